"""Actor / critic convnets (PyTorch-ROCm; the only place MFMA is used on this path).

Same architecture, layer names and initialisation order as the reference (actor/network.py:10-96,
critic/network.py:12-47) so that a reference ``state_dict`` / ``best_model.pth`` loads unchanged: conv 5x5 -> 4x4 -> 4x4
(11 -> 7 -> 4 -> 1), fc1, (fc2: present but unused, exactly like the reference), fc3.  Inputs are channels-last
[B,11,11,C] as produced by the feature kernels.  The reference's global ``torch.autograd.set_detect_anomaly(True)``
(critic/network.py:9) is deliberately not reproduced.
"""
from __future__ import annotations

import os
from typing import Dict

import torch
from torch import nn

# MIOpen's exhaustive solver search (needed: its fast mode picks kernels that make a COMA update 6x slower) also times the
# naive reference convolutions, which take 0.3-0.5 s per run on the 12k-sample minibatches: 40 s of warm-up per process
# for nothing.  Leave them out of the search unless the user says otherwise.
for _v in ("FWD", "BWD", "WRW"):
    os.environ.setdefault(f"MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_{_v}", "0")


# conv3 as a GEMM (see _ConvTrunk.trunk); IPPMARL_CONV3_GEMM=0 keeps the convolution call
CONV3_AS_GEMM = os.environ.get("IPPMARL_CONV3_GEMM", "1") != "0"
# bias + ReLU after conv1 / conv2 as one pass (see _BiasReLU); IPPMARL_FUSED_BIAS_RELU=0 keeps PyTorch's separate passes
FUSED_BIAS_RELU = os.environ.get("IPPMARL_FUSED_BIAS_RELU", "1") != "0"
# conv2's input gradient as GEMM + col2im (see _ConvDataGradAsGemm); IPPMARL_CONV2_BWD_GEMM=0 keeps the library's kernel
CONV2_BWD_DATA_AS_GEMM = os.environ.get("IPPMARL_CONV2_BWD_GEMM", "1") != "0"


def epsilon_schedule(params: Dict, num_episode: int) -> float:
    """Linear anneal eps_max -> eps_min over eps_anneal_phase episodes (actor/network.py:53-58)."""
    m = params["experiment"]["missions"]
    if num_episode > m["eps_anneal_phase"]:
        return m["eps_min"]
    return m["eps_max"] - num_episode / m["eps_anneal_phase"] * (m["eps_max"] - m["eps_min"])


# ActorNetwork.get_action_index: the team's action probabilities of the sweep in progress, per network object (kept out of the
# module's __dict__: whole-module pickles are the reference's checkpoint format)
import weakref  # noqa: E402

_TEAM_PROBS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


class _ConvDataGradAsGemm(torch.autograd.Function):
    """conv2d whose INPUT gradient is a GEMM + col2im instead of the library's implicit-GEMM backward-data kernel.

    Measured on MI355X (profiles/r02/coma_update_flops.json): MIOpen's float32 forward and weight-gradient kernels run this
    4x4 convolution at 132 / 136 TFLOP/s (84-87 % of the float32 matrix peak), its backward-data kernel at 48 -- 44 % of a
    COMA update's kernel time.  The input gradient of a stride-1 convolution is  fold(grad_out[B*L, O] x W[O, C*kh*kw])  (L =
    output positions): one hipBLASLt GEMM at the forward's rate plus a memory-bound scatter.  Forward, weight and bias
    gradients stay with the library."""

    COLS_LIMIT = 3 << 30   # bytes of the column matrix of one slice (below the 4 GB at which 32-bit byte offsets wrap)

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.conv2d(x, weight, bias)

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        grad_x = grad_w = grad_b = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            _, grad_w, grad_b = torch.ops.aten.convolution_backward(
                grad_out, x, weight, [weight.shape[0]], [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                [False, True, bool(ctx.needs_input_grad[2])])
            if not ctx.needs_input_grad[2]:
                grad_b = None
        if ctx.needs_input_grad[0]:
            B, O, Ho, Wo = grad_out.shape
            C, K = weight.shape[1], weight.shape[2]
            g = grad_out.permute(0, 2, 3, 1).reshape(B * Ho * Wo, O)                  # [B*L, O] (a view for channels-last)
            if grad_out.is_cuda and C % 4 == 0 and weight.shape[2] == weight.shape[3]:
                # tap-major, channel-minor columns: the gather kernel of libippmarl reads and writes channels-last rows.
                # The column matrix is K*K times the size of grad_out (3.2 GB for a 12 288-sample minibatch of conv2): batches
                # whose columns would pass COLS_LIMIT bytes go through in slices of the batch
                from . import _ffi
                lib, stream = _ffi.load_library(), torch.cuda.current_stream(grad_out.device).cuda_stream
                w2 = weight.permute(0, 2, 3, 1).reshape(O, -1)                        # [O, K*K*C]
                grad_x = torch.empty(B, Ho + K - 1, Wo + K - 1, C, dtype=grad_out.dtype, device=grad_out.device)
                g = g.contiguous()
                per_sample = Ho * Wo * K * K * C * g.element_size()
                step = max(1, min(B, _ConvDataGradAsGemm.COLS_LIMIT // per_sample))
                for lo in range(0, B, step):
                    nb = min(step, B - lo)
                    cols = g[lo * Ho * Wo:(lo + nb) * Ho * Wo] @ w2                   # [nb*L, K*K*C]
                    _ffi.check(lib.ippm_col2im_nhwc(_ffi.ptr(cols), grad_x[lo:lo + nb].data_ptr(), nb, Ho, Wo, K, C, stream), "ippm_col2im_nhwc")
                grad_x = grad_x.permute(0, 3, 1, 2)                                   # logical NCHW, channels-last strides
            else:
                cols = g @ weight.reshape(O, -1)                                       # [B*L, C*kh*kw]
                cols = cols.view(B, Ho * Wo, -1).transpose(1, 2)                       # [B, C*kh*kw, L]
                grad_x = torch.nn.functional.fold(cols, x.shape[-2:], weight.shape[-2:])
        return grad_x, grad_w, grad_b


class _BiasReLU(torch.autograd.Function):
    """relu(x + bias) for a channels-last activation tensor as ONE pass in each direction (libippmarl's ippm_bias_relu_nhwc
    / ippm_bias_relu_backward_nhwc) instead of the bias pass + ReLU pass (and threshold pass + column reduction) PyTorch runs
    around a MIOpen convolution.  ``x`` is the bias-free convolution output and is overwritten."""

    @staticmethod
    def usable(x: torch.Tensor, bias) -> bool:
        c = x.shape[1]
        return (FUSED_BIAS_RELU and bias is not None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and c % 4 == 0
                and 256 % (c // 4) == 0 and x.is_contiguous(memory_format=torch.channels_last))

    @staticmethod
    def _lib():
        from . import _ffi
        return _ffi, _ffi.load_library()

    @staticmethod
    def forward(ctx, x, bias):
        ffi, lib = _BiasReLU._lib()
        b = bias.detach().contiguous()
        # (channels-last storage: dense [rows, C]; `usable` has checked that)
        ffi.check(lib.ippm_bias_relu_nhwc(x.data_ptr(), ffi.ptr(b), x.numel() // x.shape[1], x.shape[1],
                                          torch.cuda.current_stream(x.device).cuda_stream), "ippm_bias_relu_nhwc")
        ctx.mark_dirty(x)
        ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, grad_y):
        (y,) = ctx.saved_tensors
        ffi, lib = _BiasReLU._lib()
        g = grad_y.contiguous(memory_format=torch.channels_last)
        grad_x = torch.empty_like(y)                      # channels-last like y
        grad_b = torch.zeros(y.shape[1], dtype=y.dtype, device=y.device)
        ffi.check(lib.ippm_bias_relu_backward_nhwc(g.data_ptr(), y.data_ptr(), grad_x.data_ptr(), ffi.ptr(grad_b),
                                                   y.numel() // y.shape[1], y.shape[1],
                                                   torch.cuda.current_stream(y.device).cuda_stream), "ippm_bias_relu_backward_nhwc")
        return grad_x, grad_b


class _ConvTrunk(nn.Module):
    def __init__(self, in_planes: int, n_actions: int):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, 256, (5, 5))
        self.conv2 = nn.Conv2d(256, 256, (4, 4))
        self.conv3 = nn.Conv2d(256, 256, (4, 4))
        self.activation = nn.ReLU()
        self.flatten = nn.Flatten()
        self.fc1 = nn.Linear(256, 256)
        self.fc2 = nn.Linear(256, 256)  # never used in forward (reference: commented out) -> never gets a gradient
        self.fc3 = nn.Linear(256, n_actions)

    def _conv_relu(self, conv: nn.Conv2d, x: torch.Tensor, data_grad_as_gemm: bool):
        """activation(conv(x)); on the device the convolution runs bias-free and bias + ReLU are one in-place pass."""
        # (both paths below, and _ConvDataGradAsGemm's backward, are written for the unpadded stride-1 layers of the reference)
        assert tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (0, 0) and tuple(conv.dilation) == (1, 1) and conv.groups == 1
        conv2d = _ConvDataGradAsGemm.apply if data_grad_as_gemm else torch.nn.functional.conv2d
        if x.is_cuda and FUSED_BIAS_RELU:
            out = conv2d(x, conv.weight, None)
            if _BiasReLU.usable(out, conv.bias):
                return _BiasReLU.apply(out, conv.bias)
            return self.activation(out + conv.bias.view(1, -1, 1, 1))
        return self.activation(conv2d(x, conv.weight, conv.bias))

    # conv1's activation [B, 256, 7, 7] float32 is 50 176 bytes per sample: 2^31 bytes at 42 799 samples, 2^32 at 85 598, and
    # the library's convolution kernels address their tensors with 32-bit byte offsets (measured: one forward pass over 131 072
    # critic states returns the results of samples 85 598.. wrapped onto samples 0..: wrong Q values, no error).  Larger batches go
    # through in slices.
    MAX_SAMPLES = 32768

    def trunk(self, x: torch.Tensor):
        if x.dim() == 3:
            x = x.unsqueeze(0)
        if x.shape[0] > self.MAX_SAMPLES:
            parts = [self.trunk(x[lo:lo + self.MAX_SAMPLES]) for lo in range(0, x.shape[0], self.MAX_SAMPLES)]
            return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
        x = x.permute(0, 3, 1, 2)  # NHWC storage -> logical NCHW (channels_last strides, no copy)
        h = self._conv_relu(self.conv1, x, False)
        h = self._conv_relu(self.conv2, h, CONV2_BWD_DATA_AS_GEMM and h.requires_grad)
        if CONV3_AS_GEMM and h.shape[-2:] == self.conv3.kernel_size:
            # conv3 sees a 4x4 map with a 4x4 kernel: ONE output position, i.e. a plain [B, 4096] x [4096, 256] product.
            # As a GEMM it goes to hipBLASLt instead of an implicit-GEMM convolution with a degenerate output tile.
            # (h is channels-last: flattening it in (H, W, C) order is a view; the weight is permuted to match.)
            w3 = self.conv3.weight.permute(0, 2, 3, 1).reshape(self.conv3.out_channels, -1)
            h = self.activation(torch.nn.functional.linear(h.permute(0, 2, 3, 1).reshape(h.shape[0], -1), w3, self.conv3.bias))
        else:
            h = self.flatten(self.activation(self.conv3(h)))
        return self.fc3(self.activation(self.fc1(h))), h


class ActorNetwork(_ConvTrunk):
    def __init__(self, params: Dict):
        self.params = params
        self.n_agents = params["experiment"]["missions"]["n_agents"]
        self.n_actions = params["experiment"]["constraints"]["num_actions"]
        super().__init__(7, self.n_actions)
        self.softmax = nn.Softmax(dim=1)
        # The plain attributes of the reference's class (actor/network.py:13-39).  A whole-module pickle of this object is
        # unpickled by the reference WITHOUT running its __init__ (checkpoint.save_actor), so everything its methods read --
        # get_action_index uses device and the epsilon schedule -- has to travel in our __dict__.
        m = params["experiment"]["missions"]
        self.mission_type = m["type"]
        self.hidden_dim = params["networks"]["actor"]["hidden_dim"]
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.log_softmax = nn.LogSoftmax(dim=1)
        self.hidden_states = [[]] * self.n_agents
        self.eps_max, self.eps_min, self.eps_anneal_phase, self.use_eps = m["eps_max"], m["eps_min"], m["eps_anneal_phase"], m["use_eps"]
        self.network_type = params["networks"]["type"]
        self.baseline = "no"

    def forward(self, input_state: torch.Tensor, eps: float):
        """-> ((1-eps) * softmax + eps / n_actions, hidden) (actor/network.py:70-88)."""
        logits, h = self.trunk(input_state)
        probs = self.softmax(logits)
        return (1 - eps) * probs + eps / self.n_actions, h

    def eps(self, num_episode: int) -> float:
        return epsilon_schedule(self.params, num_episode)

    def get_action_index(self, batch_memory, action_mask_1d, agent_id, t, num_episode: int, mode: str):
        """Single-agent action choice of the drop-in Agent.step (actor/network.py:41-68,90-96): masked eps-softmax,
        torch.multinomial in training, argmax in evaluation.
        The agents of a team are served one after the other within a step (coma_wrapper.py:97-104) and their observations are all
        in the memory before the first one acts (coma_wrapper.py:37-71), so the first call of a sweep runs ONE forward pass for the
        whole team and the following agents take their row of it (a batch-1 pass per agent was a quarter of the seam's step)."""
        device = self.conv1.weight.device
        eps = epsilon_schedule(self.params, num_episode)
        team = _TEAM_PROBS.get(self)
        key = (id(batch_memory), t, eps)
        if team is None or team[0] != key or agent_id <= team[1] or agent_id >= team[2].shape[0]:
            lens = {len(batch_memory.transitions[j]) for j in range(self.n_agents)} if hasattr(batch_memory, "transitions") else set()
            if len(lens) == 1 and agent_id == 0:       # every agent holds an observation of this step: the whole team at once
                obs = torch.stack([batch_memory.get(-1, j, "observation") for j in range(self.n_agents)]).to(device).float()
            else:
                obs = batch_memory.get(-1, agent_id, "observation").unsqueeze(0).to(device).float()
            with torch.no_grad():
                probs_all, _ = self.forward(obs, eps)
            team = [key, agent_id, probs_all if obs.shape[0] > 1 else None]
            row = probs_all[agent_id if obs.shape[0] > 1 else 0]
            if team[2] is None:
                team = None
        else:
            row = team[2][agent_id]
        if team is not None:
            team[1] = agent_id
        _TEAM_PROBS[self] = team
        mask = torch.as_tensor(action_mask_1d).to(device)
        probs = row * mask
        chosen = torch.argmax(probs) if mode == "eval" else torch.multinomial(probs, 1, replacement=True)
        return probs, chosen, mask, eps


class CriticNetwork(_ConvTrunk):
    def __init__(self, params: Dict):
        self.params = params
        self.n_agents = params["experiment"]["missions"]["n_agents"]
        self.n_actions = params["experiment"]["constraints"]["num_actions"]
        super().__init__(12, self.n_actions)

    def forward(self, input_state: torch.Tensor):
        """-> (Q [B,A], log_softmax over dim 0 (a metric the reference logs; critic/network.py:43-47))."""
        q, _ = self.trunk(input_state)
        q = q.squeeze()
        with torch.no_grad():
            log_probs = torch.log_softmax(q, dim=0)
        return q, log_probs
