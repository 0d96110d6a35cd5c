"""Synthetic random-field terrain, generated on the GPU for a whole batch of episodes.

The reference synthesises a power-law Gaussian random field for every episode (white noise -> FFT -> sqrt(P(k)) ->
inverse FFT -> min/max normalise -> ``>= 0.5``; ``mapping/ground_truths.py:16-40`` with
``P(k) = k**-cluster_radius`` from ``mapping/simulations.py:34-40``) and then overwrites it with the half-plane split
it actually flies over.  ``VecEnv.reset(..., terrain="random_field")`` flies over that field instead: it is the
"synthetic random-field terrain" of the benchmark.  The device noise is Philox-keyed by (seed, episode, cell); it is
*not* NumPy's legacy normal stream, so for a given episode this field differs from the one the reference would have
discarded — tests that compare against the oracle hand the generated truth to both sides.

Device work (csrc/terrain.hip): power-of-two grids draw the half spectrum directly and invert it in two hand-written
LDS passes — rocFFT's batched 2-D real transforms were 4x slower on this shape — the second of them run twice, for the
field's min / max and for its threshold bits, so that the field is never stored (``ippm_terrain_truth``;
``ippm_terrain_field`` + ``ippm_terrain_pack`` are the same arithmetic with the field written out); other grid sizes
(the default 493 x 493) use ``ippm_terrain_noise`` + rocFFT via torch.fft and ``ippm_terrain_pack``.
"""
from typing import Optional

import numpy as np
import torch

from . import _ffi


def fft_index_row(n: int) -> np.ndarray:
    """Wave numbers in the order the reference enumerates them (ground_truths.py:9-13).  For odd n the list is one
    short, which leaves the last amplitude row/column at zero; that is kept."""
    half = n // 2
    return np.array(list(range(0, half + 1)) + [-i for i in reversed(range(1, half))], dtype=np.float64)


def amplitude_table(gx: int, gy: int, cluster_radius: float) -> np.ndarray:
    """sqrt(P(|k|)) with P(k) = k**-cluster_radius and P(0) = 0, float64 [gx, gy]."""
    kx, ky = fft_index_row(gx), fft_index_row(gy)
    k = np.sqrt(kx[:, None] ** 2 + ky[None, :] ** 2)
    amp = np.zeros((gx, gy))
    sub = np.zeros_like(k)
    sub[k > 0] = np.sqrt(k[k > 0] ** (-float(cluster_radius)))
    amp[: len(kx), : len(ky)] = sub
    return amp


class RandomFieldTerrain:
    def __init__(self, derived, ctx: "_ffi.Context", device: torch.device, cluster_radius: float, chunk_bytes: int = 1 << 30):
        self.d, self.ctx, self.device = derived, ctx, device
        gx, gy = derived.grid_x, derived.grid_y
        amp = amplitude_table(gx, gy, cluster_radius)
        self.real_fft = gx % 2 == 0 and gy % 2 == 0   # the odd-size quirk breaks the Hermitian symmetry of the table
        if self.real_fft:
            amp = amp[:, : gy // 2 + 1]
        self.amp = torch.from_numpy(np.ascontiguousarray(amp).astype(np.float32)).to(device)
        # the hand-written passes exist for 128 / 256 / 512 / 1024 cells a side (terrain_pow2_check in csrc/terrain.hip); every other
        # size, smaller powers of two included, draws the spectrum with ippm_terrain_noise and inverts it with rocFFT
        self.native = all(n in (128, 256, 512, 1024) for n in (gx, gy))
        self.chunk = max(1, chunk_bytes // (gx * gy * 4))
        self._noise: Optional[torch.Tensor] = None
        self._work: Optional[torch.Tensor] = None

    def generate(self, episode: torch.Tensor, truth: torch.Tensor, stream: int) -> None:
        """Writes the packed truth of ``episode[e]`` into ``truth[e]`` (both device tensors)."""
        gx, gy = self.d.grid_x, self.d.grid_y
        E = episode.numel()
        for lo in range(0, E, self.chunk):
            n = min(self.chunk, E - lo)
            ep, tr = episode[lo:lo + n], truth[lo:lo + n]
            if self.native:
                if self._work is None or self._work.shape[0] < n:
                    self._work = torch.empty(n, gy // 2 + 1, gx, 2, dtype=torch.float32, device=self.device)
                    self._keys = torch.zeros(4 * n + 4, dtype=torch.int32, device=self.device)
                # (the field itself is never stored: second pass once for its min / max, once more for the threshold bits;
                #  _keys = per env (min, max, and for IPPM_TERRAIN_ONE_LAUNCH=1: arrivals, fault word), then that form's ticket counter)
                self.ctx.call("ippm_terrain_truth", _ffi.ptr(ep), _ffi.ptr(self.amp), _ffi.ptr(self._work), _ffi.ptr(self._keys), _ffi.ptr(tr),
                              n, stream)
                continue
            if self._noise is None or self._noise.shape[0] < n:
                self._noise = torch.empty(n, gx, gy, dtype=torch.float32, device=self.device)
            noise = self._noise[:n]
            self.ctx.call("ippm_terrain_noise", _ffi.ptr(ep), _ffi.ptr(noise), n, stream)
            if self.real_fft:
                field = torch.fft.irfft2(torch.fft.rfft2(noise) * self.amp, s=(gx, gy))
            else:
                field = torch.fft.ifft2(torch.fft.fft2(noise) * self.amp).real
            field = field.contiguous()
            self.ctx.call("ippm_terrain_pack", _ffi.ptr(field), None, _ffi.ptr(tr), n, stream)
