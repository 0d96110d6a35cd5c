#!/usr/bin/env python
"""FLOP accounting of the COMA learning path from a rocprofv3 kernel trace of tools/train_profile.py:

    rocprofv3 --kernel-trace -d DIR -o t -- python tools/train_profile.py      (ENVS=1024)
    python tools/coma_flops.py DIR/.../t_results.db 1024 > profiles/rNN/coma_update_flops.json

Only the script's TIMED round counts (everything from its second episode reset on: the warm-up round before it is where
MIOpen's exhaustive search tries dozens of candidate kernels).  That round's work is known exactly (1 rollout, 1 TD-target
pass, 1 update of data_passes x batch_number minibatches), so the analytic FLOPs of every convolution / GEMM class are divided
by the summed duration of that class's kernels and compared with the MI355X float32 matrix peak (157.3 TFLOP/s,
MI355X_MICROARCH.md)."""
import json
import sqlite3
import sys

import pandas as pd

FP32_MATRIX_PEAK_TFLOPS = 157.3
db, E = sys.argv[1], int(sys.argv[2])
N, T, A = 4, 15, 6
DATA_PASSES, BATCHES = 5, 5
n = E * N * T                      # transitions per update
bs = n // BATCHES


def layer_macs(c_in):              # multiply-accumulates per sample
    return {"conv1": 49 * c_in * 25 * 256, "conv2": 16 * 256 * 16 * 256, "conv3": 256 * 16 * 256, "fc": 256 * 256 + 256 * A}


actor, critic = layer_macs(7), layer_macs(12)
# forward-only samples: rollouts (actor), TD targets (target critic), post-step Q (critic); forward+backward samples: minibatches
fwd_only = {"actor": T * E * N, "critic": n + DATA_PASSES * BATCHES * bs}
fwd_bwd = {"actor": DATA_PASSES * BATCHES * bs, "critic": DATA_PASSES * BATCHES * bs}
flops = {"conv_fwd": 0.0, "conv_bwd_data": 0.0, "conv_wrw": 0.0, "gemm": 0.0}
for net, macs in (("actor", actor), ("critic", critic)):
    s_f, s_b = fwd_only[net] + fwd_bwd[net], fwd_bwd[net]
    conv = macs["conv1"] + macs["conv2"] + macs["conv3"]
    # ippmarl.networks runs conv3 (4x4 kernel on a 4x4 input = one dot product per output channel) as a plain GEMM in all
    # three directions and conv2's input gradient as a GEMM + col2im, so those FLOPs are done by hipBLASLt kernels
    conv_miopen = macs["conv1"] + macs["conv2"]
    flops["conv_fwd"] += 2.0 * s_f * conv_miopen
    flops["conv_wrw"] += 2.0 * s_b * conv_miopen
    flops["gemm"] += 2.0 * macs["fc"] * (s_f + 2 * s_b) + 2.0 * macs["conv3"] * (s_f + 2 * s_b) + 2.0 * s_b * macs["conv2"]

con = sqlite3.connect(db)
k = pd.read_sql("select * from kernels", con)
name = [c for c in k.columns if c in ("name", "kernel_name")][0]
k["us"] = (k["end"] - k["start"]) / 1e3
resets = k[k[name].str.contains("k_reset_scalars")].sort_values("start")
k = k[k["start"] >= resets["start"].iloc[-1]]   # the timed round


def klass(s):
    s = s.lower()
    if "wrw" in s or "wgrad" in s or "bwdwrw" in s:
        return "conv_wrw"
    if "bwd" in s and ("igemm" in s or "conv" in s):
        return "conv_bwd_data"
    if "fwd" in s and ("igemm" in s or "conv" in s):
        return "conv_fwd"
    if "igemm" in s or "conv" in s or "naive_conv" in s:
        return "conv_other"
    if s.startswith("cijk") or "gemm" in s:
        return "gemm"
    return None


k["class"] = k[name].map(klass)
out = {"envs": E, "transitions_per_update": n, "minibatch": bs, "fp32_matrix_peak_TFLOPs": FP32_MATRIX_PEAK_TFLOPS, "classes": {}}
for c, g in k[k["class"].notna()].groupby("class"):
    t_s = g["us"].sum() * 1e-6
    f = flops.get(c)
    out["classes"][c] = {"kernel_time_s": t_s, "launches": int(len(g)), "analytic_GFLOP": None if f is None else f / 1e9,
                         "TFLOPs": None if f is None else f / t_s / 1e12,
                         "frac_of_fp32_matrix_peak": None if f is None else f / t_s / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
                         "top_kernels": g.groupby(name)["us"].sum().sort_values(ascending=False).head(3).round(0).to_dict()}
tot_f = sum(flops.values())
tot_t = sum(v["kernel_time_s"] for v in out["classes"].values())
out["total"] = {"analytic_GFLOP": tot_f / 1e9, "kernel_time_s": tot_t, "TFLOPs": tot_f / tot_t / 1e12,
                "frac_of_fp32_matrix_peak": tot_f / tot_t / 1e12 / FP32_MATRIX_PEAK_TFLOPS}
other = k[k["class"].isna()].groupby(name)["us"].sum().sort_values(ascending=False).head(8).round(0).to_dict()
out["other_kernels_us"] = other
print(json.dumps(out, indent=1))
