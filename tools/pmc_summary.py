#!/usr/bin/env python
"""profiles/rNN/pmc_summary.json from rocprofv3's sqlite output of three runs of `bench.py --calib`:
a FETCH_SIZE pass, a WRITE_SIZE pass (each `--pmc X --kernel-trace` only) and a plain `--kernel-trace` run for durations.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE count in KiB; on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x.  Instead of assuming the factor, both counters are calibrated on a kernel
with known traffic captured in the same run: k_stream_copy (ippm_stream_copy), which reads and writes exactly the local-map
tensor with 16 B per lane.

    python tools/pmc_summary.py <fetch.db> <write.db> <trace.db> <n_envs> <n_agents> <grid> [out.json [envs_per_launch]]

envs_per_launch (default n_envs): bench.py --streams P steps the batch as P sub-batches, one launch per sub-batch and kernel; the
calibration copy moves one sub-batch's local maps, and every per-launch figure below is a sub-batch's.
"""
import json
import sqlite3
import sys

import pandas as pd


def segment(names):
    """(lo, hi) of the rows that belong to the measured stretch of a `bench.py --calib` process, given the kernel names in launch
    order: behind the three marker copies in front of the loops and, when the process has one, behind the TWO marker copies in
    front of the roofline leg (bench.py --streams P > 1: the loops before it step sub-batches side by side, the leg steps the whole
    batch per launch) -- up to the next copy (the copy-rate yardstick)."""
    is_copy = [("k_stream_copy" in n) for n in names]
    n = len(names)
    lo = next((i + 3 for i in range(n - 2) if is_copy[i] and is_copy[i + 1] and is_copy[i + 2]), None)
    if lo is None:
        return 0, n
    hi = next((i for i in range(lo, n) if is_copy[i]), n)
    if hi + 1 < n and is_copy[hi + 1] and not (hi + 2 < n and is_copy[hi + 2]):     # exactly two copies: the roofline leg follows
        lo = hi + 2
        hi = next((i for i in range(lo, n) if is_copy[i]), n)
    return lo, hi


def counters(path, counter):
    df = pd.read_sql("select kernel_name, dispatch_id, counter_name, value from counters_collection", sqlite3.connect(path))
    df = df[df["counter_name"] == counter]
    per_dispatch = df.groupby(["dispatch_id", "kernel_name"])["value"].sum().reset_index().sort_values("dispatch_id").reset_index(drop=True)
    lo, hi = segment(list(per_dispatch["kernel_name"]))
    copies = per_dispatch[per_dispatch["kernel_name"].str.contains("k_stream_copy")]
    part = pd.concat([per_dispatch.iloc[lo:hi], copies])
    return part.groupby("kernel_name")["value"].agg(["mean", "count", "max"])


def durations(path):
    df = pd.read_sql("select * from kernels", sqlite3.connect(path))
    name = [c for c in df.columns if c in ("name", "kernel_name")][0]
    df = df.sort_values("start").reset_index(drop=True)
    df["us"] = (df["end"] - df["start"]) / 1e3
    lo, hi = segment(list(df[name]))
    part = pd.concat([df.iloc[lo:hi], df[df[name].str.contains("k_stream_copy")]])
    return part.rename(columns={name: "kernel"})[["kernel", "us"]]


def pick(df, pat):
    rows = df[df.index.str.contains(pat, regex=False)]
    return rows.sort_values("count", ascending=False).iloc[0] if len(rows) else None


def main():
    fetch_db, write_db, trace_db, n_envs, n_agents, grid = sys.argv[1:7]
    out_path = sys.argv[7] if len(sys.argv) > 7 else "pmc_summary.json"
    n_envs, n_agents, grid = int(n_envs), int(n_agents), int(grid)
    per_launch = int(sys.argv[8]) if len(sys.argv) > 8 else n_envs
    copy_bytes = per_launch * n_agents * grid * grid * 4   # read once and written once by k_stream_copy
    fetch, write, trace = counters(fetch_db, "FETCH_SIZE"), counters(write_db, "WRITE_SIZE"), durations(trace_db)
    cal_f, cal_w = pick(fetch, "k_stream_copy"), pick(write, "k_stream_copy")
    f_scale = copy_bytes / (cal_f["mean"] * 1024.0)  # true bytes per reported KiB
    w_scale = copy_bytes / (cal_w["mean"] * 1024.0)
    summary = {"envs_per_gpu": n_envs, "envs_per_launch": per_launch, "n_agents": n_agents, "grid": grid, "calibration": {
        "kernel": "k_stream_copy of the local maps (ippm_stream_copy, 16 B per lane)", "bytes_each_way": copy_bytes,
        "FETCH_SIZE_reported_KiB": float(cal_f["mean"]), "WRITE_SIZE_reported_KiB": float(cal_w["mean"]),
        "fetch_correction": f_scale, "write_correction": w_scale}}
    # (prefixes: the most frequent instantiation of each kernel is the step's -- <6, ..> at 4 UAVs, <10, ..> at 8, <18, ..> at 16)
    for key, pat in (("k_sense_update", "k_sense_tiles<"), ("k_fuse_tiles", "k_fuse_tiles<"), ("k_fuse_rows", "k_fuse_rows<"),
                     ("k_plan_step", "k_plan_step"), ("k_reset_maps", "k_reset_maps"), ("k_stream_copy", "k_stream_copy")):
        f, w = pick(fetch, pat), pick(write, pat)
        if f is None or w is None:
            continue
        pat = f.name            # the instantiation's full name
        w = pick(write, pat)
        t = trace[trace["kernel"].str.contains(pat, regex=False)]["us"]
        if w is None or not len(t):
            continue
        rd = float(f["mean"]) * 1024.0 * f_scale
        wr = float(w["mean"]) * 1024.0 * w_scale
        summary[key] = {"kernel": pat, "launches": int(f["count"]), "avg_us": float(t.mean()), "hbm_read_bytes_per_launch": rd,
                        "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                        "hbm_GBps": (rd + wr) / (float(t.mean()) * 1e-6) / 1e9}
    with open(out_path, "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
