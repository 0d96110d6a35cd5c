#!/usr/bin/env python
"""Per-(kernel, grid size) duration table from rocprofv3's rocpd sqlite output: python tools/trace_summary.py <db> [min_us]"""
import sqlite3
import sys

import pandas as pd

pd.set_option("display.width", 250)
db = sqlite3.connect(sys.argv[1])
df = pd.read_sql("select * from kernels", db)
name = [c for c in df.columns if c in ("name", "kernel_name")][0]
df["us"] = (df["end"] - df["start"]) / 1e3
grid = "grid_size" if "grid_size" in df.columns else [c for c in df.columns if "grid" in c][0]
df["k"] = df[name].str.replace("void ", "").str.slice(0, 44)
g = df.groupby(["k", grid])["us"].agg(["count", "mean", "min", "max", "sum"]).sort_values("sum", ascending=False)
print(g[g["sum"] > (float(sys.argv[2]) if len(sys.argv) > 2 else 50)].round(1).to_string())

# (bench.py --calib brackets its loops between three k_stream_copy launches and the copy-rate yardstick at the end: only what
#  lies between them is the bench loop -- the placement search runs before, the COMA leg after)
copies = df.sort_values("start").reset_index(drop=True)
is_copy = copies["k"].str.contains("k_stream_copy").to_numpy()
lo = next((i + 3 for i in range(len(copies) - 2) if is_copy[i] and is_copy[i + 1] and is_copy[i + 2]), None)
if lo is not None:
    hi = next((i for i in range(lo, len(copies)) if is_copy[i]), len(copies))
    df = copies.iloc[lo:hi].copy()
# idle time in front of each kernel of the env step (end of the previous kernel on the device -> start of this one)
# (the three launches of a step: the plan kernel, the fusion in either of its forms, K3 in either of its forms)
step = df[df["k"].str.contains("k_plan_step|k_sense_tiles|k_sense_update|k_fuse_tiles|k_fuse_rows")].sort_values("start").reset_index(drop=True)
if len(step) > 12:
    step["gap_us"] = (step["start"] - step["end"].shift(1)) / 1e3
    tail = step.iloc[len(step) // 2:]
    print("\nidle time before each step kernel (second half of the bench loop):")
    print(tail.groupby("k")["gap_us"].agg(["count", "mean", "min", "max"]).round(1).to_string())
    span = (tail["end"].iloc[-1] - tail["start"].iloc[0]) / 1e3
    kinds = tail["k"].str.extract(r"(k_plan_step|k_sense|k_fuse)")[0]
    triples = int(kinds.value_counts().min())     # complete steps in the window
    assert kinds.nunique() == 3, "the window does not hold all three kernels of a step: %s" % sorted(kinds.dropna().unique())
    print("span per step (plan + fusion + K3; resets included in the span): %.1f us, of which these three kernels %.1f us"
          % (span / triples, tail["us"].sum() / triples))
