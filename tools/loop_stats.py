#!/usr/bin/env python
"""Per-kernel durations of the bench LOOP from rocprofv3's per-dispatch kernel trace (csv) of `bench.py --calib`: only the
dispatches between the three k_stream_copy launches of bench.py --calib (right before its warm-up loop, i.e. after
VecEnv.tune_placement's episodes on candidate allocations) and the copy-rate yardstick that follows the roofline leg: warm-up,
timed loop and roofline leg, the launches whose durations bench.py's own roofline entries report.
    python tools/loop_stats.py <..._kernel_trace.csv> [out.csv]"""
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
is_copy = df["Kernel_Name"].str.contains("k_stream_copy").to_numpy()
start = next((i + 3 for i in range(len(df) - 2) if is_copy[i] and is_copy[i + 1] and is_copy[i + 2]), None)
if start is None:
    raise SystemExit("no run of three k_stream_copy launches in the trace (run bench.py with --calib)")
end = next((i for i in range(start, len(df)) if is_copy[i]), len(df))   # bench.py's copy-rate yardstick follows the roofline leg
loop = df.iloc[start:end].copy()
loop["us"] = (loop["End_Timestamp"] - loop["Start_Timestamp"]) / 1e3
loop["Name"] = loop["Kernel_Name"].str.replace("void ", "", regex=False)
g = loop.groupby("Name")["us"].agg(Calls="count", TotalUs="sum", AverageUs="mean", MinUs="min", MaxUs="max").sort_values("TotalUs", ascending=False)
g["Percentage"] = 100.0 * g["TotalUs"] / g["TotalUs"].sum()
out = g.round(3).reset_index()
if len(sys.argv) > 2:
    out.to_csv(sys.argv[2], index=False)
pd.set_option("display.width", 250)
pd.set_option("display.max_colwidth", 90)
print(f"{len(loop)} dispatches after the marker (of {len(df)} in the process)")
print(out.head(14).to_string(index=False))
