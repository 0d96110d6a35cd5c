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
end = next((i for i in range(start, len(df)) if is_copy[i]), len(df))   # the next marker: the roofline leg's (--streams > 1) or the yardstick


def table(part, path):
    part = part.copy()
    part["us"] = (part["End_Timestamp"] - part["Start_Timestamp"]) / 1e3
    part["Name"] = part["Kernel_Name"].str.replace("void ", "", regex=False)
    g = part.groupby("Name")["us"].agg(Calls="count", TotalUs="sum", AverageUs="mean", MinUs="min", MaxUs="max").sort_values("TotalUs", ascending=False)
    g["Percentage"] = 100.0 * g["TotalUs"] / g["TotalUs"].sum()
    out = g.round(3).reset_index()
    if path:
        out.to_csv(path, index=False)
    print(out.head(14).to_string(index=False))


pd.set_option("display.width", 250)
pd.set_option("display.max_colwidth", 90)
loop = df.iloc[start:end]
out_path = sys.argv[2] if len(sys.argv) > 2 else None
# bench.py --streams P --calib puts TWO copies between the loops whose kernels overlap across streams (warm-up, timed region, steady-state
# leg, the overlapped_us stretch) and the roofline leg, whose sub-batches are stepped one after the other: every launch alone
two = end + 1 < len(df) and is_copy[end] and is_copy[end + 1] and not (end + 2 < len(df) and is_copy[end + 2])
print(f"{len(loop)} dispatches after the marker (of {len(df)} in the process)" + (": the loops with the streams' kernels side by side" if two else ""))
table(loop, (out_path.replace(".csv", "_overlapped.csv") if two else out_path) if out_path else None)
if two:
    leg_end = next((i for i in range(end + 2, len(df)) if is_copy[i]), len(df))
    leg = df.iloc[end + 2:leg_end]
    print(f"\n{len(leg)} dispatches of the roofline leg (every launch alone on the device)")
    table(leg, out_path)
