#!/usr/bin/env python
"""Sets rocprofv3's memory-side read counters against the known byte / line counts of tools/probe/fetch_calib.cpp.
    python tools/fetch_calib_summary.py <probe stdout> <counter csv> [<counter csv> ...]  > profiles/r06/fetch_calibration.json"""
import json
import sys

import pandas as pd

known = {}
for line in open(sys.argv[1]):
    if line.startswith("{"):
        r = json.loads(line)
        known[r["kernel"]] = r
frames = [pd.read_csv(p) for p in sys.argv[2:]]
c = pd.concat(frames)
c["kernel"] = c["Kernel_Name"].str.replace("void ", "", regex=False).str.replace(r"\(.*", "", regex=True)
per = c.groupby(["kernel", "Counter_Name", "Dispatch_Id"])["Counter_Value"].sum().reset_index()
mean = per.groupby(["kernel", "Counter_Name"])["Counter_Value"].mean().unstack()
out = {"what": "rocprofv3 memory-side read counters against known bytes of read-only kernels (tools/probe/fetch_calib.cpp): k_stream = full-line "
               "streaming, k_rows<START> = 90-row footprints of 23 x 16 B per row starting START bytes into a 128-byte line (K3's shape)",
       "kernels": []}
for k, kn in known.items():
    if k not in mean.index:
        continue
    row = {n: (None if pd.isna(v) else float(v)) for n, v in mean.loc[k].items()}
    rec = dict(kn)
    rec["counters_per_launch"] = row
    if row.get("FETCH_SIZE"):
        fb = row["FETCH_SIZE"] * 1024.0
        rec["FETCH_SIZE_bytes"] = fb
        rec["requested_over_FETCH"] = kn["requested_bytes"] / fb
        rec["line128_over_FETCH"] = kn["line128_bytes"] / fb
        rec["half64_over_FETCH"] = kn["half64_bytes"] / fb
        rec["sector32_over_FETCH"] = kn["sector32_bytes"] / fb
    if row.get("TCC_EA0_RDREQ_sum"):
        rq, r32, bub = row["TCC_EA0_RDREQ_sum"], row.get("TCC_EA0_RDREQ_32B_sum") or 0.0, row.get("TCC_BUBBLE_sum") or 0.0
        rec["requests"] = {"all": rq, "of_32B": r32, "bubble_128B": bub, "line128_per_request": kn["line128_bytes"] / 128.0 / rq,
                           "half64_per_request": kn["half64_bytes"] / 64.0 / rq, "sector32_per_request": kn["sector32_bytes"] / 32.0 / rq}
    out["kernels"].append(rec)
s = next((r for r in out["kernels"] if r["kernel"] == "k_stream" and "FETCH_SIZE_bytes" in r), None)
if s:
    corr = s["requested_bytes"] / s["FETCH_SIZE_bytes"]
    out["stream_correction"] = corr
    for r in out["kernels"]:
        if "FETCH_SIZE_bytes" in r:
            r["corrected_FETCH_over_line128"] = r["FETCH_SIZE_bytes"] * corr / r["line128_bytes"]
            r["corrected_FETCH_over_requested"] = r["FETCH_SIZE_bytes"] * corr / r["requested_bytes"]
print(json.dumps(out, indent=1))
