#!/usr/bin/env python
"""Per-kernel means of the counters in rocprofv3's rocpd sqlite output (p_results.db): python tools/pmc_db_summary.py <db>..."""
import sqlite3
import sys

import pandas as pd

pd.set_option("display.width", 250)
pd.set_option("display.max_columns", 40)
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    df = pd.read_sql("select kernel_name, grid_size, dispatch_id, counter_name, value from counters_collection", db)
    df["k"] = df["kernel_name"].str.replace("void ", "").str.slice(0, 26) + " g" + df["grid_size"].astype(str)
    g = df.groupby(["k", "counter_name"])["value"].sum() / df.groupby(["k", "counter_name"])["dispatch_id"].nunique()
    t = g.unstack()
    try:   # the same pass's kernel trace: avg duration per (kernel, grid) -- GRBM_GUI_ACTIVE / us = the clock the kernel ran at
        kd = pd.read_sql("select * from kernels", db)
        name = [c for c in kd.columns if c in ("name", "kernel_name")][0]
        gcol = [c for c in kd.columns if c in ("grid_size", "grid_size_x", "grid_x")]
        kd["k"] = kd[name].str.replace("void ", "").str.slice(0, 26) + (" g" + kd[gcol[0]].astype(str) if gcol and gcol[0] == "grid_size" else "")
        if gcol and gcol[0] == "grid_size":
            t["avg_us"] = ((kd["end"] - kd["start"]) / 1e3).groupby(kd["k"]).mean()
    except Exception as exc:
        print("(no durations:", exc, ")")
    keep = [k for k in t.index if k.startswith("k_")]
    print(path)
    print(t.loc[keep].round(0).to_string())
