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
    keep = [k for k in t.index if k.startswith("k_")]
    print(path)
    print(t.loc[keep].round(0).to_string())
