"""Prints the figures of a bench.py JSON line that one looks at between two kernel experiments."""
import json
import sys

try:
    lines = [ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")]
    d = json.loads(lines[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "n_gpus", "ranks", "faults", "resets_timed")})
    if d.get("steady_state"):
        print("steady_state", {k: d["steady_state"][k] for k in ("ms_per_step", "value", "steps", "resets")})
    for r in d.get("roofline_kernels") or []:
        print((r.get("kernel") or "")[:48], {k: round(r[k], 3) if isinstance(r.get(k), float) else r.get(k)
                                            for k in ("avg_launch_us", "min_launch_us", "timed_launches", "frac", "us_per_step") if k in r})
    if d.get("roofline"):
        print("whole_step", d["roofline"].get("whole_step"))
    print("coma", d.get("coma_training"))
    print("collective", d.get("collective"))
    print("dropin_seam", d.get("dropin_seam"))
    print("placement", [(p or {}).get("map_kernels_us_per_step") for p in (d.get("placement") if isinstance(d.get("placement"), list) else [d.get("placement")])])
    print("cpu_baseline", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:   # noqa: BLE001
    print("bench parse failed:", e)
