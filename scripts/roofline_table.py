#!/usr/bin/env python
"""Collect the committed bench lines (profiles/*.json) into one table: workload, ms/step, samples/s, algorithmic
bytes per step, fraction of the 8 TB/s HBM figure and of the device-copy bandwidth measured in the same run, the
dominant kernel and its own fraction.  Usage: python scripts/roofline_table.py > profiles/r01_roofline_table.md"""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*.json"))):
    try:
        d = json.load(open(p))
    except Exception:
        continue
    if not isinstance(d, dict) or "ms_per_step" not in d:
        continue
    c, r, s = d.get("config", {}), d.get("roofline", {}), d.get("step_roofline", {})
    rows.append((os.path.basename(p), c.get("workload", "?"), c.get("real_tensor_io", "f32"), c.get("engine_path", "?"),
                 d["n_gpus"], d["ms_per_step"], d.get("cold_start", {}).get("ms_per_step"), d["value"], s.get("alg_bytes_per_step", 0) / 1e6,
                 s.get("frac_of_8TBs"), s.get("frac_of_measured_copy"), r.get("kernel", "?"),
                 r.get("ms_per_launch"), r.get("frac"), r.get("frac_of_measured_copy")))
print("| file | workload | real I/O | path | GPUs | ms/step | cold-start ms/step | samples/s | BYTES_ALG (MB) | step / 8 TB/s | step / measured copy |"
      " dominant kernel | ms/launch | kernel / 8 TB/s | kernel / measured copy |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
fmt = lambda v, n=3: "—" if v is None else f"{v:.{n}f}"
for (f, w, io, path, n, ms, cold, val, mb, f8, fc, k, kms, kf8, kfc) in rows:
    print(f"| `{f}` | {w} | {io} | {path} | {n} | {ms:.4f} | {fmt(cold, 4)} | {val:,.0f} | {mb:,.1f} | {fmt(f8)} | {fmt(fc)} | `{k}` |"
          f" {fmt(kms, 4)} | {fmt(kf8)} | {fmt(kfc)} |")
