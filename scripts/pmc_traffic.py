#!/usr/bin/env python3
"""HBM traffic per launch of every kernel from the PMC passes of scripts/pmc_passes.sh (tcc1 = FETCH_SIZE, tcc2 = WRITE_SIZE):
   RT_HIP_SLOTS=1 scripts/pmc_passes.sh <tag> "tcc1 tcc2"  &&  python scripts/pmc_traffic.py gpurun_out/pmc_<tag> > profiles/r01_traffic.json
bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE counts half of a 16-B-per-lane read stream (MI355X_MICROARCH.md,
HBM section); checked on k_accumulate (its float4 reads are known exactly) and k_raygen (WRITE_SIZE = 56 B x paths + 16 B x paths)."""
import csv, io, json, os, re, subprocess, sys
out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_summary.py"), sys.argv[1]], capture_output=True, text=True).stdout
kernels = {}
for r in csv.DictReader(io.StringIO(out)):
    m = re.search(r"k_[a-z_]+", r["kernel"])
    if not m or not r.get("FETCH_SIZE") or not r.get("WRITE_SIZE"):
        continue
    n = int(r["dispatches"])
    f, w = float(r["FETCH_SIZE"]) * 1024 / n, float(r["WRITE_SIZE"]) * 1024 / n
    kernels[m.group(0)] = {"launches_profiled": n, "fetch_size_bytes_per_launch": f, "write_size_bytes_per_launch": w, "hbm_bytes_per_launch": 2 * f + w}
json.dump({"_how": __doc__.strip(), "workload": "Night Spheres 1920x1080, 64 passes per batch, depth 10, 2 light samples", "kernels": kernels}, sys.stdout, indent=1)
