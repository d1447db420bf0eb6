#!/usr/bin/env python3
"""HBM-side traffic per launch of every kernel from the PMC passes of scripts/pmc_passes.sh (tcc1 = FETCH_SIZE, tcc2 = WRITE_SIZE):
   RT_HIP_SLOTS=1 scripts/pmc_passes.sh <tag> "tcc1 tcc2"  &&  python scripts/pmc_traffic.py gpurun_out/pmc_<tag> [workload label] > profiles/r02_traffic.json
FETCH_SIZE / WRITE_SIZE are the L2's memory-side request counters (Infinity Cache hits included: an upper bound on HBM bytes).
Calibration (MI355X_MICROARCH.md, HBM section, and tools/calib_fetch.hip -> profiles/r02_fetch_calibration.csv):
  * a coalesced 16-B-per-lane read stream is reported at HALF its bytes -> the streaming kernels (k_raygen, k_shade, k_accumulate, k_add,
    k_resolve) use bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; checked on k_accumulate (its float4 reads are known exactly) and
    k_raygen (WRITE_SIZE = 56 B x paths + 16 B x paths);
  * the traversal kernels' node fetch — every lane reads its own 64-B record as 3 x dwordx4 + dwordx2 — is reported IN FULL
    (calib_fetch: 1.048 M KiB for 1 GiB of records) -> k_closest*, k_shadow*, k_trace* use bytes = (FETCH_SIZE + WRITE_SIZE) x 1024, and
    whoever knows the ray count adds the half of their coalesced path-state reads that FETCH_SIZE left out (32 B per bounce ray,
    64 B per shadow request): scripts/run_configs.py and bench.py do."""
import csv, io, json, os, re, subprocess, sys
out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_summary.py"), sys.argv[1]], capture_output=True, text=True).stdout
tot = {}      # kernel class -> [launches, FETCH_SIZE KiB, WRITE_SIZE KiB]: the instantiations of one kernel (k_closest<.., PRIMARY> and the bounce form) are one class
for r in csv.DictReader(io.StringIO(out)):
    m = re.search(r"k_[a-z_]+", r["kernel"])
    if not m or not r.get("FETCH_SIZE") or not r.get("WRITE_SIZE"):
        continue
    t = tot.setdefault(m.group(0), [0, 0.0, 0.0])
    t[0] += int(r["dispatches"]); t[1] += float(r["FETCH_SIZE"]); t[2] += float(r["WRITE_SIZE"])
kernels = {}
for name, (n, fk, wk) in tot.items():
    f, w = fk * 1024 / n, wk * 1024 / n
    gather = name.startswith(("k_closest", "k_shadow", "k_trace"))
    kernels[name] = {"launches_profiled": n, "fetch_size_bytes_per_launch": f, "write_size_bytes_per_launch": w, "fetch_factor": 1 if gather else 2,
                     "hbm_bytes_per_launch": (1 if gather else 2) * f + w}
json.dump({"_how": __doc__.strip(), "workload": sys.argv[2] if len(sys.argv) > 2 else "Night Spheres 1920x1080, 64 passes per batch, depth 10, 2 light samples",
           "kernels": kernels}, sys.stdout, indent=1)
