#!/usr/bin/env python3
"""Per-kernel derived metrics from a scripts/pmc_passes.sh output dir: python scripts/pmc_report.py gpurun_out/pmc_<tag>"""
import csv, io, subprocess, sys, os, re
out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "pmc_summary.py"), sys.argv[1]], capture_output=True, text=True).stdout
for r in csv.DictReader(io.StringIO(out)):
    m = re.search(r"k_[a-z_]+", r["kernel"])
    if not m: continue
    g = lambda k: float(r[k]) if r.get(k) else 0.0
    npass = 2 if g("TCP_GATE_EN1_sum") and g("FETCH_SIZE") else 1
    cyc = g("GRBM_GUI_ACTIVE") / 8 / npass
    simdq = cyc / 4 * 1024 or 1
    print(f"{m.group(0):13s} disp {r['dispatches']:>4s} ms@2.1GHz {cyc/2.1e6:8.2f} VALUbusy {g('SQ_ACTIVE_INST_VALU')/simdq:.2f} lanes {g('SQ_THREAD_CYCLES_VALU')/max(g('SQ_ACTIVE_INST_VALU'),1):4.1f} "
          f"valu {g('SQ_INSTS_VALU'):.3g} salu {g('SQ_INSTS_SALU'):.3g} vmem {g('SQ_INSTS_VMEM_RD'):.3g} lds {g('SQ_INSTS_LDS'):.3g} wait {g('SQ_WAIT_ANY')/max(g('SQ_WAVE_CYCLES'),1):.2f} "
          f"occ {g('SQ_WAVE_CYCLES')/simdq:.1f} TAbusy {g('TA_TA_BUSY_sum')/256/max(cyc,1):.2f} fetchGB {g('FETCH_SIZE')*1024/1e9:.2f} writeGB {g('WRITE_SIZE')*1024/1e9:.2f} "
          f"L2hit {g('TCC_HIT_sum')/max(g('TCC_HIT_sum')+g('TCC_MISS_sum'),1):.2f}")
