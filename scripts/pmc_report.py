#!/usr/bin/env python3
"""Per-kernel derived metrics from a scripts/pmc_passes.sh output dir: python scripts/pmc_report.py gpurun_out/pmc_<tag>
Every pass is normalised by ITS OWN GRBM_GUI_ACTIVE (the passes are separate runs); a metric whose counters were not collected
prints as '-'.  SQ_* cycle counters count quad-cycles per SIMD (MI355X_MICROARCH.md): busy fraction = counter / (cycles / 4 * 1024 SIMDs)."""
import csv, glob, os, re, sys, collections

def load(root):
    per_pass = {}
    for d in sorted(glob.glob(os.path.join(root, "*", ""))):
        name = os.path.basename(os.path.dirname(d))
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                m = re.search(r"k_[a-z_]+", r["Kernel_Name"])
                if not m: continue
                agg[m.group(0)][r["Counter_Name"]] += float(r["Counter_Value"]); disp[m.group(0)].add(r["Dispatch_Id"])
        per_pass[name] = (agg, disp)
    return per_pass

def main():
    per_pass = load(sys.argv[1])
    kernels = sorted({k for agg, _ in per_pass.values() for k in agg})
    def get(k, c):
        """(value, quad-cycle capacity of the pass that holds it, dispatches)"""
        for agg, disp in per_pass.values():
            if c in agg.get(k, {}):
                cyc = agg[k].get("GRBM_GUI_ACTIVE", 0.0) / 8
                return agg[k][c], (cyc / 4 * 1024) if cyc else None, len(disp[k]), cyc
        return None, None, 0, 0
    def frac(k, c):
        v, cap, _, _ = get(k, c)
        return f"{v / cap:.2f}" if v is not None and cap else "-"
    def ratio(k, a, b):
        x, y = get(k, a)[0], get(k, b)[0]
        return f"{x / y:.2f}" if x is not None and y else "-"
    def raw(k, c):
        v = get(k, c)[0]
        return f"{v:.3g}" if v is not None else "-"
    for k in kernels:
        _, _, n, cyc = get(k, "SQ_ACTIVE_INST_VALU")
        if not n:
            _, _, n, cyc = get(k, "GRBM_GUI_ACTIVE")
        print(f"{k:22s} dispatches {n:4d}  ms_at_2.1GHz {cyc / 2.1e6 if cyc else 0:9.2f}  valu_busy {frac(k, 'SQ_ACTIVE_INST_VALU')}  lanes_per_valu_inst {ratio(k, 'SQ_THREAD_CYCLES_VALU', 'SQ_ACTIVE_INST_VALU')}"
              f"  lds_busy {frac(k, 'SQ_ACTIVE_INST_LDS')}  lds_bank_conflict {frac(k, 'SQ_LDS_BANK_CONFLICT')}  lds_idx_active {frac(k, 'SQ_LDS_IDX_ACTIVE')}  wait_lds {ratio(k, 'SQ_WAIT_INST_LDS', 'SQ_WAVE_CYCLES')}"
              f"  wait_any {ratio(k, 'SQ_WAIT_ANY', 'SQ_WAVE_CYCLES')}  occupancy_waves_per_simd {frac(k, 'SQ_WAVE_CYCLES')}"
              f"  insts: valu {raw(k, 'SQ_INSTS_VALU')} salu {raw(k, 'SQ_INSTS_SALU')} lds {raw(k, 'SQ_INSTS_LDS')} vmem_rd {raw(k, 'SQ_INSTS_VMEM_RD')} vmem_wr {raw(k, 'SQ_INSTS_VMEM_WR')}"
              f"  fetch_GB {(get(k, 'FETCH_SIZE')[0] or 0) * 1024 / 1e9:.2f}  write_GB {(get(k, 'WRITE_SIZE')[0] or 0) * 1024 / 1e9:.2f}  l2_hit {ratio(k, 'TCC_HIT_sum', 'TCC_REQ_sum')}")

if __name__ == "__main__":
    main()
