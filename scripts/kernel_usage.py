#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in rt_lib.hip (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, no GPU needed).
usage: python scripts/kernel_usage.py [only=spheres] [extra hipcc flags...]   e.g. -DRT_CP_WAVES=2"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -Wno-unused-function".split()
CFG = {"9": "Spheres", "12": "Mesh", "371": "Boxes", "4095": "All"}

def main():
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(ROOT, "raytracer_amd", "csrc", f) for f in ("rt_lib.hip", "rt_variant_spheres.hip", "rt_variant_mesh.hip", "rt_variant_boxes.hip", "rt_variant_all.hip")]
    only = [a for a in sys.argv[1:] if a.startswith("only=")]
    if only:
        srcs = [f for f in srcs if only[0][5:] in f]
    extra = [a for a in sys.argv[1:] if not a.startswith("only=")]
    def run(src):
        return subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
    with ThreadPoolExecutor(5) as ex:
        res = list(ex.map(run, srcs))
    for p in res:
        if p.returncode:
            sys.stderr.write(p.stderr[-4000:]); sys.exit(1)
    class P: pass
    p = P(); p.stderr = "".join(r.stderr for r in res)
    blocks = re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]
    names = [b.split("\n")[0].split(" [")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
    print(f"{'kernel':58s} {'vgpr':>4s} {'spill':>5s} {'scratch':>7s} {'waves':>5s} {'lds':>6s}")
    for b, d in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        d = re.sub(r"rtd::Cfg<(\d+)u, \d+u, \d+u, \d+u>", lambda m: CFG.get(m.group(1), "Cfg" + m.group(1)), d)
        d = re.sub(r"^void ", "", d).split("(")[0].replace("rtd::", "")
        row = (g("VGPRs"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"))
        print(f"{d:58s} {row[0]:4d} {row[1]:5d} {row[2]:7d} {row[3]:5d} {row[4]:6d}")

if __name__ == "__main__":
    main()
