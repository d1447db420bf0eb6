#!/usr/bin/env python3
"""Raw device counters of one short run of a config (measurement builds -DRT_SPW_PROF=1/2 put rounds and lanes into the shadow counters)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import run_configs as rc
import raytracer_amd as rt
from raytracer_amd.core import RenderOptions
c = sys.argv[1] if len(sys.argv) > 1 else "c5"
label, mk, W, H, passes, depth, ls, ppb = rc.CONFIGS[c]
sc = mk(); dev = rt.DeviceScene(sc)
job = rt.RenderJob(sc, RenderOptions(RenderWidth=W, RenderHeight=H, Passes=ppb, MaxBounceDepth=depth, LightSampleCountHint=ls, PassesPerBatch=ppb), device_scene=dev)
s0 = rc.snap(job); job.RenderPasses(0, ppb, 1); job.UpdateImage(); s1 = rc.snap(job)
print(json.dumps({"config": c, "lib": os.environ.get("RT_HIP_LIB", "in-tree"), **{k: s1[k] - s0[k] for k in s0}}))
