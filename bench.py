#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-sample trace loop on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[1], "Night Spheres" 1920x1080 — BuiltinScenes.Testing with a
seeded generator: 479 spheres + ground plane + 1 DiffuseSphereLight, depth 10, 2 light samples.
One step = PASSES_PER_STEP (256) passes over the whole 1080p frame = 531 M samples traced, shaded and accumulated: the
config's 256 spp in full, as four wavefront batches of 64 passes, two in flight (batches are sized for 288 GB of HBM: the deep
bounces of a small batch are latency-bound launches).  Scene tables, BVH and all path state are resident in HBM before the
timed region.  With N > 1 every rank renders its own passes (rank, rank+N, ...) — weak scaling — and the timed
region ends with the one sum-reduce of the colour buffers to rank 0 (RCCL) and the RGB24 resolve.

    python bench.py --gpus N --single-process      (no torchrun)

drives the N GPUs from ONE process through the library's own sharding (rt_render_opts.n_devices / device_ids: peer copies of
the scene, one host thread per device, sums gathered over xGMI) — what a C# host would use; same workload, same JSON line.
RT_BENCH_ONE_GPU=1 maps every shard to device 0 (rehearsal on a 1-GPU box).

Adds to the JSON line: "roofline" (dominant kernel, algorithmic flops / event-timed duration, SURVEY §8d cost
model) and "cpu_baseline" (oracle/ — the C++ restatement of RenderInternal — on the host cores, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PASSES_PER_STEP = 256           # one step = BASELINE config 2 in full: 1920 x 1080 x 256 spp (four 64-pass batches, two in flight)
PASSES_PER_BATCH = 64     # passes per wavefront batch (133 M paths, 24 GB of path state); the library keeps two batches in flight
WIDTH, HEIGHT = 1920, 1080
MAX_DEPTH, LIGHT_SAMPLES = 10, 2
VALU_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 32 lanes x 2 (FMA) x 2.4 GHz
HBM_PEAK_GBS = 8000.0

# SURVEY §8d cost model: flops / bytes per box test and per primitive test
F_NODE, B_NODE = 22, 32
F_PRIM = {"sphere": 28, "plane": 12}
B_PRIM = {"sphere": 16, "plane": 24}


def algorithmic_work(stats_delta, which):
    """Flops and bytes of the closest-hit or the any-hit kernel from the counters (identical definitions in oracle and kernels).
    Night Spheres holds one unbounded InfinitePlane (tested once per ray, outside the BVH) and spheres (every BVH leaf)."""
    if which == "closest":
        rays, V, T = stats_delta["ray_count"], stats_delta["node_visits"], stats_delta["prim_tests"]
        extra = 0
    else:
        rays, V, T = stats_delta["shadow_rays"], stats_delta["shadow_node_visits"], stats_delta["shadow_prim_tests"]
        extra = 60 * rays                                  # light sampling + attenuation per generated shadow ray
    t_plane = min(rays, T)
    t_sphere = T - t_plane
    flops = F_NODE * V + F_PRIM["sphere"] * t_sphere + F_PRIM["plane"] * t_plane + extra
    byts = B_NODE * V + B_PRIM["sphere"] * t_sphere + B_PRIM["plane"] * t_plane
    return flops, byts


def stats_dict(job):
    s = job.UpdateStats()
    return {"ray_count": s.RayCount, "shadow_rays": s.ShadowRays, "node_visits": s.NodeVisits, "prim_tests": s.PrimTests,
            "shadow_node_visits": s.ShadowNodeVisits, "shadow_prim_tests": s.ShadowPrimTests, "pixels": s.PixelsRendered,
            "scatter": s.MaterialScatterCount, "limit": s.BounceLimitExceeded}


def main():
    global PASSES_PER_STEP, PASSES_PER_BATCH
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--passes-per-step", type=int, default=PASSES_PER_STEP)
    ap.add_argument("--passes-per-batch", type=int, default=PASSES_PER_BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--single-process", action="store_true", help="one process drives --gpus devices through the library's in-process sharding")
    args = ap.parse_args()
    PASSES_PER_STEP, PASSES_PER_BATCH = args.passes_per_step, args.passes_per_batch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.single_process:
        return single_process(args)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library is the only trace path (no CPU fallback)")
    if os.environ.get("RT_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") over xGMI is the real path; RT_BENCH_BACKEND=gloo + RT_BENCH_ONE_GPU=1 rehearse the N > 1 code on a 1-GPU box
        backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import raytracer_amd as rt
    from raytracer_amd.core import RenderOptions

    scene = rt.BuiltinScenes.Testing()
    t0 = time.perf_counter()
    dev_scene = rt.DeviceScene(scene, local_rank)
    scene_upload_s = time.perf_counter() - t0
    total_steps = args.warmup + args.steps
    opt = RenderOptions(RenderWidth=WIDTH, RenderHeight=HEIGHT, Passes=total_steps * PASSES_PER_STEP * world, MaxBounceDepth=MAX_DEPTH,
                        LightSampleCountHint=LIGHT_SAMPLES, PassesPerBatch=PASSES_PER_BATCH)
    job = rt.RenderJob(scene, opt, device=local_rank, rank=rank, world_size=world, device_scene=dev_scene)

    def step(i):     # pass p of this rank's step i: global pass index rank + world * (i * PPS + j)
        job.RenderPasses(rank + world * i * PASSES_PER_STEP, PASSES_PER_STEP, world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    s0 = stats_dict(job)
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        step(i)
    if world > 1:
        job.ReduceToRoot()
    if rank == 0:
        job.UpdateImage(total_samples=total_steps * PASSES_PER_STEP * world if world > 1 else None)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    s1 = stats_dict(job)
    d = {k: s1[k] - s0[k] for k in s0}
    if world > 1:
        t = torch.tensor([d["ray_count"], d["shadow_rays"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        rays_all, shadow_all = float(t[0].item()), float(t[1].item())
    else:
        rays_all, shadow_all = float(d["ray_count"]), float(d["shadow_rays"])

    samples = WIDTH * HEIGHT * PASSES_PER_STEP * args.steps * world
    out = {
        "metric": "Msamples/s", "value": samples / dt / 1e6, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "Night Spheres (BuiltinScenes.Testing, seed 0x5EED) 1920x1080, depth 10, 2 light samples, "
                               f"{PASSES_PER_STEP} passes per step ({args.steps * PASSES_PER_STEP * world} spp timed)",
                   "objects": len(scene.SceneObjects), "lights": len(scene.Lights), "parallelism": f"pass-sharded x{world}"},
        "mrays_per_s": {"closest": rays_all / dt / 1e6, "shadow": shadow_all / dt / 1e6, "total": (rays_all + shadow_all) / dt / 1e6},
        "scene_upload_and_bvh_s": scene_upload_s,
        "ray_depth_counts": job.RenderStats.RayDepthCounts[:MAX_DEPTH + 1],
        "scene_info": dev_scene.Info(),
    }

    if rank == 0 and world == 1 and not args.no_roofline:
        # a separate, event-timed leg (same workload, 1 more step, batches one at a time so that kernels do not overlap):
        # per-kernel durations from HIP events on the library's own stream (rt_render_set_profiling)
        job.SetProfiling(True)
        p0 = stats_dict(job)
        step(total_steps)
        prof = job.KernelProfile()
        job.SetProfiling(False)
        p1 = stats_dict(job)
        pd = {k: p1[k] - p0[k] for k in p0}
        traffic = {}
        # PMC FETCH_SIZE/WRITE_SIZE of this workload (separate rocprofv3 passes: scripts/pmc_passes.sh, scripts/pmc_traffic.py)
        tpath = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r02_traffic.json", "r01_traffic.json")) if os.path.exists(q)), "")
        if tpath and PASSES_PER_BATCH == 64:
            traffic = {k: v["hbm_bytes_per_launch"] for k, v in json.load(open(tpath))["kernels"].items()}
        dominant = max(("closest", "shadow", "shade"), key=lambda k: prof[k]["total_ms"])

        def valu_roofline(which):
            flops, byts = algorithmic_work(pd, which)
            sec, launches = prof[which]["total_ms"] / 1e3, max(prof[which]["launches"], 1)
            tr = traffic.get("k_" + which)
            if tr:   # FETCH_SIZE counts the coalesced path-state reads at half (scripts/pmc_traffic.py): add the other half back
                tr += 0.5 * (32 * pd["ray_count"] if which == "closest" else 64 * pd["shadow_rays"] / LIGHT_SAMPLES) / launches
            return {"kernel": "k_" + which, "bound": "valu", "achieved": flops / sec / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": flops / sec / 1e12 / VALU_PEAK_TFLOPS, "traffic": tr, "launches": launches, "avg_launch_ms": prof[which]["total_ms"] / launches,
                    "algorithmic_flops_per_launch": flops / launches, "algorithmic_bytes_per_launch": byts / launches,
                    "hbm_gbs_from_traffic": (tr * launches / sec / 1e9) if tr else None,
                    "note": "FP32 vector peak (= the f32 MFMA peak; nothing here is a dense contraction). It counts an FMA as 2 flops and the kernels are built "
                            "-ffp-contract=off for bit-parity, so half of it is their ceiling. algorithmic_bytes are BVH node / primitive bytes (SURVEY 8d): "
                            "this scene's tree lives in LDS, so they never reach HBM; `traffic` is what does (path state), from PMC counters."}

        def shade_roofline():
            # k_shade moves path state: 64 B read per alive path (56 state + 8 hit), 56 B written per survivor, 48 B per shadow request, 16 B per finished path
            alive, scat, fin = pd["ray_count"], pd["scatter"], pd["pixels"]
            byts = 64 * alive + 56 * (scat - pd["limit"]) + 48 * (pd["shadow_rays"] // LIGHT_SAMPLES) + 16 * fin
            sec, launches = prof["shade"]["total_ms"] / 1e3, max(prof["shade"]["launches"], 1)
            return {"kernel": "k_shade", "bound": "hbm", "achieved": byts / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byts / sec / 1e9 / HBM_PEAK_GBS,
                    "traffic": traffic.get("k_shade"), "launches": launches, "avg_launch_ms": prof["shade"]["total_ms"] / launches,
                    "algorithmic_bytes_per_launch": byts / launches}

        out["roofline"] = shade_roofline() if dominant == "shade" else valu_roofline(dominant)
        out["roofline_others"] = [valu_roofline(k) for k in ("closest", "shadow") if k != dominant] + ([shade_roofline()] if dominant != "shade" else [])
        out["kernel_ms"] = {k: round(v["total_ms"], 3) for k, v in prof.items()}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import binding as ob
        threads = min(16, os.cpu_count() or 1)
        cw, ch, cp = 1920, 1080, 10          # ~20 M samples: 10 s of work for 16 host threads
        copt = RenderOptions(RenderWidth=cw, RenderHeight=ch, Passes=cp, MaxBounceDepth=MAX_DEPTH, LightSampleCountHint=LIGHT_SAMPLES)
        osc = ob.OracleScene(dev_scene.flat)
        t0 = time.perf_counter()
        osc.render(copt, threads=threads)
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": cw * ch * cp / cdt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
                               "sample": f"same scene and options at {cw}x{ch}, {cp} passes ({cw * ch * cp} samples, {cdt:.1f} s)"}

    if rank == 0:
        print(json.dumps(out))
    job.Dispose()
    if world > 1:
        dist.destroy_process_group()


def single_process(args):
    """--single-process: N devices behind one rt_render handle (RenderJob(devices=[...]))."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library is the only trace path (no CPU fallback)")
    import raytracer_amd as rt
    from raytracer_amd.core import RenderOptions
    n = args.gpus
    devices = [0] * n if os.environ.get("RT_BENCH_ONE_GPU") == "1" else list(range(n))
    scene = rt.BuiltinScenes.Testing()
    t0 = time.perf_counter()
    dev_scene = rt.DeviceScene(scene, devices[0])
    scene_upload_s = time.perf_counter() - t0
    total_steps = args.warmup + args.steps
    per_step = PASSES_PER_STEP * n                      # weak scaling: every device renders PASSES_PER_STEP passes per step
    opt = RenderOptions(RenderWidth=WIDTH, RenderHeight=HEIGHT, Passes=total_steps * per_step, MaxBounceDepth=MAX_DEPTH,
                        LightSampleCountHint=LIGHT_SAMPLES, PassesPerBatch=PASSES_PER_BATCH)
    job = rt.RenderJob(scene, opt, device=devices[0], device_scene=dev_scene, devices=devices)
    for i in range(args.warmup):
        job.RenderPasses(i * per_step, per_step, 1)
    torch.cuda.synchronize()
    s0 = stats_dict(job)
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        job.RenderPasses(i * per_step, per_step, 1)     # returns when every shard has finished its passes
    job.UpdateImage()                                   # gathers the shards' sums onto devices[0] and resolves
    dt = time.perf_counter() - t0
    s1 = stats_dict(job)
    d = {k: s1[k] - s0[k] for k in s0}
    samples = WIDTH * HEIGHT * per_step * args.steps
    out = {"metric": "Msamples/s", "value": samples / dt / 1e6, "unit": "Msamples/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "Night Spheres (BuiltinScenes.Testing, seed 0x5EED) 1920x1080, depth 10, 2 light samples, "
                                  f"{PASSES_PER_STEP} passes per step per GPU ({args.steps * per_step} spp timed)",
                      "objects": len(scene.SceneObjects), "lights": len(scene.Lights), "parallelism": f"one process, pass-sharded over devices {devices} inside the library"},
           "mrays_per_s": {"closest": d["ray_count"] / dt / 1e6, "shadow": d["shadow_rays"] / dt / 1e6, "total": (d["ray_count"] + d["shadow_rays"]) / dt / 1e6},
           "scene_upload_and_bvh_s": scene_upload_s, "scene_info": dev_scene.Info()}
    print(json.dumps(out))
    job.Dispose()


if __name__ == "__main__":
    main()
