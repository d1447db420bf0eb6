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
model) and "cpu_baseline" (oracle/ — the C++ restatement of RenderInternal — on the host cores, rank 0, N=1), and
  "strong_scaling": ONE fixed job — config 2's 256 passes — split over the N ranks (256 / N passes each), timed end to end with the
                    reduce and the resolve, the reduce's milliseconds beside it (BASELINE's configs 4 and 5 are jobs of that shape);
  "value_exact":    (N = 1) the same workload with RT_RENDER_EXACT_LIGHT_SAMPLING | RT_RENDER_EXACT_OCCLUSION;
  "configs":        (N = 1) short runs of configs 1, 3, 4 and 5 at full scene size and resolution (scripts/run_configs.py),
                    each with Msamples/s, its dominant kernel, algorithmic flops / bytes, event-timed milliseconds and `frac`;
  "world_size", "backend", "rccl_version": what the process group saw.
--scaling strong makes the fixed 256-pass job the timed region itself (value = its Msamples/s, "scaling": "strong").
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
PASSES_PER_BATCH = int(os.environ.get("RT_BENCH_PPB", "64"))     # passes per wavefront batch (64: 133 M paths, 24 GB of path state); the library keeps two batches in flight (RT_BENCH_PPB: A/B runs)
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
            "scatter": s.MaterialScatterCount, "limit": s.BounceLimitExceeded, "depth_hist": list(s.RayDepthCounts)}


def stats_delta(b, a):
    return {k: [x - y for x, y in zip(b[k], a[k])] if isinstance(b[k], list) else b[k] - a[k] for k in a}


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
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak", help="weak: every GPU renders 256 passes per step; strong: ONE 256-pass job per step, split over the GPUs")
    ap.add_argument("--no-configs", action="store_true", help="leave out the short runs of configs 1, 3, 4, 5")
    ap.add_argument("--no-exact", action="store_true", help="leave out the exact-mode run of the same workload")
    ap.add_argument("--no-strong", action="store_true", help="leave out the strong-scaling leg")
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

    backend = dist.get_backend() if world > 1 else "none"
    try:
        rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:      # noqa: BLE001 — a torch build without the binding
        rccl = None
    scene = rt.BuiltinScenes.Testing()
    t0 = time.perf_counter()
    dev_scene = rt.DeviceScene(scene, local_rank)
    scene_upload_s = time.perf_counter() - t0
    strong = args.scaling == "strong"

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def strong_job(reps, warm):
        """ONE job of PASSES_PER_STEP passes, split over the ranks (rank r renders passes r, r + N, ...: 256 / N each), timed end to end:
        zeroed sums -> passes -> reduce to rank 0 -> RGB24 on rank 0's host.  `reps` such jobs after `warm` untimed ones.  The batch size
        is the library's choice for a call of that many passes (rt_render_opts.max_passes_per_call)."""
        opt = RenderOptions(RenderWidth=WIDTH, RenderHeight=HEIGHT, Passes=PASSES_PER_STEP, MaxBounceDepth=MAX_DEPTH, LightSampleCountHint=LIGHT_SAMPLES,
                            PassesPerBatch=int(os.environ.get("RT_BENCH_STRONG_PPB", "0")))
        job = rt.RenderJob(scene, opt, device=local_rank, rank=rank, world_size=world, device_scene=dev_scene)
        first, n_mine, stride = rt.render_job.shard_passes(PASSES_PER_STEP, rank, world)
        t_render = t_reduce = t_resolve = 0.0
        t_all = 0.0
        for i in range(warm + reps):
            fence()
            t0 = time.perf_counter()
            if world > 1:
                job._accum_tensor.zero_(); torch.cuda.synchronize()         # a new job starts from empty sums
            job.RenderPasses(i * PASSES_PER_STEP + first, n_mine, stride)
            t1 = time.perf_counter()
            if world > 1:
                job.ReduceToRoot()                                          # the one exchange step (RCCL reduce of W x H x 3 floats)
            t2 = time.perf_counter()
            if rank == 0:
                job.UpdateImage(total_samples=PASSES_PER_STEP if world > 1 else None)
            fence()
            t3 = time.perf_counter()
            if i >= warm:
                t_all += t3 - t0; t_render += t1 - t0; t_reduce += t2 - t1; t_resolve += t3 - t2
        job.Dispose()
        t_all, t_render, t_reduce = max_over_ranks(t_all), max_over_ranks(t_render), max_over_ranks(t_reduce)
        samples = WIDTH * HEIGHT * PASSES_PER_STEP * reps
        return {"job": f"config 2 as one job: {WIDTH}x{HEIGHT} x {PASSES_PER_STEP} passes split over {world} GPU(s), sums zeroed -> passes -> reduce -> RGB24 on the host",
                "jobs_timed": reps, "passes_per_gpu": n_mine, "seconds_per_job": t_all / reps, "value": samples / t_all / 1e6, "per_gpu": samples / t_all / 1e6 / world,
                "unit": "Msamples/s", "render_ms": t_render / reps * 1e3, "reduce_ms": t_reduce / reps * 1e3, "resolve_and_barrier_ms": t_resolve / reps * 1e3}, samples, t_all

    if strong:
        sres, samples, dt = strong_job(args.steps, args.warmup)
        out = {"metric": "Msamples/s", "value": sres["value"], "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "Night Spheres (BuiltinScenes.Testing, seed 0x5EED) 1920x1080, depth 10, 2 light samples, "
                                      f"one {PASSES_PER_STEP}-pass job per step split over the GPUs",
                          "objects": len(scene.SceneObjects), "lights": len(scene.Lights), "parallelism": f"pass-sharded x{world}"},
               "strong_scaling": sres, "world_size": world, "backend": backend, "rccl_version": rccl, "scene_info": dev_scene.Info()}
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.destroy_process_group()
        return

    total_steps = args.warmup + args.steps
    opt = RenderOptions(RenderWidth=WIDTH, RenderHeight=HEIGHT, Passes=total_steps * PASSES_PER_STEP * world, MaxBounceDepth=MAX_DEPTH,
                        LightSampleCountHint=LIGHT_SAMPLES, PassesPerBatch=PASSES_PER_BATCH)
    job = rt.RenderJob(scene, opt, device=local_rank, rank=rank, world_size=world, device_scene=dev_scene)

    def step(i):     # pass p of this rank's step i: global pass index rank + world * (i * PPS + j)
        job.RenderPasses(rank + world * i * PASSES_PER_STEP, PASSES_PER_STEP, world)

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
    dt = max_over_ranks(time.perf_counter() - t0)
    s1 = stats_dict(job)
    d = stats_delta(s1, s0)
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
        "world_size": world, "backend": backend, "rccl_version": rccl,
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
        pd = stats_delta(p1, p0)
        traffic = {}
        # PMC FETCH_SIZE/WRITE_SIZE of this workload (separate rocprofv3 passes: scripts/pmc_passes.sh, scripts/pmc_traffic.py)
        tpath = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json")) if os.path.exists(q)), "")
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
            # What k_shade moves today (rt_kernels.h: k_shade; Night Spheres: tree in LDS, so k_closest finished the misses — finish_miss — and left
            # the depth-0 ray's 32 B; compact 16-B shadow requests below the depth limit):
            #   8 B  per queue position (the counting phase reads every hit id)
            #   32 B per depth-0 hit (the stored camera ray), 56 B per deeper hit (its state)
            #   56 B per survivor (next state), 16 B per request of a survivor, 48 B per request of a path at the depth limit, 16 B per path ended there
            alive, hits, lim = pd["ray_count"], pd["scatter"], pd["limit"]
            hits0 = sum(pd["depth_hist"])                   # a path is in the histogram once it has hit something: the hits of depth 0
            requests = pd["shadow_rays"] // LIGHT_SAMPLES   # one light
            byts = 8 * alive + 32 * hits0 + 56 * (hits - hits0) + 56 * (hits - lim) + 16 * (requests - lim) + 48 * lim + 16 * lim
            sec, launches = prof["shade"]["total_ms"] / 1e3, max(prof["shade"]["launches"], 1)
            return {"kernel": "k_shade", "bound": "hbm", "achieved": byts / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byts / sec / 1e9 / HBM_PEAK_GBS,
                    "traffic": traffic.get("k_shade"), "launches": launches, "avg_launch_ms": prof["shade"]["total_ms"] / launches,
                    "algorithmic_bytes_per_launch": byts / launches}

        out["roofline"] = shade_roofline() if dominant == "shade" else valu_roofline(dominant)
        out["roofline_others"] = [valu_roofline(k) for k in ("closest", "shadow") if k != dominant] + ([shade_roofline()] if dominant != "shade" else [])
        out["kernel_ms"] = {k: round(v["total_ms"], 3) for k, v in prof.items()}

    job.Dispose()

    if not args.no_strong:
        out["strong_scaling"] = strong_job(2, 1)[0]

    if rank == 0 and world == 1 and not args.no_exact:
        # the same workload in the exact modes: the reference's correctly rounded light sampling, and occlusion answers equal to its every-object scan
        eopt = RenderOptions(RenderWidth=WIDTH, RenderHeight=HEIGHT, Passes=3 * PASSES_PER_STEP, MaxBounceDepth=MAX_DEPTH, LightSampleCountHint=LIGHT_SAMPLES,
                             PassesPerBatch=PASSES_PER_BATCH, ExactLightSampling=True, ExactOcclusion=True)
        ejob = rt.RenderJob(scene, eopt, device=local_rank, device_scene=dev_scene)
        ejob.RenderPasses(0, PASSES_PER_BATCH, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ejob.RenderPasses(PASSES_PER_STEP, PASSES_PER_STEP, 1)
        ejob.UpdateImage()
        edt = time.perf_counter() - t0
        ejob.Dispose()
        out["value_exact"] = {"value": WIDTH * HEIGHT * PASSES_PER_STEP / edt / 1e6, "unit": "Msamples/s", "ms_per_step": edt * 1e3, "steps": 1,
                              "flags": "RT_RENDER_EXACT_LIGHT_SAMPLING | RT_RENDER_EXACT_OCCLUSION", "vs_default": WIDTH * HEIGHT * PASSES_PER_STEP / edt / 1e6 / out["value"]}

    if rank == 0 and world == 1 and not args.no_configs:
        # BASELINE's other configs, short runs at full scene size and resolution (c2 is the line itself)
        import importlib.util
        spec = importlib.util.spec_from_file_location("run_configs", os.path.join(ROOT, "scripts", "run_configs.py"))
        rc = importlib.util.module_from_spec(spec); spec.loader.exec_module(rc)
        tpath = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r03_traffic_configs.json", "r02_traffic_configs.json")) if os.path.exists(q)), "")
        ctraffic = json.load(open(tpath)) if tpath else {}
        out["configs"] = []
        for c in ("c1", "c3", "c4", "c5"):
            r = rc.run_config(c, short=True, traffic=ctraffic)
            rl = r["roofline"]
            out["configs"].append({"config": c, "workload": r["workload"], "passes": r["passes"], "msamples_per_s": r["msamples_per_s"], "mrays_per_s": r["mrays_per_s"],
                                   "dominant_kernel": rl["kernel"], "bound": rl["bound"], "frac": rl["frac"], "frac_valu": rl["frac_valu"], "frac_hbm_model": rl["frac_hbm_model"],
                                   "frac_hbm_pmc": rl.get("frac_hbm_pmc"), "traffic": rl["traffic"], "launches": rl["launches"], "avg_launch_ms": rl["avg_launch_ms"],
                                   "algorithmic_flops": rl["algorithmic_flops"], "algorithmic_bytes": rl["algorithmic_bytes"], "box_tests": rl["box_tests_V"], "prim_tests": rl["prim_tests_T"],
                                   "kernel_ms_one_batch": r["kernel_ms_one_batch"], "scene_create_s": r["scene_create_s"],
                                   "tree": {k: r["scene_info"][k] for k in ("n_objects", "n_nodes", "wide_nodes", "object_list", "variant", "built_on_device")}})

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
    d = stats_delta(s1, s0)
    samples = WIDTH * HEIGHT * per_step * args.steps
    out = {"metric": "Msamples/s", "value": samples / dt / 1e6, "unit": "Msamples/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "Night Spheres (BuiltinScenes.Testing, seed 0x5EED) 1920x1080, depth 10, 2 light samples, "
                                  f"{PASSES_PER_STEP} passes per step per GPU ({args.steps * per_step} spp timed)",
                      "objects": len(scene.SceneObjects), "lights": len(scene.Lights), "parallelism": f"one process, pass-sharded over devices {devices} inside the library"},
           "mrays_per_s": {"closest": d["ray_count"] / dt / 1e6, "shadow": d["shadow_rays"] / dt / 1e6, "total": (d["ray_count"] + d["shadow_rays"]) / dt / 1e6},
           "scene_upload_and_bvh_s": scene_upload_s, "scene_info": dev_scene.Info()}
    job.Dispose()
    if not args.no_strong:
        # ONE job of PASSES_PER_STEP passes behind one handle: the library splits every call's passes over the devices; the read gathers the
        # shards' sums over xGMI onto devices[0] (rt_lib.hip: combine_shards) and resolves.  A new handle per job = sums that start at zero.
        sopt = RenderOptions(RenderWidth=WIDTH, RenderHeight=HEIGHT, Passes=PASSES_PER_STEP, MaxBounceDepth=MAX_DEPTH, LightSampleCountHint=LIGHT_SAMPLES,
                             PassesPerBatch=int(os.environ.get("RT_BENCH_STRONG_PPB", "0")))
        sjob = rt.RenderJob(scene, sopt, device=devices[0], device_scene=dev_scene, devices=devices)
        reps, t_render, t_read = 2, 0.0, 0.0
        for i in range(1 + reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sjob.RenderPasses(i * PASSES_PER_STEP, PASSES_PER_STEP, 1)
            t1 = time.perf_counter()
            sjob.UpdateImage()
            t2 = time.perf_counter()
            if i >= 1:
                t_render += t1 - t0; t_read += t2 - t1
        sjob.Dispose()
        ssamples = WIDTH * HEIGHT * PASSES_PER_STEP * reps
        out["strong_scaling"] = {"job": f"config 2 as one job: {WIDTH}x{HEIGHT} x {PASSES_PER_STEP} passes split over {n} device(s) inside the library -> gather + add + RGB24 on the host",
                                 "jobs_timed": reps, "passes_per_gpu": (PASSES_PER_STEP + n - 1) // n, "seconds_per_job": (t_render + t_read) / reps,
                                 "value": ssamples / (t_render + t_read) / 1e6, "per_gpu": ssamples / (t_render + t_read) / 1e6 / n, "unit": "Msamples/s",
                                 "render_ms": t_render / reps * 1e3, "gather_add_resolve_ms": t_read / reps * 1e3}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
