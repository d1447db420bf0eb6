"""RenderJob: the reference's entry point (Ararem.RayTracer.Core/RenderJob.cs:24-62, 69-126, 695-735) with its
RenderInternal body replaced by the HIP library behind include/rt_hip.h.

    job = RenderJob(scene, RenderOptions(...))
    job.StartOrGetRenderAsync()          # RenderJob.cs:708 — starts the render thread once, returns the task
    job.RenderTask.result()              # or job.Wait()
    job.Image  -> (H, W, 3) uint8, row 0 = top (Image<Rgb24>, RenderJob.cs:679)
    job.RenderStats

Multi-GPU: one process per GPU.  Rank r of G renders passes {p : p mod G == r} (samples are independent,
RenderJob.cs:73-78) into its own colour sums; `ReduceToRoot()` combines them with one sum-reduce (RCCL over
xGMI when the sums live in device memory).  The counter RNG is keyed by (pixel, pass), so the image does not
depend on G beyond float-sum association.

The HIP library is mandatory: there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
import threading
from concurrent.futures import Future
from typing import Optional

import numpy as np

from . import capi
from .core import RenderOptions, RenderStats, Scene
from .flatten import FlatScene, flatten_scene


def shard_passes(total_passes: int, rank: int, world_size: int):
    """(first_pass, n_passes, stride) of rank `rank`: passes rank, rank+G, rank+2G, ... below total_passes."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    n = (total_passes - rank + world_size - 1) // world_size if total_passes > rank else 0
    return rank, n, world_size


class OperationCanceledError(RuntimeError):
    """System.OperationCanceledException (RenderJob.cs:114-118)."""


class DeviceScene:
    """rt_scene handle (flattened Scene + BVH on one device)."""

    def __init__(self, scene_or_flat, device: int = -1):
        self.flat: FlatScene = scene_or_flat if isinstance(scene_or_flat, FlatScene) else flatten_scene(scene_or_flat)
        self._lib = capi.load()
        self.handle = C.c_void_p()
        capi.check(self._lib.rt_scene_create(C.byref(self.flat.desc), device, C.byref(self.handle)))
        self._device = int(self.Info()["device"])          # the device rt_scene_create used (for -1: HIP's current device of this thread at that time)

    def DeviceOrdinal(self) -> int:
        """The HIP device the scene lives on (-1 at construction = HIP's current device at that time)."""
        return self._device

    def Info(self) -> dict:
        """Shape of the device BVH and the kernel variant chosen (rt_scene_get_info)."""
        si = capi.SceneInfo()
        capi.check(self._lib.rt_scene_get_info(self.handle, C.byref(si)))
        return {"n_objects": si.n_objects, "n_unbounded": si.n_unbounded, "n_nodes": si.n_nodes, "bvh_depth": si.bvh_depth,
                "tree_in_lds": bool(si.tree_in_lds), "lds_image_bytes": si.lds_image_bytes, "built_on_device": bool(si.built_on_device), "object_list": bool(si.object_list),
                "variant": si.variant.decode(), "wide_nodes": si.wide_nodes, "wide_levels": si.wide_levels, "device": si.device}

    def TraceClosest(self, origins, directions, kMin=0.001, kMax=np.inf, seed=0):
        """RenderJob.TryFindClosestHit (RenderJob.cs:581-617) on a batch of rays -> structured array of rt_hit."""
        rays = _pack_rays(origins, directions)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        capi.check(self._lib.rt_trace_closest(self.handle, rays.ctypes.data, len(rays), kMin, kMax, seed, hits.ctypes.data))
        return hits

    def TraceAny(self, origins, directions, kMin, kMax):
        """RenderJob.AnyIntersectionFast (RenderJob.cs:575) on a batch of rays -> bool array."""
        rays = _pack_rays(origins, directions)
        n = len(rays)
        k0 = np.ascontiguousarray(np.broadcast_to(np.asarray(kMin, np.float32), (n,)))
        k1 = np.ascontiguousarray(np.broadcast_to(np.asarray(kMax, np.float32), (n,)))
        out = np.zeros(n, dtype=np.uint8)
        capi.check(self._lib.rt_trace_any(self.handle, rays.ctypes.data, k0.ctypes.data, k1.ctypes.data, n, out.ctypes.data))
        return out.astype(bool)

    def Dispose(self):
        """rt_scene_destroy refuses while a render still uses the scene (destroy the RenderJobs first); the handle is kept then."""
        if self.handle:
            if self._lib.rt_scene_destroy(self.handle) == 0:
                self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass


HIT_DTYPE = np.dtype([("k", "<f4"), ("object", "<u4"), ("material", "<u4"), ("normal", "<f4", 3), ("uv", "<f4", 2),
                      ("world_point", "<f4", 3), ("outside_face", "<u4")])
assert HIT_DTYPE.itemsize == C.sizeof(capi.Hit)


def _pack_rays(origins, directions) -> np.ndarray:
    o = np.asarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.asarray(directions, dtype=np.float32).reshape(-1, 3)
    if len(o) != len(d):
        raise ValueError("origins and directions differ in length")
    return np.ascontiguousarray(np.concatenate([o, d], axis=1))


def make_opts(o: RenderOptions, device=-1, accum_device=0, stream=0, devices=None, max_passes_per_call=0) -> capi.RenderOpts:
    ro = capi.RenderOpts()
    ro.struct_size = C.sizeof(capi.RenderOpts)
    ro.width, ro.height = o.RenderWidth, o.RenderHeight
    ro.k_min, ro.k_max = o.KMin, o.KMax
    ro.max_bounce_depth, ro.light_sample_count_hint = o.MaxBounceDepth, o.LightSampleCountHint
    ro.debug_visualisation = o.DebugVisualisation
    ro.seed = o.Seed
    ro.passes_per_batch = o.PassesPerBatch                # 0 = the library's choice: min(64, 128 M paths / pixels, max_passes_per_call)
    ro.max_passes_per_call = max_passes_per_call or max(1, o.Passes)      # never path state for more passes than the job renders
    ro.device = device
    ro.stream = stream or None
    ro.accum_device = accum_device or None
    ro.flags = ((capi.RENDER_EXACT_LIGHT_SAMPLING if getattr(o, "ExactLightSampling", False) else 0) | (capi.RENDER_EXACT_OCCLUSION if getattr(o, "ExactOcclusion", False) else 0) |
                (capi.RENDER_VALIDATE if getattr(o, "Validate", False) else 0))
    if devices:
        if len(devices) > capi.RT_MAX_DEVICES:
            raise ValueError(f"at most {capi.RT_MAX_DEVICES} devices")
        ro.n_devices = len(devices)
        for i, d in enumerate(devices):
            ro.device_ids[i] = int(d)
    return ro


class RenderJob:
    def __init__(self, scene: Scene, renderOptions: RenderOptions, device: int = -1, rank: int = 0, world_size: int = 1,
                 process_group=None, device_scene: Optional[DeviceScene] = None, devices=None):
        """devices=[d0, d1, ...]: ONE process, several GPUs — the library shards every call's passes over them and combines the sums
        on d0 (rt_render_opts.n_devices).  rank / world_size: one process per GPU, combined by ReduceToRoot (RCCL)."""
        self.Scene, self.RenderOptions = scene, renderOptions
        self.rank, self.world_size, self.process_group = rank, world_size, process_group
        self._lib = capi.load()
        if devices and device_scene is None:
            device = devices[0]
        self.device_scene = device_scene or DeviceScene(scene, device)
        W, H = renderOptions.RenderWidth, renderOptions.RenderHeight
        self.ImageBuffer = np.zeros((H, W, 3), dtype=np.uint8)
        self._accum_tensor = None
        accum_ptr = 0
        if world_size > 1:
            if devices:
                raise ValueError("devices= (one process, several GPUs) and world_size > 1 (one process per GPU) exclude each other")
            # the sums live in a torch tensor so torch.distributed (RCCL) can reduce them in place — on the SCENE's device
            import torch
            self._accum_tensor = torch.zeros(H * W * 3, dtype=torch.float32, device=torch.device("cuda", self.device_scene.DeviceOrdinal()))
            torch.cuda.synchronize(self._accum_tensor.device)
            accum_ptr = self._accum_tensor.data_ptr()
        # (a rank never asks for more than its own share of the job's passes in one call: the library sizes its batches by that)
        self._opts = make_opts(renderOptions, device=-1, accum_device=accum_ptr, devices=devices,
                               max_passes_per_call=shard_passes(renderOptions.Passes, rank, world_size)[1] if world_size > 1 else 0)
        self.handle = C.c_void_p()
        capi.check(self._lib.rt_render_create(self.device_scene.handle, C.byref(self._opts), C.byref(self.handle)))
        self.RenderStats = RenderStats(TotalTruePixels=W * H, TotalPasses=renderOptions.Passes)
        self.RenderTask: Optional[Future] = None
        self.RenderCompleted = False
        self.Cancelled = False
        self._started = threading.Lock()
        self._cancel = threading.Event()
        self._passes_done = 0            # this rank's passes
        self.PassesSoFar = 0             # advanced after every rt_render_passes call (what a progress display reads while the task runs)
        self.passes_per_call = 0         # 0 = all of this rank's passes in one rt_render_passes call

    # ---- RenderJob.cs:708-723 -----------------------------------------------------------------------------------
    def StartOrGetRenderAsync(self, cancellation: Optional[threading.Event] = None) -> Future:
        if self._started.acquire(blocking=False):
            if cancellation is not None:
                self._cancel = cancellation
            self.RenderTask = Future()
            threading.Thread(target=self._run, name="RenderJob", daemon=True).start()
        return self.RenderTask

    def Wait(self):
        self.StartOrGetRenderAsync()
        return self.RenderTask.result()

    def _run(self):
        try:
            self.RenderInternal()
            self.RenderTask.set_result(self)
        except BaseException as e:       # noqa: BLE001 — the task carries the exception, as a faulted Task does
            self.RenderTask.set_exception(e)

    # ---- RenderJob.cs:69-126 -------------------------------------------------------------------------------------
    def RenderInternal(self):
        first, n, stride = shard_passes(self.RenderOptions.Passes, self.rank, self.world_size)
        try:
            self.RenderPasses(first, n, stride)
        except OperationCanceledError:
            self.Cancelled = True        # RenderJob.cs:114-118: the reference logs, leaves the pass loop and lets the task complete
        if self.world_size > 1:
            self.ReduceToRoot()
        if self.rank == 0:
            # (after a cancel the ranks may have rendered different numbers of passes: resolve with what was rendered)
            self.UpdateImage(total_samples=self._reduced_passes if self.world_size > 1 else None)
        self.UpdateStats(reduce=self.world_size > 1)
        self.RenderCompleted = True

    def RenderPasses(self, first_pass: int, n_passes: int, stride: int = 1):
        """The pass loop; cancellation is honoured between rt_render_passes calls (the reference checks its
        token per Parallel.For iteration, RenderJob.cs:97,114-118)."""
        chunk = self.passes_per_call or max(1, n_passes)
        done = 0
        while done < n_passes:
            if self._cancel.is_set():
                self._passes_done += done
                raise OperationCanceledError("render cancelled")
            m = min(chunk, n_passes - done)
            capi.check(self._lib.rt_render_passes(self.handle, first_pass + done * stride, m, stride))
            done += m
            self.PassesSoFar += m
        self._passes_done += n_passes

    def ReduceToRoot(self):
        """The one exchange step: sum-reduce the colour sums (and the pass counts) to rank 0 — RCCL over xGMI on device tensors."""
        import torch
        import torch.distributed as dist
        dist.reduce(self._accum_tensor, dst=0, op=dist.ReduceOp.SUM, group=self.process_group)
        n = torch.tensor([self._passes_done], dtype=torch.int64, device=self._accum_tensor.device)
        dist.reduce(n, dst=0, op=dist.ReduceOp.SUM, group=self.process_group)
        self._reduced_passes = int(n.item())

    # ---- RenderJob.cs:646-654, 679-688 ---------------------------------------------------------------------------
    def UpdateImage(self, total_samples: Optional[int] = None):
        if total_samples is None:
            capi.check(self._lib.rt_render_read_rgb24(self.handle, self.ImageBuffer.ctypes.data))
        else:
            import torch
            torch.cuda.synchronize()
            capi.check(self._lib.rt_render_resolve_device(self.handle, C.c_void_p(self._accum_tensor.data_ptr()), total_samples, self.ImageBuffer.ctypes.data))
        return self.ImageBuffer

    @property
    def Image(self) -> np.ndarray:
        return self.ImageBuffer

    def ReadAccum(self):
        """rawColourBuffer (H, W, 3) float32 sums in image-row order and the sample count (RenderJob.cs:669-672)."""
        W, H = self.RenderOptions.RenderWidth, self.RenderOptions.RenderHeight
        buf = np.zeros((H, W, 3), dtype=np.float32)
        n = C.c_uint32()
        capi.check(self._lib.rt_render_read_accum(self.handle, buf.ctypes.data, C.byref(n)))
        return buf, int(n.value)

    def UpdateStats(self, reduce: bool = False) -> RenderStats:
        """RenderStats from the device counters.  reduce=True (all ranks must call): the counters of every rank summed onto rank 0,
        so that rank 0's RenderStats describes the whole job as the reference's single RenderStats does."""
        st = capi.Stats()
        capi.check(self._lib.rt_render_stats(self.handle, C.byref(st)))
        if reduce and self.world_size > 1:
            import torch
            import torch.distributed as dist
            dev = self._accum_tensor.device
            raw = np.frombuffer(bytes(st), dtype=np.uint64).astype(np.int64)
            t = torch.from_numpy(raw.copy()).to(dev)
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM, group=self.process_group)
            if self.rank == 0:
                C.memmove(C.byref(st), t.cpu().numpy().astype(np.uint64).tobytes(), C.sizeof(st))
        s = self.RenderStats
        s.PixelsRendered, s.PassesRendered = st.pixels_rendered, st.passes_rendered
        s.MaterialScatterCount, s.MaterialAbsorbedCount = st.material_scatter_count, st.material_absorbed_count
        s.AabbMisses, s.HittableMisses, s.HittableIntersections = st.aabb_misses, st.hittable_misses, st.hittable_intersections
        s.SkyRays, s.BounceLimitExceeded, s.RayCount, s.ShadowRays = st.sky_rays, st.bounce_limit_exceeded, st.ray_count, st.shadow_rays
        s.NodeVisits, s.PrimTests, s.ShadowNodeVisits, s.ShadowPrimTests = st.node_visits, st.prim_tests, st.shadow_node_visits, st.shadow_prim_tests
        s.NormalsWrongMagnitude, s.UVInvalid = st.normals_wrong_magnitude, st.uv_invalid
        s.RayDepthCounts = [int(x) for x in st.ray_depth_counts]
        return s

    def ValidatorCounts(self):
        """GraphicsValidator.Errors for this job (GraphicsValidator.cs:18): (normals_wrong, uv_invalid), one uint64 per scene object
        (RenderOptions.Validate renders only)."""
        n = self.device_scene.flat.n_objects
        nw, uv = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        capi.check(self._lib.rt_render_validator_counts(self.handle, nw.ctypes.data, uv.ctypes.data, n))
        return nw, uv

    def SetProfiling(self, enable: bool):
        capi.check(self._lib.rt_render_set_profiling(self.handle, 1 if enable else 0))

    def KernelProfile(self) -> dict:
        kp = capi.KernelProfile()
        capi.check(self._lib.rt_render_kernel_profile(self.handle, C.byref(kp)))
        return kp.as_dict()

    # ---- RenderJob.cs:733 -------------------------------------------------------------------------------------------
    def Dispose(self):
        if self.handle:
            self._lib.rt_render_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass
