"""GPU parity at BASELINE.json's full scene sizes (configs 2-5).

The scenes are the full-size ones (485 objects / Cornell box / 4 999 122 triangles / 100 000 spheres + 64 lights); what is bounded
is the number of rays and samples the CPU oracle has to trace so that each test stays well under two minutes:
  * fixed ray sets against the oracle on the full scene: object and material ids bit-exact, K <= 1e-5 relative, any-hit booleans equal;
  * an analytic class of rays whose answer needs no oracle (vertical rays onto the height-field mesh: the triangle is known from the cell);
  * images at the configured resolution against the oracle at the same (seed, pixel, pass) keys where the oracle can afford it,
    and size-independent properties where it cannot: every sample accounted for, partition independence of pass sharding,
    batching invariance bit for bit, finite sums, terminal events add up.
"""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs
from raytracer_amd.core import RenderOptions
from raytracer_amd.flatten import flatten_scene
from oracle import binding as ob

import helpers

pytestmark = pytest.mark.gpu
NO_HIT = 0xFFFFFFFF


def _ray_parity(sc, flat, dev, orc, n_closest, n_list, seed, n_any=120_000):
    rs = np.random.default_rng(seed)
    o1, d1 = helpers.camera_rays(sc, 64, 64, n_closest, rs)
    o2, d2 = helpers.secondary_rays(flat, n_closest, rs)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    g = dev.TraceClosest(o, d, 0.001, np.inf, seed=7)
    r = orc.trace_closest(o, d, 0.001, np.inf, seed=7)
    helpers.assert_closest_parity(g, r)
    hit = r["object"] != NO_HIT
    assert hit.sum() > n_closest // 4
    # Any-hit.  The reference asks every object for every shadow ray (RenderJob.cs:575, BinaryBvhNode.cs:41): seconds per ray on these
    # scenes.  So (1) the oracle's every-object scan and its culled form (oracle.cpp CullTree) are compared on n_list rays of THIS scene
    # (tests/test_oracle_culled_any.py does it with 100 k rays on every small scene), then (2) the device answers n_any (>= 100 k)
    # rays of the kinds a render traces — camera rays, incoherent rays, rays from surface points towards the lights — against the
    # culled form, bit for bit, through k_trace_any.
    oa, da, kmin, kmax = helpers.shadow_ray_set(sc, flat, orc, n_any, seed + 100)
    ref = orc.trace_any(oa, da, kmin, kmax, threads=16, culled=True)
    sub = rs.choice(len(oa), n_list, replace=False)
    assert np.array_equal(orc.trace_any(oa[sub], da[sub], kmin[sub], kmax[sub], threads=16), ref[sub])
    got = dev.TraceAny(oa, da, kmin, kmax)
    assert np.array_equal(got, ref), f"{(got != ref).sum()} of {len(ref)} any-hit answers differ from the oracle's"
    assert 0.02 < ref.mean() < 0.98
    return int(hit.sum())


def _render_vs_culled_oracle(sc, dev_flat, orc, opt, min_close, monkeypatch):
    """An image at a size the oracle can only afford with culled shadow rays, in exact-lights mode (every shadow ray bit-identical),
    through each of the three shadow kernels (RT_HIP_PERSISTENT_SHADOW forced off / on, RT_HIP_SHADOW_RAYS forced on)."""
    orc.set_culled_any(True)
    ref, rst = orc.render(opt, threads=16)
    orc.set_culled_any(False)
    # every shadow kernel the library has for this scene: request lanes (k_shadow), persistent request lanes, ray lanes (k_shadow_rays)
    for var, forced, what in (("RT_HIP_PERSISTENT_SHADOW", "0", "k_shadow"), ("RT_HIP_PERSISTENT_SHADOW", "1", "k_shadow_persistent"), ("RT_HIP_SHADOW_RAYS", "1", "k_shadow_rays")):
        monkeypatch.setenv(var, forced)
        dev = rt.DeviceScene(dev_flat)
        monkeypatch.delenv(var)
        job = rt.RenderJob(sc, opt, device_scene=dev); job.Wait(); got, _ = job.ReadAccum(); st = job.RenderStats; job.Dispose(); dev.Dispose()
        close = _image_close(got, ref)
        assert close >= min_close, f"{what}: only {close:.5f} of pixels match"
        assert (st.RayCount, st.ShadowRays) == (rst["ray_count"], rst["shadow_rays"]), what


def _image_close(got, ref):
    fin = np.isfinite(ref).all(axis=2) & np.isfinite(got).all(axis=2)
    assert fin.mean() > 0.999
    err = np.abs(got - ref)[fin]
    scale = np.maximum(np.abs(ref)[fin], 1e-3)
    close = (err <= 1e-3 * scale + 1e-5).all(axis=1).mean()
    for c in range(3):
        m_ref, m_got = ref[fin][:, c].mean(), got[fin][:, c].mean()
        assert abs(m_got - m_ref) <= 0.005 * max(abs(m_ref), 1e-6) + 1e-6, f"channel {c}: {m_got} vs {m_ref}"
    return close


def _properties(sc, dev, opt):
    """What must hold at any size: accounting, partition independence, batching invariance."""
    W, H, P = opt.RenderWidth, opt.RenderHeight, opt.Passes
    job = rt.RenderJob(sc, opt, device_scene=dev); job.Wait()
    a, n = job.ReadAccum()
    st = job.RenderStats
    assert n == P and st.PixelsRendered == W * H * P and st.PassesRendered == P
    assert st.SkyRays + st.BounceLimitExceeded == st.PixelsRendered            # every sample ends in exactly one terminal event (no builtin material absorbs)
    assert st.RayCount >= st.PixelsRendered and st.RayCount == st.PixelsRendered + st.MaterialScatterCount - st.BounceLimitExceeded
    assert np.isfinite(a).all() and (a >= 0).all()
    img = job.Image
    assert img.shape == (H, W, 3) and img.dtype == np.uint8
    job.Dispose()
    parts = []
    for r in range(2):                                                          # what two GPUs would render
        j = rt.RenderJob(sc, opt, device_scene=dev)
        j.RenderPasses(*rt.shard_passes(P, r, 2))
        parts.append(j.ReadAccum()[0]); j.Dispose()
    assert np.allclose(parts[0] + parts[1], a, rtol=1e-5, atol=1e-6)
    opt1 = RenderOptions(RenderWidth=W, RenderHeight=H, Passes=P, MaxBounceDepth=opt.MaxBounceDepth, LightSampleCountHint=opt.LightSampleCountHint,
                         Seed=opt.Seed, PassesPerBatch=1)
    j = rt.RenderJob(sc, opt1, device_scene=dev); j.Wait(); b, _ = j.ReadAccum(); j.Dispose()
    assert np.array_equal(a, b)                                                 # batches of 1 pass on alternating streams: not a bit changes
    return a, st


def test_config2_night_spheres_1080p():
    sc = bs.Testing(); flat = flatten_scene(sc); dev, orc = rt.DeviceScene(flat), ob.OracleScene(flat)
    assert dev.Info()["variant"] == "spheres" and dev.Info()["tree_in_lds"]
    _ray_parity(sc, flat, dev, orc, 100_000, 20_000, 21)
    opt = RenderOptions(RenderWidth=1920, RenderHeight=1080, Passes=2, MaxBounceDepth=10, LightSampleCountHint=2, Seed=0xC2)
    got, st = _properties(sc, dev, opt)
    ref, rst = orc.render(opt, threads=16)
    assert _image_close(got, ref) > 0.99
    opt_x = RenderOptions(RenderWidth=1920, RenderHeight=1080, Passes=2, MaxBounceDepth=10, LightSampleCountHint=2, Seed=0xC2, ExactLightSampling=True, ExactOcclusion=True)
    job = rt.RenderJob(sc, opt_x, device_scene=dev); job.Wait(); gx, _ = job.ReadAccum(); job.Dispose()
    assert _image_close(gx, ref) >= 0.999                                         # 2 M pixels, every shadow ray identical to the oracle's
    assert abs(st.RayCount - rst["ray_count"]) <= 0.002 * rst["ray_count"] and abs(st.ShadowRays - rst["shadow_rays"]) <= 0.002 * rst["shadow_rays"]


def test_config1_demo_as_stated():
    """BASELINE config 1 exactly as stated — BuiltinScenes.Demo, 480 x 270, 64 passes, depth 8, 2 light samples (8.3 M samples) — against
    the oracle path by path in the exact modes: >= 99.9 % of the pixels within 1e-3, every counter of rays equal."""
    sc = bs.Demo(); flat = flatten_scene(sc); dev, orc = rt.DeviceScene(flat), ob.OracleScene(flat)
    opt = RenderOptions(RenderWidth=480, RenderHeight=270, Passes=64, MaxBounceDepth=8, LightSampleCountHint=2, Seed=0xC1, ExactLightSampling=True, ExactOcclusion=True)
    job = rt.RenderJob(sc, opt, device_scene=dev); job.Wait(); got, n = job.ReadAccum(); st = job.RenderStats; job.Dispose()
    ref, rst = orc.render(opt, threads=16)
    assert n == 64 and st.PixelsRendered == 480 * 270 * 64
    assert _image_close(got, ref) >= 0.999
    assert (st.RayCount, st.ShadowRays) == (rst["ray_count"], rst["shadow_rays"])
    dev.Dispose()


@pytest.mark.parametrize("intensity,direct", [(0.1, False), (0.1, True), (1.0, False), (1.0, True)])      # the four variants of Renders/Refractive.../info.md:8-9
def test_config3_cornell_depth32_1080p(intensity, direct):
    sc = bs.CornellBox(intensity, direct); flat = flatten_scene(sc); dev, orc = rt.DeviceScene(flat), ob.OracleScene(flat)
    assert dev.Info()["variant"] == "boxes" and dev.Info()["object_list"]
    _ray_parity(sc, flat, dev, orc, 50_000, 20_000, 22)
    opt = RenderOptions(RenderWidth=1920, RenderHeight=1080, Passes=2, MaxBounceDepth=32, LightSampleCountHint=2, Seed=0xC3)
    got, st = _properties(sc, dev, opt)
    assert st.RayDepthCounts[32] > 0                                             # dielectric paths do reach the bounce limit
    opt_s = RenderOptions(RenderWidth=480, RenderHeight=270, Passes=4, MaxBounceDepth=32, LightSampleCountHint=2, Seed=0xC3)
    job = rt.RenderJob(sc, opt_s, device_scene=dev); job.Wait(); g, _ = job.ReadAccum(); job.Dispose()
    ref, _ = orc.render(opt_s, threads=16)
    assert _image_close(g, ref) > 0.97
    # the full-size frame in the exact modes: one pass of 1920 x 1080 at depth 32, path by path against the oracle
    opt_x = RenderOptions(RenderWidth=1920, RenderHeight=1080, Passes=1, MaxBounceDepth=32, LightSampleCountHint=2, Seed=0xC3, ExactLightSampling=True, ExactOcclusion=True)
    job = rt.RenderJob(sc, opt_x, device_scene=dev); job.Wait(); gx, _ = job.ReadAccum(); sx = job.RenderStats; job.Dispose()
    refx, rsx = orc.render(opt_x, threads=16)
    assert _image_close(gx, refx) >= 0.999
    assert (sx.RayCount, sx.ShadowRays) == (rsx["ray_count"], rsx["shadow_rays"])
    dev.Dispose()


def test_config4_five_million_triangles(monkeypatch):
    sc = bs.TriangleMesh(); flat = flatten_scene(sc)
    assert flat.n_objects == 4_999_122
    dev = rt.DeviceScene(flat)
    info = dev.Info()
    assert info["variant"] == "mesh" and not info["tree_in_lds"] and info["n_nodes"] == flat.n_objects - 1
    # (1) analytic: a vertical ray over cell (i, j) hits that cell's lower or upper triangle; ids follow TriangleArray's order
    grid = 1581
    rs = np.random.default_rng(4)
    n = 200_000
    fx, fz = rs.uniform(0.02, 0.98, n), rs.uniform(0.02, 0.98, n)
    ci, cj = rs.integers(0, grid, n), rs.integers(0, grid, n)
    xs = np.linspace(-10.0, 10.0, grid + 1, dtype=np.float32).astype(np.float64)
    x = xs[ci] + fx * (xs[ci + 1] - xs[ci]); z = xs[cj] + fz * (xs[cj + 1] - xs[cj])
    keep = np.abs(fx + fz - 1.0) > 0.02                                          # stay off the shared diagonal
    o = np.stack([x, np.full(n, 50.0), z], axis=1).astype(np.float32)[keep]
    d = np.broadcast_to(np.array([0, -1, 0], np.float32), o.shape).copy()
    cell = (ci * grid + cj)[keep]
    upper = (fx + fz > 1.0)[keep]                                                # p00 p01 p10 is the lower triangle, p11 p10 p01 the upper one
    want = np.where(upper, grid * grid + cell, cell).astype(np.uint32)
    g = dev.TraceClosest(o, d, 0.001, np.inf)
    assert np.array_equal(g["object"], want), f"{(g['object'] != want).sum()} of {len(want)} vertical rays hit the wrong triangle"
    assert (np.abs(g["world_point"][:, 1]) < 1.2).all() and np.allclose(g["k"], 50.0 - g["world_point"][:, 1], atol=1e-4)
    # (2) the oracle on the full mesh
    orc = ob.OracleScene(flat)
    _ray_parity(sc, flat, dev, orc, 20_000, 300, 23)
    # (3) 4K image properties, and the oracle at a size it can afford
    opt = RenderOptions(RenderWidth=3840, RenderHeight=2160, Passes=2, MaxBounceDepth=8, LightSampleCountHint=2, Seed=0xC4)
    _properties(sc, dev, opt)
    # (the oracle's shadow rays scan all 5 M triangles, as the reference's do: a thumbnail is what it can afford)
    opt_s = RenderOptions(RenderWidth=24, RenderHeight=14, Passes=1, MaxBounceDepth=4, LightSampleCountHint=1, Seed=0xC4)
    job = rt.RenderJob(sc, opt_s, device_scene=dev); job.Wait(); gi, _ = job.ReadAccum(); job.Dispose()
    ref, _ = orc.render(opt_s, threads=16)
    assert _image_close(gi, ref) > 0.95
    # with culled shadow rays the oracle affords a real picture of the mesh: 480 x 270, depth 8, both shadow kernels
    dev.Dispose()
    _render_vs_culled_oracle(sc, flat, orc, RenderOptions(RenderWidth=480, RenderHeight=270, Passes=2, MaxBounceDepth=8, LightSampleCountHint=2, Seed=0xC4,
                                                          ExactLightSampling=True, ExactOcclusion=True), 0.999, monkeypatch)


def test_config5_hundred_thousand_spheres_64_lights(monkeypatch):
    sc = bs.ManySpheresManyLights(); flat = flatten_scene(sc); dev, orc = rt.DeviceScene(flat), ob.OracleScene(flat)
    assert flat.n_objects >= 100_000 and flat.n_lights == 64 and dev.Info()["variant"] == "spheres" and not dev.Info()["tree_in_lds"]
    _ray_parity(sc, flat, dev, orc, 50_000, 3_000, 24)
    opt = RenderOptions(RenderWidth=3840, RenderHeight=2160, Passes=1, MaxBounceDepth=10, LightSampleCountHint=1, Seed=0xC5, PassesPerBatch=1)
    job = rt.RenderJob(sc, opt, device_scene=dev); job.Wait()
    a, n = job.ReadAccum(); st = job.RenderStats
    assert st.PixelsRendered == 3840 * 2160 and st.SkyRays + st.BounceLimitExceeded == st.PixelsRendered
    assert st.ShadowRays == 64 * st.MaterialScatterCount                         # 64 lights x 1 sample per shaded hit (no volumetric material here)
    assert np.isfinite(a).all()
    job.Dispose()
    # the oracle scans all 100 000 objects for each of the 64 shadow rays per hit: a thumbnail is what it can afford
    opt_s = RenderOptions(RenderWidth=32, RenderHeight=18, Passes=1, MaxBounceDepth=4, LightSampleCountHint=1, Seed=0xC5)
    j = rt.RenderJob(sc, opt_s, device_scene=dev); j.Wait(); g, _ = j.ReadAccum(); j.Dispose()
    ref, _ = orc.render(opt_s, threads=16)
    assert _image_close(g, ref) > 0.95
    # with culled shadow rays: 320 x 180 at the configured depth (64 shadow rays per hit: 30 M of them), both shadow kernels
    dev.Dispose()
    _render_vs_culled_oracle(sc, flat, orc, RenderOptions(RenderWidth=320, RenderHeight=180, Passes=2, MaxBounceDepth=10, LightSampleCountHint=1, Seed=0xC5,
                                                          ExactLightSampling=True, ExactOcclusion=True), 0.999, monkeypatch)


@pytest.mark.parametrize("builder", ["device", "ploc", "sah"])
def test_config5_device_built_trees_give_the_host_trees_hits(builder, monkeypatch):
    """100 000 spheres: the trees built on the GPU (rt_lbvh.hip, rt_ploc.hip, rt_sah.hip) against the host's SAH tree — 200 k closest hits bit for
    bit (K included: the leaf boxes and intersectors are the same, only the order of the tests differs) and 200 k any-hit answers."""
    sc = bs.ManySpheresManyLights(); flat = flatten_scene(sc)
    host = rt.DeviceScene(flat)
    monkeypatch.setenv("RT_HIP_BVH", builder)
    dev = rt.DeviceScene(flat)
    monkeypatch.delenv("RT_HIP_BVH")
    assert dev.Info()["built_on_device"] and not host.Info()["built_on_device"] and dev.Info()["n_nodes"] == host.Info()["n_nodes"]
    rs = np.random.default_rng(55)
    o1, d1 = helpers.camera_rays(sc, 64, 64, 100_000, rs)
    o2, d2 = helpers.secondary_rays(flat, 100_000, rs)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    a, b = dev.TraceClosest(o, d, 0.001, np.inf, seed=3), host.TraceClosest(o, d, 0.001, np.inf, seed=3)
    helpers.assert_closest_parity(a, b, flat=flat, rays=(o, d))
    kmin = np.full(len(o), 0.01, np.float32); kmax = rs.uniform(0.5, 60.0, len(o)).astype(np.float32)
    assert np.array_equal(dev.TraceAny(o, d, kmin, kmax), host.TraceAny(o, d, kmin, kmax))
    if builder == "sah":
        # the SAH builder is the default for big scenes because its tree costs rays what the host's costs them (VERDICT r01 item 8:
        # within 5 %): box tests per closest ray and per shadow ray of a small render, both trees
        opt = RenderOptions(RenderWidth=320, RenderHeight=180, Passes=1, MaxBounceDepth=10, LightSampleCountHint=1, Seed=5)
        per_ray = []
        for scene in (dev, host):
            job = rt.RenderJob(sc, opt, device_scene=scene); job.Wait(); st = job.UpdateStats(); job.Dispose()
            per_ray.append((st.NodeVisits / st.RayCount, st.ShadowNodeVisits / st.ShadowRays))
        assert per_ray[0][0] <= 1.05 * per_ray[1][0] and per_ray[0][1] <= 1.05 * per_ray[1][1], per_ray
    dev.Dispose(); host.Dispose()


def test_check1_two_million_rays_night_spheres():
    """SURVEY §8d check 1 at its stated size: >= 1 M primary + 1 M secondary rays from the oracle on config 2's scene."""
    sc = bs.Testing(); flat = flatten_scene(sc); dev, orc = rt.DeviceScene(flat), ob.OracleScene(flat)
    rs = np.random.default_rng(2_000_000)
    o1, d1 = helpers.camera_rays(sc, 1920, 1080, 1_000_000, rs)
    o2, d2 = helpers.secondary_rays(flat, 1_000_000, rs)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    g = dev.TraceClosest(o, d, 0.001, np.inf, seed=3)
    r = orc.trace_closest(o, d, 0.001, np.inf, seed=3)
    excluded = helpers.assert_closest_parity(g, r)
    assert len(excluded) <= 20                                                    # "expected ~ 0"
    assert (r["object"] != NO_HIT).sum() > 500_000


@pytest.mark.parametrize("name,w,h,passes,depth", [("Testing", 480, 270, 256, 10), ("Demo", 480, 270, 256, 8), ("CornellBox", 480, 270, 256, 32)])
def test_check2_rmse_against_the_oracles_noise_floor(name, w, h, passes, depth):
    """SURVEY §8d check 2 — converged images at high spp: 480 x 270 at 256 spp (33 M samples a picture; the oracle renders two of them)
    in the default modes.  With DIFFERENT seeds on the two sides (so nothing is compared path by path), the device image's RMSE
    against an oracle image is at most 1.5 x the RMSE between two independent oracle images at the same spp, per channel, and
    the channel means agree within 0.5 % (+ the standard error of the mean at this spp)."""
    sc = helpers.small_scenes()[name](); flat = flatten_scene(sc); dev, orc = rt.DeviceScene(flat), ob.OracleScene(flat)
    def opt(seed): return RenderOptions(RenderWidth=w, RenderHeight=h, Passes=passes, MaxBounceDepth=depth, LightSampleCountHint=2, Seed=seed)
    job = rt.RenderJob(sc, opt(101), device_scene=dev); job.Wait(); got, _ = job.ReadAccum(); job.Dispose()
    ref_a, _ = orc.render(opt(202), threads=16)
    ref_b, _ = orc.render(opt(303), threads=16)
    got, ref_a, ref_b = got / passes, ref_a / passes, ref_b / passes
    fin = np.isfinite(got).all(axis=2) & np.isfinite(ref_a).all(axis=2) & np.isfinite(ref_b).all(axis=2)
    assert fin.mean() > 0.999
    for c in range(3):
        floor = np.sqrt(np.mean((ref_a[fin][:, c] - ref_b[fin][:, c]) ** 2))
        rmse = np.sqrt(np.mean((got[fin][:, c] - ref_a[fin][:, c]) ** 2))
        assert rmse <= 1.5 * floor + 1e-6, f"channel {c}: RMSE {rmse} vs noise floor {floor}"
        sem = floor / np.sqrt(fin.sum())
        m_ref, m_got = ref_a[fin][:, c].mean(), got[fin][:, c].mean()
        assert abs(m_got - m_ref) <= 0.005 * abs(m_ref) + 4 * sem + 1e-6, f"channel {c}: mean {m_got} vs {m_ref}"
