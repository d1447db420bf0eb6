"""The N > 1 path of RenderJob on a 1-GPU box: two ranks (both on cuda:0, gloo standing in for RCCL) shard the passes, reduce the
device-resident colour sums to rank 0 and resolve there; the result equals the single-process render up to float-sum association."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, PASSES = 160, 90, 6


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracer_amd as rt
    from raytracer_amd.core import RenderOptions
    sc = rt.BuiltinScenes.Testing()
    opt = RenderOptions(RenderWidth=W, RenderHeight=H, Passes=PASSES, MaxBounceDepth=6, Seed=4242)
    job = rt.RenderJob(sc, opt, device=0, rank=rank, world_size=world)
    job.Wait()                                   # RenderInternal: own passes -> ReduceToRoot -> (rank 0) resolve
    st = job.RenderStats                         # RenderInternal sum-reduced every rank's counters onto rank 0 (UpdateStats(reduce=True))
    own = job._passes_done * W * H
    if rank == 0:
        acc = job._accum_tensor.cpu().numpy().reshape(H, W, 3)
        np.savez(out_path, acc=acc, img=job.Image, pixels=np.array([st.PixelsRendered]), rays=np.array([st.RayCount]), passes=np.array([st.PassesRendered]))
    else:
        assert st.PixelsRendered == own          # the other ranks keep their own counts
    dist.barrier()
    job.Dispose()
    dist.destroy_process_group()


def test_two_ranks_reduce_to_the_single_process_image(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    import raytracer_amd as rt
    from raytracer_amd.core import RenderOptions
    opt = RenderOptions(RenderWidth=W, RenderHeight=H, Passes=PASSES, MaxBounceDepth=6, Seed=4242)
    job = rt.RenderJob(rt.BuiltinScenes.Testing(), opt); job.Wait()
    ref, n = job.ReadAccum()
    got = np.load(out)
    assert int(got["pixels"][0]) == W * H * PASSES == job.RenderStats.PixelsRendered       # rank 0's RenderStats describes the whole job
    assert int(got["rays"][0]) == job.RenderStats.RayCount and int(got["passes"][0]) == PASSES
    assert np.allclose(got["acc"], ref, rtol=1e-5, atol=1e-6)
    assert (np.abs(got["img"].astype(int) - job.Image.astype(int)) <= 1).all()
    job.Dispose()
