"""RCCL sanity on one GPU (world_size 1): init_process_group('nccl'), barrier, reduce and all_reduce of a device tensor — the calls bench.py and
RenderJob make at N > 1 (their N > 1 path is rehearsed over gloo on this box; 8 RCCL ranks are the driver's to launch)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
dist.barrier(); dist.reduce(t, dst=0); dist.all_reduce(t); torch.cuda.synchronize()
print("rccl", torch.cuda.nccl.version(), "ok", float(t[12345]))
dist.destroy_process_group()
