#!/usr/bin/env python3
"""The sharded march across REAL ranks (one GPU each, RCCL): sdfv_slab_march over the library communicator -- a 1080p frame of
the z-sharded 256^3 * N grid marched where it lies, timed, and compared bit for bit with the single-GPU march over the gathered
grid.  Step 6 of tools/first_contact.sh:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/sharded_march_ranks.py [side=256]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
pkg = importlib.import_module("sdf-viewer_amd")
par = importlib.import_module("sdf-viewer_amd.parallel")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(device)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
dist.all_reduce(torch.empty(1, device=device))
W, H = 1920, 1080
gdims = par.weak_scaling_dims(side, world, "slab")
prm = pkg.default_params()
slab = par.alloc_slab(gdims, rank, world, device)
grid = pkg.make_grid(gdims, z_begin=slab.z_begin, z_end=slab.z_end)
par.enter_stage("sharded_march_ranks: SlabFiller(transport=rccl) + first step")
filler = par.SlabFiller(pkg, prm, gdims, slab, rank, world, transport="rccl")
filler.step()
torch.cuda.synchronize()
ggrid = pkg.make_grid(gdims)
rp = pkg.default_render_params(ggrid)
cam = pkg.camera_look_at(eye=(1.5, 2.0, 3.5), aspect=W / H)
par.enter_stage("sharded_march_ranks: first sharded frame")
got = par.raymarch_sharded(pkg, rp, grid, slab, cam, W, H, rank, world, comm=filler.comm)
torch.cuda.synchronize()
n = 10
dist.barrier()
t = time.perf_counter()
for _ in range(n):
    got = par.raymarch_sharded(pkg, rp, grid, slab, cam, W, H, rank, world, comm=filler.comm)
torch.cuda.synchronize()
dist.barrier()
ms = (time.perf_counter() - t) / n * 1e3
par.enter_stage("sharded_march_ranks: gather_replica + whole-grid march")
full0, full1 = par.gather_replica(slab, gdims, world, comm=filler.comm)
want = pkg.raymarch(rp, full0, full1, cam, W, H)[0]
t = time.perf_counter()
for _ in range(n):
    pkg.raymarch(rp, full0, full1, cam, W, H)
torch.cuda.synchronize()
whole_ms = (time.perf_counter() - t) / n * 1e3
same = torch.equal(got.view(torch.int32), want.view(torch.int32)) and bool((want[..., 3] > 0).any())
flag = torch.tensor([1.0 if same else 0.0], device=device)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print(json.dumps({"what": "sdfv_slab_march over the library RCCL communicator, one rank per GPU", "ranks": world, "grid": list(gdims),
                      "image": [W, H], "ms_per_frame": round(ms, 4), "Mrays_s": round(W * H / ms / 1e3, 1),
                      "whole_grid_march_on_one_gpu_ms": round(whole_ms, 4), "verified": bool(flag.item() == 1.0)}), flush=True)
torch.cuda.synchronize()
filler.comm.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if flag.item() == 1.0 else 1)
