#!/usr/bin/env python3
"""Rehearsal of the multi-rank path on however many GPUs the box has (ranks share GPUs; gloo moves the halo rows):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 \
        scripts/multirank_check.py

Every rank builds ITS slab (+ halo rows) of one global synthetic frame, runs the box list {7, 5} with a colour-halo
exchange before each pass (exactly bench.py's step), and rank 0 compares the gathered owned rows with one process
filtering the whole frame: the two must be bit-identical.  With nccl on one GPU per rank the same code runs over xGMI."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import rpf_pkg

rpf_pkg.load()
from raytracer_rpf_amd import feature_buffer as fb, hip, slabs

W, H, S = 257, 96, int(os.environ.get("SPP", "8"))
boxes = (7, 5)
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
backend = os.environ.get("RPF_DIST_BACKEND", "gloo")
local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
ctx = hip.Context(local)
xp = fb.torch_backend(dev)
halo = fb.halo_rows(max(boxes))
slab = slabs.slab_for(H, world, rank, halo)
H_buf, r0, r1 = slabs.buffer_rows(slab)
planes = fb.synth_planes(W, H_buf, S, row0=slab.row0 - slab.halo_top, xp=xp, mode="clustered", sigma_f=1e-3, sigma_c=0.01).contiguous()
colour = planes[2:5].to(torch.float64).contiguous()
for box in boxes:
    slabs.exchange_halo(colour, slab, rank, world)
    d = hip.make_desc(W, H_buf, S, boxes=(box,), row_begin=r0, row_end=r1, policy=hip.DEGEN_EPS)
    ctx.filter_device(d, planes.data_ptr(), colour.data_ptr(), torch.cuda.current_stream().cuda_stream)
own = colour[:, r0:r1].cpu()
if world > 1:
    parts = [None] * world
    dist.gather_object(own, parts if rank == 0 else None, dst=0)
else:
    parts = [own]
if rank == 0:
    got = torch.cat(parts, dim=1)
    full = fb.synth_planes(W, H, S, xp=xp, mode="clustered", sigma_f=1e-3, sigma_c=0.01).contiguous()
    c = full[2:5].to(torch.float64).contiguous()
    ctx.filter_device(hip.make_desc(W, H, S, boxes=boxes, policy=hip.DEGEN_EPS), full.data_ptr(), c.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
    same = torch.equal(got, c.cpu())
    act = float((c.cpu() - full[2:5].cpu().double()).norm() / c.cpu().norm())
    print("multirank_check: world %d backend %s %dx%dx%d boxes %s: slabs == full frame bit-for-bit: %s (filter activity %.2e)"
          % (world, backend, W, H, S, boxes, same, act), flush=True)
    if not same:
        sys.exit(1)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
