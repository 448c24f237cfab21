"""CPU test (-m "not gpu") of the multi-GPU path's host logic: row-slab partitioning and the neighbour halo
exchange, world_size 2 and 3 over gloo.  (On the GPUs the same code runs over RCCL; the filter itself needs
a device and is covered by the -m gpu row-slab parity test.)"""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, S, halo, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import rpf_pkg
    rpf_pkg.load()
    import torch
    import torch.distributed as dist
    from raytracer_rpf_amd import feature_buffer as fb
    from raytracer_rpf_amd import slabs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.from_numpy(fb.synth_planes(W, H, S, seed=17))[2:5].to(torch.float64)
        slab = slabs.slab_for(H, world, rank, halo)
        H_buf, rb, re = slabs.buffer_rows(slab)
        buf = torch.full((3, H_buf, W, S), -1.0, dtype=torch.float64)
        buf[:, rb:re] = full[:, slab.row0:slab.row1]          # each rank knows only its own rows
        slabs.exchange_halo(buf, slab, rank, world)           # halo rows arrive from the neighbours
        want = full[:, slab.row0 - slab.halo_top:slab.row1 + slab.halo_bottom]
        ok = bool(torch.equal(buf, want))
        # the preallocated plan (what bench.py drives every step) moves the same rows, twice in a row
        buf2 = torch.full((3, H_buf, W, S), -1.0, dtype=torch.float64)
        buf2[:, rb:re] = full[:, slab.row0:slab.row1]
        plan = slabs.HaloPlan(buf2, slab, rank, world)
        for _ in range(2):
            plan.exchange(buf2)
            ok = ok and bool(torch.equal(buf2, want))
        q.put((rank, ok, (slab.row0, slab.row1, slab.halo_top, slab.halo_bottom)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 13), (3, 20)])
def test_halo_exchange_over_gloo(world, H):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, 6, 2, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    rows = [r[2] for r in res]
    assert rows[0][0] == 0 and rows[-1][1] == H and all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
    assert rows[0][2] == 0 and rows[-1][3] == 0 and all(r[2] == 3 for r in rows[1:])


def test_partition_arithmetic():
    from raytracer_rpf_amd import slabs
    for H in (7, 1080, 2160, 8640):
        for world in (1, 2, 3, 4, 8):
            parts = [slabs.partition_rows(H, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == H
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
    s = slabs.slab_for(2160, 8, 0, 3)
    assert (s.halo_top, s.halo_bottom) == (0, 3) and slabs.buffer_rows(s) == (273, 0, 270)
    s = slabs.slab_for(2160, 8, 4, 3)
    assert slabs.buffer_rows(s) == (276, 3, 273)
    # a slab thinner than the halo its neighbours need cannot be served from owned rows: rejected, not silently wrong
    with pytest.raises(ValueError):
        slabs.slab_for(40, 8, 3, 27)   # box 55 needs 27 halo rows, 40/8 = 5 rows per rank
    assert slabs.slab_for(40, 1, 0, 27) == slabs.Slab(0, 40, 0, 0)


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must spawn its ranks itself (a child
    torch.distributed.run, before anything touches a GPU) and print rank 0's one JSON line; --rendezvous-only stops after
    the gloo rendezvous + an all-reduce so the launcher path runs here, without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--rendezvous-only"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rendezvous"] == "ok" and d["sum_of_ranks_plus_one"] == 3.0
