#!/usr/bin/env python3
"""bench.py -- RPF filter-pass throughput on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one filter pass (box 7: FillMeanAndStddev + the fused per-pixel kernel, rpf.cpp:497-733) over one
synthetic feature buffer that is already resident in HBM.  N = 1: BASELINE.json configs[1], 1920x1080x8spp.
N > 1: weak scaling -- the image is 1920 x (1080*N) rows, row-tiled one 1080-row slab per rank; every step
re-exchanges the colour halo rows with the neighbouring ranks (RCCL send/recv) and then filters the slab.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_SAMPLE = 88.0   # 19 fp32 read + 3 fp32 written per sample per pass


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--rows-per-gpu", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--box", type=int, default=7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fast-weights", action="store_true", help="opt-in fp32 pair weights (RPF_FLAG_FAST_WEIGHTS)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="rpf_set_option override (diagnostics: stage_mask, binning, waves_per_pixel, table_in_lds, lds_pad)")
    ap.add_argument("--allow-nonfinite", action="store_true", help="profiling variants whose results are wrong on purpose")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) is the measured path; gloo lets several ranks share one GPU to "
                         "rehearse the multi-rank code path (halo rows staged through the host)")
    return ap.parse_args()


def load_traffic(workload_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc pass (profiles/)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload_key)
    except (OSError, ValueError):
        return None


def cpu_baseline(torch, planes_dev, slab_rows, W, S, box, target_s, gpu_colour_dev):
    """Oracle (CPU port) timed on a bounded sample of the same workload: R full-width rows (+ halo rows) cut
    out of the very buffer the GPU filtered; also reports the GPU/oracle rel-L2 on those rows."""
    import numpy as np
    import pyoracle as O
    O.build()
    b = (box - 1) // 2
    cores = min(16, len(os.sched_getaffinity(0)))  # a 1-GPU box's CPU share
    r0 = min(400, max(b, slab_rows // 3))

    def run(R):
        lo, hi = r0 - b, r0 + R + b
        sub = planes_dev[:, lo:hi].contiguous().cpu().numpy()
        d = O.make_desc(W, hi - lo, S, box=box, row_begin=b, row_end=b + R, n_threads=cores)
        t = time.perf_counter()
        r = O.filter_pass(sub, d, debug=False)
        dt = time.perf_counter() - t
        return r, dt, lo

    R = 8
    r, dt, lo = run(R)
    rate = R * W * S / dt
    R2 = int(max(4, min(slab_rows - r0 - b - 1, 512, target_s * rate / (W * S))))
    if R2 > R:
        R = R2
        r, dt, lo = run(R)
    got = gpu_colour_dev[:, r0:r0 + R].cpu().numpy()
    want = r["colour"][:, b:b + R]
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    return {"value": R * W * S / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d full-width rows (%d samples, mean N %.0f) of the same buffer, oracle/rpf_oracle.c "
                      "fp64 OpenMP, %.1f s" % (R, R * W * S, r["sum_nbhd"] / (R * W), dt),
            "gpu_vs_oracle_rel_l2": rel}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import rpf_pkg
    rpf_pkg.load()
    from raytracer_rpf_amd import feature_buffer as fb
    from raytracer_rpf_amd import hip, slabs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    if args.dist_backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    W, S, box = args.width, args.spp, args.box
    H_total = args.rows_per_gpu * world
    halo = fb.halo_rows(box)
    slab = slabs.slab_for(H_total, world, rank, halo)
    H_buf, row_begin, row_end = slabs.buffer_rows(slab)

    # synthetic feature buffer generated directly in HBM; halo rows come from the generator too (setup,
    # untimed) -- they are what the neighbour rank generates for those image rows
    xp = fb.torch_backend(dev)
    planes = fb.synth_planes(W, H_buf, S, row0=slab.row0 - slab.halo_top, xp=xp, mode="smooth",
                             sigma_f=0.05, sigma_c=1e-4).contiguous()
    colour0 = planes[2:5].to(torch.float64).contiguous()
    colour = colour0.clone()
    ctx = hip.Context(local_rank)
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    flags = hip.FLAG_TIMING | (hip.FLAG_FAST_WEIGHTS if args.fast_weights else 0)
    desc = hip.make_desc(W, H_buf, S, boxes=(box,), row_begin=row_begin, row_end=row_end, flags=flags)
    stream = torch.cuda.current_stream().cuda_stream

    kernel_ms = []

    def step():
        colour.copy_(colour0)                                   # the pass input (unfiltered colours)
        slabs.exchange_halo(colour, slab, rank, world)          # RCCL neighbour exchange of the colour halo
        ctx.filter_device(desc, planes.data_ptr(), colour.data_ptr(), stream, allow_nonfinite=args.allow_nonfinite)
        kernel_ms.append(ctx.counters().filter_kernel_ms)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kernel_ms.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    cnt = ctx.counters()
    n_own = slab.row1 - slab.row0
    samples_per_step_all = H_total * W * S
    value = samples_per_step_all * args.steps / elapsed / 1e6
    k_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
    algo_bytes = ALGO_BYTES_PER_SAMPLE * n_own * W * S
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    workload = "synthetic %dx%dx%dspp (smooth, sigma_f=0.05), box %d, 1 pass, 19-dim fp32 planes" % (
        W, args.rows_per_gpu, S, box)
    out = {
        "metric": "RPF Msamples/sec filtered at 1080p×8spp",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64+f32 pair weights" if args.fast_weights else "f64", "data": "synthetic",
        "config": {"workload": workload, "image": "%dx%d" % (W, H_total), "rows_per_gpu": args.rows_per_gpu,
                   "mean_nbhd": cnt.sum_nbhd / float(n_own * W), "max_nbhd": cnt.max_nbhd,
                   "beta_map": "REF_GCC11_O3", "degenerate_policy": "REF_ABORT",
                   "nonfinite_pixels": cnt.nonfinite_pixels,
                   "parallelism": "row slabs x%d, 3-row colour halo over %s send/recv" % (
                       world, "RCCL" if args.dist_backend == "nccl" else "gloo (rehearsal: ranks share GPUs)")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic("%dx%dx%d_box%d" % (W, args.rows_per_gpu, S, box)),
                     "kernel": "filter_pixel_kernel", "kernel_ms": k_ms, "kernel_launches_per_step": cnt.filter_kernel_launches,
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "note": "the kernel is LDS-atomic/fp64-VALU bound (~350 ops/B, SURVEY 8d), not HBM bound; with "
                             "box*box*S > 512 a step is nbhd_count + classify + one filter launch per occupied size class, "
                             "and kernel_ms is their sum"},
    }
    # the pipe that actually bounds the fused kernel: LDS histogram atomics (19 marginal + 96 joint histograms per
    # pixel, one increment per neighbourhood sample each).  Peak = measured ds_add_rtn_u32 rate on random cells
    # (profiles/r01_lds_atomic_microbench.txt: 9.8 LDS cycles per 64 increments per CU at 8 waves/CU) x 256 CUs x 2.4 GHz.
    incr = 115.0 * cnt.sum_nbhd
    lds_peak = 256 * (64.0 / 9.8) * 2.4e9
    out["lds_atomic_roofline"] = {"bound": "lds_atomics", "achieved": incr / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0,
                                  "peak": lds_peak / 1e12, "unit": "T increments/s",
                                  "frac": (incr / (k_ms * 1e-3)) / lds_peak if k_ms > 0 else 0.0,
                                  "note": "whole-kernel time in the denominator; the MI stage alone is ~45% of it (scripts/ablate.sh)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(torch, planes, n_own, W, S, box, args.cpu_seconds, colour)
        if not args.fast_weights:
            # the opt-in fp32 pair-weight mode, measured on the same buffer for information (never `value`)
            ref = colour.clone()
            d2 = hip.make_desc(W, H_buf, S, boxes=(box,), row_begin=row_begin, row_end=row_end,
                               flags=hip.FLAG_TIMING | hip.FLAG_FAST_WEIGHTS)
            ms = []
            for _ in range(3):
                colour.copy_(colour0)
                ctx.filter_device(d2, planes.data_ptr(), colour.data_ptr(), stream)
                ms.append(ctx.counters().filter_kernel_ms)
            torch.cuda.synchronize()
            rel = float(((colour - ref).norm() / ref.norm()).item())
            out["fast_weights_f32"] = {"kernel_ms": min(ms), "Msamples_per_s_kernel": n_own * W * S / (min(ms) * 1e-3) / 1e6,
                                       "rel_l2_vs_f64_path": rel}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
