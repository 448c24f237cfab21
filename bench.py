#!/usr/bin/env python3
"""bench.py -- RPF filter-pass throughput on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one filter pass (box 7: FillMeanAndStddev + the fused per-pixel kernel, rpf.cpp:497-733) over one
synthetic feature buffer that is already resident in HBM.

--workload cfg2 (default; BASELINE.json configs[1], the configuration the metric is quoted on): 1920x1080x8spp, the
    reference's 19-dim fp32 planes.  N > 1: weak scaling -- the image is 1920 x (1080*N) rows, row-tiled one 1080-row
    slab per rank; every step re-exchanges the colour halo rows with the neighbouring ranks (RCCL send/recv) and then
    filters the slab.  Beside `value`, the line carries `scaling_4k32`: STRONG scaling of BASELINE configs[3]
    (3840x2160x32spp, 2160/N rows per rank, the north_star's 8-GPU shape), two timed steps.
--workload cfg5: BASELINE configs[4], 8192 px wide x 64 spp, 27-dim sample vectors in fp16 feature storage, one row
    slab per rank generated on the device in row chunks (the full 8192^2 frame is 232 GB of features); 66 algorithmic
    bytes per sample (27 x fp16 read + 3 x fp32 written).
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

WORKLOADS = {
    # name: width, rows per GPU, spp, layout kwargs, algorithmic bytes per sample per pass
    "cfg2": dict(width=1920, rows=1080, spp=8, layout={}, algo_bytes=88.0, label="19-dim fp32 planes"),
    "cfg5": dict(width=8192, rows=512, spp=64, layout=dict(n_random=4, n_feat=18), algo_bytes=66.0,
                 label="27-dim fp16 planes"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--rows-per-gpu", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--box", type=int, default=7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scaling-4k32", action="store_true", help="skip the strong-scaling 3840x2160x32 object")
    ap.add_argument("--fast-weights", action="store_true", help="opt-in fp32 pair weights (RPF_FLAG_FAST_WEIGHTS)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="rpf_set_option override (diagnostics: stage_mask, binning, waves_per_pixel, table_in_lds, lds_pad, screen)")
    ap.add_argument("--allow-nonfinite", action="store_true", help="profiling variants whose results are wrong on purpose")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) is the measured path; gloo lets several ranks share one GPU to "
                         "rehearse the multi-rank code path (halo rows staged through the host)")
    return ap.parse_args()


def load_traffic(workload_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/traffic.json):
    measured once per kernel revision on the GPU box, NOT in this run (refresh it whenever the kernel changes)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload_key)
    except (OSError, ValueError):
        return None


def cpu_baseline(torch, planes_dev, slab_rows, W, S, box, target_s, gpu_colour_dev, layout):
    """Oracle (CPU port) timed on a bounded sample of the same workload: R full-width rows (+ halo rows) cut
    out of the very buffer the GPU filtered; also reports the GPU/oracle rel-L2 on those rows."""
    import numpy as np
    import pyoracle as O
    O.build()
    b = (box - 1) // 2
    cores = min(16, len(os.sched_getaffinity(0)))  # a 1-GPU box's CPU share
    r0 = min(400, max(b, slab_rows // 3))
    # wide 64-spp rows cost the oracle minutes each: a column window bounds the sample instead of fewer than one row
    xw = W if W * S <= 1920 * 16 else max(64, (1920 * 16 // S) // 64 * 64)
    x0 = 0 if xw == W else (W - xw) // 2

    def run(R):
        lo, hi = r0 - b, r0 + R + b
        xa, xb = (0, W) if xw == W else (x0 - b, x0 + xw + b)
        sub = planes_dev[:, lo:hi, xa:xb].float().contiguous().cpu().numpy()
        d = O.make_desc(xb - xa, hi - lo, S, box=box, row_begin=b, row_end=b + R, n_threads=cores, **layout)
        t = time.perf_counter()
        r = O.filter_pass(sub, d, debug=False)
        dt = time.perf_counter() - t
        return r, dt, (xb - xa)

    R = 8 if xw == W else 1
    r, dt, wsub = run(R)
    rate = R * wsub * S / dt
    R2 = int(max(R, min(slab_rows - r0 - b - 1, 512, target_s * rate / (wsub * S))))
    if R2 > R:
        R = R2
        r, dt, wsub = run(R)
    xo = 0 if xw == W else b
    got = gpu_colour_dev[:, r0:r0 + R, x0:x0 + xw].cpu().numpy()
    want = r["colour"][:, b:b + R, xo:xo + xw]
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    n_px = R * wsub   # pixels the oracle filtered (the window's side columns included: they are work done)
    return {"value": n_px * S / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d rows x %d px (%d samples, mean N %.0f) of the same buffer, oracle/rpf_oracle.c "
                      "fp64 OpenMP, %.1f s" % (R, wsub, n_px * S, r["sum_nbhd"] / n_px, dt),
            "gpu_vs_oracle_rel_l2": rel}


class Job:
    """one row-tiled workload on this rank: slab buffers in HBM, the halo plan, a step function"""

    def __init__(self, torch, dist, hip, fb, slabs, args, dev, rank, world, W, rows_total, S, box, layout, ctx):
        self.torch, self.dist, self.world, self.rank = torch, dist, world, rank
        self.W, self.S, self.box, self.ctx = W, S, box, ctx
        halo = fb.halo_rows(box)
        self.slab = slabs.slab_for(rows_total, world, rank, halo)
        self.H_buf, rb, re = slabs.buffer_rows(self.slab)
        self.n_own = self.slab.row1 - self.slab.row0
        f16 = bool(layout)
        xp = fb.torch_backend(dev)
        gen = dict(row0=self.slab.row0 - self.slab.halo_top, xp=xp, mode="smooth", sigma_f=0.05, sigma_c=1e-4,
                   dtype="f16" if f16 else "f32", **layout)
        # synthetic feature buffer generated directly in HBM, in row chunks (the generator's fp64 temporaries are ~60x
        # its output); halo rows come from the generator too (setup, untimed) -- what the neighbour rank generates
        chunk = max(1, min(self.H_buf, (1 << 25) // (W * S)))
        self.planes = fb.synth_planes_chunked(W, self.H_buf, S, rows_per_chunk=chunk, **gen).contiguous()
        self.colour0 = self.planes[2:5].to(torch.float64).contiguous()
        self.colour = self.colour0.clone()
        flags = hip.FLAG_TIMING | (hip.FLAG_FAST_WEIGHTS if args.fast_weights else 0)
        self.desc = hip.make_desc(W, self.H_buf, S, boxes=(box,), row_begin=rb, row_end=re, flags=flags,
                                  plane_dtype=hip.PLANES_F16 if f16 else hip.PLANES_F32, **layout)
        self.plan = slabs.HaloPlan(self.colour, self.slab, rank, world)
        self.stream = torch.cuda.current_stream().cuda_stream
        self.kernel_ms = []
        self.allow_nonfinite = args.allow_nonfinite

    def step(self):
        self.colour.copy_(self.colour0)                         # the pass input (unfiltered colours)
        self.plan.exchange(self.colour)                         # RCCL neighbour exchange of the colour halo
        self.ctx.filter_device(self.desc, self.planes.data_ptr(), self.colour.data_ptr(), self.stream,
                               allow_nonfinite=self.allow_nonfinite)
        self.kernel_ms.append(self.ctx.counters().filter_kernel_ms)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, steps, warmup, dev, backend):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks"""
        for _ in range(warmup):
            self.step()
        self.kernel_ms.clear()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            t = self.torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=self.torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed


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

    wl = WORKLOADS[args.workload]
    W = args.width or wl["width"]
    rows_per_gpu = args.rows_per_gpu or wl["rows"]
    S = args.spp or wl["spp"]
    box, layout = args.box, wl["layout"]
    ctx = hip.Context(local_rank)
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))

    job = Job(torch, dist, hip, fb, slabs, args, dev, rank, world, W, rows_per_gpu * world, S, box, layout, ctx)
    elapsed = job.timed(args.steps, args.warmup, dev, args.dist_backend)
    cnt = ctx.counters()
    n_own = job.n_own
    H_total = rows_per_gpu * world
    value = H_total * W * S * args.steps / elapsed / 1e6
    k_ms = sum(job.kernel_ms) / max(len(job.kernel_ms), 1)
    algo_bytes = wl["algo_bytes"] * n_own * W * S
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    workload = "synthetic %dx%dx%dspp (smooth, sigma_f=0.05), box %d, 1 pass, %s" % (W, rows_per_gpu, S, box, wl["label"])
    out = {
        "metric": "RPF Msamples/sec filtered at 1080p×8spp",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64+f32 pair weights" if args.fast_weights else "f64", "data": "synthetic",
        "config": {"workload": workload, "image": "%dx%d" % (W, H_total), "rows_per_gpu": rows_per_gpu,
                   "mean_nbhd": cnt.sum_nbhd / float(n_own * W), "max_nbhd": cnt.max_nbhd,
                   "beta_map": "REF_GCC11_O3", "degenerate_policy": "REF_ABORT",
                   "nonfinite_pixels": cnt.nonfinite_pixels,
                   "parallelism": "row slabs x%d, 3-row colour halo over %s send/recv" % (
                       world, "RCCL" if args.dist_backend == "nccl" else "gloo (rehearsal: ranks share GPUs)")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": load_traffic("%dx%dx%d_box%d" % (W, rows_per_gpu, S, box)) if not layout else None,
                     "traffic_source": "profiles/traffic.json (rocprofv3 --pmc passes of the same command, committed per "
                                       "kernel revision; not re-measured by this run)",
                     "kernel": "filter_pixel_kernel", "kernel_ms": k_ms, "kernel_launches_per_step": cnt.filter_kernel_launches,
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "note": "the kernel is issue / LDS / texture-path bound (~350 ops/B, SURVEY 8d), not HBM bound; with "
                             "box*box*S > 512 a step is nbhd_count + classify + one filter launch per occupied size class, "
                             "and kernel_ms is their sum"},
    }
    # the histogram work that dominates the fused kernel: one LDS atomic increment per neighbourhood sample for each
    # of the 96 joint histograms of a pixel (the reference layout; marginals are read off the joints).  Peak = measured
    # ds_add_rtn_u32 rate on random cells (profiles/r01_lds_atomic_microbench.txt: 9.8 LDS cycles per 64 increments per CU
    # at 8 waves/CU) x 256 CUs x 2.4 GHz -- the builder's own microbenchmark, not a guide number.
    if not layout:
        incr = 96.0 * cnt.sum_nbhd
        lds_peak = 256 * (64.0 / 9.8) * 2.4e9
        out["lds_atomic_roofline"] = {"bound": "lds_atomics", "achieved": incr / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0,
                                      "peak": lds_peak / 1e12, "unit": "T increments/s",
                                      "frac": (incr / (k_ms * 1e-3)) / lds_peak if k_ms > 0 else 0.0,
                                      "note": "whole-kernel time in the denominator; the MI stage alone is ~35% of it (scripts/ablate_mi.sh)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(torch, job.planes, n_own, W, S, box, args.cpu_seconds, job.colour, layout)
        if not args.fast_weights and not layout:
            # the opt-in fp32 pair-weight mode, measured on the same buffer for information (never `value`)
            ref = job.colour.clone()
            d2 = hip.make_desc(W, job.H_buf, S, boxes=(box,), row_begin=job.desc.row_begin, row_end=job.desc.row_end,
                               flags=hip.FLAG_TIMING | hip.FLAG_FAST_WEIGHTS)
            ms = []
            for _ in range(3):
                job.colour.copy_(job.colour0)
                ctx.filter_device(d2, job.planes.data_ptr(), job.colour.data_ptr(), job.stream)
                ms.append(ctx.counters().filter_kernel_ms)
            torch.cuda.synchronize()
            rel = float(((job.colour - ref).norm() / ref.norm()).item())
            out["fast_weights_f32"] = {"kernel_ms": min(ms), "Msamples_per_s_kernel": n_own * W * S / (min(ms) * 1e-3) / 1e6,
                                       "rel_l2_vs_f64_path": rel}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not layout:
        # the headline buffer is close to the identity regime (most cross weights underflow), so its GPU-vs-oracle
        # figure says little about stage 4: the same comparison on a small filter-ACTIVE buffer (clustered generator)
        import numpy as np
        import pyoracle as O
        Wp, Hp, Sp = 256, 48, 8
        pl = fb.synth_planes(Wp, Hp, Sp, seed=11, sigma_f=1e-3, sigma_c=0.01, mode="clustered")
        got = ctx.filter_pass_debug(pl, hip.make_desc(Wp, Hp, Sp), box=box, debug=False)["colour"]
        want = O.filter_pass(pl, O.make_desc(Wp, Hp, Sp, box=box), debug=False)["colour"]
        cin = pl[2:5].astype(np.float64)
        out["parity_probe"] = {"buffer": "clustered %dx%dx%d, sigma_f=1e-3 (filter-active)" % (Wp, Hp, Sp),
                               "activity_rel_l2": float(np.linalg.norm(want - cin) / np.linalg.norm(cin)),
                               "gpu_vs_oracle_rel_l2": float(np.linalg.norm(got - want) / np.linalg.norm(want))}
        # ... and on a 32-spp buffer, whose neighbourhoods (~1500 samples) run the four-wave kernels and their far-pair screen
        Wq, Hq, Sq = 48, 20, 32
        pl = fb.synth_planes(Wq, Hq, Sq, seed=12, sigma_f=0.02, sigma_c=0.01, mode="clustered")
        r = ctx.filter_pass_debug(pl, hip.make_desc(Wq, Hq, Sq), box=box, debug=False)
        want = O.filter_pass(pl, O.make_desc(Wq, Hq, Sq, box=box), debug=False)["colour"]
        cin = pl[2:5].astype(np.float64)
        out["parity_probe_32spp"] = {"buffer": "clustered %dx%dx%d, sigma_f=0.02" % (Wq, Hq, Sq), "max_nbhd": int(r["max_nbhd"]),
                                     "activity_rel_l2": float(np.linalg.norm(want - cin) / np.linalg.norm(cin)),
                                     "gpu_vs_oracle_rel_l2": float(np.linalg.norm(r["colour"] - want) / np.linalg.norm(want))}

    # ---- strong scaling of BASELINE configs[3]: 3840x2160x32 spp row-tiled over the N ranks ---------------------------
    if args.workload == "cfg2" and not args.no_scaling_4k32 and not args.option:
        del job
        torch.cuda.empty_cache()
        W4, H4, S4 = 3840, 2160, 32
        j4 = Job(torch, dist, hip, fb, slabs, args, dev, rank, world, W4, H4, S4, box, {}, ctx)
        steps4 = 2
        e4 = j4.timed(steps4, 1, dev, args.dist_backend)
        k4 = sum(j4.kernel_ms) / max(len(j4.kernel_ms), 1)
        c4 = ctx.counters()
        out["scaling_4k32"] = {
            "workload": "synthetic 3840x2160x32spp (smooth, sigma_f=0.05), box 7, 1 pass, 19-dim fp32 planes, %d rows per rank"
                        % (j4.n_own), "scaling": "strong", "n_gpus": world, "steps": steps4, "warmup": 1,
            "value": W4 * H4 * S4 * steps4 / e4 / 1e6, "unit": "Msamples/s", "ms_per_step": e4 / steps4 * 1e3,
            "kernel_ms_rank0": k4, "mean_nbhd_rank0": c4.sum_nbhd / float(j4.n_own * W4),
            "roofline_frac_rank0": (88.0 * j4.n_own * W4 * S4 / (k4 * 1e-3) / 1e9 / HBM_PEAK_GBS) if k4 > 0 else 0.0}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
