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
--workload cfg1: BASELINE configs[0]'s shape, 400x400x8 spp, one pass, EPS policy, on the stand-in for a captured pbrt buffer
    (pbrt cannot be built here): the seeded generator with 94 % flat-quad pixels (a zero-variance normal: the strict 3-sigma
    test rejects every neighbour, N = S) and a 1e-5 in-pixel jitter elsewhere -- mean N ~ 9, p99 ~ 26, as SURVEY F10 measured on
    a captured killeroo buffer; cpu_baseline = the oracle on the FULL frame.
--workload cfg3: BASELINE configs[2]'s shape, 1920x1080x16 spp, four passes {7,7,5,5} per step, EPS policy, same
    small-neighbourhood generator; `value` counts every pass (W*H*S*4 samples per step).
Run without torch.distributed.run and --gpus N > 1, the script starts the N ranks itself (a child torch.distributed.run,
before this process touches a GPU) and passes rank 0's line through.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

SMOOTH = dict(mode="smooth", sigma_f=0.05, sigma_c=1e-4)   # large neighbourhoods (mean N ~ 0.75 box^2 S): the throughput workload
# the stand-in for a captured pbrt buffer (93.9 % of the pixels have a zero-variance feature => N = S; mean N 9.8,
# p99 26): 94 % flat-quad pixels, the others with an in-pixel jitter of 1e-5 (N = S ... 4S)
SMALLN = dict(mode="smooth", sigma_f=1e-5, sigma_c=1e-4, flat_frac=0.94)
WORKLOADS = {
    # name: width, rows per GPU, spp, layout kwargs, algorithmic bytes per sample per pass,
    #       generator, box list of one step, degenerate policy
    "cfg2": dict(width=1920, rows=1080, spp=8, layout={}, algo_bytes=88.0, label="19-dim fp32 planes", gen=SMOOTH,
                 boxes=None, policy="REF_ABORT"),
    "cfg5": dict(width=8192, rows=512, spp=64, layout=dict(n_random=4, n_feat=18), algo_bytes=66.0,
                 label="27-dim fp16 planes", gen=SMOOTH, boxes=None, policy="REF_ABORT"),
    "cfg1": dict(width=400, rows=400, spp=8, layout={}, algo_bytes=88.0, label="19-dim fp32 planes", gen=SMALLN,
                 boxes=(7,), policy="EPS"),
    "cfg3": dict(width=1920, rows=1080, spp=16, layout={}, algo_bytes=88.0, label="19-dim fp32 planes", gen=SMALLN,
                 boxes=(7, 7, 5, 5), policy="EPS"),
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
    ap.add_argument("--no-multi-inprocess", action="store_true", help="skip the rpf_multi_filter (one process, all visible GPUs) object")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="ranks meet over gloo, all-reduce one number and rank 0 prints {n_gpus, rendezvous}: exercises the "
                         "self-spawn / launcher path without a GPU (tests/test_slabs_gloo.py)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `torch.distributed.run` (this process
    has not touched a GPU and never will), pass the child's stdout -- rank 0's JSON line -- through and exit with its code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # only rank 0's JSON line goes to stdout (gloo's C++ side prints its "[Gloo] Rank ..." banner there too)
    pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in pr.stdout:
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return pr.wait()


def rendezvous_only(args):
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"n_gpus": world, "rendezvous": "ok", "sum_of_ranks_plus_one": float(t.item())}))


def load_traffic(workload_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/traffic.json):
    measured once per kernel revision on the GPU box, NOT in this run (refresh it whenever the kernel changes)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload_key)
    except (OSError, ValueError):
        return None


def cpu_baseline(torch, planes_dev, slab_rows, W, S, box, target_s, gpu_colour_dev, layout, policy="REF_ABORT",
                 full_frame=False):
    """Oracle (CPU port) timed on a bounded sample of the same workload: R full-width rows (+ halo rows) cut
    out of the very buffer the GPU filtered (full_frame: the whole buffer); also reports the GPU/oracle rel-L2 on those
    rows (gpu_colour_dev = the colours after ONE pass with `box`)."""
    import numpy as np
    import pyoracle as O
    O.build()
    b = (box - 1) // 2
    cores = min(16, len(os.sched_getaffinity(0)))  # a 1-GPU box's CPU share
    pol = O.DEGEN_EPS if policy == "EPS" else O.DEGEN_REF_ABORT
    if full_frame:
        sub = planes_dev.float().contiguous().cpu().numpy()
        d = O.make_desc(W, slab_rows, S, box=box, n_threads=cores, policy=pol, **layout)
        t = time.perf_counter()
        r = O.filter_pass(sub, d, debug=False)
        dt = time.perf_counter() - t
        want, got = r["colour"], gpu_colour_dev.cpu().numpy()
        return {"value": slab_rows * W * S / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
                "sample": "the full %dx%d frame (%d samples, mean N %.1f), oracle/rpf_oracle.c fp64 OpenMP, %.2f s"
                          % (W, slab_rows, slab_rows * W * S, r["sum_nbhd"] / (slab_rows * W), dt),
                "gpu_vs_oracle_rel_l2": float(np.linalg.norm(got - want) / np.linalg.norm(want))}
    r0 = min(400, max(b, slab_rows // 3))
    # wide 64-spp rows cost the oracle minutes each: a column window bounds the sample instead of fewer than one row
    xw = W if W * S <= 1920 * 16 else max(64, (1920 * 16 // S) // 64 * 64)
    x0 = 0 if xw == W else (W - xw) // 2

    def run(R):
        lo, hi = r0 - b, r0 + R + b
        xa, xb = (0, W) if xw == W else (x0 - b, x0 + xw + b)
        sub = planes_dev[:, lo:hi, xa:xb].float().contiguous().cpu().numpy()
        d = O.make_desc(xb - xa, hi - lo, S, box=box, row_begin=b, row_end=b + R, n_threads=cores, policy=pol, **layout)
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


def nbhd_statistics(ctx, W, H_buf, rb, re, S):
    """mean / percentiles of the neighbourhood size over the filtered rows (last pass of the last step)"""
    import numpy as np
    n = ctx.nbhd(W, H_buf)[rb:re].ravel()
    q = np.percentile(n, [50, 90, 99])
    return {"mean": float(n.mean()), "p50": float(q[0]), "p90": float(q[1]), "p99": float(q[2]), "max": int(n.max()),
            "frac_N_eq_S": float((n == S).mean())}


def multi_inprocess(torch, hip, fb, dev, box):
    """rpf_multi_filter -- the reference's own call shape (one process, RPFIntegrator::Render, rpf.cpp:737-805) -- on the
    3840x2160x32 frame over every visible device (one visible device: two row slabs on it), two passes {7,7} so that the
    colour-halo refresh between the passes (hipMemcpyPeerAsync between neighbouring devices) is part of what is timed."""
    import numpy as np
    W4, H4, S4 = 3840, 2160, 32
    ndev = torch.cuda.device_count()
    devices = list(range(ndev)) if ndev > 1 else [0, 0]
    planes = np.empty((19, H4, W4, S4), np.float32)                    # 20 GB of host memory: the film a renderer would hold
    chunk = max(1, (1 << 25) // (W4 * S4))
    xp = fb.torch_backend(dev)
    for r in range(0, H4, chunk):
        n = min(chunk, H4 - r)
        planes[:, r:r + n] = fb.synth_planes(W4, n, S4, row0=r, xp=xp, **SMOOTH).cpu().numpy()
    torch.cuda.empty_cache()
    boxes = (box, box)
    with hip.MultiContext(devices) as mc:
        t0 = time.perf_counter()
        _, prgb, st = mc.filter(planes, hip.make_desc(W4, H4, S4, boxes=boxes))
        wall = time.perf_counter() - t0
        c = mc.counters()
    return {"workload": "synthetic 3840x2160x32spp (smooth), boxes %s, rpf_multi_filter from host buffers" % (boxes,),
            "devices": devices, "wall_s": wall, "filter_kernel_ms": c.filter_kernel_ms,
            "Msamples_per_s_kernels": W4 * H4 * S4 * len(boxes) / (c.filter_kernel_ms * 1e-3) / 1e6,
            "Msamples_per_s_wall_incl_pcie": W4 * H4 * S4 * len(boxes) / wall / 1e6, "status": int(st),
            "mean_nbhd": c.sum_nbhd / float(W4 * H4), "nonfinite_pixels": int(c.nonfinite_pixels),
            "pixel_rgb_checksum": float(np.asarray(prgb, np.float64).sum()),
            "note": "wall clock includes the upload of 20 GB of planes from pageable memory and the download of the filtered "
                    "samples (never `value`); filter_kernel_ms = per pass the slowest slab, summed over the passes"}


class Job:
    """one row-tiled workload on this rank: slab buffers in HBM, the halo plan, a step function"""

    def __init__(self, torch, dist, hip, fb, slabs, args, dev, rank, world, W, rows_total, S, box, layout, ctx,
                 gen_kw=None, boxes=None, policy="REF_ABORT"):
        self.torch, self.dist, self.world, self.rank = torch, dist, world, rank
        self.W, self.S, self.box, self.ctx = W, S, box, ctx
        self.boxes = tuple(boxes) if boxes else (box,)
        if len(self.boxes) > 1 and world > 1:
            sys.exit("a multi-pass step needs the whole image on one rank (rpf_filter_device: n_box > 1 on a sub-slab is refused); "
                     "use rpf_multi_filter for multi-pass multi-GPU")
        halo = fb.halo_rows(max(self.boxes))
        self.slab = slabs.slab_for(rows_total, world, rank, halo)
        self.H_buf, rb, re = slabs.buffer_rows(self.slab)
        self.n_own = self.slab.row1 - self.slab.row0
        f16 = bool(layout)
        xp = fb.torch_backend(dev)
        gen = dict(row0=self.slab.row0 - self.slab.halo_top, xp=xp, dtype="f16" if f16 else "f32", **layout)
        gen.update(gen_kw or SMOOTH)
        # synthetic feature buffer generated directly in HBM, in row chunks (the generator's fp64 temporaries are ~60x
        # its output); halo rows come from the generator too (setup, untimed) -- what the neighbour rank generates
        chunk = max(1, min(self.H_buf, (1 << 25) // (W * S)))
        self.planes = fb.synth_planes_chunked(W, self.H_buf, S, rows_per_chunk=chunk, **gen).contiguous()
        self.colour0 = self.planes[2:5].to(torch.float64).contiguous()
        self.colour = self.colour0.clone()
        flags = hip.FLAG_TIMING | (hip.FLAG_FAST_WEIGHTS if args.fast_weights else 0)
        self.desc = hip.make_desc(W, self.H_buf, S, boxes=self.boxes, row_begin=rb, row_end=re, flags=flags,
                                  policy=hip.DEGEN_EPS if policy == "EPS" else hip.DEGEN_REF_ABORT,
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
    if os.environ.get("RPF_BENCH_WATCHDOG"):   # profiling runs: print every thread's Python stack if the run is still going after N s
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["RPF_BENCH_WATCHDOG"]), exit=False)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))      # before anything touches a GPU
    if args.rendezvous_only:
        return rendezvous_only(args)
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
        sys.exit("bench.py --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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
    boxes = wl["boxes"] or (box,)
    if wl["boxes"]:
        box = boxes[0]
    policy = wl["policy"]
    ctx = hip.Context(local_rank)
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v, 0))

    job = Job(torch, dist, hip, fb, slabs, args, dev, rank, world, W, rows_per_gpu * world, S, box, layout, ctx,
              gen_kw=wl["gen"], boxes=boxes, policy=policy)
    elapsed = job.timed(args.steps, args.warmup, dev, args.dist_backend)
    cnt = ctx.counters()
    n_own = job.n_own
    n_pass = len(boxes)
    H_total = rows_per_gpu * world
    value = H_total * W * S * n_pass * args.steps / elapsed / 1e6
    k_ms = sum(job.kernel_ms) / max(len(job.kernel_ms), 1)
    algo_bytes = wl["algo_bytes"] * n_own * W * S * n_pass
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    gen = wl["gen"]
    workload = "synthetic %dx%dx%dspp (%s, sigma_f=%g%s), %s, %s" % (
        W, rows_per_gpu, S, gen["mode"], gen["sigma_f"], ", flat_frac=%g" % gen["flat_frac"] if gen.get("flat_frac") else "",
        "box %d, 1 pass" % box if n_pass == 1 else "%d passes %s per step" % (n_pass, "{%s}" % ",".join(map(str, boxes))), wl["label"])
    out = {
        "metric": "RPF Msamples/sec filtered at 1080p×8spp",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64+f32 pair weights" if args.fast_weights else "f64", "data": "synthetic",
        "config": {"workload": workload, "image": "%dx%d" % (W, H_total), "rows_per_gpu": rows_per_gpu,
                   "passes_per_step": n_pass,
                   "mean_nbhd": cnt.sum_nbhd / float(n_own * W), "max_nbhd": cnt.max_nbhd,
                   "nbhd": nbhd_statistics(ctx, W, job.H_buf, job.desc.row_begin, job.desc.row_end, S),
                   "beta_map": "REF_GCC11_O3", "degenerate_policy": policy,
                   "nonfinite_pixels": cnt.nonfinite_pixels, "redo_pixels": cnt.redo_pixels,
                   "library": os.path.relpath(hip.LIB_PATH, ROOT) + (" (RPF_HIP_LIB override)" if os.environ.get("RPF_HIP_LIB") else ""),
                   "parallelism": "row slabs x%d, 3-row colour halo over %s send/recv" % (
                       world, "RCCL" if args.dist_backend == "nccl" else "gloo (rehearsal: ranks share GPUs)")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": load_traffic("%s_%dx%dx%d_box%d" % (args.workload, W, rows_per_gpu, S, box)),
                     "traffic_source": "profiles/traffic.json (rocprofv3 --pmc passes of the same command, committed per "
                                       "kernel revision; not re-measured by this run)",
                     "kernel": "filter_pixel_kernel", "kernel_ms": k_ms, "kernel_launches_per_step": cnt.filter_kernel_launches,
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "note": "the kernel is issue / LDS / texture-path bound (~350 ops/B, SURVEY 8d), not HBM bound; with "
                             "box*box*S > 512 a step is nbhd_count + classify + one filter launch per occupied size class "
                             "(per pass), and kernel_ms / algorithmic bytes are their sums over the step"},
    }
    # the histogram work that dominates the fused kernel: one LDS atomic increment per neighbourhood sample for each
    # of the 96 joint histograms of a pixel (the reference layout; marginals are read off the joints).  Peak = measured
    # ds_add_rtn_u32 rate on random cells (profiles/r01_lds_atomic_microbench.txt: 9.8 LDS cycles per 64 increments per CU
    # at 8 waves/CU) x 256 CUs x 2.4 GHz -- the builder's own microbenchmark, not a guide number: a diagnostic, not a roofline.
    if not layout and n_pass == 1:
        incr = 96.0 * cnt.sum_nbhd
        lds_peak = 256 * (64.0 / 9.8) * 2.4e9
        out["lds_atomic_diagnostic"] = {"achieved": incr / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0,
                                        "own_microbenchmark_rate": lds_peak / 1e12, "unit": "T increments/s",
                                        "ratio": (incr / (k_ms * 1e-3)) / lds_peak if k_ms > 0 else 0.0,
                                        "note": "whole-kernel time in the denominator; the MI stage alone is ~35% of it (scripts/ablate_mi.sh)"}
    single = rank == 0 and world == 1
    if single and not args.no_cpu_baseline:
        if n_pass > 1:   # the oracle sample is one pass: compare it with the colours after the first pass alone
            d1 = hip.make_desc(W, job.H_buf, S, boxes=(box,), row_begin=job.desc.row_begin, row_end=job.desc.row_end,
                               policy=job.desc.degenerate_policy)
            one = job.colour0.clone()
            ctx.filter_device(d1, job.planes.data_ptr(), one.data_ptr(), job.stream, allow_nonfinite=args.allow_nonfinite)
        else:
            one = job.colour
        out["cpu_baseline"] = cpu_baseline(torch, job.planes, n_own, W, S, box, args.cpu_seconds, one, layout, policy,
                                           full_frame=(args.workload == "cfg1"))
        if not args.fast_weights and args.workload == "cfg2":
            # the opt-in fp32 pair-weight mode, measured on the same buffer for information (never `value`)
            ref = job.colour.clone()
            d2 = hip.make_desc(W, job.H_buf, S, boxes=(box,), row_begin=job.desc.row_begin, row_end=job.desc.row_end,
                               policy=job.desc.degenerate_policy, flags=hip.FLAG_TIMING | hip.FLAG_FAST_WEIGHTS)
            ms = []
            for _ in range(3):
                job.colour.copy_(job.colour0)
                ctx.filter_device(d2, job.planes.data_ptr(), job.colour.data_ptr(), job.stream)
                ms.append(ctx.counters().filter_kernel_ms)
            torch.cuda.synchronize()
            rel = float(((job.colour - ref).norm() / ref.norm()).item())
            out["fast_weights_f32"] = {"kernel_ms": min(ms), "Msamples_per_s_kernel": n_own * W * S / (min(ms) * 1e-3) / 1e6,
                                       "rel_l2_vs_f64_path": rel}

    if single and not args.no_cpu_baseline and args.workload == "cfg2":
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
        # ... and on a 32-spp buffer whose neighbourhoods (~1000 samples) run the four-wave kernels, the split route and
        # their far-pair screen, with a jitter small enough that the filter moves the colours (activity >= 1e-3)
        Wq, Hq, Sq = 48, 20, 32
        pl = fb.synth_planes(Wq, Hq, Sq, seed=12, sigma_f=1e-3, sigma_c=0.01, mode="clustered")
        r = ctx.filter_pass_debug(pl, hip.make_desc(Wq, Hq, Sq), box=box, debug=False)
        want = O.filter_pass(pl, O.make_desc(Wq, Hq, Sq, box=box), debug=False)["colour"]
        cin = pl[2:5].astype(np.float64)
        out["parity_probe_32spp"] = {"buffer": "clustered %dx%dx%d, sigma_f=1e-3" % (Wq, Hq, Sq), "max_nbhd": int(r["max_nbhd"]),
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
        del j4
        torch.cuda.empty_cache()
        # ---- the same frame through the one-process multi-GPU entry point (every visible device) ----------------------
        if single and not args.no_multi_inprocess:
            try:
                out["multi_inprocess"] = multi_inprocess(torch, hip, fb, dev, box)
            except Exception as e:  # a diagnostic leg must not cost the headline line
                out["multi_inprocess"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
