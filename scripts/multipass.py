#!/usr/bin/env python3
"""BASELINE configs[2]-style run on synthetic data: 1920x1080x16spp, 4 passes (box list 7,7,5,5), one GPU, EPS policy
(what real pbrt buffers need).  Prints per-pass-averaged throughput and checks determinism + finiteness."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rpf_pkg
rpf_pkg.load()
from raytracer_rpf_amd import feature_buffer as fb, hip

W, H, S = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 16
boxes = tuple(int(b) for b in os.environ.get('BOXES', '7,7,5,5').split(','))
dev = torch.device("cuda", 0)
planes = fb.synth_planes(W, H, S, xp=fb.torch_backend(dev), mode="clustered", sigma_f=1e-3, sigma_c=0.01).contiguous()
col0 = planes[2:5].to(torch.float64).contiguous()
ctx = hip.Context(0)
desc = hip.make_desc(W, H, S, boxes=boxes, policy=hip.DEGEN_EPS, flags=hip.FLAG_TIMING)
outs = []
for _ in range(2):
    c = col0.clone()
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.filter_device(desc, planes.data_ptr(), c.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    outs.append(c)
cnt = ctx.counters()
print(json.dumps({"workload": "%dx%dx%d boxes %s EPS clustered" % (W, H, S, boxes), "seconds": dt,
                  "Msamples_per_s": W * H * S * len(boxes) / dt / 1e6, "filter_kernel_ms_total": cnt.filter_kernel_ms,
                  "deterministic": bool(torch.equal(outs[0], outs[1])), "finite": bool(torch.isfinite(outs[0]).all()),
                  "nonfinite_pixels": cnt.nonfinite_pixels, "mean_nbhd_last_pass": cnt.sum_nbhd / (W * H),
                  "activity_rel_l2": float(((outs[0] - col0).norm() / col0.norm()).item())}))
