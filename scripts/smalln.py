#!/usr/bin/env python3
"""Small-neighbourhood regime (what captured pbrt buffers look like, SURVEY F10: N = S for most pixels): the same
generator with a tiny in-pixel feature jitter so the 3-sigma test rejects nearly every neighbour.  EPS policy."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rpf_pkg
rpf_pkg.load()
from raytracer_rpf_amd import feature_buffer as fb, hip
W, H, S = 1920, int(os.environ.get('ROWS', '1080')), int(os.environ.get('SPP', '8'))
dev = torch.device("cuda", 0)
for sf in (1e-5, 3e-3):
    planes = fb.synth_planes(W, H, S, xp=fb.torch_backend(dev), mode="smooth", sigma_f=sf, sigma_c=1e-4).contiguous()
    col0 = planes[2:5].to(torch.float64).contiguous()
    ctx = hip.Context(0)
    desc = hip.make_desc(W, H, S, boxes=(7,), policy=hip.DEGEN_EPS, flags=hip.FLAG_TIMING)
    for _ in range(2):
        c = col0.clone()
        ctx.filter_device(desc, planes.data_ptr(), c.data_ptr(), torch.cuda.current_stream().cuda_stream)
    cnt = ctx.counters()
    print(json.dumps({"spp": S, "rows": H, "sigma_f": sf, "mean_nbhd": cnt.sum_nbhd / (W * H), "max_nbhd": cnt.max_nbhd, "kernel_ms": cnt.filter_kernel_ms,
                      "Msamples_per_s": W * H * S / (cnt.filter_kernel_ms * 1e-3) / 1e6, "nonfinite_pixels": cnt.nonfinite_pixels}))
