#!/usr/bin/env python3
"""Run one pass twice on the same buffer and report pixels whose filtered colours differ bitwise (there must be none)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rpf_pkg
rpf_pkg.load()
from raytracer_rpf_amd import feature_buffer as fb, hip
W, H, S = int(os.environ.get("WIDTH", "3840")), int(os.environ.get("ROWS", "96")), int(os.environ.get("SPP", "32"))
dev = torch.device("cuda", 0)
planes = fb.synth_planes(W, H, S, xp=fb.torch_backend(dev), mode=os.environ.get("MODE", "smooth"),
                         sigma_f=float(os.environ.get("SF", "0.05")), sigma_c=1e-4).contiguous()
col0 = planes[2:5].to(torch.float64).contiguous()
ctx = hip.Context(0)
desc = hip.make_desc(W, H, S, policy=hip.DEGEN_EPS)
outs = []
for _ in range(3):
    c = col0.clone()
    ctx.filter_device(desc, planes.data_ptr(), c.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    outs.append(c)
for k in (1, 2):
    diff = (outs[0] != outs[k]).any(dim=0).any(dim=-1)   # [H][W]
    nd = int(diff.sum())
    print("run 0 vs %d: %d differing pixels of %d" % (k, nd, W * H), flush=True)
    if nd:
        ys, xs = torch.nonzero(diff, as_tuple=True)
        rel = ((outs[0] - outs[k]).abs().amax() / outs[0].abs().amax()).item()
        print("  max abs diff / max: %.3e ; rows %d..%d cols %d..%d" % (rel, ys.min(), ys.max(), xs.min(), xs.max()))
        try:  # neighbourhood sizes of the differing pixels (which size class / kernel family they ran in)
            import numpy as np
            N = np.asarray(ctx.nbhd(W, H)).reshape(H, W)
            nn = N[ys.cpu().numpy(), xs.cpu().numpy()]
            edges = [1, 9, 17, 33, 65, 129, 257, 449, 833, 1601, 3137, 1 << 30]  # class c: N <= 8, 16, ..., 3136, more
            hist_d, _ = np.histogram(nn, bins=edges)
            hist_a, _ = np.histogram(N.ravel(), bins=edges)
            print("  N classes (N <= %s, more): differing %s of all %s" % ([e - 1 for e in edges[1:-1]], hist_d.tolist(), hist_a.tolist()))
        except Exception as e:
            print("  (nbhd query failed: %s)" % e)
        for i in range(min(nd, 6)):
            y, x = int(ys[i]), int(xs[i])
            d = (outs[0][:, y, x] - outs[k][:, y, x])
            print("  pixel (x=%d,y=%d): samples differing %d/%d, max |d| %.3e" % (x, y, int((d != 0).any(dim=0).sum()), S, d.abs().max().item()))
