# stage ablation in the small-neighbourhood regime (sigma_f = 1e-5 => mean N ~ 17)
for m in -1 0 1 3 7; do MASK=$m python - <<PY
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, rpf_pkg
rpf_pkg.load()
from raytracer_rpf_amd import feature_buffer as fb, hip
W, H, S = 1920, 1080, 8
dev = torch.device("cuda", 0)
planes = fb.synth_planes(W, H, S, xp=fb.torch_backend(dev), mode="smooth", sigma_f=1e-5, sigma_c=1e-4).contiguous()
col0 = planes[2:5].to(torch.float64).contiguous()
ctx = hip.Context(0)
ctx.set_option("stage_mask", int(os.environ["MASK"]))
desc = hip.make_desc(W, H, S, boxes=(7,), policy=hip.DEGEN_EPS, flags=hip.FLAG_TIMING)
for _ in range(2):
    c = col0.clone()
    ctx.filter_device(desc, planes.data_ptr(), c.data_ptr(), torch.cuda.current_stream().cuda_stream)
print("mask", os.environ["MASK"], "kernel_ms %.1f" % ctx.counters().filter_kernel_ms)
PY
done
