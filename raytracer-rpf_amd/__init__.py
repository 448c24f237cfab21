"""raytracer-rpf_amd: MI355X-native Random Parameter Filtering pass (drop-in for the reference's
RPFIntegrator::ApplyRPFFilter, /root/reference/src/custom/rpf.cpp:497-733).

Import name: ``raytracer_rpf_amd`` (see rpf_pkg.load()).  Submodules:
  feature_buffer  SoA layout helpers + seeded synthetic generator
  hip             ctypes binding of the C-ABI library (include/rpf_hip.h); fails loudly if it is missing
  slabs           row-slab partitioning + halo exchange over torch.distributed (RCCL / gloo)
"""
