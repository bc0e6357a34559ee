"""cbet_raytracing_3d_amd -- MI355X-native ray integrator behind the C ABI of include/cbet_mi355x.h.

api     ctypes binding of libcbet_mi355x.so (no CPU fallback; raises if the library is missing)
tracer  torch-held device buffers + the multi-GPU pass (torch.distributed / RCCL)
build   hipcc build of the library for gfx950
"""
__all__ = ["api", "tracer", "build"]
