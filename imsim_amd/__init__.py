"""imsim_amd -- MI355X-native stamp-rendering hot path behind imSim's plugin surface.

Only the hot path lives here: host-side mirrors of the reference interfaces
plus the HIP kernels in csrc/ behind the C-ABI of include/imsim_hip.h.  There is no CPU fallback.
"""
__version__ = "0.1.0"
