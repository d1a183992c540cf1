"""eskf_lio_amd — MI355X-native Voxelized-GICP registration path of LimHaeryong/ESKF_LIO.

The product is the C-ABI HIP module (include/vgicp_hip.h, csrc/) and the C++ host mirror of the
reference's ICP / LocalMap interface (host/).  The Python modules are thin plumbing over the C ABI for
tests and bench.py: `capi` (ctypes binding), `synth` (seeded synthetic maps/scans), `distributed`
(point sharding + RCCL unique-id hand-off).  There is no CPU fallback anywhere in this package.
"""
__all__ = ["capi", "synth", "distributed"]
