"""sos_slam_amd -- MI355X (gfx950) hot path of SOS-SLAM's photometric bundle adjustment and coarse tracker.

The compute path lives in csrc/ (hand-written HIP kernels behind the C-ABI of include/sos_slam.h plus
the C++ host facade); this package holds the ctypes binding, the synthetic window generator and the
multi-GPU sharding helpers.  Importing the package never touches the GPU; `sos_slam_amd.lib.load()`
fails loudly when the HIP library is missing.
"""
__all__ = ["synth"]
