"""The one numerical constant of the MU update.

``eps`` guards every division and logarithm of the update rules (reference: torchnmf/constants.py:3).  The HIP
kernels hard-code the same value as ``nmfmu::kEps`` (csrc/nmfmu_fused.h); tests/test_host_logic.py checks that the
two agree bit for bit.
"""
import struct

#: single-precision machine epsilon, 2**-23 = 1.1920928955078125e-07
eps: float = 2.0 ** -23

assert struct.unpack('<I', struct.pack('<f', eps))[0] == 0x34000000
