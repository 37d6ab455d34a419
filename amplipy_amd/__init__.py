"""amplipy_amd: MI355X-native trim + pileup + call engine behind AmpliPy's surface.

The package holds only what the hot path needs:
``csrc/`` (HIP kernels + the C-ABI ``libamplihip.so``), the ctypes binding, the
packed read batch, and the host-side mirror of AmpliPy's function seams.
"""
VERSION = "0.1.0"
AMPLIPY_VERSION = "0.0.2"  # the reference version whose behaviour is reproduced
