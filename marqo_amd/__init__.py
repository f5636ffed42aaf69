"""marqo_amd — MI355X-native engine for Marqo's s2_inference.vectorise() hot path.

Python host code mirrors the reference's s2_inference interface (marqo_amd.s2_inference) and
reaches hand-written gfx950 HIP kernels through the C ABI in include/marqo_hip.h
(libmarqo_hip.so, bound with ctypes in marqo_amd._lib).
"""
__version__ = "0.1.0"
