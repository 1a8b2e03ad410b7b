"""tredparse_amd -- MI355X-native hot path for tredparse (template SW + (h1,h2) likelihood grid).

Only what the path needs lives here: csrc/ (HIP kernels + C ABI -> libtredgpu.so), the ctypes
binding, and the host-side mirrors of the reference's operator interfaces for this path.
"""
__version__ = "0.5.0"
