"""zero_amd — MI355X-native encoder + adapter + surrogate-decoder training path.

Host side mirrors the reference's plugin interface (build_encoder / build_decoder /
models.adapter.Zero / methods.forward_hf / distributed.reduce_sum); arithmetic runs in
hand-written gfx950 HIP kernels behind the C ABI of include/zero_hip.h.
"""
__version__ = "0.1.0"
