"""ctypes binding of libzero_hip.so (the C ABI declared in include/zero_hip.h).

There is NO fallback: if the library is missing the import fails loudly, so a GPU run can
never silently route around the HIP kernels.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZERO_AMD_LIB: an alternative build of the same ABI (A/B runs of two kernel versions on one device, -DZK_ABLATION
# timing builds); never a fallback — the file must exist
LIB_PATH = os.environ.get("ZERO_AMD_LIB") or os.path.join(_HERE, "libzero_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C zero_amd/csrc` (or __graft_entry__.build()); "
        "zero_amd has no non-HIP fallback path")

lib = ctypes.CDLL(LIB_PATH)

_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float

# name -> argtypes, in the order of include/zero_hip.h
SIGNATURES = {
    "zk_abi_version": [],
    "zk_gemm_bf16": [_P, _P, _P, _L, _L, _L, _L, _L, _L, _I, _I, _I, _P, _P, _L, _P, _P, _L, _F, _P, _I, _P, _L, _P],
    "zk_gemv_bf16": [_P, _P, _P, _L, _L, _L, _L, _L, _L, _P, _L, _P],
    "zk_gemv_fused_bf16": [_P, _P, _P, _L, _L, _L, _L, _L, _L, _P, _L, _I, _P, _F, _P],
    "zk_gemm_bf16_plan": [_P, _P, _P, _L, _L, _L, _L, _L, _L, _I, _I, _I, _P, _P, _L, _P, _P, _L, _F, _P, _I, _P, _L, _P, _P],
    "zk_gemm_last_plan": [],
    "zk_gemm_plan_query": [_L, _L, _L, _L, _L, _L, _I, _I, _I, _I, _I, _L, _I, _I, _L, _I, _L, _P],
    "zk_attn_fwd": [_P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _L, _L, _F, _I, _P],
    "zk_attn_fwd_w64": [_P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _L, _L, _F, _I, _P],
    "zk_attn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _F, _I, _P],
    "zk_attn_bwd_rope": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _F, _I,
                         _P, _P, _P],
    "zk_attn_decode": [_P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _F, _P],
    "zk_attn_decode_fused": [_P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _L, _L, _F, _P],
    "zk_attn_decode_fused_static": [_P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _L, _F, _P],
    "zk_llama_decode_attn": [_P, _P, _P, _I, _P],
    "zk_llama_decode_mlp": [_P, _P, _P, _I, _P],
    "zk_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _L, _L, _F, _P],
    "zk_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _P],
    "zk_rmsnorm_fwd": [_P, _P, _P, _P, _L, _L, _F, _P],
    "zk_rmsnorm_bwd": [_P, _P, _P, _P, _P, _P, _L, _L, _P],
    "zk_rmsnorm_dweight": [_P, _P, _P, _P, _L, _L, _L, _P],
    "zk_rope": [_P, _P, _P, _L, _L, _L, _L, _L, _I, _P],
    "zk_swiglu_fwd": [_P, _P, _L, _L, _P],
    "zk_swiglu_bwd": [_P, _P, _P, _L, _L, _P],
    "zk_colsum": [_P, _P, _L, _L, _L, _L, _P],
    "zk_act_bwd": [_P, _P, _P, _L, _I, _P],
    "zk_patch_im2col": [_P, _P, _L, _L, _L, _L, _L, _P],
    "zk_embed_finish": [_P, _P, _P, _P, _L, _L, _L, _P],
    "zk_transpose_bf16": [_P, _P, _L, _L, _L, _L, _P],
    "zk_gather_rows": [_P, _P, _P, _P, _L, _L, _I, _P],
    "zk_cross_entropy": [_P, _P, _P, _P, _P, _L, _L, _L, _P],
    "zk_sumsq": [_P, _P, _P, _L, _F, _P],
    "zk_adamw_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _L, _F, _P, _F, _P],
    "zk_f32_to_bf16": [_P, _P, _L, _F, _I, _P],
    "zk_f32_to_bf16_multi": [_P, _P, _P, _P, _P, _P, _I, _F, _P],
    "zk_u8_resample_rows": [_P, _L, _L, _L, _L, _L, _L, c_uint32, _P, _P, _I, _L, _P, _P],
    "zk_u8_resample_cols": [_P, _L, _L, _P, _P, _I, _L, _P, _P],
    "zk_u8_patches_normalize": [_P, _L, _L, _L, _L, _L, c_uint32, _L, _L, _L, _P, _P, _I, _P],
}

for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = header / library mismatch
    _fn.argtypes = _args
    _fn.restype = c_int

ABI_VERSION = 6
if lib.zk_abi_version() != ABI_VERSION:
    raise ImportError("libzero_hip.so ABI version mismatch; rebuild with `make -C zero_amd/csrc`")


class DecodeLayer(ctypes.Structure):
    """zk_decode_layer_t of include/zero_hip.h."""
    _fields_ = ([(n, c_int64) for n in ("B", "d", "nh", "nkv", "hd", "F", "cache_rows", "pos")] + [("eps", c_float)] +
                [(n, c_void_p) for n in ("ln1_w", "wqkv", "wo", "ln2_w", "wgu", "wdown", "cos_t", "sin_t", "k_cache",
                                         "v_cache", "lens", "h", "qkv", "ao", "gu", "hm", "attn_ws")] +
                [("attn_ws_floats", c_int64)])


class GemmPlan(ctypes.Structure):
    """zk_gemm_plan_t of include/zero_hip.h (per-call kernel plan; the library holds no tuning state)."""
    _fields_ = [(n, c_int) for n in ("tile", "splits", "stage_mode", "ring", "lean", "pring", "band", "phase_delay",
                                     "ablation")]

    @classmethod
    def default(cls):
        return cls(-1, 0, -1, -1, -1, -1, 0, -1, 0)


class ZeroHipError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        raise ZeroHipError(f"{what} failed with code {rc} "
                           f"({ {-22: 'invalid argument / unsupported shape', -7: 'beyond the stated size limit'}.get(rc, 'launch failure')})")
