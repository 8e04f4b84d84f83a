"""ctypes binding of libcape_hip.so (C ABI declared in include/cape_hip.h).

Fails loudly: a missing / unloadable library raises ImportError at import time; a kernel entry
point that returns nonzero raises RuntimeError with `cape_last_error()`.
"""
import ctypes
import os

# torch must be imported (and with it the HIP runtime it bundles) BEFORE libcape_hip.so is loaded: the loader resolves
# libamdhip64.so.7 by soname, and a process must hold exactly one HIP runtime -- the one torch allocates memory and
# streams with.  Loading this library first would bind it to /opt/rocm's copy and every kernel launch would then see
# "no ROCm-capable device" for torch's pointers (observed when a CPU-only test imported this module first).
import torch  # noqa: F401  (import order is load order)

from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_longlong, c_size_t, c_uint32, c_uint64, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CAPE_HIP_LIB", os.path.join(_HERE, "..", "csrc", "libcape_hip.so"))
LIB_PATH = os.path.abspath(LIB_PATH)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"libcape_hip.so not found at {LIB_PATH}; build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C category-agnostic-pose-estimation_amd/csrc`).  There is no CPU fallback.")
try:
    _lib = ctypes.CDLL(LIB_PATH)
except OSError as e:  # pragma: no cover
    raise ImportError(f"cannot load {LIB_PATH}: {e}") from e

_lib.cape_last_error.restype = c_char_p
_lib.cape_abi_version.restype = c_int


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("M", c_int), ("N", c_int), ("K", c_int), ("a_mode", c_int), ("b_mode", c_int),
        ("A", c_void_p), ("lda", c_longlong), ("B", c_void_p), ("ldb", c_longlong), ("C", c_void_p), ("ldc", c_longlong),
        ("cN", c_int), ("cH", c_int), ("cW", c_int), ("cC", c_int), ("cKH", c_int), ("cKW", c_int),
        ("cStride", c_int), ("cPad", c_int), ("cOH", c_int), ("cOW", c_int), ("cO", c_int),
        ("scale", c_void_p), ("bias", c_void_p), ("residual", c_void_p), ("ldr", c_longlong),
        ("relu", c_int), ("accumulate", c_int), ("split_k", c_int), ("dropout_p", c_float),
        ("rng_state", c_void_p), ("rng_stream", c_uint32), ("colsum_out", c_void_p), ("precision", c_int),
        ("B_packed", c_void_p),
        ("mask_src", c_void_p), ("ldm", c_longlong), ("mask_scale", c_float),
        ("batch", c_int), ("batch_div", c_int), ("sA0", c_longlong), ("sA1", c_longlong), ("sB0", c_longlong), ("sB1", c_longlong),
        ("sC0", c_longlong), ("sC1", c_longlong),
        ("res_cols", c_int), ("sBias0", c_longlong), ("sBias1", c_longlong),
        ("cPadX", c_int), ("cKHp", c_int), ("cKWp", c_int), ("cTapH0", c_int), ("cTapHS", c_int), ("cTapW0", c_int), ("cTapWS", c_int),
    ]


class PackItem(ctypes.Structure):
    _fields_ = [("B", c_void_p), ("out", c_void_p), ("ldb", c_longlong), ("N", c_int), ("K", c_int), ("b_mode", c_int), ("pad", c_int)]


class DecodeLinearDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_int), ("K", c_int), ("Nout", c_int),
        ("X", c_void_p), ("ldx", c_longlong), ("in_gamma", c_void_p), ("in_beta", c_void_p), ("in_add", c_void_p), ("ld_add", c_longlong),
        ("W", c_void_p), ("ldw", c_longlong), ("bias", c_void_p),
        ("X2", c_void_p), ("ldx2", c_longlong), ("K2", c_int), ("W2", c_void_p), ("ldw2", c_longlong), ("n2", c_int),
        ("R", c_void_p), ("ldr", c_longlong), ("res_gamma", c_void_p), ("res_beta", c_void_p),
        ("relu", c_int), ("nseg", c_int), ("seg", c_int), ("out", c_void_p * 3), ("ldo", c_longlong * 3),
    ]


class DecodeTailDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_int), ("L", c_int),
        ("P4", c_void_p), ("ldp", c_longlong), ("g3", c_void_p), ("b3", c_void_p),
        ("W1", c_void_p), ("B1", c_void_p), ("W2", c_void_p), ("B2", c_void_p), ("W3", c_void_p), ("B3", c_void_p),
        ("ref", c_void_p),
        ("Wc", c_void_p), ("Bc", c_void_p), ("ncls", c_int),
        ("Wp", c_void_p), ("Bp", c_void_p), ("gp", c_void_p), ("bp", c_void_p),
        ("dim_t", c_void_p), ("vr", c_void_p),
        ("ref_out", c_void_p), ("ld_ref", c_longlong),
        ("qpos_out", c_void_p), ("refin_out", c_void_p),
        ("cls_out", c_void_p), ("ld_cls", c_longlong),
        ("hs_out", c_void_p), ("ld_hs", c_longlong),
    ]


class AugItem(ctypes.Structure):
    """field-for-field `cape_augment_item` (include/cape_hip.h)"""
    _fields_ = [("src", c_void_p), ("aug", c_void_p), ("out", c_void_p), ("stat", c_void_p), ("h", c_int), ("w", c_int),
                ("M", c_float * 6), ("color_on", c_int), ("order", c_int * 4),
                ("bright", c_float), ("contrast", c_float), ("sat", c_float), ("hue", c_float),
                ("mode", c_int), ("noise_std", c_float), ("seed", c_uint32), ("blur_k", c_int), ("blur_w", c_float * 49)]


DECODE_MAX_LAYERS = 8
GEMM_GROUP_MAX = 32          # CAPE_GEMM_GROUP_MAX
SUMSQ_PARTS = 256            # CAPE_SUMSQ_PARTS


class DecodeLayerDesc(ctypes.Structure):
    """field-for-field `cape_decode_layer_desc` (include/cape_hip.h)"""
    _fields_ = [(n, c_void_p) for n in (
        "w_qkv", "b_qkv", "w_qin", "k_cache", "v_cache", "w_o", "b_o", "ln2_g", "ln2_b",
        "w_sq", "b_sq", "sup_k", "sup_v", "sup_mask", "w_so", "b_so", "lns_g", "lns_b",
        "w_off", "b_off", "value", "w_mo", "b_mo", "ln1_g", "ln1_b",
        "w1", "b1", "w2", "b2", "ln3_g", "ln3_b", "m1w", "m1b", "m2w", "m2b", "m3w", "m3b")]


class DecodeStepDesc(ctypes.Structure):
    """field-for-field `cape_decode_step_desc`"""
    _fields_ = [
        ("N", c_int), ("n_layers", c_int), ("step", c_int), ("T", c_int), ("P", c_int), ("S", c_int), ("L", c_int),
        ("n_points", c_int), ("ncls", c_int), ("ffn_dim", c_int),
        ("shapes", c_int * 8), ("level_start", c_int * 4),
        ("emb", c_void_p), ("qpos0", c_void_p), ("refin0", c_void_p), ("ref0", c_void_p), ("vr", c_void_p), ("dim_t", c_void_p),
        ("class_w", c_void_p), ("class_b", c_void_p),
        ("pos_w", c_void_p), ("pos_b", c_void_p), ("pos_gamma", c_void_p), ("pos_beta", c_void_p),
        ("out_logits", c_void_p), ("ld_logits", c_longlong),
        ("out_coords", c_void_p), ("ld_coords", c_longlong),
        ("out_hs", c_void_p), ("ld_hs", c_longlong),
        ("layers", DecodeLayerDesc * DECODE_MAX_LAYERS),
    ]


P, I, LL, F, U32 = c_void_p, c_int, c_longlong, c_float, c_uint32
_SIGS = {
    "cape_rng_advance": [P, P],
    "cape_stream_fork": [P, P],
    "cape_stream_join": [P, P],
    "cape_gemm_f32": [POINTER(GemmDesc), P],
    "cape_gemm_group_f32": [POINTER(GemmDesc), I, I, P],
    "cape_pack_weights": [P, I, I, P],
    "cape_colsum_f32": [P, LL, I, LL, I, I, P, I, P],
    "cape_add_layernorm_fwd": [P, P, P, P, P, P, P, P, P, I, I, F, P, U32, P],
    "cape_add_layernorm_bwd": [P, P, P, P, P, P, P, P, P, P, P, I, I, F, P, U32, P],
    "cape_groupnorm_fwd": [P, P, P, P, LL, P, P, I, I, I, I, P, c_size_t, P],
    "cape_groupnorm_bwd": [P, LL, P, P, P, P, P, P, P, I, I, I, I, P, c_size_t, P],
    "cape_msda_fwd": [P, P, P, P, P, P, I, I, I, I, I, P],
    "cape_msda_bwd": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P],
    "cape_msda_bwd_ex": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "cape_msda_bwd_atomic": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P],
    "cape_attn_fwd": [P, P, P, P, P, LL, LL, LL, LL, LL, LL, LL, LL, I, I, I, I, F, I, I, P, F, P, U32, P],
    "cape_attn_bwd": [P, P, P, P, P, P, P, P, P, LL, LL, LL, LL, LL, LL, LL, LL, I, I, I, I, F, I, I, P, F, P, U32, P],
    "cape_flash_attn_fwd": [P, P, P, P, P, LL, LL, LL, LL, LL, LL, LL, LL, I, I, I, I, F, I, I, P, F, P, U32, P],
    "cape_flash_attn_bwd": [P, P, P, P, P, P, P, P, P, P, LL, LL, LL, LL, LL, LL, LL, LL, I, I, I, I, F, I, I, P, F, P, U32, P],
    "cape_attn_softmax_fwd": [P, P, P, I, I, I, I, F, I, I, P, F, P, U32, P],
    "cape_attn_softmax_bwd": [P, P, I, I, I, I, F, F, P, U32, P],
    "cape_add_f32": [P, P, P, LL, P],
    "cape_level_embed_add": [P, P, P, P, I, I, I, I, P],
    "cape_add_n_f32": [POINTER(c_void_p), I, P, LL, P],
    "cape_add_n_rows_f32": [POINTER(c_void_p), POINTER(c_longlong), I, P, LL, LL, I, P],
    "cape_augment_batch": [P, I, I, I, P, P, P],
    "cape_gelu_f32": [P, P, LL, P],
    "cape_support_masks": [P, P, P, I, I, I, P],
    "cape_interleave2x2_f32": [POINTER(c_void_p), P, P, I, I, I, I, P],
    "cape_gelu_bwd_f32": [P, P, P, LL, P],
    "cape_scale_residual_bwd_f32": [P, P, P, P, P, LL, I, P],
    "cape_scale_residual_f32": [P, P, P, P, LL, I, P],
    "cape_nchw_to_nhwc": [P, P, I, I, I, I, I, P],
    "cape_bn_fold": [P, P, P, P, F, P, P, I, P],
    "cape_maxpool3x3s2_nhwc": [P, P, I, I, I, I, P],
    "cape_bn_relu_bwd": [P, P, P, P, P, LL, I, I, P],
    "cape_affine_act_f32": [P, P, P, P, LL, I, I, P],
    "cape_relu_drop_bwd": [P, P, P, LL, F, P],
    "cape_pos_sine_level": [P, P, P, P, LL, I, I, I, I, P],
    "cape_token_embed_fwd": [P, P, P, P, P, P, P, P, P, P, LL, I, I, P],
    "cape_token_embed_bwd": [P, P, P, P, P, P, P, P, P, P, LL, I, I, I, P],
    "cape_query_sine_fwd": [P, P, P, LL, P],
    "cape_query_sine_bwd": [P, P, P, P, I, LL, P],
    "cape_refine_fwd": [P, P, P, LL, P],
    "cape_refine_bwd": [P, P, P, P, P, I, LL, P],
    "cape_sigmoid_fwd": [P, P, LL, P],
    "cape_sigmoid_bwd": [P, P, P, I, LL, P],
    "cape_ref_scale_fwd": [P, P, P, LL, I, I, P],
    "cape_ref_scale_bwd": [P, P, P, I, LL, I, I, P],
    "cape_support_embed_fwd": [P, P, P, P, P, P, P, I, I, I, P],
    "cape_support_embed_bwd": [P, P, P, P, P, I, I, I, P],
    "cape_adjacency": [P, P, P, P, I, I, P],
    "cape_gcn_aggregate_fwd": [P, P, P, I, I, I, P],
    "cape_gcn_aggregate_bwd": [P, P, P, P, I, I, I, P],
    "cape_zero_rows": [P, P, LL, I, P],
    "cape_loss_fwd_bwd": [P, P, P, P, P, P, F, F, F, P, P, P, P, I, LL, P],
    "cape_sumsq": [P, LL, P, P],
    "cape_adamw_step": [P, P, P, P, LL, F, F, F, F, F, F, P, I, P, P, P],
    "cape_step_increment": [P, P],
    "cape_decode_next_tokens": [P, P, P, P, P, P, I, I, I, I, I, I, P],
    "cape_decode_advance": [P, LL, P, LL, P, P, P, I, I, I, I, I, I, I, P, I, I, P, P, P],
    "cape_decode_linear": [POINTER(DecodeLinearDesc), P],
    "cape_decode_tail": [POINTER(DecodeTailDesc), P],
    "cape_decode_step": [POINTER(DecodeStepDesc), P],
}
EXPORTS = ["cape_last_error", "cape_abi_version", "cape_groupnorm_workspace_bytes", "cape_packed_weight_bytes"] + list(_SIGS)
_lib.cape_packed_weight_bytes.argtypes = [I, I]
_lib.cape_packed_weight_bytes.restype = c_size_t
_lib.cape_groupnorm_workspace_bytes.argtypes = [I, I, I]
_lib.cape_groupnorm_workspace_bytes.restype = c_size_t

for _name, _args in _SIGS.items():
    _fn = getattr(_lib, _name)      # AttributeError here = header/library mismatch: fail loudly
    _fn.argtypes = _args
    _fn.restype = c_int


def last_error() -> str:
    return _lib.cape_last_error().decode()


def abi_version() -> int:
    return _lib.cape_abi_version()


def call(name, *args):
    rc = getattr(_lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {last_error()}")


def raw():
    return _lib
