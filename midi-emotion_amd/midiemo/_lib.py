"""ctypes binding of libmidiemo_hip.so (C-ABI declared in include/midiemo.h).

There is NO fallback: if the library is missing or a symbol is absent the
import of the op layer raises, and every op raises RuntimeError on a non-zero
status code."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIDIEMO_LIB") or os.path.join(HERE, "libmidiemo_hip.so")

ME_F32, ME_BF16, ME_F16 = 0, 1, 2
ME_COND_NONE, ME_COND_CONCAT, ME_COND_TOKEN = 0, 1, 2
ME_EPI_RELU, ME_EPI_OUT_F32, ME_EPI_RELU_BWD = 1, 2, 4
ME_WS_GEMM_TN, ME_WS_RGA_PT, ME_WS_RGA_DGT, ME_WS_RGA_MT, ME_WS_GEMM_TN_GROUP, ME_WS_EMBED_BWD, ME_WS_SUMSQ, ME_WS_RELU_MASK, ME_WS_DEC_TOKEN = 1, 2, 3, 4, 5, 6, 7, 8, 9
ME_TN_MAX_GROUP = 5
ABI_VERSION = 24
ME_LO8 = 0x100             # OR-ed into the dtype of me_resid_ln_fwd / me_embed_fwd: 8-bit low halves of the residual stream
ME_DEC_MAX_LAYERS, ME_DEC_TOKEN_ROWS = 16, 4
# words of the loss-scaler state (me_scaler_step; enum ME_SCALER_* of include/midiemo.h)
ME_SCALER_SCALE, ME_SCALER_INV, ME_SCALER_TRACKER, ME_SCALER_STEP, ME_SCALER_FOUND_INF, ME_SCALER_SKIPPED, ME_SCALER_WORDS = 0, 1, 2, 3, 4, 5, 8

ERRORS = {0: "ME_OK", -1: "ME_ERR_BAD_DTYPE", -2: "ME_ERR_BAD_SHAPE", -3: "ME_ERR_ALIGNMENT",
          -4: "ME_ERR_LAUNCH", -5: "ME_ERR_NULL", -6: "ME_ERR_WORKSPACE"}

_p, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_i64, _u64, _u32 = ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint32

# name -> argtypes, exactly the prototypes of include/midiemo.h
SIGNATURES = {
    "me_abi_version": [],
    "me_cast_transpose": [_p, _i, _i, _p, _i, _p, _i, _i, _p],
    "me_cast_transpose_multi": [_p, _i, _i, _i, _p],
    "me_embed_fwd": [_p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _u64, _p],
    "me_embed_bwd": [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _u64, _p, ctypes.c_size_t, _p],
    "me_key_pad_mask": [_p, _p, _i, _i, _i, _i, _p],
    "me_gemm_nt": [_p, _i, _p, _i, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "me_gemm_nt_relu_mask": [_p, _i, _p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    "me_workspace_bytes": [_i, _i, _i, _i, _i],
    "me_gemm_tn_acc": [_p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _p, ctypes.c_size_t, _i, _p],
    "me_gemm_tn_acc_group": [_p, _i, _i, _p, ctypes.c_size_t, _i, _p],
    "me_rga_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "me_rga_pack_rel": [_p, _p, _i, _i, _i, _p],
    "me_rga_bwd": [_p] * 11 + [_i, _i, _i, _i, _i, _i, _i, _i, _p],
    "me_rga_bwd_phases": [_p] * 11 + [_i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "me_resid_ln_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _f, _u64, _u32, _i, _p],
    "me_resid_ln_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _u64, _u32, _i, _p],
    "me_ce_fwd": [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "me_ce_bwd": [_p, _i, _p, _p, _p, _i, _p, _f, _p, _p, _i, _i, _i, _i, _i, _p],
    "me_sumsq": [_p, _i64, _p, _p, ctypes.c_size_t, _p],
    "me_adamw_step": [_p, _p, _p, _p, _i64, _p, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _p, _p],
    "me_scaler_step": [_p, _p, _f, _f, _i, _p],
    "me_dec_qkv": [_p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    "me_dec_embed_qkv": [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    "me_dec_embed_qkv_attn": [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    "me_dec_ln_qkv_attn": [_p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    "me_dec_attn": [_p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    "me_dec_proj_resid": [_p, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p],
    "me_dec_ln_proj": [_p, _p, _p, _f, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "me_dec_token": [_p, _p, _p, _p, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, ctypes.c_size_t, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _f, _i, _i, _p],
    "me_dec_token_blocks": [_i, _i, _i, _i],
    "me_greedy_pick": [_p, _i, _i, _p, _i, _p, _i, _p],
    "me_sample_topk_topp": [_p, _i, _i, _p, _i, _p, _i, _f, _p, _p, _p, _p, _p, _i, _p],
    "me_sample_step": [_p, _i, _i, _p, _i, _p, _p, _p, _f, _f, _f, _i, _f, _p, _i, _p, _i, _p, _p, _i, _p],
    "me_decode_commit": [_p, _p, _i, _p, _i, _p],
    "me_greedy_pick_commit": [_p, _i, _i, _p, _i, _p, _p, _i, _p, _i, _p],
}



class TnItem(ctypes.Structure):
    """struct me_tn_item of include/midiemo.h (one weight gradient of a grouped launch)."""
    _fields_ = [("A", _p), ("lda", _i), ("B", _p), ("ldb", _i), ("dW", _p), ("lddw", _i), ("dbias", _p), ("N", _i), ("K", _i)]


class DecLayer(ctypes.Structure):
    """struct me_dec_layer of include/midiemo.h (one layer's pointers for me_dec_token; the table lives in DEVICE memory)."""
    _fields_ = [(n, _p) for n in ("Wqkv", "bqkv", "Wo", "bo", "W1", "b1", "W2", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "E", "kcache", "vcache")]


_lib = None


def load():
    """Load the shared library and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it brings its own HIP runtime (libamdhip64); loading this library before it pulls a second copy
    # from /opt/rocm into the process and whichever initialises second reports "no ROCm-capable device"
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libmidiemo_hip.so not built (%s). Run `python __graft_entry__.py` or "
            "`python midi-emotion_amd/midiemo/build.py`; there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = ctypes.c_size_t if name == "me_workspace_bytes" else ctypes.c_int
    v = lib.me_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError("libmidiemo_hip.so ABI %d != binding ABI %d" % (v, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, ERRORS.get(rc, "unknown"), rc))
