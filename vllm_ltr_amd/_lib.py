"""ctypes binding of libltr_hip.so (include/ltr_hip.h).

The product path has no CPU fallback: if the library is missing, or a call
returns an error code, this raises.  PyTorch is used by the callers only for
device memory, streams and torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# LTR_LIB (diag): another build of the same library (lab variants of one kernel file); there is still no fallback
LIB_PATH = os.environ.get("LTR_LIB") or os.path.join(_HERE, "csrc", "libltr_hip.so")

LTR_W_F32, LTR_W_F16 = 0, 1
LTR_WS_SCORE, LTR_WS_RANK = 0, 1
LTR_RANK_USE_PRI, LTR_RANK_ASCENDING = 1, 2
LTR_WT_GLOBAL_COUNT, LTR_WL_COUNT = 7, 12

# every symbol include/ltr_hip.h declares (tests check the library exports them all)
SYMBOLS = ("ltr_abi_version", "ltr_last_error", "ltr_create", "ltr_destroy", "ltr_workspace_bytes",
           "ltr_set_chunk_tokens", "ltr_score", "ltr_forward_hidden", "ltr_embed_gather", "ltr_pool_head",
           "ltr_rank_step", "ltr_age_update", "ltr_budget_prefix", "ltr_profile_enable", "ltr_profile_read",
           "ltr_head_create", "ltr_head_destroy", "ltr_head_score", "ltr_reserve_select", "ltr_listmle",
           "ltr_neuralndcg", "ltr_neuralndcg_workspace_bytes",
           "ltr_status", "ltr_queue_step", "ltr_train_create", "ltr_train_destroy", "ltr_train_workspace_bytes",
           "ltr_train_step", "ltr_train_read", "ltr_attention", "ltr_train_attention",
           "ltr_train_attention_workspace_bytes")
ABI_VERSION = 7


LTR_E_INVAL, LTR_E_RANGE = -22, -34


class LtrError(RuntimeError):
    """``code``: the LTR_E_* value the library returned (0 when raised by the host side)."""

    def __init__(self, msg: str, code: int = 0):
        super().__init__(msg)
        self.code = int(code)


class ModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "vocab_size", "hidden_size", "ffn_dim", "num_layers", "num_heads", "word_embed_proj_dim",
        "pos_rows", "num_labels", "pre_ln", "weight_dtype", "flags")]


LTR_F_NO_LN_FOLD, LTR_F_ONE_PASS = 1, 4          # (2, 8: the lane flags of ABI 5, accepted and ignored)


class HeadDesc(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("n_fc", C.c_int32), ("fc_sizes", C.c_int32 * 8),
                ("input_norm", C.c_int32), ("activation", C.c_int32), ("d_output", C.c_int32),
                ("output_activation", C.c_int32), ("weight_dtype", C.c_int32)]


ACTIVATIONS = {None: 0, "Identity": 0, "ReLU": 1, "Sigmoid": 2, "Tanh": 3, "GELU": 4, "SiLU": 5}


class TrainConfig(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("loss", C.c_int32), ("listmle_eps", C.c_float), ("pad_value", C.c_float),
                ("dropout", C.c_float), ("precision", C.c_int32), ("seed", C.c_uint64)]


LOSSES = {"listMLE": 0, "mse": 1, "crossentropy": 2, "neuralNDCG": 3}
TRAIN_PRECISIONS = {None: 0, "split": 1, "f32": 2}


class ProfileStats(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("work", C.c_double * 8), ("launches", C.c_int64 * 8)]


PROFILE_KINDS = ("gemm", "attn", "embed", "ln", "pool", "gemm_small")

_lib = None
_load_lock = threading.Lock()


def load() -> C.CDLL:
    """Load libltr_hip.so (built in-tree by ``python -m vllm_ltr_amd.csrc.build``).  Thread-safe."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:
        if _lib is None:
            _lib = _load()
    return _lib


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise LtrError(f"{LIB_PATH} not found - build it with `python -m vllm_ltr_amd.csrc.build` "
                       "(there is no CPU fallback for the ranking path)")
    # PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7, same as /opt/rocm's).
    # It must be in the process BEFORE libltr_hip.so is loaded so that the library binds to the
    # SAME HIP runtime: the stream handles and device pointers handed over come from torch's
    # runtime, and a second runtime instance does not see them ("no ROCm-capable device").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_size_t
    lib.ltr_abi_version.restype = C.c_int
    lib.ltr_last_error.restype = C.c_char_p
    lib.ltr_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(vp), i32, vp, C.POINTER(vp)]
    lib.ltr_status.argtypes = [vp, vp]
    lib.ltr_destroy.argtypes = [vp]
    lib.ltr_workspace_bytes.argtypes = [vp, i32, i64, i64]
    lib.ltr_workspace_bytes.restype = sz
    lib.ltr_set_chunk_tokens.argtypes = [vp, i32]
    lib.ltr_score.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, sz, vp]
    lib.ltr_forward_hidden.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]
    lib.ltr_embed_gather.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    lib.ltr_pool_head.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    lib.ltr_attention.argtypes = [vp, vp, vp, i32, i32, vp, vp, sz, vp]
    lib.ltr_rank_step.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, u32, vp, vp, sz, vp]
    lib.ltr_age_update.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, vp]
    lib.ltr_queue_step.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, u32, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, sz, vp]
    lib.ltr_budget_prefix.argtypes = [vp, vp, vp, vp, i32, i64, i64, vp, vp, vp, vp]
    lib.ltr_profile_enable.argtypes = [vp, i32]
    lib.ltr_profile_read.argtypes = [vp, C.POINTER(ProfileStats), i32]
    lib.ltr_reserve_select.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, C.c_int64, vp, vp, vp, vp]
    lib.ltr_listmle.argtypes = [vp, vp, vp, i32, i32, C.c_float, C.c_float, vp, vp, vp, vp]
    lib.ltr_neuralndcg.argtypes = [vp, vp, i32, i32, C.c_float, i32, C.c_float, vp, vp, vp, vp, sz, vp]
    lib.ltr_neuralndcg_workspace_bytes.argtypes = [i32, i32]
    lib.ltr_neuralndcg_workspace_bytes.restype = sz
    lib.ltr_head_create.argtypes = [C.POINTER(HeadDesc), C.POINTER(vp), i32, C.POINTER(vp)]
    lib.ltr_head_destroy.argtypes = [vp]
    lib.ltr_head_score.argtypes = [vp, vp, vp, i32, vp, vp]
    lib.ltr_train_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(vp), i32, C.POINTER(TrainConfig), vp, C.POINTER(vp)]
    lib.ltr_train_destroy.argtypes = [vp]
    lib.ltr_train_workspace_bytes.argtypes = [vp, i64, i64]
    lib.ltr_train_workspace_bytes.restype = sz
    lib.ltr_train_step.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, sz, vp]
    lib.ltr_train_read.argtypes = [vp, i32, i32, vp, sz, C.POINTER(sz), vp]
    lib.ltr_train_attention_workspace_bytes.argtypes = [i32, i64, i64]
    lib.ltr_train_attention_workspace_bytes.restype = sz
    lib.ltr_train_attention.argtypes = [i32, vp, vp, vp, i32, i32, vp, vp, vp, sz, vp]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("ltr_last_error", "ltr_workspace_bytes", "ltr_abi_version", "ltr_train_workspace_bytes",
                        "ltr_train_attention_workspace_bytes", "ltr_neuralndcg_workspace_bytes"):
            fn.restype = C.c_int
    if lib.ltr_abi_version() != ABI_VERSION:
        raise LtrError(f"{LIB_PATH} has ABI version {lib.ltr_abi_version()}, this binding needs {ABI_VERSION}: rebuild")
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ltr_last_error().decode(errors="replace")
        raise LtrError(f"{what} failed with code {rc}: {msg}", rc)
