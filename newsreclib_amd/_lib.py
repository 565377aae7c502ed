"""ctypes binding of the C ABI declared in include/newsreclib_amd.h.

The product path has NO fallback: if the HIP library is missing or a call fails, a
``RuntimeError`` is raised (a silent PyTorch/eager path would void every parity claim).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_uint8, c_uint32, c_uint64, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libnewsreclib_amd.so")
ABI_VERSION = 16


class NrlBlockParams(ctypes.Structure):
    _fields_ = [
        ("in_proj_weight", c_void_p), ("in_proj_bias", c_void_p),
        ("out_proj_weight", c_void_p), ("out_proj_bias", c_void_p),
        ("att_weight", c_void_p), ("att_bias", c_void_p), ("att_query", c_void_p),
        ("embed_dim", c_int32), ("num_heads", c_int32), ("query_dim", c_int32), ("gemm_engine", c_int32),
        ("options", c_int32),
    ]


OPTIONS_EXPLICIT = 0x40000000


class NrlBlockGrads(ctypes.Structure):
    _fields_ = [
        ("in_proj_weight", c_void_p), ("in_proj_bias", c_void_p),
        ("out_proj_weight", c_void_p), ("out_proj_bias", c_void_p),
        ("att_weight", c_void_p), ("att_bias", c_void_p), ("att_query", c_void_p),
    ]


class NrlCnnParams(ctypes.Structure):
    _fields_ = [
        ("conv_weight", c_void_p), ("conv_bias", c_void_p), ("att_weight", c_void_p), ("att_bias", c_void_p),
        ("att_query", c_void_p),
        ("embed_dim", c_int32), ("num_filters", c_int32), ("window", c_int32), ("query_dim", c_int32),
    ]


class NrlCnnGrads(ctypes.Structure):
    _fields_ = [("conv_weight", c_void_p), ("conv_bias", c_void_p), ("att_weight", c_void_p),
                ("att_bias", c_void_p), ("att_query", c_void_p)]


class NrlGruParams(ctypes.Structure):
    _fields_ = [("weight_ih", c_void_p), ("weight_hh", c_void_p), ("bias_ih", c_void_p), ("bias_hh", c_void_p),
                ("input_dim", c_int32), ("hidden_dim", c_int32)]


class NrlGruGrads(ctypes.Structure):
    _fields_ = [("weight_ih", c_void_p), ("weight_hh", c_void_p), ("bias_ih", c_void_p), ("bias_hh", c_void_p)]


class NrlAddAttParams(ctypes.Structure):
    _fields_ = [("att_weight", c_void_p), ("att_bias", c_void_p), ("att_query", c_void_p),
                ("dim", c_int32), ("query_dim", c_int32)]


class NrlAddAttGrads(ctypes.Structure):
    _fields_ = [("att_weight", c_void_p), ("att_bias", c_void_p), ("att_query", c_void_p)]


class NrlMhaParams(ctypes.Structure):
    _fields_ = [("in_proj_weight", c_void_p), ("in_proj_bias", c_void_p), ("out_proj_weight", c_void_p),
                ("out_proj_bias", c_void_p), ("embed_dim", c_int32), ("num_heads", c_int32), ("scale", ctypes.c_float),
                ("gemm_engine", c_int32)]


class NrlMhaGrads(ctypes.Structure):
    _fields_ = [("in_proj_weight", c_void_p), ("in_proj_bias", c_void_p), ("out_proj_weight", c_void_p),
                ("out_proj_bias", c_void_p)]


# name -> (restype, argtypes); mirrors include/newsreclib_amd.h one to one
SIGNATURES = {
    "nrl_abi_version": (c_int32, []),
    "nrl_build_id": (c_char_p, []),
    "nrl_last_error": (c_char_p, []),
    "nrl_set_gemm_engine": (c_int32, [c_int32]),
    "nrl_get_gemm_engine": (c_int32, []),
    "nrl_set_option": (c_int32, [c_char_p, c_int32]),
    "nrl_get_options": (c_int32, []),
    "nrl_prof_enable": (c_int32, [c_int32]),
    "nrl_prof_read": (c_int32, [POINTER(c_double), POINTER(c_int64), POINTER(c_double)]),
    "nrl_dropout_key": (c_uint32, [c_uint64, c_uint32]),
    "nrl_dropout_mask": (c_int32, [c_void_p, c_int64, c_double, c_uint64, c_uint32, c_void_p]),
    "nrl_sort_positions_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "nrl_sort_positions": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nrl_news_encoder_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32, c_int32, c_int32]),
    "nrl_news_encoder_fwd": (c_int32, [POINTER(NrlBlockParams), c_void_p, c_int64, c_void_p, c_int64, c_int32,
                                       c_double, c_uint64, c_uint32, c_int32, c_void_p, c_void_p, c_size_t,
                                       c_void_p]),
    "nrl_token_table_supported": (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    "nrl_token_table_bytes": (c_size_t, [c_int64, c_int32, c_int32, c_int32]),
    "nrl_token_table_build": (c_int32, [POINTER(NrlBlockParams), c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "nrl_news_encoder_fwd_table_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "nrl_news_encoder_fwd_table": (c_int32, [POINTER(NrlBlockParams), c_void_p, c_size_t, c_int64, c_void_p, c_int64, c_int32,
                                             c_void_p, c_void_p, c_size_t, c_void_p]),
    "nrl_news_encoder_bwd": (c_int32, [POINTER(NrlBlockParams), POINTER(NrlBlockGrads), c_void_p, c_void_p, c_int64,
                                       c_void_p, c_void_p, c_int64, c_int32, c_double, c_uint64, c_uint32,
                                       c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "nrl_user_encoder_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int32, c_int32, c_int32]),
    "nrl_user_encoder_fwd": (c_int32, [POINTER(NrlBlockParams), c_void_p, c_int64, c_int64, c_double, c_uint64,
                                       c_uint32, c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nrl_user_encoder_bwd": (c_int32, [POINTER(NrlBlockParams), POINTER(NrlBlockGrads), c_void_p, c_int64,
                                       c_int64, c_double, c_uint64, c_uint32, c_int32, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_void_p]),
    "nrl_user_encoder_bwd_phase": (c_int32, [POINTER(NrlBlockParams), POINTER(NrlBlockGrads), c_void_p, c_int64,
                                             c_int64, c_double, c_uint64, c_uint32, c_int32, c_void_p, c_void_p, c_int32,
                                             c_void_p, c_size_t, c_void_p]),
    "nrl_to_dense_batch_fwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p]),
    "nrl_to_dense_batch_bwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int64, c_void_p,
                                         c_void_p]),
    "nrl_hist_mean_fwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p]),
    "nrl_hist_mean_bwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p]),
    "nrl_dot_scores_fwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p]),
    "nrl_dot_scores_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p,
                                     c_void_p, c_void_p]),
    "nrl_ce_loss_fwd_bwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p, c_void_p,
                                      c_void_p]),
    "nrl_supcon_loss_fwd_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, ctypes.c_float,
                                          ctypes.c_float, c_void_p, c_void_p, c_void_p]),
    "nrl_adam_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double, c_double,
                                c_double, c_int64, c_float, c_int32, c_void_p]),
    "nrl_adam_rows_mark": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "nrl_adam_rows_advance": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                                        c_int64, c_int64, c_int64, c_int32, c_double, c_double, c_double, c_double, c_float,
                                        c_void_p, c_int32, c_void_p]),
    "nrl_dropout_add_layernorm_fwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_double,
                                                c_uint64, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nrl_dropout_add_layernorm_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_double,
                                                c_uint64, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nrl_cnn_encoder_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "nrl_cnn_encoder_fwd": (c_int32, [POINTER(NrlCnnParams), c_void_p, c_int64, c_void_p, c_int64, c_int32,
                                      c_double, c_uint64, c_uint32, c_int32, c_void_p, c_void_p, c_size_t,
                                      c_void_p]),
    "nrl_cnn_encoder_bwd": (c_int32, [POINTER(NrlCnnParams), POINTER(NrlCnnGrads), c_void_p, c_int64, c_void_p,
                                      c_void_p, c_int64, c_int32, c_double, c_uint64, c_uint32, c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "nrl_cnn_mhsa_encoder_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "nrl_cnn_mhsa_encoder_fwd": (c_int32, [POINTER(NrlCnnParams), POINTER(NrlBlockParams), c_void_p, c_int64, c_void_p,
                                           c_int64, c_int32, c_double, c_uint64, c_uint32, c_int32, c_void_p, c_void_p,
                                           c_size_t, c_void_p]),
    "nrl_cnn_mhsa_encoder_bwd": (c_int32, [POINTER(NrlCnnParams), POINTER(NrlCnnGrads), POINTER(NrlBlockParams),
                                           POINTER(NrlBlockGrads), c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                           c_int32, c_double, c_uint64, c_uint32, c_void_p, c_void_p, c_size_t,
                                           c_void_p]),
    "nrl_embedding_rows_fwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_double, c_uint64, c_uint32,
                                         c_void_p, c_void_p]),
    "nrl_embedding_rows_bwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_double, c_uint64, c_uint32,
                                         c_void_p, c_void_p]),
    "nrl_gru_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int32, c_int32]),
    "nrl_gru_fwd": (c_int32, [POINTER(NrlGruParams), c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32,
                              c_void_p, c_void_p, c_size_t, c_void_p]),
    "nrl_gru_bwd": (c_int32, [POINTER(NrlGruParams), POINTER(NrlGruGrads), c_void_p, c_void_p, c_void_p, c_int64,
                              c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nrl_additive_attention_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int32, c_int32]),
    "nrl_additive_attention_fwd": (c_int32, [POINTER(NrlAddAttParams), c_void_p, c_int64, c_int64, c_int32, c_void_p,
                                             c_void_p, c_size_t, c_void_p]),
    "nrl_additive_attention_bwd": (c_int32, [POINTER(NrlAddAttParams), POINTER(NrlAddAttGrads), c_void_p, c_int64,
                                             c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nrl_mha_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int32, c_int32]),
    "nrl_mha_fwd": (c_int32, [POINTER(NrlMhaParams), c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_size_t,
                              c_void_p]),
    "nrl_mha_bwd": (c_int32, [POINTER(NrlMhaParams), POINTER(NrlMhaGrads), c_void_p, c_int64, c_int64, c_void_p,
                              c_void_p, c_void_p, c_size_t, c_void_p]),
    "nrl_linear_act_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "nrl_linear_act_fwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    "nrl_linear_act_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nrl_embedding_gather": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "nrl_linear_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "nrl_linear_fwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                 c_size_t, c_void_p]),
    "nrl_linear_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_size_t, c_void_p]),
    "nrl_sdpa_supported": (c_int32, [c_int64, c_int32, c_int32, c_int32]),
    "nrl_sdpa_fwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_float, c_double,
                               c_uint64, c_uint32, c_void_p, c_void_p, c_void_p]),
    "nrl_sdpa_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                               c_int32, c_float, c_double, c_uint64, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nrl_linear_fwd_img": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                     c_size_t, c_int32, c_void_p]),
    "nrl_linear_gelu_fwd_img": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                          c_size_t, c_int32, c_void_p]),
    "nrl_linear_dgrad_gelu_img": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_size_t,
                                            c_int32, c_void_p]),
    "nrl_linear_gelu_supported": (c_int32, [c_int32]),
    "nrl_linear3_supported": (c_int32, [c_int32, c_int32]),
    "nrl_linear3_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "nrl_linear3_fwd_img": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                      c_void_p, c_void_p, c_size_t, c_int32, c_void_p]),
    "nrl_linear3_dgrad_img": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                        c_void_p, c_size_t, c_int32, c_void_p]),
    "nrl_linear_dgrad_add_img": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_size_t,
                                           c_int32, c_void_p]),
    "nrl_embedding_grad": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p]),
    "nrl_linear_bwd_img": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_int32, c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load (once) and type the shared library; raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64: it must be the HIP runtime of the process (it owns the device
    # context, streams and allocator the C ABI is handed), so torch is loaded BEFORE this library binds to one.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension was not built (run `python -m newsreclib_amd._build` "
            "or `__graft_entry__.build()`); newsreclib_amd has no CPU/PyTorch fallback by design.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    got = lib.nrl_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"ABI version mismatch: library {got}, binding {ABI_VERSION} (rebuild)")
    # the library must have been built from the sources lying beside it (content hash, not file times)
    from . import _build
    if os.path.isdir(_build.CSRC):
        want, have = _build.source_hash(), lib.nrl_build_id().decode()
        if want != have:
            raise RuntimeError(f"{LIB_PATH} was built from other sources (build id {have}, sources {want}): run "
                               "`python -m newsreclib_amd._build` or `__graft_entry__.build()`")
    _lib = lib
    eng = os.environ.get("NRL_GEMM_ENGINE")
    if eng:
        set_gemm_engine(eng)
    return lib


ENGINES = {"f32": 0, "bf16x3": 1}


def set_gemm_engine(name: str) -> None:
    """"f32" = exact fp32 MFMA; "bf16x3" = fp32 via three bf16 MFMAs per product (default)."""
    if name not in ENGINES:
        raise ValueError(f"unknown GEMM engine {name!r}; choose from {sorted(ENGINES)}")
    check(load().nrl_set_gemm_engine(ENGINES[name]), "nrl_set_gemm_engine")


def get_gemm_engine() -> str:
    code = load().nrl_get_gemm_engine()
    return {v: k for k, v in ENGINES.items()}[code]


# bit order of the switch mask (include/newsreclib_amd.h, nrl_set_option)
OPTION_NAMES = ("news_fused", "news_fused_bwd", "news_attn_mfma", "news_planes", "news_od_planes", "news_aa_planes",
                "wgrad_2step", "wgrad_ws", "rowpanel", "x3_dma", "news_tail", "news_tail_bwd", "user_fork", "news_fork",
                "news_qkv_planes", "news_pad_share", "news_tail_od", "user_proj")


def set_option(name: str, value: bool) -> None:
    """Kernel-selection switch ("news_fused", "rowpanel", "x3_dma") for A/B measurements and equivalence tests."""
    check(load().nrl_set_option(name.encode(), int(bool(value))), "nrl_set_option")


def options_mask() -> int:
    """Bit mask of the kernel-selection switches (they choose private workspace formats: an autograd forward records it
    and its backward refuses to run under another one)."""
    return int(load().nrl_get_options())


def options_word() -> int:
    """The per-call switch word of the current process defaults (``NrlBlockParams.options``): captured by an autograd
    forward and handed to its backward, so the backward reads the workspace in the format the forward wrote even if the
    defaults changed in between (and two modules may run under different switches)."""
    return OPTIONS_EXPLICIT | options_mask()


def require_options(mask: int, what: str) -> None:
    if options_mask() != mask:
        raise RuntimeError(f"newsreclib_amd: a kernel-selection switch (nrl_set_option / NRL_* environment) changed between "
                           f"the forward and the backward of {what}; the saved workspace is in the forward's format")


def engine_code() -> int:
    """The per-call engine value (1 = f32, 2 = bf16x3) of the current process default: captured by every
    autograd forward and handed to its backward, so a default changed in between cannot mix engines."""
    return load().nrl_get_gemm_engine() + 1


def require_engine(code: int, what: str) -> None:
    """Backward of an entry point whose params carry no per-call engine: refuse to run under another engine
    than the forward did (the saved workspace holds that engine's weight planes / layouts)."""
    if engine_code() != code:
        raise RuntimeError(f"newsreclib_amd: the GEMM engine changed between the forward and the backward of {what}; "
                           "set the engine before the forward and keep it until the backward has run")


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().nrl_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
