"""ctypes binding of libiisan_hip.so (the C ABI declared in include/iisan_hip.h).

The product path has NO fallback: if the library is missing or a symbol is absent, importing the ops fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libiisan_hip.so")

IISAN_F16, IISAN_BF16, IISAN_F32 = 0, 1, 2
MAX_LAYERS, MAX_SIDE = 48, 16

vp, i32, i64, f32, u64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_size_t


class LayerWeights(C.Structure):
    _fields_ = [(n, vp) for n in ("qkv_w", "qkv_b", "o_w", "o_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                                  "ln1_w", "ln1_b", "ln2_w", "ln2_b", "qkv_w32", "fc1_w32")]


class VitWeights(C.Structure):
    _fields_ = [("hidden", i32), ("layers", i32), ("heads", i32), ("mlp", i32),
                ("image", i32), ("patch", i32), ("channels", i32), ("dtype16", i32), ("eps", f32), ("full_blocks", i32),
                ("patch_w", vp), ("patch_b", vp), ("cls_token", vp), ("pos_emb", vp),
                ("layer", LayerWeights * MAX_LAYERS), ("folded", vp)]


class BertWeights(C.Structure):
    _fields_ = [("hidden", i32), ("layers", i32), ("heads", i32), ("mlp", i32),
                ("vocab", i32), ("max_pos", i32), ("dtype16", i32), ("eps", f32), ("full_blocks", i32),
                ("word_emb", vp), ("pos_emb", vp), ("type_emb", vp), ("emb_ln_w", vp), ("emb_ln_b", vp),
                ("layer", LayerWeights * MAX_LAYERS)]


class SideCfg(C.Structure):
    _fields_ = [("n_side", i32), ("dim_cv", i32), ("dim_text", i32), ("down", i32), ("emb", i32),
                ("gated", i32), ("gelu", i32), ("remove_first", i32), ("tap_stride_cv", i32),
                ("tap_stride_text", i32), ("tap_index", i32 * MAX_SIDE), ("first_index", i32),
                ("versa", i32), ("n_side_text", i32), ("tap_index_text", i32 * MAX_SIDE), ("first_index_text", i32),
                ("taps_exact16", i32)]


class SasrecCfg(C.Structure):
    _fields_ = [("seq", i32), ("emb", i32), ("heads", i32), ("blocks", i32), ("dropout", f32), ("seed", u64)]


# name -> (restype, argtypes); must list EVERY symbol include/iisan_hip.h declares (tests/test_abi.py checks)
SIGNATURES = {
    "iisan_version": (C.c_char_p, []),
    "iisan_arch": (C.c_char_p, []),
    "iisan_last_error": (C.c_char_p, []),
    "iisan_vit_forward_taps_ws_bytes": (sz, [C.POINTER(VitWeights), i64, i64]),
    "iisan_vit_fold_bytes": (sz, [C.POINTER(VitWeights)]),
    "iisan_vit_fold_layernorm": (i32, [C.POINTER(VitWeights), vp, sz, vp]),
    "iisan_vit_forward_taps": (i32, [C.POINTER(VitWeights), vp, i64, C.POINTER(i32), i32, vp, i64, vp, sz, vp]),
    "iisan_vit_forward_taps_u8": (i32, [C.POINTER(VitWeights), vp, i64, C.POINTER(i32), i32, vp, i64, vp, sz, vp]),
    "iisan_bert_forward_taps_ws_bytes": (sz, [C.POINTER(BertWeights), i64, i32, i64]),
    "iisan_bert_forward_taps": (i32, [C.POINTER(BertWeights), vp, i64, i32, C.POINTER(i32), i32, vp, i64, vp, sz, vp]),
    "iisan_side_net_ws_bytes": (sz, [C.POINTER(SideCfg), i64]),
    "iisan_side_net_num_params": (i32, [C.POINTER(SideCfg)]),
    "iisan_side_net_fwd": (i32, [C.POINTER(SideCfg), vp, vp, i64, C.POINTER(vp), vp, vp, sz, C.POINTER(u64), vp]),
    "iisan_side_net_bwd": (i32, [C.POINTER(SideCfg), vp, vp, i64, C.POINTER(vp), vp, C.POINTER(vp), vp, sz, u64, vp]),
    "iisan_linear_fwd": (i32, [vp, vp, vp, vp, i64, i32, i32, vp]),
    "iisan_linear_bwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "iisan_sasrec_ws_bytes": (sz, [C.POINTER(SasrecCfg), i64]),
    "iisan_sasrec_fwd": (i32, [C.POINTER(SasrecCfg), vp, vp, i64, C.POINTER(vp), vp, vp, sz, vp]),
    "iisan_sasrec_bwd": (i32, [C.POINTER(SasrecCfg), vp, vp, i64, C.POINTER(vp), vp, vp, C.POINTER(vp), vp, sz, vp]),
    "iisan_inbatch_ce_ws_bytes": (sz, [i64, i32]),
    "iisan_inbatch_ce_fwd": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, i32, vp, vp, sz, C.POINTER(u64), vp]),
    "iisan_inbatch_ce_bwd": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, f32, vp, vp, vp, sz, u64, vp]),
    "iisan_score_rank": (i32, [vp, vp, i64, i64, i32, vp, i32, vp, vp, vp]),
    "iisan_score_topk_ws_bytes": (sz, [i64, i64, i32]),
    "iisan_score_topk": (i32, [vp, vp, i64, i64, i32, vp, i32, i32, vp, vp, vp, sz, vp]),
    "iisan_adam_step": (i32, [vp, vp, vp, vp, i64, C.POINTER(i64), C.POINTER(f32), i32, i32, f32, f32, f32, f32, vp]),
    "iisan_gemm16": (i32, [i32, i32, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "iisan_layernorm768": (i32, [i32, vp, vp, vp, f32, vp, vp, i64, vp]),
    "iisan_attention16": (i32, [i32, vp, vp, vp, i64, i32, i32, vp]),
    "iisan_attention_cls16": (i32, [i32, vp, vp, vp, i64, i32, i32, vp]),
    "iisan_gemm32": (i32, [vp, vp, vp, vp, i64, i32, i64, i32, i32, i32, i32, vp]),
    "iisan_gemm_x3_ws_bytes": (sz, [i64, i32, i64]),
    "iisan_gemm_x3": (i32, [vp, vp, vp, vp, i64, i32, i64, i32, i32, i32, vp, sz, vp]),
    "iisan_gather_taps": (i32, [i32, vp, i64, vp, vp, i64, i64, vp]),
    "iisan_cast16": (i32, [i32, vp, vp, i64, vp]),
    # kernel-level entry points of the encoder GEMM's epilogue families (tests, tools)
    "iisan_gemm16_ld": (i32, [i32, i32, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    "iisan_gemm16_f32": (i32, [vp, vp, vp, i64, i32, i32, i32, vp]),
    "iisan_gemm16_lna": (i32, [i32, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "iisan_fold_ln_weights": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, vp]),
    "iisan_gemm16_stream": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "iisan_stream_stats_finalize": (i32, [vp, i32, i64, vp, vp, vp, f32, i64, i32, vp]),
    "iisan_gemm16_h256_applicable": (i32, [i32, i64, i32, i32, i32, i32, i32]),
    # DEV section: the library's only process-global state (development switches + measurement hooks)
    "iisan_dev_set": (i32, [C.c_char_p, i64]),
    "iisan_dev_get": (i64, [C.c_char_p]),
    "iisan_dev_state": (sz, [C.c_char_p, sz, i32]),
    "iisan_dev_reset": (None, []),
    "iisan_timing_enable": (None, [i32]),
    "iisan_timing_only_stream": (None, [vp, i32]),
    "iisan_timing_collect": (i64, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "iisan_timing_last_bytes": (C.c_double, []),
}

_lib = None


class IisanHipError(RuntimeError):
    pass


def load():
    """Load the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IisanHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C iisan_amd/csrc`).  There is no CPU fallback for the IISAN hot path.")
    # torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so).  If this library were loaded before torch the
    # system runtime it links against would be resident first, and a process that then initialises torch's GPU state on
    # top of it reports "no ROCm-capable device" (seen with build() followed by smoke() in one interpreter).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is absent: loud by design
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # development aid (profiling a non-default route under rocprofv3): IISAN_DEV_KNOBS="sanb_fused=0,gemm32_k64=1" sets the named
    # development switches (include/iisan_hip.h, DEV section) once at load time.  Unset in every product, test and bench run — and
    # never silent: one stderr line says which kernels were re-routed, and bench.py copies `dev_knobs()` into its `config`.
    try:
        for kv in filter(None, os.environ.get("IISAN_DEV_KNOBS", "").split(",")):
            name, val = kv.split("=")
            dev_set(name.strip(), int(val))
    except Exception:
        # a bad name / value must not leave a half-configured library cached behind a failed load(): the next load() would return
        # it with the earlier switches applied and without the stderr line below
        lib.iisan_dev_reset()
        _lib = None
        raise
    if dev_knobs():
        import sys
        print(f"iisan_amd: development switches {dev_knobs()!r} are set: product kernel routes are overridden for this process",
              file=sys.stderr, flush=True)
    return lib


def dev_knobs() -> str:
    """Every development switch of this process that is NOT at its library default ('' = none: the product routes)."""
    return dev_state()


def dev_set(name: str, value: int) -> None:
    """Set one named development switch of the library (tests / bench A/Bs only; raises on an unknown name)."""
    lib = load()
    if lib.iisan_dev_set(name.encode(), int(value)) != 0:
        raise IisanHipError(lib.iisan_last_error().decode())


def dev_get(name: str) -> int:
    lib = load()
    v = lib.iisan_dev_get(name.encode())
    if v == -(1 << 63):
        raise IisanHipError(lib.iisan_last_error().decode())
    return v


def dev_state(all_knobs: bool = False) -> str:
    """'name=value,...' of every development switch that is NOT at its library default ('' = the product routes)."""
    lib = load()
    n = lib.iisan_dev_state(None, 0, int(all_knobs))
    buf = C.create_string_buffer(n + 1)
    lib.iisan_dev_state(buf, n + 1, int(all_knobs))
    return buf.value.decode()


def dev_reset() -> None:
    load().iisan_dev_reset()


class dev:
    """Context manager: `with _lib.dev(ln_fold=0, gemm16_variant=4): ...` sets the switches and restores their previous values."""

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        self.old = {k: dev_get(k) for k in self.knobs}
        for k, v in self.knobs.items():
            dev_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            dev_set(k, v)
        return False


def check(rc: int, what: str):
    if rc != 0:
        raise IisanHipError(f"{what} failed (code {rc}): {load().iisan_last_error().decode()}")
