"""ctypes binding of libcm3p_hip.so (declared in include/cm3p_hip.h).

This is the only place the C ABI is crossed.  Every wrapper passes raw device pointers (`tensor.data_ptr()`), sizes
and the current HIP stream, checks the integer return code and raises on failure.  There is no CPU fallback: if the
library cannot be loaded, or a tensor is not on a GPU, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

import torch

F32, BF16 = 0, 1
EPI_BF16, EPI_F32, EPI_F32_RESID, EPI_F32_BIAS = 0, 1, 2, 5
ABI_VERSION = 16

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CM3P_HIP_LIB") or os.path.join(_HERE, "csrc", "libcm3p_hip.so")  # env override: kernel experiments

_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float
_RETURNS_INT64 = {"cm3p_attn_bwd_fused_workspace_bytes", "cm3p_token_order_workspace_ints"}  # size queries that do not fit an int

# name -> argtypes, mirrors include/cm3p_hip.h one to one
SIGNATURES = {
    "cm3p_abi_version": [],
    "cm3p_layernorm_fwd": [_P, _I, _P, _P, _P, _P, _P, _L, _I, _F, _P],
    "cm3p_layernorm_bwd_blocks": [_L],
    "cm3p_layernorm_bwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P],
    "cm3p_embed_ln_fwd": [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _L, _I, _F, _L, _P],
    "cm3p_embed_ln_bwd": [_P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _L, _I, _L, _L, _P],
    "cm3p_embed_ln_bwd_sorted_chunk": [],
    "cm3p_embed_ln_bwd_sorted": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _L, _L, _P],
    "cm3p_token_order_workspace_ints": [_L, _L],
    "cm3p_token_order": [_P, _L, _L, _P, _P, _P, _P],
    "cm3p_audio_slots": [_P, _L, _L, _P, _P, _P],
    "cm3p_gemm_bf16": [_P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _I, _I, _I, _I, _P, _P],
    "cm3p_qkv_gemm_rope": [_P, _P, _P, _L, _L, _L, _P, _P, _I, _I, _I, _F, _P],
    "cm3p_gemm_wgrad_splits": [_L, _L, _L],
    "cm3p_gemm8p_set_grid": [_I],
    "cm3p_gemm8p_get_grid": [],
    "cm3p_build_ablation_flags": [],
    "cm3p_debug_set_dma_audit": [_P],
    "cm3p_cast_f32_bf16": [_P, _P, _L, _P],
    "cm3p_cast_f32_bf16_t": [_P, _P, _P, _L, _L, _P],
    "cm3p_cast_f32_bf16_t_multi": [_P, _I, _L, _P],
    "cm3p_add_f32": [_P, _P, _I, _P, _P, _L, _P],
    "cm3p_rope_table": [_P, _L, _P, _I, _P, _P, _P],
    "cm3p_rope_apply": [_P, _P, _P, _I, _I, _I, _L, _I, _P],
    "cm3p_attn_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "cm3p_attn_fwd_impl": [_I, _I, _I, _I],
    "cm3p_attn_probs": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "cm3p_attn_generic_supported": [_I],
    "cm3p_attn_fwd_generic": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "cm3p_attn_bwd_generic": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "cm3p_rope_apply_generic": [_P, _P, _P, _I, _I, _I, _I, _L, _I, _P],
    "cm3p_attn_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _L, _I, _I, _P],
    "cm3p_geglu_fwd": [_P, _P, _L, _I, _P],
    "cm3p_gemm_geglu": [_P, _P, _P, _L, _L, _L, _P],
    "cm3p_geglu_bwd": [_P, _P, _P, _L, _I, _P],
    "cm3p_gelu_fwd": [_P, _P, _L, _P],
    "cm3p_gelu_bwd": [_P, _P, _P, _L, _P],
    "cm3p_im2col_k3": [_P, _I, _P, _I, _I, _I, _I, _I, _P],
    "cm3p_col2im_k3": [_P, _P, _I, _I, _I, _I, _I, _P],
    "cm3p_bias_gelu_fwd": [_P, _P, _P, _P, _L, _I, _P],
    "cm3p_bias_gelu_bwd_blocks": [_L],
    "cm3p_bias_gelu_bwd": [_P, _I, _P, _P, _P, _P, _P, _L, _I, _P],
    "cm3p_pool_chunks": [_I],
    "cm3p_pool_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cm3p_pool_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cm3p_gemm_f32": [_P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _L, _F, _I, _P],
    "cm3p_l2norm_fwd": [_P, _P, _P, _I, _I, _P],
    "cm3p_l2norm_bwd": [_P, _P, _P, _P, _I, _I, _P],
    "cm3p_cross_entropy": [_P, _I, _I, _L, _L, _P, _P, _F, _P, _P, _P],
    "cm3p_cross_entropy_masked": [_P, _L, _I, _L, _P, _L, _F, _P, _P, _P, _P],
    "cm3p_inv_valid_count": [_P, _L, _L, _P, _P],
    "cm3p_ce_masked_stats": [_P, _L, _I, _L, _P, _L, _P, _P, _P],
    "cm3p_ce_masked_dlogits_blocks": [_L],
    "cm3p_ce_masked_dlogits_bf16": [_P, _L, _I, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P],
    "cm3p_add_bias_f32": [_P, _P, _L, _I, _P],
    "cm3p_colsum_blocks": [_L],
    "cm3p_colsum_f32": [_P, _P, _P, _L, _I, _P],
    "cm3p_first_zero_index": [_P, _I, _I, _P, _P],
    "cm3p_scale_exp": [_P, _P, _P, _L, _P],
    "cm3p_scale_by": [_P, _P, _P, _L, _P],
    "cm3p_dot_f32": [_P, _P, _P, _L, _P],
    "cm3p_sum_f32": [_P, _P, _L, _F, _I, _P],
    "cm3p_pointwise_loss": [_P, _P, _P, _P, _L, _I, _P],
    "cm3p_attn_fwd_varlen": [_P, _P, _P, _P, _I, _I, _L, _I, _I, _F, _I, _P],
    "cm3p_attn_bwd_fused_workspace_bytes": [_I, _I, _I],
    "cm3p_attn_bwd_fused_slab_group": [_I],
    "cm3p_attn_bwd_fused": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _F, _P, _P, _L, _I, _I, _P, _L, _P],
    "cm3p_attn_bwd_varlen": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _I, _F, _P, _P, _I, _I, _P],
    "cm3p_gather_rows_f32": [_P, _P, _P, _L, _I, _P],
    "cm3p_scatter_rows_f32": [_P, _P, _P, _L, _I, _P],
    "cm3p_gemm_bf16_batched": [_P, _P, _P, _P, _I, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _I, _F, _F, _P],
    "cm3p_muon_partials": [_I, _I],
    "cm3p_muon_momentum": [_P, _P, _P, _P, _I, _I, _I, _I, _L, _F, _I, _I, _P],
    "cm3p_muon_normalize": [_P, _P, _I, _I, _I, _L, _F, _P],
    "cm3p_muon_apply": [_P, _P, _I, _I, _I, _I, _L, _F, _F, _P],
    "cm3p_adamw_multi": [_P, _P, _P, _P, _P, _I, _L, _F, _F, _F, _F, _F, _P],
}

_lib = None


class Cm3pHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """dlopen the kernel library (once) and attach the prototypes.  Raises if it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Cm3pHipError(
            f"{LIB_PATH} not found: build it with `python -m cm3p_amd.build` (or __graft_entry__.build()). "
            "cm3p_amd has no CPU or PyTorch fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = c_int64 if name in _RETURNS_INT64 else c_int
    if lib.cm3p_abi_version() != ABI_VERSION:
        raise Cm3pHipError(f"ABI mismatch: library {lib.cm3p_abi_version()} vs binding {ABI_VERSION}; rebuild")
    if lib.cm3p_build_ablation_flags() != 0 and os.environ.get("CM3P_ALLOW_ABLATED_LIB") != "1":
        raise Cm3pHipError(
            f"{LIB_PATH} was built with timing-only ablation macros (mask {lib.cm3p_build_ablation_flags():#x}): its results are wrong by "
            "construction.  Rebuild with `python -m cm3p_amd.build --force`; kernel-timing scripts set CM3P_ALLOW_ABLATED_LIB=1.")
    _lib = lib
    return lib


def _check(rc: int, name: str):
    if rc != 0:
        raise Cm3pHipError(f"{name} failed with code {rc} ({'invalid argument' if rc == -1 else 'launch failure'})")


class _DevPtr(int):
    """A device address that remembers which GPU it lives on (ctypes takes it as the plain integer it is)."""


def ptr(t: torch.Tensor | None, dtype: torch.dtype | None = None):
    """Device address of a contiguous GPU tensor.  `dtype`: what the C entry point reads there - a kernel handed int32 token ids
    where it reads int64 walks off the end of the embedding table (a GPU memory fault, not an exception), so every index / mask /
    single-dtype argument is checked here and a mismatch is a TypeError.  The address carries its device index: call() launches
    on that device and refuses arguments that live on different GPUs."""
    if t is None:
        return None
    if not t.is_cuda:
        raise Cm3pHipError("cm3p_amd kernels need GPU tensors; there is no CPU fallback")
    if not t.is_contiguous():
        raise Cm3pHipError("cm3p_amd kernels need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"cm3p_amd kernel argument has dtype {t.dtype}, the kernel reads {dtype}")
    p = _DevPtr(t.data_ptr())
    p.dev = t.device.index
    return p


class _StreamOfCall:
    """Placeholder for "the current HIP stream of the device the tensor arguments live on"; call() resolves it."""


_STREAM = _StreamOfCall()


def stream():
    return _STREAM


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise Cm3pHipError(f"unsupported dtype {t.dtype}")


# Optional per-launch timing with HIP events on the launching stream (bench.py's roofline leg).  When `_prof` is a list,
# every call appends (tag, start_event, end_event, work) where `work` is the algorithmic flop (or byte) count the
# caller attached.  Off (None) by default: zero overhead on the product path.
_prof = None
_prof_only = None
HBM_BOUND_TAGS = set()  # tags whose `work` is algorithmic bytes rather than FLOPs (filled by kernels.py)


_prof_every = 1
_prof_seen = 0


def profile_begin(only=None, every: int = 1):
    """Start timing C-ABI calls with HIP events on the current stream.  `only`: time just the calls with this tag - an event pair
    around every launch costs the stream about 2.5 us each, ~5 ms of a C2 step, so the judged region of bench.py brackets only
    the dominant kernel's launches - and of those only every `every`-th one (the first, the every+1-th, ...)."""
    global _prof, _prof_only, _prof_every, _prof_seen
    _prof = []
    _prof_only = None if only is None else ({only} if isinstance(only, str) else set(only))
    _prof_every, _prof_seen = max(1, int(every)), 0


def profile_end():
    """-> {tag: (launches, total_ms, total_work)} after synchronising."""
    global _prof, _prof_only
    rec, _prof, _prof_only = _prof or [], None, None
    torch.cuda.synchronize()
    out = {}
    for tag, e0, e1, work in rec:
        n, ms, w = out.get(tag, (0, 0.0, 0.0))
        out[tag] = (n + 1, ms + e0.elapsed_time(e1), w + (work or 0.0))
    return out


def _launch(name: str, args):
    """Resolve the launch device from the pointer arguments (all must agree), run the C entry point under that device with its
    current stream.  torch ops guard the device themselves; a drop-in user who did `model.to('cuda:1')` without
    `torch.cuda.set_device(1)` must not get device-0 launches over device-1 pointers (a memory fault or silent peer access)."""
    dev = None
    for a in args:
        if type(a) is _DevPtr:
            if dev is None:
                dev = a.dev
            elif a.dev != dev:
                raise Cm3pHipError(f"{name}: tensor arguments live on different GPUs (cuda:{dev} and cuda:{a.dev})")
    if dev is None:
        dev = torch.cuda.current_device()
    fn = getattr(load(), name)
    if dev == torch.cuda.current_device():
        h = torch.cuda.current_stream(dev).cuda_stream
        _check(fn(*[h if a is _STREAM else a for a in args]), name)
    else:
        with torch.cuda.device(dev):
            h = torch.cuda.current_stream(dev).cuda_stream
            _check(fn(*[h if a is _STREAM else a for a in args]), name)


def call(name: str, *args, tag: str | None = None, work: float | None = None):
    if _prof is None or (_prof_only is not None and (tag or name) not in _prof_only):
        _launch(name, args)
        return
    if _prof_every > 1:
        global _prof_seen
        _prof_seen += 1
        if (_prof_seen - 1) % _prof_every:
            _launch(name, args)
            return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    _launch(name, args)
    e1.record()
    _prof.append((tag or name, e0, e1, work))


def query(name: str, *args) -> int:
    """Host-only helpers that return a size, not an error code."""
    return getattr(load(), name)(*args)
