"""Typed Python wrappers over the C ABI: allocate outputs (torch = device memory only), launch, return tensors.

No autograd here and no torch compute ops: everything numerical happens inside libcm3p_hip.so.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib
from ._lib import BF16, EPI_BF16, EPI_F32, EPI_F32_RESID, F32, call, dt, ptr, query, stream

Tensor = torch.Tensor

# C-ABI calls whose `work` is algorithmic BYTES (HBM-bound row-wise kernels); every other `work` is matmul FLOPs
_lib.HBM_BOUND_TAGS.update({"cm3p_layernorm_fwd", "cm3p_layernorm_bwd", "cm3p_geglu_fwd", "cm3p_geglu_bwd", "attn_bwd_prep_kernel [global]",
                            "attn_bwd_dq_reduce_kernel [global]", "attn_bwd_prep_kernel [global, varlen]", "attn_bwd_dq_reduce_kernel [global, varlen]"})


def _empty(shape, dtype, like: Tensor) -> Tensor:
    return torch.empty(shape, dtype=dtype, device=like.device)


# ------------------------------------------------------------------------------------------------ norms
def layernorm_fwd(x: Tensor, weight: Tensor, eps: float, want_f32: bool, want_bf16: bool, want_stats: bool = True):
    """x [rows, H] fp32|bf16 -> (y_f32|None, y_bf16|None, mean|None, rstd|None)."""
    rows, H = x.shape
    y32 = _empty((rows, H), torch.float32, x) if want_f32 else None
    y16 = _empty((rows, H), torch.bfloat16, x) if want_bf16 else None
    mean = _empty((rows,), torch.float32, x) if want_stats else None
    rstd = _empty((rows,), torch.float32, x) if want_stats else None
    call("cm3p_layernorm_fwd", ptr(x), dt(x), ptr(weight, torch.float32), ptr(y32), ptr(y16), ptr(mean, torch.float32), ptr(rstd, torch.float32), rows, H, eps, stream(),
         work=float(rows) * H * (x.element_size() + (4 if want_f32 else 0) + (2 if want_bf16 else 0)))
    return y32, y16, mean, rstd


def layernorm_bwd(dy: Tensor, x: Tensor, weight: Tensor, mean: Tensor, rstd: Tensor, dres: Optional[Tensor],
                  want_bf16: bool, inplace: bool = True):
    """-> (dx_f32 = dres + LN'(dy), dx_bf16|None, dw[H]).  With `inplace` dx_f32 overwrites dres."""
    rows, H = x.shape
    dx32 = dres if (inplace and dres is not None) else _empty((rows, H), torch.float32, x)
    dx16 = _empty((rows, H), torch.bfloat16, x) if want_bf16 else None
    nblk = query("cm3p_layernorm_bwd_blocks", rows)
    part = _empty((nblk, H), torch.float32, x)
    dw = _empty((H,), torch.float32, x)
    call("cm3p_layernorm_bwd", ptr(dy), dt(dy), ptr(x), ptr(weight, torch.float32), ptr(mean, torch.float32), ptr(rstd, torch.float32), ptr(dres), ptr(dx32), ptr(dx16),
         ptr(part), ptr(dw), rows, H, stream(),
         work=float(rows) * H * (dy.element_size() + 4 + (4 if dres is not None else 0) + 4 + (2 if want_bf16 else 0)))
    return dx32, dx16, dw


def embed_ln_fwd(ids: Tensor, table: Tensor, weight: Tensor, eps: float, slot: Optional[Tensor] = None,
                 override: Optional[Tensor] = None, want_bf16: bool = False):
    T = ids.numel()
    H = table.shape[1]
    y32 = _empty((T, H), torch.float32, table)
    y16 = _empty((T, H), torch.bfloat16, table) if want_bf16 else None
    mean = _empty((T,), torch.float32, table)
    rstd = _empty((T,), torch.float32, table)
    call("cm3p_embed_ln_fwd", ptr(ids, torch.int64), ptr(table), dt(table), ptr(slot, torch.int32), ptr(override), dt(override) if override is not None else F32,
         ptr(weight, torch.float32), ptr(y32), ptr(y16), ptr(mean, torch.float32), ptr(rstd, torch.float32), T, H, eps, table.shape[0], stream())
    return y32, y16, mean, rstd


def embed_ln_bwd(dy: Tensor, ids: Tensor, table: Tensor, weight: Tensor, mean: Tensor, rstd: Tensor, padding_idx: int,
                 slot: Optional[Tensor] = None, override: Optional[Tensor] = None, want_table_grad: bool = True):
    T = ids.numel()
    V, H = table.shape
    if want_table_grad and T > 0 and os.environ.get("CM3P_EMBED_BWD", "sorted") != "atomic":
        return _embed_ln_bwd_sorted(dy, ids, table, weight, mean, rstd, padding_idx, slot, override)
    d_table = torch.zeros((V, H), dtype=torch.float32, device=table.device) if want_table_grad else None
    # zeros, not empty: in unpadded execution placeholder tokens at masked positions are dropped from the packed rows, their
    # rows are never written by the kernel, and zero is their true gradient
    d_ovr = torch.zeros(override.shape, dtype=torch.float32, device=table.device) if override is not None else None
    nblk = query("cm3p_layernorm_bwd_blocks", T)
    part = _empty((nblk, H), torch.float32, table)
    dw = _empty((H,), torch.float32, table)
    call("cm3p_embed_ln_bwd", ptr(dy), ptr(ids, torch.int64), ptr(table), dt(table), ptr(slot, torch.int32), ptr(override),
         dt(override) if override is not None else F32, ptr(weight, torch.float32), ptr(mean, torch.float32), ptr(rstd, torch.float32), ptr(d_table), ptr(d_ovr), ptr(part), ptr(dw),
         T, H, padding_idx, V, stream())
    return d_table, d_ovr, dw


def _embed_ln_bwd_sorted(dy, ids, table, weight, mean, rstd, padding_idx, slot, override):
    """The embedding backward without atomics (cm3p_embed_ln_bwd_sorted): tokens are visited in id order, every sum has a fixed
    order (reproducible bit for bit, which the atomic kernel is not) and tokens that share an id do not serialise on one row.
    The sort and the run numbering are a counting sort in six small launches (cm3p_token_order; no host read)."""
    T = ids.numel()
    V, H = table.shape
    dev = table.device
    chunk = query("cm3p_embed_ln_bwd_sorted_chunk")
    flat = ids.reshape(-1).to(torch.int64).contiguous()
    order, run_of = token_order(flat, V, chunk)
    R = min(T, V + 3 + T // chunk)
    run_rows = _empty((R, H), torch.float32, table)
    run_ids = torch.empty((R,), dtype=torch.int64, device=dev)
    d_table = _empty((V, H), torch.float32, table)
    d_ovr = torch.zeros(override.shape, dtype=torch.float32, device=dev) if override is not None else None
    nblk = (-(-T // chunk) + 3) // 4
    part = _empty((nblk, H), torch.float32, table)
    dw = _empty((H,), torch.float32, table)
    call("cm3p_embed_ln_bwd_sorted", ptr(dy), ptr(flat, torch.int64), ptr(order, torch.int64), ptr(run_of, torch.int32), ptr(table), dt(table),
         ptr(slot, torch.int32), ptr(override), dt(override) if override is not None else F32, ptr(weight, torch.float32),
         ptr(mean, torch.float32), ptr(rstd, torch.float32), ptr(d_table), ptr(d_ovr), ptr(run_rows), ptr(run_ids, torch.int64),
         ptr(part), ptr(dw), T, H, padding_idx, V, stream())
    return d_table, d_ovr, dw


def token_order(flat_ids: Tensor, V: int, chunk: int):
    """-> (order [T] int64: token indices by ascending clamp(id, -1, V), ties in token order; run_of [T] int32: run number of every
    sorted position, a new run where the id changes and at every multiple of `chunk`).  cm3p_token_order (a counting sort on the
    device); vocabularies beyond its histogram (V + 2 > 12288) and CM3P_TOKEN_ORDER=torch take the torch.sort / cumsum route that
    defines the result."""
    T = flat_ids.numel()
    dev = flat_ids.device
    n_ws = query("cm3p_token_order_workspace_ints", T, V) if os.environ.get("CM3P_TOKEN_ORDER", "hip") != "torch" else 0
    if n_ws > 0:
        order = torch.empty((T,), dtype=torch.int64, device=dev)
        run_of = torch.empty((T,), dtype=torch.int32, device=dev)
        ws = torch.empty((n_ws,), dtype=torch.int32, device=dev)
        call("cm3p_token_order", ptr(flat_ids, torch.int64), T, V, ptr(order, torch.int64), ptr(run_of, torch.int32), ptr(ws, torch.int32), stream())
        return order, run_of
    key = flat_ids.clamp(-1, V)  # ids outside the table get no gradient: two keys for all of them bound the number of runs
    sk, order = torch.sort(key, stable=True)
    start = torch.ones(T, dtype=torch.bool, device=dev)
    if T > 1:
        start[1:] = sk[1:] != sk[:-1]
    start |= (torch.arange(T, device=dev) % chunk) == 0
    return order, (torch.cumsum(start, 0) - 1).to(torch.int32)


def audio_slots(ids: Tensor, audio_token_id: int):
    T = ids.numel()
    slot = torch.empty((T,), dtype=torch.int32, device=ids.device)
    count = torch.empty((1,), dtype=torch.int32, device=ids.device)
    call("cm3p_audio_slots", ptr(ids, torch.int64), T, audio_token_id, ptr(slot, torch.int32), ptr(count), stream())
    return slot, count


# ------------------------------------------------------------------------------------------------ GEMM
def _wgrad_splits(M: int, N: int, K: int) -> int:
    return query("cm3p_gemm_wgrad_splits", M, N, K)


def _big_gemm_kernel(M: int, N: int) -> str:
    """Which 256 x 256 kernel cm3p_gemm_bf16 takes for a big shape (csrc/gemm.hip big_gemm): the half-tile-ring kernel of gemm8p.hip
    unless the extents are not multiples of 8 or CM3P_GEMM_IMPL=256 selects the r01 kernel."""
    return "gemm8p_kernel" if (M % 8 == 0 and N % 8 == 0 and not os.environ.get("CM3P_GEMM_IMPL", "").startswith("2")) else "gemm256_kernel"


def _g8p_rebal(K: int, kchunk: int) -> bool:
    """gemm8p.hip's instance choice (cm3p_gemm8p_dispatch): REBAL when every work item has an even number of 64-deep k-tiles."""
    if os.environ.get("CM3P_G8P_REBAL", "1").startswith("0"):
        return False
    splits = max(1, -(-K // kchunk))
    last = K - (splits - 1) * kchunk
    return (kchunk // 64) % 2 == 0 and ((K // 64) % 2 == 0 if splits == 1 else (last > 0 and (last // 64) % 2 == 0))


def _gemm_tag(M: int, N: int, K: int, a_kc, b_kc, epilogue, split_k: int) -> str:
    """Profiler tag = the kernel the library will pick (same rule as cm3p_gemm_bf16 in csrc/gemm.hip), spelled like rocprof."""
    q = 128 if K % 128 == 0 else 64  # (csrc/gemm.hip: k-split lengths are multiples of 128 where K allows it)
    kchunk = K if split_k <= 1 else -(-(-(-K // split_k)) // q) * q
    big = K % 64 == 0 and kchunk % 64 == 0 and (-(-M // 256)) * (-(-N // 256)) * max(1, -(-K // kchunk)) >= 200
    b2s = lambda v: "true" if v else "false"
    if not big:
        return f"gemm_bf16_kernel<{b2s(a_kc)}, {b2s(b_kc)}, {epilogue}>"
    name = _big_gemm_kernel(M, N)
    tail = f", {b2s(_g8p_rebal(K, kchunk))}" if name == "gemm8p_kernel" else ""
    return f"{name}<{b2s(a_kc)}, {b2s(b_kc)}, {epilogue}{tail}>"


def gemm(a: Tensor, b: Tensor, M: int, N: int, K: int, a_kc: bool, b_kc: bool, epilogue: int,
         resid: Optional[Tensor] = None, out: Optional[Tensor] = None, split_k: int = 1) -> Tensor:
    """C[m,n] = sum_k A(m,k) B(n,k) (+resid).  a/b bf16, row-major 2-D; see include/cm3p_hip.h for the layouts."""
    lda, ldb = a.shape[1], b.shape[1]
    if out is None:
        out = _empty((M, N), torch.bfloat16 if epilogue == EPI_BF16 else torch.float32, a)
    ws = _empty((split_k, M, N), torch.float32, a) if split_k > 1 else None
    call("cm3p_gemm_bf16", ptr(a), ptr(b), ptr(out), ptr(resid), M, N, K, lda, ldb, N, int(a_kc), int(b_kc), epilogue, split_k,
         ptr(ws), stream(), tag=_gemm_tag(M, N, K, a_kc, b_kc, epilogue, split_k), work=2.0 * M * N * K)
    return out


def linear_fwd(x: Tensor, w: Tensor, resid: Optional[Tensor] = None) -> Tensor:
    """x [T,K] bf16, w [N,K] bf16 -> x w^T as bf16, or fp32 resid + x w^T."""
    T, K = x.shape
    N = w.shape[0]
    return gemm(x, w, T, N, K, True, True, EPI_F32_RESID if resid is not None else EPI_BF16, resid)


SOFTMAX_Q_SCALE = 64 ** -0.5 * 1.4426950408889634  # scale * log2(e) for head_dim 64: what q_prescaled attention expects in q


def qkv_linear_rope(x: Tensor, w: Tensor, cos: Tensor, sin: Tensor, S: int, per_batch: bool, q_scale: float = 1.0) -> Tensor:
    """Fused Wqkv projection + RoPE on the q and k thirds: x [T,H] bf16, w [3H,H] bf16 -> qkv [T,3H] bf16 (rotated).
    q_scale multiplies the rotated q third in fp32 before its bf16 rounding (SOFTMAX_Q_SCALE for `prescaled` attention)."""
    T, Kd = x.shape
    N = w.shape[0]
    out = _empty((T, N), torch.bfloat16, x)
    call("cm3p_qkv_gemm_rope", ptr(x), ptr(w), ptr(out), T, N, Kd, ptr(cos, torch.float32), ptr(sin, torch.float32), S, int(per_batch), 2 * N // 3, float(q_scale), stream(),
         tag=_gemm_tag(T, N, Kd, True, True, 3, 1) if (2 * N // 3) % 256 == 0 else "gemm_bf16_kernel<true, true, 3>",
         work=2.0 * T * N * Kd)
    return out


def linear_dgrad(dy: Tensor, w: Tensor, w_t: Optional[Tensor] = None) -> Tensor:
    """dy [T,N] bf16, w [N,K] bf16 -> dx [T,K] bf16 = dy w.  With w_t = w^T [K,N] (cast_bf16_with_transpose) the contraction index
    is contiguous in both operands - the faster form of the 256 x 256 kernel (same products, same accumulation order)."""
    T, N = dy.shape
    K = w.shape[1]
    if w_t is not None:
        return gemm(dy, w_t, T, K, N, True, True, EPI_BF16)
    return gemm(dy, w, T, K, N, True, False, EPI_BF16)


def linear_wgrad(dy: Tensor, x: Tensor) -> Tensor:
    """dy [T,N] bf16, x [T,K] bf16 -> dW [N,K] fp32 = dy^T x (split-K over tokens, deterministic combine)."""
    T, N = dy.shape
    K = x.shape[1]
    return gemm(dy, x, N, K, T, False, False, EPI_F32, split_k=_wgrad_splits(N, K, T))


def cast_bf16(x: Tensor) -> Tensor:
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    call("cm3p_cast_f32_bf16", ptr(x, torch.float32), ptr(y), x.numel(), stream())
    return y


def cast_bf16_with_transpose(x: Tensor):
    """fp32 [rows, cols] -> (bf16 [rows, cols], bf16 [cols, rows]) in one pass."""
    rows, cols = x.shape
    y = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    yt = torch.empty((cols, rows), dtype=torch.bfloat16, device=x.device)
    call("cm3p_cast_f32_bf16_t", ptr(x, torch.float32), ptr(y), ptr(yt), rows, cols, stream())
    return y, yt


def cast_bf16_with_transpose_many(ws: list) -> list:
    """[fp32 2-D weights, extents multiples of 8] -> [(bf16 copy, bf16 transpose)] in ONE launch (cm3p_cast_f32_bf16_t_multi): the values
    of cast_bf16_with_transpose per matrix.  One bf16 allocation holds all copies; the address table travels with one small H2D copy."""
    if not ws:
        return []
    dev = ws[0].device
    numel = sum(w.numel() for w in ws)
    store = torch.empty((2 * numel,), dtype=torch.bfloat16, device=dev)
    base, out, table, off, blocks = store.data_ptr(), [], [], 0, 0
    for w in ws:
        rows, cols = w.shape
        y = store[off:off + rows * cols].view(rows, cols)
        yt = store[off + rows * cols:off + 2 * rows * cols].view(cols, rows)
        table += [w.data_ptr(), base + 2 * off, base + 2 * (off + rows * cols), rows, cols, blocks]
        out.append((y, yt))
        off += 2 * rows * cols
        blocks += (-(-rows // 64)) * (-(-cols // 64))
    # The table goes through PINNED host memory: torch's caching host allocator keeps a pinned block out of circulation until the copy
    # that reads it has run, so the asynchronous upload cannot race the temporary's release (r04 advisor: from a pageable temporary that
    # is freed on return, HIP is free to pin and copy lazily - the kernel would then dereference garbage addresses).  A blocking copy would
    # be safe too but makes the host wait for the stream, and the host runs two to three steps ahead of the GPU (tools/host_lead.py).
    tab = torch.tensor(table, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
    call("cm3p_cast_f32_bf16_t_multi", ptr(tab, torch.int64), len(ws), blocks, stream(), work=8.0 * numel)
    return out


def add_f32(a: Tensor, b: Tensor, want_bf16: bool = False, inplace: bool = True):
    y32 = a if inplace else torch.empty_like(a)
    y16 = torch.empty(a.shape, dtype=torch.bfloat16, device=a.device) if want_bf16 else None
    call("cm3p_add_f32", ptr(a), ptr(b), dt(b), ptr(y32), ptr(y16), a.numel(), stream())
    return y32, y16


# ------------------------------------------------------------------------------------------------ RoPE / attention
def rope_table(position_ids: Tensor, inv_freq: Tensor):
    n = position_ids.numel()
    half = inv_freq.numel()
    cos = torch.empty((n, half), dtype=torch.float32, device=inv_freq.device)
    sin = torch.empty((n, half), dtype=torch.float32, device=inv_freq.device)
    call("cm3p_rope_table", ptr(position_ids, torch.int64), n, ptr(inv_freq, torch.float32), half, ptr(cos, torch.float32), ptr(sin, torch.float32), stream())
    return cos, sin


def rope_apply_(qkv: Tensor, cos: Tensor, sin: Tensor, B: int, S: int, nh: int, per_batch: bool, inverse: bool = False):
    call("cm3p_rope_apply", ptr(qkv), ptr(cos, torch.float32), ptr(sin, torch.float32), B, S, nh, S if per_batch else 0, int(inverse), stream())
    return qkv


def gemm8p_set_grid(workgroups: int) -> None:
    """Workgroups per launch of the big-shape GEMM kernel (0: one per CU, the default; > CU count: the surplus starts as CUs come free -
    what bench.py selects next to RCCL).  Process-wide; include/cm3p_hip.h: cm3p_gemm8p_set_grid."""
    rc = _lib.load().cm3p_gemm8p_set_grid(int(workgroups))
    if rc != 0:
        raise ValueError(f"cm3p_gemm8p_set_grid({workgroups}) -> {rc}")


def gemm8p_get_grid() -> int:
    return int(_lib.load().cm3p_gemm8p_get_grid())


def _attn_fwd_name(window: int, prescaled: bool, masked: bool, S: int, nh: int) -> str:
    """The forward kernel a call lands on, as rocprofv3 names it.  The routing is the library's (cm3p_attn_fwd_impl: global layers with
    pre-scaled q run the pipelined kernel of csrc/attention_fwd.hip unless CM3P_ATTN_FWD_IMPL=wave3 or the sequence is too long for its
    32-bit row offsets); this function only spells the answer (r05 advisor: it used to re-derive the rule)."""
    if query("cm3p_attn_fwd_impl", S, nh, window, int(prescaled)):
        return "attn_fwd_g_kernel<4, " + ("true>" if masked else "false>")
    return "attn_fwd_kernel<1, %s, " + ("true>" if window >= 0 else "false>")


def attn_fwd(qkv: Tensor, key_mask: Optional[Tensor], B: int, S: int, nh: int, window: int, scale: float, prescaled: bool = False):
    """prescaled: the q third already carries scale * log2(e) (qkv_linear_rope(..., q_scale=SOFTMAX_Q_SCALE))."""
    out = torch.empty((B * S, nh * 64), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, nh, S), dtype=torch.float32, device=qkv.device)
    keys = S if window < 0 else min(S, 2 * window + 1)
    call("cm3p_attn_fwd", ptr(qkv), ptr(out), ptr(lse, torch.float32), ptr(key_mask, torch.uint8), B, S, nh, window, scale, int(prescaled), stream(),
         tag=_attn_tag(_attn_fwd_name(window, prescaled, key_mask is not None, S, nh), window, prescaled), work=4.0 * B * nh * S * keys * 64)
    return out, lse


# ---- head sizes other than 64 (csrc/attention_generic.hip: plain fp32 kernels so that every reference configuration runs)
def attn_generic_supported(head_dim: int) -> bool:
    return bool(query("cm3p_attn_generic_supported", int(head_dim)))


def rope_apply_generic_(qkv: Tensor, cos: Tensor, sin: Tensor, B: int, S: int, nh: int, hd: int, per_batch: bool, inverse: bool = False):
    call("cm3p_rope_apply_generic", ptr(qkv, torch.bfloat16), ptr(cos, torch.float32), ptr(sin, torch.float32), B, S, nh, hd, S if per_batch else 0, int(inverse), stream())
    return qkv


def attn_fwd_generic(qkv: Tensor, key_mask: Optional[Tensor], B: int, S: int, nh: int, hd: int, window: int, scale: float):
    out = torch.empty((B * S, nh * hd), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, nh, S), dtype=torch.float32, device=qkv.device)
    call("cm3p_attn_fwd_generic", ptr(qkv, torch.bfloat16), ptr(out), ptr(lse, torch.float32), ptr(key_mask, torch.uint8), B, S, nh, hd, window, scale, stream())
    return out, lse


def attn_bwd_generic(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, key_mask: Optional[Tensor], B: int, S: int, nh: int, hd: int,
                     window: int, scale: float) -> Tensor:
    """-> dqkv with the q / k thirds still in the ROTATED frame (rope_apply_generic_(..., inverse=True) finishes them)."""
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    call("cm3p_attn_bwd_generic", ptr(qkv, torch.bfloat16), ptr(out, torch.bfloat16), ptr(dout, torch.bfloat16), ptr(lse, torch.float32), ptr(delta), ptr(dqkv),
         ptr(key_mask, torch.uint8), B, S, nh, hd, window, scale, stream())
    return dqkv


def attn_probs(qkv: Tensor, lse: Tensor, key_mask: Optional[Tensor], B: int, S: int, nh: int, window: int, scale: float, prescaled: bool = False) -> Tensor:
    """output_attentions: the probabilities [B, nh, S, S] fp32 of one layer, from qkv and the lse attn_fwd stored (inspection path)."""
    probs = torch.empty((B, nh, S, S), dtype=torch.float32, device=qkv.device)
    call("cm3p_attn_probs", ptr(qkv), ptr(lse, torch.float32), ptr(key_mask, torch.uint8), ptr(probs), B, S, nh, window, scale, int(prescaled), stream())
    return probs


ATTN_BWD_DQ, ATTN_BWD_DKV = 1, 2  # stages of cm3p_attn_bwd (include/cm3p_hip.h)


def _attn_tag(fmt: str, window: int, prescaled: bool, varlen: bool = False) -> str:
    """Profiler tag = the kernel's name as rocprofv3 prints it (the template argument is the q_prescaled mode) + which layers it
    served: "attn_bwd_dkv3_kernel<true> [global]".  bench.py matches the part before " [" against the rocprof / PMC rows."""
    name = fmt % ("true" if prescaled else "false") if "%s" in fmt else fmt
    return f"{name} [{'global' if window < 0 else 'local'}{', varlen' if varlen else ''}]"


ATTN_BWD_FUSED_PREP, ATTN_BWD_FUSED_MAIN, ATTN_BWD_FUSED_REDUCE = 1, 2, 4  # stages of cm3p_attn_bwd_fused
ATTN_BWD_FUSED_MAIN_EVEN, ATTN_BWD_FUSED_MAIN_ODD = 8, 16  # the storing launch of _MAIN (first key block of every slab group) and all adding ones
ATTN_BWD_FUSED_MAIN_ADD1 = 32  # the adding launches one by one: _ADD1 << (p - 1) = position p of the group
_fused_ws: dict = {}  # (device index, stream) -> byte tensor: the fused backward's workspace, grown on demand, shared by all layers


_other_caches: list = []  # dicts other modules keep device memory in (encoder.py: the forward-only bf16 weight copies)


def release_workspaces() -> None:
    """Drop the cached workspaces of the fused attention backward (1.65 GB at C2, 3.3 GB at C4 per device and stream) and the bf16
    weight copies kept for forward-only calls: call it between jobs of different sequence lengths if the memory matters; the next
    call allocates what it needs."""
    _fused_ws.clear()
    for c in _other_caches:
        c.clear()


def attn_bwd_fused_enabled() -> bool:
    """Global layers run the five-product kernel of csrc/attention_bwd_fused.hip unless CM3P_ATTN_BWD_FUSED=0 (then the
    query-parallel + key-parallel pair of csrc/attention_bwd.hip; same results up to bf16 rounding of partial sums)."""
    return os.environ.get("CM3P_ATTN_BWD_FUSED", "1") != "0"


def _attn_bwd_fused(qkv, out, dout, lse, key_mask, cu, B, S, total, nh, scale, rope, per_batch, prescaled) -> Tensor:
    dqkv = torch.empty_like(qkv)
    cos, sin = rope if rope is not None else (None, None)
    need = query("cm3p_attn_bwd_fused_workspace_bytes", B, S, nh)
    key = (qkv.device.index, torch.cuda.current_stream(qkv.device).cuda_stream)
    ws = _fused_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _fused_ws[key] = torch.empty(need, dtype=torch.uint8, device=qkv.device)
    varlen = cu is not None
    rows = total if varlen else B * S
    fl = 2.0 * B * nh * S * S * 64  # one S x S x 64 product per (batch, head); SURVEY.md 8(d) credits four to the backward
    nkb = -(-S // 256)  # 256-key blocks; G of them share a dQ slab: the first of each group stores (one launch), the others add (G - 1 launches)
    G = query("cm3p_attn_bwd_fused_slab_group", S)  # (the C side derives it, CM3P_FUSED_SLAB_GROUP included: one parse, one answer)
    n_store = -(-nkb // G)
    slabs = float(need) * 2.0 / G  # (the workspace is sized for groups of 2)
    # one C call per launch, so that every profiler tag is ONE kernel (one rocprof row): the storing launch, then the adding launches
    adds = [(ATTN_BWD_FUSED_MAIN_ADD1 << (p - 1), "attn_bwd_fused_kernel<%s, true>", 4.0 * fl * len(range(p, nkb, G)) / nkb)
            for p in range(1, min(G, nkb))]
    for stage, name, work in (
        (ATTN_BWD_FUSED_PREP, "attn_bwd_prep_kernel", 2.0 * rows * nh * 64 * 2 + 12.0 * rows * nh),
        (ATTN_BWD_FUSED_MAIN_EVEN, "attn_bwd_fused_kernel<%s, false>", 4.0 * fl * n_store / nkb),
        *adds,
        (ATTN_BWD_FUSED_REDUCE, "attn_bwd_dq_reduce_kernel", slabs + rows * nh * 64 * 2.0),
    ):
        call("cm3p_attn_bwd_fused", ptr(qkv), ptr(out), ptr(dout), ptr(lse, torch.float32), ptr(dqkv), ptr(key_mask, torch.uint8),
             ptr(cu, torch.int32), B, S, total if varlen else 0, nh, scale, ptr(cos, torch.float32), ptr(sin, torch.float32),
             S if (per_batch and not varlen) else 0, stage, int(prescaled), ptr(ws), ws.numel(), stream(),
             tag=_attn_tag(name, -1, prescaled, varlen), work=None if varlen else work)  # (packed rows: S is only the longest sequence)
    return dqkv


def attn_bwd(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, key_mask: Optional[Tensor], B: int, S: int, nh: int,
             window: int, scale: float, rope: Optional[tuple] = None, per_batch: bool = False, prescaled: bool = False) -> Tensor:
    """rope = (cos, sin): also applies the inverse rotary rotation to dq / dk (backward of the fused Wqkv+RoPE GEMM).
    The two kernels are issued as two C calls so that each has its own profiler tag (one rocprof row per tag).  `work` is the
    algorithmic count of SURVEY.md section 8(d) (backward = 2 x forward = four matmuls: dQ is the dq kernel's, dP / dV / dK the
    dkv kernel's); the scores each kernel recomputes are not credited."""
    if window < 0 and attn_bwd_fused_enabled():
        return _attn_bwd_fused(qkv, out, dout, lse, key_mask, None, B, S, 0, nh, scale, rope, per_batch, prescaled)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    keys = S if window < 0 else min(S, 2 * window + 1)
    cos, sin = rope if rope is not None else (None, None)
    # global layers: the hand-scheduled kernels of csrc/attention_bwd.hip; sliding-window layers: the band kernels of attention.hip
    names = ("attn_bwd_dq3_kernel", "attn_bwd_dkv3_kernel<%s>") if window < 0 else ("attn_bwd_dq_kernel<%s>", "attn_bwd_dkv_kernel<%s>")
    for stage, name, products in ((ATTN_BWD_DQ, names[0], 1), (ATTN_BWD_DKV, names[1], 3)):
        call("cm3p_attn_bwd", ptr(qkv), ptr(out), ptr(dout), ptr(lse, torch.float32), ptr(delta), ptr(dqkv), ptr(key_mask, torch.uint8), B, S, nh,
             window, scale, ptr(cos, torch.float32), ptr(sin, torch.float32), S if per_batch else 0, stage, int(prescaled), stream(), tag=_attn_tag(name, window, prescaled),
             work=2.0 * products * B * nh * S * keys * 64)
    return dqkv


def attn_fwd_varlen(qkv: Tensor, cu: Tensor, B: int, max_s: int, nh: int, window: int, scale: float, prescaled: bool = False):
    """Packed sequences: qkv [total, 3, nh, 64], cu int32 [B+1] -> out [total, nh*64], lse [nh, total]."""
    total = qkv.shape[0]
    out = torch.empty((total, nh * 64), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((nh, total), dtype=torch.float32, device=qkv.device)
    call("cm3p_attn_fwd_varlen", ptr(qkv), ptr(out), ptr(lse, torch.float32), ptr(cu, torch.int32), B, max_s, total, nh, window, scale, int(prescaled), stream(),
         tag=_attn_tag(_attn_fwd_name(window, prescaled, False, max_s, nh), window, prescaled, True))
    return out, lse


def attn_bwd_varlen(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, cu: Tensor, B: int, max_s: int, nh: int, window: int,
                    scale: float, rope: Optional[tuple] = None, prescaled: bool = False) -> Tensor:
    """rope = (cos, sin) per packed token [total, 32]: also applies the inverse rotation to dq / dk."""
    if window < 0 and attn_bwd_fused_enabled():
        return _attn_bwd_fused(qkv, out, dout, lse, None, cu, B, max_s, qkv.shape[0], nh, scale, rope, False, prescaled)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    cos, sin = rope if rope is not None else (None, None)
    names = ("attn_bwd_dq3_kernel", "attn_bwd_dkv3_kernel<%s>") if window < 0 else ("attn_bwd_dq_kernel<%s>", "attn_bwd_dkv_kernel<%s>")
    for stage, name in ((ATTN_BWD_DQ, names[0]), (ATTN_BWD_DKV, names[1])):
        call("cm3p_attn_bwd_varlen", ptr(qkv), ptr(out), ptr(dout), ptr(lse, torch.float32), ptr(delta), ptr(dqkv), ptr(cu, torch.int32), B, max_s,
             qkv.shape[0], nh, window, scale, ptr(cos, torch.float32), ptr(sin, torch.float32), stage, int(prescaled), stream(), tag=_attn_tag(name, window, prescaled, True))
    return dqkv


def gather_rows(src: Tensor, idx: Tensor) -> Tensor:
    """src [R, H] fp32, idx int64 [n] -> [n, H] (the row selection of _unpad_cm3p_input)."""
    out = torch.empty((idx.numel(), src.shape[1]), dtype=torch.float32, device=src.device)
    call("cm3p_gather_rows_f32", ptr(src), ptr(idx, torch.int64), ptr(out), idx.numel(), src.shape[1], stream())
    return out


def scatter_rows(src: Tensor, idx: Tensor, rows: int) -> Tensor:
    """src [n, H] fp32 -> [rows, H] with row idx[i] = src[i] and zeros elsewhere (_pad_cm3p_output)."""
    out = torch.zeros((rows, src.shape[1]), dtype=torch.float32, device=src.device)
    call("cm3p_scatter_rows_f32", ptr(src), ptr(idx, torch.int64), ptr(out), idx.numel(), src.shape[1], stream())
    return out


# ------------------------------------------------------------------------------------------------ MLP pieces
def geglu_fwd(h: Tensor) -> Tensor:
    T, I2 = h.shape
    g = torch.empty((T, I2 // 2), dtype=torch.bfloat16, device=h.device)
    call("cm3p_geglu_fwd", ptr(h), ptr(g), T, I2 // 2, stream(), work=6.0 * T * (I2 // 2))
    return g


def gemm_geglu_supported(T: int, I: int, Kd: int) -> bool:
    """cm3p_gemm_geglu's shape rule (csrc/gemm.hip): the 256 x 256 ring kernel's big shapes."""
    return Kd % 64 == 0 and I % 32 == 0 and T % 8 == 0 and (-(-T // 256)) * (-(-2 * I // 256)) >= 200 \
        and not os.environ.get("CM3P_GEMM_IMPL", "").startswith("2")


def geglu_interleave_index(I: int, device) -> Tensor:
    """Row order of the interleaved Wi copy cm3p_gemm_geglu reads: row 64 q + r <- Wi row 32 q + r (r < 32) or I + 32 q + r - 32."""
    n = torch.arange(2 * I, device=device)
    r = n % 64
    j = (n // 64) * 32 + r % 32
    return torch.where(r < 32, j, I + j)


def gemm_geglu(x: Tensor, w_interleaved: Tensor) -> Tensor:
    """x [T,K] bf16, interleaved Wi [2I,K] bf16 -> gelu_erf(x Wi[:I]^T) * (x Wi[I:]^T) as bf16 [T,I] in one kernel (forward-only)."""
    T, Kd = x.shape
    I = w_interleaved.shape[0] // 2
    a = _empty((T, I), torch.bfloat16, x)
    call("cm3p_gemm_geglu", ptr(x), ptr(w_interleaved), ptr(a), T, I, Kd, stream(),
         tag=f"gemm8p_kernel<true, true, 6, {'true' if _g8p_rebal(Kd, Kd) else 'false'}>", work=2.0 * T * 2 * I * Kd)
    return a


def geglu_bwd(dg: Tensor, h: Tensor) -> Tensor:
    dh = torch.empty_like(h)
    call("cm3p_geglu_bwd", ptr(dg), ptr(h), ptr(dh), h.shape[0], h.shape[1] // 2, stream(), work=10.0 * h.shape[0] * (h.shape[1] // 2))
    return dh


def gelu_fwd(x: Tensor) -> Tensor:
    y = torch.empty_like(x)
    call("cm3p_gelu_fwd", ptr(x), ptr(y), x.numel(), stream())
    return y


def gelu_bwd(dy: Tensor, x: Tensor) -> Tensor:
    dx = torch.empty_like(x)
    call("cm3p_gelu_bwd", ptr(dy), ptr(x), ptr(dx), x.numel(), stream())
    return dx


# ------------------------------------------------------------------------------------------------ audio front end
def im2col_k3(x: Tensor, token_major: bool, B: int, C: int, T_in: int, stride: int):
    T_out = (T_in - 1) // stride + 1
    p = torch.empty((B * T_out, C * 3), dtype=torch.bfloat16, device=x.device)
    call("cm3p_im2col_k3", ptr(x, torch.bfloat16 if token_major else torch.float32), int(token_major), ptr(p), B, C, T_in, T_out, stride, stream())
    return p, T_out


def col2im_k3(dp: Tensor, B: int, C: int, T_in: int, T_out: int, stride: int) -> Tensor:
    dx = torch.empty((B, T_in, C), dtype=torch.bfloat16, device=dp.device)
    call("cm3p_col2im_k3", ptr(dp), ptr(dx), B, C, T_in, T_out, stride, stream())
    return dx


def bias_gelu_fwd(z: Tensor, bias: Tensor, want_bf16: bool, want_f32: bool):
    R, C = z.shape
    a16 = torch.empty((R, C), dtype=torch.bfloat16, device=z.device) if want_bf16 else None
    a32 = torch.empty((R, C), dtype=torch.float32, device=z.device) if want_f32 else None
    call("cm3p_bias_gelu_fwd", ptr(z), ptr(bias), ptr(a16), ptr(a32), R, C, stream())
    return a16, a32


def bias_gelu_bwd(da: Tensor, z: Tensor, bias: Tensor):
    R, C = z.shape
    dz = torch.empty((R, C), dtype=torch.bfloat16, device=z.device)
    part = torch.empty((query("cm3p_bias_gelu_bwd_blocks", R), C), dtype=torch.float32, device=z.device)
    db = torch.empty((C,), dtype=torch.float32, device=z.device)
    call("cm3p_bias_gelu_bwd", ptr(da), dt(da), ptr(z), ptr(bias), ptr(dz), ptr(part), ptr(db), R, C, stream())
    return dz, db


# ------------------------------------------------------------------------------------------------ pooling
def pool_fwd(h: Tensor, mask: Optional[Tensor], Bn: int, S: int, cls: bool):
    H = h.shape[-1]
    pooled = torch.empty((Bn, H), dtype=torch.float32, device=h.device)
    count = torch.empty((Bn,), dtype=torch.float32, device=h.device)
    part = None if cls else torch.empty((Bn, query("cm3p_pool_chunks", S), H), dtype=torch.float32, device=h.device)
    call("cm3p_pool_fwd", ptr(h), ptr(mask, torch.int64), ptr(pooled), ptr(part), ptr(count), Bn, S, H, int(cls), stream())
    return pooled, count


def pool_bwd(dpooled: Tensor, mask: Optional[Tensor], count: Tensor, Bn: int, S: int, cls: bool) -> Tensor:
    H = dpooled.shape[-1]
    dh = torch.empty((Bn * S, H), dtype=torch.float32, device=dpooled.device)
    call("cm3p_pool_bwd", ptr(dpooled), ptr(mask, torch.int64), ptr(count), ptr(dh), Bn, S, H, int(cls), stream())
    return dh


# ------------------------------------------------------------------------------------------------ fp32 head
def gemm_f32(a: Tensor, b: Tensor, M: int, N: int, K: int, a_strides, b_strides, alpha: float = 1.0,
             out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    call("cm3p_gemm_f32", ptr(a), ptr(b), ptr(out), M, N, K, a_strides[0], a_strides[1], b_strides[0], b_strides[1], N, alpha,
         int(accumulate), stream())
    return out


def l2norm_fwd(x: Tensor):
    rows, D = x.shape
    y = torch.empty_like(x)
    norm = torch.empty((rows,), dtype=torch.float32, device=x.device)
    call("cm3p_l2norm_fwd", ptr(x), ptr(y), ptr(norm), rows, D, stream())
    return y, norm


def l2norm_bwd(dy: Tensor, y: Tensor, norm: Tensor) -> Tensor:
    dx = torch.empty_like(y)
    call("cm3p_l2norm_bwd", ptr(dy), ptr(y), ptr(norm), ptr(dx), y.shape[0], y.shape[1], stream())
    return dx


def cross_entropy(logits: Tensor, rows: int, cols: int, row_stride: int, col_stride: int, target: Tensor,
                  row_offset: Optional[Tensor], grad_scale: float, dlogits: Optional[Tensor]) -> Tensor:
    loss_rows = torch.empty((rows,), dtype=torch.float32, device=logits.device)
    call("cm3p_cross_entropy", ptr(logits, torch.float32), rows, cols, row_stride, col_stride, ptr(row_offset, torch.int64), ptr(target, torch.int64), grad_scale,
         ptr(loss_rows), ptr(dlogits), stream())
    return loss_rows


def cross_entropy_masked(logits: Tensor, cols: int, target: Tensor, ignore_index: int, grad_scale: float, inv_count: Tensor,
                         want_grad: bool):
    rows, pitch = logits.shape
    loss_rows = torch.empty((rows,), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits) if want_grad else None
    call("cm3p_cross_entropy_masked", ptr(logits, torch.float32), rows, cols, pitch, ptr(target, torch.int64), ignore_index, grad_scale, ptr(inv_count, torch.float32),
         ptr(loss_rows), ptr(dlogits), stream())
    return loss_rows, dlogits


def ce_masked_stats(logits: Tensor, cols: int, target: Tensor, ignore_index: int):
    """-> (loss_rows, lse_rows), zero for ignored rows."""
    rows, pitch = logits.shape
    loss_rows = torch.empty((rows,), dtype=torch.float32, device=logits.device)
    lse_rows = torch.empty((rows,), dtype=torch.float32, device=logits.device)
    call("cm3p_ce_masked_stats", ptr(logits, torch.float32), rows, cols, pitch, ptr(target, torch.int64), ignore_index, ptr(loss_rows), ptr(lse_rows, torch.float32), stream())
    return loss_rows, lse_rows


def ce_masked_dlogits_bf16(logits: Tensor, cols: int, target: Tensor, ignore_index: int, lse_rows: Tensor, scale_a: Tensor, scale_b: Tensor):
    """-> (dlogits bf16 [rows, pitch], column sums fp32 [pitch]) of scale_a * scale_b * (softmax - onehot) on the labelled rows."""
    rows, pitch = logits.shape
    dl = torch.empty((rows, pitch), dtype=torch.bfloat16, device=logits.device)
    part = torch.empty((query("cm3p_ce_masked_dlogits_blocks", rows), pitch), dtype=torch.float32, device=logits.device)
    colsum = torch.empty((pitch,), dtype=torch.float32, device=logits.device)
    call("cm3p_ce_masked_dlogits_bf16", ptr(logits, torch.float32), rows, cols, pitch, ptr(target, torch.int64), ignore_index, ptr(lse_rows, torch.float32), ptr(scale_a, torch.float32), ptr(scale_b, torch.float32),
         ptr(dl), ptr(part), ptr(colsum), stream())
    return dl, colsum


def inv_valid_count(target: Tensor, ignore_index: int) -> Tensor:
    out = torch.empty((1,), dtype=torch.float32, device=target.device)
    call("cm3p_inv_valid_count", ptr(target, torch.int64), target.numel(), ignore_index, ptr(out), stream())
    return out


def add_bias_(x: Tensor, bias: Tensor) -> Tensor:
    call("cm3p_add_bias_f32", ptr(x), ptr(bias), x.shape[0], x.shape[1], stream())
    return x


def colsum_f32(x: Tensor) -> Tensor:
    rows, cols = x.shape
    part = torch.empty((query("cm3p_colsum_blocks", rows), cols), dtype=torch.float32, device=x.device)
    out = torch.empty((cols,), dtype=torch.float32, device=x.device)
    call("cm3p_colsum_f32", ptr(x), ptr(part), ptr(out), rows, cols, stream())
    return out


def pointwise_loss(x: Tensor, y: Tensor, kind: int):
    """Mean MSE (kind 0) / BCE-with-logits (kind 1) of fp32 x against y -> (loss [1], dloss/dx)."""
    out = torch.empty((1,), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    call("cm3p_pointwise_loss", ptr(x), ptr(y), ptr(out), ptr(dx), x.numel(), kind, stream())
    return out, dx


def first_zero_index(classes: Tensor) -> Tensor:
    B, V = classes.shape
    idx = torch.empty((B,), dtype=torch.int64, device=classes.device)
    call("cm3p_first_zero_index", ptr(classes, torch.int64), B, V, ptr(idx, torch.int64), stream())
    return idx


def scale_exp(x: Tensor, log_scale: Tensor) -> Tensor:
    y = torch.empty_like(x)
    call("cm3p_scale_exp", ptr(x), ptr(log_scale), ptr(y), x.numel(), stream())
    return y


def scale_by(x: Tensor, scale: Tensor) -> Tensor:
    y = torch.empty_like(x)
    call("cm3p_scale_by", ptr(x), ptr(scale), ptr(y), x.numel(), stream())
    return y


def dot_f32(a: Tensor, b: Tensor) -> Tensor:
    out = torch.empty((1,), dtype=torch.float32, device=a.device)
    call("cm3p_dot_f32", ptr(a), ptr(b), ptr(out), a.numel(), stream())
    return out


def sum_f32(x: Tensor, scale: float, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    if out is None:
        out = torch.empty((1,), dtype=torch.float32, device=x.device)
    call("cm3p_sum_f32", ptr(x), ptr(out), x.numel(), scale, int(accumulate), stream())
    return out
