"""The ModernBERT-shaped encoder tower of CM3P on HIP kernels.

`CM3PEncoder` stands where the reference instantiates `transformers.ModernBertModel`
(ref:cm3p/modeling_cm3p.py:305,491,537) and keeps its parameter names, so `state_dict()` keys are unchanged:
    embeddings.tok_embeddings.weight, embeddings.norm.weight, layers.N.{attn_norm,mlp_norm}.weight,
    layers.N.attn.{Wqkv,Wo}.weight, layers.N.mlp.{Wi,Wo}.weight, final_norm.weight
The nn.Linear / nn.LayerNorm / nn.Embedding members are parameter containers only; their torch forward is never
called.  Every encoder layer is ONE autograd node (`_EncoderLayerFn`; `_FinalNormFn` closes the stack) whose forward and
backward are sequences of C-ABI launches (cm3p_amd/kernels.py), restating
TF:models/modernbert/modeling_modernbert.py:262-333,434-478:

    per layer:  xn = LN(x) [identity for layer 0] -> qkv = xn Wqkv^T -> RoPE(q,k) -> flash attention (global, or
                |i-j| <= 64 band; key padding) -> x += o Wo^T -> xn = LN(x) -> h = xn Wi^T -> g = gelu(h[:I]) * h[I:]
                -> x += g Wo^T;   after the last layer: final LN.

Numerics: the residual stream, LayerNorm statistics, softmax and all accumulations are fp32; GEMM operands and saved
activations are bf16 (what HF Trainer's bf16 autocast gives the reference's nn.Linear calls).
"""
from __future__ import annotations

import os
import weakref
from typing import Optional

import torch
from torch import nn

from . import kernels as K

Tensor = torch.Tensor


def _check_supported(cfg):
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    problems = []
    if H % nh or H // nh not in (16, 32, 64):
        # 64: the MFMA kernels every default tower runs on; 16 / 32: csrc/attention_generic.hip (plain fp32 kernels, so that the
        # reference's small test configurations run as well)
        problems.append(f"head_dim must be 64 (or 16 / 32 on the generic kernels); hidden_size={H}, heads={nh}")
    if getattr(cfg, "hidden_activation", "gelu") != "gelu":
        problems.append("hidden_activation must be 'gelu'")
    for flag in ("norm_bias", "attention_bias", "mlp_bias"):
        if getattr(cfg, flag, False):
            problems.append(f"{flag}=True is not supported (the reference configs never set it)")
    for p in ("attention_dropout", "embedding_dropout", "mlp_dropout"):
        if getattr(cfg, p, 0.0) != 0.0:
            problems.append(f"{p} must be 0.0")
    if cfg.intermediate_size % 8 or H % 8:
        problems.append("hidden_size and intermediate_size must be multiples of 8")
    if problems:
        raise NotImplementedError("cm3p_amd HIP encoder: " + "; ".join(problems))


class CM3PAttentionParams(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.Wqkv = nn.Linear(cfg.hidden_size, 3 * cfg.hidden_size, bias=False)
        self.Wo = nn.Linear(cfg.hidden_size, cfg.hidden_size, bias=False)


class CM3PMLPParams(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.Wi = nn.Linear(cfg.hidden_size, 2 * cfg.intermediate_size, bias=False)
        self.Wo = nn.Linear(cfg.intermediate_size, cfg.hidden_size, bias=False)


class CM3PEncoderLayerParams(nn.Module):
    def __init__(self, cfg, layer_idx: int):
        super().__init__()
        # layer 0 has no attn_norm (TF:...modeling_modernbert.py:309-310)
        self.attn_norm = nn.Identity() if layer_idx == 0 else nn.LayerNorm(cfg.hidden_size, eps=cfg.norm_eps, bias=False)
        self.attn = CM3PAttentionParams(cfg)
        self.mlp_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.norm_eps, bias=False)
        self.mlp = CM3PMLPParams(cfg)


class CM3PEmbeddingParams(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.tok_embeddings = nn.Embedding(cfg.vocab_size, cfg.hidden_size, padding_idx=getattr(cfg, "pad_token_id", None))
        self.norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.norm_eps, bias=False)


def _f32(w: Tensor) -> Tensor:
    return w if w.dtype == torch.float32 else w.float()


def _bf16_weight(w: Tensor) -> Tensor:
    """bf16 GEMM operand of a master weight (what autocast's per-call cast produces)."""
    w = w.detach()
    if w.dtype == torch.bfloat16:
        return w
    if w.dtype != torch.float32:  # fp16 / fp64 checkpoints: widen first (dtype plumbing; the cast kernel reads fp32)
        w = w.float()
    return K.cast_bf16(w.contiguous())


_eval_weights: dict = {}  # id(parameter) -> (weakref to it, its _version, its data_ptr, bf16 copy): forward-only calls
K._other_caches.append(_eval_weights)  # (kernels.release_workspaces() empties it)


def invalidate_weight_cache() -> None:
    """Drop the bf16 weight copies kept across forward-only calls.  Called by every forward that will be followed by a backward (a
    training step means an optimizer is about to rewrite the weights, whatever it is and however it writes them) and by
    `module.train()` / `module.eval()` mode flips of an encoder; call it by hand after writing weights in a way that leaves no trace
    on the parameter (`p.data.copy_(...)`, an EMA swap through `.data`, a kernel that takes `p.data_ptr()`) between two
    forward-only calls."""
    _eval_weights.clear()


def _bf16_weight_cached(w: Tensor, geglu_rows: bool = False) -> Tensor:
    """Forward-only calls (no backward will follow: evaluation, embedding extraction) reuse the bf16 copy of a master weight for
    as long as the weight is the same object with the same version counter and storage - every in-place torch op (torch optimizers,
    load_state_dict) bumps the counter, `p.data = ...` changes the address, and writers that go through raw addresses bump it by hand
    (cm3p_amd.Muon.step does).  Writes through `p.data` bump nothing: see invalidate_weight_cache().  A training step re-casts every
    weight once anyway and empties the cache.
    geglu_rows: the copy of a Wi weight with its rows in the order cm3p_gemm_geglu reads (kernels.geglu_interleave_index)."""
    key = (id(w), geglu_rows)
    hit = _eval_weights.get(key)
    if hit is not None and hit[0]() is w and hit[1] == w._version and hit[2] == w.data_ptr():
        return hit[3]
    wb = _bf16_weight(w)
    if geglu_rows:
        wb = wb.index_select(0, K.geglu_interleave_index(w.shape[0] // 2, wb.device)).contiguous()
    if wb.data_ptr() != w.data_ptr():  # (a bf16 master weight is its own operand: nothing to keep)
        if len(_eval_weights) > 4096:  # dead entries of models that are gone
            for k in [k for k, v in _eval_weights.items() if v[0]() is None]:
                del _eval_weights[k]
        _eval_weights[key] = (weakref.ref(w), w._version, w.data_ptr(), wb)
    return wb


def _bf16_weight_pair(w: Tensor, want_t: bool):
    """-> (bf16 W [out, in], bf16 W^T [in, out] or None).  The transpose feeds the input-gradient GEMM (dx = dy W) with the
    contraction index contiguous in both operands; it is made in the same pass as the cast when a backward will follow and the
    extents allow it (multiples of 8), otherwise dgrad reads W itself (k-strided operand, same result)."""
    if not want_t:
        return _bf16_weight_cached(w), None
    w = w.detach()
    if w.dtype == torch.float32 and w.dim() == 2 and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0:
        return K.cast_bf16_with_transpose(w.contiguous())
    return _bf16_weight(w), None


class _Geometry:
    """Static description of one forward call of the stack (no tensors that need grad)."""

    __slots__ = ("B", "S", "H", "I", "nh", "L", "eps", "windows", "key_mask", "rope", "per_batch_pos", "save", "cu", "max_s", "checkpoint",
                 "handoff", "attn_out", "wcast", "hd")


def _hand_upstream(geo: _Geometry, gx32: Tensor, gx16: Optional[Tensor]) -> None:
    """A node's backward leaves the bf16 twin of the fp32 gradient it returns here; the node below (the next one the autograd
    engine runs on this chain) picks it up instead of re-reading 4 bytes per element to cast it again."""
    geo.handoff = (gx32, gx16)


def _take_from_downstream(geo: _Geometry, dy: Tensor):
    """-> (fp32 gradient this node may update in place, its bf16 twin).  The fast path applies when `dy` is exactly the tensor
    the node above handed up (nothing else consumed that activation, so autograd passed it through untouched); the handoff
    keeps that tensor alive, so an equal address cannot be a recycled allocation.  Anything else (a hook, a second consumer)
    gets a private copy and a fresh cast."""
    h, geo.handoff = geo.handoff, None
    if h is not None and h[1] is not None and h[0].data_ptr() == dy.data_ptr() and h[0].shape == dy.shape and dy.dtype == torch.float32 \
            and dy.is_contiguous():
        return h[0], h[1]
    g = dy.float().contiguous()
    if g.data_ptr() == dy.data_ptr():
        g = g.clone()
    return g, K.cast_bf16(g)


def _layer_forward(geo: _Geometry, i: int, x: Tensor, wb, want_stats: bool):
    """One encoder layer on [T, H] rows: -> (x_out, activations needed by its backward)."""
    w_an, Wqkv_b, Wo_b, w_mn, Wi_b, Wo2_b = wb[:6]
    B, S, nh = geo.B, geo.S, geo.nh
    scale = geo.hd ** -0.5
    cos, sin = geo.rope[i]
    if i == 0:
        xn, mean_a, rstd_a = K.cast_bf16(x), None, None
    else:
        _, xn, mean_a, rstd_a = K.layernorm_fwd(x, w_an, geo.eps, False, True, want_stats)
    if geo.hd != 64:
        # head sizes 16 / 32: plain projection, rotary embedding as its own pass (fp32 on the bf16 projection, one rounding: the
        # reference's order), attention on the generic kernels (csrc/attention_generic.hip); padded execution only
        qkv = K.linear_fwd(xn, Wqkv_b)
        K.rope_apply_generic_(qkv, cos, sin, B, S, nh, geo.hd, geo.per_batch_pos)
        o, lse = K.attn_fwd_generic(qkv, geo.key_mask, B, S, nh, geo.hd, geo.windows[i], scale)
    # projection + RoPE in one kernel; the q third also takes the softmax's scale * log2(e) before its one bf16 rounding, so the
    # attention kernels exponentiate the MFMA's scores as they come (prescaled=True everywhere below)
    elif geo.cu is not None:  # unpadded batch: packed rows, per-token rotary tables
        qkv = K.qkv_linear_rope(xn, Wqkv_b, cos, sin, S, geo.per_batch_pos, q_scale=K.SOFTMAX_Q_SCALE)
        o, lse = K.attn_fwd_varlen(qkv, geo.cu, B, geo.max_s, nh, geo.windows[i], scale, prescaled=True)
    else:
        qkv = K.qkv_linear_rope(xn, Wqkv_b, cos, sin, S, geo.per_batch_pos, q_scale=K.SOFTMAX_Q_SCALE)
        o, lse = K.attn_fwd(qkv, geo.key_mask, B, S, nh, geo.windows[i], scale, prescaled=True)
        if geo.attn_out is not None and len(geo.attn_out) == i:  # output_attentions (not again when a checkpointed layer is recomputed)
            geo.attn_out.append(K.attn_probs(qkv, lse, geo.key_mask, B, S, nh, geo.windows[i], scale, prescaled=True))
    x_mid = K.linear_fwd(o, Wo_b, resid=x)
    _, xn2, mean_m, rstd_m = K.layernorm_fwd(x_mid, w_mn, geo.eps, False, True, want_stats)
    if len(wb) > 6 and wb[6] is not None:  # forward-only call: Wi and GeGLU in one kernel, h and g never exist (bit-identical a)
        h, g = None, K.gemm_geglu(xn2, wb[6])
    else:
        h = K.linear_fwd(xn2, Wi_b)
        g = K.geglu_fwd(h)
    x_out = K.linear_fwd(g, Wo2_b, resid=x_mid)
    return x_out, (x, xn, mean_a, rstd_a, qkv, o, lse, x_mid, xn2, mean_m, rstd_m, h, g)


class _EncoderLayerFn(torch.autograd.Function):
    """One ModernBERT encoder layer (TF:...modeling_modernbert.py:318-333): x [T,H] fp32 + its weights -> x_out [T,H] fp32.

    One autograd node per layer, so a layer's weight gradients reach their parameters (and a DistributedDataParallel
    bucket's all-reduce starts) while the layers below are still in backward.  The fp32 residual-stream gradient travels
    through autograd; its bf16 twin - what the next dgrad / wgrad GEMMs read - travels beside it (_hand_upstream)."""

    @staticmethod
    def forward(ctx, geo: _Geometry, i: int, x: Tensor, *weights: Tensor):
        it = iter(weights)
        w_an = None if i == 0 else _f32(next(it))
        Wqkv, Wo, w_mn, Wi, Wo2 = next(it), next(it), _f32(next(it)), next(it), next(it)
        # forward-only and a shape of the ring kernel: the Wi GEMM stores gelu(h) * g itself (CM3P_GEGLU_FUSED=0: the two-kernel path)
        fuse_geglu = (not geo.save) and Wi.dim() == 2 and K.gemm_geglu_supported(x.shape[0], Wi.shape[0] // 2, Wi.shape[1]) \
            and os.environ.get("CM3P_GEGLU_FUSED", "1") != "0"
        # (training: the casts of every layer were made in one launch at the top of the stack, _run_stack)
        pre = geo.wcast if geo.save and geo.wcast is not None else {}
        pairs = [pre.get(id(w)) or _bf16_weight_pair(w, geo.save) if not (fuse_geglu and w is Wi) else (None, None) for w in (Wqkv, Wo, Wi, Wo2)]
        wb = (w_an, pairs[0][0], pairs[1][0], w_mn, pairs[2][0], pairs[3][0], _bf16_weight_cached(Wi, True) if fuse_geglu else None)
        x_out, acts = _layer_forward(geo, i, x, wb, geo.save)
        if geo.save:
            # gradient checkpointing (ref: supports_gradient_checkpointing, TF GradientCheckpointingLayer): keep only the
            # layer input; its activations are recomputed by the same kernels (bit-identical) in the backward pass
            ctx.saved = (x,) if geo.checkpoint else acts
            ctx.geo, ctx.i, ctx.wb = geo, i, wb
            ctx.wt = tuple(p[1] for p in pairs)  # W^T copies for the input-gradient GEMMs
            ctx.wdtypes = [w.dtype for w in weights]
        return x_out

    @staticmethod
    def backward(ctx, dy: Tensor):
        geo, i = ctx.geo, ctx.i
        B, S, nh = geo.B, geo.S, geo.nh
        scale = geo.hd ** -0.5
        need = ctx.needs_input_grad  # (geo, i, x, *weights)
        need_w = list(need[3:])
        if i == 0:
            need_w.insert(0, False)  # (no attn_norm in layer 0)
        n_an, n_qkv, n_o, n_mn, n_i, n_o2 = need_w
        gx32, gx16 = _take_from_downstream(geo, dy)
        if geo.checkpoint:
            _, acts = _layer_forward(geo, i, ctx.saved[0], ctx.wb, True)
        else:
            acts = ctx.saved
        x, xn, mean_a, rstd_a, qkv, o, lse, x_mid, xn2, mean_m, rstd_m, h, g = acts
        del acts
        w_an, Wqkv_b, Wo_b, w_mn, Wi_b, Wo2_b = ctx.wb[:6]
        Wqkv_t, Wo_t, Wi_t, Wo2_t = ctx.wt
        ctx.saved = ctx.wb = ctx.wt = None  # release activations as we go
        # ---- MLP branch: x_out = x_mid + g Wo2^T
        dg = K.linear_dgrad(gx16, Wo2_b, Wo2_t)
        dWo2 = K.linear_wgrad(gx16, g) if n_o2 else None
        dh = K.geglu_bwd(dg, h)
        del dg, g
        dxn2 = K.linear_dgrad(dh, Wi_b, Wi_t)
        dWi = K.linear_wgrad(dh, xn2) if n_i else None
        del dh, h, xn2
        gx32, gx16, dw_mn = K.layernorm_bwd(dxn2, x_mid, w_mn, mean_m, rstd_m, gx32, True)
        del dxn2, x_mid
        # ---- attention branch: x_mid = x + o Wo^T
        do = K.linear_dgrad(gx16, Wo_b, Wo_t)
        dWo = K.linear_wgrad(gx16, o) if n_o else None
        # attention backward; the inverse rotary rotation of dq / dk is applied in its epilogue
        if geo.hd != 64:
            dqkv = K.attn_bwd_generic(qkv, o, do, lse, geo.key_mask, B, S, nh, geo.hd, geo.windows[i], scale)
            K.rope_apply_generic_(dqkv, geo.rope[i][0], geo.rope[i][1], B, S, nh, geo.hd, geo.per_batch_pos, inverse=True)
        elif geo.cu is not None:
            dqkv = K.attn_bwd_varlen(qkv, o, do, lse, geo.cu, B, geo.max_s, nh, geo.windows[i], scale, geo.rope[i], prescaled=True)
        else:
            dqkv = K.attn_bwd(qkv, o, do, lse, geo.key_mask, B, S, nh, geo.windows[i], scale, geo.rope[i], geo.per_batch_pos,
                              prescaled=True)
        del do, o, qkv
        dWqkv = K.linear_wgrad(dqkv, xn) if n_qkv else None
        if i == 0:
            dw_an = None
            if need[2]:
                dxn = K.linear_dgrad(dqkv, Wqkv_b, Wqkv_t)
                gx32, _ = K.add_f32(gx32, dxn, want_bf16=False)
            _hand_upstream(geo, gx32, None)
        elif need[2] or n_an:
            dxn = K.linear_dgrad(dqkv, Wqkv_b, Wqkv_t)
            gx32, gx16, dw_an = K.layernorm_bwd(dxn, x, w_an, mean_a, rstd_a, gx32, True)
            _hand_upstream(geo, gx32, gx16)
        else:  # everything below this layer is frozen: the chain ends here
            dw_an = None
        grads = [dWqkv, dWo, dw_mn, dWi, dWo2] if i == 0 else [dw_an, dWqkv, dWo, dw_mn, dWi, dWo2]
        out = [(gw if gw.dtype == dt else gw.to(dt)) if (nd and gw is not None) else None for gw, dt, nd in zip(grads, ctx.wdtypes, need[3:])]
        return (None, None, gx32 if need[2] else None, *out)


class _FinalNormFn(torch.autograd.Function):
    """final_norm of the stack (TF:...modeling_modernbert.py:472): x [T,H] fp32 -> LayerNorm(x) fp32; its backward starts the
    chain of bf16 gradient twins."""

    @staticmethod
    def forward(ctx, geo: _Geometry, x: Tensor, norm_w: Tensor):
        w = _f32(norm_w.detach())
        y, _, mean, rstd = K.layernorm_fwd(x, w, geo.eps, True, False, geo.save)
        if geo.save:
            ctx.geo, ctx.pack, ctx.wdtype = geo, (x, w, mean, rstd), norm_w.dtype
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, w, mean, rstd = ctx.pack
        gx32, gx16, dw = K.layernorm_bwd(dy.contiguous(), x, w, mean, rstd, None, True, inplace=False)
        ctx.pack = None
        _hand_upstream(ctx.geo, gx32, gx16)
        return None, gx32 if ctx.needs_input_grad[1] else None, dw.to(ctx.wdtype) if ctx.needs_input_grad[2] else None


class _EmbedLNFn(torch.autograd.Function):
    """LayerNorm(tok_embeddings[ids]) with optional audio rows scattered over the placeholder tokens
    (TF:...modeling_modernbert.py:64-71, ref:cm3p/modeling_cm3p.py:592,603-605)."""

    @staticmethod
    def forward(ctx, ids: Tensor, table: Tensor, norm_w: Tensor, eps: float, padding_idx: int, slot: Optional[Tensor],
                audio_rows: Optional[Tensor]):
        w = _f32(norm_w.detach())
        tab = table.detach()
        ar = audio_rows.detach().contiguous() if audio_rows is not None else None
        y, _, mean, rstd = K.embed_ln_fwd(ids, tab, w, eps, slot, ar)
        ctx.pack = (ids, tab, w, mean, rstd, slot, ar, padding_idx)
        ctx.dtypes = (table.dtype, norm_w.dtype, audio_rows.dtype if audio_rows is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        ids, tab, w, mean, rstd, slot, ar, padding_idx = ctx.pack
        d_table, d_audio, dw = K.embed_ln_bwd(dy.contiguous(), ids, tab, w, mean, rstd, padding_idx, slot, ar,
                                              want_table_grad=ctx.needs_input_grad[1])
        td, wd, ad = ctx.dtypes
        if d_table is not None and d_table.dtype != td:
            d_table = d_table.to(td)
        if d_audio is not None and d_audio.dtype != ad:
            d_audio = d_audio.to(ad)
        return None, d_table, dw.to(wd), None, None, None, d_audio


class _PadRowsFn(torch.autograd.Function):
    """Packed rows -> padded [rows, H] with zeros at the padding positions (ref:cm3p/modeling_cm3p.py:106-134 _pad_cm3p_output);
    backward gathers the rows back."""

    @staticmethod
    def forward(ctx, y: Tensor, idx: Tensor, n_valid: int, rows: int):
        ctx.idx, ctx.n_valid, ctx.packed_rows = idx, n_valid, y.shape[0]
        return K.scatter_rows(y[:n_valid].contiguous() if n_valid != y.shape[0] else y.contiguous(), idx, rows)

    @staticmethod
    def backward(ctx, dy: Tensor):
        g = K.gather_rows(dy.contiguous(), ctx.idx)
        if ctx.packed_rows != ctx.n_valid:  # alignment rows of the packed layout take no gradient
            g = torch.cat((g, torch.zeros((ctx.packed_rows - ctx.n_valid, g.shape[1]), dtype=g.dtype, device=g.device)))
        return g, None, None, None


class _LayerNormFn(torch.autograd.Function):
    """Plain LayerNorm of given rows (the `inputs_embeds` entry of ModernBertEmbeddings, TF:...modeling_modernbert.py:67-68)."""

    @staticmethod
    def forward(ctx, x: Tensor, norm_w: Tensor, eps: float):
        w = _f32(norm_w.detach())
        xd = x.detach().contiguous()
        y, _, mean, rstd = K.layernorm_fwd(xd, w, eps, True, False)
        ctx.pack = (xd, w, mean, rstd)
        ctx.dtypes = (x.dtype, norm_w.dtype)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        xd, w, mean, rstd = ctx.pack
        x32 = xd if xd.dtype == torch.float32 else xd.float()
        dx, _, dw = K.layernorm_bwd(dy.contiguous(), x32, w, mean, rstd, None, False, inplace=False)
        return (dx if ctx.dtypes[0] == torch.float32 else dx.to(ctx.dtypes[0])), dw.to(ctx.dtypes[1]), None


class CM3PEncoder(nn.Module):
    """Drop-in for the `ModernBertModel` member of the reference towers (same submodule / parameter names)."""

    def __init__(self, config):
        super().__init__()
        _check_supported(config)
        self.config = config
        self.embeddings = CM3PEmbeddingParams(config)
        self.layers = nn.ModuleList([CM3PEncoderLayerParams(config, i) for i in range(config.num_hidden_layers)])
        self.final_norm = nn.LayerNorm(config.hidden_size, eps=config.norm_eps, bias=False)
        self.gradient_checkpointing = False  # set by PreTrainedModel.gradient_checkpointing_enable()
        self._inv_freq_cache = {}
        # init roles (TF:...modeling_modernbert.py:372-386): 'in'/'embedding' std = initializer_range,
        # 'out' std = initializer_range / sqrt(2 L); consumed by CM3PPreTrainedModel._init_weights
        cutoff = config.initializer_cutoff_factor or 3
        std_in = config.initializer_range
        std_out = config.initializer_range / (2.0 * config.num_hidden_layers) ** 0.5
        self.embeddings.tok_embeddings._cm3p_init = (std_in, cutoff)
        for layer in self.layers:
            layer.attn.Wqkv._cm3p_init = (std_in, cutoff)
            layer.attn.Wo._cm3p_init = (std_out, cutoff)
            layer.mlp.Wi._cm3p_init = (std_in, cutoff)
            layer.mlp.Wo._cm3p_init = (std_out, cutoff)

    def train(self, mode: bool = True):
        if mode != self.training:
            invalidate_weight_cache()  # HF Trainer flips the mode around every evaluation: copies never cross a train/eval boundary
        return super().train(mode)

    def get_input_embeddings(self):
        return self.embeddings.tok_embeddings

    def set_input_embeddings(self, value):
        self.embeddings.tok_embeddings = value

    def _inv_freq(self, theta: float, device) -> Tensor:
        key = (float(theta), str(device))
        if key not in self._inv_freq_cache:
            # TF:...modeling_modernbert.py:141, evaluated on the host exactly as the reference does
            hd = self.config.hidden_size // self.config.num_attention_heads
            inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
            self._inv_freq_cache[key] = inv.to(device)
        return self._inv_freq_cache[key]

    def _stack_weights(self):
        """Per layer: [attn_norm (layers > 0)], Wqkv, Wo, mlp_norm, Wi, Wo(mlp) - the argument order of _EncoderLayerFn."""
        ws = []
        for i, layer in enumerate(self.layers):
            w = [layer.attn_norm.weight] if i > 0 else []
            ws.append(w + [layer.attn.Wqkv.weight, layer.attn.Wo.weight, layer.mlp_norm.weight, layer.mlp.Wi.weight, layer.mlp.Wo.weight])
        return ws

    def forward(self, input_ids: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None,
                position_ids: Optional[Tensor] = None, inputs_embeds: Optional[Tensor] = None,
                audio_slot: Optional[Tensor] = None, audio_rows: Optional[Tensor] = None, unpad: bool = False,
                output_hidden_states: bool = False, cu_seqlens: Optional[Tensor] = None, max_seqlen: Optional[int] = None,
                output_attentions: bool = False):
        """-> last_hidden_state (B, S, H) fp32 [, tuple of L+1 detached hidden states (the stack's input and every layer's output,
        TF:...modeling_modernbert.py:457-470) when output_hidden_states].  Exactly one of input_ids / inputs_embeds.
        output_attentions: -> (last_hidden_state, hidden states or None, tuple of L attention-probability tensors (B, nh, S, S) fp32,
        detached): what the reference returns as `attentions` (TF runs its eager attention for such a call); padded execution only.

        unpad: run the stack on the valid tokens only, packed back to back (what the reference's flash_attention_2 path does,
        ref:cm3p/modeling_cm3p.py:911-931); padding positions of the result are zero.  Used when the mask is a right-padded
        one with at least one padded position; otherwise the padded path runs (same values on the valid positions)."""
        cfg = self.config
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if cfg.hidden_size // cfg.num_attention_heads != 64:
            # the generic attention kernels (head_dim 16 / 32) know the padded layout only
            if cu_seqlens is not None:
                raise NotImplementedError("unpadded inputs need head_dim 64 (the generic attention kernels run padded batches)")
            if output_attentions:
                raise NotImplementedError("output_attentions needs head_dim 64")
            unpad = False
        if cu_seqlens is not None:
            if output_attentions:
                raise NotImplementedError("output_attentions with unpadded inputs: attention probabilities are (B, nh, S, S) tensors of a padded batch")
            return self._forward_prepacked(input_ids, position_ids, audio_slot, audio_rows, cu_seqlens, max_seqlen, output_hidden_states)
        if output_attentions:
            unpad = False  # the probabilities are laid out per padded (batch, head, query, key)
        if input_ids is not None and input_ids.dtype != torch.int64:
            input_ids = input_ids.to(torch.int64)  # nn.Embedding takes IntTensor or LongTensor; the kernels index with int64
        ref = input_ids if input_ids is not None else inputs_embeds
        if not ref.is_cuda:
            raise RuntimeError("cm3p_amd runs on MI355X only: inputs must be CUDA/HIP tensors (no CPU fallback)")
        B, S = ref.shape[0], ref.shape[1]
        H = cfg.hidden_size
        dev = ref.device

        packed = None
        if unpad and attention_mask is not None and input_ids is not None:
            packed = self._plan_unpadded(attention_mask.reshape(B, S), position_ids)

        if packed is not None:
            idx, cu, max_s, n_valid, n_rows, pos = packed
            ids = input_ids.contiguous().view(-1)[idx]  # integer row selection (the reference's _unpad_cm3p_input)
            slot_p = None
            if audio_slot is not None:  # audio placeholders are valid tokens: their slot numbers travel with them
                slot_p = audio_slot.view(-1)[idx]
            if n_rows != n_valid:  # alignment rows: one extra pseudo-sequence of pad tokens, no gradient flows into it
                ids = torch.cat((ids, ids.new_zeros(n_rows - n_valid)))
                if slot_p is not None:
                    slot_p = torch.cat((slot_p, slot_p.new_full((n_rows - n_valid,), -1)))
            pad = self.embeddings.tok_embeddings.padding_idx
            x0 = _EmbedLNFn.apply(ids, self.embeddings.tok_embeddings.weight, self.embeddings.norm.weight, cfg.norm_eps,
                                  -1 if pad is None else pad, None if slot_p is None else slot_p.contiguous(), audio_rows)
        elif input_ids is not None:
            pad = self.embeddings.tok_embeddings.padding_idx
            x0 = _EmbedLNFn.apply(input_ids.contiguous().view(-1), self.embeddings.tok_embeddings.weight,
                                  self.embeddings.norm.weight, cfg.norm_eps, -1 if pad is None else pad, audio_slot, audio_rows)
        else:
            x0 = _LayerNormFn.apply(inputs_embeds.reshape(B * S, H), self.embeddings.norm.weight, cfg.norm_eps)

        y, hiddens, attns = self._run_stack(x0, B, S, attention_mask, position_ids, packed, output_hidden_states, dev, output_attentions)
        if hiddens is not None:
            if packed is not None:
                hiddens = [K.scatter_rows(h[:n_valid].contiguous(), idx, B * S) for h in hiddens]
            hiddens = tuple(h.view(B, S, H) for h in hiddens)
        if packed is not None:
            y = _PadRowsFn.apply(y, idx, n_valid, B * S)
        y = y.view(B, S, H)
        if output_attentions:
            return y, hiddens, attns
        return (y, hiddens) if output_hidden_states else y

    def _run_stack(self, x0: Tensor, B: int, S: int, attention_mask, position_ids, packed, output_hidden_states: bool, dev,
                   output_attentions: bool = False):
        """The L encoder layers + final norm on [rows, H]; `packed` = (idx, cu, max_s, n_valid, n_rows, pos) for unpadded execution."""
        cfg = self.config
        H = cfg.hidden_size
        geo = _Geometry()
        geo.B, geo.S, geo.H, geo.I, geo.nh, geo.L = B, S, H, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_hidden_layers
        geo.hd = H // cfg.num_attention_heads
        geo.eps = cfg.norm_eps
        geo.windows = [-1 if cfg.is_global_layer(i) else cfg.half_window for i in range(geo.L)]
        geo.key_mask = None
        geo.cu = None
        geo.checkpoint = bool(self.gradient_checkpointing and self.training)
        geo.max_s = S
        if packed is not None:
            idx, cu, max_s, n_valid, n_rows, pos = packed
            geo.B = cu.numel() - 1  # (+1 when alignment rows form a pseudo-sequence)
            geo.S = max_s
            geo.cu, geo.max_s = cu, max_s
            geo.per_batch_pos = True  # rotary tables are per packed token
            pos_tok = pos
        else:
            if attention_mask is not None:
                # padding term of the reference mask depends on the key only (TF:masking_utils.py:168-179)
                geo.key_mask = (attention_mask.reshape(B, S) != 0).to(torch.uint8).contiguous()
            if position_ids is None:
                position_ids = torch.arange(S, device=dev).unsqueeze(0)
            geo.per_batch_pos = position_ids.shape[0] != 1
            if geo.per_batch_pos and position_ids.shape[0] != B:
                raise ValueError("position_ids must be (1, S) or (B, S)")
            pos_tok = position_ids.contiguous().to(torch.int64)
        tables = {}
        for is_global, theta in ((True, cfg.global_rope_theta), (False, cfg.local_rope_theta)):
            if any(cfg.is_global_layer(i) == is_global for i in range(geo.L)):
                tables[is_global] = K.rope_table(pos_tok, self._inv_freq(theta, dev))
        geo.rope = [tables[cfg.is_global_layer(i)] for i in range(geo.L)]
        weights = self._stack_weights()
        geo.save = torch.is_grad_enabled() and (x0.requires_grad or self.final_norm.weight.requires_grad
                                                or any(w.requires_grad for ws in weights for w in ws))
        if geo.save and _eval_weights:
            invalidate_weight_cache()  # a backward will follow, so an optimizer will: no copy made before this step may outlive it
        geo.wcast = None
        if geo.save and os.environ.get("CM3P_CAST_BATCHED", "1") != "0":
            # bf16 copies + transposes of every projection weight of the stack in ONE launch (r03: 4 launches per layer)
            elig = [w for ws in weights for w in ws
                    if w.dim() == 2 and w.dtype == torch.float32 and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0 and w.is_contiguous()]
            if elig:
                geo.wcast = {id(w): pair for w, pair in zip(elig, K.cast_bf16_with_transpose_many([w.detach() for w in elig]))}
        geo.handoff = None
        geo.attn_out = [] if output_attentions else None
        # output_hidden_states: the stack's input and every layer's output (TF:...modeling_modernbert.py:457-470), detached
        hiddens = [x0.detach()] if output_hidden_states else None
        x = x0
        for i in range(geo.L):
            x = _EncoderLayerFn.apply(geo, i, x, *weights[i])
            if hiddens is not None:
                hiddens.append(x.detach())
        y = _FinalNormFn.apply(geo, x, self.final_norm.weight)
        attns = tuple(geo.attn_out) if geo.attn_out is not None else None
        geo.attn_out = None
        return y, hiddens, attns

    def _forward_prepacked(self, input_ids: Tensor, position_ids: Optional[Tensor], audio_slot, audio_rows, cu_seqlens: Tensor,
                           max_seqlen: Optional[int], output_hidden_states: bool):
        """Caller-supplied unpadded inputs (ref:cm3p/modeling_cm3p.py:911-931 when `indices` / `cu_seqlens` / `max_seqlen` are given;
        the layout of _unpad_cm3p_input, :65-104): input_ids (total,), cu_seqlens (batch + 1,) -> last_hidden_state (total, H),
        NOT re-padded (the reference's encoder leaves caller-packed rows packed as well)."""
        cfg = self.config
        if input_ids is None or input_ids.dim() != 1:
            raise ValueError("with cu_seqlens, input_ids must be the 1-D unpadded token tensor (total_nnz,)")
        if not input_ids.is_cuda:
            raise RuntimeError("cm3p_amd runs on MI355X only: inputs must be CUDA/HIP tensors (no CPU fallback)")
        dev = input_ids.device
        ids = input_ids.to(torch.int64).contiguous()
        total = ids.numel()
        cu = cu_seqlens.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
        if cu.numel() < 2:
            raise ValueError("cu_seqlens needs at least two entries (batch + 1)")
        lens = cu[1:] - cu[:-1]
        # ONE host read validates the description (a wrong cu_seqlens would send the kernels past the rows)
        first, last, mx, mn = torch.stack((cu[0], cu[-1], lens.max(), lens.min())).tolist()
        if first != 0 or last != total or mn <= 0:
            raise ValueError(f"cu_seqlens must start at 0, increase strictly and end at the token count ({total}); got end {last}, min length {mn}")
        max_s = int(mx) if max_seqlen is None else int(max_seqlen)
        if max_s < mx:
            raise ValueError(f"max_seqlen {max_s} is smaller than the longest sequence ({mx})")
        n_rows = (total + 63) // 64 * 64  # alignment rows: one pseudo-sequence of pad tokens, no gradient flows into it
        if position_ids is None:
            pos = torch.arange(total, device=dev) - torch.repeat_interleave(cu[:-1].to(torch.int64), lens.to(torch.int64))
        else:
            pos = position_ids.reshape(-1).to(torch.int64)
            if pos.numel() != total:
                raise ValueError("position_ids must be unpadded like input_ids")
        slot_p = audio_slot
        if n_rows != total:
            ids = torch.cat((ids, ids.new_zeros(n_rows - total)))
            pos = torch.cat((pos, torch.arange(n_rows - total, device=dev)))
            cu = torch.cat((cu, cu.new_full((1,), n_rows)))
            if slot_p is not None:
                slot_p = torch.cat((slot_p, slot_p.new_full((n_rows - total,), -1)))
        pad = self.embeddings.tok_embeddings.padding_idx
        x0 = _EmbedLNFn.apply(ids, self.embeddings.tok_embeddings.weight, self.embeddings.norm.weight, cfg.norm_eps,
                              -1 if pad is None else pad, None if slot_p is None else slot_p.contiguous(), audio_rows)
        packed = (None, cu, max(max_s, n_rows - total), total, n_rows, pos.contiguous())
        y, hiddens, _ = self._run_stack(x0, cu.numel() - 1, max_s, None, None, packed, output_hidden_states, dev)
        if n_rows != total:
            y = y[:total]
            if hiddens is not None:
                hiddens = [h[:total] for h in hiddens]
        return (y, tuple(hiddens)) if output_hidden_states else y

    @staticmethod
    def _plan_unpadded(mask: Tensor, position_ids: Optional[Tensor]):
        """Index bookkeeping of _unpad_cm3p_input (ref:cm3p/modeling_cm3p.py:88-104) on the device plus ONE host read (the
        reference reads max_seqlen the same way).  -> (indices, cu_seqlens, max_seqlen, n_valid, n_rows, positions) or None
        when unpadding does not apply: nothing padded, an empty row, or a mask that is not right-padded (then the padded path
        keeps the reference's sdpa semantics exactly)."""
        B, S = mask.shape
        m = mask != 0
        lens = m.sum(dim=1, dtype=torch.int32)
        last = (m * torch.arange(1, S + 1, device=mask.device)).amax(dim=1).to(torch.int32)  # 1 + index of the last valid key
        total, max_s, min_s, prefix = torch.stack((lens.sum(), lens.max(), lens.min(), (last == lens).all().to(torch.int32))).tolist()
        if total == B * S or min_s == 0 or not prefix:
            return None
        idx = torch.nonzero(m.flatten()).flatten()
        n_rows = (total + 63) // 64 * 64  # keeps the token-contraction GEMMs (weight gradients) on the 256 x 256 kernel
        seq = lens if n_rows == total else torch.cat((lens, lens.new_full((1,), n_rows - total)))
        cu = torch.zeros(seq.numel() + 1, dtype=torch.int32, device=mask.device)
        cu[1:] = torch.cumsum(seq, 0)
        if position_ids is None:
            pos = idx % S
        else:
            pos = position_ids.expand(B, S).reshape(-1)[idx]
        if n_rows != total:
            pos = torch.cat((pos, torch.arange(n_rows - total, device=mask.device)))
        return idx, cu, max(int(max_s), n_rows - total), int(total), int(n_rows), pos.contiguous().to(torch.int64)
