"""Seam 2 of SURVEY.md section 8(b): the HIP attention kernels as an attention function of the `transformers` registry.

`transformers` looks attention up by name (TF:models/modernbert/modeling_modernbert.py:282-297: `ALL_ATTENTION_FUNCTIONS.get_interface(
config._attn_implementation, ...)`) and builds the mask through a second registry indexed by the SAME name (TF:masking_utils.py:711-725,
1059, 1306).  Importing this module registers `"cm3p_hip"` in both, after which any model of the installed `transformers` whose
attention goes through the registry - the reference's `ModernBertModel` towers included - runs its attention on libcm3p_hip.so with

    config._attn_implementation = "cm3p_hip"          # ref:configs/train/default.yaml:8 -> ref:train.py:275

This is the NARROW seam: it cannot fuse LayerNorm / RoPE / GeGLU / the residual adds (the product replaces the modeling module for that,
cm3p_amd/modeling_cm3p.py); q, k, v arrive rotated and head-major and are re-packed into the [B, S, 3, nh, hd] layout the kernels read.
What it does carry over unchanged is the mask RULE, which is the reason for having it tested:

    key kv is visible to query q of batch b  iff  padding[b, kv]  and  (global layer  or  |q - kv| <= sliding_window - 1)

`ModernBertAttention.sliding_window` is `config.sliding_window + 1` = 65 for the default 128-token local attention (TF:...modeling_
modernbert.py:250-253: flash-attention's inclusive convention), so the band handed to the kernels is `sliding_window - 1` = 64 on each
side.  The (B, 1, S, S) mask is never built: the mask function registered here returns the 2-D key-padding mask as bytes (or None) and
the kernels rebuild the band from the window argument.  Rows with no visible key come out as exact zeros, as torch SDPA returns them.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import kernels as K

Tensor = torch.Tensor
NAME = "cm3p_hip"


def cm3p_hip_mask(batch_size: int, q_length: int, kv_length: int, q_offset: int = 0, kv_offset: int = 0, mask_function=None,
                  attention_mask: Optional[Tensor] = None, local_size: Optional[int] = None, use_vmap: bool = False, **kwargs) -> Optional[Tensor]:
    """Mask builder registered under the attention function's name (signature of TF:masking_utils.py:372 `sdpa_mask`).  Returns what
    `cm3p_hip_attention` wants as its `attention_mask`: the key-padding mask [B, kv_length] as uint8 (1 = visible), or None when every
    key is visible.  The sliding window is NOT folded in - the attention function receives it as `sliding_window` - and the
    bidirectional-skip rule of the reference (`_ignore_bidirectional_mask_sdpa`, TF:masking_utils.py:308-336) needs no counterpart:
    with no padding the kernels mask nothing but the band."""
    if use_vmap:
        # or_mask_function / and_mask_function overlays change the visibility rule itself; the kernels implement padding + band only
        raise NotImplementedError("cm3p_hip attention: custom or_/and_ mask functions are not supported (key padding and the sliding window only)")
    if q_length != kv_length or q_offset or kv_offset:
        raise NotImplementedError("cm3p_hip attention: encoder self-attention only (q_length == kv_length, no cache offsets)")
    if attention_mask is None:
        return None
    if attention_mask.dim() != 2:
        raise NotImplementedError("cm3p_hip attention: a 2-D (batch, key) padding mask is expected")
    return (attention_mask != 0).to(torch.uint8).contiguous()


class _AttnFn(torch.autograd.Function):
    """softmax(q k^T * scaling + mask) v on packed qkv [B, S, 3, nh, hd] bf16 -> [B, S, nh * hd] bf16; backward through the same library."""

    @staticmethod
    def forward(ctx, qkv: Tensor, key_mask: Optional[Tensor], B: int, S: int, nh: int, hd: int, window: int, scale: float):
        if hd == 64:
            out, lse = K.attn_fwd(qkv, key_mask, B, S, nh, window, scale, prescaled=False)
        else:
            out, lse = K.attn_fwd_generic(qkv, key_mask, B, S, nh, hd, window, scale)
        ctx.save_for_backward(qkv, out, lse)
        ctx.key_mask, ctx.geom = key_mask, (B, S, nh, hd, window, scale)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        qkv, out, lse = ctx.saved_tensors
        B, S, nh, hd, window, scale = ctx.geom
        dout = dout.contiguous()
        if hd == 64:
            dqkv = K.attn_bwd(qkv, out, dout, lse, ctx.key_mask, B, S, nh, window, scale, rope=None, prescaled=False)
        else:
            dqkv = K.attn_bwd_generic(qkv, out, dout, lse, ctx.key_mask, B, S, nh, hd, window, scale)
        return dqkv, None, None, None, None, None, None, None


def cm3p_hip_attention(module, query: Tensor, key: Tensor, value: Tensor, attention_mask: Optional[Tensor], dropout: float = 0.0,
                       scaling: Optional[float] = None, sliding_window: Optional[int] = None, **kwargs):
    """Attention function with the registry's calling convention (TF:integrations/sdpa_attention.py:77-166 is the one it stands in for):
    query / key / value (B, nh, S, hd), already rotated; -> (attn_output (B, S, nh, hd), None)."""
    if dropout:
        raise NotImplementedError("cm3p_hip attention: attention dropout must be 0 (the reference configs never set it)")
    if getattr(module, "is_causal", False) or kwargs.get("is_causal"):
        raise NotImplementedError("cm3p_hip attention: non-causal (encoder) attention only")
    B, nh, S, hd = query.shape
    if key.shape != query.shape or value.shape != query.shape:
        raise NotImplementedError("cm3p_hip attention: self-attention with equal q / k / v shapes (no grouped-query heads)")
    if attention_mask is not None and (attention_mask.dim() != 2 or attention_mask.dtype != torch.uint8):
        # a 4-D mask means the model built it under another name (or the caller passed one in): refuse rather than guess its rule
        raise NotImplementedError("cm3p_hip attention: expected the (batch, key) byte mask of cm3p_hip_mask; register the model's mask "
                                  "function under the same name (AttentionMaskInterface) and pass a 2-D attention_mask")
    scale = float(scaling) if scaling is not None else hd ** -0.5
    # TF:...modeling_modernbert.py:250-253: local layers pass config.sliding_window + 1 (inclusive flash-attention convention), global None
    window = -1 if sliding_window is None else int(sliding_window) - 1
    out = _run(query, key, value, attention_mask, window, scale)
    return out.view(B, S, nh, hd).to(query.dtype), None


def _run(query: Tensor, key: Tensor, value: Tensor, key_mask: Optional[Tensor], window: int, scale: float) -> Tensor:
    """(B, nh, S, hd) x 3 + the byte mask + the half-window (-1: global) -> [B * S, nh * hd] bf16 on the HIP kernels.  The only function
    of this module that touches the library (tests/test_hf_attention.py swaps it for a dense fp32 restatement of the same rule to check,
    on the CPU, that the seam hands the reference's visibility rule through unchanged)."""
    B, nh, S, hd = query.shape
    if not query.is_cuda:
        raise RuntimeError("cm3p_hip attention needs the tensors on an MI355X: there is no CPU path")
    if hd != 64 and not K.attn_generic_supported(hd):
        raise NotImplementedError(f"cm3p_hip attention: head_dim {hd} is not supported (64, or 16 / 32 on the generic kernels)")
    # -> [B, S, 3, nh, hd] bf16: a copy - this seam's price (the product's Wqkv GEMM writes that layout directly)
    qkv = torch.stack((query, key, value), dim=1).permute(0, 3, 1, 2, 4).to(torch.bfloat16).contiguous()
    return _AttnFn.apply(qkv, key_mask, B, S, nh, hd, window, scale)


def register() -> None:
    """Idempotent: put the attention function and its mask builder into the two registries of the installed `transformers`."""
    from transformers import AttentionInterface, AttentionMaskInterface

    AttentionInterface.register(NAME, cm3p_hip_attention)
    AttentionMaskInterface.register(NAME, cm3p_hip_mask)


register()
