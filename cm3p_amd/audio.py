"""Audio front end and multi-modal projector of the beatmap tower on HIP kernels (ref:cm3p/modeling_cm3p.py:470-528).

conv1d(k3,p1) -> GELU -> conv1d(k3,s2,p1) -> GELU are im2col + bf16 MFMA GEMM + a fused bias/GELU pass; activations are
token-major [B, T, C] throughout, so the reference's permute(0, 2, 1) is free.  The projector is
Linear(2048->768) -> GELU -> Linear(768->768) on rows of 4 concatenated frames.
"""
from __future__ import annotations

import torch

from . import kernels as K
from ._lib import EPI_F32
from .encoder import _bf16_weight, _f32

Tensor = torch.Tensor


class _ConvGeluFn(torch.autograd.Function):
    """gelu(conv1d(x, W, b, kernel 3, padding 1, stride)) -> token-major [B, T_out, C_out] (bf16 or fp32)."""

    @staticmethod
    def forward(ctx, x: Tensor, W: Tensor, bias: Tensor, stride: int, token_major: bool, out_f32: bool):
        Co, Ci, _ = W.shape
        xd = x.detach().contiguous()
        if token_major:
            B, T_in, C_in = xd.shape
        else:
            B, C_in, T_in = xd.shape
        if C_in != Ci:
            raise ValueError(f"Conv1d expects {Ci} input channels, got {C_in}")  # (nn.Conv1d raises too; the kernel would read out of bounds)
        # the channel-major kernel reads fp32, the token-major one bf16: bf16 / fp16 / fp64 mel features (the reference's
        # extraction script passes bf16, ref:extract_beatmap_embeddings.py:228-230) are widened or narrowed first
        want = torch.bfloat16 if token_major else torch.float32
        if xd.dtype != want:
            xd = xd.to(want)
        patches, T_out = K.im2col_k3(xd, token_major, B, Ci, T_in, stride)
        Wb = _bf16_weight(W.detach().reshape(Co, Ci * 3))
        b32 = _f32(bias.detach()).contiguous()
        z = K.gemm(patches, Wb, B * T_out, Co, Ci * 3, True, True, EPI_F32)
        a16, a32 = K.bias_gelu_fwd(z, b32, not out_f32, out_f32)
        ctx.pack = (patches, Wb, z, b32, B, Ci, Co, T_in, T_out, stride, token_major, W.dtype, bias.dtype)
        out = a32 if out_f32 else a16
        return out.view(B, T_out, Co)

    @staticmethod
    def backward(ctx, da: Tensor):
        patches, Wb, z, b32, B, Ci, Co, T_in, T_out, stride, token_major, wd, bd = ctx.pack
        da = da.contiguous().view(B * T_out, Co)
        dz, db = K.bias_gelu_bwd(da, z, b32)
        dW = K.linear_wgrad(dz, patches).view(Co, Ci, 3)
        dx = None
        if ctx.needs_input_grad[0]:
            if not token_major:  # (the mel features are data; the reference never differentiates through them either)
                raise NotImplementedError("gradient w.r.t. channel-major conv input (input_features) is not implemented")
            dp = K.linear_dgrad(dz, Wb)
            dx = K.col2im_k3(dp, B, Ci, T_in, T_out, stride)
        return dx, dW.to(wd), db.to(bd), None, None, None


def audio_frontend(input_features: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """(B, n_mels, T) fp32 -> (B, T/2, hidden) fp32, ready for the audio encoder's embedding LayerNorm."""
    a1 = _ConvGeluFn.apply(input_features, w1, b1, 1, False, False)  # bf16 token-major (B, T, C)
    return _ConvGeluFn.apply(a1, w2, b2, 2, True, True)


class _ProjectorFn(torch.autograd.Function):
    """linear_2(gelu(linear_1(h))) (CM3PMultiModalProjector, ref:cm3p/modeling_cm3p.py:470-481); h fp32 [R, 4*hidden]."""

    @staticmethod
    def forward(ctx, h: Tensor, W1: Tensor, W2: Tensor):
        hb = K.cast_bf16(h.detach().contiguous())
        W1b, W2b = _bf16_weight(W1), _bf16_weight(W2)
        z1 = K.linear_fwd(hb, W1b)
        a1 = K.gelu_fwd(z1)
        R, D = a1.shape
        out = K.gemm(a1, W2b, R, W2b.shape[0], D, True, True, EPI_F32)
        ctx.pack = (hb, W1b, W2b, z1, a1, W1.dtype, W2.dtype)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        hb, W1b, W2b, z1, a1, d1, d2 = ctx.pack
        dob = K.cast_bf16(dout.contiguous())
        da1 = K.linear_dgrad(dob, W2b)
        dW2 = K.linear_wgrad(dob, a1)
        dz1 = K.gelu_bwd(da1, z1)
        R, Kin = hb.shape
        dh = K.gemm(dz1, W1b, R, Kin, dz1.shape[1], True, False, EPI_F32) if ctx.needs_input_grad[0] else None
        dW1 = K.linear_wgrad(dz1, hb)
        return dh, dW1.to(d1), dW2.to(d2)


def audio_projector(h: Tensor, W1: Tensor, W2: Tensor) -> Tensor:
    return _ProjectorFn.apply(h, W1, W2)
