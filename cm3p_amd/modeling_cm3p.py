"""CM3P model classes on the MI355X kernels, behind the reference's HuggingFace surface.

Drop-in for `cm3p.modeling_cm3p` on the contrastive training path (ref:cm3p/modeling_cm3p.py): same class names,
`forward` signature (incl. `return_loss=True` and `labels`, which HF Trainer introspects), `CM3POutput` field order,
submodule / state-dict names and Auto* registration, so `train.py` drives it unchanged.
All arithmetic runs in libcm3p_hip.so through autograd nodes defined here and in encoder.py; there is no PyTorch-op
fallback and inputs must live on the GPU.

Scope: CM3PModel's contrastive branch with both towers and the audio front end, plus the "next" rows:
CM3PModel's MLM head (`has_decoder_head`, loss_type "ForMaskedLM": `loss += 0.5 * mlm_loss`, the v7 recipe), padded AND
unpadded ("varlen") execution of the towers, and the stand-alone variants the reference's train.py imports
(`CM3PForMaskedLM`, `CM3PForBeatmapClassification`, `CM3PBeatmapModelWithProjection`, `CM3PMetadataModelWithProjection`),
all implemented on the same kernels below.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Optional

import torch
from torch import nn
from transformers import AutoModel
from transformers.modeling_outputs import BaseModelOutput, BaseModelOutputWithPooling
from transformers.modeling_utils import PreTrainedModel
from transformers.utils import ModelOutput

from . import kernels as K
from .configuration_cm3p import CM3PAudioConfig, CM3PBeatmapConfig, CM3PConfig, CM3PMetadataConfig
from .encoder import CM3PEncoder, _f32, _PadRowsFn

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------- outputs
@dataclass
class CM3PAudioModelOutput(BaseModelOutput):
    audio_embeds: Optional[torch.FloatTensor] = None


@dataclass
class CM3PBeatmapModelOutput(BaseModelOutputWithPooling):
    beatmap_embeds: Optional[torch.FloatTensor] = None
    audio_model_output: Optional[CM3PAudioModelOutput] = None


@dataclass
class CM3PMetadataModelOutput(BaseModelOutput):
    metadata_embeds: Optional[torch.FloatTensor] = None


@dataclass
class CM3POutput(ModelOutput):
    """Field ORDER is part of the interface: Trainer drops `loss` and passes the rest positionally, and the reference's
    compute_metrics indexes [0] and [4] (ref:cm3p/modeling_cm3p.py:237-244, ref:train.py:77,101)."""

    loss: Optional[torch.FloatTensor] = None
    logits_per_beatmap: Optional[Tensor] = None
    logits_per_metadata: Optional[Tensor] = None
    metadata_embeds: Optional[torch.FloatTensor] = None
    beatmap_embeds: Optional[torch.FloatTensor] = None
    logits: Optional[torch.FloatTensor] = None
    metadata_model_output: BaseModelOutputWithPooling = None
    beatmap_model_output: BaseModelOutputWithPooling = None

    def to_tuple(self) -> tuple[Any]:
        nested = ("metadata_model_output", "beatmap_model_output")
        return tuple(self[k] if k not in nested else getattr(self, k).to_tuple() for k in self.keys())


# ----------------------------------------------------------------------------------------------- autograd nodes
class _PoolFn(torch.autograd.Function):
    """cls: h[:, 0]; else masked mean in fp32 (ref:cm3p/modeling_cm3p.py:385-396,631-642)."""

    @staticmethod
    def forward(ctx, h: Tensor, mask: Optional[Tensor], cls: bool):
        Bn, S, H = h.shape
        m = mask.contiguous().to(torch.int64) if mask is not None else None
        pooled, count = K.pool_fwd(h.detach().contiguous(), m, Bn, S, cls)
        ctx.pack = (m, count, Bn, S, cls)
        return pooled

    @staticmethod
    def backward(ctx, dp: Tensor):
        m, count, Bn, S, cls = ctx.pack
        dh = K.pool_bwd(dp.contiguous(), m, count, Bn, S, cls)
        return dh.view(Bn, S, -1), None, None


class _ProjectFn(torch.autograd.Function):
    """y = x W^T in fp32: the bias-free projection heads (ref:cm3p/modeling_cm3p.py:761-762,959,971)."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor):
        x32, w32 = x.detach().contiguous(), _f32(w.detach()).contiguous()
        R, Hin = x32.shape
        P = w32.shape[0]
        y = K.gemm_f32(x32, w32, R, P, Hin, (Hin, 1), (Hin, 1))
        ctx.pack = (x32, w32, w.dtype)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x32, w32, wd = ctx.pack
        dy = dy.contiguous()
        R, Hin = x32.shape
        P = w32.shape[0]
        dx = K.gemm_f32(dy, w32, R, Hin, P, (P, 1), (1, Hin)) if ctx.needs_input_grad[0] else None  # dy W
        dw = K.gemm_f32(dy, x32, P, Hin, R, (1, P), (1, Hin)) if ctx.needs_input_grad[1] else None  # dy^T x
        if dw is not None and dw.dtype != wd:
            dw = dw.to(wd)
        return dx, dw


class _L2NormFn(torch.autograd.Function):
    """x / sqrt(sum x^2), no eps (ref:cm3p/modeling_cm3p.py:54-62,960,972)."""

    @staticmethod
    def forward(ctx, x: Tensor):
        y, norm = K.l2norm_fwd(x.detach().contiguous())
        ctx.pack = (y, norm)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        y, norm = ctx.pack
        return K.l2norm_bwd(dy.contiguous(), y, norm)


class _LogitsFn(torch.autograd.Function):
    """logits = (A B^T) * exp(logit_scale) (ref:cm3p/modeling_cm3p.py:976-977).  A [M,P], B [N,P] fp32."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor, log_scale: Tensor):
        a32, b32 = a.detach().contiguous(), b.detach().contiguous()
        s32 = _f32(log_scale.detach()).reshape(1).contiguous()
        M, P = a32.shape
        N = b32.shape[0]
        raw = K.gemm_f32(a32, b32, M, N, P, (P, 1), (P, 1))
        logits = K.scale_exp(raw, s32)
        ctx.pack = (a32, b32, s32, logits, log_scale.dtype)
        return logits

    @staticmethod
    def backward(ctx, dl: Tensor):
        a32, b32, s32, logits, sd = ctx.pack
        dl = dl.contiguous()
        M, P = a32.shape
        N = b32.shape[0]
        draw = K.scale_exp(dl, s32)
        da = K.gemm_f32(draw, b32, M, P, N, (N, 1), (1, P)) if ctx.needs_input_grad[0] else None  # draw B
        db = K.gemm_f32(draw, a32, N, P, M, (1, N), (1, P)) if ctx.needs_input_grad[1] else None  # draw^T A
        ds = None
        if ctx.needs_input_grad[2]:
            ds = K.dot_f32(dl, logits).reshape(()).to(sd)  # d/ds [raw * e^s] = logits
        return da, db, ds


class _CrossEntropySumFn(torch.autograd.Function):
    """loss = sum_k coef_k * mean_rows CE_k over strided views of the given logits tensors: the pieces of cm3p_loss
    (ref:cm3p/modeling_cm3p.py:33-51) without materialising `.t()` / `.permute().reshape()` / gathered rows.
    spec = (tensor index, rows, cols, row_stride, col_stride, row_offset|None, target, coef)."""

    @staticmethod
    def forward(ctx, specs, *logits: Tensor):
        ls = [l.detach().contiguous() for l in logits]
        grads = [torch.zeros_like(l) for l in ls]
        loss = None
        for (ti, rows, cols, rs, cs, roff, target, coef) in specs:
            lr = K.cross_entropy(ls[ti], rows, cols, rs, cs, target, roff, coef / rows, grads[ti])
            loss = K.sum_f32(lr, coef / rows, out=loss, accumulate=loss is not None)
        ctx.grads = grads
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g: Tensor):
        g = g.reshape(1).contiguous()
        return (None, *[K.scale_by(d, g) for d in ctx.grads])


def cm3p_loss_hip(logits_per_metadata: Tensor, metadata_variation_classes: Optional[Tensor] = None) -> Tensor:
    """cm3p_loss on the GPU (ref:cm3p/modeling_cm3p.py:33-51), both directions from one logits tensor."""
    L = logits_per_metadata
    dev = L.device
    if L.dim() == 3:
        Bm, V, Bb = L.shape
        if Bm != Bb:
            raise AssertionError("metadata and beatmap batch sizes differ")
        idx = K.first_zero_index(metadata_variation_classes.contiguous().to(torch.int64))  # (classes == 0).argmax(1)
        rows_b = torch.arange(Bm, device=dev, dtype=torch.int64)
        # metadata loss: rows L[b, idx[b], :]  -> row offset (b*V + idx[b]) * Bb          (integer index arithmetic only)
        roff = (rows_b * V + idx) * Bb
        # beatmap loss over L.permute(2,0,1).reshape(Bb, Bm*V): element (bm, m*V+v) = L[m, v, bm]; target m*V + idx[m]
        tgt_b = rows_b * V + idx
        specs = [(0, Bm, Bb, 0, 1, roff, rows_b, 0.5), (0, Bb, Bm * V, 1, Bb, None, tgt_b, 0.5)]
    else:
        Bm, Bb = L.shape
        t = torch.arange(Bm, device=dev, dtype=torch.int64)
        specs = [(0, Bm, Bb, Bb, 1, None, t, 0.5), (0, Bb, Bm, 1, Bb, None, t, 0.5)]
    return _CrossEntropySumFn.apply(specs, L)


def _mlm_head_forward(h: Tensor, Wd: Tensor, bd: Optional[Tensor], norm_w: Tensor, Wdec: Tensor, bdec: Optional[Tensor], eps: float):
    """decoder(norm(gelu(dense(h)))) (CM3PPredictionHead + decoder, ref:cm3p/modeling_cm3p.py:991,1229-1238) -> (fp32 logits
    [T, Vp], what the backward needs, parameter dtypes); Vp = vocab rounded up to a multiple of 64, so that the decoder's input-
    gradient GEMM, which contracts over Vp, runs on the 256 x 256 kernel (pad columns hold the zero pad weights' product;
    callers slice)."""
    from ._lib import EPI_F32, EPI_F32_BIAS
    from .encoder import _bf16_weight

    T, H = h.shape
    V = Wdec.shape[0]
    Vp = (V + 63) // 64 * 64
    hb = K.cast_bf16(h.detach().contiguous())
    Wd_b = _bf16_weight(Wd)
    Wdec_b = _bf16_weight(Wdec)
    if Vp != V:
        pad = torch.zeros((Vp, H), dtype=torch.bfloat16, device=h.device)
        pad[:V].copy_(Wdec_b)  # layout only: pad the vocabulary to the GEMM's column granularity
        Wdec_b = pad
    bd32 = _f32(bd.detach()).contiguous() if bd is not None else torch.zeros((H,), dtype=torch.float32, device=h.device)
    w32 = _f32(norm_w.detach()).contiguous()
    z = K.gemm(hb, Wd_b, T, H, H, True, True, EPI_F32)
    _, a32 = K.bias_gelu_fwd(z, bd32, False, True)
    _, y, mean, rstd = K.layernorm_fwd(a32, w32, eps, False, True)
    if bdec is not None:  # the decoder bias is added while the GEMM stores its tiles (no extra pass over [T, Vp])
        bp = torch.zeros((Vp,), dtype=torch.float32, device=h.device)
        bp[:V].copy_(_f32(bdec.detach()))
        logits = K.gemm(y, Wdec_b, T, Vp, H, True, True, EPI_F32_BIAS, resid=bp)
    else:
        logits = K.gemm(y, Wdec_b, T, Vp, H, True, True, EPI_F32)
    pack = (hb, Wd_b, Wdec_b, bd32, w32, z, a32, y, mean, rstd, V)
    meta = (Wd.dtype, bd.dtype if bd is not None else None, norm_w.dtype, Wdec.dtype, bdec.dtype if bdec is not None else None)
    return logits, pack, meta


def _mlm_head_backward(pack, meta, need_dh: bool, dlb: Tensor, dbdec_full: Optional[Tensor]):
    """dlb: bf16 [T, Vp] gradient of the padded logits; dbdec_full: its fp32 column sums.  -> gradients of
    (h, Wd, bd, norm_w, Wdec, bdec)."""
    from ._lib import EPI_F32

    hb, Wd_b, Wdec_b, bd32, w32, z, a32, y, mean, rstd, V = pack
    dWd_t, dbd_t, dnw_t, dWdec_t, dbdec_t = meta
    T, H = hb.shape
    dy = K.linear_dgrad(dlb, Wdec_b)
    dWdec = K.linear_wgrad(dlb, y)[:V]
    dbdec = dbdec_full[:V] if dbdec_t is not None else None
    da, _, dnw = K.layernorm_bwd(dy, a32, w32, mean, rstd, None, False, inplace=False)
    dz, dbd = K.bias_gelu_bwd(da, z, bd32)
    dh = K.gemm(dz, Wd_b, T, H, H, True, False, EPI_F32) if need_dh else None
    dWd = K.linear_wgrad(dz, hb)
    return (dh, dWd.to(dWd_t), dbd.to(dbd_t) if dbd_t is not None else None, dnw.to(dnw_t), dWdec.to(dWdec_t),
            dbdec.to(dbdec_t) if dbdec is not None else None)


class _MLMHeadFn(torch.autograd.Function):
    """The MLM head alone (logits wanted, no labels): see _mlm_head_forward."""

    @staticmethod
    def forward(ctx, h: Tensor, Wd: Tensor, bd: Optional[Tensor], norm_w: Tensor, Wdec: Tensor, bdec: Optional[Tensor], eps: float):
        logits, ctx.pack, ctx.meta = _mlm_head_forward(h, Wd, bd, norm_w, Wdec, bdec, eps)
        return logits

    @staticmethod
    def backward(ctx, dl: Tensor):
        dl = dl.contiguous()
        dbdec_full = K.colsum_f32(dl) if ctx.meta[4] is not None else None
        return (*_mlm_head_backward(ctx.pack, ctx.meta, ctx.needs_input_grad[0], K.cast_bf16(dl), dbdec_full), None)


class _MLMHeadLossFn(torch.autograd.Function):
    """MLM head + ForMaskedLMLoss in one node (ref:cm3p/modeling_cm3p.py:987-996; TF:loss/loss_utils.py:32-46,74-91: mean cross
    entropy over labels != -100, or the sum divided by `num_items_in_batch` when the Trainer supplies it) -> (fp32 logits
    [T, Vp], loss).  Keeping both in one node lets the backward write the logits' gradient once, in bf16, straight into the
    layout the decoder's dgrad / wgrad GEMMs read, together with the decoder-bias gradient: no [T, Vp] fp32 gradient tensor,
    no scale / cast / column-sum passes over it, and the 85 % of rows without a label are never read."""

    @staticmethod
    def forward(ctx, h: Tensor, Wd: Tensor, bd: Optional[Tensor], norm_w: Tensor, Wdec: Tensor, bdec: Optional[Tensor], eps: float,
                labels: Tensor, num_items: Optional[Tensor]):
        logits, ctx.pack, ctx.meta = _mlm_head_forward(h, Wd, bd, norm_w, Wdec, bdec, eps)
        V = Wdec.shape[0]
        lab = labels.reshape(-1).contiguous().to(torch.int64)
        if lab.numel() != logits.shape[0]:
            raise ValueError(f"labels have {lab.numel()} entries, the logits {logits.shape[0]} rows")
        if num_items is None:
            inv = K.inv_valid_count(lab, -100)
        else:
            n = num_items if torch.is_tensor(num_items) else torch.tensor(float(num_items))
            inv = (1.0 / n.to(device=logits.device, dtype=torch.float32)).reshape(1).contiguous()  # scalar plumbing
        loss_rows, lse_rows = K.ce_masked_stats(logits, V, lab, -100)
        loss = K.scale_by(K.sum_f32(loss_rows, 1.0), inv)
        ctx.ce = (logits, lab, lse_rows, inv, V)
        ctx.set_materialize_grads(False)
        return logits, loss.reshape(())

    @staticmethod
    def backward(ctx, dl_ext: Optional[Tensor], dloss: Optional[Tensor]):
        logits, lab, lse_rows, inv, V = ctx.ce
        dlb = colsum = None
        if dloss is not None:
            dlb, colsum = K.ce_masked_dlogits_bf16(logits, V, lab, -100, lse_rows, dloss.reshape(1).contiguous().float(), inv)
        if dl_ext is not None:  # the logits were also used outside the loss: add that gradient (not the Trainer's path)
            dl_ext = dl_ext.contiguous().float()
            cs = K.colsum_f32(dl_ext)
            if dlb is None:
                dlb, colsum = K.cast_bf16(dl_ext), cs
            else:
                _, dlb = K.add_f32(dl_ext, dlb, want_bf16=True, inplace=False)
                colsum, _ = K.add_f32(colsum, cs, want_bf16=False, inplace=False)
        if dlb is None:
            return (None,) * 9
        return (*_mlm_head_backward(ctx.pack, ctx.meta, ctx.needs_input_grad[0], dlb, colsum), None, None, None)


class _AddScaledFn(torch.autograd.Function):
    """a + c * b for 0-dim fp32 tensors on the device (`loss += 0.5 * mlm_loss`, ref:cm3p/modeling_cm3p.py:996)."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor, c: float):
        ctx.c = c
        out = a.detach().reshape(1).clone()
        K.sum_f32(b.detach().reshape(1).contiguous(), c, out=out, accumulate=True)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g: Tensor):
        gc = torch.empty((1,), dtype=torch.float32, device=g.device)
        K.sum_f32(g.reshape(1).contiguous(), ctx.c, out=gc)
        return g, gc.reshape(()), None


# ----------------------------------------------------------------------------------------------- base class
class CM3PPreTrainedModel(PreTrainedModel):
    config_class = CM3PConfig
    base_model_prefix = "cm3p"
    supports_gradient_checkpointing = True  # per-layer recompute inside encoder._EncoderLayerFn
    _supports_flash_attn_2 = True
    _supports_flash_attn = True
    _supports_sdpa = True
    _supports_flex_attn = False

    def _check_and_adjust_attn_implementation(self, attn_implementation, *args, **kwargs):
        # Attention always runs in the HIP flash kernels; the configured string ("sdpa", "flash_attention_2", "eager", as
        # `train.py` copies it from the Hydra config, ref:train.py:275) selects nothing and needs no extra package.
        return attn_implementation if attn_implementation is not None else "sdpa"

    @torch.no_grad()
    def _init_weights(self, module):
        """Initialisation rules of the reference wrapper (ref:cm3p/modeling_cm3p.py:262-297) and of the encoder it
        wraps (TF:models/modernbert/modeling_modernbert.py:353-408: truncated normal, 'in' std 0.02, 'out' 0.02/sqrt(2L)).
        HF only visits modules that own parameters, so the encoder's role-dependent stds ride on leaf tags."""
        cfg = self.config
        tag = getattr(module, "_cm3p_init", None)
        if tag is not None:  # encoder leaves, tagged with their role's std by CM3PEncoder.__init__
            std, cutoff = tag
            nn.init.trunc_normal_(module.weight, mean=0.0, std=std, a=-cutoff * std, b=cutoff * std)
        elif isinstance(module, nn.LayerNorm):
            module.weight.fill_(1.0)
        elif isinstance(module, (nn.Linear, nn.Conv1d)):
            nn.init.normal_(module.weight, std=cfg.initializer_range)
            if module.bias is not None:
                module.bias.zero_()
        elif isinstance(module, CM3PModel):
            nn.init.normal_(module.metadata_projection.weight, std=module.metadata_embed_dim ** -0.5 * cfg.initializer_factor)
            nn.init.normal_(module.beatmap_projection.weight, std=module.beatmap_embed_dim ** -0.5 * cfg.initializer_factor)
            module.logit_scale.fill_(cfg.logit_scale_init_value)


def _require_gpu(t: Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"cm3p_amd: {what} must be on the GPU; this build has no CPU path (use the reference package on CPU)")


# ----------------------------------------------------------------------------------------------- towers
class CM3PMetadataTransformer(nn.Module):
    """ref:cm3p/modeling_cm3p.py:300-403."""

    def __init__(self, config: CM3PMetadataConfig):
        super().__init__()
        self.config = config
        self.encoder = CM3PEncoder(config)
        self.unpad_inputs = None  # as CM3PBeatmapTransformer.unpad_inputs: run padded (B, L) batches on their valid tokens only

    def get_input_embeddings(self):
        return self.encoder.get_input_embeddings()

    def set_input_embeddings(self, value):
        self.encoder.set_input_embeddings(value)

    def forward(self, input_ids: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None, indices=None, cu_seqlens=None,
                max_seqlen=None, batch_size=None, seq_len=None, output_attentions=None, output_hidden_states=None,
                output_pooler: bool = True) -> BaseModelOutputWithPooling:
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        if indices is not None or cu_seqlens is not None:
            # caller-supplied unpadded rows (ref:cm3p/modeling_cm3p.py:359-372): the encoder runs packed; the reference has no pooling
            # for them in this tower (ref:cm3p/modeling_cm3p.py:383-384)
            if cu_seqlens is None:
                raise ValueError("unpadded inputs need cu_seqlens (and max_seqlen)")
            if output_attentions:
                raise NotImplementedError("output_attentions with unpadded inputs: attention probabilities are (B, nh, S, S) tensors of a padded batch")
            if output_pooler:  # (before the encoder runs: the reference's own message, ref:cm3p/modeling_cm3p.py:383-384)
                raise NotImplementedError("Pooling with unpadded input is not implemented yet.")
            _require_gpu(input_ids, "input_ids")
            h = self.encoder(input_ids=input_ids, cu_seqlens=cu_seqlens, max_seqlen=max_seqlen, output_hidden_states=bool(output_hidden_states))
            hiddens = None
            if output_hidden_states:
                h, hiddens = h
            return BaseModelOutputWithPooling(last_hidden_state=h, pooler_output=None, hidden_states=hiddens, attentions=None)
        _require_gpu(input_ids, "input_ids")
        is_3d = input_ids.dim() == 3
        B0 = input_ids.size(0)
        ids2, am2 = input_ids, attention_mask
        if is_3d:  # (B, V, L) -> (B*V, L), ref:cm3p/modeling_cm3p.py:351-357
            ids2 = input_ids.reshape(-1, input_ids.size(-1))
            am2 = attention_mask.reshape(-1, attention_mask.size(-1)) if attention_mask is not None else None
        # output_attentions: the probabilities of every layer, (B[*V], nh, L, L) fp32 (a separate inspection kernel: the flash
        # kernels never materialise them; the reference switches to eager attention for such a call)
        # unpadded execution of a padded batch (metadata rows are usually much shorter than the padded length - the 1000-variation
        # evaluation is mostly padding): asked for explicitly or, like the reference, by attn_implementation == flash_attention_2;
        # the encoder re-pads its output and keeps the padded path where packing does not apply
        unpad = self.unpad_inputs if self.unpad_inputs is not None else getattr(self.config, "_attn_implementation", None) == "flash_attention_2"
        h = self.encoder(input_ids=ids2, attention_mask=am2, output_hidden_states=bool(output_hidden_states),
                         output_attentions=bool(output_attentions), unpad=bool(unpad) and not output_attentions)
        hiddens = attns = None
        if output_attentions:
            h, hiddens, attns = h
        elif output_hidden_states:
            h, hiddens = h
        pooled = _PoolFn.apply(h, am2, bool(self.config.cls_embed)) if output_pooler else None
        if is_3d:
            h = h.view(B0, -1, h.size(-2), h.size(-1))
            if pooled is not None:
                pooled = pooled.view(B0, -1, pooled.size(-1))
            if hiddens is not None:
                hiddens = tuple(t.view(B0, -1, t.size(-2), t.size(-1)) for t in hiddens)
        return BaseModelOutputWithPooling(last_hidden_state=h, pooler_output=pooled, hidden_states=hiddens, attentions=attns)


class CM3PMultiModalProjector(nn.Module):
    """Parameter container for linear_1 / linear_2 (ref:cm3p/modeling_cm3p.py:470-481)."""

    def __init__(self, config: CM3PAudioConfig):
        super().__init__()
        if config.projector_hidden_act != "gelu":
            raise NotImplementedError("projector_hidden_act must be 'gelu'")
        self.linear_1 = nn.Linear(config.projector_intermediate_size, config.projector_dim, bias=False)
        self.linear_2 = nn.Linear(config.projector_dim, config.projector_dim, bias=False)


class CM3PAudioEncoder(nn.Module):
    """ref:cm3p/modeling_cm3p.py:484-528: conv1d x2 + GELU -> encoder -> 4-frame concat -> projector."""

    def __init__(self, config: CM3PAudioConfig):
        super().__init__()
        self.config = config
        self.conv1 = nn.Conv1d(config.n_mels, config.hidden_size, kernel_size=3, padding=1)
        self.conv2 = nn.Conv1d(config.hidden_size, config.hidden_size, kernel_size=3, stride=2, padding=1)
        self.encoder = CM3PEncoder(config)
        self.multi_modal_projector = CM3PMultiModalProjector(config)

    def forward(self, input_features: Tensor, output_attentions=None, output_hidden_states=None) -> CM3PAudioModelOutput:
        from .audio import audio_frontend, audio_projector

        _require_gpu(input_features, "input_features")
        x = audio_frontend(input_features, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias)  # (B, T/2, H)
        B, T2, _ = x.shape
        pos = torch.arange(T2, device=x.device).unsqueeze(0).repeat(B, 1)  # explicit per-row positions, :506-507
        attns = None
        if output_attentions:
            h, _, attns = self.encoder(inputs_embeds=x, position_ids=pos, output_attentions=True)
        else:
            h = self.encoder(inputs_embeds=x, position_ids=pos)
        audio_embeds = audio_projector(h.reshape(-1, self.config.projector_intermediate_size),
                                       self.multi_modal_projector.linear_1.weight, self.multi_modal_projector.linear_2.weight)
        return CM3PAudioModelOutput(audio_embeds=audio_embeds, last_hidden_state=h, hidden_states=None, attentions=attns)


class CM3PBeatmapTransformer(nn.Module):
    """ref:cm3p/modeling_cm3p.py:531-650."""

    def __init__(self, config: CM3PBeatmapConfig):
        super().__init__()
        self.config = config
        self.audio_encoder = CM3PAudioEncoder(config.audio_config)
        self.encoder = CM3PEncoder(config)
        self.unpad_inputs = None  # None: follow config._attn_implementation == 'flash_attention_2' (the reference's rule)

    def get_input_embeddings(self):
        return self.encoder.get_input_embeddings()

    def set_input_embeddings(self, value):
        self.encoder.set_input_embeddings(value)

    def forward(self, input_ids: Optional[Tensor] = None, input_features: Optional[Tensor] = None,
                attention_mask: Optional[Tensor] = None, sliding_window_mask=None, position_ids: Optional[Tensor] = None,
                inputs_embeds: Optional[Tensor] = None, indices=None, cu_seqlens=None, max_seqlen=None, batch_size=None,
                seq_len=None, output_attentions=None, output_hidden_states=None, output_pooler: bool = True) -> CM3PBeatmapModelOutput:
        audio_out = None
        ohs = bool(output_hidden_states)
        oat = bool(output_attentions)
        if oat and (indices is not None or cu_seqlens is not None):
            raise NotImplementedError("output_attentions with unpadded inputs: attention probabilities are (B, nh, S, S) tensors of a padded batch")
        if indices is not None or cu_seqlens is not None:
            # Caller-supplied unpadded rows (ref:cm3p/modeling_cm3p.py:911-931 with indices / cu_seqlens / max_seqlen given, layout of
            # _unpad_cm3p_input :65-104): input_ids (total_nnz,), last_hidden_state stays (total_nnz, H), CLS pooling reads row
            # cu_seqlens[:-1] of it (:624-627); mean pooling of unpadded rows is not implemented in the reference either (:628-629).
            if cu_seqlens is None:
                raise ValueError("unpadded inputs need cu_seqlens (and max_seqlen)")
            if inputs_embeds is not None:
                raise NotImplementedError("unpadded inputs_embeds are not supported; pass unpadded input_ids")
            _require_gpu(input_ids, "input_ids")
            slot = rows = None
            if input_features is not None:
                audio_out = self.audio_encoder(input_features)
                rows = audio_out.audio_embeds
                slot, count = K.audio_slots(input_ids.contiguous().view(-1).to(torch.int64), int(self.config.audio_token_id))
                n = int(count.item())
                if n != rows.shape[0]:
                    raise RuntimeError(f"shape mismatch: {n} audio placeholder tokens but {rows.shape[0]} audio embeddings")
            if output_pooler and not self.config.cls_embed:  # before the encoder runs (ref:cm3p/modeling_cm3p.py:628-629)
                raise NotImplementedError("Pooling with unpadded input is not implemented yet.")
            h = self.encoder(input_ids=input_ids, position_ids=position_ids, audio_slot=slot, audio_rows=rows, cu_seqlens=cu_seqlens,
                             max_seqlen=max_seqlen, output_hidden_states=ohs)
            hiddens = None
            if ohs:
                h, hiddens = h
            pooled = None
            if output_pooler:
                if not self.config.cls_embed:
                    raise NotImplementedError("Pooling with unpadded input is not implemented yet.")
                first = cu_seqlens[:-1].to(device=h.device, dtype=torch.int64).contiguous()
                pooled = _TakeRowsFn.apply(h, first)
            return CM3PBeatmapModelOutput(last_hidden_state=h, pooler_output=pooled, hidden_states=hiddens, attentions=None,
                                          audio_model_output=audio_out)
        if inputs_embeds is not None:
            if input_features is not None:
                raise NotImplementedError("input_features together with inputs_embeds is not supported")
            _require_gpu(inputs_embeds, "inputs_embeds")
            h = self.encoder(inputs_embeds=inputs_embeds, attention_mask=attention_mask, position_ids=position_ids,
                             output_hidden_states=ohs, output_attentions=oat)
        else:
            _require_gpu(input_ids, "input_ids")
            slot = rows = None
            if input_features is not None:
                audio_out = self.audio_encoder(input_features)
                rows = audio_out.audio_embeds
                slot, count = K.audio_slots(input_ids.contiguous().view(-1).to(torch.int64), int(self.config.audio_token_id))
                # the reference's masked assignment raises on a count mismatch (ref:cm3p/modeling_cm3p.py:603-605)
                n = int(count.item())
                if n != rows.shape[0]:
                    raise RuntimeError(f"shape mismatch: {n} audio placeholder tokens but {rows.shape[0]} audio embeddings")
            # unpadded execution when asked for explicitly (unpad_inputs) or, like the reference, when the configuration says
            # flash_attention_2 (ref:cm3p/modeling_cm3p.py:911-931); the encoder falls back to the padded path where it does not apply
            unpad = self.unpad_inputs if self.unpad_inputs is not None else \
                getattr(self.config, "_attn_implementation", None) == "flash_attention_2"
            h = self.encoder(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids, audio_slot=slot,
                             audio_rows=rows, unpad=bool(unpad), output_hidden_states=ohs, output_attentions=oat)
        hiddens = attns = None
        if oat:
            h, hiddens, attns = h
        elif ohs:
            h, hiddens = h
        pooled = _PoolFn.apply(h, attention_mask, bool(self.config.cls_embed)) if output_pooler else None
        return CM3PBeatmapModelOutput(last_hidden_state=h, pooler_output=pooled, hidden_states=hiddens, attentions=attns,
                                      audio_model_output=audio_out)


class CM3PPredictionHead(nn.Module):
    """Parameter container of the MLM head (ref:cm3p/modeling_cm3p.py:1229-1238): dense -> GELU -> LayerNorm."""

    def __init__(self, config: CM3PBeatmapConfig):
        super().__init__()
        if config.classifier_activation != "gelu" or config.norm_bias:
            raise NotImplementedError("MLM head: classifier_activation must be 'gelu' and norm_bias False")
        self.dense = nn.Linear(config.hidden_size, config.hidden_size, config.classifier_bias)
        self.norm = nn.LayerNorm(config.hidden_size, eps=config.norm_eps, bias=False)


class CM3PMetadataModel(CM3PPreTrainedModel):
    config_class = CM3PMetadataConfig

    def __init__(self, config: CM3PMetadataConfig):
        super().__init__(config)
        self.metadata_model = CM3PMetadataTransformer(config)
        self.post_init()

    def get_input_embeddings(self) -> nn.Module:
        return self.metadata_model.encoder.embeddings.tok_embeddings

    def set_input_embeddings(self, value):
        self.metadata_model.encoder.embeddings.tok_embeddings = value

    def forward(self, *args, **kwargs):
        return self.metadata_model(*args, **kwargs)


class CM3PBeatmapModel(CM3PPreTrainedModel):
    config_class = CM3PBeatmapConfig
    main_input_name = "input_ids"

    def __init__(self, config: CM3PBeatmapConfig):
        super().__init__(config)
        self.beatmap_model = CM3PBeatmapTransformer(config)
        self.post_init()

    def get_input_embeddings(self) -> nn.Module:
        return self.beatmap_model.encoder.embeddings.tok_embeddings

    def set_input_embeddings(self, value):
        self.beatmap_model.encoder.embeddings.tok_embeddings = value

    def forward(self, *args, **kwargs):
        return self.beatmap_model(*args, **kwargs)


# ----------------------------------------------------------------------------------------------- the dual tower
_tower_streams: dict = {}  # device -> the stream the metadata tower runs on beside the beatmap tower


def _tower_stream(device) -> "torch.cuda.Stream":
    st = _tower_streams.get(device)
    if st is None:
        st = _tower_streams[device] = torch.cuda.Stream(device)
    return st


def _tensors_of(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            yield from _tensors_of(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors_of(o)


class CM3PModel(CM3PPreTrainedModel):
    config_class = CM3PConfig

    def __init__(self, config: CM3PConfig):
        super().__init__(config)
        if not isinstance(config.metadata_config, CM3PMetadataConfig):
            raise TypeError(f"config.metadata_config is expected to be of type CM3PMetadataConfig but is of type {type(config.metadata_config)}.")
        if not isinstance(config.beatmap_config, CM3PBeatmapConfig):
            raise TypeError(f"config.beatmap_config is expected to be of type CM3PBeatmapConfig but is of type {type(config.beatmap_config)}.")
        if config.has_decoder_head and config.loss_type != "ForMaskedLM":
            # the reference silently falls back to a shifted causal-LM loss for any other value (TF:modeling_utils.py:4655-4667)
            raise NotImplementedError('has_decoder_head needs loss_type="ForMaskedLM" (the published v7 recipe)')
        self.loss_type = config.loss_type
        self.projection_dim = config.projection_dim
        self.metadata_embed_dim = config.metadata_config.hidden_size
        self.beatmap_embed_dim = config.beatmap_config.hidden_size
        self.metadata_model = CM3PMetadataTransformer(config.metadata_config)
        self.beatmap_model = CM3PBeatmapTransformer(config.beatmap_config)
        self.beatmap_projection = nn.Linear(self.beatmap_embed_dim, self.projection_dim, bias=False)
        self.metadata_projection = nn.Linear(self.metadata_embed_dim, self.projection_dim, bias=False)
        self.logit_scale = nn.Parameter(torch.tensor(float(config.logit_scale_init_value)))
        if config.has_decoder_head:  # MLM head on the beatmap tower (ref:cm3p/modeling_cm3p.py:765-767)
            bc = config.beatmap_config
            self.head = CM3PPredictionHead(bc)
            self.decoder = nn.Linear(bc.hidden_size, bc.vocab_size, bias=bc.decoder_bias)
        # Opt-in: in-batch negatives across all ranks of the default process group (new behaviour).
        self.gather_negatives = False
        self.unpad_inputs = None  # True / False overrides the reference's rule (unpad iff attn_implementation is flash_attention_2)
        self.post_init()

    def _overlap_towers(self, input_ids, metadata_ids) -> bool:
        """Run the metadata tower on a second stream beside the beatmap tower?  (CM3P_TOWER_OVERLAP=0 switches it off.)"""
        if input_ids is None or metadata_ids is None or not metadata_ids.is_cuda or self.gather_negatives:
            return False
        if os.environ.get("CM3P_TOWER_OVERLAP", "1") == "0":
            return False
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            return False
        return True

    def get_metadata_features(self, input_ids=None, output_attentions=None, output_hidden_states=None) -> Tensor:
        out = self.metadata_model(input_ids=input_ids)
        p = out.pooler_output
        return _ProjectFn.apply(p.reshape(-1, p.size(-1)), self.metadata_projection.weight).view(*p.shape[:-1], -1)

    def get_beatmap_features(self, input_ids=None, input_features=None, attention_mask=None, position_ids=None, inputs_embeds=None,
                             output_attentions=None, output_hidden_states=None) -> Tensor:
        out = self.beatmap_model(input_ids=input_ids, input_features=input_features, attention_mask=attention_mask,
                                 position_ids=position_ids, inputs_embeds=inputs_embeds)
        return _ProjectFn.apply(out.pooler_output, self.beatmap_projection.weight)

    def forward(
        self,
        input_ids: Optional[torch.LongTensor] = None,
        input_features: Optional[torch.FloatTensor] = None,
        metadata_ids: Optional[torch.LongTensor] = None,
        attention_mask: Optional[Tensor] = None,
        metadata_attention_mask: Optional[Tensor] = None,
        position_ids: Optional[torch.LongTensor] = None,
        inputs_embeds: Optional[torch.FloatTensor] = None,
        metadata_variation_classes: Optional[torch.LongTensor] = None,
        labels: Optional[Tensor] = None,
        indices: Optional[Tensor] = None,
        cu_seqlens: Optional[Tensor] = None,
        max_seqlen: Optional[int] = None,
        batch_size: Optional[int] = None,
        seq_len: Optional[int] = None,
        return_loss: Optional[bool] = True,
        output_attentions: Optional[bool] = None,
        output_hidden_states: Optional[bool] = None,
        output_logits: Optional[bool] = None,
        **kwargs,
    ) -> CM3POutput:
        """Contrastive forward (ref:cm3p/modeling_cm3p.py:849-1012).  Whatever `attn_implementation` says, attention runs in the
        HIP flash kernels, which take the key-padding mask as it is: padded batches need no host-side unpadding.  Unpadded
        execution is supported as well - `self.unpad_inputs = True` packs the valid tokens of a padded batch, and a caller may hand
        over rows it has unpadded itself (`indices` / `cu_seqlens` / `max_seqlen` / `batch_size` / `seq_len`, the arguments of the
        reference's flash_attention_2 branch, ref:cm3p/modeling_cm3p.py:911-931) - with the same results."""
        output_logits = output_logits if output_logits is not None else self.config.has_decoder_head
        if metadata_ids is not None and metadata_ids.dim() == 3 and return_loss and metadata_variation_classes is None:
            raise ValueError("When providing multiple metadata variations, metadata_variation_classes must be provided in order to compute loss correctly.")
        if output_logits and not self.config.has_decoder_head:
            raise ValueError("Cannot return logits when the model is not configured with a decoder head.")

        beatmap_embeds = beatmap_outputs = metadata_embeds = metadata_outputs = beatmap_pending = None
        logits_per_beatmap = logits_per_metadata = None
        loss = 0 if return_loss else None

        def run_metadata():
            mo = self.metadata_model(input_ids=metadata_ids, attention_mask=metadata_attention_mask,
                                     output_attentions=output_attentions, output_hidden_states=output_hidden_states)
            p = mo.pooler_output
            me = _L2NormFn.apply(_ProjectFn.apply(p.reshape(-1, p.size(-1)), self.metadata_projection.weight))
            return mo, me.view(*p.shape[:-1], -1)

        # The two towers do not meet before the logits.  On one GPU the metadata tower (a few hundred launches of a few workgroups:
        # 8192 tokens at C2) runs on a second stream BESIDE the beatmap tower instead of after it; autograd runs each node's
        # backward on its forward stream and orders the gradients across streams, so the backward overlaps the same way.
        # Not with gathered negatives / more than one rank (DDP's bucket hooks take the stream of the last gradient of a bucket).
        # (both towers learn the unpadding rule before either is launched: with the towers overlapped the metadata tower goes first)
        if self.unpad_inputs is not None or getattr(self.config, "_attn_implementation", None) == "flash_attention_2":
            self.beatmap_model.unpad_inputs = True if self.unpad_inputs is None else bool(self.unpad_inputs)
            self.metadata_model.unpad_inputs = self.beatmap_model.unpad_inputs
        side = None
        if self._overlap_towers(input_ids, metadata_ids):
            main = torch.cuda.current_stream(metadata_ids.device)
            side = _tower_stream(metadata_ids.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                metadata_outputs, metadata_embeds = run_metadata()

        if input_ids is not None:
            try:
                beatmap_outputs = self.beatmap_model(input_ids=input_ids, input_features=input_features, attention_mask=attention_mask,
                                                     position_ids=position_ids, inputs_embeds=inputs_embeds, indices=indices,
                                                     cu_seqlens=cu_seqlens, max_seqlen=max_seqlen, batch_size=batch_size, seq_len=seq_len,
                                                     output_attentions=output_attentions, output_hidden_states=output_hidden_states)
                beatmap_embeds = _L2NormFn.apply(_ProjectFn.apply(beatmap_outputs.pooler_output, self.beatmap_projection.weight))
            except BaseException:
                if side is not None:  # the metadata tower is in flight on the other stream: join it before its tensors are dropped
                    main.wait_stream(side)
                raise
            if self.gather_negatives and metadata_ids is not None:
                if metadata_ids.dim() == 2:
                    # start the all-gather now: it runs on the process group's side stream under the whole metadata tower and is
                    # joined just before the logits (cm3p_amd/dist.py)
                    from .dist import start_gather

                    beatmap_pending = start_gather(beatmap_embeds)
                else:
                    from .dist import warn_variations_stay_local

                    warn_variations_stay_local()

        if side is not None:
            # join: everything the caller (and the logits) will read was produced on `side`; its memory belongs to that stream's pool
            main.wait_stream(side)
            for t in _tensors_of((metadata_embeds, dict(metadata_outputs))):
                t.record_stream(main)
        elif metadata_ids is not None:
            try:
                metadata_outputs, metadata_embeds = run_metadata()
            except BaseException:
                if beatmap_pending is not None:  # never leave a collective un-joined behind an exception
                    beatmap_pending.wait()
                raise

        if metadata_embeds is not None and beatmap_embeds is not None:
            me2 = metadata_embeds.reshape(-1, metadata_embeds.size(-1))
            if self.gather_negatives and metadata_embeds.dim() == 2:
                from .dist import gathered_contrastive

                logits_per_metadata, logits_per_beatmap, gl = gathered_contrastive(me2, beatmap_embeds, self.logit_scale,
                                                                                   beatmap_pending=beatmap_pending)
                if return_loss:
                    loss = gl
            else:
                lpm = _LogitsFn.apply(me2, beatmap_embeds, self.logit_scale)
                if metadata_embeds.dim() == 3:
                    logits_per_metadata = lpm.view(metadata_embeds.size(0), metadata_embeds.size(1), -1)
                    logits_per_beatmap = logits_per_metadata.permute(2, 0, 1)
                else:
                    logits_per_metadata = lpm
                    logits_per_beatmap = lpm.t()
                if return_loss:
                    loss = cm3p_loss_hip(logits_per_metadata, metadata_variation_classes)

        logits = None
        if output_logits:  # MLM head; with labels, loss += 0.5 * masked-LM loss (ref:cm3p/modeling_cm3p.py:987-996)
            if beatmap_outputs is None:
                raise ValueError("output_logits needs input_ids")
            hs = beatmap_outputs.last_hidden_state
            V = self.config.beatmap_config.vocab_size
            head_args = (hs.reshape(-1, hs.size(-1)), self.head.dense.weight, self.head.dense.bias, self.head.norm.weight,
                         self.decoder.weight, self.decoder.bias, self.config.beatmap_config.norm_eps)
            mlm = None
            if labels is not None and return_loss:
                lp, mlm = _MLMHeadLossFn.apply(*head_args, labels, kwargs.get("num_items_in_batch"))
            else:
                lp = _MLMHeadFn.apply(*head_args)
            if hs.dim() == 3:
                logits = lp.view(hs.size(0), hs.size(1), -1)[..., :V]
            else:
                # caller-supplied unpadded rows: logits (total_nnz, V), re-padded like the reference's flash_attention_2 branch
                # (_pad_cm3p_output, ref:cm3p/modeling_cm3p.py:999-1001) when the padded geometry was given
                logits = lp[:, :V]
                if indices is not None and batch_size is not None and seq_len is not None:
                    idx64 = indices.to(device=lp.device, dtype=torch.int64).reshape(-1).contiguous()
                    rows = int(batch_size) * int(seq_len)
                    # the scatter kernel writes row indices[i] of a [batch_size * seq_len] buffer for every packed row i: a wrong
                    # `indices` would be an out-of-bounds write on the GPU, not an exception - checked on the host (one read)
                    if idx64.numel() != lp.shape[0]:
                        raise ValueError(f"indices has {idx64.numel()} entries but there are {lp.shape[0]} unpadded rows")
                    if idx64.numel():
                        lo, hi = torch.stack((idx64.min(), idx64.max())).tolist()
                        if lo < 0 or hi >= rows:
                            raise ValueError(f"indices must lie in [0, batch_size * seq_len = {rows}); got [{lo}, {hi}]")
                    # the reference re-pads under no_grad only when labels are given and repad_logits_with_grad is off
                    # (ref:cm3p/modeling_cm3p.py:1000-1001); otherwise the padded logits stay differentiable
                    keep_grad = labels is None or bool(getattr(self.config.beatmap_config, "repad_logits_with_grad", False))
                    if keep_grad and lp.requires_grad:
                        padded = _PadRowsFn.apply(lp, idx64, lp.shape[0], rows)
                    else:
                        padded = K.scatter_rows(lp.detach().contiguous(), idx64, rows)
                    logits = padded.view(int(batch_size), int(seq_len), -1)[..., :V]
            if mlm is not None:
                if torch.is_tensor(loss):
                    loss = _AddScaledFn.apply(loss, mlm, 0.5)
                else:
                    loss = _AddScaledFn.apply(torch.zeros((), dtype=torch.float32, device=mlm.device), mlm, 0.5)

        return CM3POutput(loss=loss, logits_per_beatmap=logits_per_beatmap, logits_per_metadata=logits_per_metadata,
                          metadata_embeds=metadata_embeds, beatmap_embeds=beatmap_embeds, logits=logits,
                          metadata_model_output=metadata_outputs, beatmap_model_output=beatmap_outputs)


# ----------------------------------------------------------------------------------------------- stand-alone variants
class CM3PMetadataModelWithProjection(CM3PPreTrainedModel):
    """Metadata tower + projection, un-normalised (ref:cm3p/modeling_cm3p.py:1015-1065)."""

    config_class = CM3PMetadataConfig

    def __init__(self, config: CM3PMetadataConfig):
        super().__init__(config)
        self.metadata_model = CM3PMetadataTransformer(config)
        self.metadata_projection = nn.Linear(config.hidden_size, config.projection_dim, bias=False)
        self.post_init()

    def get_input_embeddings(self) -> nn.Module:
        return self.metadata_model.get_input_embeddings()

    def set_input_embeddings(self, value):
        self.metadata_model.set_input_embeddings(value)

    def forward(self, input_ids: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None, output_attentions=None,
                output_hidden_states=None) -> CM3PMetadataModelOutput:
        out = self.metadata_model(input_ids=input_ids, attention_mask=attention_mask, output_attentions=output_attentions,
                                  output_hidden_states=output_hidden_states)
        p = out.pooler_output
        emb = _ProjectFn.apply(p.reshape(-1, p.size(-1)), self.metadata_projection.weight).view(*p.shape[:-1], -1)
        return CM3PMetadataModelOutput(metadata_embeds=emb, last_hidden_state=out.last_hidden_state, hidden_states=out.hidden_states,
                                       attentions=out.attentions)


class CM3PBeatmapModelWithProjection(CM3PPreTrainedModel):
    """Beatmap tower + projection, un-normalised (ref:cm3p/modeling_cm3p.py:1068-1128)."""

    config_class = CM3PBeatmapConfig

    def __init__(self, config: CM3PBeatmapConfig):
        super().__init__(config)
        self.beatmap_model = CM3PBeatmapTransformer(config)
        self.beatmap_projection = nn.Linear(config.hidden_size, config.projection_dim, bias=False)
        self.post_init()

    def get_input_embeddings(self) -> nn.Module:
        return self.beatmap_model.get_input_embeddings()

    def set_input_embeddings(self, value):
        self.beatmap_model.set_input_embeddings(value)

    def forward(self, input_ids: Optional[Tensor] = None, input_features: Optional[Tensor] = None,
                attention_mask: Optional[Tensor] = None, position_ids: Optional[Tensor] = None,
                inputs_embeds: Optional[Tensor] = None, output_attentions=None, output_hidden_states=None) -> CM3PBeatmapModelOutput:
        out = self.beatmap_model(input_ids=input_ids, input_features=input_features, attention_mask=attention_mask,
                                 position_ids=position_ids, inputs_embeds=inputs_embeds, output_attentions=output_attentions,
                                 output_hidden_states=output_hidden_states)
        emb = _ProjectFn.apply(out.pooler_output, self.beatmap_projection.weight)
        return CM3PBeatmapModelOutput(beatmap_embeds=emb, pooler_output=out.pooler_output, last_hidden_state=out.last_hidden_state,
                                      hidden_states=out.hidden_states, attentions=out.attentions, audio_model_output=out.audio_model_output)


class _TakeRowsFn(torch.autograd.Function):
    """x [R, H] fp32 -> x[idx] (boolean-mask row selection of the sparse MLM head); backward scatters into zeros."""

    @staticmethod
    def forward(ctx, x: Tensor, idx: Tensor):
        ctx.idx, ctx.rows = idx, x.shape[0]
        return K.gather_rows(x.contiguous(), idx)

    @staticmethod
    def backward(ctx, dy: Tensor):
        return K.scatter_rows(dy.contiguous(), ctx.idx, ctx.rows), None


class CM3PForMaskedLM(CM3PPreTrainedModel):
    """Beatmap tower + MLM head + ForMaskedLM loss (ref:cm3p/modeling_cm3p.py:1241-1377), including `sparse_prediction`
    (head and decoder on the labelled positions only, :1349-1357)."""

    config_class = CM3PBeatmapConfig
    _tied_weights_keys = {"decoder.weight": "beatmap_model.encoder.embeddings.tok_embeddings.weight"}

    def __init__(self, config: CM3PBeatmapConfig):
        super().__init__(config)
        self.beatmap_model = CM3PBeatmapTransformer(config)
        self.head = CM3PPredictionHead(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=config.decoder_bias)
        self.sparse_prediction = bool(getattr(config, "sparse_prediction", False))
        self.sparse_pred_ignore_index = int(getattr(config, "sparse_pred_ignore_index", -100))
        self.post_init()

    def get_output_embeddings(self):
        return self.decoder

    def set_output_embeddings(self, new_embeddings: nn.Linear):
        self.decoder = new_embeddings

    def get_input_embeddings(self) -> nn.Module:
        return self.beatmap_model.get_input_embeddings()

    def forward(self, input_ids: Optional[Tensor] = None, input_features: Optional[Tensor] = None,
                attention_mask: Optional[Tensor] = None, sliding_window_mask=None, position_ids: Optional[Tensor] = None,
                inputs_embeds: Optional[Tensor] = None, labels: Optional[Tensor] = None, indices=None, cu_seqlens=None,
                max_seqlen=None, batch_size=None, seq_len=None, output_attentions=None, output_hidden_states=None, **kwargs):
        from transformers.modeling_outputs import MaskedLMOutput

        out = self.beatmap_model(input_ids=input_ids, input_features=input_features, attention_mask=attention_mask,
                                 position_ids=position_ids, inputs_embeds=inputs_embeds, indices=indices, cu_seqlens=cu_seqlens,
                                 output_attentions=output_attentions, output_hidden_states=output_hidden_states, output_pooler=False)
        hs = out.last_hidden_state
        Bq, Sq, Hq = hs.shape
        V = self.config.vocab_size
        rows = hs.reshape(Bq * Sq, Hq)
        sparse = self.sparse_prediction and labels is not None
        if sparse:  # keep the labelled tokens only (row selection by index; the count needs one host read, as in the reference)
            labels = labels.reshape(-1)
            idx = torch.nonzero(labels != self.sparse_pred_ignore_index).flatten()
            rows = _TakeRowsFn.apply(rows, idx)
            labels = labels[idx]
        head_args = (rows, self.head.dense.weight, self.head.dense.bias, self.head.norm.weight, self.decoder.weight, self.decoder.bias,
                     self.config.norm_eps)
        loss = None
        if labels is not None:
            lp, loss = _MLMHeadLossFn.apply(*head_args, labels, kwargs.get("num_items_in_batch"))
        else:
            lp = _MLMHeadFn.apply(*head_args)
        logits = lp[..., :V] if sparse else lp.view(Bq, Sq, -1)[..., :V]
        return MaskedLMOutput(loss=loss, logits=logits, hidden_states=out.hidden_states, attentions=out.attentions)


class _AddBiasFn(torch.autograd.Function):
    """x [rows, n] fp32 + bias [n]; backward: db = column sums."""

    @staticmethod
    def forward(ctx, x: Tensor, bias: Tensor):
        ctx.bdtype = bias.dtype
        return K.add_bias_(x.detach().clone().contiguous(), _f32(bias.detach()).contiguous())

    @staticmethod
    def backward(ctx, dy: Tensor):
        dy = dy.contiguous()
        return dy, K.colsum_f32(dy).to(ctx.bdtype)


class _PointwiseLossFn(torch.autograd.Function):
    """Mean MSELoss (kind 0) / BCEWithLogitsLoss (kind 1)."""

    @staticmethod
    def forward(ctx, x: Tensor, y: Tensor, kind: int):
        loss, dx = K.pointwise_loss(x.detach().contiguous(), y.detach().to(torch.float32).contiguous(), kind)
        ctx.dx = dx
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g: Tensor):
        return K.scale_by(ctx.dx, g.reshape(1).contiguous()), None, None


@dataclass
class BeatmapClassifierOutput(ModelOutput):
    loss: Optional[torch.FloatTensor] = None
    logits: Optional[torch.FloatTensor] = None
    hidden_states: Optional[tuple] = None
    attentions: Optional[tuple] = None


class CM3PForBeatmapClassification(CM3PPreTrainedModel):
    """Beatmap tower + linear classifier on the pooled output with the three HF loss flavours
    (ref:cm3p/modeling_cm3p.py:1137-1225)."""

    config_class = CM3PBeatmapConfig
    base_model_prefix = "beatmap_model"

    def __init__(self, config: CM3PBeatmapConfig):
        super().__init__(config)
        self.num_labels = config.num_labels
        self.beatmap_model = CM3PBeatmapTransformer(config)
        self.classifier = nn.Linear(config.hidden_size, config.num_labels) if config.num_labels > 0 else nn.Identity()
        self.post_init()

    def forward(self, input_ids: Optional[Tensor] = None, input_features: Optional[Tensor] = None,
                attention_mask: Optional[Tensor] = None, position_ids: Optional[Tensor] = None,
                inputs_embeds: Optional[Tensor] = None, labels: Optional[Tensor] = None, output_attentions=None,
                output_hidden_states=None) -> BeatmapClassifierOutput:
        out = self.beatmap_model(input_ids=input_ids, input_features=input_features, attention_mask=attention_mask,
                                 position_ids=position_ids, inputs_embeds=inputs_embeds, output_attentions=output_attentions,
                                 output_hidden_states=output_hidden_states)
        logits = out.pooler_output
        if self.num_labels > 0:
            logits = _AddBiasFn.apply(_ProjectFn.apply(out.pooler_output, self.classifier.weight), self.classifier.bias)
        loss = None
        if labels is not None:
            labels = labels.to(logits.device)
            if self.config.problem_type is None:  # the reference's (HF's) inference rule, :1198-1204
                if self.num_labels == 1:
                    self.config.problem_type = "regression"
                elif self.num_labels > 1 and labels.dtype in (torch.long, torch.int):
                    self.config.problem_type = "single_label_classification"
                else:
                    self.config.problem_type = "multi_label_classification"
            want = logits.shape[0] if self.config.problem_type == "single_label_classification" else logits.numel()
            if labels.numel() != want:  # (the torch losses raise on a size mismatch; the kernels would read past the labels)
                raise ValueError(f"labels have {labels.numel()} entries, expected {want}")
            if self.config.problem_type == "regression":
                loss = _PointwiseLossFn.apply(logits.reshape(-1), labels.reshape(-1), 0)
            elif self.config.problem_type == "single_label_classification":
                n = logits.shape[0]
                spec = [(0, n, self.num_labels, self.num_labels, 1, None, labels.reshape(-1).to(torch.int64).contiguous(), 1.0)]
                loss = _CrossEntropySumFn.apply(spec, logits)
            else:
                loss = _PointwiseLossFn.apply(logits.reshape(-1), labels.reshape(-1), 1)
        return BeatmapClassifierOutput(loss=loss, logits=logits, hidden_states=out.hidden_states, attentions=out.attentions)



def _register():
    for cfg, cls in ((CM3PMetadataConfig, CM3PMetadataModel), (CM3PBeatmapConfig, CM3PBeatmapModel), (CM3PConfig, CM3PModel)):
        try:
            AutoModel.register(cfg, cls)
        except ValueError:
            pass
    try:
        from transformers import AutoModelForMaskedLM, AutoModelForSequenceClassification

        AutoModelForMaskedLM.register(CM3PBeatmapConfig, CM3PForMaskedLM)
        AutoModelForSequenceClassification.register(CM3PBeatmapConfig, CM3PForBeatmapClassification)
    except ValueError:
        pass


_register()

__all__ = [
    "CM3PModel", "CM3PPreTrainedModel", "CM3PMetadataModel", "CM3PBeatmapModel", "CM3PMetadataTransformer",
    "CM3PBeatmapTransformer", "CM3PAudioEncoder", "CM3POutput", "CM3PForMaskedLM", "CM3PForBeatmapClassification",
    "CM3PMetadataModelWithProjection", "CM3PBeatmapModelWithProjection", "cm3p_loss_hip",
]
