"""CPU oracle: a restatement of CM3P's contrastive hot path in plain torch ops.

TEST INFRASTRUCTURE ONLY.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this module; the product (`cm3p_amd/`) never does and has no CPU fallback.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here against fixtures in
`tests/golden/` that were produced by running the reference itself (`tests/golden/make_golden.py`:
reference wrapper imported unmodified from /root/reference + installed transformers 5.15.0, CPU, fp32,
attn_implementation="sdpa").

Conventions: `ref:` = /root/reference/, `TF:` = the installed third-party transformers 5.15.0 that
carries the encoder arithmetic (pinned 4.55.0 by ref:Dockerfile:4; see SURVEY.md §8c for the skew).
A model is a flat `dict[str, Tensor]` using the reference's state-dict keys plus a plain config dict;
no `transformers` import is needed to run the oracle.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------------
# configuration helpers (defaults follow ref:cm3p/configuration_cm3p.py:10-335)
# --------------------------------------------------------------------------------------------------
_METADATA_DEFAULTS = dict(
    cls_embed=True, vocab_size=1000, hidden_size=256, intermediate_size=512, num_hidden_layers=6,
    num_attention_heads=4, norm_eps=1e-5, global_rope_theta=10000.0, global_attn_every_n_layers=1,
    local_attention=128, local_rope_theta=10000.0,
)
_AUDIO_DEFAULTS = dict(
    hidden_size=512, intermediate_size=1024, num_hidden_layers=6, num_attention_heads=8, norm_eps=1e-5,
    global_rope_theta=160000.0, global_attn_every_n_layers=3, local_attention=128, local_rope_theta=10000.0,
    projector_intermediate_size=2048, projector_dim=768, n_mels=80,
)
_BEATMAP_DEFAULTS = dict(
    cls_embed=True, vocab_size=3167, hidden_size=768, intermediate_size=1152, num_hidden_layers=22,
    num_attention_heads=12, norm_eps=1e-5, global_rope_theta=160000.0, global_attn_every_n_layers=3,
    local_attention=128, local_rope_theta=10000.0, audio_token_id=3166,
)


def resolve_config(cfg: Optional[dict]) -> dict:
    """Fill a (possibly partial) CM3PConfig kwargs dict with the reference's defaults."""
    cfg = dict(cfg or {})
    b = dict(_BEATMAP_DEFAULTS)
    b.update(cfg.get("beatmap_config") or {})
    a = dict(_AUDIO_DEFAULTS)
    a.update(b.get("audio_config") or {})
    b["audio_config"] = a
    m = dict(_METADATA_DEFAULTS)
    m.update(cfg.get("metadata_config") or {})
    return dict(
        projection_dim=cfg.get("projection_dim", 512),
        logit_scale_init_value=cfg.get("logit_scale_init_value", 2.6592),
        beatmap_config=b,
        metadata_config=m,
    )


def layer_is_global(cfg: dict, i: int) -> bool:
    """TF:models/modernbert/configuration_modernbert.py:115-120: layer i is global iff i % n == 0."""
    return i % cfg["global_attn_every_n_layers"] == 0


# --------------------------------------------------------------------------------------------------
# encoder pieces (TF:models/modernbert/modeling_modernbert.py)
# --------------------------------------------------------------------------------------------------
def layer_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """Bias-free LayerNorm, `nn.LayerNorm(H, eps, bias=False)` (TF:...modeling_modernbert.py:61,312-314,420)."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * weight


def rope_inv_freq(theta: float, head_dim: int) -> Tensor:
    """TF:...modeling_modernbert.py:141: inv_freq[k] = 1 / theta^(2k/dim), k < dim/2, fp32."""
    return 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))


def rope_cos_sin(position_ids: Tensor, theta: float, head_dim: int) -> tuple[Tensor, Tensor]:
    """TF:...modeling_modernbert.py:146-163: fp32 cos/sin of pos*inv_freq, duplicated to head_dim."""
    inv_freq = rope_inv_freq(theta, head_dim)
    freqs = position_ids[:, :, None].float() * inv_freq[None, None, :]  # (Bp, S, d/2)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x: Tensor) -> Tensor:
    """TF:...modeling_modernbert.py:188-192 (half-split / NeoX convention)."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor) -> tuple[Tensor, Tensor]:
    """TF:...modeling_modernbert.py:196-219: rotate in fp32, cast back.  q,k: (B, nh, S, d)."""
    dt = q.dtype
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    q2 = q.float() * cos + rotate_half(q.float()) * sin
    k2 = k.float() * cos + rotate_half(k.float()) * sin
    return q2.to(dt), k2.to(dt)


def attention_allowed(attention_mask: Optional[Tensor], B: int, S: int, window: Optional[int]) -> Optional[Tensor]:
    """Bool (B,1,S,S) mask, True = attend, or None when the reference skips mask creation.

    attend(b,q,kv) = padding[b,kv] AND (global OR |q-kv| <= window)
    (TF:masking_utils.py:141-151 overlay, :168-179 padding term, :308-336 skip rule:
    no mask when nothing is padded and (global, or kv_len < window)).
    `window` is config.sliding_window = local_attention // 2 for local layers, None for global.
    """
    no_pad = attention_mask is None or bool(attention_mask.bool().all())
    if no_pad and (window is None or S < window):
        return None
    idx = torch.arange(S)
    allowed = torch.ones(S, S, dtype=torch.bool)
    if window is not None:
        allowed = (idx[:, None] - idx[None, :]).abs() <= window
    allowed = allowed[None, None].expand(B, 1, S, S)
    if attention_mask is not None:
        allowed = allowed & attention_mask.bool()[:, None, None, :]
    return allowed


def sdpa(q: Tensor, k: Tensor, v: Tensor, allowed: Optional[Tensor], scale: float, eager: bool = False) -> Tensor:
    """softmax(q k^T * scale + mask) v, non-causal (TF:integrations/sdpa_attention.py:153-163).

    Rows with no allowed key come out as exact zeros (what torch's CPU SDPA returns);
    the eager branch reproduces that explicitly.
    """
    if not eager:
        return F.scaled_dot_product_attention(q, k, v, attn_mask=allowed, dropout_p=0.0, scale=scale, is_causal=False)
    s = (q @ k.transpose(-1, -2)) * scale
    if allowed is not None:
        s = s.masked_fill(~allowed, float("-inf"))
    p = torch.softmax(s.float(), dim=-1).to(q.dtype)
    if allowed is not None:
        p = torch.where(allowed.any(dim=-1, keepdim=True), p, torch.zeros_like(p))
    return p @ v


def eager_attention_probs(q: Tensor, k: Tensor, allowed: Optional[Tensor], scale: float) -> Tensor:
    """What `output_attentions=True` returns per layer: TF then runs eager_attention_forward
    (TF:models/modernbert/modeling_modernbert.py:133-170): softmax(scale * q k^T + additive mask, fp32).  The additive mask is
    finfo.min (finite) on the invisible keys, so a row with no visible key comes out uniform over ALL keys, not NaN."""
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    if allowed is not None:
        s = s + torch.where(allowed, 0.0, torch.finfo(torch.float32).min)
    return torch.softmax(s, dim=-1)


def encoder_layer(x: Tensor, sd: dict, prefix: str, i: int, cfg: dict, cos_sin, allowed, eager: bool, attn_out: Optional[list] = None) -> Tensor:
    """ModernBertEncoderLayer.forward (TF:...modeling_modernbert.py:318-333) with attention :262-301, MLP :89-91."""
    nh = cfg["num_attention_heads"]
    H = cfg["hidden_size"]
    d = H // nh
    eps = cfg["norm_eps"]
    p = f"{prefix}layers.{i}."
    B, S, _ = x.shape

    h = x if i == 0 else layer_norm(x, sd[p + "attn_norm.weight"], eps)  # layer 0: Identity (:309-310)
    qkv = F.linear(h, sd[p + "attn.Wqkv.weight"]).view(B, S, 3, nh, d)
    q, k, v = (t.transpose(1, 2) for t in qkv.unbind(dim=2))
    q, k = apply_rope(q, k, *cos_sin)
    if attn_out is not None:
        attn_out.append(eager_attention_probs(q, k, allowed, d ** -0.5))
    a = sdpa(q, k, v, allowed, d ** -0.5, eager).transpose(1, 2).reshape(B, S, H)
    x = x + F.linear(a, sd[p + "attn.Wo.weight"])

    h = layer_norm(x, sd[p + "mlp_norm.weight"], eps)
    hi, gate = F.linear(h, sd[p + "mlp.Wi.weight"]).chunk(2, dim=-1)
    x = x + F.linear(F.gelu(hi) * gate, sd[p + "mlp.Wo.weight"])  # exact-erf GELU (ACT2FN["gelu"])
    return x


def encoder(sd: dict, prefix: str, cfg: dict, *, input_ids: Optional[Tensor] = None,
            inputs_embeds: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None,
            position_ids: Optional[Tensor] = None, eager: bool = False, collect: Optional[list] = None,
            attn_out: Optional[list] = None) -> Tensor:
    """ModernBertModel.forward (TF:...modeling_modernbert.py:434-478).  attn_out: receives every layer's eager attention
    probabilities (B, nh, S, S) - the `attentions` of an output_attentions=True call."""
    if inputs_embeds is None:
        inputs_embeds = F.embedding(input_ids, sd[prefix + "embeddings.tok_embeddings.weight"])
    B, S, H = inputs_embeds.shape
    d = H // cfg["num_attention_heads"]
    if position_ids is None:
        position_ids = torch.arange(S).unsqueeze(0)
    x = layer_norm(inputs_embeds, sd[prefix + "embeddings.norm.weight"], cfg["norm_eps"])
    if collect is not None:
        collect.append(x)
    window = cfg["local_attention"] // 2
    masks = {True: attention_allowed(attention_mask, B, S, None), False: attention_allowed(attention_mask, B, S, window)}
    rope = {
        True: rope_cos_sin(position_ids, cfg["global_rope_theta"], d),
        False: rope_cos_sin(position_ids, cfg["local_rope_theta"], d),
    }
    for i in range(cfg["num_hidden_layers"]):
        g = layer_is_global(cfg, i)
        x = encoder_layer(x, sd, prefix, i, cfg, tuple(t.to(x.dtype) for t in rope[g]), masks[g], eager, attn_out)
        if collect is not None:
            collect.append(x)
    return layer_norm(x, sd[prefix + "final_norm.weight"], cfg["norm_eps"])


# --------------------------------------------------------------------------------------------------
# towers (ref:cm3p/modeling_cm3p.py)
# --------------------------------------------------------------------------------------------------
def pool(last_hidden_state: Tensor, attention_mask: Optional[Tensor], cls_embed: bool) -> Tensor:
    """ref:cm3p/modeling_cm3p.py:385-396 / :631-642: index-0 pooling, or masked mean in fp32."""
    if cls_embed:
        return last_hidden_state[..., 0, :]
    if attention_mask is None:
        return last_hidden_state.mean(dim=-2)
    m = attention_mask.unsqueeze(-1).float()
    pooled = (last_hidden_state * m).sum(dim=-2) / torch.clamp(m.sum(dim=-2), min=1e-9)
    return pooled.to(last_hidden_state.dtype)


def audio_encoder(sd: dict, prefix: str, cfg: dict, input_features: Tensor, eager: bool = False) -> Tensor:
    """CM3PAudioEncoder.forward + projector (ref:cm3p/modeling_cm3p.py:470-528) -> (B*T/8, projector_dim)."""
    x = F.gelu(F.conv1d(input_features, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1).contiguous()
    pos = torch.arange(x.size(1)).unsqueeze(0).repeat(x.size(0), 1)
    h = encoder(sd, prefix + "encoder.", cfg, inputs_embeds=x, position_ids=pos, eager=eager)
    h = h.reshape(-1, cfg["projector_intermediate_size"])
    h = F.gelu(F.linear(h, sd[prefix + "multi_modal_projector.linear_1.weight"]))
    return F.linear(h, sd[prefix + "multi_modal_projector.linear_2.weight"])


def beatmap_tower(sd: dict, cfg: dict, input_ids: Tensor, attention_mask: Optional[Tensor],
                  input_features: Optional[Tensor] = None, eager: bool = False, collect: Optional[list] = None):
    """CM3PBeatmapTransformer.forward (ref:cm3p/modeling_cm3p.py:547-650) -> (last_hidden, pooled, audio_embeds)."""
    prefix = "beatmap_model."
    emb = F.embedding(input_ids, sd[prefix + "encoder.embeddings.tok_embeddings.weight"])
    audio_embeds = None
    if input_features is not None:
        audio_embeds = audio_encoder(sd, prefix + "audio_encoder.", cfg["audio_config"], input_features, eager)
        emb = emb.clone()
        # boolean-mask assignment in row-major (b, s) order; count must match exactly (:603-605)
        emb[input_ids == cfg["audio_token_id"]] = audio_embeds.to(emb.dtype)
    h = encoder(sd, prefix + "encoder.", cfg, inputs_embeds=emb, attention_mask=attention_mask, eager=eager, collect=collect)
    return h, pool(h, attention_mask, cfg["cls_embed"]), audio_embeds


def metadata_tower(sd: dict, cfg: dict, input_ids: Tensor, attention_mask: Optional[Tensor], eager: bool = False):
    """CM3PMetadataTransformer.forward (ref:cm3p/modeling_cm3p.py:315-403); 3-D (B,V,L) is flattened :351-357."""
    is_3d = input_ids.dim() == 3
    B0 = input_ids.size(0)
    ids2, am2 = input_ids, attention_mask
    if is_3d:
        ids2 = input_ids.reshape(-1, input_ids.size(-1))
        am2 = attention_mask.reshape(-1, attention_mask.size(-1)) if attention_mask is not None else None
    h = encoder(sd, "metadata_model.encoder.", cfg, input_ids=ids2, attention_mask=am2, eager=eager)
    if is_3d:
        h = h.view(B0, -1, h.size(-2), h.size(-1))
    return h, pool(h, attention_mask, cfg["cls_embed"])


# --------------------------------------------------------------------------------------------------
# contrastive head (ref:cm3p/modeling_cm3p.py:27-62, 958-985)
# --------------------------------------------------------------------------------------------------
def l2_normalize(x: Tensor) -> Tensor:
    """x / (sum x^2)^0.5, no eps (ref:cm3p/modeling_cm3p.py:54-62,960,972)."""
    return x / torch.pow(torch.sum(torch.pow(x, 2), dim=-1, keepdim=True), 0.5)


def true_variation_index(metadata_variation_classes: Tensor) -> Tensor:
    """ref:cm3p/modeling_cm3p.py:40: first slot whose class is 0 (0 if none)."""
    return (metadata_variation_classes == 0).int().argmax(dim=1)


def cm3p_loss(similarity: Tensor, metadata_variation_classes: Optional[Tensor] = None) -> Tensor:
    """ref:cm3p/modeling_cm3p.py:33-51."""
    if similarity.dim() == 3:
        Bm, V, Bb = similarity.shape
        assert Bm == Bb
        t = true_variation_index(metadata_variation_classes)
        metadata_loss = F.cross_entropy(similarity[torch.arange(Bm), t], torch.arange(Bm))
        bsim = similarity.permute(2, 0, 1).reshape(Bb, -1)
        target = torch.arange(0, bsim.size(1), V) + t
        beatmap_loss = F.cross_entropy(bsim, target)
    else:
        n = similarity.size(0)
        metadata_loss = F.cross_entropy(similarity, torch.arange(n))
        beatmap_loss = F.cross_entropy(similarity.t(), torch.arange(n))
    return (metadata_loss + beatmap_loss) / 2.0


def contrastive_head(sd: dict, beatmap_pooled: Tensor, metadata_pooled: Tensor,
                     metadata_variation_classes: Optional[Tensor] = None):
    """Projections, L2 norm, logits and loss (ref:cm3p/modeling_cm3p.py:958-985)."""
    be = l2_normalize(F.linear(beatmap_pooled, sd["beatmap_projection.weight"]))
    me = l2_normalize(F.linear(metadata_pooled, sd["metadata_projection.weight"]))
    logits_per_metadata = torch.matmul(me, be.t()) * sd["logit_scale"].exp()
    loss = cm3p_loss(logits_per_metadata, metadata_variation_classes)
    return dict(loss=loss, logits_per_metadata=logits_per_metadata, metadata_embeds=me, beatmap_embeds=be)


def mlm_head(sd: dict, cfg: dict, last_hidden_state: Tensor) -> Tensor:
    """decoder(head(h)): dense -> GELU -> LayerNorm(no bias) -> Linear(+bias) (ref:cm3p/modeling_cm3p.py:991,1229-1238)."""
    h = F.linear(last_hidden_state, sd["head.dense.weight"], sd.get("head.dense.bias"))
    h = layer_norm(F.gelu(h), sd["head.norm.weight"], cfg["norm_eps"])
    return F.linear(h, sd["decoder.weight"], sd.get("decoder.bias"))


def masked_lm_loss(logits: Tensor, labels: Tensor, vocab_size: int, num_items_in_batch=None) -> Tensor:
    """ForMaskedLMLoss (TF:loss/loss_utils.py:32-46,74-91)."""
    lg, lb = logits.float().view(-1, vocab_size), labels.view(-1)
    if num_items_in_batch is None:
        return F.cross_entropy(lg, lb, ignore_index=-100, reduction="mean")
    return F.cross_entropy(lg, lb, ignore_index=-100, reduction="sum") / num_items_in_batch


def forward(sd: dict, cfg: dict, *, input_ids: Tensor, metadata_ids: Tensor, attention_mask: Optional[Tensor] = None,
            metadata_attention_mask: Optional[Tensor] = None, input_features: Optional[Tensor] = None,
            metadata_variation_classes: Optional[Tensor] = None, labels: Optional[Tensor] = None, eager: bool = False,
            collect: Optional[list] = None) -> dict:
    """CM3PModel.forward, contrastive branch (ref:cm3p/modeling_cm3p.py:849-1012)."""
    cfg = resolve_config(cfg)
    if metadata_ids.dim() == 3 and metadata_variation_classes is None:
        raise ValueError("When providing multiple metadata variations, metadata_variation_classes must be provided "
                         "in order to compute loss correctly.")  # ref:cm3p/modeling_cm3p.py:904-905
    bh, bp, audio_embeds = beatmap_tower(sd, cfg["beatmap_config"], input_ids, attention_mask, input_features, eager, collect)
    mh, mp = metadata_tower(sd, cfg["metadata_config"], metadata_ids, metadata_attention_mask, eager)
    out = contrastive_head(sd, bp, mp, metadata_variation_classes)
    lpm = out["logits_per_metadata"]
    out["logits_per_beatmap"] = lpm.permute(2, 0, 1) if lpm.dim() == 3 else lpm.t()
    out.update(beatmap_last_hidden_state=bh, beatmap_pooler_output=bp, metadata_last_hidden_state=mh,
               metadata_pooler_output=mp, audio_embeds=audio_embeds)
    if "decoder.weight" in sd:  # has_decoder_head: logits + 0.5 * masked-LM loss (ref:cm3p/modeling_cm3p.py:987-996)
        out["logits"] = mlm_head(sd, cfg["beatmap_config"], bh)
        if labels is not None:
            out["mlm_loss"] = masked_lm_loss(out["logits"], labels, cfg["beatmap_config"]["vocab_size"])
            out["loss"] = out["loss"] + 0.5 * out["mlm_loss"]
    return out


# --------------------------------------------------------------------------------------------------
# synthetic weights / batches for the bench's cpu_baseline leg and the GPU parity tests
# --------------------------------------------------------------------------------------------------
def init_state_dict(cfg: dict, seed: int = 0, with_audio: bool = True, dtype=torch.float32) -> dict:
    """Random weights with the reference's state-dict keys (SURVEY §5 checkpoint row) and init scales
    (TF:...modeling_modernbert.py:353-408: trunc-normal std 0.02 'in', 0.02/sqrt(2L) 'out';
    ref:cm3p/modeling_cm3p.py:262-297 for conv / projections)."""
    cfg = resolve_config(cfg)
    g = torch.Generator().manual_seed(seed)
    sd: dict[str, Tensor] = {}

    def tn(shape, std):
        t = torch.empty(shape, dtype=torch.float32)
        torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=g)
        return t.to(dtype)

    def enc(prefix, c, embeddings=True):
        H, I, L = c["hidden_size"], c["intermediate_size"], c["num_hidden_layers"]
        out_std = 0.02 / math.sqrt(2.0 * L)
        if embeddings:
            sd[prefix + "embeddings.tok_embeddings.weight"] = tn((c["vocab_size"], H), 0.02)
        else:
            sd[prefix + "embeddings.tok_embeddings.weight"] = tn((1, H), 0.02)
        sd[prefix + "embeddings.norm.weight"] = torch.ones(H, dtype=dtype)
        for i in range(L):
            p = f"{prefix}layers.{i}."
            if i > 0:
                sd[p + "attn_norm.weight"] = torch.ones(H, dtype=dtype)
            sd[p + "attn.Wqkv.weight"] = tn((3 * H, H), 0.02)
            sd[p + "attn.Wo.weight"] = tn((H, H), out_std)
            sd[p + "mlp_norm.weight"] = torch.ones(H, dtype=dtype)
            sd[p + "mlp.Wi.weight"] = tn((2 * I, H), 0.02)
            sd[p + "mlp.Wo.weight"] = tn((H, I), out_std)
        sd[prefix + "final_norm.weight"] = torch.ones(H, dtype=dtype)

    b, m = cfg["beatmap_config"], cfg["metadata_config"]
    enc("beatmap_model.encoder.", b)
    enc("metadata_model.encoder.", m)
    if with_audio:
        a = b["audio_config"]
        ap = "beatmap_model.audio_encoder."
        sd[ap + "conv1.weight"] = (torch.randn(a["hidden_size"], a["n_mels"], 3, generator=g) * 0.02).to(dtype)
        sd[ap + "conv1.bias"] = torch.zeros(a["hidden_size"], dtype=dtype)
        sd[ap + "conv2.weight"] = (torch.randn(a["hidden_size"], a["hidden_size"], 3, generator=g) * 0.02).to(dtype)
        sd[ap + "conv2.bias"] = torch.zeros(a["hidden_size"], dtype=dtype)
        enc(ap + "encoder.", a, embeddings=False)
        sd[ap + "multi_modal_projector.linear_1.weight"] = (
            torch.randn(a["projector_dim"], a["projector_intermediate_size"], generator=g) * 0.02).to(dtype)
        sd[ap + "multi_modal_projector.linear_2.weight"] = (
            torch.randn(a["projector_dim"], a["projector_dim"], generator=g) * 0.02).to(dtype)
    P = cfg["projection_dim"]
    sd["beatmap_projection.weight"] = (torch.randn(P, b["hidden_size"], generator=g) * b["hidden_size"] ** -0.5).to(dtype)
    sd["metadata_projection.weight"] = (torch.randn(P, m["hidden_size"], generator=g) * m["hidden_size"] ** -0.5).to(dtype)
    sd["logit_scale"] = torch.tensor(cfg["logit_scale_init_value"], dtype=dtype)
    return sd


def synthetic_batch(cfg: dict, B: int, S: int, L: int, seed: int = 1234, padded: bool = False,
                    audio_T: Optional[int] = None) -> dict:
    """SURVEY §8(d) synthetic inputs: ids ~ U{3..vocab-4}, all-ones masks (or right-padded, length ~ U{S/2..S})."""
    cfg = resolve_config(cfg)
    b, m = cfg["beatmap_config"], cfg["metadata_config"]
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, b["vocab_size"] - 3, (B, S), generator=g, dtype=torch.int64)
    mask = torch.ones(B, S, dtype=torch.int64)
    mids = torch.randint(3, m["vocab_size"] - 3, (B, L), generator=g, dtype=torch.int64)
    mmask = torch.ones(B, L, dtype=torch.int64)
    if padded:
        lens = torch.randint(S // 2, S + 1, (B,), generator=g)
        mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
        ids = ids * mask
    out = dict(input_ids=ids, attention_mask=mask, metadata_ids=mids, metadata_attention_mask=mmask)
    if audio_T is not None:
        n = audio_T // 8
        ids[:, 0] = b.get("audio_sos_token_id", 3164)
        ids[:, 1:1 + n] = b["audio_token_id"]
        ids[:, 1 + n] = b.get("audio_eos_token_id", 3165)
        mask[:, : n + 2] = 1
        out["input_features"] = torch.randn(B, b["audio_config"]["n_mels"], audio_T, generator=g)
    return out
