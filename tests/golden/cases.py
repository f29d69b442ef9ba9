"""Golden-case definitions shared by the fixture generator and the parity tests.

Pure data + seeded synthetic input builders; nothing here imports the reference.
Token-id ranges follow SURVEY.md §8(d): ids avoid pad 0 / bos 1 / eos 2 and the three audio ids at
the top of the beatmap vocabulary (ref:cm3p/configuration_cm3p.py:187-189).
"""
from __future__ import annotations

import copy

import torch

# head_dim is 64 in every tower of the default config (768/12, 256/4, 512/8), which is what the HIP
# attention kernels are built for.  "d64" cases therefore use hidden = 64 * heads.
_BEATMAP_D64 = dict(
    vocab_size=200, hidden_size=128, intermediate_size=192, num_hidden_layers=4, num_attention_heads=2,
    max_position_embeddings=1024, global_attn_every_n_layers=3, local_attention=128,
    global_rope_theta=160000.0, local_rope_theta=10000.0,
    audio_sos_token_id=197, audio_eos_token_id=198, audio_token_id=199, cls_embed=True,
    audio_config=dict(
        hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
        max_position_embeddings=1024, global_attn_every_n_layers=3, local_attention=128,
        projector_intermediate_size=256, projector_dim=128, n_mels=16,
    ),
)
_METADATA_D64 = dict(
    vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
    max_position_embeddings=128, global_attn_every_n_layers=1, local_attention=128, cls_embed=True,
)

# BASELINE.json configs[0] / SURVEY §8(d) C1: H=64, I=96, L=3, 4 heads (head_dim 16), P=32.  CPU only.
_BEATMAP_C1 = dict(
    vocab_size=200, hidden_size=64, intermediate_size=96, num_hidden_layers=3, num_attention_heads=4,
    max_position_embeddings=1024, global_attn_every_n_layers=3, local_attention=128,
    audio_sos_token_id=197, audio_eos_token_id=198, audio_token_id=199, cls_embed=True,
    audio_config=dict(
        hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
        max_position_embeddings=1024, projector_intermediate_size=256, projector_dim=64, n_mels=16,
    ),
)
_METADATA_C1 = dict(
    vocab_size=100, hidden_size=64, intermediate_size=96, num_hidden_layers=3, num_attention_heads=4,
    max_position_embeddings=128, cls_embed=True,
)


def _cfg(beatmap, metadata, projection_dim=32, **over):
    c = dict(
        projection_dim=projection_dim,
        beatmap_config=copy.deepcopy(beatmap),
        metadata_config=copy.deepcopy(metadata),
    )
    c["beatmap_config"]["projection_dim"] = projection_dim
    c["metadata_config"]["projection_dim"] = projection_dim
    for k, v in over.items():
        tower, key = k.split("__")
        c[tower][key] = v
    return c


CASES = {
    # name: (config kwargs, batch, beatmap seq, metadata seq, options)
    "c1_tiny_nopad": dict(cfg=_cfg(_BEATMAP_C1, _METADATA_C1), B=4, S=512, L=32, pad=False),
    "d64_cls_nopad": dict(cfg=_cfg(_BEATMAP_D64, _METADATA_D64), B=4, S=512, L=32, pad=False),
    "d64_mean_pad": dict(
        cfg=_cfg(_BEATMAP_D64, _METADATA_D64, beatmap_config__cls_embed=False, metadata_config__cls_embed=False),
        B=4, S=512, L=32, pad=True,
    ),
    # one row is so short that local-layer queries > 64 past the last valid key see no key at all
    "d64_mean_longpad": dict(
        cfg=_cfg(_BEATMAP_D64, _METADATA_D64, beatmap_config__cls_embed=False, metadata_config__cls_embed=False),
        B=3, S=384, L=32, pad=True, short_row=100,
    ),
    # S < 64: the sdpa mask-skip rule (TF:masking_utils.py:308-336) applies to local layers too
    "d64_short_seq": dict(cfg=_cfg(_BEATMAP_D64, _METADATA_D64), B=4, S=48, L=16, pad=False),
    # odd, non-tile-aligned lengths with padding
    "d64_ragged": dict(
        cfg=_cfg(_BEATMAP_D64, _METADATA_D64, beatmap_config__cls_embed=False),
        B=5, S=203, L=19, pad=True,
    ),
    # 3-D metadata variations (B, V, L) with variation classes
    "d64_variations": dict(
        cfg=_cfg(_BEATMAP_D64, _METADATA_D64, metadata_config__cls_embed=False),
        B=4, S=256, L=24, pad=True, V=3,
    ),
    # MLM head: has_decoder_head + loss_type "ForMaskedLM" (the v7 recipe), labels = ids at ~15 % of the positions, else -100
    "d64_mlm": dict(
        cfg=dict(_cfg(_BEATMAP_D64, _METADATA_D64, beatmap_config__cls_embed=False), has_decoder_head=True, loss_type="ForMaskedLM"),
        B=3, S=160, L=16, pad=True, mlm=True,
    ),
    # audio-fused: input_features (B, n_mels, T) with T/8 placeholders per row
    "d64_audio": dict(cfg=_cfg(_BEATMAP_D64, _METADATA_D64), B=3, S=256, L=16, pad=True, audio_T=320),
}


def make_inputs(name: str) -> dict[str, torch.Tensor]:
    """Seeded synthetic batch for a case (int64 ids / masks, fp32 features)."""
    case = CASES[name]
    cfg = case["cfg"]
    B, S, L = case["B"], case["S"], case["L"]
    g = torch.Generator().manual_seed(1234 + sum(map(ord, name)))
    bv = cfg["beatmap_config"]["vocab_size"]
    mv = cfg["metadata_config"]["vocab_size"]
    out: dict[str, torch.Tensor] = {}

    ids = torch.randint(3, bv - 3, (B, S), generator=g, dtype=torch.int64)
    mask = torch.ones(B, S, dtype=torch.int64)
    if case["pad"]:
        lens = torch.randint(S // 2, S + 1, (B,), generator=g)
        lens[0] = S  # keep one full row
        if "short_row" in case:
            lens[1] = case["short_row"]
        for b in range(B):
            mask[b, lens[b]:] = 0
            ids[b, lens[b]:] = 0

    if "audio_T" in case:
        T = case["audio_T"]
        n_audio = T // 8  # conv stride 2, then 4 frames per token (ref:cm3p/modeling_cm3p.py:518)
        bc = cfg["beatmap_config"]
        ids[:, 0] = bc["audio_sos_token_id"]
        ids[:, 1:1 + n_audio] = bc["audio_token_id"]
        ids[:, 1 + n_audio] = bc["audio_eos_token_id"]
        mask[:, : n_audio + 2] = 1
        out["input_features"] = torch.randn(B, bc["audio_config"]["n_mels"], T, generator=g)

    out["input_ids"] = ids
    out["attention_mask"] = mask

    if case.get("mlm"):
        pick = (torch.rand(B, S, generator=g) < 0.15) & mask.bool()
        pick[0, 1] = True  # at least one label
        out["labels"] = torch.where(pick, ids, torch.full_like(ids, -100))

    V = case.get("V")
    mshape = (B, V, L) if V else (B, L)
    mids = torch.randint(3, mv - 3, mshape, generator=g, dtype=torch.int64)
    mmask = torch.ones(mshape, dtype=torch.int64)
    if case["pad"]:
        mlens = torch.randint(L // 2, L + 1, mshape[:-1], generator=g)
        idx = torch.arange(L).expand(mshape)
        mmask = (idx < mlens.unsqueeze(-1)).to(torch.int64)
        mids = mids * mmask
    out["metadata_ids"] = mids
    out["metadata_attention_mask"] = mmask
    if V:
        # class 0 = the original metadata (exactly one per row, at a varying slot); >0 variations; -1 padding
        classes = torch.randint(1, 4, (B, V), generator=g, dtype=torch.int64)
        slot = torch.randint(0, V, (B,), generator=g)
        classes[torch.arange(B), slot] = 0
        classes[B - 1, (slot[B - 1] + 1) % V] = -1
        out["metadata_variation_classes"] = classes
    return out
