"""Seeded synthetic batches of the shapes BASELINE.json names: ids ~ U{3 .. vocab-4} (avoiding pad / bos /
eos and the three audio ids at the top of the beatmap vocabulary), all-ones masks or right padding with lengths
~ U{S/2 .. S}; the audio-fused layout puts [AUDIO_BOS][AUDIO]xN[AUDIO_EOS] first (ref:cm3p/tokenization_cm3p.py:218-220)."""
from __future__ import annotations

from typing import Optional

import torch


def synthetic_batch(config, B: int, S: int, L: int, seed: int = 1234, padded: bool = False, audio_T: Optional[int] = None) -> dict:
    b, m = config.beatmap_config, config.metadata_config
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, b.vocab_size - 3, (B, S), generator=g, dtype=torch.int64)
    mask = torch.ones(B, S, dtype=torch.int64)
    mids = torch.randint(3, m.vocab_size - 3, (B, L), generator=g, dtype=torch.int64)
    mmask = torch.ones(B, L, dtype=torch.int64)
    if padded:
        lens = torch.randint(S // 2, S + 1, (B,), generator=g)
        mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
        ids = ids * mask
    out = dict(input_ids=ids, attention_mask=mask, metadata_ids=mids, metadata_attention_mask=mmask)
    if audio_T is not None:
        n = audio_T // 8
        ids[:, 0] = b.audio_sos_token_id
        ids[:, 1:1 + n] = b.audio_token_id
        ids[:, 1 + n] = b.audio_eos_token_id
        mask[:, : n + 2] = 1
        out["input_features"] = torch.randn(B, b.audio_config.n_mels, audio_T, generator=g)
    return out
