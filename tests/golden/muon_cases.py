"""Muon golden-case definitions shared by the fixture generator and the tests (pure data; imports nothing of the reference)."""
from __future__ import annotations

import torch

# name -> shape.  Names drive the AdamW/Muon routing of ref:train.py:331-340; shapes cover tall / wide / square, a group of
# two same-shaped weights, a 3-D conv weight (flattened to 24 x 30), extents that are not multiples of 8 or 4, a 1-D and a
# 0-D parameter, an embedding (AdamW by name) and a matrix with >= 10000 rows (AdamW by the rule at ref:utils/muon_utils.py:104).
SHAPES = {
    "enc.layers.0.attn.Wqkv.weight": (96, 32),
    "enc.layers.0.attn.Wo.weight": (32, 32),
    "enc.layers.0.mlp.Wi.weight": (96, 32),
    "enc.layers.0.mlp.Wo.weight": (32, 48),
    "enc.layers.1.attn.Wo.weight": (32, 32),
    "audio.conv1.weight": (24, 10, 3),
    "odd.weight": (13, 21),
    "beatmap_projection.weight": (16, 32),
    "enc.embeddings.tok_embeddings.weight": (50, 32),
    "enc.final_norm.weight": (32,),
    "audio.conv1.bias": (24,),
    "logit_scale": (),
    "many_rows.weight": (10000, 8),
}

VARIANTS = {
    # the v7 recipe's shape: nesterov, 6 iterations, adamw_lr = lr / 4 (ref:train.py:344-352); the lr moves at step 3 like a scheduler
    "main": dict(lr=0.02, momentum=0.95, nesterov=True, ns_steps=6, adamw_lr=0.005, adamw_betas=(0.9, 0.95), adamw_eps=1e-8,
                 adamw_wd=0.01, lrs=(0.02, 0.02, 0.012)),
    # plain momentum, 5 iterations, lerp weights >= 0.5 (the other branch of torch.lerp)
    "plain": dict(lr=0.01, momentum=0.9, nesterov=False, ns_steps=5, adamw_lr=0.01, adamw_betas=(0.4, 0.3), adamw_eps=1e-6,
                  adamw_wd=0.0, lrs=(0.01, 0.01, 0.01)),
}
N_STEPS = 3


def initial_params() -> dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(11)
    return {k: torch.randn(s, generator=g) * 0.5 for k, s in SHAPES.items()}


def gradients(step: int) -> dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(100 + step)
    out = {k: torch.randn(s, generator=g) * 0.1 for k, s in SHAPES.items()}
    if step == 1:
        out["enc.layers.1.attn.Wo.weight"] = None  # a parameter without a gradient is skipped (ref:utils/muon_utils.py:153-154)
    return out
