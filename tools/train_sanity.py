#!/usr/bin/env python3
"""A few real optimizer steps at the default architecture (development aid): default CM3P config, synthetic batch, the Muon
optimizer of cm3p_amd.muon with the reference's parameter split (ref:train.py:331-340).  The same batch is repeated, so the loss
must fall monotonically-ish from ~ln(B) and stay finite.

    python tools/train_sanity.py [--batch 8] [--seq 4096] [--steps 12] [--mlm]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import CM3PConfig, CM3PModel  # noqa: E402
from cm3p_amd.muon import Muon  # noqa: E402
from cm3p_amd.synthetic import synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--mlm", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    kw = dict(has_decoder_head=True, loss_type="ForMaskedLM") if args.mlm else {}
    cfg = CM3PConfig(beatmap_config=dict(cls_embed=args.mlm), metadata_config=dict(cls_embed=args.mlm), **kw)
    torch.manual_seed(0)
    model = CM3PModel(cfg).to(dev).train()
    for p in model.beatmap_model.audio_encoder.parameters():
        p.requires_grad_(False)
    batch = {k: v.to(dev) for k, v in synthetic_batch(cfg, args.batch, args.seq, 256, seed=5).items()}
    if args.mlm:
        g = torch.Generator().manual_seed(1)
        pick = (torch.rand(args.batch, args.seq, generator=g) < 0.15).to(dev)
        batch["labels"] = torch.where(pick, batch["input_ids"], torch.full_like(batch["input_ids"], -100))
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    adamw = [p for n, p in named if any(k in n.lower() for k in ("embed", "proj_out")) or p.ndim <= 1]
    muon = [p for n, p in named if not (any(k in n.lower() for k in ("embed", "proj_out")) or p.ndim <= 1)]
    opt = Muon(muon, lr=4e-4, momentum=0.95, adamw_params=adamw, adamw_lr=4e-4 * 0.5)
    losses = []
    for step in range(args.steps):
        opt.zero_grad(set_to_none=True)
        out = model(**batch)
        out.loss.backward()
        opt.step()
        losses.append(out.loss.detach().item())
        print(f"step {step:2d} loss {losses[-1]:.5f}", flush=True)
    assert all(l == l and abs(l) < 1e4 for l in losses), "non-finite loss"
    assert losses[-1] < losses[0], "loss did not fall on a repeated batch"
    print("ok: loss", losses[0], "->", losses[-1])


if __name__ == "__main__":
    main()
