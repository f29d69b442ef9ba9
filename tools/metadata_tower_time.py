#!/usr/bin/env python3
"""How much of the C2 step is the metadata tower (32 x 256 tokens, H = 256, 6 layers)?  fwd + bwd of the tower + projection alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import CM3PConfig, CM3PModel  # noqa: E402
from cm3p_amd.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda", 0)
cfg = CM3PConfig(beatmap_config=dict(cls_embed=False), metadata_config=dict(cls_embed=False))
torch.manual_seed(0)
model = CM3PModel(cfg).to(dev).train()
b = {k: v.to(dev) for k, v in synthetic_batch(cfg, 32, 4096, 256, seed=1).items()}


def run():
    for p in model.parameters():
        p.grad = None
    out = model.metadata_model(input_ids=b["metadata_ids"], attention_mask=b.get("metadata_attention_mask"))
    out.pooler_output.float().sum().backward()


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print(f"metadata tower fwd + bwd: {e0.elapsed_time(e1) / 20:.3f} ms per step")
