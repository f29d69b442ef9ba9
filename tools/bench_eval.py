#!/usr/bin/env python3
"""Forward-only throughput of the two evaluation consumers of the hot-path kernels (development aid,
the judged number comes from bench.py):

  extract     ref:extract_beatmap_embeddings.py:217-234 - model(input_ids, attention_mask, return_loss=False) under no_grad,
              default config, B x 4096 beatmap tokens -> beatmap_embeds
  variations  evaluation with V metadata variations per row (ref:configs/train/default.yaml:147 test_metadata_variations: 1000):
              a (B, V, 256) metadata batch through the metadata tower + logits + loss

    python tools/bench_eval.py [extract] [variations] [--batch 32] [--variations 1000] [--var-batch 8] [--iters 5]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import tower_flops_fwd  # noqa: E402
from cm3p_amd import CM3PConfig, CM3PModel, _lib  # noqa: E402
from cm3p_amd.synthetic import synthetic_batch  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def breakdown(fn, top=8):
    _lib.profile_begin()
    fn()
    prof = _lib.profile_end()
    return {k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:top]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["extract", "variations"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--variations", type=int, default=1000)
    ap.add_argument("--var-batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--meta-valid", type=float, default=1.0,
                    help="variations: valid length of a metadata row ~ U{1..meta_valid * L} (right-padded); below 1 the run is repeated with "
                         "unpadded execution (model.unpad_inputs = True)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = CM3PConfig(beatmap_config=dict(cls_embed=False), metadata_config=dict(cls_embed=False))
    torch.manual_seed(0)
    model = CM3PModel(cfg).to(dev).eval()
    S, L = 4096, 256
    if "extract" in args.what:
        b = {k: v.to(dev) for k, v in synthetic_batch(cfg, args.batch, S, L, seed=1234).items()}

        def run():
            with torch.no_grad():
                return model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], return_loss=False).beatmap_embeds

        ms = timed(run, args.iters)
        fl = tower_flops_fwd(cfg.beatmap_config, args.batch * S, S)
        print(json.dumps({"path": "extract", "batch": args.batch, "seq": S, "ms": ms, "beatmaps_per_s": args.batch / ms * 1e3,
                          "tokens_per_s": args.batch * S / ms * 1e3, "tflops": fl / ms / 1e9, "kernels_ms": breakdown(run)}))
    if "variations" in args.what:
        B, V = args.var_batch, args.variations
        b = {k: v.to(dev) for k, v in synthetic_batch(cfg, B, S, L, seed=99).items()}
        g = torch.Generator().manual_seed(7)
        mids = torch.randint(3, cfg.metadata_config.vocab_size - 3, (B, V, L), generator=g).to(dev)
        mmask = torch.ones(B, V, L, dtype=torch.int64, device=dev)
        classes = torch.randint(1, 4, (B, V), generator=g)
        classes[:, 0] = 0  # one original per row
        classes = classes.to(dev)
        with torch.no_grad():
            bm = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], return_loss=False).beatmap_embeds

        def run():
            with torch.no_grad():  # the metadata side of the evaluation step (the beatmap side is the `extract` path above)
                return model.get_metadata_features(metadata_ids=mids.view(B * V, L), metadata_attention_mask=mmask.view(B * V, L)) \
                    if hasattr(model, "get_metadata_features") else None

        def run_full():
            with torch.no_grad():
                return model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], metadata_ids=mids, metadata_attention_mask=mmask,
                             metadata_variation_classes=classes, return_loss=True).loss

        if args.meta_valid < 1.0:
            lens = torch.randint(1, max(2, int(args.meta_valid * L)) + 1, (B, V), generator=g).to(dev)
            mmask = (torch.arange(L, device=dev)[None, None, :] < lens[..., None]).to(torch.int64)
            ms_pad = timed(run_full, args.iters)
            model.unpad_inputs = True
            ms_unp = timed(run_full, args.iters)
            model.unpad_inputs = None
            print(json.dumps({"path": "variations, right-padded metadata", "valid_fraction": float(mmask.float().mean()), "ms_padded_execution": ms_pad,
                              "ms_unpadded_execution": ms_unp}))
            mmask = torch.ones(B, V, L, dtype=torch.int64, device=dev)
        ms = timed(run_full, args.iters)
        fl_m = tower_flops_fwd(cfg.metadata_config, B * V * L, L)
        fl_b = tower_flops_fwd(cfg.beatmap_config, B * S, S)
        print(json.dumps({"path": "variations", "batch": B, "variations": V, "metadata_seq": L, "ms": ms,
                          "metadata_sequences_per_s": B * V / ms * 1e3, "metadata_tokens_per_s": B * V * L / ms * 1e3,
                          "tflops": (fl_m + fl_b) / ms / 1e9, "metadata_tower_tflop": fl_m / 1e12, "kernels_ms": breakdown(run_full, 10)}))
        del bm


if __name__ == "__main__":
    main()
