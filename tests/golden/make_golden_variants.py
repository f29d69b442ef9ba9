#!/usr/bin/env python3
"""Golden fixtures for the stand-alone model classes (CM3PBeatmapModelWithProjection, CM3PMetadataModelWithProjection,
CM3PForMaskedLM), produced by running the REFERENCE classes on the CPU (fp32, sdpa) with the d64 weights.

Build container only:   python tests/golden/make_golden_variants.py   -> tests/golden/variants_d64.safetensors
"""
from __future__ import annotations

import os
import sys

import torch
from safetensors.torch import load_file, save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

from cases import CASES, make_inputs  # noqa: E402
from make_golden import _shim_config  # noqa: E402  (also imports the reference package and asserts where it came from)

from cm3p import CM3PConfig  # noqa: E402
from cm3p.modeling_cm3p import (CM3PBeatmapModelWithProjection, CM3PForBeatmapClassification, CM3PForMaskedLM,  # noqa: E402
                                CM3PMetadataModelWithProjection)


def main():
    torch.set_num_threads(8)
    name = "d64_mlm"
    cfg = CM3PConfig(**CASES[name]["cfg"])
    for sub in (cfg.metadata_config, cfg.beatmap_config, cfg.beatmap_config.audio_config):
        _shim_config(sub)
    weights = load_file(os.path.join(HERE, "weights_d64.safetensors"))
    extra = {k[2:]: v for k, v in load_file(os.path.join(HERE, f"{name}.safetensors")).items() if k.startswith("w.")}
    weights.update(extra)
    inputs = make_inputs(name)
    blob = {}

    bm = CM3PBeatmapModelWithProjection._from_config(cfg.beatmap_config, attn_implementation="sdpa").float().train()
    missing = bm.load_state_dict({k: v for k, v in weights.items() if k in bm.state_dict()}, strict=True)
    out = bm(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"])
    out.beatmap_embeds.square().sum().backward()
    blob["bproj.beatmap_embeds"] = out.beatmap_embeds.detach().contiguous()
    blob["bproj.grad.beatmap_projection.weight"] = bm.beatmap_projection.weight.grad.clone()
    blob["bproj.grad.beatmap_model.encoder.layers.1.attn.Wqkv.weight"] = bm.beatmap_model.encoder.layers[1].attn.Wqkv.weight.grad.clone()

    mm = CM3PMetadataModelWithProjection._from_config(cfg.metadata_config, attn_implementation="sdpa").float().train()
    mm.load_state_dict({k: v for k, v in weights.items() if k in mm.state_dict()}, strict=True)
    out = mm(input_ids=inputs["metadata_ids"], attention_mask=inputs["metadata_attention_mask"])
    blob["mproj.metadata_embeds"] = out.metadata_embeds.detach().contiguous()

    ml = CM3PForMaskedLM._from_config(cfg.beatmap_config, attn_implementation="sdpa").float().train()
    sd = {k: v for k, v in weights.items() if k in ml.state_dict()}
    print("ForMaskedLM keys missing from the weight file:", sorted(set(ml.state_dict()) - set(sd)))
    ml.load_state_dict(sd, strict=False)
    tied = ml.decoder.weight.data_ptr() == ml.beatmap_model.get_input_embeddings().weight.data_ptr()
    print("decoder.weight tied to the token embeddings:", tied)
    out = ml(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"], labels=inputs["labels"])
    out.loss.backward()
    blob["mlm.loss"] = out.loss.detach().reshape(1)
    blob["mlm.logits"] = out.logits.detach().contiguous()
    blob["mlm.tied"] = torch.tensor([int(tied)])
    blob["mlm.grad.head.dense.weight"] = ml.head.dense.weight.grad.clone()
    blob["mlm.grad.decoder.weight"] = ml.decoder.weight.grad.clone()
    blob["mlm.grad.beatmap_model.encoder.layers.0.mlp.Wi.weight"] = ml.beatmap_model.encoder.layers[0].mlp.Wi.weight.grad.clone()
    # sparse_prediction: head + decoder on the labelled positions only (ref:cm3p/modeling_cm3p.py:1349-1357)
    import copy as _copy
    sc = _copy.deepcopy(cfg.beatmap_config)
    sc.sparse_prediction = True
    sp = CM3PForMaskedLM._from_config(sc, attn_implementation="sdpa").float().train()
    sp.load_state_dict({k: v for k, v in weights.items() if k in sp.state_dict()}, strict=False)
    out = sp(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"], labels=inputs["labels"])
    out.loss.backward()
    blob["mlm_sparse.loss"] = out.loss.detach().reshape(1)
    blob["mlm_sparse.logits"] = out.logits.detach().contiguous()
    blob["mlm_sparse.grad.decoder.weight"] = sp.decoder.weight.grad.clone()
    blob["mlm_sparse.grad.beatmap_model.encoder.layers.0.mlp.Wi.weight"] = sp.beatmap_model.encoder.layers[0].mlp.Wi.weight.grad.clone()
    print("sparse MLM logits", tuple(out.logits.shape), "loss", float(out.loss))

    # classifier variant, the three loss flavours HF infers from num_labels / label dtype (ref:cm3p/modeling_cm3p.py:1196-1218)
    g = torch.Generator().manual_seed(99)
    B = inputs["input_ids"].shape[0]
    for tag, nl, labels in (("cls_ce", 5, torch.randint(0, 5, (B,), generator=g)),
                            ("cls_mse", 1, torch.randn(B, generator=g)),
                            ("cls_bce", 4, (torch.rand(B, 4, generator=g) > 0.5).float())):
        import copy
        bc = copy.deepcopy(cfg.beatmap_config)
        bc.num_labels = nl
        bc.problem_type = None
        cl = CM3PForBeatmapClassification._from_config(bc, attn_implementation="sdpa").float().train()
        cl.load_state_dict({k: v for k, v in weights.items() if k in cl.state_dict()}, strict=False)
        with torch.no_grad():
            cl.classifier.weight.copy_(torch.randn(cl.classifier.weight.shape, generator=g) * 0.3)
            cl.classifier.bias.copy_(torch.randn(cl.classifier.bias.shape, generator=g) * 0.1)
        out = cl(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"], labels=labels)
        out.loss.backward()
        blob[f"{tag}.labels"] = labels.contiguous()
        blob[f"{tag}.w.classifier.weight"] = cl.classifier.weight.detach().clone()
        blob[f"{tag}.w.classifier.bias"] = cl.classifier.bias.detach().clone()
        blob[f"{tag}.loss"] = out.loss.detach().reshape(1)
        blob[f"{tag}.logits"] = out.logits.detach().contiguous()
        blob[f"{tag}.grad.classifier.weight"] = cl.classifier.weight.grad.clone()
        blob[f"{tag}.grad.classifier.bias"] = cl.classifier.bias.grad.clone()
        blob[f"{tag}.grad.beatmap_model.encoder.final_norm.weight"] = cl.beatmap_model.encoder.final_norm.weight.grad.clone()
        print(tag, "problem_type ->", cl.config.problem_type, "loss", float(out.loss))
    save_file(blob, os.path.join(HERE, "variants_d64.safetensors"))
    for k, v in blob.items():
        print(f"{k:70s} {tuple(v.shape)}")


if __name__ == "__main__":
    main()
