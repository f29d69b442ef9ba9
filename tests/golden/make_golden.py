#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE itself (CPU, fp32, attn_implementation="sdpa").

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py            # writes tests/golden/*.safetensors

The reference package is imported from /root/reference unmodified.  It targets transformers 4.55; the
installed transformers is 5.15, whose ModernBertModel reads four derived attributes the reference's
config classes do not define.  `_shim_config` adds exactly the derivation the installed
ModernBertConfig performs; no reference file is touched or copied.

What is stored per case (all fp32 / int64, CPU):
  inputs, loss, logits_per_metadata, metadata_embeds, beatmap_embeds, both pooler outputs,
  last_hidden_state of both towers (selected cases), per-layer hidden states (one case),
  audio_embeds (audio case), and gradients of a fixed subset of parameters.
Weights are stored once per architecture (`weights_<arch>.safetensors`).
"""
from __future__ import annotations

import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

from cases import CASES, make_inputs  # noqa: E402

import cm3p  # noqa: E402  (the reference package)
from cm3p import CM3PConfig, CM3PModel  # noqa: E402

assert cm3p.__file__.startswith("/root/reference/"), cm3p.__file__

GRAD_KEYS = [
    "logit_scale",
    "beatmap_projection.weight",
    "metadata_projection.weight",
    "beatmap_model.encoder.embeddings.tok_embeddings.weight",
    "beatmap_model.encoder.embeddings.norm.weight",
    "beatmap_model.encoder.layers.0.attn.Wqkv.weight",
    "beatmap_model.encoder.layers.0.mlp.Wi.weight",
    "beatmap_model.encoder.layers.1.attn_norm.weight",
    "beatmap_model.encoder.layers.1.attn.Wqkv.weight",
    "beatmap_model.encoder.layers.1.attn.Wo.weight",
    "beatmap_model.encoder.layers.2.mlp.Wo.weight",
    "beatmap_model.encoder.final_norm.weight",
    "metadata_model.encoder.embeddings.tok_embeddings.weight",
    "metadata_model.encoder.layers.0.attn.Wqkv.weight",
    "metadata_model.encoder.layers.1.mlp.Wi.weight",
    "metadata_model.encoder.final_norm.weight",
]
AUDIO_GRAD_KEYS = [
    "beatmap_model.audio_encoder.conv1.weight",
    "beatmap_model.audio_encoder.conv1.bias",
    "beatmap_model.audio_encoder.conv2.weight",
    "beatmap_model.audio_encoder.encoder.layers.0.attn.Wqkv.weight",
    "beatmap_model.audio_encoder.encoder.layers.1.mlp.Wo.weight",
    "beatmap_model.audio_encoder.multi_modal_projector.linear_1.weight",
    "beatmap_model.audio_encoder.multi_modal_projector.linear_2.weight",
]
FULL_HIDDEN_CASES = {"d64_mean_pad", "d64_mean_longpad", "d64_short_seq", "c1_tiny_nopad"}
PER_LAYER_CASE = "d64_mean_pad"


def _shim_config(c):
    """Derive the 5.x attribute names from the reference's 4.55-style fields."""
    n = c.global_attn_every_n_layers
    c.layer_types = ["sliding_attention" if i % n else "full_attention" for i in range(c.num_hidden_layers)]
    c.rope_parameters = {
        "full_attention": {"rope_type": "default", "rope_theta": c.global_rope_theta},
        "sliding_attention": {"rope_type": "default", "rope_theta": c.local_rope_theta},
    }
    c.sliding_window = c.local_attention // 2
    if not hasattr(c, "pad_token_id"):
        c.pad_token_id = None


def build_model(cfg_kwargs) -> CM3PModel:
    cfg = CM3PConfig(**cfg_kwargs)
    for sub in (cfg.metadata_config, cfg.beatmap_config, cfg.beatmap_config.audio_config):
        _shim_config(sub)
    torch.manual_seed(0)
    model = CM3PModel._from_config(cfg, attn_implementation="sdpa")
    # Re-draw the weights at O(1) activation scale so that attention, RoPE and the norms are all
    # numerically visible in the outputs (std-0.02 init makes softmax ~uniform and hides indexing bugs).
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name == "logit_scale":
                continue
            if "norm" in name:
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif "tok_embeddings" in name:
                p.copy_(torch.randn(p.shape, generator=g))
                p[0].zero_()  # padding_idx row stays zero like nn.Embedding's init
            elif p.ndim >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * fan_in ** -0.5)
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return model.float().train()


def arch_of(name: str) -> str:
    return "c1" if name.startswith("c1") else "d64"


def main():
    torch.set_num_threads(8)
    models = {}
    for name, case in CASES.items():
        arch = arch_of(name)
        model = build_model(case["cfg"])
        if arch not in models:
            sd = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}
            save_file(sd, os.path.join(HERE, f"weights_{arch}.safetensors"))
            models[arch] = {k: v for k, v in sd.items()}
        else:
            # same seed + same shapes -> same weights regardless of cls_embed
            for k, v in model.state_dict().items():
                if k in models[arch]:
                    assert torch.equal(v, models[arch][k]), k

        inputs = make_inputs(name)
        hiddens = {}
        hooks = []
        if name == PER_LAYER_CASE:
            enc = model.beatmap_model.encoder
            hooks.append(enc.embeddings.register_forward_hook(
                lambda m, i, o: hiddens.__setitem__("beatmap_hidden_emb", o.detach().clone())))
            for li, layer in enumerate(enc.layers):
                hooks.append(layer.register_forward_hook(
                    lambda m, i, o, li=li: hiddens.__setitem__(f"beatmap_hidden_{li}", o.detach().clone())))

        model.zero_grad(set_to_none=True)
        out = model(**inputs)
        out.loss.backward()
        for h in hooks:
            h.remove()

        blob = {f"in.{k}": v.contiguous() for k, v in inputs.items()}
        extra = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items() if k not in models[arch]}
        blob.update({f"w.{k}": v for k, v in extra.items()})  # parameters this case adds to the shared weights (MLM head)
        if out.logits is not None:
            blob["logits"] = out.logits.detach().contiguous()
        blob["loss"] = out.loss.detach().reshape(1)
        blob["logits_per_metadata"] = out.logits_per_metadata.detach().contiguous()
        blob["metadata_embeds"] = out.metadata_embeds.detach().contiguous()
        blob["beatmap_embeds"] = out.beatmap_embeds.detach().contiguous()
        blob["beatmap_pooler_output"] = out.beatmap_model_output.pooler_output.detach().contiguous()
        blob["metadata_pooler_output"] = out.metadata_model_output.pooler_output.detach().contiguous()
        if name in FULL_HIDDEN_CASES:
            blob["beatmap_last_hidden_state"] = out.beatmap_model_output.last_hidden_state.detach().contiguous()
            blob["metadata_last_hidden_state"] = out.metadata_model_output.last_hidden_state.detach().contiguous()
        if "input_features" in inputs:
            am = out.beatmap_model_output.audio_model_output
            blob["audio_embeds"] = am.audio_embeds.detach().contiguous()
        blob.update(hiddens)
        params = dict(model.named_parameters())
        keys = GRAD_KEYS + (AUDIO_GRAD_KEYS if "input_features" in inputs else [])
        keys = keys + [k for k in ("head.dense.weight", "head.norm.weight", "decoder.weight", "decoder.bias") if k in params]
        for k in keys:
            if k in params and params[k].grad is not None:
                blob[f"grad.{k}"] = params[k].grad.detach().clone().contiguous()

        # sanity: the reference's eager attention path agrees with its sdpa path on the same inputs
        model.config._attn_implementation = "eager"
        for sub in (model.config.metadata_config, model.config.beatmap_config, model.config.beatmap_config.audio_config):
            sub._attn_implementation = "eager"
        with torch.no_grad():
            out_e = model(**inputs)
        d = (out_e.logits_per_metadata - out.logits_per_metadata).abs().max().item()
        save_file(blob, os.path.join(HERE, f"{name}.safetensors"))
        size = sum(v.numel() * v.element_size() for v in blob.values()) / 1e6
        print(f"{name:20s} loss={out.loss.item():.7f}  sdpa-vs-eager max|dlogits|={d:.2e}  {size:.2f} MB")


if __name__ == "__main__":
    main()
