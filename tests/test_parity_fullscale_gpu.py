"""Parity at the BASELINE shapes and FULL depth: the 22-layer beatmap / 6-layer metadata (/ 6-layer audio) default
architecture with reference-init weights, HIP model against the CPU oracle (= the reference's fp32 sdpa path restated,
pinned by tests/test_oracle_golden.py), at the north-star tolerance.  Follows ref:cm3p/modeling_cm3p.py:849-1012.

Shapes (BASELINE.json configs; the batch is cut to what the oracle finishes in seconds on the GPU box's 16 host cores):
  C2  beatmap S = 4096, metadata L = 256, B = 2          loss, logits, embeddings, gradients after one backward
  C4  beatmap S = 8192, B = 1 with V = 2 metadata variations (a (1, 2, L) metadata batch makes the beatmap half of the
      loss a 2-class problem, so a single 8192-token sequence still has a non-degenerate loss and gradients), one run with
      right padding so that BOTH the padded and the unpadded (varlen) executions are checked against the same oracle result
  C5  C2's shape + input_features (B, 80, 1600) with 200 audio placeholders per row (B = 2)

Tolerances (bf16 GEMM / attention operands with fp32 accumulation against an all-fp32 reference):
  loss |diff| <= 1e-3 (north star);  logits max|diff| <= 2e-2;  embeddings rel-L2 <= 2e-2;  gradients rel-L2 <= 6e-2.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
CFG = dict(beatmap_config=dict(cls_embed=False), metadata_config=dict(cls_embed=False))  # ref:configs/model/default.yaml

GRAD_KEYS = [
    "beatmap_model.encoder.layers.0.attn.Wqkv.weight",      # global layer, no attn_norm
    "beatmap_model.encoder.layers.1.attn.Wqkv.weight",      # sliding-window layer
    "beatmap_model.encoder.layers.10.mlp.Wi.weight",
    "beatmap_model.encoder.layers.21.attn.Wo.weight",       # last layer (global)
    "beatmap_model.encoder.layers.12.mlp_norm.weight",
    "beatmap_model.encoder.embeddings.tok_embeddings.weight",
    "beatmap_model.encoder.final_norm.weight",
    "beatmap_projection.weight",
    "metadata_projection.weight",
    "metadata_model.encoder.layers.3.mlp.Wo.weight",
]


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


@pytest.fixture(scope="module")
def weights():
    from oracle import cm3p_oracle as O

    torch.set_num_threads(min(16, torch.get_num_threads() if torch.get_num_threads() > 1 else 16))
    return O.init_state_dict(CFG, seed=0, with_audio=True)


def _hip_model(sd):
    from cm3p_amd import CM3PConfig, CM3PModel

    model = CM3PModel(CM3PConfig(**CFG))
    model.load_state_dict(sd, strict=True)
    return model.to(DEV).train()


def _oracle_step(sd, batch, keys):
    """Oracle forward + backward on the CPU -> (outputs, {key: grad})."""
    from oracle import cm3p_oracle as O

    leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
    sd2 = dict(sd)
    sd2.update(leaves)
    out = O.forward(sd2, CFG, **batch)
    out["loss"].backward()
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}, {k: v.grad for k, v in leaves.items()}


def _check(model, batch, want, want_grads, tag):
    model.zero_grad(set_to_none=True)
    out = model(**{k: v.to(DEV) for k, v in batch.items()})
    out.loss.backward()
    torch.cuda.synchronize()
    dl = abs(out.loss.item() - want["loss"].item())
    assert dl <= 1e-3, f"{tag}: loss {out.loss.item():.6f} vs oracle {want['loss'].item():.6f} (|diff| {dl:.2e})"
    dlog = (out.logits_per_metadata.float().cpu() - want["logits_per_metadata"]).abs().max().item()
    assert dlog <= 2e-2, f"{tag}: max|dlogits| {dlog:.3e}"
    assert _rel(out.beatmap_embeds, want["beatmap_embeds"]) <= 2e-2, tag
    assert _rel(out.metadata_embeds, want["metadata_embeds"]) <= 2e-2, tag
    assert _rel(out.beatmap_model_output.pooler_output, want["beatmap_pooler_output"]) <= 2e-2, tag
    params = dict(model.named_parameters())
    checked = 0
    for k, g_want in want_grads.items():
        if g_want is None:
            continue
        g = params[k].grad
        assert g is not None and torch.isfinite(g).all(), f"{tag}: {k}"
        if g_want.norm() < 1e-12:
            continue
        r = _rel(g, g_want)
        assert r <= 6e-2, f"{tag}: grad {k} rel-L2 {r:.3e}"
        checked += 1
    return out, checked


def test_c2_shape_full_depth_forward_backward(weights):
    """BASELINE configs[1] shape: S = 4096 / L = 256, all 22 + 6 layers, B = 2, one optimizer-free step."""
    from oracle import cm3p_oracle as O

    batch = O.synthetic_batch(CFG, B=2, S=4096, L=256, seed=1234)
    want, grads = _oracle_step(weights, batch, GRAD_KEYS)
    model = _hip_model(weights)
    _, checked = _check(model, batch, want, grads, "C2")
    assert checked >= 8


def test_c4_shape_seq8192_padded_and_unpadded(weights):
    """BASELINE configs[3] shape: one 8192-token beatmap (valid length 7001, right padded) against two metadata variations;
    the padded kernels and the unpadded (varlen) kernels must both match the oracle, and each other on the valid rows."""
    from oracle import cm3p_oracle as O

    S, L, valid = 8192, 256, 7001
    batch = O.synthetic_batch(CFG, B=1, S=S, L=L, seed=4321)
    mask = (torch.arange(S)[None, :] < valid).to(torch.int64)
    batch["attention_mask"] = mask
    batch["input_ids"] = batch["input_ids"] * mask
    g = torch.Generator().manual_seed(99)
    batch["metadata_ids"] = torch.randint(3, 997, (1, 2, L), generator=g, dtype=torch.int64)
    batch["metadata_attention_mask"] = torch.ones(1, 2, L, dtype=torch.int64)
    batch["metadata_variation_classes"] = torch.tensor([[1, 0]], dtype=torch.int64)  # the true metadata is variation 1
    keys = [k for k in GRAD_KEYS if not k.startswith("metadata_model")]
    want, grads = _oracle_step(weights, batch, keys)
    assert want["loss"].item() > 1e-3  # non-degenerate by construction

    from cm3p_amd import _lib

    model = _hip_model(weights)
    model.unpad_inputs = False
    out_p, checked = _check(model, batch, want, grads, "C4 padded")
    assert checked >= 6
    hp = out_p.beatmap_model_output.last_hidden_state.detach().float().cpu()
    assert _rel(hp[0, :valid], want["beatmap_last_hidden_state"][0, :valid]) <= 2e-2

    model.unpad_inputs = True
    _lib.profile_begin()
    out_u, checked = _check(model, batch, want, grads, "C4 unpadded")
    tags = set(_lib.profile_end())
    assert any("varlen" in t for t in tags), tags
    assert checked >= 6
    hu = out_u.beatmap_model_output.last_hidden_state.detach().float().cpu()
    assert _rel(hu[0, :valid], want["beatmap_last_hidden_state"][0, :valid]) <= 2e-2
    assert hu[0, valid:].abs().max().item() == 0.0  # _pad_cm3p_output zero-fills the padding rows
    assert _rel(hu[0, :valid], hp[0, :valid]) <= 5e-3


def test_c5_shape_audio_fused_full_depth(weights):
    """BASELINE configs[4] shape: C2 + input_features (B, 80, 1600), 200 audio placeholders per row, 6-layer audio encoder."""
    from oracle import cm3p_oracle as O

    batch = O.synthetic_batch(CFG, B=2, S=4096, L=256, seed=777, audio_T=1600)
    assert int((batch["input_ids"] == 3166).sum()) == 2 * 200
    keys = GRAD_KEYS + [
        "beatmap_model.audio_encoder.conv1.weight",
        "beatmap_model.audio_encoder.encoder.layers.3.attn.Wqkv.weight",
        "beatmap_model.audio_encoder.multi_modal_projector.linear_2.weight",
    ]
    want, grads = _oracle_step(weights, batch, keys)
    model = _hip_model(weights)
    out, checked = _check(model, batch, want, grads, "C5")
    assert checked >= 11
    got_audio = out.beatmap_model_output.audio_model_output.audio_embeds
    assert got_audio.shape == want["audio_embeds"].shape == (400, 768)
    assert _rel(got_audio, want["audio_embeds"]) <= 2e-2
