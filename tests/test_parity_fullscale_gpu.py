"""Parity at the BASELINE shapes and FULL depth: the 22-layer beatmap / 6-layer metadata (/ 6-layer audio) default
architecture with reference-init weights, HIP model against the CPU oracle (= the reference's fp32 sdpa path restated,
pinned by tests/test_oracle_golden.py), at the north-star tolerance.  Follows ref:cm3p/modeling_cm3p.py:849-1012.

Shapes (BASELINE.json configs; the batch is cut to what the oracle finishes in seconds on the GPU box's 16 host cores):
  C2  beatmap S = 4096, metadata L = 256, B = 2          loss, logits, embeddings, gradients after one backward
  C4  beatmap S = 8192, B = 1 with V = 2 metadata variations (a (1, 2, L) metadata batch makes the beatmap half of the
      loss a 2-class problem, so a single 8192-token sequence still has a non-degenerate loss and gradients), one run with
      right padding so that BOTH the padded and the unpadded (varlen) executions are checked against the same oracle result
  C5  C2's shape + input_features (B, 80, 1600) with 200 audio placeholders per row (B = 2)

  C2 / C4 at the REAL batch (B = 32 at S = 4096, B = 16 at S = 8192), forward only: the towers are per-sample independent without
      padding (ref:cm3p/modeling_cm3p.py:942-972), so the oracle side is B single-sample tower forwards + one contrastive head on
      the stacked pooled vectors - the B-way softmax of the real batch feels a logit error that a 2-way softmax hides

Tolerances (bf16 GEMM / attention operands with fp32 accumulation against an all-fp32 reference): loss |diff| <= 1e-3 is the north
star and stays; every other bound is <= 3 x the error MEASURED on MI355X (TOL below; the measured values of every run are written to
gpurun_out/parity_errors.json - or $CM3P_PARITY_JSON - and the copy committed as profiles/r03_parity_errors.json is what TOL was set
from), so a drift of a small factor inside the old, wide bounds is no longer invisible.
"""
import json
import os
import time

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


# <= 3 x measured (profiles/r03_parity_errors.json); loss: the north-star bound
# measured maxima (r03, MI355X): loss 1.4e-4, logits 2.40e-3, embeds 1.49e-3, pooled 1.47e-3, grad 6.55e-3, hidden 2.9e-4, audio 5.26e-3
TOL = dict(loss=1e-3, logits=7e-3, embeds=4.4e-3, pooled=4.4e-3, grad=1.9e-2, hidden=8.7e-4, audio=1.5e-2)
MEASURED: dict = {}


def _record(tag: str, key: str, value: float):
    MEASURED.setdefault(tag, {})[key] = float(value)


@pytest.fixture(scope="module", autouse=True)
def _dump_measured_errors():
    yield
    path = os.environ.get("CM3P_PARITY_JSON", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_errors.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(dict(tolerances=TOL, measured=MEASURED), f, indent=1, sort_keys=True)
    except OSError:
        pass


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
    dlog = (out.logits_per_metadata.float().cpu() - want["logits_per_metadata"]).abs().max().item()
    errs = dict(loss=dl, logits=dlog, beatmap_embeds=_rel(out.beatmap_embeds, want["beatmap_embeds"]),
                metadata_embeds=_rel(out.metadata_embeds, want["metadata_embeds"]),
                beatmap_pooled=_rel(out.beatmap_model_output.pooler_output, want["beatmap_pooler_output"]))
    params = dict(model.named_parameters())
    grad_errs = {}
    for k, g_want in want_grads.items():
        if g_want is None:
            continue
        g = params[k].grad
        assert g is not None and torch.isfinite(g).all(), f"{tag}: {k}"
        if g_want.norm() < 1e-12:
            continue
        grad_errs[k] = _rel(g, g_want)
    for k, v in errs.items():
        _record(tag, k, v)
    for k, v in grad_errs.items():
        _record(tag, "grad." + k, v)
    assert dl <= TOL["loss"], f"{tag}: loss {out.loss.item():.6f} vs oracle {want['loss'].item():.6f} (|diff| {dl:.2e})"
    assert dlog <= TOL["logits"], f"{tag}: max|dlogits| {dlog:.3e}"
    assert errs["beatmap_embeds"] <= TOL["embeds"] and errs["metadata_embeds"] <= TOL["embeds"], (tag, errs)
    assert errs["beatmap_pooled"] <= TOL["pooled"], (tag, errs)
    for k, r in grad_errs.items():
        assert r <= TOL["grad"], f"{tag}: grad {k} rel-L2 {r:.3e}"
    return out, len(grad_errs)


def _oracle_real_batch(sd, batch):
    """Oracle outputs at the full batch from single-sample tower forwards (no padding: samples do not interact before the head)."""
    from oracle import cm3p_oracle as O

    cfg = O.resolve_config(CFG)
    bp, mp = [], []
    t0 = time.time()
    with torch.no_grad():
        for i in range(batch["input_ids"].shape[0]):
            _, p, _ = O.beatmap_tower(sd, cfg["beatmap_config"], batch["input_ids"][i:i + 1], batch["attention_mask"][i:i + 1])
            _, q = O.metadata_tower(sd, cfg["metadata_config"], batch["metadata_ids"][i:i + 1], batch["metadata_attention_mask"][i:i + 1])
            bp.append(p)
            mp.append(q)
        out = O.contrastive_head(sd, torch.cat(bp), torch.cat(mp), None)
    out["beatmap_pooler_output"] = torch.cat(bp)
    return out, time.time() - t0


def _check_real_batch(weights, B, S, seed, tag):
    from oracle import cm3p_oracle as O

    batch = O.synthetic_batch(CFG, B=B, S=S, L=256, seed=seed)
    want, secs = _oracle_real_batch(weights, batch)
    model = _hip_model(weights).eval()
    with torch.no_grad():
        out = model(**{k: v.to(DEV) for k, v in batch.items()})
    torch.cuda.synchronize()
    dl = abs(out.loss.item() - want["loss"].item())
    dlog = (out.logits_per_metadata.float().cpu() - want["logits_per_metadata"]).abs().max().item()
    errs = dict(loss=dl, logits=dlog, beatmap_embeds=_rel(out.beatmap_embeds, want["beatmap_embeds"]),
                metadata_embeds=_rel(out.metadata_embeds, want["metadata_embeds"]),
                beatmap_pooled=_rel(out.beatmap_model_output.pooler_output, want["beatmap_pooler_output"]), oracle_seconds=secs,
                oracle_loss=want["loss"].item())
    for k, v in errs.items():
        _record(tag, k, v)
    assert out.logits_per_metadata.shape == (B, B)
    assert dl <= TOL["loss"], f"{tag}: loss {out.loss.item():.6f} vs oracle {want['loss'].item():.6f} (|diff| {dl:.2e})"
    assert dlog <= TOL["logits"], f"{tag}: max|dlogits| {dlog:.3e}"
    assert errs["beatmap_embeds"] <= TOL["embeds"] and errs["metadata_embeds"] <= TOL["embeds"] and errs["beatmap_pooled"] <= TOL["pooled"], (tag, errs)


def test_c2_loss_at_the_real_batch_of_32(weights):
    """BASELINE configs[1] AS BENCHED: B = 32 x S = 4096 / L = 256, forward; a 32-way in-batch softmax on both sides."""
    _check_real_batch(weights, 32, 4096, 2024, "C2 B=32 forward")


def test_c4_loss_at_the_real_batch_of_16(weights):
    """BASELINE configs[3] AS BENCHED: B = 16 x S = 8192, forward."""
    _check_real_batch(weights, 16, 8192, 2025, "C4 B=16 forward")


def _check_backward_at_real_batch(weights, B, S, seed, rows, tag):
    """The beatmap tower's BACKWARD at the benched batch (r04 verdict: the 131 072-row grids, the 1.65 / 3.3 GB dQ-slab workspace with
    64-bit offsets and the four-key-block slab groups had only ever been compared with themselves).  The upstream gradient is non-zero on
    `rows` only - loss = sum_i <pooled[rows[i]], r_i> - so every backward launch has the full-batch grid and workspace while the expected
    weight gradients are the sum of len(rows) single-sample oracle backward passes (samples do not interact inside a tower without
    padding, ref:cm3p/modeling_cm3p.py:942-972); embedding rows that only the OTHER samples' tokens touch must receive exact zeros."""
    from oracle import cm3p_oracle as O

    batch = O.synthetic_batch(CFG, B=B, S=S, L=256, seed=seed)
    ids, mask = batch["input_ids"], batch["attention_mask"]
    for b in rows:  # token ids 5 .. 8 are left to the samples WITHOUT an upstream gradient: their embedding rows must stay exactly zero
        ids[b][(ids[b] >= 5) & (ids[b] <= 8)] = 9
    g = torch.Generator().manual_seed(seed + 1)
    r = torch.randn(len(rows), 768, generator=g)
    keys = [k for k in GRAD_KEYS if k.startswith("beatmap_model.")]
    leaves = {k: weights[k].clone().requires_grad_(True) for k in keys}
    sd2 = dict(weights)
    sd2.update(leaves)
    cfg = O.resolve_config(CFG)
    t0 = time.time()
    want_pooled = []
    for i, b in enumerate(rows):
        _, p, _ = O.beatmap_tower(sd2, cfg["beatmap_config"], ids[b:b + 1], mask[b:b + 1])
        (p * r[i:i + 1]).sum().backward()
        want_pooled.append(p.detach())
    secs = time.time() - t0

    model = _hip_model(weights)
    model.zero_grad(set_to_none=True)
    out = model.beatmap_model(input_ids=ids.to(DEV), attention_mask=mask.to(DEV))
    pooled = out.pooler_output
    assert pooled.shape == (B, 768)
    (pooled[torch.tensor(rows, device=DEV)] * r.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    _record(tag, "oracle_seconds", secs)
    _record(tag, "pooled", _rel(pooled[torch.tensor(rows, device=DEV)], torch.cat(want_pooled)))
    params = dict(model.named_parameters())
    checked = 0
    for k in keys:
        gw, gg = leaves[k].grad, params[k].grad
        assert gg is not None and torch.isfinite(gg).all(), f"{tag}: {k}"
        e = _rel(gg, gw)
        _record(tag, "grad." + k, e)
        assert e <= TOL["grad"], f"{tag}: grad {k} rel-L2 {e:.3e}"
        checked += 1
    assert checked >= 5
    # token-embedding rows touched by none of the selected samples: exact zeros (the other B - len(rows) samples contribute nothing)
    table_grad = params["beatmap_model.encoder.embeddings.tok_embeddings.weight"].grad
    touched = torch.zeros(table_grad.shape[0], dtype=torch.bool)
    touched[ids[rows].reshape(-1)] = True
    others = torch.zeros_like(touched)
    others[ids.reshape(-1)] = True
    only_others = (others & ~touched).to(DEV)
    assert int(only_others.sum()) >= 4
    assert table_grad[only_others].abs().max().item() == 0.0
    _record(tag, "rows_only_other_samples_touch", float(only_others.sum()))


def test_c2_backward_at_the_real_batch_of_32(weights):
    """BASELINE configs[1] AS BENCHED, backward: B = 32 x S = 4096, upstream gradient on 4 of the 32 samples."""
    _check_backward_at_real_batch(weights, 32, 4096, 3031, [0, 9, 22, 31], "C2 B=32 backward")


def test_c4_backward_at_the_real_batch_of_16(weights):
    """BASELINE configs[3] AS BENCHED, backward: B = 16 x S = 8192 (four key blocks per dQ slab), upstream gradient on 3 of the 16."""
    _check_backward_at_real_batch(weights, 16, 8192, 3032, [1, 8, 15], "C4 B=16 backward")


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
    _record("C4 padded", "hidden", _rel(hp[0, :valid], want["beatmap_last_hidden_state"][0, :valid]))
    assert _rel(hp[0, :valid], want["beatmap_last_hidden_state"][0, :valid]) <= TOL["hidden"]

    model.unpad_inputs = True
    _lib.profile_begin()
    out_u, checked = _check(model, batch, want, grads, "C4 unpadded")
    tags = set(_lib.profile_end())
    assert any("varlen" in t for t in tags), tags
    assert checked >= 6
    hu = out_u.beatmap_model_output.last_hidden_state.detach().float().cpu()
    _record("C4 unpadded", "hidden", _rel(hu[0, :valid], want["beatmap_last_hidden_state"][0, :valid]))
    _record("C4 unpadded", "hidden_vs_padded", _rel(hu[0, :valid], hp[0, :valid]))
    assert _rel(hu[0, :valid], want["beatmap_last_hidden_state"][0, :valid]) <= TOL["hidden"]
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
    _record("C5", "audio_embeds", _rel(got_audio, want["audio_embeds"]))
    assert _rel(got_audio, want["audio_embeds"]) <= TOL["audio"]
