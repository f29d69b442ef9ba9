"""Pins the CPU oracle (oracle/cm3p_oracle.py) to fixtures produced by the reference itself.

Fixtures: tests/golden/*.safetensors, written by tests/golden/make_golden.py from the reference's CPU
sdpa path (fp32).  Integer tensors must match bit-for-bit, fp32 values to <= 1e-5.
"""
import os

import pytest
import torch
from safetensors.torch import load_file

from cases import CASES, make_inputs
from oracle import cm3p_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    arch = "c1" if name.startswith("c1") else "d64"
    sd = load_file(os.path.join(GOLD, f"weights_{arch}.safetensors"))
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    sd.update({k[2:]: v for k, v in blob.items() if k.startswith("w.")})  # parameters only this case has (MLM head)
    return sd, blob


def _close(a, b, tol, what):
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, f"{what}: max|err| {err:.3e} > {tol:.0e} * {scale:.2f}"


@pytest.mark.parametrize("name", list(CASES))
def test_inputs_are_reproducible(name):
    """The seeded input builder reproduces the stored inputs bit-for-bit (ids, masks, classes)."""
    _, blob = _load(name)
    inputs = make_inputs(name)
    for k, v in inputs.items():
        assert torch.equal(v, blob[f"in.{k}"]), k


@pytest.mark.parametrize("eager", [False, True])
@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_reference(name, eager):
    sd, blob = _load(name)
    inputs = {k[3:]: v for k, v in blob.items() if k.startswith("in.")}
    collect = []
    with torch.no_grad():
        out = O.forward(sd, CASES[name]["cfg"], eager=eager, collect=collect, **inputs)
    tol = 1e-5 if not eager else 2e-5
    for key in ("loss", "logits_per_metadata", "metadata_embeds", "beatmap_embeds",
                "beatmap_pooler_output", "metadata_pooler_output",
                "beatmap_last_hidden_state", "metadata_last_hidden_state", "audio_embeds", "logits"):
        if key in blob:
            _close(out[key].reshape(blob[key].shape), blob[key], tol, f"{name}:{key}")
    if "beatmap_hidden_emb" in blob:
        _close(collect[0], blob["beatmap_hidden_emb"], tol, "embeddings")
        for i, h in enumerate(collect[1:]):
            _close(h, blob[f"beatmap_hidden_{i}"], tol, f"layer {i}")


@pytest.mark.parametrize("name", list(CASES))
def test_backward_matches_reference(name):
    sd, blob = _load(name)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    inputs = {k[3:]: v for k, v in blob.items() if k.startswith("in.")}
    out = O.forward(sd, CASES[name]["cfg"], **inputs)
    out["loss"].backward()
    n = 0
    for k, v in blob.items():
        if k.startswith("grad."):
            _close(sd[k[5:]].grad, v, 2e-5, k)
            n += 1
    assert n >= 10


def test_fully_masked_rows_are_exact_zeros():
    """Local-layer queries > 64 past the last valid key see no key: the reference's SDPA returns exact zeros
    there, so the hidden state of such a row after the attention residual is finite."""
    sd, blob = _load("d64_mean_longpad")
    assert torch.isfinite(blob["beatmap_last_hidden_state"]).all()
    q = torch.randn(1, 1, 8, 16)
    allowed = torch.zeros(1, 1, 8, 8, dtype=torch.bool)
    allowed[..., :4, :] = True
    for eager in (False, True):
        o = O.sdpa(q, q, q, allowed, 0.25, eager=eager)
        assert torch.equal(o[..., 4:, :], torch.zeros(1, 1, 4, 16))


def test_true_variation_index_and_targets():
    """Integer work in the 3-D loss is bit-exact: first class-0 slot, target = b*V + slot (ref:modeling_cm3p.py:40-46)."""
    classes = torch.tensor([[1, 0, 2], [0, 0, 3], [2, 3, -1], [3, 1, 0]])
    assert O.true_variation_index(classes).tolist() == [1, 0, 0, 2]


def test_loss_3d_needs_classes():
    sd, blob = _load("d64_variations")
    inputs = {k[3:]: v for k, v in blob.items() if k.startswith("in.")}
    inputs.pop("metadata_variation_classes")
    with pytest.raises(ValueError):
        O.forward(sd, CASES["d64_variations"]["cfg"], **inputs)
