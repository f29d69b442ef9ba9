"""The Muon restatement (oracle/muon_oracle.py) against the fixture the reference's own optimizer produced (CPU)."""
from __future__ import annotations

import os
import sys

import pytest
import torch
from safetensors.torch import load_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from muon_cases import N_STEPS, SHAPES, VARIANTS, gradients, initial_params  # noqa: E402

from oracle import muon_oracle as MO  # noqa: E402

GOLD = load_file(os.path.join(HERE, "golden", "muon_steps.safetensors"))


def _routing():
    named = list(initial_params().items())
    muon, adamw = MO.split_like_train_py(named)
    listed_adamw = {n for n, _ in adamw}
    return {n: MO.routes_to_muon(p, n in listed_adamw) for n, p in named}


def test_routing_matches_the_reference():
    use = _routing()
    for name in SHAPES:
        assert int(use[name]) == int(GOLD[f"main.use_muon.{name}"].item()), name
    assert not use["many_rows.weight"] and not use["enc.embeddings.tok_embeddings.weight"] and use["audio.conv1.weight"]


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_oracle_reproduces_reference_steps(variant):
    hp = VARIANTS[variant]
    params = {k: v.clone() for k, v in initial_params().items()}
    states: dict = {}
    use = _routing()
    for s in range(N_STEPS):
        MO.step(params, gradients(s), states, use, lr=hp["lrs"][s], base_lr=hp["lr"], momentum=hp["momentum"], nesterov=hp["nesterov"],
                ns_steps=hp["ns_steps"], adamw_lr=hp["adamw_lr"], adamw_betas=hp["adamw_betas"], adamw_eps=hp["adamw_eps"],
                adamw_wd=hp["adamw_wd"])
        for k, p in params.items():
            want = GOLD[f"{variant}.step{s}.{k}"]
            if use[k]:
                # same op sequence, but bf16 CPU matmul kernels may block differently on another host: compare the step
                # taken, not bits.  The update has unit-order singular values, so lr * 2^-7 per element is one bf16 ulp of it.
                torch.testing.assert_close(p, want, rtol=0, atol=hp["lrs"][s] * 8e-2, msg=lambda m: f"{k} step {s}: {m}")
            else:
                torch.testing.assert_close(p, want, rtol=1e-6, atol=1e-7, msg=lambda m: f"{k} step {s}: {m}")
    for k in params:
        for sk in ("momentum_buffer", "moment1", "moment2"):
            key = f"{variant}.state.{sk}.{k}"
            if key in GOLD:
                torch.testing.assert_close(states[k][sk], GOLD[key], rtol=1e-6, atol=1e-8)
        key = f"{variant}.state.step.{k}"
        if key in GOLD:
            assert states[k]["step"] == int(GOLD[key].item())


def test_newton_schulz_lands_in_the_designed_band():
    g = torch.Generator().manual_seed(3)
    for shape in ((96, 32), (32, 48), (64, 64)):
        G = torch.randn(shape, generator=g)
        o = MO.newton_schulz5(G, 6)
        assert o.dtype == torch.bfloat16 and o.shape == G.shape
        assert MO.orthogonality_defect(o) < 0.6           # "S' ~ Uniform(0.5, 1.5)" (ref:utils/muon_utils.py:41)
        o32 = MO.newton_schulz5(G, 6, exact=False)
        rel = (o.float() - o32).norm() / o32.norm()
        assert rel < 6e-2, rel                            # what bf16 rounding costs the reference itself (3-4 %)
