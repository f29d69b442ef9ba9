"""Muon optimizer step on the GPU (cm3p_amd/muon.py through the C ABI) against the reference-made fixture and the oracle.

Tolerances (floating point, stated here):
  * AdamW-branch parameters and all optimizer state: rtol 2e-6 / atol 1e-7 - same fp32 formula, same operation order.
  * Muon-branch parameters: the Newton-Schulz iteration runs in bf16 in the reference (every tensor op rounds) and with
    fp32 accumulators + one rounding per fused epilogue here, so the two differ by bf16 noise that the iteration carries
    along.  The reference's own distance to the same iteration in fp32 is 3-4 % (Frobenius, of the step); the bar is
      (1) ||step_hip - step_ref||_F <= 6 % of ||step_ref||_F, and
      (2) HIP is no farther from the fp32 iteration than 1.25 x the reference is.
"""
from __future__ import annotations

import copy
import os
import sys

import pytest
import torch
from safetensors.torch import load_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from muon_cases import N_STEPS, SHAPES, VARIANTS, gradients, initial_params  # noqa: E402

from oracle import muon_oracle as MO  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = load_file(os.path.join(HERE, "golden", "muon_steps.safetensors"))


def _split(params):
    muon, adamw = MO.split_like_train_py(list(params.items()))
    return [p for _, p in muon], [p for _, p in adamw]


def _make(variant, device="cuda"):
    from cm3p_amd.muon import Muon

    hp = VARIANTS[variant]
    params = {k: torch.nn.Parameter(v.clone().to(device)) for k, v in initial_params().items()}
    muon, adamw = _split(params)
    opt = Muon(muon_params=muon, lr=hp["lr"], momentum=hp["momentum"], nesterov=hp["nesterov"], ns_steps=hp["ns_steps"],
               adamw_params=adamw, adamw_lr=hp["adamw_lr"], adamw_betas=hp["adamw_betas"], adamw_eps=hp["adamw_eps"], adamw_wd=hp["adamw_wd"])
    return hp, params, opt


def _run_step(hp, params, opt, s):
    for group in opt.param_groups:
        group["lr"] = hp["lrs"][s]
    for k, g in gradients(s).items():
        params[k].grad = None if g is None else g.to(params[k].device)
    opt.step()


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_steps_match_the_reference_fixture(variant):
    hp, params, opt = _make(variant)
    # the same trajectory with the Newton-Schulz iteration in fp32 (oracle, exact=False), for bar (2)
    p32 = {k: v.clone() for k, v in initial_params().items()}
    st32: dict = {}
    named = list(initial_params().items())
    listed = {n for n, _ in MO.split_like_train_py(named)[1]}
    use = {n: MO.routes_to_muon(p, n in listed) for n, p in named}
    prev = {k: v.clone() for k, v in initial_params().items()}
    for s in range(N_STEPS):
        _run_step(hp, params, opt, s)
        torch.cuda.synchronize()
        # fp32 trajectory restarts from the reference's previous parameters so that only this step's arithmetic differs
        p32 = {k: prev[k].clone() for k in prev}
        # (st32's momentum buffers are fp32 in every implementation and follow the same recurrence, so they carry over)
        MO.step(p32, gradients(s), st32, use, lr=hp["lrs"][s], base_lr=hp["lr"], momentum=hp["momentum"], nesterov=hp["nesterov"],
                ns_steps=hp["ns_steps"], adamw_lr=hp["adamw_lr"], adamw_betas=hp["adamw_betas"], adamw_eps=hp["adamw_eps"],
                adamw_wd=hp["adamw_wd"], exact=False)
        for k in SHAPES:
            want = GOLD[f"{variant}.step{s}.{k}"]
            got = params[k].detach().cpu()
            if gradients(s)[k] is None:
                assert torch.equal(got, prev[k]), k          # no gradient -> untouched
            elif use[k]:
                step_ref, step_hip, step_f32 = want - prev[k], got - prev[k], p32[k] - prev[k]
                d_ref = (step_hip - step_ref).norm() / step_ref.norm()
                assert d_ref <= 6e-2, (k, s, float(d_ref))
                hip_to_f32 = (step_hip - step_f32).norm() / step_f32.norm()
                ref_to_f32 = (step_ref - step_f32).norm() / step_f32.norm()
                assert hip_to_f32 <= 1.25 * ref_to_f32 + 1e-3, (k, s, float(hip_to_f32), float(ref_to_f32))
            else:
                torch.testing.assert_close(got, want, rtol=2e-6, atol=1e-7, msg=lambda m: f"{k} step {s}: {m}")
        # continue from the reference's parameters: errors of earlier steps must not mask later ones
        with torch.no_grad():
            for k in SHAPES:
                prev[k] = GOLD[f"{variant}.step{s}.{k}"].clone()
                params[k].copy_(prev[k])
    for k, p in params.items():
        st = opt.state[p]
        assert st["use_muon"] == int(GOLD[f"{variant}.use_muon.{k}"].item())
        for sk in ("momentum_buffer", "moment1", "moment2"):
            key = f"{variant}.state.{sk}.{k}"
            if key in GOLD:
                assert st[sk].shape == GOLD[key].shape, (k, sk)
                torch.testing.assert_close(st[sk].cpu(), GOLD[key], rtol=2e-6, atol=1e-7)
        key = f"{variant}.state.step.{k}"
        if key in GOLD:
            assert st["step"] == int(GOLD[key].item())


def _bf16_mm_ref(a, b):
    return a.float() @ b.float()


@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
@pytest.mark.parametrize("dims", [(3, 40, 24, 56), (2, 128, 256, 64), (1, 200, 136, 72), (16, 768, 1152, 192), (34, 504, 488, 128)])
def test_batched_gemm_axpby(form, dims):
    """cm3p_gemm_bf16_batched against fp32 matmul of the same bf16 operands: |err| <= bf16 rounding of the result.  The last two
    cases have >= 128 tiles of 256 x 256 and go to the half-tile-ring kernel (gemm8p.hip, one work item per matrix and tile;
    partial edge tiles, an odd and an even k-tile count); the rest stay on the 128 x 128 kernel."""
    from cm3p_amd import _lib

    n, M, N, K = dims
    g = torch.Generator().manual_seed(M * 31 + N)
    a = torch.randn(n, M, K, generator=g).bfloat16().cuda()
    b = torch.randn(n, N, K, generator=g).bfloat16().cuda()
    r = torch.randn(n, M, N, generator=g).bfloat16().cuda()
    alpha, beta = 0.75, -1.5
    want = alpha * torch.einsum("bmk,bnk->bmn", a.float(), b.float()) + beta * r.float()
    if form == "nt":
        A, B, a_kc, b_kc, lda, ldb = a, b, 1, 1, K, K
    elif form == "nn":
        A, B, a_kc, b_kc, lda, ldb = a, b.transpose(1, 2).contiguous(), 1, 0, K, N
    else:
        A, B, a_kc, b_kc, lda, ldb = a.transpose(1, 2).contiguous(), b.transpose(1, 2).contiguous(), 0, 0, M, N
    c = torch.empty(n, M, N, dtype=torch.bfloat16, device="cuda")
    _lib.call("cm3p_gemm_bf16_batched", A.data_ptr(), B.data_ptr(), c.data_ptr(), r.data_ptr(), n, M, N, K, lda, ldb, N,
              A[0].numel(), B[0].numel(), M * N, M * N, a_kc, b_kc, alpha, beta, _lib.stream())
    torch.cuda.synchronize()
    err = (c.float() - want).abs()
    assert float((err - (want.abs() * 2 ** -8 + 1e-3)).max()) <= 0, float(err.max())


def test_model_sized_group_against_the_oracle():
    """Two 2304 x 768 (tall) and two 768 x 1152 (wide) weights - the Wqkv/Wi and MLP-Wo shapes of the default model."""
    from cm3p_amd.muon import Muon

    g = torch.Generator().manual_seed(5)
    shapes = {"a.Wqkv.weight": (2304, 768), "b.Wi.weight": (2304, 768), "a.Wo.weight": (768, 1152), "b.Wo.weight": (768, 1152)}
    init = {k: torch.randn(s, generator=g) * 0.02 for k, s in shapes.items()}
    grads = {k: torch.randn(s, generator=g) * 1e-3 for k, s in shapes.items()}
    params = {k: torch.nn.Parameter(v.clone().cuda()) for k, v in init.items()}
    opt = Muon(muon_params=list(params.values()), lr=4e-4)
    for k, p in params.items():
        p.grad = grads[k].cuda()
    opt.step()
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for k in shapes:
        ref, f32 = init[k].clone(), init[k].clone()
        MO.muon_matrix_update(ref, grads[k], {}, 4e-4, 0.95, True, 6, exact=True)
        MO.muon_matrix_update(f32, grads[k], {}, 4e-4, 0.95, True, 6, exact=False)
        step_hip, step_ref, step_f32 = params[k].detach().cpu() - init[k], ref - init[k], f32 - init[k]
        d = (step_hip - step_ref).norm() / step_ref.norm()
        assert d <= 6e-2, (k, float(d))
        assert (step_hip - step_f32).norm() <= 1.25 * (step_ref - step_f32).norm() + 1e-3 * step_f32.norm(), k


def test_step_is_deterministic_and_checkpointable():
    hp, params, opt = _make("main")
    _run_step(hp, params, opt, 0)
    sd = copy.deepcopy(opt.state_dict())  # state_dict() hands out the live tensors
    snap = {k: p.detach().clone() for k, p in params.items()}

    _run_step(hp, params, opt, 1)
    after_a = {k: p.detach().clone() for k, p in params.items()}

    # a fresh optimizer restored from the checkpoint, same parameters, same gradients -> the same bits
    hp2, params2, opt2 = _make("main")
    with torch.no_grad():
        for k in params2:
            params2[k].copy_(snap[k])
    opt2.load_state_dict(sd)
    _run_step(hp2, params2, opt2, 1)
    torch.cuda.synchronize()
    for k in params:
        assert torch.equal(after_a[k], params2[k].detach()), k


def test_refuses_cpu_parameters():
    from cm3p_amd import _lib
    from cm3p_amd.muon import Muon

    p = torch.nn.Parameter(torch.randn(16, 8))
    opt = Muon(muon_params=[p], lr=0.01)
    p.grad = torch.randn(16, 8)
    with pytest.raises(_lib.Cm3pHipError):
        opt.step()
