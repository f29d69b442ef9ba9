"""CPU restatement of the reference's Muon optimizer step.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(cm3p_amd/muon.py) never does and has no CPU path.

Follows ref:utils/muon_utils.py:
  * newton_schulz5      <- zeropower_via_newtonschulz5, :35-57 (bf16 quintic iteration, coefficients :45)
  * muon_matrix_update  <- Muon.step, Muon branch, :150-176
  * adamw_like_update   <- Muon.step, "AdamW" branch, :178-203 (note :203 steps with lr/scale, not adamw_lr/scale; :202
                           decays with adamw_lr - kept as is)
  * routes_to_muon      <- the parameter routing of Muon.__init__, :101-124, and of ref:train.py:331-340

Parity: PINNED by tests/golden/muon_steps.safetensors, produced by running the reference's own Muon class on the CPU
(tests/golden/make_golden_muon.py); tests/test_muon_oracle.py checks this restatement against it.

`exact=True` rounds to bf16 after every tensor op exactly where the reference's bf16 tensors do; `exact=False` keeps the
whole iteration in fp32 ("what the arithmetic means"), which the tests use to show that the HIP path's single-rounding
epilogues are no farther from the fp32 iteration than the reference's own bf16 rounding is.
"""
from __future__ import annotations

import torch

Tensor = torch.Tensor

NS_COEFFS = (3.4445, -4.7750, 2.0315)  # ref:utils/muon_utils.py:45


def newton_schulz5(G: Tensor, steps: int, eps: float = 1e-7, exact: bool = True) -> Tensor:
    """Approximate U V^T of G = U S V^T.  Returns bf16 (exact) or fp32, shape of G."""
    assert G.ndim == 2
    a, b, c = NS_COEFFS
    if exact:
        X = G.to(torch.bfloat16)
        X = X / (X.norm() + eps)          # bf16 norm, bf16 sum, bf16 quotient (:47)
    else:
        X = G.to(torch.bfloat16).float()  # the bf16 cast of the input is part of the algorithm's definition
        X = X / (X.norm() + eps)
    tall = G.shape[0] > G.shape[1]
    if tall:
        X = X.T
    for _ in range(steps):
        A = X @ X.T
        B = b * A + (c * A) @ A           # `c * A @ A` parses as (c * A) @ A (:52)
        X = a * X + B @ X
    return X.T if tall else X


def muon_matrix_update(p: Tensor, g: Tensor, state: dict, lr: float, momentum: float, nesterov: bool, ns_steps: int,
                       exact: bool = True) -> None:
    """In place on p and state['momentum_buffer'] (created as zeros on first use, 2-D like the flattened gradient)."""
    g2 = g.reshape(g.shape[0], -1)
    if "momentum_buffer" not in state:
        state["momentum_buffer"] = torch.zeros_like(g2)
    buf = state["momentum_buffer"]
    buf.mul_(momentum).add_(g2)
    # without nesterov the reference orthogonalises the raw gradient, not the buffer (:163-164 rebinds g only in the
    # nesterov branch); kept as is
    u = g2.add(buf, alpha=momentum) if nesterov else g2
    o = newton_schulz5(u, ns_steps, exact=exact)
    o = o * max(1, o.shape[0] / o.shape[1]) ** 0.5     # bf16 * python float -> bf16 when exact
    p.add_(o.reshape(p.shape).to(p.dtype), alpha=-lr)


def adamw_like_update(p: Tensor, g: Tensor, state: dict, lr: float, adamw_lr_ratio: float, betas, eps: float, wd: float) -> None:
    if "step" not in state:
        state["step"] = 0
        state["moment1"] = torch.zeros_like(g)
        state["moment2"] = torch.zeros_like(g)
    state["step"] += 1
    t = state["step"]
    m1, m2 = state["moment1"], state["moment2"]
    m1.lerp_(g, 1 - betas[0])
    m2.lerp_(g.square(), 1 - betas[1])
    u = m1 / (eps + m2.sqrt())
    scale = (1 - betas[0] ** t) / (1 - betas[1] ** t) ** 0.5
    p.mul_(1 - lr * adamw_lr_ratio * wd)
    p.add_(u, alpha=-lr / scale)


def routes_to_muon(p: Tensor, listed_as_adamw: bool) -> bool:
    """Muon.__init__'s rule (:101-124): >= 2-D and fewer than 10000 rows, unless handed over as an adamw parameter."""
    return (not listed_as_adamw) and p.ndim >= 2 and p.shape[0] < 10000


def split_like_train_py(named_params) -> tuple[list, list]:
    """ref:train.py:331-340: names containing 'embed' or 'proj_out', and every <= 1-D tensor, go to AdamW."""
    adamw, muon = [], []
    for name, p in named_params:
        if any(k in name.lower() for k in ("embed", "proj_out")) or p.ndim <= 1:
            adamw.append((name, p))
        else:
            muon.append((name, p))
    return muon, adamw


def step(params: dict, grads: dict, states: dict, use_muon: dict, *, lr: float, momentum: float = 0.95, nesterov: bool = True,
         ns_steps: int = 6, adamw_lr: float | None = None, adamw_betas=(0.95, 0.95), adamw_eps: float = 1e-8, adamw_wd: float = 0.0,
         base_lr: float | None = None, exact: bool = True) -> None:
    """One optimizer step over name-keyed dicts (in place).  `base_lr` is the constructor's lr (it fixes
    adamw_lr_ratio = adamw_lr / base_lr, :87); `lr` is the group's current lr (a scheduler may have moved it)."""
    base_lr = lr if base_lr is None else base_lr
    ratio = (0.002 if adamw_lr is None else adamw_lr) / base_lr
    for name, p in params.items():
        g = grads.get(name)
        if g is None:
            continue
        st = states.setdefault(name, {})
        if use_muon[name]:
            muon_matrix_update(p, g, st, lr, momentum, nesterov, ns_steps, exact=exact)
        else:
            adamw_like_update(p, g, st, lr, ratio, adamw_betas, adamw_eps, adamw_wd)


def ns_flops(rows: int, cols: int, steps: int) -> float:
    """MFMA flops of the Newton-Schulz iteration for one rows x cols matrix."""
    s, l = min(rows, cols), max(rows, cols)
    return steps * (2.0 * s * s * l + 2.0 * s * s * s + 2.0 * s * s * l)


def orthogonality_defect(o: Tensor) -> float:
    """max |singular value - 1| of the update direction; the quintic lands in roughly [0.5, 1.5] by design (:38-43)."""
    sv = torch.linalg.svdvals(o.float())
    return float((sv - 1).abs().max())


__all__ = ["newton_schulz5", "muon_matrix_update", "adamw_like_update", "routes_to_muon", "split_like_train_py", "step",
           "ns_flops", "orthogonality_defect", "NS_COEFFS"]
