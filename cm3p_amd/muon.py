"""Muon optimizer step on MI355X - drop-in for the reference's `utils.muon_utils.Muon` (ref:utils/muon_utils.py:60-203).

Same constructor, same parameter routing, same per-parameter state keys (`use_muon`, `momentum_buffer`, `moment1`,
`moment2`, `step`), so `Trainer` checkpoints written by either implementation load into the other.  What differs is how a
step runs:

  * the reference walks the parameters one by one (per 2-D weight: ~8 small torch ops + 18 small GEMMs, ~6000 launches per
    step for the default model);
  * here same-shaped weights form a GROUP that is processed as one strided batch: one fused momentum / bf16-cast /
    sum-of-squares pass, one normalisation pass, then per Newton-Schulz iteration three batched MFMA GEMMs whose epilogues
    apply the polynomial (`A = X X^T`, `B = b A + c A A`, `X' = a X + B X`), and one apply pass.  Tall weights iterate on
    `X^T X` in place of transposing (the quintic is symmetric under transposition), so no transpose is ever materialised.
    All other parameters take the reference's AdamW-like rule in ONE multi-tensor launch.

Parameters, gradients and state stay in torch's own allocations; kernels reach them through device tables of addresses.
Reductions are fixed-order, so data-parallel ranks that hold identical (all-reduced) gradients compute bit-identical
updates - replicas stay in sync without a parameter broadcast, exactly as with the reference's replicated step.

GPU only: raises if a parameter is not an fp32 CUDA tensor.  There is no CPU or torch fallback.
"""
from __future__ import annotations

from typing import Generator

import torch

from . import _lib
from ._lib import call, query, stream

NS_A, NS_B, NS_C = 3.4445, -4.7750, 2.0315  # ref:utils/muon_utils.py:45
NS_EPS = 1e-7                                # ref:utils/muon_utils.py:35


def _up8(n: int) -> int:
    return (n + 7) // 8 * 8


class _Group:
    """Workspaces of one (rows, cols) group of `n` matrices.  bf16 images are zero-initialised: padding stays zero."""

    def __init__(self, n: int, rows: int, cols: int, device):
        self.n, self.rows, self.cols = n, rows, cols
        self.rp, self.cp = _up8(rows), _up8(cols)
        self.s = min(self.rp, self.cp)
        self.x_stride = self.rp * self.cp
        bf = dict(dtype=torch.bfloat16, device=device)
        self.X = torch.zeros(n, self.rp, self.cp, **bf)
        self.X2 = torch.zeros(n, self.rp, self.cp, **bf)
        self.A = torch.empty(n, self.s, self.s, **bf)
        self.B = torch.empty(n, self.s, self.s, **bf)
        self.nparts = query("cm3p_muon_partials", rows, cols)
        self.partials = torch.empty(n, self.nparts, dtype=torch.float32, device=device)


def newton_schulz_batched(ws: _Group, steps: int) -> torch.Tensor:
    """Runs `steps` quintic iterations on ws.X (normalised, bf16, [n, rp, cp]); returns the buffer holding the result."""
    n, rp, cp, s = ws.n, ws.rp, ws.cp, ws.s
    X, Y = ws.X, ws.X2
    st = stream()
    tall = rp > cp
    flops_xx = 2.0 * n * s * s * max(rp, cp)
    for _ in range(steps):
        xp, yp, ap, bp = X.data_ptr(), Y.data_ptr(), ws.A.data_ptr(), ws.B.data_ptr()
        if tall:
            # A = X^T X: both operands contraction-strided (the wgrad form); contraction over the rp rows
            call("cm3p_gemm_bf16_batched", xp, xp, ap, None, n, s, s, rp, cp, cp, s, ws.x_stride, ws.x_stride, s * s, 0, 0, 0,
                 1.0, 0.0, st, tag="muon_ns_gemm", work=flops_xx)
        else:
            # A = X X^T: both operands k-contiguous
            call("cm3p_gemm_bf16_batched", xp, xp, ap, None, n, s, s, cp, cp, cp, s, ws.x_stride, ws.x_stride, s * s, 0, 1, 1,
                 1.0, 0.0, st, tag="muon_ns_gemm", work=flops_xx)
        # B = b A + c A A   (A symmetric: A A = A A^T, k-contiguous form)
        call("cm3p_gemm_bf16_batched", ap, ap, bp, ap, n, s, s, s, s, s, s, s * s, s * s, s * s, s * s, 1, 1, NS_C, NS_B, st,
             tag="muon_ns_gemm", work=2.0 * n * s * s * s)
        if tall:
            # X' = a X + X B   (B symmetric: X B = X B^T): [rp, s] = X[rp, s] . B[s, s]^T
            call("cm3p_gemm_bf16_batched", xp, bp, yp, xp, n, rp, cp, s, cp, s, cp, ws.x_stride, s * s, ws.x_stride, ws.x_stride,
                 1, 1, 1.0, NS_A, st, tag="muon_ns_gemm", work=flops_xx)
        else:
            # X' = a X + B X: B k-contiguous, X contraction-strided (the dgrad form)
            call("cm3p_gemm_bf16_batched", bp, xp, yp, xp, n, rp, cp, s, s, cp, cp, s * s, ws.x_stride, ws.x_stride, ws.x_stride,
                 1, 0, 1.0, NS_A, st, tag="muon_ns_gemm", work=flops_xx)
        X, Y = Y, X
    return X


class Muon(torch.optim.Optimizer):
    """Muon - MomentUm Orthogonalized by Newton-schulz; argument meaning as ref:utils/muon_utils.py:72-89."""

    def __init__(self, muon_params, lr=0.004, momentum=0.95, nesterov=True, ns_steps=6,
                 adamw_params=None, adamw_lr=0.002, adamw_betas=(0.95, 0.95), adamw_eps=1e-8, adamw_wd=0):
        defaults = dict(lr=lr, momentum=momentum, nesterov=nesterov, ns_steps=ns_steps, adamw_lr_ratio=adamw_lr / lr,
                        adamw_betas=adamw_betas, adamw_eps=adamw_eps, adamw_wd=adamw_wd)
        if isinstance(muon_params, Generator):
            muon_params = list(muon_params)
        if isinstance(adamw_params, Generator):
            adamw_params = list(adamw_params)
        elif adamw_params is None:
            adamw_params = []
        muon_params, adamw_params = list(muon_params), list(adamw_params)
        super().__init__([*muon_params, *adamw_params], defaults)

        def each(params):
            if len(params) and isinstance(params[0], dict):
                for group in params:
                    yield from group["params"]
            else:
                yield from params

        # routing flags live in the state (ints, so they pickle with the checkpoint): ref:utils/muon_utils.py:101-124
        for p in each(muon_params):
            self.state[p]["use_muon"] = 1 if (p.ndim >= 2 and p.size(0) < 10000) else 0
        for p in each(adamw_params):
            self.state[p]["use_muon"] = 0

        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world_size = torch.distributed.get_world_size()
            self.rank = torch.distributed.get_rank()
        else:
            self.world_size, self.rank = 1, 0
        self._workspaces: dict = {}

    # ------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _require_gpu_fp32(p, g):
        if type(p.data) is not torch.Tensor or type(g) is not torch.Tensor:
            raise NotImplementedError("cm3p_amd Muon: DTensor / tensor-subclass parameters are not supported")
        if not (p.is_cuda and g.is_cuda):
            raise _lib.Cm3pHipError("cm3p_amd Muon runs on the GPU only; there is no CPU fallback")
        if p.dtype != torch.float32 or g.dtype != torch.float32:
            raise _lib.Cm3pHipError(f"cm3p_amd Muon needs fp32 parameters and gradients (got {p.dtype} / {g.dtype})")
        if not (p.is_contiguous() and g.is_contiguous()):
            raise _lib.Cm3pHipError("cm3p_amd Muon needs contiguous parameters and gradients")

    def _workspace(self, key, n, rows, cols, device) -> _Group:
        ws = self._workspaces.get(key)
        if ws is None or ws.n != n:
            ws = self._workspaces[key] = _Group(n, rows, cols, device)
        return ws

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()

        for gi, group in enumerate(self.param_groups):
            lr, momentum = group["lr"], group["momentum"]
            shape_groups: dict = {}
            adamw = []
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                self._require_gpu_fp32(p, g)
                state = self.state[p]
                if state["use_muon"] == 1:
                    rows, cols = g.shape[0], g.numel() // g.shape[0]
                    if "momentum_buffer" not in state:
                        state["momentum_buffer"] = torch.zeros((rows, cols), dtype=g.dtype, device=g.device)
                    shape_groups.setdefault((rows, cols), []).append((p, g, state["momentum_buffer"]))
                else:
                    if "step" not in state:
                        state["step"] = 0
                        state["moment1"] = torch.zeros_like(g)
                        state["moment2"] = torch.zeros_like(g)
                    state["step"] += 1
                    adamw.append((p, g, state))
            if not shape_groups and not adamw:
                continue
            device = (adamw[0][0] if adamw else next(iter(shape_groups.values()))[0][0]).device

            # launches below take raw addresses (no tensor to read the device from): make the parameters' GPU current
            with torch.cuda.device(device):
                self._step_group(gi, group, shape_groups, adamw, device, lr, momentum)
            # The kernels wrote the parameters through raw addresses: tell autograd (and every cache keyed on a parameter's version
            # counter - encoder._bf16_weight_cached keeps bf16 weight copies across forward-only calls) that they changed, as an
            # in-place torch op would have.
            for items in shape_groups.values():
                for p, _, _ in items:
                    _bump_version(p)
            for p, _, _ in adamw:
                _bump_version(p)
        return loss

    def _step_group(self, gi, group, shape_groups, adamw, device, lr, momentum):
        # every address table of this step in one host buffer -> one H2D copy
        table: list[int] = []
        layout = {}
        for key, items in shape_groups.items():
            for j, name in enumerate(("p", "g", "buf")):
                layout[(key, name)] = len(table)
                table.extend(it[j].data_ptr() for it in items)
        # the AdamW rule's bias-correction scale depends on the per-parameter step count: one launch per distinct count
        adamw_by_step: dict = {}
        for p, g, state in adamw:
            adamw_by_step.setdefault(state["step"], []).append((p, g, state))
        for t, items in adamw_by_step.items():
            for name, col in (("p", [p.data_ptr() for p, _, _ in items]), ("g", [g.data_ptr() for _, g, _ in items]),
                              ("m1", [s["moment1"].data_ptr() for _, _, s in items]),
                              ("m2", [s["moment2"].data_ptr() for _, _, s in items]), ("n", [p.numel() for p, _, _ in items])):
                layout[("adamw", t, name)] = len(table)
                table.extend(col)
        dev_table = torch.tensor(table, dtype=torch.int64).to(device, non_blocking=True)
        base = dev_table.data_ptr()

        # Groups whose GEMMs fill the chip (the 256 x 256 ring kernel's rule in csrc/gemm.hip) run on the caller's stream; the small
        # ones (metadata tower, projections: a few workgroups per launch, ~110 launches) and the AdamW tensors run beside them on a
        # second stream - the groups are independent, the small launches fit into the big GEMMs' partial last waves.  Joined below.
        def fills_chip(key, n):
            s_ = min(_up8(key[0]), _up8(key[1]))
            return n * ((s_ + 255) // 256) ** 2 >= 128

        main_s = torch.cuda.current_stream(device)
        side_s = self._side_stream(device)
        side_s.wait_stream(main_s)
        ordered = sorted(shape_groups.items(), key=lambda kv: not fills_chip(kv[0], len(kv[1])))
        for key, items in ordered:
            with torch.cuda.stream(main_s if fills_chip(key, len(items)) else side_s):
                self._muon_group(gi, group, key, items, layout, base, device, lr, momentum)
        with torch.cuda.stream(side_s):
            self._adamw_tensors(group, adamw_by_step, layout, base, lr)
        main_s.wait_stream(side_s)
        # dev_table must outlive the launches above: they are stream-ordered (the second stream is joined) before any later reuse of
        # its memory by torch's caching allocator on the caller's stream.

    def _side_stream(self, device):
        streams = self.__dict__.setdefault("_side_streams", {})
        if device not in streams:
            streams[device] = torch.cuda.Stream(device)
        return streams[device]

    def _muon_group(self, gi, group, key, items, layout, base, device, lr, momentum):
        st = stream()
        rows, cols = key
        n = len(items)
        ws = self._workspace((gi, key), n, rows, cols, device)
        aligned = all(it[1].data_ptr() % 16 == 0 and it[2].data_ptr() % 16 == 0 for it in items)
        numel = rows * cols
        call("cm3p_muon_momentum", base + 8 * layout[(key, "g")], base + 8 * layout[(key, "buf")], ws.X.data_ptr(),
             ws.partials.data_ptr(), n, rows, cols, ws.cp, ws.x_stride, float(momentum), int(bool(group["nesterov"])),
             int(aligned), st, tag="muon_momentum", work=14.0 * n * numel)
        call("cm3p_muon_normalize", ws.X.data_ptr(), ws.partials.data_ptr(), n, rows, cols, ws.x_stride, NS_EPS, st,
             tag="muon_normalize", work=4.0 * n * ws.x_stride)
        out = newton_schulz_batched(ws, int(group["ns_steps"]))
        call("cm3p_muon_apply", base + 8 * layout[(key, "p")], out.data_ptr(), n, rows, cols, ws.cp, ws.x_stride,
             float(max(1, rows / cols) ** 0.5), float(-lr), st, tag="muon_apply", work=10.0 * n * numel)
        if out is ws.X2:  # keep "X holds the next step's input, X2 is scratch" (odd iteration counts swap them)
            ws.X, ws.X2 = ws.X2, ws.X

    def _adamw_tensors(self, group, adamw_by_step, layout, base, lr):
        st = stream()
        b1, b2 = group["adamw_betas"]
        adamw_lr = lr * group["adamw_lr_ratio"]
        for t, items in adamw_by_step.items():
            scale = (1 - b1 ** t) / (1 - b2 ** t) ** 0.5
            call("cm3p_adamw_multi", base + 8 * layout[("adamw", t, "p")], base + 8 * layout[("adamw", t, "g")],
                 base + 8 * layout[("adamw", t, "m1")], base + 8 * layout[("adamw", t, "m2")], base + 8 * layout[("adamw", t, "n")],
                 len(items), max(p.numel() for p, _, _ in items), float(1 - b1), float(1 - b2), float(group["adamw_eps"]),
                 float(1 - adamw_lr * group["adamw_wd"]), float(-lr / scale), st, tag="adamw_multi",
                 work=28.0 * sum(p.numel() for p, _, _ in items))


def _bump_version(p: torch.Tensor) -> None:
    """What any in-place torch op does to its output's version counter; a raw-pointer writer has to do it by hand."""
    try:
        torch.autograd.graph.increment_version(p)
    except AttributeError:  # older torch
        torch._C._increment_version(p)


__all__ = ["Muon", "newton_schulz_batched"]
