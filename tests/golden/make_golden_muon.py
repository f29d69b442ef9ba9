#!/usr/bin/env python3
"""Generate the Muon golden fixture by running the REFERENCE's own optimizer class on the CPU.

Build container only (needs /root/reference):   python tests/golden/make_golden_muon.py
Writes tests/golden/muon_steps.safetensors: for each variant, the parameters after every step and the final optimizer state.
The parameter split is the one ref:train.py:331-340 performs.
"""
from __future__ import annotations

import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

from muon_cases import N_STEPS, VARIANTS, gradients, initial_params  # noqa: E402

import utils.muon_utils as ref_muon  # noqa: E402  (the reference module)

assert ref_muon.__file__.startswith("/root/reference/"), ref_muon.__file__


def main():
    torch.set_num_threads(4)
    blob = {}
    for vname, hp in VARIANTS.items():
        params = {k: torch.nn.Parameter(v.clone()) for k, v in initial_params().items()}
        adamw = [p for n, p in params.items() if any(kw in n.lower() for kw in ("embed", "proj_out")) or p.ndim <= 1]
        adamw_ids = {id(p) for p in adamw}
        muon = [p for p in params.values() if id(p) not in adamw_ids]
        opt = ref_muon.Muon(muon_params=muon, lr=hp["lr"], momentum=hp["momentum"], nesterov=hp["nesterov"], ns_steps=hp["ns_steps"],
                            adamw_params=adamw, adamw_lr=hp["adamw_lr"], adamw_betas=hp["adamw_betas"], adamw_eps=hp["adamw_eps"],
                            adamw_wd=hp["adamw_wd"])
        for s in range(N_STEPS):
            for group in opt.param_groups:
                group["lr"] = hp["lrs"][s]
            for k, g in gradients(s).items():
                params[k].grad = None if g is None else g.clone()
            opt.step()
            for k, p in params.items():
                blob[f"{vname}.step{s}.{k}"] = p.detach().clone().contiguous()
        for k, p in params.items():
            st = opt.state[p]
            blob[f"{vname}.use_muon.{k}"] = torch.tensor([st["use_muon"]], dtype=torch.int64)
            for sk in ("momentum_buffer", "moment1", "moment2"):
                if sk in st:
                    blob[f"{vname}.state.{sk}.{k}"] = st[sk].detach().clone().contiguous()
            if "step" in st:
                blob[f"{vname}.state.step.{k}"] = torch.tensor([st["step"]], dtype=torch.int64)
        print(vname, "ok", sum(v.numel() for v in blob.values()))
    save_file(blob, os.path.join(HERE, "muon_steps.safetensors"))


if __name__ == "__main__":
    main()
