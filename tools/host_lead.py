#!/usr/bin/env python3
"""How far does the host run ahead of the GPU in the C2 step?  Prints, per step, the host time to ISSUE the step (forward call, backward
call) and the GPU time of the step; and the host's lead at the end of the forward and of the backward (events).

    python tools/host_lead.py [--steps 6] [--workload c2]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--workload", default="c2")
    args = ap.parse_args()
    w = bench.WORKLOADS[args.workload]
    from cm3p_amd import CM3PConfig, CM3PModel

    torch.manual_seed(0)
    config = CM3PConfig()
    model = CM3PModel(config).to("cuda")
    batch = bench.make_batch(config, w, 0, torch.device("cuda"))

    def step():
        for p in model.parameters():
            p.grad = None
        t0 = time.perf_counter()
        out = model(**batch)
        t1 = time.perf_counter()
        out.loss.backward()
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    evs = []
    t_start = time.perf_counter()
    host = []
    for _ in range(args.steps):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
        host.append(step() + (time.perf_counter() - t_start,))
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
    t_issued = time.perf_counter() - t_start
    torch.cuda.synchronize()
    t_done = time.perf_counter() - t_start
    print(f"{args.steps} steps: host finished issuing after {t_issued * 1e3:.1f} ms, GPU finished after {t_done * 1e3:.1f} ms")
    for i, (f, b, t) in enumerate(host):
        print(f"  step {i}: host forward {f * 1e3:6.1f} ms  backward {b * 1e3:6.1f} ms  (issued at {t * 1e3:7.1f} ms)   GPU step {evs[i].elapsed_time(evs[i + 1]):6.1f} ms")


if __name__ == "__main__":
    main()
