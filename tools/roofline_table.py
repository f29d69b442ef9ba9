#!/usr/bin/env python3
"""Per-kernel roofline table AND per-family budget of one workload from the committed rocprofv3 passes under profiles/:
time per launch and per step (kernel stats), fabric bytes per launch (FETCH x 2 + WRITE), the rate they make, matrix-pipe busy and clock;
then the step cut into kernel families (GEMM / global attention / sliding-window attention / HBM-bound passes / tails) with each family's
algorithmic work, what it achieves against its roofline, and how far it is from the north-star step time (r05 verdict item 5: the next
reader takes the gap per family from a file instead of reconstructing it).
(Averages are over ALL launches of a row: the beatmap tower's and the 16 x smaller metadata tower's.)

    python tools/roofline_table.py [c2|c4] [steps in the trace = 8] [--round r06] [--json out.json]      -> markdown on stdout

A profiled step runs slower than the bench line of the same build (the profiler lowers the clock, MI355X_MICROARCH.md "DVFS give-back" (2)):
the budget therefore carries both the profiled sum and, when profiles/<round>_<workload>_bench.json is there, the same shares scaled to
the un-profiled step time.
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_TFLOPS, HBM_PEAK_GBS, TARGET_FRAC = 2500.0, 8000.0, 0.40

# SURVEY.md section 8(d) counting, beatmap tower of the default config (H 768, I 1152, 22 layers of which 8 global): per workload the
# tokens, the sequence length and the algorithmic TFLOP of one step (fwd + bwd = 3 x fwd) by family
SHAPES = {"c2": dict(T=32 * 4096, S=4096), "c4": dict(T=16 * 8192, S=8192)}


def algorithmic_tflop(wl: str) -> dict:
    T, S = SHAPES[wl]["T"], SHAPES[wl]["S"]
    H, I, L, G = 768, 1152, 22, 8
    lin = 3.0 * T * L * (8.0 * H * H + 6.0 * H * I)
    glob = 3.0 * T * G * 4.0 * S * H
    band = 3.0 * T * (L - G) * 4.0 * min(S, 129) * H
    return {"gemm": lin / 1e12, "attn_global": glob / 1e12, "attn_band": band / 1e12}


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def family(k: str) -> str:
    if k.startswith(("gemm8p_kernel", "gemm256_kernel", "gemm_bf16_kernel", "splitk_reduce", "reduce_many")):
        return "gemm"
    if k.startswith(("attn_fwd_g_kernel", "attn_bwd_fused_kernel", "attn_bwd_dq_reduce", "attn_bwd_prep", "attn_bwd_dq3", "attn_bwd_dkv3")):
        return "attn_global"
    if k.startswith(("attn_fwd_kernel", "attn_bwd_dq_kernel", "attn_bwd_dkv_kernel")):
        return "attn_band"
    if k.startswith(("layernorm_", "geglu_", "colsum_kernel")):
        return "hbm"
    return "tails"


FAMILY_NOTE = {
    "gemm": "bf16 MFMA; the fp32 + residual and the split-K instances are co-limited by their HBM epilogues (DESIGN section 4)",
    "attn_global": "bf16 MFMA (forward 2 products, backward 5 executed / 4 credited) + the dQ hand-off (prep, slabs, reduce: HBM)",
    "attn_band": "HBM for the bytes the pair moves (q / k / v / dO read twice in the backward), MFMA work negligible",
    "hbm": "HBM: LayerNorm forward / backward, GeGLU forward / backward (algorithmic bytes in bench.py's hbm_kernels)",
    "tails": "embedding, pooling, head, casts, index bookkeeping, torch glue",
}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = {sys.argv[i][2:]: sys.argv[i + 1] for i in range(1, len(sys.argv) - 1) if sys.argv[i].startswith("--")}
    wl = args[0] if args else "c2"
    steps = int(args[1]) if len(args) > 1 else 8
    rnd = opts.get("round", "r06")
    P = os.path.join(ROOT, "profiles")
    rows = list(csv.DictReader(open(os.path.join(P, f"{rnd}_{wl}_kernel_stats.csv"))))
    traffic = json.load(open(os.path.join(P, f"traffic_{wl}.json")))
    busy = json.load(open(os.path.join(P, f"mfma_util_{wl}.json")))
    print(f"### {wl}: kernels of one step ({rnd}, rocprofv3 --kernel-trace --stats over {steps} steps)\n")
    print("| kernel (rocprof row) | launches / step | µs / launch | ms / step | MB / launch (PMC) | TB/s | matrix-busy | clock GHz |")
    print("|---|---|---|---|---|---|---|---|")
    fam_ms: dict = {}
    fam_launch: dict = {}
    for r in rows:
        ms = float(r["TotalDurationNs"]) / steps / 1e6
        k = short(r["Name"])
        f = family(k)
        fam_ms[f] = fam_ms.get(f, 0.0) + ms
        fam_launch[f] = fam_launch.get(f, 0.0) + int(r["Calls"]) / steps
        if ms < 0.9:
            continue
        us = float(r["AverageNs"]) / 1e3
        tb = traffic.get(k)
        b = busy.get(k, {})
        print(f"| `{k}` | {int(r['Calls']) / steps:.0f} | {us:.1f} | {ms:.2f} | {tb / 1e6:.0f} | {tb / (us * 1e-6) / 1e12:.2f} | "
              + (f"{100 * b['mfma_util']:.1f} % | {b['clock_ghz']:.2f} |" if b.get("mfma_util") else "- | - |") if tb else
              f"| `{k}` | {int(r['Calls']) / steps:.0f} | {us:.1f} | {ms:.2f} | - | - | - | - |")

    # ---- per-family budget
    alg = algorithmic_tflop(wl)
    total_alg = sum(alg.values())
    target_ms = total_alg / (TARGET_FRAC * PEAK_TFLOPS) * 1e3
    prof_ms = sum(fam_ms.values())
    bench_ms = None
    try:
        bench = json.loads(open(os.path.join(P, f"{rnd}_{wl}_bench.json")).read().strip().splitlines()[-1])
        bench_ms = bench["ms_per_step"]
    except Exception:
        pass
    scale = (bench_ms / prof_ms) if bench_ms else 1.0
    print(f"\n### {wl}: the step by kernel family\n")
    print(f"Algorithmic work {total_alg:.1f} TFLOP per step (SURVEY.md section 8d); north-star bar {TARGET_FRAC:.0%} of {PEAK_TFLOPS:.0f} TFLOP/s = "
          f"**{target_ms:.1f} ms**; kernel time under the profiler {prof_ms:.1f} ms"
          + (f", un-profiled step {bench_ms:.1f} ms (shares below scaled by {scale:.3f})" if bench_ms else "") + ".\n")
    print("| family | launches / step | ms / step (profiled) | ms / step (scaled to the bench line) | algorithmic TFLOP | achieved PFLOP/s | of bf16 peak | "
          "ms at the bar (its share of the target) | distance to it | what bounds it |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    budget = {}
    for f in ("gemm", "attn_global", "attn_band", "hbm", "tails"):
        ms_p = fam_ms.get(f, 0.0)
        ms_s = ms_p * scale
        a = alg.get(f)
        # a family's share of the target: MFMA families pro rata of their algorithmic work at the step's 40 %; the others have no FLOPs to
        # be credited with, so they are the price the MFMA families must make room for (listed as is)
        if a:
            pf = a / (ms_s * 1e-3) / 1e3
            at_bar = a / total_alg * target_ms
            print(f"| {f} | {fam_launch.get(f, 0):.0f} | {ms_p:.2f} | {ms_s:.2f} | {a:.1f} | {pf:.2f} | {pf * 1e3 / PEAK_TFLOPS:.3f} | {at_bar:.1f} | "
                  f"{ms_s - at_bar:+.1f} | {FAMILY_NOTE[f]} |")
        else:
            at_bar = None
            print(f"| {f} | {fam_launch.get(f, 0):.0f} | {ms_p:.2f} | {ms_s:.2f} | - | - | - | 0 (no credited work) | {ms_s:+.1f} | {FAMILY_NOTE[f]} |")
        budget[f] = {"launches_per_step": fam_launch.get(f, 0.0), "ms_profiled": ms_p, "ms_scaled": ms_s, "algorithmic_tflop": a,
                     "ms_at_bar": at_bar, "distance_ms": ms_s - (at_bar or 0.0)}
    print(f"| **sum** | {sum(fam_launch.values()):.0f} | {prof_ms:.2f} | {prof_ms * scale:.2f} | {total_alg:.1f} | "
          f"{total_alg / (prof_ms * scale * 1e-3) / 1e3:.2f} | {total_alg / (prof_ms * scale * 1e-3) / PEAK_TFLOPS:.3f} | {target_ms:.1f} | "
          f"{prof_ms * scale - target_ms:+.1f} | |")
    if "json" in opts:
        with open(opts["json"], "w") as fh:
            json.dump({"workload": wl, "round": rnd, "target_ms": target_ms, "profiled_ms": prof_ms, "bench_ms": bench_ms, "families": budget}, fh, indent=1)


if __name__ == "__main__":
    main()
