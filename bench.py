#!/usr/bin/env python3
"""Headline benchmark: CM3P contrastive training step (dual-tower forward + in-batch CLIP loss + backward) on MI355X.

    python bench.py --gpus N --steps K --warmup W                    # any N; for N > 1 it starts its own ranks (child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                       # N > 1 under an external launcher, one rank per GPU over RCCL

A step is one pass of the hot path over one synthetic batch already resident in HBM: forward of both towers, projections,
L2 norm, logits, symmetric cross-entropy, full backward (optimizer excluded).  Workload at every N is
BASELINE.json configs[1] per GPU ("C2": default CM3P config, beatmap seq 4096 / metadata seq 256, bf16 GEMM operands,
batch 32 per GPU); for N > 1 the batch is sharded over ranks (weak scaling), embeddings are all-gathered for global
in-batch negatives and gradients are all-reduced (configs[2], "C3").  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      the dominant kernel = the C-ABI tag with the largest share of one fully bracketed step, no filter (every tag
                is ONE kernel, spelled like its rocprofv3 row), timed live with HIP events on the launching stream over the
                timed steps: algorithmic FLOPs per launch (SURVEY.md section 8d counting: attention backward = 2 x forward,
                recomputed scores not credited) / average launch duration vs the dense bf16 MFMA peak.  `traffic` and
                `mfma_busy` come from the committed rocprofv3 PMC passes of this same command (files named in the object).
  cpu_baseline  the CPU oracle (oracle/cm3p_oracle.py: the reference's fp32 sdpa path restated) timed on the host's cores on
                a bounded sample of the same workload (B = 2 at the real sequence lengths, 1 warm-up +
                2 timed iterations), rank 0 at N = 1 only.
  secondary     N = 1 only: BASELINE configs[3] ("C4": beatmap seq 8192, batch 16) timed for --secondary-steps (10) steps after 3
                warm-up steps behind the judged region, so the north-star target (fraction of bf16 MFMA peak on fwd+bwd at seq 8192)
                is a first-class measurement of the same run (r05 verdict item 5: it used to be 3 steps after 2).
  comm          N > 1 only: the step re-timed without the gradient all-reduce (DDP no_sync) and without the embedding
                all-gather -> exposed_allreduce_ms / exposed_allgather_ms per step (max over ranks).  These legs and the
                per-rank times run behind a deadline on rank 0 (CM3P_BENCH_DIAG_DEADLINE_S, 300 s): if a diagnostic collective
                never returns, the judged line - complete before any of them started - is printed without them.  The leg that
                needs a second DDP wrapper (bf16 all-reduce A/B) runs only with --diagnose.
  replicas      N > 1 only: after the warm-up every rank's parameter gradients are check-summed bit for bit (int32 view, 64-bit
                sum) and the checksums compared with one all-reduce: DDP must leave identical gradients on every rank, and the run
                aborts (non-zero exit, rank named) if it does not; per-rank peak device memory and the attention workspace size.
A rank that fails (process-group init, an RCCL error, a kernel error) prints `[bench] rank R: ...` and exits non-zero; the process
group carries a timeout, so the other ranks do not wait on a barrier for ever.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NOMINAL_CLOCK_GHZ = 2.4
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
PROFILE_EVERY = 3  # of the dominant kernel's launches inside the timed region, every third one is timed with HIP events
HBM_PEAK_GBS = 8000.0  # same guide: "HBM3E peak BW 8.0 TB/s spec" (6.29 TB/s measured with a float4 copy)

WORKLOADS = {
    # name: (per-GPU batch, beatmap seq, metadata seq, audio frames or None)
    "c2": dict(B=32, S=4096, L=256, audio_T=None, desc="default CM3P config, beatmap seq=4096, metadata seq=256, batch 32/GPU"),
    "c4": dict(B=16, S=8192, L=256, audio_T=None, desc="default CM3P config, beatmap seq=8192, metadata seq=256, batch 16/GPU"),
    "c5": dict(B=32, S=4096, L=256, audio_T=1600, desc="default CM3P config + audio-fused path (1600 mel frames), batch 32/GPU"),
    # not a BASELINE config: SURVEY.md section 8(f) rank 2, the published v7 recipe's step (ref:configs/train/v7.yaml:25-33) =
    # C2 shapes + CLS pooling + the MLM head on the beatmap tower with loss += 0.5 * masked-LM loss (15 % of positions labelled)
    "v7": dict(B=32, S=4096, L=256, audio_T=None, mlm=True, desc="C2 shapes with the v7 recipe: cls_embed pooling + MLM head (has_decoder_head, "
               "loss_type ForMaskedLM, masked_lm_prob 0.15), batch 32/GPU"),
}


def tower_flops_fwd(cfg, tokens: int, S: int) -> float:
    """Algorithmic matmul FLOPs of one encoder forward: per token per layer 8H^2 + 6HI for the four
    linears plus 4*k*H for attention with k = S (global) or min(S, 2*half_window+1) (local band at its true width)."""
    H, I, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    total = 0.0
    for i in range(L):
        k = S if cfg.is_global_layer(i) else min(S, 2 * cfg.half_window + 1)
        total += 8.0 * H * H + 6.0 * H * I + 4.0 * k * H
    return total * tokens


def step_flops(config, w) -> float:
    B, S, L = w["B"], w["S"], w["L"]
    f = tower_flops_fwd(config.beatmap_config, B * S, S) + tower_flops_fwd(config.metadata_config, B * L, L)
    if w["audio_T"]:
        T2 = w["audio_T"] // 2
        f += tower_flops_fwd(config.beatmap_config.audio_config, B * T2, T2)
    if w.get("mlm"):  # CM3PPredictionHead dense (H x H) + decoder (H x vocab) on every position
        bc = config.beatmap_config
        f += 2.0 * B * S * bc.hidden_size * (bc.hidden_size + bc.vocab_size)
    return 3.0 * f  # backward = 2 x forward, no recompute credit


def make_batch(config, w, rank: int, device):
    from cm3p_amd.synthetic import synthetic_batch

    batch = {k: v.to(device) for k, v in synthetic_batch(config, w["B"], w["S"], w["L"], seed=1234 + rank, audio_T=w["audio_T"]).items()}
    if w.get("padded"):
        # SURVEY.md section 8(d) "padded variant": per-row valid length ~ U{S/2 .. S}, right-padded with pad id 0
        g = torch.Generator().manual_seed(4321 + rank)
        S = w["S"]
        lens = torch.randint(S // 2, S + 1, (w["B"],), generator=g)
        lens[0] = S
        valid = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).to(device)
        batch["attention_mask"] = valid.to(batch["attention_mask"].dtype)
        batch["input_ids"] = batch["input_ids"] * valid.to(batch["input_ids"].dtype)
        w["valid_token_fraction"] = float(lens.sum()) / (w["B"] * S)
    if w.get("mlm"):
        # ref:configs/train/v7.yaml:31-33 (labels "masked_lm", masked_lm_prob 0.15): labelled positions carry the token id, the rest -100
        g = torch.Generator().manual_seed(987 + rank)
        pick = (torch.rand(w["B"], w["S"], generator=g) < 0.15).to(device)
        batch["labels"] = torch.where(pick & (batch["attention_mask"] != 0), batch["input_ids"], torch.full_like(batch["input_ids"], -100))
    return batch


def pmc_value(kind: str, workload: str, tag: str, field: str = ""):
    """-> (value, file) from the committed rocprofv3 PMC passes of this same command, or (None, None).
    kind "traffic": HBM bytes per launch of `tag` (profiles/traffic_<workload>.json, written by tools/pmc_traffic.py: FETCH_SIZE
    and WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950);
    kind "mfma_util": matrix-pipe busy fraction (profiles/mfma_util_<workload>.json, tools/pmc_mfma.py)."""
    rel = os.path.join("profiles", f"{kind}_{workload}.json")
    try:
        d = json.load(open(os.path.join(ROOT, rel)))
    except Exception:
        return None, None
    for key in (tag, tag.split(" [")[0], tag.split("<")[0]):  # exact tag, the kernel name without its " [global]" note, the bare name
        if key in d:
            v = d[key]
            if field:
                return (v.get(field) if isinstance(v, dict) else None), rel
            return ((v.get("busy_frac", v.get("mfma_util"))) if isinstance(v, dict) else v), rel
    return None, rel


def host_cores() -> int:
    """Threads this process may really use: affinity mask, capped by the cgroup CPU quota (a GPU box hands one GPU's
    share of a large host, 16 cores, to the job; spawning one thread per visible core would oversubscribe it)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("CM3P_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(workload: str) -> dict:
    """Time the CPU oracle on a bounded sample: B = 2 at the workload's real sequence lengths, fp32
    sdpa path, all host cores, 1 warm-up + 2 timed forward+backward iterations (B = 1 and 1 timed iteration for the 8192-token
    workload, whose iteration is ~3x longer)."""
    from oracle import cm3p_oracle as O

    w = WORKLOADS[workload]
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle on {cores} host threads ...", file=sys.stderr, flush=True)
    cfg = {}
    sd = {k: v.requires_grad_(v.dtype.is_floating_point) for k, v in O.init_state_dict(cfg, seed=0, with_audio=bool(w["audio_T"])).items()}
    Bc, iters = (2, 2) if w["S"] <= 4096 else (1, 1)
    batch = O.synthetic_batch(cfg, Bc, w["S"], w["L"], seed=1234, audio_T=w["audio_T"])

    def one():
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        O.forward(sd, cfg, **batch)["loss"].backward()
        return time.perf_counter() - t0

    warm = one()
    print(f"[bench] cpu_baseline: warm-up iteration {warm:.1f} s", file=sys.stderr, flush=True)
    times = []
    for _ in range(iters):
        times.append(one())
        print(f"[bench] cpu_baseline: timed iteration {times[-1]:.1f} s", file=sys.stderr, flush=True)
    dt = sum(times) / len(times)
    try:
        cpu = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        cpu = "unknown"
    return {
        "value": Bc / dt, "unit": "pairs/s", "cores": cores, "kind": "port", "cpu": cpu,
        "sample": f"forward+backward of B={Bc} at the same sequence lengths (beatmap {w['S']}, metadata {w['L']}), fp32 sdpa path, "
                  f"1 warm-up ({warm:.1f} s) + {iters} timed iteration(s) (mean {dt:.1f} s); the full B={w['B']} step does not fit host memory",
    }


def optimizer_leg(model, steps: int = 3):
    """The Muon step that follows the hot path, timed OUTSIDE the judged region and reported next to
    it (§8d: "optimizer excluded from the roofline figure, reported separately").  Parameter split as ref:train.py:331-340,
    hyper-parameters of the v7 recipe; uses the gradients the last timed step left behind."""
    from cm3p_amd import _lib
    from cm3p_amd.muon import Muon

    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and p.grad is not None]
    adamw = [p for n, p in named if any(k in n.lower() for k in ("embed", "proj_out")) or p.ndim <= 1]
    ids = {id(p) for p in adamw}
    muon = [p for _, p in named if id(p) not in ids]
    opt = Muon(muon_params=muon, lr=4e-4, adamw_params=adamw, adamw_lr=1e-4, adamw_betas=(0.9, 0.999), adamw_wd=0.0)
    opt.step()  # allocates state and workspaces
    torch.cuda.synchronize()
    _lib.profile_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    prof = _lib.profile_end()
    n_gemm, gemm_ms, gemm_flop = prof.get("muon_ns_gemm", (0, 0.0, 0.0))
    return {
        "kind": "muon (Newton-Schulz x6, batched over same-shaped weights) + multi-tensor AdamW rule",
        "ms": ms, "muon_matrices": len(muon), "adamw_tensors": len(adamw),
        "ns_gemm_tflop": gemm_flop / steps / 1e12, "ns_gemm_ms": gemm_ms / steps,
        "ns_gemm_tflops_achieved": gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else None,
        "kernels_ms": {k: round(v[1] / steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
        "launches_per_step": sum(v[0] for v in prof.values()) // steps,
    }


def hbm_rows(bracketed_step: dict) -> list:
    """The HBM-bound tags of one fully bracketed step: achieved GB/s on algorithmic bytes, as a fraction of the 8 TB/s of MI355X_MICROARCH.md."""
    from cm3p_amd import _lib

    rows = []
    for tag, (n, ms, work) in sorted(bracketed_step.items(), key=lambda kv: -kv[1][1]):
        if tag in _lib.HBM_BOUND_TAGS and work and ms > 0 and ms >= 0.5:
            gbs = work / (ms * 1e-3) / 1e9
            rows.append({"kernel": tag, "launches": n, "ms_per_step": round(ms, 3), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4)})
    return rows


def roofline_object(tag: str, timed_launches, bracketed_step: dict, workload: str) -> dict:
    """The `roofline` object of one workload: `timed_launches` = (launches, total ms, total algorithmic work) of the dominant tag as timed
    with HIP events inside the timed steps, `bracketed_step` = every tag of one fully bracketed step (for the kernel's share)."""
    from cm3p_amd import _lib

    n, ms, work = timed_launches
    total_ms = sum(v[1] for v in bracketed_step.values())
    hbm_bound = tag in _lib.HBM_BOUND_TAGS  # (their `work` is algorithmic bytes)
    peak, unit, scale = (HBM_PEAK_GBS, "GB/s", 1e9) if hbm_bound else (BF16_MFMA_PEAK_TFLOPS, "TFLOP/s", 1e12)
    achieved = work / (ms * 1e-3) / scale if ms > 0 else 0.0
    traffic, traffic_file = pmc_value("traffic", workload, tag)
    busy, busy_file = pmc_value("mfma_util", workload, tag)
    clock, clock_file = pmc_value("mfma_util", workload, tag, "clock_ghz")  # (GRBM_GUI_ACTIVE / 8 / duration of the same counter pass)
    held = {}
    if not hbm_bound and clock:
        # `frac` stays against the nominal peak (2.4 GHz); the part lowers its clock under matrix load (MI355X_MICROARCH.md "DVFS give-back",
        # an MFMA-only stream of the attention forward holds 1.94 GHz), so the same rate against the peak AT the clock
        # this kernel held is reported beside it - information, not the judged number.  The clock is NOT this run's: it comes from the
        # committed counter pass named in clock_source (another run of the same command, under the profiler, possibly another box; boxes
        # spread about +-1.5 %), so the quotient mixes two measurements and is labelled as such (r05 advisor).
        held = {"clock_ghz": round(clock, 3), "clock_source": clock_file,
                "clock_provenance": "GRBM_GUI_ACTIVE / 8 / kernel duration of a committed rocprofv3 counter pass of this command - a profiled run, not this one",
                "frac_at_held_clock": achieved / (peak * clock / NOMINAL_CLOCK_GHZ)}
    return {
        "kernel": tag, "bound": "hbm" if hbm_bound else "mfma", "achieved": achieved, "peak": peak, "unit": unit,
        "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_file,
        "mfma_busy": busy, "mfma_busy_source": busy_file, **held,
        "launches": n, "launches_sampled_every": PROFILE_EVERY, "avg_launch_ms": ms / n,
        "share_of_kernel_time": (ms / n) * bracketed_step[tag][0] / total_ms,
        "work_per_launch": work / n,
        "counting": "SURVEY.md 8(d): matmul FLOPs 2mnk, attention backward = 2 x forward (recomputed scores not credited)",
    }


def scaling_diagnosis(args, world, rank, device, model, step, timed, n, my_ms, ms_per_step, backend):
    """N > 1 only, all OUTSIDE the judged region and BEHIND the deadline of main() (a diagnostic that hangs cannot cost the run its judged
    line): what a single scaling run needs to say besides its one number (r04 verdict item 3).  The leg that builds a second DDP wrapper
    (b) runs only with --diagnose (r05 advisor: it had never executed over RCCL).
    The step is re-timed (a) with a surplus grid for the ring-kernel GEMMs - 1024 workgroups instead of one per CU, the arrangement
    that loses least while RCCL's channel workgroups hold CUs (+7..10 % per GEMM against +26..58 %) - and (b) with the
    gradient all-reduce in bf16 (a second DDP wrapper around the same module with bf16_compress_hook); together with the no_sync / rank-local
    legs of `comm` a reader can tell "all-reduce exposed" (a, b no better; exposed_allreduce_ms large) from "GEMMs starved of CUs by the
    communication kernels" (a better) from "bandwidth-bound all-reduce" (b better).  Also: per-rank step times (skew = one slow rank),
    RCCL's version, and the fact that N = 1 runs the metadata tower beside the beatmap tower while N > 1 does not."""
    import torch.distributed as dist

    from cm3p_amd import kernels as K

    out = {}
    t = torch.tensor([my_ms], device=device, dtype=torch.float64)
    ts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(ts, t)
    per_rank = [round(float(x.item()), 3) for x in ts]
    out["ms_per_step_per_rank"] = per_rank
    out["rank_skew_ms"] = round(max(per_rank) - min(per_rank), 3)
    # (the surplus-grid A/B of the ring-kernel GEMMs is taken in the warm-up, where its winner is selected: comm.gemm_grid)
    # bf16 gradient all-reduce (skipped when the judged run already used it); every rank takes the same path: an unsupported dtype
    # raises on all of them at the same collective
    if args.grad_compress == "none" and args.diagnose:
        try:
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks

            ddp16 = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index], gradient_as_bucket_view=True, bucket_cap_mb=32)
            ddp16.register_comm_hook(None, default_hooks.bf16_compress_hook)

            def step16():
                for p in model.parameters():
                    p.grad = None
                o = ddp16(**step.batch)
                o.loss.backward()

            ms_16 = timed(step16, n)
            del ddp16
            out["grad_compress_ab"] = {"ms_per_step_fp32_allreduce": ms_per_step, "ms_per_step_bf16_allreduce": ms_16, "delta_ms": ms_16 - ms_per_step}
        except Exception as e:  # noqa: BLE001
            out["grad_compress_ab"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    out["backend"] = backend
    try:
        out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception:  # noqa: BLE001
        out["rccl_version"] = None
    out["ranks_seen"] = dist.get_world_size()
    out["tower_overlap"] = {"n_gpus_1": "metadata tower on a second stream beside the beatmap tower (about -0.7 ms per C2 step)",
                            "this_run": "off (gathered negatives: the metadata tower's output is awaited by the all-gather)"}
    out["reading"] = ("exposed_allreduce_ms ~ 0 and gemm_grid.selected = one per CU: communication hidden, a scaling loss is elsewhere (rank_skew_ms: one slow rank); "
                      "gemm_grid.selected = 1024: the ring-kernel GEMMs were waiting for CUs RCCL holds, the judged region ran with the surplus grid; "
                      "grad_compress_ab.delta_ms < 0 with exposed_allreduce_ms > 0: the all-reduce is bandwidth-bound and exposed")
    return out


def launch_ranks(n: int) -> int:
    """Run this same command as n ranks under `python -m torch.distributed.run` (one rank per GPU, rendezvous on 127.0.0.1 and a
    free port) as a child process and return its exit code."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] --gpus {n} without WORLD_SIZE: launching {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool's driver
    env.setdefault("OMP_NUM_THREADS", "4")
    # rank 0's JSON line is the only thing that belongs on stdout; whatever else the ranks or their libraries print there (gloo's
    # "[Gloo] Rank 0 is connected ..." banner goes to stdout) is passed on to stderr, so that the caller can parse stdout as it does at N = 1
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        at = line.find('{"metric"')
        if at >= 0:
            if at:
                sys.stderr.write(line[:at] + "\n")
            sys.stdout.write(line[at:])
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
            sys.stderr.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="override the per-GPU batch (rehearsals only; the judged run uses the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing in the timed region")
    ap.add_argument("--no-optimizer", action="store_true", help="skip the separately reported Muon optimizer-step timing")
    ap.add_argument("--padded", action="store_true", help="not the judged configuration: right-padded rows, valid length ~ U{S/2..S}")
    ap.add_argument("--unpad", action="store_true", help="with --padded: run the beatmap tower on the valid tokens only (unpadded execution)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C4 (seq 8192) line reported next to the judged C2 number at N = 1")
    ap.add_argument("--secondary-steps", type=int, default=10, help="timed steps of the C4 (seq 8192) leg at N = 1 (after 3 warm-up steps)")
    ap.add_argument("--diagnose", action="store_true",
                    help="N > 1: also re-time the step through a second DDP wrapper with a bf16 all-reduce hook (scaling_diagnosis); off by default - "
                         "the judged run keeps to the legs that only reuse the judged region's own collectives")
    ap.add_argument("--grad-compress", default="none", choices=["none", "bf16"],
                    help="N > 1: DDP communication hook (bf16 halves the 545 MB fp32 gradient all-reduce; off in the judged run)")
    args = ap.parse_args()

    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher.  Nothing in this process has touched the GPU yet (importing
        # torch does not), and it never will: the ranks are CHILD processes of torch.distributed.run (never os.exec*), their
        # stdout / stderr pass through unchanged - rank 0's one JSON line included - and this process exits with their code.
        raise SystemExit(launch_ranks(args.gpus))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (`python bench.py --gpus N` does it by itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and cm3p_amd has no CPU path")
    # one rank per GPU; CM3P_BENCH_BACKEND=gloo lets several ranks share one card for rehearsing the N>1 code path
    backend = os.environ.get("CM3P_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rank that dies (or a collective that never completes) must end the job with an error, not hang the others on a barrier
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("CM3P_BENCH_PG_TIMEOUT_S", "600")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)

    from cm3p_amd import CM3PConfig, CM3PModel, _lib

    if os.environ.get("CM3P_BENCH_INJECT_FAILURE_RANK") == str(rank):  # tests/test_bench_gpu.py: the failure contract of run()
        raise RuntimeError("injected failure (CM3P_BENCH_INJECT_FAILURE_RANK)")
    w = dict(WORKLOADS[args.workload])
    if args.padded:
        w["padded"] = True
        w["desc"] += " [padded variant: valid length ~ U{S/2..S}" + (", unpadded execution]" if args.unpad else ", padded execution]")
    if args.batch:
        w["B"] = args.batch
        w["desc"] += f" [batch overridden to {args.batch}/GPU]"
    if w.get("mlm"):
        config = CM3PConfig(beatmap_config=dict(cls_embed=True), metadata_config=dict(cls_embed=True), has_decoder_head=True,
                            loss_type="ForMaskedLM")  # ref:configs/train/v7.yaml:25-33
    else:
        config = CM3PConfig(beatmap_config=dict(cls_embed=False), metadata_config=dict(cls_embed=False))  # ref:configs/model/default.yaml
    torch.manual_seed(0)
    model = CM3PModel(config).to(device).train()  # random init of the named architecture, fp32 master weights
    if not w["audio_T"]:
        for p in model.beatmap_model.audio_encoder.parameters():
            p.requires_grad_(False)
    step_model = model
    if world > 1:
        model.gather_negatives = True
        step_model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev_index], gradient_as_bucket_view=True,
                                                               bucket_cap_mb=32)  # per-layer autograd nodes: buckets fill (and reduce) while backward runs
        if args.grad_compress == "bf16":
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks

            step_model.register_comm_hook(None, default_hooks.bf16_compress_hook)
    batch = make_batch(config, w, rank, device)
    if args.unpad:
        model.unpad_inputs = True

    def step():
        for p in model.parameters():
            p.grad = None
        out = step_model(**batch)
        out.loss.backward()
        return out.loss

    step.batch = batch  # (scaling_diagnosis runs the same batch through a second DDP wrapper)
    for _ in range(args.warmup):
        step()

    replicas = None
    if world > 1:
        # Replica consistency: after a DDP step every rank holds the SAME averaged gradients, bit for bit.
        if args.warmup == 0:
            step()
        from cm3p_amd import kernels as _K
        from cm3p_amd.dist import replica_report

        replicas = replica_report(model.parameters(), device, float(torch.cuda.max_memory_allocated(device)),
                                  float(sum(t.numel() for t in _K._fused_ws.values())))
        print(f"[bench] rank {rank}: gradient checksum {replicas['gradient_checksum']} peak memory "
              f"{replicas['peak_memory_gb_per_rank'][rank]:.1f} GiB", file=sys.stderr, flush=True)
        if not replicas["identical_on_all_ranks"]:
            raise RuntimeError(f"replicas diverged: gradient checksums differ across ranks (min {replicas['checksum_min']}, max {replicas['checksum_max']})")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fn, n):
        """n calls of fn between fences -> ms per call, max over ranks (outside the judged region)."""
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / n * 1e3

    # N > 1: the grid of the ring-kernel GEMMs is chosen in the warm-up, by measurement (r04 verdict item 3).  One workgroup per CU is
    # right when the chip is theirs (N = 1: a surplus grid costs 0.6-1 %); beside RCCL's channel workgroups, which hold CUs while a
    # bucket's all-reduce runs, a workgroup that finds no CU starts late and still owns a full share of the work items (+26..58 % per
    # GEMM with 32 CUs held against +7..10 % with 1024 workgroups).  Two steps each way, max over ranks (the same
    # number on every rank, so every rank takes the same decision); the judged region then runs the faster one and `comm` says which.
    grid_choice = None
    if world > 1:
        from cm3p_amd import kernels as _K

        from cm3p_amd.dist import choose_gemm_grid

        g0 = _K.gemm8p_get_grid()
        ms_g0 = timed(step, 2)
        _K.gemm8p_set_grid(1024)
        ms_g1 = timed(step, 2)
        grid_choice = choose_gemm_grid(g0, ms_g0, ms_g1, backend)  # (both times are max-over-ranks: every rank takes the same decision)
        _K.gemm8p_set_grid(grid_choice["grid"])

    # Per-kernel timing.  One extra UNTIMED step with a HIP-event pair around every C-ABI call gives the breakdown and names the
    # dominant single kernel; inside the timed region only every third launch of that kernel is bracketed (every launch of every
    # kernel bracketed costs the stream ~5 ms per C2 step, all launches of the dominant one ~0.5 ms, which would be charged to the
    # judged number; a stride of 3 walks evenly through the five shapes per layer that share the dominant tag).
    profile = not args.no_profile
    prof_all, dom_tag = {}, None
    if profile:
        fence()
        _lib.profile_begin()
        step()
        prof_all = _lib.profile_end()
        # largest share of the step, no filter by kernel family: every tag is one kernel (tags that carry no algorithmic work
        # count - the packed-sequence attention launches, whose lengths live on the device - cannot be priced and are passed over)
        priced = {k: v for k, v in prof_all.items() if v[2]}
        dom_tag = max((priced or prof_all).items(), key=lambda kv: kv[1][1])[0]
    fence()
    if profile:
        _lib.profile_begin(only=dom_tag, every=PROFILE_EVERY)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed  # this rank's own clock over the judged region (the reported time is the max over ranks)
    prof = _lib.profile_end() if profile else {}
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    pairs_per_s = world * w["B"] * args.steps / elapsed
    flops = step_flops(config, w)

    def judged_line(extra=None):
        """The contract's keys: everything the judged region measured.  Built BEFORE any diagnostic leg runs."""
        d = {
            "metric": "contrastive training steps/sec (global beatmap-metadata pairs/sec)",
            "value": pairs_per_s,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "steps_per_s": 1e3 / ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {w['desc']}", "global_batch": world * w["B"], "beatmap_seq": w["S"],
                       "metadata_seq": w["L"], "parallelism": f"dp{world}" + ("+allgather-negatives" if world > 1 else ""),
                       "weights": "random init (reference init rules), fp32 master / bf16 GEMM operands", "loss": float(loss.item()),
                       **({"valid_token_fraction": w["valid_token_fraction"]} if "valid_token_fraction" in w else {})},
            "step_tflops_algorithmic": flops / 1e12,
            "step_mfma_frac": flops / (ms_per_step * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
        }
        if rank == 0 and prof:
            # dominant SINGLE kernel (tags of C-ABI calls that launch several kernels are listed in the breakdown only, so
            # that the figure can be checked against one row of the rocprofv3 --stats summary); its launches were timed with
            # HIP events inside the timed region, on the stream they run on
            d["roofline"] = roofline_object(dom_tag, prof[dom_tag], prof_all, args.workload)
            d["kernel_breakdown_ms_per_step"] = {k: round(v[1], 3) for k, v in sorted(prof_all.items(), key=lambda kv: -kv[1][1])[:12]}
            d["kernel_breakdown_source"] = "one extra untimed step with every launch bracketed by HIP events"
            # the streaming kernels against the HBM roofline (north star: "rocprof HBM GB/s ... vs gfx950 peak"): algorithmic bytes per launch
            # (what one pass must read and write) / the launch's duration in that bracketed step, both towers' launches averaged
            d["hbm_kernels"] = hbm_rows(prof_all)
        if replicas is not None:
            d["replicas"] = replicas
        d.update(extra or {})
        return d

    result = judged_line()
    # N > 1: everything from here to the print is diagnostics (more collectives, on a code path the judged region did not take).  A rank
    # that hangs in one of them must not cost the run its number: rank 0 arms a deadline; when it fires, the judged line goes out as it
    # stands (with comm.diagnosis_error saying so) and the process ends - the launcher then tears the other ranks down.
    deadline = None
    if world > 1 and rank == 0:
        import threading

        def _give_up():
            print(f"[bench] rank 0: diagnostics did not return within {deadline_s} s - printing the judged line without them", file=sys.stderr, flush=True)
            # (the text was made before the first diagnostic ran: nothing here may touch the GPU or a collective, which is what may be stuck)
            print(judged_text, flush=True)
            os._exit(0)

        deadline_s = int(os.environ.get("CM3P_BENCH_DIAG_DEADLINE_S", "300"))
        judged_text = json.dumps({**result, "comm": {"diagnosis_error": f"deadline of {deadline_s} s passed before the diagnostic legs returned",
                                                     "gemm_grid": grid_choice}})
        deadline = threading.Timer(deadline_s, _give_up)
        deadline.daemon = True
        deadline.start()
    comm = None
    if world > 1:
        # SURVEY.md section 8(d) step-time split: the same step without the gradient all-reduce (DDP no_sync), then also without
        # the embedding all-gather (rank-local negatives); differences = what each collective leaves exposed per step
        n_comm = max(2, min(args.steps, 5))

        def step_nosync():
            with step_model.no_sync():
                step()

        ms_nosync = timed(step_nosync, n_comm)
        model.gather_negatives = False
        ms_local = timed(step_nosync, n_comm)
        model.gather_negatives = True
        comm = {"ms_per_step": ms_per_step, "ms_per_step_no_allreduce": ms_nosync, "ms_per_step_no_allreduce_no_allgather": ms_local,
                "exposed_allreduce_ms": ms_per_step - ms_nosync, "exposed_allgather_ms": ms_nosync - ms_local,
                "gradient_bytes_per_step": sum(p.numel() * p.element_size() for p in model.parameters() if p.requires_grad),
                "grad_compress": args.grad_compress, "bucket_cap_mb": 32, "steps_per_leg": n_comm}
        comm["gemm_grid"] = grid_choice
        try:  # diagnostics must never cost the run its judged line (every rank takes the same path through it)
            comm.update(scaling_diagnosis(args, world, rank, device, model, step, timed, n_comm, elapsed_local / args.steps * 1e3, ms_per_step, backend))
        except Exception as e:  # noqa: BLE001
            comm["diagnosis_error"] = f"{type(e).__name__}: {e}"[:300]
    if not args.no_optimizer:
        opt_info = optimizer_leg(model)  # every rank steps (replicas must stay identical); rank 0 reports
        if rank == 0:
            result["optimizer_step"] = opt_info
    if deadline is not None:
        deadline.cancel()
    if rank == 0 and comm is not None:
        result["comm"] = comm
    if world == 1 and args.workload == "c2" and not args.no_secondary and not args.padded and not args.batch:
        # BASELINE configs[3] next to the judged number: the north-star target is quoted at seq 8192
        w4 = dict(WORKLOADS["c4"])
        del batch
        torch.cuda.empty_cache()
        batch = make_batch(config, w4, rank, device)
        step()
        prof4_all, dom4 = {}, None
        if profile:  # the same two-stage measurement as the judged workload: one fully bracketed step names the dominant kernel ...
            fence()
            _lib.profile_begin()
            step()
            prof4_all = _lib.profile_end()
            priced4 = {k: v for k, v in prof4_all.items() if v[2]}
            dom4 = max((priced4 or prof4_all).items(), key=lambda kv: kv[1][1])[0]
            fence()
            _lib.profile_begin(only=dom4, every=PROFILE_EVERY)  # ... whose launches are then timed live inside the timed steps
        for _ in range(2):  # with the first step above and the bracketed one: >= 3 warm-up steps at this shape
            step()
        n4 = max(1, args.secondary_steps)
        ms4 = timed(step, n4)
        prof4 = _lib.profile_end() if profile else {}
        f4 = step_flops(config, w4)
        result["secondary"] = {"workload": f"c4: {w4['desc']}", "steps": n4, "warmup": 3 + int(profile), "ms_per_step": ms4,
                               "value": w4["B"] / (ms4 * 1e-3),
                               "unit": "pairs/s", "step_tflops_algorithmic": f4 / 1e12,
                               "step_mfma_frac": f4 / (ms4 * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, "target_mfma_frac": 0.40,
                               "target_ms_per_step": f4 / (0.40 * BF16_MFMA_PEAK_TFLOPS * 1e12) * 1e3}
        if prof4.get(dom4):
            result["secondary"]["roofline"] = roofline_object(dom4, prof4[dom4], prof4_all, "c4")
            result["secondary"]["kernel_breakdown_ms_per_step"] = {k: round(v[1], 3) for k, v in sorted(prof4_all.items(), key=lambda kv: -kv[1][1])[:12]}
            result["secondary"]["hbm_kernels"] = hbm_rows(prof4_all)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not w.get("mlm"):  # (the oracle's timed leg covers the BASELINE workloads)
            del batch
            torch.cuda.empty_cache()
            result["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _arm_hang_report():
    """CM3P_BENCH_HANG_REPORT_S=n: after n seconds every thread's Python stack goes to stderr (a rank that sits in a collective the others
    never join says where), repeated every n seconds.  Off by default."""
    n = int(os.environ.get("CM3P_BENCH_HANG_REPORT_S", "0") or 0)
    if n > 0:
        import faulthandler

        faulthandler.dump_traceback_later(n, repeat=True, file=sys.stderr)


def run():
    """main() with the failure contract of a multi-rank job: say which rank failed and exit non-zero (the launcher then tears the
    other ranks down; the process-group timeout covers a rank that dies without a Python exception)."""
    rank = os.environ.get("RANK", "0")
    _arm_hang_report()
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 - report and die, whatever it was
        import traceback

        traceback.print_exc()
        print(f"[bench] rank {rank}: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        sys.stderr.flush()
        os._exit(3)  # no destructor may wait on a collective the other ranks will never join


if __name__ == "__main__":
    run()
