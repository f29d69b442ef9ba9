"""The driver's contract with bench.py: one JSON line on stdout with the agreed keys (run at a rehearsal batch so that it takes
seconds; the judged run uses the defaults)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                 ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (k, d.get(k))
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["unit"] == "pairs/s"
    assert d["value"] > 0 and abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]  # pairs/s = batch / step time
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] in ("TFLOP/s", "GB/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and ("traffic" in r)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str) and c["unit"] == "pairs/s"


def test_bench_two_ranks_share_the_card_over_gloo():
    """The N > 1 code path of bench.py (DDP wrapper, gathered negatives, barrier + max-over-ranks timing, rank 0 prints) rehearsed
    with two ranks on the one GPU of this box; the driver's runs use one rank per GPU over RCCL."""
    env = dict(os.environ, CM3P_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--no-optimizer", "--diagnose"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"].startswith("dp2")
    assert d["value"] > 0 and abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]  # whole-job pairs/s
    assert "cpu_baseline" not in d  # (rank 0 at N = 1 only)
    r = d["replicas"]  # every rank holds bit-identical gradients after a DDP step; memory is reported per rank
    assert r["identical_on_all_ranks"] is True and len(r["peak_memory_gb_per_rank"]) == 2 and min(r["peak_memory_gb_per_rank"]) > 0
    assert d["comm"]["gradient_bytes_per_step"] > 0
    # the scaling run explains itself (r04 verdict item 3): per-rank times, the measured grid choice, the bf16 all-reduce leg, what ran
    c = d["comm"]
    assert len(c["ms_per_step_per_rank"]) == 2 and c["rank_skew_ms"] >= 0 and c["ranks_seen"] == 2 and c["backend"] == "gloo"
    assert set(c["gemm_grid"]["candidates"]) == {"one per CU", "1024"} and c["gemm_grid"]["selected"] == "one per CU"  # (surplus grid only over RCCL)
    assert "delta_ms" in c["grad_compress_ab"], c["grad_compress_ab"]  # --diagnose: the bf16 all-reduce leg really ran (r05 advisor: "error" used to pass)
    assert "this_run" in c["tower_overlap"] and "reading" in c


def test_bench_starts_its_own_ranks_when_called_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's N = 1 command with another --gpus): the process
    must start its two ranks itself as child processes, relay rank 0's single JSON line and exit with their code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["CM3P_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-optimizer"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly one line on stdout: the launcher adds nothing of its own there
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["replicas"]["identical_on_all_ranks"] is True
    # and a failing rank's exit code comes back through the launcher
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "2", "--no-optimizer"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(env, CM3P_BENCH_INJECT_FAILURE_RANK="0", CM3P_BENCH_PG_TIMEOUT_S="60"))
    assert p.returncode != 0 and "[bench] rank 0: RuntimeError: injected failure" in p.stderr


# (r06: a four-rank rehearsal on the one card - `--gpus 4 --batch 1` over gloo - passed in 12 s on one box of the pool and crawled on the
# next, every rank minutes per step in its first gathered forward (CM3P_BENCH_HANG_REPORT_S stack dumps, gpurun_out/r6/hang4_*.err): four
# processes whose kernels each own every CU are time-sliced by the queue scheduler.  A test that can outlast the box's watchdog does not
# belong in the suite; world sizes above two are rehearsed on CPU tensors (tests/test_dist_gloo.py: gathered loss and gradients at
# (8, 1) and (8, 4), replica report and grid vote at world 8).)


def test_bench_prints_the_judged_line_when_a_diagnostic_never_returns():
    """The deadline around the N > 1 diagnostics: with it at 0 s rank 0 prints the judged line - complete before any diagnostic
    collective started - and leaves; the line carries everything the contract names and says why `comm` is short."""
    p = _torchrun(2, dict(CM3P_BENCH_BACKEND="gloo", CM3P_BENCH_DIAG_DEADLINE_S="0", CM3P_BENCH_PG_TIMEOUT_S="60"),
                  "--steps", "1", "--warmup", "1", "--batch", "2", "--no-optimizer", port="29551")
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, (lines, p.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["replicas"]["identical_on_all_ranks"] is True and "roofline" in d
    assert "deadline" in d["comm"]["diagnosis_error"]


def _torchrun(n, extra_env, *args, port="29543"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **extra_env)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", str(n), *args],
                          capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)


def test_bench_names_the_failing_rank_and_exits_non_zero():
    """A rank that fails must end the job with an error that names it - not leave the other ranks on a barrier."""
    p = _torchrun(2, dict(CM3P_BENCH_BACKEND="gloo", CM3P_BENCH_INJECT_FAILURE_RANK="1", CM3P_BENCH_PG_TIMEOUT_S="60"),
                  "--steps", "1", "--warmup", "1", "--batch", "2", "--no-optimizer", port="29545")
    assert p.returncode != 0
    assert "[bench] rank 1: RuntimeError: injected failure" in p.stderr, p.stderr[-2000:]
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")]  # no result line from a failed job


def test_bench_two_ranks_over_rccl_when_the_box_has_two_gpus():
    """First contact of the N > 1 path with RCCL: one rank per GPU, exactly as the driver launches it.  Skipped on a one-GPU box."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    p = _torchrun(2, {}, "--steps", "2", "--warmup", "1", "--batch", "4", "--no-optimizer", port="29547")
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["replicas"]["identical_on_all_ranks"] is True
    assert d["comm"]["exposed_allreduce_ms"] is not None
