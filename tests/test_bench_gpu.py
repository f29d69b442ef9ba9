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
                        "--no-optimizer"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"].startswith("dp2")
    assert d["value"] > 0 and abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]  # whole-job pairs/s
    assert "cpu_baseline" not in d  # (rank 0 at N = 1 only)
