"""Every LDS-DMA staging stream reads inside the tensor it was given (r03 verdict item 7).

A stray READ of an LDS-DMA stream changes no result and faults only when it crosses into an unmapped page (that is how the r03
over-read of the GEMM's staging stream was found), so no parity test can see one.  The audit twin of the library
(libcm3p_hip_audit.so, built by cm3p_amd.build from the same sources with -DCM3P_DMA_AUDIT=1) records the lowest and highest byte
every staging helper reads per operand; tools/dma_audit.py runs the edge shapes through the C ABI (one process) and this test asserts
that every recorded address lay inside [base, base + bytes) of the corresponding tensor."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUDIT_LIB = os.path.join(ROOT, "cm3p_amd", "csrc", "libcm3p_hip_audit.so")


def test_the_shipped_library_has_no_audit_hooks():
    from cm3p_amd import _lib

    lib = _lib.load()
    assert lib.cm3p_build_ablation_flags() == 0
    assert lib.cm3p_debug_set_dma_audit(None) == -1  # CM3P_ERR_INVALID: nothing to switch on


@pytest.mark.gpu
def test_every_lds_dma_stream_stays_inside_its_operand():
    from cm3p_amd import build as B

    B.build_audit(verbose=False)  # opt-in twin (python -m cm3p_amd.build --audit): compiled here when missing or stale
    assert os.path.exists(AUDIT_LIB)
    env = dict(os.environ, CM3P_HIP_LIB=AUDIT_LIB, CM3P_ALLOW_ABLATED_LIB="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dma_audit.py")], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    bad = [r for r in rows if r.get("ok") is False]
    assert p.returncode == 0 and not bad, (bad[:5], p.stderr[-2000:])
    cases = [r for r in rows if "case" in r]
    assert len(cases) >= 40 and rows[-1] == {"failed": 0}
    # both big-shape GEMM kernels, both attention backward families and the packed-sequence forms were exercised
    names = " | ".join(r["case"] for r in cases)
    for needle in ("gemm 8p fwd", "gemm 256 fwd", "wgrad", "dgrad", "RoPE", "GeGLU", "batched", "sliding-window backward stage 1", "stage 2",
                   "fused global backward (even", "fused global backward (odd", "packed", "global forward (pipelined) B=", "global forward (pipelined), packed"):
        assert needle in names, needle
