"""Builds libcm3p_hip.so (the C-ABI kernel library) in-tree with hipcc for gfx950.

No torch involvement: plain `hipcc -shared -fPIC`.  The .so lands next to the sources so that it travels to the GPU
box with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["norm.hip", "elementwise.hip", "gemm.hip", "gemm256.hip", "gemm8p.hip", "attention.hip", "attention_fwd.hip", "attention_bwd.hip", "attention_bwd_fused.hip", "attention_generic.hip", "head.hip", "conv.hip", "muon.hip"]
LIB = os.path.join(CSRC, "libcm3p_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]
# hand-scheduled kernels: SLP-packing adjacent f32 multiplies into v_pk_mul_f32 costs register shuffles (v_mov / v_perm /
# v_alignbit) around the bf16 packs and packed f32 VALU is slower beside MFMAs (MI355X_MICROARCH.md, cycle constants)
EXTRA_FLAGS = {"attention_fwd.hip": ["-fno-slp-vectorize"], "attention_bwd.hip": ["-fno-slp-vectorize"], "attention_bwd_fused.hip": ["-fno-slp-vectorize"]}


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, check_isa: bool = True, audit: bool = False) -> str:
    """Compile what is stale and link.  Objects whose correctness depends on the emitted instruction pattern (counted vmcnt rings,
    inline-asm MFMAs: isa_check.CHECKS) are re-checked every time they are recompiled; a failed check fails the build, so a
    different hipcc or flag set cannot silently ship a kernel whose waits no longer cover its loads."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "attn_common.h"), os.path.join(CSRC, "..", "..", "include", "cm3p_hip.h")]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    checks = []
    if check_isa:
        from . import isa_check

        for src, _ in procs:
            if src in isa_check.CHECKS:
                cmd = [hipcc, *[f for f in FLAGS if f != "-Wall"], *EXTRA_FLAGS.get(src, []), "-S", "--cuda-device-only", "-o", "-", os.path.join(CSRC, src)]
                checks.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    for src, p in checks:
        isa, _ = p.communicate()
        try:
            if p.returncode != 0:
                raise RuntimeError(f"hipcc -S failed on {src}")
            isa_check.CHECKS[src](isa)
        except Exception:
            os.remove(os.path.join(CSRC, src.replace(".hip", ".o")))  # never link (or leave behind) an unchecked object
            raise
        if verbose:
            print(f"ISA pattern of {src}: ok", flush=True)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if audit:  # the debug twin is opt-in (r04 advisor): `python -m cm3p_amd.build --audit`, __graft_entry__.build(), the audit test's fixture
        build_audit(force=force, verbose=verbose)
    return LIB


# The bounds-audit twin of the library (csrc/common.h, CM3P_DMA_AUDIT): the objects that stage operands by LDS-DMA are compiled a second
# time with the recording hooks in; everything else is shared with the shipped library.  Debug artefact, loaded only by
# tests/test_dma_audit_gpu.py (cm3p_build_ablation_flags() is non-zero for it, so _lib.load() refuses it as a product library).
AUDIT_SOURCES = ["gemm.hip", "gemm256.hip", "gemm8p.hip", "attention.hip", "attention_fwd.hip", "attention_bwd_fused.hip"]
AUDIT_LIB = os.path.join(CSRC, "libcm3p_hip_audit.so")


def build_audit(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "attn_common.h"), os.path.join(CSRC, "..", "..", "include", "cm3p_hip.h")]
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if src not in AUDIT_SOURCES:
            objs.append(os.path.join(CSRC, src.replace(".hip", ".o")))
            continue
        o = os.path.join(CSRC, src.replace(".hip", ".audit.o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc, *[f for f in FLAGS if f != "-Wall"], *EXTRA_FLAGS.get(src, []), "-DCM3P_DMA_AUDIT=1", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src} (audit build)")
    if force or procs or _stale(AUDIT_LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", AUDIT_LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return AUDIT_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, audit="--audit" in sys.argv)
