"""Compile-time checks of the generated ISA for the kernels whose CORRECTNESS depends on what hipcc emits (no GPU needed).

Four files synchronise LDS-DMA rings with counted `s_waitcnt vmcnt(N)` and / or issue inline-asm MFMAs that hipcc pads no
hazards for: attention_fwd.hip (r05), attention_bwd.hip, attention_bwd_fused.hip and gemm8p.hip.  A different hipcc, other flags or an innocent source edit
can add a spill reload, a scratch access or an accumulator copy to their loops; the counted waits then cover the wrong loads and
the results are silently wrong.  `build.py` runs these checks whenever it recompiles one of the files (a failed check fails the
build) and tests/test_kernel_isa.py runs them on every test run.
"""
from __future__ import annotations

import os
import re
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


class IsaCheckError(AssertionError):
    pass


def _need(cond, msg):
    if not cond:
        raise IsaCheckError(msg)


def compile_isa(src: str, flags: list[str], hipcc: str | None = None) -> str:
    hipcc = hipcc or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *[f for f in flags if f != "-Wall"], "-S", "--cuda-device-only", "-o", "-", os.path.join(CSRC, src)]
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def kernel_bodies(isa: str, name: str) -> dict[str, str]:
    """mangled symbol -> text of every kernel whose mangled name contains `name` (comment lines and inline-asm markers removed)."""
    isa = re.sub(r"^\s*;.*\n", "", isa, flags=re.M)
    out = {}
    for m in re.finditer(r"^(_ZN\S*" + re.escape(name) + r"\S*):", isa, re.M):
        out[m.group(1)] = isa[m.start():isa.index(".Lfunc_end", m.start())]
    return out


def no_scratch(isa: str, what: str):
    for key in ("private_segment_fixed_size", "vgpr_spill_count"):
        for m in re.finditer(r"\." + key + r":\s*(\d+)", isa):
            _need(int(m.group(1)) == 0, f"{what}: {key} = {m.group(1)} (a spilled register turns ring loads into synchronous round trips "
                                         "and breaks the counted vmcnt waits)")


def loops(body: str):
    """(label, text) of every backward branch target .. branch segment."""
    labels = {m.group(1): m.start() for m in re.finditer(r"^(\.LBB\d+_\d+):", body, re.M)}
    for m in re.finditer(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", body):
        if m.group(1) in labels and labels[m.group(1)] < m.start():
            yield m.group(1), body[labels[m.group(1)]:m.start()]


# ------------------------------------------------------------------------------------------------ gemm8p.hip
def check_gemm8p(isa: str):
    """The half-tile ring of gemm8p.hip: per k-tile exactly 8 LDS-DMA instructions (4 half-tiles x 2 pieces per wave), 8 raw
    barriers, 64 MFMAs and ONE vector-memory wait, `s_waitcnt vmcnt(6)`, right in front of a barrier; no scratch anywhere.
    Plain instances (one k-tile per loop iteration): the B-lo fragment reads are retired by `lgkmcnt(8 | 15)` in front of the first
    barrier of phase 1.  REBAL instances (two k-tiles per iteration, B register sets swapping roles): no such wait - B-lo is read
    three phases before its slot is restaged.  The MFMAs are inline assembly with tied accumulators: D and C are the same
    registers in every accumulating MFMA."""
    no_scratch(isa, "gemm8p.hip")
    bodies = kernel_bodies(isa, "gemm8p_kernel")
    _need(len(bodies) >= 36, f"gemm8p.hip: {len(bodies)} kernel instances found")
    for sym, body in bodies.items():
        m = re.search(r"gemm8p_kernelILb[01]ELb[01]ELi\d+ELb([01])EEEv", sym)
        _need(m, f"{sym}: template arguments not recognised")
        rebal = m.group(1) == "1"
        n = 2 if rebal else 1
        ring = [seg for _, seg in loops(body) if seg.count("v_mfma_f32_16x16x32_bf16") == 64 * n]
        _need(ring, f"{sym}: steady-state k-loop ({64 * n} MFMAs) not found")
        seg = min(ring, key=len)
        dma = len(re.findall(r"\bglobal_load_lds_dwordx4\b", seg))
        _need(dma == 8 * n, f"{sym}: {dma} LDS-DMA instructions per loop iteration, expected {8 * n}")
        _need(seg.count("s_barrier") == 8 * n, f"{sym}: {seg.count('s_barrier')} barriers per loop iteration, expected {8 * n}")
        waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", seg)
        _need(waits == ["6"] * n, f"{sym}: vector-memory waits in the k-loop are {waits}, expected {n} x vmcnt(6)")
        for m in re.finditer(r"s_waitcnt vmcnt\(6\)", seg):
            _need(re.match(r"\s*s_barrier", seg[m.end():]), f"{sym}: a counted wait is not directly in front of a barrier")
        _need("scratch_" not in seg and not re.search(r"\b(buffer_|flat_|global_load_dword|global_store)", seg),
              f"{sym}: stray memory instruction in the k-loop")
        lg = re.findall(r"s_waitcnt lgkmcnt\((8|15)\)\s*\n\s*s_barrier", seg)
        _need(len(lg) == (0 if rebal else 1), f"{sym}: the B-lo retire wait (lgkmcnt(8|15) in front of phase 1's barrier) appears {len(lg)} times")
        for m in re.finditer(r"v_mfma_f32_16x16x32_bf16 (v\[\d+:\d+\]), v\[\d+:\d+\], v\[\d+:\d+\], (\S+)", seg):
            _need(m.group(1) == m.group(2), f"{sym}: an accumulating MFMA of the k-loop has D != C ({m.group(0)})")
        # whole kernel: prologue 14 pieces + the inlined k-tile bodies (first + steady, x 2 for REBAL) of 8
        total = len(re.findall(r"\bglobal_load_lds_dwordx4\b", body))
        _need(total == 14 + 16 * n, f"{sym}: {total} LDS-DMA instructions in the kernel, expected {14 + 16 * n}")
        _need(body.count("s_waitcnt vmcnt(0)") >= 1, f"{sym}: the final drain of the ring is missing")
        _need(re.search(r"s_nop 15\s*\n\s*s_nop 7", body), f"{sym}: the epilogue's wait states behind the asm MFMAs are missing")


# ------------------------------------------------------------------------------------------------ attention_bwd.hip
def check_attention_bwd(isa: str):
    no_scratch(isa, "attention_bwd.hip")
    for name, mfma_per_block in (("attn_bwd_dkv3_kernel", 64), ("attn_bwd_dq3_kernel", 48)):
        for sym, body in kernel_bodies(isa, name).items():
            blocks = [b for b in re.split(r"\n(?=\.LBB\d+_\d+:)", body) if b.count("v_mfma_f32_32x32x16_bf16") >= mfma_per_block]
            _need(blocks, f"{sym}: main loop not found")
            for b in blocks:
                _need("v_accvgpr_write" not in b and "v_accvgpr_read" not in b, f"{sym}: accumulator copies in the main loop "
                                                                                 "(an AGPR operand re-materialised in front of an asm MFMA is read stale)")
                _need("scratch_" not in b, f"{sym}: scratch access in the main loop")
                _need(re.search(r"v_mfma_f32_32x32x16_bf16 v\[\d+:\d+\], v\[\d+:\d+\], a\[\d+:\d+\]", b), f"{sym}: VGPR-result MFMA form missing")
                _need(re.search(r"v_mfma_f32_32x32x16_bf16 a\[\d+:\d+\], v\[\d+:\d+\], v\[\d+:\d+\], a\[\d+:\d+\]", b), f"{sym}: AGPR-accumulating MFMA form missing")


# ------------------------------------------------------------------------------------------------ attention_bwd_fused.hip
def check_attention_bwd_fused(isa: str):
    """The six-tile ring of attention_bwd_fused.hip.  Per tile and wave: five LDS-DMA instructions, then either two 16-byte stores
    (the instances that store their dQ partial: counted wait vmcnt(9) = 2 + 5 + 2) or eight packed-bf16 atomic adds (the ADD
    instances, which add theirs to a slab another launch stored: vmcnt(21) = 8 + 5 + 8); nothing else touches vector memory, no
    accumulator copies, no scratch."""
    no_scratch(isa, "attention_bwd_fused.hip")
    for variant in ("ILb1ELb0E", "ILb0ELb0E", "ILb1ELb1E", "ILb0ELb1E"):
        add = variant.endswith("ELb1E")
        bodies = kernel_bodies(isa, "attn_bwd_fused_kernel" + variant)
        _need(len(bodies) == 1, f"attn_bwd_fused_kernel{variant}: {len(bodies)} instances")
        sym, body = next(iter(bodies.items()))
        ring = [seg for _, seg in loops(body) if seg.count("v_mfma_f32_32x32x16_bf16") == 480]
        _need(ring, f"{sym}: unrolled six-tile ring not found")
        ring = ring[-1]
        _need("v_accvgpr_write" not in ring and "v_accvgpr_read" not in ring and "scratch_" not in ring, f"{sym}: accumulator copies / scratch in the ring")
        _need(ring.count("global_load_lds_dwordx4") == 24 and ring.count("global_load_lds_dword ") == 6, f"{sym}: LDS-DMA count per ring changed")
        stores, atomics = len(re.findall(r"\bglobal_store_dwordx4\b", ring)), len(re.findall(r"\bglobal_atomic_pk_add_bf16\b", ring))
        _need((stores, atomics) == ((0, 48) if add else (12, 0)) and len(re.findall(r"\bglobal_(load|store|atomic)_", ring)) == 30 + stores + atomics,
              f"{sym}: vector-memory instruction count per ring changed ({stores} stores, {atomics} atomics)")
        _need(not re.search(r"\bbuffer_|\bflat_", ring), f"{sym}: buffer / flat instruction in the ring")
        _need(ring.count("s_barrier") == 6, f"{sym}: barriers per ring")
        n = "21" if add else "9"
        _need(re.findall(r"s_waitcnt vmcnt\((\d+)\)", ring) == [n] * 6 and "vmcnt(0)" not in ring, f"{sym}: the ring's waits are not six vmcnt({n})")
        for m in re.finditer(r"s_barrier", ring):
            _need(f"vmcnt({n})" in ring[max(0, m.start() - 400):m.start()], f"{sym}: a barrier without its counted wait")
        _need(re.search(r"v_mfma_f32_32x32x16_bf16 v\[\d+:\d+\], a\[\d+:\d+\], v\[\d+:\d+\], v\[\d+:\d+\]", ring), f"{sym}: dQ MFMA form")
        _need(ring.count("v_exp_f32") == 384 and len(re.findall(r"v_exp_f32_e64 v\d+, v\d+ clamp", ring)) == 384, f"{sym}: exponentials / clamp")


# ------------------------------------------------------------------------------------------------ attention_fwd.hip
def hot_path(lines: list[str], lo: int, hi: int):
    """(index, line) of lines lo..hi as a wave executes them when every forward conditional branch that jumps over a block is TAKEN:
    the blocks such branches skip are the cold blocks of a hand-placed stream (masked tile, reference move)."""
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    i = lo
    while i <= hi:
        l = lines[i]
        yield i, l
        m = re.search(r"s_cbranch_\S+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and i < labels[m.group(1)] <= hi:
            i = labels[m.group(1)]
            continue
        if m and labels.get(m.group(1)) == lo:  # (a rotated loop: the last cold block sits between this branch and the latch)
            return
        i += 1


def check_attention_fwd(isa: str):
    """The pipelined global forward (attention_fwd.hip).  Its loop body is four tiles (= the ring); per tile and wave exactly four
    LDS-DMA pieces (+ one 64-byte mask piece in the MASK instance) behind ONE counted wait, vmcnt(4) (5), directly in front of the one
    barrier; nothing else touches vector memory and nothing is spilled - a compiler-issued scratch reload or global load would be
    counted by vmcnt and retire in order, i.e. drain the ring.  The score product must be the VGPR-result form with both operands in
    AGPRs, the output product the AGPR-accumulating form.  On the HOT path (cold blocks skipped) per 16 MFMAs: 32 exponentials, at
    most 110 vector instructions in all, no v_mov, and accumulator copies only as the loop header's one block (r04 verdict item 1e)."""
    no_scratch(isa, "attention_fwd.hip")
    for mask in (False, True):
        bodies = kernel_bodies(isa, "attn_fwd_g_kernelILi4ELb%dE" % int(mask))
        _need(len(bodies) == 1, f"attn_fwd_g_kernel<4, {mask}>: {len(bodies)} instances")
        sym, body = next(iter(bodies.items()))
        rings = [seg for _, seg in loops(body) if seg.count("v_mfma_f32_32x32x16_bf16") == 256]
        _need(len(rings) == 2, f"{sym}: {len(rings)} four-tile loops (256 MFMAs) found, expected the all-visible sweep and the masked sweep")
        nd = "5" if mask else "4"
        lines = body.split("\n")
        for which, ring in enumerate(rings):  # 0: every key visible (no masking code), 1: one branch per block
            what = f"{sym} loop {which}"
            _need(ring.count("global_load_lds_dwordx4") == 16 and ring.count("global_load_lds_ubyte") == (4 if mask else 0), f"{what}: LDS-DMA count per loop trip changed")
            _need(len(re.findall(r"\bglobal_(load|store|atomic)_", ring)) == 16 + (4 if mask else 0) and not re.search(r"\bbuffer_|\bflat_|scratch_", ring),
                  f"{what}: stray vector-memory instruction in the loop")
            _need(ring.count("s_barrier") == 4, f"{what}: {ring.count('s_barrier')} barriers per loop trip, expected 4")
            _need(re.findall(r"s_waitcnt vmcnt\((\d+)\)", ring) == [nd] * 4, f"{what}: the loop's vector-memory waits are not four vmcnt({nd})")
            for m in re.finditer(r"s_waitcnt vmcnt\(\d+\)", ring):
                _need(re.match(r"\s*s_barrier", ring[m.end():]), f"{what}: a counted wait is not directly in front of its barrier")
            _need(len(re.findall(r"v_mfma_f32_32x32x16_bf16 v\[\d+:\d+\], a\[\d+:\d+\], a\[\d+:\d+\], v\[\d+:\d+\]", ring)) == 128, f"{what}: score MFMA form (VGPR result, AGPR operands)")
            _need(len(re.findall(r"v_mfma_f32_32x32x16_bf16 (a\[\d+:\d+\]), v\[\d+:\d+\], v\[\d+:\d+\], \1", ring)) == 128, f"{what}: output MFMA form (AGPR accumulator tied to itself)")
            # the hot path of the loop trip
            first = body[:body.index(ring)].count("\n")
            hot = [l.split()[0] for _, l in hot_path(lines, first, first + ring.count("\n")) if l.startswith("\t") and l.split()]
            n_mfma = sum(o.startswith("v_mfma") for o in hot)
            _need(n_mfma == 256, f"{what}: {n_mfma} MFMAs on the hot path of a loop trip")
            _need(sum(o.startswith("v_exp_f32") for o in hot) == 512, f"{what}: exponentials on the hot path")
            vec = sum(o.startswith("v_") and not o.startswith("v_mfma") for o in hot)
            _need(vec <= 110 * 16, f"{what}: {vec / 16:.1f} vector instructions per 16 MFMAs on the hot path (bound: 110)")
            acc = sum("accvgpr" in o for o in hot)
            _need(acc <= 32, f"{what}: {acc} accumulator copies on the hot path of a loop trip (the loop header's block is 32)")
            _need(sum(o.startswith("v_mov_b32") for o in hot) <= 8, f"{what}: v_mov on the hot path")
            nbr = sum(o.startswith("s_cbranch") for o in hot)
            _need(nbr <= (17 if which == 0 else 49), f"{what}: {nbr} conditional branches on the hot path (all-visible sweep: one per period + the loop's)")
        _need(body.count("s_waitcnt vmcnt(0)") >= 2, f"{sym}: the prologue wait / the final drain of the ring is missing")


CHECKS = {"gemm8p.hip": check_gemm8p, "attention_fwd.hip": check_attention_fwd, "attention_bwd.hip": check_attention_bwd,
          "attention_bwd_fused.hip": check_attention_bwd_fused}


def check_file(src: str, flags: list[str]) -> None:
    CHECKS[src](compile_isa(src, flags))
