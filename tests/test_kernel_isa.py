"""Compile-time check of the hand-scheduled attention backward kernels (csrc/attention_bwd.hip, attention_bwd_fused.hip), no GPU needed.

Those kernels issue their score MFMAs as inline asm (VGPR results, AGPR-resident stationary operands).  hipcc pads no hazards for
an asm statement, so two properties of the generated ISA are part of their correctness and are pinned here:
  * no v_accvgpr_write / v_accvgpr_read inside the main loops: an AGPR operand that the compiler re-materialises right in front of
    an asm MFMA is read stale (this happened once: dO also fed VALU code, NaN gradients);
  * no scratch memory and no register spills (a spilled staging register turns every prefetch into a synchronous round trip);
  * (fused kernel) the vector-memory operations of one tile are exactly five LDS-DMA loads and two stores and every workgroup
    barrier sits right behind `s_waitcnt vmcnt(9)`: the counted wait is only correct for that issue pattern, and a
    compiler-inserted spill or reload would silently change it.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cm3p_amd", "csrc", "attention_bwd.hip")
SRC_FUSED = os.path.join(ROOT, "cm3p_amd", "csrc", "attention_bwd_fused.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    import sys

    sys.path.insert(0, ROOT)
    from cm3p_amd.build import EXTRA_FLAGS, FLAGS

    out = tmp_path_factory.mktemp("isa") / "attention_bwd.s"
    cmd = [HIPCC, *[f for f in FLAGS if f not in ("-Wall",)], *EXTRA_FLAGS.get("attention_bwd.hip", []), "-S", "--cuda-device-only", "-o", str(out), SRC]
    subprocess.run(cmd, check=True, capture_output=True)
    return out.read_text()


@pytest.fixture(scope="module")
def isa_fused(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    import sys

    sys.path.insert(0, ROOT)
    from cm3p_amd.build import EXTRA_FLAGS, FLAGS

    out = tmp_path_factory.mktemp("isa") / "attention_bwd_fused.s"
    cmd = [HIPCC, *[f for f in FLAGS if f not in ("-Wall",)], *EXTRA_FLAGS.get("attention_bwd_fused.hip", []), "-S", "--cuda-device-only", "-o", str(out),
           SRC_FUSED]
    subprocess.run(cmd, check=True, capture_output=True)
    return out.read_text()


def _kernel_body(isa: str, name: str) -> str:
    m = re.search(r"^_ZN\S*" + name + r"\S*:", isa, re.M)
    assert m, name
    return isa[m.start():isa.index(".Lfunc_end", m.start())]


@pytest.mark.parametrize("name,mfma_per_block", [("attn_bwd_dkv3_kernel", 64), ("attn_bwd_dq3_kernel", 48)])
def test_hand_scheduled_loops_have_no_agpr_copies(isa, name, mfma_per_block):
    body = _kernel_body(isa, name)
    loops = [b for b in re.split(r"\n(?=\.LBB\d+_\d+:)", body) if b.count("v_mfma_f32_32x32x16_bf16") >= mfma_per_block]
    assert loops, "main loop not found"
    for b in loops:
        assert "v_accvgpr_write" not in b and "v_accvgpr_read" not in b
        assert "scratch_" not in b
        # the asm MFMAs really are the VGPR-result / AGPR-operand form, the accumulating ones the AGPR form
        assert re.search(r"v_mfma_f32_32x32x16_bf16 v\[\d+:\d+\], v\[\d+:\d+\], a\[\d+:\d+\]", b)
        assert re.search(r"v_mfma_f32_32x32x16_bf16 a\[\d+:\d+\], v\[\d+:\d+\], v\[\d+:\d+\], a\[\d+:\d+\]", b)


def test_no_scratch_in_any_attention_backward_kernel(isa):
    for m in re.finditer(r"\.private_segment_fixed_size:\s*(\d+)", isa):
        assert int(m.group(1)) == 0
    for m in re.finditer(r"\.vgpr_spill_count:\s*(\d+)", isa):
        assert int(m.group(1)) == 0


@pytest.mark.parametrize("variant", ["ILb1E", "ILb0E"])  # q_prescaled and plain-q instantiations
def test_fused_backward_ring(isa_fused, variant):
    body = _kernel_body(isa_fused, "attn_bwd_fused_kernel" + variant)
    # the unrolled ring: the back edge's target .. the back edge (6 tiles x 80 MFMAs)
    labels = {m.group(1): m.start() for m in re.finditer(r"^(\.LBB\d+_\d+):", body, re.M)}
    ring = None
    for m in re.finditer(r"s_cbranch\S*\s+(\.LBB\d+_\d+)", body):
        if m.group(1) in labels and labels[m.group(1)] < m.start():
            seg = body[labels[m.group(1)]:m.start()]
            if seg.count("v_mfma_f32_32x32x16_bf16") == 480:
                ring = seg
    assert ring, "unrolled six-tile ring not found"
    assert "v_accvgpr_write" not in ring and "v_accvgpr_read" not in ring and "scratch_" not in ring
    assert ring.count("global_load_lds_dwordx4") == 24 and ring.count("global_load_lds_dword ") == 6
    assert len(re.findall(r"\bglobal_store_dwordx4\b", ring)) == 12 and len(re.findall(r"\bglobal_(load|store)_", ring)) == 42
    assert not re.search(r"\bbuffer_|\bflat_", ring)
    # one barrier per tile, each right behind the counted wait; no other vmcnt wait in the ring
    assert ring.count("s_barrier") == 6
    assert re.findall(r"s_waitcnt vmcnt\((\d+)\)", ring) == ["9"] * 6 and "vmcnt(0)" not in ring
    for m in re.finditer(r"s_barrier", ring):
        assert "vmcnt(9)" in ring[max(0, m.start() - 400):m.start()]
    # dQ MFMAs: VGPR result, AGPR-resident K^T operand
    assert re.search(r"v_mfma_f32_32x32x16_bf16 v\[\d+:\d+\], a\[\d+:\d+\], v\[\d+:\d+\], v\[\d+:\d+\]", ring)
    # exponentials carry the clamp
    assert ring.count("v_exp_f32") == 384 and len(re.findall(r"v_exp_f32_e64 v\d+, v\d+ clamp", ring)) == 384


def test_no_scratch_in_fused_kernel(isa_fused):
    for m in re.finditer(r"\.private_segment_fixed_size:\s*(\d+)", isa_fused):
        assert int(m.group(1)) == 0
    for m in re.finditer(r"\.vgpr_spill_count:\s*(\d+)", isa_fused):
        assert int(m.group(1)) == 0
