"""Compile-time check of the hand-scheduled attention backward kernels (csrc/attention_bwd.hip), no GPU needed.

Those kernels issue their score MFMAs as inline asm (VGPR results, AGPR-resident stationary operands).  hipcc pads no hazards for
an asm statement, so two properties of the generated ISA are part of their correctness and are pinned here:
  * no v_accvgpr_write / v_accvgpr_read inside the main loops: an AGPR operand that the compiler re-materialises right in front of
    an asm MFMA is read stale (this happened once: dO also fed VALU code, NaN gradients);
  * no scratch memory and no register spills (a spilled staging register turns every prefetch into a synchronous round trip).
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cm3p_amd", "csrc", "attention_bwd.hip")
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
