"""Compile-time checks of the kernels whose correctness depends on the emitted ISA (no GPU needed): the counted-vmcnt LDS-DMA rings
of gemm8p.hip and attention_bwd_fused.hip and the inline-asm MFMAs of attention_bwd.hip / attention_bwd_fused.hip.  The checks live in
cm3p_amd/isa_check.py because cm3p_amd/build.py runs the same ones on every rebuild of these objects (a failed check fails the
build); here they run on every test run, plus a negative control that the checker really looks at the ring."""
import os

import pytest

from cm3p_amd import build, isa_check

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


@pytest.fixture(scope="module")
def isa():
    cache = {}

    def get(src):
        if src not in cache:
            cache[src] = isa_check.compile_isa(src, build.FLAGS + build.EXTRA_FLAGS.get(src, []))
        return cache[src]

    return get


@pytest.mark.parametrize("src", sorted(isa_check.CHECKS))
def test_isa_pattern(isa, src):
    isa_check.CHECKS[src](isa(src))


def test_gemm8p_checker_rejects_a_drained_ring(isa):
    """Negative control: the same ISA with the counted wait replaced by a drain, or with one more DMA in the loop, must fail."""
    text = isa("gemm8p.hip")
    with pytest.raises(isa_check.IsaCheckError):
        isa_check.check_gemm8p(text.replace("s_waitcnt vmcnt(6)", "s_waitcnt vmcnt(0)"))
    with pytest.raises(isa_check.IsaCheckError):
        isa_check.check_gemm8p(text.replace("s_waitcnt lgkmcnt(8)", "s_waitcnt lgkmcnt(12)"))


def test_fused_checker_rejects_a_spill(isa):
    text = isa("attention_bwd_fused.hip")
    with pytest.raises(isa_check.IsaCheckError):
        isa_check.check_attention_bwd_fused(text.replace(".vgpr_spill_count: 0", ".vgpr_spill_count: 3", 1))
