"""The C-ABI library loads without a GPU and exports every function include/cm3p_hip.h declares; the ctypes binding
table covers exactly the same set (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cm3p_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(cm3p_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from cm3p_amd import _lib, build

    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.LIB_PATH


def test_header_declares_functions():
    names = _declared()
    assert len(names) >= 30 and "cm3p_gemm_bf16" in names and "cm3p_attn_bwd" in names


def test_every_declared_symbol_is_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header(lib_path):
    from cm3p_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.cm3p_abi_version() == _lib.ABI_VERSION
    # argument counts in the binding equal the parameter counts in the header
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, argtypes in _lib.SIGNATURES.items():
        m = re.search(r"\b(?:int|int64_t)\s+" + name + r"\s*\((.*?)\)\s*;", text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(argtypes), (name, n, len(argtypes))


def test_host_only_queries(lib_path):
    from cm3p_amd import _lib

    assert _lib.query("cm3p_pool_chunks", 4096) == 32
    assert 1 <= _lib.query("cm3p_layernorm_bwd_blocks", 131072) <= 2048
    # 64-bit size query: the C2 workspace of the fused attention backward is past 2^31 bytes
    assert _lib.query("cm3p_attn_bwd_fused_workspace_bytes", 32, 4096, 12) == 32 * 12 * (76 * 512 + 8 * 4160 * 128)  # (two 256-key blocks share a slab since ABI 13)


def test_cpu_tensors_are_refused(lib_path):
    import torch

    from cm3p_amd import _lib, kernels

    x = torch.zeros(4, 64)
    with pytest.raises(_lib.Cm3pHipError):
        kernels.layernorm_fwd(x, torch.ones(64), 1e-5, True, False)


def test_pointer_arguments_on_different_gpus_are_refused():
    """call() launches on the device its tensor arguments live on and refuses a mix (a device-0 launch over device-1 pointers
    is a memory fault or a silent peer access); checked before anything touches a GPU, so it runs here."""
    import pytest

    from cm3p_amd import _lib

    a, b = _lib._DevPtr(4096), _lib._DevPtr(8192)
    a.dev, b.dev = 0, 1
    with pytest.raises(_lib.Cm3pHipError, match="different GPUs"):
        _lib._launch("cm3p_cast_f32_bf16", (a, b, 16, _lib.stream()))


def test_fused_attention_backward_rejects_bad_arguments_without_a_gpu(lib_path):
    """Argument validation happens before any HIP call: NULL tensors, a window request, an undersized workspace or an unknown
    stage mask return CM3P_ERR_INVALID (-1) instead of launching."""
    from cm3p_amd import _lib

    lib = _lib.load()
    need = lib.cm3p_attn_bwd_fused_workspace_bytes(2, 512, 4)
    assert need == 2 * 4 * ((8 + 12) * 512 + 1 * (512 + 64) * 128)  # (S = 512: two key blocks, one shared slab)
    fake = 4096  # an aligned, never dereferenced address: every call below must fail validation first
    args = dict(qkv=fake, out=fake, dout=fake, lse=fake, dqkv=fake, key_mask=None, cu=None, B=2, S=512, total=0, nh=4, scale=0.125, cos=None, sin=None,
                pbs=0, stages=7, pre=1, ws=fake, ws_bytes=need, stream=None)

    def call(**kw):
        a = dict(args, **kw)
        return lib.cm3p_attn_bwd_fused(a["qkv"], a["out"], a["dout"], a["lse"], a["dqkv"], a["key_mask"], a["cu"], a["B"], a["S"], a["total"], a["nh"],
                                       a["scale"], a["cos"], a["sin"], a["pbs"], a["stages"], a["pre"], a["ws"], a["ws_bytes"], a["stream"])

    assert call(qkv=None) == -1
    assert call(ws=None) == -1
    assert call(ws_bytes=need - 1) == -1
    assert call(stages=0) == -1 and call(stages=256) == -1
    assert call(cos=fake, sin=None) == -1
    assert call(cu=fake, total=0) == -1  # packed rows need their total
    assert call(qkv=fake + 2) == -1  # 16-byte alignment


def test_shipped_library_carries_no_timing_ablations(lib_path, tmp_path):
    """The hand-scheduled kernels have compile-time timing probes (CM3P_ABL / CM3P_FABL / CM3P_BABL / CM3P_G256_ABL / CM3P_G8P_ABL)
    whose results are wrong by construction.  The in-tree library reports a zero mask, and the binding refuses a library that
    does not (checked on an object built with one probe set, linked with the shipped objects; compile only, no GPU)."""
    import subprocess

    from cm3p_amd import _lib, build

    lib = ctypes.CDLL(lib_path)
    assert lib.cm3p_build_ablation_flags() == 0
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    obj = tmp_path / "gemm8p_abl.o"
    subprocess.run([hipcc, *build.FLAGS, "-DCM3P_G8P_ABL=1", "-c", os.path.join(build.CSRC, "gemm8p.hip"), "-o", str(obj)], check=True, capture_output=True)
    objs = [os.path.join(build.CSRC, s.replace(".hip", ".o")) for s in build.SOURCES if s != "gemm8p.hip"]
    bad = tmp_path / "libablated.so"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(bad), *objs, str(obj)], check=True, capture_output=True)
    assert ctypes.CDLL(str(bad)).cm3p_build_ablation_flags() == 1 << 4
    code = ("import os, sys; os.environ['CM3P_HIP_LIB'] = sys.argv[1]\n"
            "from cm3p_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.Cm3pHipError as e:\n    print('REFUSED' if 'ablation' in str(e) else e)\n")
    out = subprocess.run([os.sys.executable, "-c", code, str(bad)], capture_output=True, text=True, cwd=ROOT)
    assert "REFUSED" in out.stdout, out.stdout + out.stderr
