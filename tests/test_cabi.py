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
    assert _lib.query("cm3p_attn_bwd_fused_workspace_bytes", 32, 4096, 12) == 32 * 12 * (76 * 512 + 16 * 4160 * 128)


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
