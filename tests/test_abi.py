"""CPU tier: the C-ABI library loads and exports every symbol include/hvs.h declares (no compute)."""

import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "hvs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hvs_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    import hirag_prod_amd as hvs
    from hirag_prod_amd import _lib

    lib = hvs.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"libhvs.so does not export {name}"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    assert lib.hvs_version() >= 100


def test_constants_match_header():
    from hirag_prod_amd import _lib

    text = open(os.path.join(ROOT, "include", "hvs.h")).read()
    assert int(re.search(r"#define HVS_MAX_K (\d+)", text).group(1)) == _lib.HVS_MAX_K
    assert int(re.search(r"#define HVS_MAX_FOLD (\d+)", text).group(1)) == _lib.HVS_MAX_FOLD
    assert int(re.search(r"#define HVS_MAX_DIST_FOLD (\d+)", text).group(1)) == _lib.HVS_MAX_DIST_FOLD
    algos = dict(re.findall(r"(HVS_ALGO_\w+) = (\d+)", text))
    assert algos and all(getattr(_lib, name) == int(value) for name, value in algos.items())
    assert sorted(_lib.ALGO_CODE.values()) == sorted(int(v) for v in algos.values())


def test_errors_cross_the_boundary_as_messages():
    import ctypes

    import hirag_prod_amd as hvs

    lib = hvs.load_library()
    h = ctypes.c_void_p()
    rc = lib.hvs_index_create(ctypes.byref(h), 0, 0, 0, None)  # dim 0 is rejected before any GPU call
    assert rc != 0 and b"dim" in lib.hvs_last_error()
    assert lib.hvs_index_destroy(None) == 0
    with pytest.raises(hvs.HvsError):
        hvs.HvsIndex(8, "f32", device=10_000)


def test_header_is_plain_c_and_cxx():
    """The boundary is a C ABI: the header must compile as C99 and as C++ without any other include."""
    import shutil
    import subprocess

    header = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "hvs.h")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", header])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", header])
