"""CPU-only: libcmtfpls.so builds/loads without a GPU and exports exactly what include/cmtfpls.h
declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cmtfpls.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cmtfpls_\w+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from cmtf_pls_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.LIB_PATH


def test_header_declares_the_hot_path():
    names = declared_symbols()
    for base in ["mode0_contract", "score", "deflate", "score_deflate", "colstats", "center"]:
        assert f"cmtfpls_{base}_f32" in names and f"cmtfpls_{base}_f64" in names
    assert "cmtfpls_rank1_f64" in names and "cmtfpls_gram_tn_f64" in names


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/cmtfpls.h but not exported"


def test_binding_table_matches_header(lib_path):
    from cmtf_pls_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.cmtfpls_abi_version() == 1
    assert lib.cmtfpls_sweep_partials() > 0
    assert lib.cmtfpls_mode0_contract_workspace_bytes(65536, 16384) > 0      # pure host arithmetic
    assert lib.cmtfpls_rank1_workspace_bytes(128, 128) > 0


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/cmtfpls.h must compile as C99 (and as C++) with no torch / HIP types."""
    import shutil
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "cmtfpls.h"\nint main(void) { return cmtfpls_abi_version == 0; }\n')
    inc = os.path.join(ROOT, "include")
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not installed")
        extra = ["-x", "c++"] if cc == "g++" else []
        p = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", "-I", inc] + extra + [str(src)], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
    text = open(os.path.join(inc, "cmtfpls.h")).read()
    assert "torch" not in text.lower().replace("torch tensor's data_ptr", "") and "hipStream_t stream" not in text
