import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the HIP library and the C demo normally arrive prebuilt (__graft_entry__.build()); if they are
    # missing, build them once here (hipcc cross-compiles for gfx950 with or without a GPU)
    lib = os.path.join(ROOT, "cmtf_pls_amd", "lib", "libcmtfpls.so")
    demo = os.path.join(ROOT, "examples", "c_abi_demo")
    if not (os.path.exists(lib) and os.path.exists(demo)):
        import subprocess
        subprocess.run(["bash", os.path.join(ROOT, "cmtf_pls_amd", "csrc", "build.sh")], check=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
