import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "small_fit: keep the one-launch small-fit path enabled (the product default)")
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


@pytest.fixture(autouse=True)
def _regular_engine_unless_asked(request):
    """Harness default of `EngineOptions`: the one-launch small-fit path (NipalsEngine._fit_small) would swallow every small
    float64 fit of the suites written to exercise the multi-launch kernels (it ignores `algorithm` and `graphs`), so engines
    constructed WITHOUT explicit options keep the regular engine here.  Tests marked `small_fit` -- the port of the
    reference's own suite, the configs[0] golden tests in both modes (`small_fit_mode`) -- see the product default; tests of
    one switch construct their own `EngineOptions(...)` and pass it to the estimator."""
    from cmtf_pls_amd.engine import EngineOptions, set_default_options
    if request.node.get_closest_marker("small_fit") is not None:
        yield                                # the process default as it stands (the product's, unless a module fixture runs both modes)
        return
    old = set_default_options(EngineOptions(small_fit=False))
    yield
    set_default_options(old)


@pytest.fixture(params=["one_launch", "regular"])
def small_fit_mode(request):
    """Estimator-level tests that must hold on BOTH paths a small float64 fit can take: the product default (the whole
    fit in one launch) and the regular multi-launch engine.  Yields the EngineOptions to pass as `options=`."""
    from cmtf_pls_amd.engine import EngineOptions
    return EngineOptions(small_fit=(request.param == "one_launch"))
