"""The C ABI without torch: examples/c_abi_demo.cpp (hipMalloc buffers, host-loop check) built against
libcmtfpls.so by csrc/build.sh, run here on the GPU box."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_demo_runs():
    exe = os.path.join(ROOT, "examples", "c_abi_demo")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C ABI demo OK" in out.stdout
    # the RCCL wrapper (cmtfpls_allreduce_sum_f64) on a one-rank communicator created without torch
    assert "through cmtfpls_allreduce_sum_f64 OK" in out.stdout, out.stdout


def test_python_quickstart_runs():
    """examples/quickstart.py end to end (fit, transform / predict, one-launch LOO, NaN imputation, coupled blocks, xcov)."""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "quickstart.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "Q2Y (leave-one-out)" in p.stdout and "ctPLS R2Y" in p.stdout
