"""The C++ deme::DEMSolver shell (dem-engine_amd/host/DEMSolver.h): a demo-style program written against the
reference's scripting API compiles with plain g++ and links only the C-ABI library."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dem-engine_amd", "host")


def test_demo_builds_and_fails_loudly_without_a_gpu():
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    exe = os.path.join(HOST, "demo_settle")
    assert os.path.exists(exe)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([exe, "4", "10"], capture_output=True, text=True)
    assert out.returncode != 0 and "HIP device" in (out.stderr + out.stdout)


@pytest.mark.gpu
def test_demo_runs_and_settles():
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(HOST, "demo_settle"), "10", "3000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "DEMO_OK clumps=1000" in out.stdout
    rows = [l for l in out.stdout.splitlines() if l.startswith("t=")]
    z = [float(l.split("zmean=")[1]) for l in rows]
    c = [int(l.split("contacts=")[1].split()[0]) for l in rows]
    assert z[-1] < z[0] and c[-1] > 100  # the bed drops and contacts form
