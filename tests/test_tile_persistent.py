"""The persistent form of the owner-tile force pass (csrc/deme_tile_p.h, DEME_TILE_PERSIST=1) against the one-workgroup-per-tile
form: same tiles, same arithmetic, same summation order -- every owner-state array, the contact list and every contact wildcard
must come out BIT-identical.  The switch is read once per process, so each run is a child process (tools/persist_compare.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("clumps,steps", [(30000, 80), (160000, 60)])
def test_persistent_tile_kernel_is_bit_identical(clumps, steps):
    """a lattice that starts overlapping (thousands of contacts from the first step, clumps flying apart: detections every 10 steps,
    tiles of every size, partial rounds) stepped by both kernels: 235 tiles (one per workgroup, one counter) and 1 250 tiles (more than
    the 1 024 workgroups the chip holds: the loop over tiles, 32 counters); the 10^6-clump run is in profiles/r06/persistent_first_attempt.txt"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "persist_compare.py"), str(clumps), str(steps)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "bit-identical" in out.stdout and "k_tile_forces<0, false>" in out.stdout, out.stdout[-1000:]
