"""bench.py's launch path on a CPU box: `python bench.py --gpus N` without a launcher must come back as N ranks of one node
(torch.distributed.run, rendezvous on 127.0.0.1), and the strong-scaling form --clumps-total must split the job's clumps over
them.  --launch-check stops after the rendezvous and the rank count (no GPU work); the measurement itself, with `rccl_ranks`
and `halo_loop` in its JSON line and a non-zero exit when the communicator does not span N ranks, needs N GPUs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None)


def test_gpus_n_without_a_launcher_spawns_n_ranks():
    out, js = _run(["--gpus", "2", "--launch-check"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert js == {"launch_check": True, "n_gpus": 2, "ranks_seen": 2, "clumps_per_gpu": 1_000_000, "scaling": "weak"}
    assert "re-executing as 2 ranks" in out.stderr


def test_strong_scaling_splits_the_total():
    out, js = _run(["--gpus", "2", "--launch-check", "--clumps-total", "10000000"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert js["ranks_seen"] == 2 and js["clumps_per_gpu"] == 5_000_000 and js["scaling"] == "strong"


def test_a_mismatched_launch_is_refused():
    # started as ONE rank by somebody else's launcher while asking for two: not a two-GPU measurement
    out, js = _run(["--gpus", "2", "--launch-check"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and js is None
    assert "was started as 1 rank" in (out.stderr + out.stdout)
