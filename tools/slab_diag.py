"""diagnostic: demo_bed in slabs under several knobs against the oracle (tests/test_host_shell.py: test_shell_runs_the_bed_in_slabs)"""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
from tests.test_host_shell import _bed_inputs, _bed_scene, HOST
pkg, orc = entry.load_package(), entry.load_oracle()
orc.build(); orc.set_num_threads(8)
n, steps = 1100, int(sys.argv[1]) if len(sys.argv) > 1 else 7000
xyz, q, kind = _bed_inputs(n)
tmp = tempfile.mkdtemp()
np.concatenate([xyz, q, kind[:, None]], 1).astype(np.float32).tofile(os.path.join(tmp, "clumps.f32"))
b = _bed_scene(pkg, xyz, q, kind, None)
p, sc = b.Initialize()
sim = orc.make_sim(pkg, p, sc)
c = np.zeros((15, 4), np.float32); c[0] = (0.01, 0, 0, 0)
sim.set_prescription(10, has=0b111, flags=0b111, coef=c)
sim.step(steps)
st = sim.download_state()
X = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:n] + np.array([p.LBFX, p.LBFY, p.LBFZ])
import json
CASES = json.loads(os.environ.get("SLAB_CASES", '[[2, "", "1000"]]'))
for slabs, halo, mig in CASES:
    env = dict(os.environ, DEME_ARITH="exact", DEME_SLABS_PER_DEVICE=str(slabs), DEME_SLAB_MIGRATE_EVERY=mig)
    if halo:
        env["DEME_SLAB_HALO"] = halo
    out = subprocess.run([os.path.join(HOST, "demo_bed"), os.path.join(tmp, "clumps.f32"), str(n), str(steps), tmp], capture_output=True, text=True, env=env)
    if out.returncode:
        print(slabs, halo, mig, "FAILED", out.stderr[-300:]); continue
    rows = np.genfromtxt(os.path.join(tmp, "clumps.csv"), delimiter=",", names=True)
    got = np.stack([rows["X"], rows["Y"], rows["Z"]], 1)
    d = np.abs(got - X).max(1)
    off = np.nonzero(d > 1e-7)[0]
    print(f"slabs {slabs} halo {halo or 'default'} migrate {mig}: |dx| {d.max():.3e} at clump {d.argmax()} x={X[d.argmax()]}, {len(off)} clumps off {off[:12].tolist()} x of those {np.round(X[off[:12], 0], 4).tolist()}; {out.stdout.strip().splitlines()[-1]}")
