import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
b = pkg.model.packed_bed(3000, seed=11, cd_freq=0, spacing_mult=3.0, init_vz=-1.0, aspect=(1.0, 1.0, 0.25))
p, sc = b.Initialize()
c0 = pkg.Context(0); c0.set_arith_mode("exact"); c0.set_params(p); c0.upload_scene(sc); c0.step(10000)
st = c0.download_state(); st = {k: st[k] for k in st if k[0] != 'a'}
c0.close()
def run(mode, n):
    c = pkg.Context(0); c.set_arith_mode(mode); c.set_params(p); c.upload_scene(sc); c.upload_state(st)
    c.step(n); s = c.download_state(); c.close(); return s
for n in (1, 10, 50, 100, 200, 500, 1000):
    e, f = run("exact", n), run("fast", n)
    X = pkg.model.decode_positions(e["voxelID"], e["locX"], e["locY"], e["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    Y = pkg.model.decode_positions(f["voxelID"], f["locX"], f["locY"], f["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    dv = max(np.abs(e[k] - f[k]).max() for k in ("vX", "vY", "vZ"))
    dw = max(np.abs(e[k] - f[k]).max() for k in ("omgBarX", "omgBarY", "omgBarZ"))
    i = int(np.argmax(np.abs(e["omgBarX"] - f["omgBarX"])))
    print(f"n={n:4d} dx={np.abs(X-Y).max():.3e} dv={dv:.3e} dw={dw:.3e} (owner {i}: exact w {e['omgBarX'][i]:.4f} fast {f['omgBarX'][i]:.4f}) vmax {np.abs(e['vZ']).max():.3f} wmax {np.abs(e['omgBarX']).max():.2f}")
