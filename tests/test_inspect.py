"""Inspectors: device reductions of deme_inspect against the oracle's restatement of the
reference's query kernels (DEMSphereQueryKernels.cu, DEMOwnerQueryKernels.cu, AuxClasses.cpp:19-170)."""
import numpy as np
import pytest

QUANTITIES = ("clump_max_z", "clump_min_z", "clump_max_absv", "clump_mass", "max_absv", "clump_kinetic_energy")


def _bed(pkg, n=1200):
    b = pkg.model.packed_bed(n, seed=21, cd_freq=0, spacing_mult=2.4, init_vz=-0.5, aspect=(1.0, 1.0, 0.6))
    p, sc = b.Initialize()
    return b, p, sc


def test_oracle_inspectors_against_numpy(pkg, orc):
    b, p, sc = _bed(pkg, 500)
    sim = orc.make_sim(pkg, p, sc)
    sim.step(120)
    st = sim.download_state()
    n = int(sc.nOwnerClumps)
    X = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    X = X + np.array([p.LBFX, p.LBFY, p.LBFZ])
    q = np.stack([st["oriQw"], st["oriQx"], st["oriQy"], st["oriQz"]], 1)
    own, comp = b.arrays["ownerClumpBody"], b.arrays["clumpComponentOffset"]
    rel = np.stack([b.arrays["CDRelPosX"], b.arrays["CDRelPosY"], b.arrays["CDRelPosZ"]], 1)[comp]
    z = X[own, 2] + pkg.io.rotate_f32(q[own], rel)[:, 2]
    r = b.arrays["Radii"][comp]
    assert abs(sim.inspect("clump_max_z") - (z + r).max()) < 1e-6
    assert abs(sim.inspect("clump_min_z") - (z - r).min()) < 1e-6
    m = b.arrays["MassProperties"][b.arrays["inertiaPropOffsets"][:n]].astype(np.float64)
    assert abs(sim.inspect("clump_mass") - m.sum()) < 1e-6 * m.sum()
    v = np.stack([st["vX"], st["vY"], st["vZ"]], 1).astype(np.float64)
    assert abs(sim.inspect("max_absv") - np.sqrt((v ** 2).sum(1)).max()) < 1e-6
    w = np.stack([st["omgBarX"], st["omgBarY"], st["omgBarZ"]], 1).astype(np.float64)[:n]
    moi = np.stack([b.arrays["moiX"], b.arrays["moiY"], b.arrays["moiZ"]], 1)[b.arrays["inertiaPropOffsets"][:n]].astype(np.float64)
    ke = 0.5 * m * (v[:n] ** 2).sum(1) + 0.5 * (moi * w ** 2).sum(1)
    assert abs(sim.inspect("clump_kinetic_energy") - ke.sum()) < 1e-5 * ke.sum()
    assert sim.inspect("clump_max_absv") >= np.sqrt((v[:n] ** 2).sum(1)).max() * 0.5
    absv = sim.inspect("absv", values=True)
    assert len(absv) == int(sc.nOwners) and np.allclose(absv, np.sqrt((v ** 2).sum(1)), rtol=1e-6)


@pytest.mark.gpu
def test_gpu_inspectors_match_oracle(pkg, orc):
    b, p, sc = _bed(pkg)
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    sim = orc.make_sim(pkg, p, sc)
    ctx.step(150)
    sim.upload_state({k: v for k, v in ctx.download_state().items() if k in
                      ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY",
                       "omgBarZ")})
    for name in QUANTITIES:
        g, o = ctx.inspect(name), sim.inspect(name)
        if name in ("clump_mass", "clump_kinetic_energy"):  # fp32 tree sum on the device vs fp64 accumulation
            assert abs(g - o) <= 2e-5 * abs(o), name
        else:  # max / min of identical per-element values: exact
            assert g == o, name
    for name, n in (("clump_max_absv", int(sc.nSpheres)), ("absv", int(sc.nOwners)), ("clump_kinetic_energy", int(sc.nOwners))):
        assert np.array_equal(ctx.inspect_values(name, n), sim.inspect(name, values=True)), name
    with pytest.raises(pkg.abi.DemeError):
        ctx.inspect("no_such_quantity")
