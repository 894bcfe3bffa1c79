"""Inspectors: device reductions of deme_inspect against the oracle's restatement of the
reference's query kernels (DEMSphereQueryKernels.cu, DEMOwnerQueryKernels.cu, AuxClasses.cpp:19-170)."""
import numpy as np
import pytest

QUANTITIES = ("clump_max_z", "clump_min_z", "clump_max_absv", "clump_mass", "max_absv", "clump_kinetic_energy")


def _bed(pkg, n=1200, volumes=False):
    b = pkg.model.packed_bed(n, seed=21, cd_freq=0, spacing_mult=2.4, init_vz=-0.5, aspect=(1.0, 1.0, 0.6))
    if volumes:  # SetVolume is a template declaration (pre-Initialize), like the reference's
        for t in b.templates:
            t.SetVolume(4.0 / 3.0 * np.pi * float((t.radii.astype(np.float64) ** 3).sum()))
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


def _f(x):
    """float literal that reads back as exactly this fp32 value"""
    return f"{float(np.float32(x)):.9g}f"


def _box_code(lo, hi):
    terms = [f"({a} >= {_f(l)}) && ({a} <= {_f(h)})" for a, l, h in zip("XYZ", lo, hi)]
    return "bool inside = " + " && ".join(terms) + ";\nreturn inside;"


def test_oracle_region_and_volume(pkg, orc):
    b, p, sc = _bed(pkg, 400)
    sim = orc.make_sim(pkg, p, sc)
    n = int(sc.nOwnerClumps)
    st = sim.download_state()
    X = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    X = (X + np.array([p.LBFX, p.LBFY, p.LBFZ])).astype(np.float32)[:n]
    lo, hi = np.percentile(X, 25, axis=0).astype(np.float32), np.percentile(X, 80, axis=0).astype(np.float32)
    inside = ((X >= lo) & (X <= hi)).all(1)
    m = b.arrays["MassProperties"][b.arrays["inertiaPropOffsets"][:n]].astype(np.float64)
    assert 0 < inside.sum() < n
    assert abs(sim.inspect("clump_mass", box=(lo, hi)) - m[inside].sum()) < 1e-6 * m.sum()
    vols = np.arange(1, int(sc.nMassProps) + 1, dtype=np.float32) * 1e-7
    sim.set_volumes(vols)
    v = vols[b.arrays["inertiaPropOffsets"][:n]].astype(np.float64)
    assert abs(sim.inspect("clump_volume") - v.sum()) < 1e-6 * v.sum()
    assert abs(sim.inspect("clump_volume", box=(lo, hi)) - v[inside].sum()) < 1e-6 * v.sum()
    far = (np.full(3, 1e3, np.float32), np.full(3, 2e3, np.float32))
    assert sim.inspect("clump_mass", box=far) == 0.0 and sim.inspect("clump_max_z", box=far) < -3e38


@pytest.mark.gpu
def test_gpu_region_inspectors_match_oracle(pkg, orc):
    """CreateInspector(quantity, region): the region string is compiled at run time; per-element membership is the same fp32
    comparison as the oracle's box, so max / min agree exactly and the sums to fp32 summation order."""
    b, p, sc = _bed(pkg, volumes=True)
    assert b.volumes().any()
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    b.compile_into(ctx)
    sim = orc.make_sim(pkg, p, sc)
    sim.set_volumes(b.volumes())
    ctx.step(150)
    st = ctx.download_state()
    sim.upload_state({k: v for k, v in st.items() if k in
                      ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY",
                       "omgBarZ")})
    n = int(sc.nOwnerClumps)
    X = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    X = (X + np.array([p.LBFX, p.LBFY, p.LBFZ]))[:n]
    lo, hi = np.percentile(X, 20, axis=0).astype(np.float32), np.percentile(X, 75, axis=0).astype(np.float32)
    rid = ctx.compile_region(_box_code(lo, hi))
    assert rid == 0 and ctx.compile_region("return Z > " + _f(hi[2]) + ";") == 1
    whole = {}
    for name in QUANTITIES + ("clump_volume",):
        g, o, whole[name] = ctx.inspect(name, region=rid), sim.inspect(name, box=(lo, hi)), ctx.inspect(name)
        if name in ("clump_mass", "clump_kinetic_energy", "clump_volume"):
            assert abs(g - o) <= 2e-5 * abs(o) and 0 < g < whole[name], name
        else:
            assert g == o, name
    assert ctx.inspect("clump_max_z", region=rid) < whole["clump_max_z"]
    top = ctx.inspect("clump_mass", region=1)
    big = (np.full(3, -1e3, np.float32), np.array([1e3, 1e3, hi[2]], np.float32))
    assert abs(top + sim.inspect("clump_mass", box=big) - whole["clump_mass"]) <= 4e-5 * whole["clump_mass"]
    with pytest.raises(pkg.abi.DemeError, match="X, Y and Z"):
        ctx.compile_region("return true;")
    with pytest.raises(pkg.abi.DemeError, match="compile"):
        ctx.compile_region("return X > nonsense_symbol;")
    with pytest.raises(pkg.abi.DemeError, match="region"):
        ctx.inspect("clump_mass", region=7)
