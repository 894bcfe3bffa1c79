"""Output writers / CSV readers against the reference's own data fixtures and formats.

tests/golden/ref_data/ holds DATA files of the reference: data/sim_data/example_cnt_pairs.csv (a contact file written by the
reference), data/clumps/3_clump.csv, ellipsoid_2_1_1.csv and 6_clump.csv (tests/test_gpu_parity.py mixes the three shapes), and
data/mesh/sphere.obj (BallDrop's projectile, tests/test_config0_balldrop.py).  CPU tests use the oracle as the state provider;
the GPU test does a write -> read -> restart round trip through the C-ABI (deme_seed_contacts)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "golden", "ref_data")
WC = ("delta_tan_x", "delta_tan_y", "delta_tan_z", "delta_time")


def test_reads_reference_clump_template(pkg):
    rel, r = pkg.io.read_clump_template_csv(os.path.join(REF, "3_clump.csv"))
    assert np.array_equal(np.c_[rel, r], pkg.model.THREE_SPHERE_CLUMP)


def test_reads_reference_contact_file(pkg):
    path = os.path.join(REF, "example_cnt_pairs.csv")
    assert pkg.io.read_contact_pairs_from_csv(path).tolist() == [[0, 1]]
    assert pkg.io.read_contact_pairs_from_csv(path, cnt_type="SA").tolist() == [[0, 0]]
    w = pkg.io.read_contact_wildcards_from_csv(path)
    # X, Y, Z are not "known" contact-file columns in the reference either (Structs.h:75-84), so they come back too
    assert list(w) == ["X", "Y", "Z", "delta_tan_x", "delta_tan_y", "delta_tan_z", "delta_time"]
    assert w["delta_time"].tolist() == [np.float32(7.84117)] and w["Z"].tolist() == [np.float32(-6.0042)]


def test_number_format_reproduces_reference_text(pkg):
    """every number in the reference-written contact file re-prints to the same text (ostream << float)"""
    lines = open(os.path.join(REF, "example_cnt_pairs.csv")).read().split()
    for line in lines[1:]:
        for tok in line.split(",")[1:]:
            assert pkg.io._g(np.float32(tok)) == tok


def _snapshot(pkg, orc, n=400, steps=150):
    b = pkg.model.packed_bed(n, seed=11, cd_freq=0, spacing_mult=2.4, init_vz=-0.5, aspect=(1.0, 1.0, 0.6))
    p, sc = b.Initialize()
    sim = orc.make_sim(pkg, p, sc)
    sim.step(steps)
    sim.compute_margins(0), sim.detect(), sim.migrate()
    sim.calc_forces(record=True)
    st = sim.download_state()
    return b, p, sim, st


def test_writers_headers_and_round_trip(pkg, orc, tmp_path):
    b, p, sim, st = _snapshot(pkg, orc)
    io = pkg.io
    cnt = sim.contacts()
    rec = sim.contact_records()
    wc = {name: sim.wildcard(w) for w, name in enumerate(WC)}
    assert (cnt[2] == 1).sum() > 50

    sph = tmp_path / "spheres.csv"
    n = io.write_sphere_file(sph, p, b.arrays, b.counts, st)
    assert n == b.counts["nSpheres"]
    assert open(sph).readline().strip() == "X,Y,Z,r,absv"  # default content QUAT|ABSV (API.h:1418)
    allf = io.OUTPUT_CONTENT
    io.write_sphere_file(sph, p, b.arrays, b.counts, st, flags=allf.ABSV | allf.VEL | allf.ANG_VEL | allf.ABS_ACC | allf.ACC |
                         allf.ANG_ACC | allf.FAMILY)
    assert open(sph).readline().strip() == ("X,Y,Z,r,absv,v_x,v_y,v_z,w_x,w_y,w_z,abs_acc,a_x,a_y,a_z,alpha_x,alpha_y,alpha_z,"
                                            "family")
    rows = np.loadtxt(sph, delimiter=",", skiprows=1)
    # sphere centres against an fp64 evaluation of the same pose
    X = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    X = X + np.array([p.LBFX, p.LBFY, p.LBFZ])
    own = b.arrays["ownerClumpBody"]
    assert np.abs(rows[::3, :3] - X[own[::3]]).max() < 0.012  # a component sits within the clump's extent
    assert np.allclose(rows[:, 3], b.arrays["Radii"][b.arrays["clumpComponentOffset"]], rtol=1e-5)

    # owner / geometry wildcard columns (OWNER_WILDCARD, GEO_WILDCARD) follow the family column, named as the model names them
    nO, nS = int(b.counts["nOwners"]), int(b.counts["nSpheres"])
    ow = {"mu_custom": np.linspace(0.1, 0.9, nO, dtype=np.float32), "cohesion": np.arange(nO, dtype=np.float32)}
    gw = {"wear": np.arange(nS, dtype=np.float32) * 0.5}
    io.write_sphere_file(sph, p, b.arrays, b.counts, st, flags=allf.FAMILY | allf.OWNER_WILDCARD | allf.GEO_WILDCARD,
                         owner_wildcards=ow, geo_wildcards=gw)
    assert open(sph).readline().strip() == "X,Y,Z,r,family,mu_custom,cohesion,wear"
    rows = np.loadtxt(sph, delimiter=",", skiprows=1)
    assert np.allclose(rows[:, 5], ow["mu_custom"][own], rtol=1e-5) and np.array_equal(rows[:, 6], ow["cohesion"][own])
    assert np.array_equal(rows[:, 7], gw["wear"])
    io.write_sphere_file(sph, p, b.arrays, b.counts, st, flags=allf.FAMILY, owner_wildcards=ow, geo_wildcards=gw)
    assert open(sph).readline().strip() == "X,Y,Z,r,family"  # bits not set: no columns

    clp = tmp_path / "clumps.csv"
    io.write_clump_file(clp, p, b.arrays, b.counts, st, b.template_names, flags=allf.OWNER_WILDCARD, owner_wildcards=ow)
    assert open(clp).readline().strip() == "X,Y,Z,Qw,Qx,Qy,Qz,clump_type,mu_custom,cohesion"
    nc = io.write_clump_file(clp, p, b.arrays, b.counts, st, b.template_names, flags=allf.ABSV | allf.VEL | allf.ANG_VEL)
    assert nc == b.counts["nOwnerClumps"]
    assert open(clp).readline().strip() == "X,Y,Z,Qw,Qx,Qy,Qz,clump_type,absv,v_x,v_y,v_z,w_x,w_y,w_z"
    xyz = io.read_clump_xyz_from_csv(clp)
    assert list(xyz) == ["0000"]  # default template name: "%04d" of the load order
    assert np.abs(xyz["0000"] - X[:nc]).max() < 1e-6
    q = io.read_clump_quat_from_csv(clp)["0000"]
    assert np.array_equal(q, np.stack([st["oriQx"], st["oriQy"], st["oriQz"], st["oriQw"]], 1)[:nc])  # 10 digits: exact
    v = io.read_clump_vel_from_csv(clp)["0000"]
    assert np.array_equal(v, np.stack([st["vX"], st["vY"], st["vZ"]], 1)[:nc])

    cf = tmp_path / "contacts.csv"
    fl = io.CNT_OUTPUT_CONTENT
    k = io.write_contact_file(cf, p, b.arrays, b.counts, st, cnt, rec, wc, flags=fl.GEO_ID | fl.FORCE | fl.CNT_POINT | fl.CNT_WILDCARD)
    # identical header to the contact file the reference ships as data
    assert open(cf).readline() == open(os.path.join(REF, "example_cnt_pairs.csv")).readline()
    F, T = rec[0], rec[1]
    active = np.sqrt(((F + T) ** 2).sum(1)) >= 1e-12
    assert k == int(active.sum()) and 0 < k <= len(F)
    pairs = io.read_contact_pairs_from_csv(cf)
    sel = active & (cnt[2] == 1)
    assert np.array_equal(pairs, np.stack([cnt[0][sel], cnt[1][sel]], 1))
    w = io.read_contact_wildcards_from_csv(cf)
    assert np.allclose(w["delta_time"], wc["delta_time"][sel], rtol=1e-5)
    # contact points lie between the two sphere centres' bounding region: within one clump diameter of A's owner
    pts = np.stack([w["X"], w["Y"], w["Z"]], 1)
    assert np.abs(pts - X[own[cnt[0][sel]]]).max() < 0.012
    # default contact content (API.h:1422): owners + geometry ids + force + point + wildcards
    io.write_contact_file(cf, p, b.arrays, b.counts, st, cnt, rec, wc)
    assert open(cf).readline().strip() == "contact_type,A,B,geoA,geoB,f_x,f_y,f_z,X,Y,Z," + ",".join(WC)
    io.write_contact_file(cf, p, b.arrays, b.counts, st, cnt, rec, wc, flags=fl.OWNER | fl.NORMAL | fl.TORQUE)
    assert open(cf).readline().strip() == "contact_type,A,B,n_x,n_y,n_z,torque_x,torque_y,torque_z"
    rows = np.loadtxt(cf, delimiter=",", skiprows=1, usecols=(3, 4, 5))
    assert np.allclose(np.linalg.norm(rows, axis=1), 1.0, atol=1e-5)


@pytest.mark.gpu
def test_restart_from_files_on_gpu(pkg, orc, tmp_path):
    """write clump + contact files after N steps, rebuild the scene from them (ReadClump*FromCsv, SetExistingContacts /
    SetExistingContactWildcards -> deme_seed_contacts), continue, and compare (a) with the oracle restarted from the same files
    and seed list -- bit for bit (exact arithmetic mode) -- and (b) with the uninterrupted run, within the file format's rounding"""
    io = pkg.io
    b = pkg.model.packed_bed(1500, seed=5, cd_freq=0, spacing_mult=2.4, init_vz=-0.5, aspect=(1.0, 1.0, 0.6))
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    ctx.set_record_contacts(True)
    ctx.step(300)
    ctx.compute_margins(0), ctx.detect(), ctx.migrate(), ctx.calc_forces()
    st = ctx.download_state()
    cnt, rec = ctx.contacts(), ctx.contact_records()
    wc = {name: ctx.wildcard(w) for w, name in enumerate(WC)}
    allf, fl = io.OUTPUT_CONTENT, io.CNT_OUTPUT_CONTENT
    io.write_clump_file(tmp_path / "c.csv", p, b.arrays, b.counts, st, b.template_names, flags=allf.VEL | allf.ANG_VEL, accuracy=10)
    # keep every list entry (force_thres < 0) so that the whole history is carried; 9 digits keep fp32 exactly
    io.write_contact_file(tmp_path / "k.csv", p, b.arrays, b.counts, st, cnt, rec, wc, flags=fl.GEO_ID | fl.CNT_WILDCARD,
                          force_thres=-1.0, precision=9)

    # ---- restart
    b2 = pkg.model.packed_bed(1500, seed=5, cd_freq=0, spacing_mult=2.4, init_vz=-0.5, aspect=(1.0, 1.0, 0.6))
    batch = b2.batches[0]
    batch.xyz[:] = io.read_clump_xyz_from_csv(tmp_path / "c.csv")["0000"]
    batch.SetOriQ(io.read_clump_quat_from_csv(tmp_path / "c.csv")["0000"])
    batch.SetVel(io.read_clump_vel_from_csv(tmp_path / "c.csv")["0000"])
    batch.SetAngVel(io.read_clump_angvel_from_csv(tmp_path / "c.csv")["0000"])
    p2, sc2 = b2.Initialize()
    ctx2 = pkg.Context(0)
    ctx2.set_params(p2), ctx2.upload_scene(sc2)
    pairs_ss = io.read_contact_pairs_from_csv(tmp_path / "k.csv", "SS")
    w_ss = io.read_contact_wildcards_from_csv(tmp_path / "k.csv", "SS")
    pairs_sa = io.read_contact_pairs_from_csv(tmp_path / "k.csv", "SA")
    w_sa = io.read_contact_wildcards_from_csv(tmp_path / "k.csv", "SA")
    ida = np.r_[pairs_ss[:, 0], pairs_sa[:, 0]]
    idb = np.r_[pairs_ss[:, 1], pairs_sa[:, 1]]
    ty = np.r_[np.full(len(pairs_ss), 1, np.uint8), np.full(len(pairs_sa), 11, np.uint8)]
    W = np.stack([np.r_[w_ss[n], w_sa[n]] for n in WC], 1)
    assert len(ida) == len(cnt[0])
    ctx2.seed_contacts(ida, idb, ty, W)
    sim = orc.make_sim(pkg, p2, sc2)  # the oracle, restarted from the same files
    sim.seed_contacts(ida, idb, ty, W)
    ctx2.step(1), sim.step(1)
    assert np.array_equal(ctx2.wildcard(3), sim.wildcard(3)) and all(np.array_equal(x, y) for x, y in zip(ctx2.contacts(), sim.contacts()))
    # the seeded history was found by the first detection: delta_time kept counting instead of restarting at h
    ctx.step(1)
    dt_a, dt_b = ctx.wildcard(3), ctx2.wildcard(3)
    assert len(dt_a) == len(dt_b) and (dt_b > 10 * p.h).sum() > 100
    assert np.allclose(dt_a, dt_b, rtol=1e-6, atol=1e-9)
    ctx.step(99), ctx2.step(99), sim.step(99)
    sa, sb, so = ctx.download_state(), ctx2.download_state(), sim.download_state()
    assert int(ctx2.counts().nContacts) == int(sim.counts().nContacts)
    for k in ("voxelID", "locX", "locY", "locZ", "vX", "vY", "vZ", "oriQw", "oriQx", "omgBarX", "omgBarZ"):
        assert np.array_equal(sb[k], so[k]), k
    for w in range(len(WC)):
        assert np.array_equal(ctx2.wildcard(w), sim.wildcard(w)), WC[w]
    n = int(sc.nOwnerClumps)
    Xa = pkg.model.decode_positions(sa["voxelID"], sa["locX"], sa["locY"], sa["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:n]
    Xb = pkg.model.decode_positions(sb["voxelID"], sb["locX"], sb["locY"], sb["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:n]
    # the clump file stores fp32 positions (the reference's format): ~3e-8 m of rounding at restart
    assert np.abs(Xa - Xb).max() < 2e-6
    assert abs(int(ctx.counts().nContacts) - int(ctx2.counts().nContacts)) <= 3


@pytest.mark.gpu
def test_update_clumps_appends_to_a_running_simulation(pkg, orc):
    """UpdateClumps (API.h:1267): a second batch poured onto a running bed.  The old clumps carry on exactly as in a run
    without the addition until the newcomers reach them; the combined scene then matches the oracle given the same state."""
    def bed():
        return pkg.model.packed_bed(1200, seed=41, cd_freq=0, spacing_mult=2.4, init_vz=-0.5, aspect=(1.0, 1.0, 0.6))
    b = bed()
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    ctx.step(150)
    ref = pkg.Context(0)
    ref.set_params(p), ref.upload_scene(sc)
    ref.step(150)
    n_old = int(sc.nOwnerClumps)
    # a layer of new clumps well above the bed, falling
    top = float(b.batches[0].xyz[:, 2].max())
    xy = b.batches[0].xyz[:300, :2]
    new = np.c_[xy, np.full(len(xy), top + 0.03, np.float32)].astype(np.float32)
    batch = b.AddClumps(b.templates[0], new)
    batch.SetVel(np.tile(np.array([0, 0, -1.0], np.float32), (len(new), 1)))
    p2, sc2 = b.UpdateClumps(ctx, time_elapsed=150 * p.h)
    assert int(sc2.nOwnerClumps) == n_old + 300 and int(ctx.counts().nContacts) == int(ref.counts().nContacts)
    ctx.step(100), ref.step(100)  # the newcomers are still in the air: the old clumps must not notice them
    a, r = ctx.download_state(), ref.download_state()
    for k in ("voxelID", "locX", "locY", "locZ", "vX", "vY", "vZ", "oriQw", "omgBarZ"):
        assert np.array_equal(a[k][:n_old], r[k][:n_old]), k
    assert np.array_equal(ctx.wildcard(3)[: len(ref.wildcard(3))].sum() > 0, True)
    # from here on: oracle with the same combined state
    sim = orc.make_sim(pkg, p2, sc2)
    sim.upload_state({k: a[k] for k in a if not k.startswith(("a", "alpha"))})
    cnt = ctx.contacts()
    sim.seed_contacts(cnt[0], cnt[1], cnt[2], np.stack([ctx.wildcard(w) for w in range(4)], 1))
    ctx.step(3000), sim.step(3000)
    g, o = ctx.download_state(), sim.download_state()
    assert int(ctx.counts().nContacts) == int(sim.counts().nContacts)
    for k in ("voxelID", "locX", "locZ", "vZ", "oriQw"):
        assert np.array_equal(g[k], o[k]), k
    newcomers_touch = (np.isin(ctx.contacts()[0] // 3, np.arange(n_old, n_old + 300))).sum()
    assert newcomers_touch > 20  # the poured layer has landed on the bed


def test_mesh_file_vtk(pkg, orc, tmp_path):
    """WriteMeshFile: legacy VTK of the mesh in its current pose (vertices rotated and translated with the owner)"""
    b = pkg.model.packed_bed(60, seed=2, cd_freq=0)
    v, f = pkg.model.plate_mesh(3, 2, 0.06, 0.04, z=0.0, wavy=0.0)
    m = b.AddMeshObject(v, f, 0)
    m.SetInitPos((0.05, 0.06, 0.01))
    m.SetInitQuat((0.0, 0.0, np.sin(np.pi / 4), np.cos(np.pi / 4)))  # 90 degrees about z
    p, sc = b.Initialize()
    sim = orc.make_sim(pkg, p, sc)
    path = tmp_path / "mesh.vtk"
    nv, nf = pkg.io.write_mesh_file(path, p, b.meshes, b.counts, sim.download_state())
    txt = open(path).read().split("\n")
    assert txt[0] == "# vtk DataFile Version 2.0" and txt[5] == "DATASET UNSTRUCTURED_GRID" and txt[6] == f"POINTS {nv} float"
    pts = np.array([[float(x) for x in l.split()] for l in txt[7:7 + nv]])
    v = np.asarray(v, np.float64)
    expect = np.stack([-v[:, 1], v[:, 0], v[:, 2]], 1) + np.array([0.05, 0.06, 0.01])  # rotated by 90 degrees, then moved
    assert nv == len(v) and nf == len(f) and np.abs(pts - expect).max() < 1e-6
    i = txt.index(f"CELLS {nf} {4 * nf}")
    cells = np.array([[int(x) for x in l.split()] for l in txt[i + 1:i + 1 + nf]])
    assert (cells[:, 0] == 3).all() and np.array_equal(cells[:, 1:], np.asarray(f))
    j = txt.index(f"CELL_TYPES {nf}")
    assert all(l.strip() == "5" for l in txt[j + 1:j + 1 + nf])
