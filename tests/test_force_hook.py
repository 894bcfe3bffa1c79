"""The jitified contact-model hook: user statement blocks written against the reference's
ingredient names compile through hipRTC.  CPU tests cover generation + compilation for gfx950 (no GPU
needed); GPU tests run the compiled model through the C-ABI."""
import os

import numpy as np
import pytest

FRICTIONLESS = r"""
if (overlapDepth > 0) {
    float E_cnt;
    matProxy2ContactParam<float>(E_cnt, E[bodyAMatType], nu[bodyAMatType], E[bodyBMatType], nu[bodyBMatType]);
    const float CoR_cnt = CoR[bodyAMatType][bodyBMatType];
    float3 rotVelCPA = cross(ARotVel, locCPA), rotVelCPB = cross(BRotVel, locCPB);
    applyOriQToVector3<float, deme::oriQ_t>(rotVelCPA.x, rotVelCPA.y, rotVelCPA.z, AOriQ.w, AOriQ.x, AOriQ.y, AOriQ.z);
    applyOriQToVector3<float, deme::oriQ_t>(rotVelCPB.x, rotVelCPB.y, rotVelCPB.z, BOriQ.w, BOriQ.x, BOriQ.y, BOriQ.z);
    const float3 velB2A = (ALinVel + rotVelCPA) - (BLinVel + rotVelCPB);
    const float projection = dot(velB2A, B2A);
    const float mass_eff = (AOwnerMass * BOwnerMass) / (AOwnerMass + BOwnerMass);
    const float sqrt_Rd = sqrt(overlapDepth * (ARadius * BRadius) / (ARadius + BRadius));
    const float Sn = 2. * E_cnt * sqrt_Rd;
    const float loge = (CoR_cnt < DEME_TINY_FLOAT) ? log(DEME_TINY_FLOAT) : log(CoR_cnt);
    const float beta = loge / sqrt(loge * loge + deme::PI_SQUARED);
    const float k_n = deme::TWO_OVER_THREE * Sn;
    const float gamma_n = deme::TWO_TIMES_SQRT_FIVE_OVER_SIX * beta * sqrt(Sn * mass_eff);
    force += (k_n * overlapDepth + gamma_n * projection) * B2A;
"""
COHESIVE = FRICTIONLESS + """
    force += -Cohesion[bodyAMatType][bodyBMatType] * B2A;   // pairwise user property
    contact_age += ts;                                        // a user contact wildcard
}
"""
PLAIN = FRICTIONLESS + "}\n"


def test_fragment_compiles_for_gfx950_without_a_gpu(pkg):
    pre = "__device__ const float Cohesion[][2] = {{0.01f, 0.02f}, {0.02f, 0.03f}};\n"
    ok, log = pkg.abi.jit_probe(COHESIVE, ["contact_age"], pre)
    assert ok, log
    ok, log = pkg.abi.jit_probe(PLAIN, [], "")
    assert ok, log


def test_user_kernel_includes_reach_the_runtime_compiler(pkg, tmp_path, monkeypatch):
    """DEMSolver::AddKernelInclude / SetKernelInclude (DEM/API.h:1362-1367; the reference's _kernelIncludes_): a model that needs
    a header of the user's compiles once the header's directory is on DEME_KERNEL_INCLUDE_PATH, and fails to without it"""
    (tmp_path / "my_contact_helpers.h").write_text("__device__ inline float my_softening(float d) { return 0.5f * d; }\n")
    body = PLAIN.replace("force += (k_n * overlapDepth", "force += my_softening(1.0f) * (k_n * overlapDepth")
    pre = "#include <my_contact_helpers.h>\n"
    monkeypatch.delenv("DEME_KERNEL_INCLUDE_PATH", raising=False)
    ok, log = pkg.abi.jit_probe(body, [], pre)
    assert not ok and "my_contact_helpers.h" in log
    monkeypatch.setenv("DEME_KERNEL_INCLUDE_PATH", str(tmp_path))
    ok, log = pkg.abi.jit_probe(body, [], pre)
    assert ok, log


def test_name_clash_and_syntax_errors_are_reported(pkg):
    ok, log = pkg.abi.jit_probe(PLAIN, ["force"], "")  # APIPrivate.cpp:1425-1465: wildcard clashes with an ingredient
    assert not ok and "force" in log
    ok, log = pkg.abi.jit_probe(PLAIN, ["a", "a"], "")
    assert not ok and "twice" in log
    ok, log = pkg.abi.jit_probe("force = undeclared_thing;", [], "")
    assert not ok and "undeclared_thing" in log


def test_builder_emits_material_arrays(pkg):
    b = pkg.SceneBuilder()
    b.LoadMaterial({"E": 1e8, "nu": 0.3, "CoR": 0.6, "Cohesion": 0.01, "Gamma": 2.0})
    b.LoadMaterial({"E": 1e9, "nu": 0.3, "CoR": 0.6, "Cohesion": 0.03, "Gamma": 4.0})
    b.SetMustPairwiseMatProp(["Cohesion"])
    b.SetMaterialPropertyPair("Cohesion", 0, 1, 0.5)
    pre = b.force_model_prerequisites()
    assert "Cohesion[][2]" in pre and "0.5f" in pre and "Gamma[] = {2.0f, 4.0f}" in pre
    b.SetPerContactWildcards(["zeta", "alpha"])
    assert b.contact_wildcards == ["alpha", "zeta"]  # std::set order


@pytest.mark.gpu
def test_custom_model_matches_builtin_and_adds_cohesion(pkg):
    def run(kind):
        b = pkg.model.packed_bed(2000, seed=31, cd_freq=0, spacing_mult=2.5, init_vz=-0.4, force_model=1)
        b.materials[0]["Cohesion"] = 0.004
        b.SetMustPairwiseMatProp(["Cohesion"])
        if kind == "plain":
            b.DefineContactForceModel(PLAIN)
            b.SetPerContactWildcards([])
        elif kind == "cohesive":
            b.DefineContactForceModel(COHESIVE)
            b.SetPerContactWildcards(["contact_age"])
        p, sc = b.Initialize()
        ctx = pkg.Context(0)
        ctx.set_params(p)
        ctx.upload_scene(sc)
        b.compile_into(ctx)
        ctx.set_record_contacts(True)
        ctx.detect()
        ctx.calc_forces()
        F = ctx.contact_records()[0]
        age = ctx.wildcard(0) if kind == "cohesive" else None
        return F, age, ctx.contacts(), p

    F0, _, c0, p = run("builtin")
    F1, _, c1, _ = run("plain")
    F2, age, c2, _ = run("cohesive")
    assert (c0[0] == c1[0]).all() and (c0[1] == c2[1]).all()
    scale = np.abs(F0).max()
    assert scale > 0 and np.abs(F1 - F0).max() <= 2e-6 * scale  # float vs double sqrt/log overloads: <= 1 ulp
    touching = np.linalg.norm(F0, axis=1) > 0
    d = np.linalg.norm(F2 - F1, axis=1)
    assert touching.sum() > 500
    # |F| is O(10 N) in fp32, so the 0.004 N shift is resolved to ~1e-5 N
    assert np.allclose(d[touching], 0.004, atol=5e-5) and (d[~touching] == 0).all()
    assert np.allclose(age[touching], p.h) and (age[~touching] == 0).all()


@pytest.mark.gpu
def test_custom_cohesive_model_forces_match_the_oracle(pkg, orc):
    """the run-time compiled fragment against the ORACLE's parametric form of the same model (oracle/deme_oracle.cpp,
    customKind 1), contact by contact: hipRTC's device log / sqrt against the host's libm leave <= 2 ulp"""
    b = pkg.model.packed_bed(2000, seed=31, cd_freq=0, spacing_mult=2.5, init_vz=-0.4, force_model=1)
    b.materials[0]["Cohesion"] = 0.004
    b.SetMustPairwiseMatProp(["Cohesion"])
    b.DefineContactForceModel(COHESIVE)
    b.SetPerContactWildcards(["contact_age"])
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    b.compile_into(ctx)
    ctx.set_record_contacts(True)
    ctx.detect(), ctx.calc_forces()
    sim = orc.make_sim(pkg, p, sc)
    sim.set_custom_model(1, [[0.004]])
    sim.detect(), sim.calc_forces(record=True)
    assert all(np.array_equal(x, y) for x, y in zip(ctx.contacts()[:3], sim.contacts()[:3]))
    F, Fo = ctx.contact_records()[0], sim.contact_records()[0]
    scale = np.abs(Fo).max()
    assert scale > 1.0 and np.abs(F - Fo).max() <= 4e-7 * scale
    assert np.array_equal(ctx.wildcard(0), sim.wildcard(0))
    g, o = ctx.download_state(), sim.download_state()
    n = int(sc.nOwnerClumps)
    for k in ("aX", "aY", "aZ", "alphaX", "alphaY", "alphaZ"):
        assert np.abs(g[k][:n] - o[k][:n]).max() <= 1e-6 * max(np.abs(o[k][:n]).max(), 1e-30), k


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_config5_polydisperse_cohesive_bed_against_the_oracle(pkg, orc, mode):
    """BASELINE configs[4] at test size: bench.build_config5 -- 1e5 single spheres of 8 radii in [r, 3r], the cohesion +
    contact-age fragment compiled by hipRTC -- stepped 60 times on the GPU and on the oracle from the same packed state.
    STATED TOLERANCE: contact sets identical; positions within 1e-8 m and velocities within 2e-5 m/s after 60 steps
    (h = 5e-6 s).  The fragment's log / sqrt come from the device math library on one side and libm on the other (<= 2 ulp),
    and in the fast mode the per-owner accumulation is the world-frame one."""
    import bench
    n = 100_000
    b = bench.build_config5(pkg, n, seed=2024, cd_freq=20)
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_arith_mode(mode)
    ctx.set_params(p), ctx.upload_scene(sc)
    b.compile_into(ctx)
    ctx.set_tile_policy(0)  # the tile pass whatever the density (by default this bed -- 200 contacts per tile -- keeps the general kernel)
    for _ in range(20):  # settle: the lattice is dropped at 1 m/s
        ctx.step(3000)
        if int(ctx.counts().nContacts) > 1.5 * n:
            break
    st = ctx.download_state()
    st = {k: st[k] for k in st if k not in ("aX", "aY", "aZ", "alphaX", "alphaY", "alphaZ")}
    a, bb, t, _ = ctx.contacts()
    age = ctx.wildcard(0)
    assert len(a) > 1.5 * n, "the bed should be packed"
    sim = orc.make_sim(pkg, p, sc)
    sim.set_custom_model(1, np.full((int(sc.nMat), int(sc.nMat)), 0.002, np.float32))
    for s in (ctx, sim):  # both continue from the same state and the same history (restart path)
        s.upload_state(st)
        s.seed_contacts(a, bb, t, age.reshape(-1, 1))
    orc.set_num_threads(min(32, os.cpu_count() or 1))
    N = 60
    ctx.step(N), sim.step(N)
    orc.set_num_threads(min(8, os.cpu_count() or 1))
    if mode == "fast":  # the user's statements compiled into the owner-tile pass (deme_jit.h + deme_tile.h MODEL 2)
        assert ctx.force_kernel()[0] == "deme_custom_tile<false>", ctx.force_kernel()
    else:
        assert ctx.force_kernel()[0] == "deme_custom_forces_ss", ctx.force_kernel()
    ga, oa = ctx.contacts(), sim.contacts()
    assert len(ga[0]) == len(oa[0]) and all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    g, o = ctx.download_state(), sim.download_state()
    X = pkg.model.decode_positions(g["voxelID"], g["locX"], g["locY"], g["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    Y = pkg.model.decode_positions(o["voxelID"], o["locX"], o["locY"], o["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    dx = np.abs(X - Y).max()
    dv = max(np.abs(g[k] - o[k]).max() for k in ("vX", "vY", "vZ"))
    dage = np.abs(ctx.wildcard(0) - sim.wildcard(0)).max()
    print(f"config5 ({mode}): {len(ga[0])} contacts, |dx| {dx:.3e} m, |dv| {dv:.3e} m/s, |d age| {dage:.3e} s after {N} steps")
    assert dx <= 1e-8 and dv <= 2e-5 and dage <= 1e-9
    ctx.close()


# what the reference's demos declare for each script under kernel/DEMUserScripts (SetPerContactWildcards /
# SetPerGeometryWildcards / SetMustPairwiseMatProp in src/demo/DEMdemo_*.cpp; the mooring scripts document theirs in comments)
HIST = ["delta_time", "delta_tan_x", "delta_tan_y", "delta_tan_z"]
USER_SCRIPTS = {
    "ForceModel2D.cu": (HIST, [], []),
    "ForceModelMooring.cu": (HIST + ["unbroken", "initialLength", "innerInteraction"], [], []),
    "ForceModelMooringPosition.cu": (HIST + ["unbroken", "initialLength", "innerInteraction", "tension"], [], []),
    "ForceModelWithCohesion.cu": (HIST, [], ["Cohesion"]),
    "ForceModelWithElectrostatic.cu": (HIST, ["Q"], []),
    "ForceModelWithFractureModel.cu": (HIST + ["unbroken", "initialLength"], [], []),
    "ForceModelWithGravity.cu": ([], ["my_mass"], []),
}


def test_reference_user_scripts_compile_unchanged(pkg):
    """every fragment file under the reference's kernel/DEMUserScripts/ goes through the force-model generator and hipRTC
    unchanged (read where it lies, never copied; skipped where the reference tree is absent).  No GPU is needed to compile."""
    d = "/root/reference/src/kernel/DEMUserScripts"
    if not os.path.isdir(d):
        pytest.skip("the reference tree is only present in the build container")
    files = sorted(f for f in os.listdir(d) if f.endswith(".cu"))
    assert len(files) >= 7 and set(files) <= set(USER_SCRIPTS), files
    for f in files:
        contact_wc, geo_wc, pair_props = USER_SCRIPTS[f]
        # pairwise material properties beyond the built-in CoR / mu / Crr become a constant table, as the set-up code emits them
        pre = "".join(f"__device__ const float {q}[][2] = {{{{50.f, 100.f}}, {{100.f, 50.f}}}};\n" for q in pair_props)
        ok, log = pkg.abi.jit_probe(open(os.path.join(d, f)).read(), contact_wc, pre, geo_wildcards=geo_wc)
        assert ok, f"{f}: {log[-1500:]}"
