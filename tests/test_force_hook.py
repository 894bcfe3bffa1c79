"""The jitified contact-model hook: user statement blocks written against the reference's
ingredient names compile through hipRTC.  CPU tests cover generation + compilation for gfx950 (no GPU
needed); GPU tests run the compiled model through the C-ABI."""
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
def test_custom_model_requires_compilation(pkg):
    b = pkg.model.packed_bed(300, seed=1, cd_freq=0, spacing_mult=2.5)
    b.DefineContactForceModel(PLAIN)
    b.SetPerContactWildcards([])
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_params(p)
    ctx.upload_scene(sc)
    with pytest.raises(pkg.abi.DemeError, match="none compiled"):
        ctx.step(1)
    with pytest.raises(pkg.abi.DemeError, match="compile"):
        ctx.compile_force_model("force = nonsense;", [], "")


ELECTRO = FRICTIONLESS + """
}
// screened Coulomb-like pair force between charged spheres (geometry wildcard `charge`, one value per sphere) acting on every
// list entry, touching or not; and a per-owner counter of touching contacts (owner wildcard `n_touch`, updated atomically)
{
    const float qq = charge_A[AGeo] * charge_B[BGeo];
    force += (float)(2.5e-3 * qq) * B2A;
    if (overlapDepth > 0) {
        atomicAdd(n_touch + AOwner, 1.0f);
        atomicAdd(n_touch_B + BOwner, 1.0f);
    }
}
"""


@pytest.mark.gpu
def test_owner_and_geometry_wildcards(pkg):
    """SetPerOwnerWildcards / SetPerGeometryWildcards (Models.h:319-360): owner arrays as `name`, `name_A`, `name_B`
    aliases, geometry arrays as `name_A[AGeo]`, `name_B[BGeo]` with B's array chosen by the contact kind"""
    b = pkg.model.packed_bed(2000, seed=31, cd_freq=0, spacing_mult=2.5, init_vz=-0.4, force_model=1)
    b.AddBCPlane((0.0, 0.0, 0.0174), (0, 0, 1), 0)  # a floor right under the lattice: sphere-analytical contacts at once
    b.DefineContactForceModel(ELECTRO)
    b.SetPerContactWildcards([])
    b.SetPerOwnerWildcards(["n_touch"])
    b.SetPerGeometryWildcards(["charge"])
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    b.compile_into(ctx)
    nS, nA, nO = int(sc.nSpheres), int(sc.nAnal), int(sc.nOwners)
    rng = np.random.default_rng(3)
    q_sph = rng.choice([-1.0, 1.0, 2.0], nS).astype(np.float32)
    q_wall = np.full(nA, 3.0, np.float32)
    ctx.set_wildcard_array("sphere", 0, q_sph)
    ctx.set_wildcard_array("analytical", 0, q_wall)
    # reference run without the extra terms: the plain frictionless fragment, stepped until the bed sits on the floor;
    # its state is then given to the context under test so that both evaluate the same configuration
    b0 = pkg.model.packed_bed(2000, seed=31, cd_freq=0, spacing_mult=2.5, init_vz=-0.4, force_model=1)
    b0.AddBCPlane((0.0, 0.0, 0.0174), (0, 0, 1), 0)
    b0.DefineContactForceModel(PLAIN)
    b0.SetPerContactWildcards([])
    p0, sc0 = b0.Initialize()
    c0 = pkg.Context(0)
    c0.set_params(p0), c0.upload_scene(sc0)
    b0.compile_into(c0)
    c0.step(300)
    st = c0.download_state()
    ctx.upload_state({k: st[k] for k in st if not k.startswith(("a", "alpha"))})
    c0.set_record_contacts(True)
    c0.compute_margins(0), c0.detect(), c0.calc_forces()
    F1 = c0.contact_records()[0]
    ctx.set_record_contacts(True)
    ctx.compute_margins(0), ctx.detect(), ctx.calc_forces()
    F2 = ctx.contact_records()[0]
    a, bb, t, _ = ctx.contacts()
    assert np.array_equal(c0.contacts()[0], a)
    qq = q_sph[a] * np.where(t == 1, q_sph[np.minimum(bb, nS - 1)], q_wall[np.minimum(bb, nA - 1)])
    extra = np.linalg.norm(F2 - F1, axis=1)
    assert (t != 1).sum() > 20 and np.allclose(extra, 2.5e-3 * np.abs(qq), rtol=2e-3, atol=2e-6)
    touching = np.linalg.norm(F1, axis=1) > 0
    own = b.arrays["ownerClumpBody"]
    ownB = np.where(t == 1, own[np.minimum(bb, nS - 1)], b.arrays["objOwner"][np.minimum(bb, nA - 1)])
    expect = np.bincount(own[a][touching], minlength=nO) + np.bincount(ownB[touching], minlength=nO)
    assert np.array_equal(ctx.wildcard_array("owner", 0, nO), expect.astype(np.float32))  # additions of 1.0f are exact
    assert expect.max() > 5


@pytest.mark.gpu
def test_user_wildcard_arrays_follow_resort_and_update(pkg):
    """owner / sphere wildcard arrays of a user model travel with their owners / spheres through ResortClumps (renumbering) and
    UpdateClumps (appending), analytical ones stay put"""
    b = pkg.model.packed_bed(1500, seed=31, cd_freq=0, spacing_mult=2.5, init_vz=-0.4, force_model=1, order="random")
    b.AddBCPlane((0.0, 0.0, 0.0174), (0, 0, 1), 0)
    b.DefineContactForceModel(ELECTRO)
    b.SetPerContactWildcards([])
    b.SetPerOwnerWildcards(["n_touch"])
    b.SetPerGeometryWildcards(["charge"])
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    b.compile_into(ctx)
    nS, nA, nO, nC = int(sc.nSpheres), int(sc.nAnal), int(sc.nOwners), int(sc.nOwnerClumps)
    rng = np.random.default_rng(5)
    q_sph = rng.uniform(-1, 1, nS).astype(np.float32)
    q_wall = np.arange(1, nA + 1, dtype=np.float32)
    ctx.set_wildcard_array("sphere", 0, q_sph)
    ctx.set_wildcard_array("analytical", 0, q_wall)
    ctx.step(30)
    touch_old = ctx.wildcard_array("owner", 0, nO)
    assert touch_old[:nC].sum() > 100
    own_old = np.asarray(b.arrays["ownerClumpBody"], np.int64).copy()
    pnew, scnew, new_of_old = b.ResortClumps(ctx, 30 * p.h)
    touch_new = ctx.wildcard_array("owner", 0, nO)
    assert np.array_equal(touch_new[new_of_old], touch_old)
    q_new = ctx.wildcard_array("sphere", 0, nS)
    own_new = np.asarray(b.arrays["ownerClumpBody"], np.int64)
    first_old, first_new = np.searchsorted(own_old, np.arange(nC)), np.searchsorted(own_new, np.arange(nC))
    for o in rng.integers(0, nC, 200):
        assert np.array_equal(q_new[first_new[new_of_old[o]]:first_new[new_of_old[o]] + 3], q_sph[first_old[o]:first_old[o] + 3])
    assert np.array_equal(ctx.wildcard_array("analytical", 0, nA), q_wall)
    ctx.step(5)  # the model runs on the permuted arrays
    # UpdateClumps: forty clumps appended; old values keep their (new) places, new entries start from zero
    extra = b.AddClumps(b.templates[0], np.array([[0.03 + 0.02 * (i % 8), 0.03 + 0.02 * (i // 8), 0.13] for i in range(40)], np.float32))  # above the bed, inside the box
    touch_before = ctx.wildcard_array("owner", 0, nO)
    q_before = ctx.wildcard_array("sphere", 0, nS)
    p2, sc2 = b.UpdateClumps(ctx, 35 * p.h)
    nO2, nS2, nC2 = int(sc2.nOwners), int(sc2.nSpheres), int(sc2.nOwnerClumps)
    assert nC2 == nC + 40 and nS2 == nS + 120
    t2, q2 = ctx.wildcard_array("owner", 0, nO2), ctx.wildcard_array("sphere", 0, nS2)
    assert np.array_equal(t2[:nC], touch_before[:nC]) and (t2[nC:nC2] == 0).all() and np.array_equal(t2[nC2:], touch_before[nC:])
    assert np.array_equal(q2[:nS], q_before) and (q2[nS:] == 0).all()
    ctx.step(5)
