"""Family motion prescriptions (the prescribed-motion switch(family) of integrateVelPos): user strings are
turned into the reference's switch bodies (APIPrivate.cpp:1600-1708), compiled at run time (hipRTC) and applied with the
per-component semantics of DEMIntegrationKernels.cu:100-236; the oracle evaluates the same expressions in a parametric
form (c0 + c1*t + c2*sinf(c3*t))."""
import numpy as np
import pytest


def test_codegen_matches_reference_form(pkg):
    b = pkg.model.SceneBuilder()
    b.SetFamilyPrescribedLinVel(3, "0.1f", "none", "0.2f*t")
    b.SetFamilyPrescribedAngVel(3, "0", "0", "3.14", dictate=False)  # merged into family 3: strings replace, flags OR
    b.AddFamilyPrescribedAcc(5, "none", "none", "-1.0f")
    b.SetFamilyPrescribedQuaternion(7, "return make_float4(0, 0, sinf(t), cosf(t));")
    vel, pos, acc = b.prescription_cases()
    assert vel.startswith(" case 3: {{vX = 0.1f;vZ = 0.2f*t;}{omgBarX = 0;omgBarY = 0;omgBarZ = 3.14;}LinVelXPrescribed = 1;"
                          "LinVelYPrescribed = 1;LinVelZPrescribed = 1;RotVelXPrescribed = 1;RotVelYPrescribed = 1;"
                          "RotVelZPrescribed = 1;break; }case 5: {{}{}LinVelXPrescribed = 0;")
    assert "case 5: {{accZ = -1.0f;}{}break; }" in acc
    assert ("case 7: {{}{float4 DEME_Presc_OriQ =  make_float4(0, 0, sinf(t), cosf(t));;oriQw = DEME_Presc_OriQ.w; "
            "oriQx = DEME_Presc_OriQ.x; oriQy = DEME_Presc_OriQ.y; oriQz = DEME_Presc_OriQ.z;}LinXPrescribed = 1;"
            "LinYPrescribed = 1;LinZPrescribed = 1;RotPrescribed = 1;break; }") in pos
    assert b.family_flags[3] & pkg.abi.FAMILY_PRESCRIBED and b.family_flags[5] & pkg.abi.FAMILY_PRESCRIBED
    with pytest.raises(ValueError):
        b.SetFamilyPrescribedLinVel(300, "0", "0", "0")


def _scene(pkg):
    """a settling bed on a piston: the floor is an analytical plane of family 10 that rises with a prescribed, accelerating
    velocity; every third clump (family 1) feels an extra, growing vertical acceleration; family 2 is dragged sideways at
    a prescribed x-velocity but still takes contact forces in y and z"""
    b = pkg.model.packed_bed(900, seed=31, cd_freq=0, spacing_mult=2.4, init_vz=-0.3, aspect=(1.0, 1.0, 0.5))
    fam = np.zeros(len(b.batches[0].xyz), np.uint8)
    fam[::3] = 1
    fam[1::7] = 2
    b.batches[0].SetFamily(fam)
    mat = 0
    piston = b.AddExternalObject()
    piston.AddPlane((0.0, 0.0, 0.0162), (0, 0, 1), mat)  # at the lowest spheres of the lattice
    piston.family = 10
    b.SetFamilyPrescribedLinVel(10, "0", "0", "0.02f + 3.0f*t")
    b.AddFamilyPrescribedAcc(1, "none", "none", "-4.0f + 16000.0f*t")
    b.SetFamilyPrescribedLinVel(2, "0.004f", "none", "none", dictate=False)
    p, sc = b.Initialize()
    return b, p, sc


def _oracle_prescriptions(sim):
    c = np.zeros((15, 4), np.float32)
    c[2] = (0.02, 3.0, 0, 0)
    sim.set_prescription(10, has=0b111, flags=0b111111, coef=c)  # v assigned, all six velocity components dictated
    c = np.zeros((15, 4), np.float32)
    c[11] = (-4.0, 16000.0, 0, 0)
    sim.set_prescription(1, has=1 << 11, flags=0, coef=c)
    c = np.zeros((15, 4), np.float32)
    c[0] = (0.004, 0, 0, 0)
    sim.set_prescription(2, has=0b1, flags=0b1, coef=c)  # only vX assigned and dictated


def test_oracle_piston_pushes_the_bed(pkg, orc):
    b, p, sc = _scene(pkg)
    sim = orc.make_sim(pkg, p, sc)
    _oracle_prescriptions(sim)
    n = int(sc.nOwnerClumps)
    sim.step(200)
    st = sim.download_state()
    t = 200 * p.h
    assert abs(st["vZ"][n] - (0.02 + 3.0 * (t - p.h))) < 1e-6 and st["vX"][n] == 0  # the piston follows its prescription
    Z = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:, 2]
    assert Z[n] + p.LBFZ > 1e-5  # the piston owner moved up from its initial position (0)
    fam = b.arrays["familyID"][:n]
    assert np.all(st["vX"][:n][fam == 2] == np.float32(0.004))  # dictated component
    assert np.abs(st["vY"][:n][fam == 2]).max() > 0  # the others still feel contacts


@pytest.mark.gpu
def test_gpu_prescriptions_match_oracle(pkg, orc):
    b, p, sc = _scene(pkg)
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    b.compile_into(ctx)
    sim = orc.make_sim(pkg, p, sc)
    _oracle_prescriptions(sim)
    ctx.step(200), sim.step(200)
    gs, os_ = ctx.download_state(), sim.download_state()
    n = int(sc.nOwnerClumps)
    assert gs["vZ"][n] == os_["vZ"][n] and gs["vZ"][n] > 0.0225  # piston: identical fp32 expression
    fam = b.arrays["familyID"][:n]
    assert np.all(gs["vX"][:n][fam == 2] == np.float32(0.004))
    X = pkg.model.decode_positions(gs["voxelID"], gs["locX"], gs["locY"], gs["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    Y = pkg.model.decode_positions(os_["voxelID"], os_["locX"], os_["locY"], os_["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    assert int(ctx.counts().nContacts) == int(sim.counts().nContacts) > 300
    # the prescription expressions are the same fp32 arithmetic on both sides: the usual trajectory tolerances hold
    assert np.abs(X - Y).max() == 0.0  # bit-identical trajectories
    V = np.stack([gs["vX"], gs["vY"], gs["vZ"]], 1)
    W = np.stack([os_["vX"], os_["vY"], os_["vZ"]], 1)
    assert np.abs(V - W).max() == 0.0
    # clearing the prescriptions restores the plain integrator for those families
    ctx.compile_prescriptions("", "", "")
    ctx.step(5)
    g2 = ctx.download_state()
    assert g2["vZ"][n] != gs["vZ"][n]  # the piston now accelerates under gravity / contacts like any free body


def _rule_scene(pkg):
    """clumps that rise above z = 0.0345 m are frozen (family 5 is fixed); clumps that fall faster than 0.6 m/s are tagged
    family 6, which gets an extra upward acceleration"""
    b = pkg.model.packed_bed(900, seed=33, cd_freq=0, spacing_mult=2.4, init_vz=-0.3, aspect=(1.0, 1.0, 0.5))
    b.ChangeFamilyWhen(0, 5, "return Z > 0.0345;")
    b.ChangeFamilyWhen(0, 6, "return vZ < -0.6;")
    b.SetFamilyFixed(5)
    b.AddFamilyPrescribedAcc(6, "none", "none", "40.0f")
    p, sc = b.Initialize()
    return b, p, sc


def _oracle_rules(sim):
    sim.add_family_rule(0, 5, 2, 0, 0.0345)  # quantity 2 = Z, op 0 = ">"
    sim.add_family_rule(0, 6, 5, 1, -0.6)    # quantity 5 = vZ, op 1 = "<"
    c = np.zeros((15, 4), np.float32)
    c[11] = (40.0, 0, 0, 0)
    sim.set_prescription(6, has=1 << 11, flags=0, coef=c)


def test_family_rules_codegen_and_oracle(pkg, orc):
    b, p, sc = _rule_scene(pkg)
    assert b.family_change_rules() == (" if (family_code == 0) { bool shouldMakeChange = false;shouldMakeChange =  Z > 0.0345;"
                                       "if (shouldMakeChange) {granData->familyID[myOwner] = 5;}}if (family_code == 0) "
                                       "{ bool shouldMakeChange = false;shouldMakeChange =  vZ < -0.6;if (shouldMakeChange) "
                                       "{granData->familyID[myOwner] = 6;}}")
    sim = orc.make_sim(pkg, p, sc)
    _oracle_rules(sim)
    sim.step(150)
    st = sim.download_state()
    n = int(sc.nOwnerClumps)
    fam = st["familyID"][:n]
    assert (fam == 5).sum() > 3 and (fam == 6).sum() > 3
    assert np.all(st["vZ"][:n][fam == 5] == 0)  # frozen where they were caught


@pytest.mark.gpu
def test_gpu_family_changes_match_oracle(pkg, orc):
    b, p, sc = _rule_scene(pkg)
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    b.compile_into(ctx)
    sim = orc.make_sim(pkg, p, sc)
    _oracle_rules(sim)
    ctx.step(150), sim.step(150)
    gs, os_ = ctx.download_state(), sim.download_state()
    n = int(sc.nOwnerClumps)
    assert np.array_equal(gs["familyID"], os_["familyID"])
    assert (gs["familyID"][:n] == 5).sum() > 3 and (gs["familyID"][:n] == 6).sum() > 3
    X = pkg.model.decode_positions(gs["voxelID"], gs["locX"], gs["locY"], gs["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    Y = pkg.model.decode_positions(os_["voxelID"], os_["locX"], os_["locY"], os_["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    assert np.abs(X - Y).max() == 0.0  # bit-identical trajectories
    assert np.all(gs["vZ"][:n][gs["familyID"][:n] == 5] == 0)
    # the staged calls (margins / detect / migrate / forces / integrate) apply the rules in the same place as deme_step
    staged = pkg.Context(0)
    staged.set_params(p), staged.upload_scene(sc)
    b.compile_into(staged)
    for _ in range(150):
        staged.compute_margins(int(p.cdUpdateFreq)), staged.detect(), staged.migrate(), staged.calc_forces(), staged.integrate()
    ss = staged.download_state()
    for k in ("familyID", "voxelID", "locX", "locY", "locZ", "vX", "vY", "vZ"):
        assert np.array_equal(ss[k], gs[k]), k
    ctx.change_family(6, 0), sim.change_family(6, 0)
    assert np.array_equal(ctx.download_state()["familyID"], sim.download_state()["familyID"])
    # a rule that reads the contact acceleration makes the step reduce a/alpha before the rules run
    ctx.compile_family_rules(" if (family_code == 0) { bool shouldMakeChange = false;shouldMakeChange =  accZ > 1e9;"
                             "if (shouldMakeChange) {granData->familyID[myOwner] = 7;}}")
    ctx.step(3)
    assert not (ctx.download_state()["familyID"] == 7).any()
