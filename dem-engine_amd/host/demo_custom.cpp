// demo_custom.cpp -- a user force model with all three kinds of wildcards, driven through the reference's scripting calls
// (cf. DEMdemo_Electrostatic.cpp / DEMdemo_Fracture*.cpp of the reference, which set geometry and owner wildcards from the
// script and read them back): frictionless Hertz + a pairwise "charge" force from a per-sphere geometry wildcard, a per-owner
// counter of touching contacts, and a contact-age contact wildcard.
//
//   ./demo_custom <outdir>   writes spheres.csv / clumps.csv with the wildcard columns and prints CHECK lines
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "DEMSolver.h"

using namespace deme;

static const char* MODEL = R"MODEL(
{
    const float qq = charge_A[AGeo] * charge_B[BGeo];
    force += demo_charge_force(qq) * B2A;  // from the user's header (AddKernelInclude)
    if (overlapDepth > 0) {
        float E_cnt;
        matProxy2ContactParam<float>(E_cnt, E[bodyAMatType], nu[bodyAMatType], E[bodyBMatType], nu[bodyBMatType]);
        const float sqrt_Rd = sqrt(overlapDepth * (ARadius * BRadius) / (ARadius + BRadius));
        force += (deme::TWO_OVER_THREE * 2.f * E_cnt * sqrt_Rd * (float)overlapDepth) * B2A;
        atomicAdd(n_touch + AOwner, 1.0f);
        atomicAdd(n_touch_B + BOwner, 1.0f);
        contact_age += ts;
    } else {
        contact_age = 0;
    }
}
)MODEL";

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    DEMSolver DEMSim;
    auto mat = DEMSim.LoadMaterial({{"E", 1e8f}, {"nu", 0.3f}, {"CoR", 0.6f}, {"mu", 0.2f}, {"Crr", 0.0f}});
    DEMSim.InstructBoxDomainDimension({0.f, 0.2f}, {0.f, 0.2f}, {0.f, 0.3f});
    DEMSim.InstructBoxDomainBoundingBC("top_open", mat);
    const float r = 0.004f;
    auto tmpl = DEMSim.LoadSphereType(2.6e3f * 4.f / 3.f * 3.14159265f * r * r * r, r, mat);
    std::vector<float3> xyz;
    for (int k = 0; k < 6; k++)
        for (int j = 0; j < 10; j++)
            for (int i = 0; i < 10; i++)
                xyz.push_back(make_float3(0.05f + i * 0.00795f, 0.05f + j * 0.00795f, 0.0045f + k * 0.0083f));  // neighbours in a layer touch
    auto batch = DEMSim.AddClumps(tmpl, xyz);
    std::vector<unsigned int> fam(xyz.size());
    for (size_t i = 0; i < fam.size(); i++)
        fam[i] = (i % 2) ? 1 : 2;
    batch->SetFamilies(fam);
    batch->SetVel(make_float3(0.f, 0.f, -0.3f));

    {   // SetExpandSafetyType (DEM/APIPublic.cpp:836-843): "auto" is the one type there is, anything else is an error
        bool threw = false;
        DEMSim.SetExpandSafetyType("auto");
        try {
            DEMSim.SetExpandSafetyType("manual");
        } catch (const std::runtime_error&) {
            threw = true;
        }
        std::printf("CHECK safety_type_throws %d\n", (int)threw);
    }
    // a header of the user's, found through DEME_KERNEL_INCLUDE_PATH (the test puts demo_helpers.h there)
    DEMSim.AddKernelInclude("demo_helpers.h");
    auto model = DEMSim.DefineContactForceModel(MODEL);
    model->SetPerContactWildcards({"contact_age"});
    model->SetPerOwnerWildcards({"n_touch"});
    model->SetPerGeometryWildcards({"charge"});
    DEMSim.SetInitTimeStep(5e-6);
    DEMSim.SetGravitationalAcceleration(make_float3(0, 0, -9.81f));
    DEMSim.SetCDUpdateFreq(10);
    DEMSim.SetExpandSafetyAdder(0.5f);
    DEMSim.SetMaxVelocity(10.f);
    DEMSim.SetInitBinSizeAsMultipleOfSmallestSphere(4.f);
    DEMSim.Initialize();

    // geometry wildcards from the script: alternating charges on the spheres, none on the walls
    const size_t nS = xyz.size();
    for (size_t i = 0; i < nS; i += 2)
        DEMSim.SetSphereWildcardValue((bodyID_t)i, "charge", std::vector<float>{1.0f, -1.0f});
    DEMSim.SetOwnerWildcardValue(3, "n_touch", 100.f, 2);       // owners 3 and 4 start from 100
    DEMSim.SetFamilyOwnerWildcardValue(2, "n_touch", 1000.f);  // family 2 (even owners) from 1000: overrides owner 4
    const std::vector<float> q = DEMSim.GetSphereWildcardValue(0, "charge", 4);
    std::printf("CHECK charge %.1f %.1f %.1f %.1f\n", q[0], q[1], q[2], q[3]);
    const std::vector<float> t0 = DEMSim.GetOwnerWildcardValue(2, "n_touch", 4);
    std::printf("CHECK n_touch_before %.1f %.1f %.1f %.1f\n", t0[0], t0[1], t0[2], t0[3]);

    DEMSim.DoDynamicsThenSync(40 * 5e-6);  // the layers are still pressed together
    const std::vector<float> all = DEMSim.GetAllOwnerWildcardValue("n_touch");
    const std::vector<float> f1 = DEMSim.GetFamilyOwnerWildcardValue(1, "n_touch");
    const double total = std::accumulate(all.begin(), all.end(), 0.0);
    std::printf("CHECK n_touch_total %.1f family1_members %zu owners %zu contacts %zu\n", total, f1.size(), all.size(), DEMSim.GetNumContacts());

    // contact wildcards from the script: ages of the pairs inside family 1 reset, the others kept
    DEMSim.SetFamilyContactWildcardValueBoth(1, "contact_age", -1.0f);
    DEMSim.SetOutputContent(FAMILY | OWNER_WILDCARD | GEO_WILDCARD);
    DEMSim.WriteSphereFile(dir + "/spheres.csv");
    DEMSim.WriteClumpFile(dir + "/clumps.csv");
    DEMSim.SetContactOutputContent(OWNER | FORCE | CNT_WILDCARD);
    DEMSim.WriteContactFile(dir + "/contacts.csv");
    {   // the same list as vectors (GetContactDetailedInfo): same rows as the file, absent fields throw
        auto info = DEMSim.GetContactDetailedInfo();
        double fsum = 0;
        size_t nSS = 0;
        for (size_t i = 0; i < info->Size(); i++) {
            const float3 f = info->GetForce()[i];
            fsum += std::sqrt((double)f.x * f.x + (double)f.y * f.y + (double)f.z * f.z);
            nSS += info->GetContactType()[i] == "SS" && info->GetAOwner()[i] < info->GetBOwner()[i];
        }
        bool threw = false;
        try {
            info->GetPoint();
        } catch (const std::runtime_error&) {
            threw = true;
        }
        std::printf("CHECK contact_info %zu %zu %.9e %zu %d\n", info->Size(), nSS, fsum, info->GetWildcard("contact_age").size(), (int)threw);
    }
    {   // the same through a tracker: geometry wildcards of the tracked batch, in geometry order
        auto tr = DEMSim.Track(batch);
        std::vector<float> g = tr->GetGeometryWildcardValues("charge");
        tr->SetGeometryWildcardValue("charge", 7.5f, 3);
        std::printf("CHECK tracker_geo %zu %.1f %.1f\n", g.size(), g[3], tr->GetGeometryWildcardValues("charge")[3]);
        tr->SetGeometryWildcardValue("charge", g[3], 3);
    }
    // owner-level getters / setters of the solver object (API.h:515-586, 699-709) and per-family output control
    DEMSim.SetOwnerVelocity(5, make_float3(0.f, 0.f, 1.f));
    DEMSim.SetOwnerFamily(7, 9);
    const size_t moved = DEMSim.ChangeClumpFamily(3, {0.0, 0.066}, {0.0, 1.0}, {0.0, 1.0});  // the three leftmost columns
    const float3 v5 = DEMSim.GetOwnerVelocity(5);
    std::printf("CHECK owner v5z %.3f fam7 %u moved %zu mass0 %.6e oriq0w %.3f\n", v5.z, DEMSim.GetOwnerFamily(7), moved,
                DEMSim.GetOwnerMass(0), DEMSim.GetOwnerOriQ(0).w);
    DEMSim.DisableFamilyOutput(3);
    DEMSim.SetOutputContent(FAMILY);
    DEMSim.WriteSphereFile(dir + "/spheres_no3.csv");
    {   // renumbering in place: owner 3's state, counter and charge travel to its new id; the run carries on
        const float t3 = DEMSim.GetOwnerWildcardValue(3, "n_touch")[0], q3 = DEMSim.GetSphereWildcardValue(3, "charge", 1)[0];
        const float3 p3 = DEMSim.GetOwnerPosition(3);
        const size_t nc = DEMSim.GetNumContacts();
        const std::vector<bodyID_t> map = DEMSim.ResortClumps();
        const float3 p3n = DEMSim.GetOwnerPosition(map[3]);
        std::printf("CHECK resort moved %d same_pos %d same_touch %d same_charge %d\n", (int)(map[3] != 3 || map[4] != 4 || map[5] != 5),
                    (int)(p3.x == p3n.x && p3.y == p3n.y && p3.z == p3n.z),
                    (int)(DEMSim.GetOwnerWildcardValue(map[3], "n_touch")[0] == t3),
                    (int)(DEMSim.GetSphereWildcardValue(map[3], "charge", 1)[0] == q3));  // one sphere per clump: sphere id = owner id
        DEMSim.DoDynamicsThenSync(20 * 5e-6);
        std::printf("CHECK resort_contacts %zu %zu\n", nc, DEMSim.GetNumContacts());
    }
    std::printf("DEMO_OK\n");
    return 0;
}
