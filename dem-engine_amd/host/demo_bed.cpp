// demo_bed.cpp -- a bed of two clump kinds built through the C++ shell from positions the caller supplies, so that the scene the
// shell hands the engine can be compared, field by field, with the one the Python set-up path (model.py) builds from the same
// inputs (tests/test_host_shell.py: the two hosts implement the same sizing / flattening / table logic independently).
//
//   DEME_DUMP_SCENE=<file> ./demo_bed <clumps.f32> <n> <steps> <outdir> [mesh.obj]
//     clumps.f32: n records of 8 floats (x y z, quaternion x y z w, kind); writes <outdir>/clumps.csv after `steps` steps;
//     with a mesh file the scene also holds that mesh (scaled, turned, fixed), a cylindrical wall, and sizes its bins by number
#include <DEM/API.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace deme;

int main(int argc, char** argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: demo_bed clumps.f32 n steps outdir\n");
        return 2;
    }
    const size_t n = (size_t)std::atol(argv[2]);
    const int steps = std::atoi(argv[3]);
    const std::string dir = argv[4];
    std::vector<float> raw(n * 8);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(raw.data(), sizeof(float), raw.size(), f) != raw.size()) {
        std::fprintf(stderr, "cannot read %zu records from %s\n", n, argv[1]);
        return 2;
    }
    std::fclose(f);

    DEMSolver DEMSim;
    DEMSim.SetVerbosity("ERROR");
    auto mat_grain = DEMSim.LoadMaterial({{"E", 1e8f}, {"nu", 0.3f}, {"CoR", 0.6f}, {"mu", 0.2f}, {"Crr", 0.0f}});
    auto mat_wall = DEMSim.LoadMaterial({{"E", 2e8f}, {"nu", 0.25f}, {"CoR", 0.5f}, {"mu", 0.4f}, {"Crr", 0.0f}});
    DEMSim.SetMaterialPropertyPair("mu", mat_grain, mat_wall, 0.35f);

    DEMSim.InstructBoxDomainDimension({0.f, 0.3f}, {0.f, 0.2f}, {0.f, 0.25f});
    DEMSim.InstructBoxDomainBoundingBC("top_open", mat_wall);

    const float r = 0.005f;
    // loaded first but sorted behind the single sphere (templates go by component count)
    auto clump3 = DEMSim.LoadClumpType(2.6e3f * 5.5886717f, make_float3(2.928f, 2.6029f, 3.9908f) * 2.6e3f, std::vector<float>{0.8f, 0.8f, 0.8f},
                                       std::vector<float3>{{0.5f, 0.341729f, 0.f}, {0.f, -0.658271f, 0.f}, {-0.5f, 0.341729f, 0.f}}, mat_grain);
    clump3->Scale(r);
    auto single = DEMSim.LoadSphereType(2.6e3f * 4.f / 3.f * 3.14159265f * 0.006f * 0.006f * 0.006f, 0.006f, mat_grain);

    std::vector<std::shared_ptr<DEMClumpTemplate>> kinds;
    std::vector<float3> xyz;
    std::vector<float4> q;
    std::vector<unsigned int> fam;
    for (size_t i = 0; i < n; i++) {
        const float* c = raw.data() + 8 * i;
        xyz.push_back(make_float3(c[0], c[1], c[2]));
        q.push_back(make_float4(c[3], c[4], c[5], c[6]));
        kinds.push_back(c[7] > 0.5f ? single : clump3);
        fam.push_back(i % 3 == 0 ? 1u : 2u);
    }
    auto batch = DEMSim.AddClumps(kinds, xyz);
    batch->SetOriQ(q);
    batch->SetVel(make_float3(0.f, 0.f, -0.4f));
    batch->SetFamilies(fam);

    // an inclined deflector plate that creeps sideways (a prescribed analytical owner), and a family that keeps its distance
    auto plate = DEMSim.AddExternalObject();
    plate->AddPlane(make_float3(0.f, 0.f, 0.f), make_float3(0.6f, 0.f, 0.8f), mat_wall);
    plate->SetInitPos(make_float3(0.02f, 0.1f, 0.01f));
    plate->SetFamily(10);
    DEMSim.SetFamilyPrescribedLinVel(10, "0.01", "0", "0");
    DEMSim.DisableContactBetweenFamilies(1, 10);
    DEMSim.SetFamilyExtraMargin(2, 2e-4f);

    const bool with_mesh = argc > 5;
    if (with_mesh) {
        auto ball = DEMSim.AddWavefrontMeshObject(argv[5], mat_wall);
        ball->Scale(0.02f);
        ball->SetInitPos(make_float3(0.15f, 0.1f, 0.03f));
        ball->SetInitQuat(make_float4(0.f, 0.38268343f, 0.f, 0.92387953f));  // 45 degrees about y
        ball->SetMass(0.2f);
        ball->SetMOI(make_float3(3e-5f, 3e-5f, 3e-5f));
        ball->SetFamily(11);
        DEMSim.SetFamilyFixed(11);
        auto drum = DEMSim.AddExternalObject();
        drum->AddCylinder(make_float3(0.15f, 0.1f, 0.f), make_float3(0.f, 0.f, 1.f), 0.16f, mat_wall, ENTITY_NORMAL_INWARD);
        drum->SetFamily(12);
        DEMSim.SetFamilyFixed(12);
    }

    DEMSim.UseFrictionalHertzianModel();
    DEMSim.SetInitTimeStep(5e-6);
    DEMSim.SetGravitationalAcceleration(make_float3(0, 0, -9.81f));
    DEMSim.SetCDUpdateFreq(10);
    DEMSim.SetExpandSafetyMultiplier(1.1f);
    DEMSim.SetExpandSafetyAdder(0.05f);
    DEMSim.SetMaxVelocity(8.f);
    DEMSim.SetErrorOutVelocity(200.f);
    if (with_mesh)
        DEMSim.SetInitBinNumTarget(200000);
    else
        DEMSim.SetInitBinSizeAsMultipleOfSmallestSphere(3.5f);
    DEMSim.SetIntegrator("extended_taylor");
    DEMSim.Initialize();

    DEMSim.DoDynamicsThenSync(steps * 5e-6);
    DEMSim.SetOutputContent({"XYZ", "QUAT", "VEL", "ANG_VEL", "FAMILY"});
    DEMSim.WriteClumpFile(dir + "/clumps.csv", 10);
    std::printf("DEMO_OK clumps=%zu contacts=%zu\n", DEMSim.GetNumClumps(), DEMSim.GetNumContacts());
    return 0;
}
