// demo_collide.cpp -- an API-compatibility client: two spheres sent against each other above a floor, the second one joining after
// Initialize(), written against the include tree a reference script uses (<DEM/API.h>, <DEM/HostSideHelpers.hpp>,
// <DEM/utils/Samplers.hpp>) and exercising the calls a script of the reference's src/demo/ makes around such a scene (materials and
// pair properties, LoadSphereType / Duplicate, AddClumps in both forms, trackers, inspectors, UpdateClumps, SetFamilyClumpMaterial,
// GetOwnerContactForces).  The scene is this test's own: 0.4 m spheres of 650 kg at +-0.5 m, 1.5 and -0.8 m/s, a plane 0.1 m below them,
// the built-in frictional Hertzian model so that the result can be checked against the CPU oracle (tests/test_host_shell.py).
//
//   ./demo_collide [frames]      prints one "STATE <owner> <x> <y> <z> <vx> <vy> <vz> <family>" line per sphere at the end
#include <DEM/API.h>
#include <DEM/HostSideHelpers.hpp>
#include <DEM/utils/Samplers.hpp>
#include <core/ApiVersion.h>
#include <core/utils/ThreadManager.h>

#include <cstdio>
#include <filesystem>

using namespace deme;

int main(int argc, char** argv) {
    const int frames = argc > 1 ? std::atoi(argv[1]) : 25;
    DEMSolver DEMSim;
    DEMSim.SetVerbosity("STEP_METRIC");
    DEMSim.SetOutputFormat(OUTPUT_FORMAT::CSV);
    DEMSim.SetContactOutputContent({"OWNER", "FORCE", "POINT", "NORMAL", "TORQUE", "CNT_WILDCARD"});
    DEMSim.EnsureKernelErrMsgLineNum();

    auto mat_type_1 = DEMSim.LoadMaterial({{"E", 3e8}, {"nu", 0.25}, {"CoR", 0.7}, {"mu", 0.4}, {"Crr", 0.02}});
    auto mat_type_2 = DEMSim.LoadMaterial({{"E", 6e8}, {"nu", 0.35}, {"CoR", 0.5}, {"mu", 0.4}, {"Crr", 0.02}});
    std::shared_ptr<DEMMaterial> mat_type_3 = DEMSim.Duplicate(mat_type_2);
    DEMSim.SetMaterialPropertyPair("CoR", mat_type_1, mat_type_2, 0.55);
    DEMSim.SetMaterialPropertyPair("CoR", mat_type_1, mat_type_3, 0.55);

    DEMSim.InstructBoxDomainDimension({-4.f, 4.f}, {-3.f, 3.f}, {-1.5f, 2.5f});
    auto sph_type_1 = DEMSim.LoadSphereType(650., 0.4, mat_type_1);
    auto sph_type_2 = DEMSim.Duplicate(sph_type_1);

    const float sphPos = 0.5f;
    auto particles1 = DEMSim.AddClumps(sph_type_1, make_float3(-sphPos, 0, 0));  // single-position form
    particles1->SetVel(make_float3(1.5f, 0, 0));
    particles1->SetFamily(0);
    auto tracker1 = DEMSim.Track(particles1);

    auto floor = DEMSim.AddBCPlane(make_float3(0, 0, -0.5f), make_float3(0, 0, 1), mat_type_2);
    (void)floor;

    auto max_z_finder = DEMSim.CreateInspector("clump_max_z");
    auto KE_finder = DEMSim.CreateInspector("clump_kinetic_energy");

    DEMSim.UseFrictionalHertzianModel();
    DEMSim.SetInitTimeStep(1e-5);
    DEMSim.SetGravitationalAcceleration(make_float3(0, 0, -9.8f));
    DEMSim.SetCDUpdateFreq(8);
    DEMSim.SetMaxVelocity(8.);
    DEMSim.SetExpandSafetyType("auto");
    DEMSim.SetExpandSafetyMultiplier(1.1f);
    DEMSim.SetIntegrator("centered_difference");
    auto jitify_options = DEMSim.GetJitifyOptions();
    jitify_options.pop_back();
    DEMSim.Initialize();
    DEMSim.UpdateSimParams();

    // clumps can join after initialization
    auto particles2 = DEMSim.AddClumps(std::vector<std::shared_ptr<DEMClumpTemplate>>(1, sph_type_2),
                                       std::vector<float3>(1, make_float3(sphPos, 0, 0)));
    particles2->SetVel(std::vector<float3>(1, make_float3(-0.8f, 0, 0)));
    particles2->SetFamily(1);
    auto tracker2 = DEMSim.Track(particles2);
    DEMSim.UpdateClumps();
    DEMSim.SetFamilyClumpMaterial(1, mat_type_3);  // the second sphere becomes material 3 before the first step

    const unsigned int sphere1ID = tracker1->GetOwnerID();
    const unsigned int sphere2ID = tracker2->GetOwnerIDs()[0];
    const float frame_time = 8e-3f;
    for (int i = 0; i < frames; i++) {
        DEMSim.DoDynamicsThenSync(frame_time);
        const float max_z = max_z_finder->GetValue(), KE = KE_finder->GetValue();
        std::vector<float3> forces, points;
        DEMSim.GetOwnerContactForces({sphere1ID, sphere2ID}, points, forces);
        if (i % 5 == 4)
            std::printf("frame %d: max z %.6f, kinetic energy %.6f, %zu force pair(s), avg contacts per sphere %.2f, families %u %u\n", i,
                        max_z, KE, points.size(), DEMSim.GetAvgSphContacts(), tracker1->GetFamily(0), tracker2->GetFamilies()[0]);
    }
    DEMSim.ShowMemStats();
    DEMSim.ShowThreadCollaborationStats();
    DEMSim.PrintKinematicScratchSpaceUsage();
    for (unsigned int id : {sphere1ID, sphere2ID}) {
        const float3 x = DEMSim.GetOwnerPosition(id), v = DEMSim.GetOwnerVelocity(id);
        std::printf("STATE %u %.9g %.9g %.9g %.9g %.9g %.9g %u\n", id, x.x, x.y, x.z, v.x, v.y, v.z, DEMSim.GetOwnerFamily(id));
    }
    return 0;
}
