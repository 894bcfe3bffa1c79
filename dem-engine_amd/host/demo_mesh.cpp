// demo_mesh.cpp -- clumps dropped onto a triangle mesh loaded from a Wavefront OBJ file (cf. the reference's
// DEMdemo_BallDrop / DEMdemo_FlexibleMesh structure): mesh owner with prescribed motion, node update through a tracker
// (deformable mesh), mesh output as VTK.
//   ./demo_mesh <mesh.obj> <outdir> [steps]
#include <cstdio>
#include <cstdlib>
#include <random>

#include "DEMSolver.h"

using namespace deme;

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: demo_mesh <mesh.obj> <outdir> [steps]\n");
        return 2;
    }
    const int steps = argc > 3 ? std::atoi(argv[3]) : 3000;
    DEMSolver DEMSim;
    auto mat = DEMSim.LoadMaterial({{"E", 1e8f}, {"nu", 0.3f}, {"CoR", 0.5f}, {"mu", 0.3f}, {"Crr", 0.0f}});
    DEMSim.InstructBoxDomainDimension({0.f, 0.2f}, {0.f, 0.2f}, {0.f, 0.3f});
    DEMSim.InstructBoxDomainBoundingBC("top_open", mat);

    auto plate = DEMSim.AddWavefrontMeshObject(argv[1], mat);
    plate->SetInitPos(make_float3(0.1f, 0.1f, 0.05f));
    plate->SetMass(1.f);
    plate->SetMOI(make_float3(1e-2f, 1e-2f, 1e-2f));
    plate->SetFamily(10);
    DEMSim.SetFamilyPrescribedLinVel(10, "0", "0", "0.2f");  // the plate rises at 0.2 m/s
    auto plate_tracker = DEMSim.Track(plate);

    auto ball = DEMSim.LoadSphereType(2.6e3f * 4.f / 3.f * 3.14159265f * 0.004f * 0.004f * 0.004f, 0.004f, mat);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> jit(-0.0003f, 0.0003f);
    std::vector<float3> xyz;
    for (int k = 0; k < 4; k++)
        for (int j = 0; j < 12; j++)
            for (int i = 0; i < 12; i++)
                xyz.push_back(make_float3(0.05f + i * 0.009f + jit(rng), 0.05f + j * 0.009f + jit(rng), 0.06f + k * 0.009f + jit(rng)));
    auto batch = DEMSim.AddClumps(ball, xyz);
    batch->SetVel(make_float3(0.f, 0.f, -0.3f));

    DEMSim.UseFrictionalHertzianModel();
    DEMSim.SetInitTimeStep(5e-6);
    DEMSim.SetGravitationalAcceleration(make_float3(0, 0, -9.81f));
    DEMSim.SetCDUpdateFreq(10);
    DEMSim.SetExpandSafetyAdder(0.3f);
    DEMSim.SetMaxVelocity(10.f);
    DEMSim.SetErrorOutVelocity(100.f);
    DEMSim.SetInitBinSizeAsMultipleOfSmallestSphere(4.f);
    DEMSim.Initialize();
    auto max_z = DEMSim.CreateInspector("clump_max_z");

    DEMSim.DoDynamicsThenSync(steps / 2 * 5e-6);
    // deform the mesh: bend the plate's nodes upwards away from its centre line (owner-local coordinates)
    std::vector<float3> nodes = plate->vertices;
    for (auto& v : nodes)
        v.z += 0.15f * v.x * v.x;
    plate_tracker->UpdateMesh(nodes);
    // pour a few more spheres onto the running simulation (AddClumps + UpdateClumps, as the reference's filling loops do)
    std::vector<float3> more;
    for (int i = 0; i < 25; i++)
        more.push_back(make_float3(0.06f + (i % 5) * 0.02f, 0.06f + (i / 5) * 0.02f, 0.14f));
    auto batch2 = DEMSim.AddClumps(ball, more);
    batch2->SetVel(make_float3(0.f, 0.f, -1.0f));
    DEMSim.UpdateClumps();
    DEMSim.DoDynamicsThenSync((steps - steps / 2) * 5e-6);

    DEMSim.WriteMeshFile(std::string(argv[2]) + "/mesh.vtk");
    DEMSim.SetOutputContent(ABSV | FAMILY);
    DEMSim.WriteSphereFile(std::string(argv[2]) + "/spheres.csv");
    std::printf("MESH triangles=%zu nodes=%zu plate_z=%.6f contacts=%zu max_z=%.5f clumps=%zu\n", plate->GetNumTriangles(),
                plate->GetNumNodes(), plate_tracker->Pos().z, DEMSim.GetNumContacts(), max_z->GetValue(), DEMSim.GetNumClumps());
    DEMSim.ShowThreadCollaborationStats();
    DEMSim.ShowTimingStats();
    std::printf("DEMO_MESH_OK\n");
    return 0;
}
