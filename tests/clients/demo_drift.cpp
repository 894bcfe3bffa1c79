// demo_drift.cpp -- a test client of the C++ shell (tests/test_host_shell.py): a layer of spheres drifting along x over a floor, far
// enough to change slabs on a decomposed run; in the middle of the run more spheres are added (AddClumps + UpdateClumps) and the
// clumps are renumbered in place (ResortClumps) -- both re-upload the scene, and a decomposed run has to be cut by where the clumps
// ARE by then, not by where their batches were loaded.  The program knows nothing of slabs: DEME_SLABS_PER_DEVICE decides.
//
//   ./demo_drift [steps per leg]     prints "POS <load-order id> x y z" for every 37th clump and a checksum line
#include <DEM/API.h>
#include <DEM/HostSideHelpers.hpp>
#include <DEM/utils/Samplers.hpp>

#include <cmath>
#include <cstdio>

using namespace deme;

int main(int argc, char** argv) {
    const int leg = argc > 1 ? std::atoi(argv[1]) : 1500;
    DEMSolver DEMSim;
    DEMSim.SetVerbosity("ERROR");
    auto mat = DEMSim.LoadMaterial({{"E", 1e8}, {"nu", 0.3}, {"CoR", 0.5}, {"mu", 0.4}, {"Crr", 0.0}});
    DEMSim.InstructBoxDomainDimension({0.f, 1.6f}, {0.f, 0.4f}, {0.f, 0.4f});
    DEMSim.InstructBoxDomainBoundingBC("all", mat);
    auto ball = DEMSim.LoadSphereType(2.6e3 * 4.0 / 3.0 * 3.14159265 * 0.004 * 0.004 * 0.004, 0.004, mat);
    std::vector<float3> xyz;
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 30; j++)
            for (int i = 0; i < 60; i++)  // three layers, slightly staggered: they land on the floor and on each other while drifting
                xyz.push_back(make_float3(0.10f + 0.0105f * i + 0.002f * (k % 2), 0.04f + 0.0105f * j + 0.002f * (k % 2), 0.006f + 0.0095f * k));
    auto batch = DEMSim.AddClumps(ball, xyz);
    batch->SetVel(make_float3(2.0f, 0.f, 0.f));
    DEMSim.UseFrictionalHertzianModel();
    DEMSim.SetInitTimeStep(5e-6);
    DEMSim.SetGravitationalAcceleration(make_float3(0, 0, -9.81f));
    DEMSim.SetCDUpdateFreq(10);
    DEMSim.SetExpandSafetyAdder(1.0f);
    DEMSim.SetSlabMigrationInterval(100);
    DEMSim.SetSlabHalo(0.03f);
    DEMSim.Initialize();
    std::printf("SLABS %u\n", DEMSim.GetNumSlabs());

    DEMSim.DoDynamicsThenSync(leg * 5e-6);  // 1.5 cm of drift per 1500 steps at 2 m/s, minus what friction takes
    // more spheres join where the bed has gone, at its speed
    std::vector<float3> more;
    for (int i = 0; i < 40; i++)
        more.push_back(make_float3(0.15f + 0.012f * i, 0.2f, 0.06f));
    auto batch2 = DEMSim.AddClumps(ball, more);
    batch2->SetVel(make_float3(2.0f, 0.f, -0.5f));
    DEMSim.UpdateClumps();
    DEMSim.DoDynamicsThenSync(leg * 5e-6);
    const std::vector<bodyID_t> map = DEMSim.ResortClumps();  // ids change; the run does not
    DEMSim.DoDynamicsThenSync(leg * 5e-6);

    const size_t n = DEMSim.GetNumClumps();
    std::vector<size_t> old_of_new(n);
    for (size_t o = 0; o < n; o++)
        old_of_new[map[o]] = o;
    double sx = 0, sy = 0, sz = 0, sv = 0;
    for (size_t id = 0; id < n; id++) {
        const float3 x = DEMSim.GetOwnerPosition((bodyID_t)id), v = DEMSim.GetOwnerVelocity((bodyID_t)id);
        sx += x.x, sy += x.y, sz += x.z, sv += std::sqrt((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z);
        if (old_of_new[id] % 37 == 0)
            std::printf("POS %zu %.7f %.7f %.7f\n", old_of_new[id], x.x, x.y, x.z);
    }
    std::printf("SUM clumps %zu contacts %zu mean_x %.7f mean_y %.7f mean_z %.7f mean_speed %.6f\n", n, DEMSim.GetNumContacts(), sx / n, sy / n,
                sz / n, sv / n);
    std::printf("DRIFT_OK\n");
    return 0;
}
