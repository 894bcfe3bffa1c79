// deme_tile_p.hip -- the translation unit of the persistent owner-tile force kernels (deme_tile_p.h), compiled apart from deme_hip.hip
// because it wants one compiler switch of its own: -mllvm -amdgpu-atomic-optimizer-strategy=None.  The atomic optimiser turns
// "thread 0 takes the next tile number from the counter" into a wave-wide reduction whose result it reads back with
// v_readfirstlane at once -- i.e. the wavefront would wait for the answer of the atomic, a round trip to memory, at the START of
// every tile instead of picking it up a tile later.  (The detection kernels of deme_hip.hip keep the optimiser: their counters are
// bumped by whole wavefronts.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <algorithm>

#define DEME_TILE_FORCE_ONLY 1
#include "../../include/deme_hip.h"
#include "deme_device.h"
#include "deme_force.h"
#include "deme_force_fast.h"
#include "deme_tile.h"
#include "deme_tile_p.h"

namespace deme_dev {

// which = model * 4 + (mesh ? 2 : 0) + (record ? 1 : 0), as launch_forces counts the instances of k_tile_forces.
// Launches min(nTiles, workgroups the device holds at once) persistent workgroups; returns a hipError_t.
int launch_tile_forces_p(int which, unsigned nCU, unsigned ldsBytes, hipStream_t st, const DevParams& dp, const TileArgs& ta) {
    auto go = [&](auto kern) -> int {
        static int perCU = 0;               // (per instance; the LDS size of a context changes rarely, and only ever the same way for
        static unsigned perCUlds = ~0u;     // all contexts of a process -- asked again whenever it does)
        if (perCUlds != ldsBytes) {
            int n = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, DEME_TILE_T, ldsBytes);
            if (e != hipSuccess)
                return (int)e;
            perCU = std::max(n, 1), perCUlds = ldsBytes;
        }
        unsigned grid = std::min<unsigned>(ta.nTiles, (unsigned)perCU * nCU);
        TileArgs tb = ta;
        tb.ctrParts = 1u;  // one counter per DEME_TILE_P_PARTS-th of the workgroups when the chip is full (see the kernel)
        if (grid >= 8u * DEME_TILE_P_PARTS)
            grid -= grid % DEME_TILE_P_PARTS, tb.ctrParts = DEME_TILE_P_PARTS;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(DEME_TILE_T), ldsBytes, st, dp, tb);
        return (int)hipGetLastError();
    };
    switch (which) {
        case 0: return go(k_tile_forces_p<0, false, false>);
        case 1: return go(k_tile_forces_p<0, false, true>);
        case 2: return go(k_tile_forces_p<0, true, false>);
        case 3: return go(k_tile_forces_p<0, true, true>);
        case 4: return go(k_tile_forces_p<1, false, false>);
        case 5: return go(k_tile_forces_p<1, false, true>);
        case 6: return go(k_tile_forces_p<1, true, false>);
        default: return go(k_tile_forces_p<1, true, true>);
    }
}

}  // namespace deme_dev
