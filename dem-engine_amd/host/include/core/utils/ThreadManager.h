// core/utils/ThreadManager.h -- the reference's kT/dT hand-over object (src/core/utils/ThreadManager.h).  This build has no
// worker threads (detection and stepping are phases on one HIP stream), so there is nothing to manage; the header exists
// because the reference demos include it.
#pragma once
