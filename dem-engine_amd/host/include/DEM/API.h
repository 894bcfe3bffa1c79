// DEM/API.h -- include path of the reference's public header (src/DEM/API.h), so that a script written against the reference
// (#include <DEM/API.h>) builds against this shell unchanged.  Everything lives in ../../DEMSolver.h.
#pragma once
#include "../../DEMSolver.h"
