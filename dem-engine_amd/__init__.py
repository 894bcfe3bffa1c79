"""dem-engine_amd -- MI355X-native hot path for Chrono DEM-Engine workloads.

This package holds only what the hot path needs:

  csrc/      hand-written HIP (gfx950) kernels, the C++ host orchestration and the
             C-ABI (include/deme_hip.h) -> libdeme_hip.so
  host/      C++ ``deme::DEMSolver`` shell above the C-ABI (drop-in for demo scripts)
  abi.py     ctypes binding of the C-ABI (what a maintainer's FFI stub would look like)
  decomp.py  slab decomposition + ghost lists for multi-GPU runs
  io.py      the reference's CSV writers / readers (sphere, clump, contact files; restart data)
  model.py   host-side model builder: the subset of ``DEMSolver`` set-up calls
             (API.h:50-1300) that defines kernel inputs, restated in numpy

There is NO CPU fallback: if libdeme_hip.so is missing, importing ``abi`` raises.
The directory name contains a hyphen (the project name); load it with
``__graft_entry__.load_package()`` which registers it as ``dem_engine_amd``.
"""
from . import abi, model, decomp, io  # noqa: F401
from .abi import Context, DemeParams, DemeScene, DemeOwnerState, DemeCounts, library_path  # noqa: F401
from .model import SceneBuilder  # noqa: F401

__all__ = ["abi", "model", "Context", "SceneBuilder", "DemeParams", "DemeScene", "DemeOwnerState", "DemeCounts"]
