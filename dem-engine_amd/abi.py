"""ctypes binding of include/deme_hip.h (libdeme_hip.so).

Fails loudly when the HIP library has not been built: there is no CPU path in
the product.  Structure layouts mirror the header field for field.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

NULL_MAPPING_PARTNER = 0xFFFFFFFF
FAMILY_MASK_ENTRIES = 32896
NUM_FAMILIES = 256
FAMILY_FIXED = 1
FAMILY_GHOST = 2
FAMILY_PRESCRIBED = 4
INTEGRATOR_FORWARD_EULER, INTEGRATOR_CENTERED_DIFFERENCE, INTEGRATOR_EXTENDED_TAYLOR = 0, 1, 2
FORCE_HERTZIAN, FORCE_HERTZIAN_FRICTIONLESS, FORCE_CUSTOM = 0, 1, 2
GHOST_BYTES = 56


ARITH_FAST, ARITH_EXACT = 1, 0


class DemeParams(C.Structure):
    _fields_ = [
        ("nvXp2", C.c_uint32), ("nvYp2", C.c_uint32), ("nvZp2", C.c_uint32),
        ("nbX", C.c_uint32), ("nbY", C.c_uint32), ("nbZ", C.c_uint32),
        ("l", C.c_double), ("voxelSize", C.c_double), ("binSize", C.c_double),
        ("LBFX", C.c_float), ("LBFY", C.c_float), ("LBFZ", C.c_float),
        ("Gx", C.c_float), ("Gy", C.c_float), ("Gz", C.c_float),
        ("h", C.c_float), ("beta", C.c_float), ("approxMaxVel", C.c_float),
        ("expSafetyMulti", C.c_float), ("expSafetyAdder", C.c_float),
        ("integrator", C.c_uint32), ("forceModel", C.c_uint32), ("nContactWildcards", C.c_uint32),
        ("cdUpdateFreq", C.c_uint32), ("errOutBinSphNum", C.c_uint32), ("errOutVel", C.c_float),
        ("timeElapsed", C.c_double),
    ]


_P = C.c_void_p


class DemeScene(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("nOwners", "nOwnerClumps", "nSpheres", "nAnal", "nTri", "nMat", "nComp", "nMassProps")] + \
               [(n, _P) for n in (
                   "voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz",
                   "vX", "vY", "vZ", "omgBarX", "omgBarY", "omgBarZ", "familyID", "inertiaPropOffsets",
                   "ownerClumpBody", "clumpComponentOffset", "sphereMaterialOffset",
                   "Radii", "CDRelPosX", "CDRelPosY", "CDRelPosZ", "MassProperties", "moiX", "moiY", "moiZ",
                   "objType", "objOwner", "objNormal", "objMaterial", "objRelPosX", "objRelPosY", "objRelPosZ",
                   "objRotX", "objRotY", "objRotZ", "objSize1", "objSize2", "objSize3", "objMass",
                   "E", "nu", "CoR", "mu", "Crr",
                   "familyMasks", "familyExtraMarginSize", "familyFlags",
                   "ownerMesh", "triNode1", "triNode2", "triNode3", "triMaterialOffset", "ownerGhost")]


# field name -> numpy dtype, for building DemeScene / DemeOwnerState from arrays
SCENE_DTYPES = {
    "voxelID": np.uint64, "locX": np.uint16, "locY": np.uint16, "locZ": np.uint16,
    "oriQw": np.float32, "oriQx": np.float32, "oriQy": np.float32, "oriQz": np.float32,
    "vX": np.float32, "vY": np.float32, "vZ": np.float32,
    "omgBarX": np.float32, "omgBarY": np.float32, "omgBarZ": np.float32,
    "familyID": np.uint8, "inertiaPropOffsets": np.uint16,
    "ownerClumpBody": np.uint32, "clumpComponentOffset": np.uint16, "sphereMaterialOffset": np.uint16,
    "Radii": np.float32, "CDRelPosX": np.float32, "CDRelPosY": np.float32, "CDRelPosZ": np.float32,
    "MassProperties": np.float32, "moiX": np.float32, "moiY": np.float32, "moiZ": np.float32,
    "objType": np.uint8, "objOwner": np.uint32, "objNormal": np.float32, "objMaterial": np.uint16,
    "objRelPosX": np.float32, "objRelPosY": np.float32, "objRelPosZ": np.float32,
    "objRotX": np.float32, "objRotY": np.float32, "objRotZ": np.float32,
    "objSize1": np.float32, "objSize2": np.float32, "objSize3": np.float32, "objMass": np.float32,
    "E": np.float32, "nu": np.float32, "CoR": np.float32, "mu": np.float32, "Crr": np.float32,
    "familyMasks": np.uint8, "familyExtraMarginSize": np.float32, "familyFlags": np.uint8,
    "ownerMesh": np.uint32, "triNode1": np.float32, "triNode2": np.float32, "triNode3": np.float32,
    "triMaterialOffset": np.uint16,
    "ownerGhost": np.uint8,
}

STATE_DTYPES = {
    "voxelID": np.uint64, "locX": np.uint16, "locY": np.uint16, "locZ": np.uint16,
    "oriQw": np.float32, "oriQx": np.float32, "oriQy": np.float32, "oriQz": np.float32,
    "vX": np.float32, "vY": np.float32, "vZ": np.float32,
    "omgBarX": np.float32, "omgBarY": np.float32, "omgBarZ": np.float32,
    "aX": np.float32, "aY": np.float32, "aZ": np.float32,
    "alphaX": np.float32, "alphaY": np.float32, "alphaZ": np.float32,
    "familyID": np.uint8,
}


class DemeOwnerState(C.Structure):
    _fields_ = [(n, _P) for n in STATE_DTYPES]


class DemeCounts(C.Structure):
    _fields_ = [("nContacts", C.c_uint64), ("nPrevContacts", C.c_uint64), ("nBinSphereTouches", C.c_uint64),
                ("nActiveBins", C.c_uint64), ("nSteps", C.c_uint64), ("nDetections", C.c_uint64),
                ("maxSpheresInBin", C.c_uint32), ("lastStatus", C.c_uint32)]


class DemeAdaptive(C.Structure):
    """include/deme_hip.h DemeAdaptive; the defaults are the reference's (DEM/Structs.h:204-216)"""
    _fields_ = [("autoBinSize", C.c_uint32), ("binObserveSteps", C.c_uint32), ("binMaxRate", C.c_float), ("binAcc", C.c_float),
                ("binUpperSafety", C.c_float), ("binLowerSafety", C.c_float), ("autoUpdateFreq", C.c_uint32),
                ("maxUpdateFreq", C.c_uint32), ("freqObserveDetections", C.c_uint32)]


def make_scene_struct(arrays, counts):
    """arrays: dict name -> numpy array (kept alive by the caller); counts: dict of n* fields."""
    sc = DemeScene()
    keep = {}
    for k, v in counts.items():
        setattr(sc, k, int(v))
    for name, dt in SCENE_DTYPES.items():
        a = arrays.get(name)
        if a is None:
            setattr(sc, name, None)
            continue
        a = np.ascontiguousarray(a, dtype=dt)
        keep[name] = a
        setattr(sc, name, a.ctypes.data if a.size else None)
    sc._keep = keep
    return sc


def make_state_struct(n_owners, arrays=None):
    """Allocate (or wrap) per-owner arrays for a state round trip."""
    st = DemeOwnerState()
    out = {}
    for name, dt in STATE_DTYPES.items():
        if arrays is not None and name in arrays and arrays[name] is not None:
            a = np.ascontiguousarray(arrays[name], dtype=dt)
        elif arrays is not None:
            setattr(st, name, None)
            continue
        else:
            a = np.zeros(n_owners, dtype=dt)
        out[name] = a
        setattr(st, name, a.ctypes.data)
    st._keep = out
    return st, out


def library_path():
    # DEME_HIP_LIB: another build of the same library (kernel A/B experiments); never a different implementation
    return os.environ.get("DEME_HIP_LIB") or os.path.join(_HERE, "csrc", "libdeme_hip.so")


_lib = None


class DemeError(RuntimeError):
    pass


def load_library():
    """Load libdeme_hip.so.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise DemeError(
            f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback in this package.")
    lib = C.CDLL(path)
    lib.deme_last_error.restype = C.c_char_p
    lib.deme_last_error.argtypes = [_P]
    lib.deme_version.restype = C.c_char_p
    lib.deme_ctx_create.argtypes = [C.c_int, C.POINTER(_P)]
    lib.deme_ctx_destroy.argtypes = [_P]
    lib.deme_ctx_destroy.restype = None
    for name, args in {
        "deme_ctx_set_stream": [_P, _P], "deme_sync": [_P], "deme_set_arith_mode": [_P, C.c_int], "deme_set_async_detection": [_P, C.c_uint32], "deme_get_arith_mode": [_P],
        "deme_set_params": [_P, C.POINTER(DemeParams)], "deme_upload_scene": [_P, C.POINTER(DemeScene)],
        "deme_upload_owner_state": [_P, C.POINTER(DemeOwnerState)],
        "deme_download_owner_state": [_P, C.POINTER(DemeOwnerState)],
        "deme_update_tri_nodes": [_P, _P, _P, _P],
        "deme_compute_margins": [_P, C.c_uint32], "deme_set_margins": [_P, _P],
        "deme_detect_contacts": [_P], "deme_migrate_history": [_P], "deme_calc_forces": [_P],
        "deme_integrate": [_P], "deme_step": [_P, C.c_uint32],
        "deme_get_counts": [_P, C.POINTER(DemeCounts)],
        "deme_download_bin_incidence": [_P, _P, _P, C.c_size_t],
        "deme_download_contacts": [_P, _P, _P, _P, _P, C.c_size_t],
        "deme_download_contact_wildcard": [_P, C.c_uint32, _P, C.c_size_t],
        "deme_upload_contact_wildcard": [_P, C.c_uint32, _P, C.c_size_t],
        "deme_seed_contacts": [_P, _P, _P, _P, _P, C.c_size_t],
        "deme_compile_prescriptions": [_P, C.c_char_p, C.c_char_p, C.c_char_p],
        "deme_upload_wildcard_array": [_P, C.c_uint32, C.c_uint32, _P, C.c_size_t],
        "deme_download_wildcard_array": [_P, C.c_uint32, C.c_uint32, _P, C.c_size_t],
        "deme_halo_stream": [_P, C.POINTER(_P)], "deme_halo_pack_async": [_P, _P, C.c_uint32, _P],
        "deme_halo_unpack_async": [_P, _P, C.c_uint32, _P], "deme_halo_sync": [_P],
        "deme_step_overlap_begin": [_P, C.POINTER(C.c_int)], "deme_step_overlap_end": [_P],
        "deme_compile_family_rules": [_P, C.c_char_p], "deme_change_family": [_P, C.c_uint32, C.c_uint32],
        "deme_set_family_material": [_P, C.c_uint32, C.c_uint32, C.c_int],
        "deme_device_memory": [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
        "deme_set_adaptive": [_P, C.POINTER(DemeAdaptive)],
        "deme_get_adaptive_state": [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
        "deme_mark_persistent_contacts": [_P, C.c_int, C.c_uint32, C.c_uint32, C.c_int],
        "deme_num_persistent_contacts": [_P, C.POINTER(C.c_size_t)],
        "deme_add_owner_acc": [_P, C.c_uint32, C.c_uint32, _P, _P],
        "deme_download_persistent_contacts": [_P, _P, _P, _P, C.c_size_t],
        "deme_upload_persistent_contacts": [_P, _P, _P, _P, C.c_size_t],
        "deme_inspect": [_P, C.c_uint32, C.POINTER(C.c_float)],
        "deme_compile_region": [_P, C.c_char_p, C.POINTER(C.c_int)],
        "deme_inspect_region": [_P, C.c_uint32, C.c_int, C.POINTER(C.c_float)],
        "deme_upload_volumes": [_P, _P, C.c_size_t], "deme_inspect_values": [_P, C.c_uint32, _P, C.c_size_t],
        "deme_set_record_contacts": [_P, C.c_int],
        "deme_download_contact_records": [_P, _P, _P, _P, _P, C.c_size_t],
        "deme_download_sphere_geometry": [_P, _P, _P, _P, _P, C.c_size_t],
        "deme_compile_force_model": [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p],
        "deme_compile_force_model_ex": [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p),
                                        C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p],
        "deme_kernel_time_ms": [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)],
        "deme_kernel_time_reset": [_P], "deme_set_timing": [_P, C.c_int],
        "deme_halo_pack": [_P, _P, C.c_uint32, _P], "deme_halo_unpack": [_P, _P, C.c_uint32, _P],
        "deme_jit_probe": [C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.c_char_p, C.c_size_t],
        "deme_jit_probe_ex": [C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32,
                              C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.c_char_p, C.c_size_t],
    }.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.deme_halo_unique_id.argtypes = [_P]
    lib.deme_halo_unique_id.restype = C.c_int
    lib.deme_halo_group_create.argtypes = [_P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]
    lib.deme_halo_group_create.restype = C.c_int
    lib.deme_halo_group_destroy.argtypes = [_P]
    lib.deme_halo_group_destroy.restype = None
    lib.deme_halo_group_last_error.argtypes = [_P]
    lib.deme_halo_group_last_error.restype = C.c_char_p
    lib.deme_halo_group_attach.argtypes = [_P, _P, C.c_int, _P, _P, C.c_uint32, _P, C.c_uint32, C.c_int, _P, _P, C.c_uint32, _P, C.c_uint32]
    lib.deme_halo_group_attach.restype = C.c_int
    for name, args in {"deme_halo_group_step": [_P, C.c_uint32], "deme_halo_group_exchange": [_P], "deme_halo_group_sync": [_P],
                       "deme_halo_group_stats": [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]}.items():
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def halo_unique_id():
    """ncclGetUniqueId through the library: 128 bytes rank 0 hands to the other ranks"""
    buf = (C.c_ubyte * 128)()
    if load_library().deme_halo_unique_id(buf) != 0:
        raise DemeError("RCCL is not available (deme_halo_unique_id)")
    return bytes(buf)


class HaloGroup:
    """The library-side ghost exchange (include/deme_hip.h, deme_halo_group_*): one RCCL communicator + the slabs this process
    holds.  unique_id None with world 1: a one-rank communicator (slabs of one process exchange by sends to self)."""

    def __init__(self, rank=0, world=1, device=0, unique_id=None):
        self.lib = load_library()
        h = _P()
        idbuf = None if unique_id is None else (C.c_ubyte * 128).from_buffer_copy(unique_id)
        rc = self.lib.deme_halo_group_create(idbuf, int(rank), int(world), int(device), C.byref(h))
        self.h = h
        self.rank, self.world = rank, world
        self._ck(rc, "deme_halo_group_create")

    def _ck(self, rc, what):
        if rc != 0:
            msg = self.lib.deme_halo_group_last_error(self.h) if self.h else b""
            raise DemeError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")

    def attach(self, ctx, part, left=None, right=None):
        """part: one entry of decomp.decompose(); left / right: the neighbour as a rank (int, another process) or as the
        Context of a slab this process holds too, None at the ends of the chain"""
        def side(nb, send, recv):
            s = np.ascontiguousarray(send, np.uint32)
            r = np.ascontiguousarray(recv, np.uint32)
            if nb is None:
                return -1, None, s, r
            if isinstance(nb, Context):
                return self.rank, nb.h, s, r
            return int(nb), None, s, r
        lr, lc, ls, lrv = side(left, part["send_left"], part["recv_left"])
        rr, rc_, rs, rrv = side(right, part["send_right"], part["recv_right"])
        self._keep = getattr(self, "_keep", []) + [(ls, lrv, rs, rrv)]
        self._ctxs = getattr(self, "_ctxs", []) + [ctx]
        self._ck(self.lib.deme_halo_group_attach(self.h, ctx.h, lr, lc, _ptr(ls), ls.size, _ptr(lrv), lrv.size, rr, rc_, _ptr(rs),
                                                  rs.size, _ptr(rrv), rrv.size), "deme_halo_group_attach")

    def step(self, n):
        self._ck(self.lib.deme_halo_group_step(self.h, int(n)), "deme_halo_group_step")

    def exchange(self):
        self._ck(self.lib.deme_halo_group_exchange(self.h), "deme_halo_group_exchange")

    def sync(self):
        self._ck(self.lib.deme_halo_group_sync(self.h), "deme_halo_group_sync")

    def stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._ck(self.lib.deme_halo_group_stats(self.h, C.byref(a), C.byref(b)), "deme_halo_group_stats")
        return int(a.value), int(b.value)

    def set_slab(self, ctx, part, halo, flip_mask=7):
        """what deme_halo_group_migrate has to know about a slab: global ids of its owners and spheres (part = an entry of
        decomp.decompose / the result of a migration), its x range and the halo thickness; flip_mask: contact wildcards that are
        B -> A vectors (bit w)"""
        og = np.ascontiguousarray(part["owner_global"], np.uint32)
        sg = np.ascontiguousarray(part["sphere_global"], np.uint32)
        lo, hi = (float(v) for v in part["edges"])
        self.lib.deme_halo_group_set_slab.argtypes = [_P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double,
                                                      C.c_uint32]
        self._ck(self.lib.deme_halo_group_set_slab(self.h, ctx.h, _ptr(og), _ptr(sg), int(part["n_own"]), len(part["ghost_left_g"]),
                                                   len(part["ghost_right_g"]), lo, hi, float(halo), int(flip_mask)), "deme_halo_group_set_slab")

    def migrate(self):
        """deme_halo_group_migrate: clumps that crossed a face move to the face neighbour (state, template ids, contact history), ghost
        sets and exchange lists are renewed -- on the device, over RCCL between ranks.  Returns the clumps this process sent away."""
        n = C.c_uint32(0)
        self.lib.deme_halo_group_migrate.argtypes = [_P, C.POINTER(C.c_uint32)]
        self._ck(self.lib.deme_halo_group_migrate(self.h, C.byref(n)), "deme_halo_group_migrate")
        for c in getattr(self, "_ctxs", []):  # the slabs have new owner / sphere counts
            cnt = self.slab_counts(c)
            c.n_owners, c.n_spheres = cnt[3], cnt[4]
        return int(n.value)

    def slab_counts(self, ctx):
        """(own clumps, ghosts from the left, ghosts from the right, owners, spheres, seeded contacts) of a slab"""
        v = (C.c_uint32 * 6)()
        self.lib.deme_halo_group_slab_counts.argtypes = [_P, _P, C.POINTER(C.c_uint32)]
        self._ck(self.lib.deme_halo_group_slab_counts(self.h, ctx.h, v), "deme_halo_group_slab_counts")
        return tuple(int(x) for x in v)

    def slab_ids(self, ctx):
        """(global clump id per owner, global sphere id per sphere, owner per sphere, component per sphere) in the slab's current numbering"""
        n_own, n_gl, n_gr, n_o, n_s, _ = self.slab_counts(ctx)
        og, sg, so, sc = np.zeros(n_o, np.uint32), np.zeros(n_s, np.uint32), np.zeros(n_s, np.uint32), np.zeros(n_s, np.uint16)
        self.lib.deme_halo_group_download_ids.argtypes = [_P, _P, _P, _P, _P, _P]
        self._ck(self.lib.deme_halo_group_download_ids(self.h, ctx.h, _ptr(og), _ptr(sg), _ptr(so), _ptr(sc)), "deme_halo_group_download_ids")
        return og, sg, so, sc

    def comm_count(self):
        """ranks of the group's RCCL communicator, as RCCL reports it (ncclCommCount)"""
        n = C.c_int(0)
        self.lib.deme_halo_group_comm_count.argtypes = [_P, C.POINTER(C.c_int)]
        self._ck(self.lib.deme_halo_group_comm_count(self.h, C.byref(n)), "deme_halo_group_comm_count")
        return int(n.value)

    def set_cross_contacts(self, evaluate_once=True):
        """one evaluation and one history per contact that straddles a cut: the left slab evaluates, the reaction on the right
        slab's clump travels back every step (deme_halo_group_set_cross_contacts); call after every slab is attached"""
        self.lib.deme_halo_group_set_cross_contacts.argtypes = [_P, C.c_int]
        self._ck(self.lib.deme_halo_group_set_cross_contacts(self.h, int(bool(evaluate_once))), "deme_halo_group_set_cross_contacts")

    def host_time(self, reset=False):
        """host microseconds spent enqueuing: (interior passes, packs, RCCL group, unpack + boundary pass + integration)"""
        us = (C.c_double * 4)()
        self.lib.deme_halo_group_host_time.argtypes = [_P, C.POINTER(C.c_double), C.c_int]
        self._ck(self.lib.deme_halo_group_host_time(self.h, us, 1 if reset else 0), "deme_halo_group_host_time")
        return tuple(us)

    def close(self):
        if getattr(self, "h", None):
            self.lib.deme_halo_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DecompPlan:
    """deme_decomp_*: the slab decomposition of a global scene, computed by the library on the host (no device needed).  The plan
    keeps pointers into the scene's arrays while it is built only; a slab's scene points into the plan (keep the plan alive while
    a slab scene is in use)."""

    def __init__(self, params, scene, n_slabs, axis=-1, halo=0.0, edges=None, shared_free=False, snap=True, spatial_order=False):
        self.lib = load_library()
        L = self.lib
        L.deme_decomp_create.argtypes = [C.POINTER(DemeParams), C.POINTER(DemeScene), C.c_uint32, C.c_int, C.c_double, _P, C.c_uint32,
                                         C.POINTER(_P), C.c_char_p, C.c_size_t]
        L.deme_decomp_destroy.argtypes = [_P]
        L.deme_decomp_destroy.restype = None
        L.deme_decomp_info.argtypes = [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_double), _P, C.POINTER(C.c_uint32)]
        L.deme_decomp_slab.argtypes = [_P, C.c_uint32, C.POINTER(DemeScene), C.POINTER(C.c_uint32), C.POINTER(_P), C.POINTER(_P),
                                       C.POINTER(_P), C.POINTER(C.c_uint32), C.POINTER(_P), C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        h = _P()
        err = C.create_string_buffer(1024)
        e = None if edges is None else np.ascontiguousarray(edges, np.float64)
        flags = (1 if shared_free else 0) | (0 if snap else 2) | (4 if spatial_order else 0)
        rc = L.deme_decomp_create(C.byref(params), C.byref(scene), int(n_slabs), int(axis), float(halo), None if e is None else _ptr(e),
                                  flags, C.byref(h), err, len(err))
        if rc != 0:
            raise DemeError(f"deme_decomp_create failed (status {rc}): {err.value.decode()}")
        self.h = h
        self._scene = scene  # (the tables a slab scene shares with the scene it was cut from)
        n, ax, hl, nf = C.c_uint32(0), C.c_int(0), C.c_double(0), C.c_uint32(0)
        L.deme_decomp_info(self.h, C.byref(n), C.byref(ax), C.byref(hl), None, C.byref(nf))
        self.n_slabs, self.axis, self.halo, self.n_free = int(n.value), int(ax.value), float(hl.value), int(nf.value)
        ed = np.zeros(self.n_slabs + 1, np.float64)
        L.deme_decomp_info(self.h, None, None, None, _ptr(ed), None)
        self.edges = ed

    def slab(self, s):
        """dict: scene (DemeScene), n_own, n_ghost_lower, n_ghost_upper, owner_global, sphere_global, send_lower, send_upper, range"""
        sc = DemeScene()
        cnt = (C.c_uint32 * 3)()
        og, sg, sl, su = _P(), _P(), _P(), _P()
        nl, nu = C.c_uint32(0), C.c_uint32(0)
        rg = (C.c_double * 2)()
        rc = self.lib.deme_decomp_slab(self.h, int(s), C.byref(sc), cnt, C.byref(og), C.byref(sg), C.byref(sl), C.byref(nl), C.byref(su),
                                       C.byref(nu), rg)
        if rc != 0:
            raise DemeError(f"deme_decomp_slab({s}) failed (status {rc})")

        def arr(ptr, n):
            if not n:
                return np.zeros(0, np.uint32)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(int(n),)).copy()
        sc._keep = self  # the plan owns the slab's arrays
        return {"scene": sc, "n_own": int(cnt[0]), "n_ghost_lower": int(cnt[1]), "n_ghost_upper": int(cnt[2]),
                "owner_global": arr(og, sc.nOwners), "sphere_global": arr(sg, sc.nSpheres), "send_lower": arr(sl, nl.value),
                "send_upper": arr(su, nu.value), "range": (float(rg[0]), float(rg[1]))}

    def close(self):
        if getattr(self, "h", None):
            self.lib.deme_decomp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def copy_rate_probe(device=0, nbytes=1 << 30, reps=5):
    """GB/s (read + written) of the library's own 16 B / lane streaming copy kernel: the attainable HBM rate of the box"""
    lib = load_library()
    v = C.c_double(0)
    lib.deme_copy_rate_probe.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    rc = lib.deme_copy_rate_probe(int(device), int(nbytes), int(reps), C.byref(v))
    if rc != 0:
        raise DemeError(f"deme_copy_rate_probe failed (status {rc})")
    return float(v.value)


def device_count():
    lib = load_library()
    n = C.c_int(0)
    lib.deme_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.deme_device_count(C.byref(n))
    return int(n.value)


class Multi:
    """deme_multi_*: one process, several devices -- what the C++ shell's DEMSolver(nGPUs / device ids) opens.  One device with
    several slabs exercises every step of it on a single GPU (the slabs exchange by sends to self)."""

    def __init__(self, devices=(0,)):
        self.lib = load_library()
        L = self.lib
        L.deme_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(_P), C.c_char_p, C.c_size_t]
        L.deme_multi_destroy.argtypes = [_P]
        L.deme_multi_destroy.restype = None
        L.deme_multi_last_error.argtypes = [_P]
        L.deme_multi_last_error.restype = C.c_char_p
        L.deme_multi_build.argtypes = [_P, C.POINTER(DemeParams), C.POINTER(DemeScene), C.c_uint32, C.c_int, C.c_double, C.c_uint32, C.c_int,
                                       C.c_uint32]
        L.deme_multi_num_slabs.argtypes = [_P, C.POINTER(C.c_uint32)]
        L.deme_multi_slab_ctx.argtypes = [_P, C.c_uint32, C.POINTER(_P)]
        L.deme_multi_set_migration.argtypes = [_P, C.c_uint32]
        L.deme_multi_step.argtypes = [_P, C.c_uint32]
        L.deme_multi_sync.argtypes = [_P]
        L.deme_multi_download_state.argtypes = [_P, C.POINTER(DemeOwnerState), C.c_uint32]
        L.deme_multi_upload_state.argtypes = [_P, C.POINTER(DemeOwnerState), C.c_uint32]
        L.deme_multi_counts.argtypes = [_P, C.POINTER(DemeCounts), C.POINTER(C.c_uint64)]
        dev = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = _P()
        err = C.create_string_buffer(1024)
        rc = L.deme_multi_create(dev, len(devices), C.byref(h), err, len(err))
        if rc != 0:
            raise DemeError(f"deme_multi_create failed (status {rc}): {err.value.decode()}")
        self.h = h
        self.n_owners = 0

    def _ck(self, rc, what):
        if rc != 0:
            msg = self.lib.deme_multi_last_error(self.h)
            raise DemeError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")

    def build(self, params, scene, slabs_per_device=1, axis=-1, halo=0.0, shared_free=False, arith=None, flip_mask=7, caller_order=False):
        a = -1 if arith is None else {"fast": 1, "exact": 0}.get(arith, arith)
        self._ck(self.lib.deme_multi_build(self.h, C.byref(params), C.byref(scene), int(slabs_per_device), int(axis), float(halo),
                                           (1 if shared_free else 0) | (8 if caller_order else 0), int(a), int(flip_mask)), "deme_multi_build")
        self.n_owners = int(scene.nOwners)

    def num_slabs(self):
        n = C.c_uint32(0)
        self._ck(self.lib.deme_multi_num_slabs(self.h, C.byref(n)), "deme_multi_num_slabs")
        return int(n.value)

    def slab_ctx(self, s):
        """the slab's context as a borrowed Context (owned by the library: do not close it)"""
        h = _P()
        self._ck(self.lib.deme_multi_slab_ctx(self.h, int(s), C.byref(h)), "deme_multi_slab_ctx")
        return Context.borrowed(h)

    def set_migration(self, every):
        self._ck(self.lib.deme_multi_set_migration(self.h, int(every)), "deme_multi_set_migration")

    def step(self, n):
        self._ck(self.lib.deme_multi_step(self.h, int(n)), "deme_multi_step")

    def sync(self):
        self._ck(self.lib.deme_multi_sync(self.h), "deme_multi_sync")

    def download_state(self):
        st, out = make_state_struct(self.n_owners)
        self._ck(self.lib.deme_multi_download_state(self.h, C.byref(st), self.n_owners), "deme_multi_download_state")
        return out

    def upload_state(self, arrays):
        st, keep = make_state_struct(self.n_owners, arrays)
        self._ck(self.lib.deme_multi_upload_state(self.h, C.byref(st), self.n_owners), "deme_multi_upload_state")

    def counts(self):
        c, mig = DemeCounts(), C.c_uint64(0)
        self._ck(self.lib.deme_multi_counts(self.h, C.byref(c), C.byref(mig)), "deme_multi_counts")
        return c, int(mig.value)

    def rebalance(self):
        """deme_multi_rebalance: (clumps moved, boundaries in force afterwards)"""
        n = C.c_uint32(0)
        e = np.zeros(self.num_slabs() + 1, np.float64)
        self.lib.deme_multi_rebalance.argtypes = [_P, C.POINTER(C.c_uint32), _P]
        self._ck(self.lib.deme_multi_rebalance(self.h, C.byref(n), _ptr(e)), "deme_multi_rebalance")
        return int(n.value), e

    def wildcard_array(self, kind, index, n):
        """a user model's wildcard array by GLOBAL id (kind: owner, sphere; n = the global scene's count)"""
        out = np.zeros(int(n), np.float32)
        self.lib.deme_multi_download_wildcard_array.argtypes = [_P, C.c_uint32, C.c_uint32, _P, C.c_size_t]
        self._ck(self.lib.deme_multi_download_wildcard_array(self.h, Context.WC_KINDS[kind], int(index), _ptr(out), out.size),
                 "deme_multi_download_wildcard_array")
        return out

    def set_wildcard_array(self, kind, index, values):
        v = np.ascontiguousarray(values, dtype=np.float32)
        self.lib.deme_multi_upload_wildcard_array.argtypes = [_P, C.c_uint32, C.c_uint32, _P, C.c_size_t]
        self._ck(self.lib.deme_multi_upload_wildcard_array(self.h, Context.WC_KINDS[kind], int(index), _ptr(v), v.size),
                 "deme_multi_upload_wildcard_array")

    def set_contact_wildcard(self, w, values):
        """a per-contact wildcard column of the merged list (every slab's copy of a pair takes the value)"""
        v = np.ascontiguousarray(values, dtype=np.float32)
        self.lib.deme_multi_upload_contact_wildcard.argtypes = [_P, C.c_uint32, _P, C.c_size_t]
        self._ck(self.lib.deme_multi_upload_contact_wildcard(self.h, int(w), _ptr(v), v.size), "deme_multi_upload_contact_wildcard")

    def seed_contacts(self, idA, idB, types, wildcards):
        """deme_multi_seed_contacts: a saved list in GLOBAL ids (restart); wildcards (n, nW) float32 or None"""
        a, b = np.ascontiguousarray(idA, np.uint32), np.ascontiguousarray(idB, np.uint32)
        t = np.ascontiguousarray(types, np.uint8)
        w = None if wildcards is None else np.ascontiguousarray(wildcards, np.float32)
        self.lib.deme_multi_seed_contacts.argtypes = [_P, _P, _P, _P, _P, C.c_size_t]
        self._ck(self.lib.deme_multi_seed_contacts(self.h, _ptr(a), _ptr(b), _ptr(t), None if w is None else _ptr(w), a.size),
                 "deme_multi_seed_contacts")

    def mark_persistent_contacts(self, mode=0, n1=0, n2=0, mark=1):
        self.lib.deme_multi_mark_persistent_contacts.argtypes = [_P, C.c_int, C.c_uint32, C.c_uint32, C.c_int]
        self._ck(self.lib.deme_multi_mark_persistent_contacts(self.h, int(mode), int(n1), int(n2), int(mark)), "deme_multi_mark_persistent_contacts")

    def persistent_contacts(self):
        n = C.c_size_t(0)
        self.lib.deme_multi_num_persistent_contacts.argtypes = [_P, C.POINTER(C.c_size_t)]
        self._ck(self.lib.deme_multi_num_persistent_contacts(self.h, C.byref(n)), "deme_multi_num_persistent_contacts")
        a, b, t = np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint8)
        self.lib.deme_multi_download_persistent_contacts.argtypes = [_P, _P, _P, _P, C.c_size_t]
        self._ck(self.lib.deme_multi_download_persistent_contacts(self.h, _ptr(a), _ptr(b), _ptr(t), n.value), "deme_multi_download_persistent_contacts")
        return a, b, t

    def set_persistent_contacts(self, idA, idB, types):
        a, b = np.ascontiguousarray(idA, np.uint32), np.ascontiguousarray(idB, np.uint32)
        t = np.ascontiguousarray(types, np.uint8)
        self.lib.deme_multi_upload_persistent_contacts.argtypes = [_P, _P, _P, _P, C.c_size_t]
        self._ck(self.lib.deme_multi_upload_persistent_contacts(self.h, _ptr(a), _ptr(b), _ptr(t), a.size), "deme_multi_upload_persistent_contacts")

    def inspect_values(self, quantity, n):
        """deme_multi_inspect_values: per sphere / per owner, by GLOBAL id (n = the global scene's count)"""
        out = np.zeros(int(n), np.float32)
        self.lib.deme_multi_inspect_values.argtypes = [_P, C.c_uint32, _P, C.c_size_t]
        self._ck(self.lib.deme_multi_inspect_values(self.h, Context.INSPECT_CODES[quantity], _ptr(out), out.size), "deme_multi_inspect_values")
        return out

    def reset(self):
        """deme_multi_reset: slabs, contexts and plan dropped, the groups opened anew; build() follows"""
        self.lib.deme_multi_reset.argtypes = [_P]
        self._ck(self.lib.deme_multi_reset(self.h), "deme_multi_reset")

    def set_rebalance(self, every_nth_migration):
        self.lib.deme_multi_set_rebalance.argtypes = [_P, C.c_uint32]
        self._ck(self.lib.deme_multi_set_rebalance(self.h, int(every_nth_migration)), "deme_multi_set_rebalance")

    def slab_counts(self, s):
        """((own, ghosts below, ghosts above, owners, spheres, seeded contacts), (lo, hi)) of a slab"""
        v, r = (C.c_uint32 * 6)(), (C.c_double * 2)()
        self.lib.deme_multi_slab_counts.argtypes = [_P, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        self._ck(self.lib.deme_multi_slab_counts(self.h, int(s), v, r), "deme_multi_slab_counts")
        return tuple(int(x) for x in v), (float(r[0]), float(r[1]))

    def add_owner_acc(self, owner, acc=None, ang_acc=None):
        """deme_multi_add_owner_acc: accelerations for the next step, by GLOBAL owner id"""
        a = None if acc is None else np.ascontiguousarray(acc, np.float32).reshape(-1, 3)
        l = None if ang_acc is None else np.ascontiguousarray(ang_acc, np.float32).reshape(-1, 3)
        n = len(a) if a is not None else len(l)
        self.lib.deme_multi_add_owner_acc.argtypes = [_P, C.c_uint32, C.c_uint32, _P, _P]
        self._ck(self.lib.deme_multi_add_owner_acc(self.h, int(owner), n, None if a is None else _ptr(a), None if l is None else _ptr(l)),
                 "deme_multi_add_owner_acc")

    def num_contacts(self):
        n = C.c_size_t(0)
        self.lib.deme_multi_num_contacts.argtypes = [_P, C.POINTER(C.c_size_t)]
        self._ck(self.lib.deme_multi_num_contacts(self.h, C.byref(n)), "deme_multi_num_contacts")
        return int(n.value)

    def contacts(self):
        """(idA, idB, type) of the merged list in GLOBAL sphere ids: a pair that straddles a cut once, canonical order"""
        n = self.num_contacts()
        a, b, t = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
        self.lib.deme_multi_download_contacts.argtypes = [_P, _P, _P, _P, C.c_size_t]
        self._ck(self.lib.deme_multi_download_contacts(self.h, _ptr(a), _ptr(b), _ptr(t), n), "deme_multi_download_contacts")
        return a, b, t

    def wildcard(self, w):
        n = self.num_contacts()
        out = np.zeros(n, np.float32)
        self.lib.deme_multi_download_contact_wildcard.argtypes = [_P, C.c_uint32, _P, C.c_size_t]
        self._ck(self.lib.deme_multi_download_contact_wildcard(self.h, int(w), _ptr(out), n), "deme_multi_download_contact_wildcard")
        return out

    def contact_records(self):
        n = self.num_contacts()
        arrs = [np.zeros((n, 3), np.float32) for _ in range(4)]
        self.lib.deme_multi_download_contact_records.argtypes = [_P, _P, _P, _P, _P, C.c_size_t]
        self._ck(self.lib.deme_multi_download_contact_records(self.h, *[_ptr(a) for a in arrs], n), "deme_multi_download_contact_records")
        return arrs

    def close(self):
        if getattr(self, "h", None):
            self.lib.deme_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def exported_symbols():
    """Names include/deme_hip.h declares (parsed from the header)."""
    import re
    hdr = os.path.join(_HERE, "..", "include", "deme_hip.h")
    txt = open(hdr).read()
    return sorted(set(re.findall(r"\b(deme_[a-z_0-9]+)\s*\(", txt)))


def _ptr(a):
    return None if a is None else a.ctypes.data


def jit_probe(src, wildcard_names=(), prerequisites="", owner_wildcards=(), geo_wildcards=()):
    """Compile-only check of a user force fragment (hipRTC, gfx950); needs no GPU.  Returns (ok, log)."""
    lib = load_library()

    def arr(ns):
        return (C.c_char_p * max(1, len(ns)))(*[s.encode() for s in ns])
    log = C.create_string_buffer(8192)
    rc = lib.deme_jit_probe_ex(src.encode(), arr(wildcard_names), len(wildcard_names), arr(owner_wildcards), len(owner_wildcards),
                               arr(geo_wildcards), len(geo_wildcards), prerequisites.encode(), log, len(log))
    return rc == 0, log.value.decode(errors="replace")


class Context:
    """One GPU context (mirrors what kT+dT own in the reference)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = _P()
        rc = self.lib.deme_ctx_create(int(device), C.byref(h))
        if rc != 0 or not h:
            raise DemeError(f"deme_ctx_create failed (status {rc}): is a HIP device visible?")
        self.h = h
        self.n_owners = 0
        self.n_spheres = 0
        self.n_wildcards = 0

    @classmethod
    def borrowed(cls, handle):
        """a context the LIBRARY owns (a slab of deme_halo_group_build / deme_multi_build): every call works, close() leaves it alone"""
        self = cls.__new__(cls)
        self.lib = load_library()
        self.h = handle
        self._borrowed = True
        v = (C.c_uint32 * 4)()
        self.lib.deme_scene_sizes.argtypes = [_P, C.POINTER(C.c_uint32)]
        self._ck(self.lib.deme_scene_sizes(self.h, v), "deme_scene_sizes")
        self.n_owners, self.n_spheres, self.n_wildcards = int(v[0]), int(v[2]), int(v[3])
        return self

    def refresh_sizes(self):
        """owner / sphere counts change when clumps migrate between slabs"""
        v = (C.c_uint32 * 4)()
        self.lib.deme_scene_sizes.argtypes = [_P, C.POINTER(C.c_uint32)]
        self._ck(self.lib.deme_scene_sizes(self.h, v), "deme_scene_sizes")
        self.n_owners, self.n_spheres = int(v[0]), int(v[2])

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                self.lib.deme_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            msg = self.lib.deme_last_error(self.h)
            raise DemeError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")

    def set_stream(self, stream_ptr):
        self._ck(self.lib.deme_ctx_set_stream(self.h, stream_ptr), "deme_ctx_set_stream")

    def sync(self):
        self._ck(self.lib.deme_sync(self.h), "deme_sync")

    def set_arith_mode(self, mode):
        """'fast' (default: world-frame force kernel, 1-ulp physics arithmetic) or 'exact' (the reference's operation order,
        bit-identical to the CPU oracle); include/deme_hip.h DEME_ARITH_*"""
        m = {"fast": ARITH_FAST, "exact": ARITH_EXACT}.get(mode, mode)
        self._ck(self.lib.deme_set_arith_mode(self.h, int(m)), "deme_set_arith_mode")

    def arith_mode(self):
        return "fast" if self.lib.deme_get_arith_mode(self.h) == ARITH_FAST else "exact"

    def set_params(self, p):
        self.n_wildcards = int(p.nContactWildcards)
        self._ck(self.lib.deme_set_params(self.h, C.byref(p)), "deme_set_params")

    def upload_scene(self, sc):
        self.n_owners, self.n_spheres = int(sc.nOwners), int(sc.nSpheres)
        self._ck(self.lib.deme_upload_scene(self.h, C.byref(sc)), "deme_upload_scene")

    def upload_state(self, arrays):
        st, _ = make_state_struct(self.n_owners, arrays)
        self._ck(self.lib.deme_upload_owner_state(self.h, C.byref(st)), "deme_upload_owner_state")

    def force_kernel(self):
        """(name of the kernel that evaluates the contact forces of the current list, largest tile's foreign owners, local list)"""
        buf = C.create_string_buffer(64)
        h, l = C.c_uint32(0), C.c_uint32(0)
        self.lib.deme_force_kernel_name.argtypes = [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        self._ck(self.lib.deme_force_kernel_name(self.h, buf, 64, C.byref(h), C.byref(l)), "deme_force_kernel_name")
        return buf.value.decode(), int(h.value), int(l.value)

    def tile_stats(self):
        """(tiles, tiles evaluated by the per-tile fallback kernel, largest halo, largest local list) of the current list"""
        out = (C.c_uint32 * 4)()
        self.lib.deme_tile_stats.argtypes = [_P, C.POINTER(C.c_uint32)]
        self._ck(self.lib.deme_tile_stats(self.h, out), "deme_tile_stats")
        return tuple(int(v) for v in out)

    def set_tile_policy(self, min_contacts_per_tile_custom):
        self.lib.deme_set_tile_policy.argtypes = [_P, C.c_uint32]
        self._ck(self.lib.deme_set_tile_policy(self.h, int(min_contacts_per_tile_custom)), "deme_set_tile_policy")

    def set_fused_step(self, on):
        """the one-kernel step (k_tile_step<M>, deme_tile_step.h); off by default: deme_set_fused_step"""
        self.lib.deme_set_fused_step.argtypes = [_P, C.c_int]
        self._ck(self.lib.deme_set_fused_step(self.h, 2 if on == "auto" else (1 if on else 0)), "deme_set_fused_step")

    def engine_order(self):
        """(reordered, spread in the caller's order, spread along the curve): deme_get_order"""
        r, sp = C.c_int(0), (C.c_double * 2)()
        self.lib.deme_get_order.argtypes = [_P, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        self._ck(self.lib.deme_get_order(self.h, C.byref(r), sp), "deme_get_order")
        return bool(r.value), float(sp[0]), float(sp[1])

    def order_renewals(self):
        n = C.c_uint64(0)
        self.lib.deme_order_renewals.argtypes = [_P, C.POINTER(C.c_uint64)]
        self._ck(self.lib.deme_order_renewals(self.h, C.byref(n)), "deme_order_renewals")
        return int(n.value)

    def renew_order(self):
        self._ck(self.lib.deme_renew_order(self.h), "deme_renew_order")

    def set_reorder(self, enable):
        self.lib.deme_set_reorder.argtypes = [_P, C.c_int]
        self._ck(self.lib.deme_set_reorder(self.h, int(bool(enable))), "deme_set_reorder")

    def download_state(self):
        st, out = make_state_struct(self.n_owners)
        self._ck(self.lib.deme_download_owner_state(self.h, C.byref(st)), "deme_download_owner_state")
        return out

    def update_tri_nodes(self, n1, n2, n3):
        arrs = [np.ascontiguousarray(x, np.float32).reshape(-1) for x in (n1, n2, n3)]
        self._ck(self.lib.deme_update_tri_nodes(self.h, *[_ptr(x) for x in arrs]), "deme_update_tri_nodes")

    def set_async_detection(self, lead_steps):
        """deme_set_async_detection: start each detection lead_steps before its list is due, beside the steps (0: lock-step)"""
        self._ck(self.lib.deme_set_async_detection(self.h, C.c_uint32(int(lead_steps))), "deme_set_async_detection")

    def compute_margins(self, drift):
        self._ck(self.lib.deme_compute_margins(self.h, int(drift)), "deme_compute_margins")

    def set_margins(self, m):
        m = np.ascontiguousarray(m, dtype=np.float32)
        assert m.size == self.n_owners
        self._ck(self.lib.deme_set_margins(self.h, m.ctypes.data), "deme_set_margins")

    def detect(self):
        self._ck(self.lib.deme_detect_contacts(self.h), "deme_detect_contacts")

    def migrate(self):
        self._ck(self.lib.deme_migrate_history(self.h), "deme_migrate_history")

    def calc_forces(self):
        self._ck(self.lib.deme_calc_forces(self.h), "deme_calc_forces")

    def integrate(self):
        self._ck(self.lib.deme_integrate(self.h), "deme_integrate")

    def step(self, n):
        self._ck(self.lib.deme_step(self.h, int(n)), "deme_step")

    def counts(self):
        c = DemeCounts()
        self._ck(self.lib.deme_get_counts(self.h, C.byref(c)), "deme_get_counts")
        return c

    def bin_incidence(self):
        n = int(self.counts().nBinSphereTouches)
        b = np.zeros(n, np.uint32)
        s = np.zeros(n, np.uint32)
        self._ck(self.lib.deme_download_bin_incidence(self.h, _ptr(b), _ptr(s), n), "deme_download_bin_incidence")
        return b, s

    def contacts(self):
        n = int(self.counts().nContacts)
        a = np.zeros(n, np.uint32)
        b = np.zeros(n, np.uint32)
        t = np.zeros(n, np.uint8)
        m = np.zeros(n, np.uint32)
        self._ck(self.lib.deme_download_contacts(self.h, _ptr(a), _ptr(b), _ptr(t), _ptr(m), n),
                 "deme_download_contacts")
        return a, b, t, m

    def wildcard(self, w):
        n = int(self.counts().nContacts)
        out = np.zeros(n, np.float32)
        self._ck(self.lib.deme_download_contact_wildcard(self.h, int(w), _ptr(out), n),
                 "deme_download_contact_wildcard")
        return out

    def set_wildcard(self, w, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        self._ck(self.lib.deme_upload_contact_wildcard(self.h, int(w), _ptr(arr), arr.size),
                 "deme_upload_contact_wildcard")

    # DEMInspector quantity names (AuxClasses.cpp:94-170)
    INSPECT_CODES = {"clump_max_z": 0, "clump_min_z": 1, "clump_max_absv": 2, "clump_mass": 3, "max_absv": 4,
                     "clump_kinetic_energy": 5, "absv": 6, "clump_volume": 7}

    def compile_region(self, code):
        """CreateInspector(quantity, region): compiles the region string once, returns the handle inspect(region=) takes."""
        rid = C.c_int(-1)
        self._ck(self.lib.deme_compile_region(self.h, code.encode(), C.byref(rid)), "deme_compile_region")
        return int(rid.value)

    def inspect(self, quantity, region=-1):
        """DEMInspector::GetValue of a named quantity (reduced on the device), optionally limited to a compiled region."""
        if quantity not in self.INSPECT_CODES:
            raise DemeError(f"{quantity} is not a known query type.")
        out = C.c_float(0)
        self._ck(self.lib.deme_inspect_region(self.h, self.INSPECT_CODES[quantity], int(region), C.byref(out)), "deme_inspect_region")
        return float(out.value)

    def upload_volumes(self, volumes):
        v = np.ascontiguousarray(volumes, np.float32)
        self._ck(self.lib.deme_upload_volumes(self.h, _ptr(v), v.size), "deme_upload_volumes")

    def inspect_values(self, quantity, n):
        """DEMInspector::GetValues: the unreduced per-sphere / per-owner array (n elements)."""
        out = np.zeros(int(n), np.float32)
        self._ck(self.lib.deme_inspect_values(self.h, self.INSPECT_CODES[quantity], _ptr(out), out.size), "deme_inspect_values")
        return out

    # ---- halo exchange overlapped with the interior force evaluation (see include/deme_hip.h)
    def halo_stream(self):
        st = _P()
        self._ck(self.lib.deme_halo_stream(self.h, C.byref(st)), "deme_halo_stream")
        return st.value

    def halo_pack_async(self, d_ids, n, d_buf):
        self._ck(self.lib.deme_halo_pack_async(self.h, _P(d_ids), int(n), _P(d_buf)), "deme_halo_pack_async")

    def halo_unpack_async(self, d_ids, n, d_buf):
        self._ck(self.lib.deme_halo_unpack_async(self.h, _P(d_ids), int(n), _P(d_buf)), "deme_halo_unpack_async")

    def halo_sync(self):
        self._ck(self.lib.deme_halo_sync(self.h), "deme_halo_sync")

    def step_overlap_begin(self):
        due = C.c_int(0)
        self._ck(self.lib.deme_step_overlap_begin(self.h, C.byref(due)), "deme_step_overlap_begin")
        return bool(due.value)

    def step_overlap_end(self):
        self._ck(self.lib.deme_step_overlap_end(self.h), "deme_step_overlap_end")

    def compile_family_rules(self, rules):
        """ChangeFamilyWhen rules: the _familyChangeRules_ text of equipFamilyOnFlyChanges (see include/deme_hip.h)."""
        self._ck(self.lib.deme_compile_family_rules(self.h, rules.encode()), "deme_compile_family_rules")

    def set_family_material(self, family, material, meshes=False):
        self._ck(self.lib.deme_set_family_material(self.h, int(family), int(material), 1 if meshes else 0), "deme_set_family_material")

    def change_family(self, frm, to):
        self._ck(self.lib.deme_change_family(self.h, int(frm), int(to)), "deme_change_family")

    def set_adaptive(self, bin_size=False, update_freq=False, bin_observe=25, bin_max_rate=0.05, bin_acc=0.1, bin_upper_safety=0.25,
                     bin_lower_safety=0.3, max_update_freq=2500, freq_observe=4):
        """UseAdaptiveBinSize / UseAdaptiveUpdateFreq and their tuning knobs (DEM/API.h:253-309), on device timers."""
        a = DemeAdaptive(int(bool(bin_size)), int(bin_observe), float(bin_max_rate), float(bin_acc), float(bin_upper_safety),
                         float(bin_lower_safety), int(bool(update_freq)), int(max_update_freq), int(freq_observe))
        self._ck(self.lib.deme_set_adaptive(self.h, C.byref(a)), "deme_set_adaptive")

    def adaptive_state(self):
        """(bin size, cdUpdateFreq, number of bin-size changes, number of update-frequency changes)"""
        b, k, nb, nk = C.c_double(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self._ck(self.lib.deme_get_adaptive_state(self.h, C.byref(b), C.byref(k), C.byref(nb), C.byref(nk)), "deme_get_adaptive_state")
        return float(b.value), int(k.value), int(nb.value), int(nk.value)

    PERSIST_ALL, PERSIST_EITHER, PERSIST_BOTH, PERSIST_PAIR = 0, 1, 2, 3

    def mark_persistent_contacts(self, mode=0, n1=0, n2=0, mark=True):
        """MarkPersistentContact / MarkFamilyPersistentContact{Either,Both,} and their Remove* inverses (DEM/API.h:874-905)."""
        self._ck(self.lib.deme_mark_persistent_contacts(self.h, int(mode), int(n1), int(n2), int(bool(mark))),
                 "deme_mark_persistent_contacts")

    def num_persistent_contacts(self):
        n = C.c_size_t(0)
        self._ck(self.lib.deme_num_persistent_contacts(self.h, C.byref(n)), "deme_num_persistent_contacts")
        return int(n.value)

    def add_owner_acc(self, owner, acc=None, ang_acc=None):
        """DEMTracker::AddAcc / AddAngAcc: n x 3 arrays for owners [owner, owner + n), consumed by the next step only"""
        a = None if acc is None else np.ascontiguousarray(acc, np.float32).reshape(-1, 3)
        l = None if ang_acc is None else np.ascontiguousarray(ang_acc, np.float32).reshape(-1, 3)
        n = len(a) if a is not None else len(l)
        self._ck(self.lib.deme_add_owner_acc(self.h, int(owner), n, None if a is None else _ptr(a), None if l is None else _ptr(l)),
                 "deme_add_owner_acc")

    def persistent_contacts(self):
        """the marked set as (idA, idB, type) arrays"""
        n = self.num_persistent_contacts()
        a, b, t = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
        self._ck(self.lib.deme_download_persistent_contacts(self.h, _ptr(a), _ptr(b), _ptr(t), n), "deme_download_persistent_contacts")
        return a, b, t

    def set_persistent_contacts(self, idA, idB, ctype):
        a, b, t = np.ascontiguousarray(idA, np.uint32), np.ascontiguousarray(idB, np.uint32), np.ascontiguousarray(ctype, np.uint8)
        self._ck(self.lib.deme_upload_persistent_contacts(self.h, _ptr(a), _ptr(b), _ptr(t), len(a)), "deme_upload_persistent_contacts")

    def compile_prescriptions(self, vel_cases, pos_cases, acc_cases):
        """Family motion prescriptions: the three switch bodies of equipFamilyPrescribedMotions (see include/deme_hip.h)."""
        self._ck(self.lib.deme_compile_prescriptions(self.h, vel_cases.encode(), pos_cases.encode(), acc_cases.encode()),
                 "deme_compile_prescriptions")

    def seed_contacts(self, idA, idB, ctype, wildcards=None):
        """Restart: saved contact pairs (geometry ids) + wildcards [n, nW] feed the next history map."""
        a = np.ascontiguousarray(idA, dtype=np.uint32)
        b = np.ascontiguousarray(idB, dtype=np.uint32)
        t = np.ascontiguousarray(ctype, dtype=np.uint8)
        w = None if wildcards is None else np.ascontiguousarray(wildcards, dtype=np.float32)
        self._keep_seed = (a, b, t, w)
        self._ck(self.lib.deme_seed_contacts(self.h, _ptr(a), _ptr(b), _ptr(t), None if w is None else _ptr(w), a.size),
                 "deme_seed_contacts")

    def set_record_contacts(self, enable=True):
        self._ck(self.lib.deme_set_record_contacts(self.h, int(bool(enable))), "deme_set_record_contacts")

    def contact_records(self):
        n = int(self.counts().nContacts)
        arrs = [np.zeros((n, 3), np.float32) for _ in range(4)]
        self._ck(self.lib.deme_download_contact_records(self.h, *[_ptr(a) for a in arrs], n),
                 "deme_download_contact_records")
        return arrs

    def sphere_geometry(self):
        n = self.n_spheres
        X, Y, Z = (np.zeros(n, np.float64) for _ in range(3))
        R = np.zeros(n, np.float32)
        self._ck(self.lib.deme_download_sphere_geometry(self.h, _ptr(X), _ptr(Y), _ptr(Z), _ptr(R), n),
                 "deme_download_sphere_geometry")
        return X, Y, Z, R

    def compile_force_model(self, src, wildcard_names=(), prerequisites="", owner_wildcards=(), geo_wildcards=()):
        def arr(names):
            return (C.c_char_p * max(1, len(names)))(*[s.encode() for s in names])
        b = src.encode()
        self._ck(self.lib.deme_compile_force_model_ex(self.h, b, len(b), arr(wildcard_names), len(wildcard_names),
                                                      arr(owner_wildcards), len(owner_wildcards), arr(geo_wildcards),
                                                      len(geo_wildcards), prerequisites.encode()), "deme_compile_force_model_ex")

    WC_KINDS = {"owner": 0, "sphere": 1, "triangle": 2, "analytical": 3}

    def set_wildcard_array(self, kind, index, values):
        """owner / geometry wildcard array `index` of the user force model (kind: owner, sphere, triangle, analytical)"""
        v = np.ascontiguousarray(values, dtype=np.float32)
        self._ck(self.lib.deme_upload_wildcard_array(self.h, self.WC_KINDS[kind], int(index), _ptr(v), v.size),
                 "deme_upload_wildcard_array")

    def wildcard_array(self, kind, index, n):
        out = np.zeros(int(n), np.float32)
        self._ck(self.lib.deme_download_wildcard_array(self.h, self.WC_KINDS[kind], int(index), _ptr(out), out.size),
                 "deme_download_wildcard_array")
        return out

    def set_timing(self, enable=True):
        """True / 1: time every launch; n > 1: every n-th launch of each timed kernel; False / 0: off"""
        self._ck(self.lib.deme_set_timing(self.h, int(enable)), "deme_set_timing")

    def kernel_time_reset(self):
        self._ck(self.lib.deme_kernel_time_reset(self.h), "deme_kernel_time_reset")

    def kernel_time_ms(self, name):
        ms = C.c_double()
        n = C.c_uint64()
        self._ck(self.lib.deme_kernel_time_ms(self.h, name.encode(), C.byref(ms), C.byref(n)), "deme_kernel_time_ms")
        return ms.value, n.value

    def halo_pack(self, d_ids_ptr, n, d_buf_ptr):
        self._ck(self.lib.deme_halo_pack(self.h, d_ids_ptr, int(n), d_buf_ptr), "deme_halo_pack")

    def halo_unpack(self, d_ids_ptr, n, d_buf_ptr):
        self._ck(self.lib.deme_halo_unpack(self.h, d_ids_ptr, int(n), d_buf_ptr), "deme_halo_unpack")
