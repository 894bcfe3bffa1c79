"""ctypes wrapper of the CPU oracle (oracle/libdeme_oracle.so) and, when present,
the reference-helper build (oracle/_ref/libdeme_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_P = C.c_void_p
FORCE_NF = 39

_orc = None
_ref = None
_variant = ""  # "" = the parity build; "_perf" = libdeme_oracle_perf.so, the build bench.py times (oracle/Makefile)
_libs = {}


def set_variant(perf):
    """Which build new simulations use: the parity build (default; what the tests compare the HIP path with) or the -O3 build
    with its list building, sorts and accumulation spread over the OpenMP team (bench.py's cpu_baseline only)."""
    global _variant, _orc
    _variant = "_perf" if perf else ""
    _orc = _libs.get(_variant)


def build_perf_native():
    """rebuild the timed variant with -march=native on THIS host (the shipped file is generic AVX2); False if there is no compiler"""
    try:
        subprocess.check_call(["make", "-C", _HERE, "perf-native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return True
    except Exception:  # noqa: BLE001
        return False


def build(force=False):
    """Compile the oracle (and oracle/_ref when the reference tree is here)."""
    so = os.path.join(_HERE, "libdeme_oracle.so")
    src = os.path.join(_HERE, "deme_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libdeme_oracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libdeme_ref.so")
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(ref_so) or
                                                 os.path.getmtime(ref_so) < os.path.getmtime(
                                                     os.path.join(_HERE, "ref_glue.cpp"))):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def lib():
    global _orc
    if _orc is None:
        build()
        if _variant and not os.path.exists(os.path.join(_HERE, f"libdeme_oracle{_variant}.so")):
            subprocess.check_call(["make", "-C", _HERE, f"libdeme_oracle{_variant}.so"], stdout=subprocess.DEVNULL)
        _orc = _libs[_variant] = C.CDLL(os.path.join(_HERE, f"libdeme_oracle{_variant}.so"))
        _orc.orc_sim_create.restype = _P
        _orc.orc_sim_create.argtypes = [_P, _P]
        for n in ("orc_sim_destroy", "orc_sim_set_params", "orc_sim_set_margins", "orc_sim_compute_margins",
                  "orc_sim_get_margins", "orc_sim_migrate", "orc_sim_calc_forces", "orc_sim_integrate",
                  "orc_sim_counts", "orc_sim_get_state", "orc_sim_set_state", "orc_sim_set_wildcard"):
            getattr(_orc, n).restype = None
        for n in ("orc_sim_get_incidence", "orc_sim_get_contacts", "orc_sim_get_wildcard", "orc_sim_get_records",
                  "orc_sim_get_sphere_geometry"):
            getattr(_orc, n).restype = C.c_size_t
        _orc.orc_sim_detect.restype = C.c_int
        _orc.orc_sim_step.restype = C.c_int
    return _orc


INSPECT_CODES = {"clump_max_z": 0, "clump_min_z": 1, "clump_max_absv": 2, "clump_mass": 3, "max_absv": 4,
                 "clump_kinetic_energy": 5, "absv": 6, "clump_volume": 7}


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libdeme_ref.so"))


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.path.join(_HERE, "_ref", "libdeme_ref.so"))
    return _ref


def _p(a):
    return C.c_void_p(a.ctypes.data)


# ---------------------------------------------------------------------------
# element-wise functions; `which` = "orc" (restatement) or "ref" (reference build)
# ---------------------------------------------------------------------------
def _pick(which):
    return (lib(), "orc_el_") if which == "orc" else (ref(), "ref_")


def decode(which, vid, sx, sy, sz, nvXp2, nvYp2, voxel, l):
    L, pre = _pick(which)
    n = len(vid)
    X, Y, Z = (np.zeros(n) for _ in range(3))
    getattr(L, pre + "decode")(C.c_size_t(n), _p(vid), _p(sx), _p(sy), _p(sz), C.c_uint(nvXp2), C.c_uint(nvYp2),
                               C.c_double(voxel), C.c_double(l), _p(X), _p(Y), _p(Z))
    return X, Y, Z


def encode(which, X, Y, Z, nvXp2, nvYp2, voxel, l):
    L, pre = _pick(which)
    n = len(X)
    vid = np.zeros(n, np.uint64)
    sx, sy, sz = (np.zeros(n, np.uint16) for _ in range(3))
    getattr(L, pre + "encode")(C.c_size_t(n), _p(X), _p(Y), _p(Z), C.c_uint(nvXp2), C.c_uint(nvYp2),
                               C.c_double(voxel), C.c_double(l), _p(vid), _p(sx), _p(sy), _p(sz))
    return vid, sx, sy, sz


def rotate(which, v, q_wxyz):
    L, pre = _pick(which)
    x, y, z = (np.ascontiguousarray(v[:, k], np.float32).copy() for k in range(3))
    qs = [np.ascontiguousarray(q_wxyz[:, k], np.float32) for k in range(4)]
    getattr(L, pre + "rotate")(C.c_size_t(len(x)), _p(x), _p(y), _p(z), *[_p(q) for q in qs])
    return np.stack([x, y, z], 1)


def rotate_d(which, v, q_wxyz):
    L, pre = _pick(which)
    x, y, z = (np.ascontiguousarray(v[:, k], np.float64).copy() for k in range(3))
    qs = [np.ascontiguousarray(q_wxyz[:, k], np.float32) for k in range(4)]
    getattr(L, pre + "rotate_d")(C.c_size_t(len(x)), _p(x), _p(y), _p(z), *[_p(q) for q in qs])
    return np.stack([x, y, z], 1)


def hamilton(which, q1, q2):
    L, pre = _pick(which)
    q1 = np.ascontiguousarray(q1, np.float32)
    q2 = np.ascontiguousarray(q2, np.float32)
    out = np.zeros_like(q1)
    getattr(L, pre + "hamilton")(C.c_size_t(len(q1)), _p(q1), _p(q2), _p(out))
    return out


def mask_pair(which, i, j):
    L, pre = _pick(which)
    i = np.ascontiguousarray(i, np.uint32)
    j = np.ascontiguousarray(j, np.uint32)
    out = np.zeros(len(i), np.uint32)
    getattr(L, pre + "mask_pair")(C.c_size_t(len(i)), _p(i), _p(j), _p(out))
    return out


def point_bin(which, X, Y, Z, bin_size, nbX, nbY):
    L, pre = _pick(which)
    out = np.zeros(len(X), np.uint32)
    getattr(L, pre + "point_bin")(C.c_size_t(len(X)), _p(X), _p(Y), _p(Z), C.c_double(bin_size), C.c_uint32(nbX),
                                  C.c_uint32(nbY), _p(out))
    return out


def spheres_overlap(which, A, rA, B, rB):
    L, pre = _pick(which)
    n = len(rA)
    A = np.ascontiguousarray(A, np.float64)
    B = np.ascontiguousarray(B, np.float64)
    rA = np.ascontiguousarray(rA, np.float64)
    rB = np.ascontiguousarray(rB, np.float64)
    t = np.zeros(n, np.uint8)
    CP = np.zeros((n, 3))
    nrm = np.zeros((n, 3), np.float32)
    d = np.zeros(n)
    getattr(L, pre + "spheres_overlap")(C.c_size_t(n), _p(A), _p(rA), _p(B), _p(rB), _p(t), _p(CP), _p(nrm), _p(d))
    return t, CP, nrm, d


def sphere_entity(which, A, radA, typeB, B, dirB, size1, nsign, beta):
    L, pre = _pick(which)
    n = len(radA)
    A = np.ascontiguousarray(A, np.float64)
    B = np.ascontiguousarray(B, np.float64)
    radA = np.ascontiguousarray(radA, np.float32)
    typeB = np.ascontiguousarray(typeB, np.uint8)
    dirB = np.ascontiguousarray(dirB, np.float32)
    size1 = np.ascontiguousarray(size1, np.float32)
    nsign = np.ascontiguousarray(nsign, np.float32)
    beta = np.ascontiguousarray(beta, np.float32)
    t = np.zeros(n, np.uint8)
    CP = np.zeros((n, 3))
    nrm = np.zeros((n, 3), np.float32)
    d = np.zeros(n)
    getattr(L, pre + "sphere_entity")(C.c_size_t(n), _p(A), _p(radA), _p(typeB), _p(B), _p(dirB), _p(size1), _p(nsign),
                                      _p(beta), _p(t), _p(CP), _p(nrm), _p(d))
    return t, CP, nrm, d


def mat_proxy(which, Y1, nu1, Y2, nu2):
    L, pre = _pick(which)
    arrs = [np.ascontiguousarray(a, np.float32) for a in (Y1, nu1, Y2, nu2)]
    E = np.zeros(len(arrs[0]), np.float32)
    G = np.zeros(len(arrs[0]), np.float32)
    getattr(L, pre + "mat_proxy")(C.c_size_t(len(E)), *[_p(a) for a in arrs], _p(E), _p(G))
    return E, G


def force(which, model, depth, fin, mu, Crr, hist):
    """fin: (n, 39) float32 (layout in oracle/deme_oracle.cpp); hist: (n,4) in -> returns (hist_out, out(n,6))."""
    L, pre = _pick(which)
    depth = np.ascontiguousarray(depth, np.float64)
    fin = np.ascontiguousarray(fin, np.float32)
    mu = np.ascontiguousarray(mu, np.float32)
    Crr = np.ascontiguousarray(Crr, np.float32)
    h = np.ascontiguousarray(hist, np.float32).copy()
    out = np.zeros((len(depth), 6), np.float32)
    getattr(L, pre + "force")(C.c_size_t(len(depth)), C.c_int(model), _p(depth), _p(fin), _p(mu), _p(Crr), _p(h),
                              _p(out))
    return h, out


# ---------------------------------------------------------------------------
# simulation object (mirrors dem_engine_amd.abi.Context method for method)
# ---------------------------------------------------------------------------
class OracleSim:
    def __init__(self, params, scene, state_dtypes, counts_cls, state_factory):
        """params/scene: the same ctypes structs the C-ABI takes."""
        self.L = lib()
        self.h = self.L.orc_sim_create(C.byref(params), C.byref(scene))
        self.n_owners = int(scene.nOwners)
        self.n_spheres = int(scene.nSpheres)
        self._counts_cls = counts_cls
        self._state_factory = state_factory
        self.params = params

    def close(self):
        if self.h:
            self.L.orc_sim_destroy(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, p):
        self.L.orc_sim_set_params(C.c_void_p(self.h), C.byref(p))

    def set_custom_model(self, kind, cohesion):
        """DEME_FORCE_CUSTOM in parametric form; kind 1 = frictionless Hertz + pairwise cohesion + a contact-age wildcard
        (the fragment of bench.py's configs[4] flavour); cohesion: nMat x nMat"""
        c = np.ascontiguousarray(cohesion, np.float32).reshape(-1)
        self.L.orc_sim_set_custom_model.restype = None
        self.L.orc_sim_set_custom_model(C.c_void_p(self.h), C.c_int(int(kind)), _p(c), C.c_size_t(c.size))

    def set_margins(self, m):
        m = np.ascontiguousarray(m, np.float32)
        self.L.orc_sim_set_margins(C.c_void_p(self.h), _p(m))

    def compute_margins(self, drift):
        self.L.orc_sim_compute_margins(C.c_void_p(self.h), C.c_uint32(drift))

    def margins(self):
        m = np.zeros(self.n_owners, np.float32)
        self.L.orc_sim_get_margins(C.c_void_p(self.h), _p(m))
        return m

    def detect(self):
        rc = self.L.orc_sim_detect(C.c_void_p(self.h))
        if rc:
            raise RuntimeError(f"oracle detect status {rc}")

    def migrate(self):
        self.L.orc_sim_migrate(C.c_void_p(self.h))

    def calc_forces(self, record=False):
        self.L.orc_sim_calc_forces(C.c_void_p(self.h), C.c_int(int(record)))

    def integrate(self):
        self.L.orc_sim_integrate(C.c_void_p(self.h))

    def step(self, n):
        rc = self.L.orc_sim_step(C.c_void_p(self.h), C.c_uint32(n))
        if rc:
            raise RuntimeError(f"oracle step status {rc}")

    def counts(self):
        c = self._counts_cls()
        self.L.orc_sim_counts(C.c_void_p(self.h), C.byref(c))
        return c

    def download_state(self):
        st, out = self._state_factory(self.n_owners)
        self.L.orc_sim_get_state(C.c_void_p(self.h), C.byref(st))
        return out

    def upload_state(self, arrays):
        st, _ = self._state_factory(self.n_owners, arrays)
        self.L.orc_sim_set_state(C.c_void_p(self.h), C.byref(st))

    def add_family_rule(self, frm, to, quantity, op, threshold):
        self.L.orc_sim_add_family_rule(C.c_void_p(self.h), C.c_uint32(frm), C.c_uint32(to), C.c_uint32(quantity), C.c_uint32(op),
                                       C.c_double(threshold))

    def change_family(self, frm, to):
        self.L.orc_sim_change_family(C.c_void_p(self.h), C.c_uint32(frm), C.c_uint32(to))

    def mark_persistent_contacts(self, mode=0, n1=0, n2=0, mark=True):
        self.L.orc_sim_mark_persistent(C.c_void_p(self.h), C.c_int(mode), C.c_uint32(n1), C.c_uint32(n2), C.c_int(int(bool(mark))))

    def num_persistent_contacts(self):
        self.L.orc_sim_num_persistent.restype = C.c_size_t
        return int(self.L.orc_sim_num_persistent(C.c_void_p(self.h)))

    def add_owner_acc(self, owner, acc=None, ang_acc=None):
        a = None if acc is None else np.ascontiguousarray(acc, np.float32).reshape(-1, 3)
        l = None if ang_acc is None else np.ascontiguousarray(ang_acc, np.float32).reshape(-1, 3)
        n = len(a) if a is not None else len(l)
        self.L.orc_sim_add_owner_acc(C.c_void_p(self.h), C.c_uint32(owner), C.c_uint32(n), None if a is None else _p(a),
                                     None if l is None else _p(l))

    def persistent_contacts(self):
        n = self.num_persistent_contacts()
        a, b, t = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
        self.L.orc_sim_get_persistent(C.c_void_p(self.h), _p(a), _p(b), _p(t))
        return a, b, t

    def set_persistent_contacts(self, idA, idB, ctype):
        a, b, t = np.ascontiguousarray(idA, np.uint32), np.ascontiguousarray(idB, np.uint32), np.ascontiguousarray(ctype, np.uint8)
        self.L.orc_sim_set_persistent(C.c_void_p(self.h), _p(a), _p(b), _p(t), C.c_size_t(len(a)))

    def set_prescription(self, family, has, flags, coef):
        c = np.ascontiguousarray(coef, np.float32).reshape(15, 4)
        self.L.orc_sim_set_prescription(C.c_void_p(self.h), C.c_uint32(family), C.c_uint32(has), C.c_uint32(flags), _p(c))

    def seed_contacts(self, idA, idB, ctype, wildcards=None):
        a = np.ascontiguousarray(idA, np.uint32)
        b = np.ascontiguousarray(idB, np.uint32)
        t = np.ascontiguousarray(ctype, np.uint8)
        w = np.ascontiguousarray(wildcards if wildcards is not None else np.zeros((len(a), 0)), np.float32)
        self.L.orc_sim_seed_contacts.restype = C.c_int
        rc = self.L.orc_sim_seed_contacts(C.c_void_p(self.h), _p(a), _p(b), _p(t), _p(w), C.c_size_t(len(a)))
        assert rc == 0, "duplicate contact pair in the seed list"

    def inspect(self, quantity, values=False, box=None):
        """box: ((xlo, ylo, zlo), (xhi, yhi, zhi)) region in the test form of the reference's region string"""
        q = INSPECT_CODES[quantity] if isinstance(quantity, str) else int(quantity)
        n = self.n_spheres if q <= 2 else self.n_owners
        red = C.c_float(0)
        vals = np.zeros(n, np.float32) if values else None
        lo = hi = None
        if box is not None:
            lo, hi = (np.ascontiguousarray(b, np.float32) for b in box)
        self.L.orc_sim_inspect_region.restype = C.c_size_t
        self.L.orc_sim_inspect_region(C.c_void_p(self.h), C.c_uint32(q), None if lo is None else _p(lo), None if hi is None else _p(hi),
                                      C.byref(red), None if vals is None else _p(vals))
        return vals if values else float(red.value)

    def set_volumes(self, volumes):
        v = np.ascontiguousarray(volumes, np.float32)
        self.L.orc_sim_set_volumes(C.c_void_p(self.h), _p(v), C.c_size_t(v.size))

    def bin_incidence(self):
        n = int(self.counts().nBinSphereTouches)
        b = np.zeros(n, np.uint32)
        s = np.zeros(n, np.uint32)
        self.L.orc_sim_get_incidence(C.c_void_p(self.h), _p(b), _p(s), C.c_size_t(n))
        return b, s

    def contacts(self):
        n = int(self.counts().nContacts)
        a, b, m = (np.zeros(n, np.uint32) for _ in range(3))
        t = np.zeros(n, np.uint8)
        self.L.orc_sim_get_contacts(C.c_void_p(self.h), _p(a), _p(b), _p(t), _p(m), C.c_size_t(n))
        return a, b, t, m

    def wildcard(self, w):
        n = int(self.counts().nContacts)
        out = np.zeros(n, np.float32)
        self.L.orc_sim_get_wildcard(C.c_void_p(self.h), C.c_uint32(w), _p(out), C.c_size_t(n))
        return out

    def set_wildcard(self, w, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        self.L.orc_sim_set_wildcard(C.c_void_p(self.h), C.c_uint32(w), _p(arr), C.c_size_t(arr.size))

    def contact_records(self):
        n = int(self.counts().nContacts)
        arrs = [np.zeros((n, 3), np.float32) for _ in range(4)]
        self.L.orc_sim_get_records(C.c_void_p(self.h), *[_p(a) for a in arrs], C.c_size_t(n))
        return arrs

    def update_tri_nodes(self, n1, n2, n3):
        arrs = [np.ascontiguousarray(x, np.float32).reshape(-1) for x in (n1, n2, n3)]
        self.L.orc_sim_set_tri_nodes(C.c_void_p(self.h), *[_p(a) for a in arrs])

    def sphere_geometry(self):
        n = self.n_spheres
        X, Y, Z = (np.zeros(n) for _ in range(3))
        R = np.zeros(n, np.float32)
        self.L.orc_sim_get_sphere_geometry(C.c_void_p(self.h), _p(X), _p(Y), _p(Z), _p(R), C.c_size_t(n))
        return X, Y, Z, R


def make_sim(pkg, p, sc):
    """Oracle twin of a dem_engine_amd.Context built from the same params/scene structs."""
    return OracleSim(p, sc, pkg.abi.STATE_DTYPES, pkg.DemeCounts, pkg.abi.make_state_struct)


def rcp_mul(x, y, fenv=False):
    x = np.ascontiguousarray(x, np.float64)
    y = np.ascontiguousarray(y, np.float64)
    r, m = np.zeros_like(x), np.zeros_like(x)
    fn = lib().orc_el_rcp_mul_fenv if fenv else lib().orc_el_rcp_mul_ru
    fn(C.c_size_t(len(x)), _p(x), _p(y), _p(r), _p(m))
    return r, m


def tri_sphere(A, B, Cc, P, r, directional=False):
    arrs = [np.ascontiguousarray(a, np.float64) for a in (A, B, Cc, P, r)]
    n = len(arrs[4])
    hit = np.zeros(n, np.uint8)
    nr, pt = np.zeros((n, 3)), np.zeros((n, 3))
    d = np.zeros(n)
    lib().orc_el_tri_sphere(C.c_size_t(n), C.c_int(int(directional)), *[_p(a) for a in arrs], _p(hit), _p(nr), _p(d), _p(pt))
    return hit, nr, d, pt


def tri_box(center, half, A, B, Cc, which="orc"):
    L, pre = _pick(which)
    arrs = [np.ascontiguousarray(a, np.float32) for a in (center, half, A, B, Cc)]
    out = np.zeros(len(arrs[1]), np.uint8)
    getattr(L, pre + "tri_box")(C.c_size_t(len(out)), *[_p(a) for a in arrs], _p(out))
    return out


def tri_bbox(A, B, Cc, binSize, nb, which="orc"):
    L_, pre = _pick(which)
    arrs = [np.ascontiguousarray(a, np.float32) for a in (A, B, Cc)]
    n = len(arrs[0])
    lo, hi = np.zeros((n, 3), np.int32), np.zeros((n, 3), np.int32)
    getattr(L_, pre + "tri_bbox")(C.c_size_t(n), *[_p(a) for a in arrs], C.c_double(binSize), C.c_uint32(nb[0]), C.c_uint32(nb[1]),
                                  C.c_uint32(nb[2]), _p(lo), _p(hi))
    return lo, hi


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(C.c_int(n))
