"""Host-side model builder: the set-up half of ``deme::DEMSolver`` restated in numpy.

Only the calls that define kernel inputs are mirrored (names and argument
meaning follow DEM/API.h); everything else in the reference's API (writers,
trackers, inspectors ...) is out of scope for the hot path.

Reference behaviour followed here (paths relative to the reference's src/):
  * voxel bit split and length unit `l`   DEM/APIPrivate.cpp:373-487 (figureOutNV)
  * bin size / bin counts                 DEM/APIPrivate.cpp:489-566, DEM/HostSideHelpers.hpp:195-207
  * world bounding planes                 DEM/APIPrivate.cpp:955-1014
  * owner / sphere flattening order       DEM/dT.cpp:700-800, DEM/kT.cpp:766-832
  * mass-property table order             DEM/APIPrivate.cpp:1802-1819 (clump templates, analytical, meshes)
  * pairwise material matrices            DEM/APIPrivate.cpp:1877-2026 (off-diagonal default = mean)
  * family mask indexing                  kernel/DEMHelperKernels.cuh:57-62
"""
import math

import numpy as np

from . import abi

VOXEL_RES_POWER2 = 16
VOXEL_COUNT_POWER2 = 64
DEFAULT_BOX_DOMAIN_ENLARGE_RATIO = 0.2
RESERVED_FAMILY_NUM = 255

# data/clumps/3_clump.csv of the reference (three spheres; x,y,z,r) -- a data fixture
THREE_SPHERE_CLUMP = np.array([[0.5, 0.341729, 0.0, 0.8],
                               [0.0, -0.658271, 0.0, 0.8],
                               [-0.5, 0.341729, 0.0, 0.8]], dtype=np.float32)
THREE_SPHERE_CLUMP_VOLUME = 5.5886717
THREE_SPHERE_CLUMP_MOI_MIXER = (2.928, 2.6029, 3.9908)  # DEMdemo_Mixer.cpp:62-67


def mask_pair(i, j):
    if i > j:
        i, j = j, i
    return (1 + j) * j // 2 + i


def encode_positions(xyz_shifted, nvXp2, nvYp2, voxel_size, l):
    """positionToVoxelID (kernel/DEMHelperKernels.cuh:138-159) on an (n,3) float64 array."""
    X = np.asarray(xyz_shifted, dtype=np.float64)
    n = np.floor(X / voxel_size).astype(np.uint64)  # positions are >= 0: truncation == floor
    sub = ((X - n.astype(np.float64) * voxel_size) / l).astype(np.uint16)
    vid = n[:, 0] + (n[:, 1] << np.uint64(nvXp2)) + (n[:, 2] << np.uint64(nvXp2 + nvYp2))
    return vid.astype(np.uint64), sub[:, 0].copy(), sub[:, 1].copy(), sub[:, 2].copy()


def decode_positions(vid, lx, ly, lz, nvXp2, nvYp2, voxel_size, l):
    vid = np.asarray(vid, dtype=np.uint64)
    vx = vid & np.uint64((1 << nvXp2) - 1)
    vy = (vid >> np.uint64(nvXp2)) & np.uint64((1 << nvYp2) - 1)
    vz = vid >> np.uint64(nvXp2 + nvYp2)
    X = vx.astype(np.float64) * voxel_size + np.asarray(lx, np.float64) * l
    Y = vy.astype(np.float64) * voxel_size + np.asarray(ly, np.float64) * l
    Z = vz.astype(np.float64) * voxel_size + np.asarray(lz, np.float64) * l
    return np.stack([X, Y, Z], axis=1)


class ClumpTemplate:
    def __init__(self, mass, moi, radii, relpos, materials):
        self.mass = float(mass)
        self.moi = tuple(float(x) for x in moi)
        self.radii = np.asarray(radii, np.float32).reshape(-1)
        self.relpos = np.asarray(relpos, np.float32).reshape(-1, 3)
        self.materials = list(materials)
        self.mark = None
        self.volume = 0.0  # SetVolume (DEMClumpTemplate::volume, read by the "clump_volume" inspector)
        self.name = None  # AssignName (Structs.h:697); default "%04d" of the load order (APIPublic.cpp:1751-1755)

    def AssignName(self, name):
        self.name = str(name)
        return self

    def SetVolume(self, v):
        self.volume = float(v)
        return self

    def Scale(self, s):
        """DEMClumpTemplate::Scale (DEM/Structs.h): lengths*s, mass*s^3, MOI*s^5."""
        # the reference's types (Structs.h:682-695): s is a float; mass and volume (float) *= a double; MOI (float3) *= that double
        # converted to float; lengths (float) *= s
        s32 = np.float32(s)
        ps = float(abs(s32))
        self.mass = float(np.float32(float(np.float32(self.mass)) * (ps * ps * ps)))
        self.volume = float(np.float32(float(np.float32(self.volume)) * (ps * ps * ps)))
        s5 = np.float32(ps * ps * ps * ps * ps)
        self.moi = tuple(float(np.float32(m) * s5) for m in self.moi)
        self.radii = (self.radii * s32).astype(np.float32)
        self.relpos = (self.relpos * s32).astype(np.float32)
        return self


class ClumpBatch:
    def __init__(self, templates, xyz):
        self.templates = templates
        self.xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
        n = len(self.xyz)
        self.vel = np.zeros((n, 3), np.float32)
        self.angvel = np.zeros((n, 3), np.float32)
        self.oriq = np.tile(np.array([0, 0, 0, 1], np.float32), (n, 1))  # x y z w (float4 order)
        self.family = np.zeros(n, np.uint8)

    def SetVel(self, v):
        self.vel[:] = np.asarray(v, np.float32)

    def SetAngVel(self, w):
        self.angvel[:] = np.asarray(w, np.float32)

    def SetOriQ(self, q_xyzw):
        self.oriq[:] = np.asarray(q_xyzw, np.float32)

    def SetFamily(self, f):
        self.family[:] = np.asarray(f, np.uint8)


class ExternObj:
    def __init__(self):
        self.comps = []  # (type, pos, rot/dir, size1, size2, size3, normal_sign, material)
        self.family = RESERVED_FAMILY_NUM
        self.init_pos = (0.0, 0.0, 0.0)
        self.init_oriq = (0.0, 0.0, 0.0, 1.0)  # x y z w
        self.mass = 1e6
        self.moi = (1e6, 1e6, 1e6)

    def AddPlane(self, pos, normal, material):
        n = np.asarray(normal, np.float32)
        inv = np.float32(1.0) / np.sqrt(np.float32(np.dot(n, n)))  # normalize(): rsqrtf * v
        n = (n * inv).astype(np.float32)
        self.comps.append((0, tuple(np.float32(pos)), tuple(n), 0.0, 0.0, 0.0, 1.0, material))

    def AddCylinder(self, pos, axis, rad, material, normal_inward=True):
        a = np.asarray(axis, np.float32)
        inv = np.float32(1.0) / np.sqrt(np.float32(np.dot(a, a)))
        a = (a * inv).astype(np.float32)
        self.comps.append((2, tuple(np.float32(pos)), tuple(a), float(rad), 0.0, 0.0, 1.0 if normal_inward else -1.0,
                           material))

    def SetFamily(self, f):
        self.family = int(f)

    def SetInitPos(self, p):
        self.init_pos = tuple(float(x) for x in p)

    def SetMass(self, m):
        self.mass = float(m)


class MeshObj:
    """DEMMeshConnected (DEM/BdrsAndObjs.h): vertices + triangle index triples, owner-local frame."""

    def __init__(self, vertices, faces, material):
        self.vertices = np.asarray(vertices, np.float32).reshape(-1, 3)
        self.faces = np.asarray(faces, np.int64).reshape(-1, 3)
        self.material = material
        self.family = RESERVED_FAMILY_NUM
        self.init_pos = (0.0, 0.0, 0.0)
        self.init_oriq = (0.0, 0.0, 0.0, 1.0)  # x y z w
        self.mass = 1.0
        self.moi = (1.0, 1.0, 1.0)
        self.vel = (0.0, 0.0, 0.0)

    def Scale(self, s):
        self.vertices = (self.vertices * np.asarray(s, np.float32)).astype(np.float32)
        return self

    def SetInitPos(self, p):
        self.init_pos = tuple(float(x) for x in p)

    def SetInitQuat(self, q_xyzw):
        self.init_oriq = tuple(float(x) for x in q_xyzw)

    def SetMass(self, m):
        self.mass = float(m)

    def SetMOI(self, moi):
        self.moi = tuple(float(x) for x in moi)

    def SetFamily(self, f):
        self.family = int(f)

    def GetNumTriangles(self):
        return len(self.faces)

    def nodes(self):
        v = self.vertices
        return v[self.faces[:, 0]], v[self.faces[:, 1]], v[self.faces[:, 2]]


class SceneBuilder:
    """Python mirror of the DEMSolver set-up surface needed by the hot path."""

    def __init__(self):
        self.materials = []
        self.pair_overrides = {}
        self.templates = []
        self.batches = []
        self.ext_objs = []
        self.meshes = []
        self.user_box_min = np.array([-10, -10, -10], np.float32)
        self.user_box_max = np.array([10, 10, 10], np.float32)
        self.target_box_min = self.user_box_min * np.float32(1.2)
        self.target_box_max = self.user_box_max * np.float32(1.2)
        self.bounding_bc = "none"
        self.bounding_mat = None
        self.h = 1e-5
        self.G = (0.0, 0.0, -9.81)
        self.cd_update_freq = 20
        # initial bin size (API.h:140-153, 1403-1412): explicit, a multiple of the smallest radius, or -- the default -- whatever
        # gives about m_target_init_bin_num bins, found by the loop of APIPrivate.cpp:525-541 starting from the multiple
        self.bin_size = None
        self.bin_multiple = 8.0
        self.target_bin_num = 1000000
        self.expand_factor = 0.0
        self.safety_multi = 1.0   # API.h:1481 m_expand_safety_multi
        self.safety_adder = 3.0   # API.h:1484 m_expand_base_vel: 3 m/s on top of every owner's speed when sizing the margins
        self.approx_max_vel = 1e15
        self.err_out_vel = 1e15
        self.err_out_bin_sph = 32768
        self.integrator = abi.INTEGRATOR_EXTENDED_TAYLOR  # API.h:1556 default
        self.force_model = abi.FORCE_HERTZIAN
        self.family_masks = np.zeros(abi.FAMILY_MASK_ENTRIES, np.uint8)
        self.family_extra = np.zeros(abi.NUM_FAMILIES, np.float32)
        self.family_flags = np.zeros(abi.NUM_FAMILIES, np.uint8)
        self.family_flags[RESERVED_FAMILY_NUM] = abi.FAMILY_FIXED  # APIPrivate.cpp:1351
        self.n_custom_wildcards = 0

    # ---- materials ---------------------------------------------------------
    def LoadMaterial(self, props):
        self.materials.append(dict(props))
        return len(self.materials) - 1

    def SetMaterialPropertyPair(self, name, m1, m2, val):
        self.pair_overrides[(name, m1, m2)] = float(val)
        self.pair_overrides[(name, m2, m1)] = float(val)

    # ---- templates ---------------------------------------------------------
    def LoadClumpType(self, mass, moi, radii, relpos, material):
        radii = np.asarray(radii, np.float32).reshape(-1)
        mats = [material] * len(radii) if np.isscalar(material) else list(material)
        t = ClumpTemplate(mass, moi, radii, relpos, mats)
        self.templates.append(t)
        return t

    def LoadSphereType(self, mass, radius, material):
        moi = 2.0 / 5.0 * mass * radius * radius
        return self.LoadClumpType(mass, (moi, moi, moi), [radius], [[0, 0, 0]], material)

    def LoadThreeSphereClump(self, scale, density, material, moi_unit=THREE_SPHERE_CLUMP_MOI_MIXER):
        """data/clumps/3_clump.csv as the Mixer demo loads it (DEMdemo_Mixer.cpp:62-67)."""
        t = self.LoadClumpType(density * THREE_SPHERE_CLUMP_VOLUME, tuple(m * density for m in moi_unit),
                               THREE_SPHERE_CLUMP[:, 3], THREE_SPHERE_CLUMP[:, :3], material)
        return t.Scale(scale)

    # ---- domain --------------------------------------------------------------
    def InstructBoxDomainDimension(self, x, y, z):
        """x,y,z: sizes (centred box) or (lo, hi) pairs.  APIPublic.cpp:845-904."""
        def rng(v):
            if np.isscalar(v):
                return (-float(v) / 2.0, float(v) / 2.0)
            return (float(min(v)), float(max(v)))
        r = [rng(x), rng(y), rng(z)]
        self.user_box_min = np.array([a for a, _ in r], np.float32)
        self.user_box_max = np.array([b for _, b in r], np.float32)
        # float arithmetic like the reference's float3 members: |hi - lo| * (0.2f / 2) (APIPublic.cpp:880-885)
        enlarge = (self.user_box_max - self.user_box_min) * np.float32(np.float32(DEFAULT_BOX_DOMAIN_ENLARGE_RATIO) / np.float32(2.0))
        self.target_box_min = (self.user_box_min - enlarge).astype(np.float32)
        self.target_box_max = (self.user_box_max + enlarge).astype(np.float32)

    def InstructBoxDomainBoundingBC(self, inst, material):
        self.bounding_bc = inst
        self.bounding_mat = material

    # ---- entities --------------------------------------------------------------
    def AddClumps(self, templates, xyz):
        xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
        if isinstance(templates, ClumpTemplate):
            templates = [templates] * len(xyz)
        b = ClumpBatch(list(templates), xyz)
        self.batches.append(b)
        return b

    def SortClumpsSpatially(self, cell=None):
        """Reorder the clumps of every batch along a Z-order curve (cells of edge `cell`, default 4 x the largest
        component radius).  No reference equivalent: owner ids follow the load order, and the engine's gathers (B owner in
        the force kernel, B-side contributions in the integrator) run ~28 % slower when neighbours in space are far apart
        in memory (measured: bench.py --order random vs lattice / Morton order).  Call before Initialize(); ids change."""
        rmax = max((float(t.radii.max()) + float(np.abs(t.relpos).max()) for t in self.templates), default=1.0)
        cell = float(cell) if cell else 4.0 * rmax
        for b in self.batches:
            order = np.argsort(morton_codes(b.xyz, cell), kind="stable")
            b.xyz, b.vel, b.angvel, b.oriq, b.family = b.xyz[order], b.vel[order], b.angvel[order], b.oriq[order], b.family[order]
            b.templates = [b.templates[i] for i in order]
        return self

    def AddExternalObject(self):
        o = ExternObj()
        self.ext_objs.append(o)
        return o

    def AddMeshObject(self, vertices, faces, material):
        """AddWavefrontMeshObject without the OBJ reader (API.h:638-645): vertices (n,3), faces (m,3)."""
        m = MeshObj(vertices, faces, material)
        self.meshes.append(m)
        return m

    def AddBCPlane(self, pos, normal, material):
        o = self.AddExternalObject()
        o.AddPlane(pos, normal, material)
        return o

    # ---- solver knobs -----------------------------------------------------------
    def SetInitTimeStep(self, h):
        self.h = float(h)

    def SetGravitationalAcceleration(self, g):
        self.G = tuple(float(x) for x in g)

    def SetCDUpdateFreq(self, k):
        self.cd_update_freq = int(k)

    def SetInitBinSize(self, s):
        self.bin_size = float(s)
        self.target_bin_num = None

    def SetInitBinSizeAsMultipleOfSmallestSphere(self, m):
        self.bin_multiple = float(np.float32(m))
        self.bin_size = None
        self.target_bin_num = None

    def SetInitBinNumTarget(self, n):
        self.target_bin_num = int(n)
        self.bin_size = None

    def SetExpandFactor(self, beta):
        self.expand_factor = float(beta)

    def SetExpandSafetyMultiplier(self, m):
        self.safety_multi = float(m)

    def SetExpandSafetyAdder(self, a):
        self.safety_adder = float(a)

    def SetMaxVelocity(self, v):
        self.approx_max_vel = float(v)

    def SetErrorOutVelocity(self, v):
        self.err_out_vel = float(v)

    def SetIntegrator(self, which):
        key = which.upper() if isinstance(which, str) else which  # the reference takes the names in lower case (API.h:124)
        self.integrator = {"FORWARD_EULER": 0, "CENTERED_DIFFERENCE": 1, "EXTENDED_TAYLOR": 2}.get(key, key)
        if not isinstance(self.integrator, int):
            raise ValueError(f"Integration type {which} is unknown")

    def UseFrictionalHertzianModel(self):
        self.force_model = abi.FORCE_HERTZIAN

    def UseFrictionlessHertzianModel(self):
        self.force_model = abi.FORCE_HERTZIAN_FRICTIONLESS

    # ---- user force model (DEMForceModel, AuxClasses.h:422-485) -----------------------------------
    def DefineContactForceModel(self, src):
        """DEMSolver::DefineContactForceModel (APIPublic.cpp:906): a C++ statement block written against the
        reference's ingredient names; compiled at run time through deme_compile_force_model (hipRTC)."""
        self.force_model = abi.FORCE_CUSTOM
        self.force_src = src
        return self

    def ReadContactForceModel(self, path):
        return self.DefineContactForceModel(open(path).read())

    def SetPerOwnerWildcards(self, names):
        """per-owner float arrays a fragment sees as `name`, `name_A`, `name_B` (Models.h:319-342); sorted like the set."""
        self.owner_wildcards = sorted(set(names))

    def SetPerGeometryWildcards(self, names):
        """per-sphere / -triangle / -analytical arrays a fragment sees as `name_A[AGeo]`, `name_B[BGeo]` (Models.h:345-360)."""
        self.geo_wildcards = sorted(set(names))

    def SetPerContactWildcards(self, names):
        """std::set<std::string> in the reference (Models.h:363): indices follow sorted order."""
        self.contact_wildcards = sorted(set(names))
        self.n_custom_wildcards = len(self.contact_wildcards)

    def SetMustPairwiseMatProp(self, names):
        self.pairwise_props = set(names) | {"CoR", "mu", "Crr"}

    def DefineCustomModelPrerequisites(self, code):
        self.user_prerequisites = code

    def force_model_prerequisites(self):
        """_materialDefs_ for the properties beyond E/nu/CoR/mu/Crr + the user's prerequisites text."""
        nm = max(1, len(self.materials))
        extra = sorted({k for m in self.materials for k in m} - {"E", "nu", "CoR", "mu", "Crr"})
        out = []
        lit = lambda v: repr(float(np.float32(v))) + "f"
        for name in extra:
            v = np.array([m.get(name, 0.0) for m in self.materials], np.float32)
            if name in getattr(self, "pairwise_props", set()):
                M = ((v[:, None] + v[None, :]) / np.float32(2.0)).astype(np.float32)
                for i in range(nm):
                    M[i, i] = v[i]
                for (n_, a, b), val in self.pair_overrides.items():
                    if n_ == name:
                        M[a, b] = np.float32(val)
                rows = ", ".join("{" + ", ".join(lit(x) for x in r) + "}" for r in M)
                out.append(f"__device__ const float {name}[][{nm}] = {{{rows}}};")
            else:
                out.append(f"__device__ const float {name}[] = {{{', '.join(lit(x) for x in v)}}};")
        return "\n".join(out) + "\n" + getattr(self, "user_prerequisites", "")

    def compile_into(self, ctx):
        """Call after ctx.upload_scene(): builds the user model and the family prescriptions into the context."""
        if self.force_model == abi.FORCE_CUSTOM:
            ctx.compile_force_model(self.force_src, getattr(self, "contact_wildcards", []), self.force_model_prerequisites(),
                                    getattr(self, "owner_wildcards", []), getattr(self, "geo_wildcards", []))
        if getattr(self, "_presc_inputs", None):
            ctx.compile_prescriptions(*self.prescription_cases())
        if getattr(self, "_family_rules", None):
            ctx.compile_family_rules(self.family_change_rules())
        vols = self.volumes()
        if vols.any():
            ctx.upload_volumes(vols)

    def volumes(self):
        """declared volume per mass-property entry (clump templates first; other owner kinds 0)"""
        v = np.zeros(int(self.counts["nMassProps"]), np.float32)
        tv = np.asarray(getattr(self, "template_volumes", []), np.float32)
        v[:len(tv)] = tv
        return v

    def _user_wildcards(self, ctx):
        """the user model's owner / geometry wildcard arrays as {(kind, slot): array}, read before a scene is uploaded again"""
        if self.force_model != abi.FORCE_CUSTOM:
            return {}
        n = {"owner": int(self.counts["nOwners"]), "sphere": int(self.counts["nSpheres"]), "triangle": int(self.counts.get("nTri", 0)),
             "analytical": int(self.counts["nAnal"])}
        out = {}
        for j in range(len(getattr(self, "owner_wildcards", []))):
            out[("owner", j)] = ctx.wildcard_array("owner", j, n["owner"])
        for j in range(len(getattr(self, "geo_wildcards", []))):
            for kind in ("sphere", "triangle", "analytical"):
                if n[kind]:
                    out[(kind, j)] = ctx.wildcard_array(kind, j, n[kind])
        return out

    def _restore_user_wildcards(self, ctx, saved, owner_dst, sphere_dst):
        """owner_dst / sphere_dst: new index of every old owner / sphere; triangles and analytical components keep theirs"""
        for (kind, j), old in saved.items():
            if kind == "owner":
                new = np.zeros(int(self.counts["nOwners"]), np.float32)
                new[owner_dst] = old
            elif kind == "sphere":
                new = np.zeros(int(self.counts["nSpheres"]), np.float32)
                new[sphere_dst] = old
            else:
                new = old
            ctx.set_wildcard_array(kind, j, new)

    def UpdateClumps(self, ctx, time_elapsed):
        """DEMSolver::UpdateClumps (API.h:1267): clumps added with AddClumps after Initialize() join the running simulation.
        Old owners keep their state, the contact list keeps its history (sphere ids are stable because new clumps are
        appended; analytical-component and triangle ids do not change).  `time_elapsed`: the simulation time to continue
        from (DemeParams.timeElapsed).  Returns the new (params, scene)."""
        old_counts = dict(self.counts)
        st = ctx.download_state()
        saved_wc = self._user_wildcards(ctx)
        cnt = ctx.contacts()
        nW = int(self.params.nContactWildcards)
        W = np.stack([ctx.wildcard(w) for w in range(nW)], 1) if nW else np.zeros((len(cnt[0]), 0), np.float32)
        p, sc = self.Initialize()
        n_old_c, n_old_o = int(old_counts["nOwnerClumps"]), int(old_counts["nOwners"])
        n_new_c = int(self.counts["nOwnerClumps"])
        if n_new_c < n_old_c or int(self.counts["nOwners"]) - n_new_c != n_old_o - n_old_c:
            raise ValueError("UpdateClumps can only append clumps")
        idx_new = np.r_[np.arange(n_old_c), n_new_c + np.arange(n_old_o - n_old_c)]
        for k in ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY",
                  "omgBarZ", "familyID"):
            self.arrays[k][idx_new] = st[k]
        sc = abi.make_scene_struct(self.arrays, self.counts)
        p.timeElapsed = float(time_elapsed)
        self.params = p
        ctx.set_params(p)
        ctx.upload_scene(sc)
        self.compile_into(ctx)
        self._restore_user_wildcards(ctx, saved_wc, idx_new, np.arange(int(old_counts["nSpheres"])))  # new spheres are appended
        if len(cnt[0]):
            ctx.seed_contacts(cnt[0], cnt[1], cnt[2], W)
        return p, sc

    def ResortClumps(self, ctx, time_elapsed, cell=None):
        """Restore spatial order in a RUNNING simulation: the clumps of every batch are renumbered along a Z-order curve of
        their current positions, and state, contact list, contact history and persistent marks follow.  Mixing destroys the
        locality the initial numbering had, and the engine's gathers slow down with it (bench.py --order random: 28 %).
        Host-side like UpdateClumps (a few seconds per million clumps): meant to be called every few thousand steps.  Returns
        (params, scene, new_owner_of_old_owner); a user model's owner / geometry wildcard arrays are permuted along.  No reference
        equivalent (its owner ids never change)."""
        st = ctx.download_state()
        saved_wc = self._user_wildcards(ctx)
        cnt = ctx.contacts()
        nW = int(self.params.nContactWildcards)
        W = np.stack([ctx.wildcard(w) for w in range(nW)], 1) if nW else np.zeros((len(cnt[0]), 0), np.float32)
        pers = ctx.persistent_contacts() if ctx.num_persistent_contacts() else None
        p_old = self.params
        n_c, n_o = int(self.counts["nOwnerClumps"]), int(self.counts["nOwners"])
        old_owner_of_sphere = np.asarray(self.arrays["ownerClumpBody"], np.int64)
        n_sph_of = np.bincount(old_owner_of_sphere, minlength=n_c)[:n_c]
        first_old = np.r_[0, np.cumsum(n_sph_of)]
        X = decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p_old.nvXp2, p_old.nvYp2, p_old.voxelSize, p_old.l)[:n_c]
        rmax = max((float(t.radii.max()) + float(np.abs(t.relpos).max()) for t in self.templates), default=1.0)
        cell = float(cell) if cell else 4.0 * rmax
        old_of_new = np.zeros(n_c, np.int64)
        o0 = 0
        for bt in self.batches:  # a batch keeps its id range; its clumps are reordered inside it
            n = len(bt.xyz)
            order = np.argsort(morton_codes((X[o0:o0 + n] - X.min(0)).astype(np.float32), cell), kind="stable")
            old_of_new[o0:o0 + n] = o0 + order
            bt.xyz, bt.vel, bt.angvel, bt.oriq, bt.family = bt.xyz[order], bt.vel[order], bt.angvel[order], bt.oriq[order], bt.family[order]
            bt.templates = [bt.templates[i] for i in order]
            o0 += n
        new_of_old = np.arange(n_o, dtype=np.int64)
        new_of_old[old_of_new] = np.arange(n_c)
        p, sc = self.Initialize()
        src = np.r_[old_of_new, np.arange(n_c, n_o)]
        for k in ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY",
                  "omgBarZ", "familyID"):
            self.arrays[k][:] = st[k][src]
        sc = abi.make_scene_struct(self.arrays, self.counts)
        # sphere ids: spheres are clump-major, a clump's spheres keep their order
        first_new = np.r_[0, np.cumsum(n_sph_of[old_of_new])]
        new_sphere = np.zeros(len(old_owner_of_sphere), np.int64)
        k_in_clump = np.arange(len(old_owner_of_sphere)) - first_old[old_owner_of_sphere]
        new_sphere[:] = first_new[new_of_old[old_owner_of_sphere]] + k_in_clump

        def remap(a, b, t):
            a, b, t = np.asarray(a, np.int64), np.asarray(b, np.int64), np.asarray(t)
            ss = t == 1
            a2 = new_sphere[a]
            b2 = np.where(ss, new_sphere[np.where(ss, b, 0)], b)
            flip = ss & (a2 > b2)  # a sphere-sphere pair is stored smaller id first
            a3, b3 = np.where(flip, b2, a2), np.where(flip, a2, b2)
            return a3.astype(np.uint32), b3.astype(np.uint32), flip

        p.timeElapsed = float(time_elapsed)
        self.params = p
        ctx.set_params(p)
        ctx.upload_scene(sc)
        self.compile_into(ctx)
        self._restore_user_wildcards(ctx, saved_wc, new_of_old, new_sphere)
        if len(cnt[0]):
            a, b2, flip = remap(*cnt[:3])
            W = W.copy()
            if self.force_model == abi.FORCE_HERTZIAN:  # its delta_tan_x/y/z point from B to A (cf. decomp.redecompose); a user
                W[flip, :3] *= -1.0                      # model's directional history is the user's to re-orient
            ctx.seed_contacts(a, b2, cnt[2], W)
        if pers is not None:
            a, b2, _ = remap(*pers)
            ctx.set_persistent_contacts(a, b2, pers[2])
        return p, sc, new_of_old

    def ChangeFamilyWhen(self, id_from, id_to, condition):
        """condition: C++ statements that `return` a bool, over X, Y, Z, vX, vY, vZ, accX, accY, accZ, pos, vel, acc, mass, ts,
        time (API.h:1024; DEMModeratorKernels.cu:10-60); checked between force evaluation and integration of every step."""
        if not hasattr(self, "_family_rules"):
            self._family_rules = []
        self._family_rules.append((int(id_from), int(id_to), str(condition)))

    def family_change_rules(self):
        """_familyChangeRules_ as equipFamilyOnFlyChanges builds it (APIPrivate.cpp:1576-1598)."""
        out = " "
        for a, b_, cond in self._family_rules:
            out += f"if (family_code == {a}) {{ bool shouldMakeChange = false;"
            out += cond.replace("return", "shouldMakeChange = ")
            out += f"if (shouldMakeChange) {{granData->familyID[myOwner] = {b_};}}}}"
        return out

    # ---- family motion prescriptions (API.h:720-838; APIPublic.cpp:1013-1330) ---------------------------------
    _PRESC_FIELDS = ("linPosX", "linPosY", "linPosZ", "linVelX", "linVelY", "linVelZ", "oriQ", "rotVelX", "rotVelY", "rotVelZ",
                     "linPosPre", "linVelPre", "rotVelPre", "accX", "accY", "accZ", "angAccX", "angAccY", "angAccZ", "accPre",
                     "angAccPre")
    _PRESC_FLAGS = ("linVelXPrescribed", "linVelYPrescribed", "linVelZPrescribed", "rotVelXPrescribed", "rotVelYPrescribed",
                    "rotVelZPrescribed", "rotPosPrescribed", "linPosXPrescribed", "linPosYPrescribed", "linPosZPrescribed")

    def _presc(self, fam, **kw):
        if int(fam) > 255:
            raise ValueError(f"You applied prescribed motion to family {fam}, but family number should not be larger than 255.")
        if not hasattr(self, "_presc_inputs"):
            self._presc_inputs = []
        info = {k: "none" for k in self._PRESC_FIELDS}
        info.update({k: False for k in self._PRESC_FLAGS})
        info["family"] = int(fam)
        info.update(kw)
        self._presc_inputs.append(info)
        self.family_flags[int(fam)] |= abi.FAMILY_PRESCRIBED

    def SetFamilyPrescribedLinVel(self, fam, velX="none", velY="none", velZ="none", dictate=True, pre="none"):
        """Strings are C++ expressions of t (and X, Y, Z, vX ... as in the reference); "none": leave that component.  With
        dictate the family ignores contact forces for its translation AND rotation (APIPublic.cpp:1013-1054)."""
        f = {k: dictate for k in ("linVelXPrescribed", "linVelYPrescribed", "linVelZPrescribed", "rotVelXPrescribed",
                                  "rotVelYPrescribed", "rotVelZPrescribed")}
        for k, v in (("linVelXPrescribed", velX), ("linVelYPrescribed", velY), ("linVelZPrescribed", velZ)):
            if v != "none":
                f[k] = True
        self._presc(fam, linVelX=velX, linVelY=velY, linVelZ=velZ, linVelPre=pre, **f)

    def SetFamilyPrescribedAngVel(self, fam, velX="none", velY="none", velZ="none", dictate=True, pre="none"):
        f = {k: dictate for k in ("linVelXPrescribed", "linVelYPrescribed", "linVelZPrescribed", "rotVelXPrescribed",
                                  "rotVelYPrescribed", "rotVelZPrescribed")}
        for k, v in (("rotVelXPrescribed", velX), ("rotVelYPrescribed", velY), ("rotVelZPrescribed", velZ)):
            if v != "none":
                f[k] = True
        self._presc(fam, rotVelX=velX, rotVelY=velY, rotVelZ=velZ, rotVelPre=pre, **f)

    def SetFamilyPrescribedPosition(self, fam, X="none", Y="none", Z="none", dictate=True, pre="none"):
        f = {k: dictate for k in ("linPosXPrescribed", "linPosYPrescribed", "linPosZPrescribed", "rotPosPrescribed")}
        for k, v in (("linPosXPrescribed", X), ("linPosYPrescribed", Y), ("linPosZPrescribed", Z)):
            if v != "none":
                f[k] = True
        self._presc(fam, linPosX=X, linPosY=Y, linPosZ=Z, linPosPre=pre, **f)

    def SetFamilyPrescribedQuaternion(self, fam, q_formula="none", dictate=True):
        """q_formula: statements that `return` a float4 (x, y, z, w), e.g. "return make_float4(0, 0, sinf(t), cosf(t));"."""
        f = {k: dictate for k in ("linPosXPrescribed", "linPosYPrescribed", "linPosZPrescribed", "rotPosPrescribed")}
        if q_formula != "none":
            f["rotPosPrescribed"] = True
        self._presc(fam, oriQ=q_formula, **f)

    def AddFamilyPrescribedAcc(self, fam, X="none", Y="none", Z="none", pre="none"):
        self._presc(fam, accX=X, accY=Y, accZ=Z, accPre=pre)

    def AddFamilyPrescribedAngAcc(self, fam, X="none", Y="none", Z="none", pre="none"):
        self._presc(fam, angAccX=X, angAccY=Y, angAccZ=Z, angAccPre=pre)

    def prescription_cases(self):
        """The three switch bodies of equipFamilyPrescribedMotions (APIPrivate.cpp:1600-1708), after the per-family merge
        of APIPrivate.cpp:843-937 (later strings replace earlier ones, flags are OR-ed)."""
        merged = {}
        for inp in self._presc_inputs:
            m = merged.setdefault(inp["family"], dict({k: "none" for k in self._PRESC_FIELDS}, **{k: False for k in self._PRESC_FLAGS}))
            for k in self._PRESC_FIELDS:
                if inp[k] != "none":
                    m[k] = inp[k]
            for k in self._PRESC_FLAGS:
                m[k] = m[k] or inp[k]
        vel = pos = acc = " "
        b = lambda v: "1" if v else "0"  # std::to_string(bool)
        for fam in sorted(merged):
            m = merged[fam]
            head = f"case {fam}: {{"
            v = head + "{"
            if m["linVelPre"] != "none":
                v += m["linVelPre"] + ";"
            for name, key in (("vX", "linVelX"), ("vY", "linVelY"), ("vZ", "linVelZ")):
                if m[key] != "none":
                    v += f"{name} = {m[key]};"
            v += "}{"
            if m["rotVelPre"] != "none":
                v += m["rotVelPre"] + ";"
            for name, key in (("omgBarX", "rotVelX"), ("omgBarY", "rotVelY"), ("omgBarZ", "rotVelZ")):
                if m[key] != "none":
                    v += f"{name} = {m[key]};"
            v += "}"
            for name, key in (("LinVelXPrescribed", "linVelXPrescribed"), ("LinVelYPrescribed", "linVelYPrescribed"),
                              ("LinVelZPrescribed", "linVelZPrescribed"), ("RotVelXPrescribed", "rotVelXPrescribed"),
                              ("RotVelYPrescribed", "rotVelYPrescribed"), ("RotVelZPrescribed", "rotVelZPrescribed")):
                v += f"{name} = {b(m[key])};"
            vel += v + "break; }"
            q = head + "{"
            if m["linPosPre"] != "none":
                q += m["linPosPre"] + ";"
            for name, key in (("X", "linPosX"), ("Y", "linPosY"), ("Z", "linPosZ")):
                if m[key] != "none":
                    q += f"{name} = {m[key]};"
            q += "}"
            if m["oriQ"] != "none":
                q += "{" + m["oriQ"].replace("return", "float4 DEME_Presc_OriQ = ") + ";"
                q += "oriQw = DEME_Presc_OriQ.w; oriQx = DEME_Presc_OriQ.x; oriQy = DEME_Presc_OriQ.y; oriQz = DEME_Presc_OriQ.z;}"
            for name, key in (("LinXPrescribed", "linPosXPrescribed"), ("LinYPrescribed", "linPosYPrescribed"),
                              ("LinZPrescribed", "linPosZPrescribed"), ("RotPrescribed", "rotPosPrescribed")):
                q += f"{name} = {b(m[key])};"
            pos += q + "break; }"
            a = head + "{"
            if m["accPre"] != "none":
                a += m["accPre"] + ";"
            for name, key in (("accX", "accX"), ("accY", "accY"), ("accZ", "accZ")):
                if m[key] != "none":
                    a += f"{name} = {m[key]};"
            a += "}{"
            if m["angAccPre"] != "none":
                a += m["angAccPre"] + ";"
            for name, key in (("angAccX", "angAccX"), ("angAccY", "angAccY"), ("angAccZ", "angAccZ")):
                if m[key] != "none":
                    a += f"{name} = {m[key]};"
            acc += a + "}break; }"
        return vel, pos, acc

    def SetFamilyFixed(self, fam):
        self.family_flags[int(fam)] |= abi.FAMILY_FIXED

    def DisableContactBetweenFamilies(self, a, b):
        self.family_masks[mask_pair(int(a), int(b))] = 1

    def SetFamilyExtraMargin(self, fam, m):
        self.family_extra[int(fam)] = np.float32(m)

    # ---- sizing ---------------------------------------------------------------
    def _figure_out_nv(self):
        size = (self.target_box_max - self.target_box_min).astype(np.float32)
        xyz = [float(size[0]), float(size[1]), float(size[2])]
        rank = [0, 1, 2]
        for i in range(2):
            for j in range(i + 1, 3):
                if xyz[i] > xyz[j]:
                    xyz[i], xyz[j] = xyz[j], xyz[i]
                    rank[i], rank[j] = rank[j], rank[i]
        user321 = list(xyz)
        more = [0, 0]
        # float arithmetic in the reference (float XYZ[3] *= 2.)
        a0, a1, a2 = np.float32(xyz[0]), np.float32(xyz[1]), np.float32(xyz[2])
        while a0 < a1:
            if math.sqrt(2.0) * float(a0) > float(a1):
                break
            more[0] += 1
            a0 = np.float32(float(a0) * 2.0)
        while a1 < a2:
            if math.sqrt(2.0) * float(a1) > float(a2):
                break
            more[1] += 1
            a1 = np.float32(float(a1) * 2.0)
        budget = VOXEL_COUNT_POWER2 - 2 * more[0] - more[1]
        base, left = budget // 3, budget % 3
        b3 = base
        b2 = b3 + more[0]
        b1 = b2 + more[1]
        while left > 0:
            if b3 < b2:
                b3 += 1
            elif b2 < b1:
                b2 += 1
            else:
                b1 += 1
            left -= 1
        bits = [b3, b2, b1]
        ls = [user321[k] / 2.0 ** VOXEL_RES_POWER2 / 2.0 ** bits[k] for k in range(3)]
        l = max(ls)
        nv = [0, 0, 0]
        for k in range(3):
            nv[rank[k]] = bits[k]
        voxel = float(1 << VOXEL_RES_POWER2) * l
        return nv, l, voxel

    @staticmethod
    def _calc_bin_num(voxel, bin_size, nv):
        nb = [int(voxel * float(1 << nv[k]) / bin_size) + 1 for k in range(3)]
        return nb, nb[0] * nb[1] * nb[2]

    # ---- flatten ----------------------------------------------------------------
    def Initialize(self):
        nv, l, voxel = self._figure_out_nv()
        lbf = self.target_box_min.astype(np.float32)
        for bt in self.batches:  # the reference's courtesy check (dT.cpp:739-744, 887-893): a warning, not an error
            out = ((bt.xyz < self.user_box_min) | (bt.xyz > self.user_box_max)).any(axis=1)
            if out.any():
                import warnings
                sx = bt.xyz[np.argmax(out)]
                warnings.warn("At least one clump is initialized with a position out of the box domain you specified. It is found at "
                              f"{sx[0]:.5g}, {sx[1]:.5g}, {sx[2]:.5g} (this message only shows one such example). This simulation is "
                              "unlikely to go as planned.", stacklevel=2)
                break

        # templates sorted by component count (APIPrivate.cpp:696-742); stable here
        order = sorted(range(len(self.templates)), key=lambda i: len(self.templates[i].radii))
        comp_prefix = {}
        radii, rel = [], []
        for mark, ti in enumerate(order):
            t = self.templates[ti]
            t.mark = mark
            comp_prefix[mark] = len(radii)
            radii.extend(t.radii.tolist())
            rel.extend(t.relpos.tolist())
        radii = np.asarray(radii, np.float32)
        rel = np.asarray(rel, np.float32).reshape(-1, 3)
        tmpl_sorted = [self.templates[i] for i in order]

        smallest = float(radii.min()) if len(radii) else 1.0
        # the reference multiplies two floats (m_binSize_as_multiple * m_smallest_radius) and stores the product in a double
        bin_size = self.bin_size if self.bin_size is not None else float(np.float32(self.bin_multiple) * np.float32(smallest))
        nb, nbins = self._calc_bin_num(voxel, bin_size, nv)
        if self.target_bin_num is not None and self.bin_size is None:
            prev = nbins
            tgt = self.target_bin_num
            while nbins < 0.67 * tgt or nbins > 1.5 * tgt:
                bin_size *= 0.8 if nbins < tgt else 1.2
                nb, nbins = self._calc_bin_num(voxel, bin_size, nv)
                if (prev < tgt <= nbins) or (prev >= tgt > nbins):
                    break
                prev = nbins
        while nbins > 0xFFFFFFFE:
            bin_size *= 1.5
            nb, nbins = self._calc_bin_num(voxel, bin_size, nv)

        # world bounding box as one analytical owner (APIPrivate.cpp:955-1014)
        ext = list(self.ext_objs)
        if self.bounding_bc != "none":
            bottom = self.bounding_bc in ("only_bottom", "top_open", "all")
            sides = self.bounding_bc in ("only_sides", "top_open", "all")
            top = self.bounding_bc == "all"
            box = ExternObj()
            c = ((self.user_box_min + self.user_box_max) / np.float32(2.0)).astype(np.float32)
            mat = self.bounding_mat
            if bottom:
                box.AddPlane((c[0], c[1], self.user_box_min[2]), (0, 0, 1), mat)
            if sides:
                box.AddPlane((self.user_box_min[0], c[1], c[2]), (1, 0, 0), mat)
                box.AddPlane((self.user_box_max[0], c[1], c[2]), (-1, 0, 0), mat)
                box.AddPlane((c[0], self.user_box_min[1], c[2]), (0, 1, 0), mat)
                box.AddPlane((c[0], self.user_box_max[1], c[2]), (0, -1, 0), mat)
            if top:
                box.AddPlane((c[0], c[1], self.user_box_max[2]), (0, 0, -1), mat)
            ext.append(box)

        # owners: clumps (batch load order) then analytical objects
        n_clumps = sum(len(b.xyz) for b in self.batches)
        n_owners = n_clumps + len(ext) + len(self.meshes)
        xyz = np.zeros((n_owners, 3), np.float32)
        oriq = np.tile(np.array([0, 0, 0, 1], np.float32), (n_owners, 1))
        vel = np.zeros((n_owners, 3), np.float32)
        ang = np.zeros((n_owners, 3), np.float32)
        fam = np.zeros(n_owners, np.uint8)
        inert = np.zeros(n_owners, np.uint16)
        sph_owner, sph_comp, sph_mat = [], [], []
        o = 0
        for b in self.batches:
            n = len(b.xyz)
            xyz[o:o + n] = b.xyz
            oriq[o:o + n] = b.oriq
            vel[o:o + n] = b.vel
            ang[o:o + n] = b.angvel
            fam[o:o + n] = b.family
            marks = np.array([t.mark for t in b.templates], np.int64)
            inert[o:o + n] = marks
            uniq = np.unique(marks)
            if len(uniq) == 1:
                t = tmpl_sorted[int(uniq[0])]
                k = len(t.radii)
                sph_owner.append(np.repeat(np.arange(o, o + n, dtype=np.uint32), k))
                sph_comp.append(np.tile(np.arange(k, dtype=np.uint16) + np.uint16(comp_prefix[int(uniq[0])]), n))
                sph_mat.append(np.tile(np.asarray(t.materials, np.uint16), n))
            else:
                for i in range(n):
                    t = tmpl_sorted[int(marks[i])]
                    k = len(t.radii)
                    sph_owner.append(np.full(k, o + i, np.uint32))
                    sph_comp.append(np.arange(k, dtype=np.uint16) + np.uint16(comp_prefix[int(marks[i])]))
                    sph_mat.append(np.asarray(t.materials, np.uint16))
            o += n
        n_tmpl = len(tmpl_sorted)
        mass = [t.mass for t in tmpl_sorted]
        self.template_volumes = [t.volume for t in tmpl_sorted]  # analytical / mesh owners: 0, like the reference (dT.cpp:607-618)
        moi = [t.moi for t in tmpl_sorted]
        obj = {k: [] for k in ("type", "owner", "normal", "mat", "px", "py", "pz", "rx", "ry", "rz", "s1", "s2", "s3",
                               "mass")}
        for ei, e in enumerate(ext):
            owner = n_clumps + ei
            xyz[owner] = np.asarray(e.init_pos, np.float32)
            oriq[owner] = np.asarray(e.init_oriq, np.float32)
            fam[owner] = e.family
            inert[owner] = n_tmpl + ei
            mass.append(e.mass)
            moi.append(e.moi)
            for (ty, pos, rot, s1, s2, s3, nsign, mat) in e.comps:
                obj["type"].append(ty), obj["owner"].append(owner), obj["normal"].append(nsign), obj["mat"].append(mat)
                obj["px"].append(pos[0]), obj["py"].append(pos[1]), obj["pz"].append(pos[2])
                obj["rx"].append(rot[0]), obj["ry"].append(rot[1]), obj["rz"].append(rot[2])
                obj["s1"].append(s1), obj["s2"].append(s2), obj["s3"].append(s3), obj["mass"].append(e.mass)

        # meshes: one owner each after the analytical owners; triangles mesh-major (APIPrivate.cpp:756-810)
        tri_owner, tri_n1, tri_n2, tri_n3, tri_mat = [], [], [], [], []
        for mi, me in enumerate(self.meshes):
            owner = n_clumps + len(ext) + mi
            xyz[owner] = np.asarray(me.init_pos, np.float32)
            oriq[owner] = np.asarray(me.init_oriq, np.float32)
            vel[owner] = np.asarray(me.vel, np.float32)
            fam[owner] = me.family
            inert[owner] = n_tmpl + len(ext) + mi
            mass.append(me.mass)
            moi.append(me.moi)
            a, b_, c_ = me.nodes()
            tri_owner.append(np.full(len(a), owner, np.uint32))
            tri_n1.append(a), tri_n2.append(b_), tri_n3.append(c_)
            tri_mat.append(np.full(len(a), me.material, np.uint16))
        shifted = (xyz - lbf[None, :]).astype(np.float32).astype(np.float64)  # float3 subtraction, dT.cpp:745
        vid, lx, ly, lz = encode_positions(shifted, nv[0], nv[1], voxel, l)

        # materials (APIPrivate.cpp:1877-2026)
        nm = max(1, len(self.materials))
        def prop(name, default=0.0):
            return np.array([m.get(name, default) for m in self.materials] or [default], np.float32)
        E, nu = prop("E"), prop("nu")
        def pair(name):
            v = prop(name)
            M = ((v[:, None] + v[None, :]) / np.float32(2.0)).astype(np.float32)
            for i in range(nm):
                M[i, i] = v[i]
            for (n_, a, b), val in self.pair_overrides.items():
                if n_ == name:
                    M[a, b] = np.float32(val)
            return M.reshape(-1)
        CoR, mu, Crr = pair("CoR"), pair("mu"), pair("Crr")

        nW = {abi.FORCE_HERTZIAN: 4, abi.FORCE_HERTZIAN_FRICTIONLESS: 0}.get(self.force_model, self.n_custom_wildcards)
        p = abi.DemeParams()
        p.nvXp2, p.nvYp2, p.nvZp2 = nv
        p.nbX, p.nbY, p.nbZ = nb
        p.l, p.voxelSize, p.binSize = l, voxel, bin_size
        p.LBFX, p.LBFY, p.LBFZ = [float(x) for x in lbf]
        p.Gx, p.Gy, p.Gz = self.G
        p.h = self.h
        p.beta = self.expand_factor
        p.approxMaxVel = self.approx_max_vel
        p.expSafetyMulti, p.expSafetyAdder = self.safety_multi, self.safety_adder
        p.integrator, p.forceModel, p.nContactWildcards = self.integrator, self.force_model, nW
        p.cdUpdateFreq = self.cd_update_freq
        p.errOutBinSphNum = self.err_out_bin_sph
        p.errOutVel = self.err_out_vel
        p.timeElapsed = 0.0

        cat = lambda lst, dt: (np.concatenate(lst).astype(dt) if lst else np.zeros(0, dt))
        arrays = {
            "voxelID": vid, "locX": lx, "locY": ly, "locZ": lz,
            "oriQw": oriq[:, 3].copy(), "oriQx": oriq[:, 0].copy(), "oriQy": oriq[:, 1].copy(), "oriQz": oriq[:, 2].copy(),
            "vX": vel[:, 0].copy(), "vY": vel[:, 1].copy(), "vZ": vel[:, 2].copy(),
            "omgBarX": ang[:, 0].copy(), "omgBarY": ang[:, 1].copy(), "omgBarZ": ang[:, 2].copy(),
            "familyID": fam, "inertiaPropOffsets": inert,
            "ownerClumpBody": cat(sph_owner, np.uint32), "clumpComponentOffset": cat(sph_comp, np.uint16),
            "sphereMaterialOffset": cat(sph_mat, np.uint16),
            "Radii": radii, "CDRelPosX": rel[:, 0].copy() if len(rel) else np.zeros(0, np.float32),
            "CDRelPosY": rel[:, 1].copy() if len(rel) else np.zeros(0, np.float32),
            "CDRelPosZ": rel[:, 2].copy() if len(rel) else np.zeros(0, np.float32),
            "MassProperties": np.asarray(mass, np.float32),
            "moiX": np.asarray([m[0] for m in moi], np.float32), "moiY": np.asarray([m[1] for m in moi], np.float32),
            "moiZ": np.asarray([m[2] for m in moi], np.float32),
            "objType": np.asarray(obj["type"], np.uint8), "objOwner": np.asarray(obj["owner"], np.uint32),
            "objNormal": np.asarray(obj["normal"], np.float32), "objMaterial": np.asarray(obj["mat"], np.uint16),
            "objRelPosX": np.asarray(obj["px"], np.float32), "objRelPosY": np.asarray(obj["py"], np.float32),
            "objRelPosZ": np.asarray(obj["pz"], np.float32),
            "objRotX": np.asarray(obj["rx"], np.float32), "objRotY": np.asarray(obj["ry"], np.float32),
            "objRotZ": np.asarray(obj["rz"], np.float32),
            "objSize1": np.asarray(obj["s1"], np.float32), "objSize2": np.asarray(obj["s2"], np.float32),
            "objSize3": np.asarray(obj["s3"], np.float32), "objMass": np.asarray(obj["mass"], np.float32),
            "E": E, "nu": nu, "CoR": CoR, "mu": mu, "Crr": Crr,
            "familyMasks": self.family_masks, "familyExtraMarginSize": self.family_extra,
            "familyFlags": self.family_flags,
        }
        n_tri = int(sum(len(x) for x in tri_owner))
        if n_tri:
            arrays.update({"ownerMesh": np.concatenate(tri_owner), "triNode1": np.concatenate(tri_n1).reshape(-1),
                           "triNode2": np.concatenate(tri_n2).reshape(-1), "triNode3": np.concatenate(tri_n3).reshape(-1),
                           "triMaterialOffset": np.concatenate(tri_mat)})
        counts = {"nOwners": n_owners, "nOwnerClumps": n_clumps, "nSpheres": len(arrays["ownerClumpBody"]),
                  "nAnal": len(obj["type"]), "nTri": n_tri, "nMat": nm, "nComp": len(radii), "nMassProps": len(mass)}
        self.params, self.arrays, self.counts = p, arrays, counts
        # clump mark -> type name for the clump output file (m_template_number_name_map, APIPrivate.cpp:721)
        self.template_names = {t.mark: (t.name if t.name is not None else "%04d" % self.templates.index(t)) for t in self.templates}
        self.scene = abi.make_scene_struct(arrays, counts)
        return p, self.scene


def icosphere(radius, subdiv=2):
    """vertices (float32) and facets of an icosahedron subdivided `subdiv` times, on the sphere of the given radius"""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1),
         (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(x, float) / np.linalg.norm(x) for x in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return (np.array(v) * radius).astype(np.float32), np.array(f, np.int64)



def balldrop_like(n_target=10000, seed=12345, projectile=None):
    """BASELINE configs[0]: a BallDrop-like scene -- ~1e4 single-sphere clumps from 11 templates
    (r = 1.25 ... 1.75 mm, rho = 2500), E 7e7, nu 0.24, CoR 0.9, mu 0.3, Crr 0, box 0.2 x 0.2 x 2 "top_open", h 2e-6, plus a
    spherical projectile mesh dropped into the bed (DEMdemo_BallDrop.cpp:53-150; seeded HCP + jitter instead of
    std::random_device / PD sampling).  projectile = (vertices, faces) of a unit sphere; default: a twice-subdivided
    icosahedron (162 vertices / 320 facets, the shape of the reference's data/mesh/sphere.obj)."""
    import math
    b = SceneBuilder()
    mat = b.LoadMaterial({"E": 7e7, "nu": 0.24, "CoR": 0.9, "mu": 0.3, "Crr": 0.0})
    b.InstructBoxDomainDimension((-0.1, 0.1), (-0.1, 0.1), (0.0, 2.0))
    b.InstructBoxDomainBoundingBC("top_open", mat)
    radii = [0.00125 + 0.00005 * i for i in range(11)]
    tmpls = [b.LoadSphereType(2500.0 * 4.0 / 3.0 * math.pi * r ** 3, r, mat) for r in radii]
    sep = 0.0033  # denser than the largest diameter (3.5 mm): the bed starts with contacts, like a poured bed
    side = int(round((n_target / 6) ** 0.5))
    pts = hcp_points([-side * sep / 2, -side * sep / 2, 0.0017], [side * sep / 2, side * sep / 2, 0.0017 + 7 * sep], sep)
    pts = pts[:n_target]  # z-major ordering: the lowest layers
    rng = np.random.default_rng(seed)
    pts = pts + ((rng.random(pts.shape) * 2 - 1) * 0.02 * sep).astype(np.float32)
    pick = rng.integers(0, len(tmpls), len(pts))
    batch = b.AddClumps([tmpls[i] for i in pick], pts)
    batch.SetVel(np.tile(np.array([0, 0, -0.2], np.float32), (len(pts), 1)))
    v, f = projectile if projectile is not None else icosphere(1.0, 2)
    proj = b.AddMeshObject(np.asarray(v, np.float32) * np.float32(0.012), f, mat)
    under = (np.abs(pts[:, 0]) < 0.008) & (np.abs(pts[:, 1]) < 0.008)  # the top layer may be partial: look under the projectile
    proj.SetInitPos((0.0, 0.0, float(pts[under, 2].max()) + 0.0015 + 0.012 + 0.0001))  # ~0.1 mm above the spheres below it
    proj.SetMass(2.6e3 * 4.0 / 3.0 * math.pi * 0.012 ** 3)
    proj.SetMOI((2.7e-7, 2.7e-7, 2.7e-7))
    proj.SetFamily(2)
    proj.vel = (0.0, 0.0, -2.0)
    b.SetInitTimeStep(2e-6)
    b.SetGravitationalAcceleration((0, 0, -9.81))
    b.SetCDUpdateFreq(0)
    b.SetInitBinSizeAsMultipleOfSmallestSphere(4.0)
    b.SetMaxVelocity(15.0)
    b.SetErrorOutVelocity(1e3)
    return b


def plate_mesh(nx, ny, size_x, size_y, z=0.0, wavy=0.0):
    """A tessellated rectangular plate (2*nx*ny triangles, normals +z), optionally with a sinusoidal relief."""
    xs = np.linspace(-size_x / 2, size_x / 2, nx + 1, dtype=np.float32)
    ys = np.linspace(-size_y / 2, size_y / 2, ny + 1, dtype=np.float32)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    Z = z + wavy * np.sin(6.0 * X / max(size_x, 1e-9)) * np.cos(5.0 * Y / max(size_y, 1e-9))
    v = np.stack([X.ravel(), Y.ravel(), Z.ravel().astype(np.float32)], 1).astype(np.float32)
    idx = lambda i, j: i * (ny + 1) + j
    f = []
    for i in range(nx):
        for j in range(ny):
            f.append((idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)))
            f.append((idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)))
    return v, np.asarray(f, np.int64)


def hcp_points(lo, hi, sep):
    """HCPSampler box sampling (DEM/utils/Samplers.hpp:498-533), float arithmetic."""
    lo = np.asarray(lo, np.float32)
    hi = np.asarray(hi, np.float32)
    dx = np.float32(sep)
    dy = np.float32(sep) * np.float32(math.sqrt(3.0) / 2)
    dz = np.float32(sep) * np.float32(math.sqrt(2.0 / 3.0))
    size = hi - lo
    nx, ny, nz = int(size[0] / dx) + 1, int(size[1] / dy) + 1, int(size[2] / dz) + 1
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    offy = np.where(k % 2 == 0, np.float32(0), dy / np.float32(3)).astype(np.float32)
    offx = np.where((j + k) % 2 == 0, np.float32(0), dx / np.float32(2)).astype(np.float32)
    x = lo[0] + (offx + i.astype(np.float32) * dx)
    y = lo[1] + (offy + j.astype(np.float32) * dy)
    z = lo[2] + (k.astype(np.float32) * dz)
    pts = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1).astype(np.float32)
    ok = np.all((pts >= lo) & (pts <= hi), axis=1)
    return pts[ok]


def random_unit_quaternions(n, rng):
    q = rng.standard_normal((n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    return q.astype(np.float32)  # x y z w


def morton_codes(pts, cell):
    """Z-order code of the cell (edge `cell`) each point falls in; 21 bits per axis."""
    q = np.floor(np.asarray(pts, np.float64) / cell).astype(np.int64)
    q -= q.min(axis=0)
    code = np.zeros(len(q), np.uint64)
    for bit in range(21):
        for ax in range(3):
            code |= ((q[:, ax].astype(np.uint64) >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + ax)
    return code


def packed_bed(n_target, seed=2024, scale=0.005, spacing_mult=3.0, jitter=0.05, aspect=(1.0, 1.0, 0.45),
               three_sphere=True, cd_freq=0, E=1e8, nu=0.3, CoR=0.6, mu=0.2, Crr=0.0, h=5e-6, bin_multiple=4.0,
               radii_poly=None, force_model=abi.FORCE_HERTZIAN, init_vz=0.0, order="lattice", slab=None):
    """BASELINE.md config-2 recipe: three-sphere clumps (3_clump.csv * scale) on an HCP lattice of
    spacing 3*scale with seeded jitter and random orientations inside a box with 5 wall planes.
    slab = (rank, n_ranks, halo): the scene of ONE x-slab of that bed -- the clumps rank owns plus those within `halo` of its faces,
    with the positions, orientations and kinds the whole bed would give them (the cheap per-clump random draws are made for the
    whole bed, the expensive flattening only for the slab).  The builder then carries slab_edges (equal-count boundaries of the whole
    bed), slab_global_ids (global clump index of every local clump) and slab_x (their x); decomp.decompose(..., edges=, only_rank=)
    turns it into rank's part without any rank ever holding the whole bed's scene."""
    rng = np.random.default_rng(seed)
    sep = spacing_mult * scale
    vol_per = sep ** 3 / math.sqrt(2.0)
    vol = n_target * vol_per
    a = np.asarray(aspect, float)
    s = (vol / float(np.prod(a))) ** (1.0 / 3.0)
    dims = a * s
    b = SceneBuilder()
    mat = b.LoadMaterial({"E": E, "nu": nu, "CoR": CoR, "mu": mu, "Crr": Crr})
    pad = 2.0 * sep
    box = dims + 2 * pad
    b.InstructBoxDomainDimension((0.0, float(box[0])), (0.0, float(box[1])), (0.0, float(box[2] * 1.3)))
    b.InstructBoxDomainBoundingBC("top_open", mat)
    pts = hcp_points([pad, pad, pad], [pad + dims[0], pad + dims[1], pad + dims[2]], sep)
    if len(pts) > n_target:
        pts = pts[:n_target]  # keep the lowest layers (z-major ordering)
    if order == "random":  # input order of the clumps: the engine keeps the caller's numbering
        pts = pts[rng.permutation(len(pts))]
    elif order == "morton":
        pts = pts[np.argsort(morton_codes(pts, 2.0 * sep), kind="stable")]
    pts = pts + ((rng.random(pts.shape) * 2 - 1) * (jitter * sep)).astype(np.float32)
    keep = None
    if slab is not None:
        from .decomp import slab_edges
        rank, n_ranks, halo = slab
        x = pts[:, 0].astype(np.float64)
        edges = slab_edges(x, n_ranks)
        keep = np.nonzero((x >= edges[rank] - halo) & (x < edges[rank + 1] + halo))[0]
        b.slab_edges, b.slab_global_ids, b.slab_x, b.slab_total = edges, keep, pts[keep, 0].copy(), len(pts)
    sub = (lambda v: v) if keep is None else (lambda v: v[keep])
    if three_sphere:
        tmpl = b.LoadThreeSphereClump(scale, 2.6e3, mat)
        batch = b.AddClumps(tmpl, sub(pts))
    else:
        radii_poly = radii_poly or [scale]
        tmpls = [b.LoadSphereType(2.6e3 * 4.0 / 3.0 * math.pi * r ** 3, r, mat) for r in radii_poly]
        pick = rng.integers(0, len(tmpls), len(pts))
        batch = b.AddClumps([tmpls[i] for i in sub(pick)], sub(pts))
    batch.SetOriQ(sub(random_unit_quaternions(len(pts), rng)))
    n_local = len(pts) if keep is None else len(keep)
    if init_vz:
        batch.SetVel(np.tile(np.array([0, 0, init_vz], np.float32), (n_local, 1)))
    b.SetInitTimeStep(h)
    b.SetGravitationalAcceleration((0, 0, -9.81))
    b.SetCDUpdateFreq(cd_freq)
    b.SetInitBinSizeAsMultipleOfSmallestSphere(bin_multiple)
    b.SetExpandSafetyAdder(0.0)  # margins from the owners' own speeds alone (the reference's default adds 3 m/s: far too wide for a bed)
    b.force_model = force_model
    b.SetMaxVelocity(5.0)
    b.SetErrorOutVelocity(1e3)
    return b
