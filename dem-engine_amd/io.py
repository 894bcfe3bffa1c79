"""Output writers and CSV readers of the reference's file formats.

Formats, column names and number formatting follow the reference writers line by line
(DEM/dT.cpp:1254-1405 writeSpheresAsCsv, :1491-1618 writeClumpsAsCsv, :1620-1848 generateContactInfo +
writeContactsAsCsv; column names DEM/Structs.h:41-97; content flags DEM/Defines.h:152-183) and the static
readers of DEMSolver (DEM/API.h:1153-1250).  Numbers are printed the way a C++ ostream prints a float:
"%.<precision>g" of the fp32 value (precision 6; the clump file takes `accuracy`, default 10).

Everything here is host-side post-processing of arrays downloaded through the C-ABI; positions are decoded in
fp32 exactly as the writers do (voxelIDToPosition<float, ...>, kernel/DEMHelperKernels.cuh:116-135).
"""
import csv

import numpy as np


class OUTPUT_CONTENT:  # DEM/Defines.h:152-170
    XYZ, QUAT, ABSV, VEL, ANG_VEL, ABS_ACC, ACC, ANG_ACC, FAMILY, MAT, OWNER_WILDCARD, GEO_WILDCARD = \
        0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024


class CNT_OUTPUT_CONTENT:  # DEM/Defines.h:172-183
    CNT_TYPE, FORCE, CNT_POINT, COMPONENT, NORMAL, TORQUE, CNT_WILDCARD, OWNER, GEO_ID, NICKNAME = \
        0, 1, 2, 4, 8, 16, 32, 64, 128, 256


DEFAULT_OUTPUT_CONTENT = OUTPUT_CONTENT.QUAT | OUTPUT_CONTENT.ABSV  # API.h:1418
DEFAULT_CNT_OUTPUT_CONTENT = (CNT_OUTPUT_CONTENT.OWNER | CNT_OUTPUT_CONTENT.GEO_ID | CNT_OUTPUT_CONTENT.FORCE |
                              CNT_OUTPUT_CONTENT.CNT_POINT | CNT_OUTPUT_CONTENT.CNT_WILDCARD)  # API.h:1422-1424

# DEM/Structs.h:60-90
CNT_FILE_KNOWN_COL_NAMES = {"A", "B", "compA", "compB", "geoA", "geoB", "nameA", "nameB", "contact_type", "f_x", "f_y", "f_z",
                            "torque_x", "torque_y", "torque_z", "n_x", "n_y", "n_z", "SS", "SA", "SM"}
CONTACT_TYPE_NAME = {0: "fake", 1: "SS", 2: "SM", 11: "SA", 12: "SA", 13: "SA", 14: "SA"}  # Structs.h:90-97


def _g(x, prec=6):
    return "%.*g" % (prec, float(x))


def decode_positions_f32(state, p):
    """voxelIDToPosition<float,...> + LBF, all fp32 (dT.cpp:1312-1320)."""
    vid = state["voxelID"].astype(np.uint64)
    vx = vid & np.uint64((1 << p.nvXp2) - 1)
    vy = (vid >> np.uint64(p.nvXp2)) & np.uint64((1 << p.nvYp2) - 1)
    vz = vid >> np.uint64(p.nvXp2 + p.nvYp2)
    vs, l = np.float32(p.voxelSize), np.float32(p.l)
    out = np.empty((len(vid), 3), np.float32)
    for k, (v, s, lbf) in enumerate(((vx, state["locX"], p.LBFX), (vy, state["locY"], p.LBFY), (vz, state["locZ"], p.LBFZ))):
        out[:, k] = (v.astype(np.float32) * vs + s.astype(np.float32) * l) + np.float32(lbf)
    return out


def rotate_f32(q_wxyz, v):
    """applyOriQToVector3<float,float> (DEMHelperKernels.cuh:162-173) on arrays [n,3]."""
    w, x, y, z = (q_wxyz[:, i].astype(np.float32) for i in range(4))
    two = np.float32(2.0)
    one = np.float32(1.0)
    vx, vy, vz = (v[:, i].astype(np.float32) for i in range(3))
    ox = (two * (w * w + x * x) - one) * vx + (two * (x * y - w * z)) * vy + (two * (x * z + w * y)) * vz
    oy = (two * (x * y + w * z)) * vx + (two * (w * w + y * y) - one) * vy + (two * (y * z - w * x)) * vz
    oz = (two * (x * z - w * y)) * vx + (two * (y * z + w * x)) * vy + (two * (w * w + z * z) - one) * vz
    return np.stack([ox, oy, oz], 1).astype(np.float32)


def _quat(state):
    return np.stack([state["oriQw"], state["oriQx"], state["oriQy"], state["oriQz"]], 1).astype(np.float32)


def _len3(a):
    return np.sqrt((a.astype(np.float32) ** 2).sum(1, dtype=np.float32)).astype(np.float32)


def _owner_columns(flags, state, owners, prec):
    """the optional per-owner columns shared by the sphere and the clump file, in the writers' order"""
    names, cols = [], []
    v = np.stack([state["vX"], state["vY"], state["vZ"]], 1)[owners]
    a = np.stack([state["aX"], state["aY"], state["aZ"]], 1)[owners]
    if flags & OUTPUT_CONTENT.ABSV:
        names += ["absv"]
        cols += [_len3(v)]
    if flags & OUTPUT_CONTENT.VEL:
        names += ["v_x", "v_y", "v_z"]
        cols += [v[:, 0], v[:, 1], v[:, 2]]
    if flags & OUTPUT_CONTENT.ANG_VEL:
        names += ["w_x", "w_y", "w_z"]
        cols += [state["omgBarX"][owners], state["omgBarY"][owners], state["omgBarZ"][owners]]
    if flags & OUTPUT_CONTENT.ABS_ACC:
        names += ["abs_acc"]
        cols += [_len3(a)]
    if flags & OUTPUT_CONTENT.ACC:
        names += ["a_x", "a_y", "a_z"]
        cols += [a[:, 0], a[:, 1], a[:, 2]]
    if flags & OUTPUT_CONTENT.ANG_ACC:
        names += ["alpha_x", "alpha_y", "alpha_z"]
        cols += [state["alphaX"][owners], state["alphaY"][owners], state["alphaZ"][owners]]
    if flags & OUTPUT_CONTENT.FAMILY:
        names += ["family"]
        cols += [state["familyID"][owners].astype(np.int64)]
    return names, cols


def _write_table(path, header, cols, prec, text_cols=()):
    n = len(cols[0]) if cols else 0
    with open(path, "w") as f:
        f.write(",".join(header) + "\n")
        for i in range(n):
            row = []
            for k, c in enumerate(cols):
                row.append(str(c[i]) if (k in text_cols or np.issubdtype(np.asarray(c).dtype, np.integer)) else _g(c[i], prec))
            f.write(",".join(row) + "\n")


def write_sphere_file(path, p, arrays, counts, state, flags=DEFAULT_OUTPUT_CONTENT, no_output_families=(), owner_wildcards=None,
                      geo_wildcards=None):
    """DEMSolver::WriteSphereFile -> writeSpheresAsCsv (dT.cpp:1254-1405): one row per sphere component.
    owner_wildcards / geo_wildcards: {name: per-owner array} / {name: per-sphere array}, written after the family column when
    the OWNER_WILDCARD / GEO_WILDCARD bits are set (dT.cpp:1292-1301, 1386-1398).  The owner value of a sphere's row is its
    OWNER's (the reference indexes the owner array with the sphere id there, dT.cpp:1390 -- not reproduced)."""
    owner = np.asarray(arrays["ownerClumpBody"], np.int64)
    keep = ~np.isin(state["familyID"][owner], list(no_output_families))
    owner = owner[keep]
    comp = np.asarray(arrays["clumpComponentOffset"], np.int64)[keep]
    com = decode_positions_f32(state, p)
    rel = np.stack([arrays["CDRelPosX"], arrays["CDRelPosY"], arrays["CDRelPosZ"]], 1).astype(np.float32)[comp]
    pos = (com[owner] + rotate_f32(_quat(state)[owner], rel)).astype(np.float32)
    header = ["X", "Y", "Z", "r"]
    cols = [pos[:, 0], pos[:, 1], pos[:, 2], np.asarray(arrays["Radii"], np.float32)[comp]]
    names, extra = _owner_columns(flags, state, owner, 6)
    if flags & OUTPUT_CONTENT.OWNER_WILDCARD:
        for name, arr in (owner_wildcards or {}).items():
            names.append(name), extra.append(np.asarray(arr, np.float32)[owner])
    if flags & OUTPUT_CONTENT.GEO_WILDCARD:
        for name, arr in (geo_wildcards or {}).items():
            names.append(name), extra.append(np.asarray(arr, np.float32)[keep])
    _write_table(path, header + names, cols + extra, 6)
    return len(owner)


def write_clump_file(path, p, arrays, counts, state, template_names, flags=DEFAULT_OUTPUT_CONTENT, accuracy=10,
                     no_output_families=(), owner_wildcards=None):
    """DEMSolver::WriteClumpFile -> writeClumpsAsCsv (dT.cpp:1491-1618): one row per clump owner; xyz, quaternion
    and clump_type are always written; owner wildcards ({name: per-owner array}) follow the family column when the
    OWNER_WILDCARD bit is set (dT.cpp:1526-1530, 1605-1611)."""
    n = int(counts["nOwnerClumps"])
    owners = np.arange(n)
    owners = owners[~np.isin(state["familyID"][:n], list(no_output_families))]
    com = decode_positions_f32(state, p)[owners]
    q = _quat(state)[owners]
    marks = np.asarray(arrays["inertiaPropOffsets"], np.int64)[owners]
    header = ["X", "Y", "Z", "Qw", "Qx", "Qy", "Qz", "clump_type"]
    cols = [com[:, 0], com[:, 1], com[:, 2], q[:, 0], q[:, 1], q[:, 2], q[:, 3], [template_names[int(m)] for m in marks]]
    names, extra = _owner_columns(flags, state, owners, accuracy)
    if flags & OUTPUT_CONTENT.OWNER_WILDCARD:
        for name, arr in (owner_wildcards or {}).items():
            names.append(name), extra.append(np.asarray(arr, np.float32)[owners])
    _write_table(path, header + names, cols + extra, accuracy, text_cols=(7,))
    return len(owners)


def contact_info(p, arrays, counts, state, contacts, records, wildcards, flags=DEFAULT_CNT_OUTPUT_CONTENT, force_thres=1e-12):
    """generateContactInfo (dT.cpp:1620-1755): dict of columns for the contacts whose |force + torque-only force|
    reaches force_thres.  contacts = (idA, idB, type, _); records = (force, torque_only, cpA_local, cpB_local) as
    returned by Context.contact_records(); wildcards = {name: array}."""
    idA, idB, ctype = (np.asarray(x) for x in contacts[:3])
    F, T, cpA = (np.asarray(x, np.float32) for x in records[:3])
    keep = _len3(F + T) >= np.float32(force_thres)
    idA, idB, ctype, F, T, cpA = idA[keep], idB[keep], ctype[keep], F[keep], T[keep], cpA[keep]
    ownerA = np.asarray(arrays["ownerClumpBody"], np.int64)[idA]
    ownerB = np.zeros(len(idA), np.int64)
    ss, sm = ctype == 1, ctype == 2
    sa = ~(ss | sm)
    ownerB[ss] = np.asarray(arrays["ownerClumpBody"], np.int64)[idB[ss]]
    if sm.any():
        ownerB[sm] = np.asarray(arrays["ownerMesh"], np.int64)[idB[sm]]
    if sa.any():
        ownerB[sa] = np.asarray(arrays["objOwner"], np.int64)[idB[sa]]
    out = {"contact_type": [CONTACT_TYPE_NAME.get(int(t), "SA") for t in ctype]}
    if flags & CNT_OUTPUT_CONTENT.OWNER:
        out["A"], out["B"] = ownerA, ownerB
    if flags & CNT_OUTPUT_CONTENT.GEO_ID:
        out["geoA"], out["geoB"] = idA.astype(np.int64), idB.astype(np.int64)
    if flags & CNT_OUTPUT_CONTENT.FORCE:
        out["f_x"], out["f_y"], out["f_z"] = F[:, 0], F[:, 1], F[:, 2]
    q = _quat(state)[ownerA]
    com = decode_positions_f32(state, p)[ownerA]
    pnt = (rotate_f32(q, cpA) + com).astype(np.float32)
    if flags & CNT_OUTPUT_CONTENT.CNT_POINT:
        out["X"], out["Y"], out["Z"] = pnt[:, 0], pnt[:, 1], pnt[:, 2]
    if flags & CNT_OUTPUT_CONTENT.NORMAL:  # contact point - sphere A centre, normalised (rsqrtf form)
        comp = np.asarray(arrays["clumpComponentOffset"], np.int64)[idA]
        rel = np.stack([arrays["CDRelPosX"], arrays["CDRelPosY"], arrays["CDRelPosZ"]], 1).astype(np.float32)[comp]
        d = (pnt - (com + rotate_f32(q, rel)).astype(np.float32)).astype(np.float32)
        nrm = (d * (np.float32(1.0) / _len3(d))[:, None]).astype(np.float32)
        out["n_x"], out["n_y"], out["n_z"] = nrm[:, 0], nrm[:, 1], nrm[:, 2]
    if flags & CNT_OUTPUT_CONTENT.TORQUE:  # torque of the torque-only force about A's centre, global frame
        qc = q * np.array([1, -1, -1, -1], np.float32)
        tl = np.cross(cpA, rotate_f32(qc, T)).astype(np.float32)
        tg = rotate_f32(q, tl)
        out["torque_x"], out["torque_y"], out["torque_z"] = tg[:, 0], tg[:, 1], tg[:, 2]
    if flags & CNT_OUTPUT_CONTENT.CNT_WILDCARD:
        for name, arr in wildcards.items():
            out[name] = np.asarray(arr, np.float32)[keep]
    return out


def write_contact_file(path, p, arrays, counts, state, contacts, records, wildcards, flags=DEFAULT_CNT_OUTPUT_CONTENT,
                       force_thres=1e-12, precision=6):
    """DEMSolver::WriteContactFile -> writeContactsAsCsv (dT.cpp:1757-1848)."""
    info = contact_info(p, arrays, counts, state, contacts, records, wildcards, flags, force_thres)
    header = list(info.keys())
    _write_table(path, header, [info[k] for k in header], precision, text_cols=(0,))
    return len(info["contact_type"])


# ---- readers (static members of DEMSolver, API.h:1153-1250) ---------------------------------------------------
def write_contact_file_including_potential_pairs(path, p, arrays, counts, state, contacts, records, wildcards,
                                                 flags=DEFAULT_CNT_OUTPUT_CONTENT):
    """WriteContactFileIncludingPotentialPairs (API.h:1110-1116): force threshold -1, i.e. every pair of the list"""
    return write_contact_file(path, p, arrays, counts, state, contacts, records, wildcards, flags=flags, force_thres=-1.0)


def _read_csv(path):
    with open(path) as f:
        rows = [r for r in csv.reader(line for line in f if line.strip() and not line.lstrip().startswith("#"))]
    header = [h.strip() for h in rows[0]]
    return header, [[c.strip() for c in r] for r in rows[1:]]


def _by_type(path, cols, type_col="clump_type"):
    header, rows = _read_csv(path)
    idx = [header.index(c) for c in cols]
    t = header.index(type_col)
    out = {}
    for r in rows:
        out.setdefault(r[t], []).append([np.float32(r[i]) for i in idx])
    return {k: np.asarray(v, np.float32) for k, v in out.items()}


def read_clump_xyz_from_csv(path):
    """ReadClumpXyzFromCsv: {clump type name: [n,3]}"""
    return _by_type(path, ("X", "Y", "Z"))


def read_clump_vel_from_csv(path):
    return _by_type(path, ("v_x", "v_y", "v_z"))


def read_clump_angvel_from_csv(path):
    return _by_type(path, ("w_x", "w_y", "w_z"))


def read_clump_quat_from_csv(path):
    """ReadClumpQuatFromCsv: {clump type name: [n,4] as (x, y, z, w)} (the float4 the reference fills)."""
    q = _by_type(path, ("Qx", "Qy", "Qz", "Qw"))
    return q


def read_contact_pairs_from_csv(path, cnt_type="SS", cnt_col="contact_type", first="geoA", second="geoB"):
    """ReadContactPairsFromCsv: geometry-id pairs of one contact type."""
    header, rows = _read_csv(path)
    t, a, b = header.index(cnt_col), header.index(first), header.index(second)
    return np.asarray([[int(r[a]), int(r[b])] for r in rows if r[t] == cnt_type], np.uint32).reshape(-1, 2)


def read_contact_wildcards_from_csv(path, cnt_type="SS", cnt_col="contact_type"):
    """ReadContactWildcardsFromCsv: every column whose name is not a known contact-file column is taken as a
    wildcard -- which, as in the reference, includes the contact point columns X, Y, Z (they are not in
    CNT_FILE_KNOWN_COL_NAMES)."""
    header, rows = _read_csv(path)
    t = header.index(cnt_col)
    names = [h for h in header if h not in CNT_FILE_KNOWN_COL_NAMES]
    return {n: np.asarray([np.float32(r[header.index(n)]) for r in rows if r[t] == cnt_type], np.float32) for n in names}


def read_clump_template_csv(path):
    """LoadClumpType(..., "file.csv") component list: x,y,z,r rows, '#' comment lines (data/clumps/*.csv)."""
    header, rows = _read_csv(path)
    ix = [header.index(c) for c in ("x", "y", "z", "r")]
    a = np.asarray([[np.float32(r[i]) for i in ix] for r in rows], np.float32)
    return a[:, :3].copy(), a[:, 3].copy()


def read_obj(path):
    """AddWavefrontMeshObject's input (API.h:638-645; the reference reads it through tinyobjloader): `v x y z` vertices and
    `f` faces whose corners are `i`, `i/t`, `i//n` or `i/t/n` (1-based, negative = relative to the end); polygons are
    fanned into triangles.  Returns (vertices float32 (n, 3), faces int64 (m, 3)); normals / texture coordinates / groups are
    ignored, like the reference ignores them for contact purposes."""
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            parts = line.split()
            if not parts or parts[0].startswith("#"):
                continue
            if parts[0] == "v":
                verts.append([float(x) for x in parts[1:4]])
            elif parts[0] == "f":
                idx = []
                for c in parts[1:]:
                    i = int(c.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    v = np.asarray(verts, np.float32).reshape(-1, 3)
    fa = np.asarray(faces, np.int64).reshape(-1, 3)
    if len(fa) and (fa.min() < 0 or fa.max() >= len(v)):
        raise ValueError(f"{path}: face index out of range")
    return v, fa


def write_mesh_file(path, p, meshes, counts, state):
    """DEMSolver::WriteMeshFile -> writeMeshesAsVtk (dT.cpp:1850-1935): every mesh of the scene in one legacy-VTK
    UNSTRUCTURED_GRID file -- POINTS (owner-local vertices rotated by the owner's quaternion, then translated to its position),
    CELLS ("3 i j k" per facet, vertex indices offset per mesh), CELL_TYPES (5 = triangle).  `meshes`: the builder's MeshObj list
    (vertices (n, 3), faces (m, 3)); mesh owners are the last len(meshes) owners."""
    n_owners = int(counts["nOwners"])
    com = decode_positions_f32(state, p)
    q = _quat(state)
    lines = ["# vtk DataFile Version 2.0", "VTK from DEM simulation", "ASCII", "", "", "DATASET UNSTRUCTURED_GRID"]
    total_v = sum(len(m.vertices) for m in meshes)
    total_f = sum(len(m.faces) for m in meshes)
    lines.append(f"POINTS {total_v} float")
    offs, off = [], 0
    for i, m in enumerate(meshes):
        owner = n_owners - len(meshes) + i
        v = np.asarray(m.vertices, np.float32).reshape(-1, 3)
        w = (rotate_f32(np.repeat(q[owner][None, :], len(v), 0), v) + com[owner]).astype(np.float32)
        lines += [f"{_g(a)} {_g(b)} {_g(c)}" for a, b, c in w]
        offs.append(off)
        off += len(v)
    lines += ["", "", f"CELLS {total_f} {4 * total_f}"]
    for m, o in zip(meshes, offs):
        lines += [f"3 {int(a) + o} {int(b) + o} {int(c) + o}" for a, b, c in np.asarray(m.faces, np.int64).reshape(-1, 3)]
    lines += ["", "", f"CELL_TYPES {total_f}"] + ["5 "] * total_f
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return total_v, total_f
