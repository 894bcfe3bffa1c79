"""Spatial slab decomposition for multi-GPU runs (no reference equivalent: the
reference only splits kT/dT across two devices, DEM/APIPublic.cpp:22-72).

The global scene (arrays of SceneBuilder.Initialize()) is cut into N slabs along x.  A rank's scene is
[its own clumps | ghosts from the left neighbour | ghosts from the right neighbour | analytical owners].
Ghosts are copies of neighbour-owned clumps whose centre lies within `halo` of the shared face; they are marked per owner
(DemeScene.ownerGhost: never integrated locally, left out of inspections), keep their TRUE family -- contact masks and family
margins hold across a cut exactly as inside a slab -- and are refreshed every step from their owner rank (56-byte ghost
records: pose, velocities, family; deme_halo_pack / deme_halo_unpack).  A local-ghost contact is evaluated on both ranks,
each applying the force to its own clump only; ghost-ghost pairs are left to the ranks that own the clumps.

Ownership and ghost lists are fixed between re-decompositions, so a clump must not drift further than `halo`
minus its reach before the next one.  `redecompose` (every few thousand steps, or when the maximum displacement
since the last one approaches that bound) gathers the owned state and the contact history of all ranks, cuts the
domain again at the current positions and re-seeds every rank's history through deme_seed_contacts -- clumps
migrate between ranks there, with their contact wildcards.
"""
import numpy as np

from . import abi


_OWNER_KEYS = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ",
               "omgBarX", "omgBarY", "omgBarZ", "familyID", "inertiaPropOffsets")
_SPHERE_KEYS = ("ownerClumpBody", "clumpComponentOffset", "sphereMaterialOffset")


def slab_edges(x, n_ranks):
    """Equal-count slab boundaries along x (quantiles), so every rank owns ~the same number of clumps."""
    qs = np.quantile(np.asarray(x, np.float64), np.linspace(0, 1, n_ranks + 1))
    qs[0], qs[-1] = -np.inf, np.inf
    return qs


def _ranges(starts, counts):
    """concatenation of arange(starts[i], starts[i] + counts[i]) without a Python loop (millions of clumps per rank)"""
    starts, counts = np.asarray(starts, np.int64), np.asarray(counts, np.int64)
    total = int(counts.sum())
    if total == 0:
        return np.zeros(0, np.int64)
    first = np.cumsum(counts) - counts  # where each run begins in the output
    return np.repeat(starts - first, counts) + np.arange(total, dtype=np.int64)


def decompose(arrays, counts, clump_x, n_ranks, halo, shared_free=False, edges=None, only_rank=None):
    """Split a global scene.  clump_x: x of every clump centre (world frame).  Returns one dict per rank:
    arrays, counts, n_own, global_ids (own clumps' global owner ids), send/recv id lists (local owner ids).
    shared_free: allow replicated owners that move under contact forces (see below); the slabs must then be stepped by
    abi.HaloGroup (deme_halo_group_step), which adds their accelerations up across the slabs every step.
    edges / only_rank: the scene holds ONE slab of a larger bed (that rank's clumps and its ghosts, in the larger bed's order --
    model.packed_bed(slab=...)): the boundaries of the whole bed are given, only that rank's part is built (the others are None),
    and `global_ids` / ghost ids are indices into the scene given, not into the larger bed."""
    n_clumps = int(counts["nOwnerClumps"])
    n_owners = int(counts["nOwners"])
    # Analytical owners and meshes are replicated on every rank.  That is exact as it stands when the owner's motion does not
    # depend on the contact forces it receives (fixed, or every velocity component dictated by a prescription): each rank applies
    # the owner's force to its own clumps, and the owner's own a/alpha are never used.  A replicated owner that moves under
    # contact forces is flagged (ownerGhost bit 1): every rank sums the contributions of the spheres it owns -- a ghost sphere's
    # contact with such an owner is left to the sphere's own rank -- and the per-rank sums are all-reduced before the integration,
    # so that every replica takes the same step.
    flags = np.asarray(arrays["familyFlags"])
    fam = np.asarray(arrays["familyID"])
    free = [int(o) for o in range(n_clumps, n_owners) if not (flags[fam[o]] & (abi.FAMILY_FIXED | abi.FAMILY_PRESCRIBED))]
    if free and not shared_free:
        raise ValueError(f"replicated owner(s) {free} (meshes / analytical bodies) move under contact forces: decompose(..., "
                         "shared_free=True) and step the slabs with abi.HaloGroup, which all-reduces their accelerations every step")
    x = np.asarray(clump_x, np.float64)[:n_clumps]
    if edges is None:
        edges = slab_edges(x, n_ranks)
    edges = np.asarray(edges, np.float64)
    # ghosts are taken from the face neighbours only: an interior slab thinner than the halo would leave clumps of the slab
    # after next within reach of the face without a ghost copy (equal-count slabs get thin where the bed is dense)
    widths = np.diff(edges)[1:-1]
    if len(widths) and float(widths.min()) < halo:
        raise ValueError(f"slab {1 + int(widths.argmin())} is {float(widths.min()):.4g} wide, thinner than the halo {halo:.4g}: "
                         "use fewer ranks or a thinner halo")
    rank_of = np.clip(np.searchsorted(edges, x, side="right") - 1, 0, n_ranks - 1)
    sph_owner_g = arrays["ownerClumpBody"]
    first_sphere = np.searchsorted(sph_owner_g, np.arange(n_owners + 1))  # spheres are clump-major
    extra_owners = np.arange(n_clumps, n_owners)  # analytical and mesh owners, kept on every rank
    out = []
    for r in range(n_ranks):
        if only_rank is not None and r != only_rank:
            out.append(None)
            continue
        own = np.nonzero(rank_of == r)[0]
        gl = np.nonzero((rank_of == r - 1) & (x >= edges[r] - halo))[0] if r > 0 else np.zeros(0, np.int64)
        gr = np.nonzero((rank_of == r + 1) & (x < edges[r + 1] + halo))[0] if r < n_ranks - 1 else np.zeros(0, np.int64)
        owners_g = np.concatenate([own, gl, gr, extra_owners]).astype(np.int64)
        new_id = np.full(n_owners, -1, np.int64)
        new_id[owners_g] = np.arange(len(owners_g))
        a = dict(arrays)
        for k in _OWNER_KEYS:
            a[k] = arrays[k][owners_g].copy()
        ghost = np.zeros(len(owners_g), np.uint8)
        ghost[len(own):len(own) + len(gl) + len(gr)] = 1
        if free:
            ghost[new_id[np.asarray(free, np.int64)]] = 2  # replicated free owners: summed across slabs (OWNER_SHARED_BIT)
        a["ownerGhost"] = ghost
        clumps_here = owners_g[:len(own) + len(gl) + len(gr)]
        sph_idx = _ranges(first_sphere[clumps_here], first_sphere[clumps_here + 1] - first_sphere[clumps_here])
        for k in _SPHERE_KEYS:
            a[k] = arrays[k][sph_idx].copy()
        a["ownerClumpBody"] = new_id[arrays["ownerClumpBody"][sph_idx]].astype(np.uint32)
        a["objOwner"] = new_id[arrays["objOwner"]].astype(np.uint32)
        if int(counts.get("nTri", 0)):
            a["ownerMesh"] = new_id[arrays["ownerMesh"]].astype(np.uint32)
        c = dict(counts)
        c.update({"nOwners": len(owners_g), "nOwnerClumps": len(clumps_here), "nSpheres": len(sph_idx)})
        out.append({"arrays": a, "counts": c, "n_own": len(own), "global_ids": own, "ghost_left_g": gl, "ghost_right_g": gr,
                    "new_id": new_id, "edges": (edges[r], edges[r + 1]), "all_edges": edges, "sphere_global": sph_idx.astype(np.int64),
                    "owner_global": owners_g})
    # send lists: what my neighbour holds as ghosts, in the neighbour's slot order (ascending global id on both sides)
    for r in range(n_ranks):
        me = out[r]
        if me is None:
            continue
        me["recv_left"] = np.arange(me["n_own"], me["n_own"] + len(me["ghost_left_g"]), dtype=np.uint32)
        me["recv_right"] = np.arange(me["n_own"] + len(me["ghost_left_g"]),
                                     me["n_own"] + len(me["ghost_left_g"]) + len(me["ghost_right_g"]), dtype=np.uint32)
        # what the neighbour holds as ghosts from me: my own clumps within the halo of the shared face (the neighbour's rule,
        # evaluated here so that a rank needs nobody else's part)
        to_left = np.nonzero((rank_of == r) & (x < edges[r] + halo))[0] if r > 0 else np.zeros(0, np.int64)
        to_right = np.nonzero((rank_of == r) & (x >= edges[r + 1] - halo))[0] if r < n_ranks - 1 else np.zeros(0, np.int64)
        me["send_left"] = me["new_id"][to_left].astype(np.uint32)
        me["send_right"] = me["new_id"][to_right].astype(np.uint32)
        assert (me["send_left"] < me["n_own"]).all() and (me["send_right"] < me["n_own"]).all()
        if r > 0 and out[r - 1] is not None:
            assert np.array_equal(to_left, out[r - 1]["ghost_right_g"])
        if r < n_ranks - 1 and out[r + 1] is not None:
            assert np.array_equal(to_right, out[r + 1]["ghost_left_g"])
    for me in out:
        if me is None:
            continue
        me["scene"] = abi.make_scene_struct(me["arrays"], me["counts"])
        del me["new_id"]
    return out


def decompose_lib(params, scene, n_ranks, halo, axis=0, edges=None, shared_free=False, snap=False, spatial_order=False):
    """The same decomposition computed by the LIBRARY (deme_decomp_create: what a C++ host calls; DEMSolver(nGPUs) goes through it).
    Returns (plan, parts) with parts[r] holding the keys of decompose() that the halo group and the migration books read: scene,
    counts, n_own, global_ids, ghost_left_g / ghost_right_g, owner_global, sphere_global, send / recv lists, edges.  Keep `plan`
    alive while a slab scene is in use (its arrays belong to the plan)."""
    plan = abi.DecompPlan(params, scene, n_ranks, axis=axis, halo=halo, edges=edges, shared_free=shared_free, snap=snap,
                          spatial_order=spatial_order)
    parts = []
    for r in range(n_ranks):
        q = plan.slab(r)
        n_own, nl, nu = q["n_own"], q["n_ghost_lower"], q["n_ghost_upper"]
        og = q["owner_global"].astype(np.int64)
        sc = q["scene"]
        parts.append({
            "scene": sc, "counts": {"nOwners": int(sc.nOwners), "nOwnerClumps": int(sc.nOwnerClumps), "nSpheres": int(sc.nSpheres)},
            "n_own": n_own, "global_ids": og[:n_own], "ghost_left_g": og[n_own:n_own + nl], "ghost_right_g": og[n_own + nl:n_own + nl + nu],
            "owner_global": og, "sphere_global": q["sphere_global"].astype(np.int64), "edges": q["range"], "all_edges": plan.edges,
            "send_left": q["send_lower"], "send_right": q["send_upper"],
            "recv_left": np.arange(n_own, n_own + nl, dtype=np.uint32), "recv_right": np.arange(n_own + nl, n_own + nl + nu, dtype=np.uint32)})
    return plan, parts


GHOST_STATE_KEYS = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ",
                    "omgBarX", "omgBarY", "omgBarZ", "familyID")


def exchange_host(parts, states):
    """Reference (host-memory) halo exchange between in-process ranks: states[r] = dict of per-owner arrays.
    Used by tests; the GPU path packs/unpacks on the device and moves the records with RCCL."""
    n = len(parts)
    for r in range(n):
        for nb, send_key, recv_key in ((r - 1, "send_left", "recv_right"), (r + 1, "send_right", "recv_left")):
            if nb < 0 or nb >= n:
                continue
            src, dst = parts[r][send_key], parts[nb][recv_key]
            assert len(src) == len(dst)
            for k in GHOST_STATE_KEYS:
                states[nb][k][dst] = states[r][k][src]


# ---- re-decomposition (migration of clumps and of their contact history between ranks) --------------------------
def _globalise_pairs(part, idA, idB, ctype):
    """local (A, B, type) pairs -> global sphere ids, stored (smaller, larger) for sphere-sphere pairs.  Returns
    (mine, gA, gB, flipped): `mine` marks the pairs this rank reports (it owns the clump of the globally smaller sphere;
    sphere-analytical / sphere-mesh pairs go with sphere A's owner), `flipped` the pairs whose local order was the other way."""
    local_owner = np.asarray(part["arrays"]["ownerClumpBody"], np.int64)
    sg = part["sphere_global"]
    ss = ctype == 1
    gA = sg[idA]
    gB = np.where(ss, sg[np.where(ss, idB, 0)], idB.astype(np.int64))  # analytical component / triangle ids are global already
    flipped = ss & (gA > gB)
    lo_local = np.where(flipped, idB, idA)  # local id of the sphere that is sphere A in the global numbering
    mine = local_owner[np.where(ss, lo_local, idA)] < part["n_own"]
    return mine, np.where(flipped, gB, gA), np.where(flipped, gA, gB), flipped


def owned_payload(part, state, contacts, wildcards, flip_sign_wildcards=(0, 1, 2), persistent=None, owner_wildcards=None,
                  sphere_wildcards=None):
    """What one rank contributes to a re-decomposition: the state of its OWN clumps and its share of the contact
    history in global ids.  A sphere-sphere pair is reported once, by the rank that owns the clump of the globally
    smaller sphere id, stored (smaller, larger); where the local numbering had it the other way round the B-to-A
    vector wildcards (flip_sign_wildcards) change sign.  Sphere-analytical contacts go with sphere A's owner.
    contacts = (idA, idB, type[, map]) local; wildcards = float[n, nW].  Optional: persistent = (idA, idB, type) of the marked
    contacts (Context.persistent_contacts()); owner_wildcards / sphere_wildcards = {name: per-local-owner / per-local-sphere
    array} of a user force model -- the values of this rank's own clumps travel, replicated owners (walls, meshes) keep
    whatever each rank holds."""
    n_own = part["n_own"]
    own_state = {k: np.asarray(state[k])[:n_own].copy() for k in GHOST_STATE_KEYS}
    idA, idB, ctype = (np.asarray(x) for x in contacts[:3])
    W = np.asarray(wildcards, np.float32).reshape(len(idA), -1)
    mine, gA, gB, flipped = _globalise_pairs(part, idA, idB, ctype)
    Wm = W[mine].copy()
    fm = flipped[mine]
    for k in flip_sign_wildcards:
        if k < Wm.shape[1]:
            Wm[fm, k] = -Wm[fm, k]
    out = {"global_ids": np.asarray(part["global_ids"], np.int64), "state": own_state, "gA": gA[mine], "gB": gB[mine],
           "type": ctype[mine], "wc": Wm}
    # replicated owners (walls, meshes, analytical bodies; the same on every rank): their CURRENT state travels too, or a moving
    # one would jump back to its pose at the previous decomposition
    n_cl_here = int(part["counts"]["nOwnerClumps"])
    out["replicated_global"] = np.asarray(part["owner_global"], np.int64)[n_cl_here:].copy()
    out["replicated_state"] = {k: np.asarray(state[k])[n_cl_here:].copy() for k in GHOST_STATE_KEYS}
    if persistent is not None:
        pA, pB, pT = (np.asarray(x) for x in persistent[:3])
        pm, pgA, pgB, _ = _globalise_pairs(part, pA, pB, pT)
        out["persistent"] = (pgA[pm], pgB[pm], pT[pm])
    local_owner = np.asarray(part["arrays"]["ownerClumpBody"], np.int64)
    n_own_sph = int((local_owner < n_own).sum())  # spheres are clump-major and a rank's own clumps come first
    n_clumps_here = int(part["counts"]["nOwnerClumps"])
    if owner_wildcards:
        out["owner_wc"] = {k: np.asarray(v, np.float32)[:n_own].copy() for k, v in owner_wildcards.items()}
        out["owner_wc_replicated"] = {k: np.asarray(v, np.float32)[n_clumps_here:].copy() for k, v in owner_wildcards.items()}
    if sphere_wildcards:
        out["sphere_wc"] = {k: np.asarray(v, np.float32)[:n_own_sph].copy() for k, v in sphere_wildcards.items()}
        out["sphere_global_own"] = np.asarray(part["sphere_global"], np.int64)[:n_own_sph].copy()
    return out


def _localise_pairs(part, n_sph_global, gA, gB, ty):
    """global pairs -> this part's local sphere ids: (keep, la, lb, flip) for the pairs with an owned clump on either side"""
    loc = np.full(n_sph_global, -1, np.int64)
    loc[part["sphere_global"]] = np.arange(len(part["sphere_global"]))
    owner_local = np.asarray(part["arrays"]["ownerClumpBody"], np.int64)
    a = loc[gA]
    ss = ty == 1
    b = np.where(ss, loc[np.where(ss, gB, 0)], gB)
    present = (a >= 0) & (b >= 0)
    own_a = np.zeros(len(a), bool)
    own_a[present] = owner_local[a[present]] < part["n_own"]
    own_b = np.zeros(len(a), bool)
    sel = present & ss
    own_b[sel] = owner_local[b[sel]] < part["n_own"]
    keep = present & (own_a | own_b)  # ghost-ghost pairs and contacts of foreign clumps stay with their owners
    # a sphere-sphere pair is stored with the smaller sphere id first (DEMContactKernels_SphereSphere.cu:199-207):
    # the local numbering may flip it
    la, lb = a[keep].copy(), b[keep].copy()
    flip = (ty[keep] == 1) & (la > lb)
    la[flip], lb[flip] = lb[flip], la[flip].copy()
    return keep, la, lb, flip


def redecompose(global_arrays, counts, payloads, n_ranks, halo, decode_x, flip_sign_wildcards=(0, 1, 2), shared_free=False, edges=None):
    """payloads: owned_payload() of every rank (in-process list, or the result of an all_gather_object).
    decode_x(arrays) -> world x of every clump centre.  Returns (global_arrays_now, new_parts, seeds) where
    seeds[r] = (idA, idB, type, wildcards) in rank r's new local sphere ids, ready for seed_contacts().
    flip_sign_wildcards: wildcards that are vectors from B to A (the Hertzian model's delta_tan_x/y/z, indices 0-2 in
    its alphabetical order): they change sign when a rank's local numbering stores the pair the other way round.
    When the payloads carry them, new_parts[r] also gets "persistent" = (idA, idB, type) for set_persistent_contacts() and
    "owner_wc" / "sphere_wc" = {name: array in rank r's new local numbering} for set_wildcard_array() (ghost copies included:
    the force model reads B's owner through them)."""
    g = dict(global_arrays)
    for k in GHOST_STATE_KEYS:
        g[k] = np.array(global_arrays[k], copy=True)
    for pl in payloads:
        for k in GHOST_STATE_KEYS:
            g[k][pl["global_ids"]] = pl["state"][k]
    if payloads and "replicated_state" in payloads[0]:  # every rank holds the same replicas: rank 0's copy speaks for all
        for k in GHOST_STATE_KEYS:
            g[k][payloads[0]["replicated_global"]] = payloads[0]["replicated_state"][k]
    parts = decompose(g, counts, decode_x(g), n_ranks, halo, shared_free=shared_free, edges=edges)
    gA = np.concatenate([pl["gA"] for pl in payloads])
    gB = np.concatenate([pl["gB"] for pl in payloads])
    ty = np.concatenate([pl["type"] for pl in payloads])
    wc = np.concatenate([pl["wc"] for pl in payloads]) if payloads else np.zeros((0, 0), np.float32)
    n_sph_global = len(global_arrays["ownerClumpBody"])
    n_clumps_global = int(counts["nOwnerClumps"])
    seeds = []
    for part in parts:
        keep, la, lb, flip = _localise_pairs(part, n_sph_global, gA, gB, ty)
        w = wc[keep].copy()
        for k in flip_sign_wildcards:
            if k < w.shape[1]:
                w[flip, k] = -w[flip, k]
        seeds.append((la.astype(np.uint32), lb.astype(np.uint32), ty[keep].astype(np.uint8), w))
    if payloads and all("persistent" in pl for pl in payloads):
        pA = np.concatenate([pl["persistent"][0] for pl in payloads])
        pB = np.concatenate([pl["persistent"][1] for pl in payloads])
        pT = np.concatenate([pl["persistent"][2] for pl in payloads])
        for part in parts:
            keep, la, lb, _ = _localise_pairs(part, n_sph_global, pA, pB, pT)
            part["persistent"] = (la.astype(np.uint32), lb.astype(np.uint32), pT[keep].astype(np.uint8))
    if payloads and all("owner_wc" in pl for pl in payloads):
        for name in payloads[0]["owner_wc"]:
            glob = np.zeros(n_clumps_global, np.float32)
            for pl in payloads:
                glob[pl["global_ids"]] = pl["owner_wc"][name]
            for r, part in enumerate(parts):
                n_here = int(part["counts"]["nOwnerClumps"])
                arr = np.concatenate([glob[part["owner_global"][:n_here]], payloads[r]["owner_wc_replicated"][name]])
                part.setdefault("owner_wc", {})[name] = arr.astype(np.float32)
    if payloads and all("sphere_wc" in pl for pl in payloads):
        for name in payloads[0]["sphere_wc"]:
            glob = np.zeros(n_sph_global, np.float32)
            for pl in payloads:
                glob[pl["sphere_global_own"]] = pl["sphere_wc"][name]
            for part in parts:
                part.setdefault("sphere_wc", {})[name] = glob[part["sphere_global"]].astype(np.float32)
    return g, parts, seeds


def redecompose_distributed(dist, rank, world, global_arrays, counts, part, state, contacts, wildcards, halo, decode_x, **kw):
    """One process per rank: all-gather the owned payloads (torch.distributed object collective -- a few tens of MB
    per million clumps, amortised over the thousands of steps between re-decompositions), cut again, and return
    this rank's (global arrays, new part, seed)."""
    payloads = [None] * world
    dist.all_gather_object(payloads, owned_payload(part, state, contacts, wildcards,
                                                   **{k: kw.pop(k) for k in ("persistent", "owner_wildcards", "sphere_wildcards") if k in kw}))
    g, parts, seeds = redecompose(global_arrays, counts, payloads, world, halo, decode_x, **kw)
    return g, parts[rank], seeds[rank]


# ---- neighbour-to-neighbour migration ("at re-bin: migration lists") ---------------------------------------------
# The all-gather above hands every rank the whole job's state: fine for tests and small jobs, O(N_total) per rank.  The functions
# below re-decompose with FIXED slab edges and face-neighbour traffic only: a clump whose centre crossed a face moves to that
# neighbour with its state, its template data and the history of its contacts; every rank then tells its neighbours which of its
# clumps lie within the halo of the shared face (the new ghost sets, with their data).  Per rank the work and the traffic are
# O(own clumps + boundary layer).  `transport(rank, to_left, to_right) -> (from_left, from_right)` moves picklable packets.
_STATIC_OWNER_KEYS = ("inertiaPropOffsets",)


def _first_sphere(part):
    own = np.asarray(part["arrays"]["ownerClumpBody"], np.int64)
    return np.searchsorted(own, np.arange(int(part["counts"]["nOwnerClumps"]) + 1))  # spheres are clump-major


def _clump_packet(part, state, local_ids, first_sphere):
    """everything another rank needs to hold these clumps (as its own or as ghosts): global ids, state, template data, spheres"""
    ids = np.asarray(local_ids, np.int64)
    a = part["arrays"]
    nsph = first_sphere[ids + 1] - first_sphere[ids]
    sidx = _ranges(first_sphere[ids], nsph)
    return {"gid": np.asarray(part["owner_global"], np.int64)[ids], "state": {k: np.asarray(state[k])[ids].copy() for k in GHOST_STATE_KEYS},
            "inertia": np.asarray(a["inertiaPropOffsets"])[ids].copy(), "nsph": nsph,
            "comp": np.asarray(a["clumpComponentOffset"])[sidx].copy(), "mat": np.asarray(a["sphereMaterialOffset"])[sidx].copy(),
            "sgid": np.asarray(part["sphere_global"], np.int64)[sidx]}


def _cat_packets(pk):
    pk = [q for q in pk if q is not None and len(q["gid"])]
    if not pk:
        return None
    out = {k: np.concatenate([q[k] for q in pk]) for k in ("gid", "inertia", "nsph", "comp", "mat", "sgid")}
    out["state"] = {k: np.concatenate([q["state"][k] for q in pk]) for k in GHOST_STATE_KEYS}
    return out


def _empty_packet(part):
    a = part["arrays"]
    return {"gid": np.zeros(0, np.int64), "inertia": np.zeros(0, np.asarray(a["inertiaPropOffsets"]).dtype), "nsph": np.zeros(0, np.int64),
            "comp": np.zeros(0, np.asarray(a["clumpComponentOffset"]).dtype), "mat": np.zeros(0, np.asarray(a["sphereMaterialOffset"]).dtype),
            "sgid": np.zeros(0, np.int64), "state": {k: np.zeros(0, abi.STATE_DTYPES[k]) for k in GHOST_STATE_KEYS}}


def migration_packets(part, state, contacts, wildcards, edges, rank, x_own, flip_sign_wildcards=(0, 1, 2)):
    """Phase 1, per rank.  x_own: current world x of the own clumps.  Returns (stay_ids, to_left, to_right, rows) where the two
    packets carry the clumps that crossed the left / right face together with the history of every contact they take part in, and
    rows = this rank's whole contact history in global sphere ids (gA, gB, type, wildcards; sphere pairs smaller id first)."""
    n_own = part["n_own"]
    x = np.asarray(x_own, np.float64)[:n_own]
    go_l = x < edges[rank]
    go_r = x >= edges[rank + 1]
    fs = _first_sphere(part)
    idA, idB, ctype = (np.asarray(v) for v in contacts[:3])
    W = np.asarray(wildcards, np.float32).reshape(len(idA), -1)
    sg = np.asarray(part["sphere_global"], np.int64)
    ss = ctype == 1
    gA = sg[idA]
    gB = np.where(ss, sg[np.where(ss, idB, 0)], idB.astype(np.int64))
    flip = ss & (gA > gB)
    Wg = W.copy()
    for k in flip_sign_wildcards:
        if k < Wg.shape[1]:
            Wg[flip, k] = -Wg[flip, k]
    rows = {"gA": np.where(flip, gB, gA), "gB": np.where(flip, gA, gB), "type": ctype.astype(np.uint8), "wc": Wg}
    local_owner = np.asarray(part["arrays"]["ownerClumpBody"], np.int64)
    oA = local_owner[idA]
    oB = np.where(ss, local_owner[np.where(ss, idB, 0)], -1)
    out = []
    for sel in (go_l, go_r):
        ids = np.nonzero(sel)[0]
        pk = _clump_packet(part, state, ids, fs) if len(ids) else _empty_packet(part)
        moving = np.zeros(int(part["counts"]["nOwners"]) + 1, bool)
        moving[ids] = True
        involved = moving[oA] | np.where(oB >= 0, moving[np.maximum(oB, 0)], False)
        pk["rows"] = {k: v[involved] for k, v in rows.items()}
        out.append(pk)
    stay = np.nonzero(~(go_l | go_r))[0]
    return stay, out[0], out[1], rows


def ghost_packets(own_packet, edges, rank, n_ranks, halo, x_own):
    """Phase 2, per rank: of the clumps this rank owns now (own_packet, x_own their world x), those within `halo` of the left /
    right face, as packets, plus their positions in own_packet (the send lists)."""
    x = np.asarray(x_own, np.float64)
    sel_l = np.nonzero(x < edges[rank] + halo)[0] if rank > 0 else np.zeros(0, np.int64)
    sel_r = np.nonzero(x >= edges[rank + 1] - halo)[0] if rank + 1 < n_ranks else np.zeros(0, np.int64)

    def sub(sel):
        first = np.cumsum(own_packet["nsph"]) - own_packet["nsph"]
        sidx = _ranges(first[sel], own_packet["nsph"][sel])
        return {"gid": own_packet["gid"][sel], "inertia": own_packet["inertia"][sel], "nsph": own_packet["nsph"][sel],
                "comp": own_packet["comp"][sidx], "mat": own_packet["mat"][sidx], "sgid": own_packet["sgid"][sidx],
                "state": {k: own_packet["state"][k][sel] for k in GHOST_STATE_KEYS}}
    return sel_l, sub(sel_l), sel_r, sub(sel_r)


def assemble_part(old_part, own_packet, send_l, send_r, ghosts_l, ghosts_r, rows_list, flip_sign_wildcards=(0, 1, 2), state=None):
    """New local scene of a rank: [own | ghosts from the left | ghosts from the right | replicated owners (walls, meshes)], its
    exchange lists, and the contact history to seed: every known row (own history first, then what arrived) whose spheres are
    present and that involves an own clump.  `state`: the old part's CURRENT per-owner state; the replicated owners take their
    rows from it (without it they would return to their pose at the previous decomposition).
    Returns (part, seed = (idA, idB, type, wildcards))."""
    a_old, c_old = old_part["arrays"], old_part["counts"]
    n_cl_old, n_ow_old = int(c_old["nOwnerClumps"]), int(c_old["nOwners"])
    packs = [own_packet, ghosts_l, ghosts_r]
    n_own, n_gl, n_gr = (len(q["gid"]) for q in packs)
    n_cl = n_own + n_gl + n_gr
    extras = np.arange(n_cl_old, n_ow_old)
    a = dict(a_old)
    rep_src = state if state is not None else a_old
    for k in GHOST_STATE_KEYS:
        a[k] = np.concatenate([q["state"][k] for q in packs] + [np.asarray(rep_src[k])[extras]]).astype(np.asarray(a_old[k]).dtype)
    a["inertiaPropOffsets"] = np.concatenate([q["inertia"] for q in packs] + [np.asarray(a_old["inertiaPropOffsets"])[extras]]).astype(
        np.asarray(a_old["inertiaPropOffsets"]).dtype)
    ghost = np.zeros(n_cl + len(extras), np.uint8)
    ghost[n_own:n_cl] = 1
    if "ownerGhost" in a_old:  # replicated free owners stay flagged
        ghost[n_cl:] = np.asarray(a_old["ownerGhost"])[extras] & 2
    a["ownerGhost"] = ghost
    nsph = np.concatenate([q["nsph"] for q in packs]).astype(np.int64)
    a["ownerClumpBody"] = np.repeat(np.arange(n_cl, dtype=np.uint32), nsph)
    a["clumpComponentOffset"] = np.concatenate([q["comp"] for q in packs]).astype(np.asarray(a_old["clumpComponentOffset"]).dtype)
    a["sphereMaterialOffset"] = np.concatenate([q["mat"] for q in packs]).astype(np.asarray(a_old["sphereMaterialOffset"]).dtype)
    shift = n_cl - n_cl_old
    a["objOwner"] = (np.asarray(a_old["objOwner"], np.int64) + shift).astype(np.uint32)
    if int(c_old.get("nTri", 0)):
        a["ownerMesh"] = (np.asarray(a_old["ownerMesh"], np.int64) + shift).astype(np.uint32)
    c = dict(c_old)
    c.update({"nOwners": n_cl + len(extras), "nOwnerClumps": n_cl, "nSpheres": int(nsph.sum())})
    sphere_global = np.concatenate([q["sgid"] for q in packs]).astype(np.int64)
    owner_global = np.concatenate([q["gid"] for q in packs] + [np.asarray(old_part["owner_global"], np.int64)[extras]])
    part = {"arrays": a, "counts": c, "n_own": n_own, "global_ids": own_packet["gid"].copy(), "ghost_left_g": ghosts_l["gid"].copy(),
            "ghost_right_g": ghosts_r["gid"].copy(), "edges": old_part["edges"], "all_edges": old_part.get("all_edges"),
            "sphere_global": sphere_global, "owner_global": owner_global,
            "send_left": np.asarray(send_l, np.uint32), "send_right": np.asarray(send_r, np.uint32),
            "recv_left": np.arange(n_own, n_own + n_gl, dtype=np.uint32), "recv_right": np.arange(n_own + n_gl, n_cl, dtype=np.uint32)}
    part["scene"] = abi.make_scene_struct(a, c)
    # history: rows in global ids -> local ids of the new numbering
    gA = np.concatenate([r["gA"] for r in rows_list])
    gB = np.concatenate([r["gB"] for r in rows_list])
    ty = np.concatenate([r["type"] for r in rows_list])
    wc = np.concatenate([r["wc"] for r in rows_list]) if rows_list else np.zeros((0, 0), np.float32)
    order = np.argsort(sphere_global, kind="stable")
    sorted_g = sphere_global[order]

    def loc(g):
        pos = np.searchsorted(sorted_g, g)
        pos = np.minimum(pos, len(sorted_g) - 1) if len(sorted_g) else pos
        ok = (sorted_g[pos] == g) if len(sorted_g) else np.zeros(len(g), bool)
        return np.where(ok, order[pos], -1)
    ss = ty == 1
    la = loc(gA)
    lb = np.where(ss, loc(np.where(ss, gB, sphere_global[0] if len(sphere_global) else 0)), gB)
    present = (la >= 0) & (lb >= 0)
    own_of = a["ownerClumpBody"].astype(np.int64)
    own_a = np.zeros(len(la), bool)
    own_a[present] = own_of[la[present]] < n_own
    own_b = np.zeros(len(la), bool)
    sel = present & ss
    own_b[sel] = own_of[lb[sel]] < n_own
    keep = present & (own_a | own_b)
    # one row per pair: the first occurrence wins (this rank's own history comes first in rows_list)
    key = np.stack([gA[keep], gB[keep], ty[keep].astype(np.int64)], 1)
    _, first = np.unique(key, axis=0, return_index=True)
    first.sort()
    la, lb, tk, w = la[keep][first], lb[keep][first], ty[keep][first], wc[keep][first].copy()
    flip = (tk == 1) & (la > lb)
    la2, lb2 = np.where(flip, lb, la), np.where(flip, la, lb)
    for k in flip_sign_wildcards:
        if k < w.shape[1]:
            w[flip, k] = -w[flip, k]
    return part, (la2.astype(np.uint32), lb2.astype(np.uint32), tk.astype(np.uint8), w)


def migrate_neighbours(rank, n_ranks, part, state, contacts, wildcards, edges, halo, decode_x, transport,
                       flip_sign_wildcards=(0, 1, 2)):
    """One rank's re-decomposition with face-neighbour traffic only (fixed slab edges).  decode_x(state-like dict) -> world x per
    owner.  transport(tag, to_left, to_right) -> (from_left, from_right) exchanges picklable packets with the face neighbours
    (None at the ends of the chain).  Returns (new part, seed) for upload_scene + seed_contacts."""
    x_all = decode_x(state)
    stay, to_l, to_r, rows = migration_packets(part, state, contacts, wildcards, edges, rank, x_all, flip_sign_wildcards)
    from_l, from_r = transport("migrants", to_l if rank > 0 else None, to_r if rank + 1 < n_ranks else None)
    if (rank == 0 and len(to_l["gid"])) or (rank + 1 == n_ranks and len(to_r["gid"])):
        raise ValueError("a clump left the decomposed range")
    fs = _first_sphere(part)
    own = _cat_packets([_clump_packet(part, state, stay, fs), from_l, from_r]) or _empty_packet(part)
    x_own = decode_x(own["state"])
    if ((x_own < edges[rank]) | (x_own >= edges[rank + 1])).any():
        raise ValueError("a clump moved further than one slab between two re-decompositions")
    sel_l, gp_l, sel_r, gp_r = ghost_packets(own, edges, rank, n_ranks, halo, x_own)
    gh_l, gh_r = transport("ghosts", gp_l if rank > 0 else None, gp_r if rank + 1 < n_ranks else None)
    rows_list = [rows] + [q["rows"] for q in (from_l, from_r) if q is not None]
    return assemble_part(part, own, sel_l, sel_r, gh_l or _empty_packet(part), gh_r or _empty_packet(part), rows_list, flip_sign_wildcards,
                         state=state)


def torch_transport(dist, rank, n_ranks):
    """transport for migrate_neighbours over torch.distributed point-to-point object messages (any backend): both directions
    of the chain in two sweeps, even ranks sending first"""
    def move(tag, to_left, to_right):
        got = {"l": None, "r": None}

        def send(obj, dst):
            dist.send_object_list([obj], dst=dst)

        def recv(src):
            box = [None]
            dist.recv_object_list(box, src=src)
            return box[0]
        for phase in (0, 1):
            if rank % 2 == phase:
                if rank + 1 < n_ranks:
                    send(to_right, rank + 1)
                    got["r"] = recv(rank + 1)
            else:
                if rank > 0:
                    got["l"] = recv(rank - 1)
                    send(to_left, rank - 1)
        return got["l"], got["r"]
    return move


def migrate_neighbours_in_process(parts, states, contacts, wildcards, edges, halo, decode_x, flip_sign_wildcards=(0, 1, 2)):
    """all slabs held by one process (tests, the one-GPU harness): the same per-rank functions with lists as the transport"""
    n = len(parts)
    phase1 = [migration_packets(parts[r], states[r], contacts[r], wildcards[r], edges, r, decode_x(states[r]), flip_sign_wildcards)
              for r in range(n)]
    owns, xs = [], []
    for r in range(n):
        stay, to_l, to_r, rows = phase1[r]
        if (r == 0 and len(to_l["gid"])) or (r + 1 == n and len(to_r["gid"])):
            raise ValueError("a clump left the decomposed range")
        from_l = phase1[r - 1][2] if r > 0 else None
        from_r = phase1[r + 1][1] if r + 1 < n else None
        own = _cat_packets([_clump_packet(parts[r], states[r], stay, _first_sphere(parts[r])), from_l, from_r]) or _empty_packet(parts[r])
        owns.append((own, from_l, from_r))
        xs.append(decode_x(own["state"]))
    gp = [ghost_packets(owns[r][0], edges, r, n, halo, xs[r]) for r in range(n)]
    out = []
    for r in range(n):
        own, from_l, from_r = owns[r]
        gh_l = gp[r - 1][3] if r > 0 else _empty_packet(parts[r])
        gh_r = gp[r + 1][1] if r + 1 < n else _empty_packet(parts[r])
        rows_list = [phase1[r][3]] + [q["rows"] for q in (from_l, from_r) if q is not None]
        out.append(assemble_part(parts[r], own, gp[r][0], gp[r][2], gh_l, gh_r, rows_list, flip_sign_wildcards, state=states[r]))
    return [o[0] for o in out], [o[1] for o in out]
