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


def decompose(arrays, counts, clump_x, n_ranks, halo):
    """Split a global scene.  clump_x: x of every clump centre (world frame).  Returns one dict per rank:
    arrays, counts, n_own, global_ids (own clumps' global owner ids), send/recv id lists (local owner ids)."""
    n_clumps = int(counts["nOwnerClumps"])
    n_owners = int(counts["nOwners"])
    # Meshes are replicated on every rank like the analytical owners.  That is exact when a mesh's motion does not depend on
    # the contact forces it receives (fixed, or every velocity component dictated by a prescription): each rank applies
    # the mesh's force to its own clumps, and the mesh's own a/alpha -- which would need an all-reduce -- are never used.
    if int(counts.get("nTri", 0)):
        flags = np.asarray(arrays["familyFlags"])
        mesh_owners = np.unique(np.asarray(arrays["ownerMesh"]))
        free = [int(o) for o in mesh_owners if not (flags[arrays["familyID"][o]] & (abi.FAMILY_FIXED | abi.FAMILY_PRESCRIBED))]
        if free:
            raise ValueError(f"mesh owner(s) {free} move under contact forces: only fixed or prescribed meshes can be replicated "
                             "across slabs (a free mesh would need an all-reduce of its accelerations every step)")
    x = np.asarray(clump_x, np.float64)[:n_clumps]
    edges = slab_edges(x, n_ranks)
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
                    "new_id": new_id, "edges": (edges[r], edges[r + 1]), "sphere_global": sph_idx.astype(np.int64),
                    "owner_global": owners_g})
    # send lists: what my neighbour holds as ghosts, in the neighbour's slot order (ascending global id on both sides)
    for r in range(n_ranks):
        me = out[r]
        me["recv_left"] = np.arange(me["n_own"], me["n_own"] + len(me["ghost_left_g"]), dtype=np.uint32)
        me["recv_right"] = np.arange(me["n_own"] + len(me["ghost_left_g"]),
                                     me["n_own"] + len(me["ghost_left_g"]) + len(me["ghost_right_g"]), dtype=np.uint32)
        me["send_left"] = me["new_id"][out[r - 1]["ghost_right_g"]].astype(np.uint32) if r > 0 else np.zeros(0, np.uint32)
        me["send_right"] = me["new_id"][out[r + 1]["ghost_left_g"]].astype(np.uint32) if r < n_ranks - 1 \
            else np.zeros(0, np.uint32)
        assert (me["send_left"] < me["n_own"]).all() and (me["send_right"] < me["n_own"]).all()
    for me in out:
        me["scene"] = abi.make_scene_struct(me["arrays"], me["counts"])
        del me["new_id"]
    return out


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


def redecompose(global_arrays, counts, payloads, n_ranks, halo, decode_x, flip_sign_wildcards=(0, 1, 2)):
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
    parts = decompose(g, counts, decode_x(g), n_ranks, halo)
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
