"""Spatial slab decomposition for multi-GPU runs (no reference equivalent: the
reference only splits kT/dT across two devices, DEM/APIPublic.cpp:22-72).

The global scene (arrays of SceneBuilder.Initialize()) is cut into N slabs along x.  A rank's scene is
[its own clumps | ghosts from the left neighbour | ghosts from the right neighbour | analytical owners].
Ghosts are copies of neighbour-owned clumps whose centre lies within `halo` of the shared face; they carry
family GHOST_FAMILY (flag DEME_FAMILY_GHOST: never integrated locally) and are refreshed every step from
their owner rank (56-byte ghost records: pose, velocities, family; deme_halo_pack / deme_halo_unpack).
A local-ghost contact is evaluated on both ranks, each applying the force to its own clump only;
ghost-ghost contacts are masked out.

Round-1 limitation (stated in DESIGN.md): ownership and ghost lists are fixed at set-up, so a clump must not
drift further than `halo` minus its reach from its initial slab -- true for settling beds, not for flows.
"""
import numpy as np

from . import abi

GHOST_FAMILY = 254
FAMILY_GHOST_FLAG = 2

_OWNER_KEYS = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ",
               "omgBarX", "omgBarY", "omgBarZ", "familyID", "inertiaPropOffsets")
_SPHERE_KEYS = ("ownerClumpBody", "clumpComponentOffset", "sphereMaterialOffset")


def slab_edges(x, n_ranks):
    """Equal-count slab boundaries along x (quantiles), so every rank owns ~the same number of clumps."""
    qs = np.quantile(np.asarray(x, np.float64), np.linspace(0, 1, n_ranks + 1))
    qs[0], qs[-1] = -np.inf, np.inf
    return qs


def decompose(arrays, counts, clump_x, n_ranks, halo):
    """Split a global scene.  clump_x: x of every clump centre (world frame).  Returns one dict per rank:
    arrays, counts, n_own, global_ids (own clumps' global owner ids), send/recv id lists (local owner ids)."""
    n_clumps = int(counts["nOwnerClumps"])
    n_owners = int(counts["nOwners"])
    assert int(counts.get("nTri", 0)) == 0, "meshes are not decomposed in this round"
    x = np.asarray(clump_x, np.float64)[:n_clumps]
    edges = slab_edges(x, n_ranks)
    rank_of = np.clip(np.searchsorted(edges, x, side="right") - 1, 0, n_ranks - 1)
    sph_owner_g = arrays["ownerClumpBody"]
    first_sphere = np.searchsorted(sph_owner_g, np.arange(n_owners + 1))  # spheres are clump-major
    extra_owners = np.arange(n_clumps, n_owners)  # analytical owners, kept on every rank
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
        fam = a["familyID"]
        fam[len(own):len(own) + len(gl) + len(gr)] = GHOST_FAMILY
        clumps_here = owners_g[:len(own) + len(gl) + len(gr)]
        sph_idx = np.concatenate([np.arange(first_sphere[o], first_sphere[o + 1]) for o in clumps_here]) \
            if len(clumps_here) else np.zeros(0, np.int64)
        for k in _SPHERE_KEYS:
            a[k] = arrays[k][sph_idx].copy()
        a["ownerClumpBody"] = new_id[arrays["ownerClumpBody"][sph_idx]].astype(np.uint32)
        a["objOwner"] = new_id[arrays["objOwner"]].astype(np.uint32)
        flags = arrays["familyFlags"].copy()
        flags[GHOST_FAMILY] |= FAMILY_GHOST_FLAG
        masks = arrays["familyMasks"].copy()
        masks[(1 + GHOST_FAMILY) * GHOST_FAMILY // 2 + GHOST_FAMILY] = 1  # ghost-ghost pairs are someone else's job
        a["familyFlags"], a["familyMasks"] = flags, masks
        c = dict(counts)
        c.update({"nOwners": len(owners_g), "nOwnerClumps": len(clumps_here), "nSpheres": len(sph_idx)})
        out.append({"arrays": a, "counts": c, "n_own": len(own), "global_ids": own, "ghost_left_g": gl, "ghost_right_g": gr,
                    "new_id": new_id, "edges": (edges[r], edges[r + 1])})
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
                    "omgBarX", "omgBarY", "omgBarZ")


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
