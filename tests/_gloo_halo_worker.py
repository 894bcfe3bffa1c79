"""Worker for tests/test_decomp.py::test_gloo_world2_halo_exchange.  Run without arguments it spawns two
processes (gloo, 127.0.0.1); each steps one slab with the CPU oracle and exchanges ghost records with its
neighbour through torch.distributed send/recv; rank 0 compares the union with a single-domain oracle run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GKEYS = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY",
         "omgBarZ")
STEPS = 80
STEPS_A, STEPS_B = 400, 20  # re-decomposition run: before / after the clumps change hands


def pack(st, ids):
    return np.concatenate([st[k][ids].astype(np.float64) if k != "voxelID" else st[k][ids].view(np.float64) for k in GKEYS])


def unpack(st, ids, flat):
    n = len(ids)
    for i, k in enumerate(GKEYS):
        col = flat[i * n:(i + 1) * n]
        st[k][ids] = col.view(np.uint64) if k == "voxelID" else col.astype(st[k].dtype)


def worker(rank, world, port):
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = g.load_package()
    orc = g.load_oracle()
    orc.set_num_threads(2)
    b = pkg.model.packed_bed(1200, seed=12, cd_freq=0, spacing_mult=2.5, init_vz=-0.4, aspect=(2.0, 1.0, 0.5))
    p, sc = b.Initialize()
    x = np.concatenate([bb.xyz for bb in b.batches])[:, 0]
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, world, halo=0.035)
    me = parts[rank]
    sim = orc.make_sim(pkg, p, me["scene"])
    nb = 1 - rank
    send_ids, recv_ids = (me["send_right"], me["recv_right"]) if rank == 0 else (me["send_left"], me["recv_left"])
    for _ in range(STEPS):
        st = sim.download_state()
        out = torch.from_numpy(pack(st, send_ids))
        inc = torch.empty(len(recv_ids) * len(GKEYS), dtype=torch.float64)
        if rank == 0:
            dist.send(out, nb), dist.recv(inc, nb)
        else:
            dist.recv(inc, nb), dist.send(out, nb)
        unpack(st, recv_ids, inc.numpy())
        sim.upload_state({k: st[k] for k in GKEYS})
        sim.step(1)
    st = sim.download_state()
    n = me["n_own"]
    X = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:n]
    gathered = [None, None]
    dist.all_gather_object(gathered, (me["global_ids"], X))
    if rank == 0:
        one = orc.make_sim(pkg, p, sc)
        one.step(STEPS)
        s1 = one.download_state()
        nC = sc.nOwnerClumps
        X1 = pkg.model.decode_positions(s1["voxelID"], s1["locX"], s1["locY"], s1["locZ"], p.nvXp2, p.nvYp2, p.voxelSize,
                                        p.l)[:nC]
        Xd = np.zeros_like(X1)
        for ids, xs in gathered:
            Xd[ids] = xs
        err = float(np.abs(Xd - X1).max())
        assert one.counts().nContacts > 100, one.counts().nContacts
        assert err < 2e-7, err
        print(f"GLOO_HALO_OK max|dx|={err:.3e} contacts={one.counts().nContacts}")
    dist.barrier()
    redecomposition_run(rank, world, dist, torch, pkg, orc)
    dist.barrier()
    dist.destroy_process_group()


def exchange_and_step(sim, me, rank, dist, torch, steps):
    nb = 1 - rank
    send_ids, recv_ids = (me["send_right"], me["recv_right"]) if rank == 0 else (me["send_left"], me["recv_left"])
    for _ in range(steps):
        st = sim.download_state()
        out = torch.from_numpy(pack(st, send_ids))
        inc = torch.empty(len(recv_ids) * len(GKEYS), dtype=torch.float64)
        if rank == 0:
            dist.send(out, nb), dist.recv(inc, nb)
        else:
            dist.recv(inc, nb), dist.send(out, nb)
        unpack(st, recv_ids, inc.numpy())
        sim.upload_state({k: st[k] for k in GKEYS})
        sim.step(1)


def redecomposition_run(rank, world, dist, torch, pkg, orc):
    """sheared bed: clumps cross the cut; ownership and contact history migrate through
    decomp.redecompose_distributed (all_gather_object) + seed_contacts"""
    b = pkg.model.packed_bed(1200, seed=9, cd_freq=0, spacing_mult=2.5, init_vz=-0.4, aspect=(2.0, 1.0, 0.5))
    p, sc = b.Initialize()
    nC = int(sc.nOwnerClumps)
    b.arrays["vX"][:nC] = np.where(np.arange(nC) % 2 == 0, 0.6, 0.3).astype(np.float32)
    sc = pkg.abi.make_scene_struct(b.arrays, b.counts)
    x = np.concatenate([bb.xyz for bb in b.batches])[:, 0]
    me = pkg.decomp.decompose(b.arrays, b.counts, x, world, halo=0.035)[rank]
    sim = orc.make_sim(pkg, p, me["scene"])
    exchange_and_step(sim, me, rank, dist, torch, STEPS_A)
    cnt = sim.contacts()
    W = np.stack([sim.wildcard(w) for w in range(int(p.nContactWildcards))], 1)

    def decode_x(arrays):
        return pkg.model.decode_positions(arrays["voxelID"], arrays["locX"], arrays["locY"], arrays["locZ"], p.nvXp2, p.nvYp2,
                                          p.voxelSize, p.l)[:, 0] + p.LBFX

    _, me2, seed = pkg.decomp.redecompose_distributed(dist, rank, world, b.arrays, b.counts, me, sim.download_state(), cnt, W,
                                                      0.035, decode_x)
    moved = len(np.setdiff1d(me2["global_ids"], me["global_ids"]))
    sim2 = orc.make_sim(pkg, p, me2["scene"])
    sim2.seed_contacts(*seed)
    exchange_and_step(sim2, me2, rank, dist, torch, STEPS_B)
    exchange_and_step(sim, me, rank, dist, torch, STEPS_B)  # the old decomposition simply continuing: the yardstick

    def owned_x(s, part):
        st = s.download_state()
        return pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize,
                                          p.l)[:part["n_own"]]

    gathered = [None, None]
    dist.all_gather_object(gathered, (me2["global_ids"], owned_x(sim2, me2), me["global_ids"], owned_x(sim, me), moved))
    if rank == 0:
        Xn, Xo = np.zeros((nC, 3)), np.zeros((nC, 3))
        for ids2, x2, ids1, x1, _ in gathered:
            Xn[ids2], Xo[ids1] = x2, x1
        err = float(np.abs(Xn - Xo).max())
        total_moved = sum(g[4] for g in gathered)
        assert total_moved > 3, total_moved
        assert err < 1e-9, err  # 5e-6 if the history is not carried (tests/test_decomp.py)
        print(f"GLOO_REDECOMP_OK max|dx|={err:.3e} moved={total_moved}")
    # ---- the same migration with face-neighbour traffic only (decomp.migrate_neighbours over point-to-point object messages):
    # fixed edges, so the ownership it produces is compared with the edges, and the run with the same yardstick
    def state_x(st):
        return pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize,
                                          p.l)[:, 0] + p.LBFX
    sim_c = orc.make_sim(pkg, p, me["scene"])
    exchange_and_step(sim_c, me, rank, dist, torch, STEPS_A)
    cnt_c = sim_c.contacts()
    W_c = np.stack([sim_c.wildcard(w) for w in range(int(p.nContactWildcards))], 1)
    me3, seed3 = pkg.decomp.migrate_neighbours(rank, world, me, sim_c.download_state(), cnt_c, W_c, me["all_edges"], 0.035, state_x,
                                               pkg.decomp.torch_transport(dist, rank, world))
    sim3 = orc.make_sim(pkg, p, me3["scene"])
    sim3.seed_contacts(*seed3)
    exchange_and_step(sim3, me3, rank, dist, torch, STEPS_B)
    exchange_and_step(sim_c, me, rank, dist, torch, STEPS_B)
    moved3 = len(np.setdiff1d(me3["global_ids"], me["global_ids"]))
    gathered = [None, None]
    dist.all_gather_object(gathered, (me3["global_ids"], owned_x(sim3, me3), me["global_ids"], owned_x(sim_c, me), moved3))
    if rank == 0:
        Xn, Xo = np.zeros((nC, 3)), np.zeros((nC, 3))
        for ids2, x2, ids1, x1, _ in gathered:
            Xn[ids2], Xo[ids1] = x2, x1
        assert sorted(np.concatenate([g[0] for g in gathered]).tolist()) == list(range(nC))
        err = float(np.abs(Xn - Xo).max())
        total_moved = sum(g[4] for g in gathered)
        assert total_moved > 3 and err < 1e-9, (total_moved, err)
        print(f"GLOO_NEIGHBOUR_MIGRATION_OK max|dx|={err:.3e} moved={total_moved}")


if __name__ == "__main__":
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
