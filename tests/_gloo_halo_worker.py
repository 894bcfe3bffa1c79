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
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
