"""Worker for tests/test_decomp.py::test_overlapped_exchange_through_torch_streams / _through_rccl_on_one_gpu (GPU).  One
process, two contexts (two slabs) on one GPU; the ghost records travel between them on torch ExternalStreams wrapping the
contexts' halo streams -- the stream / event choreography bench.py uses -- either as device copies or
(DEME_OVERLAP_TEST_RCCL=1) through RCCL itself: a world-size-1 nccl group delivers a send to self to the receive posted in the
same batch.  torch is imported first: importing it after libdeme_hip.so would bring a second ROCm runtime into the process."""
import os
import sys

import torch  # noqa: F401  (first)
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

GKEYS = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY", "omgBarZ")


def main():
    pkg = g.load_package()
    n_clumps = int(os.environ.get("DEME_OVERLAP_TEST_CLUMPS", "3000"))  # (a 400k-clump run of this worker is how the deferral
    # logic was checked at scale: 1e8 owner-steps, bit-identical)
    b = pkg.model.packed_bed(n_clumps, seed=6, cd_freq=7, spacing_mult=2.5, init_vz=-0.4, aspect=(2.0, 1.0, 0.5))
    p, sc = b.Initialize()
    x = np.concatenate([bb.xyz for bb in b.batches])[:, 0]
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 2, halo=0.035)
    dev = torch.device("cuda", 0)
    gb = pkg.abi.GHOST_BYTES

    def make(pt):
        ctx = pkg.Context(0)
        ctx.set_params(p), ctx.upload_scene(pt["scene"])
        return ctx

    def ids_of(pt):
        return {k: torch.from_numpy(pt[k].astype(np.int32)).to(dev) for k in ("send_left", "send_right", "recv_left", "recv_right")}

    ids = [ids_of(pt) for pt in parts]
    n01, n10 = len(parts[0]["send_right"]), len(parts[1]["send_left"])
    steps = int(os.environ.get("DEME_OVERLAP_TEST_STEPS", "60"))
    # ---- ordered exchange (reference)
    plain = [make(pt) for pt in parts]
    s01 = torch.empty(max(1, n01) * gb, dtype=torch.uint8, device=dev)
    s10 = torch.empty(max(1, n10) * gb, dtype=torch.uint8, device=dev)
    for _ in range(steps):
        plain[0].halo_pack(ids[0]["send_right"].data_ptr(), n01, s01.data_ptr())
        plain[1].halo_pack(ids[1]["send_left"].data_ptr(), n10, s10.data_ptr())
        plain[0].sync(), plain[1].sync()
        plain[1].halo_unpack(ids[1]["recv_left"].data_ptr(), n01, s01.data_ptr())
        plain[0].halo_unpack(ids[0]["recv_right"].data_ptr(), n10, s10.data_ptr())
        plain[0].step(1), plain[1].step(1)
    # ---- overlapped exchange: send and receive buffers per rank, torch copies on the halo streams stand in for RCCL
    over = [make(pt) for pt in parts]
    ext = [torch.cuda.ExternalStream(c.halo_stream()) for c in over]
    send = [torch.empty(max(1, n01) * gb, dtype=torch.uint8, device=dev), torch.empty(max(1, n10) * gb, dtype=torch.uint8, device=dev)]
    recv = [torch.empty(max(1, n10) * gb, dtype=torch.uint8, device=dev), torch.empty(max(1, n01) * gb, dtype=torch.uint8, device=dev)]
    packed = [torch.cuda.Event(), torch.cuda.Event()]
    delivered = torch.cuda.Event()
    use_rccl = os.environ.get("DEME_OVERLAP_TEST_RCCL") == "1"
    if use_rccl:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    n_split = 0
    for _ in range(steps):
        n_split += sum(int(not c.step_overlap_begin()) for c in over)
        over[0].halo_pack_async(ids[0]["send_right"].data_ptr(), n01, send[0].data_ptr())
        over[1].halo_pack_async(ids[1]["send_left"].data_ptr(), n10, send[1].data_ptr())
        for r in (0, 1):
            packed[r].record(ext[r])  # "send posted": the peer's receive may proceed once my pack has run
        if use_rccl:
            # both slabs' ghost records through RCCL itself: a world-size-1 communicator delivers a send to self to the receive
            # posted in the same group, in order -- send[0] -> recv[1], send[1] -> recv[0].  One batch on slab 0's halo stream.
            with torch.cuda.stream(ext[0]):
                ext[0].wait_event(packed[1])
                ops = [dist.P2POp(dist.isend, send[0], 0), dist.P2POp(dist.irecv, recv[1], 0),
                       dist.P2POp(dist.isend, send[1], 0), dist.P2POp(dist.irecv, recv[0], 0)]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                delivered.record(ext[0])
            ext[1].wait_event(delivered)
        else:
            for r in (0, 1):
                with torch.cuda.stream(ext[r]):
                    ext[r].wait_event(packed[1 - r])
                    recv[r].copy_(send[1 - r], non_blocking=True)  # what irecv delivers
        over[0].halo_unpack_async(ids[0]["recv_right"].data_ptr(), n10, recv[0].data_ptr())
        over[1].halo_unpack_async(ids[1]["recv_left"].data_ptr(), n01, recv[1].data_ptr())
        over[0].step_overlap_end(), over[1].step_overlap_end()
        # a send buffer may only be re-packed after the peer has copied it: in the real exchange RCCL's send completes on
        # the sender's halo stream; here the peer's copy is on the peer's stream, so wait for it explicitly
        torch.cuda.synchronize()
    assert n_split > steps, n_split
    for a, c in zip(plain, over):
        sa, sb = a.download_state(), c.download_state()
        for k in GKEYS:
            assert np.array_equal(sa[k], sb[k]), k
        assert int(a.counts().nContacts) == int(c.counts().nContacts) > 100
    print("OVERLAP_TORCH_OK split steps", n_split, "through RCCL" if use_rccl else "through device copies")
    if use_rccl:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
