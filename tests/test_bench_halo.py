"""bench.py's per-step ghost exchange (Halo.step) with fakes for the context, torch and torch.distributed: the call
sequence of the overlapped path and the P2P op list for an end rank and a middle rank.  (The real thing needs one GPU per
rank; the stream / event choreography itself is tested on one GPU in tests/test_decomp.py.)"""
import contextlib
import types

import numpy as np


class FakeTensor:
    def __init__(self, n, tag):
        self.n, self.tag = n, tag

    def data_ptr(self):
        return hash(self.tag) & 0xFFFFFFF


class FakeTorch:
    uint8 = "u8"

    class cuda:
        @staticmethod
        def ExternalStream(ptr):
            return ("ext", ptr)

        @staticmethod
        def current_device():
            return 0

        @staticmethod
        @contextlib.contextmanager
        def stream(s):
            FakeTorch.log.append(("enter_stream", s))
            yield
            FakeTorch.log.append(("exit_stream", s))

    log = []

    @staticmethod
    def device(kind, idx):
        return (kind, idx)

    @staticmethod
    def from_numpy(a):
        t = FakeTensor(len(a), ("ids", len(a)))
        t.to = lambda dev: t
        return t

    @staticmethod
    def empty(n, dtype=None, device=None):
        FakeTorch.counter = getattr(FakeTorch, "counter", 0) + 1
        return FakeTensor(n, ("buf", FakeTorch.counter))


class FakeDist:
    isend, irecv = "isend", "irecv"

    def __init__(self, log):
        self.log = log

    def P2POp(self, fn, tensor, peer):
        return (fn, tensor, peer)

    def batch_isend_irecv(self, ops):
        self.log.append(("batch", [(fn, peer) for fn, _, peer in ops]))
        return [types.SimpleNamespace(wait=lambda: self.log.append(("wait",))) for _ in ops]


class FakeCtx:
    def __init__(self, log):
        self.log = log

    def halo_stream(self):
        return 1234

    def __getattr__(self, name):
        return lambda *a: self.log.append((name,) + tuple(a[1:2]))  # keep the count argument


def _halo(rank, world):
    import bench
    FakeTorch.log = log = []
    part = {"send_left": np.arange(3), "send_right": np.arange(5), "recv_left": np.arange(4), "recv_right": np.arange(6)}
    pkg = types.SimpleNamespace(abi=types.SimpleNamespace(GHOST_BYTES=56))
    h = bench.Halo(pkg, FakeCtx(log), part, rank, world, FakeTorch, FakeDist(log))
    return h, log


def test_overlapped_step_sequence_middle_rank():
    h, log = _halo(1, 3)
    assert h.overlap and h.bytes_per_step == 56 * (3 + 5)
    for _ in range(2):  # the plan is built once and reused
        del log[:]
        h.step()
        names = [e[0] for e in log]
        assert names == ["step_overlap_begin", "halo_pack_async", "halo_pack_async", "enter_stream", "batch", "wait", "wait", "wait",
                         "wait", "exit_stream", "halo_unpack_async", "halo_unpack_async", "step_overlap_end"]
        assert [e[1] for e in log if e[0] == "halo_pack_async"] == [3, 5]      # send_left, send_right counts
        assert [e[1] for e in log if e[0] == "halo_unpack_async"] == [4, 6]    # recv_left, recv_right counts
        assert [e for e in log if e[0] == "batch"][0][1] == [("isend", 0), ("irecv", 0), ("isend", 2), ("irecv", 2)]


def test_overlapped_step_sequence_end_ranks():
    h, log = _halo(0, 2)
    h.step()
    assert [e for e in log if e[0] == "batch"][0][1] == [("isend", 1), ("irecv", 1)]
    assert [e[1] for e in log if e[0] == "halo_pack_async"] == [5] and [e[1] for e in log if e[0] == "halo_unpack_async"] == [6]
    h, log = _halo(1, 2)
    h.step()
    assert [e for e in log if e[0] == "batch"][0][1] == [("isend", 0), ("irecv", 0)]
    assert [e[1] for e in log if e[0] == "halo_pack_async"] == [3] and [e[1] for e in log if e[0] == "halo_unpack_async"] == [4]


def test_ordered_exchange_sequence():
    import bench
    FakeTorch.log = log = []
    part = {"send_left": np.arange(3), "send_right": np.arange(5), "recv_left": np.arange(4), "recv_right": np.arange(6)}
    pkg = types.SimpleNamespace(abi=types.SimpleNamespace(GHOST_BYTES=56))
    h = bench.Halo(pkg, FakeCtx(log), part, 0, 2, FakeTorch, FakeDist(log), overlap=False)
    h.step()
    assert [e[0] for e in log] == ["halo_pack", "batch", "wait", "wait", "halo_unpack", "step"]
