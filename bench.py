#!/usr/bin/env python3
"""bench.py -- clump*steps/s of the MI355X-native DEM hot path on BASELINE.json configs[1]
(1M three-sphere clumps in a box, gravity settling), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the whole bed: (contact detection every --cd-freq steps:
margins, binning, bin-sorted sweep, history map) + contact forces + fused accumulation/integration.
State is resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

N > 1 (weak scaling): ONE bed N times as long in x is cut into N slabs (dem-engine_amd/decomp.py); each rank
steps its slab and, every step, exchanges ghost-clump records (56 B each: pose, velocities) with its face
neighbours over RCCL (torch.distributed P2P, backend nccl).  See DESIGN.md section 6.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# the hosts of this pool only support dmabuf IPC: without this RCCL's cross-process buffers fail with
# "hipIpcGetMemHandle: invalid argument".  Normally exported already; set before torch / the HIP runtime load.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)
GUIDE_COPY_GBS = 6290.0  # ... its measured float4 copy
README_CLUMP_STEPS_PER_S = 1e6 * 1e6 / 3600.0  # reference README.md:48, two RTX 3080 ("around 1 hour")
HALO = 0.03  # ghost layer thickness [m]: two lattice spacings (clump reach 7.3 mm)


def build_bed(pkg, n_clumps, seed, cd_freq, x_mult=1, order="lattice", bin_multiple=5.0, slab=None):
    """slab = (rank, n_ranks, halo): only that x-slab of the bed (own clumps + ghosts) is flattened into a scene"""
    b = pkg.model.packed_bed(n_clumps * x_mult, seed=seed, cd_freq=cd_freq, aspect=(1.0 * x_mult, 1.0, 0.05),
                             spacing_mult=3.0, jitter=0.05, bin_multiple=bin_multiple, init_vz=-1.0, order=order, slab=slab)
    b.SetExpandSafetyMultiplier(1.2)
    b.SetExpandSafetyAdder(0.02)
    return b


# BASELINE configs[4] flavour (--config5): polydisperse single spheres, radii from 8 templates in [r, 3r], a user force
# fragment (frictionless Hertz + pairwise cohesion + a contact-age wildcard) compiled at run time through hipRTC; written
# against the reference's ingredient names like DEMUserScripts/ForceModelWithCohesion.cu
COHESIVE_FRAGMENT = r"""
if (overlapDepth > 0) {
    float E_cnt;
    matProxy2ContactParam<float>(E_cnt, E[bodyAMatType], nu[bodyAMatType], E[bodyBMatType], nu[bodyBMatType]);
    const float CoR_cnt = CoR[bodyAMatType][bodyBMatType];
    float3 rotVelCPA = cross(ARotVel, locCPA), rotVelCPB = cross(BRotVel, locCPB);
    applyOriQToVector3<float, deme::oriQ_t>(rotVelCPA.x, rotVelCPA.y, rotVelCPA.z, AOriQ.w, AOriQ.x, AOriQ.y, AOriQ.z);
    applyOriQToVector3<float, deme::oriQ_t>(rotVelCPB.x, rotVelCPB.y, rotVelCPB.z, BOriQ.w, BOriQ.x, BOriQ.y, BOriQ.z);
    const float3 velB2A = (ALinVel + rotVelCPA) - (BLinVel + rotVelCPB);
    const float projection = dot(velB2A, B2A);
    const float mass_eff = (AOwnerMass * BOwnerMass) / (AOwnerMass + BOwnerMass);
    const float sqrt_Rd = sqrt(overlapDepth * (ARadius * BRadius) / (ARadius + BRadius));
    const float Sn = 2. * E_cnt * sqrt_Rd;
    const float loge = (CoR_cnt < DEME_TINY_FLOAT) ? log(DEME_TINY_FLOAT) : log(CoR_cnt);
    const float beta = loge / sqrt(loge * loge + deme::PI_SQUARED);
    const float k_n = deme::TWO_OVER_THREE * Sn;
    const float gamma_n = deme::TWO_TIMES_SQRT_FIVE_OVER_SIX * beta * sqrt(Sn * mass_eff);
    force += (k_n * overlapDepth + gamma_n * projection) * B2A;
    force += -Cohesion[bodyAMatType][bodyBMatType] * B2A;
    contact_age += ts;
}
"""


def build_config5(pkg, n, seed, cd_freq):
    r = 0.003
    radii = [r * (1.0 + 2.0 * i / 7.0) for i in range(8)]
    b = pkg.model.packed_bed(n, seed=seed, cd_freq=cd_freq, scale=3.0 * r, aspect=(1.0, 1.0, 0.05), spacing_mult=2.05, jitter=0.02,
                             bin_multiple=4.0, init_vz=-1.0, three_sphere=False, radii_poly=radii)
    b.materials[0]["Cohesion"] = 0.002
    b.SetMustPairwiseMatProp(["Cohesion"])
    b.DefineContactForceModel(COHESIVE_FRAGMENT)
    b.SetPerContactWildcards(["contact_age"])
    b.SetExpandSafetyMultiplier(1.2)
    b.SetExpandSafetyAdder(0.02)
    return b


def force_kernel_bytes(n_owners, n_spheres, n_contacts, n_w):
    """Algorithmic HBM bytes of ONE contact-force launch (SURVEY 8d / DESIGN.md 3.2):
    N_c*(9 + 8*n_w) + N_o*57 + N_s*7  (contact ids+type, wildcards read+write, owner and sphere state
    once each).  The per-contact contribution records this design writes instead of atomics are
    implementation traffic and are NOT counted."""
    return n_contacts * (9 + 8 * n_w) + n_owners * 57 + n_spheres * 7


def _time_oracle(sim, orc, threads, budget_s, chunk, max_steps):
    """steps/s of the oracle at a given OpenMP team size, bounded by wall time and by a step count"""
    orc.set_num_threads(threads)
    sim.step(2)  # thread team spin-up
    t0 = time.perf_counter()
    steps = 0
    while steps < max_steps and time.perf_counter() - t0 < budget_s:
        sim.step(chunk)
        steps += chunk
    return steps, time.perf_counter() - t0


def cpu_baseline(pkg, seed, cd_freq, budget_s=24.0, main=None):
    """The CPU oracle (oracle/, a port: the reference has no CPU path) on the host cores of this box, in the shape SURVEY 8d
    asks for -- a down-scaled configs[1] (1e5 clumps of the same recipe, pre-settled on the GPU so the bed is packed like the
    measured one) single-thread and all-core, plus configs[0] (BallDrop-like, ~1e4 single spheres) -- each leg BOUNDED in wall
    time (the contract wants the default bench to finish within minutes; 1e5 clumps x 1e3 steps single-thread would take
    several minutes by itself), so the step counts actually run are part of the sample description."""
    orc = entry.load_oracle()
    # the timed build: -O3, list building / sorts / history map / accumulation spread over the OpenMP team (oracle/Makefile
    # libdeme_oracle_perf.so, ORC_PERF); rebuilt with -march=native on this host when a compiler is here.  The parity build the
    # tests use stays untouched.
    native = orc.build_perf_native()
    orc.set_variant(True)
    ncpu = os.cpu_count() or 1
    n = 100_000
    b = build_bed(pkg, n, seed, cd_freq=cd_freq)
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_params(p)
    ctx.upload_scene(sc)
    ctx.step(20000)
    st = ctx.download_state()
    ctx.close()
    sim = orc.make_sim(pkg, p, sc)
    sim.upload_state({k: st[k] for k in st if k not in ("aX", "aY", "aZ", "alphaX", "alphaY", "alphaZ")})
    orc.set_num_threads(min(32, ncpu))
    sim.step(cd_freq + 1)  # first detection + page-in
    s1, t1 = _time_oracle(sim, orc, 1, budget_s * 0.3, cd_freq or 5, 1000)  # whole K-cycles: the detection's share is in
    legs = []
    for th in sorted({min(t, ncpu) for t in (8, 16, 32, 64, ncpu)}):
        sN, tN = _time_oracle(sim, orc, th, budget_s * 0.08, cd_freq or 20, 1000)
        legs.append((n * sN / tN, th, sN))
    best = max(legs)
    sA, tA = _time_oracle(sim, orc, best[1], budget_s * 0.3, cd_freq or 40, 1000)  # SURVEY 8d's shape: 1e5 clumps x 1e3 steps, if the budget allows
    nc = int(sim.counts().nContacts)
    out = {"value": n * sA / tA, "unit": "clump*steps/s", "cores": int(best[1]), "kind": "port",
           "sample": f"{n} three-sphere clumps (configs[1] recipe down-scaled, packed state, {nc} contacts, cd every {cd_freq}) x {sA} "
                     f"steps on {best[1]} OpenMP threads (fastest of {[l[1] for l in legs]}); oracle/deme_oracle.cpp built -O3 "
                     f"{'-march=native on this host' if native else '-march=x86-64-v3'} with ORC_PERF: incidence list, sorts, history map and per-owner "
                     f"accumulation run on the whole team (the parity build of the tests is the -O2 -ffp-contract=off one)",
           "host_cores": ncpu,
           "single_thread": {"value": n * s1 / t1, "cores": 1, "steps": s1},
           "all_core_small_bed": {"value": [l[0] for l in legs if l[1] == max(x[1] for x in legs)][0], "cores": max(x[1] for x in legs),
                                  "note": "1e5 clumps do not feed a team this wide: this leg times OpenMP barriers; see all_core for the measured bed"},
           "by_threads": {str(l[1]): l[0] for l in legs}}
    try:  # configs[0]: the BallDrop-like scene of tests/test_config0_balldrop.py (plumbing case), a few seconds
        b0 = pkg.model.balldrop_like(seed=12345) if hasattr(pkg.model, "balldrop_like") else None
        if b0 is not None:
            p0, sc0 = b0.Initialize()
            sim0 = orc.make_sim(pkg, p0, sc0)
            orc.set_num_threads(min(16, ncpu))
            sim0.step(5)
            s0, t0 = _time_oracle(sim0, orc, min(16, ncpu), budget_s * 0.08, 10, 1000)
            out["config0"] = {"value": int(sc0.nOwnerClumps) * s0 / t0, "cores": min(16, ncpu), "steps": s0,
                              "clumps": int(sc0.nOwnerClumps), "triangles": int(sc0.nTri)}
    except Exception as e:  # noqa: BLE001 -- the side leg must never cost the bench line
        out["config0"] = {"error": f"{type(e).__name__}: {e}"}
    # the all-core figure on the MEASURED bed (the 1e5-clump bed above does not feed a 256-thread team: its "all-core" leg timed
    # OpenMP barriers).  One K-cycle is too long to run here, so it is composed: one step with its detection + a few plain steps,
    # value = clumps * K / (t_detection_step + (K - 1) * t_step).  Bounded: the leg is skipped if the first step alone eats the budget.
    if main is not None and ncpu > best[1]:
        try:
            p1, sc1, st1 = main
            n1 = int(sc1.nOwnerClumps)
            sim1 = orc.make_sim(pkg, p1, sc1)
            sim1.upload_state({k: st1[k] for k in st1 if k not in ("aX", "aY", "aZ", "alphaX", "alphaY", "alphaZ")})
            orc.set_num_threads(ncpu)
            t_ = time.perf_counter()
            sim1.step(1)  # list building from scratch + one step (page-in included: the dearer side for the CPU)
            t_first = time.perf_counter() - t_
            t_ = time.perf_counter()
            sim1.step(1)
            t_det = time.perf_counter() - t_ if not cd_freq else None  # (cd_freq 0: every step detects)
            m, t_plain = 0, 0.0
            while m < 6 and t_plain < 0.25 * budget_s and t_first < 0.5 * budget_s:
                t_ = time.perf_counter()
                sim1.step(1)
                t_plain += time.perf_counter() - t_
                m += 1
            if m:
                K = max(int(cd_freq), 1)
                t_step = t_plain / m
                t_cycle = t_first + (K - 1) * t_step if cd_freq else (t_det or t_step) * K
                out["all_core"] = {"value": n1 * K / t_cycle, "cores": ncpu, "clumps": n1,
                                   "sample": f"the measured bed ({n1} clumps): one step with its detection {t_first:.2f} s + {m} plain steps at {t_step:.3f} s each, "
                                             f"composed into a {K}-step cycle"}
            del sim1
        except Exception as e:  # noqa: BLE001
            out["all_core"] = {"error": f"{type(e).__name__}: {e}"}
    orc.set_variant(False)
    return out


def pmc_traffic(n_contacts, kernel):
    """roofline.traffic: HBM bytes per launch of the force kernel from the rocprofv3 PMC passes of this same
    command (FETCH_SIZE and WRITE_SIZE in separate passes, corrected with the factors calibrated in the same
    passes; profiles/<round>/traffic.json written by profiles/make_traffic.py).  Counters cannot be read from
    inside the process, so the committed summary of the newest round is quoted -- only when it was taken on the same
    kernel and workload."""
    rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", d, "traffic.json")))
    if not rounds:
        return {"traffic": None}
    path = os.path.join(ROOT, "profiles", rounds[-1], "traffic.json")
    t = json.load(open(path))
    if t.get("force_kernel", "k_calc_forces<0, 0>") != kernel:
        return {"traffic": None, "traffic_note": f"profiles/{rounds[-1]}/traffic.json was taken on {t.get('force_kernel')}"}
    if abs(t["contacts"] - n_contacts) > 0.05 * n_contacts:
        return {"traffic": None, "traffic_note": f"profiles/{rounds[-1]}/traffic.json was taken on a different workload"}
    lo = t["kernels"][kernel].get("traffic_lower_bound")
    return {"traffic": t["traffic_bytes_per_launch"], "traffic_lower_bound": lo,
            "traffic_source": f"profiles/{rounds[-1]}/traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; 2 x FETCH + WRITE as the guide prescribes -- "
                              "an upper bound here: the in-pass calibration shows gathers are not half-counted like streams; lower bound beside it)"}


def attainable_copy_gbs(torch, gib=1, reps=5):
    """the attainable HBM rate on this box beside the nominal peak -- a device-to-device copy of `gib` GiB (beyond the
    256 MiB Infinity Cache), read + written bytes over the best of `reps` timings."""
    src = torch.empty(gib << 28, dtype=torch.float32, device="cuda").fill_(1.0)
    dst = torch.empty_like(src)
    dst.copy_(src)
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    del src, dst
    return 2.0 * (gib << 30) / (best * 1e-3) / 1e9


def pmc_calibration(torch):
    """Known-byte kernels for calibrating FETCH_SIZE / WRITE_SIZE inside a rocprofv3 --pmc pass
    (MI355X_MICROARCH.md, HBM section): a 1 GiB streaming copy and 4 M random 64-byte row gathers out of
    a 512 MiB table (both beyond the 256 MiB Infinity Cache)."""
    g = torch.Generator(device="cuda").manual_seed(1)
    src = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_(generator=g)
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)  # 1 GiB read + 1 GiB written per launch
    table = src.view(-1, 16)[: 1 << 23]  # 8 Mi rows x 64 B
    idx = torch.randint(0, table.shape[0], (1 << 22,), device="cuda", generator=g)
    out = torch.empty((1 << 22, 16), dtype=torch.float32, device="cuda")
    for _ in range(3):
        torch.index_select(table, 0, idx, out=out)  # 256 MiB gathered in 64 B rows + 32 MiB of indices; 256 MiB written
    torch.cuda.synchronize()


class Halo:
    """Per-step ghost exchange with the face neighbours (RCCL P2P through torch.distributed)."""

    def __init__(self, pkg, ctx, part, rank, world, torch, dist, via_host=False, overlap=True):
        self.ctx, self.rank, self.world, self.torch, self.dist = ctx, rank, world, torch, dist
        # overlap: ghost traffic on the context's halo stream while the compute stream evaluates the owner runs that read
        # no ghost (deme_step_overlap_begin / _end); otherwise everything is ordered on the compute stream
        self.overlap = overlap and not via_host
        self.halo_stream = None
        if self.overlap:
            try:
                self.halo_stream = torch.cuda.ExternalStream(ctx.halo_stream())
            except Exception as e:  # an older torch without ExternalStream: fall back to the ordered exchange
                print(f"[bench] halo overlap disabled: {e}", file=sys.stderr)
                self.overlap = False
        self.via_host = via_host  # plumbing test on a box with fewer GPUs than ranks: gloo, staged through host memory
        dev = torch.device("cuda", torch.cuda.current_device())
        keys = ("send_left", "send_right", "recv_left", "recv_right")
        self.ids = {k: torch.from_numpy(part[k].astype(np.int32)).to(dev) for k in keys}
        gb = pkg.abi.GHOST_BYTES
        self.buf = {k: torch.empty(max(1, len(part[k])) * gb, dtype=torch.uint8, device=dev) for k in keys}
        self.n = {k: len(part[k]) for k in keys}
        self.bytes_per_step = gb * (self.n["send_left"] + self.n["send_right"])
        self._plan = None

    def neighbours(self):
        """[(side, rank)] of the face neighbours in the x-slab chain"""
        return [(s, nb) for s, nb in (("left", self.rank - 1), ("right", self.rank + 1)) if 0 <= nb < self.world]

    def probe_overlap(self):
        """One tiny exchange with the face neighbours on the halo stream before anything depends on it.  If torch / RCCL refuse
        the external stream the call raises on every rank alike; the ranks then agree (all-reduce MIN) to use the ordered
        exchange on the compute stream instead.  Returns whether the overlapped path stays on."""
        if not self.overlap:
            return False
        t, d = self.torch, self.dist
        ok = 1
        try:
            dev = t.device("cuda", t.cuda.current_device())
            ops, keep = [], []
            for _, nb in self.neighbours():
                a, b = t.full((16,), self.rank, dtype=t.uint8, device=dev), t.full((16,), 255, dtype=t.uint8, device=dev)
                keep.append((a, b, nb))
                ops += [d.P2POp(d.isend, a, nb), d.P2POp(d.irecv, b, nb)]
            with t.cuda.stream(self.halo_stream):
                for w in d.batch_isend_irecv(ops):
                    w.wait()
            self.halo_stream.synchronize()
            ok = int(all(int(b[0].item()) == nb for _, b, nb in keep))
        except Exception as e:  # noqa: BLE001 -- whatever the stack refuses, the answer is the ordered path
            print(f"[bench] rank {self.rank}: halo-stream exchange refused ({type(e).__name__}: {e}); ordered exchange instead",
                  file=sys.stderr, flush=True)
            ok = 0
        flag = t.tensor([ok], dtype=t.int32, device="cuda")
        d.all_reduce(flag, op=d.ReduceOp.MIN)
        self.overlap = bool(int(flag.item()))
        return self.overlap

    def step(self):
        """one time step including the ghost exchange"""
        if not self.overlap:
            self.exchange()
            self.ctx.step(1)
            return
        c = self.ctx
        if self._plan is None:  # pointers, counts and the P2P op list never change: built once (the loop body is host-bound work)
            sides = self.neighbours()
            pack = [(self.ids["send_" + s].data_ptr(), self.n["send_" + s], self.buf["send_" + s].data_ptr()) for s, _ in sides]
            unpack = [(self.ids["recv_" + s].data_ptr(), self.n["recv_" + s], self.buf["recv_" + s].data_ptr()) for s, _ in sides]
            ops = []
            for s, nb in sides:
                ops.append(self.dist.P2POp(self.dist.isend, self.buf["send_" + s], nb))
                ops.append(self.dist.P2POp(self.dist.irecv, self.buf["recv_" + s], nb))
            self._plan = (pack, unpack, ops)
        pack, unpack, ops = self._plan
        c.step_overlap_begin()
        for a in pack:
            c.halo_pack_async(*a)
        with self.torch.cuda.stream(self.halo_stream):  # RCCL orders the transfers against the halo stream
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for a in unpack:
            c.halo_unpack_async(*a)
        c.step_overlap_end()

    def exchange(self):
        c, d = self.ctx, self.dist
        ops = []
        sides = self.neighbours()
        for side, nb in sides:
            s, r = "send_" + side, "recv_" + side
            c.halo_pack(self.ids[s].data_ptr(), self.n[s], self.buf[s].data_ptr())
            if self.via_host:
                continue
            ops.append(d.P2POp(d.isend, self.buf[s], nb))
            ops.append(d.P2POp(d.irecv, self.buf[r], nb))
        if self.via_host:
            c.sync()
            host = {}
            for side, nb in sides:
                host["send_" + side] = self.buf["send_" + side].cpu()
                host["recv_" + side] = self.torch.empty(self.buf["recv_" + side].numel(), dtype=self.torch.uint8)
                ops.append(d.P2POp(d.isend, host["send_" + side], nb))
                ops.append(d.P2POp(d.irecv, host["recv_" + side], nb))
            for w in d.batch_isend_irecv(ops):
                w.wait()
            for side, nb in sides:
                self.buf["recv_" + side].copy_(host["recv_" + side])
            self.torch.cuda.current_stream().synchronize()
        elif ops:
            for w in d.batch_isend_irecv(ops):
                w.wait()
        for side, nb in sides:
            r = "recv_" + side
            c.halo_unpack(self.ids[r].data_ptr(), self.n[r], self.buf[r].data_ptr())


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: run the same command line as N ranks of ONE node under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and pass its exit code on.  A scaling run can then not
    silently measure one rank."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n_gpus} without a launcher: re-executing as {n_gpus} ranks under torch.distributed.run", file=sys.stderr, flush=True)
    return subprocess.call(cmd)


def globalise_part(part, b):
    """A rank that built only its own slab of the bed (model.packed_bed(slab=...)) numbers clumps and spheres by position in that
    scene; the library's migration needs ids that mean the same on every rank: the clump's index in the whole bed
    (b.slab_global_ids), 3 g + k for its spheres (the bench bed: three spheres per clump), replicated owners behind all clumps."""
    if getattr(b, "slab_global_ids", None) is None:
        return part
    q = dict(part)
    og = np.asarray(part["owner_global"], np.int64)
    n_cl = int(part["counts"]["nOwnerClumps"])
    gids = np.asarray(b.slab_global_ids, np.int64)
    glob = np.concatenate([gids[og[:n_cl]], int(b.slab_total) + np.arange(len(og) - n_cl)])
    sg = np.asarray(part["sphere_global"], np.int64)
    scene_owner = np.asarray(b.arrays["ownerClumpBody"], np.int64)[sg]
    first = np.searchsorted(np.asarray(b.arrays["ownerClumpBody"], np.int64), scene_owner)
    q["owner_global"] = glob
    q["sphere_global"] = 3 * gids[scene_owner] + (sg - first)
    return q


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--clumps", type=int, default=1_000_000, help="clumps per GPU (weak scaling)")
    ap.add_argument("--clumps-total", type=int, default=0,
                    help="STRONG scaling: this many clumps in the whole job, cut into --gpus slabs (BASELINE configs[2] as written: "
                         "--clumps-total 10000000 at 2 and 8 GPUs); overrides --clumps")
    ap.add_argument("--launch-check", action="store_true",
                    help="plumbing check of the launch path only: rendezvous, count the ranks, print a one-line JSON and exit "
                         "(no GPU work; what tests/test_bench_launch.py runs on a CPU box)")
    ap.add_argument("--cd-freq", type=int, default=40,
                    help="contact detection every K steps (0: every step); 40 = the setting of the demo the config-2 recipe "
                         "comes from (DEMdemo_Mixer.cpp:87); the reference's built-in default is 20 (API.h:1509)")
    ap.add_argument("--presettle", type=int, default=30000, help="untimed steps that let the lattice settle")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--order", default="morton", choices=["lattice", "morton", "random"],
                    help="numbering of the clumps: morton = along a Z-order curve (what SceneBuilder.ResortClumps gives a running "
                         "simulation; the default), lattice = the sampler's row-major order (+5 %% step time), random (+30 %%)")
    ap.add_argument("--mesh-triangles", type=int, default=0,
                    help="BASELINE configs[3] flavour: put a wavy, fixed plate of about this many triangles under the bed")
    ap.add_argument("--mesh-update-every", type=int, default=0,
                    help="with --mesh-triangles: the plate is a DEFORMABLE mesh whose node coordinates the script rewrites every this "
                         "many steps (a travelling ripple, through deme_update_tri_nodes = DEMTracker::UpdateMesh, the pattern of "
                         "DEMdemo_FlexibleMesh.cpp:203-255); every update makes the next step start with a contact detection")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE configs[4] flavour: polydisperse spheres + a user cohesion model compiled at run time")
    ap.add_argument("--custom-model", action="store_true",
                    help="the headline bed of three-sphere clumps with the --config5 fragment as its force model (run-time compiled, "
                         "one contact wildcard): the tile pass of a user model at the headline's contact density")
    ap.add_argument("--tile-policy", type=int, default=-1, metavar="N",
                    help="deme_set_tile_policy: a run-time compiled model takes the tile pass when a tile holds >= N contacts on average "
                         "(default: the library's 320; 0 = always)")
    ap.add_argument("--bin-multiple", type=float, default=5.0,
                    help="bin edge as a multiple of the smallest sphere radius (SetInitBinSizeAsMultipleOfSmallestSphere)")
    ap.add_argument("--async-detection", type=int, default=0, metavar="D",
                    help="start each contact detection D steps before its list is due, on a stream of its own beside the steps "
                         "(deme_set_async_detection; slabs of a halo group included).  0 = lock-step")
    ap.add_argument("--cross-contacts", default="both", choices=["both", "once"],
                    help="contacts that straddle a cut: evaluated by both slabs, each with its own history (default), or by the left "
                         "slab only, which returns the reaction every step (deme_halo_group_set_cross_contacts)")
    ap.add_argument("--adaptive", default="off", choices=["off", "bin", "freq", "both"],
                    help="let the engine tune the bin size / the update frequency on device timers during the pre-settling and "
                         "warm-up (the reference's default mode); frozen before the timed region.  Default: off (fixed K and bin size)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: order the ghost exchange on the compute stream")
    ap.add_argument("--halo", default="library", choices=["library", "python"],
                    help="N > 1: who drives the per-step ghost exchange -- the library's C++ loop calling RCCL itself "
                         "(deme_halo_group_step, default) or the round-1 Python loop over torch.distributed P2P")
    ap.add_argument("--slabs", type=int, default=1,
                    help="1-GPU harness of the N > 1 path: cut this rank's bed into S x-slabs held by this one process, exchanged "
                         "through RCCL sends to self by the library loop (measures the loop's host cost; not a scaling number)")
    ap.add_argument("--migrate-every", type=int, default=0, metavar="M",
                    help="slab runs with the library loop: every M steps the slabs hand over the clumps that crossed a cut "
                         "(deme_halo_group_migrate: device-side, RCCL between ranks) INSIDE the timed loop; 0 = never")
    ap.add_argument("--drift", type=float, default=0.0, metavar="VX",
                    help="give every clump this lateral velocity [m/s] after the pre-settling, so that clumps really cross the cuts "
                         "(with --migrate-every)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--state-cache", default="",
                    help="profiling aid (1 GPU): file that keeps the pre-settled bed (owner state, contact list, wildcards) so "
                         "that repeated rocprofv3 passes of the same command skip the 30 000 untimed steps; created when absent")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--watchdog", type=float, default=1500.0,
                    help="seconds after which the process gives up with exit code 3 (a rank stuck in a collective must not hold "
                         "the others, and the launcher, for ever); 0 = none")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if args.clumps_total:
        args.clumps = max(1, args.clumps_total // max(1, args.gpus))

    # Only the JSON line may reach stdout: RCCL prints a version banner there when a communicator is created.  The real stdout
    # is kept aside and file descriptor 1 points at stderr for everything else this process (and its C libraries) writes.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.watchdog > 0:
        import threading

        def _give_up():
            print(f"[bench] rank {os.environ.get('RANK', '0')}: not done after {args.watchdog:.0f} s -- giving up", file=sys.stderr, flush=True)
            os._exit(3)
        wd = threading.Timer(args.watchdog, _give_up)
        wd.daemon = True
        wd.start()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started as {world} rank(s): launch it plainly (it spawns its own ranks) or with "
                         f"torch.distributed.run --nproc-per-node {args.gpus}")
    if args.launch_check:  # no GPU work: the rendezvous and the rank count only
        seen = 1
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t_ = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(t_)
            seen = int(t_.item())
            dist.destroy_process_group()
        if rank == 0:
            os.write(json_fd, (json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": seen,
                                           "clumps_per_gpu": args.clumps, "scaling": "strong" if args.clumps_total else "weak"}) + "\n").encode())
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # DEME_BENCH_VIA_HOST=1: plumbing test of the N > 1 path on a box with fewer GPUs than ranks (ranks share GPUs,
    # ghosts travel over gloo through host memory); never set for a measurement
    via_host = os.environ.get("DEME_BENCH_VIA_HOST") == "1"
    if via_host:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if via_host:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if via_host else "cuda"

    pkg = entry.load_package()
    if args.config5:
        b = build_config5(pkg, args.clumps * world, args.seed, args.cd_freq)
    else:
        # N > 1: every rank builds its own slab of the N-times-longer bed (its clumps and their ghosts), not the whole bed
        b = build_bed(pkg, args.clumps, args.seed, args.cd_freq, x_mult=world, order=args.order, bin_multiple=args.bin_multiple,
                      slab=(rank, world, HALO) if (world > 1 and not args.mesh_triangles) else None)
    if args.custom_model and not args.config5:
        b.materials[0]["Cohesion"] = 0.002
        b.SetMustPairwiseMatProp(["Cohesion"])
        b.DefineContactForceModel(COHESIVE_FRAGMENT)
        b.SetPerContactWildcards(["contact_age"])
    mesh_obj = None
    if args.mesh_triangles:
        lo, hi = b.user_box_min, b.user_box_max
        n_side = max(2, int(round((args.mesh_triangles / 2) ** 0.5)))
        v, f = pkg.model.plate_mesh(n_side, n_side, float(hi[0] - lo[0]) * 0.98, float(hi[1] - lo[1]) * 0.98, z=0.0, wavy=0.002)
        m = b.AddMeshObject(v, f, 0)
        m.SetInitPos(((lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 0.021))  # just under the lowest spheres of the lattice
        mesh_obj = m
    p, sc = b.Initialize()
    halo, part = None, None
    group, extra_ctx, slab_parts = None, [], None
    if world > 1:
        if getattr(b, "slab_edges", None) is not None:
            part = pkg.decomp.decompose(b.arrays, b.counts, b.slab_x, world, HALO, edges=b.slab_edges, only_rank=rank)[rank]
        else:
            x = np.concatenate([bb.xyz for bb in b.batches])[:, 0]
            part = pkg.decomp.decompose(b.arrays, b.counts, x, world, HALO)[rank]
        sc = part["scene"]
        n_own = part["n_own"]
    elif args.slabs > 1:
        x = np.concatenate([bb.xyz for bb in b.batches])[:, 0]
        slab_parts = pkg.decomp.decompose(b.arrays, b.counts, x, args.slabs, HALO)
        n_own = int(sc.nOwnerClumps)
        sc_full = sc
        sc = slab_parts[0]["scene"]
    else:
        n_own = int(sc.nOwnerClumps)
    ctx = pkg.Context(local_rank)
    if world > 1:
        # ghost traffic (RCCL, ordered against torch's current stream) and the DEM kernels share one
        # non-default torch stream, so pack -> send/recv -> unpack -> step need no host synchronisation
        side = torch.cuda.Stream()
        torch.cuda.set_stream(side)
        ctx.set_stream(side.cuda_stream)
    ctx.set_params(p)
    ctx.upload_scene(sc)
    b.compile_into(ctx)  # user force model / prescriptions, if the scene has any
    if args.tile_policy >= 0:
        ctx.set_tile_policy(args.tile_policy)
    if args.async_detection:
        ctx.set_async_detection(args.async_detection)
    if args.adaptive != "off":
        ctx.set_adaptive(bin_size=args.adaptive in ("bin", "both"), update_freq=args.adaptive in ("freq", "both"),
                         bin_observe=5, max_update_freq=200, freq_observe=3)
    if world > 1 and args.halo == "library" and not via_host:
        # the library's own loop: RCCL communicator from a unique id that rank 0 generates and torch.distributed hands round.
        # Whatever goes wrong while setting it up (RCCL not loadable, communicator refused) is agreed on by all ranks, which then
        # fall back to the Python loop over torch.distributed -- a scaling run should not die of a set-up problem.
        ok = 1
        try:
            uid = [pkg.abi.halo_unique_id() if rank == 0 else None]
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: {e}", file=sys.stderr, flush=True)
            uid, ok = [None], 0
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is None:
            ok = 0
        if ok:
            try:
                group = pkg.abi.HaloGroup(rank=rank, world=world, device=local_rank, unique_id=uid[0])
                group.attach(ctx, part, left=rank - 1 if rank > 0 else None, right=rank + 1 if rank + 1 < world else None)
                if args.migrate_every:
                    group.set_slab(ctx, globalise_part(part, b), HALO)
                if args.cross_contacts == "once":
                    group.set_cross_contacts(True)
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: library halo loop unavailable ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
                ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not int(flag.item()):
            if os.environ.get("DEME_BENCH_STRICT", "0") not in ("", "0"):
                # a scaling measurement is a measurement of the LIBRARY loop: no silent change of what is being timed
                raise SystemExit(f"[bench] rank {rank}: the library's halo loop could not be set up on every rank and DEME_BENCH_STRICT is set "
                                 f"(the Python loop over torch.distributed is not what --gpus {world} is meant to measure)")
            if group is not None:
                group.close()
            group = None
            args.halo = "python"
    if world > 1 and group is None:
        halo = Halo(pkg, ctx, part, rank, world, torch, dist, via_host=via_host, overlap=not args.no_overlap)
        if not via_host:
            halo.probe_overlap()
    elif slab_parts is not None:
        for pt in slab_parts[1:]:
            c2 = pkg.Context(local_rank)
            c2.set_params(p), c2.upload_scene(pt["scene"])
            b.compile_into(c2)
            if args.async_detection:
                c2.set_async_detection(args.async_detection)
            extra_ctx.append(c2)
        all_ctx = [ctx] + extra_ctx
        group = pkg.abi.HaloGroup(rank=0, world=1, device=local_rank)
        for i, (c_, pt) in enumerate(zip(all_ctx, slab_parts)):
            group.attach(c_, pt, left=all_ctx[i - 1] if i else None, right=all_ctx[i + 1] if i + 1 < len(all_ctx) else None)
            if args.migrate_every:
                group.set_slab(c_, pt, HALO)
        if args.cross_contacts == "once":
            group.set_cross_contacts(True)

    host_enqueue = {"s": 0.0, "steps": 0}
    migration = {"calls": 0, "moved": 0, "since": 0, "s": 0.0}
    mesh_state = {"since": 0, "updates": 0, "t": 0.0}

    def deform_mesh():
        # the plate's nodes in the mesh's own frame with a ripple that travels in x: what a co-simulated structural solver would
        # hand back (DEMdemo_FlexibleMesh.cpp:216-250)
        mesh_state["t"] += args.mesh_update_every * float(p.h)
        v = mesh_obj.vertices.copy()
        v[:, 2] += np.float32(2e-4) * np.sin(np.float32(40.0) * v[:, 0] - np.float32(2000.0 * mesh_state["t"])).astype(np.float32)
        f = mesh_obj.faces
        ctx.update_tri_nodes(v[f[:, 0]], v[f[:, 1]], v[f[:, 2]])
        mesh_state["updates"] += 1

    def run(n):
        if mesh_obj is not None and args.mesh_update_every > 0 and halo is None and group is None:
            left = n
            while left > 0:
                k = min(left, args.mesh_update_every - mesh_state["since"])
                ctx.step(k)
                left -= k
                mesh_state["since"] += k
                if mesh_state["since"] >= args.mesh_update_every:
                    deform_mesh()
                    mesh_state["since"] = 0
            return
        if group is not None:
            left = n
            while left > 0:
                k = min(left, args.migrate_every - migration["since"]) if args.migrate_every else left
                t_ = time.perf_counter()
                group.step(k)
                host_enqueue["s"] += time.perf_counter() - t_
                host_enqueue["steps"] += k
                left -= k
                migration["since"] += k
                if args.migrate_every and migration["since"] >= args.migrate_every:
                    t_ = time.perf_counter()
                    migration["moved"] += group.migrate()
                    migration["s"] += time.perf_counter() - t_
                    migration["calls"] += 1
                    migration["since"] = 0
        elif halo is None:
            ctx.step(n)
        else:
            for _ in range(n):
                halo.step()

    def local_sync():
        if group is not None:
            group.sync()
        ctx.sync()
        torch.cuda.synchronize()

    def barrier():
        local_sync()
        if world > 1:
            dist.barrier()

    # untimed pre-settling: the lattice (no contacts at t=0) is dropped at 1 m/s and compacts; the run stops early once the
    # (job-wide) contact count has plateaued
    done, last_nc = 0, -1
    t_pre = time.perf_counter()
    cache = args.state_cache if (args.state_cache and world == 1) else ""
    if cache and os.path.exists(cache):  # resume the settled bed through the restart path (deme_seed_contacts)
        z = np.load(cache)
        ctx.upload_state({k: z[k] for k in z.files if k not in ("idA", "idB", "ctype", "wc", "presettle")})
        ctx.seed_contacts(z["idA"], z["idB"], z["ctype"], z["wc"] if z["wc"].size else None)
        done = args.presettle = int(z["presettle"])
    while done < args.presettle:
        chunk = min(1000, args.presettle - done)
        run(chunk)
        done += chunk
        nc = int(ctx.counts().nContacts) + sum(int(c_.counts().nContacts) for c_ in extra_ctx)
        if world > 1:  # the same stopping rule on the job's total, so that every N times a bed in the same state
            tot_nc = torch.tensor([float(nc)], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tot_nc, op=dist.ReduceOp.SUM)
            nc = int(tot_nc.item())
        if rank == 0 and args.verbose:
            print(f"[presettle] step {done} contacts {nc} t {time.perf_counter() - t_pre:.1f}s", file=sys.stderr, flush=True)
        if done >= 4000 and last_nc > 0 and abs(nc - last_nc) < 0.002 * nc:
            break
        last_nc = nc
    args.presettle = done
    if cache and not os.path.exists(cache):
        st = ctx.download_state()
        a_, b_, t_, _ = ctx.contacts()
        nW = int(p.nContactWildcards)
        wc = np.stack([ctx.wildcard(w) for w in range(nW)], 1) if nW else np.zeros((0, 0), np.float32)
        np.savez(cache, idA=a_, idB=b_, ctype=t_, wc=wc, presettle=done,
                 **{k: st[k] for k in st if k not in ("aX", "aY", "aZ", "alphaX", "alphaY", "alphaZ")})
    # Phase the K-step cadence so that the timed region carries its contact detections whatever --steps is: single steps until
    # a detection has just run, then as many as put the next one on the FIRST timed step (after the warm-up).  The region then
    # holds ceil(steps / K) detections -- never fewer than its share (round 1's pre-settling in multiples of K left none in a
    # 20-step region).  Every rank steps in lock-step, so the phase is the same job-wide.
    K = int(args.cd_freq)
    if K > 1 and args.adaptive == "off":
        d0 = int(ctx.counts().nDetections)
        for _ in range(K + 1):
            run(1)
            if int(ctx.counts().nDetections) > d0:
                break
        # stepsSinceCD == 1 now; the first timed step detects when it equals K there.  With --async-detection D the cycle of a
        # detection BEGINS D steps before its list is due (owner snapshot, then part 1 and 2 beside the next D steps): that
        # beginning is put on the first timed step instead, so that the region holds all of the detection's work
        lead = args.async_detection if 0 < args.async_detection < K else 0
        run((K - lead - 1 - args.warmup) % K)
    if args.drift:
        for c_ in [ctx] + extra_ctx:
            st_ = c_.download_state()
            st_["vX"] = st_["vX"] + np.float32(args.drift)  # (fixed owners are held at rest by the integrator, ghosts are refreshed below)
            c_.upload_state({k: st_[k] for k in st_ if k not in ("aX", "aY", "aZ", "alphaX", "alphaY", "alphaZ")})
        if group is not None:
            group.exchange()
    run(args.warmup)
    migration.update(calls=0, moved=0, s=0.0)
    adaptive_state = None
    if args.adaptive != "off":  # what the controllers settled on; frozen for the timed region
        adaptive_state = ctx.adaptive_state()
        ctx.set_adaptive()
        args.cd_freq = adaptive_state[1]
        if rank == 0 and args.verbose:
            print(f"[adaptive] bin size {adaptive_state[0]:.6g} K {adaptive_state[1]} changes {adaptive_state[2:]}", file=sys.stderr)
    # every 8th launch of the force / integration kernels is bracketed with HIP events (the detection always is: its timer
    # only ticks once per K steps); timing every launch costs 4.7 % of the step in dispatch gaps
    stride = max(1, min(8, args.steps // 5))  # short runs (--steps < 40) time more of their launches so that the mean exists (>= 5 samples)
    ctx.set_timing(0 if os.environ.get("DEME_BENCH_NO_KERNEL_TIMING") else stride)
    ctx.kernel_time_reset()
    det_before = int(ctx.counts().nDetections)
    host_enqueue["s"], host_enqueue["steps"] = 0.0, 0
    if group is not None:
        group.host_time(reset=True)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    local_sync()
    dt_local = time.perf_counter() - t0  # this rank's own clock, before it waits for the others (reported per rank; not the metric)
    barrier()
    dt = time.perf_counter() - t0
    total_clumps = n_own
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = torch.tensor([float(n_own)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt, total_clumps = float(tmax.item()), int(tot.item())
    # per rank: what it holds and how long ITS timed region took -- the first real multi-GPU run shows imbalance at a glance
    per_rank = None
    if world > 1:
        n_ghost = int(sc.nOwnerClumps) - int(n_own)
        mine = torch.tensor([float(n_own), float(n_ghost), float(ctx.counts().nContacts), 1e3 * dt_local / args.steps], dtype=torch.float64, device=red_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"owners": [int(v[0].item()) for v in allr], "ghosts": [int(v[1].item()) for v in allr],
                    "contacts": [int(v[2].item()) for v in allr], "ms_per_step": [round(float(v[3].item()), 5) for v in allr]}
    n_det = int(ctx.counts().nDetections) - det_before
    f_ms, f_n = ctx.kernel_time_ms("calc_forces")
    i_ms, _ = ctx.kernel_time_ms("integrate")
    d_ms, d_n = ctx.kernel_time_ms("detect")
    d_async = None
    if args.async_detection:  # the two parts of an asynchronous detection: beside the steps on its own stream / at the swap
        p1, n1 = ctx.kernel_time_ms("detect_async_part1")
        p2, _ = ctx.kernel_time_ms("detect_async_part2")
        if n1:
            d_async = {"part1_beside_the_steps_ms": p1, "part2_at_the_swap_ms": p2, "count": int(n1)}
            d_ms = (d_ms * d_n + (p1 + p2) * n1) / (d_n + n1)
    c = ctx.counts()
    value = total_clumps * args.steps / dt
    fbytes = force_kernel_bytes(int(sc.nOwners), int(sc.nSpheres), int(c.nContacts), int(p.nContactWildcards))
    fk_name, tile_halo, tile_list = ctx.force_kernel()
    fused = fk_name.startswith("k_tile_step")
    # the one-kernel step (deme_tile_step.h) integrates too: its algorithmic bytes are the force pass's (owner state is read ONCE) plus
    # the integrator's write-back, 54 B per owner; what the kernel pays for closing its tiles -- the contacts that
    # straddle two tiles are evaluated by both -- is NOT counted
    force_only_bytes = fbytes
    if fused:
        fbytes += 54 * int(sc.nOwners)
    achieved = fbytes / (f_ms * 1e-3) / 1e9 if f_ms > 0 else 0.0
    par = f"{world} x-slab(s)"
    if group is not None:
        n_ex, ex_bytes = group.stats()
        nst = max(1, host_enqueue["steps"])
        phases = " / ".join("%.1f" % (u / nst) for u in group.host_time())
        par = (f"{world if world > 1 else args.slabs} x-slab(s){' held by ONE process on one GPU (harness)' if world == 1 else ''}; ghost exchange every "
               f"step by the library's C++ loop: ncclSend / ncclRecv in one group on an exchange stream, overlapped with the interior "
               f"force pass ({ex_bytes} B sent per step by this process); host time to enqueue a step "
               f"{1e6 * host_enqueue['s'] / nst:.1f} us (interior pass / pack / RCCL group / unpack + boundary pass + integration: "
               f"{phases} us)")
    if halo:
        par += (", overlapped with the interior force evaluation on a second stream" if halo.overlap else "")
        par += f", ghost exchange every step over {'gloo via host memory (PLUMBING TEST, not a measurement)' if via_host else 'RCCL'} ({halo.bytes_per_step} B sent per step by rank 0)"
    # the attainable HBM rate of THIS box beside the nominal 8 TB/s -- the library's own 16 B / lane streaming copy kernel
    # (deme_copy_rate_probe; 1 GiB each way, beyond the 256 MiB Infinity Cache); the guide's figure for such a copy is 6.29 TB/s.
    # (Round 4 used torch's device-to-device copy here, which reaches 4.7-5.4 TB/s: a fraction of THAT flattered the kernel.)
    copy_gbs = None
    if rank == 0 and not args.no_cpu_baseline:  # (profiling and A/B runs pass --no-cpu-baseline: their traces hold the bench's kernels only)
        try:
            copy_gbs = pkg.abi.copy_rate_probe(local_rank)
        except Exception as e:  # (a box short of 2 GiB of free HBM)
            print(f"[bench] attainable-rate probe skipped: {e}", file=sys.stderr)
    halo_loop = "library" if group is not None else ("python" if halo is not None else None)
    rccl_ranks = None
    if group is not None:
        rccl_ranks = group.comm_count()
    elif halo is not None and not via_host:
        rccl_ranks = dist.get_world_size()
    out = {
        "metric": "clump*steps/s", "value": value, "unit": "clump*steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if args.clumps_total else "weak",
        # who moved the ghosts, and how many ranks the communicator that moved them spans (ncclCommCount of the library's own
        # communicator; the world size of torch's for the Python loop): a run whose count differs from --gpus exits non-zero
        "halo_loop": halo_loop, "rccl_ranks": rccl_ranks, "per_rank": per_rank,
        "migration": ({"every_steps": args.migrate_every, "calls_in_timed_region": migration["calls"], "clumps_moved": migration["moved"],
                       "host_seconds": migration["s"], "drift_m_per_s": args.drift} if args.migrate_every else None),
        "vs_baseline": value / README_CLUMP_STEPS_PER_S, "dtype": "f32 physics / f64 geometry", "arith_mode": ctx.arith_mode(), "data": "synthetic",
        "config": {"workload": ("BASELINE configs[4] flavour: polydisperse spheres (8 templates, r..3r) with a run-time compiled "
                                "cohesion model" if args.config5 else
                                f"BASELINE configs[1]'s bed ({args.clumps} three-sphere clumps) with the run-time compiled cohesion model of configs[4]" if args.custom_model else
                                f"BASELINE configs[{2 if args.clumps_total else 1}]: {args.clumps} three-sphere clumps (3_clump.csv x0.005) per GPU in a box, gravity settling"
                                + (f"; one bed {world} times as long cut into {world} x-slabs (configs[2] flavour)" if world > 1 else ""))
                               + (f" + {int(sc.nTri)}-triangle plate (configs[3] flavour)" if int(sc.nTri) else "")
                               + (f", a deformable mesh: nodes rewritten every {args.mesh_update_every} steps ({mesh_state['updates']} updates so far)"
                                  if (int(sc.nTri) and args.mesh_update_every) else ""),
                   "clumps_total": total_clumps, "owners_this_rank": int(sc.nOwners), "spheres_this_rank": int(sc.nSpheres),
                   "contacts_this_rank": int(c.nContacts), "bin_sphere_touches": int(c.nBinSphereTouches),
                   "triangles": int(sc.nTri), "cd_every": args.cd_freq, "presettle_steps": args.presettle,
                   "clump_numbering": args.order, "bin_multiple": args.bin_multiple, "async_detection_lead": args.async_detection,
                   "cross_cut_contacts": (args.cross_contacts if group is not None else None),
                   "margin_safety": {"multiplier": float(p.expSafetyMulti), "adder_m_per_s": float(p.expSafetyAdder)},
                   "force_model": ("user fragment via hipRTC: frictionless Hertz + cohesion, 1 wildcard" if (args.config5 or args.custom_model)
                                   else "Hertzian (history, 4 wildcards)"), "integrator": "extended Taylor", "h": p.h,
                   "parallelism": par,
                   "adaptive": (None if adaptive_state is None else
                                {"mode": args.adaptive, "bin_size": adaptive_state[0], "cd_every": adaptive_state[1],
                                 "bin_size_changes": adaptive_state[2], "update_freq_changes": adaptive_state[3]}),
                   "vs_baseline_ref": "reference README.md:48, ~1h for 1e6 clumps x 1e6 steps on 2x RTX 3080"},
        "roofline": {"kernel": fk_name + (" (hipRTC)" if (args.config5 or args.custom_model) else ""), "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "attainable_copy_GBs": copy_gbs, "attainable_copy_kernel": "deme_copy_rate_probe (libdeme_hip: hand-written 16 B / lane copy, best of the plain and the four-in-flight streaming form at 8 / 16 / 32 workgroups per CU; 1 GiB read + 1 GiB written)",
                     "guide_copy_GBs": GUIDE_COPY_GBS, "frac_of_attainable": (achieved / copy_gbs if copy_gbs else None),
                     "frac_of_guide_copy": achieved / GUIDE_COPY_GBS,
                     "algorithmic_bytes_per_launch": fbytes, "avg_launch_ms": f_ms, "launches": int(f_n),
                     "fused_step": ({"what": "contact forces + accumulation + integration in one kernel (closed owner tiles)",
                                     "force_pass_bytes": force_only_bytes, "integration_write_back_bytes": 54 * int(sc.nOwners),
                                     "frac_on_the_force_pass_bytes_alone": force_only_bytes / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if f_ms > 0 else None}
                                    if fused else None),
                     # SURVEY 8d adds 24 B per owner for a force kernel that reduces in-kernel, as the owner-tile pass does (its
                     # per-owner sums are its output); `achieved` / `frac` above do NOT count them (the conservative figure)
                     "frac_counting_the_in_kernel_reduction": ((fbytes + 24 * int(sc.nOwners)) / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                               if (f_ms > 0 and fk_name.startswith("k_tile")) else None),
                     "tile": ({"owners_per_tile": 128, "largest_tile_foreign_owners": tile_halo, "largest_tile_local_list": tile_list}
                              if fk_name.startswith("k_tile") else None),
                     "launch_sampling": f"every {stride}{'th' if stride > 3 else ('st', 'nd', 'rd')[stride - 1]} launch inside the timed region is bracketed with HIP events"},
        "kernels_ms": {"calc_forces": f_ms, "integrate": i_ms, "detect_update": d_ms, "detect_updates": int(n_det),
                       # the same step with the detection spread over its K steps (what a run of many K-cycles converges to;
                       # `ms_per_step` above is the measured wall time of exactly `steps` steps, ceil(steps / K) detections included)
                       "amortised_ms_per_step": (f_ms + i_ms + (d_ms / args.cd_freq if args.cd_freq else d_ms)),
                       "detections_in_timed_region": int(n_det), "async_detection": d_async},
    }
    assert n_det >= 1 or args.adaptive != "off", "the timed region contains no contact detection: the phase alignment failed"
    out["roofline"].update(pmc_traffic(int(c.nContacts), out["roofline"]["kernel"]))
    if os.environ.get("DEME_PMC_CALIB") == "1":
        pmc_calibration(torch)
    if rank == 0:
        main_bed = None
        if not args.no_cpu_baseline and world == 1 and group is None and not (args.config5 or args.custom_model or args.mesh_triangles):
            main_bed = (p, sc, ctx.download_state())
        out["cpu_baseline"] = cpu_baseline(pkg, args.seed, args.cd_freq, main=main_bed) if (not args.no_cpu_baseline and world == 1) else None
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if world > 1 and not via_host and rccl_ranks != world:
        raise SystemExit(f"[bench] the ghost exchange ran on a communicator of {rccl_ranks} rank(s), not {world}: this is not a {world}-GPU measurement")


if __name__ == "__main__":
    main()
