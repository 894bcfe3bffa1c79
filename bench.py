#!/usr/bin/env python3
"""bench.py -- clump*steps/s of the MI355X-native DEM hot path on BASELINE.json configs[1]
(1M three-sphere clumps in a box, gravity settling), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the whole bed: (contact detection every --cd-freq steps:
margins, binning, bin-sorted sweep, history map) + contact forces + integration.  State is resident in
HBM before the timed region.  Prints ONE JSON line on rank 0.

N > 1: the bed is cut into N slabs along x (weak scaling: every rank simulates its own
--clumps-sized slab); see DESIGN.md section "Multi-GPU" for what is and is not exchanged.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)
README_CLUMP_STEPS_PER_S = 1e6 * 1e6 / 3600.0  # reference README.md:48, two RTX 3080 ("around 1 hour")


def build_bed(pkg, n_clumps, seed, cd_freq, rank=0):
    b = pkg.model.packed_bed(n_clumps, seed=seed + rank, cd_freq=cd_freq, aspect=(1.0, 1.0, 0.05),
                             spacing_mult=3.0, jitter=0.05, bin_multiple=4.0, init_vz=-1.0)
    b.SetExpandSafetyMultiplier(1.2)
    b.SetExpandSafetyAdder(0.02)
    return b


def force_kernel_bytes(n_owners, n_spheres, n_contacts, n_w):
    """Algorithmic HBM bytes of ONE contact-force launch (SURVEY 8d / DESIGN.md):
    N_c*(9 + 8*n_w) + N_o*57 + N_s*7  (contact ids+type, wildcards read+write, owner and sphere state
    once each).  The per-contact contribution records this design writes instead of atomics are
    implementation traffic and are NOT counted."""
    return n_contacts * (9 + 8 * n_w) + n_owners * 57 + n_spheres * 7


def cpu_baseline(pkg, seed, budget_s=15.0):
    """The CPU oracle (oracle/, a port: the reference has no CPU path) on a bounded sample of the
    same workload recipe, all host cores via OpenMP."""
    orc = entry.load_oracle()
    n = 20000
    b = build_bed(pkg, n, seed, cd_freq=10)
    p, sc = b.Initialize()
    sim = orc.make_sim(pkg, p, sc)
    sim.step(10)  # first detection + page-in
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget_s:
        sim.step(10)
        steps += 10
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "clump*steps/s", "cores": int(orc.num_threads()), "kind": "port",
            "sample": f"{n} three-sphere clumps x {steps} steps (same recipe, cd every 10), oracle/deme_oracle.cpp -O2 OpenMP, "
                      f"{int(sim.counts().nContacts)} contacts at end"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--clumps", type=int, default=1_000_000, help="clumps per GPU")
    ap.add_argument("--cd-freq", type=int, default=10, help="contact detection every K steps (0: every step)")
    ap.add_argument("--presettle", type=int, default=30000, help="untimed steps that let the lattice start settling")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"

    pkg = entry.load_package()
    b = build_bed(pkg, args.clumps, args.seed, args.cd_freq, rank)
    p, sc = b.Initialize()
    ctx = pkg.Context(local_rank)
    ctx.set_params(p)
    ctx.upload_scene(sc)

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # untimed pre-settling: the lattice (no contacts at t=0) is dropped at 1 m/s and compacts; stop when the
    # contact count has plateaued or the step budget is spent
    done, last_nc, log = 0, -1, []
    t_pre = time.perf_counter()
    while done < args.presettle:
        chunk = min(1000, args.presettle - done)
        ctx.step(chunk)
        done += chunk
        nc = int(ctx.counts().nContacts)
        log.append((done, nc))
        if rank == 0 and args.verbose:
            st = ctx.download_state()
            vmax = float(np.sqrt(st["vX"] ** 2 + st["vY"] ** 2 + st["vZ"] ** 2).max())
            print(f"[presettle] step {done} contacts {nc} vmax {vmax:.3f} t {time.perf_counter() - t_pre:.1f}s",
                  file=sys.stderr, flush=True)
        if world == 1 and done >= 4000 and last_nc > 0 and abs(nc - last_nc) < 0.002 * nc:
            break
        last_nc = nc
    args.presettle = done
    ctx.step(args.warmup)
    ctx.set_timing(True)
    ctx.kernel_time_reset()
    barrier()
    t0 = time.perf_counter()
    ctx.step(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    f_ms, f_n = ctx.kernel_time_ms("calc_forces")
    i_ms, i_n = ctx.kernel_time_ms("integrate")
    d_ms, d_n = ctx.kernel_time_ms("detect")
    c = ctx.counts()
    n_clumps = int(sc.nOwnerClumps)
    total_clumps = n_clumps * world
    value = total_clumps * args.steps / dt
    fbytes = force_kernel_bytes(int(sc.nOwners), int(sc.nSpheres), int(c.nContacts), int(p.nContactWildcards))
    achieved = fbytes / (f_ms * 1e-3) / 1e9 if f_ms > 0 else 0.0
    out = {
        "metric": "clump*steps/s", "value": value, "unit": "clump*steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": value / README_CLUMP_STEPS_PER_S, "dtype": "f32 physics / f64 geometry", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 1M three-sphere clumps (3_clump.csv x0.005) in a box, gravity settling",
                   "clumps_per_gpu": n_clumps, "spheres_per_gpu": int(sc.nSpheres), "contacts": int(c.nContacts),
                   "bin_sphere_touches": int(c.nBinSphereTouches), "cd_every": args.cd_freq,
                   "presettle_steps": args.presettle, "force_model": "Hertzian (history, 4 wildcards)",
                   "integrator": "extended Taylor", "h": p.h, "parallelism": f"slab x{world}",
                   "vs_baseline_ref": "reference README.md:48, ~1h for 1e6 clumps x 1e6 steps on 2x RTX 3080"},
        "roofline": {"kernel": "k_calc_forces<0>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": fbytes, "avg_launch_ms": f_ms, "launches": int(f_n)},
        "kernels_ms": {"calc_forces": f_ms, "integrate": i_ms, "detect_update": d_ms, "detect_updates": int(d_n)},
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, args.seed)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
