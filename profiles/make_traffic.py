#!/usr/bin/env python3
"""profiles/<round>/traffic.json from the two rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE).

    python profiles/make_traffic.py profiles/r01 r01g

Corrections (MI355X_MICROARCH.md, HBM section + the calibration kernels bench.py launches under
DEME_PMC_CALIB=1 inside the same passes):
  * FETCH_SIZE is in KiB and reports exactly 1/2 of a wide coalesced streaming read: the 1 GiB device copy in the
    pass reads 1 GiB and is reported as 524288 KiB  -> factor 2;
  * WRITE_SIZE is in KiB and exact for streaming writes (same copy: 1048576 KiB)               -> factor 1;
  * a random 64-byte row gather is reported at ~124 B per row (4 Mi rows: 523505 KiB incl. 16 MiB of
    half-counted indices), i.e. NOT halved -- so for a kernel that mixes streams and gathers "2 x FETCH_SIZE" is
    an upper bound.  Both the prescribed figure (2 x FETCH + WRITE) and the lower bound (streams doubled, the
    remainder taken at face value) are stored; bench.py quotes the prescribed one.
"""
import json
import re
import sys


def read(path, counter):
    """{kernel: (calls, value of `counter`)} from a summarize_pmc.py table (any number of counter columns)"""
    rows, cols = {}, None
    for line in open(path):
        if line.startswith("#"):
            continue
        if line.startswith("kernel"):
            cols = line.split()[2:]
            continue
        parts = line.rstrip().split()
        n = len(cols)
        vals = parts[-n:]
        rows[" ".join(parts[:-(n + 1)])] = (int(parts[-(n + 1)]), float(vals[cols.index(counter)]))
    return rows


def main():
    d, tag = sys.argv[1], sys.argv[2]
    fetch, write = read(f"{d}/{tag}_fetch_pmc.txt", "FETCH_SIZE"), read(f"{d}/{tag}_write_pmc.txt", "WRITE_SIZE")
    bench = json.load(open(f"{d}/{tag}_bench.json"))
    nc = bench["config"]["contacts_this_rank"]
    out = {"contacts": nc, "source": [f"{tag}_fetch_pmc.txt", f"{tag}_write_pmc.txt"], "unit": "bytes per launch", "kernels": {}}
    # (the profiler prints all template arguments -- <MODEL, MESH, REC> since round 4 --, the library's deme_force_kernel_name the first two)
    force = next(k for k in ("k_tile_forces<0, false, false>", "k_tile_forces<0, false>", "k_tile_forces<0>", "k_forces_fast<0>", "k_calc_forces<0, 0>") if k in fetch)
    out["force_kernel"] = "k_tile_forces<0, false>" if force.startswith("k_tile_forces<0, false") else force
    for k in (force, "k_integrate<true>", "k_sweep"):
        if k not in fetch or k not in write:
            continue
        f_kib, w_kib = fetch[k][1], write[k][1]
        out["kernels"][k] = {"FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib, "traffic_prescribed": int((2 * f_kib + w_kib) * 1024)}
    # force kernel: coalesced streams read per contact = gather record (16; 8 in the tile pass) + wildcards (16)
    if force != out["force_kernel"]:
        out["kernels"][out["force_kernel"]] = out["kernels"].pop(force)
        force = out["force_kernel"]
    fk = out["kernels"][force]
    stream_read = nc * (24 if force.startswith("k_tile") else 32)
    raw = fk["FETCH_SIZE_KiB"] * 1024
    fk["traffic_lower_bound"] = int(stream_read + max(0.0, raw - stream_read / 2) + fk["WRITE_SIZE_KiB"] * 1024)
    out["traffic_bytes_per_launch"] = fk["traffic_prescribed"]
    out["calibration"] = {"copy_1GiB_FETCH_KiB": fetch.get("__amd_rocclr_copyBuffer"), "copy_1GiB_WRITE_KiB": write.get("__amd_rocclr_copyBuffer"),
                          "gather_4Mi_rows_64B_FETCH_KiB": fetch.get("at::native::vectorized_gather_kernel<16, long>"),
                          "note": "copyBuffer rows average 3 calibration copies of 1 GiB with small state copies; see module docstring"}
    out["integrator"] = out["kernels"].get("k_integrate<true>")
    json.dump(out, open(f"{d}/traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
