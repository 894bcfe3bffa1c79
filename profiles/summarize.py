#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV into a short per-kernel table.

    python profiles/summarize.py <kernel_trace.csv> <out.txt> [last_n_force_launches]

Only dispatches from the last N launches of the contact-force kernel onward are counted (the timed
region of bench.py); kernel names are shortened.  The big trace is not kept.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("deme_dev::", "")
    m = re.search(r"rocprim::[A-Za-z0-9_]+::detail::(?:trampoline_kernel<.*?detail::)?([a-z_]+(?:<[^<>]*>)?)", name)
    if name.startswith("rocprim") and m:
        return "rocprim:" + m.group(1)[:50]
    return name[:60]


def main():
    path, out = sys.argv[1], sys.argv[2]
    last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")))
    rows.sort()
    start = 0
    if last_n:
        # the hot variant of the contact-force kernel: built-in models, or the run-time compiled user model
        idx = [i for i, r in enumerate(rows) if ("k_calc_forces" in r[2] and "1>" not in short(r[2])) or "deme_custom_forces_ss" in r[2] or "k_forces_fast" in r[2] or "k_tile_forces" in r[2] or "k_tile_step" in r[2]]
        if len(idx) >= last_n:
            start = idx[-last_n]
    agg = defaultdict(lambda: [0, 0, 10 ** 18, 0, "", ""])
    for s, e, n, vg, lds in rows[start:]:
        a = agg[short(n)]
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
        a[4], a[5] = vg, lds
    tot = sum(a[1] for a in agg.values())
    with open(out, "w") as f:
        f.write(f"# source: {path}; dispatches counted: {len(rows) - start}; region = last {last_n} force launches onward\n")
        f.write(f"{'kernel':62s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} vgpr lds\n")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k:62s} {a[0]:7d} {a[1] / 1e3:12.1f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} "
                    f"{100.0 * a[1] / tot:6.2f} {a[4]} {a[5]}\n")
        f.write(f"# total kernel time {tot / 1e3:.1f} us\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
