#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel over the timed region of bench.py.

    python profiles/summarize_pmc.py <counter_collection.csv> <out.txt> <last_n_force_launches>
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("deme_dev::", "")
    if name.startswith("rocprim"):
        m = re.search(r"detail::(?:trampoline_kernel<.*?detail::)?([a-z_]+)", name)
        return "rocprim:" + (m.group(1)[:40] if m else "?")
    return name[:50]


def main():
    path, out, last_n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rows = list(csv.DictReader(open(path)))
    # one row per (dispatch, counter)
    disp = defaultdict(dict)
    names = {}
    for r in rows:
        d = int(r["Dispatch_Id"])
        disp[d][r["Counter_Name"]] = float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
    ids = sorted(disp)
    f_ids = [d for d in ids if ("k_calc_forces" in names[d] or "k_forces_fast" in names[d] or "k_tile_forces" in names[d] or "k_tile_step" in names[d] or "deme_custom_forces" in names[d])]
    start = f_ids[-last_n] if len(f_ids) >= last_n else ids[0]
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for d in ids:
        if d < start:
            continue
        k = short(names[d])
        cnt[k] += 1
        for c, v in disp[d].items():
            agg[k][c] += v
    counters = sorted({c for k in agg for c in agg[k]})
    with open(out, "w") as f:
        f.write(f"# source {path}; per-dispatch averages over the region from the last {last_n} force launches\n")
        f.write(f"{'kernel':52s} {'calls':>6s} " + " ".join(f"{c:>18s}" for c in counters) + "\n")
        for k in sorted(agg, key=lambda k: -cnt[k]):
            f.write(f"{k:52s} {cnt[k]:6d} " + " ".join(f"{agg[k][c] / cnt[k]:18.1f}" for c in counters) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
