#!/usr/bin/env python3
"""Timeline of ONE contact detection out of a rocprofv3 --kernel-trace CSV: every dispatch between the last k_margins launch and the
next force launch, with its start offset, duration and the idle gap in front of it (sizing read-backs show up as gaps).

    python profiles/timeline.py <kernel_trace.csv> <out.txt>
"""
import csv
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from summarize import short  # noqa: E402

path, out = sys.argv[1], sys.argv[2]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marg = [i for i, r in enumerate(rows) if "k_margins" in r[2]]
i0 = marg[-1]
i1 = next(i for i in range(i0, len(rows)) if "k_tile_forces" in rows[i][2] or "k_forces_fast" in rows[i][2] or "k_calc_forces" in rows[i][2] or "custom" in rows[i][2])
t0, prev_end = rows[i0][0], rows[i0][0]
busy = gaps = 0
with open(out, "w") as f:
    f.write(f"# one detection: from k_margins to the next force launch ({i1 - i0} dispatches)\n{'start_us':>9s} {'dur_us':>8s} {'gap_us':>8s}  kernel\n")
    for s, e, n in rows[i0:i1 + 1]:
        gap = max(0, s - prev_end)
        f.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:8.1f}  {short(n)}\n")
        busy += e - s
        gaps += gap
        prev_end = max(prev_end, e)
    f.write(f"# span {(rows[i1][0] - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us (incl. the force launch that ends it), gaps {gaps / 1e3:.1f} us\n")
print(open(out).read())
