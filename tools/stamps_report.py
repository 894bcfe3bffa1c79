#!/usr/bin/env python3
"""Report of a DEME_TILE_STAMPS dump (16 words per tile: 100 MHz wall clock at the phase boundaries, see deme_tile.h).  usage: stamps_report.py file"""
import sys
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
t0 = d[:, 0].astype(np.int64)
ok = t0 > 0
base = t0[ok].min()
st = (d[ok, :12].astype(np.int64) - base) * 10e-3
st[d[ok, :12] == 0] = np.nan
nct = d[ok, 13].astype(int)
nh = d[ok, 14].astype(int)
end = st[:, 11]
life = end - st[:, 0]
print(f"tiles {ok.sum()} of {len(d)}; kernel span {np.nanmax(end):.1f} us; tile life mean {life.mean():.2f} median {np.median(life):.2f} p10 {np.percentile(life,10):.2f} p90 {np.percentile(life,90):.2f} us")
print(f"contacts per tile mean {nct.mean():.0f} max {nct.max()}; foreign owners mean {nh.mean():.0f} max {nh.max()}")
nr = (nct + 255) // 256
for r in range(1, 6):
    m = nr == r
    if m.sum() == 0:
        continue
    rounds = [round(float(np.nanmean(st[m, 3 + k] - st[m, 2 + k])), 2) for k in range(r)]
    print(f"  {r} round(s): {m.sum():5d} tiles, life {life[m].mean():.2f}, start -> staged {np.nanmean(st[m,2]-st[m,0]):.2f}, rounds {rounds}, epilogue {np.nanmean(end[m]-st[m,2+r]):.2f} us")
ev = np.concatenate([np.stack([st[:, 0], np.ones(len(st))], 1), np.stack([end, -np.ones(len(st))], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
row = []
for T in list(range(0, int(np.nanmax(end)) + 5, 5)):
    i = np.searchsorted(ev[:, 0], T)
    if 0 < i < len(conc):
        row.append(f"{T}:{int(conc[i-1])}")
print("tiles in process at t [us]:", " ".join(row))
clk = d[ok, 15].astype(np.float64)
if (clk > 0).any():
    m = (clk > 0) & (life > 1.0)
    print(f"shader clock over the tiles' lives (s_memtime cycles / wall time): median {np.median(clk[m] / life[m]) / 1e3:.3f} GHz, p10 {np.percentile(clk[m] / life[m], 10) / 1e3:.3f}, p90 {np.percentile(clk[m] / life[m], 90) / 1e3:.3f}")
hw = d[ok, 12]
xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(int)
cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int); sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(int)
wave = (hw & np.uint64(0xF)).astype(int); simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(int)
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
# gaps between consecutive tiles on the same (CU, wave slot): how long a slot stands empty / a persistent workgroup needs between tiles
slot = key * 64 + simd * 16 + wave
gaps = []
for s in np.unique(slot):
    m = np.where(slot == s)[0]
    o = m[np.argsort(st[m, 0])]
    g = st[o[1:], 0] - end[o[:-1]]
    gaps.append(g[(g > -0.5) & (g < 20)])
gaps = np.concatenate(gaps) if gaps else np.array([0.0])
print(f"gap between a tile's end and the next tile's start on the same wave slot: mean {gaps.mean():.2f} median {np.median(gaps):.2f} p90 {np.percentile(gaps,90):.2f} us ({len(gaps)} pairs)")
u, c = np.unique(key, return_counts=True)
print(f"distinct CUs {len(u)}; tiles per CU min {c.min()} mean {c.mean():.1f} max {c.max()}; per XCD {np.bincount(xcc).tolist()}")
