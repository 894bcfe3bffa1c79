#!/usr/bin/env python3
"""Live VGPRs per instruction of one kernel's gfx950 assembly (hipcc --save-temps, the function's text cut out by awk):
backward liveness over the block graph; prints the peak, where it is, and the pressure at every label.
usage: vgpr_live.py k.s [N]   (N: list the N highest-pressure lines)"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
topn = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ins = []  # (lineno, op, args, label_before)
labels = {}
for i, l in enumerate(src):
    t = l.split(';')[0].rstrip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    t = t.strip()
    if not t or t.startswith('.') or t.endswith(':') or t.startswith(';;'):
        continue
    parts = t.split(None, 1)
    ins.append((i + 1, parts[0], parts[1] if len(parts) > 1 else ''))
def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out
n = len(ins)
DEF = [set() for _ in range(n)]
USE = [set() for _ in range(n)]
succ = [[] for _ in range(n)]
for k, (ln, op, args) in enumerate(ins):
    a = [x.strip() for x in args.split(',')] if args else []
    nodst = op.startswith(('global_store', 'ds_write', 'scratch_store', 'buffer_store', 'v_cmp_', 'v_cmpx', 's_', 'ds_append')) or \
        (op.startswith('global_atomic') and 'sc0' not in args) or op.startswith(('v_readlane', 'v_readfirstlane'))
    if op.startswith(('v_readlane', 'v_readfirstlane')):
        for x in a[1:]:
            USE[k].update(regs(x))
    elif nodst:
        for x in a:
            USE[k].update(regs(x))
    elif a:
        DEF[k].update(regs(a[0]))
        for x in a[1:]:
            USE[k].update(regs(x))
        if op.startswith(('v_writelane', 'v_mac', 'v_fmac', 'v_pk_fmac', 'v_dot2c')) or op.endswith('_sdwa'):
            USE[k].update(regs(a[0]))  # read-modify-write
    if op == 's_endpgm':
        continue
    if op == 's_branch':
        succ[k].append(labels[a[0]])
        continue
    if op.startswith('s_cbranch'):
        succ[k].append(labels[a[0]])
    if k + 1 < n:
        succ[k].append(k + 1)
live_in = [set() for _ in range(n)]
changed = True
while changed:
    changed = False
    for k in range(n - 1, -1, -1):
        out = set()
        for s in succ[k]:
            out |= live_in[s]
        # (predicated writes do not kill: treat every VALU def under exec as a kill anyway -- an upper bound would need exec tracking)
        new = USE[k] | (out - DEF[k])
        if new != live_in[k]:
            live_in[k] = new
            changed = True
press = [len(live_in[k] | DEF[k]) for k in range(n)]
order = sorted(range(n), key=lambda k: -press[k])
print('instructions', n, 'peak live VGPRs', press[order[0]])
shown = 0
last = -100
for k in order:
    if abs(k - last) < 40:
        continue
    print(f'  {press[k]:4d} live at asm line {ins[k][0]}: {ins[k][1]} {ins[k][2][:60]}')
    last = k
    shown += 1
    if shown >= topn:
        break
inv = {v: k for k, v in labels.items()}
for lab, k in sorted(labels.items(), key=lambda kv: kv[1]):
    if k < n:
        print(f'  {lab:12s} line {ins[k][0]:5d} live-in {len(live_in[k])}')
if len(sys.argv) > 3:  # vgpr_live.py k.s N detail: where the registers live at the peak were last written (linear order)
    k0 = order[0]
    rows = []
    for r in sorted(live_in[k0] | DEF[k0]):
        d = max((k for k in range(k0) if r in DEF[k]), default=None)
        rows.append((ins[d][0] if d is not None else 0, r, (ins[d][1] + ' ' + ins[d][2][:50]) if d is not None else 'entry'))
    for ln, r, t in sorted(rows):
        print(f'    v{r:<3d} last written at line {ln}: {t}')
