#!/usr/bin/env python3
"""Approximate VGPR liveness over the kernel ISA (straight-line view): for every stage segment (split at s_memtime)
lists how many VGPRs / AGPRs are live THROUGH it (last touched before, next read after, untouched inside) and how
many it touches itself.  Usage: asm_liveness.py file.s [segment-to-detail]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
ins = []   # (lineno, op, defs:set, uses:set, seg)
seg = 0
rx = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')
def regs(tok):
    out = []
    for m in rx.finditer(tok):
        if m.group(1): out.append((m.group(1), int(m.group(2))))
        else: out += [(m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1)]
    return out
for ln, l in enumerate(lines):
    t = l.split(';')[0].strip()
    if not t or t.startswith('.') or t.endswith(':'): continue
    parts = t.split(None, 1)
    op = parts[0]
    if op == 's_memtime': seg += 1; continue
    ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
    nodef = op.startswith(('ds_write', 'global_store', 'scratch_store', 'buffer_store', 'flat_store', 'v_cmp', 'v_readlane', 'v_readfirstlane', 's_', 'ds_bpermute_b32x'))
    d, u = set(), set()
    for i, o in enumerate(ops):
        r = regs(o)
        if i == 0 and not nodef:
            d.update(r)
            if op.startswith(('v_fmac', 'v_writelane', 'v_mac', 'v_accvgpr_write')) and not op.startswith('v_accvgpr_write'): u.update(r)
        else: u.update(r)
    ins.append((ln, op, d, u, seg))
nseg = seg + 1
first = {}; last = {}
for idx, (ln, op, d, u, sg) in enumerate(ins):
    first.setdefault(sg, idx); last[sg] = idx
detail = int(sys.argv[2]) if len(sys.argv) > 2 else -1
print("seg  through(v) through(a) touched(v)")
for sg in range(nseg):
    if sg not in first: continue
    a, b = first[sg], last[sg]
    touched = set()
    for k in range(a, b + 1): touched |= ins[k][2] | ins[k][3]
    through = []
    allregs = set()
    for r in [('v', i) for i in range(256)] + [('a', i) for i in range(256)]:
        if r in touched: continue
        # next access after b
        nxt = None
        for k in range(b + 1, min(len(ins), b + 40000)):
            if r in ins[k][3]: nxt = 'use'; break
            if r in ins[k][2]: nxt = 'def'; break
        if nxt != 'use': continue
        prv = None
        for k in range(a - 1, -1, -1):
            if r in ins[k][2] or r in ins[k][3]: prv = k; break
        if prv is None: continue
        through.append((r, prv))
    tv = [x for x in through if x[0][0] == 'v']; ta = [x for x in through if x[0][0] == 'a']
    print("%3d  %9d %10d %10d" % (sg, len(tv), len(ta), len([r for r in touched if r[0] == 'v'])))
    if sg == detail:
        for r, prv in sorted(tv, key=lambda x: x[1]):
            print("   %s%d last touched line %d: %s" % (r[0], r[1], ins[prv][0] + 1, lines[ins[prv][0]].strip()[:90]))
