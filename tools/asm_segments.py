#!/usr/bin/env python3
"""Splits the kernel ISA (tools/kernel_resources.sh with KEEP=file) at the s_memtime stage stamps and prints, per
segment, the instruction count and the spill traffic (scratch loads / stores, AGPR copies, SGPR lane spills)."""
import collections
import sys
lines = open(sys.argv[1]).read().split('\n')
segs = []
cur = collections.Counter()
for l in lines:
    t = l.strip()
    if not t or t.startswith(('.', ';')) or t.endswith(':'):
        continue
    op = t.split()[0]
    if op == 's_memtime':
        segs.append(cur)
        cur = collections.Counter()
        continue
    cur[op] += 1
segs.append(cur)
print("seg instrs  sld  sst  acc   wl   rl  f64   ds  glob  nop wait")
for i, c in enumerate(segs):
    n = sum(c.values())
    pre = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print("%3d %6d %4d %4d %4d %4d %4d %4d %4d %5d %4d %4d" % (
        i, n, pre('scratch_load'), pre('scratch_store'), c['v_accvgpr_read_b32'] + c['v_accvgpr_write_b32'],
        c['v_writelane_b32'], c['v_readlane_b32'], sum(v for k, v in c.items() if 'f64' in k), pre('ds_'),
        pre('global_'), c['s_nop'], c['s_waitcnt']))
