#!/usr/bin/env python3
"""Instruction-class breakdown per basic block of one kernel in a hipcc -S listing.
usage: isa_blocks.py listing.s kernel-name-substring [min_instrs]"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 10
text = open(src).read().split('\n')
start = next(i for i, l in enumerate(text) if l.startswith('_Z') and key in l and l.split()[0].endswith(':'))
end = next(i for i in range(start, len(text)) if 's_endpgm' in text[i])
lines = text[start:end + 1]
blocks = []; cur = ['entry', []]
for l in lines:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur); cur = [m.group(1), []]
    elif l.startswith('\t') and not l.startswith('\t;') and not l.startswith('\t.') and l.strip():
        cur[1].append(l.strip())
blocks.append(cur)
def cls(i):
    op = i.split()[0]
    if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): return 'lane'
    if op.startswith('ds_'): return 'ds'
    if op.startswith(('global_', 'scratch_', 'buffer_')): return 'vmem'
    if op.startswith('s_load'): return 'smem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith(('s_cbranch', 's_branch')): return 'branch'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_') and 'f64' in op: return 'v64'
    if op.startswith(('v_fma_f32', 'v_fmac_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_pk_')): return 'vf32'
    if op.startswith('v_cndmask'): return 'cnd'
    if op.startswith('v_cvt'): return 'cvt'
    if op.startswith('v_'): return 'vint'
    return 'other'
tot = {}
for name, ins in blocks:
    c = {}
    for i in ins:
        k = cls(i); c[k] = c.get(k, 0) + 1; tot[k] = tot.get(k, 0) + 1
    if len(ins) >= minlen:
        print(f'{name:12s} {len(ins):5d}', ' '.join(f'{k}={v}' for k, v in sorted(c.items())))
print('TOTAL', sum(tot.values()), ' '.join(f'{k}={v}' for k, v in sorted(tot.items())))
