#!/usr/bin/env python3
"""VGPR liveness of one kernel from hipcc -S output: live-register count per line (approximate: first operand = def for non-store
instructions), printed as the maximum per block of lines.  Usage: isa_pressure.py file.s kernel-name-substring [lines-per-block]"""
import re, sys
s = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r'^(\S+):\s*; @', s, re.M) if sys.argv[2] in m.group(1)]
name = names[0]
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
step = int(sys.argv[3]) if len(sys.argv) > 3 else 200
def vregs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.append(int(m.group(3)))
    return out
NODEF = ('global_store', 'scratch_store', 'ds_write', 'buffer_store', 'flat_store', 'v_cmp', 'v_accvgpr_write', 's_', 'v_readfirstlane', 'v_readlane', 'global_atomic', 'ds_add', 'v_cmpx')
ins = []   # (defs, uses, label, branch targets, falls through)
labels = {}
for k, x in enumerate(body):
    t = x.strip()
    if x.startswith('.LBB'):
        labels[x.split(':')[0]] = len(ins)
        continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    t = t.split(';')[0].strip()
    op = t.split()[0]
    rest = t[len(op):]
    ops = [o.strip() for o in rest.split(',')] if rest.strip() else []
    defs, uses = [], []
    if op.startswith(NODEF):
        for o in ops: uses += vregs(o)
    else:
        if ops:
            defs = vregs(ops[0])
            for o in ops[1:]: uses += vregs(o)
            if op.startswith('v_mfma') or op.startswith('v_fmac') or op.startswith('v_mac') or 'v_dot2c' in op or op.startswith('v_pk_fmac'): uses += defs
    tgt = None; fall = True
    if op.startswith('s_cbranch'): tgt = ops[0]
    elif op == 's_branch': tgt = ops[0]; fall = False
    elif op == 's_endpgm': fall = False
    ins.append([set(defs), set(uses), k, tgt, fall])
n = len(ins)
succ = [[] for _ in range(n)]
for a in range(n):
    if ins[a][4] and a + 1 < n: succ[a].append(a + 1)
    if ins[a][3] and ins[a][3] in labels and labels[ins[a][3]] < n: succ[a].append(labels[ins[a][3]])
live_in = [set() for _ in range(n)]
changed = True
it = 0
while changed and it < 50:
    changed = False; it += 1
    for a in range(n - 1, -1, -1):
        out = set()
        for b in succ[a]: out |= live_in[b]
        new = (out - ins[a][0]) | ins[a][1]
        if new != live_in[a]: live_in[a] = new; changed = True
print(name, 'instructions', n, 'iterations', it)
cur = 0
for a in range(0, n, step):
    blk = range(a, min(n, a + step))
    mx = max(blk, key=lambda q: len(live_in[q]))
    print(f'lines {ins[a][2]:6d}-{ins[min(n - 1, a + step - 1)][2]:6d}  max live {len(live_in[mx]):3d} at line {ins[mx][2]}')
