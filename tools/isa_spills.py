#!/usr/bin/env python3
"""Where does a kernel spill?  For every scratch slot: the lines of its stores and reloads (hipcc -S output).
Usage: isa_spills.py file.s kernel-name-substring"""
import re, sys, collections
s = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r'^(\S+):\s*; @', s, re.M) if sys.argv[2] in m.group(1)]
name = names[0]
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
st = collections.defaultdict(list); ld = collections.defaultdict(list)
for k, x in enumerate(body):
    m = re.search(r'scratch_(load|store)_dword(x\d)?\s.*?(?:offset:(\d+))?\s*;', x)
    if m:
        off = int(m.group(3) or 0)
        (st if m.group(1) == 'store' else ld)[off].append(k)
labels = [(k, x.split(':')[0]) for k, x in enumerate(body) if x.startswith('.LBB')]
bars = [k for k, x in enumerate(body) if 's_barrier' in x]
print('barriers at', bars)
print('mfma ranges:', end=' ')
mf = [k for k, x in enumerate(body) if 'v_mfma' in x]
rng = []
for k in mf:
    if rng and k - rng[-1][1] < 40: rng[-1][1] = k
    else: rng.append([k, k])
print(rng)
for off in sorted(set(st) | set(ld)):
    print(off, 'st', st[off], 'ld', ld[off])
