#!/usr/bin/env python3
"""Flag kernels whose ISA waits for a vector-memory load right after issuing it, again and again (a compiler that ran out of registers —
or per-element branches — turns a batch of independent loads into dependent round trips: found in round 5 in the persistent stack's
publish phase and tail).  Usage: isa_serial_loads.py file.s [min-count]"""
import re, sys, subprocess
s = open(sys.argv[1]).read()
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for m in re.finditer(r'^(\S+):\s*; @', s, re.M):
    name = m.group(1); i = m.end(); j = s.find('.Lfunc_end', i)
    body = [x.strip() for x in s[i:j].split('\n') if x.strip() and not x.strip().startswith(';')]
    n = 0
    for k, x in enumerate(body):
        if x.startswith(('global_load', 'buffer_load')) and 'lds' not in x:
            for y in body[k + 1:k + 4]:
                if y.startswith('s_waitcnt vmcnt(0)'):
                    n += 1
                    break
                if y.startswith(('global_load', 'buffer_load')):
                    break
    if n >= thr:
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:120]
        print(f'{n:4d} load -> vmcnt(0) pairs  {dem}')
