#!/usr/bin/env python3
"""Coarse map of a kernel's ISA (hipcc -S --cuda-device-only): per block of lines the count of MFMAs, scratch, LDS, global, AGPR moves.
Usage: isa_map.py file.s kernel-name-substring [lines-per-block]"""
import re, sys
s = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r'^(\S+):\s*; @', s, re.M) if sys.argv[2] in m.group(1)]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 150
for name in names[:1]:
    i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
    body = s[i:j].split('\n')
    print(name, len(body), 'lines')
    for k in range(0, len(body), step):
        blk = body[k:k + step]
        c = lambda p: sum(1 for x in blk if re.search(p, x))
        print(k, 'mfma', c('v_mfma'), 'scr_ld', c('scratch_load'), 'scr_st', c('scratch_store'), 'ds', c(r'\bds_'), 'glob', c('global_load'),
              'gst', c('global_store|global_atomic'), 'bar', c('s_barrier'), 'accrd', c('v_accvgpr_read'), 'accwr', c('v_accvgpr_write'), 'nop', c('s_nop'),
              [x.split(':')[0] for x in blk if x.startswith('.LBB')][:5])
