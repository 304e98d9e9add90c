#!/usr/bin/env python3
"""Static instruction mix of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only).
usage: isa_mix.py file.s [name-filter]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter()
    for i in ins:
        if i.startswith('v_pk'): c['v_pk'] += 1
        elif i.startswith(('v_rcp', 'v_rsq', 'v_sqrt', 'v_log', 'v_exp')): c['trans'] += 1
        elif i.startswith(('v_cndmask', 'v_mov')): c['mov/sel'] += 1
        elif i.startswith('v_cmp'): c['cmp'] += 1
        elif '_f64' in i and i.startswith('v_'): c['v_f64'] += 1
        elif '_f32' in i and i.startswith('v_'): c['v_f32'] += 1
        elif i.startswith('v_'): c['v_int'] += 1
        elif i.startswith('s_waitcnt') or i.startswith('s_nop'): c['wait'] += 1
        elif i.startswith('s_'): c['salu'] += 1
        elif i.startswith('ds_'): c['lds'] += 1
        elif i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['vmem'] += 1
        else: c['other'] += 1
    print(name[-60:], len(ins), dict(sorted(c.items())))
