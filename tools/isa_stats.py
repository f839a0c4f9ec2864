#!/usr/bin/env python3
"""Instruction census of one kernel in a hipcc -S dump: isa_stats.py file.s <mangled-name-substring>"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i] and all('s_endpgm' not in x for x in s[i+1:i+1]))
# last s_endpgm before .section/.rodata of this function
for i in range(start, len(s)):
    if s[i].strip().startswith('.Lfunc_end'):
        end = i; break
ops = []
for l in s[start+1:end]:
    t = l.strip()
    if not t or t[0] in '.;/' or t.endswith(':'): continue
    ops.append(t.split()[0])
c = Counter(ops)
print(len(ops), 'instructions')
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    if any(k.startswith(p) for p in ('s_load', 'global_', 'buffer_', 'ds_', 'scratch_', 'v_mul_f64', 'v_add_f64', 'v_fma_f64', 'v_mfma', 's_waitcnt', 's_barrier', 'v_cndmask', 's_cbranch')):
        print('  %-28s %d' % (k, v))
first = {}
for i, o in enumerate(ops): first.setdefault(o, i)
print('first:', {k: first[k] for k in first if k.startswith(('s_load', 'global_load', 'global_store'))})
