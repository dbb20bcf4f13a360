#!/usr/bin/env python3
"""Per-kernel census of memory instructions and s_waitcnt vmcnt values in a hipcc -S listing (developer tool):
   scripts/asm_census.py file.s [name-substring].  Many vmcnt(0) next to in-loop loads = the prefetch is not a prefetch."""
import re, sys, collections
s = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):.*\n', s, re.M):
    name = m.group(1)
    if sub not in name: continue
    a = m.end(); b = s.index('.Lfunc_end', a)
    w = collections.Counter(); nl = ns = br = 0
    for l in s[a:b].split('\n'):
        t = l.strip()
        if t.startswith(('global_load', 'buffer_load')): nl += 1
        elif t.startswith(('global_store', 'buffer_store')): ns += 1
        elif t.startswith('s_cbranch'): br += 1
        mm = re.match(r's_waitcnt.*vmcnt\((\d+)\)', t)
        if mm: w[int(mm.group(1))] += 1
    print(name[:64].ljust(64), 'loads', nl, 'stores', ns, 'branches', br, 'vmcnt', dict(sorted(w.items())))
