#!/usr/bin/env python3
"""Instruction-mix summary of the align kernels from a hipcc -S dump (developer tool)."""
import collections
import re
import sys

s = open(sys.argv[1]).read()
funcs = re.split(r'\n(?=_ZN6agatha\w+:)', s)
for fn in funcs:
    m = re.match(r'(_ZN6agatha\d+align_kernel\w*ILi(\d+)ELi(\d+)E\w+):', fn)
    if not m:
        continue
    body = fn[:fn.find('.section')] if '.section' in fn else fn
    ins = []
    for l in body.split('\n'):
        t = l.strip()
        if not l.startswith('\t') or not t or t.startswith('.') or t.startswith(';'):
            continue
        ins.append(t.split()[0])
    c = collections.Counter(ins)
    g = lambda pred: sum(v for k, v in c.items() if pred(k))
    print(m.group(2), m.group(3), "total", len(ins), "valu", g(lambda k: k.startswith('v_')),
          "salu", g(lambda k: k.startswith('s_')), "accvgpr", g(lambda k: 'accvgpr' in k),
          "scratch", g(lambda k: 'scratch' in k), "vlane", g(lambda k: 'lane' in k and k.startswith('v_')),
          "cndmask", g(lambda k: 'cndmask' in k), "vcmp", g(lambda k: k.startswith('v_cmp')),
          "branch", g(lambda k: k.startswith('s_cbranch')), "bperm", g(lambda k: 'bpermute' in k))
    if len(sys.argv) > 2 and sys.argv[2] == m.group(2) + "," + m.group(3):
        print(c.most_common(50))
