#!/bin/bash
# developer tool: compile ONE instantiation of the packed-int16 kernel (default 16,3,-1) to ISA and print the instruction
# counts of its larger basic blocks (the assembler's .if / .endif around the cut caps are evaluated)
G=${1:-16}; P=${2:-3}; T0=${3:--1}
cd "$(dirname "$0")/../agatha_amd/csrc"
mkdir -p /tmp/isa16
FLAGS=$(grep '^CXXFLAGS' Makefile | sed 's/CXXFLAGS *= *//; s/\$(ARCH)/gfx950/')
hipcc $FLAGS -O3 -DAGATHA16_NT0=$((-T0)) -DAGATHA16_ONLY_G=$G -DAGATHA16_ONLY_P=$P -S --cuda-device-only -o /tmp/isa16/one.s align16_inst.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs:|SGPRs:|Spill|Occupancy|Scratch"
python3 - <<'PY'
import re, collections
out, skip = [], False
for l in open('/tmp/isa16/one.s').read().split('\n'):
    t = l.strip()
    if t.startswith('.if '): skip = not eval(t[4:]); continue
    if t == '.endif': skip = False; continue
    if not skip: out.append(l)
s = '\n'.join(out)
i = s.index('_ZN6agatha14align16_kernel'); i = s.index(':\n', i)
fn = s[i:]; fn = fn[:fn.index('.Lfunc_end')]
tot = collections.Counter()
for b in re.split(r'\n(?=\.LBB\d+_\d+:)', fn):
    lines = b.split('\n')
    body = [l.strip() for l in lines[1:] if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(x.split()[0] for x in body)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    if len(body) >= 40:
        print(lines[0].split(':')[0][:10].ljust(10), 'instructions', len(body), 'valu', valu, 'lds', sum(v for k, v in c.items() if k.startswith('ds_')),
              'salu', sum(v for k, v in c.items() if k.startswith('s_')), 'scratch', sum(v for k, v in c.items() if 'scratch' in k))
PY
