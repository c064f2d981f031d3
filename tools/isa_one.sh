#!/bin/bash
# developer tool: compile ONE instantiation (default 32,3) to ISA and print the instruction mix
G=${1:-32}; S=${2:-3}
cd /root/repo/agatha_amd/csrc
mkdir -p /tmp/isa
python3 - "$G" "$S" <<'PY'
import re,sys
G,S=sys.argv[1],sys.argv[2]
s=open('align_kernel.hip').read()
a=s.index('static const Cfg kCfgs[] = {'); b=s.index('};',a)
s=s[:a]+'static const Cfg kCfgs[] = {\n    {%s, %s, launch_align_t<%s, %s>},\n'%(G,S,G,S)+s[b:]
open('/tmp/isa/one.hip','w').write(s)
PY
cp kernels.h device_common.h /tmp/isa/
FLAGS=$(grep '^CXXFLAGS' Makefile | sed 's/CXXFLAGS *= *//; s/\$(ARCH)/gfx950/')
hipcc $FLAGS -S --cuda-device-only -o /tmp/isa/one.s /tmp/isa/one.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "align_kernel" | grep -E "error|VGPRs:|SGPRs:|Spill|Occupancy|Scratch"
python3 isa_stats.py /tmp/isa/one.s $G,$S
python3 - <<'PY'
import re,collections
s=open('/tmp/isa/one.s').read()
i=s.index('_ZN6agatha12align_kernel'); fn=s[i:]; fn=fn[:fn.index('.section')]
blocks=re.split(r'\n(?=\.LBB\d+_\d+:|; %bb\.\d+:)', fn)
c=collections.Counter()
for b in blocks:
    if 'in Loop' not in b.split('\n')[0]: continue
    for l in b.split('\n')[1:]:
        t=l.strip()
        if l.startswith('\t') and t and not t.startswith('.') and not t.startswith(';'): c[t.split()[0]]+=1
print("IN-LOOP total",sum(c.values()),"valu",sum(v for k,v in c.items() if k.startswith('v_')),"salu",sum(v for k,v in c.items() if k.startswith('s_')))
PY
