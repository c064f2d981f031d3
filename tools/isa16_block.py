"""Developer tool: print one basic block (or its instruction histogram) of the ISA listing tools/isa16_one.sh leaves in /tmp/isa16/one.s
    python tools/isa16_block.py .LBB0_609 [full]"""
import re, collections, sys
out, skip = [], False
for l in open('/tmp/isa16/one.s').read().split('\n'):
    t = l.strip()
    if t.startswith('.if '): skip = not eval(t[4:]); continue
    if t == '.endif': skip = False; continue
    if not skip: out.append(l)
s = '\n'.join(out)
i = s.index('_ZN6agatha14align16_kernel'); i = s.index(':\n', i)
fn = s[i:]; fn = fn[:fn.index('.Lfunc_end')]
for b in re.split(r'\n(?=\.LBB\d+_\d+:)', fn):
    lines = b.split('\n')
    if lines[0].split(':')[0] != sys.argv[1]: continue
    body = [l.strip() for l in lines[1:] if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    if len(sys.argv) > 2: print('\n'.join(body))
    else:
        c = collections.Counter(x.split()[0] for x in body)
        for k, v in c.most_common(): print(v, k)
