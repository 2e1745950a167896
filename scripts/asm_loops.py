#!/usr/bin/env python3
"""List the loops (backward branches) of one kernel in hipcc's .s output with their instruction counts."""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
show = len(sys.argv) > 3
s = open(path).read()
m = re.search(r'^(_Z\w*%s\w*):.*\n' % pat, s, re.M)
name = m.group(1)
body = s[m.end():s.index('.Lfunc_end', m.end())]
lines = [l.split(';')[0].strip() for l in body.split('\n')]
ins = []; labels = {}
for l in lines:
    if not l or l.startswith((';', '//')): continue
    if l.endswith(':'):
        labels[l[:-1]] = len(ins); continue
    if l.startswith('.'): continue
    ins.append(l.split(';')[0].strip())
print(name, 'instructions:', len(ins))
loops = []
for i, x in enumerate(ins):
    mm = re.match(r's_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)', x)
    if mm:
        t = mm.group(1) or mm.group(2)
        if t in labels and labels[t] <= i:
            loops.append((labels[t], i, t))
for a, b, t in sorted(loops, key=lambda z: z[1] - z[0]):
    c = collections.Counter(y.split()[0] for y in ins[a:b + 1])
    cls = collections.Counter()
    for k, v in c.items():
        cls['valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else 'lds' if k.startswith('ds_') else 'vmem' if k.startswith(('buffer_', 'global_', 'flat_', 'scratch_')) else 'other'] += v
    print(f'loop {t}: {b - a + 1} instrs', dict(cls), c.most_common(8))
    if show and t == sys.argv[3]:
        print('\n'.join(ins[a:b + 1]))
