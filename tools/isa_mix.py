#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S --cuda-device-only listing.
usage: isa_mix.py file.s <substring of the mangled kernel name>"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(_Z\w*' + re.escape(key) + r'\w*):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M)
if not m:
    sys.exit("kernel not found")
print(m.group(1))
body = m.group(2)
blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
cur = 'entry'
for i, b in enumerate(blocks):
    if i % 2 == 1:
        cur = b
        continue
    ins = [l.strip().split()[0] for l in b.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    c = collections.Counter(ins)
    nm = sum(v for k, v in c.items() if 'mfma' in k)
    if nm == 0:
        continue
    valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
    print(cur, 'mfma', nm, 'valu', valu, 'ds', sum(v for k, v in c.items() if k.startswith('ds_')),
          'global', sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'scratch_'))),
          'salu', sum(v for k, v in c.items() if k.startswith('s_')), 'total', len(ins))
    print('    ', [(k, v) for k, v in c.most_common(16) if 'mfma' not in k])
