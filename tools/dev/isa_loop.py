#!/usr/bin/env python3
"""Instruction mix of the loop around a kernel's largest basic block (k_multi / k_small: the loop of steps) + the kernels' spill counts:
   hipcc ... -S --cuda-device-only -o /tmp/k.s ba_kernels.hip;  python3 tools/dev/isa_loop.py /tmp/k.s _ZN2ba7k_multiILi8ELi1ELb1ELb1E"""
import re, sys
from collections import Counter
t = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else '_ZN2ba7k_multiILi8ELi1ELb1ELb1E'
s0 = next(i for i, l in enumerate(t) if l.startswith(pat))
e0 = next(i for i in range(s0, len(t)) if t[i].startswith('.Lfunc_end'))
L = t[s0:e0]
blocks = []; cur = None
for i, l in enumerate(L):
    m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
    if m:
        cur = [m.group(1), i, 0, m.group(2)]; blocks.append(cur)
    elif cur and l.strip().startswith('v_'):
        cur[2] += 1
big = max(blocks, key=lambda b: b[2])
hdr = re.search(r'Header=(BB\d+_\d+)', big[3]).group(1)
start = next(i for i, l in enumerate(L) if re.match(r'^\.L%s:' % hdr, l))
c = Counter(); ops = Counter()
for i in range(start, len(L)):
    l = L[i]
    m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
    if m and ('Header=%s' % hdr) not in m.group(2) and i > start:
        break
    t2 = l.strip()
    if not t2 or t2.startswith(';') or t2.startswith('.'):
        continue
    op = t2.split()[0]; c[op.split('_')[0]] += 1
    if op.startswith('v_'):
        ops[op] += 1
print('largest block', big[:3], 'loop header', hdr)
print(dict(c)); print(ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30))
y = '\n'.join(t); y = y[y.index('amdhsa.kernels:'):]
for m in re.finditer(r'\.name:\s+(\S+).*?(?=\n  - \.agpr|\Z)', y, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    if 'k_multi' in m.group(1) or 'k_small' in m.group(1):
        print(m.group(1)[8:40], 'vgpr_spill', g('vgpr_spill_count'), 'sgpr_spill', g('sgpr_spill_count'), 'scratch', g('private_segment_fixed_size'))
