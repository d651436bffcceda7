#!/usr/bin/env python3
"""Estimated VALU issue cycles of one basic block: isa_cost.py <file.s> <label> (e.g. .LBB0_282)
2 cycles for the plain VOP1/VOP2 forms measured at the fast rate (profiles/r02_valu_rate.md), 4 for everything else."""
import re, sys
from collections import Counter
FAST = {"v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_lshrrev_b32_e32", "v_ashrrev_i32_e32", "v_and_b32_e32", "v_or_b32_e32",
        "v_xor_b32_e32", "v_not_b32_e32", "v_mov_b32_e32", "v_add_f32_e32", "v_sub_f32_e32", "v_mul_f32_e32", "v_add_u16_e32",
        "v_sub_u16_e32", "v_max_u16_e32", "v_max_i16_e32", "v_min_i16_e32", "v_min_u16_e32", "v_lshlrev_b16_e32", "v_ashrrev_i16_e32",
        "v_lshrrev_b16_e32", "v_mul_lo_u16_e32"}
lines = open(sys.argv[1]).read().split("\n")
s = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2] + ":"))
e = next(i for i in range(s + 1, len(lines)) if re.match(r"^\.LBB\d+_\d+:", lines[i]))
c = Counter()
cyc = 0
for l in lines[s + 1:e]:
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    op = t.split()[0]
    c[op] += 1
    if op.startswith("v_"):
        # an SGPR or literal-free check is not attempted: SGPR sources make the fast forms slow, count them as slow
        fast = op in FAST and not re.search(r"\bs\d+\b|\bs\[", t.split(None, 1)[1] if " " in t else "")
        cyc += 2 if fast else 4
nv = sum(v for k, v in c.items() if k.startswith("v_"))
print(f"{sys.argv[2]}: VALU {nv}, est. VALU cycles {cyc}, SALU {sum(v for k, v in c.items() if k.startswith('s_') and k not in ('s_nop', 's_waitcnt'))}, s_nop {c['s_nop']}, DS {sum(v for k, v in c.items() if k.startswith('ds_'))}")
if len(sys.argv) > 3:
    for k, v in c.most_common():
        print(f"  {v:4d} {k}")
