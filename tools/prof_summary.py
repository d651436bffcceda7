#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace / PMC) into a small text file for profiles/."""
import sqlite3, sys

db, out = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else db
c = sqlite3.connect(db)
lines = [f"# {title}", "", "## kernel summary (rocprofv3 --kernel-trace --stats; view top_kernels; durations in ns)", "",
         "| kernel | calls | total_ns | avg_ns | % |", "|---|---|---|---|---|"]
for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
    lines.append(f"| `{name}` | {calls} | {total * 1e3:.0f} | {avg * 1e3:.0f} | {pct:.3f} |")
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
want = [x for x in ("name", "grid_size", "workgroup_size", "lds_block_size", "scratch_size", "vgpr_count", "accum_vgpr_count", "sgpr_count", "start", "end", "duration") if x in cols]
if want:
    lines += ["", "## dispatches (view kernels)", "", "| " + " | ".join(want) + " |", "|" + "---|" * len(want)]
    for row in c.execute(f"select {', '.join(want)} from kernels order by start limit 40"):
        lines.append("| " + " | ".join(str(x) for x in row) + " |")
try:
    rows = c.execute("select * from counters_collection limit 2000").fetchall()
    if rows:
        ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        lines += ["", "## counters (view counters_collection)", "", "| " + " | ".join(ccols) + " |", "|" + "---|" * len(ccols)]
        for row in rows[:200]:
            lines.append("| " + " | ".join(str(x) for x in row) + " |")
except Exception as e:
    lines.append(f"(no counters: {e})")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:30]))
