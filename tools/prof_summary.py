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
    ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kcol = next((x for x in ccols if x in ("kernel_name", "name", "kernel")), None)
    if kcol and "counter_name" in ccols and "value" in ccols and "dispatch_id" in ccols:
        # counters are stored per dimension instance (XCD / SE / ...): sum them per dispatch, then average over dispatches
        agg = c.execute(f"select {kcol}, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by {kcol}, counter_name").fetchall()
        if agg:
            lines += ["", "## counter totals (summed over all hardware instances; per dispatch = total / dispatches)", "",
                      "| kernel | counter | dispatches | total | per dispatch |", "|---|---|---|---|---|"]
            for k, cn, nd, tot in agg:
                lines.append(f"| `{k}` | {cn} | {nd} | {tot:.0f} | {tot / max(nd, 1):.0f} |")
    rows = c.execute("select * from counters_collection limit 2000").fetchall()
    if rows:
        lines += ["", "## counters (view counters_collection)", "", "| " + " | ".join(ccols) + " |", "|" + "---|" * len(ccols)]
        for row in rows[:40]:
            lines.append("| " + " | ".join(str(x) for x in row) + " |")
except Exception as e:
    lines.append(f"(no counters: {e})")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:30]))
