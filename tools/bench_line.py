#!/usr/bin/env python3
"""Condense bench.py's JSON line: value, ms/step, arena size (stdin -> one short line)."""
import json
import sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
other, found = [], False
for l in sys.stdin:
    if l.startswith("{"):
        j = json.loads(l)
        found = True
        print(tag, j["value"], "GCUPS", j["ms_per_step"], "ms", j["config"]["trace_arena_gb"], "GB", j["roofline"]["frac"])
    else:
        other.append(l.rstrip())
if not found:
    print(tag, "NO RESULT:", " | ".join(other[-6:]))
