#!/bin/bash
# same-box sweep of the scheduling switches on config 3: tools/dev/knobs.sh "VAR=val ..." ...
cd "$(dirname "$0")/../.."
for cfg in "$@"; do
  echo -n "[$cfg] "
  env $cfg timeout 300 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 tools/bench_line.py ""
done
