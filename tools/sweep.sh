#!/bin/bash
# usage: tools/sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...   -> one condensed bench line per environment set
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 ${BENCH_ARGS} 2>&1 | python3 tools/bench_line.py "[$cfg]" | tee -a gpurun_out/sweep.log
done
