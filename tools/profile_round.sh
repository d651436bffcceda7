#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline refers to (run on the GPU box):  tools/profile_round.sh r01
#   1. kernel trace + stats of the default bench command (100k pairs)      -> gpurun_out/<tag>_kernel_trace_stats.md
#   2. separate PMC passes (no tracing domains besides --kernel-trace): FETCH_SIZE, WRITE_SIZE, SQ instruction mix
#                                                                            -> gpurun_out/<tag>_pmc_*.md, <tag>_roofline.json (tools/roofline_from_pmc.py)
# The program after `--` is python3 itself (no env / shell hop), inputs are generated in-process (--gen-workers 1).
tag=${1:-r01}
pairs=${2:-100000}
cd "$(dirname "$0")/.."
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
B="bench.py --pairs $pairs --gen-workers 1 --no-cpu-baseline --no-e2e"
run() {  # name, rocprof args..., -- bench args
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 "$@" > $out/${tag}_${name}.log 2>&1
  local db=$(find /tmp/prof_$name -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py "$db" $out/${tag}_${name}.md "round ${tag#r}: rocprofv3 $* (pairs=$pairs)" > /dev/null
  grep '^{' $out/${tag}_${name}.log | tail -1 > $out/${tag}_${name}.json
}
run kernel_trace_stats --kernel-trace --stats -d /tmp/prof_kernel_trace_stats -o kt -- python3 $B --steps 3 --warmup 1
run pmc_fetch --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_pmc_fetch -o f -- python3 $B --steps 1 --warmup 0
run pmc_write --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_pmc_write -o w -- python3 $B --steps 1 --warmup 0
run pmc_sq --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d /tmp/prof_pmc_sq -o s -- python3 $B --steps 1 --warmup 0
# 3. the side configurations (32..256 blocks: k_quad / k_align / k_walk), kernel trace only            -> gpurun_out/<tag>_sec_<config>.md
for cfg in "c2 200000" "c2t 200000" "c4 400000" "c4t 400000" "c5 80000"; do
  set -- $cfg
  rm -rf /tmp/prof_sec_$1
  BA_GEN_WORKERS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sec_$1 -o kt -- python3 tools/dev/sec.py $1 $2 > $out/${tag}_sec_$1.log 2>&1
  db=$(find /tmp/prof_sec_$1 -name '*.db' | head -1)
  [ -n "$db" ] && python3 tools/prof_summary.py "$db" $out/${tag}_sec_$1.md "round ${tag#r}: rocprofv3 --kernel-trace --stats -- python3 tools/dev/sec.py $1 $2 ($(grep GCUPS $out/${tag}_sec_$1.log | tail -1))" > /dev/null
done
python3 tools/roofline_from_pmc.py $tag $pairs $out/${tag}_roofline.json > /dev/null
ls -la $out
