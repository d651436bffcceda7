#!/bin/bash
# Instruction counts per configuration: tools/pmc_cases.sh "<trace> <min> <max>" ...   (20000 pairs, 1 launch each)
cd "$(dirname "$0")/.."
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg
  name=t$1_$2_$3
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d /tmp/pmc_$name -o p -- python3 tools/gpu_perf_one.py 20000 $1 1 $2 $3 > $out/pmc_$name.log 2>&1
  db=$(find /tmp/pmc_$name -name '*.db' | head -1)
  python3 tools/prof_summary.py "$db" $out/pmc_$name.md "$cfg" > /dev/null
  echo "== $cfg: $(grep '^pairs=' $out/pmc_$name.log)"
  grep "k_align" $out/pmc_$name.md | grep "SQ_" | sed 's/.*| \(SQ_[A-Z_]*\) | [0-9]* | \([0-9]*\) |.*/   \1 \2/'
done 2>&1 | tee $out/pmc_cases.txt
