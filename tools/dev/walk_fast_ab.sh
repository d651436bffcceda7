#!/bin/bash
# round 6: the lanes' walk with runs of diagonal moves at once (tb_diag) against the round-5 form and its knobs, same box.
#   variants: tools/dev/variant.sh fast0 "-DBA_TB_FAST=0" 1 8; fc2 "-DBA_TB_FCELLS=2"; r3 "-DBA_TB_DIAG_RUNS=3"; nd2 "-DBA_TB_DIAG2=0 -DBA_TB_FCELLS=2"; w0 "-DBA_TB_FAST=0" 0 1
cd "$(dirname "$0")/../.."
for r in 1 2; do
for l in "" libblock_aligner_hip_fast0.so libblock_aligner_hip_fc2.so libblock_aligner_hip_nd2.so; do
  [ -n "$l" ] && [ ! -f block_aligner_amd/lib/$l ] && continue
  echo -n "[c3 100k ${l:-main}] "; BA_LIB=$l python tools/dev/c3.py 100000 2>&1 | tail -1
done; done
for n in 12500 25000; do for l in "" libblock_aligner_hip_fast0.so; do echo -n "[c3 $n ${l:-main}] "; BA_LIB=$l python tools/dev/c3.py $n 2>&1 | tail -1; done; done
for c in "c2t 200000" "c4t 400000" "c5 80000"; do for l in "" libblock_aligner_hip_w0.so; do
  echo -n "[$c ${l:-main}] "; BA_LIB=$l python tools/dev/sec.py $c 2>&1 | tail -1
done; done
