#!/bin/bash
# round 6: wave priorities of k_multi's solo mode (MQ_SOLO_PRIO, default 2) and traceback waves (MQ_TB_PRIO, default 1) with the cheaper lane walk, same box
cd "$(dirname "$0")/../.."
for r in 1 2; do for l in "" libblock_aligner_hip_sp1.so libblock_aligner_hip_sp3.so libblock_aligner_hip_tp0.so libblock_aligner_hip_tp2.so; do
  echo -n "[c3 100k ${l:-main}] "; BA_LIB=$l timeout 120 python tools/dev/c3.py 100000 2>&1 | tail -1 | cut -c1-120; done; done
for l in "" libblock_aligner_hip_tp0.so libblock_aligner_hip_tp2.so; do echo -n "[c3 12.5k ${l:-main}] "; BA_LIB=$l timeout 120 python tools/dev/c3.py 12500 2>&1 | tail -1 | cut -c1-120; done
