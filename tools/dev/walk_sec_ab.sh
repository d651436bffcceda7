#!/bin/bash
# round 6: the walk kernels' knobs on the 32..256 configurations with traceback, same box (variants of the k0_p1 object: tools/dev/variant.sh w0 "-DBA_TB_FAST=0" 0 1 ...)
cd "$(dirname "$0")/../.."
for r in 1 2; do for c in "c2t 200000" "c4t 400000" "c5 80000"; do for l in "" libblock_aligner_hip_w0.so libblock_aligner_hip_w4.so libblock_aligner_hip_wfc2.so; do
  [ -n "$l" ] && [ ! -f block_aligner_amd/lib/$l ] && continue
  echo -n "[$c ${l:-main}] "; BA_LIB=$l python tools/dev/sec.py $c 2>&1 | tail -1
done; done; done
