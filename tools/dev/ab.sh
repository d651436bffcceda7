#!/bin/bash
# Same-box A/B of two builds of the library (box-to-box differences are ~1 %, larger than most kernel micro-changes):
#   make -C block_aligner_amd/csrc OBJ=_build_b LIB=../lib/libblock_aligner_hip_b.so      (variant B from the working tree)
#   tools/dev/ab.sh [rounds] [extra library names ...]   -> alternating bench lines for A (libblock_aligner_hip.so), B and the extras
cd "$(dirname "$0")/../.."
rounds=${1:-2}; shift
libs="libblock_aligner_hip.so libblock_aligner_hip_b.so $*"
run() {
  python3 - "$1" <<'PY' 2>&1 | python3 tools/bench_line.py "[$1]"
import sys, os, runpy
sys.path.insert(0, os.getcwd())
import block_aligner_amd.hip as H
H.LIB_PATH = os.path.join(os.path.dirname(H.LIB_PATH), sys.argv[1])
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "2", "--warmup", "1"]
runpy.run_path("bench.py", run_name="__main__")
PY
}
for r in $(seq $rounds); do for l in $libs; do run $l; done; done
