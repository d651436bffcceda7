#!/bin/bash
# k_walk plan sweep on the side configurations (GPU box): tools/dev/walk_sweep.sh
cd "$(dirname "$0")/../.."
for cfg in "c4t 400000" "c2t 200000" "c4t 50000"; do
  for v in "1024 3" "1024 4" "1024 6" "2048 4" "2048 8" "512 3"; do
    set -- $v
    echo -n "max=$1 frac=$2: "; BA_WALK_WAVE_MAX=$1 BA_WALK_WAVE_FRAC=$2 python tools/dev/sec.py $cfg | tail -1
  done
done
