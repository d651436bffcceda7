#!/bin/bash
# per-kernel times of a secondary configuration (run on the GPU box): tools/dev/prof_sec.sh c2t 200000 [tag]
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp BA_GEN_WORKERS=1
tag=${3:-$1}
rm -rf /tmp/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o kt -- python3 tools/dev/sec.py $1 $2 > gpurun_out/sec_$tag.log 2>&1
db=$(find /tmp/prof_$tag -name '*.db' | head -1)
[ -n "$db" ] && python3 tools/prof_summary.py "$db" gpurun_out/sec_$tag.md "sec.py $1 $2" > /dev/null
tail -1 gpurun_out/sec_$tag.log
sed -n 5,14p gpurun_out/sec_$tag.md | cut -c1-200
