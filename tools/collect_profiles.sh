#!/bin/bash
# Copy a round's profile summaries from gpurun_out/ (scratch) to profiles/ (tracked) and stamp the roofline record with the commit
# whose kernel sources it was measured on:  tools/collect_profiles.sh r03
tag=${1:-r03}
cd "$(dirname "$0")/.."
for f in kernel_trace_stats.md pmc_fetch.md pmc_write.md pmc_sq.md sec_c2.md sec_c2t.md sec_c4.md sec_c4t.md sec_c5.md roofline.json; do
  [ -f gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f
done
[ -f gpurun_out/${tag}_kernel_trace_stats.json ] && cp gpurun_out/${tag}_kernel_trace_stats.json profiles/${tag}_bench_under_kernel_trace.json
python3 - "$tag" <<'PY'
import json, subprocess, sys, os
sys.path.insert(0, "tools")
from kernel_hash import kernel_hash
tag = sys.argv[1]
p = f"profiles/{tag}_roofline.json"
r = json.load(open(p))
here = kernel_hash()
r["commit"] = subprocess.check_output(["git", "rev-parse", "--short=12", "HEAD"]).decode().strip() + (" (working tree: kernel sources match)" if r["kernel_sources_sha16"] == here else " (NOTE: kernel sources have changed since this profile: sha16 now %s)" % here)
json.dump(r, open(p, "w"), indent=1)
print(p, r["commit"], r["kernel_sources_sha16"])
PY
