# Same-box A/B of two library builds on the secondary configurations (tools/bench_configs.py); see tools/dev/ab.sh for building variant B.
cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
for r in 1 2; do for l in libblock_aligner_hip.so libblock_aligner_hip_b.so; do echo "== $l"; python3 - $l <<'PY' 2>&1 | grep -v inputs
import sys, os, runpy
sys.path.insert(0, os.getcwd())
import block_aligner_amd.hip as H
H.LIB_PATH = os.path.join(os.path.dirname(H.LIB_PATH), sys.argv[1])
sys.argv = ["bench_configs.py"]
runpy.run_path("tools/bench_configs.py", run_name="__main__")
PY
done; done
