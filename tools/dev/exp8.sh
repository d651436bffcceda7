cd /tmp; export TMPDIR=/tmp BA_GEN_WORKERS=1
rm -rf /tmp/ps; rocprofv3 --kernel-trace --stats -d /tmp/ps -o x -- python3 /root/repo/tools/dev/sec.py c5 80000 > /tmp/ps.log 2>&1
tail -1 /tmp/ps.log
python3 /root/repo/tools/prof_summary.py $(find /tmp/ps -name "*.db" | head -1) /root/repo/gpurun_out/sec_c5.md "c5 80000" | head -12
