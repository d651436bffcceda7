cd /tmp; export TMPDIR=/tmp
for mode in base skip; do
  rm -rf /tmp/pp
  if [ $mode = skip ]; then export BA_SKIP_WALK=1; else unset BA_SKIP_WALK; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d /tmp/pp -o x -- python3 /root/repo/tools/dev/c3.py 100000 > /tmp/pp.log 2>&1
  python3 /root/repo/tools/prof_summary.py $(find /tmp/pp -name "*.db" | head -1) /tmp/pp.md "x" > /dev/null
  echo "== $mode"; grep k_multi /tmp/pp.md | grep "SQ_" | awk -F'|' '{print $3, $5}'
  rm -rf /tmp/pp
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY -d /tmp/pp -o x -- python3 /root/repo/tools/dev/c3.py 100000 > /tmp/pp.log 2>&1
  python3 /root/repo/tools/prof_summary.py $(find /tmp/pp -name "*.db" | head -1) /tmp/pp.md "x" > /dev/null
  grep k_multi /tmp/pp.md | grep "SQ_" | awk -F'|' '{print $3, $5}'
done
