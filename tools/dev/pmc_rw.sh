cd /tmp; export TMPDIR=/tmp
for mode in base skip; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pp; 
    if [ $mode = skip ]; then export BA_SKIP_WALK=1; else unset BA_SKIP_WALK; fi
    rocprofv3 --kernel-trace --pmc $c -d /tmp/pp -o x -- python3 /root/repo/tools/dev/c3.py 100000 > /tmp/pp.log 2>&1
    python3 /root/repo/tools/prof_summary.py $(find /tmp/pp -name "*.db" | head -1) /tmp/pp.md "x" > /dev/null
    echo "$mode $c: $(grep k_multi /tmp/pp.md | grep $c | head -1)"
  done
done
