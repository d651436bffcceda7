# k_multi against the per-pair kernel by pair length (pairs of MF_LEN bases, ~9 % edits, 100-base tails, block 128..512, 30 k pairs), same box
for L in 600 1000 1500; do echo "== length $L"; MF_LEN=$L python tools/dev/multi_from.py 30000 2>&1 | grep "n="; done
