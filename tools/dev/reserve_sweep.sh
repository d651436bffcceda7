# config 3: how many of the batch's last hand-offs are left to the emptied fill waves' whole-wave walks (BA_TB_RESERVE; default: fill waves / 2 = 2048)
for v in 2048 4096 6144 8192 12288; do echo -n "[reserve $v] "; BA_TB_RESERVE=$v python tools/dev/c3.py 100000 2>&1 | tail -1; done
echo -n "[default again] "; python tools/dev/c3.py 100000 2>&1 | tail -1
