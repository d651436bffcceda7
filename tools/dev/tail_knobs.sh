# config 3: pairs kept out of the slots at the end of the batch (BA_MQ_DRAIN; default: fill waves / 4) and the walkers' look-ahead, same box
for e in "X=1" "BA_MQ_DRAIN=0" "BA_MQ_DRAIN=512" "BA_MQ_DRAIN=2048" "BA_WORK_CHUNK=2" "X=2"; do echo -n "[$e] "; env $e python tools/dev/c3.py 100000 2>&1 | tail -1; done
