# same-box sweep of development switches on config 3 (development library): tools/dev/knob_sweep.sh "VAR=val ..." ...
for c in "$@"; do
  echo -n "[$c] "; env $c python tools/dev/c3.py ${KNOB_N:-100000} 2>&1 | tail -1
done
