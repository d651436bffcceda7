# a sized batch's ranges launched together against one after the other
python tools/dev/sized_line.py 20000 2>&1 | grep "^sized"
BA_SIZED_SERIAL=1 python tools/dev/sized_line.py 20000 2>&1 | grep "^sized" | sed 's/^/[serial] /'
