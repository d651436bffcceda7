#!/bin/bash
# round 6: the short-pair indel set (tools/dev/multi_from.py: where k_multi's traced solo driver costs most) under the register budgets of the three geometries
cd "$(dirname "$0")/../.."
for g in 0 3 2; do echo "== BA_MQ_GEOM=$g"; BA_MQ_GEOM=$g timeout 200 python tools/dev/multi_from.py 20000 60000 200000 2>&1 | grep -v Warn | grep "k_multi" | cut -c1-100; done
timeout 200 python tools/dev/multi_from.py 20000 60000 200000 2>&1 | grep "k_align" | cut -c1-100
