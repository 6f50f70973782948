#!/bin/bash
# the headline frame under the launch-geometry switches (development aid): value and ms per step of bench.py, 200 steps each
#   bash tools/headline_variants.sh "VAR=value [VAR=value]" ...   (no arguments: the plain run)
cd ${GRAFT_REPO_ROOT:-.}
run() { echo -n "$1: "; env $1 RT_TRACE_GEOM=1 timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-two-ctx 2> /tmp/err.txt | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['value'], b['ms_per_step'], b.get('alt_precision',{}).get('value'))" || { tail -3 /tmp/err.txt; exit 1; }; grep "trace geometry" /tmp/err.txt | sort | uniq -c | head -3; }
run X=0
for v in "$@"; do run "$v"; done
