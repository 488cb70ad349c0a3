#!/bin/bash
# stage A and stages B/C on disjoint CU sets (EDGEHIP_A_CUS, with --overlap): frames/s against the share of stage A
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --cpu-frames 0 ${OVL:---overlap} 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); print(j['value'], j['ms_per_step'], j['config'].get('estimation_ok'))"; }
echo -n "serial            "; OVL=" " run X=1
echo -n "overlap           "; run X=1
for k in ${KS:-64 80 96 112 128}; do echo -n "overlap, A on $k CUs "; run EDGEHIP_A_CUS=$k; done
