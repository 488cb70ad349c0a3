cd $GRAFT_REPO_ROOT
for m in 0 1 2 0 1 2; do
  echo -n "FWD_MODE=$m  "
  EDGEHIP_FWD_MODE=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], 'tvr', k['B.try_velrot'], 'fwd', k['C.forward_match'], 'rot', k['C.rotate'])"
done
