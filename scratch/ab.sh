#!/bin/bash
# alternating A/B/C of trees on one box: new = repo root, others under scratch/
for i in 1 2 3; do
  for t in new ab_mid ab_old; do
    d=$GRAFT_REPO_ROOT; [ "$t" != new ] && d=$GRAFT_REPO_ROOT/scratch/$t
    (cd $d && timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --instrumented-steps 0 --resident-steps 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$t',d['ms_per_step'],d['host_seconds_last_step'])")
  done
done
