#!/bin/bash
# whole fits with the int8 (A) and the MX (B) form of the bit-plane products, same box, alternating:  bash profiles/tools/format_ab.sh
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --instrumented-steps 0 --resident-steps 6"
for v in A B A B A B A B A B; do
  case $v in A) o="--option bp_format=int8";; B) o="--option bp_format=mx6";; esac
  $B $o 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', '$o', d['ms_per_step'], 'ms from host;', round(1e8/d["value_resident"],2) if d["value_resident"] else None, 'ms resident')"
done
