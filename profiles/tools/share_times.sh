#!/bin/bash
# what DESIGN section 6's projection is computed from: the headline fit, the same on ONE device context, and the busiest rank's share of
# the ten iterations at 2 / 4 / 8 GPUs (5 / 3 / 2 iterations) on the one GPU there is.   bash profiles/tools/share_times.sh
run() { env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --instrumented-steps 0 --resident-steps 0 $ARGS 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);h=d['host_seconds_last_step'];print('$LABEL', d['ms_per_step'], 'ms per fit; stage', h['stage'], 'prologue', h['prologue'], 'device_stages', h['device_stages'], 'sharding:', d['config']['sharding'])"; }
LABEL="10 iterations, default contexts:" ARGS="" run A=1
LABEL="10 iterations, ONE context:    " ARGS="" run DDX_STREAMS=1
LABEL=" 5 iterations (share at 2 GPUs):" ARGS="--iters 5" run A=1
LABEL=" 3 iterations (share at 4 GPUs):" ARGS="--iters 3" run A=1
LABEL=" 2 iterations (share at 8 GPUs):" ARGS="--iters 2" run A=1
LABEL=" 1 iteration                   :" ARGS="--iters 1" run A=1
