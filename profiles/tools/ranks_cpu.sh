#!/bin/bash
# CPU time of an N-rank bench run on ONE GPU (gloo collectives, the ranks share the device: the fit time means nothing, the host side does):
#   bash profiles/tools/ranks_cpu.sh 8 [extra env, e.g. DDX_UPLOAD_SHARE=0]
# prints cgroup CPU seconds per fit (all ranks together) and the bench line's ms per fit
n=${1:-8}; shift
steps=3; warm=1
u0=$(awk '/usage_usec/{print $2}' /sys/fs/cgroup/cpu.stat); t0=$(awk '/throttled_usec/{print $2}' /sys/fs/cgroup/cpu.stat)
env DDX_BENCH_BACKEND=gloo "$@" timeout 420 python bench.py --gpus $n --steps $steps --warmup $warm --no-cpu-baseline --instrumented-steps 0 --resident-steps 0 > /tmp/ranks_cpu.json 2> /tmp/ranks_cpu.err
u1=$(awk '/usage_usec/{print $2}' /sys/fs/cgroup/cpu.stat); t1=$(awk '/throttled_usec/{print $2}' /sys/fs/cgroup/cpu.stat)
python - "$n" "$steps" "$warm" "$u0" "$u1" "$t0" "$t1" "$*" <<'PY'
import json, sys
n, steps, warm, u0, u1, t0, t1 = (int(x) for x in sys.argv[1:8])
try:
    d = json.loads(open("/tmp/ranks_cpu.json").read().strip().splitlines()[-1])
    ms = f'{d["ms_per_step"]} ms per fit, {d["host_cpu_seconds_per_step"]} CPU-s per fit over all ranks ({d["host_cpu_seconds_per_step_note"][-30:]}); host {d["host_seconds_last_step"]}' 
except Exception as e:
    ms = f"no bench line ({e})"
print(f"{n} ranks on one GPU [{sys.argv[8]}]: {ms}; whole run (data generation of {n} ranks + {steps + warm} fits): {(u1 - u0) / 1e6:.1f} CPU-s, throttled {(t1 - t0) / 1e6:.1f} s")
PY
tail -2 /tmp/ranks_cpu.err
