#!/bin/bash
# runs the soak with host_wait=block; if the process is still alive 60 s after its loop finished, dumps every thread's stack
python scratch/soak_block.py block 120 > gpurun_out/hang_soak.log 2>&1 &
PID=$!
for i in $(seq 1 400); do
  kill -0 $PID 2>/dev/null || break
  if grep -q "loop done" gpurun_out/hang_soak.log 2>/dev/null; then
    n=$((n+1))
    if [ "$n" -gt 45 ]; then break; fi
  fi
  sleep 1
done
if kill -0 $PID 2>/dev/null; then
  echo "still alive: dumping stacks" >> gpurun_out/hang_soak.log
  timeout 120 /opt/rocm/bin/rocgdb -p $PID -batch -ex "set pagination off" -ex "thread apply all bt 25" > gpurun_out/hang_bt.txt 2>&1
  kill -9 $PID
else
  echo "exited by itself" >> gpurun_out/hang_soak.log
fi
tail -5 gpurun_out/hang_soak.log
grep -c "^Thread" gpurun_out/hang_bt.txt 2>/dev/null
