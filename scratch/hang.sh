#!/bin/bash
# 1. the soak with host_wait=block and ddx_destroy's steps on stderr; 2. the same under rocgdb, interrupted if it hangs
DDX_OPTIONS="upload_debug=1" timeout 200 python scratch/soak_block.py block 120 > gpurun_out/hang_soak.log 2> gpurun_out/hang_soak.err
echo "rc=$?" >> gpurun_out/hang_soak.log
grep -a "ddx_destroy" gpurun_out/hang_soak.err | tail -12 > gpurun_out/hang_steps.txt
grep -a -c "ddx_destroy.*done" gpurun_out/hang_soak.err >> gpurun_out/hang_steps.txt
timeout -s INT 240 /opt/rocm/bin/rocgdb -q -batch -ex "set pagination off" -ex "handle SIGSEGV nostop noprint pass" -ex run -ex "thread apply all bt 30" --args python scratch/soak_block.py block 120 > gpurun_out/hang_gdb.txt 2>&1
tail -5 gpurun_out/hang_soak.log; cat gpurun_out/hang_steps.txt; grep -c "^Thread" gpurun_out/hang_gdb.txt
