#!/bin/bash
# soak with the library's own sleeping waits: must exit by itself; CPU seconds per fit for both modes
for mode in block spin; do
  timeout 300 python scratch/soak_block.py $mode 120 > gpurun_out/soak_$mode.log 2> gpurun_out/soak_$mode.err
  echo "mode $mode rc=$?" >> gpurun_out/soak_$mode.log
  tail -6 gpurun_out/soak_$mode.log
done
