#!/bin/bash
out=gpurun_out; mkdir -p $out
bash profiles/collect.sh r06h > $out/r06h_collect.log 2>&1; tail -2 $out/r06h_collect.log
bash profiles/pmc_sq.sh r06h > $out/r06h_pmc_sq.log 2>&1; tail -12 $out/r06h_pmc_sq.txt
DDX_OPTIONS=bp_format=mx6 bash profiles/pmc_sq.sh r06h_mx > $out/r06h_mx_pmc_sq.log 2>&1; grep k_bp_ $out/r06h_mx_pmc_sq.txt | head
timeout 600 python profiles/tools/soak.py > $out/r06h_soak.txt 2>&1; echo "soak rc=$?"; tail -3 $out/r06h_soak.txt
