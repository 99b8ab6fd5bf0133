#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null > gpurun_out/counters_list.txt
grep -o -E "\b(TA|TD|TCP|SQ|TCC|GRBM|SPI)_[A-Za-z0-9_]+" gpurun_out/counters_list.txt | sort -u > gpurun_out/counter_names.txt
wc -l gpurun_out/counter_names.txt
grep -E "^(TA_|TD_)" gpurun_out/counter_names.txt | tr '\n' ' '
echo
grep -E "^TCP_" gpurun_out/counter_names.txt | tr '\n' ' '
echo
grep -E "^SQ_" gpurun_out/counter_names.txt | tr '\n' ' '
