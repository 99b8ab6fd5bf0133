#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02exp3
mkdir -p $OUT
env PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_LAUNCH=1 PIRE_HIP_TILED_VARIANT=9 timeout 300 python bench.py --set c2_single --no-cpu --steps 2 --warmup 1 2>&1 | grep "pire_hip:" | sort | uniq -c | tee $OUT/launch.log
env PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_LAUNCH=1 PIRE_HIP_TILED_VARIANT=0 timeout 300 python bench.py --set c2_single --no-cpu --steps 2 --warmup 1 2>&1 | grep "pire_hip:" | sort | uniq -c | tee -a $OUT/launch.log
rocminfo | grep -i "lds\|wave\|workgroup\|Compute Unit" | sort | uniq -c | head -20 | tee -a $OUT/launch.log
