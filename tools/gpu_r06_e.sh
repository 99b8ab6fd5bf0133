#!/bin/bash
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
for order in 0 1; do
  echo "== stage clocks, PIRE_HIP_NO_LENGTH_ORDER=$order"
  PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_RAGGED_CLOCKS=1 PIRE_HIP_NO_LENGTH_ORDER=$order WIDE_CASE_LEGS=wide timeout 600 python tools/wide_case.py --points blacklist_1k:urls --log2-urls 23 --reps 3 2>&1 | grep "ragged clocks" | tail -3 | cut -c1-400
done
