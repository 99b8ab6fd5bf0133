#!/bin/bash
# two strings per lane with the per-wave "direct" mode (skip the attempt on the rows alone after a chunk that left them)
export PYTHONPATH=. PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_wide.py tests/test_selftest.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python tools/wide_case.py --log2-strings 20 --points dict_1k:k128,dict_1k:k512,dict_1k:k1000,dict_10k:k512,dict_10k:k2048,dict_10k:k10000,set_b_mix:mix --out gpurun_out/r05k_wide_curve.jsonl 2>&1 | grep -v amdgpu | cut -c1-400 | tail -12
