#!/bin/bash
# Round 2, experiments on the tiled kernel: wave priority while walking, de-phased waves, no compact tier, strides;
# plus the PCIe probe for the host-pointer mode.
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02exp1
mkdir -p $OUT
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], r["kernel"], "avg", r["kernel_avg_ms"], "min", r["kernel_min_ms"], "frac", r["frac"], "cold", d.get("value_before_adapt"), d["match_counts"]["final"])'
for rep in 1 2; do
for v in 0 5 6 7 8; do
  env PIRE_HIP_TILED_VARIANT=$v timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "$P" "v$v" | tee -a $OUT/variants.log
done
done
for st in 4096 4224 4352 4112 8192; do
  timeout 300 python bench.py --no-cpu --steps 30 --warmup 10 --stride $st 2>&1 | tail -1 | python -c "$P" "stride $st" | tee -a $OUT/stride.log
done
for lg in 18 19 21 22; do
  timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 --log2-strings $lg 2>&1 | tail -1 | python -c "$P" "2^$lg strings" | tee -a $OUT/sizes.log
done
echo "== PCIe probe"
timeout 300 tools/pcie_probe 2>&1 | tee $OUT/pcie_probe.log
