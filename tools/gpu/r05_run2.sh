#!/bin/bash
# round 5, GPU call: wide walk after the sampling fix -- parity, curve, counters of the timed launches
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=. PYTHONUNBUFFERED=1 TMPDIR=/tmp
TAG=${1:-r05b}
timeout 900 python -m pytest tests/test_wide.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/wide_case.py --points "${2:-set_b_mix:mix,dict_1k:k32,dict_1k:k128,dict_1k:k512,dict_1k:k1000,dict_10k:k32,dict_10k:k2048,dict_10k:k10000}" --log2-strings ${3:-20} --out gpurun_out/${TAG}_wide_curve.jsonl > gpurun_out/${TAG}_wide_curve.log 2>&1; echo "wide_case rc=$?"
python - <<PY
import json
for l in open("gpurun_out/${TAG}_wide_curve.jsonl"):
    d=json.loads(l)
    print(d["set"], d["corpus"], "visited", d["distinct_states_visited_in_sample"], "rows", d["wide_rows"], "| " + " | ".join("%s %s %.0f GB/s twice %.4f out_wide %.4f par %s" % (k, d[k]["kernel"], d[k]["GBps"], d[k].get("wave_chunk_share_walked_twice_by_the_wide_walk", -1), d[k]["measured_share_outside_wide_rows"], d[k]["parity_all_strings"]) for k in ("dense","wide","auto")))
PY
OUT=gpurun_out/${TAG}_pmc
mkdir -p $OUT
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python bench.py --set dict_1k --corpus k128 --walk 2 --steps 5 --warmup 1 --settle 10 --no-cpu --cold-launches 0 > $OUT/pmc_$i.log 2>&1 || echo "pmc pass $i failed"
done
python tools/summarize_pmc.py $OUT --last 5 > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
grep -h '"value"' $OUT/pmc_1.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench line: value', d['value'], 'kernel', d['roofline']['kernel'], 'kernel_avg_ms', d['roofline']['kernel_avg_ms'], 'traps', d.get('traps'), 'walk', d['config']['walk'])"
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
