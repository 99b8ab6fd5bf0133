#!/bin/bash
# round 5: the whole GPU suite at HEAD, the headline line with the counters reduced once per fence / after every pass /
# through RCCL on one rank, config C4 at its stated shape on one GPU, the multi-rank launch path over gloo, prefix at size
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=. PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r05_pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $O/r05_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r05_smoke.log
one() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.load(open("$O/$name.json")); c=d["config"]
    print("value", d["value"], "ms/step", d["ms_per_step"], "kernel", d["roofline"]["kernel_avg_ms"], "frac", d["roofline"]["frac"], "reduce", c.get("counter_reduce_ms"), c.get("reduce_backend"), "parity", d.get("cpu_baseline",{}).get("parity_vs_gpu"))
except Exception as e: print("unreadable", e)
PY
)"; }
one r05_bench_n1 --steps 20 --warmup 5
one r05_bench_n1_defaults
one r05_bench_force_dist_nccl --steps 20 --warmup 5 --force-dist --backend nccl --no-cpu
one r05_bench_force_dist_nccl_every_step --steps 20 --warmup 5 --force-dist --backend nccl --no-cpu --reduce-every-step
one r05_bench_reduce_every_step --steps 20 --warmup 5 --no-cpu --reduce-every-step
one r05_bench_c4_shard --c4 --steps 10 --warmup 3
timeout 900 python bench.py --gpus 8 --backend gloo --steps 5 --warmup 2 --settle 5 --no-cpu --cold-launches 0 > $O/r05_bench_8ranks_gloo.json 2> $O/r05_bench_8ranks_gloo.err; echo "8 ranks gloo rc=$?"; cut -c1-200 $O/r05_bench_8ranks_gloo.json; grep -o '"per_rank_[A-Za-z_]*": \[[^]]*\]' $O/r05_bench_8ranks_gloo.json
timeout 900 python bench.py --gpus 2 --backend gloo --c4 --steps 3 --warmup 1 --settle 3 --no-cpu --cold-launches 0 > $O/r05_bench_c4_2ranks_gloo.json 2> $O/r05_bench_c4_2ranks_gloo.err; echo "c4 2 ranks gloo rc=$?"; cut -c1-200 $O/r05_bench_c4_2ranks_gloo.json; grep -o '"per_rank_[A-Za-z_]*": \[[^]]*\]' $O/r05_bench_c4_2ranks_gloo.json
for lg in 18 20 21; do PREFIX_LOG2_STRINGS=$lg PREFIX_SETTLE=30 timeout 600 python tools/prefix_case.py 2>&1 | grep -v "^adapt"; done > $O/r05_prefix_sizes.log 2>&1; cat $O/r05_prefix_sizes.log
