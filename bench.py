#!/usr/bin/env python3
"""bench.py -- scanned GB/s of the MI355X scan path on BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torch.distributed environment launches the N ranks itself (it re-executes
itself under torch.distributed.run on 127.0.0.1); under an external launcher it is one of the ranks.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 8 regexps glued with
Scanner::Glue ("set_a", patterns of the reference's own tools/bench/run-bench), 2^20 strings x 4 KiB per GPU,
synthetic corpus generated on the device (oracle/corpus.h definition, planted witnesses), Begin().Run().End()
per string.  One "step" = one pass of the hot path over the whole per-GPU batch + the match-count reduce.
Inputs are resident in HBM when the timed region starts.  Multi-GPU: one process per GPU, the corpus is sharded
by string index (rank r owns global strings [r*n, (r+1)*n)), no data-path collective; the only exchange is the
all-reduce (RCCL) of the uint64[regexps+2] match counters per step.  Weak scaling.

Wide working sets (BASELINE config 5 as north_star means it: transitions that do not fit the 255 dense rows):
`--set dict_1k|dict_10k --corpus k32|k128|k512|k2048|k<words>` -- a dictionary scanner built the way the reference's
samples/blacklist/blacklist.cpp builds one, over records of dictionary labels and filler words -- and `--set set_b_mix
--corpus mix`; the line then carries `working_set` (distinct states visited, shares outside the dense / wide rows).

Other workloads (informational lines of the same shape): `--set c2_single|set_b|set_d` (BASELINE configs 2 / 5a),
`--set slow_x40_utf8` (config 5b, SlowScanner), `--corpus cxx` (the reference's own benchmark corpus,
tools/bench/test_file, repeated as tools/bench/run-bench does, instead of the synthetic corpus; `--one-string` scans it
as ONE string as the reference's bench does, through the segmented scan).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import hashlib  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)
SEED = 0x5EED5EED
SEED_HELDOUT = 0x0DDBA11   # the corpus the dense-row ranking is learned on (never the timed one)
WORKLOADS = {"set_a": "C3: 8 regexps glued via Scanner::Glue, LDS-resident dense rows",
             "c2_single": "C2: single Scanner hello\\s+w.+d$",
             "set_b": "C5a: 8 glued regexps, 8952-state table with HBM-resident transitions",
             "set_d": "8 glued unanchored regexps (pire_ut.cpp patterns)",
             "dict_1k": "C5 wide: dictionary Scanner of 1 000 domains (samples/blacklist construction, Surround()ed), 4 084 states",
             "dict_10k": "C5 wide: dictionary Scanner of 10 000 domains (samples/blacklist construction, Surround()ed), 30 202 states",
             "set_b_mix": "C5a table (8 glued regexps, 8 952 states) over fragments that keep its patterns half matched"}


def kernel_sources_sha16() -> str:
    """sha256 (16 hex digits) over the sources the headline kernel is compiled from: a committed PMC file
    (profiles/*_pmc_traffic.json, written by tools/make_pmc_json.py) carries the hash of the sources it was measured
    with, and a file whose hash is not this one is refused -- its traffic figure describes another kernel."""
    h = hashlib.sha256()
    for name in ("tiled.hip", "device_common.h", "internal.h"):
        with open(os.path.join(ROOT, "pire_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log2-strings", type=int, default=20, help="strings per GPU = 2^this (headline: 20)")
    ap.add_argument("--strings", type=int, default=0, help="strings per GPU, any number (overrides --log2-strings)")
    ap.add_argument("--len", type=int, default=4096, help="bytes per string (headline: 4096)")
    ap.add_argument("--stride", type=int, default=0, help="bytes between strings in memory (default: --len, contiguous)")
    ap.add_argument("--set", default="set_a", help="golden pattern set (set_a = headline)")
    ap.add_argument("--corpus", default="synthetic",
                    help="cxx: the reference's own benchmark corpus (tools/bench/test_file, C++ text) repeated to the "
                         "batch size as run-bench:126-138 does, instead of the synthetic corpus; for the wide sets of "
                         "tests/golden/wide.json (--set dict_1k|dict_10k|set_b_mix): one of their token corpora (k32 ... / mix)")
    ap.add_argument("--walk", type=int, default=0, choices=[0, 1, 2],
                    help="pire_hip_config.walk_variant: 0 the library's choice between the dense rows and the class-indexed "
                         "walk, 1 always the dense rows, 2 always the class-indexed walk (same results)")
    ap.add_argument("--zip", type=int, default=0, choices=[0, 1, 2],
                    help="pire_hip_config.zip_variant: 0 the library's choice of the class-indexed walk's LDS image (zipped once adapt() "
                         "has measured the scans leaving the plain rows), 1 never zipped, 2 always (same results)")
    ap.add_argument("--one-string", action="store_true", help="with --corpus cxx: the whole text as ONE string")
    ap.add_argument("--cpu-sample-log2", type=int, default=20, help="strings in the CPU baseline sample = 2^this")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-adapt", action="store_true", help="do not learn the dense-row ranking on the held-out corpus")
    ap.add_argument("--settle", type=int, default=60,
                    help="untimed passes in front of the warm-up passes of every timed leg, so that the GPU's clocks "
                         "have settled (an idle MI355X needs 20-30 launches, profiles/r03_warmup_curve.log); 0 = none")
    ap.add_argument("--cold-launches", type=int, default=20,
                    help="launches of the from-idle leg (`cold_start` in the line): after the timed region the GPU is "
                         "left idle for --cold-idle-ms, then this many passes are timed; 0 = no such leg")
    ap.add_argument("--cold-idle-ms", type=int, default=300)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even for ONE rank (world-1 process group): the counters then go "
                         "through dist.all_reduce of the chosen backend -- with nccl, through RCCL -- on a one-GPU box")
    ap.add_argument("--c4", action="store_true",
                    help="BASELINE configs[3] at its stated shape: 64 Mi x 4 KiB strings over 8 GPUs = 2^23 strings (32 GiB) per "
                         "GPU (same as --log2-strings 23; the CPU baseline then checks the first 2^22 strings of rank 0)")
    ap.add_argument("--reduce-every-step", action="store_true",
                    help="all-reduce the match counters after every pass (round 4's form: 31-44 us per step on one rank, the "
                         "all-reduce kernel waits for a CU of a GPU the persistent scan fills) instead of once per fence")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to "
                    "exercise the multi-rank control flow on a box with fewer GPUs than ranks)")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: run the N ranks under torch.distributed.run ourselves."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def cxx_corpus_bytes() -> np.ndarray:
    """The C++ text of the `--corpus cxx` workload: the reference's own benchmark corpus, tools/bench/test_file
    (20 485 bytes; data fixture tests/golden/ref_bench_test_file.gz).  run-bench:126-138 doubles it until the big file
    is large enough; doubling a file is repeating it, so the batch is this text repeated."""
    from pire_amd import workloads as W

    return np.frombuffer(W.ref_bench_file(), dtype=np.uint8)


def cpu_baseline(blob, host_strings, length, gpu_idx, gpu_fin):
    """The reference's own Runner(sc).Begin().Run().End() (oracle/_ref, kind 'reference') -- or the C port
    (kind 'port') if the prebuilt reference library is not on this box -- timed on this host's cores over
    `host_strings` (the first strings of rank 0's batch, [k, length] u8); also the parity check of the GPU results
    on that sample.  The only place of the benchmark that touches oracle/ (test infrastructure)."""
    from oracle import binding as ob

    cores = os.cpu_count() or 1
    threads = min(cores, 256)
    sample = host_strings.shape[0]
    text = host_strings.reshape(-1)
    offs = np.arange(sample + 1, dtype=np.uint64) * length
    nbytes = sample * length
    out = {"cores": threads, "host_cores": cores, "sample": f"first {sample} strings x {length} B of rank 0's batch "
           f"({nbytes / 2**20:.0f} MiB), Begin().Run().End() per string"}
    runs = {}
    if ob.ref_available():
        try:
            ref = ob.RefScanner.load(blob)
            out["kind"] = "reference"
            for label, kind, thr in (("scanner_1t", 0, 1), ("nonreloc_1t", 1, 1), ("scanner_all", 0, threads),
                                     ("nonreloc_all", 1, threads)):
                sub = sample if thr > 1 else max(sample // 16, 1)
                best = None
                for _ in range(3 if thr > 1 else 1):      # all-core leg: best of 3 passes over the sample
                    t0 = time.time()
                    idx, fin = ref.run(text[:sub * length], offs[:sub + 1], kind=kind, threads=thr)
                    dt = time.time() - t0
                    best = dt if best is None else min(best, dt)
                runs[label] = {"GBps": round(sub * length / best / 1e9, 4), "seconds": round(best, 3), "threads": thr,
                               "strings": sub}
                if label == "scanner_all":
                    cpu_idx, cpu_fin = idx, fin
            out["value"] = runs["scanner_all"]["GBps"]
            out["impl"] = "Pire::Scanner (unmodified reference, g++ -O2), std::thread sharding by string index is ours"
        except Exception as e:   # prebuilt .so not loadable here: fall back to the port, say so
            out["ref_error"] = str(e)
            runs = {}
    if not runs:
        o = ob.OracleScanner(blob)
        out["kind"] = "port"
        for label, thr in (("port_1t", 1), ("port_all", threads)):
            sub = sample if thr > 1 else max(sample // 16, 1)
            t0 = time.time()
            idx, fin = o.run(text[:sub * length], offs[:sub + 1], threads=thr)
            dt = time.time() - t0
            runs[label] = {"GBps": round(sub * length / dt / 1e9, 4), "seconds": round(dt, 3), "threads": thr,
                           "strings": sub}
            if label == "port_all":
                cpu_idx, cpu_fin = idx, fin
        out["value"] = runs["port_all"]["GBps"]
        out["impl"] = "oracle/pire_oracle.c byte-wise port, pthread sharding"
    out["unit"] = "GB/s"
    out["runs"] = runs
    out["parity_vs_gpu"] = bool((cpu_idx == gpu_idx[:sample]).all() and (cpu_fin == gpu_fin[:sample]).all())
    out["distinct_end_states_in_sample"] = int(len(np.unique(cpu_idx)))
    return out


def bench_slow(args):
    """BASELINE config 5b (informational, `--set slow_x40_utf8`): Pire::SlowScanner, NFA simulation.  Same JSON
    shape as the headline; the kernel is pirehip::SlowScanKernel (VALU/LDS bound, not HBM bound)."""
    import torch

    import pire_amd
    from pire_amd import binding as pb
    from pire_amd import workloads as W

    case = W.slow_case(args.set)
    blob = W.load_blob(case["blob"])
    table = pire_amd.SlowTable(blob)
    torch.cuda.set_device(0)
    n, length = args.strings or (1 << args.log2_strings), args.len
    gap = int(case["pattern"].split("{")[1].split("}")[0]) if "{" in case["pattern"] else 40
    plants = pb.make_plants([(b"x" + b"y" * min(gap, 60), True), (b"zx" + b"w" * min(gap - 1, 59), True)])   # witness (for x.{40}$) / near miss
    text = torch.empty((n, length), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    pire_amd.corpus_fill_device(text.data_ptr(), SEED, 0, n, length, length, plants, stream)
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    flags = pb.FLAG_BEGIN | pb.FLAG_END

    def step():
        cnt.zero_()
        table.run_strided_device(text.data_ptr(), n, length, length, flags, fin.data_ptr(), 0, cnt.data_ptr(), stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = [a.elapsed_time(b) for a, b in ev]
    value = float(n) * length * args.steps / elapsed / 1e9
    kernel, symbol = pb.last_kernel(), pb.last_kernel_symbol()
    # The bound of this scanner is the LDS gather, not HBM (work per byte is proportional to the active
    # states).  Per input byte a lane does one letter lookup (u8) and, in the list forms, one table lookup per list
    # slot: 4 slots of 8 bytes (SlowScanKernel, <= 256 states) or up to 16 slots of 4 bytes (SlowListKernel).  The LDS
    # serves 64 banks x 4 bytes per clock and CU; conflict-free, a wave instruction of 64 lanes costs 64 * width / 256
    # clocks (2 for a u8 / b32 read -- two half-waves --, 4 for a b64 read).  Roofline = bytes/s the chip could scan if
    # every lookup were conflict-free and nothing else took time, at the 2.4 GHz peak clock, 256 CUs.
    # Round 4: the list kernel walks only the groups of four slots in use (one, on this pattern), and what bounds it is
    # not the LDS but VALU issue: 30.8 VALU instructions per wave and input byte (rocprofv3 PMC of this very workload,
    # profiles/r04_slow_list_pmc.txt) x 4 clocks each on one of a CU's 4 SIMDs.  Both bounds are stated, `frac` is
    # against the lower one.
    valu_per_wave_byte = {"slow_list": 30.8}.get(kernel)
    clocks_per_wave_byte = {"slow": 2 + 4 * 4, "slow_list": 2 + 4 * 2}.get(kernel)
    if clocks_per_wave_byte:
        lds_peak = 256 * 2.4e9 / clocks_per_wave_byte * 64 / 1e9
        bound, peak = "lds", lds_peak
        model = (f"{clocks_per_wave_byte} conflict-free LDS clocks per 64 input bytes of a wave "
                 f"(1 letter lookup + {'4 slots x b64' if kernel == 'slow' else '4 slots in use x b32; all 16: 34 clocks'}), "
                 f"256 CUs x 2.4 GHz = {lds_peak:.0f} GB/s")
        if valu_per_wave_byte:
            valu_peak = 256 * 4 * 2.4e9 / (valu_per_wave_byte * 4) * 64 / 1e9
            model += (f"; VALU issue: {valu_per_wave_byte} instructions per wave and byte (measured, PMC) x 4 clocks, 1 024 SIMDs x "
                      f"2.4 GHz = {valu_peak:.0f} GB/s")
            if valu_peak < lds_peak:
                bound, peak = "valu", valu_peak
    else:   # wave-per-string form: latency bound per step (fences + atomics), no closed-form throughput bound
        peak, bound, model = None, "latency", "one wave per string: ~1 us per byte and wave (set clear, scatter, fence)"
    achieved = n * (length + 1) / (np.mean(ms) * 1e-3) / 1e9
    res = {"metric": "scanned GB/s, SlowScanner (config 5b)", "value": round(value, 2), "unit": "GB/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": f"C5b: SlowScanner {case['pattern']!r} ({case['options'] or 'latin1'}), "
                                  f"{case['geometry']['states']} NFA states, {n} x {length} B strings",
                      "strings_per_gpu": n, "string_bytes": length},
           # (every line's roofline is the path's stated bound, HBM read bandwidth; the kernel's own instruction-count
           # bound -- what actually limits an NFA simulation -- is kept next to it as `kernel_model`)
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "kernel_model": {"bound": bound, "peak": round(peak, 1) if peak else None,
                                         "frac": round(achieved / peak, 4) if peak else None, "model": model},
                        "kernel": symbol, "kernel_avg_ms": round(float(np.mean(ms)), 4)},
           "match_counts": {"final": int(cnt[0].item()), "strings": int(cnt[1].item())}}
    if not args.no_cpu:
        from oracle import binding as ob   # the cpu_baseline leg: the checker, never the thing measured

        sample = min(n, 1 << args.cpu_sample_log2)   # the whole batch when asked for (C5b at its stated size: 16 GiB, ~8 s on 256 cores)
        host = ob.corpus_fill(SEED, 0, sample, length, plants, threads=min(os.cpu_count() or 1, 64))
        offs = np.arange(sample + 1, dtype=np.uint64) * length
        threads = min(os.cpu_count() or 1, 256)
        if ob.ref_available():
            ref = ob.RefSlowScanner.load(blob)
            t0 = time.time()
            rf, _ = ref.run(host.reshape(-1), offs, threads=threads)
            dt = time.time() - t0
            kind = "reference"
        else:
            o = ob.OracleSlowScanner(blob)
            t0 = time.time()
            rf, _ = o.run(host.reshape(-1), offs)
            dt = time.time() - t0
            kind, threads = "port", 1
        res["cpu_baseline"] = {"value": round(sample * length / dt / 1e9, 4), "unit": "GB/s", "cores": threads,
                               "kind": kind, "sample": f"first {sample} strings x {length} B",
                               "parity_vs_gpu": bool((rf == fin[:sample].cpu().numpy()).all())}
    print(json.dumps(res))


def bench_c1(args):
    """BASELINE config C1 (informational, `--set c1_nonreloc`): ONE NonrelocScanner, pattern hello\\s+w.+d$, 10 000 x 256 B
    ASCII strings on the reference's own CPU Run() -- "plumbing, no GPU".  The measured thing here IS the reference
    (oracle/_ref, the unmodified library): `value` = its all-core rate, n_gpus 0.  When a GPU is present the same batch
    also goes through pire_hip_run_strided and every result is compared (`gpu_parity`)."""
    from oracle import binding as ob   # this config is the CPU baseline itself
    from pire_amd import workloads as W

    big = W.pattern_set("c2_single")
    blob = W.load_blob(big["blob"])
    n, length = args.strings or 10000, 256 if args.len == 4096 else args.len
    host = ob.corpus_fill(SEED, 0, n, length, W.plants_for(big), threads=min(os.cpu_count() or 1, 16))
    offs = np.arange(n + 1, dtype=np.uint64) * length
    ref = ob.RefScanner.load(blob) if ob.ref_available() else None
    runs, want = {}, None
    cores = min(os.cpu_count() or 1, 256)
    legs = (("nonreloc_1t", 1, 1), ("nonreloc_all", 1, cores), ("scanner_1t", 0, 1)) if ref else (("port_1t", 0, 1),)
    for label, kind, thr in legs:
        best = None
        for _ in range(max(3, args.steps // 10)):
            t0 = time.perf_counter()
            if ref:
                idx, fin = ref.run(host.reshape(-1), offs, kind=kind, threads=thr)
            else:
                idx, fin = ob.OracleScanner(blob).run(host.reshape(-1), offs, threads=thr)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        runs[label] = {"GBps": round(n * length / best / 1e9, 4), "seconds": round(best, 6), "threads": thr}
        if want is None:
            want = (idx, fin)
        assert (idx == want[0]).all() and (fin == want[1]).all()
    # a 2.5 MB batch: one thread is often faster than spawning all of them; the line reports the better of the two
    top = max((k for k in runs if not k.startswith("scanner")), key=lambda k: runs[k]["GBps"])
    res = {"metric": "scanned GB/s, config C1 (plumbing): NonrelocScanner hello\\s+w.+d$ on the reference CPU Run()",
           "value": runs[top]["GBps"], "unit": "GB/s", "n_gpus": 0, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(runs[top]["seconds"] * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": f"C1: single NonrelocScanner {big['patterns'][0]!r}, {n} x {length} B ASCII strings, "
                                  "reference CPU Run() (Begin().Run().End() per string)", "corpus_seed": SEED},
           "cpu_baseline": {"value": runs[top]["GBps"], "unit": "GB/s", "cores": runs[top]["threads"],
                            "kind": "reference" if ref else "port", "sample": f"the whole batch, {n} x {length} B", "runs": runs},
           "match_counts": {"final": int(want[1].sum()), "strings": n, "distinct_end_states": int(len(np.unique(want[0])))}}
    try:
        import torch

        if torch.cuda.is_available():
            import pire_amd

            t = pire_amd.Table(blob)
            gi, gf = t.run_strided_host(host)
            res["gpu_parity"] = bool((gi == want[0]).all() and (gf == want[1]).all())
    except ImportError:
        pass
    print(json.dumps(res))


def main():
    args = parse()
    if args.c4:
        args.log2_strings = 23
        args.cpu_sample_log2 = max(args.cpu_sample_log2, 22)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.set.startswith("slow_"):
        return bench_slow(args)
    if args.set == "c1_nonreloc":
        return bench_c1(args)
    import torch
    import torch.distributed as dist

    import pire_amd
    from pire_amd import binding as pb
    from pire_amd import distributed as pd
    from pire_amd import workloads as W

    rank, local, world = pd.world_info()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks over RCCL need {world} GPUs, this box has {ndev} "
                         "(--backend gloo runs the ranks on the GPUs there are, for control-flow tests)")
    local_dev = local % max(ndev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1 or args.force_dist:
        pd.init(args.backend, dev)   # "nccl" is RCCL on ROCm
        pd.FORCE_SINGLE_RANK_COLLECTIVES = bool(args.force_dist)
    # who is here: every rank's host / device / PCI bus id, gathered through the process group itself.  Over RCCL two ranks on ONE
    # GPU would report N x the throughput of a single device as "N GPUs": refused.
    rank_devices = pd.describe_ranks(dev)
    clashes = pd.ranks_sharing_a_device(rank_devices)
    if args.backend == "nccl" and world > 1 and clashes:
        raise SystemExit(f"ranks share a GPU {clashes}: {rank_devices} -- one process per GPU (torch.distributed.run --nproc-per-node N "
                         "on a node with N GPUs); --backend gloo runs the launch path on fewer GPUs for control-flow tests")

    # The library re-ranks a table's dense rows by itself when scans keep leaving them (pire_hip_config.auto_adapt).
    # Here the ranking is learned explicitly, on a held-out corpus (step 1 below), and must not move afterwards: off.
    pb.set_config(auto_adapt=1, walk_variant=args.walk, zip_variant=args.zip)
    wide_entry = None
    try:
        wide_entry = W.wide_set(args.set)   # tests/golden/wide.json: dictionary scanners, token-mixture corpora
    except KeyError:
        pass
    if wide_entry:
        if args.corpus == "synthetic":
            args.corpus = sorted(wide_entry["samples"])[0]
        big = {"blob": wide_entry["blob"],
               "patterns": wide_entry.get("patterns") or [f"{wide_entry['words']} fixed strings joined with Fsm::operator|=, "
                                                          f"{wide_entry['mode']} (samples/blacklist/blacklist.cpp:65-76)"]}
    else:
        big = W.pattern_set(args.set)
    blob = W.load_blob(big["blob"])
    table = pire_amd.Table(blob)
    table.upload()
    plants = None if wide_entry else W.plants_for(big)
    n = args.strings or (1 << args.log2_strings)
    length = args.len
    stride = args.stride or length
    assert stride >= length and (args.corpus == "synthetic" or stride == length)
    stream = torch.cuda.current_stream().cuda_stream

    # rank r owns global strings shard_range(n*world, r, world) = [r*n, (r+1)*n): generated in place, never
    # crosses PCIe or xGMI
    first, last = pd.shard_range(n * world, rank, world)
    assert last - first == n
    base_text = None
    wide_base = None
    wide_order = None
    if wide_entry:
        # `nbase` distinct records of the token corpus (built on the host, numpy), repeated over the batch on the device:
        # record i of rank r = base[(first + i + 1237 * (i // nbase)) % nbase]
        nbase = min(n, 16384)
        assert n % nbase == 0
        wide_base = W.wide_records(wide_entry, args.corpus, SEED, nbase, length)
        # ... every repeat rotated by its own number of records (not a multiple of 64): a plain repeat has a period of
        # nbase / 64 = 256 tasks -- the number of CUs --, and a kernel that hands task b + 256 w to wave w of block b then
        # walks the SAME 64 records in all 16 waves of a CU, whose table loads hit each other's lines in the L1
        # (dict_10k / k10000: 1.04 instead of 0.54 TB/s; found when the wide kernels' task order changed)
        wide_order = W.rotated_repeat_order(n, nbase, first)
        text = torch.as_tensor(wide_base, device=dev).index_select(0, torch.as_tensor(wide_order, device=dev)).contiguous()
    elif args.corpus == "cxx":
        # the reference's benchmark text repeated over the whole (global) batch; this rank's shard starts at byte first*length
        base_text = cxx_corpus_bytes()
        f = len(base_text)
        total = n * length
        reps = total // f + 3
        start = (first * length) % f
        tiled = torch.as_tensor(base_text, device=dev).repeat(reps)
        text = tiled[start:start + total].clone().view(n, length)
        del tiled
    else:
        text = torch.empty((n, stride), dtype=torch.uint8, device=dev)
        pire_amd.corpus_fill_device(text.data_ptr(), SEED, first, n, length, stride, plants, stream)
    run_n, run_len = (1, n * length) if args.one_string else (n, length)
    run_stride = run_len if args.one_string else stride
    out_idx = torch.empty(run_n, dtype=torch.int32, device=dev)
    out_fin = torch.empty(run_n, dtype=torch.uint8, device=dev)
    flags = pb.FLAG_BEGIN | pb.FLAG_END
    # The match counters ACCUMULATE on the device (out_counts is added to, never cleared by the library): one row per leg
    # -- the untimed passes of a leg into a row of their own -- and ONE all-reduce of the leg's row per fence (RCCL over
    # xGMI for N > 1).  The counters are sums, so the total over K passes reduced once is the sum of K reductions; a pass's
    # own counts are total / K (checked: the batch is the same every pass).  Round 4 reduced after every pass: on one rank
    # through RCCL that cost 31-44 us per 0.66 ms step (5-7 %), the all-reduce kernel waiting for a CU of a GPU whose every
    # CU's LDS the persistent scan holds (`--reduce-every-step` restores it for comparison).
    settle = max(0, args.settle)
    cold_steps = max(1, min(args.steps, 10))
    cold_launches = max(0, args.cold_launches)
    total_steps = 3 * (settle + args.warmup) + 2 * cold_steps + args.steps + cold_launches + 24
    counts_all = torch.zeros((total_steps + 16 if args.reduce_every_step else 16, table.RegexpsCount + 2), dtype=torch.int64,
                             device=dev)
    row_no = [0]
    last_row = [None]
    pending = []   # outstanding all-reduces, oldest first (--reduce-every-step)
    per_rank = []  # wall seconds of the timed leg, per rank
    leg_stats = {}

    def step(tbl, counts, ev=None):
        last_row[0] = counts
        if ev:
            ev[0].record()
        tbl.run_strided_device(text.data_ptr(), run_n, run_len, run_stride, flags, out_idx.data_ptr(),
                               out_fin.data_ptr(), counts.data_ptr(), 0, stream)
        if ev:
            ev[1].record()
        if args.reduce_every_step:
            while len(pending) > 2:
                pending.pop(0).wait()   # stream-level wait for the reduction issued two steps ago: long finished
            h = pd.allreduce_counts(counts, async_op=True)
            if h is not None:
                pending.append(h)

    def next_row():
        row_no[0] += 1
        row = counts_all[row_no[0] % counts_all.shape[0]]
        row.zero_()
        return row

    def fence():
        while pending:
            pending.pop(0).wait()
        pd.barrier()
        torch.cuda.synchronize()

    def timed(tbl, steps, warm):
        """`warm` untimed passes, then exactly `steps` timed ones between two fences; the timed passes' counters are
        all-reduced ONCE, inside the timed region, in front of its closing fence.  The events are made before the first
        fence: nothing but the fence itself (microseconds) separates the untimed passes from the timed ones, so the GPU
        does not fall idle in between (see `settle` below)."""
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        red = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        warm_row = next_row()
        for _ in range(warm):
            step(tbl, next_row() if args.reduce_every_step else warm_row)
        fence()
        row = next_row()
        rows = [next_row() for _ in range(steps)] if args.reduce_every_step else [row] * steps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(tbl, rows[k], events[k])
        if not args.reduce_every_step:
            red[0].record()
            pd.allreduce_counts(row)          # the path's only exchange: 80 B of counters, once per fence
            red[1].record()
        fence()
        mine = time.perf_counter() - t0
        per_rank[:] = pd.gather_over_ranks(mine, dev)      # this leg's wall time of every rank (the last leg's is reported)
        elapsed = pd.max_over_ranks(mine, dev)
        kms = [a.elapsed_time(b) for a, b in events]
        leg_stats.clear()
        leg_stats.update({"counts_row": last_row[0],
                          "passes_in_row": 1 if args.reduce_every_step else steps,
                          "reduce_ms": None if args.reduce_every_step else round(red[0].elapsed_time(red[1]), 4),
                          "kernel_ms_of_ranks": pd.gather_over_ranks(float(np.mean(kms)), dev)})
        return elapsed, kms

    # --- 1. the dense-row ranking is learned on a HELD-OUT corpus: same generator, another seed, a quarter of the size.
    # pire_hip_table_adapt() (shim: Table<Scanner>::Adapt()) re-ranks the LDS-resident rows from the visit counters the
    # scans left on the device; results never depend on the ranking.  The timed corpus is never seen by adapt().
    adapted_rows = 0
    heldout = None
    if not args.no_adapt:
        n2 = max(64, min(n, 1 << 18))
        if wide_entry:
            # a held-out base of 16 384 records, repeated (rotated) like the timed text: a working set of 10 000 states whose deep
            # ones carry 1e-5 of the steps each leaves a visit sample of every one of them only in a batch of this size (the
            # samples are one lane per wave and 128-byte tile, and one re-walk in 64) -- round 5 ranked such tables from 64 MB
            nb2 = min(n2, 16384)
            base2 = torch.as_tensor(W.wide_records(wide_entry, args.corpus, SEED_HELDOUT, nb2, length), device=dev)
            text2 = base2.index_select(0, torch.as_tensor(W.rotated_repeat_order(n2, nb2), device=dev)).contiguous()
            del base2
            heldout = f"token corpus {args.corpus!r}, seed {SEED_HELDOUT:#x}, {nb2} records repeated to {n2}"
        elif args.corpus == "cxx":
            # another cut of the same kind of text: the file reversed line by line would be another language; use the
            # text shifted by half a file and scanned as records of the same length
            f = len(base_text)
            t2 = torch.as_tensor(np.roll(base_text, f // 2 + 977), device=dev).repeat(n2 * length // f + 2)
            text2 = t2[:n2 * length].clone().view(n2, length)
            heldout = f"the same text shifted by {f // 2 + 977} bytes, {n2} records"
        else:
            text2 = torch.empty((n2, stride), dtype=torch.uint8, device=dev)
            pire_amd.corpus_fill_device(text2.data_ptr(), SEED_HELDOUT, 0, n2, length, stride, plants, stream)
            heldout = f"synthetic corpus, seed {SEED_HELDOUT:#x}, {n2} strings"
        n2r, l2r, s2r = (1, n2 * length, n2 * length) if args.one_string else (n2, length, stride)
        # (a wide table: a second round -- once adapt() has seen the scans leave the dense rows the library takes the
        # class-indexed walk, whose own visit samples then rank the states beyond the dense rows)
        for _ in range(4 if wide_entry else 1):
            for _ in range(3):
                table.run_strided_device(text2.data_ptr(), n2r, l2r, s2r, flags, out_idx.data_ptr(), out_fin.data_ptr(), 0, 0,
                                         stream)
            torch.cuda.synchronize()
            adapted_rows += table.adapt()
        del text2
    # --- 2. the table as pire_hip_table_create ranks it (a-priori byte model, adaptation switched OFF: auto_adapt = 1): a second
    # handle, a short timed leg on the timed corpus, reported as value_ranking_frozen (rounds 1-5 called it value_before_adapt).
    # `settle`: an MI355X that has been idle for more than ~1 ms restarts its power management transient -- two or three
    # launches at boost clocks, a dip to 0.8-0.9 ms per launch, recovery after 20-30 launches (~25 ms); profiles/
    # r03_warmup_curve.log.  Every timed leg is therefore preceded by `settle` untimed passes (default 60, 40 ms of
    # scanning) issued back to back with the W warm-up passes: the metric is the throughput of sustained scanning.
    cold_table = pire_amd.Table(blob)
    cold_table.upload()
    cold_elapsed, cold_ms = timed(cold_table, cold_steps, settle + args.warmup)
    del cold_table
    # --- 2b. (round 6) the same for a caller that only ENQUEUES and never calls adapt(): a third handle under the library's default
    # policy (auto_adapt = 0: such calls start a re-ranking in the background and swap it in at a later launch boundary; the walk
    # follows the trap signal meanwhile) -- a handful of passes on the timed corpus with nothing but the caller's own stream
    # synchronisation between them, then a short timed leg.  Reported as value_enqueue_only_no_adapt.
    enq = None
    if not args.no_adapt and not args.one_string:
        pb.set_config(auto_adapt=0)
        enq_table = pire_amd.Table(blob)
        enq_table.upload()
        enq_row = next_row()
        passes = 0
        for passes in range(1, 13):
            step(enq_table, enq_row)
            torch.cuda.synchronize()
            time.sleep(0.03)
            if enq_table.refresh_info().adaptations >= 2:
                break
        enq_elapsed, enq_ms = timed(enq_table, cold_steps, settle + args.warmup)
        ei = enq_table.refresh_info()
        enq = {"value": round(float(n) * length * cold_steps * world / enq_elapsed / 1e9, 2), "kernel": pb.last_kernel_symbol(),
               "kernel_avg_ms": round(float(np.mean(enq_ms)), 4), "passes_before_the_timed_leg": passes,
               "adaptations_in_the_background": int(ei.adaptations), "explicit_adapt_calls": 0,
               "what": "a fresh table, pire_hip_config defaults, PIRE_HIP_RUN_ON_DEVICE calls only; between the passes the caller "
                       "synchronises its own stream and sleeps 30 ms"}
        del enq_table
        pb.set_config(auto_adapt=1)
    # --- 3. the timed region: settle + W warm-up passes, fence, exactly K passes, fence
    elapsed, kernel_ms = timed(table, args.steps, settle + args.warmup)
    per_rank_timed = list(per_rank)   # of THIS leg (the from-idle leg below overwrites per_rank)
    kernel_name = pb.last_kernel_symbol()   # the instantiation the library actually launched

    timed_stats = dict(leg_stats)
    total_counts = timed_stats["counts_row"].cpu().numpy().astype(np.uint64)   # the timed passes', reduced over the ranks
    assert (total_counts % timed_stats["passes_in_row"] == 0).all(), "the same batch every pass: the totals are multiples of the pass count"
    total_counts //= timed_stats["passes_in_row"]   # one pass's
    # --- 4. from idle: what a caller gets who scans one batch on a GPU that was doing nothing.  Behind the
    # timed region and outside it: the device is drained, left idle for 0.3 s (its clocks drop), then `cold_launches`
    # passes are timed back to back with no warm-up at all (the first ones run in the power management's transient,
    # profiles/r03_warmup_curve.log).  Reported beside the sustained `value`, never instead of it.
    from_idle = None
    if cold_launches:
        fence()
        time.sleep(args.cold_idle_ms / 1e3)
        cold_el, cold_start_ms = timed(table, cold_launches, 0)
        from_idle = (cold_el, cold_start_ms)
    gpu_idx = out_idx.cpu().numpy().astype(np.uint32)
    gpu_fin = out_fin.cpu().numpy()

    if rank == 0:
        scanned = float(n) * length * args.steps * world
        frozen_value = round(float(n) * length * cold_steps * world / cold_elapsed / 1e9, 2)
        value = scanned / elapsed / 1e9
        algo_bytes = n * length + 5 * run_n                # input read once + u32 state idx + u8 final per string
        avg_ms = float(np.mean(kernel_ms))
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        info = table.refresh_info()
        # HBM traffic per launch from the committed PMC passes (collected separately, as rocprofv3 requires)
        traffic = traffic_src = lds = traffic_note = None
        sources = kernel_sources_sha16()
        for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    pmc = json.load(f)
                if pmc.get("workload") != f"{args.set} 2^{args.log2_strings} x {length}" or args.strings or args.corpus != "synthetic":
                    continue
                if pmc.get("kernel_sources_sha16") != sources:
                    # counters of another build of the kernel say nothing about this one
                    traffic_note = (f"profiles/{name} was collected with kernel sources {pmc.get('kernel_sources_sha16', 'unknown')}, "
                                    f"these are {sources}: refused (re-run the --pmc passes, tools/make_pmc_json.py)")
                    continue
                traffic, traffic_src = pmc["hbm_bytes_per_launch"], name
                if "lds_idx_active_cycles_per_launch" in pmc:
                    # the secondary bound SURVEY 8(d) asks for: the dependent LDS gather (one ds_read_u8 per byte)
                    lds = {"bank_conflict_over_idx_active": round(pmc["lds_bank_conflict_cycles_per_launch"] /
                                                                  pmc["lds_idx_active_cycles_per_launch"], 3),
                           "lds_cycles_per_lookup": round(pmc["lds_idx_active_cycles_per_launch"] /
                                                          pmc["lds_instructions_per_launch"], 2),
                           "source": f"profiles/{name} (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS)"}
                break
            except (OSError, ValueError, KeyError):
                pass
        if wide_entry:
            data = (f"token corpus {args.corpus!r} of {args.set} (pire_amd/workloads.py wide_records, seed {SEED:#x}): "
                    f"{wide_base.shape[0]} distinct records repeated over the batch")
            shape = f"{'2^%d' % args.log2_strings if not args.strings else n} x {length} B records per GPU"
        elif args.corpus == "cxx":
            data = (f"C++ source text: the reference's tools/bench/test_file ({len(base_text)} bytes) repeated to the batch size, "
                    "as tools/bench/run-bench:126-138 builds its big file")
            shape = f"ONE string of {n * length} B" if args.one_string else f"{n} x {length} B records"
        else:
            data = "synthetic"
            shape = f"{'2^%d' % args.log2_strings if not args.strings else n} x {length} B strings per GPU"
        res = {
            "metric": "scanned GB/s (whole node) + ns/byte, 8-regex glued Scanner, 4KiB strings",
            "value": round(value, 2),
            "unit": "GB/s",
            "ns_per_byte": round(1.0 / value, 6),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle": settle,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": data,
            # (round 6) "before adapt()" = what a caller gets who never calls pire_hip_table_adapt(): the library's default policy,
            # calls that only enqueue (leg 2b).  The table whose ranking is FROZEN as created (auto_adapt = 1, leg 2) beside it.
            "value_before_adapt": enq["value"] if enq else frozen_value,
            "value_before_adapt_is": ("value_enqueue_only_no_adapt: a fresh table, library defaults, no adapt() call, no device-wide "
                                      "synchronisation" if enq else "value_ranking_frozen (this run made no enqueue-only leg)"),
            "value_enqueue_only_no_adapt": enq["value"] if enq else None,
            "value_ranking_frozen": frozen_value,
            "enqueue_only_no_adapt": enq,
            "config": {
                "workload": f"{WORKLOADS.get(args.set, args.set)} ({args.set}), {shape}, "
                            f"Begin().Run().End() per string, match counters accumulated on the device and reduced "
                            f"{'after every pass' if args.reduce_every_step else 'once per timed leg'}",
                "patterns": big["patterns"],
                "walk": {"variant_asked": args.walk, "zip_variant_asked": args.zip, "wide_rows": info.wide_states,
                         "states_with_a_row_of_their_own": info.zip_full_states or info.wide_states,
                         "image": "zipped (header + <= 3 exceptions for the states without a row)" if info.zip_full_states else "plain rows",
                         "plan_share_outside_plain_rows": round(float(info.zip_plain_outside_share), 6),
                         "plan_share_outside_zipped_tier": round(float(info.zip_outside_share), 6),
                         "wide_lds_bytes": info.wide_lds_bytes,
                         "measured_share_outside_dense_rows": round(float(info.outside_dense_share), 6),
                         "measured_share_outside_wide_rows": round(float(info.outside_wide_share), 6),
                         "shares_measured": bool(info.shares_measured)},
                "table": {"states": info.states, "letters": info.letters, "regexps": info.regexps,
                          "ref_buf_bytes": int(info.ref_buf_size), "lds_dense_rows": info.hot_states,
                          "lds_table_bytes": info.lds_table_bytes, "rows_promoted_by_adapt": adapted_rows,
                          "ranking_learned_on": heldout},
                "strings_per_gpu": run_n, "string_bytes": run_len, "string_stride": run_stride, "corpus_seed": SEED,
                "parallelism": f"shard-by-string x{world}",
                "ranks_seen": len(rank_devices), "per_rank_device": rank_devices, "ranks_sharing_a_device": clashes,
                "shard_ranges": [list(pd.shard_range(n * world, r, world)) for r in range(world)],
                "reduce_backend": pd.backend_description(), "rccl_version": pd.rccl_version(),
                "per_rank_GBps": [round(float(n) * length * args.steps / t / 1e9, 1) for t in per_rank_timed],
                "per_rank_kernel_ms": [round(x, 4) for x in timed_stats["kernel_ms_of_ranks"]],
                "counter_reduce": ("after every pass, asynchronously (round 4's form)" if args.reduce_every_step else
                                   "once per fence: the counters accumulate on the device over the timed passes"),
                "counter_reduce_ms": timed_stats["reduce_ms"],
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": f"profiles/{traffic_src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, same kernel sources "
                                  f"{sources})" if traffic else traffic_note,
                "lds_gather": lds,
                "kernel": kernel_name, "kernel_avg_ms": round(avg_ms, 4),
                "kernel_min_ms": round(float(np.min(kernel_ms)), 4),
                "kernel_avg_ms_ranking_frozen": round(float(np.mean(cold_ms)), 4),
                "algorithmic_bytes_per_launch": algo_bytes,
                "frac_of_measured_copy_ceiling_6290": round(achieved / 6290.0, 4),
            },
            "match_counts": {"final": int(total_counts[0]), "strings": int(total_counts[1]),
                             "per_regexp": [int(c) for c in total_counts[2:]]},
        }
        # how often the timed corpus left the dense rows: the table's cold-state samples since the ranking was frozen (one
        # rotating lane of 64 per chunk that ended outside the dense rows), read by an adapt() AFTER every
        # timed leg -- it re-ranks a table nobody uses any more
        launches_since_ranking = settle + args.warmup + args.steps + (cold_launches if from_idle else 0)
        table.adapt()
        samples = int(table.refresh_info().last_trap_samples)
        wide_chunks = int(table.info.last_wide_trap_chunks)
        res["traps"] = {"cold_samples": samples, "launches": launches_since_ranking,
                        "wide_walk_wave_chunk_share_walked_twice": round(wide_chunks / max(1.0, launches_since_ranking * float(n) * length / 1024.0), 6),
                        "cold_lane_chunk_share": round(samples * 64.0 / max(1.0, launches_since_ranking * float(n) * length / 16.0), 8),
                        "what": "share of (lane, 16-byte chunk) pairs of the timed corpus that ended in a state without a dense row"}
        if from_idle:
            cel, cms = from_idle
            cold_achieved = algo_bytes / (float(np.mean(cms)) * 1e-3) / 1e9
            res["cold_start"] = {
                "what": f"{cold_launches} passes back to back after {args.cold_idle_ms} ms of idle GPU, no warm-up, adapted table; "
                        "outside the timed region",
                "value": round(float(n) * length * cold_launches * world / cel / 1e9, 2), "unit": "GB/s",
                "ms_per_step": round(cel / cold_launches * 1e3, 4),
                "kernel_avg_ms": round(float(np.mean(cms)), 4), "kernel_first_ms": round(float(cms[0]), 4),
                "kernel_max_ms": round(float(np.max(cms)), 4),
                "frac": round(cold_achieved / HBM_PEAK_GBS, 4)}
        assert int(total_counts[1]) == run_n * world, "match-count reduce lost strings"
        if not args.no_cpu and world == 1 and not args.one_string:   # the reported CPU baseline belongs to the N=1 line
            sample = min(n, 1 << args.cpu_sample_log2)
            if wide_entry:
                sample = min(sample, wide_base.shape[0])
                host = wide_base[:sample]
                # every string of the batch: the repeats of a base record must all have ended where the first one did
                nb = wide_base.shape[0]
                res["parity_of_repeats"] = bool((gpu_idx == gpu_idx[:nb][(wide_order - first) % nb]).all() and (gpu_fin == gpu_fin[:nb][(wide_order - first) % nb]).all())
            elif args.corpus == "cxx":
                f = len(base_text)
                host = np.resize(base_text, sample * length + f)[:sample * length].reshape(sample, length)
            else:
                from oracle import binding as ob   # host twin of the corpus generator (oracle/corpus.c)

                host = ob.corpus_fill(SEED, 0, sample, length, plants, threads=min(os.cpu_count() or 1, 256))
            res["cpu_baseline"] = cpu_baseline(blob, host, length, gpu_idx, gpu_fin)
            if wide_entry:
                from oracle import binding as ob   # the checker's visit counts: how many states this corpus walks through

                k = min(256, host.shape[0])
                v = np.sort(ob.OracleScanner(blob).visit_counts(host[:k].reshape(-1), np.arange(k + 1, dtype=np.uint64) * length))[::-1]
                cum = np.cumsum(v.astype(np.float64)) / max(float(v.sum()), 1.0)
                res["working_set"] = {"distinct_states_visited": int((v > 0).sum()), "sample": f"first {k} records",
                                      "share_of_steps_outside_the_255_most_visited": round(1.0 - float(cum[min(254, len(cum) - 1)]), 6),
                                      "share_of_steps_outside_the_wide_rows_most_visited":
                                          round(1.0 - float(cum[min(max(info.wide_states, 1) - 1, len(cum) - 1)]), 6)}
        print(json.dumps(res))
        sys.stdout.flush()
    if world > 1 or args.force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
