#!/usr/bin/env python3
"""Regenerates the measurement tables of DESIGN.md sections 5.1-5.4 (between their BEGIN / END markers) from the
profiles/r06_* files that tools/collect_final_r06.sh leaves: no number in those tables is typed by hand."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles") + "/"


def J(n):
    return json.load(open(P + n))


def headline():
    h, hd = J("r06_bench_n1.json"), J("r06_bench_n1_defaults.json")
    tr = open(P + "r06_bench_trace_timed_region.txt").read()
    m = re.search(r"the last 20 \(the timed region\): avg ([\d.]+) ms, min ([\d.]+), max ([\d.]+)", tr)
    rf, cs, cb = h["roofline"], h["cold_start"], h["cpu_baseline"]
    fd = J("r06_bench_force_dist_nccl.json")
    fe = J("r06_bench_force_dist_nccl_every_step.json")
    rb = fd.get("reduce_backend") or fd.get("config", {}).get("reduce_backend", "?")
    alg = rf["algorithmic_bytes_per_launch"]
    return f"""| | value | source |
|---|---|---|
| `python bench.py --steps 20 --warmup 5` (the driver's command) | **{h['value']:.0f} GB/s**, {h['ms_per_step']:.3f} ms per step; no `adapt()` call, enqueue-only {h['value_before_adapt']:.0f}, ranking frozen as created {h.get('value_ranking_frozen', h['value_before_adapt']):.0f} | `profiles/r06_bench_n1.json` |
| `python bench.py` (defaults: 50 steps) | {hd['value']:.0f} GB/s | `r06_bench_n1_defaults.json` |
| kernel `ScanTiledKernel<16,2,nt,5>`, HIP events in `bench.py` | avg {rf['kernel_avg_ms']:.4f} ms (min {rf['kernel_min_ms']:.4f}) → {rf['achieved']:.0f} GB/s algorithmic = **{rf['frac']:.3f} of 8 TB/s** ({rf['frac_of_measured_copy_ceiling_6290']:.3f} × the measured copy ceiling of the part) | same |
| the same 20 launches in `rocprofv3 --kernel-trace --stats` | avg {m.group(1)} ms (min {m.group(2)}, max {m.group(3)}) → {alg / float(m.group(1)) / 1e6:.0f} GB/s = {alg / float(m.group(1)) / 1e6 / 8000:.3f} | `r06_bench_kernel_stats.csv`, `r06_bench_trace_timed_region.txt` |
| HBM traffic per launch (PMC, separate passes, gfx950 corrections) | {rf['traffic'] / 1e9:.3f} GB = {rf['traffic'] / alg:.3f} × algorithmic ({alg / 1e9:.3f} GB: 4 096 B of text + 5 B of results per string) | `r06_pmc_traffic.json`, `r06_bench_pmc_summary.txt` |
| LDS gather | {rf['lds_gather']['lds_cycles_per_lookup']} LDS cycles per lookup instruction, conflicts / active = {rf['lds_gather']['bank_conflict_over_idx_active']} | same |
| dense rows left (trap samples) | {h['traps']['cold_samples']} in {h['traps']['launches']} launches | `r06_bench_n1.json` `traps` |
| from idle: 20 launches after 300 ms of idle GPU | {cs['value']:.0f} GB/s ({cs['frac']:.2f}); first launch {cs['kernel_first_ms']:.3f} ms, slowest {cs['kernel_max_ms']:.3f} | `r06_bench_n1.json` `cold_start` |
| reference `Pire::Scanner` on the box's host cores, whole batch, parity with the GPU **{cb['parity_vs_gpu']}** | {cb['runs']['scanner_1t']['GBps']:.2f} GB/s on 1 core, {cb['value']:.1f} on {cb['cores']} | `r06_bench_n1.json` `cpu_baseline` |
| one rank through RCCL (`--force-dist --backend nccl`), counters reduced once per fence | {fd['value']:.0f} GB/s, {fd['ms_per_step']:.4f} ms per step ({(fd['ms_per_step'] / h['ms_per_step'] - 1) * 100:+.1f} % against the line without a process group); the reduce itself {fd['config']['counter_reduce_ms']} ms per fence; `reduce_backend` = "{str(rb)[:11]}" | `r06_bench_force_dist_nccl.json` |
| ... reduced after every pass (round 4's form) | {fe['value']:.0f} GB/s, {fe['ms_per_step']:.4f} ms per step ({(fe['ms_per_step'] / h['ms_per_step'] - 1) * 100:+.1f} %) | `r06_bench_force_dist_nccl_every_step.json` |"""


def ragged():
    cur = {}
    for line in open(P + "r06_ragged_cases.log"):
        mm = re.match(r"variant=(\d): (\w+) (\w+): (\d+) strings, ([\d.]+) GiB.*mean ([\d.]+) ms -> ([\d.]+) GB/s", line)
        if mm:
            v, k, c, n, g, ms, gb = mm.groups()
            cur.setdefault(c, {})[v] = (k, float(gb), float(ms), n, g)
    desc = {"urls": "URLs 20–199 B", "loglines": "log lines 64–1 023 B", "uniform2k": "0–2 047 B", "uniform8k": "0–8 191 B",
            "fixed4096": "4 096 B each", "urls_x4": "URLs × 4", "loglines_x4": "log lines × 4", "uniform2k_x4": "0–2 047 B × 4"}
    out = "| batch (`set_a`, device pointers) | strings / GiB | ragged kernel (variant 1) | default routing | |\n|---|---|---|---|---|\n"
    for c, d in cur.items():
        a, b = d.get("1"), d.get("0")
        out += f"| {desc.get(c, c)} | 2^{int(a[3]).bit_length() - 1} / {a[4]} | {a[1]:.0f} GB/s | **{b[1]:.0f}** ({b[0]}) | {b[1] / a[1]:.2f} × |\n"
    u, l, l4 = cur["urls"]["0"][1], cur["loglines"]["0"][1], cur["loglines_x4"]["0"][1]
    out += f"""
Unchanged kernels (round 4's), re-measured at HEAD.  Batches the host knows to be below 160 MiB, and device-offset batches of fewer
than 2^20 strings, keep the ragged kernel; 0–8 191 B × 2^17 leaves half of the lanes without a string (section 7).  PMC of both kernels:
`r06_ragged_pmc_{{urls,loglines}}.txt`, `r06_stream_pmc_{{urls,loglines}}.txt`."""
    return out


def configs():
    def line(name, what, f):
        d = J(f)
        c, r = d.get("cpu_baseline", {}), d.get("roofline", {})
        return f"| {name} | {what} | {d['value']:.0f} GB/s | {r.get('frac')} ({r.get('bound')}) | {c.get('parity_vs_gpu')} ({c.get('sample', '')[:40]}…) | `{f}` |\n"
    out = "| config | workload | value | roofline frac | parity vs the reference | file |\n|---|---|---|---|---|---|\n"
    out += line("C2", "1 pattern, 2^20 × 4 KiB", "r06_bench_c2.json")
    out += line("C4 shard", "8 patterns, 2^23 × 4 KiB (32 GiB)", "r06_bench_c4_shard.json")
    out += line("C5a", "8 URL-classifier patterns, 2^20 × 16 KiB", "r06_bench_c5a.json")
    out += line("C5b", "`SlowScanner` x.{40}$ UTF-8, 2^20 × 16 KiB", "r06_bench_c5b.json")
    out += line("set_d", "8 unanchored `pire_ut.cpp` patterns, 2^20 × 4 KiB", "r06_bench_set_d.json")
    out += line("C++ text", "the reference's benchmark corpus as 4 KiB records", "r06_bench_cxx_records.json")
    out += line("C++ text, one string", "… as ONE 1 GiB string (segmented scan)", "r06_bench_cxx_one_string.json")
    c1 = J("r06_bench_c1_nonreloc.json")
    out += f"| C1 | `NonrelocScanner` `hello\\s+w.+d$`, 10 000 × 256 B | reference CPU `Run()` {c1['value']:.2f} GB/s; the GPU leg is in the same line | — | see file | `r06_bench_c1_nonreloc.json` |\n"
    tests = open(P + "r06_final_pytest_gpu.log").read()
    passed = re.search(r"(\d+) passed", tests)
    out += f"""
`set_d`: {J('r06_bench_set_d.json')['roofline']['frac']:.2f} -- what the LDS allows for its addresses: section 5.5.  The multi-rank launch path on the one GPU (gloo): 8 ranks
`r06_bench_8ranks_gloo.json`, config C4's shape on 2 ranks `r06_bench_c4_2ranks_gloo.json`.  Secondary kernels at HEAD:
`r06_final_{{prefix,half_final,counting,capture,actions,long_strings,pair,host_call_latency,counting_variants,capture_variants,
half_final_variants,slow_ragged}}.log`, `r06_counting_kernel_stats.txt`; the whole run: `r06_final_run.log` (`r06_final_pytest_gpu.log`:
{passed.group(1) if passed else '?'} GPU tests passed at HEAD)."""
    return out


def wide():
    """Section 5.4: throughput against working-set size (tools/wide_case.py, 2^20 x 4 KiB records / 2^23 URLs)."""
    out = ("| table (states x letters) | corpus | states visited (sample) | steps outside 255 rows / outside the 'plain' rows of the class-indexed walk (ideal ranking) | "
           "dense rows | **plain rows** (two strings per lane) | **zipped image**, one / two strings per lane | zipped tier: states (with a row of their own), measured share of the steps outside it | library's choice |\n|---|---|---|---|---|---|---|---|---|\n")
    for line in open(P + "r06_wide_curve.jsonl"):
        d = json.loads(line)
        shape = f"{d['strings']:,} URLs, {d['GiB']} GiB".replace(",", " ") if "GiB" in d else "2^20 × 4 KiB"
        plain = d.get("wide2") or d.get("wide")
        z1, z2 = d.get("zip"), d.get("zip2")
        zz = z2 or z1
        a = d["auto"]
        out += (f"| `{d['set']}` ({d['states']} × {d['letters']}) | `{d['corpus']}`, {shape} | {d['distinct_states_visited_in_sample']} | "
                f"{d['ideal_share_outside_255_rows'] * 100:.1f} % / {d['ideal_share_outside_wide_rows'] * 100:.1f} % | {d['dense']['GBps']:.0f} GB/s ({d['dense']['kernel']}) | "
                f"**{plain['GBps']:.0f}** ({plain['tier_states']} rows, {plain['measured_share_outside_wide_rows'] * 100:.1f} % outside) | "
                f"**{z1['GBps']:.0f}**{(' / **%.0f**' % z2['GBps']) if z2 else ''} | {zz['tier_states']} ({zz['states_with_a_row']}), {zz['measured_share_outside_wide_rows'] * 100:.2f} % | "
                f"{a['kernel']} {a['GBps']:.0f} ({a.get('symbol', '').split('::')[-1]}) |\n")
    out += ("\nEvery string of every batch equal to the oracle's answer (`parity_all_strings` in `profiles/r06_wide_curve.jsonl`).  `bench.py` lines (ranking learned on a held-out "
            "corpus of the batch's size, CPU baseline = the reference on all cores, parity of the whole batch; `enqueue only` = a fresh table, no `adapt()` call, calls that only enqueue):\n\n")
    out += "| `bench.py --set … --corpus …` | value | kernel | roofline frac (HBM) | tier: states (rows), share outside | ranking frozen as created (`auto_adapt = 1`) | `value_before_adapt`: enqueue only, no `adapt()` | reference on the host cores | file |\n|---|---|---|---|---|---|---|---|---|\n"
    for f, what in (("set_b_mix_mix", "set_b_mix mix"), ("dict_1k_k32", "dict_1k k32"), ("dict_1k_k128", "dict_1k k128"), ("dict_1k_k512", "dict_1k k512"),
                    ("dict_1k_k1000", "dict_1k k1000"), ("dict_10k_k32", "dict_10k k32"), ("dict_10k_k512", "dict_10k k512"), ("dict_10k_k2048", "dict_10k k2048"),
                    ("dict_10k_k10000", "dict_10k k10000"), ("c5_dict_10k_16k", "dict_10k k10000 --len 16384 (C5's shape: 2^20 × 16 KiB)"),
                    ("dict_utf8_1k_k32", "dict_utf8_1k k32 (113 letter classes)"), ("dict_utf8_1k_k1000", "dict_utf8_1k k1000"),
                    ("dict_utf8_5k_k512", "dict_utf8_5k k512"), ("dict_utf8_5k_k5000", "dict_utf8_5k k5000"),
                    ("dict_10k_k10000_plain_rows", "dict_10k k10000 --zip 1 (plain rows forced: round 5's walk)"),
                    ("dict_1k_k1000_plain_rows", "dict_1k k1000 --zip 1"),
                    ("dict_1k_k128_dense_rows", "dict_1k k128 --walk 1 (dense rows forced)")):
        d = J(f"r06_bench_{f}.json")
        c = d.get("cpu_baseline", {})
        w = d["config"]["walk"]
        e = d.get("enqueue_only_no_adapt") or {}
        out += (f"| {what} | **{d['value']:.0f} GB/s** | `{d['roofline']['kernel'].split('::')[-1]}` | {d['roofline']['frac']:.3f} | "
                f"{w['wide_rows']} ({w['states_with_a_row_of_their_own']}), {w['measured_share_outside_wide_rows'] * 100:.2f} % | {d.get('value_ranking_frozen', d['value_before_adapt']):.0f} | "
                f"{('%.0f after %d passes' % (e['value'], e['passes_before_the_timed_leg'])) if e else '—'} | "
                f"{('%.1f GB/s on %d cores, parity %s' % (c['value'], c['cores'], c['parity_vs_gpu'])) if c else '—'} | `r06_bench_{f}.json` |\n")
    return out


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    for name, fn in (("headline", headline), ("ragged", ragged), ("configs", configs), ("wide", wide)):
        a = s.index(f"<!-- BEGIN:{name}")
        a = s.index("\n", a) + 1
        b = s.index(f"<!-- END:{name} -->")
        s = s[:a] + fn().strip("\n") + "\n" + s[b:]
    open(path, "w").write(s)
    print("DESIGN.md: %d bytes" % len(s))


if __name__ == "__main__":
    main()
