#!/usr/bin/env python3
"""Throughput against working-set size (round 5): dictionary scanners and token-mixture corpora through
pire_hip_run_strided with the dense rows (walk_variant 1), the class-indexed walk (2) and the library's own choice (0).

    python tools/wide_case.py [--log2-strings 18] [--len 4096] [--points set:corpus,...] [--out file.jsonl]

One JSON line per point: table geometry, distinct states the corpus visits (oracle, on a sample), the library's measured
shares outside the dense / wide rows, GB/s of each walk at settled clocks, trap shares, and parity of EVERY string of the
batch against the oracle.  Blacklist scanners (anchored, the sample's own wrapping) are measured on URL batches through
pire_hip_run.  The oracle is the checker here, never the thing measured."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from pire_amd import workloads as W

DEFAULT = ("set_b_mix:mix,dict_1k:k32,dict_1k:k128,dict_1k:k512,dict_1k:k1000,dict_10k:k32,dict_10k:k512,dict_10k:k2048,"
           "dict_10k:k10000,blacklist_1k:urls,blacklist_10k:urls")


def timed(launch, total_bytes, reps):
    settle = max(20, int(40.0 / max(total_bytes / 2.5e9, 0.05)))
    for _ in range(settle):
        launch()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        launch()
        b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in ev]
    return float(np.mean(ts)), float(np.min(ts))


def working_set(o, text, offs, wide):
    v = np.sort(o.visit_counts(text, offs))[::-1].astype(np.float64)
    cum = np.cumsum(v) / max(v.sum(), 1.0)
    return {"distinct_states_visited_in_sample": int((v > 0).sum()),
            "ideal_share_outside_255_rows": round(1.0 - float(cum[min(254, len(cum) - 1)]), 6),
            "ideal_share_outside_wide_rows": round(1.0 - float(cum[min(max(wide, 1) - 1, len(cum) - 1)]), 6)}


def point_records(entry, corpus, n, length, reps):
    blob = W.load_blob(entry["blob"])
    o = ob.OracleScanner(blob)
    nbase = min(n, 16384)   # (as bench.py: with 4 096 two waves of a CU could meet part of the same records, tests/test_workloads.py)
    base = W.wide_records(entry, corpus, 0x5EED5EED, nbase, length)
    offs = np.arange(nbase + 1, dtype=np.uint64) * length
    oi, of = o.run(base.reshape(-1), offs, threads=8)
    # (every repeat of the base rotated by its own number of records: a plain repeat has a period of 64 tasks, and a kernel
    # whose block b hands task b + 256 w to its wave w then walks the same 64 records in all waves of a CU -- their table
    # loads hit each other's lines in the L1)
    order = W.rotated_repeat_order(n, nbase)
    text = torch.as_tensor(base, device="cuda").index_select(0, torch.as_tensor(order, device="cuda")).contiguous()
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    t = pire_amd.Table(blob)
    t.upload()
    info = t.info
    res = {"set": entry["name"], "corpus": corpus, "strings": n, "string_bytes": length, "states": info.states, "letters": info.letters,
           "dense_rows": info.hot_states, "wide_rows": info.wide_states, "wide_lds_bytes": info.wide_lds_bytes}
    res.update(working_set(o, base[:1024].reshape(-1), offs[:1025], info.wide_states))

    def launch():
        t.run_strided_device(text.data_ptr(), n, length, length, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)

    def parity():
        gi = idx.cpu().numpy().astype(np.uint32)
        gf = fin.cpu().numpy()
        return bool((gi == oi[order]).all() and (gf == of[order]).all())

    total = n * length
    only = set(os.environ.get("WIDE_CASE_LEGS", "dense,wide,wide2,zip,zip2,auto").split(","))
    for variant, zipv, label in ((1, 1, "dense"), (2, 1, "wide"), (3, 1, "wide2"), (2, 2, "zip"), (3, 2, "zip2"), (0, 0, "auto")):
        if label not in only:
            continue
        pb.set_config(walk_variant=variant, zip_variant=zipv, auto_adapt=1)
        for _ in range(4):   # the ranking learned from the batch itself, with the walk that is measured
            launch()
            torch.cuda.synchronize()
            t.adapt()
        mean, best = timed(launch, total, reps)
        kernel = pb.last_kernel()
        ok = parity()
        t.adapt()
        i2 = t.refresh_info()
        launches = max(20, int(40.0 / max(total / 2.5e9, 0.05))) + reps
        res[label] = {"kernel": kernel, "GBps": round(total / mean / 1e6, 1), "GBps_best": round(total / best / 1e6, 1),
                      "ms": round(mean, 4), "parity_all_strings": ok,
                      "trap_samples": int(i2.last_trap_samples),
                      "wave_chunk_share_walked_twice_by_the_wide_walk": round(i2.last_wide_trap_chunks / max(1.0, launches * total / 1024.0), 6),
                      "measured_share_outside_dense_rows": round(float(i2.outside_dense_share), 6),
                      "measured_share_outside_wide_rows": round(float(i2.outside_wide_share), 6),
                      "tier_states": int(i2.wide_states), "states_with_a_row": int(i2.zip_full_states) or int(i2.wide_states),
                      "plan_share_outside_plain_rows": round(float(i2.zip_plain_outside_share), 6),
                      "plan_share_outside_zipped_tier": round(float(i2.zip_outside_share), 6),
                      "symbol": pb.last_kernel_symbol()}
    return res


def point_urls(entry, n, reps):
    blob = W.load_blob(entry["blob"])
    o = ob.OracleScanner(blob)
    nbase = min(n, 1 << 16)
    btext, boffs = W.wide_urls(entry, 0x5EED5EED, nbase)
    oi, of = o.run(btext, boffs, threads=8)
    rep = n // nbase
    lens = np.tile(np.diff(boffs), rep)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    text = torch.as_tensor(btext, device="cuda").repeat(rep).contiguous()
    doffs = torch.as_tensor(offs.astype(np.int64), device="cuda")
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    t = pire_amd.Table(blob)
    t.upload()
    info = t.info
    total = int(offs[-1])
    res = {"set": entry["name"], "corpus": "urls", "strings": n, "GiB": round(total / 2**30, 3), "states": info.states,
           "letters": info.letters, "dense_rows": info.hot_states, "wide_rows": info.wide_states}
    res.update(working_set(o, btext[:int(boffs[4096])], boffs[:4097], info.wide_states))

    def launch():
        t.run_device(text.data_ptr(), doffs.data_ptr(), n, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)

    # (*_stream: ragged_variant = 2, the stream kernel on the same walk -- opt-in; the others: the library's routing)
    only = set(os.environ.get("WIDE_CASE_LEGS", "dense,wide,wide_stream,zip,zip_stream,auto").split(","))
    for variant, zipv, raggedv, label in ((1, 1, 0, "dense"), (2, 1, 0, "wide"), (2, 1, 2, "wide_stream"), (2, 2, 0, "zip"),
                                          (2, 2, 2, "zip_stream"), (0, 0, 0, "auto")):
        if label not in only:
            continue
        pb.set_config(walk_variant=variant, zip_variant=zipv, ragged_variant=raggedv, auto_adapt=1)
        for _ in range(3):
            launch()
            torch.cuda.synchronize()
            t.adapt()
        mean, best = timed(launch, total, reps)
        try:   # tuning build (PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_STREAM_CLOCKS=1): the stream kernel's stage clocks
            import ctypes as C
            out = (C.c_double * 8)()
            if pb.lib().pire_hip_debug_stream_clocks(out) == 0:
                print("%s stream clocks, us per wave: search %.2f | table copy %.2f | positions + lane search %.2f | window loop %.2f | "
                      "flush %.2f | whole wave %.2f ; %.1f phases per wave, %d wave-launches" % (label, *out[:7], int(out[7])), flush=True)
        except AttributeError:
            pass
        gi = idx.cpu().numpy().astype(np.uint32).reshape(rep, nbase)
        gf = fin.cpu().numpy().reshape(rep, nbase)
        kernel = pb.last_kernel()
        t.adapt()
        i2 = t.refresh_info()
        res[label] = {"kernel": kernel, "GBps": round(total / mean / 1e6, 1), "GBps_best": round(total / best / 1e6, 1),
                      "ms": round(mean, 4), "parity_all_strings": bool((gi == oi[None, :]).all() and (gf == of[None, :]).all()),
                      "measured_share_outside_dense_rows": round(float(i2.outside_dense_share), 6),
                      "measured_share_outside_wide_rows": round(float(i2.outside_wide_share), 6),
                      "tier_states": int(i2.wide_states), "states_with_a_row": int(i2.zip_full_states) or int(i2.wide_states),
                      "symbol": pb.last_kernel_symbol()}
    res["listed_share"] = round(float(of.mean()), 4)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-strings", type=int, default=18)
    ap.add_argument("--len", type=int, default=4096)
    ap.add_argument("--log2-urls", type=int, default=23)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--points", default=DEFAULT)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    out = open(args.out, "w") if args.out else None
    for spec in args.points.split(","):
        name, corpus = spec.split(":")
        entry = W.wide_set(name)
        t0 = time.time()
        if corpus == "urls":
            res = point_urls(entry, 1 << args.log2_urls, args.reps)
        else:
            res = point_records(entry, corpus, 1 << args.log2_strings, args.len, args.reps)
        res["seconds"] = round(time.time() - t0, 1)
        line = json.dumps(res)
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main())
