#!/usr/bin/env python3
"""Walks with actions on tables whose scans visit thousands of states: pire_hip_prefix (LongestPrefix /
ShortestPrefix, run.h:277-311) and pire_hip_run_half_final (half_final.h:137-164) on a blacklist scanner's URL batch and on a
dictionary scanner's log lines, device pointers -- the dense rows (walk_variant = 1: what these entry points took until round 6)
against the class-indexed walk (walk_variant = 2), every answer compared with the oracle on the base sample.

  python tools/actions_wide_case.py [--points blacklist_1k:urls,dict_1k:k128] [--log2-strings 21]
"""
import argparse
import json
import os

import numpy as np
import torch

import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from pire_amd import workloads as W


def timed(launch, total_bytes, reps):
    settle = max(10, int(20.0 / max(total_bytes / 2.5e9, 0.05)))
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


def batch(entry, corpus, n):
    """(base text, base offsets, repeats): the base sample is repeated to n strings."""
    nbase = min(n, 1 << 16)
    if corpus == "urls":
        btext, boffs = W.wide_urls(entry, 0x5EED5EED, nbase)
    else:   # log lines: the corpus' records cut into lines of 64..1023 bytes
        rng = np.random.RandomState(4)
        lens = rng.randint(64, 1024, size=nbase).astype(np.uint64)
        boffs = np.zeros(nbase + 1, dtype=np.uint64)
        boffs[1:] = np.cumsum(lens)
        total = int(boffs[-1])
        planted = corpus.endswith("+w")   # "+w": a word of the dictionary written into every 10th line (the corpora themselves hold none)
        btext = W.wide_records(entry, corpus[:-2] if planted else corpus, 0x5EED5EED, (total + 1023) // 1024, 1024).reshape(-1)[:total].copy()
        if planted:
            words = W.dictionary_words(entry)
            for i in range(0, nbase, 10):
                w = words[rng.randint(0, len(words))]
                at = int(boffs[i]) + rng.randint(0, int(lens[i]) - len(w))
                btext[at:at + len(w)] = np.frombuffer(w, dtype=np.uint8)
    return btext, boffs, n // nbase


def point(entry, corpus, n, reps):
    blob = W.load_blob(entry["blob"])
    o = ob.OracleScanner(blob)
    btext, boffs, rep = batch(entry, corpus, n)
    nbase = len(boffs) - 1
    n = nbase * rep
    lens = np.tile(np.diff(boffs), rep)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    total = int(offs[-1])
    text = torch.as_tensor(np.ascontiguousarray(btext), device="cuda").repeat(rep).contiguous()
    doffs = torch.as_tensor(offs.astype(np.int64), device="cuda")
    out_len = torch.empty(n, dtype=torch.int64, device="cuda")
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    t = pire_amd.Table(blob)
    t.upload()
    info = t.info
    R = info.regexps
    res = torch.empty((n, max(R, 1)), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    # (through BeginMark and EndMark: the dictionary scanners are built with Fsm::Surround, the blacklist scanners anchored)
    want = {("prefix", True): o.prefix(btext, boffs, True, True, True), ("prefix", False): o.prefix(btext, boffs, False, True, True)}
    hi, hf, hr = o.run_half_final(btext, boffs)
    out = {"set": entry["name"], "corpus": corpus, "strings": n, "GiB": round(total / 2**30, 3), "states": info.states,
           "letters": info.letters, "regexps": R, "share_of_strings_with_a_prefix": round(float((want[("prefix", True)] >= 0).mean()), 4)}
    only = set(os.environ.get("ACTIONS_WIDE_LEGS", "dense,wide").split(","))
    for variant, label in ((1, "dense"), (2, "wide")):
        if label not in only:
            continue
        pb.set_config(walk_variant=variant, zip_variant=0, auto_adapt=1)
        leg = {}

        def run_plain():
            t.run_device(text.data_ptr(), doffs.data_ptr(), n, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)

        for _ in range(3):   # the ranking learned from plain scans of the batch with the walk that is measured
            run_plain()
            torch.cuda.synchronize()
            t.adapt()
        for longest in (True, False):
            def launch():
                t.prefix_device(text.data_ptr(), doffs.data_ptr(), n, longest, out_len.data_ptr(), through_begin=True, through_end=True, stream=stream)
            mean, best = timed(launch, total, reps)
            got = out_len.cpu().numpy().reshape(rep, nbase)
            leg["LongestPrefix" if longest else "ShortestPrefix"] = {
                "kernel": pb.last_kernel(), "GBps": round(total / mean / 1e6, 1), "GBps_best": round(total / best / 1e6, 1),
                "ms": round(mean, 4), "parity_all_strings": bool((got == want[("prefix", longest)][None, :]).all())}

        def launch_hf():
            t.run_half_final_device(text.data_ptr(), doffs.data_ptr(), n, 3, idx.data_ptr(), fin.data_ptr(), res.data_ptr(), stream)
        mean, best = timed(launch_hf, total, reps)
        gi = idx.cpu().numpy().astype(np.uint32).reshape(rep, nbase)
        gr = res.cpu().numpy().astype(np.uint64).reshape(rep, nbase, max(R, 1))[:, :, :R]
        leg["HalfFinal"] = {"kernel": pb.last_kernel(), "GBps": round(total / mean / 1e6, 1), "GBps_best": round(total / best / 1e6, 1),
                            "ms": round(mean, 4), "parity_all_strings": bool((gi == hi[None, :]).all() and (gr == hr[None, :, :]).all())}
        i2 = t.refresh_info()
        leg["tier_states"] = int(i2.wide_states)
        leg["zipped"] = bool(i2.zip_full_states)
        out[label] = leg
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", default="blacklist_1k:urls,dict_1k:k128,dict_1k:k128+w,dict_10k:k2048+w")
    ap.add_argument("--log2-strings", type=int, default=21)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    for pt in args.points.split(","):
        name, corpus = pt.split(":")
        print(json.dumps(point(W.wide_set(name), corpus, 1 << args.log2_strings, args.reps)), flush=True)


if __name__ == "__main__":
    main()
