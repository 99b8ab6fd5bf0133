#!/usr/bin/env python3
"""Copies the reference's benchmark corpus (tools/bench/test_file, 20 485 bytes of C++ text, "don't change its
contents so that results can be compared to the previous runs") into tests/golden/ as a DATA fixture, gzip-compressed,
and records its SHA-256.  Run in the build container, where /root/reference exists; the GPU box only has the fixture.

The reference's tools/bench/run-bench:126-138 builds its big file by doubling this file until it reaches the wanted
size; bench.py --corpus cxx (pire_amd/workloads.py ref_bench_corpus) does the same from the fixture.
"""
import gzip
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/tools/bench/test_file"

data = open(SRC, "rb").read()
with open(os.path.join(HERE, "ref_bench_test_file.gz"), "wb") as f:
    f.write(gzip.compress(data, 9, mtime=0))
with open(os.path.join(HERE, "ref_bench_test_file.json"), "w") as f:
    json.dump({"source": "tools/bench/test_file", "bytes": len(data), "sha256": hashlib.sha256(data).hexdigest()}, f)
print(len(data), hashlib.sha256(data).hexdigest())
