"""Benchmark workloads: the committed scanner images and corpus plants the benchmarks run on.

The product does not compile regexps (that stays in the reference library on the host), so
a benchmark needs serialised scanners: the fixtures `tests/golden/make_golden.py` generated with the unmodified
reference (Scanner::Save() bytes + the witnesses planted into the synthetic corpus).  This module only reads them;
it needs neither the oracle nor the test helpers."""
import gzip
import json
import os

from .binding import make_plants

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
_cache = None


def _golden():
    global _cache
    if _cache is None:
        with open(os.path.join(GOLDEN, "cases.json")) as f:
            _cache = json.load(f)
    return _cache


def load_blob(rel: str) -> bytes:
    with open(os.path.join(GOLDEN, rel), "rb") as f:
        data = f.read()
    return gzip.decompress(data) if rel.endswith(".gz") else data


def pattern_set(name: str) -> dict:
    """One of the glued 8-regexp sets (set_a = the headline, set_b, set_d, c2_single): its fixture record."""
    for b in _golden()["big"]:
        if b["name"] == name:
            return b
    raise KeyError(name)


def slow_case(name: str) -> dict:
    for c in _golden()["slow"]:
        if c["name"] == name:
            return c
    with open(os.path.join(GOLDEN, "slow_wide.json")) as f:   # SlowScanners of more than 256 NFA states
        for c in json.load(f)["slow_wide"]:
            if c["name"] == name:
                return c
    raise KeyError(name)


def plants_for(big: dict):
    """Corpus plants of a pattern set: witness r at the tail / at a generated offset, as the fixture says."""
    return make_plants([(bytes.fromhex(h), t) for h, t in zip(big["witnesses_hex"], big["witness_at_tail"])])


def ref_bench_file() -> bytes:
    """The reference's benchmark corpus, tools/bench/test_file (20 485 bytes of C++ text), shipped as a data fixture
    (tests/golden/make_ref_corpus.py); its SHA-256 is checked."""
    import hashlib

    data = load_blob("ref_bench_test_file.gz")
    with open(os.path.join(GOLDEN, "ref_bench_test_file.json")) as f:
        meta = json.load(f)
    assert len(data) == meta["bytes"] and hashlib.sha256(data).hexdigest() == meta["sha256"], "corpus fixture damaged"
    return data


def ref_bench_corpus(total_bytes: int) -> bytes:
    """The big file of the reference's tools/bench/run-bench:126-138: test_file doubled (`cat big big > big.new`) until
    it is at least `total_bytes` long -- NOT truncated there, exactly like the script; callers cut what they need."""
    data = ref_bench_file()
    while len(data) < total_bytes:
        data = data + data
    return data


# ---- wide working sets (round 5): dictionary scanners and token-mixture corpora ---------------------------------------
# The headline corpus (random printable text + planted witnesses) keeps a glued table inside ~100 states.  What the
# reference's own deployment example scans with (samples/blacklist/blacklist.cpp: thousands of domains in one Scanner)
# visits thousands.  tests/golden/make_golden_wide.py compiles such scanners with the unmodified reference and writes
# tests/golden/wide.json; the corpora are defined HERE (numpy only, deterministic in their seed) so that bench.py, the
# tests and the fixture generator build the same bytes.

_wide = None


def _wide_json():
    global _wide
    if _wide is None:
        with open(os.path.join(GOLDEN, "wide.json")) as f:
            _wide = json.load(f)
    return _wide


def wide_sets():
    return _wide_json()["wide"]


def wide_set(name: str) -> dict:
    for w in wide_sets():
        if w["name"] == name:
            return w
    raise KeyError(name)


def dictionary_words(entry: dict):
    """The words of a dictionary scanner, in the order the fixture generator drew them (bytes)."""
    data = load_blob(entry["words_file"]).split(b"\n")
    return [w for w in data if w][:entry["words"]]


_SYL = ("ba be bi bo bu ca ce ci co cu da de di do du fa fe fi fo ga ge go ha he hi ho ja jo ka ke ki ko la le li lo lu ma me "
        "mi mo mu na ne ni no nu pa pe pi po ra re ri ro ru sa se si so su ta te ti to tu va ve vi vo wa we wi xa ya yo za ze "
        "zo an en in on un ar er or st tr ch sh th net web shop news mail blog soft tech data cloud game play free best top "
        "my the e i x 24 365 1 2").split()
_TLD = "com com com com net org ru ru de info biz co.uk io cn fr it nl com.br pl in".split()


def synthetic_domains(n: int, seed: int = 1):
    """n distinct made-up domain names (no network here for a real list): 2-4 syllables, sometimes a hyphen, a TLD; in
    the order drawn."""
    import numpy as np

    rng = np.random.RandomState(seed)
    out, seen = [], set()
    while len(out) < n:
        k = rng.randint(2, 5)
        w = "".join(_SYL[rng.randint(len(_SYL))] for _ in range(k))
        if rng.rand() < 0.08:
            w = w[:len(w) // 2] + "-" + w[len(w) // 2:]
        w = w + "." + _TLD[rng.randint(len(_TLD))]
        if w not in seen:
            seen.add(w)
            out.append(w.encode())
    return out


# Mixed-script words for a LARGE-ALPHABET table (BASELINE config 5's wording): Cyrillic, Greek, accented and
# capitalised Latin syllables, digits -- as UTF-8 some 110 distinct byte values occur, so the scanner the reference compiles
# from them (lexer with Encodings::Utf8(), pire/encoding.cpp:99-111) has more than 100 letter classes.
_SYL_CYR = ("ба бе би бо бу ва ве ви во га го да де ди до ду жа же жи за зе зо ка ке ки ко ку ла ле ли ло лу ма ме ми мо му на не ни но "
            "ну па пе пи по ра ре ри ро ру са се си со су та те ти то ту фа фе хо ца че ша ше щи эк юр яр ый ов ев ин ск ст пр тр "
            "град мир свет дом код сеть игра новь").split()
_SYL_GRE = "αλ βα γε δι εκ ζω ηλ θε ικ κα λο μα νε ξυ ομ πα ρο σι τα υπ φι χο ψη ωρ ος ης ον πολ λογ".split()
_SYL_ACC = ("ré né dé té lé mè prè für grü mü hö kö schö nä lä stra ße ça gar çon ñor se mañ ña pão ção île tô rê sû blå sø ær "
            "ký ží čes řek ło ść").split()
_SYL_CAP = ("Ba Be Co Da Fi Go Ha Jo Ka Le Mi No Pa Ra Sa Ti Vo Wa Net Web Shop News Mail Blog Soft Tech Data Cloud Game Play Qu Xe "
            "Yo Ze Ul Ow Il El Ar Un").split()


def synthetic_words_utf8(n: int, seed: int = 2):
    """n distinct made-up words (bytes, UTF-8), letters and digits only (each is parsed as a PATTERN by the reference's lexer):
    2-4 syllables of one script -- 45 % Cyrillic, 20 % lower-case Latin, 15 % Greek, 10 % accented Latin, 10 % capitalised
    Latin with a number behind it; in the order drawn."""
    import numpy as np

    rng = np.random.RandomState(seed)
    lat = [x for x in _SYL if x.isalpha()]
    out, seen = [], set()
    while len(out) < n:
        r = rng.rand()
        k = rng.randint(2, 5)
        if r < 0.45:
            w = "".join(_SYL_CYR[rng.randint(len(_SYL_CYR))] for _ in range(k))
        elif r < 0.65:
            w = "".join(lat[rng.randint(len(lat))] for _ in range(k))
        elif r < 0.80:
            w = "".join(_SYL_GRE[rng.randint(len(_SYL_GRE))] for _ in range(k))
        elif r < 0.90:
            w = "".join(_SYL_ACC[rng.randint(len(_SYL_ACC))] for _ in range(k))
        else:
            w = "".join(_SYL_CAP[rng.randint(len(_SYL_CAP))] for _ in range(min(k, 3))) + str(rng.randint(0, 1000))
        if w not in seen:
            seen.add(w)
            out.append(w.encode("utf-8"))
    return out


def token_stream(seed: int, tokens, weights, nbytes: int):
    """`nbytes` bytes of text: tokens drawn independently with the given weights, back to back (vectorised)."""
    import numpy as np

    rng = np.random.RandomState(seed)
    lens = np.array([len(t) for t in tokens], dtype=np.int64)
    flat = np.frombuffer(b"".join(tokens), dtype=np.uint8)
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    p = np.asarray(weights, dtype=np.float64)
    p = p / p.sum()
    mean = float((p * lens).sum())
    out = np.empty(0, dtype=np.uint8)
    while out.size < nbytes:
        k = int((nbytes - out.size) / mean * 1.05) + 64
        idx = rng.choice(len(tokens), size=k, p=p)
        ln = lens[idx]
        end = np.cumsum(ln)
        src = np.repeat(starts[idx] - (end - ln), ln) + np.arange(int(end[-1]))
        out = np.concatenate((out, flat[src]))
    return out[:nbytes]


def wide_tokens(entry: dict, corpus: str):
    """(tokens, weights) of a named corpus of a wide set.
    dictionary scanners -- corpus 'k<N>': the labels (word minus its TLD: never a whole dictionary word, so the Surround()ed
    scanner is never absorbed in its accepting state) of the first N words, half of the items, the rest filler words of
    the same syllables; every item followed by a separator.
    set_b -- corpus 'mix': fragments that keep several of the 8 glued patterns half matched at once."""
    import numpy as np

    if entry["kind"] == "dictionary":
        k = int(corpus[1:])
        words = dictionary_words(entry)[:k]
        rng = np.random.RandomState(77)
        if entry.get("script") == "utf8":
            # the words minus their last letter (never a whole dictionary word) and filler of the same scripts' syllables
            labels = sorted({w.decode("utf-8")[:-1].encode("utf-8") for w in words})
            syl = _SYL_CYR * 3 + [x for x in _SYL if x.isalpha()] + _SYL_GRE + _SYL_ACC + _SYL_CAP
            filler = sorted({"".join(syl[rng.randint(len(syl))] for _ in range(rng.randint(1, 4))).encode("utf-8") for _ in range(4096)})
            # ... none of which may CONTAIN a word of the whole dictionary (two-syllable words occur inside longer ones): the
            # Surround()ed scanner would be absorbed in its accepting state and the rest of the record skipped
            every = dictionary_words(entry)
            first = {}
            for w in every:
                first.setdefault(w[:2], []).append(w)

            def clean(tok):
                return not any(tok.startswith(w, i) for i in range(len(tok) - 1) for w in first.get(tok[i:i + 2], ()))

            labels = [t for t in labels if clean(t)]
            filler = [t for t in filler if clean(t)]
        else:
            labels = sorted({w.split(b".")[0] for w in words})
            filler = sorted({"".join(_SYL[rng.randint(len(_SYL))] for _ in range(rng.randint(1, 4))).encode() for _ in range(4096)})
        seps = [b" ", b"/", b"\n", b"=", b"_"]
        toks, wts = [], []
        for group, share in ((labels, 0.5), (filler, 0.5)):
            for t in group:
                for s in seps:
                    toks.append(t + s)
                    wts.append(share / len(group) / len(seps))
        return toks, wts
    if entry["kind"] == "glued" and corpus == "mix":
        alpha = b"ABCDEFGHIJKLMNOPQRSTUVWXYZ"
        toks = [b"hello ", b"hello  w", b"w", b" ", b"  ", b"d", b"http://", b"http://ab.", b"cd.", b"xyz.", b"foo", b"bar", b"baz",
                b"foobar", b"qu", b"abc@", b"def.co", b"(123) ", b"456-78", b"123-", b"-456", b"X", b"Y", b"Z", b"abc ", b"\n"]
        toks += [alpha[:k] for k in range(1, 26)] + [b"X" + alpha[:k] for k in range(1, 20, 2)]
        rng = np.random.RandomState(78)
        toks += sorted({"".join(_SYL[rng.randint(len(_SYL))] for _ in range(rng.randint(1, 3))).encode() + b" " for _ in range(64)})
        return toks, [1.0] * len(toks)
    raise KeyError((entry["name"], corpus))


def wide_records(entry: dict, corpus: str, seed: int, n: int, length: int):
    """[n, length] u8: the token stream of the corpus cut into records."""
    toks, wts = wide_tokens(entry, corpus)
    return token_stream(seed, toks, wts, n * length).reshape(n, length)


def rotated_repeat_order(n: int, nbase: int, first: int = 0):
    """Record index (into a base of `nbase` distinct records) of every string of a batch of `n` that repeats the base --
    every repeat rotated by its own number of records: string i = base[(first + i + 1237 * (i // nbase)) % nbase].

    A plain repeat has a period of nbase / 64 tasks; with nbase = 16 384 that is 256 -- the number of CUs -- and a kernel that
    hands task b + 256 w to wave w of block b then walks the SAME 64 records in all 16 waves of a CU, whose table loads hit
    each other's lines in the L1: the corpora with 30 % of the steps outside the wide rows measured 1.04 instead of 0.54
    TB/s that way (lesson 21).  1237 is odd (every alignment of a 64-record task occurs) and no
    two waves of a block meet the same records (tests/test_workloads.py)."""
    import numpy as np

    i = np.arange(n, dtype=np.int64)
    return (i % nbase + (i // nbase) * 1237 + first) % nbase


def wide_urls(entry: dict, seed: int, n: int, listed_share: float = 0.25):
    """(text u8, offsets u64[n+1]): URLs for a blacklist scanner (samples/blacklist/blacklist.cpp:78-85 reads one per line):
    scheme, 0-2 subdomain labels, a host -- a word of the dictionary with probability `listed_share`, else a made-up domain
    of the same kind --, a path."""
    import numpy as np

    rng = np.random.RandomState(seed)
    words = dictionary_words(entry)
    other = synthetic_domains(4096, seed=991)
    subs = [b"", b"", b"www.", b"m.", b"shop.", b"mail.", b"a.b."]
    paths = [b"", b"/", b"/index.html", b"/a/b/c?d=e", b"/news/2010/07/14/some-long-article-title.html", b"/img/logo.png",
             b"/search?q=perl+incompatible+regular+expressions&lang=en", b"/~user/dir/",
             b"/catalog/section/12/item/34567/reviews?page=3&sort=date&order=desc#comment-991",
             b"/cgi-bin/view.pl?doc=/pub/docs/pire/README&format=text&session=4f2a91c07d55e38b",
             b"/static/js/vendor/jquery-1.4.2.min.js?v=20100714", b"/forum/viewtopic.php?f=12&t=34567&start=40&hilit=regexp+scanner",
             b"/download/releases/0.0.6/pire-0.0.6.tar.gz", b"/blog/2010/07/yet-another-post-about-finite-automata-and-their-tables/"]
    schemes = [b"http://", b"http://", b"https://", b"ftp://", b""]
    parts = []
    listed = rng.rand(n) < listed_share
    wi = rng.randint(len(words), size=n)
    oi = rng.randint(len(other), size=n)
    si, pi, ci = rng.randint(len(subs), size=n), rng.randint(len(paths), size=n), rng.randint(len(schemes), size=n)
    for i in range(n):
        parts.append(schemes[ci[i]] + subs[si[i]] + (words[wi[i]] if listed[i] else other[oi[i]]) + paths[pi[i]])
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(u) for u in parts], dtype=np.uint64)
    return np.frombuffer(b"".join(parts), dtype=np.uint8), offs
