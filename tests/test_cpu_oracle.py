"""CPU tests of the oracle (test infrastructure) against the committed golden vectors, against
known answers from the literature the reference cites, and of its two restatement levels against
each other.  No GPU."""
import hashlib
import itertools
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bwt_golden.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# Known answers independent of this repo: Burrows & Wheeler, SRC Research Report 124 (the paper
# /root/reference/README.md:55-57 names as the definition): S = "abraca" -> L = "caraab", I = 1.
KAT = {"abraca": "caraab", "banana": "nnbaaa", "abracadabra": "rdarcaaaabb", "mississippi": "pssmipissii"}


@pytest.mark.parametrize("s,l", sorted(KAT.items()))
def test_known_answers(oracle, s, l):
    t = np.frombuffer(s.encode(), dtype=np.uint8)
    assert bytes(oracle.bwt_naive(t)).decode() == l
    assert bytes(oracle.bwt(t)).decode() == l


def test_bw_paper_index(oracle):
    # "abraca": the original string is row I = 1 of the sorted rotation matrix
    t = np.frombuffer(b"abraca", dtype=np.uint8)
    sa = oracle.sa_circular(t)
    assert int(np.nonzero(sa == 0)[0][0]) == 1


def test_golden_literals(oracle):
    for s, l in GOLD["literal"].items():
        t = np.frombuffer(s.encode(), dtype=np.uint8)
        assert bytes(oracle.bwt(t)).decode() == l


@pytest.mark.parametrize("name,sigma,L", [("len6_1bit", 2, 6), ("len6_2bit", 4, 6), ("len8_1bit", 2, 8), ("len8_2bit", 4, 8)])
def test_golden_exhaustive(oracle, name, sigma, L):
    h = hashlib.sha256()
    for tup in itertools.product(range(sigma), repeat=L):
        h.update(bytes(oracle.bwt(np.array(tup, dtype=np.uint8))))
    assert h.hexdigest() == GOLD["exhaustive"][name]


def test_exhaustive_naive_vs_doubling(oracle):
    for sigma, L in [(2, 10), (3, 6), (8, 3)]:
        for tup in itertools.product(range(sigma), repeat=L):
            t = np.array(tup, dtype=np.uint8)
            assert np.array_equal(oracle.bwt(t), oracle.bwt_naive(t)), tup


@pytest.mark.parametrize("n", [1024, 65536, 1 << 20])
def test_golden_random_bytestream(oracle, n):
    assert sha(oracle.bwt(oracle.synth_bytes(n, 0xB3B3B3B3))) == GOLD["random_bytestream"][str(n)]


@pytest.mark.parametrize("nb", [1000, 65536])
def test_golden_random_pacterm(oracle, nb):
    b = oracle.synth_bases(nb, 0xB3B3B3B3 + 1)
    buf = oracle.pac_encode(b)
    g = GOLD["random_pacterm"][str(nb)]
    assert sha(buf) == g["pac_sha256"]
    s = oracle.symbols_from_file_bytes(buf, "pacterm")
    bw = oracle.bwt(s)
    assert sha(bw) == g["sha256"] and int(np.nonzero(bw == 0)[0][0]) == g["zero_at"]
    assert sha(oracle.bwt(b)) == GOLD["random_pac"][str(nb)]


def test_restatement_equals_definition_exhaustive(oracle):
    # the Appendix-A merge recurrence proven on every string, every blocking
    for sigma, L in [(2, 8), (3, 6), (4, 5)]:
        for tup in itertools.product(range(sigma), repeat=L):
            t = np.array(tup, dtype=np.uint8)
            ref = oracle.bwt(t)
            for B in range(1, L + 1):
                out, p0, _ = oracle.b3m(t, B, 2, sigma)
                assert np.array_equal(out, ref), (tup, B)


@pytest.mark.parametrize("B", [1, 2, 3, 5, 8, 16])
def test_restatement_random_and_repeats(oracle, B):
    for t in (oracle.synth_bases(20000, 11), oracle.synth_repeat_bases(30000, 12)):
        s = np.concatenate([t + 1, [0]]).astype(np.uint8)
        ref = oracle.bwt(s)
        out, p0, _ = oracle.b3m(s, B, 4, 5)
        assert np.array_equal(out, ref)
        sa = oracle.sa_circular(s)
        assert p0 == int(np.nonzero(sa == 0)[0][0])
        out, p0, _ = oracle.b3m(t, B, 4, 4)          # circular, no terminator
        assert np.array_equal(out, oracle.bwt(t))


def test_leaf_window_sort_equals_global_filter(oracle):
    # the reference's scheme (sort T[a, a+m+lcpnext] as a terminated string) against filtering
    # the global circular suffix array
    t = oracle.synth_repeat_bases(40000, 3)
    sa = oracle.sa_circular(t)
    for a, m in [(0, 10000), (10000, 15000), (25000, 15000), (39000, 1000)]:
        ref = sa[(sa >= a) & (sa < a + m)]
        got = oracle.leaf_sort(t, a, m, largelcpthres=64)   # small threshold: exercises the unbounded retry
        assert np.array_equal(got, ref)
    assert oracle.lcpnext(np.array([0, 1, 0, 1, 0, 1, 2], dtype=np.uint8), 0, 2, 2) == 4


def test_golden_intermediates(oracle):
    b = oracle.synth_bases(65536, 0xB3B3B3B3 + 1)
    s = np.concatenate([b + 1, [0]]).astype(np.uint8)
    n = len(s)
    for k in range(4):
        g = GOLD["intermediate"]["leaf%d" % k]
        sa = oracle.leaf_sort(s, g["a"], g["m"])
        bw, p0, gt = oracle.leaf_products(s, g["a"], g["m"], sa)
        assert sha(sa.astype(np.uint32)) == g["sa"] and sha(bw) == g["bwt"] and p0 == g["p0"] and sha(gt[1:]) == g["gt"]
    assert n == 65537


def test_pac_round_trip(oracle):
    for l in [0, 1, 2, 3, 4, 5, 7, 8, 9, 1000, 1001, 1002, 1003]:
        b = oracle.synth_bases(l, 5) if l else np.zeros(0, dtype=np.uint8)
        buf = oracle.pac_encode(b)
        assert len(buf) == l // 4 + 2            # file size is floor(l/4)+2
        assert np.array_equal(oracle.pac_decode(buf), b)
    assert len(oracle.pac_encode(np.zeros(5_000_000, dtype=np.uint8))) == 1_250_002


def test_compactstream_codec_round_trip():
    # row f4: the compact array image (recalled layout, see oracle.compact_encode) encodes and decodes every width
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    for bits in (1, 2, 3, 4, 7, 8):
        for n in (0, 1, 31, 32, 33, 1000):
            sym = rng.integers(0, 1 << bits, n, dtype=np.uint8)
            buf = O.compact_encode(sym, bits)
            assert len(buf) == 32 + 8 * ((n * bits + 63) // 64)
            assert np.array_equal(O.symbols_from_file_bytes(buf, "compactstream"), sym)
    # 2-bit compact words are a .pac payload: same bit order
    b = rng.integers(0, 4, 1001, dtype=np.uint8)
    assert np.array_equal(O.compact_encode(b, 2)[32:32 + 250], O.pac_encode(b)[:250])


def test_utf8_symbols_and_wide_bwt():
    # row f4: inputtype=utf-8 makes the code points the text; bwt_wide against sorted rotations (README.md:55-59), both of its paths
    from oracle import oracle as O
    txt = "a\u00f1b \u00f1and\u00fa \u03b1\u03b2\u03b3 a\u00f1b \u00f1and\u00fa \U0001F600 a\u00f1b"
    s = O.symbols_from_file_bytes(np.frombuffer(txt.encode("utf-8"), dtype=np.uint8), "utf-8")
    assert list(s) == [ord(c) for c in txt]
    rng = np.random.default_rng(2)
    for w in (s, rng.integers(0, 1000, 2000), rng.integers(0, 3, 500) * 100000, np.array([7, 7, 7, 7])):
        rot = sorted(range(len(w)), key=lambda i: list(w[i:]) + list(w[:i]))
        assert np.array_equal(O.bwt_wide(w), np.array([w[(i - 1) % len(w)] for i in rot]))
