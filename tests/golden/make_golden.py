#!/usr/bin/env python3
"""Generates tests/golden/bwt_golden.json from the oracle (oracle/b3m_oracle.c).

The reference binary cannot run here (its algorithm lives in libmaus2, absent: SURVEY.md 8c) and
ships no fixtures, so these vectors pin the DEFINITION (/root/reference/README.md:55-59,
/root/reference/src/lcpbit.cpp:3667-3669) as computed by two independent sorters of the oracle that
must agree (naive rotation sort for the small cases, circular prefix doubling for the rest).
Families mirror /root/reference/src/lcpbit.cpp:4053-4066 (fixed string, exhaustive
length x alphabet, seeded random).  Run:  python tests/golden/make_golden.py
"""
import hashlib
import itertools
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    g = {"literal": {}, "exhaustive": {}, "random_bytestream": {}, "random_pacterm": {}, "random_pac": {}, "intermediate": {}}
    for s in ["abbab#", "abab", "aaaa", "a", "ab", "mississippi", "abcabcabc", "banana", "abracadabra"]:
        t = np.frombuffer(s.encode(), dtype=np.uint8)
        b1, b2 = O.bwt_naive(t), O.bwt(t)
        assert np.array_equal(b1, b2)
        g["literal"][s] = bytes(b1).decode()
    for name, sigma, L in [("len6_1bit", 2, 6), ("len6_3bit", 8, 6), ("len8_2bit", 4, 8), ("len8_1bit", 2, 8), ("len6_2bit", 4, 6)]:
        h = hashlib.sha256()
        for tup in itertools.product(range(sigma), repeat=L):
            t = np.array(tup, dtype=np.uint8)
            b = O.bwt(t)
            if L <= 6 or sigma <= 2:
                assert np.array_equal(b, O.bwt_naive(t))
            h.update(bytes(b))
        g["exhaustive"][name] = h.hexdigest()
    for n in [1024, 65536, 1 << 20]:
        t = O.synth_bytes(n, 0xB3B3B3B3 + 0)
        g["random_bytestream"][str(n)] = sha(O.bwt(t))
    for nb in [1000, 65536, 5_000_000]:
        b = O.synth_bases(nb, 0xB3B3B3B3 + 1)
        s = np.concatenate([b + 1, [0]]).astype(np.uint8)
        bw = O.bwt(s)
        g["random_pacterm"][str(nb)] = {"sha256": sha(bw), "zero_at": int(np.nonzero(bw == 0)[0][0]), "pac_sha256": sha(O.pac_encode(b))}
        g["random_pac"][str(nb)] = sha(O.bwt(b)) if nb < 5_000_000 else None
    # intermediates of a 4-block balanced tree over the 64 K case (pacterm)
    b = O.synth_bases(65536, 0xB3B3B3B3 + 1)
    s = np.concatenate([b + 1, [0]]).astype(np.uint8)
    n = len(s)
    bs = (n + 3) // 4
    leaves = []
    for k in range(4):
        a = k * bs
        m = min(bs, n - a)
        sa = O.leaf_sort(s, a, m)
        bw, p0, gt = O.leaf_products(s, a, m, sa)
        leaves.append((a, m, sa, bw, p0, gt))
        g["intermediate"]["leaf%d" % k] = {"a": a, "m": m, "sa": sha(sa.astype(np.uint32)), "bwt": sha(bw), "p0": p0, "gt": sha(gt[1:])}
    for k in (0, 2):
        a, m, sa, bw, p0, gt = leaves[k]
        a2, m2, sa2, bw2, p02, gt2 = leaves[k + 1]
        rank_end = O.rank_of(s, sa, (a2 + m2) % n)
        G, gtP, rankR = O.gap(s, a, m, m2, bw, p0, gt2, rank_end, 5)
        out, p0P = O.interleave(G, bw, bw2, p0)
        g["intermediate"]["merge%d%d" % (k, k + 1)] = {"rank_end": rank_end, "G": sha(G), "bwt": sha(out), "p0": p0P}
    with open(os.path.join(HERE, "bwt_golden.json"), "w") as f:
        json.dump(g, f, indent=1, sort_keys=True)
    print("wrote bwt_golden.json")


if __name__ == "__main__":
    main()
