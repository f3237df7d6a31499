"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY (ctypes/numpy face of b3m_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package bwtb3m_amd never does.

What it restates (see b3m_oracle.c header for the parity status):
  * BWT by definition -- /root/reference/README.md:55-59, /root/reference/src/lcpbit.cpp:3667-3669
  * balanced block merging (leaf sort, gap array by LF walk, interleave) --
    SURVEY.md 3.1 / Appendix A; rank conventions /root/reference/src/bwtdecodeblock.cpp:232-284
  * input decoding for pac / pacterm in numpy, independent of
    the product's C++ reader, so that reader and writer are checked against each other.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libb3m_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "b3m_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libb3m_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        i64, p = ctypes.c_int64, ctypes.c_void_p
        L.b3mo_sa_naive.argtypes = [p, i64, p]
        L.b3mo_sa_circular.argtypes = [p, i64, p]
        L.b3mo_bwt_from_sa.argtypes = [p, i64, p, p]
        L.b3mo_bwt.argtypes = [p, i64, p]
        L.b3mo_lcpnext.argtypes = [p, i64, i64, i64, i64]
        L.b3mo_lcpnext.restype = i64
        L.b3mo_leaf_sort_window.argtypes = [p, i64, i64, i64, i64, p]
        L.b3mo_leaf_products.argtypes = [p, i64, i64, i64, p, p, p, p]
        L.b3mo_rank_of.argtypes = [p, i64, p, i64, i64]
        L.b3mo_rank_of.restype = i64
        L.b3mo_gap.argtypes = [p, i64, i64, i64, i64, p, i64, p, i64, ctypes.c_int, p, p, p]
        L.b3mo_gap_range.argtypes = [p, i64, i64, i64, i64, p, i64, p, ctypes.c_int, i64, i64, i64, p, p]
        L.b3mo_interleave.argtypes = [p, p, i64, p, i64, p, i64, p]
        L.b3mo_b3m.argtypes = [p, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, i64, p, p, p]
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("oracle %s failed rc=%d" % (what, rc))


def sa_naive(t):
    t, tp = _u8(t)
    sa = np.empty(len(t), dtype=np.int64)
    _check(lib().b3mo_sa_naive(tp, len(t), sa.ctypes.data), "sa_naive")
    return sa


def sa_circular(t):
    t, tp = _u8(t)
    sa = np.empty(len(t), dtype=np.int64)
    _check(lib().b3mo_sa_circular(tp, len(t), sa.ctypes.data), "sa_circular")
    return sa


def bwt_from_sa(t, sa):
    """BWT[i] = s[((SA[i]+n)-1) % n] -- /root/reference/src/lcpbit.cpp:3667-3669"""
    t = np.ascontiguousarray(t, dtype=np.uint8)
    n = len(t)
    return t[(np.asarray(sa, dtype=np.int64) + n - 1) % n] if n else t.copy()


def bwt(t):
    t, tp = _u8(t)
    out = np.empty(len(t), dtype=np.uint8)
    if len(t):
        _check(lib().b3mo_bwt(tp, len(t), out.ctypes.data), "bwt")
    return out


def bwt_naive(t):
    return bwt_from_sa(t, sa_naive(t))


def lcpnext(t, a, m, thres=16384):
    t, tp = _u8(t)
    return int(lib().b3mo_lcpnext(tp, len(t), a, m, thres))


def leaf_sort(t, a, m, largelcpthres=16384):
    """(row a7) absolute positions of the block's suffixes in rotation order."""
    t, tp = _u8(t)
    sa = np.empty(m, dtype=np.int64)
    _check(lib().b3mo_leaf_sort_window(tp, len(t), a, m, largelcpthres, sa.ctypes.data), "leaf_sort")
    return sa


def leaf_products(t, a, m, sa):
    """block BWT, p0 (rank of S_a), gt[0..m] with gt[k] = [S_{a+k} > S_a]."""
    t, tp = _u8(t)
    sa = np.ascontiguousarray(sa, dtype=np.int64)
    b = np.empty(m, dtype=np.uint8)
    gt = np.zeros(m + 1, dtype=np.uint8)
    p0 = ctypes.c_int64(0)
    _check(lib().b3mo_leaf_products(tp, len(t), a, m, sa.ctypes.data, b.ctypes.data,
                                    ctypes.byref(p0), gt.ctypes.data), "leaf_products")
    return b, int(p0.value), gt


def rank_of(t, sa, x):
    t, tp = _u8(t)
    sa = np.ascontiguousarray(sa, dtype=np.int64)
    return int(lib().b3mo_rank_of(tp, len(t), sa.ctypes.data, len(sa), x))


def gap(t, a, m, mr, bwtL, p0L, gtR, rank_end, sigma):
    """(row a8) returns G[0..m], gtP tail (m+mr+1 array with [m..m+mr] filled), rankR[0..mr)."""
    t, tp = _u8(t)
    bwtL, bp = _u8(bwtL)
    gtR, gp = _u8(gtR)
    G = np.zeros(m + 1, dtype=np.uint32)
    gtP = np.zeros(m + mr + 1, dtype=np.uint8)
    rankR = np.zeros(mr, dtype=np.int64)
    _check(lib().b3mo_gap(tp, len(t), a, m, mr, bp, p0L, gp, rank_end, sigma,
                          G.ctypes.data, gtP.ctypes.data, rankR.ctypes.data), "gap")
    return G, gtP, rankR


def gap_range(t, a, m, mr, bwtL, p0L, gtR, sigma, k_hi, k_lo, rho):
    """ranks in L of R's positions k_lo..k_hi-1 walked from the anchor rho = rank_L(S_{a+m+k_hi});
    returns (rankR[k_hi-k_lo], gtP array of m+mr+1 with the walked positions filled)"""
    t, tp = _u8(t)
    bwtL, bp = _u8(bwtL)
    gtR, gp = _u8(gtR)
    rankR = np.zeros(max(1, k_hi - k_lo), dtype=np.int64)
    gtP = np.zeros(m + mr + 1, dtype=np.uint8)
    _check(lib().b3mo_gap_range(tp, len(t), a, m, mr, bp, p0L, gp, sigma, k_hi, k_lo, rho,
                                rankR.ctypes.data, gtP.ctypes.data), "gap_range")
    return rankR[: k_hi - k_lo], gtP


def interleave(G, bwtL, bwtR, p0L):
    """(row a10) merged BWT and the new rank of L's p0 entry."""
    G = np.ascontiguousarray(G, dtype=np.uint32)
    bwtL, lp = _u8(bwtL)
    bwtR, rp = _u8(bwtR)
    out = np.empty(len(bwtL) + len(bwtR), dtype=np.uint8)
    p0 = ctypes.c_int64(-1)
    _check(lib().b3mo_interleave(G.ctypes.data, lp, len(bwtL), rp, len(bwtR), out.ctypes.data,
                                 p0L, ctypes.byref(p0)), "interleave")
    return out, int(p0.value)


def b3m(t, numblocks, numthreads=1, sigma=256, largelcpthres=16384):
    """Whole restated path; returns (bwt, p0, stats[leaf, gap, interleave, total] seconds)."""
    t, tp = _u8(t)
    out = np.empty(len(t), dtype=np.uint8)
    p0 = ctypes.c_int64(-1)
    stats = np.zeros(4, dtype=np.float64)
    _check(lib().b3mo_b3m(tp, len(t), numblocks, numthreads, sigma, largelcpthres,
                          out.ctypes.data, ctypes.byref(p0), stats.ctypes.data), "b3m")
    return out, int(p0.value), stats


# --- input formats, numpy restatement ----------------------------------

def pac_encode(bases):
    """bases: array of 0..3.  BWA .pac: base i in byte i>>2 at bit ((~i)&3)<<1; ceil(l/4)
    data bytes, one extra 0 byte if l%4==0, then one byte l%4."""
    bases = np.asarray(bases, dtype=np.uint8)
    l = len(bases)
    nb = (l + 3) // 4
    pad = np.zeros(nb * 4, dtype=np.uint8)
    pad[:l] = bases
    q = pad.reshape(nb, 4)
    data = ((q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]).astype(np.uint8)
    tail = [0, 0] if l % 4 == 0 else [l % 4]
    return np.concatenate([data, np.array(tail, dtype=np.uint8)])


def pac_decode(buf):
    buf = np.asarray(buf, dtype=np.uint8)
    if len(buf) < 1:
        raise ValueError("pac file too short")
    last = int(buf[-1])
    if len(buf) < 2 or last > 3:
        raise ValueError("not a pac file")
    l = (len(buf) - 2) * 4 + last  # file size is always floor(l/4)+2
    data = buf[: (l + 3) // 4]
    out = np.empty(len(data) * 4, dtype=np.uint8)
    out[0::4] = data >> 6
    out[1::4] = (data >> 4) & 3
    out[2::4] = (data >> 2) & 3
    out[3::4] = data & 3
    return out[:l]


def compact_encode(symbols, bits):
    """libmaus2 CompactArray file image [UPSTREAM-RECALL, unpinned: no sample file or source in /root/reference; producers
    /root/reference/src/fagzToCompact4.cpp:113 (2 bits) and fagzToCompact.cpp:85 (3 bits)]: four big-endian u64 -- bits per
    symbol, number of symbols, number of 64-bit words, the word count again -- then the words, big-endian, symbols packed
    from the most significant bit down."""
    sym = np.asarray(symbols, dtype=np.uint64)
    n = len(sym)
    nbits = n * bits
    nw = (nbits + 63) // 64
    bitarr = np.zeros(nw * 64, dtype=np.uint8)
    for k in range(bits):
        bitarr[k:n * bits:bits] = ((sym >> np.uint64(bits - 1 - k)) & np.uint64(1)).astype(np.uint8)
    data = np.packbits(bitarr)
    hdr = np.array([bits, n, nw, nw], dtype=">u8").view(np.uint8)
    return np.concatenate([hdr, data]).astype(np.uint8)


def compact_decode(buf):
    buf = np.asarray(buf, dtype=np.uint8)
    bits, n, nw, nw2 = (int(x) for x in buf[:32].view(">u8"))
    assert nw == nw2 and 1 <= bits <= 8 and len(buf) >= 32 + 8 * nw and nw * 64 >= n * bits, "not a compact array file"
    bitarr = np.unpackbits(buf[32:32 + 8 * nw])[: n * bits].reshape(n, bits).astype(np.uint16)
    w = (1 << np.arange(bits - 1, -1, -1)).astype(np.uint16)
    return (bitarr * w).sum(axis=1).astype(np.uint8)


def symbols_from_file_bytes(buf, inputtype):
    """Logical symbol sequence of an input file (row a3)."""
    if inputtype == "bytestream":
        return np.asarray(buf, dtype=np.uint8)
    if inputtype == "compactstream":
        return compact_decode(buf)
    if inputtype == "pac":
        return pac_decode(buf)
    if inputtype == "pacterm":
        b = pac_decode(buf)
        return np.concatenate([b + 1, np.zeros(1, dtype=np.uint8)]).astype(np.uint8)
    if inputtype == "utf-8":
        # the text is the file's sequence of Unicode code points (role of libmaus2's utf-8 input wrapper; decoder call site
        # /root/reference/src/bwtb3mdecoderl.cpp:47-57 writes them back with UTF8::encodeUTF8)
        return np.frombuffer(bytes(np.asarray(buf, dtype=np.uint8)).decode("utf-8").encode("utf-32-le"), dtype="<u4").astype(np.int64)
    raise ValueError("oracle: unsupported inputtype " + inputtype)


def bwt_wide(sym):
    """BWT (rotation order, README.md:55-59) of a text over any integer alphabet, as int64.  Texts of <= 256 distinct symbols go through the
    byte BWT on their ranks (order-preserving); larger alphabets through a plain prefix-doubling suffix sort in numpy."""
    sym = np.asarray(sym, dtype=np.int64)
    n = len(sym)
    if n == 0:
        return sym.copy()
    vals, rk = np.unique(sym, return_inverse=True)
    if len(vals) <= 256:
        return vals[bwt(rk.astype(np.uint8)).astype(np.int64)]
    r = rk.astype(np.int64)
    h = 1
    idx = np.arange(n, dtype=np.int64)
    while True:
        key2 = r[(idx + h) % n]
        order = np.lexsort((key2, r))
        a, b = r[order], key2[order]
        newr = np.empty(n, dtype=np.int64)
        newr[order] = np.concatenate([[0], np.cumsum((a[1:] != a[:-1]) | (b[1:] != b[:-1]))])
        r = newr
        h *= 2
        if r.max() == n - 1 or h >= n:
            break
    order = np.argsort(r, kind="stable")
    return sym[(order - 1) % n]


# --- deterministic synthetic inputs (splitmix64, seed 0xB3B3B3B3 + cfg) ----------

def splitmix64(seed, count):
    """count 64-bit outputs of splitmix64 started at `seed` (vectorised)."""
    M = (1 << 64) - 1
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed & M) + idx * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def synth_bases(nbases, seed):
    """uniform 2-bit bases: 32 bases per splitmix64 word, most significant pair first."""
    words = splitmix64(seed, (nbases + 31) // 32)
    shifts = np.arange(62, -2, -2, dtype=np.uint64)
    b = ((words[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)
    return b[:nbases]


def synth_bytes(nbytes, seed):
    words = splitmix64(seed, (nbytes + 7) // 8)
    return words.astype(">u8").view(np.uint8)[:nbytes].copy()


def synth_repeat_bases(nbases, seed, unit=37, copies=3):
    """repeat-rich correctness input: tandem repeats + mutated segment copies (not timed)."""
    b = synth_bases(nbases, seed).copy()
    r = splitmix64(seed ^ 0x5EED, 64)
    n = nbases
    if n < 8 * unit:
        return b
    # tandem repeat of a short unit
    s = int(r[0] % np.uint64(max(1, n // 4)))
    L = min(n // 8, unit * 20)
    b[s:s + L] = np.resize(b[s:s + unit], L)
    # copies of a long segment with sparse mutations
    seg = n // 10
    src = int(r[1] % np.uint64(n - seg))
    for c in range(copies):
        dst = int(r[2 + c] % np.uint64(n - seg))
        b[dst:dst + seg] = b[src:src + seg]
        mut = (r[8 + c * 4: 8 + c * 4 + 4] % np.uint64(seg)).astype(np.int64)
        b[dst + mut] = (b[dst + mut] + 1) & 3
    # homopolymer run
    s2 = int(r[20] % np.uint64(n - n // 50 - 1))
    b[s2:s2 + n // 50] = 0
    return b
