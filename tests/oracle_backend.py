"""Oracle-backed compute backend for bwtb3m_amd.distributed -- TEST ONLY.

Lets the multi-GPU driver's host logic (tree plan, chain ranges, gap-bucket all-to-all, slice
bookkeeping, gt bit plumbing) run under gloo on CPU tensors.  Every compute step is the CPU
restatement in oracle/; the product never imports this."""
import numpy as np
import torch

from oracle import oracle as O


def _bits_to_words(bits_abs):
    nw = (len(bits_abs) + 31) // 32
    pad = np.zeros(nw * 32, dtype=np.uint32)
    pad[: len(bits_abs)] = bits_abs
    w = (pad.reshape(nw, 32) << np.arange(32, dtype=np.uint32)).sum(axis=1, dtype=np.uint64).astype(np.uint32)
    return w.view(np.int32)


def _words_to_bits(words, count):
    w = words.numpy().view(np.uint32)
    idx = np.arange(count)
    return ((w[idx >> 5] >> (idx & 31)) & 1).astype(np.uint8)


class OracleBackend:
    """same interface as bwtb3m_amd.distributed.HipBackend (this rank's b3m_dist_plan), computed by the oracle on CPU tensors"""

    def __init__(self, sym, itype):
        self.sym = np.ascontiguousarray(sym, dtype=np.uint8)
        self.itype = itype
        self.sigma = {"pacterm": 5, "pac": 4}.get(itype, 256)
        self.sub = 1 if itype == "pacterm" else 0

    def setup(self, n, itype, max_leaf, max_L, max_R):
        self.n = n
        self.limits = (max_leaf, max_L, max_R)

    def gt_words(self, n):
        return (n + 1 + 31) // 32 + 2

    def set_text(self, d_text):
        pass

    def leaf(self, a, m):
        assert m <= self.limits[0]
        n = self.n
        sa = O.leaf_sort(self.sym, a, m)
        bw, p0, gt = O.leaf_products(self.sym, a, m, sa)
        self.sa = sa
        absb = np.zeros(self.gt_words(n) * 32, dtype=np.uint8)
        absb[a + 1: a + m + 1] = gt[1:]
        h5 = np.bincount(self.sym[a:a + m], minlength=256) if self.itype == "bytestream" else np.bincount(self.sym[a:a + m], minlength=5)[:5]
        return torch.from_numpy(bw), torch.from_numpy(_bits_to_words(absb)), p0, [int(x) for x in h5]

    def rank_of_into(self, positions, accum):
        accum += torch.from_numpy(np.array([O.rank_of(self.sym, self.sa, int(x)) for x in positions], dtype=np.int64))

    def bits_replace(self, dst, seg, seg_w0, b, e):
        from bwtb3m_amd.distributed import _replace_bits_seg
        _replace_bits_seg(dst, seg, seg_w0, b, e)

    def set_endbit(self, gt, bit, anchor, p0L):
        from bwtb3m_amd.distributed import _set_bit
        _set_bit(gt, bit, 1 if int(anchor[0]) > p0L else 0)

    def packed_len(self, count):
        return count if self.itype == "bytestream" else (count + 3) // 4

    def pack2(self, bwt):
        if self.itype == "bytestream":
            return bwt.clone()
        b = bwt.numpy().astype(np.int64)
        codes = (b - self.sub) & 3
        pad = np.zeros((len(codes) + 3) // 4 * 4, dtype=np.int64)
        pad[: len(codes)] = codes
        q = pad.reshape(-1, 4)
        return torch.from_numpy((q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8))

    def build_dict(self, parts, m, p0L):
        assert m <= self.limits[1]
        out = []
        for t, cnt in parts:
            if self.itype == "bytestream":
                out.append(t.numpy()[:cnt].copy())
                continue
            v = t.numpy().astype(np.int64)
            codes = np.stack([(v >> (2 * k)) & 3 for k in range(4)], axis=1).reshape(-1)[:cnt]
            out.append((codes + self.sub).astype(np.uint8))
        self.bwtL = np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)
        assert len(self.bwtL) == m
        self.p0L = p0L

    def walk(self, a, m, mr, symstart, gt_abs, anchors, chain_first, num_chains, chain_len, bounds):
        n = self.n
        # the C array handed in must be the one of L's text
        hist = np.bincount(self.sym[a:a + m], minlength=5)
        if self.itype == "bytestream":
            assert list(symstart) == [int(x) for x in np.concatenate([[0], np.cumsum(np.bincount(self.sym[a:a + m], minlength=256)[:255])])]
        elif self.itype == "pacterm":
            assert list(symstart) == [int(hist[0] + hist[1:1 + c].sum()) for c in range(4)]
        else:
            assert list(symstart) == [int(hist[:c].sum()) for c in range(4)]
        nb = self.gt_words(n) * 32
        gtb = _words_to_bits(gt_abs, nb)
        gtR = gtb[a + m: a + m + mr + 1].copy()
        out_bits = np.zeros(nb, dtype=np.uint8)
        ranks = []
        for i in range(num_chains):
            c = chain_first + i
            k_hi = mr - c * chain_len
            k_lo = max(0, k_hi - chain_len)
            rk, gtP = O.gap_range(self.sym, a, m, mr, self.bwtL, self.p0L, gtR, self.sigma, k_hi, k_lo, int(anchors[i]))      # (anchors: int64 tensor)
            ranks.append(rk)
            out_bits[a + m + k_lo: a + m + k_hi] = gtP[m + k_lo: m + k_hi]
        r = np.sort(np.concatenate(ranks)) if ranks else np.zeros(0, dtype=np.int64)
        cuts = np.searchsorted(r, np.asarray(bounds, dtype=np.int64), side="left").astype(np.int64)
        return torch.from_numpy(r.astype(np.int32)), cuts, torch.from_numpy(_bits_to_words(out_bits))

    def l_bytes(self, a, l0, l1):
        out = self.bwtL[l0:l1].copy()
        if l0 <= self.p0L < l1:
            out[self.p0L - l0] = self.sym[(a - 1) % self.n]      # the entry of L's first suffix: the symbol in front of L
        return torch.from_numpy(out)

    def _unpack(self, t, cnt):
        if self.itype == "bytestream":
            return t.numpy()[:cnt].copy()
        v = t.numpy().astype(np.int64)
        codes = np.stack([(v >> (2 * k)) & 3 for k in range(4)], axis=1).reshape(-1)[:cnt]
        return (codes + self.sub).astype(np.uint8)

    def r_bytes(self, parts, q0, q1):
        allr = np.concatenate([self._unpack(t, cnt) for t, cnt in parts]) if parts else np.zeros(0, dtype=np.uint8)
        assert q1 <= len(allr)
        return torch.from_numpy(allr[q0:q1].copy())

    def sort_u32(self, keys, nbits):
        return torch.sort(keys)[0]

    def interleave(self, sr_local, bwtL, bwtR):
        m = bwtL.numel()
        G = np.bincount(sr_local.numpy().astype(np.int64), minlength=m + 1).astype(np.uint32)
        out, _ = O.interleave(G, bwtL.numpy(), bwtR.numpy(), 0)
        return torch.from_numpy(out)
