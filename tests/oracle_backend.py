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
    def __init__(self, sym, itype):
        self.sym = np.ascontiguousarray(sym, dtype=np.uint8)
        self.sigma = {"pacterm": 5, "pac": 4}.get(itype, 256)

    def gt_words(self, n):
        return (n + 1 + 31) // 32 + 2

    def block_sort(self, d_text, n, itype, a, m):
        sa = O.leaf_sort(self.sym, a, m)
        bw, p0, gt = O.leaf_products(self.sym, a, m, sa)
        absb = np.zeros(self.gt_words(n) * 32, dtype=np.uint8)
        absb[a + 1: a + m + 1] = gt[1:]
        return torch.from_numpy(sa), torch.from_numpy(bw), torch.from_numpy(_bits_to_words(absb)), p0

    def rank_of(self, d_text, n, itype, sa, positions):
        return np.array([O.rank_of(self.sym, sa.numpy(), int(x)) for x in positions], dtype=np.int64)

    def gap_chains(self, d_text, n, itype, a, m, mr, bwtL, p0L, gt_abs, anchors, chain_first, num_chains, chain_len):
        nb = self.gt_words(n) * 32
        gtb = _words_to_bits(gt_abs, nb)
        gtR = gtb[a + m: a + m + mr + 1].copy()
        out_bits = np.zeros(nb, dtype=np.uint8)
        ranks = []
        for i in range(num_chains):
            c = chain_first + i
            k_hi = mr - c * chain_len
            k_lo = max(0, k_hi - chain_len)
            rk, gtP = O.gap_range(self.sym, a, m, mr, bwtL.numpy(), p0L, gtR, self.sigma, k_hi, k_lo, int(anchors[i]))
            ranks.append(rk)
            out_bits[a + m + k_lo: a + m + k_hi] = gtP[m + k_lo: m + k_hi]
        r = np.sort(np.concatenate(ranks)) if ranks else np.zeros(0, dtype=np.int64)
        return torch.from_numpy(r.astype(np.int32)), torch.from_numpy(_bits_to_words(out_bits))

    def sort_u32(self, keys, nbits):
        return torch.sort(keys)[0]

    def interleave(self, sr_local, bwtL, bwtR):
        m = bwtL.numel()
        G = np.bincount(sr_local.numpy().astype(np.int64), minlength=m + 1).astype(np.uint32)
        out, _ = O.interleave(G, bwtL.numpy(), bwtR.numpy(), 0)
        return torch.from_numpy(out)
