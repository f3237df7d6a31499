"""Multi-GPU balanced block merging: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).  SURVEY.md 8(e).

Partitioning (strong scaling: the input is fixed, the work is split):
  * the packed text is replicated on every GPU (0.75 GB at 3 Gbase);
  * the text is cut into B >= P leaf blocks; rank g sorts blocks g, g + P, ... (no communication);
  * a merge (L, R) is executed by ALL ranks that own a block of either side together (with B > P a rank may hold a slice of BOTH sides,
    and with P = 1 the group is one rank: every collective below still runs, on device tensors, through the process group's backend):
      - BWT(L) is broadcast inside the group (every walker needs the rank dictionary),
      - R's text positions are cut into one chain range per rank; each rank walks its range
        (b3m_gap_chains) and gets the ranks in L of those suffixes of R,
      - the rank space [0, |L|] (= the gap array G) is cut into one slice per rank; every rank
        buckets its produced ranks by owning slice: ALL-TO-ALL of gap buckets,
      - the owner sorts what it received (its slice of the sparse gap array), pulls the
        contiguous range of BWT(R) that falls into its slice (second all-to-all) and interleaves:
        it then owns a contiguous slice of the merged BWT.
    No all-reduce of data on this path (a ring would be bound by one xGMI link; the all-to-all
    drives all links).  Small vectors (sizes, anchors, gt bit words) use all_gather/all_reduce.

The compute steps go through a backend object.  The product has exactly one backend, HipBackend
(C ABI of libb3m.so on CUDA tensors); the CPU tests inject an oracle-backed one to exercise this
host logic under gloo -- nothing in this module can fall back to the CPU by itself.
"""
import ctypes
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------------
# plan (pure host logic, identical on every rank)
# ---------------------------------------------------------------------------------------------------
def chains_for(mr, world_hint=1):
    """chain table of a merge: same rule as the single-GPU engine (engine.hip: chains_for)"""
    target = 256 * 1024 * max(1, world_hint)
    chain_len = (max(32, (mr + target - 1) // target) + 3) & ~3      # whole quads of steps (gap_walk_slots)
    return chain_len, (mr + chain_len - 1) // chain_len


class Node:
    """a node of the merge tree: text range [a, a + m), its leaf blocks, the ranks that own one of them (ascending: a node's BWT is the
    concatenation of its members' slices in this order), and its HOLDER -- the owner of its first leaf -- which keeps the node's gt bits"""

    def __init__(self, a, m, leaves, world):
        self.a, self.m, self.leaves = a, m, list(leaves)
        self.ranks = sorted({k % world for k in self.leaves})
        self.holder = self.leaves[0] % world
        self.key = self.leaves[0]


class Merge:
    def __init__(self, L, R, idx, world):
        self.L, self.R, self.idx = L, R, idx
        self.group = sorted(set(L.ranks) | set(R.ranks))
        self.parent = Node(L.a, L.m + R.m, L.leaves + R.leaves, world)
        self.chain_len, self.num_chains = chains_for(R.m, len(self.group))
        self.anchor_off = 0


class TreePlan:
    """balanced binary merge tree over `leaves` >= world leaf blocks; leaf k belongs to rank k % world: "GPU g sorts blocks g, g + P, ..."
    (SURVEY.md 8(e), 3.1 steps 3, 5).  Default: one leaf per rank."""

    def __init__(self, n, world, leaves=None):
        B = world if leaves is None else int(leaves)
        assert 1 <= world <= B <= n, "need world <= leaves <= n"
        bs = (n + B - 1) // B
        self.n, self.world, self.bs, self.B = n, world, bs, B
        self.leaves = []
        for k in range(B):
            a = k * bs
            m = min(bs, n - a)
            assert m >= 1, "more leaf blocks than symbols"
            self.leaves.append(Node(a, m, [k], world))
        self.levels = []
        self.merges = []
        cur = self.leaves
        off = 0
        while len(cur) > 1:
            nxt, lv = [], []
            for i in range(0, len(cur) - 1, 2):
                mg = Merge(cur[i], cur[i + 1], len(self.merges), world)
                mg.anchor_off = off
                off += mg.num_chains
                self.merges.append(mg)
                lv.append(mg)
                nxt.append(mg.parent)
            if len(cur) & 1:
                nxt.append(cur[-1])
            self.levels.append(lv)
            cur = nxt
        self.root = cur[0]
        self.total_anchors = off

    def anchor_positions(self, mg):
        rend = mg.R.a + mg.R.m
        return (rend - np.arange(mg.num_chains, dtype=np.int64) * mg.chain_len) % self.n

    def leaves_of(self, rank):
        return [k for k in range(self.B) if k % self.world == rank]

    def merges_with_leaf_on_left(self, k):
        return [mg for mg in self.merges if k in mg.L.leaves]


def split_even(total, parts):
    """boundaries of `parts` nearly equal contiguous pieces of range(total)"""
    return [(total * i) // parts for i in range(parts + 1)]


# ---------------------------------------------------------------------------------------------------
# the product backend: this rank's b3m_dist_plan (HIP kernels, one persistent workspace) through the C ABI
# ---------------------------------------------------------------------------------------------------
class HipBackend:
    """include/b3m.h, "multi-GPU block merge": every compute step of a rank.  The workspace (padded text, leaf sort, rank dictionary,
    walk / sort buffers) is allocated once in setup(); the tensors that travel (slices, packed slices, rank buckets) are torch's."""

    def __init__(self, device):
        from . import engine  # noqa: F401  (fails loudly if libb3m.so is missing)
        self.engine = engine
        self.device = device
        self._h = None

    def setup(self, n, itype, max_leaf, max_L, max_R):
        eng = self.engine
        self.close()
        self.n, self.type = n, eng._type_id(itype)
        self._h = eng.lib().b3m_dist_create(n, self.type, max_leaf, max_L, max_R)
        if not self._h:
            raise eng._lib.B3MError(eng.lib().b3m_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self.engine.lib().b3m_dist_destroy(self._h)
            self._h = None

    __del__ = close

    def gt_words(self, n):
        return (n + 1 + 31) // 32 + 2

    def set_text(self, d_text):
        eng = self.engine
        eng.check(eng.lib().b3m_dist_set_text(self._h, eng._ptr(d_text), eng._stream()))

    def leaf(self, a, m):
        """-> (bwt, gt_abs words with this leaf's bits, p0, symbol counts of the leaf's text: 0..4 for 2-bit alphabets, 0..255 for bytes)"""
        eng = self.engine
        bw = torch.empty(m, dtype=torch.uint8, device=self.device)
        gt_abs = torch.zeros(self.gt_words(self.n), dtype=torch.int32, device=self.device)
        p0 = ctypes.c_uint64(0)
        h5 = (ctypes.c_uint64 * 5)()
        eng.check(eng.lib().b3m_dist_leaf(self._h, a, m, eng._ptr(bw), eng._ptr(gt_abs), ctypes.byref(p0), h5, eng._stream()))
        if self.type == eng.INPUT_TYPES["bytestream"]:
            h256 = (ctypes.c_uint64 * 256)()
            eng.check(eng.lib().b3m_dist_leaf_hist(self._h, h256))
            return bw, gt_abs, int(p0.value), [int(x) for x in h256]
        return bw, gt_abs, int(p0.value), [int(x) for x in h5]

    def packed_len(self, count):
        """bytes of the transport form of `count` BWT symbols (pack2): 4 per byte for 2-bit alphabets, as they are for bytes"""
        return count if self.type == self.engine.INPUT_TYPES["bytestream"] else (count + 3) // 4

    def rank_of_into(self, positions, accum):
        """accum (int64 device tensor, one entry per position) += rank of the rotation starting at each position among the suffixes of the
        leaf sorted last: the chain anchors stay on the device (b3m_dist_rank_of_dev)"""
        eng = self.engine
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        assert accum.dtype == torch.int64 and accum.is_contiguous() and accum.numel() == len(pos)
        if len(pos):
            eng.check(eng.lib().b3m_dist_rank_of_dev(self._h, pos.ctypes.data, len(pos), eng._ptr(accum), eng._stream()))

    def bits_replace(self, dst, seg, seg_w0, b, e):
        """dst bits [b,e) = bits [b,e) of `seg`, a word range that starts at word seg_w0 of the absolute-position bit array (b3m_bits_copy)"""
        eng = self.engine
        if e > b:
            assert seg.is_contiguous() and dst.is_contiguous() and (b >> 5) >= seg_w0 and ((e - 1) >> 5) - seg_w0 < seg.numel()
            eng.check(eng.lib().b3m_bits_copy(ctypes.c_void_p(seg.data_ptr() - 4 * seg_w0), eng._ptr(dst), b, e, eng._stream()))

    def set_endbit(self, gt, bit, anchor, p0L):
        """bit `bit` of gt = [anchor[0] > p0L] (anchor: device int64 tensor)"""
        eng = self.engine
        eng.check(eng.lib().b3m_dist_endbit(self._h, eng._ptr(gt), bit, eng._ptr(anchor), p0L, eng._stream()))

    def pack2(self, bwt):
        eng = self.engine
        m = bwt.numel()
        pk = torch.empty(self.packed_len(m), dtype=torch.uint8, device=self.device)
        eng.check(eng.lib().b3m_dist_pack2(self._h, eng._ptr(bwt), m, eng._ptr(pk), eng._stream()))
        return pk

    def build_dict(self, parts, m, p0L):
        eng = self.engine
        k = len(parts)
        ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t, _ in parts])
        syms = (ctypes.c_uint64 * k)(*[int(c) for _, c in parts])
        eng.check(eng.lib().b3m_dist_build_dict(self._h, ptrs, syms, k, m, p0L, eng._stream()))

    def walk(self, a, m, mr, symstart, gt_abs, anchors, chain_first, num_chains, chain_len, bounds):
        """-> (ascending ranks of the walked suffixes of R (int32 tensor), cuts[j] = #ranks < bounds[j] (numpy), gt_abs_out words)"""
        eng = self.engine
        cap = max(1, num_chains * chain_len)
        sr = torch.empty(cap, dtype=torch.int32, device=self.device)
        gt_out = torch.zeros(self.gt_words(self.n), dtype=torch.int32, device=self.device)
        cnt = ctypes.c_uint64(0)
        assert anchors.dtype == torch.int64 and anchors.is_contiguous() and anchors.numel() == num_chains
        bd = np.ascontiguousarray(bounds, dtype=np.uint64)
        cuts = np.zeros(len(bd), dtype=np.uint64)
        ss = (ctypes.c_uint64 * len(symstart))(*[int(x) for x in symstart])      # 4 symbol starts (2-bit alphabets) or 256 (bytes)
        eng.check(eng.lib().b3m_dist_walk(self._h, a, m, mr, ss, eng._ptr(gt_abs), eng._ptr(gt_out), eng._ptr(anchors), chain_first, num_chains, chain_len,
                                          bd.ctypes.data, len(bd), eng._ptr(sr), cuts.ctypes.data, ctypes.byref(cnt), eng._stream()))
        return sr[: cnt.value], cuts.astype(np.int64), gt_out

    def l_bytes(self, a, l0, l1):
        """BWT(L)[l0, l1) of the node L that starts at text position a, out of the dictionary (its p0 entry: the symbol in front of L, from the text)"""
        eng = self.engine
        out = torch.empty(l1 - l0, dtype=torch.uint8, device=self.device)
        eng.check(eng.lib().b3m_dist_l_bytes(self._h, a, l0, l1, eng._ptr(out), eng._stream()))
        return out

    def r_bytes(self, parts, q0, q1):
        """BWT(R)[q0, q1) out of R's packed slices (the merge's one all-gather brought them with L's)"""
        eng = self.engine
        out = torch.empty(q1 - q0, dtype=torch.uint8, device=self.device)
        if q1 > q0:
            k = len(parts)
            ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t, _ in parts])
            syms = (ctypes.c_uint64 * k)(*[int(c) for _, c in parts])
            eng.check(eng.lib().b3m_dist_r_bytes(self._h, ptrs, syms, k, q0, q1, eng._ptr(out), eng._stream()))
        return out

    def sort_u32(self, keys, nbits):
        eng = self.engine
        out = torch.empty_like(keys)
        if keys.numel():
            eng.check(eng.lib().b3m_dist_sort_u32(self._h, eng._ptr(keys), keys.numel(), nbits, eng._ptr(out), eng._stream()))
        return out

    def interleave(self, sr_local, bwtL, bwtR):
        eng = self.engine
        m, mr = bwtL.numel(), bwtR.numel()
        out = torch.empty(m + mr, dtype=torch.uint8, device=self.device)
        if m + mr:
            eng.check(eng.lib().b3m_dist_interleave(self._h, eng._ptr(sr_local), eng._ptr(bwtL), m, eng._ptr(bwtR), mr, eng._ptr(out), eng._stream()))
        return out


# ---------------------------------------------------------------------------------------------------
# collectives helpers
# ---------------------------------------------------------------------------------------------------
def _staged(t):
    """gloo cannot move device tensors: with a gloo process group and tensors on the GPU (the 1-GPU
    rehearsal of the N>1 path) collectives are staged through host memory.  Transport only."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_gather_i64(vals, group, device):
    """every member contributes a small int64 vector; returns a (members, len) numpy array"""
    t = torch.tensor(vals, dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else device)
    gs = dist.get_world_size(group)
    out = [torch.empty_like(t) for _ in range(gs)]
    dist.all_gather(out, t, group=group)
    return torch.stack(out).cpu().numpy()


def _all_gather_padded(mine, chunk, group):
    """ONE all-gather of every member's 1-D uint8 tensor, each padded to `chunk` bytes: -> (members, chunk) tensor on mine's device"""
    gs = dist.get_world_size(group)
    pad = torch.empty(chunk, dtype=mine.dtype, device=mine.device)
    pad[: mine.numel()] = mine
    if _staged(pad):
        h = pad.cpu()
        parts = [torch.empty_like(h) for _ in range(gs)]
        dist.all_gather(parts, h, group=group)
        return torch.stack(parts).to(mine.device)
    out = torch.empty(gs * chunk, dtype=mine.dtype, device=mine.device)
    if dist.get_backend(group) == "gloo":
        parts = list(out.view(gs, chunk).unbind(0))
        dist.all_gather(parts, pad, group=group)
    else:
        dist.all_gather_into_tensor(out, pad, group=group)
    return out.view(gs, chunk)


class _PendingA2A:
    """an all-to-all(v) in flight (dist.all_to_all_single(async_op=True)): RCCL runs it on its own stream while this rank's next kernels
    (the next tile of the chain walk) run on the compute stream; result() waits and hands the received elements over"""

    def __init__(self, send, send_counts, recv_counts, group):
        self.dev = send.device
        self.staged = _staged(send)
        self.src = (send.cpu() if self.staged else send).contiguous()          # kept alive until the collective is through
        self.recv = torch.empty(int(sum(recv_counts)), dtype=send.dtype, device=self.src.device)
        self.work = dist.all_to_all_single(self.recv, self.src, output_split_sizes=[int(x) for x in recv_counts],
                                           input_split_sizes=[int(x) for x in send_counts], group=group, async_op=True)

    def result(self):
        self.work.wait()
        self.src = None
        return self.recv.to(self.dev) if self.staged else self.recv


def _broadcast(t, src, group):
    if _staged(t):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


def _all_reduce_sum(t, group=None):
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)


def _set_bit(words, x, value):
    w, b = x >> 5, x & 31
    mask = (1 << b)
    mask = mask - (1 << 32) if mask >= (1 << 31) else mask
    if value:
        words[w] |= mask
    else:
        words[w] &= ~mask


def _m32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def _replace_bits_seg(dst, seg, seg_w0, b, e):
    """dst bits [b,e) = bits [b,e) of `seg`, a word range starting at word seg_w0; int32 word
    tensors indexed by absolute position.  Edge words are merged under a mask."""
    if e <= b:
        return
    w0, w1 = b >> 5, (e - 1) >> 5
    for w in sorted({w0, w1}):
        lo, hi = max(b, w << 5) - (w << 5), min(e, (w + 1) << 5) - (w << 5)
        mk = _m32(((1 << hi) - 1) & ~((1 << lo) - 1))
        dst[w] = (dst[w] & ~mk) | (seg[w - seg_w0] & mk)
    if w1 - w0 >= 2:
        dst[w0 + 1:w1] = seg[w0 + 1 - seg_w0:w1 - seg_w0]


# ---------------------------------------------------------------------------------------------------
# the distributed hot path
# ---------------------------------------------------------------------------------------------------
def _exchange_words(send_parts, recv_counts, group):
    """one all-to-all(v) of int32 word segments: send_parts[j] goes to member j (empty tensors allowed); -> list of the segments received,
    recv_counts[j] words from member j (every member can compute both sides from the plan)"""
    send = torch.cat(send_parts) if any(t.numel() for t in send_parts) else send_parts[0][:0]
    got = _PendingA2A(send.contiguous(), [t.numel() for t in send_parts], recv_counts, group).result()
    out, o = [], 0
    for c in recv_counts:
        out.append(got[o:o + int(c)])
        o += int(c)
    return out


class DistributedBwt:
    """construct the BWT of a replicated packed text on dist.get_world_size() GPUs by balanced block merging over `leaves` >= world leaf
    blocks (rank g sorts blocks g, g + P, ...).

    run() returns this rank's contiguous slice of the BWT, its offset, and p0 (rank of the rotation starting at 0).  Groups are
    created once (new_group is collective over ALL ranks; one per distinct rank set); the backend's workspace is allocated once, in __init__.

    gt bits: the HOLDER of a node (the owner of its first leaf) keeps the node's bits.  In a merge the holder of R sends every walker the
    words of its own chain range (one all-to-all(v): |R| / 8 bytes in all, not |L| / 8 + |R| / 8 to every member), the walkers send the
    new bits of their ranges to the holder of the parent (= the holder of L, whose bits stay where they are)."""

    def __init__(self, n, itype, backend, device, leaves=None):
        self.n, self.itype, self.be, self.device = n, itype, backend, torch.device(device)
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.plan = TreePlan(n, self.world, leaves)
        self.groups, by_set = {}, {}
        for mg in self.plan.merges:          # same order on every rank
            key = tuple(mg.group)
            if key not in by_set:
                by_set[key] = dist.new_group(ranks=mg.group)
            self.groups[mg.idx] = by_set[key]
        self._group_sets = by_set
        mine = [mg for mg in self.plan.merges if self.rank in mg.group]
        my_leaves = self.plan.leaves_of(self.rank)
        assert my_leaves, "every rank owns a leaf block"
        self.be.setup(n, itype, max(self.plan.leaves[k].m for k in my_leaves), max([mg.L.m for mg in mine] + [1]), max([mg.R.m for mg in mine] + [1]))
        self.timing = {}
        self.sync_phases = False           # a device synchronisation at every phase border: the phase split of ONE extra step of the bench, never the timed steps

    def warm_communicators(self):
        """first use of every group this rank belongs to (RCCL creates a communicator on a group's first collective): a tiny all-reduce each, in the
        same order on every rank, so that the timed steps and their watchdog see collectives, not communicator setup"""
        dev = "cpu" if dist.get_backend() == "gloo" else self.device
        for key, grp in self._group_sets.items():
            if self.rank in key:
                t = torch.ones(1, dtype=torch.int64, device=dev)
                dist.all_reduce(t, group=grp)
                assert int(t[0]) == len(key)

    def _now(self):
        if self.sync_phases and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return time.perf_counter()

    def _symstart(self, leaves):
        """C array of a node's text: symbols smaller than code c (pacterm: code c = symbol c + 1, the terminator is smaller than every base)"""
        h = self.leaf_hist[leaves].sum(axis=0)
        if self.itype == "bytestream":
            return [int(x) for x in np.concatenate([[0], np.cumsum(h[:255])])]
        if self.itype == "pacterm":
            return [int(h[0] + h[1:1 + c].sum()) for c in range(4)]
        return [int(h[:c].sum()) for c in range(4)]

    def run(self, d_text):
        n, be, plan, me = self.n, self.be, self.plan, self.rank
        dev = self.device
        tm = self.timing = {"leaf": 0.0, "anchors": 0.0, "gather": 0.0, "dict": 0.0, "walk": 0.0, "all_to_all": 0.0, "sort": 0.0, "interleave": 0.0}
        be.set_text(d_text)
        # ---- my leaf blocks g, g + P, ...: sort, the anchors (z-ranks) of every merge the leaf is on the left of, histogram ----
        # (the ranks never visit the host: partial ranks are added on the device, summed over the ranks by one all-reduce, and the walks read them there)
        anchors = torch.zeros(max(1, plan.total_anchors), dtype=torch.int64, device=dev)
        slices, node_p0, hists = {}, {}, []       # node key -> (my slice of its BWT, offset of the slice inside the node's BWT)
        gt_cur = None
        for k in plan.leaves_of(me):
            t0 = self._now()
            leaf = plan.leaves[k]
            bw, gt_leaf, p0, h5 = be.leaf(leaf.a, leaf.m)
            if gt_cur is None:
                gt_cur = gt_leaf                  # absolute-position bit array: this rank's leaves and the nodes it holds occupy disjoint ranges of it
            else:
                be.bits_replace(gt_cur, gt_leaf, 0, leaf.a + 1, leaf.a + leaf.m + 1)
            slices[leaf.key] = (bw, 0)
            node_p0[leaf.key] = p0
            hists.append([k] + [int(x) for x in h5])
            t1 = self._now()
            tm["leaf"] += t1 - t0
            for mg in plan.merges_with_leaf_on_left(k):
                be.rank_of_into(plan.anchor_positions(mg), anchors[mg.anchor_off:mg.anchor_off + mg.num_chains])
            tm["anchors"] += self._now() - t1
        t1 = self._now()
        _all_reduce_sum(anchors)
        per = (plan.B + self.world - 1) // self.world
        width = len(hists[0])
        table = np.full((per, width), -1, dtype=np.int64)
        table[:len(hists)] = np.array(hists, dtype=np.int64)
        allh = _all_gather_i64(table.reshape(-1).tolist(), None, dev).reshape(-1, width)
        self.leaf_hist = np.zeros((plan.B, width - 1), dtype=np.int64)
        for row in allh:
            if row[0] >= 0:
                self.leaf_hist[int(row[0])] = row[1:]
        tm["anchors"] += self._now() - t1
        # ---- merges, level by level, in the same order on every rank -----------------------------------------
        for mg in plan.merges:
            if me in mg.group:
                gt_cur = self._merge(mg, slices, node_p0, gt_cur, anchors, tm)
        out, off = slices[plan.root.key]
        return out, off, node_p0[plan.root.key]

    def _merge(self, mg, slices, node_p0, gt_cur, anchors, tm):
        """one merge of the tree, this rank's part.  What the host sees of it: slice sizes and the split sizes of the all-to-alls (and the
        new p0 at the end); anchors, gt bits, ranks and symbols stay on the device.  Collectives per merge: one small all-gather (slice
        sizes, p0), ONE all-gather of the packed slices of both sides, one all-to-all(v) of gt words (R's holder -> the walkers' chain ranges),
        one small all-gather of the bucket sizes per tile, one all-to-all(v) of the gap buckets per tile, one all-to-all(v) of gt words (walkers ->
        the parent's holder), one small all-gather (new p0, offsets)."""
        n, be, me, dev = self.n, self.be, self.rank, self.device
        grp = self.groups[mg.idx]
        members = mg.group
        gs = len(members)
        gi = members.index(me)
        a, m, mr = mg.L.a, mg.L.m, mg.R.m
        rbeg, rend = a + m, a + m + mr
        t0 = self._now()
        # 1. slice sizes of both sides and L's p0 (the symbol of L's p0 entry, which the packed form cannot carry, is the text symbol
        #    in front of L: the library reads it there).  A member may hold a slice of L, of R, or (more leaves than ranks) of both.
        myL, myR = slices.pop(mg.L.key, None), slices.pop(mg.R.key, None)
        empty = torch.empty(0, dtype=torch.uint8, device=dev)
        bwL, bwR = (myL[0] if myL else empty), (myR[0] if myR else empty)
        info = _all_gather_i64([bwL.numel(), bwR.numel(), node_p0.get(mg.L.key, -1) if myL else -1], grp, dev)
        sizesL, sizesR = info[:, 0], info[:, 1]
        p0L = int(info[:, 2].max())
        assert int(sizesL.sum()) == m and int(sizesR.sum()) == mr and p0L >= 0, "slice tables inconsistent"
        node_p0.pop(mg.L.key, None); node_p0.pop(mg.R.key, None)
        # 2. the packed slices of BOTH sides to every member in ONE all-gather (2 bits per symbol; bytes: as they are).  L's slices become the rank
        #    dictionary where it is used; an owner later takes its range of BWT(R) out of R's slices, so no second exchange of symbols is needed.
        #    (Padded to the largest slice: a member's share of a node differs from the others' by the rounding of the rank slices.)
        chunkL = max(1, max(be.packed_len(int(c)) for c in sizesL))
        chunkR = max(1, max(be.packed_len(int(c)) for c in sizesR))
        both = torch.zeros(chunkL + chunkR, dtype=torch.uint8, device=dev)
        if bwL.numel():
            pk = be.pack2(bwL); both[:pk.numel()] = pk
        if bwR.numel():
            pk = be.pack2(bwR); both[chunkL:chunkL + pk.numel()] = pk
        allpk = _all_gather_padded(both, chunkL + chunkR, grp)
        partsL = [(allpk[j, :chunkL], int(sizesL[j])) for j in range(gs) if sizesL[j]]
        partsR = [(allpk[j, chunkL:], int(sizesR[j])) for j in range(gs) if sizesR[j]]
        del bwL, bwR, myL, myR, both
        # 3. gt bits of R's positions: R's holder has them; member j walks the chains [cb[j], cb[j+1]) = the positions (rbeg + klo_j, rbeg + khi_j]
        #    and gets the words of that range only
        cb = split_even(mg.num_chains, gs)
        def chain_bits(j):                     # absolute bit range [b, e) member j's walk reads (one bit of margin on both sides, inside R)
            khi = mr - cb[j] * mg.chain_len
            klo = max(0, mr - cb[j + 1] * mg.chain_len)
            if cb[j + 1] == cb[j]:
                return rbeg, rbeg
            return max(rbeg, rbeg + klo - 1), min(rend + 1, rbeg + khi + 2)
        def words(b, e):
            return (b >> 5, ((e - 1) >> 5) + 1) if e > b else (0, 0)
        hR, hP = members.index(mg.R.holder), members.index(mg.L.holder)
        need = [words(*chain_bits(j)) for j in range(gs)]
        noseg = torch.empty(0, dtype=torch.int32, device=dev)
        send = [gt_cur[w0:w1].contiguous() if gi == hR else noseg for (w0, w1) in need]
        got = _exchange_words(send, [(need[gi][1] - need[gi][0]) if j == hR else 0 for j in range(gs)], grp)
        if gi != hR and need[gi][1] > need[gi][0]:
            b, e = chain_bits(gi)
            be.bits_replace(gt_cur, got[hR], need[gi][0], b, e)
        t1 = self._now()
        tm["gather"] += t1 - t0
        # 4. dictionary, then my chain range; the ranks come back sorted and already cut at the borders of the group's rank slices
        be.build_dict(partsL, m, p0L)
        t1b = self._now()
        tm["dict"] += t1b - t1
        c0, c1 = cb[gi], cb[gi + 1]
        lam = split_even(m + 1, gs)           # owner j holds ranks [lam[j], lam[j+1])
        # B3M_DIST_TILES = t > 1: my chain range goes in t tiles; while tile i+1 walks, the gap buckets of tile i are already on their way
        # (RCCL runs the all-to-all on its own stream), at the price of one small all-gather of bucket sizes per tile.  Default: one tile,
        # one all-gather of the bucket sizes, one all-to-all(v).
        ntile = max(1, min(int(os.environ.get("B3M_DIST_TILES", "1")), max(1, c1 - c0)))
        tb = split_even(c1 - c0, ntile)
        symstart = self._symstart(mg.L.leaves)
        flights, rc, oseg = [], None, None
        w0, w1 = rbeg >> 5, (rend >> 5) + 1
        t_walk = t_a2a = 0.0
        for t in range(ntile):
            tw0 = self._now()
            ca, cz = c0 + tb[t], c0 + tb[t + 1]
            anc = anchors[mg.anchor_off + ca: mg.anchor_off + cz]
            sr_t, cut, gt_out = be.walk(a, m, mr, symstart, gt_cur, anc, ca, cz - ca, mg.chain_len, lam)
            seg = gt_out[w0:w1]
            oseg = seg.clone() if oseg is None else oseg | seg          # new gt bits of the walked positions (disjoint between tiles and walkers)
            del gt_out
            tw1 = self._now()
            t_walk += tw1 - tw0
            send_counts = np.diff(cut)
            rc_t = _all_gather_i64(list(send_counts), grp, dev)          # rc_t[i][j] = i sends to j in this tile
            flights.append(_PendingA2A(sr_t, send_counts, rc_t[:, gi], grp))
            rc = rc_t if rc is None else rc + rc_t
            t_a2a += self._now() - tw1
        t2 = self._now()
        tm["walk"] += t_walk
        # new gt bits of R's positions relative to S_a: member j made the bits of [rbeg + klo_j, rbeg + khi_j); they go to the parent's holder
        def made_bits(j):
            khi = mr - cb[j] * mg.chain_len
            klo = max(0, mr - cb[j + 1] * mg.chain_len)
            return (rbeg + klo, rbeg + khi) if cb[j + 1] > cb[j] else (rbeg, rbeg)
        made = [words(*made_bits(j)) for j in range(gs)]
        mw0, mw1 = made[gi]
        send = [oseg[mw0 - w0:mw1 - w0].contiguous() if (j == hP and mw1 > mw0) else noseg for j in range(gs)]
        got = _exchange_words(send, [(made[j][1] - made[j][0]) if gi == hP else 0 for j in range(gs)], grp)
        if gi == hP:
            for j in range(gs):
                b, e = made_bits(j)
                if e > b:
                    be.bits_replace(gt_cur, got[j], made[j][0], b, e)      # L's bits (a, rbeg) stay; R's become relative to S_a
            be.set_endbit(gt_cur, rend, anchors[mg.anchor_off: mg.anchor_off + 1], p0L)
        t2b = self._now()
        tm["gather"] += t2b - t2
        # 5. the gap buckets of my slice of the rank space [0, m], from every walker and tile
        got = torch.cat([f.result() for f in flights]) if len(flights) > 1 else flights[0].result()
        del flights
        t3 = self._now()
        tm["all_to_all"] += t_a2a + (t3 - t2b)
        nbits = max(1, int(m + 1).bit_length())
        sr_slice = be.sort_u32(got, min(32, nbits))
        t4 = self._now()
        tm["sort"] += t4 - t3
        # 6. the range of BWT(R) that falls into my slice, out of R's packed slices
        cj = rc.sum(axis=0)                   # cj[j] = suffixes of R owned by j
        Q = np.concatenate([[0], np.cumsum(cj)])
        bwtR_part = be.r_bytes(partsR, int(Q[gi]), int(Q[gi + 1]))
        # 7. interleave my slice; its range of BWT(L) comes out of the dictionary
        l0, l1 = lam[gi], (lam[gi + 1] if gi + 1 < gs else m)
        l1 = min(l1, m)
        sr_local = (sr_slice - l0) if l0 else sr_slice
        out = be.interleave(sr_local, be.l_bytes(a, l0, l1), bwtR_part)
        del partsL, partsR, allpk
        p0P, new_off = self._p0_after_merge(sr_slice, p0L, l0, l1, int(Q[gi]), out.numel(), gi, grp)
        slices[mg.parent.key] = (out, new_off)
        node_p0[mg.parent.key] = p0P
        tm["interleave"] += self._now() - t4
        return gt_cur

    def _p0_after_merge(self, sr_slice, p0L, l0, l1, q0, out_len, gi, grp):
        """new rank of L's p0 entry, p0L + #{q : sr[q] <= p0L} (known to the owner of rank p0L), and the offset of my slice of the parent:
        the one place after a merge's interleave where a number leaves the device"""
        cand = -1
        if l0 <= p0L < l1:
            le = int(torch.searchsorted(sr_slice, torch.tensor([p0L], dtype=torch.int32, device=sr_slice.device), right=True)[0])
            cand = p0L + q0 + le
        fin = _all_gather_i64([out_len, cand], grp, sr_slice.device)
        return int(fin[:, 1].max()), int(fin[:gi, 0].sum())


class PrefixShardedBwt:
    """Multi-GPU construction WITHOUT an exchange step, for a text that is resident on every GPU:
    rank k sorts only the suffixes whose rank falls into k's range (whole classes of the leading
    8 bits, balanced from the text's own histogram, so all ranks agree without communicating) and
    owns that contiguous slice of the BWT.  Strong scaling of a fixed input; nothing but the
    optional result gather touches the interconnect.  The block-merging path with its all-to-all of
    gap buckets (DistributedBwt) is what a text that cannot be sorted in one piece needs."""

    def __init__(self, n, itype, device, isasamplingrate=0, allow_fallback_without_samples=False):
        from . import engine
        self.isasamplingrate = int(isasamplingrate)
        self.allow_fallback_without_samples = allow_fallback_without_samples
        self.samples_dropped = False       # True once the fallback below has been taken by a job that asked for samples
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.n, self.itype, self.device = n, itype, device
        self.plan = engine.Plan(n, itype, isasamplingrate=isasamplingrate, shard=self.rank, nshards=self.world)
        self.out = torch.empty(self.plan.slice_capacity, dtype=torch.uint8, device=device)
        self.tie_finish_runs = 0           # steps that needed the collective finish of long repeats
        self.local_ties = False            # the union of the shards' lists did not fit a shard's workspace: every shard finishes its own repeats from then on
        if self.world > 1:
            self.plan.set_tie_mode(engine.Plan.TIES_LIST)
        self.fallback = None               # the block-merge driver, once some shard has refused its range (skewed text): see run()

    def run(self, d_text):
        """One step.  A shard whose range does not fit its workspace -- one level-0 digit of a skewed text heavier than a shard's share plus slack --
        refuses; all ranks then agree (one small all-gather, which every step makes anyway) to construct this text by block merging instead
        (DistributedBwt, the mode with the all-to-all of gap buckets), this step and every later one."""
        from . import _lib
        if self.fallback is not None:
            return self.fallback.run(d_text)
        refused = 0
        try:
            sl = self.plan.run(d_text, self.out)
        except _lib.B3MError as ex:
            if self.world == 1 or "suffix-range shard larger than the workspace" not in str(ex) or self.n >= (1 << 32):
                raise
            refused, sl = 1, self.out[:0]
        if self.world > 1:
            # one small all-gather per step: who refused, and (long repeats, row a6) what the local rounds left tied on any shard
            t, x, depth, complete = (0, 0, 0, True) if refused or self.local_ties else self.plan.list_counts()
            pend = _all_gather_i64([t, sl.numel(), refused, x, depth, 1 if complete else 0], None, self.out.device)
            if int(pend[:, 2].sum()) > 0:
                self.refusal = "suffix-range shards refused by rank(s) %s: block merging instead" % [int(i) for i in np.nonzero(pend[:, 2])[0]]
                if self.isasamplingrate and not self.allow_fallback_without_samples:
                    # the block-merge driver emits no (rank,pos) samples: a caller that asked for them must not get slices without them silently
                    raise _lib.B3MError(self.refusal + "; the block-merge mode emits no (rank,pos) samples (isasamplingrate=%d was asked for): "
                                        "construct with allow_fallback_without_samples=True to accept BWT slices alone" % self.isasamplingrate)
                self.plan.close()
                self.plan = None
                self.samples_dropped = bool(self.isasamplingrate)
                del self.out, sl
                torch.cuda.empty_cache()
                self.fallback = DistributedBwt(self.n, self.itype, HipBackend(self.device), self.device)
                return self.fallback.run(d_text)
            # ties (row a6): every shard kept a LIST of what its rounds left tied (and of what its direct comparisons placed among long repeats); the records -- 16 bytes
            # per listed suffix -- go to every rank in two all-gathers and every rank finishes the union by prefix doubling on the list (engine.finish_shard_lists).
            # A union too large for a shard's workspace: from this step on the shards finish their repeats themselves (key extension, no collective step)
            if int(pend[:, 0].sum()) > 0:
                from . import engine
                tied, placed = int(pend[:, 0].sum()), int(pend[:, 3].sum())
                fits = _all_gather_i64([1 if self.plan.list_fits(tied, placed) else 0], None, sl.device)
                if int(fits[:, 0].min()) == 0:
                    self.plan.set_tie_mode(engine.Plan.TIES_LOCAL)
                    self.local_ties = True
                    sl = self.plan.run(d_text, self.out)
                else:
                    depths = [int(d) for d, t in zip(pend[:, 4], pend[:, 0]) if t]
                    gathered = lambda mine: (_all_gather_v(mine[0].reshape(-1), 2 * pend[:, 0]).reshape(-1, 2), _all_gather_v(mine[1].reshape(-1), 2 * pend[:, 3]).reshape(-1, 2),
                                             min(depths), max(depths), bool(pend[:, 5].min()))
                    engine.finish_shard_lists(self.plan, sl, gathered)
                    self.tie_finish_runs += 1
        r = self.plan.result()
        off = r["slice_offset"] if self.world > 1 else 0
        return sl, off, (r["p0"] if r["has_p0"] else None)


def _no_plan(job):
    from . import _lib
    return _lib.B3MError("this job fell back from suffix-range shards to block merging (%s): it has BWT slices, but no shard plan, no (rank,pos) samples "
                         "and no per-kernel times" % job.refusal)


def _sharded_result(self):
    """the shard plan's b3m_result of the last step; after the fallback to block merging there is none (a clear error, not a NULL handle)"""
    if self.plan is None:
        raise _no_plan(self)
    return self.plan.result()


def _sharded_samples(self):
    if self.plan is None:
        raise _no_plan(self)
    return self.plan.samples()


PrefixShardedBwt.result = _sharded_result
PrefixShardedBwt.samples = _sharded_samples


def _all_gather_v(t, sizes):
    """concatenation of every rank's 1-D tensor (sizes[i] entries from rank i), on every rank"""
    mx = int(max(1, sizes.max()))
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    dev = pad.device
    if _staged(pad):
        pad = pad.cpu()
    parts = [torch.empty_like(pad) for _ in range(len(sizes))]
    dist.all_gather(parts, pad)
    return torch.cat([p[: int(sz)] for p, sz in zip(parts, sizes)]).to(dev)


def gather_result(slice_t, group=None):
    """collect the distributed BWT on every rank (tests / verification only)"""
    ws = dist.get_world_size(group)
    sizes = _all_gather_i64([slice_t.numel()], group, slice_t.device)[:, 0]
    mx = int(sizes.max())
    pad = torch.zeros(mx, dtype=slice_t.dtype, device=slice_t.device)
    pad[: slice_t.numel()] = slice_t
    dev = pad.device
    if _staged(pad):
        pad = pad.cpu()
    parts = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: int(s)] for p, s in zip(parts, sizes)]).to(dev)


# ---------------------------------------------------------------------------------------------------
# bench entry (called by bench.py when WORLD_SIZE > 1)
# ---------------------------------------------------------------------------------------------------
def _run_mode(job, d_text, steps, warmup, dev, fam_of=None):
    from . import benchutil
    for _ in range(warmup):
        job.run(d_text)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    fam = {}
    for _ in range(steps):
        out, off, p0 = job.run(d_text)
        if fam_of is not None:       # reads the HIP-event timeline of the finished run (b3m_plan_run is blocking)
            benchutil.add_kernel_times(fam, fam_of(job))
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    return float(dt[0]), out, off, p0, fam


def _check_slices(out, off, p0, d_text, n, itype, dev):
    """size-independent properties of a distributed result (what the N>1 bench can afford at full size): the slices tile the rank space,
    their symbol histograms add up to the text's (computed with torch alone), and for pacterm the slot p0 holds the one terminator.
    Collective; returns a description or raises."""
    from . import benchutil
    info = _all_gather_i64([int(off), int(out.numel())], None, dev)
    pos = 0
    for o, ln in sorted((int(x[0]), int(x[1])) for x in info):
        if o != pos:
            raise RuntimeError("slices do not tile the rank space: a slice starts at %d, expected %d" % (o, pos))
        pos += ln
    if pos != n:
        raise RuntimeError("slices cover %d of %d ranks" % (pos, n))
    h = torch.zeros(256, dtype=torch.int64, device=dev)
    for o in range(0, out.numel(), 1 << 28):
        h += torch.bincount(out[o:o + (1 << 28)].to(torch.int64), minlength=256)
    _all_reduce_sum(h)
    want = benchutil.text_histogram(d_text, n, itype)
    if h.cpu().tolist() != want:
        raise RuntimeError("symbol histogram of the slices differs from the text's")
    term = ""
    if itype == "pacterm":
        mine = p0 is not None and int(off) <= int(p0) < int(off) + out.numel()      # (a shard without the rotation starting at 0 reports None)
        ok = torch.tensor([1 if (mine and int(out[int(p0) - int(off)]) == 0) else 0], dtype=torch.int64, device=dev)
        _all_reduce_sum(ok)
        if int(ok[0]) != 1:
            raise RuntimeError("BWT[p0] is not the terminator")
        term = "; BWT[p0]=terminator"
    return "slices of all %d ranks tile [0, n); sum of their symbol histograms == histogram(text)%s" % (dist.get_world_size(), term)


def bench_main(a, rank, local_rank, world):
    """N>1 leg of bench.py: one process per GPU (torchrun), RCCL process group.  Headline = the mode asked for (default:
    suffix-range shards); the other mode runs too (one step) and both are in the ONE JSON line (`config.modes`).  A failure or a
    hung collective of the HEADLINE mode ends the process with a non-zero status and no line; one of the other mode is named in
    `config.modes[<mode>].error` and the headline line stands."""
    import sys
    import threading
    from . import benchutil, synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % max(1, ndev))
    torch.cuda.set_device(dev)
    # B3M_BENCH_BACKEND=gloo: rehearsal of the N>1 flow with several ranks on ONE GPU (RCCL needs a GPU
    # per rank); collectives are then staged through host memory
    backend = os.environ.get("B3M_BENCH_BACKEND", "nccl")
    # a peer that has left must show as a collective that does not complete (this module's own watchdog deals with that), not as an abort
    # of this process from the process group's error-handling thread
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    inject = os.environ.get("B3M_BENCH_INJECT", "")          # "<mode>:<rank>": that rank raises when it starts that mode (tests of the failure paths)
    if backend == "nccl":
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    itype_cfg, nsyms_cfg, _ = synth.CONFIGS[a.config]
    want = int(a.size) if a.size else nsyms_cfg
    itype, d_text, n = synth.config_input_device(a.config, size=want, device=dev)
    nbases = n - 1 if itype == "pacterm" else n
    mode = getattr(a, "mode", "prefix")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def _bail():
        sys.stderr.write("[bench] rank %d: watchdog fired after %d s (a collective never completed): exit 3\n" % (rank, a.blocks_timeout))
        sys.stderr.flush()
        os._exit(3)

    # the shards emit what the plan of b3m_compute_bwt emits (samples at b3m_plan_sampling_rate), as the one-GPU line does
    from . import _lib
    plan_rate = int(_lib.lib().b3m_plan_sampling_rate(getattr(a, "isasamplingrate", 0), getattr(a, "bwtonly", 0), n))

    # leaf blocks of the block-merge mode: --blocks B (>= world; rank g sorts blocks g, g + P, ...), default one per rank -- and four for a world of ONE rank
    # (`WORLD_SIZE=1 B3M_BENCH_BACKEND=nccl python bench.py --gpus 1`: the one-GPU box's way to run every collective of a merge through RCCL)
    nleaves = max(world, int(getattr(a, "blocks", 0) or 0)) if (getattr(a, "blocks", 0) or world > 1) else 4

    def make(m):
        # (a text too skewed for suffix-range shards is constructed by block merging instead; the line then says so and claims no samples)
        return (DistributedBwt(n, itype, HipBackend(dev), dev, leaves=nleaves) if m == "blocks"
                else PrefixShardedBwt(n, itype, dev, isasamplingrate=plan_rate, allow_fallback_without_samples=True))

    def guarded(seconds, bail, fn):
        wd = threading.Timer(seconds, bail)
        wd.daemon = True
        wd.start()
        try:
            return fn()
        finally:
            wd.cancel()

    def measure(m, steps, warmup, bail=None):
        bail = bail or _bail
        if inject == "%s:%d" % (m, rank):
            raise RuntimeError("injected failure (B3M_BENCH_INJECT=%s)" % inject)
        # plan + workspace, and the first use of every process group this mode needs (RCCL sets a communicator up on a group's first collective:
        # seconds each on a fresh node), under a watchdog of their own: the steps' watchdog then times collectives, not their setup
        def setup():
            job = make(m)
            if m == "blocks":
                job.warm_communicators()
            return job
        job = guarded(2 * a.blocks_timeout, bail, setup)

        def steps_():
            dt, out, off, p0, fam = _run_mode(job, d_text, steps, warmup, dev, fam_of=(lambda j: j.plan.kernel_times() if getattr(j, "fallback", None) is None else []) if m != "blocks" else None)
            checked = _check_slices(out, off, p0, d_text, n, itype, dev) if getattr(a, "check", True) else None
            del out
            merger = job if m == "blocks" else getattr(job, "fallback", None)
            phases, levels = {}, 0
            if merger is not None:
                # the phase split comes from ONE extra step with a device synchronisation at every phase border (the timed steps have none)
                merger.sync_phases = True
                job.run(d_text)
                merger.sync_phases = False
                phases = {k: round(v, 4) for k, v in merger.timing.items()}
                levels = len(merger.plan.levels)
            if getattr(job, "fallback", None) is not None:      # a shard refused its range (skewed text): the steps ran as block merges
                fam = {}
            return {"dt": dt, "steps": steps, "fam": fam, "phases": phases, "levels": levels, "job": job, "checked": checked,
                    "tie_finish_runs": getattr(job, "tie_finish_runs", 0), "refusal": getattr(job, "refusal", None),
                    "samples_dropped": getattr(job, "samples_dropped", False)}
        return guarded(a.blocks_timeout, bail, steps_)

    other = "blocks" if mode != "blocks" else "prefix"
    try:
        if a.verify:
            raise SystemExit("bench.py --verify with --gpus > 1: use tests/ (the oracle is test infrastructure, not part of the package)")
        res_main = measure(mode, a.steps, a.warmup)
        res_main.pop("job", None)
    except BaseException as ex:   # noqa: BLE001
        # peers may be stuck in a collective this rank has left: no barrier, no teardown -- fail at once
        sys.stderr.write("[bench] rank %d failed: %r\n" % (rank, ex))
        sys.stderr.flush()
        os._exit(2)

    reported = threading.Event()

    def report(res_other):
        """rank 0's ONE JSON line; res_other: the other mode's measurement, None (not run) or {"error": ...}"""
        def summary(r):
            if "error" in r:
                return r
            out = {"ms_per_step": round(r["dt"] / r["steps"] * 1e3, 3), "value": round(nbases * r["steps"] / r["dt"] / 1e9, 4), "steps": r["steps"],
                   "merge_levels": r["levels"], "checked": r["checked"], "tie_finish_runs": r["tie_finish_runs"]}
            if r["phases"]:
                out["rank0_phase_s"] = r["phases"]
                out["rank0_phase_s_what"] = ("one extra step with a device synchronisation at every phase border: leaf sort, anchors (z-ranks), gather (all-gather of both "
                                             "sides' packed slices + gt words), dict (rank dictionary of BWT(L)), walk (LF chains), all_to_all (gap buckets), sort, interleave")
            if r.get("refusal"):
                out["fell_back"] = r["refusal"]
            return out
        dt, steps, fam, levels = res_main["dt"], res_main["steps"], res_main["fam"], res_main["levels"]
        alg_job = nbases * (9.5 + 41.0 * levels)
        if fam:
            # dominant kernel of rank 0's shard (every rank runs the same kernels on its own suffix range)
            roof = benchutil.dominant_roofline(fam, steps, dt, nbases, levels, 0, root, peak_gpus=world, shards=world if mode == "prefix" else 1)
            roof["rank"] = 0
        else:
            roof = {"bound": "hbm", "kernel": "whole job (SURVEY.md 8d alg_bytes(n,B))", "achieved": round(alg_job / (dt / steps) / 1e9, 1),
                    "peak": 8000.0 * world, "unit": "GB/s", "frac": round(alg_job / (dt / steps) / 1e9 / (8000.0 * world), 5), "traffic": None}
        names = {"blocks": "%d GPU(s): %d leaf blocks (rank g sorts blocks g, g + P, ...), balanced merges, %s all-to-all of gap buckets" % (world, nleaves, "RCCL" if backend == "nccl" else backend),
                 "prefix": "%d GPUs: suffix-range shards of the replicated text, no exchange step" % world}
        modes = {mode: summary(res_main)}
        # only suffix-range shards emit samples; a shard job that fell back to block merging emitted none
        with_samples = mode == "prefix" and not res_main.get("refusal")
        if res_other is not None:
            modes[other] = summary(res_other)
        res = {
            "metric": "Gbase/s BWT construction (3 Gbase pacterm) at 1/2/4/8 MI355X; % HBM roofline",
            "value": round(nbases * steps / dt / 1e9, 4), "unit": "Gbase/s", "n_gpus": world, "steps": steps, "warmup": a.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: %d-symbol synthetic %s" % (a.config.upper(), n, itype), "blocks": nleaves if mode == "blocks" else 1, "backend": backend,
                       "merge_levels": levels, "mode": mode, "parallelism": names[mode], "modes": modes, "checked": res_main["checked"],
                       "value_is": "device-resident: packed text replicated in every GPU's HBM -> BWT slices%s in HBM" % (" + (rank,pos) samples" if with_samples else "")},
            "roofline": roof,
        }
        if with_samples:
            res["config"]["isasamplingrate"] = {"plan": plan_rate, "what": "suffix-range shards emit the (rank,pos) samples of the plan b3m_compute_bwt creates (b3m_plan_sampling_rate)"}
        elif mode == "prefix":
            res["config"]["isasamplingrate"] = {"plan": 0, "what": "the shards fell back to block merging, which emits no (rank,pos) samples"}
        if reported.is_set():          # (the watchdog of the other mode and the main thread can both get here: one line)
            return
        reported.set()
        print(json.dumps(res), flush=True)

    # the other mode (one step) rides in the same line.  It must not be able to take the headline with it: if it fails or a collective of it never
    # completes (it has never run on this many real GPUs before the driver's own run), rank 0 still prints the line, with the failure named in
    # `config.modes`, and every rank leaves at once with status 0 (peers may be stuck in a collective: no barrier, no teardown)
    res_other = None
    if getattr(a, "also_blocks", True) and n < (1 << 32):
        def _bail_other():
            sys.stderr.write("[bench] rank %d: mode %s: watchdog fired after %d s; the headline line stands\n" % (rank, other, a.blocks_timeout))
            sys.stderr.flush()
            if rank == 0:
                report({"error": "a collective of this mode did not complete within %d s" % a.blocks_timeout})
            os._exit(0)
        try:
            torch.cuda.empty_cache()
            res_other = measure(other, max(3, min(a.steps, 5)), 1, bail=_bail_other)
            res_other.pop("job", None)
        except BaseException as ex:   # noqa: BLE001
            sys.stderr.write("[bench] rank %d: mode %s failed: %r; the headline line stands\n" % (rank, other, ex))
            sys.stderr.flush()
            if rank == 0:
                report({"error": repr(ex)[:300]})
            os._exit(0)
    if rank == 0:
        report(res_other)
    sys.stdout.flush()
    # the line is out; a peer that left early (its watchdog, an error of the other mode) must not hold this rank in the last barrier for ever
    def _leave():
        os._exit(0)
    wd = threading.Timer(30, _leave)
    wd.daemon = True
    wd.start()
    dist.barrier()
    dist.destroy_process_group()
    wd.cancel()
