"""Device-level API over libb3m: plans and the stage entry points of include/b3m.h.

torch is used only as plumbing: device buffers and the current HIP stream.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import INPUT_TYPES, check, lib


def _type_id(inputtype):
    if isinstance(inputtype, int):
        return inputtype
    t = lib().b3m_parse_input_type(inputtype.encode())
    if t < 0:
        raise _lib.B3MError(lib().b3m_last_error().decode())
    return t


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def num_symbols(file_bytes, inputtype):
    """getFileSize of the input wrappers: symbols in a packed input"""
    t = _type_id(inputtype)
    nb = len(file_bytes)
    if t == INPUT_TYPES["bytestream"]:
        return nb
    if t in (INPUT_TYPES["pac"], INPUT_TYPES["pacterm"]):
        if nb < 2 or int(file_bytes[-1]) > 3:
            raise _lib.B3MError("not a .pac file")
        l = (nb - 2) * 4 + int(file_bytes[-1])
        return l + (1 if t == INPUT_TYPES["pacterm"] else 0)
    if t == INPUT_TYPES["compactstream"]:
        return compact_info(file_bytes)[1]
    raise _lib.B3MError("input type not supported by the GPU engine yet")


class Plan:
    """b3m_plan: all workspace allocated at creation; run() = one pass of the hot path."""

    def __init__(self, n, inputtype, blocksize=0, mem=0, isasamplingrate=0, shard=0, nshards=1):
        self.n = int(n)
        self.type = _type_id(inputtype)
        self.nshards = int(nshards)
        if self.nshards > 1:
            self._h = lib().b3m_plan_create_sharded(self.n, self.type, int(shard), self.nshards, int(mem), int(isasamplingrate))
        else:
            self._h = lib().b3m_plan_create(self.n, self.type, int(blocksize), int(mem), int(isasamplingrate))
        if not self._h:
            raise _lib.B3MError(lib().b3m_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().b3m_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        if lib is not None:          # (module globals are gone when the interpreter shuts down)
            self.close()

    @property
    def num_blocks(self):
        return int(lib().b3m_plan_num_blocks(self._h))

    @property
    def host_tier(self):
        """0, or the number of sequential shards of a plan in the host tier (b3m_plan_host_tier)"""
        return int(lib().b3m_plan_host_tier(self._h))

    @property
    def warning(self):
        """'' or what the caller should be told (a mem= target that cannot be met: b3m_plan_warning)"""
        return lib().b3m_plan_warning(self._h).decode()

    @property
    def workspace_bytes(self):
        return int(lib().b3m_plan_workspace_bytes(self._h))

    @property
    def text_bytes(self):
        return int(lib().b3m_plan_text_bytes(self._h))

    @property
    def slice_capacity(self):
        return int(lib().b3m_plan_slice_capacity(self._h))

    def text_buffer(self):
        """the plan's own resident text buffer as a uint8 CUDA tensor (a view, b3m_plan_text_buffer): a text copied in here and passed to
        run() is used in place"""
        ptr = lib().b3m_plan_text_buffer(self._h)

        class _View:
            __cuda_array_interface__ = {"shape": (self.text_bytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(_View(), device="cuda")

    def run(self, d_text, d_out=None):
        """one pass of the hot path.  Sharded plans return the view of their slice (see result()).  symbols32 plans: text and result are
        uint8 views of n uint32 words (4n bytes each)."""
        assert d_text.is_cuda and d_text.dtype == torch.uint8 and d_text.numel() >= self.text_bytes
        cap = self.slice_capacity * (4 if self.type == _lib.SYMBOLS32 else 1)
        if d_out is None:
            d_out = torch.empty(cap, dtype=torch.uint8, device=d_text.device)
        assert d_out.is_cuda and d_out.numel() >= cap
        check(lib().b3m_plan_run(self._h, _ptr(d_text), _ptr(d_out), _stream()))
        if self.nshards > 1:
            return d_out[: self.result()["slice_len"]]
        return d_out

    def run_host(self, h_text, h_out, packed=False):
        """the same pass from host memory to host memory (b3m_plan_run_host): h_text / h_out are CPU uint8 tensors, pinned
        for asynchronous copies.  packed: the BWT arrives 2 bits per symbol in the .pac layout ((n+3)//4 bytes, b3m_plan_set_packed_output).
        Returns the HIP-event / wall-clock splits of the run."""
        check(lib().b3m_plan_set_packed_output(self._h, 1 if packed else 0))
        assert not h_text.is_cuda and not h_out.is_cuda and h_text.numel() >= self.text_bytes
        assert h_out.numel() >= ((self.n + 3) // 4 if packed else self.n)
        t = _lib.E2ETimes()
        check(lib().b3m_plan_run_host(self._h, _ptr(h_text), _ptr(h_out), ctypes.byref(t), _stream()))
        return {k: getattr(t, k) for k, _ in _lib.E2ETimes._fields_}

    @staticmethod
    def pack_pac(bwt, pacterm):
        """what a packed hand-over of the BWT tensor `bwt` (uint8 symbols, any device) holds: see b3m_plan_set_packed_output"""
        v = bwt.to(torch.int16)
        if pacterm:
            v = torch.clamp(v - 1, min=0)
        n = v.numel()
        v = torch.nn.functional.pad(v & 3, (0, (-n) % 4)).view(-1, 4)
        return ((v[:, 0] << 6) | (v[:, 1] << 4) | (v[:, 2] << 2) | v[:, 3]).to(torch.uint8)

    # ---- suffix-range shards and long repeats (b3m_plan_tie_*): see finish_shard_ties() ----
    def keep_sa(self, on=True):
        check(lib().b3m_plan_set_keep_sa(self._h, 1 if on else 0))

    def has_sa(self):
        return bool(lib().b3m_plan_has_sa(self._h))

    def tie_pending(self):
        c = ctypes.c_uint64(0)
        check(lib().b3m_plan_tie_pending(self._h, ctypes.byref(c)))
        return int(c.value)

    def tie_export(self, device):
        """-> (this shard's suffix array slice (int32 tensor, rank order), its tied slots as (global slot, global label) pairs (int32 [k, 2]))"""
        r = self.result()
        k = self.tie_pending()
        sa = torch.empty(max(1, r["slice_len"]), dtype=torch.int32, device=device)
        pairs = torch.empty((max(1, k), 2), dtype=torch.int32, device=device)
        check(lib().b3m_plan_tie_export(self._h, _ptr(sa), _ptr(pairs), _stream()))
        return sa[: r["slice_len"]], pairs[:k]

    def tie_finish(self, sa_global, pairs_all, my_first, d_bwt_slice):
        check(lib().b3m_plan_tie_finish(self._h, _ptr(sa_global), _ptr(pairs_all), pairs_all.shape[0], int(my_first), _ptr(d_bwt_slice), _stream()))

    # ---- the same bound from lists instead of suffix arrays (include/b3m.h: B3M_TIES_LIST): see finish_shard_lists() ----
    TIES_SA, TIES_LIST, TIES_LOCAL = 0, 1, 2

    def set_tie_mode(self, mode):
        check(lib().b3m_plan_set_tie_mode(self._h, int(mode)))

    def list_counts(self):
        """-> (tied, placed, depth, complete) of this shard's last run in list mode"""
        t, x, d, c = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_int(0)
        check(lib().b3m_plan_list_counts(self._h, ctypes.byref(t), ctypes.byref(x), ctypes.byref(d), ctypes.byref(c)))
        return int(t.value), int(x.value), int(d.value), bool(c.value)

    def list_fits(self, tied_total, placed_total):
        return bool(lib().b3m_plan_list_fits(self._h, int(tied_total), int(placed_total)))

    def list_export(self, device):
        """-> (tied records, placed records): int64 tensors [k, 2] = (global rank | first-of-group << 63, position)"""
        t, x, _, _ = self.list_counts()
        tied = torch.empty((max(1, t), 2), dtype=torch.int64, device=device)
        placed = torch.empty((max(1, x), 2), dtype=torch.int64, device=device)
        check(lib().b3m_plan_list_export(self._h, _ptr(tied), _ptr(placed), _stream()))
        return tied[:t], placed[:x]

    def list_finish(self, tied_all, placed_all, depth_min, depth_max, complete, d_bwt_slice):
        check(lib().b3m_plan_list_finish(self._h, _ptr(tied_all), tied_all.shape[0], _ptr(placed_all), placed_all.shape[0], int(depth_min), int(depth_max),
                                         1 if complete else 0, _ptr(d_bwt_slice), _stream()))

    def result(self):
        r = _lib.Result()
        check(lib().b3m_plan_result(self._h, ctypes.byref(r)))
        return {k: getattr(r, k) for k, _ in _lib.Result._fields_}

    def samples(self):
        ns = self.result()["num_samples"]
        a = np.zeros((ns, 2), dtype=np.uint64)
        if ns:
            check(lib().b3m_plan_samples(self._h, a.ctypes.data, ns))
        return a

    def histogram(self):
        h = np.zeros(256, dtype=np.uint64)
        check(lib().b3m_plan_histogram(self._h, h.ctypes.data))
        return h

    def lf_walk(self, d_bwt, sarate=0):
        """checkbwt + sampled SA on the device: returns (symbol mismatches,
        chain mismatches, SA samples SA[0], SA[sarate], ... as uint64 numpy array or None)"""
        mm = np.zeros(2, dtype=np.uint64)
        sa = np.zeros((self.n + sarate - 1) // sarate, dtype=np.uint64) if sarate else None
        check(lib().b3m_plan_lf_walk(self._h, _ptr(d_bwt), int(sarate), sa.ctypes.data if sarate else None, mm.ctypes.data, _stream()))
        return int(mm[0]), int(mm[1]), sa

    def kernel_times(self):
        arr = (_lib.KernelTime * 16)()
        check(lib().b3m_plan_kernel_times(self._h, arr, 16))
        out = []
        for k in arr:
            if k.name:
                out.append({"name": k.name.decode(), "ms": k.ms, "launches": int(k.launches), "alg_bytes": int(k.alg_bytes)})
        return out


def compact_info(file_bytes):
    """(bits per symbol, number of symbols, offset of the word stream) of a compactstream file image (b3m_compact_info)"""
    fb = np.ascontiguousarray(file_bytes[:32], dtype=np.uint8)
    bits, n, off = ctypes.c_uint32(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(lib().b3m_compact_info(fb.ctypes.data, len(file_bytes), ctypes.byref(bits), ctypes.byref(n), ctypes.byref(off)))
    return int(bits.value), int(n.value), int(off.value)


def compact_to_plan_input(file_bytes, device="cuda"):
    """compactstream file image -> (inputtype to plan with, device tensor to run it on, n): 2-bit files are a .pac payload in all but the
    header, other widths are decoded to one byte per symbol on the device (b3m_compact_decode)"""
    bits, n, off = compact_info(file_bytes)
    fb = np.ascontiguousarray(file_bytes, dtype=np.uint8)
    words = torch.zeros(len(fb) - off + 16, dtype=torch.uint8, device=device)       # padded: the pac plan copies n/4 + 2 bytes
    words[: len(fb) - off] = torch.from_numpy(fb[off:].copy()).to(device)
    if bits == 2:
        return "pac", words, n
    out = torch.empty(max(n, 1), dtype=torch.uint8, device=device)
    check(lib().b3m_compact_decode(_ptr(words), n, bits, _ptr(out), _stream()))
    return "bytestream", out[:n], n


def rerun_for_ties(plan, d_text, out):
    """a shard that ran BWT-only kept no suffix array: sort it once more with the array kept (only when some shard reports long repeats)"""
    if not plan.has_sa():
        plan.keep_sa(True)
        plan.run(d_text, out)
    return plan


def finish_shard_ties(plan, out_slice, gather_sa, gather_pairs):
    """Suffix-range shards of a repeat-rich text: finish what the shards' local rounds left tied (row a6).  gather_sa(slice) -> every shard's
    suffix array slice concatenated in rank order (n entries); gather_pairs(pairs) -> (all shards' pairs concatenated in shard order, index of
    this shard's first pair, total).  Called by EVERY shard (collectively) when any shard has plan.tie_pending() > 0; a shard whose BWT-only
    run kept no suffix array must have been run again with plan.keep_sa() first (rerun_for_ties)."""
    sa, pairs = plan.tie_export(out_slice.device)
    sa_global = gather_sa(sa)
    pairs_all, my_first = gather_pairs(pairs)
    if pairs_all.shape[0]:
        plan.tie_finish(sa_global.contiguous(), pairs_all.contiguous(), my_first, out_slice)


def finish_shard_lists(plan, out_slice, gather):
    """The same from lists (plans in list mode, Plan.set_tie_mode(Plan.TIES_LIST)): gather((tied, placed, depth, complete)) -> (every shard's tied records concatenated
    in shard order, all placed records, smallest and largest depth over the shards that reported tied suffixes, all complete).  Called by EVERY shard (collectively) after
    its run; a union that does not fit a shard's workspace (Plan.list_fits) is the caller's to handle: TIES_LOCAL and run again."""
    tied, placed = plan.list_export(out_slice.device)
    _, _, depth, complete = plan.list_counts()
    tied_all, placed_all, dmin, dmax, all_complete = gather((tied, placed, depth, complete))
    if tied_all.shape[0]:
        plan.list_finish(tied_all.contiguous(), placed_all.contiguous(), dmin, dmax, all_complete, out_slice)


def bwt_device(file_bytes, inputtype, blocksize=0, isasamplingrate=0, keep_sa=False):
    """convenience: host packed bytes -> (bwt numpy, result dict, plan); keep_sa: the sort keeps its whole suffix array (b3m_plan_set_keep_sa)"""
    fb = np.ascontiguousarray(file_bytes, dtype=np.uint8)
    if _type_id(inputtype) == INPUT_TYPES["compactstream"]:
        itype, d, n = compact_to_plan_input(fb)
        plan = Plan(n, itype, blocksize=blocksize, isasamplingrate=isasamplingrate)
        try:
            if keep_sa:
                plan.keep_sa(True)
            out = plan.run(d)
            torch.cuda.synchronize()
            return out.cpu().numpy(), plan.result(), plan
        except Exception:
            plan.close()
            raise
    n = num_symbols(fb, inputtype)
    plan = Plan(n, inputtype, blocksize=blocksize, isasamplingrate=isasamplingrate)
    try:
        if keep_sa:
            plan.keep_sa(True)
        d = torch.from_numpy(fb).cuda()
        out = plan.run(d)
        torch.cuda.synchronize()
        return out.cpu().numpy(), plan.result(), plan
    except Exception:
        plan.close()
        raise


def bwt_symbols32(symbols, isasamplingrate=0):
    """texts over alphabets wider than a byte (B3M_INPUT_SYMBOLS32: what inputtype=utf-8 ranks a text of more than 256 distinct code points
    into): symbols = integers below 2^21 -> (bwt as uint32 numpy array, result dict, plan, the BWT's device tensor as uint8 view)"""
    sym = np.ascontiguousarray(symbols, dtype=np.uint32)
    if len(sym) and int(sym.max()) >= 1 << 21:
        raise _lib.B3MError("symbols32: symbols must be below 2^21")
    plan = Plan(len(sym), _lib.SYMBOLS32, isasamplingrate=isasamplingrate)
    try:
        d = torch.from_numpy(sym.view(np.uint8).copy()).cuda()
        out = plan.run(d)
        torch.cuda.synchronize()
        return out.cpu().numpy().view(np.uint32).copy(), plan.result(), plan, out
    except Exception:
        plan.close()
        raise


def lf_walk(d_text, n, inputtype, d_bwt, p0, hist, samples, sarate=0):
    """b3m_lf_walk: checkbwt + sampled SA over a BWT that no single plan produced (gathered shard slices); samples: (k, 2) uint64 (rank,pos) pairs in
    position order.  Returns (symbol mismatches, chain mismatches, SA samples or None)."""
    smp = np.ascontiguousarray(samples, dtype=np.uint64)
    h = np.zeros(256, dtype=np.uint64)
    h[: len(hist)] = np.asarray(hist, dtype=np.uint64)
    mm = np.zeros(2, dtype=np.uint64)
    sa = np.zeros((n + sarate - 1) // sarate, dtype=np.uint64) if sarate else None
    check(lib().b3m_lf_walk(_ptr(d_text), n, _type_id(inputtype), _ptr(d_bwt), int(p0), h.ctypes.data, smp.ctypes.data, smp.shape[0], int(sarate),
                            sa.ctypes.data if sarate else None, None, mm.ctypes.data, _stream()))
    return int(mm[0]), int(mm[1]), sa


def histogram(d_text, n, inputtype, start, length):
    h = np.zeros(256, dtype=np.uint64)
    check(lib().b3m_histogram(_ptr(d_text), n, _type_id(inputtype), start, length, h.ctypes.data, _stream()))
    return h


def block_sort(d_text, n, inputtype, start, length, want_gt=True):
    """row a7: returns (sa u32 tensor, bwt u8 tensor, gt bit words tensor or None, p0)"""
    dev = d_text.device
    sa = torch.empty(length, dtype=torch.int32, device=dev)
    bw = torch.empty(length, dtype=torch.uint8, device=dev)
    gt = torch.zeros((length + 1 + 31) // 32, dtype=torch.int32, device=dev) if want_gt else None
    p0 = ctypes.c_uint64(0)
    check(lib().b3m_block_sort(_ptr(d_text), n, _type_id(inputtype), start, length, _ptr(sa), _ptr(bw), _ptr(gt),
                               ctypes.byref(p0), _stream()))
    return sa, bw, gt, int(p0.value)


def rank_of(d_text, n, inputtype, d_sa, positions):
    pos = np.ascontiguousarray(positions, dtype=np.uint64)
    out = np.zeros(len(pos), dtype=np.uint64)
    check(lib().b3m_rank_of(_ptr(d_text), n, _type_id(inputtype), _ptr(d_sa), d_sa.numel(), pos.ctypes.data, len(pos),
                            out.ctypes.data, _stream()))
    return out


def gap(d_text, n, inputtype, a, m, mr, d_bwtL, p0L, d_gtR, anchors, chain_len, want_dense=True, want_gt=True):
    """rows a8/a9: returns (sorted ranks u32 tensor, dense G tensor or None, gtP words or None)"""
    dev = d_text.device
    anc = np.ascontiguousarray(anchors, dtype=np.uint64)
    sr = torch.empty(mr, dtype=torch.int32, device=dev)
    G = torch.empty(m + 1, dtype=torch.int32, device=dev) if want_dense else None
    gtP = torch.zeros((m + mr + 1 + 31) // 32, dtype=torch.int32, device=dev) if want_gt else None
    check(lib().b3m_gap(_ptr(d_text), n, _type_id(inputtype), a, m, mr, _ptr(d_bwtL), p0L, _ptr(d_gtR), anc.ctypes.data,
                        len(anc), chain_len, _ptr(sr), _ptr(G), _ptr(gtP), _stream()))
    return sr, G, gtP


def interleave(d_sr, d_bwtL, d_bwtR, p0L):
    """row a10: returns (merged bwt tensor, new rank of L's p0 entry)"""
    m, mr = d_bwtL.numel(), d_bwtR.numel()
    out = torch.empty(m + mr, dtype=torch.uint8, device=d_bwtL.device)
    p0 = ctypes.c_uint64(0)
    check(lib().b3m_interleave(_ptr(d_sr), _ptr(d_bwtL), m, _ptr(d_bwtR), mr, _ptr(out), p0L, ctypes.byref(p0), _stream()))
    return out, int(p0.value)
