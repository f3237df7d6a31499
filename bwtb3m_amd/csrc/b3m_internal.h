// b3m_internal.h -- shared host-side declarations of libb3m (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <stdexcept>
#include "../../include/b3m.h"
#include "switches.h"

namespace b3m {

// ---- errors -------------------------------------------------------------------------------
void set_error(const std::string &msg);
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

#define B3M_HIP(expr)                                                                          \
	do {                                                                                       \
		hipError_t e__ = (expr);                                                               \
		if (e__ != hipSuccess)                                                                 \
			throw b3m::Error(std::string(#expr) + ": " + hipGetErrorString(e__) + " (" +       \
			                 __FILE__ + ":" + std::to_string(__LINE__) + ")");                 \
	} while (0)
#define B3M_REQUIRE(cond, msg)                                                                 \
	do { if (!(cond)) throw b3m::Error(std::string(msg)); } while (0)

// ---- text view (device) ---------------------------------------------------------------------
// Packed input exactly as in the file, copied into a buffer padded with >= 16 zero bytes so
// that 3-word look-ahead reads never leave the allocation.
struct TextView {
	const uint8_t *b;   // bytes (symbols32: n uint32 words)
	uint64_t n;         // number of symbols (pacterm: bases + 1)
	uint64_t nbases;    // pac/pacterm: number of 2-bit bases; bytestream: n
	int type;           // B3M_INPUT_*
};

// Symbols per 64-bit sort key and tag width, by input type (DESIGN.md "sort key").
//   bytestream: 8 symbols, no tag.  pac: 32 symbols, no tag (circular, no terminator).
//   pacterm: KSYM_PACTERM (29) symbols in the high bits, 5-bit tag in the low bits:
//            tag = offset of the terminator inside the window, 31 = no terminator.
constexpr int KSYM_BYTES = 8;
constexpr int KSYM_PAC = 32;
// 2 * K + 5 tag bits + 1 <= 64: K <= 29.  29 since round 4 (21 before): a key round of the list-based rounds covers 29 bases, so a repeat-rich text needs fewer
// slot-rounds (genome-like text at C3 size: 227.6 -> 215.8 ms, profiles/NOTES_r04.md); the price is 64-bit keys in the tiles of those rounds (no room for the slot in
// the key's low bits: two compares per pair) and 8 instead of 6 byte-wise splits of a large group.  -DB3M_KSYM_PACTERM=21 builds the old form.
#ifndef B3M_KSYM_PACTERM
#define B3M_KSYM_PACTERM 29
#endif
constexpr int KSYM_PACTERM = B3M_KSYM_PACTERM;
constexpr int KSYM_WIDE = 3, WIDE_BITS = 21;      // symbols32: three 21-bit symbols per key (63 bits)
constexpr int TAGBITS_PACTERM = 5;
constexpr uint32_t TAG_NORMAL = 31;

inline int key_symbols(int type) { return type == B3M_INPUT_BYTESTREAM ? KSYM_BYTES : type == B3M_INPUT_PAC ? KSYM_PAC : type == B3M_INPUT_SYMBOLS32 ? KSYM_WIDE : KSYM_PACTERM; }
inline int key_bits(int type) { return type == B3M_INPUT_PACTERM ? 2 * KSYM_PACTERM + TAGBITS_PACTERM : type == B3M_INPUT_SYMBOLS32 ? KSYM_WIDE * WIDE_BITS : 64; }
inline uint64_t packed_text_bytes(uint64_t n, int type)
{
	if (type == B3M_INPUT_BYTESTREAM) return n;
	if (type == B3M_INPUT_SYMBOLS32) return 4 * n;
	uint64_t l = (type == B3M_INPUT_PACTERM) ? n - 1 : n;
	return l / 4 + 2;                    // BWA .pac file size
}
inline uint64_t num_bases(uint64_t n, int type) { return type == B3M_INPUT_PACTERM ? n - 1 : n; }

// ---- kernel-family timing ---------------------------------------------------------------------
enum KFam { KF_KEYS = 0, KF_RADIX_HIST, KF_SCAN, KF_RADIX_SCATTER, KF_TIES, KF_EMIT, KF_RANKBUILD,
            KF_ANCHOR, KF_GAPWALK, KF_INTERLEAVE, KF_MISC, KF_MSD_HIST, KF_MSD_SCATTER, KF_LDS_SORT, KF_MSD_SCATTER_TEXT, KF_COUNT };
extern const char *const kfam_names[KF_COUNT];

struct Timeline {
	bool enabled = false;
	hipStream_t stream = nullptr;
	struct Ev { hipEvent_t a, b; int fam; uint64_t bytes; };
	std::vector<Ev> evs;
	std::vector<hipEvent_t> pool;
	hipEvent_t get();
	size_t begin(int fam, uint64_t alg_bytes);     // -> index of the entry (scopes may nest or live on different streams)
	void end(size_t idx);
	void collect(double ms[KF_COUNT], uint64_t launches[KF_COUNT], uint64_t bytes[KF_COUNT]);
	void reset();
	~Timeline();
};
struct Scope {
	Timeline *t; size_t idx = 0;
	Scope(Timeline *tl, int fam, uint64_t bytes = 0) : t(tl) { if (t && t->enabled) idx = t->begin(fam, bytes); }
	~Scope() { if (t && t->enabled) t->end(idx); }
};

// ---- device helpers (launchers defined in the .hip files) ------------------------------------------
struct ScanWs { uint32_t *sums = nullptr; uint64_t cap = 0; };   // scratch for device scans
uint64_t scan_ws_elems(uint64_t n);
void exclusive_sum_u32(uint32_t *d, uint64_t n, uint32_t *ws, hipStream_t s);      // in place
void inclusive_max_u32(uint32_t *d, uint64_t n, uint32_t *ws, hipStream_t s);      // in place

struct RadixWs {
	uint32_t *tile_hist = nullptr;   // [256][numTiles]
	uint32_t *scan_ws = nullptr;
	uint32_t *ghist = nullptr;       // [8][256] device
	uint64_t cap_elems = 0;
};
uint64_t radix_tiles(uint64_t n);
uint64_t radix_ws_bytes(uint64_t n);
void radix_ws_carve(RadixWs &ws, void *base, uint64_t n);
// LSD radix sort of (key,value) pairs on bits [0,nbits). Result ends in (*k_res,*v_res) which
// is one of the two buffer pairs.  Stable.  v may be null for keys only.
void radix_sort_u64(uint64_t *kA, uint32_t *vA, uint64_t *kB, uint32_t *vB, uint64_t n, int nbits,
                    RadixWs &ws, hipStream_t s, Timeline *tl, uint64_t **k_res, uint32_t **v_res);
// MSD sort of 32-bit keys (k_msd.hip): false = not applicable (few keys / skewed), kA then holds a permutation of the input
uint64_t sort_u32_msd_scratch(uint64_t n);
bool sort_u32_msd(uint32_t *kA, uint32_t *kB, uint64_t n, int nbits, uint32_t *scratch, uint64_t scratch_elems, uint32_t *scan_ws,
                  hipStream_t s, Timeline *tl, uint32_t **k_res);
// rank_list = true: the keys are ranks < 2^nbits plus 0xFFFFFFFF padding that may come back as 2^nbits - 1 (large lists
// then take the MSD path); false: stable sort on the low nbits that keeps every key's high bits
void radix_sort_u32(uint32_t *kA, uint32_t *kB, uint64_t n, int nbits, RadixWs &ws, hipStream_t s,
                    Timeline *tl, uint32_t **k_res, bool rank_list = false);

// ---- (rank,pos) samples written where a suffix is finally placed ---------------------------------
// The MSD sort of a whole-text block (or of a suffix-range shard of one) emits the sample of position x = k * rate into the fixed
// slot k - first the moment it knows the suffix's final rank: in the finish kernels for records alone in their key, in the tie fix
// and in the list-based rounds' emit for the others.  No suffix array has to be kept for the sake of one entry in `rate`
// (row a12: the reference always writes .preisa, /root/reference/src/checkbwt.cpp:131-148).
// Divisibility of a 32-bit position without a division: low tz bits zero and (x >> tz) * inv <= lim (inv = inverse of the odd
// part of the rate modulo 2^32); positions above 2^32 need a power-of-two rate (sample_in_kernel_ok).
struct SampleOut {
	uint64_t *pairs = nullptr;       // null: no samples
	uint64_t rank_off = 0;           // global rank of the sort's first slot (suffix-range shards)
	uint64_t first = 0, cap = 0;     // slots [first, first + cap) are this block's
	uint32_t rate = 1, tz = 0, pmask = 0xFFFFFFFFu, inv = 1, lim = 0xFFFFFFFFu, pow2 = 1;
};
SampleOut make_sample_out(uint64_t *pairs, uint64_t rate, uint64_t first, uint64_t cap);
bool sample_in_kernel_ok(uint64_t rate, uint64_t n);

// ---- block sort (row a7) ---------------------------------------------------------------------
struct MsdTile { uint32_t in_off, count, hist_base, stride; };
struct MsdWs {
	uint32_t *hist = nullptr; uint64_t hist_cap = 0;        // flattened [bucket][digit][tile] counts / offsets
	MsdTile *tiles = nullptr; uint64_t tiles_cap = 0;       // tile descriptors (device)
	uint32_t *sub_start = nullptr;                           // starts of the buckets leaving level 0
	uint32_t *bucket_tab = nullptr;                          // per bucket entering level 1: start, hist_off, ntiles
	uint32_t *lvl_start = nullptr;                           // starts of the final buckets
	uint32_t *overflow = nullptr; uint64_t overflow_cap = 0; // [0] = count, then (start, size) pairs
	uint32_t *biglist = nullptr; uint64_t big_cap = 0;       // [0] = count, then ids of final buckets above the LDS capacity
	uint32_t *hugelist = nullptr; uint64_t huge_cap = 0;     // [0] = count, then ids of final buckets above F_HUGE records
	uint32_t *midlist = nullptr; uint64_t mid_cap = 0;       // [0] = count, then ids of final buckets for the 1024-thread finish: 1x..2x the LDS capacity, or listed by the gathering finishes
	void *lists = nullptr;                                   // four FinishLists structs (device): what the finish kernels read when they must list a bucket (level 1: two, level 2: two)
	// level 2 (buckets above twice the finish's capacity get one more 8-bit partition, in parallel tiles, through the level-1 kernels):
	uint32_t *mid2 = nullptr, *big2 = nullptr, *huge2 = nullptr; uint64_t list2_cap = 0;      // its own bucket lists
	uint32_t *lvl2 = nullptr; uint64_t lvl2_buckets = 0, lvl2_words = 0;     // starts[256 nb], ends[256 nb], bucket table[3 nb] for up to lvl2_buckets listed buckets
	uint32_t *runpre = nullptr;                              // the same rows as start | records before the run << 14 (for the gathering finish)
	uint32_t *runtab = nullptr; uint64_t runtab_cap = 0;     // level-1 run table: row (level-0 bucket, digit), column = chunk: start | length << 16
	uint32_t *scan_ws = nullptr;
	// chunk-local level 0: run table (row = digit, column = chunk), the rows' prefixes, first runs of level 1's chunks, gather tiles
	uint32_t *tab0 = nullptr, *pre0 = nullptr; uint64_t tab0_cap = 0;
	uint32_t *tile_fr = nullptr; uint64_t tile_fr_cap = 0;
	void *gtiles = nullptr; uint64_t gtiles_cap = 0;
	uint32_t *tfirst = nullptr;
	uint32_t *cbase0 = nullptr; uint64_t cbase0_cap = 0;    // one-pass shards: where the records of level-0 chunk c start (+ two scalars behind the columns)
	// k_bucket_finish recursing into its sub-buckets above the wave sort's capacity: two sets of (start[], end[], list[0] = count, ids) for up to sub_cap of them
	uint32_t *sub_st[2] = {nullptr, nullptr}, *sub_en[2] = {nullptr, nullptr}, *sub_list[2] = {nullptr, nullptr}; uint64_t sub_cap = 0;
};
uint64_t msd_ws_bytes(uint64_t cap, uint64_t span);
void msd_ws_carve(MsdWs &ws, void *base, uint64_t cap, uint64_t span);
// a caller that streams the BWT away while it is being made: the MSD finish runs in `nslices` slices of whole level-0
// buckets and calls fn(ctx, first, last) on the host once the kernels that make bwt[first, last) final are enqueued
struct SliceHook { int nslices; void (*fn)(void *ctx, uint64_t first, uint64_t last); void *ctx; };
// a text that is still arriving (host-to-host runs: the H2D copy goes out in pieces on another stream): after event i the first
// bytes_ready[i] bytes of the packed text are in place.  The chunk-local level 0 needs no global pass over the text before it starts,
// so it runs piece by piece behind the copy; every other path waits for the last event first.
struct TextArrival { int n; hipEvent_t ev[16]; uint64_t bytes_ready[16]; };
// suffixes that the direct comparisons of the record sort (k_tie_fix: records with equal key fields, groups of up to 32) have placed for good although they agree with a
// neighbour on `keys` sort keys or more: prefix doubling on a list alone must be able to look their rank up (tie_rounds_sparse); keys = 0: none are collected
struct TieExtras { uint32_t *cnt = nullptr, *slot = nullptr, *pos_lo = nullptr; uint8_t *pos_hi = nullptr; uint32_t cap = 0, keys = 0; };
struct BlockSortWs {
	uint64_t cap = 0;                 // max block length
	bool no_defer = false;            // suffix-range shards run one after the other by ONE plan (texts below 2^32 symbols in the host tier): nobody gathers the shards' suffix
	                                  // arrays, so a shard finishes its long repeats itself, by key-extension rounds to the end -- unless the plan keeps a list of tied suffixes
	                                  // over all its shards (sparse_room entries still free there): then the shard stops after the usual few rounds and hands over what is left
	                                  // (out.tie_left, out.tie_depth), which the plan finishes by prefix doubling on that list alone (engine.hip: sparse_tie_finish)
	uint64_t sparse_room = 0;
	TieExtras extras;                 // (with sparse_room: where the shard lists what its direct comparisons placed among long repeats)
	uint64_t *keyA = nullptr, *keyB = nullptr;   // LSD: key ping-pong ; MSD: 8-byte records A/B
	uint32_t *idxA = nullptr, *idxB = nullptr;   // LSD: idx ping-pong ; MSD: idxA = suffix array
	uint8_t *sa_hi = nullptr;                    // texts of 2^32 symbols and more: bits 32.. of the suffix array entries (MSD path)
	uint32_t *t[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // tie-list buffers (cap+1 each); t[5] = ISA of the doubling rounds.  Texts of 2^32 symbols and more:
	                                                                           // t[0] = tie bitmap (cap bits), t[1..5] = list buffers of tie_cap+1 entries
	// LAZY tie buffers (one-block record sorts of texts below 2^32 symbols: the plans of the headline): t[1..5] (20 of the ~50 bytes per suffix of a sort's workspace)
	// serve only the list-based rounds of repeat-rich text; they are allocated by blocksort_ensure_ties() the first time such a sort needs them and kept
	bool lazy_ties = false; void *tie_alloc = nullptr;
	uint64_t tie_cap = 0;                        // tied suffixes the list-based rounds can take
	uint8_t *tie_hi = nullptr;                   // texts of 2^32 symbols and more: bits 32.. of the positions on the list (tie_cap+1 bytes)
	uint32_t *scan_ws = nullptr;
	RadixWs radix;
	MsdWs msd;
	uint32_t *d_scalars = nullptr;    // small device scalars
	const SliceHook *slice_hook = nullptr;
	const TextArrival *arrival = nullptr;
	uint64_t *d_symhist_out = nullptr; // optional: the MSD path adds the block's symbol histogram here (it falls out of the level-0 digit counts)
	const SampleOut *samples = nullptr; // optional (whole-text blocks / their shards, MSD path): (rank,pos) samples are emitted by the sort itself (out.samples_done)
};
// cap = suffixes sorted at once; span = length of the text range they are drawn from (> cap for a suffix-range shard; 0 = cap)
uint64_t blocksort_ws_bytes(uint64_t cap, uint64_t span = 0, bool lazy_ties = false);
void blocksort_ws_carve(BlockSortWs &ws, void *base, uint64_t cap, uint64_t span = 0, bool lazy_ties = false);
void blocksort_ensure_ties(BlockSortWs &ws, uint64_t need = 0);       // no-op unless the tie buffers are lazy and not there yet; texts of 2^32 symbols and more: buffers for `need` tied suffixes
void blocksort_release(BlockSortWs &ws);           // frees what blocksort_ensure_ties allocated
struct BlockSortOut {
	uint32_t *d_sa = nullptr;         // points into ws (valid until next sort) : sorted positions
	uint8_t *d_sa_hi = nullptr;       // their bits 32.. (texts of 2^32 symbols and more), else null
	uint64_t tie_rounds = 0, tie_elements = 0;
	uint64_t msd_overflow_buckets = 0;
	bool used_msd = false;
	uint64_t tie_left = 0;            // suffix-range shard: slots still tied after the local rounds (finished globally), their list and labels in ws
	uint32_t *tie_U = nullptr, *tie_gid = nullptr;
	uint64_t tie_depth = 0;           // symbols every group of that list agrees on (in the text itself: the key rounds have seen them)
	uint64_t extras = 0;              // entries the shard's direct comparisons added to ws.extras (may exceed its cap: then the plan's list is incomplete)
	bool tie_refused = false;         // a sequential shard had more tied suffixes left than the plan's list had room for: it finished them itself (key extension to the end)
	bool late_writes = false;         // slices handed to the hook were written again later (listed buckets, overflow list): copy the whole result again
	bool symhist_done = false;        // ws.d_symhist_out was filled
	bool samples_done = false;        // ws.samples was served: every sample of a suffix sorted here sits in its slot
	bool reran_with_sa = false;       // a BWT-only sort met long repeats and started over with the suffix array kept (prefix doubling reads every rank)
	bool want_ext = false;            // a block of a merge tree with long ties: sort it again together with an overhang (block_sort does)
	uint64_t slow_ties = 0;           // slots that prefix doubling inside the block's domain could not finish (repeat longer than the overhang)
	uint64_t overhang = 0;            // symbols behind the block that were sorted with it
	uint64_t kept_len = 0;            // suffixes sorted here (= len unless this is a suffix-range shard)
	uint64_t slice_offset = 0;        // global rank of the first of them
};
// sorts the block and writes its BWT (d_bwt, len bytes) and p0 (device u32: rank of S_start)
// need_sa = false: the caller only wants the BWT (out.d_sa is null; saves 4 B/suffix of writes on the MSD path)
// shard/nshards > 1: suffix-range shard of a whole-text block (MSD path only): only the suffixes whose
// rank falls into the shard's range are sorted; out.kept_len / out.slice_offset describe the slice
void block_sort(const TextView &T, uint64_t start, uint64_t len, BlockSortWs &ws, hipStream_t s,
                Timeline *tl, BlockSortOut &out, uint8_t *d_bwt, uint32_t *d_p0, int shard = 0, int nshards = 1, bool need_sa = true);
// own / allow_ext: see tie_rounds_domain.  own = 0: every suffix of [start, start+len) belongs to the block
void block_sort_lsd(const TextView &T, uint64_t start, uint64_t len, BlockSortWs &ws, hipStream_t s, Timeline *tl, BlockSortOut &out, uint64_t own = 0, bool allow_ext = false);
bool msd_applicable(const TextView &T, uint64_t len);
void block_sort_msd(const TextView &T, uint64_t start, uint64_t len, BlockSortWs &ws, hipStream_t s, Timeline *tl, BlockSortOut &out,
                    uint8_t *d_bwt, uint32_t *d_p0, int shard, int nshards, bool need_sa, uint64_t own = 0, bool allow_ext = false);
// prefix doubling for the ties of a block of a merge tree, inside the domain [start, start+D) (the block's `own` suffixes + overhang)
uint64_t tie_rounds_domain(uint32_t *idx, uint64_t u, uint64_t *kx, uint64_t *ky, uint32_t *B[6], uint32_t *scan_ws, RadixWs &radix,
                           uint64_t start, uint64_t D, uint64_t own, uint32_t *isa, uint64_t h0, hipStream_t s, uint64_t *left, uint64_t *h_blocked);
void isa_from_sa_rel(const uint32_t *d_sa, uint64_t len, uint64_t start, uint32_t *d_isa, hipStream_t s);
bool no_tree_doubling();          // B3M_NO_TREE_DOUBLING=1: key-extension rounds only (measurement switch)
uint64_t block_overhang();        // symbols of overhang a plan leaves room for (B3M_OVERHANG, default 2^20)
// idx_hi / tmp_hi: texts of 2^32 symbols and more (bits 32.. of the suffix array entries; a byte buffer as long as the list): key extension only
uint64_t tie_rounds(const TextView &T, uint32_t *idx, uint64_t u, uint64_t depth0, uint64_t *kx, uint64_t *ky, uint32_t *B[6],
                    uint32_t *scan_ws, RadixWs &radix, uint64_t len_for_labels, hipStream_t s, uint64_t *left,
                    uint64_t max_rounds = 0, uint32_t *isa = nullptr, uint64_t h0 = 0, uint8_t *idx_hi = nullptr, uint8_t *tmp_hi = nullptr,
                    uint64_t sym_off = 0, bool isa_sparse = false, bool labels_are_heads = false);      // labels_are_heads: B[1][s] is the list index of the first slot of s's group (what this function leaves behind; saves the first round a scan).  isa_sparse: entries may be unset (isa_sparse_from_slots); sym_off: symbols every group of the list shares in front of depth0's key (the record sort's reading: key_at_off)
void isa_from_sa(const uint32_t *d_sa, uint64_t len, uint32_t *d_isa, hipStream_t s);
// prefix doubling on a list alone (k_blocksort.hip): the plans that run suffix-range shards one after the other finish the union of what the shards left tied with it
void sparse_rank_placed(const uint64_t *pos, uint64_t cnt, const uint64_t *xpos, const uint64_t *xrank, uint64_t nx, uint64_t *rank, hipStream_t s);
uint64_t tie_rounds_sparse(const TextView &T, uint32_t *idx, uint8_t *idx_hi, uint8_t *tmp_hi, uint64_t slots, uint64_t u, uint64_t *kx, uint64_t *ky, uint32_t *B[6],
                           uint32_t *scan_ws, RadixWs &radix, const uint64_t *pos, uint64_t *rank, const uint64_t *gslot, uint64_t cnt, uint32_t *d_missing,
                           uint64_t d0, uint64_t X, hipStream_t s, uint64_t *left, uint32_t *missing);
// the inverse suffix array with entries for the listed slots only (their suffix -> the slot), every other one unset: tie_rounds(..., isa_sparse = true)
void isa_sparse_from_slots(const uint32_t *d_sa, uint64_t len, const uint32_t *d_slots, uint64_t total, uint32_t *d_isa, hipStream_t s);
void emit_bwt_list(const TextView &T, const uint32_t *U, uint64_t u, const uint32_t *d_sa, uint64_t start, uint8_t *d_bwt, uint32_t *d_p0, hipStream_t s,
                   const SampleOut *smp = nullptr, const uint8_t *d_sa_hi = nullptr);
// products of a sorted block
void emit_bwt(const TextView &T, const uint32_t *d_sa, uint64_t len, uint64_t start, uint8_t *d_bwt,
              uint32_t *d_p0 /*device scalar*/, hipStream_t s);
void emit_gt(const uint32_t *d_sa, uint64_t len, uint64_t start, uint32_t p0, uint32_t *d_gt_abs /*bit x = gt of absolute position x*/,
             hipStream_t s);
// gt bits of the block's own positions (start, start+len) from the text alone (no SA needed)
void emit_gt_from_text(const TextView &T, uint64_t start, uint64_t len, uint32_t *d_gt_abs, hipStream_t s);
// sample of position k*rate -> pairs[k - first] = (rank + rank_off, pos); slots of other shards stay untouched
void emit_samples(const uint32_t *d_sa, const uint8_t *d_sa_hi, uint64_t len, uint64_t rate, uint64_t rank_off, uint64_t *d_pairs, uint64_t first, uint64_t cap,
                  hipStream_t s);
struct RankQueries { int n = 0; const uint64_t *pos[8]; uint32_t *accum[8]; uint64_t count[8]; };      // up to 8 lists of (position -> rank added to accum)
void rank_of_positions_multi(const TextView &T, const uint32_t *d_sa, uint64_t len, const RankQueries &Q, hipStream_t s);
void rank_of_positions(const TextView &T, const uint32_t *d_sa, uint64_t len, const uint64_t *d_pos, uint64_t count,
                       uint32_t *d_rank_accum /* += */, hipStream_t s);
void compact_decode(const uint8_t *d_words, uint64_t n, uint32_t bits, uint8_t *d_out, hipStream_t s);
void histogram_symbols(const TextView &T, uint64_t start, uint64_t len, uint64_t *d_hist256, hipStream_t s);

// ---- merge (rows a8-a10) ------------------------------------------------------------------------
constexpr uint32_t RANK_BLK_SYMS = 192;  // 64-byte line: 4 x u32 counts + 48 bytes of 2-bit codes
uint64_t rank_blocks(uint64_t m);
// p0 (rank of S_a inside L; that BWT entry belongs to position a-1 and is never counted) stays on
// the device: it is remapped by every merge, so kernels read it through a pointer.
struct RankDict { const uint32_t *blocks; uint64_t m; const uint32_t *p0ptr; uint32_t C[4]; };
// builds the 2-bit rank dictionary of BWT(L); scratch: 4 * rank_blocks(m) u32 + scan ws
void rank_build(const uint8_t *d_bwtL, uint64_t m, const uint32_t *d_p0, int type, uint32_t *d_blocks,
                uint32_t *d_scratch, uint32_t *scan_ws, hipStream_t s);
// multi-GPU merges: 2-bit transport form of a block BWT and the dictionary built from such slices
constexpr int DICT_MAX_PARTS = 8;
void pack2(const uint8_t *d_bwt, uint64_t m, int type, uint8_t *d_packed, hipStream_t s);
void rank_build_from_parts(const uint8_t *const *d_parts, const uint64_t *part_syms, int nparts, uint64_t m, uint64_t p0, uint32_t *d_blocks,
                           uint32_t *d_scratch, uint32_t *scan_ws, hipStream_t s);
void parts_bytes(const uint8_t *const *d_parts, const uint64_t *part_syms, int nparts, uint64_t q0, uint64_t q1, int type, uint8_t *d_out, hipStream_t s);
void dict_bytes(const TextView &T, const uint32_t *d_blocks, uint64_t l0, uint64_t l1, uint64_t p0, uint64_t p0_pos, uint8_t *d_out, hipStream_t s);
void rank_cuts(const uint32_t *d_sr, uint64_t cnt, const uint64_t *d_bounds, int nb, uint64_t *d_cuts, hipStream_t s);
struct GapWalkArgs {
	TextView T; RankDict rd;
	uint64_t a, m, mr;
	const uint32_t *gt_cur;    // absolute-position bit array (level input)
	uint32_t *gt_next;         // absolute-position bit array (level output), pre-zeroed over R
	const uint32_t *anchors;   // num_chains ranks
	uint64_t num_chains, chain_len;
	uint64_t chain_first;     // global index of this launch's first chain (multi-GPU: each rank walks a chain range)
	uint32_t *ranks_out;       // gap_walk_slots() entries in any order, unused slots = 0xFFFFFFFF
};
void gap_walk(const GapWalkArgs &A, hipStream_t s);
uint64_t gap_walk_slots(uint64_t num_chains, uint64_t chain_len);   // entries of ranks_out the walk fills (chains padded to whole quads of steps)
// byte alphabets: count rows every 64 symbols of BWT(L) (bytes, 64-byte aligned, readable to the next multiple of 64) and the walk over them
uint64_t byte_rank_rows(uint64_t m);
uint64_t byte_rank_tiles(uint64_t m);
void byte_rank_build(const uint8_t *d_bwtL, uint64_t m, const uint32_t *d_p0, uint32_t *d_occ, uint32_t *d_th, hipStream_t s);
void gap_walk_bytes(const TextView &T, const uint32_t *d_occ, const uint8_t *d_bwtL, const uint32_t *d_p0, const uint32_t *d_C, uint64_t a, uint64_t m, uint64_t mr,
                    const uint32_t *gt_cur, uint32_t *gt_next, const uint32_t *d_anchors, uint64_t num_chains, uint64_t chain_len, uint32_t *ranks_out, hipStream_t s,
                    uint64_t chain_first = 0);      // chain_first: global index of this launch's first chain (multi-GPU: each rank walks a chain range)
// LF walks over the final BWT between (rank,pos) anchors: SA samples + verification (8f rows 1, 3)
// d_blocks: rank dictionary of the whole BWT (rank_build64), p0 = rank of the terminator entry (~0: none), C = symbol starts
// (counts are 64-bit: d_sb = 4 * rank_superblocks(m) u64, the counts in front of every superblock of lines; the line headers count from there)
uint64_t rank_superblocks(uint64_t m);
void rank_build64(const uint8_t *d_bwt, uint64_t m, const uint64_t *d_p0, int type, uint32_t *d_blocks, uint32_t *d_scratch, uint32_t *scan_ws, uint64_t *d_sb, hipStream_t s);
// extras of a walk: ISA samples (isa_out[p / isarate] = rank of the rotation at p, for the multiples of isarate: bwtcomputessa makes X.isa from a sparser X.preisa this way) and
// nocheck (no text to compare the symbols with: T.b is not read; the chain closures are still checked)
struct LfExtra { uint64_t isarate = 0; uint64_t *d_isa_out = nullptr; bool nocheck = false;
                 uint64_t chain_first = 0, chain_count = 0; };      // chain_count > 0: only the chains [chain_first, chain_first + chain_count) (2-bit walk: one device's share of a run over several)
void lf_walk(const TextView &T, const uint32_t *d_blocks, const uint64_t *d_sb, uint64_t p0, const uint64_t C[4], const uint64_t *d_anchors, uint64_t nanchors, uint64_t sarate,
             uint64_t *d_sa_out, unsigned long long *d_mismatch, hipStream_t s, const LfExtra &X = LfExtra());
void lf_walk_bytes(const TextView &T, const uint8_t *d_bwt, uint64_t *d_keys, uint64_t *d_keys2, uint32_t *d_vals, uint32_t *d_vals2, uint32_t *d_lf,
                   RadixWs &rw, const uint64_t *d_anchors, uint64_t nanchors, uint64_t sarate, uint64_t *d_sa_out, unsigned long long *d_mismatch, hipStream_t s,
                   const LfExtra &X = LfExtra());
void interleave(const uint32_t *d_sr, const uint8_t *d_bwtL, uint64_t m, const uint8_t *d_bwtR, uint64_t mr,
                uint8_t *d_out, uint32_t *d_tile_q /* scratch: tiles+1 */, hipStream_t s);
uint64_t interleave_tiles(uint64_t total);
// new ranks after a merge: L entries r -> r + #{q: sr[q] <= r}; R entries q -> q + sr[q]
void remap_ranks(const uint32_t *d_sr, uint64_t mr, uint64_t *d_pairs, uint64_t count, uint64_t posL_begin, uint64_t posR_begin,
                 uint64_t posR_end, hipStream_t s);
void dense_gap_from_sorted(const uint32_t *d_sr, uint64_t mr, uint32_t *d_G, uint64_t m, hipStream_t s);
void copy_bits(const uint32_t *src, uint32_t *dst, uint64_t bit_begin, uint64_t bit_end, hipStream_t s);
void fill_u32(uint32_t *d, uint32_t v, uint64_t n, hipStream_t s);

}  // namespace b3m
