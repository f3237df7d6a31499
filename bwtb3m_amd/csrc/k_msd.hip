// k_msd.hip -- the fast block sort for 2-bit inputs (pac/pacterm): MSD partition of 8-byte
// records down to buckets that fit the LDS of one workgroup, sorted there.
//
// Why: an LSD radix sort of (u64 key, u32 position) pairs moves ~24 B per suffix per pass plus a
// gather for the BWT symbol (~240 B/suffix in all).  Here a suffix is ONE 8-byte record
//        [ 29 key bits | 3 bits predecessor symbol | 32 bits position ]
// and the passes are (C3: 3 Gbase, D0 = 11, D1 = 8, 2^19 final buckets of ~5.7 K records):
//
//   level 0 (k_l0_pass)       : text -> records, by the first D0 <= 12 bits.  The records do not exist yet, so a wide
//                               fan-out is cheap: a chunk of 65,280 POSITIONS is counting-sorted as 16-bit indices inside
//                               LDS (text of the chunk: 16 KB) and the records are generated in digit order.  Whole blocks:
//                               CHUNK-LOCAL (template flag LOCAL): the chunk's records go to the chunk's own index range, a
//                               stream of whole lines, and a run table (row = digit, column = chunk) says where each digit
//                               sits; no counting pass.  Suffix-range shards (they keep 1/N of every chunk) and texts of
//                               2^32 symbols and more: the scattering form (counting pass, [digit][tile] scan, one run per
//                               digit and chunk written to the digit's bucket).
//   level 1 (k_l1_gather /    : the next D1 <= 8 bits, chunk-local too: 16 Ki records of ONE level-0 bucket are sorted by
//            k_l1_local)        their digit in LDS and written contiguously to the bucket's range (its rank range); behind a
//                               chunk-local level 0 they are gathered from the bucket's row of runs (slot -> run map in LDS),
//                               behind the scattering one they are a stream copy.  Its own run table for the finish.
//   finish  (k_finish_fast /  : one workgroup per final bucket: its ~90 runs of ~64 records gathered ONCE into registers,
//            finish_bucket)     counting sort on 13 more key bits in LDS, rank inside the bin, BWT byte (+ SA entry) dropped
//                               at the final place, key ties marked.
//   ties    (k_tie_collect,   : runs of equal 29+D0 known bits ordered from the text.
//            k_tie_fix_list)
//
// HBM traffic per suffix: 0.25 (text) + 8 | 8 (+4 of unaligned runs) + 8 | 8 (+2) + 1 (+4) = ~43 B (PMC: 128 GB per 3 Gbase step).
// Buckets above the LDS capacity (skewed input) take k_bucket_finish: one more 8-bit partition
// through HBM, then one wave per sub-bucket.  Bins / tie groups above 32 records (repeats), sub-
// buckets above 512 and buckets above 2^20 records go to the overflow list and are finished by the
// list-based rounds of k_blocksort.hip (key extension, then prefix doubling), from depth 0.
// pacterm: the terminator and everything behind it read as base 0 for bucketing; the exact tie
// resolution (key_at with terminator tags) restores "terminator < A" where it matters.
#include "b3m_internal.h"
#include "b3m_text.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace b3m {

constexpr int MSD_THREADS = 256;
constexpr int MSD_ITEMS = 16;
constexpr int MSD_CHUNK = MSD_THREADS * MSD_ITEMS;     // 4096 records staged through LDS at a time
constexpr int MSD_SUPER = 16 * MSD_CHUNK;              // 65536 records per workgroup
constexpr int L1_THREADS = 1024;                      // level 1: 16 records per thread, chunks of 16 Ki records sorted in LDS (128 KB)
constexpr int L1_CHUNK = L1_THREADS * MSD_ITEMS;
constexpr int L1_TCH = 4;                              // chunks per workgroup (its columns of the run table leave as 16-byte pieces)
constexpr int MSD_MAXBITS = 8;                         // fan-out of a record pass (9 bits measured in round 1: level 1 +2.3 ms, level 0 at 10 bits -1.6 ms; again in round 3 with the
                                                       // gathering kernels and the XCD-aware finish, 10 + 9 against 11 + 8 bits at C3: level 0 -0.27, level 1 -0.40, finish +0.63, ties +0.64 ms
                                                       // (a record then knows 39 bits of its rotation, not 40): +0.55 ms in all)
constexpr int MSD_MAXBINS = 1 << MSD_MAXBITS;
constexpr uint64_t MSD_MIN_LEN = 1ull << 18;
// level 0
constexpr int W0_THREADS = 1024;
constexpr int W0A_CHUNK = 65280, W0A_TCH = 4, W0A_BINS = 2048;    // <= 11 bits: 65,280 positions at a time (16-bit indices), 4 chunks per tile
constexpr int W0B_CHUNK = 32768, W0B_TCH = 8, W0B_BINS = 4096;    // 12 bits
constexpr int W0_MAXBITS = 12;
constexpr int W0_MAXBINS = 1 << W0_MAXBITS;
// finish
constexpr int F_CAP = 6144;                            // records of a final bucket sorted in LDS
constexpr uint32_t LVL2_AVG = 3072;                    // level 2: records a sub-bucket is to average (the digit width is chosen per bucket)
constexpr uint32_t F_HUGE = 1u << 20;                      // buckets above this go to the list-based rounds directly where one workgroup would have to partition them (k_bucket_finish) ...
constexpr uint32_t F_HUGE2 = 1u << 26;                     // ... and above this where level 2 partitions them in parallel tiles
constexpr uint32_t LVL2_MIN = 32768;                       // listed buckets above this many records take level 2 (256 sub-buckets for the gathering finish); smaller ones one workgroup
                                                           // each (k_bucket_finish: sub-buckets for its wave sort, up to 512 records; on skewed text the heaviest sub-bucket is ~6x the
                                                           // average).  1 Gbase at 80 % A+T: 98304 -> 97.6 ms, 49152 -> 81.3, 24576 -> 79.4; 65 % A+T: 24.0 ms with all three
constexpr int F_AVG = 5760;                            // target average bucket size
constexpr int F_BINBITS = 13;
constexpr int F_BINS = 1 << F_BINBITS;                 // counting-sort bins of the finish (16-bit counters)
constexpr int LS_CAP = 512;                            // sub-bucket capacity of the wave sort
constexpr int LS_BINBITS = 8;
constexpr int LS_BINS = 1 << LS_BINBITS;               // bin-scatter table of the wave sort
constexpr int LS_TIE_CAP = 32;                         // bins / tie groups up to this size are finished in-kernel
constexpr int LS_WAVES = MSD_THREADS / 64;
// record = [key field | 3 bits predecessor symbol | pb bits position], pb = 32 for texts below 2^32 symbols, 36 above
// (then the key field has 25 bits instead of 29: more key ties for k_tie_fix, nothing else changes)
__host__ __device__ inline int rec_key_shift(int pb) { return pb + 3; }
__host__ __device__ inline int rec_key_bits(int pb) { return 61 - pb; }
constexpr int POSBITS_BIG = 36;
// suffix array entries: low 32 bits, and for texts of 2^32 symbols and more a byte array with the bits above
struct SaOut { uint32_t *lo; uint8_t *hi; };
__device__ __forceinline__ void sa_set(const SaOut &A, uint64_t slot, uint64_t pos) { A.lo[slot] = (uint32_t)pos; if (A.hi) A.hi[slot] = (uint8_t)(pos >> 32); }
__device__ __forceinline__ uint64_t sa_get(const SaOut &A, uint64_t slot) { return (uint64_t)A.lo[slot] | (A.hi ? (uint64_t)A.hi[slot] << 32 : 0ull); }

static inline uint32_t nblk(uint64_t n, uint32_t t = 256) { return (uint32_t)((n + t - 1) / t); }
static inline uint64_t al(uint64_t x) { return (x + 255) & ~255ull; }

// final buckets of <= F_AVG suffixes on average: t bits, D0 from the text, the rest (<= 8) from records
struct MsdDigits { int D0, D1; };
static MsdDigits msd_digits(uint64_t len)
{
	int t = 0; while ((len >> t) > (uint64_t)F_AVG) ++t;
	MsdDigits d;
	if (t <= W0_MAXBITS) { d.D0 = std::max(t, 1); d.D1 = 0; }
	else { d.D0 = 11; d.D1 = t - d.D0; if (d.D1 > MSD_MAXBITS) { d.D1 = MSD_MAXBITS; d.D0 = std::min(W0_MAXBITS, t - d.D1); } }
	// test switch: B3M_MSD_D1=k gives level 1 k bits whatever the size, so that small inputs (which the oracle can check) run the
	// kernels of the large geometries (k_l1_local, the gathering finishes, k_finish_fast from 4 bits on)
	if (sw().MSD_D1.set) {
		const int k = (int)sw().MSD_D1.v;
		if (k >= 1 && k <= MSD_MAXBITS && t > k) { d.D1 = k; d.D0 = std::min(W0_MAXBITS, t - k); }
	}
	// test switch: B3M_MSD_D0=k caps level 0 at k bits (with B3M_MSD_D1: fewer partition bits than the size asks for, which is
	// what a 30 Gbase text has; a suffix-range shard then takes the fine level 0 of block_sort_msd on an input the oracle can check)
	if (sw().MSD_D0.set) { const int k = (int)sw().MSD_D0.v; if (k >= 1 && k < d.D0) d.D0 = k; }
	return d;
}

bool msd_applicable(const TextView &T, uint64_t len)
{
	return (T.type == B3M_INPUT_PAC || T.type == B3M_INPUT_PACTERM) && len >= MSD_MIN_LEN;
}

static const uint64_t MSD_L0_MAX = (uint64_t)W0_MAXBINS + 2;                            // buckets leaving level 0
static const uint64_t MSD_TAB_MAX = ((uint64_t)W0_MAXBINS << MSD_MAXBITS) + 2;          // final buckets

// level 0: 4096 * (span/2^18 + 2) counters over the text the block spans (a suffix-range shard scans the WHOLE
// text but keeps cap records); level 1: 256 * (cap/2^16 + 4098)
static uint64_t msd_hist_cap(uint64_t cap, uint64_t span) { return std::max(cap, span) / 64 + cap / 256 + (4ull << 20); }
static uint64_t msd_tiles_cap(uint64_t cap) { return cap / MSD_SUPER + MSD_L0_MAX + 2; }
static uint64_t msd_ovf_cap(uint64_t cap) { return cap / 33 + 4096; }                   // (start, size) pairs: every listed bin holds > 32 records
static uint64_t msd_big_cap(uint64_t cap) { return cap / F_CAP + 16; }
static uint64_t msd_lvl2_buckets(uint64_t cap) { return cap / (2 * F_CAP) + 16; }       // listed buckets level 2 takes at once (each above twice the finish's capacity)
// sub-buckets of one class of level 2: a bucket of sz records makes 2^d2 < 2 sz / LVL2_AVG of them (2 for d2 = 1); every one may be listed again
// (k_finish_fast lists sparse buckets too), each at most once
static uint64_t msd_list2_cap(uint64_t cap) { return 2 * cap / LVL2_AVG + 2 * msd_lvl2_buckets(cap) + 256; }
static uint64_t msd_lvl2_words(uint64_t cap) { return 2 * msd_list2_cap(cap) + 3 * msd_lvl2_buckets(cap) + 16; }   // starts, ends, bucket table
// entries of the chunk-local level 0's run table: 2048 rows x (whole tiles of 4 chunks of 65,280 positions); 0 = no such path (small blocks, 2^32 and more)
static uint64_t msd_local0_cap(uint64_t cap)
{
	if (cap < MSD_MIN_LEN || cap >= (1ull << 32)) return 0;
	const uint64_t tile = (uint64_t)W0A_CHUNK * W0A_TCH;
	return (uint64_t)W0A_BINS * (((cap + 16 + tile - 1) / tile) * W0A_TCH);
}
// columns of the level-0 run table over a text range of `span` positions (a one-pass shard's chunks cover the whole block, not its own cap)
static uint64_t msd_cols0(uint64_t span)
{
	const uint64_t tile = (uint64_t)W0A_CHUNK * W0A_TCH;
	return ((span + 16 + tile - 1) / tile) * W0A_TCH;
}
static uint64_t msd_runtab_cap(uint64_t cap) { return (cap / L1_CHUNK + MSD_L0_MAX) * (uint64_t)MSD_MAXBINS; }   // one column per chunk, 2^D1 rows per level-0 bucket
static uint64_t msd_sub_cap(uint64_t cap) { return cap / LS_CAP + 64; }                 // sub-buckets above the wave sort's capacity (each holds more than LS_CAP records)

uint64_t msd_ws_bytes(uint64_t cap, uint64_t span)
{
	if (cap < MSD_MIN_LEN) return 4096;
	uint64_t hist = msd_hist_cap(cap, span);
	return al(4 * hist) + al(sizeof(MsdTile) * msd_tiles_cap(cap)) + al(8 * MSD_L0_MAX) + al(4 * 3 * MSD_L0_MAX) + al(4 * MSD_TAB_MAX)
	     + al(4 * (2 * msd_ovf_cap(cap) + 4)) + al(4 * (msd_big_cap(cap) + 2)) + al(4 * (MSD_TAB_MAX + 2)) + al(4 * (cap / F_HUGE + 18)) + 2 * al(4 * msd_runtab_cap(cap)) + 256 + al(4 * scan_ws_elems(hist)) + 4096
	     + 2 * al(4 * (msd_list2_cap(cap) + 2)) + al(4 * (cap / F_HUGE + 18)) + al(4 * msd_lvl2_words(cap)) + 6 * al(4 * (msd_sub_cap(cap) + 2))
	     + (msd_local0_cap(cap) ? 2 * al(4 * msd_local0_cap(cap)) + al(4 * (cap / L1_CHUNK + 2 * MSD_L0_MAX)) + al(32 * msd_tiles_cap(cap)) + al(4 * MSD_L0_MAX)
	                              + al(4 * (msd_cols0(std::max(cap, span)) + 32)) : 0);
}
void msd_ws_carve(MsdWs &ws, void *base, uint64_t cap, uint64_t span)
{
	if (cap < MSD_MIN_LEN) { ws = MsdWs(); return; }
	uint8_t *p = (uint8_t *)base;
	uint64_t hist = msd_hist_cap(cap, span);
	ws.hist = (uint32_t *)p; ws.hist_cap = hist; p += al(4 * hist);
	ws.tiles = (MsdTile *)p; ws.tiles_cap = msd_tiles_cap(cap); p += al(sizeof(MsdTile) * ws.tiles_cap);
	ws.sub_start = (uint32_t *)p; p += al(8 * MSD_L0_MAX);          // level-0 bucket starts; first the u64 row totals
	ws.bucket_tab = (uint32_t *)p; p += al(4 * 3 * MSD_L0_MAX);
	ws.lvl_start = (uint32_t *)p; p += al(4 * MSD_TAB_MAX);
	ws.overflow = (uint32_t *)p; ws.overflow_cap = msd_ovf_cap(cap); p += al(4 * (2 * ws.overflow_cap + 4));
	ws.biglist = (uint32_t *)p; ws.big_cap = msd_big_cap(cap); p += al(4 * (ws.big_cap + 2));
	ws.hugelist = (uint32_t *)p; ws.huge_cap = cap / F_HUGE + 16; p += al(4 * (ws.huge_cap + 2));
	ws.midlist = (uint32_t *)p; ws.mid_cap = MSD_TAB_MAX; p += al(4 * (ws.mid_cap + 2));       // any final bucket can be listed here (skew, sparse buckets), each at most once
	ws.runtab = (uint32_t *)p; ws.runtab_cap = msd_runtab_cap(cap); p += al(4 * ws.runtab_cap);
	ws.runpre = (uint32_t *)p; p += al(4 * ws.runtab_cap);
	ws.lists = p; p += 256;
	ws.list2_cap = msd_list2_cap(cap);
	ws.mid2 = (uint32_t *)p; p += al(4 * (ws.list2_cap + 2));
	ws.big2 = (uint32_t *)p; p += al(4 * (ws.list2_cap + 2));
	ws.huge2 = (uint32_t *)p; p += al(4 * (cap / F_HUGE + 18));
	ws.lvl2 = (uint32_t *)p; ws.lvl2_buckets = msd_lvl2_buckets(cap); ws.lvl2_words = msd_lvl2_words(cap); p += al(4 * msd_lvl2_words(cap));
	ws.sub_cap = msd_sub_cap(cap);
	for (int i = 0; i < 2; ++i) {
		ws.sub_st[i] = (uint32_t *)p; p += al(4 * (ws.sub_cap + 2));
		ws.sub_en[i] = (uint32_t *)p; p += al(4 * (ws.sub_cap + 2));
		ws.sub_list[i] = (uint32_t *)p; p += al(4 * (ws.sub_cap + 2));
	}
	ws.scan_ws = (uint32_t *)p; p += al(4 * scan_ws_elems(hist));
	// chunk-local level 0 (whole blocks below 2^32): its run table and the rows' prefixes, level 1's gather tiles
	ws.tab0 = nullptr; ws.pre0 = nullptr; ws.tab0_cap = 0; ws.cbase0 = nullptr; ws.cbase0_cap = 0;
	if (msd_local0_cap(cap)) {
		ws.tab0_cap = msd_local0_cap(cap);
		ws.tab0 = (uint32_t *)p; p += al(4 * ws.tab0_cap);
		ws.pre0 = (uint32_t *)p; p += al(4 * ws.tab0_cap);
		ws.tile_fr_cap = cap / L1_CHUNK + 2 * MSD_L0_MAX; ws.tile_fr = (uint32_t *)p; p += al(4 * ws.tile_fr_cap);
		ws.gtiles_cap = msd_tiles_cap(cap); ws.gtiles = p; p += al(32 * ws.gtiles_cap);
		ws.tfirst = (uint32_t *)p; p += al(4 * MSD_L0_MAX);
		ws.cbase0_cap = msd_cols0(std::max(cap, span)); ws.cbase0 = (uint32_t *)p; p += al(4 * (ws.cbase0_cap + 32));     // + 32: the reservation counter, the overflow flag, the "below" count and five symbol counts behind the columns
	}
}

// XCD-aware workgroup order (8 XCDs, each with its own L2; workgroup i lands on XCD i % 8): workgroup i takes item (i % 8) * (n / 8) + i / 8
// (bijective for any n), so that the workgroups resident on one XCD at a time hold NEIGHBOURING items.  Neighbouring final buckets (and
// neighbouring level-0 buckets over the same chunks) read neighbouring runs: the 128-byte line at a run's unaligned edge belongs to both, and
// is then fetched into that XCD's L2 once instead of by two XCDs.
__device__ __forceinline__ uint32_t xcd_order(uint32_t i, uint32_t n)
{
	const uint32_t q = n >> 3, r = n & 7u, x = i & 7u;
	return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (i >> 3);
}
// 64 bits = the 32 bases from `pos` on, with the bucketing convention for the end of the text:
// pacterm: the terminator and everything behind it read as 0; pac: circular.
__device__ inline uint64_t window_tail(const TextView &T, uint64_t pos)
{
	uint64_t x = 0;
	if (T.type == B3M_INPUT_PAC) {
		uint64_t p = pos % T.n;
		for (int t = 0; t < 32; ++t) { x = (x << 2) | base_at(T, p); if (++p == T.n) p = 0; }
	} else {
		for (int t = 0; t < 32; ++t) { uint64_t p = pos + t; x = (x << 2) | (p < T.nbases ? base_at(T, p) : 0u); }
	}
	return x;
}
__device__ __forceinline__ uint64_t window_safe(const TextView &T, uint64_t pos)
{
	return (pos + 32 <= T.nbases) ? window32(T, pos) : window_tail(T, pos);
}
// output symbol preceding position i (the BWT symbol of the rotation starting at i)
__device__ __forceinline__ uint32_t pred_sym(const TextView &T, uint64_t i)
{
	if (T.type == B3M_INPUT_PACTERM) return i == 0 ? 0u : base_at(T, i - 1) + 1u;     // T[n-1] is the terminator
	return base_at(T, i ? i - 1 : T.n - 1);
}
__device__ __forceinline__ uint64_t make_rec(uint64_t x, int D1, uint32_t pred, uint64_t idx, uint32_t *dig, int pb)
{
	*dig = (uint32_t)(x >> (64 - D1));
	const int kb = rec_key_bits(pb);
	uint64_t kf = (x >> (64 - D1 - kb)) & ((1ull << kb) - 1);
	return (kf << rec_key_shift(pb)) | ((uint64_t)pred << pb) | idx;
}

// ------------------------------------------------------------------------------------------------
// level 0: text -> records
// ------------------------------------------------------------------------------------------------
struct L0Args {
	TextView T; uint64_t start, len;
	uint64_t P0;                 // absolute position of item 0 of tile 0 (multiple of 16)
	uint64_t *rec_out;
	uint32_t *hist;              // COUNT_ONLY: counts out [bin][tile] ; else scanned offsets in
	int D;                       // digit bits (up to 15: a suffix-range shard splits its digit range further)
	uint32_t keep_lo, keep_hi;   // only positions with keep_lo <= digit < keep_hi take part; bin = digit - keep_lo, at most MAXBINS bins
	int pb;                      // position bits of a record
	uint32_t *runtab0;           // LOCAL: row (digit) x column (chunk): start inside the chunk | length << 16
	uint32_t nchunks0;           // LOCAL: columns
	uint32_t tile_base;          // LOCAL: first tile of this launch (the text may arrive in pieces: one launch per piece)
	uint32_t tile_stride;        // COUNT_ONLY: workgroup i counts tile i * tile_stride into column i (0 = 1; a sample of the tiles: shard borders)
	// SHARD (chunk-local level 0 of a suffix-range shard: ONE pass over the text): a chunk keeps the positions of the shard's digits only and
	// reserves room for them at the end of what the chunks before it (in time) took; where, says cbase[column]
	uint32_t *cbase; uint32_t *alloc; uint32_t cap;
	unsigned long long *below;   // += positions of the block with a digit below keep_lo: the global rank of the shard's first suffix
	unsigned long long *symhist; // SHARD, may be null: += the block's symbol histogram (256 entries; every position is looked at anyway)
	// EDGE (a suffix-range shard whose borders fall INSIDE a level-0 digit: skewed text): of the digit keep_lo only the positions whose 64-bit window (32 bases,
	// window_safe) is >= key_lo take part (edge & 1), of the digit keep_hi - 1 only those below key_hi (edge & 2); `below` then also counts the positions of digit
	// keep_lo in front of key_lo (COUNT_ONLY passes count it too: the two-pass form gets the shard's first rank from there)
	uint64_t key_lo, key_hi; uint32_t edge;
};

// EDGE: does the position with digit dg0 and window x (fetched on demand: only the two edge digits need it) belong to the shard?  *blw: it lies in front of the shard
template <typename F>
__device__ __forceinline__ bool l0_edge_keep(const L0Args &A, const uint32_t dg0, const uint32_t klo, const uint32_t nbin, F window, bool *blw)
{
	const uint32_t bin = dg0 - klo;
	bool in = bin < nbin;
	*blw = dg0 < klo;
	const bool at_lo = (A.edge & 1u) && dg0 == klo, at_hi = (A.edge & 2u) && in && bin == nbin - 1u;
	if (at_lo || at_hi) {
		const uint64_t x = window();
		if (at_lo && x < A.key_lo) { in = false; *blw = true; }
		if (at_hi && x >= A.key_hi) in = false;
	}
	return in;
}

// window / predecessor of an arbitrary position (chunks at the block edges or near the text end)
__device__ __forceinline__ bool l0_slow(const TextView &T, uint64_t i, uint64_t start, uint64_t end, uint64_t *x, uint32_t *pr)
{
	if (i < start || i >= end) return false;
	*x = window_safe(T, i); *pr = pred_sym(T, i);
	return true;
}

// CHUNK positions are counting-sorted at a time, TILE = TCH chunks share one row of the histogram matrix.
// Two geometries: 65,280-position chunks for up to 2048 bins (32-record runs on average at 11 bits; the
// index array alone is 128 KB of LDS, everything else is squeezed into the rest) and 32,768-position
// chunks for 4096 bins.
// LOCAL (whole-text blocks, no shard): the chunk's records go out in digit order to the chunk's OWN index range of the record buffer,
// a stream of whole lines instead of 2048 scattered 32-record runs, and where digit d sits inside chunk c is one entry of a run table
// (row d, column c); no counting pass and no [digit][tile] offsets are needed.  Level 1 gathers a bucket from its row's runs.
// SHARD = 1: the one-pass level 0 of a suffix-range shard (L0Args).  SHARD = 2: the same with 512 threads and an index array of W0S_SIDX entries
// (three or more shards: a chunk keeps a third of its positions at most; one that keeps more raises a flag and the host takes SHARD = 1), so that TWO
// workgroups share a CU: what paces this pass is the chain of latencies per chunk (text load, barriers, the reservation, the stores), not the
// atomics of the few positions kept.  (Two workgroups of 1024 threads need 64 registers per thread: 161 spilled, the pass went from 3.0 to 7.5 ms.)
constexpr int W0S_SIDX = 24576, W0S_THREADS = 512;
template <bool COUNT_ONLY, int CHUNK, int TCH, int MAXBINS, bool LOCAL = false, int SHARD = 0, bool EDGE = false>
__global__ __launch_bounds__(SHARD == 2 ? W0S_THREADS : W0_THREADS, SHARD == 2 ? 4 : 1) void k_l0_pass(L0Args A)
{
	static_assert(!SHARD || (LOCAL && !COUNT_ONLY), "a one-pass shard is a chunk-local level 0");
	static_assert(!EDGE || COUNT_ONLY || !LOCAL || SHARD, "borders inside a digit belong to suffix-range shards");
	constexpr int TH = SHARD == 2 ? W0S_THREADS : W0_THREADS;
	constexpr int WORDS = CHUNK / 16;
	static_assert(CHUNK % 16 == 0 && CHUNK <= 65536, "chunk-local positions are 16-bit");
	__shared__ uint32_t cur[MAXBINS];                            // count, then exclusive start, then end of every digit run
	__shared__ uint32_t gbase[(COUNT_ONLY || LOCAL) ? 1 : MAXBINS];
	__shared__ uint16_t sidx[COUNT_ONLY ? 1 : SHARD == 2 ? W0S_SIDX : CHUNK];     // chunk-local positions in digit order
	__shared__ uint32_t stext[WORDS + 4];                        // [0] = the word before the chunk
	__shared__ uint32_t wsum[TH / 64];
	__shared__ uint32_t s_base;
	uint32_t n_below = 0;                                        // SHARD: this thread's positions with a digit below the shard's
	uint32_t n_s1 = 0, n_s2 = 0, n_s3 = 0, n_all = 0, n_term = 0; // SHARD: this thread's positions by leading base 1, 2, 3, all of them, the terminator
	const uint32_t tid = threadIdx.x;
	const int bins = (int)(A.keep_hi - A.keep_lo);           // bin = digit - keep_lo; digits outside [keep_lo, keep_hi) are not this sort's (suffix-range shard)
	const uint32_t klo = A.keep_lo, nbin = (uint32_t)bins;
	const int dsh = 64 - A.D, dsh32 = 32 - A.D;
	const uint32_t tcol = blockIdx.x, ntiles = gridDim.x;      // column of the [bin][tile] matrix this workgroup owns
	const uint32_t tile = (LOCAL ? A.tile_base : 0u) + ((COUNT_ONLY && A.tile_stride > 1u) ? tcol * A.tile_stride : tcol);
	const uint64_t end = A.start + A.len;
	const uint32_t padd = A.T.type == B3M_INPUT_PACTERM ? 1u : 0u;
	if (COUNT_ONLY) { for (int b = tid; b < bins; b += TH) cur[b] = 0; }
	else if (!LOCAL) { for (int b = tid; b < bins; b += TH) gbase[b] = A.hist[tile + (uint64_t)b * ntiles]; }
	__syncthreads();
	// LOCAL: the text words of the NEXT chunk are requested while this chunk's records are generated and stored (a few registers per thread; the
	// load's latency used to open every chunk's chain with nothing else in flight: one workgroup per CU)
	constexpr bool PF = LOCAL && SHARD != 2;                      // (the 512-thread form would hold eight words per thread: spills)
	constexpr int PFN = PF ? (WORDS + 4 + TH - 1) / TH : 1;
	uint32_t pf[PFN]; int pf_c = -1;                              // pf_c: the chunk whose words pf[] holds (-1: none)
	for (int c = 0; c < TCH; ++c) {
		const uint64_t cpos = A.P0 + (uint64_t)tile * (TCH * CHUNK) + (uint64_t)c * CHUNK;
		if (LOCAL && (cpos >= end || cpos + CHUNK <= A.start)) {
			// a chunk outside the block: its column of the run table is empty (written here: the table is not cleared beforehand)
			const uint32_t col = tile * TCH + c;
			for (int b = tid; b < bins; b += TH) A.runtab0[(uint64_t)b * A.nchunks0 + col] = 0u;
			if (SHARD && tid == 0) A.cbase[col] = 0u;
			continue;
		}
		if (cpos >= end) break;
		if (cpos + CHUNK <= A.start) continue;
		// the whole chunk inside the block, every window inside the text, a word in front of it
		const bool fast = cpos >= A.start && cpos >= 16 && cpos + CHUNK <= end && cpos + CHUNK + 48 <= A.T.nbases;
		if (fast) {
			if (PF && pf_c == c) {
				#pragma unroll
				for (int j = 0; j < PFN; ++j) { const int w = j * TH + (int)tid; if (w < WORDS + 4) stext[w] = __builtin_bswap32(pf[j]); }
			} else {
				const uint8_t *p = A.T.b + (((cpos >> 4) - 1) << 2);
				for (int j = tid; j < WORDS + 4; j += TH) stext[j] = be32(p + 4 * j);
			}
		}
		if (!COUNT_ONLY) { for (int b = tid; b < bins; b += TH) cur[b] = 0; }
		__syncthreads();
		// ---- count ----
		// a suffix-range shard keeps 1/N of the positions: the count remembers WHICH (16 bits per word of 16 positions, in registers: the
		// same thread takes the same words in the place phase) so that the place phase only visits those
		constexpr int NJ = (WORDS + TH - 1) / TH;
		constexpr bool KM = !COUNT_ONLY && (!LOCAL || SHARD);
		uint32_t km[KM ? (NJ + 1) / 2 : 1];
		const bool sparse = KM && (EDGE || 2u * nbin <= (1u << A.D));      // (EDGE: the count's verdict on every position is kept, the place phase does not test again)
		if (KM) { for (int j = 0; j < (NJ + 1) / 2; ++j) km[j] = 0u; }
		if (fast) {
			#pragma unroll
			for (int j = 0; j < NJ; ++j) {
				const uint32_t w = j * TH + tid;
				if (WORDS % TH != 0 && w >= (uint32_t)WORDS) break;
				// the digit needs <= 12 bits of the window: 32-bit funnel shifts of two words do
				const uint32_t w0 = stext[w + 1], w1 = stext[w + 2];
				uint32_t mk = 0;
				if (SHARD) {          // the 16 positions of this word start with its 16 bases
					const uint32_t lo = w0 & 0x55555555u, hi = (w0 >> 1) & 0x55555555u;
					n_s1 += __popc(lo & ~hi); n_s2 += __popc(hi & ~lo); n_s3 += __popc(lo & hi); n_all += 16u;
				}
				#pragma unroll
				for (int s = 0; s < 16; ++s) {
					const uint32_t xh = s ? __funnelshift_l(w1, w0, 2 * s) : w0;
					const uint32_t dg0 = xh >> dsh32;
					const uint32_t bin = dg0 - klo;
					if (EDGE) {
						bool blw;
						const bool in = l0_edge_keep(A, dg0, klo, nbin, [&]() { const uint32_t w2 = stext[w + 3]; return ((uint64_t)xh << 32) | (s ? __funnelshift_l(w2, w1, 2 * s) : w1); }, &blw);
						n_below += blw ? 1u : 0u;
						if (in) { atomicAdd(&cur[bin], 1u); mk |= 1u << s; }
						continue;
					}
					if (SHARD) n_below += dg0 < klo ? 1u : 0u;
					if (bin < nbin) { atomicAdd(&cur[bin], 1u); mk |= 1u << s; }
				}
				if (KM) km[j >> 1] |= mk << (16 * (j & 1));
			}
		} else {
			#pragma nounroll
			for (uint32_t li = tid; li < (uint32_t)CHUNK; li += TH) {
				uint64_t x; uint32_t pr;
				if (l0_slow(A.T, cpos + li, A.start, end, &x, &pr)) {
					const uint32_t bin = (uint32_t)(x >> dsh) - klo;
					bool blw = (uint32_t)(x >> dsh) < klo, in = bin < nbin;
					if (EDGE) { in = l0_edge_keep(A, (uint32_t)(x >> dsh), klo, nbin, [&]() { return x; }, &blw); if (!SHARD) n_below += blw ? 1u : 0u; }
					if (SHARD) {
						n_below += blw ? 1u : 0u;
						const uint64_t ap = cpos + li;
						if (A.T.type == B3M_INPUT_PACTERM && ap == A.T.n - 1) n_term += 1u;
						else { const uint32_t bs = base_at(A.T, ap); n_s1 += bs == 1u; n_s2 += bs == 2u; n_s3 += bs == 3u; n_all += 1u; }
					}
					if (in) atomicAdd(&cur[bin], 1u);
				}
			}
		}
		if (COUNT_ONLY) { __syncthreads(); continue; }
		__syncthreads();
		// ---- exclusive scan of the counts, in place ----
		{
			constexpr int per = MAXBINS / TH;           // consecutive bins per thread
			uint32_t loc[per]; uint32_t sum = 0;
			#pragma unroll
			for (int k = 0; k < per; ++k) { int b = tid * per + k; loc[k] = b < bins ? cur[b] : 0; sum += loc[k]; }
			uint32_t inc = wave_incl_sum(sum);
			uint32_t w = tid >> 6, l = tid & 63;
			if (l == 63) wsum[w] = inc;
			__syncthreads();
			uint32_t pre = 0;
			for (uint32_t i = 0; i < w; ++i) pre += wsum[i];
			uint32_t run = pre + inc - sum;
			if (SHARD == 2) {
				// more positions kept than the index array holds (the shard's digits are much denser here than on average): nothing of this
				// chunk is placed, the flag tells the host to take the form with the whole index array
				uint32_t all = 0;
				for (uint32_t i = 0; i < (uint32_t)(TH / 64); ++i) all += wsum[i];
				if (all > (uint32_t)W0S_SIDX) {
					const uint32_t col = tile * TCH + c;
					if (tid == 0) { atomicOr(A.alloc + 1, 1u); A.cbase[col] = 0u; }
					for (int b = tid; b < bins; b += TH) A.runtab0[(uint64_t)b * A.nchunks0 + col] = 0u;
					__syncthreads();
					continue;
				}
			}
			#pragma unroll
			for (int k = 0; k < per; ++k) { int b = tid * per + k; if (b < bins) cur[b] = run; run += loc[k]; }
		}
		__syncthreads();
		// ---- place the chunk-local positions in digit order (order inside a digit is irrelevant) ----
		if (fast && sparse) {
			#pragma unroll
			for (int j = 0; j < NJ; ++j) {
				const uint32_t w = j * TH + tid;
				if (WORDS % TH != 0 && w >= (uint32_t)WORDS) break;
				uint32_t mk = (km[KM ? (j >> 1) : 0] >> (16 * (j & 1))) & 0xFFFFu;
				if (mk == 0) continue;
				const uint32_t w0 = stext[w + 1], w1 = stext[w + 2];
				while (mk) {
					const uint32_t sft = (uint32_t)__ffs(mk) - 1u; mk &= mk - 1u;
					const uint32_t bin = (__funnelshift_l(w1, w0, 2u * sft) >> dsh32) - klo;       // (a shift of 0 returns w0)
					sidx[atomicAdd(&cur[bin], 1u)] = (uint16_t)(w * 16 + sft);
				}
			}
		} else if (fast) {
			#pragma unroll
			for (int j = 0; j < NJ; ++j) {
				const uint32_t w = j * TH + tid;
				if (WORDS % TH != 0 && w >= (uint32_t)WORDS) break;
				const uint32_t w0 = stext[w + 1], w1 = stext[w + 2];
				#pragma unroll
				for (int s = 0; s < 16; ++s) {
					const uint32_t bin = ((s ? __funnelshift_l(w1, w0, 2 * s) : w0) >> dsh32) - klo;
					if (bin < nbin) sidx[atomicAdd(&cur[bin], 1u)] = (uint16_t)(w * 16 + s);
				}
			}
		} else {
			#pragma nounroll
			for (uint32_t li = tid; li < (uint32_t)CHUNK; li += TH) {
				uint64_t x; uint32_t pr;
				if (l0_slow(A.T, cpos + li, A.start, end, &x, &pr)) {
					const uint32_t bin = (uint32_t)(x >> dsh) - klo;
					bool in = bin < nbin, blw;
					if (EDGE) in = l0_edge_keep(A, (uint32_t)(x >> dsh), klo, nbin, [&]() { return x; }, &blw);
					if (in) sidx[atomicAdd(&cur[bin], 1u)] = (uint16_t)li;
				}
			}
		}
		__syncthreads();
		// ---- generate the records in digit order: cur[d] is now the END of run d ----
		const uint32_t nvalid = cur[bins - 1];
		if (SHARD) {
			// the chunk's records go behind what earlier chunks reserved (any order: level 1 finds them through the run table and cbase)
			if (tid == 0) { const uint32_t bs = nvalid ? atomicAdd(A.alloc, nvalid) : 0u; s_base = bs; A.cbase[tile * TCH + c] = bs; }
			__syncthreads();
		}
		const uint32_t obase = SHARD ? s_base : 0u;
		if (PF && c + 1 < TCH) {
			const uint64_t npos = cpos + CHUNK;
			if (npos >= A.start && npos >= 16 && npos + CHUNK <= end && npos + CHUNK + 48 <= A.T.nbases) {       // (the next chunk will be `fast`)
				const uint32_t *p = reinterpret_cast<const uint32_t *>(A.T.b + (((npos >> 4) - 1) << 2));
				#pragma unroll
				for (int j = 0; j < PFN; ++j) { const int w = j * TH + (int)tid; pf[j] = w < WORDS + 4 ? p[w] : 0u; }
				pf_c = c + 1;
			}
		}
		for (uint32_t q = tid; q < nvalid; q += TH) {
			const uint32_t li = sidx[q];
			uint32_t dgt; uint64_t r;
			if (fast) {
				// 32-bit funnel shifts only: xh:xl = the 64-bit window at base li, key field = its 29 bits after the digit
				const uint32_t w = li >> 4, sh = 2u * (li & 15u);
				// (the 40 bits a record takes from its window lie in w0 and w1 unless it starts in the last three bases of its word; the predecessor base lies
				// in the word in front only for the first base of a word: the two rare words are read by the lanes that need them -- these reads are random
				// and bank conflicts are what the generate phase waits for)
				const uint32_t w0 = stext[w + 1], w1 = stext[w + 2];
				uint32_t w2 = 0u, wp = 0u;
				if (sh + (uint32_t)A.D + (uint32_t)rec_key_bits(A.pb) > 64u) w2 = stext[w + 3];
				if (sh == 0u) wp = stext[w];
				asm volatile("" : "+v"(w2), "+v"(wp));                   // (keeps the two reads under their conditions: they are safe to speculate and would be hoisted)
				const uint32_t xh = __funnelshift_l(w1, w0, sh), xl = __funnelshift_l(w2, w1, sh);
				const uint32_t pr = (__funnelshift_l(w0, wp, sh) & 3u) + padd;
				dgt = xh >> dsh32;
				const uint32_t kf = __funnelshift_l(xl, xh, (uint32_t)A.D) >> (32 - rec_key_bits(A.pb));
				r = ((uint64_t)((kf << 3) | pr) << A.pb) | (cpos + li);
			} else {
				uint64_t x; uint32_t pr;
				l0_slow(A.T, cpos + li, A.start, end, &x, &pr);
				r = make_rec(x, A.D, pr, cpos + li, &dgt, A.pb);
			}
			if (SHARD) { if ((uint64_t)obase + q < (uint64_t)A.cap) __builtin_nontemporal_store(r, &A.rec_out[(uint64_t)obase + q]); continue; }      // (past the workspace: the host sees alloc > cap and takes the two-pass path)
			if (LOCAL) { __builtin_nontemporal_store(r, &A.rec_out[(cpos - A.P0) + q]); continue; }
			const uint32_t bin = dgt - klo;
			const uint32_t rs = bin ? cur[bin - 1] : 0u;          // start of run bin
			A.rec_out[(uint64_t)gbase[bin] + (q - rs)] = r;
		}
		if (LOCAL) {
			const uint32_t col = tile * TCH + c;
			for (int b = tid; b < bins; b += TH) { const uint32_t st0 = b ? cur[b - 1] : 0u; A.runtab0[(uint64_t)b * A.nchunks0 + col] = st0 | ((cur[b] - st0) << 16); }
			__syncthreads();
			continue;
		}
		__syncthreads();
		for (int b = tid; b < bins; b += TH) gbase[b] += cur[b] - (b ? cur[b - 1] : 0u);
		__syncthreads();
	}
	if (COUNT_ONLY) {
		__syncthreads();
		for (int b = tid; b < bins; b += TH) A.hist[tcol + (uint64_t)b * ntiles] = cur[b];
	}
	if (SHARD || (COUNT_ONLY && EDGE)) {
		#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			n_below += __shfl_xor(n_below, o, 64);
			n_s1 += __shfl_xor(n_s1, o, 64); n_s2 += __shfl_xor(n_s2, o, 64); n_s3 += __shfl_xor(n_s3, o, 64); n_all += __shfl_xor(n_all, o, 64); n_term += __shfl_xor(n_term, o, 64);
		}
		// one atomic per workgroup and counter (one per wave was 16 x 11,500 workgroups on six addresses: the pass took 7 ms instead of 3)
		__syncthreads();
		uint32_t *red = cur;                                     // (dead: the last chunk's run table has been written)
		if ((tid & 63u) == 0u) { uint32_t *r = red + 8u * (tid >> 6); r[0] = n_below; r[1] = n_s1; r[2] = n_s2; r[3] = n_s3; r[4] = n_all; r[5] = n_term; }
		__syncthreads();
		if (tid < 6u) {
			unsigned long long v = 0;
			for (uint32_t w = 0; w < (uint32_t)(TH / 64); ++w) v += red[8u * w + tid];
			red[256 + tid * 2] = (uint32_t)v; red[257 + tid * 2] = (uint32_t)(v >> 32);
		}
		__syncthreads();
		if (tid == 0) {
			auto tot = [&](int i) { return (unsigned long long)red[256 + 2 * i] | ((unsigned long long)red[257 + 2 * i] << 32); };
			if (tot(0)) atomicAdd(A.below, tot(0));
			if (A.symhist && tot(4) + tot(5)) {
				const uint32_t off = A.T.type == B3M_INPUT_PACTERM ? 1u : 0u;
				atomicAdd(&A.symhist[off], tot(4) - tot(1) - tot(2) - tot(3));
				if (tot(1)) atomicAdd(&A.symhist[off + 1], tot(1));
				if (tot(2)) atomicAdd(&A.symhist[off + 2], tot(2));
				if (tot(3)) atomicAdd(&A.symhist[off + 3], tot(3));
				if (tot(5)) atomicAdd(&A.symhist[0], tot(5));
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------
// level 1: records -> records, CHUNK-LOCAL.  A chunk of THREADS*16 consecutive records of ONE level-0
// bucket is sorted by its next D1 bits inside LDS and goes back to the SAME index range of the other
// record buffer, whole lines only (a stream copy: what bounded the old level 1 was the partial lines of
// its scattered 64-record runs, and it needed a counting pass over all records first to know where the
// runs go).  Where digit d of chunk c sits is one entry of the run table, row (bucket, d), column c:
//        start inside the chunk | length << 16
// The final bucket (bucket, d) is the union of row (bucket, d)'s runs; the finish kernel gathers it.
// No histogram pass, no scan over [digit][tile] matrices, no global offsets at this level.
// ------------------------------------------------------------------------------------------------
struct L1Args {
	const uint64_t *rec_in;
	uint64_t *rec_out;
	const MsdTile *tiles;        // in_off, count, hist_base = table index of (digit 0, first chunk of the tile), stride = chunks of the bucket
	uint32_t *runtab;
	int D;                       // digit bits of this pass
	int shift;                   // digit = (rec >> shift) & (2^D - 1)
};

template <int THREADS, int TCH>
__global__ __launch_bounds__(THREADS) void k_l1_local(L1Args A)
{
	constexpr int CHUNK = THREADS * MSD_ITEMS;               // records sorted in LDS at a time
	__shared__ uint32_t cnt[MSD_MAXBINS];
	__shared__ uint32_t binstart[MSD_MAXBINS];
	__shared__ uint32_t tabst[TCH][MSD_MAXBINS];
	__shared__ __attribute__((aligned(16))) uint64_t srec[CHUNK];
	__shared__ uint32_t wsum[THREADS / 64];
	static_assert(THREADS >= MSD_MAXBINS, "one bin per thread");
	const uint32_t tid = threadIdx.x;
	const int bins = 1 << A.D;
	const MsdTile tile = A.tiles[blockIdx.x];
	const uint32_t nchunks = (tile.count + CHUNK - 1) / CHUNK;       // <= TCH
	uint64_t rec[MSD_ITEMS];
	// all 16 loads of a chunk go out back to back (clamped index, no branch around a load); the loads of chunk c+1 are
	// issued before chunk c is copied out, so that the read stream and the write stream overlap inside the one workgroup a CU holds
	{
		const uint32_t nv0 = tile.count < (uint32_t)CHUNK ? tile.count : (uint32_t)CHUNK;
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) { uint32_t e = s * THREADS + tid; rec[s] = A.rec_in[(uint64_t)tile.in_off + (e < nv0 ? e : nv0 - 1)]; }
	}
	for (uint32_t c = 0; c < nchunks; ++c) {
		const uint32_t cbase = c * CHUNK;
		const uint32_t nvalid = tile.count - cbase < (uint32_t)CHUNK ? tile.count - cbase : (uint32_t)CHUNK;
		for (int b = tid; b < bins; b += THREADS) cnt[b] = 0;
		__syncthreads();
		uint32_t dg[MSD_ITEMS]; uint32_t slot[MSD_ITEMS];
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) dg[s] = (uint32_t)(rec[s] >> A.shift) & (uint32_t)(bins - 1);
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) if ((uint32_t)(s * THREADS) + tid < nvalid) slot[s] = atomicAdd(&cnt[dg[s]], 1u);
		__syncthreads();
		{
			const uint32_t v = (int)tid < bins ? cnt[tid] : 0u;
			uint32_t inc = wave_incl_sum(v);
			uint32_t w = tid >> 6, l = tid & 63;
			if (l == 63) wsum[w] = inc;
			__syncthreads();
			uint32_t pre = 0;
			for (uint32_t i = 0; i < w; ++i) pre += wsum[i];
			if ((int)tid < bins) { binstart[tid] = pre + inc - v; tabst[c][tid] = (pre + inc - v) | (v << 16); }
		}
		__syncthreads();
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) if ((uint32_t)(s * THREADS) + tid < nvalid) srec[binstart[dg[s]] + slot[s]] = rec[s];
		if (c + 1 < nchunks) {
			const uint32_t nb1 = cbase + CHUNK, nv1 = tile.count - nb1 < (uint32_t)CHUNK ? tile.count - nb1 : (uint32_t)CHUNK;
			#pragma unroll
			for (int s = 0; s < MSD_ITEMS; ++s) { uint32_t e = s * THREADS + tid; rec[s] = A.rec_in[(uint64_t)tile.in_off + nb1 + (e < nv1 ? e : nv1 - 1)]; }
		}
		__syncthreads();
		uint64_t *out = A.rec_out + (uint64_t)tile.in_off + cbase;
		for (uint32_t p = tid; p < nvalid; p += THREADS) __builtin_nontemporal_store(srec[p], &out[p]);
	}
	__syncthreads();
	// the tile's columns of the run table: digit d -> nchunks consecutive entries of row d
	for (int d = tid; d < bins; d += THREADS) {
		uint32_t *row = A.runtab + (uint64_t)tile.hist_base + (uint64_t)d * tile.stride;
		for (uint32_t c = 0; c < nchunks; ++c) row[c] = tabst[c][d];
	}
}

// ------------------------------------------------------------------------------------------------
// Chunk-local level 0: its run table (row = digit, column = chunk of W0A_CHUNK positions) -> bucket sizes, then the prefix of
// every run inside its row and, for every 16 Ki-record chunk of level 1, the run its first slot lies in.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows0_totals(const uint32_t *tab0, uint32_t nchunks0, unsigned long long *tot)
{
	__shared__ unsigned long long ws[4];
	const uint32_t *row = tab0 + (uint64_t)blockIdx.x * nchunks0;
	unsigned long long c = 0;
	for (uint32_t i = threadIdx.x; i < nchunks0; i += 256) c += row[i] >> 16;
	#pragma unroll
	for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
	if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) tot[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// pre0[d][c] = records of row d in the chunks before c; tile_fr[tfirst[d] + k] = the run that holds slot k * L1_CHUNK of bucket d
// (k < chunks of the bucket), tile_fr[tfirst[d] + chunks] = nchunks0
__global__ __launch_bounds__(256) void k_rows0_prefix(const uint32_t *tab0, uint32_t nchunks0, uint32_t *pre0, const uint32_t *tfirst, uint32_t *tile_fr)
{
	__shared__ uint32_t ws[4];
	const uint32_t d = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t *row = tab0 + (uint64_t)d * nchunks0;
	uint32_t *prow = pre0 + (uint64_t)d * nchunks0;
	uint32_t *fr = tile_fr + tfirst[d];
	uint32_t run = 0;
	for (uint32_t i0 = 0; i0 < nchunks0; i0 += 256) {
		const uint32_t i = i0 + tid;
		const uint32_t len = i < nchunks0 ? row[i] >> 16 : 0u;
		const uint32_t inc = wave_incl_sum(len);
		__syncthreads();
		if (lane == 63) ws[wave] = inc;
		__syncthreads();
		uint32_t pre = run + inc - len;
		for (uint32_t w = 0; w < wave; ++w) pre += ws[w];
		if (i < nchunks0) {
			prow[i] = pre;
			for (uint32_t k = (pre + (uint32_t)L1_CHUNK - 1) / (uint32_t)L1_CHUNK; (uint64_t)k * L1_CHUNK < (uint64_t)pre + len; ++k) fr[k] = i;
		}
		run += ws[0] + ws[1] + ws[2] + ws[3];
	}
	if (tid == 0) fr[(run + (uint32_t)L1_CHUNK - 1) / (uint32_t)L1_CHUNK] = nchunks0;
}

// ------------------------------------------------------------------------------------------------
// Level 1 behind a chunk-local level 0: the same chunk-local sort as k_l1_local, but the 16 Ki records of a chunk are GATHERED from
// the runs of the bucket's row (~512 runs of ~32 records on uniform input) through the slot -> run map of the gathering finish
// (group descriptors of up to four runs per 64 slots; a lane of a group with more searches the run list), and written contiguously
// to the bucket's own range of the output buffer.
// ------------------------------------------------------------------------------------------------
// ---- slot -> run map of a gathering kernel, held in REGISTERS (round 5) ----
// A thread's item s is slot s * THREADS + tid of its chunk, so the 64-slot group of item s is the same for all lanes of a wave: group s * W + wave (W waves).
// The descriptor of that group -- up to four runs: where the second, third and fourth begin inside the 64 slots, the address of the group's slot 0 and what to
// add from each run border on -- is built by LANE s of the wave and fetched by v_readlane with a constant lane: five scalar registers per item instead of two
// 16-byte LDS broadcasts per record (level 1 and the finish issue ~6 LDS instructions per record; these two were a third of them).
struct GDesc { uint32_t pk, A, d1, d2, d3; };      // pk = o1 | o2 << 8 | o3 << 16 | (more than four runs) << 24; address of slot x = A + x + [x >= o1] d1 + [x >= o2] d2 + [x >= o3] d3
// built from the run list of the chunk: rr[j] = {first slot of run j inside the chunk, index of its first record taken}, rr[nr], rr[nr + 1] = {nv, 0};
// gfirst[g] = last run that begins at or before slot 64 g (the list is written by the threads that own the runs)
__device__ __forceinline__ GDesc gdesc_build(const uint2 *rr, const uint16_t *gfirst, const uint32_t g, const uint32_t nr)
{
	const uint32_t sl = g << 6;
	const uint32_t j0 = gfirst[g], j1 = gfirst[g + 1];
	const uint2 q0 = rr[j0];
	uint32_t o1 = 64u, o2 = 64u, o3 = 64u, b1 = 0u, b2 = 0u, b3 = 0u, found = 0;
	const uint32_t b0 = q0.y + (sl - q0.x);
	for (uint32_t i = j0 + 1; i <= j1 && i < nr; ++i) {
		const uint2 qi = rr[i];
		if (rr[i + 1].x == qi.x || qi.x >= sl + 64u) continue;
		const uint32_t oo = qi.x - sl, bb = qi.y - oo;
		if (found == 0) { o1 = oo; b1 = bb; } else if (found == 1) { o2 = oo; b2 = bb; } else if (found == 2) { o3 = oo; b3 = bb; }
		++found;
	}
	GDesc d;
	d.pk = o1 | (o2 << 8) | (o3 << 16) | (found > 3 ? 1u << 24 : 0u);
	d.A = b0; d.d1 = b1 - b0; d.d2 = b2 - b1; d.d3 = b3 - b2;      // (absent runs: border 64, never reached)
	return d;
}
// the descriptor lane `s` of this wave holds, in scalar registers; address of slot x (< 64) of that group
#define GDESC_LANE(D, s, pk_, A_, d1_, d2_, d3_) \
	const uint32_t pk_ = (uint32_t)__builtin_amdgcn_readlane((int)(D).pk, s), A_ = (uint32_t)__builtin_amdgcn_readlane((int)(D).A, s), d1_ = (uint32_t)__builtin_amdgcn_readlane((int)(D).d1, s), \
	               d2_ = (uint32_t)__builtin_amdgcn_readlane((int)(D).d2, s), d3_ = (uint32_t)__builtin_amdgcn_readlane((int)(D).d3, s);
__device__ __forceinline__ uint32_t gdesc_adr(const uint32_t pk, const uint32_t A, const uint32_t d1, const uint32_t d2, const uint32_t d3, const uint32_t x)
{
	return A + x + (x >= (pk & 0xFFu) ? d1 : 0u) + (x >= ((pk >> 8) & 0xFFu) ? d2 : 0u) + (x >= ((pk >> 16) & 0xFFu) ? d3 : 0u);
}

struct L1GTile { uint32_t out_off, count, hist_base, stride, d, tfr, slot0, pad; };    // tfr: index into tile_fr of the tile's first chunk; slot0: its first slot in the bucket
struct L1GArgs {
	const uint64_t *rec_in; uint64_t *rec_out;
	const L1GTile *tiles;
	uint32_t *runtab;            // this level's run table (as k_l1_local)
	const uint32_t *tab0, *pre0, *tile_fr; uint32_t nchunks0; uint32_t chunk0;      // level 0's table, row prefixes, first runs; index stride of a level-0 chunk
	int D, shift;
	const uint32_t *cbase;       // null, or (one-pass shard) where the records of level-0 chunk c start: cbase[c] instead of c * chunk0
	uint32_t xcd;                // 1: the tiles are listed tile-major (tile j of every bucket, then j + 1) and taken in xcd_order
};
constexpr int L1G_RUNS = 1536;       // runs of one chunk kept in LDS (uniform input: ~512); more: every lane searches the row in HBM

template <int THREADS, int TCH>
__global__ __launch_bounds__(THREADS) void k_l1_gather(L1GArgs A)
{
	constexpr int CHUNK = THREADS * MSD_ITEMS, W = THREADS / 64;
	constexpr int NG = CHUNK / 64;
	__shared__ uint32_t cnt[MSD_MAXBINS];
	__shared__ uint32_t binstart[MSD_MAXBINS];
	__shared__ uint32_t tabst[TCH][MSD_MAXBINS];
	__shared__ __attribute__((aligned(16))) uint64_t srec[CHUNK + 2];
	__shared__ uint32_t wsum[THREADS / 64];
	__shared__ __attribute__((aligned(8))) uint2 rr[L1G_RUNS + 2];       // x = first slot of the run inside the chunk, y = index of its first record taken
	__shared__ uint16_t gfirst[NG + 2];
	__shared__ uint32_t any_w;
	static_assert(THREADS >= MSD_MAXBINS && MSD_ITEMS <= 64, "one bin per thread; one descriptor per lane");
	const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
	const int bins = 1 << A.D;
	const L1GTile tile = A.tiles[A.xcd ? xcd_order(blockIdx.x, gridDim.x) : blockIdx.x];
	const uint32_t nchunks = (tile.count + CHUNK - 1) / CHUNK;       // <= TCH
	const uint32_t *row0 = A.tab0 + (uint64_t)tile.d * A.nchunks0, *prow = A.pre0 + (uint64_t)tile.d * A.nchunks0;
	uint64_t rec[MSD_ITEMS];
	// the 16 records of a thread = slots s * THREADS + tid of chunk c of the tile (all loads back to back, clamped slot).  Barriers inside:
	// every thread calls it.  The run list (rr, gfirst) is only needed until the loads are issued; the slot -> run map itself lives in registers (GDesc).
	auto load_chunk = [&](const uint32_t c) {
		const uint32_t cbase = c * CHUNK;
		const uint32_t nv = tile.count - cbase < (uint32_t)CHUNK ? tile.count - cbase : (uint32_t)CHUNK;
		const uint32_t S0 = tile.slot0 + cbase;
		const uint32_t fr = A.tile_fr[tile.tfr + c];
		uint32_t fre = A.tile_fr[tile.tfr + c + 1]; if (fre >= A.nchunks0) fre = A.nchunks0 - 1;
		const uint32_t nr = fre - fr + 1;
		const uint32_t glast = (nv - 1) >> 6, xlast = (nv - 1) & 63u;
		if (nr > (uint32_t)L1G_RUNS) {
			// (skew: the bucket is rare, its chunk spans more level-0 chunks than the list holds) every lane searches the row's prefixes in HBM
			#pragma unroll                                          // (a loop with a dynamic index would put rec[] into scratch memory)
			for (int s = 0; s < MSD_ITEMS; ++s) {
				uint32_t e = s * THREADS + tid; e = e < nv ? e : nv - 1;
				const uint32_t slot = S0 + e;
				uint32_t lo = fr, hi = fre + 1;                       // largest i in [fr, fre] with prow[i] <= slot
				while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (prow[mid] <= slot) lo = mid; else hi = mid; }
				rec[s] = A.rec_in[(A.cbase ? (uint64_t)A.cbase[lo] : (uint64_t)lo * A.chunk0) + (row0[lo] & 0xFFFFu) + (slot - prow[lo])];
			}
			return;
		}
		__syncthreads();                                              // (the list of the chunk before is no longer read)
		for (uint32_t j = tid; j < nr; j += THREADS) {
			const uint32_t e = row0[fr + j], pre = prow[fr + j], len = e >> 16;
			const uint32_t lo = pre > S0 ? pre - S0 : 0u, skip = S0 > pre ? S0 - pre : 0u;
			uint32_t hi = pre + len > S0 ? pre + len - S0 : 0u; hi = hi < nv ? hi : nv;
			rr[j] = make_uint2(lo < nv ? lo : nv, (A.cbase ? A.cbase[fr + j] : (fr + j) * A.chunk0) + (e & 0xFFFFu) + skip);
			for (uint32_t g = (lo + 63) >> 6; (g << 6) < hi; ++g) gfirst[g] = (uint16_t)j;
		}
		if (tid == 0) { rr[nr] = make_uint2(nv, 0u); rr[nr + 1] = make_uint2(nv, 0u); gfirst[glast + 1] = (uint16_t)(nr - 1); any_w = 0u; }
		__syncthreads();
		GDesc D{0u, 0u, 0u, 0u, 0u};
		{
			const uint32_t g = lane * W + wave;                         // lane s holds the descriptor of this wave's item s
			if (lane < (uint32_t)MSD_ITEMS && g <= glast) { D = gdesc_build(rr, gfirst, g, nr); if (D.pk >> 24) any_w = 1u; }
		}
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) {
			GDESC_LANE(D, s, pk, A0, d1, d2, d3)
			const uint32_t g = (uint32_t)s * W + wave;                  // (the same for the whole wave)
			rec[s] = 0;
			if (g <= glast) rec[s] = A.rec_in[gdesc_adr(pk, A0, d1, d2, d3, g == glast && lane > xlast ? xlast : lane)];
		}
		__syncthreads();
		if (any_w) {
			// groups that meet more than four runs (3e-4 of them on uniform text): their lanes search the list
			#pragma unroll
			for (int s = 0; s < MSD_ITEMS; ++s) {
				if (((uint32_t)__builtin_amdgcn_readlane((int)D.pk, s) >> 24) == 0u) continue;
				uint32_t e = s * THREADS + tid; e = e < nv ? e : nv - 1;
				uint32_t lo = gfirst[e >> 6], hi = nr;                    // largest j with rr[j].x <= e
				while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (rr[mid].x <= e) lo = mid; else hi = mid; }
				rec[s] = A.rec_in[rr[lo].y + (e - rr[lo].x)];
			}
		}
	};
	load_chunk(0);
	for (uint32_t c = 0; c < nchunks; ++c) {
		const uint32_t cbase = c * CHUNK;
		const uint32_t nvalid = tile.count - cbase < (uint32_t)CHUNK ? tile.count - cbase : (uint32_t)CHUNK;
		const uint32_t oidx = tile.out_off + cbase, a = oidx & 1u;   // staging index = place + the parity of the output index: pairs of records are 16-byte aligned on both sides
		for (int b = tid; b < bins; b += THREADS) cnt[b] = 0;
		__syncthreads();
		uint32_t dg[MSD_ITEMS]; uint32_t slot[MSD_ITEMS];
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) dg[s] = (uint32_t)(rec[s] >> A.shift) & (uint32_t)(bins - 1);
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) if ((uint32_t)(s * THREADS) + tid < nvalid) slot[s] = atomicAdd(&cnt[dg[s]], 1u);
		__syncthreads();
		{
			const uint32_t v = (int)tid < bins ? cnt[tid] : 0u;
			uint32_t inc = wave_incl_sum(v);
			uint32_t l = tid & 63;
			if (l == 63) wsum[wave] = inc;
			__syncthreads();
			uint32_t pre = 0;
			for (uint32_t i = 0; i < wave; ++i) pre += wsum[i];
			if ((int)tid < bins) { binstart[tid] = pre + inc - v; tabst[c][tid] = (pre + inc - v) | (v << 16); }
		}
		__syncthreads();
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) if ((uint32_t)(s * THREADS) + tid < nvalid) srec[binstart[dg[s]] + slot[s] + a] = rec[s];
		if (c + 1 < nchunks) load_chunk(c + 1);                       // issued before this chunk is copied out: read and write streams overlap
		__syncthreads();
		uint64_t *o0 = A.rec_out + ((uint64_t)oidx - a);              // even index: element j of the staging array goes to o0[j], j in [a, a + nvalid)
		for (uint32_t p = tid; 2 * p < a + nvalid; p += THREADS) {
			const uint32_t j = 2 * p;
			if (j >= a && j + 2 <= a + nvalid) {
				const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&srec[j]);
				__builtin_nontemporal_store(v.x, &o0[j]); __builtin_nontemporal_store(v.y, &o0[j + 1]);
			} else {
				if (j >= a) __builtin_nontemporal_store(srec[j], &o0[j]);
				if (j + 1 >= a && j + 1 < a + nvalid) __builtin_nontemporal_store(srec[j + 1], &o0[j + 1]);
			}
		}
	}
	__syncthreads();
	for (int d = tid; d < bins; d += THREADS) {
		uint32_t *row = A.runtab + (uint64_t)tile.hist_base + (uint64_t)d * tile.stride;
		for (uint32_t c = 0; c < nchunks; ++c) row[c] = tabst[c][d];
	}
}

// ------------------------------------------------------------------------------------------------
// The same level 1 with TWO workgroups per CU (round 5).  k_l1_gather is one 1024-thread workgroup per CU (its 16 Ki records staged in 128 KB of
// LDS): the chain load -> count -> scan -> place -> store of a chunk is what paces it, and nothing of one chunk overlaps anything of the next.  Two
// smaller workgroups per CU overlap their chains (chunks of 8 Ki records: 11.3 -> 10.05 ms in round 3), but halving the chunk halves the runs the
// finish gathers.  Here the chunk stays 16 Ki records and the STAGING shrinks instead: a 512-thread workgroup keeps its 32 records per thread in
// registers, learns every record's place inside the sorted chunk from the counting sort (an arrival slot per record, 14 bits), and sends the chunk out
// in PASSES of L1M_CAP places through a 64-KB staging array (which the run list of the gather shares: it is dead by then): pass k takes the records whose
// place lies in [k CAP, (k + 1) CAP).  70 KB of LDS, at most 128 registers: two workgroups per CU; same input, same output, same run table as k_l1_gather.
// ------------------------------------------------------------------------------------------------
constexpr int L1M_THREADS = 512, L1M_ITEMS = L1_CHUNK / L1M_THREADS, L1M_CAP = 8192;
static_assert(L1M_ITEMS * L1M_THREADS == L1_CHUNK && L1M_ITEMS % 2 == 0, "k_l1_gather_mp: the chunk is level 1's");

// one chunk of k_l1_gather_mp.  FULL: all CHUNK slots hold records (every chunk of a bucket but its last): no per-item guards
struct L1MShared {
	uint32_t cnt[MSD_MAXBINS];                                            // arrival counters, then the first place of every digit
	uint32_t wsum[L1M_THREADS / 64];
	uint32_t any_w;                                                       // a 64-slot group of this chunk meets more than four runs
	union {                                                               // the run list is dead once the records are requested; the staging array is first written two barriers later
		struct { __attribute__((aligned(8))) uint2 rr[L1G_RUNS + 2]; uint16_t gfirst[L1_CHUNK / 64 + 2]; } map;
		__attribute__((aligned(16))) uint64_t srec[L1M_CAP + 2];
	} u;
};
template <bool FULL>
__device__ __forceinline__ void l1m_chunk(const L1GArgs &A, const L1GTile &tile, const uint32_t c, L1MShared &S, uint32_t *tabst_c, const uint32_t *row0, const uint32_t *prow)
{
	constexpr int THREADS = L1M_THREADS, ITEMS = L1M_ITEMS, CHUNK = L1_CHUNK, CAP = L1M_CAP, W = THREADS / 64;
	uint32_t tid = threadIdx.x;
	asm volatile("" : "+v"(tid));                                         // (opaque per chunk: the 32 slot numbers, group indices and LDS addresses derived from it were hoisted out of the chunk loop and spilled)
	const uint32_t wave = tid >> 6, lane = tid & 63u;
	const int bins = 1 << A.D;
	const uint32_t cbase = c * CHUNK;
	const uint32_t nv = FULL ? (uint32_t)CHUNK : tile.count - cbase;
	const uint32_t glast = (nv - 1) >> 6, xlast = (nv - 1) & 63u;
	uint64_t rec[ITEMS];
	// ---- the chunk's records: slots s * THREADS + tid, through the slot -> run map ----
	{
		const uint32_t S0 = tile.slot0 + cbase;
		const uint32_t fr = A.tile_fr[tile.tfr + c];
		uint32_t fre = A.tile_fr[tile.tfr + c + 1]; if (fre >= A.nchunks0) fre = A.nchunks0 - 1;
		const uint32_t nr = fre - fr + 1;
		if (nr > (uint32_t)L1G_RUNS) {
			// (skew: the bucket is rare, its chunk spans more level-0 chunks than the list holds) every lane searches the row's prefixes in HBM
			#pragma unroll
			for (int s = 0; s < ITEMS; ++s) {
				uint32_t e = s * THREADS + tid; if (!FULL) e = e < nv ? e : nv - 1;
				const uint32_t slot = S0 + e;
				uint32_t lo = fr, hi = fre + 1;                           // largest i in [fr, fre] with prow[i] <= slot
				while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (prow[mid] <= slot) lo = mid; else hi = mid; }
				rec[s] = A.rec_in[(A.cbase ? (uint64_t)A.cbase[lo] : (uint64_t)lo * A.chunk0) + (row0[lo] & 0xFFFFu) + (slot - prow[lo])];
			}
			for (int b = tid; b < bins; b += THREADS) S.cnt[b] = 0;
		} else {
			for (uint32_t j = tid; j < nr; j += THREADS) {
				const uint32_t e = row0[fr + j], pre = prow[fr + j], len = e >> 16;
				const uint32_t lo = pre > S0 ? pre - S0 : 0u, skip = S0 > pre ? S0 - pre : 0u;
				uint32_t hi = pre + len > S0 ? pre + len - S0 : 0u; hi = hi < nv ? hi : nv;
				S.u.map.rr[j] = make_uint2(lo < nv ? lo : nv, (A.cbase ? A.cbase[fr + j] : (fr + j) * A.chunk0) + (e & 0xFFFFu) + skip);
				for (uint32_t g = (lo + 63) >> 6; (g << 6) < hi; ++g) S.u.map.gfirst[g] = (uint16_t)j;
			}
			if (tid == 0) { S.u.map.rr[nr] = make_uint2(nv, 0u); S.u.map.rr[nr + 1] = make_uint2(nv, 0u); S.u.map.gfirst[glast + 1] = (uint16_t)(nr - 1); S.any_w = 0u; }
			for (int b = tid; b < bins; b += THREADS) S.cnt[b] = 0;
			__syncthreads();
			GDesc D{0u, 0u, 0u, 0u, 0u};
			{
				const uint32_t g = lane * W + wave;                         // lane s holds the descriptor of this wave's item s
				if (lane < (uint32_t)ITEMS && g <= glast) { D = gdesc_build(S.u.map.rr, S.u.map.gfirst, g, nr); if (D.pk >> 24) S.any_w = 1u; }
			}
			#pragma unroll
			for (int s = 0; s < ITEMS; ++s) {
				GDESC_LANE(D, s, pk, A0, d1, d2, d3)
				if (FULL) rec[s] = A.rec_in[gdesc_adr(pk, A0, d1, d2, d3, lane)];
				else {
					const uint32_t g = (uint32_t)s * W + wave;              // (the same for the whole wave: a group behind the chunk's last record is skipped by all its lanes)
					rec[s] = 0;
					if (g <= glast) rec[s] = A.rec_in[gdesc_adr(pk, A0, d1, d2, d3, g == glast && lane > xlast ? xlast : lane)];
				}
			}
			__syncthreads();                                              // (any_w is set; the list stays in place until the barrier behind the loads)
			if (S.any_w) {
				// groups that meet more than four runs (3e-4 of them on uniform text): their lanes search the list
				#pragma unroll
				for (int s = 0; s < ITEMS; ++s) {
					if (((uint32_t)__builtin_amdgcn_readlane((int)D.pk, s) >> 24) == 0u) continue;
					uint32_t e = s * THREADS + tid; if (!FULL) e = e < nv ? e : nv - 1;
					uint32_t lo = S.u.map.gfirst[e >> 6], hi = nr;                // largest j with rr[j].x <= e
					while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (S.u.map.rr[mid].x <= e) lo = mid; else hi = mid; }
					rec[s] = A.rec_in[S.u.map.rr[lo].y + (e - S.u.map.rr[lo].x)];
				}
			}
		}
	}
	__syncthreads();                                                      // (the counters are zero; the staging array of the chunk before has left; the run list is dead)
	// ---- count: arrival slot of every record inside its digit (two per register) ----
	uint32_t place[ITEMS / 2];
	#pragma unroll
	for (int s = 0; s < ITEMS; s += 2) {
		uint32_t p0 = 0u, p1 = 0u;
		if (FULL || (uint32_t)(s * THREADS) + tid < nv) p0 = atomicAdd(&S.cnt[(uint32_t)(rec[s] >> A.shift) & (uint32_t)(bins - 1)], 1u);
		if (FULL || (uint32_t)((s + 1) * THREADS) + tid < nv) p1 = atomicAdd(&S.cnt[(uint32_t)(rec[s + 1] >> A.shift) & (uint32_t)(bins - 1)], 1u);
		place[s >> 1] = p0 | (p1 << 16);
	}
	__syncthreads();
	{
		const uint32_t v = (int)tid < bins ? S.cnt[tid] : 0u;
		const uint32_t inc = wave_incl_sum(v);
		if ((tid & 63u) == 63u) S.wsum[wave] = inc;
		__syncthreads();
		uint32_t pre = 0;
		for (uint32_t i = 0; i < wave; ++i) pre += S.wsum[i];
		if ((int)tid < bins) { S.cnt[tid] = pre + inc - v; tabst_c[tid] = (pre + inc - v) | (v << 16); }
	}
	__syncthreads();
	// ---- every record's place inside the sorted chunk = first place of its digit + arrival slot (0xFFFF: no record) ----
	// (the counters' addresses are computed again, not kept from the count phase: 32 registers held across the scan were 22 records spilled)
	#pragma unroll
	for (int s = 0; s < ITEMS; ++s) { uint32_t h = (uint32_t)(rec[s] >> 32); asm volatile("" : "+v"(h)); rec[s] = (rec[s] & 0xFFFFFFFFull) | ((uint64_t)h << 32); }
	#pragma unroll
	for (int s = 0; s < ITEMS; s += 2) {
		uint32_t p0 = (place[s >> 1] & 0xFFFFu) + S.cnt[(uint32_t)(rec[s] >> A.shift) & (uint32_t)(bins - 1)];
		uint32_t p1 = (place[s >> 1] >> 16) + S.cnt[(uint32_t)(rec[s + 1] >> A.shift) & (uint32_t)(bins - 1)];
		if (!FULL) { if ((uint32_t)(s * THREADS) + tid >= nv) p0 = 0xFFFFu; if ((uint32_t)((s + 1) * THREADS) + tid >= nv) p1 = 0xFFFFu; }
		place[s >> 1] = p0 | (p1 << 16);
	}
	// ---- the chunk leaves in passes of CAP places; staging index = place + the parity of the output index, so that pairs of records are 16-byte aligned on both sides ----
	for (uint32_t k0 = 0; k0 < nv; k0 += CAP) {
		if (k0) __syncthreads();                                          // (the pass before has been read out of the staging array)
		const uint32_t oidx = tile.out_off + cbase + k0, a = oidx & 1u;
		#pragma unroll
		for (int s = 0; s < ITEMS; s += 2) {
			const uint32_t q0 = (place[s >> 1] & 0xFFFFu) - k0, q1 = (place[s >> 1] >> 16) - k0;      // (no record: 0xFFFF - k0 >= CAP)
			if (q0 < (uint32_t)CAP) S.u.srec[q0 + a] = rec[s];
			if (q1 < (uint32_t)CAP) S.u.srec[q1 + a] = rec[s + 1];
		}
		__syncthreads();
		const uint32_t nk = nv - k0 < (uint32_t)CAP ? nv - k0 : (uint32_t)CAP;
		uint64_t *o0 = A.rec_out + ((uint64_t)oidx - a);                  // even index: 16-byte aligned; element j of the staging array goes to o0[j], j in [a, a + nk)
		for (uint32_t p = tid; 2 * p < a + nk; p += THREADS) {
			const uint32_t j = 2 * p;
			if (j >= a && j + 2 <= a + nk) {
				const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&S.u.srec[j]);
				__builtin_nontemporal_store(v.x, &o0[j]); __builtin_nontemporal_store(v.y, &o0[j + 1]);
			} else {
				if (j >= a) __builtin_nontemporal_store(S.u.srec[j], &o0[j]);
				if (j + 1 >= a && j + 1 < a + nk) __builtin_nontemporal_store(S.u.srec[j + 1], &o0[j + 1]);
			}
		}
	}
}

template <int TCH>
__global__ __launch_bounds__(L1M_THREADS, 4) void k_l1_gather_mp(L1GArgs A)
{
	__shared__ L1MShared S;
	__shared__ uint32_t tabst[TCH][MSD_MAXBINS];
	static_assert(L1M_THREADS >= MSD_MAXBINS && L1_CHUNK / 64 <= L1M_THREADS && 0xFFFF - (L1_CHUNK - 1) >= L1M_CAP, "one bin / one group per thread; the 'no record' place is in no pass");
	const uint32_t tid = threadIdx.x;
	const int bins = 1 << A.D;
	const L1GTile tile = A.tiles[A.xcd ? xcd_order(blockIdx.x, gridDim.x) : blockIdx.x];
	const uint32_t nchunks = (tile.count + L1_CHUNK - 1) / L1_CHUNK;       // <= TCH
	const uint32_t *row0 = A.tab0 + (uint64_t)tile.d * A.nchunks0, *prow = A.pre0 + (uint64_t)tile.d * A.nchunks0;
	for (uint32_t c = 0; c < nchunks; ++c) {
		if (tile.count - c * L1_CHUNK >= (uint32_t)L1_CHUNK) l1m_chunk<true>(A, tile, c, S, tabst[c], row0, prow);
		else l1m_chunk<false>(A, tile, c, S, tabst[c], row0, prow);
	}
	__syncthreads();
	for (int d = tid; d < bins; d += L1M_THREADS) {
		uint32_t *row = A.runtab + (uint64_t)tile.hist_base + (uint64_t)d * tile.stride;
		for (uint32_t c = 0; c < nchunks; ++c) row[c] = tabst[c][d];
	}
}

// sizes and starts of the final buckets from the run table: one workgroup per level-0 bucket; wave w walks the
// rows d = w, w+4, ... (coalesced) and writes, for the gathering finish, the same row as start | prefix << 14
// (prefix = records of the row's earlier runs, clamped: rows of buckets above the finish's capacity are never read
// in that form).  Then the 2^D sizes are scanned.  btab = {start[nb], tab_off[nb], nch[nb]}.
constexpr uint32_t RUN_PRE_MAX = (1u << 18) - 1;
__global__ __launch_bounds__(256) void k_run_totals(const uint32_t *runtab, uint32_t *runpre, const uint32_t *btab, uint32_t nb, int D, uint32_t len, uint32_t *lvl_start, uint32_t *lvl_end = nullptr)
{
	__shared__ uint32_t sz[MSD_MAXBINS];
	__shared__ uint32_t wsum[4];
	const uint32_t b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t bins = 1u << D;
	const uint32_t nch = btab[2 * nb + b];
	const uint32_t *tab = runtab + btab[nb + b];
	uint32_t *pre = runpre + btab[nb + b];
	for (uint32_t d = wave; d < bins; d += 4) {
		uint32_t run = 0;
		for (uint32_t c0 = 0; c0 < nch; c0 += 64) {
			const uint32_t c = c0 + lane;
			const uint32_t e = c < nch ? tab[(uint64_t)d * nch + c] : 0u;
			const uint32_t cnt = e >> 16;
			const uint32_t inc = wave_incl_sum(cnt);
			const uint32_t ex = run + inc - cnt;
			if (c < nch) pre[(uint64_t)d * nch + c] = (e & 0x3FFFu) | ((ex < RUN_PRE_MAX ? ex : RUN_PRE_MAX) << 14);
			run += __shfl(inc, 63, 64);
		}
		if (lane == 0) sz[d] = run;
	}
	__syncthreads();
	const uint32_t v = tid < bins ? sz[tid] : 0u;
	uint32_t inc = wave_incl_sum(v);
	if (lane == 63) wsum[wave] = inc;
	__syncthreads();
	uint32_t pre0 = 0;
	for (uint32_t i = 0; i < wave; ++i) pre0 += wsum[i];
	if (tid < bins) lvl_start[((uint64_t)b << D) + tid] = btab[b] + pre0 + inc - v;
	if (lvl_end && tid < bins) lvl_end[((uint64_t)b << D) + tid] = btab[b] + pre0 + inc;         // (level 2: the buckets of this launch are not neighbours in rank order)
	if (b == nb - 1 && tid == 0 && !lvl_end) lvl_start[(uint64_t)nb << D] = len;
}

__global__ void k_add5_u64(const unsigned long long *src, unsigned long long *dst) { if (threadIdx.x < 5) dst[threadIdx.x] += src[threadIdx.x]; }
// totals[d] = number of records with level-0 digit d (sum of row d of the [digit][tile] counts)
__global__ __launch_bounds__(256) void k_row_totals(const uint32_t *hist, uint32_t ntiles, unsigned long long *totals)
{
	__shared__ unsigned long long part[256];
	unsigned long long s = 0;
	for (uint32_t t = threadIdx.x; t < ntiles; t += 256) s += hist[(uint64_t)blockIdx.x * ntiles + t];
	part[threadIdx.x] = s;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
	if (threadIdx.x == 0) totals[blockIdx.x] = part[0];
}

// symbol histogram of the block from the level-0 digit totals: the leading base of a rotation is the top two bits of
// its digit.  pacterm: symbol = base + 1; the terminator position was counted as base 0 (its window reads as zeros).
__global__ __launch_bounds__(256) void k_symhist_from_totals(const unsigned long long *totals, uint32_t bins, int D, int pacterm, int has_term, unsigned long long *hist256)
{
	__shared__ unsigned long long part[4][256];
	unsigned long long s[4] = {0, 0, 0, 0};
	for (uint32_t d = threadIdx.x; d < bins; d += 256) s[d >> (D - 2)] += totals[d];
	for (int c = 0; c < 4; ++c) part[c][threadIdx.x] = s[c];
	__syncthreads();
	if (threadIdx.x < 4) {
		unsigned long long t = 0;
		for (int i = 0; i < 256; ++i) t += part[threadIdx.x][i];
		if (pacterm) {
			if (threadIdx.x == 0 && has_term) { t -= 1; atomicAdd(&hist256[0], 1ull); }
			atomicAdd(&hist256[threadIdx.x + 1], t);
		} else atomicAdd(&hist256[threadIdx.x], t);
	}
}

// starts of the buckets leaving a level: bucket_tab = {start[nb], hist_off[nb], ntiles[nb]} of the
// buckets entering it; out[(b << D) | d] = start of digit d inside bucket b; out[nb << D] = len
__global__ void k_next_starts(const uint32_t *scanned, const uint32_t *bucket_tab, uint32_t nb, int D, uint32_t len, uint32_t *out)
{
	uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t total = nb << D;
	if (g > total) return;
	if (g == total) { out[g] = len; return; }
	uint32_t b = g >> D, d = g & ((1u << D) - 1);
	uint32_t nt = bucket_tab[2 * nb + b];
	out[g] = nt ? scanned[bucket_tab[nb + b] + (uint64_t)d * nt] : bucket_tab[b];
}

// starts of the nb buckets leaving level 0 (nb need not be a power of two); out[nb] = len
__global__ void k_l0_starts(const uint32_t *scanned, uint32_t ntiles, uint32_t nb, uint32_t len, uint32_t *out)
{
	uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g > nb) return;
	out[g] = g == nb ? len : scanned[(uint64_t)g * ntiles];
}

// ------------------------------------------------------------------------------------------------
// wave sort: one WAVE sorts one sub-bucket (<= 512 records, ~180 on uniform input) with wave-level
// synchronisation only.  The remaining key bits are close to uniform, so one bin-scatter (LDS
// atomics into 256 bins) leaves almost every record in its final bin; then rank inside the bin by
// (key field, arrival slot) and permute in place.  Records whose whole key field ties are marked
// in the tie bitmap for k_tie_fix.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct SortOut { SaOut sa; uint8_t *bwt; uint64_t start; uint32_t *p0; uint32_t *overflow; uint32_t overflow_cap; uint32_t *tiebits; int pb; SampleOut smp; };

constexpr uint32_t OVF_ONE_BIN = 0x80000000u;
// a sub-bucket the in-kernel sort cannot finish: (start, size) goes to the overflow list (lane 0 only)
// (word 1 of the table: set once a range was pushed whose records are only known to share their BUCKET; ranges that are one bin of a finish
// share the bin's bits as well, and the rounds on the list start behind what every range shares)
__device__ __forceinline__ void ovf_push(const SortOut &O, uint32_t b0, uint32_t m, bool one_bin = false)
{
	uint32_t k = atomicAdd(&O.overflow[0], 1u);
	// 8-byte aligned pairs: u64 = (size | one bin << 31) << 32 | start (a range is below 2^31 records: the finish that lists a bin holds a few thousand, the
	// whole-bucket paths stop at the sort's 2^32 - 1 records ... of which a single bucket never holds half)
	if (k < O.overflow_cap) { O.overflow[2 + 2 * k] = b0; O.overflow[3 + 2 * k] = m | (one_bin ? OVF_ONE_BIN : 0u); }
	if (!one_bin) atomicOr(&O.overflow[1], (m & OVF_ONE_BIN) ? 3u : 1u);       // (bit 1: a range of 2^31 records and more -- one bucket of a text of one symbol -- the flag bit is part of its size)
}

// one wave sorts the m <= LS_CAP records it holds in r (lane l: records l, l+64, ...), destined
// for the slots [b0, b0+m).  outb (>= m u64) and cb (2*LS_BINS u32) are this wave's private LDS.
// bins_deep: a bin of this sort knows at least the bits a bin of the main finish knows (the host's arithmetic): its overflow ranges may start where theirs do
template <bool WRITE_SA>
__device__ __forceinline__ void wave_sort_bucket(uint32_t b0, uint32_t m, int sortbits, uint64_t *outb, uint32_t *cb, const SortOut &O,
                                                  const uint64_t (&r)[LS_CAP / 64], bool bins_deep)
{
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t *cnt = cb, *bst = cb + LS_BINS;
	const SaOut sa = O.sa; uint8_t *bwt = O.bwt; const uint64_t start = O.start; uint32_t *p0 = O.p0;
	const int REC_KEY_SHIFT = rec_key_shift(O.pb);
	const uint64_t posmask = (1ull << O.pb) - 1;
	uint32_t *tiebits = O.tiebits;
	if (m == 0) return;
	const int binbits = sortbits < LS_BINBITS ? (sortbits > 0 ? sortbits : 0) : LS_BINBITS;
	const int bshift = REC_KEY_SHIFT + sortbits - binbits;
	const uint32_t bmask = (1u << binbits) - 1;
	constexpr int PER = LS_CAP / 64;               // 8 records per lane; rounds t with t*64 >= m are skipped (m is wave-uniform)
	uint32_t sl[PER];
	#pragma unroll
	for (int k = 0; k < LS_BINS / 64; ++k) cnt[k * 64 + lane] = 0;
	wave_sync();
	#pragma unroll
	for (int t = 0; t < PER; ++t) {
		uint32_t e = t * 64 + lane;
		sl[t] = 0;
		if (t * 64 < (int)m) { if (e < m) sl[t] = atomicAdd(&cnt[(uint32_t)(r[t] >> bshift) & bmask], 1u); }
	}
	wave_sync();
	{
		constexpr int per = LS_BINS / 64;           // consecutive bins per lane
		uint32_t loc[per]; uint32_t sum = 0;
		#pragma unroll
		for (int k = 0; k < per; ++k) { loc[k] = cnt[lane * per + k]; sum += loc[k]; }
		uint32_t run = wave_incl_sum(sum) - sum;
		#pragma unroll
		for (int k = 0; k < per; ++k) { bst[lane * per + k] = run; run += loc[k]; }
	}
	wave_sync();
	#pragma unroll
	for (int t = 0; t < PER; ++t) {
		uint32_t e = t * 64 + lane;
		if (t * 64 < (int)m) { if (e < m) outb[bst[(uint32_t)(r[t] >> bshift) & bmask] + sl[t]] = r[t]; }
	}
	wave_sync();
	// rank inside the bin: position by (key field, slot); bins of <= 4 records without a loop
	bool ovf = false;
	uint32_t dest[PER];
	#pragma unroll
	for (int t = 0; t < PER; ++t) {
		uint32_t e = t * 64 + lane;
		dest[t] = 0;
		if (t * 64 < (int)m) {
			if (e < m) {
				uint32_t bin = (uint32_t)(r[t] >> bshift) & bmask;
				uint32_t c = cnt[bin], s0 = bst[bin], less = 0;
				if (c > 1) {
					const uint64_t kx = r[t] >> REC_KEY_SHIFT;
					if (c <= 4) {
						#pragma unroll
						for (uint32_t j = 0; j < 4; ++j) {
							uint64_t ky = outb[s0 + (j < c ? j : c - 1)] >> REC_KEY_SHIFT;
							less += (j < c && (ky < kx || (ky == kx && j < sl[t]))) ? 1u : 0u;
						}
					} else if (c <= (uint32_t)LS_TIE_CAP) {
						for (uint32_t j = 0; j < c; ++j) {
							uint64_t ky = outb[s0 + j] >> REC_KEY_SHIFT;
							less += (ky < kx || (ky == kx && j < sl[t])) ? 1u : 0u;
						}
					} else { ovf = true; less = sl[t]; }
				}
				dest[t] = s0 + less;
			}
		}
	}
	wave_sync();
	#pragma unroll
	for (int t = 0; t < PER; ++t) if (t * 64 + lane < m) outb[dest[t]] = r[t];
	wave_sync();
	const bool any_hot = __ballot(ovf) != 0;
	if (any_hot) {
		// bins beyond LS_TIE_CAP (records that share every bit down to the bin: a repeat family's k-mer): THOSE go to the list-based rounds, each a range
		// of its own that starts at the bin's depth; the rest of the sub-bucket is finished here.  (The whole sub-bucket used to go, at the depth of its
		// bucket: 278 M of the 360 M listed suffixes of a genome-like 3-Gbase text were such ranges.)
		constexpr int per = LS_BINS / 64;
		#pragma unroll
		for (int k = 0; k < per; ++k) { const uint32_t c = cnt[lane * per + k]; if (c > (uint32_t)LS_TIE_CAP) ovf_push(O, b0 + bst[lane * per + k], c, bins_deep); }
	}
	// tie groups (equal key field): bit p of the tie bitmap = "record p has the key field of
	// record p-1"; k_tie_fix resolves the runs against the text.  Scattered single-bit atomics:
	// no hot address.  tied: records that k_tie_fix will look at (they need their SA entry).
	#pragma unroll
	for (int t = 0; t < PER; ++t) {
		uint32_t e = t * 64 + lane;
		if (t * 64 < (int)m) {
			if (e < m) {
				const uint64_t q = outb[e];
				const uint64_t kf = q >> REC_KEY_SHIFT;
				const uint64_t i = q & posmask;
				if (any_hot && cnt[(uint32_t)(q >> bshift) & bmask] > (uint32_t)LS_TIE_CAP) {      // in a listed bin: the rounds read its position, write its byte, say its sample
					sa_set(sa, b0 + e, i);
					continue;
				}
				bool sp = e > 0 && (outb[e - 1] >> REC_KEY_SHIFT) == kf;
				bool sn = e + 1 < m && (outb[e + 1] >> REC_KEY_SHIFT) == kf;
				if (sp) { uint32_t g = b0 + e; atomicOr(&tiebits[g >> 5], 1u << (g & 31)); }
				if (WRITE_SA || sp || sn) sa_set(sa, b0 + e, i);       // without consumers only k_tie_fix reads positions
				bwt[b0 + e] = (uint8_t)((q >> O.pb) & 7u);
				if (i == start) *p0 = b0 + e;
				if (!(sp || sn) && sample_hit(O.smp, (uint32_t)i)) sample_put(O.smp, b0 + e, i);     // (tied records: k_tie_fix, once their order is known)
			}
		}
	}
}

// FINISH: one workgroup per final bucket (<= F_CAP records), read ONCE into registers (14 per thread)
// and counting-sorted inside LDS on the top F_BINBITS of its free key bits (8192 bins of 16-bit
// counters for ~5.7 K records: almost every record is alone in its bin).  What moves through LDS is
// a 4-byte entry (remaining key bits | index inside the bucket), not the record: the owner of a
// record scans its bin, learns its final place and whether its whole key field ties with a
// neighbour, and drops the BWT byte (and the SA entry) there; the bucket leaves as coalesced rows.
// Buckets above F_CAP are listed for k_bucket_finish; a bucket with a bin above LS_TIE_CAP (a 16-mer
// repeated inside the bucket) goes to the overflow list.
// THREADS = 512: every final bucket, 6144 records at most (two workgroups per CU).  THREADS = 1024: the buckets
// between 6144 and 12288 records that the first launch listed (`midlist`; skewed base composition makes many),
// one workgroup per CU.
// debug build (-DB3M_CHECK_LDS, `make debug`): every LDS index of the finish is checked against its array and traps
// ... and so is every bucket id taken from a list and every bucket's record range (GLB_CHK)
#ifdef B3M_CHECK_LDS
#define LDS_CHK(idx, cap) do { if ((uint32_t)(idx) >= (uint32_t)(cap)) __builtin_trap(); } while (0)
#define GLB_CHK(cond) do { if (!(cond)) __builtin_trap(); } while (0)
#else
#define LDS_CHK(idx, cap) do { } while (0)
#define GLB_CHK(cond) do { } while (0)
#endif

constexpr int G_NCH = 512;               // runs (chunks of the level-0 bucket) a gathering finish maps in LDS

// where the records of a final bucket are: GATHER = false: rec[bstart[b] ...) contiguous (no level 1, or a bucket the
// compaction kernel has copied together); GATHER = true: row (b >> D1, b & (2^D1 - 1)) of the level-1 run table
struct FinishSrc {
	const uint64_t *rec;
	const uint32_t *bstart;      // starts of the final buckets in the OUTPUT (rank) order, nbF + 1 entries
	const uint32_t *bend;        // null: bucket b ends where b + 1 starts; else its end (level 2: the sub-buckets of listed buckets, which are not neighbours in rank order)
	const uint32_t *runtab;      // GATHER: run table of level 1 (start | length << 16), read by the compaction kernel
	const uint32_t *runpre;      // GATHER: the same rows as start | records before the run << 14 (k_run_totals)
	const uint32_t *btab;        // GATHER: {start[nb0], tab_off[nb0], nch[nb0]} of the level-0 buckets
	uint32_t nb0; int D1; uint32_t chunk;
	uint32_t nbF, nrec;          // number of final buckets and of records (bounds of every bucket id / record range: checked in the debug build)
	uint32_t xcd;                // 1: workgroups take the buckets of a launch in xcd_order
};
__device__ __forceinline__ uint32_t bucket_end(const FinishSrc &S, uint32_t b) { return S.bend ? S.bend[b] : S.bstart[b + 1]; }
struct FinishLists { uint32_t *midlist; uint32_t mid_cap; uint32_t *biglist; uint32_t big_cap; uint32_t *hugelist; uint32_t huge_cap; uint32_t huge_min; };

// what a gathering finish needs to know of a bucket before it starts: fetched one bucket AHEAD by k_finish_rows, so that
// the only global-memory round trip a bucket waits for is that of its records
struct RowAhead { uint32_t s1; uint32_t v, vn; };          // end of the bucket in rank order; this thread's word of the row and the next one
struct RowBucket { uint32_t nch, tab_off, start; };        // level-0 bucket: its chunks, its rows in the run table, its first record

template <bool WRITE_SA, int THREADS, bool GATHER>
__device__ __forceinline__ void finish_bucket(const uint32_t b, const uint32_t s0, const uint32_t m, const RowBucket &RB, const RowAhead &RA,
                                              const FinishSrc &S, int fbits, const FinishLists *Lp, const SortOut &O)
{
	constexpr int F_THREADS = THREADS, F_CAP = THREADS * 12, F_PER = 12, F_WAVES = THREADS / 64;
	constexpr int IDXB = F_CAP <= 8192 ? 13 : 14;             // index field of an entry
	constexpr int ENT_N = F_CAP + 4, CW_N = F_BINS / 2 + 4, OBWT_N = F_CAP + 8, GF_N = F_CAP / 64 + 2;
	__shared__ __attribute__((aligned(16))) uint32_t ent[ENT_N];          // (low key bits << IDXB) | index, in bin order
	__shared__ __attribute__((aligned(16))) uint32_t cw[CW_N];            // 16-bit counters, then exclusive starts, two per word
	__shared__ __attribute__((aligned(16))) uint8_t obwt[OBWT_N];
	__shared__ uint32_t osa[WRITE_SA ? F_CAP : 1];
	__shared__ uint32_t wsum[F_WAVES];
	__shared__ __attribute__((aligned(8))) uint2 rr[GATHER ? G_NCH + 1 : 1];   // GATHER: per run {records of the bucket before it, index of its first record}
	__shared__ uint16_t gfirst[GATHER ? GF_N : 1];            // GATHER: run holding slot 64 g
	__shared__ __attribute__((aligned(16))) uint4 gdesc[GATHER ? GF_N : 1];    // GATHER: where the 64 slots of group g are (see below)
	__shared__ uint32_t gdesc2[GATHER ? GF_N : 1];
	__shared__ uint32_t s_more;
	static_assert(F_CAP <= (1 << IDXB), "index field of an entry");
	static_assert(F_PER % 6 == 0, "rank phase works on six records at a time");
	static_assert(!GATHER || THREADS >= G_NCH, "one run per thread");
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	if (m == 0) return;
	const uint32_t nch = GATHER ? RB.nch : 0u;
	if (m > (uint32_t)F_CAP || nch > (uint32_t)G_NCH) {
		if (tid == 0) {
			const FinishLists L = *Lp;                          // cold: the lists live in device memory, not in the kernel's registers
			if (L.midlist && m <= 2u * F_CAP) { uint32_t k = atomicAdd(&L.midlist[0], 1u); if (k < L.mid_cap) L.midlist[1 + k] = b; }
			else if (m > L.huge_min) {    // one workgroup would take too long: the whole bucket goes to the overflow list (k_huge_spill)
				uint32_t k = atomicAdd(&L.hugelist[0], 1u); if (k < L.huge_cap) L.hugelist[1 + k] = b;
				ovf_push(O, s0, m);
			} else { uint32_t k = atomicAdd(&L.biglist[0], 1u); if (k < L.big_cap) L.biglist[1 + k] = b; }
		}
		return;
	}
	uint64_t r[F_PER];
	if (GATHER) {
		// ---- map the bucket's slots onto the runs of its row (start | records before the run << 14, k_run_totals) ----
		const uint32_t glast = (m - 1) >> 6;
		{
			const uint32_t v = RA.v, vn = tid + 1 < nch ? RA.vn : (m << 14);
			const uint32_t pre = v >> 14, end = vn >> 14;
			if (tid < nch) {
				LDS_CHK(tid, G_NCH);
				rr[tid] = make_uint2(pre, RB.start + tid * S.chunk + (v & 0x3FFFu));
				for (uint32_t g = (pre + 63) >> 6; (g << 6) < end; ++g) { LDS_CHK(g, GF_N); gfirst[g] = (uint16_t)tid; }     // the run holding slot 64 g (empty runs write nothing)
			}
			if (tid == 0) { rr[nch] = make_uint2(m, 0u); LDS_CHK(glast + 1, GF_N); gfirst[glast + 1] = (uint16_t)(nch - 1); s_more = 0u; }
		}
		__syncthreads();
		// one descriptor per group of 64 slots: the group starts inside run j0 and up to three more runs begin inside it
		// (runs are ~64 records): {o1 | o2 << 8 | o3 << 16, A, B - o1, C - o2} and D - o3: slot x of the group is record
		// A + x below o1, (B - o1) + x below o2, ...  A bucket with a group that meets more than four runs (many short
		// runs: skewed composition) is listed and takes the copy-together path instead.
		if (tid <= glast) {
			const uint32_t sl = tid << 6;
			LDS_CHK(tid + 1, GF_N);
			const uint32_t j0 = gfirst[tid], j1 = gfirst[tid + 1];
			const uint2 q0 = rr[j0];
			uint32_t o1 = 64u, o2 = 64u, o3 = 64u, b1 = 0u, b2 = 0u, b3 = 0u, found = 0;
			for (uint32_t i = j0 + 1; i <= j1 && i < nch; ++i) {
				LDS_CHK(i + 1, G_NCH + 1);
				const uint2 qi = rr[i];
				if (rr[i + 1].x == qi.x || qi.x >= sl + 64u) continue;         // empty, or begins beyond the group
				const uint32_t oo = qi.x - sl, bb = qi.y - oo;
				if (found == 0) { o1 = oo; b1 = bb; } else if (found == 1) { o2 = oo; b2 = bb; } else if (found == 2) { o3 = oo; b3 = bb; }
				++found;
			}
			if (found > 3) s_more = 1u;
			LDS_CHK(tid, GF_N);
			gdesc[tid] = make_uint4(o1 | (o2 << 8) | (o3 << 16), q0.y + (sl - q0.x), b1, b2);
			gdesc2[tid] = b3;
		}
		__syncthreads();
		if (s_more) {
			if (tid == 0) { const FinishLists L = *Lp; uint32_t k = atomicAdd(&L.midlist[0], 1u); if (k < L.mid_cap) L.midlist[1 + k] = b; }
			return;
		}
		uint32_t adr[F_PER];
		#pragma unroll
		for (int k = 0; k < F_PER; ++k) {
			const uint32_t eu = k * F_THREADS + tid;
			const uint32_t e = eu < m ? eu : m - 1;
			uint32_t g = (uint32_t)(k * F_WAVES) + wave; g = g < glast ? g : glast;       // wave-uniform
			LDS_CHK(g, GF_N);
			const uint4 d = gdesc[g];
			const uint32_t d3 = gdesc2[g];
			const uint32_t x = e - (g << 6);
			const uint32_t o1 = d.x & 0xFFu, o2 = (d.x >> 8) & 0xFFu, o3 = d.x >> 16;
			adr[k] = (x >= o3 ? d3 : x >= o2 ? d.w : x >= o1 ? d.z : d.y) + x;
		}
		#pragma unroll
		for (int k = 0; k < F_PER; ++k) r[k] = S.rec[adr[k]];
	} else {
		#pragma unroll
		for (int k = 0; k < F_PER; ++k) { uint32_t e = k * F_THREADS + tid; r[k] = S.rec[(uint64_t)s0 + (e < m ? e : m - 1)]; }
	}
	// the key field sits in the high dword (from bit hks up), the predecessor symbol in the 3 bits below it
	const int binbits = fbits < F_BINBITS ? fbits : F_BINBITS;
	const int lowbits = fbits - binbits;                      // <= 16
	const int hks = rec_key_shift(O.pb) - 32, hps = O.pb - 32;      // key field and predecessor symbol inside the high dword
	const int bsh = hks + lowbits;
	const uint32_t bmask = (1u << binbits) - 1, lmask = (1u << lowbits) - 1;
	const uint32_t apad = (uint32_t)((reinterpret_cast<uintptr_t>(O.bwt) + s0) & 3u);   // obwt keeps the byte phase of the global address
	#pragma unroll
	for (int k = 0; k < (F_BINS / 2) / F_THREADS; ++k) cw[k * F_THREADS + tid] = 0;
	__syncthreads();
	// ---- count: slot of every record inside its bin.  No branch around an LDS access (a returning
	// atomic or a load inside its own if-block is waited for on the spot: 12 serial round trips per
	// phase): slots past the bucket's end add 0 to the spare word behind the table ----
	uint32_t bs[F_PER];                                       // bin (F_BINS = none), then slot | start of the bin << 16
	{
		uint32_t old[F_PER];
		#pragma unroll
		for (int k = 0; k < F_PER; ++k) {
			const uint32_t e = k * F_THREADS + tid;
			const uint32_t bin = e < m ? (((uint32_t)(r[k] >> 32) >> bsh) & bmask) : (uint32_t)F_BINS;
			bs[k] = bin;
			LDS_CHK(bin >> 1, CW_N);
			old[k] = atomicAdd(&cw[bin >> 1], e < m ? ((bin & 1u) ? 0x10000u : 1u) : 0u);
		}
		#pragma unroll
		for (int k = 0; k < F_PER; ++k) bs[k] = bs[k] | (((bs[k] & 1u) ? old[k] >> 16 : old[k] & 0xFFFFu) << 16);   // bin | slot << 16
	}
	__syncthreads();
	// ---- exclusive scan over the bins, in place: thread = 16 consecutive bins = 8 words ----
	uint32_t cmax = 0;                                        // largest of this thread's bins
	{
		constexpr int WPT = (F_BINS / 2) / F_THREADS;       // 8
		uint32_t w[WPT]; uint32_t sum = 0;
		#pragma unroll
		for (int k = 0; k < WPT; ++k) {
			w[k] = cw[tid * WPT + k];
			uint32_t lo = w[k] & 0xFFFFu, hi = w[k] >> 16;
			sum += lo + hi; cmax = cmax > lo ? cmax : lo; cmax = cmax > hi ? cmax : hi;
		}
		uint32_t inc = wave_incl_sum(sum);
		if (lane == 63) wsum[wave] = inc;
		__syncthreads();
		uint32_t run = inc - sum;
		for (uint32_t i = 0; i < wave; ++i) run += wsum[i];
		#pragma unroll
		for (int k = 0; k < WPT; ++k) {
			uint32_t lo = w[k] & 0xFFFFu, hi = w[k] >> 16;
			cw[tid * WPT + k] = run | ((run + lo) << 16);
			run += lo + hi;
		}
		if (tid == F_THREADS - 1) cw[F_BINS / 2] = run;      // = m: end of the last bin
	}
	__syncthreads();
	const uint16_t *cs = reinterpret_cast<const uint16_t *>(cw);       // start of bin b = cs[b], its end = cs[b + 1]
	// a bin above LS_TIE_CAP (a 16-mer repeated in the text) goes to the overflow list on its own
	if (cmax > (uint32_t)LS_TIE_CAP) {
		#pragma unroll 1
		for (uint32_t bb = tid * (F_BINS / F_THREADS); bb < (tid + 1) * (F_BINS / F_THREADS); ++bb) {
			const uint32_t st = cs[bb], c = (uint32_t)cs[bb + 1] - st;
			if (c > (uint32_t)LS_TIE_CAP) ovf_push(O, s0 + st, c, true);
		}
	}
	uint32_t stv[F_PER], env[F_PER];                          // start and end of the record's bin (spare bin: m, 0)
	#pragma unroll
	for (int k = 0; k < F_PER; ++k) { const uint32_t bin = bs[k] & 0xFFFFu; LDS_CHK(bin + 1, 2 * CW_N); stv[k] = cs[bin]; env[k] = cs[bin + 1]; }
	#pragma unroll
	for (int k = 0; k < F_PER; ++k) {
		const uint32_t e = k * F_THREADS + tid;
		if (e < m) { LDS_CHK(stv[k] + (bs[k] >> 16), ENT_N); ent[stv[k] + (bs[k] >> 16)] = ((((uint32_t)(r[k] >> 32) >> hks) & lmask) << IDXB) | e; }      // a store: nothing waits for it
	}
	__syncthreads();
	// ---- the owner of a record scans its bin: final place, ties.  Three counts over the bin's
	// entries y (own entry x, xk = x without its index bits): y < x, y < xk (smaller key) and
	// y < xk + 2^13 (key not larger) give the place, "an equal key precedes me" and "my key ties".
	// dest: place | hot bin << 29 | tied << 30 | has an equal predecessor << 31
	uint32_t dest[F_PER];
	#pragma unroll
	for (int h = 0; h < F_PER; h += 6) {                       // six records at a time: their 24 entry loads fly together
		uint32_t y[6][4];
		#pragma unroll
		for (int k = 0; k < 6; ++k) {
			const uint32_t st = (h + k) * F_THREADS + tid < m ? stv[h + k] : 0u;     // beyond the bin: the next entries (ent is padded), masked out below
			LDS_CHK(st + 3, ENT_N);
			#pragma unroll
			for (int j = 0; j < 4; ++j) y[k][j] = ent[st + j];
		}
		#pragma unroll
		for (int k = 0; k < 6; ++k) {
			const int kk = h + k;
			const uint32_t e = kk * F_THREADS + tid;
			const uint32_t st = stv[kk], c = env[kk] - st, slot = bs[kk] >> 16;
			const uint32_t xk = (((uint32_t)(r[kk] >> 32) >> hks) & lmask) << IDXB, x = xk | e, xn = xk + (1u << IDXB);
			uint32_t lt = 0, ltk = 0, lek = 0;
			#pragma unroll
			for (uint32_t j = 0; j < 4; ++j) {
				const uint32_t v = j < c ? y[k][j] : 0xFFFFFFFFu;
				lt += v < x ? 1u : 0u; ltk += v < xk ? 1u : 0u; lek += v < xn ? 1u : 0u;
			}
			if (e < m && c > 4 && c <= (uint32_t)LS_TIE_CAP) {     // rare: longer bins with a loop
				for (uint32_t j = 4; j < c; ++j) {
					LDS_CHK(st + j, ENT_N);
					const uint32_t v = ent[st + j];
					lt += v < x ? 1u : 0u; ltk += v < xk ? 1u : 0u; lek += v < xn ? 1u : 0u;
				}
			}
			dest[kk] = c > (uint32_t)LS_TIE_CAP ? (st + slot) | (1u << 29)       // hot bin: arrival order, positions into sa
			                                    : (st + lt) | (lek - ltk > 1 ? 1u << 30 : 0u) | (lt > ltk ? 1u << 31 : 0u);
		}
	}
	// ---- drop BWT byte (and SA entry) at the final place; key-field ties are marked for k_tie_fix ----
	#pragma unroll
	for (int k = 0; k < F_PER; ++k) {
		uint32_t e = k * F_THREADS + tid;
		if (e < m) {
			const uint32_t dp = dest[k] & 0xFFFFu, g = s0 + dp;
			const uint64_t i = r[k] & ((1ull << O.pb) - 1);
			LDS_CHK(apad + dp, OBWT_N); LDS_CHK(dp, m);
			obwt[apad + dp] = (uint8_t)(((uint32_t)(r[k] >> 32) >> hps) & 7u);     // hot bins: overwritten by the fallback's emit
			if (WRITE_SA) { osa[dp] = (uint32_t)i; if (O.sa.hi) O.sa.hi[g] = (uint8_t)(i >> 32); }
			else if (dest[k] & (3u << 29)) sa_set(O.sa, g, i);       // without consumers only k_tie_fix / the fallback read positions
			if (dest[k] & (1u << 31)) atomicOr(&O.tiebits[g >> 5], 1u << (g & 31));
			if (i == O.start) *O.p0 = g;
			if (!(dest[k] & (3u << 29)) && sample_hit(O.smp, (uint32_t)i)) sample_put(O.smp, g, i);     // (tied / hot records: where they are finally placed)
		}
	}
	__syncthreads();
	// ---- the bucket leaves as rows: BWT bytes as dwords (partial first / last dword bytewise) ----
	{
		uint8_t *gb = O.bwt + ((ptrdiff_t)s0 - (ptrdiff_t)apad);   // 4-byte aligned (may point before the block's first byte: never written)
		const uint32_t lo = apad, hi = apad + m;               // valid bytes of obwt
		const uint32_t ndw = (hi + 3) >> 2;
		for (uint32_t j = tid; j < ndw; j += F_THREADS) {
			LDS_CHK(4 * j + 3, OBWT_N);
			if (4 * j >= lo && 4 * j + 4 <= hi) reinterpret_cast<uint32_t *>(gb)[j] = reinterpret_cast<const uint32_t *>(obwt)[j];
			else { for (uint32_t q = 4 * j; q < 4 * j + 4; ++q) if (q >= lo && q < hi) gb[q] = obwt[q]; }
		}
	}
	if (WRITE_SA) {
		#pragma unroll
		for (int k = 0; k < F_PER; ++k) { uint32_t e = k * F_THREADS + tid; if (e < m) O.sa.lo[s0 + e] = osa[e]; }
	}
}

template <bool WRITE_SA, int THREADS>
__global__ __launch_bounds__(THREADS, 4) void k_finish_lds(FinishSrc S, int fbits, uint32_t b_first, const uint32_t *todo, uint32_t todo_cap, const FinishLists *L, SortOut O)
{
	const RowBucket RB{0u, 0u, 0u}; const RowAhead RA{0u, 0u, 0u};
	if (!todo) {
		const uint32_t b = b_first + blockIdx.x;
		GLB_CHK(b < S.nbF);
		const uint32_t s0 = S.bstart[b], m = bucket_end(S, b) - s0;
		GLB_CHK(s0 <= S.nrec && m <= S.nrec - s0);
		finish_bucket<WRITE_SA, THREADS, false>(b, s0, m, RB, RA, S, fbits, L, O); return;
	}
	// listed buckets: a fixed grid strides over the list (an empty list costs nothing).  The count is what the listing kernels asked for;
	// what the list HOLDS is at most its capacity (the round-2 abort of `d1.log`: 126 buckets asked for 102 slots, this loop read the
	// 24 ids behind the list -- run-table words -- and S.bstart[] of those faulted: profiles/NOTES_r03.md)
	uint32_t cnt = todo[0]; cnt = cnt < todo_cap ? cnt : todo_cap;
	for (uint32_t it = blockIdx.x; it < cnt; it += gridDim.x) {
		const uint32_t b = todo[1 + it];
		GLB_CHK(b < S.nbF);
		const uint32_t s0 = S.bstart[b], m = bucket_end(S, b) - s0;
		GLB_CHK(s0 <= S.nrec && m <= S.nrec - s0);
		finish_bucket<WRITE_SA, THREADS, false>(b, s0, m, RB, RA, S, fbits, L, O);
		__syncthreads();
	}
}

// The gathering finish: one workgroup per final bucket, consecutive workgroups = consecutive digits of one level-0 bucket
// (they read neighbouring runs of the same chunks at about the same time).  Bounds, level-0 bucket entry and the thread's
// words of the run-table row are all requested before anything waits.
template <bool WRITE_SA, int THREADS>
__global__ __launch_bounds__(THREADS, 4) void k_finish_rows(FinishSrc S, int fbits, uint32_t b_first, const FinishLists *L, SortOut O)
{
	const uint32_t tid = threadIdx.x, b = b_first + blockIdx.x;
	const uint32_t b0 = b >> S.D1;
	const RowBucket RB{S.btab[2 * S.nb0 + b0], S.btab[S.nb0 + b0], S.btab[b0]};
	GLB_CHK(b < S.nbF && b0 < S.nb0);
	const uint32_t s0 = S.bstart[b], s1 = bucket_end(S, b);
	GLB_CHK(s0 <= s1 && s1 <= S.nrec);
	if (s1 == s0) return;                                      // empty (a suffix-range shard keeps a digit range; nch may be 0)
	const uint32_t tc = tid < RB.nch ? tid : RB.nch - 1, tn = tc + 1 < RB.nch ? tc + 1 : tc;
	const uint32_t *row = S.runpre + RB.tab_off + (uint64_t)(b & ((1u << S.D1) - 1u)) * RB.nch;
	const RowAhead RA{s1, row[tc], row[tn]};
	finish_bucket<WRITE_SA, THREADS, true>(b, s0, s1 - s0, RB, RA, S, fbits, L, O);
}

// ------------------------------------------------------------------------------------------------
// THE FAST FINISH (gathering, 512 threads, what C3 runs): the same counting sort as finish_bucket with the per-record
// instruction count cut (the kernel is instruction-issue bound, not HBM bound):
//   * one 32-bit table word per bin (count, then start | count << 16): one LDS atomic and one LDS read per record;
//   * an entry is (free key bits << 13 | index) cut to 32 bits: above the low key bits it keeps the low TAGB = 19 - lowbits bits
//     of its bin.  Relative to the first value of the record's own bin, modulo 2^32, the entries of that bin are below
//     2^(13 + lowbits) and those of other bins are not -- unless a bin is an exact multiple of 2^TAGB bins further on, so the
//     four entries a record reads from its bin's start need no "is this still my bin" mask.  Such a bin within three
//     entries means at most two entries in 2^TAGB - 1 bins, hence an aligned block of 2^(TAGB-2) empty bins: every wave
//     checks its bins for such blocks and a bucket that has one (tiny, or skewed) is listed for the copy-together path;
//   * items whose 512 slots all lie inside the bucket run without per-lane guards;
//   * the run-to-slot map is two 16-byte broadcast reads per 64 slots.
// Requires TAGB >= 7, i.e. at most 12 low key bits (level 1 of 4 bits or more); other geometries keep finish_bucket.
// ------------------------------------------------------------------------------------------------
template <bool WRITE_SA>
__global__ __launch_bounds__(512, 4) void k_finish_fast(FinishSrc S, int fbits, uint32_t b_first, const FinishLists *Lp, SortOut O)
{
	constexpr int F_THREADS = 512, F_CAP = 6144, F_PER = 12, F_WAVES = 8, IDXB = 13;
	constexpr int ENT_N = F_CAP + 8, TAB_N = F_BINS + 4, OBWT_N = F_CAP + 8, GF_N = F_CAP / 64 + 2;
	__shared__ __attribute__((aligned(16))) uint32_t ent[ENT_N];
	__shared__ __attribute__((aligned(16))) uint32_t tab[TAB_N];
	__shared__ __attribute__((aligned(16))) uint8_t obwt[OBWT_N];
	__shared__ uint32_t wsum[F_WAVES];
	__shared__ __attribute__((aligned(8))) uint2 rr[G_NCH + 1];
	__shared__ uint16_t gfirst[GF_N];
	__shared__ __attribute__((aligned(16))) uint4 gda[GF_N], gdb[GF_N];      // {o1, o2, o3, A} and {B - o1, C - o2, D - o3, -}
	__shared__ uint32_t s_slow;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, b = b_first + (S.xcd ? xcd_order(blockIdx.x, gridDim.x) : blockIdx.x);
	const uint32_t b0 = b >> S.D1;
	const uint32_t nch = S.btab[2 * S.nb0 + b0], tab_off = S.btab[S.nb0 + b0], bstart0 = S.btab[b0];
	const uint32_t s0 = S.bstart[b], m = bucket_end(S, b) - s0;
	GLB_CHK(b < S.nbF && b0 < S.nb0 && s0 <= S.nrec && m <= S.nrec - s0);
	if (m == 0) return;
	if (m > (uint32_t)F_CAP || nch > (uint32_t)G_NCH) {
		if (tid == 0) {
			const FinishLists L = *Lp;
			if (m <= 2u * F_CAP) { uint32_t k = atomicAdd(&L.midlist[0], 1u); if (k < L.mid_cap) L.midlist[1 + k] = b; }
			else if (m > L.huge_min) { uint32_t k = atomicAdd(&L.hugelist[0], 1u); if (k < L.huge_cap) L.hugelist[1 + k] = b; ovf_push(O, s0, m); }
			else { uint32_t k = atomicAdd(&L.biglist[0], 1u); if (k < L.big_cap) L.biglist[1 + k] = b; }
		}
		return;
	}
	// ---- slots -> runs (see finish_bucket) ----
	const uint32_t glast = (m - 1) >> 6;
	{
		const uint32_t *row = S.runpre + tab_off + (uint64_t)(b & ((1u << S.D1) - 1u)) * nch;
		const uint32_t tc = tid < nch ? tid : nch - 1, tn = tc + 1 < nch ? tc + 1 : tc;
		const uint32_t v = row[tc], vn = tid + 1 < nch ? row[tn] : (m << 14);
		const uint32_t pre = v >> 14, end = vn >> 14;
		if (tid < nch) {
			rr[tid] = make_uint2(pre, bstart0 + tid * S.chunk + (v & 0x3FFFu));
			for (uint32_t g = (pre + 63) >> 6; (g << 6) < end; ++g) gfirst[g] = (uint16_t)tid;
		}
		if (tid == 0) { rr[nch] = make_uint2(m, 0u); gfirst[glast + 1] = (uint16_t)(nch - 1); s_slow = 0u; }
		#pragma unroll
		for (int k = 0; k < F_BINS / F_THREADS; ++k) tab[k * F_THREADS + tid] = 0;
		if (tid < 8) ent[m + tid] = 0xFFFFFFFFu;                // what a record reads past the last entry compares greater
	}
	__syncthreads();
	if (tid <= glast) {
		const uint32_t sl = tid << 6;
		const uint32_t j0 = gfirst[tid], j1 = gfirst[tid + 1];
		const uint2 q0 = rr[j0];
		uint32_t o1 = 64u, o2 = 64u, o3 = 64u, b1 = 0u, b2 = 0u, b3 = 0u, found = 0;
		for (uint32_t i = j0 + 1; i <= j1 && i < nch; ++i) {
			const uint2 qi = rr[i];
			if (rr[i + 1].x == qi.x || qi.x >= sl + 64u) continue;
			const uint32_t oo = qi.x - sl, bb = qi.y - oo;
			if (found == 0) { o1 = oo; b1 = bb; } else if (found == 1) { o2 = oo; b2 = bb; } else if (found == 2) { o3 = oo; b3 = bb; }
			++found;
		}
		if (found > 3) s_slow = 1u;
		gda[tid] = make_uint4(o1, o2, o3, q0.y + (sl - q0.x));
		gdb[tid] = make_uint4(b1, b2, b3, 0u);
	}
	__syncthreads();
	if (s_slow) {
		if (tid == 0) { const FinishLists L = *Lp; uint32_t k = atomicAdd(&L.midlist[0], 1u); if (k < L.mid_cap) L.midlist[1 + k] = b; }
		return;
	}
	const uint32_t kfull = m / F_THREADS;                        // items 0 .. kfull-1 have all their 512 slots inside the bucket
	uint64_t r[F_PER];
	{
		uint32_t adr[F_PER];
		#pragma unroll
		for (int k = 0; k < F_PER; ++k) {
			const uint32_t eu = k * F_THREADS + tid;
			const uint32_t e = eu < m ? eu : m - 1;
			uint32_t g = (uint32_t)(k * F_WAVES) + wave; g = g < glast ? g : glast;
			const uint4 da = gda[g], db = gdb[g];
			const uint32_t x = e - (g << 6);
			adr[k] = (x >= da.z ? db.z : x >= da.y ? db.y : x >= da.x ? db.x : da.w) + x;
		}
		#pragma unroll
		for (int k = 0; k < F_PER; ++k) r[k] = S.rec[adr[k]];
	}
	const int binbits = F_BINBITS, lowbits = fbits - binbits;
	const int hks = rec_key_shift(O.pb) - 32, hps = O.pb - 32;
	const int bsh = hks + lowbits;
	const int tagb = 32 - IDXB - lowbits;                        // bin bits an entry keeps (7..11)
	const uint32_t bsl = (uint32_t)(IDXB + lowbits);
	const uint32_t apad = (uint32_t)((reinterpret_cast<uintptr_t>(O.bwt) + s0) & 3u);
	// ---- count.  Items below kfull run without per-lane guards (the branch on k < kfull is scalar) ----
	uint32_t slot[F_PER];
	#pragma unroll
	for (int k = 0; k < F_PER; ++k) {
		const uint32_t bin = ((uint32_t)(r[k] >> 32) >> bsh) & (uint32_t)(F_BINS - 1);
		if ((uint32_t)k < kfull) slot[k] = atomicAdd(&tab[bin], 1u);
		else { slot[k] = 0; if (k * F_THREADS + tid < m) slot[k] = atomicAdd(&tab[bin], 1u); }
	}
	__syncthreads();
	// ---- scan: thread = 16 consecutive bins; tab[bin] = start | count << 16 ----
	{
		constexpr int WPT = F_BINS / F_THREADS;                  // 16
		uint32_t w[WPT]; uint32_t sum = 0, cmax = 0;
		#pragma unroll
		for (int q = 0; q < WPT / 4; ++q) {
			const uint4 v = reinterpret_cast<const uint4 *>(tab)[tid * (WPT / 4) + q];
			w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
		}
		#pragma unroll
		for (int k = 0; k < WPT; ++k) { sum += w[k]; cmax = cmax > w[k] ? cmax : w[k]; }
		{
			// an aligned block of 2^(tagb-2) empty bins = G = 2^(tagb-6) lanes (16 bins each) with nothing: bin tags could alias (see the header)
			unsigned long long g = __ballot(sum != 0);
			const int G = 1 << ((tagb < 11 ? tagb : 11) - 6);      // (11 bin bits and more: a half wave's 512 bins; with all 13 kept nothing can alias)
			for (int sft = 1; sft < G; sft <<= 1) g |= g >> sft;
			const unsigned long long gm = G == 2 ? 0x5555555555555555ull : G == 4 ? 0x1111111111111111ull : G == 8 ? 0x0101010101010101ull
			                            : G == 16 ? 0x0001000100010001ull : 0x0000000100000001ull;
			if ((g & gm) != gm && lane == 0) s_slow = 1u;
		}
		uint32_t inc = wave_incl_sum(sum);
		if (lane == 63) wsum[wave] = inc;
		__syncthreads();
		uint32_t run = inc - sum;
		for (uint32_t i = 0; i < wave; ++i) run += wsum[i];
		#pragma unroll
		for (int k = 0; k < WPT; ++k) { const uint32_t c = w[k]; w[k] = run | (c << 16); run += c; }
		#pragma unroll
		for (int q = 0; q < WPT / 4; ++q) reinterpret_cast<uint4 *>(tab)[tid * (WPT / 4) + q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
		__syncthreads();
		if (s_slow) {
			if (tid == 0) { const FinishLists L = *Lp; uint32_t k = atomicAdd(&L.midlist[0], 1u); if (k < L.mid_cap) L.midlist[1 + k] = b; }
			return;
		}
		// a 16-mer repeated inside the bucket: that bin goes to the overflow list on its own -- only once the bucket is known to stay
		// here (a bucket handed to the slower path lists its bins there: pushed from both places, the list would hold the slots twice)
		if (cmax > (uint32_t)LS_TIE_CAP) {
			#pragma unroll 1
			for (int k = 0; k < WPT; ++k) { const uint32_t c = w[k] >> 16; if (c > (uint32_t)LS_TIE_CAP) ovf_push(O, s0 + (w[k] & 0xFFFFu), c, true); }
		}
	}
	// ---- entries into bin order ----
	uint32_t tb[F_PER];                                           // start | count << 16 of the record's bin
	#pragma unroll
	for (int k = 0; k < F_PER; ++k) tb[k] = tab[((uint32_t)(r[k] >> 32) >> bsh) & (uint32_t)(F_BINS - 1)];
	#pragma unroll
	for (int k = 0; k < F_PER; ++k) {
		const uint32_t hi = (uint32_t)(r[k] >> 32), e = k * F_THREADS + tid;
		const uint32_t x = ((hi >> hks) << IDXB) | e;             // cut to 32 bits: tagb bin bits | low key bits | index
		if ((uint32_t)k < kfull) ent[(tb[k] & 0xFFFFu) + slot[k]] = x;
		else if (e < m) ent[(tb[k] & 0xFFFFu) + slot[k]] = x;
	}
	__syncthreads();
	// ---- rank inside the bin from the four entries at its start; ties of the whole key field.  Relative to xk0 (the record's
	// entry without its index) modulo 2^32: t < 8192 <=> same key, t < index <=> same key and an earlier record; relative to
	// the bin's first value B: below the own offset <=> sorts before me.  Everything rare (bins above four entries, ties, the
	// block's first suffix) sits behind ONE branch per record. ----
	const uint32_t start_lo = (uint32_t)O.start;
	uint32_t dpv[WRITE_SA ? F_PER : 1];                          // WRITE_SA: the record's place; its position goes through `ent` once every entry has been read
	auto rank_drop = [&](const int k, const bool guard) {
		const uint32_t hi = (uint32_t)(r[k] >> 32), e = k * F_THREADS + tid;
		const bool valid = !guard || e < m;
		const uint32_t st = valid ? (tb[k] & 0xFFFFu) : 0u, c = tb[k] >> 16;
		const uint32_t B = (hi >> bsh) << bsl;                    // first value of the record's bin
		const uint32_t xk0 = (hi >> hks) << IDXB, dx = (xk0 | e) - B;
		const uint32_t y0 = ent[st], y1 = ent[st + 1], y2 = ent[st + 2], y3 = ent[st + 3];
		uint32_t lt = (y0 - B < dx ? 1u : 0u) + (y1 - B < dx ? 1u : 0u) + (y2 - B < dx ? 1u : 0u) + (y3 - B < dx ? 1u : 0u);
		const uint32_t t0 = y0 - xk0, t1 = y1 - xk0, t2 = y2 - xk0, t3 = y3 - xk0;
		// (same key <=> t = the other entry's index < F_CAP.  Not "< 8192": the 0xFFFFFFFF padding behind the bucket's last entry reads as
		// index 8191 of the all-ones key, and a record with that key then looked tied -- harmless until the samples: a "tied" record leaves
		// its sample to the tie fix, which never saw a group there; 14 of 7.3 M samples at 30 Gbase, found by the C5 test's slot census)
		constexpr uint32_t SAMEKEY = (uint32_t)F_CAP;
		uint32_t eq = (t0 < SAMEKEY ? 1u : 0u) + (t1 < SAMEKEY ? 1u : 0u) + (t2 < SAMEKEY ? 1u : 0u) + (t3 < SAMEKEY ? 1u : 0u);
		const uint32_t lo32 = (uint32_t)r[k];
		uint32_t dp = st + lt;
		const bool smp = sample_hit(O.smp, lo32);
		if (valid && (eq > 1 || c > 4 || lo32 == start_lo || smp)) {    // rare
			bool prev = t0 < e || t1 < e || t2 < e || t3 < e;
			const bool hot = c > (uint32_t)LS_TIE_CAP;
			if (c > 4 && !hot) {
				#pragma unroll 1
				for (uint32_t j = 4; j < c; ++j) {
					const uint32_t y = ent[st + j], t = y - xk0;
					lt += y - B < dx ? 1u : 0u; eq += t < SAMEKEY ? 1u : 0u; prev = prev || t < e;
				}
			}
			dp = hot ? st + slot[k] : st + lt;
			const uint32_t g = s0 + dp;
			const uint64_t pos = r[k] & ((1ull << O.pb) - 1);
			if (!WRITE_SA && (hot || eq > 1)) sa_set(O.sa, g, pos);       // only k_tie_fix / the fallback read positions
			if (prev && !hot) atomicOr(&O.tiebits[g >> 5], 1u << (g & 31));
			if (pos == O.start) *O.p0 = g;
			if (smp && !hot && eq <= 1) sample_put(O.smp, g, pos);            // (rank,pos) sample of a record alone in its key: its place is final
		}
		if (WRITE_SA) dpv[k] = valid ? dp : 0xFFFFFFFFu;
		if (valid) {
			obwt[apad + dp] = (uint8_t)((hi >> hps) & 7u);       // hot bins: overwritten by the fallback's emit
			if (WRITE_SA && O.sa.hi) O.sa.hi[s0 + dp] = (uint8_t)(hi & ((1u << hps) - 1u));
		}
	};
	#pragma unroll
	for (int k = 0; k < F_PER; ++k) { if ((uint32_t)k < kfull) rank_drop(k, false); else rank_drop(k, true); }
	__syncthreads();
	{
		uint8_t *gb = O.bwt + ((ptrdiff_t)s0 - (ptrdiff_t)apad);
		const uint32_t lo = apad, hi = apad + m;
		const uint32_t ndw = (hi + 3) >> 2;
		for (uint32_t j = tid; j < ndw; j += F_THREADS) {
			if (4 * j >= lo && 4 * j + 4 <= hi) reinterpret_cast<uint32_t *>(gb)[j] = reinterpret_cast<const uint32_t *>(obwt)[j];
			else { for (uint32_t q = 4 * j; q < 4 * j + 4; ++q) if (q >= lo && q < hi) gb[q] = obwt[q]; }
		}
	}
	if (WRITE_SA) {
		// the positions in place order through the entry array (dead now: an extra 24 KB of LDS for them halved the occupancy)
		#pragma unroll
		for (int k = 0; k < F_PER; ++k) if (dpv[k] != 0xFFFFFFFFu) ent[dpv[k]] = (uint32_t)r[k];
		__syncthreads();
		#pragma unroll
		for (int k = 0; k < F_PER; ++k) { uint32_t e = k * F_THREADS + tid; if (e < m) O.sa.lo[s0 + e] = ent[e]; }
	}
}

// ------------------------------------------------------------------------------------------------
// THE FINISH, third form (sorts that keep no suffix array: what the shipped one-block plan runs).  The same counting sort as
// k_finish_fast, but no record computes its rank:
//   * the owner of a record drops a 4-byte ENTRY (low key bits | index | "special" | predecessor symbol) at its bin's start + arrival slot;
//   * the bins that hold two entries or more (15 % of the bins) go on a work list in LDS and are put in order, in place, by ALL threads,
//     three or four bins each, their table words and entries in flight together, up to four (eight) entries through a sorting network in
//     registers: no chain of dependent LDS round trips, no imbalance between the threads that scanned many such bins and few;
//   * key ties (0.3 % of the records) are marked in the tie bitmap; the positions k_tie_fix needs of them are fetched again, by one
//     thread per tied entry, while the bucket leaves, and stored at the very end;
//   * the bucket leaves straight from the entry array: four entries = four BWT bytes = one dword store.  A record whose position is a
//     multiple of the sampling rate, or the block's first suffix, leaves its position in a short list; the thread that sees its (special)
//     entry at its final place writes the (rank,pos) sample / p0.
// Against k_finish_fast: no reads of the bin's entries by every record (4 of its ~15 LDS operations per record), no byte staging array,
// no bin tags (so no aliasing rule, and sparse buckets stay here instead of being listed), 16-bit table words (start | count capped at 7),
// a third of the vector instructions; 51 KB of LDS and 80 registers: three workgroups per CU instead of two.
// Dense sampling (more specials than the list holds) still works, through fetches of the records, but k_finish_fast is the kernel for it.
// ------------------------------------------------------------------------------------------------
constexpr int V3_NCH = 256;              // runs a bucket may be spread over (more: listed)
constexpr int V3_WL = 1536;              // work list: bins with two entries or more (uniform input: 1280 +- 33 of a full bucket's 8192)
constexpr int V3_TL = 256;               // tied entries whose records are fetched again
constexpr int V3_BL = 128;               // bins of 9 .. LS_TIE_CAP entries: ordered by half a wave each (skewed text; a uniform bucket has none)
constexpr int V3_SP = 64;                // positions of special records
constexpr uint64_t V3_MIN_RATE = 256;    // sampling rates below this take k_finish_fast (F_CAP / rate specials per bucket must fit V3_SP)

#define B3M_CE(a, b) { const uint32_t lo_ = a < b ? a : b, hi_ = a < b ? b : a; a = lo_; b = hi_; }

// THREADS = 512 (12 records per thread, 80 registers: three workgroups = 24 waves per CU) or 1024 (6 records per thread: at 64 registers two
// workgroups = 32 waves per CU, all the CU holds; B3M_FINISH_V3=4)
// NOMAP (timing only, WRONG results: B3M_FINISH_V3=5): the bucket's records are read from a contiguous range instead of through the slot -> run map: what the map's two
// phases and the dependent loads in front of the records cost
template <int WPC, int THREADS = 512, bool NOMAP = false>
__global__ __launch_bounds__(THREADS, THREADS == 512 ? 2 * WPC : 8) void k_finish_v3(FinishSrc S, int fbits, uint32_t b_first, const FinishLists *Lp, SortOut O)
{
	constexpr int F_THREADS = THREADS, F_CAP = 6144, F_PER = F_CAP / THREADS, F_WAVES = THREADS / 64;
	constexpr int NQ = THREADS == 512 ? 4 : (V3_WL + THREADS - 1) / THREADS;      // work-list items per thread
	static_assert(F_CAP % THREADS == 0 && NQ * THREADS >= V3_WL && (F_BINS / 2) / THREADS >= 4, "k_finish_v3: thread count");
	constexpr int ENT_N = F_CAP + 16, TABW_N = F_BINS / 2 + 4, GF_N = F_CAP / 64 + 2;
	constexpr int E_IDX = 4, E_LOW = 17;                              // entry = low key bits << 17 | index << 4 | special << 3 | predecessor symbol
	constexpr uint32_t E_SPECIAL = 8u, T_START = 0x1FFFu;             // table half-word after the scan: start (13 bits) | min(count, 7) << 13
	constexpr uint32_t E_PEND = 1u << 31;                             // (set once the entry is in place) a tied record whose position is still to be fetched: no room on the tie list
	__shared__ __attribute__((aligned(16))) uint32_t ent[ENT_N];      // entry of the record at place p of the bucket: ent[apad + p] (apad: byte phase of the BWT address)
	__shared__ __attribute__((aligned(16))) uint32_t tabw[TABW_N];    // two 16-bit words per dword: counts, then start | capped count; half-word F_BINS = m
	__shared__ uint32_t wsum[F_WAVES];
	__shared__ __attribute__((aligned(8))) uint2 rr[V3_NCH + 1];
	__shared__ uint16_t gfirst[GF_N];
	__shared__ __attribute__((aligned(16))) uint4 gda[GF_N], gdb[GF_N];
	__shared__ uint16_t wl[V3_WL];                                    // bins with two entries or more
	__shared__ uint32_t tl[V3_TL];                                    // tied entries: place << 13 | index
	__shared__ __attribute__((aligned(8))) uint2 sp[V3_SP];           // special records: {position bits 0..31, index | position bits 32.. << 16}
	__shared__ uint16_t bl[V3_BL];                                    // bins above eight entries
	__shared__ uint32_t s_slow, s_wl, s_tl, s_sp, s_bl;
	__shared__ uint32_t lds_pad[WPC == 2 ? 6144 : 1];                 // (WPC = 2, the A/B form: enough LDS to keep a third workgroup off the CU)
#ifdef B3M_PHASE_EXPERIMENT
	const uint32_t xf = S.xcd >> 8;             // (`make experiment`, timing only: B3M_FINISH_EXPERIMENT = leave after the records are loaded (1), the count (2), the scan (3), the entries are placed (4), the listed bins (5))
#else
	constexpr uint32_t xf = 0u;                  // (the exits below cost the kernel 0.15 ms when they are compiled in: 11.31 vs 11.16 ms)
#endif
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, b = b_first + ((S.xcd & 1u) ? xcd_order(blockIdx.x, gridDim.x) : blockIdx.x);
	const uint32_t b0 = b >> S.D1;
	const uint32_t nch = S.btab[2 * S.nb0 + b0], tab_off = S.btab[S.nb0 + b0], bstart0 = S.btab[b0];
	const uint32_t s0 = S.bstart[b], m = bucket_end(S, b) - s0;
	GLB_CHK(b < S.nbF && b0 < S.nb0 && s0 <= S.nrec && m <= S.nrec - s0);
	if (m == 0) return;
	if (m > (uint32_t)F_CAP || nch > (uint32_t)V3_NCH) {
		if (tid == 0) {
			const FinishLists L = *Lp;
			if (m <= 2u * F_CAP) { uint32_t k = atomicAdd(&L.midlist[0], 1u); if (k < L.mid_cap) L.midlist[1 + k] = b; }
			else if (m > L.huge_min) { uint32_t k = atomicAdd(&L.hugelist[0], 1u); if (k < L.huge_cap) L.hugelist[1 + k] = b; ovf_push(O, s0, m); }
			else { uint32_t k = atomicAdd(&L.biglist[0], 1u); if (k < L.big_cap) L.biglist[1 + k] = b; }
		}
		return;
	}
	if (WPC == 2 && fbits < 0) lds_pad[tid] = 0;                 // (never: keeps the padding array alive)
	const uint32_t apad = (uint32_t)((reinterpret_cast<uintptr_t>(O.bwt) + s0) & 3u);
	// ---- slots -> runs (see finish_bucket).  The map a thread needs for its OWN records lives in registers (GDesc: lane k of a wave builds the descriptor of the
	// 64-slot group of the wave's item k, every lane fetches it with v_readlane): the records are requested before the barrier that publishes the LDS copy of the
	// map, which only the rare re-fetches use (tied records, dense samples) ----
	const uint32_t glast = (m - 1) >> 6, xlast = (m - 1) & 63u;
	GDesc D{64u | (64u << 8) | (64u << 16), 0u, 0u, 0u, 0u};
	if (NOMAP) {
		if (tid <= glast) { gda[tid] = make_uint4(64u, 64u, 64u, s0 + (tid << 6)); gdb[tid] = make_uint4(0u, 0u, 0u, 0u); }
		{ const uint32_t g = lane * F_WAVES + wave; D.A = s0 + ((g < glast ? g : glast) << 6); }
		if (tid == 0) { s_slow = 0u; s_tl = 0u; s_sp = 0u; s_bl = 0u; }
		#pragma unroll
		for (int k = 0; k < (F_BINS / 2) / F_THREADS; ++k) tabw[k * F_THREADS + tid] = 0;
		if (tid < 4) { ent[tid] = 0u; tabw[F_BINS / 2 + tid] = 0u; }
		if (tid < 12) ent[apad + m + tid] = 0u;
	} else {
		{
			const uint32_t *row = S.runpre + tab_off + (uint64_t)(b & ((1u << S.D1) - 1u)) * nch;
			const uint32_t tc = tid < nch ? tid : nch - 1, tn = tc + 1 < nch ? tc + 1 : tc;
			const uint32_t v = row[tc], vn = tid + 1 < nch ? row[tn] : (m << 14);
			const uint32_t pre = v >> 14, end = vn >> 14;
			if (tid < nch) {
				rr[tid] = make_uint2(pre, bstart0 + tid * S.chunk + (v & 0x3FFFu));
				for (uint32_t g = (pre + 63) >> 6; (g << 6) < end; ++g) gfirst[g] = (uint16_t)tid;
			}
			if (tid == 0) { rr[nch] = make_uint2(m, 0u); gfirst[glast + 1] = (uint16_t)(nch - 1); s_slow = 0u; s_tl = 0u; s_sp = 0u; s_bl = 0u; }
			#pragma unroll
			for (int k = 0; k < (F_BINS / 2) / F_THREADS; ++k) tabw[k * F_THREADS + tid] = 0;
			if (tid < 4) { ent[tid] = 0u; tabw[F_BINS / 2 + tid] = 0u; }      // the places in front of the bucket's first and behind its last entry: no symbol, not special
			if (tid < 12) ent[apad + m + tid] = 0u;
		}
		__syncthreads();
		if (lane < (uint32_t)F_PER) {
			const uint32_t g = lane * F_WAVES + wave;                   // (a group behind the bucket's last one: its lanes fetch the last record again)
			D = gdesc_build(rr, gfirst, g < glast ? g : glast, nch);
			if (D.pk >> 24) s_slow = 1u;                                // more than four runs in 64 slots: the bucket is listed
			if (g <= glast) {
				const uint32_t b1 = D.A + D.d1, b2 = b1 + D.d2, b3 = b2 + D.d3;
				gda[g] = make_uint4(D.pk & 0xFFu, (D.pk >> 8) & 0xFFu, (D.pk >> 16) & 0xFFu, D.A);
				gdb[g] = make_uint4(b1, b2, b3, 0u);
			}
		}
	}
	// the record of index e of the bucket (slot e of its rank range before sorting), through the LDS copy of the slot -> run map
	auto rec_adr = [&](const uint32_t e) -> uint32_t {
		const uint32_t g = e >> 6, x = e & 63u;
		const uint4 da = gda[g], db = gdb[g];
		return (x >= da.z ? db.z : x >= da.y ? db.y : x >= da.x ? db.x : da.w) + x;
	};
	const uint64_t posmask = (1ull << O.pb) - 1;
	const uint32_t kfull = m / F_THREADS;                        // items 0 .. kfull-1 have all their 512 slots inside the bucket
	// ---- the records, and all this thread needs of them from here on: the entry, and the bin (later: | arrival slot << 16) ----
	uint32_t ev[F_PER], bs[F_PER];
	{
		uint64_t r[F_PER];
		#pragma unroll
		for (int k = 0; k < F_PER; ++k) {
			GDESC_LANE(D, k, pk, A0, d1, d2, d3)
			// (the group of item k is the same for the whole wave: read into a SCALAR register, whether it lies wholly inside the bucket is decided by the scalar unit and
			// only the bucket's last group clamps its lanes -- five vector instructions per record less in a kernel whose vector ALUs are ~85 % busy: round 5's SQ counters)
			const uint32_t g = (uint32_t)(k * F_WAVES) + (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
			uint32_t x = lane;
			if (g >= glast) x = (g > glast || lane > xlast) ? xlast : lane;      // (the bucket's last group, or one behind it: its lanes fetch the last record again)
			r[k] = S.rec[(pk >> 24) ? A0 : gdesc_adr(pk, A0, d1, d2, d3, x)];
		}
		__syncthreads();                                             // (the loads are in flight; s_slow and the LDS copy of the map are published)
		if (s_slow) {
			if (tid == 0) { const FinishLists L = *Lp; uint32_t k = atomicAdd(&L.midlist[0], 1u); if (k < L.mid_cap) L.midlist[1 + k] = b; }
			return;
		}
		const int lowbits = fbits - F_BINBITS;                   // <= 12
		const int hks = rec_key_shift(O.pb) - 32, hps = O.pb - 32;
		const int bsh = hks + lowbits;
		const uint32_t lmask = (1u << lowbits) - 1u;
		const uint32_t start_lo = (uint32_t)O.start;
		#pragma unroll
		for (int k = 0; k < F_PER; ++k) {
			const uint32_t hi = (uint32_t)(r[k] >> 32), lo32 = (uint32_t)r[k], e = k * F_THREADS + tid;
			const bool special = lo32 == start_lo || sample_hit(O.smp, lo32);
			ev[k] = (((hi >> hks) & lmask) << E_LOW) | (e << E_IDX) | (special ? E_SPECIAL : 0u) | ((hi >> hps) & 7u);
			bs[k] = (hi >> bsh) & (uint32_t)(F_BINS - 1);
			if (special && ((uint32_t)k < kfull || e < m)) {     // rare: the position waits in LDS for whoever finds the entry at its final place
				const uint32_t j = atomicAdd(&s_sp, 1u);
				if (j < (uint32_t)V3_SP) sp[j] = make_uint2(lo32, e | ((hi & ((1u << hps) - 1u)) << 16));
			}
		}
	}
	if (xf == 1u) { if (ev[0] == 0xFFFFFFFFu && bs[1] == 0xFFFFFFFFu) O.bwt[s0] = 0; return; }
	// ---- count: arrival slot of every record inside its bin (16-bit counters, two per word) ----
	#pragma unroll
	for (int k = 0; k < F_PER; ++k) {
		const uint32_t bin = bs[k];
		if ((uint32_t)k < kfull || k * F_THREADS + tid < m) {
			LDS_CHK(bin >> 1, TABW_N);
			// (the half-word by shift amount, its old value by one bit-field extract: the select forms took three vector instructions each, and this kernel keeps the
			// vector ALUs ~85 % busy -- SQ counters of round 5: 1376 vector instructions per wave x 4.2 M waves x 4 cycles = 9.4 of its 10.8 ms)
			const uint32_t sh = (bin & 1u) << 4;
			const uint32_t old = atomicAdd(&tabw[bin >> 1], 1u << sh);
			bs[k] = bin | (__builtin_amdgcn_ubfe(old, sh, 16u) << 16);
		}
	}
	__syncthreads();
	if (xf == 2u) { if (bs[0] == 0xFFFFFFFFu) O.bwt[s0] = (uint8_t)ev[1]; return; }
	// ---- scan: thread = 16 consecutive bins = 8 words; half-word = start | min(count, 7) << 13; the bins with two entries or more go on the
	// work list (their number rides in the high half of the scanned sum) ----
	{
		constexpr int WPT = (F_BINS / 2) / F_THREADS;            // 8
		uint32_t w[WPT]; uint32_t sum = 0, cmax = 0, multi = 0;
		#pragma unroll
		for (int q = 0; q < WPT / 4; ++q) {
			const uint4 v = reinterpret_cast<const uint4 *>(tabw)[tid * (WPT / 4) + q];
			w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
		}
		#pragma unroll
		for (int k = 0; k < WPT; ++k) {
			const uint32_t lo = w[k] & 0xFFFFu, hi = w[k] >> 16;
			sum += lo + hi; cmax = cmax > lo ? cmax : lo; cmax = cmax > hi ? cmax : hi;
			multi |= ((lo > 1u ? 1u : 0u) | (hi > 1u ? 2u : 0u)) << (2 * k);
		}
		const uint32_t both = sum | ((uint32_t)__popc(multi) << 16);
		uint32_t inc = wave_incl_sum(both);
		if (lane == 63) wsum[wave] = inc;
		__syncthreads();
		uint32_t pre = inc - both;
		for (uint32_t i = 0; i < wave; ++i) pre += wsum[i];
		uint32_t run = pre & 0xFFFFu, wat = pre >> 16;
		if (tid == F_THREADS - 1) s_wl = (pre + both) >> 16;
		while (multi) {                                          // (stores only: nothing waits)
			const uint32_t k = (uint32_t)__ffs(multi) - 1u; multi &= multi - 1u;
			LDS_CHK(tid * (F_BINS / F_THREADS) + k, F_BINS);
			if (wat < (uint32_t)V3_WL) wl[wat] = (uint16_t)(tid * (F_BINS / F_THREADS) + k);
			++wat;
		}
		// a 16-mer repeated inside the bucket: that bin goes to the overflow list on its own
		if (cmax > (uint32_t)LS_TIE_CAP) {
			uint32_t st = run;
			#pragma unroll 1
			for (int k = 0; k < 2 * WPT; ++k) {
				const uint32_t c = (k & 1) ? w[k >> 1] >> 16 : w[k >> 1] & 0xFFFFu;
				if (c > (uint32_t)LS_TIE_CAP) ovf_push(O, s0 + st, c, true);
				st += c;
			}
		}
		#pragma unroll
		for (int k = 0; k < WPT; ++k) {
			const uint32_t lo = w[k] & 0xFFFFu, hi = w[k] >> 16;
			w[k] = (run | ((lo < 7u ? lo : 7u) << 13)) | (((run + lo) | ((hi < 7u ? hi : 7u) << 13)) << 16);
			run += lo + hi;
		}
		#pragma unroll
		for (int q = 0; q < WPT / 4; ++q) reinterpret_cast<uint4 *>(tabw)[tid * (WPT / 4) + q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
		if (tid == F_THREADS - 1) tabw[F_BINS / 2] = run;        // = m: "start" of the bin behind the last one
	}
	__syncthreads();
	const uint16_t *tab16 = reinterpret_cast<const uint16_t *>(tabw);
	if (xf == 3u) { if (bs[0] == 0xFFFFFFFFu && tab16[tid] == 0xFFFFu) O.bwt[s0] = (uint8_t)ev[1]; return; }
	// ---- entries into bin order (arrival order inside a bin) ----
	uint32_t tid_again = tid;
	asm volatile("" : "+v"(tid_again));                             // (the "slot inside the bucket" masks of the count phase are computed again, not kept: they were spilled)
	#pragma unroll
	for (int k = 0; k < F_PER; ++k) {
		const uint32_t e = k * F_THREADS + tid_again, bin = bs[k] & 0xFFFFu;
		const uint32_t t = tab16[bin];
		const uint32_t at = (t & T_START) + (bs[k] >> 16);
		if ((uint32_t)k < kfull || e < m) {
			LDS_CHK(apad + at, ENT_N); LDS_CHK(at, m);
			ent[apad + at] = ev[k];
			if ((t >> 13) == 7u) {                                // maybe a hot bin (above LS_TIE_CAP entries: on the overflow list): the list-based rounds read positions
				const uint32_t c = ((uint32_t)tab16[bin + 1] & T_START) - (t & T_START);
				if (c > (uint32_t)LS_TIE_CAP) sa_set(O.sa, s0 + at, S.rec[rec_adr(e)] & posmask);
			}
		}
	}
	__syncthreads();
	if (xf == 4u) { if (ent[tid] == 0xFFFFFFFFu) O.bwt[s0] = 1; return; }
	// ---- the listed bins: in order, in place; key ties noted ----
	// the entry x at place p ties with a neighbour; returns what to OR into the entry (no room on the list: its position is fetched when the bucket leaves,
	// by whoever writes its byte -- a bucket of an A+T-rich prefix ties a third of its records, and fetching them here, one after the other, cost 100 us per bucket)
	auto tie_note = [&](const uint32_t p, const uint32_t x, const bool has_pred) -> uint32_t {
		const uint32_t g = s0 + p;
		if (has_pred) atomicOr(&O.tiebits[g >> 5], 1u << (g & 31));       // bit p = "the record at place p has the key field of the one before it"
		const uint32_t j = atomicAdd(&s_tl, 1u);
		if (j < (uint32_t)V3_TL) { tl[j] = (p << 13) | ((x >> E_IDX) & 0x1FFFu); return 0u; }
		return E_PEND;
	};
	auto ekey = [&](const uint32_t x) -> uint32_t { return (x & ~E_PEND) >> E_LOW; };
	// a bin of cc entries from place st1 (entry index apad + st1) on, one entry after the other (rare: more than eight entries, or no room on the list)
	auto fix_serial = [&](const uint32_t at0, const uint32_t cc) {
		#pragma unroll 1
		for (uint32_t i = 1; i < cc; ++i) {
			const uint32_t x = ent[at0 + i]; uint32_t j = i;
			while (j > 0) { const uint32_t y = ent[at0 + j - 1]; if (y <= x) break; ent[at0 + j] = y; --j; }
			ent[at0 + j] = x;
		}
	};
	auto ties_serial = [&](const uint32_t at0, const uint32_t cc) {
		#pragma unroll 1
		for (uint32_t i = 0; i < cc; ++i) {
			const uint32_t x = ent[at0 + i];
			const bool tp = i > 0 && ekey(ent[at0 + i - 1]) == ekey(x), tn = i + 1 < cc && ekey(ent[at0 + i + 1]) == ekey(x);
			if (!(tp || tn)) continue;
			ent[at0 + i] = (x & ~E_SPECIAL) | tie_note(at0 - apad + i, x, tp);
		}
	};
	const uint32_t nwl = s_wl;
	if (nwl <= (uint32_t)V3_WL) {
		uint32_t bin[NQ], st[NQ], c[NQ], e0[NQ], e1[NQ], e2[NQ], e3[NQ];
		#pragma unroll
		for (int q = 0; q < NQ; ++q) { const uint32_t i = q * F_THREADS + tid; bin[q] = i < nwl ? wl[i] : 0xFFFFu; }
		#pragma unroll
		for (int q = 0; q < NQ; ++q) {
			const uint32_t t = tab16[bin[q] & (uint32_t)(F_BINS - 1)];
			st[q] = apad + (t & T_START); c[q] = bin[q] == 0xFFFFu ? 0u : t >> 13;
		}
		#pragma unroll
		for (int q = 0; q < NQ; ++q) { LDS_CHK(st[q] + 3, ENT_N); e0[q] = ent[st[q]]; e1[q] = ent[st[q] + 1]; e2[q] = ent[st[q] + 2]; e3[q] = ent[st[q] + 3]; }
		#pragma unroll
		for (int q = 0; q < NQ; ++q) {
			if ((uint32_t)(q * F_THREADS) >= nwl) break;         // (uniform)
			if (c[q] < 2u) continue;                               // (no bin)
			LDS_CHK(st[q] - apad + c[q], m + 1);
			if (c[q] <= 4u) {
				uint32_t x0 = e0[q], x1 = e1[q], x2 = c[q] > 2u ? e2[q] : 0xFFFFFFFFu, x3 = c[q] > 3u ? e3[q] : 0xFFFFFFFFu;
				B3M_CE(x0, x1) B3M_CE(x2, x3) B3M_CE(x0, x2) B3M_CE(x1, x3) B3M_CE(x1, x2)
				const bool t01 = (x0 >> E_LOW) == (x1 >> E_LOW), t12 = c[q] > 2u && (x1 >> E_LOW) == (x2 >> E_LOW), t23 = c[q] > 3u && (x2 >> E_LOW) == (x3 >> E_LOW);     // (the fill values sort behind the entries)
				if (t01 || t12 || t23) {          // what a tied record has to say about samples or p0 is said by k_tie_fix, once its place is final
					if (t01) x0 &= ~E_SPECIAL;
					if (t01 || t12) x1 &= ~E_SPECIAL;
					if (t12 || t23) x2 &= ~E_SPECIAL;
					if (t23) x3 &= ~E_SPECIAL;
					const uint32_t p = st[q] - apad;
					if (t01) x0 |= tie_note(p, x0, false);
					if (t01 || t12) x1 |= tie_note(p + 1, x1, t01);
					if (t12 || t23) x2 |= tie_note(p + 2, x2, t12);
					if (t23) x3 |= tie_note(p + 3, x3, true);
				}
				ent[st[q]] = x0; ent[st[q] + 1] = x1;
				if (c[q] > 2u) { ent[st[q] + 2] = x2; if (c[q] > 3u) ent[st[q] + 3] = x3; }
				continue;
			}
			// five entries and more (0.5 % of the listed bins; the count is capped at 7 in the table: the next bin's start tells)
			const uint32_t cc = ((uint32_t)tab16[bin[q] + 1] & T_START) - (st[q] - apad);
			if (cc > (uint32_t)LS_TIE_CAP) continue;              // hot: left in arrival order for the list-based rounds
			LDS_CHK(st[q] - apad + cc, m + 1);
			if (cc <= 8u) {
				uint32_t x[8];
				x[0] = e0[q]; x[1] = e1[q]; x[2] = e2[q]; x[3] = e3[q];
				#pragma unroll
				for (int i = 4; i < 8; ++i) x[i] = ent[st[q] + i];
				#pragma unroll
				for (int i = 5; i < 8; ++i) if ((uint32_t)i >= cc) x[i] = 0xFFFFFFFFu;
				// 19 comparators (Batcher's odd-even merge sort of 8)
				B3M_CE(x[0], x[1]) B3M_CE(x[2], x[3]) B3M_CE(x[4], x[5]) B3M_CE(x[6], x[7])
				B3M_CE(x[0], x[2]) B3M_CE(x[1], x[3]) B3M_CE(x[4], x[6]) B3M_CE(x[5], x[7])
				B3M_CE(x[1], x[2]) B3M_CE(x[5], x[6])
				B3M_CE(x[0], x[4]) B3M_CE(x[1], x[5]) B3M_CE(x[2], x[6]) B3M_CE(x[3], x[7])
				B3M_CE(x[2], x[4]) B3M_CE(x[3], x[5])
				B3M_CE(x[1], x[2]) B3M_CE(x[3], x[4]) B3M_CE(x[5], x[6])
				#pragma unroll
				for (int i = 0; i < 8; ++i) if ((uint32_t)i < cc) ent[st[q] + i] = x[i];
			} else {
				// nine entries and more: to the half-waves below
				const uint32_t j = atomicAdd(&s_bl, 1u);
				if (j < (uint32_t)V3_BL) { bl[j] = (uint16_t)bin[q]; continue; }
				fix_serial(st[q], cc);
			}
			ties_serial(st[q], cc);
		}
	} else {
		// more bins with two entries or more than the list holds (a bucket of few distinct keys): every scanner, its own, one after the other
		for (uint32_t bb = tid * (F_BINS / F_THREADS); bb < (tid + 1) * (F_BINS / F_THREADS); ++bb) {
			const uint32_t st1 = (uint32_t)tab16[bb] & T_START, cc = ((uint32_t)tab16[bb + 1] & T_START) - st1;
			if (cc < 2u || cc > (uint32_t)LS_TIE_CAP) continue;
			fix_serial(apad + st1, cc);
			ties_serial(apad + st1, cc);
		}
	}
	__syncthreads();
	// ---- bins of nine entries and more: half a wave each, every lane counts the entries smaller than its own (entries are distinct: the index is in them) ----
	{
		const uint32_t nbl = s_bl < (uint32_t)V3_BL ? s_bl : (uint32_t)V3_BL;      // (the same for every thread: the barrier inside is taken by all or none)
		if (nbl) {
			const uint32_t l = tid & 31u;
			for (uint32_t i = tid >> 5; i < ((nbl + 1u) & ~1u); i += F_THREADS / 32) {        // (both halves of a wave make the same number of trips)
				uint32_t at0 = 0, cc = 0;
				if (i < nbl) { const uint32_t bn = bl[i], st1 = (uint32_t)tab16[bn] & T_START; at0 = apad + st1; cc = ((uint32_t)tab16[bn + 1] & T_START) - st1; }
				const uint32_t x = l < cc ? ent[at0 + l] : 0xFFFFFFFFu;
				uint32_t less = 0;
				for (uint32_t j = 0; j < (uint32_t)LS_TIE_CAP; ++j) if (j < cc) less += ent[at0 + j] < x ? 1u : 0u;
				wave_sync();
				if (l < cc) ent[at0 + less] = x;
				wave_sync();
				if (l < cc) {
					const uint32_t y = ent[at0 + l];
					const bool tp = l > 0 && ekey(ent[at0 + l - 1]) == ekey(y), tn = l + 1 < cc && ekey(ent[at0 + l + 1]) == ekey(y);
					if (tp || tn) ent[at0 + l] = (y & ~E_SPECIAL) | tie_note(at0 - apad + l, y, tp);
				}
			}
			__syncthreads();
		}
	}
	// ---- the records of the tied entries: requested now, stored when the bucket has left ----
	const uint32_t ntl = s_tl < (uint32_t)V3_TL ? s_tl : (uint32_t)V3_TL;
	uint64_t trec = 0; uint32_t tplace = 0;
	if (tid < ntl) { const uint32_t it = tl[tid]; tplace = it >> 13; trec = S.rec[rec_adr(it & 0x1FFFu)]; }
	if (xf == 5u) { if (ent[tid] == 0xFFFFFFFFu) O.bwt[s0] = 1; return; }
	// ---- the bucket leaves: four entries = four BWT bytes = one dword (partial first / last dword bytewise) ----
	{
		uint8_t *gb = O.bwt + ((ptrdiff_t)s0 - (ptrdiff_t)apad);      // 4-byte aligned (may point before the block's first byte: never written)
		const uint32_t lo = apad, hi = apad + m;
		const uint32_t ndw = (hi + 3) >> 2;
		const uint32_t nsp = s_sp;
		for (uint32_t j = tid; j < ndw; j += F_THREADS) {
			const uint4 v = reinterpret_cast<const uint4 *>(ent)[j];
			const uint32_t bytes = (v.x & 7u) | ((v.y & 7u) << 8) | ((v.z & 7u) << 16) | ((v.w & 7u) << 24);
			if (4 * j >= lo && 4 * j + 4 <= hi) reinterpret_cast<uint32_t *>(gb)[j] = bytes;
			else { for (uint32_t q = 4 * j; q < 4 * j + 4; ++q) if (q >= lo && q < hi) gb[q] = (uint8_t)(bytes >> (8 * (q & 3u))); }
			if ((v.x | v.y | v.z | v.w) & E_PEND) {
				// tied records the list had no room for: their positions now (up to four loads in flight)
				const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
				uint64_t rc[4];
				#pragma unroll
				for (int q = 0; q < 4; ++q) rc[q] = (vv[q] & E_PEND) ? S.rec[rec_adr((vv[q] >> E_IDX) & 0x1FFFu)] : 0ull;
				#pragma unroll
				for (int q = 0; q < 4; ++q) if (vv[q] & E_PEND) sa_set(O.sa, (uint64_t)s0 + (4 * j + q - apad), rc[q] & posmask);
			}
			if ((v.x | v.y | v.z | v.w) & E_SPECIAL) {
				// a (rank,pos) sample or the block's first suffix, alone in its key: its place is final (entries outside the bucket are zero)
				const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
				#pragma unroll
				for (int q = 0; q < 4; ++q) {
					if (!(vv[q] & E_SPECIAL)) continue;
					const uint32_t idx = (vv[q] >> E_IDX) & 0x1FFFu;
					uint64_t pos = ~0ull;
					for (uint32_t i = 0; i < nsp && i < (uint32_t)V3_SP; ++i) { const uint2 z = sp[i]; if ((z.y & 0xFFFFu) == idx) pos = (uint64_t)z.x | ((uint64_t)(z.y >> 16) << 32); }
					if (pos == ~0ull) pos = S.rec[rec_adr(idx)] & posmask;      // (more specials than the list holds: dense sampling)
					const uint32_t g = s0 + (4 * j + q - apad);
					if (pos == O.start) *O.p0 = g;
					if (sample_hit(O.smp, (uint32_t)pos)) sample_put(O.smp, g, pos);
				}
			}
		}
	}
	if (tid < ntl) sa_set(O.sa, s0 + tplace, trec & posmask);
}

// Buckets the gathering finish has listed (above its LDS capacity, or spread over more runs than it maps): their runs are
// copied together into [bstart[b], bstart[b+1]) of `out` (the record buffer level 1 has read and no longer needs), where
// the mid / big / huge paths find them contiguous.  mid and big buckets (<= 2^20 records): one workgroup each; huge ones:
// every workgroup takes segments of 256 runs.
__device__ __forceinline__ void compact_segments(const uint32_t b, const FinishSrc &S, uint64_t *out, uint32_t seg_first, uint32_t seg_step)
{
	__shared__ uint32_t s_off[257];
	__shared__ uint32_t s_src[256];
	__shared__ uint32_t red[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	GLB_CHK(b < S.nbF);
	const uint32_t b0 = b >> S.D1, nch = S.btab[2 * S.nb0 + b0];
	const uint32_t *row = S.runtab + S.btab[S.nb0 + b0] + (uint64_t)(b & ((1u << S.D1) - 1u)) * nch;
	const uint32_t dst0 = S.bstart[b], cbase = S.btab[b0];
	for (uint32_t seg = seg_first; (uint64_t)seg * 256 < nch; seg += seg_step) {
		// records of the runs before this segment
		uint32_t part = 0;
		for (uint32_t t = tid; t < seg * 256; t += 256) part += row[t] >> 16;
		#pragma unroll
		for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
		__syncthreads();
		if (lane == 0) red[wave] = part;
		__syncthreads();
		const uint32_t base = red[0] + red[1] + red[2] + red[3];
		const uint32_t c = seg * 256 + tid;
		const uint32_t rent = c < nch ? row[c] : 0u, rc = rent >> 16;
		const uint32_t inc = wave_incl_sum(rc);
		__syncthreads();
		if (lane == 63) red[wave] = inc;
		__syncthreads();
		uint32_t ex = inc - rc;
		for (uint32_t i = 0; i < wave; ++i) ex += red[i];
		s_off[tid] = ex; s_src[tid] = cbase + c * S.chunk + (rent & 0xFFFFu);
		if (tid == 255) s_off[256] = ex + rc;
		__syncthreads();
		const uint32_t total = s_off[256];
		for (uint32_t x = tid; x < total; x += 256) {
			uint32_t lo = 0, hi = 256;                          // largest run with s_off <= x
			while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_off[mid] <= x) lo = mid; else hi = mid; }
			out[(uint64_t)dst0 + base + x] = S.rec[(uint64_t)s_src[lo] + (x - s_off[lo])];
		}
		__syncthreads();
	}
}
__global__ __launch_bounds__(256) void k_compact_listed(FinishSrc S, uint64_t *out, const uint32_t *midlist, uint32_t mid_cap, const uint32_t *biglist, uint32_t list_cap,
                                                         const uint32_t *hugelist, uint32_t huge_cap)
{
	uint32_t nm = midlist[0]; if (nm > mid_cap) nm = mid_cap;
	uint32_t nbg = biglist[0]; if (nbg > list_cap) nbg = list_cap;
	for (uint32_t it = blockIdx.x; it < nm + nbg; it += gridDim.x)
		compact_segments(it < nm ? midlist[1 + it] : biglist[1 + it - nm], S, out, 0, 1);
	uint32_t nh = hugelist[0]; if (nh > huge_cap) nh = huge_cap;
	for (uint32_t h = 0; h < nh; ++h) compact_segments(hugelist[1 + h], S, out, blockIdx.x, gridDim.x);
}

// HUGE BUCKETS (low-complexity input): every workgroup copies its share of the positions into the suffix array;
// the bucket is already on the overflow list
__global__ __launch_bounds__(256) void k_huge_spill(const uint64_t *rec, const uint32_t *bstart, const uint32_t *bend, const uint32_t *hugelist, uint32_t huge_cap, SaOut sa, int pb)
{
	uint32_t nh = hugelist[0]; if (nh > huge_cap) nh = huge_cap;
	for (uint32_t h = 0; h < nh; ++h) {
		const uint32_t b = hugelist[1 + h], s0 = bstart[b], m = (bend ? bend[b] : bstart[b + 1]) - s0;
		for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < m; e += (uint64_t)gridDim.x * 256) sa_set(sa, (uint64_t)s0 + e, rec[s0 + e] & ((1ull << pb) - 1));
	}
}

// listed buckets with no key bits left for another partition: the whole bucket to the overflow list, its positions to the suffix array
__global__ __launch_bounds__(256) void k_list_to_overflow(const uint64_t *rec, const uint32_t *bstart, const uint32_t *bend, const uint32_t *list, uint32_t cap, SaOut sa, int pb, SortOut O)
{
	uint32_t nl = list[0]; if (nl > cap) nl = cap;
	for (uint32_t h = 0; h < nl; ++h) {
		const uint32_t b = list[1 + h], s0 = bstart[b], m = (bend ? bend[b] : bstart[b + 1]) - s0;
		if (blockIdx.x == 0 && threadIdx.x == 0) ovf_push(O, s0, m);
		for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < m; e += (uint64_t)gridDim.x * 256) sa_set(sa, (uint64_t)s0 + e, rec[s0 + e] & ((1ull << pb) - 1));
	}
}

// BIG BUCKETS (above F_CAP; skewed input): one workgroup per listed bucket counts the bucket's next
// 8 bits, scatters it into its own range of the other record buffer (chunks of 4096 through LDS,
// as k_msd_pass), and its four waves then sort the 256 sub-buckets one after the other.
// where a pass of k_bucket_finish lists its sub-buckets above the wave sort's capacity for the NEXT pass (null list: there is none)
struct SubNext { uint32_t *st, *en, *list; uint32_t cap; };
// flags: 1 = a bin of the wave sort knows at least what a bin of the main finish knows, 2 = so does a whole sub-bucket of this pass
template <bool WRITE_SA>
__device__ __forceinline__ void big_bucket(const uint32_t b, const uint64_t *rec_in, uint64_t *rec_tmp, const uint32_t *bstart, const uint32_t *bend,
                                           int D, int shift, int sortbits, const SortOut &O, const SubNext &nx, int flags)
{
	constexpr int CH = MSD_THREADS * MSD_ITEMS;                 // 4096
	__shared__ uint32_t cnt[MSD_MAXBINS];
	__shared__ uint32_t binstart[MSD_MAXBINS];
	__shared__ uint32_t gbase[MSD_MAXBINS];
	__shared__ uint32_t sub[MSD_MAXBINS + 4];
	__shared__ uint32_t wsum[8];
	__shared__ __attribute__((aligned(16))) uint64_t big[CH];     // scatter staging, later the waves' sort buffers (32 KB)
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t s0 = bstart[b], s1 = bend ? bend[b] : bstart[b + 1], m = s1 - s0;
	const uint32_t bins = 1u << D, dmask = bins - 1;
	// ---- pass 1: histogram of the next digit over the bucket ----
	cnt[tid] = 0;
	__syncthreads();
	for (uint32_t e0 = 0; e0 < m; e0 += MSD_THREADS * 8) {        // 8 loads in flight per thread
		uint64_t v[8];
		#pragma unroll
		for (int k = 0; k < 8; ++k) { uint32_t e = e0 + k * MSD_THREADS + tid; v[k] = rec_in[(uint64_t)s0 + (e < m ? e : m - 1)]; }
		#pragma unroll
		for (int k = 0; k < 8; ++k) { uint32_t e = e0 + k * MSD_THREADS + tid; if (e < m) atomicAdd(&cnt[(uint32_t)(v[k] >> shift) & dmask], 1u); }
	}
	__syncthreads();
	{
		uint32_t v = cnt[tid], inc = wave_incl_sum(v);
		if (lane == 63) wsum[wave] = inc;
		__syncthreads();
		uint32_t pre = 0;
		for (uint32_t i = 0; i < wave; ++i) pre += wsum[i];
		uint32_t ex = pre + inc - v;
		sub[tid] = ex; gbase[tid] = s0 + ex;
		if (tid == MSD_THREADS - 1) sub[MSD_MAXBINS] = ex + v;
	}
	__syncthreads();
	// ---- pass 2: scatter the bucket into rec_tmp[s0, s1) ----
	for (uint32_t cbase = 0; cbase < m; cbase += CH) {
		cnt[tid] = 0;
		__syncthreads();
		uint64_t rec[MSD_ITEMS]; uint32_t dg[MSD_ITEMS], slot[MSD_ITEMS]; uint32_t vmask = 0;
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) {          // loads first, uses later: one wait for all 16
			uint32_t e = cbase + s * MSD_THREADS + tid;
			rec[s] = rec_in[(uint64_t)s0 + (e < m ? e : m - 1)];
			if (e < m) vmask |= 1u << s;
		}
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) dg[s] = (uint32_t)(rec[s] >> shift) & dmask;
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) if (vmask & (1u << s)) slot[s] = atomicAdd(&cnt[dg[s]], 1u);
		__syncthreads();
		{
			uint32_t v = cnt[tid], inc = wave_incl_sum(v);
			if (lane == 63) wsum[wave] = inc;
			__syncthreads();
			uint32_t pre = 0;
			for (uint32_t i = 0; i < wave; ++i) pre += wsum[i];
			binstart[tid] = pre + inc - v;
		}
		__syncthreads();
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) if (vmask & (1u << s)) big[binstart[dg[s]] + slot[s]] = rec[s];
		__syncthreads();
		const uint32_t nvalid = m - cbase < (uint32_t)CH ? m - cbase : (uint32_t)CH;
		for (uint32_t p = tid; p < nvalid; p += MSD_THREADS) { uint64_t q = big[p]; uint32_t d = (uint32_t)(q >> shift) & dmask; rec_tmp[(uint64_t)gbase[d] + (p - binstart[d])] = q; }
		__syncthreads();
		gbase[tid] += cnt[tid];
		__syncthreads();
	}
	__threadfence_block();
	__syncthreads();
	// ---- pass 3: the four waves sort the sub-buckets, each wave on its own slice of LDS ----
	uint64_t *outb = big + wave * LS_CAP;                                                  // 4 x 4 KB
	uint32_t *cb = reinterpret_cast<uint32_t *>(big + LS_WAVES * LS_CAP) + wave * 2 * LS_BINS;
	for (uint32_t d = wave; d < bins; d += LS_WAVES) {
		const uint32_t lo = sub[d], mm = sub[d + 1] - lo;
		if (mm == 0) continue;
		if (mm > (uint32_t)LS_CAP) {
			// too large for a wave: while key bits are left, the next pass partitions it again (it lies contiguous in rec_tmp); with none left its records
			// share their whole key field, and the list-based rounds take it from there
			if (nx.list && sortbits > 0) {
				uint32_t k = 0;
				if (lane == 0) k = atomicAdd(&nx.list[0], 1u);
				k = __shfl(k, 0, 64);
				if (k < nx.cap) { if (lane == 0) { nx.st[k] = s0 + lo; nx.en[k] = s0 + lo + mm; nx.list[1 + k] = k; } continue; }
			}
			if (lane == 0) ovf_push(O, s0 + lo, mm, sortbits <= 0 && (flags & 2));
			for (uint32_t e = lane; e < mm; e += 64) sa_set(O.sa, (uint64_t)s0 + lo + e, rec_tmp[(uint64_t)s0 + lo + e] & ((1ull << O.pb) - 1));
			continue;
		}
		uint64_t r[LS_CAP / 64];
		#pragma unroll
		for (int t = 0; t < LS_CAP / 64; ++t) { uint32_t e = t * 64 + lane; r[t] = rec_tmp[(uint64_t)s0 + lo + (e < mm ? e : mm - 1)]; }
		wave_sort_bucket<WRITE_SA>(s0 + lo, mm, sortbits, outb, cb, O, r, (flags & 1) != 0);
		wave_sync();
	}
}

template <bool WRITE_SA>
__global__ __launch_bounds__(MSD_THREADS) void k_bucket_finish(const uint64_t *rec_in, uint64_t *rec_tmp, const uint32_t *bstart, const uint32_t *bend, const uint32_t *biglist, uint32_t big_cap,
                                                                 int D, int shift, int sortbits, SortOut O, SubNext nx, int flags)
{
	uint32_t cnt = biglist[0]; cnt = cnt < big_cap ? cnt : big_cap;          // (what the list holds, not what was asked for)
	for (uint32_t it = blockIdx.x; it < cnt; it += gridDim.x) {        // a fixed grid strides over the list
		big_bucket<WRITE_SA>(biglist[1 + it], rec_in, rec_tmp, bstart, bend, D, shift, sortbits, O, nx, flags);
		__syncthreads();
	}
}

// one thread per word of the tie bitmap: every run of set bits starting in the word is a tie group
// (head = the position before the run).  The group's rotations agree on their first eq_syms
// symbols; they are ordered by the next 31 bases (one gather each), by full comparison if that
// does not decide.  Groups are at most LS_TIE_CAP long (longer ones are on the overflow list and
// set no bits); the length guard only keeps a corrupt bitmap from running away.
// The launch covers the records [lo, hi) (tie groups never cross a bucket, so slices of whole buckets are independent).
__device__ __forceinline__ void tie_group(const TextView &T, const uint32_t *tiebits, uint64_t len, uint64_t q, uint32_t eq_syms, const SaOut &sa, uint8_t *bwt, uint64_t start, uint32_t *p0,
                                          const SampleOut &smp, const TieExtras &X)
{
	{
		{
		uint64_t pos = q - 1;                                    // q = first tied record; head = q-1
		uint32_t g = 2;
		for (uint64_t z = q + 1; z < len && g <= (uint32_t)LS_TIE_CAP && ((tiebits[z >> 5] >> (z & 31)) & 1u); ++z) ++g;
		if (g > (uint32_t)LS_TIE_CAP) return;     // cannot happen (see above)
		if (g == 2 && !X.keys) {
			// two records (nearly every group of a text without repeats): both positions, both symbols and both windows are fetched side by side, not as the
			// chain of loads the insertion below makes of them (ties of C3: 1.06 -> 0.93 ms)
			const uint64_t y = sa_get(sa, pos), x = sa_get(sa, pos + 1);
			const uint8_t by = bwt[pos], bx = bwt[pos + 1];
			const bool ey = y + eq_syms + 32 <= T.nbases, ex = x + eq_syms + 32 <= T.nbases;
			const uint64_t dy = ey ? window32(T, y + eq_syms) >> 2 : 0, dx = ex ? window32(T, x + eq_syms) >> 2 : 0;
			const bool y_le_x = (ex && ey && dx != dy) ? dy < dx : compare_rotations(T, y, x) <= 0;
			if (!y_le_x) { sa_set(sa, pos, x); sa_set(sa, pos + 1, y); bwt[pos] = bx; bwt[pos + 1] = by; }
			const uint64_t f = y_le_x ? y : x, l2 = y_le_x ? x : y;
			if (f == start) *p0 = (uint32_t)pos;
			if (l2 == start) *p0 = (uint32_t)(pos + 1);
			if (sample_hit(smp, (uint32_t)f)) sample_put(smp, pos, f);
			if (sample_hit(smp, (uint32_t)l2)) sample_put(smp, pos + 1, l2);
			return;
		}
		for (uint32_t a = 1; a < g; ++a) {
			uint64_t x = sa_get(sa, pos + a); uint8_t bx = bwt[pos + a];
			bool ex = (uint64_t)x + eq_syms + 32 <= T.nbases;
			uint64_t dx = ex ? window32(T, (uint64_t)x + eq_syms) >> 2 : 0;
			uint32_t j = a;
			while (j > 0) {
				uint64_t y = sa_get(sa, pos + j - 1);
				bool ey = (uint64_t)y + eq_syms + 32 <= T.nbases;
				uint64_t dy = ey ? window32(T, (uint64_t)y + eq_syms) >> 2 : 0;
				bool y_le_x;
				if (ex && ey && dx != dy) y_le_x = dy < dx;
				else y_le_x = compare_rotations(T, y, x) <= 0;
				if (y_le_x) break;
				sa_set(sa, pos + j, y); bwt[pos + j] = bwt[pos + j - 1]; --j;
			}
			sa_set(sa, pos + j, x); bwt[pos + j] = bx;
		}
		for (uint32_t a = 0; a < g; ++a) {
			const uint64_t x = sa_get(sa, pos + a);
			if (x == start) *p0 = (uint32_t)(pos + a);
			if (sample_hit(smp, (uint32_t)x)) sample_put(smp, pos + a, x);
		}
		if (X.keys) {
			// (sequential suffix-range shards) members that agree with a neighbour on X.keys sort keys or more are listed with their slot: they are placed for good,
			// but the doubling rounds on the plan's list of tied suffixes must find their rank (b3m_internal.h: TieExtras)
			bool prev = false;
			for (uint32_t a = 0; a < g; ++a) {
				const uint64_t x = sa_get(sa, pos + a);
				bool next = false;
				if (a + 1 < g) {
					const uint64_t y = sa_get(sa, pos + a + 1);
					next = true;
					for (uint32_t d = 0; d < X.keys && next; ++d) next = key_at(T, x, d) == key_at(T, y, d);
				}
				if (prev || next) {
					const uint32_t k = atomicAdd(X.cnt, 1u);
					if (k < X.cap) { X.slot[k] = (uint32_t)(pos + a); X.pos_lo[k] = (uint32_t)x; if (X.pos_hi) X.pos_hi[k] = (uint8_t)(x >> 32); }
				}
				prev = next;
			}
		}
		}
	}
}
// starts of the tie groups of a word of the bitmap: set bits whose predecessor bit is clear, inside [lo, hi)
__device__ __forceinline__ uint32_t tie_starts(const uint32_t *tiebits, uint64_t w, uint64_t lo, uint64_t hi)
{
	const uint32_t bits = tiebits[w];
	if (!bits) return 0u;
	const uint32_t prevbit = w ? (tiebits[w - 1] >> 31) : 0u;
	uint32_t starts = bits & ~((bits << 1) | prevbit);
	if ((w << 5) < lo) starts &= ~((1u << (lo - (w << 5))) - 1u);             // runs of the neighbouring slices are theirs
	if ((w << 5) + 32 > hi) starts &= (1u << (hi - (w << 5))) - 1u;
	return starts;
}
__global__ void k_tie_fix(TextView T, const uint32_t *tiebits, uint64_t len, uint64_t lo, uint64_t hi, uint32_t eq_syms, SaOut sa, uint8_t *bwt, uint64_t start, uint32_t *p0, SampleOut smp,
                          TieExtras X)
{
	uint64_t w = (lo >> 5) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if ((w << 5) >= hi) return;
	uint32_t starts = tie_starts(tiebits, w, lo, hi);
	while (starts) {
		uint32_t k = __ffs(starts) - 1; starts &= starts - 1;
		tie_group(T, tiebits, len, (w << 5) + k, eq_syms, sa, bwt, start, p0, smp, X);
	}
}
// The same in two steps (whole blocks): the groups are few (0.3 % of the records) and each costs a chain of dependent gathers, so a
// thread per bitmap word keeps every wave alive for one lane's sake.  Step 1 lists the group starts (a stream over the bitmap, one
// atomic per wave), step 2 runs one group per lane, densely.
constexpr int TIE_WPT = 32;                   // bitmap words per thread of k_tie_collect: one atomic on the list counter per 8192 words
__global__ __launch_bounds__(256) void k_tie_collect(const uint32_t *tiebits, uint64_t len, uint32_t *list, uint32_t cap, uint32_t *counter)
{
	__shared__ uint32_t ws[4];
	__shared__ uint32_t s_base;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint64_t w0 = (uint64_t)blockIdx.x * (256 * TIE_WPT);
	uint32_t c = 0;
	#pragma unroll 4
	for (int i = 0; i < TIE_WPT; ++i) { const uint64_t w = w0 + (uint64_t)i * 256 + tid; if ((w << 5) < len) c += __popc(tie_starts(tiebits, w, 0, len)); }
	uint32_t inc = c;
	#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= (uint32_t)o) inc += t; }
	if (lane == 63) ws[wave] = inc;
	__syncthreads();
	if (tid == 0) { const uint32_t total = ws[0] + ws[1] + ws[2] + ws[3]; s_base = total ? atomicAdd(counter, total) : 0u; }
	__syncthreads();
	uint32_t at = s_base + inc - c;
	for (uint32_t i = 0; i < wave; ++i) at += ws[i];
	if (c == 0) return;
	for (int i = 0; i < TIE_WPT; ++i) {
		const uint64_t w = w0 + (uint64_t)i * 256 + tid;
		if ((w << 5) >= len) break;
		uint32_t starts = tie_starts(tiebits, w, 0, len);
		while (starts) {
			const uint32_t k = __ffs(starts) - 1; starts &= starts - 1;
			if (at < cap) list[at] = (uint32_t)((w << 5) + k);
			++at;
		}
	}
}
__global__ __launch_bounds__(256) void k_tie_fix_list(TextView T, const uint32_t *tiebits, uint64_t len, const uint32_t *list, const uint32_t *counter, uint32_t cap,
                                                      uint32_t eq_syms, SaOut sa, uint8_t *bwt, uint64_t start, uint32_t *p0, SampleOut smp, TieExtras X)
{
	uint32_t cnt = *counter; cnt = cnt < cap ? cnt : cap;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (uint64_t)gridDim.x * blockDim.x)
		tie_group(T, tiebits, len, list[i], eq_syms, sa, bwt, start, p0, smp, X);
}

// sorted (size << 32 | start) pairs -> sizes (scanned into list offsets by the caller) and starts
__global__ void k_ovf_split(const uint64_t *pairs, uint32_t nov, uint32_t *sizes, uint32_t *b0, uint32_t size_mask)
{
	uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j > nov) return;
	if (j == nov) { sizes[j] = 0; return; }
	uint64_t p = pairs[j];
	sizes[j] = (uint32_t)(p >> 32) & size_mask; b0[j] = (uint32_t)p;
}

// list of the suffix-array slots of the overflow buckets: U[s] = slot, gid[s] = list index of the bucket's first slot
__global__ void k_fill_overflow_list(const uint32_t *off, const uint32_t *b0, uint32_t nov, uint64_t total, uint32_t *U, uint32_t *gid)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= total) return;
	uint32_t lo = 0, hi = nov;          // last j with off[j] <= s
	while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (off[mid] <= s) lo = mid; else hi = mid; }
	U[s] = b0[lo] + (uint32_t)(s - off[lo]);
	if (gid) gid[s] = off[lo];                 // (label: the list index of the range's first slot, see tie_rounds: labels_are_heads)
}

// the same with one wave per range (a repeat-rich genome: a million ranges of a few hundred slots): consecutive lanes write consecutive slots, no search
__global__ __launch_bounds__(256) void k_fill_overflow_ranges(const uint32_t *off, const uint32_t *b0, uint32_t nov, uint32_t *U, uint32_t *gid)
{
	const uint32_t lane = threadIdx.x & 63u, nw = gridDim.x * 4u;
	for (uint32_t j = blockIdx.x * 4u + (threadIdx.x >> 6); j < nov; j += nw) {
		const uint32_t o = off[j], sz = off[j + 1] - o, b = b0[j];
		for (uint32_t t = lane; t < sz; t += 64u) { U[o + t] = b + t; if (gid) gid[o + t] = o; }      // (label: the list index of the range's first slot)
	}
}

// ---- suffix-range shards whose borders fall INSIDE a level-0 digit (skewed text) ----
// A shard is a range of 64-bit windows [key_lo, key_hi) (32 bases, the record sort's reading of the text: window_safe) instead of a range of whole digits.  The
// borders are quantiles of a fixed sample of the block's positions (every stride-th one: the same sample, hence the same borders, on every rank); what must be exact --
// a shard's size and the rank of its first suffix -- is counted by the level-0 passes themselves (k_l0_pass<.., EDGE>).  A window is a prefix of its suffix under a
// reading in which the terminator and what follows are the smallest base, so window order never contradicts suffix order and equal windows share a shard.
__global__ __launch_bounds__(256) void k_sample_windows(TextView T, uint64_t start, uint64_t stride, uint32_t ns, uint64_t *out)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < ns) out[i] = window_safe(T, start + (uint64_t)i * stride);
}
constexpr uint32_t SHARD_KEY_SAMPLES = 1u << 17;
struct ShardKeys {
	bool on = false;
	std::vector<uint64_t> sorted;          // the sample, ascending
	int nshards = 1;
	uint64_t border(int k) const { return k <= 0 ? 0 : sorted[(size_t)((uint64_t)sorted.size() * (unsigned)k / (unsigned)nshards)]; }      // key_lo of shard k (k < nshards)
	bool edge_lo(int k) const { return k > 0 && border(k) != 0; }
	bool edge_hi(int k) const { return k + 1 < nshards; }
	bool empty(int k) const { return edge_hi(k) && border(k + 1) <= border(k); }
	// the shard's range of D-bit digits: [lo, hi), hi = one behind the digit of the last window it may hold
	uint32_t dig_lo(int k, int D) const { return (uint32_t)(border(k) >> (64 - D)); }
	uint32_t dig_hi(int k, int D) const { if (empty(k)) return dig_lo(k, D); return edge_hi(k) ? (uint32_t)((border(k + 1) - 1) >> (64 - D)) + 1u : (1u << D); }
	double fraction(int k) const          // of the block's suffixes, by the sample
	{
		const auto lb = [&](uint64_t key) { return (double)(std::lower_bound(sorted.begin(), sorted.end(), key) - sorted.begin()); };
		const double a = k > 0 ? lb(border(k)) : 0.0, b = edge_hi(k) ? lb(border(k + 1)) : (double)sorted.size();
		return b > a ? (b - a) / (double)sorted.size() : 0.0;
	}
};
static void shard_keys_sample(ShardKeys &K, const TextView &T, uint64_t start, uint64_t len, int nshards, uint64_t *d_scratch, hipStream_t s)
{
	const uint32_t ns = (uint32_t)std::min<uint64_t>(len, SHARD_KEY_SAMPLES);
	const uint64_t stride = len / ns;
	hipLaunchKernelGGL(k_sample_windows, dim3((ns + 255) / 256), dim3(256), 0, s, T, start, stride, ns, d_scratch);
	K.sorted.resize(ns);
	B3M_HIP(hipMemcpyAsync(K.sorted.data(), d_scratch, 8ull * ns, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	std::sort(K.sorted.begin(), K.sorted.end());
	K.nshards = nshards; K.on = true;
}
// do whole-digit borders leave a shard above its share by more than half the slack a plan gives it (n / 64: engine.hip, shard_cap)?  cum: digit totals (any scale)
static bool digit_borders_unbalanced(const std::vector<uint64_t> &cum, uint32_t bins0, int nshards, uint64_t len_in)
{
	if (sw().SHARD_KEYS.set) return sw().SHARD_KEYS.v != 0;      // test switch: 1 = always borders by window, 0 = never
	if (!cum[bins0]) return false;
	uint32_t prev = 0;
	for (int k = 1; k <= nshards; ++k) {
		uint32_t d = bins0;
		if (k < nshards) { const uint64_t target = (uint64_t)((unsigned __int128)cum[bins0] * (unsigned)k / (unsigned)nshards); d = 0; while (d < bins0 && cum[d] < target) ++d; }
		const double est = (double)(cum[d] - cum[prev]) / (double)cum[bins0] * (double)len_in;
		if (est > (double)len_in / nshards + (double)len_in / 128.0) return true;
		prev = d;
	}
	return false;
}

static uint32_t big_grid() { const int v = (int)sw().BIG_GRID.get(2048); return (uint32_t)(v < 64 ? 64 : v); }      // (measurement switch: workgroups that stride over the listed buckets)
void block_sort_msd(const TextView &T, uint64_t start, uint64_t len_in, BlockSortWs &ws, hipStream_t s, Timeline *tl, BlockSortOut &out,
                    uint8_t *d_bwt, uint32_t *d_p0, int shard, int nshards, bool need_sa, uint64_t own, bool allow_ext)
{
	uint64_t len = len_in;           // becomes the number of KEPT suffixes after level 0 (suffix-range shard)
	MsdWs &M = ws.msd;
	// B3M_DEBUG=2: the stream is drained after every stage and the stage is named on stderr (which kernel a fault belongs to)
	const int dbg_stages = (int)sw().DEBUG.get(0);
	// (a GPU memory fault ends the process inside the runtime: the last "stage done" line then names the stage BEFORE the faulting one)
	auto stage = [&](const char *name) {
		if (dbg_stages < 2) return;
		const hipError_t e = hipStreamSynchronize(s);
		fprintf(stderr, "[b3m] stage %s: %s (start %llu len %llu)\n", e == hipSuccess ? "done" : "FAILED", name, (unsigned long long)start, (unsigned long long)len_in);
		fflush(stderr);
		B3M_HIP(e);
	};
	stage("enter block_sort_msd");
	B3M_REQUIRE(M.hist != nullptr, "MSD workspace missing");
	const MsdDigits dg = msd_digits(len_in);
	B3M_REQUIRE(dg.D0 >= 1 && dg.D0 <= W0_MAXBITS && dg.D1 >= 0 && dg.D1 <= MSD_MAXBITS, "internal: MSD digit split");
	const bool big = T.n >= (1ull << 32);                    // positions need more than 32 bits: 36-bit record positions, suffix array in two arrays
	B3M_REQUIRE(!big || ws.sa_hi != nullptr, "internal: workspace has no room for positions above 2^32");
	B3M_REQUIRE(T.n < (1ull << POSBITS_BIG), "texts of 2^36 symbols and more are not supported");
	const int pb = big ? POSBITS_BIG : 32;
	const int REC_KEY_BITS = rec_key_bits(pb);
	const SaOut sa{ws.idxA, big ? ws.sa_hi : nullptr};
	out.used_msd = true;
	// ---- level 0: text -> records (keyA), partitioned by the first D0 bits ----
	const uint64_t P0 = start & ~15ull;
	const uint64_t span = start + len_in - P0;
	bool geoA = dg.D0 <= 11;
	uint64_t tile0 = geoA ? (uint64_t)W0A_CHUNK * W0A_TCH : (uint64_t)W0B_CHUNK * W0B_TCH;
	uint32_t ntA = (uint32_t)((span + tile0 - 1) / tile0);
	uint32_t bins0 = 1u << dg.D0;          // buckets leaving level 0 (a shard: the bins of its own digit range, see below)
	int D0e = dg.D0;                        // digit bits of level 0 as it finally runs
	uint32_t *hist0 = M.hist;               // row 0 of the [bin][tile] matrix of that pass
	B3M_REQUIRE((uint64_t)bins0 * ntA <= M.hist_cap, "MSD histogram workspace too small");
	L0Args A{};
	A.T = T; A.start = start; A.len = len; A.P0 = P0; A.rec_out = ws.keyA; A.hist = M.hist; A.D = dg.D0;
	A.keep_lo = 0; A.keep_hi = bins0; A.pb = pb;
	// Chunk-local level 0 (unsharded blocks with a level 1, 32-bit positions): no counting pass, the records of a chunk stay in the
	// chunk's own index range, level 1 gathers (k_l0_pass<.., LOCAL>, k_l1_gather).  B3M_L0_LOCAL=0: the scattering form.
	const int l0_local_on = (int)sw().L0_LOCAL.get(1);
	const uint32_t nchunks0 = (uint32_t)((span + W0A_CHUNK - 1) / W0A_CHUNK);
	bool local0 = l0_local_on && nshards <= 1 && geoA && dg.D1 > 0 && !big && span <= ws.cap && M.tab0 != nullptr
	              && (uint64_t)bins0 * ((uint64_t)ntA * W0A_TCH) <= M.tab0_cap && span < (1ull << 32);
	const uint32_t ncol0 = ntA * (uint32_t)W0A_TCH;             // columns of the level-0 run table (whole tiles)
	(void)nchunks0;
	unsigned long long *d_tot = reinterpret_cast<unsigned long long *>(M.sub_start);   // free until level 0 is done
	if (!local0 && ws.arrival) for (int i = 0; i < ws.arrival->n; ++i) B3M_HIP(hipStreamWaitEvent(s, ws.arrival->ev[i], 0));
	// ---- a suffix-range shard in ONE pass over the text (B3M_SHARD_ONEPASS=0: the counting pass + the scattering pass) ----
	// The shard's borders need digit totals of the whole text before a record can be kept or dropped -- but only to balance the shards, and
	// every rank must arrive at the SAME borders without talking: a count over every 16th tile does both (the same sample on every rank;
	// the shards then differ by the sampling error, ~0.1 %).  What must be exact is counted by the pass itself: the shard's size (what its chunks
	// reserved) and the global rank of its first suffix (the positions with a smaller digit, A.below).  The pass is the chunk-local level 0
	// restricted to the shard's digits: a chunk sorts the 1/N of its positions it keeps and puts them behind what earlier chunks reserved
	// (cbase[column]); level 1 gathers through the run table as for a whole block.
	bool shard1 = false, borders_fixed = false; uint32_t fixed_lo = 0, fixed_hi = 0;
	ShardKeys SK;                          // borders inside digits (skewed text), once either form of the shard's level 0 has asked for them
	{
		const int onepass_on = (int)sw().SHARD_ONEPASS.get(1);
		int tfin = 0; while ((len_in >> tfin) > (uint64_t)F_AVG) ++tfin;
		if (nshards > 1 && onepass_on && l0_local_on && geoA && dg.D1 > 0 && !big && M.tab0 != nullptr && M.cbase0 != nullptr && span < (1ull << 32)
		    && ncol0 <= M.cbase0_cap && !allow_ext && own == 0 && tfin - dg.D0 - dg.D1 <= 0) {
			const uint32_t stride = ntA >= 64 ? 16u : 1u, ns = (ntA + stride - 1) / stride;
			{
				Scope sc(tl, KF_MSD_HIST, span / 4 / stride);
				A.tile_stride = stride;
				hipLaunchKernelGGL((k_l0_pass<true, W0A_CHUNK, W0A_TCH, W0A_BINS>), dim3(ns), dim3(W0_THREADS), 0, s, A);
				A.tile_stride = 0;
				hipLaunchKernelGGL(k_row_totals, dim3(bins0), dim3(256), 0, s, M.hist, ns, d_tot);
			}
			std::vector<unsigned long long> tot(bins0);
			B3M_HIP(hipMemcpyAsync(tot.data(), d_tot, 8ull * bins0, hipMemcpyDeviceToHost, s));
			B3M_HIP(hipStreamSynchronize(s));
			std::vector<uint64_t> cum(bins0 + 1, 0);
			for (uint32_t d = 0; d < bins0; ++d) cum[d + 1] = cum[d] + tot[d];
			auto bound = [&](int k) -> uint32_t {
				if (k <= 0) return 0; if (k >= nshards) return bins0;
				const uint64_t target = (uint64_t)((unsigned __int128)cum[bins0] * (unsigned)k / (unsigned)nshards);
				uint32_t d = 0; while (d < bins0 && cum[d] < target) ++d;
				return d;
			};
			uint32_t lo = bound(shard), hi = bound(shard + 1);
			double est = cum[bins0] ? (double)(cum[hi] - cum[lo]) / (double)cum[bins0] * (double)len_in : 0.0;
			if (digit_borders_unbalanced(cum, bins0, nshards, len_in)) {
				// skewed text: whole digits do not balance the shards; borders by 64-bit window instead (the same decision and the same sample on every rank)
				shard_keys_sample(SK, T, start, len_in, nshards, ws.keyB, s);
				lo = SK.dig_lo(shard, dg.D0); hi = SK.dig_hi(shard, dg.D0); est = SK.fraction(shard) * (double)len_in;
				if (dbg_stages >= 1) fprintf(stderr, "[b3m] shard %d of %d: borders by window: [%016llx, %016llx) = digits %u..%u, ~%.0f suffixes\n", shard, nshards,
				                             (unsigned long long)SK.border(shard), (unsigned long long)(SK.edge_hi(shard) ? SK.border(shard + 1) : ~0ull), lo, hi, est);
			}
			// (whether THIS shard then takes the one pass or, for want of room, the two: its borders are these -- the other shards' are)
			borders_fixed = true; fixed_lo = lo; fixed_hi = hi;
			const uint64_t capr = std::min<uint64_t>(ws.cap, 0xFFFFFFFFull);
			const unsigned two_mask = (unsigned)sw().SHARD_TWO_PASS.get(0);      // test switch: these shards (bit k = shard k) take the two passes, the others one
			if (hi > lo && (uint64_t)(hi - lo) * ncol0 <= M.tab0_cap && est <= 0.97 * (double)capr && !((two_mask >> (shard & 31)) & 1u)) {
				uint32_t *d_alloc = M.cbase0 + M.cbase0_cap;
				unsigned long long *d_below = reinterpret_cast<unsigned long long *>(d_alloc + 2);
				B3M_HIP(hipMemsetAsync(d_alloc, 0, 32 + 64, s));
				L0Args B = A;
				B.keep_lo = lo; B.keep_hi = hi; B.runtab0 = M.tab0; B.nchunks0 = ncol0; B.tile_base = 0;
				B.cbase = M.cbase0; B.alloc = d_alloc; B.below = d_below; B.cap = (uint32_t)capr;
				if (SK.on) { B.key_lo = SK.border(shard); B.key_hi = SK.edge_hi(shard) ? SK.border(shard + 1) : 0; B.edge = (SK.edge_lo(shard) ? 1u : 0u) | (SK.edge_hi(shard) ? 2u : 0u); }
				// the block's symbol histogram rides along (no second pass over the text for it); a fallback to the two passes must not count twice
				const bool want_hist = ws.d_symhist_out != nullptr;
				unsigned long long *d_h5 = reinterpret_cast<unsigned long long *>(d_alloc + 8);      // five counters behind the scalars, added to the caller's array on success
				B.symhist = want_hist ? d_h5 : nullptr;
				// three shards and more: 512-thread workgroups with a third of the index array, two to a CU (B3M_SHARD_WG2=0: always the 1024-thread form);
				// a chunk that keeps more than that array holds raises the flag and the pass is repeated in the other form
				const int wg2_on = (int)sw().SHARD_WG2.get(1);
				uint32_t h_sc[4] = {0, 0, 0, 0};
				for (int form = (wg2_on && est <= 0.30 * (double)len_in) ? 2 : 1; form >= 1; --form) {
					if (form == 1 && h_sc[1]) B3M_HIP(hipMemsetAsync(d_alloc, 0, 32 + 64, s));      // (the flagged attempt's counts start over)
					Scope sc(tl, KF_MSD_SCATTER_TEXT, span / 4 + 8 * (uint64_t)est);
					if (SK.on) {
						if (form == 2) hipLaunchKernelGGL((k_l0_pass<false, W0A_CHUNK, W0A_TCH, W0A_BINS, true, 2, true>), dim3(ntA), dim3(W0S_THREADS), 0, s, B);
						else hipLaunchKernelGGL((k_l0_pass<false, W0A_CHUNK, W0A_TCH, W0A_BINS, true, 1, true>), dim3(ntA), dim3(W0_THREADS), 0, s, B);
					}
					else if (form == 2) hipLaunchKernelGGL((k_l0_pass<false, W0A_CHUNK, W0A_TCH, W0A_BINS, true, 2>), dim3(ntA), dim3(W0S_THREADS), 0, s, B);
					else hipLaunchKernelGGL((k_l0_pass<false, W0A_CHUNK, W0A_TCH, W0A_BINS, true, 1>), dim3(ntA), dim3(W0_THREADS), 0, s, B);
					B3M_HIP(hipMemcpyAsync(h_sc, d_alloc, 16, hipMemcpyDeviceToHost, s));
					B3M_HIP(hipStreamSynchronize(s));
					if (!h_sc[1]) break;
					if (dbg_stages >= 1) fprintf(stderr, "[b3m] one-pass shard %d of %d: a chunk keeps more positions than the 512-thread form's index array holds: the pass again with 1024 threads\n", shard, nshards);
				}
				stage("level 0 (one-pass shard)");
				if ((uint64_t)h_sc[0] <= capr) {
					shard1 = true; local0 = true;
					if (want_hist) { hipLaunchKernelGGL(k_add5_u64, dim3(1), dim3(64), 0, s, d_h5, reinterpret_cast<unsigned long long *>(ws.d_symhist_out)); out.symhist_done = true; }
					bins0 = hi - lo; len = h_sc[0];
					out.slice_offset = (uint64_t)h_sc[2] | ((uint64_t)h_sc[3] << 32);
					A.keep_lo = lo; A.keep_hi = hi;
					Scope sc(tl, KF_SCAN, 4ull * bins0 * ncol0);
					if (bins0) hipLaunchKernelGGL(k_rows0_totals, dim3(bins0), dim3(256), 0, s, M.tab0, ncol0, d_tot);
				}
				// (more than the workspace holds, against the estimate: nothing is lost, the two passes below start over and say what is too large)
			}
		}
	}
	if (shard1) {
		// (the pass has counted the block's symbols itself)
	} else if (local0) {
		{
			Scope sc(tl, KF_MSD_SCATTER_TEXT, span / 4 + 8 * len);
			A.runtab0 = M.tab0; A.nchunks0 = ncol0;                  // (every column is written by its chunk, empty ones too)
			// a text that is still arriving: tiles go out as the bytes they read (their positions, a word in front, 48 bases behind) are in place
			uint32_t t_done = 0;
			if (ws.arrival) {
				for (int i = 0; i < ws.arrival->n && t_done < ntA; ++i) {
					const uint64_t bytes = ws.arrival->bytes_ready[i];
					const uint64_t pos_ok = bytes >= 32 ? (bytes - 32) * 4 : 0;                       // positions whose windows are complete
					uint32_t t_end = i + 1 == ws.arrival->n ? ntA : (uint32_t)std::min<uint64_t>(ntA, pos_ok > P0 ? (pos_ok - P0) / tile0 : 0);
					B3M_HIP(hipStreamWaitEvent(s, ws.arrival->ev[i], 0));
					if (t_end > t_done) {
						A.tile_base = t_done;
						hipLaunchKernelGGL((k_l0_pass<false, W0A_CHUNK, W0A_TCH, W0A_BINS, true>), dim3(t_end - t_done), dim3(W0_THREADS), 0, s, A);
						t_done = t_end;
					}
				}
			}
			if (t_done < ntA) {
				A.tile_base = t_done;
				hipLaunchKernelGGL((k_l0_pass<false, W0A_CHUNK, W0A_TCH, W0A_BINS, true>), dim3(ntA - t_done), dim3(W0_THREADS), 0, s, A);
			}
		}
		Scope sc(tl, KF_SCAN, 4ull * bins0 * ncol0);
		stage("level 0 (chunk-local)");
		hipLaunchKernelGGL(k_rows0_totals, dim3(bins0), dim3(256), 0, s, M.tab0, ncol0, d_tot);
	} else {
		Scope sc(tl, KF_MSD_HIST, span / 4);
		if (geoA) hipLaunchKernelGGL((k_l0_pass<true, W0A_CHUNK, W0A_TCH, W0A_BINS>), dim3(ntA), dim3(W0_THREADS), 0, s, A);
		else hipLaunchKernelGGL((k_l0_pass<true, W0B_CHUNK, W0B_TCH, W0B_BINS>), dim3(ntA), dim3(W0_THREADS), 0, s, A);
	}
	bool totals_ready = local0 && !shard1;  // d_tot holds this pass's digit totals (NOT out.symhist_done: a pass that starts over with the suffix array kept inherits that flag, while d_tot has been overwritten since)
	if (ws.d_symhist_out && dg.D0 >= 2 && !shard1) {
		if (!local0) hipLaunchKernelGGL(k_row_totals, dim3(bins0), dim3(256), 0, s, M.hist, ntA, d_tot);
		totals_ready = true;
		hipLaunchKernelGGL(k_symhist_from_totals, dim3(1), dim3(256), 0, s, d_tot, bins0, dg.D0, T.type == B3M_INPUT_PACTERM ? 1 : 0,
		                   (T.type == B3M_INPUT_PACTERM && start + len_in == T.n) ? 1 : 0, reinterpret_cast<unsigned long long *>(ws.d_symhist_out));
		out.symhist_done = true;
	}
	if (!shard1) out.slice_offset = 0;
	if (nshards > 1 && !shard1) {
		// suffix-range shard: this rank keeps the level-0 digits [lo, hi) holding the ranks
		// [shard*n/N, (shard+1)*n/N) as evenly as whole digits allow.  The totals come from the
		// text alone, so every rank derives the same boundaries without talking to anyone.
		if (!totals_ready) hipLaunchKernelGGL(k_row_totals, dim3(bins0), dim3(256), 0, s, M.hist, ntA, d_tot);
		std::vector<unsigned long long> tot(bins0);
		B3M_HIP(hipMemcpyAsync(tot.data(), d_tot, 8ull * bins0, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipStreamSynchronize(s));
		std::vector<uint64_t> cum(bins0 + 1, 0);
		for (uint32_t d = 0; d < bins0; ++d) cum[d + 1] = cum[d] + tot[d];
		auto bound = [&](int k) -> uint32_t {          // first digit whose cumulative count reaches k*n/N
			if (k <= 0) return 0; if (k >= nshards) return bins0;
			uint64_t target = (uint64_t)((unsigned __int128)cum[bins0] * (unsigned)k / (unsigned)nshards);
			uint32_t d = 0; while (d < bins0 && cum[d] < target) ++d;
			return d;
		};
		uint32_t lo = bound(shard), hi = bound(shard + 1);
		if (borders_fixed) { lo = fixed_lo; hi = fixed_hi; }      // the borders every shard of this text uses (from the sampled count above); sizes and offsets below are exact
		else if (digit_borders_unbalanced(cum, bins0, nshards, len_in)) shard_keys_sample(SK, T, start, len_in, nshards, ws.keyB, s);      // skewed text: borders by 64-bit window
		if (SK.on) { lo = SK.dig_lo(shard, dg.D0); hi = SK.dig_hi(shard, dg.D0); }
		out.slice_offset = cum[lo];
		len = cum[hi] - cum[lo];             // (borders by window: an upper bound until the count below)
		if (!SK.on && !(len <= ws.cap && len < (1ull << 32))) {
			uint64_t mx = 0; for (uint32_t d = lo; d < hi; ++d) mx = std::max<uint64_t>(mx, tot[d]);
			throw Error("suffix-range shard larger than the workspace (skewed input): shard " + std::to_string(shard) + " of " + std::to_string(nshards) + " would hold " +
			            std::to_string(len) + " suffixes (level-0 digits " + std::to_string(lo) + ".." + std::to_string(hi) + ", the heaviest has " + std::to_string(mx) +
			            "), the workspace " + std::to_string(ws.cap) + (ws.no_defer
			                ? ": the shards of this plan run one after the other inside a memory target (host tier); a shard is whole level-0 digits, and this text's heaviest one does not fit: "
			                  "raise mem= (about 30 bytes per suffix of the heaviest digit, beside the text) or leave it out"
			                : ": use block mode or fewer shards"));
		}
		// The shard's level 0 only has bins for its own digits [lo, hi).  Where the text is so long that D0 + D1 bits leave final
		// buckets above the LDS sort's size (n > ~6 G: 30 Gbase has 2^23 of them), the shard spends the bins it does not need for
		// other shards' digits on S more bits of its own: one more counting pass over the text, and level 0 partitions by D0 + S bits.
		int t = 0; while ((len_in >> t) > (uint64_t)F_AVG) ++t;
		int S = 0;
		while (hi > lo && S < t - dg.D0 - dg.D1 && dg.D0 + S < 15 && ((uint64_t)(hi - lo) << (S + 1)) <= (uint64_t)W0_MAXBINS) ++S;
		const int no_fine = (int)sw().NO_FINE_SHARDS.get(0);      // measurement switch
		if (no_fine) S = 0;
		if (sw().FINE_MAXS.set && S > (int)sw().FINE_MAXS.v) S = (int)sw().FINE_MAXS.v;
		A.keep_lo = lo << S; A.keep_hi = hi << S; A.D = dg.D0 + S; D0e = dg.D0 + S;
		if (SK.on) {
			// the window borders cut finer than (lo, hi) << S: the digits that can hold a window of the shard, and the two edge tests inside the first and the last
			A.keep_lo = SK.dig_lo(shard, D0e); A.keep_hi = SK.dig_hi(shard, D0e);
			A.key_lo = SK.border(shard); A.key_hi = SK.edge_hi(shard) ? SK.border(shard + 1) : 0; A.edge = (SK.edge_lo(shard) ? 1u : 0u) | (SK.edge_hi(shard) ? 2u : 0u);
		}
		bins0 = A.keep_hi - A.keep_lo;
		if (SK.on) {
			// always counted again: the rows of the first count hold whole digits.  The pass also counts what lies in front of the shard (its first rank), an empty shard's too
			geoA = bins0 <= (uint32_t)W0A_BINS;
			tile0 = geoA ? (uint64_t)W0A_CHUNK * W0A_TCH : (uint64_t)W0B_CHUNK * W0B_TCH;
			ntA = (uint32_t)((span + tile0 - 1) / tile0);
			B3M_REQUIRE((uint64_t)std::max(bins0, 1u) * ntA <= M.hist_cap, "MSD histogram workspace too small");
			unsigned long long *d_below = d_tot + W0_MAXBINS;      // (behind the totals of at most W0_MAXBINS rows)
			B3M_HIP(hipMemsetAsync(d_below, 0, 8, s));
			A.below = d_below;
			{
				Scope sc(tl, KF_MSD_HIST, span / 4);
				if (geoA) hipLaunchKernelGGL((k_l0_pass<true, W0A_CHUNK, W0A_TCH, W0A_BINS, false, 0, true>), dim3(ntA), dim3(W0_THREADS), 0, s, A);
				else hipLaunchKernelGGL((k_l0_pass<true, W0B_CHUNK, W0B_TCH, W0B_BINS, false, 0, true>), dim3(ntA), dim3(W0_THREADS), 0, s, A);
				if (bins0) hipLaunchKernelGGL(k_row_totals, dim3(bins0), dim3(256), 0, s, M.hist, ntA, d_tot);
			}
			std::vector<unsigned long long> kt(bins0); unsigned long long below = 0;
			if (bins0) B3M_HIP(hipMemcpyAsync(kt.data(), d_tot, 8ull * bins0, hipMemcpyDeviceToHost, s));
			B3M_HIP(hipMemcpyAsync(&below, d_below, 8, hipMemcpyDeviceToHost, s));
			B3M_HIP(hipStreamSynchronize(s));
			len = 0; for (uint32_t d = 0; d < bins0; ++d) len += kt[d];
			out.slice_offset = below;
			if (dbg_stages >= 1) fprintf(stderr, "[b3m] shard %d of %d: borders by window: [%016llx, %016llx) = %u digits of %d bits, %llu suffixes behind %llu\n", shard, nshards,
			                             (unsigned long long)A.key_lo, (unsigned long long)(SK.edge_hi(shard) ? A.key_hi : ~0ull), bins0, D0e, (unsigned long long)len, below);
			if (!(len <= ws.cap && len < (1ull << 32)))
				throw Error("suffix-range shard larger than the workspace (skewed input): shard " + std::to_string(shard) + " of " + std::to_string(nshards) + " would hold " +
				            std::to_string(len) + " suffixes between two 32-base windows, the workspace " + std::to_string(ws.cap) +
				            (ws.no_defer ? ": raise mem= or leave it out" : ": use block mode or fewer shards"));
		}
		else if (S == 0) hist0 = M.hist + (uint64_t)lo * ntA;      // the rows of the first count serve
		else if (bins0) {
			geoA = bins0 <= (uint32_t)W0A_BINS;
			tile0 = geoA ? (uint64_t)W0A_CHUNK * W0A_TCH : (uint64_t)W0B_CHUNK * W0B_TCH;
			ntA = (uint32_t)((span + tile0 - 1) / tile0);
			B3M_REQUIRE((uint64_t)bins0 * ntA <= M.hist_cap, "MSD histogram workspace too small");
			Scope sc(tl, KF_MSD_HIST, span / 4);
			if (geoA) hipLaunchKernelGGL((k_l0_pass<true, W0A_CHUNK, W0A_TCH, W0A_BINS>), dim3(ntA), dim3(W0_THREADS), 0, s, A);
			else hipLaunchKernelGGL((k_l0_pass<true, W0B_CHUNK, W0B_TCH, W0B_BINS>), dim3(ntA), dim3(W0_THREADS), 0, s, A);
		}
		A.hist = hist0;
	}
	out.kept_len = len;
	B3M_REQUIRE(len < (1ull << 32), "more than 2^32 suffixes in one sort: use suffix-range shards");
	if (len == 0) { out.d_sa = sa.lo; out.d_sa_hi = sa.hi; return; }          // empty shard (one huge digit took the neighbours' share)
	if (!local0) {
		{
			Scope sc(tl, KF_SCAN, 8ull * bins0 * ntA);
			exclusive_sum_u32(hist0, (uint64_t)bins0 * ntA, M.scan_ws, s);
		}
		Scope sc(tl, KF_MSD_SCATTER_TEXT, span / 4 + 8 * len);
		if (SK.on) {
			if (geoA) hipLaunchKernelGGL((k_l0_pass<false, W0A_CHUNK, W0A_TCH, W0A_BINS, false, 0, true>), dim3(ntA), dim3(W0_THREADS), 0, s, A);
			else hipLaunchKernelGGL((k_l0_pass<false, W0B_CHUNK, W0B_TCH, W0B_BINS, false, 0, true>), dim3(ntA), dim3(W0_THREADS), 0, s, A);
		}
		else if (geoA) hipLaunchKernelGGL((k_l0_pass<false, W0A_CHUNK, W0A_TCH, W0A_BINS>), dim3(ntA), dim3(W0_THREADS), 0, s, A);
		else hipLaunchKernelGGL((k_l0_pass<false, W0B_CHUNK, W0B_TCH, W0B_BINS>), dim3(ntA), dim3(W0_THREADS), 0, s, A);
	}
	{
		uint32_t tab0[3] = {0, 0, ntA};               // the one bucket entering level 0: {start, hist_off, ntiles}
		B3M_HIP(hipMemcpyAsync(M.bucket_tab, tab0, 12, hipMemcpyHostToDevice, s));
		B3M_HIP(hipStreamSynchronize(s));
	}
	const uint32_t eq_syms = (uint32_t)((D0e + REC_KEY_BITS) / 2);       // symbols known equal when key fields tie
	uint32_t *d_tiebits = ws.t[0];      // tie bitmap: len bits
	const uint64_t *cur = ws.keyA; uint64_t *other = ws.keyB;
	uint32_t nbF = bins0;               // final buckets
	const int consumed = dg.D1;         // key-field bits used by level 1
	bool sliced = false, lvl2_ran = false, listed_first = false;      // listed_first: the first finish launch listed buckets (they are written after their slice has left)
	// B3M_L1_ORDER=1 (measurement switch): level 1's tiles listed tile-major (tile j of every bucket, then j + 1) and taken in XCD-aware order, so that tile j of
	// buckets d and d + 1 -- neighbouring runs of the same level-0 chunks -- run side by side on one XCD.  It takes the reads from 36.1 to 25.4 GB (PMC) and
	// the kernel from 11.31 to 11.52 ms: level 1 is not waiting for those bytes (profiles/NOTES_r03.md).  Default: bucket by bucket, launch order.
	const int l1_order = (int)sw().L1_ORDER.get(0);
	const int xcd_on = (int)sw().XCD_ORDER.get(1);       // measurement switch: 0 = workgroups take their items in launch order
	FinishSrc FS{};                     // where the finish finds the records of a final bucket
	FS.bstart = M.lvl_start; FS.xcd = (xcd_on ? 1u : 0u) | ((uint32_t)sw().FINISH_EXPERIMENT.get(0) << 8);      // (the experiment field is only ever set in the experiment build)
	bool gather = false;
	if (dg.D1 == 0) {
		hipLaunchKernelGGL(k_l0_starts, dim3(nblk((uint64_t)bins0 + 1)), dim3(256), 0, s, hist0, ntA, bins0, (uint32_t)len, M.lvl_start);
		FS.rec = ws.keyA;
	} else {
		// ---- level 1: every chunk of L1_CHUNK records of a level-0 bucket sorted in LDS, in place; tiles of <= L1_TCH chunks built on the host ----
		std::vector<uint32_t> st(bins0 + 1);
		if (local0) {
			// bucket sizes = row sums of level 0's run table; their prefix is uploaded where the scattering form computes it
			std::vector<unsigned long long> tot(bins0);
			B3M_HIP(hipMemcpyAsync(tot.data(), d_tot, 8ull * bins0, hipMemcpyDeviceToHost, s));
			B3M_HIP(hipStreamSynchronize(s));
			uint64_t acc = 0;
			for (uint32_t b = 0; b < bins0; ++b) { st[b] = (uint32_t)acc; acc += tot[b]; }
			st[bins0] = (uint32_t)acc;
			B3M_REQUIRE(acc == len, "internal: level 0's run table does not add up to the block");
			B3M_HIP(hipMemcpyAsync(M.sub_start, st.data(), 4ull * (bins0 + 1), hipMemcpyHostToDevice, s));
		} else {
			hipLaunchKernelGGL(k_l0_starts, dim3(nblk((uint64_t)bins0 + 1)), dim3(256), 0, s, hist0, ntA, bins0, (uint32_t)len, M.sub_start);
			B3M_HIP(hipMemcpyAsync(st.data(), M.sub_start, 4ull * (bins0 + 1), hipMemcpyDeviceToHost, s));
			B3M_HIP(hipStreamSynchronize(s));
		}
		std::vector<MsdTile> tiles;
		std::vector<L1GTile> gtiles;
		std::vector<uint32_t> tfirst(local0 ? bins0 + 1 : 0);
		std::vector<uint32_t> tab(3ull * bins0);
		uint64_t toff = 0;
		const uint32_t TILE = (uint32_t)L1_CHUNK * L1_TCH;
		for (uint32_t b = 0; b < bins0; ++b) {
			const uint32_t sz = st[b + 1] - st[b];
			const uint32_t nch = (sz + L1_CHUNK - 1) / L1_CHUNK;
			tab[b] = st[b]; tab[bins0 + b] = (uint32_t)toff; tab[2ull * bins0 + b] = nch;
			if (local0) tfirst[b] = b ? tfirst[b - 1] + (st[b] - st[b - 1] + (uint32_t)L1_CHUNK - 1) / (uint32_t)L1_CHUNK + 1 : 0u;
			for (uint32_t j = 0; j * TILE < sz; ++j) {
				if (local0) {
					L1GTile t{}; t.out_off = st[b] + j * TILE; t.count = std::min<uint32_t>(TILE, sz - j * TILE);
					t.hist_base = (uint32_t)(toff + (uint64_t)j * L1_TCH); t.stride = nch; t.d = b; t.tfr = tfirst[b] + j * (uint32_t)L1_TCH; t.slot0 = j * TILE;
					gtiles.push_back(t);
					continue;
				}
				MsdTile t; t.in_off = st[b] + j * TILE; t.count = std::min<uint32_t>(TILE, sz - j * TILE);
				t.hist_base = (uint32_t)(toff + (uint64_t)j * L1_TCH); t.stride = nch;
				tiles.push_back(t);
			}
			toff += ((uint64_t)1 << dg.D1) * nch;
		}
		B3M_REQUIRE(toff <= M.runtab_cap && tiles.size() <= M.tiles_cap && toff < (1ull << 32), "MSD workspace too small");
		B3M_HIP(hipMemcpyAsync(M.bucket_tab, tab.data(), 4 * tab.size(), hipMemcpyHostToDevice, s));
		uint32_t ntB = 0;
		if (local0) {
			if (l1_order) {
				// tile-major (a counting pass over the list, no sort on the host while the GPU waits)
				uint32_t maxj = 0;
				for (const L1GTile &t : gtiles) maxj = std::max(maxj, t.slot0 / TILE);
				std::vector<uint32_t> at(maxj + 2, 0);
				for (const L1GTile &t : gtiles) ++at[t.slot0 / TILE + 1];
				for (uint32_t j = 0; j <= maxj; ++j) at[j + 1] += at[j];
				std::vector<L1GTile> g2(gtiles.size());
				for (const L1GTile &t : gtiles) g2[at[t.slot0 / TILE]++] = t;
				gtiles.swap(g2);
			}
			tfirst[bins0] = tfirst[bins0 - 1] + (st[bins0] - st[bins0 - 1] + (uint32_t)L1_CHUNK - 1) / (uint32_t)L1_CHUNK + 1;
			B3M_REQUIRE(gtiles.size() <= M.gtiles_cap && tfirst[bins0] <= M.tile_fr_cap, "MSD workspace too small");
			B3M_HIP(hipMemcpyAsync(M.gtiles, gtiles.data(), sizeof(L1GTile) * gtiles.size(), hipMemcpyHostToDevice, s));
			B3M_HIP(hipMemcpyAsync(M.tfirst, tfirst.data(), 4ull * (bins0 + 1), hipMemcpyHostToDevice, s));
			B3M_HIP(hipStreamSynchronize(s));      // host vectors must outlive the copies
			{
				Scope sc(tl, KF_SCAN, 8ull * bins0 * ncol0);
				hipLaunchKernelGGL(k_rows0_prefix, dim3(bins0), dim3(256), 0, s, M.tab0, ncol0, M.pre0, M.tfirst, M.tile_fr);
			}
			stage("rows0 prefix");
			L1GArgs Gq{};
			Gq.rec_in = ws.keyA; Gq.rec_out = ws.keyB; Gq.tiles = reinterpret_cast<const L1GTile *>(M.gtiles); Gq.runtab = M.runtab;
			Gq.tab0 = M.tab0; Gq.pre0 = M.pre0; Gq.tile_fr = M.tile_fr; Gq.nchunks0 = ncol0; Gq.chunk0 = (uint32_t)W0A_CHUNK; Gq.D = dg.D1; Gq.shift = 64 - dg.D1;
			Gq.cbase = shard1 ? M.cbase0 : nullptr; Gq.xcd = l1_order ? 1u : 0u;
			ntB = (uint32_t)gtiles.size();
			Scope sc(tl, KF_MSD_SCATTER, 16 * len);
			if (sw().L1_MP.get(0)) hipLaunchKernelGGL((k_l1_gather_mp<L1_TCH>), dim3(ntB), dim3(L1M_THREADS), 0, s, Gq);
			else hipLaunchKernelGGL((k_l1_gather<L1_THREADS, L1_TCH>), dim3(ntB), dim3(L1_THREADS), 0, s, Gq);
			stage("level 1 (gather)");
		} else {
			B3M_HIP(hipMemcpyAsync(M.tiles, tiles.data(), sizeof(MsdTile) * tiles.size(), hipMemcpyHostToDevice, s));
			B3M_HIP(hipStreamSynchronize(s));      // host vectors must outlive the copies
			L1Args Bq{};
			Bq.rec_in = ws.keyA; Bq.rec_out = ws.keyB; Bq.tiles = M.tiles; Bq.runtab = M.runtab; Bq.D = dg.D1; Bq.shift = 64 - dg.D1;
			ntB = (uint32_t)tiles.size();
			Scope sc(tl, KF_MSD_SCATTER, 16 * len);
			hipLaunchKernelGGL((k_l1_local<L1_THREADS, L1_TCH>), dim3(ntB), dim3(L1_THREADS), 0, s, Bq);
		}
		nbF = bins0 << dg.D1;
		B3M_REQUIRE((uint64_t)nbF + 1 <= MSD_TAB_MAX, "MSD level table too small");
		{
			Scope sc(tl, KF_SCAN, 4ull * toff);
			hipLaunchKernelGGL(k_run_totals, dim3(bins0), dim3(256), 0, s, M.runtab, M.runpre, M.bucket_tab, bins0, dg.D1, (uint32_t)len, M.lvl_start);
		}
		stage("level 1 + run totals");
		FS.rec = ws.keyB; FS.runtab = M.runtab; FS.runpre = M.runpre; FS.btab = M.bucket_tab; FS.nb0 = bins0; FS.D1 = dg.D1; FS.chunk = (uint32_t)L1_CHUNK;
		gather = true;
	}
	// ---- finish: every final bucket sorted inside LDS; compulsory traffic = records in, BWT byte (+ SA entry) out ----
	FS.nbF = nbF; FS.nrec = (uint32_t)len;
	const int fbits = REC_KEY_BITS - consumed;            // free key bits of a record entering the finish
	B3M_REQUIRE(fbits >= 1 && fbits <= REC_KEY_BITS && fbits - std::min(fbits, F_BINBITS) <= 16, "internal: free key bits of the finish");
	const uint32_t ovf_cap = (uint32_t)M.overflow_cap;
	SortOut O{sa, d_bwt, start, d_p0, M.overflow, ovf_cap, d_tiebits, pb, SampleOut()};
	if (ws.samples) {
		// (rank,pos) samples leave with the records: only sorts of the whole text (or of a suffix-range shard of it) know global ranks
		// (... or leaf-local ones: a block of a merge tree, sorted as it stands; not the second sort of a block with its overhang, whose ranks count the overhang's suffixes)
		B3M_REQUIRE(own == 0, "internal: in-kernel samples need a block sorted as it stands");
		B3M_REQUIRE(sample_in_kernel_ok(ws.samples->rate, T.n), "internal: sampling rate the kernels cannot test");
		O.smp = *ws.samples; O.smp.rank_off = out.slice_offset;
		out.samples_done = true;
	}
	{
		Scope sc(tl, KF_LDS_SORT, (need_sa ? 13 : 9) * len);
		B3M_HIP(hipMemsetAsync(M.overflow, 0, 8, s));
		B3M_HIP(hipMemsetAsync(M.biglist, 0, 4, s));
		B3M_HIP(hipMemsetAsync(M.hugelist, 0, 4, s));
		B3M_HIP(hipMemsetAsync(d_tiebits, 0, 4 * ((len + 31) / 32 + 1), s));
		B3M_HIP(hipMemsetAsync(M.midlist, 0, 4, s));
		const uint32_t gmid = (uint32_t)std::min<uint64_t>(len / F_CAP + 1, 1024);
		// (buckets up to F_HUGE2 records go through level 2 where there is one: tiles in parallel; without it one workgroup partitions a listed bucket: F_HUGE)
		const uint32_t huge_min = (gather && !sw().NO_LEVEL2.set) ? F_HUGE2 : F_HUGE;
		const FinishLists hl[2] = {{M.midlist, (uint32_t)M.mid_cap, M.biglist, (uint32_t)M.big_cap, M.hugelist, (uint32_t)M.huge_cap, huge_min},
		                           {nullptr, 0u, M.biglist, (uint32_t)M.big_cap, M.hugelist, (uint32_t)M.huge_cap, huge_min}};
		FinishLists *dl = reinterpret_cast<FinishLists *>(M.lists);       // [0]: first launch (lists mid / big / huge), [1]: the mid launch
		B3M_HIP(hipMemcpyAsync(dl, hl, sizeof hl, hipMemcpyHostToDevice, s));
		B3M_HIP(hipStreamSynchronize(s));                                 // hl lives on this stack frame
		// The first launch goes out in slices of whole level-0 buckets (one slice unless the caller streams the result away:
		// ws.slice_hook), each followed by the tie fix of its records: after that a slice of the BWT is final unless one
		// of its buckets was listed (out.late_writes).
		const uint32_t nsl = (ws.slice_hook && ws.slice_hook->nslices > 1 && dg.D1 > 0) ? (uint32_t)std::min<uint64_t>(ws.slice_hook->nslices, bins0) : 1u;
		std::vector<uint32_t> lvl0;                                       // starts of the level-0 buckets (rank order) at the slice borders
		if (nsl > 1) {
			lvl0.resize(nsl + 1);
			std::vector<uint32_t> hs(bins0 + 1);
			B3M_HIP(hipMemcpyAsync(hs.data(), M.sub_start, 4ull * (bins0 + 1), hipMemcpyDeviceToHost, s));
			B3M_HIP(hipStreamSynchronize(s));
			for (uint32_t i = 0; i <= nsl; ++i) lvl0[i] = hs[(uint64_t)bins0 * i / nsl];
		}
		const int no_fast = (int)sw().NO_FAST_FINISH.get(0);      // measurement switch
		for (uint32_t sl = 0; sl < nsl; ++sl) {
			const uint32_t l0a = (uint32_t)((uint64_t)bins0 * sl / nsl), l0b = (uint32_t)((uint64_t)bins0 * (sl + 1) / nsl);
			const uint32_t bfa = nsl > 1 ? l0a << dg.D1 : 0u, bfb = nsl > 1 ? l0b << dg.D1 : nbF;
			const uint64_t ra = nsl > 1 ? lvl0[sl] : 0, rb = nsl > 1 ? lvl0[sl + 1] : len;
			if (bfb == bfa) continue;
			if (gather && !no_fast && fbits >= F_BINBITS && fbits - F_BINBITS <= 12) {
				// sorts that keep no suffix array take k_finish_v3 (32-bit positions; sampling off or sparse enough for its list of special records).
				// B3M_FINISH_V3 (measurement switch): 0 = k_finish_fast<false> instead, 2 = k_finish_v3 held at two workgroups per CU
				int v3 = (int)sw().FINISH_V3.get(1);
				if (big || (O.smp.pairs && O.smp.rate < V3_MIN_RATE)) v3 = 0;
				if (need_sa) hipLaunchKernelGGL((k_finish_fast<true>), dim3(bfb - bfa), dim3(512), 0, s, FS, fbits, bfa, dl, O);
				else if (v3 == 2) hipLaunchKernelGGL(k_finish_v3<2>, dim3(bfb - bfa), dim3(512), 0, s, FS, fbits, bfa, dl, O);
				else if (v3 == 4) hipLaunchKernelGGL((k_finish_v3<3, 1024>), dim3(bfb - bfa), dim3(1024), 0, s, FS, fbits, bfa, dl, O);
#ifdef B3M_PHASE_EXPERIMENT
				else if (sw().FINISH_NOMAP.on()) hipLaunchKernelGGL((k_finish_v3<3, 512, true>), dim3(bfb - bfa), dim3(512), 0, s, FS, fbits, bfa, dl, O);      // timing only, WRONG results
#endif
				else if (v3) hipLaunchKernelGGL(k_finish_v3<3>, dim3(bfb - bfa), dim3(512), 0, s, FS, fbits, bfa, dl, O);
				else hipLaunchKernelGGL((k_finish_fast<false>), dim3(bfb - bfa), dim3(512), 0, s, FS, fbits, bfa, dl, O);
			} else if (gather) {
				if (need_sa) hipLaunchKernelGGL((k_finish_rows<true, 512>), dim3(bfb - bfa), dim3(512), 0, s, FS, fbits, bfa, dl, O);
				else hipLaunchKernelGGL((k_finish_rows<false, 512>), dim3(bfb - bfa), dim3(512), 0, s, FS, fbits, bfa, dl, O);
			} else {
				if (need_sa) hipLaunchKernelGGL((k_finish_lds<true, 512>), dim3(bfb - bfa), dim3(512), 0, s, FS, fbits, bfa, (const uint32_t *)nullptr, 0u, dl, O);
				else hipLaunchKernelGGL((k_finish_lds<false, 512>), dim3(bfb - bfa), dim3(512), 0, s, FS, fbits, bfa, (const uint32_t *)nullptr, 0u, dl, O);
			}
			if (nsl > 1) {
				if (rb > ra) hipLaunchKernelGGL(k_tie_fix, dim3(nblk((rb + 31) / 32 - ra / 32)), dim3(256), 0, s, T, ws.t[0], len, ra, rb, eq_syms, sa, d_bwt, start, d_p0, O.smp, TieExtras{});
				ws.slice_hook->fn(ws.slice_hook->ctx, ra, rb);
			}
		}
		stage("first finish launch");
		// k_bucket_finish and its follow-up passes: a listed bucket gets an 8-bit partition by one workgroup, whose waves sort the sub-buckets up to LS_CAP
		// records; a sub-bucket above that is listed for the NEXT pass (the next 8 key bits), until the key bits are used up: a repeat family's k-mers end
		// as hot bins / whole-key sub-buckets at the depth the records know, not as whole sub-buckets at the depth of their bucket.  cons: key bits consumed
		// in front of the first pass, bits: key bits left to it.  Passes on empty lists return at once (the lists are read on the device: no host round trip).
		auto bucket_finish_chain = [&](const uint64_t *rin, uint64_t *rtmp, const uint32_t *bst, const uint32_t *ben, const uint32_t *list, uint32_t list_cap,
		                               int cons, int bits, uint32_t grid) {
			const int kb_main = consumed + std::min(fbits, F_BINBITS);      // key bits a bin of the main finish knows: what "one bin" promises the list rounds
			const int no_chain = (int)sw().NO_FINISH_CHAIN.get(0);      // measurement switch: one pass, as before
			for (int pass = 0; pass < 3; ++pass) {
				const int D = pass == 0 ? 8 : std::min(8, bits);
				const int sortbits = bits - D, shift = 64 - cons - D;
				const bool more = sortbits > 0 && pass < 2 && !no_chain && M.sub_cap;
				const int set = pass & 1;
				SubNext nx{nullptr, nullptr, nullptr, 0u};
				if (more) { B3M_HIP(hipMemsetAsync(M.sub_list[set], 0, 4, s)); nx = SubNext{M.sub_st[set], M.sub_en[set], M.sub_list[set], (uint32_t)M.sub_cap}; }
				const int known_sub = cons + std::min(std::max(bits, 0), D), known_bin = cons + std::min(std::max(bits, 0), D + LS_BINBITS);
				const int flags = no_chain ? 0 : ((known_bin >= kb_main ? 1 : 0) | (known_sub >= kb_main ? 2 : 0));
				if (need_sa) hipLaunchKernelGGL(k_bucket_finish<true>, dim3(grid), dim3(MSD_THREADS), 0, s, rin, rtmp, bst, ben, list, list_cap, D, shift, sortbits, O, nx, flags);
				else hipLaunchKernelGGL(k_bucket_finish<false>, dim3(grid), dim3(MSD_THREADS), 0, s, rin, rtmp, bst, ben, list, list_cap, D, shift, sortbits, O, nx, flags);
				if (!more) break;
				const uint64_t *t = rin; rin = rtmp; rtmp = const_cast<uint64_t *>(t);
				bst = M.sub_st[set]; ben = M.sub_en[set]; list = M.sub_list[set]; list_cap = (uint32_t)M.sub_cap; grid = 2048;
				cons += D; bits = sortbits;
			}
		};
		uint32_t lc1[3] = {0, 0, 0};                                      // buckets the first launch listed: mid, big, huge
		if (gather) {
			// what the gathering finish could not take is copied together into the buffer level 1 has read; the paths below read it there.
			// A fixed grid strides over the lists: empty lists (uniform input) cost a launch and nothing else.
			hipLaunchKernelGGL(k_compact_listed, dim3(2048), dim3(256), 0, s, FS, ws.keyA, M.midlist, (uint32_t)M.mid_cap, M.biglist, (uint32_t)M.big_cap, M.hugelist, (uint32_t)M.huge_cap);
			FS.rec = ws.keyA; cur = ws.keyA; other = ws.keyB;
			stage("compact listed");
			B3M_HIP(hipMemcpyAsync(&lc1[0], M.midlist, 4, hipMemcpyDeviceToHost, s));
			B3M_HIP(hipMemcpyAsync(&lc1[1], M.biglist, 4, hipMemcpyDeviceToHost, s));
			B3M_HIP(hipMemcpyAsync(&lc1[2], M.hugelist, 4, hipMemcpyDeviceToHost, s));
			B3M_HIP(hipStreamSynchronize(s));
			listed_first = (lc1[0] | lc1[1] | lc1[2]) != 0;
		}
		if (!gather || lc1[0]) {
		if (need_sa) hipLaunchKernelGGL((k_finish_lds<true, 1024>), dim3(gmid), dim3(1024), 0, s, FS, fbits, 0u, (const uint32_t *)M.midlist, (uint32_t)M.mid_cap, dl + 1, O);
		else hipLaunchKernelGGL((k_finish_lds<false, 1024>), dim3(gmid), dim3(1024), 0, s, FS, fbits, 0u, (const uint32_t *)M.midlist, (uint32_t)M.mid_cap, dl + 1, O);
		stage("mid finish");
		}
		// buckets above the LDS capacity (none on uniform input): one more 8-bit partition through HBM.
		if (len > huge_min && (!gather || lc1[2])) hipLaunchKernelGGL(k_huge_spill, dim3(2048), dim3(256), 0, s, cur, (const uint32_t *)M.lvl_start, (const uint32_t *)nullptr, M.hugelist, (uint32_t)M.huge_cap, sa, pb);
		// LEVEL 2 (skewed base composition, low-complexity text): the listed buckets above twice the finish's capacity are to level 1's kernels what
		// the level-0 buckets were: tiles of 16 Ki records of ONE bucket sorted by their next 8 bits in LDS (k_l1_local: a stream copy from where
		// k_compact_listed put them), a run table, and the gathering finish on 256 sub-buckets per bucket -- every tile and every sub-bucket a
		// workgroup of its own.  (Before: one workgroup per listed bucket partitioned it alone, chunk by chunk, and sub-buckets above 512 records
		// went to the list-based rounds: 80 % A+T cost 10x the uniform text.)  What is still too large after it takes the old paths.
		bool lvl2_done = false; (void)lvl2_done;
		const int no_lvl2 = (int)sw().NO_LEVEL2.get(0);      // measurement switch
		uint32_t nbig1 = lc1[1];                                          // listed buckets left to k_bucket_finish
		if (gather && lc1[1] && !no_lvl2 && fbits > 8 && lc1[1] <= M.big_cap) {
			std::vector<uint32_t> all(lc1[1]), ids, small, ls((size_t)nbF + 1);
			B3M_HIP(hipMemcpyAsync(all.data(), M.biglist + 1, 4ull * lc1[1], hipMemcpyDeviceToHost, s));
			B3M_HIP(hipMemcpyAsync(ls.data(), M.lvl_start, 4ull * (nbF + 1), hipMemcpyDeviceToHost, s));
			B3M_HIP(hipStreamSynchronize(s));
			const uint32_t lvl2_min = (uint32_t)sw().LVL2_MIN.get(LVL2_MIN);      // test switch: small inputs through level 2
			// a bucket's fan-out is chosen from its own size: the sub-buckets are to average LVL2_AVG records (half a finish workgroup's capacity: on
			// skewed text the heaviest is a few times the average), so a class of buckets per digit width D2 = 1 .. 8, one after the other.  (With 8 bits
			// for all of them a 65 K bucket made 256 sub-buckets of 256 records: 1.3 M workgroups, 24 ms of the 80 a Gbase at 80 % A+T took.)
			const uint32_t lvl2_avg = sw().LVL2_AVG.set ? std::min<uint32_t>(LVL2_AVG, (uint32_t)sw().LVL2_AVG.v) : LVL2_AVG;      // (measurement switch: smaller only, the tables are sized for LVL2_AVG)
			std::vector<uint32_t> cls[MSD_MAXBITS + 1];
			const int d2_max = std::min(MSD_MAXBITS, fbits - 1);
			for (uint32_t b : all) {
				const uint32_t sz = ls[b + 1] - ls[b];
				if (sz <= lvl2_min) { small.push_back(b); continue; }
				int d2 = 1; while (d2 < d2_max && (sz >> d2) > lvl2_avg) ++d2;
				cls[d2].push_back(b);
			}
			const uint32_t TILE = (uint32_t)L1_CHUNK * L1_TCH;
			// classes the tables have no room for stay with k_bucket_finish
			struct Cls { int d2; std::vector<MsdTile> tiles; std::vector<uint32_t> tab; uint64_t toff; uint32_t nb; };
			std::vector<Cls> run;
			for (int d2 = 1; d2 <= MSD_MAXBITS; ++d2) {
				const uint32_t nb2 = (uint32_t)cls[d2].size();
				if (!nb2) continue;
				Cls c; c.d2 = d2; c.nb = nb2; c.tab.resize(3ull * nb2); c.toff = 0;
				for (uint32_t i = 0; i < nb2; ++i) {
					const uint32_t b = cls[d2][i], st = ls[b], sz = ls[b + 1] - st, nch = (sz + L1_CHUNK - 1) / L1_CHUNK;
					c.tab[i] = st; c.tab[nb2 + i] = (uint32_t)c.toff; c.tab[2ull * nb2 + i] = nch;
					for (uint32_t j = 0; j * TILE < sz; ++j) {
						MsdTile t; t.in_off = st + j * TILE; t.count = std::min<uint32_t>(TILE, sz - j * TILE);
						t.hist_base = (uint32_t)(c.toff + (uint64_t)j * L1_TCH); t.stride = nch;
						c.tiles.push_back(t);
					}
					c.toff += ((uint64_t)1 << d2) * nch;
				}
				const uint64_t nsub = (uint64_t)nb2 << d2;
				if (nb2 <= M.lvl2_buckets && 2 * nsub + 3ull * nb2 + 16 <= M.lvl2_words && nsub <= M.list2_cap && c.toff <= M.runtab_cap && c.tiles.size() <= M.tiles_cap && c.toff < (1ull << 32))
					run.push_back(std::move(c));
				else small.insert(small.end(), cls[d2].begin(), cls[d2].end());
			}
			if (!run.empty()) {
				// the listed buckets level 2 does not take stay on the list of k_bucket_finish
				nbig1 = (uint32_t)small.size();
				B3M_HIP(hipMemcpyAsync(M.biglist, &nbig1, 4, hipMemcpyHostToDevice, s));
				if (nbig1) B3M_HIP(hipMemcpyAsync(M.biglist + 1, small.data(), 4ull * nbig1, hipMemcpyHostToDevice, s));
				B3M_HIP(hipStreamSynchronize(s));
			}
			for (Cls &c : run) {
				const int d2 = c.d2;
				const uint32_t nb2 = c.nb, nbF2 = nb2 << d2;
				uint32_t *d_st2 = M.lvl2, *d_en2 = M.lvl2 + nbF2, *d_tab2 = M.lvl2 + 2ull * nbF2;
				const FinishLists hl2[2] = {{M.mid2, (uint32_t)M.list2_cap, M.big2, (uint32_t)M.list2_cap, M.huge2, (uint32_t)M.huge_cap, F_HUGE},
				                            {nullptr, 0u, M.big2, (uint32_t)M.list2_cap, M.huge2, (uint32_t)M.huge_cap, F_HUGE}};
				B3M_HIP(hipMemcpyAsync(dl + 2, hl2, sizeof hl2, hipMemcpyHostToDevice, s));
				B3M_HIP(hipMemcpyAsync(M.tiles, c.tiles.data(), sizeof(MsdTile) * c.tiles.size(), hipMemcpyHostToDevice, s));
				B3M_HIP(hipMemcpyAsync(d_tab2, c.tab.data(), 4 * c.tab.size(), hipMemcpyHostToDevice, s));
				B3M_HIP(hipMemsetAsync(M.mid2, 0, 4, s)); B3M_HIP(hipMemsetAsync(M.big2, 0, 4, s)); B3M_HIP(hipMemsetAsync(M.huge2, 0, 4, s));
				B3M_HIP(hipStreamSynchronize(s));                         // host vectors must outlive the copies
				L1Args Bq{};
				Bq.rec_in = ws.keyA; Bq.rec_out = ws.keyB; Bq.tiles = M.tiles; Bq.runtab = M.runtab; Bq.D = d2; Bq.shift = 64 - consumed - d2;
				hipLaunchKernelGGL((k_l1_local<L1_THREADS, L1_TCH>), dim3((uint32_t)c.tiles.size()), dim3(L1_THREADS), 0, s, Bq);
				hipLaunchKernelGGL(k_run_totals, dim3(nb2), dim3(256), 0, s, M.runtab, M.runpre, d_tab2, nb2, d2, (uint32_t)len, d_st2, d_en2);
				stage("level 2: partition");
				FinishSrc F2{};
				F2.rec = ws.keyB; F2.bstart = d_st2; F2.bend = d_en2; F2.runtab = M.runtab; F2.runpre = M.runpre; F2.btab = d_tab2; F2.nb0 = nb2; F2.D1 = d2;
				F2.chunk = (uint32_t)L1_CHUNK; F2.nbF = nbF2; F2.nrec = (uint32_t)len;
				const int fbits2 = fbits - d2;
				if (!no_fast && fbits2 >= F_BINBITS && fbits2 - F_BINBITS <= 12) {
					int v3 = (int)sw().FINISH_V3.get(1);
					if (big || (O.smp.pairs && O.smp.rate < V3_MIN_RATE)) v3 = 0;
					if (need_sa) hipLaunchKernelGGL((k_finish_fast<true>), dim3(nbF2), dim3(512), 0, s, F2, fbits2, 0u, dl + 2, O);
					else if (v3) hipLaunchKernelGGL(k_finish_v3<3>, dim3(nbF2), dim3(512), 0, s, F2, fbits2, 0u, dl + 2, O);
					else hipLaunchKernelGGL((k_finish_fast<false>), dim3(nbF2), dim3(512), 0, s, F2, fbits2, 0u, dl + 2, O);
				} else {
					if (need_sa) hipLaunchKernelGGL((k_finish_rows<true, 512>), dim3(nbF2), dim3(512), 0, s, F2, fbits2, 0u, dl + 2, O);
					else hipLaunchKernelGGL((k_finish_rows<false, 512>), dim3(nbF2), dim3(512), 0, s, F2, fbits2, 0u, dl + 2, O);
				}
				// what level 2 could not take either: copied together (into the buffer it has read), then the 1024-thread finish, one more
				// partition by one workgroup per bucket, or the overflow list
				hipLaunchKernelGGL(k_compact_listed, dim3(2048), dim3(256), 0, s, F2, ws.keyA, M.mid2, (uint32_t)M.list2_cap, M.big2, (uint32_t)M.list2_cap, M.huge2, (uint32_t)M.huge_cap);
				F2.rec = ws.keyA;
				const uint32_t gmid2 = (uint32_t)std::min<uint64_t>(nbF2, 1024);
				if (need_sa) hipLaunchKernelGGL((k_finish_lds<true, 1024>), dim3(gmid2), dim3(1024), 0, s, F2, fbits2, 0u, (const uint32_t *)M.mid2, (uint32_t)M.list2_cap, dl + 3, O);
				else hipLaunchKernelGGL((k_finish_lds<false, 1024>), dim3(gmid2), dim3(1024), 0, s, F2, fbits2, 0u, (const uint32_t *)M.mid2, (uint32_t)M.list2_cap, dl + 3, O);
				hipLaunchKernelGGL(k_huge_spill, dim3(2048), dim3(256), 0, s, (const uint64_t *)ws.keyA, (const uint32_t *)d_st2, (const uint32_t *)d_en2, M.huge2, (uint32_t)M.huge_cap, sa, pb);
				const uint32_t gbig2 = (uint32_t)std::min<uint64_t>(nbF2, big_grid());
				if (fbits2 > 8) {
					bucket_finish_chain((const uint64_t *)ws.keyA, ws.keyB, (const uint32_t *)d_st2, (const uint32_t *)d_en2, M.big2, (uint32_t)M.list2_cap, consumed + d2, fbits2, gbig2);
				} else {
					hipLaunchKernelGGL(k_list_to_overflow, dim3(64), dim3(256), 0, s, (const uint64_t *)ws.keyA, (const uint32_t *)d_st2, (const uint32_t *)d_en2, M.big2, (uint32_t)M.list2_cap, sa, pb, O);
				}
				stage("level 2: finish");
				lvl2_ran = true;
				// a list that ran over would have dropped sub-buckets silently
				uint32_t l2[3] = {0, 0, 0};
				B3M_HIP(hipMemcpyAsync(&l2[0], M.mid2, 4, hipMemcpyDeviceToHost, s)); B3M_HIP(hipMemcpyAsync(&l2[1], M.big2, 4, hipMemcpyDeviceToHost, s)); B3M_HIP(hipMemcpyAsync(&l2[2], M.huge2, 4, hipMemcpyDeviceToHost, s));
				B3M_HIP(hipStreamSynchronize(s));
				B3M_REQUIRE(l2[0] <= M.list2_cap && l2[1] <= M.list2_cap && l2[2] <= M.huge_cap, "internal: a bucket list of level 2 ran over");
				if (sw().DEBUG.set)
					fprintf(stderr, "[b3m] level 2: %u buckets by %d bits, %zu tiles, fbits %d; listed again: mid %u big %u huge %u\n", nb2, d2, c.tiles.size(), fbits2, l2[0], l2[1], l2[2]);
			}
		}
		if (!gather || nbig1) {
		// one workgroup per listed bucket: the listed buckets up to LVL2_MIN records (and all of them where there is no level 2: small blocks without a level 1)
		const uint32_t gbig = (uint32_t)std::min<uint64_t>(len / F_CAP + 1, big_grid());
		bucket_finish_chain(cur, other, (const uint32_t *)M.lvl_start, (const uint32_t *)nullptr, M.biglist, (uint32_t)M.big_cap, consumed, fbits, gbig);
		}
		stage("big / huge finish");
		sliced = nsl > 1;
	}
	if (sliced) {
		// every slice has had its tie fix; only buckets that were listed (skewed input) still have open ties, and they were
		// written after their slice was handed to the hook
		uint32_t lc[3] = {0, 0, 0};
		B3M_HIP(hipMemcpyAsync(&lc[0], M.midlist, 4, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipMemcpyAsync(&lc[1], M.biglist, 4, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipMemcpyAsync(&lc[2], M.hugelist, 4, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipStreamSynchronize(s));
		// (the counts of the first launch: level 2 takes its buckets OFF the big list, which then reads 0 although they were finished -- and their key ties
		// marked -- after the slices had left; a host-to-host run of such a text then skipped the tie fix of those buckets: pairs of neighbouring BWT bytes in
		// either order, found by tools/fuzz_large.py in round 3)
		out.late_writes = listed_first || (lc[0] | lc[1] | lc[2]) != 0;
	}
	if (!sliced || out.late_writes) {
		Scope sc(tl, KF_TIES, len / 8);
		const int tie_two_step = (int)sw().TIE_TWO_STEP.get(1);      // measurement switch
		// (sequential shards of a plan that keeps a list of tied suffixes: what the direct comparisons place among long repeats is listed too, TieExtras)
		TieExtras xtra{};
		if (nshards > 1 && start == 0 && len_in == T.n && ws.no_defer && ws.sparse_room > 0 && ws.extras.cnt) { xtra = ws.extras; xtra.keys = 4; }      // (4 = R0, the key rounds a shard runs before it hands over)
		if (tie_two_step && len >= (1ull << 22) && len < (1ull << 32)) {
			// both record buffers are free by now: the list of group starts goes into one of them
			uint32_t *glist = reinterpret_cast<uint32_t *>(ws.keyB), *gcnt = reinterpret_cast<uint32_t *>(M.lists) + 60;
			const uint32_t gcap = (uint32_t)std::min<uint64_t>(2 * ws.cap, 0xFFFFFFF0ull);
			B3M_HIP(hipMemsetAsync(gcnt, 0, 4, s));
			hipLaunchKernelGGL(k_tie_collect, dim3(nblk((len + 31) / 32, 256 * TIE_WPT)), dim3(256), 0, s, ws.t[0], len, glist, gcap, gcnt);
			hipLaunchKernelGGL(k_tie_fix_list, dim3(8192), dim3(256), 0, s, T, ws.t[0], len, glist, gcnt, gcap, eq_syms, sa, d_bwt, start, d_p0, O.smp, xtra);
			if (sw().DEBUG.set) {
				uint32_t ng = 0; B3M_HIP(hipMemcpyAsync(&ng, gcnt, 4, hipMemcpyDeviceToHost, s)); B3M_HIP(hipStreamSynchronize(s));
				fprintf(stderr, "[b3m] tie fix: %u groups of records with equal key fields\n", ng);
			}
		} else
			hipLaunchKernelGGL(k_tie_fix, dim3(nblk((len + 31) / 32)), dim3(256), 0, s, T, ws.t[0], len, (uint64_t)0, len, eq_syms, sa, d_bwt, start, d_p0, O.smp, xtra);
	}
	stage("tie fix");
	uint32_t novw[2] = {0, 0}, nlist[3] = {0, 0, 0};
	B3M_HIP(hipMemcpyAsync(novw, M.overflow, 8, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipMemcpyAsync(&nlist[0], M.midlist, 4, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipMemcpyAsync(&nlist[1], M.biglist, 4, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipMemcpyAsync(&nlist[2], M.hugelist, 4, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	const uint32_t nov = novw[0];
	// what every range of the overflow list shares: the bits of its bucket, and those of its bin when every range is one bin of a finish (a repeated
	// 16-mer at the usual geometry).  The rounds on the list start behind it (in the record sort's reading of the text: key_at_off)
	const uint64_t ovf_syms = (uint64_t)(D0e + consumed + ((novw[1] & 1u) ? 0 : std::min(fbits, F_BINBITS))) / 2;
	if (sw().DEBUG.set) fprintf(stderr, "[b3m] block_sort_msd len %llu D0 %d D1 %d: listed mid %u big %u huge %u (caps %llu %llu %llu), overflow ranges %u\n",
	                                 (unsigned long long)len, D0e, dg.D1, nlist[0], nlist[1], nlist[2], (unsigned long long)M.mid_cap, (unsigned long long)M.big_cap, (unsigned long long)M.huge_cap, nov);
	// a list that ran over would have dropped buckets silently
	B3M_REQUIRE(nlist[0] <= M.mid_cap && nlist[1] <= M.big_cap && nlist[2] <= M.huge_cap, "internal: a bucket list of the MSD finish ran over");
	(void)lvl2_ran;
	out.msd_overflow_buckets = nov;
	if (nov) out.late_writes = true;
	bool rerun_with_sa = false;
	if (nov) {
		// ---- fallback: the overflow list (hot bins, oversized sub-buckets, huge buckets) goes through the list-based rounds ----
		Scope sc(tl, KF_TIES, 0);
		B3M_REQUIRE(nov <= ovf_cap, "too many overflow ranges for the workspace");
		// order the (start, size) pairs by start and turn the sizes into list offsets, all on the device
		// (a repeat-rich genome lists millions of bins); the record buffers are free by now
		uint64_t *pk = reinterpret_cast<uint64_t *>(M.overflow + 2), *rk = nullptr;
		radix_sort_u64(pk, nullptr, ws.keyA, nullptr, nov, 32, ws.radix, s, nullptr, &rk, nullptr);
		if (rk == pk) { B3M_HIP(hipMemcpyAsync(ws.keyA, pk, 8ull * nov, hipMemcpyDeviceToDevice, s)); rk = ws.keyA; }
		if (sw().DEBUG.get(0) >= 3) {
			std::vector<uint64_t> hp(nov);
			B3M_HIP(hipMemcpyAsync(hp.data(), rk, 8ull * nov, hipMemcpyDeviceToHost, s)); B3M_HIP(hipStreamSynchronize(s));
			uint64_t nb = 0, eb = 0, no = 0, eo = 0, big_o = 0;
			for (uint64_t v : hp) { const uint32_t sz = (uint32_t)(v >> 32); if ((sz & OVF_ONE_BIN) && !(novw[1] & 2u)) { ++nb; eb += sz & ~OVF_ONE_BIN; } else { ++no; eo += sz; big_o = std::max<uint64_t>(big_o, sz); } }
			fprintf(stderr, "[b3m] overflow ranges: %llu single bins with %llu records; %llu other ranges with %llu records (largest %llu)\n", (unsigned long long)nb, (unsigned long long)eb,
			        (unsigned long long)no, (unsigned long long)eo, (unsigned long long)big_o);
		}
		// the overflow table is dead now: reuse it for (off, b0)
		uint32_t *d_off = M.overflow, *d_b0 = M.overflow + nov + 1;
		const uint32_t size_mask = (novw[1] & 2u) ? 0xFFFFFFFFu : ~OVF_ONE_BIN;
		hipLaunchKernelGGL(k_ovf_split, dim3(nblk((uint64_t)nov + 1)), dim3(256), 0, s, rk, nov, d_off, d_b0, size_mask);
		exclusive_sum_u32(d_off, (uint64_t)nov + 1, ws.scan_ws, s);
		uint32_t total32 = 0;
		B3M_HIP(hipMemcpyAsync(&total32, d_off + nov, 4, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipStreamSynchronize(s));
		const uint64_t total = total32;
		blocksort_ensure_ties(ws, total);                      // (lazy in one-block plans: uniform text never comes here; texts of 2^32 symbols and more: larger buffers on demand)
		// texts of 2^32 symbols and more: list buffers of tie_cap entries (the tie bitmap sits in t[0]), positions in two arrays
		uint32_t *Bf[6] = {ws.t[0], ws.t[1], ws.idxB, ws.t[2], ws.t[3], ws.t[4]};
		if (big) {
			Bf[0] = ws.t[1]; Bf[1] = ws.t[2]; Bf[2] = ws.t[3]; Bf[3] = ws.t[4]; Bf[4] = ws.t[5]; Bf[5] = ws.idxB;
			B3M_REQUIRE(ws.t[5] != nullptr && ws.tie_hi != nullptr, "internal: workspace has no list buffers for positions above 2^32");
			B3M_REQUIRE(total <= ws.tie_cap, "repeat-rich input of 2^32 symbols and more: " + std::to_string(total) + " suffixes are left to the list-based rounds, whose buffers hold " +
			            std::to_string(ws.tie_cap) + ": use more shards");
		}
		const bool by_range = total / nov >= 16 && !sw().FILL_BY_SLOT.on();      // (few slots per range: one thread per slot and a search keep the lanes busy)
		auto fill_list = [&](uint32_t *U, uint32_t *gid) {
			if (by_range) hipLaunchKernelGGL(k_fill_overflow_ranges, dim3(std::min<uint32_t>((nov + 3) / 4, 32768u)), dim3(256), 0, s, (const uint32_t *)d_off, (const uint32_t *)d_b0, nov, U, gid);
			else hipLaunchKernelGGL(k_fill_overflow_list, dim3(nblk(total)), dim3(256), 0, s, d_off, d_b0, nov, total, U, gid);
		};
		fill_list(Bf[0], Bf[1]);
		// the list of slots is needed twice more (ranks of everything ever listed; the symbols of the finished slots): a copy behind the rounds' keys in the
		// first record buffer (the rounds use total 64-bit keys of it) instead of two more binary searches per slot (3.1 ms each for the 332 M of a genome-like text)
		uint32_t *U_keep = (!big && 12 * total + 64 <= 8 * ws.cap) ? reinterpret_cast<uint32_t *>(ws.keyA + total + 4) : nullptr;
		if (U_keep) B3M_HIP(hipMemcpyAsync(U_keep, Bf[0], 4 * total, hipMemcpyDeviceToDevice, s));
		auto slots_again = [&](uint32_t *dst) {
			if (U_keep) B3M_HIP(hipMemcpyAsync(dst, U_keep, 4 * total, hipMemcpyDeviceToDevice, s));
			else fill_list(dst, nullptr);
		};
		stage("overflow list filled");
		B3M_HIP(hipStreamSynchronize(s));
		uint64_t left = 0;
		out.tie_elements = total;
		// A block that is the whole text can rank any suffix: after a few key-extension rounds (enough for
		// everything but long repeats) the rest is finished by prefix doubling, log2(longest repeat) rounds
		// instead of (longest repeat)/21.  Doubling reads the rank of EVERY suffix, i.e. the complete
		// suffix array: a run that was asked for the BWT only starts over with the array kept.
		const bool whole = start == 0 && len_in == T.n && nshards <= 1;
		// a suffix-range shard of the whole text cannot rank the suffixes of the other shards: after the same few key-extension
		// rounds it hands what is left to the caller, who finishes it by prefix doubling over ALL shards' suffix arrays
		// (b3m_plan_tie_export / b3m_plan_tie_finish): O(log longest repeat) rounds there too
		const bool defer = nshards > 1 && start == 0 && len_in == T.n && !ws.no_defer;
		const uint64_t R0 = 4;
		const uint64_t off0 = sw().TIE_FROM_0.set ? 0 : ovf_syms;      // (switch: the rounds start at symbol 0, as they did before)
		// a block of a merge tree: the same few rounds, then prefix doubling inside the block and its overhang (tie_rounds_domain):
		// the first time round the caller is asked to sort the block again with the overhang behind it (out.want_ext)
		const bool tree = !whole && !defer && !no_tree_doubling() && !(nshards > 1 && ws.no_defer);      // (a sequential shard holds a rank range of the text, not a block: key extension to the end)
		// 36-bit positions: key extension until nothing is left (an inverse suffix array of 2^32 and more entries does not fit beside the sort: no doubling,
		// O(longest repeat / 21) rounds)
		// ... unless the plan that runs the shards one after the other keeps a list of what they leave tied (ws.sparse_room): then the shard stops after the
		// usual few rounds, as every other sort does, and the plan finishes the union of those lists by prefix doubling ON THE LIST (engine.hip: sparse_tie_finish)
		const bool sdefer = nshards > 1 && start == 0 && len_in == T.n && ws.no_defer && ws.sparse_room > 0;
		uint64_t h0_whole = 1; while (2 * h0_whole <= R0 * (uint64_t)key_symbols(T.type)) h0_whole *= 2;
		uint32_t *U1_keep = U_keep ? U_keep + total + 16 : nullptr; uint64_t u1 = 0;
		const bool keep_first = whole && !big && U_keep && 16 * total + 256 <= 8 * ws.cap && (R0 - 1) * (uint64_t)key_symbols(T.type) >= h0_whole &&
		                        !sw().ISA_EVER_LISTED.on();      // (measurement switch: ranks for every slot ever listed, as before)
		if (big) out.tie_rounds = tie_rounds(T, sa.lo, total, 0, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, len, s, &left, sdefer ? R0 : 0, nullptr, 0, sa.hi, ws.tie_hi, off0, false, true);
		else if (keep_first) {
			// (a whole-text block) the list as the first round leaves it is kept: the doubling rounds behind the R0 key rounds ask for the ranks of suffixes that agree
			// with another one on (R0 - 1) K symbols or more, i.e. of suffixes on THAT list -- ranks for those 163 M instead of all 332 M ever listed (genome-like text at C3 size)
			out.tie_rounds = tie_rounds(T, sa.lo, total, 0, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, len, s, &left, 1, nullptr, 0, nullptr, nullptr, off0, false, true);
			if (left) {
				u1 = left;
				B3M_HIP(hipMemcpyAsync(U1_keep, Bf[0], 4 * u1, hipMemcpyDeviceToDevice, s));
				out.tie_rounds += tie_rounds(T, sa.lo, left, 1, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, len, s, &left, R0 - 1, nullptr, 0, nullptr, nullptr, off0, false, true);
			}
		} else
		out.tie_rounds = tie_rounds(T, sa.lo, total, 0, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, len, s, &left, (whole || defer || tree || sdefer) ? R0 : 0, nullptr, 0, nullptr, nullptr, off0, false, true);
		if (sdefer && left && out.tie_rounds >= R0) {
			if (left <= ws.sparse_room) { out.tie_left = left; out.tie_U = Bf[0]; out.tie_gid = Bf[1]; out.tie_depth = off0 + R0 * (uint64_t)key_symbols(T.type); }
			else {         // no room on the plan's list: to the end here, as without it
				out.tie_refused = true;
				out.tie_rounds += tie_rounds(T, sa.lo, left, R0, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, len, s, &left, 0, nullptr, 0, big ? sa.hi : nullptr, big ? ws.tie_hi : nullptr, off0, false, true);
			}
		}
		if (big || sdefer) { /* finished above, or handed over */ }
		else if (left && tree && out.tie_rounds >= R0) {
			if (allow_ext) { out.want_ext = true; goto leave_fallback; }
			if (!need_sa) { rerun_with_sa = true; goto leave_fallback; }
			const uint64_t K = (uint64_t)key_symbols(T.type);
			uint64_t h0 = 1; while (2 * h0 <= R0 * K) h0 *= 2;
			uint64_t hb = 0;
			isa_from_sa_rel(sa.lo, len, start, ws.t[5], s);
			out.tie_rounds += tie_rounds_domain(sa.lo, left, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, start, len, own ? own : len, ws.t[5], h0, s, &left, &hb);
			if (left) { out.slow_ties = left; out.tie_rounds += tie_rounds(T, sa.lo, left, hb / K, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, len, s, &left, 0); }
		}
		if (!big && left && defer && out.tie_rounds >= R0) {
			if (!need_sa) { rerun_with_sa = true; goto leave_fallback; }
			out.tie_left = left; out.tie_U = Bf[0]; out.tie_gid = Bf[1];
		}
		if (!big && left && whole && out.tie_rounds >= R0) {
			if (!need_sa) { rerun_with_sa = true; goto leave_fallback; }
			uint64_t h0 = 1; while (2 * h0 <= R0 * (uint64_t)key_symbols(T.type)) h0 *= 2;
			// ranks for the suffixes that were ever listed; the few others doubling asks for are searched (B3M_ISA_FULL=1: all n entries, as before)
			const bool sparse = !sw().ISA_FULL.on();
			if (sparse && u1) isa_sparse_from_slots(sa.lo, len, U1_keep, u1, ws.t[5], s);
			else if (sparse) {
				slots_again(Bf[4]);
				isa_sparse_from_slots(sa.lo, len, Bf[4], total, ws.t[5], s);
			} else isa_from_sa(sa.lo, len, ws.t[5], s);
			out.tie_rounds += tie_rounds(T, sa.lo, left, 0, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, len, s, &left, 0, ws.t[5], h0, nullptr, nullptr, 0, sparse, true);
		}
		// BWT symbols (and p0) of the slots finished here
		// (into a buffer the rounds only use as scratch: a shard's deferred list lives on in two of the others)
		slots_again(Bf[5]);
		emit_bwt_list(T, Bf[5], total, sa.lo, start, d_bwt, d_p0, s, &O.smp, sa.hi);
		stage("fallback rounds + emit");
		leave_fallback:;
	}
	if (rerun_with_sa) {
		ws.d_symhist_out = nullptr;       // the histogram was added by this pass already
		block_sort_msd(T, start, len_in, ws, s, tl, out, d_bwt, d_p0, shard, nshards, true, own, allow_ext);
		out.reran_with_sa = true;         // (the plan remembers: its next runs keep the suffix array from the start)
		return;
	}
	B3M_HIP(hipGetLastError());
	out.d_sa = need_sa ? sa.lo : nullptr;      // without need_sa only the entries of tied / overflow records were written
	out.d_sa_hi = need_sa ? sa.hi : nullptr;
}


// ================================================================================================
// Sort of 32-bit keys (the rank lists of the merges: SURVEY.md 8a row a9) by the same means as the
// block sort: two unstable 8-bit partitions through LDS (64-key runs), then every final bucket's
// remaining <= 15 bits are counted in an LDS histogram and the keys are written back in order (keys
// only, so the sorted bucket IS its histogram).  32 B of traffic per key instead of the 48 B of a
// four-pass LSD radix sort, on scatter passes that are not bound by stable ranking.
// ================================================================================================
constexpr int U32_LOWBITS = 15;
constexpr int U32_BINS = 1 << U32_LOWBITS;
constexpr int U32_THREADS = 1024;
constexpr uint64_t U32_MIN_N = 1ull << 23;             // below this the LSD passes are as fast
constexpr uint32_t U32_MAX_BUCKET = 65535;             // 16-bit counters: a larger final bucket (skewed ranks) -> LSD fallback

struct U32Args {
	const uint32_t *in; uint32_t *out;
	const MsdTile *tiles;
	uint32_t *hist;
	int D, shift;
};

template <bool COUNT_ONLY, int THREADS>
__global__ __launch_bounds__(THREADS) void k_u32_pass(U32Args A)
{
	constexpr int CHUNK = THREADS * MSD_ITEMS;
	__shared__ uint32_t cnt[MSD_MAXBINS];
	__shared__ uint32_t binstart[COUNT_ONLY ? 1 : MSD_MAXBINS];
	__shared__ uint32_t gbase[COUNT_ONLY ? 1 : MSD_MAXBINS];
	__shared__ uint32_t skey[COUNT_ONLY ? 1 : CHUNK];
	__shared__ uint32_t wsum[THREADS / 64];
	const uint32_t tid = threadIdx.x;
	const int bins = 1 << A.D;
	const MsdTile tile = A.tiles[blockIdx.x];
	if (COUNT_ONLY) { for (int b = tid; b < bins; b += THREADS) cnt[b] = 0; }
	else { for (int b = tid; b < bins; b += THREADS) gbase[b] = A.hist[tile.hist_base + (uint64_t)b * tile.stride]; }
	__syncthreads();
	const uint32_t nchunks = (tile.count + CHUNK - 1) / CHUNK;
	for (uint32_t c = 0; c < nchunks; ++c) {
		const uint32_t cbase = c * CHUNK;
		if (!COUNT_ONLY) { for (int b = tid; b < bins; b += THREADS) cnt[b] = 0; __syncthreads(); }
		uint32_t key[MSD_ITEMS], dg[MSD_ITEMS], slot[MSD_ITEMS]; uint32_t vmask = 0;
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) {                  // loads first, uses later
			uint32_t e = cbase + s * THREADS + tid;
			key[s] = A.in[(uint64_t)tile.in_off + (e < tile.count ? e : tile.count - 1)];
			if (e < tile.count) vmask |= 1u << s;
		}
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) dg[s] = (key[s] >> A.shift) & (uint32_t)(bins - 1);
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) if (vmask & (1u << s)) slot[s] = atomicAdd(&cnt[dg[s]], 1u);
		if (COUNT_ONLY) continue;
		__syncthreads();
		{
			static_assert(COUNT_ONLY || THREADS >= MSD_MAXBINS, "one bin per thread");
			const uint32_t v = (int)tid < bins ? cnt[tid] : 0u;
			uint32_t inc = wave_incl_sum(v);
			uint32_t w = tid >> 6, l = tid & 63;
			if (l == 63) wsum[w] = inc;
			__syncthreads();
			uint32_t pre = 0;
			for (uint32_t i = 0; i < w; ++i) pre += wsum[i];
			if ((int)tid < bins) binstart[tid] = pre + inc - v;
		}
		__syncthreads();
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) if (vmask & (1u << s)) skey[binstart[dg[s]] + slot[s]] = key[s];
		__syncthreads();
		const uint32_t nvalid = binstart[bins - 1] + cnt[bins - 1];
		for (uint32_t p = tid; p < nvalid; p += THREADS) {
			uint32_t q = skey[p];
			const uint32_t dd = (q >> A.shift) & (uint32_t)(bins - 1);
			A.out[(uint64_t)gbase[dd] + (p - binstart[dd])] = q;
		}
		__syncthreads();
		for (int b = tid; b < bins; b += THREADS) gbase[b] += cnt[b];
		__syncthreads();
	}
	if (COUNT_ONLY) {
		__syncthreads();
		for (int b = tid; b < bins; b += THREADS) A.hist[tile.hist_base + (uint64_t)b * tile.stride] = cnt[b];
	}
}

// one workgroup per final bucket (< 2^16 keys): histogram of the low bits in 16-bit LDS counters (64 KB: two
// workgroups per CU), exclusive scan in place, and every output slot finds its bin by binary search in
// the scanned histogram (15 LDS reads): coalesced loads and stores
__global__ __launch_bounds__(U32_THREADS) void k_u32_finish(const uint32_t *in, uint32_t *out, const uint32_t *bstart, int lowbits)
{
	__shared__ __attribute__((aligned(16))) uint32_t hw[U32_BINS / 2 + 4];     // two 16-bit counters per word, then exclusive starts
	__shared__ uint32_t wsum[U32_THREADS / 64];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t b = blockIdx.x;
	const uint32_t s0 = bstart[b], m = bstart[b + 1] - s0;
	if (m == 0) return;
	const uint32_t bins = 1u << lowbits, mask = bins - 1;
	const uint32_t prefix = b << lowbits;
	for (uint32_t i = tid; i < U32_BINS / 2 + 4; i += U32_THREADS) hw[i] = 0;
	__syncthreads();
	for (uint32_t e0 = 0; e0 < m; e0 += 8 * U32_THREADS) {       // eight loads in flight per thread, then their atomics
		uint32_t v[8];
		#pragma unroll
		for (int k = 0; k < 8; ++k) { uint32_t e = e0 + k * U32_THREADS + tid; v[k] = in[(uint64_t)s0 + (e < m ? e : m - 1)]; }
		#pragma unroll
		for (int k = 0; k < 8; ++k) {
			uint32_t e = e0 + k * U32_THREADS + tid, bin = v[k] & mask;
			if (e < m) atomicAdd(&hw[bin >> 1], (bin & 1u) ? 0x10000u : 1u);
		}
	}
	__syncthreads();
	{
		constexpr int WPT = (U32_BINS / 2) / U32_THREADS;       // 16 words = 32 bins per thread
		uint32_t w[WPT]; uint32_t sum = 0;
		#pragma unroll
		for (int k = 0; k < WPT; ++k) { w[k] = hw[tid * WPT + k]; sum += (w[k] & 0xFFFFu) + (w[k] >> 16); }
		uint32_t inc = wave_incl_sum(sum);
		if (lane == 63) wsum[wave] = inc;
		__syncthreads();
		uint32_t run = inc - sum;
		for (uint32_t i = 0; i < wave; ++i) run += wsum[i];
		#pragma unroll
		for (int k = 0; k < WPT; ++k) {
			uint32_t lo = w[k] & 0xFFFFu, hi = w[k] >> 16;
			hw[tid * WPT + k] = run | ((run + lo) << 16);       // starts < 2^16 because m < 2^16
			run += lo + hi;
		}
	}
	__syncthreads();
	const uint16_t *hs = reinterpret_cast<const uint16_t *>(hw);  // hs[bin] = first slot of the bin
	for (uint32_t s = tid; s < m; s += U32_THREADS) {
		uint32_t lo = 0, hi = bins;                          // largest bin with hs[bin] <= s
		while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (hs[mid] <= s) lo = mid; else hi = mid; }
		out[(uint64_t)s0 + s] = prefix | lo;
	}
}

// workspace: `scratch` (u32) of at least sort_u32_msd_scratch(n) entries, `scan_ws` for exclusive_sum_u32
uint64_t sort_u32_msd_scratch(uint64_t n)
{
	uint64_t tiles = n / MSD_SUPER + 260;
	return 2 * 256 * tiles + 4 * tiles + 65540 + 3 * 260 + 64;
}
// returns false (nothing sorted) when the keys are too few or too skewed for this path
bool sort_u32_msd(uint32_t *kA, uint32_t *kB, uint64_t n, int nbits, uint32_t *scratch, uint64_t scratch_elems, uint32_t *scan_ws,
                  hipStream_t s, Timeline *tl, uint32_t **k_res)
{
	if (n < U32_MIN_N || n >= (1ull << 32) || nbits <= U32_LOWBITS + 8 || nbits > 32 || scratch_elems < sort_u32_msd_scratch(n)) return false;
	const int U = nbits - U32_LOWBITS;                       // upper bits: two levels of <= 8
	const int D1 = std::min(8, U), D2 = U - D1;
	if (D2 > 8 || D2 < 1) return false;
	const uint64_t tiles_max = n / MSD_SUPER + 260;
	uint32_t *histA = scratch, *histB = histA + 256 * tiles_max;
	MsdTile *d_tiles = reinterpret_cast<MsdTile *>(histB + 256 * tiles_max);
	uint32_t *starts = reinterpret_cast<uint32_t *>(d_tiles + tiles_max), *tab = starts + 65540;
	// ---- level A: regular tiles ----
	const uint32_t ntA = (uint32_t)((n + MSD_SUPER - 1) / MSD_SUPER);
	std::vector<MsdTile> tiles(ntA);
	for (uint32_t i = 0; i < ntA; ++i) { tiles[i].in_off = i * (uint32_t)MSD_SUPER; tiles[i].count = (uint32_t)std::min<uint64_t>(MSD_SUPER, n - (uint64_t)i * MSD_SUPER); tiles[i].hist_base = i; tiles[i].stride = ntA; }
	B3M_HIP(hipMemcpyAsync(d_tiles, tiles.data(), sizeof(MsdTile) * ntA, hipMemcpyHostToDevice, s));
	uint32_t tab0[3] = {0, 0, ntA};
	B3M_HIP(hipMemcpyAsync(tab, tab0, 12, hipMemcpyHostToDevice, s));
	B3M_HIP(hipStreamSynchronize(s));
	U32Args A{kA, kB, d_tiles, histA, D1, nbits - D1};
	{
		Scope sc(tl, KF_RADIX_HIST, 4 * n);
		hipLaunchKernelGGL((k_u32_pass<true, 256>), dim3(ntA), dim3(256), 0, s, A);
	}
	{
		Scope sc(tl, KF_SCAN, 8ull * 256 * ntA);
		exclusive_sum_u32(histA, (uint64_t)(1u << D1) * ntA, scan_ws, s);
	}
	{
		Scope sc(tl, KF_RADIX_SCATTER, 8 * n);
		hipLaunchKernelGGL((k_u32_pass<false, U32_THREADS>), dim3(ntA), dim3(U32_THREADS), 0, s, A);
	}
	const uint32_t nbA = 1u << D1;
	hipLaunchKernelGGL(k_next_starts, dim3(nblk((uint64_t)nbA + 1)), dim3(256), 0, s, histA, tab, 1u, D1, (uint32_t)n, starts);
	std::vector<uint32_t> st(nbA + 1);
	B3M_HIP(hipMemcpyAsync(st.data(), starts, 4ull * (nbA + 1), hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	// ---- level B: tiles inside every level-A bucket ----
	tiles.clear();
	std::vector<uint32_t> tb(3ull * nbA);
	uint64_t hoff = 0;
	for (uint32_t b = 0; b < nbA; ++b) {
		uint32_t sz = st[b + 1] - st[b];
		uint32_t nt = (sz + MSD_SUPER - 1) / MSD_SUPER;
		tb[b] = st[b]; tb[nbA + b] = (uint32_t)hoff; tb[2ull * nbA + b] = nt;
		for (uint32_t j = 0; j < nt; ++j) {
			MsdTile t; t.in_off = st[b] + j * (uint32_t)MSD_SUPER; t.count = std::min<uint32_t>((uint32_t)MSD_SUPER, sz - j * (uint32_t)MSD_SUPER);
			t.hist_base = (uint32_t)(hoff + j); t.stride = nt;
			tiles.push_back(t);
		}
		hoff += ((uint64_t)1 << D2) * nt;
	}
	B3M_REQUIRE(tiles.size() <= tiles_max && hoff <= 256 * tiles_max, "rank sort workspace too small");
	B3M_HIP(hipMemcpyAsync(d_tiles, tiles.data(), sizeof(MsdTile) * tiles.size(), hipMemcpyHostToDevice, s));
	B3M_HIP(hipMemcpyAsync(tab, tb.data(), 4 * tb.size(), hipMemcpyHostToDevice, s));
	B3M_HIP(hipStreamSynchronize(s));
	const uint32_t ntB = (uint32_t)tiles.size();
	U32Args Bq{kB, kA, d_tiles, histB, D2, nbits - D1 - D2};
	{
		Scope sc(tl, KF_RADIX_HIST, 4 * n);
		hipLaunchKernelGGL((k_u32_pass<true, 256>), dim3(ntB), dim3(256), 0, s, Bq);
	}
	{
		Scope sc(tl, KF_SCAN, 8ull * hoff);
		exclusive_sum_u32(histB, hoff, scan_ws, s);
	}
	{
		Scope sc(tl, KF_RADIX_SCATTER, 8 * n);
		hipLaunchKernelGGL((k_u32_pass<false, U32_THREADS>), dim3(ntB), dim3(U32_THREADS), 0, s, Bq);
	}
	const uint32_t nbF = nbA << D2;
	hipLaunchKernelGGL(k_next_starts, dim3(nblk((uint64_t)nbF + 1)), dim3(256), 0, s, histB, tab, nbA, D2, (uint32_t)n, starts);
	// skew check: a final bucket handled by one workgroup must stay small
	std::vector<uint32_t> fs(nbF + 1);
	B3M_HIP(hipMemcpyAsync(fs.data(), starts, 4ull * (nbF + 1), hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	uint32_t mx = 0;
	for (uint32_t b = 0; b < nbF; ++b) mx = std::max(mx, fs[b + 1] - fs[b]);
	if (mx > U32_MAX_BUCKET) return false;                   // kA holds a permutation of the input: the caller sorts it the slow way
	{
		Scope sc(tl, KF_RADIX_SCATTER, 8 * n);
		hipLaunchKernelGGL(k_u32_finish, dim3(nbF), dim3(U32_THREADS), 0, s, kA, kB, starts, U32_LOWBITS);
	}
	B3M_HIP(hipGetLastError());
	*k_res = kB;
	return true;
}

}  // namespace b3m
