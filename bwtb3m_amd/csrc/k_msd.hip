// k_msd.hip -- the fast block sort for 2-bit inputs (pac/pacterm): two MSD partition passes of
// 8-byte records followed by an in-LDS sort of every final bucket.
//
// Why: an LSD radix sort of (u64 key, u32 position) pairs moves ~24 B per suffix per pass plus a
// gather for the BWT symbol (~240 B/suffix in all).  Here a suffix is ONE 8-byte record
//        [ 29 key bits | 3 bits predecessor symbol | 32 bits position ]
// built straight from the packed text, partitioned twice by the leading bits of its rotation
// (D1 then D2 bits, ~2^21 final buckets of 1-2 K suffixes), and each final bucket is sorted inside
// the LDS of one workgroup, ties resolved exactly against the text, and written as BWT symbol +
// suffix-array entry.  HBM traffic: 0.5 (text, twice) + 8 + 8 + 16 + 8 + 5 = ~46 B per suffix.
//
//   pass A (k_msd_pass<true,*>)  : text -> records, partitioned by the first D1 bits
//   pass B (k_msd_pass<false,*>) : records of every A bucket partitioned by the next D2 bits
//   pass C (k_lds_sort)          : one workgroup per final bucket: stable LDS radix sort on the
//                                  remaining 29-D2 key bits (64-lane ballot match-any ranking),
//                                  exact tie resolution, emit
// Partition order inside a bucket is irrelevant (the next level re-sorts), so slots come from LDS
// atomics; each workgroup owns a "super tile" of 64 Ki records processed in 4 Ki chunks through
// LDS so that every digit's run leaves the CU as one contiguous segment, and the histogram matrix
// has only len*bins/65536 entries.
// Buckets larger than the LDS capacity, or tie groups deeper than the in-kernel limit (skewed or
// repetitive input), are finished by the list-based key-extension rounds of k_blocksort.hip.
// pacterm: the terminator and everything behind it read as base 0 for bucketing; the exact tie
// resolution (key_at with terminator tags) restores "terminator < A" where it matters.
#include "b3m_internal.h"
#include "b3m_text.h"
#include <algorithm>
#include <vector>

namespace b3m {

constexpr int MSD_THREADS = 256;
constexpr int MSD_ITEMS = 16;
constexpr int MSD_CHUNK = MSD_THREADS * MSD_ITEMS;     // 4096 records staged through LDS at a time
constexpr int MSD_SUPER = 16 * MSD_CHUNK;              // 65536 records per workgroup
constexpr int MSD_MAXBITS = 8;                         // fan-out per pass: full 128-byte runs per chunk
constexpr int MSD_MAXBINS = 1 << MSD_MAXBITS;
constexpr int MSD_MAXLEVELS = 3;
constexpr uint64_t MSD_MIN_LEN = 1ull << 18;
constexpr int LS_CAP = 512;                            // final bucket capacity: one WAVE sorts one bucket
constexpr int LS_BINBITS = 9;
constexpr int LS_BINS = 1 << LS_BINBITS;               // bin-scatter table of the wave sort
constexpr int LS_TIE_CAP = 32;                         // bins / tie groups up to this size are finished in-kernel
constexpr int LS_WAVES = MSD_THREADS / 64;             // independent waves (buckets) per workgroup
constexpr int REC_KEY_SHIFT = 35;
constexpr int REC_KEY_BITS = 29;

static inline uint32_t nblk(uint64_t n, uint32_t t = 256) { return (uint32_t)((n + t - 1) / t); }
static inline uint64_t al(uint64_t x) { return (x + 255) & ~255ull; }

// total partition bits t: final buckets of <= 256 suffixes on average (capacity 512, one wave
// each); split over ceil(t/8) levels as evenly as possible
struct MsdDigits { int L; int bits[MSD_MAXLEVELS]; int total; };
static MsdDigits msd_digits(uint64_t len)
{
	int t = 0; while (((uint64_t)256 << t) < len) ++t;
	t = std::max(4, std::min(MSD_MAXLEVELS * MSD_MAXBITS, t));
	MsdDigits d; d.total = t; d.L = std::max(2, (t + MSD_MAXBITS - 1) / MSD_MAXBITS);
	for (int l = 0; l < d.L; ++l) d.bits[l] = t / d.L + (l < t % d.L ? 1 : 0);
	return d;
}

bool msd_applicable(const TextView &T, uint64_t len)
{
	return (T.type == B3M_INPUT_PAC || T.type == B3M_INPUT_PACTERM) && len >= MSD_MIN_LEN;
}

static const uint64_t MSD_NB_MAX = (1ull << (MSD_MAXLEVELS * MSD_MAXBITS)) + 2;       // final buckets
static const uint64_t MSD_TAB_MAX = (1ull << ((MSD_MAXLEVELS - 1) * MSD_MAXBITS)) + 2; // buckets entering a level

uint64_t msd_ws_bytes(uint64_t cap)
{
	if (cap < MSD_MIN_LEN) return 4096;
	uint64_t tiles = cap / MSD_SUPER + MSD_TAB_MAX + 2;
	uint64_t hist = (uint64_t)MSD_MAXBINS * tiles;
	return al(4 * hist) + al(sizeof(MsdTile) * tiles) + al(4 * MSD_NB_MAX) + al(4 * 4 * MSD_TAB_MAX) + al(4 * (MSD_NB_MAX + 1)) + al(4 * scan_ws_elems(hist)) + 4096;
}
void msd_ws_carve(MsdWs &ws, void *base, uint64_t cap)
{
	if (cap < MSD_MIN_LEN) { ws = MsdWs(); return; }
	uint8_t *p = (uint8_t *)base;
	uint64_t tiles = cap / MSD_SUPER + MSD_TAB_MAX + 2;
	uint64_t hist = (uint64_t)MSD_MAXBINS * tiles;
	ws.hist = (uint32_t *)p; ws.hist_cap = hist; p += al(4 * hist);
	ws.tiles = (MsdTile *)p; ws.tiles_cap = tiles; p += al(sizeof(MsdTile) * tiles);
	ws.sub_start = (uint32_t *)p; p += al(4 * MSD_NB_MAX);
	ws.bucket_tab = (uint32_t *)p; ws.lvl_start = ws.bucket_tab + 3 * MSD_TAB_MAX; p += al(4 * 4 * MSD_TAB_MAX);
	ws.overflow = (uint32_t *)p; ws.overflow_cap = MSD_NB_MAX; p += al(4 * (MSD_NB_MAX + 1));
	ws.scan_ws = (uint32_t *)p;
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
__device__ __forceinline__ uint64_t make_rec(uint64_t x, int D1, uint32_t pred, uint32_t idx, uint32_t *dig)
{
	*dig = (uint32_t)(x >> (64 - D1));
	uint64_t kf = (x >> (64 - D1 - REC_KEY_BITS)) & ((1ull << REC_KEY_BITS) - 1);
	return (kf << REC_KEY_SHIFT) | ((uint64_t)pred << 32) | idx;
}

struct MsdArgs {
	TextView T; uint64_t start, len;
	uint64_t P0;                 // FROM_TEXT: absolute position of item 0 of tile 0 (multiple of 16)
	const uint64_t *rec_in;      // !FROM_TEXT
	uint64_t *rec_out;
	const MsdTile *tiles;        // null: regular tiling (pass A)
	uint32_t *hist;              // COUNT_ONLY: counts out ; else scanned offsets in
	int D;                       // digit bits of this pass
	int shift;                   // !FROM_TEXT: digit = (rec >> shift) & (2^D - 1)
	uint32_t keep_lo, keep_hi;   // FROM_TEXT scatter: only records with keep_lo <= digit < keep_hi are kept (suffix-range shard)
};

template <bool FROM_TEXT, bool COUNT_ONLY>
__global__ __launch_bounds__(MSD_THREADS) void k_msd_pass(MsdArgs A)
{
	__shared__ uint32_t cnt[MSD_MAXBINS];
	__shared__ uint32_t binstart[COUNT_ONLY ? 1 : MSD_MAXBINS];
	__shared__ uint32_t gbase[COUNT_ONLY ? 1 : MSD_MAXBINS];
	__shared__ uint64_t srec[COUNT_ONLY ? 1 : MSD_CHUNK];
	__shared__ uint8_t sdig[(COUNT_ONLY || !FROM_TEXT) ? 1 : MSD_CHUNK];   // level 0 keeps its digit out of the record
	__shared__ uint32_t wsum[8];
	const uint32_t tid = threadIdx.x;
	const int bins = 1 << A.D;
	MsdTile tile;
	if (A.tiles) tile = A.tiles[blockIdx.x];
	else { tile.in_off = blockIdx.x * (uint32_t)MSD_SUPER; tile.count = MSD_SUPER; tile.hist_base = blockIdx.x; tile.stride = gridDim.x; }
	if (COUNT_ONLY) { for (int b = tid; b < bins; b += MSD_THREADS) cnt[b] = 0; }
	else { for (int b = tid; b < bins; b += MSD_THREADS) gbase[b] = A.hist[tile.hist_base + (uint64_t)b * tile.stride]; }
	__syncthreads();
	const uint32_t nchunks = (tile.count + MSD_CHUNK - 1) / MSD_CHUNK;
	for (uint32_t c = 0; c < nchunks; ++c) {
		const uint32_t cbase = c * MSD_CHUNK;
		const uint32_t cn = tile.count - cbase < (uint32_t)MSD_CHUNK ? tile.count - cbase : (uint32_t)MSD_CHUNK;
		if (!COUNT_ONLY) { for (int b = tid; b < bins; b += MSD_THREADS) cnt[b] = 0; __syncthreads(); }
		uint64_t rec[MSD_ITEMS]; uint32_t dg[MSD_ITEMS]; uint32_t slot[MSD_ITEMS]; uint32_t vmask = 0;
		if (FROM_TEXT) {
			// thread = one 32-bit word of packed text = 16 consecutive positions
			uint64_t pos0 = A.P0 + tile.in_off + cbase + (uint64_t)tid * 16;
			const uint64_t end = A.start + A.len;
			if (pos0 < end && pos0 + 16 > A.start) {
				const bool fast = pos0 + 16 + 32 <= A.T.nbases && pos0 >= A.start && pos0 + 16 <= end && pos0 > 0;
				if (fast) {
					// the whole word lies inside the block and far from the text end: 3 words -> 16 records
					const uint8_t *p = A.T.b + ((pos0 >> 4) << 2);
					uint64_t wp = be32(p - 4), w0 = be32(p), w1 = be32(p + 4), w2 = be32(p + 8);
					uint64_t hi = (w0 << 32) | w1;
					const uint32_t padd = A.T.type == B3M_INPUT_PACTERM ? 1u : 0u;
					#pragma unroll
					for (int s = 0; s < MSD_ITEMS; ++s) {
						uint64_t x = s ? ((hi << (2 * s)) | (w2 >> (32 - 2 * s))) : hi;
						uint32_t pr = (s ? (uint32_t)(w0 >> (32 - 2 * s)) & 3u : (uint32_t)wp & 3u) + padd;
						rec[s] = make_rec(x, A.D, pr, (uint32_t)(pos0 + s), &dg[s]);
						if (COUNT_ONLY || (dg[s] >= A.keep_lo && dg[s] < A.keep_hi)) vmask |= 1u << s;
					}
				} else {
					// block edges / end of the text: one position at a time, rolled (rare: keep the code small)
					#pragma nounroll
					for (int s = 0; s < MSD_ITEMS; ++s) {
						uint64_t i = pos0 + s;
						uint64_t rr = 0; uint32_t dd = 0; bool ok = false;
						if (i >= A.start && i < end) {
							rr = make_rec(window_safe(A.T, i), A.D, pred_sym(A.T, i), (uint32_t)i, &dd);
							ok = COUNT_ONLY || (dd >= A.keep_lo && dd < A.keep_hi);
						}
						#pragma unroll
						for (int q = 0; q < MSD_ITEMS; ++q) if (q == s) { rec[q] = rr; dg[q] = dd; if (ok) vmask |= 1u << q; }
					}
				}
			}
		} else {
			// all 16 loads go out back to back (clamped index, no branch around a load: a load
			// inside its own if-block gets its own s_waitcnt and the chunk takes 16 round trips)
			#pragma unroll
			for (int s = 0; s < MSD_ITEMS; ++s) {
				uint32_t e = cbase + s * MSD_THREADS + tid;
				uint32_t ec = e < tile.count ? e : tile.count - 1;
				rec[s] = A.rec_in[(uint64_t)tile.in_off + ec];
				if (e < tile.count) vmask |= 1u << s;
			}
			#pragma unroll
			for (int s = 0; s < MSD_ITEMS; ++s) dg[s] = (uint32_t)(rec[s] >> A.shift) & (uint32_t)(bins - 1);
		}
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) if (vmask & (1u << s)) slot[s] = atomicAdd(&cnt[dg[s]], 1u);
		if (COUNT_ONLY) continue;
		__syncthreads();
		// exclusive scan of cnt over the bins -> start of every digit run inside the chunk
		{
			const int per = MSD_MAXBINS / MSD_THREADS;        // consecutive bins per thread
			uint32_t loc[per]; uint32_t sum = 0;
			#pragma unroll
			for (int k = 0; k < per; ++k) { int b = tid * per + k; loc[k] = b < bins ? cnt[b] : 0; sum += loc[k]; }
			uint32_t inc = wave_incl_sum(sum);
			uint32_t w = tid >> 6, l = tid & 63;
			if (l == 63) wsum[w] = inc;
			__syncthreads();
			uint32_t pre = 0;
			for (uint32_t i = 0; i < w; ++i) pre += wsum[i];
			uint32_t run = pre + inc - sum;
			#pragma unroll
			for (int k = 0; k < per; ++k) { int b = tid * per + k; if (b < bins) binstart[b] = run; run += loc[k]; }
		}
		__syncthreads();
		#pragma unroll
		for (int s = 0; s < MSD_ITEMS; ++s) if (vmask & (1u << s)) { uint32_t p = binstart[dg[s]] + slot[s]; srec[p] = rec[s]; if (FROM_TEXT) sdig[p] = (uint8_t)dg[s]; }
		__syncthreads();
		// number of valid records of this chunk = binstart[last] + cnt[last]; copy out run by run
		const uint32_t nvalid = binstart[bins - 1] + cnt[bins - 1];
		(void)cn;
		for (uint32_t p = tid; p < nvalid; p += MSD_THREADS) {
			uint64_t q = srec[p];
			const uint32_t d = FROM_TEXT ? (uint32_t)sdig[p] : ((uint32_t)(q >> A.shift) & (uint32_t)(bins - 1));
			A.rec_out[(uint64_t)gbase[d] + (p - binstart[d])] = q;
		}
		__syncthreads();
		for (int b = tid; b < bins; b += MSD_THREADS) gbase[b] += cnt[b];
		__syncthreads();
	}
	if (COUNT_ONLY) {
		__syncthreads();
		for (int b = tid; b < bins; b += MSD_THREADS) A.hist[tile.hist_base + (uint64_t)b * tile.stride] = cnt[b];
	}
}

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

// ------------------------------------------------------------------------------------------------
// final level: one WAVE sorts one bucket (<= 512 records, ~180 on uniform input) in its own 8 KB
// of LDS; the four waves of a workgroup are independent, so there is no workgroup barrier and a
// CU keeps 20 buckets in flight.  The remaining key bits of a bucket are close to uniform, so one
// bin-scatter (LDS atomics into 512 bins) leaves almost every record in its final bin.  Then:
//   rank inside the bin by (key field, arrival slot), bins permuted in place
//   records whose key field equals a neighbour's fetch the next 31 bases of their rotation (all
//   gathers of the wave in flight together)                        -> deep key in LDS
//   the first record of every tie group insertion-sorts its group on the deep keys (LDS only;
//   the text is consulted again only if those still tie or the text end is near).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct SortOut { uint32_t *sa; uint8_t *bwt; uint64_t start; uint32_t *p0; uint32_t *overflow; uint32_t overflow_cap; uint32_t *tiebits; };

// one wave sorts the final bucket g = records [b0, b0+m) of `rec`; outb (LS_CAP u64) and cb
// (2*LS_BINS u32) are this wave's private LDS
// the records of a bucket, 8 per lane (lane l holds records l, l+64, ...); issued early by the caller
// so that they are in flight while the previous bucket is being sorted
__device__ __forceinline__ void wave_load_bucket(const uint64_t *rec, uint32_t b0, uint32_t m, uint64_t (&r)[LS_CAP / 64])
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t mm = m < (uint32_t)LS_CAP ? m : (uint32_t)LS_CAP;
	#pragma unroll
	for (int t = 0; t < LS_CAP / 64; ++t) { uint32_t e = t * 64 + lane; r[t] = mm ? rec[(uint64_t)b0 + (e < mm ? e : mm - 1)] : 0; }
}

template <bool WRITE_SA>
__device__ __forceinline__ void wave_sort_bucket(const TextView &T, const uint64_t *rec, uint32_t g, uint32_t b0, uint32_t m, int sortbits,
                                                  uint64_t *outb, uint32_t *cb, const SortOut &O, const uint64_t (&r)[LS_CAP / 64])
{
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t *cnt = cb, *bst = cb + LS_BINS;
	uint32_t *sa = O.sa; uint8_t *bwt = O.bwt; const uint64_t start = O.start; uint32_t *p0 = O.p0;
	uint32_t *overflow = O.overflow; const uint32_t overflow_cap = O.overflow_cap; uint32_t *tiebits = O.tiebits;
	if (m == 0) return;
	if (m > (uint32_t)LS_CAP) {
		if (lane == 0) { uint32_t k = atomicAdd(&overflow[0], 1u); if (k < overflow_cap) overflow[1 + k] = g; }
		for (uint32_t e = lane; e < m; e += 64) sa[b0 + e] = (uint32_t)rec[b0 + e];
		return;
	}
	const int binbits = sortbits < LS_BINBITS ? (sortbits > 0 ? sortbits : 0) : LS_BINBITS;
	const int bshift = REC_KEY_SHIFT + sortbits - binbits;
	const uint32_t bmask = (1u << binbits) - 1;
	constexpr int PER = LS_CAP / 64;               // 8 records per lane
	uint32_t sl[PER];
	#pragma unroll
	for (int k = 0; k < LS_BINS / 64; ++k) cnt[k * 64 + lane] = 0;
	wave_sync();
	#pragma unroll
	for (int t = 0; t < PER; ++t) {
		uint32_t e = t * 64 + lane;
		sl[t] = 0;
		if (e < m) sl[t] = atomicAdd(&cnt[(uint32_t)(r[t] >> bshift) & bmask], 1u);
	}
	wave_sync();
	{
		constexpr int per = LS_BINS / 64;           // 8 consecutive bins per lane
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
		if (e < m) outb[bst[(uint32_t)(r[t] >> bshift) & bmask] + sl[t]] = r[t];
	}
	wave_sync();
	// rank inside the bin: position by (key field, slot); bins of <= 4 records without a loop
	bool ovf = false;
	uint32_t dest[PER];
	#pragma unroll
	for (int t = 0; t < PER; ++t) {
		uint32_t e = t * 64 + lane;
		dest[t] = 0;
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
	wave_sync();
	#pragma unroll
	for (int t = 0; t < PER; ++t) if (t * 64 + lane < m) outb[dest[t]] = r[t];
	wave_sync();
	// tie groups (equal key field): bit p of the tie bitmap = "record p has the key field of
	// record p-1"; k_tie_fix resolves the runs against the text.  Scattered single-bit atomics:
	// no hot address.  tied: records that k_tie_fix will look at (they need their SA entry).
	uint32_t tiedm = 0;
	#pragma unroll
	for (int t = 0; t < PER; ++t) {
		uint32_t e = t * 64 + lane;
		if (e < m) {
			uint64_t kf = outb[e] >> REC_KEY_SHIFT;
			bool sp = e > 0 && (outb[e - 1] >> REC_KEY_SHIFT) == kf;
			bool sn = e + 1 < m && (outb[e + 1] >> REC_KEY_SHIFT) == kf;
			if (sp) { uint32_t q = b0 + e; atomicOr(&tiebits[q >> 5], 1u << (q & 31)); }
			if (sp || sn) tiedm |= 1u << t;
		}
	}
	if (__ballot(ovf)) {
		if (lane == 0) { uint32_t k = atomicAdd(&overflow[0], 1u); if (k < overflow_cap) overflow[1 + k] = g; }
		for (uint32_t e = lane; e < m; e += 64) sa[b0 + e] = (uint32_t)rec[b0 + e];
		return;
	}
	#pragma unroll
	for (int t = 0; t < PER; ++t) {
		uint32_t e = t * 64 + lane;
		if (e < m) {
			uint64_t q = outb[e];
			uint32_t i = (uint32_t)q;
			if (WRITE_SA || (tiedm & (1u << t))) sa[b0 + e] = i;       // without consumers only k_tie_fix reads positions
			bwt[b0 + e] = (uint8_t)((q >> 32) & 7u);
			if (i == (uint32_t)start) *p0 = b0 + e;
		}
	}
}


// final level as its own launch (kept for reference / experiments): one wave per final bucket
template <bool WRITE_SA>
__global__ __launch_bounds__(MSD_THREADS) void k_lds_sort(TextView T, const uint64_t *rec, const uint32_t *sub_start, uint32_t nbuckets, int sortbits, SortOut O)
{
	__shared__ uint64_t s_rec[LS_WAVES][LS_CAP];
	__shared__ __attribute__((aligned(16))) uint32_t s_cb[LS_WAVES][2 * LS_BINS];   // counts | starts
	const uint32_t wave = threadIdx.x >> 6;
	const uint32_t g = blockIdx.x * LS_WAVES + wave;
	if (g >= nbuckets) return;
	const uint32_t b0 = sub_start[g], b1 = sub_start[g + 1];
	uint64_t r[LS_CAP / 64];
	wave_load_bucket(rec, b0, b1 - b0, r);
	wave_sort_bucket<WRITE_SA>(T, rec, g, b0, b1 - b0, sortbits, s_rec[wave], s_cb[wave], O, r);
}

// LAST PARTITION LEVEL + FINAL SORT FUSED: one workgroup owns one bucket entering the last level
// (~46 K records at 3 Gbase): it counts the bucket's next digit, scatters the bucket into its
// own range of the other record buffer (chunks of 4096 through LDS, as k_msd_pass), and its four
// waves then sort the 256 sub-buckets one after the other.  What the scatter writes is read back
// by the same CU microseconds later: that 24 GB write + 24 GB read stay in L2 / Infinity Cache
// instead of HBM, and the separate counting pass, scan and host round trip of the level are gone.
template <bool WRITE_SA>
__global__ __launch_bounds__(MSD_THREADS) void k_bucket_finish(TextView T, const uint64_t *rec_in, uint64_t *rec_tmp, const uint32_t *bstart, uint32_t nb,
                                                                 int D, int shift, int sortbits, uint32_t *sub_start, SortOut O)
{
	constexpr int CH = MSD_THREADS * MSD_ITEMS;                 // 4096
	__shared__ uint32_t cnt[MSD_MAXBINS];
	__shared__ uint32_t binstart[MSD_MAXBINS];
	__shared__ uint32_t gbase[MSD_MAXBINS];
	__shared__ uint32_t sub[MSD_MAXBINS + 4];
	__shared__ uint32_t wsum[8];
	__shared__ __attribute__((aligned(16))) uint64_t big[CH];     // scatter staging, later the waves' sort buffers (32 KB)
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t b = blockIdx.x;
	const uint32_t s0 = bstart[b], s1 = bstart[b + 1], m = s1 - s0;
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
		if (tid < bins) sub_start[((uint64_t)b << D) + tid] = s0 + ex;
		if (b + 1 == nb && tid == 0) sub_start[(uint64_t)nb << D] = s1;
	}
	__syncthreads();
	if (m == 0) return;
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
	uint32_t *cb = reinterpret_cast<uint32_t *>(big + LS_WAVES * LS_CAP) + wave * 2 * LS_BINS;   // 4 x 4 KB
	uint64_t rn[LS_CAP / 64];
	if (wave < bins) wave_load_bucket(rec_tmp, s0 + sub[wave], sub[wave + 1] - sub[wave], rn);
	for (uint32_t d = wave; d < bins; d += LS_WAVES) {
		uint32_t lo = sub[d], hi = sub[d + 1];
		uint64_t r[LS_CAP / 64];
		#pragma unroll
		for (int t = 0; t < LS_CAP / 64; ++t) r[t] = rn[t];
		if (d + LS_WAVES < bins) wave_load_bucket(rec_tmp, s0 + sub[d + LS_WAVES], sub[d + LS_WAVES + 1] - sub[d + LS_WAVES], rn);   // next one flies meanwhile
		wave_sort_bucket<WRITE_SA>(T, rec_tmp, (b << D) | d, s0 + lo, hi - lo, sortbits, outb, cb, O, r);
		wave_sync();
	}
}

// one thread per word of the tie bitmap: every run of set bits starting in the word is a tie group
// (head = the position before the run).  The group's rotations agree on their first eq_syms
// symbols; they are ordered by the next 31 bases (one gather each), by full comparison if that
// does not decide.  Groups longer than LS_TIE_CAP are skipped: such a group fills one bin of
// k_lds_sort beyond LS_TIE_CAP, so its whole bucket was flagged for the overflow path.
__global__ void k_tie_fix(TextView T, const uint32_t *tiebits, uint64_t len, uint32_t eq_syms, uint32_t *sa, uint8_t *bwt, uint64_t start, uint32_t *p0)
{
	uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t nwords = (len + 31) >> 5;
	if (w >= nwords) return;
	uint32_t bits = tiebits[w];
	if (!bits) return;
	uint32_t prevbit = w ? (tiebits[w - 1] >> 31) : 0u;
	uint32_t starts = bits & ~((bits << 1) | prevbit);          // set bits whose predecessor bit is clear
	while (starts) {
		uint32_t k = __ffs(starts) - 1; starts &= starts - 1;
		uint64_t q = (w << 5) + k;                               // first tied record; head = q-1
		uint64_t pos = q - 1;
		uint32_t g = 2;
		for (uint64_t z = q + 1; z < len && g <= (uint32_t)LS_TIE_CAP && ((tiebits[z >> 5] >> (z & 31)) & 1u); ++z) ++g;
		if (g > (uint32_t)LS_TIE_CAP) continue;   // lies in one bin of > LS_TIE_CAP records: that bucket went to the overflow path
		for (uint32_t a = 1; a < g; ++a) {
			uint32_t x = sa[pos + a]; uint8_t bx = bwt[pos + a];
			bool ex = (uint64_t)x + eq_syms + 32 <= T.nbases;
			uint64_t dx = ex ? window32(T, (uint64_t)x + eq_syms) >> 2 : 0;
			uint32_t j = a;
			while (j > 0) {
				uint32_t y = sa[pos + j - 1];
				bool ey = (uint64_t)y + eq_syms + 32 <= T.nbases;
				uint64_t dy = ey ? window32(T, (uint64_t)y + eq_syms) >> 2 : 0;
				bool y_le_x;
				if (ex && ey && dx != dy) y_le_x = dy < dx;
				else y_le_x = compare_rotations(T, y, x) <= 0;
				if (y_le_x) break;
				sa[pos + j] = y; bwt[pos + j] = bwt[pos + j - 1]; --j;
			}
			sa[pos + j] = x; bwt[pos + j] = bx;
		}
		for (uint32_t a = 0; a < g; ++a) if (sa[pos + a] == (uint32_t)start) *p0 = (uint32_t)(pos + a);
	}
}

// list of the suffix-array slots of the overflow buckets: U[s] = slot, gid[s] = bucket start
__global__ void k_fill_overflow_list(const uint32_t *off, const uint32_t *b0, uint32_t nov, uint64_t total, uint32_t *U, uint32_t *gid)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= total) return;
	uint32_t lo = 0, hi = nov;          // last j with off[j] <= s
	while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (off[mid] <= s) lo = mid; else hi = mid; }
	U[s] = b0[lo] + (uint32_t)(s - off[lo]);
	if (gid) gid[s] = b0[lo];
}

void block_sort_msd(const TextView &T, uint64_t start, uint64_t len_in, BlockSortWs &ws, hipStream_t s, Timeline *tl, BlockSortOut &out,
                    uint8_t *d_bwt, uint32_t *d_p0, int shard, int nshards, bool need_sa)
{
	uint64_t len = len_in;           // becomes the number of KEPT suffixes after level 0 (suffix-range shard)
	MsdWs &M = ws.msd;
	B3M_REQUIRE(M.hist != nullptr, "MSD workspace missing");
	const MsdDigits dg = msd_digits(len_in);
	uint64_t *rin = ws.keyB, *rout = ws.keyA;      // level 0 writes keyA
	uint32_t *sa = ws.idxA;
	out.used_msd = true;
	// ---- level 0: text -> records, partitioned by the first bits[0] bits ----
	const uint64_t P0 = start & ~15ull;
	const uint64_t span = start + len_in - P0;
	const uint32_t ntA = (uint32_t)((span + MSD_SUPER - 1) / MSD_SUPER);
	const uint32_t bins0 = 1u << dg.bits[0];
	B3M_REQUIRE((uint64_t)bins0 * ntA <= M.hist_cap, "MSD histogram workspace too small");
	MsdArgs A{};
	A.T = T; A.start = start; A.len = len; A.P0 = P0; A.rec_in = nullptr; A.rec_out = rout; A.tiles = nullptr; A.hist = M.hist; A.D = dg.bits[0]; A.shift = 0;
	A.keep_lo = 0; A.keep_hi = bins0;
	{
		Scope sc(tl, KF_MSD_HIST, span / 4);
		hipLaunchKernelGGL((k_msd_pass<true, true>), dim3(ntA), dim3(MSD_THREADS), 0, s, A);
	}
	out.slice_offset = 0;
	if (nshards > 1) {
		// suffix-range shard: this rank keeps the level-0 digits [lo, hi) holding the ranks
		// [shard*n/N, (shard+1)*n/N) as evenly as whole digits allow.  The totals come from the
		// text alone, so every rank derives the same boundaries without talking to anyone.
		unsigned long long *d_tot = reinterpret_cast<unsigned long long *>(M.sub_start);   // free until the levels are done
		hipLaunchKernelGGL(k_row_totals, dim3(bins0), dim3(256), 0, s, M.hist, ntA, d_tot);
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
		A.keep_lo = lo; A.keep_hi = hi;
		out.slice_offset = cum[lo];
		len = cum[hi] - cum[lo];
		B3M_REQUIRE(len <= ws.cap, "suffix-range shard larger than the workspace (skewed input): use block mode or fewer shards");
		if (lo) B3M_HIP(hipMemsetAsync(M.hist, 0, 4ull * lo * ntA, s));
		if (hi < bins0) B3M_HIP(hipMemsetAsync(M.hist + (uint64_t)hi * ntA, 0, 4ull * (bins0 - hi) * ntA, s));
	}
	out.kept_len = len;
	if (len == 0) { out.d_sa = sa; return; }          // empty shard (one huge digit took the neighbours' share)
	{
		Scope sc(tl, KF_SCAN, 8ull * bins0 * ntA);
		exclusive_sum_u32(M.hist, (uint64_t)bins0 * ntA, M.scan_ws, s);
	}
	{
		Scope sc(tl, KF_MSD_SCATTER_TEXT, span / 4 + 8 * len);
		hipLaunchKernelGGL((k_msd_pass<true, false>), dim3(ntA), dim3(MSD_THREADS), 0, s, A);
	}
	// bucket starts leaving level 0: a one-bucket table {start 0, hist_off 0, ntiles ntA}
	uint32_t nb = 1;                    // buckets entering the level just done
	{
		uint32_t tab0[3] = {0, 0, ntA};
		B3M_HIP(hipMemcpyAsync(M.bucket_tab, tab0, 12, hipMemcpyHostToDevice, s));
		B3M_HIP(hipStreamSynchronize(s));
	}
	int consumed = 0;                   // key-field bits used by levels >= 1
	uint32_t nfinal = 0;
	const uint32_t eq_syms = (uint32_t)((dg.bits[0] + REC_KEY_BITS) / 2);    // symbols known equal when key fields tie
	uint32_t *d_tiebits = ws.t[0];      // tie bitmap: len bits
	for (int lv = 0; lv + 1 < dg.L; ++lv) {
		const int D = dg.bits[lv];
		const uint32_t nnext = nb << D;
		const bool last = lv + 2 == dg.L;         // the next level is the last one: fused with the final sort
		uint32_t *d_starts = last ? M.lvl_start : M.sub_start;
		B3M_REQUIRE(!last || nnext + 1 <= MSD_TAB_MAX, "MSD level table too small");
		hipLaunchKernelGGL(k_next_starts, dim3(nblk((uint64_t)nnext + 1)), dim3(256), 0, s, M.hist, M.bucket_tab, nb, D, (uint32_t)len, d_starts);
		const int Dn = dg.bits[lv + 1];
		consumed += Dn;
		std::swap(rin, rout);
		if (last) {
			// ---- last partition level + final sort, one workgroup per bucket ----
			// compulsory HBM traffic: the records once in, the BWT byte (and the SA entry) out; the
			// scatter inside the bucket and its re-read are meant to stay in L2 / Infinity Cache
			Scope sc(tl, KF_LDS_SORT, (need_sa ? 13 : 9) * len);
			B3M_HIP(hipMemsetAsync(M.overflow, 0, 4, s));
			B3M_HIP(hipMemsetAsync(d_tiebits, 0, 4 * ((len + 31) / 32 + 1), s));
			SortOut O{sa, d_bwt, start, d_p0, M.overflow, (uint32_t)std::min<uint64_t>(M.overflow_cap - 2, (uint64_t)nnext << Dn), d_tiebits};
			if (need_sa)
				hipLaunchKernelGGL(k_bucket_finish<true>, dim3(nnext), dim3(MSD_THREADS), 0, s, T, rin, rout, d_starts, nnext, Dn, 64 - consumed, REC_KEY_BITS - consumed, M.sub_start, O);
			else
				hipLaunchKernelGGL(k_bucket_finish<false>, dim3(nnext), dim3(MSD_THREADS), 0, s, T, rin, rout, d_starts, nnext, Dn, 64 - consumed, REC_KEY_BITS - consumed, M.sub_start, O);
			nfinal = nnext << Dn;
			break;
		}
		// ---- build the tiles of the next level on the host ----
		std::vector<uint32_t> st(nnext + 1);
		B3M_HIP(hipMemcpyAsync(st.data(), M.sub_start, 4ull * (nnext + 1), hipMemcpyDeviceToHost, s));
		B3M_HIP(hipStreamSynchronize(s));
		std::vector<MsdTile> tiles;
		std::vector<uint32_t> tab(3ull * nnext);
		uint64_t hoff = 0;
		for (uint32_t b = 0; b < nnext; ++b) {
			uint32_t sz = st[b + 1] - st[b];
			uint32_t nt = (sz + MSD_SUPER - 1) / MSD_SUPER;
			tab[b] = st[b]; tab[nnext + b] = (uint32_t)hoff; tab[2ull * nnext + b] = nt;
			for (uint32_t j = 0; j < nt; ++j) {
				MsdTile t; t.in_off = st[b] + j * (uint32_t)MSD_SUPER; t.count = std::min<uint32_t>((uint32_t)MSD_SUPER, sz - j * (uint32_t)MSD_SUPER);
				t.hist_base = (uint32_t)(hoff + j); t.stride = nt;
				tiles.push_back(t);
			}
			hoff += ((uint64_t)1 << Dn) * nt;
		}
		B3M_REQUIRE(hoff <= M.hist_cap && tiles.size() <= M.tiles_cap && hoff < (1ull << 32) && 3ull * nnext <= 3 * MSD_TAB_MAX, "MSD workspace too small");
		B3M_HIP(hipMemcpyAsync(M.tiles, tiles.data(), sizeof(MsdTile) * tiles.size(), hipMemcpyHostToDevice, s));
		B3M_HIP(hipMemcpyAsync(M.bucket_tab, tab.data(), 4 * tab.size(), hipMemcpyHostToDevice, s));
		B3M_HIP(hipStreamSynchronize(s));      // host vectors must outlive the copies
		// ---- partition level lv+1 ----
		MsdArgs Bq{};
		Bq.T = T; Bq.start = start; Bq.len = len; Bq.P0 = 0; Bq.rec_in = rin; Bq.rec_out = rout; Bq.tiles = M.tiles; Bq.hist = M.hist; Bq.D = Dn;
		Bq.shift = 64 - consumed; Bq.keep_lo = 0; Bq.keep_hi = 0xFFFFFFFFu;
		const uint32_t ntB = (uint32_t)tiles.size();
		{
			Scope sc(tl, KF_MSD_HIST, 8 * len);
			hipLaunchKernelGGL((k_msd_pass<false, true>), dim3(ntB), dim3(MSD_THREADS), 0, s, Bq);
		}
		{
			Scope sc(tl, KF_SCAN, 8ull * hoff);
			exclusive_sum_u32(M.hist, hoff, M.scan_ws, s);
		}
		{
			Scope sc(tl, KF_MSD_SCATTER, 16 * len);
			hipLaunchKernelGGL((k_msd_pass<false, false>), dim3(ntB), dim3(MSD_THREADS), 0, s, Bq);
		}
		nb = nnext;
	}
	B3M_REQUIRE(nfinal != 0, "internal: MSD needs at least two levels");
	{
		Scope sc(tl, KF_TIES, len / 8);
		hipLaunchKernelGGL(k_tie_fix, dim3(nblk((len + 31) / 32)), dim3(256), 0, s, T, ws.t[0], len, eq_syms, sa, d_bwt, start, d_p0);
	}
	uint32_t nov = 0;
	B3M_HIP(hipMemcpyAsync(&nov, M.overflow, 4, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	out.msd_overflow_buckets = nov;
	if (nov) {
		// ---- fallback: oversized / deeply tied buckets go through the key-extension rounds ----
		Scope sc(tl, KF_TIES, 0);
		std::vector<uint32_t> ids(nov), sub(nfinal + 1);
		B3M_HIP(hipMemcpyAsync(ids.data(), M.overflow + 1, 4ull * nov, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipMemcpyAsync(sub.data(), M.sub_start, 4ull * (nfinal + 1), hipMemcpyDeviceToHost, s));
		B3M_HIP(hipStreamSynchronize(s));
		std::sort(ids.begin(), ids.end());
		std::vector<uint32_t> off(nov + 1), b0(nov);
		uint64_t total = 0;
		for (uint32_t j = 0; j < nov; ++j) { off[j] = (uint32_t)total; b0[j] = sub[ids[j]]; total += sub[ids[j] + 1] - sub[ids[j]]; }
		off[nov] = (uint32_t)total;
		// the overflow table is dead now: reuse it for (off, b0)
		B3M_REQUIRE(2ull * nov + 2 <= M.overflow_cap, "overflow table too small");
		uint32_t *d_off = M.overflow, *d_b0 = M.overflow + nov + 1;
		B3M_HIP(hipMemcpyAsync(d_off, off.data(), 4ull * (nov + 1), hipMemcpyHostToDevice, s));
		B3M_HIP(hipMemcpyAsync(d_b0, b0.data(), 4ull * nov, hipMemcpyHostToDevice, s));
		uint32_t *Bf[6] = {ws.t[0], ws.t[1], ws.idxB, ws.t[2], ws.t[3], ws.t[4]};
		hipLaunchKernelGGL(k_fill_overflow_list, dim3(nblk(total)), dim3(256), 0, s, d_off, d_b0, nov, total, Bf[0], Bf[1]);
		B3M_HIP(hipStreamSynchronize(s));
		uint64_t left = 0;
		out.tie_elements = total;
		out.tie_rounds = tie_rounds(T, sa, total, 0, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, len, s, &left);
		// BWT symbols (and p0) of the slots finished here
		hipLaunchKernelGGL(k_fill_overflow_list, dim3(nblk(total)), dim3(256), 0, s, d_off, d_b0, nov, total, ws.t[0], (uint32_t *)nullptr);
		emit_bwt_list(T, ws.t[0], total, sa, start, d_bwt, d_p0, s);
	}
	B3M_HIP(hipGetLastError());
	out.d_sa = need_sa ? sa : nullptr;      // without need_sa only the entries of tied / overflow records were written
}

}  // namespace b3m
