// k_blocksort.hip -- suffix-block sort (the role of
// libmaus2::suffixsort::BwtMergeBlockSortRequest::dispatch) and the products of a sorted block.
//
// The packed text of the WHOLE input is HBM-resident, so block suffixes are compared as
// rotations of the whole circular text directly: no lcpnext look-ahead window (row a6) is
// needed.  Sort = 64-bit key of the first K symbols (radix sort) + key-extension rounds for the
// suffixes still tied (next K symbols, segmented by tie group) until all groups are singletons
// or n symbols have been compared (equal rotations of a periodic text).
#include "b3m_internal.h"
#include <chrono>
#include "b3m_text.h"

namespace b3m {

// ------------------------------------------------------------------------------------------------
// K1: keys of the first K symbols, one thread per 16 bases (one 32-bit word of packed text)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_keys_packed(TextView T, uint64_t start, uint64_t len, uint64_t *keys, uint32_t *idx)
{
	// thread handles block-relative symbols [16*g, 16*g+16) where g counted from the word holding `start`
	uint64_t first_word = start >> 4;
	uint64_t w = first_word + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t pos0 = w << 4;
	if (pos0 >= start + len) return;
	const int K = (T.type == B3M_INPUT_PAC) ? KSYM_PAC : KSYM_PACTERM;
	bool fast = pos0 + 15 + 32 <= T.nbases && pos0 >= start && pos0 + 16 <= start + len;
	if (fast) {
		const uint8_t *p = T.b + (w << 2);
		uint64_t w0 = be32(p), w1 = be32(p + 4), w2 = be32(p + 8);
		uint64_t hi = (w0 << 32) | w1;
		#pragma unroll
		for (int s = 0; s < 16; ++s) {
			uint64_t x = s ? ((hi << (2 * s)) | (w2 >> (32 - 2 * s))) : hi;
			uint64_t k = (T.type == B3M_INPUT_PAC) ? x : (((x >> (64 - 2 * KSYM_PACTERM)) << TAGBITS_PACTERM) | TAG_NORMAL);
			uint64_t r = pos0 + s - start;
			keys[r] = k;
			idx[r] = (uint32_t)(pos0 + s);
		}
	} else {
		(void)K;
		for (int s = 0; s < 16; ++s) {
			uint64_t i = pos0 + s;
			if (i < start || i >= start + len) continue;
			keys[i - start] = key_at(T, i, 0);
			idx[i - start] = (uint32_t)i;
		}
	}
}

__global__ __launch_bounds__(256) void k_keys_generic(TextView T, uint64_t start, uint64_t len, uint64_t *keys, uint32_t *idx)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= len) return;
	keys[r] = key_at(T, start + r, 0);
	idx[r] = (uint32_t)(start + r);
}

// ------------------------------------------------------------------------------------------------
// K3/K4: tie groups and key-extension rounds, on a shrinking list
//
// State of a round: U[s] = positions (ascending) of the suffixes still tied, gid[s] = label of the
// tie group of slot s (equal label <=> same group; labels ascend with s).  A round gives every
// listed suffix the key of its next K symbols, sorts the list by (group, key), writes the suffixes
// back into their group's slots and splits the groups where keys differ; slots whose group became
// a singleton leave the list.  Only the list is touched, never the whole block.
// ------------------------------------------------------------------------------------------------
// head labels from the sorted first keys: hid[r] = r if keys[r] starts a new group else 0
__global__ void k_heads_from_keys(const uint64_t *keys, uint64_t len, uint32_t *hid)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= len) return;
	hid[r] = (r == 0 || keys[r] != keys[r - 1]) ? (uint32_t)r : 0u;
}
// after the max-scan hid[s] = slot of the head of s's group.  flag[s] = 1 if the group has >= 2 members
__global__ void k_tie_flags(const uint32_t *hid, uint64_t u, uint32_t *flag)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s > u) return;
	if (s == u) { flag[s] = 0; return; }
	bool h0 = hid[s] == (uint32_t)s, h1 = (s + 1 == u) ? true : (hid[s + 1] == (uint32_t)(s + 1));
	flag[s] = (h0 && h1) ? 0u : 1u;
}
// compaction: U == nullptr means the identity list (first round after the radix sort).  hid[s] = list index of the head of s's group (max-scanned)
__global__ void k_tie_compact(const uint32_t *flag_scan, const uint32_t *hid, const uint32_t *U, uint64_t u, uint32_t *U2, uint32_t *gid2)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	bool h0 = hid[s] == (uint32_t)s, h1 = (s + 1 == u) ? true : (hid[s + 1] == (uint32_t)(s + 1));
	if (!(h0 && h1)) { uint32_t p = flag_scan[s]; U2[p] = U ? U[s] : (uint32_t)s; gid2[p] = flag_scan[hid[s]]; }      // label = index of the group's head in the NEW list (the head is kept with its group)
}
// keys of the next K symbols for the list; value = slot
// (idx_hi: bits 32.. of the positions, texts of 2^32 symbols and more; null otherwise)
// (off: the symbols in front of the key, see key_at_off)
__global__ void k_tie_keys(TextView T, const uint32_t *U, const uint32_t *idx, const uint8_t *idx_hi, uint64_t u, uint64_t off, uint64_t *nk, uint32_t *slot)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	const uint32_t r = U[s];
	nk[s] = key_at_off(T, (uint64_t)idx[r] | (idx_hi ? (uint64_t)idx_hi[r] << 32 : 0ull), off);
	slot[s] = (uint32_t)s;
}
// prefix doubling: the key of a listed suffix is the current rank (ISA: exact for resolved suffixes,
// first slot of the tie group for listed ones) of the suffix h symbols further on
__global__ void k_tie_keys_isa(const uint32_t *isa, uint64_t n, uint64_t h, const uint32_t *U, const uint32_t *idx, uint64_t u, uint64_t *nk, uint32_t *slot)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	uint64_t q = (uint64_t)idx[U[s]] + h; if (q >= n) q -= n;
	nk[s] = isa[q];
	slot[s] = (uint32_t)s;
}
__global__ void k_tie_gather_isa(const uint32_t *isa, uint64_t n, uint64_t h, const uint32_t *U, const uint32_t *idx, const uint32_t *slot_final,
                                 uint64_t u, uint32_t *tmp_idx, uint64_t *tmp_key)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	uint32_t i = idx[U[slot_final[s]]];
	uint64_t q = (uint64_t)i + h; if (q >= n) q -= n;
	tmp_idx[s] = i;
	tmp_key[s] = isa[q];
}
__global__ void k_isa_init(const uint32_t *sa, uint64_t len, uint32_t *isa)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r < len) isa[sa[r]] = (uint32_t)r;
}
// ---- the inverse suffix array, written where it is read ----
// Doubling reads the rank of suffix i+h for the LISTED suffixes i only; filling all n entries is n random 4-byte writes (137 ms at 3 Gbase, a
// third of a repeat-rich genome's whole sort).  Sparse form: every entry starts UNSET, the suffixes that were ever listed get theirs (exact for
// those the key rounds finished, the group's first slot for the rest: k_isa_slots, k_isa_list), and a round first collects the targets that are
// still UNSET -- suffixes the record sort placed for good, near the end of a repeat mostly -- and finds their ranks by binary search in the
// suffix array (k_isa_need, k_isa_search).  Slots of groups still tied hold their members in any order: a suffix that is not in the group compares
// the same way with all of them, so the search steps over them correctly.  Too many such targets (n/64): the whole array is filled after all.
constexpr uint32_t ISA_UNSET = 0xFFFFFFFFu;
__global__ void k_isa_slots(const uint32_t *U0, uint64_t total, const uint32_t *sa, uint32_t *isa)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s < total) { const uint32_t r = U0[s]; isa[sa[r]] = r; }
}
__global__ void k_isa_need(const uint32_t *isa, uint64_t n, uint64_t h, const uint32_t *U, const uint32_t *idx, uint64_t u, uint32_t *Q, uint32_t *cnt)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t q = 0; bool need = false;
	if (s < u) { q = (uint64_t)idx[U[s]] + h; if (q >= n) q -= n; need = isa[q] == ISA_UNSET; }
	const uint64_t mask = __ballot(need);
	if (!mask) return;
	const uint32_t lane = lane_id(), first = (uint32_t)__ffsll((long long)mask) - 1u;
	uint32_t base = 0;
	if (lane == first) base = atomicAdd(cnt, (uint32_t)__popcll(mask));
	base = __shfl(base, (int)first, 64);
	if (need) Q[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)q;
}
__global__ void k_isa_search(TextView T, const uint32_t *sa, uint64_t n, const uint32_t *Q, uint32_t nq, uint32_t *isa)
{
	uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nq) return;
	const uint32_t q = Q[t];
	uint64_t lo = 0, hi = n;
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		const uint32_t p = sa[mid];
		if (p == q) { lo = mid; break; }
		if (compare_rotations(T, p, q) < 0) lo = mid + 1; else hi = mid;
	}
	isa[q] = (uint32_t)lo;
}
// hid[s] = s where the label changes (heads of the listed groups), else 0
__global__ void k_heads_from_gid(const uint32_t *gid, uint64_t u, uint32_t *hid)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s < u) hid[s] = (s == 0 || gid[s] != gid[s - 1]) ? (uint32_t)s : 0u;
}
// listed suffixes: rank = first slot of their group (hid = list index of the group's head, after the max-scan)
__global__ void k_isa_list(const uint32_t *U, const uint32_t *hid, const uint32_t *idx, uint64_t u, uint32_t *isa)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s < u) isa[idx[U[s]]] = U[hid[s]];
}
// second sort key: group label of the slot (as u64 so that the pair sorter can be reused)
__global__ void k_tie_gidkeys(const uint32_t *gid, const uint32_t *slot_sorted, uint64_t u, uint64_t *gk)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	gk[s] = gid[slot_sorted[s]];
}
// gather (idx, key) of the final order into temporaries (the key is recomputed from the text)
__global__ void k_tie_gather(TextView T, const uint32_t *U, const uint32_t *idx, const uint8_t *idx_hi, const uint32_t *slot_final,
                             uint64_t u, uint64_t off, uint32_t *tmp_idx, uint8_t *tmp_hi, uint64_t *tmp_key)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	const uint32_t r = U[slot_final[s]];
	const uint32_t i = idx[r];
	const uint32_t h = idx_hi ? idx_hi[r] : 0u;
	tmp_idx[s] = i;
	if (idx_hi) tmp_hi[s] = (uint8_t)h;
	tmp_key[s] = key_at_off(T, (uint64_t)i | ((uint64_t)h << 32), off);
}
// write back into the tie slots (the sorted group labels equal the old ones slot by slot) and
// mark the new group heads.  (lpos: the list handed in is a sub-list -- whole groups, in order -- of the list hid belongs to: lpos[s] = index there)
__global__ void k_tie_apply(const uint32_t *U, const uint32_t *gid, const uint32_t *tmp_idx, const uint8_t *tmp_hi, const uint64_t *tmp_key,
                            uint64_t u, uint32_t *idx, uint8_t *idx_hi, uint32_t *hid, const uint32_t *lpos)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	idx[U[s]] = tmp_idx[s];
	if (idx_hi) idx_hi[U[s]] = tmp_hi[s];
	bool head = (s == 0) || gid[s] != gid[s - 1] || tmp_key[s] != tmp_key[s - 1];
	const uint32_t at = lpos ? lpos[s] : (uint32_t)s;
	hid[at] = head ? at : 0u;
}
__global__ void k_emit_bwt_list(TextView T, const uint32_t *U, uint64_t u, const uint32_t *sa, const uint8_t *sa_hi, uint64_t start, uint8_t *bwt, uint32_t *p0, SampleOut smp)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	uint32_t r = U[s];
	uint64_t i = (uint64_t)sa[r] | (sa_hi ? (uint64_t)sa_hi[r] << 32 : 0ull);
	uint64_t j = i ? i - 1 : T.n - 1;
	if (T.type == B3M_INPUT_SYMBOLS32) reinterpret_cast<uint32_t *>(bwt)[r] = sym_at(T, j);      // (wide symbols: the BWT array holds uint32)
	else bwt[r] = (uint8_t)sym_at(T, j);
	if (i == start) *p0 = r;
	if (sample_hit(smp, (uint32_t)i)) sample_put(smp, r, i);      // the listed slots are final here: their samples could not be written earlier
}

// ------------------------------------------------------------------------------------------------
// A round inside LDS.  Two radix sorts of the whole list per round (by key, then stably by group) cost ten to twelve passes
// over it, whatever the groups look like; once the groups are small -- after the first round or two of a repeat-rich genome they
// are -- a workgroup can take a TILE of the list (whole groups, at most TT_CAP slots), fetch the keys, order every group in LDS
// and write the suffixes back: one pass.  Tile c = the groups whose head lies in the list slots [c * TT_STRIDE, (c+1) * TT_STRIDE);
// a group above TT_GROUP members belongs to no tile and goes through the radix sorts as a sub-list of its own
// (k_tie_large_*).  Groups of up to TT_COUNT members are ordered by counting (each member counts the smaller ones), larger
// ones by a bitonic network over the tile, keyed (group, key).
// ------------------------------------------------------------------------------------------------
constexpr int MP_GROUP = 512, MP_ROW = 9;               // pieces above MP_GROUP slots are split further; table row = piece start >> MP_ROW  (256 / 8 measured: 208 -> 216 ms on the genome-like text)
// (tiles of 2048 slots / 256 threads, 32 KB of LDS: four workgroups per CU; 4096 / 512 until round 4: 227 -> 222 ms on the genome-like text.  -DB3M_TT_LARGE: the old form)
#if defined(B3M_TT_TINY)
constexpr int TT_STRIDE = 512, TT_GROUP = 512, TT_CAP = 1024, TT_THREADS = 128, TT_PER = TT_CAP / TT_THREADS, TT_COUNT = 512;
#elif defined(B3M_TT_LARGE)
constexpr int TT_STRIDE = 2048, TT_GROUP = 2048, TT_CAP = 4096, TT_THREADS = 512, TT_PER = TT_CAP / TT_THREADS, TT_COUNT = 512;
#else
constexpr int TT_STRIDE = 1024, TT_GROUP = 1024, TT_CAP = 2048, TT_THREADS = 256, TT_PER = TT_CAP / TT_THREADS, TT_COUNT = 512;
#endif
static int tt_count_max()      // measurement switch (+ the experiment build's timing-only flags in the high half: never set in the product library)
{
	const int v = (int)sw().TT_COUNT.get(TT_COUNT);
	return (v < 8 ? 8 : (v > TT_COUNT ? TT_COUNT : v)) | ((int)sw().TT_EXPERIMENT.get(0) << 16);
}
static_assert(TT_STRIDE - 1 + TT_GROUP <= TT_CAP, "a tile holds the groups of its stride");

// flag[s] = 1 if the group of slot s has more than TT_GROUP members (hid[s] = list index of the group's head); flag[u] = 0
__device__ __forceinline__ bool tie_large(const uint32_t *hid, const uint32_t *gid, uint64_t u, uint64_t s, uint32_t group = TT_GROUP)
{
	const uint64_t far = (uint64_t)hid[s] + group;
	return far < u && gid[far] == gid[s];
}
__global__ void k_tie_large_flags(const uint32_t *hid, const uint32_t *gid, uint64_t u, uint32_t *flag)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s > u) return;
	flag[s] = (s < u && tie_large(hid, gid, u, s)) ? 1u : 0u;
}
__global__ void k_tie_large_compact(const uint32_t *flag_scan, const uint32_t *hid, const uint32_t *gid, const uint32_t *U, uint64_t u,
                                    uint32_t *U2, uint32_t *gid2, uint32_t *lpos)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	if (tie_large(hid, gid, u, s)) { const uint32_t p = flag_scan[s]; U2[p] = U[s]; gid2[p] = gid[s]; lpos[p] = (uint32_t)s; }
}

// MODE 0: keys from the text; 1: ranks (isa); 2: keys and suffixes are given per slot (kin, vin: what the splits of the large groups left, k_msd_*)
// and a group also starts where the splits marked a boundary (pb[s] & MSD_BND)
constexpr uint32_t MSD_BND = 0x80000000u, MSD_LOC = 0x40000000u, MSD_PID = 0x3FFFFFFFu;      // (MSD_LOC: the slot's pair lives in the second of the two buffers)
template <int MODE>
__device__ __forceinline__ bool tile_head(const uint32_t *gid, const uint32_t *pb, uint64_t s)
{
	return s == 0 || gid[s] != gid[s - 1] || (MODE == 2 && (pb[s] & MSD_BND));
}
template <int MODE>
__global__ __launch_bounds__(TT_THREADS) void k_tie_tile(TextView T, const uint32_t *U, const uint32_t *gid, uint64_t u, uint32_t *idx,
                                                         const uint32_t *isa, uint64_t off, uint32_t *hid,
                                                         const uint64_t *kin, const uint32_t *vin, const uint32_t *pb,
                                                         const uint64_t *kin1, const uint32_t *vin1, int kshift, int count_max)
{
	// kshift = 12 where the key leaves room (pacterm: 48 bits, ranks: 32): the slot inside the tile rides in the low bits, every key of the tile is
	// distinct and one 64-bit compare per pair orders them (0: 64-bit keys, ties are broken by the slot in a second compare)
#ifdef B3M_PHASE_EXPERIMENT
	const int xflags = count_max >> 16; count_max &= 0xFFFF;      // (`make experiment`, timing only, WRONG results: 1 = no ordering, 2 = no write-back, 4 = leave after the fetch)
#else
	constexpr int xflags = 0;
#endif
	constexpr bool ISA = MODE == 1;
	constexpr uint32_t GROUP = MODE == 2 ? MP_GROUP : TT_GROUP;       // (MODE 2: a group above MP_GROUP is a piece of one key, k_msd_leftover's)
	__shared__ uint64_t sk[TT_CAP];                  // keys
	__shared__ uint32_t sv[TT_CAP];                  // suffixes
	__shared__ uint16_t sg[TT_CAP];                  // group of the slot, counted inside the tile from 1 (0xFFFF: padding)
	__shared__ __attribute__((aligned(4))) uint16_t gs[TT_CAP + 2];              // first slot of group g; gs[groups + 1] = m
	__shared__ uint32_t wtot[TT_THREADS / 64];
	__shared__ uint32_t red[4];
	const uint32_t tid = threadIdx.x;
	const uint64_t c0 = (uint64_t)blockIdx.x * TT_STRIDE, c1 = c0 + TT_STRIDE < u ? c0 + TT_STRIDE : u;
	if (tid == 0) { red[0] = 0xFFFFFFFFu; red[1] = 0u; red[2] = 0xFFFFFFFFu; red[3] = 0u; }
	__syncthreads();
	for (uint64_t s = c0 + tid; s < c1; s += TT_THREADS)
		if (tile_head<MODE>(gid, pb, s)) { atomicMin(&red[0], (uint32_t)s); atomicMax(&red[1], (uint32_t)s); }
	__syncthreads();
	const uint32_t a = red[0], hl = red[1];
	if (a == 0xFFFFFFFFu) return;                    // no group starts here (the inside of a large one)
	// the end of the last group: the first head behind the stride, looked for as far as a group of TT_GROUP reaches
	const uint64_t lim = (uint64_t)hl + GROUP + 1 < u ? (uint64_t)hl + GROUP + 1 : u;
	for (uint64_t s = c1 + tid; s < lim; s += TT_THREADS) if (tile_head<MODE>(gid, pb, s)) atomicMin(&red[2], (uint32_t)s);
	__syncthreads();
	uint64_t e;
	if (red[2] != 0xFFFFFFFFu) e = red[2];
	else e = (u - hl <= (uint64_t)GROUP) ? u : hl;          // runs to the end of the list, or is a large group: not this tile's
	const uint32_t m = (uint32_t)(e - a);
	if (m == 0) return;
	uint32_t P = 64; while (P < m) P <<= 1;
	// ---- fetch: suffix, key, head mark ----
	uint32_t hsum = 0;
	#pragma unroll
	for (int k = 0; k < TT_PER; ++k) {
		const uint32_t p = tid * TT_PER + k;          // consecutive slots per thread: the head marks are summed in place
		uint32_t head = 0;
		if (p < m) {
			const uint64_t s = (uint64_t)a + p;
			const bool loc1 = MODE == 2 && (pb[s] & MSD_LOC);
			const uint32_t i = MODE == 2 ? (loc1 ? vin1[s] : vin[s]) : idx[U[s]];
			uint64_t key;
			if (MODE == 2) key = loc1 ? kin1[s] : kin[s];
			else if (ISA) { uint64_t q = (uint64_t)i + off; if (q >= T.n) q -= T.n; key = isa[q]; }
			else key = key_at_off(T, i, off);
			head = (p == 0 || tile_head<MODE>(gid, pb, s)) ? 1u : 0u;
			sk[p] = (key << kshift) | (kshift ? (uint64_t)p : 0ull); sv[p] = i;
			if (head) hsum += 1;
			sg[p] = (uint16_t)(hsum | (head << 15));   // partial count inside the thread (< 16), head mark in bit 15
		} else if (p < P) { sk[p] = ~0ull; sv[p] = 0u; sg[p] = 0xFFFFu; }
	}
	// heads in front of this thread's slots: scan of the 512 counts (inside the waves, then the eight wave totals)
	uint32_t before;
	{
		const uint32_t inc = wave_incl_sum(hsum);
		if ((tid & 63u) == 63u) wtot[tid >> 6] = inc;
		__syncthreads();
		before = inc - hsum;
		for (uint32_t w = 0; w < (tid >> 6); ++w) before += wtot[w];
	}
	#pragma unroll
	for (int k = 0; k < TT_PER; ++k) {
		const uint32_t p = tid * TT_PER + k;
		if (p < m) {
			const uint32_t w = sg[p], g = before + (w & 0x7FFFu);
			sg[p] = (uint16_t)g;
			if (w & 0x8000u) gs[g] = (uint16_t)p;
			if (p == m - 1) gs[g + 1] = (uint16_t)m;
		}
	}
	__syncthreads();
	// largest group of the tile
	#pragma unroll
	for (int k = 0; k < TT_PER; ++k) {
		const uint32_t p = tid * TT_PER + k;
		if (p < m && (p == 0 || sg[p] != sg[p - 1])) { const uint32_t g = sg[p]; atomicMax(&red[3], (uint32_t)gs[g + 1] - (uint32_t)gs[g]); }
	}
	__syncthreads();
	const uint32_t gmax = red[3];
	if (xflags & 4) return;
	if (xflags & 1) { /* timing only: the keys stay as they were fetched */ }
	else if (gmax <= (uint32_t)count_max) {
		// ---- every member counts the members of its group in front of it in (key, slot) order ----
		uint64_t mk[TT_PER]; uint32_t mv[TT_PER]; uint32_t dst[TT_PER];
		#pragma unroll
		for (int k = 0; k < TT_PER; ++k) {
			const uint32_t p = tid + k * TT_THREADS;
			dst[k] = 0xFFFFFFFFu;
			if (p < m) {
				const uint32_t g = sg[p], b = gs[g], en = gs[g + 1];
				const uint64_t my = sk[p];
				// (eight keys per step: the loads of a step are in flight together; one key per step waited for the LDS at every member -- 10 of the 22 ms the tiles of
				// a genome-like text's first round took, tools/tt_experiment.py)
				uint32_t less = 0, j = b;
				for (; j + 8u <= en; j += 8u) {
					uint64_t kk[8];
					#pragma unroll
					for (int q = 0; q < 8; ++q) kk[q] = sk[j + q];
					#pragma unroll
					for (int q = 0; q < 8; ++q) less += kk[q] < my ? 1u : 0u;
				}
				for (; j < en; ++j) less += sk[j] < my ? 1u : 0u;
				mk[k] = my; mv[k] = sv[p]; dst[k] = b + less;
			}
		}
		__syncthreads();
		if (!kshift) {
			// 64-bit keys carry no slot: members with EQUAL keys counted the same place.  They stay one group, in any order: each takes the next free slot
			// behind that place (16-bit counters, two per word, in the group-start table, which nobody reads any more) -- one LDS atomic per member instead
			// of a second compare per pair
			uint32_t *eq = reinterpret_cast<uint32_t *>(gs);
			for (uint32_t t = tid; t < (uint32_t)(TT_CAP + 2) / 2u; t += TT_THREADS) eq[t] = 0u;
			__syncthreads();
			#pragma unroll
			for (int k = 0; k < TT_PER; ++k) if (dst[k] != 0xFFFFFFFFu) {
				const uint32_t d = dst[k], old = atomicAdd(&eq[d >> 1], (d & 1u) ? 0x10000u : 1u);
				dst[k] = d + ((d & 1u) ? (old >> 16) : (old & 0xFFFFu));
			}
		}
		#pragma unroll
		for (int k = 0; k < TT_PER; ++k) if (dst[k] != 0xFFFFFFFFu) { sk[dst[k]] = mk[k]; sv[dst[k]] = mv[k]; }
		__syncthreads();
	} else {
		// ---- bitonic network over the tile, keyed (group, key) ----
		for (uint32_t kk = 2; kk <= P; kk <<= 1) {
			for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
				for (uint32_t t = tid; t < (P >> 1); t += TT_THREADS) {
					const uint32_t i1 = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), i2 = i1 | j;
					const bool up = (i1 & kk) == 0u;
					const uint32_t g1 = sg[i1], g2 = sg[i2];
					const uint64_t k1 = sk[i1], k2 = sk[i2];
					const bool gt = g1 > g2 || (g1 == g2 && k1 > k2);
					if (gt == up) {
						sg[i1] = (uint16_t)g2; sg[i2] = (uint16_t)g1; sk[i1] = k2; sk[i2] = k1;
						const uint32_t v1 = sv[i1]; sv[i1] = sv[i2]; sv[i2] = v1;
					}
				}
				__syncthreads();
			}
		}
	}
	// ---- back into the slots; every slot learns the list index of the head of its group as it is now (a tile holds whole groups: no scan over the list is needed) ----
	{
		uint32_t hl[TT_PER], last = 0;                 // (local index of the last head at or in front of the slot) + 1, inside this thread's consecutive slots; 0: none yet
		#pragma unroll
		for (int k = 0; k < TT_PER; ++k) {
			const uint32_t p = tid * TT_PER + k;
			if (p < m) {
				const bool head = p == 0 || sg[p] != sg[p - 1] || (sk[p] >> kshift) != (sk[p - 1] >> kshift);
				if (head) last = p + 1;
			}
			hl[k] = last;
		}
		const uint32_t inc = wave_incl_max(last);
		uint32_t carry = __shfl_up(inc, 1, 64);
		if ((tid & 63u) == 0u) carry = 0u;
		__syncthreads();                               // (wtot: the head counts of the fetch are read by now)
		if ((tid & 63u) == 63u) wtot[tid >> 6] = inc;
		__syncthreads();
		for (uint32_t w = 0; w < (tid >> 6); ++w) carry = carry > wtot[w] ? carry : wtot[w];
		#pragma unroll
		for (int k = 0; k < TT_PER; ++k) {
			const uint32_t p = tid * TT_PER + k;
			if (p < m && !(xflags & 2)) {
				const uint64_t s = (uint64_t)a + p;
				idx[U[s]] = sv[p];
				hid[s] = a + (hl[k] ? hl[k] : carry) - 1u;
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Large groups inside a round: MSD splits on the round's own key.  Two radix sorts of a sub-list cost ten passes whatever its groups look
// like; a group of a repeat family is mostly its consensus, so splitting it by the key's leading byte, then the pieces still above
// MP_GROUP by the next byte, ... touches 1 + 0.66 + 0.43 + ... of it, and whatever falls to MP_GROUP or below is finished by the tiles ON THE
// SAME KEY (so the whole list still advances by one key per round).  A piece that is still large after the key's last byte holds one
// key: it stays a group.
//   K/V, K2/V2: key and suffix of every list slot, in the first pair of buffers or -- MSD_LOC, after an odd number of splits -- in the second;
//   pid[s] = (start slot of the large piece s is in) + 1, or 0, and MSD_BND where a piece starts; tab[row * 256 + d], row = piece start >> 11
//   (pieces are longer than 512 slots: a row has one piece): counts, then places.
// A chunk of 1024 slots meets at most three pieces (one that began in front of it, two that begin inside): three LDS histograms.
// ------------------------------------------------------------------------------------------------
constexpr int MP_CHUNK = 2 << MP_ROW, MP_THREADS = MP_CHUNK / 4, MP_PER = MP_CHUNK / MP_THREADS;
static_assert(MP_CHUNK <= 2 * MP_GROUP && (1 << MP_ROW) <= MP_GROUP, "at most two pieces begin inside a chunk, one piece per table row");
static_assert(MP_GROUP <= TT_COUNT, "what the splits leave is ordered by counting");

template <bool ISA>
__global__ void k_tie_keyvals(TextView T, const uint32_t *U, const uint32_t *idx, uint64_t u, const uint32_t *isa, uint64_t off, uint64_t *K, uint32_t *V)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	const uint32_t i = idx[U[s]];
	uint64_t key;
	if (ISA) { uint64_t q = (uint64_t)i + off; if (q >= T.n) q -= T.n; key = isa[q]; }
	else key = key_at_off(T, i, off);
	K[s] = key; V[s] = i;
}
// pid of the large groups (hid[s] = list index of the group's head); *nlarge += their slots
// (one atomic per workgroup on the counter: one per wave of a 360 M list took 9 ms); act[chunk] = the chunk has slots of pieces: the splits skip the others
__device__ __forceinline__ void msd_block_count(uint32_t mine, uint32_t *total, uint8_t *act)
{
	__shared__ uint32_t acc;
	if (threadIdx.x == 0) acc = 0u;
	__syncthreads();
	const uint32_t w = wave_incl_sum(mine);
	if (lane_id() == 63u && w) atomicAdd(&acc, w);
	__syncthreads();
	if (threadIdx.x == 0) { act[blockIdx.x] = acc ? 1 : 0; if (acc) atomicAdd(total + (blockIdx.x & 63u), acc); }      // (64 counters, summed by the host)
}
// PIECES OF MIDDLE SIZE (above MP_GROUP, up to 8192 slots) leave the split passes: on repeat families a pass by one byte of the key sheds a third of a piece at best
// (copies at 10 % divergence agree on the next four symbols two times in three), so a piece of a few thousand slots took all eight passes over the list before
// (genome-like 3 Gbase: 47 M of the 163 M slots of the second key round, 8 x (count + scatter + relabel)).  Such a piece is listed (first slot | size << 32, the
// buffer its pairs are in) and ONE workgroup orders it completely in LDS (k_msd_piece_sort).
struct PieceList { unsigned long long *rec; uint32_t *cnt; uint32_t cap; uint32_t maxsize; };      // maxsize: the largest piece the sorter takes (2048, 4096 or 8192)
__device__ __forceinline__ void piece_put(const PieceList &P, uint64_t start, uint32_t size, uint32_t loc)
{
	const uint32_t at = atomicAdd(P.cnt, 1u);
	if (at < P.cap) P.rec[at] = (unsigned long long)start | ((unsigned long long)size << 32) | (loc ? 1ull << 63 : 0ull);      // (more than the list holds: the host sees cnt > cap and takes the passes for all)
}
__global__ __launch_bounds__(MP_THREADS) void k_msd_init(const uint32_t *hid, const uint32_t *gid, uint64_t u, uint32_t *pid, uint32_t *nlarge, uint8_t *act, PieceList PL)
{
	uint32_t mine = 0;
	#pragma unroll
	for (int k = 0; k < MP_PER; ++k) {
		const uint64_t s = (uint64_t)blockIdx.x * MP_CHUNK + threadIdx.x + (uint64_t)k * MP_THREADS;
		if (s >= u) continue;
		bool large = tie_large(hid, gid, u, s, MP_GROUP);
		if (large && PL.cap && !tie_large(hid, gid, u, s, PL.maxsize)) {
			// a group of middle size: its head lists it (the size by bisection between MP_GROUP and maxsize members), nobody takes part in the passes
			large = false;
			if (hid[s] == (uint32_t)s) {
				const uint32_t g = gid[s];
				uint64_t lo = s + MP_GROUP, hi = s + PL.maxsize < u ? s + PL.maxsize : u;      // gid[lo] == g; hi: first slot known to be outside (or u)
				while (hi - lo > 1) { const uint64_t mid = lo + (hi - lo) / 2; if (gid[mid] == g) lo = mid; else hi = mid; }
				piece_put(PL, s, (uint32_t)(hi - s), 0u);
			}
		}
		pid[s] = large ? hid[s] + 1u : 0u;
		mine += large ? 1u : 0u;
	}
	msd_block_count(mine, nlarge, act);
}
// the chunk's pieces: ps[0] = start of the one that began in front of the chunk, ps[1] / ps[2] = the first / last start inside
__device__ __forceinline__ void msd_chunk_pieces(const uint32_t (&q)[MP_PER], uint64_t c0, uint32_t *ps)
{
	if (threadIdx.x == 0) { ps[0] = 0u; ps[1] = 0xFFFFFFFFu; ps[2] = 0u; }
	__syncthreads();
	#pragma unroll
	for (int k = 0; k < MP_PER; ++k) if (q[k]) {
		const uint32_t st = q[k] - 1u;
		if ((uint64_t)st < c0) ps[0] = st; else { atomicMin(&ps[1], st); atomicMax(&ps[2], st); }
	}
	__syncthreads();
}
__device__ __forceinline__ uint32_t msd_local_piece(uint32_t st, uint64_t c0, const uint32_t *ps) { return (uint64_t)st < c0 ? 0u : (st == ps[1] ? 1u : 2u); }

__global__ __launch_bounds__(MP_THREADS) void k_msd_count(const uint64_t *K, const uint32_t *pid, uint64_t u, int shift, uint32_t *tab, const uint8_t *act)
{
	__shared__ uint32_t h[3 * 256];
	__shared__ uint32_t ps[3];
	if (!act[blockIdx.x]) return;
	const uint32_t tid = threadIdx.x;
	const uint64_t c0 = (uint64_t)blockIdx.x * MP_CHUNK;
	uint32_t q[MP_PER];
	#pragma unroll
	for (int k = 0; k < MP_PER; ++k) { const uint64_t s = c0 + tid + (uint64_t)k * MP_THREADS; q[k] = s < u ? pid[s] & MSD_PID : 0u; }
	for (uint32_t t = tid; t < 768u; t += MP_THREADS) h[t] = 0u;
	msd_chunk_pieces(q, c0, ps);
	#pragma unroll
	for (int k = 0; k < MP_PER; ++k) if (q[k]) {
		const uint64_t s = c0 + tid + (uint64_t)k * MP_THREADS;
		atomicAdd(&h[msd_local_piece(q[k] - 1u, c0, ps) * 256u + (uint32_t)((K[s] >> shift) & 255u)], 1u);
	}
	__syncthreads();
	for (uint32_t t = tid; t < 768u; t += MP_THREADS) if (h[t]) atomicAdd(&tab[(uint64_t)(ps[t >> 8] >> MP_ROW) * 256u + (t & 255u)], h[t]);
}
// counts -> places inside the piece (exclusive); one wave per table row, four bytes per lane (a row belongs to the piece that starts in its
// 512 slots: the rows of two chunks are those of chunk 2r and 2r + 1)
__global__ __launch_bounds__(256) void k_msd_scan(uint32_t *tab, uint64_t rows, const uint8_t *act, uint64_t nchunk)
{
	const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= rows) return;
	const uint64_t c = r >> 1;                                    // (MP_CHUNK = 2 << MP_ROW)
	if (c < nchunk && !act[c]) return;
	uint4 *row = reinterpret_cast<uint4 *>(tab + r * 256u) + lane_id();
	const uint4 v = *row;
	const uint32_t t = v.x + v.y + v.z + v.w, inc = wave_incl_sum(t);
	uint4 o; o.x = inc - t; o.y = o.x + v.x; o.z = o.y + v.y; o.w = o.z + v.z;
	*row = o;
}
static_assert(MP_CHUNK == (2 << MP_ROW), "k_msd_scan: two table rows per chunk");
// the pairs to their byte's place inside the piece (any order inside it), from one pair of buffers to the other; slots outside the pieces stay where they are
__global__ __launch_bounds__(MP_THREADS) void k_msd_scatter(const uint64_t *K, const uint32_t *V, const uint32_t *pid, uint64_t u, int shift, uint32_t *tab,
                                                            uint64_t *K2, uint32_t *V2, const uint8_t *act)
{
	__shared__ uint32_t h[3 * 256], hb[3 * 256];
	__shared__ uint32_t ps[3];
	if (!act[blockIdx.x]) return;
	const uint32_t tid = threadIdx.x;
	const uint64_t c0 = (uint64_t)blockIdx.x * MP_CHUNK;
	uint32_t q[MP_PER], v[MP_PER]; uint64_t key[MP_PER];
	#pragma unroll
	for (int k = 0; k < MP_PER; ++k) {
		const uint64_t s = c0 + tid + (uint64_t)k * MP_THREADS;
		q[k] = 0u; key[k] = 0; v[k] = 0;
		if (s < u) { q[k] = pid[s] & MSD_PID; if (q[k]) { key[k] = K[s]; v[k] = V[s]; } }
	}
	for (uint32_t t = tid; t < 768u; t += MP_THREADS) h[t] = 0u;
	msd_chunk_pieces(q, c0, ps);
	#pragma unroll
	for (int k = 0; k < MP_PER; ++k) if (q[k]) atomicAdd(&h[msd_local_piece(q[k] - 1u, c0, ps) * 256u + (uint32_t)((key[k] >> shift) & 255u)], 1u);
	__syncthreads();
	for (uint32_t t = tid; t < 768u; t += MP_THREADS) {
		const uint32_t c = h[t];
		hb[t] = c ? atomicAdd(&tab[(uint64_t)(ps[t >> 8] >> MP_ROW) * 256u + (t & 255u)], c) : 0u;
		h[t] = 0u;
	}
	__syncthreads();
	#pragma unroll
	for (int k = 0; k < MP_PER; ++k) {
		const uint64_t s = c0 + tid + (uint64_t)k * MP_THREADS;
		if (s >= u || !q[k]) continue;
		const uint32_t st = q[k] - 1u, t = msd_local_piece(st, c0, ps) * 256u + (uint32_t)((key[k] >> shift) & 255u);
		const uint64_t dest = (uint64_t)st + hb[t] + atomicAdd(&h[t], 1u);
		K2[dest] = key[k]; V2[dest] = v[k];
	}
}
// after the scatter the table holds the END of every byte's range (= the start of the next byte's): the new pieces are the ranges above TT_GROUP
// (K: where this split wrote; loc: MSD_LOC if that is the second pair of buffers)
__global__ __launch_bounds__(MP_THREADS) void k_msd_repid(const uint64_t *K, uint32_t *pid, uint64_t u, int shift, const uint32_t *tab, uint32_t loc, uint32_t *nleft, uint8_t *act, PieceList PL)
{
	if (!act[blockIdx.x]) return;
	uint32_t mine = 0;
	#pragma unroll
	for (int k = 0; k < MP_PER; ++k) {
		const uint64_t s = (uint64_t)blockIdx.x * MP_CHUNK + threadIdx.x + (uint64_t)k * MP_THREADS;
		if (s >= u) continue;
		const uint32_t p = pid[s], q = p & MSD_PID;
		if (!q) continue;
		const uint32_t st = q - 1u, d = (uint32_t)((K[s] >> shift) & 255u);
		const uint32_t *row = tab + (uint64_t)(st >> MP_ROW) * 256u;
		const uint32_t e = row[d], b = d ? row[d - 1] : 0u;
		bool large = e - b > (uint32_t)MP_GROUP;
		if (large && shift > 0 && PL.cap && e - b <= PL.maxsize) {      // a new piece of middle size: listed by its first slot, out of the passes (after the key's last byte a piece holds one key: it stays)
			large = false;
			if (s == (uint64_t)st + b) piece_put(PL, s, e - b, loc);
		}
		pid[s] = (large ? st + b + 1u : 0u) | (p & MSD_BND) | (s == (uint64_t)st + b ? MSD_BND : 0u) | loc;
		mine += large ? 1u : 0u;
	}
	msd_block_count(mine, nleft, act);
}
// one workgroup per listed piece: its pairs ordered by the whole key in LDS (bitonic network over the next power of two, padded with keys above every real one),
// written back where they were; then the labels the passes would have left: a boundary where the key changes, and a run of one key above MP_GROUP slots keeps
// its first slot as its piece (k_msd_leftover makes it a group; k_tie_tile<2> takes the rest)
constexpr uint32_t PS_PAD = 0xFFFFFFFFu;               // the suffix of a padding entry (positions are below 2^32 - 1): orders it behind a real entry with the same key (pac keys use all 64 bits)
template <int CAP, int PS_THREADS>
__global__ __launch_bounds__(PS_THREADS) void k_msd_piece_sort(PieceList PL, uint32_t npieces, uint64_t *K0, uint32_t *V0, uint64_t *K1, uint32_t *V1, uint32_t *pid)
{
	__shared__ uint64_t sk[CAP];
	__shared__ uint32_t sv[CAP];
	const uint32_t tid = threadIdx.x;
	for (uint32_t pi = blockIdx.x; pi < npieces; pi += gridDim.x) {
		const unsigned long long rec = PL.rec[pi];
		const uint64_t st = rec & 0xFFFFFFFFull;
		const uint32_t m = (uint32_t)(rec >> 32) & 0x7FFFFFFFu, loc = (uint32_t)(rec >> 63);
		uint64_t *K = loc ? K1 : K0; uint32_t *V = loc ? V1 : V0;
		uint32_t P = 2 * PS_THREADS; while (P < m) P <<= 1;
		for (uint32_t j = tid; j < P; j += PS_THREADS) { sk[j] = j < m ? K[st + j] : ~0ull; sv[j] = j < m ? V[st + j] : PS_PAD; }
		__syncthreads();
		for (uint32_t k = 2; k <= P; k <<= 1)
			for (uint32_t jj = k >> 1; jj > 0; jj >>= 1) {
				for (uint32_t t = tid; t < (P >> 1); t += PS_THREADS) {
					const uint32_t i = ((t & ~(jj - 1u)) << 1) | (t & (jj - 1u)), l = i | jj;      // i: bit jj clear; partner l
					const bool up = (i & k) == 0u;
					const uint64_t a = sk[i], b = sk[l];
					const uint32_t va = sv[i], vb = sv[l];
					const bool gt = a > b || (a == b && va == PS_PAD && vb != PS_PAD);      // (padding behind everything real)
					if (gt == up && (a != b || va == PS_PAD || vb == PS_PAD)) { sk[i] = b; sk[l] = a; sv[i] = vb; sv[l] = va; }
				}
				__syncthreads();
			}
		// runs of one key: the first slot of slot j's run by a backward look (runs are short unless the repeat goes on: bounded by a scan over 64-slot blocks)
		for (uint32_t j = tid; j < m; j += PS_THREADS) { K[st + j] = sk[j]; V[st + j] = sv[j]; }
		__syncthreads();
		// sv is free now: sv[j] = j where a run begins, else 0 -> inclusive max-scan = the run's first slot
		for (uint32_t j = tid; j < P; j += PS_THREADS) sv[j] = (j < m && (j == 0 || sk[j] != sk[j - 1])) ? j : 0u;
		__syncthreads();
		for (uint32_t d = 1; d < P; d <<= 1) {
			uint32_t tmp[CAP / PS_THREADS];
			#pragma unroll
			for (uint32_t q = 0; q < (uint32_t)(CAP / PS_THREADS); ++q) { const uint32_t j = tid + q * PS_THREADS; tmp[q] = (j < P && j >= d) ? max(sv[j], sv[j - d]) : (j < P ? sv[j] : 0u); }
			__syncthreads();
			#pragma unroll
			for (uint32_t q = 0; q < (uint32_t)(CAP / PS_THREADS); ++q) { const uint32_t j = tid + q * PS_THREADS; if (j < P) sv[j] = tmp[q]; }
			__syncthreads();
		}
		// the run's end: the next run's first slot, found from the run starts (a run start j knows its predecessor run ended at j): sk reused as "end of the run that starts here"
		for (uint32_t j = tid; j < m; j += PS_THREADS) {
			const bool last_of_run = j + 1 == m || sv[j + 1] != sv[j];
			if (last_of_run) reinterpret_cast<uint32_t *>(sk)[sv[j]] = j + 1u;      // (32-bit view of the key array: the keys have left)
		}
		__syncthreads();
		for (uint32_t j = tid; j < m; j += PS_THREADS) {
			const uint32_t rs = sv[j], re = reinterpret_cast<uint32_t *>(sk)[rs];
			const bool big = re - rs > (uint32_t)MP_GROUP;
			pid[st + j] = (big ? (uint32_t)(st + rs) + 1u : 0u) | (j == rs ? MSD_BND : 0u) | (loc ? MSD_LOC : 0u);
		}
		__syncthreads();
	}
}
// pieces still large after the key's last byte hold one key each: they stay groups
__global__ void k_msd_leftover(const uint32_t *U, const uint32_t *gid, const uint32_t *pid, const uint32_t *V, const uint32_t *V1, uint64_t u, uint32_t *idx, uint32_t *hid)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u || !(pid[s] & MSD_PID)) return;
	idx[U[s]] = (pid[s] & MSD_LOC) ? V1[s] : V[s];
	hid[s] = (pid[s] & MSD_PID) - 1u;              // the piece of one key is a group: its first slot is the head of all its members
}

// ------------------------------------------------------------------------------------------------
// K5: products
// ------------------------------------------------------------------------------------------------
__global__ void k_emit_bwt(TextView T, const uint32_t *sa, uint64_t len, uint64_t start, uint8_t *bwt, uint32_t *p0)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= len) return;
	uint64_t i = sa[r];
	uint64_t j = i ? i - 1 : T.n - 1;
	if (bwt) { if (T.type == B3M_INPUT_SYMBOLS32) reinterpret_cast<uint32_t *>(bwt)[r] = sym_at(T, j); else bwt[r] = (uint8_t)sym_at(T, j); }
	if (i == start) *p0 = (uint32_t)r;
}
// gt bit of absolute position x: [S_x > S_start], for the block's own positions (x != start)
__global__ void k_emit_gt(const uint32_t *sa, uint64_t len, uint64_t start, uint32_t p0, uint32_t *gt)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= len) return;
	uint32_t x = sa[r];
	if (r > p0 && x != start) atomicOr(&gt[x >> 5], 1u << (x & 31));
}
// gt bits without a scatter: bit x = [S_x > S_start] is a property of the TEXT, so it is computed
// by streaming the packed text: one thread per 16 positions compares every rotation with the
// block's first one (first 32 bases in one window; the full comparison only if those tie).
// Writes 16-bit halves of the absolute-position bit array: no atomics.
__global__ __launch_bounds__(256) void k_gt_from_text(TextView T, uint64_t start, uint64_t len, uint16_t *gt16)
{
	uint64_t w = (start >> 4) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t pos0 = w << 4;
	const uint64_t end = start + len;             // bits for x in (start, end) ; bit `end` is set by the caller
	if (pos0 >= end) return;
	const bool fast0 = start + 32 <= T.nbases;
	const uint64_t ref = fast0 ? window32(T, start) : 0;
	bool fast = fast0 && pos0 + 16 + 32 <= T.nbases;
	uint64_t w0 = 0, w1 = 0, w2 = 0;
	if (fast) { const uint8_t *p = T.b + (w << 2); w0 = be32(p); w1 = be32(p + 4); w2 = be32(p + 8); }
	uint64_t hi = (w0 << 32) | w1;
	uint32_t bits = 0, keep = 0;                  // keep: positions of this half-word outside (start, end)
	#pragma unroll
	for (int s = 0; s < 16; ++s) {
		uint64_t x = pos0 + s;
		if (x <= start || x >= end) { keep |= 1u << s; continue; }
		int c;
		if (fast) {
			uint64_t wx = s ? ((hi << (2 * s)) | (w2 >> (32 - 2 * s))) : hi;
			c = wx != ref ? (wx > ref ? 1 : -1) : compare_rotations(T, x, start);
		} else c = compare_rotations(T, x, start);
		if (c > 0) bits |= 1u << s;
	}
	uint16_t old = keep ? gt16[w] : (uint16_t)0;  // ragged ends of the block: keep the neighbours' bits
	gt16[w] = (uint16_t)((old & keep) | bits);
}
void emit_gt_from_text(const TextView &T, uint64_t start, uint64_t len, uint32_t *d_gt_abs, hipStream_t s)
{
	if (!len) return;
	uint64_t words = ((start + len + 15) >> 4) - (start >> 4);
	hipLaunchKernelGGL(k_gt_from_text, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, s, T, start, len, reinterpret_cast<uint16_t *>(d_gt_abs));
}

// (rank,pos) samples: the sample of position x = k*rate has the fixed slot k - first: no counter
__global__ void k_emit_samples(const uint32_t *sa, const uint8_t *sa_hi, uint64_t len, uint64_t rate, uint64_t rank_off, uint64_t *pairs, uint64_t first, uint64_t cap)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= len) return;
	const uint64_t x = (uint64_t)sa[r] | (sa_hi ? (uint64_t)sa_hi[r] << 32 : 0ull);
	const uint32_t r32 = (uint32_t)rate;
	bool hit; uint64_t q;
	if ((r32 & (r32 - 1)) == 0) { hit = (x & (r32 - 1)) == 0; q = x >> (31 - __clz(r32)); }     // power of two: no division
	else { q = x / r32; hit = q * r32 == x; }
	if (hit) {
		uint64_t k = q - first;
		if (k < cap) { pairs[2 * k] = r + rank_off; pairs[2 * k + 1] = x; }
	}
}
// rank of S_x among the sorted block: lower bound by rotation comparison; one thread per query
__global__ void k_rank_of(TextView T, const uint32_t *sa, uint64_t len, const uint64_t *pos, uint64_t count, uint32_t *accum)
{
	uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= count) return;
	uint64_t x = pos[q] % T.n;
	uint64_t lo = 0, hi = len;
	while (lo < hi) {
		uint64_t mid = (lo + hi) >> 1;
		uint64_t y = sa[mid];
		int c = (y == x) ? 0 : compare_rotations(T, y, x);
		if (c < 0) lo = mid + 1; else hi = mid;
	}
	atomicAdd(&accum[q], (uint32_t)lo);
}
__global__ __launch_bounds__(256) void k_histogram(TextView T, uint64_t start, uint64_t len, unsigned long long *hist)
{
	__shared__ uint32_t h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < len; r += stride)
		atomicAdd(&h[sym_at(T, start + r)], 1u);
	__syncthreads();
	if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline uint32_t nblk(uint64_t n, uint32_t t = 256) { return (uint32_t)((n + t - 1) / t); }
static inline uint64_t al(uint64_t x) { return (x + 255) & ~255ull; }
static inline bool msd_applicable_len(uint64_t cap) { return cap >= (1ull << 18); }      // (MSD_MIN_LEN of k_msd.hip: smaller sorts take the key/index path, which lists ties always)

// compactstream (row a3/f4, role of CompactInputTypes::linear_wrapper): symbol i = `bits` bits at bit offset i*bits of the
// big-endian word stream, most significant bit first -> one byte per symbol (alphabets above 4 symbols take the byte paths)
__global__ void k_compact_decode(const uint8_t *words, uint64_t n, uint32_t bits, uint8_t *out)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t bo = i * bits, by = bo >> 3; const uint32_t sh = (uint32_t)(bo & 7);
	const uint32_t two = ((uint32_t)words[by] << 8) | (uint32_t)words[by + 1];     // the file is padded to whole 64-bit words (+ 8 bytes here)
	out[i] = (uint8_t)((two >> (16 - sh - bits)) & ((1u << bits) - 1u));
}
void compact_decode(const uint8_t *d_words, uint64_t n, uint32_t bits, uint8_t *d_out, hipStream_t s)
{
	if (n) hipLaunchKernelGGL(k_compact_decode, dim3(nblk(n)), dim3(256), 0, s, d_words, n, bits, d_out);
}

static uint64_t big_tie_cap(uint64_t cap) { return cap / 16 + (1ull << 20); }
uint64_t blocksort_ws_bytes(uint64_t cap, uint64_t span, bool lazy_ties)
{
	if (span < cap) span = cap;
	if (span >= (1ull << 32) || !msd_applicable_len(cap)) lazy_ties = false;
	uint64_t b = 0;
	b += 2 * al(8 * cap) + 2 * al(4 * cap);           // key/record and idx ping-pong
	if (span >= (1ull << 32)) b += al(cap);           // bits 32.. of the suffix array entries
	// texts of 2^32 symbols and more: the tie bitmap, and list buffers for cap/16 tied suffixes (key-extension rounds with 36-bit positions;
	// no inverse suffix array: 5 bytes per TEXT symbol would not fit beside a shard's workspace)
	if (span >= (1ull << 32)) b += al(cap / 8 + 64) + 5 * al(4 * (big_tie_cap(cap) + 1)) + al(big_tie_cap(cap) + 1);
	else b += (lazy_ties ? 1 : 6) * al(4 * (cap + 1));  // tie-list buffers + inverse suffix array of the doubling rounds (lazy: the tie bitmap's buffer only)
	b += al(4 * scan_ws_elems(cap + 1));
	b += al(radix_ws_bytes(cap));
	b += al(msd_ws_bytes(cap, span));
	b += 4096;
	return b;
}
void blocksort_ws_carve(BlockSortWs &ws, void *base, uint64_t cap, uint64_t span, bool lazy_ties)
{
	if (span < cap) span = cap;
	if (span >= (1ull << 32) || !msd_applicable_len(cap)) lazy_ties = false;
	blocksort_release(ws);
	ws.lazy_ties = lazy_ties;
	uint8_t *p = (uint8_t *)base;
	ws.cap = cap;
	ws.keyA = (uint64_t *)p; p += al(8 * cap);
	ws.keyB = (uint64_t *)p; p += al(8 * cap);
	ws.idxA = (uint32_t *)p; p += al(4 * cap);
	ws.idxB = (uint32_t *)p; p += al(4 * cap);
	ws.sa_hi = nullptr;
	if (span >= (1ull << 32)) { ws.sa_hi = p; p += al(cap); }
	ws.tie_cap = cap; ws.tie_hi = nullptr;
	if (span >= (1ull << 32)) {
		ws.t[0] = (uint32_t *)p; p += al(cap / 8 + 64);
		ws.tie_cap = big_tie_cap(cap);
		for (int i = 1; i < 6; ++i) { ws.t[i] = (uint32_t *)p; p += al(4 * (ws.tie_cap + 1)); }
		ws.tie_hi = p; p += al(ws.tie_cap + 1);
	} else if (lazy_ties) { for (int i = 1; i < 6; ++i) ws.t[i] = nullptr; ws.t[0] = (uint32_t *)p; p += al(4 * (cap + 1)); }
	else for (int i = 0; i < 6; ++i) { ws.t[i] = (uint32_t *)p; p += al(4 * (cap + 1)); }
	ws.scan_ws = (uint32_t *)p; p += al(4 * scan_ws_elems(cap + 1));
	radix_ws_carve(ws.radix, p, cap); p += al(radix_ws_bytes(cap));
	msd_ws_carve(ws.msd, p, cap, span); p += al(msd_ws_bytes(cap, span));
	ws.d_scalars = (uint32_t *)p;
}

void blocksort_ensure_ties(BlockSortWs &ws, uint64_t need)
{
	if (ws.sa_hi) {
		// texts of 2^32 symbols and more: list buffers for a sixteenth of the sort are part of the workspace (a text with few repeats never needs more); a repeat-rich one
		// (a tenth to a half of a genome's suffixes share their first 16 bases with 32 others) gets larger ones on demand, kept for the plan's later runs
		if (need <= ws.tie_cap) return;
		const uint64_t c = std::min<uint64_t>(ws.cap, need + need / 8 + (1ull << 20));
		const uint64_t one = al(4 * (c + 1));
		void *a = nullptr;
		if (hipMalloc(&a, 5 * one + al(c + 1)) != hipSuccess) {
			(void)hipGetLastError();
			throw Error("repeat-rich input of 2^32 symbols and more: " + std::to_string(need) + " suffixes are left to the list-based rounds and there is no HBM left for their buffers (" +
			            std::to_string((5 * one + c) / 1000000) + " MB beside the plan's workspace): use more shards (a smaller mem=)");
		}
		if (ws.tie_alloc) (void)hipFree(ws.tie_alloc);
		ws.tie_alloc = a;
		for (int i = 1; i < 6; ++i) ws.t[i] = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(a) + (uint64_t)(i - 1) * one);
		ws.tie_hi = static_cast<uint8_t *>(a) + 5 * one;
		ws.tie_cap = c;
		return;
	}
	if (!ws.lazy_ties || ws.t[1]) return;
	const uint64_t one = al(4 * (ws.cap + 1));
	if (hipMalloc(&ws.tie_alloc, 5 * one) != hipSuccess) {
		(void)hipGetLastError();
		ws.tie_alloc = nullptr;
		throw Error("repeat-rich input: no HBM left for the buffers of the list-based rounds (" + std::to_string(5 * one / 1000000) + " MB beside the plan's workspace): use blocks (blocksize= / mem=)");
	}
	for (int i = 1; i < 6; ++i) ws.t[i] = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(ws.tie_alloc) + (uint64_t)(i - 1) * one);
}
void blocksort_release(BlockSortWs &ws)
{
	if (ws.tie_alloc) { (void)hipFree(ws.tie_alloc); ws.tie_alloc = nullptr; for (int i = 1; i < 6; ++i) ws.t[i] = nullptr; if (ws.sa_hi) { ws.tie_hi = nullptr; ws.tie_cap = 0; } }
}

// Rounds on the list (U, gid) of u tied slots.  idx = the suffix array being built.  Buffers: kx/ky
// (u64, cap >= u) and six u32 buffers of cap+1: B[0]=U, B[1]=gid, B[2..3] sort values, B[4]=hid,
// B[5]=flag (B is permuted in place: on return B[0]/B[1] hold the list that is left).
//   isa == nullptr: key-extension rounds: round r sorts every group by the symbols [(depth0+r)*K, +K)
//                   of its rotations; at most max_rounds of them (0 = until nothing is left).
//   isa != nullptr: prefix doubling from h0 symbols (every listed group must agree on >= h0 symbols,
//                   isa must be initialised: whole-text blocks only): round r sorts by the rank of
//                   the suffix h0*2^r further on and refreshes the ranks of the listed suffixes.
// Returns the number of rounds run; *left = tied slots remaining.
// one round by radix sorts on the list (Ux, gidx) of cnt slots: by key, then stably by group label; the suffixes go back into their group's slots and
// the new heads into hid (at lpos[.] when the list is a sub-list of the one hid belongs to)
static void tie_round_lsd(const TextView &T, uint32_t *idx, uint8_t *idx_hi, uint8_t *tmp_hi, const uint32_t *Ux, const uint32_t *gidx, uint64_t cnt,
                          uint64_t *kx, uint64_t *ky, uint32_t *vx, uint32_t *vy, uint32_t *hid, const uint32_t *lpos,
                          const uint32_t *isa, uint64_t h, uint64_t off, int gbits, RadixWs &radix, hipStream_t s)
{
	if (isa) hipLaunchKernelGGL(k_tie_keys_isa, dim3(nblk(cnt)), dim3(256), 0, s, isa, T.n, h, Ux, idx, cnt, kx, vx);
	else hipLaunchKernelGGL(k_tie_keys, dim3(nblk(cnt)), dim3(256), 0, s, T, Ux, idx, (const uint8_t *)idx_hi, cnt, off, kx, vx);
	uint64_t *rk; uint32_t *rv;
	radix_sort_u64(kx, vx, ky, vy, cnt, isa ? gbits : key_off_bits(T.type), radix, s, nullptr, &rk, &rv);
	uint64_t *gk = (rk == kx) ? ky : kx;
	uint32_t *vidle = (rv == vx) ? vy : vx;
	hipLaunchKernelGGL(k_tie_gidkeys, dim3(nblk(cnt)), dim3(256), 0, s, gidx, rv, cnt, gk);
	uint64_t *rk2; uint32_t *rv2;
	radix_sort_u64(gk, rv, rk, vidle, cnt, gbits, radix, s, nullptr, &rk2, &rv2);
	uint64_t *tmp_key = (rk2 == gk) ? rk : gk;
	uint32_t *tmp_idx = (rv2 == rv) ? vidle : rv;
	if (isa) hipLaunchKernelGGL(k_tie_gather_isa, dim3(nblk(cnt)), dim3(256), 0, s, isa, T.n, h, Ux, idx, rv2, cnt, tmp_idx, tmp_key);
	else hipLaunchKernelGGL(k_tie_gather, dim3(nblk(cnt)), dim3(256), 0, s, T, Ux, idx, (const uint8_t *)idx_hi, rv2, cnt, off, tmp_idx, tmp_hi, tmp_key);
	hipLaunchKernelGGL(k_tie_apply, dim3(nblk(cnt)), dim3(256), 0, s, Ux, gidx, tmp_idx, (const uint8_t *)tmp_hi, tmp_key, cnt, idx, idx_hi, hid, lpos);
}

uint64_t tie_rounds(const TextView &T, uint32_t *idx, uint64_t u, uint64_t depth0, uint64_t *kx, uint64_t *ky, uint32_t *B[6],
                    uint32_t *scan_ws, RadixWs &radix, uint64_t len_for_labels, hipStream_t s, uint64_t *left,
                    uint64_t max_rounds, uint32_t *isa, uint64_t h0, uint8_t *idx_hi, uint8_t *tmp_hi, uint64_t sym_off, bool isa_sparse, bool labels_are_heads)
{
	B3M_REQUIRE(!(idx_hi && isa), "internal: prefix doubling is not available for positions above 2^32");
	B3M_REQUIRE(!idx_hi || tmp_hi, "internal: tie rounds with 36-bit positions need a byte buffer");
	const uint64_t K = (uint64_t)key_symbols(T.type);
	int gbits = 1; while ((1ull << gbits) < len_for_labels + 1 && gbits < 32) ++gbits;
	uint32_t *U = B[0], *gid = B[1], *vx = B[2], *vy = B[3], *hid = B[4], *flag = B[5];
	uint64_t rounds = 0;
	uint32_t h_u = 0;
	uint64_t h = h0;
	// labels: any values that are equal inside a group and differ between neighbouring groups -- or, from the first compaction of this function on (and where the caller
	// says so), the LIST INDEX OF THE GROUP'S HEAD: what the rounds need of a group (is it large, where does it start) is then read from the label, without a scan over the list
	bool heads = labels_are_heads;
	auto head_index = [&]() -> const uint32_t * {         // list index of the head of every slot's group
		if (heads) return gid;
		hipLaunchKernelGGL(k_heads_from_gid, dim3(nblk(u)), dim3(256), 0, s, gid, u, hid);
		inclusive_max_u32(hid, u, scan_ws, s);
		return hid;
	};
	if (isa && u) hipLaunchKernelGGL(k_isa_list, dim3(nblk(u)), dim3(256), 0, s, U, head_index(), idx, u, isa);
	const int dbg = (int)sw().DEBUG.get(0);
	// tiles (k_tie_tile): 32-bit positions; B3M_NO_TIE_TILES=1 keeps every round on the radix sorts
	const bool tiles_ok = !idx_hi && !sw().NO_TIE_TILES.on();
	// large groups split inside the round (k_msd_*); B3M_NO_TIE_MSD=1: through the radix sorts as a sub-list
	const bool msd_on = !sw().NO_TIE_MSD.on();
	bool any_large = true;                             // (groups only shrink: once none is large, none will be)
	for (uint64_t depth = depth0; u > 0; ++depth) {
		const uint64_t off = sym_off + depth * K;       // symbols in front of this round's key
		if (isa ? h >= T.n : off >= T.n) break;         // compared >= n symbols: equal rotations
		if (max_rounds && rounds >= max_rounds) break;
		++rounds;
		const auto t_round = std::chrono::steady_clock::now();
		double ms_splits = 0;
		if (dbg >= 3) {      // census of the tie groups entering this round: elements in groups of size < 2^k
			std::vector<uint32_t> g(u);
			B3M_HIP(hipMemcpyAsync(g.data(), gid, 4ull * u, hipMemcpyDeviceToHost, s)); B3M_HIP(hipStreamSynchronize(s));
			uint64_t by[34] = {0}, ng = 0, big = 0;
			for (uint64_t a = 0; a < u;) { uint64_t b = a + 1; while (b < u && g[b] == g[a]) ++b; int k = 0; while ((1ull << k) < b - a + 1) ++k; by[k] += b - a; ++ng; if (b - a > big) big = b - a; a = b; }
			fprintf(stderr, "[b3m] tie round %llu (%s, depth %llu): %llu listed in %llu groups, largest %llu; elements by group size <2^k:", (unsigned long long)rounds, isa ? "doubling" : "keys",
			        (unsigned long long)(isa ? h : off), (unsigned long long)u, (unsigned long long)ng, (unsigned long long)big);
			for (int k = 0; k < 34; ++k) if (by[k]) fprintf(stderr, " %d:%llu", k, (unsigned long long)by[k]);
			fprintf(stderr, "\n");
		}
		if (isa && isa_sparse) {
			// targets without a rank yet (see k_isa_slots): listed in kx, found in the suffix array
			uint32_t *Q = reinterpret_cast<uint32_t *>(kx), *qcnt = flag, nq = 0;
			B3M_HIP(hipMemsetAsync(qcnt, 0, 4, s));
			hipLaunchKernelGGL(k_isa_need, dim3(nblk(u)), dim3(256), 0, s, (const uint32_t *)isa, T.n, h, U, idx, u, Q, qcnt);
			B3M_HIP(hipMemcpyAsync(&nq, qcnt, 4, hipMemcpyDeviceToHost, s));
			B3M_HIP(hipStreamSynchronize(s));
			if (dbg >= 3) fprintf(stderr, "[b3m]   doubling by %llu: %u of %llu targets have no rank yet\n", (unsigned long long)h, nq, (unsigned long long)u);
			const uint64_t nq_max = sw().ISA_SEARCH_MAX.set ? (uint64_t)sw().ISA_SEARCH_MAX.v : T.n / 64;      // (switch: 0 = fill the array as soon as one target is unset)
			if (nq > nq_max) {
				isa_from_sa(idx, len_for_labels, isa, s);
				hipLaunchKernelGGL(k_heads_from_gid, dim3(nblk(u)), dim3(256), 0, s, gid, u, hid);
				inclusive_max_u32(hid, u, scan_ws, s);
				hipLaunchKernelGGL(k_isa_list, dim3(nblk(u)), dim3(256), 0, s, U, hid, idx, u, isa);
				isa_sparse = false;
			} else if (nq) hipLaunchKernelGGL(k_isa_search, dim3(nblk(nq)), dim3(256), 0, s, T, (const uint32_t *)idx, T.n, (const uint32_t *)Q, nq, isa);
		}
		const uint32_t ntile = (uint32_t)((u + TT_STRIDE - 1) / TT_STRIDE);
		bool used_lsd = false;
		const int kshift = (isa || key_off_bits(T.type) <= 52) ? 12 : 0;      // (k_tie_tile: the slot in the key's low bits)
		if (tiles_ok && msd_on && u >= 8192 && u < (1ull << 30)) {
			// ---- tiles; large groups are split by the key's bytes first ----
			uint32_t *pid = flag, *tab = hid;               // (hid is written last, by the tiles; flag is free until the round's compaction)
			uint32_t *cnt = radix.tile_hist;                 // 64 words (the radix sorts' tile histograms: 256 words per 4096 keys, idle in such a round)
			const uint64_t rows = (u >> MP_ROW) + 2;
			const uint32_t nchunk = (uint32_t)((u + MP_CHUNK - 1) / MP_CHUNK);
			uint8_t *act = reinterpret_cast<uint8_t *>(radix.tile_hist + 66);   // a byte per chunk
			// the list of middle-sized pieces (k_msd_piece_sort) behind the chunk flags: at most one piece per MP_GROUP slots
			PieceList PL{};
			PL.cnt = cnt + 64;
			PL.rec = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(act) + (((uint64_t)nchunk + 15) & ~15ull) + 8);
			PL.rec = reinterpret_cast<unsigned long long *>(((uintptr_t)PL.rec + 7) & ~(uintptr_t)7);
			PL.cap = sw().NO_PIECE_SORT.on() ? 0u : (uint32_t)(u / MP_GROUP + 2);
			PL.maxsize = 4096;      // (genome-like 3 Gbase, one process: no piece sort 179.1 ms, 8192 slots / 1024 threads 182.2, 4096 / 512 172.3, 2048 / 256 173.9: profiles/r05_ab_piece.log)
			if (sw().PIECE_CAP.set) PL.maxsize = sw().PIECE_CAP.v <= 2048 ? 2048u : sw().PIECE_CAP.v <= 4096 ? 4096u : 8192u;
			uint32_t h_n = 0, h_c[65], n_pieces = 0;
			auto read_count = [&]() { B3M_HIP(hipMemcpyAsync(h_c, cnt, 260, hipMemcpyDeviceToHost, s)); B3M_HIP(hipStreamSynchronize(s)); uint64_t t = 0; for (int i = 0; i < 64; ++i) t += h_c[i];
			                          n_pieces = h_c[64]; return (uint32_t)std::min<uint64_t>(t, 0xFFFFFFFFull); };
			if (any_large) {
				const uint32_t *hix = head_index();
				B3M_HIP(hipMemsetAsync(cnt, 0, 260, s));
				hipLaunchKernelGGL(k_msd_init, dim3(nchunk), dim3(MP_THREADS), 0, s, hix, (const uint32_t *)gid, u, pid, cnt, act, PL);
				h_n = read_count();
				B3M_REQUIRE(n_pieces <= PL.cap, "internal: more pieces of middle size than slots allow");
				if (h_n == 0 && n_pieces == 0) any_large = false;
			}
			if (h_n == 0 && n_pieces == 0) {
				if (isa) hipLaunchKernelGGL(k_tie_tile<1>, dim3(ntile), dim3(TT_THREADS), 0, s, T, U, gid, u, idx, (const uint32_t *)isa, h, hid, (const uint64_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint64_t *)nullptr, (const uint32_t *)nullptr, kshift, tt_count_max());
				else hipLaunchKernelGGL(k_tie_tile<0>, dim3(ntile), dim3(TT_THREADS), 0, s, T, U, gid, u, idx, (const uint32_t *)nullptr, off, hid, (const uint64_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint64_t *)nullptr, (const uint32_t *)nullptr, kshift, tt_count_max());
			} else {
				if (isa) hipLaunchKernelGGL(k_tie_keyvals<true>, dim3(nblk(u)), dim3(256), 0, s, T, U, idx, u, (const uint32_t *)isa, h, kx, vx);
				else hipLaunchKernelGGL(k_tie_keyvals<false>, dim3(nblk(u)), dim3(256), 0, s, T, U, idx, u, (const uint32_t *)nullptr, off, kx, vx);
				int cur = 0;                                  // the pieces' pairs are in (kx, vx) / (ky, vy)
				const int kb = isa ? 32 : (key_off_bits(T.type) + 7) / 8 * 8;
				uint32_t passes = 0;
				for (int shift = kb - 8; shift >= 0 && h_n; shift -= 8) {
					B3M_HIP(hipMemsetAsync(tab, 0, 4ull * rows * 256, s));
					uint64_t *K0 = cur ? ky : kx, *K1 = cur ? kx : ky; uint32_t *V0 = cur ? vy : vx, *V1 = cur ? vx : vy;
					hipLaunchKernelGGL(k_msd_count, dim3(nchunk), dim3(MP_THREADS), 0, s, (const uint64_t *)K0, (const uint32_t *)pid, u, shift, tab, (const uint8_t *)act);
					hipLaunchKernelGGL(k_msd_scan, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, s, tab, rows, (const uint8_t *)act, (uint64_t)nchunk);
					hipLaunchKernelGGL(k_msd_scatter, dim3(nchunk), dim3(MP_THREADS), 0, s, (const uint64_t *)K0, (const uint32_t *)V0, (const uint32_t *)pid, u, shift, tab, K1, V1, (const uint8_t *)act);
					cur ^= 1;
					B3M_HIP(hipMemsetAsync(cnt, 0, 256, s));
					hipLaunchKernelGGL(k_msd_repid, dim3(nchunk), dim3(MP_THREADS), 0, s, (const uint64_t *)K1, pid, u, shift, (const uint32_t *)tab, cur ? MSD_LOC : 0u, cnt, act, PL);
					h_n = read_count();
					B3M_REQUIRE(n_pieces <= PL.cap, "internal: more pieces of middle size than slots allow");
					++passes;
				}
				if (n_pieces) {
					const dim3 grid(std::min<uint32_t>(n_pieces, 16384u));
					if (PL.maxsize <= 2048) hipLaunchKernelGGL((k_msd_piece_sort<2048, 256>), grid, dim3(256), 0, s, PL, n_pieces, kx, vx, ky, vy, pid);
					else if (PL.maxsize <= 4096) hipLaunchKernelGGL((k_msd_piece_sort<4096, 512>), grid, dim3(512), 0, s, PL, n_pieces, kx, vx, ky, vy, pid);
					else hipLaunchKernelGGL((k_msd_piece_sort<8192, 1024>), grid, dim3(1024), 0, s, PL, n_pieces, kx, vx, ky, vy, pid);
				}
				ms_splits = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_round).count();
				if (dbg >= 3) fprintf(stderr, "[b3m]   large groups: %u split passes, %u slots in pieces of one key, %u pieces of middle size ordered in LDS\n", passes, h_n, n_pieces);
				// (the table in hid is dead: the tiles and the pieces of one key write the heads there)
				hipLaunchKernelGGL(k_tie_tile<2>, dim3(ntile), dim3(TT_THREADS), 0, s, T, U, gid, u, idx, (const uint32_t *)nullptr, off, hid, (const uint64_t *)kx, (const uint32_t *)vx, (const uint32_t *)pid,
				                   (const uint64_t *)ky, (const uint32_t *)vy, kshift, tt_count_max());
				if (h_n || n_pieces) hipLaunchKernelGGL(k_msd_leftover, dim3(nblk(u)), dim3(256), 0, s, (const uint32_t *)U, (const uint32_t *)gid, (const uint32_t *)pid, (const uint32_t *)vx, (const uint32_t *)vy, u, idx, hid);
			}
		} else {
		// ---- which slots belong to groups too large for a tile ----
		uint64_t nL = u;
		const uint32_t *hix_large = hid;
		if (tiles_ok) {
			nL = 0;
			if (any_large) {
				hix_large = head_index();
				hipLaunchKernelGGL(k_tie_large_flags, dim3(nblk(u + 1)), dim3(256), 0, s, hix_large, (const uint32_t *)gid, u, flag);
				exclusive_sum_u32(flag, u + 1, scan_ws, s);
				uint32_t h_nl = 0;
				B3M_HIP(hipMemcpyAsync(&h_nl, flag + u, 4, hipMemcpyDeviceToHost, s));
				B3M_HIP(hipStreamSynchronize(s));
				nL = h_nl;
				if (nL == 0) any_large = false;
			}
			if (nL > u / 2) nL = u;                      // mostly large groups: the radix sorts take the whole list
		}
		used_lsd = nL > 0;
		if (nL == u) tie_round_lsd(T, idx, idx_hi, tmp_hi, U, gid, u, kx, ky, vx, vy, hid, nullptr, isa, h, off, gbits, radix, s);
		else {
			// the sub-list of the large groups (nL <= u/2 slots) and its sort buffers share the round's buffers: kx/ky hold u keys each
			const uint64_t hu = (u + 1) / 2;
			uint32_t *U_L = reinterpret_cast<uint32_t *>(kx + hu), *gid_L = U_L + nL, *lpos = reinterpret_cast<uint32_t *>(ky + hu);
			if (nL) hipLaunchKernelGGL(k_tie_large_compact, dim3(nblk(u)), dim3(256), 0, s, flag, hix_large, (const uint32_t *)gid, U, u, U_L, gid_L, lpos);
			if (isa) hipLaunchKernelGGL(k_tie_tile<1>, dim3(ntile), dim3(TT_THREADS), 0, s, T, U, gid, u, idx, (const uint32_t *)isa, h, hid, (const uint64_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint64_t *)nullptr, (const uint32_t *)nullptr, kshift, tt_count_max());
			else hipLaunchKernelGGL(k_tie_tile<0>, dim3(ntile), dim3(TT_THREADS), 0, s, T, U, gid, u, idx, (const uint32_t *)nullptr, off, hid, (const uint64_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint64_t *)nullptr, (const uint32_t *)nullptr, kshift, tt_count_max());
			if (nL) tie_round_lsd(T, idx, nullptr, nullptr, U_L, gid_L, nL, kx, ky, vx, vy, hid, lpos, isa, h, off, gbits, radix, s);
			if (dbg >= 3) fprintf(stderr, "[b3m]   tiles: %u, %llu slots of large groups through the radix sorts\n", ntile, (unsigned long long)nL);
		}
		}
		// the tiles and the pieces of one key wrote every slot's head index; the radix-sort path wrote it at the heads only: one max-scan spreads it
		if (used_lsd) inclusive_max_u32(hid, u, scan_ws, s);
		if (isa) hipLaunchKernelGGL(k_isa_list, dim3(nblk(u)), dim3(256), 0, s, U, hid, idx, u, isa);   // every key of this round has been read
		hipLaunchKernelGGL(k_tie_flags, dim3(nblk(u + 1)), dim3(256), 0, s, hid, u, flag);
		exclusive_sum_u32(flag, u + 1, scan_ws, s);
		B3M_HIP(hipMemcpyAsync(&h_u, flag + u, 4, hipMemcpyDeviceToHost, s));
		// the two value buffers are free now: the compacted list goes there, then the pairs swap
		uint32_t *U2 = vx, *gid2 = vy;
		hipLaunchKernelGGL(k_tie_compact, dim3(nblk(u)), dim3(256), 0, s, flag, hid, U, u, U2, gid2);
		B3M_HIP(hipStreamSynchronize(s));
		if (dbg >= 2) fprintf(stderr, "[b3m]   round %llu: %.2f ms on %llu slots (%.2f ms until the splits were through%s), %u left\n", (unsigned long long)rounds,
		                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_round).count(), (unsigned long long)u, ms_splits,
		                      dbg >= 3 ? "; includes the census" : "", h_u);
#ifdef B3M_PHASE_EXPERIMENT
		if (sw().TT_EXPERIMENT.on()) throw Error("TT_EXPERIMENT: timing-only round done");
#endif
		vx = U; vy = gid; U = U2; gid = gid2;
		heads = true;                                    // (k_tie_compact: the new labels are the heads' new list indices)
		u = h_u;
		h *= 2;
	}
	B[0] = U; B[1] = gid; B[2] = vx; B[3] = vy;
	if (left) *left = u;
	return rounds;
}
// ------------------------------------------------------------------------------------------------
// Prefix doubling ON A LIST ALONE (row a6 for suffix-range shards that ONE plan runs one after the other: texts of 2^32 symbols and more, the host
// tier).  Nobody holds the suffix array of the whole text there, so there is no inverse suffix array to read ranks from.  What there is: the list of
// every suffix that agrees with another one on X symbols or more (each shard hands over what its few key-extension rounds left tied), in rank order,
// i.e. a suffix array of the listed suffixes among themselves (idx, cnt slots).  That is enough: when every group on the list agrees on d >= 2X
// symbols, the suffix h = d - X symbols behind a listed one agrees with its neighbour's on d - h = X symbols, so it is ON THE LIST, and its rank
// there (its slot when it is placed for good, the first slot of its group otherwise) is a key that orders the group.  Members with equal keys agree
// on h + d = 2d - X symbols: (d - X) doubles per round, O(log longest repeat) rounds, every one of them on the list only.  Ranks are looked up by
// position: pos[] ascending, rank[k] = rank of the suffix at pos[k] (binary search; the list is a small fraction of the text).
// ------------------------------------------------------------------------------------------------
struct SparseRanks { const uint64_t *pos; uint64_t *rank; uint64_t cnt; const uint64_t *gslot; uint32_t *missing; };
__device__ __forceinline__ uint64_t sparse_find(const uint64_t *pos, uint64_t cnt, uint64_t x)
{
	uint64_t lo = 0, hi = cnt;
	while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (pos[mid] < x) lo = mid + 1; else hi = mid; }
	return (lo < cnt && pos[lo] == x) ? lo : ~0ull;
}
__device__ __forceinline__ uint64_t sparse_key(const SparseRanks &R, const TextView &T, uint64_t i, uint64_t h)
{
	uint64_t q = i + h;
	if (q >= T.n) { if (T.type == B3M_INPUT_PACTERM) return 0; q -= T.n; }      // (pacterm: a tied suffix has d > h symbols in front of the terminator: not reached)
	const uint64_t k = sparse_find(R.pos, R.cnt, q);
	if (k == ~0ull) { atomicAdd(R.missing, 1u); return 0; }
	return R.rank[k] + 1u;
}
// ranks of the suffixes the record sort placed for good among long repeats (TieExtras): xpos / xrank, nx of them, in any order
__global__ void k_sp_rank_placed(const uint64_t *pos, uint64_t cnt, const uint64_t *xpos, const uint64_t *xrank, uint64_t nx, uint64_t *rank)
{
	uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= nx) return;
	const uint64_t k = sparse_find(pos, cnt, xpos[x]);
	if (k != ~0ull) rank[k] = xrank[x];
}
__global__ void k_sp_init(SparseRanks R, const uint32_t *idx, const uint8_t *idx_hi, uint64_t slots)
{
	uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= slots) return;
	const uint64_t k = sparse_find(R.pos, R.cnt, (uint64_t)idx[e] | (idx_hi ? (uint64_t)idx_hi[e] << 32 : 0ull));
	if (k == ~0ull) atomicAdd(R.missing, 1u); else R.rank[k] = R.gslot[e];
}
__global__ void k_sp_refresh(SparseRanks R, const uint32_t *U, const uint32_t *hid, const uint32_t *idx, const uint8_t *idx_hi, uint64_t u)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	const uint32_t r = U[s];
	const uint64_t k = sparse_find(R.pos, R.cnt, (uint64_t)idx[r] | (idx_hi ? (uint64_t)idx_hi[r] << 32 : 0ull));
	if (k == ~0ull) atomicAdd(R.missing, 1u); else R.rank[k] = R.gslot[U[hid[s]]];
}
__global__ void k_sp_keys(SparseRanks R, TextView T, uint64_t h, const uint32_t *U, const uint32_t *idx, const uint8_t *idx_hi, uint64_t u, uint64_t *nk, uint32_t *slot)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	const uint32_t r = U[s];
	nk[s] = sparse_key(R, T, (uint64_t)idx[r] | (idx_hi ? (uint64_t)idx_hi[r] << 32 : 0ull), h);
	slot[s] = (uint32_t)s;
}
__global__ void k_sp_gather(SparseRanks R, TextView T, uint64_t h, const uint32_t *U, const uint32_t *idx, const uint8_t *idx_hi, const uint32_t *slot_final,
                            uint64_t u, uint32_t *tmp_idx, uint8_t *tmp_hi, uint64_t *tmp_key)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	const uint32_t r = U[slot_final[s]];
	const uint32_t i = idx[r], hi = idx_hi ? idx_hi[r] : 0u;
	tmp_idx[s] = i;
	if (idx_hi) tmp_hi[s] = (uint8_t)hi;
	tmp_key[s] = sparse_key(R, T, (uint64_t)i | ((uint64_t)hi << 32), h);
}
void sparse_rank_placed(const uint64_t *pos, uint64_t cnt, const uint64_t *xpos, const uint64_t *xrank, uint64_t nx, uint64_t *rank, hipStream_t s)
{
	if (nx) hipLaunchKernelGGL(k_sp_rank_placed, dim3(nblk(nx)), dim3(256), 0, s, pos, cnt, xpos, xrank, nx, rank);
}
// idx / idx_hi: the listed suffixes in rank order, `slots` of them (changed in place), gslot[e] = rank of slot e in the whole text; B[0] / B[1]: the u slots still
// tied and their group labels; pos / rank: the table, cnt >= slots entries (pos ascending; the ranks of entries that are not slots of the list are in place,
// sparse_rank_placed; the slots' are filled here); d0: symbols every group agrees on (> X; 2X and more is what makes every round a doubling); X: see above.
// Returns the rounds run; *left = slots still tied (rotations equal over n symbols); *missing = lookups that found no entry (must be 0).
uint64_t tie_rounds_sparse(const TextView &T, uint32_t *idx, uint8_t *idx_hi, uint8_t *tmp_hi, uint64_t slots, uint64_t u, uint64_t *kx, uint64_t *ky, uint32_t *B[6],
                           uint32_t *scan_ws, RadixWs &radix, const uint64_t *pos, uint64_t *rank, const uint64_t *gslot, uint64_t cnt, uint32_t *d_missing,
                           uint64_t d0, uint64_t X, hipStream_t s, uint64_t *left, uint32_t *missing)
{
	B3M_REQUIRE(!idx_hi || tmp_hi, "internal: tie rounds with 36-bit positions need a byte buffer");
	B3M_REQUIRE(d0 > X && X >= 1, "internal: prefix doubling on a list needs groups that agree on more symbols than the listing threshold");
	int kbits = 1; while ((1ull << kbits) < T.n + 2 && kbits < 64) ++kbits;       // keys: rank in the whole text + 1
	int gbits = 1; while ((1ull << gbits) < slots + 1 && gbits < 32) ++gbits;     // labels: slots of the list
	uint32_t *U = B[0], *gid = B[1], *vx = B[2], *vy = B[3], *hid = B[4], *flag = B[5];
	SparseRanks R{pos, rank, cnt, gslot, d_missing};
	B3M_HIP(hipMemsetAsync(d_missing, 0, 4, s));
	const int dbg = (int)sw().DEBUG.get(0);
	uint64_t rounds = 0, d = d0;
	uint32_t h_u = 0;
	if (u) {
		// ranks: its own for everything that is placed for good, the first of its group's for what is still tied
		hipLaunchKernelGGL(k_sp_init, dim3(nblk(slots)), dim3(256), 0, s, R, (const uint32_t *)idx, (const uint8_t *)idx_hi, slots);
		hipLaunchKernelGGL(k_heads_from_gid, dim3(nblk(u)), dim3(256), 0, s, gid, u, hid);
		inclusive_max_u32(hid, u, scan_ws, s);
		hipLaunchKernelGGL(k_sp_refresh, dim3(nblk(u)), dim3(256), 0, s, R, (const uint32_t *)U, (const uint32_t *)hid, (const uint32_t *)idx, (const uint8_t *)idx_hi, u);
	}
	while (u > 0 && d < T.n) {
		const uint64_t h = d - X;
		++rounds;
		hipLaunchKernelGGL(k_sp_keys, dim3(nblk(u)), dim3(256), 0, s, R, T, h, (const uint32_t *)U, (const uint32_t *)idx, (const uint8_t *)idx_hi, u, kx, vx);
		uint64_t *rk; uint32_t *rv;
		radix_sort_u64(kx, vx, ky, vy, u, kbits, radix, s, nullptr, &rk, &rv);
		uint64_t *gk = (rk == kx) ? ky : kx;
		uint32_t *vidle = (rv == vx) ? vy : vx;
		hipLaunchKernelGGL(k_tie_gidkeys, dim3(nblk(u)), dim3(256), 0, s, gid, rv, u, gk);
		uint64_t *rk2; uint32_t *rv2;
		radix_sort_u64(gk, rv, rk, vidle, u, gbits, radix, s, nullptr, &rk2, &rv2);
		uint64_t *tmp_key = (rk2 == gk) ? rk : gk;
		uint32_t *tmp_idx = (rv2 == rv) ? vidle : rv;
		hipLaunchKernelGGL(k_sp_gather, dim3(nblk(u)), dim3(256), 0, s, R, T, h, (const uint32_t *)U, (const uint32_t *)idx, (const uint8_t *)idx_hi, (const uint32_t *)rv2, u, tmp_idx, tmp_hi, tmp_key);
		hipLaunchKernelGGL(k_tie_apply, dim3(nblk(u)), dim3(256), 0, s, U, gid, tmp_idx, (const uint8_t *)tmp_hi, tmp_key, u, idx, idx_hi, hid, (const uint32_t *)nullptr);
		inclusive_max_u32(hid, u, scan_ws, s);
		hipLaunchKernelGGL(k_sp_refresh, dim3(nblk(u)), dim3(256), 0, s, R, (const uint32_t *)U, (const uint32_t *)hid, (const uint32_t *)idx, (const uint8_t *)idx_hi, u);      // every key of this round has been read
		hipLaunchKernelGGL(k_tie_flags, dim3(nblk(u + 1)), dim3(256), 0, s, hid, u, flag);
		exclusive_sum_u32(flag, u + 1, scan_ws, s);
		B3M_HIP(hipMemcpyAsync(&h_u, flag + u, 4, hipMemcpyDeviceToHost, s));
		// the sorts' value buffers are free now: the compacted list goes there, then the pairs swap
		uint32_t *U2 = vx, *gid2 = vy;
		hipLaunchKernelGGL(k_tie_compact, dim3(nblk(u)), dim3(256), 0, s, flag, hid, U, u, U2, gid2);
		B3M_HIP(hipStreamSynchronize(s));
		if (dbg >= 2) fprintf(stderr, "[b3m]   list round %llu: by the rank %llu symbols on (groups agree on %llu), %llu slots, %u left\n", (unsigned long long)rounds, (unsigned long long)h,
		                      (unsigned long long)d, (unsigned long long)u, h_u);
		vx = U; vy = gid; U = U2; gid = gid2;
		u = h_u;
		d = 2 * d - X;
	}
	B[0] = U; B[1] = gid; B[2] = vx; B[3] = vy;
	if (left) *left = u;
	uint32_t miss = 0;
	B3M_HIP(hipMemcpyAsync(&miss, d_missing, 4, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	if (missing) *missing = miss;
	return rounds;
}

// ------------------------------------------------------------------------------------------------
// Ties of a block of a merge tree (the bound on a block's repeat cost the reference
// gets from lcpnext / largelcpthres, src/bwtb3m.cpp:55).  The block [start, start+own) is sorted together
// with an overhang [start+own, start+D) of the text behind it; after a few key-extension rounds the tied
// suffixes are finished by prefix doubling INSIDE that domain: the key of suffix i is the current rank of
// suffix i+h if i+h lies in the domain and that rank is usable.  Where it is not:
//   * an overhang suffix simply leaves the list ("dead": it is only there to lend ranks, it is filtered out
//     of the result) and its rank becomes unknown for whoever refers to it later;
//   * a group with a suffix of the block among them stops doubling as a whole ("blocked": its members' ranks
//     become unknown as well, so nothing that refers to them advances past what is known) and is finished by
//     the key-extension rounds afterwards.  That needs a repeat that reaches from the block to within its own
//     length of the domain's end, i.e. one longer than the overhang.
// isa is indexed by position - start.  Keys: 0 = blocked group, 1 + rank, D + 1 = dead.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t ISA_UNKNOWN = 0xFFFFFFFFu;
__global__ void k_isa_init_rel(const uint32_t *sa, uint64_t len, uint64_t start, uint32_t *isa)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r < len) isa[sa[r] - start] = (uint32_t)r;
}
__global__ void k_isa_list_rel(const uint32_t *U, const uint32_t *hid, const uint32_t *idx, uint64_t u, uint64_t start, uint32_t *isa)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s < u) isa[idx[U[s]] - start] = U[hid[s]];
}
__device__ inline uint32_t dom_ref(const uint32_t *isa, uint64_t i_rel, uint64_t h, uint64_t D)
{
	const uint64_t q = i_rel + h;
	return q < D ? isa[q] : ISA_UNKNOWN;
}
__global__ void k_dom_clear(const uint32_t *gid, uint64_t u, uint32_t *mark)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s < u) mark[gid[s]] = 0;
}
// mark[label] = 1 for the groups that stop doubling: a member of the block whose reference is unknown, or a member whose own rank was given up earlier
__global__ void k_dom_mark(const uint32_t *isa, uint64_t start, uint64_t h, uint64_t D, uint64_t own, const uint32_t *U, const uint32_t *gid, const uint32_t *idx,
                           uint64_t u, uint32_t *mark, uint32_t *any)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	const uint64_t i = (uint64_t)idx[U[s]] - start;
	if (isa[i] == ISA_UNKNOWN || (i < own && dom_ref(isa, i, h, D) == ISA_UNKNOWN)) { mark[gid[s]] = 1; *any = 1; }
}
__device__ inline uint64_t dom_key(const uint32_t *isa, uint64_t i_rel, uint64_t h, uint64_t D, uint32_t blocked)
{
	if (blocked) return 0;
	const uint32_t r = dom_ref(isa, i_rel, h, D);
	return r == ISA_UNKNOWN ? D + 1 : (uint64_t)r + 1;
}
__global__ void k_dom_keys(const uint32_t *isa, uint64_t start, uint64_t h, uint64_t D, const uint32_t *U, const uint32_t *gid, const uint32_t *idx, uint64_t u,
                           const uint32_t *mark, uint64_t *nk, uint32_t *slot)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	nk[s] = dom_key(isa, (uint64_t)idx[U[s]] - start, h, D, mark[gid[s]]);
	slot[s] = (uint32_t)s;
}
// (the sorted group labels equal the old ones slot by slot: gid[s] is the label of whoever ends up in slot s)
__global__ void k_dom_gather(const uint32_t *isa, uint64_t start, uint64_t h, uint64_t D, const uint32_t *U, const uint32_t *gid, const uint32_t *idx,
                             const uint32_t *slot_final, uint64_t u, const uint32_t *mark, uint32_t *tmp_idx, uint64_t *tmp_key)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	const uint32_t i = idx[U[slot_final[s]]];
	tmp_idx[s] = i;
	tmp_key[s] = dom_key(isa, (uint64_t)i - start, h, D, mark[gid[s]]);
}
// ranks of the listed suffixes after the round: first slot of their group; unknown for the dead and the blocked
__global__ void k_dom_isa_list(const uint32_t *U, const uint32_t *hid, const uint32_t *idx, const uint64_t *tmp_key, uint64_t D, uint64_t u, uint64_t start, uint32_t *isa)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	const uint64_t k = tmp_key[s];
	isa[idx[U[s]] - start] = (k == 0 || k == D + 1) ? ISA_UNKNOWN : U[hid[s]];
}
// who stays on the list: members of groups of >= 2 that are not dead
__global__ void k_dom_flags(const uint32_t *hid, const uint64_t *tmp_key, uint64_t D, uint64_t u, uint32_t *flag)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s > u) return;
	if (s == u) { flag[s] = 0; return; }
	bool h0 = hid[s] == (uint32_t)s, h1 = (s + 1 == u) ? true : (hid[s + 1] == (uint32_t)(s + 1));
	flag[s] = ((h0 && h1) || tmp_key[s] == D + 1) ? 0u : 1u;
}
__global__ void k_dom_compact(const uint32_t *flag_scan, const uint32_t *label, const uint32_t *U, uint64_t u, uint32_t *U2, uint32_t *gid2)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= u) return;
	const uint32_t p = flag_scan[s];
	if (flag_scan[s + 1] != p) { U2[p] = U[s]; gid2[p] = label[s]; }
}
// the overhang leaves the list before the key-extension rounds: only the block's own suffixes must end up ordered
__global__ void k_dom_own_flags(const uint32_t *U, const uint32_t *idx, uint64_t start, uint64_t own, uint64_t u, uint32_t *flag)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s > u) return;
	flag[s] = (s < u && (uint64_t)idx[U[s]] - start < own) ? 1u : 0u;
}
// Returns the rounds run; *left = slots still listed (blocked groups, the block's own suffixes only), in B[0]/B[1];
// *h_blocked = symbols every group left on the list agrees on.
uint64_t tie_rounds_domain(uint32_t *idx, uint64_t u, uint64_t *kx, uint64_t *ky, uint32_t *B[6], uint32_t *scan_ws, RadixWs &radix,
                           uint64_t start, uint64_t D, uint64_t own, uint32_t *isa, uint64_t h0, hipStream_t s, uint64_t *left, uint64_t *h_blocked)
{
	int gbits = 1; while ((1ull << gbits) < D + 2 && gbits < 33) ++gbits;
	uint32_t *U = B[0], *gid = B[1], *vx = B[2], *vy = B[3], *hid = B[4], *flag = B[5];
	uint64_t rounds = 0, hb = 0;
	uint32_t h_u = 0, h_any = 0;
	// ranks of the listed suffixes: first slot of their group
	hipLaunchKernelGGL(k_heads_from_gid, dim3(nblk(u)), dim3(256), 0, s, gid, u, hid);
	inclusive_max_u32(hid, u, scan_ws, s);
	hipLaunchKernelGGL(k_isa_list_rel, dim3(nblk(u)), dim3(256), 0, s, U, hid, idx, u, start, isa);
	for (uint64_t h = h0; u > 0 && h < D; h *= 2) {
		++rounds;
		B3M_HIP(hipMemsetAsync(flag, 0, 4, s));
		hipLaunchKernelGGL(k_dom_clear, dim3(nblk(u)), dim3(256), 0, s, gid, u, hid);
		hipLaunchKernelGGL(k_dom_mark, dim3(nblk(u)), dim3(256), 0, s, isa, start, h, D, own, U, gid, idx, u, hid, flag);
		B3M_HIP(hipMemcpyAsync(&h_any, flag, 4, hipMemcpyDeviceToHost, s));
		hipLaunchKernelGGL(k_dom_keys, dim3(nblk(u)), dim3(256), 0, s, isa, start, h, D, U, gid, idx, u, hid, kx, vx);
		uint64_t *rk; uint32_t *rv;
		radix_sort_u64(kx, vx, ky, vy, u, gbits, radix, s, nullptr, &rk, &rv);
		uint64_t *gk = (rk == kx) ? ky : kx;
		uint32_t *vidle = (rv == vx) ? vy : vx;
		hipLaunchKernelGGL(k_tie_gidkeys, dim3(nblk(u)), dim3(256), 0, s, gid, rv, u, gk);
		uint64_t *rk2; uint32_t *rv2;
		radix_sort_u64(gk, rv, rk, vidle, u, 32, radix, s, nullptr, &rk2, &rv2);
		uint64_t *tmp_key = (rk2 == gk) ? rk : gk;
		uint32_t *tmp_idx = (rv2 == rv) ? vidle : rv;
		hipLaunchKernelGGL(k_dom_gather, dim3(nblk(u)), dim3(256), 0, s, isa, start, h, D, U, gid, idx, rv2, u, hid, tmp_idx, tmp_key);
		hipLaunchKernelGGL(k_tie_apply, dim3(nblk(u)), dim3(256), 0, s, U, gid, tmp_idx, (const uint8_t *)nullptr, tmp_key, u, idx, (uint8_t *)nullptr, hid, (const uint32_t *)nullptr);
		inclusive_max_u32(hid, u, scan_ws, s);
		hipLaunchKernelGGL(k_dom_isa_list, dim3(nblk(u)), dim3(256), 0, s, U, hid, idx, tmp_key, D, u, start, isa);
		hipLaunchKernelGGL(k_dom_flags, dim3(nblk(u + 1)), dim3(256), 0, s, hid, tmp_key, D, u, flag);
		exclusive_sum_u32(flag, u + 1, scan_ws, s);
		B3M_HIP(hipMemcpyAsync(&h_u, flag + u, 4, hipMemcpyDeviceToHost, s));
		uint32_t *U2 = vx, *gid2 = vy;
		hipLaunchKernelGGL(k_dom_compact, dim3(nblk(u)), dim3(256), 0, s, flag, hid, U, u, U2, gid2);
		B3M_HIP(hipStreamSynchronize(s));
		vx = U; vy = gid; U = U2; gid = gid2;
		u = h_u;
		if (h_any && !hb) hb = h;
	}
	if (u) {
		// what is left stopped doubling: only the block's own suffixes go on
		hipLaunchKernelGGL(k_dom_own_flags, dim3(nblk(u + 1)), dim3(256), 0, s, U, idx, start, own, u, flag);
		exclusive_sum_u32(flag, u + 1, scan_ws, s);
		B3M_HIP(hipMemcpyAsync(&h_u, flag + u, 4, hipMemcpyDeviceToHost, s));
		hipLaunchKernelGGL(k_dom_compact, dim3(nblk(u)), dim3(256), 0, s, flag, gid, U, u, vx, vy);
		B3M_HIP(hipStreamSynchronize(s));
		std::swap(U, vx); std::swap(gid, vy);
		u = h_u;
	}
	B[0] = U; B[1] = gid; B[2] = vx; B[3] = vy;
	if (left) *left = u;
	if (h_blocked) *h_blocked = hb ? hb : h0;
	return rounds;
}
void isa_from_sa_rel(const uint32_t *d_sa, uint64_t len, uint64_t start, uint32_t *d_isa, hipStream_t s)
{
	if (len) hipLaunchKernelGGL(k_isa_init_rel, dim3(nblk(len)), dim3(256), 0, s, d_sa, len, start, d_isa);
}
void isa_from_sa(const uint32_t *d_sa, uint64_t len, uint32_t *d_isa, hipStream_t s)
{
	if (len) hipLaunchKernelGGL(k_isa_init, dim3(nblk(len)), dim3(256), 0, s, d_sa, len, d_isa);
}
void isa_sparse_from_slots(const uint32_t *d_sa, uint64_t len, const uint32_t *d_slots, uint64_t total, uint32_t *d_isa, hipStream_t s)
{
	B3M_HIP(hipMemsetAsync(d_isa, 0xFF, 4ull * len, s));
	if (total) hipLaunchKernelGGL(k_isa_slots, dim3(nblk(total)), dim3(256), 0, s, d_slots, total, d_sa, d_isa);
}

bool no_tree_doubling() { return sw().NO_TREE_DOUBLING.on(); }
uint64_t block_overhang() { return sw().OVERHANG.set ? (uint64_t)sw().OVERHANG.v : (1ull << 20); }

void block_sort_lsd(const TextView &T, uint64_t start, uint64_t len, BlockSortWs &ws, hipStream_t s, Timeline *tl, BlockSortOut &out, uint64_t own, bool allow_ext)
{
	// K1 keys
	{
		Scope sc(tl, KF_KEYS, len / 4 + 12 * len);
		if (T.type == B3M_INPUT_BYTESTREAM || T.type == B3M_INPUT_SYMBOLS32)
			hipLaunchKernelGGL(k_keys_generic, dim3(nblk(len)), dim3(256), 0, s, T, start, len, ws.keyA, ws.idxA);
		else {
			uint64_t words = ((start + len + 15) >> 4) - (start >> 4);
			hipLaunchKernelGGL(k_keys_packed, dim3(nblk(words)), dim3(256), 0, s, T, start, len, ws.keyA, ws.idxA);
		}
	}
	// K2 radix sort by the first key
	uint64_t *k = nullptr; uint32_t *v = nullptr;
	radix_sort_u64(ws.keyA, ws.idxA, ws.keyB, ws.idxB, len, key_bits(T.type), ws.radix, s, tl, &k, &v);
	uint64_t *k_other = (k == ws.keyA) ? ws.keyB : ws.keyA;
	uint32_t *v_other = (v == ws.idxA) ? ws.idxB : ws.idxA;
	// K3 tie groups of the sorted keys -> list
	uint32_t *hid = ws.t[3], *flag = ws.t[4];
	uint32_t h_u = 0;
	{
		Scope sc(tl, KF_TIES, 24 * len);
		hipLaunchKernelGGL(k_heads_from_keys, dim3(nblk(len)), dim3(256), 0, s, k, len, hid);
		inclusive_max_u32(hid, len, ws.scan_ws, s);
		hipLaunchKernelGGL(k_tie_flags, dim3(nblk(len + 1)), dim3(256), 0, s, hid, len, flag);
		exclusive_sum_u32(flag, len + 1, ws.scan_ws, s);
		B3M_HIP(hipMemcpyAsync(&h_u, flag + len, 4, hipMemcpyDeviceToHost, s));
		hipLaunchKernelGGL(k_tie_compact, dim3(nblk(len)), dim3(256), 0, s, flag, hid, (const uint32_t *)nullptr, len, ws.t[0], ws.t[1]);
		B3M_HIP(hipStreamSynchronize(s));
	}
	out.tie_elements = h_u;
	if (h_u) {
		// K4 rounds; the sorted keys are dead, so both key arrays serve as sort buffers
		Scope sc(tl, KF_TIES, 60ull * h_u);
		uint32_t *B[6] = {ws.t[0], ws.t[1], v_other, ws.t[2], ws.t[3], ws.t[4]};
		uint64_t left = 0;
		// whole text in one block: a few key-extension rounds, then prefix doubling (see tie_rounds)
		const bool whole = start == 0 && len == T.n;
		const bool tree = !whole && !no_tree_doubling();       // a block of a merge tree: tie_rounds_domain
		const uint64_t R0 = 4;
		out.tie_rounds = tie_rounds(T, v, h_u, 1, k, k_other, B, ws.scan_ws, ws.radix, len, s, &left, (whole || tree) ? R0 : 0);
		if (left && tree && out.tie_rounds >= R0) {
			if (allow_ext) { out.want_ext = true; out.d_sa = v; return; }
			const uint64_t K = (uint64_t)key_symbols(T.type);
			uint64_t h0 = 1; while (2 * h0 <= (R0 + 1) * K) h0 *= 2;
			uint64_t hb = 0;
			isa_from_sa_rel(v, len, start, ws.t[5], s);
			out.tie_rounds += tie_rounds_domain(v, left, k, k_other, B, ws.scan_ws, ws.radix, start, len, own ? own : len, ws.t[5], h0, s, &left, &hb);
			if (left) { out.slow_ties = left; out.tie_rounds += tie_rounds(T, v, left, hb / K, k, k_other, B, ws.scan_ws, ws.radix, len, s, &left, 0); }
		}
		if (left && whole && out.tie_rounds >= R0) {
			uint64_t h0 = 1; while (2 * h0 <= (R0 + 1) * (uint64_t)key_symbols(T.type)) h0 *= 2;
			isa_from_sa(v, len, ws.t[5], s);
			out.tie_rounds += tie_rounds(T, v, left, 0, k, k_other, B, ws.scan_ws, ws.radix, len, s, &left, 0, ws.t[5], h0);
		}
	}
	B3M_HIP(hipGetLastError());
	out.d_sa = v;
}

// overhang filter: flag[r] = 1 if sa[r] belongs to the block (position < end); after the scan the block's suffixes are copied in order
__global__ void k_own_flags(const uint32_t *sa, uint64_t len, uint64_t end, uint32_t *flag)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r <= len) flag[r] = (r < len && (uint64_t)sa[r] < end) ? 1u : 0u;
}
__global__ void k_own_compact(const uint32_t *sa, const uint32_t *flag_scan, uint64_t len, uint32_t *out)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= len) return;
	const uint32_t p = flag_scan[r];
	if (flag_scan[r + 1] != p) out[p] = sa[r];
}
void block_sort(const TextView &T, uint64_t start, uint64_t len, BlockSortWs &ws, hipStream_t s, Timeline *tl, BlockSortOut &out,
                uint8_t *d_bwt, uint32_t *d_p0, int shard, int nshards, bool need_sa)
{
	B3M_REQUIRE(nshards > 1 || len <= ws.cap, "block_sort: block larger than workspace");
	B3M_REQUIRE(nshards <= 1 || msd_applicable(T, len), "suffix-range shards need a 2-bit input of at least 2^18 symbols");
	B3M_REQUIRE(T.n < (1ull << 32) || msd_applicable(T, len), "texts of 2^32 symbols and more need a pac/pacterm input (record sort with 36-bit positions)");
	out = BlockSortOut();
	if (len == 0) return;
	// a block of a merge tree may be sorted together with an overhang of the text behind it (tie_rounds_domain): as much as the workspace has room for
	const bool tree = nshards <= 1 && !(start == 0 && len == T.n) && T.n < (1ull << 32);
	const uint64_t H = tree ? std::min(std::min<uint64_t>(ws.cap - len, T.n - (start + len)), block_overhang()) : 0;
	const bool msd = msd_applicable(T, len);
	if (!msd) blocksort_ensure_ties(ws);               // (the key/index path lists its ties in them from the start)
	if (!msd && ws.arrival) for (int i = 0; i < ws.arrival->n; ++i) B3M_HIP(hipStreamWaitEvent(s, ws.arrival->ev[i], 0));      // only the MSD path starts behind a text that is still arriving
	if (msd) block_sort_msd(T, start, len, ws, s, tl, out, d_bwt, d_p0, shard, nshards, need_sa, 0, H > 0);
	else block_sort_lsd(T, start, len, ws, s, tl, out, 0, H > 0);
	if (out.want_ext) {
		blocksort_ensure_ties(ws);
		// long ties: the block and its overhang as one block (suffix array kept, BWT into scratch), then the overhang is filtered out
		const uint64_t first_rounds = out.tie_rounds, first_elems = out.tie_elements;
		uint64_t *hist_keep = ws.d_symhist_out; ws.d_symhist_out = nullptr;      // the histogram of the block was added by the first pass (MSD) or is the caller's business
		const bool hist_done = out.symhist_done;
		BlockSortOut o2;
		const SampleOut *smp_keep = ws.samples; ws.samples = nullptr;            // (ranks inside the extended domain are not the block's: the caller samples the filtered suffix array)
		if (msd) block_sort_msd(T, start, len + H, ws, s, tl, o2, reinterpret_cast<uint8_t *>(ws.t[1]), d_p0, 0, 1, true, len, false);
		else block_sort_lsd(T, start, len + H, ws, s, tl, o2, len, false);
		ws.samples = smp_keep;
		ws.d_symhist_out = hist_keep;
		Scope sc(tl, KF_EMIT, 14 * (len + H));
		uint32_t *flag = ws.t[3], *sa_own = ws.t[2];
		B3M_REQUIRE(o2.d_sa != nullptr && o2.d_sa != sa_own && o2.d_sa != flag, "internal: buffers of the overhang filter");
		hipLaunchKernelGGL(k_own_flags, dim3(nblk(len + H + 1)), dim3(256), 0, s, o2.d_sa, len + H, start + len, flag);
		exclusive_sum_u32(flag, len + H + 1, ws.scan_ws, s);
		hipLaunchKernelGGL(k_own_compact, dim3(nblk(len + H)), dim3(256), 0, s, o2.d_sa, flag, len + H, sa_own);
		B3M_HIP(hipMemsetAsync(d_p0, 0xFF, 4, s));
		emit_bwt(T, sa_own, len, start, d_bwt, d_p0, s);
		out = o2;
		out.d_sa = sa_own; out.d_sa_hi = nullptr; out.kept_len = len; out.slice_offset = 0; out.overhang = H; out.want_ext = false;
		out.tie_rounds += first_rounds; out.tie_elements = first_elems; out.symhist_done = hist_done; out.late_writes = true;
		return;
	}
	if (msd) return;
	out.kept_len = len;
	Scope sc(tl, KF_EMIT, 6 * len);
	emit_bwt(T, out.d_sa, len, start, d_bwt, d_p0, s);
}

void emit_bwt_list(const TextView &T, const uint32_t *U, uint64_t u, const uint32_t *d_sa, uint64_t start, uint8_t *d_bwt, uint32_t *d_p0, hipStream_t s,
                   const SampleOut *smp, const uint8_t *d_sa_hi)
{
	if (u) hipLaunchKernelGGL(k_emit_bwt_list, dim3(nblk(u)), dim3(256), 0, s, T, U, u, d_sa, d_sa_hi, start, d_bwt, d_p0, smp ? *smp : SampleOut());
}

void emit_bwt(const TextView &T, const uint32_t *d_sa, uint64_t len, uint64_t start, uint8_t *d_bwt, uint32_t *d_p0, hipStream_t s)
{
	if (len) hipLaunchKernelGGL(k_emit_bwt, dim3(nblk(len)), dim3(256), 0, s, T, d_sa, len, start, d_bwt, d_p0);
}
void emit_gt(const uint32_t *d_sa, uint64_t len, uint64_t start, uint32_t p0, uint32_t *d_gt_abs, hipStream_t s)
{
	if (len) hipLaunchKernelGGL(k_emit_gt, dim3(nblk(len)), dim3(256), 0, s, d_sa, len, start, p0, d_gt_abs);
}
void emit_samples(const uint32_t *d_sa, const uint8_t *d_sa_hi, uint64_t len, uint64_t rate, uint64_t rank_off, uint64_t *d_pairs, uint64_t first, uint64_t cap, hipStream_t s)
{
	B3M_REQUIRE(rate < (1ull << 32), "sampling rate out of range");
	if (len && rate) hipLaunchKernelGGL(k_emit_samples, dim3(nblk(len)), dim3(256), 0, s, d_sa, d_sa_hi, len, rate, rank_off, d_pairs, first, cap);
}
// in-kernel sampling (SampleOut): any rate below 2^32 for 32-bit positions; texts of 2^32 symbols and more need a power of two
bool sample_in_kernel_ok(uint64_t rate, uint64_t n)
{
	if (rate == 0 || rate >= (1ull << 32)) return false;
	return n < (1ull << 32) || (rate & (rate - 1)) == 0;
}
SampleOut make_sample_out(uint64_t *pairs, uint64_t rate, uint64_t first, uint64_t cap)
{
	B3M_REQUIRE(rate >= 1 && rate < (1ull << 32), "sampling rate out of range");
	SampleOut S;
	S.pairs = pairs; S.first = first; S.cap = cap; S.rate = (uint32_t)rate;
	uint32_t odd = (uint32_t)rate; S.tz = 0;
	while (!(odd & 1u)) { odd >>= 1; ++S.tz; }
	S.pmask = (uint32_t)((1ull << S.tz) - 1);
	S.pow2 = odd == 1u;
	uint32_t inv = odd;                                           // Newton: inv * odd == 1 (mod 2^32)
	for (int i = 0; i < 5; ++i) inv *= 2u - odd * inv;
	S.inv = inv; S.lim = 0xFFFFFFFFu / odd;
	return S;
}
// several lists of queries against the same sorted block in one launch (a leaf's end rank and its anchors for every merge it is on the left of: the searches are chains
// of dependent loads, one launch of all of them costs what the longest list costs)
__global__ void k_rank_of_multi(TextView T, const uint32_t *sa, uint64_t len, RankQueries Q)
{
	uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int g = 0;
	while (g < Q.n && q >= Q.count[g]) { q -= Q.count[g]; ++g; }
	if (g >= Q.n) return;
	const uint64_t x = Q.pos[g][q] % T.n;
	const uint64_t kx = key_at(T, x, 0);          // the query's first key once, not at every step of the search
	uint64_t lo = 0, hi = len;
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		const uint64_t y = sa[mid];
		int c = 0;
		if (y != x) { const uint64_t ky = key_at(T, y, 0); c = ky != kx ? (ky < kx ? -1 : 1) : compare_rotations(T, y, x); }
		if (c < 0) lo = mid + 1; else hi = mid;
	}
	atomicAdd(&Q.accum[g][q], (uint32_t)lo);
}
void rank_of_positions_multi(const TextView &T, const uint32_t *d_sa, uint64_t len, const RankQueries &Q, hipStream_t s)
{
	uint64_t total = 0;
	for (int g = 0; g < Q.n; ++g) total += Q.count[g];
	if (total) hipLaunchKernelGGL(k_rank_of_multi, dim3(nblk(total, 64)), dim3(64), 0, s, T, d_sa, len, Q);
}
void rank_of_positions(const TextView &T, const uint32_t *d_sa, uint64_t len, const uint64_t *d_pos, uint64_t count,
                       uint32_t *d_rank_accum, hipStream_t s)
{
	if (count) hipLaunchKernelGGL(k_rank_of, dim3(nblk(count, 64)), dim3(64), 0, s, T, d_sa, len, d_pos, count, d_rank_accum);
}
// 2-bit inputs: 16 bases per 32-bit word counted with popcounts, per-thread counters, one
// atomic per wave and symbol at the end (no LDS-atomic contention on 4 hot counters)
__global__ __launch_bounds__(256) void k_histogram_packed(TextView T, uint64_t start, uint64_t len, unsigned long long *hist)
{
	const uint64_t e = start + len;
	uint64_t w0 = (start + 15) >> 4, w1 = e >> 4;                  // whole words inside the range ...
	uint64_t wl = T.nbases >> 4; if (wl > w1) wl = w1;             // ... and inside the bases
	if (wl < w0) wl = w0;
	uint32_t c1 = 0, c2 = 0, c3 = 0; uint64_t nb = 0;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	auto count = [&](uint32_t x) {                                  // byte order is irrelevant for counting
		uint32_t lo = x & 0x55555555u, hi = (x >> 1) & 0x55555555u;
		c1 += __popc(lo & ~hi); c2 += __popc(hi & ~lo); c3 += __popc(lo & hi); nb += 16;
	};
	// bulk: 16-byte loads (64 bases), two in flight per thread; the words before / after the 16-byte aligned part singly
	const uint64_t q0 = (w0 + 3) >> 2, q1 = wl >> 2;               // whole groups of 4 words
	if (q1 > q0) {
		const uint4 *t4 = reinterpret_cast<const uint4 *>(T.b);
		const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
		for (uint64_t q = q0 + gid; q < q1; q += 2 * stride) {
			uint4 a = t4[q];
			uint4 b = q + stride < q1 ? t4[q + stride] : make_uint4(0, 0, 0, 0);
			count(a.x); count(a.y); count(a.z); count(a.w);
			if (q + stride < q1) { count(b.x); count(b.y); count(b.z); count(b.w); }
		}
		if (gid < (q0 << 2) - w0) count(*reinterpret_cast<const uint32_t *>(T.b + ((w0 + gid) << 2)));
		if (gid < wl - (q1 << 2)) count(*reinterpret_cast<const uint32_t *>(T.b + (((q1 << 2) + gid) << 2)));
	} else {
		for (uint64_t w = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < wl; w += stride) count(*reinterpret_cast<const uint32_t *>(T.b + (w << 2)));
	}
	uint64_t c0 = nb - c1 - c2 - c3;
	uint64_t d1 = c1, d2 = c2, d3 = c3;
	for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_down(c0, o, 64); d1 += __shfl_down(d1, o, 64); d2 += __shfl_down(d2, o, 64); d3 += __shfl_down(d3, o, 64); }
	int off = T.type == B3M_INPUT_PACTERM ? 1 : 0;
	if ((threadIdx.x & 63) == 0) {
		if (c0) atomicAdd(&hist[off + 0], (unsigned long long)c0);
		if (d1) atomicAdd(&hist[off + 1], (unsigned long long)d1);
		if (d2) atomicAdd(&hist[off + 2], (unsigned long long)d2);
		if (d3) atomicAdd(&hist[off + 3], (unsigned long long)d3);
	}
	// ragged ends, one symbol at a time: [start, head_end) and [tail_beg, e)
	if (blockIdx.x == 0) {
		uint64_t head_end = (w0 << 4) < e ? (w0 << 4) : e;
		if (wl <= w0) head_end = e;                                // no bulk word at all
		uint64_t tail_beg = (wl << 4) > head_end ? (wl << 4) : head_end;
		for (uint64_t i = start + threadIdx.x; i < head_end; i += blockDim.x) atomicAdd(&hist[sym_at(T, i)], 1ull);
		for (uint64_t i = tail_beg + threadIdx.x; i < e; i += blockDim.x) atomicAdd(&hist[sym_at(T, i)], 1ull);
	}
}

void histogram_symbols(const TextView &T, uint64_t start, uint64_t len, uint64_t *d_hist256, hipStream_t s)
{
	if (!len) return;
	B3M_REQUIRE(T.type != B3M_INPUT_SYMBOLS32, "internal: no 256-bin histogram of a symbols32 text");
	if (T.type != B3M_INPUT_BYTESTREAM && len >= 4096) {
		uint32_t grid = nblk(len / 16, 256 * 8); if (grid > 2048) grid = 2048; if (grid == 0) grid = 1;
		hipLaunchKernelGGL(k_histogram_packed, dim3(grid), dim3(256), 0, s, T, start, len, (unsigned long long *)d_hist256);
		return;
	}
	uint32_t grid = nblk(len, 256 * 64); if (grid > 4096) grid = 4096; if (grid == 0) grid = 1;
	hipLaunchKernelGGL(k_histogram, dim3(grid), dim3(256), 0, s, T, start, len, (unsigned long long *)d_hist256);
}

}  // namespace b3m
