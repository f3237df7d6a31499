// k_scan_radix.hip -- device-wide scans and the LDS-staged LSD radix sort.
//
// Radix pass = three launches (reduce-then-scan form):
//   k_tile_hist    : per-tile digit histogram in LDS  -> tile_hist[digit][tile]
//   exclusive scan : over the flat [256][numTiles] matrix = global offset of (digit, tile)
//   k_scatter      : each wave ranks its 64 keys per round with ballot match-any
//                    (8 ballots -> mask of lanes sharing my digit, popcount below my lane =
//                    rank inside the wave), per-wave running digit counts live in LDS, the
//                    tile is reordered through LDS so every digit's run leaves as one
//                    contiguous segment, then written to HBM.
// 64-wide wavefronts throughout (no 32-lane idioms).  Stable, so LSD over digits is correct.
#include "b3m_internal.h"
#include "b3m_text.h"

namespace b3m {

// ================================================================================================
// scans
// ================================================================================================
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

struct OpSum { __device__ static uint32_t id() { return 0; } __device__ static uint32_t f(uint32_t a, uint32_t b) { return a + b; } };
struct OpMax { __device__ static uint32_t id() { return 0; } __device__ static uint32_t f(uint32_t a, uint32_t b) { return a > b ? a : b; } };

template <class Op> __device__ __forceinline__ uint32_t wave_incl(uint32_t v)
{
	#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { uint32_t u = __shfl_up(v, o, 64); if ((int)lane_id() >= o) v = Op::f(v, u); }
	return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix,
// *total = block aggregate
template <class Op> __device__ __forceinline__ uint32_t block_excl(uint32_t v, uint32_t *total, uint32_t *lds /*>=5*/)
{
	uint32_t inc = wave_incl<Op>(v);
	uint32_t w = threadIdx.x >> 6, l = lane_id();
	if (l == 63) lds[w] = inc;
	__syncthreads();
	uint32_t pre = Op::id(), tot = Op::id();
	#pragma unroll
	for (uint32_t i = 0; i < SCAN_THREADS / 64; ++i) { uint32_t x = lds[i]; if (i < w) pre = Op::f(pre, x); tot = Op::f(tot, x); }
	uint32_t up = __shfl_up(inc, 1, 64);
	uint32_t ex = Op::f(pre, l ? up : Op::id());
	*total = tot;
	__syncthreads();
	return ex;
}

// EXCL: exclusive (sum) ; otherwise inclusive (max)
template <class Op, bool EXCL>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tiles(uint32_t *d, uint64_t n, uint32_t *sums)
{
	__shared__ uint32_t lds[8];
	uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
	uint32_t v[SCAN_ITEMS];
	uint32_t acc = Op::id();
	#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = (base + i < n) ? d[base + i] : Op::id(); acc = Op::f(acc, v[i]); }
	uint32_t tot;
	uint32_t ex = block_excl<Op>(acc, &tot, lds);
	uint32_t run = ex;
	#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; ++i) {
		uint32_t nv = Op::f(run, v[i]);
		if (base + i < n) d[base + i] = EXCL ? run : nv;
		run = nv;
	}
	if (threadIdx.x == 0 && sums) sums[blockIdx.x] = tot;
}

template <class Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(uint32_t *d, uint64_t n, const uint32_t *sums /*exclusive-scanned*/)
{
	uint32_t add = sums[blockIdx.x];
	uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
	#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) d[base + i] = Op::f(d[base + i], add);
}

uint64_t scan_ws_elems(uint64_t n)
{
	uint64_t tot = 0;
	while (n > SCAN_TILE) { n = (n + SCAN_TILE - 1) / SCAN_TILE; tot += n + 8; }
	return tot + 16;
}

template <class Op, bool EXCL> static void scan_rec(uint32_t *d, uint64_t n, uint32_t *ws, hipStream_t s)
{
	if (n == 0) return;
	uint64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
	if (tiles == 1) {
		hipLaunchKernelGGL((k_scan_tiles<Op, EXCL>), dim3(1), dim3(SCAN_THREADS), 0, s, d, n, (uint32_t *)nullptr);
		return;
	}
	hipLaunchKernelGGL((k_scan_tiles<Op, EXCL>), dim3((uint32_t)tiles), dim3(SCAN_THREADS), 0, s, d, n, ws);
	// tile aggregates -> exclusive prefix of aggregates (always exclusive, also for the inclusive max)
	scan_rec<Op, true>(ws, tiles, ws + tiles + 8, s);
	hipLaunchKernelGGL((k_scan_apply<Op>), dim3((uint32_t)tiles), dim3(SCAN_THREADS), 0, s, d, n, ws);
}

void exclusive_sum_u32(uint32_t *d, uint64_t n, uint32_t *ws, hipStream_t s) { scan_rec<OpSum, true>(d, n, ws, s); }
void inclusive_max_u32(uint32_t *d, uint64_t n, uint32_t *ws, hipStream_t s) { scan_rec<OpMax, false>(d, n, ws, s); }

// ================================================================================================
// radix sort
// ================================================================================================
constexpr int RX_THREADS = 256;
constexpr int RX_ITEMS = 16;                       // keys per thread
constexpr int RX_TILE = RX_THREADS * RX_ITEMS;     // 4096 keys per workgroup
constexpr int RX_BITS = 8;
constexpr int RX_BINS = 1 << RX_BITS;

uint64_t radix_tiles(uint64_t n) { return (n + RX_TILE - 1) / RX_TILE; }
uint64_t radix_ws_bytes(uint64_t n)
{
	uint64_t tiles = radix_tiles(n) + 1;
	uint64_t e = RX_BINS * tiles + scan_ws_elems(RX_BINS * tiles) + 8 * RX_BINS + 64;
	return e * 4;
}
void radix_ws_carve(RadixWs &ws, void *base, uint64_t n)
{
	uint64_t tiles = radix_tiles(n) + 1;
	uint32_t *p = (uint32_t *)base;
	ws.tile_hist = p; p += RX_BINS * tiles;
	ws.scan_ws = p; p += scan_ws_elems(RX_BINS * tiles);
	ws.ghist = p;
	ws.cap_elems = n;
}

// all digit histograms of the key in one read: ghist[pass][256] (used only to skip constant digits)
template <typename KeyT>
__global__ __launch_bounds__(RX_THREADS) void k_global_hist(const KeyT *keys, uint64_t n, int npasses, uint32_t *ghist)
{
	__shared__ uint32_t h[8][RX_BINS];
	for (int i = threadIdx.x; i < 8 * RX_BINS; i += RX_THREADS) ((uint32_t *)h)[i] = 0;
	__syncthreads();
	uint64_t stride = (uint64_t)gridDim.x * RX_THREADS;
	for (uint64_t i = (uint64_t)blockIdx.x * RX_THREADS + threadIdx.x; i < n; i += stride) {
		KeyT k = keys[i];
		for (int p = 0; p < npasses; ++p) atomicAdd(&h[p][(uint32_t)(k >> (p * RX_BITS)) & (RX_BINS - 1)], 1u);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < npasses * RX_BINS; i += RX_THREADS) {
		uint32_t v = ((uint32_t *)h)[i];
		if (v) atomicAdd(&ghist[i], v);
	}
}

// THREADS x 16 keys per workgroup.  The length of a digit's run leaving the workgroup is what
// bounds the scatter (k_msd.hip, NOTES_r01.md), so the tile is as large as the LDS allows.
template <typename KeyT, int THREADS>
__global__ __launch_bounds__(THREADS) void k_tile_hist(const KeyT *keys, uint64_t n, int shift, uint32_t *tile_hist, uint32_t num_tiles)
{
	__shared__ uint32_t h[RX_BINS];
	if (threadIdx.x < RX_BINS) h[threadIdx.x] = 0;
	__syncthreads();
	uint64_t base = (uint64_t)blockIdx.x * (THREADS * RX_ITEMS);
	#pragma unroll
	for (int t = 0; t < RX_ITEMS; ++t) {
		uint64_t i = base + (uint64_t)t * THREADS + threadIdx.x;
		if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & (RX_BINS - 1)], 1u);
	}
	__syncthreads();
	if (threadIdx.x < RX_BINS) tile_hist[(uint64_t)threadIdx.x * num_tiles + blockIdx.x] = h[threadIdx.x];
}

template <typename KeyT, bool HAS_VAL, int THREADS>
__global__ __launch_bounds__(THREADS) void k_scatter(const KeyT *kin, const uint32_t *vin, KeyT *kout, uint32_t *vout,
                                                       uint64_t n, int shift, const uint32_t *tile_off, uint32_t num_tiles)
{
	constexpr int WAVES = THREADS / 64, TILE = THREADS * RX_ITEMS;
	// per-wave running digit counts, later turned into LDS positions of each (wave,digit) run
	__shared__ uint32_t whist[WAVES][RX_BINS];
	__shared__ uint32_t gbase[RX_BINS];          // global position of each digit's run minus its start inside the tile
	__shared__ KeyT skey[TILE];
	__shared__ uint32_t sval[HAS_VAL ? TILE : 1];
	__shared__ uint32_t scan_lds[8];

	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	for (int i = tid; i < WAVES * RX_BINS; i += THREADS) ((uint32_t *)whist)[i] = 0;
	__syncthreads();

	const uint64_t tile_base = (uint64_t)blockIdx.x * TILE;
	const uint64_t wbase = tile_base + (uint64_t)wave * (64 * RX_ITEMS);
	const uint32_t tile_n = (uint32_t)((n - tile_base) < (uint64_t)TILE ? (n - tile_base) : (uint64_t)TILE);

	KeyT key[RX_ITEMS];
	uint32_t val[RX_ITEMS];
	uint32_t rnk[RX_ITEMS];
	volatile uint32_t *wh = whist[wave];
	#pragma unroll
	for (int t = 0; t < RX_ITEMS; ++t) {
		uint64_t i = wbase + (uint64_t)t * 64 + lane;
		bool valid = i < n;
		key[t] = valid ? kin[i] : (KeyT)0;
		if (HAS_VAL) val[t] = valid ? vin[i] : 0u;
		uint32_t d = (uint32_t)(key[t] >> shift) & (RX_BINS - 1);
		// match-any over the 8 digit bits: mask of valid lanes holding the same digit
		uint64_t m = __ballot(valid);
		#pragma unroll
		for (int b = 0; b < RX_BITS; ++b) {
			uint64_t bal = __ballot((d >> b) & 1u);
			m &= ((d >> b) & 1u) ? bal : ~bal;
		}
		uint32_t below = __popcll(m & ((1ull << lane) - 1ull));
		uint32_t cnt = __popcll(m);
		uint32_t old = 0;
		if (valid && below == 0) { old = wh[d]; wh[d] = old + cnt; }   // leader = lowest lane of the group
		int leader = m ? (__ffsll((long long)m) - 1) : 0;
		old = __shfl(old, leader, 64);
		rnk[t] = old + below;
	}
	__syncthreads();
	// digit totals over the waves -> exclusive scan over the 256 digits (threads 0..255) -> LDS run starts
	{
		uint32_t tot = 0;
		if (tid < RX_BINS) { for (int w = 0; w < WAVES; ++w) tot += whist[w][tid]; }
		uint32_t inc = wave_incl<OpSum>(tot);
		if (tid < RX_BINS && lane == 63) scan_lds[wave] = inc;
		__syncthreads();
		if (tid < RX_BINS) {
			uint32_t pre = 0;
			for (uint32_t i = 0; i < wave; ++i) pre += scan_lds[i];
			uint32_t ex = pre + inc - tot;
			gbase[tid] = tile_off[(uint64_t)tid * num_tiles + blockIdx.x] - ex;
			uint32_t run = ex;
			for (int w = 0; w < WAVES; ++w) { uint32_t c = whist[w][tid]; whist[w][tid] = run; run += c; }
		}
	}
	__syncthreads();
	#pragma unroll
	for (int t = 0; t < RX_ITEMS; ++t) {
		uint64_t i = wbase + (uint64_t)t * 64 + lane;
		if (i < n) {
			uint32_t d = (uint32_t)(key[t] >> shift) & (RX_BINS - 1);
			uint32_t p = whist[wave][d] + rnk[t];
			skey[p] = key[t];
			if (HAS_VAL) sval[p] = val[t];
		}
	}
	__syncthreads();
	// coalesced-by-run write: consecutive threads write consecutive slots of the reordered tile
	#pragma unroll
	for (int t = 0; t < RX_ITEMS; ++t) {
		uint32_t p = (uint32_t)t * THREADS + tid;
		if (p < tile_n) {
			KeyT k = skey[p];
			uint32_t d = (uint32_t)(k >> shift) & (RX_BINS - 1);
			uint32_t g = gbase[d] + p;
			kout[g] = k;
			if (HAS_VAL) vout[g] = sval[p];
		}
	}
}

template <typename KeyT>
static void radix_sort_impl(KeyT *kA, uint32_t *vA, KeyT *kB, uint32_t *vB, uint64_t n, int nbits, RadixWs &ws,
                            hipStream_t s, Timeline *tl, KeyT **k_res, uint32_t **v_res)
{
	*k_res = kA; if (v_res) *v_res = vA;
	if (n <= 1 || nbits <= 0) return;
	B3M_REQUIRE(n <= ws.cap_elems, "radix workspace too small");
	B3M_REQUIRE(n < (1ull << 32), "radix sort: n >= 2^32 not supported");
	int npasses = (nbits + RX_BITS - 1) / RX_BITS;
	// 8 Ki keys per workgroup: 32-key runs (4-byte keys at 16 Ki per workgroup measured slower: one workgroup per CU)
	constexpr int TH = 512;
	uint32_t tiles = (uint32_t)((n + (uint64_t)TH * RX_ITEMS - 1) / ((uint64_t)TH * RX_ITEMS));
	// which digits are constant over all keys?  (one read of the keys, one small D2H)
	uint32_t hg[8 * RX_BINS];
	{
		Scope sc(tl, KF_RADIX_HIST, n * sizeof(KeyT));
		B3M_HIP(hipMemsetAsync(ws.ghist, 0, sizeof(hg), s));
		uint32_t grid = (uint32_t)((n + RX_THREADS * 16 - 1) / (RX_THREADS * 16));
		if (grid > 2048) grid = 2048;
		hipLaunchKernelGGL((k_global_hist<KeyT>), dim3(grid), dim3(RX_THREADS), 0, s, kA, n, npasses, ws.ghist);
	}
	B3M_HIP(hipMemcpyAsync(hg, ws.ghist, sizeof(uint32_t) * npasses * RX_BINS, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	KeyT *kin = kA, *kout = kB; uint32_t *vin = vA, *vout = vB;
	for (int p = 0; p < npasses; ++p) {
		bool constant = false;
		for (int b = 0; b < RX_BINS; ++b) if (hg[p * RX_BINS + b] == (uint32_t)n) { constant = true; break; }
		if (constant) continue;
		int shift = p * RX_BITS;
		{
			Scope sc(tl, KF_RADIX_HIST, n * sizeof(KeyT));
			hipLaunchKernelGGL((k_tile_hist<KeyT, TH>), dim3(tiles), dim3(TH), 0, s, kin, n, shift, ws.tile_hist, tiles);
		}
		{
			Scope sc(tl, KF_SCAN, (uint64_t)RX_BINS * tiles * 8);
			exclusive_sum_u32(ws.tile_hist, (uint64_t)RX_BINS * tiles, ws.scan_ws, s);
		}
		{
			uint64_t rec = sizeof(KeyT) + (vA ? 4 : 0);
			Scope sc(tl, KF_RADIX_SCATTER, 2 * n * rec);
			if (vA) hipLaunchKernelGGL((k_scatter<KeyT, true, TH>), dim3(tiles), dim3(TH), 0, s, kin, vin, kout, vout, n, shift, ws.tile_hist, tiles);
			else hipLaunchKernelGGL((k_scatter<KeyT, false, TH>), dim3(tiles), dim3(TH), 0, s, kin, (const uint32_t *)nullptr, kout, (uint32_t *)nullptr, n, shift, ws.tile_hist, tiles);
		}
		KeyT *tk = kin; kin = kout; kout = tk;
		uint32_t *tv = vin; vin = vout; vout = tv;
	}
	B3M_HIP(hipGetLastError());
	*k_res = kin; if (v_res) *v_res = vin;
}

void radix_sort_u64(uint64_t *kA, uint32_t *vA, uint64_t *kB, uint32_t *vB, uint64_t n, int nbits, RadixWs &ws,
                    hipStream_t s, Timeline *tl, uint64_t **k_res, uint32_t **v_res)
{
	radix_sort_impl<uint64_t>(kA, vA, kB, vB, n, nbits, ws, s, tl, k_res, v_res);
}
void radix_sort_u32(uint32_t *kA, uint32_t *kB, uint64_t n, int nbits, RadixWs &ws, hipStream_t s, Timeline *tl, uint32_t **k_res, bool rank_list)
{
	// large rank lists: unstable MSD partitions + LDS histograms (k_msd.hip); its scratch is carved from the tile histogram area
	if (rank_list && n <= ws.cap_elems) {
		const uint64_t have = (uint64_t)RX_BINS * (radix_tiles(ws.cap_elems) + 1);
		if (sort_u32_msd(kA, kB, n, nbits, ws.tile_hist, have, ws.scan_ws, s, tl, k_res)) return;
	}
	radix_sort_impl<uint32_t>(kA, nullptr, kB, nullptr, n, nbits, ws, s, tl, k_res, nullptr);
}

}  // namespace b3m
