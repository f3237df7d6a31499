// k_merge.hip -- gap/rank merge of two adjacent blocks (the gap
// computation and interleave halves of libmaus2's mergeBlocks, and the sparse gap array).
//
//   rank dictionary : BWT(L) as 64-byte lines {4 x u32 symbol counts | 192 two-bit codes}; one
//                     rank query = one line.
//   LF walk         : one lane = one chain of consecutive positions of R walked backwards from
//                     an anchor (SURVEY.md Appendix A recurrence); every step emits the rank of
//                     one suffix of R among the suffixes of L.
//   sparse gap      : the emitted ranks are radix-sorted; sorted_ranks[q] is the rank in L of
//                     the q-th smallest suffix of R (run lengths of equal values = G).
//   interleave      : output tile by output tile; positions q + sorted_ranks[q] take BWT(R)[q],
//                     the holes take BWT(L) in order.
// Integer/bit work bound by HBM latency and bandwidth; no MFMA.
#include "b3m_internal.h"
#include "b3m_text.h"
#include <cstdlib>

namespace b3m {

static inline uint32_t nblk(uint64_t n, uint32_t t = 256) { return (uint32_t)((n + t - 1) / t); }

uint64_t rank_blocks(uint64_t m) { return m / RANK_BLK_SYMS + 1; }

// ------------------------------------------------------------------------------------------------
// K6 rank dictionary
// ------------------------------------------------------------------------------------------------
// pass 1: one thread per payload word = 16 consecutive symbols: pack the 2-bit codes, count them
// with popcounts (the p0 entry is stored as code 0 and not counted).  wc[w] = the four counts of
// the word, one byte each.
template <typename P0T>
__global__ __launch_bounds__(256) void k_rank_pack_words(const uint8_t *bwt, uint64_t m, const P0T *p0ptr, int sub, uint32_t *blocks,
                                                           uint32_t *wc, uint64_t nwords)
{
	uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords) return;
	const uint64_t p0 = sizeof(P0T) == 4 && *p0ptr == (P0T)0xFFFFFFFFu ? ~0ull : (uint64_t)*p0ptr;
	uint64_t base = w * 16;
	uint32_t word = 0, valid = 0;           // valid: bit 2k set if symbol k counts
	if (base + 16 <= m) {
		uint8_t v[16];
		__builtin_memcpy(v, bwt + base, 16);
		#pragma unroll
		for (int k = 0; k < 16; ++k) word |= (((uint32_t)v[k] - (uint32_t)sub) & 3u) << (2 * k);
		valid = 0x55555555u;
	} else {
		for (int k = 0; k < 16; ++k) if (base + k < m) { word |= (((uint32_t)bwt[base + k] - (uint32_t)sub) & 3u) << (2 * k); valid |= 1u << (2 * k); }
	}
	if (p0 >= base && p0 < base + 16) { uint32_t k = (uint32_t)(p0 - base); word &= ~(3u << (2 * k)); valid &= ~(1u << (2 * k)); }
	uint32_t lo = word & 0x55555555u, hi = (word >> 1) & 0x55555555u;
	uint32_t c1 = __popc(lo & ~hi & valid), c2 = __popc(hi & ~lo & valid), c3 = __popc(lo & hi & valid);
	uint32_t c0 = __popc(valid) - c1 - c2 - c3;
	uint64_t b = w / 12, j = w - b * 12;
	blocks[b * 16 + 4 + j] = word;
	wc[w] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
}
// pass 2: per 64-byte line: sum of its 12 word counts -> cnt[4][nb]
__global__ void k_rank_line_counts(const uint32_t *wc, uint64_t nb, uint32_t *cnt)
{
	uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= nb) return;
	uint32_t s = 0, t = 0;                  // byte-wise sums: 12 * 16 = 192 < 256 per lane of the packed word
	#pragma unroll
	for (int j = 0; j < 12; ++j) { uint32_t x = wc[b * 12 + j]; s += x & 0x00FF00FFu; t += (x >> 8) & 0x00FF00FFu; }
	cnt[0 * nb + b] = s & 0xFFFFu; cnt[2 * nb + b] = s >> 16;
	cnt[1 * nb + b] = t & 0xFFFFu; cnt[3 * nb + b] = t >> 16;
}
// pass 3 (after 4 exclusive scans): headers
__global__ void k_rank_headers(uint32_t *blocks, const uint32_t *cnt, uint64_t nb)
{
	uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= nb) return;
	uint32_t *out = blocks + b * 16;
	out[0] = cnt[0 * nb + b]; out[1] = cnt[1 * nb + b]; out[2] = cnt[2 * nb + b]; out[3] = cnt[3 * nb + b];
}

// ---- 2-bit transport form of a block BWT (multi-GPU merges: BWT(L) travels packed, 4 symbols per byte) ----------------
// code = symbol - sub (pacterm: base + 1 -> base); symbol k of the slice in byte k>>2 at bits 2*(k&3).  The slice's p0 entry
// (and the pacterm terminator, which is the p0 entry of every node that holds block 0) packs as garbage: its true byte
// travels beside the slice and the dictionary never counts it.
__global__ __launch_bounds__(256) void k_pack2(const uint8_t *bwt, uint64_t m, int sub, uint8_t *packed)
{
	uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (q * 4 >= m) return;
	uint32_t v = 0;
	for (int k = 0; k < 4; ++k) if (q * 4 + k < m) v |= (((uint32_t)bwt[q * 4 + k] - (uint32_t)sub) & 3u) << (2 * k);
	packed[q] = (uint8_t)v;
}
void pack2(const uint8_t *d_bwt, uint64_t m, int type, uint8_t *d_packed, hipStream_t s)
{
	if (m) hipLaunchKernelGGL(k_pack2, dim3(nblk((m + 3) / 4)), dim3(256), 0, s, d_bwt, m, type == B3M_INPUT_PACTERM ? 1 : 0, d_packed);
}
// rank dictionary payload + word counts from up to DICT_MAX_PARTS packed slices laid end to end (symbol g of L lies in the part
// whose [off, off + syms) holds g)
struct PackedParts { const uint8_t *p[DICT_MAX_PARTS]; uint64_t off[DICT_MAX_PARTS + 1]; int n; };
__global__ __launch_bounds__(256) void k_rank_words_from_parts(PackedParts P, uint64_t m, uint64_t p0, uint32_t *blocks, uint32_t *wc, uint64_t nwords)
{
	uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords) return;
	const uint64_t base = w * 16;
	uint32_t word = 0, valid = 0;
	int j = 0;
	while (j + 1 < P.n && P.off[j + 1] <= base) ++j;
	for (int k = 0; k < 16; ++k) {
		const uint64_t g = base + k;
		if (g >= m) break;
		while (j + 1 < P.n && P.off[j + 1] <= g) ++j;
		const uint64_t l = g - P.off[j];
		word |= (((uint32_t)P.p[j][l >> 2] >> (2 * (l & 3))) & 3u) << (2 * k);
		valid |= 1u << (2 * k);
	}
	if (p0 >= base && p0 < base + 16) { uint32_t k = (uint32_t)(p0 - base); word &= ~(3u << (2 * k)); valid &= ~(1u << (2 * k)); }
	uint32_t lo = word & 0x55555555u, hi = (word >> 1) & 0x55555555u;
	uint32_t c1 = __popc(lo & ~hi & valid), c2 = __popc(hi & ~lo & valid), c3 = __popc(lo & hi & valid);
	uint32_t c0 = __popc(valid) - c1 - c2 - c3;
	uint64_t b = w / 12, jj = w - b * 12;
	blocks[b * 16 + 4 + jj] = word;
	wc[w] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
}
// bytes of BWT(L)[l0, l1) back from the dictionary payload (the p0 entry gets its true byte)
__global__ __launch_bounds__(256) void k_dict_bytes(TextView T, const uint32_t *blocks, uint64_t l0, uint64_t l1, uint64_t p0, uint64_t p0_pos, int add, uint8_t *out)
{
	uint64_t g = l0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= l1) return;
	const uint64_t b = g / RANK_BLK_SYMS; const uint32_t off = (uint32_t)(g - b * RANK_BLK_SYMS);
	const uint32_t word = blocks[b * 16 + 4 + (off >> 4)];
	uint32_t v = ((word >> (2 * (off & 15))) & 3u) + (uint32_t)add;
	if (g == p0) v = sym_at(T, p0_pos);                 // the entry of the node's first suffix: the symbol in front of the node, from the text
	out[g - l0] = (uint8_t)v;
}
void dict_bytes(const TextView &T, const uint32_t *d_blocks, uint64_t l0, uint64_t l1, uint64_t p0, uint64_t p0_pos, uint8_t *d_out, hipStream_t s)
{
	if (l1 > l0) hipLaunchKernelGGL(k_dict_bytes, dim3(nblk(l1 - l0)), dim3(256), 0, s, T, d_blocks, l0, l1, p0, p0_pos, T.type == B3M_INPUT_PACTERM ? 1 : 0, d_out);
}
// bytes [q0, q1) of a block BWT that arrived as packed slices laid end to end (multi-GPU merges: an owner's range of BWT(R) out of the ONE all-gather
// that carried both sides' slices; R never holds the terminator and its p0 entry is a real symbol of the text, so the codes are the whole truth)
__global__ __launch_bounds__(256) void k_parts_bytes(PackedParts P, uint64_t q0, uint64_t q1, int add, uint8_t *out)
{
	const uint64_t g = q0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= q1) return;
	int j = 0;
	while (j + 1 < P.n && P.off[j + 1] <= g) ++j;
	const uint64_t l = g - P.off[j];
	out[g - q0] = (uint8_t)((((uint32_t)P.p[j][l >> 2] >> (2 * (l & 3))) & 3u) + (uint32_t)add);
}
void parts_bytes(const uint8_t *const *d_parts, const uint64_t *part_syms, int nparts, uint64_t q0, uint64_t q1, int type, uint8_t *d_out, hipStream_t s)
{
	B3M_REQUIRE(nparts >= 1 && nparts <= DICT_MAX_PARTS, "too many packed slices");
	PackedParts P{}; P.n = nparts; P.off[0] = 0;
	for (int i = 0; i < nparts; ++i) { P.p[i] = d_parts[i]; P.off[i + 1] = P.off[i] + part_syms[i]; }
	B3M_REQUIRE(q0 <= q1 && q1 <= P.off[nparts], "range outside the packed slices");
	if (q1 > q0) hipLaunchKernelGGL(k_parts_bytes, dim3(nblk(q1 - q0)), dim3(256), 0, s, P, q0, q1, type == B3M_INPUT_PACTERM ? 1 : 0, d_out);
}
// cuts[j] = #{q : sr[q] < bounds[j]} over an ascending rank list
__global__ void k_cuts(const uint32_t *sr, uint64_t cnt, const uint64_t *bounds, int nb, uint64_t *cuts)
{
	int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= nb) return;
	uint64_t lo = 0, hi = cnt;
	while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if ((uint64_t)sr[mid] < bounds[j]) lo = mid + 1; else hi = mid; }
	cuts[j] = lo;
}
void rank_cuts(const uint32_t *d_sr, uint64_t cnt, const uint64_t *d_bounds, int nb, uint64_t *d_cuts, hipStream_t s)
{
	if (nb) hipLaunchKernelGGL(k_cuts, dim3(nblk(nb, 64)), dim3(64), 0, s, d_sr, cnt, d_bounds, nb, d_cuts);
}

// d_scratch: 4*nb u32 (line counts) followed by 12*nb u32 (word counts)
template <typename P0T>
static void rank_build_t(const uint8_t *d_bwtL, uint64_t m, const P0T *d_p0, int type, uint32_t *d_blocks, uint32_t *d_scratch,
                         uint32_t *scan_ws, hipStream_t s)
{
	B3M_REQUIRE(type == B3M_INPUT_PAC || type == B3M_INPUT_PACTERM, "gap merge needs a 2-bit alphabet (pac/pacterm)");
	uint64_t nb = rank_blocks(m);
	int sub = (type == B3M_INPUT_PACTERM) ? 1 : 0;
	uint32_t *cnt = d_scratch, *wc = d_scratch + 4 * nb;
	hipLaunchKernelGGL(k_rank_pack_words<P0T>, dim3(nblk(nb * 12)), dim3(256), 0, s, d_bwtL, m, d_p0, sub, d_blocks, wc, nb * 12);
	hipLaunchKernelGGL(k_rank_line_counts, dim3(nblk(nb)), dim3(256), 0, s, wc, nb, cnt);
	for (int c = 0; c < 4; ++c) exclusive_sum_u32(cnt + c * nb, nb, scan_ws, s);
	hipLaunchKernelGGL(k_rank_headers, dim3(nblk(nb)), dim3(256), 0, s, d_blocks, cnt, nb);
}

void rank_build(const uint8_t *d_bwtL, uint64_t m, const uint32_t *d_p0, int type, uint32_t *d_blocks, uint32_t *d_scratch, uint32_t *scan_ws, hipStream_t s)
{
	rank_build_t<uint32_t>(d_bwtL, m, d_p0, type, d_blocks, d_scratch, scan_ws, s);
}
// ---- the dictionary with 64-bit counts (LF walks over the FINAL BWT: at 30 Gbase every base occurs 7.5 G times) -------------------------------
// A line header keeps 32 bits: the counts since the start of the line's SUPERBLOCK (2^sbshift lines = 2^sbshift x 192 symbols < 2^32), and a table
// sb[superblock][4] of 64-bit counts says what lies in front of the superblock.  The exclusive sums over the lines are taken modulo 2^32 as before
// (wrap-around is harmless: the difference of two of them less than 2^32 apart is exact); the table is the running sum of those differences.
__global__ __launch_bounds__(256) void k_rank_sb_diffs(const uint32_t *cnt, uint64_t nb, int sbshift, uint64_t nsb, uint64_t *sb)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= 4 * nsb) return;
	const uint64_t j = i >> 2; const uint32_t c = (uint32_t)(i & 3);
	sb[i] = j == 0 ? 0ull : (uint64_t)(uint32_t)(cnt[c * nb + (j << sbshift)] - cnt[c * nb + ((j - 1) << sbshift)]);
}
__global__ void k_rank_sb_prefix(uint64_t *sb, uint64_t nsb)
{
	const uint32_t c = threadIdx.x;
	if (c >= 4) return;
	uint64_t acc = 0;
	for (uint64_t j = 0; j < nsb; ++j) { acc += sb[4 * j + c]; sb[4 * j + c] = acc; }
}
__global__ void k_rank_headers_sb(uint32_t *blocks, const uint32_t *cnt, uint64_t nb, int sbshift)
{
	uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= nb) return;
	const uint64_t b0 = (b >> sbshift) << sbshift;
	uint32_t *out = blocks + b * 16;
	#pragma unroll
	for (int c = 0; c < 4; ++c) out[c] = cnt[c * nb + b] - cnt[c * nb + b0];
}
int rank_sb_shift()
{
	// (test switch: small superblocks put texts the oracle can sort across many of them)
	const int v = (int)sw().RANK_SB_SHIFT.get(16);
	return v < 1 ? 1 : (v > 24 ? 24 : v);
}
uint64_t rank_superblocks(uint64_t m) { return (rank_blocks(m) >> rank_sb_shift()) + 1; }
// d_p0: rank of the terminator entry, a 64-bit device scalar (~0: none); d_sb: 4 * rank_superblocks(m) u64
void rank_build64(const uint8_t *d_bwt, uint64_t m, const uint64_t *d_p0, int type, uint32_t *d_blocks, uint32_t *d_scratch, uint32_t *scan_ws, uint64_t *d_sb, hipStream_t s)
{
	B3M_REQUIRE(type == B3M_INPUT_PAC || type == B3M_INPUT_PACTERM, "the LF walk's rank dictionary needs a 2-bit alphabet (pac/pacterm)");
	const uint64_t nb = rank_blocks(m), nsb = rank_superblocks(m);
	const int sub = (type == B3M_INPUT_PACTERM) ? 1 : 0, sh = rank_sb_shift();
	uint32_t *cnt = d_scratch, *wc = d_scratch + 4 * nb;
	hipLaunchKernelGGL(k_rank_pack_words<uint64_t>, dim3(nblk(nb * 12)), dim3(256), 0, s, d_bwt, m, d_p0, sub, d_blocks, wc, nb * 12);
	hipLaunchKernelGGL(k_rank_line_counts, dim3(nblk(nb)), dim3(256), 0, s, wc, nb, cnt);
	for (int c = 0; c < 4; ++c) exclusive_sum_u32(cnt + c * nb, nb, scan_ws, s);
	hipLaunchKernelGGL(k_rank_sb_diffs, dim3(nblk(4 * nsb)), dim3(256), 0, s, cnt, nb, sh, nsb, d_sb);
	hipLaunchKernelGGL(k_rank_sb_prefix, dim3(1), dim3(64), 0, s, d_sb, nsb);
	hipLaunchKernelGGL(k_rank_headers_sb, dim3(nblk(nb)), dim3(256), 0, s, d_blocks, cnt, nb, sh);
}

// the dictionary of BWT(L) from its packed slices (see k_pack2); d_scratch / scan_ws as rank_build
void rank_build_from_parts(const uint8_t *const *d_parts, const uint64_t *part_syms, int nparts, uint64_t m, uint64_t p0, uint32_t *d_blocks,
                           uint32_t *d_scratch, uint32_t *scan_ws, hipStream_t s)
{
	B3M_REQUIRE(nparts >= 1 && nparts <= DICT_MAX_PARTS, "too many packed slices for one dictionary");
	PackedParts P{}; P.n = nparts; P.off[0] = 0;
	for (int i = 0; i < nparts; ++i) { P.p[i] = d_parts[i]; P.off[i + 1] = P.off[i] + part_syms[i]; }
	B3M_REQUIRE(P.off[nparts] == m, "packed slices do not add up to |L|");
	uint64_t nb = rank_blocks(m);
	uint32_t *cnt = d_scratch, *wc = d_scratch + 4 * nb;
	hipLaunchKernelGGL(k_rank_words_from_parts, dim3(nblk(nb * 12)), dim3(256), 0, s, P, m, p0, d_blocks, wc, nb * 12);
	hipLaunchKernelGGL(k_rank_line_counts, dim3(nblk(nb)), dim3(256), 0, s, wc, nb, cnt);
	for (int c = 0; c < 4; ++c) exclusive_sum_u32(cnt + c * nb, nb, scan_ws, s);
	hipLaunchKernelGGL(k_rank_headers, dim3(nblk(nb)), dim3(256), 0, s, d_blocks, cnt, nb);
}

// rank_c(BWT_L, rho): occurrences of code c in BWT_L[0, rho), p0 entry never counted.
// rank_c(B,r) counts B[0,r) -- convention of /root/reference/src/bwtdecodeblock.cpp:232-241.
template <typename RT>
__device__ __forceinline__ uint32_t rank_query_t(const uint32_t *blocks, RT p0, uint32_t code, RT rho)
{
	const RT b = rho / RANK_BLK_SYMS; const uint32_t off = (uint32_t)(rho - b * RANK_BLK_SYMS);
	const uint4 *line = reinterpret_cast<const uint4 *>(blocks + (uint64_t)b * 16);
	uint4 h = line[0], p0v = line[1], p1v = line[2], p2v = line[3];
	uint32_t hd[4] = {h.x, h.y, h.z, h.w};
	uint32_t w[12] = {p0v.x, p0v.y, p0v.z, p0v.w, p1v.x, p1v.y, p1v.z, p1v.w, p2v.x, p2v.y, p2v.z, p2v.w};
	uint32_t cnt = hd[code];
	uint32_t pat = code * 0x55555555u;
	#pragma unroll
	for (int i = 0; i < 12; ++i) {
		int nsym = (int)off - 16 * i;
		uint32_t mask = nsym >= 16 ? 0x55555555u : (nsym <= 0 ? 0u : (0x55555555u & ((1u << (2 * nsym)) - 1u)));
		uint32_t y = w[i] ^ pat;
		uint32_t mm = ~(y | (y >> 1)) & mask;
		cnt += __popc(mm);
	}
	// the p0 entry is stored as code 0 but was not counted in the headers
	if (code == 0 && p0 >= b * RANK_BLK_SYMS && p0 < rho) cnt -= 1;
	return cnt;
}
// the same with 64-bit counts: superblock table + line header (counts since the superblock's start) + the line's own symbols
__device__ __forceinline__ uint64_t rank_query64(const uint32_t *blocks, const uint64_t *sb, int sbshift, uint64_t p0, uint32_t code, uint64_t rho)
{
	const uint64_t b = rho / RANK_BLK_SYMS;
	return sb[((b >> sbshift) << 2) + code] + (uint64_t)rank_query_t<uint64_t>(blocks, p0, code, rho);
}
__device__ __forceinline__ uint32_t rank_query(const RankDict &rd, uint32_t p0, uint32_t code, uint32_t rho) { return rank_query_t<uint32_t>(rd.blocks, p0, code, rho); }

// ------------------------------------------------------------------------------------------------
// K7 LF walk
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gap_walk(GapWalkArgs A)
{
	uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= A.num_chains) return;
	const TextView &T = A.T;
	const uint64_t Rbeg = A.a + A.m, Rend = Rbeg + A.mr;
	uint64_t hi = Rend - (A.chain_first + c) * A.chain_len;   // chain covers positions [lo, hi)
	uint64_t lo = (hi - Rbeg > A.chain_len) ? hi - A.chain_len : Rbeg;
	uint32_t rho = A.anchors[c];                          // rank_L(S_hi)
	const uint32_t p0 = *A.rd.p0ptr;
	const uint32_t last_code = base_at(T, A.a + A.m - 1);   // T[a+m-1], never the terminator
	const bool term = T.type == B3M_INPUT_PACTERM;
	// The only load a step must wait for is its rank line: the text (16 bases per big-endian word) and the
	// gt bits (32 per word) are read a word AHEAD of the walk, which moves towards lower positions.
	const uint32_t *tw = reinterpret_cast<const uint32_t *>(T.b);
	uint64_t wi = (hi - 1) >> 4;
	uint32_t word = hi > lo ? __builtin_bswap32(tw[wi]) : 0u, wnext = (hi > lo && wi > 0) ? __builtin_bswap32(tw[wi - 1]) : 0u;
	uint64_t gtw_idx = hi >> 5;
	uint32_t gtw = A.gt_cur[gtw_idx], gtnext = gtw_idx > 0 ? A.gt_cur[gtw_idx - 1] : 0u;
	uint32_t outw = 0; uint64_t outw_idx = ~0ull;        // gt_next word being assembled
	// the end bit of the parent: gt_P[Rend] is stored per node by the host from the anchor
	uint64_t t = 0;
	for (uint64_t j = hi; j-- > lo; ++t) {
		// symbol T[j]
		if ((j >> 4) != wi) { wi = j >> 4; word = wnext; wnext = wi > 0 ? __builtin_bswap32(tw[wi - 1]) : 0u; }
		const bool is_term = term && j == T.n - 1;
		const uint32_t code = is_term ? 0u : (word >> (2u * (15u - (uint32_t)(j & 15)))) & 3u;
		// gt_R[j+1] relative to S_{Rbeg}; for j+1 == Rend it is the node's end bit kept at bit Rend
		uint64_t g = j + 1;
		if ((g >> 5) != gtw_idx) { gtw_idx = g >> 5; gtw = gtnext; gtnext = gtw_idx > 0 ? A.gt_cur[gtw_idx - 1] : 0u; }
		uint32_t gtbit = (gtw >> (g & 31)) & 1u;
		uint32_t nrho;
		if (is_term) nrho = 0;                              // S_{n-1} = "$..." is the smallest rotation
		else nrho = A.rd.C[code] + rank_query(A.rd, p0, code, rho) + ((last_code == code && gtbit) ? 1u : 0u);
		rho = nrho;
		A.ranks_out[t * A.num_chains + c] = rho;
		// gt_P[j] = [S_j > S_a] = rank_L(j) > p0_L
		uint32_t ob = rho > p0 ? 1u : 0u;
		if ((j >> 5) != outw_idx) {
			if (outw_idx != ~0ull && outw) atomicOr(&A.gt_next[outw_idx], outw);
			outw_idx = j >> 5; outw = 0;
		}
		outw |= ob << (j & 31);
	}
	if (outw_idx != ~0ull && outw) atomicOr(&A.gt_next[outw_idx], outw);
	// pad the unused slots of a short last chain
	for (; t < A.chain_len; ++t) A.ranks_out[t * A.num_chains + c] = 0xFFFFFFFFu;
}

// FOUR LANES PER CHAIN.  The one-lane walk asks the vector memory pipe for 4 x 16 bytes of a different line in every lane: 256
// line look-ups per wave and step.  Here the four lanes of a quad share a chain and a step is ONE 16-byte load per lane (lane 0
// of the quad: the counts, lanes 1..3: four payload words each): 16 lines per wave-instruction, all 64 bytes of each used.  The
// partial counts meet by two quad shuffles; every lane then knows the new rank.  Ranks leave as whole lines: lane k of a quad
// keeps the rank of step 4i + k and the quads of a wave store 16 x 16 contiguous bytes every fourth step (layout
// [step / 4][chain][4]; the rank list is sorted afterwards, its order is free).
__global__ __launch_bounds__(256) void k_gap_walk4(GapWalkArgs A)
{
	const uint64_t c = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
	const uint32_t sub = threadIdx.x & 3u;
	if (c >= A.num_chains) return;                            // whole quads leave together
	const TextView &T = A.T;
	const uint64_t Rbeg = A.a + A.m, Rend = Rbeg + A.mr;
	uint64_t hi = Rend - (A.chain_first + c) * A.chain_len;   // chain covers positions [lo, hi)
	uint64_t lo = (hi - Rbeg > A.chain_len) ? hi - A.chain_len : Rbeg;
	uint32_t rho = A.anchors[c];                              // rank_L(S_hi)
	const uint32_t p0 = *A.rd.p0ptr;
	const uint32_t last_code = base_at(T, A.a + A.m - 1);
	const bool term = T.type == B3M_INPUT_PACTERM;
	const uint32_t *tw = reinterpret_cast<const uint32_t *>(T.b);
	uint64_t wi = (hi - 1) >> 4;
	uint32_t word = hi > lo ? __builtin_bswap32(tw[wi]) : 0u, wnext = (hi > lo && wi > 0) ? __builtin_bswap32(tw[wi - 1]) : 0u;
	uint64_t gtw_idx = hi >> 5;
	uint32_t gtw = A.gt_cur[gtw_idx], gtnext = gtw_idx > 0 ? A.gt_cur[gtw_idx - 1] : 0u;
	uint32_t outw = 0; uint64_t outw_idx = ~0ull;
	const uint32_t C0 = A.rd.C[0], C1 = A.rd.C[1], C2 = A.rd.C[2], C3 = A.rd.C[3];
	uint32_t keep = 0xFFFFFFFFu;
	uint64_t t = 0;
	const uint64_t steps4 = (A.chain_len + 3) & ~3ull;       // slots of a chain, padded to whole quads of steps
	uint32_t *outp = A.ranks_out + c * 4 + sub;               // + (t / 4) * 4 * num_chains
	for (uint64_t j = hi; j-- > lo; ++t) {
		if ((j >> 4) != wi) { wi = j >> 4; word = wnext; wnext = wi > 0 ? __builtin_bswap32(tw[wi - 1]) : 0u; }
		const bool is_term = term && j == T.n - 1;
		const uint32_t code = is_term ? 0u : (word >> (2u * (15u - (uint32_t)(j & 15)))) & 3u;
		uint64_t g = j + 1;
		if ((g >> 5) != gtw_idx) { gtw_idx = g >> 5; gtw = gtnext; gtnext = gtw_idx > 0 ? A.gt_cur[gtw_idx - 1] : 0u; }
		const uint32_t gtbit = (gtw >> (g & 31)) & 1u;
		// one rank line, 16 bytes per lane
		const uint32_t b = rho / RANK_BLK_SYMS, off = rho - b * RANK_BLK_SYMS;
		const uint4 q = reinterpret_cast<const uint4 *>(A.rd.blocks + (uint64_t)b * 16)[sub];
		uint32_t part;
		if (sub == 0) part = code == 0 ? q.x : code == 1 ? q.y : code == 2 ? q.z : q.w;
		else {
			const uint32_t pat = code * 0x55555555u;
			const int base = 64 * ((int)sub - 1);             // symbols before this lane's four words
			const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
			part = 0;
			#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int nsym = (int)off - base - 16 * i;
				const uint32_t mask = nsym >= 16 ? 0x55555555u : (nsym <= 0 ? 0u : (0x55555555u & ((1u << (2 * nsym)) - 1u)));
				const uint32_t y = w4[i] ^ pat;
				part += __popc(~(y | (y >> 1)) & mask);
			}
		}
		part += __shfl_xor(part, 1, 64);
		part += __shfl_xor(part, 2, 64);
		uint32_t cnt = part;
		if (code == 0 && p0 >= b * RANK_BLK_SYMS && p0 < rho) cnt -= 1;          // the p0 entry is stored as code 0 but never counted
		const uint32_t Cc = code == 0 ? C0 : code == 1 ? C1 : code == 2 ? C2 : C3;
		rho = is_term ? 0u : Cc + cnt + ((last_code == code && gtbit) ? 1u : 0u);
		if ((t & 3) == sub) keep = rho;
		if ((t & 3) == 3) { outp[(t >> 2) * 4 * A.num_chains] = keep; keep = 0xFFFFFFFFu; }
		if (sub == 0) {
			const uint32_t ob = rho > p0 ? 1u : 0u;
			if ((j >> 5) != outw_idx) {
				if (outw_idx != ~0ull && outw) atomicOr(&A.gt_next[outw_idx], outw);
				outw_idx = j >> 5; outw = 0;
			}
			outw |= ob << (j & 31);
		}
	}
	if (sub == 0 && outw_idx != ~0ull && outw) atomicOr(&A.gt_next[outw_idx], outw);
	// the open quad of steps, then the unused slots of a short chain
	if (t & 3) { outp[(t >> 2) * 4 * A.num_chains] = keep; t = (t + 3) & ~3ull; }
	for (; t < steps4; t += 4) outp[(t >> 2) * 4 * A.num_chains] = 0xFFFFFFFFu;
}

// slots the walk writes for a merge's chain table: the caller sizes and sorts this many entries (unused ones hold 0xFFFFFFFF)
uint64_t gap_walk_slots(uint64_t num_chains, uint64_t chain_len) { return num_chains * ((chain_len + 3) & ~3ull); }

void gap_walk(const GapWalkArgs &A, hipStream_t s)
{
	if (!A.num_chains) return;
	const int one_lane = (int)sw().WALK_ONE_LANE.get(0);   // measurement switch
	if (one_lane) {
		// the one-lane kernel fills [chain_len][num_chains]; the rest of the padded table must read as unused
		const uint64_t used = A.num_chains * A.chain_len, all = gap_walk_slots(A.num_chains, A.chain_len);
		if (all > used) fill_u32(A.ranks_out + used, 0xFFFFFFFFu, all - used, s);
		hipLaunchKernelGGL(k_gap_walk, dim3(nblk(A.num_chains, 256)), dim3(256), 0, s, A);
	} else hipLaunchKernelGGL(k_gap_walk4, dim3(nblk(A.num_chains * 4, 256)), dim3(256), 0, s, A);
}

// ------------------------------------------------------------------------------------------------
// BYTE ALPHABETS (bytestream, compactstream above 2 bits): rank structure and LF walk of the gap merge (rows a8/a9 for
// sigma <= 256).  BWT(L) stays as bytes; occ[b][c] = occurrences of c in BWT_L[0, 64 b) (the p0 entry never counted),
// one 1-KB row per 64 symbols (16 B per symbol of L).  A rank query = one count (4 bytes of a random row) + the
// <= 64 bytes of the block before the rank (one line), compared as bytes with SWAR arithmetic.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t BR_BLK = 64;                 // symbols per count row
constexpr uint32_t BR_TILE = 256;               // rows per build tile (16 Ki symbols)
uint64_t byte_rank_rows(uint64_t m) { return m / BR_BLK + 1; }
uint64_t byte_rank_tiles(uint64_t m) { return (byte_rank_rows(m) + BR_TILE - 1) / BR_TILE; }
// pass 1: histogram of every tile -> th[c][tile]
__global__ __launch_bounds__(256) void k_br_tile_hist(const uint8_t *bwt, uint64_t m, const uint32_t *p0ptr, uint32_t *th, uint64_t ntiles)
{
	__shared__ uint32_t h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	const uint64_t t = blockIdx.x, lo = t * BR_TILE * BR_BLK, hi = lo + (uint64_t)BR_TILE * BR_BLK < m ? lo + (uint64_t)BR_TILE * BR_BLK : m;
	const uint64_t p0 = *p0ptr;
	for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) if (i != p0) atomicAdd(&h[bwt[i]], 1u);
	__syncthreads();
	th[(uint64_t)threadIdx.x * ntiles + t] = h[threadIdx.x];
}
// pass 2: exclusive scan of every symbol's row of tile counts (one workgroup per symbol)
__global__ __launch_bounds__(256) void k_br_colscan(uint32_t *th, uint64_t ntiles)
{
	__shared__ uint32_t wsum[4];
	__shared__ uint32_t carry;
	uint32_t *row = th + (uint64_t)blockIdx.x * ntiles;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	if (tid == 0) carry = 0;
	__syncthreads();
	for (uint64_t base = 0; base < ntiles; base += 256) {
		const uint64_t i = base + tid;
		const uint32_t v = i < ntiles ? row[i] : 0u;
		const uint32_t inc = wave_incl_sum(v);
		if (lane == 63) wsum[wave] = inc;
		__syncthreads();
		uint32_t pre = carry;
		for (uint32_t w = 0; w < wave; ++w) pre += wsum[w];
		if (i < ntiles) row[i] = pre + inc - v;
		__syncthreads();
		if (tid == 255) carry = pre + inc;
		__syncthreads();
	}
}
// pass 3: the count rows of a tile, row by row (thread c keeps the running count of symbol c)
__global__ __launch_bounds__(256) void k_br_rows(const uint8_t *bwt, uint64_t m, const uint32_t *p0ptr, const uint32_t *th, uint64_t ntiles, uint32_t *occ, uint64_t nrows)
{
	__shared__ uint32_t run[256];
	const uint64_t t = blockIdx.x;
	const uint32_t c = threadIdx.x;
	run[c] = th[(uint64_t)c * ntiles + t];
	const uint64_t p0 = *p0ptr;
	__syncthreads();
	for (uint32_t r = 0; r < BR_TILE; ++r) {
		const uint64_t row = t * BR_TILE + r;
		if (row >= nrows) break;
		occ[row * 256 + c] = run[c];
		__syncthreads();
		const uint64_t i = row * BR_BLK + c;
		if (c < BR_BLK && i < m && i != p0) atomicAdd(&run[bwt[i]], 1u);
		__syncthreads();
	}
}
// d_occ: 256 * byte_rank_rows(m) u32; d_th: 256 * byte_rank_tiles(m) u32 scratch
void byte_rank_build(const uint8_t *d_bwtL, uint64_t m, const uint32_t *d_p0, uint32_t *d_occ, uint32_t *d_th, hipStream_t s)
{
	const uint64_t nrows = byte_rank_rows(m), ntiles = byte_rank_tiles(m);
	hipLaunchKernelGGL(k_br_tile_hist, dim3((uint32_t)ntiles), dim3(256), 0, s, d_bwtL, m, d_p0, d_th, ntiles);
	hipLaunchKernelGGL(k_br_colscan, dim3(256), dim3(256), 0, s, d_th, ntiles);
	hipLaunchKernelGGL(k_br_rows, dim3((uint32_t)ntiles), dim3(256), 0, s, d_bwtL, m, d_p0, d_th, ntiles, d_occ, nrows);
}
// bytes of w equal to c (SWAR): x = w ^ cccc; a byte of x is zero iff equal
__device__ __forceinline__ uint32_t eq_bytes(uint32_t w, uint32_t c4)
{
	const uint32_t x = w ^ c4;
	const uint32_t t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;      // high bit of every byte set iff the byte is non-zero
	return __popc(~t & 0x80808080u);
}
__device__ __forceinline__ uint32_t byte_rank_query(const uint32_t *occ, const uint8_t *bwt, uint32_t p0, uint32_t c, uint32_t rho)
{
	const uint32_t b = rho / BR_BLK, off = rho - b * BR_BLK;
	uint32_t cnt = occ[(uint64_t)b * 256 + c];
	const uint4 *line = reinterpret_cast<const uint4 *>(bwt + (uint64_t)b * BR_BLK);     // bwt is 64-byte aligned and padded
	const uint32_t c4 = c * 0x01010101u;
	#pragma unroll
	for (int q = 0; q < 4; ++q) {
		if ((uint32_t)(16 * q) >= off) break;
		const uint4 v = line[q];
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
		#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int nb = (int)off - 16 * q - 4 * i;          // bytes of this word before the rank
			if (nb <= 0) break;
			uint32_t ww = w[i];
			if (nb < 4) ww = (ww & ((1u << (8 * nb)) - 1u)) | (~c4 & ~((1u << (8 * nb)) - 1u));     // bytes at and behind the rank never match
			cnt += eq_bytes(ww, c4);
		}
	}
	if (p0 >= b * BR_BLK && p0 < rho && bwt[p0] == c) cnt -= 1;      // the p0 entry is never counted
	return cnt;
}
struct ByteWalkArgs {
	TextView T;
	const uint32_t *occ; const uint8_t *bwtL; const uint32_t *p0ptr; const uint32_t *C;      // C: 256 symbol starts of L's text (device)
	uint64_t a, m, mr;
	const uint32_t *gt_cur; uint32_t *gt_next;
	const uint32_t *anchors; uint64_t num_chains, chain_len, chain_first;
	uint32_t *ranks_out;       // [chain_len][num_chains], unused slots 0xFFFFFFFF (then up to gap_walk_slots: filled by the caller)
};
__global__ __launch_bounds__(256) void k_gap_walk_bytes(ByteWalkArgs A)
{
	uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= A.num_chains) return;
	const TextView &T = A.T;
	const uint64_t Rbeg = A.a + A.m, Rend = Rbeg + A.mr;
	uint64_t hi = Rend - (A.chain_first + c) * A.chain_len;
	uint64_t lo = (hi - Rbeg > A.chain_len) ? hi - A.chain_len : Rbeg;
	uint32_t rho = A.anchors[c];
	const uint32_t p0 = *A.p0ptr;
	const uint32_t last_sym = T.b[A.a + A.m - 1];
	uint32_t outw = 0; uint64_t outw_idx = ~0ull;
	uint64_t t = 0;
	for (uint64_t j = hi; j-- > lo; ++t) {
		const uint32_t sym = T.b[j];
		const uint64_t g = j + 1;
		const uint32_t gtbit = (A.gt_cur[g >> 5] >> (g & 31)) & 1u;
		rho = A.C[sym] + byte_rank_query(A.occ, A.bwtL, p0, sym, rho) + ((last_sym == sym && gtbit) ? 1u : 0u);
		A.ranks_out[t * A.num_chains + c] = rho;
		const uint32_t ob = rho > p0 ? 1u : 0u;
		if ((j >> 5) != outw_idx) {
			if (outw_idx != ~0ull && outw) atomicOr(&A.gt_next[outw_idx], outw);
			outw_idx = j >> 5; outw = 0;
		}
		outw |= ob << (j & 31);
	}
	if (outw_idx != ~0ull && outw) atomicOr(&A.gt_next[outw_idx], outw);
	for (; t < A.chain_len; ++t) A.ranks_out[t * A.num_chains + c] = 0xFFFFFFFFu;
}
void gap_walk_bytes(const TextView &T, const uint32_t *d_occ, const uint8_t *d_bwtL, const uint32_t *d_p0, const uint32_t *d_C, uint64_t a, uint64_t m, uint64_t mr,
                    const uint32_t *gt_cur, uint32_t *gt_next, const uint32_t *d_anchors, uint64_t num_chains, uint64_t chain_len, uint32_t *ranks_out, hipStream_t s, uint64_t chain_first)
{
	if (!num_chains) return;
	ByteWalkArgs A{T, d_occ, d_bwtL, d_p0, d_C, a, m, mr, gt_cur, gt_next, d_anchors, num_chains, chain_len, chain_first, ranks_out};
	const uint64_t used = num_chains * chain_len, all = gap_walk_slots(num_chains, chain_len);
	if (all > used) fill_u32(ranks_out + used, 0xFFFFFFFFu, all - used, s);
	hipLaunchKernelGGL(k_gap_walk_bytes, dim3(nblk(num_chains, 256)), dim3(256), 0, s, A);
}

// ------------------------------------------------------------------------------------------------
// LF walks over the FINAL BWT (the sampled SA stage of bwtcomputessa,
// /root/reference/src/bwtcomputessa.cpp:39-51, and the verification walk of checkbwt,
// /root/reference/src/checkbwt.cpp:176-241).  Chain k starts at anchor k+1 (rank r of the rotation
// at position p) and walks back to anchor k: at every step the BWT symbol at r must equal T[p-1]
// (checked), lf(r) = C[c] + rank_c(BWT, r) (convention /root/reference/src/bwtdecodeblock.cpp:280-284),
// and SA[r] = p is stored when r is a multiple of the SA sampling rate.  The rank the chain
// arrives with must be anchor k's (checked).
// ------------------------------------------------------------------------------------------------
struct RankDict64 { const uint32_t *blocks; const uint64_t *sb; int sbshift; uint64_t m; uint64_t p0; uint64_t C[4]; };     // p0 = ~0: no terminator entry
struct LfWalkArgs {
	TextView T; RankDict64 rd;
	const uint64_t *anchors;        // (rank, pos) pairs sorted by pos; pos of anchor 0 is 0
	uint64_t nanchors;
	uint64_t sarate;                // 0: no SA samples
	uint64_t *sa_out;               // ceil(n / sarate) positions
	unsigned long long *mismatch;   // [0] symbol mismatches, [1] chain closure mismatches
	uint64_t isarate; uint64_t *isa_out; int nocheck;      // LfExtra
	uint64_t kfirst, kcount;        // the chains this launch walks
};
__global__ __launch_bounds__(256) void k_lf_walk(LfWalkArgs A)
{
	uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= A.kcount) return;
	k += A.kfirst;
	const TextView &T = A.T;
	const bool term = T.type == B3M_INPUT_PACTERM;
	const uint64_t p0 = A.rd.p0;
	uint64_t kn = k + 1 == A.nanchors ? 0 : k + 1;
	uint64_t r = A.anchors[2 * kn];
	uint64_t p = kn ? A.anchors[2 * kn + 1] : T.n;     // the last chain starts at position n == 0
	const uint64_t pend = A.anchors[2 * k + 1];
	uint32_t bad = 0;
	// the text is read 16 bases (one big-endian word) at a time, one word ahead of the walk
	const bool chk = !A.nocheck;
	const uint32_t *tw = reinterpret_cast<const uint32_t *>(T.b);
	uint64_t wi = p > pend ? (p - 1) >> 4 : 0;
	uint32_t word = (chk && p > pend) ? __builtin_bswap32(tw[wi]) : 0u, wnext = (chk && p > pend && wi > 0) ? __builtin_bswap32(tw[wi - 1]) : 0u;
	while (p > pend) {
		if (A.sarate && r % A.sarate == 0) A.sa_out[r / A.sarate] = (p == T.n ? 0 : p);
		if (A.isarate && (p == T.n ? 0 : p) % A.isarate == 0) A.isa_out[(p == T.n ? 0 : p) / A.isarate] = r;
		if (r >= T.n) { bad += 1; break; }                 // a corrupt BWT can send the walk anywhere: stop the chain (counted: symbol mismatch + closure)
		const uint64_t q = p - 1;
		if (chk && (q >> 4) != wi) { wi = q >> 4; word = wnext; wnext = wi > 0 ? __builtin_bswap32(tw[wi - 1]) : 0u; }
		const uint32_t base = (word >> (2u * (15u - (uint32_t)(q & 15)))) & 3u;
		const uint32_t st = term ? (q == T.n - 1 ? 0u : base + 1u) : base;     // symbol preceding the rotation at p
		uint64_t nr;
		if (term && r == p0) {                          // the BWT holds the terminator here
			bad += (chk && st != 0) ? 1u : 0u;
			nr = 0;                                      // the rotation starting with the terminator is the smallest
		} else {
			const uint64_t b = r / RANK_BLK_SYMS; const uint32_t off = (uint32_t)(r - b * RANK_BLK_SYMS);
			uint32_t word = A.rd.blocks[b * 16 + 4 + (off >> 4)];
			uint32_t code = (word >> (2 * (off & 15))) & 3u;
			uint32_t sb = term ? code + 1 : code;
			bad += (chk && sb != st) ? 1u : 0u;
			nr = A.rd.C[code] + rank_query64(A.rd.blocks, A.rd.sb, A.rd.sbshift, p0, code, r);
		}
		r = nr; --p;
	}
	if (bad) atomicAdd(&A.mismatch[0], (unsigned long long)bad);
	if (r != A.anchors[2 * k]) atomicAdd(&A.mismatch[1], 1ull);
}
void lf_walk(const TextView &T, const uint32_t *d_blocks, const uint64_t *d_sb, uint64_t p0, const uint64_t C[4], const uint64_t *d_anchors, uint64_t nanchors, uint64_t sarate,
             uint64_t *d_sa_out, unsigned long long *d_mismatch, hipStream_t s, const LfExtra &X)
{
	LfWalkArgs A{T, RankDict64{d_blocks, d_sb, rank_sb_shift(), T.n, p0, {C[0], C[1], C[2], C[3]}}, d_anchors, nanchors, sarate, d_sa_out, d_mismatch,
	             X.isarate, X.d_isa_out, X.nocheck ? 1 : 0, 0, nanchors};
	if (X.chain_count) { A.kfirst = std::min(X.chain_first, nanchors); A.kcount = std::min(X.chain_count, nanchors - A.kfirst); }
	if (A.kcount) hipLaunchKernelGGL(k_lf_walk, dim3(nblk(A.kcount)), dim3(256), 0, s, A);
}

// Byte alphabets (byte LF walk): no rank dictionary -- the whole LF array is built by ONE
// stable counting sort of the BWT by symbol (lf[i] = slot of i in that order, /root/reference/src/bwtdecodeblock.cpp:280-284)
// and the chains walk it directly.
__global__ void k_bwt_keys(const uint8_t *bwt, uint64_t n, uint64_t *keys, uint32_t *vals, int wide)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { keys[i] = wide ? reinterpret_cast<const uint32_t *>(bwt)[i] : bwt[i]; vals[i] = (uint32_t)i; }
}
__global__ void k_lf_from_sorted(const uint32_t *sorted_idx, uint64_t n, uint32_t *lf)
{
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n) lf[sorted_idx[j]] = (uint32_t)j;
}
struct LfWalkBytesArgs {
	TextView T; const uint8_t *bwt; const uint32_t *lf;
	const uint64_t *anchors; uint64_t nanchors; uint64_t sarate; uint64_t *sa_out; unsigned long long *mismatch;
	uint64_t isarate; uint64_t *isa_out; int nocheck;      // LfExtra
};
__global__ __launch_bounds__(256) void k_lf_walk_bytes(LfWalkBytesArgs A)
{
	uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= A.nanchors) return;
	const TextView &T = A.T;
	uint64_t kn = k + 1 == A.nanchors ? 0 : k + 1;
	uint32_t r = (uint32_t)A.anchors[2 * kn];
	uint64_t p = kn ? A.anchors[2 * kn + 1] : T.n;     // the last chain starts at position n == 0
	const uint64_t pend = A.anchors[2 * k + 1];
	uint32_t bad = 0;
	while (p > pend) {
		if (A.sarate && r % A.sarate == 0) A.sa_out[r / A.sarate] = (p == T.n ? 0 : p);
		if (A.isarate && (p == T.n ? 0 : p) % A.isarate == 0) A.isa_out[(p == T.n ? 0 : p) / A.isarate] = r;
		if (!A.nocheck) {
			const uint32_t have = T.type == B3M_INPUT_SYMBOLS32 ? reinterpret_cast<const uint32_t *>(A.bwt)[r] : A.bwt[r];
			bad += have != sym_at(T, p - 1);            // the symbol preceding the rotation at p
		}
		r = A.lf[r]; --p;
	}
	if (bad) atomicAdd(&A.mismatch[0], (unsigned long long)bad);
	if (r != (uint32_t)A.anchors[2 * k]) atomicAdd(&A.mismatch[1], 1ull);
}
// d_keys/d_keys2 (n u64 each), d_vals/d_vals2 (n u32 each): sort buffers; d_lf: n u32
void lf_walk_bytes(const TextView &T, const uint8_t *d_bwt, uint64_t *d_keys, uint64_t *d_keys2, uint32_t *d_vals, uint32_t *d_vals2, uint32_t *d_lf,
                   RadixWs &rw, const uint64_t *d_anchors, uint64_t nanchors, uint64_t sarate, uint64_t *d_sa_out, unsigned long long *d_mismatch, hipStream_t s, const LfExtra &X)
{
	const uint64_t n = T.n;
	const int wide = T.type == B3M_INPUT_SYMBOLS32;      // (d_bwt: n uint32 symbols below 2^21)
	hipLaunchKernelGGL(k_bwt_keys, dim3(nblk(n)), dim3(256), 0, s, d_bwt, n, d_keys, d_vals, wide);
	uint64_t *rk; uint32_t *rv;
	radix_sort_u64(d_keys, d_vals, d_keys2, d_vals2, n, wide ? WIDE_BITS : 8, rw, s, nullptr, &rk, &rv);
	hipLaunchKernelGGL(k_lf_from_sorted, dim3(nblk(n)), dim3(256), 0, s, rv, n, d_lf);
	LfWalkBytesArgs A{T, d_bwt, d_lf, d_anchors, nanchors, sarate, d_sa_out, d_mismatch, X.isarate, X.d_isa_out, X.nocheck ? 1 : 0};
	if (nanchors) hipLaunchKernelGGL(k_lf_walk_bytes, dim3(nblk(nanchors)), dim3(256), 0, s, A);
}

// ------------------------------------------------------------------------------------------------
// K9 interleave
// ------------------------------------------------------------------------------------------------
constexpr int IL_THREADS = 256;
constexpr int IL_ITEMS = 16;
constexpr int IL_TILE = IL_THREADS * IL_ITEMS;   // 4096 output symbols per workgroup

uint64_t interleave_tiles(uint64_t total) { return (total + IL_TILE - 1) / IL_TILE; }

// tile_q[t] = number of R suffixes whose output position q + sr[q] is < t*IL_TILE: f(q) = q + sr[q] is strictly increasing, so every tile border
// finds its q by a binary search of its own (31 dependent loads for one of the 0.7 M tiles of a 3-Gbase level; the first form streamed the whole
// rank list through one thread per rank: 8.6 of the 207 ms of C3 with 8 blocks)
__global__ void k_tile_bounds(const uint32_t *sr, uint64_t mr, uint32_t *tile_q, uint64_t tiles)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t > tiles) return;
	const uint64_t border = t * IL_TILE;
	uint64_t lo = 0, hi = mr;                                       // smallest q with f(q) >= border (mr: none)
	while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (mid + (uint64_t)sr[mid] < border) lo = mid + 1; else hi = mid; }
	tile_q[t] = (uint32_t)lo;
}

// One workgroup per tile of 4096 output symbols.  The symbols of R drop into their places in LDS (place = q + sr[q]); the holes take the
// next symbols of BWT(L), whose segment for the tile is staged in LDS by whole words; every thread then assembles its 16 output symbols
// in registers and stores them as ONE 16-byte word (the first form stored single bytes: 8.7 ms per level of C3 with 8 blocks).
__global__ __launch_bounds__(IL_THREADS) void k_interleave(const uint32_t *sr, const uint8_t *bwtL, uint64_t m, const uint8_t *bwtR,
                                                             uint64_t mr, uint8_t *out, const uint32_t *tile_q)
{
	__shared__ __attribute__((aligned(16))) uint8_t sym[IL_TILE];
	__shared__ __attribute__((aligned(16))) uint8_t isr[IL_TILE];
	__shared__ __attribute__((aligned(16))) uint8_t lseg[IL_TILE + 32];
	__shared__ uint32_t lds[IL_THREADS / 64];
	const uint32_t tid = threadIdx.x;
	const uint64_t total = m + mr;
	const uint64_t tbase = (uint64_t)blockIdx.x * IL_TILE;
	const uint32_t tn = (uint32_t)((total - tbase) < (uint64_t)IL_TILE ? (total - tbase) : (uint64_t)IL_TILE);
	const uint32_t q0 = tile_q[blockIdx.x], q1 = tile_q[blockIdx.x + 1];
	for (int i = tid; i < IL_TILE / 4; i += IL_THREADS) reinterpret_cast<uint32_t *>(isr)[i] = 0u;
	// the tile's holes take BWT(L)[l0, l0 + nh): staged by whole (aligned) words
	const uint64_t l0 = tbase - q0;
	const uint32_t nh = tn - (q1 - q0);
	const uint8_t *lsrc = bwtL + l0;
	const uint32_t lofs = (uint32_t)(reinterpret_cast<uintptr_t>(lsrc) & 3u);      // BWT(L) starts anywhere (a block inside a larger array)
	{
		const uint8_t *w0 = lsrc - lofs, *lo = bwtL, *hi = bwtL + m;
		const uint32_t nw = (lofs + nh + 3) >> 2;
		for (uint32_t i = tid; i < nw; i += IL_THREADS) {
			const uint8_t *wa = w0 + 4 * (uint64_t)i;
			uint32_t v;
			if (wa >= lo && wa + 4 <= hi) v = *reinterpret_cast<const uint32_t *>(wa);
			else { v = 0; for (int k = 0; k < 4; ++k) if (wa + k >= lo && wa + k < hi) v |= (uint32_t)wa[k] << (8 * k); }      // the array's first / last word
			reinterpret_cast<uint32_t *>(lseg)[i] = v;
		}
	}
	__syncthreads();
	for (uint32_t q = q0 + tid; q < q1; q += IL_THREADS) {
		uint32_t p = (uint32_t)((uint64_t)q + sr[q] - tbase);
		sym[p] = bwtR[q];
		isr[p] = 1;
	}
	__syncthreads();
	const uint32_t base = tid * IL_ITEMS;
	const uint4 fl = reinterpret_cast<const uint4 *>(isr)[tid], sv = reinterpret_cast<const uint4 *>(sym)[tid];
	const uint32_t f[4] = {fl.x, fl.y, fl.z, fl.w}, sy[4] = {sv.x, sv.y, sv.z, sv.w};
	uint32_t rcount = __popc(f[0]) + __popc(f[1]) + __popc(f[2]) + __popc(f[3]);          // flags are 0/1 bytes
	uint32_t valid = base < tn ? (tn - base < (uint32_t)IL_ITEMS ? tn - base : (uint32_t)IL_ITEMS) : 0u;
	uint32_t holes = valid - (rcount < valid ? rcount : valid);
	uint32_t inc = wave_incl_sum(holes);
	uint32_t w = tid >> 6, l = lane_id();
	if (l == 63) lds[w] = inc;
	__syncthreads();
	uint32_t pre = 0;
	for (uint32_t i = 0; i < w; ++i) pre += lds[i];
	uint32_t ex = lofs + pre + inc - holes;                             // index into lseg of this thread's first hole symbol
	uint32_t o[4] = {0u, 0u, 0u, 0u};
	#pragma unroll
	for (int i = 0; i < IL_ITEMS; ++i) {
		const uint32_t isR = (f[i >> 2] >> (8 * (i & 3))) & 0xFFu;
		uint32_t v;
		if (isR) v = (sy[i >> 2] >> (8 * (i & 3))) & 0xFFu; else { v = lseg[ex]; ++ex; }
		o[i >> 2] |= v << (8 * (i & 3));
	}
	if (valid == (uint32_t)IL_ITEMS) reinterpret_cast<uint4 *>(out + tbase)[tid] = make_uint4(o[0], o[1], o[2], o[3]);
	else for (uint32_t i = 0; i < valid; ++i) out[tbase + base + i] = (uint8_t)(o[i >> 2] >> (8 * (i & 3)));
}

void interleave(const uint32_t *d_sr, const uint8_t *d_bwtL, uint64_t m, const uint8_t *d_bwtR, uint64_t mr, uint8_t *d_out,
                uint32_t *d_tile_q, hipStream_t s)
{
	uint64_t total = m + mr;
	if (!total) return;
	uint64_t tiles = interleave_tiles(total);
	hipLaunchKernelGGL(k_tile_bounds, dim3(nblk(tiles + 1)), dim3(256), 0, s, d_sr, mr, d_tile_q, tiles);
	hipLaunchKernelGGL(k_interleave, dim3((uint32_t)tiles), dim3(IL_THREADS), 0, s, d_sr, d_bwtL, m, d_bwtR, mr, d_out, d_tile_q);
}

// ------------------------------------------------------------------------------------------------
// rank bookkeeping across a merge
// ------------------------------------------------------------------------------------------------
// pairs = (rank, pos) u64; entries with pos in [posL_begin, posR_begin) belong to L, in
// [posR_begin, posR_end) to R; others untouched.
__global__ void k_remap(const uint32_t *sr, uint64_t mr, uint64_t *pairs, uint64_t count, uint64_t pl, uint64_t pr, uint64_t pe)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= count) return;
	uint64_t r = pairs[2 * i], pos = pairs[2 * i + 1];
	if (pos >= pl && pos < pr) {
		// L entry: r + #{q : sr[q] <= r}
		uint64_t lo = 0, hi = mr;
		while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if ((uint64_t)sr[mid] <= r) lo = mid + 1; else hi = mid; }
		pairs[2 * i] = r + lo;
	} else if (pos >= pr && pos < pe) {
		pairs[2 * i] = r + (uint64_t)sr[r];
	}
}
void remap_ranks(const uint32_t *d_sr, uint64_t mr, uint64_t *d_pairs, uint64_t count, uint64_t posL_begin, uint64_t posR_begin,
                 uint64_t posR_end, hipStream_t s)
{
	if (count) hipLaunchKernelGGL(k_remap, dim3(nblk(count)), dim3(256), 0, s, d_sr, mr, d_pairs, count, posL_begin, posR_begin, posR_end);
}

__global__ void k_dense_gap(const uint32_t *sr, uint64_t mr, uint32_t *G)
{
	uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= mr) return;
	// run heads add the run length
	if (q == 0 || sr[q] != sr[q - 1]) {
		uint64_t e = q + 1;
		while (e < mr && sr[e] == sr[q]) ++e;
		G[sr[q]] = (uint32_t)(e - q);
	}
}
void dense_gap_from_sorted(const uint32_t *d_sr, uint64_t mr, uint32_t *d_G, uint64_t m, hipStream_t s)
{
	B3M_HIP(hipMemsetAsync(d_G, 0, 4 * (m + 1), s));
	if (mr) hipLaunchKernelGGL(k_dense_gap, dim3(nblk(mr)), dim3(256), 0, s, d_sr, mr, d_G);
}

// dst bits [b,e) = src bits [b,e) (other bits of dst untouched)
__global__ void k_copy_bits(const uint32_t *src, uint32_t *dst, uint64_t b, uint64_t e)
{
	uint64_t w = (b >> 5) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if ((w << 5) >= e) return;
	uint64_t lo = w << 5, hi = lo + 32;
	uint32_t mask = 0xFFFFFFFFu;
	if (lo < b) mask &= ~((1u << (b - lo)) - 1u);
	if (hi > e) mask &= ((1u << (e - lo)) - 1u);
	if (mask == 0xFFFFFFFFu) dst[w] = src[w];
	else { atomicAnd(&dst[w], ~mask); atomicOr(&dst[w], src[w] & mask); }
}
void copy_bits(const uint32_t *src, uint32_t *dst, uint64_t bit_begin, uint64_t bit_end, hipStream_t s)
{
	if (bit_end <= bit_begin) return;
	uint64_t words = ((bit_end + 31) >> 5) - (bit_begin >> 5);
	hipLaunchKernelGGL(k_copy_bits, dim3(nblk(words)), dim3(256), 0, s, src, dst, bit_begin, bit_end);
}
__global__ void k_fill(uint32_t *d, uint32_t v, uint64_t n)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) d[i] = v;
}
void fill_u32(uint32_t *d, uint32_t v, uint64_t n, hipStream_t s)
{
	if (n) hipLaunchKernelGGL(k_fill, dim3(nblk(n)), dim3(256), 0, s, d, v, n);
}

}  // namespace b3m
