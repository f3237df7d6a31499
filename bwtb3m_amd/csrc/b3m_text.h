// b3m_text.h -- device-side access to the packed input (the role of
// {Byte,Pac,PacTerm}InputTypes::circular_wrapper).  Symbols are decoded in registers straight
// from the packed bytes; nothing is unpacked to HBM.
#pragma once
#include "b3m_internal.h"

namespace b3m {

__device__ __forceinline__ uint32_t base_at(const TextView &T, uint64_t i)
{
	// BWA .pac: base i in byte i>>2, bit position ((~i)&3)<<1 (first base in the two MSBs)
	return (T.b[i >> 2] >> (((~i) & 3) << 1)) & 3u;
}

// symbol in the output alphabet at position i (i < n)
__device__ __forceinline__ uint32_t sym_at(const TextView &T, uint64_t i)
{
	if (T.type == B3M_INPUT_BYTESTREAM) return T.b[i];
	if (T.type == B3M_INPUT_SYMBOLS32) return reinterpret_cast<const uint32_t *>(T.b)[i];
	if (T.type == B3M_INPUT_PAC) return base_at(T, i);
	return (i == T.n - 1) ? 0u : base_at(T, i) + 1u;   // pacterm
}

__device__ __forceinline__ uint32_t be32(const uint8_t *p)   // p 4-byte aligned
{
	return __builtin_bswap32(*reinterpret_cast<const uint32_t *>(p));
}

// 64 bits = 32 consecutive 2-bit bases starting at base `pos`, first base in the MSBs.
// Requires the 12 bytes starting at the word holding `pos` to be readable (padded buffer).
__device__ __forceinline__ uint64_t window32(const TextView &T, uint64_t pos)
{
	const uint8_t *p = T.b + ((pos >> 4) << 2);
	uint64_t w0 = be32(p), w1 = be32(p + 4), w2 = be32(p + 8);
	uint32_t s = 2u * (uint32_t)(pos & 15);
	uint64_t hi = (w0 << 32) | w1;
	return s ? ((hi << s) | (w2 >> (32 - s))) : hi;
}

// Sort key of the rotation starting at i, for the symbols [depth*K, (depth+1)*K) of it.
// Order of keys == lexicographic order of those K symbols (with the pacterm terminator
// smaller than every base, encoded by the tag: see b3m_internal.h).
__device__ inline uint64_t key_at(const TextView &T, uint64_t i, uint64_t depth)
{
	if (T.type == B3M_INPUT_PACTERM) {
		constexpr int K = KSYM_PACTERM;
		uint64_t pos = i + depth * K;
		if (pos + K <= T.nbases)        // window holds bases only
			return ((window32(T, pos) >> (64 - 2 * K)) << TAGBITS_PACTERM) | TAG_NORMAL;
		if (pos > T.nbases) return 0;   // already past the terminator: cannot be tied
		uint64_t k = 0; uint32_t t = 0;
		for (; t < (uint32_t)K; ++t) {
			uint64_t p = pos + t;
			if (p == T.nbases) break;    // terminator at offset t
			k = (k << 2) | base_at(T, p);
		}
		k <<= 2 * (K - t);
		return (k << TAGBITS_PACTERM) | t;
	}
	if (T.type == B3M_INPUT_PAC) {
		uint64_t pos = (i + depth * KSYM_PAC) % T.n;
		if (pos + KSYM_PAC <= T.n) return window32(T, pos);
		uint64_t k = 0;
		for (int t = 0; t < KSYM_PAC; ++t) { k = (k << 2) | base_at(T, pos); if (++pos == T.n) pos = 0; }
		return k;
	}
	if (T.type == B3M_INPUT_SYMBOLS32) {      // three 21-bit symbols, circular
		const uint32_t *w = reinterpret_cast<const uint32_t *>(T.b);
		uint64_t pos = (i + depth * KSYM_WIDE) % T.n, k = 0;
		for (int t = 0; t < KSYM_WIDE; ++t) { k = (k << WIDE_BITS) | w[pos]; if (++pos == T.n) pos = 0; }
		return k;
	}
	// bytestream: 8 bytes, circular
	uint64_t pos = (i + depth * KSYM_BYTES) % T.n;
	uint64_t k = 0;
	if (pos + 8 <= T.n) {
		for (int t = 0; t < 8; ++t) k = (k << 8) | T.b[pos + t];
	} else {
		for (int t = 0; t < 8; ++t) { k = (k << 8) | T.b[pos]; if (++pos == T.n) pos = 0; }
	}
	return k;
}

// Key of the tie rounds (k_blocksort.hip): the K symbols [off, off + K) of the rotation starting at i, for a list whose groups agree on their
// first `off` symbols IN THE READING OF THE RECORD SORT (k_msd.hip: the terminator and what lies behind it read as base 0).  pacterm: a suffix whose
// terminator lies in front of the window is smaller than every member of its group that is longer (what it agreed with was padding) and
// shorter ones come first, so its key is its length (below 64); every other key is key_at's plus 64.  A group never holds two suffixes with the
// terminator in the same place, so such members leave the list in the round that sees them.
__device__ inline uint64_t key_at_off(const TextView &T, uint64_t i, uint64_t off)
{
	if (T.type == B3M_INPUT_PACTERM) {
		constexpr int K = KSYM_PACTERM;
		const uint64_t pos = i + off;
		if (pos + K <= T.nbases) return (((window32(T, pos) >> (64 - 2 * K)) << TAGBITS_PACTERM) | TAG_NORMAL) + 64u;
		if (pos > T.nbases) { const uint64_t t = T.nbases - i; return t < 63u ? t : 63u; }
		uint64_t k = 0; uint32_t t = 0;
		for (; t < (uint32_t)K; ++t) {
			uint64_t p = pos + t;
			if (p == T.nbases) break;
			k = (k << 2) | base_at(T, p);
		}
		k <<= 2 * (K - t);
		return ((k << TAGBITS_PACTERM) | t) + 64u;
	}
	uint64_t pos = (i + off) % T.n;
	if (T.type == B3M_INPUT_SYMBOLS32) {
		const uint32_t *w = reinterpret_cast<const uint32_t *>(T.b);
		uint64_t k = 0;
		for (int t = 0; t < KSYM_WIDE; ++t) { k = (k << WIDE_BITS) | w[pos]; if (++pos == T.n) pos = 0; }
		return k;
	}
	const int K = T.type == B3M_INPUT_PAC ? KSYM_PAC : KSYM_BYTES;
	if (T.type == B3M_INPUT_PAC) {
		if (pos + KSYM_PAC <= T.n) return window32(T, pos);
		uint64_t k = 0;
		for (int t = 0; t < K; ++t) { k = (k << 2) | base_at(T, pos); if (++pos == T.n) pos = 0; }
		return k;
	}
	uint64_t k = 0;
	for (int t = 0; t < K; ++t) { k = (k << 8) | T.b[pos]; if (++pos == T.n) pos = 0; }
	return k;
}
inline int key_off_bits(int type) { return type == B3M_INPUT_PACTERM ? 2 * KSYM_PACTERM + TAGBITS_PACTERM + 1 : type == B3M_INPUT_SYMBOLS32 ? KSYM_WIDE * WIDE_BITS : 64; }

// compare rotations S_x and S_y (x != y): -1, 0 (equal over >= n symbols), +1
__device__ inline int compare_rotations(const TextView &T, uint64_t x, uint64_t y)
{
	int K = T.type == B3M_INPUT_BYTESTREAM ? KSYM_BYTES : T.type == B3M_INPUT_PAC ? KSYM_PAC : T.type == B3M_INPUT_SYMBOLS32 ? KSYM_WIDE : KSYM_PACTERM;
	for (uint64_t d = 0; d * K < T.n; ++d) {
		uint64_t kx = key_at(T, x, d), ky = key_at(T, y, d);
		if (kx != ky) return kx < ky ? -1 : 1;
	}
	return 0;
}

// ---- (rank,pos) samples (SampleOut, b3m_internal.h) ------------------------------------------------
// is the position with these low 32 bits a multiple of the rate?  (sampling off: pmask is all ones, only 0 passes and sample_put drops it)
__device__ __forceinline__ bool sample_hit(const SampleOut &S, uint32_t lo32)
{
	bool h = (lo32 & S.pmask) == 0u;
	// rates that are not powers of two: the exact test only for the few positions the mask lets through, BEHIND a branch (the empty asm keeps the compiler from
	// computing it for every record and selecting afterwards: a shift, a 32-bit multiply -- quarter rate -- and three selects per record of k_finish_v3, whose
	// vector ALUs are ~85 % busy: round 5)
	if (h && !S.pow2) { uint32_t q = (lo32 >> S.tz) * S.inv; asm volatile("" : "+v"(q)); h = q <= S.lim; }
	return h;
}
// (only for positions sample_hit accepted: the quotient by the odd part of the rate is then the product with its inverse modulo
// 2^32 -- no division in the kernels; rates that are not powers of two come with 32-bit positions only, sample_in_kernel_ok)
__device__ __forceinline__ void sample_put(const SampleOut &S, uint64_t rank, uint64_t pos)
{
	if (!S.pairs) return;
	const uint64_t k = (S.pow2 ? pos >> S.tz : (uint64_t)(((uint32_t)pos >> S.tz) * S.inv)) - S.first;
	if (k < S.cap) { S.pairs[2 * k] = rank + S.rank_off; S.pairs[2 * k + 1] = pos; }
}

// ---- wave helpers (64 lanes) -----------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Inclusive scans over the 64 lanes through DPP (cross-lane operands of the VALU itself): row_shr 1/2/4/8 inside each row of 16 lanes, then
// row_bcast:15 (lane 15 of rows 0 and 2 into rows 1 and 3) and row_bcast:31 (lane 31 into rows 2 and 3): six vector instructions, no LDS
// pipe (the `__shfl_up` form is six ds_bpermute round trips; every phase that scans -- level 0, level 1, the finish, the scan kernels -- sits
// between two barriers and waits for exactly that chain).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false); }     // lanes without a source (or outside the row mask) read 0
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v)
{
	v += dpp_or0<0x111, 0xf>(v); v += dpp_or0<0x112, 0xf>(v); v += dpp_or0<0x114, 0xf>(v); v += dpp_or0<0x118, 0xf>(v);
	v += dpp_or0<0x142, 0xa>(v); v += dpp_or0<0x143, 0xc>(v);
	return v;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v)      // (unsigned: 0 is the neutral element)
{
	uint32_t u;
	u = dpp_or0<0x111, 0xf>(v); v = v > u ? v : u; u = dpp_or0<0x112, 0xf>(v); v = v > u ? v : u; u = dpp_or0<0x114, 0xf>(v); v = v > u ? v : u; u = dpp_or0<0x118, 0xf>(v); v = v > u ? v : u;
	u = dpp_or0<0x142, 0xa>(v); v = v > u ? v : u; u = dpp_or0<0x143, 0xc>(v); v = v > u ? v : u;
	return v;
}

}  // namespace b3m
