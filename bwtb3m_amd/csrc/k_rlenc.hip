// k_rlenc.hip -- the RL .bwt container written from a BWT that is still in HBM (role of
// libmaus2::huffman::RLEncoderStd, usage /root/reference/src/lcpbit.cpp:3677-3681).  The host encoder (host/formats.cpp)
// spends its time finding runs and writing two Huffman code streams per block bit by bit: 5.5 s at 3 Gbase.  Here the
// device finds the runs, counts every block's symbol / run-length frequencies, and packs both code streams of every block at
// their final byte offsets; the host only builds the per-block code tables and headers from the frequencies (the SAME code as
// the host encoder: the files are byte-identical) and writes the file.
//   1. k_rl_count  : run heads per tile of 4096 symbols              (one read of the BWT)
//   2. k_rl_runs   : start position and symbol of every run           (second read; 5 B per run written)
//   3. k_rl_hist   : per block of `rpb` runs: symbol counts, length counts (1..255, 256 = escape), Elias bits of the escaped lengths, symbols
//      host        : code tables + header per block (rl_block_header), sizes of the two streams from the counts, byte offsets
//   4. k_rl_pack   : one workgroup per block: bit offsets by a scan over its runs' code lengths, codes OR-ed into the zeroed output
// Bit order MSB first (a stream is bytes; the 32-bit words the kernel ORs are byte-swapped).
#include "b3m_internal.h"
#include "host/formats.h"
#include "host/parfile.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <functional>
#include <memory>
#include <thread>

namespace b3m {

static inline uint32_t nblk(uint64_t n, uint32_t t = 256) { return (uint32_t)((n + t - 1) / t); }
namespace {
struct Buf {
	void *p = nullptr;
	explicit Buf(uint64_t bytes) { B3M_HIP(hipMalloc(&p, bytes ? bytes : 16)); }
	~Buf() { if (p) (void)hipFree(p); }
	template <class X> X *as() { return (X *)p; }
	Buf(const Buf &) = delete; Buf &operator=(const Buf &) = delete;
};
}

constexpr int RL_TILE = 4096;            // symbols per workgroup in the run search (256 threads x 16)
constexpr int RL_HIST = 516;             // per block: [0,256) symbol counts, [256,513) length counts (index 256 + min(len,256)), [513] Elias bits, [514] symbols lo, [515] symbols hi

__device__ __forceinline__ uint32_t elias_bits(uint64_t x)
{
	const uint64_t y = x + 1;
	const int N = 63 - __clzll((long long)y);
	const int L = 31 - __clz(N + 1);
	return (uint32_t)(2 * L + 1 + N);
}

__device__ __forceinline__ uint32_t wave_incl(uint32_t v)
{
	const uint32_t lane = threadIdx.x & 63u;
	#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(v, o, 64); if (lane >= (uint32_t)o) v += t; }
	return v;
}

// heads of tile t: positions i with i == 0 or bwt[i] != bwt[i-1]
__global__ __launch_bounds__(256) void k_rl_count(const uint8_t *bwt, uint64_t n, uint32_t *tile_heads)
{
	__shared__ uint32_t ws[4];
	const uint64_t base = (uint64_t)blockIdx.x * RL_TILE + (uint64_t)threadIdx.x * 16;
	uint32_t c = 0;
	if (base < n) {
		uint8_t prev = base ? bwt[base - 1] : (uint8_t)~bwt[0];
		#pragma unroll
		for (int j = 0; j < 16; ++j) { if (base + j < n) { const uint8_t v = bwt[base + j]; c += v != prev; prev = v; } }
	}
	#pragma unroll
	for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
	if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) tile_heads[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// run r: rstart[r] = position of its first symbol, rsym[r] = the symbol; rstart[R] = n
__global__ __launch_bounds__(256) void k_rl_runs(const uint8_t *bwt, uint64_t n, const uint32_t *tile_first, uint64_t R, uint32_t *rstart, uint8_t *rsym)
{
	__shared__ uint32_t ws[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint64_t base = (uint64_t)blockIdx.x * RL_TILE + (uint64_t)tid * 16;
	uint8_t v[16]; uint32_t heads = 0, c = 0;
	if (base < n) {
		uint8_t prev = base ? bwt[base - 1] : (uint8_t)~bwt[0];
		#pragma unroll
		for (int j = 0; j < 16; ++j) { v[j] = base + j < n ? bwt[base + j] : prev; if (base + j < n && v[j] != prev) { heads |= 1u << j; ++c; } prev = v[j]; }
	}
	const uint32_t inc = wave_incl(c);
	if (lane == 63) ws[wave] = inc;
	__syncthreads();
	uint64_t r = (uint64_t)tile_first[blockIdx.x] + inc - c;
	for (uint32_t i = 0; i < wave; ++i) r += ws[i];
	#pragma unroll
	for (int j = 0; j < 16; ++j) if (heads & (1u << j)) { rstart[r] = (uint32_t)(base + j); rsym[r] = v[j]; ++r; }
	if (blockIdx.x == 0 && tid == 0) rstart[R] = (uint32_t)n;       // (n < 2^32)
}

__global__ __launch_bounds__(256) void k_rl_hist(const uint32_t *rstart, const uint8_t *rsym, uint64_t R, uint32_t rpb, uint32_t *hist)
{
	__shared__ uint32_t h[RL_HIST];
	const uint64_t r0 = (uint64_t)blockIdx.x * rpb, r1 = r0 + rpb < R ? r0 + rpb : R;
	for (uint32_t i = threadIdx.x; i < (uint32_t)RL_HIST; i += 256) h[i] = 0;
	__syncthreads();
	for (uint64_t r = r0 + threadIdx.x; r < r1; r += 256) {
		const uint32_t len = rstart[r + 1] - rstart[r];
		atomicAdd(&h[rsym[r]], 1u);
		if (len >= 256u) { atomicAdd(&h[256 + 256], 1u); atomicAdd(&h[513], elias_bits(len)); } else atomicAdd(&h[256 + len], 1u);
	}
	__syncthreads();
	if (threadIdx.x == 0) { const uint64_t syms = (uint64_t)rstart[r1] - rstart[r0]; h[514] = (uint32_t)syms; h[515] = (uint32_t)(syms >> 32); }
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < (uint32_t)RL_HIST; i += 256) hist[(uint64_t)blockIdx.x * RL_HIST + i] = h[i];
}

// appends `bits` (<= 32) bits of v to a stream that starts at bit `bitpos` of `out` (bytes, MSB first); words are OR-ed in
struct BitSink {
	uint32_t *out; uint64_t w; uint64_t acc; int fill;
	__device__ void open(uint8_t *base, uint64_t bitpos) { out = reinterpret_cast<uint32_t *>(base); w = bitpos >> 5; fill = (int)(bitpos & 31); acc = 0; }
	__device__ void put(uint64_t v, int bits)
	{
		if (bits == 0) return;
		acc = (acc << bits) | (v & (bits >= 64 ? ~0ull : ((1ull << bits) - 1)));
		fill += bits;
		while (fill >= 32) {
			const uint32_t word = (uint32_t)(acc >> (fill - 32));
			if (word) atomicOr(&out[w], __builtin_bswap32(word));
			++w; fill -= 32;
		}
	}
	__device__ void close() { if (fill > 0) { const uint32_t word = (uint32_t)(acc << (32 - fill)); if (word) atomicOr(&out[w], __builtin_bswap32(word)); } }
};

// tables: per block 513 entries (code << 8 | length): [0,256) symbols, [256,513) lengths (256 + min(len,256))
__global__ __launch_bounds__(256) void k_rl_pack(const uint32_t *rstart, const uint8_t *rsym, uint64_t R, uint32_t rpb, const uint64_t *tables,
                                                 const uint64_t *off_sym, const uint64_t *off_len, uint8_t *out)
{
	__shared__ uint64_t tab[513];
	__shared__ uint32_t ws1[4], ws2[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint64_t r0 = (uint64_t)blockIdx.x * rpb, r1 = r0 + rpb < R ? r0 + rpb : R;
	for (uint32_t i = tid; i < 513u; i += 256) tab[i] = tables[(uint64_t)blockIdx.x * 513 + i];
	__syncthreads();
	const uint32_t per = (rpb + 255) / 256;                   // consecutive runs per thread
	const uint64_t a = r0 + (uint64_t)tid * per, b = a + per < r1 ? a + per : r1;
	uint32_t b1 = 0, b2 = 0;
	for (uint64_t r = a; r < b; ++r) {
		const uint32_t len = rstart[r + 1] - rstart[r];
		b1 += (uint32_t)(tab[rsym[r]] & 255u);
		if (len >= 256u) b2 += (uint32_t)(tab[512] & 255u) + elias_bits(len); else b2 += (uint32_t)(tab[256 + len] & 255u);
	}
	const uint32_t i1 = wave_incl(b1), i2 = wave_incl(b2);
	if (lane == 63) { ws1[wave] = i1; ws2[wave] = i2; }
	__syncthreads();
	uint64_t p1 = i1 - b1, p2 = i2 - b2;
	for (uint32_t i = 0; i < wave; ++i) { p1 += ws1[i]; p2 += ws2[i]; }
	BitSink s1, s2;
	s1.open(out, off_sym[blockIdx.x] * 8 + p1);
	s2.open(out, off_len[blockIdx.x] * 8 + p2);
	for (uint64_t r = a; r < b; ++r) {
		const uint32_t len = rstart[r + 1] - rstart[r];
		const uint64_t e = tab[rsym[r]];
		s1.put(e >> 8, (int)(e & 255u));
		if (len >= 256u) {
			const uint64_t c = tab[512];
			s2.put(c >> 8, (int)(c & 255u));
			const uint64_t y = (uint64_t)len + 1;
			const int N = 63 - __clzll((long long)y), L = 31 - __clz(N + 1);
			s2.put((uint64_t)(N + 1), 2 * L + 1);           // L zeros, then N + 1 in L + 1 bits
			s2.put(y & ((1ull << N) - 1), N);
		} else { const uint64_t c = tab[256 + len]; s2.put(c >> 8, (int)(c & 255u)); }
	}
	s1.close(); s2.close();
}

// block b's header bytes (code tables, built on the host) dropped at the block's start in the output image: one workgroup per block
__global__ __launch_bounds__(64) void k_rl_headers(const uint8_t *blob, const uint64_t *src_off, const uint64_t *dst_off, uint8_t *out)
{
	const uint64_t a = src_off[blockIdx.x], e = src_off[blockIdx.x + 1], d = dst_off[blockIdx.x];
	for (uint64_t i = a + threadIdx.x; i < e; i += 64) out[d + (i - a)] = blob[i];
}

// d_symbols: n bytes in HBM.  Writes the same file as rl_encode_file.  Returns false if this input is not for the device encoder (a code
// longer than 32 bits, a stretch of 2^31 symbols with fewer runs than one block holds): the caller uses the host encoder then.
// The BWT is taken in SEGMENTS of at most 2^31 symbols (run starts are 32-bit offsets into the segment; B3M_RL_SEG: test switch): a segment
// begins at a run head, encodes the whole blocks of runs_per_block runs that lie inside it -- never its last run, which may go on behind the
// segment's end -- and the next segment begins where its last block ended.  The blocks of the file are the same ones, whatever the segments.
// h_symbols != null: the BWT lies in host memory (host tier: it never was in HBM as a whole) and every segment is brought to a staging buffer of seg_max bytes first
// The encoder proper hands every segment's image (in HBM, freed when the sink returns unless the sink takes the pointer) and its blocks (positions relative to the image)
// to a sink: the file writer below puts it at the next file offset; the encoder of ONE PART of a file (several devices, each with a range of whole blocks) keeps it.
using RLSink = std::function<void(Buf &image, uint64_t bytes, std::vector<RLIndexEntry> &blocks)>;
static bool rl_encode_stream(const uint8_t *d_symbols, const uint8_t *h_symbols, uint64_t n, uint64_t runs_per_block, unsigned numthreads, const int64_t *symmap,
                             void *hip_stream, uint64_t seg_limit, const RLSink &sink);
static bool rl_encode_impl(const std::string &fn, const uint8_t *d_symbols, const uint8_t *h_symbols, uint64_t n, uint64_t runs_per_block, unsigned numthreads, const int64_t *symmap,
                           void *hip_stream, uint64_t seg_limit)
{
	int dev = 0;
	B3M_HIP(hipGetDevice(&dev));
	std::unique_ptr<ParFile> pf;
	std::vector<RLIndexEntry> index;
	uint64_t fpos = 8;                                    // file offset of the next block (8 bytes of n in front)
	const bool ok = rl_encode_stream(d_symbols, h_symbols, n, runs_per_block, numthreads, symmap, hip_stream, seg_limit, [&](Buf &image, uint64_t bytes, std::vector<RLIndexEntry> &blocks) {
		if (!pf) pf.reset(new ParFile(fn));
		pf->put_device(fpos, image.p, bytes, dev, std::min(numthreads ? numthreads : 1u, 8u));
		for (const RLIndexEntry &e : blocks) index.push_back({fpos + e.pos, e.runs, e.syms});
		fpos += bytes;
	});
	if (!ok) return false;
	std::vector<uint8_t> head, tail;
	rl_container_ends(n, fpos - 8, index, head, tail);
	pf->put(0, head.data(), head.size());
	pf->put(fpos, tail.data(), tail.size());
	pf->close();
	return true;
}
bool rl_encode_device(const std::string &fn, const uint8_t *d_symbols, uint64_t n, uint64_t runs_per_block, unsigned numthreads, const int64_t *symmap, void *hip_stream)
{
	return rl_encode_impl(fn, d_symbols, nullptr, n, runs_per_block, numthreads, symmap, hip_stream, 0);
}
bool rl_encode_device_from_host(const std::string &fn, const uint8_t *h_symbols, uint64_t n, uint64_t runs_per_block, unsigned numthreads, const int64_t *symmap, void *hip_stream,
                                uint64_t seg_limit)
{
	return rl_encode_impl(fn, nullptr, h_symbols, n, runs_per_block, numthreads, symmap, hip_stream, seg_limit);
}
static bool rl_encode_stream(const uint8_t *d_symbols, const uint8_t *h_symbols, uint64_t n, uint64_t runs_per_block, unsigned numthreads, const int64_t *symmap,
                             void *hip_stream, uint64_t seg_limit, const RLSink &sink)
{
	hipStream_t s = (hipStream_t)hip_stream;
	if (!runs_per_block) runs_per_block = 64 * 1024;
	if (!numthreads) numthreads = 1;
	if (n == 0 || runs_per_block > (1u << 24)) return false;
	const uint32_t rpb = (uint32_t)runs_per_block;
	uint64_t seg_max = 1ull << 31;
	if (seg_limit >= 1024 && seg_limit < seg_max) seg_max = seg_limit;
	if (sw().RL_SEG.set) { const uint64_t v = (uint64_t)sw().RL_SEG.v; if (v >= 1024 && v < seg_max) seg_max = v; }
	std::unique_ptr<Buf> stage;                           // host source: the segment's symbols in HBM
	if (h_symbols) stage.reset(new Buf(std::min(seg_max, n) + 16));
	for (uint64_t a = 0; a < n;) {
		const uint64_t len = std::min<uint64_t>(seg_max, n - a);
		const bool last = a + len == n;
		const uint8_t *sym = d_symbols ? d_symbols + a : stage->as<uint8_t>();
		if (h_symbols) B3M_HIP(hipMemcpyAsync(stage->p, h_symbols + a, len, hipMemcpyHostToDevice, s));
		const uint64_t ntiles = (len + RL_TILE - 1) / RL_TILE;
		Buf tiles(4 * (ntiles + 1)), scan_ws(4 * scan_ws_elems(ntiles + 1));
		hipLaunchKernelGGL(k_rl_count, dim3((uint32_t)ntiles), dim3(256), 0, s, sym, len, tiles.as<uint32_t>());
		B3M_HIP(hipMemsetAsync(tiles.as<uint32_t>() + ntiles, 0, 4, s));
		exclusive_sum_u32(tiles.as<uint32_t>(), ntiles + 1, scan_ws.as<uint32_t>(), s);
		uint32_t R32 = 0;
		B3M_HIP(hipMemcpyAsync(&R32, tiles.as<uint32_t>() + ntiles, 4, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipStreamSynchronize(s));
		const uint64_t Rall = R32;                          // run heads inside the segment
		B3M_REQUIRE(Rall >= 1, "internal: no runs found");
		const uint64_t R = last ? Rall : ((Rall - 1) / rpb) * rpb;      // runs this segment encodes: whole blocks, not its last run
		if (R == 0) return false;                           // (2^31 symbols in fewer runs than a block holds: the host encoder's case)
		const uint64_t nblocks = (R + rpb - 1) / rpb;
		Buf rstart(4 * (Rall + 2)), rsym(Rall + 16), hist(4ull * RL_HIST * nblocks);
		hipLaunchKernelGGL(k_rl_runs, dim3((uint32_t)ntiles), dim3(256), 0, s, sym, len, tiles.as<uint32_t>(), Rall, rstart.as<uint32_t>(), rsym.as<uint8_t>());
		hipLaunchKernelGGL(k_rl_hist, dim3((uint32_t)nblocks), dim3(256), 0, s, rstart.as<uint32_t>(), rsym.as<uint8_t>(), R, rpb, hist.as<uint32_t>());
		std::vector<uint32_t> h((size_t)RL_HIST * nblocks);
		uint32_t used32 = 0;                                // symbols the encoded runs cover = start of run R
		B3M_HIP(hipMemcpyAsync(h.data(), hist.p, 4ull * RL_HIST * nblocks, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipMemcpyAsync(&used32, rstart.as<uint32_t>() + R, 4, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipStreamSynchronize(s));
		const uint64_t used = last ? len : used32;
		B3M_REQUIRE(used >= 1 && used <= len, "internal: segment of the run-length encoder out of range");
		// ---- host: tables, headers, sizes ----
		std::vector<uint64_t> tables(513 * nblocks), off_sym(nblocks), off_len(nblocks), bytes1(nblocks), bytes2(nblocks);
		std::vector<std::vector<uint8_t>> headers(nblocks);
		std::atomic<uint64_t> next(0); std::atomic<bool> too_long(false);
		auto work = [&]() {
			RLBlockTables t;
			for (;;) {
				const uint64_t b = next.fetch_add(1);
				if (b >= nblocks) return;
				const uint32_t *hb = h.data() + (size_t)RL_HIST * b;
				const uint64_t nr = std::min<uint64_t>(rpb, R - b * rpb);
				rl_block_header(hb, hb + 256, nr, symmap, headers[b], t);
				uint64_t s1 = 0, s2 = hb[513];
				uint64_t *tb = tables.data() + 513 * b;
				for (int c = 0; c < 256; ++c) { tb[c] = (t.scode[c] << 8) | t.slen[c]; s1 += (uint64_t)hb[c] * t.slen[c]; if (t.slen[c] > 32) too_long = true; }
				for (int l = 0; l <= 256; ++l) { tb[256 + l] = (t.ccode[l] << 8) | t.clen[l]; s2 += (uint64_t)hb[256 + l] * t.clen[l]; if (t.clen[l] > 32) too_long = true; }
				bytes1[b] = (s1 + 7) / 8; bytes2[b] = (s2 + 7) / 8;
			}
		};
		{
			std::vector<std::thread> th;
			const unsigned T = (unsigned)std::min<uint64_t>(numthreads, nblocks);
			for (unsigned i = 1; i < T; ++i) th.emplace_back(work);
			work();
			for (auto &x : th) x.join();
		}
		if (too_long) return false;
		uint64_t pos = 0;                                   // offsets inside this segment's image
		std::vector<uint64_t> hdst(nblocks);
		std::vector<RLIndexEntry> index;                    // this segment's blocks, positions relative to its image
		for (uint64_t b = 0; b < nblocks; ++b) {
			const uint32_t *hb = h.data() + (size_t)RL_HIST * b;
			index.push_back({pos, std::min<uint64_t>(rpb, R - b * rpb), (uint64_t)hb[514] | ((uint64_t)hb[515] << 32)});
			hdst[b] = pos;
			off_sym[b] = pos + headers[b].size();
			off_len[b] = off_sym[b] + bytes1[b];
			pos = off_len[b] + bytes2[b];
		}
		const uint64_t total = pos;
		Buf d_out(((total + 3) & ~3ull) + 16), d_tab(8 * tables.size()), d_o1(8 * nblocks), d_o2(8 * nblocks);
		B3M_HIP(hipMemsetAsync(d_out.p, 0, ((total + 3) & ~3ull) + 16, s));
		B3M_HIP(hipMemcpyAsync(d_tab.p, tables.data(), 8 * tables.size(), hipMemcpyHostToDevice, s));
		B3M_HIP(hipMemcpyAsync(d_o1.p, off_sym.data(), 8 * nblocks, hipMemcpyHostToDevice, s));
		B3M_HIP(hipMemcpyAsync(d_o2.p, off_len.data(), 8 * nblocks, hipMemcpyHostToDevice, s));
		hipLaunchKernelGGL(k_rl_pack, dim3((uint32_t)nblocks), dim3(256), 0, s, rstart.as<uint32_t>(), rsym.as<uint8_t>(), R, rpb, d_tab.as<uint64_t>(),
		                   d_o1.as<uint64_t>(), d_o2.as<uint64_t>(), d_out.as<uint8_t>());
		// the headers join the code streams in the device image (after the pack kernel: its word-wise ORs and these byte stores may share a word),
		// and the image leaves for the file in pieces through pinned staging buffers, written at their offsets by a few threads
		std::vector<uint8_t> blob; std::vector<uint64_t> hsrc(nblocks + 1);
		for (uint64_t b = 0; b < nblocks; ++b) { hsrc[b] = blob.size(); blob.insert(blob.end(), headers[b].begin(), headers[b].end()); }
		hsrc[nblocks] = blob.size();
		Buf d_blob(blob.size() + 16), d_hs(8 * (nblocks + 1)), d_hd(8 * nblocks);
		B3M_HIP(hipMemcpyAsync(d_blob.p, blob.data(), blob.size(), hipMemcpyHostToDevice, s));
		B3M_HIP(hipMemcpyAsync(d_hs.p, hsrc.data(), 8 * (nblocks + 1), hipMemcpyHostToDevice, s));
		B3M_HIP(hipMemcpyAsync(d_hd.p, hdst.data(), 8 * nblocks, hipMemcpyHostToDevice, s));
		hipLaunchKernelGGL(k_rl_headers, dim3((uint32_t)nblocks), dim3(64), 0, s, d_blob.as<uint8_t>(), d_hs.as<uint64_t>(), d_hd.as<uint64_t>(), d_out.as<uint8_t>());
		B3M_HIP(hipStreamSynchronize(s));
		B3M_HIP(hipGetLastError());
		sink(d_out, total, index);
		a += used;
	}
	return true;
}

// ---- one file from SEVERAL devices (b3m_compute_bwt with num_gpus > 1: every device codes the blocks that begin in its slice of the BWT) ----
// A block is rpb runs, counted from the start of the whole BWT, so a device must know how many run heads lie in front of its slice and where the first block of its
// own begins.  rl_slice_scan: heads per tile of the slice, scanned (position 0 counts as a head: the caller takes it back when the slice continues its neighbour's
// last run), with the slice's first and last symbol; rl_head_position: the position of the slice's ordinal-th head.  Current device.
void rl_slice_scan(const uint8_t *d_symbols, uint64_t len, void *hip_stream, std::vector<uint32_t> &tile_first, uint8_t *first_sym, uint8_t *last_sym)
{
	hipStream_t s = (hipStream_t)hip_stream;
	tile_first.assign(1, 0u);
	*first_sym = *last_sym = 0;
	if (!len) return;
	B3M_REQUIRE(len < (1ull << 32), "internal: a slice of 2^32 symbols and more in the run-length encoder");
	const uint64_t ntiles = (len + RL_TILE - 1) / RL_TILE;
	Buf tiles(4 * (ntiles + 1)), scan_ws(4 * scan_ws_elems(ntiles + 1));
	hipLaunchKernelGGL(k_rl_count, dim3((uint32_t)ntiles), dim3(256), 0, s, d_symbols, len, tiles.as<uint32_t>());
	B3M_HIP(hipMemsetAsync(tiles.as<uint32_t>() + ntiles, 0, 4, s));
	exclusive_sum_u32(tiles.as<uint32_t>(), ntiles + 1, scan_ws.as<uint32_t>(), s);
	tile_first.resize(ntiles + 1);
	B3M_HIP(hipMemcpyAsync(tile_first.data(), tiles.p, 4 * (ntiles + 1), hipMemcpyDeviceToHost, s));
	B3M_HIP(hipMemcpyAsync(first_sym, d_symbols, 1, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipMemcpyAsync(last_sym, d_symbols + len - 1, 1, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
}
uint64_t rl_head_position(const uint8_t *d_symbols, uint64_t len, const std::vector<uint32_t> &tile_first, uint64_t ordinal, void *hip_stream)
{
	hipStream_t s = (hipStream_t)hip_stream;
	const uint64_t ntiles = tile_first.size() - 1;
	B3M_REQUIRE(ntiles >= 1 && ordinal < tile_first[ntiles], "internal: run head out of range");
	const uint64_t t = (uint64_t)(std::upper_bound(tile_first.begin(), tile_first.end(), (uint32_t)ordinal) - tile_first.begin()) - 1;      // last tile whose first head is <= ordinal
	const uint64_t base = t * RL_TILE, cnt = std::min<uint64_t>(RL_TILE, len - base);
	std::vector<uint8_t> sym(cnt + 1);
	B3M_HIP(hipMemcpyAsync(sym.data() + (base ? 0 : 1), d_symbols + (base ? base - 1 : 0), cnt + (base ? 1 : 0), hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	if (!base) sym[0] = (uint8_t)~sym[1];                  // (position 0 is a head: the kernel's convention)
	uint64_t want = ordinal - tile_first[t];
	for (uint64_t i = 0; i < cnt; ++i) if (sym[i + 1] != sym[i]) { if (!want) return base + i; --want; }
	throw Error("internal: run head not found in its tile");
}
RLPart::~RLPart() { if (!images.empty()) { (void)hipSetDevice(dev); for (auto &im : images) if (im.first) (void)hipFree(im.first); } }
// every run of d_symbols[0, len) -- a range that begins at a block's first run and ends where another block begins (or at the end of the BWT) -- coded into images that stay
// in HBM (current device); false: not an input for the device encoder (as rl_encode_device)
bool rl_encode_part(const uint8_t *d_symbols, uint64_t len, uint64_t runs_per_block, unsigned numthreads, const int64_t *symmap, void *hip_stream, RLPart &out)
{
	B3M_HIP(hipGetDevice(&out.dev));
	out.bytes = 0; out.index.clear();
	return rl_encode_stream(d_symbols, nullptr, len, runs_per_block, numthreads, symmap, hip_stream, 0, [&](Buf &image, uint64_t bytes, std::vector<RLIndexEntry> &blocks) {
		out.images.emplace_back(image.p, bytes);
		image.p = nullptr;                                  // (the part owns it now)
		for (const RLIndexEntry &e : blocks) out.index.push_back({out.bytes + e.pos, e.runs, e.syms});
		out.bytes += bytes;
	});
}

}  // namespace b3m

extern "C" int b3m_rl_write_device(const char *fn, const uint8_t *d_symbols, uint64_t n, uint64_t runs_per_block, void *stream)
{
	try {
		b3m::switches_load();
		int cnt = 0;
		if (hipGetDeviceCount(&cnt) != hipSuccess || cnt == 0) throw b3m::Error("no HIP device: this library has no CPU fallback");
		if (!fn || !d_symbols) throw b3m::Error("b3m_rl_write_device: null argument");
		if (!b3m::rl_encode_device(fn, d_symbols, n, runs_per_block, std::max(1u, std::thread::hardware_concurrency()), nullptr, stream))
			throw b3m::Error("b3m_rl_write_device: this input needs the host encoder (b3m_rl_write): empty, 2^32 symbols or more, or a code longer than 32 bits");
		return 0;
	} catch (const std::exception &e) { b3m::set_error(e.what()); return 1; }
}
