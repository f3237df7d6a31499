// Microbenchmark of the partition scatter's WRITE path (no reads): S output streams per workgroup,
// chunks of 4096 8-byte records leave the CU as one contiguous run per stream.
//   mode 0: every run exactly 16 records, cursors 128-byte aligned   (whole aligned lines)
//   mode 1: run lengths vary around 16 (binomial), cursors unaligned  (what the sort does)
//   mode 2: every run exactly 16 records, cursors offset by 5 records (every line in two parts, same workgroup)
//   mode 3: mode 1 lengths, but each stream's cursor is rounded UP to a multiple of 16 before every
//           run (holes in the output; timing of "padded" layouts)
//   mode 4: mode 1, but a run's leading partial line is held back and written with the next chunk's
//           run of the same stream by the same instruction sequence (merge in registers not modelled;
//           here simply: run starts are rounded DOWN to a line, rewriting the previous tail -> full lines only)
// build: hipcc -O3 --offload-arch=gfx950 scatter_write.hip -o scatter_write ; run: ./scatter_write [S=256] [GB=24]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int THREADS = 256, MAXS = 2048;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// POL: 0 plain store, 1 sc0 sc1 (system-scope write-through), 2 nt, 3 sc1
template <int POL> __device__ __forceinline__ void st(uint64_t *p, uint64_t v)
{
	if (POL == 0) *p = v;
	else if (POL == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
	else if (POL == 2) asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
	else asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

template <int MODE, int CHUNK, int POL, bool READ>
__global__ __launch_bounds__(THREADS) void k_scatter(uint64_t *out, const uint64_t *in, uint64_t per_stream_per_wg, int S, int chunks)
{
	__shared__ uint32_t binstart[MAXS + 1];
	__shared__ uint32_t cnt[MAXS];
	__shared__ uint64_t cursor[MAXS];
	__shared__ uint16_t sdig[CHUNK];
	__shared__ uint32_t wsum[4];
	const uint32_t tid = threadIdx.x;
	// stream s of this workgroup owns out[(s * gridDim.x + blockIdx.x) * per_stream_per_wg ...)
	for (int s = tid; s < S; s += THREADS) cursor[s] = ((uint64_t)s * gridDim.x + blockIdx.x) * per_stream_per_wg + (MODE == 2 ? 5 : 0);
	__syncthreads();
	uint64_t acc = 0;
	for (int c = 0; c < chunks; ++c) {
		if (READ) { for (int p = tid; p < CHUNK; p += THREADS) acc ^= in[((uint64_t)blockIdx.x * chunks + c) * CHUNK + p]; }
		// run lengths of this chunk
		if (MODE == 0 || MODE == 2) { for (int s = tid; s < S; s += THREADS) cnt[s] = CHUNK / S; }
		else {
			for (int s = tid; s < S; s += THREADS) cnt[s] = 0;
			__syncthreads();
			for (int p = tid; p < CHUNK; p += THREADS) atomicAdd(&cnt[mix(p * 2654435761u + c * 40503u + blockIdx.x * 9176u) % (uint32_t)S], 1u);
		}
		__syncthreads();
		{
			constexpr int per = MAXS / THREADS;
			uint32_t loc[per], sum = 0;
			for (int k = 0; k < per; ++k) { int s = tid * per + k; loc[k] = s < S ? cnt[s] : 0; sum += loc[k]; }
			uint32_t inc = sum;
			for (int o = 1; o < 64; o <<= 1) { uint32_t u = __shfl_up(inc, o, 64); if ((int)(tid & 63) >= o) inc += u; }
			if ((tid & 63) == 63) wsum[tid >> 6] = inc;
			__syncthreads();
			uint32_t run = inc - sum;
			for (uint32_t i = 0; i < (tid >> 6); ++i) run += wsum[i];
			for (int k = 0; k < per; ++k) { int s = tid * per + k; if (s < S) binstart[s] = run; run += loc[k]; }
			if (tid == THREADS - 1) binstart[S] = run;
		}
		__syncthreads();
		for (int s = tid; s < S; s += THREADS) for (uint32_t p = binstart[s]; p < binstart[s + 1]; ++p) sdig[p] = (uint16_t)s;
		if (MODE == 3) { for (int s = tid; s < S; s += THREADS) cursor[s] = (cursor[s] + 15) & ~15ull; }
		__syncthreads();
		for (int p = tid; p < CHUNK; p += THREADS) {
			uint32_t s = sdig[p];
			uint64_t cur = cursor[s];
			if (MODE == 4) {
				// full lines only: the run is extended backwards to its line start (rewrites the previous tail)
				uint64_t a0 = cur & ~15ull;
				uint32_t off = p - binstart[s];
				uint32_t lead = (uint32_t)(cur - a0);
				st<POL>(&out[cur + off], ((uint64_t)p << 32) | c);
				if (off < lead) st<POL>(&out[a0 + off], ((uint64_t)p << 32) | c);      // the held-back part, rewritten
			} else {
				st<POL>(&out[cur + (p - binstart[s])], (((uint64_t)p << 32) | c) ^ acc);
			}
		}
		__syncthreads();
		for (int s = tid; s < S; s += THREADS) cursor[s] += cnt[s];
		__syncthreads();
	}
}

template <int MODE, int CHUNK, int POL, bool READ> float run(uint64_t *d, const uint64_t *in, uint64_t total_recs, int S, int wgs)
{
	int chunks = (int)(total_recs / ((uint64_t)wgs * CHUNK));
	uint64_t per = ((uint64_t)chunks * CHUNK / S) * 2 + 64;      // room for imbalance / padding
	per = (per + 15) & ~15ull;
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	hipLaunchKernelGGL((k_scatter<MODE, CHUNK, POL, READ>), dim3(wgs), dim3(THREADS), 0, 0, d, in, per, S, 2);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	hipLaunchKernelGGL((k_scatter<MODE, CHUNK, POL, READ>), dim3(wgs), dim3(THREADS), 0, 0, d, in, per, S, chunks);
	CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	double gb = (double)wgs * chunks * CHUNK * 8 / 1e9;
	printf("mode %d chunk %5d pol %d read %d S=%4d wgs=%d chunks=%d : %6.2f ms  %.1f GB written  %.2f TB/s (write only)\n", MODE, CHUNK, POL, (int)READ, S, wgs, chunks, ms, gb, gb / ms);
	return ms;
}

int main(int argc, char **argv)
{
	double GB = 24;
	uint64_t recs = (uint64_t)(GB * 1e9 / 8);
	uint64_t alloc = (uint64_t)(60e9 / 8);
	uint64_t *d, *in; CK(hipMalloc(&d, alloc * 8)); CK(hipMalloc(&in, recs * 8 + (1 << 20)));
	CK(hipMemset(d, 0, alloc * 8)); CK(hipMemset(in, 1, recs * 8 + (1 << 20)));
	const int W = 45776;
	// 1. L2 residency: same 16-record runs, smaller chunks (less data between the two touches of a line)
	run<0, 4096, 0, false>(d, in, recs, 256, W);
	run<1, 4096, 0, false>(d, in, recs, 256, W);
	run<1, 1024, 0, false>(d, in, recs, 64, W * 4);
	run<0, 1024, 0, false>(d, in, recs, 64, W * 4);
	run<1, 1024, 0, false>(d, in, recs, 64, W);
	// 2. store policies on the unaligned pattern
	run<1, 4096, 1, false>(d, in, recs, 256, W);
	run<1, 4096, 2, false>(d, in, recs, 256, W);
	run<1, 4096, 3, false>(d, in, recs, 256, W);
	run<0, 4096, 1, false>(d, in, recs, 256, W);
	// 3. with the read stream
	run<0, 4096, 0, true>(d, in, recs, 256, W);
	run<1, 4096, 0, true>(d, in, recs, 256, W);
	run<1, 4096, 0, true>(d, in, recs, 128, W);
	run<1, 4096, 0, true>(d, in, recs, 64, W);
	CK(hipFree(d)); CK(hipFree(in));
	return 0;
}
