// Microbenchmark behind the gap walk's "48 G steps/s" (VERDICT r2 item 9): how many INDEPENDENT random line reads per second does the
// chip deliver, as a function of the table size (125 MB / 500 MB / 2 GB), the line size (64 B = one rank line of k_gap_walk4, 128 B),
// the number of lanes that share a line (1: every lane its own line; 4: a quad reads one 64-byte line as 4 x 16 bytes, the form
// k_gap_walk4 uses; 8: 8 x 16 bytes = a 128-byte line) and the number of requests each lane keeps in flight (1, 2, 4, 8 independent
// chains).  Every chain is a dependent sequence: the next line index is a hash of the data just read, as in an LF walk.
// build: hipcc -O3 --offload-arch=gfx950 random_lines.hip -o random_lines ; run: ./random_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// SHARE lanes read one line of SHARE * 16 bytes (SHARE = 4: 64 B, 8: 128 B); SHARE = 1: one lane reads 16 bytes of its own 64-byte line
// (one request per lane).  INFL independent chains per lane group.
template <int SHARE, int INFL>
__global__ __launch_bounds__(256) void k_walk(const uint4 *tab, uint32_t nlines, uint32_t line16, int steps, uint32_t *sink)
{
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t sub = SHARE > 1 ? (threadIdx.x & (SHARE - 1)) : 0u;
	const uint32_t chain0 = (gid / SHARE) * INFL;
	uint32_t idx[INFL];
	#pragma unroll
	for (int k = 0; k < INFL; ++k) idx[k] = mix(chain0 + k + 12345u) % nlines;
	uint32_t acc = 0;
	for (int s = 0; s < steps; ++s) {
		uint4 q[INFL];
		#pragma unroll
		for (int k = 0; k < INFL; ++k) q[k] = tab[(uint64_t)idx[k] * line16 + sub];
		#pragma unroll
		for (int k = 0; k < INFL; ++k) {
			uint32_t v = q[k].x ^ q[k].y ^ q[k].z ^ q[k].w;
			if (SHARE > 1) { for (int o = 1; o < SHARE; o <<= 1) v ^= __shfl_xor(v, o, 64); }      // the whole group sees the whole line (as the rank counts of a walk do)
			acc += v;
			idx[k] = mix(v + idx[k] + (uint32_t)s) % nlines;
		}
	}
	if (acc == 0x12345678u) sink[gid & 1023] = acc;
}

template <int SHARE, int INFL>
static void run(const uint4 *tab, uint64_t bytes, uint32_t linebytes, uint32_t *sink, uint32_t groups)
{
	const uint32_t line16 = linebytes / 16, nlines = (uint32_t)(bytes / linebytes);
	const int steps = 256;
	const uint32_t threads = groups * SHARE;
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	hipLaunchKernelGGL((k_walk<SHARE, INFL>), dim3(threads / 256), dim3(256), 0, 0, tab, nlines, line16, 8, sink);      // warm-up
	CK(hipEventRecord(a));
	hipLaunchKernelGGL((k_walk<SHARE, INFL>), dim3(threads / 256), dim3(256), 0, 0, tab, nlines, line16, steps, sink);
	CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
	const double reads = (double)groups * INFL * steps;
	printf("table %5.0f MB  line %3u B  %d lane(s)/line  %d in flight per group  %7u groups: %6.2f ms  %6.1f G lines/s  %5.2f TB/s\n",
	       bytes / 1e6, linebytes, SHARE, INFL, groups, ms, reads / ms / 1e6, reads * linebytes / ms / 1e9);
}

int main()
{
	const uint64_t maxb = 2ull << 30;
	uint4 *tab; uint32_t *sink;
	CK(hipMalloc(&tab, maxb)); CK(hipMalloc(&sink, 4096));
	{
		// random content (the chains' next indices are hashes of it)
		uint32_t *h = (uint32_t *)malloc(64 << 20);
		uint32_t x = 1; for (size_t i = 0; i < (64u << 20) / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = x; }
		for (uint64_t o = 0; o < maxb; o += 64 << 20) CK(hipMemcpy((char *)tab + o, h, 64 << 20, hipMemcpyHostToDevice));
		free(h);
	}
	const uint64_t sizes[3] = {125ull << 20, 500ull << 20, 2ull << 30};
	for (int si = 0; si < 3; ++si) {
		const uint64_t bytes = sizes[si];
		// as many groups as k_gap_walk4 has chains (512 Ki), and four times that
		for (uint32_t groups : {1u << 19, 1u << 21}) {
			run<4, 1>(tab, bytes, 64, sink, groups);
			run<4, 2>(tab, bytes, 64, sink, groups);
			run<4, 4>(tab, bytes, 64, sink, groups);
			run<1, 1>(tab, bytes, 64, sink, groups);
			run<1, 4>(tab, bytes, 64, sink, groups);
			run<1, 8>(tab, bytes, 64, sink, groups);
			run<8, 1>(tab, bytes, 128, sink, groups);
			run<8, 4>(tab, bytes, 128, sink, groups);
		}
	}
	return 0;
}
