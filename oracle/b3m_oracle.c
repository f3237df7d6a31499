/*
 * oracle/b3m_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's BWT-construction path (bwtb3m, balanced
 * block merging).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product (bwtb3m_amd/) never does.
 *
 * PARITY STATUS
 *   The algorithm body of the reference lives in libmaus2 (>= 2.0.349,
 *   /root/reference/configure.ac:39), which is absent from /root/reference and
 *   from this image, so the reference cannot be built here.
 *   - decoded BWT symbols: pinned by DEFINITION -- /root/reference/README.md:55-59
 *     (original Burrows-Wheeler definition, circular comparisons, no terminator)
 *     and its executable form BWT[i] = s[(SA[i]+n-1)%n] at
 *     /root/reference/src/lcpbit.cpp:3667-3669.  Two independent sorters here
 *     (naive rotation sort and circular prefix doubling) must agree.
 *   - block-merging intermediates (leaf SA, gap arrays G, gt bits) follow the
 *     conventions visible in the reference tree: 0-based rank_c(B,r) counts
 *     occurrences in B[0,r) (/root/reference/src/bwtdecodeblock.cpp:232-241,
 *     262-276), lf(i) = D[sym] + rank_sym(i) (bwtdecodeblock.cpp:280-284),
 *     LF by histogram prefix sums (/root/reference/src/lcpbit.cpp:3198-3277,
 *     3362-3366).  The merge recurrence is SURVEY.md Appendix A; it is proved
 *     here by exhaustive small-string tests against the definition.
 *   - RL .bwt container bytes: PARITY UNPINNED (no libmaus2-written sample).
 *
 * All symbol arrays are one byte per symbol (already unpacked).  Positions are
 * int64_t in the API; internals use uint32_t and require n < 2^31.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* 1. Definition-level oracle                                                 */
/* ------------------------------------------------------------------------- */

/* Naive rotation sort, O(n^2 log n).  README.md:55-59: the input string is
 * circular for the sake of comparisons, no terminator.  Ties (equal rotations
 * of a periodic string) keep position order; they cannot change the BWT. */
static const uint8_t *g_t;
static int64_t g_n;
static int cmp_rot(const void *pa, const void *pb)
{
	int64_t a = *(const int64_t *)pa, b = *(const int64_t *)pb;
	for (int64_t k = 0; k < g_n; ++k) {
		uint8_t ca = g_t[(a + k) % g_n], cb = g_t[(b + k) % g_n];
		if (ca != cb) return ca < cb ? -1 : 1;
	}
	return a < b ? -1 : (a > b);
}
EXPORT int b3mo_sa_naive(const uint8_t *t, int64_t n, int64_t *sa)
{
	for (int64_t i = 0; i < n; ++i) sa[i] = i;
	g_t = t; g_n = n;
	qsort(sa, (size_t)n, sizeof(int64_t), cmp_rot);
	return 0;
}

/* Circular prefix doubling (Manber-Myers bucket refinement), O(n log maxlcp).
 * After the round with offset h the array is sorted by the first 2h symbols of
 * every rotation; stops when all ranks are distinct or 2h >= n (equal
 * rotations: remaining ties are genuine and harmless). */
EXPORT int b3mo_sa_circular(const uint8_t *t, int64_t n64, int64_t *sa_out)
{
	if (n64 <= 0) return 0;
	if (n64 >= ((int64_t)1 << 31)) return -1;
	uint32_t n = (uint32_t)n64;
	uint32_t *sa = malloc(4 * (size_t)n), *sa2 = malloc(4 * (size_t)n);
	uint32_t *rk = malloc(4 * (size_t)n), *rk2 = malloc(4 * (size_t)n);
	uint32_t *nxt = malloc(4 * (size_t)n);
	if (!sa || !sa2 || !rk || !rk2 || !nxt) return -2;
	/* round 0: counting sort by first symbol; rank = head index of the group */
	{
		uint32_t cnt[257]; memset(cnt, 0, sizeof cnt);
		for (uint32_t i = 0; i < n; ++i) cnt[t[i] + 1]++;
		for (int c = 0; c < 256; ++c) cnt[c + 1] += cnt[c];
		uint32_t head[256]; memcpy(head, cnt, sizeof head);
		for (uint32_t i = 0; i < n; ++i) { sa[cnt[t[i]]++] = i; }
		for (uint32_t i = 0; i < n; ++i) rk[i] = head[t[i]];
	}
	for (uint64_t h = 1; h < n; h *= 2) {
		/* nxt[g] = next free slot of the group whose head index is g */
		int all_distinct = 1;
		for (uint32_t k = 0; k < n; ++k) nxt[k] = k; /* only heads are used */
		/* visit rotations in order of their first h symbols; predecessor j = s-h gets
		 * the next slot of its own group => sorted by (rk[j], rk[j+h]) */
		for (uint32_t k = 0; k < n; ++k) {
			uint32_t s = sa[k];
			uint32_t j = (uint32_t)(((uint64_t)s + n - (h % n)) % n);
			sa2[nxt[rk[j]]++] = j;
		}
		/* new ranks */
		uint32_t hh = (uint32_t)(h % n);
		rk2[sa2[0]] = 0;
		for (uint32_t k = 1; k < n; ++k) {
			uint32_t a = sa2[k - 1], b = sa2[k];
			uint32_t a2 = a + hh; if (a2 >= n) a2 -= n;
			uint32_t b2 = b + hh; if (b2 >= n) b2 -= n;
			if (rk[a] == rk[b] && rk[a2] == rk[b2]) { rk2[b] = rk2[a]; all_distinct = 0; }
			else rk2[b] = k;
		}
		uint32_t *tmp = sa; sa = sa2; sa2 = tmp;
		tmp = rk; rk = rk2; rk2 = tmp;
		if (all_distinct) break;
	}
	for (uint32_t i = 0; i < n; ++i) sa_out[i] = sa[i];
	free(sa); free(sa2); free(rk); free(rk2); free(nxt);
	return 0;
}

/* BWT[i] = s[((SA[i]+n)-1) % n]   -- /root/reference/src/lcpbit.cpp:3667-3669 */
EXPORT int b3mo_bwt_from_sa(const uint8_t *t, int64_t n, const int64_t *sa, uint8_t *bwt)
{
	for (int64_t i = 0; i < n; ++i) bwt[i] = t[((sa[i] + n) - 1) % n];
	return 0;
}

EXPORT int b3mo_bwt(const uint8_t *t, int64_t n, uint8_t *bwt)
{
	int64_t *sa = malloc(8 * (size_t)(n > 0 ? n : 1));
	if (!sa) return -2;
	int rc = b3mo_sa_circular(t, n, sa);
	if (!rc) b3mo_bwt_from_sa(t, n, sa, bwt);
	free(sa);
	return rc;
}

/* ------------------------------------------------------------------------- */
/* 2. Restatement of balanced block merging                                   */
/* ------------------------------------------------------------------------- */

/* Compare rotations S_x and S_y of the circular text; 0 if equal over n symbols. */
static int cmp_suffix(const uint8_t *t, uint32_t n, uint32_t x, uint32_t y)
{
	for (uint32_t k = 0; k < n; ++k) {
		uint8_t a = t[x], b = t[y];
		if (a != b) return a < b ? -1 : 1;
		if (++x == n) x = 0;
		if (++y == n) y = 0;
	}
	return 0;
}

/* lcpnext: max LCP between the suffix at the next block's
 * start and any suffix starting inside the block; bounded search first
 * (largelcpthres, /root/reference/src/bwtb3m.cpp:55), unbounded for offenders.
 * Z-function of P = T[e, e+K) over the block text. */
EXPORT int64_t b3mo_lcpnext(const uint8_t *t, int64_t n64, int64_t a, int64_t m, int64_t thres)
{
	uint32_t n = (uint32_t)n64;
	uint64_t K = thres > 0 ? (uint64_t)thres : 16384;
	for (;;) {
		if (K > n) K = n;
		/* string X = P (K symbols from e=a+m) followed by block text + K look-ahead */
		uint64_t len = K + (uint64_t)m + K;
		uint8_t *x = malloc(len);
		uint32_t *z = malloc(4 * len);
		if (!x || !z) { free(x); free(z); return -2; }
		uint32_t e = (uint32_t)((a + m) % n64);
		for (uint64_t i = 0; i < K; ++i) x[i] = t[(e + i) % n];
		for (uint64_t i = 0; i < (uint64_t)m + K; ++i) x[K + i] = t[((uint64_t)a + i) % n];
		uint64_t l = 0, r = 0; z[0] = (uint32_t)len;
		uint64_t best = 0;
		for (uint64_t i = 1; i < len; ++i) {
			uint64_t zi = 0;
			if (i < r) { zi = z[i - l]; if (zi > r - i) zi = r - i; }
			while (i + zi < len && zi < K && x[zi] == x[i + zi]) ++zi;
			z[i] = (uint32_t)zi;
			if (i + zi > r) { l = i; r = i + zi; }
			if (i >= K && i < K + (uint64_t)m && zi > best) best = zi;
		}
		free(x); free(z);
		if (best < K || K >= n) return (int64_t)best;
		K *= 4;
	}
}

/* Leaf block sort (row a7): SA of the block's suffixes as rotations of the whole
 * text, via the reference's scheme: sort the window T[a, a+m+lcpnext] as a
 * TERMINATED string (running off the end is smaller than any symbol) by prefix
 * doubling and keep suffixes starting in the block.  This is the role divsufsort
 * plays in BwtMergeBlockSortRequest (sorter type at
 * /root/reference/src/lcpbit.cpp:3658).
 * sa_out[r] = absolute text position. */
EXPORT int b3mo_leaf_sort_window(const uint8_t *t, int64_t n64, int64_t a, int64_t m,
                                 int64_t largelcpthres, int64_t *sa_out)
{
	uint32_t n = (uint32_t)n64;
	if (m >= n64) return b3mo_sa_circular(t, n64, sa_out); /* single block: circular sort */
	int64_t lcpnext = b3mo_lcpnext(t, n64, a, m, largelcpthres);
	if (lcpnext < 0) return -2;
	uint64_t len64 = (uint64_t)m + (uint64_t)lcpnext + 1;
	if (len64 > (uint64_t)m + n) len64 = (uint64_t)m + n;
	uint32_t len = (uint32_t)len64;
	uint8_t *w = malloc(len);
	uint32_t *b0 = malloc(4 * (size_t)len), *b1 = malloc(4 * (size_t)len);
	uint32_t *r0 = malloc(4 * (size_t)len), *r1 = malloc(4 * (size_t)len), *nx = malloc(4 * (size_t)len);
	if (!w || !b0 || !b1 || !r0 || !r1 || !nx) return -2;
	for (uint32_t i = 0; i < len; ++i) w[i] = t[((uint64_t)a + i) % n];
	/* inline doubling so that we know the final buffer */
	uint32_t *sa = b0, *sa2 = b1, *rk = r0, *rk2 = r1;
	{
		uint32_t cnt[257]; memset(cnt, 0, sizeof cnt);
		for (uint32_t i = 0; i < len; ++i) cnt[w[i] + 1]++;
		for (int c = 0; c < 256; ++c) cnt[c + 1] += cnt[c];
		uint32_t head[256]; memcpy(head, cnt, sizeof head);
		for (uint32_t i = 0; i < len; ++i) sa[cnt[w[i]]++] = i;
		for (uint32_t i = 0; i < len; ++i) rk[i] = head[w[i]];
		for (uint64_t h = 1; h < len; h *= 2) {
			int all_distinct = 1;
			for (uint32_t k = 0; k < len; ++k) nx[k] = k;
			for (uint32_t j = len - (uint32_t)h; j < len; ++j) sa2[nx[rk[j]]++] = j;
			for (uint32_t k = 0; k < len; ++k) {
				uint32_t s = sa[k];
				if (s >= h) { uint32_t j = s - (uint32_t)h; sa2[nx[rk[j]]++] = j; }
			}
			rk2[sa2[0]] = 0;
			for (uint32_t k = 1; k < len; ++k) {
				uint32_t x = sa2[k - 1], y = sa2[k];
				int64_t x2 = (uint64_t)x + h < len ? (int64_t)rk[x + h] : -1;
				int64_t y2 = (uint64_t)y + h < len ? (int64_t)rk[y + h] : -1;
				if (rk[x] == rk[y] && x2 == y2) { rk2[y] = rk2[x]; all_distinct = 0; }
				else rk2[y] = k;
			}
			uint32_t *tmp = sa; sa = sa2; sa2 = tmp;
			tmp = rk; rk = rk2; rk2 = tmp;
			if (all_distinct) break;
		}
	}
	int64_t o = 0;
	for (uint32_t k = 0; k < len; ++k)
		if (sa[k] < (uint64_t)m) sa_out[o++] = (int64_t)(((uint64_t)a + sa[k]) % n);
	free(w); free(b0); free(b1); free(r0); free(r1); free(nx);
	return o == m ? 0 : -3;
}

/* Leaf products: block BWT, p0 = rank of S_a, gt bits relative to S_a.
 * gt[x-a] for x in (a, a+m] : [S_x > S_a]  (bit m = the one look-ahead comparison). */
EXPORT int b3mo_leaf_products(const uint8_t *t, int64_t n64, int64_t a, int64_t m,
                              const int64_t *sa, uint8_t *bwt, int64_t *p0_out, uint8_t *gt /* m+1 */)
{
	uint32_t n = (uint32_t)n64;
	int64_t p0 = -1;
	for (int64_t r = 0; r < m; ++r) {
		bwt[r] = t[((uint64_t)sa[r] + n - 1) % n];
		if (sa[r] == a) p0 = r;
	}
	if (p0 < 0) return -3;
	*p0_out = p0;
	gt[0] = 0;
	for (int64_t r = 0; r < m; ++r) gt[sa[r] - a] = (uint8_t)(r > p0);
	gt[0] = 0;
	gt[m] = (uint8_t)(cmp_suffix(t, n, (uint32_t)((a + m) % n64), (uint32_t)a) > 0);
	return 0;
}

/* rank of S_x among the (sorted) suffixes sa[0,m): #{ i : S_i < S_x }  */
EXPORT int64_t b3mo_rank_of(const uint8_t *t, int64_t n64, const int64_t *sa, int64_t m, int64_t x)
{
	uint32_t n = (uint32_t)n64;
	int64_t lo = 0, hi = m;
	while (lo < hi) {
		int64_t mid = (lo + hi) >> 1;
		if (cmp_suffix(t, n, (uint32_t)sa[mid], (uint32_t)(x % n64)) < 0) lo = mid + 1; else hi = mid;
	}
	return lo;
}

/* Rank dictionary over BWT(L): occ[b][c] = #c in bwtL[0, b*BS) with the entry at
 * p0L (it belongs to position a-1, outside L) never counted; C[c] = #{i in L:
 * T[i] < c}.  rank_c(B,r) counts B[0,r) -- /root/reference/src/bwtdecodeblock.cpp:232-241. */
#define LR_BS 64
typedef struct {
	uint32_t *occ; uint64_t *C; int sigma; const uint8_t *bwtL; int64_t m, p0L; uint8_t lastL;
} lrank_t;

static int lrank_build(lrank_t *lr, const uint8_t *t, uint32_t n, int64_t a, int64_t m,
                       const uint8_t *bwtL, int64_t p0L, int sigma)
{
	uint64_t nb = (uint64_t)m / LR_BS + 1;
	lr->occ = calloc(nb * (size_t)sigma, 4);
	lr->C = calloc((size_t)sigma + 1, 8);
	uint32_t *run = calloc((size_t)sigma, 4);
	if (!lr->occ || !lr->C || !run) return -2;
	lr->sigma = sigma; lr->bwtL = bwtL; lr->m = m; lr->p0L = p0L;
	for (int64_t r = 0; r < m; ++r) {
		if (r % LR_BS == 0) memcpy(lr->occ + (r / LR_BS) * sigma, run, 4 * (size_t)sigma);
		if (r != p0L) run[bwtL[r]]++;
	}
	if (m % LR_BS == 0) memcpy(lr->occ + (m / LR_BS) * sigma, run, 4 * (size_t)sigma);
	free(run);
	for (int64_t i = 0; i < m; ++i) lr->C[t[((uint64_t)a + i) % n] + 1]++;
	for (int c = 0; c < sigma; ++c) lr->C[c + 1] += lr->C[c];
	lr->lastL = t[((uint64_t)a + m - 1) % n];
	return 0;
}
static void lrank_free(lrank_t *lr) { free(lr->occ); free(lr->C); }

/* One LF chain of SURVEY.md Appendix A over R positions k = k_hi-1 .. k_lo
 * (k relative to R's start a+m); rho = rank_L(S_{a+m+k_hi}) on entry.
 *   rank_L(j) = C_L[c] + rank_c(BWT_L, rho) + [T[a+m-1] = c and gt_R[j+1]]
 * (the "- [T[a-1] = c and p0 < rho]" term is the never-counted p0 entry). */
static void gap_chain(const uint8_t *t, uint32_t n, int64_t a, const lrank_t *lr,
                      const uint8_t *gtR, int64_t k_hi, int64_t k_lo, int64_t rho,
                      uint32_t *G, uint8_t *gtP, int64_t *rankR, int atomic)
{
	int64_t m = lr->m; int sigma = lr->sigma;
	for (int64_t k = k_hi - 1; k >= k_lo; --k) {
		uint8_t c = t[((uint64_t)a + m + k) % n];
		int64_t b = rho / LR_BS;
		int64_t cnt = lr->occ[b * sigma + c];
		for (int64_t r = b * LR_BS; r < rho; ++r) cnt += (r != lr->p0L && lr->bwtL[r] == c);
		rho = (int64_t)lr->C[c] + cnt + ((lr->lastL == c && gtR[k + 1]) ? 1 : 0);
		if (atomic) {
			#pragma omp atomic
			G[rho]++;
		} else G[rho]++;
		if (gtP) gtP[m + k] = (uint8_t)(rho > lr->p0L);
		if (rankR) rankR[k] = rho;
	}
}

/* Gap array (row a8), single chain from the end of R.
 *   L = [a, a+m), R = [a+m, a+m+mr).  gtR[k], k in [1, mr], = [S_{a+m+k} > S_{a+m}].
 *   rank_end = rank of S_{a+m+mr} in L.  sigma = alphabet size.
 * Out: G[0..m] counts (sum = mr); gtP[k] for k in [m, m+mr] = [S_{a+k} > S_a]
 * (may be NULL); rankR[k] = rank_L(a+m+k) for k in [0,mr) (may be NULL). */
EXPORT int b3mo_gap(const uint8_t *t, int64_t n64, int64_t a, int64_t m, int64_t mr,
                    const uint8_t *bwtL, int64_t p0L, const uint8_t *gtR, int64_t rank_end,
                    int sigma, uint32_t *G, uint8_t *gtP, int64_t *rankR)
{
	lrank_t lr;
	int rc = lrank_build(&lr, t, (uint32_t)n64, a, m, bwtL, p0L, sigma);
	if (rc) return rc;
	memset(G, 0, 4 * (size_t)(m + 1));
	if (gtP) gtP[m + mr] = (uint8_t)(rank_end > p0L);
	gap_chain(t, (uint32_t)n64, a, &lr, gtR, mr, 0, rank_end, G, gtP, rankR, 0);
	lrank_free(&lr);
	return 0;
}

/* One chain of the walk from an arbitrary anchor: positions k = k_hi-1 .. k_lo of R, rho =
 * rank_L(S_{a+m+k_hi}).  rankR[k-k_lo] and gtP[m+k] are written for the walked positions only
 * (what one rank of the multi-GPU driver computes for one chain). */
EXPORT int b3mo_gap_range(const uint8_t *t, int64_t n64, int64_t a, int64_t m, int64_t mr,
                          const uint8_t *bwtL, int64_t p0L, const uint8_t *gtR, int sigma,
                          int64_t k_hi, int64_t k_lo, int64_t rho, int64_t *rankR, uint8_t *gtP)
{
	lrank_t lr;
	(void)mr;
	int rc = lrank_build(&lr, t, (uint32_t)n64, a, m, bwtL, p0L, sigma);
	if (rc) return rc;
	uint32_t *G = calloc((size_t)m + 1, 4);
	gap_chain(t, (uint32_t)n64, a, &lr, gtR, k_hi, k_lo, rho, G, gtP, rankR ? rankR - k_lo : NULL, 0);
	free(G);
	lrank_free(&lr);
	return 0;
}

/* Interleave (row a10): for r in 0..m: emit G[r] symbols of BWT(R), then BWT(L)[r].
 * p0P = new rank of L's p0 entry. */
EXPORT int b3mo_interleave(const uint32_t *G, const uint8_t *bwtL, int64_t m,
                           const uint8_t *bwtR, int64_t mr, uint8_t *out,
                           int64_t p0L, int64_t *p0P)
{
	int64_t o = 0, q = 0;
	for (int64_t r = 0; r <= m; ++r) {
		for (uint32_t g = 0; g < G[r]; ++g) out[o++] = bwtR[q++];
		if (r < m) { if (r == p0L && p0P) *p0P = o; out[o++] = bwtL[r]; }
	}
	return (q == mr && o == m + mr) ? 0 : -3;
}

/* Whole path: numblocks leaves of (nearly) equal size, balanced binary merge
 * tree; numthreads leaves sorted concurrently, merges walked by numthreads LF
 * chains started from anchors, interleave split at equal parts of the rank range
 * (numthreads = /root/reference/src/bwtb3m.cpp:48-50).
 * This is what bench.py times as cpu_baseline kind "port".
 * stats (optional, 4 doubles): wall seconds in leaf sort, gap, interleave, total. */
typedef struct { int64_t a, m, p0; uint8_t *bwt; uint8_t *gt; int lo, hi; } node_t;

static double now_s(void)
{
#ifdef _OPENMP
	return omp_get_wtime();
#else
	return 0.0;
#endif
}

EXPORT int b3mo_b3m(const uint8_t *t, int64_t n, int numblocks, int numthreads, int sigma,
                    int64_t largelcpthres, uint8_t *bwt_out, int64_t *p0_out, double *stats)
{
	if (n <= 0) return 0;
	if (numblocks < 1) numblocks = 1;
	if ((int64_t)numblocks > n) numblocks = (int)n;
	if (numthreads < 1) numthreads = 1;
#ifdef _OPENMP
	omp_set_num_threads(numthreads);
#endif
	double t0 = now_s();
	int B = numblocks;
	int64_t bs = (n + B - 1) / B;
	B = (int)((n + bs - 1) / bs);
	node_t *cur = calloc((size_t)B, sizeof(node_t));
	int64_t **leaf_sa = calloc((size_t)B, sizeof(int64_t *));
	int64_t *leaf_a = calloc((size_t)B, 8), *leaf_m = calloc((size_t)B, 8);
	int err = 0;
	#pragma omp parallel for schedule(dynamic, 1)
	for (int k = 0; k < B; ++k) {
		int64_t a = (int64_t)k * bs, e = a + bs; if (e > n) e = n;
		int64_t m = e - a;
		cur[k].a = a; cur[k].m = m; cur[k].lo = k; cur[k].hi = k + 1;
		leaf_a[k] = a; leaf_m[k] = m;
		leaf_sa[k] = malloc(8 * (size_t)m);
		cur[k].bwt = malloc((size_t)m);
		cur[k].gt = malloc((size_t)m + 1);
		int rc = b3mo_leaf_sort_window(t, n, a, m, largelcpthres, leaf_sa[k]);
		if (!rc) rc = b3mo_leaf_products(t, n, a, m, leaf_sa[k], cur[k].bwt, &cur[k].p0, cur[k].gt);
		if (rc) err = rc;
	}
	double t1 = now_s(), tgap = 0, til = 0;
	int cnt = B;
	while (cnt > 1 && !err) {
		int nn = (cnt + 1) / 2, nm = cnt / 2;
		node_t *nx = calloc((size_t)nn, sizeof(node_t));
		if (cnt & 1) nx[nn - 1] = cur[cnt - 1];
		int S = (numthreads + nm - 1) / nm;           /* chains per merge */
		lrank_t *lr = calloc((size_t)nm, sizeof(lrank_t));
		uint32_t **G = calloc((size_t)nm, sizeof(uint32_t *));
		int64_t *anch = calloc((size_t)nm * (size_t)(S + 1), 8);
		double s0 = now_s();
		#pragma omp parallel for schedule(dynamic, 1)
		for (int p = 0; p < nm; ++p) {
			node_t *L = &cur[2 * p], *R = &cur[2 * p + 1];
			node_t P; P.a = L->a; P.m = L->m + R->m; P.lo = L->lo; P.hi = R->hi; P.p0 = 0;
			P.bwt = malloc((size_t)P.m); P.gt = malloc((size_t)P.m + 1);
			memcpy(P.gt, L->gt, (size_t)L->m);
			nx[p] = P;
			G[p] = calloc((size_t)L->m + 1, 4);
			if (lrank_build(&lr[p], t, (uint32_t)n, L->a, L->m, L->bwt, L->p0, sigma)) err = -2;
		}
		/* anchors: rank in L of the suffix at each chain start = sum over L's leaves */
		#pragma omp parallel for schedule(dynamic, 1) collapse(2)
		for (int p = 0; p < nm; ++p)
			for (int s = 0; s < S; ++s) {
				node_t *L = &cur[2 * p], *R = &cur[2 * p + 1];
				int64_t k_hi = R->m - (R->m * s) / S;
				int64_t x = (R->a + k_hi) % n, rk = 0;
				for (int k = L->lo; k < L->hi; ++k) rk += b3mo_rank_of(t, n, leaf_sa[k], leaf_m[k], x);
				anch[(size_t)p * (S + 1) + s] = rk;
			}
		#pragma omp parallel for schedule(dynamic, 1) collapse(2)
		for (int p = 0; p < nm; ++p)
			for (int s = 0; s < S; ++s) {
				node_t *L = &cur[2 * p], *R = &cur[2 * p + 1];
				int64_t k_hi = R->m - (R->m * s) / S, k_lo = R->m - (R->m * (s + 1)) / S;
				int64_t rho = anch[(size_t)p * (S + 1) + s];
				if (s == 0) nx[p].gt[L->m + R->m] = (uint8_t)(rho > L->p0);
				gap_chain(t, (uint32_t)n, L->a, &lr[p], R->gt, k_hi, k_lo, rho, G[p], nx[p].gt, NULL, 1);
			}
		double s1 = now_s();
		/* interleave, split over S parts of the rank range */
		int64_t *qoff = calloc((size_t)nm * (size_t)(S + 1), 8);
		#pragma omp parallel for schedule(dynamic, 1) collapse(2)
		for (int p = 0; p < nm; ++p)
			for (int s = 0; s < S; ++s) {
				int64_t m = cur[2 * p].m;
				int64_t r0 = ((m + 1) * s) / S, r1 = ((m + 1) * (s + 1)) / S, sum = 0;
				for (int64_t r = r0; r < r1; ++r) sum += G[p][r];
				qoff[(size_t)p * (S + 1) + s + 1] = sum;
			}
		for (int p = 0; p < nm; ++p)
			for (int s = 0; s < S; ++s) qoff[(size_t)p * (S + 1) + s + 1] += qoff[(size_t)p * (S + 1) + s];
		#pragma omp parallel for schedule(dynamic, 1) collapse(2)
		for (int p = 0; p < nm; ++p)
			for (int s = 0; s < S; ++s) {
				node_t *L = &cur[2 * p], *R = &cur[2 * p + 1];
				int64_t m = L->m;
				int64_t r0 = ((m + 1) * s) / S, r1 = ((m + 1) * (s + 1)) / S;
				int64_t q = qoff[(size_t)p * (S + 1) + s], o = r0 + q;
				for (int64_t r = r0; r < r1; ++r) {
					for (uint32_t g = 0; g < G[p][r]; ++g) nx[p].bwt[o++] = R->bwt[q++];
					if (r < m) { if (r == L->p0) nx[p].p0 = o; nx[p].bwt[o++] = L->bwt[r]; }
				}
			}
		double s2 = now_s();
		tgap += s1 - s0; til += s2 - s1;
		for (int p = 0; p < nm; ++p) {
			lrank_free(&lr[p]); free(G[p]);
			free(cur[2 * p].bwt); free(cur[2 * p].gt); free(cur[2 * p + 1].bwt); free(cur[2 * p + 1].gt);
		}
		free(lr); free(G); free(anch); free(qoff); free(cur);
		cur = nx; cnt = nn;
	}
	if (!err) {
		memcpy(bwt_out, cur[0].bwt, (size_t)n);
		if (p0_out) *p0_out = cur[0].p0;
	}
	for (int k = 0; k < cnt; ++k) { free(cur[k].bwt); free(cur[k].gt); }
	free(cur);
	for (int k = 0; k < B; ++k) free(leaf_sa[k]);
	free(leaf_sa); free(leaf_a); free(leaf_m);
	double t2 = now_s();
	if (stats) { stats[0] = t1 - t0; stats[1] = tgap; stats[2] = til; stats[3] = t2 - t0; }
	return err;
}
