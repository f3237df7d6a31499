// engine.hip -- plan/orchestration of balanced block merging on one MI355X and the C ABI of the
// device-level entry points (include/b3m.h).  Role of
// libmaus2::suffixsort::bwtb3m::BwtMergeSortTemplate::computeBwt steps 3-9.
#include "b3m_internal.h"
#include "b3m_text.h"
#include <atomic>
#include <thread>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <cstdio>
#include <exception>
#include <functional>
#include <map>
#include <mutex>

namespace b3m {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

const char *const kfam_names[KF_COUNT] = {"keys", "radix_hist", "scan", "radix_scatter", "ties", "emit",
                                          "rank_build", "anchor_rank", "gap_walk", "interleave", "misc",
                                          "msd_hist", "msd_scatter", "bucket_finish", "msd_scatter_text"};

// ---- Timeline ------------------------------------------------------------------------------------
hipEvent_t Timeline::get()
{
	if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
	hipEvent_t e; B3M_HIP(hipEventCreate(&e)); return e;
}
size_t Timeline::begin(int fam, uint64_t bytes)
{
	Ev e; e.a = get(); e.b = get(); e.fam = fam; e.bytes = bytes;
	B3M_HIP(hipEventRecord(e.a, stream));
	evs.push_back(e);
	return evs.size() - 1;
}
void Timeline::end(size_t idx) { B3M_HIP(hipEventRecord(evs[idx].b, stream)); }
void Timeline::collect(double ms[KF_COUNT], uint64_t launches[KF_COUNT], uint64_t bytes[KF_COUNT])
{
	for (int i = 0; i < KF_COUNT; ++i) { ms[i] = 0; launches[i] = 0; bytes[i] = 0; }
	for (auto &e : evs) {
		float t = 0; (void)hipEventSynchronize(e.b); (void)hipEventElapsedTime(&t, e.a, e.b);
		ms[e.fam] += t; launches[e.fam]++; bytes[e.fam] += e.bytes;
	}
}
void Timeline::reset() { for (auto &e : evs) { pool.push_back(e.a); pool.push_back(e.b); } evs.clear(); }
Timeline::~Timeline() { reset(); for (auto e : pool) (void)hipEventDestroy(e); }

// ---- small kernels -------------------------------------------------------------------------------
// gt end bit of a leaf: [rank_leaf(S_end) > p0]
__global__ void k_set_endbit(const uint32_t *rank, const uint64_t *p0pair, uint32_t *gt, uint64_t bit)
{
	if (threadIdx.x == 0 && blockIdx.x == 0 && (uint64_t)rank[0] > p0pair[0]) atomicOr(&gt[bit >> 5], 1u << (bit & 31));
}
// p0 (rank of the block's first suffix inside the sort just done, 0xFFFFFFFF if that suffix lies in another suffix-range
// shard) -> (rank + off, pos); an absent p0 leaves the pair as it is (~0: set before the leaves)
__global__ void k_p0_to_pair(const uint32_t *p0, uint64_t *pair, uint64_t pos, uint64_t off)
{
	if (threadIdx.x == 0 && blockIdx.x == 0 && p0[0] != 0xFFFFFFFFu) { pair[0] = (uint64_t)p0[0] + off; pair[1] = pos; }
}
__global__ void k_emit_gt_ptr(const uint32_t *sa, uint64_t len, uint64_t start, const uint32_t *p0, uint32_t *gt)
{
	uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= len) return;
	uint32_t x = sa[r];
	if (r > *p0 && x != start) atomicOr(&gt[x >> 5], 1u << (x & 31));
}
__global__ void k_add_u32_to_i64(const uint32_t *src, long long *dst, uint64_t count)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < count) dst[i] += (long long)src[i];
}
__global__ void k_i64_to_u32(const long long *src, uint32_t *dst, uint64_t count)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < count) dst[i] = (uint32_t)src[i];
}
__global__ void k_endbit_i64(const long long *anchor, uint64_t p0L, uint32_t *gt, uint64_t bit)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		if ((uint64_t)anchor[0] > p0L) atomicOr(&gt[bit >> 5], 1u << (bit & 31)); else atomicAnd(&gt[bit >> 5], ~(1u << (bit & 31)));
	}
}
__global__ void k_clear_bits(uint32_t *dst, uint64_t b, uint64_t e)   // clears bits [b,e)
{
	uint64_t w = (b >> 5) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if ((w << 5) >= e) return;
	uint64_t lo = w << 5, hi = lo + 32;
	uint32_t mask = 0xFFFFFFFFu;
	if (lo < b) mask &= ~((1u << (b - lo)) - 1u);
	if (hi > e) mask &= ((1u << (e - lo)) - 1u);
	if (mask == 0xFFFFFFFFu) dst[w] = 0; else atomicAnd(&dst[w], ~mask);
}
static void clear_bits(uint32_t *dst, uint64_t b, uint64_t e, hipStream_t s)
{
	if (e <= b) return;
	uint64_t words = ((e + 31) >> 5) - (b >> 5);
	hipLaunchKernelGGL(k_clear_bits, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, s, dst, b, e);
}
// abs bit (start+k) |= rel bit k for k in [1,count]
__global__ void k_place_bits(const uint32_t *rel, uint64_t count, uint64_t start, uint32_t *abs)
{
	uint64_t k = 1 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k > count) return;
	if ((rel[k >> 5] >> (k & 31)) & 1u) { uint64_t x = start + k; atomicOr(&abs[x >> 5], 1u << (x & 31)); }
}
// walk prologue: fill the RankDict scalars that live on the device (p0 of L, end bit of P)
__global__ void k_merge_prologue(const uint64_t *p0pairL, const uint32_t *anchors, uint32_t *gt_next, uint64_t Rend, uint32_t *p0_u32)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		uint32_t p0 = (uint32_t)p0pairL[0];
		p0_u32[0] = p0;
		if (anchors[0] > p0) atomicOr(&gt_next[Rend >> 5], 1u << (Rend & 31));
	}
}

}  // namespace b3m

using namespace b3m;

// ================================================================================================
// plan
// ================================================================================================
struct Leaf { uint64_t a, m; };
struct NodeT { uint64_t a, m; int lo, hi; };
struct MergeT {
	NodeT L, R; int level;
	uint64_t chain_len, num_chains, slots;
	uint64_t anchor_off;    // offset into anchors / anchor positions arrays
};

struct b3m_plan {
	uint64_t n = 0; int type = 0; uint64_t blocksize = 0; uint64_t isarate = 0;
	int shard = 0, nshards = 1; uint64_t shard_cap = 0;   // suffix-range shard of a single-block plan
	int tie_mode = 0;                                     // suffix-range shard: B3M_TIES_* (how long repeats that reach over shards are finished)
	int keep_sa = 0;                                      // b3m_plan_set_keep_sa: the next runs keep the whole suffix array of the sort (tie export needs it)
	int sa_hint = 0;                                      // the last run's BWT-only sort had to start over with the suffix array kept (repeat-rich text): the next ones keep it from the start
	int seq_shards = 1;                                   // texts of 2^32 symbols and more on one GPU: the suffix-range shards of the one block, one after the other
	bool host_tier = false;                               // ... and a text below 2^32 symbols whose memory target no merge tree can meet: the same sequential shards, their slices
	                                                      // meant to leave for pinned host memory as they are finished (b3m_plan_run_host): the BWT is never in HBM as a whole
	uint64_t seq_half = 0;                                // sequential shards: d_bwt[0] holds two slices (one is copied out while the next shard sorts)
	std::vector<Leaf> leaves;
	std::vector<MergeT> merges;
	std::vector<std::vector<int>> level_merges;         // merge ids per level
	std::vector<std::vector<NodeT>> level_nodes;         // nodes entering each level
	int levels = 0;
	uint64_t ws_bytes = 0;
	// device memory
	uint8_t *arena = nullptr;
	uint8_t *d_text = nullptr; uint64_t text_cap = 0;
	uint8_t *d_bwt[2] = {nullptr, nullptr};
	uint32_t *d_gt[2] = {nullptr, nullptr}; uint64_t gt_words = 0;
	BlockSortWs bs;
	uint32_t *d_rank_blocks = nullptr, *d_rank_scratch = nullptr, *d_scan_ws = nullptr;
	uint32_t *d_ranks[2] = {nullptr, nullptr}; uint64_t max_slots = 0;
	uint32_t *d_tile_q = nullptr;
	RadixWs merge_radix;
	uint32_t *d_anchors = nullptr; uint64_t *d_anchor_pos = nullptr; uint64_t total_anchors = 0;
	uint64_t *d_hist = nullptr;            // [B][256]
	uint64_t *d_p0pairs = nullptr;         // [B][2]
	uint32_t *d_p0tmp = nullptr;           // [B] u32 p0 of leaves, + scratch
	uint32_t *d_endrank = nullptr;         // [B]
	uint64_t *d_endpos = nullptr;          // [B]
	uint64_t *d_samples = nullptr; uint32_t *d_sample_cnt = nullptr;
	std::vector<uint64_t> sample_off;      // per leaf offset (pairs)
	uint64_t total_samples = 0;
	// results
	b3m_result res{};
	std::vector<uint64_t> h_samples;       // (rank,pos) pairs in position order; filled on demand from h_raw (samples_of)
	uint64_t *h_raw = nullptr;             // pinned image of d_samples, copied on the run's own stream (a pageable copy of 47 MB cost 30 ms per C3 step)
	bool h_raw_pinned = false;
	bool samples_pending = false;          // h_raw is newer than h_samples
	bool samples_lazy = false;             // a large sample table (dense sampling: 750 MB at rate 64 and 3 Gbase) stays in HBM after a run and comes to the host when somebody
	                                       // asks for it (samples_of); b3m_compute_bwt streams X.preisa from the device image instead (b3m_plan_samples_device)
	bool raw_stale = false;                // lazy: h_raw is older than d_samples
	std::vector<uint64_t> h_hist;          // 256
	Timeline tl;
	double kms[KF_COUNT]{}; uint64_t klaunch[KF_COUNT]{}; uint64_t kbytes[KF_COUNT]{};
	uint64_t tie_left = 0; uint32_t *tie_U = nullptr, *tie_gid = nullptr;   // suffix-range shard: what its local rounds left tied (b3m_plan_tie_*)
	uint32_t *d_sa_slice = nullptr; uint64_t slice_off = 0, slice_len = 0;
	hipStream_t copy_stream = nullptr;     // host-memory runs: slices of the BWT leave on this stream while later buckets sort
	uint32_t *d_isa_cache = nullptr;       // suffix-range shards of a repeat-rich text: the inverse suffix array of the collective tie finish (4 n bytes, allocated at the first
	                                       // finish and kept: no hipMalloc per step on that path either)
	std::string warning;                   // what the caller should be told about this plan (a mem= target that cannot be met)
	int packed_out = 0;                    // host-memory runs hand the BWT over 2-bit packed (b3m_plan_set_packed_output)
	uint8_t *d_pack = nullptr;             // its device image
	b3m_e2e_times e2e{};
	// sequential suffix-range shards (texts of 2^32 symbols and more, the host tier): what the shards' few key-extension rounds leave tied, in rank order over all shards
	// (slot inside the shard, position, first of its group), and what their direct comparisons placed among long repeats (TieExtras); finished by prefix doubling on
	// that list alone when the last shard is through (sparse_tie_finish).  10 bytes per entry, cap entries of each kind.
	struct SparseTies {
		uint64_t cap = 0;
		uint32_t *slot = nullptr, *pos_lo = nullptr; uint8_t *pos_hi = nullptr, *head = nullptr;       // tied
		uint32_t *xslot = nullptr, *xpos_lo = nullptr; uint8_t *xpos_hi = nullptr; uint32_t *xcnt = nullptr;      // placed, but among long repeats
		struct Part { uint64_t first, count, xfirst, xcount, slice_off; };
		std::vector<Part> parts;
		uint64_t count = 0, xcount = 0, dmin = 0, dmax = 0;
		bool complete = true;
	} sp;
};

static inline uint64_t al256(uint64_t x) { return (x + 255) & ~255ull; }

// The arena of a plan.  One hipMalloc of 91 GB (C3) lands on physical pages that put level 1 of the record sort -- the gather between the two 24 GB record buffers -- into one
// of two modes, 10.8 or 11.4 ms, for as long as the arena lives (same virtual address either way; in a fresh process about three allocations in five get the slow one:
// profiles/NOTES_r04.md, tools/l1_mode.py).  Backed by separate physical allocations of 1 GB each, mapped into one reserved address range (hipMemCreate / hipMemMap), the
// first arena of a process runs it in 10.3-10.9 ms (fourteen fresh processes of fourteen, pieces of 64 MB to 4 GB; tools/arena_chunks.py) and nothing else gets slower.  256 MB pieces do the same for level 1
// but cost the list rounds of a repeat-rich text 1 % (random access likes large pieces); pieces for the record buffers alone do NOT give the fast mode; and a merge tree's
// gap walk loses 5 % on any mapped arena (95 instead of 91 ms), so only one-block, unsharded plans of 1 GB and more take this form.  B3M_ARENA_CHUNK_MB=c: pieces of c MB
// (0: one hipMalloc, as before).  Any failure of the mapping calls falls back to hipMalloc.
struct ArenaVmm { std::vector<hipMemGenericAllocationHandle_t> handles; uint64_t chunk = 0, size = 0; void *va = nullptr; };
static std::map<const void *, ArenaVmm> g_arena_vmm;        // (per plan)
static std::mutex g_arena_mu;
static ArenaVmm &arena_vmm_of(const void *plan) { std::lock_guard<std::mutex> g(g_arena_mu); return g_arena_vmm[plan]; }
static void arena_vmm_drop(const void *plan) { std::lock_guard<std::mutex> g(g_arena_mu); g_arena_vmm.erase(plan); }
static void arena_vmm_undo(ArenaVmm &V, uint64_t mapped)
{
	if (mapped) (void)hipMemUnmap(V.va, mapped);
	for (auto h : V.handles) (void)hipMemRelease(h);
	if (V.va) (void)hipMemAddressFree(V.va, V.size);
	(void)hipGetLastError();
	V = ArenaVmm{};
}
// rec_off / rec_len: the part of the arena that gets the pieces (all of it, as plan_create_impl calls this; in front of it and behind it: one physical allocation each)
static uint8_t *arena_alloc(uint64_t bytes, ArenaVmm &V, bool chunked_ok, uint64_t rec_off, uint64_t rec_len)
{
	const uint64_t chunk = (uint64_t)sw().ARENA_CHUNK_MB.get(1024) << 20;
	uint8_t *ptr = nullptr;
	if (chunk && chunked_ok && bytes >= (1ull << 30) && rec_len >= 2 * chunk) {
		int dev = 0; B3M_HIP(hipGetDevice(&dev));
		hipMemAllocationProp prop{};
		prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
		size_t gran = 0;
		bool ok = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran && chunk % gran == 0;
		uint64_t size = 0, r0 = 0, r1 = 0;
		if (ok) {
			size = (bytes + gran - 1) / gran * gran;
			r0 = (rec_off + gran - 1) / gran * gran;                       // the pieces start at the first granule inside the record buffers ...
			r1 = r0 + (rec_off + rec_len - r0) / chunk * chunk;            // ... and cover whole pieces of them
			V.chunk = chunk; V.size = size;
			ok = hipMemAddressReserve(&V.va, size, chunk, nullptr, 0) == hipSuccess;
		}
		uint64_t mapped = 0;
		auto back = [&](uint64_t o, uint64_t len) {                      // one physical allocation for [o, o + len) (in 8 GB pieces if the driver will not give it whole)
			if (!ok || !len) return;
			hipMemGenericAllocationHandle_t h;
			if (hipMemCreate(&h, len, &prop, 0) == hipSuccess) {
				V.handles.push_back(h);
				ok = hipMemMap((uint8_t *)V.va + o, len, 0, h, 0) == hipSuccess;
				if (ok) mapped = o + len;
				return;
			}
			(void)hipGetLastError();
			const uint64_t big = 8ull << 30;
			for (uint64_t q = 0; ok && q < len; q += big) {
				const uint64_t l = std::min(big, len - q);
				ok = hipMemCreate(&h, l, &prop, 0) == hipSuccess;
				if (!ok) break;
				V.handles.push_back(h);
				ok = hipMemMap((uint8_t *)V.va + o + q, l, 0, h, 0) == hipSuccess;
				if (ok) mapped = o + q + l;
			}
		};
		back(0, r0);
		for (uint64_t o = r0; ok && o < r1; o += chunk) {
			hipMemGenericAllocationHandle_t h;
			ok = hipMemCreate(&h, chunk, &prop, 0) == hipSuccess;
			if (!ok) break;
			V.handles.push_back(h);
			ok = hipMemMap((uint8_t *)V.va + o, chunk, 0, h, 0) == hipSuccess;
			if (ok) mapped = o + chunk;
		}
		back(r1, size - r1);
		if (ok) {
			hipMemAccessDesc acc{};
			acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
			ok = hipMemSetAccess(V.va, size, &acc, 1) == hipSuccess;
		}
		if (ok) return (uint8_t *)V.va;
		arena_vmm_undo(V, mapped);                  // (the mapping calls are not available, or memory is short in pieces: one hipMalloc says so in the usual way)
	}
	B3M_HIP(hipMalloc((void **)&ptr, bytes));
	return ptr;
}
static void arena_free(uint8_t *ptr, ArenaVmm &V)
{
	if (!ptr) return;
	if (!V.chunk) { (void)hipFree(ptr); return; }
	(void)hipMemUnmap(ptr, V.size);
	for (auto h : V.handles) (void)hipMemRelease(h);
	(void)hipMemAddressFree(ptr, V.size);
	V = ArenaVmm{};
}

// entries of each kind a plan of sequential shards keeps for the long repeats of its text (b3m_plan::SparseTies): a tenth of a shard's sort (the finish works in the
// sort's buffers: 2 + 3 * 2 tenths of its record buffers), i.e. 2 bytes per suffix of a shard beside the ~50 of its sort.  B3M_SPARSE_TIES=c: at most c entries (test
// switch; 0: the shards finish their repeats themselves, as before round 4)
static uint64_t sparse_ties_cap(uint64_t n, uint64_t shard_cap)
{
	uint64_t c = shard_cap / 10;
	if (sw().SPARSE_TIES.set) c = std::min<uint64_t>(c, (uint64_t)sw().SPARSE_TIES.v);
	return c;
}
static void chains_for(uint64_t mr, uint64_t &chain_len, uint64_t &num_chains)
{
	const uint64_t target = 512ull * 1024;      // lanes to fill: 256 CUs x 32 waves x 64
	const uint64_t min_len = 32;
	chain_len = std::max<uint64_t>(min_len, (mr + target - 1) / target);
	chain_len = (chain_len + 3) & ~3ull;         // whole quads of steps: the walk pads a chain's slots to a multiple of 4, and padding keys pile up in the rank sort
	num_chains = (mr + chain_len - 1) / chain_len;
	if (num_chains == 0) { num_chains = 1; }
}

static void build_tree(b3m_plan *p)
{
	uint64_t n = p->n, bs = p->blocksize;
	uint64_t B = (n + bs - 1) / bs;
	bs = (n + B - 1) / B;                       // equalise: b = ceil(n / numblocks)
	B = (n + bs - 1) / bs;
	p->blocksize = bs;
	for (uint64_t k = 0; k < B; ++k) { uint64_t a = k * bs; p->leaves.push_back({a, std::min(bs, n - a)}); }
	std::vector<NodeT> cur;
	for (uint64_t k = 0; k < B; ++k) cur.push_back({p->leaves[k].a, p->leaves[k].m, (int)k, (int)k + 1});
	int level = 0; uint64_t aoff = 0;
	while (cur.size() > 1) {
		p->level_nodes.push_back(cur);
		std::vector<NodeT> nx; std::vector<int> ids;
		for (size_t i = 0; i + 1 < cur.size(); i += 2) {
			MergeT M; M.L = cur[i]; M.R = cur[i + 1]; M.level = level;
			chains_for(M.R.m, M.chain_len, M.num_chains);
			M.slots = gap_walk_slots(M.num_chains, M.chain_len);
			M.anchor_off = aoff; aoff += M.num_chains;
			ids.push_back((int)p->merges.size());
			p->merges.push_back(M);
			nx.push_back({M.L.a, M.L.m + M.R.m, M.L.lo, M.R.hi});
			p->max_slots = std::max(p->max_slots, M.slots);
		}
		if (cur.size() & 1) nx.push_back(cur.back());
		p->level_merges.push_back(ids);
		cur = nx; ++level;
	}
	p->levels = level;
	p->total_anchors = aoff;
}

namespace {
struct DevBuf {
	void *p = nullptr;
	explicit DevBuf(uint64_t bytes) { B3M_HIP(hipMalloc(&p, bytes ? bytes : 16)); }
	~DevBuf() { if (p) (void)hipFree(p); }
	template <class X> X *as() { return (X *)p; }
};
struct PaddedText {
	DevBuf buf; TextView T;
	PaddedText(const void *d_text, uint64_t n, int type, hipStream_t s) : buf(al256(packed_text_bytes(n, type) + 64))
	{
		B3M_REQUIRE(type == B3M_INPUT_BYTESTREAM || type == B3M_INPUT_PAC || type == B3M_INPUT_PACTERM, "unsupported input type");
		B3M_REQUIRE(n >= 1 && n < (1ull << 32), "n out of range");
		uint64_t tb = packed_text_bytes(n, type);
		B3M_HIP(hipMemsetAsync(buf.p, 0, al256(tb + 64), s));
		B3M_HIP(hipMemcpyAsync(buf.p, d_text, tb, hipMemcpyDeviceToDevice, s));
		T = TextView{(const uint8_t *)buf.p, n, num_bases(n, type), type};
	}
};
void require_device()
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) throw Error("no HIP device available: libb3m has no CPU fallback");
}
}  // namespace

extern "C" {

int b3m_version(void) { return B3M_VERSION; }
const char *b3m_last_error(void) { return g_err.c_str(); }
const char *b3m_switch_name(int i) { const SwitchDoc *d; const int n = switches_describe(&d); return i >= 0 && i < n ? d[i].name : nullptr; }
const char *b3m_switch_doc(int i) { const SwitchDoc *d; const int n = switches_describe(&d); return i >= 0 && i < n ? d[i].doc : nullptr; }

static const char *type_names[7] = {"bytestream", "compactstream", "pac", "pacterm", "lz4", "utf-8", "symbols32"};
int b3m_parse_input_type(const char *name)
{
	if (name) for (int i = 0; i < 6; ++i) if (!strcmp(name, type_names[i])) return i;
	// the message of the reference's input-type switch, verbatim (/root/reference/src/checkbwt.cpp:274: the text thrown by
	// BwtMergeSort::computeBwt, which that program repeats)
	set_error("libmaus2::suffixsort::bwtb3m::BwtMergeSort::computeBwt: unknown/unsupported input type");
	return -1;
}
const char *b3m_input_type_name(int t) { return (t >= 0 && t < 7) ? type_names[t] : "unknown"; }      // (symbols32 has a name but is no file type: not parsed)

#define B3M_TRY try {
#define B3M_CATCH(ret)                                                          \
	} catch (const std::exception &e) { b3m::set_error(e.what()); return ret; } \
	catch (...) { b3m::set_error("unknown error"); return ret; }

void b3m_plan_destroy(b3m_plan *p);
void b3m_dist_destroy(struct b3m_dist_plan *p);
static b3m_plan *plan_create_impl(uint64_t n, int inputtype, uint64_t blocksize, uint64_t mem, uint64_t isasamplingrate, int shard, int nshards)
{
	b3m_plan *p = nullptr;
	B3M_TRY
	switches_load();
	B3M_REQUIRE(inputtype == B3M_INPUT_BYTESTREAM || inputtype == B3M_INPUT_PAC || inputtype == B3M_INPUT_PACTERM || inputtype == B3M_INPUT_SYMBOLS32,
	            std::string("input type ") + b3m_input_type_name(inputtype) + " is not supported by the GPU engine yet");
	B3M_REQUIRE(n >= 1, "empty input");
	// symbols32 (utf-8 texts of more than 256 distinct code points): one block through the key/index sort, uint32 symbols in and out
	B3M_REQUIRE(inputtype != B3M_INPUT_SYMBOLS32 || (n < (1ull << 32) && nshards == 1 && (blocksize == 0 || blocksize >= n)),
	            "symbols32 texts are sorted as one block of fewer than 2^32 symbols (no blocksize=, no shards)");
	if (inputtype == B3M_INPUT_SYMBOLS32) blocksize = n;
	B3M_REQUIRE(inputtype != B3M_INPUT_PACTERM || n >= 1, "pacterm needs n >= 1");
	B3M_REQUIRE(n < (1ull << 36), "n >= 2^36 symbols is not supported");
	B3M_REQUIRE(n < (1ull << 32) || ((inputtype == B3M_INPUT_PAC || inputtype == B3M_INPUT_PACTERM) && blocksize == 0),
	            "texts of 2^32 symbols and more: pac/pacterm only, sorted as suffix-range shards of one block (no blocksize=)");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) throw Error("no HIP device available: libb3m has no CPU fallback");
	size_t free_b = 0, total_b = 0;
	B3M_HIP(hipMemGetInfo(&free_b, &total_b));
	uint64_t budget = mem ? std::min<uint64_t>(mem, free_b) : (uint64_t)(free_b * 0.92);
	p = new b3m_plan();
	p->n = n; p->type = inputtype; p->isarate = isasamplingrate;
	p->shard = shard; p->nshards = nshards;
	if (nshards > 1) {
		B3M_REQUIRE(shard >= 0 && shard < nshards, "shard index out of range");
		B3M_REQUIRE((inputtype == B3M_INPUT_PAC || inputtype == B3M_INPUT_PACTERM) && n >= (1ull << 18), "suffix-range shards need a pac/pacterm input of at least 2^18 symbols");
		blocksize = n;                                         // one block = the whole text, every rank sorts a rank range of it
		p->shard_cap = n / nshards + n / 64 + (1ull << 20);    // whole level-0 digits: ~n/256 of imbalance, generous slack
		if (p->shard_cap > n) p->shard_cap = n;
		B3M_REQUIRE(p->shard_cap < (1ull << 32), "a suffix-range shard must stay below 2^32 suffixes: use more shards");
		p->tie_mode = n >= (1ull << 32) ? B3M_TIES_LOCAL : B3M_TIES_SA;
	} else if (n >= (1ull << 32)) {
		// one GPU, more than 2^32 suffixes: the same shards, run one after the other inside b3m_plan_run (the packed text is
		// resident once; every shard scans it again and keeps its range of level-0 digits)
		p->seq_shards = (int)(n / (3ull << 30)) + 1;
		blocksize = n;
		p->shard_cap = n / p->seq_shards + n / 64 + (1ull << 20);
	}
	// HOST TIER.  The part of a merge tree's workspace that does not shrink with the blocks -- text, two BWT arrays, gt bits, the ranks of a merge: 7.25 bytes
	// per symbol -- is what `mem=` could not get below (the reference meets a small mem= with files, /root/reference/README.md:42,46).  Sequential suffix-range
	// shards need the text (0.25 B/symbol) and ONE shard's sort; their slices of the BWT are final when the shard is through and leave for pinned host memory
	// (b3m_plan_run_host: two slice buffers, the copy of one behind the sort of the next).  Taken when no merge tree fits the target (B3M_FORCE_SPILL=S: test switch).
	if (nshards == 1 && p->seq_shards == 1 && blocksize == 0 && (inputtype == B3M_INPUT_PAC || inputtype == B3M_INPUT_PACTERM) && n >= (1ull << 18)) {
		int S = (int)sw().FORCE_SPILL.get(0);
		if (S < 2 && mem) {
			const uint64_t fixed_tree = packed_text_bytes(n, inputtype) + 2 * n + n / 2 + (64ull << 20);
			const uint64_t bmin = std::max<uint64_t>(1024, n / 1024);
			const bool tree_fits = fixed_tree + blocksort_ws_bytes(bmin) + 9 * n / 2 <= budget;      // (the test the block-size loop below ends on)
			// (the arena of such a plan, as it is carved below: text, two slice buffers, one shard's sort, the sample table, small tables)
			auto seq_arena = [&](uint64_t cap) {
				const uint64_t nsmp = isasamplingrate ? (n + isasamplingrate - 1) / isasamplingrate : 0;
				return al256(20 * sparse_ties_cap(n, cap) + 64) + al256(al256(packed_text_bytes(n, inputtype) + 64)) + al256(2 * al256(cap + 16) + 16) + al256(blocksort_ws_bytes(cap, n, true)) + al256(16 * (nsmp + 1)) + (1ull << 20);
			};
			if (!tree_fits)
				for (int t = 2; t <= 4096 && S < 2; ++t) {
					const uint64_t cap = std::min<uint64_t>(n, n / t + n / 64 + (1ull << 20));
					if (seq_arena(cap) <= budget) S = t;
					if (n / t < n / 64) break;                     // (more shards no longer shrink the slack-dominated capacity)
				}
		}
		if (S >= 2) {
			p->seq_shards = S; p->host_tier = true;
			blocksize = n;
			p->shard_cap = std::min<uint64_t>(n, n / (uint64_t)S + n / 64 + (1ull << 20));
		}
	}
	// per-suffix bytes of the leaf sort workspace ~ 2*(8+4) + 1 + 12 + radix scratch: see blocksort_ws_bytes
	if (blocksize == 0) {
		// fixed part: text, 2 bwt arrays, gt; merge part needs ranks (8 B per R suffix) + rank blocks
		uint64_t fixed = packed_text_bytes(n, inputtype) + 2 * n + n / 2 + (64ull << 20);
		uint64_t b = n;
		// byte alphabets: count rows of BWT(L), 16 B per symbol of the largest left side, instead of 0.33 B
		const uint64_t merge_part = inputtype == B3M_INPUT_BYTESTREAM ? 9 * n / 2 + 9 * n : 9 * n / 2;
		// (a budget below the fixed part -- text, two BWT arrays, gt bits, the ranks of a merge -- cannot be met by smaller blocks: the split
		// stops at 1024 leaf blocks instead of going down to 1 Ki-symbol blocks, which is correct but takes minutes)
		const uint64_t bmin = std::max<uint64_t>(1024, n / 1024);
		while (b > bmin && fixed + blocksort_ws_bytes(b) + (b < n ? merge_part : 0) > budget) b = (b + 1) / 2;
		blocksize = b;
	}
	if (blocksize > n) blocksize = n;
	p->blocksize = blocksize;
	build_tree(p);
	const uint64_t B = p->leaves.size();
	const bool bytes = inputtype == B3M_INPUT_BYTESTREAM;
	// ---- carve the arena ----
	uint64_t off = 0;
	auto take = [&](uint64_t bytes) { uint64_t o = off; off += al256(bytes); return o; };
	p->text_cap = al256(packed_text_bytes(n, inputtype) + 64);
	uint64_t o_text = take(p->text_cap);
	// (sequential shards: a device-resident run writes into the caller's n bytes; a host-memory run needs two slices here, not n bytes)
	p->seq_half = p->seq_shards > 1 ? al256(p->shard_cap + 16) : 0;
	const uint64_t bwt_cap = (nshards > 1) ? p->shard_cap : (p->seq_shards > 1 ? 2 * p->seq_half : n);
	const bool sharded_sort = nshards > 1 || p->seq_shards > 1;
	uint64_t o_bwt0 = take(bwt_cap + 16), o_bwt1 = take(B > 1 ? n + 16 : 16);
	p->gt_words = (n + 1 + 31) / 32 + 2;
	uint64_t o_gt0 = take(B > 1 ? 4 * p->gt_words : 16), o_gt1 = take(B > 1 ? 4 * p->gt_words : 16);      // (gt bits: merges only)
	// blocks of a merge tree are sorted with room for an overhang behind them (long repeats: tie_rounds_domain)
	const uint64_t ovh = (B > 1 && !sharded_sort) ? std::min(block_overhang(), p->blocksize / 8) : 0;
	const uint64_t sort_cap = sharded_sort ? p->shard_cap : p->blocksize + ovh;
	const uint64_t sort_span = std::max(p->blocksize, sort_cap);
	// one block (or a suffix-range shard of one) below 2^32 symbols: the tie-list buffers of the sort are allocated when a repeat-rich text first
	// needs them (20 of ~50 bytes per suffix: the arena of C3 is 90 GB instead of 150)
	const bool lazy_ties = B == 1 && n < (1ull << 32) && inputtype != B3M_INPUT_BYTESTREAM && inputtype != B3M_INPUT_SYMBOLS32;      // (byte alphabets sort by key/index pairs and list their ties always)
	uint64_t o_bs = take(blocksort_ws_bytes(sort_cap, sort_span, lazy_ties));
	uint64_t maxL = 0, maxTot = 0;
	for (auto &M : p->merges) { maxL = std::max(maxL, M.L.m); maxTot = std::max(maxTot, M.L.m + M.R.m); }
	uint64_t nbk = rank_blocks(maxL);
	uint64_t o_rb = 0, o_rs = 0, o_sw = 0, o_r0 = 0, o_r1 = 0, o_tq = 0, o_mr = 0;
	if (B > 1) {
		// byte alphabets: count rows (1 KB per 64 symbols of L), an aligned copy of BWT(L) + the 256 symbol starts + the build's tile counts
		if (bytes) { o_rb = take(1024 * byte_rank_rows(maxL)); o_rs = take(maxL + 256 + 1024 * byte_rank_tiles(maxL) + 1024); o_sw = take(64); }
		else { o_rb = take(64 * nbk); o_rs = take(64 * nbk + 64); o_sw = take(4 * scan_ws_elems(nbk) + 64); }
		o_r0 = take(4 * p->max_slots + 64); o_r1 = take(4 * p->max_slots + 64);
		o_tq = take(4 * (interleave_tiles(maxTot) + 2));
		o_mr = take(radix_ws_bytes(p->max_slots));
	}
	uint64_t o_an = take(4 * (p->total_anchors + 1)), o_ap = take(8 * (p->total_anchors + 1));
	uint64_t o_hist = take(8 * 256 * B), o_p0p = take(16 * B), o_p0t = take(4 * (B + 4)), o_er = take(4 * B), o_ep = take(8 * B);
	uint64_t nsamp = 0;
	p->sample_off.assign(B + 1, 0);
	if (isasamplingrate) {
		for (uint64_t k = 0; k < B; ++k) {
			uint64_t a = p->leaves[k].a, e = a + p->leaves[k].m;
			uint64_t c = (e + isasamplingrate - 1) / isasamplingrate - (a + isasamplingrate - 1) / isasamplingrate;
			p->sample_off[k + 1] = p->sample_off[k] + c;
		}
		nsamp = p->sample_off[B];
	}
	p->total_samples = nsamp;
	uint64_t o_sm = take(16 * (nsamp + 1)), o_sc = take(4 * (B + 1));
	p->sp.cap = (p->seq_shards > 1 || nshards > 1) ? sparse_ties_cap(n, p->shard_cap) : 0;
	uint64_t o_sp = take(p->sp.cap ? 20 * p->sp.cap + 64 : 16);
	p->ws_bytes = off;
	if (mem && off > mem) {
		// a memory target below what the smallest blocks need is exceeded, but never silently (the reference's mem= counts host RAM and
		// falls back to files, /root/reference/README.md:42,46; here everything is HBM-resident: text, two BWT arrays, gt bits, the ranks of a merge)
		char buf[320];
		snprintf(buf, sizeof buf, "mem=%llu cannot be met: %llu leaf blocks of %llu symbols still need a workspace of %llu bytes (%.2f bytes per symbol; "
		         "the part that does not shrink with the blocks is about 7.25 bytes per symbol); using that", (unsigned long long)mem,
		         (unsigned long long)B, (unsigned long long)p->blocksize, (unsigned long long)off, (double)off / (double)n);
		p->warning = buf;
	}
	if (off > free_b) {
		char buf[256];
		snprintf(buf, sizeof buf, "workspace of %.2f GB does not fit the %.2f GB of free HBM (block size %llu)", off / 1e9, free_b / 1e9,
		         (unsigned long long)p->blocksize);
		throw Error(buf);
	}
	p->arena = arena_alloc(off, arena_vmm_of(p), nshards <= 1 && B == 1, 0, off);
	uint8_t *A = p->arena;
	p->d_text = A + o_text; p->d_bwt[0] = A + o_bwt0; p->d_bwt[1] = A + o_bwt1;
	p->d_gt[0] = (uint32_t *)(A + o_gt0); p->d_gt[1] = (uint32_t *)(A + o_gt1);
	blocksort_ws_carve(p->bs, A + o_bs, sort_cap, sort_span, lazy_ties);
	if (B > 1) {
		p->d_rank_blocks = (uint32_t *)(A + o_rb); p->d_rank_scratch = (uint32_t *)(A + o_rs); p->d_scan_ws = (uint32_t *)(A + o_sw);
		p->d_ranks[0] = (uint32_t *)(A + o_r0); p->d_ranks[1] = (uint32_t *)(A + o_r1);
		p->d_tile_q = (uint32_t *)(A + o_tq);
		radix_ws_carve(p->merge_radix, A + o_mr, p->max_slots);
	}
	p->d_anchors = (uint32_t *)(A + o_an); p->d_anchor_pos = (uint64_t *)(A + o_ap);
	p->d_hist = (uint64_t *)(A + o_hist); p->d_p0pairs = (uint64_t *)(A + o_p0p); p->d_p0tmp = (uint32_t *)(A + o_p0t);
	p->d_endrank = (uint32_t *)(A + o_er); p->d_endpos = (uint64_t *)(A + o_ep);
	p->d_samples = (uint64_t *)(A + o_sm); p->d_sample_cnt = (uint32_t *)(A + o_sc);
	if (p->sp.cap) {
		const uint64_t c = p->sp.cap;
		p->sp.slot = (uint32_t *)(A + o_sp); p->sp.pos_lo = p->sp.slot + c; p->sp.xslot = p->sp.pos_lo + c; p->sp.xpos_lo = p->sp.xslot + c;
		p->sp.pos_hi = (uint8_t *)(p->sp.xpos_lo + c); p->sp.head = p->sp.pos_hi + c; p->sp.xpos_hi = p->sp.head + c;
		p->sp.xcnt = (uint32_t *)(A + o_sp + 19 * c + ((64 - (19 * c) % 64) % 64));      // (19 bytes per entry pair used; the counter behind them, aligned)
	}
	// static tables: anchor positions, leaf end positions
	std::vector<uint64_t> ap(p->total_anchors + 1, 0), ep(B, 0);
	for (auto &M : p->merges) {
		uint64_t Rend = M.R.a + M.R.m;
		for (uint64_t c = 0; c < M.num_chains; ++c) ap[M.anchor_off + c] = Rend - c * M.chain_len;
	}
	for (uint64_t k = 0; k < B; ++k) ep[k] = p->leaves[k].a + p->leaves[k].m;
	B3M_HIP(hipMemcpy(p->d_anchor_pos, ap.data(), 8 * ap.size(), hipMemcpyHostToDevice));
	B3M_HIP(hipMemcpy(p->d_endpos, ep.data(), 8 * B, hipMemcpyHostToDevice));
	p->h_hist.assign(256, 0);
	p->samples_lazy = 16 * nsamp > (64ull << 20);
	if (nsamp && !p->samples_lazy) {
		// pinned if the host can spare it (a dense sampling of a long text may not fit what the driver is willing to pin)
		if (hipHostMalloc((void **)&p->h_raw, 16 * nsamp, hipHostMallocDefault) == hipSuccess) p->h_raw_pinned = true;
		else { (void)hipGetLastError(); p->h_raw = (uint64_t *)malloc(16 * nsamp); B3M_REQUIRE(p->h_raw != nullptr, "out of host memory for the (rank,pos) samples"); }
	}
	return p;
	B3M_CATCH((b3m_plan_destroy(p), (b3m_plan *)nullptr))
}

b3m_plan *b3m_plan_create(uint64_t n, int inputtype, uint64_t blocksize, uint64_t mem, uint64_t isasamplingrate)
{
	return plan_create_impl(n, inputtype, blocksize, mem, isasamplingrate, 0, 1);
}
b3m_plan *b3m_plan_create_sharded(uint64_t n, int inputtype, int shard, int nshards, uint64_t mem, uint64_t isasamplingrate)
{
	return plan_create_impl(n, inputtype, 0, mem, isasamplingrate, shard, nshards);
}
uint64_t b3m_plan_slice_capacity(const b3m_plan *p) { return p ? (p->nshards > 1 ? p->shard_cap : p->n) : 0; }

void b3m_plan_destroy(b3m_plan *p)
{
	if (!p) return;
	if (p->h_raw) { if (p->h_raw_pinned) (void)hipHostFree(p->h_raw); else free(p->h_raw); }     // (a lazy table's image is malloc'ed: h_raw_pinned stays false)
	if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
	if (p->d_pack) (void)hipFree(p->d_pack);
	if (p->d_isa_cache) (void)hipFree(p->d_isa_cache);
	blocksort_release(p->bs);
	arena_free(p->arena, arena_vmm_of(p)); arena_vmm_drop(p);
	delete p;
}
uint64_t b3m_plan_num_blocks(const b3m_plan *p) { return p ? p->leaves.size() : 0; }
/* 0, or the number of sequential suffix-range shards of a plan in the HOST TIER: a text below 2^32 symbols whose memory target no merge tree can meet.  Such a
 * plan keeps the text and ONE shard's sort in HBM; b3m_plan_run_host sends every slice of the BWT to (pinned) host memory as its shard finishes. */
int b3m_plan_host_tier(const b3m_plan *p) { return p && p->host_tier ? p->seq_shards : 0; }
const char *b3m_plan_warning(const b3m_plan *p) { return p ? p->warning.c_str() : ""; }
uint64_t b3m_plan_workspace_bytes(const b3m_plan *p) { return p ? p->ws_bytes : 0; }
uint64_t b3m_plan_text_bytes(const b3m_plan *p) { return p ? packed_text_bytes(p->n, p->type) : 0; }
void *b3m_plan_text_buffer(b3m_plan *p) { return p ? (void *)p->d_text : nullptr; }

namespace {
// BWT symbols -> the .pac layout (4 symbols per byte, the first in the top bits); sub = 1 for pacterm (symbols 1..4 -> 0..3, the
// terminator's slot reads 0).  One thread per 16 symbols = 4 output bytes.
__global__ void k_pack_pac(const uint8_t *bwt, uint64_t w_first, uint64_t w_last, uint32_t sub, uint8_t *out, uint64_t n)
{
	const uint64_t w = w_first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= w_last) return;
	const uint4 v = reinterpret_cast<const uint4 *>(bwt)[w];
	const uint32_t in[4] = {v.x, v.y, v.z, v.w};
	uint32_t o = 0;
	for (int k = 0; k < 4; ++k) {
		uint32_t b = 0;
		for (int j = 0; j < 4; ++j) {
			uint32_t c = (in[k] >> (8 * j)) & 0xFFu;
			if (16 * w + 4 * k + j >= n) c = sub;                 // padding slots behind the last symbol read 0 whatever the buffer holds there
			c = c >= sub ? c - sub : 0u;
			b |= (c & 3u) << (6 - 2 * j);
		}
		o |= b << (8 * k);
	}
	reinterpret_cast<uint32_t *>(out)[w] = o;
}
struct HostSink {            // SliceHook context of a host-memory run
	b3m_plan *p; hipStream_t s; uint8_t *d_bwt; uint8_t *h_bwt; std::vector<hipEvent_t> evs; uint64_t copied = 0;
	// symbols [first, last) are final on the device: pack them if asked for, then send them off on the copy stream
	static void slice(void *ctx, uint64_t first, uint64_t last)
	{
		HostSink *k = (HostSink *)ctx;
		if (last <= first) return;
		b3m_plan *p = k->p;
		uint64_t b0 = first, b1 = last;                 // byte range that leaves
		const uint8_t *src = k->d_bwt;
		if (p->packed_out) {
			// whole bytes of four symbols; the byte that straddles `first` was left to this slice by the one before
			b0 = first / 4; b1 = last == p->n ? (p->n + 3) / 4 : last / 4;
			if (b1 > b0) hipLaunchKernelGGL(k_pack_pac, dim3((uint32_t)(((b1 + 3) / 4 - b0 / 4 + 255) / 256)), dim3(256), 0, k->s, k->d_bwt, b0 / 4, (b1 + 3) / 4,
			                                p->type == B3M_INPUT_PACTERM ? 1u : 0u, p->d_pack, p->n);
			src = p->d_pack;
		}
		hipEvent_t e = p->tl.get();
		B3M_HIP(hipEventRecord(e, k->s));
		B3M_HIP(hipStreamWaitEvent(p->copy_stream, e, 0));
		if (b1 > b0) B3M_HIP(hipMemcpyAsync(k->h_bwt + b0, src + b0, b1 - b0, hipMemcpyDeviceToHost, p->copy_stream));
		k->evs.push_back(e); k->copied += last - first;
	}
};
}

// ---- sequential suffix-range shards and long repeats (row a6): the plan's list of what the shards left tied (b3m_plan::SparseTies) ----
__global__ void k_sp_export(const uint32_t *U, const uint32_t *gid, const uint32_t *sa, const uint8_t *sa_hi, uint64_t left, uint32_t *slot, uint32_t *pos_lo, uint8_t *pos_hi, uint8_t *head)
{
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= left) return;
	const uint32_t r = U[j];
	slot[j] = r; pos_lo[j] = sa[r]; pos_hi[j] = sa_hi ? sa_hi[r] : (uint8_t)0;
	head[j] = (j == 0 || gid[j] != gid[j - 1]) ? 1 : 0;
}
// the list as the rounds want it: U = every slot, hid = heads (max-scanned into group labels by the caller)
__global__ void k_sp_list(const uint8_t *head, uint64_t cnt, uint32_t *U, uint32_t *hid)
{
	uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e < cnt) { U[e] = (uint32_t)e; hid[e] = head[e] ? (uint32_t)e : 0u; }
}
__global__ void k_sp_positions(const uint32_t *lo, const uint8_t *hi, uint64_t cnt, uint64_t *out)
{
	uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e < cnt) out[e] = (uint64_t)lo[e] | ((uint64_t)hi[e] << 32);
}
__global__ void k_sp_gslots(const uint32_t *slot, uint64_t first, uint64_t count, uint64_t slice_off, uint64_t *out)
{
	uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e < count) out[first + e] = slice_off + slot[first + e];
}
// the finished list back into the result: BWT symbol at the suffix's global rank (into the BWT -- the part [first, first + len) of it that d_bwt holds -- or as
// (rank, symbol) for a result that has left for host memory), p0, the (rank,pos) sample
__global__ void k_sp_patch(TextView T, const uint32_t *idx, const uint8_t *idx_hi, const uint64_t *gslot, uint64_t cnt, uint8_t *bwt, uint64_t first, uint64_t len, uint8_t *sym_out,
                           uint64_t *p0pair, uint64_t *samples, uint64_t rate)
{
	uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= cnt) return;
	const uint64_t i = (uint64_t)idx[e] | (idx_hi ? (uint64_t)idx_hi[e] << 32 : 0ull), g = gslot[e];
	const uint8_t c = (uint8_t)sym_at(T, i ? i - 1 : T.n - 1);
	if (!bwt) sym_out[e] = c;
	if (g - first >= len) return;                  // another shard's slot
	if (bwt) bwt[g - first] = c;
	if (i == 0) { p0pair[0] = g; p0pair[1] = 0; }
	if (samples && i % rate == 0) { samples[2 * (i / rate)] = g; samples[2 * (i / rate) + 1] = i; }
}
// records of the lists that travel between shards (b3m_plan_list_export): (global rank | first of its group << 63, position)
constexpr uint64_t REC_HEAD = 1ull << 63;
__global__ void k_sp_records(const uint32_t *slot, const uint32_t *pos_lo, const uint8_t *pos_hi, const uint8_t *head, uint64_t cnt, uint64_t slice_off, uint64_t *rec)
{
	uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= cnt) return;
	rec[2 * e] = (slice_off + slot[e]) | ((head && head[e]) ? REC_HEAD : 0ull);
	rec[2 * e + 1] = (uint64_t)pos_lo[e] | ((uint64_t)pos_hi[e] << 32);
}
__global__ void k_sp_from_records(const uint64_t *rec, uint64_t cnt, uint32_t *idx, uint8_t *idx_hi, uint32_t *U, uint32_t *hid, uint64_t *gslot)
{
	uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= cnt) return;
	const uint64_t r = rec[2 * e], q = rec[2 * e + 1];
	idx[e] = (uint32_t)q; if (idx_hi) idx_hi[e] = (uint8_t)(q >> 32);
	U[e] = (uint32_t)e; hid[e] = (e == 0 || (r & REC_HEAD)) ? (uint32_t)e : 0u;
	gslot[e] = r & ~REC_HEAD;
}
__global__ void k_sp_placed_from_records(const uint64_t *rec, uint64_t cnt, uint64_t *xpos, uint64_t *xrank)
{
	uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e < cnt) { xrank[e] = rec[2 * e] & ~REC_HEAD; xpos[e] = rec[2 * e + 1]; }
}
__global__ void k_sp_tied_positions(const uint32_t *idx, const uint8_t *idx_hi, uint64_t cnt, uint64_t *out)
{
	uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e < cnt) out[e] = (uint64_t)idx[e] | (idx_hi ? (uint64_t)idx_hi[e] << 32 : 0ull);
}
// The finish of a list of Tn tied and Ex placed suffixes (of one plan's sequential shards, or of all shards of a run over several GPUs), in the buffers of the plan's sort.
//   load_tied(idx, idx_hi, U, hid, gslot): positions of the listed suffixes in rank order, U = every slot, hid[e] = e at the first of a group else 0, gslot = global ranks
//   load_placed(xpos, xrank): the placed ones, any order
//   d_bwt: the part [bwt_first, bwt_first + bwt_len) of the BWT on the device, or null: the symbols go to h_bwt (the whole BWT in host memory, every slice arrived)
struct SparseBufs { uint32_t *idx; uint8_t *idx_hi; uint32_t *U, *hid; uint64_t *gslot, *xpos, *xrank; };
typedef std::function<void(const SparseBufs &)> SparseLoad;
static void sparse_finish_core(b3m_plan *p, const TextView &T, uint64_t Tn, uint64_t Ex, bool complete, uint64_t dmin, uint64_t dmax, const SparseLoad &load_tied, const SparseLoad &load_placed,
                               uint8_t *d_bwt, uint64_t bwt_first, uint64_t bwt_len, uint8_t *h_bwt, hipStream_t s)
{
	if (Tn == 0) return;
	if (!complete) Ex = 0;
	const uint64_t M = Tn + Ex;
	BlockSortWs &ws = p->bs;
	const bool big = p->n >= (1ull << 32);
	const int dbg = (int)sw().DEBUG.get(0);
	blocksort_ensure_ties(ws, Tn);
	uint32_t *Bf[6] = {ws.t[0], ws.t[1], ws.idxB, ws.t[2], ws.t[3], ws.t[4]};
	if (big) { Bf[0] = ws.t[1]; Bf[1] = ws.t[2]; Bf[2] = ws.t[3]; Bf[3] = ws.t[4]; Bf[4] = ws.t[5]; Bf[5] = ws.idxB; }
	B3M_REQUIRE(Tn <= (big ? ws.tie_cap : ws.cap) && 2 * Tn + 3 * M + 64 <= ws.cap, "the list of tied suffixes (" + std::to_string(Tn) + " + " + std::to_string(Ex) +
	            ") is larger than the buffers of its finish (a shard's sort of " + std::to_string(ws.cap) + " suffixes)");
	uint32_t *idx = ws.idxA; uint8_t *idx_hi = big ? ws.sa_hi : nullptr, *tmp_hi = big ? ws.tie_hi : nullptr;
	// behind the rounds' keys (Tn of each record buffer): global ranks of the list's slots, the positions to look ranks up by, their ranks
	uint64_t *gslot = ws.keyA + Tn + 8, *posS = gslot + Tn + 8, *rank = posS + M + 8;         // keyA: Tn + Tn + M + Ex + M (+ pads)
	uint64_t *ptmp = ws.keyB + Tn + 8;                                                       // keyB: Tn + M
	uint32_t *d_missing = reinterpret_cast<uint32_t *>(ptmp + M + 8);
	load_tied(SparseBufs{idx, idx_hi, Bf[0], Bf[1], gslot, nullptr, nullptr});
	inclusive_max_u32(Bf[1], Tn, ws.scan_ws, s);
	const uint64_t K = (uint64_t)key_symbols(T.type), X = dmax;
	uint64_t left = 0, rounds = 0;
	if (!complete) {
		// a shard finished its own repeats (no room on the list) or placed more suffixes among long repeats than the list holds: the targets of a doubling round
		// could be missing, so the list is finished as a shard would have finished it, by key extension to the end
		rounds = tie_rounds(T, idx, Tn, 0, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, Tn, s, &left, 0, nullptr, 0, idx_hi, tmp_hi, dmin, false, true);
	} else {
		// key extension until every group agrees on 2X symbols and more, then doubling on the list (tie_rounds_sparse)
		const uint64_t rt0 = (2 * X + 1 - dmin + K - 1) / K;
		rounds = tie_rounds(T, idx, Tn, 0, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, Tn, s, &left, rt0, nullptr, 0, idx_hi, tmp_hi, dmin, false, true);
		if (left && rounds >= rt0) {
			hipLaunchKernelGGL(k_sp_tied_positions, dim3((uint32_t)((Tn + 255) / 256)), dim3(256), 0, s, (const uint32_t *)idx, (const uint8_t *)idx_hi, Tn, posS);
			// the placed ones: positions behind the list's in the table to be, a copy of them and their global ranks beside it until the table is in order
			uint64_t *xpos = ptmp;
			if (Ex) {
				load_placed(SparseBufs{nullptr, nullptr, nullptr, nullptr, nullptr, xpos, rank});
				B3M_HIP(hipMemcpyAsync(posS + Tn, xpos, 8 * Ex, hipMemcpyDeviceToDevice, s));
			}
			int nbits = 1; while ((1ull << nbits) < p->n + 1 && nbits < 64) ++nbits;
			uint64_t *sorted = nullptr;
			uint64_t *stmp = rank + Ex + 8;         // (M entries: inside keyA's 3 M behind gslot)
			radix_sort_u64(posS, nullptr, stmp, nullptr, M, nbits, ws.radix, s, nullptr, &sorted, nullptr);
			uint64_t *rtab = sorted == posS ? stmp : posS;
			sparse_rank_placed(sorted, M, xpos, rank, Ex, rtab, s);
			uint32_t missing = 0;
			const uint64_t d0 = dmin + rt0 * K;
			if (dbg >= 2) fprintf(stderr, "[b3m] list finish: %llu tied + %llu placed entries, %llu key rounds to %llu symbols, %llu left for doubling (threshold %llu)\n",
			                      (unsigned long long)Tn, (unsigned long long)Ex, (unsigned long long)rounds, (unsigned long long)d0, (unsigned long long)left, (unsigned long long)X);
			rounds += tie_rounds_sparse(T, idx, idx_hi, tmp_hi, Tn, left, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, sorted, rtab, gslot, M, d_missing, d0, X, s, &left, &missing);
			B3M_REQUIRE(missing == 0, "internal: prefix doubling on the list of tied suffixes met " + std::to_string(missing) + " suffixes that are not on it");
		}
	}
	p->res.tie_rounds += rounds;
	// symbols, p0 and samples of everything that was on the list
	uint8_t *sym = reinterpret_cast<uint8_t *>(ptmp);
	uint64_t *smp = (p->isarate && p->total_samples) ? p->d_samples : nullptr;
	hipLaunchKernelGGL(k_sp_patch, dim3((uint32_t)((Tn + 255) / 256)), dim3(256), 0, s, T, (const uint32_t *)idx, (const uint8_t *)idx_hi, (const uint64_t *)gslot, Tn, d_bwt, bwt_first, bwt_len, sym,
	                   p->d_p0pairs, smp, p->isarate ? p->isarate : 1);
	if (!d_bwt) {
		std::vector<uint64_t> hg(Tn); std::vector<uint8_t> hs(Tn);
		B3M_HIP(hipMemcpyAsync(hg.data(), gslot, 8 * Tn, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipMemcpyAsync(hs.data(), sym, Tn, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipStreamSynchronize(s));
		B3M_HIP(hipStreamSynchronize(p->copy_stream));            // every slice has arrived
		for (uint64_t e = 0; e < Tn; ++e) h_bwt[hg[e]] = hs[e];
	}
}
/* the list of ONE plan's sequential shards; d_bwt: the whole BWT on the device (a device-resident run), or null: the symbols go to h_bwt */
static void sparse_tie_finish(b3m_plan *p, const TextView &T, uint8_t *d_bwt, uint8_t *h_bwt, hipStream_t s)
{
	b3m_plan::SparseTies &L = p->sp;
	const uint64_t Tn = L.count, Ex = L.xcount;
	auto load_tied = [&](const SparseBufs &b) {
		B3M_HIP(hipMemcpyAsync(b.idx, L.pos_lo, 4 * Tn, hipMemcpyDeviceToDevice, s));
		if (b.idx_hi) B3M_HIP(hipMemcpyAsync(b.idx_hi, L.pos_hi, Tn, hipMemcpyDeviceToDevice, s));
		hipLaunchKernelGGL(k_sp_list, dim3((uint32_t)((Tn + 255) / 256)), dim3(256), 0, s, (const uint8_t *)L.head, Tn, b.U, b.hid);
		for (const auto &pt : L.parts)
			if (pt.count) hipLaunchKernelGGL(k_sp_gslots, dim3((uint32_t)((pt.count + 255) / 256)), dim3(256), 0, s, (const uint32_t *)L.slot, pt.first, pt.count, pt.slice_off, b.gslot);
	};
	auto load_placed = [&](const SparseBufs &b) {
		hipLaunchKernelGGL(k_sp_positions, dim3((uint32_t)((Ex + 255) / 256)), dim3(256), 0, s, (const uint32_t *)L.xpos_lo, (const uint8_t *)L.xpos_hi, Ex, b.xpos);
		for (const auto &pt : L.parts)
			if (pt.xcount) hipLaunchKernelGGL(k_sp_gslots, dim3((uint32_t)((pt.xcount + 255) / 256)), dim3(256), 0, s, (const uint32_t *)L.xslot, pt.xfirst, pt.xcount, pt.slice_off, b.xrank);
	};
	sparse_finish_core(p, T, Tn, Ex, L.complete, L.dmin, L.dmax, load_tied, load_placed, d_bwt, 0, p->n, h_bwt, s);
}

// the run's samples as a host list, in position order (slot k of the pinned image holds position k * rate); a shard keeps only the
// samples of its own suffixes (the other slots still read "empty"; ranks are global already)
static const std::vector<uint64_t> &samples_of(b3m_plan *p)
{
	if (p->samples_pending) {
		p->h_samples.clear();
		if (p->samples_lazy && p->raw_stale) {
			if (!p->h_raw) { p->h_raw = (uint64_t *)malloc(16 * p->total_samples); B3M_REQUIRE(p->h_raw != nullptr, "out of host memory for the (rank,pos) samples"); }
			B3M_HIP(hipMemcpy(p->h_raw, p->d_samples, 16 * p->total_samples, hipMemcpyDeviceToHost));
			p->raw_stale = false;
		}
		const uint64_t *raw = p->h_raw;
		if (p->nshards > 1) {
			for (uint64_t i = 0; i < p->total_samples; ++i)
				if (raw[2 * i] != ~0ull) { p->h_samples.push_back(raw[2 * i]); p->h_samples.push_back(raw[2 * i + 1]); }
		} else {
			// every multiple of the rate is a suffix of the text: a slot that still reads "empty" means a sort kernel lost a sample
			for (uint64_t i = 0; i < p->total_samples; ++i)
				B3M_REQUIRE(raw[2 * i] != ~0ull, "internal: the (rank,pos) sample of position " + std::to_string(i * p->isarate) + " was not written by the sort");
			p->h_samples.assign(raw, raw + 2 * p->total_samples);
		}
		p->res.num_samples = p->h_samples.size() / 2;
		p->samples_pending = false;
	}
	return p->h_samples;
}

static int plan_run_impl(b3m_plan *p, const void *d_text_in, const void *h_text_in, uint8_t *d_bwt_out, uint8_t *h_bwt_out, void *stream)
{
	B3M_TRY
	switches_load();
	const bool host = h_text_in != nullptr;
	B3M_REQUIRE(p && (host ? (h_bwt_out != nullptr) : (d_text_in && d_bwt_out)), "b3m_plan_run: null argument");
	hipStream_t s = (hipStream_t)stream;
	hipEvent_t evh0 = nullptr, evh1 = nullptr, evh2 = nullptr, evh3 = nullptr;
	HostSink sink{p, s, nullptr, h_bwt_out, {}, 0};
	SliceHook hook{8, &HostSink::slice, &sink};
	// whatever way this function is left: the sort workspace must not keep pointers into this stack frame, and no copy of a
	// host-memory run may still be in flight into the caller's buffers
	struct Guard {
		b3m_plan *p; bool host;
		~Guard()
		{
			p->bs.slice_hook = nullptr; p->bs.arrival = nullptr; p->bs.samples = nullptr; p->bs.d_symhist_out = nullptr;
			if (host && p->copy_stream && std::uncaught_exceptions()) (void)hipStreamSynchronize(p->copy_stream);
		}
	} guard{p, h_text_in != nullptr};
	double wall0 = 0;
	if (host) {
		B3M_REQUIRE(p->nshards == 1, "host-memory runs need an unsharded plan");
		B3M_REQUIRE(p->type != B3M_INPUT_SYMBOLS32, "host-memory runs: symbols32 texts are device-resident only");
		if (!p->copy_stream) B3M_HIP(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
		d_bwt_out = p->d_bwt[0];                    // one block: the leaf sort writes here; several: the last level is copied here
		sink.d_bwt = d_bwt_out;
		B3M_REQUIRE(!(p->seq_shards > 1 && p->packed_out), "host-memory runs of sequential shards hand the BWT over one byte per symbol");
		if (p->packed_out && !p->d_pack) B3M_HIP(hipMalloc((void **)&p->d_pack, (((p->n + 3) / 4 + 15) & ~15ull) + 16));
		p->e2e = b3m_e2e_times{};
	}
	const uint64_t n = p->n, B = p->leaves.size();
	Timeline *tl = &p->tl;
	tl->reset(); tl->stream = s; tl->enabled = !sw().NO_TIMELINE.on();      // (measurement switch: no per-family HIP events)
	hipEvent_t ev0 = tl->get(), ev1 = tl->get();
	TextArrival arrival{}; bool streaming_text = false;
	if (host) {
		evh0 = tl->get(); evh1 = tl->get(); evh2 = tl->get(); evh3 = tl->get();
		B3M_HIP(hipStreamSynchronize(s));
		wall0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
		B3M_HIP(hipEventRecord(evh0, s));
	}
	B3M_HIP(hipEventRecord(ev0, s));
	// resident, padded copy of the packed text
	uint64_t tb = packed_text_bytes(n, p->type);
	if (!host && d_text_in == (const void *)p->d_text) {
		// the caller loaded the text into the plan's own buffer (b3m_plan_text_buffer): used in place, only the padding is cleared
		B3M_HIP(hipMemsetAsync(p->d_text + tb, 0, p->text_cap - tb, s));
	} else {
		if (host) {
			// the text arrives in pieces on the copy stream; a single-block sort starts its (chunk-local) level 0 behind the first piece,
			// everything else waits for the last one (TextArrival)
			hipEvent_t e0 = tl->get();
			B3M_HIP(hipEventRecord(e0, s));
			B3M_HIP(hipStreamWaitEvent(p->copy_stream, e0, 0));
			tl->pool.push_back(e0);
			B3M_HIP(hipMemsetAsync(p->d_text + (tb & ~63ull), 0, p->text_cap - (tb & ~63ull), p->copy_stream));
			const int np = tb >= (64ull << 20) ? 8 : 1;
			arrival.n = np;
			for (int i = 0; i < np; ++i) {
				const uint64_t a0 = (tb * (uint64_t)i / np) & ~255ull, a1 = i + 1 == np ? tb : (tb * (uint64_t)(i + 1) / np) & ~255ull;
				B3M_HIP(hipMemcpyAsync(p->d_text + a0, (const uint8_t *)h_text_in + a0, a1 - a0, hipMemcpyHostToDevice, p->copy_stream));
				arrival.ev[i] = i + 1 == np ? evh1 : tl->get();
				B3M_HIP(hipEventRecord(arrival.ev[i], p->copy_stream));
				arrival.bytes_ready[i] = a1;
			}
			// (pacterm only: in a circular pac text the BWT symbol of position 0 is the text's LAST symbol, which the first tile would read before it has arrived)
			streaming_text = B == 1 && p->nshards == 1 && p->seq_shards <= 1 && p->type == B3M_INPUT_PACTERM;
			if (!streaming_text) B3M_HIP(hipStreamWaitEvent(s, evh1, 0));
		} else {
			B3M_HIP(hipMemsetAsync(p->d_text + (tb & ~63ull), 0, p->text_cap - (tb & ~63ull), s));
			B3M_HIP(hipMemcpyAsync(p->d_text, d_text_in, tb, hipMemcpyDeviceToDevice, s));
		}
	}
	TextView T{p->d_text, n, num_bases(n, p->type), p->type};
	B3M_HIP(hipMemsetAsync(p->d_hist, 0, 8 * 256 * B, s));
	B3M_HIP(hipMemsetAsync(p->d_anchors, 0, 4 * (p->total_anchors + 1), s));
	B3M_HIP(hipMemsetAsync(p->d_endrank, 0, 4 * B, s));
	B3M_HIP(hipMemsetAsync(p->d_sample_cnt, 0, 4 * (B + 1), s));
	B3M_HIP(hipMemsetAsync(p->d_p0pairs, 0xFF, 16 * B, s));                // "p0 not seen" until a sort finds the block's first suffix
	if (B > 1) B3M_HIP(hipMemsetAsync(p->d_gt[0], 0, 4 * p->gt_words, s));
	p->res = b3m_result{};
	p->res.n = n; p->res.num_blocks = B; p->res.levels = p->levels;
	// ---- leaves (row a7) ----
	for (uint64_t k = 0; k < B; ++k) {
		const Leaf &lf = p->leaves[k];
		BlockSortOut so;
		// (rank,pos) samples of a whole-text block (or its shards) are written by the sort itself, where each suffix is finally placed
		// (SampleOut): no suffix array is kept for them.  Blocks of a merge tree keep theirs anyway (anchors), the LSD path always has one; since round 5 their record
		// sort writes the samples too (leaf-local ranks, carried through the merges as before) instead of a pass over the leaf's suffix array afterwards
		// (k_emit_samples: 3.9 ms per step of C3 in 8 blocks); a leaf that is sorted again with an overhang (long repeats) still takes that pass.
		const bool smp_in_kernel = p->isarate != 0 && sample_in_kernel_ok(p->isarate, n);
		const bool need_sa = B > 1 || p->keep_sa || p->sa_hint || (p->isarate != 0 && !smp_in_kernel);       // anchors / the tie export / samples the kernels cannot test for read the suffix array
		SampleOut smp_cfg;
		if (smp_in_kernel) smp_cfg = make_sample_out(p->d_samples + 2 * p->sample_off[k], p->isarate, (lf.a + p->isarate - 1) / p->isarate, p->sample_off[k + 1] - p->sample_off[k]);
		const int nsh = p->seq_shards > 1 ? p->seq_shards : p->nshards;
		uint64_t acc = 0;                                    // suffixes sorted by the earlier shards of this run = global rank of this shard's first
		// slots of other shards stay "empty"; samples written by the sort itself: a slot nobody wrote is found by samples_of
		if (p->isarate && (nsh > 1 || smp_in_kernel) && k == 0) B3M_HIP(hipMemsetAsync(p->d_samples, 0xFF, 16 * (p->total_samples + 1), s));
		const bool seq_host = host && p->seq_shards > 1;         // the slices leave for host memory one by one: two slice buffers in d_bwt[0]
		std::vector<hipEvent_t> gone;                            // gone[sh]: slice sh has left its buffer
		p->bs.no_defer = p->seq_shards > 1 || (p->nshards > 1 && p->tie_mode != B3M_TIES_SA);
		// what the shards leave tied goes onto the plan's list (SparseTies) and is finished when the last shard is through
		b3m_plan::SparseTies &L = p->sp;
		L.parts.clear(); L.count = L.xcount = 0; L.dmin = ~0ull; L.dmax = 0; L.complete = true;
		const bool sparse = (p->seq_shards > 1 || (p->nshards > 1 && p->tie_mode == B3M_TIES_LIST)) && L.cap > 0 && B == 1;
		if (sparse) B3M_HIP(hipMemsetAsync(L.xcnt, 0, 4, s));
		for (int sh = (p->seq_shards > 1 ? 0 : p->shard); sh < (p->seq_shards > 1 ? p->seq_shards : p->shard + 1); ++sh) {
			uint8_t *slice_dst = seq_host ? p->d_bwt[0] + (uint64_t)(sh & 1) * p->seq_half : nullptr;
			if (seq_host && sh >= 2) B3M_HIP(hipStreamWaitEvent(s, gone[(size_t)sh - 2], 0));
			B3M_HIP(hipMemsetAsync(p->d_p0tmp + k, 0xFF, 4, s));
			// one block: its BWT is the result and goes straight to the caller's buffer
			p->bs.d_symhist_out = (sh == 0 || p->seq_shards == 1) ? p->d_hist + 256 * k : nullptr;      // the histogram of the whole block falls out of every shard's level-0 totals: taken once
			p->bs.slice_hook = (host && B == 1 && nsh == 1) ? &hook : nullptr;
			p->bs.arrival = streaming_text ? &arrival : nullptr;
			p->bs.samples = smp_in_kernel ? &smp_cfg : nullptr;
			p->bs.sparse_room = sparse ? L.cap - L.count : 0;
			p->bs.extras = TieExtras{};
			if (sparse) { p->bs.extras.cnt = L.xcnt; p->bs.extras.slot = L.xslot; p->bs.extras.pos_lo = L.xpos_lo; p->bs.extras.pos_hi = L.xpos_hi; p->bs.extras.cap = (uint32_t)L.cap; }
			block_sort(T, lf.a, lf.m, p->bs, s, tl, so, seq_host ? slice_dst : (B == 1 ? d_bwt_out + acc : p->d_bwt[0] + lf.a), p->d_p0tmp + k, sh, nsh, need_sa);
			if (seq_host) {
				// the shard's slice is final: off to host memory on the copy stream, behind the sort of the next shard
				B3M_REQUIRE(so.kept_len <= p->seq_half, "internal: a sequential shard's slice is larger than its buffer");
				hipEvent_t done = tl->get(), left = tl->get();
				B3M_HIP(hipEventRecord(done, s));
				B3M_HIP(hipStreamWaitEvent(p->copy_stream, done, 0));
				if (so.kept_len) B3M_HIP(hipMemcpyAsync(h_bwt_out + acc, slice_dst, so.kept_len, hipMemcpyDeviceToHost, p->copy_stream));
				B3M_HIP(hipEventRecord(left, p->copy_stream));
				gone.push_back(left); sink.evs.push_back(done); sink.evs.push_back(left);
				sink.copied += so.kept_len;
			}
			p->bs.d_symhist_out = nullptr; p->bs.slice_hook = nullptr; p->bs.arrival = nullptr; p->bs.samples = nullptr;
			if (host && B == 1 && so.late_writes && !seq_host) sink.copied = 0;       // slices were rewritten after they left: everything is copied again below
			if (p->seq_shards > 1) B3M_REQUIRE(so.slice_offset == acc, "internal: suffix-range shards do not tile the rank space");
			if (so.reran_with_sa) p->sa_hint = 1;
			p->tie_left = so.tie_left; p->tie_U = so.tie_U; p->tie_gid = so.tie_gid;
			p->d_sa_slice = so.d_sa; p->slice_off = so.slice_offset; p->slice_len = so.kept_len;
			if (p->seq_shards > 1 || sparse) {
				B3M_REQUIRE(!so.tie_left || (sparse && L.count + so.tie_left <= L.cap), "internal: deferred ties in a sequential shard");
				if (sparse) {
					uint32_t xc = 0;
					B3M_HIP(hipMemcpyAsync(&xc, L.xcnt, 4, hipMemcpyDeviceToHost, s));
					B3M_HIP(hipStreamSynchronize(s));
					if (xc > L.cap) { L.complete = false; xc = (uint32_t)L.cap; B3M_HIP(hipMemcpyAsync(L.xcnt, &xc, 4, hipMemcpyHostToDevice, s)); B3M_HIP(hipStreamSynchronize(s)); }
					if (so.tie_refused) L.complete = false;
					if (so.tie_left) {
						hipLaunchKernelGGL(k_sp_export, dim3((uint32_t)((so.tie_left + 255) / 256)), dim3(256), 0, s, (const uint32_t *)so.tie_U, (const uint32_t *)so.tie_gid, (const uint32_t *)p->bs.idxA,
						                   (const uint8_t *)(n >= (1ull << 32) ? p->bs.sa_hi : nullptr), so.tie_left, L.slot + L.count, L.pos_lo + L.count, L.pos_hi + L.count, L.head + L.count);
						L.dmin = std::min(L.dmin, so.tie_depth); L.dmax = std::max(L.dmax, so.tie_depth);
					}
					L.parts.push_back({L.count, so.tie_left, L.xcount, (uint64_t)xc - L.xcount, so.slice_offset});
					L.count += so.tie_left; L.xcount = xc;
				}
				so.tie_left = 0; p->tie_left = 0;
			}
			p->res.tie_rounds = std::max(p->res.tie_rounds, so.tie_rounds);
			p->res.tie_elements += so.tie_elements;
			p->res.tie_overhang = std::max<uint64_t>(p->res.tie_overhang, so.overhang);
			p->res.tie_slow += so.slow_ties;
			Scope sc(tl, KF_EMIT, 6 * lf.m);
			hipLaunchKernelGGL(k_p0_to_pair, dim3(1), dim3(64), 0, s, p->d_p0tmp + k, p->d_p0pairs + 2 * k, lf.a, so.slice_offset);
			if (!so.symhist_done && (sh == 0 || p->seq_shards == 1) && p->type != B3M_INPUT_SYMBOLS32) histogram_symbols(T, lf.a, lf.m, p->d_hist + 256 * k, s);
			if (p->isarate && !so.samples_done) {
				B3M_REQUIRE(so.d_sa != nullptr || so.kept_len == 0, "internal: samples asked for, but the sort kept no suffix array");
				emit_samples(so.d_sa, so.d_sa_hi, so.kept_len, p->isarate, so.slice_offset, p->d_samples + 2 * p->sample_off[k], (lf.a + p->isarate - 1) / p->isarate,
				             p->sample_off[k + 1] - p->sample_off[k], s);
			}
			acc += so.kept_len;
		}
		if (p->seq_shards > 1) {
			B3M_REQUIRE(acc == lf.m, "internal: suffix-range shards do not cover the block"); p->res.slice_offset = 0; p->res.slice_len = acc;
			p->tie_left = 0;
			if (sparse && L.count) {
				Scope sc(tl, KF_TIES, 0);
				sparse_tie_finish(p, T, seq_host ? nullptr : d_bwt_out, h_bwt_out, s);
			}
		}
		else { p->res.slice_offset = so.slice_offset; p->res.slice_len = so.kept_len; }
		{
			if (B > 1 && p->type == B3M_INPUT_BYTESTREAM) hipLaunchKernelGGL(k_emit_gt_ptr, dim3((uint32_t)((lf.m + 255) / 256)), dim3(256), 0, s, so.d_sa, lf.m, lf.a, p->d_p0tmp + k, p->d_gt[0]);
			else if (B > 1) emit_gt_from_text(T, lf.a, lf.m, p->d_gt[0], s);
		}
		if (B > 1) {
			Scope sc(tl, KF_ANCHOR, 0);
			// end bit: [S_{a+m} > S_a], and the anchors of every merge that has this leaf on its left side: one launch of all the searches
			RankQueries Q;
			auto add = [&](const uint64_t *pos, uint32_t *acc, uint64_t c) {
				if (Q.n == 8) { rank_of_positions_multi(T, so.d_sa, lf.m, Q, s); Q.n = 0; }
				Q.pos[Q.n] = pos; Q.accum[Q.n] = acc; Q.count[Q.n] = c; ++Q.n;
			};
			add(p->d_endpos + k, p->d_endrank + k, 1);
			for (auto &M : p->merges)
				if ((int)k >= M.L.lo && (int)k < M.L.hi) add(p->d_anchor_pos + M.anchor_off, p->d_anchors + M.anchor_off, M.num_chains);
			rank_of_positions_multi(T, so.d_sa, lf.m, Q, s);
			hipLaunchKernelGGL(k_set_endbit, dim3(1), dim3(64), 0, s, p->d_endrank + k, p->d_p0pairs + 2 * k, p->d_gt[0], lf.a + lf.m);
		}
	}
	// leaf histograms -> host (row a4); also needed for the C arrays of the merges
	std::vector<uint64_t> hh(256 * B);
	B3M_HIP(hipMemcpyAsync(hh.data(), p->d_hist, 8 * 256 * B, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	std::fill(p->h_hist.begin(), p->h_hist.end(), 0);
	for (uint64_t k = 0; k < B; ++k) for (int c = 0; c < 256; ++c) p->h_hist[c] += hh[256 * k + c];
	// ---- merges (rows a8-a10) ----
	int cur = 0;
	for (int lv = 0; lv < p->levels; ++lv) {
		int nxt = cur ^ 1;
		B3M_HIP(hipMemcpyAsync(p->d_gt[nxt], p->d_gt[cur], 4 * p->gt_words, hipMemcpyDeviceToDevice, s));
		const auto &nodes = p->level_nodes[lv];
		if (nodes.size() & 1) {       // carried node
			const NodeT &c = nodes.back();
			B3M_HIP(hipMemcpyAsync(p->d_bwt[nxt] + c.a, p->d_bwt[cur] + c.a, c.m, hipMemcpyDeviceToDevice, s));
		}
		for (int id : p->level_merges[lv]) {
			const MergeT &M = p->merges[id];
			const uint64_t a = M.L.a, m = M.L.m, mr = M.R.m, Rbeg = a + m, Rend = Rbeg + mr;
			uint32_t *d_p0u = p->d_p0tmp + B;   // scratch u32
			RankDict rd{};
			{
				Scope sc(tl, KF_RANKBUILD, m + m / 3);
				// p0 of L lives on the device (remapped after every merge): no host round trip
				clear_bits(p->d_gt[nxt], Rbeg, Rend + 1, s);
				hipLaunchKernelGGL(k_merge_prologue, dim3(1), dim3(64), 0, s, p->d_p0pairs + 2 * M.L.lo, p->d_anchors + M.anchor_off,
				                   p->d_gt[nxt], Rend, d_p0u);
				uint64_t hl[256] = {0};
				for (int k = M.L.lo; k < M.L.hi; ++k) for (int c = 0; c < 256; ++c) hl[c] += hh[256 * k + c];
				if (p->type == B3M_INPUT_BYTESTREAM) {
					// byte alphabet: aligned copy of BWT(L) + count rows every 64 symbols + the 256 symbol starts of L's text
					uint8_t *lcopy = reinterpret_cast<uint8_t *>(p->d_rank_scratch);
					uint32_t *d_C = reinterpret_cast<uint32_t *>(lcopy + ((m + 255) & ~255ull)), *d_th = d_C + 256;
					uint32_t hC[256]; { uint64_t acc = 0; for (int c = 0; c < 256; ++c) { hC[c] = (uint32_t)acc; acc += hl[c]; } }
					B3M_HIP(hipMemcpyAsync(lcopy, p->d_bwt[cur] + a, m, hipMemcpyDeviceToDevice, s));
					B3M_HIP(hipMemsetAsync(lcopy + m, 0, ((m + 255) & ~255ull) - m, s));
					B3M_HIP(hipMemcpyAsync(d_C, hC, 1024, hipMemcpyHostToDevice, s));
					B3M_HIP(hipStreamSynchronize(s));                  // hC is on this stack frame
					byte_rank_build(lcopy, m, d_p0u, p->d_rank_blocks, d_th, s);
				} else
				rank_build(p->d_bwt[cur] + a, m, d_p0u, p->type, p->d_rank_blocks, p->d_rank_scratch, p->d_scan_ws, s);
				rd.blocks = p->d_rank_blocks; rd.m = m; rd.p0ptr = d_p0u;
				if (p->type == B3M_INPUT_PACTERM) { uint64_t acc = hl[0]; for (int c = 0; c < 4; ++c) { rd.C[c] = (uint32_t)acc; acc += hl[c + 1]; } }
				else { uint64_t acc = 0; for (int c = 0; c < 4; ++c) { rd.C[c] = (uint32_t)acc; acc += hl[c]; } }
			}
			{
				Scope sc(tl, KF_GAPWALK, mr * (64 + 4));
				if (p->type == B3M_INPUT_BYTESTREAM) {
					uint8_t *lcopy = reinterpret_cast<uint8_t *>(p->d_rank_scratch);
					uint32_t *d_C = reinterpret_cast<uint32_t *>(lcopy + ((m + 255) & ~255ull));
					gap_walk_bytes(T, p->d_rank_blocks, lcopy, d_p0u, d_C, a, m, mr, p->d_gt[cur], p->d_gt[nxt], p->d_anchors + M.anchor_off, M.num_chains, M.chain_len,
					               p->d_ranks[0], s);
				} else {
				GapWalkArgs G{};
				G.T = T; G.rd = rd; G.a = a; G.m = m; G.mr = mr;
				G.gt_cur = p->d_gt[cur]; G.gt_next = p->d_gt[nxt];
				G.anchors = p->d_anchors + M.anchor_off; G.num_chains = M.num_chains; G.chain_len = M.chain_len;
				G.ranks_out = p->d_ranks[0];
				gap_walk(G, s);
				}
			}
			uint32_t *sr = nullptr;
			{
				int nbits = 1; while ((1ull << nbits) <= m + 1 && nbits < 32) ++nbits;
				radix_sort_u32(p->d_ranks[0], p->d_ranks[1], M.slots, nbits, p->merge_radix, s, tl, &sr, true);
			}
			{
				Scope sc(tl, KF_INTERLEAVE, 4 * mr + 2 * (m + mr));
				interleave(sr, p->d_bwt[cur] + a, m, p->d_bwt[cur] + Rbeg, mr, p->d_bwt[nxt] + a, p->d_tile_q, s);
				// ranks of p0 entries and of the (rank,pos) samples move with the merge
				remap_ranks(sr, mr, p->d_p0pairs + 2 * M.L.lo, (uint64_t)(M.R.hi - M.L.lo), a, Rbeg, Rend, s);
				if (p->isarate) {
					uint64_t o0 = p->sample_off[M.L.lo], o1 = p->sample_off[M.R.hi];
					remap_ranks(sr, mr, p->d_samples + 2 * o0, o1 - o0, a, Rbeg, Rend, s);
				}
			}
		}
		cur = nxt;
	}
	if (B > 1 && !host) B3M_HIP(hipMemcpyAsync(d_bwt_out, p->d_bwt[cur], n, hipMemcpyDeviceToDevice, s));
	if (host) {
		B3M_HIP(hipEventRecord(evh2, s));                                   // construction done
		if (sink.copied != n) {                                                 // nothing streamed (several blocks, small input) or rewritten slices: one copy of the whole result
			B3M_HIP(hipStreamSynchronize(p->copy_stream));
			const uint8_t *src = B > 1 ? p->d_bwt[cur] : d_bwt_out;
			if (p->packed_out) {
				const uint64_t nb4 = (n + 3) / 4;
				hipLaunchKernelGGL(k_pack_pac, dim3((uint32_t)(((nb4 + 3) / 4 + 255) / 256)), dim3(256), 0, s, src, (uint64_t)0, (nb4 + 3) / 4, p->type == B3M_INPUT_PACTERM ? 1u : 0u, p->d_pack, p->n);
				B3M_HIP(hipMemcpyAsync(h_bwt_out, p->d_pack, nb4, hipMemcpyDeviceToHost, s));
			} else B3M_HIP(hipMemcpyAsync(h_bwt_out, src, n, hipMemcpyDeviceToHost, s));
			B3M_HIP(hipEventRecord(evh3, s));
		} else {
			B3M_HIP(hipEventRecord(evh3, p->copy_stream));
			B3M_HIP(hipStreamWaitEvent(s, evh3, 0));
		}
	}
	uint64_t p0pair[2] = {0, 0};
	B3M_HIP(hipMemcpyAsync(p0pair, p->d_p0pairs, 16, hipMemcpyDeviceToHost, s));
	// the samples leave on the same stream into pinned memory; the host-side list is made from them when somebody asks (samples_of)
	if (p->isarate && p->total_samples && !p->samples_lazy) B3M_HIP(hipMemcpyAsync(p->h_raw, p->d_samples, 16 * p->total_samples, hipMemcpyDeviceToHost, s));
	p->raw_stale = p->samples_lazy;
	B3M_HIP(hipEventRecord(ev1, s));
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	p->res.p0 = p0pair[0];                 // a global rank (k_p0_to_pair adds the shard's offset)
	p->res.has_p0 = 1;
	if (p0pair[0] == ~0ull) { p->res.has_p0 = 0; B3M_REQUIRE(p->nshards > 1, "internal: the block's first suffix was not found"); }   // it lies in another GPU's shard
	float ms = 0; (void)hipEventElapsedTime(&ms, ev0, ev1);
	p->res.ms_total = ms;
	if (host) {
		B3M_HIP(hipStreamSynchronize(p->copy_stream));
		const double wall1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
		float a = 0, b = 0, c = 0;
		(void)hipEventElapsedTime(&a, evh0, evh1); (void)hipEventElapsedTime(&b, evh1, evh2); (void)hipEventElapsedTime(&c, evh2, evh3);
		p->e2e.h2d_ms = a; p->e2e.construct_ms = b; p->e2e.d2h_tail_ms = c > 0 ? c : 0; p->e2e.total_ms = wall1 - wall0;
		p->e2e.streamed_bytes = sink.copied == n ? (p->packed_out ? (n + 3) / 4 : n) : 0;
		for (auto e : sink.evs) tl->pool.push_back(e);
		tl->pool.push_back(evh0); tl->pool.push_back(evh1); tl->pool.push_back(evh2); tl->pool.push_back(evh3);
		for (int i = 0; i + 1 < arrival.n; ++i) tl->pool.push_back(arrival.ev[i]);
	}
	tl->collect(p->kms, p->klaunch, p->kbytes);
	tl->pool.push_back(ev0); tl->pool.push_back(ev1);
	p->res.ms_sort = p->kms[KF_KEYS] + p->kms[KF_RADIX_HIST] + p->kms[KF_SCAN] + p->kms[KF_RADIX_SCATTER] + p->kms[KF_TIES] + p->kms[KF_EMIT]
	               + p->kms[KF_MSD_HIST] + p->kms[KF_MSD_SCATTER] + p->kms[KF_MSD_SCATTER_TEXT] + p->kms[KF_LDS_SORT];
	p->res.ms_gap = p->kms[KF_RANKBUILD] + p->kms[KF_ANCHOR] + p->kms[KF_GAPWALK];
	p->res.ms_interleave = p->kms[KF_INTERLEAVE];
	p->h_samples.clear();
	p->samples_pending = p->isarate && p->total_samples;
	p->res.num_samples = p->nshards > 1 ? 0 : p->total_samples;      // (a shard's own count: samples_of)
	return 0;
	B3M_CATCH(1)
}

int b3m_plan_run(b3m_plan *p, const void *d_text_in, uint8_t *d_bwt_out, void *stream)
{
	if (!d_text_in) { set_error("b3m_plan_run: null argument"); return 1; }
	return plan_run_impl(p, d_text_in, nullptr, d_bwt_out, nullptr, stream);
}
int b3m_plan_run_host(b3m_plan *p, const void *h_text, uint8_t *h_bwt_out, b3m_e2e_times *times, void *stream)
{
	if (!h_text) { set_error("b3m_plan_run_host: null argument"); return 1; }
	int rc = plan_run_impl(p, nullptr, h_text, nullptr, h_bwt_out, stream);
	if (rc == 0 && times) *times = p->e2e;
	return rc;
}
int b3m_plan_set_packed_output(b3m_plan *p, int on)
{
	B3M_TRY
	B3M_REQUIRE(p != nullptr, "b3m_plan_set_packed_output: null plan");
	B3M_REQUIRE(!on || p->type == B3M_INPUT_PAC || p->type == B3M_INPUT_PACTERM, "a 2-bit packed BWT needs a pac/pacterm input");
	p->packed_out = on ? 1 : 0;
	return 0;
	B3M_CATCH(1)
}
void *b3m_host_alloc(uint64_t bytes)
{
	void *q = nullptr;
	if (hipHostMalloc(&q, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc failed"); return nullptr; }
	return q;
}
void b3m_host_free(void *q) { if (q) (void)hipHostFree(q); }

/* compactstream input (libmaus2 CompactArray file, layout [UPSTREAM-RECALL], see include/b3m.h) */
int b3m_compact_info(const uint8_t *file_head32, uint64_t file_size, uint32_t *bits_out, uint64_t *n_out, uint64_t *data_off_out)
{
	B3M_TRY
	B3M_REQUIRE(file_head32 && file_size >= 32, "not a compact array file (shorter than its header)");
	auto be64 = [&](int k) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v = (v << 8) | file_head32[8 * k + i]; return v; };
	const uint64_t bits = be64(0), n = be64(1), nw = be64(2), nw2 = be64(3);
	B3M_REQUIRE(bits >= 1 && bits <= 8, "compactstream: 1..8 bits per symbol are supported");
	B3M_REQUIRE(nw == nw2 && nw <= (file_size - 32) / 8 && (n == 0 || (n * bits + 63) / 64 <= nw), "not a compact array file (word count does not fit)");
	if (bits_out) *bits_out = (uint32_t)bits;
	if (n_out) *n_out = n;
	if (data_off_out) *data_off_out = 32;
	return 0;
	B3M_CATCH(1)
}
/* d_words: the word stream (file image from data_off on) readable up to ceil(n*bits/8)+1 bytes; d_out: n bytes */
int b3m_compact_decode(const void *d_words, uint64_t n, uint32_t bits, uint8_t *d_out, void *stream)
{
	B3M_TRY
	require_device();
	B3M_REQUIRE(d_words && d_out && bits >= 1 && bits <= 8, "b3m_compact_decode: bad argument");
	compact_decode((const uint8_t *)d_words, n, bits, d_out, (hipStream_t)stream);
	B3M_HIP(hipGetLastError());
	return 0;
	B3M_CATCH(1)
}

/* ---- suffix-range shards and long repeats (row a6): a shard's local sort stops after a few key-extension rounds; what is still tied is
 * finished by prefix doubling over the suffix arrays of ALL shards, in O(log longest repeat) rounds -- the bound the reference gets per block
 * from lcpnext / largelcpthres (/root/reference/src/bwtb3m.cpp:55, ChangeLog:205-211).  Every rank runs the same rounds on the union list. ---- */
__global__ void k_tie_pairs(const uint32_t *U, const uint32_t *gid, uint64_t u, uint32_t off, uint32_t *pairs)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s < u) { pairs[2 * s] = U[s] + off; pairs[2 * s + 1] = gid[s] + off; }
}
__global__ void k_tie_unpairs(const uint32_t *pairs, uint64_t u, uint32_t *U, uint32_t *gid)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s < u) { U[s] = pairs[2 * s]; gid[s] = pairs[2 * s + 1]; }
}
__global__ void k_sub_u32(const uint32_t *pairs, uint64_t u, uint32_t off, uint32_t *out)
{
	uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s < u) out[s] = pairs[2 * s] - off;
}
int b3m_plan_set_keep_sa(b3m_plan *p, int on) { if (!p) { set_error("null argument"); return 1; } p->keep_sa = on ? 1 : 0; return 0; }
int b3m_plan_has_sa(const b3m_plan *p) { return p && p->d_sa_slice != nullptr; }
int b3m_plan_tie_pending(const b3m_plan *p, uint64_t *count_out)
{
	if (!p || !count_out) { set_error("null argument"); return 1; }
	*count_out = p->tie_left; return 0;
}
/* d_sa_slice_out: slice_len u32 (this shard's suffix array, rank order); d_pairs_out: 2 * pending u32 = (global slot, global group label) */
int b3m_plan_tie_export(b3m_plan *p, uint32_t *d_sa_slice_out, uint32_t *d_pairs_out, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && (d_sa_slice_out || p->slice_len == 0), "b3m_plan_tie_export: null argument");
	B3M_REQUIRE(p->nshards > 1 && p->n < (1ull << 32), "tie export: suffix-range shards of a text below 2^32 symbols only");
	hipStream_t s = (hipStream_t)stream;
	B3M_REQUIRE(p->d_sa_slice || p->slice_len == 0, "tie export: the shard kept no suffix array: b3m_plan_set_keep_sa(plan, 1) and run it again");
	if (p->slice_len) B3M_HIP(hipMemcpyAsync(d_sa_slice_out, p->d_sa_slice, 4 * p->slice_len, hipMemcpyDeviceToDevice, s));
	if (p->tie_left) {
		B3M_REQUIRE(d_pairs_out, "b3m_plan_tie_export: null argument");
		hipLaunchKernelGGL(k_tie_pairs, dim3((uint32_t)((p->tie_left + 255) / 256)), dim3(256), 0, s, p->tie_U, p->tie_gid, p->tie_left, (uint32_t)p->slice_off, d_pairs_out);
	}
	B3M_HIP(hipStreamSynchronize(s));
	return 0;
	B3M_CATCH(1)
}
/* d_sa_global: n u32, all shards' slices in rank order (changed in place); d_pairs_all: the exported pairs of all shards, concatenated in shard
 * order (total of them); my_first: index of this shard's first pair.  Repairs this shard's slice of the BWT in d_bwt_slice (+ p0, samples). */
int b3m_plan_tie_finish(b3m_plan *p, uint32_t *d_sa_global, const uint32_t *d_pairs_all, uint64_t total, uint64_t my_first, uint8_t *d_bwt_slice, void *stream)
{
	B3M_TRY
	switches_load();
	B3M_REQUIRE(p && d_sa_global && d_pairs_all && (d_bwt_slice || p->slice_len == 0), "b3m_plan_tie_finish: null argument");      // (an empty shard takes part with an empty slice)
	B3M_REQUIRE(p->nshards > 1 && p->n < (1ull << 32), "tie finish: suffix-range shards of a text below 2^32 symbols only");
	hipStream_t s = (hipStream_t)stream;
	const uint64_t n = p->n;
	BlockSortWs &ws = p->bs;
	B3M_REQUIRE(total <= ws.cap, "tie finish: more tied suffixes over all shards than one shard's workspace holds (low-complexity input)");
	if (total == 0) return 0;
	blocksort_ensure_ties(ws);
	TextView T{p->d_text, n, num_bases(n, p->type), p->type};
	if (!p->d_isa_cache) B3M_HIP(hipMalloc((void **)&p->d_isa_cache, 4 * n));
	uint32_t *const d_isa = p->d_isa_cache;
	isa_from_sa(d_sa_global, n, d_isa, s);
	uint32_t *Bf[6] = {ws.t[0], ws.t[1], ws.idxB, ws.t[2], ws.t[3], ws.t[4]};
	hipLaunchKernelGGL(k_tie_unpairs, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, d_pairs_all, total, Bf[0], Bf[1]);
	const uint64_t R0 = 4;
	uint64_t h0 = 1; while (2 * h0 <= R0 * (uint64_t)key_symbols(T.type)) h0 *= 2;
	uint64_t left = 0;
	const uint64_t rounds = tie_rounds(T, d_sa_global, total, 0, ws.keyA, ws.keyB, Bf, ws.scan_ws, ws.radix, n, s, &left, 0, d_isa, h0);
	// (left > 0 here means rotations that are EQUAL over all n symbols -- a circular pac text with a period: the doubling rounds only stop early when h >= n --
	// and any order of equal rotations gives the same BWT; a list above the workspace was refused above)
	p->res.tie_rounds += rounds;
	// my slice back into the workspace, the BWT symbols of the slots that were open here, p0, samples
	const uint64_t mine = p->tie_left;
	if (p->slice_len) B3M_HIP(hipMemcpyAsync(p->d_sa_slice, d_sa_global + p->slice_off, 4 * p->slice_len, hipMemcpyDeviceToDevice, s));
	if (mine) {
		uint32_t *Ul = ws.t[0];
		hipLaunchKernelGGL(k_sub_u32, dim3((uint32_t)((mine + 255) / 256)), dim3(256), 0, s, d_pairs_all + 2 * my_first, mine, (uint32_t)p->slice_off, Ul);
		B3M_HIP(hipMemsetAsync(p->d_p0tmp, 0xFF, 4, s));
		emit_bwt_list(T, Ul, mine, p->d_sa_slice, 0, d_bwt_slice, p->d_p0tmp, s);
		hipLaunchKernelGGL(k_p0_to_pair, dim3(1), dim3(64), 0, s, p->d_p0tmp, p->d_p0pairs, (uint64_t)0, p->slice_off);
		if (p->isarate && p->total_samples) {
			B3M_HIP(hipMemsetAsync(p->d_samples, 0xFF, 16 * (p->total_samples + 1), s));
			emit_samples(p->d_sa_slice, nullptr, p->slice_len, p->isarate, p->slice_off, p->d_samples, 0, p->total_samples, s);
		}
	}
	uint64_t p0pair[2] = {~0ull, 0};
	B3M_HIP(hipMemcpyAsync(p0pair, p->d_p0pairs, 16, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	if (p0pair[0] != ~0ull) { p->res.p0 = p0pair[0]; p->res.has_p0 = 1; }
	if (mine && p->isarate && p->total_samples) {
		if (p->samples_lazy) p->raw_stale = true;
		else B3M_HIP(hipMemcpy(p->h_raw, p->d_samples, 16 * p->total_samples, hipMemcpyDeviceToHost));
		p->samples_pending = true;
	}
	p->tie_left = 0;
	return 0;
	B3M_CATCH(1)
}

/* ---- the same bound from lists instead of suffix arrays (include/b3m.h: B3M_TIES_LIST) ---- */
int b3m_plan_set_tie_mode(b3m_plan *p, int mode)
{
	if (!p || mode < B3M_TIES_SA || mode > B3M_TIES_LOCAL) { set_error("b3m_plan_set_tie_mode: bad argument"); return 1; }
	if (p->nshards <= 1) { set_error("b3m_plan_set_tie_mode: not a suffix-range shard"); return 1; }
	if (mode == B3M_TIES_SA && p->n >= (1ull << 32)) { set_error("b3m_plan_set_tie_mode: texts of 2^32 symbols and more have no suffix-array form of the tie finish"); return 1; }
	p->tie_mode = mode; return 0;
}
int b3m_plan_list_counts(const b3m_plan *p, uint64_t *tied_out, uint64_t *placed_out, uint64_t *depth_out, int *complete_out)
{
	if (!p || !tied_out || !placed_out || !depth_out || !complete_out) { set_error("null argument"); return 1; }
	*tied_out = p->sp.count; *placed_out = p->sp.xcount; *depth_out = p->sp.count ? p->sp.dmax : 0; *complete_out = p->sp.complete ? 1 : 0;
	return 0;
}
/* does the union of all shards' lists fit the buffers this shard would finish it in (its sort's workspace)?  If not: B3M_TIES_LOCAL and run again */
int b3m_plan_list_fits(const b3m_plan *p, uint64_t tied_total, uint64_t placed_total)
{
	if (!p) return 0;
	const BlockSortWs &ws = p->bs;
	const bool big = p->n >= (1ull << 32);
	(void)big;      // (its list buffers grow on demand: blocksort_ensure_ties)
	return 2 * tied_total + 3 * (tied_total + placed_total) + 64 <= ws.cap;
}
int b3m_plan_list_export(b3m_plan *p, uint64_t *d_tied_out, uint64_t *d_placed_out, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && p->nshards > 1 && p->tie_mode == B3M_TIES_LIST, "b3m_plan_list_export: a suffix-range shard in list mode (b3m_plan_set_tie_mode) is needed");
	const b3m_plan::SparseTies &L = p->sp;
	B3M_REQUIRE((d_tied_out || !L.count) && (d_placed_out || !L.xcount), "b3m_plan_list_export: null argument");
	hipStream_t s = (hipStream_t)stream;
	if (L.count) hipLaunchKernelGGL(k_sp_records, dim3((uint32_t)((L.count + 255) / 256)), dim3(256), 0, s, (const uint32_t *)L.slot, (const uint32_t *)L.pos_lo, (const uint8_t *)L.pos_hi,
	                                (const uint8_t *)L.head, L.count, p->slice_off, d_tied_out);
	if (L.xcount) hipLaunchKernelGGL(k_sp_records, dim3((uint32_t)((L.xcount + 255) / 256)), dim3(256), 0, s, (const uint32_t *)L.xslot, (const uint32_t *)L.xpos_lo, (const uint8_t *)L.xpos_hi,
	                                 (const uint8_t *)nullptr, L.xcount, p->slice_off, d_placed_out);
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	return 0;
	B3M_CATCH(1)
}
/* d_tied_all: the tied records of ALL shards, concatenated in shard order (2 * tied_total u64); d_placed_all: their placed records, any order; depth_min / depth_max: over
 * the shards that reported tied suffixes; complete: every shard's list was.  Repairs this shard's slice of the BWT in d_bwt_slice (+ p0, samples). */
int b3m_plan_list_finish(b3m_plan *p, const uint64_t *d_tied_all, uint64_t tied_total, const uint64_t *d_placed_all, uint64_t placed_total, uint64_t depth_min, uint64_t depth_max,
                         int complete, uint8_t *d_bwt_slice, void *stream)
{
	B3M_TRY
	switches_load();
	B3M_REQUIRE(p && p->nshards > 1 && p->tie_mode == B3M_TIES_LIST, "b3m_plan_list_finish: a suffix-range shard in list mode (b3m_plan_set_tie_mode) is needed");
	B3M_REQUIRE((d_tied_all || !tied_total) && (d_placed_all || !placed_total) && (d_bwt_slice || p->slice_len == 0), "b3m_plan_list_finish: null argument");
	if (tied_total == 0) return 0;
	B3M_REQUIRE(depth_min >= 1 && depth_max >= depth_min, "b3m_plan_list_finish: bad depths");
	hipStream_t s = (hipStream_t)stream;
	const uint64_t n = p->n;
	TextView T{p->d_text, n, num_bases(n, p->type), p->type};
	auto load_tied = [&](const SparseBufs &b) {
		hipLaunchKernelGGL(k_sp_from_records, dim3((uint32_t)((tied_total + 255) / 256)), dim3(256), 0, s, d_tied_all, tied_total, b.idx, b.idx_hi, b.U, b.hid, b.gslot);
	};
	auto load_placed = [&](const SparseBufs &b) {
		hipLaunchKernelGGL(k_sp_placed_from_records, dim3((uint32_t)((placed_total + 255) / 256)), dim3(256), 0, s, d_placed_all, placed_total, b.xpos, b.xrank);
	};
	sparse_finish_core(p, T, tied_total, placed_total, complete != 0, depth_min, depth_max, load_tied, load_placed, d_bwt_slice ? d_bwt_slice : p->d_bwt[0], p->slice_off, p->slice_len, nullptr, s);
	uint64_t p0pair[2] = {~0ull, 0};
	B3M_HIP(hipMemcpyAsync(p0pair, p->d_p0pairs, 16, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	if (p0pair[0] != ~0ull) { p->res.p0 = p0pair[0]; p->res.has_p0 = 1; }
	if (p->isarate && p->total_samples) {
		if (p->samples_lazy) p->raw_stale = true;
		else B3M_HIP(hipMemcpy(p->h_raw, p->d_samples, 16 * p->total_samples, hipMemcpyDeviceToHost));
		p->samples_pending = true;
	}
	return 0;
	B3M_CATCH(1)
}

int b3m_plan_result(const b3m_plan *p, b3m_result *out)
{
	if (!p || !out) { set_error("null argument"); return 1; }
	B3M_TRY
	if (p->samples_pending && p->nshards > 1) (void)samples_of(const_cast<b3m_plan *>(p));      // a shard's sample count is known once its slots have been looked at
	*out = p->res; return 0;
	B3M_CATCH(1)
}
int b3m_plan_samples(const b3m_plan *p, uint64_t *pairs_out, uint64_t cap)
{
	if (!p || !pairs_out) { set_error("null argument"); return 1; }
	B3M_TRY
	const std::vector<uint64_t> &h = samples_of(const_cast<b3m_plan *>(p));
	uint64_t c = std::min<uint64_t>(cap, h.size() / 2);
	memcpy(pairs_out, h.data(), 16 * c);
	return 0;
	B3M_CATCH(1)
}
int b3m_plan_histogram(const b3m_plan *p, uint64_t *hist256)
{
	if (!p || !hist256) { set_error("null argument"); return 1; }
	if (p->type == B3M_INPUT_SYMBOLS32) { set_error("b3m_plan_histogram: a symbols32 text has no 256-entry histogram (count the symbols the caller ranked)"); return 1; }
	memcpy(hist256, p->h_hist.data(), 8 * 256);
	return 0;
}
int b3m_plan_kernel_times(const b3m_plan *p, b3m_kernel_time *out, int cap)
{
	if (!p || !out) { set_error("null argument"); return 1; }
	int k = 0;
	for (int i = 0; i < KF_COUNT && k < cap; ++i, ++k) { out[k].name = kfam_names[i]; out[k].ms = p->kms[i]; out[k].launches = p->klaunch[i]; out[k].alg_bytes = p->kbytes[i]; }
	return 0;
}

// ================================================================================================
// stage-level entry points
// ================================================================================================

int b3m_histogram(const void *d_text, uint64_t n, int inputtype, uint64_t start, uint64_t len, uint64_t *hist_out, void *stream)
{
	B3M_TRY
	switches_load();
	require_device();
	hipStream_t s = (hipStream_t)stream;
	B3M_REQUIRE(start + len <= n, "range out of bounds");
	PaddedText pt(d_text, n, inputtype, s);
	DevBuf h(8 * 256);
	B3M_HIP(hipMemsetAsync(h.p, 0, 8 * 256, s));
	histogram_symbols(pt.T, start, len, h.as<uint64_t>(), s);
	B3M_HIP(hipMemcpyAsync(hist_out, h.p, 8 * 256, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	return 0;
	B3M_CATCH(1)
}

int b3m_block_sort(const void *d_text, uint64_t n, int inputtype, uint64_t start, uint64_t len, uint32_t *d_sa_out,
                   uint8_t *d_bwt_out, uint32_t *d_gt_out, uint64_t *p0_out, void *stream)
{
	B3M_TRY
	switches_load();
	require_device();
	hipStream_t s = (hipStream_t)stream;
	B3M_REQUIRE(start + len <= n && len >= 1, "range out of bounds");
	PaddedText pt(d_text, n, inputtype, s);
	const uint64_t cap = len + std::min(std::min(block_overhang(), len / 8), n - (start + len));     // room for an overhang behind the block (tie_rounds_domain)
	DevBuf ws(blocksort_ws_bytes(cap));
	BlockSortWs bs; blocksort_ws_carve(bs, ws.p, cap);
	BlockSortOut so;
	DevBuf sc(64), bwt_tmp(d_bwt_out ? 16 : len);
	uint8_t *bw = d_bwt_out ? d_bwt_out : bwt_tmp.as<uint8_t>();
	block_sort(pt.T, start, len, bs, s, nullptr, so, bw, sc.as<uint32_t>());
	uint32_t p0 = 0;
	B3M_HIP(hipMemcpyAsync(&p0, sc.p, 4, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	if (p0_out) *p0_out = p0;
	if (d_sa_out) B3M_HIP(hipMemcpyAsync(d_sa_out, so.d_sa, 4 * len, hipMemcpyDeviceToDevice, s));
	if (d_gt_out) {
		// relative bit array: bit k = [S_{start+k} > S_start], k in [1,len]
		uint64_t words_abs = (n + 1 + 31) / 32 + 2;
		DevBuf gt(4 * words_abs);
		B3M_HIP(hipMemsetAsync(gt.p, 0, 4 * words_abs, s));
		emit_gt(so.d_sa, len, start, p0, gt.as<uint32_t>(), s);
		// end bit
		DevBuf q(8 + 8);
		uint64_t endpos = start + len;
		B3M_HIP(hipMemcpyAsync(q.p, &endpos, 8, hipMemcpyHostToDevice, s));
		B3M_HIP(hipMemsetAsync((uint8_t *)q.p + 8, 0, 8, s));
		rank_of_positions(pt.T, so.d_sa, len, q.as<uint64_t>(), 1, (uint32_t *)((uint8_t *)q.p + 8), s);
		uint32_t er = 0;
		B3M_HIP(hipMemcpyAsync(&er, (uint8_t *)q.p + 8, 4, hipMemcpyDeviceToHost, s));
		// shift absolute -> relative on the host (stage API is for tests; sizes are small)
		std::vector<uint32_t> habs(words_abs);
		B3M_HIP(hipMemcpyAsync(habs.data(), gt.p, 4 * words_abs, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipStreamSynchronize(s));
		uint64_t wrel = (len + 1 + 31) / 32;
		std::vector<uint32_t> hrel(wrel, 0);
		for (uint64_t k = 1; k < len; ++k) { uint64_t x = start + k; if ((habs[x >> 5] >> (x & 31)) & 1u) hrel[k >> 5] |= 1u << (k & 31); }
		if (er > p0) hrel[len >> 5] |= 1u << (len & 31);
		B3M_HIP(hipMemcpy(d_gt_out, hrel.data(), 4 * wrel, hipMemcpyHostToDevice));
	}
	B3M_HIP(hipStreamSynchronize(s));
	return 0;
	B3M_CATCH(1)
}

int b3m_rank_of(const void *d_text, uint64_t n, int inputtype, const uint32_t *d_sa, uint64_t len, const uint64_t *positions,
                uint64_t count, uint64_t *ranks_out, void *stream)
{
	B3M_TRY
	switches_load();
	require_device();
	hipStream_t s = (hipStream_t)stream;
	PaddedText pt(d_text, n, inputtype, s);
	DevBuf pos(8 * count), rk(4 * count);
	B3M_HIP(hipMemcpyAsync(pos.p, positions, 8 * count, hipMemcpyHostToDevice, s));
	B3M_HIP(hipMemsetAsync(rk.p, 0, 4 * count, s));
	rank_of_positions(pt.T, d_sa, len, pos.as<uint64_t>(), count, rk.as<uint32_t>(), s);
	std::vector<uint32_t> h(count);
	B3M_HIP(hipMemcpyAsync(h.data(), rk.p, 4 * count, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	for (uint64_t i = 0; i < count; ++i) ranks_out[i] = h[i];
	return 0;
	B3M_CATCH(1)
}

int b3m_gap(const void *d_text, uint64_t n, int inputtype, uint64_t a, uint64_t m, uint64_t mr, const uint8_t *d_bwtL, uint64_t p0L,
            const uint32_t *d_gtR, const uint64_t *anchors, uint64_t num_chains, uint64_t chain_len, uint32_t *d_sorted_ranks,
            uint32_t *d_G_dense, uint32_t *d_gtP, void *stream)
{
	B3M_TRY
	switches_load();
	require_device();
	hipStream_t s = (hipStream_t)stream;
	B3M_REQUIRE(inputtype == B3M_INPUT_PAC || inputtype == B3M_INPUT_PACTERM, "gap merge needs a 2-bit alphabet (pac/pacterm)");
	B3M_REQUIRE(a + m + mr <= n && m >= 1 && mr >= 1, "range out of bounds");
	B3M_REQUIRE(chain_len >= 1 && num_chains * chain_len >= mr && (num_chains - 1) * chain_len < mr, "chains do not tile R");
	PaddedText pt(d_text, n, inputtype, s);
	const uint64_t Rbeg = a + m, Rend = Rbeg + mr;
	uint64_t nb = rank_blocks(m);
	DevBuf blocks(64 * nb), scratch(64 * nb + 64), sws(4 * scan_ws_elems(nb) + 64);
	DevBuf p0d(16);
	{ uint32_t p032 = (uint32_t)p0L; B3M_HIP(hipMemcpy(p0d.p, &p032, 4, hipMemcpyHostToDevice)); }
	rank_build(d_bwtL, m, p0d.as<uint32_t>(), inputtype, blocks.as<uint32_t>(), scratch.as<uint32_t>(), sws.as<uint32_t>(), s);
	// C array from the text histogram of L
	DevBuf hb(8 * 256);
	B3M_HIP(hipMemsetAsync(hb.p, 0, 8 * 256, s));
	histogram_symbols(pt.T, a, m, hb.as<uint64_t>(), s);
	uint64_t hl[256];
	B3M_HIP(hipMemcpyAsync(hl, hb.p, 8 * 256, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	RankDict rd{};
	rd.blocks = blocks.as<uint32_t>(); rd.m = m; rd.p0ptr = p0d.as<uint32_t>();
	if (inputtype == B3M_INPUT_PACTERM) { uint64_t acc = hl[0]; for (int c = 0; c < 4; ++c) { rd.C[c] = (uint32_t)acc; acc += hl[c + 1]; } }
	else { uint64_t acc = 0; for (int c = 0; c < 4; ++c) { rd.C[c] = (uint32_t)acc; acc += hl[c]; } }
	// absolute gt arrays
	uint64_t words_abs = (n + 1 + 31) / 32 + 2;
	DevBuf gtc(4 * words_abs), gtn(4 * words_abs);
	{
		uint64_t wrel = (mr + 1 + 31) / 32;
		std::vector<uint32_t> hrel(wrel), habs(words_abs, 0);
		B3M_HIP(hipMemcpy(hrel.data(), d_gtR, 4 * wrel, hipMemcpyDeviceToHost));
		for (uint64_t k = 1; k <= mr; ++k) if ((hrel[k >> 5] >> (k & 31)) & 1u) { uint64_t x = Rbeg + k; habs[x >> 5] |= 1u << (x & 31); }
		B3M_HIP(hipMemcpy(gtc.p, habs.data(), 4 * words_abs, hipMemcpyHostToDevice));
	}
	B3M_HIP(hipMemsetAsync(gtn.p, 0, 4 * words_abs, s));
	std::vector<uint32_t> a32(num_chains);
	for (uint64_t c = 0; c < num_chains; ++c) a32[c] = (uint32_t)anchors[c];
	DevBuf an(4 * num_chains);
	B3M_HIP(hipMemcpy(an.p, a32.data(), 4 * num_chains, hipMemcpyHostToDevice));
	uint64_t slots = gap_walk_slots(num_chains, chain_len);
	DevBuf r0(4 * slots), r1(4 * slots), rws(radix_ws_bytes(slots));
	RadixWs rw; radix_ws_carve(rw, rws.p, slots);
	GapWalkArgs G{};
	G.T = pt.T; G.rd = rd; G.a = a; G.m = m; G.mr = mr; G.gt_cur = gtc.as<uint32_t>(); G.gt_next = gtn.as<uint32_t>();
	G.anchors = an.as<uint32_t>(); G.num_chains = num_chains; G.chain_len = chain_len; G.ranks_out = r0.as<uint32_t>();
	gap_walk(G, s);
	uint32_t *sr = nullptr;
	int nbits = 1; while ((1ull << nbits) <= m + 1 && nbits < 32) ++nbits;
	radix_sort_u32(r0.as<uint32_t>(), r1.as<uint32_t>(), slots, nbits, rw, s, nullptr, &sr, true);
	if (d_sorted_ranks) B3M_HIP(hipMemcpyAsync(d_sorted_ranks, sr, 4 * mr, hipMemcpyDeviceToDevice, s));
	if (d_G_dense) dense_gap_from_sorted(sr, mr, d_G_dense, m, s);
	if (d_gtP) {
		std::vector<uint32_t> habs(words_abs);
		B3M_HIP(hipMemcpyAsync(habs.data(), gtn.p, 4 * words_abs, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipStreamSynchronize(s));
		uint64_t tot = m + mr;
		uint64_t wrel = (tot + 1 + 31) / 32;
		std::vector<uint32_t> hrel(wrel, 0);
		for (uint64_t k = m; k < tot; ++k) { uint64_t x = a + k; if ((habs[x >> 5] >> (x & 31)) & 1u) hrel[k >> 5] |= 1u << (k & 31); }
		if (anchors[0] > p0L) hrel[tot >> 5] |= 1u << (tot & 31);
		B3M_HIP(hipMemcpy(d_gtP, hrel.data(), 4 * wrel, hipMemcpyHostToDevice));
	}
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	(void)Rend;
	return 0;
	B3M_CATCH(1)
}

int b3m_interleave(const uint32_t *d_sorted_ranks, const uint8_t *d_bwtL, uint64_t m, const uint8_t *d_bwtR, uint64_t mr, uint8_t *d_out,
                   uint64_t p0L, uint64_t *p0P_out, void *stream)
{
	B3M_TRY
	switches_load();
	require_device();
	hipStream_t s = (hipStream_t)stream;
	B3M_REQUIRE(m + mr < (1ull << 32), "block too large");
	DevBuf tq(4 * (interleave_tiles(m + mr) + 2));
	interleave(d_sorted_ranks, d_bwtL, m, d_bwtR, mr, d_out, tq.as<uint32_t>(), s);
	if (p0P_out) {
		uint64_t pr[2] = {p0L, 0};
		DevBuf pp(16);
		B3M_HIP(hipMemcpyAsync(pp.p, pr, 16, hipMemcpyHostToDevice, s));
		remap_ranks(d_sorted_ranks, mr, pp.as<uint64_t>(), 1, 0, 1, 1, s);
		B3M_HIP(hipMemcpyAsync(pr, pp.p, 16, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipStreamSynchronize(s));
		*p0P_out = pr[0];
	}
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	return 0;
	B3M_CATCH(1)
}


/* Multi-GPU form of b3m_gap (see include/b3m.h): walks only the chains [chain_first,
 * chain_first+num_chains) of the merge's global chain table, with gt bit arrays indexed by
 * absolute text position. */
int b3m_gap_chains(const void *d_text, uint64_t n, int inputtype, uint64_t a, uint64_t m, uint64_t mr, const uint8_t *d_bwtL, uint64_t p0L,
                   const uint32_t *d_gt_abs_in, const uint64_t *anchors, uint64_t chain_first, uint64_t num_chains, uint64_t chain_len,
                   uint32_t *d_sorted_ranks, uint64_t *count_out, uint32_t *d_gt_abs_out, void *stream)
{
	B3M_TRY
	switches_load();
	require_device();
	hipStream_t s = (hipStream_t)stream;
	B3M_REQUIRE(inputtype == B3M_INPUT_PAC || inputtype == B3M_INPUT_PACTERM, "gap merge needs a 2-bit alphabet (pac/pacterm)");
	B3M_REQUIRE(a + m + mr <= n && m >= 1 && mr >= 1 && chain_len >= 1, "range out of bounds");
	uint64_t total_chains = (mr + chain_len - 1) / chain_len;
	B3M_REQUIRE(chain_first + num_chains <= total_chains, "chain range out of bounds");
	if (num_chains == 0) { if (count_out) *count_out = 0; return 0; }
	PaddedText pt(d_text, n, inputtype, s);
	uint64_t nb = rank_blocks(m);
	DevBuf blocks(64 * nb), scratch(64 * nb + 64), sws(4 * scan_ws_elems(nb) + 64), p0d(16), hb(8 * 256);
	{ uint32_t p032 = (uint32_t)p0L; B3M_HIP(hipMemcpy(p0d.p, &p032, 4, hipMemcpyHostToDevice)); }
	rank_build(d_bwtL, m, p0d.as<uint32_t>(), inputtype, blocks.as<uint32_t>(), scratch.as<uint32_t>(), sws.as<uint32_t>(), s);
	B3M_HIP(hipMemsetAsync(hb.p, 0, 8 * 256, s));
	histogram_symbols(pt.T, a, m, hb.as<uint64_t>(), s);
	uint64_t hl[256];
	B3M_HIP(hipMemcpyAsync(hl, hb.p, 8 * 256, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	RankDict rd{};
	rd.blocks = blocks.as<uint32_t>(); rd.m = m; rd.p0ptr = p0d.as<uint32_t>();
	if (inputtype == B3M_INPUT_PACTERM) { uint64_t acc = hl[0]; for (int c = 0; c < 4; ++c) { rd.C[c] = (uint32_t)acc; acc += hl[c + 1]; } }
	else { uint64_t acc = 0; for (int c = 0; c < 4; ++c) { rd.C[c] = (uint32_t)acc; acc += hl[c]; } }
	std::vector<uint32_t> a32(num_chains);
	for (uint64_t c = 0; c < num_chains; ++c) a32[c] = (uint32_t)anchors[c];
	DevBuf an(4 * num_chains);
	B3M_HIP(hipMemcpy(an.p, a32.data(), 4 * num_chains, hipMemcpyHostToDevice));
	uint64_t slots = gap_walk_slots(num_chains, chain_len);
	DevBuf r0(4 * slots), r1(4 * slots), rws(radix_ws_bytes(slots));
	RadixWs rw; radix_ws_carve(rw, rws.p, slots);
	GapWalkArgs G{};
	G.T = pt.T; G.rd = rd; G.a = a; G.m = m; G.mr = mr; G.gt_cur = d_gt_abs_in; G.gt_next = d_gt_abs_out;
	G.anchors = an.as<uint32_t>(); G.num_chains = num_chains; G.chain_len = chain_len; G.chain_first = chain_first;
	G.ranks_out = r0.as<uint32_t>();
	gap_walk(G, s);
	uint32_t *sr = nullptr;
	int nbits = 1; while ((1ull << nbits) <= m + 1 && nbits < 32) ++nbits;
	radix_sort_u32(r0.as<uint32_t>(), r1.as<uint32_t>(), slots, nbits, rw, s, nullptr, &sr, true);
	// positions covered by this chain range
	uint64_t hi = mr - chain_first * chain_len;
	uint64_t last = chain_first + num_chains;
	uint64_t lo = (last * chain_len >= mr) ? 0 : mr - last * chain_len;
	uint64_t cnt = hi - lo;
	B3M_HIP(hipMemcpyAsync(d_sorted_ranks, sr, 4 * cnt, hipMemcpyDeviceToDevice, s));
	if (count_out) *count_out = cnt;
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	return 0;
	B3M_CATCH(1)
}


int b3m_sort_u32(const uint32_t *d_keys, uint64_t count, int nbits, uint32_t *d_out, void *stream)
{
	B3M_TRY
	switches_load();
	require_device();
	hipStream_t s = (hipStream_t)stream;
	if (count == 0) return 0;
	B3M_REQUIRE(nbits >= 1 && nbits <= 32, "nbits out of range");
	DevBuf a(4 * count), b(4 * count), rws(radix_ws_bytes(count));
	RadixWs rw; radix_ws_carve(rw, rws.p, count);
	B3M_HIP(hipMemcpyAsync(a.p, d_keys, 4 * count, hipMemcpyDeviceToDevice, s));
	uint32_t *res = nullptr;
	radix_sort_u32(a.as<uint32_t>(), b.as<uint32_t>(), count, nbits, rw, s, nullptr, &res, true);
	B3M_HIP(hipMemcpyAsync(d_out, res, 4 * count, hipMemcpyDeviceToDevice, s));
	B3M_HIP(hipStreamSynchronize(s));
	return 0;
	B3M_CATCH(1)
}

int b3m_bits_place(const uint32_t *d_rel, uint64_t count, uint64_t start, uint32_t *d_abs, void *stream)
{
	B3M_TRY
	require_device();
	hipStream_t s = (hipStream_t)stream;
	if (count) hipLaunchKernelGGL(k_place_bits, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, s, d_rel, count, start, d_abs);
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	return 0;
	B3M_CATCH(1)
}


/* b3m_block_sort with the gt bits written on the device into an ABSOLUTE-position bit array
 * (d_gt_abs, (n+1+31)/32+2 words, OR-ed in): bit x = [S_x > S_start] for x in (start, start+len].
 * No host round trip over the block: the form the multi-GPU driver uses. */
int b3m_block_sort_abs(const void *d_text, uint64_t n, int inputtype, uint64_t start, uint64_t len, uint32_t *d_sa_out,
                       uint8_t *d_bwt_out, uint32_t *d_gt_abs, uint64_t *p0_out, void *stream)
{
	B3M_TRY
	switches_load();
	require_device();
	hipStream_t s = (hipStream_t)stream;
	B3M_REQUIRE(start + len <= n && len >= 1, "range out of bounds");
	PaddedText pt(d_text, n, inputtype, s);
	const uint64_t cap = len + std::min(std::min(block_overhang(), len / 8), n - (start + len));     // room for an overhang behind the block (tie_rounds_domain)
	DevBuf ws(blocksort_ws_bytes(cap));
	BlockSortWs bs; blocksort_ws_carve(bs, ws.p, cap);
	BlockSortOut so;
	DevBuf sc(64);     // [0] p0 u32, [2..3] p0 pair u64, [4] end rank, [6..7] end position u64
	uint32_t *d_p0 = sc.as<uint32_t>();
	uint64_t *d_pair = (uint64_t *)((uint8_t *)sc.p + 8);
	uint32_t *d_er = (uint32_t *)((uint8_t *)sc.p + 32);
	uint64_t *d_ep = (uint64_t *)((uint8_t *)sc.p + 40);
	B3M_HIP(hipMemsetAsync(sc.p, 0, 64, s));
	DevBuf bwt_tmp(d_bwt_out ? 16 : len);
	block_sort(pt.T, start, len, bs, s, nullptr, so, d_bwt_out ? d_bwt_out : bwt_tmp.as<uint8_t>(), d_p0);
	if (d_gt_abs) {
		hipLaunchKernelGGL(k_p0_to_pair, dim3(1), dim3(64), 0, s, d_p0, d_pair, start, (uint64_t)0);
		emit_gt_from_text(pt.T, start, len, d_gt_abs, s);
		uint64_t endpos = start + len;
		B3M_HIP(hipMemcpyAsync(d_ep, &endpos, 8, hipMemcpyHostToDevice, s));
		rank_of_positions(pt.T, so.d_sa, len, d_ep, 1, d_er, s);
		hipLaunchKernelGGL(k_set_endbit, dim3(1), dim3(64), 0, s, d_er, d_pair, d_gt_abs, start + len);
	}
	uint32_t p0 = 0;
	B3M_HIP(hipMemcpyAsync(&p0, d_p0, 4, hipMemcpyDeviceToHost, s));
	if (d_sa_out) B3M_HIP(hipMemcpyAsync(d_sa_out, so.d_sa, 4 * len, hipMemcpyDeviceToDevice, s));
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	if (p0_out) *p0_out = p0;
	return 0;
	B3M_CATCH(1)
}


/* SURVEY.md 8f rows 1 + 3 on the device: LF walks over the final BWT of the plan's last run between
 * its (rank,pos) samples.  Verifies every BWT symbol against the text and every chain against the
 * next anchor (checkbwt), and, if sarate > 0, fills sa_out[i] = SA[i*sarate] (bwtcomputessa).
 * pac/pacterm, unsharded plans created with isasamplingrate > 0. */
/* LF walks over a final 2-bit BWT on the current device: dictionary with 64-bit counts, one chain per anchor.  hist: symbol counts of the text (pacterm: [0] the
 * terminator, [1..4] the bases; pac: [0..3]); anchors: K (rank,pos) pairs in position order, the first at position 0 (host); p0 = rank of the terminator entry
 * (~0: none).  SA samples go to sa_out (host) and/or stay in d_sa_keep (device, ceil(n/sarate) u64) for a caller that streams them out itself. */
static void lf_walk_2bit(const TextView &T, const uint8_t *d_bwt, uint64_t p0, const uint64_t *hist, const uint64_t *anchors, uint64_t K, uint64_t sarate,
                         uint64_t *sa_out, uint64_t *d_sa_keep, uint64_t *mismatches_out, hipStream_t s, uint64_t isarate = 0, uint64_t *isa_out = nullptr, bool nocheck = false,
                         uint64_t chain_first = 0, uint64_t chain_count = 0)
{
	const uint64_t n = T.n;
	B3M_REQUIRE(K >= 1 && anchors[1] == 0, "anchor at position 0 missing");
	const uint64_t nsa = sarate ? (n + sarate - 1) / sarate : 0;
	const uint64_t nb = rank_blocks(n), nsb = rank_superblocks(n);
	const uint64_t nisa = (isarate && isa_out) ? (n + isarate - 1) / isarate : 0;
	DevBuf blocks(64 * nb), scratch(64 * nb + 64), sws(4 * scan_ws_elems(nb) + 64), sb(32 * nsb + 64), p0d(16), danc(16 * K), dsa(d_sa_keep ? 16 : 8 * (nsa + 1)), dmis(16), disa(8 * (nisa + 1));
	uint64_t *d_sa = d_sa_keep ? d_sa_keep : dsa.as<uint64_t>();
	if (nisa) B3M_HIP(hipMemsetAsync(disa.p, 0xFF, 8 * nisa, s));
	LfExtra X; X.isarate = nisa ? isarate : 0; X.d_isa_out = disa.as<uint64_t>(); X.nocheck = nocheck; X.chain_first = chain_first; X.chain_count = chain_count;
	B3M_HIP(hipMemcpyAsync(p0d.p, &p0, 8, hipMemcpyHostToDevice, s));
	B3M_HIP(hipMemcpyAsync(danc.p, anchors, 16 * K, hipMemcpyHostToDevice, s));
	B3M_HIP(hipMemsetAsync(dmis.p, 0, 16, s));
	if (nsa) B3M_HIP(hipMemsetAsync(d_sa, 0xFF, 8 * nsa, s));
	rank_build64(d_bwt, n, p0d.as<uint64_t>(), T.type, blocks.as<uint32_t>(), scratch.as<uint32_t>(), sws.as<uint32_t>(), sb.as<uint64_t>(), s);
	uint64_t C[4];
	if (T.type == B3M_INPUT_PACTERM) { uint64_t acc = hist[0]; for (int c = 0; c < 4; ++c) { C[c] = acc; acc += hist[c + 1]; } }
	else { uint64_t acc = 0; for (int c = 0; c < 4; ++c) { C[c] = acc; acc += hist[c]; } }
	lf_walk(T, blocks.as<uint32_t>(), sb.as<uint64_t>(), p0, C, danc.as<uint64_t>(), K, sarate, d_sa, dmis.as<unsigned long long>(), s, X);
	unsigned long long mm[2] = {0, 0};
	B3M_HIP(hipMemcpyAsync(mm, dmis.p, 16, hipMemcpyDeviceToHost, s));
	if (nsa && sa_out) B3M_HIP(hipMemcpyAsync(sa_out, d_sa, 8 * nsa, hipMemcpyDeviceToHost, s));
	if (nisa) B3M_HIP(hipMemcpyAsync(isa_out, disa.p, 8 * nisa, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	mismatches_out[0] = mm[0]; mismatches_out[1] = mm[1];
}

/* byte alphabets (and 32-bit symbols): the walk over the LF array of the whole BWT (n < 2^32) */
static void lf_walk_bytes_any(const TextView &T, const uint8_t *d_bwt, const uint64_t *anchors, uint64_t K, uint64_t sarate, uint64_t *sa_out, uint64_t *mismatches_out, hipStream_t s,
                              uint64_t isarate = 0, uint64_t *isa_out = nullptr, bool nocheck = false)
{
	const uint64_t n = T.n;
	B3M_REQUIRE(n < (1ull << 32), "byte LF walk: n out of range");
	B3M_REQUIRE(K >= 1 && anchors[1] == 0, "anchor at position 0 missing");
	const uint64_t nsa = sarate ? (n + sarate - 1) / sarate : 0;
	const uint64_t nisa = (isarate && isa_out) ? (n + isarate - 1) / isarate : 0;
	DevBuf k0(8 * n), k1(8 * n), v0(4 * n), v1(4 * n), lf(4 * n), rws(radix_ws_bytes(n)), danc(16 * K), dsa(8 * (nsa + 1)), dmis(16), disa(8 * (nisa + 1));
	RadixWs rw; radix_ws_carve(rw, rws.p, n);
	if (nisa) B3M_HIP(hipMemsetAsync(disa.p, 0xFF, 8 * nisa, s));
	LfExtra X; X.isarate = nisa ? isarate : 0; X.d_isa_out = disa.as<uint64_t>(); X.nocheck = nocheck;
	B3M_HIP(hipMemcpyAsync(danc.p, anchors, 16 * K, hipMemcpyHostToDevice, s));
	B3M_HIP(hipMemsetAsync(dmis.p, 0, 16, s));
	B3M_HIP(hipMemsetAsync(dsa.p, 0xFF, 8 * (nsa + 1), s));
	lf_walk_bytes(T, d_bwt, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint32_t>(), v1.as<uint32_t>(), lf.as<uint32_t>(), rw,
	              danc.as<uint64_t>(), K, sarate, dsa.as<uint64_t>(), dmis.as<unsigned long long>(), s, X);
	unsigned long long mm[2] = {0, 0};
	B3M_HIP(hipMemcpyAsync(mm, dmis.p, 16, hipMemcpyDeviceToHost, s));
	if (nsa && sa_out) B3M_HIP(hipMemcpyAsync(sa_out, dsa.p, 8 * nsa, hipMemcpyDeviceToHost, s));
	if (nisa) B3M_HIP(hipMemcpyAsync(isa_out, disa.p, 8 * nisa, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	mismatches_out[0] = mm[0]; mismatches_out[1] = mm[1];
}

/* the walks as one job description: everything b3m_lf_walk takes, plus ISA samples (isa_out[i] = rank of the rotation at position i * isarate, host) and a text that may
 * be absent (d_text = NULL: the symbols are not compared, the chain closures still are): what bwtcomputessa needs (/root/reference/src/bwtcomputessa.cpp:39-51) */
int b3m_lf_walk_job(const b3m_lf_job *j, uint64_t *mismatches_out, void *stream)
{
	B3M_TRY
	switches_load();
	require_device();
	B3M_REQUIRE(j && j->d_bwt && j->hist256 && j->sample_pairs && mismatches_out && j->num_samples >= 1, "b3m_lf_walk_job: null argument");
	B3M_REQUIRE(j->inputtype == B3M_INPUT_PAC || j->inputtype == B3M_INPUT_PACTERM || j->inputtype == B3M_INPUT_BYTESTREAM, "b3m_lf_walk_job: bytestream, pac and pacterm texts");
	static const uint8_t no_text[64] = {0};
	const bool nocheck = j->d_text == nullptr;
	TextView T{nocheck ? no_text : (const uint8_t *)j->d_text, j->n, num_bases(j->n, j->inputtype), j->inputtype};
	if (j->inputtype == B3M_INPUT_BYTESTREAM) {
		B3M_REQUIRE(!j->d_sa_out, "b3m_lf_walk_job: SA samples of a byte text go to host memory (sa_out)");
		lf_walk_bytes_any(T, j->d_bwt, j->sample_pairs, j->num_samples, j->sarate, j->sa_out, mismatches_out, (hipStream_t)stream, j->isarate, j->isa_out, nocheck);
		return 0;
	}
	lf_walk_2bit(T, j->d_bwt, j->inputtype == B3M_INPUT_PACTERM ? j->p0 : ~0ull, j->hist256, j->sample_pairs, j->num_samples, j->sarate, j->sa_out, j->d_sa_out, mismatches_out,
	             (hipStream_t)stream, j->isarate, j->isa_out, nocheck);
	return 0;
	B3M_CATCH(1)
}

/* ONE DEVICE'S SHARE of the walks of a run over several devices (b3m_compute_bwt with num_gpus > 1): the chains [chain_first, chain_first + chain_count) of the job
 * (chain k runs from sample k + 1 back to sample k), over the whole BWT on the current device; d_sa_out (device, ceil(n / sarate) u64, required when sarate > 0) is
 * filled with ~0 first and receives the SA samples these chains meet: the shares of all devices together hold every sample once.  pac / pacterm. */
int b3m_lf_walk_chains(const b3m_lf_job *j, uint64_t chain_first, uint64_t chain_count, uint64_t *mismatches_out, void *stream)
{
	B3M_TRY
	switches_load();
	require_device();
	B3M_REQUIRE(j && j->d_bwt && j->d_text && j->hist256 && j->sample_pairs && mismatches_out && j->num_samples >= 1, "b3m_lf_walk_chains: null argument");
	B3M_REQUIRE(j->inputtype == B3M_INPUT_PAC || j->inputtype == B3M_INPUT_PACTERM, "b3m_lf_walk_chains: pac and pacterm texts");
	B3M_REQUIRE(chain_count >= 1 && chain_first + chain_count <= j->num_samples, "b3m_lf_walk_chains: chain range out of range");
	B3M_REQUIRE(!j->sarate || j->d_sa_out, "b3m_lf_walk_chains: the SA samples of a share stay on the device (d_sa_out)");
	B3M_REQUIRE(!j->sa_out && !j->isa_out, "b3m_lf_walk_chains: host outputs belong to the whole job (b3m_lf_walk_job)");
	TextView T{(const uint8_t *)j->d_text, j->n, num_bases(j->n, j->inputtype), j->inputtype};
	lf_walk_2bit(T, j->d_bwt, j->inputtype == B3M_INPUT_PACTERM ? j->p0 : ~0ull, j->hist256, j->sample_pairs, j->num_samples, j->sarate, nullptr, j->d_sa_out, mismatches_out,
	             (hipStream_t)stream, 0, nullptr, false, chain_first, chain_count);
	return 0;
	B3M_CATCH(1)
}
/* d_dst[i] = d_src[i] where d_src[i] != ~0 (count u64 on the current device): the shares of b3m_lf_walk_chains put together */
__global__ void k_merge_set_u64(uint64_t *dst, const uint64_t *src, uint64_t count)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < count && src[i] != ~0ull) dst[i] = src[i];
}
/* (rank,pos) sample tables of suffix-range shards put together on the device: slot i of d_src with a rank != ~0 goes to slot i of d_dst; stats_out[0] += slots both
 * tables had ranked, stats_out[1] = slots of d_dst that are still empty afterwards (host memory) */
__global__ void k_merge_pairs(uint64_t *dst, const uint64_t *src, uint64_t npairs, unsigned long long *stats)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= npairs) return;
	const uint64_t r = src[2 * i];
	if (r != ~0ull) {
		if (dst[2 * i] != ~0ull) atomicAdd(&stats[0], 1ull);
		dst[2 * i] = r; dst[2 * i + 1] = src[2 * i + 1];
	} else if (dst[2 * i] == ~0ull) atomicAdd(&stats[1], 1ull);
}
int b3m_merge_sample_pairs(uint64_t *d_dst, const uint64_t *d_src, uint64_t npairs, uint64_t *stats_out, void *stream)
{
	B3M_TRY
	require_device();
	B3M_REQUIRE(d_dst && d_src && stats_out, "b3m_merge_sample_pairs: null argument");
	hipStream_t s = (hipStream_t)stream;
	DevBuf st(16);
	B3M_HIP(hipMemsetAsync(st.p, 0, 16, s));
	if (npairs) hipLaunchKernelGGL(k_merge_pairs, dim3((uint32_t)((npairs + 255) / 256)), dim3(256), 0, s, d_dst, d_src, npairs, st.as<unsigned long long>());
	unsigned long long h[2] = {0, 0};
	B3M_HIP(hipMemcpyAsync(h, st.p, 16, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	stats_out[0] += h[0]; stats_out[1] = h[1];
	return 0;
	B3M_CATCH(1)
}
int b3m_merge_samples(uint64_t *d_dst, const uint64_t *d_src, uint64_t count, void *stream)
{
	B3M_TRY
	require_device();
	B3M_REQUIRE(d_dst && d_src, "b3m_merge_samples: null argument");
	if (count) hipLaunchKernelGGL(k_merge_set_u64, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_dst, d_src, count);
	B3M_HIP(hipStreamSynchronize((hipStream_t)stream));
	return 0;
	B3M_CATCH(1)
}

/* the same as a stage of its own, for a BWT that no single plan produced (the slices of suffix-range shards gathered on one GPU): d_text = the packed text on
 * the current device with at least 8 readable bytes behind it */
int b3m_lf_walk(const void *d_text, uint64_t n, int inputtype, const uint8_t *d_bwt, uint64_t p0, const uint64_t *hist256, const uint64_t *sample_pairs,
                uint64_t num_samples, uint64_t sarate, uint64_t *sa_out, uint64_t *d_sa_out, uint64_t *mismatches_out, void *stream)
{
	B3M_TRY
	require_device();
	B3M_REQUIRE(d_text && d_bwt && hist256 && sample_pairs && mismatches_out && num_samples >= 1, "b3m_lf_walk: null argument");
	B3M_REQUIRE(inputtype == B3M_INPUT_PAC || inputtype == B3M_INPUT_PACTERM || inputtype == B3M_INPUT_BYTESTREAM, "b3m_lf_walk: bytestream, pac and pacterm texts");
	TextView T{(const uint8_t *)d_text, n, num_bases(n, inputtype), inputtype};
	if (inputtype == B3M_INPUT_BYTESTREAM) {
		B3M_REQUIRE(!d_sa_out, "b3m_lf_walk: SA samples of a byte text go to host memory (sa_out)");
		lf_walk_bytes_any(T, d_bwt, sample_pairs, num_samples, sarate, sa_out, mismatches_out, (hipStream_t)stream);
		return 0;
	}
	lf_walk_2bit(T, d_bwt, inputtype == B3M_INPUT_PACTERM ? p0 : ~0ull, hist256, sample_pairs, num_samples, sarate, sa_out, d_sa_out, mismatches_out, (hipStream_t)stream);
	return 0;
	B3M_CATCH(1)
}

static int plan_lf_walk_impl(b3m_plan *p, const uint8_t *d_bwt, uint64_t sarate, uint64_t *sa_out, uint64_t *d_sa_keep, uint64_t *mismatches_out, void *stream);
int b3m_plan_lf_walk(b3m_plan *p, const uint8_t *d_bwt, uint64_t sarate, uint64_t *sa_out, uint64_t *mismatches_out, void *stream)
{
	return plan_lf_walk_impl(p, d_bwt, sarate, sa_out, nullptr, mismatches_out, stream);
}
/* the same with the SA samples left in device memory (d_sa_out: ceil(n/sarate) u64), for a caller that streams them to a file itself; pac / pacterm */
int b3m_plan_lf_walk_dev(b3m_plan *p, const uint8_t *d_bwt, uint64_t sarate, uint64_t *d_sa_out, uint64_t *mismatches_out, void *stream)
{
	if (!d_sa_out || !sarate) { set_error("b3m_plan_lf_walk_dev: no output array / sarate = 0"); return 1; }
	return plan_lf_walk_impl(p, d_bwt, sarate, nullptr, d_sa_out, mismatches_out, stream);
}
/* the pinned image of the last run's (rank,pos) samples: slot i belongs to position i * rate; a slot that reads (~0, .) is another shard's (suffix-range
 * shards) -- an unsharded plan has none of those (checked, as b3m_plan_samples does).  Valid until the plan runs again or is destroyed. */
int b3m_plan_samples_raw(b3m_plan *p, const uint64_t **slots_out, uint64_t *nslots_out)
{
	B3M_TRY
	B3M_REQUIRE(p && slots_out && nslots_out, "b3m_plan_samples_raw: null argument");
	if (p->samples_lazy && p->isarate) (void)samples_of(p);          // (a large table lives in HBM: this brings it over; b3m_plan_samples_device does not)
	else if (p->nshards == 1 && p->isarate && p->samples_pending) {
		// every multiple of the rate is a suffix of the text: a slot that still reads "empty" means a sort kernel lost a sample (looked at in place, by a
		// few threads: at rate 64 the image of a 3-Gbase text is 750 MB, and the copy samples_of makes of it is not needed here)
		const uint64_t *raw = p->h_raw; const uint64_t N = p->total_samples;
		std::atomic<uint64_t> bad(~0ull);
		auto scan = [&](uint64_t a, uint64_t e) { for (uint64_t i = a; i < e; ++i) if (raw[2 * i] == ~0ull) { uint64_t cur = bad.load(); while (i < cur && !bad.compare_exchange_weak(cur, i)) {} return; } };
		const unsigned T = N > (1u << 20) ? 4 : 1;
		std::vector<std::thread> th;
		for (unsigned t = 1; t < T; ++t) th.emplace_back(scan, N * t / T, N * (t + 1) / T);
		scan(0, N / T);
		for (auto &x : th) x.join();
		B3M_REQUIRE(bad.load() == ~0ull, "internal: the (rank,pos) sample of position " + std::to_string(bad.load() * p->isarate) + " was not written by the sort");
	}
	*slots_out = p->h_raw; *nslots_out = p->isarate ? p->total_samples : 0;
	return 0;
	B3M_CATCH(1)
}
__global__ void k_count_empty_slots(const uint64_t *slots, uint64_t n, unsigned long long *out)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const bool e = i < n && slots[2 * i] == ~0ull;
	const unsigned long long m = __ballot(e);
	if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}
/* the device image of the last run's (rank,pos) samples (slot i = position i * rate; (~0, .) = another shard's) and the number of its slots nobody wrote
 * (an unsharded plan must have none: the caller checks).  Valid until the plan runs again or is destroyed. */
int b3m_plan_samples_device(b3m_plan *p, const uint64_t **d_slots_out, uint64_t *nslots_out, uint64_t *empty_out, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && d_slots_out && nslots_out, "b3m_plan_samples_device: null argument");
	hipStream_t s = (hipStream_t)stream;
	const uint64_t N = p->isarate ? p->total_samples : 0;
	*d_slots_out = p->d_samples; *nslots_out = N;
	if (empty_out) {
		*empty_out = 0;
		if (N) {
			DevBuf c(16);
			B3M_HIP(hipMemsetAsync(c.p, 0, 8, s));
			hipLaunchKernelGGL(k_count_empty_slots, dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, s, p->d_samples, N, c.as<unsigned long long>());
			unsigned long long h = 0;
			B3M_HIP(hipMemcpyAsync(&h, c.p, 8, hipMemcpyDeviceToHost, s));
			B3M_HIP(hipStreamSynchronize(s));
			*empty_out = h;
		}
	}
	return 0;
	B3M_CATCH(1)
}
static int plan_lf_walk_impl(b3m_plan *p, const uint8_t *d_bwt, uint64_t sarate, uint64_t *sa_out, uint64_t *d_sa_keep, uint64_t *mismatches_out, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && d_bwt && mismatches_out, "b3m_plan_lf_walk: null argument");
	B3M_REQUIRE(p->nshards == 1, "the LF walk needs the whole BWT (unsharded plan)");
	B3M_REQUIRE(p->isarate && !samples_of(p).empty(), "the plan has no (rank,pos) samples: create it with isasamplingrate > 0 and run it");
	hipStream_t s = (hipStream_t)stream;
	const uint64_t n = p->n;
	TextView T{p->d_text, n, num_bases(n, p->type), p->type};
	// anchors = the samples, already in position order
	uint64_t K = p->h_samples.size() / 2;
	const std::vector<uint64_t> &anc = p->h_samples;
	B3M_REQUIRE(anc[1] == 0, "anchor at position 0 missing");
	if (p->type == B3M_INPUT_BYTESTREAM || p->type == B3M_INPUT_SYMBOLS32) {
		B3M_REQUIRE(!d_sa_keep, "b3m_plan_lf_walk_dev: pac / pacterm plans only");
		// byte alphabet (and wide symbols): the LF array itself (one stable sort of the BWT by symbol) instead of a rank dictionary
		lf_walk_bytes_any(T, d_bwt, anc.data(), K, sarate, sa_out, mismatches_out, s);
		return 0;
	}
	// 2-bit alphabets: rank dictionary of the whole BWT with 64-bit counts (a symbol may occur 2^32 times and more: 30 Gbase)
	const uint64_t p064 = (p->type == B3M_INPUT_PACTERM) ? p->res.p0 : ~0ull;   // pac: no terminator entry
	lf_walk_2bit(T, d_bwt, p064, p->h_hist.data(), anc.data(), K, sarate, sa_out, d_sa_keep, mismatches_out, s);
	return 0;
	B3M_CATCH(1)
}


// ================================================================================================
// b3m_dist_plan: this rank's part of the multi-GPU block merge (north-star exchange path).
// One process per GPU; the collectives (RCCL all-to-all of gap buckets, broadcasts of packed BWT(L) slices and gt words) are
// issued by the host side between these calls (bwtb3m_amd/distributed.py, torch.distributed).  Everything here works in ONE
// workspace allocated at creation: padded text, leaf sort, rank dictionary, walk / sort buffers.  No hipMalloc per call.
// ================================================================================================
struct b3m_dist_plan {
	uint64_t n = 0; int type = 0;
	uint64_t max_leaf = 0, max_L = 0, max_R = 0;
	uint8_t *arena = nullptr; uint64_t ws_bytes = 0;
	uint8_t *d_text = nullptr; uint64_t text_cap = 0;
	BlockSortWs bs;
	uint32_t *d_sa_last = nullptr; uint64_t sa_len = 0;            // suffix array of the last leaf sort (inside bs)
	uint32_t *d_blocks = nullptr, *d_scratch = nullptr, *d_scan_ws = nullptr;   // rank dictionary of BWT(L)
	uint64_t dict_m = 0, dict_p0 = 0;
	uint32_t *d_r0 = nullptr, *d_r1 = nullptr; RadixWs radix;       // rank lists (walk output, received buckets)
	uint32_t *d_tile_q = nullptr;
	uint32_t *d_anchors = nullptr; uint64_t anchors_cap = 0;
	uint8_t *d_lcopy = nullptr; uint32_t *d_C = nullptr, *d_th = nullptr;   // byte alphabets: aligned copy of BWT(L), the 256 symbol starts of L's text, the build's tile counts (d_blocks = count rows)
	uint64_t leaf_hist[256] = {0};                                 // symbol counts of the last leaf's text
	uint64_t *d_small = nullptr;                                   // [0..1] p0 pair, [2] end rank (u32), [3] end position, [8..8+64) bounds, [72..72+64) cuts, [136..] query positions / ranks
	uint32_t *d_p0u = nullptr;
};
static const uint64_t DIST_SMALL = 136 + 2 * 4096;

b3m_dist_plan *b3m_dist_create(uint64_t n, int inputtype, uint64_t max_leaf, uint64_t max_L, uint64_t max_R)
{
	b3m_dist_plan *p = nullptr;
	B3M_TRY
	switches_load();
	require_device();
	B3M_REQUIRE(inputtype == B3M_INPUT_PAC || inputtype == B3M_INPUT_PACTERM || inputtype == B3M_INPUT_BYTESTREAM, "the multi-GPU block merge takes pac, pacterm and bytestream texts");
	B3M_REQUIRE(n >= 1 && n < (1ull << 32), "n out of range for the block-merge mode (use suffix-range shards above 2^32)");
	const bool bytes = inputtype == B3M_INPUT_BYTESTREAM;
	B3M_REQUIRE(max_leaf >= 1 && max_leaf <= n && max_L <= n && max_R <= n, "dist plan: sizes out of range");
	p = new b3m_dist_plan();
	p->n = n; p->type = inputtype; p->max_leaf = max_leaf; p->max_L = max_L; p->max_R = max_R;
	uint64_t off = 0;
	auto take = [&](uint64_t bytes) { uint64_t o = off; off += al256(bytes); return o; };
	p->text_cap = al256(packed_text_bytes(n, inputtype) + 64);
	const uint64_t nbk = rank_blocks(max_L ? max_L : 1);
	const uint64_t chains_max = max_R / 32 + 2;
	const uint64_t leaf_cap = max_leaf + (max_leaf < n ? std::min(block_overhang(), max_leaf / 8) : 0);      // room for an overhang (tie_rounds_domain)
	// byte alphabets: count rows (1 KB per 64 symbols of L), an aligned copy of BWT(L) + the 256 symbol starts + the build's tile counts (as the one-GPU plan)
	const uint64_t mL = max_L ? max_L : 1;
	uint64_t o_text = take(p->text_cap), o_bs = take(blocksort_ws_bytes(leaf_cap)), o_rb = take(bytes ? 1024 * byte_rank_rows(mL) : 64 * nbk),
	         o_rs = take(bytes ? al256(mL) + 256 + 1024 + 1024 * byte_rank_tiles(mL) + 1024 : 64 * nbk + 64),
	         o_sw = take(4 * scan_ws_elems(nbk) + 64), o_r0 = take(4 * (max_R + max_R / 8 + 256)), o_r1 = take(4 * (max_R + max_R / 8 + 256)), o_rx = take(radix_ws_bytes(max_R + max_R / 8 + 256)),
	         o_tq = take(4 * (interleave_tiles(max_L + max_R + 1) + 2)), o_an = take(4 * chains_max), o_sm = take(8 * DIST_SMALL), o_p0 = take(64);
	p->ws_bytes = off;
	size_t free_b = 0, total_b = 0;
	B3M_HIP(hipMemGetInfo(&free_b, &total_b));
	if (off > free_b) throw Error("dist plan: workspace does not fit the free HBM");
	B3M_HIP(hipMalloc((void **)&p->arena, off));
	uint8_t *A = p->arena;
	p->d_text = A + o_text;
	blocksort_ws_carve(p->bs, A + o_bs, leaf_cap);
	p->d_blocks = (uint32_t *)(A + o_rb); p->d_scratch = (uint32_t *)(A + o_rs); p->d_scan_ws = (uint32_t *)(A + o_sw);
	if (bytes) { p->d_lcopy = A + o_rs; p->d_C = (uint32_t *)(p->d_lcopy + al256(mL) + 256); p->d_th = p->d_C + 256; }
	p->d_r0 = (uint32_t *)(A + o_r0); p->d_r1 = (uint32_t *)(A + o_r1);
	radix_ws_carve(p->radix, A + o_rx, max_R + max_R / 8 + 256);
	p->d_tile_q = (uint32_t *)(A + o_tq);
	p->d_anchors = (uint32_t *)(A + o_an); p->anchors_cap = chains_max;
	p->d_small = (uint64_t *)(A + o_sm); p->d_p0u = (uint32_t *)(A + o_p0);
	return p;
	B3M_CATCH((b3m_dist_destroy(p), (b3m_dist_plan *)nullptr))
}
void b3m_dist_destroy(b3m_dist_plan *p)
{
	if (!p) return;
	if (p->arena) (void)hipFree(p->arena);
	delete p;
}
uint64_t b3m_dist_workspace_bytes(const b3m_dist_plan *p) { return p ? p->ws_bytes : 0; }

static TextView dist_text(const b3m_dist_plan *p) { return TextView{p->d_text, p->n, num_bases(p->n, p->type), p->type}; }

int b3m_dist_set_text(b3m_dist_plan *p, const void *d_text, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && d_text, "b3m_dist_set_text: null argument");
	hipStream_t s = (hipStream_t)stream;
	const uint64_t tb = packed_text_bytes(p->n, p->type);
	B3M_HIP(hipMemsetAsync(p->d_text + (tb & ~63ull), 0, p->text_cap - (tb & ~63ull), s));
	B3M_HIP(hipMemcpyAsync(p->d_text, d_text, tb, hipMemcpyDeviceToDevice, s));
	return 0;
	B3M_CATCH(1)
}

/* leaf [a, a+m): BWT bytes, gt bits OR-ed into the absolute-position bit array d_gt_abs ((n+1+31)/32+2 words), rank of S_a;
 * hist5_out (host, may be NULL): symbol counts 0..4 of the leaf's text.  The suffix array stays in the workspace for b3m_dist_rank_of. */
int b3m_dist_leaf(b3m_dist_plan *p, uint64_t a, uint64_t m, uint8_t *d_bwt_out, uint32_t *d_gt_abs, uint64_t *p0_out, uint64_t *hist5_out, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && d_bwt_out && d_gt_abs, "b3m_dist_leaf: null argument");
	B3M_REQUIRE(a + m <= p->n && m >= 1 && m <= p->max_leaf, "leaf out of range");
	hipStream_t s = (hipStream_t)stream;
	const TextView T = dist_text(p);
	BlockSortOut so;
	uint64_t *d_pair = p->d_small, *d_ep = p->d_small + 3; uint32_t *d_er = (uint32_t *)(p->d_small + 2);
	uint64_t *d_hist = p->d_small + 136;                    // 256 u64, free until the first rank query
	B3M_HIP(hipMemsetAsync(p->d_small, 0xFF, 16, s));
	B3M_HIP(hipMemsetAsync(p->d_small + 2, 0, 8, s));
	B3M_HIP(hipMemsetAsync(p->d_p0u, 0xFF, 4, s));
	B3M_HIP(hipMemsetAsync(d_hist, 0, 8 * 256, s));
	p->bs.d_symhist_out = d_hist;
	block_sort(T, a, m, p->bs, s, nullptr, so, d_bwt_out, p->d_p0u);
	p->bs.d_symhist_out = nullptr;
	if (!so.symhist_done) histogram_symbols(T, a, m, d_hist, s);
	p->d_sa_last = so.d_sa; p->sa_len = m;
	hipLaunchKernelGGL(k_p0_to_pair, dim3(1), dim3(64), 0, s, p->d_p0u, d_pair, a, (uint64_t)0);
	if (p->type == B3M_INPUT_BYTESTREAM) hipLaunchKernelGGL(k_emit_gt_ptr, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, s, so.d_sa, m, a, p->d_p0u, d_gt_abs);
	else emit_gt_from_text(T, a, m, d_gt_abs, s);
	const uint64_t endpos = a + m;
	B3M_HIP(hipMemcpyAsync(d_ep, &endpos, 8, hipMemcpyHostToDevice, s));
	rank_of_positions(T, so.d_sa, m, d_ep, 1, d_er, s);
	hipLaunchKernelGGL(k_set_endbit, dim3(1), dim3(64), 0, s, d_er, d_pair, d_gt_abs, a + m);
	uint64_t pr[2] = {0, 0}, *hh = p->leaf_hist;
	B3M_HIP(hipMemcpyAsync(pr, d_pair, 16, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipMemcpyAsync(hh, d_hist, 8 * 256, hipMemcpyDeviceToHost, s));
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	if (p0_out) *p0_out = pr[0];
	if (hist5_out) for (int c = 0; c < 5; ++c) hist5_out[c] = hh[c];
	return 0;
	B3M_CATCH(1)
}

/* the 256 symbol counts of the text of the leaf sorted last (byte alphabets: the C arrays of the merges; hist5_out of b3m_dist_leaf holds the first five) */
int b3m_dist_leaf_hist(const b3m_dist_plan *p, uint64_t *hist256_out)
{
	if (!p || !hist256_out) { set_error("null argument"); return 1; }
	memcpy(hist256_out, p->leaf_hist, 8 * 256);
	return 0;
}

/* ranks of the rotations starting at positions[] (host) among the suffixes of the last leaf sorted here: the chain anchors (z-ranks) */
int b3m_dist_rank_of(b3m_dist_plan *p, const uint64_t *positions, uint64_t count, uint64_t *ranks_out, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && p->d_sa_last, "b3m_dist_rank_of: no leaf sorted yet");
	hipStream_t s = (hipStream_t)stream;
	const TextView T = dist_text(p);
	std::vector<uint32_t> h;
	for (uint64_t o = 0; o < count; o += 4096) {
		const uint64_t c = std::min<uint64_t>(4096, count - o);
		uint64_t *d_pos = p->d_small + 136; uint32_t *d_rk = (uint32_t *)(p->d_small + 136 + 4096);
		B3M_HIP(hipMemcpyAsync(d_pos, positions + o, 8 * c, hipMemcpyHostToDevice, s));
		B3M_HIP(hipMemsetAsync(d_rk, 0, 4 * c, s));
		rank_of_positions(T, p->d_sa_last, p->sa_len, d_pos, c, d_rk, s);
		h.resize(c);
		B3M_HIP(hipMemcpyAsync(h.data(), d_rk, 4 * c, hipMemcpyDeviceToHost, s));
		B3M_HIP(hipStreamSynchronize(s));
		for (uint64_t i = 0; i < c; ++i) ranks_out[o + i] = h[i];
	}
	return 0;
	B3M_CATCH(1)
}

int b3m_dist_rank_of_dev(b3m_dist_plan *p, const uint64_t *positions, uint64_t count, int64_t *d_accum, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && p->d_sa_last && (count == 0 || (positions && d_accum)), "b3m_dist_rank_of_dev: no leaf sorted yet / null argument");
	hipStream_t s = (hipStream_t)stream;
	const TextView T = dist_text(p);
	for (uint64_t o = 0; o < count; o += 4096) {
		const uint64_t c = std::min<uint64_t>(4096, count - o);
		uint64_t *d_pos = p->d_small + 136; uint32_t *d_rk = (uint32_t *)(p->d_small + 136 + 4096);
		B3M_HIP(hipMemcpyAsync(d_pos, positions + o, 8 * c, hipMemcpyHostToDevice, s));
		B3M_HIP(hipMemsetAsync(d_rk, 0, 4 * c, s));
		rank_of_positions(T, p->d_sa_last, p->sa_len, d_pos, c, d_rk, s);
		hipLaunchKernelGGL(k_add_u32_to_i64, dim3((uint32_t)((c + 255) / 256)), dim3(256), 0, s, d_rk, reinterpret_cast<long long *>(d_accum) + o, c);
		B3M_HIP(hipStreamSynchronize(s));            // (`positions` is the caller's host memory, the staging area is reused by the next piece)
	}
	return 0;
	B3M_CATCH(1)
}

int b3m_dist_endbit(b3m_dist_plan *p, uint32_t *d_gt_abs, uint64_t bit, const int64_t *d_anchor, uint64_t p0L, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && d_gt_abs && d_anchor && bit <= p->n, "b3m_dist_endbit: bad argument");
	hipLaunchKernelGGL(k_endbit_i64, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<const long long *>(d_anchor), p0L, d_gt_abs, bit);
	return 0;
	B3M_CATCH(1)
}

int b3m_bits_copy(const uint32_t *d_src_abs, uint32_t *d_dst_abs, uint64_t bit_begin, uint64_t bit_end, void *stream)
{
	B3M_TRY
	require_device();
	B3M_REQUIRE(d_src_abs && d_dst_abs, "b3m_bits_copy: null argument");
	copy_bits(d_src_abs, d_dst_abs, bit_begin, bit_end, (hipStream_t)stream);
	return 0;
	B3M_CATCH(1)
}

int b3m_dist_pack2(b3m_dist_plan *p, const uint8_t *d_bwt, uint64_t m, uint8_t *d_packed, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && (m == 0 || (d_bwt && d_packed)), "b3m_dist_pack2: null argument");
	if (p->type == B3M_INPUT_BYTESTREAM) { if (m) B3M_HIP(hipMemcpyAsync(d_packed, d_bwt, m, hipMemcpyDeviceToDevice, (hipStream_t)stream)); return 0; }      // byte alphabets travel as they are
	pack2(d_bwt, m, p->type, d_packed, (hipStream_t)stream);
	return 0;
	B3M_CATCH(1)
}

/* rank dictionary of BWT(L) (|L| = m, p0L = rank of L's first suffix) from the packed slices of L's owners, in rank order */
int b3m_dist_build_dict(b3m_dist_plan *p, const uint8_t *const *d_parts, const uint64_t *part_syms, int nparts, uint64_t m, uint64_t p0L, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && d_parts && part_syms && m >= 1 && m <= p->max_L, "b3m_dist_build_dict: bad argument");
	if (p->type == B3M_INPUT_BYTESTREAM) {
		// byte alphabet: the slices side by side as the aligned copy of BWT(L), then the count rows every 64 symbols
		hipStream_t s = (hipStream_t)stream;
		uint64_t at = 0;
		for (int i = 0; i < nparts; ++i) { if (part_syms[i]) B3M_HIP(hipMemcpyAsync(p->d_lcopy + at, d_parts[i], part_syms[i], hipMemcpyDeviceToDevice, s)); at += part_syms[i]; }
		B3M_REQUIRE(at == m, "b3m_dist_build_dict: the slices do not add up to |L|");
		B3M_HIP(hipMemsetAsync(p->d_lcopy + m, 0, al256(m) + 256 - m, s));
		const uint32_t p032 = (uint32_t)p0L;
		B3M_HIP(hipMemcpyAsync(p->d_p0u, &p032, 4, hipMemcpyHostToDevice, s));
		byte_rank_build(p->d_lcopy, m, p->d_p0u, p->d_blocks, p->d_th, s);
		p->dict_m = m; p->dict_p0 = p0L;
		return 0;
	}
	rank_build_from_parts(d_parts, part_syms, nparts, m, p0L, p->d_blocks, p->d_scratch, p->d_scan_ws, (hipStream_t)stream);
	p->dict_m = m; p->dict_p0 = p0L;
	return 0;
	B3M_CATCH(1)
}

/* this rank's chains of the merge (L = [a, a+m), R = [a+m, a+m+mr)) walked through the dictionary: ascending ranks of the walked suffixes of R in
 * d_sorted_out (capacity num_chains*chain_len), their number in *count_out, and cuts_out[j] = #{ranks < bounds[j]} (host; bounds = borders of the
 * rank slices of the group: the gap buckets of the all-to-all).  symstart[4]: C array of L (symbols smaller than code c in L's text).
 * gt arrays as b3m_gap_chains.  One host synchronisation (the bucket sizes are split sizes of the collective). */
int b3m_dist_walk(b3m_dist_plan *p, uint64_t a, uint64_t m, uint64_t mr, const uint64_t *symstart, const uint32_t *d_gt_abs_in, uint32_t *d_gt_abs_out,
                  const int64_t *d_anchors, uint64_t chain_first, uint64_t num_chains, uint64_t chain_len,
                  const uint64_t *bounds, int nbounds, uint32_t *d_sorted_out, uint64_t *cuts_out, uint64_t *count_out, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && symstart && d_gt_abs_in && d_gt_abs_out && count_out && (num_chains == 0 || d_anchors), "b3m_dist_walk: null argument");
	B3M_REQUIRE(p->dict_m == m, "b3m_dist_walk: the dictionary in the workspace is not this merge's");
	B3M_REQUIRE(a + m + mr <= p->n && mr >= 1 && chain_len >= 1 && nbounds >= 0 && nbounds <= 64, "b3m_dist_walk: range out of bounds");
	const uint64_t total_chains = (mr + chain_len - 1) / chain_len;
	B3M_REQUIRE(chain_first + num_chains <= total_chains, "chain range out of bounds");
	hipStream_t s = (hipStream_t)stream;
	*count_out = 0;
	if (num_chains == 0) { for (int j = 0; j < nbounds; ++j) cuts_out[j] = 0; return 0; }
	const uint64_t slots = gap_walk_slots(num_chains, chain_len);
	B3M_REQUIRE(slots <= p->max_R + p->max_R / 8 + 256 && num_chains <= p->anchors_cap, "b3m_dist_walk: chain range larger than the workspace");
	const TextView T = dist_text(p);
	// the anchors (ranks of the chain tops in L, summed over L's leaves and all-reduced) arrive as a device array
	hipLaunchKernelGGL(k_i64_to_u32, dim3((uint32_t)((num_chains + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const long long *>(d_anchors), p->d_anchors, num_chains);
	const uint32_t p032 = (uint32_t)p->dict_p0;
	B3M_HIP(hipMemcpyAsync(p->d_p0u, &p032, 4, hipMemcpyHostToDevice, s));
	if (nbounds) B3M_HIP(hipMemcpyAsync(p->d_small + 8, bounds, 8 * nbounds, hipMemcpyHostToDevice, s));
	if (p->type == B3M_INPUT_BYTESTREAM) {
		uint32_t hC[256];
		for (int c = 0; c < 256; ++c) hC[c] = (uint32_t)symstart[c];
		B3M_HIP(hipMemcpyAsync(p->d_C, hC, 1024, hipMemcpyHostToDevice, s));
		B3M_HIP(hipStreamSynchronize(s));                  // hC is on this stack frame
		gap_walk_bytes(T, p->d_blocks, p->d_lcopy, p->d_p0u, p->d_C, a, m, mr, d_gt_abs_in, d_gt_abs_out, p->d_anchors, num_chains, chain_len, p->d_r0, s, chain_first);
	} else {
	RankDict rd{};
	rd.blocks = p->d_blocks; rd.m = m; rd.p0ptr = p->d_p0u;
	for (int c = 0; c < 4; ++c) rd.C[c] = (uint32_t)symstart[c];
	GapWalkArgs G{};
	G.T = T; G.rd = rd; G.a = a; G.m = m; G.mr = mr; G.gt_cur = d_gt_abs_in; G.gt_next = d_gt_abs_out;
	G.anchors = p->d_anchors; G.num_chains = num_chains; G.chain_len = chain_len; G.chain_first = chain_first; G.ranks_out = p->d_r0;
	gap_walk(G, s);
	}
	uint32_t *sr = nullptr;
	int nbits = 1; while ((1ull << nbits) <= m + 1 && nbits < 32) ++nbits;
	radix_sort_u32(p->d_r0, p->d_r1, slots, nbits, p->radix, s, nullptr, &sr, true);
	const uint64_t hi = mr - chain_first * chain_len, last = chain_first + num_chains;
	const uint64_t lo = (last * chain_len >= mr) ? 0 : mr - last * chain_len;
	const uint64_t cnt = hi - lo;
	if (d_sorted_out) B3M_HIP(hipMemcpyAsync(d_sorted_out, sr, 4 * cnt, hipMemcpyDeviceToDevice, s));
	if (nbounds) {
		rank_cuts(sr, cnt, p->d_small + 8, nbounds, p->d_small + 72, s);
		B3M_HIP(hipMemcpyAsync(cuts_out, p->d_small + 72, 8 * nbounds, hipMemcpyDeviceToHost, s));
	}
	B3M_HIP(hipStreamSynchronize(s));
	B3M_HIP(hipGetLastError());
	*count_out = cnt;
	return 0;
	B3M_CATCH(1)
}

/* bytes of BWT(L)[l0, l1) from the dictionary in the workspace; L = [a, a + dict_m): its p0 entry belongs to position a - 1 (circular), whose symbol
 * the 2-bit dictionary cannot carry (the terminator): it is read from the text */
int b3m_dist_l_bytes(b3m_dist_plan *p, uint64_t a, uint64_t l0, uint64_t l1, uint8_t *d_out, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && a < p->n && l0 <= l1 && l1 <= p->dict_m && (l1 == l0 || d_out), "b3m_dist_l_bytes: bad argument");
	if (p->type == B3M_INPUT_BYTESTREAM) { if (l1 > l0) B3M_HIP(hipMemcpyAsync(d_out, p->d_lcopy + l0, l1 - l0, hipMemcpyDeviceToDevice, (hipStream_t)stream)); return 0; }
	dict_bytes(dist_text(p), p->d_blocks, l0, l1, p->dict_p0, a ? a - 1 : p->n - 1, d_out, (hipStream_t)stream);
	return 0;
	B3M_CATCH(1)
}

/* bytes [q0, q1) of BWT(R) out of R's packed slices (the same all-gather that brought L's): an owner's range of the right side */
int b3m_dist_r_bytes(b3m_dist_plan *p, const uint8_t *const *d_parts, const uint64_t *part_syms, int nparts, uint64_t q0, uint64_t q1, uint8_t *d_out, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && d_parts && part_syms && nparts >= 1 && q0 <= q1 && (q1 == q0 || d_out), "b3m_dist_r_bytes: bad argument");
	hipStream_t s = (hipStream_t)stream;
	if (p->type == B3M_INPUT_BYTESTREAM) {
		uint64_t at = 0;
		for (int i = 0; i < nparts; ++i) {
			const uint64_t b = std::max(q0, at), e = std::min(q1, at + part_syms[i]);
			if (e > b) B3M_HIP(hipMemcpyAsync(d_out + (b - q0), d_parts[i] + (b - at), e - b, hipMemcpyDeviceToDevice, s));
			at += part_syms[i];
		}
		B3M_REQUIRE(q1 <= at, "b3m_dist_r_bytes: range outside the slices");
		return 0;
	}
	parts_bytes(d_parts, part_syms, nparts, q0, q1, p->type, d_out, s);
	return 0;
	B3M_CATCH(1)
}

/* ascending sort of the rank buckets this rank received (all keys < 2^nbits), in the workspace */
int b3m_dist_sort_u32(b3m_dist_plan *p, const uint32_t *d_keys, uint64_t count, int nbits, uint32_t *d_out, void *stream)
{
	B3M_TRY
	hipStream_t s = (hipStream_t)stream;
	if (count == 0) return 0;
	B3M_REQUIRE(p && d_keys && d_out && nbits >= 1 && nbits <= 32, "b3m_dist_sort_u32: bad argument");
	B3M_REQUIRE(count <= p->max_R + p->max_R / 8 + 256, "b3m_dist_sort_u32: more keys than the workspace holds");
	B3M_HIP(hipMemcpyAsync(p->d_r0, d_keys, 4 * count, hipMemcpyDeviceToDevice, s));
	uint32_t *res = nullptr;
	radix_sort_u32(p->d_r0, p->d_r1, count, nbits, p->radix, s, nullptr, &res, true);
	B3M_HIP(hipMemcpyAsync(d_out, res, 4 * count, hipMemcpyDeviceToDevice, s));
	return 0;
	B3M_CATCH(1)
}

/* interleave of this rank's slice: d_sr = its (slice-relative) sorted ranks, m symbols of BWT(L), mr of BWT(R) */
int b3m_dist_interleave(b3m_dist_plan *p, const uint32_t *d_sr, const uint8_t *d_bwtL, uint64_t m, const uint8_t *d_bwtR, uint64_t mr, uint8_t *d_out, void *stream)
{
	B3M_TRY
	B3M_REQUIRE(p && m + mr <= p->max_L + p->max_R + 1, "b3m_dist_interleave: slice larger than the workspace");
	interleave(d_sr, d_bwtL, m, d_bwtR, mr, d_out, p->d_tile_q, (hipStream_t)stream);
	return 0;
	B3M_CATCH(1)
}

}  // extern "C"
