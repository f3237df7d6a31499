// compute_bwt.cpp -- b3m_compute_bwt: the whole job of BwtMergeSort::computeBwt
// (/root/reference/src/bwtb3m.cpp:63; steps of SURVEY.md 3.1): read the input file, keep the
// packed text resident in HBM, run the plan, write X.bwt (RL), X.hist, X.meta, X.preisa | X.sa X.isa.
//
// Three stages, each overlapped inside as far as its data allow:
//   in   : the file goes straight to HBM (pread into pinned staging pieces by a few threads, one HIP stream each) WHILE the plan is created;
//   sort : num_gpus = 1: b3m_plan_run.  num_gpus = N > 1 ("one host thread per GPU drives its stream(s)"; the reference's own
//          parallelism knob behind this call is numthreads=, /root/reference/src/bwtb3m.cpp:48-50): one host thread per device, each with the
//          plan of one suffix-range shard (b3m_plan_create_sharded: no exchange step; long repeats: the shards' lists of tied suffixes are gathered by peer
//          copies and every shard finishes the union, b3m_plan_list_*), the slices gathered on the first device;
//   out  : two host threads: X.bwt (run-length / Huffman coded from the BWT in HBM, streamed to the file in pieces) beside X.sa / X.isa (LF walks on a
//          stream of their own, the samples streamed from HBM) or X.preisa (the plan's pinned sample image, written in place).
#include "formats.h"
#include "parfile.h"
#include "../switches.h"
#include "../../../include/b3m.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <stdexcept>
#include <thread>
#include <unistd.h>

namespace b3m { void set_error(const std::string &msg); }

namespace {
struct Log {
	int fd;
	void operator()(const char *fmt, ...) const
	{
		if (fd < 0) return;
		char buf[1024];
		va_list ap; va_start(ap, fmt);
		int k = vsnprintf(buf, sizeof buf, fmt, ap);
		va_end(ap);
		if (k > 0) { ssize_t w = write(fd, buf, (size_t)std::min<int>(k, (int)sizeof buf - 1)); (void)w; }
	}
};
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define B3MCHK(x) do { if (x) throw std::runtime_error(b3m_last_error()); } while (0)

// fn(k) on one host thread per k in [0, count); the first failure is rethrown here (b3m_last_error is thread-local: the text travels with the exception)
void on_threads(int count, const std::function<void(int)> &fn)
{
	std::vector<std::string> err((size_t)count);
	std::vector<std::thread> th;
	for (int k = 0; k < count; ++k)
		th.emplace_back([&, k]() { try { fn(k); } catch (const std::exception &e) { err[(size_t)k] = *e.what() ? e.what() : "unknown error"; } });
	for (auto &t : th) t.join();
	for (int k = 0; k < count; ++k) if (!err[(size_t)k].empty()) throw std::runtime_error(count > 1 ? "GPU worker " + std::to_string(k) + ": " + err[(size_t)k] : err[(size_t)k]);
}

// what the construction stage leaves for the output stage
struct Built {
	int dev = 0;                         // device that holds d_bwt and d_text
	uint8_t *d_bwt = nullptr;            // n symbols (symbols32: n uint32)
	const void *d_text = nullptr;        // packed text on `dev` (LF walks)
	b3m_plan *plan = nullptr;            // one GPU: the plan (its samples, its LF walk); several: null
	uint64_t p0 = 0, num_blocks = 1, levels = 0;
	uint64_t hist[256] = {0};
	const uint64_t *pairs = nullptr; uint64_t npairs = 0;     // (rank,pos) samples in position order, host memory
	const uint64_t *d_pairs = nullptr;                        // or (X.preisa only) their image in HBM
	uint64_t *d_pairs_own = nullptr;                          // ... which this struct owns (several GPUs: the shards' tables put together on the first device)
	std::vector<uint64_t> pairs_store;
	double ms_sort = 0, ms_gap = 0, ms_interleave = 0;
};

struct Shard {
	int dev = 0; b3m_plan *plan = nullptr; uint8_t *d_out = nullptr; b3m_result res{};
	uint64_t tied = 0, placed = 0, depth = 0;                // what the shard's run left on its list (b3m_plan_list_counts)
	uint64_t *d_tied = nullptr, *d_placed = nullptr;         // its records (b3m_plan_list_export)
	uint64_t *d_tied_all = nullptr, *d_placed_all = nullptr; // the union on this shard's device (b3m_plan_list_finish)
	uint8_t *d_full = nullptr;                               // the whole BWT on this shard's device (bwtonly=0: every device walks its share of the LF chains)
};

// the devices of an N-GPU run: 0 .. N-1, or what B3M_DEVICES lists (test switch: "0,0,0,0" runs four shards' worth of host threads, plans and peer
// copies on one GPU)
std::vector<int> device_list(int N)
{
	int ndev = 0;
	HIPCHK(hipGetDeviceCount(&ndev));
	std::vector<int> d;
	if (const char *e = b3m::sw().DEVICES[0] ? b3m::sw().DEVICES : nullptr) {
		for (const char *p = e; *p;) { char *q; long v = strtol(p, &q, 10); if (q == p) break; d.push_back((int)v); p = *q == ',' ? q + 1 : q; }
		if ((int)d.size() != N) throw std::runtime_error("B3M_DEVICES must list exactly numgpus devices");
		for (int v : d) if (v < 0 || v >= ndev) throw std::runtime_error("B3M_DEVICES names a device this machine does not have");
		return d;
	}
	if (N > ndev) throw std::runtime_error("numgpus=" + std::to_string(N) + " but this machine has " + std::to_string(ndev) + " GPU(s)");
	for (int k = 0; k < N; ++k) d.push_back(k);
	return d;
}

void free_shards(std::vector<Shard> &sh)
{
	for (auto &s : sh) {
		(void)hipSetDevice(s.dev);
		if (s.plan) b3m_plan_destroy(s.plan);
		for (void *p : {(void *)s.d_out, (void *)s.d_tied, (void *)s.d_placed, (void *)s.d_tied_all, (void *)s.d_placed_all, (void *)s.d_full}) if (p) (void)hipFree(p);
		s = Shard{};
	}
}

// N suffix-range shards, one host thread and one device each; d_text0: the packed text on devs[0].  Leaves the whole BWT on devs[0].
// gather: 0 = the slices stay where they are (bwtonly=1: every device codes its own part of X.bwt), 1 = the whole BWT on the first device (the form before round 5),
// 2 = on EVERY device (bwtonly=0: every device also walks its share of the LF chains)
void construct_sharded(const std::vector<int> &devs, int type, uint64_t n, uint64_t mem, uint64_t rate, const void *d_text0, uint64_t tb, Built &B, std::vector<Shard> &sh,
                       const Log &log, int gather)
{
	const int N = (int)devs.size();
	sh.assign((size_t)N, Shard{});
	for (int k = 0; k < N; ++k) sh[(size_t)k].dev = devs[(size_t)k];
	// direct peer access between the devices of the run (xGMI), where the platform grants it: the text, the slices and -- long repeats -- the suffix array
	// slices travel by hipMemcpyPeer, which falls back to staging through the host without it
	{
		std::map<int, int> distinct;
		for (int d : devs) distinct[d] = 1;
		for (auto &a : distinct) {
			HIPCHK(hipSetDevice(a.first));
			for (auto &b : distinct) if (b.first != a.first) { int can = 0; if (hipDeviceCanAccessPeer(&can, a.first, b.first) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(b.first, 0); (void)hipGetLastError(); }
		}
	}
	auto run_shard = [&](Shard &s) {
		B3MCHK(b3m_plan_run(s.plan, b3m_plan_text_buffer(s.plan), s.d_out, nullptr));
		B3MCHK(b3m_plan_result(s.plan, &s.res));
	};
	const double t0 = now();
	on_threads(N, [&](int k) {
		Shard &s = sh[(size_t)k];
		HIPCHK(hipSetDevice(s.dev));
		s.plan = b3m_plan_create_sharded(n, type, k, N, mem, rate);
		if (!s.plan) throw std::runtime_error(b3m_last_error());
		HIPCHK(hipMalloc((void **)&s.d_out, b3m_plan_slice_capacity(s.plan) + 16));
		HIPCHK(hipMemcpyPeer(b3m_plan_text_buffer(s.plan), s.dev, d_text0, devs[0], tb));
		B3MCHK(b3m_plan_set_tie_mode(s.plan, B3M_TIES_LIST));
		run_shard(s);
	});
	const double t1 = now();
	// long repeats (row a6): every shard stopped after a few key-extension rounds and kept a LIST of what was left tied (and of what its direct comparisons placed
	// among long repeats); the lists are gathered by peer copies -- 16 bytes per listed suffix, not the suffix arrays -- and every shard finishes the union by prefix
	// doubling on it (b3m_plan_list_*), then repairs its own slice.  A union too large for a shard's workspace: the shards with leftovers sort again and finish them
	// themselves (B3M_TIES_LOCAL: key extension, correct but not bounded)
	uint64_t total = 0, placed = 0, dmin = ~0ull, dmax = 0; int complete = 1;
	for (auto &s : sh) {
		int c = 1;
		B3MCHK(b3m_plan_list_counts(s.plan, &s.tied, &s.placed, &s.depth, &c));
		total += s.tied; placed += s.placed; complete &= c;
		if (s.tied) { dmin = std::min(dmin, s.depth); dmax = std::max(dmax, s.depth); }
	}
	bool local = false;
	if (total) {
		bool fits = true;
		for (auto &s : sh) fits = fits && b3m_plan_list_fits(s.plan, total, placed);
		if (!fits) {
			local = true;
			on_threads(N, [&](int k) {
				Shard &s = sh[(size_t)k];
				if (!s.tied) return;
				HIPCHK(hipSetDevice(s.dev));
				B3MCHK(b3m_plan_set_tie_mode(s.plan, B3M_TIES_LOCAL));
				run_shard(s);
			});
		} else {
			on_threads(N, [&](int k) {
				Shard &s = sh[(size_t)k];
				HIPCHK(hipSetDevice(s.dev));
				HIPCHK(hipMalloc((void **)&s.d_tied, 16 * s.tied + 16));
				HIPCHK(hipMalloc((void **)&s.d_placed, 16 * s.placed + 16));
				B3MCHK(b3m_plan_list_export(s.plan, s.d_tied, s.d_placed, nullptr));
			});
			on_threads(N, [&](int k) {
				Shard &s = sh[(size_t)k];
				HIPCHK(hipSetDevice(s.dev));
				HIPCHK(hipMalloc((void **)&s.d_tied_all, 16 * total + 16));
				HIPCHK(hipMalloc((void **)&s.d_placed_all, 16 * placed + 16));
				uint64_t at = 0, xat = 0;
				for (int j = 0; j < N; ++j) {
					const Shard &o = sh[(size_t)j];
					if (o.tied) HIPCHK(hipMemcpyPeer(s.d_tied_all + 2 * at, s.dev, o.d_tied, o.dev, 16 * o.tied));
					if (o.placed) HIPCHK(hipMemcpyPeer(s.d_placed_all + 2 * xat, s.dev, o.d_placed, o.dev, 16 * o.placed));
					at += o.tied; xat += o.placed;
				}
				B3MCHK(b3m_plan_list_finish(s.plan, s.d_tied_all, total, s.d_placed_all, placed, dmin, dmax, complete, s.d_out, nullptr));
				B3MCHK(b3m_plan_result(s.plan, &s.res));
				(void)hipFree(s.d_tied_all); s.d_tied_all = nullptr;
				(void)hipFree(s.d_placed_all); s.d_placed_all = nullptr;
			});
		}
	}
	const double t2 = now();
	// the slices tile the rank space: gathered on the first device (the RL encoder and the LF walks read the whole BWT)
	uint64_t pos = 0;
	for (auto &s : sh) {
		if (s.res.slice_offset != pos) throw std::runtime_error("internal: the shards' slices do not tile the rank space");
		pos += s.res.slice_len;
	}
	if (pos != n) throw std::runtime_error("internal: the shards' slices do not cover the rank space");
	B.dev = devs[0];
	if (gather) on_threads(N, [&](int k) {
		Shard &d = sh[(size_t)k];
		if (gather == 1 && k) return;
		HIPCHK(hipSetDevice(d.dev));
		HIPCHK(hipMalloc((void **)&d.d_full, n + 16));
		for (const Shard &s : sh) if (s.res.slice_len) HIPCHK(hipMemcpyPeer(d.d_full + s.res.slice_offset, d.dev, s.d_out, s.dev, s.res.slice_len));
		HIPCHK(hipDeviceSynchronize());                   // (device-to-device copies return before they are done; the readers use streams that do not wait for the null stream)
	});
	B.d_bwt = sh[0].d_full;                              // (null without a gather; owned by the shard)
	HIPCHK(hipSetDevice(B.dev));
	B.d_text = b3m_plan_text_buffer(sh[0].plan);
	B.num_blocks = 1; B.levels = 0;
	int have_p0 = 0;
	for (auto &s : sh) {
		if (s.res.has_p0) { B.p0 = s.res.p0; ++have_p0; }
		B.ms_sort = std::max(B.ms_sort, s.res.ms_sort);
	}
	if (have_p0 != 1) throw std::runtime_error("internal: the rotation starting at position 0 was placed by " + std::to_string(have_p0) + " shards");
	HIPCHK(hipSetDevice(sh[0].dev));
	B3MCHK(b3m_plan_histogram(sh[0].plan, B.hist));          // (every shard's level 0 sees the whole text: any shard has the text's histogram)
	// samples: every sampled position got its rank from exactly one shard (slot i of each shard's pinned image = position i * rate)
	if (rate && gather != 2 && 16 * ((n + rate - 1) / rate) > (64ull << 20)) {      // (gather 2 = bwtonly=0: the LF walks read the pairs from host memory)
		// a large table (bwtonly=1 samples every 64th position: 750 MB at 3 Gbase) never comes to the host: the shards' device images are put together on the first
		// device and X.preisa is streamed from there, as on one GPU
		const uint64_t nslots = (n + rate - 1) / rate;
		HIPCHK(hipSetDevice(B.dev));
		HIPCHK(hipMalloc((void **)&B.d_pairs_own, 16 * nslots + 16));
		HIPCHK(hipMemset(B.d_pairs_own, 0xFF, 16 * nslots));
		uint64_t *tmp = nullptr;
		uint64_t stats[2] = {0, nslots};
		try {
			for (int k = 0; k < N; ++k) {
				const uint64_t *d_slots = nullptr; uint64_t c = 0;
				HIPCHK(hipSetDevice(sh[(size_t)k].dev));
				B3MCHK(b3m_plan_samples_device(sh[(size_t)k].plan, &d_slots, &c, nullptr, nullptr));
				if (c != nslots) throw std::runtime_error("internal: sample tables differ");
				HIPCHK(hipDeviceSynchronize());
				HIPCHK(hipSetDevice(B.dev));
				const uint64_t *src = d_slots;
				if (sh[(size_t)k].dev != B.dev) {
					if (!tmp) HIPCHK(hipMalloc((void **)&tmp, 16 * nslots + 16));
					HIPCHK(hipMemcpyPeer(tmp, B.dev, d_slots, sh[(size_t)k].dev, 16 * nslots));
					src = tmp;
				}
				B3MCHK(b3m_merge_sample_pairs(B.d_pairs_own, src, nslots, stats, nullptr));
			}
		} catch (...) { if (tmp) { (void)hipSetDevice(B.dev); (void)hipFree(tmp); } throw; }
		if (tmp) (void)hipFree(tmp);
		if (stats[0]) throw std::runtime_error("internal: " + std::to_string(stats[0]) + " sampled positions were ranked by two shards");
		if (stats[1]) throw std::runtime_error("internal: the (rank,pos) samples of " + std::to_string(stats[1]) + " positions were written by no shard");
		B.d_pairs = B.d_pairs_own; B.npairs = nslots;
	} else if (rate) {
		uint64_t nslots = 0; std::vector<const uint64_t *> raw((size_t)N, nullptr);
		for (int k = 0; k < N; ++k) { uint64_t c = 0; HIPCHK(hipSetDevice(sh[(size_t)k].dev)); B3MCHK(b3m_plan_samples_raw(sh[(size_t)k].plan, &raw[(size_t)k], &c)); if (k && c != nslots) throw std::runtime_error("internal: sample tables differ"); nslots = c; }
		B.pairs_store.assign(2 * nslots, ~0ull);
		for (int k = 0; k < N; ++k)
			for (uint64_t i = 0; i < nslots; ++i) if (raw[(size_t)k][2 * i] != ~0ull) {
				if (B.pairs_store[2 * i] != ~0ull) throw std::runtime_error("internal: a sampled position was ranked by two shards");
				B.pairs_store[2 * i] = raw[(size_t)k][2 * i]; B.pairs_store[2 * i + 1] = raw[(size_t)k][2 * i + 1];
			}
		for (uint64_t i = 0; i < nslots; ++i) if (B.pairs_store[2 * i] == ~0ull) throw std::runtime_error("internal: the (rank,pos) sample of position " + std::to_string(i * rate) + " was written by no shard");
		B.pairs = B.pairs_store.data(); B.npairs = nslots;
	}
	HIPCHK(hipSetDevice(B.dev));
	std::map<int, int> distinct;
	for (int d : devs) distinct[d] = 1;
	log("[V] %d suffix-range shards on %d device(s): plans + sort %.3f s, tie finish %.3f s (%llu tied + %llu placed suffixes on the shards' lists%s), %s %.3f s\n", N,
	    (int)distinct.size(), t1 - t0, t2 - t1, (unsigned long long)total, (unsigned long long)placed, local ? ": too many for a shard's workspace, finished shard by shard" : "",
	    gather == 2 ? "BWT gathered on every device" : gather == 1 ? "gather" : "slices stay on their devices, samples merged", now() - t2);
}

// the whole BWT on the first device after all (a part of the distributed output stage could not be used)
void gather_on_first(std::vector<Shard> &sh, uint64_t n, Built &B)
{
	Shard &d = sh[0];
	HIPCHK(hipSetDevice(d.dev));
	if (!d.d_full) {
		HIPCHK(hipMalloc((void **)&d.d_full, n + 16));
		for (const Shard &s : sh) if (s.res.slice_len) HIPCHK(hipMemcpyPeer(d.d_full + s.res.slice_offset, d.dev, s.d_out, s.dev, s.res.slice_len));
		HIPCHK(hipDeviceSynchronize());
	}
	B.d_bwt = d.d_full;
}

// X.bwt from the slices where they lie: a block of the container is runs_per_block runs counted from the start of the BWT, so the devices first count the run heads of
// their slices (one small exchange of counts through this host process: the role of an all-gather), every device finds where the first block that BEGINS in its slice
// starts, takes the symbols from there to the next device's first block (its own slice and the head of the following one, by peer copy) and codes them -- runs,
// frequencies, both code streams on the device, tables on its share of the host threads -- and writes its image at the offset the sizes in front of it give.  The
// file is the one-device file byte for byte.  false: not an input for the device encoder (see rl_encode_device); nothing usable was written.
bool encode_sharded(const std::string &fn, std::vector<Shard> &sh, uint64_t n, uint64_t rpb, unsigned numthreads, const int64_t *symmap, const Log &log)
{
	const int N = (int)sh.size();
	const double t0 = now();
	struct Part { std::vector<uint32_t> tile_first; uint8_t first = 0, last = 0; bool cont = false; uint64_t heads = 0, G = 0, b0 = 0, P = 0, Pend = 0; bool starts = false;
	              std::unique_ptr<b3m::RLPart> img; bool ok = true; uint64_t base = 0; };
	std::vector<Part> pt((size_t)N);
	std::vector<hipStream_t> st((size_t)N, nullptr);
	struct Streams { std::vector<Shard> &sh; std::vector<hipStream_t> &st; ~Streams() { for (size_t k = 0; k < st.size(); ++k) if (st[k]) { (void)hipSetDevice(sh[k].dev); (void)hipStreamDestroy(st[k]); } } } guard{sh, st};
	on_threads(N, [&](int k) {
		Shard &s = sh[(size_t)k];
		HIPCHK(hipSetDevice(s.dev));
		HIPCHK(hipStreamCreateWithFlags(&st[(size_t)k], hipStreamNonBlocking));
		b3m::rl_slice_scan(s.d_out, s.res.slice_len, st[(size_t)k], pt[(size_t)k].tile_first, &pt[(size_t)k].first, &pt[(size_t)k].last);
	});
	// run heads in front of every slice; the first block that begins inside it
	uint64_t G = 0; int prev = -1;
	for (int k = 0; k < N; ++k) {
		Part &p = pt[(size_t)k];
		if (!sh[(size_t)k].res.slice_len) continue;
		p.cont = prev >= 0 && pt[(size_t)prev].last == p.first;          // the slice goes on with its neighbour's last run: its position 0 is no head
		p.heads = (uint64_t)p.tile_first.back() - (p.cont ? 1 : 0);
		p.G = G; p.b0 = (G + rpb - 1) / rpb;
		p.starts = p.b0 * rpb < G + p.heads;
		G += p.heads; prev = k;
	}
	const uint64_t runs = G;
	on_threads(N, [&](int k) {
		Part &p = pt[(size_t)k]; Shard &s = sh[(size_t)k];
		if (!p.starts) return;
		HIPCHK(hipSetDevice(s.dev));
		p.P = s.res.slice_offset + b3m::rl_head_position(s.d_out, s.res.slice_len, p.tile_first, p.b0 * rpb - p.G + (p.cont ? 1 : 0), st[(size_t)k]);
	});
	{
		uint64_t end = n;
		for (int k = N - 1; k >= 0; --k) if (pt[(size_t)k].starts) { pt[(size_t)k].Pend = end; end = pt[(size_t)k].P; }
		if (end != 0) throw std::runtime_error("internal: the first block of the run-length container does not begin at symbol 0");
	}
	int nparts = 0;
	for (auto &p : pt) nparts += p.starts ? 1 : 0;
	const unsigned thr = std::max(1u, numthreads / (unsigned)std::max(1, nparts));
	on_threads(N, [&](int k) {
		Part &p = pt[(size_t)k]; Shard &d = sh[(size_t)k];
		if (!p.starts) return;
		HIPCHK(hipSetDevice(d.dev));
		const uint64_t len = p.Pend - p.P;
		uint8_t *buf = nullptr;
		HIPCHK(hipMalloc((void **)&buf, len + 16));
		try {
			for (const Shard &s : sh) {
				const uint64_t lo = std::max(p.P, s.res.slice_offset), hi = std::min(p.Pend, s.res.slice_offset + s.res.slice_len);
				if (lo < hi) HIPCHK(hipMemcpyPeerAsync(buf + (lo - p.P), d.dev, s.d_out + (lo - s.res.slice_offset), s.dev, hi - lo, st[(size_t)k]));
			}
			HIPCHK(hipStreamSynchronize(st[(size_t)k]));
			p.img.reset(new b3m::RLPart());
			p.ok = b3m::rl_encode_part(buf, len, rpb, thr, symmap, st[(size_t)k], *p.img);
		} catch (...) { (void)hipFree(buf); throw; }
		(void)hipFree(buf);
	});
	for (auto &p : pt) if (p.starts && !p.ok) return false;
	// offsets; every part holds whole blocks: as many as lie between its first block and the next part's
	std::vector<b3m::RLIndexEntry> index;
	uint64_t fpos = 8, nextb = (runs + rpb - 1) / rpb;
	for (int k = N - 1; k >= 0; --k) {
		Part &p = pt[(size_t)k];
		if (!p.starts) continue;
		if (p.img->index.size() != nextb - p.b0) throw std::runtime_error("internal: a device's part of the run-length container holds " + std::to_string(p.img->index.size()) +
		                                                                  " blocks, " + std::to_string(nextb - p.b0) + " begin in its range");
		nextb = p.b0;
	}
	for (int k = 0; k < N; ++k) {
		Part &p = pt[(size_t)k];
		if (!p.starts) continue;
		p.base = fpos;
		for (const b3m::RLIndexEntry &e : p.img->index) index.push_back({fpos + e.pos, e.runs, e.syms});
		fpos += p.img->bytes;
	}
	const double t1 = now();
	b3m::ParFile pf(fn);
	on_threads(N, [&](int k) {
		Part &p = pt[(size_t)k];
		if (!p.starts) return;
		uint64_t at = p.base;
		for (auto &im : p.img->images) { pf.put_device(at, im.first, im.second, p.img->dev, std::min(thr, 8u)); at += im.second; }
	});
	std::vector<uint8_t> head, tail;
	b3m::rl_container_ends(n, fpos - 8, index, head, tail);
	pf.put(0, head.data(), head.size());
	pf.put(fpos, tail.data(), tail.size());
	pf.close();
	log("[V] X.bwt coded by %d of %d devices, each the blocks that begin in its slice (%llu runs, %llu blocks): run heads + coding %.3f s, images to the file %.3f s\n", nparts, N,
	    (unsigned long long)runs, (unsigned long long)index.size(), t1 - t0, now() - t1);
	return true;
}

// X.sa's samples from the LF walks, every device its share of the chains over its own copy of the BWT (Shard::d_full) and the packed text of its plan; the shares are put
// together on the first device (d_sa: ceil(n / sarate) u64 there).  Every symbol and every chain is verified on the way, as on one device.
void walk_sharded(std::vector<Shard> &sh, uint64_t n, int type, const Built &B, uint64_t sarate, uint64_t *d_sa, uint64_t mm[2], const Log &log)
{
	const int N = (int)sh.size();
	const uint64_t nsa = (n + sarate - 1) / sarate, K = B.npairs;
	const double t0 = now();
	std::vector<uint64_t *> part((size_t)N, nullptr);
	std::vector<uint64_t> m0((size_t)N, 0), m1((size_t)N, 0);
	struct Free { std::vector<Shard> &sh; std::vector<uint64_t *> &p; ~Free() { for (size_t k = 1; k < p.size(); ++k) if (p[k]) { (void)hipSetDevice(sh[k].dev); (void)hipFree(p[k]); } } } guard{sh, part};
	part[0] = d_sa;
	on_threads(N, [&](int k) {
		Shard &s = sh[(size_t)k];
		const uint64_t c0 = K * (uint64_t)k / (uint64_t)N, c1 = K * (uint64_t)(k + 1) / (uint64_t)N;
		HIPCHK(hipSetDevice(s.dev));
		if (k) HIPCHK(hipMalloc((void **)&part[(size_t)k], 8 * nsa + 16));
		hipStream_t st = nullptr;
		HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
		b3m_lf_job j{};
		j.d_text = b3m_plan_text_buffer(s.plan); j.n = n; j.inputtype = type; j.d_bwt = s.d_full; j.p0 = B.p0; j.hist256 = B.hist;
		j.sample_pairs = B.pairs; j.num_samples = K; j.sarate = sarate; j.d_sa_out = part[(size_t)k];
		uint64_t m[2] = {0, 0};
		int rc = 0;
		if (c1 > c0) rc = b3m_lf_walk_chains(&j, c0, c1 - c0, m, st);
		else HIPCHK(hipMemset(part[(size_t)k], 0xFF, 8 * nsa));
		(void)hipStreamDestroy(st);
		if (rc) throw std::runtime_error(b3m_last_error());
		m0[(size_t)k] = m[0]; m1[(size_t)k] = m[1];
	});
	const double t1 = now();
	mm[0] = mm[1] = 0;
	for (int k = 0; k < N; ++k) { mm[0] += m0[(size_t)k]; mm[1] += m1[(size_t)k]; }
	HIPCHK(hipSetDevice(sh[0].dev));
	uint64_t *tmp = nullptr;
	if (N > 1) HIPCHK(hipMalloc((void **)&tmp, 8 * nsa + 16));
	try {
		for (int k = 1; k < N; ++k) {
			HIPCHK(hipMemcpyPeer(tmp, sh[0].dev, part[(size_t)k], sh[(size_t)k].dev, 8 * nsa));
			B3MCHK(b3m_merge_samples(d_sa, tmp, nsa, nullptr));
		}
	} catch (...) { if (tmp) (void)hipFree(tmp); throw; }
	if (tmp) (void)hipFree(tmp);
	log("[V] LF walks on %d devices, %llu chains each: %.3f s; SA samples put together on the first: %.3f s\n", N, (unsigned long long)(K / (uint64_t)N), t1 - t0, now() - t1);
}
}  // namespace

extern "C" uint64_t b3m_plan_sampling_rate(uint64_t isasamplingrate, int bwtonly, uint64_t n)
{
	const uint64_t isarate = isasamplingrate ? isasamplingrate : (bwtonly ? 64 : 262144);
	if (bwtonly) return isarate;
	// bwtonly=0: the samples are also the anchors of the LF walks that make .sa / .isa, one chain per sample.  The walks are fastest from ~10^5 chains on
	// (C3, tools/rate_sweep.py: 137 / 127 / 123 / 478 ms at rates 1024 / 4096 / 16384 / 262144), the sort pays for every sample (finish 12.4 / 11.4 / 11.2 / 11.1 ms
	// and 16 bytes to pinned host memory each): the sparsest of 16384, 4096, 1024 that divides the rate and still leaves 2^17 chains; short texts: 1024
	for (uint64_t r : {16384ull, 4096ull, 1024ull})
		if (isarate > r && isarate % r == 0 && (r == 1024 || n / r >= (1ull << 17))) return r;
	return isarate;
}

extern "C" int b3m_compute_bwt(const b3m_options *o, int log_fd)
{
	b3m::switches_load();
	void *d_text = nullptr; Built B; std::vector<Shard> shards; int dev0 = 0; bool own_bwt = false;
	void *h_text = nullptr; uint8_t *h_bwt = nullptr;      // host tier: the text comes from, the BWT goes to pinned host memory
	auto cleanup = [&]() {
		free_shards(shards);
		(void)hipSetDevice(dev0);
		if (B.plan) b3m_plan_destroy(B.plan);
		if (d_text) (void)hipFree(d_text);
		if (own_bwt && B.d_bwt) (void)hipFree(B.d_bwt);
		if (B.d_pairs_own) (void)hipFree(B.d_pairs_own);
		if (h_text) b3m_host_free(h_text);
		if (h_bwt) b3m_host_free(h_bwt);
		b3m::staging_release();
	};
	try {
		if (!o || !o->input_fn || !*o->input_fn) throw std::runtime_error("no input file given");
		Log log{o->verbose ? log_fd : -1};
		const char *tname = (o->inputtype && *o->inputtype) ? o->inputtype : "bytestream";
		int type = b3m_parse_input_type(tname);
		if (type < 0) throw std::runtime_error(b3m_last_error());
		if (type != B3M_INPUT_BYTESTREAM && type != B3M_INPUT_PAC && type != B3M_INPUT_PACTERM && type != B3M_INPUT_COMPACTSTREAM && type != B3M_INPUT_UTF8)
			throw std::runtime_error(std::string("inputtype=") + tname + " is not supported by this engine yet (supported: bytestream, compactstream, pac, pacterm, utf-8)");
		std::string outfn = (o->output_fn && *o->output_fn) ? o->output_fn : (std::string((o->tmpprefix && *o->tmpprefix) ? o->tmpprefix : "bwtb3m_out") + ".bwt");
		std::string prefix = b3m::clip_off(outfn, ".bwt");
		const unsigned io_threads = std::max(1u, std::min(o->numthreads ? o->numthreads : 1u, 8u));
		const int ngpus = o->num_gpus > 1 ? o->num_gpus : 1;
		double t0 = now();
		// ---- in: what kind of file, how many symbols ----------------------------------------------------------------------------
		const int filetype = type;
		const std::string infn = o->input_fn;
		const uint64_t fsize = b3m::file_size(infn);
		std::vector<uint8_t> file;                            // only the types that are decoded on the host keep a host image
		uint64_t n = 0;
		// compactstream (row f4): 2-bit files are a .pac payload behind a 32-byte header; other widths are decoded to bytes on the device
		uint32_t cbits = 0; uint64_t coff = 0;
		int64_t symmap_store[256]; const int64_t *symmap = nullptr;
		std::vector<int64_t> wide_map;                        // utf-8 above 256 distinct code points: rank -> code point (the text goes as symbols32)
		const bool direct = type == B3M_INPUT_BYTESTREAM || type == B3M_INPUT_PAC || type == B3M_INPUT_PACTERM;   // the file's bytes ARE the packed text
		if (type == B3M_INPUT_COMPACTSTREAM) {
			file = b3m::read_file(infn);
			if (b3m_compact_info(file.data(), file.size(), &cbits, &n, &coff)) throw std::runtime_error(b3m_last_error());
			file.erase(file.begin(), file.begin() + (long)coff);
			file.resize(file.size() + 16, 0);
			type = cbits == 2 ? B3M_INPUT_PAC : B3M_INPUT_BYTESTREAM;
		} else if (type == B3M_INPUT_UTF8) {
			// utf-8 (row f4): the text is the sequence of code points.  The code points that occur are ranked (order kept) into a byte
			// alphabet, the ranks are sorted and merged as a bytestream, and the files name the code points again (symmap).
			file = b3m::read_file(infn);
			std::vector<uint32_t> cps = b3m::utf8_decode(file);
			std::vector<uint8_t> seen(0x110000, 0);
			for (uint32_t c : cps) seen[c] = 1;
			std::vector<int32_t> rank(0x110000, -1);
			uint32_t k = 0;
			for (uint32_t c = 0; c < 0x110000; ++c) if (seen[c]) { if (k < 256) symmap_store[k] = (int64_t)c; rank[c] = (int32_t)k++; }
			n = cps.size();
			if (k > 256) {
				// more than a byte alphabet holds: the ranks go as 32-bit symbols (one block, key/index sort over three 21-bit symbols per key)
				if (n >= (1ull << 32)) throw std::runtime_error("inputtype=utf-8: texts of more than 256 distinct code points must stay below 2^32 code points");
				wide_map.resize(k);
				for (uint32_t c = 0; c < 0x110000; ++c) if (rank[c] >= 0) wide_map[(size_t)rank[c]] = (int64_t)c;
				file.assign(4 * cps.size() + 16, 0);
				uint32_t *w = reinterpret_cast<uint32_t *>(file.data());
				for (size_t i = 0; i < cps.size(); ++i) w[i] = (uint32_t)rank[cps[i]];
				type = B3M_INPUT_SYMBOLS32;
			} else {
				for (uint32_t j = k; j < 256; ++j) symmap_store[j] = (int64_t)0x110000 + j;      // (unused ranks: never written)
				file.resize(cps.size());
				for (size_t i = 0; i < cps.size(); ++i) file[i] = (uint8_t)rank[cps[i]];
				symmap = symmap_store;
				type = B3M_INPUT_BYTESTREAM;
				file.resize(file.size() + 16, 0);
			}
		} else if (type == B3M_INPUT_BYTESTREAM) n = fsize;
		else {
			uint8_t last = 255;
			if (fsize >= 2) b3m::read_range(infn, fsize - 1, &last, 1, 1);
			if (fsize < 2 || last > 3) throw std::runtime_error(std::string(o->input_fn) + " is not a .pac file");
			n = (fsize - 2) * 4 + last + (type == B3M_INPUT_PACTERM ? 1 : 0);
		}
		if (n == 0) throw std::runtime_error("input is empty");
		log("[V] input %s type %s symbols %llu\n", o->input_fn, tname, (unsigned long long)n);
		// bwtonly=1 lowers the default ISA sampling rate to 64 (/root/reference/ChangeLog:281); 0 means "default for the mode"
		uint64_t isarate = o->isasamplingrate ? o->isasamplingrate : (o->bwtonly ? 64 : 262144);
		uint64_t sarate = o->sasamplingrate ? o->sasamplingrate : 32;
		const bool want_sa = !o->bwtonly;                     // sampled SA/ISA stage (byte alphabets walk the LF array)
		// anchors of the LF walks: a divisor of the output rate, dense enough to fill the GPU
		const uint64_t anchor_rate = b3m_plan_sampling_rate(o->isasamplingrate, (int)o->bwtonly, n);
		// byte alphabets are sorted as ONE block (no byte-alphabet merge): a memory target that would force several
		// blocks (the reference's scripts pass mem= for host RAM) is not applied to them
		uint64_t mem = o->mem;
		const bool wide = type == B3M_INPUT_SYMBOLS32;
		if ((type == B3M_INPUT_BYTESTREAM || wide) && mem) { log("[V] mem= is not applied to bytestream input (sorted as one block in HBM)\n"); mem = 0; }
		const bool two_bit = type == B3M_INPUT_PAC || type == B3M_INPUT_PACTERM;
		const bool sharded = ngpus > 1;
		if (sharded && !(two_bit && n >= (1ull << 18) && !o->blocksize))
			throw std::runtime_error("numgpus > 1 constructs pac / pacterm texts of at least 2^18 symbols as suffix-range shards (no blocksize=)");
		std::vector<int> devs;
		if (sharded) { devs = device_list(ngpus); dev0 = devs[0]; } else HIPCHK(hipGetDevice(&dev0));
		HIPCHK(hipSetDevice(dev0));
		// ---- in: the text to HBM, beside the creation of the plan -------------------------------------------------------------------
		const double tp0 = now();
		uint64_t tb = 0; double t_read = 0;
		// with a memory target the plan comes first: it says whether the text goes to HBM or -- host tier: no merge tree meets the target, the text is sorted as
		// sequential suffix-range shards whose slices leave for pinned host memory -- is handed over from host memory (b3m_plan_run_host)
		int spill = 0;
		if (!sharded && mem && direct && two_bit) {
			B.plan = b3m_plan_create(n, type, o->blocksize, mem, anchor_rate);
			if (!B.plan) throw std::runtime_error(b3m_last_error());
			spill = b3m_plan_host_tier(B.plan);
		}
		if (spill) {
			tb = b3m_plan_text_bytes(B.plan);
			if (tb != fsize) throw std::runtime_error("internal: packed size mismatch");
			h_text = b3m_host_alloc(tb + 64);
			h_bwt = (uint8_t *)b3m_host_alloc(n + 16);
			if (!h_text || !h_bwt) throw std::runtime_error("out of pinned host memory for the host tier (text and BWT)");
			const double r0 = now();
			b3m::read_range(infn, 0, h_text, tb, io_threads);
			t_read = now() - r0;
		} else
		if (filetype == B3M_INPUT_COMPACTSTREAM && cbits != 2) {
			void *d_words = nullptr;
			HIPCHK(hipMalloc(&d_words, file.size()));
			HIPCHK(hipMemcpy(d_words, file.data(), file.size(), hipMemcpyHostToDevice));
			HIPCHK(hipMalloc(&d_text, n + 16));
			if (b3m_compact_decode(d_words, n, cbits, (uint8_t *)d_text, nullptr)) { (void)hipFree(d_words); throw std::runtime_error(b3m_last_error()); }
			HIPCHK(hipDeviceSynchronize());
			(void)hipFree(d_words);
			tb = n;
		} else {
			// bytes of the packed text (include/b3m.h: bytestream n, symbols32 4 n, pac / pacterm the BWA .pac size; checked against the plan's own figure below)
			tb = wide ? 4 * n : (two_bit ? (type == B3M_INPUT_PACTERM ? n - 1 : n) / 4 + 2 : n);
			HIPCHK(hipMalloc(&d_text, tb + 64));
			HIPCHK(hipMemset((uint8_t *)d_text + (tb & ~63ull), 0, tb + 64 - (tb & ~63ull)));
		}
		std::string read_err;
		std::thread reader([&]() {
			try {
				const double r0 = now();
				if (spill) return;
				if (direct) {
					if (tb != fsize) throw std::runtime_error("internal: packed size mismatch");
					b3m::read_range_to_device(infn, 0, d_text, tb, dev0, io_threads);
				}
				else if (!(filetype == B3M_INPUT_COMPACTSTREAM && cbits != 2)) {
					HIPCHK(hipSetDevice(dev0));
					if (tb > file.size()) throw std::runtime_error("internal: packed size mismatch");
					HIPCHK(hipMemcpy(d_text, file.data(), tb, hipMemcpyHostToDevice));
				}
				t_read = now() - r0;
			} catch (const std::exception &e) { read_err = e.what(); }
		});
		std::string plan_err;
		if (!sharded && !B.plan) {
			B.plan = b3m_plan_create(n, type, o->blocksize, mem, anchor_rate);
			if (!B.plan) plan_err = b3m_last_error();
		}
		reader.join();
		if (!plan_err.empty()) throw std::runtime_error(plan_err);
		if (!read_err.empty()) throw std::runtime_error(read_err);
		const double tp1 = now();
		// ---- sort -------------------------------------------------------------------------------------------------------------------
		double t1 = now(), t2 = t1;
		// a shard that refuses its range (one 32-base window holds more suffixes than a shard's workspace: borders fall inside level-0 digits, but never inside a
		// window) must not end the job -- the reference never fails because of mem= or its thread count (/root/reference/src/bwtb3m.cpp:48-55)
		const auto refusal = [](const std::string &why) { return why.find("suffix-range shard larger than the workspace") != std::string::npos; };
		bool run_sharded = sharded;
		if (run_sharded) {
			log("[V] input read into HBM in %.3f s\n", tp1 - tp0);
			t1 = now();
			// (the whole BWT on every device for the LF walks of bwtonly=0; bwtonly=1 needs it nowhere: the slices are coded where they lie)
			try { construct_sharded(devs, type, n, mem, anchor_rate, d_text, tb, B, shards, log, want_sa ? 2 : 0); own_bwt = false; }
			catch (const std::exception &e) {
				const std::string why = e.what();
				if (!refusal(why) || n >= (1ull << 32)) throw;
				free_shards(shards);
				HIPCHK(hipSetDevice(dev0));
				if (B.d_pairs_own) (void)hipFree(B.d_pairs_own);
				B = Built{}; own_bwt = false;          // (the shards owned their copies of the BWT)
				Log{log_fd}("[W] numgpus=%d: %s -- constructing on one device instead\n", ngpus, why.c_str());
				B.plan = b3m_plan_create(n, type, 0, mem, anchor_rate);
				if (B.plan && b3m_plan_host_tier(B.plan)) { b3m_plan_destroy(B.plan); B.plan = b3m_plan_create(n, type, 0, 0, anchor_rate); }      // (the text is in HBM already)
				if (!B.plan) throw std::runtime_error(b3m_last_error());
				run_sharded = false;
			}
			t2 = now();
		}
		if (!run_sharded) {
			b3m_plan *plan = B.plan;
			if (*b3m_plan_warning(plan)) Log{log_fd}("[W] %s\n", b3m_plan_warning(plan));       // (whatever verbose= says)
			if (spill) log("[V] blocks %llu workspace %.2f GB; plan created, %.3f GB of text read into pinned host memory (%.3f s) and %.3f GB pinned for the BWT: %.3f s\n",
			               (unsigned long long)b3m_plan_num_blocks(plan), b3m_plan_workspace_bytes(plan) / 1e9, tb / 1e9, t_read, n / 1e9, tp1 - tp0);
			else
			log("[V] blocks %llu workspace %.2f GB; plan created and input read into HBM in %.3f s (the file's %.3f GB through pinned pieces: %.3f s, beside the plan)\n",
			    (unsigned long long)b3m_plan_num_blocks(plan), b3m_plan_workspace_bytes(plan) / 1e9, tp1 - tp0, tb / 1e9, t_read);
			if (b3m_plan_text_bytes(plan) != tb && !(filetype == B3M_INPUT_COMPACTSTREAM && cbits != 2)) throw std::runtime_error("internal: packed size mismatch");
			if (spill) {
				log("[V] host tier: mem=%llu is below what a merge tree needs (7.25 bytes per symbol whatever the blocks): %d sequential suffix-range shards, the text from and the "
				    "BWT to pinned host memory\n", (unsigned long long)mem, spill);
				t1 = now();
				if (b3m_plan_run_host(plan, h_text, h_bwt, nullptr, nullptr)) {
					const std::string why = b3m_last_error();
					if (!refusal(why)) throw std::runtime_error(why);
					// the host tier cannot hold this text's heaviest window in a shard: the smallest-block merge tree instead, above the target (as before the host tier existed)
					const uint64_t bmin = std::max<uint64_t>(1024, n / 1024);
					Log{log_fd}("[W] mem=%llu: %s -- constructing by block merging with %llu-symbol blocks instead, above the memory target\n", (unsigned long long)mem, why.c_str(),
					            (unsigned long long)bmin);
					b3m_plan_destroy(plan); B.plan = plan = nullptr;
					HIPCHK(hipMalloc(&d_text, tb + 64));
					HIPCHK(hipMemset((uint8_t *)d_text + (tb & ~63ull), 0, tb + 64 - (tb & ~63ull)));
					HIPCHK(hipMemcpy(d_text, h_text, tb, hipMemcpyHostToDevice));
					b3m_host_free(h_text); h_text = nullptr; b3m_host_free(h_bwt); h_bwt = nullptr;
					B.plan = plan = b3m_plan_create(n, type, bmin, 0, anchor_rate);
					if (!plan) throw std::runtime_error(b3m_last_error());
					spill = 0;
				}
				t2 = now();
				if (spill && want_sa) {
					// the LF walks that make X.sa / X.isa read the whole BWT and a rank dictionary of it: 1.7 bytes per symbol of HBM beside the plan
					const uint64_t need = b3m_plan_workspace_bytes(plan) + n + 2 * (n / 3) + 8 * ((n + sarate - 1) / sarate);
					if (need > mem) Log{log_fd}("[W] mem=%llu is exceeded by the LF walks of bwtonly=0 (the whole BWT and its rank dictionary in HBM: %.2f GB with the plan); "
					                            "bwtonly=1 stays inside it\n", (unsigned long long)mem, need / 1e9);
					HIPCHK(hipMalloc((void **)&B.d_bwt, n + 16)); own_bwt = true;
					HIPCHK(hipMemcpy(B.d_bwt, h_bwt, n, hipMemcpyHostToDevice));
				}
			}
			if (!spill) {
			HIPCHK(hipMalloc((void **)&B.d_bwt, (wide ? 4 * n : n) + 16)); own_bwt = true;
			t1 = now();
			if (b3m_plan_run(plan, d_text, B.d_bwt, nullptr)) throw std::runtime_error(b3m_last_error());
			t2 = now();
			}
			b3m_result res; b3m_plan_result(plan, &res);
			B.dev = dev0; B.d_text = b3m_plan_text_buffer(plan); B.p0 = res.p0; B.num_blocks = res.num_blocks; B.levels = res.levels;
			B.ms_sort = res.ms_sort; B.ms_gap = res.ms_gap; B.ms_interleave = res.ms_interleave;
			if (!wide) b3m_plan_histogram(plan, B.hist);
			if (res.num_samples && !want_sa) {
				// X.preisa is the sample table as it stands in HBM: streamed to the file from there
				uint64_t c = 0, empty = 0;
				B3MCHK(b3m_plan_samples_device(plan, &B.d_pairs, &c, &empty, nullptr));
				if (empty) throw std::runtime_error("internal: " + std::to_string(empty) + " (rank,pos) samples were not written by the sort");
				B.npairs = c;
			} else if (res.num_samples) { uint64_t c = 0; B3MCHK(b3m_plan_samples_raw(plan, &B.pairs, &c)); B.npairs = c; }
		}
		// (this process launches every kernel for the first time: `sort` includes the loading of their code objects, 3-14 ms at 3 Gbase; with
		// HIP_ENABLE_DEFERRED_LOADING=0 the figure is bench.py's ms_per_step and the 2.4 s of loading move in front of the plan: tools/cli_sort_time.py)
		log("[V] constructed BWT in %.3f s (sort %.1f ms incl. first-launch code loading, gap %.1f ms, interleave %.1f ms), %.3f Gsym/s\n", t2 - t1, B.ms_sort, B.ms_gap,
		    B.ms_interleave, n / (t2 - t1) / 1e9);
		// ---- out: X.bwt beside X.sa/X.isa | X.preisa -----------------------------------------------------------------------------
		const unsigned enc_threads = o->numthreads ? o->numthreads : 1;
		double t_bwt = 0, t_idx = 0;
		std::string err_bwt, err_idx;
		std::thread bwt_writer([&]() {
			try {
				const double w0 = now();
				HIPCHK(hipSetDevice(B.dev));
				hipStream_t s = nullptr;
				HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
				// X.bwt: written from the BWT in HBM (runs, frequencies and code streams on the device); the host encoder takes what that
				// one does not (2^32 symbols and more) and B3M_HOST_RL=1 asks for it (both write the same bytes)
				if (wide) {
					std::vector<uint32_t> bw(n);
					HIPCHK(hipMemcpy(bw.data(), B.d_bwt, 4 * n, hipMemcpyDeviceToHost));
					b3m::rl_encode_file_wide(outfn, bw.data(), n, 64 * 1024, wide_map.data());
					std::map<int64_t, uint64_t> hm;
					for (uint64_t i = 0; i < n; ++i) hm[wide_map[bw[i]]]++;
					b3m::write_hist_map(prefix + ".hist", hm);
				} else if (run_sharded && !b3m::sw().HOST_RL.set && encode_sharded(outfn, shards, n, 64 * 1024, enc_threads, symmap, log)) {
					HIPCHK(hipSetDevice(B.dev));
					b3m::write_hist(prefix + ".hist", B.hist, symmap);
				} else if (!B.d_bwt && h_bwt) {
					// host tier, bwtonly=1: the BWT is in pinned host memory; the device encoder takes it one segment at a time (a sixteenth of the memory target)
					const uint64_t seg = std::max<uint64_t>(1ull << 24, mem / 16);
					if (b3m::sw().HOST_RL.set || !b3m::rl_encode_device_from_host(outfn, h_bwt, n, 64 * 1024, enc_threads, symmap, s, seg))
						b3m::rl_encode_file(outfn, h_bwt, n, 64 * 1024, enc_threads, symmap);
					b3m::write_hist(prefix + ".hist", B.hist, symmap);
				} else {
					if (run_sharded && !B.d_bwt) gather_on_first(shards, n, B);       // (the slices could not be coded where they lie: a text for the host encoder)
					HIPCHK(hipSetDevice(B.dev));
					if (b3m::sw().HOST_RL.set || !b3m::rl_encode_device(outfn, B.d_bwt, n, 64 * 1024, enc_threads, symmap, s)) {
						std::vector<uint8_t> bwt(n);
						HIPCHK(hipMemcpy(bwt.data(), B.d_bwt, n, hipMemcpyDeviceToHost));
						b3m::rl_encode_file(outfn, bwt.data(), n, 64 * 1024, enc_threads, symmap);
					}
					b3m::write_hist(prefix + ".hist", B.hist, symmap);
				}
				b3m::write_meta(prefix + ".meta", {n, B.num_blocks, B.levels, B.p0, isarate});
				(void)hipStreamDestroy(s);
				t_bwt = now() - w0;
			} catch (const std::exception &e) { err_bwt = e.what(); }
		});
		try {
			const double w0 = now();
			if (want_sa) {
				// bwtonly=0 (/root/reference/README.md:63-67): sampled suffix array and its inverse by LF
				// walks over the final BWT, every symbol and every chain verified on the way
				hipStream_t s = nullptr;
				HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
				const uint64_t nsa = (n + sarate - 1) / sarate;
				uint64_t mm[2] = {0, 0};
				uint64_t head[2] = {sarate, nsa};
				b3m::ParFile pf(prefix + ".sa");
				pf.reserve(16 + 8 * nsa);
				pf.put(0, head, 16);
				if (two_bit) {
					// the SA samples stay in HBM and leave for the file in pieces (0.75 GB at 3 Gbase and rate 32)
					uint64_t *d_sa = nullptr;
					HIPCHK(hipMalloc((void **)&d_sa, 8 * nsa + 16));
					int rc = 0;
					if (run_sharded && shards.size() > 1 && B.npairs >= shards.size()) {
						try { walk_sharded(shards, n, type, B, sarate, d_sa, mm, log); } catch (...) { (void)hipFree(d_sa); throw; }
						HIPCHK(hipSetDevice(B.dev));
					}
					else rc = B.plan ? b3m_plan_lf_walk_dev(B.plan, B.d_bwt, sarate, d_sa, mm, s)
					                 : b3m_lf_walk(B.d_text, n, type, B.d_bwt, B.p0, B.hist, B.pairs, B.npairs, sarate, nullptr, d_sa, mm, s);
					if (rc) { (void)hipFree(d_sa); throw std::runtime_error(b3m_last_error()); }
					if (!(mm[0] || mm[1])) pf.put_device(16, d_sa, 8 * nsa, B.dev, io_threads);
					(void)hipFree(d_sa);
				} else {
					std::vector<uint64_t> sa(nsa);
					if (b3m_plan_lf_walk(B.plan, B.d_bwt, sarate, sa.data(), mm, s)) throw std::runtime_error(b3m_last_error());
					pf.put_host(16, sa.data(), 8 * nsa, io_threads);
				}
				pf.close();
				(void)hipStreamDestroy(s);
				if (mm[0] || mm[1]) throw std::runtime_error("internal error: the LF walk over the constructed BWT does not reproduce the text");
				std::vector<uint64_t> isa((n + isarate - 1) / isarate);
				for (uint64_t i = 0; i < B.npairs; ++i) { uint64_t pos = B.pairs[2 * i + 1]; if (pos % isarate == 0) isa[pos / isarate] = B.pairs[2 * i]; }
				b3m::write_sampled(prefix + ".isa", isarate, isa);
				log("[V] sampled SA (rate %llu) and ISA (rate %llu) by LF walks, BWT verified against the text\n", (unsigned long long)sarate, (unsigned long long)isarate);
			} else {
				b3m::ParFile pf(prefix + ".preisa");
				pf.reserve(16 * B.npairs);
				if (B.d_pairs) pf.put_device(0, B.d_pairs, 16 * B.npairs, B.dev, io_threads);
				else pf.put_host(0, B.pairs, 16 * B.npairs, io_threads);
				pf.close();
			}
			t_idx = now() - w0;
		} catch (const std::exception &e) { err_idx = e.what(); }
		bwt_writer.join();
		if (!err_bwt.empty()) throw std::runtime_error(err_bwt);
		if (!err_idx.empty()) throw std::runtime_error(err_idx);
		double t3 = now();
		log("[V] wrote %s (+.hist .meta %s) in %.3f s (RL coding + .bwt %.3f s beside %s %.3f s); total %.3f s\n", outfn.c_str(), want_sa ? ".sa .isa" : ".preisa", t3 - t2, t_bwt,
		    want_sa ? "LF walks + .sa/.isa" : ".preisa", t_idx, t3 - t0);
		cleanup();
		return 0;
	} catch (const std::exception &e) {
		b3m::set_error(e.what());
		cleanup();
		return 1;
	}
}
