// compute_bwt.cpp -- b3m_compute_bwt: the whole job of BwtMergeSort::computeBwt
// (/root/reference/src/bwtb3m.cpp:63; steps of SURVEY.md 3.1): read the input file, keep the
// packed text resident in HBM, run the plan, write X.bwt (RL), X.hist, X.meta, X.preisa.
#include "formats.h"
#include "../../../include/b3m.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
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
}

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
	void *d_text = nullptr; uint8_t *d_bwt = nullptr; b3m_plan *plan = nullptr;
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
		double t0 = now();
		std::vector<uint8_t> file = b3m::read_file(o->input_fn);
		uint64_t n;
		// compactstream (row f4): 2-bit files are a .pac payload behind a 32-byte header; other widths are decoded to bytes on the device
		uint32_t cbits = 0; uint64_t coff = 0; const int filetype = type;
		int64_t symmap_store[256]; const int64_t *symmap = nullptr;
		std::vector<int64_t> wide_map;                        // utf-8 above 256 distinct code points: rank -> code point (the text goes as symbols32)
		if (type == B3M_INPUT_COMPACTSTREAM) {
			if (b3m_compact_info(file.data(), file.size(), &cbits, &n, &coff)) throw std::runtime_error(b3m_last_error());
			file.erase(file.begin(), file.begin() + (long)coff);
			file.resize(file.size() + 16, 0);
			type = cbits == 2 ? B3M_INPUT_PAC : B3M_INPUT_BYTESTREAM;
		} else if (type == B3M_INPUT_UTF8) {
			// utf-8 (row f4): the text is the sequence of code points.  The code points that occur are ranked (order kept) into a byte
			// alphabet, the ranks are sorted and merged as a bytestream, and the files name the code points again (symmap).
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
		} else if (type == B3M_INPUT_BYTESTREAM) n = file.size();
		else {
			if (file.size() < 2 || file.back() > 3) throw std::runtime_error(std::string(o->input_fn) + " is not a .pac file");
			n = (file.size() - 2) * 4 + file.back() + (type == B3M_INPUT_PACTERM ? 1 : 0);
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
		const double tp0 = now();
		plan = b3m_plan_create(n, type, o->blocksize, mem, anchor_rate);
		if (!plan) throw std::runtime_error(b3m_last_error());
		if (*b3m_plan_warning(plan)) Log{log_fd}("[W] %s\n", b3m_plan_warning(plan));       // (whatever verbose= says)
		log("[V] blocks %llu workspace %.2f GB, plan created in %.3f s (file read %.3f s)\n", (unsigned long long)b3m_plan_num_blocks(plan), b3m_plan_workspace_bytes(plan) / 1e9,
		    now() - tp0, tp0 - t0);
		uint64_t tb = b3m_plan_text_bytes(plan);
		HIPCHK(hipMalloc((void **)&d_bwt, wide ? 4 * n : n));
		if (filetype == B3M_INPUT_COMPACTSTREAM && cbits != 2) {
			void *d_words = nullptr;
			HIPCHK(hipMalloc(&d_words, file.size()));
			HIPCHK(hipMemcpy(d_words, file.data(), file.size(), hipMemcpyHostToDevice));
			HIPCHK(hipMalloc(&d_text, n + 16));
			if (b3m_compact_decode(d_words, n, cbits, (uint8_t *)d_text, nullptr)) { (void)hipFree(d_words); throw std::runtime_error(b3m_last_error()); }
			HIPCHK(hipDeviceSynchronize());
			(void)hipFree(d_words);
		} else {
			if (tb > file.size()) throw std::runtime_error("internal: packed size mismatch");
			HIPCHK(hipMalloc(&d_text, tb + 16));
			HIPCHK(hipMemcpy(d_text, file.data(), tb, hipMemcpyHostToDevice));
		}
		double t1 = now();
		if (b3m_plan_run(plan, d_text, d_bwt, nullptr)) throw std::runtime_error(b3m_last_error());
		double t2 = now();
		b3m_result res; b3m_plan_result(plan, &res);
		// (this process launches every kernel for the first time: `sort` includes the loading of their code objects, 3-14 ms at 3 Gbase; with
		// HIP_ENABLE_DEFERRED_LOADING=0 the figure is bench.py's ms_per_step and the 2.4 s of loading move in front of the plan: tools/cli_sort_time.py)
		log("[V] constructed BWT in %.3f s (sort %.1f ms incl. first-launch code loading, gap %.1f ms, interleave %.1f ms), %.3f Gsym/s\n", t2 - t1, res.ms_sort, res.ms_gap,
		    res.ms_interleave, n / (t2 - t1) / 1e9);
		// X.bwt: written from the BWT in HBM (runs, frequencies and code streams on the device); the host encoder takes what that
		// one does not (2^32 symbols and more) and B3M_HOST_RL=1 asks for it (both write the same bytes)
		const unsigned enc_threads = o->numthreads ? o->numthreads : 1;
		if (wide) {
			std::vector<uint32_t> bw(n);
			HIPCHK(hipMemcpy(bw.data(), d_bwt, 4 * n, hipMemcpyDeviceToHost));
			b3m::rl_encode_file_wide(outfn, bw.data(), n, 64 * 1024, wide_map.data());
			std::map<int64_t, uint64_t> hm;
			for (uint64_t i = 0; i < n; ++i) hm[wide_map[bw[i]]]++;
			b3m::write_hist_map(prefix + ".hist", hm);
		} else
		if (getenv("B3M_HOST_RL") || !b3m::rl_encode_device(outfn, d_bwt, n, 64 * 1024, enc_threads, symmap, nullptr)) {
			std::vector<uint8_t> bwt(n);
			HIPCHK(hipMemcpy(bwt.data(), d_bwt, n, hipMemcpyDeviceToHost));
			b3m::rl_encode_file(outfn, bwt.data(), n, 64 * 1024, enc_threads, symmap);
		}
		if (!wide) {
			uint64_t hist[256];
			b3m_plan_histogram(plan, hist);
			b3m::write_hist(prefix + ".hist", hist, symmap);
		}
		std::vector<uint64_t> pairs(2 * res.num_samples);
		if (res.num_samples) b3m_plan_samples(plan, pairs.data(), res.num_samples);
		b3m::write_meta(prefix + ".meta", {n, res.num_blocks, res.levels, res.p0, isarate});
		if (want_sa) {
			// bwtonly=0 (/root/reference/README.md:63-67): sampled suffix array and its inverse by LF
			// walks over the final BWT, every symbol and every chain verified on the way
			std::vector<uint64_t> sa((n + sarate - 1) / sarate);
			uint64_t mm[2] = {0, 0};
			if (b3m_plan_lf_walk(plan, d_bwt, sarate, sa.data(), mm, nullptr)) throw std::runtime_error(b3m_last_error());
			if (mm[0] || mm[1]) throw std::runtime_error("internal error: the LF walk over the constructed BWT does not reproduce the text");
			b3m::write_sampled(prefix + ".sa", sarate, sa);
			std::vector<uint64_t> isa((n + isarate - 1) / isarate);
			for (uint64_t i = 0; i < res.num_samples; ++i) { uint64_t pos = pairs[2 * i + 1]; if (pos % isarate == 0) isa[pos / isarate] = pairs[2 * i]; }
			b3m::write_sampled(prefix + ".isa", isarate, isa);
			log("[V] sampled SA (rate %llu) and ISA (rate %llu) by LF walks, BWT verified against the text\n", (unsigned long long)sarate, (unsigned long long)isarate);
		} else {
			b3m::write_preisa(prefix + ".preisa", pairs.data(), res.num_samples);
		}
		double t3 = now();
		log("[V] wrote %s (+.hist .meta .preisa) in %.3f s; total %.3f s\n", outfn.c_str(), t3 - t2, t3 - t0);
		b3m_plan_destroy(plan);
		(void)hipFree(d_text); (void)hipFree(d_bwt);
		return 0;
	} catch (const std::exception &e) {
		b3m::set_error(e.what());
		if (plan) b3m_plan_destroy(plan);
		if (d_text) (void)hipFree(d_text);
		if (d_bwt) (void)hipFree(d_bwt);
		return 1;
	}
}
