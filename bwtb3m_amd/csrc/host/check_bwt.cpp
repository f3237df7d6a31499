// check_bwt.cpp -- b3m_check_bwt: the reference's verifier (/root/reference/src/checkbwt.cpp:26-243: `checkbwt [-i type] X.bwt text`) on the device.
// The reference takes (rank,pos) pairs of X.preisa as the ends of its chains, loads an LF structure of X.bwt and walks every chain backwards over
// the text, comparing the symbol the LF step names with the text's (:176-243).  Here: X.bwt is decoded (host), BWT and text go to HBM, the anchors
// are ALL pairs of X.preisa (or the ranks of X.isa where only that exists: bwtonly=0), one chain per anchor (b3m_lf_walk: every symbol and every
// chain closure checked).
#include "formats.h"
#include "../switches.h"
#include "parfile.h"
#include "../../../include/b3m.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <unistd.h>

namespace b3m { void set_error(const std::string &msg); }

namespace {
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
void logf(int fd, const char *fmt, ...)
{
	if (fd < 0) return;
	char buf[512];
	va_list ap; va_start(ap, fmt);
	int k = vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (k > 0) { ssize_t w = write(fd, buf, (size_t)std::min<int>(k, (int)sizeof buf - 1)); (void)w; }
}
}

extern "C" int b3m_check_bwt(const char *bwt_fn, const char *text_fn, const char *inputtype, int log_fd, uint64_t *mismatches_out)
{
	b3m::switches_load();
	void *d_text = nullptr; uint8_t *d_bwt = nullptr;
	auto cleanup = [&]() { if (d_text) (void)hipFree(d_text); if (d_bwt) (void)hipFree(d_bwt); b3m::staging_release(); };
	try {
		if (!bwt_fn || !text_fn || !mismatches_out) throw std::runtime_error("b3m_check_bwt: null argument");
		mismatches_out[0] = mismatches_out[1] = 0;
		const char *tname = (inputtype && *inputtype) ? inputtype : "bytestream";
		const int type = b3m_parse_input_type(tname);
		if (type < 0) throw std::runtime_error(b3m_last_error());
		if (type != B3M_INPUT_BYTESTREAM && type != B3M_INPUT_PAC && type != B3M_INPUT_PACTERM)
			throw std::runtime_error(std::string("checkbwt: inputtype=") + tname + " is not supported by this engine's verifier (supported: bytestream, pac, pacterm)");
		const std::string bwtname = bwt_fn, prefix = b3m::clip_off(bwtname, ".bwt");
		const unsigned threads = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
		// ---- text ----
		const uint64_t fsize = b3m::file_size(text_fn);
		uint64_t n;
		if (type == B3M_INPUT_BYTESTREAM) n = fsize;
		else {
			uint8_t last = 255;
			if (fsize >= 2) b3m::read_range(text_fn, fsize - 1, &last, 1, 1);
			if (fsize < 2 || last > 3) throw std::runtime_error(std::string(text_fn) + " is not a .pac file");
			n = (fsize - 2) * 4 + last + (type == B3M_INPUT_PACTERM ? 1 : 0);
		}
		const uint64_t nb = b3m::rl_length(bwtname);
		if (nb != n) throw std::runtime_error("checkbwt: " + bwtname + " holds " + std::to_string(nb) + " symbols, the text " + std::to_string(n));
		if (n == 0) throw std::runtime_error("checkbwt: empty text");
		int dev = 0;
		HIPCHK(hipGetDevice(&dev));
		HIPCHK(hipMalloc(&d_text, fsize + 64));
		HIPCHK(hipMemset((uint8_t *)d_text + (fsize & ~63ull), 0, fsize + 64 - (fsize & ~63ull)));
		b3m::read_range_to_device(text_fn, 0, d_text, fsize, dev, 8);
		// ---- X.bwt: decoded on the host (the blocks in parallel), then to HBM ----
		{
			std::vector<uint8_t> sym(n);
			bool bad = false;
			if (b3m::rl_decode_array(bwtname, sym.data(), n, threads, &bad) != n || bad) throw std::runtime_error("checkbwt: " + bwtname + " does not decode to " + std::to_string(n) + " byte symbols");
			HIPCHK(hipMalloc((void **)&d_bwt, n + 16));
			HIPCHK(hipMemcpy(d_bwt, sym.data(), n, hipMemcpyHostToDevice));
		}
		// ---- anchors: (rank,pos) pairs of X.preisa (/root/reference/src/checkbwt.cpp:131-148), else the ranks of X.isa; in position order ----
		std::vector<uint64_t> pairs;
		if (b3m::file_exists(prefix + ".preisa")) {
			std::vector<uint8_t> raw = b3m::read_file(prefix + ".preisa");
			if (raw.size() % 16) throw std::runtime_error(prefix + ".preisa: size is not a multiple of 16");
			pairs.resize(raw.size() / 8);
			memcpy(pairs.data(), raw.data(), raw.size());
		} else if (b3m::file_exists(prefix + ".isa")) {
			std::vector<uint8_t> raw = b3m::read_file(prefix + ".isa");
			if (raw.size() < 16) throw std::runtime_error(prefix + ".isa is too short");
			uint64_t rate, cnt; memcpy(&rate, raw.data(), 8); memcpy(&cnt, raw.data() + 8, 8);
			if (!rate || raw.size() != 16 + 8 * cnt) throw std::runtime_error(prefix + ".isa: bad header");
			pairs.resize(2 * cnt);
			for (uint64_t i = 0; i < cnt; ++i) { memcpy(&pairs[2 * i], raw.data() + 16 + 8 * i, 8); pairs[2 * i + 1] = i * rate; }
		} else throw std::runtime_error("checkbwt: neither " + prefix + ".preisa nor " + prefix + ".isa exists (the (rank,pos) pairs the chains start from)");
		const uint64_t K = pairs.size() / 2;
		if (!K) throw std::runtime_error("checkbwt: no (rank,pos) pairs");
		{
			bool sorted = true;
			for (uint64_t i = 1; i < K && sorted; ++i) sorted = pairs[2 * i + 1] > pairs[2 * i - 1];
			if (!sorted) {
				std::vector<std::pair<uint64_t, uint64_t>> v(K);
				for (uint64_t i = 0; i < K; ++i) v[i] = {pairs[2 * i + 1], pairs[2 * i]};
				std::sort(v.begin(), v.end());
				for (uint64_t i = 0; i < K; ++i) { pairs[2 * i] = v[i].second; pairs[2 * i + 1] = v[i].first; }
			}
			for (uint64_t i = 0; i < K; ++i) if (pairs[2 * i] >= n || pairs[2 * i + 1] >= n) throw std::runtime_error("checkbwt: a (rank,pos) pair lies outside the text");
		}
		if (pairs[1] != 0) throw std::runtime_error("checkbwt: no (rank,pos) pair for position 0");
		// ---- symbol counts: X.hist where it exists (reader /root/reference/src/bwtdecodeblock.cpp:150-173), else counted ----
		uint64_t hist[256] = {0};
		if (b3m::file_exists(prefix + ".hist")) { for (auto &kv : b3m::read_hist(prefix + ".hist")) if (kv.first >= 0 && kv.first < 256) hist[kv.first] = kv.second; }
		else if (b3m_histogram(d_text, n, type, 0, n, hist, nullptr)) throw std::runtime_error(b3m_last_error());
		logf(log_fd, "[V] checkbwt: %llu symbols, %llu chains (one per (rank,pos) pair)\n", (unsigned long long)n, (unsigned long long)K);
		uint64_t mm[2] = {0, 0};
		if (b3m_lf_walk(d_text, n, type, d_bwt, pairs[0], hist, pairs.data(), K, 0, nullptr, nullptr, mm, nullptr)) throw std::runtime_error(b3m_last_error());
		mismatches_out[0] = mm[0]; mismatches_out[1] = mm[1];
		if (mm[0] || mm[1]) logf(log_fd, "[E] failure: %llu symbols of the BWT differ from the text, %llu chains do not arrive at their pair's rank\n", (unsigned long long)mm[0], (unsigned long long)mm[1]);
		logf(log_fd, "[V] gok=%d\n", (mm[0] || mm[1]) ? 0 : 1);                     // the reference's last line (/root/reference/src/checkbwt.cpp:241)
		cleanup();
		return 0;
	} catch (const std::exception &e) {
		b3m::set_error(e.what());
		cleanup();
		return 1;
	}
}

// b3m_compute_ssa: bwtcomputessa (/root/reference/src/bwtcomputessa.cpp:24-58; the pipeline of /root/reference/README.md:63-67): X.sa and X.isa from X.bwt and the hint
// file X.preisa, without the text.  The symbols say which walk: {0 once, 1..4} = a pacterm text (2-bit rank dictionary, the 0 is the terminator entry), at most {0..3} = a
// circular 2-bit text, anything else the byte LF array (fewer than 2^32 symbols).
extern "C" int b3m_compute_ssa(const char *bwt_fn, uint64_t sasamplingrate, uint64_t isasamplingrate, int log_fd)
{
	b3m::switches_load();
	uint8_t *d_bwt = nullptr;
	auto cleanup = [&]() { if (d_bwt) (void)hipFree(d_bwt); b3m::staging_release(); };
	try {
		if (!bwt_fn) throw std::runtime_error("b3m_compute_ssa: null argument");
		const uint64_t sarate = sasamplingrate ? sasamplingrate : 32, isarate = isasamplingrate ? isasamplingrate : 32;
		const std::string bwtname = bwt_fn, prefix = b3m::clip_off(bwtname, ".bwt");
		const unsigned threads = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
		const uint64_t n = b3m::rl_length(bwtname);
		if (n == 0) throw std::runtime_error("bwtcomputessa: " + bwtname + " is empty");
		std::vector<uint8_t> sym(n);
		bool bad = false;
		if (b3m::rl_decode_array(bwtname, sym.data(), n, threads, &bad) != n || bad) throw std::runtime_error("bwtcomputessa: " + bwtname + " does not decode to " + std::to_string(n) + " byte symbols");
		// symbol counts: X.hist (reader /root/reference/src/bwtdecodeblock.cpp:150-173) or counted from the BWT (a permutation of the text)
		uint64_t hist[256] = {0};
		if (b3m::file_exists(prefix + ".hist")) { for (auto &kv : b3m::read_hist(prefix + ".hist")) if (kv.first >= 0 && kv.first < 256) hist[kv.first] = kv.second; }
		else for (uint64_t i = 0; i < n; ++i) hist[sym[i]]++;
		int maxsym = 0; for (int c = 0; c < 256; ++c) if (hist[c]) maxsym = c;
		const int type = (maxsym <= 4 && maxsym == 4 && hist[0] == 1) ? B3M_INPUT_PACTERM : (maxsym <= 3 ? B3M_INPUT_PAC : B3M_INPUT_BYTESTREAM);
		uint64_t p0 = 0;
		if (type == B3M_INPUT_PACTERM) { const void *z = memchr(sym.data(), 0, n); if (!z) throw std::runtime_error("bwtcomputessa: X.hist names a terminator the BWT does not hold"); p0 = (uint64_t)((const uint8_t *)z - sym.data()); }
		if (type == B3M_INPUT_BYTESTREAM && n >= (1ull << 32)) throw std::runtime_error("bwtcomputessa: byte alphabets of 2^32 symbols and more are not supported");
		if (!b3m::file_exists(prefix + ".preisa")) throw std::runtime_error("bwtcomputessa: " + prefix + ".preisa does not exist (the hint file bwtb3m writes with bwtonly=1)");
		std::vector<uint8_t> raw = b3m::read_file(prefix + ".preisa");
		if (raw.size() % 16 || raw.empty()) throw std::runtime_error(prefix + ".preisa: size is not a positive multiple of 16");
		std::vector<uint64_t> pairs(raw.size() / 8);
		memcpy(pairs.data(), raw.data(), raw.size());
		const uint64_t K = pairs.size() / 2;
		{
			bool sorted = true;
			for (uint64_t i = 1; i < K && sorted; ++i) sorted = pairs[2 * i + 1] > pairs[2 * i - 1];
			if (!sorted) {
				std::vector<std::pair<uint64_t, uint64_t>> v(K);
				for (uint64_t i = 0; i < K; ++i) v[i] = {pairs[2 * i + 1], pairs[2 * i]};
				std::sort(v.begin(), v.end());
				for (uint64_t i = 0; i < K; ++i) { pairs[2 * i] = v[i].second; pairs[2 * i + 1] = v[i].first; }
			}
			for (uint64_t i = 0; i < K; ++i) if (pairs[2 * i] >= n || pairs[2 * i + 1] >= n) throw std::runtime_error("bwtcomputessa: a (rank,pos) pair of " + prefix + ".preisa lies outside the text");
		}
		if (pairs[1] != 0) throw std::runtime_error("bwtcomputessa: " + prefix + ".preisa has no pair for position 0");
		HIPCHK(hipMalloc((void **)&d_bwt, n + 16));
		HIPCHK(hipMemcpy(d_bwt, sym.data(), n, hipMemcpyHostToDevice));
		std::vector<uint8_t>().swap(sym);
		std::vector<uint64_t> sa((n + sarate - 1) / sarate), isa((n + isarate - 1) / isarate);
		b3m_lf_job job;
		memset(&job, 0, sizeof job);
		job.d_text = nullptr; job.n = n; job.inputtype = type; job.d_bwt = d_bwt; job.p0 = p0; job.hist256 = hist;
		job.sample_pairs = pairs.data(); job.num_samples = K; job.sarate = sarate; job.sa_out = sa.data(); job.isarate = isarate; job.isa_out = isa.data();
		uint64_t mm[2] = {0, 0};
		logf(log_fd, "[V] bwtcomputessa: %llu symbols (%s walk), %llu chains, SA rate %llu, ISA rate %llu\n", (unsigned long long)n,
		     type == B3M_INPUT_BYTESTREAM ? "byte LF array" : (type == B3M_INPUT_PACTERM ? "2-bit dictionary, terminated" : "2-bit dictionary"), (unsigned long long)K,
		     (unsigned long long)sarate, (unsigned long long)isarate);
		if (b3m_lf_walk_job(&job, mm, nullptr)) throw std::runtime_error(b3m_last_error());
		if (mm[1]) throw std::runtime_error("bwtcomputessa: " + std::to_string(mm[1]) + " chains do not arrive at the rank of their pair: " + bwtname + " and " + prefix + ".preisa do not belong together");
		for (uint64_t v : sa) if (v == ~0ull) throw std::runtime_error("bwtcomputessa: internal: a suffix array sample was not reached");
		for (uint64_t v : isa) if (v == ~0ull) throw std::runtime_error("bwtcomputessa: internal: an inverse suffix array sample was not reached");
		b3m::write_sampled(prefix + ".sa", sarate, sa);
		b3m::write_sampled(prefix + ".isa", isarate, isa);
		logf(log_fd, "[V] wrote %s.sa and %s.isa\n", prefix.c_str(), prefix.c_str());
		cleanup();
		return 0;
	} catch (const std::exception &e) {
		b3m::set_error(e.what());
		cleanup();
		return 1;
	}
}
