// cli_bwtdecodeblock.cpp -- named consumer of the container's seek index: text[low, low+len) out of X.bwt by LF steps from the nearest
// sampled position, through X.bwt.sparserank (written on first use).  Same surface as /root/reference/src/bwtdecodeblock.cpp:
//   bwtdecodeblock <in.bwt> <in.hist> <sorted isa> <text.compact> <low> <len>          (:324-333)
// symbols print as the reference prints them (:297-300): 0 -> newline, s -> "ACGTN"[s - 1]; every block is compared with the compact
// text file (:371-386; a libmaus2 CompactArray image, layout recalled, see include/b3m.h).  Host code only.
#include "arginfo.h"
#include "formats.h"
#include "../../../include/b3m.h"
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <stdexcept>

#include "options.cpp"

namespace {
uint64_t be64(const uint8_t *p) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v = (v << 8) | p[i]; return v; }
struct SortedIsa {                                  // (pos, rank) pairs in position order, big-endian (sortPreIsa)
	std::vector<uint8_t> d; uint64_t k;
	explicit SortedIsa(const std::string &fn) : d(b3m::read_file(fn)) {
		if (d.size() % 16) throw std::runtime_error("WordPairAccessor::init(): File size " + std::to_string(d.size()) + " is not a multiple of 16");
		k = d.size() / 16;
		if (!k) throw std::runtime_error(fn + " holds no samples");
	}
	std::pair<uint64_t, uint64_t> get(uint64_t i) const { return {be64(d.data() + 16 * i), be64(d.data() + 16 * i + 8)}; }
	std::pair<uint64_t, uint64_t> next_larger(uint64_t p) const {       // first sample at a position >= p, else the first sample (the text is circular)
		uint64_t lo = 0, hi = k;
		while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (get(mid).first < p) lo = mid + 1; else hi = mid; }
		return get(lo == k ? 0 : lo);
	}
};
// symbols of a compact array image (b3m_compact_info gives the layout), decoded on the host
struct CompactText {
	std::vector<uint8_t> d; uint32_t bits = 0; uint64_t n = 0, off = 0;
	explicit CompactText(const std::string &fn) : d(b3m::read_file(fn)) {
		if (d.size() < 32 || b3m_compact_info(d.data(), d.size(), &bits, &n, &off)) throw std::runtime_error(std::string(fn) + ": " + b3m_last_error());
	}
	uint32_t get(uint64_t i) const {
		uint32_t v = 0;
		for (uint32_t b = 0; b < bits; ++b) { const uint64_t bit = i * bits + b; const uint64_t word = bit / 64; const uint8_t byte = d[off + 8 * word + (bit % 64) / 8]; v = (v << 1) | ((byte >> (7 - bit % 8)) & 1u); }
		return v;
	}
};
}

int main(int argc, char *argv[])
{
	try {
		b3m::ArgInfo arginfo(argc, argv);
		if (arginfo.help || arginfo.restargs.size() < 6)
			throw std::runtime_error(std::string("usage: ") + argv[0] + " <in.bwt> <in.hist> <sortedisa> <text.compact> <low> <len>\n");
		const std::string bwt = arginfo.restargs[0], histfn = arginfo.restargs[1], sortedisa = arginfo.restargs[2], compact = arginfo.restargs[3];
		uint64_t low = b3m::parse_numeric(arginfo.restargs[4]), len = b3m::parse_numeric(arginfo.restargs[5]);
		const uint64_t n = b3m::rl_length(bwt);
		if (low + len > n) throw std::runtime_error("low+len=" + std::to_string(low + len) + " > n=" + std::to_string(n));
		std::map<int64_t, uint64_t> hm = b3m::read_hist(histfn);
		if (hm.empty()) throw std::runtime_error("LFStep::getMaxSym: no symbols");
		const int64_t minsym = hm.begin()->first, maxsym = hm.rbegin()->first;
		if (maxsym < 0) throw std::runtime_error("maxsym < 0");
		if (!b3m::file_exists(bwt + ".sparserank")) b3m::rl_block_sym_histograms(bwt, bwt + ".sparserank", minsym, maxsym);
		b3m::RLSparseRank SR(bwt, histfn, minsym, maxsym);
		SortedIsa W(sortedisa);
		CompactText text(compact);
		static const char remap[] = "ACGTN";
		const uint64_t bs = 2048;
		while (len) {
			const uint64_t todo = std::min(len, bs);
			std::cerr << "[V] decoding [" << low << "," << low + todo << ")" << std::endl;
			// from the nearest sample at or behind the block's end, LF steps walk the text backwards
			const std::pair<uint64_t, uint64_t> PP = W.next_larger(low + todo);
			uint64_t p = PP.first % n, r = PP.second;
			const uint64_t target = (low + todo) % n;
			while (p != target) { r = SR.lf(r).second; p = p ? p - 1 : n - 1; }
			std::string block(todo, 0);
			for (uint64_t i = 0; i < todo; ++i) {
				const std::pair<int64_t, uint64_t> P = SR.lf(r);
				block[todo - i - 1] = P.first ? (P.first <= 5 ? remap[P.first - 1] : '?') : '\n';
				r = P.second;
			}
			std::cout << block;
			for (uint64_t i = 0; i < todo; ++i) {
				const uint32_t c = text.get(low + i);
				const char want = c ? (c <= 5 ? remap[c - 1] : '?') : '\n';
				if (block[i] != want) throw std::runtime_error("decoded block differs from " + compact + " at position " + std::to_string(low + i));
			}
			len -= todo; low += todo;
		}
		std::cout.flush();
		return EXIT_SUCCESS;
	} catch (std::exception const &ex) {
		std::cerr << ex.what() << std::endl;
		return EXIT_FAILURE;
	}
}
