// formats.cpp -- see formats.h.  Pure host code (no HIP).
#include "formats.h"
#include "../../../include/b3m.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <queue>
#include <stdexcept>
#include <atomic>
#include <thread>
#include <sys/stat.h>

namespace b3m {

void set_error(const std::string &msg);

std::vector<uint8_t> read_file(const std::string &fn)
{
	FILE *f = fopen(fn.c_str(), "rb");
	if (!f) throw std::runtime_error("cannot open " + fn + " for reading");
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	std::vector<uint8_t> v((size_t)sz);
	if (sz && fread(v.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); throw std::runtime_error("short read on " + fn); }
	fclose(f);
	return v;
}
void write_file(const std::string &fn, const void *data, size_t bytes)
{
	FILE *f = fopen(fn.c_str(), "wb");
	if (!f) throw std::runtime_error("cannot open " + fn + " for writing");
	if (bytes && fwrite(data, 1, bytes, f) != bytes) { fclose(f); throw std::runtime_error("short write on " + fn); }
	fclose(f);
}
bool file_exists(const std::string &fn) { struct stat st; return stat(fn.c_str(), &st) == 0; }

std::string clip_off(const std::string &s, const std::string &suffix)
{
	if (s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0) return s.substr(0, s.size() - suffix.size());
	return s;
}

static void put_be64(std::vector<uint8_t> &o, uint64_t v) { for (int i = 7; i >= 0; --i) o.push_back((uint8_t)(v >> (8 * i))); }
static uint64_t get_be64(const uint8_t *p) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v = (v << 8) | p[i]; return v; }

// ---- bit I/O, MSB first -------------------------------------------------------------------------
struct BitWriter {
	std::vector<uint8_t> &out; uint64_t acc = 0; int used = 0;      // `used` pending bits in the low end of acc
	explicit BitWriter(std::vector<uint8_t> &o) : out(o) {}
	void put(uint64_t v, int bits)
	{
		while (bits > 0) {
			int take = std::min(bits, 56 - used);                    // keep room for byte-wise flushing
			acc = (acc << take) | ((v >> (bits - take)) & ((take == 64 ? 0 : (1ull << take)) - 1));
			used += take; bits -= take;
			while (used >= 8) { out.push_back((uint8_t)(acc >> (used - 8))); used -= 8; }
		}
	}
	void align() { if (used) { out.push_back((uint8_t)(acc << (8 - used))); used = 0; } acc = 0; }
	// Elias-delta of x+1 (x >= 0)
	void elias(uint64_t x) { uint64_t y = x + 1; int N = 63 - __builtin_clzll(y); int L = 31 - __builtin_clz((unsigned)(N + 1)); put(0, L); put((uint64_t)(N + 1), L + 1); put(y & ((N ? (1ull << N) : 1ull) - 1), N); }
};
struct BitReader {
	const uint8_t *p, *e; uint64_t buf = 0; int nb = 0;             // nb valid bits in the low end of buf
	BitReader(const uint8_t *b, const uint8_t *end) : p(b), e(end) {}
	void fill(int need) { while (nb < need && p < e) { buf = (buf << 8) | *p++; nb += 8; } }
	// the next `bits` (<= 32) bits without consuming them; zero-padded past the end of the data
	uint32_t peek(int bits) { fill(bits); return nb >= bits ? (uint32_t)((buf >> (nb - bits)) & ((1ull << bits) - 1)) : (uint32_t)((buf << (bits - nb)) & ((1ull << bits) - 1)); }
	void skip(int bits) { fill(bits); if (nb < bits) throw std::runtime_error("RL decoder: unexpected end of data"); nb -= bits; }
	int bit() { fill(1); if (!nb) throw std::runtime_error("RL decoder: unexpected end of data"); return (int)((buf >> --nb) & 1); }
	uint64_t get(int bits) { uint64_t v = 0; while (bits > 0) { int t = std::min(bits, 32); v = (v << t) | peek(t); skip(t); bits -= t; } return v; }
	void align() { nb -= nb & 7; }
	uint64_t elias() { int L = 0; while (!bit()) ++L; uint64_t N1 = (1ull << L) | get(L); int N = (int)N1 - 1; uint64_t y = (1ull << N) | get(N); return y - 1; }
};

// ---- canonical Huffman ----------------------------------------------------------------------------
struct Canon {
	std::vector<std::pair<int64_t, int>> syms;         // (symbol, length) in canonical order (length, symbol)
	std::map<int64_t, std::pair<uint64_t, int>> enc;  // symbol -> (code, length)
	// decoding tables per length
	std::vector<uint64_t> first_code; std::vector<int> first_idx, count;
	static const int FAST = 11;                        // codes up to this length decode with one table lookup
	std::vector<uint32_t> fast;                        // index by the next FAST bits: (symbol index << 8) | length, 0 = longer code
	void finish()
	{
		std::sort(syms.begin(), syms.end(), [](const std::pair<int64_t, int> &a, const std::pair<int64_t, int> &b) { return a.second != b.second ? a.second < b.second : a.first < b.first; });
		int maxl = 0; for (auto &s : syms) maxl = std::max(maxl, s.second);
		first_code.assign(maxl + 2, 0); first_idx.assign(maxl + 2, 0); count.assign(maxl + 2, 0);
		uint64_t code = 0; int prev = syms.empty() ? 0 : syms[0].second;
		for (size_t i = 0; i < syms.size(); ++i) {
			int l = syms[i].second;
			code <<= (l - prev); prev = l;
			if (!count[l]) { first_code[l] = code; first_idx[l] = (int)i; }
			count[l]++;
			enc[syms[i].first] = {code, l};
			code++;
		}
		fast.assign((size_t)1 << FAST, 0);
		for (size_t i = 0; i < syms.size(); ++i) {
			int l = syms[i].second; if (l > FAST) continue;
			uint64_t c0 = enc[syms[i].first].first << (FAST - l);
			for (uint64_t k = 0; k < (1ull << (FAST - l)); ++k) fast[(size_t)(c0 + k)] = ((uint32_t)i << 8) | (uint32_t)l;
		}
	}
	static Canon from_freq(const std::map<int64_t, uint64_t> &freq)
	{
		Canon c;
		if (freq.size() == 1) { c.syms.push_back({freq.begin()->first, 1}); c.finish(); return c; }
		struct Node { uint64_t w; int l, r; };
		std::vector<Node> nodes; std::vector<int64_t> leafsym;
		typedef std::pair<uint64_t, int> QE;
		std::priority_queue<QE, std::vector<QE>, std::greater<QE>> q;
		for (auto &kv : freq) { nodes.push_back({kv.second, -1, -1}); leafsym.push_back(kv.first); q.push({kv.second, (int)nodes.size() - 1}); }
		while (q.size() > 1) {
			QE a = q.top(); q.pop(); QE b = q.top(); q.pop();
			nodes.push_back({a.first + b.first, a.second, b.second});
			q.push({a.first + b.first, (int)nodes.size() - 1});
		}
		std::vector<std::pair<int, int>> st; st.push_back({(int)nodes.size() - 1, 0});
		while (!st.empty()) {
			auto cur = st.back(); st.pop_back();
			const Node &nd = nodes[cur.first];
			if (nd.l < 0) c.syms.push_back({leafsym[cur.first], std::max(cur.second, 1)});
			else { st.push_back({nd.l, cur.second + 1}); st.push_back({nd.r, cur.second + 1}); }
		}
		c.finish();
		return c;
	}
	void serialise(BitWriter &w) const { w.elias(syms.size()); for (auto &s : syms) { w.elias((uint64_t)s.first); w.elias((uint64_t)s.second); } }
	static Canon deserialise(BitReader &r)
	{
		Canon c; uint64_t k = r.elias();
		for (uint64_t i = 0; i < k; ++i) { int64_t s = (int64_t)r.elias(); int l = (int)r.elias(); c.syms.push_back({s, l}); }
		c.finish();
		return c;
	}
	void put(BitWriter &w, int64_t sym) const { auto it = enc.find(sym); w.put(it->second.first, it->second.second); }
	int64_t get(BitReader &r) const
	{
		uint32_t f = fast[r.peek(FAST)];
		if (f) { r.skip((int)(f & 255u)); return syms[f >> 8].first; }
		uint64_t code = 0;
		for (int l = 1; l < (int)count.size(); ++l) {
			code = (code << 1) | (uint64_t)r.bit();
			if (count[l] && code >= first_code[l] && code < first_code[l] + (uint64_t)count[l]) return syms[first_idx[l] + (int)(code - first_code[l])].first;
		}
		throw std::runtime_error("RL decoder: bad Huffman code");
	}
};

static const uint64_t RL_ESCAPE = 256;   // run lengths >= 256 are coded as escape + elias(len)

// header of one block (run count, escape flag, the two code tables, byte-aligned) and its code tables, from the block's frequencies:
// sf[s] = runs of byte symbol s, cf[l] = runs of length l (1..255), cf[256] = runs of 256 and more (escape + Elias-coded length).
// Shared by the host encoder below and the device encoder (k_rlenc.hip), which packs the code streams on the GPU.
void rl_block_header(const uint32_t *sf, const uint32_t *cf, uint64_t nr, const int64_t *symmap, std::vector<uint8_t> &header, RLBlockTables &t)
{
	std::map<int64_t, uint64_t> sh, ch;
	for (int s = 0; s < 256; ++s) if (sf[s]) sh[s] = sf[s];
	for (uint64_t l = 0; l <= RL_ESCAPE; ++l) if (cf[l]) ch[(int64_t)l] = cf[l];
	Canon sc = Canon::from_freq(sh), cc = Canon::from_freq(ch);
	memset(&t, 0, sizeof t);
	for (auto &kv : sc.enc) { t.scode[kv.first] = kv.second.first; t.slen[kv.first] = (uint8_t)kv.second.second; }
	for (auto &kv : cc.enc) { t.ccode[kv.first] = kv.second.first; t.clen[kv.first] = (uint8_t)kv.second.second; }
	header.clear();
	BitWriter w(header);
	w.elias(nr);
	w.put(cf[RL_ESCAPE] ? 1 : 0, 1);
	if (symmap) {          // the table names the mapped symbols (the map is monotone: the canonical order is the same)
		w.elias(sc.syms.size());
		for (auto &s : sc.syms) { w.elias((uint64_t)symmap[s.first]); w.elias((uint64_t)s.second); }
	} else sc.serialise(w);
	cc.serialise(w);
	w.align();
}

static void encode_block(const RLRun *runs, uint64_t nr, std::vector<uint8_t> &out, const int64_t *symmap)
{
	// symbols are bytes and coded lengths are 1..256 (256 = escape): flat counters and code tables
	uint32_t sf[256] = {0}, cf[RL_ESCAPE + 1] = {0};
	for (uint64_t i = 0; i < nr; ++i) {
		sf[(uint8_t)runs[i].sym]++;
		if (runs[i].len >= RL_ESCAPE) cf[RL_ESCAPE]++; else cf[runs[i].len]++;
	}
	RLBlockTables t;
	rl_block_header(sf, cf, nr, symmap, out, t);
	BitWriter w(out);
	out.reserve(out.size() + nr / 2 + 4096);
	for (uint64_t i = 0; i < nr; ++i) { uint8_t s = (uint8_t)runs[i].sym; w.put(t.scode[s], t.slen[s]); }
	w.align();
	for (uint64_t i = 0; i < nr; ++i) {
		if (runs[i].len >= RL_ESCAPE) { w.put(t.ccode[RL_ESCAPE], t.clen[RL_ESCAPE]); w.elias(runs[i].len); } else w.put(t.ccode[runs[i].len], t.clen[runs[i].len]);
	}
	w.align();
}

void rl_container_ends(uint64_t n, uint64_t blocks_bytes, const std::vector<RLIndexEntry> &index, std::vector<uint8_t> &head, std::vector<uint8_t> &tail)
{
	head.clear(); tail.clear();
	put_be64(head, n);
	for (auto &e : index) { put_be64(tail, e.pos); put_be64(tail, e.runs); put_be64(tail, e.syms); }
	put_be64(tail, index.size());
	put_be64(tail, 8 + blocks_bytes);
}
void rl_write_container(const std::string &fn, uint64_t n, const uint8_t *blocks, uint64_t blocks_bytes, const std::vector<RLIndexEntry> &index)
{
	FILE *f = fopen(fn.c_str(), "wb");
	if (!f) throw std::runtime_error("cannot open " + fn + " for writing");
	std::vector<uint8_t> head, tail;
	rl_container_ends(n, blocks_bytes, index, head, tail);
	fwrite(head.data(), 1, head.size(), f);
	if (blocks_bytes && fwrite(blocks, 1, blocks_bytes, f) != blocks_bytes) { fclose(f); throw std::runtime_error("short write to " + fn); }
	fwrite(tail.data(), 1, tail.size(), f);
	if (fclose(f)) throw std::runtime_error("error closing " + fn);
}

void rl_encode_file(const std::string &fn, const uint8_t *symbols, uint64_t n, uint64_t runs_per_block, unsigned numthreads, const int64_t *symmap)
{
	if (!runs_per_block) runs_per_block = 64 * 1024;
	if (!numthreads) numthreads = 1;
	FILE *f = fopen(fn.c_str(), "wb");
	if (!f) throw std::runtime_error("cannot open " + fn + " for writing");
	std::vector<uint8_t> head; put_be64(head, n);
	fwrite(head.data(), 1, head.size(), f);
	uint64_t filepos = 8;
	std::vector<RLIndexEntry> index;
	// Blocks hold exactly runs_per_block runs, so a block's first symbol depends on every run before it.
	// Pass 1 (parallel): run starts per 64 KiB segment -> prefix sums.  Pass 2 (parallel, in batches
	// written in order): every worker finds its block's first run by a search in the prefix sums plus a
	// scan inside one segment, collects the block's runs itself and encodes them.
	const uint64_t SEG = 64 * 1024;
	const uint64_t nseg = (n + SEG - 1) / SEG;
	std::vector<uint64_t> segruns(nseg + 1, 0);             // segruns[s] = runs starting before segment s
	{
		std::atomic<uint64_t> next(0);
		auto count = [&]() {
			for (;;) {
				uint64_t s = next.fetch_add(64);
				if (s >= nseg) return;
				for (uint64_t q = s; q < std::min(nseg, s + 64); ++q) {
					uint64_t lo = q * SEG, hi = std::min(n, lo + SEG), c = 0;
					for (uint64_t i = lo; i < hi; ++i) c += (i == 0 || symbols[i] != symbols[i - 1]);
					segruns[q + 1] = c;
				}
			}
		};
		std::vector<std::thread> th;
		for (unsigned t = 1; t < numthreads; ++t) th.emplace_back(count);
		count();
		for (auto &x : th) x.join();
		for (uint64_t s = 0; s < nseg; ++s) segruns[s + 1] += segruns[s];
	}
	const uint64_t total_runs = segruns[nseg];
	const uint64_t nblocks = (total_runs + runs_per_block - 1) / runs_per_block;
	// position of the first symbol of run r (r < total_runs)
	auto run_start = [&](uint64_t r) -> uint64_t {
		uint64_t s = (uint64_t)(std::upper_bound(segruns.begin(), segruns.end(), r) - segruns.begin()) - 1;   // segruns[s] <= r < segruns[s+1]
		uint64_t k = segruns[s];
		for (uint64_t i = s * SEG;; ++i) if (i == 0 || symbols[i] != symbols[i - 1]) { if (k == r) return i; ++k; }
	};
	const uint64_t batch = (uint64_t)numthreads * 16;
	for (uint64_t b0 = 0; b0 < nblocks; b0 += batch) {
		const uint64_t nb = std::min(batch, nblocks - b0);
		std::vector<std::vector<uint8_t>> bufs(nb);
		std::vector<uint64_t> bsyms(nb, 0), bruns(nb, 0);
		std::atomic<uint64_t> next_block(0);
		auto work = [&]() {
			std::vector<RLRun> runs; runs.reserve((size_t)runs_per_block);
			for (;;) {
				uint64_t b = next_block.fetch_add(1);
				if (b >= nb) return;
				const uint64_t r0 = (b0 + b) * runs_per_block, r1 = std::min(total_runs, r0 + runs_per_block);
				runs.clear();
				uint64_t i = run_start(r0), syms = 0;
				while (runs.size() < r1 - r0) {
					const uint8_t s = symbols[i]; uint64_t j = i + 1;
					while (j < n && symbols[j] == s) ++j;
					runs.push_back({(int64_t)s, j - i}); syms += j - i;
					i = j;
				}
				bsyms[b] = syms; bruns[b] = runs.size();
				encode_block(runs.data(), runs.size(), bufs[b], symmap);
			}
		};
		std::vector<std::thread> th;
		unsigned T = (unsigned)std::min<uint64_t>(numthreads, nb);
		for (unsigned t = 1; t < T; ++t) th.emplace_back(work);
		work();
		for (auto &x : th) x.join();
		for (uint64_t b = 0; b < nb; ++b) {
			index.push_back({filepos, bruns[b], bsyms[b]});
			fwrite(bufs[b].data(), 1, bufs[b].size(), f);
			filepos += bufs[b].size();
		}
	}
	std::vector<uint8_t> tail;
	for (auto &e : index) { put_be64(tail, e.pos); put_be64(tail, e.runs); put_be64(tail, e.syms); }
	put_be64(tail, index.size());
	put_be64(tail, filepos);
	fwrite(tail.data(), 1, tail.size(), f);
	if (fclose(f)) throw std::runtime_error("error closing " + fn);
}

// the same container for symbols of any width (utf-8 texts of more than 256 distinct code points): ranks[i] names the symbol symmap[ranks[i]];
// the code tables are built per block from maps instead of the flat byte counters (one thread: such texts are small)
void rl_encode_file_wide(const std::string &fn, const uint32_t *ranks, uint64_t n, uint64_t runs_per_block, const int64_t *symmap)
{
	if (!runs_per_block) runs_per_block = 64 * 1024;
	std::vector<uint8_t> blocks; std::vector<RLIndexEntry> index;
	std::vector<RLRun> runs; runs.reserve((size_t)runs_per_block);
	auto flush = [&]() {
		if (runs.empty()) return;
		std::map<int64_t, uint64_t> sh, ch; uint64_t syms = 0;
		for (auto &r : runs) { sh[r.sym]++; ch[(int64_t)std::min<uint64_t>(r.len, RL_ESCAPE)]++; syms += r.len; }
		Canon sc = Canon::from_freq(sh), cc = Canon::from_freq(ch);
		std::vector<uint8_t> out;
		BitWriter w(out);
		w.elias(runs.size());
		w.put(ch.count((int64_t)RL_ESCAPE) ? 1 : 0, 1);
		sc.serialise(w); cc.serialise(w);
		w.align();
		for (auto &r : runs) sc.put(w, r.sym);
		w.align();
		for (auto &r : runs) { if (r.len >= RL_ESCAPE) { cc.put(w, (int64_t)RL_ESCAPE); w.elias(r.len); } else cc.put(w, (int64_t)r.len); }
		w.align();
		index.push_back({8 + blocks.size(), runs.size(), syms});
		blocks.insert(blocks.end(), out.begin(), out.end());
		runs.clear();
	};
	for (uint64_t i = 0; i < n;) {
		uint64_t j = i + 1;
		while (j < n && ranks[j] == ranks[i]) ++j;
		runs.push_back({symmap ? symmap[ranks[i]] : (int64_t)ranks[i], j - i});
		if (runs.size() == runs_per_block) flush();
		i = j;
	}
	flush();
	rl_write_container(fn, n, blocks.data(), blocks.size(), index);
}

static std::vector<RLIndexEntry> rl_index_of(const std::vector<uint8_t> &d, const std::string &fn)
{
	if (d.size() < 24) throw std::runtime_error(fn + " is not an RL .bwt file (too short)");
	uint64_t ipos = get_be64(d.data() + d.size() - 8);
	if (ipos + 16 > d.size()) throw std::runtime_error(fn + ": bad index offset");
	uint64_t nb = get_be64(d.data() + d.size() - 16);
	if (ipos + nb * 24 + 16 != d.size()) throw std::runtime_error(fn + ": bad index size");
	std::vector<RLIndexEntry> v(nb);
	for (uint64_t b = 0; b < nb; ++b) { const uint8_t *p = d.data() + ipos + 24 * b; v[b] = {get_be64(p), get_be64(p + 8), get_be64(p + 16)}; }
	return v;
}
std::vector<RLIndexEntry> rl_index(const std::string &fn) { return rl_index_of(read_file(fn), fn); }

uint64_t rl_length(const std::string &fn)
{
	FILE *f = fopen(fn.c_str(), "rb");
	if (!f) throw std::runtime_error("cannot open " + fn + " for reading");
	uint8_t b[8];
	if (fread(b, 1, 8, f) != 8) { fclose(f); throw std::runtime_error(fn + " is not an RL .bwt file (too short)"); }
	fclose(f);
	return get_be64(b);
}

// runs of ONE block of the container (the block that starts at file offset `pos`): what a seek through the trailing index decodes
static void rl_decode_block_at(const std::vector<uint8_t> &d, uint64_t pos, uint64_t ipos, std::vector<RLRun> &runs)
{
	if (pos < 8 || pos >= ipos) throw std::runtime_error("RL decoder: block offset outside the file");
	BitReader r(d.data() + pos, d.data() + ipos);
	const uint64_t nr = r.elias();
	const bool esc = r.bit();
	Canon sc = Canon::deserialise(r), cc = Canon::deserialise(r);
	r.align();
	runs.resize(nr);
	for (uint64_t i = 0; i < nr; ++i) runs[i].sym = sc.get(r);
	r.align();
	for (uint64_t i = 0; i < nr; ++i) {
		const int64_t c = cc.get(r);
		runs[i].len = (esc && c == (int64_t)RL_ESCAPE) ? r.elias() : (uint64_t)c;
	}
}

// ---- X.bwt.sparserank + LF steps on the container (role of RLDecoder::getBlockSymHistograms and of the SparseRank consumer,
// /root/reference/src/bwtdecodeblock.cpp:134-365): per block of the container the counts of the symbols minsym..maxsym BEFORE the
// block, (maxsym - minsym + 1) 8-byte big-endian numbers per block (reader: :228-236) ----
void rl_block_sym_histograms(const std::string &bwtfn, const std::string &outfn, int64_t minsym, int64_t maxsym)
{
	if (maxsym < minsym) throw std::runtime_error("sparserank: empty symbol range");
	std::vector<uint8_t> d = read_file(bwtfn);
	std::vector<RLIndexEntry> idx = rl_index_of(d, bwtfn);
	const uint64_t ipos = get_be64(d.data() + d.size() - 8);
	const uint64_t k = (uint64_t)(maxsym - minsym + 1);
	std::vector<uint64_t> H(k, 0);
	std::vector<uint8_t> o; o.reserve(idx.size() * k * 8);
	std::vector<RLRun> runs;
	for (auto &e : idx) {
		for (uint64_t c = 0; c < k; ++c) put_be64(o, H[c]);
		rl_decode_block_at(d, e.pos, ipos, runs);
		for (auto &r : runs) {
			if (r.sym < minsym || r.sym > maxsym) throw std::runtime_error("sparserank: symbol outside [minsym,maxsym]");
			H[(uint64_t)(r.sym - minsym)] += r.len;
		}
	}
	write_file(outfn, o.data(), o.size());
}

RLSparseRank::RLSparseRank(const std::string &bwtfn, const std::string &histfn, int64_t mins, int64_t maxs)
	: data(read_file(bwtfn)), minsym(mins), maxsym(maxs)
{
	index = rl_index_of(data, bwtfn);
	ipos = get_be64(data.data() + data.size() - 8);
	n = get_be64(data.data());
	sparse = read_file(bwtfn + ".sparserank");
	const uint64_t k = (uint64_t)(maxsym - minsym + 1);
	if (sparse.size() != index.size() * k * 8) throw std::runtime_error(bwtfn + ".sparserank does not fit the index of " + bwtfn);
	vstart.resize(index.size() + 1, 0);
	for (size_t b = 0; b < index.size(); ++b) vstart[b + 1] = vstart[b] + index[b].syms;
	std::map<int64_t, uint64_t> hm = read_hist(histfn);
	if (!hm.empty() && hm.begin()->first >= 0) {           // D[sym] = symbols smaller than sym (bwtdecodeblock.cpp:150-173)
		D.assign((size_t)hm.rbegin()->first + 1, 0);
		uint64_t acc = 0;
		for (auto &kv : hm) { D[(size_t)kv.first] = acc; acc += kv.second; }
	}
	cached = ~0ull;
}
// symbol at rank i and the number of its occurrences before i
std::pair<int64_t, uint64_t> RLSparseRank::inverse_select(uint64_t i)
{
	if (i >= n) throw std::runtime_error("sparserank: rank out of range");
	const size_t b = (size_t)(std::upper_bound(vstart.begin(), vstart.end(), i) - vstart.begin()) - 1;      // findVBlock
	if (cached != b) { rl_decode_block_at(data, index[b].pos, ipos, runs); cached = b; }
	const uint64_t k = (uint64_t)(maxsym - minsym + 1);
	std::vector<uint64_t> H(k);
	for (uint64_t c = 0; c < k; ++c) H[c] = get_be64(sparse.data() + (b * k + c) * 8);
	uint64_t offset = i - vstart[b];
	for (auto &r : runs) {
		if (offset >= r.len) { H[(uint64_t)(r.sym - minsym)] += r.len; offset -= r.len; }
		else { H[(uint64_t)(r.sym - minsym)] += offset; return {r.sym, H[(uint64_t)(r.sym - minsym)]}; }
	}
	throw std::runtime_error("sparserank: block shorter than its index entry says");
}
std::pair<int64_t, uint64_t> RLSparseRank::lf(uint64_t i)
{
	std::pair<int64_t, uint64_t> p = inverse_select(i);
	if (p.first < 0 || (size_t)p.first >= D.size()) throw std::runtime_error("sparserank: symbol missing in the histogram");
	return {p.first, D[(size_t)p.first] + p.second};
}

uint64_t rl_decode_file(const std::string &fn, void (*sink)(void *, int64_t, uint64_t), void *ctx)
{
	std::vector<uint8_t> d = read_file(fn);
	if (d.size() < 24) throw std::runtime_error(fn + " is not an RL .bwt file (too short)");
	uint64_t n = get_be64(d.data());
	uint64_t ipos = get_be64(d.data() + d.size() - 8);
	if (ipos + 16 > d.size() || ipos < 8) throw std::runtime_error(fn + ": bad index offset");
	BitReader r(d.data() + 8, d.data() + ipos);
	uint64_t got = 0;
	std::vector<int64_t> syms;
	while (got < n) {
		r.align();
		uint64_t nr = r.elias();
		bool esc = r.bit();
		Canon sc = Canon::deserialise(r), cc = Canon::deserialise(r);
		r.align();
		syms.resize(nr);
		for (uint64_t i = 0; i < nr; ++i) syms[i] = sc.get(r);
		r.align();
		for (uint64_t i = 0; i < nr; ++i) {
			int64_t c = cc.get(r);
			uint64_t len = (esc && c == (int64_t)RL_ESCAPE) ? r.elias() : (uint64_t)c;
			sink(ctx, syms[i], len);
			got += len;
		}
		r.align();
		if (nr == 0) throw std::runtime_error(fn + ": empty block");
	}
	if (got != n) throw std::runtime_error(fn + ": symbol count mismatch");
	return n;
}

// decode straight into an array, the blocks in parallel (block positions and symbol counts come from the index)
uint64_t rl_decode_array(const std::string &fn, uint8_t *out, uint64_t cap, unsigned numthreads, bool *bad_symbol)
{
	std::vector<uint8_t> d = read_file(fn);
	if (d.size() < 24) throw std::runtime_error(fn + " is not an RL .bwt file (too short)");
	const uint64_t n = get_be64(d.data());
	std::vector<RLIndexEntry> idx = rl_index_of(d, fn);
	std::vector<uint64_t> off(idx.size() + 1, 0);
	for (size_t b = 0; b < idx.size(); ++b) off[b + 1] = off[b] + idx[b].syms;
	if (off.back() != n) throw std::runtime_error(fn + ": symbol count mismatch");
	const uint64_t ipos = get_be64(d.data() + d.size() - 8);
	std::atomic<uint64_t> next(0); std::atomic<int> bad(0), fail(0);
	auto work = [&]() {
		try {
			std::vector<int64_t> syms;
			for (;;) {
				uint64_t b = next.fetch_add(1);
				if (b >= idx.size()) return;
				if (idx[b].pos < 8 || idx[b].pos >= ipos) throw std::runtime_error("bad block position");
				BitReader r(d.data() + idx[b].pos, d.data() + ipos);
				uint64_t nr = r.elias(); bool esc = r.bit();
				if (nr != idx[b].runs) throw std::runtime_error("run count mismatch");
				Canon sc = Canon::deserialise(r), cc = Canon::deserialise(r);
				r.align();
				syms.resize(nr);
				for (uint64_t i = 0; i < nr; ++i) syms[i] = sc.get(r);
				r.align();
				uint64_t pos = off[b];
				for (uint64_t i = 0; i < nr; ++i) {
					int64_t c = cc.get(r);
					uint64_t len = (esc && c == (int64_t)RL_ESCAPE) ? r.elias() : (uint64_t)c;
					if (syms[i] < 0 || syms[i] > 255) bad = 1;
					if (pos + len > off[b + 1]) throw std::runtime_error("block overruns its symbol count");
					if (out && pos < cap) memset(out + pos, (int)syms[i], (size_t)std::min<uint64_t>(len, cap - pos));
					pos += len;
				}
				if (pos != off[b + 1]) throw std::runtime_error("block symbol count mismatch");
			}
		} catch (const std::exception &) { fail = 1; }
	};
	std::vector<std::thread> th;
	unsigned T = std::max(1u, std::min<unsigned>(numthreads, (unsigned)idx.size()));
	for (unsigned t = 1; t < T; ++t) th.emplace_back(work);
	work();
	for (auto &x : th) x.join();
	if (fail) throw std::runtime_error(fn + ": corrupt RL block");
	if (bad_symbol) *bad_symbol = bad != 0;
	return n;
}

// ---- siblings -----------------------------------------------------------------------------------------
void write_hist(const std::string &fn, const uint64_t hist256[256], const int64_t *symmap)
{
	std::vector<uint8_t> o; uint64_t cnt = 0;
	for (int i = 0; i < 256; ++i) cnt += hist256[i] != 0;
	put_be64(o, cnt);
	for (int i = 0; i < 256; ++i) if (hist256[i]) { put_be64(o, (uint64_t)(symmap ? symmap[i] : i)); put_be64(o, hist256[i]); }
	write_file(fn, o.data(), o.size());
}
void write_hist_map(const std::string &fn, const std::map<int64_t, uint64_t> &hist)
{
	std::vector<uint8_t> o;
	put_be64(o, hist.size());
	for (auto &kv : hist) { put_be64(o, (uint64_t)kv.first); put_be64(o, kv.second); }
	write_file(fn, o.data(), o.size());
}
std::vector<uint32_t> utf8_decode(const std::vector<uint8_t> &f)
{
	std::vector<uint32_t> out; out.reserve(f.size());
	const size_t n = f.size();
	auto bad = [&](size_t i) { throw std::runtime_error("utf-8 input: invalid byte sequence at offset " + std::to_string(i)); };
	for (size_t i = 0; i < n;) {
		const uint8_t c = f[i];
		uint32_t cp; int k;
		if (c < 0x80) { cp = c; k = 0; }
		else if (c >= 0xC2 && c <= 0xDF) { cp = c & 0x1Fu; k = 1; }
		else if (c >= 0xE0 && c <= 0xEF) { cp = c & 0x0Fu; k = 2; }
		else if (c >= 0xF0 && c <= 0xF4) { cp = c & 0x07u; k = 3; }
		else { bad(i); return out; }
		if (i + k >= n + (k ? 0 : 1)) bad(i);
		for (int j = 1; j <= k; ++j) { const uint8_t d = f[i + j]; if ((d & 0xC0u) != 0x80u) bad(i); cp = (cp << 6) | (d & 0x3Fu); }
		if ((k == 2 && cp < 0x800u) || (k == 3 && cp < 0x10000u) || cp > 0x10FFFFu || (cp >= 0xD800u && cp <= 0xDFFFu)) bad(i);
		out.push_back(cp);
		i += (size_t)k + 1;
	}
	return out;
}
std::map<int64_t, uint64_t> read_hist(const std::string &fn)
{
	std::vector<uint8_t> d = read_file(fn);
	std::map<int64_t, uint64_t> m;
	if (d.size() < 8) throw std::runtime_error(fn + ": short .hist");
	uint64_t cnt = get_be64(d.data());
	if (d.size() != 8 + 16 * cnt) throw std::runtime_error(fn + ": bad .hist size");
	for (uint64_t i = 0; i < cnt; ++i) m[(int64_t)get_be64(d.data() + 8 + 16 * i)] = get_be64(d.data() + 16 + 16 * i);
	return m;
}
void write_preisa(const std::string &fn, const uint64_t *pairs, uint64_t count) { write_file(fn, pairs, 16 * count); }
void write_sampled(const std::string &fn, uint64_t rate, const std::vector<uint64_t> &values)
{
	std::vector<uint64_t> o; o.push_back(rate); o.push_back(values.size()); o.insert(o.end(), values.begin(), values.end());
	write_file(fn, o.data(), 8 * o.size());
}
void write_meta(const std::string &fn, const std::vector<uint64_t> &numbers)
{
	std::vector<uint8_t> o; for (uint64_t v : numbers) put_be64(o, v);
	write_file(fn, o.data(), o.size());
}

}  // namespace b3m

// ---- C ABI ---------------------------------------------------------------------------------------------
extern "C" {
int b3m_rl_write(const char *fn, const uint8_t *symbols, uint64_t n, uint64_t runs_per_block)
{
	try { b3m::rl_encode_file(fn, symbols, n, runs_per_block, std::max(1u, std::thread::hardware_concurrency())); return 0; }
	catch (const std::exception &e) { b3m::set_error(e.what()); return 1; }
}
int b3m_rl_read(const char *fn, uint8_t *symbols_out, uint64_t cap, uint64_t *n_out)
{
	try {
		bool bad = false;
		uint64_t n = b3m::rl_decode_array(fn, symbols_out, cap, std::max(1u, std::thread::hardware_concurrency()), &bad);
		if (n_out) *n_out = n;
		if (bad) { b3m::set_error("symbol outside 0..255"); return 1; }
		return 0;
	} catch (const std::exception &e) { b3m::set_error(e.what()); return 1; }
}
int b3m_rl_read_symbols(const char *fn, int64_t *symbols_out, uint64_t cap, uint64_t *n_out)
{
	try {
		struct Ctx { int64_t *out; uint64_t cap, at; } c{symbols_out, cap, 0};
		uint64_t n = b3m::rl_decode_file(fn, [](void *q, int64_t sym, uint64_t len) {
			Ctx *k = (Ctx *)q;
			for (uint64_t i = 0; i < len; ++i, ++k->at) if (k->out && k->at < k->cap) k->out[k->at] = sym;
		}, &c);
		if (n_out) *n_out = n;
		return 0;
	} catch (const std::exception &e) { b3m::set_error(e.what()); return 1; }
}
int b3m_rl_length(const char *fn, uint64_t *n_out)
{
	try { uint64_t n = b3m::rl_length(fn); if (n_out) *n_out = n; return 0; }
	catch (const std::exception &e) { b3m::set_error(e.what()); return 1; }
}
}
