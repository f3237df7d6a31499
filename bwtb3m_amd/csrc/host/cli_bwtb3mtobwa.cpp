// cli_bwtb3mtobwa.cpp -- SURVEY.md 8f row 2: convert X.bwt (+ X.sa) of a pacterm construction to the
// files `bwa bwtupdate` expects.  Same surface as /root/reference/src/bwtb3mtobwa.cpp
// (usage :38-44, MausFmToBwaConversion::rewrite(in,outbwt,outsa,1) :29; pipeline
// /root/reference/README.md:106-111).
//
// BWA layout (recall of bwa's bwt.c bwt_dump_bwt / bwt_dump_sa; no BWA source or sample index is
// available here, so byte identity with `bwa index` output is PARITY-UNPINNED):
//   out.bwt : u64 primary | u64 L2[1..4] | u32 words, 16 symbols each, first symbol in the two MSBs,
//             BWT of the text WITHOUT the entry of the terminator
//   out.sa  : u64 primary | u64 L2[1..4] | u64 sa_intv | u64 seq_len | u64 SA[k*sa_intv], k = 1 .. n_sa-1
//             (n_sa = (seq_len + sa_intv) / sa_intv ; SA[0] = position of the terminator is not stored)
// primary = rank of the rotation starting at text position 0 = index of the terminator in the BWT.
#include "arginfo.h"
#include "../../../include/b3m.h"
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "options.cpp"

static std::vector<uint64_t> read_u64_file(const std::string &fn)
{
	FILE *f = fopen(fn.c_str(), "rb");
	if (!f) throw std::runtime_error("cannot open " + fn + " for reading");
	fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
	std::vector<uint64_t> v((size_t)sz / 8);
	if (sz % 8 || (sz && fread(v.data(), 8, v.size(), f) != v.size())) { fclose(f); throw std::runtime_error(fn + ": not a u64 array file"); }
	fclose(f);
	return v;
}
static void write_all(const std::string &fn, const std::vector<uint8_t> &b)
{
	FILE *f = fopen(fn.c_str(), "wb");
	if (!f) throw std::runtime_error("cannot open " + fn + " for writing");
	if (!b.empty() && fwrite(b.data(), 1, b.size(), f) != b.size()) { fclose(f); throw std::runtime_error("short write on " + fn); }
	fclose(f);
}
static void put64(std::vector<uint8_t> &o, uint64_t v) { for (int i = 0; i < 8; ++i) o.push_back((uint8_t)(v >> (8 * i))); }
static void put32(std::vector<uint8_t> &o, uint32_t v) { for (int i = 0; i < 4; ++i) o.push_back((uint8_t)(v >> (8 * i))); }

int main(int argc, char *argv[])
{
	try {
		b3m::ArgInfo arginfo(argc, argv);
		if (arginfo.help || arginfo.restargs.size() < 3)
			throw std::runtime_error(std::string("usage: ") + argv[0] + " <in.bwt> <out.bwt> <out.sa>\n");
		const std::string in = arginfo.restargs[0], outbwt = arginfo.restargs[1], outsa = arginfo.restargs[2];
		uint64_t n = 0;
		if (b3m_rl_length(in.c_str(), &n)) throw std::runtime_error(b3m_last_error());
		if (n < 2) throw std::runtime_error("input BWT too short");
		std::vector<uint8_t> sym(n);
		if (b3m_rl_read(in.c_str(), sym.data(), n, &n)) throw std::runtime_error(b3m_last_error());
		// exactly one terminator (symbol 0), bases are 1..4  (pacterm)
		uint64_t primary = n, cnt[5] = {0, 0, 0, 0, 0};
		for (uint64_t i = 0; i < n; ++i) {
			if (sym[i] > 4) throw std::runtime_error("symbol outside 0..4: the input is not the BWT of a pacterm text");
			if (sym[i] == 0) { if (primary != n) throw std::runtime_error("more than one terminator in the BWT"); primary = i; }
			cnt[sym[i]]++;
		}
		if (primary == n) throw std::runtime_error("no terminator in the BWT: construct it with inputtype=pacterm");
		const uint64_t seq_len = n - 1;
		uint64_t L2[5] = {0, 0, 0, 0, 0};
		for (int c = 0; c < 4; ++c) L2[c + 1] = L2[c] + cnt[c + 1];
		std::vector<uint8_t> ob;
		put64(ob, primary);
		for (int c = 1; c <= 4; ++c) put64(ob, L2[c]);
		uint64_t words = (seq_len + 15) >> 4, k = 0;
		std::vector<uint32_t> w(words, 0);
		for (uint64_t i = 0; i < n; ++i) if (i != primary) { w[k >> 4] |= (uint32_t)(sym[i] - 1) << ((15 - (k & 15)) << 1); ++k; }
		for (uint64_t i = 0; i < words; ++i) put32(ob, w[i]);
		write_all(outbwt, ob);
		// sampled suffix array written by bwtb3m with bwtonly=0: rate, count, SA[0], SA[rate], ...
		std::string safn = in.size() > 4 && in.compare(in.size() - 4, 4, ".bwt") == 0 ? in.substr(0, in.size() - 4) + ".sa" : in + ".sa";
		std::vector<uint64_t> sa = read_u64_file(safn);
		if (sa.size() < 2 || sa[1] + 2 != sa.size()) throw std::runtime_error(safn + ": not a sampled suffix array (run bwtb3m without bwtonly=1)");
		const uint64_t intv = sa[0];
		const uint64_t n_sa = (seq_len + intv) / intv;
		if (sa[1] < n_sa) throw std::runtime_error(safn + ": too few samples");
		std::vector<uint8_t> os;
		put64(os, primary);
		for (int c = 1; c <= 4; ++c) put64(os, L2[c]);
		put64(os, intv); put64(os, seq_len);
		for (uint64_t i = 1; i < n_sa; ++i) put64(os, sa[2 + i]);
		write_all(outsa, os);
		return EXIT_SUCCESS;
	} catch (std::exception const &ex) {
		std::cerr << ex.what() << std::endl;
		return EXIT_FAILURE;
	}
}
