// cli_sortPreIsa.cpp -- X.preisa ((rank, pos) pairs, host-endian) -> the "sorted isa" the block decoder searches: (pos, rank) pairs in
// position order, 8-byte big-endian numbers.  Same surface as /root/reference/src/sortPreIsa.cpp (:28-34 arguments, :62 the swap,
// :95-96 the output form); the reference sorts through temporary files, this one in memory.
#include "arginfo.h"
#include "formats.h"
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <stdexcept>

#include "options.cpp"

int main(int argc, char *argv[])
{
	try {
		b3m::ArgInfo arginfo(argc, argv);
		if (arginfo.help || arginfo.restargs.size() < 2) throw std::runtime_error(std::string("usage: ") + argv[0] + " [verbose=1] <in.preisa> <out.sortedisa>\n");
		const bool verbose = arginfo.get_numeric("verbose", 1) != 0;
		std::vector<uint8_t> in = b3m::read_file(arginfo.restargs[0]);
		if (in.size() % 16) throw std::runtime_error("Uneven number of words in input file.");
		const uint64_t k = in.size() / 16;
		const uint64_t *w = reinterpret_cast<const uint64_t *>(in.data());
		std::vector<std::pair<uint64_t, uint64_t>> v(k);
		for (uint64_t i = 0; i < k; ++i) v[i] = {w[2 * i + 1], w[2 * i]};        // (pos, rank)
		std::sort(v.begin(), v.end());
		std::vector<uint8_t> out; out.reserve(16 * k);
		auto be = [&](uint64_t x) { for (int b = 7; b >= 0; --b) out.push_back((uint8_t)(x >> (8 * b))); };
		for (uint64_t i = 0; i < k; ++i) {
			if (i && v[i].first == v[i - 1].first) throw std::runtime_error("position sampled twice in " + arginfo.restargs[0]);
			be(v[i].first); be(v[i].second);
		}
		b3m::write_file(arginfo.restargs[1], out.data(), out.size());
		if (verbose) std::cerr << "[V] " << k << " pairs, first pos " << (k ? v[0].first : 0) << " first dif " << (k > 1 ? v[1].first - v[0].first : 0) << std::endl;
		return EXIT_SUCCESS;
	} catch (std::exception const &ex) {
		std::cerr << ex.what() << std::endl;
		return EXIT_FAILURE;
	}
}
