// cli_checkbwt.cpp -- the reference's verifier, same surface as /root/reference/src/checkbwt.cpp:248-285:
//   checkbwt [-i <inputtype>] <X.bwt> <text>
// walks the LF chains of X.bwt between the (rank,pos) pairs of X.preisa (or X.isa) over the text on the device and prints "[V] gok=1" (or 0) on
// stderr; like the reference it leaves with EXIT_SUCCESS once the walk has run (:243), whatever gok says, and with EXIT_FAILURE on an exception.
#include "../../../include/b3m.h"
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

int main(int argc, char *argv[])
{
	try {
		std::string itype = "bytestream";
		std::vector<std::string> rest;
		for (int i = 1; i < argc; ++i) {
			const std::string a = argv[i];
			if (a == "-i" && i + 1 < argc) itype = argv[++i];
			else if (a.rfind("-i", 0) == 0 && a.size() > 2) itype = a.substr(2);
			else if (a == "-V" && i + 1 < argc) ++i;                                  // (progress interval of the reference: accepted, unused)
			else if (a == "-h" || a == "--help") rest.clear(), i = argc;
			else rest.push_back(a);
		}
		if (rest.size() < 2) throw std::runtime_error(std::string("usage: ") + argv[0] + " [-i <inputtype>] <in.bwt> <text>");
		uint64_t mm[2] = {0, 0};
		if (b3m_check_bwt(rest[0].c_str(), rest[1].c_str(), itype.c_str(), 2, mm)) throw std::runtime_error(b3m_last_error());
		return EXIT_SUCCESS;
	} catch (std::exception const &ex) {
		std::cerr << ex.what() << std::endl;
		return EXIT_FAILURE;
	}
}
