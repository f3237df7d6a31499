// cli_bwtb3mdecoderl.cpp -- named consumer of the .bwt container: decode the run-length file to
// raw bytes on stdout, same surface as /root/reference/src/bwtb3mdecoderl.cpp (:24-68, usage :75-84).
#include "arginfo.h"
#include "../../../include/b3m.h"
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <vector>

#include "options.cpp"

int main(int argc, char *argv[])
{
	try {
		b3m::ArgInfo arginfo(argc, argv);
		if (arginfo.help || arginfo.restargs.size() < 1)
			throw std::runtime_error(std::string("usage: ") + argv[0] + " [inputtype=bytestream|utf-8] <in.bwt>\n");
		std::string inputtype = arginfo.get("inputtype", "bytestream");
		if (inputtype != "bytestream" && inputtype != "utf-8")   // the two the reference knows (:30-67)
			throw std::runtime_error("input type " + inputtype + " is currently not supported for bwtb3mdecode");
		if (inputtype == "utf-8") {
			// every symbol written as the UTF-8 form of its code point (:47-57)
			uint64_t n = 0;
			if (b3m_rl_length(arginfo.restargs[0].c_str(), &n)) throw std::runtime_error(b3m_last_error());
			std::vector<int64_t> sym(n);
			if (b3m_rl_read_symbols(arginfo.restargs[0].c_str(), sym.data(), n, &n)) throw std::runtime_error(b3m_last_error());
			std::vector<uint8_t> out; out.reserve(n + n / 2);
			for (uint64_t i = 0; i < n; ++i) {
				const uint32_t c = (uint32_t)sym[i];
				if (sym[i] < 0 || sym[i] > 0x10FFFF) throw std::runtime_error("unsupported character value " + std::to_string(sym[i]) + " for inputtype=utf-8");
				if (c < 0x80) out.push_back((uint8_t)c);
				else if (c < 0x800) { out.push_back((uint8_t)(0xC0 | (c >> 6))); out.push_back((uint8_t)(0x80 | (c & 0x3F))); }
				else if (c < 0x10000) { out.push_back((uint8_t)(0xE0 | (c >> 12))); out.push_back((uint8_t)(0x80 | ((c >> 6) & 0x3F))); out.push_back((uint8_t)(0x80 | (c & 0x3F))); }
				else { out.push_back((uint8_t)(0xF0 | (c >> 18))); out.push_back((uint8_t)(0x80 | ((c >> 12) & 0x3F))); out.push_back((uint8_t)(0x80 | ((c >> 6) & 0x3F))); out.push_back((uint8_t)(0x80 | (c & 0x3F))); }
			}
			fwrite(out.data(), 1, out.size(), stdout);
			fflush(stdout);
			return EXIT_SUCCESS;
		}
		uint64_t n = 0;
		if (b3m_rl_length(arginfo.restargs[0].c_str(), &n)) throw std::runtime_error(b3m_last_error());
		std::vector<uint8_t> buf(n);
		if (b3m_rl_read(arginfo.restargs[0].c_str(), buf.data(), n, &n)) {
			// symbols >= 256 are rejected in bytestream mode (:36-42)
			throw std::runtime_error(std::string("unsupported character value for inputtype=bytestream: ") + b3m_last_error());
		}
		fwrite(buf.data(), 1, n, stdout);
		fflush(stdout);
		return EXIT_SUCCESS;
	} catch (std::exception const &ex) {
		std::cerr << ex.what() << std::endl;
		return EXIT_FAILURE;
	}
}
