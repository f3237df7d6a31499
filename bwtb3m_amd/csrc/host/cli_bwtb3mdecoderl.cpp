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
			throw std::runtime_error(std::string("usage: ") + argv[0] + " <in.bwt>\n");
		std::string inputtype = arginfo.get("inputtype", "bytestream");
		if (inputtype != "bytestream")   // the reference also knows utf-8; nothing else (:61-67)
			throw std::runtime_error("input type " + inputtype + " is currently not supported for bwtb3mdecode");
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
