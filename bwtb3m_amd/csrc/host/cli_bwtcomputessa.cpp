// cli_bwtcomputessa.cpp -- the post-stage of bwtonly=1, same surface as /root/reference/src/bwtcomputessa.cpp:24-58:
//   bwtcomputessa [sasamplingrate=32] [isasamplingrate=32] <X.bwt>
// X.sa and X.isa from X.bwt and the hint file X.preisa (/root/reference/README.md:63-67), by LF walks on the device.  Options of the reference that steer its
// external-memory sort (tmpprefix, copyinputtomemory, threads, maxsortmem, maxtmpfiles, ref_isa, ref_sa) are accepted and unused: nothing leaves HBM here.
#include "arginfo.h"
#include "../../../include/b3m.h"
#include <cstdlib>
#include <iostream>
#include <stdexcept>

#include "options.cpp"

int main(int argc, char *argv[])
{
	try {
		b3m::ArgInfo arginfo(argc, argv);
		if (arginfo.help || arginfo.restargs.size() < 1) {
			std::cerr << "usage: " << argv[0] << " <bwt>" << std::endl;                 // (:25-29)
			return EXIT_FAILURE;
		}
		const uint64_t sarate = arginfo.get_numeric("sasamplingrate", 32), isarate = arginfo.get_numeric("isasamplingrate", 32);
		if (b3m_compute_ssa(arginfo.restargs[0].c_str(), sarate, isarate, arginfo.get_numeric("verbose", 0) ? 2 : -1)) throw std::runtime_error(b3m_last_error());
		return EXIT_SUCCESS;
	} catch (std::exception const &ex) {
		std::cerr << ex.what() << std::endl;
		return EXIT_FAILURE;
	}
}
