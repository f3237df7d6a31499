// cli_bwtb3m.cpp -- the bwtb3m command line, identical surface to /root/reference/src/bwtb3m.cpp:
//   bwtb3m [key=value]... <inputfile>      (help text :32-60, error path :67-71, "[M]" line :65)
#include "arginfo.h"
#include "../../../include/b3m.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <sys/resource.h>

#include "options.cpp"   // ArgInfo grammar is header-light: compiled into the CLI as well

static std::string format_time(double s)
{
	const uint64_t is = (uint64_t)s;
	char buf[64]; snprintf(buf, sizeof buf, "%02llu:%02llu:%02llu.%03llu", (unsigned long long)(is / 3600), (unsigned long long)((is / 60) % 60),
	                       (unsigned long long)(is % 60), (unsigned long long)((s - (double)is) * 1000));
	return buf;
}

int main(int argc, char *argv[])
{
	try {
		b3m::ArgInfo arginfo(argc, argv);
		auto t0 = std::chrono::steady_clock::now();
		b3m_options def; b3m_options_init(&def);
		if (arginfo.help || arginfo.restargs.empty()) {
			std::ostringstream str;
			str << "This is bwtb3m (MI355X-native engine) version " << b3m_version() << std::endl;
			str << std::endl;
			str << "usage: " << arginfo.progname << " [options] <inputfile>" << std::endl;
			str << std::endl;
			str << "options:" << std::endl;
			str << "inputtype=[<bytestream>] (bytestream,compactstream,pac,pacterm,lz4,utf-8)" << std::endl;
			str << "outputfilename=[<" << arginfo.default_tmp_file_name() + ".bwt" << ">] (name of output .bwt file)" << std::endl;
			str << "sasamplingrate=[" << def.sasamplingrate << "] sampling rate for sampled suffix array" << std::endl;
			str << "isasamplingrate=[" << def.isasamplingrate << "] sampling rate for sampled inverse suffix array" << std::endl;
			str << "mem=[" << def.mem << "] memory target (0: all free HBM)" << std::endl;
			str << "numthreads=[" << def.numthreads << "] number of threads" << std::endl;
			str << "bwtonly=[" << def.bwtonly << "] compute BWT only (no sampled suffix array and reverse)" << std::endl;
			str << std::string("tmpprefix=[") + arginfo.default_tmp_file_name() + "] (prefix for tmp files)" << std::endl;
			str << "sparsetmpprefix=[tmpprefix] (prefix for sparse gap tmp files)" << std::endl;
			str << "copyinputtomemory=[" << def.copyinputtomemory << "] (copy input file to memory)" << std::endl;
			str << "largelcpthres=[" << def.largelcpthres << "] (large LCP value threshold)" << std::endl;
			str << "verbose=[" << def.verbose << "] (verbosity level)" << std::endl;
			str << "numgpus=[" << def.num_gpus << "] (extension: GPUs of this node to construct on, one host thread and one suffix-range shard each; pac/pacterm)" << std::endl;
			throw std::runtime_error(str.str());
		}
		b3m_options o = def;
		std::string tmpprefix = arginfo.get("tmpprefix", arginfo.default_tmp_file_name());
		std::string inputtype = arginfo.get("inputtype", "bytestream");
		std::string outfn = arginfo.get("outputfilename", tmpprefix + ".bwt");
		std::string sparse = arginfo.get("sparsetmpprefix", tmpprefix);
		o.input_fn = arginfo.restargs[0].c_str();
		o.inputtype = inputtype.c_str();
		o.output_fn = outfn.c_str();
		o.tmpprefix = tmpprefix.c_str();
		o.sparsetmpprefix = sparse.c_str();
		o.bwtonly = (uint32_t)arginfo.get_numeric("bwtonly", def.bwtonly);
		o.mem = arginfo.get_numeric("mem", def.mem);
		o.sasamplingrate = arginfo.get_numeric("sasamplingrate", def.sasamplingrate);
		// default 262144, 64 when bwtonly=1 (/root/reference/ChangeLog:281)
		o.isasamplingrate = arginfo.get_numeric("isasamplingrate", o.bwtonly ? 64 : def.isasamplingrate);
		o.numthreads = (uint32_t)arginfo.get_numeric("numthreads", def.numthreads);
		o.copyinputtomemory = (uint32_t)arginfo.get_numeric("copyinputtomemory", 0);
		o.largelcpthres = arginfo.get_numeric("largelcpthres", def.largelcpthres);
		o.verbose = (uint32_t)arginfo.get_numeric("verbose", 0);
		o.blocksize = arginfo.get_numeric("blocksize", 0);
		o.num_gpus = (int32_t)arginfo.get_numeric("numgpus", (uint64_t)def.num_gpus);
		if (b3m_compute_bwt(&o, 2)) throw std::runtime_error(b3m_last_error());
		struct rusage ru; getrusage(RUSAGE_SELF, &ru);
		double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		std::cerr << "[M] MemUsage(maxrss=" << ru.ru_maxrss / 1024 << "MB) runtime " << format_time(el) << std::endl;
	} catch (std::exception const &ex) {
		std::cerr << ex.what() << std::endl;
		return EXIT_FAILURE;
	}
}
