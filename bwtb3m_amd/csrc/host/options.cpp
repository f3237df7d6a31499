// options.cpp -- BwtMergeSortOptions defaults and ArgInfo grammar.
#include "arginfo.h"
#include "../../../include/b3m.h"
#include <cctype>
#include <cstring>
#include <ctime>
#include <stdexcept>
#include <thread>
#include <unistd.h>

namespace b3m {

uint64_t parse_numeric(const std::string &v)
{
	if (v.empty()) throw std::runtime_error("empty numeric value");
	size_t i = 0; uint64_t x = 0;
	while (i < v.size() && isdigit((unsigned char)v[i])) { x = x * 10 + (uint64_t)(v[i] - '0'); ++i; }
	if (i == 0) throw std::runtime_error("cannot parse numeric value " + v);
	if (i < v.size()) {
		char c = (char)tolower((unsigned char)v[i]);
		uint64_t mul = c == 'k' ? 1024ull : c == 'm' ? 1024ull * 1024 : c == 'g' ? 1024ull * 1024 * 1024 : c == 't' ? 1024ull * 1024 * 1024 * 1024 : 0;
		if (!mul || i + 1 != v.size()) throw std::runtime_error("cannot parse numeric value " + v);
		x *= mul;
	}
	return x;
}

ArgInfo::ArgInfo(int argc, char **argv)
{
	progname = argc ? argv[0] : "";
	int i = 1;
	for (; i < argc; ++i) {
		std::string a = argv[i];
		if (a == "-h" || a == "--help") { help = true; continue; }
		size_t eq = a.find('=');
		if (eq == std::string::npos || eq == 0) break;
		argmap[a.substr(0, eq)] = a.substr(eq + 1);
	}
	for (; i < argc; ++i) restargs.push_back(argv[i]);
}
std::string ArgInfo::get(const std::string &k, const std::string &def) const
{
	auto it = argmap.find(k);
	return it == argmap.end() ? def : it->second;
}
uint64_t ArgInfo::get_numeric(const std::string &k, uint64_t def) const
{
	auto it = argmap.find(k);
	return it == argmap.end() ? def : parse_numeric(it->second);
}
std::string ArgInfo::default_tmp_file_name() const
{
	char host[256] = "localhost";
	gethostname(host, sizeof host - 1);
	std::string p = progname;
	size_t sl = p.find_last_of('/');
	if (sl != std::string::npos) p = p.substr(sl + 1);
	return p + "_" + host + "_" + std::to_string((long)getpid()) + "_" + std::to_string((long)time(nullptr));
}

}  // namespace b3m

extern "C" void b3m_options_init(b3m_options *o)
{
	if (!o) return;
	memset(o, 0, sizeof *o);
	o->inputtype = "bytestream";
	o->mem = 0;                               // reference default 2 GiB of host RAM; here 0 = all free HBM
	o->sasamplingrate = 32;                   // getDefaultSaSamplingRate
	o->isasamplingrate = 262144;              // getDefaultIsaSamplingRate
	o->largelcpthres = 16384;                 // getDefaultLargeLCPThres
	o->numthreads = std::max(1u, std::thread::hardware_concurrency());
	o->bwtonly = 0;
	o->copyinputtomemory = 0;
	o->verbose = 0;
	o->num_gpus = 1;
	o->blocksize = 0;
}
