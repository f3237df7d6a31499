// arginfo.h -- key=value command line grammar of libmaus2::util::ArgInfo as used by the reference
// CLIs (/root/reference/src/bwtb3m.cpp:29-32): leading key=value tokens are options, the rest are
// positional (restargs); numeric values accept k/m/g suffixes (/root/reference/README.md:42).
#pragma once
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

namespace b3m {

struct ArgInfo {
	std::string progname;
	std::map<std::string, std::string> argmap;
	std::vector<std::string> restargs;
	bool help = false;
	ArgInfo(int argc, char **argv);
	bool has(const std::string &k) const { return argmap.count(k) != 0; }
	std::string get(const std::string &k, const std::string &def) const;
	uint64_t get_numeric(const std::string &k, uint64_t def) const;   // getValueUnsignedNumeric
	// <prog>_<host>_<pid>_<time>  (/root/reference/README.md:39,45)
	std::string default_tmp_file_name() const;
};
uint64_t parse_numeric(const std::string &v);   // throws on junk

}  // namespace b3m
