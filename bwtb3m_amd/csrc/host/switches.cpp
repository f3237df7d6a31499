// switches.cpp -- the one place where libb3m reads its environment (../switches.h).
#include "../switches.h"
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace b3m {

namespace {
Switches g_sw;
std::mutex g_mu;
bool g_loaded = false;
Sw read_one(const char *var)
{
	Sw s;
	if (const char *e = getenv(var)) { s.set = true; s.v = strtoll(e, nullptr, 0); }
	return s;
}
void load_locked()
{
	Switches n;
#define B3M_SW_READ(name, doc) n.name = read_one("B3M_" #name);
	B3M_SWITCH_LIST(B3M_SW_READ)
#undef B3M_SW_READ
	if (const char *e = getenv("B3M_DEVICES")) { strncpy(n.DEVICES, e, sizeof(n.DEVICES) - 1); }
#ifdef B3M_PHASE_EXPERIMENT
	n.FINISH_EXPERIMENT = read_one("B3M_FINISH_EXPERIMENT");
	n.TT_EXPERIMENT = read_one("B3M_TT_EXPERIMENT");
	n.FINISH_NOMAP = read_one("B3M_FINISH_NOMAP");
#endif
	g_sw = n; g_loaded = true;
}
const SwitchDoc g_docs[] = {
#define B3M_SW_DOC(name, doc) {"B3M_" #name, doc},
	B3M_SWITCH_LIST(B3M_SW_DOC)
#undef B3M_SW_DOC
	{"B3M_DEVICES", "test: \"0,0,..\" lists the devices of numgpus=N (one device N times on a one-GPU box)"},
};
}  // namespace

void switches_load() { std::lock_guard<std::mutex> g(g_mu); load_locked(); }
const Switches &sw()
{
	if (!g_loaded) { std::lock_guard<std::mutex> g(g_mu); if (!g_loaded) load_locked(); }
	return g_sw;
}
int switches_describe(const SwitchDoc **out) { *out = g_docs; return (int)(sizeof(g_docs) / sizeof(g_docs[0])); }

}  // namespace b3m
