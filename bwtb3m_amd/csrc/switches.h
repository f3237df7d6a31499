// switches.h -- every environment switch of libb3m, in ONE place.
//
// The library reads its environment nowhere else: switches_load() fills the struct below (at plan creation, at the start of a run and at
// the stage entry points of the C ABI -- never per kernel launch), and the code reads sw().NAME.  tools/README.md lists the same names.
// A switch is a test switch (small inputs through the kernels of the large geometries), a measurement switch (an older form kept for A/B
// runs) or a deployment knob; NONE of them may change a result.  Switches that do (timing-only experiments that skip work) exist only in
// the experiment build (-DB3M_PHASE_EXPERIMENT, `make experiment` -> libb3m_exp.so): tests/test_cpu_host.py greps the product library for them.
#pragma once
#include <stdint.h>

namespace b3m {

struct Sw {
	bool set = false; long long v = 0;
	long long get(long long dflt) const { return set ? v : dflt; }       // the value, or dflt when the variable is not in the environment
	bool on() const { return set && v != 0; }                             // present and non-zero
};

// name (the variable is B3M_<name>), what it is for
#define B3M_SWITCH_LIST(X)                                                                                                             \
	X(DEBUG,            "1: one line per stage on stderr; 2: per round of the list-based rounds; 3: census of group sizes")            \
	X(NO_TIMELINE,      "measurement: 1 = no per-family HIP events (b3m_plan_kernel_times reports nothing)")                           \
	X(ARENA_CHUNK_MB,   "deployment: size of the physical pieces a one-block plan's arena is mapped from (default 1024; 0 = one hipMalloc)") \
	X(SPARSE_TIES,      "test: cap on the plan's list of tied suffixes of sequential shards (0 = shards finish their repeats themselves)") \
	X(FORCE_SPILL,      "test: S >= 2 sequential suffix-range shards (the host tier) whatever mem= says")                               \
	X(MSD_D1,           "test: level 1 gets this many bits whatever the size (small inputs through the kernels of the large geometries)") \
	X(MSD_D0,           "test: cap on level 0's bits")                                                                                   \
	X(BIG_GRID,         "measurement: workgroups that stride over the listed buckets (default 2048)")                                   \
	X(L0_LOCAL,         "measurement: 0 = the scattering level 0 (counting pass + [digit][tile] scan) instead of the chunk-local one")   \
	X(SHARD_ONEPASS,    "measurement: 0 = suffix-range shards take the two-pass level 0")                                                \
	X(SHARD_TWO_PASS,   "test: bit k set = shard k takes the two-pass level 0 (mixed sets of shards must agree on their borders)")       \
	X(SHARD_KEYS,       "test: 1 = suffix-range shards always take borders by 64-bit window (inside level-0 digits), 0 = never (default: skewed text only)") \
	X(SHARD_WG2,        "measurement: 0 = one-pass shards never take the 512-thread form")                                               \
	X(NO_FINE_SHARDS,   "measurement: 1 = shards do not split their digit range further")                                                \
	X(FINE_MAXS,        "measurement: cap on the extra bits of a shard's fine level 0")                                                  \
	X(L1_ORDER,         "measurement: 1 = level 1's tiles listed tile-major and taken in XCD order (edge lines fetched once)")           \
	X(XCD_ORDER,        "measurement: 0 = workgroups take their items in launch order")                                                  \
	X(NO_LEVEL2,        "measurement: 1 = listed buckets never take level 2")                                                            \
	X(NO_FAST_FINISH,   "measurement: 1 = the generic finish instead of k_finish_fast / k_finish_v3")                                   \
	X(FINISH_V3,        "measurement: 0 = k_finish_fast, 1 = k_finish_v3 (default), 2 = two workgroups per CU, 4 = 1024 threads")        \
	X(NO_FINISH_CHAIN,  "measurement: 1 = the listed buckets' finish in one pass, as before the recursion")                             \
	X(LVL2_MIN,         "test: listed buckets above this many records take level 2 (default 32768)")                                     \
	X(LVL2_AVG,         "measurement: records a level-2 sub-bucket is to average (smaller than the default 3072 only)")                 \
	X(TIE_TWO_STEP,     "measurement: 0 = one thread per word of the tie bitmap instead of collect + fix")                               \
	X(FILL_BY_SLOT,     "measurement: 1 = the overflow ranges' slot list filled by one thread per slot")                                \
	X(TIE_FROM_0,       "measurement: present = the list-based rounds start at symbol 0")                                                \
	X(ISA_EVER_LISTED,  "measurement: 1 = ranks for every slot ever listed, not only those the first key round left")                   \
	X(ISA_FULL,         "measurement: 1 = the whole inverse suffix array is filled before the doubling rounds")                         \
	X(ISA_SEARCH_MAX,   "measurement: unset targets above which the inverse suffix array is filled whole (default n / 64)")             \
	X(TT_COUNT,         "measurement: groups above this many members send a tile of the list rounds to the bitonic network")            \
	X(NO_TIE_TILES,     "measurement: 1 = no LDS tiles in the list-based rounds (radix sorts only)")                                    \
	X(NO_PIECE_SORT,    "measurement: 1 = pieces of 513..8192 slots go through the byte-split passes like larger ones (the form before round 5)")       \
	X(PIECE_CAP,        "measurement: largest piece ordered by one workgroup in LDS: 2048, 4096 (default) or 8192 slots")                             \
	X(NO_TIE_MSD,       "measurement: 1 = large groups through the LSD radix sorts")                                                    \
	X(NO_TREE_DOUBLING, "measurement: 1 = blocks of a merge tree finish long repeats by key extension only")                            \
	X(OVERHANG,         "test: symbols of overhang behind a block of a merge tree (default 2^20)")                                      \
	X(RANK_SB_SHIFT,    "test: log2 of the lines per superblock of the 64-bit rank dictionary (default 16)")                            \
	X(WALK_ONE_LANE,    "measurement: 1 = the gap walk with one lane per chain")                                                        \
	X(RL_SEG,           "test: segment length of the device RL encoder (default 2^31 symbols)")                                         \
	X(HOST_RL,          "measurement: present = the RL container is coded on the host")                                                 \
	X(FILE_MMAP,        "measurement: 1 = output files through a shared mapping instead of pwrite")                                     \
	X(L1_MP,            "measurement: 1 = level 1's gather as two 512-thread workgroups per CU that send the chunk out in passes (k_l1_gather_mp: slower)")

struct Switches {
#define B3M_SW_FIELD(name, doc) Sw name;
	B3M_SWITCH_LIST(B3M_SW_FIELD)
#undef B3M_SW_FIELD
	char DEVICES[256] = {0};          // test: "0,0,.." lists the devices of numgpus=N (one device N times on a one-GPU box)
	// read from the environment by the experiment build only (host/switches.cpp under -DB3M_PHASE_EXPERIMENT); never set in the product library.  WRONG results, timing only:
	Sw FINISH_EXPERIMENT;             // leave k_finish_v3 after phase 1..5
	Sw TT_EXPERIMENT;                 // 1 no ordering, 2 no write-back, 4 leave after the fetch in k_tie_tile
	Sw FINISH_NOMAP;                  // the finish reads a contiguous range instead of its runs
};

const Switches &sw();                 // what the last switches_load() read
void switches_load();                 // reads the environment (plan creation, start of a run, stage entry points)
struct SwitchDoc { const char *name, *doc; };
int switches_describe(const SwitchDoc **out);   // the list above (tools/README.md is checked against it)

}  // namespace b3m
