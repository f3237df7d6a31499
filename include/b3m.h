/*
 * include/b3m.h -- C ABI of the MI355X-native BWT-construction engine (libb3m.so).
 *
 * Drop-in boundary for the hot path of gt1/bwtb3m.  The reference has no FFI of its
 * own; its only library seam for this path is the static C++ call
 *     libmaus2::suffixsort::bwtb3m::BwtMergeSortOptions options(arginfo);
 *     libmaus2::suffixsort::bwtb3m::BwtMergeSort::computeBwt(options,&std::cerr);
 * at /root/reference/src/bwtb3m.cpp:62-63.  b3m_options / b3m_compute_bwt replace
 * exactly that pair; the stage-level entry points below replace the libmaus2
 * internals that call reaches so that each stage can be
 * checked against the oracle on its own.  INTEGRATION.md shows the reference-side
 * binding.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary: every function returns 0 on success,
 *     non-zero on error; b3m_last_error() gives the message (thread-local).
 *   - pointers named d_* are HIP device pointers on the current device; everything
 *     else is host memory.  All buffers are caller-owned unless stated otherwise.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - blocking, not re-entrant per process (matches the reference).
 *   - symbols: bytestream 0..255; pac 0..3; pacterm 1..4 plus one 0 at position n-1
 *     (/root/reference/src/bwtdecodeblock.cpp:307-310: 0 = terminator, sym-1 = base).
 *   - there is NO CPU fallback: without a usable HIP device every compute entry
 *     point fails with an error.
 */
#ifndef B3M_H
#define B3M_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define B3M_VERSION 1

/* input types, /root/reference/src/bwtb3m.cpp:43, enum use at /root/reference/src/checkbwt.cpp:257-270 */
enum {
	B3M_INPUT_BYTESTREAM = 0,
	B3M_INPUT_COMPACTSTREAM = 1, /* b3m_compute_bwt / b3m_compact_*: 2-bit files as pac, other widths through the byte paths */
	B3M_INPUT_PAC = 2,
	B3M_INPUT_PACTERM = 3,
	B3M_INPUT_LZ4 = 4,           /* parsed, not yet constructed */
	B3M_INPUT_UTF8 = 5,          /* b3m_compute_bwt: the code points that occur are ranked; up to 256 of them: plans take the ranks as bytestream, more: as symbols32 */
	B3M_INPUT_SYMBOLS32 = 6      /* plans only (no file type of the reference): n uint32 symbols below 2^21, circular, no terminator; what a utf-8 text of more than
	                                256 distinct code points is ranked into (/root/reference/src/bwtb3mdecoderl.cpp:47-57 writes the code points back).  One block,
	                                n < 2^32; b3m_plan_run reads 4n bytes and writes n uint32 symbols (4n bytes) to d_bwt_out */
};

int         b3m_version(void);
const char *b3m_last_error(void);
/* The library's environment switches (csrc/switches.h is the one place that reads the environment: at plan creation, at the start of a run and at the stage entry
   points, never per kernel launch): name / description of switch i, NULL behind the last one.  None of them changes a result; timing-only experiments that do are
   compiled into the experiment build alone (`make experiment`). */
const char *b3m_switch_name(int i);
const char *b3m_switch_doc(int i);

/* parseInputType: name -> enum, -1 (and error text of /root/reference/src/checkbwt.cpp:274) if unknown */
int         b3m_parse_input_type(const char *name);
const char *b3m_input_type_name(int inputtype);

/* ---- whole job: replaces BwtMergeSortOptions + BwtMergeSort::computeBwt ------------------- */

/* Mirrors BwtMergeSortOptions (getters printed at /root/reference/src/bwtb3m.cpp:43-56,
 * defaults /root/reference/README.md:38-49).  Strings are caller-owned, NUL-terminated;
 * NULL/empty = default. */
typedef struct b3m_options {
	const char *input_fn;          /* positional <inputfile> */
	const char *inputtype;         /* default "bytestream" */
	const char *output_fn;         /* outputfilename=, default <tmpprefix>.bwt */
	const char *tmpprefix;         /* tmpprefix= */
	const char *sparsetmpprefix;   /* sparsetmpprefix=, default tmpprefix (unused: gap runs stay in HBM) */
	uint64_t    mem;               /* mem=, memory target; here: HBM budget, 0 = all free HBM */
	uint64_t    sasamplingrate;    /* default 32 */
	uint64_t    isasamplingrate;   /* default 262144 (64 when bwtonly=1, ChangeLog:281) */
	uint64_t    largelcpthres;     /* default 16384 (not needed on the GPU path, accepted) */
	uint32_t    numthreads;        /* host threads for file I/O + RL encoding */
	uint32_t    bwtonly;           /* default 0 */
	uint32_t    copyinputtomemory; /* accepted; the packed input is always HBM-resident */
	uint32_t    verbose;
	int32_t     num_gpus;          /* extension (CLI key numgpus=): GPUs of this node to construct on, devices 0 .. num_gpus-1 (default 1).  > 1: one host thread
	                                  per device, each with one suffix-range shard of the text (pac / pacterm, n >= 2^18; no exchange step; a shard is a range of
	                                  32-base windows, so its borders may fall inside a level-0 digit of a skewed text; long repeats are finished over the
	                                  gathered LISTS of tied suffixes, b3m_plan_list_*); the slices are gathered on device 0, which writes the same files as
	                                  num_gpus = 1.  A shard that still refuses its range (one window above a shard's workspace) sends the job to one device with a [W] line.  (One process per GPU over RCCL instead: bwtb3m_amd.distributed.) */
	uint64_t    blocksize;         /* extension: leaf block size in symbols, 0 = derive from mem */
} b3m_options;

void b3m_options_init(b3m_options *o);                       /* getDefault*() values */
/* file -> X.bwt (+ X.hist, X.meta, X.preisa | X.sa/X.isa), log lines to log_fd (2 = stderr) */
int  b3m_compute_bwt(const b3m_options *o, int log_fd);
/* the (rank,pos) sampling rate of the plan b3m_compute_bwt creates for these options (0 = the mode's default: 262144, or 64 with
 * bwtonly=1, /root/reference/src/bwtb3m.cpp:46, ChangeLog:281) and a text of n symbols.  bwtonly=0 samples more densely than the .isa
 * file needs: the samples are also the anchors of the LF walks that make .sa/.isa (/root/reference/src/hwtPreIsaToIsa.cpp:107-172), one
 * chain per sample, and must fill the GPU: the sparsest of 16384 / 4096 / 1024 that divides the rate and leaves 2^17 chains (short
 * texts: 1024).  bench.py times a plan created with exactly this rate. */
uint64_t b3m_plan_sampling_rate(uint64_t isasamplingrate, int bwtonly, uint64_t n);

/* ---- device-resident construction (what bench.py times) ----------------------------------- */

typedef struct b3m_plan b3m_plan;

/* n = number of symbols (pacterm: bases + 1).  blocksize = leaf block size in symbols
 * (0 = single block if it fits `mem`, else derived from mem).  isasamplingrate = 0: no
 * (rank,pos) samples.  All workspace is allocated here, none in b3m_plan_run. */
b3m_plan *b3m_plan_create(uint64_t n, int inputtype, uint64_t blocksize, uint64_t mem,
                          uint64_t isasamplingrate);
/* Suffix-range shard for N GPUs holding the same text (multi-GPU without an exchange step): the plan
 * of shard k sorts only the suffixes whose rank lies in k's range -- whole classes of the leading 8
 * bits, balanced from the histogram of the text, identical on every rank -- and b3m_plan_run writes
 * that contiguous slice of the BWT (result.slice_offset / slice_len; d_bwt_out needs
 * b3m_plan_slice_capacity() bytes).  pac/pacterm, n >= 2^18. */
b3m_plan *b3m_plan_create_sharded(uint64_t n, int inputtype, int shard, int nshards, uint64_t mem,
                                  uint64_t isasamplingrate);
uint64_t  b3m_plan_slice_capacity(const b3m_plan *p);
void      b3m_plan_destroy(b3m_plan *p);
uint64_t  b3m_plan_num_blocks(const b3m_plan *p);
/* HOST TIER (role of the reference's mem= / tmpprefix= files, /root/reference/README.md:42,46, /root/reference/src/bwtb3m.cpp:47,53): a pac / pacterm plan whose
 * memory target no merge tree can meet (the tree's text, two BWT arrays, gt bits and rank lists are 7.25 bytes per symbol whatever the blocks) sorts the text
 * as SEQUENTIAL suffix-range shards instead: HBM holds the text (0.25 B/symbol) and one shard's sort; b3m_plan_run_host sends every shard's slice of the BWT to
 * (pinned) host memory through two slice buffers while the next shard sorts, so the BWT is never in HBM as a whole.  Returns the number of shards, 0 for any
 * other plan.  (b3m_plan_run of such a plan still writes the caller's n device bytes.)  Long repeats are finished shard by shard, by key-extension rounds. */
int       b3m_plan_host_tier(const b3m_plan *p);
/* "" or what the caller should be told about the plan: a mem= target below the footprint of the smallest blocks is exceeded, and said
 * so here (b3m_compute_bwt prints it as a [W] line whatever verbose= says; mem= itself: /root/reference/README.md:42) */
const char *b3m_plan_warning(const b3m_plan *p);
uint64_t  b3m_plan_workspace_bytes(const b3m_plan *p);
uint64_t  b3m_plan_text_bytes(const b3m_plan *p);            /* packed input size in bytes */
/* the plan's own resident text buffer (device memory, b3m_plan_text_bytes() bytes and padding behind them): a text loaded HERE and passed
 * to b3m_plan_run as d_text is used in place (no copy into the plan: 0.75 GB read + written at 3 Gbase); any other d_text is copied here */
void     *b3m_plan_text_buffer(b3m_plan *p);

/* d_text: packed input exactly as in the file (bytestream: n bytes; pac/pacterm: BWA .pac
 * payload, base i in byte i>>2 at bit ((~i)&3)<<1), b3m_plan_text_bytes() readable bytes.
 * d_bwt_out: n bytes, one symbol per byte, in rank order (symbols32: n uint32, text and result). */
int b3m_plan_run(b3m_plan *p, const void *d_text, uint8_t *d_bwt_out, void *stream);

/* The same pass from HOST memory to HOST memory: what SURVEY.md 8(d) calls end to end (input in the page cache -> BWT symbols in
 * host memory; role of the file -> memory and memory -> encoder hand-offs around BwtMergeSort::computeBwt,
 * /root/reference/src/bwtb3m.cpp:63).  h_text: b3m_plan_text_bytes() bytes, h_bwt_out: n bytes; pinned memory (b3m_host_alloc) makes
 * both copies asynchronous.  One block: the BWT leaves the device in slices of whole buckets, in rank order, while later buckets are
 * still being sorted; a pacterm text of 64 MB and more arrives in 8 pieces and the (chunk-local) level 0 of the sort runs behind the copy,
 * piece by piece.  times (may be NULL): HIP-event splits of this run. */
typedef struct b3m_e2e_times {
	double h2d_ms;         /* packed text, host -> HBM */
	double construct_ms;   /* first kernel .. last kernel (slices of the BWT already leave during this) */
	double d2h_tail_ms;    /* device -> host copies still running after the last kernel */
	double total_ms;       /* host wall clock: call entry (stream idle) .. last byte in host memory */
	uint64_t streamed_bytes; /* bytes of the BWT that left in slices during construction (0: one copy at the end) */
} b3m_e2e_times;
int   b3m_plan_run_host(b3m_plan *p, const void *h_text, uint8_t *h_bwt_out, b3m_e2e_times *times, void *stream);
/* pac / pacterm plans: b3m_plan_run_host hands the BWT over as it takes the text, 2 bits per symbol in the .pac layout (4 symbols per
 * byte, the first in the top two bits: h_bwt_out then holds (n+3)/4 bytes).  pac: the symbol values; pacterm: symbol - 1, and the one
 * slot that holds the terminator (rank b3m_result.p0) reads 0.  A quarter of the bytes cross PCIe; the encoder behind the hand-off
 * (RLEncoderStd, /root/reference/src/lcpbit.cpp:3677-3681) consumes symbols either way. */
int   b3m_plan_set_packed_output(b3m_plan *p, int on);
void *b3m_host_alloc(uint64_t bytes);     /* pinned host memory (hipHostMalloc) */
void  b3m_host_free(void *p);

/* row f4, inputtype=compactstream (producers /root/reference/src/fagzToCompact4.cpp:113 (2 bits), fagzToCompact.cpp:85 (3 bits)): a libmaus2
 * CompactArray file.  Layout [UPSTREAM-RECALL, unpinned]: four big-endian u64 (bits per symbol, number of symbols, number of 64-bit words,
 * the word count again), then the words, big-endian, symbols from the most significant bit down.  2-bit files are sorted like a .pac
 * payload (same bit order); other widths are decoded to one byte per symbol on the device and take the byte paths.
 * Circular, no terminator (as bytestream).  b3m_compute_bwt accepts the type; at plan level: b3m_compact_info + (bits == 2 ?
 * plan of type pac on the word stream : b3m_compact_decode + plan of type bytestream). */
int b3m_compact_info(const uint8_t *file_head32, uint64_t file_size, uint32_t *bits_out, uint64_t *n_out, uint64_t *data_off_out);
int b3m_compact_decode(const void *d_words, uint64_t n, uint32_t bits, uint8_t *d_out, void *stream);

/* Suffix-range shards and long repeats (the reference bounds the repeat cost of a block with lcpnext / largelcpthres,
 * /root/reference/src/bwtb3m.cpp:55, ChangeLog:205-211).  A shard's b3m_plan_run stops its local sort after a few key-extension rounds; if
 * b3m_plan_tie_pending reports suffixes still tied on ANY shard, the caller gathers every shard's suffix array slice (b3m_plan_tie_export; over
 * RCCL all-gather between GPUs) and tied-slot list, and every shard calls b3m_plan_tie_finish on the same global arrays: prefix doubling,
 * O(log longest repeat) rounds, after which the shard's slice of the BWT (and p0, samples) is final.  bwtb3m_amd.distributed does this. */
int b3m_plan_set_keep_sa(b3m_plan *p, int on);   /* the next runs keep the shard's whole suffix array (a BWT-only run writes only what k_tie_fix reads) */
int b3m_plan_has_sa(const b3m_plan *p);          /* did the last run keep it */
int b3m_plan_tie_pending(const b3m_plan *p, uint64_t *count_out);
int b3m_plan_tie_export(b3m_plan *p, uint32_t *d_sa_slice_out, uint32_t *d_pairs_out, void *stream);
int b3m_plan_tie_finish(b3m_plan *p, uint32_t *d_sa_global, const uint32_t *d_pairs_all, uint64_t total, uint64_t my_first, uint8_t *d_bwt_slice, void *stream);

/* The same bound WITHOUT the shards' suffix arrays (round 4; the only form for texts of 2^32 symbols and more): with b3m_plan_set_tie_mode(plan, B3M_TIES_LIST) a shard's run
 * stops after the same few rounds and keeps a LIST of what is left -- (global rank | first-of-group flag in bit 63, position) per tied suffix, in rank order -- plus the
 * suffixes its direct comparisons placed for good among long repeats ("placed": their rank is final, the doubling rounds must be able to look it up).  The caller gathers
 * every shard's records (b3m_plan_list_export; 16 bytes per suffix on the list instead of 4 per suffix of the text: one small all-gather between GPUs), concatenates the tied
 * ones in shard order, and every shard calls b3m_plan_list_finish on the union: prefix doubling on the list alone (the rank of the suffix d - X symbols on is looked up among
 * the listed ones), after which the shard's slice of the BWT, p0 and samples are final.  b3m_plan_list_counts: complete = 0 means the shard's list overflowed (it then
 * finished its own repeats by key extension, or dropped placed entries): the union is still finished correctly, by key extension (not bounded).
 * B3M_TIES_LOCAL: the shard finishes its long repeats itself by key extension (no collective step at all).  b3m_compute_bwt (numgpus > 1) uses the list. */
#define B3M_TIES_SA    0    /* b3m_plan_tie_* over all shards' suffix arrays (default below 2^32 symbols) */
#define B3M_TIES_LIST  1
#define B3M_TIES_LOCAL 2    /* (default for shards of texts of 2^32 symbols and more) */
int b3m_plan_set_tie_mode(b3m_plan *p, int mode);
int b3m_plan_list_counts(const b3m_plan *p, uint64_t *tied_out, uint64_t *placed_out, uint64_t *depth_out, int *complete_out);
int b3m_plan_list_fits(const b3m_plan *p, uint64_t tied_total, uint64_t placed_total);      /* 1: this shard's workspace can finish a union list of that size */
int b3m_plan_list_export(b3m_plan *p, uint64_t *d_tied_out /* 2 * tied */, uint64_t *d_placed_out /* 2 * placed */, void *stream);
int b3m_plan_list_finish(b3m_plan *p, const uint64_t *d_tied_all, uint64_t tied_total, const uint64_t *d_placed_all, uint64_t placed_total, uint64_t depth_min, uint64_t depth_max,
                         int complete, uint8_t *d_bwt_slice, void *stream);

typedef struct b3m_result {
	uint64_t n, num_blocks, levels;
	uint64_t p0;                 /* rank of the rotation starting at position 0 (pacterm: BWA "primary") */
	uint64_t num_samples;        /* (rank,pos) pairs available via b3m_plan_samples */
	uint64_t tie_rounds;         /* key-extension rounds needed by the leaf sorts (max) */
	uint64_t tie_elements;       /* suffixes that needed more than the first key (sum) */
	double   ms_total, ms_sort, ms_gap, ms_interleave;   /* HIP-event times of the last run */
	uint64_t slice_offset, slice_len;   /* suffix-range shard: global rank of the first symbol written, and how many (else 0, n) */
	uint64_t has_p0;                    /* 0 if the rotation starting at 0 lies in another shard */
	uint64_t tie_overhang;              /* blocks of a merge tree with long repeats: symbols behind the block sorted with it (max; 0 = no block needed it) */
	uint64_t tie_slow;                  /* suffixes whose repeat reached past the overhang: finished by key-extension rounds (sum) */
} b3m_result;
int b3m_plan_result(const b3m_plan *p, b3m_result *out);
/* copies min(cap, num_samples) (rank,pos) u64 pairs to host memory, in position order (pos = 0, rate, 2*rate, ...) */
int b3m_plan_samples(const b3m_plan *p, uint64_t *pairs_out, uint64_t cap);

/* row a4: symbol histogram of the last run, hist256[sym] (the content of X.hist) */
int b3m_plan_histogram(const b3m_plan *p, uint64_t *hist256);

/* SURVEY.md 8f rows 1 + 3: parallel LF walks over the final BWT (d_bwt = output of the plan's last run)
 * between the plan's (rank,pos) samples.  Role of BwtComputeSSA::computeSSA
 * (/root/reference/src/bwtcomputessa.cpp:39-51) and of checkbwt's verification walk
 * (/root/reference/src/checkbwt.cpp:176-243).  mismatches_out[0] = positions where the BWT symbol differs
 * from the text, [1] = chains that did not arrive at the next anchor's rank (both 0 for a correct BWT).
 * sarate > 0: sa_out[i] = SA[i*sarate], ceil(n/sarate) entries (host).  Unsharded plans (texts of 2^32 symbols and more included: the plan's
 * sequential shards leave one BWT; the rank dictionary counts in 64 bits, so a symbol may occur 2^32 times and more). */
int b3m_plan_lf_walk(b3m_plan *p, const uint8_t *d_bwt, uint64_t sarate, uint64_t *sa_out,
                     uint64_t *mismatches_out, void *stream);

/* the same with the SA samples left in device memory (d_sa_out: ceil(n/sarate) u64): b3m_compute_bwt streams them into X.sa from there; pac / pacterm */
int b3m_plan_lf_walk_dev(b3m_plan *p, const uint8_t *d_bwt, uint64_t sarate, uint64_t *d_sa_out, uint64_t *mismatches_out, void *stream);
/* the pinned host image of the last run's (rank,pos) samples, in place: slot i belongs to position i * rate; a slot that reads (~0, .) is another shard's
 * (suffix-range shards only).  Valid until the plan runs again or is destroyed.  X.preisa is this image (/root/reference/src/hwtPreIsaToIsa.cpp:64-77). */
int b3m_plan_samples_raw(b3m_plan *p, const uint64_t **slots_out, uint64_t *nslots_out);

/* the DEVICE image of the same table (slots as above) and the number of slots nobody wrote (empty_out may be NULL).  A table above 64 MB (dense sampling:
 * bwtonly=1 samples every 64th position, 750 MB at 3 Gbase) is not copied to the host by b3m_plan_run at all; the host-side accessors fetch it when asked,
 * b3m_compute_bwt streams X.preisa from here. */
int b3m_plan_samples_device(b3m_plan *p, const uint64_t **d_slots_out, uint64_t *nslots_out, uint64_t *empty_out, void *stream);

/* The same walks as a stage of its own, over a BWT that no single plan produced -- the slices of suffix-range shards gathered on one GPU (b3m_compute_bwt
 * with num_gpus > 1).  pac / pacterm.  d_text: the packed text on the current device, 8 readable bytes behind it; p0: rank of the terminator entry (pacterm);
 * hist256: symbol counts of the text (X.hist); sample_pairs: num_samples (rank,pos) pairs in position order, the first at position 0 (host).  The rank
 * dictionary counts in 64 bits (superblock table + 32-bit line headers): a symbol may occur 2^32 times and more.  sa_out (host) and/or d_sa_out (device),
 * each ceil(n/sarate) u64, may be NULL. */
int b3m_lf_walk(const void *d_text, uint64_t n, int inputtype, const uint8_t *d_bwt, uint64_t p0, const uint64_t *hist256, const uint64_t *sample_pairs,
                uint64_t num_samples, uint64_t sarate, uint64_t *sa_out, uint64_t *d_sa_out, uint64_t *mismatches_out, void *stream);

/* The walks as one job description: what b3m_lf_walk takes, plus ISA samples (isa_out[i] = rank of the rotation at position i * isarate: host memory, ceil(n/isarate)
 * entries) and a text that may be absent (d_text = NULL: symbols are not compared, chain closures still are).  Role of BwtComputeSSA::computeSSA
 * (/root/reference/src/bwtcomputessa.cpp:39-51): X.sa and X.isa at any rates from X.bwt and the sparser pairs of X.preisa, no text needed. */
typedef struct b3m_lf_job {
	const void    *d_text;          /* packed text on the device, or NULL */
	uint64_t       n; int inputtype;
	const uint8_t *d_bwt; uint64_t p0; const uint64_t *hist256;
	const uint64_t *sample_pairs; uint64_t num_samples;      /* (rank,pos) in position order, the first at position 0 (host) */
	uint64_t       sarate;  uint64_t *sa_out;  uint64_t *d_sa_out;
	uint64_t       isarate; uint64_t *isa_out;
} b3m_lf_job;
int b3m_lf_walk_job(const b3m_lf_job *job, uint64_t *mismatches_out, void *stream);

/* One device's share of the same job (b3m_compute_bwt with num_gpus > 1 splits the LF walks that make X.sa over its devices: /root/reference/src/bwtb3m.cpp:48-50,
 * numthreads= is the reference's knob for the same stage): the chains [chain_first, chain_first + chain_count) -- chain k runs from sample k + 1 back to sample k --
 * over the whole BWT on the current device.  job->d_sa_out (device, ceil(n/sarate) u64) is set to ~0 and receives the SA samples these chains meet; job->sa_out and
 * job->isa_out must be NULL, job->d_text is required.  b3m_merge_samples puts shares together: d_dst[i] = d_src[i] where d_src[i] != ~0. */
int b3m_lf_walk_chains(const b3m_lf_job *job, uint64_t chain_first, uint64_t chain_count, uint64_t *mismatches_out, void *stream);
int b3m_merge_samples(uint64_t *d_dst, const uint64_t *d_src, uint64_t count, void *stream);
/* the (rank,pos) sample tables of suffix-range shards (b3m_plan_samples_device: slot i = position i * rate, rank ~0 = another shard's) put together on the device:
 * slots of d_src that hold a rank go to d_dst (npairs slots of 16 bytes each, current device); stats_out (host): [0] += slots both had ranked (must stay 0),
 * [1] = slots of d_dst still empty afterwards. */
int b3m_merge_sample_pairs(uint64_t *d_dst, const uint64_t *d_src, uint64_t npairs, uint64_t *stats_out, void *stream);
/* bwtcomputessa as one call (/root/reference/README.md:63-67: after bwtonly=1, X.preisa is the hint file for "the bwtcomputessa program to compute a sampled suffix array
 * and sampled inverse suffix array"): X.bwt is decoded, the chains between the pairs of X.preisa are walked on the device, X.sa and X.isa are written (layout of
 * /root/reference/src/sasubsample.cpp:34-57).  Defaults of the reference's program: both rates 32.  bin/bwtcomputessa is the program. */
int b3m_compute_ssa(const char *bwt_fn, uint64_t sasamplingrate, uint64_t isasamplingrate, int log_fd);

/* The reference's verifier as one call (role of `checkbwt [-i type] X.bwt text`, /root/reference/src/checkbwt.cpp:26-243): X.bwt is decoded, BWT and text go to
 * HBM, and one LF chain per (rank,pos) pair of X.preisa -- or of X.isa, where only that exists -- is walked over the text on the device; every symbol and every
 * chain closure is checked.  mismatches_out as b3m_plan_lf_walk; log lines ("[V] gok=1", the reference's last line) to log_fd.  bytestream, pac, pacterm.
 * bin/checkbwt is the program. */
int b3m_check_bwt(const char *bwt_fn, const char *text_fn, const char *inputtype, int log_fd, uint64_t *mismatches_out);

/* per-kernel-family timing of the last run (HIP events on the run's stream) */
typedef struct b3m_kernel_time { const char *name; double ms; uint64_t launches; uint64_t alg_bytes; } b3m_kernel_time;
int b3m_plan_kernel_times(const b3m_plan *p, b3m_kernel_time *out, int cap);

/* ---- stage-level entry points (device buffers), one per hot-path row of SURVEY.md 8a -------- */

/* row a4: symbol histogram of text[start, start+len) -> hist_out[256] (host) */
int b3m_histogram(const void *d_text, uint64_t n, int inputtype, uint64_t start, uint64_t len,
                  uint64_t *hist_out, void *stream);

/* row a7, BwtMergeBlockSortRequest: suffixes starting in [start, start+len) sorted as
 * rotations of the whole circular text.
 *   d_sa_out  : len u32 absolute positions in rank order (may be NULL)
 *   d_bwt_out : len bytes, bwt[r] = T[(sa[r]-1) mod n]   (may be NULL)
 *   d_gt_out  : (len+1 + 31)/32 u32 words, bit k = [S_{start+k} > S_{start}], k in [1,len]
 *               (may be NULL)
 *   p0_out    : rank of S_start inside the block */
int b3m_block_sort(const void *d_text, uint64_t n, int inputtype, uint64_t start, uint64_t len,
                   uint32_t *d_sa_out, uint8_t *d_bwt_out, uint32_t *d_gt_out,
                   uint64_t *p0_out, void *stream);

/* same sort, gt bits written on the device into an ABSOLUTE-position bit array d_gt_abs
 * ((n+1+31)/32+2 u32 words, OR-ed in; caller zeroes): bit x = [S_x > S_start], x in (start, start+len] */
int b3m_block_sort_abs(const void *d_text, uint64_t n, int inputtype, uint64_t start, uint64_t len,
                       uint32_t *d_sa_out, uint8_t *d_bwt_out, uint32_t *d_gt_abs,
                       uint64_t *p0_out, void *stream);

/* rank of the rotation starting at each positions[i] among a sorted block (d_sa, len):
 * #{ r : S_{sa[r]} < S_x }.  Used for the LF-chain anchors (the reference's z-ranks). */
int b3m_rank_of(const void *d_text, uint64_t n, int inputtype, const uint32_t *d_sa, uint64_t len,
                const uint64_t *positions, uint64_t count, uint64_t *ranks_out, void *stream);

/* row a8 + a9, gap computation of mergeBlocks: L = [a, a+m), R = [a+m, a+m+mr).
 *   d_bwtL/p0L : block BWT of L and the rank of S_a in it (that entry belongs to a-1)
 *   d_gtR      : bit k = [S_{a+m+k} > S_{a+m}], k in [1, mr]
 *   anchors    : host array of `num_chains` ranks: rank in L of S_{a+m+mr - c*chain_len}
 *   Output (sparse gap array, the in-HBM form of SparseGammaGap): d_sorted_ranks[q], q in
 *   [0,mr), = rank in L of the q-th smallest suffix of R, ascending.  Optionally the dense
 *   G[0..m] (u32 counts) and gt bits of the parent for positions of R:
 *   d_gtP bit (m+k) = [S_{a+m+k} > S_a], k in [0, mr]. */
int b3m_gap(const void *d_text, uint64_t n, int inputtype, uint64_t a, uint64_t m, uint64_t mr,
            const uint8_t *d_bwtL, uint64_t p0L, const uint32_t *d_gtR,
            const uint64_t *anchors, uint64_t num_chains, uint64_t chain_len,
            uint32_t *d_sorted_ranks, uint32_t *d_G_dense, uint32_t *d_gtP, void *stream);

/* Multi-GPU form of b3m_gap: one rank walks only the chains [chain_first, chain_first+num_chains)
 * of the merge's chain table (chain c covers R positions [a+m+mr-(c+1)*chain_len, a+m+mr-c*chain_len)
 * clipped to R).  gt arrays are indexed by ABSOLUTE text position ((n+1+31)/32+2 u32 words):
 *   d_gt_abs_in  bit x = [S_x > S_{a+m}] for x in (a+m, a+m+mr]
 *   d_gt_abs_out |= bit j = [S_j > S_a] for the walked positions j (caller zeroes it)
 * anchors[i] = rank in L of the suffix at the top end of chain chain_first+i.
 * d_sorted_ranks (capacity num_chains*chain_len) receives *count_out ascending ranks. */
int b3m_gap_chains(const void *d_text, uint64_t n, int inputtype, uint64_t a, uint64_t m, uint64_t mr,
                   const uint8_t *d_bwtL, uint64_t p0L, const uint32_t *d_gt_abs_in,
                   const uint64_t *anchors, uint64_t chain_first, uint64_t num_chains, uint64_t chain_len,
                   uint32_t *d_sorted_ranks, uint64_t *count_out, uint32_t *d_gt_abs_out, void *stream);

/* row a9 helper: ascending sort of u32 keys that are all < 2^nbits (the sparse gap array is a sorted
 * rank list; the multi-GPU driver re-sorts the rank buckets it receives).  Large lists take the MSD
 * partition + LDS histogram path of the merges, small ones LSD radix passes. */
int b3m_sort_u32(const uint32_t *d_keys, uint64_t count, int nbits, uint32_t *d_out, void *stream);
/* abs bit (start+k) |= rel bit k, k in [1,count]: block-relative gt bits of b3m_block_sort ->
 * the absolute-position bit array b3m_gap_chains works on. */
int b3m_bits_place(const uint32_t *d_rel, uint64_t count, uint64_t start, uint32_t *d_abs, void *stream);

/* row a10, interleave: out = for r in 0..m: G[r] symbols of BWT(R), then BWT(L)[r], with G
 * given as the sorted rank list.  p0P_out = new rank of L's p0 entry. */
int b3m_interleave(const uint32_t *d_sorted_ranks, const uint8_t *d_bwtL, uint64_t m,
                   const uint8_t *d_bwtR, uint64_t mr, uint8_t *d_out,
                   uint64_t p0L, uint64_t *p0P_out, void *stream);

/* ---- multi-GPU block merge: this rank's part, one persistent workspace -------------- */
/* One process per GPU.  The collectives between these calls -- broadcast of the packed BWT(L) slices, RCCL all-to-all(v) of the gap buckets
 * (rank lists cut at the borders of the group's rank slices) and of the BWT(R) ranges -- are issued by the host side
 * (bwtb3m_amd/distributed.py over torch.distributed); the reference has no counterpart (single process, OpenMP: SURVEY.md 5).
 * max_leaf / max_L / max_R: largest leaf this rank sorts, largest left / right side of a merge it takes part in.
 * pac, pacterm (2-bit rank lines; BWT(L) travels 4 symbols per byte) and bytestream (count rows every 64 symbols; BWT(L) travels as it is:
 * b3m_dist_pack2 copies, b3m_dist_walk takes 256 symbol starts instead of 4, b3m_dist_leaf_hist has the leaf's 256 symbol counts). */
typedef struct b3m_dist_plan b3m_dist_plan;
b3m_dist_plan *b3m_dist_create(uint64_t n, int inputtype, uint64_t max_leaf, uint64_t max_L, uint64_t max_R);
void     b3m_dist_destroy(b3m_dist_plan *p);
uint64_t b3m_dist_workspace_bytes(const b3m_dist_plan *p);
int b3m_dist_set_text(b3m_dist_plan *p, const void *d_text, void *stream);
int b3m_dist_leaf(b3m_dist_plan *p, uint64_t a, uint64_t m, uint8_t *d_bwt_out, uint32_t *d_gt_abs, uint64_t *p0_out, uint64_t *hist5_out, void *stream);
int b3m_dist_leaf_hist(const b3m_dist_plan *p, uint64_t *hist256_out);
int b3m_dist_rank_of(b3m_dist_plan *p, const uint64_t *positions, uint64_t count, uint64_t *ranks_out, void *stream);
/* the same, the ranks ADDED to a device array (int64: the chain anchors of a merge are sums over the leaves of its left side, all-reduced
 * between the GPUs without a detour through host memory) */
int b3m_dist_rank_of_dev(b3m_dist_plan *p, const uint64_t *positions, uint64_t count, int64_t *d_accum, void *stream);
int b3m_dist_pack2(b3m_dist_plan *p, const uint8_t *d_bwt, uint64_t m, uint8_t *d_packed, void *stream);   /* 4 symbols per byte */
int b3m_dist_build_dict(b3m_dist_plan *p, const uint8_t *const *d_parts, const uint64_t *part_syms, int nparts, uint64_t m, uint64_t p0L, void *stream);
int b3m_dist_walk(b3m_dist_plan *p, uint64_t a, uint64_t m, uint64_t mr, const uint64_t *symstart, const uint32_t *d_gt_abs_in, uint32_t *d_gt_abs_out,
                  const int64_t *d_anchors, uint64_t chain_first, uint64_t num_chains, uint64_t chain_len,
                  const uint64_t *bounds, int nbounds, uint32_t *d_sorted_out, uint64_t *cuts_out, uint64_t *count_out, void *stream);
/* bytes of BWT(L)[l0, l1) out of the dictionary; L's p0 entry reads the symbol in front of L (position a - 1, circular), taken from the text */
int b3m_dist_l_bytes(b3m_dist_plan *p, uint64_t a, uint64_t l0, uint64_t l1, uint8_t *d_out, void *stream);
/* bytes [q0, q1) of BWT(R) out of R's packed slices in rank order (d_parts[i]: part_syms[i] symbols in the form of b3m_dist_pack2): the members of a merge
 * exchange the packed slices of BOTH sides in ONE all-gather; the dictionary is built from L's, an owner takes its range of BWT(R) from R's here */
int b3m_dist_r_bytes(b3m_dist_plan *p, const uint8_t *const *d_parts, const uint64_t *part_syms, int nparts, uint64_t q0, uint64_t q1, uint8_t *d_out, void *stream);
/* gt plumbing of a merge on the device: bit `bit` of d_gt_abs = [d_anchor[0] > p0L] (the end bit of the parent node), and
 * bits [bit_begin, bit_end) of one absolute-position bit array copied into another (partial edge words merged under a mask) */
int b3m_dist_endbit(b3m_dist_plan *p, uint32_t *d_gt_abs, uint64_t bit, const int64_t *d_anchor, uint64_t p0L, void *stream);
int b3m_bits_copy(const uint32_t *d_src_abs, uint32_t *d_dst_abs, uint64_t bit_begin, uint64_t bit_end, void *stream);
int b3m_dist_sort_u32(b3m_dist_plan *p, const uint32_t *d_keys, uint64_t count, int nbits, uint32_t *d_out, void *stream);
int b3m_dist_interleave(b3m_dist_plan *p, const uint32_t *d_sr, const uint8_t *d_bwtL, uint64_t m, const uint8_t *d_bwtR, uint64_t mr, uint8_t *d_out, void *stream);

/* ---- output formats (host): rows a11/a12 ------------------------------------------------------ */

/* RL-encoded .bwt container (role of libmaus2::huffman::RLEncoderStd, usage
 * /root/reference/src/lcpbit.cpp:3677-3681).  Container byte layout is PARITY-UNPINNED, see
 * DESIGN.md. */
int b3m_rl_write(const char *fn, const uint8_t *symbols, uint64_t n, uint64_t runs_per_block);
/* the same file (byte for byte) from symbols that are still in HBM: the device finds the runs, counts every block's symbol / run-length
 * frequencies and packs both code streams at their final offsets; the host builds the code tables and headers.  Texts of any length: the BWT is taken in
 * segments of at most 2^31 symbols that end on block borders. */
int b3m_rl_write_device(const char *fn, const uint8_t *d_symbols, uint64_t n, uint64_t runs_per_block, void *stream);
/* decodeRun() loop of /root/reference/src/bwtb3mdecoderl.cpp:34: n_out = symbol count;
 * symbols_out may be NULL to query the length only */
int b3m_rl_read(const char *fn, uint8_t *symbols_out, uint64_t cap, uint64_t *n_out);
/* the same with the symbols as the file names them (inputtype=utf-8: Unicode code points): the (sym:int64, len) pairs of decodeRun() expanded */
int b3m_rl_read_symbols(const char *fn, int64_t *symbols_out, uint64_t cap, uint64_t *n_out);
int b3m_rl_length(const char *fn, uint64_t *n_out);           /* RLDecoder::getLength, checkbwt.cpp:57 */

#ifdef __cplusplus
}
#endif
#endif
