// formats.h -- host-side file formats around the hot path.
#pragma once
#include <stdint.h>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace b3m {

std::vector<uint8_t> read_file(const std::string &fn);
void write_file(const std::string &fn, const void *data, size_t bytes);
bool file_exists(const std::string &fn);

// ---- RL .bwt container (role of libmaus2::huffman::RLEncoderStd / RLDecoder) -------------------
// Layout follows what is known of the upstream container and is
// PARITY-UNPINNED: no libmaus2-written sample exists in /root/reference, so byte identity with
// upstream cannot be checked.  Bit order MSB-first.
//   u64 BE  n (symbols)
//   blocks, each byte-aligned:
//       elias(numruns) | 1 bit cntescape | symbol code table | run-length code table | align
//       numruns symbol codes | align | numruns run-length codes (escape: code + elias(len)) | align
//   index: per block (u64 BE file offset, u64 BE runs, u64 BE symbols), u64 BE numblocks
//   u64 BE  offset of the index            (last 8 bytes of the file)
struct RLRun { int64_t sym; uint64_t len; };
// symmap (may be null): the symbol value written for byte b is symmap[b] (monotone; utf-8 input: code points ranked into a byte alphabet)
void rl_encode_file(const std::string &fn, const uint8_t *symbols, uint64_t n, uint64_t runs_per_block, unsigned numthreads, const int64_t *symmap = nullptr);
// symbols of any width: ranks[i] < k names symmap[ranks[i]] (symmap null: the rank itself); same container, tables from maps
void rl_encode_file_wide(const std::string &fn, const uint32_t *ranks, uint64_t n, uint64_t runs_per_block, const int64_t *symmap);
// decodeRun() stream: calls sink(sym,len) per run; returns n
uint64_t rl_decode_file(const std::string &fn, void (*sink)(void *, int64_t, uint64_t), void *ctx);
uint64_t rl_length(const std::string &fn);
// decode into out[0..cap) with the blocks in parallel; returns n
uint64_t rl_decode_array(const std::string &fn, uint8_t *out, uint64_t cap, unsigned numthreads, bool *bad_symbol);
struct RLIndexEntry { uint64_t pos, runs, syms; };
// pieces shared with the device encoder (csrc/k_rlenc.hip): code tables + header bytes of one block from its frequencies, and the container
// around already encoded blocks (index entries: file position, runs, symbols of every block)
struct RLBlockTables { uint64_t scode[256]; uint8_t slen[256]; uint64_t ccode[257]; uint8_t clen[257]; };
void rl_block_header(const uint32_t *sf256, const uint32_t *cf257, uint64_t nr, const int64_t *symmap, std::vector<uint8_t> &header, RLBlockTables &t);
// the same file written from symbols that are still in HBM (csrc/k_rlenc.hip): runs, frequencies and both code streams on the device, tables
// and headers here.  false: not an input for it (empty, 2^32 symbols or more, a code above 32 bits): use rl_encode_file
bool rl_encode_device(const std::string &fn, const uint8_t *d_symbols, uint64_t n, uint64_t runs_per_block, unsigned numthreads, const int64_t *symmap, void *hip_stream);
// what stands in front of and behind the blocks of a container: n; the index, its size and its offset
// the same from a BWT in (pinned) host memory, brought to HBM one segment of at most seg_limit symbols at a time (host tier)
bool rl_encode_device_from_host(const std::string &fn, const uint8_t *h_symbols, uint64_t n, uint64_t runs_per_block, unsigned numthreads, const int64_t *symmap, void *hip_stream,
                                uint64_t seg_limit);
// one file from several devices: every device codes the blocks that BEGIN in its slice of the BWT (csrc/k_rlenc.hip, used by b3m_compute_bwt with num_gpus > 1)
struct RLPart {
	int dev = 0; uint64_t bytes = 0;
	std::vector<std::pair<void *, uint64_t>> images;     // device pointers (owned) and their sizes, in file order
	std::vector<RLIndexEntry> index;                      // the part's blocks; pos relative to the part's first byte
	RLPart() = default; ~RLPart();
	RLPart(const RLPart &) = delete; RLPart &operator=(const RLPart &) = delete;
};
void rl_slice_scan(const uint8_t *d_symbols, uint64_t len, void *hip_stream, std::vector<uint32_t> &tile_first, uint8_t *first_sym, uint8_t *last_sym);
uint64_t rl_head_position(const uint8_t *d_symbols, uint64_t len, const std::vector<uint32_t> &tile_first, uint64_t ordinal, void *hip_stream);
bool rl_encode_part(const uint8_t *d_symbols, uint64_t len, uint64_t runs_per_block, unsigned numthreads, const int64_t *symmap, void *hip_stream, RLPart &out);
void rl_container_ends(uint64_t n, uint64_t blocks_bytes, const std::vector<RLIndexEntry> &index, std::vector<uint8_t> &head, std::vector<uint8_t> &tail);
void rl_write_container(const std::string &fn, uint64_t n, const uint8_t *blocks, uint64_t blocks_bytes, const std::vector<RLIndexEntry> &index);
std::vector<RLIndexEntry> rl_index(const std::string &fn);

// ---- consumers of the container's seek index (role of RLDecoder::getBlockSymHistograms / SparseRank, /root/reference/src/bwtdecodeblock.cpp:134-365) ----
// X.bwt.sparserank: per block of the container, the counts of minsym..maxsym BEFORE the block, 8-byte big-endian each
void rl_block_sym_histograms(const std::string &bwtfn, const std::string &outfn, int64_t minsym, int64_t maxsym);
struct RLSparseRank {
	std::vector<uint8_t> data, sparse; std::vector<RLIndexEntry> index; std::vector<uint64_t> vstart, D; uint64_t ipos = 0, n = 0;
	int64_t minsym, maxsym; std::vector<RLRun> runs; uint64_t cached;
	RLSparseRank(const std::string &bwtfn, const std::string &histfn, int64_t minsym, int64_t maxsym);      // reads bwtfn + ".sparserank"
	std::pair<int64_t, uint64_t> inverse_select(uint64_t i);     // (symbol at rank i, its occurrences before i)
	std::pair<int64_t, uint64_t> lf(uint64_t i);                 // (symbol at rank i, rank of the suffix one position to the left)
};

// ---- siblings -------------------------------------------------------------------------------------
// X.hist: NumberMapSerialisation::serialiseMap<int64,uint64> (reader at
// /root/reference/src/bwtdecodeblock.cpp:150-173): u64 BE count, then (i64 BE key, u64 BE value)
void write_hist(const std::string &fn, const uint64_t hist256[256], const int64_t *symmap = nullptr);
void write_hist_map(const std::string &fn, const std::map<int64_t, uint64_t> &hist);
std::map<int64_t, uint64_t> read_hist(const std::string &fn);
// X.preisa: raw host-endian u64 pairs (rank,pos) (/root/reference/src/hwtPreIsaToIsa.cpp:64-77)
void write_preisa(const std::string &fn, const uint64_t *pairs, uint64_t count);
// X.sa / X.isa: u64 rate, u64 count, count x u64, host-endian (/root/reference/src/sasubsample.cpp:34-57)
void write_sampled(const std::string &fn, uint64_t rate, const std::vector<uint64_t> &values);
// X.meta: 8-byte big-endian numbers (NumberSerialisation); content is engine-specific
void write_meta(const std::string &fn, const std::vector<uint64_t> &numbers);

// inputtype=utf-8: the text is the file's sequence of Unicode code points (strict decoding: no overlong forms, surrogates,
// values above U+10FFFF or truncated sequences); throws with the byte offset of the first bad sequence
std::vector<uint32_t> utf8_decode(const std::vector<uint8_t> &file);

// clipOff(bwtname, ".bwt")  (/root/reference/src/checkbwt.cpp:31)
std::string clip_off(const std::string &s, const std::string &suffix);

}  // namespace b3m
