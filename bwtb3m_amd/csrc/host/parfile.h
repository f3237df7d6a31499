// parfile.h -- file <-> HBM transport of b3m_compute_bwt (role of the file -> memory and memory -> encoder hand-offs around
// BwtMergeSort::computeBwt, /root/reference/src/bwtb3m.cpp:63; the reference's numthreads= is what parallelises its I/O too, :48-50).
// An output file is written piecewise at its final offsets by a few host threads; pieces that still live in HBM come through pinned
// staging buffers, each thread on a HIP stream of its own (non-blocking streams: the copies overlap the kernels of other streams, e.g.
// the LF walks that make X.sa while X.bwt is being written).  The input file is read the same way, straight into device memory.
#pragma once
#include <stdint.h>
#include <string>

namespace b3m {

class ParFile {
public:
	explicit ParFile(const std::string &fn);             // created / truncated; throws std::runtime_error
	~ParFile();
	// the final size, if known: the file is extended and mapped, and pieces are copied into the mapping instead of written with pwrite (buffered writes to
	// ONE file take its inode lock one at a time, whatever the number of threads; page faults on a shared mapping do not)
	void reserve(uint64_t total_bytes);
	void put(uint64_t off, const void *p, uint64_t bytes);                                      // this thread
	void put_host(uint64_t off, const void *p, uint64_t bytes, unsigned threads);               // host memory, split over threads
	void put_device(uint64_t off, const void *d_p, uint64_t bytes, int device, unsigned threads); // device memory through pinned staging
	void close();                                        // throws on a failed close; the destructor closes silently
	const std::string &name() const { return fn_; }
private:
	std::string fn_; int fd_; uint8_t *map_ = nullptr; uint64_t map_bytes_ = 0;
	void store(uint64_t off, const void *p, uint64_t bytes);
};

uint64_t file_size(const std::string &fn);
// bytes [off, off + bytes) of the file -> host memory / device memory (pread by `threads` threads; device: through pinned staging)
void read_range(const std::string &fn, uint64_t off, void *dst, uint64_t bytes, unsigned threads);
void read_range_to_device(const std::string &fn, uint64_t off, void *d_dst, uint64_t bytes, int device, unsigned threads);
// the pinned staging buffers live until this is called (b3m_compute_bwt calls it before it returns)
void staging_release();

}  // namespace b3m
