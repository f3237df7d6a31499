// parfile.cpp -- see parfile.h
#include "parfile.h"
#include "../switches.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <stdexcept>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace b3m {

namespace {
constexpr uint64_t CHUNK = 8ull << 20;      // one piece: 8 MB (a page-cache write of it takes milliseconds, its PCIe copy ~0.15 ms)

#define PF_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// pinned staging buffers, handed out to the worker threads of one transfer and taken back afterwards
struct Staging {
	std::mutex mu;
	std::vector<void *> idle;
	void *take()
	{
		{
			std::lock_guard<std::mutex> g(mu);
			if (!idle.empty()) { void *p = idle.back(); idle.pop_back(); return p; }
		}
		void *p = nullptr;
		PF_HIP(hipHostMalloc(&p, CHUNK, hipHostMallocDefault));
		return p;
	}
	void give(void *p) { std::lock_guard<std::mutex> g(mu); idle.push_back(p); }
	void release() { std::lock_guard<std::mutex> g(mu); for (void *p : idle) (void)hipHostFree(p); idle.clear(); }
} g_staging;

void pwrite_all(int fd, const void *p, uint64_t bytes, uint64_t off, const std::string &fn)
{
	const uint8_t *b = (const uint8_t *)p;
	while (bytes) {
		ssize_t w = pwrite(fd, b, bytes, (off_t)off);
		if (w < 0) { if (errno == EINTR) continue; throw std::runtime_error("write to " + fn + " failed: " + strerror(errno)); }
		b += w; off += (uint64_t)w; bytes -= (uint64_t)w;
	}
}
void pread_all(int fd, void *p, uint64_t bytes, uint64_t off, const std::string &fn)
{
	uint8_t *b = (uint8_t *)p;
	while (bytes) {
		ssize_t r = pread(fd, b, bytes, (off_t)off);
		if (r < 0) { if (errno == EINTR) continue; throw std::runtime_error("read of " + fn + " failed: " + strerror(errno)); }
		if (r == 0) throw std::runtime_error("short read on " + fn);
		b += r; off += (uint64_t)r; bytes -= (uint64_t)r;
	}
}

// runs work(chunk index) for every chunk on up to `threads` threads; the first exception is rethrown on the caller's thread
template <class F> void for_chunks(uint64_t bytes, unsigned threads, F work)
{
	const uint64_t nchunks = (bytes + CHUNK - 1) / CHUNK;
	if (nchunks == 0) return;
	const unsigned T = (unsigned)std::min<uint64_t>(std::max(1u, threads), nchunks);
	std::atomic<uint64_t> next(0);
	std::mutex emu; std::string err;
	auto run = [&](unsigned t) {
		try { work(t, next, nchunks); }
		catch (const std::exception &e) { std::lock_guard<std::mutex> g(emu); if (err.empty()) err = e.what(); next = nchunks; }
	};
	std::vector<std::thread> th;
	for (unsigned t = 1; t < T; ++t) th.emplace_back(run, t);
	run(0);
	for (auto &x : th) x.join();
	if (!err.empty()) throw std::runtime_error(err);
}
}  // namespace

ParFile::ParFile(const std::string &fn) : fn_(fn), fd_(-1)
{
	fd_ = open(fn.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);      // (read-write: a shared writable mapping needs it)
	if (fd_ < 0) throw std::runtime_error("cannot open " + fn + " for writing");
}
ParFile::~ParFile() { if (map_) (void)munmap(map_, map_bytes_); if (fd_ >= 0) (void)::close(fd_); }
void ParFile::close()
{
	if (map_) { (void)munmap(map_, map_bytes_); map_ = nullptr; }
	if (fd_ >= 0) { int r = ::close(fd_); fd_ = -1; if (r) throw std::runtime_error("error closing " + fn_); }
}
void ParFile::reserve(uint64_t total_bytes)
{
	// measured at C3 (tools/e2e_c3.py --mmap-ab, 0.94 GB .bwt + 0.75 GB .sa): the mapping is SLOWER than pwrite from 8 threads (0.72-0.96 s against
	// 0.38-0.48 s for the .bwt: a page fault per 4 KB costs more than the inode lock): off unless B3M_FILE_MMAP=1 asks for it
	if (b3m::sw().FILE_MMAP.get(0) != 1 || total_bytes < (64ull << 20) || map_) return;
	if (ftruncate(fd_, (off_t)total_bytes) != 0) return;
	void *m = mmap(nullptr, total_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0);
	if (m == MAP_FAILED) return;                                                  // (a file system without shared mappings: pwrite it is)
	map_ = (uint8_t *)m; map_bytes_ = total_bytes;
}
void ParFile::store(uint64_t off, const void *p, uint64_t bytes)
{
	if (map_ && off + bytes <= map_bytes_) memcpy(map_ + off, p, bytes);
	else pwrite_all(fd_, p, bytes, off, fn_);
}
void ParFile::put(uint64_t off, const void *p, uint64_t bytes) { store(off, p, bytes); }

void ParFile::put_host(uint64_t off, const void *p, uint64_t bytes, unsigned threads)
{
	const uint8_t *b = (const uint8_t *)p;
	for_chunks(bytes, threads, [&](unsigned, std::atomic<uint64_t> &next, uint64_t nchunks) {
		for (;;) {
			const uint64_t c = next.fetch_add(1);
			if (c >= nchunks) return;
			const uint64_t o = c * CHUNK, len = std::min(CHUNK, bytes - o);
			store(off + o, b + o, len);
		}
	});
}

void ParFile::put_device(uint64_t off, const void *d_p, uint64_t bytes, int device, unsigned threads)
{
	const uint8_t *b = (const uint8_t *)d_p;
	for_chunks(bytes, threads, [&](unsigned, std::atomic<uint64_t> &next, uint64_t nchunks) {
		PF_HIP(hipSetDevice(device));
		hipStream_t s = nullptr;
		PF_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
		void *stage = nullptr;
		try {
			stage = g_staging.take();
			for (;;) {
				const uint64_t c = next.fetch_add(1);
				if (c >= nchunks) break;
				const uint64_t o = c * CHUNK, len = std::min(CHUNK, bytes - o);
				PF_HIP(hipMemcpyAsync(stage, b + o, len, hipMemcpyDeviceToHost, s));
				PF_HIP(hipStreamSynchronize(s));
				store(off + o, stage, len);
			}
		} catch (...) { if (stage) g_staging.give(stage); (void)hipStreamDestroy(s); throw; }
		g_staging.give(stage);
		(void)hipStreamDestroy(s);
	});
}

uint64_t file_size(const std::string &fn)
{
	struct stat st;
	if (stat(fn.c_str(), &st) != 0) throw std::runtime_error("cannot open " + fn + " for reading");
	return (uint64_t)st.st_size;
}

void read_range(const std::string &fn, uint64_t off, void *dst, uint64_t bytes, unsigned threads)
{
	int fd = open(fn.c_str(), O_RDONLY);
	if (fd < 0) throw std::runtime_error("cannot open " + fn + " for reading");
	uint8_t *b = (uint8_t *)dst;
	try {
		for_chunks(bytes, threads, [&](unsigned, std::atomic<uint64_t> &next, uint64_t nchunks) {
			for (;;) {
				const uint64_t c = next.fetch_add(1);
				if (c >= nchunks) return;
				const uint64_t o = c * CHUNK, len = std::min(CHUNK, bytes - o);
				pread_all(fd, b + o, len, off + o, fn);
			}
		});
	} catch (...) { (void)::close(fd); throw; }
	(void)::close(fd);
}

void read_range_to_device(const std::string &fn, uint64_t off, void *d_dst, uint64_t bytes, int device, unsigned threads)
{
	int fd = open(fn.c_str(), O_RDONLY);
	if (fd < 0) throw std::runtime_error("cannot open " + fn + " for reading");
	uint8_t *b = (uint8_t *)d_dst;
	try {
		for_chunks(bytes, threads, [&](unsigned, std::atomic<uint64_t> &next, uint64_t nchunks) {
			PF_HIP(hipSetDevice(device));
			hipStream_t s = nullptr;
			PF_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
			void *stage = nullptr;
			try {
				stage = g_staging.take();
				for (;;) {
					const uint64_t c = next.fetch_add(1);
					if (c >= nchunks) break;
					const uint64_t o = c * CHUNK, len = std::min(CHUNK, bytes - o);
					pread_all(fd, stage, len, off + o, fn);
					PF_HIP(hipMemcpyAsync(b + o, stage, len, hipMemcpyHostToDevice, s));
					PF_HIP(hipStreamSynchronize(s));
				}
			} catch (...) { if (stage) g_staging.give(stage); (void)hipStreamDestroy(s); throw; }
			g_staging.give(stage);
			(void)hipStreamDestroy(s);
		});
	} catch (...) { (void)::close(fd); throw; }
	(void)::close(fd);
}

void staging_release() { g_staging.release(); }

}  // namespace b3m
