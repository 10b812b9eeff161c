// write_calib.cpp -- how fast can ONE output file be filled on this host?  (The SAM text of `ngm-hip` is 422 bytes per read:
// 15 M reads/s are 6.3 GB/s into one file.)  Variants: pwrite from 1..N threads into disjoint ranges (buffered writes take the
// inode lock), chunk sizes, a shared mapping filled by N threads with and without MADV_POPULATE_WRITE ahead of the copies,
// and N separate files as the no-contention ceiling.
//   g++ -O2 -pthread write_calib.cpp -o write_calib && ./write_calib <dir> [GiB = 4]
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statfs.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
	const std::string dir = argc > 1 ? argv[1] : "/tmp";
	const double gib = argc > 2 ? atof(argv[2]) : 4.0;
	const size_t total = (size_t) (gib * (1 << 30)) & ~((size_t) (1 << 21) - 1);
	struct statfs sf;
	if (statfs(dir.c_str(), &sf) == 0) printf("dir %s  f_type 0x%lx  free %.1f GiB\n", dir.c_str(), (unsigned long) sf.f_type, (double) sf.f_bavail * sf.f_bsize / (1 << 30));
	std::vector<char> src(64 << 20);
	for (size_t i = 0; i < src.size(); ++i) src[i] = (char) ('A' + (i * 7) % 23);
	const std::string fn = dir + "/write_calib.bin";
	auto report = [&](const char *what, int threads, size_t chunk, double dt) {
		printf("%-34s threads %2d  chunk %8zu KiB  %.3f s  %.2f GB/s\n", what, threads, chunk >> 10, dt, total / dt / 1e9);
		fflush(stdout);
	};
	// --- pwrite, T threads, disjoint interleaved chunks of one file
	for (int T : {1, 2, 4, 8, 16}) for (size_t chunk : {(size_t) 256 << 10, (size_t) 4 << 20, (size_t) 32 << 20}) {
		if (T > 1 && chunk == ((size_t) 256 << 10) && T != 4) continue;
		unlink(fn.c_str());
		const int fd = open(fn.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
		if (fd < 0) { perror("open"); return 1; }
		const size_t n_chunks = total / chunk;
		std::atomic<size_t> next{0};
		const double t0 = now();
		std::vector<std::thread> th;
		for (int t = 0; t < T; ++t) th.emplace_back([&] {
			for (;;) {
				const size_t c = next++;
				if (c >= n_chunks) break;
				size_t off = c * chunk, left = chunk;
				while (left) { const ssize_t w = pwrite(fd, src.data() + (off % (src.size() - chunk + 1)) % 4096, left, (off_t) off); if (w <= 0) { perror("pwrite"); exit(1); } off += w; left -= w; }
			}
		});
		for (auto &x : th) x.join();
		const double dt = now() - t0;
		close(fd);
		report("pwrite, one file", T, chunk, dt);
	}
	// --- N separate files (ceiling without inode-lock contention)
	for (int T : {4, 16}) {
		const size_t chunk = (size_t) 4 << 20;
		std::vector<int> fds(T);
		for (int t = 0; t < T; ++t) { const std::string f2 = fn + "." + std::to_string(t); unlink(f2.c_str()); fds[t] = open(f2.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); }
		const double t0 = now();
		std::vector<std::thread> th;
		for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
			for (size_t off = 0; off < total / T; off += chunk) { if (pwrite(fds[t], src.data(), chunk, (off_t) off) != (ssize_t) chunk) { perror("pwrite"); exit(1); } }
		});
		for (auto &x : th) x.join();
		const double dt = now() - t0;
		for (int t = 0; t < T; ++t) { close(fds[t]); unlink((fn + "." + std::to_string(t)).c_str()); }
		report("pwrite, one file per thread", T, chunk, dt);
	}
	// --- shared mapping filled by T threads; populate: 0 none, 1 MADV_POPULATE_WRITE by the copying thread per chunk, 2 by T/2 extra threads running ahead
	for (int populate : {0, 1}) for (int T : {4, 16, 32}) {
		const size_t chunk = (size_t) 4 << 20;
		unlink(fn.c_str());
		const int fd = open(fn.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
		if (ftruncate(fd, (off_t) total) != 0) { perror("ftruncate"); return 1; }
		const double t0 = now();
		char *m = (char *) mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		if (m == MAP_FAILED) { perror("mmap"); return 1; }
		const size_t n_chunks = total / chunk;
		std::atomic<size_t> next{0};
		std::vector<std::thread> th;
		for (int t = 0; t < T; ++t) th.emplace_back([&] {
			for (;;) {
				const size_t c = next++;
				if (c >= n_chunks) break;
				if (populate == 1 && madvise(m + c * chunk, chunk, MADV_POPULATE_WRITE) != 0) { static std::atomic<bool> once{false}; if (!once.exchange(true)) perror("madvise(MADV_POPULATE_WRITE)"); }
				memcpy(m + c * chunk, src.data(), chunk);
			}
		});
		for (auto &x : th) x.join();
		munmap(m, total);
		const double dt = now() - t0;
		close(fd);
		report(populate ? "mmap + MADV_POPULATE_WRITE + memcpy" : "mmap + memcpy (page faults)", T, chunk, dt);
	}
	unlink(fn.c_str());
	return 0;
}
