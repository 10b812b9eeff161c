// CPU-only driver for tests/test_host_units.py: the header-only host pieces of the product that need no GPU.
//   bam <out.bam>        a small BAM file written with csrc/bam_writer.h (header, dictionary, records in several BGZF chunks, EOF)
//   pool                 ngm::ThreadPool::parallel_for: sums, concurrent callers, nested use
#include <atomic>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../nextgenmap_amd/csrc/bam_writer.h"
#include "../../nextgenmap_amd/csrc/thread_pool.h"

static int write_bam(const char *path) {
	std::string raw, file;
	ngm::bam::put_header(raw, "@HD\tVN:1.0\tSO:unsorted\n@PG\tID:ngm\n", {"chr1", "chrTwo"}, {1000000, 54321});
	if (!ngm::bam::bgzf_compress(raw.data(), raw.size(), file)) return 2;
	// three chunks of records, each compressed on its own (as the formatter threads do): they must concatenate into one file
	for (int chunk = 0; chunk < 3; ++chunk) {
		std::string recs;
		for (int i = 0; i < 700; ++i) {
			const int id = chunk * 700 + i;
			char name[32];
			const int nl = snprintf(name, sizeof(name), "read%05d", id);
			std::string seq(100 + id % 51, 'A');
			for (size_t t = 0; t < seq.size(); ++t) seq[t] = "ACGTN"[(id + 3 * t) % 5];
			std::string qual(seq.size(), 'I');
			for (size_t t = 0; t < qual.size(); ++t) qual[t] = (char) (33 + (id + t) % 40);
			char cigar[64];
			snprintf(cigar, sizeof(cigar), "%dS%dM2I%dM1D%dM", id % 7 + 1, 20, 30, (int) seq.size() - 53 - (id % 7 + 1));
			ngm::bam::Tags tg;
			tg.add_int("AS", 1234 - id); tg.add_int("NM", id % 9); tg.add_float("XI", 0.9876f); tg.add_string("MD", "50A49", 5);
			const bool unmapped = id % 97 == 0;
			ngm::bam::put_record(recs, name, (size_t) nl, unmapped ? 4u : (id % 2 ? 16u : 0u), unmapped ? -1 : id % 2, unmapped ? -1 : 1000 + 37 * id, unmapped ? 0 : 60,
					unmapped ? nullptr : cigar, seq.data(), seq.size(), id % 5 == 0 ? nullptr : qual.data(), -1, -1, 0, tg);
		}
		if (!ngm::bam::bgzf_compress(recs.data(), recs.size(), file)) return 3;
	}
	ngm::bam::bgzf_eof(file);
	FILE *f = fopen(path, "wb");
	if (!f) return 4;
	fwrite(file.data(), 1, file.size(), f);
	fclose(f);
	printf("min_bin %u %u %u %u %u\n", ngm::bam::min_bin(0, 1), ngm::bam::min_bin(16383, 16385), ngm::bam::min_bin(1 << 20, (1 << 20) + 150), ngm::bam::min_bin(-1, -1),
			ngm::bam::min_bin(100000000, 100000200));
	return 0;
}

static int test_pool() {
	ngm::ThreadPool &pool = ngm::ThreadPool::instance();
	// every index exactly once, any grain
	for (int n : {0, 1, 7, 1000, 100003}) for (int grain : {1, 64, 4096}) {
		std::vector<std::atomic<int>> seen(n);
		for (auto &s : seen) s = 0;
		pool.parallel_for(n, [&](int lo, int hi) { for (int i = lo; i < hi; ++i) seen[i].fetch_add(1); }, grain);
		for (int i = 0; i < n; ++i) if (seen[i] != 1) { printf("pool: index %d of %d seen %d times (grain %d)\n", i, n, (int) seen[i], grain); return 1; }
	}
	// several callers at once (the mapper instances and the CLI stages share the pool), one of them nesting
	std::atomic<long long> total{0};
	std::vector<std::thread> callers;
	for (int c = 0; c < 6; ++c) callers.emplace_back([&, c] {
		for (int rep = 0; rep < 20; ++rep) {
			long long local = 0;
			std::atomic<long long> sum{0};
			pool.parallel_for(50000, [&](int lo, int hi) {
				long long s = 0;
				for (int i = lo; i < hi; ++i) s += i;
				if (c == 0 && lo == 0) pool.parallel_for(1000, [&](int a, int b) { sum += (b - a); }, 100);  // nested
				sum += s;
			}, 512);
			local = sum;
			total += local - (c == 0 ? 1000 : 0);
		}
	});
	for (auto &t : callers) t.join();
	const long long expect = 6LL * 20 * (50000LL * 49999 / 2);
	printf("pool threads %d total %lld expect %lld\n", pool.size(), (long long) total, expect);
	return total == expect ? 0 : 1;
}

int main(int argc, char **argv) {
	if (argc >= 3 && !strcmp(argv[1], "bam")) return write_bam(argv[2]);
	if (argc >= 2 && !strcmp(argv[1], "pool")) return test_pool();
	return 64;
}
