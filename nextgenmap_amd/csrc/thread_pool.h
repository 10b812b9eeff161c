// thread_pool.h -- one persistent pool of host threads per process for the per-read host stages (pair selection,
// CIGAR / position formatting in mapper.cpp; FASTQ parsing and SAM formatting in ngm_cli.cpp).
// NextGenMap runs these stages on its CS threads (src/NGM.cpp:232-279 hands out batches under a mutex); here a batch is
// cut into chunks that idle pool threads pick up, several callers (mapper instances) may have loops in flight at once, and
// the calling thread works on its own loop too, so a loop never waits for a free pool thread.
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace ngm {

class ThreadPool {
public:
	// threads: 0 = from the machine: hardware threads divided by the ranks on this node (torchrun sets LOCAL_WORLD_SIZE),
	// NGM_HIP_HOST_THREADS overrides
	static ThreadPool &instance() {
		static ThreadPool pool(default_threads());
		return pool;
	}
	static int default_threads() {
		if (const char *e = getenv("NGM_HIP_HOST_THREADS")) return std::max(1, atoi(e));
		const char *e = getenv("LOCAL_WORLD_SIZE");
		if (!e) e = getenv("WORLD_SIZE");
		const int ranks = std::max(1, e ? atoi(e) : 1);
		// (not capped by the quota: the host stages of the mapping path are short bursts between GPU stages -- far below the quota on
		// average -- and a burst on 64 threads ends sooner than on 16; loops that keep every thread busy for long pass cpu_quota() as
		// their limit instead: BAM records + deflate)
		return std::max(1, std::min(64, (int) std::thread::hardware_concurrency() / ranks));
	}
	// CPUs this process may use at once: the cgroup's CFS quota (containers: /sys/fs/cgroup/cpu.max "quota period") when there is one.
	// More threads than that, busy for long, do not just queue -- the group is throttled for the rest of every period, all threads at
	// once, and 64 busy threads under a 16-CPU quota get LESS done than 16 (profiles/r04_cpu_quota_probe.txt: 12.2 x against 15.6 x one thread)
	static int cpu_quota() {
		int cpus = 1 << 20;
		if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
			char q[64] = {0};
			long long period = 0;
			if (fscanf(f, "%63s %lld", q, &period) == 2 && q[0] != 'm' && period > 0) cpus = (int) std::max(1LL, (atoll(q) + period - 1) / period);
			fclose(f);
		} else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
			long long quota = -1, period = 100000;
			if (fscanf(g, "%lld", &quota) != 1) quota = -1;
			fclose(g);
			if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
			if (quota > 0 && period > 0) cpus = (int) std::max(1LL, (quota + period - 1) / period);
		}
		return cpus;
	}
	int size() const { return (int) workers_.size() + 1; }

	// f(lo, hi) over [0, n) in chunks of at least min_grain items; returns when all of it is done
	// max_threads > 0: at most that many threads work on this loop at a time (sustained CPU-bound loops under a CPU quota)
	template <typename F>
	void parallel_for(int n, F &&f, int min_grain = 1024, int max_threads = 0) {
		if (n <= 0) return;
		int chunks = std::min(size() * 4, std::max(1, n / std::max(1, min_grain)));
		if (chunks <= 1 || workers_.empty()) { f(0, n); return; }
		auto job = std::make_shared<Job>();
		job->fn = [&f](int lo, int hi) { f(lo, hi); };
		job->n = n; job->chunks = chunks; job->next.store(0); job->done.store(0);
		job->limit = max_threads; job->active.store(1);   // (the caller)
		{
			std::lock_guard<std::mutex> lk(mu_);
			jobs_.push_back(job);
		}
		cv_.notify_all();
		run_chunks(*job);  // the caller helps
		{
			std::unique_lock<std::mutex> lk(job->mu);
			job->cv.wait(lk, [&] { return job->done.load() == job->chunks; });
		}
		std::lock_guard<std::mutex> lk(mu_);
		jobs_.erase(std::remove(jobs_.begin(), jobs_.end(), job), jobs_.end());
	}

	~ThreadPool() {
		{
			std::lock_guard<std::mutex> lk(mu_);
			stop_ = true;
		}
		cv_.notify_all();
		for (auto &t : workers_) t.join();
	}

private:
	struct Job {
		std::function<void(int, int)> fn;
		int n = 0, chunks = 0;
		std::atomic<int> next{0}, done{0}, active{0};
		int limit = 0;
		std::mutex mu;
		std::condition_variable cv;
	};
	explicit ThreadPool(int threads) {
		for (int t = 1; t < threads; ++t) workers_.emplace_back([this] { worker(); });
	}
	static void run_chunks(Job &j) {
		for (;;) {
			const int c = j.next.fetch_add(1);
			if (c >= j.chunks) return;
			const int lo = (int) ((long long) j.n * c / j.chunks), hi = (int) ((long long) j.n * (c + 1) / j.chunks);
			j.fn(lo, hi);
			if (j.done.fetch_add(1) + 1 == j.chunks) {
				std::lock_guard<std::mutex> lk(j.mu);
				j.cv.notify_all();
			}
		}
	}
	void worker() {
		for (;;) {
			std::shared_ptr<Job> job;
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [&] {
					if (stop_) return true;
					for (auto &j : jobs_) if (j->next.load() < j->chunks && (j->limit <= 0 || j->active.load() < j->limit)) return true;
					return false;
				});
				if (stop_) return;
				for (auto &j : jobs_) if (j->next.load() < j->chunks && (j->limit <= 0 || j->active.load() < j->limit)) { job = j; j->active.fetch_add(1); break; }
			}
			if (job) { run_chunks(*job); job->active.fetch_sub(1); }
		}
	}
	std::vector<std::thread> workers_;
	std::vector<std::shared_ptr<Job>> jobs_;
	std::mutex mu_;
	std::condition_variable cv_;
	bool stop_ = false;
};

}  // namespace ngm
