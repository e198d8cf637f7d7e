// store.cpp — write-back cache of the per-view result files + the background workers of the driver.
//
// The reference's inter-pass API is the file system: every pass of every view re-reads its own previous
// results (depths.dmb, APD_normals.dmb, weak.bin, selected_views.bin, radius.bin: 625 MB at 6208x4128) and the
// depth maps of all its source views (9 x 100 MB) and writes five files (main.cpp:365-376, APD.cpp:1147-1195,
// 1428-1456).  Measured end to end at full resolution (profiles/r03_e2e_before.txt) that was 0.4 s of reads and
// 0.17 s of writes per view next to 1.3 s of kernels, on the critical path.  Here:
//   * PublishResult(file, mat) keeps `mat` in memory under the file's path and hands the write to a background
//     thread (temporary name + rename, so a reader of the folder never sees a torn file);
//   * LoadResult(file, ...) serves later passes from memory and falls back to the file (cache disabled, evicted,
//     another process wrote it);
//   * RunInBackground(job) runs a view's post-processing (visibility-mask clean-up, publishing) while the GPU is
//     already on the next view; a path announced with ExpectResult() makes LoadResult wait for its job.
// The files on disk end up byte-identical to the synchronous pipeline's; FlushResults() joins everything: before
// the fusion, at exit, and — with more than one rank — before every inter-pass barrier (host/main.cpp), because
// another rank may read this rank's depths.dmb in the next pass.
#include "APD.h"
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <set>
#include <thread>
#include <map>

namespace {
struct Store {
	std::mutex m;
	std::condition_variable cv;
	std::map<std::string, Mat> cache;        // path -> latest content
	std::map<std::string, uint64_t> stamp;   // path -> publish counter (for eviction order)
	std::set<std::string> writing;           // queued or in flight
	std::set<std::string> expected;          // announced by a background job, not yet published
	std::deque<std::pair<std::string, Mat>> write_queue;
	std::deque<std::function<void()>> jobs;
	size_t cached_bytes = 0, pending_bytes = 0;
	size_t limit_bytes = (size_t)32 << 30, pending_limit = (size_t)6 << 30;
	uint64_t clock = 0;
	int jobs_running = 0;
	bool stop = false, enabled = true;
	std::thread writer, worker;
	std::string write_error;
};
Store& g = *new Store;   // never destroyed: a print-and-exit on any thread must not run into joinable std::thread destructors

size_t bytes_of(const Mat& m) { return m.step * (size_t)m.rows; }

void evict_locked() {   // oldest entries that are not waiting to be written
	while (g.cached_bytes > g.limit_bytes) {
		std::string victim;
		uint64_t best = ~0ull;
		for (const auto& kv : g.cache)
			if (!g.writing.count(kv.first) && g.stamp[kv.first] < best) { best = g.stamp[kv.first]; victim = kv.first; }
		if (victim.empty()) return;
		g.cached_bytes -= bytes_of(g.cache[victim]);
		g.cache.erase(victim);
		g.stamp.erase(victim);
	}
}

void writer_loop() {
	std::unique_lock<std::mutex> lk(g.m);
	for (;;) {
		g.cv.wait(lk, [] { return g.stop || !g.write_queue.empty(); });
		if (g.write_queue.empty()) { if (g.stop) return; continue; }
		auto item = g.write_queue.front();
		g.write_queue.pop_front();
		lk.unlock();
		path tmp = item.first;
		tmp += ".part";
		std::error_code ec;
		bool ok = WriteBinMat(tmp, item.second);
		if (ok) { std::filesystem::rename(tmp, item.first, ec); ok = !ec; }
		lk.lock();
		if (!ok && g.write_error.empty()) g.write_error = "cannot write " + item.first;
		g.pending_bytes -= bytes_of(item.second);
		// a newer version of the same path may be queued behind this one: only then is it still "writing"
		bool again = false;
		for (const auto& q : g.write_queue) again = again || q.first == item.first;
		if (!again) g.writing.erase(item.first);
		evict_locked();
		g.cv.notify_all();
	}
}
void worker_loop() {
	std::unique_lock<std::mutex> lk(g.m);
	for (;;) {
		g.cv.wait(lk, [] { return g.stop || !g.jobs.empty(); });
		if (g.jobs.empty()) { if (g.stop) return; continue; }
		auto job = std::move(g.jobs.front());
		g.jobs.pop_front();
		g.jobs_running++;
		lk.unlock();
		job();
		lk.lock();
		g.jobs_running--;
		g.cv.notify_all();
	}
}
void start_threads_locked() {
	if (!g.writer.joinable()) g.writer = std::thread(writer_loop);
	if (!g.worker.joinable()) g.worker = std::thread(worker_loop);
}
}  // namespace

void SetResultCache(bool enabled, size_t limit_bytes) {
	std::lock_guard<std::mutex> lk(g.m);
	g.enabled = enabled;
	if (limit_bytes) g.limit_bytes = limit_bytes;
}

void PublishResult(const path& file, const Mat& m) {
	const std::string key = file.string();
	std::unique_lock<std::mutex> lk(g.m);
	if (!g.enabled) {
		lk.unlock();
		path tmp = file;
		tmp += ".part";
		if (!WriteBinMat(tmp, m)) DvpFatal("cannot write " + tmp.string());
		std::error_code ec;
		std::filesystem::rename(tmp, file, ec);   // (the throwing overload would end in std::terminate, past the agreed multi-rank abort)
		if (ec) DvpFatal("cannot rename " + tmp.string() + " to " + file.string() + ": " + ec.message());
		lk.lock();
		g.expected.erase(key);
		g.cv.notify_all();
		return;
	}
	start_threads_locked();
	if (!g.write_error.empty()) { const std::string e = g.write_error; lk.unlock(); DvpFatal(e); }
	g.cv.wait(lk, [&] { return g.pending_bytes + bytes_of(m) <= g.pending_limit || g.write_queue.empty(); });   // bounded backlog
	auto it = g.cache.find(key);
	if (it != g.cache.end()) g.cached_bytes -= bytes_of(it->second);
	g.cache[key] = m;
	g.stamp[key] = ++g.clock;
	g.cached_bytes += bytes_of(m);
	g.writing.insert(key);
	g.write_queue.emplace_back(key, m);
	g.pending_bytes += bytes_of(m);
	g.expected.erase(key);
	evict_locked();
	g.cv.notify_all();
}

void ExpectResult(const path& file) {
	std::lock_guard<std::mutex> lk(g.m);
	g.expected.insert(file.string());
}

bool LoadResult(const path& file, Mat& m, bool will_modify) {
	const std::string key = file.string();
	{
		std::unique_lock<std::mutex> lk(g.m);
		g.cv.wait(lk, [&] { return !g.expected.count(key); });   // its background job has not published it yet
		auto it = g.cache.find(key);
		if (it != g.cache.end()) {
			if (!will_modify) { m = it->second; return true; }
			// the caller is going to write into the buffer (state maps the engine downloads into): it gets its own copy
			// unless nobody else can still see this one
			if (g.writing.count(key)) { const Mat shared = it->second; lk.unlock(); m = shared.clone(); return true; }
			m = it->second;
			g.cached_bytes -= bytes_of(it->second);
			g.cache.erase(it);
			g.stamp.erase(key);
			return true;
		}
		g.cv.wait(lk, [&] { return !g.writing.count(key); });   // evicted while queued cannot happen; defensive
		// a failed background write means the file on disk is stale or missing: never compute from it
		if (!g.write_error.empty()) { const std::string e = g.write_error; lk.unlock(); DvpFatal(e + " (noticed by LoadResult of " + key + ")"); }
	}
	if (!std::filesystem::exists(file)) return false;
	return ReadBinMat(file, m);
}

bool ResultExists(const path& file) {
	const std::string key = file.string();
	{
		std::lock_guard<std::mutex> lk(g.m);
		if (g.cache.count(key) || g.expected.count(key) || g.writing.count(key)) return true;
	}
	return std::filesystem::exists(file);
}

void RunInBackground(std::function<void()> job) {
	std::unique_lock<std::mutex> lk(g.m);
	if (!g.enabled) { lk.unlock(); job(); return; }
	start_threads_locked();
	g.cv.wait(lk, [] { return g.jobs.size() < 2; });   // at most two views behind the GPU
	g.jobs.push_back(std::move(job));
	g.cv.notify_all();
}

void WaitBackgroundJobs() {
	std::unique_lock<std::mutex> lk(g.m);
	g.cv.wait(lk, [] { return g.jobs.empty() && g.jobs_running == 0; });
}

void FlushResults(bool drop_cache) {
	std::unique_lock<std::mutex> lk(g.m);
	g.cv.wait(lk, [] { return g.jobs.empty() && g.jobs_running == 0; });
	g.cv.wait(lk, [] { return g.write_queue.empty() && g.writing.empty(); });
	if (drop_cache) { g.cache.clear(); g.stamp.clear(); g.cached_bytes = 0; }
	if (!g.write_error.empty()) { const std::string e = g.write_error; lk.unlock(); DvpFatal(e); }
}

void ShutdownResultStore() {
	// (the cache is not emptied: handing tens of GB of maps back block by block took 0.4 s of a 32 s job, and the process is about to end)
	FlushResults(false);
	{
		std::lock_guard<std::mutex> lk(g.m);
		g.stop = true;
		g.cv.notify_all();
	}
	if (g.writer.joinable()) g.writer.join();
	if (g.worker.joinable()) g.worker.join();
	g.stop = false;
}
