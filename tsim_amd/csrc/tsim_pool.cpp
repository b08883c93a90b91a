// tsim_pool.cpp - the process-wide worker pool behind tsim_parallel_for (host only, no HIP).
//
// tsim_program_finalize packs a program's levels and, inside a level, its graphs in parallel.  Round 5 created the threads
// per call - one per level, up to eight more per level for the graphs: 88 std::thread constructions for the cultivation
// shape, ~3 of the 6.5 ms its packing took on a 256-CPU host, and unbounded in the number of components.
// Now: up to 32 workers, created once per process on first use (fewer if a thread cannot be created, none at all is fine);
// a parallel_for publishes a job - an index counter, a bound, a function - and its CALLER works on it like any worker,
// so a nested parallel_for (a level's graphs, from a worker that packs that level) always makes progress on its own job
// and never waits for a free thread.
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct Job {
  std::function<void(size_t)> fn;
  size_t n = 0;
  int max_workers = 0;             // helpers besides the caller
  std::atomic<size_t> next{0};
  std::atomic<size_t> done{0};
  std::atomic<int> helpers{0};
};

struct Pool {
  std::mutex m;
  std::condition_variable cv;
  std::vector<std::shared_ptr<Job>> jobs;  // jobs that may still have indices to hand out
  std::vector<std::thread> threads;
  bool started = false;

  static void run(Job &j) {
    for (;;) {
      const size_t i = j.next.fetch_add(1, std::memory_order_relaxed);
      if (i >= j.n) return;
      j.fn(i);
      j.done.fetch_add(1, std::memory_order_release);
    }
  }
  void worker() {
    for (;;) {
      std::shared_ptr<Job> j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] {
          for (auto &q : jobs)
            if (q->next.load(std::memory_order_relaxed) < q->n && q->helpers.load(std::memory_order_relaxed) < q->max_workers) return true;
          return false;
        });
        for (auto &q : jobs)
          if (q->next.load(std::memory_order_relaxed) < q->n && q->helpers.load(std::memory_order_relaxed) < q->max_workers) {
            j = q;
            q->helpers.fetch_add(1, std::memory_order_relaxed);
            break;
          }
      }
      if (j) run(*j);
    }
  }
  void start() {
    std::lock_guard<std::mutex> lk(m);
    if (started) return;
    started = true;
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned want = hw > 2 ? (hw - 1 < 32u ? hw - 1 : 32u) : 0u;
    try {
      for (unsigned t = 0; t < want; ++t) threads.emplace_back([this] { worker(); });
    } catch (...) {  // fewer workers: the callers do the rest themselves
    }
    for (auto &t : threads) t.detach();  // (never joined: the process may exit while they sleep - as with the stream pool)
  }
};

Pool &pool() {
  static Pool *p = new Pool;  // never destroyed
  return *p;
}

}  // namespace

void tsim_parallel_for_impl(size_t n, int max_threads, const std::function<void(size_t)> &fn) {
  if (n == 0) return;
  if (n == 1 || max_threads <= 1) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  Pool &P = pool();
  P.start();
  auto j = std::make_shared<Job>();
  j->fn = fn;
  j->n = n;
  j->max_workers = (int)(n - 1 < (size_t)(max_threads - 1) ? n - 1 : (size_t)(max_threads - 1));
  {
    std::lock_guard<std::mutex> lk(P.m);
    P.jobs.push_back(j);
  }
  P.cv.notify_all();
  Pool::run(*j);  // the caller works too
  while (j->done.load(std::memory_order_acquire) < n) std::this_thread::yield();  // indices still in a helper's hands
  {
    std::lock_guard<std::mutex> lk(P.m);
    for (size_t k = 0; k < P.jobs.size(); ++k)
      if (P.jobs[k] == j) {
        P.jobs.erase(P.jobs.begin() + (long)k);
        break;
      }
  }
}
