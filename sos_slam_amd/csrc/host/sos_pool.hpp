// sos_pool.hpp -- helper threads for the facade's per-keyframe graph walks.
//
// The reference runs exactly these loops on its IndexThreadReduce workers (FS/FullSystemOptimize.cpp:125-182: linearizeAll_Reductor over
// the active residuals; util/IndexThreadReduce.h).  Here the per-residual arithmetic is on the device and what is left on the host is
// pointer chasing over ~2 x 38 k heap objects per walk: packWindow's record walk and the consumer of
// linearizeAll(true).  They are split into contiguous ranges whose results (record slots, per-range lists) are laid out by a prefix sum,
// so the output is the serial loop's output byte for byte whatever the thread count -- no reduction order to worry about.
//
// WalkPool::get().run(parts, fn): fn(part) for part in [0, parts), the caller takes parts itself; returns when all are done.  Helpers
// spin briefly (a walk is 0.1 - 1 ms: a condition-variable wake-up alone is ~50 us) and then sleep until the next call.
#pragma once

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace sos {

class WalkPool {
 public:
  static WalkPool &get() {
    static WalkPool p;
    return p;
  }
  // number of threads a walk is split over (the caller included); 1 = serial.  Default: min(4, hardware threads / 2), at least 1
  int threads() const { return nthreads_; }
  void setThreads(int n) {
    if (n < 1) n = 1;
    if (n > 16) n = 16;
    std::unique_lock<std::mutex> lk(m_);
    nthreads_ = n;
  }
  template <class F> void run(int parts, F &&fn) {
    // one walk at a time owns the helpers: a second caller (two systems driven from two threads of one process) walks by itself
    std::unique_lock<std::mutex> owner(run_m_, std::try_to_lock);
    if (parts <= 1 || nthreads_ <= 1 || !owner.owns_lock()) {
      for (int p = 0; p < parts; p++) fn(p);
      return;
    }
    ensureWorkers(nthreads_ - 1);
    std::function<void(int)> f(std::ref(fn));
    {
      std::unique_lock<std::mutex> lk(m_);
      job_ = &f;
      parts_ = parts;
      next_.store(0, std::memory_order_relaxed);
      done_.store(0, std::memory_order_relaxed);
      active_ = std::min((int)workers_.size(), nthreads_ - 1);
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    work();
    // every part has been handed out; wait until the helpers have finished theirs ...
    while (done_.load(std::memory_order_acquire) < parts) std::this_thread::yield();
    {
      std::unique_lock<std::mutex> lk(m_);
      job_ = nullptr;  // (a helper that wakes up late finds nothing to join)
    }
    // ... and have left the hand-out loop: the next run resets its counters, which nobody may still be reading
    while (busy_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  }
  ~WalkPool() {
    {
      std::unique_lock<std::mutex> lk(m_);
      stop_ = true;
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (std::thread &t : workers_) t.join();
  }

 private:
  WalkPool() {
    const unsigned hw = std::thread::hardware_concurrency();
    nthreads_ = (int)std::max(1u, std::min(4u, hw / 2));
  }
  void work() {
    for (;;) {
      const int p = next_.fetch_add(1, std::memory_order_relaxed);
      if (p >= parts_) break;
      (*job_)(p);
      done_.fetch_add(1, std::memory_order_release);
    }
  }
  void ensureWorkers(int n) {
    std::unique_lock<std::mutex> lk(m_);
    while ((int)workers_.size() < n) {
      const int id = (int)workers_.size();
      workers_.emplace_back([this, id] { loop(id); });
    }
  }
  void loop(int id) {
    unsigned long seen = 0;
    for (;;) {
      // spin for a while on the generation counter (back-to-back walks of one keyframe), then sleep
      int spins = 0;
      while (gen_.load(std::memory_order_acquire) == seen && ++spins < 2000) std::this_thread::yield();
      if (gen_.load(std::memory_order_acquire) == seen) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
      }
      seen = gen_.load(std::memory_order_acquire);
      if (stop_) return;
      bool mine;
      {
        std::unique_lock<std::mutex> lk(m_);
        mine = job_ != nullptr && id < active_;
        if (mine) busy_.fetch_add(1, std::memory_order_acq_rel);  // (decided under the lock run() clears job_ under)
      }
      if (mine) {
        work();
        busy_.fetch_sub(1, std::memory_order_acq_rel);
      }
    }
  }
  std::mutex m_, run_m_;
  std::condition_variable cv_;
  std::vector<std::thread> workers_;
  std::function<void(int)> *job_ = nullptr;
  std::atomic<unsigned long> gen_{0};
  std::atomic<int> next_{0}, done_{0}, busy_{0};
  int parts_ = 0, active_ = 0, nthreads_ = 1;
  bool stop_ = false;
};

}  // namespace sos
