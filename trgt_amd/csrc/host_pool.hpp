// trgt_amd/csrc/host_pool.hpp -- persistent host worker threads for the per-locus glue of trgt_locus_batch
// (the reference runs one rayon task per locus, src/commands/genotype.rs:179-187; here the host part of a locus is a few
// microseconds, so what matters is not paying a thread spawn per parallel section).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace trgt {

class HostPool {
 public:
  explicit HostPool(int n_threads) : n_(n_threads < 1 ? 1 : n_threads) {
    for (int t = 1; t < n_; ++t) th_.emplace_back([this, t]() { loop(t); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  int size() const { return n_; }

  // f(index, worker) for index in [0, n); indices are handed out in blocks of `grain` (dynamic balancing: loci differ in cost)
  template <typename F>
  void parallel_for(int64_t n, int64_t grain, F f) {
    if (n <= 0) return;
    if (n_ == 1 || n <= grain) { for (int64_t i = 0; i < n; ++i) f(i, 0); return; }
    std::atomic<int64_t> next{0};
    auto body = [&](int t) {
      for (;;) {
        const int64_t b = next.fetch_add(grain, std::memory_order_relaxed);
        if (b >= n) break;
        const int64_t e = b + grain < n ? b + grain : n;
        for (int64_t i = b; i < e; ++i) f(i, t);
      }
    };
    run(body);
  }

 private:
  void run(const std::function<void(int)>& body) {
    {
      std::lock_guard<std::mutex> g(m_);
      job_ = &body; pending_ = n_ - 1; ++gen_;
    }
    cv_.notify_all();
    body(0);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this]() { return pending_ == 0; });
    job_ = nullptr;
  }
  void loop(int t) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)>* job;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&]() { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        job = job_;
      }
      if (job) (*job)(t);
      {
        std::lock_guard<std::mutex> g(m_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  int n_;
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* job_ = nullptr;
  uint64_t gen_ = 0;
  int pending_ = 0;
  bool stop_ = false;
};

}  // namespace trgt
