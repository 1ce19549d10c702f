// Internals shared by the host toolbox's translation units (toolbox.cpp: the synchronous flows; pipe.cpp: contexts over one or more
// GPUs with asynchronous jobs).  Not part of the C ABI.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../../include/zkp_toolbox.h"

struct zkp_statement {
  std::string label;
  std::vector<std::string> secrets;
  struct Point { std::string name; bool common; uint32_t rank; };
  std::vector<Point> points;
  uint32_t ni = 0, ns = 0;
  struct Constraint { uint32_t lhs; std::vector<std::pair<uint32_t, uint32_t>> lc; };
  std::vector<Constraint> cons;
  uint32_t terms = 0;
  // The allocation sequence IS part of the statement: every allocate_scalar / allocate_point call appends to the
  // transcript when it is made (prover.rs:52-73, verifier.rs:57-77), in whatever order the caller makes them.
  struct Alloc { bool is_point; uint32_t idx; };
  std::vector<Alloc> alloc;
  // number of leading scalar allocations (hashed once for a batch that starts from equal transcripts)
  uint32_t scalar_prefix() const {
    uint32_t n = 0;
    while (n < alloc.size() && !alloc[n].is_point) ++n;
    return n;
  }
};

namespace zkp {
namespace host {

constexpr size_t TB = ZKP_TRANSCRIPT_BYTES;

// Persistent worker pool: the three host phases of a batch call (prefix, phase A, phase B) would otherwise each
// pay thread creation for up to 64 threads (~1-2 ms, more than the GPU part of the call).
class Pool {
 public:
  static Pool& instance() { static Pool p; return p; }
  // runs fn(0) .. fn(k-1) concurrently (fn(0) on the calling thread) and returns when all are done
  void run(unsigned k, const std::function<void(unsigned)>& fn) {
    if (k <= 1) { fn(0); return; }
    std::unique_lock<std::mutex> call_lock(call_mu_);          // one parallel region at a time
    ensure_workers(k - 1);
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = &fn;
      active_ = k - 1;
      pending_ = k - 1;
      ++generation_;
    }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }
  ~Pool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++generation_; }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }

 private:
  void ensure_workers(unsigned k) {
    while (workers_.size() < k) {
      const unsigned id = (unsigned)workers_.size();
      unsigned gen;
      { std::lock_guard<std::mutex> lk(mu_); gen = generation_; }
      workers_.emplace_back([this, id, gen] { loop(id, gen); });
    }
  }
  void loop(unsigned id, unsigned seen) {
    for (;;) {
      const std::function<void(unsigned)>* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
        if (id < active_) job = job_;
      }
      if (job) {
        (*job)(id + 1);
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) done_cv_.notify_all();
      }
    }
  }
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> workers_;
  const std::function<void(unsigned)>* job_ = nullptr;
  unsigned active_ = 0, pending_ = 0, generation_ = 0;
  bool stop_ = false;
};

template <typename F>
inline void parallel_for(uint32_t n, int n_threads, F&& body) {
  unsigned t = n_threads > 0 ? (unsigned)n_threads : std::thread::hardware_concurrency();
  if (t == 0) t = 1;
  t = std::min<unsigned>(t, (n + 31) / 32 ? (n + 31) / 32 : 1);      // at least ~32 proofs per thread
  t = std::min<unsigned>(t, 128);
  if (t <= 1) { body(0u, n); return; }
  const uint32_t chunk = (n + t - 1) / t;
  Pool::instance().run(t, [&](unsigned k) {
    const uint32_t lo = std::min<uint32_t>(n, k * chunk), hi = std::min<uint32_t>(n, lo + chunk);
    if (lo < hi) body(lo, hi);
  });
}

// ---- fused (all-on-device) flows: zkp_mi355x.h (2c) ------------------------------------------------------------

struct FusedView {
  zkp_fused_statement fs{};
  std::vector<uint32_t> lhs, off, csc, cpt, order, seq;
  std::vector<const char*> slabels, plabels;
  explicit FusedView(const zkp_statement& st) {
    const uint32_t nc = (uint32_t)st.cons.size(), np = (uint32_t)st.points.size();
    auto pid = [&](uint32_t v) { return st.points[v].common ? st.points[v].rank : st.ns + st.points[v].rank; };
    lhs.resize(nc);
    off.assign(nc + 1, 0);
    for (uint32_t k = 0; k < nc; ++k) {
      lhs[k] = pid(st.cons[k].lhs);
      for (const auto& term : st.cons[k].lc) { csc.push_back(term.first); cpt.push_back(pid(term.second)); }
      off[k + 1] = (uint32_t)csc.size();
    }
    plabels.resize(np);
    for (uint32_t v = 0; v < np; ++v) { order.push_back(pid(v)); plabels[pid(v)] = st.points[v].name.c_str(); }
    for (const auto& s : st.secrets) slabels.push_back(s.c_str());
    for (const auto& al : st.alloc) seq.push_back(al.is_point ? pid(al.idx) : (0x80000000u | al.idx));
    fs.alloc_seq = seq.data();
    fs.shape = zkp_batch_statement{(uint32_t)st.secrets.size(), st.ns, st.ni, nc, lhs.data(), off.data(), csc.data(), cpt.data()};
    fs.label = st.label.c_str();
    fs.secret_labels = slabels.data();
    fs.point_labels = plabels.data();
    fs.alloc_order = order.data();
  }
};

// the device transcript programs need all blobs at one STROBE position; small batches stay with the host threads
// (a lone wavefront needs ~10 us per Keccak permutation, the host ~0.4 us)
inline bool use_fused(const uint8_t* ts, uint32_t N) {
  if (N < zkp_toolbox_get_fused_min_batch() || N == 0) return false;
  for (uint32_t j = 1; j < N; ++j)
    if (std::memcmp(ts + TB * (size_t)j + 200, ts + 200, 3) != 0) return false;
  return true;
}


// getrandom() until `len` bytes are there; false = the operating system gave none (callers fail closed: ZKP_TB_NO_ENTROPY)
bool os_entropy(uint8_t* out, size_t len);
// what `thread_rng()` is to the reference: a ChaCha20 stream keyed from the operating system (toolbox.cpp)
bool os_random(uint8_t* out, size_t len);

}  // namespace host
}  // namespace zkp
