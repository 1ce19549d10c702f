// zkp_pipe: engine contexts over ONE OR SEVERAL GPUs behind one object, with asynchronous jobs on host buffers
// (include/zkp_toolbox.h, "pipelines and device groups").
//
// Two things the reference's callers cannot get from one synchronous call per batch:
//   * throughput through the boundary: a lone call chain leaves most of the chip idle (narrow kernels, copies in front and behind).
//     zkp_*_submit puts a job on the next free context and returns; with three or more jobs in flight the copies of one job ride the
//     SDMA engines while the kernels of the others fill the CUs.  Entropy and batch weights are drawn on the device from a ChaCha20
//     stream keyed with 40 bytes of getrandom() per job (what thread_rng() is to prover.rs:82 / batch_verifier.rs:179).
//   * more than one GPU from one process: BatchVerifier is one in-process object (batch_verifier.rs:30-43); a pipe created over
//     device ids {0 .. 7} shards contiguous proof ranges [g N / G, (g + 1) N / G) over its contexts, one host thread per device,
//     common points replicated, and ANDs the verdicts on the host -- SURVEY 8(e) as written, no torch, no RCCL.
// Host buffers that are not pinned are staged through per-context pinned rings (a host memcpy at 26 - 130 GB/s, see
// profiles/r04_pcie_copy_rates.txt); pinned buffers (zkp_host_alloc / zkp_host_register) are handed to the DMA engines as they are.
#include <sys/random.h>

#include <algorithm>
#include <deque>
#include <memory>

#include "toolbox_internal.hpp"

using namespace zkp::host;

namespace {

struct PinBuf {
  uint8_t* p = nullptr;
  size_t cap = 0;
  int device = 0;                    // the ring lives on the NUMA node of the GPU it feeds (zkp_host_alloc_on)
  int ensure(size_t n) {
    if (n <= cap) return ZKP_OK;
    if (p) zkp_host_free(p);
    p = nullptr;
    cap = 0;
    void* q = nullptr;
    const size_t want = n + n / 4 + 4096;
    const int rc = zkp_host_alloc_on(&q, want, device);
    if (rc) return rc;
    p = static_cast<uint8_t*>(q);
    cap = want;
    return ZKP_OK;
  }
  ~PinBuf() { if (p) zkp_host_free(p); }
};

struct Slot {
  zkp_ctx* ctx = nullptr;
  int device = 0;                    // HIP ordinal
  int group = 0;                     // position in the device list the pipe was created over: one host thread per entry (synchronous calls, submitter threads)
  std::atomic<bool> busy{false};     // a job is on the context's stream (set by whoever submitted it, cleared by zkp_job_wait)
  std::atomic<bool> reserved{false}; // submitter threads: the caller has handed the slot to a job that its device's thread has yet to put on the stream
  std::mutex ctx_mu;                 // poll / wait on the context: one thread at a time (a kick from another thread just skips a context somebody holds)
  zkp_job* cur = nullptr;            // submitter threads: the OUTER job whose inner job is on this context (touched by the device's thread only)
  PinBuf in, out;
};

// One host thread per entry of the device list (round 5): asynchronous submits of a pipe over several GPUs are carried out here -- staging
// memcpy, ~40 enqueue calls and the fixed-base table check of a job cost the submitting thread 0.1 - 0.4 ms, and ONE caller thread doing
// that for eight GPUs is host-bound near three of them (profiles/r05_pipe_host_scaling.txt).  Between submits the thread polls the busy
// contexts of ITS device, so a finished job's copies out start without the caller (what kick_all does on the caller's thread otherwise).
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  struct Item { std::function<void()> run; struct zkp_job* outer; };
  std::deque<Item> q;
  bool stop = false;                 // zkp_pipe_destroy: whatever is still queued or on a stream is DISCARDED -- nothing is submitted any more, no output is copied
};

struct OutCopy { uint8_t* user; const uint8_t* staged; size_t bytes; };

void par_memcpy(void* dst, const void* src, size_t bytes, bool use_pool) {
  constexpr size_t chunk = 64u << 10;
  if (!bytes) return;
  if (!use_pool || bytes < (4u << 20)) { std::memcpy(dst, src, bytes); return; }
  const uint32_t n = (uint32_t)((bytes + chunk - 1) / chunk);
  parallel_for(n, 8, [&](uint32_t lo, uint32_t hi) {
    const size_t a = (size_t)lo * chunk, b = std::min(bytes, (size_t)hi * chunk);
    std::memcpy(static_cast<uint8_t*>(dst) + a, static_cast<const uint8_t*>(src) + a, b - a);
  });
}

}  // namespace

struct zkp_pipe {
  std::vector<std::unique_ptr<Slot>> slots;
  std::vector<int> devices;          // as given (duplicates allowed: several contexts per GPU)
  size_t next = 0;
  std::mutex err_mu;                 // (the per-device threads of the synchronous calls may all report)
  std::string last_error;
  int threaded = -1;                 // submitter threads: -1 = default (on when the pipe spans more than one entry of the device list), 0 / 1 = zkp_pipe_set_submit_threads
  std::vector<std::unique_ptr<Worker>> workers;      // one per entry of the device list, started with the first threaded submit
  bool use_threads() const { return threaded < 0 ? devices.size() > 1 : threaded != 0; }
  int find_free() {
    for (size_t i = 0; i < slots.size(); ++i) {
      const size_t k = (next + i) % slots.size();
      if (!slots[k]->busy.load(std::memory_order_acquire) && !slots[k]->reserved.load(std::memory_order_acquire)) { next = (k + 1) % slots.size(); return (int)k; }
    }
    return -1;
  }
};

struct zkp_job {
  zkp_pipe* pipe = nullptr;
  int slot = -1;
  // submitter threads: the handle the caller holds is an OUTER job; the device's thread performs the submit and leaves the real job in `inner`
  bool outer = false;
  zkp_job* inner = nullptr;
  std::mutex mu;
  std::condition_variable cv;
  int state = 0;                     // outer: 0 = queued, 1 = on the stream (inner is set), 2 = the submit failed, 3 = retired by the device's thread: rc holds
  int rc = 0;                        //        what zkp_job_wait returns (states 2 and 3)
  bool immediate = false;            // the call ran synchronously inside submit (small or ragged batch: host-transcript route)
  int immediate_rc = 0;
  char kind = 0;                     // 'P', 'V', 'E', 'B'
  bool use_pool = true;
  std::vector<OutCopy> outs;
  int invalid_point = 0;
  uint32_t N = 0, K = 0;
  uint8_t* results = nullptr;        // 'V' / 'E': the caller's [N]
  int* verdicts = nullptr;           // 'B': the caller's [K]
};

namespace {

// Assigns every buffer of a job either the caller's own memory (pinned: DMA straight from / to it) or a place in the slot's pinned
// rings.  Sizes are summed first (ensure() may reallocate), then the pointers are handed out.
struct Stager {
  Slot& s;
  bool use_pool;
  std::vector<OutCopy>& outs;
  size_t in_off = 0, out_off = 0;
  static size_t pad(size_t n) { return (n + 255) & ~(size_t)255; }
  const uint8_t* in(const uint8_t* user, size_t bytes) {
    if (!user || !bytes || zkp_host_is_pinned(user)) return user;
    uint8_t* d = s.in.p + in_off;
    in_off += pad(bytes);
    par_memcpy(d, user, bytes, use_pool);
    return d;
  }
  // rows [rows][N][elem] cut out of the caller's [rows][stride][elem]; *stride_out = what the job has to be told
  const uint8_t* in_rows(const uint8_t* user, size_t rows, size_t N, size_t stride, size_t elem, uint32_t* stride_out) {
    *stride_out = (uint32_t)stride;
    if (!user || !rows || !N || zkp_host_is_pinned(user)) return user;
    uint8_t* d = s.in.p + in_off;
    in_off += pad(rows * N * elem);
    if (stride == N) par_memcpy(d, user, rows * N * elem, use_pool);
    else for (size_t r = 0; r < rows; ++r) par_memcpy(d + r * N * elem, user + r * stride * elem, N * elem, use_pool);
    *stride_out = (uint32_t)N;
    return d;
  }
  uint8_t* out(uint8_t* user, size_t bytes) {
    if (!user || !bytes || zkp_host_is_pinned(user)) return user;
    uint8_t* d = s.out.p + out_off;
    out_off += pad(bytes);
    outs.push_back({user, d, bytes});
    return d;
  }
};

bool draw_seed(uint8_t seed[40]) { return os_entropy(seed, 40); }

// the transcripts of a fused job must stand at one STROBE position (one shared blob does by construction)
bool fused_ok(const uint8_t* ts, uint32_t N, bool shared) {
  if (N == 0 || N < zkp_toolbox_get_fused_min_batch()) return false;
  return shared || use_fused(ts, N);
}

// ---- the slow path of a job: small or ragged batches run the synchronous call now, on gathered copies ------------------------
struct Gathered {
  std::vector<uint8_t> ts, inst, w;
  void transcripts(const uint8_t* t, uint32_t N, bool shared) {
    ts.resize(TB * (size_t)N);
    if (shared) for (uint32_t j = 0; j < N; ++j) std::memcpy(ts.data() + TB * (size_t)j, t, TB);
    else std::memcpy(ts.data(), t, ts.size());
  }
  void rows(std::vector<uint8_t>& dst, const uint8_t* src, size_t rows, size_t N, size_t stride, size_t elem) {
    dst.resize(rows * N * elem);
    for (size_t r = 0; r < rows && src; ++r) std::memcpy(dst.data() + r * N * elem, src + r * stride * elem, N * elem);
  }
};

// text for a negative code: the engine's own codes (> -10) come with zkp_last_error(); the toolbox's (<= -10) name themselves
std::string err_text(int rc) {
  if (rc > -10) return zkp_last_error();
  switch (rc) {
    case ZKP_TB_BAD_STATEMENT: return "malformed statement descriptor or NULL / inconsistent argument";
    case ZKP_TB_INVALID_POINT: return "a point handed to the prover does not decode";
    case ZKP_TB_NO_ENTROPY: return "getrandom() failed";
    default: return "toolbox error " + std::to_string(rc);
  }
}
int fail_pipe(zkp_pipe* p, int rc, const std::string& what) {
  if (p) {
    std::lock_guard<std::mutex> lk(p->err_mu);
    p->last_error = what + (rc < 0 && rc > -10 ? std::string(": ") + zkp_last_error() : std::string());
  }
  return rc;
}

}  // namespace

extern "C" {

uint32_t zkp_pipe_shard_plan(uint32_t n_items, uint32_t unit, uint32_t n_contexts, uint32_t fused_min_batch, uint32_t* lo, uint32_t* hi) {
  if (!n_items || !n_contexts || !lo || !hi) return 0;
  const uint32_t u = std::max<uint32_t>(1, unit);
  const uint64_t min_items = std::max<uint64_t>(1, ((uint64_t)std::max<uint32_t>(1, fused_min_batch) + u - 1) / u);
  const uint32_t G = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(n_contexts, n_items), n_items / min_items));
  uint32_t k = 0;
  for (uint32_t g = 0; g < G; ++g) {
    const uint32_t a = (uint32_t)((uint64_t)g * n_items / G), b = (uint32_t)((uint64_t)(g + 1) * n_items / G);
    if (a < b) { lo[k] = a; hi[k] = b; ++k; }
  }
  return k;
}

int zkp_pipe_create(zkp_pipe** out, const int* device_ids, int n_devices, int contexts_per_device) {
  if (!out) return ZKP_TB_BAD_STATEMENT;
  *out = nullptr;
  if (!device_ids || n_devices <= 0 || contexts_per_device <= 0 || (int64_t)n_devices * contexts_per_device > 1024) return ZKP_TB_BAD_STATEMENT;
  std::unique_ptr<zkp_pipe> p(new zkp_pipe());
  p->devices.assign(device_ids, device_ids + n_devices);
  // slot order = device-major round robin: consecutive jobs go to different GPUs first, then to the next context of each
  for (int k = 0; k < contexts_per_device; ++k)
    for (int d = 0; d < n_devices; ++d) {
      std::unique_ptr<Slot> s(new Slot());
      s->device = device_ids[d];
      s->group = d;
      s->in.device = s->out.device = device_ids[d];
      const int rc = zkp_ctx_create(&s->ctx, device_ids[d]);
      if (rc) {
        for (auto& q : p->slots) zkp_ctx_destroy(q->ctx);
        return rc;
      }
      p->slots.push_back(std::move(s));
    }
  *out = p.release();
  return ZKP_TB_OK;
}

void zkp_pipe_destroy(zkp_pipe* p) {
  if (!p) return;
  for (auto& w : p->workers) {                           // what is still queued is dropped, what is on a stream discarded; then the threads end
    { std::lock_guard<std::mutex> lk(w->mu); w->stop = true; }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
  }
  // Jobs still in flight belong to zkp_job handles nobody waited for; their owners (and the buffers and verdict words the jobs name) may be
  // gone already.  Discard them: the kernels are waited for, no copy out is issued any more, nothing is written to caller memory.
  for (auto& s : p->slots) {
    if (s->ctx) { (void)zkp_ctx_job_discard(s->ctx); zkp_ctx_destroy(s->ctx); }
  }
  delete p;
}
int zkp_pipe_num_contexts(const zkp_pipe* p) { return p ? (int)p->slots.size() : 0; }
int zkp_pipe_num_devices(const zkp_pipe* p) { return p ? (int)p->devices.size() : 0; }
zkp_ctx* zkp_pipe_context(zkp_pipe* p, int i) { return (p && i >= 0 && (size_t)i < p->slots.size()) ? p->slots[i]->ctx : nullptr; }
int zkp_pipe_context_device(const zkp_pipe* p, int i) { return (p && i >= 0 && (size_t)i < p->slots.size()) ? p->slots[i]->device : -1; }
int zkp_pipe_jobs_in_flight(const zkp_pipe* p) {
  int n = 0;
  if (p) for (auto& s : p->slots) n += (s->busy.load() || s->reserved.load()) ? 1 : 0;
  return n;
}
int zkp_pipe_set_submit_threads(zkp_pipe* p, int on) {
  if (!p) return ZKP_TB_BAD_STATEMENT;
  for (auto& s : p->slots)
    if (s->busy.load() || s->reserved.load()) return fail_pipe(p, ZKP_TB_PIPE_FULL, "zkp_pipe_set_submit_threads: wait for the jobs in flight first");
  p->threaded = on < 0 ? -1 : (on ? 1 : 0);
  return ZKP_TB_OK;
}
// (submitter threads and the per-device threads of the synchronous calls report asynchronously: the text is copied under err_mu into a buffer of the CALLING
// thread, valid until that thread asks again)
const char* zkp_pipe_last_error(const zkp_pipe* p) {
  if (!p) return "";
  static thread_local std::string text;
  { std::lock_guard<std::mutex> lk(const_cast<zkp_pipe*>(p)->err_mu); text = p->last_error; }
  return text.c_str();
}

}  // extern "C"

namespace {

// Jobs whose kernels have finished get their copies out started (zkp_ctx_job_poll issues them: zkp_mi355x.h 2d): called whenever the caller
// is in the pipe anyway, so that a finished job's outputs are already travelling when somebody waits for it.
void kick_slot(Slot* s) {
  if (!s->busy.load(std::memory_order_acquire)) return;
  std::unique_lock<std::mutex> lk(s->ctx_mu, std::try_to_lock);     // (somebody is waiting on it or kicking it already)
  if (lk.owns_lock() && s->busy.load(std::memory_order_acquire)) (void)zkp_ctx_job_poll(s->ctx);
}
void kick_all(zkp_pipe* p) {
  for (auto& s : p->slots) kick_slot(s.get());
}
int wait_inner(zkp_job* jp);
// state >= 2 hands the job back to whoever holds the handle: zkp_job_wait may return and free it the moment it sees the state, so the notification goes out
// UNDER the job's mutex (the waiter cannot leave cv.wait -- it re-takes o->mu -- before this thread has let go of the condition variable), and nothing
// touches `o` afterwards.  pipe_gone: the pipe is being destroyed under the job -- the handle must not reach into it any more.
void finish_outer(zkp_job* o, int state, int rc, bool pipe_gone = false) {
  std::lock_guard<std::mutex> lk(o->mu);
  o->rc = rc;
  if (pipe_gone) o->pipe = nullptr;
  o->state = state;
  o->cv.notify_all();
}
// the device's thread: jobs of this device whose kernels are done get their copies out issued (poll), finished ones are retired completely --
// staged outputs copied to the caller's buffers, verdict words written -- so that the caller's zkp_job_wait only picks up the result
void retire_group(zkp_pipe* p, int group, bool discard) {
  for (auto& sp : p->slots) {
    Slot* s = sp.get();
    if (s->group != group || !s->cur || !s->busy.load(std::memory_order_acquire)) continue;
    zkp_job* o = s->cur;
    if (discard) {
      // the pipe is going away under a job nobody waited for: its owner, buffers and verdict words may be gone.  Kernels are waited for, no copy out is
      // issued, nothing is written to caller memory (zkp_ctx_job_discard); whoever still holds the handle gets an error
      { std::lock_guard<std::mutex> lk(s->ctx_mu); (void)zkp_ctx_job_discard(s->ctx); s->busy.store(false, std::memory_order_release); }
      s->cur = nullptr;
      delete o->inner;
      o->inner = nullptr;
      finish_outer(o, 3, ZKP_TB_BAD_STATEMENT, true);
      continue;
    }
    bool done;
    { std::lock_guard<std::mutex> lk(s->ctx_mu); done = zkp_ctx_job_poll(s->ctx) != 0; }
    if (!done) continue;
    s->cur = nullptr;
    finish_outer(o, 3, wait_inner(o->inner));
  }
}

void worker_loop(zkp_pipe* p, int group, Worker* w) {
  for (;;) {
    Worker::Item it{nullptr, nullptr};
    bool stopping;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      // wake for work -- or, while jobs of this device are on their streams, every 200 us to move them along (a finished job's copies out are issued by the
      // first poll after its kernels); with nothing in flight the thread sleeps until somebody queues work
      bool in_flight = false;
      for (auto& sp : p->slots) in_flight = in_flight || (sp->group == group && sp->cur);
      if (in_flight) w->cv.wait_for(lk, std::chrono::microseconds(200), [&] { return w->stop || !w->q.empty(); });
      else w->cv.wait(lk, [&] { return w->stop || !w->q.empty(); });
      stopping = w->stop;
      if (!w->q.empty()) {
        it = std::move(w->q.front());
        w->q.pop_front();
      }
    }
    if (it.outer) {
      if (stopping) finish_outer(it.outer, 2, ZKP_TB_BAD_STATEMENT, true); // never submitted: the buffers it names may be gone
      else it.run();
    }
    retire_group(p, group, stopping);
    if (stopping && !it.outer) return;                       // (queue drained, every job of this device discarded)
  }
}
Worker* worker_of(zkp_pipe* p, int group) {
  if (p->workers.size() != p->devices.size()) {
    p->workers.clear();
    for (size_t g = 0; g < p->devices.size(); ++g) p->workers.emplace_back(new Worker());
  }
  Worker* w = p->workers[group].get();
  if (!w->th.joinable()) w->th = std::thread(worker_loop, p, group, w);
  return w;
}

// what every submit starts with; *slot_out < 0 with rc == 0 never happens
int begin_job(zkp_pipe* p, const zkp_statement* st, zkp_job** job, int want_slot, std::unique_ptr<zkp_job>& j, Slot** slot_out) {
  if (!p || !st || !job) return ZKP_TB_BAD_STATEMENT;
  *job = nullptr;
  if (want_slot < 0) kick_all(p);                        // (the sharded synchronous calls and the submitter threads pass their slot: each looks after its own device)
  int k = want_slot;
  if (k < 0) k = p->find_free();
  else if ((size_t)k >= p->slots.size() || p->slots[k]->busy.load()) k = -1;
  if (k < 0) return fail_pipe(p, ZKP_TB_PIPE_FULL, "every context of the pipe has a job in flight: zkp_job_wait one first");
  j.reset(new zkp_job());
  j->pipe = p;
  j->slot = k;
  *slot_out = p->slots[k].get();
  return ZKP_TB_OK;
}
int finish_immediate(zkp_job** job, std::unique_ptr<zkp_job>& j, int rc) {
  j->immediate = true;
  j->immediate_rc = rc;
  *job = j.release();
  return ZKP_TB_OK;
}
int finish_submitted(zkp_pipe* p, Slot* s, zkp_job** job, std::unique_ptr<zkp_job>& j, int rc, const char* what) {
  if (rc) return fail_pipe(p, rc, what);
  s->busy.store(true, std::memory_order_release);
  *job = j.release();
  return ZKP_TB_OK;
}

int prove_submit_on(zkp_pipe* p, int want_slot, bool use_pool, const zkp_statement* st, uint32_t N, uint32_t flags, const uint8_t* ts, const uint8_t* secrets,
                    const uint8_t* inst, uint32_t inst_stride, const uint8_t* common, const uint8_t* entropy, uint8_t* ts_out, uint8_t* challenges,
                    uint8_t* responses, uint8_t* commitments, zkp_job** job) {
  std::unique_ptr<zkp_job> j;
  Slot* s = nullptr;
  int rc = begin_job(p, st, job, want_slot, j, &s);
  if (rc) return rc;
  j->kind = 'P';
  j->N = N;
  j->use_pool = use_pool;
  if (N == 0) return finish_immediate(job, j, ZKP_TB_OK);
  if (!ts || (st->ni && inst_stride < N)) return fail_pipe(p, ZKP_TB_BAD_STATEMENT, "prove: transcripts == NULL or inst_stride < N");
  const bool shared = (flags & ZKP_JOB_SHARED_TRANSCRIPT) != 0;
  const uint32_t m = (uint32_t)st->secrets.size(), nc = (uint32_t)st->cons.size(), ni = st->ni, ns = st->ns;
  if (!fused_ok(ts, N, shared)) {
    Gathered g;
    g.transcripts(ts, N, shared);
    g.rows(g.inst, inst, ni, N, inst_stride, 32);
    rc = zkp_prove_batch(s->ctx, st, N, g.ts.data(), secrets, g.inst.data(), common, entropy, 0, challenges, responses, commitments);
    if (ts_out && rc >= 0) std::memcpy(ts_out, g.ts.data(), g.ts.size());
    return finish_immediate(job, j, rc);
  }
  uint8_t seed[40];
  if (!entropy && !draw_seed(seed)) return fail_pipe(p, ZKP_TB_NO_ENTROPY, "getrandom() failed");
  if (ns && N >= 32) { rc = zkp_ctx_prepare_fixed_points(s->ctx, ns, common); if (rc) return fail_pipe(p, rc, "zkp_ctx_prepare_fixed_points"); }
  const size_t in_bytes = (shared ? TB : TB * (size_t)N) + 32 * (size_t)N * m + 32 * (size_t)ni * N + 32 * (size_t)ns + 32 * (size_t)N + 8 * 256;
  const size_t out_bytes = TB * (size_t)N + 32 * (size_t)N + 32 * (size_t)N * m + 32 * (size_t)N * nc + 8 * 256;
  if ((rc = s->in.ensure(in_bytes)) || (rc = s->out.ensure(out_bytes))) return fail_pipe(p, rc, "pinned staging");
  Stager sg{*s, use_pool, j->outs};
  uint32_t istr = inst_stride;
  const uint8_t* a_ts = sg.in(ts, shared ? TB : TB * (size_t)N);
  const uint8_t* a_sec = sg.in(secrets, 32 * (size_t)N * m);
  const uint8_t* a_inst = sg.in_rows(inst, ni, N, inst_stride, 32, &istr);
  const uint8_t* a_com = sg.in(common, 32 * (size_t)ns);
  const uint8_t* a_ent = sg.in(entropy, 32 * (size_t)N);
  uint8_t* o_ts = sg.out(ts_out, TB * (size_t)N);
  uint8_t* o_chal = sg.out(challenges, 32 * (size_t)N);
  uint8_t* o_resp = sg.out(responses, 32 * (size_t)N * m);
  uint8_t* o_coms = sg.out(commitments, 32 * (size_t)N * nc);
  FusedView fv(*st);
  rc = zkp_fused_prove_submit(s->ctx, &fv.fs, N, flags & ZKP_JOB_SHARED_TRANSCRIPT, a_ts, a_sec, a_inst, istr, a_com, a_ent, entropy ? nullptr : seed, o_ts, o_chal,
                              o_resp, o_coms, &j->invalid_point);
  return finish_submitted(p, s, job, j, rc, "zkp_fused_prove_submit");
}

int verify_compact_submit_on(zkp_pipe* p, int want_slot, bool use_pool, const zkp_statement* st, uint32_t N, uint32_t flags, const uint8_t* ts, const uint8_t* inst,
                             uint32_t inst_stride, const uint8_t* common, const uint8_t* challenges, const uint8_t* responses, uint8_t* ts_out, uint8_t* results,
                             zkp_job** job) {
  std::unique_ptr<zkp_job> j;
  Slot* s = nullptr;
  int rc = begin_job(p, st, job, want_slot, j, &s);
  if (rc) return rc;
  j->kind = 'V';
  j->N = N;
  j->results = results;
  j->use_pool = use_pool;
  if (N == 0) return finish_immediate(job, j, ZKP_TB_OK);
  if (!ts || !results || (st->ni && inst_stride < N)) return fail_pipe(p, ZKP_TB_BAD_STATEMENT, "verify: transcripts / results == NULL or inst_stride < N");
  const bool shared = (flags & ZKP_JOB_SHARED_TRANSCRIPT) != 0;
  const uint32_t m = (uint32_t)st->secrets.size(), ni = st->ni, ns = st->ns;
  if (!fused_ok(ts, N, shared)) {
    Gathered g;
    g.transcripts(ts, N, shared);
    g.rows(g.inst, inst, ni, N, inst_stride, 32);
    rc = zkp_verify_compact_batch(s->ctx, st, N, g.ts.data(), g.inst.data(), common, challenges, responses, 0, results);
    if (ts_out && rc >= 0) std::memcpy(ts_out, g.ts.data(), g.ts.size());
    return finish_immediate(job, j, rc);
  }
  if (ns && N >= 32) { rc = zkp_ctx_prepare_fixed_points(s->ctx, ns, common); if (rc) return fail_pipe(p, rc, "zkp_ctx_prepare_fixed_points"); }
  const size_t in_bytes = (shared ? TB : TB * (size_t)N) + 32 * (size_t)ni * N + 32 * (size_t)ns + 32 * (size_t)N + 32 * (size_t)N * m + 8 * 256;
  const size_t out_bytes = TB * (size_t)N + N + 8 * 256;
  if ((rc = s->in.ensure(in_bytes)) || (rc = s->out.ensure(out_bytes))) return fail_pipe(p, rc, "pinned staging");
  Stager sg{*s, use_pool, j->outs};
  uint32_t istr = inst_stride;
  const uint8_t* a_ts = sg.in(ts, shared ? TB : TB * (size_t)N);
  const uint8_t* a_inst = sg.in_rows(inst, ni, N, inst_stride, 32, &istr);
  const uint8_t* a_com = sg.in(common, 32 * (size_t)ns);
  const uint8_t* a_chal = sg.in(challenges, 32 * (size_t)N);
  const uint8_t* a_resp = sg.in(responses, 32 * (size_t)N * m);
  uint8_t* o_ts = sg.out(ts_out, TB * (size_t)N);
  uint8_t* o_res = sg.out(results, N);
  FusedView fv(*st);
  rc = zkp_fused_verify_compact_submit(s->ctx, &fv.fs, N, flags & ZKP_JOB_SHARED_TRANSCRIPT, a_ts, a_inst, istr, a_com, a_chal, a_resp, o_ts, o_res);
  return finish_submitted(p, s, job, j, rc, "zkp_fused_verify_compact_submit");
}

int verify_each_submit_on(zkp_pipe* p, int want_slot, bool use_pool, const zkp_statement* st, uint32_t N, uint32_t flags, const uint8_t* ts, const uint8_t* inst,
                          uint32_t inst_stride, const uint8_t* common, const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16,
                          uint8_t* ts_out, uint8_t* results, zkp_job** job) {
  std::unique_ptr<zkp_job> j;
  Slot* s = nullptr;
  int rc = begin_job(p, st, job, want_slot, j, &s);
  if (rc) return rc;
  j->kind = 'E';
  j->N = N;
  j->results = results;
  j->use_pool = use_pool;
  if (N == 0) return finish_immediate(job, j, ZKP_TB_OK);
  if (!ts || !results || (st->ni && inst_stride < N)) return fail_pipe(p, ZKP_TB_BAD_STATEMENT, "verify: transcripts / results == NULL or inst_stride < N");
  const bool shared = (flags & ZKP_JOB_SHARED_TRANSCRIPT) != 0;
  const uint32_t m = (uint32_t)st->secrets.size(), nc = (uint32_t)st->cons.size(), ni = st->ni, ns = st->ns;
  if (!fused_ok(ts, N, shared)) {
    Gathered g;
    g.transcripts(ts, N, shared);
    g.rows(g.inst, inst, ni, N, inst_stride, 32);
    rc = zkp_verify_batchable_each(s->ctx, st, N, g.ts.data(), g.inst.data(), common, commitments, responses, weights16, 0, results);
    if (ts_out && rc >= 0) std::memcpy(ts_out, g.ts.data(), g.ts.size());
    return finish_immediate(job, j, rc);
  }
  uint8_t seed[40];
  if (!weights16 && nc && !draw_seed(seed)) return fail_pipe(p, ZKP_TB_NO_ENTROPY, "getrandom() failed");
  if (ns && N >= 32) { rc = zkp_ctx_prepare_fixed_points(s->ctx, ns, common); if (rc) return fail_pipe(p, rc, "zkp_ctx_prepare_fixed_points"); }
  const size_t in_bytes = (shared ? TB : TB * (size_t)N) + 32 * (size_t)ni * N + 32 * (size_t)ns + 32 * (size_t)N * nc + 32 * (size_t)N * m + 16 * (size_t)N * nc + 8 * 256;
  const size_t out_bytes = TB * (size_t)N + N + 8 * 256;
  if ((rc = s->in.ensure(in_bytes)) || (rc = s->out.ensure(out_bytes))) return fail_pipe(p, rc, "pinned staging");
  Stager sg{*s, use_pool, j->outs};
  uint32_t istr = inst_stride;
  const uint8_t* a_ts = sg.in(ts, shared ? TB : TB * (size_t)N);
  const uint8_t* a_inst = sg.in_rows(inst, ni, N, inst_stride, 32, &istr);
  const uint8_t* a_com = sg.in(common, 32 * (size_t)ns);
  const uint8_t* a_coms = sg.in(commitments, 32 * (size_t)N * nc);
  const uint8_t* a_resp = sg.in(responses, 32 * (size_t)N * m);
  const uint8_t* a_w = sg.in(weights16, 16 * (size_t)N * nc);
  uint8_t* o_ts = sg.out(ts_out, TB * (size_t)N);
  uint8_t* o_res = sg.out(results, N);
  FusedView fv(*st);
  rc = zkp_fused_verify_batchable_submit(s->ctx, &fv.fs, N, flags & ZKP_JOB_SHARED_TRANSCRIPT, a_ts, a_inst, istr, a_com, a_coms, a_resp, a_w, (weights16 || !nc) ? nullptr : seed,
                                         o_ts, o_res);
  return finish_submitted(p, s, job, j, rc, "zkp_fused_verify_batchable_submit");
}

int batch_many_submit_on(zkp_pipe* p, int want_slot, bool use_pool, const zkp_statement* st, uint32_t K, uint32_t N_each, uint32_t flags, const uint8_t* ts,
                         const uint8_t* inst, uint32_t inst_stride, const uint8_t* common, const uint8_t* commitments, const uint8_t* responses,
                         const uint8_t* weights16, uint32_t w_stride, uint8_t* ts_out, int* verdicts, zkp_job** job) {
  std::unique_ptr<zkp_job> j;
  Slot* s = nullptr;
  int rc = begin_job(p, st, job, want_slot, j, &s);
  if (rc) return rc;
  if (!verdicts || K == 0 || N_each == 0 || (uint64_t)K * N_each > 0x7fffffffull) return fail_pipe(p, ZKP_TB_BAD_STATEMENT, "batch verification: verdicts == NULL, or n_batches / N_each out of range");
  const uint32_t N = K * N_each;
  j->kind = 'B';
  j->N = N;
  j->K = K;
  j->verdicts = verdicts;
  j->use_pool = use_pool;
  if (!ts || (st->ni && inst_stride < N) || (weights16 && w_stride < N)) return fail_pipe(p, ZKP_TB_BAD_STATEMENT, "batch verification: transcripts == NULL or a stride smaller than the proof count");
  const bool shared = (flags & ZKP_JOB_SHARED_TRANSCRIPT) != 0;
  const uint32_t m = (uint32_t)st->secrets.size(), nc = (uint32_t)st->cons.size(), ni = st->ni, ns = st->ns;
  if (!fused_ok(ts, N, shared)) {
    Gathered g;
    g.transcripts(ts, N, shared);
    g.rows(g.inst, inst, ni, N, inst_stride, 32);
    if (weights16) g.rows(g.w, weights16, nc, N, w_stride, 16);
    rc = zkp_batch_verify_many(s->ctx, st, K, N_each, N, g.ts.data(), g.inst.data(), common, commitments, responses, weights16 ? g.w.data() : nullptr, 0, verdicts);
    if (ts_out && rc >= 0) std::memcpy(ts_out, g.ts.data(), g.ts.size());
    return finish_immediate(job, j, rc);
  }
  uint8_t seed[40];
  if (!weights16 && nc && !draw_seed(seed)) return fail_pipe(p, ZKP_TB_NO_ENTROPY, "getrandom() failed");
  const size_t in_bytes = (shared ? TB : TB * (size_t)N) + 32 * (size_t)ni * N + 32 * (size_t)ns + 32 * (size_t)N * nc + 32 * (size_t)N * m + 16 * (size_t)N * nc + 8 * 256;
  const size_t out_bytes = TB * (size_t)N + 8 * 256;
  if ((rc = s->in.ensure(in_bytes)) || (rc = s->out.ensure(out_bytes))) return fail_pipe(p, rc, "pinned staging");
  Stager sg{*s, use_pool, j->outs};
  uint32_t istr = inst_stride, wstr = w_stride;
  const uint8_t* a_ts = sg.in(ts, shared ? TB : TB * (size_t)N);
  const uint8_t* a_inst = sg.in_rows(inst, ni, N, inst_stride, 32, &istr);
  const uint8_t* a_com = sg.in(common, 32 * (size_t)ns);
  const uint8_t* a_coms = sg.in(commitments, 32 * (size_t)N * nc);
  const uint8_t* a_resp = sg.in(responses, 32 * (size_t)N * m);
  const uint8_t* a_w = sg.in_rows(weights16, nc, N, w_stride, 16, &wstr);
  uint8_t* o_ts = sg.out(ts_out, TB * (size_t)N);
  FusedView fv(*st);
  rc = zkp_fused_batch_verify_many_submit(s->ctx, &fv.fs, K, N_each, flags & ZKP_JOB_SHARED_TRANSCRIPT, a_ts, a_inst, istr, a_com, a_coms, a_resp, a_w, wstr,
                                          (weights16 || !nc) ? nullptr : seed, o_ts, verdicts);
  return finish_submitted(p, s, job, j, rc, "zkp_fused_batch_verify_many_submit");
}

// submitter threads: the caller reserves a context and gets an OUTER handle at once; the device's thread carries out the submit (do_submit(slot, &inner))
template <typename F>
int submit_threaded(zkp_pipe* p, const zkp_statement* st, zkp_job** job, F&& do_submit) {
  if (!p || !st || !job) return ZKP_TB_BAD_STATEMENT;
  *job = nullptr;
  const int k = p->find_free();
  if (k < 0) return fail_pipe(p, ZKP_TB_PIPE_FULL, "every context of the pipe has a job in flight: zkp_job_wait one first");
  Slot* s = p->slots[k].get();
  s->reserved.store(true, std::memory_order_release);
  zkp_job* o = new zkp_job();
  o->pipe = p;
  o->slot = k;
  o->outer = true;
  Worker* w = worker_of(p, s->group);
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->q.push_back(Worker::Item{[s, o, k, fn = std::forward<F>(do_submit)]() mutable {
      zkp_job* in = nullptr;
      const int rc = fn(k, &in);
      if (rc) { finish_outer(o, 2, rc); return; }
      o->inner = in;
      if (in->immediate) { finish_outer(o, 3, wait_inner(in)); return; }      // (small or ragged batch: it ran inside the submit)
      s->cur = o;
      { std::lock_guard<std::mutex> lk2(o->mu); o->state = 1; }
    }, o});
  }
  w->cv.notify_one();
  *job = o;
  return ZKP_TB_OK;
}

}  // namespace

extern "C" {

int zkp_prove_batch_submit(zkp_pipe* p, const zkp_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts, const uint8_t* secrets,
                           const uint8_t* inst_points, uint32_t inst_stride, const uint8_t* common_points, const uint8_t* entropy, uint8_t* transcripts_out,
                           uint8_t* challenges, uint8_t* responses, uint8_t* commitments, zkp_job** job) {
  if (p && p->use_threads())
    return submit_threaded(p, st, job, [=](int k, zkp_job** in) {
      return prove_submit_on(p, k, false, st, N, flags, transcripts, secrets, inst_points, inst_stride, common_points, entropy, transcripts_out, challenges, responses, commitments, in);
    });
  return prove_submit_on(p, -1, true, st, N, flags, transcripts, secrets, inst_points, inst_stride, common_points, entropy, transcripts_out, challenges, responses,
                         commitments, job);
}
int zkp_verify_compact_batch_submit(zkp_pipe* p, const zkp_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts, const uint8_t* inst_points,
                                    uint32_t inst_stride, const uint8_t* common_points, const uint8_t* challenges, const uint8_t* responses,
                                    uint8_t* transcripts_out, uint8_t* results, zkp_job** job) {
  if (p && p->use_threads())
    return submit_threaded(p, st, job, [=](int k, zkp_job** in) {
      return verify_compact_submit_on(p, k, false, st, N, flags, transcripts, inst_points, inst_stride, common_points, challenges, responses, transcripts_out, results, in);
    });
  return verify_compact_submit_on(p, -1, true, st, N, flags, transcripts, inst_points, inst_stride, common_points, challenges, responses, transcripts_out, results, job);
}
int zkp_verify_batchable_each_submit(zkp_pipe* p, const zkp_statement* st, uint32_t N, uint32_t flags, const uint8_t* transcripts, const uint8_t* inst_points,
                                     uint32_t inst_stride, const uint8_t* common_points, const uint8_t* commitments, const uint8_t* responses,
                                     const uint8_t* weights16, uint8_t* transcripts_out, uint8_t* results, zkp_job** job) {
  if (p && p->use_threads())
    return submit_threaded(p, st, job, [=](int k, zkp_job** in) {
      return verify_each_submit_on(p, k, false, st, N, flags, transcripts, inst_points, inst_stride, common_points, commitments, responses, weights16, transcripts_out, results, in);
    });
  return verify_each_submit_on(p, -1, true, st, N, flags, transcripts, inst_points, inst_stride, common_points, commitments, responses, weights16, transcripts_out,
                               results, job);
}
int zkp_batch_verify_many_submit(zkp_pipe* p, const zkp_statement* st, uint32_t n_batches, uint32_t N_each, uint32_t flags, const uint8_t* transcripts,
                                 const uint8_t* inst_points, uint32_t inst_stride, const uint8_t* common_points, const uint8_t* commitments,
                                 const uint8_t* responses, const uint8_t* weights16, uint32_t weights_stride, uint8_t* transcripts_out, int* verdicts, zkp_job** job) {
  if (p && p->use_threads())
    return submit_threaded(p, st, job, [=](int k, zkp_job** in) {
      return batch_many_submit_on(p, k, false, st, n_batches, N_each, flags, transcripts, inst_points, inst_stride, common_points, commitments, responses, weights16, weights_stride,
                                  transcripts_out, verdicts, in);
    });
  return batch_many_submit_on(p, -1, true, st, n_batches, N_each, flags, transcripts, inst_points, inst_stride, common_points, commitments, responses, weights16,
                              weights_stride, transcripts_out, verdicts, job);
}

int zkp_job_context_index(const zkp_job* j) { return j ? j->slot : -1; }
int zkp_job_done(const zkp_job* jc) {
  if (!jc) return 1;
  zkp_job* j = const_cast<zkp_job*>(jc);
  if (j->outer) {                                          // the device's thread moves the job along; the caller only looks at its state
    std::lock_guard<std::mutex> lk(j->mu);
    return j->state >= 2 ? 1 : 0;
  }
  if (j->immediate) return 1;
  Slot* s = j->pipe->slots[j->slot].get();
  std::unique_lock<std::mutex> lk(s->ctx_mu, std::try_to_lock);
  return lk.owns_lock() ? zkp_ctx_job_poll(s->ctx) : 0;
}

int zkp_job_wait(zkp_job* jp) {
  if (!jp) return ZKP_TB_BAD_STATEMENT;
  if (!jp->outer) return wait_inner(jp);
  std::unique_ptr<zkp_job> o(jp);
  int rc;
  zkp_pipe* pipe;
  {
    std::unique_lock<std::mutex> lk(o->mu);
    o->cv.wait(lk, [&] { return o->state >= 2; });
    rc = o->rc;
    pipe = o->pipe;                                        // NULL: zkp_pipe_destroy discarded the job -- the pipe and its slots are gone, only the handle is left to free
  }
  if (pipe) pipe->slots[o->slot]->reserved.store(false, std::memory_order_release);
  return rc;
}

}  // extern "C"

namespace {
// retires a job that is on a context's stream (or ran inside its submit): blocks until it is done, copies staged outputs, writes the verdicts
int wait_inner(zkp_job* jp) {
  std::unique_ptr<zkp_job> j(jp);
  if (j->immediate) return j->immediate_rc;
  Slot* s = j->pipe->slots[j->slot].get();
  if (j->use_pool) kick_all(j->pipe);                    // (use_pool = false: a per-device thread -- other threads own the other slots)
  int rc;
  {
    std::lock_guard<std::mutex> lk(s->ctx_mu);
    rc = zkp_ctx_job_wait(s->ctx);
    s->busy.store(false, std::memory_order_release);
  }
  if (rc) {
    // the device failed underneath the job: nothing it wrote may be read as a proof or as "verified"
    if (j->results) std::memset(j->results, 1, j->N);
    if (j->verdicts) for (uint32_t b = 0; b < j->K; ++b) j->verdicts[b] = ZKP_TB_VERIFICATION_FAILURE;
    return fail_pipe(j->pipe, rc, "zkp_ctx_job_wait");
  }
  for (const OutCopy& o : j->outs) par_memcpy(o.user, o.staged, o.bytes, j->use_pool);
  if (j->kind == 'P') return j->invalid_point ? ZKP_TB_INVALID_POINT : ZKP_TB_OK;
  if (j->kind == 'B') for (uint32_t b = 0; b < j->K; ++b) j->verdicts[b] = j->verdicts[b] ? ZKP_TB_VERIFICATION_FAILURE : ZKP_TB_OK;
  return ZKP_TB_OK;
}
}  // namespace

// ---- synchronous calls over every context of a pipe: contiguous proof ranges, one host thread per device ------------------
namespace {

struct Shard { uint32_t lo, hi; int slot; };

// ranges [g n / G, (g + 1) n / G) over the first G = min(#contexts, n) contexts (SURVEY 8(e) / DESIGN section 8); empty ones dropped
// -- and no more contexts than keep every range on the fused device route (>= fused_min_batch proofs each): a range below it would run the
// host-transcript route synchronously inside submit, one range after the other (N = 100 over 24 contexts: 24 ranges of 4 proofs, each a full
// GPU call chain; one range of 100 instead).  `unit` = proofs per schedulable item (a whole batch for the K-batch call).  Range g goes to slot g:
// zkp_pipe_create lays the slots out round robin over the device list (slot = k * n_devices + d), so G <= #contexts ranges reach the GPUs evenly
// before any GPU gets a second one.  The arithmetic is zkp_pipe_shard_plan (exported: the CPU suite checks it without a GPU).
std::vector<Shard> make_shards(const zkp_pipe* p, uint32_t n, uint32_t unit = 1) {
  std::vector<uint32_t> lo(p->slots.size()), hi(p->slots.size());
  const uint32_t G = zkp_pipe_shard_plan(n, unit, (uint32_t)p->slots.size(), zkp_toolbox_get_fused_min_batch(), lo.data(), hi.data());
  std::vector<Shard> out;
  for (uint32_t g = 0; g < G; ++g) out.push_back({lo[g], hi[g], (int)g});
  return out;
}

// One host thread per DEVICE submits that device's shards (each to its own context) and waits for them.  fn(shard, job**) submits;
// the returned code of a shard is what the single-context call would have returned for that range.
template <typename Submit>
int run_shards(zkp_pipe* p, const std::vector<Shard>& shards, Submit&& submit, std::vector<int>& codes) {
  codes.assign(shards.size(), ZKP_TB_OK);
  for (auto& s : p->slots)
    if (s->busy.load() || s->reserved.load()) return fail_pipe(p, ZKP_TB_PIPE_FULL, "the synchronous calls of a pipe need all of its contexts: wait for the submitted jobs first");
  std::vector<int> devs;                                    // entries of the pipe's device list that have work (an ordinal listed twice = two threads)
  for (const Shard& sh : shards) {
    const int d = p->slots[sh.slot]->group;
    if (std::find(devs.begin(), devs.end(), d) == devs.end()) devs.push_back(d);
  }
  std::vector<std::string> errs(devs.size());
  auto work = [&](size_t di) {
    std::vector<std::pair<size_t, zkp_job*>> mine;
    for (size_t i = 0; i < shards.size(); ++i) {
      if (p->slots[shards[i].slot]->group != devs[di]) continue;
      zkp_job* j = nullptr;
      const int rc = submit(shards[i], &j);
      if (rc) { codes[i] = rc; if (rc < 0) errs[di] = err_text(rc); continue; }
      mine.emplace_back(i, j);
    }
    for (auto& e : mine) {
      codes[e.first] = zkp_job_wait(e.second);
      if (codes[e.first] < 0) errs[di] = err_text(codes[e.first]);
    }
  };
  if (devs.size() <= 1) {
    if (!devs.empty()) work(0);
  } else {
    std::vector<std::thread> th;
    for (size_t di = 1; di < devs.size(); ++di) th.emplace_back(work, di);
    work(0);
    for (auto& t : th) t.join();
  }
  for (size_t di = 0; di < devs.size(); ++di)
    if (!errs[di].empty()) { std::lock_guard<std::mutex> lk(p->err_mu); p->last_error = "device " + std::to_string(p->devices[devs[di]]) + ": " + errs[di]; }
  for (int c : codes) if (c < 0) return c;                 // an infrastructure failure anywhere: nothing may be trusted
  return ZKP_TB_OK;
}

}  // namespace

extern "C" {

int zkp_pipe_prove_batch(zkp_pipe* p, const zkp_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* secrets, const uint8_t* inst_points,
                         const uint8_t* common_points, const uint8_t* entropy, uint8_t* challenges, uint8_t* responses, uint8_t* commitments) {
  if (!p || !st) return ZKP_TB_BAD_STATEMENT;
  if (N == 0) return ZKP_TB_OK;
  if (!transcripts) return ZKP_TB_BAD_STATEMENT;
  const size_t m = st->secrets.size(), nc = st->cons.size();
  const std::vector<Shard> shards = make_shards(p, N);
  const bool threads = p->devices.size() > 1;
  std::vector<int> codes;
  const int rc = run_shards(p, shards, [&](const Shard& sh, zkp_job** j) {
    const size_t a = sh.lo;
    return prove_submit_on(p, sh.slot, !threads, st, sh.hi - sh.lo, 0, transcripts + TB * a, secrets ? secrets + 32 * a * m : nullptr,
                           inst_points ? inst_points + 32 * a : nullptr, N, common_points, entropy ? entropy + 32 * a : nullptr, transcripts + TB * a,
                           challenges ? challenges + 32 * a : nullptr, responses ? responses + 32 * a * m : nullptr, commitments ? commitments + 32 * a * nc : nullptr, j);
  }, codes);
  if (rc) return rc;
  for (int c : codes) if (c) return c;                     // ZKP_TB_INVALID_POINT of any range
  return ZKP_TB_OK;
}

int zkp_pipe_verify_compact_batch(zkp_pipe* p, const zkp_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* inst_points,
                                  const uint8_t* common_points, const uint8_t* challenges, const uint8_t* responses, uint8_t* results) {
  if (!p || !st) return ZKP_TB_BAD_STATEMENT;
  if (N == 0) return ZKP_TB_OK;
  if (!transcripts || !results || !challenges) return ZKP_TB_BAD_STATEMENT;
  const size_t m = st->secrets.size();
  const std::vector<Shard> shards = make_shards(p, N);
  const bool threads = p->devices.size() > 1;
  std::vector<int> codes;
  const int rc = run_shards(p, shards, [&](const Shard& sh, zkp_job** j) {
    const size_t a = sh.lo;
    return verify_compact_submit_on(p, sh.slot, !threads, st, sh.hi - sh.lo, 0, transcripts + TB * a, inst_points ? inst_points + 32 * a : nullptr, N, common_points,
                                    challenges + 32 * a, responses ? responses + 32 * a * m : nullptr, transcripts + TB * a, results + a, j);
  }, codes);
  if (rc) return rc;
  for (int c : codes) if (c) return c;
  return ZKP_TB_OK;
}

int zkp_pipe_verify_batchable_each(zkp_pipe* p, const zkp_statement* st, uint32_t N, uint8_t* transcripts, const uint8_t* inst_points,
                                   const uint8_t* common_points, const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16, uint8_t* results) {
  if (!p || !st) return ZKP_TB_BAD_STATEMENT;
  if (N == 0) return ZKP_TB_OK;
  if (!transcripts || !results) return ZKP_TB_BAD_STATEMENT;
  const size_t m = st->secrets.size(), nc = st->cons.size();
  const std::vector<Shard> shards = make_shards(p, N);
  const bool threads = p->devices.size() > 1;
  std::vector<int> codes;
  const int rc = run_shards(p, shards, [&](const Shard& sh, zkp_job** j) {
    const size_t a = sh.lo;
    return verify_each_submit_on(p, sh.slot, !threads, st, sh.hi - sh.lo, 0, transcripts + TB * a, inst_points ? inst_points + 32 * a : nullptr, N, common_points,
                                 commitments ? commitments + 32 * a * nc : nullptr, responses ? responses + 32 * a * m : nullptr,
                                 weights16 ? weights16 + 16 * a * nc : nullptr, transcripts + TB * a, results + a, j);
  }, codes);
  if (rc) return rc;
  for (int c : codes) if (c) return c;
  return ZKP_TB_OK;
}

// One verdict for the whole batch (batch_verifier.rs:230-234): every range is a batch check of its own -- own weights, own sums of the
// static coefficients (batch_verifier.rs:173-206 per range) -- and the batch verifies iff every range does (host AND).
int zkp_pipe_batch_verify(zkp_pipe* p, const zkp_statement* st, uint32_t N, uint32_t n_transcripts, uint8_t* transcripts, const uint8_t* inst_points,
                          const uint8_t* common_points, const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16) {
  if (!p || !st) return ZKP_TB_BAD_STATEMENT;
  if (n_transcripts != N) return ZKP_TB_BATCH_SIZE_MISMATCH;           // batch_verifier.rs:72-74
  if (N == 0) return zkp_batch_verify(p->slots[0]->ctx, st, 0, 0, transcripts, inst_points, common_points, commitments, responses, weights16, 0);
  if (!transcripts) return ZKP_TB_BAD_STATEMENT;
  const size_t m = st->secrets.size(), nc = st->cons.size();
  const std::vector<Shard> shards = make_shards(p, N);
  const bool threads = p->devices.size() > 1;
  std::vector<int> codes, verdicts(shards.size(), ZKP_TB_VERIFICATION_FAILURE);
  const int rc = run_shards(p, shards, [&](const Shard& sh, zkp_job** j) {
    const size_t a = sh.lo;
    return batch_many_submit_on(p, sh.slot, !threads, st, 1, sh.hi - sh.lo, 0, transcripts + TB * a, inst_points ? inst_points + 32 * a : nullptr, N, common_points,
                                commitments ? commitments + 32 * a * nc : nullptr, responses ? responses + 32 * a * m : nullptr,
                                weights16 ? weights16 + 16 * a : nullptr, N, transcripts + TB * a, &verdicts[&sh - shards.data()], j);
  }, codes);
  if (rc) return rc;
  for (int c : codes) if (c) return c;
  for (int v : verdicts) if (v != ZKP_TB_OK) return ZKP_TB_VERIFICATION_FAILURE;
  return ZKP_TB_OK;
}

// K independent batches: whole batches go to the contexts, [g K / G, (g + 1) K / G) each
int zkp_pipe_batch_verify_many(zkp_pipe* p, const zkp_statement* st, uint32_t K, uint32_t N_each, uint32_t n_transcripts, uint8_t* transcripts,
                               const uint8_t* inst_points, const uint8_t* common_points, const uint8_t* commitments, const uint8_t* responses,
                               const uint8_t* weights16, int* verdicts) {
  if (!p || !st || !verdicts || K == 0 || N_each == 0 || (uint64_t)K * N_each > 0x7fffffffull) return ZKP_TB_BAD_STATEMENT;
  const uint32_t N = K * N_each;
  if (n_transcripts != N) return ZKP_TB_BATCH_SIZE_MISMATCH;
  if (!transcripts) return ZKP_TB_BAD_STATEMENT;
  const size_t m = st->secrets.size(), nc = st->cons.size();
  const std::vector<Shard> shards = make_shards(p, K, N_each);
  const bool threads = p->devices.size() > 1;
  std::vector<int> codes;
  const int rc = run_shards(p, shards, [&](const Shard& sh, zkp_job** j) {
    const size_t a = (size_t)sh.lo * N_each;
    return batch_many_submit_on(p, sh.slot, !threads, st, sh.hi - sh.lo, N_each, 0, transcripts + TB * a, inst_points ? inst_points + 32 * a : nullptr, N, common_points,
                                commitments ? commitments + 32 * a * nc : nullptr, responses ? responses + 32 * a * m : nullptr,
                                weights16 ? weights16 + 16 * a : nullptr, N, transcripts + TB * a, verdicts + sh.lo, j);
  }, codes);
  if (rc) return rc;
  for (int c : codes) if (c) return c;
  return ZKP_TB_OK;
}

int zkp_pipe_batch_verify_locate(zkp_pipe* p, const zkp_statement* st, uint32_t N, uint32_t n_transcripts, uint8_t* transcripts, const uint8_t* inst_points,
                                 const uint8_t* common_points, const uint8_t* commitments, const uint8_t* responses, const uint8_t* weights16, uint8_t* results) {
  if (!p || !st || !results || (N && !transcripts)) return ZKP_TB_BAD_STATEMENT;
  if (n_transcripts != N) return ZKP_TB_BATCH_SIZE_MISMATCH;
  std::memset(results, 0, N);
  const std::vector<uint8_t> saved(transcripts, transcripts + TB * (size_t)N);      // the per-proof pass starts where the batch check started
  const int rc = zkp_pipe_batch_verify(p, st, N, n_transcripts, transcripts, inst_points, common_points, commitments, responses, weights16);
  if (rc != ZKP_TB_VERIFICATION_FAILURE || N == 0) return rc;
  std::vector<uint8_t> again(saved);
  const int rc2 = zkp_pipe_verify_batchable_each(p, st, N, again.data(), inst_points, common_points, commitments, responses, nullptr, results);
  if (rc2 < 0) return rc2;
  return ZKP_TB_VERIFICATION_FAILURE;
}

}  // extern "C"
