// C-ABI of the engine (include/ehx.h): process-global space registry, key <-> dense id map
// (ANNIndex's key_to_label_/label_to_key_, embeddinghub/embeddingstore/index.h:30-32), HBM
// residency and capacity doubling (index.cc:29-32), and the kNN pipelines that chain the gfx950
// kernels.  No vector arithmetic happens on the host: if the device is unavailable every compute
// entry point fails with EHX_ENODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <random>
#include <set>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/ehx.h"
#include "ehx_env.h"
#include "ehx_kernels.h"

using namespace ehx;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      (void)hipGetLastError();                                                                \
      return fail(_e == hipErrorOutOfMemory ? EHX_ENOMEM : EHX_ENODEVICE, "%s failed: %s (%s:%d)", \
                  #expr, hipGetErrorString(_e), __FILE__, __LINE__);                          \
    }                                                                                         \
  } while (0)

struct Engine {
  std::mutex mu;
  bool inited = false;
  int device = 0;            // devices[0]: where unsharded spaces live
  std::vector<int> devices;  // ehx_init's device list: shard i of a sharded space lives on devices[i % size]
  int n_cus = 256;
  std::unordered_map<std::string, std::unique_ptr<ehx_space>> spaces;
  std::vector<std::unique_ptr<ehx_space>> graveyard;  // dropped spaces (tombstones), freed by ehx_shutdown
};
Engine& engine() {
  static Engine e;
  return e;
}

inline uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int ensure(size_t want, bool zero = false) {
    if (want <= n) return EHX_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    HIP_TRY(hipMalloc((void**)&p, want * sizeof(T)));
    if (zero) {
      // (the fill runs on the NULL stream; the spaces' streams are non-blocking, i.e. not ordered with it: wait)
      HIP_TRY(hipMemset(p, 0, want * sizeof(T)));
      HIP_TRY(hipStreamSynchronize(nullptr));
    }
    n = want;
    return EHX_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

}  // namespace

// Persistent host threads of a sharded space: worker i drives shard i + 1 (the caller's thread drives shard 0).  Round 2
// started G - 1 std::threads per CALL; these live as long as the space and sleep on a condition variable between jobs.
struct ShardWorkers {
  std::mutex run_mu;  // one job at a time
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::vector<std::thread> th;
  const std::function<int(size_t)>* job = nullptr;
  uint64_t gen = 0;
  size_t pending = 0;
  bool stop = false;
  std::vector<int> rcs;
  std::vector<std::string> errs;

  explicit ShardWorkers(size_t G) : rcs(G, 0), errs(G) {
    for (size_t i = 1; i < G; ++i) th.emplace_back([this, i] { loop(i); });
  }
  ~ShardWorkers() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_go.notify_all();
    for (auto& t : th) t.join();
  }
  void loop(size_t i) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<int(size_t)>* f;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_go.wait(lk, [&] { return stop || gen != seen; });
        if (stop) return;
        seen = gen;
        f = job;
      }
      const int rc = (*f)(i);
      std::string err = rc ? g_err : "";
      {
        std::lock_guard<std::mutex> lk(mu);
        rcs[i] = rc;
        errs[i] = std::move(err);
        if (--pending == 0) cv_done.notify_all();
      }
    }
  }

  int run(const std::function<int(size_t)>& f) {
    std::lock_guard<std::mutex> one(run_mu);
    {
      std::lock_guard<std::mutex> lk(mu);
      job = &f;
      pending = th.size();
      ++gen;
    }
    cv_go.notify_all();
    rcs[0] = f(0);
    errs[0] = rcs[0] ? g_err : "";
    {
      std::unique_lock<std::mutex> lk(mu);
      cv_done.wait(lk, [&] { return pending == 0; });
      job = nullptr;
    }
    for (size_t i = 0; i < rcs.size(); ++i)
      if (rcs[i]) {
        snprintf(g_err, sizeof(g_err), "shard %zu: %s", i, errs[i].c_str());
        return rcs[i];
      }
    return EHX_OK;
  }
};

struct ehx_space {
  std::string name;
  uint32_t dims = 0, ld = 0;
  int metric = EHX_METRIC_L2SQ;
  ehx_params params{};
  bool frozen = false;
  bool dropped = false;        // ehx_space_drop ran: HBM released, the host object stays (tombstone) so that a
                               // thread still holding the handle fails with EHX_ENOTFOUND instead of touching
                               // freed memory; reclaimed by ehx_shutdown
  bool implicit_keys = false;  // rows appended by ehx_fill_synthetic: key == decimal row id
  std::atomic<bool> poisoned{false};  // single-copy graph space (x_perm): an in-place overwrite of committed rows failed
                               // between the raw upload and the permutation — those rows sit in raw order inside a
                               // permuted store; searches and Gets refuse (EHX_EINTERNAL) instead of answering wrongly
  std::shared_mutex mu;        // writers: set/drop/reserve ; readers: knn/get
  std::mutex wmu;              // every mutator takes wmu first, then mu: writers are serialised among themselves, and
                               // a batch of fresh keys does its upload / statistics / scan copies holding wmu only —
                               // the rows land beyond the published row count — and takes mu just to publish
  hipStream_t wstream = nullptr;  // the writers' stream (uploads, row statistics, derived copies)
  hipEvent_t wev = nullptr;       // blocking-sync event: a writer waiting for its stream sleeps instead of spinning
  hipEvent_t sev[2] = {nullptr, nullptr};  // "upload out of staging half i has finished" (ping-pong staging)
                                  // inside the HIP runtime beside the threads that launch searches
  int device = 0;              // HIP device of this space's HBM state
  // Row sharding behind the C ABI (ehx_params.shards > 1): the PARENT keeps the key maps and no rows; global row g
  // lives in shard g % G at local row g / G (streamed Sets stay balanced); the shards are ordinary
  // keyless spaces, one per device of ehx_init's list, searched concurrently and merged on shard 0's device.
  bool keyless = false;            // a shard: rows are addressed by local id only, hidden from ehx_space_open
  std::vector<ehx_space*> shards;  // parent only (the shards are owned by the registry under hidden names)
  std::unique_ptr<ShardWorkers> workers;  // parent only: one persistent host thread per shard beyond the first
  hipEvent_t xev = nullptr;        // shard only: "my local top-k has reached the gather buffer" (the parent's stream waits)
  DevBuf<unsigned char> dOutPack;  // shard only: ids | distances | counts of one batch, contiguous: ONE peer copy
  DevBuf<unsigned char> dGPack;    // parent scratch on shards[0]'s device: the G packed results, one slot per shard

  // HBM-resident state
  void* dX = nullptr;        // [cap][ld] rows, fp32 or fp16 (x_half)
  int x_half = 0;            // EHX_DTYPE_F16: rows stored as IEEE binary16 (flat mode only)
  size_t esz = sizeof(float);  // bytes per stored element
  char* xrow(uint64_t id) const { return (char*)dX + id * ld * esz; }
  const float* xf32() const { return (const float*)dX; }
  float2* dRowp = nullptr;   // [cap]
  float* dInv = nullptr;     // [cap] (cosine)
  float* dMaxSumsq = nullptr;  // device scalar: largest |x|^2 ever written (certification margin, cert_margin)
  float* dXs = nullptr;      // [cap][ld] graph mode: the search copy (permuted blocks, cosine rows normalised)
  bool x_perm = false;       // graph mode, fp32 rows (round 4): the rows are stored ONCE — dX holds them in the search
                             // copy's block order, RAW; dXs is the same pointer; cosine rows are scaled by inv_norm on
                             // the fly in the kernels (GraphArgs / InsertArgs .xscale); Get undoes the permutation
  DevBuf<uint64_t> dPermIds; // rows of a batch written in place (non-contiguous ids), for launch_permute_blocks
  uint64_t cap = 0, n = 0;
  // fp16-MFMA filter scan (k_flat16.hip): unit-normalised binary16 scan copy of the rows
  bool has16 = false;          // the space keeps the fp16 scan copy (maintained on every write, whatever use16 says)
  bool use16 = false;          // ... and scans with the fp16 filter right now (ehx_space_set_scan switches it)
  __half* dX16 = nullptr;      // [cap][ld16] in the stage-blocked scan16_index layout
  float2* dRowp16 = nullptr;   // [cap]
  uint32_t ld16 = 0;
  unsigned long long* dUnsafe = nullptr;  // rows the filter cannot bound (then every scan is the fp32 scan)
  uint64_t h_unsafe = 0;
  // int8-MFMA filter scan (k_flati8.hip): per-row-scaled int8 scan copy of the unit-normalised rows
  bool has8 = false;           // the space keeps the int8 scan copy (flat spaces whose row length makes it pay)
  int8_t* dX8 = nullptr;       // [cap][ld8] in the stage-blocked scan8_index layout
  float4* dRowp8 = nullptr;    // [cap + 512] (A, B, C, D)
  float4* dTilep8 = nullptr;   // [cap/256 + 2]
  float* dTileg8 = nullptr;    // [cap/256 + 2][16] per-lane-group max |A| (k_misc.hip: rows of a tile ordered by step)
  uint8_t* dPerm8 = nullptr;   // [cap] position -> row index inside the tile
  DevBuf<uint64_t> dTileList;  // scratch of launch_make_scan8
  uint32_t ld8 = 0;
  unsigned long long* dUnsafe8 = nullptr;
  uint64_t h_unsafe8 = 0;
  uint64_t i8_min_rows = 16384;  // below this the fp16 filter serves (sample pass + cascade need a few thousand rows)
  uint32_t scan_sel = EHX_SCAN_AUTO;  // EHX_SCAN_*: what ehx_space_set_scan selected

  // graph (graph mode): imported adjacency, re-laid-out for the GPU (k_graph.hip)
  uint32_t* dAdj0 = nullptr;     // [g_n][2M]
  uint32_t* dUpStart = nullptr;  // [g_n]
  uint32_t* dUpLists = nullptr;  // [*][M]
  uint64_t g_n = 0;              // rows covered by the graph (0 = no graph)
  uint32_t g_entry = 0;
  int g_maxlevel = -1;
  DevBuf<uint32_t> dVisited;
  unsigned long long* hUncertPin = nullptr;  // pinned landing place of a batch's verdict (uncertified-query count)
  // one query per call against a small flat shard: one launch, host-visible in / out (knn_host_direct)
  char* hOnePin = nullptr;                   // host-coherent pinned: query | ids[64] | dist[64] | count | flag
  DevBuf<uint64_t> dOnePart;                 // [n_blocks][64] workgroup lists
  uint32_t* dOneTicket = nullptr;
  uint32_t one_seq = 0;
  std::atomic<uint64_t> n_one_launch{0};
  char* hSmallPin = nullptr;                 // pinned staging of small host calls: [queries | ids, distances, counts]
  DevBuf<uint64_t> dSmallOut;                // their results, one block (one device-to-host copy)
  // Host-pointer batches (ehx_knn with more than a handful of queries): every call in flight owns a SLOT — pinned
  // staging for its queries and results, device buffers for both, a copy stream — so that the upload of call i + 1
  // and the download of call i - 1 run beside the scan of call i (which alone needs scratch_mu).  One caller sees its
  // own copies in series as before; two or more callers keep the scan kernels back to back.
  struct HostSlot {
    hipStream_t st = nullptr;
    hipEvent_t in_ev = nullptr, done_ev = nullptr;
    char* pin = nullptr;
    size_t pin_bytes = 0;
    DevBuf<float> dq;
    DevBuf<unsigned char> dout;
    bool busy = false;
  };
  static constexpr int kHostSlots = 3;
  HostSlot hslot[kHostSlots];
  std::mutex hs_mu;
  std::condition_variable hs_cv;
  // Adaptation of the int8 pipeline's candidate list (i8_adapt): batches run in either scratch set, under the pipeline
  // lock or not (knn_host_direct), so the score lives under its own small mutex and the lengths are atomics — a batch
  // reads them ONCE, at its start.
  std::mutex i8_adapt_mu;
  uint32_t i8_fb_score = 0;      // recent batches that lost queries to the next engine (i8_adapt_mu)
  std::atomic<uint32_t> i8_width{kMerged8};  // width of the int8 pipeline's candidate list (doubles when batches lose
                                 // queries; create_one seeds it from the row length)
  std::atomic<uint32_t> i8_kprime_min{0};    // floor of the list's logical length k' (raised when queries lose their
                                 // certificate to a short list; flat_pass8 picks k' from the row count above it)
  std::atomic<uint32_t> i8_kprime_last{0};   // the k' the last batch ran with (statistics only)
  bool vis_dirty = false;    // a search that clears its bitmaps with a memset BEFORE the kernel leaves them marked; the
                             // visit-log mode needs them all-zero at launch
  // GPU-side insertion state
  uint64_t g_cap_rows = 0;       // rows the adjacency arrays are sized for
  uint64_t g_lists_cap = 0, g_lists_used = 0;  // upper-level lists (M ids each)
  std::vector<int32_t> h_levels;  // level of every node in the graph
  std::default_random_engine level_rng;  // hnswlib: level_generator_ (libstdc++ minstd_rand0)
  bool level_rng_seeded = false;
  uint64_t g_stale_updates = 0;  // rows overwritten in place after their insertion (no graph repair)
  DevBuf<uint32_t> dInsIds, dInsSel, dInsVislog, dItemTgt, dItemKind, dItemOff, dItemIds;
  DevBuf<uint32_t> dLinkHead, dLinkNext, dLinkCount;  // bulk build: device-side link work items (k_insert.hip)
  DevBuf<uint64_t> dLinkTouched;
  DevBuf<int32_t> dInsLevels, dItemLevel;
  unsigned long long* dGraphCounters = nullptr;  // n_dist, n_hops0, n_hops_up, n_prefetch_hit, [4..11] profile builds

  // key map (explicit keys only)
  // key <-> row id.  Their own lock (taken INSIDE mu when both are held, or alone): a streamed batch inserts its
  // 8192 keys — milliseconds of hashing and allocation — without stopping the searches, which only need mu for the
  // device arrays and the row count; the row count is published after the keys, so every id a search can return
  // already has its key.
  std::shared_mutex kmu;
  std::unordered_map<std::string, uint64_t> key_to_id;
  std::vector<std::string> id_to_key;

  // scratch for the kNN pipeline (serialised by scratch_mu)
  std::mutex scratch_mu;
  hipStream_t stream = nullptr;
  DevBuf<float> dQraw, dQ;
  DevBuf<uint64_t> dCand, dPart, dMerged, dOutIds, dGthr;
  DevBuf<float> dOutDist;
  DevBuf<uint32_t> dOutCount;
  unsigned long long* dUncert = nullptr;
  // filter scratch: fp16 queries, per-query (gamma, u, v), per-query certification flags, re-run buffers
  DevBuf<__half> dQ16;
  DevBuf<float> dQgamma, dFbQ, dFbDist, dSample;
  DevBuf<float2> dQuv;
  DevBuf<uint32_t> dUflags, dFbCnt, dFbIdx;
  DevBuf<uint64_t> dFbIds;
  unsigned long long* dUncert16 = nullptr;  // queries the filter pass could not certify
  // int8 filter scratch: everything ONE in-flight batch of the int8 pipeline owns — prepared queries, query tiles +
  // parameters, per-pass thresholds, sample scores, pools, running best list, verdict, timing events.  TWO sets: a host
  // caller's batch can be enqueued behind another caller's on the space's stream while that one still waits for its
  // verdict (knn_host_direct), so the scan kernels of consecutive batches run back to back with no host in between.
  struct I8Set {
    DevBuf<float> dQ;
    DevBuf<int8_t> dQ8;
    DevBuf<float4> dQp8;
    DevBuf<float2> dQuv;
    DevBuf<float> dThr8, dSample8;
    DevBuf<uint64_t> dPool, dMerged8;
    DevBuf<uint32_t> dI8Ctl;  // [q_rows] pool counts | [q_rows] overflow flags | [256] lock-step counters
    DevBuf<uint32_t> dUflags;
    DevBuf<uint64_t> dCnt;    // [8] epilogue counters of diagnosis builds (EHX_I8_COUNT); the set's own: nothing shared
    unsigned long long* dUncert = nullptr;
    unsigned long long* hUncertPin = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // start | scan start | scan end | all enqueued work done
    hipEvent_t verdict = nullptr;                            // blocking-sync: the verdict has landed in hUncertPin
    std::atomic<bool> ev_valid{false};
    uint64_t ev_seq = 0;     // value of ehx_space::ev_counter when ev[] was last recorded (ehx_stats: which set is newest)
    hipEvent_t ring[64][2] = {};
    uint64_t ring_count = 0;
    std::mutex mu;
  };
  I8Set i8set[2];
  std::atomic<uint64_t> ev_counter{0};
  std::atomic<uint32_t> i8_next_set{0};
  std::atomic<uint64_t> n_filter_queries{0}, n_filter_fallback{0}, n_exhaustive{0}, n_uncertified_final{0};
  std::atomic<uint64_t> n_i8_queries{0}, n_i8_fallback{0};
  float* hStage = nullptr;  // pinned staging (Set / Get / query upload)
  size_t hStageBytes = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  std::atomic<bool> ev_valid{false};
  uint64_t ev_seq = 0;
  // ring of (start, stop) event pairs around the scan kernel: per-launch durations for the roofline
  static constexpr int kRing = 64;
  hipEvent_t ring[kRing][2] = {};
  uint64_t ring_count = 0;

  // micro-batcher: concurrent small ehx_knn calls are coalesced into one device batch
  struct KnnReq {
    const float* q;
    size_t nq;
    uint32_t k;
    uint64_t* ids;
    float* dist;
    uint32_t* cnt;
    int rc = 0;
    bool done = false;
    char err[256] = "";
  };
  std::mutex bq_mu;
  std::condition_variable bq_cv;
  std::vector<KnnReq*> bq;
  bool bq_leader = false;
  std::atomic<uint64_t> n_coalesced_batches{0}, n_coalesced_queries{0};
  // write-combiner: concurrent single-row ehx_set calls (runner/copy.go: 500 goroutines per chunk) become one batch
  struct SetReq {
    const char* key;
    size_t klen;
    const float* vec;
    int rc = 0;
    bool done = false;
    char err[256] = "";
  };
  std::mutex wq_mu;
  std::condition_variable wq_cv;
  std::vector<SetReq*> wq;
  bool wq_leader = false;
  std::atomic<uint64_t> n_combined_sets{0}, n_combined_batches{0};

  // stats
  std::atomic<uint64_t> n_queries{0}, n_dist{0}, n_rerank{0}, bytes_algo{0};

  // frees every device / pinned resource (idempotent); the host-side object stays usable as a tombstone
  void release_device() {
    auto fr = [](auto*& p) {
      if (p) (void)hipFree(p);
      p = nullptr;
    };
    if (dXs == (float*)dX) dXs = nullptr;  // (single-copy graph spaces: the same allocation)
    fr(dX);
    fr(dXs);
    dPermIds.release();
    fr(dRowp);
    fr(dInv);
    fr(dMaxSumsq);
    fr(dX16);
    fr(dRowp16);
    fr(dUnsafe);
    fr(dX8);
    fr(dRowp8);
    fr(dTilep8);
    fr(dTileg8);
    fr(dPerm8);
    dTileList.release();
    fr(dUnsafe8);
    for (auto& c : i8set) {
      c.dQ.release();
      c.dQ8.release();
      c.dQp8.release();
      c.dQuv.release();
      c.dThr8.release();
      c.dSample8.release();
      c.dPool.release();
      c.dMerged8.release();
      c.dI8Ctl.release();
      c.dUflags.release();
      fr(c.dUncert);
      if (c.hUncertPin) (void)hipHostFree(c.hUncertPin);
      c.hUncertPin = nullptr;
      for (auto& e : c.ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
      }
      if (c.verdict) (void)hipEventDestroy(c.verdict);
      c.verdict = nullptr;
      c.ev_valid = false;
      for (auto& pr : c.ring)
        for (auto& e : pr) {
          if (e) (void)hipEventDestroy(e);
          e = nullptr;
        }
    }
    dGPack.release();
    dOutPack.release();
    if (xev) (void)hipEventDestroy(xev);
    xev = nullptr;
    fr(dUncert16);
    if (hUncertPin) (void)hipHostFree(hUncertPin);
    hUncertPin = nullptr;
    if (hSmallPin) (void)hipHostFree(hSmallPin);
    hSmallPin = nullptr;
    if (hOnePin) (void)hipHostFree(hOnePin);
    hOnePin = nullptr;
    dOnePart.release();
    fr(dOneTicket);
    for (auto& h : hslot) {
      if (h.pin) (void)hipHostFree(h.pin);
      h.pin = nullptr;
      h.pin_bytes = 0;
      h.dq.release();
      h.dout.release();
      if (h.in_ev) (void)hipEventDestroy(h.in_ev);
      if (h.done_ev) (void)hipEventDestroy(h.done_ev);
      if (h.st) (void)hipStreamDestroy(h.st);
      h.in_ev = h.done_ev = nullptr;
      h.st = nullptr;
    }
    fr(dAdj0);
    fr(dUpStart);
    fr(dUpLists);
    fr(dGraphCounters);
    fr(dUncert);
    dQ16.release();
    dQgamma.release();
    dSample.release();
    dFbQ.release();
    dFbDist.release();
    dQuv.release();
    dUflags.release();
    dFbCnt.release();
    dFbIdx.release();
    dFbIds.release();
    dVisited.release();
    dInsIds.release();
    dInsSel.release();
    dInsVislog.release();
    dItemTgt.release();
    dItemKind.release();
    dItemOff.release();
    dItemIds.release();
    dInsLevels.release();
    dItemLevel.release();
    dLinkHead.release();
    dLinkNext.release();
    dLinkCount.release();
    dLinkTouched.release();
    dQraw.release();
    dQ.release();
    dCand.release();
    dPart.release();
    dMerged.release();
    dGthr.release();
    dOutIds.release();
    dSmallOut.release();
    dOutDist.release();
    dOutCount.release();
    if (hStage) (void)hipHostFree(hStage);
    hStage = nullptr;
    hStageBytes = 0;
    for (auto& e : ev) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
    ev_valid = false;
    for (auto& pr : ring)
      for (auto& e : pr) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
      }
    if (stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
    if (wstream) (void)hipStreamDestroy(wstream);
    wstream = nullptr;
    if (wev) (void)hipEventDestroy(wev);
    wev = nullptr;
    for (auto& e : sev) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
    cap = 0;
    n = 0;
    g_n = 0;
  }
  ~ehx_space() { release_device(); }
};

namespace {

int ensure_stage(ehx_space* s, size_t bytes) {
  if (bytes <= s->hStageBytes) return EHX_OK;
  if (s->hStage) (void)hipHostFree(s->hStage);
  s->hStage = nullptr;
  s->hStageBytes = 0;
  HIP_TRY(hipHostMalloc((void**)&s->hStage, bytes, hipHostMallocDefault));
  s->hStageBytes = bytes;
  return EHX_OK;
}

// grow HBM arrays to hold `rows` rows (multiple of 256, zero-initialised, rowp = pad).
int grow(ehx_space* s, uint64_t rows) {
  uint64_t want = round_up(rows < 256 ? 256 : rows, 256);
  if (want <= s->cap) return EHX_OK;
  HIP_TRY(hipDeviceSynchronize());  // no search may still read the old arrays
  char* nx = nullptr;
  float2* nr = nullptr;
  float* ni = nullptr;
  HIP_TRY(hipMalloc((void**)&nx, want * s->ld * s->esz));
  hipError_t e1 = hipMalloc((void**)&nr, want * sizeof(float2));
  hipError_t e2 = hipMalloc((void**)&ni, want * sizeof(float));
  if (e1 != hipSuccess || e2 != hipSuccess) {
    (void)hipFree(nx);
    if (nr) (void)hipFree(nr);
    if (ni) (void)hipFree(ni);
    return fail(EHX_ENOMEM, "hipMalloc failed growing space '%s' to %llu rows", s->name.c_str(),
                (unsigned long long)want);
  }
  const uint64_t keep = s->n;
  if (keep) {
    HIP_TRY(hipMemcpyAsync(nx, s->dX, keep * s->ld * s->esz, hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(nr, s->dRowp, keep * sizeof(float2), hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(ni, s->dInv, keep * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
  }
  HIP_TRY(hipMemsetAsync(nx + keep * s->ld * s->esz, 0, (want - keep) * s->ld * s->esz, s->stream));
  HIP_TRY(hipMemsetAsync(ni + keep, 0, (want - keep) * sizeof(float), s->stream));
  HIP_TRY(launch_rowp_pad(nr, keep, want - keep, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (s->has16) {
    __half* nx16 = nullptr;
    float2* nr16 = nullptr;
    // (+ tail padding: the scan's DMA reads three stage blocks / two tiles of row parameters ahead)
    hipError_t e3 = hipMalloc((void**)&nx16, (want * s->ld16 + kScan16TailPadHalves) * sizeof(__half));
    hipError_t e4 = hipMalloc((void**)&nr16, (want + 2 * kTileRows16) * sizeof(float2));
    if (e3 != hipSuccess || e4 != hipSuccess) {
      if (nx16) (void)hipFree(nx16);
      if (nr16) (void)hipFree(nr16);
      (void)hipFree(nx);
      (void)hipFree(nr);
      (void)hipFree(ni);
      return fail(EHX_ENOMEM, "hipMalloc failed growing the scan copy of space '%s' to %llu rows", s->name.c_str(),
                  (unsigned long long)want);
    }
    // the scan copy is stored in whole 256-row tiles (scan16_index): copy the tiles that hold rows
    const uint64_t keep16 = round_up(keep, kTileRows16);
    if (keep) {
      HIP_TRY(hipMemcpyAsync(nx16, s->dX16, keep16 * s->ld16 * sizeof(__half), hipMemcpyDeviceToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(nr16, s->dRowp16, keep * sizeof(float2), hipMemcpyDeviceToDevice, s->stream));
    }
    HIP_TRY(hipMemsetAsync(nx16 + keep16 * s->ld16, 0,
                           ((want - keep16) * s->ld16 + kScan16TailPadHalves) * sizeof(__half), s->stream));
    HIP_TRY(launch_rowp_pad(nr16, keep, want + 2 * kTileRows16 - keep, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->dX16) (void)hipFree(s->dX16);
    if (s->dRowp16) (void)hipFree(s->dRowp16);
    s->dX16 = nx16;
    s->dRowp16 = nr16;
  }
  if (s->has8) {
    int8_t* nx8 = nullptr;
    float4* nr8 = nullptr;
    float4* nt8 = nullptr;
    const uint64_t tiles = want / kTileRows16;
    hipError_t e5 = hipMalloc((void**)&nx8, want * s->ld8 + kScan8TailPadBytes);
    hipError_t e6 = hipMalloc((void**)&nr8, (want + 2 * kTileRows16) * sizeof(float4));
    hipError_t e7 = hipMalloc((void**)&nt8, (tiles + 2) * sizeof(float4));
    float* ng8 = nullptr;
    uint8_t* np8 = nullptr;
    hipError_t e8 = hipMalloc((void**)&ng8, (tiles + 2) * 16 * sizeof(float));
    hipError_t e9 = hipMalloc((void**)&np8, want);
    if (e5 != hipSuccess || e6 != hipSuccess || e7 != hipSuccess || e8 != hipSuccess || e9 != hipSuccess) {
      if (nx8) (void)hipFree(nx8);
      if (nr8) (void)hipFree(nr8);
      if (nt8) (void)hipFree(nt8);
      if (ng8) (void)hipFree(ng8);
      if (np8) (void)hipFree(np8);
      (void)hipFree(nx);
      (void)hipFree(nr);
      (void)hipFree(ni);
      return fail(EHX_ENOMEM, "hipMalloc failed growing the int8 scan copy of space '%s' to %llu rows", s->name.c_str(),
                  (unsigned long long)want);
    }
    const uint64_t keep8 = round_up(keep, kTileRows16), keep_tiles = keep8 / kTileRows16;
    if (keep) {
      HIP_TRY(hipMemcpyAsync(nx8, s->dX8, keep8 * s->ld8, hipMemcpyDeviceToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(nr8, s->dRowp8, keep8 * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(nt8, s->dTilep8, keep_tiles * sizeof(float4), hipMemcpyDeviceToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(ng8, s->dTileg8, keep_tiles * 16 * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(np8, s->dPerm8, keep8, hipMemcpyDeviceToDevice, s->stream));
    }
    HIP_TRY(hipMemsetAsync(ng8 + keep_tiles * 16, 0, (tiles + 2 - keep_tiles) * 16 * sizeof(float), s->stream));
    HIP_TRY(launch_perm8_pad(np8, keep8, want - keep8, s->stream));
    HIP_TRY(hipMemsetAsync(nx8 + keep8 * s->ld8, 0, (want - keep8) * s->ld8 + kScan8TailPadBytes, s->stream));
    HIP_TRY(launch_rowp8_pad(nr8, keep8, want + 2 * kTileRows16 - keep8, s->stream));
    HIP_TRY(launch_tilep8_pad(nt8, keep_tiles, tiles + 2 - keep_tiles, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->dX8) (void)hipFree(s->dX8);
    if (s->dRowp8) (void)hipFree(s->dRowp8);
    if (s->dTilep8) (void)hipFree(s->dTilep8);
    if (s->dTileg8) (void)hipFree(s->dTileg8);
    if (s->dPerm8) (void)hipFree(s->dPerm8);
    s->dTileg8 = ng8;
    s->dPerm8 = np8;
    s->dX8 = nx8;
    s->dRowp8 = nr8;
    s->dTilep8 = nt8;
  }
  if (s->params.mode == EHX_MODE_GRAPH && !s->x_perm) {
    float* nxs = nullptr;
    if (hipMalloc((void**)&nxs, want * s->ld * sizeof(float)) != hipSuccess) {
      (void)hipFree(nx);
      (void)hipFree(nr);
      (void)hipFree(ni);
      return fail(EHX_ENOMEM, "hipMalloc failed growing the search copy of space '%s' to %llu rows", s->name.c_str(),
                  (unsigned long long)want);
    }
    if (keep) HIP_TRY(hipMemcpyAsync(nxs, s->dXs, keep * s->ld * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(hipMemsetAsync(nxs + keep * s->ld, 0, (want - keep) * s->ld * sizeof(float), s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->dXs) (void)hipFree(s->dXs);
    s->dXs = nxs;
  }
  if (s->dX) (void)hipFree(s->dX);
  if (s->dRowp) (void)hipFree(s->dRowp);
  if (s->dInv) (void)hipFree(s->dInv);
  s->dX = nx;
  if (s->x_perm) s->dXs = (float*)nx;  // one allocation: the rows ARE the search copy
  s->dRowp = nr;
  s->dInv = ni;
  s->cap = want;
  return EHX_OK;
}

// capacity policy of ANNIndex::set (index.cc:29-32): double when the next label hits capacity
int ensure_rows(ehx_space* s, uint64_t rows) {
  if (rows < s->cap) return EHX_OK;
  uint64_t want = s->cap ? s->cap : 256;
  while (want <= rows) want *= 2;
  return grow(s, want);
}

bool valid_space(ehx_space* s) { return s != nullptr; }

// work enqueued on stream `st` from here on starts after every search of this space that is already in flight (whatever
// stream it was given, whichever scratch set it runs in)
int wait_searches_in_flight(ehx_space* s, hipStream_t st) {
  if (s->ev_valid) HIP_TRY(hipStreamWaitEvent(st, s->ev[3], 0));
  for (auto& o : s->i8set)
    if (o.ev_valid) HIP_TRY(hipStreamWaitEvent(st, o.ev[3], 0));
  return EHX_OK;
}

// ---- graph mode: GPU-side insertion of rows [id0, id0+count) (already in HBM, stats computed) ----
// hnswlib addPoint semantics (index.cc:36).  batch == 1: strictly sequential (the reference's
// mutex-serialised order); batch > 1: rounds of concurrent inserts against the graph as it was before
// the round (the analogue of hnswlib's multi-threaded add_items).
int graph_ensure_arrays(ehx_space* s) {
  const uint32_t M0 = 2 * s->params.M;
  if (s->g_cap_rows >= s->cap && s->dAdj0) return EHX_OK;
  HIP_TRY(hipDeviceSynchronize());
  uint32_t* na = nullptr;
  uint32_t* nu = nullptr;
  HIP_TRY(hipMalloc((void**)&na, s->cap * M0 * sizeof(uint32_t)));
  HIP_TRY(hipMalloc((void**)&nu, s->cap * sizeof(uint32_t)));
  HIP_TRY(hipMemset(na, 0xFF, s->cap * M0 * sizeof(uint32_t)));
  HIP_TRY(hipMemset(nu, 0xFF, s->cap * sizeof(uint32_t)));
  if (s->dAdj0 && s->g_n) {
    HIP_TRY(hipMemcpy(na, s->dAdj0, s->g_n * M0 * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy(nu, s->dUpStart, s->g_n * sizeof(uint32_t), hipMemcpyDeviceToDevice));
  }
  HIP_TRY(hipStreamSynchronize(nullptr));  // (fills and copies above ran on the NULL stream; ours are non-blocking)
  if (s->dAdj0) (void)hipFree(s->dAdj0);
  if (s->dUpStart) (void)hipFree(s->dUpStart);
  s->dAdj0 = na;
  s->dUpStart = nu;
  s->g_cap_rows = s->cap;
  return EHX_OK;
}

int graph_ensure_lists(ehx_space* s, uint64_t lists) {
  if (lists <= s->g_lists_cap && s->dUpLists) return EHX_OK;
  uint64_t want = s->g_lists_cap ? s->g_lists_cap : 1024;
  while (want < lists) want *= 2;
  HIP_TRY(hipDeviceSynchronize());
  uint32_t* nl = nullptr;
  HIP_TRY(hipMalloc((void**)&nl, want * s->params.M * sizeof(uint32_t)));
  HIP_TRY(hipMemset(nl, 0xFF, want * s->params.M * sizeof(uint32_t)));
  if (s->dUpLists && s->g_lists_used)
    HIP_TRY(hipMemcpy(nl, s->dUpLists, s->g_lists_used * s->params.M * sizeof(uint32_t), hipMemcpyDeviceToDevice));
  HIP_TRY(hipStreamSynchronize(nullptr));
  if (s->dUpLists) (void)hipFree(s->dUpLists);
  s->dUpLists = nl;
  s->g_lists_cap = want;
  return EHX_OK;
}

// hnswlib addPoint for rows [id0, id0 + count), already in HBM.  Rounds of P rows (P = 1: hnswlib's sequential
// insertion, the oracle's graph; P > 1: the analogue of its multi-threaded add_items) — and NO host work between a
// round's kernels: the levels of all rows are drawn up front (the generator's sequence does not depend on the graph),
// so the entry point and top level of every round are known to the host in advance; the search kernel writes the new
// nodes' own lists and registers the reverse links per adjacency list on the device, the link kernel applies them.
// The whole build is enqueued on the space's stream and waited for once.  (Round 2 paid three stream synchronisations,
// a std::map regrouping on the host, five small uploads and a 5-GB bitmap memset per round.)
int graph_insert(ehx_space* s, uint64_t id0, uint64_t count, uint32_t batch) {
  if (count == 0) return EHX_OK;
  if (id0 != s->g_n)
    return fail(EHX_EUNSUPPORTED, "graph mode: rows must be inserted in id order (graph covers %llu, next row %llu)",
                (unsigned long long)s->g_n, (unsigned long long)id0);
  const uint32_t M = s->params.M, M0 = 2 * M;
  if (M0 > 64 || M < 2) return fail(EHX_EUNSUPPORTED, "M=%u not supported by the insertion kernels (2M <= 64)", M);
  uint32_t efc = s->params.ef_construction > M ? s->params.ef_construction : M;  // max(efC, M)
  if (efc > 2048) return fail(EHX_EUNSUPPORTED, "ef_construction=%u exceeds 2048", efc);
  int rc;
  if ((rc = graph_ensure_arrays(s))) return rc;
  if (!s->level_rng_seeded) {
    s->level_rng.seed((unsigned)s->params.seed);
    s->level_rng_seeded = true;
  }
  const double mult = 1.0 / log(1.0 * M);
  hipStream_t st = s->stream;
  const uint64_t end = id0 + count;
  // ---- levels of every new row (getRandomLevel: -log(U(0,1)) * mult, a fresh distribution object per draw) ----
  // Nothing of this call is committed (level generator, h_levels, g_lists_used) before every allocation it needs has
  // succeeded: a call that fails for memory leaves the space exactly as it found it, and a retry draws the same levels.
  const std::default_random_engine rng_before = s->level_rng;
  auto undo = [&](int code) {
    s->level_rng = rng_before;
    return code;
  };
  std::vector<int32_t> h_lv(count);
  std::vector<uint32_t> h_upstart(count);
  uint64_t new_lists = 0;
  int top = s->g_n ? s->g_maxlevel : 0;
  for (uint64_t i = 0; i < count; ++i) {
    std::uniform_real_distribution<double> distribution(0.0, 1.0);
    const int level = (int)(-log(distribution(s->level_rng)) * mult);
    h_lv[i] = level;
    h_upstart[i] = level > 0 ? (uint32_t)(s->g_lists_used + new_lists) : 0xFFFFFFFFu;
    new_lists += (uint64_t)level;
    if (level > top) top = level;
  }
  if ((rc = graph_ensure_lists(s, s->g_lists_used + new_lists))) return undo(rc);
  if ((rc = s->dInsLevels.ensure(count))) return undo(rc);
  // ---- round schedule ----
  const uint64_t round_cap = batch > 1 ? batch : 4096;
  // Rows of one round do not see each other, so a round never exceeds a small share of the graph it joins: 1/128, at
  // most `round_cap` rows — and 1/256 when the graph stays small (below 128 Ki nodes after this call: there every node is
  // an early node, and hnswlib-python's add_items with 64 threads is blind to only 64 / n of the graph).  Measured
  // against the oracle's sequentially built graphs at equal ef (tests/test_graph_scale.py, recall@10 over 4096 queries,
  // worst ef; profiles/r03_*_graph_scale_report*.jsonl): share 1/16 — 20 k x 768 Gaussian rows -0.009, 200 k x 768
  // -0.0015; 1/64 — -0.005 and -0.001, but 200 k x 768 STRUCTURED rows (bench.py's manifold data, where recall is
  // 0.97 and neighbours are real) -0.0052; 1/128 — structured -0.0017; 1/256 — -0.0013.  A round costs ~2 ms however
  // few rows it holds (one wave's ef_construction search is a millisecond of dependent steps), so the small shares are
  // paid once, while the graph is small: 2 M x 768 takes 18.4 s with 1/16 and 19.2 s with 1/64.
  // EHX_BUILD_DIV overrides the share (A/B runs).
  const uint64_t div_env = env().build_div;
  const uint64_t div = div_env >= 2 ? div_env : (end < (128u << 10) ? 256 : 128);
  auto round_size = [&](uint64_t g_n, uint64_t left) {
    uint64_t P = 1;
    if (batch != 1 && g_n >= 64) {
      P = g_n / div;
      if (P > round_cap) P = round_cap;
      if (P < 1) P = 1;
    }
    return P > left ? left : P;
  };
  uint64_t n_rounds = 0, max_P = 1;
  for (uint64_t g = s->g_n, pos = id0; pos < end; ++n_rounds) {
    const uint64_t P = g ? round_size(g, end - pos) : 1;
    if (P > max_P) max_P = P;
    g += P;
    pos += P;
  }
  const uint32_t vis_words = (uint32_t)((s->cap + 31) / 32);
  const uint32_t vislog_cap = 32768;
  const uint64_t max_pairs = max_P * (uint64_t)(top + 1) * M;
  if (max_pairs >= 0xFFFFFFFFull) return undo(fail(EHX_EUNSUPPORTED, "graph build: round too large"));
  // (the bitmaps are zero when allocated and every search clears the bits it set: no per-round memset)
  if ((rc = s->dVisited.ensure(max_P * vis_words, true))) return undo(rc);
  if ((rc = s->dInsVislog.ensure(max_P * (uint64_t)vislog_cap))) return undo(rc);
  if ((rc = s->dLinkHead.ensure(s->cap + s->g_lists_cap, true))) return undo(rc);  // all zero between rounds
  if ((rc = s->dLinkNext.ensure(max_pairs))) return undo(rc);
  if ((rc = s->dLinkTouched.ensure(max_pairs))) return undo(rc);
  if ((rc = s->dLinkCount.ensure(n_rounds))) return undo(rc);
  // ---- commit: from here on the rows are on their way into the graph ----
  s->g_lists_used += new_lists;
  s->h_levels.insert(s->h_levels.end(), h_lv.begin(), h_lv.end());
  // the new nodes' up_start entries and levels (their adjacency rows are still all-0xFF; nothing reaches a node
  // before the round that links it)
  HIP_TRY(hipMemcpyAsync(s->dUpStart + id0, h_upstart.data(), count * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(s->dInsLevels.p, h_lv.data(), count * sizeof(int32_t), hipMemcpyHostToDevice, st));
  if (s->vis_dirty) {  // a search that clears its bitmaps before its kernel left them marked
    HIP_TRY(hipMemsetAsync(s->dVisited.p, 0, s->dVisited.n * sizeof(uint32_t), st));
    s->vis_dirty = false;
  }
  HIP_TRY(hipMemsetAsync(s->dLinkCount.p, 0, n_rounds * sizeof(uint32_t), st));
  InsertArgs a{};  // (zeroed: a null link_head / sel switches those outputs off in the kernels)
  a.X = (s->x_half || s->x_perm) ? nullptr : s->xf32();  // (graph kernels read the search copy; X: fp32 ablation builds only)
  a.Xs = s->dXs;
  a.inv_norm = s->dInv;
  a.xscale = (s->x_perm && s->metric == EHX_METRIC_COSINE) ? s->dInv : nullptr;
  a.adj0 = s->dAdj0;
  a.up_start = s->dUpStart;
  a.up_lists = s->dUpLists;
  a.visited = s->dVisited.p;
  a.vislog = s->dInsVislog.p;
  a.new_ids = nullptr;
  a.sel = nullptr;
  a.ef = efc;
  a.dims = s->dims;
  a.ld = s->ld;
  a.M = M;
  a.M0 = M0;
  a.vis_words = vis_words;
  a.vislog_cap = vislog_cap;
  a.metric = s->metric;
  a.exclude_self = 0;
  a.head_rows = (uint32_t)s->cap;
  a.link_head = s->dLinkHead.p;
  a.link_next = s->dLinkNext.p;
  a.link_touched = (uint2*)s->dLinkTouched.p;
  // EHX_BUILD_TRACE=1: progress to stderr (costs a stream synchronisation every 128 rounds)
  const bool trace = env().build_trace;
  const auto t_build0 = std::chrono::steady_clock::now();
  uint64_t pos = id0, round = 0;
  while (pos < end) {
    if (s->g_n == 0) {  // very first node: becomes the entry point, nothing to link
      s->g_entry = (uint32_t)pos;
      s->g_maxlevel = h_lv[0];
      s->g_n = 1;
      pos += 1;
      round += 1;
      continue;
    }
    const uint64_t P = round_size(s->g_n, end - pos);
    a.id0 = (uint32_t)pos;
    a.new_levels = s->dInsLevels.p + (pos - id0);
    a.max_sel_levels = (uint32_t)s->g_maxlevel + 1;
    a.entry_point = s->g_entry;
    a.max_level = s->g_maxlevel;
    a.link_count = s->dLinkCount.p + round;
    HIP_TRY(launch_insert_search(a, (uint32_t)P, st));
    const uint64_t pairs = P * a.max_sel_levels * M;
    HIP_TRY(launch_insert_link_dev(a, (uint32_t)std::min<uint64_t>(pairs, 32768), st));
    // entry point / top level (hnswlib: a node with a higher level becomes the entry point)
    for (uint64_t i = 0; i < P; ++i) {
      const int lv = h_lv[pos - id0 + i];
      if (lv > s->g_maxlevel) {
        s->g_entry = (uint32_t)(pos + i);
        s->g_maxlevel = lv;
      }
    }
    s->g_n += P;
    pos += P;
    round += 1;
    if (trace && ((round & 127) == 0 || pos >= end)) {
      HIP_TRY(hipStreamSynchronize(st));
      fprintf(stderr, "[ehx build] round %llu of %llu, rows %llu, %.1f s\n", (unsigned long long)round,
              (unsigned long long)n_rounds, (unsigned long long)s->g_n,
              std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build0).count());
    }
  }
  HIP_TRY(hipStreamSynchronize(st));
  // A bulk build gives its scratch back: one visited bitmap per insertion in flight is cap / 8 bytes each — 5.1 GB for
  // rounds of 4096 rows on a 10 M-row index, four times what a 1024-query search batch needs (it re-allocates its own,
  // zeroed, at its first call: ~1 ms).  Streamed Sets (small calls) keep theirs.
  // (EHX_BUILD_SCRATCH_KEEP=<bytes>: what a build may keep, whatever its size — tests release at small sizes with 0)
  const long long keep_env = env().build_scratch_keep;
  const bool give_back = keep_env >= 0 ? s->dVisited.n * sizeof(uint32_t) > (unsigned long long)keep_env
                                       : (end - id0 >= 65536 && s->dVisited.n * sizeof(uint32_t) > (1ull << 30));
  if (give_back) {
    s->dVisited.release();
    s->dInsVislog.release();
    s->dLinkNext.release();
    s->dLinkTouched.release();
    s->vis_dirty = false;
  }
  return EHX_OK;
}

// ---- graph mode: hnswlib updatePoint(data, id, 1.0) for a row overwritten in place (index.cc:21-36:
// an existing key keeps its label and addPoint takes its update branch) ----
int graph_update(ehx_space* s, uint32_t id) {
  if (id >= s->g_n) return EHX_OK;
  if (s->g_entry == id && s->g_n == 1) return EHX_OK;
  const uint32_t M = s->params.M, M0 = 2 * M;
  const uint32_t efc = s->params.ef_construction > M ? s->params.ef_construction : M;
  hipStream_t st = s->stream;
  const int level = s->h_levels[id];
  int rc;
  InsertArgs a{};  // (zeroed: a null link_head / sel switches those outputs off in the kernels)
  a.X = (s->x_half || s->x_perm) ? nullptr : s->xf32();  // (graph kernels read the search copy; X: fp32 ablation builds only)
  a.Xs = s->dXs;
  a.inv_norm = s->dInv;
  a.xscale = (s->x_perm && s->metric == EHX_METRIC_COSINE) ? s->dInv : nullptr;
  a.adj0 = s->dAdj0;
  a.up_start = s->dUpStart;
  a.up_lists = s->dUpLists;
  a.ef = efc;
  a.dims = s->dims;
  a.ld = s->ld;
  a.M = M;
  a.M0 = M0;
  a.metric = s->metric;
  a.entry_point = s->g_entry;
  a.max_level = s->g_maxlevel;
  a.exclude_self = 1;
  auto read_list = [&](uint32_t node, int layer, std::vector<uint32_t>* out) -> int {
    const uint32_t width = layer == 0 ? M0 : M;
    uint32_t buf[64];
    const uint32_t* src;
    if (layer == 0) {
      src = s->dAdj0 + (size_t)node * M0;
    } else {
      uint32_t us = 0;
      HIP_TRY(hipMemcpy(&us, s->dUpStart + node, 4, hipMemcpyDeviceToHost));
      src = s->dUpLists + ((size_t)us + (uint32_t)(layer - 1)) * M;
    }
    HIP_TRY(hipMemcpy(buf, src, width * 4, hipMemcpyDeviceToHost));
    out->clear();
    for (uint32_t j = 0; j < width && buf[j] != 0xFFFFFFFFu; ++j) out->push_back(buf[j]);
    return EHX_OK;
  };
  // part 1: the one-hop neighbours re-select their links among {id} u one-hop u two-hop
  std::vector<uint32_t> one, two, h_neigh, h_off, h_cand;
  for (int layer = 0; layer <= level; ++layer) {
    if ((rc = read_list(id, layer, &one))) return rc;
    if (one.empty()) continue;
    std::set<uint32_t> sCand;
    sCand.insert(id);
    for (uint32_t o : one) {
      sCand.insert(o);
      if ((rc = read_list(o, layer, &two))) return rc;
      for (uint32_t t : two) sCand.insert(t);
    }
    h_neigh.assign(one.begin(), one.end());
    std::sort(h_neigh.begin(), h_neigh.end());
    h_neigh.erase(std::unique(h_neigh.begin(), h_neigh.end()), h_neigh.end());
    h_off.clear();
    h_cand.clear();
    for (uint32_t ng : h_neigh) {
      h_off.push_back((uint32_t)h_cand.size());
      for (uint32_t c : sCand)
        if (c != ng) h_cand.push_back(c);
    }
    h_off.push_back((uint32_t)h_cand.size());
    const uint32_t n_items = (uint32_t)h_neigh.size();
    if ((rc = s->dItemTgt.ensure(n_items))) return rc;
    if ((rc = s->dItemOff.ensure(n_items + 1))) return rc;
    if ((rc = s->dItemIds.ensure(h_cand.size() ? h_cand.size() : 1))) return rc;
    HIP_TRY(hipMemcpyAsync(s->dItemTgt.p, h_neigh.data(), n_items * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->dItemOff.p, h_off.data(), (n_items + 1) * 4, hipMemcpyHostToDevice, st));
    if (!h_cand.empty())
      HIP_TRY(hipMemcpyAsync(s->dItemIds.p, h_cand.data(), h_cand.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(launch_update_neigh(a, n_items, s->dItemTgt.p, layer, s->dItemOff.p, s->dItemIds.p, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  // part 2: repairConnectionsForUpdate = search from the entry point with the new vector, drop the
  // node itself from the results, reconnect with isUpdate semantics
  const uint32_t vis_words = (uint32_t)((s->cap + 31) / 32);
  const uint32_t vislog_cap = 32768;
  const uint32_t max_sel_levels = (uint32_t)s->g_maxlevel + 1;
  if ((rc = s->dInsIds.ensure(1))) return rc;
  if ((rc = s->dInsLevels.ensure(1))) return rc;
  if ((rc = s->dInsSel.ensure((size_t)max_sel_levels * (1 + M)))) return rc;
  if ((rc = s->dVisited.ensure(vis_words, true))) return rc;
  if ((rc = s->dInsVislog.ensure(vislog_cap))) return rc;
  const int32_t lv32 = level;
  HIP_TRY(hipMemcpyAsync(s->dInsIds.p, &id, 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(s->dInsLevels.p, &lv32, 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(s->dVisited.p, 0, vis_words * sizeof(uint32_t), st));
  a.visited = s->dVisited.p;
  a.vislog = s->dInsVislog.p;
  a.new_ids = s->dInsIds.p;
  a.new_levels = s->dInsLevels.p;
  a.sel = s->dInsSel.p;
  a.vis_words = vis_words;
  a.vislog_cap = vislog_cap;
  a.max_sel_levels = max_sel_levels;
  HIP_TRY(launch_insert_search(a, 1, st));
  std::vector<uint32_t> h_sel((size_t)max_sel_levels * (1 + M));
  HIP_TRY(hipMemcpyAsync(h_sel.data(), s->dInsSel.p, h_sel.size() * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  std::vector<uint32_t> h_tgt, h_kind, h_ioff, h_inc;
  std::vector<int32_t> h_tlevel;
  for (int l = 0; l <= level; ++l) {
    const uint32_t* o = &h_sel[(size_t)l * (1 + M)];
    const uint32_t c = o[0];
    if (c == 0) continue;  // level skipped by hnswlib: lists untouched
    h_tgt.push_back(id);
    h_tlevel.push_back(l);
    h_kind.push_back(1u);
    h_ioff.push_back((uint32_t)h_inc.size());
    for (uint32_t j = 0; j < c; ++j) h_inc.push_back(o[1 + j]);
    for (uint32_t j = 0; j < c; ++j) {  // reverse links, in hnswlib's order (selectedNeighbors order)
      h_tgt.push_back(o[1 + j]);
      h_tlevel.push_back(l);
      h_kind.push_back(2u);
      h_ioff.push_back((uint32_t)h_inc.size());
      h_inc.push_back(id);
    }
  }
  h_ioff.push_back((uint32_t)h_inc.size());
  const uint32_t n_items = (uint32_t)h_tgt.size();
  if (n_items) {
    if ((rc = s->dItemTgt.ensure(n_items))) return rc;
    if ((rc = s->dItemLevel.ensure(n_items))) return rc;
    if ((rc = s->dItemKind.ensure(n_items))) return rc;
    if ((rc = s->dItemOff.ensure(n_items + 1))) return rc;
    if ((rc = s->dItemIds.ensure(h_inc.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(s->dItemTgt.p, h_tgt.data(), n_items * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->dItemLevel.p, h_tlevel.data(), n_items * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->dItemKind.p, h_kind.data(), n_items * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->dItemOff.p, h_ioff.data(), (n_items + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->dItemIds.p, h_inc.data(), h_inc.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(launch_insert_link(a, n_items, s->dItemTgt.p, s->dItemLevel.p, s->dItemKind.p, s->dItemOff.p,
                               s->dItemIds.p, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return EHX_OK;
}

struct ScanPlan {
  uint32_t q_tiles, q_rows, n_tiles, n_chunks, tiles_per_chunk, kprime, xcd_map, grid;
};

// plan one scan pass over `n_tiles` row tiles
ScanPlan plan_scan(uint32_t nq, uint32_t n_tiles, uint32_t k, int n_cus) {
  ScanPlan p;
  p.q_tiles = (nq + kTileQ - 1) / kTileQ;
  p.q_rows = p.q_tiles * kTileQ;
  p.n_tiles = n_tiles;
  p.kprime = k + 8;  // EHX_MAX_K + 8 = 56 < kCandSlots: a compacted candidate list always has free slots
  // one persistent workgroup per CU: grid ~= n_cus, split as q_tiles x n_chunks
  uint32_t chunks = (uint32_t)n_cus / p.q_tiles;
  if (chunks < 1) chunks = 1;
  if (chunks >= 8) chunks &= ~7u;
  if (chunks > p.n_tiles) chunks = p.n_tiles ? p.n_tiles : 1;
  p.n_chunks = chunks;
  p.tiles_per_chunk = p.n_tiles ? (p.n_tiles + chunks - 1) / chunks : 0;
  p.grid = p.q_tiles * p.n_chunks;
  p.xcd_map = (p.n_chunks % 8 == 0) ? 1u : 0u;
  return p;
}

// graph pipeline: prepared queries -> zero visited bitmaps -> one-wave-per-query search
int knn_graph_locked(ehx_space* s, hipStream_t st, size_t nq, const float* d_queries, uint32_t k, uint64_t* d_ids,
                     float* d_dist, uint32_t* d_count) {
  if (s->poisoned.load())
    return fail(EHX_EINTERNAL, "graph space: an in-place overwrite failed half way (rows left in raw order); drop and rebuild it");
  if (s->g_n != s->n)
    return fail(EHX_EUNSUPPORTED,
                "graph mode: the graph covers %llu of %llu rows (rows were written while graph building was "
                "switched off, build_batch = 0xFFFFFFFF: import the graph with ehx_graph_import)",
                (unsigned long long)s->g_n, (unsigned long long)s->n);
  uint32_t ef = s->params.ef > k ? s->params.ef : k;  // searchKnn: max(ef_, k)
  if (ef > 4096) return fail(EHX_EUNSUPPORTED, "ef=%u exceeds 4096", ef);
  const uint32_t q_rows = (uint32_t)nq;
  int rc;
  if ((rc = s->dQ.ensure((size_t)q_rows * s->ld))) return rc;
  const uint32_t vis_words = (uint32_t)((s->n + 31) / 32);
  // the bitmaps are all-zero between kernels (every kernel that marks rows clears them again): zeroed once, on
  // allocation
  if ((rc = s->dVisited.ensure((size_t)nq * vis_words, true))) return rc;
  const bool use_vislog = env().graph_vislog;  // (EHX_GRAPH_VISLOG=0: per-batch memset of the bitmaps instead, A/B runs)
  // Measured (r02, batch 1024, memset inside the timed region; gpurun_out of scripts/gpu_session_n.sh): the memset
  // costs n/8 bytes per query, streamed; the log costs one store per visited row plus one RANDOM 4-byte store per row
  // when the query clears its words — ~27 ef of them.  6.25 M x 128: ef 50 log 0.41 / memset 0.47 ms, ef 200 1.11 /
  // 1.11, ef 800 4.27 / 3.89; 2 M x 768: ef 100 2.06 / 2.05, ef 400 6.99 / 6.81; small bitmaps (1 M x 128): the
  // memset is nearly free.  Hence: the log when the bitmaps are large AND the index has more than 32 000 rows per ef.
  const bool log_now = use_vislog && (size_t)nq * vis_words * sizeof(uint32_t) >= (192u << 20) &&
                       s->n >= (uint64_t)32000 * ef;
  const uint32_t vislog_cap = log_now ? 48u * ef + 256u : 0u;
  if ((rc = s->dInsVislog.ensure((size_t)nq * (vislog_cap ? vislog_cap : 1u)))) return rc;
  if (!s->dGraphCounters) {
    HIP_TRY(hipMalloc((void**)&s->dGraphCounters, kGraphCounters * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(s->dGraphCounters, 0, kGraphCounters * sizeof(unsigned long long)));
  }
  {
    int rcw = wait_searches_in_flight(s, st);
    if (rcw) return rcw;
  }
  HIP_TRY(hipEventRecord(s->ev[0], st));
  HIP_TRY(launch_prep_queries(d_queries, (uint32_t)nq, s->dims, s->ld, q_rows, s->metric, s->dQ.p, st));
  GraphArgs a;
  a.Q = s->dQ.p;
  a.X = (s->x_half || s->x_perm) ? nullptr : s->xf32();  // (graph kernels read the search copy; X: fp32 ablation builds only)
  a.Xs = s->dXs;
  a.inv_norm = s->dInv;
  a.xscale = (s->x_perm && s->metric == EHX_METRIC_COSINE) ? s->dInv : nullptr;
  a.adj0 = s->dAdj0;
  a.up_start = s->dUpStart;
  a.up_lists = s->dUpLists;
  a.visited = s->dVisited.p;
  a.vislog = s->dInsVislog.p;
  a.vislog_cap = vislog_cap;
  a.out_ids = d_ids;
  a.out_dist = d_dist;
  a.out_count = d_count;
  a.counters = s->dGraphCounters;
  a.nq = (uint32_t)nq;
  a.k = k;
  a.ef = ef;
  a.ef_cap = ef;
  a.n = (uint32_t)s->n;
  a.dims = s->dims;
  a.ld = s->ld;
  a.M = s->params.M;
  a.M0 = 2 * s->params.M;
  a.vis_words = vis_words;
  a.entry_point = s->g_entry;
  a.max_level = s->g_maxlevel;
  a.metric = s->metric;
  hipEvent_t* pr = s->ring[s->ring_count % ehx_space::kRing];
  HIP_TRY(hipEventRecord(s->ev[1], st));
  HIP_TRY(hipEventRecord(pr[0], st));
  // (inside the timed kernel region: clearing the bitmaps is part of what a batch costs, log or memset)
  if (log_now && s->vis_dirty)  // (the whole buffer: an earlier, larger batch may have marked words beyond this one's)
    HIP_TRY(hipMemsetAsync(s->dVisited.p, 0, s->dVisited.n * sizeof(uint32_t), st));
  else if (!log_now)
    HIP_TRY(hipMemsetAsync(s->dVisited.p, 0, (size_t)nq * vis_words * sizeof(uint32_t), st));
  s->vis_dirty = !log_now;
  HIP_TRY(launch_graph_search(a, st));
  HIP_TRY(hipEventRecord(pr[1], st));
  HIP_TRY(hipEventRecord(s->ev[2], st));
  s->ring_count++;
  HIP_TRY(hipEventRecord(s->ev[3], st));
  s->ev_valid = true;
  s->ev_seq = ++s->ev_counter;
  s->n_queries += nq;
  return EHX_OK;
}

// one flat pipeline: prepared queries -> scan -> merge -> canonical re-rank.
//   f16 = false: the fp32 MFMA scan (k_flat8.hip), exact on its own.
//   f16 = true : the fp16 MFMA filter scan (k_flat16.hip); per-query certification flags land in
//                s->dUflags and the caller re-runs the unflagged remainder through the fp32 scan.
int flat_pass(ehx_space* s, hipStream_t st, size_t nq, const float* d_queries, uint32_t k, uint64_t* d_ids,
              float* d_dist, uint32_t* d_count, bool f16, bool count_stats) {
  Engine& E = engine();
  // A cascade of scan passes over growing row ranges (one tile per workgroup, then x8 per pass): after
  // every pass the per-workgroup candidate lists are merged into the query's running best-64 and its
  // k'-th best key — an upper bound of the final k'-th best — becomes the threshold the next pass starts
  // from.  A pass over 8x the rows seen so far appends only ~7 k' candidates per query, so nearly every
  // tile epilogue stays on its branch-free fast path; a single pass would have every workgroup warm its
  // thresholds up from +inf (~k' ln(rows/k') appends per list).
  const uint32_t tile_rows = f16 ? kTileRows16 : kTileRows;
  const uint32_t n_tiles = (uint32_t)((s->n + tile_rows - 1) / tile_rows);
  const uint32_t lpc = f16 ? 2u : scan_lists_per_chunk();
  struct Pass {
    uint32_t tile0;
    ScanPlan plan;
  };
  // Filter scan: a SAMPLE pass first — the first 8 tiles (2048 rows) are scanned in dump mode (all scores to
  // HBM, no candidate lists) and sample_select turns them into the k'-th best score per query, so not even
  // the first real pass has to warm its lists up from +inf (which costs ~3 list compactions per list).
  constexpr uint32_t kSampleTiles = 8;
  const bool sample = f16 && n_tiles >= 256;
  std::vector<Pass> passes;
  {
    const ScanPlan whole = plan_scan((uint32_t)nq, n_tiles, k, E.n_cus);
    uint32_t done = 0;
    if (sample) {
      // the filter scan gets its first thresholds from the sample pass below, so its cascade can start
      // wide (128 tiles) and grow x16: three scan launches at 10 M rows
      uint32_t cum = 128;
      while (cum * 2 < n_tiles) {
        passes.push_back({done, plan_scan((uint32_t)nq, cum - done, k, E.n_cus)});
        done = cum;
        cum *= 16;
      }
    } else if (lpc == 2 && n_tiles >= 16 * whole.n_chunks) {
      uint32_t cum = whole.n_chunks;  // pass 0: one tile per workgroup
      while (cum * 2 < n_tiles) {
        passes.push_back({done, plan_scan((uint32_t)nq, cum - done, k, E.n_cus)});
        done = cum;
        cum *= 8;
      }
    }
    passes.push_back({done, plan_scan((uint32_t)nq, n_tiles - done, k, E.n_cus)});
  }
  ScanPlan p = passes.back().plan;  // (q_tiles, q_rows, kprime are the same for every pass)
  if (f16) {
    // the filter keeps k' = k + 22 candidates (<= 56): the certification needs the k'-th lower bound to
    // clear the k-th exact distance by the fp16 error bound, so it wants more slack than the fp32 scan
    const uint32_t kp = k + 22 > 56 ? (k + 8 > 56 ? k + 8 : 56) : k + 22;
    for (auto& ps : passes) ps.plan.kprime = kp;
    p.kprime = kp;
  } else if (s->dims > 1024) {
    // fp32 scan at large d: the certification margin grows like d * 2^-24 (cert_margin) while the gap between the
    // k-th and the k'-th best of isotropic data shrinks like ln(k'/k) / sqrt(d): widen k' (up to the 56 a
    // 64-slot list allows) so that typical data still certifies instead of falling to the exhaustive pass
    const double grow = std::exp(std::min(4.0, 5.3e-7 * std::pow((double)s->dims, 1.5)));
    uint32_t kp = (uint32_t)std::ceil((double)k * grow);
    kp = std::min<uint32_t>(56, std::max<uint32_t>(k + 8, kp));
    for (auto& ps : passes) ps.plan.kprime = kp;
    p.kprime = kp;
  }
  uint32_t lists_total = 0, grid_max = 0;  // every pass reuses the same list slots
  for (auto& ps : passes) {
    lists_total = std::max(lists_total, ps.plan.n_chunks * lpc);
    grid_max = std::max(grid_max, ps.plan.grid);
  }
  int rc;
  if ((rc = s->dQ.ensure((size_t)p.q_rows * s->ld))) return rc;
  if ((rc = s->dCand.ensure((size_t)grid_max * 512 * kCandSlots))) return rc;
  if ((rc = s->dPart.ensure((size_t)p.q_rows * lists_total * p.kprime))) return rc;
  if ((rc = s->dMerged.ensure((size_t)p.q_rows * 64))) return rc;
  if ((rc = s->dGthr.ensure((size_t)p.q_rows + 8))) return rc;  // +8: instrumentation slots of profiling builds
  if (!s->dUncert) {
    HIP_TRY(hipMalloc((void**)&s->dUncert, 2 * sizeof(unsigned long long)));  // [0] uncertified, [1] scan error
    HIP_TRY(hipMemset(s->dUncert, 0, 2 * sizeof(unsigned long long)));
  }
  if ((rc = s->dUflags.ensure(p.q_rows))) return rc;
  if (!s->dUncert16) {
    HIP_TRY(hipMalloc((void**)&s->dUncert16, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(s->dUncert16, 0, sizeof(unsigned long long)));
  }
  if (f16) {
    if ((rc = s->dQ16.ensure(scanq16_halves(p.q_rows, s->ld16)))) return rc;
    if ((rc = s->dQgamma.ensure(p.q_rows))) return rc;
    if ((rc = s->dQuv.ensure(p.q_rows))) return rc;
    if (sample && (rc = s->dSample.ensure((size_t)kSampleTiles * kTileRows16 * p.q_rows))) return rc;
  }
  // scratch buffers are shared by all callers: order this pipeline after the previous one even
  // when it was enqueued on a different stream
  {
    int rcw = wait_searches_in_flight(s, st);
    if (rcw) return rcw;
  }
  HIP_TRY(hipEventRecord(s->ev[0], st));
  HIP_TRY(launch_prep_queries(d_queries, (uint32_t)nq, s->dims, s->ld, p.q_rows, s->metric, s->dQ.p, st));
  if (f16)
    HIP_TRY(launch_prep_queries16(d_queries, (uint32_t)nq, s->dims, s->ld16, p.q_rows, s->metric, s->dQ16.p,
                                  s->dQgamma.p, s->dQuv.p, st));
  if (s->n == 0) {
    // empty space: every query returns count 0
    HIP_TRY(hipMemsetAsync(s->dMerged.p, 0xFF, (size_t)p.q_rows * 64 * sizeof(uint64_t), st));
    HIP_TRY(hipEventRecord(s->ev[1], st));
    HIP_TRY(hipEventRecord(s->ev[2], st));
  } else {
    ScanArgs a;
    a.Q = s->dQ.p;
    a.X = s->dX;
    a.x_half = (uint32_t)s->x_half;
    a.rowp = s->dRowp;
    a.cand = s->dCand.p;
    a.part = s->dPart.p;
    a.n = (uint32_t)s->n;
    a.ld = s->ld;
    a.q_tiles = p.q_tiles;
    a.kprime = p.kprime;
    a.lists_total = lists_total;
    a.err = (uint32_t*)(s->dUncert + 1);
    a.gthr = (unsigned long long*)s->dGthr.p;
    ScanArgs16 h;
    h.Q = s->dQ16.p;
    h.X = s->dX16;
    h.rowp = s->dRowp16;
    h.qgamma = s->dQgamma.p;
    h.eps = scan16_eps(s->dims);
    h.cos = s->metric == EHX_METRIC_COSINE;
    h.cand = a.cand;
    h.part = a.part;
    h.n = a.n;
    h.ld = s->ld16;
    h.q_tiles = a.q_tiles;
    h.kprime = a.kprime;
    h.lists_total = lists_total;
    h.err = a.err;
    h.gthr = a.gthr;
    auto scan = [&](const ScanPlan& pl, uint32_t tile0, uint32_t list0) -> hipError_t {
      if (f16) {
        h.tile0 = tile0;
        h.n_tiles = pl.n_tiles;
        h.n_chunks = pl.n_chunks;
        h.tiles_per_chunk = pl.tiles_per_chunk;
        h.xcd_map = pl.xcd_map;
        h.list0 = list0;
        return launch_flat_scan16(h, st);
      }
      a.tile0 = tile0;
      a.n_tiles = pl.n_tiles;
      a.n_chunks = pl.n_chunks;
      a.tiles_per_chunk = pl.tiles_per_chunk;
      a.xcd_map = pl.xcd_map;
      a.list0 = list0;
      return launch_flat_scan(a, st);
    };
    HIP_TRY(hipMemsetAsync(s->dGthr.p, 0xFF, (size_t)p.q_rows * sizeof(uint64_t), st));
    hipEvent_t* pr = s->ring[s->ring_count % ehx_space::kRing];
    HIP_TRY(hipEventRecord(s->ev[1], st));
    HIP_TRY(hipEventRecord(pr[0], st));
    if (sample) {
      ScanPlan sp = plan_scan((uint32_t)nq, kSampleTiles, k, E.n_cus);
      sp.kprime = p.kprime;
      h.dump = s->dSample.p;
      HIP_TRY(scan(sp, 0, 0));
      h.dump = nullptr;
      HIP_TRY(launch_sample_select(s->dSample.p, kSampleTiles * kTileRows16, p.q_rows, (uint32_t)nq, p.kprime,
                                   (unsigned long long*)s->dGthr.p, st));
    }
    for (size_t i = 0; i < passes.size(); ++i) {
      const bool last = i + 1 == passes.size();
      HIP_TRY(scan(passes[i].plan, passes[i].tile0, 0));
      if (last) {  // (the final merge is outside the timed scan phase)
        HIP_TRY(hipEventRecord(pr[1], st));
        HIP_TRY(hipEventRecord(s->ev[2], st));
        s->ring_count++;
      }
      HIP_TRY(launch_flat_merge(s->dPart.p, (uint32_t)nq, passes[i].plan.n_chunks * lpc, p.kprime, s->dMerged.p, st,
                                lists_total, i > 0, last ? nullptr : (unsigned long long*)s->dGthr.p));
    }
  }
  RerankArgs r;
  r.Q = s->dQ.p;
  r.X = s->dX;
  r.x_half = (uint32_t)s->x_half;
  r.inv_norm = s->dInv;
  r.merged = s->dMerged.p;
  r.out_ids = d_ids;
  r.out_dist = d_dist;
  r.out_count = d_count;
  r.n_uncertified = s->dUncert16;  // verdict counter of this pass (the caller reads and clears it)
  r.nq = (uint32_t)nq;
  r.k = k;
  r.kprime = p.kprime;
  r.n = (uint32_t)s->n;
  r.dims = s->dims;
  r.ld = s->ld;
  r.metric = s->metric;
  if (f16) r.quv = s->dQuv.p;
  r.max_sumsq = s->dMaxSumsq;
  r.uncert_flags = s->dUflags.p;
  HIP_TRY(launch_rerank(r, st));
  HIP_TRY(hipEventRecord(s->ev[3], st));
  s->ev_valid = true;
  s->ev_seq = ++s->ev_counter;
  if (count_stats) {
    s->n_queries += nq;
    s->n_dist += (uint64_t)nq * s->n;
    // SURVEY §8d brute force bytes per batch: N*d*s + B*d*4 + B*k*12 (s = bytes per element the scan reads)
    s->bytes_algo += s->n * s->dims * (uint64_t)(f16 ? 2 : s->esz) + (uint64_t)nq * s->dims * 4ull +
                     (uint64_t)nq * k * 12ull;
  }
  s->n_rerank += (uint64_t)nq * p.kprime;
  return EHX_OK;
}

// which scan engine answers first on this space right now: EHX_ENGINE_* (include/ehx.h)
int resolve_engine(const ehx_space* s) {
  if (s->params.mode != EHX_MODE_FLAT || s->scan_sel == EHX_SCAN_F32 || s->n == 0) return EHX_ENGINE_F32;
  if (s->scan_sel == EHX_SCAN_AUTO && s->has8 && s->h_unsafe8 == 0 && s->n >= s->i8_min_rows) return EHX_ENGINE_I8;
  if (s->has16 && s->h_unsafe == 0) return EHX_ENGINE_F16;
  return EHX_ENGINE_F32;
}

// The int8 filter pipeline (k_flati8.hip, k_select.hip): prepared queries -> sample pass (first thresholds) ->
// cascade of collect passes, x4 in rows, each followed by select256 (running best 256 + the next threshold) ->
// rerank256 (canonical distances of the k' = 128 best lower bounds, top-k, certificate).  Per-query verdicts land in
// s->dUflags / s->dUncert16 like those of flat_pass.
// `set`: which of the space's two scratch sets (ehx_space::I8Set) this batch runs in; the caller holds that set's mutex.
int flat_pass8(ehx_space* s, int set, hipStream_t st, size_t nq, const float* d_queries, uint32_t k, uint64_t* d_ids,
               float* d_dist, uint32_t* d_count, bool count_stats, uint32_t* kprime_used = nullptr) {
  Engine& E = engine();
  ehx_space::I8Set& sc = s->i8set[set];
  const uint32_t growth = env().i8_growth;
  const double safety = [] {
    // rank of the next pass's threshold = 256 x (share of the rows seen) x safety.  2: the last pass of a 10 M-row batch
    // runs under the 138th best of the first 27 % (about 510 rows below it overall: the list still fills to 256, the
    // certificate's floor is unchanged) instead of the 256th (950 rows): fewer alarms, fewer keys — 7.64 -> 7.46 ms per
    // batch, 0 fallbacks over 25 batches; 1.5: 7.44; 1: a query falls to the next engine (profiles/r03_e_i8_safety_sweep.jsonl)
    return env().i8_safety;  // (EHX_I8_SAFETY; 1e9: always the 256th best)
  }();
  // EHX_I8_SYNC: lock-step of the query-tile workgroups that stream one row chunk (k_flati8.hip).  "rev": by ring
  // revolution (rounds 2-4: 9 % of the scan time in round 2, 60 % on round 4's kernel); N > 0: by tile, tolerance N tiles
  // (round 5); 0 / unset: off.
  const int sync_mode = env().i8_sync;
  const bool use_sync = sync_mode != 0;
  constexpr uint32_t kSampleTiles = 8;
  // The int8 bound leaves ~60-75 rows per query ON AVERAGE that it cannot exclude from the top-10 at 10 M rows
  // (scripts/studies/int8_filter_bound.py), with a heavy tail — a query whose 10th neighbour is unusually far has
  // several times as many — and a query whose list is too short costs a whole scan by the next engine.  So the
  // threshold rank is the full width of the running list; the re-rank reads only as many candidates as it needs.
  // The list is 256 keys wide to begin with.  Longer rows leave more survivors (the bound is ~0.02 in dot units whatever
  // the dimension, while the spread of the dot products shrinks like 1/sqrt(d)): at 12.5 M x 1536 a fifth of the
  // queries needed more than 256 candidates and went to the next engine, which doubled the batch time.  A space whose
  // batches keep losing queries that way doubles its list (knn_device_locked), up to kMerged8Max.
  const uint32_t width = s->i8_width.load(std::memory_order_relaxed);  // (read once: another batch may widen it meanwhile)
  const long kprime_env = env().i8_kprime;
  // How many candidates a query keeps is what the scan's epilogue pays for (every key collected is a trip through its
  // slow path, and the waves of a workgroup wait for each other at every stage: 1.25 M x 768 collected 470 keys per
  // query, 70 % of the epilogue's tests alarmed).  The rows a query cannot exclude grow with the index (60-75 on average
  // at 10 M x 768, fewer on a shard of 1 M), so the list's LOGICAL length k' follows the row count — 128 below 4 M rows,
  // else the full width (64 was tried: no query lost at 1 - 1.25 M x 768 in 50 batches, 3 of 256 at 100 k x 768); rows
  // of 1024 dims and more always get the full width (the bound is ~1.3e-2 in dot units whatever d while the scores'
  // spread shrinks like 1/sqrt(d): 18 000 x 2048 needs its 256) — and, like the width, doubles when queries lose their
  // certificate because the list was too short (knn_device_locked).
  // Short rows need fewer still: the bound is a smaller share of the scores' spread (0.15 sigma at d = 128 against 0.36
  // at d = 768), and on short rows the hit path is what a tile's time is made of (two stages of matrix work per tile at
  // d = 128).  Measured, 46 000 queries each, 0 fallbacks (profiles/r04_p_kprime_short_rows.jsonl): 6.25 M x 128 k' 256 /
  // 128 / 64 = 1.380 / 1.277 / 1.238 ms per batch, 1 M x 128 0.447 (128) / 0.414 (64); 4 M x 384 and 10 M x 256 are fine
  // with 128 (1.762 against 1.825, 2.620 against 2.722 ms) and lose queries by the hundred with 64.
  uint32_t kp_auto = s->n >= 4000000 ? 256u : 128u;
  if (s->dims <= 128) kp_auto = 64u;
  else if (s->dims < 512) kp_auto = 128u;
  if (s->dims >= 1024) kp_auto = width;
  const uint32_t kp_want = std::max(kp_auto, s->i8_kprime_min.load(std::memory_order_relaxed));
  const uint32_t kprime = kprime_env >= 64 ? (uint32_t)std::min<long>(kprime_env, (long)width) : std::min(kp_want, width);
  s->i8_kprime_last.store(kprime, std::memory_order_relaxed);
  if (kprime_used) *kprime_used = kprime;
  const uint32_t n_tiles = (uint32_t)((s->n + kTileRows16 - 1) / kTileRows16);
  struct Pass {
    uint32_t tile0;
    ScanPlan plan;
  };
  // First pass: up to 512 tiles (131 072 rows) under a threshold taken from the sample at a LOW rank, chosen so that
  // the pass collects ~1000 keys per query (any threshold is sound, see sample_select256_kernel); then x4 in rows per
  // pass under the 256th best so far.
  const uint32_t kFirstTiles = env().i8_first_tiles;  // (EHX_I8_FIRST_TILES: sweeps of the cascade's shape on small shards)
  std::vector<Pass> passes;
  {
    uint32_t done = 0, cum = kFirstTiles;
    while ((uint64_t)cum * 2 < n_tiles) {
      passes.push_back({done, plan_scan((uint32_t)nq, cum - done, k, E.n_cus)});
      done = cum;
      cum *= growth;
    }
    passes.push_back({done, plan_scan((uint32_t)nq, n_tiles - done, k, E.n_cus)});
  }
  // rank of the threshold the select after pass i publishes for pass i + 1: the kprime-th best is always valid; while
  // only a share f of the rows has been seen, the final kprime-th best is expected near rank kprime * f of the prefix, so
  // rank kprime * f * safety (>= 16) is a much tighter threshold that is still above it — fewer keys collected, fewer
  // epilogue alarms in the middle passes.  Sound whatever happens (the certificate uses the smallest threshold ever
  // applied, qparams.w).
  auto rank_after = [&](size_t i) -> uint32_t {
    if (i + 1 >= passes.size()) return kprime;
    const double f = (double)((uint64_t)(passes[i + 1].tile0) * kTileRows16) / (double)s->n;
    return (uint32_t)std::min<double>(kprime, std::max<double>(16.0, std::ceil(kprime * f * safety)));
  };
  // The first pass runs under a threshold taken from the sample at a LOW rank, chosen for the number of keys the pass
  // should collect: a single pass has to fill the list with room to spare (2 x its logical length — every key collected
  // is a trip through the epilogue's slow path, and on a 20 000-row space 4 x meant a hit in 2.6 % of all pairs); with more
  // passes to come it only has to deliver the next threshold's rank (twice over, at least min(512, 2 k') keys: round 2
  // collected 1024 and spent more than half of the first pass in the epilogue's slow path).
  const uint64_t first_rows = (uint64_t)passes.front().plan.n_tiles * kTileRows16;
  const uint64_t first_keys_env = env().i8_first_keys;  // (EHX_I8_FIRST_KEYS: keys per query the first pass aims for)
  const uint64_t first_floor = first_keys_env ? first_keys_env : std::min<uint64_t>(512, 2ull * kprime);
  const uint64_t first_keys = std::min<uint64_t>(
      2048, passes.size() == 1 ? 2ull * kprime : std::max<uint64_t>(first_floor, 2ull * rank_after(0)));
  const uint32_t sample_rank =
      (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(8, first_keys * kSampleTiles * kTileRows16 / first_rows));
  const ScanPlan p = passes.back().plan;  // (q_tiles, q_rows are the same for every pass)
  uint32_t grid_max = 0, chunks_max = 0;
  for (auto& ps : passes) {
    grid_max = std::max(grid_max, ps.plan.grid);
    chunks_max = std::max(chunks_max, ps.plan.n_chunks);
  }
  if (chunks_max > 256) return fail(EHX_EINTERNAL, "scan plan with %u chunks", chunks_max);
  int rc;
  if ((rc = sc.dQ.ensure((size_t)p.q_rows * s->ld))) return rc;
  if ((rc = sc.dQ8.ensure(scanq8_bytes(p.q_rows, s->ld8)))) return rc;
  if ((rc = sc.dQp8.ensure(p.q_rows))) return rc;
  if ((rc = sc.dQuv.ensure(p.q_rows))) return rc;
  if ((rc = sc.dThr8.ensure(p.q_rows))) return rc;
  if ((rc = sc.dSample8.ensure((size_t)kSampleTiles * kTileRows16 * p.q_rows))) return rc;
  if ((rc = sc.dCnt.ensure(8, true))) return rc;   // (the set's own: this function runs outside the pipeline lock too)
  if ((rc = sc.dPool.ensure((size_t)p.q_rows * kPoolCap))) return rc;
  if ((rc = sc.dMerged8.ensure((size_t)p.q_rows * width))) return rc;
  if ((rc = sc.dI8Ctl.ensure((size_t)p.q_rows * 2 + kSyncWordsI8))) return rc;
  if ((rc = sc.dUflags.ensure(p.q_rows))) return rc;
  if (!sc.dUncert) {
    HIP_TRY(hipMalloc((void**)&sc.dUncert, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(sc.dUncert, 0, sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc((void**)&sc.hUncertPin, sizeof(unsigned long long), hipHostMallocDefault));
  }
  uint32_t* pool_cnt = sc.dI8Ctl.p;
  uint32_t* ovf = sc.dI8Ctl.p + p.q_rows;
  uint32_t* sync = sc.dI8Ctl.p + 2 * (size_t)p.q_rows;
  if (!sc.ev[0]) {
    for (auto& e : sc.ev) HIP_TRY(hipEventCreate(&e));
    for (auto& pr2 : sc.ring)
      for (auto& e : pr2) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventCreateWithFlags(&sc.verdict, hipEventBlockingSync | hipEventDisableTiming));
  }
  // (a caller's stream other than the space's own: searches already in flight there and here finish first)
  if ((rc = wait_searches_in_flight(s, st))) return rc;
  HIP_TRY(hipEventRecord(sc.ev[0], st));
  HIP_TRY(launch_prep_queries_i8(d_queries, (uint32_t)nq, s->dims, s->ld, s->ld8, p.q_rows, s->metric, sc.dQ.p,
                                 sc.dQ8.p, sc.dQp8.p, sc.dQuv.p, sc.dThr8.p, sc.dI8Ctl.p, st));
  ScanArgsI8 a;
  a.Q = sc.dQ8.p;
  a.X = s->dX8;
  a.rowp = s->dRowp8;
  a.tilep = s->dTilep8;
  a.tileg = s->dTileg8;
  a.perm = s->dPerm8;
  a.qparams = sc.dQp8.p;
  a.thr = sc.dThr8.p;
  a.cand = sc.dCnt.p;
  a.pool = sc.dPool.p;
  a.pool_cnt = pool_cnt;
  a.ovf = ovf;
  a.pool_cap = kPoolCap;
  a.n = (uint32_t)s->n;
  a.ld = s->ld8;
  a.q_tiles = p.q_tiles;
  auto scan = [&](const ScanPlan& pl, uint32_t tile0) -> hipError_t {
    a.tile0 = tile0;
    a.n_tiles = pl.n_tiles;
    a.n_chunks = pl.n_chunks;
    a.tiles_per_chunk = pl.tiles_per_chunk;
    a.xcd_map = pl.xcd_map;
    return launch_flat_scan_i8(a, st);
  };
  hipEvent_t* pr = sc.ring[sc.ring_count % 64];
  HIP_TRY(hipEventRecord(sc.ev[1], st));
  HIP_TRY(hipEventRecord(pr[0], st));
  {  // sample pass: lower bounds of the first 2048 rows -> thr[q] = the k'-th best of them
    ScanPlan sp = plan_scan((uint32_t)nq, kSampleTiles, k, E.n_cus);
    a.dump = sc.dSample8.p;
    a.sync = nullptr;
    HIP_TRY(scan(sp, 0));
    a.dump = nullptr;
    HIP_TRY(launch_sample_select256(sc.dSample8.p, kSampleTiles * kTileRows16, p.q_rows, (uint32_t)nq, sample_rank,
                                    sc.dThr8.p, st));
  }
  for (size_t i = 0; i < passes.size(); ++i) {
    const bool last = i + 1 == passes.size();
    a.sync = nullptr;
    if (use_sync && passes[i].plan.xcd_map && p.q_tiles > 1 && passes[i].plan.tiles_per_chunk >= 4) {
      a.sync = sync;
      a.sync_tol = sync_mode > 0 ? (uint32_t)sync_mode : 0u;
      if (i > 0) HIP_TRY(hipMemsetAsync(sync, 0, kSyncWordsI8 * sizeof(uint32_t), st));
    }
    HIP_TRY(scan(passes[i].plan, passes[i].tile0));
    if (last) {  // (the last select and the re-rank are outside the timed scan phase, like flat_pass's final merge)
      HIP_TRY(hipEventRecord(pr[1], st));
      HIP_TRY(hipEventRecord(sc.ev[2], st));
      sc.ring_count++;
    }
    HIP_TRY(launch_select256(sc.dPool.p, pool_cnt, kPoolCap, (uint32_t)nq, rank_after(i), sc.dMerged8.p, width, i > 0,
                             sc.dThr8.p, sc.dQp8.p, st));
  }
  Rerank256Args r;
  r.Q = sc.dQ.p;
  r.X = s->dX;
  r.x_half = (uint32_t)s->x_half;
  r.inv_norm = s->dInv;
  r.merged = sc.dMerged8.p;
  r.width = width;
  r.ovf = ovf;
  r.quv = sc.dQuv.p;
  r.qparams = sc.dQp8.p;
  r.max_sumsq = s->dMaxSumsq;
  r.out_ids = d_ids;
  r.out_dist = d_dist;
  r.out_count = d_count;
  r.n_uncertified = sc.dUncert;
  r.uncert_flags = sc.dUflags.p;
  r.nq = (uint32_t)nq;
  r.k = k;
  r.kprime = kprime;
  r.n = (uint32_t)s->n;
  r.dims = s->dims;
  r.ld = s->ld;
  r.metric = s->metric;
  HIP_TRY(launch_rerank256(r, st));
  if (env().i8_count) {  // diagnosis builds (-DEHX_I8_COUNT=1): the scan's epilogue counters of this batch
    unsigned long long c[8] = {0};
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(c, sc.dCnt.p, sizeof(c), hipMemcpyDeviceToHost));
    fprintf(stderr, "[i8 count] tests %llu alarms %llu row-block alarms %llu trips %llu (cumulative)\n", c[0], c[1], c[2], c[3]);
  }
  if (env().i8_debug) {  // diagnosis only: what the uncertified queries of this batch look like
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> fl(nq), ov(nq);
    std::vector<float4> qp(nq);
    std::vector<float2> uv(nq);
    std::vector<uint64_t> mg(nq * width);
    std::vector<float> od(nq * k);
    HIP_TRY(hipMemcpy(fl.data(), sc.dUflags.p, nq * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ov.data(), ovf, nq * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(qp.data(), sc.dQp8.p, nq * sizeof(float4), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(uv.data(), sc.dQuv.p, nq * sizeof(float2), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mg.data(), sc.dMerged8.p, nq * width * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(od.data(), d_dist, nq * k * 4, hipMemcpyDeviceToHost));
    int shown = 0;
    for (size_t q = 0; q < nq && shown < 6; ++q) {
      if (!fl[q]) continue;
      ++shown;
      auto S = [&](int i) { return mg[q * width + i] == ~0ull ? INFINITY : ordered_to_f32((uint32_t)(mg[q * width + i] >> 32)); };
      fprintf(stderr, "[i8 debug] q=%zu ovf=%u tmin=%g S[0]=%g S[63]=%g S[127]=%g S[255]=%g kth_dist=%g u=%g v=%g\n", q, ov[q],
              qp[q].w, S(0), S(63), S(127), S(255), od[q * k + k - 1], uv[q].x, uv[q].y);
    }
  }
  HIP_TRY(hipEventRecord(sc.ev[3], st));
  sc.ev_valid = true;
  sc.ev_seq = ++s->ev_counter;
  if (count_stats) {
    s->n_queries += nq;
    s->n_dist += (uint64_t)nq * s->n;
    // SURVEY §8d brute force bytes per batch: N*d*s + B*d*4 + B*k*12 (s = 1: the int8 scan copy)
    s->bytes_algo += s->n * (uint64_t)s->dims + (uint64_t)nq * s->dims * 4ull + (uint64_t)nq * k * 12ull;
  }
  s->n_rerank += (uint64_t)nq * kprime;
  return EHX_OK;
}

// Exhaustive canonical pass: the canonical distance of every row for `nq` queries (k_flat.hip:
// exhaustive_kernel), merged and emitted through the re-rank with the certification switched off (the keys
// are exact).  Serves (a) queries no matrix-core scan can certify and (b) requests with k > EHX_MAX_K, which
// it answers in pages of 64 results (each page keeps the keys strictly above the previous page's last).
int exhaustive_pass(ehx_space* s, hipStream_t st, size_t nq, const float* d_queries, uint32_t k, uint64_t* d_ids,
                    float* d_dist, uint32_t* d_count) {
  // rows per workgroup: 8192 when there are queries enough to fill the chip; fewer queries get smaller blocks (down to
  // one 64-row step) so that about 2048 workgroups share the shard — the keys are exact whatever the partition
  const uint32_t kRowsPerBlock =
      (uint32_t)std::min<uint64_t>(8192, std::max<uint64_t>(64, ((uint64_t)s->n * nq / 2048 + 63) / 64 * 64));
  const uint32_t n_blocks = (uint32_t)((s->n + kRowsPerBlock - 1) / kRowsPerBlock);
  const uint32_t pages = (k + 63) / 64;
  int rc;
  if ((rc = s->dQ.ensure(nq * s->ld))) return rc;
  if ((rc = s->dPart.ensure(nq * n_blocks * 64))) return rc;
  if ((rc = s->dMerged.ensure(nq * 64))) return rc;
  if ((rc = s->dUflags.ensure(nq))) return rc;
  if (pages > 1 && (rc = s->dGthr.ensure(nq + 8))) return rc;
  if (!s->dUncert16) {
    HIP_TRY(hipMalloc((void**)&s->dUncert16, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(s->dUncert16, 0, sizeof(unsigned long long)));
  }
  {
    int rcw = wait_searches_in_flight(s, st);
    if (rcw) return rcw;
  }
  HIP_TRY(hipEventRecord(s->ev[0], st));
  HIP_TRY(launch_prep_queries(d_queries, (uint32_t)nq, s->dims, s->ld, (uint32_t)nq, s->metric, s->dQ.p, st));
  HIP_TRY(hipEventRecord(s->ev[1], st));
  for (uint32_t pg = 0; pg < pages; ++pg) {
    const uint64_t* floor = pg ? s->dGthr.p : nullptr;
    HIP_TRY(launch_exhaustive(s->dQ.p, s->dX, s->x_half, s->dInv, (uint32_t)s->n, s->dims, s->ld, s->metric,
                              kRowsPerBlock, n_blocks, (uint32_t)nq, floor, s->dPart.p, st));
    HIP_TRY(launch_flat_merge(s->dPart.p, (uint32_t)nq, n_blocks, 64, s->dMerged.p, st, n_blocks));
    if (pg + 1 < pages) HIP_TRY(launch_set_floor(s->dMerged.p, (uint32_t)nq, s->dGthr.p, st));
    RerankArgs r;
    r.Q = s->dQ.p;
    r.X = s->dX;
    r.x_half = (uint32_t)s->x_half;
    r.inv_norm = s->dInv;
    r.merged = s->dMerged.p;
    r.out_ids = d_ids;
    r.out_dist = d_dist;
    r.out_count = d_count;
    r.n_uncertified = s->dUncert16;
    r.nq = (uint32_t)nq;
    r.k = std::min<uint32_t>(64, k - pg * 64);
    r.kprime = 64;
    r.n = (uint32_t)s->n;
    r.dims = s->dims;
    r.ld = s->ld;
    r.metric = s->metric;
    r.uncert_flags = s->dUflags.p;
    r.exact_keys = 1;
    r.out_stride = k;
    r.out_offset = pg * 64;
    HIP_TRY(launch_rerank(r, st));
    if (pg + 1 == pages) HIP_TRY(hipEventRecord(s->ev[2], st));
  }
  HIP_TRY(hipEventRecord(s->ev[3], st));
  s->ev_valid = true;
  s->ev_seq = ++s->ev_counter;
  s->n_dist += (uint64_t)nq * s->n * pages;
  return EHX_OK;
}

// Adaptation of the int8 list after a batch of `nq` queries that ran with logical length `kprime`, lost `n_failed`
// queries to the next engine, `n_short` of them because their candidate LIST was too short.  Called for EVERY int8
// batch, clean ones included, from both paths (knn_device_locked; knn_host_direct's pipelined stage, which used to
// skip it for clean batches: its score never decayed, and two losing batches any distance apart widened the list).
// The list is too short for this data when batches keep losing queries to the next engine — which re-reads every
// row for them, nearly a batch's worth of time however few they are (12.5 M x 1536: 13 queries in 10 batches cost
// 45 % of the run).  Only queries whose LIST was the failing part count (the re-rank flags them 2: a pool overflow,
// exact ties at the threshold or lost candidates are not cured by width, and a width, once raised, stays).  A batch of
// at least 64 queries that loses more than 2 % of them that way widens the list at once; otherwise every losing
// batch adds 4 to a score that decays by 1 per clean batch, and 8 widens (two losing batches close together).
void i8_adapt(ehx_space* s, size_t nq, size_t n_failed, size_t n_short, uint32_t kprime) {
  std::lock_guard<std::mutex> l(s->i8_adapt_mu);
  if (n_short == 0) s->i8_fb_score = s->i8_fb_score ? s->i8_fb_score - 1 : 0;
  else s->i8_fb_score += 4;
  const uint32_t width = s->i8_width.load(std::memory_order_relaxed);
  if (!((nq >= 64 && n_short * 50 > nq) || s->i8_fb_score >= 8)) return;
  // (a batch that ran with an older, shorter list than the space has by now says nothing about the present one)
  if (kprime < std::min(width, std::max(s->i8_kprime_min.load(std::memory_order_relaxed), kprime))) {
    s->i8_fb_score = 0;
    return;
  }
  if (!(width < kMerged8Max || kprime < width)) return;
  if (kprime < width) s->i8_kprime_min.store(std::min(width, 2 * kprime), std::memory_order_relaxed);  // k' first
  else {
    s->i8_width.store(width * 2, std::memory_order_relaxed);
    s->i8_kprime_min.store(width * 2, std::memory_order_relaxed);
  }
  s->i8_fb_score = 0;
  if (env().i8_trace)
    fprintf(stderr, "[ehx i8] %zu of %zu queries uncertified (%zu by a short list): candidate list now %u of %u\n", n_failed,
            nq, n_short, std::max(s->i8_kprime_min.load(), kprime), s->i8_width.load());
}

// Device pipeline of a flat space: up to three stages, each run only for the queries the previous one
// could not certify, so the answer is always the exhaustive fp32 answer in the oracle's arithmetic:
//   0. int8 matrix-core filter scan + certified re-rank      (all queries; spaces with the int8 scan copy, >= i8_min_rows)
//   1. fp16 matrix-core filter scan + certified re-rank      (what stage 0 could not certify / spaces without it)
//   2. fp32 matrix-core scan + certified re-rank              (what stage 1 could not certify / fp32-only spaces)
//   3. canonical distance of every row                        (what stage 2 could not certify: near-ties finer
//                                                              than the certification margin; kMaxExhaustive
//                                                              queries per launch group, as many groups as needed)
// One host round trip (8 bytes) per stage to read its verdict.
// i8_failed (optional): the int8 stage of this very batch has already run — in one of the scratch sets, outside the
// pipeline lock (knn_host_direct) — and left the answers of every other query in the output arrays; these queries
// (i8_short of them because their candidate list was too short) continue with the next engine.
int knn_device_locked(ehx_space* s, hipStream_t st, size_t nq, const float* d_queries, uint32_t k,
                      uint64_t* d_ids, float* d_dist, uint32_t* d_count, const std::vector<uint32_t>* i8_failed = nullptr,
                      size_t i8_short = 0, uint32_t i8_kprime_in = 0) {
  if (k == 0 || nq == 0) return EHX_OK;
  if (nq > (1u << 24)) return fail(EHX_EINVAL, "too many queries in one call: %zu", nq);
  if (s->params.mode == EHX_MODE_GRAPH) {
    // searchKnn(q, k) keeps max(ef, k) results and returns the k best (index.cc:41): any k the result list holds
    if (k > EHX_MAX_K_PAGED) return fail(EHX_EUNSUPPORTED, "graph mode: k=%u exceeds %u", k, EHX_MAX_K_PAGED);
    return knn_graph_locked(s, st, nq, d_queries, k, d_ids, d_dist, d_count);
  }
  if (k > EHX_MAX_K) {
    // beyond the candidate capacity of one scan pass: the exhaustive canonical pass, paged (exact, HBM-bound —
    // the whole shard is read once per page of 64 results and per query)
    if (k > EHX_MAX_K_PAGED) return fail(EHX_EUNSUPPORTED, "k=%u exceeds EHX_MAX_K_PAGED=%u", k, EHX_MAX_K_PAGED);
    if (s->n == 0) {
      HIP_TRY(hipMemsetAsync(d_count, 0, nq * sizeof(uint32_t), st));
      return EHX_OK;
    }
    int rc2 = exhaustive_pass(s, st, nq, d_queries, k, d_ids, d_dist, d_count);
    if (rc2) return rc2;
    s->n_queries += nq;
    s->n_exhaustive += nq;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemsetAsync(s->dUncert16, 0, sizeof(unsigned long long), st));
    return EHX_OK;
  }
  // ONE query against a small shard — the reference's own usage: one NearestNeighbor RPC, one query (server.cc:172-210;
  // BASELINE configs[0]: 10 k x 128).  The matrix-core engines are built for batches: their dozen launches (sample pass,
  // cascade, selects, re-rank) take ~0.55 ms for a single query on 10 k rows, where the exhaustive canonical pass — the
  // oracle's arithmetic over every row, exact by construction, three launches — reads the rows once.  Concurrent single
  // queries never get here alone: ehx_knn coalesces them into device batches.  (EHX_SMALL_EXACT_BYTES=0 switches it off.)
  const uint64_t small_bytes = env().small_exact_bytes;
  if (nq == 1 && s->scan_sel == EHX_SCAN_AUTO && s->n > 0 && (uint64_t)s->n * s->ld * s->esz <= small_bytes) {
    int rc2 = exhaustive_pass(s, st, nq, d_queries, k, d_ids, d_dist, d_count);
    if (rc2) return rc2;
    s->n_queries += nq;
    s->n_exhaustive += nq;
    if (s->dUncert16) HIP_TRY(hipMemsetAsync(s->dUncert16, 0, sizeof(unsigned long long), st));
    return EHX_OK;   // (no wait here: the caller's copy-back or stream order is the wait)
  }
  constexpr size_t kMaxExhaustive = 32;
  enum { kI8, kFilter, kF32, kExhaustive };
  int rc;
  size_t n_short = 0;  // of the last stage's uncertified queries: those whose candidate LIST was too short (flag 2)
  uint32_t i8_kprime = i8_kprime_in;  // the k' this batch's int8 stage ran with
  // run one stage on `subset` (nullptr = every query); *unc = global indices it could not certify
  auto stage = [&](int kind, const std::vector<uint32_t>* subset, bool count_stats, std::vector<uint32_t>* unc) -> int {
    const size_t m = subset ? subset->size() : nq;
    const float* q = d_queries;
    uint64_t* oi = d_ids;
    float* od = d_dist;
    uint32_t* oc = d_count;
    if (subset) {
      if ((rc = s->dFbQ.ensure(m * s->dims))) return rc;
      if ((rc = s->dFbIds.ensure(m * k))) return rc;
      if ((rc = s->dFbDist.ensure(m * k))) return rc;
      if ((rc = s->dFbCnt.ensure(m))) return rc;
      if ((rc = s->dFbIdx.ensure(m))) return rc;
      // (the index list comes from pageable host memory: the runtime stages it before the call returns)
      HIP_TRY(hipMemcpyAsync(s->dFbIdx.p, subset->data(), m * sizeof(uint32_t), hipMemcpyHostToDevice, st));
      HIP_TRY(launch_gather_queries(d_queries, s->dFbIdx.p, (uint32_t)m, s->dims, s->dFbQ.p, st));
      q = s->dFbQ.p;
      oi = s->dFbIds.p;
      od = s->dFbDist.p;
      oc = s->dFbCnt.p;
    }
    // (the int8 stage runs in scratch set 0 here, held for the stage and its verdict: host batches may be using both sets
    // through knn_host_direct's pipelined path at the same time)
    std::unique_lock<std::mutex> set_lock(s->i8set[0].mu, std::defer_lock);
    if (kind == kI8) set_lock.lock();
    if (kind == kExhaustive) rc = exhaustive_pass(s, st, m, q, k, oi, od, oc);
    else if (kind == kI8) rc = flat_pass8(s, 0, st, m, q, k, oi, od, oc, count_stats, &i8_kprime);
    else rc = flat_pass(s, st, m, q, k, oi, od, oc, kind == kFilter, count_stats);
    if (rc) return rc;
    unsigned long long* d_unc = kind == kI8 ? s->i8set[0].dUncert : s->dUncert16;
    const uint32_t* d_flags = kind == kI8 ? s->i8set[0].dUflags.p : s->dUflags.p;
    if (subset) {
      HIP_TRY(launch_scatter_results(oi, od, oc, s->dFbIdx.p, (uint32_t)m, k, d_ids, d_dist, d_count, st));
      HIP_TRY(hipEventRecord(s->ev[3], st));
    }
    // verdict
    unc->clear();
    // (into PINNED host memory: a copy to pageable memory goes through a staging buffer and a copy kernel)
    if (!s->hUncertPin) HIP_TRY(hipHostMalloc((void**)&s->hUncertPin, sizeof(unsigned long long), hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(s->hUncertPin, d_unc, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const unsigned long long n_unc = *s->hUncertPin;
#if defined(EHX_ABL) && EHX_ABL
    return EHX_OK;  // profiling builds with ablated (wrong-by-construction) kernels: time the first stage only
#endif
    if (n_unc == 0) return EHX_OK;
    HIP_TRY(hipMemsetAsync(d_unc, 0, sizeof(unsigned long long), st));
    std::vector<uint32_t> flags(m);
    HIP_TRY(hipMemcpyAsync(flags.data(), d_flags, m * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    n_short = 0;
    for (size_t j = 0; j < m; ++j)
      if (flags[j]) {
        unc->push_back(subset ? (*subset)[j] : (uint32_t)j);
        n_short += flags[j] == 2u;
      }
    return EHX_OK;
  };

  std::vector<uint32_t> todo, next;
  bool all = true;  // `todo` = every query
  bool counted = false;
  const int eng = resolve_engine(s);
  if (eng == EHX_ENGINE_I8) {
    if (i8_failed) {
      next = *i8_failed;
      n_short = i8_short;
    } else if ((rc = stage(kI8, nullptr, true, &next))) {
      return rc;
    }
    counted = true;
    s->n_i8_queries += nq;
    s->n_i8_fallback += next.size();
    i8_adapt(s, nq, next.size(), n_short, i8_kprime);
    if (next.empty()) return EHX_OK;
    todo.swap(next);
    all = todo.size() * 2 > nq;
  }
  if ((eng == EHX_ENGINE_I8 || eng == EHX_ENGINE_F16) && s->has16 && s->h_unsafe == 0) {
    const size_t m = all ? nq : todo.size();
    if ((rc = stage(kFilter, all ? nullptr : &todo, !counted, &next))) return rc;
    counted = true;
    s->n_filter_queries += m;
    s->n_filter_fallback += next.size();
    if (next.empty()) return EHX_OK;
    todo.swap(next);
    all = todo.size() * 2 > nq;  // most of the batch: just run it all through the fp32 scan
  }
  if ((rc = stage(kF32, all ? nullptr : &todo, !counted, &next))) return rc;
  if (next.empty()) return EHX_OK;
  todo.swap(next);
  // Whatever the matrix-core scans could not certify is answered by the exhaustive canonical pass, kMaxExhaustive
  // queries at a time (bounded scratch): an EHX_OK result is always the certified exhaustive top-k.
  std::vector<uint32_t> chunk;
  for (size_t i0 = 0; i0 < todo.size(); i0 += kMaxExhaustive) {
    chunk.assign(todo.begin() + i0, todo.begin() + std::min(todo.size(), i0 + kMaxExhaustive));
    if ((rc = stage(kExhaustive, &chunk, false, &next))) return rc;
    s->n_exhaustive += chunk.size();
    if (!next.empty()) {  // cannot happen: exact keys are never flagged
      s->n_uncertified_final += next.size();
      return fail(EHX_EINTERNAL, "%zu queries left uncertified by the exhaustive canonical pass", next.size());
    }
  }
  return EHX_OK;
}

int key_for_id(ehx_space* s, uint64_t id, std::string* out) {
  std::shared_lock<std::shared_mutex> kl(s->kmu);
  if (id < s->id_to_key.size() && !s->implicit_keys) {
    *out = s->id_to_key[id];
    return EHX_OK;
  }
  if (s->implicit_keys && id < s->n) {
    *out = std::to_string(id);
    return EHX_OK;
  }
  return EHX_ENOTFOUND;
}

int lookup_key(ehx_space* s, const char* key, size_t klen, uint64_t* id) {
  if (s->implicit_keys) {
    // decimal row id
    if (klen == 0 || klen > 20) return EHX_ENOTFOUND;
    uint64_t v = 0;
    for (size_t i = 0; i < klen; ++i) {
      if (key[i] < '0' || key[i] > '9') return EHX_ENOTFOUND;
      v = v * 10 + (uint64_t)(key[i] - '0');
    }
    if (v >= s->n) return EHX_ENOTFOUND;
    *id = v;
    return EHX_OK;
  }
  std::shared_lock<std::shared_mutex> kl(s->kmu);
  auto it = s->key_to_id.find(std::string(key, klen));
  if (it == s->key_to_id.end()) return EHX_ENOTFOUND;
  *id = it->second;
  return EHX_OK;
}


// resolve the keys of a batch to row ids (upsert: an existing key keeps its label, index.cc:21-35); a key repeated
// inside the batch resolves to one row and the LAST vector wins, as sequential Sets would leave it.  Fresh keys are
// resolved against a batch-local map and committed to key_to_id / id_to_key only after their rows are in HBM with
// statistics: a failing upload leaves the key maps and the row count untouched.
void resolve_keys(ehx_space* s, size_t n, const char* const* keys, const size_t* klens, std::vector<uint64_t>* ids,
                         uint64_t* next_out, std::vector<std::string>* new_keys) {
  ids->resize(n);
  uint64_t next = s->n;
  std::shared_lock<std::shared_mutex> kl(s->kmu);
  std::unordered_map<std::string, uint64_t> fresh;
  fresh.reserve(n);
  new_keys->reserve(n);
  for (size_t i = 0; i < n; ++i) {
    std::string k(keys[i], klens[i]);
    auto it = s->key_to_id.find(k);
    if (it != s->key_to_id.end()) {
      (*ids)[i] = it->second;
      continue;
    }
    auto f = fresh.try_emplace(k, next);  // (one hash for "seen in this batch?" and the insert)
    if (!f.second) {
      (*ids)[i] = f.first->second;
      continue;
    }
    (*ids)[i] = next;
    new_keys->push_back(std::move(k));
    ++next;
  }
  *next_out = next;
}

// =====================================================================================================
// Row-sharded spaces behind the C ABI (ehx_params.shards = G > 1).
// One process drives the G devices of ehx_init's list: global row g lives in shard g % G at local row g / G; a
// search runs on every shard concurrently (one host thread and one stream per shard), each shard's local top-k
// (k * 12 + 4 bytes per query) is copied peer-to-peer over xGMI into one gather buffer on shard 0's device and
// merge_lists_kernel — the same kernel the multi-process path uses behind its RCCL all-gather — turns local rows
// into global ids (local * G + shard) and merges by (distance, id).  No other exchange step exists.
// =====================================================================================================
inline bool is_parent(const ehx_space* s) { return !s->shards.empty(); }

// f(i) for every shard, each on its own persistent thread (shard 0 on the caller's); first failure wins.  A shard that
// was dropped meanwhile (ehx_space_drop marks the parent first, so this only guards a handle that outlived its space)
// answers EHX_ENOTFOUND instead of touching released buffers.
int for_each_shard(ehx_space* p, const std::function<int(size_t)>& f) {
  if (!p->workers) return fail(EHX_EINTERNAL, "space '%s' has no shard workers", p->name.c_str());
  return p->workers->run([&](size_t i) -> int {
    if (p->shards[i]->dropped) return fail(EHX_ENOTFOUND, "Not found");
    return f(i);
  });
}

int write_rows_locked_fwd(ehx_space* s, size_t n, const std::vector<uint64_t>& ids, uint64_t next, const float* vecs);
int fill_synthetic_locked(ehx_space* s, uint64_t seed, uint64_t row0, uint64_t n_rows, int normalize, uint64_t stride);

// parent locked exclusively by the caller
int sharded_set_batch(ehx_space* p, size_t n, const char* const* keys, const size_t* klens, const float* vecs) {
  if (p->frozen) return fail(EHX_EIMMUTABLE, "Cannot write to immutable space");
  if (p->implicit_keys) return fail(EHX_EINVAL, "space '%s' holds synthetic rows with implicit keys", p->name.c_str());
  const uint64_t G = p->shards.size();
  std::vector<uint64_t> ids;
  std::vector<std::string> new_keys;
  uint64_t next = 0;
  resolve_keys(p, n, keys, klens, &ids, &next, &new_keys);
  std::vector<std::vector<uint64_t>> lids(G);
  std::vector<std::vector<float>> rows(G);
  for (size_t i = 0; i < n; ++i) {
    const uint64_t sh = ids[i] % G;
    lids[sh].push_back(ids[i] / G);
    rows[sh].insert(rows[sh].end(), vecs + i * p->dims, vecs + (i + 1) * p->dims);
  }
  std::vector<uint64_t> before(G);
  for (size_t i = 0; i < G; ++i) before[i] = p->shards[i]->n;
  // Capacity first, on every shard, before any shard writes a row: the allocation that fails a batch half way (an
  // out-of-memory while one shard grows) then fails it before anything changed — graph shards cannot take rows back
  // once they are linked.
  int rc = for_each_shard(p, [&](size_t i) -> int {
    if (lids[i].empty()) return EHX_OK;
    ehx_space* c = p->shards[i];
    std::lock_guard<std::mutex> cg(c->wmu);
    std::unique_lock<std::shared_mutex> wl(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t next_local = (next + G - 1 - i) / G;
    int r = ensure_rows(c, next_local);
    if (!r && c->params.mode == EHX_MODE_GRAPH) r = graph_ensure_arrays(c);
    return r;
  });
  if (rc) return rc;
  rc = for_each_shard(p, [&](size_t i) -> int {
    if (lids[i].empty()) return EHX_OK;
    ehx_space* c = p->shards[i];
    std::lock_guard<std::mutex> cg(c->wmu);
    std::unique_lock<std::shared_mutex> wl(c->mu);
    const uint64_t next_local = (next + G - 1 - i) / G;  // globals below `next` that belong to shard i
    return write_rows_locked_fwd(c, lids[i].size(), lids[i], next_local, rows[i].data());
  });
  if (rc) {
    // A failing shard (e.g. out of memory while growing) must not leave the others ahead of the parent: their published
    // row counts go back to what they were, so no search returns a global id the parent has no key for.  (Rows of
    // EXISTING keys that the batch rewrote on the shards that succeeded stay rewritten — a failed batch may have
    // applied part of its updates, as a failed sequence of single Sets would; graph shards keep the nodes they linked.)
    const std::string msg = g_err;
    for (size_t i = 0; i < G; ++i) {
      ehx_space* c = p->shards[i];
      std::lock_guard<std::mutex> cg(c->wmu);
      std::unique_lock<std::shared_mutex> wl(c->mu);
      if (c->n > before[i] && c->params.mode != EHX_MODE_GRAPH) c->n = before[i];
    }
    snprintf(g_err, sizeof(g_err), "%s", msg.c_str());
    return rc;
  }
  const uint64_t old_n = p->n;
  {
    std::unique_lock<std::shared_mutex> kl(p->kmu);
    for (size_t i = 0; i < new_keys.size(); ++i) p->key_to_id.emplace(new_keys[i], old_n + i);
    for (auto& k : new_keys) p->id_to_key.push_back(std::move(k));
  }
  p->n = next;
  return EHX_OK;
}

int sharded_fill_synthetic(ehx_space* p, uint64_t seed, uint64_t row0, uint64_t n_rows, int normalize) {
  if (p->frozen) return fail(EHX_EIMMUTABLE, "Cannot write to immutable space");
  if (!p->implicit_keys && p->n != 0) return fail(EHX_EINVAL, "space '%s' already holds keyed rows", p->name.c_str());
  const uint64_t G = p->shards.size(), n0 = p->n;
  int rc = for_each_shard(p, [&](size_t i) -> int {
    // globals n0 .. n0+n_rows-1 with g % G == i: g0, g0 + G, ...; generator row of global g = row0 + (g - n0)
    const uint64_t g0 = n0 + ((i + G - n0 % G) % G);
    if (g0 >= n0 + n_rows) return EHX_OK;
    const uint64_t cnt = (n0 + n_rows - 1 - g0) / G + 1;
    ehx_space* c = p->shards[i];
    std::lock_guard<std::mutex> cg(c->wmu);
    std::unique_lock<std::shared_mutex> wl(c->mu);
    return fill_synthetic_locked(c, seed, row0 + (g0 - n0), cnt, normalize, G);
  });
  if (rc) return rc;
  p->implicit_keys = true;
  p->n += n_rows;
  return EHX_OK;
}

// queries are on the host (d_queries == nullptr) or on device `qdev`; outputs likewise.  Parent locked shared.
// Per batch and shard: the queries in, the shard's own pipeline, ONE peer copy of its packed local top-k
// (ids | distances | counts: 12 k + 4 bytes per query) into its slot of the gather buffer on shard 0's device, and an
// event; the parent's stream waits for the G events (no host synchronisation per shard), merges, and hands the result
// over.  k up to 1024 like an unsharded space (every shard pages its own exhaustive pass beyond 48; the merge walks
// the lists beyond 64).
int sharded_knn(ehx_space* p, size_t nq, const float* h_queries, const float* d_queries, int qdev, uint32_t k,
                uint64_t* out_ids, float* out_dist, uint32_t* out_count, bool out_on_device, hipStream_t caller_stream) {
  if (k == 0 || nq == 0) return EHX_OK;
  if (k > 1024) return fail(EHX_EUNSUPPORTED, "k=%u exceeds 1024", k);
  const size_t G = p->shards.size();
  const int home = p->shards[0]->device;
  std::lock_guard<std::mutex> sl(p->scratch_mu);
  int rc;
  HIP_TRY(hipSetDevice(home));
  const size_t o_dist = nq * k * sizeof(uint64_t), o_cnt = o_dist + nq * k * sizeof(float);
  const size_t P = (o_cnt + nq * sizeof(uint32_t) + 15) / 16 * 16;  // one shard's packed result
  if ((rc = p->dGPack.ensure(G * P))) return rc;
  if ((rc = p->dOutIds.ensure(nq * k))) return rc;
  if ((rc = p->dOutDist.ensure(nq * k))) return rc;
  if ((rc = p->dOutCount.ensure(nq))) return rc;
  if (d_queries) {  // the caller's stream produced the queries: they must be complete before the shards read them
    HIP_TRY(hipSetDevice(qdev));
    HIP_TRY(hipStreamSynchronize(caller_stream));
  }
  const size_t qbytes = nq * p->dims * sizeof(float);
  rc = for_each_shard(p, [&](size_t i) -> int {
    ehx_space* c = p->shards[i];
    std::shared_lock<std::shared_mutex> rl(c->mu);
    std::lock_guard<std::mutex> cl(c->scratch_mu);
    HIP_TRY(hipSetDevice(c->device));
    int r;
    if ((r = c->dQraw.ensure(nq * c->dims))) return r;
    if ((r = c->dOutPack.ensure(P))) return r;
    if (!c->xev) HIP_TRY(hipEventCreateWithFlags(&c->xev, hipEventDisableTiming));
    if (d_queries) HIP_TRY(hipMemcpyPeerAsync(c->dQraw.p, c->device, d_queries, qdev, qbytes, c->stream));
    else HIP_TRY(hipMemcpyAsync(c->dQraw.p, h_queries, qbytes, hipMemcpyHostToDevice, c->stream));
    unsigned char* pk = c->dOutPack.p;
    if ((r = knn_device_locked(c, c->stream, nq, c->dQraw.p, k, (uint64_t*)pk, (float*)(pk + o_dist),
                               (uint32_t*)(pk + o_cnt))))
      return r;
    // the one exchange step
    HIP_TRY(hipMemcpyPeerAsync(p->dGPack.p + i * P, home, pk, c->device, P, c->stream));
    HIP_TRY(hipEventRecord(c->xev, c->stream));
    return EHX_OK;
  });
  if (rc) {
    // a shard failed: the others may still be writing into the gather buffer and their own scratch — drain them
    // before the error leaves (the next call reuses both); g_err keeps the failing shard's message
    for (ehx_space* c : p->shards)
      if (hipSetDevice(c->device) == hipSuccess) (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
    return rc;
  }
  HIP_TRY(hipSetDevice(home));
  for (size_t i = 0; i < G; ++i) HIP_TRY(hipStreamWaitEvent(p->stream, p->shards[i]->xev, 0));
  const unsigned char* gp = p->dGPack.p;
  HIP_TRY(launch_merge_lists((const uint64_t*)gp, (const float*)(gp + o_dist), (const uint32_t*)(gp + o_cnt),
                             (uint32_t)nq, k, (uint32_t)G, p->dOutIds.p, p->dOutDist.p, p->dOutCount.p, p->stream, P, P,
                             P, (uint64_t)G, 1));
  if (out_on_device) {
    HIP_TRY(hipMemcpyPeerAsync(out_ids, qdev, p->dOutIds.p, home, nq * k * sizeof(uint64_t), p->stream));
    HIP_TRY(hipMemcpyPeerAsync(out_dist, qdev, p->dOutDist.p, home, nq * k * sizeof(float), p->stream));
    HIP_TRY(hipMemcpyPeerAsync(out_count, qdev, p->dOutCount.p, home, nq * sizeof(uint32_t), p->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(out_ids, p->dOutIds.p, nq * k * sizeof(uint64_t), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(out_dist, p->dOutDist.p, nq * k * sizeof(float), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(out_count, p->dOutCount.p, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, p->stream));
  }
  HIP_TRY(hipStreamSynchronize(p->stream));  // (the shards' scratch may be reused by the next call: all of it is done)
  p->n_queries += nq;
  return EHX_OK;
}

}  // namespace

extern "C" {

int ehx_abi_version(void) { return EHX_ABI_VERSION; }
const char* ehx_last_error(void) { return g_err; }

int ehx_init(const int* device_ids, int n_devices) {
  Engine& E = engine();
  std::lock_guard<std::mutex> lk(E.mu);
  if (E.inited) return EHX_OK;
  (void)env();   // every environment knob is read here, once (ehx_env.h)
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return fail(EHX_ENODEVICE, "no HIP device available (%s); the engine has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  }
  std::vector<int> devs;
  if (device_ids && n_devices > 0) devs.assign(device_ids, device_ids + n_devices);
  else devs.push_back(0);
  int n_cus = 0;
  for (int dev : devs) {
    if (dev < 0 || dev >= count) return fail(EHX_EINVAL, "device id %d out of range (0..%d)", dev, count - 1);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(EHX_ENODEVICE, "device %d is %s; this engine is built for gfx950 only", dev, prop.gcnArchName);
    if (!n_cus) n_cus = prop.multiProcessorCount;
  }
  // The shards of one space exchange their local top-k lists by peer copies over xGMI: every ordered pair of the
  // devices is opened EXPLICITLY, and a pair that cannot be opened fails the call, naming the two devices — a node whose
  // links are down must not quietly run its exchange step through host memory (EHX_ALLOW_NO_PEER=1 accepts it: the
  // copies then stage through the host, correct but slow).
  const bool strict = !env().allow_no_peer;
  for (int a : devs)
    for (int b : devs) {
      if (a == b) continue;
      int can = 0;
      hipError_t pe = hipDeviceCanAccessPeer(&can, a, b);
      if (pe == hipSuccess && can) {
        pe = hipSetDevice(a);
        if (pe == hipSuccess) pe = hipDeviceEnablePeerAccess(b, 0);
        if (pe == hipErrorPeerAccessAlreadyEnabled) pe = hipSuccess;
      } else if (pe == hipSuccess) {
        pe = hipErrorPeerAccessUnsupported;
      }
      (void)hipGetLastError();
      if (pe != hipSuccess && strict)
        return fail(EHX_ENODEVICE, "device %d cannot open peer access to device %d (%s): the shards' exchange step needs "
                                   "it (EHX_ALLOW_NO_PEER=1 stages the copies through the host instead)",
                    a, b, hipGetErrorString(pe));
    }
  HIP_TRY(hipSetDevice(devs[0]));
  E.devices = devs;
  E.device = devs[0];
  E.n_cus = n_cus > 0 ? n_cus : 256;
  E.inited = true;
  return EHX_OK;
}

int ehx_shutdown(void) {
  Engine& E = engine();
  std::lock_guard<std::mutex> lk(E.mu);
  if (E.inited) (void)hipDeviceSynchronize();
  E.spaces.clear();
  E.graveyard.clear();  // (handles of dropped spaces die here: no call may be in flight during shutdown)
  return EHX_OK;
}

// one space on one device (E.mu held by the caller); parent = true: the key-map holder of a sharded space (no HBM)
static int create_one(Engine& E, const std::string& nm, uint32_t dims, int metric, int dtype, const ehx_params* params,
                      int device, bool keyless, bool parent, ehx_space** out) {
  int rc;
  HIP_TRY(hipSetDevice(device));
  if (E.spaces.count(nm)) return fail(EHX_EEXISTS, "space '%s' already exists", nm.c_str());
  std::unique_ptr<ehx_space> s(new ehx_space);
  s->name = nm;
  s->device = device;
  s->keyless = keyless;
  s->dims = dims;
  s->ld = (uint32_t)round_up(dims, kBK);
  s->metric = metric;
  s->x_half = dtype == EHX_DTYPE_F16;
  s->esz = s->x_half ? 2 : sizeof(float);
  if (params) s->params = *params;
  {
    // Single-copy storage of a graph space's rows (round 4): the search copy — rows in the 4 x 4 block order the graph
    // kernels read — is the ONLY copy.  Free for L2^2 and inner product (their search copy was a permuted duplicate).
    // A cosine space's search copy used to hold the NORMALISED rows; with one copy the kernels form x * inv_norm on the
    // fly — bit-identical results, half the HBM (10 M x 768: 61 -> 31 GB).  The scale must be REQUESTED BEFORE the
    // row's ring loads (ehx_kernels.h, wave_group_dists_t): sunk below them it is the youngest load when the first
    // product needs it and the wait drains the ring — that cost 13 % at 2 M x 768 (profiles/r04_r_*); requested first
    // the cost is 2 % (profiles/r04_s_*, r04_t_*).  EHX_GRAPH_TWO_COPIES=1: raw rows + search copy as in rounds 1-3.
    s->x_perm = params && params->mode == EHX_MODE_GRAPH && !s->x_half && !env().graph_two_copies;
  }
  if (parent) s->params.shards = params->shards;
  if (s->params.mode != EHX_MODE_FLAT && s->params.mode != EHX_MODE_GRAPH)
    return fail(EHX_EINVAL, "unknown mode %u", s->params.mode);
  if (s->params.M == 0) s->params.M = 16;
  if (s->params.ef_construction == 0) s->params.ef_construction = 200;
  if (s->params.ef == 0) s->params.ef = 10;
  if (s->params.seed == 0) s->params.seed = 100;
  if (s->params.mode == EHX_MODE_GRAPH) {
    // one wave per list: a level-0 list holds 2M ids, so M <= 32 (search, import and GPU-side insertion alike)
    if (s->params.M < 2 || s->params.M > 32)
      return fail(EHX_EUNSUPPORTED, "graph mode: M=%u outside [2, 32]", s->params.M);
  }
  if (s->params.scan > EHX_SCAN_F16) return fail(EHX_EINVAL, "unknown scan engine %u", s->params.scan);
  {
    const bool env_f32 = env().scan_f32;  // EHX_SCAN=f32: every space scans in fp32 (A/B runs, profiling)
    const bool env_f16 = env().scan_f16;  // EHX_SCAN=f16: no int8 scan copy (A/B runs)
    s->use16 = s->params.mode == EHX_MODE_FLAT && s->params.scan != EHX_SCAN_F32 && !env_f32;
    s->has16 = s->use16;
    s->scan_sel = s->use16 ? s->params.scan : (uint32_t)EHX_SCAN_F32;
    s->ld16 = (uint32_t)round_up(dims, 128);
    s->ld8 = (uint32_t)round_up(dims, 64);
    // the int8 scan copy pays when its rows are clearly shorter than the fp16 copy's (padded to whole 64-byte stages
    // here — round 2 padded to 256 bytes, a whole ring revolution, which kept this engine off 128-dim rows and wasted
    // a quarter of the work at 384 — and to a revolution of 128 halves there)
    // ... and while its error bound (~1.3e-2 in dot units whatever d) stays well below the spread of the scores
    // (~1/sqrt(d) for isotropic rows): beyond d = 2048 most queries would lose their certificate at 256 candidates
    // (measured at d = 4096: 12 of 20) and pay for a second pass, so longer rows start at the fp16 filter
    s->has8 = s->has16 && s->params.scan == EHX_SCAN_AUTO && !env_f16 && dims <= 2048 &&
              (uint64_t)s->ld8 * 10 < (uint64_t)s->ld16 * 2 * 8;
    // the candidate list starts wider on long rows: the bound is ~1.3e-2 in dot units whatever d, the spread of the
    // scores shrinks like 1/sqrt(d) — at d = 1536 (12.5 M rows) a fifth of the queries needed more than 256 candidates
    // and the first batches paid a second engine's pass for them until the list had widened by itself
    s->i8_width = dims >= 1536 ? 2 * kMerged8 : kMerged8;
    if (env().i8_width) s->i8_width = env().i8_width;
    if (env().i8_min_rows) s->i8_min_rows = env().i8_min_rows;
    if (s->has16) {
      HIP_TRY(hipMalloc((void**)&s->dUnsafe, sizeof(unsigned long long)));
      HIP_TRY(hipMemset(s->dUnsafe, 0, sizeof(unsigned long long)));
    }
    if (s->has8) {
      HIP_TRY(hipMalloc((void**)&s->dUnsafe8, sizeof(unsigned long long)));
      HIP_TRY(hipMemset(s->dUnsafe8, 0, sizeof(unsigned long long)));
    }
  }
  HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&s->wstream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&s->wev, hipEventBlockingSync | hipEventDisableTiming));
  for (auto& e : s->sev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventBlockingSync | hipEventDisableTiming));
  if (!parent) {
    HIP_TRY(hipMalloc((void**)&s->dMaxSumsq, sizeof(float)));
    HIP_TRY(hipMemset(s->dMaxSumsq, 0, sizeof(float)));
    for (auto& e : s->ev) HIP_TRY(hipEventCreate(&e));
    for (auto& pr : s->ring)
      for (auto& e : pr) HIP_TRY(hipEventCreate(&e));
    uint64_t cap0 = s->params.initial_capacity ? s->params.initial_capacity : 128;  // index.h:21
    if ((rc = grow(s.get(), cap0))) return rc;
  }
  *out = s.get();
  E.spaces[nm] = std::move(s);
  return EHX_OK;
}

int ehx_space_create(const char* name, size_t name_len, uint32_t dims, int metric, int dtype,
                     const ehx_params* params, ehx_space** out) {
  if (!name || !out) return fail(EHX_EINVAL, "name/out must not be NULL");
  if (dims == 0 || dims > (1u << 16)) return fail(EHX_EINVAL, "dims=%u out of range", dims);
  if (metric < EHX_METRIC_L2SQ || metric > EHX_METRIC_COSINE) return fail(EHX_EINVAL, "unknown metric %d", metric);
  if (dtype != EHX_DTYPE_F32 && dtype != EHX_DTYPE_F16) return fail(EHX_EUNSUPPORTED, "dtype %d not supported", dtype);
  for (size_t i = 0; i < name_len; ++i)
    if (name[i] == '\x01') return fail(EHX_EINVAL, "space names must not contain byte 0x01 (reserved for shards)");
  int rc = ehx_init(nullptr, 0);
  if (rc) return rc;
  Engine& E = engine();
  std::lock_guard<std::mutex> lk(E.mu);
  const std::string nm(name, name_len);
  const uint32_t G = params ? params->shards : 0;
  if (G <= 1) return create_one(E, nm, dims, metric, dtype, params, E.device, false, false, out);
  if (G > 64) return fail(EHX_EINVAL, "shards=%u exceeds 64", G);
  if (params->mode == EHX_MODE_GRAPH && params->build_batch == 0xFFFFFFFFu)
    return fail(EHX_EUNSUPPORTED, "sharded graph spaces build their graphs on the GPUs (no import)");
  // the parent: key maps, routing, merge scratch on shard 0's device; then one keyless space per shard
  ehx_space* parent = nullptr;
  ehx_params pp = *params;
  if ((rc = create_one(E, nm, dims, metric, dtype, &pp, E.devices[0], false, true, &parent))) return rc;
  ehx_params cp = *params;
  cp.shards = 0;
  cp.initial_capacity = (params->initial_capacity + G - 1) / G;
  for (uint32_t i = 0; i < G; ++i) {
    ehx_space* c = nullptr;
    const std::string cn = nm + '\x01' + std::to_string(i);
    rc = create_one(E, cn, dims, metric, dtype, &cp, E.devices[i % E.devices.size()], true, false, &c);
    if (rc) {
      for (ehx_space* d : parent->shards) E.spaces.erase(d->name);
      E.spaces.erase(nm);
      return rc;
    }
    parent->shards.push_back(c);
  }
  parent->workers = std::make_unique<ShardWorkers>(G);
  *out = parent;
  return EHX_OK;
}

int ehx_space_open(const char* name, size_t name_len, ehx_space** out) {
  if (!name || !out) return fail(EHX_EINVAL, "name/out must not be NULL");
  Engine& E = engine();
  std::lock_guard<std::mutex> lk(E.mu);
  auto it = E.spaces.find(std::string(name, name_len));
  if (it == E.spaces.end() || it->second->keyless) return fail(EHX_ENOTFOUND, "Not found");
  *out = it->second.get();
  return EHX_OK;
}

int ehx_space_drop(ehx_space* s) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  Engine& E = engine();
  if (is_parent(s)) {
    // The PARENT is marked first, under its writer mutex and its lock held exclusively — every search or write that
    // runs on the shards holds the parent's lock shared or exclusively, so none is in flight now and none starts
    // later (they find `dropped`) — then the shards go (each leaves its own tombstone), then the parent itself below.
    std::vector<ehx_space*> kids;
    {
      std::lock_guard<std::mutex> wg(s->wmu);
      std::unique_lock<std::shared_mutex> wl(s->mu);
      if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
      s->dropped = true;
      kids = s->shards;
    }
    for (ehx_space* c : kids) (void)ehx_space_drop(c);
    s->workers.reset();  // (idle: nothing can reach for_each_shard any more)
  }
  std::unique_ptr<ehx_space> owned;
  {
    std::lock_guard<std::mutex> lk(E.mu);
    auto it = E.spaces.find(s->name);
    if (it == E.spaces.end() || it->second.get() != s) return fail(EHX_ENOTFOUND, "Not found");
    owned = std::move(it->second);
    E.spaces.erase(it);  // the name is free again; late users of the handle see the tombstone below
  }
  {
    std::lock_guard<std::mutex> wg(s->wmu);  // (a streaming writer may be uploading without holding mu)
    // Wait for every in-flight user (readers hold mu shared, writers exclusive), then release the HBM.  The host
    // object is NOT freed: threads that fetched the handle before the drop, or are parked on its mutexes /
    // condition variable, find `dropped` set and return EHX_ENOTFOUND.
    std::unique_lock<std::shared_mutex> wl(s->mu);
    std::lock_guard<std::mutex> sl(s->scratch_mu);
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    s->dropped = true;
    s->release_device();
    {
      std::unique_lock<std::shared_mutex> kl(s->kmu);
      s->key_to_id.clear();
      s->id_to_key.clear();
      s->id_to_key.shrink_to_fit();
    }
    s->h_levels.clear();
    s->h_levels.shrink_to_fit();
  }
  std::lock_guard<std::mutex> lk(E.mu);
  E.graveyard.push_back(std::move(owned));
  return EHX_OK;
}

int ehx_space_freeze(ehx_space* s) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  std::lock_guard<std::mutex> wg(s->wmu);
  std::unique_lock<std::shared_mutex> wl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  s->frozen = true;
  for (ehx_space* c : s->shards) {
    std::unique_lock<std::shared_mutex> cl(c->mu);
    c->frozen = true;
  }
  return EHX_OK;
}

int ehx_space_size(ehx_space* s, uint64_t* n) {
  if (!valid_space(s) || !n) return fail(EHX_EINVAL, "NULL argument");
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  *n = s->n;
  return EHX_OK;
}

int ehx_space_dims(ehx_space* s, uint32_t* dims) {
  if (!valid_space(s) || !dims) return fail(EHX_EINVAL, "NULL argument");
  *dims = s->dims;
  return EHX_OK;
}

int ehx_space_reserve(ehx_space* s, uint64_t rows) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  std::lock_guard<std::mutex> wg(s->wmu);
  std::unique_lock<std::shared_mutex> wl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (is_parent(s)) {
    const uint64_t G = s->shards.size();
    for (ehx_space* c : s->shards) {
      int rc = ehx_space_reserve(c, (rows + G - 1) / G);
      if (rc) return rc;
    }
    return EHX_OK;
  }
  HIP_TRY(hipSetDevice(s->device));
  return grow(s, rows);
}

int ehx_space_set_ef(ehx_space* s, uint32_t ef) {
  if (!valid_space(s) || ef == 0) return fail(EHX_EINVAL, "bad argument");
  std::unique_lock<std::shared_mutex> wl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  s->params.ef = ef;
  for (ehx_space* c : s->shards) {
    int rc = ehx_space_set_ef(c, ef);
    if (rc) return rc;
  }
  return EHX_OK;
}

int ehx_space_set_scan(ehx_space* s, uint32_t scan) {
  if (!valid_space(s) || scan > EHX_SCAN_F16) return fail(EHX_EINVAL, "bad argument");
  std::unique_lock<std::shared_mutex> wl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (is_parent(s)) {
    for (ehx_space* c : s->shards) {
      int rc = ehx_space_set_scan(c, scan);
      if (rc) return rc;
    }
    s->params.scan = scan;
    return EHX_OK;
  }
  if (scan != EHX_SCAN_F32 && !s->has16)
    return fail(EHX_EUNSUPPORTED, "space '%s' was created without the filter scan copies", s->name.c_str());
  s->params.scan = scan;
  s->scan_sel = scan;
  s->use16 = scan != EHX_SCAN_F32;
  return EHX_OK;
}

int ehx_space_scan_engine(ehx_space* s, uint32_t* engine) {
  if (!valid_space(s) || !engine) return fail(EHX_EINVAL, "NULL argument");
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (is_parent(s)) return ehx_space_scan_engine(s->shards[0], engine);
  *engine = (uint32_t)resolve_engine(s);
  return EHX_OK;
}

static int set_batch_locked(ehx_space* s, size_t n, const char* const* keys, const size_t* klens, const float* vecs);
static int write_rows_locked(ehx_space* s, size_t n, const std::vector<uint64_t>& ids, uint64_t next, const float* vecs,
                             std::vector<std::string>* new_keys, bool append_only = false);
static bool all_fresh_keys(const ehx_space* s, size_t n, const std::vector<uint64_t>& ids);

int ehx_set_batch(ehx_space* s, size_t n, const char* const* keys, const size_t* klens, const float* vecs) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (n == 0) return EHX_OK;
  if (!keys || !klens || !vecs) return fail(EHX_EINVAL, "NULL argument");
  std::lock_guard<std::mutex> wg(s->wmu);
  if (!is_parent(s) && s->params.mode == EHX_MODE_FLAT) {
    // Streaming fast path (copy.go's BatchSet chunks, MultiSet): a batch made only of fresh keys is a pure append.
    // The key lookup needs the lock shared only, and the upload runs with no lock on the space at all.
    std::vector<uint64_t> ids;
    std::vector<std::string> new_keys;
    uint64_t next = 0;
    bool fast = false;
    {
      std::shared_lock<std::shared_mutex> rl(s->mu);
      if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
      if (s->keyless) return fail(EHX_EINVAL, "a shard is written through its parent space");
      if (!s->frozen && !s->implicit_keys) {
        resolve_keys(s, n, keys, klens, &ids, &next, &new_keys);
        fast = new_keys.size() == n && all_fresh_keys(s, n, ids);
      }
    }
    if (fast) return write_rows_locked(s, n, ids, next, vecs, &new_keys, true);
  }
  std::unique_lock<std::shared_mutex> wl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (s->keyless) return fail(EHX_EINVAL, "a shard is written through its parent space");
  auto write = [&](size_t cnt, const char* const* ks, const size_t* kl, const float* v) -> int {
    return is_parent(s) ? sharded_set_batch(s, cnt, ks, kl, v) : set_batch_locked(s, cnt, ks, kl, v);
  };
  if (s->params.mode == EHX_MODE_GRAPH && n > 1 && s->params.build_batch != 0xFFFFFFFFu) {
    // graph mode replays a batch in call order; when it re-writes keys (known ones, or the same key
    // twice) every row must be in HBM exactly when its turn comes, so such batches go row by row
    bool rewrite = false;
    {
      std::set<std::string> seen;
      std::shared_lock<std::shared_mutex> kl(s->kmu);
      for (size_t i = 0; i < n && !rewrite; ++i) {
        std::string k(keys[i], klens[i]);
        rewrite = s->key_to_id.count(k) != 0 || !seen.insert(std::move(k)).second;
      }
    }
    if (rewrite) {
      for (size_t i = 0; i < n; ++i) {
        int rc = write(1, keys + i, klens + i, vecs + i * s->dims);
        if (rc) return rc;
      }
      return EHX_OK;
    }
  }
  return write(n, keys, klens, vecs);
}

// rows -> pinned staging (fp16 spaces: rounded to binary16, round-to-nearest-even, on the way); large slabs are split
// over four threads
static void stage_rows(char* dst, const float* src, size_t elems, bool half) {
  auto work = [=](size_t e0, size_t e1) {
    if (half) {
      _Float16* h = (_Float16*)dst;
      for (size_t e = e0; e < e1; ++e) h[e] = (_Float16)src[e];
    } else {
      memcpy(dst + e0 * sizeof(float), src + e0, (e1 - e0) * sizeof(float));
    }
  };
  constexpr size_t kThreads = 4;
  if (elems * sizeof(float) < (2u << 20)) {
    work(0, elems);
    return;
  }
  const size_t per = ((elems + kThreads - 1) / kThreads + 63) & ~(size_t)63;
  std::thread th[kThreads - 1];
  size_t started = 0, done_to = std::min(elems, per);  // [0, per) is this thread's share
  for (size_t t = 1; t < kThreads && t * per < elems; ++t) {
    try {
      th[t - 1] = std::thread(work, t * per, std::min(elems, (t + 1) * per));
      ++started;
      done_to = std::min(elems, (t + 1) * per);
    } catch (const std::system_error&) {
      break;  // no thread to be had (a process at its thread limit): the caller's thread copies the rest
    }
  }
  work(0, std::min(elems, per));
  for (size_t t = 0; t < started; ++t) th[t].join();
  if (done_to < elems) work(done_to, elems);
}

// wait for a stream of the space: the writers' stream through the blocking event, any other by hipStreamSynchronize
static int sync_stream(ehx_space* s, hipStream_t st) {
  if (st == s->wstream && s->wev) {
    HIP_TRY(hipEventRecord(s->wev, st));
    HIP_TRY(hipEventSynchronize(s->wev));
  } else {
    HIP_TRY(hipStreamSynchronize(st));
  }
  return EHX_OK;
}

// (re)build the derived copies of rows [row0, row0+n) after they were written; must follow row_stats:
// graph mode: the search copy; flat fp32 spaces: the fp16 scan copy
// exclusive: no search can be reading the space (the caller holds s->mu exclusively and the writer's stream has waited
// for the searches in flight) — rows below the published count may then move inside their tiles; otherwise every row
// of [row0, row0 + n) lies beyond the published row count.  n_after: the row count once this write is published.
static int refresh_scan16(ehx_space* s, uint64_t row0, uint64_t n, hipStream_t st, bool exclusive, uint64_t n_after) {
  if (!st) st = s->stream;
  if (s->dXs && !s->x_perm && n)
    HIP_TRY(launch_make_search_copy(s->dX, s->x_half, s->dInv, row0, n, s->ld, s->metric, s->dXs, st));
  if ((!s->has16 && !s->has8) || n == 0) return EHX_OK;  // (kept current whatever engine is selected right now)
  unsigned long long u = 0, u8 = 0;
  if (s->has16) {
    HIP_TRY(launch_make_scan16(s->dX, s->x_half, row0, n, s->dims, s->ld, s->ld16, s->metric, s->dX16, s->dRowp16,
                               s->dUnsafe, st));
    HIP_TRY(hipMemcpyAsync(&u, s->dUnsafe, sizeof(u), hipMemcpyDeviceToHost, st));
  }
  if (s->has8) {
    // Full tiles are stored ordered by quantisation step (k_misc.hip).  Re-ordering moves rows inside a tile, so it
    // happens only where no scan can look: the fresh rows of an append (a tile that straddles the published row count
    // keeps the row order, for good), or anywhere under an exclusive writer — which re-makes whole tiles, because a
    // rewritten row of an ordered tile no longer sits where its id says.
    uint64_t r8 = row0, e8 = row0 + n;
    const bool sort_tiles = env().i8_sort;
    if (exclusive) {
      {  // searches still in flight on other streams
        int rcw = wait_searches_in_flight(s, st);
        if (rcw) return rcw;
      }
      r8 = row0 & ~(uint64_t)255;
      e8 = std::min<uint64_t>(round_up(row0 + n, 256), std::max<uint64_t>(n_after, row0 + n));
    }
    int rc8;
    const uint64_t slo = sort_tiles ? r8 : 0, shi = sort_tiles ? e8 : 0;
    if ((rc8 = s->dTileList.ensure((make_scan8_scratch_bytes(r8, e8 - r8, slo, shi) + 7) / 8))) return rc8;
    HIP_TRY(launch_make_scan8(s->dX, s->x_half, r8, e8 - r8, s->dims, s->ld, s->ld8, s->metric, s->dX8, s->dRowp8,
                              s->dTilep8, s->dPerm8, s->dTileg8, slo, shi, s->dTileList.p, s->dUnsafe8, st));
    if (s->dTileList.n > (64u << 20) / 8) {  // (a bulk load's scratch — 9 bytes per row — is not kept)
      HIP_TRY(hipStreamSynchronize(st));
      s->dTileList.release();
    }
    HIP_TRY(hipMemcpyAsync(&u8, s->dUnsafe8, sizeof(u8), hipMemcpyDeviceToHost, st));
  }
  {
    int rcs = sync_stream(s, st);
    if (rcs) return rcs;
  }
  s->h_unsafe = u;
  s->h_unsafe8 = u8;
  return EHX_OK;
}


static int set_batch_locked(ehx_space* s, size_t n, const char* const* keys, const size_t* klens, const float* vecs) {
  if (s->frozen) return fail(EHX_EIMMUTABLE, "Cannot write to immutable space");
  if (s->implicit_keys) return fail(EHX_EINVAL, "space '%s' holds synthetic rows with implicit keys", s->name.c_str());
  std::vector<uint64_t> ids;
  std::vector<std::string> new_keys;
  uint64_t next = 0;
  resolve_keys(s, n, keys, klens, &ids, &next, &new_keys);
  return write_rows_locked(s, n, ids, next, vecs, &new_keys);
}

// every key of the batch is new and distinct: the rows are a pure append
static bool all_fresh_keys(const ehx_space* s, size_t n, const std::vector<uint64_t>& ids) {
  for (size_t i = 0; i < n; ++i)
    if (ids[i] != s->n + i) return false;
  return true;
}

// rows `vecs[i]` -> row ids[i] of the space (ids < next; ids >= s->n are appended, dense), then statistics, derived
// copies, graph; finally publishes the keys (new_keys, in id order from s->n) and the new row count `next`.
//   append_only = false: the caller holds s->mu exclusively (rows may be rewritten in place, graphs change).
//   append_only = true : flat spaces, every id >= s->n.  The caller holds s->wmu only: searches keep running while
//     the rows are uploaded, described and copied BEYOND the published row count (every kernel masks rows >= n),
//     on the writers' stream; s->mu is taken exclusively just to grow the arrays (rare) and to publish.
static int write_rows_locked(ehx_space* s, size_t n, const std::vector<uint64_t>& ids, uint64_t next, const float* vecs,
                             std::vector<std::string>* new_keys, bool append_only) {
  if (s->frozen) return fail(EHX_EIMMUTABLE, "Cannot write to immutable space");
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t ws = s->wstream ? s->wstream : s->stream;
  // rows rewritten in place: in-flight device searches (enqueued without the lock being held any more) finish first
  if (!append_only) {
    int rcw = wait_searches_in_flight(s, ws);
    if (rcw) return rcw;
  }
  const uint64_t old_n = s->n;
  int rc;
  if (append_only && next >= s->cap) {
    std::unique_lock<std::shared_mutex> gl(s->mu);  // the arrays move: no search may be running
    if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
    rc = ensure_rows(s, next);
  } else {
    rc = ensure_rows(s, next);
  }
  if (rc) return rc;
  // upload through pinned staging in slabs; rows may be non-contiguous (updates) so copy per row
  // (fp16 spaces: rows are rounded to binary16, round-to-nearest-even, while they are staged)
  // Two staging halves, ping-pong: slab i is copied into its half (by up to four host threads — one core moves
  // ~8 GB/s, a 25-MB chunk of copy.go's 8192 x 768 rows would spend 3 ms there) while slab i-1 is on the wire.
  const size_t row_bytes = (size_t)s->dims * s->esz;
  const size_t slab_rows = std::max<size_t>(1, std::min<size_t>(n, (8u << 20) / row_bytes));
  const size_t half_bytes = (slab_rows * row_bytes + 255) & ~(size_t)255;
  if ((rc = ensure_stage(s, 2 * half_bytes))) return rc;
  uint64_t min_id = ~0ull, max_id = 0;
  size_t slab = 0;
  // single-copy graph space: everything fallible that does not depend on the upload happens BEFORE the first row lands
  // (the id list of a non-contiguous batch and its device buffer); a failure after an in-place upload of committed rows
  // poisons the space (the rows would stay in raw order inside a permuted store)
  bool perm_run = true;
  std::vector<uint64_t> perm_uniq;
  bool touches_committed = false;
  if (s->x_perm) {
    for (size_t i = 1; i < n && perm_run; ++i) perm_run = ids[i] == ids[0] + i;
    if (!perm_run) {
      perm_uniq.assign(ids.begin(), ids.begin() + n);
      std::sort(perm_uniq.begin(), perm_uniq.end());
      perm_uniq.erase(std::unique(perm_uniq.begin(), perm_uniq.end()), perm_uniq.end());
      if ((rc = s->dPermIds.ensure(perm_uniq.size()))) return rc;
    }
    for (size_t i = 0; i < n && !touches_committed; ++i) touches_committed = ids[i] < old_n;
  }
  struct Poison {   // armed while raw rows may sit in a permuted store
    ehx_space* s;
    bool armed = false;
    ~Poison() { if (armed) s->poisoned.store(true); }
  } poison{s};
  for (size_t i0 = 0; i0 < n; i0 += slab_rows, ++slab) {
    if (touches_committed) poison.armed = true;
    const size_t m = std::min(slab_rows, n - i0);
    char* stage = (char*)s->hStage + (slab & 1) * half_bytes;
    if (slab >= 2) HIP_TRY(hipEventSynchronize(s->sev[slab & 1]));  // the upload that last used this half
    stage_rows(stage, vecs + i0 * s->dims, m * s->dims, s->x_half);
    // contiguous run of fresh ids -> one 2D copy; otherwise row by row
    bool contiguous = true;
    for (size_t i = 1; i < m; ++i)
      if (ids[i0 + i] != ids[i0] + i) { contiguous = false; break; }
    if (contiguous) {
      HIP_TRY(hipMemcpy2DAsync(s->xrow(ids[i0]), (size_t)s->ld * s->esz, stage, row_bytes,
                               row_bytes, m, hipMemcpyHostToDevice, ws));
    } else {
      for (size_t i = 0; i < m; ++i)
        HIP_TRY(hipMemcpyAsync(s->xrow(ids[i0 + i]), stage + i * row_bytes, row_bytes,
                               hipMemcpyHostToDevice, ws));
    }
    HIP_TRY(hipEventRecord(s->sev[slab & 1], ws));
    for (size_t i = 0; i < m; ++i) {
      min_id = std::min(min_id, ids[i0 + i]);
      max_id = std::max(max_id, ids[i0 + i]);
    }
  }
  // (the stream is waited for below, before the commit: both halves are free again when this call returns)
  if (s->x_perm) {
    // single-copy graph space: the rows just written go into the search copy's block order, in place, exactly once
    // each (the permutation is its own inverse: a row written twice in this batch is permuted once)
    if (perm_run) {
      HIP_TRY(launch_permute_blocks((float*)s->dX, s->ld, ids[0], n, nullptr, ws));
    } else {
      HIP_TRY(hipMemcpyAsync(s->dPermIds.p, perm_uniq.data(), perm_uniq.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ws));
      HIP_TRY(launch_permute_blocks((float*)s->dX, s->ld, 0, perm_uniq.size(), s->dPermIds.p, ws));
    }
    HIP_TRY(hipStreamSynchronize(ws));  // (the list lives on this stack frame; the rows are in block order from here on)
    poison.armed = false;
  }
  // per-row statistics over the touched id range (idempotent for untouched rows in between)
  HIP_TRY(launch_row_stats(s->dX, s->x_half, min_id, max_id - min_id + 1, s->dims, s->ld, s->metric, s->dInv,
                           s->dRowp, s->dMaxSumsq, ws, s->x_perm ? 1 : 0));
  if ((rc = refresh_scan16(s, min_id, max_id - min_id + 1, ws, !append_only, next))) return rc;
  if ((rc = sync_stream(s, ws))) return rc;
  // commit: the rows are resident and described — publish the keys and the new row count
  // (the keys first, under their own lock — searches keep running — then the row count, under the space's lock for
  // the length of one store)
  if (new_keys) {
    std::shared_lock<std::shared_mutex> rl(s->mu, std::defer_lock);
    if (append_only) {
      rl.lock();  // (shared: keeps a drop out, not the searches)
      if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
    }
    std::unique_lock<std::shared_mutex> kl(s->kmu);
    for (size_t i = 0; i < new_keys->size(); ++i) s->key_to_id.emplace((*new_keys)[i], old_n + i);
    for (auto& k : *new_keys) s->id_to_key.push_back(std::move(k));
  }
  {
    std::unique_lock<std::shared_mutex> pl(s->mu, std::defer_lock);
    if (append_only) {
      pl.lock();
      if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
    }
    s->n = next;
  }
  if (s->params.mode == EHX_MODE_GRAPH) {
    // new rows join the graph one at a time, in id order (ANNIndex::set -> addPoint, index.cc:36);
    // rows overwritten in place keep their links (hnswlib's updatePoint repair is not built yet)
    // in call order: a fresh key is an insertion, a known key hnswlib's update-in-place
    if (s->g_n == old_n && s->params.build_batch != 0xFFFFFFFFu) {
      if ((rc = graph_ensure_arrays(s))) return rc;
      // Opt-in bulk write (ehx_params.build_batch > 1 given explicitly): a batch made only of fresh keys
      // joins the graph in concurrent rounds of up to build_batch rows — hnswlib's multi-threaded
      // add_items (offlinehub.py:89) — instead of one row per round.
      bool all_fresh = s->params.build_batch > 1 && next - old_n == n;
      for (size_t i = 0; i < n && all_fresh; ++i) all_fresh = ids[i] == old_n + i;
      if (all_fresh) return graph_insert(s, old_n, n, s->params.build_batch);
      for (size_t i = 0; i < n; ++i) {
        if (ids[i] >= s->g_n) {
          if ((rc = graph_insert(s, ids[i], 1, 1))) return rc;
        } else {
          if ((rc = graph_update(s, (uint32_t)ids[i]))) return rc;
        }
      }
    } else {
      s->g_stale_updates += n - (next - old_n);
    }
  }
  return EHX_OK;
}

namespace {
int write_rows_locked_fwd(ehx_space* s, size_t n, const std::vector<uint64_t>& ids, uint64_t next, const float* vecs) {
  return write_rows_locked(s, n, ids, next, vecs, nullptr);
}
}  // namespace

// Single-row Sets (the reference's usage: one Set per RPC / per goroutine, runner/copy.go:146-161 runs 500 at a time)
// are combined like the single-query searches are: the first caller becomes the leader, takes every request that
// queued up meanwhile (up to 4096) and writes them as ONE batch; under load the batch size grows by itself.  A call
// returns after its row is published, so a following ehx_knn from the same thread sees it (index_test.cc:39-49).
constexpr size_t kCombineMaxBatch = 4096;

int ehx_set(ehx_space* s, const char* key, size_t klen, const float* vec) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (!key || !vec) return fail(EHX_EINVAL, "key / vector is NULL");
  ehx_space::SetReq me;
  me.key = key;
  me.klen = klen;
  me.vec = vec;
  std::unique_lock<std::mutex> lk(s->wq_mu);
  s->wq.push_back(&me);
  std::vector<ehx_space::SetReq*> group;
  std::vector<const char*> ks;
  std::vector<size_t> kl;
  std::vector<float> rows;
  while (!me.done) {
    if (s->wq_leader) {
      s->wq_cv.wait(lk, [&] { return me.done || !s->wq_leader; });
      continue;
    }
    s->wq_leader = true;
    while (!me.done && !s->wq.empty()) {
      const size_t m = std::min(s->wq.size(), kCombineMaxBatch);
      group.assign(s->wq.begin(), s->wq.begin() + m);
      s->wq.erase(s->wq.begin(), s->wq.begin() + m);
      lk.unlock();
      int rc;
      if (m == 1) {
        const char* k1[1] = {group[0]->key};
        size_t l1[1] = {group[0]->klen};
        rc = ehx_set_batch(s, 1, k1, l1, group[0]->vec);
      } else {
        ks.resize(m);
        kl.resize(m);
        rows.resize(m * s->dims);
        for (size_t i = 0; i < m; ++i) {
          ks[i] = group[i]->key;
          kl[i] = group[i]->klen;
          memcpy(rows.data() + i * s->dims, group[i]->vec, s->dims * sizeof(float));
        }
        rc = ehx_set_batch(s, m, ks.data(), kl.data(), rows.data());
        s->n_combined_batches += 1;
        s->n_combined_sets += m;
      }
      lk.lock();
      for (auto* r : group) {
        r->rc = rc;
        if (rc) snprintf(r->err, sizeof(r->err), "%s", g_err);
        r->done = true;
      }
      s->wq_cv.notify_all();
    }
    s->wq_leader = false;
    s->wq_cv.notify_all();
  }
  lk.unlock();
  if (me.rc) snprintf(g_err, sizeof(g_err), "%s", me.err);
  return me.rc;
}

int ehx_get_by_id(ehx_space* s, uint64_t id, float* out_vec) {
  if (!valid_space(s) || !out_vec) return fail(EHX_EINVAL, "NULL argument");
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (id >= s->n) return fail(EHX_ENOTFOUND, "Not found");
  if (is_parent(s)) return ehx_get_by_id(s->shards[id % s->shards.size()], id / s->shards.size(), out_vec);
  HIP_TRY(hipSetDevice(s->device));
  if (s->x_half) {
    std::vector<_Float16> h(s->dims);
    HIP_TRY(hipMemcpy(h.data(), s->xrow(id), (size_t)s->dims * 2, hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < s->dims; ++c) out_vec[c] = (float)h[c];
    return EHX_OK;
  }
  if (s->x_perm) {  // single-copy graph space: the row is stored in the search copy's block order — undo it here
    if (s->poisoned.load())
      return fail(EHX_EINTERNAL, "graph space: an in-place overwrite failed half way (rows left in raw order); drop and rebuild it");
    std::vector<float> h(s->ld);
    HIP_TRY(hipMemcpy(h.data(), s->xrow(id), (size_t)s->ld * sizeof(float), hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < s->dims; ++c) out_vec[c] = h[search_copy_pos(c)];
    return EHX_OK;
  }
  HIP_TRY(hipMemcpy(out_vec, s->xrow(id), (size_t)s->dims * sizeof(float), hipMemcpyDeviceToHost));
  return EHX_OK;
}

int ehx_get(ehx_space* s, const char* key, size_t klen, float* out_vec) {
  if (!valid_space(s) || !key || !out_vec) return fail(EHX_EINVAL, "NULL argument");
  uint64_t id;
  {
    std::shared_lock<std::shared_mutex> rl(s->mu);
    if (s->dropped || lookup_key(s, key, klen, &id)) return fail(EHX_ENOTFOUND, "Not found");
  }
  return ehx_get_by_id(s, id, out_vec);
}

int ehx_key_of(ehx_space* s, uint64_t id, char* out_key, size_t cap, size_t* klen) {
  if (!valid_space(s) || !klen) return fail(EHX_EINVAL, "NULL argument");
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  std::string k;
  if (key_for_id(s, id, &k)) return fail(EHX_ENOTFOUND, "Not found");
  *klen = k.size();
  if (out_key && cap) memcpy(out_key, k.data(), std::min(cap, k.size()));
  return EHX_OK;
}

int ehx_knn_device(ehx_space* s, void* stream, size_t n_queries, const float* d_queries, uint32_t k,
                   uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (n_queries && k && (!d_queries || !d_out_ids || !d_out_dist || !d_out_count))
    return fail(EHX_EINVAL, "NULL device pointer");
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (is_parent(s)) {
    if (n_queries == 0 || k == 0) return EHX_OK;
    hipPointerAttribute_t at;
    HIP_TRY(hipPointerGetAttributes(&at, d_queries));
    return sharded_knn(s, n_queries, nullptr, d_queries, at.device, k, d_out_ids, d_out_dist, d_out_count, true,
                       (hipStream_t)stream);
  }
  std::lock_guard<std::mutex> sl(s->scratch_mu);
  HIP_TRY(hipSetDevice(s->device));
  return knn_device_locked(s, (hipStream_t)stream, n_queries, d_queries, k, d_out_ids, d_out_dist, d_out_count);
}

static int knn_host_direct(ehx_space* s, size_t n_queries, const float* queries, uint32_t k, uint64_t* out_ids,
                           float* out_dist, uint32_t* out_count) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (n_queries == 0) return EHX_OK;
  if (!out_count) return fail(EHX_EINVAL, "out_count is NULL");
  if (k == 0) {
    for (size_t i = 0; i < n_queries; ++i) out_count[i] = 0;
    return EHX_OK;
  }
  if (!queries || !out_ids || !out_dist) return fail(EHX_EINVAL, "NULL argument");
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (is_parent(s)) return sharded_knn(s, n_queries, queries, nullptr, 0, k, out_ids, out_dist, out_count, false, nullptr);
  HIP_TRY(hipSetDevice(s->device));
  int rc;
  const size_t qbytes = n_queries * s->dims * sizeof(float);
  // A small call (the reference's request shape: one query, ten keys) is all fixed cost: its queries go through a
  // pinned staging buffer (an asynchronous copy instead of the runtime's pageable-memory path) and its three result
  // arrays come back as ONE block into pinned memory instead of three blocking copies.
  constexpr size_t kSmallCall = 32u << 10;
  const size_t nk = n_queries * k;
  const size_t out_bytes = nk * (sizeof(uint64_t) + sizeof(float)) + n_queries * sizeof(uint32_t);
  if (qbytes <= kSmallCall && out_bytes <= kSmallCall) {
    std::lock_guard<std::mutex> sl(s->scratch_mu);
    // ONE query against a small flat shard — the reference's request (server.cc:172-210; BASELINE configs[0]): a single
    // launch reads the query from host-visible memory, scans every row in the oracle's arithmetic, and the last
    // workgroup writes the answer into host-visible memory and raises a flag this thread spins on (k_flat.hip:
    // single_query_kernel).  10 k x 128: ~130 us through the three-launch path -> see DESIGN §e.
    const uint64_t one_bytes = env().small_exact_bytes;
    const bool one_on = env().one_launch;
    if (one_on && n_queries == 1 && k <= 64 && s->params.mode == EHX_MODE_FLAT && s->scan_sel == EHX_SCAN_AUTO && s->n > 0 &&
        s->ld <= 4096 && (uint64_t)s->n * s->ld * s->esz <= one_bytes) {
      constexpr size_t kOneQ = 16384;   // query slot (ld <= 4096 floats)
      if (!s->hOnePin) {
        HIP_TRY(hipHostMalloc((void**)&s->hOnePin, kOneQ + 2048, hipHostMallocCoherent | hipHostMallocMapped));
        memset(s->hOnePin, 0, kOneQ + 2048);
      }
      if (!s->dOneTicket) {
        HIP_TRY(hipMalloc((void**)&s->dOneTicket, sizeof(uint32_t)));
        HIP_TRY(hipMemset(s->dOneTicket, 0, sizeof(uint32_t)));
      }
      const uint32_t rpb = (uint32_t)std::max<uint64_t>(64, ((s->n + 1023) / 1024 + 63) / 64 * 64);  // <= 1024 workgroups
      const uint32_t n_blocks = (uint32_t)((s->n + rpb - 1) / rpb);
      if ((rc = s->dOnePart.ensure((size_t)n_blocks * 64))) return rc;
      if ((rc = wait_searches_in_flight(s, s->stream))) return rc;  // (device searches queued on other streams)
      char* h = s->hOnePin;
      memcpy(h, queries, qbytes);
      SingleQueryArgs a;
      a.q_in = (const float*)h;
      a.X = s->dX;
      a.inv_norm = s->dInv;
      a.part = s->dOnePart.p;
      a.ticket = s->dOneTicket;
      a.out_ids = (uint64_t*)(h + kOneQ);
      a.out_dist = (float*)(h + kOneQ + 512);
      a.out_count = (uint32_t*)(h + kOneQ + 768);
      a.done_flag = (uint32_t*)(h + kOneQ + 1024);
      a.seq = ++s->one_seq ? s->one_seq : ++s->one_seq;   // (never 0: the buffer starts zeroed)
      a.x_half = (uint32_t)s->x_half;
      a.n = (uint32_t)s->n;
      a.dims = s->dims;
      a.ld = s->ld;
      a.rows_per_block = rpb;
      a.k = k;
      a.metric = s->metric;
      HIP_TRY(launch_single_query(a, n_blocks, s->stream));
      volatile uint32_t* flag = (volatile uint32_t*)a.done_flag;
      bool seen = false;
      for (uint32_t spin = 0; spin < 4000000u; ++spin) {   // ~ tens of milliseconds at most, then ask the runtime
        if (*flag == a.seq) {
          seen = true;
          break;
        }
        __builtin_ia32_pause();
      }
      if (!seen) {
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (*flag != a.seq) return fail(EHX_EINTERNAL, "single-query kernel finished without publishing its result");
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      memcpy(out_ids, a.out_ids, k * sizeof(uint64_t));
      memcpy(out_dist, a.out_dist, k * sizeof(float));
      out_count[0] = *a.out_count;
      s->n_queries += 1;
      s->n_exhaustive += 1;
      s->n_one_launch += 1;
      s->n_dist += s->n;
      return EHX_OK;
    }
    if ((rc = s->dQraw.ensure(n_queries * s->dims))) return rc;
    if (!s->hSmallPin) HIP_TRY(hipHostMalloc((void**)&s->hSmallPin, 2 * kSmallCall, hipHostMallocDefault));
    if ((rc = s->dSmallOut.ensure(kSmallCall / sizeof(uint64_t)))) return rc;
    uint64_t* d_ids = s->dSmallOut.p;
    float* d_dist = (float*)(d_ids + nk);
    uint32_t* d_cnt = (uint32_t*)(d_dist + nk);
    memcpy(s->hSmallPin, queries, qbytes);
    HIP_TRY(hipMemcpyAsync(s->dQraw.p, s->hSmallPin, qbytes, hipMemcpyHostToDevice, s->stream));
    if ((rc = knn_device_locked(s, s->stream, n_queries, s->dQraw.p, k, d_ids, d_dist, d_cnt))) return rc;
    char* h = s->hSmallPin + kSmallCall;
    HIP_TRY(hipMemcpyAsync(h, d_ids, out_bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    memcpy(out_ids, h, nk * sizeof(uint64_t));
    memcpy(out_dist, h + nk * sizeof(uint64_t), nk * sizeof(float));
    memcpy(out_count, h + nk * (sizeof(uint64_t) + sizeof(float)), n_queries * sizeof(uint32_t));
    return EHX_OK;
  }
  // A batch: through a slot of its own (see ehx_space::HostSlot) — only the device pipeline itself is serialised (the
  // pipeline's lock is NOT held while the queries are staged: the first version took it on entry and two callers ran
  // strictly one after the other).
  ehx_space::HostSlot* hs = nullptr;
  {
    std::unique_lock<std::mutex> hl(s->hs_mu);
    s->hs_cv.wait(hl, [&] {
      for (auto& h : s->hslot)
        if (!h.busy) return true;
      return false;
    });
    for (auto& h : s->hslot)
      if (!h.busy) {
        hs = &h;
        break;
      }
    hs->busy = true;
  }
  struct Release {
    ehx_space* s;
    ehx_space::HostSlot* h;
    bool ok = false;   // set on the success path; an early error return may leave copies / kernels of this call in flight
    ~Release() {
      if (!ok) {  // drain them before the slot's pinned and device buffers go to the next caller
        if (h->st) (void)hipStreamSynchronize(h->st);
        if (s->stream) (void)hipStreamSynchronize(s->stream);
      }
      {
        std::lock_guard<std::mutex> hl(s->hs_mu);
        h->busy = false;
      }
      s->hs_cv.notify_one();
    }
  } release{s, hs};
  const size_t ids_b = nk * sizeof(uint64_t), dist_b = nk * sizeof(float);
  const size_t need = qbytes + out_bytes;
  if (!hs->st) {
    HIP_TRY(hipStreamCreateWithFlags(&hs->st, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&hs->in_ev, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&hs->done_ev, hipEventDisableTiming));
  }
  if (hs->pin_bytes < need) {
    HIP_TRY(hipStreamSynchronize(hs->st));
    if (hs->pin) (void)hipHostFree(hs->pin);
    hs->pin = nullptr;
    hs->pin_bytes = 0;
    HIP_TRY(hipHostMalloc((void**)&hs->pin, need, hipHostMallocDefault));
    hs->pin_bytes = need;
  }
  if ((rc = hs->dq.ensure(n_queries * s->dims))) return rc;
  if ((rc = hs->dout.ensure(out_bytes))) return rc;
  uint64_t* d_ids = (uint64_t*)hs->dout.p;
  float* d_dist = (float*)(hs->dout.p + ids_b);
  uint32_t* d_cnt = (uint32_t*)(hs->dout.p + ids_b + dist_b);
  memcpy(hs->pin, queries, qbytes);
  HIP_TRY(hipMemcpyAsync(hs->dq.p, hs->pin, qbytes, hipMemcpyHostToDevice, hs->st));
  HIP_TRY(hipEventRecord(hs->in_ev, hs->st));
  // The int8 engine's first stage — all of a batch unless queries lose their certificate — runs in one of the space's two
  // scratch sets WITHOUT the pipeline-wide lock: this call's launches queue up on the space's stream behind the other
  // caller's while that one still waits for its verdict, so the scan kernels of consecutive batches run back to back with
  // no host round trip (launches, verdict copy, thread wake-up: ~0.1 ms per batch) between them.  A batch that does lose
  // queries is re-run through the full engine chain under the lock (rare; the chain also adapts the list's length).
  const bool pipe_on = env().host_pipeline;
  bool done = false, have_failed = false;
  std::vector<uint32_t> failed;
  size_t n_short = 0;
  uint32_t kprime_used = 0;
  if (pipe_on && s->params.mode == EHX_MODE_FLAT && k <= EHX_MAX_K && s->n > 0 && resolve_engine(s) == EHX_ENGINE_I8) {
    const int set = (int)(s->i8_next_set.fetch_add(1, std::memory_order_relaxed) & 1u);   // consecutive batches alternate
    ehx_space::I8Set& sc = s->i8set[set];
    std::lock_guard<std::mutex> l(sc.mu);
    HIP_TRY(hipStreamWaitEvent(s->stream, hs->in_ev, 0));
    if ((rc = flat_pass8(s, set, s->stream, n_queries, hs->dq.p, k, d_ids, d_dist, d_cnt, false, &kprime_used))) return rc;
    HIP_TRY(hipMemcpyAsync(sc.hUncertPin, sc.dUncert, sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipEventRecord(sc.verdict, s->stream));
    HIP_TRY(hipEventRecord(hs->done_ev, s->stream));
    HIP_TRY(hipEventSynchronize(sc.verdict));
    s->n_queries += n_queries;
    s->n_dist += (uint64_t)n_queries * s->n;
    s->bytes_algo += s->n * (uint64_t)s->dims + (uint64_t)n_queries * s->dims * 4ull + (uint64_t)n_queries * k * 12ull;
    if (*sc.hUncertPin == 0) {
      done = true;
      s->n_i8_queries += n_queries;
      i8_adapt(s, n_queries, 0, 0, kprime_used);   // a clean batch: the score decays
    } else {  // which queries, and why: the engine chain continues with them (below, under the pipeline lock)
      HIP_TRY(hipMemsetAsync(sc.dUncert, 0, sizeof(unsigned long long), s->stream));
      std::vector<uint32_t> flags(n_queries);
      HIP_TRY(hipMemcpyAsync(flags.data(), sc.dUflags.p, n_queries * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream));
      for (size_t j = 0; j < n_queries; ++j)
        if (flags[j]) {
          failed.push_back((uint32_t)j);
          n_short += flags[j] == 2u;
        }
      have_failed = true;
    }
  }
  if (!done) {
    std::lock_guard<std::mutex> sl2(s->scratch_mu);
    HIP_TRY(hipStreamWaitEvent(s->stream, hs->in_ev, 0));
    if ((rc = knn_device_locked(s, s->stream, n_queries, hs->dq.p, k, d_ids, d_dist, d_cnt, have_failed ? &failed : nullptr,
                                n_short, kprime_used)))
      return rc;
    HIP_TRY(hipEventRecord(hs->done_ev, s->stream));
  }
  char* ho = hs->pin + qbytes;
  HIP_TRY(hipStreamWaitEvent(hs->st, hs->done_ev, 0));
  HIP_TRY(hipMemcpyAsync(ho, hs->dout.p, out_bytes, hipMemcpyDeviceToHost, hs->st));
  HIP_TRY(hipStreamSynchronize(hs->st));
  memcpy(out_ids, ho, ids_b);
  memcpy(out_dist, ho + ids_b, dist_b);
  memcpy(out_count, ho + ids_b + dist_b, n_queries * sizeof(uint32_t));
  release.ok = true;
  return EHX_OK;
}

// Small calls (the reference's usage: one query per RPC, server.cc:172-210; Go Nearest, online.go:63) are
// coalesced: the first caller becomes the leader, gathers every request that queued up meanwhile
// (same k, up to 1024 queries), runs ONE device batch and hands the results back.  An uncontended call
// runs immediately; under load the batch size grows by itself with the scan time.
constexpr size_t kCoalesceMaxCall = 64;     // calls above this size already are batches
constexpr size_t kCoalesceMaxBatch = 1024;

int ehx_knn(ehx_space* s, size_t n_queries, const float* queries, uint32_t k, uint64_t* out_ids,
            float* out_dist, uint32_t* out_count) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (n_queries == 0) return EHX_OK;
  if (n_queries > kCoalesceMaxCall || k == 0 || !queries || !out_ids || !out_dist || !out_count)
    return knn_host_direct(s, n_queries, queries, k, out_ids, out_dist, out_count);
  ehx_space::KnnReq me;
  me.q = queries;
  me.nq = n_queries;
  me.k = k;
  me.ids = out_ids;
  me.dist = out_dist;
  me.cnt = out_count;
  std::unique_lock<std::mutex> lk(s->bq_mu);
  s->bq.push_back(&me);
  std::vector<ehx_space::KnnReq*> group;
  std::vector<float> q;
  std::vector<uint64_t> ids;
  std::vector<float> dist;
  std::vector<uint32_t> cnt;
  while (!me.done) {
    if (s->bq_leader) {  // somebody else is serving: wait for my result, or for the leadership to come free
      s->bq_cv.wait(lk, [&] { return me.done || !s->bq_leader; });
      continue;
    }
    // Leader: serve groups until my own request has been answered, then hand the role to a waiter (a leader
    // that kept serving while the queue refills would delay its own, already answered, caller without bound).
    s->bq_leader = true;
    while (!me.done && !s->bq.empty()) {
      // one group = the oldest request's k, in arrival order, up to kCoalesceMaxBatch queries
      group.clear();
      const uint32_t gk = s->bq.front()->k;
      size_t total = 0;
      for (auto it = s->bq.begin(); it != s->bq.end();) {
        if ((*it)->k == gk && total + (*it)->nq <= kCoalesceMaxBatch) {
          total += (*it)->nq;
          group.push_back(*it);
          it = s->bq.erase(it);
        } else {
          ++it;
        }
      }
      lk.unlock();
      int rc;
      if (group.size() == 1) {
        ehx_space::KnnReq* r = group[0];
        rc = knn_host_direct(s, r->nq, r->q, gk, r->ids, r->dist, r->cnt);
      } else {
        q.resize(total * s->dims);
        ids.resize(total * gk);
        dist.resize(total * gk);
        cnt.resize(total);
        size_t off = 0;
        for (auto* r : group) {
          memcpy(q.data() + off * s->dims, r->q, r->nq * s->dims * sizeof(float));
          off += r->nq;
        }
        rc = knn_host_direct(s, total, q.data(), gk, ids.data(), dist.data(), cnt.data());
        off = 0;
        for (auto* r : group) {
          if (rc == EHX_OK) {
            memcpy(r->ids, ids.data() + off * gk, r->nq * gk * sizeof(uint64_t));
            memcpy(r->dist, dist.data() + off * gk, r->nq * gk * sizeof(float));
            memcpy(r->cnt, cnt.data() + off, r->nq * sizeof(uint32_t));
          }
          off += r->nq;
        }
        s->n_coalesced_batches += 1;
        s->n_coalesced_queries += total;
      }
      lk.lock();
      for (auto* r : group) {
        r->rc = rc;
        if (rc) snprintf(r->err, sizeof(r->err), "%s", g_err);
        r->done = true;
      }
      s->bq_cv.notify_all();
    }
    s->bq_leader = false;
    s->bq_cv.notify_all();  // a waiter whose request is still queued takes over
  }
  lk.unlock();
  if (me.rc) snprintf(g_err, sizeof(g_err), "%s", me.err);
  return me.rc;
}

int ehx_knn_keys(ehx_space* s, size_t n_queries, const float* queries, uint32_t k, uint64_t* out_ids,
                 float* out_dist, uint32_t* out_count, char* key_arena, size_t arena_cap, uint64_t* key_off) {
  if (!key_off || (!key_arena && arena_cap)) return fail(EHX_EINVAL, "NULL argument");
  int rc = ehx_knn(s, n_queries, queries, k, out_ids, out_dist, out_count);
  if (rc) return rc;
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  uint64_t off = 0;
  std::string key;
  for (size_t i = 0; i < n_queries; ++i) {
    for (uint32_t j = 0; j < k; ++j) {
      key_off[i * k + j] = off;
      if (j < out_count[i] && key_for_id(s, out_ids[i * k + j], &key) == EHX_OK) {
        if (off + key.size() > arena_cap) return fail(EHX_ERANGE, "key arena too small");
        memcpy(key_arena + off, key.data(), key.size());
        off += key.size();
      }
    }
  }
  key_off[n_queries * k] = off;
  return EHX_OK;
}

int ehx_knn_by_key(ehx_space* s, const char* key, size_t klen, uint32_t k, uint64_t* out_ids, float* out_dist,
                   uint32_t* out_count) {
  if (!valid_space(s) || !key || !out_count) return fail(EHX_EINVAL, "NULL argument");
  uint64_t id;
  std::vector<float> v(s->dims);
  {
    std::shared_lock<std::shared_mutex> rl(s->mu);
    if (s->dropped || lookup_key(s, key, klen, &id)) return fail(EHX_ENOTFOUND, "Not found");
  }
  int rc = ehx_get_by_id(s, id, v.data());  // Version::get(key), server.cc:195
  if (rc) return rc;
  const uint32_t kk = k + 1;               // server.cc:198
  std::vector<uint64_t> ids(kk);
  std::vector<float> dist(kk);
  uint32_t cnt = 0;
  if ((rc = ehx_knn(s, 1, v.data(), kk, ids.data(), dist.data(), &cnt))) return rc;
  // server.cc:205-207: erase own key if present, else drop the last
  uint32_t o = 0;
  bool removed = false;
  for (uint32_t j = 0; j < cnt; ++j) {
    if (!removed && ids[j] == id) {
      removed = true;
      continue;
    }
    if (o < k) {
      if (out_ids) out_ids[o] = ids[j];
      if (out_dist) out_dist[o] = dist[j];
      ++o;
    }
  }
  if (!removed && cnt == kk && o == k) { /* last one already dropped by the o<k bound */ }
  *out_count = o;
  return EHX_OK;
}

int ehx_merge_topk_strided_device(void* stream, size_t n_queries, uint32_t k, uint32_t n_lists, const uint64_t* d_ids,
                                  size_t ids_stride, const float* d_dist, size_t dist_stride, const uint32_t* d_count,
                                  size_t count_stride, uint64_t* d_out_ids, float* d_out_dist,
                                  uint32_t* d_out_count) {
  if (n_queries == 0 || k == 0) return EHX_OK;
  if (k > 64 && n_lists > 64) return fail(EHX_EUNSUPPORTED, "merging k > 64 takes at most 64 lists (%u)", n_lists);
  if (!d_ids || !d_dist || !d_out_ids || !d_out_dist) return fail(EHX_EINVAL, "NULL device pointer");
  if (ids_stride % 8 || dist_stride % 4 || count_stride % 4) return fail(EHX_EINVAL, "misaligned list stride");
  int rc = ehx_init(nullptr, 0);
  if (rc) return rc;
  HIP_TRY(launch_merge_lists(d_ids, d_dist, d_count, (uint32_t)n_queries, k, n_lists, d_out_ids, d_out_dist,
                             d_out_count, (hipStream_t)stream, ids_stride, dist_stride, count_stride));
  return EHX_OK;
}

int ehx_merge_topk_device(void* stream, size_t n_queries, uint32_t k, uint32_t n_lists, const uint64_t* d_ids,
                          const float* d_dist, const uint32_t* d_count, uint64_t* d_out_ids, float* d_out_dist,
                          uint32_t* d_out_count) {
  return ehx_merge_topk_strided_device(stream, n_queries, k, n_lists, d_ids, n_queries * k * sizeof(uint64_t), d_dist,
                                       n_queries * k * sizeof(float), d_count, n_queries * sizeof(uint32_t),
                                       d_out_ids, d_out_dist, d_out_count);
}

int ehx_gen_rows_device(void* stream, uint64_t seed, uint64_t row0, uint64_t n_rows, uint32_t dims, int normalize,
                        float* d_out) {
  if (!d_out && n_rows) return fail(EHX_EINVAL, "NULL device pointer");
  if (dims % 4 != 0) return fail(EHX_EINVAL, "ehx_gen_rows_device needs dims %% 4 == 0 (got %u)", dims);
  int rc = ehx_init(nullptr, 0);
  if (rc) return rc;
  HIP_TRY(launch_gen_rows(seed, row0, n_rows, dims, dims, normalize, d_out, (hipStream_t)stream));
  return EHX_OK;
}

namespace {
// rows row0, row0 + stride, ... of dataset `seed` appended to the space (locked exclusively by the caller)
int fill_synthetic_locked(ehx_space* s, uint64_t seed, uint64_t row0, uint64_t n_rows, int normalize, uint64_t stride) {
  if (s->frozen) return fail(EHX_EIMMUTABLE, "Cannot write to immutable space");
  if (!s->implicit_keys && s->n != 0)
    return fail(EHX_EINVAL, "space '%s' already holds keyed rows", s->name.c_str());
  HIP_TRY(hipSetDevice(s->device));
  if (s->n + n_rows >= (1ull << 32)) return fail(EHX_EUNSUPPORTED, "a shard holds at most 2^32-1 rows");
  int rc = grow(s, s->n + n_rows);
  if (rc) return rc;
  s->implicit_keys = true;
  if (s->x_half) {
    // generate fp32 slabs, round them into the fp16 rows
    const uint64_t slab = 1u << 16;
    DevBuf<float> tmp;
    if ((rc = tmp.ensure(std::min<uint64_t>(slab, n_rows) * s->ld))) return rc;
    for (uint64_t r0 = 0; r0 < n_rows; r0 += slab) {
      const uint64_t m = std::min<uint64_t>(slab, n_rows - r0);
      HIP_TRY(launch_gen_rows(seed, row0 + r0 * stride, m, s->dims, s->ld, normalize, tmp.p, s->stream, stride));
      HIP_TRY(launch_store_rows_f16(tmp.p, s->ld, nullptr, s->n + r0, m, s->dims, s->ld, (__half*)s->dX, s->stream));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    tmp.release();
  } else {
    HIP_TRY(launch_gen_rows(seed, row0, n_rows, s->dims, s->ld, normalize, (float*)s->xrow(s->n), s->stream, stride));
    if (s->x_perm) HIP_TRY(launch_permute_blocks((float*)s->dX, s->ld, s->n, n_rows, nullptr, s->stream));
  }
  HIP_TRY(launch_row_stats(s->dX, s->x_half, s->n, n_rows, s->dims, s->ld, s->metric, s->dInv, s->dRowp, s->dMaxSumsq,
                           s->stream, s->x_perm ? 1 : 0));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if ((rc = refresh_scan16(s, s->n, n_rows, nullptr, true, s->n + n_rows))) return rc;
  const uint64_t old_n = s->n;
  s->n += n_rows;
  if (s->params.mode == EHX_MODE_GRAPH && s->g_n == old_n && s->params.build_batch != 0xFFFFFFFFu) {
    if ((rc = graph_insert(s, old_n, n_rows, s->params.build_batch))) return rc;
  }
  return EHX_OK;
}
}  // namespace

int ehx_fill_synthetic(ehx_space* s, uint64_t seed, uint64_t row0, uint64_t n_rows, int normalize) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (n_rows == 0) return EHX_OK;
  std::lock_guard<std::mutex> wg(s->wmu);
  std::unique_lock<std::shared_mutex> wl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (s->keyless) return fail(EHX_EINVAL, "a shard is written through its parent space");
  if (is_parent(s)) return sharded_fill_synthetic(s, seed, row0, n_rows, normalize);
  return fill_synthetic_locked(s, seed, row0, n_rows, normalize, 1);
}

int ehx_graph_import(ehx_space* s, uint64_t n, const uint32_t* level0, const int32_t* levels, uint64_t n_upper,
                     const uint32_t* upper_node, const int32_t* upper_level, const uint64_t* upper_off,
                     const uint32_t* upper_ids, uint32_t entry_point, int32_t max_level) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  std::lock_guard<std::mutex> wg(s->wmu);
  std::unique_lock<std::shared_mutex> wl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (is_parent(s)) return fail(EHX_EUNSUPPORTED, "sharded spaces build their graphs on the GPUs (no import)");
  if (s->params.mode != EHX_MODE_GRAPH) return fail(EHX_EINVAL, "space '%s' is not in graph mode", s->name.c_str());
  if (n != s->n) return fail(EHX_EINVAL, "graph has %llu nodes but the space holds %llu rows", (unsigned long long)n,
                             (unsigned long long)s->n);
  if (n == 0) return EHX_OK;
  if (!level0 || !levels || (n_upper && (!upper_node || !upper_level || !upper_off || !upper_ids)))
    return fail(EHX_EINVAL, "NULL argument");
  if (entry_point >= n || max_level < 0) return fail(EHX_EINVAL, "bad entry point / max level");
  const uint32_t M = s->params.M, M0 = 2 * M;
  if (M0 > 64) return fail(EHX_EUNSUPPORTED, "M=%u: level-0 degree exceeds one wave", M);
  HIP_TRY(hipSetDevice(s->device));
  // host-side re-layout (pure index shuffling, no vector arithmetic)
  std::vector<uint32_t> adj((size_t)n * M0, 0xFFFFFFFFu), up_start(n, 0xFFFFFFFFu);
  for (uint64_t i = 0; i < n; ++i) {
    const uint32_t* row = level0 + i * (1 + M0);
    const uint32_t c = row[0];
    if (c > M0) return fail(EHX_EINVAL, "node %llu: level-0 degree %u > %u", (unsigned long long)i, c, M0);
    for (uint32_t j = 0; j < c; ++j) {
      if (row[1 + j] >= n) return fail(EHX_EINVAL, "node %llu: neighbour id out of range", (unsigned long long)i);
      // a list holds distinct ids (hnswlib invariant); the search kernel's visited test relies on it
      for (uint32_t t = 0; t < j; ++t)
        if (row[1 + t] == row[1 + j])
          return fail(EHX_EINVAL, "node %llu: neighbour %u listed twice at level 0", (unsigned long long)i, row[1 + j]);
      adj[i * M0 + j] = row[1 + j];
    }
  }
  uint64_t total_lists = 0;
  for (uint64_t i = 0; i < n; ++i) {
    if (levels[i] < 0 || levels[i] > max_level) return fail(EHX_EINVAL, "node %llu: bad level", (unsigned long long)i);
    if (levels[i] > 0) {
      up_start[i] = (uint32_t)total_lists;
      total_lists += (uint64_t)levels[i];
    }
  }
  std::vector<uint32_t> lists((size_t)(total_lists ? total_lists : 1) * M, 0xFFFFFFFFu);
  for (uint64_t u = 0; u < n_upper; ++u) {
    const uint32_t node = upper_node[u];
    const int32_t lv = upper_level[u];
    if (node >= n || lv < 1 || lv > levels[node]) return fail(EHX_EINVAL, "upper list %llu: bad node/level", (unsigned long long)u);
    const uint64_t c = upper_off[u + 1] - upper_off[u];
    if (c > M) return fail(EHX_EINVAL, "upper list %llu: degree %llu > %u", (unsigned long long)u, (unsigned long long)c, M);
    uint32_t* dst = &lists[((size_t)up_start[node] + (uint32_t)(lv - 1)) * M];
    for (uint64_t j = 0; j < c; ++j) {
      const uint32_t v = upper_ids[upper_off[u] + j];
      if (v >= n) return fail(EHX_EINVAL, "upper list %llu: neighbour id out of range", (unsigned long long)u);
      dst[j] = v;
    }
  }
  HIP_TRY(hipDeviceSynchronize());
  if (s->dAdj0) (void)hipFree(s->dAdj0);
  if (s->dUpStart) (void)hipFree(s->dUpStart);
  if (s->dUpLists) (void)hipFree(s->dUpLists);
  s->dAdj0 = s->dUpStart = s->dUpLists = nullptr;
  s->g_n = 0;
  HIP_TRY(hipMalloc((void**)&s->dAdj0, adj.size() * 4));
  HIP_TRY(hipMalloc((void**)&s->dUpStart, up_start.size() * 4));
  HIP_TRY(hipMalloc((void**)&s->dUpLists, lists.size() * 4));
  HIP_TRY(hipMemcpy(s->dAdj0, adj.data(), adj.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->dUpStart, up_start.data(), up_start.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->dUpLists, lists.data(), lists.size() * 4, hipMemcpyHostToDevice));
  s->g_n = n;
  s->g_entry = entry_point;
  s->g_maxlevel = max_level;
  s->g_cap_rows = n;  // imported arrays are exactly n rows: re-grown on the next insert
  s->g_lists_cap = total_lists ? total_lists : 1;
  s->g_lists_used = total_lists;
  s->h_levels.assign(levels, levels + n);
  // continue the level sequence where a sequential build of these n nodes would have left it
  s->level_rng.seed((unsigned)s->params.seed);
  s->level_rng_seeded = true;
  for (uint64_t i = 0; i < n; ++i) {
    std::uniform_real_distribution<double> distribution(0.0, 1.0);
    (void)distribution(s->level_rng);
  }
  return EHX_OK;
}

int ehx_graph_export(ehx_space* s, uint32_t* level0, int32_t* levels, uint32_t* up_start, uint32_t* up_lists,
                     uint64_t up_lists_cap, uint64_t* n_lists, uint32_t* entry_point, int32_t* max_level) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (is_parent(s)) return fail(EHX_EUNSUPPORTED, "graph export works per shard");
  if (s->params.mode != EHX_MODE_GRAPH) return fail(EHX_EINVAL, "space '%s' is not in graph mode", s->name.c_str());
  const uint64_t n = s->g_n;
  const uint32_t M = s->params.M, M0 = 2 * M;
  if (n_lists) *n_lists = s->g_lists_used;
  if (entry_point) *entry_point = s->g_entry;
  if (max_level) *max_level = s->g_maxlevel;
  if (n == 0) return EHX_OK;
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipDeviceSynchronize());
  if (level0) {
    std::vector<uint32_t> adj(n * M0);
    HIP_TRY(hipMemcpy(adj.data(), s->dAdj0, adj.size() * 4, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; ++i) {
      uint32_t c = 0;
      for (uint32_t j = 0; j < M0; ++j) {
        const uint32_t v = adj[i * M0 + j];
        level0[i * (1 + M0) + 1 + j] = v == 0xFFFFFFFFu ? 0u : v;
        if (v != 0xFFFFFFFFu) c = j + 1;
      }
      level0[i * (1 + M0)] = c;
    }
  }
  if (levels) memcpy(levels, s->h_levels.data(), n * sizeof(int32_t));
  if (up_start) HIP_TRY(hipMemcpy(up_start, s->dUpStart, n * 4, hipMemcpyDeviceToHost));
  if (up_lists) {
    if (up_lists_cap < s->g_lists_used) return fail(EHX_ERANGE, "upper-list buffer too small");
    if (s->g_lists_used)
      HIP_TRY(hipMemcpy(up_lists, s->dUpLists, s->g_lists_used * M * 4, hipMemcpyDeviceToHost));
  }
  return EHX_OK;
}

int ehx_stats(ehx_space* s, ehx_stats_t* out) {
  if (!valid_space(s) || !out) return fail(EHX_EINVAL, "NULL argument");
  std::shared_lock<std::shared_mutex> rl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (is_parent(s)) {  // the shards' work counters added up; the slowest shard's times (they run concurrently)
    memset(out, 0, sizeof(*out));
    for (ehx_space* c : s->shards) {
      ehx_stats_t t;
      int rc = ehx_stats(c, &t);
      if (rc) return rc;
      out->capacity += t.capacity;
      out->n_dist += t.n_dist;
      out->n_hops += t.n_hops;
      out->n_rerank += t.n_rerank;
      out->n_uncertified += t.n_uncertified;
      out->bytes_algorithmic += t.bytes_algorithmic;
      out->n_filter_queries += t.n_filter_queries;
      out->n_filter_fallback += t.n_filter_fallback;
      out->n_exhaustive += t.n_exhaustive;
      out->n_i8_queries += t.n_i8_queries;
      out->n_i8_fallback += t.n_i8_fallback;
      out->last_scan_ms = std::max(out->last_scan_ms, t.last_scan_ms);
      out->last_total_ms = std::max(out->last_total_ms, t.last_total_ms);
      out->scan_ms_mean = std::max(out->scan_ms_mean, t.scan_ms_mean);
      out->scan_launches = std::max(out->scan_launches, t.scan_launches);
    }
    out->n_rows = s->n;
    out->n_queries = s->n_queries;
    return EHX_OK;
  }
  std::lock_guard<std::mutex> sl(s->scratch_mu);
  HIP_TRY(hipSetDevice(s->device));
  memset(out, 0, sizeof(*out));
  out->n_rows = s->n;
  out->capacity = s->cap;
  out->n_queries = s->n_queries;
  out->n_dist = s->n_dist;
  out->n_rerank = s->n_rerank;
  out->bytes_algorithmic = s->bytes_algo;
  out->n_filter_queries = s->n_filter_queries;
  out->n_filter_fallback = s->n_filter_fallback;
  out->n_exhaustive = s->n_exhaustive;
  out->n_uncertified = s->n_uncertified_final;
  out->n_i8_queries = s->n_i8_queries;
  out->n_i8_fallback = s->n_i8_fallback;
  if (s->dGraphCounters) {
    unsigned long long g[3] = {0, 0, 0};
    HIP_TRY(hipMemcpy(g, s->dGraphCounters, sizeof(g), hipMemcpyDeviceToHost));
    out->n_dist += g[0];
    out->n_hops = g[1] + g[2];
    // n_dist*d*4 + n_hops0*(4+4*2M) + n_hops_up*(4+4*M)
    out->bytes_algorithmic += g[0] * s->dims * 4ull + g[1] * (4ull + 8ull * s->params.M) + g[2] * (4ull + 4ull * s->params.M);
  }
  if (s->dUncert) {
    unsigned long long u[2] = {0, 0};
    HIP_TRY(hipMemcpy(u, s->dUncert, sizeof(u), hipMemcpyDeviceToHost));
    (void)u[0];
    if (u[1]) return fail(EHX_EINTERNAL, "scan kernel tripped its bounded-retry guard %llu times", u[1]);
  }
  {
    // scan times: the space's own ring (graph, fp16 / fp32 engines) and the rings of the int8 pipeline's two scratch sets
    double sum = 0;
    uint64_t got = 0;
    float ms = 0;
    uint64_t newest = 0;  // last_scan_ms / last_total_ms: of the event set that was recorded LAST
    if (s->ev_valid) {
      HIP_TRY(hipEventSynchronize(s->ev[3]));
      newest = s->ev_seq;
      if (hipEventElapsedTime(&ms, s->ev[1], s->ev[2]) == hipSuccess) out->last_scan_ms = ms;
      if (hipEventElapsedTime(&ms, s->ev[0], s->ev[3]) == hipSuccess) out->last_total_ms = ms;
      const uint64_t m = s->ring_count < (uint64_t)ehx_space::kRing ? s->ring_count : (uint64_t)ehx_space::kRing;
      for (uint64_t i = 0; i < m; ++i)
        if (hipEventElapsedTime(&ms, s->ring[i][0], s->ring[i][1]) == hipSuccess) {
          sum += ms;
          ++got;
        }
    }
    for (auto& c : s->i8set) {
      std::lock_guard<std::mutex> cl(c.mu);
      if (!c.ev_valid) continue;
      HIP_TRY(hipEventSynchronize(c.ev[3]));
      if (c.ev_seq > newest) {
        newest = c.ev_seq;
        if (hipEventElapsedTime(&ms, c.ev[1], c.ev[2]) == hipSuccess) out->last_scan_ms = ms;
        if (hipEventElapsedTime(&ms, c.ev[0], c.ev[3]) == hipSuccess) out->last_total_ms = ms;
      }
      const uint64_t m = c.ring_count < 64 ? c.ring_count : 64;
      for (uint64_t i = 0; i < m; ++i)
        if (hipEventElapsedTime(&ms, c.ring[i][0], c.ring[i][1]) == hipSuccess) {
          sum += ms;
          ++got;
        }
    }
    out->scan_launches = got;
    out->scan_ms_mean = got ? sum / (double)got : 0.0;
    (void)hipGetLastError();  // an event that was never recorded is not an error of the caller's next launch
  }
  return EHX_OK;
}

int ehx_graph_counters(ehx_space* s, uint64_t* out, uint32_t n_out) {
  if (!valid_space(s) || !out) return fail(EHX_EINVAL, "NULL argument");
  if (n_out > kGraphCounters) return fail(EHX_EINVAL, "at most %u counters", kGraphCounters);
  if (is_parent(s)) {
    for (uint32_t i = 0; i < n_out; ++i) out[i] = 0;
    uint64_t t[kGraphCounters];
    for (ehx_space* c : s->shards) {
      int rc = ehx_graph_counters(c, t, n_out);
      if (rc) return rc;
      for (uint32_t i = 0; i < n_out; ++i) out[i] += t[i];
    }
    return EHX_OK;
  }
  std::lock_guard<std::mutex> sl(s->scratch_mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  unsigned long long g[kGraphCounters] = {};
  if (s->dGraphCounters) {
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpy(g, s->dGraphCounters, sizeof(g), hipMemcpyDeviceToHost));
  }
  for (uint32_t i = 0; i < n_out; ++i) out[i] = g[i];
  return EHX_OK;
}

int ehx_stats_reset(ehx_space* s) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (is_parent(s)) {
    for (ehx_space* c : s->shards) {
      int rc = ehx_stats_reset(c);
      if (rc) return rc;
    }
    s->n_queries = 0;
    return EHX_OK;
  }
  std::lock_guard<std::mutex> sl(s->scratch_mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  HIP_TRY(hipSetDevice(s->device));
  s->n_queries = 0;
  s->n_dist = 0;
  s->n_rerank = 0;
  s->bytes_algo = 0;
  s->n_filter_queries = 0;
  s->n_filter_fallback = 0;
  s->n_exhaustive = 0;
  s->n_uncertified_final = 0;
  s->n_i8_queries = 0;
  s->n_i8_fallback = 0;
  s->ring_count = 0;
  for (auto& c : s->i8set) {
    std::lock_guard<std::mutex> cl(c.mu);
    c.ring_count = 0;
  }
  if (s->dUncert) HIP_TRY(hipMemset(s->dUncert, 0, 2 * sizeof(unsigned long long)));
  if (s->dGraphCounters) HIP_TRY(hipMemset(s->dGraphCounters, 0, kGraphCounters * sizeof(unsigned long long)));
  return EHX_OK;
}

}  // extern "C"
