// Internal declarations shared by the translation units of the C-ABI library (not installed, not part of the ABI):
//   ehx_space.cpp   engine state, error text, HBM residency and capacity doubling, key <-> id maps
//   ehx_flat.cpp    the exact flat chain: scan plans, fp32 / fp16 / int8 pipelines, exhaustive pass, engine fall-through
//   ehx_graph.cpp   graph mode: GPU-side insertion, update-in-place repair, the graph search pipeline
//   ehx_shards.cpp  row-sharded spaces inside one process (ehx_params.shards)
//   ehx_write.cpp   Set / BatchSet: staging, upload, row statistics, scan copies, write combiner
//   ehx_search.cpp  ehx_knn_device / ehx_knn (host pointers, micro-batcher) / keys / merge
//   ehx_api.cpp     init, registry, Get, synthetic fill, graph import / export, statistics
#pragma once
// (was the head of ehx_api.cpp) C-ABI of the engine (include/ehx.h): process-global space registry, key <-> dense id map
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


struct ehx_space;

namespace ehx_impl {

extern thread_local char g_err[512];   // text of the calling thread's last error

int fail(int code, const char* fmt, ...);   // sets the thread's error text (ehx_last_error), returns `code`

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
Engine& engine();   // the process-global engine state (ehx_space.cpp)

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

}  // namespace ehx_impl
using namespace ehx_impl;

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
  bool implicit_keys = false;  // rows appended by ehx_fill_synthetic / ehx_fill_manifold: key == decimal row id ...
  uint64_t implicit_n = 0;     // ... for rows [0, implicit_n); rows Set afterwards (unsharded spaces) carry their own keys:
                               // id_to_key[id - implicit_n].  A Set of the key "123" on such a space rewrites row 123.
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
  uint64_t cap = 0;
  // published row count.  Atomic (round 6): an appending Set publishes its rows with ONE release store under the space's lock
  // held SHARED — rows below the new count are resident and described before the store, the arrays do not move while any search
  // holds the lock shared — where it used to take the lock exclusively "for the length of one store": glibc's rwlock prefers
  // readers, two pipelined search callers overlap without a gap, and the writer waited ~4 batches per chunk (12.5 M x 1536
  // under search: 63 ms per 8192-row chunk against 3 un-contended).  A search may read the count more than once; every use
  // tolerates a later, larger value (more rows valid than tiles scanned: "a search sees a prefix of the completed Sets").
  std::atomic<uint64_t> n{0};
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
  uint64_t h_margin8 = 0;      // tiles written so far with a lane group whose min B lies > 0.1 % above the tile's (dUnsafe8[1])
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
    DevBuf<uint32_t> dI8Ctl;  // [q_rows] pool counts | [q_rows] overflow flags | [kSyncWordsI8] lock-step progress words
    DevBuf<uint32_t> dUflags;
    DevBuf<uint64_t> dCnt;    // [8] epilogue counters of diagnosis builds (EHX_I8_COUNT); the set's own: nothing shared
    unsigned long long* dUncert = nullptr;
    unsigned long long* hUncertPin = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // start of the last TIMED batch | (unused) | its end | all enqueued
                                                              // work done (every batch: what writers and other streams wait for)
    // Timing events are recorded on every EHX_STATS_EVERY-th batch of the set only (round 6: an event record between two
    // kernels idles the queue ~6 us — start, scan start, scan end and end were 24 us of a 0.9-ms batch): batches N - 1, 2 N - 1
    // ... go into the ring behind scan_ms_mean; batch 0 is timed too (a caller that runs one batch and asks) but stays out of
    // the ring — the first batch behind a reset starts on an idle queue and ran 5-10 % long in the bench's 10-step runs
    hipStream_t ev3_stream = nullptr;   // the stream ev[3] was last recorded on (work queued there later is behind it anyway)
    uint64_t batches = 0;        // batches run in this set since the last ehx_stats_reset
    bool timed_valid = false;    // ev[0] / ev[2] / last_scan[] hold a recorded batch
    hipEvent_t last_scan[2] = {nullptr, nullptr};            // scan start / end of the last batch: two of ring[][]'s events
    hipEvent_t verdict = nullptr;                            // blocking-sync: the verdict has landed in hUncertPin
    std::atomic<bool> ev_valid{false};
    uint64_t ev_seq = 0;     // value of ehx_space::ev_counter when ev[] was last recorded (ehx_stats: which set is newest)
    hipEvent_t ring[64][2] = {};
    uint64_t ring_count = 0;
    hipEvent_t first_pair[2] = {nullptr, nullptr};   // scan start / end of the set's FIRST batch after a reset (not in the ring)
    std::mutex mu;
  };
  I8Set i8set[2];
  std::atomic<uint64_t> ev_counter{0};
  std::atomic<uint32_t> i8_next_set{0};
  std::mutex i8_enqueue_mu;  // held while ONE host batch's int8 stage is enqueued on the space's stream (not while its
                             // verdict is awaited): two callers in different scratch sets must not interleave their
                             // launches — the batches' kernels would alternate on the stream and every per-batch scan
                             // time (the timing ring, ehx_stats) would span both
  std::atomic<uint64_t> n_filter_queries{0}, n_filter_fallback{0}, n_exhaustive{0}, n_uncertified_final{0};
  std::atomic<uint64_t> n_i8_queries{0}, n_i8_fallback{0};
  float* hStage = nullptr;  // pinned staging (Set / Get / query upload)
  size_t hStageBytes = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t ev3_stream = nullptr;   // the stream ev[3] was last recorded on
  // graph search (round 6): timing events on batch 0 and on every EHX_STATS_EVERY-th batch only — six event records per batch
  // idled the queue ~36 us of a 0.35-ms batch.  scan_ev: the timed batch's scan window (a ring pair, or ev[1] / ev[2] for
  // batch 0, which stays out of the ring); ev_end: recorded behind ev[3] on timed batches; end_sampled: the last batch was
  // a graph search (the other engines record ev[0..3] on every batch and clear it)
  hipEvent_t scan_ev[2] = {nullptr, nullptr};
  hipEvent_t ev_end = nullptr;
  bool end_sampled = false, g_timed_valid = false;
  uint64_t g_batches = 0;
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
      for (auto& e : c.first_pair) {
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
    if (ev_end) (void)hipEventDestroy(ev_end);
    ev_end = nullptr;
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


namespace ehx_impl {
// ---- ehx_space.cpp ----
int ensure_stage(ehx_space* s, size_t bytes);
int grow(ehx_space* s, uint64_t rows);
int ensure_rows(ehx_space* s, uint64_t rows);
inline bool valid_space(ehx_space* s) { return s != nullptr; }
inline bool is_parent(const ehx_space* s) { return !s->shards.empty(); }

// work enqueued on stream `st` from here on starts after every search of this space that is already in flight (whatever
// stream it was given, whichever scratch set it runs in)
int wait_searches_in_flight(ehx_space* s, hipStream_t st);
int key_for_id(ehx_space* s, uint64_t id, std::string* out);
int lookup_key(ehx_space* s, const char* key, size_t klen, uint64_t* id);
bool implicit_id(const ehx_space* s, const char* key, size_t klen, uint64_t* id);  // (caller holds kmu or mu)
void resolve_keys(ehx_space* s, size_t n, const char* const* keys, const size_t* klens, std::vector<uint64_t>* ids,
                         uint64_t* next_out, std::vector<std::string>* new_keys);

// ---- ehx_graph.cpp ----
int graph_ensure_arrays(ehx_space* s);
int graph_ensure_lists(ehx_space* s, uint64_t lists);
int graph_insert(ehx_space* s, uint64_t id0, uint64_t count, uint32_t batch);
int graph_update(ehx_space* s, uint32_t id);
struct GraphOneLaunch {   // one query per call in one launch (knn_graph_locked)
  const float* q_host;     // the raw query, host-visible
  uint32_t* done_flag;     // host-visible; the kernel stores `seq` there when the results are written
  uint32_t seq;
};
int knn_graph_locked(ehx_space* s, hipStream_t st, size_t nq, const float* d_queries, uint32_t k, uint64_t* d_ids,
                     float* d_dist, uint32_t* d_count, const GraphOneLaunch* one = nullptr);

// ---- ehx_flat.cpp ----
int flat_pass(ehx_space* s, hipStream_t st, size_t nq, const float* d_queries, uint32_t k, uint64_t* d_ids,
              float* d_dist, uint32_t* d_count, bool f16, bool count_stats);
int resolve_engine(const ehx_space* s);
int flat_pass8(ehx_space* s, int set, hipStream_t st, size_t nq, const float* d_queries, uint32_t k, uint64_t* d_ids,
               float* d_dist, uint32_t* d_count, bool count_stats, uint32_t* kprime_used = nullptr);
int exhaustive_pass(ehx_space* s, hipStream_t st, size_t nq, const float* d_queries, uint32_t k, uint64_t* d_ids,
                    float* d_dist, uint32_t* d_count);
void i8_adapt(ehx_space* s, size_t nq, size_t n_failed, size_t n_short, uint32_t kprime);
int knn_device_locked(ehx_space* s, hipStream_t st, size_t nq, const float* d_queries, uint32_t k,
                      uint64_t* d_ids, float* d_dist, uint32_t* d_count, const std::vector<uint32_t>* i8_failed = nullptr,
                      size_t i8_short = 0, uint32_t i8_kprime_in = 0);

// ---- ehx_write.cpp ----
int sync_stream(ehx_space* s, hipStream_t st);
int refresh_scan16(ehx_space* s, uint64_t row0, uint64_t n, hipStream_t st, bool exclusive, uint64_t n_after);
int write_rows_locked_fwd(ehx_space* s, size_t n, const std::vector<uint64_t>& ids, uint64_t next, const float* vecs);

// ---- ehx_api.cpp ----
int fill_synthetic_locked(ehx_space* s, uint64_t seed, uint64_t row0, uint64_t n_rows, int normalize, uint64_t stride,
                          uint32_t latent = 0);

// ---- ehx_shards.cpp ----
int sharded_set_batch(ehx_space* p, size_t n, const char* const* keys, const size_t* klens, const float* vecs);
int sharded_fill_synthetic(ehx_space* p, uint64_t seed, uint64_t row0, uint64_t n_rows, int normalize, uint32_t latent = 0);
int sharded_knn(ehx_space* p, size_t nq, const float* h_queries, const float* d_queries, int qdev, uint32_t k,
                uint64_t* out_ids, float* out_dist, uint32_t* out_count, bool out_on_device, hipStream_t caller_stream);

}  // namespace ehx_impl
