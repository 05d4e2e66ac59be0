// Environment knobs of the engine, ALL of them, read ONCE per process (at ehx_init, or at the first call that needs one
// — whichever comes first) into one table.  None is part of the C ABI's contract: they exist for A/B runs, sweeps and
// diagnosis, every default is the measured best, and INTEGRATION.md §"Environment knobs" lists them with what they do.
// Nothing else in csrc/ calls getenv.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace ehx {

struct Env {
  // ---- engine selection ----
  bool scan_f32 = false;          // EHX_SCAN=f32: every space scans in fp32 (no filter copies)
  bool scan_f16 = false;          // EHX_SCAN=f16: no int8 scan copy (the fp16 filter serves)
  uint64_t small_exact_bytes = 512ull << 20;  // EHX_SMALL_EXACT_BYTES: one query against a shard up to this size takes the
                                  // exhaustive canonical pass (0: never)
  bool one_launch = true;         // EHX_ONE_LAUNCH=0: that pass as three launches instead of one
  bool host_pipeline = true;      // EHX_HOST_PIPELINE=0: host batches take the pipeline lock for the whole call
  bool allow_no_peer = false;     // EHX_ALLOW_NO_PEER=1: in-process shards without peer access (copies stage through the host)
  // ---- int8 filter scan ----
  uint32_t i8_growth = 0;         // EHX_I8_GROWTH [2, 64]: rows of pass i+1 / rows of pass i (0: automatic, ehx_flat.cpp)
  double i8_safety = 2.0;         // EHX_I8_SAFETY >= 1: slack of the rank a middle pass's threshold is taken at
  int i8_sync = 0;                // EHX_I8_SYNC: 0 off, N > 0 lock-step by tile with tolerance N tiles
  long i8_kprime = 0;             // EHX_I8_KPRIME >= 64: fixed logical length of the candidate list (0: automatic)
  uint32_t i8_first_tiles = 0;    // EHX_I8_FIRST_TILES [64, 65536]: tiles of the cascade's first pass (0: automatic)
  uint64_t i8_first_keys = 0;     // EHX_I8_FIRST_KEYS: keys per query the first pass aims for (0: automatic)
  uint32_t i8_width = 0;          // EHX_I8_WIDTH = 256 | 512 | 1024: initial width of the list (0: by row length)
  uint64_t i8_min_rows = 0;       // EHX_I8_MIN_ROWS >= 4096: below this many rows the fp16 filter serves (0: 16384)
  bool i8_sort = true;            // EHX_I8_SORT=0: tiles keep their row order (no per-tile ordering by quantisation step)
  bool i8_trace = false;          // EHX_I8_TRACE: every adaptation of the candidate list on stderr
  bool i8_count = false;          // EHX_I8_COUNT: (builds with -DEHX_I8_COUNT=1) epilogue counters per batch on stderr
  bool i8_debug = false;          // EHX_I8_DEBUG: what the uncertified queries of a batch look like, on stderr
  bool i8_qres = true;            // EHX_I8_QRES=0: short rows through the query ring instead of the resident query tile
  bool i8_half = true;            // EHX_I8_HALF=0: rows of <= 128 dims through full-tile workgroups (one per CU)
  uint32_t stats_every = 2;       // EHX_STATS_EVERY [1, 1024]: the int8 chain brackets the scan phase of every N-th batch of a scratch set with timing events
  bool i8_groupb = true;          // EHX_I8_GROUPB=0: L2^2 spaces scan under one min B per tile (no per-group B margins)
  uint32_t i8_skew = 64;          // EHX_I8_SKEW: half-tile workgroups: start skew of a SIMD's second wave, x 64 cycles (0: none)
  bool rerank_staged = true;      // EHX_RERANK_STAGED=0: every lane of the re-rank walks its own row
  // ---- graph mode ----
  uint64_t build_div = 0;         // EHX_BUILD_DIV >= 2: a bulk-build round is at most 1/DIV of the graph it joins
  bool build_trace = false;       // EHX_BUILD_TRACE: build progress on stderr (a stream sync every 128 rounds)
  long long build_scratch_keep = -1;  // EHX_BUILD_SCRATCH_KEEP: bytes of scratch a bulk build may keep (-1: automatic)
  bool graph_vislog = true;       // EHX_GRAPH_VISLOG=0: visited bitmaps cleared by a memset per batch, never by the visit log
  int graph_help = -1;            // EHX_GRAPH_HELP = 0 | 1: the wide walk's helper wave off / on whatever the shape (-1: by shape)
  uint32_t graph_width = 0;       // EHX_GRAPH_WIDTH = 1 | 2 | 4: expansions per step of every graph search (0: the space's search_width)
};

inline const Env& env() {
  static const Env e = [] {
    Env v;
    auto str = [](const char* n) { return getenv(n); };
    auto flag = [&](const char* n, bool dflt) {   // "0" = off, anything else = on
      const char* g = str(n);
      return g ? atoi(g) != 0 : dflt;
    };
    if (const char* g = str("EHX_SCAN")) {
      v.scan_f32 = !strcmp(g, "f32");
      v.scan_f16 = !strcmp(g, "f16");
    }
    if (const char* g = str("EHX_SMALL_EXACT_BYTES")) v.small_exact_bytes = strtoull(g, nullptr, 10);
    v.one_launch = flag("EHX_ONE_LAUNCH", true);
    v.host_pipeline = flag("EHX_HOST_PIPELINE", true);
    v.allow_no_peer = flag("EHX_ALLOW_NO_PEER", false);
    if (const char* g = str("EHX_I8_GROWTH")) {
      const long x = atol(g);
      v.i8_growth = (uint32_t)(x < 2 ? 2 : (x > 64 ? 64 : x));
    }
    if (const char* g = str("EHX_I8_SAFETY")) {
      const double x = atof(g);
      v.i8_safety = x < 1.0 ? 1.0 : x;
    }
    if (const char* g = str("EHX_I8_SYNC")) {
      const int x = atoi(g);
      v.i8_sync = x < 0 ? 0 : (x > 64 ? 64 : x);
    }
    if (const char* g = str("EHX_I8_KPRIME")) v.i8_kprime = atol(g);
    if (const char* g = str("EHX_I8_FIRST_TILES")) {
      const long x = atol(g);
      v.i8_first_tiles = (uint32_t)(x < 64 ? 64 : (x > 65536 ? 65536 : x));
    }
    if (const char* g = str("EHX_I8_FIRST_KEYS")) {
      const long x = atol(g);
      v.i8_first_keys = (uint64_t)(x < 0 ? 0 : x);
    }
    if (const char* g = str("EHX_I8_WIDTH")) {
      const long x = atol(g);
      if (x == 256 || x == 512 || x == 1024) v.i8_width = (uint32_t)x;
    }
    if (const char* g = str("EHX_I8_MIN_ROWS")) {
      const uint64_t x = strtoull(g, nullptr, 10);
      v.i8_min_rows = x < 4096 ? 4096 : x;
    }
    v.i8_sort = flag("EHX_I8_SORT", true);
    v.i8_trace = str("EHX_I8_TRACE") != nullptr;
    v.i8_count = str("EHX_I8_COUNT") != nullptr;
    v.i8_debug = str("EHX_I8_DEBUG") != nullptr;
    v.i8_qres = flag("EHX_I8_QRES", true);
    v.i8_half = flag("EHX_I8_HALF", true);
    v.i8_groupb = flag("EHX_I8_GROUPB", true);
    if (const char* g = str("EHX_STATS_EVERY")) {
      const long x = atol(g);
      v.stats_every = (uint32_t)(x < 1 ? 1 : (x > 1024 ? 1024 : x));
    }
    if (const char* g = str("EHX_I8_SKEW")) {
      const long x = atol(g);
      v.i8_skew = (uint32_t)(x < 0 ? 0 : (x > 4096 ? 4096 : x));
    }
    v.rerank_staged = flag("EHX_RERANK_STAGED", true);
    if (const char* g = str("EHX_BUILD_DIV")) {
      const long x = atol(g);
      v.build_div = (uint64_t)(x < 0 ? 0 : x);
    }
    v.build_trace = str("EHX_BUILD_TRACE") != nullptr;
    if (const char* g = str("EHX_BUILD_SCRATCH_KEEP")) v.build_scratch_keep = atoll(g);
    v.graph_vislog = flag("EHX_GRAPH_VISLOG", true);
    if (const char* g = str("EHX_GRAPH_HELP")) v.graph_help = atoi(g) != 0;
    if (const char* g = str("EHX_GRAPH_WIDTH")) {
      const long x = atol(g);
      if (x == 1 || x == 2 || x == 4) v.graph_width = (uint32_t)x;
    }
    return v;
  }();
  return e;
}

}  // namespace ehx
