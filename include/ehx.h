/*
 * ehx.h — C ABI of the MI355X-native nearest-neighbour engine ("ehx") that drops in behind
 * featureform/embeddinghub's two vector-store boundaries.
 *
 * This header is the drop-in boundary: plain pointers and sizes, C linkage, no C++/torch
 * types, no callbacks, and the engine never retains a caller pointer after a call returns
 * (cgo rule).  Each entry point cites the reference interface it replaces
 * (paths relative to the reference checkout):
 *
 *   embeddingstore (C++):  embeddinghub/embeddingstore/index.h:19-33, index.cc:10-52 (ANNIndex)
 *                          embeddinghub/embeddingstore/server.cc:172-210 (NearestNeighbor RPC)
 *                          embeddinghub/embeddingstore/embedding_store.proto:9-19 (service)
 *   Go provider:           provider/online.go:42-70 (OnlineStore / VectorStore / VectorStoreTable /
 *                          BatchOnlineTable), provider/redis.go:245-262,454-493
 *   Python offline index:  embeddinghub/sdk/python/offlinehub.py:27-141
 *
 * INTEGRATION.md shows the cgo / C++ / ctypes stubs a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns EHX_OK (0) or a negative EHX_E* code; ehx_last_error() returns a
 *     thread-local message for the last failing call on the calling thread;
 *   - all entry points are re-entrant; ehx_set* is linearizable with respect to a following
 *     ehx_knn* from the same thread (index_test.cc:39-49 relies on it);
 *   - caller allocates every output buffer; inputs are copied before the call returns;
 *   - ids are dense row ids in first-Set order (index.cc:24-28 labels), keys are arbitrary bytes;
 *   - results are nearest first (index.cc:43-50 reverses hnswlib's heap for the same reason);
 *   - all vector arithmetic runs in hand-written HIP kernels on gfx950; there is no CPU
 *     fallback: without a usable device every compute entry point fails with EHX_ENODEVICE.
 */
#ifndef EHX_H_
#define EHX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EHX_ABI_VERSION 5

/* ---- error codes (shim mapping: gRPC status for contract 1, fferr type for contract 2) ---- */
enum {
  EHX_OK = 0,
  EHX_EINVAL = -1,        /* INVALID_ARGUMENT (server.cc:184-188) / fferr.NewInvalidArgumentError   */
  EHX_ENOTFOUND = -2,     /* NOT_FOUND "Not found" (server.cc:178) / DatasetNotFound, EntityNotFound */
  EHX_EEXISTS = -3,       /* ALREADY_EXISTS / fferr.NewDatasetAlreadyExistsError (redis.go:161)       */
  EHX_EIMMUTABLE = -4,    /* FAILED_PRECONDITION "Cannot write to immutable space" (server.cc:125-127) */
  EHX_ENODEVICE = -5,     /* no usable gfx950 device / HIP runtime error: INTERNAL                      */
  EHX_ENOMEM = -6,        /* RESOURCE_EXHAUSTED                                                          */
  EHX_ERANGE = -7,        /* caller buffer too small (key arena)                                         */
  EHX_EUNSUPPORTED = -8,  /* UNIMPLEMENTED (e.g. k above EHX_MAX_K)                                      */
  EHX_EINTERNAL = -9
};

/* ---- metric / dtype / mode ---- */
enum {
  EHX_METRIC_L2SQ = 0,   /* sum (a-b)^2, no sqrt — hnswlib::L2Space, index.cc:13 (embeddingstore default) */
  EHX_METRIC_IP = 1,     /* 1 - sum a*b   — hnswlib::InnerProductSpace                                     */
  EHX_METRIC_COSINE = 2  /* normalise on Set and on query, then IP — Go providers (redis.go:253)           */
};
enum {
  EHX_DTYPE_F32 = 0, /* rows stored as given (the reference's std::vector<float>, index.h:14)                     */
  EHX_DTYPE_F16 = 1  /* rows rounded to IEEE binary16 (nearest-even) when written and widened exactly to fp32
                        wherever they are read — results are those of an F32 space fed the rounded rows; halves
                        the footprint of the stored rows (BASELINE.json configs[5]); the API still speaks fp32
                        (Get returns the widened stored values).  Graph mode: the fp32 search copy is made from
                        the rounded rows (6 instead of 8 bytes per stored dimension)                              */
};
enum {
  EHX_MODE_FLAT = 0,  /* exhaustive scan on the matrix cores + canonical re-rank: exact kNN */
  EHX_MODE_GRAPH = 1  /* HNSW-style level-0 best-first search over an HBM-resident graph      */
};

/* scan engine selection of a flat space (ehx_params.scan, ehx_space_set_scan); results are identical for all */
enum {
  EHX_SCAN_AUTO = 0, /* the fastest certified filter the space supports: int8 where it pays, else fp16 */
  EHX_SCAN_F32 = 1,  /* fp32 matrix-core scan only */
  EHX_SCAN_F16 = 2   /* fp16 matrix-core filter (never the int8 one) */
};
/* what ehx_space_scan_engine reports: the engine that answers first right now */
enum { EHX_ENGINE_F32 = 0, EHX_ENGINE_F16 = 1, EHX_ENGINE_I8 = 2 };

#define EHX_MAX_K 48u /* largest k served by one scan pass (k + 8 slack < 64 candidate slots) */
#define EHX_MAX_K_PAGED 1024u /* flat mode serves EHX_MAX_K < k <= this exactly too, by the exhaustive canonical pass in
                                  pages of 64 results: the whole shard is read once per page and query — fine for the
                                  occasional large request, not a batch path; graph mode serves any k <= this from its
                                  result list of max(ef, k) entries, as hnswlib's searchKnn does */

typedef struct ehx_space ehx_space; /* opaque; owned by the process-global registry */

/* Index parameters.  Zero-initialise and set what you need; 0 means "reference default"
 * (hnswlib defaults as used by index.cc:14-15: M=16, ef_construction=200, seed=100, ef=10). */
typedef struct ehx_params {
  uint32_t mode;            /* EHX_MODE_*                                   */
  uint32_t M;               /* graph degree (level 0 holds 2*M); 2..32, GPU-side insertion needs M <= 31 */
  uint32_t ef_construction; /*                                              */
  uint32_t ef;              /* search ef; effective ef = max(ef, k)         */
  uint64_t seed;            /* level generator seed                         */
  uint64_t initial_capacity;/* rows; 0 -> 128 (index.h:21), doubles on fill */
  uint32_t build_batch;     /* graph mode: rows inserted concurrently per round by bulk loads; 0 = auto
                               (ehx_fill_synthetic: up to 4096 per round; ehx_set / ehx_set_batch: strictly
                               sequential = hnswlib's single-threaded addPoint order, graph bit-identical to
                               the oracle's), 1 = strictly sequential everywhere, N > 1 = rounds of up to N
                               rows, ALSO for an ehx_set_batch made only of fresh keys (hnswlib-python's
                               multi-threaded add_items), 0xFFFFFFFF = no graph building (import one).    */
  uint32_t scan;            /* flat mode: EHX_SCAN_AUTO (0) = a matrix-core FILTER scan (int8 on rows of up to 2048 dims and
                               >= 16 Ki rows, else fp16) in front of the canonical fp32 re-rank; every query a
                               filter cannot certify is re-run by the next engine (int8 -> fp16 -> fp32 scan ->
                               exhaustive canonical pass) — results identical to EHX_SCAN_F32 (1) = fp32
                               matrix-core scan only; EHX_SCAN_F16 (2) = start from the fp16 filter.          */
  uint32_t shards;          /* 0 / 1: one shard on ehx_init's first device.  G > 1: the space is ROW-SHARDED inside this
                               process — row g lives in shard g % G on device_ids[(g % G) % n_devices] of ehx_init;
                               ehx_set* route by row id, ehx_knn* search every shard concurrently, copy each local
                               top-k peer-to-peer (xGMI) to shard 0's device and merge there; ids stay the dense
                               global row ids; k as for an unsharded space.  Graph import / export are per-shard operations.   */
  uint32_t search_width;    /* graph mode: level-0 expansions per search step.  0 / 1: the strict walk — one node at a time
                               in hnswlib's order (searchBaseLayerST): ids, distances and work counters identical to the
                               reference algorithm's on the same graph.  2 / 4: the WIDE walk, a throughput mode — a step
                               expands the 2 / 4 closest unexpanded candidates together (same ef bound, same termination
                               rule, same canonical distances): fewer dependent memory round trips per query, a few per
                               cent more rows fetched, recall@k within 0.005 of the strict walk at equal ef (the gate of
                               tests/test_graph_wide.py); the result's tail may differ from hnswlib's.  Short lists are
                               walked more narrowly whatever is set here: ef < 16 (the reference's default 10 included) is
                               always the strict walk, ef < 64 expands at most 2 per step; graphs of M > 16 are always
                               searched strictly. */
  uint32_t reserved[4];
} ehx_params;

/* Work and time counters, same definitions as the oracle. */
typedef struct ehx_stats_t {
  uint64_t n_rows;           /* rows currently stored                                                */
  uint64_t capacity;         /* rows allocated in HBM                                                */
  uint64_t n_queries;        /* queries served since creation / last reset                           */
  uint64_t n_dist;           /* distances actually evaluated (one per vector row fetched)            */
  uint64_t n_hops;           /* graph nodes expanded (graph mode)                                    */
  uint64_t n_rerank;         /* candidates re-ranked in canonical order                              */
  uint64_t n_uncertified;    /* always 0: a call never returns EHX_OK with an uncertified result (what the matrix-
                                core scans cannot certify is answered by the exhaustive canonical pass; kept in
                                the struct as the audit counter of that invariant)                  */
  uint64_t bytes_algorithmic;/* SURVEY §8d algorithmic bytes of the scans/searches served            */
  double   last_scan_ms;     /* device time of the last scan/search kernel (HIP events)              */
  double   last_total_ms;    /* device time of the last full ehx_knn* pipeline                       */
  double   scan_ms_mean;     /* mean device time of the last <=64 scan/search kernel launches        */
  uint64_t scan_launches;    /* number of launches averaged in scan_ms_mean                          */
  uint64_t n_filter_queries; /* queries answered through the fp16 filter scan                        */
  uint64_t n_filter_fallback;/* ... of which the filter could not certify and the fp32 scan re-ran   */
  uint64_t n_exhaustive;     /* queries answered by the exhaustive canonical pass (fp32 scan uncertified) */
  uint64_t n_i8_queries;     /* queries answered through the int8 filter scan                         */
  uint64_t n_i8_fallback;    /* ... of which it could not certify (pool overflow / margin): next engine */
} ehx_stats_t;

/* ---- process / device ---- */

/* Once per process (idempotent).  device_ids==NULL or n_devices==0 -> device 0.  Unsharded spaces live on the first
 * listed device; the shards of a space created with ehx_params.shards = G are spread over the whole list (one
 * process drives all of them: what a Go / C++ caller of this ABI uses on an 8-GPU node).  The one-process-per-GPU
 * form (torch.distributed + RCCL all-gather, embeddinghub_amd/sharded.py) lists one device per process.
 * With more than one device listed, peer access is opened explicitly between every ordered pair (the shards' exchange
 * step is a peer copy over xGMI); a pair that cannot be opened fails the call with EHX_ENODEVICE naming the two devices
 * (environment EHX_ALLOW_NO_PEER=1: accept it, the copies then stage through host memory). */
int ehx_init(const int* device_ids, int n_devices);
int ehx_shutdown(void);
int ehx_abi_version(void);
const char* ehx_last_error(void);

/* ---- spaces: CreateSpace/DeleteSpace/FreezeSpace (embedding_store.proto:10-12),
 *      VectorStore.CreateIndex/DeleteIndex + OnlineStore.CreateTable/GetTable (online.go:42-59) ---- */
int ehx_space_create(const char* name, size_t name_len, uint32_t dims, int metric, int dtype,
                     const ehx_params* params /* may be NULL */, ehx_space** out);
int ehx_space_open(const char* name, size_t name_len, ehx_space** out); /* EHX_ENOTFOUND if absent */
int ehx_space_drop(ehx_space* s);
int ehx_space_freeze(ehx_space* s);            /* later ehx_set* -> EHX_EIMMUTABLE (version.cc:47-50) */
int ehx_space_size(ehx_space* s, uint64_t* n); /* number of distinct keys                              */
int ehx_space_dims(ehx_space* s, uint32_t* dims);
int ehx_space_reserve(ehx_space* s, uint64_t rows);
int ehx_space_set_ef(ehx_space* s, uint32_t ef);
/* graph spaces: expansions per search step from now on (ehx_params.search_width: 0 / 1 strict, 2 / 4 wide); EHX_EINVAL
 * for any other value.  The graph is the same for every width: only the order it is walked in changes. */
int ehx_space_set_search_width(ehx_space* s, uint32_t width);
/* flat spaces: switch between EHX_SCAN_AUTO (certified filter scan, the default), EHX_SCAN_F16 and EHX_SCAN_F32.
 * Results are identical; A/B measurement and diagnosis.  The scan copies stay current whatever is selected. */
int ehx_space_set_scan(ehx_space* s, uint32_t scan);
/* the engine that answers first on this space right now: EHX_ENGINE_* */
int ehx_space_scan_engine(ehx_space* s, uint32_t* engine);

/* ---- writes: Set/MultiSet (server.cc:113-149 -> Version::set -> ANNIndex::set, index.cc:20-37),
 *      OnlineStoreTable.Set / BatchOnlineTable.BatchSet (online.go:50-53,66-70) ---- */
int ehx_set(ehx_space* s, const char* key, size_t klen, const float* vec /* dims floats */);
int ehx_set_batch(ehx_space* s, size_t n, const char* const* keys, const size_t* klens,
                  const float* vecs /* n x dims, row-major */);

/* ---- reads: Get (server.cc:95-111), OnlineStoreTable.Get (online.go:52) ---- */
int ehx_get(ehx_space* s, const char* key, size_t klen, float* out_vec /* dims floats */);
int ehx_get_by_id(ehx_space* s, uint64_t id, float* out_vec);
/* key of a row id: copies up to cap bytes, stores the full length in *klen */
int ehx_key_of(ehx_space* s, uint64_t id, char* out_key, size_t cap, size_t* klen);

/* ---- kNN: ANNIndex::approx_nearest (index.cc:39-52), NearestNeighbor RPC (server.cc:172-210),
 *      VectorStoreTable.Nearest (online.go:63), offlinehub.Index.nearest_neighbor (offlinehub.py:102-131).
 *      Batched: n_queries x dims row-major in, n_queries x k out, nearest first.
 *      out_count[i] <= k is the number of valid results of query i (fewer than k only when the
 *      space holds fewer than k rows — the reference has UB there, index.cc:46-50). ---- */
int ehx_knn(ehx_space* s, size_t n_queries, const float* queries, uint32_t k, uint64_t* out_ids,
            float* out_dist, uint32_t* out_count);
/* as ehx_knn, plus keys: key j of query i is key_arena[key_off[i*k+j] .. key_off[i*k+j+1]) ;
 * key_off has n_queries*k+1 entries; missing results have empty keys. */
int ehx_knn_keys(ehx_space* s, size_t n_queries, const float* queries, uint32_t k, uint64_t* out_ids,
                 float* out_dist, uint32_t* out_count, char* key_arena, size_t arena_cap,
                 uint64_t* key_off);
/* server.cc:193-207 / offlinehub.py:113-130: query by stored key, ask for k+1, drop the key
 * itself if present else drop the last.  EHX_ENOTFOUND for an unknown key (the reference has UB). */
int ehx_knn_by_key(ehx_space* s, const char* key, size_t klen, uint32_t k, uint64_t* out_ids,
                   float* out_dist, uint32_t* out_count);
/* as ehx_knn_by_key, plus the neighbours' KEYS in one call — the whole NearestNeighbor RPC by key (server.cc:172-210:
 * look the key up, search k+1, drop the key itself, answer with keys) without k calls of ehx_key_of: key j is
 * key_arena[key_off[j] .. key_off[j+1]), key_off has k+1 entries; EHX_ERANGE if the arena is too small.
 * out_ids / out_dist may be NULL. */
int ehx_knn_by_key_keys(ehx_space* s, const char* key, size_t klen, uint32_t k, uint64_t* out_ids, float* out_dist,
                        uint32_t* out_count, char* key_arena, size_t arena_cap, uint64_t* key_off);

/* ---- device-resident entry points (inputs/outputs already in HBM; `stream` is a hipStream_t
 *      passed as void*, NULL = default stream).  These are what a batching shim and bench.py
 *      call; ehx_knn above is ehx_knn_device plus the H2D/D2H copies.
 *      Completion: the results are in place when `stream` reaches the point the call returned at — wait for the stream
 *      (or an event recorded on it) before reading them.  Most paths have waited for the device themselves by the time
 *      they return (every flat stage reads its verdict back); ONE query against a small flat shard (the exhaustive
 *      canonical pass, EHX_SMALL_EXACT_BYTES) and graph searches return with work still queued on `stream`.  A writer
 *      that rewrites rows in place waits for searches in flight on whatever stream they were given; a writer that only
 *      appends does not need to (it touches rows beyond the row count the search was launched with, and growing the
 *      arrays takes the space exclusively and drains the device first). ---- */
int ehx_knn_device(ehx_space* s, void* stream, size_t n_queries, const float* d_queries, uint32_t k,
                   uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count);
/* k-way merge of per-shard results (RCCL all-gather output): lists laid out
 * [n_lists][n_queries][k]; ids must already be global.  Ordered by (dist, id). */
int ehx_merge_topk_device(void* stream, size_t n_queries, uint32_t k, uint32_t n_lists,
                          const uint64_t* d_ids, const float* d_dist, const uint32_t* d_count,
                          uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_count);
/* Same, with list l of each array starting l * stride BYTES after list 0 — e.g. the three arrays of every
 * shard packed into one buffer so that ONE all-gather moves them (embeddinghub_amd/sharded.py). */
int ehx_merge_topk_strided_device(void* stream, size_t n_queries, uint32_t k, uint32_t n_lists,
                                  const uint64_t* d_ids, size_t ids_stride, const float* d_dist, size_t dist_stride,
                                  const uint32_t* d_count, size_t count_stride, uint64_t* d_out_ids,
                                  float* d_out_dist, uint32_t* d_out_count);

/* ---- synthetic workload (EHX-GAUSS-1, include/ehx_datagen.h) ---- */
/* Appends rows [row0, row0+n_rows) of dataset `seed` to the space, generated on the device;
 * normalize!=0 L2-normalises each generated row first.  Rows get implicit decimal keys. */
int ehx_fill_synthetic(ehx_space* s, uint64_t seed, uint64_t row0, uint64_t n_rows, int normalize);
int ehx_gen_rows_device(void* stream, uint64_t seed, uint64_t row0, uint64_t n_rows, uint32_t dims,
                        int normalize, float* d_out /* n_rows x dims */);
/* EHX-MANIFOLD-1 (include/ehx_datagen.h): rows WITH STRUCTURE — points of a latent_dims-dimensional (1..64) linear
 * subspace of the dims-dimensional space plus 5 % isotropic noise — the workload on which a graph index meets a recall
 * target at a small ef (isotropic Gaussian rows at d = 768 defeat any graph: SURVEY.md §7).  Generated on
 * the device like ehx_fill_synthetic — a 10 M x 768 corpus never exists on the host; the host-side restatement
 * (oracle/datagen_oracle.hpp) produces the same bytes. */
int ehx_fill_manifold(ehx_space* s, uint64_t seed, uint64_t row0, uint64_t n_rows, uint32_t latent_dims, int normalize);
int ehx_gen_manifold_rows_device(void* stream, uint64_t seed, uint64_t row0, uint64_t n_rows, uint32_t dims,
                                 uint32_t latent_dims, int normalize, float* d_out /* n_rows x dims */);

/* ---- graph import (graph mode; persistence and strict-parity checks).  Graph spaces also build their
 * graph on the GPU as rows are Set (hnswlib addPoint semantics, k_insert.hip). ----
 * level0: n x (1 + 2M) u32 rows = [count, ids...]; levels: n ints; upper lists are passed as a CSR
 * over (node, level>=1): upper_off has n_upper+1 entries into upper_ids, upper_node/upper_level
 * name each list. */
int ehx_graph_import(ehx_space* s, uint64_t n, const uint32_t* level0, const int32_t* levels,
                     uint64_t n_upper, const uint32_t* upper_node, const int32_t* upper_level,
                     const uint64_t* upper_off, const uint32_t* upper_ids, uint32_t entry_point,
                     int32_t max_level);

/* Export of the graph built on the GPU (or imported): level0 [n][1+2M] u32 = (count, ids...), levels [n],
 * up_start [n] (index of a node's first upper list or 0xFFFFFFFF; lists of levels 1..L are consecutive),
 * up_lists [n_lists][M] padded with 0xFFFFFFFF.  Any output pointer may be NULL. */
int ehx_graph_export(ehx_space* s, uint32_t* level0, int32_t* levels, uint32_t* up_start, uint32_t* up_lists,
                     uint64_t up_lists_cap, uint64_t* n_lists, uint32_t* entry_point, int32_t* max_level);

/* ---- stats ---- */
int ehx_stats(ehx_space* s, ehx_stats_t* out);
int ehx_stats_reset(ehx_space* s);
/* Graph-search work counters since the last reset, the terms of SURVEY §8d's bytes-per-query formula plus a
 * kernel diagnostic: out[0] = rows fetched (n_dist), out[1] = level-0 expansions, out[2] = upper-level
 * expansions, out[3] = level-0 expansions whose adjacency row and visited words had been requested one
 * expansion ahead (k_graph.hip; wide walk: whose adjacency row the previous step had predicted and requested);
 * out[4] = steps of the wide walk (out[1] / out[4] = expansions per step); out[4..11] are phase timers in
 * -DEHX_GRAPH_PROFILE ablation builds (strict walk).  n_out <= 12.  Zeros for a space that never ran a graph search. */
int ehx_graph_counters(ehx_space* s, uint64_t* out, uint32_t n_out);

#ifdef __cplusplus
}
#endif
#endif /* EHX_H_ */
