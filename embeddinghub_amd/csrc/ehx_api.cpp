// C-ABI of the engine (include/ehx.h): init / shutdown, the space registry, Get, synthetic fill, graph import / export,
// statistics.  The rest of the ABI lives in ehx_write.cpp (Set) and ehx_search.cpp (kNN); ehx_internal.h maps the files.
// No vector arithmetic happens on the host: if the device is unavailable every compute entry point fails with
// EHX_ENODEVICE.
#include "ehx_internal.h"
#include "../../include/ehx_datagen.h"

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
    // the cost is 2 % (profiles/r04_s_*, r04_t_*).  (The two-copy layout of rounds 1-3 stayed selectable by an environment switch until round 6; profiles/r04_q-u hold its A/B.)
    s->x_perm = params && params->mode == EHX_MODE_GRAPH && !s->x_half;
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
  if (!(s->params.search_width <= 2 || s->params.search_width == 4))
    return fail(EHX_EINVAL, "search_width %u: 0, 1, 2 or 4", s->params.search_width);
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
      // [0] rows the int8 filter cannot bound, [1] tiles with a lane group whose B margin matters (L2^2 on rows whose norms vary)
      HIP_TRY(hipMalloc((void**)&s->dUnsafe8, 2 * sizeof(unsigned long long)));
      HIP_TRY(hipMemset(s->dUnsafe8, 0, 2 * sizeof(unsigned long long)));
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
    HIP_TRY(hipEventCreate(&s->ev_end));
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

int ehx_space_set_search_width(ehx_space* s, uint32_t width) {
  if (!valid_space(s) || !(width <= 2 || width == 4)) return fail(EHX_EINVAL, "search_width must be 0, 1, 2 or 4");
  std::unique_lock<std::shared_mutex> wl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  s->params.search_width = width;
  for (ehx_space* c : s->shards) {
    int rc = ehx_space_set_search_width(c, width);
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

int ehx_gen_rows_device(void* stream, uint64_t seed, uint64_t row0, uint64_t n_rows, uint32_t dims, int normalize,
                        float* d_out) {
  if (!d_out && n_rows) return fail(EHX_EINVAL, "NULL device pointer");
  if (dims % 4 != 0) return fail(EHX_EINVAL, "ehx_gen_rows_device needs dims %% 4 == 0 (got %u)", dims);
  int rc = ehx_init(nullptr, 0);
  if (rc) return rc;
  HIP_TRY(launch_gen_rows(seed, row0, n_rows, dims, dims, normalize, d_out, (hipStream_t)stream));
  return EHX_OK;
}

int ehx_gen_manifold_rows_device(void* stream, uint64_t seed, uint64_t row0, uint64_t n_rows, uint32_t dims,
                                 uint32_t latent_dims, int normalize, float* d_out) {
  if (!d_out && n_rows) return fail(EHX_EINVAL, "NULL device pointer");
  if (dims % 4 != 0) return fail(EHX_EINVAL, "ehx_gen_manifold_rows_device needs dims %% 4 == 0 (got %u)", dims);
  if (latent_dims == 0 || latent_dims > EHX_MANIFOLD_MAX_LATENT)
    return fail(EHX_EINVAL, "latent_dims %u outside [1, %u]", latent_dims, EHX_MANIFOLD_MAX_LATENT);
  int rc = ehx_init(nullptr, 0);
  if (rc) return rc;
  HIP_TRY(launch_gen_rows(seed, row0, n_rows, dims, dims, normalize, d_out, (hipStream_t)stream, 1, latent_dims));
  return EHX_OK;
}

}  // extern "C"

namespace ehx_impl {
// rows row0, row0 + stride, ... of dataset `seed` appended to the space (locked exclusively by the caller)
int fill_synthetic_locked(ehx_space* s, uint64_t seed, uint64_t row0, uint64_t n_rows, int normalize, uint64_t stride,
                          uint32_t latent) {
  if (s->frozen) return fail(EHX_EIMMUTABLE, "Cannot write to immutable space");
  if (s->implicit_n != s->n)   // (generated rows form the head of a space: their keys are their decimal row ids)
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
      HIP_TRY(launch_gen_rows(seed, row0 + r0 * stride, m, s->dims, s->ld, normalize, tmp.p, s->stream, stride, latent));
      HIP_TRY(launch_store_rows_f16(tmp.p, s->ld, nullptr, s->n + r0, m, s->dims, s->ld, (__half*)s->dX, s->stream));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    tmp.release();
  } else {
    HIP_TRY(launch_gen_rows(seed, row0, n_rows, s->dims, s->ld, normalize, (float*)s->xrow(s->n), s->stream, stride, latent));
    if (s->x_perm) HIP_TRY(launch_permute_blocks((float*)s->dX, s->ld, s->n, n_rows, nullptr, s->stream));
  }
  HIP_TRY(launch_row_stats(s->dX, s->x_half, s->n, n_rows, s->dims, s->ld, s->metric, s->dInv, s->dRowp, s->dMaxSumsq,
                           s->stream, s->x_perm ? 1 : 0));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if ((rc = refresh_scan16(s, s->n, n_rows, nullptr, true, s->n + n_rows))) return rc;
  const uint64_t old_n = s->n;
  s->n += n_rows;
  {
    std::unique_lock<std::shared_mutex> kl(s->kmu);
    s->implicit_n = s->n;
  }
  if (s->params.mode == EHX_MODE_GRAPH && s->g_n == old_n && s->params.build_batch != 0xFFFFFFFFu) {
    if ((rc = graph_insert(s, old_n, n_rows, s->params.build_batch))) return rc;
  }
  return EHX_OK;
}
}  // namespace ehx_impl

extern "C" {

static int fill_generated(ehx_space* s, uint64_t seed, uint64_t row0, uint64_t n_rows, int normalize, uint32_t latent) {
  if (!valid_space(s)) return fail(EHX_EINVAL, "space is NULL");
  if (n_rows == 0) return EHX_OK;
  std::lock_guard<std::mutex> wg(s->wmu);
  std::unique_lock<std::shared_mutex> wl(s->mu);
  if (s->dropped) return fail(EHX_ENOTFOUND, "Not found");
  if (s->keyless) return fail(EHX_EINVAL, "a shard is written through its parent space");
  if (is_parent(s)) return sharded_fill_synthetic(s, seed, row0, n_rows, normalize, latent);
  return fill_synthetic_locked(s, seed, row0, n_rows, normalize, 1, latent);
}

int ehx_fill_synthetic(ehx_space* s, uint64_t seed, uint64_t row0, uint64_t n_rows, int normalize) {
  return fill_generated(s, seed, row0, n_rows, normalize, 0);
}

int ehx_fill_manifold(ehx_space* s, uint64_t seed, uint64_t row0, uint64_t n_rows, uint32_t latent_dims, int normalize) {
  if (latent_dims == 0 || latent_dims > EHX_MANIFOLD_MAX_LATENT)
    return fail(EHX_EINVAL, "latent_dims %u outside [1, %u]", latent_dims, EHX_MANIFOLD_MAX_LATENT);
  return fill_generated(s, seed, row0, n_rows, normalize, latent_dims);
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
      if (s->end_sampled) {   // graph search: the last TIMED batch (batch 0 and every EHX_STATS_EVERY-th)
        if (s->g_timed_valid) {
          HIP_TRY(hipEventSynchronize(s->ev_end));
          if (s->scan_ev[0] && s->scan_ev[1] && hipEventElapsedTime(&ms, s->scan_ev[0], s->scan_ev[1]) == hipSuccess)
            out->last_scan_ms = ms;
          if (hipEventElapsedTime(&ms, s->ev[0], s->ev_end) == hipSuccess) out->last_total_ms = ms;
        }
      } else {
        if (hipEventElapsedTime(&ms, s->ev[1], s->ev[2]) == hipSuccess) out->last_scan_ms = ms;
        if (hipEventElapsedTime(&ms, s->ev[0], s->ev[3]) == hipSuccess) out->last_total_ms = ms;
      }
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
      if (c.timed_valid) HIP_TRY(hipEventSynchronize(c.ev[2]));   // (recorded right behind ev[3] on timed batches)
      if (c.timed_valid && c.ev_seq > newest) {   // (the set's last TIMED batch: every EHX_STATS_EVERY-th)
        newest = c.ev_seq;
        if (c.last_scan[0] && c.last_scan[1] && hipEventElapsedTime(&ms, c.last_scan[0], c.last_scan[1]) == hipSuccess)
          out->last_scan_ms = ms;
        if (hipEventElapsedTime(&ms, c.ev[0], c.ev[2]) == hipSuccess) out->last_total_ms = ms;
      }
      const uint64_t m = c.ring_count < 64 ? c.ring_count : 64;
      for (uint64_t i = 0; i < m; ++i)
        if (hipEventElapsedTime(&ms, c.ring[i][0], c.ring[i][1]) == hipSuccess) {
          sum += ms;
          ++got;
        }
    }
    if (got == 0 && out->last_scan_ms > 0.0) {   // (only first batches so far: the int8 chain keeps them out of its ring)
      sum = out->last_scan_ms;
      got = 1;
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
  s->g_batches = 0;   // (the next graph batch is a timed one)
  for (auto& c : s->i8set) {
    std::lock_guard<std::mutex> cl(c.mu);
    c.ring_count = 0;
    c.batches = 0;   // (the next batch of the set is a timed one)
  }
  if (s->dUncert) HIP_TRY(hipMemset(s->dUncert, 0, 2 * sizeof(unsigned long long)));
  if (s->dGraphCounters) HIP_TRY(hipMemset(s->dGraphCounters, 0, kGraphCounters * sizeof(unsigned long long)));
  return EHX_OK;
}

}  // extern "C"

