// Engine state, error text, HBM residency of a space (capacity doubling, index.cc:29-32), key <-> dense id maps
// (ANNIndex's key_to_label_ / label_to_key_, embeddinghub/embeddingstore/index.h:30-32).
#include "ehx_internal.h"

namespace ehx_impl {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

Engine& engine() {
  static Engine e;
  return e;
}


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


// work enqueued on stream `st` from here on starts after every search of this space that is already in flight (whatever
// stream it was given, whichever scratch set it runs in)
int wait_searches_in_flight(ehx_space* s, hipStream_t st) {
  // (an event recorded on `st` itself orders nothing that the stream's own order does not: no wait packet — three of them per
  // batch of the host pipeline, every batch on the space's stream, were ~10 us of queue time; round 6)
  if (s->ev_valid && s->ev3_stream != st) HIP_TRY(hipStreamWaitEvent(st, s->ev[3], 0));
  for (auto& o : s->i8set)
    if (o.ev_valid && o.ev3_stream != st) HIP_TRY(hipStreamWaitEvent(st, o.ev[3], 0));
  return EHX_OK;
}

int key_for_id(ehx_space* s, uint64_t id, std::string* out) {
  std::shared_lock<std::shared_mutex> kl(s->kmu);
  if (id < s->implicit_n) {
    *out = std::to_string(id);
    return EHX_OK;
  }
  if (id - s->implicit_n < s->id_to_key.size()) {
    *out = s->id_to_key[id - s->implicit_n];
    return EHX_OK;
  }
  return EHX_ENOTFOUND;
}

// the row a decimal key names among the implicitly keyed rows [0, implicit_n) (canonical decimals only: "007" is a key
// of its own)
bool implicit_id(const ehx_space* s, const char* key, size_t klen, uint64_t* id) {
  if (s->implicit_n == 0 || klen == 0 || klen > 20 || (klen > 1 && key[0] == '0')) return false;
  uint64_t v = 0;
  for (size_t i = 0; i < klen; ++i) {
    if (key[i] < '0' || key[i] > '9') return false;
    v = v * 10 + (uint64_t)(key[i] - '0');
  }
  if (v >= s->implicit_n) return false;
  *id = v;
  return true;
}

int lookup_key(ehx_space* s, const char* key, size_t klen, uint64_t* id) {
  std::shared_lock<std::shared_mutex> kl(s->kmu);
  if (implicit_id(s, key, klen, id)) return EHX_OK;
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
    if (implicit_id(s, keys[i], klens[i], &(*ids)[i])) continue;
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

}  // namespace ehx_impl
