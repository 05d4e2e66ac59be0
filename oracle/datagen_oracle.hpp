// ORACLE — TEST INFRASTRUCTURE ONLY (see hnsw_oracle.hpp header).
//
// Host-side restatement of the synthetic-data generator specified in
// include/ehx_datagen.h ("EHX-GAUSS-1"), written independently of the device
// implementation in embeddinghub_amd/csrc so the two can be checked bit-for-bit
// against each other (tests/test_datagen.py).  Workload definition: SURVEY.md §8d
// (Philox4x32-10 counter RNG -> Box-Muller N(0,1), corpus seed 20250211, query seed
// 20250212, optional L2 normalisation with x*(1/(sqrt(sum x^2)+1e-30f))).
//
// Every floating-point step is either a single IEEE operation or an explicit fmaf,
// so the result does not depend on the host libm; compile with -ffp-contract=off.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

namespace oracle {
namespace datagen {

struct U4 {
  uint32_t v[4];
};

// Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants).
static inline U4 philox4x32_10(U4 ctr, uint32_t k0, uint32_t k1) {
  const uint64_t M0 = 0xD2511F53ull, M1 = 0xCD9E8D57ull;
  for (int round = 0; round < 10; round++) {
    uint64_t p0 = M0 * ctr.v[0];
    uint64_t p1 = M1 * ctr.v[2];
    U4 nxt;
    nxt.v[0] = (uint32_t)(p1 >> 32) ^ ctr.v[1] ^ k0;
    nxt.v[1] = (uint32_t)p1;
    nxt.v[2] = (uint32_t)(p0 >> 32) ^ ctr.v[3] ^ k1;
    nxt.v[3] = (uint32_t)p0;
    ctr = nxt;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return ctr;
}

static inline float bits_to_float(uint32_t b) {
  float f;
  std::memcpy(&f, &b, 4);
  return f;
}
static inline uint32_t float_to_bits(float f) {
  uint32_t b;
  std::memcpy(&b, &f, 4);
  return b;
}

// ln(u) for u in (0,1): exponent/mantissa split + degree-18 Taylor of ln(1+f),
// |f| <= sqrt(2)-1, Horner with fmaf.
static inline float det_log(float u) {
  uint32_t b = float_to_bits(u);
  int e = (int)(b >> 23) - 127;
  float m = bits_to_float((b & 0x007FFFFFu) | 0x3F800000u);
  if (m > 0x1.6a09e6p+0f) {
    m = m * 0.5f;
    e = e + 1;
  }
  float f = m - 1.0f;
  static const float c[19] = {0.0f,
                              0x1.000000p+0f,  -0x1.000000p-1f, 0x1.555556p-2f,  -0x1.000000p-2f,
                              0x1.99999ap-3f,  -0x1.555556p-3f, 0x1.24924ap-3f,  -0x1.000000p-3f,
                              0x1.c71c72p-4f,  -0x1.99999ap-4f, 0x1.745d18p-4f,  -0x1.555556p-4f,
                              0x1.3b13b2p-4f,  -0x1.24924ap-4f, 0x1.111112p-4f,  -0x1.000000p-4f,
                              0x1.e1e1e2p-5f,  -0x1.c71c72p-5f};
  float p = c[18];
  for (int n = 17; n >= 1; n--) p = std::fmaf(p, f, c[n]);
  p = p * f;
  return std::fmaf((float)e, 0x1.62e430p-1f, p);
}

// sin/cos of a in [0, pi/2): Taylor in a^2, Horner with fmaf.
static inline void det_sincos(float a, float* s, float* c) {
  float a2 = a * a;
  float ps = -0x1.ae7f3ep-41f;
  ps = std::fmaf(ps, a2, 0x1.612462p-33f);
  ps = std::fmaf(ps, a2, -0x1.ae6456p-26f);
  ps = std::fmaf(ps, a2, 0x1.71de3ap-19f);
  ps = std::fmaf(ps, a2, -0x1.a01a02p-13f);
  ps = std::fmaf(ps, a2, 0x1.111112p-7f);
  ps = std::fmaf(ps, a2, -0x1.555556p-3f);
  ps = std::fmaf(ps, a2, 1.0f);
  *s = ps * a;
  float pc = 0x1.ae7f3ep-45f;
  pc = std::fmaf(pc, a2, -0x1.93974ap-37f);
  pc = std::fmaf(pc, a2, 0x1.1eed8ep-29f);
  pc = std::fmaf(pc, a2, -0x1.27e4fcp-22f);
  pc = std::fmaf(pc, a2, 0x1.a01a02p-16f);
  pc = std::fmaf(pc, a2, -0x1.6c16c2p-10f);
  pc = std::fmaf(pc, a2, 0x1.555556p-5f);
  pc = std::fmaf(pc, a2, -0.5f);
  pc = std::fmaf(pc, a2, 1.0f);
  *c = pc;
}

// Four N(0,1) values for (seed, row, column block cb) = columns 4cb..4cb+3; `stream` = the counter's last word
// (0: EHX-GAUSS-1; 1 / 2: the basis and the latents of EHX-MANIFOLD-1).
static inline void normal4(uint64_t seed, uint64_t row, uint32_t cb, float z[4], uint32_t stream = 0u) {
  U4 ctr = {{(uint32_t)row, (uint32_t)(row >> 32), cb, stream}};
  U4 x = philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  for (int p = 0; p < 2; p++) {
    uint32_t xa = x.v[2 * p], xb = x.v[2 * p + 1];
    float u1 = ((float)(xa >> 9) + 0.5f) * 0x1p-23f;
    float r = sqrtf(-2.0f * det_log(u1));
    uint32_t t = xb >> 8;
    uint32_t quad = t >> 22;
    float a = ((float)(t & 0x3FFFFFu) * 0x1p-22f) * 0x1.921fb6p+0f;
    float s, c;
    det_sincos(a, &s, &c);
    float cs, sn;
    switch (quad) {
      case 0: cs = c; sn = s; break;
      case 1: cs = -s; sn = c; break;
      case 2: cs = -c; sn = -s; break;
      default: cs = s; sn = -c; break;
    }
    z[2 * p] = r * cs;
    z[2 * p + 1] = r * sn;
  }
}

// One row of `dim` values; optionally L2-normalised (sequential fp32 sum,
// x * (1/(sqrt(sum)+1e-30f)) — the hnswlib-python convention).
static inline void gen_row(uint64_t seed, uint64_t row, size_t dim, bool normalize, float* out) {
  for (size_t cb = 0; cb * 4 < dim; cb++) {
    float z[4];
    normal4(seed, row, (uint32_t)cb, z);
    for (int j = 0; j < 4 && cb * 4 + j < dim; j++) out[cb * 4 + j] = z[j];
  }
  if (normalize) {
    float norm = 0.0f;
    for (size_t i = 0; i < dim; i++) norm += out[i] * out[i];
    norm = 1.0f / (sqrtf(norm) + 1e-30f);
    for (size_t i = 0; i < dim; i++) out[i] = out[i] * norm;
  }
}

// EHX-MANIFOLD-1 (include/ehx_datagen.h): x[r][c] = sum_j l[r][j] * b[j][c] (sequential, non-fused) + 0.05 * e[r][c] with
// basis b (seed 20250213, stream 1, scaled by 1 / sqrt(R)), latents l (stream 2) and the EHX-GAUSS-1 noise e of the same
// seed; optionally L2-normalised like gen_row.  `basis` (R x dim, from manifold_basis) is shared by all rows.
static inline void manifold_basis(size_t dim, uint32_t R, float* basis) {
  const float rs = 1.0f / sqrtf((float)R);
  for (uint32_t j = 0; j < R; j++)
    for (size_t cb = 0; cb * 4 < dim; cb++) {
      float z[4];
      normal4(20250213ull, (uint64_t)j, (uint32_t)cb, z, 1u);
      for (int i = 0; i < 4 && cb * 4 + i < dim; i++) basis[(size_t)j * dim + cb * 4 + i] = z[i] * rs;
    }
}
static inline void gen_manifold_row(uint64_t seed, uint64_t row, size_t dim, uint32_t R, const float* basis, bool normalize,
                                    float* out) {
  float lat[64 + 4];
  for (uint32_t jb = 0; jb * 4 < R; jb++) normal4(seed, row, jb, lat + jb * 4, 2u);
  for (size_t cb = 0; cb * 4 < dim; cb++) {
    float e[4];
    normal4(seed, row, (uint32_t)cb, e, 0u);
    for (int i = 0; i < 4 && cb * 4 + i < dim; i++) {
      const size_t c = cb * 4 + i;
      float acc = 0.0f;
      for (uint32_t j = 0; j < R; j++) {
        const float prod = lat[j] * basis[(size_t)j * dim + c];
        acc = acc + prod;
      }
      const float ne = 0.05f * e[i];
      out[c] = acc + ne;
    }
  }
  if (normalize) {
    float norm = 0.0f;
    for (size_t i = 0; i < dim; i++) norm += out[i] * out[i];
    norm = 1.0f / (sqrtf(norm) + 1e-30f);
    for (size_t i = 0; i < dim; i++) out[i] = out[i] * norm;
  }
}

}  // namespace datagen
}  // namespace oracle
