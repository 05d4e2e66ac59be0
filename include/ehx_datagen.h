/*
 * ehx_datagen.h — EHX-GAUSS-1: the deterministic synthetic workload of SURVEY.md §8d / BASELINE.md.
 *
 * Element (row r, column c) of dataset `seed` is an N(0,1) sample defined by
 *
 *   x[0..3] = Philox4x32-10( counter = (r_lo, r_hi, c/4, 0), key = (seed_lo, seed_hi) )
 *   pair p = (c%4)/2 uses (xa, xb) = (x[2p], x[2p+1]):
 *     u1   = (float(xa >> 9) + 0.5f) * 2^-23                       in (0,1), exact in fp32
 *     rad  = sqrt_rn( -2.0f * LOG(u1) )
 *     t    = xb >> 8 ; quad = t >> 22 ; a = (float(t & 0x3FFFFF) * 2^-22) * fl(pi/2)
 *     (s,c) = SINCOS(a), rotated by quad*90deg
 *     z[2p] = rad * cos , z[2p+1] = rad * sin                       (Box-Muller)
 *   LOG  : exponent/mantissa split (mantissa folded to [sqrt(2)/2, sqrt(2))), degree-18 Taylor
 *          of ln(1+f) evaluated by Horner with fused multiply-add, + e*fl(ln 2) by one fma
 *   SINCOS: Taylor polynomials in a^2 (sin to a^15, cos to a^16), Horner with fma
 *
 * Every step is a single IEEE-754 fp32 operation or an explicit fma, so host and device
 * produce bit-identical values (no libm).  Translation units that include this header MUST be
 * compiled with -ffp-contract=off (HIP's __fmul_rn/__fadd_rn are plain * and + and would be
 * contracted; __fsqrt_rn is the approximate native sqrt — neither is used here).  Cosine workloads L2-normalise each row with the
 * hnswlib-python convention: norm = 1/(sqrt(sum_i x_i^2) + 1e-30f) (sequential fp32 sum,
 * non-fused), x_i *= norm.  Corpus seed 20250211, query seed 20250212.
 *
 * EHX-MANIFOLD-1 (round 6): rows WITH STRUCTURE — points of an R-dimensional linear subspace of the `dims`-dimensional
 * space plus 5 % isotropic noise — on which a graph index reaches recall >= 0.95 at a small ef (isotropic Gaussian rows
 * at d = 768 are adversarial for any graph: SURVEY.md §7).  Same building blocks, so host and device agree bit for bit:
 *
 *   basis   b[j][c] = N4( key = EHX_SEED_BASIS, counter = (j, 0, c/4, 1) )[c%4] * rs,   rs = 1 / sqrt_rn(float(R))   j < R
 *   latent  l[r][j] = N4( key = seed,           counter = (r_lo, r_hi, j/4, 2) )[j%4]                                 j < R
 *   noise   e[r][c] = N4( key = seed,           counter = (r_lo, r_hi, c/4, 0) )[c%4]          (the EHX-GAUSS-1 element)
 *   x[r][c] = ( ... ((0 + l0*b0c) + l1*b1c) ... + l(R-1)*b(R-1)c ) + 0.05f * e[r][c]     (sequential, non-fused, fp32)
 *
 * (N4 = the Philox4x32-10 -> Box-Muller block above with the counter's last word as the STREAM: 0 = EHX-GAUSS-1.)
 * Cosine workloads normalise rows as above.  Queries: the same generator with the query seed — drawn independently of
 * the rows (erratum of round 3).
 *
 * This header is the product's implementation (device + host inline); oracle/datagen_oracle.hpp
 * is an independent restatement of the same spec used only by the tests.
 */
#ifndef EHX_DATAGEN_H_
#define EHX_DATAGEN_H_

#include <stdint.h>

#define EHX_SEED_CORPUS 20250211ull
#define EHX_SEED_QUERY 20250212ull
#define EHX_SEED_BASIS 20250213ull  /* EHX-MANIFOLD-1: the subspace is the same for every dataset seed */
#define EHX_MANIFOLD_MAX_LATENT 64u

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define EHX_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#include <string.h>
#define EHX_HD static inline
#endif

namespace ehx_datagen {

struct u32x4 {
  uint32_t x, y, z, w;
};

EHX_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

EHX_HD u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = mulhi32(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = mulhi32(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    u32x4 n;
    n.x = hi1 ^ c.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ c.w ^ k1;
    n.w = lo0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

EHX_HD float f_fma(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fmaf(a, b, c);
#else
  return fmaf(a, b, c);
#endif
}
EHX_HD float f_mul(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return a * b; /* never contracted: built with -ffp-contract=off */
#else
  return a * b;
#endif
}
EHX_HD float f_add(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return a + b;
#else
  return a + b;
#endif
}
EHX_HD float f_sqrt(float a) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_sqrtf(a); /* correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt) */
#else
  return sqrtf(a);
#endif
}
EHX_HD float f_div(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return a / b; /* correctly rounded, same flag */
#else
  return a / b;
#endif
}
EHX_HD float u32_as_f32(uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(b);
#else
  float f;
  memcpy(&f, &b, 4);
  return f;
#endif
}
EHX_HD uint32_t f32_as_u32(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(f);
#else
  uint32_t b;
  memcpy(&b, &f, 4);
  return b;
#endif
}

EHX_HD float det_log(float u) {
  const uint32_t b = f32_as_u32(u);
  int e = (int)(b >> 23) - 127;
  float m = u32_as_f32((b & 0x007FFFFFu) | 0x3F800000u);
  if (m > 0x1.6a09e6p+0f) {
    m = f_mul(m, 0.5f);
    e += 1;
  }
  const float f = f_add(m, -1.0f);
  float p = -0x1.c71c72p-5f;           /* -1/18 */
  p = f_fma(p, f, 0x1.e1e1e2p-5f);     /* +1/17 */
  p = f_fma(p, f, -0x1.000000p-4f);
  p = f_fma(p, f, 0x1.111112p-4f);
  p = f_fma(p, f, -0x1.24924ap-4f);
  p = f_fma(p, f, 0x1.3b13b2p-4f);
  p = f_fma(p, f, -0x1.555556p-4f);
  p = f_fma(p, f, 0x1.745d18p-4f);
  p = f_fma(p, f, -0x1.99999ap-4f);
  p = f_fma(p, f, 0x1.c71c72p-4f);
  p = f_fma(p, f, -0x1.000000p-3f);
  p = f_fma(p, f, 0x1.24924ap-3f);
  p = f_fma(p, f, -0x1.555556p-3f);
  p = f_fma(p, f, 0x1.99999ap-3f);
  p = f_fma(p, f, -0x1.000000p-2f);
  p = f_fma(p, f, 0x1.555556p-2f);
  p = f_fma(p, f, -0x1.000000p-1f);
  p = f_fma(p, f, 0x1.000000p+0f);
  p = f_mul(p, f);
  return f_fma((float)e, 0x1.62e430p-1f, p);
}

EHX_HD void det_sincos(float a, float* s, float* c) {
  const float a2 = f_mul(a, a);
  float ps = -0x1.ae7f3ep-41f;
  ps = f_fma(ps, a2, 0x1.612462p-33f);
  ps = f_fma(ps, a2, -0x1.ae6456p-26f);
  ps = f_fma(ps, a2, 0x1.71de3ap-19f);
  ps = f_fma(ps, a2, -0x1.a01a02p-13f);
  ps = f_fma(ps, a2, 0x1.111112p-7f);
  ps = f_fma(ps, a2, -0x1.555556p-3f);
  ps = f_fma(ps, a2, 1.0f);
  *s = f_mul(ps, a);
  float pc = 0x1.ae7f3ep-45f;
  pc = f_fma(pc, a2, -0x1.93974ap-37f);
  pc = f_fma(pc, a2, 0x1.1eed8ep-29f);
  pc = f_fma(pc, a2, -0x1.27e4fcp-22f);
  pc = f_fma(pc, a2, 0x1.a01a02p-16f);
  pc = f_fma(pc, a2, -0x1.6c16c2p-10f);
  pc = f_fma(pc, a2, 0x1.555556p-5f);
  pc = f_fma(pc, a2, -0.5f);
  pc = f_fma(pc, a2, 1.0f);
  *c = pc;
}

/* z[0..3] = columns 4*cb .. 4*cb+3 of row `row` of stream `stream` (0 = EHX-GAUSS-1). */
EHX_HD void normal4_stream(uint64_t seed, uint64_t row, uint32_t cb, uint32_t stream, float z[4]) {
  u32x4 ctr;
  ctr.x = (uint32_t)row;
  ctr.y = (uint32_t)(row >> 32);
  ctr.z = cb;
  ctr.w = stream;
  const u32x4 r = philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t xs[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const uint32_t xa = xs[2 * p], xb = xs[2 * p + 1];
    const float u1 = f_mul(f_add((float)(xa >> 9), 0.5f), 0x1p-23f);
    const float rad = f_sqrt(f_mul(-2.0f, det_log(u1)));
    const uint32_t t = xb >> 8;
    const uint32_t quad = t >> 22;
    const float a = f_mul(f_mul((float)(t & 0x3FFFFFu), 0x1p-22f), 0x1.921fb6p+0f);
    float s, c;
    det_sincos(a, &s, &c);
    const float cs = quad == 0 ? c : (quad == 1 ? -s : (quad == 2 ? -c : s));
    const float sn = quad == 0 ? s : (quad == 1 ? c : (quad == 2 ? -s : -c));
    z[2 * p] = f_mul(rad, cs);
    z[2 * p + 1] = f_mul(rad, sn);
  }
}
EHX_HD void normal4(uint64_t seed, uint64_t row, uint32_t cb, float z[4]) { normal4_stream(seed, row, cb, 0u, z); }

/* EHX-MANIFOLD-1 pieces: b[j][4 cb .. 4 cb + 3] of the R-dimensional basis, and l[row][4 jb .. 4 jb + 3] */
EHX_HD void manifold_basis4(uint32_t j, uint32_t cb, uint32_t R, float b[4]) {
  normal4_stream(EHX_SEED_BASIS, (uint64_t)j, cb, 1u, b);
  const float rs = f_div(1.0f, f_sqrt((float)R));
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = f_mul(b[i], rs);
}
EHX_HD void manifold_latent4(uint64_t seed, uint64_t row, uint32_t jb, float l[4]) { normal4_stream(seed, row, jb, 2u, l); }
/* x[row][4 cb + i] from the row's latents l[0..R), the basis column values bcol[j] = b[j][4 cb + i] and the noise e */
EHX_HD float manifold_element(const float* l, const float* bcol, uint32_t R, float e) {
  float acc = 0.0f;
  for (uint32_t j = 0; j < R; ++j) acc = f_add(acc, f_mul(l[j], bcol[j]));
  return f_add(acc, f_mul(0.05f, e));
}

}  // namespace ehx_datagen
#endif /* EHX_DATAGEN_H_ */
