// gkg_common.h — shared device helpers + error plumbing for libgkg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gkg_hip.h"

namespace gkg {

// Feature element load/store: fp32 as is, bf16 carried as raw uint16_t (widening is exact).
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const uint16_t* p) { return __uint_as_float(((uint32_t)*p) << 16); }
__device__ __forceinline__ float ldf(const _Float16* p) { return (float)*p; }            // fp16: widening is exact
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(_Float16* p, float v) { *p = (_Float16)v; }           // round-to-nearest-even
__device__ __forceinline__ void stf(uint16_t* p, float v) {
  // round-to-nearest-even; plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
  const __bf16 b = (__bf16)v;
  *p = __builtin_bit_cast(uint16_t, b);
}

// 4 consecutive features: one 16-byte (fp32) or 8-byte (bf16, round-to-nearest-even) store.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)lo) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)hi) << 16);
}
__device__ __forceinline__ void stf4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void stf4(uint16_t* p, const float4& v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}

// "XM layout" (include/gkg_hip.h): the grouped projection's [x | m] operand buffer, (T, 2C) fp32 with C = 4h.  Row t is
//   [x_0 | m_0 | x_1 | m_1 | x_2 | m_2 | x_3 | m_3],  x_q = x[t][q h .. (q+1) h),  m_q = m[t][q h .. (q+1) h):
// conv group q of BasicConv (reference torch_nn.py:61, groups = 4) reads the 2h contiguous floats at column 2 q h.  Channel ch
// of x sits at column xm_col(ch, h), of m at xm_col(ch, h) + h.  A token-major tensor is passed as (pointer, row pitch, chunk):
// chunk == 0 is the plain (T, C) matrix, chunk == h the x half of an XM buffer (pitch 2C).  4 consecutive channels never
// straddle a chunk (h % 4 == 0).
// (ch / chunk as a multiply-high: exact for ch < 2^32 / chunk, chunk >= 2; the reciprocal is loop-invariant)
__device__ __forceinline__ int xm_col(int ch, int chunk) {
  return chunk > 0 ? ch + (int)__umulhi((unsigned)ch, 0xffffffffu / (unsigned)chunk + 1u) * chunk : ch;
}

// GELU, erf form (the reference's nn.GELU(), torch_nn.py:24) and its derivative.  erf comes from Abramowitz & Stegun 7.1.26,
// |error| <= 1.5e-7 — at the level of fp32 rounding for O(1) activations and four orders below this path's 1e-3 contract —
// in ~15 vector instructions instead of erff's ~40 (the compile-time ablation of the fused inference kernel showed erff as
// a quarter of its time; bn_bwd_stats<GELU> ran 2.2x slower than its plain form).  exp(-x^2) with x = z / sqrt 2 is the
// Gaussian of the derivative as well: one exponential serves both.
__device__ __forceinline__ void gelu_parts(float z, float& cdf, float& gauss) {
  const float x = fabsf(z) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, x, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  gauss = __expf(-x * x);                              // exp(-z^2 / 2)
  cdf = 0.5f * (1.0f + copysignf(1.0f - p * t * gauss, z));
}
__device__ __forceinline__ float gelu_f(float z) {
  float cdf, gauss;
  gelu_parts(z, cdf, gauss);
  return z * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float z) {
  float cdf, gauss;
  gelu_parts(z, cdf, gauss);
  return cdf + z * 0.39894228040143267794f * gauss;
}

// Train-mode BN whose statistics came out of the projection kernel's epilogue as fp64 column sums (gkg_linear_bn_fwd* with
// train == 2): the CONSUMER of the projection — the BN-apply pass — derives scale / shift itself instead of a one-block
// finalize launch in between.  Every workgroup computes the coefficients of its channels once (into LDS), the first
// workgroup of each group also writes what the backward needs (a, c, mean, invstd) and updates the running statistics, and
// clears `zero_buf`: the OTHER of two alternating scratch buffers, i.e. what the previous projection accumulated into (see
// gkg_bn_bwd_atomic for the protocol).  Same arithmetic as bn_sums_finalize_kernel.
struct BnDerive {
  const double* sums;       // [nb][2][C]; null: the caller passes a / c
  const float* gamma; const float* beta; const float* bias;
  float* running_mean; float* running_var; long long* nbt;
  float* a_out; float* c_out; float* mean_out; float* invstd_out;
  int R; float momentum, eps;
  double* zero_buf; size_t zero_doubles;
};

__device__ __forceinline__ void bn_derive_channel(const BnDerive& d, size_t o, double S, double Q, bool side, float& av, float& cv) {
  const double m = S / d.R;
  double var = Q / d.R - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)d.eps));
  av = d.gamma[o] * is;
  cv = d.beta[o] - av * (float)m;
  if (side) {
    d.a_out[o] = av; d.c_out[o] = cv; d.mean_out[o] = (float)m; d.invstd_out[o] = is;
    if (d.running_mean) {
      const float bv = d.bias ? d.bias[o] : 0.f;
      d.running_mean[o] = (1.f - d.momentum) * d.running_mean[o] + d.momentum * ((float)m + bv);
      const double unb = d.R > 1 ? var * (double)d.R / (double)(d.R - 1) : var;
      d.running_var[o] = (1.f - d.momentum) * d.running_var[o] + d.momentum * (float)unb;
    }
  }
}

}  // namespace gkg

namespace gkg {
// gkg_gemm_x6.hip: BN scale / shift / saved statistics / running-stat update from the fp64 column sums (re-zeroes them)
hipError_t launch_bn_sums_finalize(double* stats, int R, int cout, int nb, const float* gamma, const float* beta,
                                   const float* bias, float* running_mean, float* running_var, float* bn_a, float* bn_c,
                                   float* bn_mean, float* bn_invstd, float momentum, float eps, long long* nbt, hipStream_t st,
                                   int nslots = 1);
}  // namespace gkg

// Records the message for gkg_last_error_string() and returns `code`.
int gkg_fail(int code, const char* msg);
int gkg_fail_hip(hipError_t e, const char* where);

// Adds algorithmic work to a profiled kernel's counter without timing anything (the k-NN entry point reports the
// contraction's 2 BG c N M flop once per call; the tile / prefilter launch scopes underneath do the timing).
void gkg_prof_add_work(int kernel_id, double work);

// Opt-in launch timing (gkg_prof_*): RAII bracket around one kernel launch on `st`.
struct GkgProfScope {
  GkgProfScope(int kernel_id, hipStream_t st, double work = 0.0);
  ~GkgProfScope();
  int slot;
  hipStream_t st;
};
