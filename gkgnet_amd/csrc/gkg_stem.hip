// gkg_stem.hip — the backbone's FIRST stem convolution: Conv2d(3 -> C1/2, 3x3, stride 2, padding 1) on the input image
// (reference gkgnet.py:79-81), optionally with the eval-mode BN + GELU behind it folded into the epilogue.
//
// With 3 input channels the library's implicit-GEMM kernels run at a few TFLOP/s (MIOpen igemm_fwd_gtcx35_nhwc_*: 756 us for
// B = 32 at 576 x 576 under bf16 autocast, 1 015 us in fp32 — 12 % of the cfg3 forward).  The layer has 27 taps per output
// value: a direct form is bound by its vector FMAs (5.7 GFLOP at cfg3) and its 340-550 MB of image + output traffic.  One
// thread = PX consecutive output pixels x all output channels: the 27 x COUT weights sit transposed in LDS ([tap][co],
// broadcast reads of 4 at a time, each serving PX pixels), the input columns of a (channel, tap row) are loaded once per
// thread, the accumulators are packed pairs (v_pk_fma_f32), the result rows — COUT contiguous values of the channels-last
// output, which is what the next (library) convolution and the blocks' token-major kernels want — go out as 16-byte stores.
#include "gkg_common.h"

#ifndef STEM_ABL
#define STEM_ABL 0     // measurement builds only (tools/ubench/stem_ablate.py): 1 no loads, 2 no stores, 4 one FMA group per tap
#endif

namespace gkg {

typedef float st_f2 __attribute__((ext_vector_type(2)));

// PX consecutive output pixels of one row per thread: every weight read from LDS serves PX pixels (one pixel per thread was
// bound by the LDS — 270 broadcast ds_read_b128 per wave and pixel column, 167 us of LDS time per CU at cfg3 — not by its
// 540 packed FMAs), and the 2 PX + 1 input columns of a (channel, row) are loaded once for all of them.
template <int COUT, typename OutT, int PX>
__global__ __launch_bounds__(256, 2) void stem_conv3x3s2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, const float* __restrict__ a,
                                                                const float* __restrict__ cs, OutT* __restrict__ out, int B,
                                                                int cin, int H, int W, int Ho, int Wo, int act) {
  __shared__ __align__(16) float wl[4 * 9 * COUT];          // [tap = c * 9 + dy * 3 + dx][co]
  const int ntap = cin * 9;
  for (int i = threadIdx.x; i < ntap * COUT; i += 256) {
    const int tap = i / COUT, co = i - tap * COUT;
    wl[i] = w[(size_t)co * ntap + tap];
  }
  __syncthreads();
  const int Wg = (Wo + PX - 1) / PX;                          // pixel groups per output row
  const long long gidx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long ngrp = (long long)B * Ho * Wg;
  if (gidx >= ngrp) return;
  const int wg = (int)(gidx % Wg);
  const int ho = (int)((gidx / Wg) % Ho);
  const int b = (int)(gidx / ((long long)Wg * Ho));
  const int wo0 = wg * PX;
  st_f2 acc[PX][COUT / 2];
#pragma unroll
  for (int px = 0; px < PX; ++px)
#pragma unroll
    for (int k = 0; k < COUT / 2; ++k) acc[px][k] = st_f2{0.f, 0.f};
  const float* xb = x + (size_t)b * cin * H * W;
  const int hi0 = 2 * ho - 1, wi0 = 2 * wo0 - 1;
  int c = 0, dy = 0;
  constexpr int NC = 2 * PX + 1;
#pragma unroll 1
  for (int cr = 0; cr < cin * 3; ++cr) {                      // one (channel, tap row) per iteration (NOT unrolled: the compiler
    const int hi = hi0 + dy;                                  // otherwise hoists the weights into registers and spills).
    float col[NC];                                            // Requesting the columns of iteration cr + 1 before the FMAs of
    const bool rowok = hi >= 0 && hi < H;                     // cr measured the same (EXPERIMENTS.md) and costs 9 registers.
    const float* xr = xb + ((size_t)c * H + (rowok ? hi : 0)) * W;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int wi = wi0 + j;
#if STEM_ABL & 1
      col[j] = (float)(wi + cr);
#else
      col[j] = (rowok && wi >= 0 && wi < W) ? xr[wi] : 0.f;
#endif
    }
    if (++dy == 3) { dy = 0; ++c; }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const float4* wt = reinterpret_cast<const float4*>(wl + (cr * 3 + dx) * COUT);
#pragma unroll
      for (int k = 0; k < ((STEM_ABL & 4) ? 1 : COUT / 4); ++k) {
        const float4 w4 = wt[k];
        const st_f2 w01 = st_f2{w4.x, w4.y}, w23 = st_f2{w4.z, w4.w};
#pragma unroll
        for (int px = 0; px < PX; ++px) {
          const float v = col[2 * px + dx];
          const st_f2 vv = st_f2{v, v};
          acc[px][2 * k] = __builtin_elementwise_fma(vv, w01, acc[px][2 * k]);
          acc[px][2 * k + 1] = __builtin_elementwise_fma(vv, w23, acc[px][2 * k + 1]);
        }
      }
    }
  }
  const long long prow = ((long long)b * Ho + ho) * Wo;
#pragma unroll
  for (int px = 0; px < PX; ++px) {
    if (wo0 + px < Wo) {
      OutT* o = out + (size_t)(prow + wo0 + px) * COUT;
#pragma unroll
      for (int k = 0; k < COUT / 4; ++k) {
        float4 r = make_float4(acc[px][2 * k][0], acc[px][2 * k][1], acc[px][2 * k + 1][0], acc[px][2 * k + 1][1]);
        if (bias) { r.x += bias[4 * k]; r.y += bias[4 * k + 1]; r.z += bias[4 * k + 2]; r.w += bias[4 * k + 3]; }
        if (a) {
          r.x = __builtin_fmaf(a[4 * k], r.x, cs[4 * k]); r.y = __builtin_fmaf(a[4 * k + 1], r.y, cs[4 * k + 1]);
          r.z = __builtin_fmaf(a[4 * k + 2], r.z, cs[4 * k + 2]); r.w = __builtin_fmaf(a[4 * k + 3], r.w, cs[4 * k + 3]);
        }
        if (act == 1) { r.x = gelu_f(r.x); r.y = gelu_f(r.y); r.z = gelu_f(r.z); r.w = gelu_f(r.w); }
#if STEM_ABL & 2
        if (r.x == 1.2345e-30f) stf4(o + 4 * k, r);
#else
        stf4(o + 4 * k, r);
#endif
      }
    }
  }
}

template <int COUT>
static hipError_t stem_launch(const float* x, const float* w, const float* bias, const float* a, const float* c, void* out, int B,
                              int cin, int H, int W, int Ho, int Wo, int act, int out_dtype, hipStream_t st) {
  constexpr int PX = COUT <= 48 ? 4 : 2;                      // accumulators: PX * COUT registers
  const long long ngrp = (long long)B * Ho * ((Wo + PX - 1) / PX);
  const dim3 grid((unsigned)((ngrp + 255) / 256));
  if (out_dtype == GKG_BF16)
    hipLaunchKernelGGL((stem_conv3x3s2_kernel<COUT, uint16_t, PX>), grid, dim3(256), 0, st, x, w, bias, a, c, (uint16_t*)out, B, cin, H, W, Ho, Wo, act);
  else
    hipLaunchKernelGGL((stem_conv3x3s2_kernel<COUT, float, PX>), grid, dim3(256), 0, st, x, w, bias, a, c, (float*)out, B, cin, H, W, Ho, Wo, act);
  return hipGetLastError();
}

}  // namespace gkg
using namespace gkg;

// 1 when gkg_stem_conv3x3s2_fwd has a form for (cin, cout): 1..4 input channels, 24 / 40 / 48 / 64 output channels (the
// widths of the t / s / m / b stems).
extern "C" int gkg_stem_conv3x3s2_supported(int cin, int cout) {
  return cin >= 1 && cin <= 4 && (cout == 24 || cout == 40 || cout == 48 || cout == 64) ? 1 : 0;
}

// out (B, Ho, Wo, cout) channels-last = act(a * (Conv2d(cin -> cout, 3x3, stride 2, padding 1)(x) + bias) + c):
//   x (B, cin, H, W) fp32 NCHW, w (cout, cin, 3, 3) fp32, bias (cout) or NULL; a / c (cout) fp32 or both NULL (no affine: the
//   plain convolution, for training); act 0 none / 1 GELU (erf); out_dtype GKG_F32 / GKG_BF16; Ho = (H + 1) / 2, Wo = (W + 1) / 2.
extern "C" int gkg_stem_conv3x3s2_fwd(const float* x, const float* w, const float* bias, const float* a, const float* c, void* out,
                                      int B, int cin, int H, int W, int cout, int act, int out_dtype, void* stream) {
  if (!x || !w || !out || ((a == nullptr) != (c == nullptr))) return gkg_fail(GKG_ERR_NULL, "gkg_stem_conv3x3s2_fwd: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || (act != 0 && act != 1) || (out_dtype != GKG_F32 && out_dtype != GKG_BF16))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_stem_conv3x3s2_fwd: bad sizes");
  if (!gkg_stem_conv3x3s2_supported(cin, cout))
    return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_stem_conv3x3s2_fwd: need cin <= 4 and cout in {24, 40, 48, 64}");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  if ((long long)B * Ho * Wo > 0x7fffffffLL * 128) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_stem_conv3x3s2_fwd: too many pixels");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  switch (cout) {
    case 24: e = stem_launch<24>(x, w, bias, a, c, out, B, cin, H, W, Ho, Wo, act, out_dtype, st); break;
    case 40: e = stem_launch<40>(x, w, bias, a, c, out, B, cin, H, W, Ho, Wo, act, out_dtype, st); break;
    case 48: e = stem_launch<48>(x, w, bias, a, c, out, B, cin, H, W, Ho, Wo, act, out_dtype, st); break;
    default: e = stem_launch<64>(x, w, bias, a, c, out, B, cin, H, W, Ho, Wo, act, out_dtype, st); break;
  }
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "stem_conv3x3s2_kernel");
}
