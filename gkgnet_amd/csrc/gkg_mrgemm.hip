// gkg_mrgemm.hip — SURVEY §8 row g1 (inference): neighbour gather + max(x_j - x_i) as the A-OPERAND PRODUCER of the grouped
// 1x1 projection, BN (eval) + GELU in its epilogue.
//
// Reference chain (mmcls/models/backbones/vig_model/torch_vertex.py:47-62 + torch_nn.py:57-69):
//     x_i, x_j = batched_index_select(...)        (B*G, c, N, k) each
//     m        = max_k(x_j - x_i)
//     u        = interleave [x_0, m_0, x_1, m_1, ...]            (B, 2C, N, 1)
//     out      = GELU(BN(Conv2d(2C, 2C, 1, groups=4)(u)))
// In inference nothing downstream needs `u` (or `m`) in memory.  One workgroup (8 waves) owns 64 tokens x one conv group q
// (wide layers) or x all 4 conv groups (narrow layers, where the whole image of the tile fits 64 KB of LDS):
//   phase 0  the tile's index rows go to LDS once (clamped int32): every channel quad of a token shares them;
//   phase 1  every thread gathers the k neighbour rows of (token, 4 channels) from L2 (token-major rows: one float4 per
//            neighbour; two items in flight per thread), takes the max of the differences — the same arithmetic, in the
//            same order, as mr_fwd_tm_kernel — and writes the interleaved [x, m] values as bf16 (round-to-nearest-even)
//            straight into the LDS image of the GEMM's A tile: 64 rows x ci = C/2 input channels per conv group;
//   phase 2  v_mfma_f32_32x32x16_bf16 over that tile: wave w takes row block (w & 1) and up to 3 (conv group, column block)
//            pairs; the weights stream from L2 as pre-arranged 16-byte fragments [q][ci/8][co][8], fetched 4 contraction
//            steps ahead into registers, fp32 accumulation;
//   phase 3  out = act(a * acc + c) (eval-mode BN folded with the conv bias: the same (a, c) the separate affine_act pass
//            used), rounded to bf16, staged through the (now free) LDS tile and stored as 16-byte row segments into the
//            token-major (T, 2C) operand of fc2.
// Replaces three launches (mr_fwd_tm -> batched GEMM -> affine_act) and the (T, 2C) bf16 + (T, 2C) fp32 round trips
// between them.  bf16 inference only (callers under bf16 autocast with gradients off: gkgnet_amd/fused.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gkg_common.h"

namespace gkg {

typedef float mg_f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 mg_bf16x8;

// Compile-time phase ablation (measurement only, tools/ubench/mrgemm_ablate.py builds private copies with -DMG_ABL=<bits>):
// 1 no neighbour gathers (the centre row stands in), 2 no MFMA loop, 4 no GELU, 8 no index staging (indices = token itself),
// 16 no output stores.  Run-time flags for the same purpose pushed the per-item arrays into scratch memory (4x slower).
#ifndef MG_ABL
#define MG_ABL 0
#endif

constexpr int MG_ROWS = 64;       // tokens per workgroup
constexpr int MG_NW = 8;          // waves per workgroup (512 threads)
constexpr int MG_MAXB = 3;        // 32 x 32 output blocks per wave (wide layers; narrow ones: 2)
constexpr int MG_D = 3;           // weight fragments are fetched this many contraction steps ahead (registers)

struct MrGemmArgs {
  const float* x;          // (B, N, C) token-major fp32
  const float* src;        // (B, M, C) or null (self graph: src = x, M = N)
  const int64_t* nn_idx;   // (B*G, N, k)
  const uint16_t* nn16;    // (B*G, N, k) compact lists (gkg_knn_fwd_tm16) instead of nn_idx, or null
  const uint4* wp;         // [4][ci_pad/8][co_pad] fragments of 8 consecutive input channels (bf16)
  const float* a;          // (4*co) BN scale
  const float* cs;         // (4*co) BN shift (conv bias folded)
  uint16_t* out;           // (T, ldo) bf16; conv group q writes columns [q*co, (q+1)*co)
  int ldo;
  int B, G, c, N, M, k, C, Cq, ci, co, ci_pad, co_pad, act;
  long long T;
  int tiles, tiles_per_xcd;
};

__device__ __forceinline__ int mg_clamp(int64_t v, int M) { return (int)(v < 0 ? 0 : (v >= M ? M - 1 : v)); }
// torch.max semantics (same as gkg_mr.hip::takes): NaN propagates, first maximum wins
__device__ __forceinline__ bool mg_takes(float v, float best) { return v > best || (v != v && best == best); }
// GELU (erf form) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): ~15 vector instructions instead of the
// ~40 of erff.  The result is rounded to bf16 (2^-9 relative), so against the exact erf it differs only where a rounding
// boundary falls within 1e-7: the compile-time ablation showed erff as 21-26 % of this kernel at every stage shape.
__device__ __forceinline__ float mg_gelu(float z) {
  const float x = fabsf(z) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, x, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float e = 1.0f - p * t * __expf(-x * x);          // erf(|z| / sqrt 2)
  return 0.5f * z * (1.0f + copysignf(e, z));
}

// QG: conv groups per workgroup.  4 (narrow layers): one workgroup owns 64 tokens x ALL channels — index rows, token rows
// and output rows are each touched once and fully coalesced; 1 (wide layers): one workgroup per (64 tokens, conv group),
// the 4 groups of a tile adjacent in dispatch order on one XCD.
// MAXB: output blocks per wave (registers); INF: (token, quad) items in flight per thread in phase 1; WPE: waves per SIMD the
// register allocation must allow (8 for the narrow layers: their phases are pure latency, occupancy is what hides it).
template <int KS, int QG, int MAXB = MG_MAXB, int INF = 2, int WPE = 4, int DEPTH = MG_D>
__global__ __launch_bounds__(64 * MG_NW, WPE) void mr_linear_bf16_kernel(MrGemmArgs g) {
  extern __shared__ __align__(16) unsigned char mg_lds[];
  constexpr int NT = 64 * MG_NW;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware map: workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  Every XCD gets a CONTIGUOUS
  // range of token tiles (whole images: a token's neighbours are rows of its own image); with QG == 1 the 4 conv groups
  // of a tile — same index rows, same token rows, different channel quarter — are adjacent in its dispatch order.
  const int lin = blockIdx.x;
  const int xcd = lin & 7, j = lin >> 3;
  const int tl = QG == 4 ? j : (j >> 2), q0 = QG == 4 ? 0 : (j & 3);
  const int tile = xcd * g.tiles_per_xcd + tl;
  if (tl >= g.tiles_per_xcd || tile >= g.tiles) return;
  const long long t0 = (long long)tile * MG_ROWS;
  const int pitch = (g.ci_pad + 8) * 2;                 // bytes; 16 B of padding: conflict-free 16-byte fragment reads
  const int qstride = MG_ROWS * pitch;                  // bytes between the A tiles of two conv groups (QG == 4)
  const int C = g.C, N = g.N, M = g.M;
  const int k = KS > 0 ? KS : g.k;

  // ---------------------------------------------------------------- phase 0: the tile's index rows -> LDS (int32, clamped)
  // k-NN groups touched by this workgroup's channels: [glo, glo + NG)
  const int ch_lo = q0 * g.Cq, ch_hi = (q0 + QG) * g.Cq - 1;
  const int glo = ch_lo / g.c, NG = ch_hi / g.c - glo + 1;
  int* ids = reinterpret_cast<int*>(mg_lds + QG * qstride);          // [64][NG][k]
  {
    const int per_tok = NG * k;
    for (int e = tid; e < MG_ROWS * per_tok; e += NT) {
      const int tok = e / per_tok, r = e - tok * per_tok;
      const int gi = r / k, jj = r - gi * k;
      const long long t = t0 + tok;
      int v = 0;
      if (t < g.T) {
        const int b = (int)(t / N), n = (int)(t - (long long)b * N);
        const size_t ie = (((size_t)b * g.G + glo + gi) * N + n) * k + jj;
        v = (MG_ABL & 8) ? min(n, M - 1) : (g.nn16 ? min((int)g.nn16[ie], M - 1) : mg_clamp(g.nn_idx[ie], M));
      }
      ids[e] = v;
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- phase 1: the A tile = interleaved [x, max-relative]
  const int Q4 = g.Cq >> 2;                              // channel quads per conv group
  const int QW = QG * Q4;                                // quads per token in this workgroup
  const float* srcb = g.src ? g.src : g.x;
  const int nitems = MG_ROWS * QW;
  struct Item { float4 xi; float4 nb[KS > 0 ? KS : 1]; int tok, qq; bool live; };
  auto fetch = [&](int it, Item& I) __attribute__((always_inline)) {
    I.live = false;
    I.tok = 0; I.qq = 0;
    if (it >= nitems) return;
    const int tok = it / QW, qq = it - tok * QW;         // qq = qi * Q4 + quad: consecutive threads, consecutive channels
    I.tok = tok; I.qq = qq;
    const long long t = t0 + tok;
    if (t >= g.T) return;
    I.live = true;
    const int b = (int)(t / N);
    const int ch = ch_lo + 4 * qq;                        // original channel (4 channels never straddle a k-NN group)
    const int* ip = ids + (tok * NG + (ch / g.c - glo)) * k;
    I.xi = *reinterpret_cast<const float4*>(g.x + (size_t)t * C + ch);
    const float* sb = srcb + (size_t)b * M * C + ch;
    if (KS > 0) {
#pragma unroll
      for (int u = 0; u < KS; ++u) I.nb[u] = (MG_ABL & 1) ? I.xi : *reinterpret_cast<const float4*>(sb + (size_t)ip[u] * C);
    }
  };
  auto finish = [&](const Item& I, int it) __attribute__((always_inline)) {
    if (it >= nitems) return;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (I.live) {
      const float4 xi = I.xi;
      float4 best;
      if (KS > 0) {
        best = make_float4(I.nb[0].x - xi.x, I.nb[0].y - xi.y, I.nb[0].z - xi.z, I.nb[0].w - xi.w);
#pragma unroll
        for (int u = 1; u < KS; ++u) {
          const float d0 = I.nb[u].x - xi.x, d1 = I.nb[u].y - xi.y, d2 = I.nb[u].z - xi.z, d3 = I.nb[u].w - xi.w;
          if (mg_takes(d0, best.x)) best.x = d0;
          if (mg_takes(d1, best.y)) best.y = d1;
          if (mg_takes(d2, best.z)) best.z = d2;
          if (mg_takes(d3, best.w)) best.w = d3;
        }
      } else {
        const long long t = t0 + I.tok;
        const int b = (int)(t / N);
        const int ch = ch_lo + 4 * I.qq;
        const int* ip = ids + (I.tok * NG + (ch / g.c - glo)) * k;
        const float* sb = srcb + (size_t)b * M * C + ch;
        const float4 n0 = *reinterpret_cast<const float4*>(sb + (size_t)ip[0] * C);
        best = make_float4(n0.x - xi.x, n0.y - xi.y, n0.z - xi.z, n0.w - xi.w);
        for (int u = 1; u < k; ++u) {
          const float4 nv = *reinterpret_cast<const float4*>(sb + (size_t)ip[u] * C);
          const float d0 = nv.x - xi.x, d1 = nv.y - xi.y, d2 = nv.z - xi.z, d3 = nv.w - xi.w;
          if (mg_takes(d0, best.x)) best.x = d0;
          if (mg_takes(d1, best.y)) best.y = d1;
          if (mg_takes(d2, best.z)) best.z = d2;
          if (mg_takes(d3, best.w)) best.w = d3;
        }
      }
      v = make_uint4(pack_bf16x2(xi.x, best.x), pack_bf16x2(xi.y, best.y), pack_bf16x2(xi.z, best.z),
                     pack_bf16x2(xi.w, best.w));
    }
    const int qi = I.qq / Q4, quad = I.qq - qi * Q4;
    *reinterpret_cast<uint4*>(mg_lds + qi * qstride + I.tok * pitch + 16 * quad) = v;
  };
  if (KS > 0 && KS <= 12 && INF == 2) {
    for (int it = tid; it < nitems; it += 2 * NT) {      // two items in flight per thread: 2 (k + 1) row loads outstanding
      Item I0, I1;
      fetch(it, I0);
      fetch(it + NT, I1);
      finish(I0, it);
      finish(I1, it + NT);
    }
  } else {
    for (int it = tid; it < nitems; it += NT) {          // long index rows: one item's k + 1 loads fill the register budget
      Item I0;
      fetch(it, I0);
      finish(I0, it);
    }
  }
  if (g.ci_pad > g.ci && tid < MG_ROWS * QG)             // contraction padding (ci % 16 == 8): one zero fragment per row
    *reinterpret_cast<uint4*>(mg_lds + (tid / MG_ROWS) * qstride + (tid % MG_ROWS) * pitch + 2 * g.ci) = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // ---------------------------------------------------------------- phase 2: (64 x ci) @ W_q^T on the bf16 matrix cores
  // wave w: row block rb = w & 1 and the (conv group, column block) pairs p = (w >> 1) + (NW / 2) u, u < MAXB
  const int l31 = lane & 31, kg = lane >> 5;
  const int rb = w & 1;
  const int ncb = g.co_pad >> 5;
  const int npairs = QG * ncb;
  const int S = g.ci_pad >> 4;
  const size_t wq_stride = (size_t)(g.ci_pad >> 3) * g.co_pad;       // fragments per conv group
  const uint4* wptr[MAXB];
  const unsigned char* aptr[MAXB];
  bool has[MAXB];
#pragma unroll
  for (int u = 0; u < MAXB; ++u) {
    const int p = (w >> 1) + (MG_NW / 2) * u;
    has[u] = p < npairs;                                             // wave-uniform
    const int qi = has[u] ? p / ncb : 0, cb = has[u] ? p - qi * ncb : 0;
    wptr[u] = g.wp + (size_t)(q0 + qi) * wq_stride + (size_t)kg * g.co_pad + 32 * cb + l31;
    aptr[u] = mg_lds + qi * qstride + (32 * rb + l31) * pitch + 16 * kg;
  }
  mg_f32x16 acc[MAXB];
#pragma unroll
  for (int u = 0; u < MAXB; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  uint4 bq[DEPTH][MAXB];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int u = 0; u < MAXB; ++u)
      bq[d][u] = (has[u] && d < S) ? wptr[u][(size_t)(2 * d) * g.co_pad] : make_uint4(0, 0, 0, 0);
  for (int s0 = 0; s0 < ((MG_ABL & 2) ? 0 : S); s0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int s = s0 + d;
      if (s < S) {                                                   // uniform
        uint4 bc[MAXB];
#pragma unroll
        for (int u = 0; u < MAXB; ++u) bc[u] = bq[d][u];
#pragma unroll
        for (int u = 0; u < MAXB; ++u)
          if (has[u] && s + DEPTH < S) bq[d][u] = wptr[u][(size_t)(2 * (s + DEPTH)) * g.co_pad];
#pragma unroll
        for (int u = 0; u < MAXB; ++u)
          if (has[u]) {
            const mg_bf16x8 av = __builtin_bit_cast(mg_bf16x8, *reinterpret_cast<const uint4*>(aptr[u] + 32 * s));
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(mg_bf16x8, bc[u]), acc[u], 0, 0, 0);
          }
      }
    }
  }

  // ---------------------------------------------------------------- phase 3: BN (eval) + activation -> bf16 -> rows out
  __syncthreads();                                       // every wave is done reading the A tiles: reuse them as the stage
#pragma unroll
  for (int u = 0; u < MAXB; ++u) {
    const int p = (w >> 1) + (MG_NW / 2) * u;
    const int qi = has[u] ? p / ncb : 0, cb = has[u] ? p - qi * ncb : 0;
    const int col = 32 * cb + l31;
    if (has[u] && col < g.co) {
      const float av = g.a[(q0 + qi) * g.co + col], cv = g.cs[(q0 + qi) * g.co + col];
      unsigned char* st = mg_lds + qi * qstride + 2 * col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * kg;
        float o = __builtin_fmaf(av, acc[u][r], cv);
        if (g.act == 1 && !(MG_ABL & 4)) o = mg_gelu(o);
        *reinterpret_cast<uint16_t*>(st + row * pitch) = __builtin_bit_cast(uint16_t, (__bf16)o);
      }
    }
  }
  __syncthreads();
  const int CH = g.co >> 3;                              // 16-byte chunks per (token, conv group) output segment
  const int CW = QG * CH;
  for (int it = tid; it < MG_ROWS * CW; it += NT) {
    const int tok = it / CW, r = it - tok * CW;
    const int qi = r / CH, ck = r - qi * CH;
    const long long t = t0 + tok;
    if (t < g.T && !(MG_ABL & 16))
      *reinterpret_cast<uint4*>(g.out + (size_t)t * g.ldo + (q0 + qi) * g.co + 8 * ck) =
          *reinterpret_cast<const uint4*>(mg_lds + qi * qstride + tok * pitch + 16 * ck);
  }
}

static size_t mg_lds_bytes(const MrGemmArgs& g, int QG) {
  // k-NN groups touched by one workgroup's channels, exactly as the kernel computes it: the maximum over the conv groups q0
  // of (last channel) / c - (first channel) / c + 1.  (c >= Cq does NOT imply one group: with G = 3, Cq = C / 4 < c = C / 3
  // and conv group 1 straddles k-NN groups 0 and 1 — ADVICE r3.)
  int NG = g.G;
  if (QG != 4) {
    NG = 1;
    for (int q0 = 0; q0 < 4; ++q0) {
      const int n = ((q0 + 1) * g.Cq - 1) / g.c - (q0 * g.Cq) / g.c + 1;
      if (n > NG) NG = n;
    }
  }
  return (size_t)QG * MG_ROWS * (g.ci_pad + 8) * 2 + (size_t)MG_ROWS * NG * g.k * 4;
}

template <int KS, int QG, int MAXB = MG_MAXB, int INF = 2, int WPE = 4, int DEPTH = MG_D>
static hipError_t mg_launch(const MrGemmArgs& g, hipStream_t st) {
  const size_t lds = mg_lds_bytes(g, QG);
  if (lds > 64 * 1024) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&mr_linear_bf16_kernel<KS, QG, MAXB, INF, WPE, DEPTH>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;
  }
  dim3 grid((unsigned)(g.tiles_per_xcd * (QG == 4 ? 1 : 4) * 8));
  hipLaunchKernelGGL((mr_linear_bf16_kernel<KS, QG, MAXB, INF, WPE, DEPTH>), grid, dim3(64 * MG_NW), lds, st, g);
  return hipGetLastError();
}

}  // namespace gkg
using namespace gkg;

// Bytes of the weight-fragment array gkg_mr_linear_bf16 expects for C channels:
// [4 conv groups][ci_pad / 8][co_pad] x 16 B, ci = co = C / 2, ci_pad = ci rounded up to 16, co_pad = co rounded up to 32;
// fragment (q, f, n) holds W[q][n][8 f .. 8 f + 7] as bf16 (zero beyond ci / co).
extern "C" size_t gkg_mr_linear_planes_bytes(int C) {
  if (C <= 0 || (C & 15)) return 0;
  const int ci = C / 2, ci_pad = (ci + 15) & ~15, co_pad = (ci + 31) & ~31;
  return (size_t)4 * (ci_pad / 8) * co_pad * 16;
}

// out (T, ldo) bf16 [:, 0 : 2C] = act(a * (Conv1x1_{groups=4}([x, max_k(src[idx] - x)] interleaved)) + c), token-major.
//   x (B, N, C) fp32, src (B, M, C) fp32 or NULL (self graph, M == N), nn_idx (B*G, N, k) int64, C = G * c,
//   wplanes: gkg_mr_linear_planes_bytes(C) bytes (layout above), a / cshift (2C) fp32, act 0 none / 1 GELU (erf).
static int mr_linear_bf16_impl(const float* x, const float* src, const int64_t* nn_idx, const uint16_t* nn16, const void* wplanes,
                               const float* a, const float* cshift, void* out, int ldo, int B, int G, int c, int N, int M, int k,
                               int act, void* stream) {
  if (!x || (!nn_idx && !nn16) || !wplanes || !a || !cshift || !out) return gkg_fail(GKG_ERR_NULL, "gkg_mr_linear_bf16: null pointer");
  if (B <= 0 || G <= 0 || c <= 0 || N <= 0 || M <= 0 || k <= 0 || k > 64 || act < 0 || act > 1)
    return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_linear_bf16: bad sizes (need > 0, k <= 64, act in 0..1)");
  const long long C = (long long)G * c;
  if ((C & 15) || (c & 3) || C > 768) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_linear_bf16: need C % 16 == 0, c % 4 == 0, C <= 768");
  if (!src && M != N) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_linear_bf16: self graph needs M == N");
  if (ldo < 2 * C || (ldo & 7)) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_linear_bf16: need ldo >= 2C and ldo % 8 == 0");
  MrGemmArgs g;
  g.x = x; g.src = src; g.nn_idx = nn_idx; g.nn16 = nn16; g.wp = (const uint4*)wplanes; g.a = a; g.cs = cshift;
  g.out = (uint16_t*)out; g.ldo = ldo;
  g.B = B; g.G = G; g.c = c; g.N = N; g.M = M; g.k = k; g.C = (int)C; g.Cq = (int)C / 4; g.ci = (int)C / 2; g.co = (int)C / 2;
  g.ci_pad = (g.ci + 15) & ~15; g.co_pad = (g.co + 31) & ~31; g.act = act;
  g.T = (long long)B * N;
  const long long tiles = (g.T + MG_ROWS - 1) / MG_ROWS;
  if (tiles * 4 > 0x7fffffffLL / 2) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_linear_bf16: too many tokens");
  g.tiles = (int)tiles;
  g.tiles_per_xcd = (int)((tiles + 7) / 8);
  hipStream_t st = (hipStream_t)stream;
  GkgProfScope prof(GKG_PROF_MR_FWD, st);
  // narrow layers: one workgroup per token tile with all 4 conv groups, while its LDS image stays below 64 KB
  const bool all_groups = mg_lds_bytes(g, 4) <= 64 * 1024 && 4 * (g.co_pad >> 5) <= (MG_NW / 2) * MG_MAXB;
  if ((g.co_pad >> 5) > (MG_NW / 2) * MG_MAXB) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_linear_bf16: C too large");
  hipError_t e;
  const bool narrow = all_groups && 4 * (g.co_pad >> 5) <= (MG_NW / 2) * 2;          // <= 2 blocks per wave: C <= 128
  if (narrow) e = k == 9 ? mg_launch<9, 4, 2, 1, 6, 2>(g, st) : (k == 18 ? mg_launch<18, 4, 2, 1, 4, 2>(g, st) : mg_launch<0, 4, 2, 1, 6, 2>(g, st));
  else if (all_groups) e = k == 9 ? mg_launch<9, 4>(g, st) : (k == 18 ? mg_launch<18, 4>(g, st) : mg_launch<0, 4>(g, st));
  else e = k == 9 ? mg_launch<9, 1>(g, st) : (k == 18 ? mg_launch<18, 1>(g, st) : mg_launch<0, 1>(g, st));
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "mr_linear_bf16_kernel");
}

extern "C" int gkg_mr_linear_bf16(const float* x, const float* src, const int64_t* nn_idx, const void* wplanes, const float* a,
                                  const float* cshift, void* out, int ldo, int B, int G, int c, int N, int M, int k, int act,
                                  void* stream) {
  return mr_linear_bf16_impl(x, src, nn_idx, nullptr, wplanes, a, cshift, out, ldo, B, G, c, N, M, k, act, stream);
}

// gkg_mr_linear_bf16 over the compact neighbour lists of gkg_knn_fwd_tm16 (u16 rows, M <= 65 536): same result.
extern "C" int gkg_mr_linear_bf16_nn16(const float* x, const float* src, const uint16_t* nn16, const void* wplanes, const float* a,
                                       const float* cshift, void* out, int ldo, int B, int G, int c, int N, int M, int k, int act,
                                       void* stream) {
  if (M > 65536) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_linear_bf16_nn16: M <= 65536 (u16 rows)");
  return mr_linear_bf16_impl(x, src, nullptr, nn16, wplanes, a, cshift, out, ldo, B, G, c, N, M, k, act, stream);
}
