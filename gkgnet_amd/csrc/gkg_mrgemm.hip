// gkg_mrgemm.hip — SURVEY §8 row g1 (inference): neighbour gather + max(x_j - x_i) as the A-OPERAND PRODUCER of the grouped
// 1x1 projection, BN (eval) + GELU in its epilogue.
//
// Reference chain (mmcls/models/backbones/vig_model/torch_vertex.py:47-62 + torch_nn.py:57-69):
//     x_i, x_j = batched_index_select(...)        (B*G, c, N, k) each
//     m        = max_k(x_j - x_i)
//     u        = interleave [x_0, m_0, x_1, m_1, ...]            (B, 2C, N, 1)
//     out      = GELU(BN(Conv2d(2C, 2C, 1, groups=4)(u)))
// In inference nothing downstream needs `u` (or `m`) in memory.  One workgroup owns 64 tokens x one conv group q:
//   phase 1  every thread gathers the k neighbour rows of (token, 4 channels) from L2 (token-major rows: one float4 per
//            neighbour), takes the max of the differences — the same arithmetic, in the same order, as mr_fwd_tm_kernel —
//            and writes the interleaved [x, m] values as bf16 (round-to-nearest-even) straight into the LDS image of the
//            GEMM's A tile: 64 rows x ci = C/2 input channels of this conv group;
//   phase 2  v_mfma_f32_32x32x16_bf16 over that tile: the wave (w & 1) takes row block (w & 1), column blocks (w >> 1),
//            (w >> 1) + 2, ...; the weights stream from L2 as pre-arranged 16-byte fragments [q][ci/8][co][8]
//            (register double-buffered), fp32 accumulation;
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

constexpr int MG_ROWS = 64;       // tokens per workgroup
constexpr int MG_MAXCB = 6;       // 32-column blocks per wave: co <= 384

struct MrGemmArgs {
  const float* x;          // (B, N, C) token-major fp32
  const float* src;        // (B, M, C) or null (self graph: src = x, M = N)
  const int64_t* nn_idx;   // (B*G, N, k)
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
__device__ __forceinline__ float mg_gelu(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f)); }

template <int KS>
__global__ __launch_bounds__(256, 2) void mr_linear_bf16_kernel(MrGemmArgs g) {
  extern __shared__ __align__(16) unsigned char mg_lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware map: workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  Every XCD gets a CONTIGUOUS
  // range of token tiles (whole images: a token's neighbours are rows of its own image), and the 4 conv groups of a tile
  // — same index rows, same token rows, different channel quarter — are adjacent in its dispatch order.
  const int lin = blockIdx.x;
  const int xcd = lin & 7, j = lin >> 3;
  const int tl = j >> 2, q = j & 3;
  const int tile = xcd * g.tiles_per_xcd + tl;
  if (tl >= g.tiles_per_xcd || tile >= g.tiles) return;
  const long long t0 = (long long)tile * MG_ROWS;
  const int pitch = (g.ci_pad + 8) * 2;                 // bytes; 16 B of padding: conflict-free 16-byte fragment reads
  const int C = g.C, N = g.N, M = g.M;
  const int k = KS > 0 ? KS : g.k;

  // ---------------------------------------------------------------- phase 1: the A tile = interleaved [x, max-relative]
  const int Q4 = g.Cq >> 2;                              // channel quads of this conv group
  const float* srcb = g.src ? g.src : g.x;
  for (int it = tid; it < MG_ROWS * Q4; it += 256) {
    const int tok = it / Q4, quad = it - tok * Q4;
    const long long t = t0 + tok;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t < g.T) {
      const int b = (int)(t / N), n = (int)(t - (long long)b * N);
      const int ch = q * g.Cq + 4 * quad;                // original channel (4 channels never straddle a k-NN group)
      const int grp = ch / g.c;
      const int64_t* ip = g.nn_idx + (((size_t)b * g.G + grp) * N + n) * k;
      const float4 xi = *reinterpret_cast<const float4*>(g.x + (size_t)t * C + ch);
      const float* sb = srcb + (size_t)b * M * C + ch;
      float4 best;
      if (KS > 0) {
        int id[KS > 0 ? KS : 1];
#pragma unroll
        for (int u = 0; u < KS; ++u) id[u] = mg_clamp(ip[u], M);
        float4 nb[KS > 0 ? KS : 1];
#pragma unroll
        for (int u = 0; u < KS; ++u) nb[u] = *reinterpret_cast<const float4*>(sb + (size_t)id[u] * C);
        best = make_float4(nb[0].x - xi.x, nb[0].y - xi.y, nb[0].z - xi.z, nb[0].w - xi.w);
#pragma unroll
        for (int u = 1; u < KS; ++u) {
          const float d0 = nb[u].x - xi.x, d1 = nb[u].y - xi.y, d2 = nb[u].z - xi.z, d3 = nb[u].w - xi.w;
          if (mg_takes(d0, best.x)) best.x = d0;
          if (mg_takes(d1, best.y)) best.y = d1;
          if (mg_takes(d2, best.z)) best.z = d2;
          if (mg_takes(d3, best.w)) best.w = d3;
        }
      } else {
        const float4 n0 = *reinterpret_cast<const float4*>(sb + (size_t)mg_clamp(ip[0], M) * C);
        best = make_float4(n0.x - xi.x, n0.y - xi.y, n0.z - xi.z, n0.w - xi.w);
        for (int u = 1; u < k; ++u) {
          const float4 nv = *reinterpret_cast<const float4*>(sb + (size_t)mg_clamp(ip[u], M) * C);
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
    *reinterpret_cast<uint4*>(mg_lds + tok * pitch + 16 * quad) = v;
  }
  if (g.ci_pad > g.ci && tid < MG_ROWS)                  // contraction padding (ci % 16 == 8): one zero fragment per row
    *reinterpret_cast<uint4*>(mg_lds + tid * pitch + 2 * g.ci) = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // ---------------------------------------------------------------- phase 2: (64 x ci) @ W_q^T on the bf16 matrix cores
  const int l31 = lane & 31, kg = lane >> 5;
  const int rb = w & 1, cb0 = w >> 1;                    // this wave: row block rb, column blocks cb0, cb0 + 2, ...
  const int ncb = g.co_pad >> 5;
  const int S = g.ci_pad >> 4;
  const uint4* wq = g.wp + (size_t)q * (g.ci_pad >> 3) * g.co_pad + l31;
  mg_f32x16 acc[MG_MAXCB];
#pragma unroll
  for (int u = 0; u < MG_MAXCB; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  uint4 bn[MG_MAXCB];
#pragma unroll
  for (int u = 0; u < MG_MAXCB; ++u)
    bn[u] = cb0 + 2 * u < ncb ? wq[(size_t)kg * g.co_pad + 32 * (cb0 + 2 * u)] : make_uint4(0, 0, 0, 0);
  const unsigned char* ap = mg_lds + (32 * rb + l31) * pitch + 16 * kg;
  for (int s = 0; s < S; ++s) {
    uint4 bc[MG_MAXCB];
#pragma unroll
    for (int u = 0; u < MG_MAXCB; ++u) bc[u] = bn[u];
    if (s + 1 < S) {
#pragma unroll
      for (int u = 0; u < MG_MAXCB; ++u)
        if (cb0 + 2 * u < ncb) bn[u] = wq[(size_t)(2 * (s + 1) + kg) * g.co_pad + 32 * (cb0 + 2 * u)];
    }
    const mg_bf16x8 av = __builtin_bit_cast(mg_bf16x8, *reinterpret_cast<const uint4*>(ap + 32 * s));
#pragma unroll
    for (int u = 0; u < MG_MAXCB; ++u)
      if (cb0 + 2 * u < ncb)                             // wave-uniform
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(mg_bf16x8, bc[u]), acc[u], 0, 0, 0);
  }

  // ---------------------------------------------------------------- phase 3: BN (eval) + activation -> bf16 -> rows out
  __syncthreads();                                       // every wave is done reading the A tile: reuse it as the stage
#pragma unroll
  for (int u = 0; u < MG_MAXCB; ++u) {
    const int col = 32 * (cb0 + 2 * u) + l31;
    if (cb0 + 2 * u < ncb && col < g.co) {
      const float av = g.a[q * g.co + col], cv = g.cs[q * g.co + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * kg;
        float o = __builtin_fmaf(av, acc[u][r], cv);
        if (g.act == 1) o = mg_gelu(o);
        *reinterpret_cast<uint16_t*>(mg_lds + row * pitch + 2 * col) = __builtin_bit_cast(uint16_t, (__bf16)o);
      }
    }
  }
  __syncthreads();
  const int CH = g.co >> 3;                              // 16-byte chunks per output row segment
  for (int it = tid; it < MG_ROWS * CH; it += 256) {
    const int tok = it / CH, ck = it - tok * CH;
    const long long t = t0 + tok;
    if (t < g.T)
      *reinterpret_cast<uint4*>(g.out + (size_t)t * g.ldo + q * g.co + 8 * ck) =
          *reinterpret_cast<const uint4*>(mg_lds + tok * pitch + 16 * ck);
  }
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
extern "C" int gkg_mr_linear_bf16(const float* x, const float* src, const int64_t* nn_idx, const void* wplanes, const float* a,
                                  const float* cshift, void* out, int ldo, int B, int G, int c, int N, int M, int k, int act,
                                  void* stream) {
  if (!x || !nn_idx || !wplanes || !a || !cshift || !out) return gkg_fail(GKG_ERR_NULL, "gkg_mr_linear_bf16: null pointer");
  if (B <= 0 || G <= 0 || c <= 0 || N <= 0 || M <= 0 || k <= 0 || k > 64 || act < 0 || act > 1)
    return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_linear_bf16: bad sizes (need > 0, k <= 64, act in 0..1)");
  const long long C = (long long)G * c;
  if ((C & 15) || (c & 3) || C > 768) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_linear_bf16: need C % 16 == 0, c % 4 == 0, C <= 768");
  if (!src && M != N) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_linear_bf16: self graph needs M == N");
  if (ldo < 2 * C || (ldo & 7)) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_linear_bf16: need ldo >= 2C and ldo % 8 == 0");
  MrGemmArgs g;
  g.x = x; g.src = src; g.nn_idx = nn_idx; g.wp = (const uint4*)wplanes; g.a = a; g.cs = cshift;
  g.out = (uint16_t*)out; g.ldo = ldo;
  g.B = B; g.G = G; g.c = c; g.N = N; g.M = M; g.k = k; g.C = (int)C; g.Cq = (int)C / 4; g.ci = (int)C / 2; g.co = (int)C / 2;
  g.ci_pad = (g.ci + 15) & ~15; g.co_pad = (g.co + 31) & ~31; g.act = act;
  g.T = (long long)B * N;
  const long long tiles = (g.T + MG_ROWS - 1) / MG_ROWS;
  if (tiles * 4 > 0x7fffffffLL / 2) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_linear_bf16: too many tokens");
  g.tiles = (int)tiles;
  g.tiles_per_xcd = (int)((tiles + 7) / 8);
  const size_t lds = (size_t)MG_ROWS * (g.ci_pad + 8) * 2;
  dim3 grid((unsigned)(g.tiles_per_xcd * 4 * 8));
  hipStream_t st = (hipStream_t)stream;
  GkgProfScope prof(GKG_PROF_MR_FWD, st);
  if (k == 9) hipLaunchKernelGGL((mr_linear_bf16_kernel<9>), grid, dim3(256), lds, st, g);
  else if (k == 18) hipLaunchKernelGGL((mr_linear_bf16_kernel<18>), grid, dim3(256), lds, st, g);
  else hipLaunchKernelGGL((mr_linear_bf16_kernel<0>), grid, dim3(256), lds, st, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "mr_linear_bf16_kernel");
}
