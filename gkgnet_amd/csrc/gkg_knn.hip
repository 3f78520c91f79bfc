// gkg_knn.hip — fused dilated k-NN graph construction for MI355X (gfx950 / CDNA4).
//
// Replaces the reference's  F.normalize -> (|x|^2 - 2 x.y^T + |y|^2) -> += relative_pos -> topk -> [::d]
// chain (mmcls/models/backbones/vig_model/torch_edge.py:9-51, 54-106, 139-176) without ever
// materialising the (BG,N,M) distance matrix.
//
// Pipeline (all on the caller's stream):
//   1. token_prep_kernel   : (one launch for queries and keys) per token, ordered-fma L2 norm over the group's channels, normalised fp32
//                            copy th (BG,cpad,T) (cpad = c rounded up to a multiple of 8, zero padded) + |th|^2.
//   2. knn_tile_kernel<KD> : one workgroup = 64 queries of one (b,g) problem x one key split; its 4 waves
//                            take key tiles of 32 round-robin.  Per key tile a wave runs the contraction on
//                            the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, == an ordered fmaf
//                            chain over channels) with keys as the A operand (streamed from L2/HBM in the
//                            reference's native channel-major layout: a lane reads y[ch][m0+lane]) and the
//                            64 staged queries as two B operands from LDS.  The two 32x32 accumulators are
//                            exchanged with v_permlane32_swap so that lane l owns query n0+l and all 32
//                            key distances of the tile; each lane keeps a sorted top-KD list of fp64
//                            (distance, index) keys in registers (v_min_f64 + v_max_f64 per slot).  The 4 per-wave lists of a
//                            query are merged through LDS; ranks 0,d,2d,.. go out as int64.
//   3. knn_merge_kernel    : only when the keys were split over several workgroups (few queries, many
//                            keys — the label graph): merges the S partial lists of each query.
//
// Arithmetic contract: include/gkg_hip.h (restated in oracle/gkg_oracle.c).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gkg_knn_common.h"
#include "gkg_knn_tile.h"

namespace gkg {

// ------------------------------------------------------------------------------------------ prep
struct PrepStrides { int G; size_t sb, sg, sc, sn; int chunk; };      // chunk > 0 (token-major fp32 only): the x half of an XM
                                                                        // buffer (gkg_common.h) — channel ch at column xm_col(ch, chunk)

// One thread per token.  The two ordered fma chains over the channels are inherently serial per token and the
// problem has few tokens per CU (cfg2: 160), so the kernel is latency-bound.  Structure: (1) pull the token's
// whole channel column into a thread-private LDS column with 16 loads in flight and no arithmetic in between
// (token-major input: float4 loads of the contiguous channel run), (2) run both chains out of LDS, (3) store
// the normalised copy coalesced along the token axis.  PT (tokens per workgroup) shrinks with c so the column
// block always fits: PT * c * 4 B <= 48 KB.
// One launch prepares the queries AND (bipartite graphs) the keys: workgroups blockIdx.x >= nbx1 take the second
// tensor — for the short label-branch problems a second dependent launch costs as much as the work itself.
// tb != null: the normalised copy is written as bf16 in OCTET-major planes (bg, cp16 / 8, n, 8), zero-padded to cp16
// channels — a matrix-core operand fragment (one token, 8 consecutive channels) is 16 contiguous bytes and the fragments of
// consecutive tokens are adjacent, so a wave's fragment load / store touches 8 full cache lines instead of 32+ partial ones
// (measured: the token-major (bg, n, cp16) form left the contraction bound by the texture-address path) — instead of fp32
// channel-major th.
// tb_lo != null (prefilter mode, see knn_pf_kernel): besides the fp32 channel-major copy th (the exact re-rank reads it) the
// two leading bf16 terms of every normalised value, hi = bf16(v) and lo = bf16(v - hi), are written as token-major
// (bg, n, cp16) planes (tb = hi, tb_lo = lo) for the matrix-core prefilter.
// Prefilter planes, round 5: channels c and c + 1 of the HI plane carry the two leading bf16 terms of the token's |th|^2 (LO plane:
// zeros) — the kernel stages a query's copy of those two channels as (1, 1), so the matrix cores add |y|^2 to every distance and
// the selection loses a v_readlane + s_nop + v_add per candidate; the planes have `rows` = Tn rounded up to 32 rows, the pad rows
// hold MASKED_SQ there (keys past M mask themselves).  cp16 >= c + 2 in that mode.
// raw != null (round 6, queries of a fused block only): the input is a projection's PRE-BN output and this pass is also its
// BN-apply (kernel argument `d`: scale / shift derived from the projection's fp64 column sums, or given) — every token is
// affine-transformed as it is gathered, stored to `raw` (a token-major view: row pitch raw_ld, chunk raw_chunk — the x half of
// the grouped projection's operand buffer) and normalised from the same registers: the Grapher's fc1 BN-apply and the k-NN's
// token preparation in ONE pass over the tokens (reference torch_vertex.py:326 -> torch_edge.py:167-173).
// res_tm / nchw (the KEYS of a label graph produced by the Grapher in front of it, reference torch_vertex.py:331 -> :392-403): the
// pass is that Grapher's fc2 BN-apply — a y + c + res_tm[token] goes to `raw` (the token-major companion the label block reads as
// values), to `nchw` (B, C, Tn: the block's channel-major output) and, normalised, into the label k-NN's workspace as its keys.
struct PrepSet { const void* t; float* th; float* sq; int Tn; PrepStrides ps; uint16_t* tb; int cp16; uint16_t* tb_lo; int rows;
                 float* raw; int raw_ld, raw_chunk; const float* res_tm; float* nchw; };

// scale / shift of the group's c channels into LDS (tab[0..c) = a, tab[c..2c) = shift): derived from the column sums like every
// BN-apply pass (gkg_common.h bn_derive_channel; the first workgroup of image 0's groups writes the saved statistics and updates the
// running ones), or copied from the caller's a / c
__device__ __forceinline__ void prep_affine_table(const BnDerive& d, const float* __restrict__ a_in, const float* __restrict__ c_in,
                                                  float* tab, int c, int C, int bg, int G, int nthreads) {
  const int g = bg % G;
  const bool side = blockIdx.x == 0 && bg < G;                 // image 0: one workgroup per group
  for (int ch = threadIdx.x; ch < c; ch += nthreads) {
    const size_t o = (size_t)g * c + ch;
    float av, cv;
    if (d.sums) bn_derive_channel(d, o, d.sums[o], d.sums[C + o], side, av, cv);
    else { av = a_in[o]; cv = c_in[o]; }
    tab[ch] = av; tab[c + ch] = cv;
  }
  if (d.sums && blockIdx.x == 0 && bg == 0) {
    if (threadIdx.x == 0 && d.nbt) *d.nbt += 1;
    for (size_t i = threadIdx.x; i < d.zero_doubles; i += nthreads) d.zero_buf[i] = 0.0;
  }
  __syncthreads();
}

template <typename T, bool NORM, int PT, bool PROD>
__device__ __forceinline__ void token_prep_body(const PrepSet& s1, const PrepSet& s2, int nbx1, int c, int cpad, const BnDerive& d,
                                                const float* __restrict__ aff_a, const float* __restrict__ aff_c) {
  extern __shared__ float col[];          // [c][PT] (+ [2][c] scale / shift: PROD)
  const bool second = (int)blockIdx.x >= nbx1;
  const PrepSet& S = second ? s2 : s1;
  constexpr bool affine = PROD;                                  // a producer launch has ONE set: the BN-applied operand
  float* tab = col + (size_t)c * PT;
  if (affine) prep_affine_table(d, aff_a, aff_c, tab, c, s1.ps.G * c, blockIdx.y, s1.ps.G, PT);
  const T* __restrict__ t = static_cast<const T*>(S.t);
  float* __restrict__ th = S.th;
  float* __restrict__ sq = S.sq;
  const int Tn = S.Tn;
  const PrepStrides ps = S.ps;
  const int n = ((int)blockIdx.x - (second ? nbx1 : 0)) * PT + threadIdx.x;
  const int bg = blockIdx.y;
  if (n >= Tn) {
    if (S.tb_lo && n < S.rows) {                    // prefilter planes: pad rows up to a multiple of 32 mask themselves
      uint4* ohi = reinterpret_cast<uint4*>(S.tb) + (size_t)bg * (S.cp16 >> 3) * S.rows + n;
      uint4* olo = reinterpret_cast<uint4*>(S.tb_lo) + (size_t)bg * (S.cp16 >> 3) * S.rows + n;
      for (int c0 = 0; c0 < S.cp16; c0 += 8) {
        uint32_t wv[4] = {0u, 0u, 0u, 0u};
        if (c >= c0 && c < c0 + 8) {
          const uint32_t mk = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)MASKED_SQ);
          wv[(c - c0) >> 1] = ((c - c0) & 1) ? (mk << 16) : mk;
        }
        ohi[(size_t)(c0 >> 3) * S.rows] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        olo[(size_t)(c0 >> 3) * S.rows] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    return;
  }
  // element (bg = b*G + g, ch, n) of the input lives at  b*sb + g*sg + ch*sc + n*sn
  const T* tp = t + (size_t)(bg / ps.G) * ps.sb + (size_t)(bg % ps.G) * ps.sg + (size_t)n * ps.sn;
  float* cp = col + threadIdx.x;
  // ---- (1) gather the column
  if (sizeof(T) == 4 && ps.sc == 1 && (c & 3) == 0 && ((ps.sn | ps.sg | ps.sb) & 3) == 0) {
    const int goff = (bg % ps.G) * (int)ps.sg;      // the group's first channel; quad q sits xm_col(goff + 4 q) - goff floats into tp
    const float* tpf = reinterpret_cast<const float*>(tp) - goff;
    int q = 0;
    float* rawp = affine ? S.raw + ((size_t)(bg / ps.G) * Tn + n) * S.raw_ld : nullptr;
    auto apply = [&](float4& v, int q4) __attribute__((always_inline)) {       // BN-apply of 4 channels + the raw store
      if (affine) {
        v.x = __builtin_fmaf(tab[q4], v.x, tab[c + q4]); v.y = __builtin_fmaf(tab[q4 + 1], v.y, tab[c + q4 + 1]);
        v.z = __builtin_fmaf(tab[q4 + 2], v.z, tab[c + q4 + 2]); v.w = __builtin_fmaf(tab[q4 + 3], v.w, tab[c + q4 + 3]);
        if (S.res_tm) {
          const float4 r4 = *reinterpret_cast<const float4*>(S.res_tm + ((size_t)(bg / ps.G) * Tn + n) * (size_t)(ps.G * c) + goff + q4);
          v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
        *reinterpret_cast<float4*>(rawp + xm_col(goff + q4, S.raw_chunk)) = v;
      }
    };
    for (; q + 8 <= (c >> 2); q += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(tpf + xm_col(goff + 4 * (q + u), ps.chunk));
#pragma unroll
      for (int u = 0; u < 8; ++u) apply(v[u], 4 * (q + u));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        cp[(4 * (q + u) + 0) * PT] = v[u].x; cp[(4 * (q + u) + 1) * PT] = v[u].y;
        cp[(4 * (q + u) + 2) * PT] = v[u].z; cp[(4 * (q + u) + 3) * PT] = v[u].w;
      }
    }
    for (; q < (c >> 2); ++q) {
      float4 v = *reinterpret_cast<const float4*>(tpf + xm_col(goff + 4 * q, ps.chunk));
      apply(v, 4 * q);
      cp[(4 * q + 0) * PT] = v.x; cp[(4 * q + 1) * PT] = v.y; cp[(4 * q + 2) * PT] = v.z; cp[(4 * q + 3) * PT] = v.w;
    }
  } else {
    int ch = 0;
    for (; ch + 16 <= c; ch += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = ldf(tp + (size_t)(ch + u) * ps.sc);
#pragma unroll
      for (int u = 0; u < 16; ++u) cp[(ch + u) * PT] = v[u];
    }
    for (; ch < c; ++ch) cp[ch * PT] = ldf(tp + (size_t)ch * ps.sc);
  }
  // ---- (2) ordered chains out of LDS (own column: no barrier needed)
  float den = 1.0f;
  if (NORM) {
    float s = 0.0f;
#pragma unroll 8
    for (int ch = 0; ch < c; ++ch) { const float v = cp[ch * PT]; s = __builtin_fmaf(v, v, s); }
    den = fmaxf(sqrtf(s), 1e-12f);
  }
  // ---- (3) normalise, store, |th|^2
  float q2 = 0.0f;
  if (S.tb_lo) {                                    // prefilter mode: bf16 hi / lo planes (token-major) + th / sq below
    // the fp32 copy and |th|^2 first (the exact re-rank reads them): the two leading bf16 terms of |th|^2 ride in channels c, c + 1
    float q2p = 0.0f;
    {
      float* op = th + (size_t)bg * cpad * Tn + n;
      float* np_ = (affine && S.nchw) ? S.nchw + ((size_t)(bg / ps.G) * (ps.G * c) + (size_t)(bg % ps.G) * c) * Tn + n : nullptr;
#pragma unroll 8
      for (int ch = 0; ch < c; ++ch) {
        float v = cp[ch * PT];
        if (np_) np_[(size_t)ch * Tn] = v;
        if (NORM) { v = v / den; cp[ch * PT] = v; }      // the planes below split the normalised value: one division per element
        op[(size_t)ch * Tn] = v;
        q2p = __builtin_fmaf(v, v, q2p);
      }
      for (int chp = c; chp < cpad; ++chp) op[(size_t)chp * Tn] = 0.0f;
      sq[(size_t)bg * Tn + n] = q2p;
    }
    const float q2h = __uint_as_float(((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)q2p)) << 16);
    const float q2l = q2p - q2h;
    // octet-major planes (bg, cp16 / 8, rows, 8): consecutive tokens are consecutive 16-byte fragments
    const int R = S.rows;
    uint4* ohi = reinterpret_cast<uint4*>(S.tb) + (size_t)bg * (S.cp16 >> 3) * R + n;
    uint4* olo = reinterpret_cast<uint4*>(S.tb_lo) + (size_t)bg * (S.cp16 >> 3) * R + n;
    for (int c0 = 0; c0 < S.cp16; c0 += 8) {
      uint32_t wh[4], wl[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e0 = c0 + 2 * u, e1 = e0 + 1;
        float v0 = e0 < c ? cp[e0 * PT] : 0.0f;
        float v1 = e1 < c ? cp[e1 * PT] : 0.0f;
        float l0 = v0 - __uint_as_float(((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)v0)) << 16);
        float l1 = v1 - __uint_as_float(((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)v1)) << 16);
        if (e0 == c) { v0 = q2h; l0 = 0.0f; } else if (e0 == c + 1) { v0 = q2l; l0 = 0.0f; }
        if (e1 == c) { v1 = q2h; l1 = 0.0f; } else if (e1 == c + 1) { v1 = q2l; l1 = 0.0f; }
        wh[u] = pack_bf16x2(v0, v1);
        wl[u] = pack_bf16x2(l0, l1);
      }
      ohi[(size_t)(c0 >> 3) * R] = make_uint4(wh[0], wh[1], wh[2], wh[3]);
      olo[(size_t)(c0 >> 3) * R] = make_uint4(wl[0], wl[1], wl[2], wl[3]);
    }
    return;
  }
  if (S.tb) {                                       // bf16 token-major copy: 16-byte stores of 8 channels
    uint4* ob = reinterpret_cast<uint4*>(S.tb) + (size_t)bg * (S.cp16 >> 3) * Tn + n;      // octet-major (bg, cp16/8, Tn, 8)
    for (int c0 = 0; c0 < S.cp16; c0 += 8) {
      uint32_t wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v0 = c0 + 2 * u < c ? cp[(c0 + 2 * u) * PT] : 0.0f;
        float v1 = c0 + 2 * u + 1 < c ? cp[(c0 + 2 * u + 1) * PT] : 0.0f;
        if (NORM) { v0 = v0 / den; v1 = v1 / den; }
        q2 = __builtin_fmaf(v0, v0, q2);
        q2 = __builtin_fmaf(v1, v1, q2);
        wv[u] = pack_bf16x2(v0, v1);
      }
      ob[(size_t)(c0 >> 3) * Tn] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
    sq[(size_t)bg * Tn + n] = q2;
    return;
  }
  float* op = th + (size_t)bg * cpad * Tn + n;
  float* np_ = (affine && S.nchw) ? S.nchw + ((size_t)(bg / ps.G) * (ps.G * c) + (size_t)(bg % ps.G) * c) * Tn + n : nullptr;
#pragma unroll 8
  for (int ch = 0; ch < c; ++ch) {
    float v = cp[ch * PT];
    if (np_) np_[(size_t)ch * Tn] = v;
    if (NORM) v = v / den;
    op[(size_t)ch * Tn] = v;
    q2 = __builtin_fmaf(v, v, q2);
  }
  for (int chp = c; chp < cpad; ++chp) op[(size_t)chp * Tn] = 0.0f;
  sq[(size_t)bg * Tn + n] = q2;
}

template <typename T, bool NORM, int PT>
__global__ __launch_bounds__(PT) void token_prep_kernel(PrepSet s1, PrepSet s2, int nbx1, int c, int cpad) {
  token_prep_body<T, NORM, PT, false>(s1, s2, nbx1, c, cpad, BnDerive{}, nullptr, nullptr);
}
// The same pass as a projection's BN-apply (PrepSet::raw): what a trace shows for the Grapher's fc1 / the label block's fc1 / the
// Grapher's fc2 in front of a label block — one launch = BN-apply (+ residual, + both layouts) + the k-NN's token preparation
template <bool NORM, int PT>
__global__ __launch_bounds__(PT) void bn_apply_knn_prep_kernel(PrepSet s1, int c, int cpad, BnDerive d, const float* __restrict__ aff_a,
                                                               const float* __restrict__ aff_c) {
  token_prep_body<float, NORM, PT, true>(s1, s1, 0x7fffffff, c, cpad, d, aff_a, aff_c);
}

// Small problems (a few ten thousand token-groups: the 18 x 18 stages, the label graphs): one thread per token-group leaves
// less than one wave per SIMD, each walking three dependent c-long passes (gather, norm chain, divide + |.|^2 chain) — the
// launch is pure latency (13-14 us at cfg2).  Cooperative form for fp32 token-major input: a 256-thread workgroup owns TK
// tokens of one (b, g); all threads gather the tile (16 lanes per token row: 256-byte runs), every wave recomputes the
// ordered norm chain of its token from LDS (no exchange), the divisions and the channel-major stores are spread over
// 256 / TK channel lanes per token, and the ordered |th|^2 chain runs on the normalised tile.  Same operations in the same
// order per value: bit-identical outputs.
template <bool NORM, int TK, bool PROD>
__device__ __forceinline__ void token_prep_coop_body(const PrepSet& s1, const PrepSet& s2, int nbx1, int c, int cpad, const BnDerive& d,
                                                     const float* __restrict__ aff_a, const float* __restrict__ aff_c) {
  extern __shared__ float col[];          // [c][TK + 1] (+ [2][c] scale / shift: PROD)
  constexpr int LP = TK + 1, CL = 256 / TK;      // channel lanes per token
  const int tid = threadIdx.x;
  const bool second = (int)blockIdx.x >= nbx1;
  const PrepSet& S = second ? s2 : s1;
  constexpr bool affine = PROD;
  float* tab = col + (size_t)c * LP;
  if (affine) prep_affine_table(d, aff_a, aff_c, tab, c, s1.ps.G * c, blockIdx.y, s1.ps.G, 256);
  const int Tn = S.Tn;
  const PrepStrides ps = S.ps;
  const int n0 = ((int)blockIdx.x - (second ? nbx1 : 0)) * TK;
  const int bg = blockIdx.y;
  const int nt = min(TK, Tn - n0);
  const float* tp = static_cast<const float*>(S.t) + (size_t)(bg / ps.G) * ps.sb + (size_t)(bg % ps.G) * ps.sg + (size_t)n0 * ps.sn;
  const int c4 = c >> 2;
  const int goff = (bg % ps.G) * (int)ps.sg;      // the group's first channel (token-major: sg == c)
  // ---- (1) gather: lane (tid & 15) walks the float4s of the rows tid >> 4, + 16, ...
  for (int q = tid & 15; q < c4; q += 16) {
    float4 v[TK / 16];
    const int xo = xm_col(goff + 4 * q, ps.chunk) - goff;
#pragma unroll
    for (int u = 0; u < TK / 16; ++u) {
      const int tok = (tid >> 4) + 16 * u;
      v[u] = tok < nt ? *reinterpret_cast<const float4*>(tp + (size_t)tok * ps.sn + xo) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (affine) {                                   // BN-apply of the 4 channels + the raw store (the x half of the operand buffer)
      const float4 a4 = *reinterpret_cast<const float4*>(tab + 4 * q), c4 = *reinterpret_cast<const float4*>(tab + c + 4 * q);
      const int ro = xm_col(goff + 4 * q, S.raw_chunk);
#pragma unroll
      for (int u = 0; u < TK / 16; ++u) {
        const int tok = (tid >> 4) + 16 * u;
        v[u].x = __builtin_fmaf(a4.x, v[u].x, c4.x); v[u].y = __builtin_fmaf(a4.y, v[u].y, c4.y);
        v[u].z = __builtin_fmaf(a4.z, v[u].z, c4.z); v[u].w = __builtin_fmaf(a4.w, v[u].w, c4.w);
        if (tok < nt) {
          const size_t trow = (size_t)(bg / ps.G) * Tn + n0 + tok;
          if (S.res_tm) {
            const float4 r4 = *reinterpret_cast<const float4*>(S.res_tm + trow * (size_t)(ps.G * c) + goff + 4 * q);
            v[u].x += r4.x; v[u].y += r4.y; v[u].z += r4.z; v[u].w += r4.w;
          }
          *reinterpret_cast<float4*>(S.raw + trow * S.raw_ld + ro) = v[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < TK / 16; ++u) {
      const int tok = (tid >> 4) + 16 * u;
      col[(4 * q + 0) * LP + tok] = v[u].x; col[(4 * q + 1) * LP + tok] = v[u].y;
      col[(4 * q + 2) * LP + tok] = v[u].z; col[(4 * q + 3) * LP + tok] = v[u].w;
    }
  }
  __syncthreads();
  const int tok = tid % TK, cl = tid / TK;
  const float* cp = col + tok;
  // ---- (2) the token's ordered norm chain (every channel lane recomputes it: no exchange)
  float den = 1.0f;
  if (NORM) {
    float s = 0.0f;
#pragma unroll 8
    for (int ch = 0; ch < c; ++ch) { const float v = cp[ch * LP]; s = __builtin_fmaf(v, v, s); }
    den = fmaxf(sqrtf(s), 1e-12f);
  }
  __syncthreads();                          // everyone has read the raw tile
  // ---- (3) normalise + store: channel lane cl takes channels cl, cl + CL, ...
  float* op = S.th + (size_t)bg * cpad * Tn + n0 + tok;
  float* np_ = (affine && S.nchw && tok < nt) ? S.nchw + ((size_t)(bg / ps.G) * (ps.G * c) + (size_t)(bg % ps.G) * c) * Tn + n0 + tok : nullptr;
#pragma unroll 4
  for (int ch = cl; ch < c; ch += CL) {
    float v = cp[ch * LP];
    if (np_) np_[(size_t)ch * Tn] = v;                    // the block's channel-major output (coalesced along the tokens)
    if (NORM) v = v / den;
    col[ch * LP + tok] = v;
    if (tok < nt) op[(size_t)ch * Tn] = v;
  }
  if (tok < nt)
    for (int chp = c + cl; chp < cpad; chp += CL) op[(size_t)chp * Tn] = 0.0f;
  __syncthreads();
  // ---- (4) ordered |th|^2 chain
  if (cl == 0 && tok < nt) {
    float q2 = 0.0f;
#pragma unroll 8
    for (int ch = 0; ch < c; ++ch) { const float v = cp[ch * LP]; q2 = __builtin_fmaf(v, v, q2); }
    S.sq[(size_t)bg * Tn + n0 + tok] = q2;
  }
}

template <bool NORM, int TK>
__global__ __launch_bounds__(256) void token_prep_coop_kernel(PrepSet s1, PrepSet s2, int nbx1, int c, int cpad) {
  token_prep_coop_body<NORM, TK, false>(s1, s2, nbx1, c, cpad, BnDerive{}, nullptr, nullptr);
}
template <bool NORM, int TK>
__global__ __launch_bounds__(256) void bn_apply_knn_prep_coop_kernel(PrepSet s1, int c, int cpad, BnDerive d, const float* __restrict__ aff_a,
                                                                     const float* __restrict__ aff_c) {
  token_prep_coop_body<NORM, TK, true>(s1, s1, 0x7fffffff, c, cpad, d, aff_a, aff_c);
}

// ------------------------------------------------------------------------------------------ split merge
// One thread per partial-list ELEMENT: its final rank = its position in its own (sorted) list + the number of
// lexicographically smaller (dist, idx) pairs in every other split's list (binary search).  Ranks are unique,
// so ranks 0,d,2d,.. < kd are written straight to their output slot — no serial merge.
__global__ __launch_bounds__(256) void knn_merge_kernel(const float* __restrict__ part_v, const int* __restrict__ part_i,
                                                        int64_t* __restrict__ nn_idx, int64_t* __restrict__ center,
                                                        uint16_t* __restrict__ nn16,
                                                        int S, int BG, int N, int M, int k, int dilation, int kd) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;        // (s, q, p) flattened as s*nq*kd + q*kd + p
  const size_t nq = (size_t)BG * N;
  const size_t stride = nq * kd;                                   // between splits
  if (e >= stride * S) return;
  const int s = (int)(e / stride);
  const size_t rem = e - (size_t)s * stride;
  const size_t q = rem / kd;
  const int p = (int)(rem - q * kd);
  const float v = part_v[e];
  const int id = part_i[e];
  // the centre plane does not depend on the distances: split 0's threads write it for every output slot, so ranks that
  // no finite candidate claims (non-finite inputs only) still carry a defined centre index
  if (center && s == 0 && p < k) center[q * k + p] = (int64_t)(q % N);
  if (id == 0x7fffffff) return;                                    // padding of a short list
  int rank = p;
  for (int s2 = 0; s2 < S && rank < kd; ++s2) {
    if (s2 == s) continue;
    const float* ov = part_v + (size_t)s2 * stride + q * kd;
    const int* oi = part_i + (size_t)s2 * stride + q * kd;
    int lo = 0, hi = kd;                                           // first entry >= (v, id)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const float mv = ov[mid]; const int mi = oi[mid];
      const bool less = (mv < v) || (mv == v && mi < id);
      if (less) lo = mid + 1; else hi = mid;
    }
    rank += lo;
  }
  if (rank < kd && rank % dilation == 0) {
    const int outj = rank / dilation;
    if (outj < k) {
      const int bc = id < M ? id : 0;               // masked tail keys can only rank here on non-finite inputs
      if (nn16) nn16[q * k + outj] = (uint16_t)bc;
      else nn_idx[q * k + outj] = bc;
    }
  }
}

}  // namespace gkg

// ================================================================================================ host side
using namespace gkg;

#if defined(KNN_TIMELINE) || defined(KNN_ABLATE)
static void* gkg_knn_tl_buf = nullptr;      // measurement builds only (tools/ubench/knn_timeline.py, knn_ablate.py)
extern "C" void gkg_debug_set_knn_timeline(void* buf) { gkg_knn_tl_buf = buf; }
#endif

// list sizes of the tile kernel: the lengths GKGNet's configurations use (k*d = 9, 18, 27; pvig_m k = 18: 18, 36) plus 16
// and 64 to cover everything else (each instantiation is a kernel of its own: 6 sizes x 2 x 2 x 3 selection forms)
static const int kListSizes[] = {9, 18, 27, 36, 64};   // k*d of GKGNet's layers + one catch-all (10..18 share the 18-entry list)

static int pick_list(int kd) {
  for (int s : kListSizes) if (s >= kd) return s;
  return -1;
}

static int pick_splits(int BG, int N, int M) {
  const int qtiles = (N + QT - 1) / QT;
  const int ktiles = (M + KT - 1) / KT;
  const long wgs = (long)qtiles * BG;
  int S = 1;
  if (wgs < 256) {                                 // fewer workgroups than CUs: split the keys
    S = (int)((512 + wgs - 1) / wgs);
    const int smax = (ktiles + 7) / 8;          // keep >= 8 key tiles (2 per wave) per split
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    if (S > 64) S = 64;
  }
  return S;
}

struct KnnPlan {
  int cpad, kd, KD, S, tps;
  size_t off_xh, off_yh, off_sqx, off_sqy, off_pv, off_pi, off_xp, off_yp, off_flags, total;
};

static int make_plan(int BG, int c, int N, int M, int k, int dilation, bool has_y, KnnPlan* p) {
  if (BG <= 0 || c <= 0 || N <= 0 || M <= 0 || k <= 0 || dilation <= 0) return GKG_ERR_SHAPE;
  if ((long)k * dilation > M) return GKG_ERR_SHAPE;
  if (!has_y && M != N) return GKG_ERR_SHAPE;
  p->cpad = (c + 7) & ~7;          // zero-padded channels: fma(0,0,acc) == acc keeps the chain exact
  p->kd = k * dilation;
  p->KD = pick_list(p->kd);
  if (p->KD < 0) return GKG_ERR_UNSUPPORTED;
  if ((size_t)p->cpad * QT * sizeof(float) > 150 * 1024) return GKG_ERR_UNSUPPORTED;   // c <= 600
  if (BG > 65535) return GKG_ERR_UNSUPPORTED;           // token_prep uses grid.y = BG
  if ((long)M + KT > (long)IDX_BITS) return GKG_ERR_UNSUPPORTED;   // key indices live in 29 bits of the list keys
  p->S = pick_splits(BG, N, M);
  const int ktiles = (M + KT - 1) / KT;
  p->tps = (ktiles + p->S - 1) / p->S;
  p->S = (ktiles + p->tps - 1) / p->tps;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = 0;
  p->off_xh = o; o = al(o + sizeof(float) * (size_t)BG * p->cpad * N);
  p->off_sqx = o; o = al(o + sizeof(float) * ((size_t)BG * N + 32));
  if (has_y) {
    p->off_yh = o; o = al(o + sizeof(float) * (size_t)BG * p->cpad * M);
    p->off_sqy = o; o = al(o + sizeof(float) * ((size_t)BG * M + 32));
  } else {
    p->off_yh = p->off_xh; p->off_sqy = p->off_sqx;
  }
  if (p->S > 1) {
    p->off_pv = o; o = al(o + sizeof(float) * (size_t)p->S * BG * N * p->kd);
    p->off_pi = o; o = al(o + sizeof(int) * (size_t)p->S * BG * N * p->kd);
    p->off_xp = p->off_yp = p->off_flags = 0;
  } else {
    p->off_pv = p->off_pi = 0;
    // prefilter mode (un-split problems): bf16 hi + lo planes (BG, T, cp16) of the queries and keys
    const size_t cp16 = (size_t)((c + 2 + 15) & ~15);               // + the two |th|^2 channels (see PrepSet)
    const size_t Nr = (size_t)((N + 31) & ~31), Mr = (size_t)((M + 31) & ~31);
    p->off_xp = o; o = al(o + 2 * sizeof(uint16_t) * (size_t)BG * Nr * cp16);
    if (has_y) { p->off_yp = o; o = al(o + 2 * sizeof(uint16_t) * (size_t)BG * Mr * cp16); }
    else p->off_yp = p->off_xp;
    p->off_flags = o; o = al(o + sizeof(int) * (size_t)((N + QT - 1) / QT) * ((BG + 7) / 8) * 8);   // one per workgroup
  }
  p->total = o;
  return 0;
}

extern "C" size_t gkg_knn_workspace_bytes(int BG, int c, int N, int M, int k, int dilation, int dtype, unsigned flags) {
  (void)dtype; (void)flags;
  KnnPlan p;
  // y presence is unknown here: budget for it (upper bound) unless the shapes make it impossible
  if (make_plan(BG, c, N, M, k, dilation, true, &p) != 0) return 0;
  return p.total;
}

// BN-apply producer of the queries (PrepSet::raw): coefficients derived from column sums (d.sums) or given (a, c)
struct PrepAffine { BnDerive d; const float* a; const float* c; };

template <typename T, int PT>
static void launch_prep_pt(const PrepSet& s1, const PrepSet* s2, int BG, int c, int cpad, bool norm, hipStream_t st, const PrepAffine& pa) {
  const int nbx1 = ((s1.tb_lo ? s1.rows : s1.Tn) + PT - 1) / PT;
  const int nbx2 = s2 ? ((s2->tb_lo ? s2->rows : s2->Tn) + PT - 1) / PT : 0;
  dim3 grid(nbx1 + nbx2, BG);
  const size_t lds = (size_t)c * PT * sizeof(float) + (s1.raw ? (size_t)2 * c * sizeof(float) : 0);
  const PrepSet second = s2 ? *s2 : s1;
  if (s1.raw) {                                  // BN-apply producer: fp32 tokens, one set
    if constexpr (sizeof(T) == 4) {
      if (norm) hipLaunchKernelGGL((bn_apply_knn_prep_kernel<true, PT>), grid, dim3(PT), lds, st, s1, c, cpad, pa.d, pa.a, pa.c);
      else hipLaunchKernelGGL((bn_apply_knn_prep_kernel<false, PT>), grid, dim3(PT), lds, st, s1, c, cpad, pa.d, pa.a, pa.c);
    }
    return;
  }
  if (norm) hipLaunchKernelGGL((token_prep_kernel<T, true, PT>), grid, dim3(PT), lds, st, s1, second, nbx1, c, cpad);
  else hipLaunchKernelGGL((token_prep_kernel<T, false, PT>), grid, dim3(PT), lds, st, s1, second, nbx1, c, cpad);
}

template <int TK>
static void launch_prep_coop(const PrepSet& s1, const PrepSet* s2, int BG, int c, int cpad, bool norm, hipStream_t st, const PrepAffine& pa) {
  const int nbx1 = (s1.Tn + TK - 1) / TK;
  const int nbx2 = s2 ? (s2->Tn + TK - 1) / TK : 0;
  dim3 grid(nbx1 + nbx2, BG);
  const size_t lds = (size_t)c * (TK + 1) * sizeof(float) + (s1.raw ? (size_t)2 * c * sizeof(float) : 0);
  const PrepSet second = s2 ? *s2 : s1;
  if (s1.raw) {
    if (norm) hipLaunchKernelGGL((bn_apply_knn_prep_coop_kernel<true, TK>), grid, dim3(256), lds, st, s1, c, cpad, pa.d, pa.a, pa.c);
    else hipLaunchKernelGGL((bn_apply_knn_prep_coop_kernel<false, TK>), grid, dim3(256), lds, st, s1, c, cpad, pa.d, pa.a, pa.c);
    return;
  }
  if (norm) hipLaunchKernelGGL((token_prep_coop_kernel<true, TK>), grid, dim3(256), lds, st, s1, second, nbx1, c, cpad);
  else hipLaunchKernelGGL((token_prep_coop_kernel<false, TK>), grid, dim3(256), lds, st, s1, second, nbx1, c, cpad);
}

static bool prep_coop_ok(const PrepSet& s) {       // fp32 token-major rows of whole float4s, fp32 channel-major output
  return !s.tb && s.ps.sc == 1 && ((s.ps.sn | s.ps.sg | s.ps.sb) & 3) == 0 && ((size_t)s.t & 15) == 0;
}

template <typename T>
static hipError_t launch_prep(const PrepSet& s1, const PrepSet* s2, int BG, int c, int cpad, bool norm, hipStream_t st,
                              const PrepAffine& pa = PrepAffine{}) {
  GkgProfScope prof(GKG_PROF_TOKEN_PREP, st);
  if (sizeof(T) == 4 && (c & 3) == 0 && prep_coop_ok(s1) && (!s2 || prep_coop_ok(*s2)) &&
      (size_t)(s1.Tn + (s2 ? s2->Tn : 0)) * BG <= 262144) {                          // latency-bound sizes only
    if (c <= 192) launch_prep_coop<64>(s1, s2, BG, c, cpad, norm, st, pa);               // <= 49 KB tile
    else if (c <= 384) launch_prep_coop<32>(s1, s2, BG, c, cpad, norm, st, pa);
    else launch_prep_coop<16>(s1, s2, BG, c, cpad, norm, st, pa);
    return hipGetLastError();
  }
  if (c <= 192) launch_prep_pt<T, 64>(s1, s2, BG, c, cpad, norm, st, pa);          // <= 48 KB column block
  else if (c <= 384) launch_prep_pt<T, 32>(s1, s2, BG, c, cpad, norm, st, pa);
  else launch_prep_pt<T, 16>(s1, s2, BG, c, cpad, norm, st, pa);                    // c <= 600 (plan limit)
  return hipGetLastError();
}

// The queries' BN-apply riding in their preparation pass (gkg_bn_apply_knn_prep)
struct KnnProducer { PrepAffine aff; float* raw; int raw_ld, raw_chunk; int as_keys; const float* res_tm; float* nchw; };

// Fused aggregation request of gkg_knn_mr_fwd_tm (token-major fp32 callers): outputs of knn_tile_kernel<..., MRF = true>.
struct KnnMrFuse {
  float* out;             // XM (B * N, 2 C): m into the m chunks; x into the x chunks unless x already lives there
  uint16_t* arg;          // (B, N, C)
  uint16_t* nn16;         // (B * G, N, k) or null
};

// The launch-plan part of the eligibility of the fused form (the caller checks dtype / channel alignment): the fp32 tile
// kernel with merged per-wave lists — no key splits, no prefilter, no single-wave form — and lists of at most 36 entries.
static bool knn_mr_plan_ok(const KnnPlan& p, int c, int N, int M, int k, bool pf, bool solo32) {
  return p.S == 1 && !pf && !solo32 && p.KD <= 36 && k <= 18 && M <= 65536 && (c & 3) == 0 &&
         (size_t)NW * p.KD * 64 * 8 + (size_t)k * 64 * 4 <= 150 * 1024;
}

static int knn_fwd_impl(const void* x, const void* y, const float* relpos, int64_t* nn_idx, int64_t* center,
                        int BG, int c, int N, int M, int k, int dilation, int dtype, unsigned flags,
                        void* workspace, size_t workspace_bytes, void* stream, int G_tm, const KnnMrFuse* mr = nullptr,
                        bool probe_only = false, uint16_t* nn16_only = nullptr, int ldx = 0, int xchunk = 0,
                        const KnnProducer* prod = nullptr) {
  // prod != null (gkg_bn_apply_knn_prep): PREPARATION ONLY — `x` is a projection's pre-BN output (B, N, C); the queries' pass
  // applies the BN, stores the result to prod->raw and leaves the normalised copies / norms (/ prefilter planes) in the workspace
  // exactly where the k-NN call with the same arguments + GKG_KNN_X_PREPARED expects them (same plan, same decisions: this IS that
  // call up to its preparation launch); y / relpos / mr are only tested for presence.
  const bool prep_only = prod != nullptr;
  const bool x_prepared = (flags & GKG_KNN_X_PREPARED) != 0, y_prepared = (flags & GKG_KNN_Y_PREPARED) != 0;
  if (y_prepared && (!y || prep_only || G_tm <= 0 || dtype != GKG_F32 || (c & 3)))
    return gkg_fail(GKG_ERR_SHAPE, "GKG_KNN_Y_PREPARED: a token-major fp32 k-NN call with keys");
  if (prep_only && x_prepared) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_apply_knn_prep: GKG_KNN_X_PREPARED makes no sense here");
  if ((prep_only || x_prepared) && (G_tm <= 0 || dtype != GKG_F32 || (c & 3)))
    return gkg_fail(GKG_ERR_UNSUPPORTED, "prepared queries: token-major fp32 callers with c % 4 == 0 only");
  // ldx / xchunk (token-major fp32 callers, G_tm > 0): x as a view — row pitch ldx floats (0: G_tm * c), chunk xchunk
  // (gkg_common.h "XM layout": xchunk = C / 4 with ldx = 2 C is the x half of the grouped projection's operand buffer)
  if (G_tm > 0) {
    const int C = G_tm * c;
    if (ldx == 0) ldx = C;
    if (ldx < C || xchunk < 0 || (xchunk > 0 && (dtype != GKG_F32 || (xchunk & 3) || (c & 3) || (ldx & 3) || C % xchunk || ldx < 2 * C)))
      return gkg_fail(GKG_ERR_SHAPE, "gkg_knn_fwd_tm: bad x view (ldx >= C; a chunk needs fp32, c % 4 == 0, chunk % 4 == 0, chunk | C, ldx >= 2 C)");
  } else if (ldx != 0 || xchunk != 0) {
    return gkg_fail(GKG_ERR_SHAPE, "gkg_knn_fwd: x views are a token-major feature");
  }
  // nn16_only (gkg_knn_fwd_tm16): the neighbour lists as u16 rows INSTEAD of the int64 plane (M <= 65 536)
  if (!probe_only && (!x || (!nn_idx && !mr && !nn16_only && !prep_only) || !workspace)) return gkg_fail(GKG_ERR_NULL, "gkg_knn_fwd: x, nn_idx and workspace must be non-null");
  if (dtype != GKG_F32 && dtype != GKG_BF16 && dtype != GKG_F16) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_knn_fwd: dtype must be GKG_F32, GKG_BF16 or GKG_F16");
  KnnPlan p;
  int rc = make_plan(BG, c, N, M, k, dilation, y != nullptr, &p);
  if (rc == GKG_ERR_SHAPE) return gkg_fail(rc, "gkg_knn_fwd: bad sizes (need >0, k*dilation <= M, M == N for the self graph)");
  if (rc != 0) return gkg_fail(rc, "gkg_knn_fwd: unsupported size (k*dilation <= 64, c <= 600, BG <= 65535)");
  if (!probe_only && workspace_bytes < p.total) return gkg_fail(GKG_ERR_WORKSPACE, "gkg_knn_fwd: workspace too small (see gkg_knn_workspace_bytes)");
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  float* xh = (float*)(ws + p.off_xh);
  float* yh = (float*)(ws + p.off_yh);
  float* sqx = (float*)(ws + p.off_sqx);
  float* sqy = (float*)(ws + p.off_sqy);
  const bool norm = (flags & GKG_KNN_NORMALIZE) != 0;
  hipError_t e;
  // input addressing: channel-major (BG,c,T)  or  token-major (B,T,C=G*c) when G_tm > 0
  auto strides = [&](int Tn, int ld, int chunk) {
    PrepStrides ps;
    if (G_tm > 0) { ps.G = G_tm; ps.sb = (size_t)Tn * ld; ps.sg = c; ps.sc = 1; ps.sn = (size_t)ld; ps.chunk = chunk; }
    else { ps.G = 1; ps.sb = (size_t)c * Tn; ps.sg = 0; ps.sc = Tn; ps.sn = 1; ps.chunk = 0; }
    return ps;
  };
  // bf16 contraction: the normalised copies are bf16 octet-major planes (they fit the fp32 copies' workspace slots); needs
  // c >= 9, i.e. cpad >= 16 (below that the fp32 staging area, which sizes the LDS, is smaller than the bf16 one) —
  // otherwise the flag is ignored
  const bool bf = (flags & GKG_KNN_BF16_CONTRACT) != 0 && p.cpad >= 16;
  const int cp16 = (c + 15) & ~15;
  const int cp16p = (c + 2 + 15) & ~15;            // prefilter planes: + the two |th|^2 channels (see PrepSet)
  const int Nr = (N + 31) & ~31, Mr = (M + 31) & ~31;
  // prefilter mode: the bit-exact contract path for normalised, un-split problems with lists up to 36 entries (LDS)
  const size_t pf_stage = (size_t)2 * QT * (cp16p + 8) * 2;
  // Where it pays (MI355X, tools/bench_ops.py, prefilter vs fp32 tile kernel, us per launch incl. the extra preparation
  // work): long key streams, where the contraction dominates — pvig_s@576 stage 3 (c = 200, 1296 x 1296) k*d = 18:
  // 437 + 92 vs 594 + 57, k*d = 27: 525 + 91 vs 827 + 56; stage 2 (c = 80, 5184 x 1296, k*d = 9): 675 + 82 vs 882 + 58;
  // stage 1 (c = 40, 20736 x 1296): 2203 + 143 vs 2382 + 97.  It loses on short streams, where its serial tail (merge,
  // exact pass, final ranking in one wave) is not amortised (cfg2 label graph 34 vs 25, C = 640 / k*d = 27 at 18 x 18: 210
  // vs 119; the cfg2 Grapher graph is a tie: 51 + 22 vs 54 + 15), and against the BUFFERED selection of narrow groups with
  // long lists (pvig_m: c = 24 / k*d = 18: 4469 vs 3149).  GKG_KNN_FORCE_PREFILTER overrides the rule (tests).
  const bool pf_pays = M >= 1024 && (p.KD <= 12 ? p.cpad >= 40 : (p.KD <= 27 && p.cpad >= 128));
  // The prefilter's error bound (knn_pf_kernel, step 3) is derived for |relative_pos| <= 1.125: the bias is the initial value of
  // the accumulators, so the rounding of the accumulation scales with |bias| + 2.  A call with a bias takes the prefilter only
  // when the caller vouches for that range (GKG_KNN_RELPOS_UNIT; GKGNet's bias -2 PE PE^T / D lies in [-1, 0]) — ADVICE r3.
  const bool rp_ok = !relpos || (flags & GKG_KNN_RELPOS_UNIT);
  const bool pf = !bf && norm && p.S == 1 && c >= 16 && p.KD <= 36 && !(flags & GKG_KNN_NO_PREFILTER) && rp_ok
                  && pf_stage <= 150 * 1024 && (pf_pays || (flags & GKG_KNN_FORCE_PREFILTER));
  size_t lds_q = (size_t)p.cpad * QT * sizeof(float);
  size_t lds_m = (size_t)NW * p.KD * 64 * 2 * sizeof(float);
  size_t lds = lds_q > lds_m ? lds_q : lds_m;
  const bool short_stream = p.tps < 10 * NW;
  // Selection mode.  Buffered selection (see the kernel) wins where the per-candidate insert dominates and its 32 KB
  // buffer does not cost occupancy — measured on MI355X (tools/bench_ops.py, direct -> buffered): pvig_m@768 k=18
  // stage 1 (c=12, kd=18) 15.97 -> 10.61 ms, stage 2 4.46 -> 3.13 ms, stage 3 (c=48, kd=36) 2.59 -> 1.55 ms; pvig_s@576
  // stage 1 2.52 -> 2.42 ms, stage 2 0.96 -> 0.89 ms, label graph over 20 736 keys 250 -> 214 us; it LOSES with wide groups
  // (c=200: query tile 51 KB + buffer -> one workgroup per CU, 607 -> 876 us) and on short streams with 9-entry lists
  // (cfg2 label graph 24.6 -> 31.5 us).  The GKG_KNN_SELECT_DIRECT / _BUFFERED flags override the rule (measurement / tests).
  const int force = (flags & GKG_KNN_SELECT_BUFFERED) ? 2 : ((flags & GKG_KNN_SELECT_DIRECT) ? 1 : 0);
  const bool lds_ok = lds_q + (size_t)KNN_BUF * 256 * 8 + NW * 64 * 4 <= 150 * 1024;
  const bool pays = lds_q <= 24 * 1024 && p.tps >= 2 * NW && (p.tps >= 4 * NW || p.KD >= 18 || p.S > 1);
  const bool buffered = lds_ok && (force == 2 || (force == 0 && pays));
  // fp32 contract forms: single-wave workgroups for the narrowest groups (see the launch below)
  const int nqt_ = (N + QT - 1) / QT;
  const bool solo32 = buffered && !pf && p.S == 1 && (size_t)nqt_ * BG >= 2048 && lds_q <= 4 * 1024;
  // Round 5: the buffered fp32 forms also exist with 6 k-pairs per operand batch — channels padded to a multiple of 12 where
  // that is no more than the multiple of 8 (c = 9..12, 17..24, 33..36, 57..60: pvig_m's groups of 12 and 24).  The workspace was sized
  // for the wider padding; cpad is only the row count / stride of the prepared copies.
  {
    const int c12 = (c + 11) / 12 * 12;
    if (buffered && !pf && !bf && !mr && !probe_only && p.KD <= 36 && (c12 % 16) != 0 && (c12 < p.cpad || (c12 == p.cpad && (p.cpad % 16) != 0))) {
      p.cpad = c12;
      lds_q = (size_t)p.cpad * QT * sizeof(float);
      lds = lds_q > lds_m ? lds_q : lds_m;
    }
  }
  if (mr || probe_only) {
    const bool ok = !bf && dtype == GKG_F32 && G_tm > 0 && knn_mr_plan_ok(p, c, N, M, k, pf, solo32);
    if (probe_only) return ok ? 0 : GKG_ERR_UNSUPPORTED;
    if (!ok) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_knn_mr_fwd_tm: this shape does not take the fused form (gkg_knn_mr_fused_supported)");
  }
  uint16_t* xpl = (uint16_t*)(ws + p.off_xp);
  uint16_t* ypl = (uint16_t*)(ws + p.off_yp);
  const bool pq = prod && !prod->as_keys, pk = prod && prod->as_keys;
  const PrepSet sx{x, xh, sqx, N, strides(N, ldx, xchunk), pf ? xpl : (bf ? (uint16_t*)xh : nullptr), pf ? cp16p : cp16,
                   pf ? xpl + (size_t)BG * Nr * cp16p : nullptr, Nr,
                   pq ? prod->raw : nullptr, pq ? prod->raw_ld : 0, pq ? prod->raw_chunk : 0, nullptr, nullptr};
  // (a keys producer reads its pre-BN input through `x`: the presence-only `y` is a dummy there)
  const PrepSet sy{pk ? x : y, yh, sqy, M, strides(M, G_tm * c, 0), pf ? ypl : (bf ? (uint16_t*)yh : nullptr), pf ? cp16p : cp16,
                   pf ? ypl + (size_t)BG * Mr * cp16p : nullptr, Mr,
                   pk ? prod->raw : nullptr, pk ? prod->raw_ld : 0, pk ? prod->raw_chunk : 0, pk ? prod->res_tm : nullptr,
                   pk ? prod->nchw : nullptr};
  if (prep_only) {                               // ONE operand (with its BN-apply): the other belongs to its own producer / the k-NN call
    if (bf) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_bn_apply_knn_prep: not with the bf16 contraction");
    if (pk && !y) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_apply_knn_prep: a keys producer needs a k-NN problem with keys");
    e = launch_prep<float>(pk ? sy : sx, nullptr, BG, c, p.cpad, norm, st, prod->aff);
    return e == hipSuccess ? 0 : gkg_fail_hip(e, "token_prep (BN-apply producer)");
  }
  if (x_prepared || y_prepared) {                // copies already in place (gkg_bn_apply_knn_prep): only what is missing
    if (bf) return gkg_fail(GKG_ERR_UNSUPPORTED, "GKG_KNN_X/Y_PREPARED: not with the bf16 contraction");
    if (!x_prepared) e = launch_prep<float>(sx, nullptr, BG, c, p.cpad, norm, st);
    else e = (y && !y_prepared) ? launch_prep<float>(sy, nullptr, BG, c, p.cpad, norm, st) : hipSuccess;
  }
  else if (dtype == GKG_F32) e = launch_prep<float>(sx, y ? &sy : nullptr, BG, c, p.cpad, norm, st);
  else if (dtype == GKG_F16) e = launch_prep<_Float16>(sx, y ? &sy : nullptr, BG, c, p.cpad, norm, st);
  else e = launch_prep<uint16_t>(sx, y ? &sy : nullptr, BG, c, p.cpad, norm, st);
  if (e != hipSuccess) return gkg_fail_hip(e, "token_prep");
  gkg_prof_add_work(GKG_PROF_KNN_TILE, 2.0 * BG * (double)c * N * (double)M);    // algorithmic flop of the distance contraction
  KnnArgs a;
  a.xh = xh; a.yh = yh; a.sqx = sqx; a.sqy = sqy; a.relpos = relpos;
  a.nn_idx = nn_idx; a.center = center;
  a.part_v = (float*)(ws + p.off_pv); a.part_i = (int*)(ws + p.off_pi);
  a.BG = BG; a.cpad = p.cpad; a.N = N; a.M = M; a.k = k; a.dilation = dilation; a.kd = p.kd;
  a.splits = p.S; a.tiles_per_split = p.tps;
  a.nqt = (N + QT - 1) / QT;
  a.xb = (const uint16_t*)xh; a.yb = (const uint16_t*)yh; a.cp16 = cp16;
  a.xb_lo = a.yb_lo = nullptr; a.margin = 0.f; a.wg_flags = nullptr;
  a.mr_x = a.mr_src = nullptr; a.mr_out = nullptr; a.mr_arg = nullptr; a.nn16 = nn16_only; a.mr_G = 1; a.mr_c = c;
  a.mr_ldx = a.mr_lds = c; a.mr_xchunk = a.mr_schunk = 0; a.mr_write_x = 1;
  dim3 grid((unsigned)(a.nqt * ((BG + 7) / 8) * 8), 1, p.S);
#if defined(KNN_TIMELINE) || defined(KNN_ABLATE)
  if (p.S == 1 && gkg_knn_tl_buf) a.part_v = (float*)gkg_knn_tl_buf;
#endif
  a.rp_major = 0; a.rp_group = 1;
  // bf16 form with a positional bias at least twice the size of the keys it meets (64 x 4 B of relative_pos against
  // 2 x cp16 B of a key per (query tile, key) pair, i.e. c <= 64) and enough query tiles to spread over the XCDs: map the
  // workgroups relative_pos-major (see the kernel).  Measured (tools/bench_knn_bf.py, us): pvig_m stages 1-2 6031 -> 5101,
  // 1590 -> 1487; pvig_s stage 1 1259 -> 1187.
  if (bf && relpos && !pf && a.nqt >= 64 && BG >= 8 && (size_t)QT * 4 >= (size_t)cp16 * 2 * 2) {
    a.rp_major = 1;
    grid = dim3((unsigned)(((a.nqt + 7) / 8) * 8 * BG), 1, p.S);
  }
  // Exact forms (fp32 tile kernel, prefilter kernel) with a positional bias that is far beyond an L2 (>= 8 MB: GKGNet-576
  // stage 1: 107 MB, stage 2: 27 MB; pvig_m stage 1: 340 MB): problem-major order streams the WHOLE bias once per problem
  // (counters, pvig_s stage 1: 7.5 GB of L2 misses per launch at B*G = 64 — the launch ran at the speed of that stream).
  // Interleave the XCD's problems in groups whose key sets stay in the L2 (<= 2 MB): the bias is fetched once per group.
  if (relpos && a.rp_major == 0 && p.S == 1 && BG >= 16 && (size_t)N * M * 4 >= ((size_t)8 << 20)) {
    const size_t key_bytes = (size_t)M * (pf ? (size_t)cp16p * 4 : (size_t)p.cpad * 4);
    const int bpx = (BG + 7) / 8;
    int g = (int)(((size_t)2 << 20) / (key_bytes ? key_bytes : 1));
    g = g > bpx ? bpx : g;
    while (g >= 2 && bpx % g) --g;               // a divisor of the XCD's problem count: the grid (and the flag array) keep their size
    if (g >= 2) { a.rp_major = 2; a.rp_group = g; }
  }
  if (pf) {
    a.xb = xpl; a.xb_lo = xpl + (size_t)BG * Nr * cp16p;
    a.yb = y ? ypl : xpl; a.yb_lo = y ? ypl + (size_t)BG * Mr * cp16p : a.xb_lo;
    a.cp16 = cp16p; a.pf_c = c; a.pf_nrows = Nr; a.pf_mrows = y ? Mr : Nr;
    // eps = split terms dropped (3 * 2^-18 * 2 = 2.3e-5) + |y|^2 entering as its two leading bf16 terms (2^-17 |y|^2 <= 7.7e-6)
    // + fp32 accumulation of the 3c + 2 products and the bias on partial sums <= |bias| + 2 + |y|^2 <= 4.125 (12.4c * 2^-24) +
    // the contract's own chain (2c * 2^-24): 14.4 c u = 8.6e-7 c; margin = 2 eps
    a.margin = 2.0f * (4.0e-5f + 9.0e-7f * (float)cp16p);
    a.wg_flags = (int*)(ws + p.off_flags);
    e = hipMemsetAsync(a.wg_flags, 0, sizeof(int) * (size_t)grid.x, st);
    if (e != hipSuccess) return gkg_fail_hip(e, "knn_pf_kernel (flags)");
    e = launch_knn_prefilter(a, grid, p.KD, st);
    if (e != hipSuccess) return gkg_fail_hip(e, "knn_pf_kernel");
    // fall through: the fp32 tile kernel below runs as the clean-up pass over the tiles the prefilter flagged
    // (a.wg_flags != null: every other workgroup exits at once)
  }
  // enough query tiles to fill the chip with single-wave workgroups (2+ waves per SIMD) and the keys not split: one list per query
  // (measured, bf16 form, 4 waves -> 1: pvig_s stage 1 1353 -> 1257 us, pvig_m stages 1-3 6661 -> 6051, 1697 -> 1592,
  // 801 -> 731; it loses when the per-wave query image costs occupancy: stage 2, c = 80, 458 -> 540)
  const bool solo = p.S == 1 && (size_t)a.nqt * BG >= 2048 && (size_t)QT * (cp16 + 8) * 2 <= 8 * 1024;
  if (bf) {
    // bf16 form: the staged query tile is 64 x (cp16 + 8) bf16 — buffered selection with the largest candidate buffer (16 or
    // 12 entries per lane) that keeps the workgroups per CU the register budget allows
    const size_t q_bf = (size_t)QT * (cp16 + 8) * 2;
    const int wgs = relpos ? (p.KD <= 18 ? 3 : 2) : (p.KD <= 12 ? 4 : (p.KD <= 27 ? 3 : 2));
    const size_t per_wg = (size_t)160 * 1024 / wgs;
    int wbuf = q_bf + 16 * 2048 + 1024 <= per_wg ? 16 : (q_bf + 12 * 2048 + 1024 <= per_wg ? 12 : 0);
    const bool pays_bf = p.tps >= 2 * NW && (p.tps >= 4 * NW || p.KD >= 18 || p.S > 1);
    if (force == 1 || (force == 0 && !pays_bf)) wbuf = 0;
    if (force == 2 && wbuf == 0) wbuf = 12;
    e = launch_knn_tile_bf(a, grid, lds, p.KD, wbuf, solo && wbuf > 0, st);
  } else {
    // fp32 contract forms: direct (guarded / guard-less for short streams) or buffered selection — gkg_knn_f32.hip
    // single-wave workgroups for the narrowest groups (pvig_m stage 1: c = 12, 36 864 queries x 2 304 keys, k*d = 18): one list
    // per query instead of four quarter-stream lists — measured on the model's activations 9 045 -> 7 872 us per launch; no
    // gain at c = 24 / 48 (2 748 -> 2 761, 1 225 -> 1 238 us), where the per-wave query image costs occupancy
    if (mr) {
      a.mr_x = (const float*)x; a.mr_src = (const float*)(y ? y : x);
      a.mr_out = mr->out; a.mr_arg = mr->arg; a.nn16 = mr->nn16; a.mr_G = G_tm; a.mr_c = c;
      a.mr_ldx = ldx; a.mr_xchunk = xchunk;
      a.mr_lds = y ? G_tm * c : ldx; a.mr_schunk = y ? 0 : xchunk;
      // x already lives in the operand buffer when the caller passes that buffer's x half as the view of x
      a.mr_write_x = !((const void*)x == (const void*)mr->out && ldx == 2 * G_tm * c && xchunk == (G_tm * c) / 4);
      e = launch_knn_tile_f32_mr(a, grid, lds, p.KD, buffered ? 2 : ((short_stream && p.KD == 9) ? 1 : 0), st);
    } else {
      e = launch_knn_tile_f32(a, grid, lds, p.KD, solo32 ? 5 : (buffered ? 2 : ((short_stream && p.KD == 9) ? 1 : 0)), st);
    }
  }
  if (e != hipSuccess) return gkg_fail_hip(e, "knn_tile_kernel");
  if (p.S > 1) {
    const size_t ne = (size_t)BG * N * p.kd * p.S;
    GkgProfScope prof(GKG_PROF_KNN_MERGE, st);
    // ranks that no finite candidate claims (non-finite inputs only) must still hold a valid index
    if (nn16_only) (void)hipMemsetAsync(nn16_only, 0, sizeof(uint16_t) * (size_t)BG * N * k, st);
    else (void)hipMemsetAsync(nn_idx, 0, sizeof(int64_t) * (size_t)BG * N * k, st);
    hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, a.part_v, a.part_i,
                       nn_idx, center, nn16_only, p.S, BG, N, M, k, dilation, p.kd);
    e = hipGetLastError();
    if (e != hipSuccess) return gkg_fail_hip(e, "knn_merge_kernel");
  }
  return 0;
}

extern "C" int gkg_knn_fwd(const void* x, const void* y, const float* relpos, int64_t* nn_idx, int64_t* center,
                           int BG, int c, int N, int M, int k, int dilation, int dtype, unsigned flags,
                           void* workspace, size_t workspace_bytes, void* stream) {
  return knn_fwd_impl(x, y, relpos, nn_idx, center, BG, c, N, M, k, dilation, dtype, flags, workspace, workspace_bytes,
                      stream, 0);
}

extern "C" int gkg_knn_fwd_tm(const void* x, int ldx, int xchunk, const void* y, const float* relpos, int64_t* nn_idx, int64_t* center,
                              int B, int G, int c, int N, int M, int k, int dilation, int dtype, unsigned flags,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (B <= 0 || G <= 0) return gkg_fail(GKG_ERR_SHAPE, "gkg_knn_fwd_tm: bad B / G");
  return knn_fwd_impl(x, y, relpos, nn_idx, center, B * G, c, N, M, k, dilation, dtype, flags, workspace, workspace_bytes,
                      stream, G, nullptr, false, nullptr, ldx, xchunk);
}

// The same graph as gkg_knn_fwd_tm with COMPACT neighbour lists: nn16 (B * G, N, k) u16 rows instead of the (2, B G, N, k) int64
// edge_index — for callers that consume the graph on the device and never hand it out (Grapher.forward discards it, reference
// torch_vertex.py:330).  At GKGNet-576's stage 1 the int64 pair is 191 MB written per launch, the compact lists 24 MB (pvig_m
// stage 1, k = 18: 1.36 GB / 170 MB).  Needs M <= 65 536.  Same bits as the int64 lists (tests/test_hip_knn_compact.py).
extern "C" int gkg_knn_fwd_tm16(const void* x, int ldx, int xchunk, const void* y, const float* relpos, uint16_t* nn16, int B, int G, int c, int N, int M,
                                int k, int dilation, int dtype, unsigned flags, void* workspace, size_t workspace_bytes, void* stream) {
  if (B <= 0 || G <= 0) return gkg_fail(GKG_ERR_SHAPE, "gkg_knn_fwd_tm16: bad B / G");
  if (!nn16) return gkg_fail(GKG_ERR_NULL, "gkg_knn_fwd_tm16: nn16 must be non-null");
  if (M > 65536) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_knn_fwd_tm16: M <= 65536 (u16 rows)");
  return knn_fwd_impl(x, y, relpos, nullptr, nullptr, B * G, c, N, M, k, dilation, dtype, flags, workspace, workspace_bytes,
                      stream, G, nullptr, false, nn16, ldx, xchunk);
}

// Row g2: k-NN + max-relative aggregation in one kernel (knn_tile_kernel<..., MRF = true>) for token-major fp32 callers.
// x (B, N, C = G c), y (B, M, C) or NULL (self graph), relative_pos (N, M) or NULL -> the graph of gkg_knn_fwd_tm (same
// contract, same bits) is built and consumed on the spot: xm_out (B N, 2 C) = the grouped projection's operand buffer of
// gkg_mr_fwd_tm mode 1 (m into the m chunks; the x chunks are written unless x IS that buffer's x half: x == xm_out,
// ldx == 2 C, xchunk == C / 4), arg_out (B, N, C) u16 = the winning neighbour rows of gkg_mr_fwd_tm arg_kind 1,
// nn16_out (B G, N, k) u16 = the neighbour lists (optional); nn_idx_out / center_out (B G, N, k) int64 = gkg_knn_fwd_tm's outputs
// for callers that return the graph (GrapherLabel), optional — otherwise no int64 index tensor and no centre plane exist.
extern "C" int gkg_knn_mr_fused_supported(int B, int G, int c, int N, int M, int k, int dilation, int has_y, int has_relpos,
                                          unsigned flags) {
  if (B <= 0 || G <= 0 || c <= 0 || ((G * c) & 15)) return 0;
  static const float dummy = 0.f;
  return knn_fwd_impl(nullptr, has_y ? &dummy : nullptr, has_relpos ? &dummy : nullptr, nullptr, nullptr, B * G, c, N, M, k,
                      dilation, GKG_F32, flags, nullptr, 0, nullptr, G, nullptr, true) == 0 ? 1 : 0;
}

extern "C" int gkg_knn_mr_fwd_tm(const float* x, int ldx, int xchunk, const float* y, const float* relpos, float* xm_out,
                                 uint16_t* arg_out, uint16_t* nn16_out, int64_t* nn_idx_out, int64_t* center_out, int B, int G, int c,
                                 int N, int M, int k, int dilation, unsigned flags, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  if (B <= 0 || G <= 0) return gkg_fail(GKG_ERR_SHAPE, "gkg_knn_mr_fwd_tm: bad B / G");
  if (!xm_out || !arg_out) return gkg_fail(GKG_ERR_NULL, "gkg_knn_mr_fwd_tm: null output");
  if ((G * c) & 15) return gkg_fail(GKG_ERR_SHAPE, "gkg_knn_mr_fwd_tm: C = G * c must be a multiple of 16");
  if (((size_t)x & 15) || ((size_t)y & 15) || ((size_t)xm_out & 15) || ((size_t)arg_out & 7) || (ldx & 3))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_knn_mr_fwd_tm: 16-byte aligned rows");
  if ((const void*)x == (const void*)xm_out && !(ldx == 2 * G * c && xchunk == (G * c) / 4))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_knn_mr_fwd_tm: x aliases xm_out but is not its x half (ldx = 2 C, xchunk = C / 4)");
  const KnnMrFuse mr{xm_out, arg_out, nn16_out};
  if (center_out && !nn_idx_out) return gkg_fail(GKG_ERR_NULL, "gkg_knn_mr_fwd_tm: center_out without nn_idx_out");
  return knn_fwd_impl(x, y, relpos, nn_idx_out, center_out, B * G, c, N, M, k, dilation, GKG_F32, flags, workspace, workspace_bytes,
                      stream, G, &mr, false, nullptr, ldx, xchunk);
}


// The Grapher's fc1 BN-apply (train mode, statistics from the projection's fp64 column sums: gkg_bn_apply_train's contract with
// nb == 1, act == 0, no residual) and the k-NN's token preparation of the SAME tokens in one pass (round 6):
//   y (B N, C) pre-BN projection output -> x = a y + c, written to `out` (row pitch ldo floats; ochunk > 0: the x half of the
//   grouped projection's operand buffer, include/gkg_hip.h "XM layout") AND normalised into `knn_workspace` — the workspace of the
//   k-NN call (gkg_knn_fwd_tm / _tm16 / gkg_knn_mr_fwd_tm) with the SAME (B, G, c, N, M, k, dilation, y / relative_pos presence,
//   flags) that follows with GKG_KNN_X_PREPARED set and x = `out`: it then launches no preparation for the queries (none at all
//   for a self graph).  fused_mr != 0: that call is gkg_knn_mr_fwd_tm.  Same arithmetic, same bits as apply + token_prep.
extern "C" int gkg_bn_apply_knn_prep(const float* y, const double* sums, const float* gamma, const float* beta, const float* bias,
                                     float* running_mean, float* running_var, long long* num_batches_tracked, float* a, float* c_out,
                                     float* mean, float* invstd, float* out, int ldo, int ochunk, int B, int G, int c, int N, int M,
                                     int k, int dilation, int has_y, int has_relpos, unsigned knn_flags, int fused_mr,
                                     int as_keys, const float* res_tm, float* out_nchw,
                                     void* knn_workspace, size_t knn_workspace_bytes, float momentum, float eps, double* zero_buf,
                                     size_t zero_doubles, void* stream) {
  if (!y || !sums || !gamma || !beta || !a || !c_out || !mean || !invstd || !out || !knn_workspace)
    return gkg_fail(GKG_ERR_NULL, "gkg_bn_apply_knn_prep: null pointer");
  if (!as_keys && (res_tm || out_nchw)) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_apply_knn_prep: res_tm / out_nchw belong to a keys producer");
  if (as_keys && (!has_y || ochunk != 0 || ((size_t)res_tm & 15))) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_apply_knn_prep: a keys producer writes plain rows for a problem with keys");
  if ((running_mean == nullptr) != (running_var == nullptr)) return gkg_fail(GKG_ERR_NULL, "gkg_bn_apply_knn_prep: running stats come in pairs");
  if (B <= 0 || G <= 0 || c <= 0 || (c & 3) || N <= 0) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_apply_knn_prep: bad sizes (c % 4 == 0)");
  const int C = G * c;
  if (ldo == 0) ldo = C;
  if (ldo < C || (ldo & 3) || ochunk < 0 || (ochunk & 3) || (ochunk > 0 && (C % ochunk || ldo < 2 * C)) || ((size_t)out & 15) || ((size_t)y & 15) ||
      (zero_doubles && !zero_buf))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_apply_knn_prep: bad output view");
  KnnProducer prod{};
  prod.aff.d = BnDerive{sums, gamma, beta, bias, running_mean, running_var, num_batches_tracked, a, c_out, mean, invstd,
                        B * (as_keys ? M : N), momentum, eps, zero_buf, zero_doubles};
  prod.raw = out; prod.raw_ld = ldo; prod.raw_chunk = ochunk;
  prod.as_keys = as_keys ? 1 : 0; prod.res_tm = res_tm; prod.nchw = out_nchw;
  static const float dummy = 0.f;
  static KnnMrFuse mr_dummy{nullptr, nullptr, nullptr};
  return knn_fwd_impl(y, has_y ? &dummy : nullptr, has_relpos ? &dummy : nullptr, nullptr, nullptr, B * G, c, N, M, k, dilation, GKG_F32,
                      knn_flags, knn_workspace, knn_workspace_bytes, stream, G, fused_mr ? &mr_dummy : nullptr, false, nullptr, 0, 0, &prod);
}
