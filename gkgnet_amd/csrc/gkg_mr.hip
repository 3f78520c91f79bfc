// gkg_mr.hip — max-relative neighbour aggregation and its backward scatter (gfx950).
//
// Forward replaces batched_index_select(x, edge_index[1]) + batched_index_select(src, edge_index[0]) +
// torch.max(x_j - x_i, -1)  (reference torch_vertex.py:49-54, torch_nn.py:84-105) without building the
// (BG,c,N,k) gathered tensors.  Backward replaces autograd's max_backward -> sub -> index_put_(accumulate).
//
// Layout: features are channel-major (BG,c,T); one (bg,ch) row of the source is contiguous and small
// (T*4 B <= 83 KB), so a workgroup keeps CH source rows in LDS and gathers / scatters there:
//   fwd : grid (n-tiles, channel chunks, BG); thread = one query n, its k indices in LDS (int32);
//         per channel: k LDS gathers, k subtracts (x_j - x_i as the reference does), first-max select;
//         writes m and the argmax byte coalesced along n.
//   bwd : grid (channel chunks, BG); LDS accumulator acc[CH][M]; all N queries of the (bg, chunk) are
//         swept, g is added at idx[n][argmax] with ds_add_f32, then the rows are stored coalesced
//         (gsrc / gx are fully overwritten -> no memset, no global atomics).
#include <type_traits>

#include "gkg_common.h"

// measurement builds only (tools/ubench/mr_fwd_ablate.py).  MR_ABL bits: 1 gathers from 4 fixed rows, 2 synthetic indices (no
// index loads), 4 no out stores, 8 no argmax stores.  MR_TL: wave-level s_memtime stamps of every 1009th workgroup, written
// behind the argmax tensor (the tool allocates the room).
#ifndef MR_ABL
#define MR_ABL 0
#endif
#ifdef MR_TL
#define MR_STAMP(p)                                                                                                     \
  do {                                                                                                                  \
    if ((threadIdx.x & 63) == 0 && blockIdx.x % 1009 == 0 && blockIdx.x / 1009 < 512)                                   \
      reinterpret_cast<unsigned long long*>(argmax + 2 * T * C)[((blockIdx.x / 1009) * 4 + (threadIdx.x >> 6)) * 8 + (p)] = \
          __builtin_readcyclecounter();                                                                                 \
  } while (0)
#else
#define MR_STAMP(p) do {} while (0)
#endif

namespace gkg {

constexpr int MR_LDS_BUDGET = 96 * 1024;   // bytes of LDS for source rows / accumulators

// Caller-provided neighbour indices are trusted to be in [0, M); clamping costs two integer ops per index and keeps a
// corrupted index tensor from becoming an out-of-bounds access.
__device__ __forceinline__ int clamp_idx(int64_t v, int M) { return (int)(v < 0 ? 0 : (v >= M ? M - 1 : v)); }

// torch.max semantics for the running maximum (reference torch_vertex.py:54): a NaN candidate replaces a non-NaN best and
// then stays (first NaN wins, like the first-max tie rule), so a diverging run surfaces NaN instead of hiding it.
__device__ __forceinline__ bool takes(float v, float best) { return v > best || (v != v && best == best); }

// ------------------------------------------------------------------------------------------ forward
// These kernels move ~1 flop per byte and, at the sizes of this path, every operand is L2/MALL resident:
// they are bound by memory *latency*, so each thread issues all of its independent loads before the
// first use (index row, centre values, then k gathers per channel from the LDS-staged source rows).
constexpr int MR_CB = 4;        // channels processed together per thread (loads batched across them)

template <typename T>
__global__ __launch_bounds__(256) void mr_fwd_kernel(const T* __restrict__ x, const T* __restrict__ src,
                                                     const int64_t* __restrict__ nn_idx, T* __restrict__ m_out,
                                                     uint8_t* __restrict__ argmax, int c, int N, int M, int k, int CH) {
  extern __shared__ float smem[];
  float* rows = smem;                                   // [CH][M]
  int* idx_s = reinterpret_cast<int*>(smem + (size_t)CH * M);   // [k][256]
  const int tid = threadIdx.x;
  const int bg = blockIdx.z;
  const int ch0 = blockIdx.y * CH;
  const int nch = min(CH, c - ch0);
  const int n0 = blockIdx.x * 256;
  const int n = n0 + tid;

  {
    const T* sp = src + ((size_t)bg * c + ch0) * M;
    const int total = nch * M;
    for (int i = tid; i < total; i += 1024) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (i + 256 * u < total) ? ldf(sp + i + 256 * u) : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) if (i + 256 * u < total) rows[i + 256 * u] = v[u];
    }
    // indices of this tile: (256, k) int64 contiguous -> idx_s[j][q]
    const int64_t* ip = nn_idx + ((size_t)bg * N + n0) * k;
    const int cnt = min(256, N - n0) * k;
    for (int i = tid; i < cnt; i += 1024) {
      int64_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (i + 256 * u < cnt) ? ip[i + 256 * u] : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = i + 256 * u;
        if (e < cnt) { const int q = e / k; idx_s[(e - q * k) * 256 + q] = clamp_idx(v[u], M); }
      }
    }
  }
  __syncthreads();
  if (n >= N) return;
  for (int cb = 0; cb < nch; cb += MR_CB) {
    float xi[MR_CB], best[MR_CB];
    int arg[MR_CB];
#pragma unroll
    for (int u = 0; u < MR_CB; ++u) {
      const int ch = min(cb + u, nch - 1);
      xi[u] = ldf(x + ((size_t)bg * c + ch0 + ch) * N + n);
    }
#pragma unroll
    for (int u = 0; u < MR_CB; ++u) {
      const float* r = rows + (size_t)min(cb + u, nch - 1) * M;
      best[u] = r[idx_s[tid]] - xi[u];
      arg[u] = 0;
    }
    for (int j = 1; j < k; ++j) {
      const int id = idx_s[j * 256 + tid];
#pragma unroll
      for (int u = 0; u < MR_CB; ++u) {
        const float v = rows[(size_t)min(cb + u, nch - 1) * M + id] - xi[u];
        if (takes(v, best[u])) { best[u] = v; arg[u] = j; }
      }
    }
#pragma unroll
    for (int u = 0; u < MR_CB; ++u) {
      if (cb + u < nch) {
        const size_t o = ((size_t)bg * c + ch0 + cb + u) * N + n;
        stf(m_out + o, best[u]);
        if (argmax) argmax[o] = (uint8_t)arg[u];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ backward
template <typename T, bool SELF>
__global__ __launch_bounds__(256) void mr_bwd_kernel(const T* __restrict__ g, const int64_t* __restrict__ nn_idx,
                                                     const uint8_t* __restrict__ argmax, T* __restrict__ gx,
                                                     T* __restrict__ gsrc, int c, int N, int M, int k, int CH) {
  extern __shared__ float smem[];
  float* acc = smem;                                    // [CH][M]
  int* idx_s = reinterpret_cast<int*>(smem + (size_t)CH * M);   // [256][k] as [q*k + j]
  const int tid = threadIdx.x;
  const int bg = blockIdx.y;
  const int ch0 = blockIdx.x * CH;
  const int nch = min(CH, c - ch0);
  const size_t gbase = ((size_t)bg * c + ch0) * N;

  {
    const int total = nch * N;                          // centre term: gx = -g (self: seeds the accumulator)
    for (int i = tid; i < total; i += 1024) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (i + 256 * u < total) ? ldf(g + gbase + i + 256 * u) : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = i + 256 * u;
        if (e < total) { if (SELF) acc[e] = -v[u]; else stf(gx + gbase + e, -v[u]); }
      }
    }
    if (!SELF) for (int i = tid; i < nch * M; i += 256) acc[i] = 0.0f;
  }
  for (int n0 = 0; n0 < N; n0 += 256) {
    __syncthreads();                                    // acc init done / previous tile's idx consumed
    const int64_t* ip = nn_idx + ((size_t)bg * N + n0) * k;
    const int cnt = min(256, N - n0) * k;
    for (int i = tid; i < cnt; i += 1024) {
      int64_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (i + 256 * u < cnt) ? ip[i + 256 * u] : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) if (i + 256 * u < cnt) idx_s[i + 256 * u] = clamp_idx(v[u], M);
    }
    __syncthreads();
    const int n = n0 + tid;
    if (n < N) {
      for (int cb = 0; cb < nch; cb += MR_CB) {
        float gv[MR_CB];
        int am[MR_CB];
#pragma unroll
        for (int u = 0; u < MR_CB; ++u) {
          const size_t o = gbase + (size_t)min(cb + u, nch - 1) * N + n;
          gv[u] = ldf(g + o);
          am[u] = argmax[o];
        }
#pragma unroll
        for (int u = 0; u < MR_CB; ++u)
          if (cb + u < nch) atomicAdd(&acc[(size_t)(cb + u) * M + idx_s[tid * k + min(am[u], k - 1)]], gv[u]);
      }
    }
  }
  __syncthreads();
  T* out = SELF ? gx : gsrc;
  const size_t obase = ((size_t)bg * c + ch0) * M;
  for (int i = tid; i < nch * M; i += 256) stf(out + obase + i, acc[i]);
}

// ------------------------------------------------------------------------------------------ large-M fallbacks
// When one source row (M*4 B) does not fit the LDS budget (label graph over >24k image tokens) the rows stay in
// L2: the forward gathers straight from global memory, the backward scatters with fp32 global atomics.
template <typename T>
__global__ __launch_bounds__(256) void mr_fwd_gather_kernel(const T* __restrict__ x, const T* __restrict__ src,
                                                            const int64_t* __restrict__ nn_idx, T* __restrict__ m_out,
                                                            uint8_t* __restrict__ argmax, int c, int N, int M, int k) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int bg = blockIdx.z;
  if (n >= N) return;
  const int64_t* ip = nn_idx + ((size_t)bg * N + n) * k;
  for (int ch = blockIdx.y; ch < c; ch += gridDim.y) {
    const size_t o = ((size_t)bg * c + ch) * N + n;
    const T* r = src + ((size_t)bg * c + ch) * M;
    const float xi = ldf(x + o);
    float best = ldf(r + clamp_idx(ip[0], M)) - xi;
    int arg = 0;
    for (int j = 1; j < k; ++j) {
      const float v = ldf(r + clamp_idx(ip[j], M)) - xi;
      if (takes(v, best)) { best = v; arg = j; }
    }
    stf(m_out + o, best);
    if (argmax) argmax[o] = (uint8_t)arg;
  }
}

__global__ __launch_bounds__(256) void mr_bwd_atomic_kernel(const float* __restrict__ g, const int64_t* __restrict__ nn_idx,
                                                            const uint8_t* __restrict__ argmax, float* __restrict__ gx,
                                                            float* __restrict__ dst, int c, int N, int M, int k, int self) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int bg = blockIdx.z;
  if (n >= N) return;
  const int64_t* ip = nn_idx + ((size_t)bg * N + n) * k;
  for (int ch = blockIdx.y; ch < c; ch += gridDim.y) {
    const size_t o = ((size_t)bg * c + ch) * N + n;
    const float gv = g[o];
    if (!self) gx[o] = -gv;
    atomicAdd(dst + ((size_t)bg * c + ch) * M + clamp_idx(ip[min((int)argmax[o], k - 1)], M), gv);
  }
}

__global__ __launch_bounds__(256) void negate_kernel(const float* __restrict__ g, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = -g[i];
}

// ------------------------------------------------------------------------------------------ token-major variants
// Inside the fused Grapher block activations are token-major (B, N, C): a neighbour is then ONE contiguous row
// segment of c floats, so the aggregation is a coalesced row gather — no LDS staging.  One thread = one token x
// 4 consecutive channels (one group): k float4 gathers + centre float4, all issued before the max chain.
//   mode 0: m (B,N,C) token-major.
//   mode 1: the grouped 1x1 projection's operand buffer XM (T, 2C) (gkg_common.h "XM layout"): the reference's interleave
//           [x_0, m_0, x_1, m_1, ...] (torch_vertex.py:57-61) + Conv2d(groups=4) channel split (torch_nn.py:61) with the
//           columns of each conv group reordered to [x chunk | m chunk] — a permutation of the weight's input columns that the
//           weight-plane build folds in, so nothing is interleaved at run time.  m goes to the m chunks; the x chunks are
//           written only when `write_x` (x does not live in the buffer already: the producer of x writes it there itself).
// x / src are token-major VIEWS (pointer, row pitch, chunk): chunk 0 = plain rows, chunk h = the x half of an XM buffer.
// AK — what `argmax` holds for the backward: 0 = the winning SLOT j (u8, the C-ABI's documented form), 1 = the winning
// neighbour's ROW INDEX itself (u16; needs M <= 65536).  With the row stored the backward scatters straight from it: no
// index-row lookup, i.e. one dependent global round trip less per query and no 72-byte index rows to fetch.
__device__ __forceinline__ int clamp_idx(uint16_t v, int M) { return min((int)v, M - 1); }     // compact lists (gkg_mr_fwd_tm16)

template <int KS, int QP, typename OutT = float, int AK = 0, typename IdxT = int64_t>   // KS: compile-time k (all k index loads and row gathers issued up front);
                            // QP: channel quads per thread (share one index row; QP*4 channels stay inside a group)
                            // OutT: element type of `out` (float, or uint16_t = bf16 for the grouped GEMM's operand)
__global__ __launch_bounds__(256) void mr_fwd_tm_kernel(const float* __restrict__ x, const float* __restrict__ src,
                                                        const IdxT* __restrict__ nn_idx, OutT* __restrict__ out,
                                                        uint8_t* __restrict__ argmax, int B, int G, int c, int N, int M,
                                                        int k_rt, int mode, int ldx, int xchunk, int lds_, int schunk,
                                                        int write_x) {
  const int k = KS > 0 ? KS : k_rt;
  const int C = G * c, CT = C / (4 * QP);                         // thread-columns per token
  const size_t T = (size_t)B * N;
  // XCD-aware map: a token's k neighbours are rows of ITS image, so all workgroups of one image get linear ids that are
  // congruent mod 8 — one XCD, one L2 — and adjacent in dispatch order: the image's rows (N*C*4 bytes: 415 KB at cfg2)
  // are fetched into that L2 once and every gather after the first hits it.  With the plain linear map the k-fold
  // re-reads were spread over all 8 L2s (measured FETCH_SIZE 40 MB per launch against 13 MB of source rows).
  const int bpi = (N * CT + 255) / 256;                          // workgroups per image
  const int lin = blockIdx.x;
  const int xcd = lin & 7, seq = lin >> 3;
  const int b = (seq / bpi) * 8 + xcd;
  if (b >= B) return;                                            // grid padded to a multiple of 8 images
  const int jt = (seq - (seq / bpi) * bpi) * 256 + (int)threadIdx.x;
  if (jt >= N * CT) return;
  const int n = jt / CT;
  const int ch = 4 * QP * (jt - n * CT);
  const size_t t = (size_t)b * N + n;
  const int g = ch / c;
  const IdxT* ip = nn_idx + (((size_t)b * G + g) * N + n) * k;
  const float* sb = src + (size_t)b * M * lds_ + xm_col(ch, schunk);
  const float* xb = x + t * (size_t)ldx + xm_col(ch, xchunk);
  float4 xi[QP], best[QP];
  int ai[QP][4];                                                  // winner per channel: neighbour slot j (AK == 0) / row index (AK == 1)
  MR_STAMP(0);
#pragma unroll
  for (int q = 0; q < QP; ++q) xi[q] = *reinterpret_cast<const float4*>(xb + 4 * q);
  // `takes` (first maximum wins, a NaN is the maximum and sticks) costs 3 compares + 2 scalar mask operations + exec-masked
  // moves per (channel, neighbour): 420 vector + 330 scalar instructions per thread (SQ counters, profiles/r04_pmc_mr_stage1.txt).
  // Fast chain: a plain '>' (identical to `takes` whenever no difference is NaN) while chk accumulates d * 0 — NaN as soon as
  // any difference is NaN or infinite; a wave in which some lane's chk is NaN (non-finite inputs only) redoes its chain with
  // `takes`.  (What bounds the kernel is neither this nor HBM nor the L2 row gathers nor the store pattern — each was
  // removed in turn, EXPERIMENTS.md round 4 — but the in-order vector-memory pipe: a wave's stores are acknowledged 4 k cycles
  // after issue and its first loads return after 8 k under load.)
  auto chain = [&](auto careful, int q, const float4* v, const int* id, int kk) {
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    for (int j = 0; j < kk; ++j) {
      const float d0 = v[j].x - xi[q].x, d1 = v[j].y - xi[q].y, d2 = v[j].z - xi[q].z, d3 = v[j].w - xi[q].w;
      const int w = AK == 1 ? id[j] : j;
      if (j == 0) { best[q] = make_float4(d0, d1, d2, d3); ai[q][0] = ai[q][1] = ai[q][2] = ai[q][3] = w; }
      else if constexpr (decltype(careful)::value) {
        if (takes(d0, best[q].x)) { best[q].x = d0; ai[q][0] = w; }
        if (takes(d1, best[q].y)) { best[q].y = d1; ai[q][1] = w; }
        if (takes(d2, best[q].z)) { best[q].z = d2; ai[q][2] = w; }
        if (takes(d3, best[q].w)) { best[q].w = d3; ai[q][3] = w; }
      } else {
        const bool t0 = d0 > best[q].x, t1 = d1 > best[q].y, t2 = d2 > best[q].z, t3 = d3 > best[q].w;
        best[q].x = t0 ? d0 : best[q].x; ai[q][0] = t0 ? w : ai[q][0];
        best[q].y = t1 ? d1 : best[q].y; ai[q][1] = t1 ? w : ai[q][1];
        best[q].z = t2 ? d2 : best[q].z; ai[q][2] = t2 ? w : ai[q][2];
        best[q].w = t3 ? d3 : best[q].w; ai[q][3] = t3 ? w : ai[q][3];
      }
      if constexpr (!decltype(careful)::value) {
        c0 = __builtin_fmaf(d0, 0.f, c0); c1 = __builtin_fmaf(d1, 0.f, c1);
        c2 = __builtin_fmaf(d2, 0.f, c2); c3 = __builtin_fmaf(d3, 0.f, c3);
      }
    }
    return (c0 + c1) + (c2 + c3);
  };
  if (KS > 0) {
    int id[KS > 0 ? KS : 1];
#pragma unroll
#if MR_ABL & 2
    for (int j = 0; j < KS; ++j) id[j] = (int)(((unsigned)n * 7u + (unsigned)j * 131u) % (unsigned)M);
#else
    for (int j = 0; j < KS; ++j) id[j] = clamp_idx(ip[j], M);
#endif
#pragma unroll
    for (int q = 0; q < QP; ++q) {
      float4 v[KS > 0 ? KS : 1];
#pragma unroll
#if MR_ABL & 1
      for (int j = 0; j < KS; ++j) v[j] = *reinterpret_cast<const float4*>(sb + (size_t)(id[j] & 3) * lds_ + 4 * q);
#else
      for (int j = 0; j < KS; ++j) v[j] = *reinterpret_cast<const float4*>(sb + (size_t)id[j] * lds_ + 4 * q);
#endif
#ifdef MR_TL
      if (v[0].x == 1.25e-33f || id[8] == -7) MR_STAMP(7);      // (forces the index loads to have arrived)
      MR_STAMP(1);
      if (v[8].y == 1.25e-33f && v[3].z == 1.5e-33f) MR_STAMP(7);
      MR_STAMP(2);
#endif
      const float chk = chain(std::false_type{}, q, v, id, KS);
      if (__builtin_amdgcn_ballot_w64(chk != chk) != 0ull) (void)chain(std::true_type{}, q, v, id, KS);
    }
  } else {
    for (int j = 0; j < k; ++j) {
      const int rid = clamp_idx(ip[j], M);
      const size_t row = (size_t)rid * lds_;
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(sb + row + 4 * q);
        const float d0 = v.x - xi[q].x, d1 = v.y - xi[q].y, d2 = v.z - xi[q].z, d3 = v.w - xi[q].w;
        const int w = AK == 1 ? rid : j;
        if (j == 0) { best[q] = make_float4(d0, d1, d2, d3); ai[q][0] = ai[q][1] = ai[q][2] = ai[q][3] = w; continue; }
        if (takes(d0, best[q].x)) { best[q].x = d0; ai[q][0] = w; }
        if (takes(d1, best[q].y)) { best[q].y = d1; ai[q][1] = w; }
        if (takes(d2, best[q].z)) { best[q].z = d2; ai[q][2] = w; }
        if (takes(d3, best[q].w)) { best[q].w = d3; ai[q][3] = w; }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const int chq = ch + 4 * q;
    if (argmax && (!(MR_ABL & 8) || best[q].x == 1.2345e-30f)) {
      if (AK == 1) *reinterpret_cast<uint2*>(argmax + 2 * (t * C + chq)) =
          make_uint2((uint32_t)ai[q][0] | ((uint32_t)ai[q][1] << 16), (uint32_t)ai[q][2] | ((uint32_t)ai[q][3] << 16));
      else *reinterpret_cast<uint32_t*>(argmax + t * C + chq) =
          (uint32_t)ai[q][0] | ((uint32_t)ai[q][1] << 8) | ((uint32_t)ai[q][2] << 16) | ((uint32_t)ai[q][3] << 24);
    }
    if ((MR_ABL & 4) && best[q].y != 1.2345e-30f) continue;
    if (mode == 0) {
      stf4(out + t * C + chq, best[q]);
    } else {
      const int Cq = C >> 2;                      // channels per conv group = chunk width of the XM buffer (C % 16 == 0)
      OutT* o = out + t * (size_t)(2 * C) + xm_col(chq, Cq);
      if (write_x) stf4(o, xi[q]);
      stf4(o + Cq, best[q]);
    }
  }
#ifdef MR_TL
  MR_STAMP(3);
  __builtin_amdgcn_s_waitcnt(0);                                 // stores acknowledged
  MR_STAMP(4);
#endif
}

// Backward, pass 1: gx[t][ch] = direct[t][ch] - gm[t][ch]   (mode 1: direct / gm are the x / m chunks of dXM)
__global__ __launch_bounds__(256) void mr_bwd_tm_init_kernel(const float* __restrict__ gin, float* __restrict__ gx,
                                                             int C, size_t T, int mode) {
  const int C4 = C >> 2;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= T * C4) return;
  const size_t t = i / C4;
  const int ch = 4 * (int)(i - t * C4);
  float4 o;
  if (mode == 0) {
    const float4 g = *reinterpret_cast<const float4*>(gin + t * C + ch);
    o = make_float4(-g.x, -g.y, -g.z, -g.w);
  } else {
    const int Cq = C >> 2;
    const float* p = gin + t * (size_t)(2 * C) + xm_col(ch, Cq);
    const float4 u0 = *reinterpret_cast<const float4*>(p), u1 = *reinterpret_cast<const float4*>(p + Cq);
    o = make_float4(u0.x - u1.x, u0.y - u1.y, u0.z - u1.z, u0.w - u1.w);
  }
  *reinterpret_cast<float4*>(gx + t * C + ch) = o;
}

// Backward in ONE kernel.  One workgroup = one image x a chunk of CW channels; the destination rows of that (image,
// chunk) live in LDS ([M][CW] fp32).  Phase 1 seeds the image: self graph -> gx seed = direct - gm (read straight
// from the upstream gradient), bipartite graph -> zeros.  Phase 2 sweeps all N queries: gm is added at
// idx[n][argmax] with ds_add_f32 and, for the bipartite graph, gx = direct - gm is stored on the way.  Phase 3
// stores the image rows.  gx / gsrc are fully overwritten, no global atomics, no separate init pass.
// MODE / AK as template constants in the LDS-image kernels: with run-time `mode` / `ak` branches the float4 pairs and the
// target array went through SCRATCH memory (24 scratch_load/store_dword between the ds_add_f32 of the inner loop,
// .private_segment_fixed_size 40 — VERDICT r3); the run-time forms (MODE < 0) remain for the global-atomic fallbacks.
template <int MODE = -1>
__device__ __forceinline__ void load_grad(const float* __restrict__ gin, size_t T, size_t t, int C, int ch, int mode,
                                          float4& direct, float4& gm) {
  if (MODE >= 0) mode = MODE;
  if (mode == 0) {
    gm = *reinterpret_cast<const float4*>(gin + t * C + ch);
    direct = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {                                     // XM layout (T, 2C): the x chunk carries the direct gradient, the m chunk gm
    const int Cq = C >> 2;
    const float* p = gin + t * (size_t)(2 * C) + xm_col(ch, Cq);
    direct = *reinterpret_cast<const float4*>(p);
    gm = *reinterpret_cast<const float4*>(p + Cq);
  }
}

// destination rows of the 4 channels of one thread: from the winning slots + the index row (AK == 0) or directly (AK == 1)
template <int AK = -1>
__device__ __forceinline__ void mr_targets(const uint8_t* __restrict__ argmax, int ak, size_t elem, const int64_t* __restrict__ ip,
                                           int k, int M, int (&j)[4]) {
  if (AK >= 0) ak = AK;
  if (ak == 1) {
    const uint2 v = *reinterpret_cast<const uint2*>(argmax + 2 * elem);
    j[0] = min((int)(v.x & 0xffff), M - 1); j[1] = min((int)(v.x >> 16), M - 1);
    j[2] = min((int)(v.y & 0xffff), M - 1); j[3] = min((int)(v.y >> 16), M - 1);
  } else {
    const uint32_t am = *reinterpret_cast<const uint32_t*>(argmax + elem);
    const int km = k - 1;
    j[0] = clamp_idx(ip[min((int)(am & 0xff), km)], M); j[1] = clamp_idx(ip[min((int)((am >> 8) & 0xff), km)], M);
    j[2] = clamp_idx(ip[min((int)((am >> 16) & 0xff), km)], M); j[3] = clamp_idx(ip[min((int)((am >> 24) & 0xff), km)], M);
  }
}

template <bool SELF, int MODE, int AK>
__global__ __launch_bounds__(256) void mr_bwd_tm_scatter_kernel(const float* __restrict__ gin, const int64_t* __restrict__ nn_idx,
                                                                const uint8_t* __restrict__ argmax, float* __restrict__ gx,
                                                                float* __restrict__ gsrc, int B, int G, int c, int N, int M,
                                                                int k, int mode, int CW, int ak) {
  extern __shared__ float acc[];                  // [M][CW]
  const int C = G * c;
  // XCD-aware map (see mr_fwd_tm_kernel): the C/CW channel-chunk workgroups of one image read interleaved pieces of the
  // same gradient rows and index rows -> same XCD, adjacent in dispatch order
  const int nchunk = C / CW;
  const int lin = blockIdx.x;
  const int xcd = lin & 7, seq = lin >> 3;
  const int b = (seq / nchunk) * 8 + xcd;
  if (b >= B) return;
  const int ch0 = (seq - (seq / nchunk) * nchunk) * CW;
  const int cw4 = CW >> 2;                        // quads per row chunk
  const int tid = threadIdx.x;
  const int qd = tid % cw4, tl = tid / cw4, TL = 256 / cw4;
  const size_t T = (size_t)B * N;
  const int ch = ch0 + 4 * qd;
  // phase 1: seed the LDS image (self graph: a token's own "direct - gm"; bipartite graph: zeros).  Seeding with the same
  // LDS atomics as the scatter, to read every gradient row once instead of twice, was measured SLOWER (31.8 -> 40 us at
  // cfg2): the second read hits L2, the four extra atomics per row do not come free.
  if (tl < TL) {
    for (int m = tl; m < M; m += TL) {
      float4 seed = make_float4(0.f, 0.f, 0.f, 0.f);
      if (SELF) {
        float4 direct, gm;
        load_grad<MODE>(gin, T, (size_t)b * N + m, C, ch, mode, direct, gm);
        seed = make_float4(direct.x - gm.x, direct.y - gm.y, direct.z - gm.z, direct.w - gm.w);
      }
      *reinterpret_cast<float4*>(acc + (size_t)m * CW + 4 * qd) = seed;
    }
  }
  __syncthreads();
  // phase 2: sweep the queries; the destination rows come straight from the saved row indices (arg kind 1)
  if (tl < TL) {
    const int g = ch / c;
    for (int n = tl; n < N; n += TL) {
      const size_t t = (size_t)b * N + n;
      const int64_t* ip = nn_idx + (((size_t)b * G + g) * N + n) * k;
      int j[4];
      mr_targets<AK>(argmax, ak, t * C + ch, ip, k, M, j);
      float4 direct, gm;
      load_grad<MODE>(gin, T, t, C, ch, mode, direct, gm);
      if (!SELF)
        *reinterpret_cast<float4*>(gx + t * C + ch) = make_float4(direct.x - gm.x, direct.y - gm.y, direct.z - gm.z, direct.w - gm.w);
      atomicAdd(acc + (size_t)j[0] * CW + 4 * qd + 0, gm.x);
      atomicAdd(acc + (size_t)j[1] * CW + 4 * qd + 1, gm.y);
      atomicAdd(acc + (size_t)j[2] * CW + 4 * qd + 2, gm.z);
      atomicAdd(acc + (size_t)j[3] * CW + 4 * qd + 3, gm.w);
    }
  }
  __syncthreads();
  if (tl < TL) {
    float* db = (SELF ? gx : gsrc) + (size_t)b * M * C + ch0;
    for (int m = tl; m < M; m += TL)
      *reinterpret_cast<float4*>(db + (size_t)m * C + 4 * qd) = *reinterpret_cast<const float4*>(acc + (size_t)m * CW + 4 * qd);
  }
}

// ---- exact integer accumulation (round 4; the default LDS-image form) -------------------------------------------------------
// Hardware fact (tools/ubench/lds_atomic_rate.hip, profiles/r04_ubench_lds_atomic_rate.txt, SQ counters in
// profiles/r04_pmc_mr_stage1.txt): on gfx950 `ds_add_f32` retires at ~170 cycles per wave instruction (2.7 cycles per LANE,
// whatever the addresses), `ds_add_u64` at 30-66, `ds_add_u32` at 16-31.  The fp32 scatter above spends two thirds of its
// time in the LDS atomic unit.  This form accumulates in 64-bit FIXED POINT instead:
//   sweep 1  over the workgroup's (image, channel chunk): gx = direct - gm (bipartite graphs) and the largest |gm| bit
//            pattern (ds_max_u32) -> e_max, the scale of this workgroup;
//   sweep 2  every gm becomes  sign * (24-bit mantissa << sh),  sh = e - e_max + SHMAX  with SHMAX = 38 - bits(N): the
//            largest value keeps 24 + SHMAX bits, the sum of up to N of them fits 62 bits, values down to 2^-SHMAX of the
//            largest keep their FULL mantissa (29 binary orders at N = 324, 23 at N = 20 736; below that the low bits are
//            truncated — against a total that is then >= 2^24 times larger); one ds_add_u64 per value;
//   store    acc * 2^(e_max - 150 - SHMAX) converted to fp32 at the end — through a double (53 bits), so the 62-bit total is
//            rounded twice, not once, and the self graphs' "direct - gm" seed is one more fp32 add.
// Integer addition is associative: the result does not depend on the order the lanes arrive in — bit-identical from run
// to run (this form also serves GKG_MR_DETERMINISTIC) — and the sum carries the rounding of its conversion only (plus the
// truncation of addends more than 2^-SHMAX below the chunk's largest), where an fp32 atomic chain rounds after every addend:
// deterministic and at least as accurate as the fp32 sum, not "correctly rounded".  Non-finite gradients (the GradScaler overflow protocol needs inf / NaN
// to reach the parameters) show up as e_max = 255: the workgroup then re-runs the fp32-atomic form on the same LDS.
__device__ __forceinline__ long long mr_to_fixed(float v, int emax, int shmax) {
  const unsigned bits = __float_as_uint(v);
  const int e = (int)((bits >> 23) & 0xff);
  const unsigned long long mant = (unsigned long long)((bits & 0x7fffffu) | (e ? 0x800000u : 0u));
  const int sh = (e ? e : 1) - emax + shmax;
  const unsigned long long mag = sh >= 0 ? (mant << sh) : (mant >> min(-sh, 63));
  return (bits >> 31) ? -(long long)mag : (long long)mag;
}
__device__ __forceinline__ float mr_from_fixed(long long a, int emax, int shmax) {
  return (float)__builtin_ldexp((double)a, emax - 150 - shmax);
}

template <bool SELF, int MODE, int AK>
__device__ __forceinline__ void mr_scatter_f32_body(float* acc, const float* __restrict__ gin, const int64_t* __restrict__ nn_idx,
                                                    const uint8_t* __restrict__ argmax, float* __restrict__ gx,
                                                    float* __restrict__ gsrc, int b, int G, int c, int N, int M, int k, int CW,
                                                    int ch0, size_t T, bool write_gx) {
  const int C = G * c;
  const int cw4 = CW >> 2;
  const int tid = threadIdx.x;
  const int qd = tid % cw4, tl = tid / cw4, TL = 256 / cw4;
  const int ch = ch0 + 4 * qd;
  if (tl < TL) {
    for (int m = tl; m < M; m += TL) {
      float4 seed = make_float4(0.f, 0.f, 0.f, 0.f);
      if (SELF) {
        float4 direct, gm;
        load_grad<MODE>(gin, T, (size_t)b * N + m, C, ch, MODE, direct, gm);
        seed = make_float4(direct.x - gm.x, direct.y - gm.y, direct.z - gm.z, direct.w - gm.w);
      }
      *reinterpret_cast<float4*>(acc + (size_t)m * CW + 4 * qd) = seed;
    }
  }
  __syncthreads();
  if (tl < TL) {
    const int g = ch / c;
    for (int n = tl; n < N; n += TL) {
      const size_t t = (size_t)b * N + n;
      const int64_t* ip = nn_idx + (((size_t)b * G + g) * N + n) * k;
      int j[4];
      mr_targets<AK>(argmax, AK, t * C + ch, ip, k, M, j);
      float4 direct, gm;
      load_grad<MODE>(gin, T, t, C, ch, MODE, direct, gm);
      if (!SELF && write_gx)
        *reinterpret_cast<float4*>(gx + t * C + ch) = make_float4(direct.x - gm.x, direct.y - gm.y, direct.z - gm.z, direct.w - gm.w);
      atomicAdd(acc + (size_t)j[0] * CW + 4 * qd + 0, gm.x);
      atomicAdd(acc + (size_t)j[1] * CW + 4 * qd + 1, gm.y);
      atomicAdd(acc + (size_t)j[2] * CW + 4 * qd + 2, gm.z);
      atomicAdd(acc + (size_t)j[3] * CW + 4 * qd + 3, gm.w);
    }
  }
  __syncthreads();
  if (tl < TL) {
    float* db = (SELF ? gx : gsrc) + (size_t)b * M * C + ch0;
    for (int m = tl; m < M; m += TL)
      *reinterpret_cast<float4*>(db + (size_t)m * C + 4 * qd) = *reinterpret_cast<const float4*>(acc + (size_t)m * CW + 4 * qd);
  }
}

template <bool SELF, int MODE, int AK>
__global__ __launch_bounds__(256) void mr_bwd_tm_scatter_i64_kernel(const float* __restrict__ gin, const int64_t* __restrict__ nn_idx,
                                                                    const uint8_t* __restrict__ argmax, float* __restrict__ gx,
                                                                    float* __restrict__ gsrc, int B, int G, int c, int N, int M,
                                                                    int k, int CW, int shmax) {
  extern __shared__ long long acc64[];            // [M][CW] + control words
  const int C = G * c;
  const int nchunk = C / CW;
  const int lin = blockIdx.x;
  const int xcd = lin & 7, seq = lin >> 3;        // XCD-aware map, as mr_bwd_tm_scatter_kernel
  const int b = (seq / nchunk) * 8 + xcd;
  if (b >= B) return;
  const int ch0 = (seq - (seq / nchunk) * nchunk) * CW;
  const int cw4 = CW >> 2;
  const int tid = threadIdx.x;
  const int qd = tid % cw4, tl = tid / cw4, TL = 256 / cw4;
  const size_t T = (size_t)B * N;
  const int ch = ch0 + 4 * qd;
  unsigned* ctl = reinterpret_cast<unsigned*>(acc64 + (size_t)M * CW);
  for (int i = tid; i < M * CW; i += 256) acc64[i] = 0;
  if (tid == 0) ctl[0] = 0;
  __syncthreads();
  // ---- sweep 1: gx of the bipartite graph, and the chunk's largest |gm| bit pattern
  unsigned mx = 0;
  if (tl < TL) {
    for (int n = tl; n < N; n += TL) {
      const size_t t = (size_t)b * N + n;
      float4 direct, gm;
      load_grad<MODE>(gin, T, t, C, ch, MODE, direct, gm);
      if (!SELF)
        *reinterpret_cast<float4*>(gx + t * C + ch) = make_float4(direct.x - gm.x, direct.y - gm.y, direct.z - gm.z, direct.w - gm.w);
      mx = max(max(mx, __float_as_uint(gm.x) & 0x7fffffffu), max(__float_as_uint(gm.y) & 0x7fffffffu,
               max(__float_as_uint(gm.z) & 0x7fffffffu, __float_as_uint(gm.w) & 0x7fffffffu)));
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
  if ((tid & 63) == 0) atomicMax(ctl, mx);
  __syncthreads();
  mx = ctl[0];
  const int emax = (int)(mx >> 23);
  if (emax == 255) {                              // inf / NaN somewhere in this chunk: the fp32-atomic form propagates them
    __syncthreads();
    mr_scatter_f32_body<SELF, MODE, AK>(reinterpret_cast<float*>(acc64), gin, nn_idx, argmax, gx, gsrc, b, G, c, N, M, k, CW, ch0, T,
                                        false);
    return;
  }
  // ---- sweep 2: exact accumulation
  if (tl < TL && mx != 0) {
    const int g = ch / c;
    const int em = max(emax, 1);
    for (int n = tl; n < N; n += TL) {
      const size_t t = (size_t)b * N + n;
      const int64_t* ip = nn_idx + (((size_t)b * G + g) * N + n) * k;
      int j[4];
      mr_targets<AK>(argmax, AK, t * C + ch, ip, k, M, j);
      float4 direct, gm;
      load_grad<MODE>(gin, T, t, C, ch, MODE, direct, gm);
      atomicAdd(reinterpret_cast<unsigned long long*>(acc64 + (size_t)j[0] * CW + 4 * qd + 0), (unsigned long long)mr_to_fixed(gm.x, em, shmax));
      atomicAdd(reinterpret_cast<unsigned long long*>(acc64 + (size_t)j[1] * CW + 4 * qd + 1), (unsigned long long)mr_to_fixed(gm.y, em, shmax));
      atomicAdd(reinterpret_cast<unsigned long long*>(acc64 + (size_t)j[2] * CW + 4 * qd + 2), (unsigned long long)mr_to_fixed(gm.z, em, shmax));
      atomicAdd(reinterpret_cast<unsigned long long*>(acc64 + (size_t)j[3] * CW + 4 * qd + 3), (unsigned long long)mr_to_fixed(gm.w, em, shmax));
    }
  }
  __syncthreads();
  // ---- store: one rounding per element (+ the token's own seed for the self graph)
  if (tl < TL) {
    const int em = max(emax, 1);
    float* db = (SELF ? gx : gsrc) + (size_t)b * M * C + ch0;
    for (int m = tl; m < M; m += TL) {
      const long long* a = acc64 + (size_t)m * CW + 4 * qd;
      float4 o = make_float4(mr_from_fixed(a[0], em, shmax), mr_from_fixed(a[1], em, shmax), mr_from_fixed(a[2], em, shmax),
                             mr_from_fixed(a[3], em, shmax));
      if (SELF) {
        float4 direct, gm;
        load_grad<MODE>(gin, T, (size_t)b * N + m, C, ch, MODE, direct, gm);
        o = make_float4(o.x + (direct.x - gm.x), o.y + (direct.y - gm.y), o.z + (direct.z - gm.z), o.w + (direct.w - gm.w));
      }
      *reinterpret_cast<float4*>(db + (size_t)m * C + 4 * qd) = o;
    }
  }
}

// ---- streaming form of the exact scatter (round 5) ----------------------------------------------------------------------
// At the HBM-bound shapes (pooled 1 296-key images under 5 184 / 20 736 queries, the 36 x 36 self graphs) BOTH kernels above take
// the same time whatever their accumulator type (tools/bench_mr_bwd.py: 420-426 us at stage 1 for fp32 and 64-bit atomics, chunk
// widths 4 / 8): they are bound by their sweep — one (index, gradient) row per thread and iteration, nothing in flight behind
// it — and the fixed-point form pays that sweep twice to learn its scale first.  This form
//   * keeps U = 4 / 8 rows per thread in flight (all winning-row / gradient loads of an iteration issued before the first use) in
//     512-thread workgroups: one workgroup per CU (83 KB of accumulators at 1 296 rows x 8 channels) still has >= 80 KB of loads
//     outstanding (1 024-thread workgroups measured the same);
//   * sweeps ONCE: the fixed-point scale is the exact maximum when a thread's first U rows are the whole sweep (the 18 x 18
//     stages: the two-sweep kernel's bits), otherwise the maximum of a strided SAMPLE of the chunk's rows (requested together
//     with the first iteration's rows: one round trip) plus HEAD binary orders of headroom.  A gradient beyond the headroom (or
//     inf / NaN) raises a flag; the
//     workgroup then discards its image and re-runs the exact two-sweep form — same result contract, rare path.  The sample,
//     the flag and therefore the path taken depend on the data only: bit-identical from run to run, like the two-sweep form
//     (the scale sits HEAD orders above the sampled maximum: values more than 2^-(SHMAX - HEAD) below THAT lose low bits —
//     17 binary orders at N = 20 736).
// ACC == 1 (measurement): fp64 LDS atomics (ds_add_f64) instead of fixed point — no scale at all, not order-independent.
template <int NT, bool SELF, int MODE, int AK>
__device__ __forceinline__ void mr_i64_two_sweep_body(long long* acc64, unsigned* ctl, const float* __restrict__ gin,
                                                      const int64_t* __restrict__ nn_idx, const uint8_t* __restrict__ argmax,
                                                      float* __restrict__ gx, float* __restrict__ gsrc, int b, int G, int c, int N,
                                                      int M, int k, int CW, int ch0, size_t T, int shmax, bool write_gx) {
  const int C = G * c;
  const int cw4 = CW >> 2;
  const int tid = threadIdx.x;
  const int qd = tid % cw4, tl = tid / cw4, TL = NT / cw4;
  const int ch = ch0 + 4 * qd;
  unsigned mx = 0;
  if (tl < TL) {
    for (int n = tl; n < N; n += TL) {
      const size_t t = (size_t)b * N + n;
      float4 direct, gm;
      load_grad<MODE>(gin, T, t, C, ch, MODE, direct, gm);
      if (!SELF && write_gx)
        *reinterpret_cast<float4*>(gx + t * C + ch) = make_float4(direct.x - gm.x, direct.y - gm.y, direct.z - gm.z, direct.w - gm.w);
      mx = max(max(mx, __float_as_uint(gm.x) & 0x7fffffffu), max(__float_as_uint(gm.y) & 0x7fffffffu,
               max(__float_as_uint(gm.z) & 0x7fffffffu, __float_as_uint(gm.w) & 0x7fffffffu)));
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
  if ((tid & 63) == 0) atomicMax(ctl, mx);
  __syncthreads();
  mx = ctl[0];
  const int emax = (int)(mx >> 23);
  if (emax == 255) {                              // inf / NaN somewhere in this chunk: the fp32-atomic form propagates them
    __syncthreads();
    float* acc = reinterpret_cast<float*>(acc64);   // fp32 atomics on the same LDS (mr_scatter_f32_body for NT threads)
    if (tl < TL) {
      for (int m = tl; m < M; m += TL) {
        float4 seed = make_float4(0.f, 0.f, 0.f, 0.f);
        if (SELF) {
          float4 direct, gm;
          load_grad<MODE>(gin, T, (size_t)b * N + m, C, ch, MODE, direct, gm);
          seed = make_float4(direct.x - gm.x, direct.y - gm.y, direct.z - gm.z, direct.w - gm.w);
        }
        *reinterpret_cast<float4*>(acc + (size_t)m * CW + 4 * qd) = seed;
      }
    }
    __syncthreads();
    if (tl < TL) {
      const int g = ch / c;
      for (int n = tl; n < N; n += TL) {
        const size_t t = (size_t)b * N + n;
        const int64_t* ip = nn_idx + (((size_t)b * G + g) * N + n) * k;
        int j[4];
        mr_targets<AK>(argmax, AK, t * C + ch, ip, k, M, j);
        float4 direct, gm;
        load_grad<MODE>(gin, T, t, C, ch, MODE, direct, gm);
        atomicAdd(acc + (size_t)j[0] * CW + 4 * qd + 0, gm.x);
        atomicAdd(acc + (size_t)j[1] * CW + 4 * qd + 1, gm.y);
        atomicAdd(acc + (size_t)j[2] * CW + 4 * qd + 2, gm.z);
        atomicAdd(acc + (size_t)j[3] * CW + 4 * qd + 3, gm.w);
      }
    }
    __syncthreads();
    if (tl < TL) {
      float* db = (SELF ? gx : gsrc) + (size_t)b * M * C + ch0;
      for (int m = tl; m < M; m += TL)
        *reinterpret_cast<float4*>(db + (size_t)m * C + 4 * qd) = *reinterpret_cast<const float4*>(acc + (size_t)m * CW + 4 * qd);
    }
    return;
  }
  if (tl < TL && mx != 0) {
    const int g = ch / c;
    const int em = max(emax, 1);
    for (int n = tl; n < N; n += TL) {
      const size_t t = (size_t)b * N + n;
      const int64_t* ip = nn_idx + (((size_t)b * G + g) * N + n) * k;
      int j[4];
      mr_targets<AK>(argmax, AK, t * C + ch, ip, k, M, j);
      float4 direct, gm;
      load_grad<MODE>(gin, T, t, C, ch, MODE, direct, gm);
      atomicAdd(reinterpret_cast<unsigned long long*>(acc64 + (size_t)j[0] * CW + 4 * qd + 0), (unsigned long long)mr_to_fixed(gm.x, em, shmax));
      atomicAdd(reinterpret_cast<unsigned long long*>(acc64 + (size_t)j[1] * CW + 4 * qd + 1), (unsigned long long)mr_to_fixed(gm.y, em, shmax));
      atomicAdd(reinterpret_cast<unsigned long long*>(acc64 + (size_t)j[2] * CW + 4 * qd + 2), (unsigned long long)mr_to_fixed(gm.z, em, shmax));
      atomicAdd(reinterpret_cast<unsigned long long*>(acc64 + (size_t)j[3] * CW + 4 * qd + 3), (unsigned long long)mr_to_fixed(gm.w, em, shmax));
    }
  }
  __syncthreads();
  if (tl < TL) {
    const int em = max(emax, 1);
    float* db = (SELF ? gx : gsrc) + (size_t)b * M * C + ch0;
    for (int m = tl; m < M; m += TL) {
      const long long* a = acc64 + (size_t)m * CW + 4 * qd;
      float4 o = make_float4(mr_from_fixed(a[0], em, shmax), mr_from_fixed(a[1], em, shmax), mr_from_fixed(a[2], em, shmax),
                             mr_from_fixed(a[3], em, shmax));
      if (SELF) {
        float4 direct, gm;
        load_grad<MODE>(gin, T, (size_t)b * N + m, C, ch, MODE, direct, gm);
        o = make_float4(o.x + (direct.x - gm.x), o.y + (direct.y - gm.y), o.z + (direct.z - gm.z), o.w + (direct.w - gm.w));
      }
      *reinterpret_cast<float4*>(db + (size_t)m * C + 4 * qd) = o;
    }
  }
}

constexpr int MR_STREAM_HEAD = 6;       // binary orders of headroom above the sampled maximum

template <int NT, bool SELF, int MODE, int AK, int ACC, int U>     // U: rows in flight per thread
__global__ __launch_bounds__(NT) void mr_bwd_tm_stream_kernel(const float* __restrict__ gin, const int64_t* __restrict__ nn_idx,
                                                               const uint8_t* __restrict__ argmax, float* __restrict__ gx,
                                                               float* __restrict__ gsrc, int B, int G, int c, int N, int M,
                                                               int k, int CW, int shmax, int arg_planes) {
  extern __shared__ long long acc64[];            // [M][CW] (fixed point, or the bits of doubles) + control words
  const int C = G * c;
  const int nchunk = C / CW;
  const int lin = blockIdx.x;
  const int xcd = lin & 7, seq = lin >> 3;        // XCD-aware map, as mr_bwd_tm_scatter_kernel
  const int b = (seq / nchunk) * 8 + xcd;
  if (b >= B) return;
  const int ch0 = (seq - (seq / nchunk) * nchunk) * CW;
  const int cw4 = CW >> 2;
  const int tid = threadIdx.x;
  const int qd = tid % cw4, tl = tid / cw4, TL = NT / cw4;
  const size_t T = (size_t)B * N;
  const int ch = ch0 + 4 * qd;
  const int g = ch / c;
  unsigned* ctl = reinterpret_cast<unsigned*>(acc64 + (size_t)M * CW);
  // arg_planes (measurement, AK == 1): the winning rows as 8-channel planes (B, C / 8, N, 8) instead of (B, N, C) rows — a chunk's
  // 16 bytes per query row are then contiguous over the rows
  const size_t pl_base = (((size_t)b * (C >> 3) + (ch >> 3)) * N) * 8 + (ch & 7);
  float4 direct[U], gm[U];
  int j[U][4];
  auto load_rows = [&](int n0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = min(n0 + u * TL, N - 1);
      const size_t t = (size_t)b * N + n;
      const int64_t* ip = AK == 1 ? nullptr : nn_idx + (((size_t)b * G + g) * N + n) * k;
      mr_targets<AK>(argmax, AK, arg_planes ? pl_base + (size_t)n * 8 : t * C + ch, ip, k, M, j[u]);
      load_grad<MODE>(gin, T, t, C, ch, MODE, direct[u], gm[u]);
    }
  };
  auto absmax4 = [](const float4& v) -> unsigned {
    return max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu),
               max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu));
  };
  // ---- prologue: the first U rows of every thread AND (longer sweeps) a strided sample of the chunk's rows are requested in one
  //      go; the LDS image is cleared in their shadow.  The fixed-point scale comes from what arrived: the exact maximum when the
  //      first iteration is the whole sweep (N <= TL * U: the 18 x 18 stages — no headroom needed, the flag cannot rise), the
  //      sample's maximum plus HEAD binary orders otherwise.
  const bool whole = N <= TL * U;                 // uniform
  unsigned mx = 0;
  float4 sd[2], sg[2];
  if (ACC == 0 && !whole) {
    const int stride = max(N / (TL * 2), 1);
#pragma unroll
    for (int u = 0; u < 2; ++u)
      load_grad<MODE>(gin, T, (size_t)b * N + min((tl + u * TL) * stride, N - 1), C, ch, MODE, sd[u], sg[u]);
  }
  load_rows(tl);
  for (int i = tid; i < M * CW; i += NT) acc64[i] = 0;
  if (tid == 0) { ctl[0] = 0; ctl[1] = 0; }
  __syncthreads();
  int em = 0;
  unsigned lim = 0xffffffffu;
  if (ACC == 0) {
    if (!whole) mx = max(absmax4(sg[0]), absmax4(sg[1]));
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (tl + u * TL < N) mx = max(mx, absmax4(gm[u]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    if ((tid & 63) == 0) atomicMax(ctl, mx);
    __syncthreads();
    mx = ctl[0];
    // an all-zero (or denormal) sample still gets a scale — whatever lies above it raises the flag
    em = min(max((int)(mx >> 23), 1) + (whole ? 0 : MR_STREAM_HEAD), 254);
    lim = (unsigned)(em + 1) << 23;               // |bits| >= lim: beyond the scale (inf / NaN always are)
  }
  bool over = false;
  auto process_rows = [&](int n0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = n0 + u * TL;
      if (n < N) {
        const size_t t = (size_t)b * N + n;
        if (!SELF)
          *reinterpret_cast<float4*>(gx + t * C + ch) =
              make_float4(direct[u].x - gm[u].x, direct[u].y - gm[u].y, direct[u].z - gm[u].z, direct[u].w - gm[u].w);
        if (ACC == 0) {
          over |= absmax4(gm[u]) >= lim;
          atomicAdd(reinterpret_cast<unsigned long long*>(acc64 + (size_t)j[u][0] * CW + 4 * qd + 0), (unsigned long long)mr_to_fixed(gm[u].x, em, shmax));
          atomicAdd(reinterpret_cast<unsigned long long*>(acc64 + (size_t)j[u][1] * CW + 4 * qd + 1), (unsigned long long)mr_to_fixed(gm[u].y, em, shmax));
          atomicAdd(reinterpret_cast<unsigned long long*>(acc64 + (size_t)j[u][2] * CW + 4 * qd + 2), (unsigned long long)mr_to_fixed(gm[u].z, em, shmax));
          atomicAdd(reinterpret_cast<unsigned long long*>(acc64 + (size_t)j[u][3] * CW + 4 * qd + 3), (unsigned long long)mr_to_fixed(gm[u].w, em, shmax));
        } else {
          double* ad = reinterpret_cast<double*>(acc64);
          unsafeAtomicAdd(ad + (size_t)j[u][0] * CW + 4 * qd + 0, (double)gm[u].x);
          unsafeAtomicAdd(ad + (size_t)j[u][1] * CW + 4 * qd + 1, (double)gm[u].y);
          unsafeAtomicAdd(ad + (size_t)j[u][2] * CW + 4 * qd + 2, (double)gm[u].z);
          unsafeAtomicAdd(ad + (size_t)j[u][3] * CW + 4 * qd + 3, (double)gm[u].w);
        }
      }
    }
  };
  // ---- the sweep: U rows per thread in flight
  process_rows(tl);
  for (int n0 = tl + TL * U; n0 < N; n0 += TL * U) {
    load_rows(n0);
    process_rows(n0);
  }
  if (ACC == 0) {
    if (over) ctl[1] = 1;
    __syncthreads();
    if (ctl[1]) {                                   // beyond the scale somewhere (inf / NaN included): the exact two-sweep form on a clean image
      __syncthreads();
      for (int i = tid; i < M * CW; i += NT) acc64[i] = 0;
      if (tid == 0) ctl[0] = 0;
      __syncthreads();
      mr_i64_two_sweep_body<NT, SELF, MODE, AK>(acc64, ctl, gin, nn_idx, argmax, gx, gsrc, b, G, c, N, M, k, CW, ch0, T, shmax, false);
      return;
    }
  } else {
    __syncthreads();
  }
  // ---- store: one conversion per element (+ the token's own seed for the self graph)
  if (tl < TL) {
    float* db = (SELF ? gx : gsrc) + (size_t)b * M * C + ch0;
    for (int m = tl; m < M; m += TL) {
      const long long* a = acc64 + (size_t)m * CW + 4 * qd;
      float4 o;
      if (ACC == 0) {
        o = make_float4(mr_from_fixed(a[0], em, shmax), mr_from_fixed(a[1], em, shmax), mr_from_fixed(a[2], em, shmax),
                        mr_from_fixed(a[3], em, shmax));
      } else {
        const double* ad = reinterpret_cast<const double*>(a);
        o = make_float4((float)ad[0], (float)ad[1], (float)ad[2], (float)ad[3]);
      }
      if (SELF) {
        float4 direct1, gm1;
        load_grad<MODE>(gin, T, (size_t)b * N + m, C, ch, MODE, direct1, gm1);
        o = make_float4(o.x + (direct1.x - gm1.x), o.y + (direct1.y - gm1.y), o.z + (direct1.z - gm1.z), o.w + (direct1.w - gm1.w));
      }
      *reinterpret_cast<float4*>(db + (size_t)m * C + 4 * qd) = o;
    }
  }
}

// Deterministic scatter (GKG_MR_DETERMINISTIC): the LDS-atomic kernel above adds the fan-in of a key in whatever order
// its lanes arrive, so gsrc differs in the last bit from run to run (like the reference's CUDA index_put_(accumulate)).
// Here a workgroup owns ONE channel quad of one image; each of its TL threads sweeps its own residue class of queries
// (n = tl, tl + TL, ...) in ascending order into a PRIVATE copy of the destination rows ([TL][M] float4 in LDS, plain
// read-modify-write, no atomics); the copies are then added in thread order.  Every sum has a fixed order -> results
// are bit-identical from run to run.
template <bool SELF, int MODE, int AK>
__global__ __launch_bounds__(64) void mr_bwd_tm_det_kernel(const float* __restrict__ gin, const int64_t* __restrict__ nn_idx,
                                                           const uint8_t* __restrict__ argmax, float* __restrict__ gx,
                                                           float* __restrict__ gsrc, int B, int G, int c, int N, int M,
                                                           int k, int mode, int TL, int ak) {
  extern __shared__ float acc[];                  // [TL][M][4]
  const int C = G * c;
  const int b = blockIdx.y;
  const int ch = 4 * blockIdx.x;
  const int tl = threadIdx.x;
  const size_t T = (size_t)B * N;
  const int g = ch / c;
  for (int i = tl; i < TL * M; i += 64) *reinterpret_cast<float4*>(acc + 4 * (size_t)i) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  if (tl < TL) {
    float* mine = acc + (size_t)tl * M * 4;
    for (int n = tl; n < N; n += TL) {
      const size_t t = (size_t)b * N + n;
      const int64_t* ip = nn_idx + (((size_t)b * G + g) * N + n) * k;
      int j[4];
      mr_targets<AK>(argmax, ak, t * C + ch, ip, k, M, j);
      float4 direct, gm;
      load_grad<MODE>(gin, T, t, C, ch, mode, direct, gm);
      if (!SELF)
        *reinterpret_cast<float4*>(gx + t * C + ch) = make_float4(direct.x - gm.x, direct.y - gm.y, direct.z - gm.z, direct.w - gm.w);
      mine[4 * j[0] + 0] += gm.x;
      mine[4 * j[1] + 1] += gm.y;
      mine[4 * j[2] + 2] += gm.z;
      mine[4 * j[3] + 3] += gm.w;
    }
  }
  __syncthreads();
  float* db = (SELF ? gx : gsrc) + (size_t)b * M * C + ch;
  for (int m = tl; m < M; m += 64) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (SELF) {                                    // centre term of the self graph: direct - gm of the token itself
      float4 direct, gm;
      load_grad<MODE>(gin, T, (size_t)b * N + m, C, ch, mode, direct, gm);
      s = make_float4(direct.x - gm.x, direct.y - gm.y, direct.z - gm.z, direct.w - gm.w);
    }
    for (int t2 = 0; t2 < TL; ++t2) {
      const float4 v = *reinterpret_cast<const float4*>(acc + ((size_t)t2 * M + m) * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(db + (size_t)m * C) = s;
  }
}

// Deterministic fallback for destination images too large for LDS (label graphs over > 9 600 keys: few queries): one
// thread per (image, channel quad) walks its queries in order and read-modify-writes the pre-seeded global rows.
template <int MODE, int AK>
__device__ __forceinline__ void det_global_walk(const float* __restrict__ gin, const int64_t* __restrict__ nn_idx,
                                                const uint8_t* __restrict__ argmax, float* __restrict__ db, int b, int g, int ch,
                                                int C, int G, int N, int M, int k, size_t T) {
  for (int n = 0; n < N; ++n) {
    const size_t t = (size_t)b * N + n;
    const int64_t* ip = nn_idx + (((size_t)b * G + g) * N + n) * k;
    int j[4];
    mr_targets<AK>(argmax, AK, t * C + ch, ip, k, M, j);
    float4 direct, gm;
    load_grad<MODE>(gin, T, t, C, ch, MODE, direct, gm);
    db[(size_t)j[0] * C + 0] += gm.x;
    db[(size_t)j[1] * C + 1] += gm.y;
    db[(size_t)j[2] * C + 2] += gm.z;
    db[(size_t)j[3] * C + 3] += gm.w;
  }
}

__global__ __launch_bounds__(64) void mr_bwd_tm_det_global_kernel(const float* __restrict__ gin, const int64_t* __restrict__ nn_idx,
                                                                  const uint8_t* __restrict__ argmax, float* __restrict__ dst,
                                                                  int B, int G, int c, int N, int M, int k, int mode, int ak) {
  const int C = G * c, C4 = C >> 2;
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= B * C4) return;
  const int b = i / C4, ch = 4 * (i - b * C4);
  const size_t T = (size_t)B * N;
  const int g = ch / c;
  float* db = dst + (size_t)b * M * C + ch;
  // the (mode, arg kind) branches are hoisted out of the walk: inside it they sent the float4s through scratch memory
  if (mode == 0) {
    if (ak) det_global_walk<0, 1>(gin, nn_idx, argmax, db, b, g, ch, C, G, N, M, k, T);
    else det_global_walk<0, 0>(gin, nn_idx, argmax, db, b, g, ch, C, G, N, M, k, T);
  } else {
    if (ak) det_global_walk<1, 1>(gin, nn_idx, argmax, db, b, g, ch, C, G, N, M, k, T);
    else det_global_walk<1, 0>(gin, nn_idx, argmax, db, b, g, ch, C, G, N, M, k, T);
  }
}

// Fallback when a destination image does not fit in LDS even for a 4-channel chunk: fp32 global atomics.
__global__ __launch_bounds__(256) void mr_bwd_tm_scatter_atomic_kernel(const float* __restrict__ gin, const int64_t* __restrict__ nn_idx,
                                                                       const uint8_t* __restrict__ argmax, float* __restrict__ dst,
                                                                       int B, int G, int c, int N, int M, int k, int mode, int ak) {
  const int C = G * c, C4 = C >> 2;
  const size_t T = (size_t)B * N;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= T * C4) return;
  const size_t t = i / C4;
  const int ch = 4 * (int)(i - t * C4);
  const int b = (int)(t / N), n = (int)(t - (size_t)b * N);
  const int g = ch / c;
  const int64_t* ip = nn_idx + (((size_t)b * G + g) * N + n) * k;
  int j[4];
  mr_targets(argmax, ak, t * C + ch, ip, k, M, j);
  float4 gm;
  if (mode == 0) {
    gm = *reinterpret_cast<const float4*>(gin + t * C + ch);
  } else {
    const int Cq = C >> 2;
    gm = *reinterpret_cast<const float4*>(gin + t * (size_t)(2 * C) + xm_col(ch, Cq) + Cq);
  }
  float* db = dst + (size_t)b * M * C + ch;
  atomicAdd(db + (size_t)j[0] * C + 0, gm.x);
  atomicAdd(db + (size_t)j[1] * C + 1, gm.y);
  atomicAdd(db + (size_t)j[2] * C + 2, gm.z);
  atomicAdd(db + (size_t)j[3] * C + 3, gm.w);
}

}  // namespace gkg

using namespace gkg;

static int pick_ch(int c, int M, int extra_bytes) {
  long ch = ((long)MR_LDS_BUDGET - extra_bytes) / ((long)M * 4);
  if (ch > c) ch = c;
  if (ch > 64) ch = 64;
  return (int)ch;
}

template <typename T>
static hipError_t mr_fwd_launch(const void* x, const void* src, const int64_t* nn_idx, void* m_out, uint8_t* argmax,
                                int BG, int c, int N, int M, int k, int CH, hipStream_t st) {
  GkgProfScope prof(GKG_PROF_MR_FWD, st);
  const size_t lds = (size_t)CH * M * 4 + (size_t)k * 256 * 4;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mr_fwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid((N + 255) / 256, (c + CH - 1) / CH, BG);
  hipLaunchKernelGGL((mr_fwd_kernel<T>), grid, dim3(256), lds, st, (const T*)x, (const T*)src, nn_idx, (T*)m_out, argmax,
                     c, N, M, k, CH);
  return hipGetLastError();
}

extern "C" int gkg_mr_fwd(const void* x, const void* src, const int64_t* nn_idx, void* m_out, uint8_t* argmax,
                          int BG, int c, int N, int M, int k, int dtype, void* stream) {
  if (!x || !nn_idx || !m_out) return gkg_fail(GKG_ERR_NULL, "gkg_mr_fwd: x, nn_idx and m_out must be non-null");
  if (BG <= 0 || c <= 0 || N <= 0 || M <= 0 || k <= 0 || k > 255) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_fwd: bad sizes (k <= 255)");
  if (!src) { if (M != N) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_fwd: self graph needs M == N"); src = x; }
  if (dtype != GKG_F32 && dtype != GKG_BF16 && dtype != GKG_F16) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_fwd: dtype");
  if (BG > 65535) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_fwd: BG <= 65535");
  const int extra = k * 256 * 4;
  // channels per workgroup: as many source rows as fit, but keep enough workgroups to fill the chip
  int CH = pick_ch(c, M, extra);
  if (CH < 1) {                      // source row does not fit in LDS: gather from L2
    GkgProfScope prof(GKG_PROF_MR_FWD, (hipStream_t)stream);
    dim3 grid((N + 255) / 256, c < 64 ? c : 64, BG);
    if (dtype == GKG_F32) hipLaunchKernelGGL((mr_fwd_gather_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)src, nn_idx, (float*)m_out, argmax, c, N, M, k);
    else if (dtype == GKG_F16) hipLaunchKernelGGL((mr_fwd_gather_kernel<_Float16>), grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (const _Float16*)src, nn_idx, (_Float16*)m_out, argmax, c, N, M, k);
    else hipLaunchKernelGGL((mr_fwd_gather_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)src, nn_idx, (uint16_t*)m_out, argmax, c, N, M, k);
    hipError_t e2 = hipGetLastError();
    return e2 == hipSuccess ? 0 : gkg_fail_hip(e2, "mr_fwd_gather_kernel");
  }
  const long ntiles = (N + 255) / 256;
  while (CH > 4 && ntiles * ((c + CH - 1) / CH) * BG < 1024) CH = (CH + 1) / 2;
  hipError_t e = dtype == GKG_F32 ? mr_fwd_launch<float>(x, src, nn_idx, m_out, argmax, BG, c, N, M, k, CH, (hipStream_t)stream)
               : dtype == GKG_F16 ? mr_fwd_launch<_Float16>(x, src, nn_idx, m_out, argmax, BG, c, N, M, k, CH, (hipStream_t)stream)
                                  : mr_fwd_launch<uint16_t>(x, src, nn_idx, m_out, argmax, BG, c, N, M, k, CH, (hipStream_t)stream);
  if (e != hipSuccess) return gkg_fail_hip(e, "mr_fwd_kernel");
  return 0;
}

template <typename T>
static hipError_t mr_bwd_launch(const void* g, const int64_t* nn_idx, const uint8_t* argmax, void* gx, void* gsrc,
                                int BG, int c, int N, int M, int k, int CH, hipStream_t st) {
  GkgProfScope prof(GKG_PROF_MR_BWD, st);
  const size_t lds = (size_t)CH * M * 4 + (size_t)k * 256 * 4;
  dim3 grid((c + CH - 1) / CH, BG);
  if (gsrc) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mr_bwd_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mr_bwd_kernel<T, false>), grid, dim3(256), lds, st, (const T*)g, nn_idx, argmax, (T*)gx, (T*)gsrc, c, N, M, k, CH);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mr_bwd_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mr_bwd_kernel<T, true>), grid, dim3(256), lds, st, (const T*)g, nn_idx, argmax, (T*)gx, (T*)nullptr, c, N, M, k, CH);
  }
  return hipGetLastError();
}

extern "C" int gkg_mr_bwd(const void* g, const int64_t* nn_idx, const uint8_t* argmax, void* gx, void* gsrc,
                          int BG, int c, int N, int M, int k, int dtype, void* stream) {
  if (!g || !nn_idx || !argmax || !gx) return gkg_fail(GKG_ERR_NULL, "gkg_mr_bwd: g, nn_idx, argmax and gx must be non-null");
  if (BG <= 0 || c <= 0 || N <= 0 || M <= 0 || k <= 0 || k > 255) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_bwd: bad sizes (k <= 255)");
  if (!gsrc && M != N) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_bwd: self graph needs M == N");
  if (dtype != GKG_F32 && dtype != GKG_BF16 && dtype != GKG_F16) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_bwd: dtype");
  if (BG > 65535) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_bwd: BG <= 65535");
  const int extra = k * 256 * 4;
  int CH = pick_ch(c, M, extra);
  if (CH < 1) {                      // accumulator row does not fit in LDS: fp32 global atomics (fp32 tensors only)
    if (dtype != GKG_F32) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_bwd: M too large for the LDS accumulator and dtype is not fp32");
    hipStream_t st = (hipStream_t)stream;
    GkgProfScope prof(GKG_PROF_MR_BWD, st);
    const size_t nx = (size_t)BG * c * N;
    if (gsrc) (void)hipMemsetAsync(gsrc, 0, sizeof(float) * (size_t)BG * c * M, st);
    else hipLaunchKernelGGL(negate_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st, (const float*)g, (float*)gx, nx);
    dim3 grid((N + 255) / 256, c < 64 ? c : 64, BG);
    hipLaunchKernelGGL(mr_bwd_atomic_kernel, grid, dim3(256), 0, st, (const float*)g, nn_idx, argmax, (float*)gx,
                       gsrc ? (float*)gsrc : (float*)gx, c, N, M, k, gsrc ? 0 : 1);
    hipError_t e2 = hipGetLastError();
    return e2 == hipSuccess ? 0 : gkg_fail_hip(e2, "mr_bwd_atomic_kernel");
  }
  while (CH > 2 && (long)((c + CH - 1) / CH) * BG < 1024) CH = (CH + 1) / 2;
  hipError_t e = dtype == GKG_F32 ? mr_bwd_launch<float>(g, nn_idx, argmax, gx, gsrc, BG, c, N, M, k, CH, (hipStream_t)stream)
               : dtype == GKG_F16 ? mr_bwd_launch<_Float16>(g, nn_idx, argmax, gx, gsrc, BG, c, N, M, k, CH, (hipStream_t)stream)
                                  : mr_bwd_launch<uint16_t>(g, nn_idx, argmax, gx, gsrc, BG, c, N, M, k, CH, (hipStream_t)stream);
  if (e != hipSuccess) return gkg_fail_hip(e, "mr_bwd_kernel");
  return 0;
}

// ------------------------------------------------------------------------------------------ token-major entry points
template <typename IdxT>
static int mr_fwd_tm_impl(const float* x, int ldx, int xchunk, const float* src, const IdxT* nn_idx, void* out, uint8_t* argmax,
                          int B, int G, int c, int N, int M, int k, int mode, int out_dtype, int arg_kind, void* stream) {
  if (arg_kind != 0 && arg_kind != 1) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_fwd_tm: arg_kind is 0 (u8 slot) or 1 (u16 row index)");
  if (arg_kind == 1 && M > 65536) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_fwd_tm: arg_kind 1 needs M <= 65536");
  if (!x || !nn_idx || !out) return gkg_fail(GKG_ERR_NULL, "gkg_mr_fwd_tm: x, nn_idx and out must be non-null");
  if (B <= 0 || G <= 0 || c <= 0 || N <= 0 || M <= 0 || k <= 0 || k > 255 || (c & 3)) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_fwd_tm: bad sizes (c % 4 == 0, k <= 255)");
  if (mode != 0 && mode != 1) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_fwd_tm: mode is 0 or 1");
  if (mode == 1 && ((G * c) & 15)) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_fwd_tm: mode 1 needs C % 16 == 0");
  if (out_dtype != GKG_F32 && out_dtype != GKG_BF16) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_fwd_tm: out_dtype is GKG_F32 or GKG_BF16");
  const int C = G * c;
  if (ldx == 0) ldx = C;
  if (ldx < C || (ldx & 3) || xchunk < 0 || (xchunk & 3) || (xchunk > 0 && (C % xchunk || ldx < 2 * C)))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_fwd_tm: bad x view (ldx >= C, ldx % 4 == 0; a chunk is a multiple of 4 dividing C, with ldx >= 2 C)");
  const bool has_src = src != nullptr;
  int lds_ = C, schunk = 0;
  if (!src) { if (M != N) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_fwd_tm: self graph needs M == N"); src = x; lds_ = ldx; schunk = xchunk; }
  // mode 1: x already lives in the operand buffer when the caller passes that buffer's x half as the view of x
  const int write_x = !(mode == 1 && (const void*)x == (const void*)out && ldx == 2 * C && xchunk == (C >> 2) && out_dtype == GKG_F32);
  if (mode == 1 && (const void*)x == (const void*)out && write_x) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_fwd_tm: x aliases out but is not its x half (ldx = 2 C, xchunk = C / 4, fp32)");
  // algorithmic bytes (SURVEY §8d "MR gather-max fwd"): x + (keys) + int64 indices + m (+ 1 byte of argmax per element)
  const double e_out = out_dtype == GKG_BF16 ? 2.0 : 4.0;
  const double work = 4.0 * B * (double)G * c * N + (has_src ? 4.0 * B * (double)G * c * M : 0.0) + (double)sizeof(IdxT) * B * (double)G * N * k
                      + e_out * B * (double)G * c * N + (argmax ? 1.0 * B * (double)G * c * N : 0.0);
  GkgProfScope prof(GKG_PROF_MR_FWD, (hipStream_t)stream, work);
  // one channel quad per thread: two quads per thread (shared index row) measured 20 % slower at cfg2 — the kernel
  // wants more threads in flight, not fewer index loads
  const long bpi = ((long)N * (G * c / 4) + 255) / 256;
  if (bpi * (((long)B + 7) / 8) * 8 > 0x7fffffffL) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_fwd_tm: problem too large for one launch");
  const dim3 grid((unsigned)(bpi * ((B + 7) / 8) * 8));
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == GKG_BF16) {
    uint16_t* o = (uint16_t*)out;
    if (k == 9) { if (arg_kind == 1) hipLaunchKernelGGL((mr_fwd_tm_kernel<9, 1, uint16_t, 1, IdxT>), grid, dim3(256), 0, st, x, src, nn_idx, o, argmax, B, G, c, N, M, k, mode, ldx, xchunk, lds_, schunk, write_x); else hipLaunchKernelGGL((mr_fwd_tm_kernel<9, 1, uint16_t, 0, IdxT>), grid, dim3(256), 0, st, x, src, nn_idx, o, argmax, B, G, c, N, M, k, mode, ldx, xchunk, lds_, schunk, write_x); }
    else { if (arg_kind == 1) hipLaunchKernelGGL((mr_fwd_tm_kernel<0, 1, uint16_t, 1, IdxT>), grid, dim3(256), 0, st, x, src, nn_idx, o, argmax, B, G, c, N, M, k, mode, ldx, xchunk, lds_, schunk, write_x); else hipLaunchKernelGGL((mr_fwd_tm_kernel<0, 1, uint16_t, 0, IdxT>), grid, dim3(256), 0, st, x, src, nn_idx, o, argmax, B, G, c, N, M, k, mode, ldx, xchunk, lds_, schunk, write_x); }
  } else {
    float* o = (float*)out;
    if (k == 9) { if (arg_kind == 1) hipLaunchKernelGGL((mr_fwd_tm_kernel<9, 1, float, 1, IdxT>), grid, dim3(256), 0, st, x, src, nn_idx, o, argmax, B, G, c, N, M, k, mode, ldx, xchunk, lds_, schunk, write_x); else hipLaunchKernelGGL((mr_fwd_tm_kernel<9, 1, float, 0, IdxT>), grid, dim3(256), 0, st, x, src, nn_idx, o, argmax, B, G, c, N, M, k, mode, ldx, xchunk, lds_, schunk, write_x); }
    else { if (arg_kind == 1) hipLaunchKernelGGL((mr_fwd_tm_kernel<0, 1, float, 1, IdxT>), grid, dim3(256), 0, st, x, src, nn_idx, o, argmax, B, G, c, N, M, k, mode, ldx, xchunk, lds_, schunk, write_x); else hipLaunchKernelGGL((mr_fwd_tm_kernel<0, 1, float, 0, IdxT>), grid, dim3(256), 0, st, x, src, nn_idx, o, argmax, B, G, c, N, M, k, mode, ldx, xchunk, lds_, schunk, write_x); }
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "mr_fwd_tm_kernel");
}

extern "C" int gkg_mr_fwd_tm(const float* x, int ldx, int xchunk, const float* src, const int64_t* nn_idx, void* out, uint8_t* argmax,
                             int B, int G, int c, int N, int M, int k, int mode, int out_dtype, int arg_kind, void* stream) {
  return mr_fwd_tm_impl<int64_t>(x, ldx, xchunk, src, nn_idx, out, argmax, B, G, c, N, M, k, mode, out_dtype, arg_kind, stream);
}

// gkg_mr_fwd_tm over the compact neighbour lists of gkg_knn_fwd_tm16 (u16 rows, M <= 65 536): same outputs, same bits; the
// index stream is a quarter of the int64 one (SURVEY §8d "MR gather-max fwd": 8 B G N k -> 2 B G N k bytes).
extern "C" int gkg_mr_fwd_tm16(const float* x, int ldx, int xchunk, const float* src, const uint16_t* nn16, void* out, uint8_t* argmax,
                               int B, int G, int c, int N, int M, int k, int mode, int out_dtype, int arg_kind, void* stream) {
  if (M > 65536) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_fwd_tm16: M <= 65536 (u16 rows)");
  return mr_fwd_tm_impl<uint16_t>(x, ldx, xchunk, src, nn16, out, argmax, B, G, c, N, M, k, mode, out_dtype, arg_kind, stream);
}

// The LDS-image backward kernels are instantiated per (self graph, mode, arg kind): 8 forms each, picked here.
template <bool DET, bool SELF, int MODE, int AK>
static void launch_tm_lds_one(dim3 grid, dim3 block, size_t lds, hipStream_t st, const float* gin, const int64_t* nn_idx,
                              const uint8_t* argmax, float* gx, float* gsrc, int B, int G, int c, int N, int M, int k, int last) {
  if (DET) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mr_bwd_tm_det_kernel<SELF, MODE, AK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mr_bwd_tm_det_kernel<SELF, MODE, AK>), grid, block, lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k, MODE, last, AK);
  } else {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mr_bwd_tm_scatter_kernel<SELF, MODE, AK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mr_bwd_tm_scatter_kernel<SELF, MODE, AK>), grid, block, lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k, MODE, last, AK);
  }
}

template <bool DET>
static void launch_tm_lds(bool self, int mode, int ak, dim3 grid, dim3 block, size_t lds, hipStream_t st, const float* gin,
                          const int64_t* nn_idx, const uint8_t* argmax, float* gx, float* gsrc, int B, int G, int c, int N, int M,
                          int k, int last) {
#define GKG_TM_CASE(S, MO, A) launch_tm_lds_one<DET, S, MO, A>(grid, block, lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k, last)
  if (self) {
    if (mode == 0) { if (ak) GKG_TM_CASE(true, 0, 1); else GKG_TM_CASE(true, 0, 0); }
    else { if (ak) GKG_TM_CASE(true, 1, 1); else GKG_TM_CASE(true, 1, 0); }
  } else {
    if (mode == 0) { if (ak) GKG_TM_CASE(false, 0, 1); else GKG_TM_CASE(false, 0, 0); }
    else { if (ak) GKG_TM_CASE(false, 1, 1); else GKG_TM_CASE(false, 1, 0); }
  }
#undef GKG_TM_CASE
}

template <bool SELF, int MODE, int AK>
static void launch_tm_i64_one(dim3 grid, size_t lds, hipStream_t st, const float* gin, const int64_t* nn_idx, const uint8_t* argmax,
                              float* gx, float* gsrc, int B, int G, int c, int N, int M, int k, int CW, int shmax) {
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mr_bwd_tm_scatter_i64_kernel<SELF, MODE, AK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((mr_bwd_tm_scatter_i64_kernel<SELF, MODE, AK>), grid, dim3(256), lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k, CW, shmax);
}
static void launch_tm_i64(bool self, int mode, int ak, dim3 grid, size_t lds, hipStream_t st, const float* gin, const int64_t* nn_idx,
                          const uint8_t* argmax, float* gx, float* gsrc, int B, int G, int c, int N, int M, int k, int CW, int shmax) {
#define GKG_TM64_CASE(S, MO, A) launch_tm_i64_one<S, MO, A>(grid, lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k, CW, shmax)
  if (self) {
    if (mode == 0) { if (ak) GKG_TM64_CASE(true, 0, 1); else GKG_TM64_CASE(true, 0, 0); }
    else { if (ak) GKG_TM64_CASE(true, 1, 1); else GKG_TM64_CASE(true, 1, 0); }
  } else {
    if (mode == 0) { if (ak) GKG_TM64_CASE(false, 0, 1); else GKG_TM64_CASE(false, 0, 0); }
    else { if (ak) GKG_TM64_CASE(false, 1, 1); else GKG_TM64_CASE(false, 1, 0); }
  }
#undef GKG_TM64_CASE
}

template <int NT, bool SELF, int MODE, int AK, int ACC, int U>
static void launch_tm_stream_one(dim3 grid, size_t lds, hipStream_t st, const float* gin, const int64_t* nn_idx, const uint8_t* argmax,
                                 float* gx, float* gsrc, int B, int G, int c, int N, int M, int k, int CW, int shmax) {
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mr_bwd_tm_stream_kernel<NT, SELF, MODE, AK, ACC, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((mr_bwd_tm_stream_kernel<NT, SELF, MODE, AK, ACC, U>), grid, dim3(NT), lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k, CW, shmax & 0xffff, shmax >> 16);
}
template <int NT, int ACC, int U>
static void launch_tm_stream_nt(bool self, int mode, int ak, dim3 grid, size_t lds, hipStream_t st, const float* gin, const int64_t* nn_idx,
                                const uint8_t* argmax, float* gx, float* gsrc, int B, int G, int c, int N, int M, int k, int CW, int shmax) {
#define GKG_TMS_CASE(S, MO, A) launch_tm_stream_one<NT, S, MO, A, ACC, U>(grid, lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k, CW, shmax)
  if (self) {
    if (mode == 0) { if (ak) GKG_TMS_CASE(true, 0, 1); else GKG_TMS_CASE(true, 0, 0); }
    else { if (ak) GKG_TMS_CASE(true, 1, 1); else GKG_TMS_CASE(true, 1, 0); }
  } else {
    if (mode == 0) { if (ak) GKG_TMS_CASE(false, 0, 1); else GKG_TMS_CASE(false, 0, 0); }
    else { if (ak) GKG_TMS_CASE(false, 1, 1); else GKG_TMS_CASE(false, 1, 0); }
  }
#undef GKG_TMS_CASE
}

extern "C" int gkg_mr_bwd_tm(const float* gin, const int64_t* nn_idx, const uint8_t* argmax, float* gx, float* gsrc,
                             int B, int G, int c, int N, int M, int k, int mode, int arg_kind, unsigned flags, void* stream) {
  if (arg_kind != 0 && arg_kind != 1) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_bwd_tm: arg_kind is 0 or 1");
  // arg_kind 1: the saved winning ROWS are the scatter targets — the index tensor is not read and may be null (the fused
  // k-NN + aggregation forward, gkg_knn_mr_fwd_tm, never materialises one)
  if (!gin || (!nn_idx && arg_kind != 1) || !argmax || !gx) return gkg_fail(GKG_ERR_NULL, "gkg_mr_bwd_tm: gin, argmax, gx (and nn_idx unless arg_kind == 1) must be non-null");
  if (B <= 0 || G <= 0 || c <= 0 || N <= 0 || M <= 0 || k <= 0 || k > 255 || (c & 3)) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_bwd_tm: bad sizes");
  if (mode != 0 && mode != 1) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_bwd_tm: mode is 0 or 1");
  if (mode == 1 && ((G * c) & 15)) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_bwd_tm: mode 1 needs C % 16 == 0");
  if (!gsrc && M != N) return gkg_fail(GKG_ERR_SHAPE, "gkg_mr_bwd_tm: self graph needs M == N");
  hipStream_t st = (hipStream_t)stream;
  // algorithmic bytes (SURVEY §8d "MR bwd"): g + int64 indices + argmax + gx (+ gsrc); arg_kind 1: u16 winning rows, no indices
  const double work = 4.0 * B * (double)G * c * N + (arg_kind == 1 ? 0.0 : 8.0 * B * (double)G * N * k) + (arg_kind == 1 ? 2.0 : 1.0) * B * (double)G * c * N
                      + 4.0 * B * (double)G * c * N + (gsrc ? 4.0 * B * (double)G * c * M : 0.0);
  GkgProfScope prof(GKG_PROF_MR_BWD, st, work);
  const int C = G * c;
  const size_t T = (size_t)B * N;
  const size_t total = T * (C / 4);
  // Exact integer accumulation (see mr_bwd_tm_scatter_i64_kernel): the default whenever an (image, channel chunk) of 64-bit
  // accumulators fits the LDS budget — order-independent, so it is also the deterministic form.  GKG_MR_FP32_ATOMICS keeps
  // the fp32-atomic kernels (measurement, tests).
  // Where it is selected (MI355X, tools/bench_mr_bwd.py, us, exact vs fp32 atomics at their best chunk widths): 18 x 18
  // images — cfg2 21.2 vs 24.4-27.6, its label graph 9.4 vs 10.9, C = 640 39.3 vs 44.9-51.5; a tie at 36 x 36 (116.7 vs
  // 113.5-118.2); it LOSES where the gradients stream from HBM and are swept twice (pooled 1 296-key images under 5 184 /
  // 20 736 queries: 209 vs 164, 492 vs 425).  Rule: destination images of up to 512 rows; with GKG_MR_DETERMINISTIC wherever it
  // fits (the private-accumulator form it replaces there is 1.6x the fp32 time).  Chunk width 8 (measured best or tied at
  // every shape: the kernel lives on workgroups per CU), 4 when 8 does not fit.
  // Streaming fixed-point form (mr_bwd_tm_stream_kernel, round 5): the default from 160 query rows per image wherever an (image,
  // channel chunk) of 64-bit accumulators fits the LDS budget — one sweep, 4 / 8 rows per thread in flight, 512 threads.  Order-
  // independent like the two-sweep form, so it serves GKG_MR_DETERMINISTIC too.  Measured against what it replaces
  // (tools/bench_mr_bwd.py, us, profiles/r05_bench_mr_bwd_stream.txt): cfg2 21.8 -> 16.6, C = 640 at 18 x 18 38.7 -> 25.1,
  // 36 x 36 self graph 117.6 -> 89.9, pooled 1 296-key images under 5 184 / 20 736 queries 165 -> 133 / 424 -> 359; the label
  // graphs (80 queries) keep the two-sweep kernel (8.6 vs 9.2-9.9).  Chunk width: the widest of 16 / 8 / 4 channels that fits.
  // Measurement: flags bits 16..17 = 1 force this form (bits 8..15 chunk width, 20..21 = 2: 256 threads, bit 22: 8 rows in
  // flight), 2 = its fp64-atomic variant (ds_add_f64: 45-98 cycles per wave instruction against 170-183 for ds_add_f32 —
  // tools/ubench/lds_atomic_rate.hip), 3 = the two-sweep kernel wherever it fits.
  const int sv = (int)((flags >> 16) & 3);
  if (N < (1 << 24) && (sv == 1 || sv == 2 || (sv == 0 && !(flags & GKG_MR_FP32_ATOMICS) && N >= 160))) {
    int CW = (int)((flags >> 8) & 0xff);
    auto sfits = [&](int cw) { return (size_t)M * cw * 8 + 16 <= (size_t)MR_LDS_BUDGET && C % cw == 0 && c % cw == 0; };
    if (!CW || !sv) CW = sfits(16) ? 16 : (sfits(8) ? 8 : 4);
    if ((sv && (size_t)M * CW * 8 + 16 <= 150 * 1024 && C % CW == 0 && c % CW == 0) || sfits(CW)) {
      // long sweeps over a pooled key image (GKGNet-576 stages 1 / 2: N = 16 M / 4 M, M = 1 296): 256-thread workgroups with 4 rows
      // in flight — two workgroups per CU overlap one's store phase with the other's sweep (round 6, on the XM gradient layout,
      // us: stage 1 379.6 -> 334.9, stage 2 145.7 -> 141.0; the self graphs keep 512 threads: 36 x 36 89.7 vs 93.5)
      const bool long_sweep = M > 512 && (long)N >= 4L * M;
      const bool nt256 = sv ? ((flags >> 20) & 3) == 2 : long_sweep;
      const bool u8 = sv ? ((flags >> 22) & 1) != 0 : (M > 512 && !long_sweep);       // 8 rows in flight
      const size_t lds = (size_t)M * CW * 8 + 16;
      int bitsN = 0;
      while ((1 << bitsN) <= N) ++bitsN;
      const dim3 grid((C / CW) * ((B + 7) / 8) * 8);
      const bool self = gsrc == nullptr;
      const int planes_bit = (sv && ((flags >> 23) & 1) && arg_kind == 1) ? (1 << 16) : 0;      // measurement: 8-channel planes of winning rows
      if (sv == 2) launch_tm_stream_nt<512, 1, 4>(self, mode, arg_kind, grid, lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k, CW, (38 - bitsN) | planes_bit);
      else if (nt256 && u8) launch_tm_stream_nt<256, 0, 8>(self, mode, arg_kind, grid, lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k, CW, (38 - bitsN) | planes_bit);
      else if (nt256) launch_tm_stream_nt<256, 0, 4>(self, mode, arg_kind, grid, lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k, CW, (38 - bitsN) | planes_bit);
      else if (u8) launch_tm_stream_nt<512, 0, 8>(self, mode, arg_kind, grid, lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k, CW, (38 - bitsN) | planes_bit);
      else launch_tm_stream_nt<512, 0, 4>(self, mode, arg_kind, grid, lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k, CW, (38 - bitsN) | planes_bit);
      hipError_t es = hipGetLastError();
      return es == hipSuccess ? 0 : gkg_fail_hip(es, "mr_bwd_tm_stream_kernel");
    }
    if (sv) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_bwd_tm: forced streaming form does not fit");
  }
  if (!(flags & GKG_MR_FP32_ATOMICS) && N < (1 << 24) && (M <= 512 || sv == 3 || (flags & GKG_MR_DETERMINISTIC))) {
    int CW = 8;
    const int forced = (int)((flags >> 8) & 0xff);          // measurement only: bits 8..15 force the chunk width
    auto fits = [&](int cw, size_t budget) { return (size_t)M * cw * 8 + 16 <= budget && C % cw == 0 && c % cw == 0; };
    if (!fits(CW, (size_t)MR_LDS_BUDGET)) CW = 4;
    if (forced) CW = forced;
    if (fits(CW, (size_t)MR_LDS_BUDGET) && (long)(C / CW) * (B + 7) < 0x7fffffffL) {
      int bitsN = 0;
      while ((1 << bitsN) <= N) ++bitsN;                  // N < 2^bitsN
      launch_tm_i64(gsrc == nullptr, mode, arg_kind, dim3((C / CW) * ((B + 7) / 8) * 8), (size_t)M * CW * 8 + 16, st, gin, nn_idx,
                    argmax, gx, gsrc, B, G, c, N, M, k, CW, 38 - bitsN);
      hipError_t ei = hipGetLastError();
      return ei == hipSuccess ? 0 : gkg_fail_hip(ei, "mr_bwd_tm_scatter_i64_kernel");
    }
  }
  if (flags & GKG_MR_DETERMINISTIC) {
    if (B > 65535) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_bwd_tm: B <= 65535");
    long TL = (long)(144 * 1024) / ((long)M * 16);
    if (TL > 64) TL = 64;
    if (TL >= 1) {
      const size_t lds = (size_t)TL * M * 16;
      launch_tm_lds<true>(gsrc == nullptr, mode, arg_kind, dim3(C / 4, B), dim3(64), lds, st, gin, nn_idx, argmax, gx, gsrc, B, G, c, N, M, k,
                          (int)TL);
    } else {                         // seed (gx = direct - gm; gsrc = 0), then the ordered global walk
      hipLaunchKernelGGL(mr_bwd_tm_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gin, gx, C, T, mode);
      if (gsrc) (void)hipMemsetAsync(gsrc, 0, sizeof(float) * (size_t)B * M * C, st);
      hipLaunchKernelGGL(mr_bwd_tm_det_global_kernel, dim3((unsigned)((B * (C / 4) + 63) / 64)), dim3(64), 0, st, gin, nn_idx,
                         argmax, gsrc ? gsrc : gx, B, G, c, N, M, k, mode, arg_kind);
    }
    hipError_t ed = hipGetLastError();
    return ed == hipSuccess ? 0 : gkg_fail_hip(ed, "mr_bwd_tm (deterministic)");
  }
  // channel chunk: largest power of two (4..64) dividing C and c whose [M][CW] fp32 image fits the LDS budget, shrunk
  // until the grid has ~2 workgroups per CU
  int CW = 64;
  while (CW > 4 && ((size_t)M * CW * 4 > (size_t)MR_LDS_BUDGET || C % CW || c % CW)) CW >>= 1;
  while (CW > 8 && (long)(C / CW) * B < 512) CW >>= 1;
  if ((flags >> 8) & 0xff) CW = (int)((flags >> 8) & 0xff);       // measurement only
  if ((size_t)M * CW * 4 <= (size_t)MR_LDS_BUDGET && C % CW == 0 && c % CW == 0 && (long)(C / CW) * (B + 7) < 0x7fffffffL) {
    const size_t lds = (size_t)M * CW * 4;
    launch_tm_lds<false>(gsrc == nullptr, mode, arg_kind, dim3((C / CW) * ((B + 7) / 8) * 8), dim3(256), lds, st, gin, nn_idx, argmax, gx, gsrc,
                         B, G, c, N, M, k, CW);
  } else {                           // destination image too large for LDS: elementwise seed + fp32 global atomics
    hipLaunchKernelGGL(mr_bwd_tm_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gin, gx, C, T, mode);
    if (gsrc) (void)hipMemsetAsync(gsrc, 0, sizeof(float) * (size_t)B * M * C, st);
    hipLaunchKernelGGL(mr_bwd_tm_scatter_atomic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gin, nn_idx,
                       argmax, gsrc ? gsrc : gx, B, G, c, N, M, k, mode, arg_kind);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "mr_bwd_tm");
}
