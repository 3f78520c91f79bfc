// gkg_knn_pf.hip — k-NN graph construction, prefilter form: approximate distance tiles on the bf16 matrix cores, exact
// contract distances only for the candidates that can matter.  Same results as knn_tile_kernel (gkg_knn.hip), bit for bit;
// selected by the host side where it is faster (long key streams).  Reference: torch_edge.py:9-51, 54-106, 139-176.
#include "gkg_knn_common.h"
#include "gkg_topk_merge.h"

// -DPF_ABL=<bits> (tools/ubench/pf_ablate.py only; results wrong, timing valid): 1 no candidate appends (tested, never stored),
// 2 no MFMAs, 4 no relative_pos loads, 8 no selection at all (the accumulators are summed into a sink), 16 no key-operand loads
#ifndef PF_ABL
#define PF_ABL 0
#endif

namespace gkg {

// ------------------------------------------------------------------------------------------ prefilter + exact re-rank
// Same contract, same bits, a third of the matrix time: the (64-query x 32-key) distance tiles are first evaluated
// APPROXIMATELY on the bf16 matrix cores — every normalised fp32 token split into its two leading bf16 terms (hi, lo),
// three of the four cross products (hi.hi + hi.lo + lo.hi), fp32 accumulation, relative_pos pre-loaded into the
// accumulators: 3 x 32 cycles per 32x32x16 block where the fp32 MFMA needs 8 x 64, and unlike the fp32 MFMA it runs
// BESIDE the selection's vector work instead of on its datapath — and only the few candidates that can be among a query's
// k*d nearest are then re-evaluated EXACTLY (the contract's ordered fmaf chain, bit for bit) and ranked.
//
// Why the result is the contract's: let d_e(m) be the contract distance of key m and d_a(m) the prefilter's, with
// |d_a - d_e| <= eps for every pair (eps below).  Let tau = the KD-th smallest d_a.  The KD keys with the smallest d_a all
// have d_e <= tau + eps, so the KD-th smallest d_e is <= tau + eps, so every key of the true top-KD has
// d_a <= d_e + eps <= tau + 2 eps.  The survivors S = {m : d_a(m) <= tau + 2 eps} therefore contain the true top-KD, and
// ranking S by the exact (d_e, m) keys gives exactly the contract's list (ties included).
//   eps: with |x^| = |y^| = 1 (normalised tokens; the kernel is only selected with GKG_KNN_NORMALIZE) sum_ch |2 x^ y^| <= 2.
//   Dropped terms of the split (lo.lo and the residuals |v - hi - lo| <= 2^-18 |v|): <= 3 * 2^-18 * 2 = 2.3e-5; fp32
//   accumulation of the 3 c products on the matrix core, any order, on partial sums <= |bias| + 2 <= 3.125 (the bias is the
//   accumulators' initial value: hence the |relative_pos| <= 1.125 precondition, GKG_KNN_RELPOS_UNIT): <= 3 c u * 3.125,
//   u = 2^-24; the contract's own chain against the real value: <= c u * 2; the final adds: a few u.
//   eps = 3e-5 + 7e-7 * cpad (11.4 c u = 6.8e-7 c) covers the sum.
// Per-wave lists: the 4 waves of a workgroup stream disjoint key tiles, so a wave may have dropped (beyond its KDW-entry
// list) a key that belongs to S.  That can only have happened if the wave's own KDW-th entry is <= tau + 2 eps.  A query
// tile with such a query, or with a query that has more than SMAX survivors (masses of exact or near ties: duplicated
// tokens, degenerate features), is NOT settled here: the workgroup sets its flag, writes nothing, and the host side's
// clean-up launch — the fp32 tile kernel restricted to the flagged workgroups — computes the tile the plain way.  On
// ordinary data no tile is flagged (KDW exceeds KD for short lists, 9 -> 12, 16 -> 18, so that the first condition needs
// >= KDW of a query's best keys in ONE wave's quarter of the key tiles: ~4^(1-KDW) per query) and the clean-up launch is
// ~1 300 workgroups that exit on their first instruction; on degenerate data the cost is bounded by prefilter + plain
// kernel.  (An in-lane exact re-scan of all M keys was the first form of this fallback: one degenerate block of the
// random-init GKGNet-576 train step took 6.9 ms instead of 0.45.)
constexpr int PF_EXTRA = 8;       // survivors beyond KD a query may have before it takes the slow path

// The contract's distance (without relative_pos) of query n and key m from the fp32 channel-major normalised copies: the
// ordered fmaf chain over the cpad (zero-padded) channels with the query pre-scaled by -2 — what knn_tile_kernel's fp32
// MFMA contraction computes, bit for bit.
__device__ __forceinline__ float pf_exact_dist(const float* __restrict__ xc, int N, const float* __restrict__ yc, int M,
                                               int cpad, float sqx, float sqy) {
  float acc = 0.0f;
#pragma unroll 8
  for (int ch = 0; ch < cpad; ++ch) acc = __builtin_fmaf(yc[(size_t)ch * M], -2.0f * xc[(size_t)ch * N], acc);
  return (sqx + acc) + sqy;                        // the contract's order; relative_pos is added by the caller
}

// PBUF — buffered selection as in knn_tile_kernel: a candidate is only tested against the lane's (possibly stale) KDW-th
// distance and, when it passes, appended to a per-lane LDS buffer of PBUF entries behind the staged queries; the lists are
// updated in wave-uniform flushes (merge network for lists of 16+ entries, sorted inserts below).  0: every candidate goes
// through the sorted insert (2 KDW + 6 instructions).
// SB — the whole contraction is ONE batch of key operands (cp16 <= 64: narrow groups, pvig_s stage 1, pvig_m): the MFMAs read
// the prefetched batch in place and the next tile's batch is requested behind them, instead of copying the batch aside and
// prefetching at once — 32 registers less, which pays for fetching relative_pos one tile AHEAD at three waves per SIMD.
// Measured before (tools/ubench/knn_timeline.py, pvig_s stage 1): every tile spent 4-5 k of its 10 k cycles between its top
// and its first MFMA result, waiting for the bias it had just requested; 1 670 -> 1 500 us.
template <int KD, int KDW, bool HAS_RP, int PBUF = 0, bool SB = false>
__global__ __launch_bounds__(256, KDW <= 16 ? 3 : 2) void knn_pf_kernel(KnnArgs a) {
  extern __shared__ float smem[];
  typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
  constexpr int SMAX = KD + PF_EXTRA;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lin = blockIdx.x;
  int bg, qt;
  if (!knn_map(a, lin, bg, qt)) return;              // XCD-aware map shared with knn_tile_kernel (gkg_knn_common.h)
  KNN_TL(0);
  const int n0 = qt * QT;
  const int N = a.N, M = a.M, cpad = a.cpad, cp16 = a.cp16, S16 = cp16 >> 4;
  const int lane_n = n0 + lane;
  const int nc = lane_n < N ? lane_n : N - 1;
  const int kk = lane >> 5, l31 = lane & 31;
  constexpr int KB = 4;                              // k16-steps per key-operand batch
  // key tiles are visited centre-out from the tile under the query tile's own image position when there is a positional
  // bias (see knn_tile_kernel): lists fill with near-final entries first; the result does not depend on the order
  const int ktiles = (M + KT - 1) / KT;
  int c0 = 0;
  if (HAS_RP) c0 = min(max((int)(((long long)(n0 + QT / 2) * M / N) / KT), 0), ktiles - 1);
  const int nleft = c0, nright = ktiles - 1 - c0, nboth = min(nleft, nright);
  auto tile_at = [&](int i) -> int {                 // i-th visited tile, 0 <= i < ktiles
    if (!HAS_RP) return i;
    if (i <= 2 * nboth) return (i & 1) ? c0 + ((i + 1) >> 1) : c0 - (i >> 1);
    return nleft >= nright ? c0 - (i - nboth) : c0 + (i - nboth);
  };
  const int t_first = tile_at(min(w, ktiles - 1));
  // octet-major planes with MR = M rounded up to 32 rows (round 5): the pad rows carry MASKED_SQ in the |y|^2 channel, so the
  // keys past M of the last tile mask themselves and no row index is clamped
  const int MR = a.pf_mrows, NR = a.pf_nrows;
  const uint4* yhp = reinterpret_cast<const uint4*>(a.yb) + (size_t)bg * (cp16 >> 3) * MR;
  const uint4* ylp = reinterpret_cast<const uint4*>(a.yb_lo) + (size_t)bg * (cp16 >> 3) * MR;
  uint4 bh_[KB], bl_[KB];
  {
    const int mk0 = t_first * KT + l31;
    const uint4* h0 = yhp + (size_t)kk * MR + mk0;
    const uint4* l0 = ylp + (size_t)kk * MR + mk0;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      bh_[u] = u < S16 ? h0[(size_t)(2 * u) * MR] : make_uint4(0, 0, 0, 0);
      bl_[u] = u < S16 ? l0[(size_t)(2 * u) * MR] : make_uint4(0, 0, 0, 0);
    }
  }
  // ---- stage the query tile: hi and lo planes scaled by -2 (exact), rows of cp16 + 8 bf16
  const int qpitch = (cp16 + 8) * 2;                 // bytes
  char* xq_hi = reinterpret_cast<char*>(smem);
  char* xq_lo = xq_hi + QT * qpitch;
  {
    const uint4* xhp = reinterpret_cast<const uint4*>(a.xb) + (size_t)bg * (cp16 >> 3) * NR;
    const uint4* xlp = reinterpret_cast<const uint4*>(a.xb_lo) + (size_t)bg * (cp16 >> 3) * NR;
    const int qc = a.pf_c;                          // channels qc, qc + 1: the keys' |y|^2 terms; a QUERY stages (1, 1) there (hi), (0, 0) (lo)
    const int chunks = cp16 >> 3;
    auto m2h = [](unsigned hv) -> unsigned {
      const unsigned e = hv & 0x7f80u;
      if (e == 0u) return 0u;                                   // zero / denormal
      if (e == 0x7f80u) return hv ^ 0x8000u;                     // inf / NaN keep their class
      if (e == 0x7f00u) return ((hv ^ 0x8000u) & 0x8000u) | 0x7f80u;   // overflow -> inf
      return (hv + 0x80u) ^ 0x8000u;
    };
    auto m2 = [&](unsigned wv) { return m2h(wv & 0xffffu) | (m2h(wv >> 16) << 16); };
    for (int i = tid; i < 2 * QT * chunks; i += 256) {
      const int pl = i >= QT * chunks;
      const int r = i - pl * QT * chunks;
      const int ck = r >> 6, q = r & 63;           // consecutive threads: consecutive queries of one octet (coalesced)
      uint4 v = make_uint4(0, 0, 0, 0);
      if (n0 + q < N) v = (pl ? xlp : xhp)[(size_t)ck * NR + n0 + q];
      v.x = m2(v.x); v.y = m2(v.y); v.z = m2(v.z); v.w = m2(v.w);
      {
        unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int r8 = qc + e - 8 * ck;           // position of channel qc + e inside this octet
          if (r8 >= 0 && r8 < 8) {
            const unsigned one = pl ? 0u : 0x3f80u;
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if ((r8 >> 1) == u) wv[u] = (r8 & 1) ? ((wv[u] & 0x0000ffffu) | (one << 16)) : ((wv[u] & 0xffff0000u) | one);
          }
        }
        v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
      }
      *reinterpret_cast<uint4*>((pl ? xq_lo : xq_hi) + q * qpitch + 16 * ck) = v;
    }
  }
  if (PBUF > 0) reinterpret_cast<float*>(xq_lo + QT * qpitch + (size_t)PBUF * 256 * 8)[tid] = INFINITY;   // shared admission bounds (see flush)
  __syncthreads();
  KNN_TL(2);

  const float* sqy = a.sqy + (size_t)bg * M;
  const bool two_blocks = n0 + 32 < N;               // wave-uniform
  TopList<KDW> top;
  top.init();
  float2* cbuf = reinterpret_cast<float2*>(xq_lo + QT * qpitch) + tid;      // [PBUF][256] behind the staged queries
  // the lane's append cursor IS its entry count (knn_tile_kernel's form): an admitted candidate costs the store + one add
  typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) v2u_t lds_v2u_t;
  unsigned cw0 = (unsigned)(size_t)(lds_v2u_t*)cbuf;
  asm volatile("" : "+v"(cw0));
  unsigned cw_lim = cw0 + (PBUF > 8 ? PBUF - 8 : 0) * 256 * 8;
  asm volatile("" : "+v"(cw_lim));
  unsigned cw = cw0;
  float thr = INFINITY;
  constexpr bool PSHARE = PBUF > 0;
  constexpr int QSH = (KD + NW - 1) / NW;
  float* ths = reinterpret_cast<float*>(xq_lo + QT * qpitch + (size_t)PBUF * 256 * 8);      // [NW][64] behind the candidate buffer (set to +inf before the staging barrier)
  const float margin_s = a.margin;
  auto flush = [&]() {
    unsigned cw_now = cw;
    asm volatile("" : "+v"(cw_now));
    int bcnt = (int)((cw_now - cw0) / (256 * 8));
#if PF_ABL & 64
    {                                               // tools/ubench/pf_ablate.py counters: flushes, entries, insert rounds (max over lanes)
      unsigned long long* ctr = reinterpret_cast<unsigned long long*>(a.part_v);
      if (lane == 0) atomicAdd(&ctr[0], 1ull);
      atomicAdd(&ctr[1], (unsigned long long)bcnt);
      int mx = bcnt;
      for (int m_ = 1; m_ < 64; m_ <<= 1) mx = max(mx, __shfl_xor(mx, m_, 64));
      if (lane == 0) atomicAdd(&ctr[2], (unsigned long long)mx);
    }
#endif
    if constexpr (PBUF >= 12 && KDW >= 16) {
      if (__builtin_amdgcn_ballot_w64(bcnt > 3) != 0ull) {
        double b[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i < PBUF) {
            const float2 e = cbuf[i * 256];
            b[i] = i < bcnt ? pack_key(e.x, __float_as_int(e.y)) : (double)INFINITY;
          } else {
            b[i] = (double)INFINITY;
          }
        }
        TopMerge16<KDW>::run(top.key, b);
        bcnt = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < (PBUF > 0 ? PBUF : 1); ++i) {   // forward branches only (see knn_tile_kernel)
      if (__builtin_amdgcn_ballot_w64(i < bcnt) == 0ull) break;
      const float2 e = cbuf[i * 256];
      const double k = i < bcnt ? pack_key(e.x, __float_as_int(e.y)) : (double)INFINITY;
      top.template insert_key<false>(k);
    }
    cw = cw0;
    thr = key_dist(top.key[KDW - 1]);
    if constexpr (PSHARE) {
      // shared admission bound (round 5; knn_tile_kernel's SHARE carried over): the Q = ceil(KD / 4) best entries of the four
      // waves' lists are >= KD distinct keys, so the query's final KD-th prefilter distance tau is <= sh = the largest of the
      // waves' Q-th entries.  A candidate beyond sh + margin is no survivor (S = {d <= tau + margin}) and cannot be one of the
      // KD best: it never needs to enter a list.  Published / read without synchronisation: a stale value is an older, larger
      // one — still a bound.  What it buys: the appends and the flushes behind them are a third of this launch at pvig_s
      // stage 1 (tools/ubench/pf_ablate.py: 1 915 -> 1 330 us without them); a wave's own KDW-th entry over its quarter of the
      // keys admits ~4 x 50 candidates per query, the shared bound ~70.
      const float mine = key_dist(top.key[QSH - 1]);              // +inf while the list holds fewer than Q entries
      ths[w * 64 + lane] = mine;
      float m = mine;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) m = fmaxf(m, ths[ww * 64 + lane]);
      if (m < INFINITY) {
        const float b = m + margin_s;
        thr = fminf(thr, __int_as_float(__float_as_int(b) + (b >= 0.0f ? 1 : -1)));       // rounded away: never tighter than sh + margin
      }
    }
  };
  // relative_pos rows of the two query blocks this lane's accumulator columns belong to
  const int nq0 = min(n0 + l31, N - 1), nq1 = min(n0 + 32 + l31, N - 1);
  // relative_pos rows (the accumulators' initial value) and |y|^2, as 16-byte buffer loads over the workgroup's 64 rows.
  // AHEAD: those of the NEXT tile are fetched during this tile's selection phase (see knn_tile_kernel's bf16 form) — only
  // where the 32 extra live registers are free (lists >= 18: already 2 waves per SIMD; measured 438 -> 416 us at pvig_s
  // stage 3).  With 12-entry lists they cost the third wave per SIMD and the kernel loses (stage 1 2230 -> 2450 us).
  constexpr bool AHEAD = KDW > 16 || SB;
  float4 rq0[4], rq1[4];
#if PF_ABL & 4
  for (int g_ = 0; g_ < 4; ++g_) { rq0[g_] = make_float4(0.f, 0.f, 0.f, 0.f); rq1[g_] = make_float4(0.f, 0.f, 0.f, 0.f); }
#endif
#if PF_ABL & (1 | 8)
  float abl_sink = INFINITY;
#endif
  const size_t rp_row0 = (size_t)min(n0, N - 1) * M;
  const size_t rp_left = ((size_t)N * M - rp_row0) * sizeof(float);
  const __amdgpu_buffer_rsrc_t rp_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(HAS_RP ? a.relpos + rp_row0 : a.sqy), 0, (int)(rp_left > 0x7fffffffull ? 0x7fffffffull : rp_left), 0x00020000);
  const unsigned rp_o0 = (unsigned)(((size_t)l31 * M + 4 * kk) * sizeof(float));
  const unsigned rp_o1 = (unsigned)(((size_t)(32 + l31) * M + 4 * kk) * sizeof(float));
  auto fetch_side = [&](int tt) __attribute__((always_inline)) {
    const int mm0 = tt * KT;
    if (HAS_RP && !(PF_ABL & 4)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        rq0[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rp_rsrc, (int)(rp_o0 + 32 * g), mm0 * 4, 0));
        rq1[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rp_rsrc, (int)(rp_o1 + 32 * g), mm0 * 4, 0));
      }
    }
  };
  if (AHEAD && w < ktiles) fetch_side(t_first);
  for (int iv = w; iv < ktiles; iv += NW) {
    const int t = tile_at(iv);
    const int t_next = iv + NW < ktiles ? tile_at(iv + NW) : t;
    const int m0 = t * KT;
    const int mk = m0 + l31;
    const int mk_next = t_next * KT + l31;
    if (!AHEAD) fetch_side(t);
    // accumulators start from relative_pos: lane (l31, kk), register 4 g + j <-> key row m0 + 8 g + 4 kk + j
    f32x16 acc0, acc1;
    if (HAS_RP) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        acc0[4 * g] = rq0[g].x; acc0[4 * g + 1] = rq0[g].y; acc0[4 * g + 2] = rq0[g].z; acc0[4 * g + 3] = rq0[g].w;
        acc1[4 * g] = rq1[g].x; acc1[4 * g + 1] = rq1[g].y; acc1[4 * g + 2] = rq1[g].z; acc1[4 * g + 3] = rq1[g].w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    }
    {
      const uint4* ykh = yhp + (size_t)kk * MR + mk;
      const uint4* ykl = ylp + (size_t)kk * MR + mk;
      const uint4* ynh = yhp + (size_t)kk * MR + mk_next;
      const uint4* ynl = ylp + (size_t)kk * MR + mk_next;
      const char* xh0 = xq_hi + l31 * qpitch + 16 * kk;
      const char* xl0 = xq_lo + l31 * qpitch + 16 * kk;
      const char* xh1 = xh0 + 32 * qpitch;
      const char* xl1 = xl0 + 32 * qpitch;
      if constexpr (SB) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          if (u < S16) {
            const bf16x8_t kh = __builtin_bit_cast(bf16x8_t, bh_[u]);
            const bf16x8_t kl = __builtin_bit_cast(bf16x8_t, bl_[u]);
            const bf16x8_t qh0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xh0 + 32 * u));
            const bf16x8_t ql0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xl0 + 32 * u));
#if PF_ABL & 2
            acc0[u] += __uint_as_float(bh_[u].x ^ bl_[u].y) + __builtin_bit_cast(float, (unsigned)(qh0[0] != ql0[1]));
            acc1[u] += __uint_as_float(bh_[u].z ^ bl_[u].w);
#else
            if (two_blocks) {
              const bf16x8_t qh1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xh1 + 32 * u));
              const bf16x8_t ql1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xl1 + 32 * u));
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh0, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh1, acc1, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql0, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql1, acc1, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh0, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh1, acc1, 0, 0, 0);
            } else {
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh0, acc0, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql0, acc0, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh0, acc0, 0, 0, 0);
            }
#endif
          }
        }
        __builtin_amdgcn_sched_barrier(0);                        // the next tile's batch: behind the MFMAs that read this one
#if !(PF_ABL & 16)
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          bh_[u] = u < S16 ? ynh[(size_t)(2 * u) * MR] : make_uint4(0, 0, 0, 0);
          bl_[u] = u < S16 ? ynl[(size_t)(2 * u) * MR] : make_uint4(0, 0, 0, 0);
        }
#endif
      } else
      for (int s0 = 0; s0 < S16; s0 += KB) {
        const bool last = s0 + KB >= S16;                         // uniform: prefetch the NEXT tile's first batch
        uint4 ah[KB], al[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) { ah[u] = bh_[u]; al[u] = bl_[u]; }
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          const int sn = last ? u : s0 + KB + u;
          bh_[u] = sn < S16 ? (last ? ynh : ykh)[(size_t)(2 * sn) * MR] : make_uint4(0, 0, 0, 0);
          bl_[u] = sn < S16 ? (last ? ynl : ykl)[(size_t)(2 * sn) * MR] : make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          if (s0 + u < S16) {
            const bf16x8_t kh = __builtin_bit_cast(bf16x8_t, ah[u]);
            const bf16x8_t kl = __builtin_bit_cast(bf16x8_t, al[u]);
            const bf16x8_t qh0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xh0 + 32 * (s0 + u)));
            const bf16x8_t ql0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xl0 + 32 * (s0 + u)));
            if (two_blocks) {                                      // the two accumulator chains alternate: no MFMA
              const bf16x8_t qh1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xh1 + 32 * (s0 + u)));
              const bf16x8_t ql1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xl1 + 32 * (s0 + u)));
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh0, acc0, 0, 0, 0);     // issues right behind the one it
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh1, acc1, 0, 0, 0);     // depends on (small terms first)
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql0, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql1, acc1, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh0, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh1, acc1, 0, 0, 0);
            } else {
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh0, acc0, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql0, acc0, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh0, acc0, 0, 0, 0);
            }
          }
        }
      }
    }
    if (iv / NW < 10) KNN_TL(3 + 2 * (iv / NW));
    if (AHEAD && iv + NW < ktiles) fetch_side(t_next); // in flight during the selection below
    // ---- lane l <- all 32 keys of query n0 + l (permlane swap as in knn_tile_kernel); approximate distance (without the
    //      query's own |x|^2, a per-query constant) = acc + |y|^2, keys past M masked by MASKED_SQ
#if PF_ABL & 8
    for (int r_ = 0; r_ < 16; ++r_) abl_sink = fminf(abl_sink, acc0[r_] + acc1[r_]);
    if (iv + NW >= ktiles) top.template insert<false>(abl_sink, 0);
#else
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float lo[4], hi[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0[4 * g + j]),
                                                         __float_as_uint(acc1[4 * g + j]), false, false);
        lo[j] = __uint_as_float(sw[0]);
        hi[j] = __uint_as_float(sw[1]);
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = 8 * g + 4 * hh + j;
          const float d = hh ? hi[j] : lo[j];     // |y|^2 is already inside (channels pf_c, pf_c + 1 of the contraction)
          if constexpr (PBUF > 0) {
#if PF_ABL & 1
            abl_sink = fminf(abl_sink, d <= thr ? d : abl_sink);
#else
            if (d <= thr) {                       // '<=': the tiles are not visited in index order (NaN fails)
              *(lds_v2u_t*)(size_t)cw = v2u_t{__float_as_uint(d), (unsigned)(m0 + row)};
              asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(cw) : "i"(256 * 8) : "memory");
            }
#endif
          } else {
            top.template insert<KDW >= 18>(d, m0 + row);
          }
        }
      }
      if (PBUF > 0 && ((g == 3 && iv + NW >= ktiles) || __builtin_amdgcn_ballot_w64(cw > cw_lim) != 0ull)) flush();
    }
#if PF_ABL & 1
    if (iv + NW >= ktiles) top.template insert<false>(abl_sink, 0);
#endif
#endif
    if (iv / NW < 10) KNN_TL(4 + 2 * (iv / NW));
  }

  // ---- the 4 per-wave lists -> LDS; wave 0 merges them per query and collects the survivors.  (A rank-counting merge on all
  //      four waves, as in knn_tile_kernel, measured the same launch time here: it trades wave 0's LDS latency, which the
  //      CU's other workgroups fill, for vector work — and vector issue is what this kernel is short of.)
  KNN_TL(28);
  __syncthreads();                       // everyone is done with the staged queries
  KNN_TL(29);
  float* lv = smem;                      // [NW][KDW][64]
  int* li = reinterpret_cast<int*>(smem + NW * KDW * 64);
  int* sidx = li + NW * KDW * 64;        // [SMAX][64] survivor key indices (ascending prefilter distance)
  int* scnt = sidx + SMAX * 64;          // [64] survivor count; -1: tile flagged for the clean-up launch
  int* npairs = scnt + 64;               // [1] (+3 pad) number of (query, survivor) pairs that need the exact distance
  uint16_t* plist = reinterpret_cast<uint16_t*>(npairs + 4);     // [SMAX * 64] those pairs, (s << 6) | q
  double* keys = reinterpret_cast<double*>(smem);                // [SMAX][64] final sort keys, over the dead list area
#pragma unroll
  for (int j = 0; j < KDW; ++j) {
    lv[(w * KDW + j) * 64 + lane] = key_dist(top.key[j]);
    li[(w * KDW + j) * 64 + lane] = key_index(top.key[j]);
  }
  if (tid == 0) *npairs = 0;
  __syncthreads();
  const float margin = a.margin;
  if (w == 0) {
    int p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    float h0 = lv[(0 * KDW) * 64 + lane], h1 = lv[(1 * KDW) * 64 + lane], h2 = lv[(2 * KDW) * 64 + lane],
          h3 = lv[(3 * KDW) * 64 + lane];
    int i0 = li[(0 * KDW) * 64 + lane], i1 = li[(1 * KDW) * 64 + lane], i2 = li[(2 * KDW) * 64 + lane],
        i3 = li[(3 * KDW) * 64 + lane];
    float tau = INFINITY;
    int cnt = 0;
    bool open = true;                    // still collecting
    float sv_d[SMAX];                    // prefilter distances of the survivors (registers: the loops are unrolled)
#pragma unroll
    for (int j = 0; j < SMAX; ++j) {
      int sel = 0; float bv = h0; int bi = i0;
      if (h1 < bv || (h1 == bv && i1 < bi)) { sel = 1; bv = h1; bi = i1; }
      if (h2 < bv || (h2 == bv && i2 < bi)) { sel = 2; bv = h2; bi = i2; }
      if (h3 < bv || (h3 == bv && i3 < bi)) { sel = 3; bv = h3; bi = i3; }
      if (sel == 0) { ++p0; h0 = p0 < KDW ? lv[(0 * KDW + p0) * 64 + lane] : INFINITY; i0 = p0 < KDW ? li[(0 * KDW + p0) * 64 + lane] : 0x7fffffff; }
      else if (sel == 1) { ++p1; h1 = p1 < KDW ? lv[(1 * KDW + p1) * 64 + lane] : INFINITY; i1 = p1 < KDW ? li[(1 * KDW + p1) * 64 + lane] : 0x7fffffff; }
      else if (sel == 2) { ++p2; h2 = p2 < KDW ? lv[(2 * KDW + p2) * 64 + lane] : INFINITY; i2 = p2 < KDW ? li[(2 * KDW + p2) * 64 + lane] : 0x7fffffff; }
      else { ++p3; h3 = p3 < KDW ? lv[(3 * KDW + p3) * 64 + lane] : INFINITY; i3 = p3 < KDW ? li[(3 * KDW + p3) * 64 + lane] : 0x7fffffff; }
      if (j == KD - 1) tau = bv;
      // merged order is ascending, so the survivors are a PREFIX of it: the first candidate that is not taken — beyond the
      // margin, or not a real key (exhausted lists, keys masked past M) — ends the collection
      const bool take = open && (j < KD || bv <= tau + margin) && (unsigned)bi < (unsigned)M;
      if (!take) open = false;
      if (take) { sidx[j * 64 + lane] = bi; cnt = j + 1; }
      sv_d[j] = take ? bv : INFINITY;
    }
    // more survivors than SMAX?  (the next head is still inside the margin)
    const float nh = fminf(fminf(h0, h1), fminf(h2, h3));
    bool slow = open && nh <= tau + margin;
    // a wave whose list is full and whose last entry is inside the margin may have dropped a survivor
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
      const float lastv = lv[(ww * KDW + KDW - 1) * 64 + lane];
      if (lastv <= tau + margin) slow = true;          // +inf (list not full) and NaN never pass
    }
    if (!(tau < INFINITY)) slow = false;               // fewer than KD finite candidates at all: nothing was dropped
    if (lane_n >= N) { slow = false; cnt = 0; }        // padding lanes of the last query tile
    // any query of the tile unsettled -> the whole tile goes to the clean-up launch (wave-uniform decision)
    const bool tile_slow = __builtin_amdgcn_ballot_w64(slow) != 0ull;
    if (tile_slow) {
      if (lane == 0) a.wg_flags[blockIdx.x] = 1;
      cnt = -1;
    }
    slow = tile_slow;
    scnt[lane] = cnt;
    // Sort keys.  A survivor more than `margin` away from both neighbours in this (sorted) list keeps its prefilter
    // distance (+ |x|^2, the term the prefilter leaves out): its order against every other survivor is already the
    // contract's (each distance is within eps of the exact one).  The others — near-ties, and everything around the KD-th
    // rank by construction — get the exact contract distance from the pair pass below.
    if (!slow) {
      const float sqxv = a.sqx[(size_t)bg * N + nc];
#pragma unroll
      for (int j = 0; j < SMAX; ++j) {
        if (j < cnt) {
          const bool near = (j > 0 && sv_d[j] - sv_d[j - 1] <= margin) || (j + 1 < SMAX && j + 1 < cnt && sv_d[j + 1] - sv_d[j] <= margin);
          if (near) {
            const int slot = atomicAdd(npairs, 1);
            plist[slot] = (uint16_t)((j << 6) | lane);
          } else {
            keys[j * 64 + lane] = pack_key(sv_d[j] + sqxv, sidx[j * 64 + lane]);
          }
        }
      }
    }
  }
  KNN_TL(30);
  __syncthreads();
  KNN_TL(25);

  // ---- exact contract distance of the pairs that need it: one thread per pair, dense over the workgroup
  const float* xcb = a.xh + (size_t)bg * cpad * N;
  const float* ycb = a.yh + (size_t)bg * cpad * M;
  {
    const int np = *npairs;
    for (int p = tid; p < np; p += 256) {
      const int e = plist[p];
      const int q = e & 63, sv = e >> 6;
      const int m = sidx[sv * 64 + q];
      const int n = min(n0 + q, N - 1);
      float d = pf_exact_dist(xcb + n, N, ycb + m, M, cpad, a.sqx[(size_t)bg * N + n], sqy[m]);
      if (HAS_RP) d = d + a.relpos[(size_t)n * M + m];
      keys[sv * 64 + q] = d == d && d < INFINITY ? pack_key(d, m) : (double)INFINITY;   // NaN / +inf never enter a list
    }
  }
  KNN_TL(26);
  __syncthreads();
  KNN_TL(27);
  if (w != 0) return;

  // ---- wave 0: rank the survivors by their keys (a flagged tile writes nothing: the clean-up launch owns it)
  const int myc = scnt[lane];
  if (__builtin_amdgcn_ballot_w64(myc < 0) != 0ull) return;
  TopList<KD> fin;
  fin.init();
#pragma unroll
  for (int sv = 0; sv < SMAX; ++sv) {
    if (__builtin_amdgcn_ballot_w64(sv < myc) == 0ull) break;
    const double kv = sv < myc ? keys[sv * 64 + lane] : (double)INFINITY;
    fin.template insert_key<false>(kv);
  }
  if (lane_n < N) {
    const size_t obase = ((size_t)bg * N + nc) * a.k;
    int outj = 0;
#pragma unroll
    for (int j = 0; j < KD; ++j) {
      if (j < a.kd && j % a.dilation == 0 && outj < a.k) {
        const int bi = key_index(fin.key[j]);
        const int bc = (unsigned)bi < (unsigned)M ? bi : 0;
        if (a.nn16) a.nn16[obase + outj] = (uint16_t)bc;
        else a.nn_idx[obase + outj] = bc;
        if (a.center) a.center[obase + outj] = lane_n;
        ++outj;
      }
    }
  }
  KNN_TL(31);
}

}  // namespace gkg

using namespace gkg;

// prefilter + exact re-rank (knn_pf_kernel): un-split, normalised, fp32-contract problems
template <int KD, int KDW, bool HAS_RP, int PBUF, bool SB = false>
static hipError_t launch_pf_v(const KnnArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  if constexpr (KDW <= 16 && HAS_RP && !SB) {                     // single-batch form (see the kernel): narrow groups
    if (a.cp16 <= 64) return launch_pf_v<KD, KDW, HAS_RP, PBUF, true>(a, grid, lds, st);
  }
  if (lds > 64 * 1024) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_pf_kernel<KD, KDW, HAS_RP, PBUF, SB>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;
  }
  hipLaunchKernelGGL((knn_pf_kernel<KD, KDW, HAS_RP, PBUF, SB>), grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

// HAS_RP is a compile-time parameter of the launcher: the relative_pos forms are instantiated in this translation unit,
// the others in gkg_knn_pf_norp.hip (the same source with GKG_KNN_NORP_PART defined), so that they compile in parallel.
template <int KD, int KDW, bool HAS_RP>
static hipError_t launch_pf(const KnnArgs& a, dim3 grid, hipStream_t st) {
  GkgProfScope prof(GKG_PROF_KNN_TILE, st);
  const size_t stage = (size_t)2 * QT * (a.cp16 + 8) * 2;
  const size_t lists = (size_t)2 * NW * KDW * 64 * 4 + (size_t)(KD + PF_EXTRA) * 64 * (4 + 2) + 64 * 4 + 16;
  // buffered selection with the largest candidate buffer (16 or 12 entries per lane: 32 / 24 KiB) that keeps the
  // workgroups per CU the register budget allows (3 for lists <= 16, 2 above)
  const size_t per_wg = (size_t)160 * 1024 / (KDW <= 16 ? 3 : 2);
  const int pbuf = stage + 16 * 2048 <= per_wg ? 16 : (stage + 12 * 2048 <= per_wg ? 12 : 0);
  const size_t need = stage + (size_t)pbuf * 2048 + (pbuf ? NW * 64 * sizeof(float) : 0);      // + the shared admission bounds
  const size_t lds = need > lists ? need : lists;
  if (pbuf == 16) return launch_pf_v<KD, KDW, HAS_RP, 16>(a, grid, lds, st);
  if (pbuf == 12) return launch_pf_v<KD, KDW, HAS_RP, 12>(a, grid, lds, st);
  return launch_pf_v<KD, KDW, HAS_RP, 0>(a, grid, lds, st);
}

namespace gkg {
#ifdef GKG_KNN_NORP_PART
hipError_t launch_knn_prefilter_norp(const KnnArgs& a, dim3 grid, int KD, hipStream_t st) {
  switch (KD) {
    case 9: return launch_pf<9, 12, false>(a, grid, st);
    case 18: return launch_pf<18, 18, false>(a, grid, st);
    case 27: return launch_pf<27, 27, false>(a, grid, st);
    default: return launch_pf<36, 36, false>(a, grid, st);
  }
}
#else
hipError_t launch_knn_prefilter_norp(const KnnArgs& a, dim3 grid, int KD, hipStream_t st);
hipError_t launch_knn_prefilter(const KnnArgs& a, dim3 grid, int KD, hipStream_t st) {
  if (!a.relpos) return launch_knn_prefilter_norp(a, grid, KD, st);
  switch (KD) {
    case 9: return launch_pf<9, 12, true>(a, grid, st);
    case 18: return launch_pf<18, 18, true>(a, grid, st);
    case 27: return launch_pf<27, 27, true>(a, grid, st);
    default: return launch_pf<36, 36, true>(a, grid, st);
  }
}
#endif
}  // namespace gkg
