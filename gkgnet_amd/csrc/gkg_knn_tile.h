// The fused k-NN tile kernel (template) and its launcher, shared by the two translation units that instantiate it:
// gkg_knn.hip (fp32 contract forms) and gkg_knn_bf.hip (bf16-contraction forms), compiled in parallel.
// Arithmetic contract: include/gkg_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "gkg_knn_common.h"
#include "gkg_topk_merge.h"


namespace gkg {

// Measured on MI355X (tools/ubench/mfma_valu_overlap.hip): v_mfma_f32_32x32x2_f32 and fp32 VALU work of the
// waves of one SIMD do NOT overlap (they share the fp32 datapath) — every vector instruction spent on the
// selection adds to the matrix time, so the per-candidate work is kept minimal: 3 adds for the distance
// (the -2 is folded into the staged queries, |y|^2 is broadcast with v_readlane and doubles as the
// out-of-range mask), 4 to pack the (distance, index) key and a 2-instruction-per-slot sorted insert.  A sparse variant (per-lane clz walk over a
// "beats my k-th best" mask with the distances parked in LDS) was measured 1.5-1.7x SLOWER: per-wave lists
// see only a quarter of the keys, so some lane of 64 passes for almost every candidate and the divergent
// loop serialises on LDS latency.
// BUF > 0 — buffered selection.  The sorted insert costs 2*KD+6 vector instructions per candidate and lane, and on gfx950
// those are paid in matrix time (fp32 MFMA and fp32 VALU do not overlap).  Once a list has seen a few dozen keys almost
// every candidate fails against its KD-th entry, but per-lane divergence rules out skipping: some lane of 64 passes for
// almost every candidate.  So a candidate is only TESTED against the lane's (possibly stale) KD-th distance — one compare —
// and, when it passes, APPENDED raw (distance, index: 8 bytes) to a per-lane LDS buffer of BUF entries; the sorted inserts
// run in wave-uniform flushes: when some lane's buffer could overflow within the next 8 candidates (and at the end of the
// stream) every lane inserts its buffered entries, i.e. max-over-lanes(count) insert steps, and refreshes its threshold.
// A stale threshold only admits MORE candidates than necessary; the insert itself re-decides with the full (distance,
// index) key, so the result is bit-identical to the direct form (ties included: keys arrive in increasing index order
// within a wave, so a later candidate equal to the KD-th entry never displaces it — the strict '<' is exact).
// BF — the contraction on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16: 16x the fp32 rate, and unlike the fp32 MFMA it
// overlaps the selection's vector work): normalised tokens rounded to bf16, products exact, fp32 accumulation, the squared
// norms added in fp32.  For callers under bf16 autocast, where the reference itself runs x.y^T in bf16 AND rounds the
// product matrix to bf16 (torch_edge.py:35-51 under autocast) — this form keeps more of the fp32 answer than that.  Not
// covered by the bit-exact index contract (the accumulation order inside the 16-deep dot product is the hardware's).
// NWV — waves per workgroup.  4 (default): the waves split the KEY tiles of one 64-query tile and their lists are merged at
// the end — the only way to fill the chip when a launch has few query tiles.  1: a workgroup is ONE wave that streams all
// keys for its 64 queries — one list per query instead of four quarter-stream lists, i.e. k·d·(1 + ln(M / k·d)) expected
// admissions instead of 4·k·d·(1 + ln(M / 4·k·d)) (k·d = 9, M = 1296: 54 vs 164), no merge, no barrier after the staging.
// Same result bit for bit (the k·d smallest (distance, index) keys do not depend on how the stream was cut up).
// MRF (round 5, SURVEY §8 row g2: "the k-NN + gather kernel") — the max-relative aggregation in the epilogue.  After the merge
// the workgroup holds the final lists of its 64 queries: they go to LDS (no int64 index tensor, no centre plane, unless the
// caller asks for the compact u16 list), and the 256 threads gather the k neighbour rows of every query from the token-major
// source (one thread = one query x 4 channels, k float4 row-segment loads + the centre, all issued before the max chain:
// mr_fwd_tm_kernel's arithmetic, same bits) and write the grouped projection's interleaved operand U[q][t][2i] = x,
// U[q][t][2i+1] = max_j (x_j - x_i) and the winning rows (u16) the backward scatters from.  Replaces the launch pair
// knn_tile_kernel -> mr_fwd_tm_kernel (reference torch_edge.py:164-176 -> torch_vertex.py:49-61); while one workgroup of a
// CU gathers, the CU's other two or three are in their matrix phase.
template <int KD, bool HAS_RP, int KU, bool GUARD = true, int BUF = 0, bool BF = false, int NWV = NW, bool MRF = false>
__global__ __launch_bounds__(64 * NWV, (BF && HAS_RP) ? (KD <= 18 ? 3 : 2) : (KD <= 12 ? (MRF ? 3 : 4) : (KD <= 27 ? 3 : 2))) void knn_tile_kernel(KnnArgs a) {
  // (MRF with the 9-entry list: three workgroups per SIMD instead of four — at 128 registers the gather's k float4 rows next to
  // the bias row of the matrix phase spilled 11-15 registers; the 18 x 18 stages it serves launch three workgroups per CU)
  static_assert(!MRF || (NWV > 1 && !BF), "fused aggregation: the merged-list forms of the fp32 contract");
  extern __shared__ float smem[];
  constexpr int TH = 64 * NWV;                  // threads
  constexpr int RPS = TH / 16;                  // query-staging rows per pass
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware workgroup -> (problem, query tile) map.  Workgroups are dealt round-robin over the 8 XCDs, each with
  // its own L2; all query tiles of one (b,g) problem stream the SAME keys, so they are given linear ids that are
  // congruent mod 8 (same XCD) and adjacent in dispatch order: the keys are then fetched into one L2 once instead
  // of once per query tile (measured: FETCH_SIZE 53 -> see profiles/).  Placement only affects speed.
  const int lin = blockIdx.x;
  KNN_TL(0);
  if (a.wg_flags && a.wg_flags[lin] == 0) return;   // clean-up pass behind knn_pf_kernel: only the tiles it flagged
  // the map (gkg_knn_common.h): problem-major, bias-major (pvig_m stage 1, bf16 form: 43.5 GB of relative_pos per launch from
  // the MALL -> 0.34 GB, against 5.4 GB of keys that now re-enter per query tile) or interleaved
  int bg, qt;
  if (!knn_map(a, lin, bg, qt)) return;         // the grid is padded (uniform exit)
  const int split = blockIdx.z;
  const int n0 = qt * QT;
  const int N = a.N, M = a.M, cpad = a.cpad;

  // ---- first key batch of this wave's first tile and the lane's |x|^2: issued before the query staging so their
  //      L2/HBM round trip overlaps it (every workgroup of the launch runs this prologue at the same time)
  const int lane_n = n0 + lane;
  const int nc = lane_n < N ? lane_n : N - 1;
  const int kk = lane >> 5;       // which k of the k-pair this lane feeds
  const int l31 = lane & 31;
  const float* yp = a.yh + (size_t)bg * cpad * M;
  const int t_begin = split * a.tiles_per_split;
  // Visiting order of the key tiles.  With a positional bias (HAS_RP: image queries over image keys, both in raster order)
  // the stream starts at the key tile under the query tile's own image position and alternates outwards: neighbours in
  // feature space are mostly neighbours in the image (and the bias favours them), so a scan from index 0 hands a query
  // ever-closer keys until its own position — the worst order for a running top-k, nearly every candidate beats the
  // threshold — while centre-out the lists fill with near-final entries first.  The result does not depend on the order
  // (inserts compare whole (distance, index) keys); only the buffered form's admission test becomes '<=' (a candidate
  // equal to the k*d-th distance may still win on its index).  On the models' activations: cfg3's k-NN kernels 4.71 ->
  // 4.42 ms, cfg5's 26.6 -> 24.1 ms; neutral for the fp32 forms (direct inserts / prefilter), whose work per candidate does
  // not depend on the threshold.
  const int t_stop = min(t_begin + a.tiles_per_split, (a.M + KT - 1) / KT);
  const int TV = max(t_stop - t_begin, 0);
  int c0 = t_begin;
  if (HAS_RP && TV > 0) c0 = min(max((int)(((long long)(n0 + QT / 2) * a.M / a.N) / KT), t_begin), t_stop - 1);
  const int nleft = c0 - t_begin, nright = t_stop - 1 - c0, nboth = min(nleft, nright);
  auto tile_at = [&](int i) -> int {             // i-th visited tile, 0 <= i < TV
    if (!HAS_RP) return t_begin + i;
    if (i <= 2 * nboth) return (i & 1) ? c0 + ((i + 1) >> 1) : c0 - (i >> 1);
    return nleft >= nright ? c0 - (i - nboth) : c0 + (i - nboth);
  };
  const int t_first = tile_at(min(w, max(TV - 1, 0)));
  float an[KU];
  constexpr int KB = 4;                          // BF: k16-steps per key-operand batch
  const int cp16 = a.cp16, S16 = cp16 >> 4;
  const uint4* ybp = BF ? reinterpret_cast<const uint4*>(a.yb) + (size_t)bg * (cp16 >> 3) * M : nullptr;   // [octet][key]
  uint4 bn_[KB];
  if (BF) {
    const int mk0 = min(t_first * KT + l31, M - 1);
    const uint4* y0 = ybp + (size_t)kk * M + mk0;
#pragma unroll
    for (int u = 0; u < KB; ++u) bn_[u] = u < S16 ? y0[(size_t)(2 * u) * M] : make_uint4(0, 0, 0, 0);
  } else {
    const int t0 = t_first;
    const int mk0 = min(t0 * KT + l31, M - 1);
    const float* y0 = yp + (size_t)kk * M + mk0;
#pragma unroll
    for (int u = 0; u < KU; ++u) an[u] = y0[(size_t)(2 * u) * M];
  }

  // The last k-pair of every contraction adds |x|^2 and |y|^2 on the matrix pipe: keys feed (1, |y|^2), queries
  // (|x|^2, 1), so acc = fma(|y|^2, 1, fma(1, |x|^2, acc)) — bit for bit ((|x|^2 + (-2 x.y)) + |y|^2), the contract's
  // order — and each candidate saves two vector adds and a v_readlane (the vector pipe is the contended one).
  // Used by the deep-batch instantiations (KU == 8: channel counts that are multiples of 16; KU == 6, round 5: multiples of 12 —
  // pvig_m's 12 / 24 channels per group run 6 + 1 / 12 + 1 k-pairs instead of 8 + 1 / 12 with three vector adds per candidate);
  // measured a few per cent slower on the KU == 4 ones (c = 200 at 36x36), which keep the three vector adds.
  constexpr bool FOLD = (KU == 8 || KU == 6) && !BF;
  const float qtail0 = (!FOLD || kk) ? 1.0f : a.sqx[(size_t)bg * N + min(n0 + l31, N - 1)];
  const float qtail1 = (!FOLD || kk) ? 1.0f : a.sqx[(size_t)bg * N + min(n0 + 32 + l31, N - 1)];
  const float sqx = FOLD ? 0.0f : a.sqx[(size_t)bg * N + nc];

  // ---- stage the query tile scaled by -2 (exact): xs[ch][64] = -2 * xh (zero for n >= N).
  //      All loads of a pass are issued before the first LDS store.
  // BF: xq[64][cp16 + 8] bf16 rows (-2 x, exact; 16 B of padding per row keeps the fragment reads conflict-free)
  const int qpitch = (cp16 + 8) * 2;             // bytes
  if (BF) {
    const uint4* xbp = reinterpret_cast<const uint4*>(a.xb) + (size_t)bg * (cp16 >> 3) * N;
    const int chunks = cp16 >> 3;                  // 16-byte chunks per query row
    for (int i = tid; i < QT * chunks; i += TH) {
      const int ck = i >> 6, q = i & 63;           // consecutive threads: consecutive queries of one octet (coalesced)
      uint4 v = make_uint4(0, 0, 0, 0);
      if (n0 + q < N) v = xbp[(size_t)ck * N + n0 + q];
      // x -> -2x on packed bf16: exponent + 1 and sign flip, zeros stay zeros (|x| <= 1 after normalisation; raw inputs
      // near the top of the range would overflow to inf like any -2x)
      auto m2h = [](unsigned hv) -> unsigned {
        const unsigned e = hv & 0x7f80u;
        if (e == 0u) return 0u;                                   // zero / denormal
        if (e == 0x7f80u) return hv ^ 0x8000u;                     // inf / NaN keep their class
        if (e == 0x7f00u) return ((hv ^ 0x8000u) & 0x8000u) | 0x7f80u;   // overflow -> inf
        return (hv + 0x80u) ^ 0x8000u;
      };
      auto m2 = [&](unsigned wv) { return m2h(wv & 0xffffu) | (m2h(wv >> 16) << 16); };
      v.x = m2(v.x); v.y = m2(v.y); v.z = m2(v.z); v.w = m2(v.w);
      *reinterpret_cast<uint4*>(reinterpret_cast<char*>(smem) + q * qpitch + 16 * ck) = v;
    }
  } else {
    const float* xp = a.xh + (size_t)bg * cpad * N;
    if ((N & 3) == 0) {
      const int q4 = (tid & 15) * 4;               // 16 float4 per 64-query row, 16 rows per pass
      const bool inb = n0 + q4 < N;                 // N % 4 == 0: a float4 is entirely in or out
      const float* src = xp + (size_t)(tid >> 4) * N + n0 + q4;
      float4* dst = reinterpret_cast<float4*>(smem + (tid >> 4) * QT + q4);
      for (int ch = 0; ch < cpad; ch += 4 * RPS) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = ch + RPS * u + (tid >> 4);
          v[u] = (inb && row < cpad) ? *reinterpret_cast<const float4*>(src + (size_t)(ch + RPS * u) * N)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ch + RPS * u + (tid >> 4) < cpad)
            dst[(ch + RPS * u) * (QT / 4)] = make_float4(-2.f * v[u].x, -2.f * v[u].y, -2.f * v[u].z, -2.f * v[u].w);
      }
    } else {
      for (int i = tid; i < cpad * QT; i += 4 * TH) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = i + TH * u;
          const int ch = j >> 6, q = j & 63;
          v[u] = (j < cpad * QT && n0 + q < N) ? xp[(size_t)ch * N + n0 + q] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i + TH * u < cpad * QT) smem[i + TH * u] = -2.f * v[u];
      }
    }
  }
  // floats of the staged query tile (bf16 form: 64 rows of cp16 + 8 bf16), behind which the candidate buffer lives
  const size_t qfloats = BF ? (size_t)QT * (a.cp16 + 8) / 2 : (size_t)cpad * QT;
  if (BUF > 0) smem[qfloats + (size_t)BUF * 2 * TH + tid] = INFINITY;   // ths (see below)
  KNN_TL(1);
  __syncthreads();
  KNN_TL(2);

  const int n = lane_n;
  const float* sqy = a.sqy + (size_t)bg * M;

  const bool two_blocks = n0 + 32 < N;   // wave-uniform

  TopList<KD> top;
  top.init();
  // buffered selection state: the buffer lives behind the staged queries, [BUF][256] x {distance, index}
  float2* cbuf = reinterpret_cast<float2*>(smem + qfloats) + tid;
  // the lane's append cursor IS its entry count (cursor - cbuf = count * TH): an admitted candidate costs the store, one add on
  // the cursor and the move of its index; the count is only formed inside a flush
  typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) v2u_t lds_v2u_t;   // a 32-bit LDS pointer: the cursor is ONE register, ds_write takes it as is
  // (kept as opaque integers: when the compiler can relate the cursor to the buffer's base it carries an OFFSET instead and
  // re-adds the base at every store)
  unsigned cw0 = (unsigned)(size_t)(lds_v2u_t*)cbuf;
  asm volatile("" : "+v"(cw0));
  unsigned cw_lim = cw0 + (BUF > 8 ? BUF - 8 : 0) * TH * 8;
  asm volatile("" : "+v"(cw_lim));
  unsigned cw = cw0;
  float thr = INFINITY;
  // Shared admission bound (SHARE: the buffered form; in the guarded direct form the two extra selects per candidate cost
  // more than the saved inserts — C = 640 / k*d = 27 at 18 x 18: 119 -> 157 us — so it keeps its own bound).  The 4 waves of the workgroup keep separate lists
  // over disjoint key tiles of the SAME 64 queries.  With Q = ceil(KD / 4): the Q best entries of each wave's list are 4 Q
  // >= KD distinct candidates, so the query's final KD-th distance is <= sh = max over the waves of their Q-th entry — a
  // much tighter bound than a wave's own KD-th entry (a wave sees a quarter of the keys; its Q-th entry is about where the
  // final KD-th will be).  Candidates with dist > sh can never reach the final list and are not inserted (dist == sh may tie
  // in by index: kept).  Each wave publishes its Q-th distance in LDS and reads the others' without synchronisation: a
  // stale value is an older, LARGER one, still a valid bound.  Results are bit-identical; the sorted inserts — the dominant
  // cost of long lists — drop by about half (k*d = 36: 136 -> 73 expected inserts per lane over 576 keys).
  constexpr bool SHARE = BUF > 0 && KD >= 16 && NWV > 1;   // 9-entry lists: Q = 3 saves too few inserts to pay for the exchange
  constexpr int QSH = (KD + NWV - 1) / NWV;
  float* ths = smem + qfloats + (size_t)(BUF > 0 ? BUF : 0) * 2 * TH;      // [NWV][64], behind the candidate buffer
  float sh = INFINITY;
  auto refresh_shared = [&]() {
    const float mine = key_dist(top.key[QSH - 1]);                // +inf while the list holds fewer than Q entries
    ths[w * 64 + lane] = mine;
    float m = mine;
#pragma unroll
    for (int ww = 0; ww < NWV; ++ww) m = fmaxf(m, ths[ww * 64 + lane]);
    sh = m;
  };
  auto next_up = [](float v) -> float {                            // smallest float > v (v finite), +inf stays +inf
    if (!(v < INFINITY)) return v;
    const int b = __float_as_int(v);
    return __int_as_float(v >= 0.0f ? b + 1 : b - 1);
  };
  // Lists of 16+ entries with a 16-entry buffer: when some lane holds more than 3 candidates the whole batch goes through a
  // merge network (sort the 16 batch keys, bitonic-merge them into the list: 63 + 120 compare-exchanges at k*d = 36,
  // generated by tools/gen_topk_merge.py) instead of max-over-lanes sorted inserts of 2 k*d + 6 instructions each — the
  // same k*d smallest (distance, index) keys.
  constexpr bool NET = BUF >= 12 && KD >= 16;
  auto flush = [&]() {
    unsigned cw_now = cw;
    asm volatile("" : "+v"(cw_now));
    int bcnt = (int)((cw_now - cw0) / (TH * 8));
#if defined(KNN_ABLATE) && KNN_ABLATE == 3
    unsigned long long* gkg_knn_ablate_counters = reinterpret_cast<unsigned long long*>(a.part_v);
    if (lane == 0) {                               // tools/ubench/knn_ablate.py: flushes, batches through the network
      atomicAdd(&gkg_knn_ablate_counters[0], 1ull);
      if (__builtin_amdgcn_ballot_w64(bcnt > 3) != 0ull) atomicAdd(&gkg_knn_ablate_counters[1], 1ull);
    }
    atomicAdd(&gkg_knn_ablate_counters[2], (unsigned long long)bcnt);
    {
      int mx = bcnt;
      for (int m_ = 1; m_ < 64; m_ <<= 1) mx = max(mx, __shfl_xor(mx, m_, 64));
      if (lane == 0) atomicAdd(&gkg_knn_ablate_counters[3], (unsigned long long)mx);
    }
#endif
    if constexpr (NET) {
      if (__builtin_amdgcn_ballot_w64(bcnt > 3) != 0ull) {
        double b[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i < BUF) {
            const float2 e = cbuf[i * TH];
            b[i] = i < bcnt ? pack_key(e.x, __float_as_int(e.y)) : (double)INFINITY;
          } else {
            b[i] = (double)INFINITY;
          }
        }
        TopMerge16<KD>::run(top.key, b);
        bcnt = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < BUF; ++i) {                 // forward branches only: a back edge makes the allocator duplicate the list
      if (__builtin_amdgcn_ballot_w64(i < bcnt) == 0ull) break;
      const float2 e = cbuf[i * TH];
      const double k = i < bcnt ? pack_key(e.x, __float_as_int(e.y)) : (double)INFINITY;
      top.template insert_key<false>(k);
    }
    cw = cw0;
    thr = key_dist(top.key[KD - 1]);               // +inf while the list is not full
    if (SHARE) {
      refresh_shared();
      thr = fminf(thr, next_up(sh));               // strict '<' against own KD-th entry, '<=' against the shared bound
    }
#if defined(KNN_ABLATE) && KNN_ABLATE == 1
    thr = -INFINITY;                               // tools/ubench/knn_ablate.py: every candidate tested, none admitted after the first flush
#endif
  };
#if defined(KNN_ABLATE) && KNN_ABLATE == 2
  float abl_sink = INFINITY;
#endif

  const int ktiles = (M + KT - 1) / KT;
  const int t_end = min(t_begin + a.tiles_per_split, ktiles);
  const int CP = cpad / 2;          // k-pairs; cpad % 8 == 0 -> CP % 4 == 0

  // The first key batch of every later tile is loaded during the PREVIOUS tile's last MFMA batch (before its selection
  // phase), so no tile starts with an exposed L2 round trip.
  // bf16 form: relative_pos is the accumulators' initial value, so its loads cannot hide under the contraction the way the
  // fp32 form's do — measured (s1 shape): 46 % of the wave cycles waiting on memory at 2.3 waves per SIMD, vector pipe 51 %
  // busy.  The tile's 2 x 16 values (and |y|^2) are therefore fetched ONE TILE AHEAD, during the previous tile's selection
  // phase (after that tile's last key-operand loads, so nothing in between waits on them).
  float4 rq0[4], rq1[4];                        // next tile: 4 groups of 4 key rows for each of the two query blocks
  float sy_n = MASKED_SQ;
  // buffer addressing over this workgroup's 64 rows of relative_pos: 16-byte loads at any dword alignment (odd M too),
  // and reads past the end of the tensor (rows >= N; the last row's masked keys) return zeros instead of faulting
  const size_t rp_row0 = (size_t)min(n0, N - 1) * M;
  const size_t rp_left = ((size_t)N * M - rp_row0) * sizeof(float);
  const __amdgpu_buffer_rsrc_t rp_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(HAS_RP ? a.relpos + rp_row0 : a.sqy), 0, (int)(rp_left > 0x7fffffffull ? 0x7fffffffull : rp_left), 0x00020000);
  const unsigned rp_o0 = (unsigned)(((size_t)l31 * M + 4 * kk) * sizeof(float));
  const unsigned rp_o1 = (unsigned)(((size_t)(32 + l31) * M + 4 * kk) * sizeof(float));
  auto fetch_side = [&](int tt) __attribute__((always_inline)) {
    const int mm0 = tt * KT;
    sy_n = (mm0 + l31 < M) ? sqy[min(mm0 + l31, M - 1)] : MASKED_SQ;
    if (HAS_RP) {
      // lane (l31, kk), group g, element j <-> key row mm0 + 8 g + 4 kk + j of query block row l31
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        rq0[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rp_rsrc, (int)(rp_o0 + 32 * g), mm0 * 4, 0));
        rq1[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rp_rsrc, (int)(rp_o1 + 32 * g), mm0 * 4, 0));
      }
    }
  };
  if (BF && w < TV) fetch_side(t_first);

  for (int iv = w; iv < TV; iv += NWV) {
    const int t = tile_at(iv);
    const int t_next = iv + NWV < TV ? tile_at(iv + NWV) : t;
    const int m0 = t * KT;
    const int mk = min(m0 + l31, M - 1);
    const int mk_next = min(t_next * KT + l31, M - 1);
    // ---- side inputs of this tile, issued first so their latency hides under the contraction:
    //      |y|^2 of key m0+l31 (broadcast per candidate with v_readlane; MASKED_SQ sends keys past M to the end of every list) and the
    //      positional bias of this lane's query row
    const float sy32 = BF ? sy_n : ((m0 + l31 < M) ? sqy[mk] : MASKED_SQ);
    float rp[KT];
    if (HAS_RP && !BF) {
      const float* rpp = a.relpos + (size_t)nc * M + m0;
      if (m0 + KT <= M && (M & 3) == 0) {
#pragma unroll
        for (int j = 0; j < KT / 4; ++j) {
          const float4 v4 = *reinterpret_cast<const float4*>(rpp + 4 * j);
          rp[4 * j + 0] = v4.x; rp[4 * j + 1] = v4.y; rp[4 * j + 2] = v4.z; rp[4 * j + 3] = v4.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < KT; ++j) rp[j] = rpp[min(j, M - 1 - m0)];
      }
    }
    // ---- contraction: acc0 = keys x (-2 queries[0..31]), acc1 = keys x (-2 queries[32..63]).
    //      Key operand double-buffered in registers (KU k-pairs per batch).
    f32x16 acc0, acc1;
    if (BF) acc0 = acc1 = f32x16{0};
    if (BF && HAS_RP) {
      // bf16 form: the accumulators START from relative_pos (fetched one tile ahead, see fetch_side): the bias rides through
      // the contraction instead of costing an add per candidate
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        acc0[4 * g] = rq0[g].x; acc0[4 * g + 1] = rq0[g].y; acc0[4 * g + 2] = rq0[g].z; acc0[4 * g + 3] = rq0[g].w;
        acc1[4 * g] = rq1[g].x; acc1[4 * g + 1] = rq1[g].y; acc1[4 * g + 2] = rq1[g].z; acc1[4 * g + 3] = rq1[g].w;
      }
    }
    if (BF) {
      // keys = A operand (lane: key l31, 8 channels 16 s + 8 kk ...), the two query blocks = B operands from LDS
      typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
      const uint4* ykp = ybp + (size_t)kk * M + mk;
      const uint4* ykn = ybp + (size_t)kk * M + mk_next;
      const char* xq0 = reinterpret_cast<const char*>(smem) + l31 * qpitch + 16 * kk;
      const char* xq1 = xq0 + 32 * qpitch;
      for (int s0 = 0; s0 < S16; s0 += KB) {
        const bool last = s0 + KB >= S16;                         // uniform: prefetch the NEXT tile's first batch
        uint4 ac[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) ac[u] = bn_[u];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          const int sn = last ? u : s0 + KB + u;
          bn_[u] = sn < S16 ? (last ? ykn : ykp)[(size_t)(2 * sn) * M] : make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          if (s0 + u < S16) {
            const bf16x8_t av = __builtin_bit_cast(bf16x8_t, ac[u]);
            const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xq0 + 32 * (s0 + u)));
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b0, acc0, 0, 0, 0);
            if (two_blocks) {
              const bf16x8_t b1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xq1 + 32 * (s0 + u)));
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b1, acc1, 0, 0, 0);
            }
          }
        }
      }
    } else {
      const float* ykp = yp + (size_t)kk * M + mk;
      const float* ykn = yp + (size_t)kk * M + mk_next;
      const float* xsp = smem + kk * QT + l31;
      float ac[KU];
      // One batch = KU k-pairs.  The first MFMA of a tile takes a zero literal as its accumulator (no 32 v_mov per tile), and the
      // query operands are read from LDS two k-pairs ahead of the MFMAs that use them (sched_group_barrier: the compiler
      // otherwise reuses ONE register pair — ds_read2, s_waitcnt lgkmcnt(0), two MFMAs, ds_read2 ... — and every k-pair
      // waits out an LDS round trip).
      const f32x16 zero16 = {0};
      // (the peeled first batch and the second operand pair in flight cost ~4 registers: not in the instantiations that sit at
      // their register cap — those keep the zero-filled accumulators and the compiler's own order)
      constexpr bool PIPE = !(HAS_RP && KU == 8 && ((KD <= 12 && !MRF) || KD == 27));
      auto batch2 = [&](int s, auto first_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
        const bool last = s + KU >= CP;                       // uniform: prefetch the NEXT tile's first batch
        const float* pb = last ? ykn : ykp + (size_t)(2 * (s + KU)) * M;
#pragma unroll
        for (int u = 0; u < KU; ++u) ac[u] = an[u];
#pragma unroll
        for (int u = 0; u < KU; ++u) an[u] = pb[(size_t)(2 * u) * M];
        __builtin_amdgcn_sched_barrier(0);      // keep the next batch's loads ahead of this batch's MFMAs
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const float b0 = xsp[(2 * (s + u)) * QT];
          const float b1 = xsp[(2 * (s + u)) * QT + 32];
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], b0, (FIRST && u == 0) ? zero16 : acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], b1, (FIRST && u == 0) ? zero16 : acc1, 0, 0, 0);
        }
        if constexpr (PIPE) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // DS reads of k-pairs 0, 1
#pragma unroll
          for (int u = 0; u < KU; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // the k-pair's two MFMAs
            if (u + 2 < KU) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
      };
      auto batch1 = [&](int s, auto first_c) __attribute__((always_inline)) {   // <= 32 queries: one query block
        constexpr bool FIRST = decltype(first_c)::value;
        const bool last = s + KU >= CP;
        const float* pb = last ? ykn : ykp + (size_t)(2 * (s + KU)) * M;
#pragma unroll
        for (int u = 0; u < KU; ++u) ac[u] = an[u];
#pragma unroll
        for (int u = 0; u < KU; ++u) an[u] = pb[(size_t)(2 * u) * M];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KU; ++u)
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], xsp[(2 * (s + u)) * QT], (FIRST && u == 0) ? zero16 : acc0, 0, 0, 0);
      };
      if (two_blocks) {
        if constexpr (PIPE) {
          batch2(0, std::true_type{});
          for (int s = KU; s < CP; s += KU) batch2(s, std::false_type{});
        } else {
          acc0 = acc1 = zero16;
          for (int s = 0; s < CP; s += KU) batch2(s, std::false_type{});
        }
        if (FOLD) {
          const float atail = kk ? sy32 : 1.0f;                // k-pair (1, |y|^2) x (|x|^2, 1)
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(atail, qtail0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(atail, qtail1, acc1, 0, 0, 0);
        }
      } else {                                    // tail query tile with <= 32 queries: one query block only
        if constexpr (PIPE) {
          batch1(0, std::true_type{});
          for (int s = KU; s < CP; s += KU) batch1(s, std::false_type{});
        } else {
          acc0 = zero16;
          for (int s = 0; s < CP; s += KU) batch1(s, std::false_type{});
        }
        if (FOLD) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kk ? sy32 : 1.0f, qtail0, acc0, 0, 0, 0);
        // lanes 32-63 hold no query here: acc1's registers are left as they are (an empty asm "defines" them — a zero fill or a
        // copy of acc0 is sunk into the join block and then runs on every tile of the common path too)
        asm volatile("" : "=v"(acc1));
      }
    }
    KNN_TL(3 + 2 * (iv / NWV));
    // next tile's relative_pos rows and |y|^2: in flight during this tile's selection phase (bf16 form; sy32 was copied)
    if (BF && iv + NWV < TV) fetch_side(t_next);
    // ---- lane l needs all 32 keys of ITS query: v_permlane32_swap exchanges the 32-lane halves of the two
    //      accumulators (vdst.hi <-> src.lo); afterwards lo = key rows (r&3)+8(r>>2), hi = those + 4.
    //      dist = ((|x|^2 + (-2 x.y)) + |y|^2) + relpos in the reference's order; with FOLD the first two adds happened
    //      in the contraction's last k-pair, otherwise |y|^2 is broadcast per candidate with v_readlane; it is
    //      MASKED_SQ for keys past M, which also masks them.
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
          const int row = 8 * g + 4 * hh + j;                              // increasing key order (tie rule)
          float dist = hh ? hi[j] : lo[j];
          if (BF) {
            // bf16 form (outside the bit-exact contract): relative_pos is already inside, and the query's own |x|^2 — one
            // constant for all of its candidates, it cannot change their order — is left out; |y|^2 stays (it also masks
            // the keys past M and keeps zero-norm keys where they belong)
            dist = dist + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sy32), row));
          } else {
            if (!FOLD) {                                                 // FOLD: both adds already happened on the matrix pipe
              const float sy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sy32), row));
              dist = (sqx + dist) + sy;
            }
            if (HAS_RP) dist = dist + rp[row];
          }
          if (BUF > 0) {
#if defined(KNN_ABLATE) && KNN_ABLATE == 2
            abl_sink = fminf(abl_sink, dist);       // tools/ubench/knn_ablate.py: the contraction + distance adds alone
#else
            if (HAS_RP ? dist <= thr : dist < thr) { // NaN fails; '<=' where the tiles are not visited in index order
              *(lds_v2u_t*)(size_t)cw = v2u_t{__float_as_uint(dist), (unsigned)(m0 + row)};
              // in place, behind the store (left to the compiler the add lands in a temporary in front of it, plus a copy back)
              asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(cw) : "i"(TH * 8) : "memory");
            }
#endif
          } else {
            top.template insert<GUARD>(dist, m0 + row);
          }
        }
      }
      // room for the next 8 candidates?  (the stream's last group flushes unconditionally)
#if defined(KNN_ABLATE) && KNN_ABLATE == 2
      if (BUF > 0 && g == 3 && iv + NWV >= TV) { top.template insert<false>(abl_sink, 0); }
#else
      if (BUF > 0 && ((g == 3 && iv + NWV >= TV) || __builtin_amdgcn_ballot_w64(cw > cw_lim) != 0ull)) flush();
#endif
    }
    KNN_TL(4 + 2 * (iv / NWV));
  }

  const int kd = a.kd;
  const bool partial = a.splits > 1;
  char* const out_base = (!MRF && a.nn16) ? reinterpret_cast<char*>(a.nn16) : reinterpret_cast<char*>(a.nn_idx);   // uniform
  const int out_sh = (!MRF && a.nn16) ? 1 : 3;
  const size_t obase = ((size_t)bg * N + nc) * a.k;
  const size_t pbase = (((size_t)split * a.BG + bg) * N + nc) * (size_t)kd;
  if constexpr (NWV == 1) {
    // ---- one wave, one list per query: the sorted list IS the answer
    if (n < N) {
      int next_rank = 0, outj = 0;
#pragma unroll
      for (int j = 0; j < KD; ++j) {
        if (j < kd) {
          const float bv = key_dist(top.key[j]);
          const int bi = key_index(top.key[j]);
          if (partial) {
            a.part_v[pbase + j] = bv;
            a.part_i[pbase + j] = bi;
          } else if (j == next_rank) {
            const int bc = (unsigned)bi < (unsigned)M ? bi : 0;             // non-finite distances only: stay in range
            // compact lists (gkg_knn_fwd_tm16: callers inside the block) or the int64 plane: ONE address, two store widths
            char* const op = out_base + ((obase + outj) << out_sh);
            if (out_sh == 1) *reinterpret_cast<uint16_t*>(op) = (uint16_t)bc;
            else *reinterpret_cast<int64_t*>(op) = bc;
            if (a.center) a.center[obase + outj] = n;
            ++outj;
            next_rank += a.dilation;
          }
        }
      }
    }
    KNN_TL(31);
    return;
  }
  // ---- merge the per-wave lists of each query: EVERY wave ranks its own entries among all lists (rank = own position +
  //      the number of smaller keys in each other list; lists are sorted, so only the first KD - j entries of another list
  //      can keep entry j below rank KD) and writes the ones that land in the output.  The packed keys are distinct across
  //      lists (disjoint key indices) except for empty slots (+inf, non-finite inputs only), whose duplicate ranks all write
  //      the same 0.  Until round 4 wave 0 merged the four lists with a serial head-of-list loop (9 dependent LDS round
  //      trips per rank) after the other waves had exited: 19 k of a cfg2 workgroup's 87 k cycles; this form takes 5 k, for
  //      any number of waves.  The launch time at cfg2 did not move (the CU's other two workgroups filled the idle SIMDs:
  //      tools/ubench/knn_timeline.py, profiles/r04_knn_timeline.txt).
  KNN_TL(28);
  __syncthreads();                       // everyone is done with xs / dmat
  KNN_TL(29);
  double* lk = reinterpret_cast<double*>(smem);      // [NWV][KD][64]
#pragma unroll
  for (int j = 0; j < KD; ++j) lk[(w * KD + j) * 64 + lane] = top.key[j];
  int* nbr = reinterpret_cast<int*>(lk + (size_t)NWV * KD * 64);      // MRF: [k][64] final neighbour rows of the 64 queries
  if constexpr (MRF) {
    for (int i = tid; i < a.k * 64; i += TH) nbr[i] = 0;               // ranks no finite candidate claims (non-finite inputs)
  }
  __syncthreads();
  KNN_TL(30);
  const int dil = a.dilation;
  const unsigned magic = dil > 1 ? 0xffffffffu / (unsigned)dil + 1u : 0u;      // r / dil for r < 2^16 (r <= KD * NWV)
  constexpr int JC = 9;                  // own entries ranked per pass (bounds the registers the ranks take next to the list)
#pragma unroll
  for (int j0 = 0; j0 < KD; j0 += JC) {
    int rank[JC];
#pragma unroll
    for (int u = 0; u < JC; ++u) rank[u] = j0 + u;
#pragma unroll 1
    for (int o = 1; o < NWV; ++o) {
      const int ww = (w + o) % NWV;
      const double* lp = lk + (size_t)ww * KD * 64 + lane;
#pragma unroll
      for (int i = 0; i < KD - j0; ++i) {
        const double ok = lp[i * 64];
#pragma unroll
        for (int u = 0; u < JC; ++u)
          if (j0 + u < KD && i < KD - (j0 + u)) rank[u] += ok < top.key[j0 + u] ? 1 : 0;
      }
    }
    if (n < N) {
#pragma unroll
      for (int u = 0; u < JC; ++u) {
        if (j0 + u < KD) {
          const int r = rank[u];
          if (r < kd) {
            const int bi = key_index(top.key[j0 + u]);
            if (partial) {
              a.part_v[pbase + r] = key_dist(top.key[j0 + u]);
              a.part_i[pbase + r] = bi;
            } else {
              const int q = dil > 1 ? (int)__umulhi((unsigned)r, magic) : r;
              if (q * dil == r) {
                const int bc = (unsigned)bi < (unsigned)M ? bi : 0;          // non-finite distances only: stay in range
                if constexpr (MRF) {
                  nbr[q * 64 + lane] = bc;
                  if (a.nn16) a.nn16[obase + q] = (uint16_t)bc;
                  if (a.nn_idx) a.nn_idx[obase + q] = bc;
                } else {
                  char* const op = out_base + ((obase + q) << out_sh);
                  if (out_sh == 1) *reinterpret_cast<uint16_t*>(op) = (uint16_t)bc;
                  else *reinterpret_cast<int64_t*>(op) = bc;
                }
                if (a.center) a.center[obase + q] = n;
              }
            }
          }
        }
      }
    }
  }
  if constexpr (MRF) {
    __syncthreads();
    const int G = a.mr_G, c = a.mr_c, C = G * c, k = a.k;
    const int b = bg / G, g = bg - b * G;
    const int f4n = c >> 2;                                    // float4 columns of a query's group segment
    const int nq = min(QT, N - n0);
    // x / src are token-major views (pointer, pitch, chunk: gkg_common.h "XM layout"); a group's c channels may span chunks,
    // a float4 never does
    const int ldx = a.mr_ldx, lds_ = a.mr_lds;
    const float* xrow0 = a.mr_x + ((size_t)b * N + n0) * ldx;
    const float* sb = a.mr_src + (size_t)b * M * lds_;
    const int Cq = C >> 2;                                     // channels per conv group (reference torch_nn.py:61) = XM chunk width
    // Two (query, 4-channel) tasks per thread at a time: all 2 (k + 1) row-segment loads are issued before the first maximum
    // chain (the gather is latency-bound: one task at a time measured +20 us on the cfg2 launch against 14.7 us for the
    // stand-alone aggregation kernel it replaces).
    const int ntask = nq * f4n;
    auto run = [&](auto careful, auto ks, const float4& xi, const int* id, const float4* v, float4& best, int (&ai)[4]) {
      constexpr int KS = decltype(ks)::value;
      float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        if (j < k) {
          const float d0 = v[j].x - xi.x, d1 = v[j].y - xi.y, d2 = v[j].z - xi.z, d3 = v[j].w - xi.w;
          const int wv = id[j];
          if (j == 0) { best = make_float4(d0, d1, d2, d3); ai[0] = ai[1] = ai[2] = ai[3] = wv; }
          else if constexpr (decltype(careful)::value) {
            if (mr_takes(d0, best.x)) { best.x = d0; ai[0] = wv; }
            if (mr_takes(d1, best.y)) { best.y = d1; ai[1] = wv; }
            if (mr_takes(d2, best.z)) { best.z = d2; ai[2] = wv; }
            if (mr_takes(d3, best.w)) { best.w = d3; ai[3] = wv; }
          } else {
            const bool t0 = d0 > best.x, t1 = d1 > best.y, t2 = d2 > best.z, t3 = d3 > best.w;
            best.x = t0 ? d0 : best.x; ai[0] = t0 ? wv : ai[0];
            best.y = t1 ? d1 : best.y; ai[1] = t1 ? wv : ai[1];
            best.z = t2 ? d2 : best.z; ai[2] = t2 ? wv : ai[2];
            best.w = t3 ? d3 : best.w; ai[3] = t3 ? wv : ai[3];
          }
          if constexpr (!decltype(careful)::value) {
            c0 = __builtin_fmaf(d0, 0.f, c0); c1 = __builtin_fmaf(d1, 0.f, c1);
            c2 = __builtin_fmaf(d2, 0.f, c2); c3 = __builtin_fmaf(d3, 0.f, c3);
          }
        }
      }
      return (c0 + c1) + (c2 + c3);
    };
    auto pass = [&](auto ks) {
      constexpr int KS = decltype(ks)::value;
      constexpr int TU = KS <= 9 ? 2 : 1;                      // tasks in flight per thread
      for (int task0 = tid; task0 < ntask; task0 += TU * TH) {
        int q[TU], chq[TU], id[TU][KS];
        float4 xi[TU], v[TU][KS];
        bool on[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          const int task = task0 + u * TH;
          on[u] = task < ntask;
          const int tk = on[u] ? task : task0;
          q[u] = tk / f4n;
          chq[u] = g * c + 4 * (tk - q[u] * f4n);
#pragma unroll
          for (int j = 0; j < KS; ++j) id[u][j] = nbr[(j < k ? j : 0) * 64 + q[u]];
          xi[u] = *reinterpret_cast<const float4*>(xrow0 + (size_t)q[u] * ldx + xm_col(chq[u], a.mr_xchunk));
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          const float* sp = sb + xm_col(chq[u], a.mr_schunk);
#pragma unroll
          for (int j = 0; j < KS; ++j) v[u][j] = *reinterpret_cast<const float4*>(sp + (size_t)id[u][j] * lds_);
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          float4 best;
          int ai[4];
          // mr_fwd_tm_kernel's chain: a plain '>' (first maximum wins) while chk accumulates d * 0 — NaN as soon as a difference
          // is NaN or infinite; such a lane (non-finite inputs only) redoes the chain with `takes` (a NaN is the maximum and sticks)
          const float chk = run(std::false_type{}, ks, xi[u], id[u], v[u], best, ai);
          if (chk != chk) (void)run(std::true_type{}, ks, xi[u], id[u], v[u], best, ai);
          if (on[u]) {
            const size_t t = (size_t)b * N + n0 + q[u];
            *reinterpret_cast<uint2*>(a.mr_arg + t * C + chq[u]) =
                make_uint2((uint32_t)ai[0] | ((uint32_t)ai[1] << 16), (uint32_t)ai[2] | ((uint32_t)ai[3] << 16));
            // the grouped projection's operand buffer XM (T, 2C): m into the m chunk; the x chunk only when x does not live there
            float* o = a.mr_out + t * (size_t)(2 * C) + xm_col(chq[u], Cq);
            if (a.mr_write_x) *reinterpret_cast<float4*>(o) = xi[u];
            *reinterpret_cast<float4*>(o + Cq) = best;
          }
        }
      }
    };
    if (KD <= 9 || k <= 9) pass(std::integral_constant<int, 9>{});          // a 9-entry list holds k <= 9 neighbours
    else if constexpr (KD > 9) pass(std::integral_constant<int, 18>{});
  }
  KNN_TL(31);
}

constexpr int KNN_BUF = 16;        // buffered selection: entries per lane (8 bytes each: 32 KB per workgroup)

template <int KD, bool HAS_RP, int KU, bool GUARD = true, int BUF = 0, bool BF = false, int NWV = NW, bool MRF = false>
static hipError_t launch_tile_v(const KnnArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  // staged query tile: fp32 [cpad][64], or (bf16 form) 64 rows of cp16 + 8 bf16
  const size_t qbytes = BF ? (size_t)QT * (a.cp16 + 8) * 2 : (size_t)a.cpad * QT * sizeof(float);
  if (BF) {                                     // the caller sized `lds` for the fp32 tile: keep only what the merge needs
    const size_t merge = NWV == 1 ? 0 : (size_t)NWV * KD * 64 * 2 * sizeof(float);
    lds = qbytes > merge ? qbytes : merge;
  }
  if (NWV == 1 && !BF) lds = qbytes;            // no merge area
  if (NWV > NW && !BF) {                        // the caller sized the merge area for 4 lists
    const size_t merge = (size_t)NWV * KD * 64 * 2 * sizeof(float);
    if (lds < merge) lds = merge;
  }
  if (BUF > 0) {                                // candidate buffer + the per-wave shared admission bounds behind the queries
    const size_t need = qbytes + (size_t)BUF * 64 * NWV * sizeof(float2) + NWV * 64 * sizeof(float);
    if (lds < need) lds = need;
  }
  if (MRF) {                                    // the final lists of the 64 queries behind the merge area
    const size_t need = (size_t)NWV * KD * 64 * 2 * sizeof(float) + (size_t)a.k * 64 * sizeof(int);
    if (lds < need) lds = need;
  }
  if (lds > 64 * 1024) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_tile_kernel<KD, HAS_RP, KU, GUARD, BUF, BF, NWV, MRF>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;
  }
  hipLaunchKernelGGL((knn_tile_kernel<KD, HAS_RP, KU, GUARD, BUF, BF, NWV, MRF>), grid, dim3(64 * NWV), lds, st, a);
  return hipGetLastError();
}

}  // namespace gkg
