// gkg_knn_common.h — declarations shared by the k-NN kernels (gkg_knn.hip: preparation, fp32 / bf16 tile kernel, split
// merge, host side; gkg_knn_pf.hip: bf16 prefilter + exact re-rank kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gkg_common.h"

// -DKNN_TIMELINE (tools/ubench/knn_timeline.py only): wave-level phase timestamps (s_memtime) of a sample of workgroups, and
// every workgroup's start / end / placement, into the buffer the host put in a.part_v (un-split launches do not use it)
#ifdef KNN_TIMELINE
#define KNN_TL(p)                                                                                                  \
  do {                                                                                                             \
    unsigned long long* tl_ = reinterpret_cast<unsigned long long*>(a.part_v);                                     \
    if (lane == 0 && (blockIdx.x % 97) < 3 && blockIdx.x / 97 < 8 && (p) < 32)                                      \
      tl_[(((blockIdx.x / 97) * 3 + blockIdx.x % 97) * 8 + w) * 32 + (p)] = __builtin_readcyclecounter();           \
    if (lane == 0 && w == 0 && blockIdx.x < 4096 && ((p) == 0 || (p) == 31)) {   /* every workgroup: start, end, where */ \
      tl_[24 * 8 * 32 + blockIdx.x * 4 + ((p) == 0 ? 0 : 1)] = __builtin_readcyclecounter();                       \
      if ((p) == 0) {                                                                                              \
        tl_[24 * 8 * 32 + blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  /* HW_ID */ \
        tl_[24 * 8 * 32 + blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); /* XCC_ID */ \
      }                                                                                                            \
    }                                                                                                              \
  } while (0)
#else
#define KNN_TL(p) do {} while (0)
#endif

namespace gkg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int QT = 64;   // queries per workgroup (one per lane)
constexpr int KT = 32;   // keys per MFMA tile
constexpr int NW = 4;    // waves per workgroup

// ------------------------------------------------------------------------------------------ top-KD list
// Sorted ascending.  A list entry is ONE fp64 key that orders exactly like the pair (distance, key index): the fp32
// distance converted to fp64 (exact; leaves the low 29 mantissa bits zero) with the index stored in those bits —
// complemented for negative distances, where a larger mantissa means a smaller value — so "equal distance -> smaller
// index first" is the plain fp64 '<'.  The sorted insert is then a v_min_f64 + v_max_f64 per slot (fp64 vector ops
// issue at the fp32 rate on CDNA3/4) instead of compare + v_med3 + two selects per slot: 2 instead of 4 vector
// instructions per slot, and the index never has to be moved separately.
//   An fp64 infinity or NaN cannot carry index bits (inf | bits is a NaN), and does not need to: v_min_f64 /
// v_max_f64 return the non-NaN operand, so a candidate whose distance is +inf or NaN (non-finite inputs only) leaves
// the list untouched, exactly like the strict '<' of a scalar insert.  Keys past M are masked with a large FINITE
// |y|^2 (MASKED_SQ) instead of +inf; they can only surface when fewer than k*d real candidates exist at all, and the
// output stage keeps indices in range for that case.  (A distance is never -0.0: |x|^2 >= +0 heads the sum.)
constexpr uint32_t IDX_BITS = 0x1fffffffu;          // 29 bits: key index < 2^29
constexpr float MASKED_SQ = 3.0e38f;

__device__ __forceinline__ double pack_key(float d, int m) {
  // low 29 bits: m for d >= 0, IDX_BITS - m (== IDX_BITS ^ m) for d < 0
  const uint32_t flip = (uint32_t)(__float_as_int(d) >> 31) & IDX_BITS;
  return __longlong_as_double(__double_as_longlong((double)d) + (long long)(flip ^ (uint32_t)m));
}
__device__ __forceinline__ float key_dist(double k) {
  return (float)__longlong_as_double(__double_as_longlong(k) & ~(long long)IDX_BITS);
}
__device__ __forceinline__ int key_index(double k) {
  const long long b = __double_as_longlong(k);
  if ((b & 0x7fffffffffffffffLL) == 0x7ff0000000000000LL) return 0x7fffffff;      // empty slot (+inf)
  const uint32_t lo = (uint32_t)b & IDX_BITS;
  return (int)(b < 0 ? IDX_BITS - lo : lo);
}

// Raw v_min_f64 / v_max_f64: the builtin forms are preceded by a canonicalising v_max_f64 x,x per operand (sNaN
// quieting under IEEE mode) — a third of the insert for nothing, the keys are never NaN.
__device__ __forceinline__ double min_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double max_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int KD>
struct TopList {
  double key[KD];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < KD; ++j) key[j] = (double)INFINITY;
  }
  // Sorted insert; a no-op for lanes whose candidate does not beat their KD-th entry, skipped when no lane of the
  // wave improves.
  template <bool GUARD>
  __device__ __forceinline__ void insert(float d, int m) { insert_key<GUARD>(pack_key(d, m)); }
  template <bool GUARD>
  __device__ __forceinline__ void insert_key(const double k) {
    // GUARD: skip the insert when no lane of the wave improves.  Pays once a wave has streamed a few hundred keys per
    // query (late candidates rarely enter a list); before that it is a compare + branch per candidate for nothing.
    if (GUARD && __builtin_amdgcn_ballot_w64(k < key[KD - 1]) == 0ull) return;
    // new key[j] = max(old key[j-1], min(old key[j], k)), from the top slot down.  Four slots at a time, the mins
    // first and then the maxes, so that no v_max_f64 issues right behind the v_min_f64 it depends on.
    constexpr int U = 4;
#pragma unroll
    for (int j0 = KD - 1; j0 >= 1; j0 -= U) {
      double t[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (j0 - u >= 1) t[u] = min_f64(key[j0 - u], k);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (j0 - u >= 1) key[j0 - u] = max_f64(key[j0 - u - 1], t[u]);
    }
    key[0] = min_f64(key[0], k);
  }
};

// ------------------------------------------------------------------------------------------ main kernel
struct KnnArgs {
  const float* xh;      // (BG, cpad, N) normalised queries
  const float* yh;      // (BG, cpad, M) normalised keys (== xh for the self graph)
  const float* sqx;     // (BG, N)
  const float* sqy;     // (BG, M) (+ >= 32 floats of readable slack)
  const float* relpos;  // (N, M) or null
  int64_t* nn_idx;      // (BG, N, k)
  int64_t* center;      // (BG, N, k) or null
  float* part_v;        // (S, BG, N, KD) partial lists when S > 1
  int* part_i;
  int BG, cpad, N, M, k, dilation, kd;
  int splits, tiles_per_split;
  int nqt;              // query tiles per problem
  int rp_major;         // workgroup -> (problem, query tile) map (knn_map): 0 problem-major, 1 = all problems of one query
                        // tile adjacent on one XCD, 2 = groups of rp_group problems of an XCD interleaved per query tile
  int rp_group;
  const uint16_t* xb;   // BF mode: (BG, N, cp16) / (BG, M, cp16) normalised bf16 token-major copies (prefilter: hi planes)
  const uint16_t* yb;
  int cp16;
  // prefilter mode (knn_pf_kernel)
  const uint16_t* xb_lo;  // lo planes
  const uint16_t* yb_lo;
  float margin;           // 2 * eps: eps bounds |prefilter distance - contract distance| (see knn_pf_kernel)
  int pf_c, pf_nrows, pf_mrows;   // prefilter planes: channel count (|th|^2 rides in channels pf_c, pf_c + 1), padded row counts
  int* wg_flags;          // [gridDim.x] or null.  knn_pf_kernel: sets [blockIdx.x] = 1 (and writes no output) for a query
                          // tile it cannot settle; knn_tile_kernel: when non-null, only flagged workgroups run (clean-up pass)
  // fused aggregation (knn_tile_kernel<..., MRF = true>): token-major fp32 centre / source rows, outputs (see the kernel)
  const float* mr_x;      // (B, N, G * mr_c) as a view: row pitch mr_ldx, chunk mr_xchunk (gkg_common.h "XM layout")
  const float* mr_src;    // (B, M, G * mr_c) likewise (mr_lds, mr_schunk)
  float* mr_out;          // XM (B * N, 2 G mr_c): the grouped projection's [x | m] operand buffer; m is written, x when mr_write_x
  uint16_t* mr_arg;       // (B, N, G * mr_c) winning neighbour rows
  uint16_t* nn16;         // (BG, N, k) compact neighbour lists, or null
  int mr_G, mr_c;
  int mr_ldx, mr_xchunk, mr_lds, mr_schunk, mr_write_x;
};

// the aggregation's maximum rule (gkg_mr.hip `takes`): the first maximum wins, a NaN is the maximum and sticks
__device__ __forceinline__ bool mr_takes(float v, float best) { return v > best || (v != v && best == best); }


// XCD-aware workgroup -> (problem bg, query tile qt) map shared by knn_tile_kernel and knn_pf_kernel (the clean-up pass of the
// latter reads the flags the former wrote per workgroup id, so both must agree).  Workgroups are dealt round-robin over the 8
// XCDs, each with its own L2; lin & 7 is the XCD, jj = lin >> 3 the position in that XCD's dispatch order.
//   0  problem-major: the query tiles of ONE problem are adjacent — its keys enter that L2 once, every problem streams the whole
//      (N, M) positional bias again (B*G x its size per launch: fine while the bias is small);
//   1  bias-major: ALL problems of one query tile are adjacent — the tile's 64 rows of bias enter the L2 once, the keys of every
//      problem re-enter per query tile (narrow groups with a huge bias: pvig_m stage 1, bf16 form);
//   2  interleaved: an XCD owns the problems bg = xcd (mod 8) as in 0, but walks them in groups of rp_group — for every query
//      tile the group's problems are adjacent.  The bias rows are fetched once per group instead of once per problem, and the
//      group's keys (rp_group key sets, sized to stay in the L2) are still reused across the query tiles.
// Returns false for the padding workgroups of the grid.  Placement only affects speed.
__device__ __forceinline__ bool knn_map(const KnnArgs& a, int lin, int& bg, int& qt) {
  const int xcd = lin & 7, jj = lin >> 3, nqt = a.nqt;
  if (a.rp_major == 1) {
    bg = jj % a.BG;
    qt = (jj / a.BG) * 8 + xcd;
  } else if (a.rp_major == 2) {
    const int g = a.rp_group, per = nqt * g;
    const int grp = jj / per, rem = jj - grp * per;
    qt = rem / g;
    bg = (grp * g + (rem - qt * g)) * 8 + xcd;
  } else {
    bg = (jj / nqt) * 8 + xcd;
    qt = jj % nqt;
  }
  return bg < a.BG && qt < nqt;
}

// gkg_knn_pf.hip: launches knn_pf_kernel for list size KD (9, 16, 18, 27 or 36)
hipError_t launch_knn_prefilter(const KnnArgs& a, dim3 grid, int KD, hipStream_t st);
// gkg_knn_bf.hip: the tile kernel's bf16-contraction forms (direct / buffered selection; solo: one wave per query tile)
hipError_t launch_knn_tile_bf(const KnnArgs& a, dim3 grid, size_t lds, int KD, int wbuf, bool solo, hipStream_t st);
// gkg_knn_f32.hip: the tile kernel's fp32-contract forms (mode 0 direct + guard, 1 direct without it, 2 buffered)
hipError_t launch_knn_tile_f32(const KnnArgs& a, dim3 grid, size_t lds, int KD, int mode, hipStream_t st);
// gkg_knn_f32_mr.hip: the same forms with the max-relative aggregation in the epilogue (modes 0-2, lists up to 36 entries)
hipError_t launch_knn_tile_f32_mr(const KnnArgs& a, dim3 grid, size_t lds, int KD, int mode, hipStream_t st);

}  // namespace gkg
