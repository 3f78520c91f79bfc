// gkg_gemm_x6.hip — the Grapher block's dense projections (forward and input-gradient) on the bf16 matrix cores at fp32
// accuracy: every fp32 operand is split into three bf16 terms, x = hi + mid + lo (exactly: each residual is computed in
// fp32 and is representable), and six of the nine cross products are kept
//     a*b ~= ah*bh + [ah*bm + am*bh + am*bm + ah*bl + al*bh]          (dropped: am*bl, al*bm, al*bl <= 2^-24 |a*b|)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16, the bracketed corrections in their own accumulator.  Measured against
// an fp64 evaluation (tools/ubench/gemm_x6_bench.hip, K = 320..1280): max error 5.0e-8 * sum|a b|, mean 4.6e-9 — three
// to four times SMALLER than the fp32-MFMA kernel of gkg_gemm.hip (1.5e-7 / 1.7e-8) and than any fp32 fma chain of that
// length; 6 bf16 MFMAs cost 6*32 cycles per 32x32x16 block where the fp32 MFMA costs 8*64.
//
// Replaces (reference torch_vertex.py:290-306 fc1 / fc2, :57-62 + torch_nn.py:57-69 BasicConv groups=4, :334-360
// FFNLabel — the 1x1 convolutions and their input gradients):
//     forward   Y[m][n]  = sum_k X[m][k]  W[n][k]     (+ train-mode BN column statistics in the epilogue)
//     dgrad     dX[m][n] = sum_k dY[m][k] W[k][n]
// Both are C = A * B^T with A a row-major fp32 activation matrix and B a WEIGHT matrix, so the weights are split once per
// optimiser step by x6_prep_kernel (both orientations, all registered layers, one launch) into
//     planes[p][k/8][n][8]  bf16      p = hi, mid, lo;  n padded to 128, k padded to 32 (zeros)
// and only the activation operand is split inside the GEMM.
//
// Kernel (256 threads = 4 waves, tile 128 rows x 32*NI columns, K-step 32):
//   * waves are stacked 4(M) x 1(N): wave w owns rows [32w, 32w+32), so each A element is split by exactly one wave.  Its
//     rows travel global -> the wave's PRIVATE LDS ring (LDS-DMA, 3 stages, no workgroup barrier on that path) -> registers
//     (2 ds_read_b128 per 16-deep half step; the ring image is XOR-swizzled on the SOURCE address, conflict-free), where
//     the 44-instruction split runs in the shadow of the MFMAs (4-5 VALU per MFMA gap: measured free up to ~5,
//     tools/ubench/mfma_bf16_fill.hip);
//   * the weight planes are DMA'd as 1-KiB contiguous pieces (64 n x 16 B) into a 2-stage image [p][k/8][n] that every
//     wave reads conflict-free with ds_read_b128 — no VALU on the B side at all;
//   * one workgroup barrier per K-step, in the MIDDLE of the step: it certifies stage kt+1 (first read right after it)
//     and releases stage kt's B image and the wave's oldest A slot for the DMA issued next; counted vmcnt leaves the
//     newest A stage in flight across it;
//   * XCD-aware 1-D grid: the column tiles of one 128-row block run back to back on one XCD, so the row block is fetched
//     from HBM into that L2 once.
#include "gkg_common.h"
#include <stdlib.h>

namespace gkg {

typedef float x6_f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 x6_bf16x8;
typedef __bf16 x6_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x6_f32x2 __attribute__((ext_vector_type(2)));

constexpr int X6_NPAD = 128;             // plane rows are padded to this many n (covers NI = 2 and 4 column tiles)

__device__ __forceinline__ unsigned x6_cvt2(float a, float b) {
  x6_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, x6_bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}

// two floats -> packed (hi, mid, lo) bf16 pairs; the residuals are exact fp32 differences
__device__ __forceinline__ void x6_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
#ifdef X6_ABLATE_NOSPLIT
  h = x6_cvt2(x0, x1); m = x6_cvt2(x1, x0); l = h ^ 0x00010001u;
  return;
#endif
  h = x6_cvt2(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = x6_cvt2(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = x6_cvt2(s0, s1);
}

// ---- weight planes ------------------------------------------------------------------------------------------------------
// One descriptor per registered projection weight w (nb, N, K) row-major (N = cout, K = cin per group):
//   forward planes  pf: [nb][3][KCf][NPf][8]   B[n][k] = w[n][k]      NPf = roundup(N, 128), KCf = roundup(K, 32) / 8
//   dgrad planes    pd: [nb][3][KCd][NPd][8]   B[n][k] = w[k][n]      NPd = roundup(K, 128), KCd = roundup(N, 32) / 8
// A work unit is one (k/8, n) pair = 8 floats in, 3 x 16 B out; unit_begin is the running unit count over descriptors.
// kperm (the grouped projection behind the max-relative aggregation, reference torch_vertex.py:57-61 + torch_nn.py:61): the
// activation operand arrives as [x chunk | m chunk] (gkg_common.h "XM layout") instead of the reference's interleave
// [x_0, m_0, x_1, m_1, ...], so the planes hold the weight's INPUT columns in that order: plane position p < K/2 is column 2p
// (an x channel), p >= K/2 column 2 (p - K/2) + 1 (its m).  The weight tensor itself keeps the reference's layout.
struct X6PrepDesc {
  const float* w;
  uint4* pf;
  uint4* pd;
  int nb, N, K;
  int unit_begin;
  int kperm;
};
__device__ __host__ __forceinline__ int x6_kperm_src(int p, int K) { const int h = K >> 1; return p < h ? 2 * p : 2 * (p - h) + 1; }

// Zero jobs riding in the same launch (round 5): a training step begins by clearing its gradient arena and the BN scratch — two
// more 5 us launches in front of the first projection.  Workgroups past the unit blocks clear up to two buffers (16-byte units).
struct X6ZeroJobs {
  uint4* p[2];
  unsigned long long n16[2];        // 16-byte units
  int unit_blocks;                  // workgroups [0, unit_blocks) split weights, the rest clear
};

__global__ __launch_bounds__(256) void x6_prep_kernel(const X6PrepDesc* __restrict__ descs, int ndesc, int total_units, X6ZeroJobs zj) {
  if ((int)blockIdx.x >= zj.unit_blocks) {
    const unsigned long long zb = blockIdx.x - zj.unit_blocks, stride = (unsigned long long)(gridDim.x - zj.unit_blocks) * 256;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      for (unsigned long long i = zb * 256 + threadIdx.x; i < zj.n16[j]; i += stride) zj.p[j][i] = make_uint4(0, 0, 0, 0);
    return;
  }
  __shared__ int begin[256];
  __shared__ int first;
  const int u0 = blockIdx.x * 256;
  if ((int)threadIdx.x < ndesc) begin[threadIdx.x] = descs[threadIdx.x].unit_begin;
  __syncthreads();
  if (threadIdx.x == 0) {                      // the block's 256 consecutive units start in this descriptor
    int d0 = 0;
    while (d0 + 1 < ndesc && begin[d0 + 1] <= u0) ++d0;
    first = d0;
  }
  __syncthreads();
  const int u = u0 + threadIdx.x;
  if (u >= total_units) return;
  int d = first;
  while (d + 1 < ndesc && begin[d + 1] <= u) ++d;
  const X6PrepDesc g = descs[d];
  int v = u - g.unit_begin;
  const int NPf = (g.N + X6_NPAD - 1) / X6_NPAD * X6_NPAD, KCf = (g.K + 31) / 32 * 4;
  const int NPd = (g.K + X6_NPAD - 1) / X6_NPAD * X6_NPAD, KCd = (g.N + 31) / 32 * 4;
  const int fwd_units = g.pf ? g.nb * KCf * NPf : 0;        // an orientation nobody asked for (null planes) has no units
  const bool dgrad = v >= fwd_units;
  if (dgrad) v -= fwd_units;
  const int NP = dgrad ? NPd : NPf, KC = dgrad ? KCd : KCf;
  const int Nn = dgrad ? g.K : g.N, Kk = dgrad ? g.N : g.K;      // extents of (n, k) in this orientation
  const int n = v % NP, kc = (v / NP) % KC, z = v / (NP * KC);
  const float* w = g.w + (size_t)z * g.N * g.K;
  float f[8];
  if (g.kperm) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kc * 8 + j;
      f[j] = (n < Nn && k < Kk) ? (dgrad ? w[(size_t)k * g.K + x6_kperm_src(n, g.K)] : w[(size_t)n * g.K + x6_kperm_src(k, g.K)]) : 0.f;
    }
  } else if (!dgrad && n < Nn && kc * 8 + 8 <= Kk && (g.K & 3) == 0) {          // forward orientation: 8 consecutive floats of row n
    const float4 v0 = *reinterpret_cast<const float4*>(w + (size_t)n * g.K + kc * 8);
    const float4 v1 = *reinterpret_cast<const float4*>(w + (size_t)n * g.K + kc * 8 + 4);
    f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kc * 8 + j;
      f[j] = (n < Nn && k < Kk) ? (dgrad ? w[(size_t)k * g.K + n] : w[(size_t)n * g.K + k]) : 0.f;
    }
  }
  uint4 h, m, l;
  x6_split2(f[0], f[1], h.x, m.x, l.x); x6_split2(f[2], f[3], h.y, m.y, l.y);
  x6_split2(f[4], f[5], h.z, m.z, l.z); x6_split2(f[6], f[7], h.w, m.w, l.w);
  uint4* P = (dgrad ? g.pd : g.pf) + (size_t)z * 3 * KC * NP;
  const size_t plane = (size_t)KC * NP, at = (size_t)kc * NP + n;
  P[at] = h; P[plane + at] = m; P[2 * plane + at] = l;
}

// ---- the GEMM -----------------------------------------------------------------------------------------------------------
struct X6Args {
  const float* A;  size_t a_bstride;  int lda;          // (nb, M, K) activations, row pitch lda floats (lda % 4 == 0)
  const uint4* P;  size_t p_bstride;                    // weight planes of this orientation, uint4 units per batch
  float* C;  size_t c_bstride;  int ldc;
  int M, N, K;
  int NP, KC;                                           // plane geometry (padded n, k/8 chunks)
  int mtiles, ntiles;
  double* sums;  int nslots, nbatch;                    // EPI_BNSTATS: [nslots][nb][2][N] fp64, accumulated with atomics
  // EPI_BNBWD (input-gradient GEMM): the tile it has just produced is the upstream gradient g of the BN (+ GELU) layer that
  // made this GEMM's input in the forward.  Its backward statistics  sum dz, sum dz * yhat  (dz = g * act'(a y + c),
  // yhat = (y - mean) * invstd) are accumulated here, from the accumulators and one read of y, instead of by a stand-alone
  // pass over g and y.  Output column n belongs to BN group n / pco, channel n % pco; y is (pnb, M, pco).
  const float* py; const float* pa; const float* pc; const float* pmean; const float* pinvstd;
  double* psums;  int pco, pact;                        // psums: [pnb][2][pco] fp64 (atomics)
  // split-K (ksplit > 1): workgroup (tile, ks) contracts K-steps [ks * kper, (ks + 1) * kper) and leaves its 128 x BN partial
  // in part[(z * tiles + tile) * ksplit + ks]; the LAST of a tile's ksplit workgroups to arrive (cnt[z * tiles + tile], reset
  // by it) adds the partials in split order — the same bits whoever arrives last — and runs the epilogue on the sum.
  int ksplit, kper;
  float* part;  unsigned* cnt;
  const float* add;                                     // X6_STORE: C = A B^T + add (same shape and pitch as C), or null
};

enum { X6_STORE = 0, X6_BNSTATS = 1, X6_BNBWD = 2 };

__device__ __forceinline__ void x6_chan_merge(double& n, double& mean, double& m2, double nb, double mb, double m2b) {
  if (nb <= 0.0) return;
  const double tot = n + nb;
  const double delta = mb - mean;
  mean += delta * (nb / tot);
  m2 += m2b + delta * delta * (n * nb / tot);
  n = tot;
}

// LDS-DMA of one 1-KiB piece: 64 lanes x 16 B from (sbase + voff) to LDS bytes [lds_dst, lds_dst + 1024).  M0 carries the
// destination and is compiler-reserved: saved and restored inside the statement.  Invisible to the compiler's waitcnt
// bookkeeping — completion is counted by hand (s_waitcnt vmcnt(N)) before the barrier that precedes the reads.
__device__ __forceinline__ void x6_dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// -DX6_TIMELINE (tools/ubench/x6_timeline.py only): phase timestamps (s_memtime) of every 16th workgroup's first wave
#ifdef X6_TIMELINE
__device__ long long* x6_tl_buf;
#define X6_TL(p)                                                                                                      \
  do {                                                                                                                \
    if (x6_tl_buf && tid == 0 && (blockIdx.x & 15) == 0 && (blockIdx.x >> 4) < 4096)                                 \
      x6_tl_buf[(blockIdx.x >> 4) * 8 + (p)] = (long long)__builtin_readcyclecounter();                               \
  } while (0)
#else
#define X6_TL(p) do {} while (0)
#endif

// BN: columns per tile — 32 NI, or 80 with NI = 3 (the last 32-column block is half used).  Every GKGNet width is a multiple of
// 80, so 80-column tiles cover 80 / 160 outputs with one / two workgroups per row block where 64-column tiles need two / three:
// at the short-K stage-1 / stage-2 shapes the launch time follows the number of requests (A is fetched once per column tile,
// B once per tile and K-step), not the arithmetic (profiles/r04_x6_one_vs_two_column_tiles.txt).
template <int NI, int EPI, int BN = 32 * NI>
__global__ __launch_bounds__(256) void gemm_x6_kernel(X6Args g) {
  static_assert(BN <= 32 * NI && BN > 32 * (NI - 1) && BN % 16 == 0 && (12 * BN * 16) % 1024 == 0, "tile width");
  constexpr int BM = 128, BK = 32, SA = 3;
  constexpr bool PARTIAL = BN != 32 * NI;                // lanes r >= BN - 32 (NI - 1) of the last block hold no column
  constexpr int A_STAGE = 4 * 4096, B_STAGE = 12 * BN * 16, B_BASE = SA * A_STAGE;
  constexpr int PIECES = 12 * BN * 16 / 1024;            // 1-KiB pieces of one B stage image
  constexpr int BI = (PIECES + 3) / 4;                   // ... per wave per K-step (wave w: pieces w, w+4, ...), and the
  constexpr int BI_MIN = PIECES / 4;                     // fewest any wave issues: what the counted waits may assume
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = blockIdx.y;
  const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
  // the column tiles and the K splits of one 128-row block run back to back on one XCD
  const int per_row = g.ntiles * g.ksplit, inner = slot % per_row;
  const int tn = inner % g.ntiles, ks = inner / g.ntiles, tm = (slot / per_row) * 8 + xcd;
  if (tm >= g.mtiles) return;
  X6_TL(0);
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = g.M, N = g.N;
  const int K = g.ksplit > 1 ? min(g.K - ks * g.kper * BK, g.kper * BK) : g.K;       // this workgroup's share of the contraction
  const float* A = g.A + (size_t)z * g.a_bstride + (size_t)ks * g.kper * BK;
  const uint4* P = g.P + (size_t)z * g.p_bstride + (size_t)ks * g.kper * 4 * g.NP;
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int nk = (K + BK - 1) / BK;
  const bool ktail = (K & (BK - 1)) != 0;

  // A pieces: LDS slot L = 64 i + lane of the wave's 32 x 128-B image holds chunk (L & 7) ^ ((row >> 1) & 7) of row L >> 3
  unsigned aoff[4];
  int achunk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int L = 64 * i + lane, row = L >> 3, chunk = (L & 7) ^ ((row >> 1) & 7);
    const int gr = min(m0 + 32 * w + row, M - 1);        // rows past the end re-read the last row (never stored)
    aoff[i] = (unsigned)((size_t)gr * g.lda + chunk * 4) * 4u;
    achunk[i] = chunk;
  }
  const size_t plane = (size_t)g.KC * g.NP * 16;
  // B pieces: slot s = 64 piece + lane of the stage image [3 planes][4 chunks][BN] holds n = s % BN of row s / BN (a piece
  // may straddle two rows when BN is not a multiple of 64: the source address is per lane, the destination linear)
  unsigned boff[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int sidx = (w + 4 * i) * 64 + lane, rowid = sidx / BN, n = sidx - rowid * BN, p = rowid >> 2, chunk = rowid & 3;
    boff[i] = (unsigned)(p * plane + ((size_t)chunk * g.NP + n0 + n) * 16);
  }
  auto dma_a = [&](int kt) {
    const char* base = (const char*)A + (size_t)kt * (BK * 4);
    const unsigned dst = lds0 + (kt % SA) * A_STAGE + w * 4096;
    const bool last = ktail && kt == nk - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned o = aoff[i];
      if (last && kt * BK + achunk[i] * 4 >= K) o -= achunk[i] * 16;      // past K: a valid address, masked after the read
      x6_dma16(base, o, dst + i * 1024);
    }
  };
  auto dma_b = [&](int kt) {
    const char* base = (const char*)P + (size_t)kt * 4 * g.NP * 16;
    const unsigned dst = lds0 + B_BASE + (kt & 1) * B_STAGE;
#pragma unroll
    for (int i = 0; i < BI; ++i)
      if ((i + 1) * 4 <= PIECES || w + 4 * i < PIECES) x6_dma16(base, boff[i], dst + (w + 4 * i) * 1024);
  };

  x6_f32x16 acc[NI], accs[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc[j][q] = 0.f; accs[j][q] = 0.f; }

  const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
  int ac[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) { ac[s][0] = r * 128 + (((4 * s + 2 * h) ^ sw) << 4); ac[s][1] = r * 128 + (((4 * s + 2 * h + 1) ^ sw) << 4); }

  float f[8];                        // raw A fragment of the next half step, split in place
  unsigned sh_[4], sm_[4], sl_[4], t0_[4], t1_[4];
  x6_bf16x8 a[2][3], b[2][NI][3];

  auto read_raw = [&](int kt, int s) {
    const char* ab = (const char*)lds + (kt % SA) * A_STAGE + w * 4096;
    const float4 f0 = *(const float4*)(ab + ac[s][0]);
    const float4 f1 = *(const float4*)(ab + ac[s][1]);
    f[0] = f0.x; f[1] = f0.y; f[2] = f0.z; f[3] = f0.w; f[4] = f1.x; f[5] = f1.y; f[6] = f1.z; f[7] = f1.w;
    if (ktail && kt == nk - 1) {
      const int kb = kt * BK + 16 * s + 8 * h;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = kb + j < K ? f[j] : 0.f;
    }
  };
  auto read_b = [&](int kt, int s, int buf) {
    const uint4* bb = lds + (B_BASE + (kt & 1) * B_STAGE) / 16 + r;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) b[buf][j][p] = __builtin_bit_cast(x6_bf16x8, bb[(p * 4 + 2 * s + h) * BN + j * 32]);
  };
  // The split of f[] as 44 single-instruction steps (asm volatile: fixed order, never sunk into another block).  Step i
  // works on pair i & 3, so consecutive steps are independent.
  auto op = [&](int i) {
    const int p = i & 3, o = i >> 2;
    float& x0 = f[2 * p]; float& x1 = f[2 * p + 1];
#ifdef X6_ABLATE_NOSPLIT          // tools/ubench/x6_nosplit.py: what the kernels cost without the operand split (3 of 11 steps kept)
    if (o != 0 && o != 5 && o != 10) return;
#endif
    if (o == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(sh_[p]) : "v"(x0), "v"(x1));
    if (o == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(t0_[p]) : "v"(sh_[p]));
    if (o == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(t1_[p]) : "v"(sh_[p]));
    if (o == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x0) : "v"(t0_[p]));
    if (o == 4) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x1) : "v"(t1_[p]));
    if (o == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(sm_[p]) : "v"(x0), "v"(x1));
    if (o == 6) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(t0_[p]) : "v"(sm_[p]));
    if (o == 7) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(t1_[p]) : "v"(sm_[p]));
    if (o == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x0) : "v"(t0_[p]));
    if (o == 9) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x1) : "v"(t1_[p]));
    if (o == 10) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(sl_[p]) : "v"(x0), "v"(x1));
  };
  auto commit = [&](int buf) {
    a[buf][0] = __builtin_bit_cast(x6_bf16x8, uint4{sh_[0], sh_[1], sh_[2], sh_[3]});
    a[buf][1] = __builtin_bit_cast(x6_bf16x8, uint4{sm_[0], sm_[1], sm_[2], sm_[3]});
    a[buf][2] = __builtin_bit_cast(x6_bf16x8, uint4{sl_[0], sl_[1], sl_[2], sl_[3]});
  };
  // One half step (16 deep): the MFMAs of buffer `cur`; in their gaps the NEXT half step's (stage nkt, half ns) B fragments
  // are read and its raw A fragment (read at the top) is split into buffer cur ^ 1.
  constexpr int SLOTS = 6 * NI, OPS_PER = (44 + SLOTS - 2) / (SLOTS - 1);
  auto half = [&](int cur, bool has_next, int nkt, int ns) {
    const uint4* bb = lds + (B_BASE + (nkt & 1) * B_STAGE) / 16 + r;
    if (has_next) read_raw(nkt, ns);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int si = j * 6 + t;
        // small terms first, the hi*hi product into its own accumulator
        if (t == 0) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][2], b[cur][j][0], accs[j], 0, 0, 0);
        if (t == 1) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], b[cur][j][2], accs[j], 0, 0, 0);
        if (t == 2) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1], b[cur][j][1], accs[j], 0, 0, 0);
        if (t == 3) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1], b[cur][j][0], accs[j], 0, 0, 0);
        if (t == 4) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], b[cur][j][1], accs[j], 0, 0, 0);
        if (t == 5) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], b[cur][j][0], acc[j], 0, 0, 0);
        if (has_next) {
          if (si < 3 * NI) {
            const int jj = si / 3, pp = si % 3;
            b[cur ^ 1][jj][pp] = __builtin_bit_cast(x6_bf16x8, bb[(pp * 4 + 2 * ns + h) * BN + jj * 32]);
          }
          if (si >= 1) {
#pragma unroll
            for (int o = 0; o < OPS_PER; ++o) { const int i = (si - 1) * OPS_PER + o; if (i < 44) op(i); }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (has_next) commit(cur ^ 1);
  };

  // ---- pipeline.  DMA issue order per wave: [A0] [B0] [A1] [B1] [A2], then per step after the barrier [B kt+2] [A kt+3]:
  // at the barrier of step kt the wave needs A kt+1 and B kt+1 and may leave A kt+2 (its newest 4 pieces) in flight.
  dma_a(0); dma_b(0);
  if (nk > 1) { dma_a(1); dma_b(1); }
  if (nk > 2) dma_a(2);
  X6_TL(1);
  if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + BI_MIN) : "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 + BI_MIN) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  X6_TL(2);
  read_raw(0, 0); read_b(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 44; ++i) op(i);
  commit(0);
  for (int kt = 0; kt + 1 < nk; ++kt) {
    half(0, true, kt, 1);                        // s = 0 on buffer 0; fetches and splits (kt, s = 1) into buffer 1
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) dma_b(kt + 2);
    if (kt + 3 < nk) dma_a(kt + 3);
    half(1, true, kt + 1, 0);                    // s = 1 on buffer 1; fetches and splits (kt+1, s = 0) into buffer 0
  }
  half(0, true, nk - 1, 1);
  half(1, false, 0, 0);
  X6_TL(3);

  // X6_BNBWD: the producer's y values of this lane's 16 rows x NI columns are fetched NOW, ahead of the tile's stores and the
  // barrier, so that their latency is off the workgroup's tail
  float yv[EPI == X6_BNBWD ? NI : 1][16];
  float pav[NI], pcv[NI], pmv[NI], piv[NI];
  if (EPI == X6_BNBWD) {
    const int rbase = m0 + 32 * w;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + j * 32 + r;
      pav[j] = pcv[j] = pmv[j] = piv[j] = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) yv[j][q] = 0.f;
      if (n < N && (!PARTIAL || j * 32 + r < BN)) {
        const int pq = n / g.pco, pi = n - pq * g.pco;
        const size_t po = (size_t)pq * g.pco + pi;
        pav[j] = g.pa[po]; pcv[j] = g.pc[po]; pmv[j] = g.pmean[po]; piv[j] = g.pinvstd[po];
        const float* yp = g.py + ((size_t)pq * M + rbase) * (size_t)g.pco + pi;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
          if (rbase + row < M) yv[j][q] = yp[(size_t)row * g.pco];
        }
      }
    }
  }
  // ---- epilogue.  Block j, register q: row 32 w + (q & 3) + 8 (q >> 2) + 4 h, column 32 j + r
  float* C = g.C + (size_t)z * g.c_bstride;
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] += accs[j][q];
  if (g.ksplit > 1) {
    // split-K: the partial goes out in register order (element (j, q) of thread tid at (j * 16 + q) * 256 + tid: every store
    // instruction of a wave is one 256-byte run).  Hand-off without cache-wide fences (an agent-scope release writes the whole
    // L2 back: measured 57-150 us per launch in this place): every payload store is write-through (relaxed agent-scope atomic
    // store = sc1), every storing wave drains its stores, the workgroup meets, ONE lane counts the arrival; the last arrival
    // reads all partials with sc1 loads (they bypass this CU's L1, the only cache that could hold a stale copy).
    const size_t tile_id = ((size_t)z * g.mtiles + tm) * g.ntiles + tn;
    typedef __attribute__((address_space(1))) float gfloat;
    typedef __attribute__((address_space(1))) unsigned gu32;
    gfloat* mine = (gfloat*)(g.part + (tile_id * g.ksplit + ks) * (size_t)(NI * 16 * 256));
    gu32* cnt = (gu32*)(g.cnt + tile_id);
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) __hip_atomic_store(mine + (j * 16 + q) * 256 + tid, acc[j][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(lds);
    if (tid == 0) *flag = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag != (unsigned)(g.ksplit - 1)) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          // no instruction: keeps the loads below the count
    if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
    const gfloat* all = (const gfloat*)(g.part + tile_id * g.ksplit * (size_t)(NI * 16 * 256));
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    for (int s2 = 0; s2 < g.ksplit; ++s2) {                         // range order: the same bits whoever arrives last
      const gfloat* ps = all + (size_t)s2 * (NI * 16 * 256) + tid;
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] += __hip_atomic_load(ps + (j * 16 + q) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + j * 32 + r;
    const bool mine = n < N && (!PARTIAL || j * 32 + r < BN);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + 32 * w + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (m < M && mine) {
        float v = acc[j][q];
        if (EPI == X6_STORE && g.add) v += g.add[(size_t)z * g.c_bstride + (size_t)m * g.ldc + n];
        C[(size_t)m * g.ldc + n] = v;
      }
    }
  }
  X6_TL(4);
#ifdef X6_TIMELINE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  X6_TL(5);
#endif
  if (EPI == X6_BNBWD) {
    // Backward statistics of the PRODUCER's BN (see X6Args): per lane the 16 rows of its column, y read with the same
    // 128-byte row segments the tile was stored with; the two half-waves and the four waves are combined through LDS in
    // fp64 and leave ONE atomic pair per column and tile, like the forward statistics below.
    __syncthreads();                                   // every wave is done with the DMA rings
    float* red = reinterpret_cast<float*>(lds);        // [4 waves][BN][2]
    const int rbase = m0 + 32 * w;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + j * 32 + r;
      float s0 = 0.f, s1 = 0.f;
      if (n < N && (!PARTIAL || j * 32 + r < BN)) {
        const float av = pav[j], cv = pcv[j], mv = pmv[j], iv = piv[j];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
          float dz = rbase + row < M ? acc[j][q] : 0.f;
          if (g.pact == 1) dz *= gelu_grad_f(__builtin_fmaf(av, yv[j][q], cv));
          s0 += dz;
          s1 += dz * ((yv[j][q] - mv) * iv);
        }
      }
      s0 += __shfl_xor(s0, 32);
      s1 += __shfl_xor(s1, 32);
      if (h == 0 && (!PARTIAL || j * 32 + r < BN)) { float* o = red + ((w * BN) + j * 32 + r) * 2; o[0] = s0; o[1] = s1; }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
      double t0 = 0.0, t1 = 0.0;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) { t0 += (double)red[(rg * BN + tid) * 2]; t1 += (double)red[(rg * BN + tid) * 2 + 1]; }
      const int n = n0 + tid, pq = n / g.pco, pi = n - pq * g.pco;
      double* sz = g.psums + (size_t)pq * 2 * g.pco + pi;
      __hip_atomic_fetch_add(sz, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(sz + g.pco, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (EPI != X6_BNSTATS) return;

  // Train-mode BN statistics of the tile's columns (as gkg_gemm.hip's epilogue): per wave (its 32 rows) centred
  // (mean, M2) from registers, Chan-merged over the four waves in fp64, then ONE fp64 atomic pair per column and tile:
  // S += n mean, Q += M2 + n mean^2.  No workgroup waits for another.
  __syncthreads();                                   // every wave is done with the DMA rings
  float* red = reinterpret_cast<float*>(lds);        // [4 waves][BN][2]
  const int rbase = m0 + 32 * w;
  const int cnt = max(0, min(32, M - rbase));
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += rbase + (q & 3) + 8 * (q >> 2) + 4 * h < M ? acc[j][q] : 0.f;
    s += __shfl_xor(s, 32);
    const float mean = cnt > 0 ? s / (float)cnt : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float d = acc[j][q] - mean;
      m2 += rbase + (q & 3) + 8 * (q >> 2) + 4 * h < M ? d * d : 0.f;
    }
    m2 += __shfl_xor(m2, 32);
    if (h == 0 && (!PARTIAL || j * 32 + r < BN)) { float* o = red + ((w * BN) + j * 32 + r) * 2; o[0] = mean; o[1] = m2; }
  }
  __syncthreads();
  if (tid < BN && n0 + tid < N) {
    double n = 0.0, mean = 0.0, m2 = 0.0;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int c = max(0, min(32, M - (m0 + rg * 32)));
      x6_chan_merge(n, mean, m2, (double)c, (double)red[(rg * BN + tid) * 2], (double)red[(rg * BN + tid) * 2 + 1]);
    }
    double* sz = g.sums + ((size_t)(tm % g.nslots) * g.nbatch + z) * 2 * N + n0 + tid;
    __hip_atomic_fetch_add(sz, n * mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(sz + N, m2 + n * mean * mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int NI, int EPI, int BN = 32 * NI>
static hipError_t x6_launch_ni(X6Args a, int nb, hipStream_t st) {
  a.mtiles = (a.M + 127) / 128;
  a.ntiles = (a.N + BN - 1) / BN;
  if (a.ksplit < 1) a.ksplit = 1;
  // A ring + two B stages (+ slack: the unused lanes of a partial last block read past their image row)
  const size_t sh = 3 * 4 * 4096 + 2 * 12 * BN * 16 + (BN != 32 * NI ? 512 : 0);
  // the attribute is per DEVICE (a per-process guard left a second GPU of the same process without it — ADVICE r3)
  static bool once[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !once[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_x6_kernel<NI, EPI, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) once[dev] = true;
  }
  const int groups = (a.mtiles + 7) / 8;
  hipLaunchKernelGGL((gemm_x6_kernel<NI, EPI, BN>), dim3(groups * 8 * a.ntiles * a.ksplit, nb), dim3(256), sh, st, a);
  return hipGetLastError();
}

// Column-tile width (32 NI).  More columns per wave amortise the 44-instruction A split over more MFMAs (NI = 1 is
// VALU-bound, NI = 2 about balanced) but leave fewer workgroups: NI = 2, and NI = 1 when that leaves the chip under-filled
// (few rows: the label branch).  Measured cold-cache per shape with tools/bench_x6.py; NI = 5 (160 columns, one
// workgroup per CU) was tried and is slower everywhere (cfg2 fc1 25.8 vs 22.9 us).
// ---- few rows: the waves of a workgroup split K, not M (round 5) --------------------------------------------------------
// gemm_x6_kernel has a floor of ~10 us per launch whatever the size (2.4 us of prologue DMA issue for three 16 KB A stages and
// two B stages per workgroup, a barrier per K-step, a 128-row epilogue) and needed a cross-workgroup split-K seam (5-13 us) to
// fill the chip with the label branch's 2 560-row matrices: ten launches of 13-28 us per cfg2 step.  Here a workgroup owns
// 32 rows x 64 columns and its four waves take a QUARTER OF THE CONTRACTION each: every wave streams its own A rows through a
// private 3 x 4 KB LDS ring (LDS-DMA, as above), reads its B fragments straight from the weight planes (they are stored in
// MFMA operand order: one 16-byte load per lane and product, L2-resident, one K-step ahead in registers), and never meets
// another wave before the end — no barrier in the loop, no partial tiles through global memory.  The four 32 x 64 partials
// are added through LDS in wave order (deterministic), and the workgroup stores its tile (+ residual) and, EPI_BNSTATS, adds
// the tile's centred column statistics to the fp64 sums like the kernel above.  2 560 x 320 -> 320: 400 workgroups of
// 3 + 3 + 2 + 2 K-steps.
// MI — 32-row blocks per workgroup (and wave): 1 (default), or 2 (opt-in: 64-row tiles halve the plane traffic at one wave per
// SIMD and a 2-stage ring; measured no better, see the launch site).
template <int EPI, int MI>
__global__ __launch_bounds__(256, MI == 1 ? 2 : 1) void gemm_x6_ks_kernel(X6Args g) {
  constexpr int NI = 2, BN = 64, BM = 32 * MI, BK = 32, SA = MI == 1 ? 3 : 2, STAGE = 4096 * MI;
  extern __shared__ uint4 lds[];                       // [4 waves][SA][STAGE] A rings; then [4][BM][64] fp32 partials (aliased)
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = blockIdx.y;
  const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
  const int tn = slot % g.ntiles, tm = (slot / g.ntiles) * 8 + xcd;      // the column tiles of a row block on one XCD
  if (tm >= g.mtiles) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = g.M, N = g.N, K = g.K;
  const int nk_total = (K + BK - 1) / BK;
  const int kbase = nk_total >> 2, krem = nk_total & 3;
  const int cnt = kbase + (w < krem ? 1 : 0);                            // this wave's K-steps
  const int kt0 = w * kbase + min(w, krem);
  const bool ktail = (K & (BK - 1)) != 0;
  const float* A = g.A + (size_t)z * g.a_bstride;
  const uint4* P = g.P + (size_t)z * g.p_bstride;
  const unsigned lds0 = (unsigned)(size_t)lds + w * (SA * STAGE);
  const char* ring = (const char*)lds + w * (SA * STAGE);

  // A pieces (as gemm_x6_kernel): LDS slot L = 64 i + lane of the BM x 128-B image holds chunk (L & 7) ^ ((row >> 1) & 7) of row L >> 3
  unsigned aoff[4 * MI];
  int achunk[4 * MI];
#pragma unroll
  for (int i = 0; i < 4 * MI; ++i) {
    const int L = 64 * i + lane, row = L >> 3, chunk = (L & 7) ^ ((row >> 1) & 7);
    const int gr = min(m0 + row, M - 1);
    aoff[i] = (unsigned)((size_t)gr * g.lda + chunk * 4) * 4u;
    achunk[i] = chunk;
  }
  auto dma_a = [&](int k) {                                              // k: this wave's step index
    const int kt = kt0 + k;
    const char* base = (const char*)A + (size_t)kt * (BK * 4);
    const unsigned dst = lds0 + (k % SA) * STAGE;
    const bool last = ktail && kt == nk_total - 1;
#pragma unroll
    for (int i = 0; i < 4 * MI; ++i) {
      unsigned o = aoff[i];
      if (last && kt * BK + achunk[i] * 4 >= K) o -= achunk[i] * 16;    // past K: a valid address, masked after the read
      x6_dma16(base, o, dst + i * 1024);
    }
  };
  const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
  const size_t plane = (size_t)g.KC * g.NP;                              // uint4 units
  auto load_b = [&](uint4 (&b)[NI][3], int k, int s2) {                  // the 6 operand fragments of half s2 of K-step k
    const uint4* base = P + ((size_t)(kt0 + k) * 4 + h + 2 * s2) * g.NP + n0 + r;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) b[j][p] = base[p * plane + j * 32];
  };
  x6_f32x16 acc[MI][NI], accs[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) { acc[mi][j][q] = 0.f; accs[mi][j][q] = 0.f; }

  uint4 b0f[NI][3], b1f[NI][3], bn0[NI][3], bn1[NI][3];      // both halves of the current K-step and of the next one
  // issue order per wave: A0 A1 A2 B0.0 B0.1 | B(k+1).0 B(k+1).1 (wait A(k), B(k)) step k, A(k+3) | ...
  if (cnt > 0) dma_a(0);
  if (cnt > 1) dma_a(1);
  if (SA > 2 && cnt > 2) dma_a(2);
  if (cnt > 0) { load_b(b0f, 0, 0); load_b(b1f, 0, 1); }
  auto mma = [&](const char* ab, int s2, bool last, int kglob, const uint4 (&bq)[NI][3]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const char* rowp = ab + (32 * mi + r) * 128;
      const float4 f0 = *(const float4*)(rowp + (((4 * s2 + 2 * h) ^ sw) << 4));
      const float4 f1 = *(const float4*)(rowp + (((4 * s2 + 2 * h + 1) ^ sw) << 4));
      float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
      if (last) {
        const int kb = kglob * BK + 16 * s2 + 8 * h;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = kb + e < K ? f[e] : 0.f;
      }
      uint4 ah, am, al;
      x6_split2(f[0], f[1], ah.x, am.x, al.x); x6_split2(f[2], f[3], ah.y, am.y, al.y);
      x6_split2(f[4], f[5], ah.z, am.z, al.z); x6_split2(f[6], f[7], ah.w, am.w, al.w);
      const x6_bf16x8 a0 = __builtin_bit_cast(x6_bf16x8, ah), a1 = __builtin_bit_cast(x6_bf16x8, am), a2 = __builtin_bit_cast(x6_bf16x8, al);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const x6_bf16x8 b0 = __builtin_bit_cast(x6_bf16x8, bq[j][0]), b1 = __builtin_bit_cast(x6_bf16x8, bq[j][1]),
                        b2 = __builtin_bit_cast(x6_bf16x8, bq[j][2]);
        // small terms first, the hi*hi product into its own accumulator (gemm_x6_kernel's order)
        accs[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, accs[mi][j], 0, 0, 0);
        accs[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, accs[mi][j], 0, 0, 0);
        accs[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, accs[mi][j], 0, 0, 0);
        accs[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, accs[mi][j], 0, 0, 0);
        accs[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, accs[mi][j], 0, 0, 0);
        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[mi][j], 0, 0, 0);
      }
    }
  };
  for (int k = 0; k < cnt; ++k) {
    const bool more = k + 1 < cnt;
    if (more) { load_b(bn0, k + 1, 0); load_b(bn1, k + 1, 1); }
    // A(k) and both halves of B(k) have landed once at most the operations issued AFTER B(k).1 are outstanding: A(k - 1 + SA)
    // (issued at the end of step k-1; k == 0: the prologue's A pieces are OLDER than B0) and the 12 loads just issued.  (The
    // compiler's own waits cover the plain loads it can see; this asm wait is for the LDS-DMA pieces it cannot.)
    const int after = (more ? 12 : 0) + ((k >= 1 && k - 1 + SA < cnt) ? 4 * MI : 0);
    if (after == 12 + 4 * MI) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(12 + 4 * MI) : "memory");
    else if (after == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (after == 4 * MI) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * MI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* ab = ring + (k % SA) * STAGE;
    const bool last = ktail && kt0 + k == nk_total - 1;
    mma(ab, 0, last, kt0 + k, b0f);
    mma(ab, 1, last, kt0 + k, b1f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // the ring slot's reads are in registers: it may be refilled
    if (more) {
      // the copies first (their wait sees only this step's loads, a whole step old), then the DMA: behind it the compiler's count
      // of outstanding loads would be short by the DMA pieces
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) { b0f[j][p] = bn0[j][p]; b1f[j][p] = bn1[j][p]; }
    }
    asm volatile("" ::: "memory");
    if (k + SA < cnt) dma_a(k + SA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                                       // every wave is done with its ring
  float* red = reinterpret_cast<float*>(lds);                            // [4][BM][64]
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 32 * mi + (q & 3) + 8 * (q >> 2) + 4 * h;
        red[(w * BM + row) * BN + j * 32 + r] = acc[mi][j][q] + accs[mi][j][q];
      }
  __syncthreads();
  float* C = g.C + (size_t)z * g.c_bstride;
  float v[8 * MI];
#pragma unroll
  for (int u = 0; u < 8 * MI; ++u) {
    const int e = tid + 256 * u, row = e >> 6, col = e & 63;
    v[u] = ((red[e] + red[BM * BN + e]) + red[2 * BM * BN + e]) + red[3 * BM * BN + e];      // wave order: the same bits every run
    const int m = m0 + row, n = n0 + col;
    if (m < M && n < N) {
      float o = v[u];
      if (EPI == X6_STORE && g.add) o += g.add[(size_t)z * g.c_bstride + (size_t)m * g.ldc + n];
      C[(size_t)m * g.ldc + n] = o;
    }
  }
  if (EPI != X6_BNSTATS) return;
  // train-mode BN statistics of the tile's columns: the summed tile back into LDS, one thread per column takes the centred
  // (mean, M2) over the tile's valid rows, ONE fp64 atomic pair per column and tile:  S += n mean,  Q += M2 + n mean^2
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 8 * MI; ++u) red[tid + 256 * u] = v[u];
  __syncthreads();
  if (tid < BN && n0 + tid < N) {
    const int rows = max(0, min(BM, M - m0));
    float sum = 0.f;
    for (int row = 0; row < rows; ++row) sum += red[row * BN + tid];
    const float mean = rows > 0 ? sum / (float)rows : 0.f;
    float m2 = 0.f;
    for (int row = 0; row < rows; ++row) { const float d = red[row * BN + tid] - mean; m2 += d * d; }
    const double nd = (double)rows, md = (double)mean;
    double* sz = g.sums + ((size_t)(tm % g.nslots) * g.nbatch + z) * 2 * N + n0 + tid;
    __hip_atomic_fetch_add(sz, nd * md, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(sz + N, (double)m2 + nd * md * md, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

                                        // bit 2 / bit 3 = always 32-row / 64-row tiles in it (A/B, tests)

template <int EPI, int MI>
static hipError_t x6_launch_ks(X6Args a, int nb, hipStream_t st) {
  a.mtiles = (a.M + 32 * MI - 1) / (32 * MI);
  a.ntiles = (a.N + 63) / 64;
  const size_t sh = MI == 1 ? 4 * 3 * 4096 : 4 * 2 * 8192;      // 48 / 64 KB of A rings (the 32 / 64 KB of partials alias them)
  if (MI == 2) {
    static bool once[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !once[dev]) {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_x6_ks_kernel<EPI, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) once[dev] = true;
    }
  }
  const int groups = (a.mtiles + 7) / 8;
  hipLaunchKernelGGL((gemm_x6_ks_kernel<EPI, MI>), dim3(groups * 8 * a.ntiles, nb), dim3(256), sh, st, a);
  return hipGetLastError();
}

// Split-K workspace (caller-owned, gkg_x6_splitk_workspace_bytes()): [1024 tile counters, zero between launches][partials].
constexpr size_t X6_SK_CNT_BYTES = 4096, X6_SK_MAX_WG = 512;
constexpr size_t X6_SK_BYTES = X6_SK_CNT_BYTES + X6_SK_MAX_WG * (size_t)(2 * 16 * 256 * 4);

template <int EPI>
static hipError_t x6_launch(X6Args a, int nb, hipStream_t st, void* sk_ws = nullptr, size_t sk_bytes = 0, unsigned flags = 0) {
  GkgProfScope prof(GKG_PROF_GEMM_X6, st, 2.0 * a.M * a.N * a.K * nb);
  const int mt = (a.M + 127) / 128;
  // Few rows under a long contraction (the label branch: 2 560 x 1280 -> 320 is 100 workgroups of 40 K-steps on 256 CUs): the
  // contraction is cut into ksplit ranges so that the launch fills the chip's 512 workgroup slots, at least 4 K-steps each
  // (the pipeline's fill is worth about two).  The last arrival per tile sums the partials in split order (deterministic) and
  // runs the normal epilogue, BN statistics included.
  const long long base2 = (long long)mt * ((a.N + 63) / 64) * nb;
  const int nk_total = (a.K + 31) / 32;
  // Few rows (<= 4 096: the label branch): the waves of a workgroup split K instead of M (gemm_x6_ks_kernel) — taken by the _sk
  // entry points, i.e. by callers that asked for the short-matrix forms
  // Where it pays (measured inside the cfg2 step, same box, us, this body vs gemm_x6_kernel with cross-workgroup split-K:
  // profiles/r05_x6_ks_vs_tile_kernel.txt): 2 560 x 320 -> 320 11.0 vs 13.1 (input gradient + residual 11.7 vs 20.7), 640 -> 320
  // 15.2 vs 21.2, 1280 -> 320 24.3 vs 29.6 (input gradient of 320 -> 1280: 25.4 vs 36.5); a tie at 320 -> 640 (16.4); it LOSES
  // where every 32-row workgroup re-reads a wide B (320 -> 1280: 30.0 vs 22.4, its transpose 27.3 vs 21.3: 196 MB of plane
  // traffic) and on the grouped 4 x (160 -> 160) products (15.5 vs 13.5).  Rule: un-grouped, at most 640 output columns.
  // The caller's `flags` (GKG_X6_NO_KS / GKG_X6_FORCE_KS: measurement, tests) override the rule per call.
  if constexpr (EPI != X6_BNBWD) {
    if (sk_ws && !(flags & GKG_X6_NO_KS) && a.M <= 4096 && (long long)((a.M + 31) / 32) * ((a.N + 63) / 64) * nb <= 65535 * 8 &&
        ((flags & GKG_X6_FORCE_KS) || (nb == 1 && a.N <= 640))) {
      // (64-row tiles — half the plane traffic, one wave per SIMD — measured no better on the long contractions they were built
      // for: 2 560 x 1280 -> 320 26.8 vs 24.3 us; what bounds those launches is the wave-serial K loop, not the B planes)
      return x6_launch_ks<EPI, 1>(a, nb, st);
    }
  }
  if (sk_ws && sk_bytes >= X6_SK_BYTES && base2 < 320 && base2 <= 1024 && nk_total >= 8 && EPI != X6_BNBWD) {
    int ks = (int)(X6_SK_MAX_WG / base2);
    if (ks > nk_total / 4) ks = nk_total / 4;
    if (ks > 8) ks = 8;
    if (ks >= 2) {
      a.kper = (nk_total + ks - 1) / ks;
      a.ksplit = (nk_total + a.kper - 1) / a.kper;               // no empty range
      a.cnt = reinterpret_cast<unsigned*>(sk_ws);
      a.part = reinterpret_cast<float*>(reinterpret_cast<char*>(sk_ws) + X6_SK_CNT_BYTES);
      return x6_launch_ni<2, EPI>(a, nb, st);
    }
  }
  int ni = base2 < 160 ? 1 : 2;
  // 80 output columns on many rows (GKGNet-576 stage 1): ONE 80-column tile per row block instead of a full and a quarter
  // 64-column tile — 663 552 rows: 80 -> 80 171 -> 133 us, + statistics 208 -> 161, 320 -> 80 345 -> 279, dgrad 320 <- 80
  // 351 -> 279.  Wider outputs measured equal or slower on 80-column tiles (160: +-3 %, 320 / 400 / 640: 3-8 % slower:
  // profiles/r04_x6_80_column_tiles_ab.txt); the BN-backward epilogue's registers do not fit two waves per SIMD there.
  if constexpr (EPI != X6_BNBWD) {
    if (a.N == 80 && ni == 2) return x6_launch_ni<3, EPI, 80>(a, nb, st);
  }
  if (ni == 1) return x6_launch_ni<1, EPI>(a, nb, st);
  return x6_launch_ni<2, EPI>(a, nb, st);
}

// ---- weight gradient ----------------------------------------------------------------------------------------------------
// dW[m][n] = sum_k dY[k][m] X[k][n]: the contraction runs over the token axis, so BOTH operands are activations (split in
// registers) and both are "k-strided" for the matrix core, whose operand lane (r, h) wants 8 consecutive k of ONE column.
// Here that is 8 dword loads of 8 consecutive rows at a fixed column — lanes r = consecutive columns, so every load
// instruction is two fully used 128-B row segments — straight from global/L2 into registers: no LDS, no barriers, each
// wave streams its own K range.  A workgroup = one 64 x 64 tile of dW x one slab of rows; its 4 waves take a quarter of
// the slab each (2 x 2 blocks of 32 x 32 per wave: 24 MFMAs per 16 rows), add their tiles through LDS and issue ONE set of
// fp32 atomics into the pre-zeroed dW (summation order over slabs is run-dependent, like the vendor library's split-K).
// Loads run three 16-row steps ahead in a 3-deep register ring; the 176-instruction split of the next step's four
// fragments sits in the MFMA gaps (7-8 per gap: this kernel is VALU-bound at ~1.6x the matrix time — for the long-K,
// small-output shapes it exists for it is the HBM stream that sets the time).
struct X6WgradArgs {
  const float* A;  size_t a_bstride;  int lda;       // dY (nb, K, M)
  const float* B;  size_t b_bstride;  int ldb;       // X  (nb, K, N)
  float* C;  size_t c_bstride;  int ldc;             // dW (nb, M, N), zero on entry
  int M, N, K;
  int mtiles, ntiles, splits, rows_per_split;        // rows_per_split % 128 == 0
  int swap;                                          // wide kernel: A / B roles exchanged, C written transposed
  int share, rot;                                    // workgroup -> (slab, tile) map of batched launches (x6w_map); 0, 0 otherwise
  int ubase, urem;                                   // LDS-DMA kernels: slab s = ubase + (s < urem) units of 128 rows (K = 128 * units)
  int kperm;                                         // X arrives as [x chunk | m chunk] (X6PrepDesc::kperm): operand column j of X is
                                                     // dW column x6_kperm_src(j, cin) — applied where the tile is added to dW
};

typedef unsigned x6_u32x4 __attribute__((ext_vector_type(4)));
typedef x6_u32x4 X6WFrag[4][3];               // [A0, A1, B0, B1][hi, mid, lo]: 4 packed bf16 pairs = one MFMA operand each

__device__ __forceinline__ x6_bf16x8 x6w_operand(const x6_u32x4& f) { return __builtin_bit_cast(x6_bf16x8, f); }
// step I (0..175) of the split of four raw fragments into `fn`: block I / 44, and within a block the four pairs
// interleaved (4 independent chains).  Single-instruction asm volatile: fixed order, not sunk, not SLP-packed.
template <int I>
__device__ __forceinline__ void x6w_op(float (&raw)[4][8], X6WFrag& fn, unsigned (&t0)[4], unsigned (&t1)[4]) {
  constexpr int blk = I / 44, u = I % 44, p = u & 3, o = u >> 2;
  float& x0 = raw[blk][2 * p]; float& x1 = raw[blk][2 * p + 1];
  unsigned w;
#ifdef X6_ABLATE_NOSPLIT
  if constexpr (o != 0 && o != 5 && o != 10) return;
#endif
  if constexpr (o == 0) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(x0), "v"(x1)); fn[blk][0][p] = w; }
  if constexpr (o == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(t0[p]) : "v"(fn[blk][0][p]));
  if constexpr (o == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(t1[p]) : "v"(fn[blk][0][p]));
  if constexpr (o == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x0) : "v"(t0[p]));
  if constexpr (o == 4) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x1) : "v"(t1[p]));
  if constexpr (o == 5) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(x0), "v"(x1)); fn[blk][1][p] = w; }
  if constexpr (o == 6) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(t0[p]) : "v"(fn[blk][1][p]));
  if constexpr (o == 7) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(t1[p]) : "v"(fn[blk][1][p]));
  if constexpr (o == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x0) : "v"(t0[p]));
  if constexpr (o == 9) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x1) : "v"(t1[p]));
  if constexpr (o == 10) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(x0), "v"(x1)); fn[blk][2][p] = w; }
}
template <int I, int END>
__device__ __forceinline__ void x6w_ops(float (&raw)[4][8], X6WFrag& fn, unsigned (&t0)[4], unsigned (&t1)[4]) {
  if constexpr (I < END && I < 176) { x6w_op<I>(raw, fn, t0, t1); x6w_ops<I + 1, END>(raw, fn, t0, t1); }
}
// MFMA slot SI (0..23) of a 16-row step: block (i, j) = SI / 6, product t = SI % 6; then 8 split steps in its shadow
template <int SI, bool SPLIT>
__device__ __forceinline__ void x6w_slots(const X6WFrag& fc, X6WFrag& fn, x6_f32x16 (&acc)[2][2], x6_f32x16 (&accs)[2][2],
                                          float (&raw)[4][8], unsigned (&t0)[4], unsigned (&t1)[4]) {
  if constexpr (SI < 24) {
    constexpr int i = SI / 12, j = (SI / 6) & 1, t = SI % 6;
    constexpr int pa = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;         // small terms first, hi*hi last into its own accumulator
    constexpr int pb = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
    if constexpr (t < 5) accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x6w_operand(fc[i][pa]), x6w_operand(fc[2 + j][pb]), accs[i][j], 0, 0, 0);
    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x6w_operand(fc[i][0]), x6w_operand(fc[2 + j][0]), acc[i][j], 0, 0, 0);
    if constexpr (SPLIT) x6w_ops<SI * 8, SI * 8 + 8>(raw, fn, t0, t1);
    __builtin_amdgcn_sched_barrier(0);
    x6w_slots<SI + 1, SPLIT>(fc, fn, acc, accs, raw, t0, t1);
  }
}

__global__ __launch_bounds__(256) void wgrad_x6_kernel(X6WgradArgs g) {
  __shared__ float red[4 * 64 * 64];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = blockIdx.y;
  // XCD-aware: the tiles of one row slab run on one XCD (its rows are fetched into that L2 once)
  const int ntile = g.mtiles * g.ntiles;
  const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
  const int split = (slot / ntile) * 8 + xcd, tile = slot % ntile;
  if (split >= g.splits) return;
  const int m0 = (tile / g.ntiles) * 64, n0 = (tile % g.ntiles) * 64;
  const int r = lane & 31, h = lane >> 5;
  const int rows_per_wave = g.rows_per_split >> 2;
  const long long k_begin = (long long)split * g.rows_per_split + (long long)w * rows_per_wave;
  const int T = rows_per_wave >> 4;                                  // 16-row steps of this wave (even)
  const char* Az = (const char*)(g.A + (size_t)z * g.a_bstride);
  const char* Bz = (const char*)(g.B + (size_t)z * g.b_bstride);
  const long long abytes = (long long)g.K * g.lda * 4, bbytes = (long long)g.K * g.ldb * 4;

  // per-lane byte offset of each fragment's first row (8 h) at its column; the 8 rows j of a fragment come from 8
  // descriptors whose base is advanced by j rows (so the hardware range check stays exact at the end of the tensor);
  // columns outside the matrix get an offset no range check passes
  unsigned off[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = m0 + 32 * i + r, cb = n0 + 32 * i + r;
    off[i] = ca < g.M ? (unsigned)((8 * h * g.lda + ca) * 4) : 0xfffffff0u;
    off[2 + i] = cb < g.N ? (unsigned)((8 * h * g.ldb + cb) * 4) : 0xfffffff0u;
  }
  auto load = [&](float (&raw)[4][8], int s) __attribute__((always_inline)) {
    const long long row0 = k_begin + 16ll * s;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long ao = (row0 + j) * g.lda * 4, bo = (row0 + j) * g.ldb * 4;
      const long long ra_ = abytes - ao, rb_ = bbytes - bo;
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(Az + (ra_ > 0 ? ao : 0)), 0,
                                                                          (int)(ra_ > 0 ? (ra_ > 0x7fffffffll ? 0x7fffffffll : ra_) : 0), 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(Bz + (rb_ > 0 ? bo : 0)), 0,
                                                                          (int)(rb_ > 0 ? (rb_ > 0x7fffffffll ? 0x7fffffffll : rb_) : 0), 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        raw[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, off[i], 0, 0));
        raw[2 + i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, off[2 + i], 0, 0));
      }
    }
  };
  float raw0[4][8], raw1[4][8];                 // raw fragments two 16-row steps deep
  X6WFrag fr0, fr1;
  unsigned t0[4], t1[4];
  x6_f32x16 acc[2][2], accs[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) { acc[i][j][q] = 0.f; accs[i][j][q] = 0.f; }

  // one 16-row step s: MFMAs on `fc`; the gaps split `rs` (step s + 1) into `fn`; then `rs` is reloaded for step s + 3.
  // The last step (no split) is peeled out of the loop: two variants of the MFMA chain inside it would put the
  // accumulators behind phi nodes (the register allocator then copies whole tiles per MFMA).
  load(raw0, 0);
  if (T > 1) load(raw1, 1);
  x6w_ops<0, 176>(raw0, fr0, t0, t1);
  if (T > 2) load(raw0, 2);
  for (int s = 0; s + 2 < T; s += 2) {          // buffer roles alternate: 2 steps per trip (T is even)
    x6w_slots<0, true>(fr0, fr1, acc, accs, raw1, t0, t1);       // step s: splits step s+1 (raw1) ...
    if (s + 3 < T) load(raw1, s + 3);                             // ... then raw1 <- step s+3
    x6w_slots<0, true>(fr1, fr0, acc, accs, raw0, t0, t1);       // step s+1: splits step s+2 (raw0)
    if (s + 4 < T) load(raw0, s + 4);
  }
  x6w_slots<0, true>(fr0, fr1, acc, accs, raw1, t0, t1);         // step T-2
  x6w_slots<0, false>(fr1, fr0, acc, accs, raw0, t0, t1);        // step T-1

  // ---- the four waves' tiles -> LDS -> one atomic add per element
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h, col = 32 * j + r;
        red[w * 4096 + row * 64 + col] = acc[i][j][q] + accs[i][j][q];
      }
  __syncthreads();
  float* C = g.C + (size_t)z * g.c_bstride;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = tid + 256 * u, row = e >> 6, col = e & 63;
    const float v = (red[e] + red[4096 + e]) + (red[8192 + e] + red[12288 + e]);
    if (m0 + row < g.M && n0 + col < g.N)
      __hip_atomic_fetch_add(C + (size_t)(m0 + row) * g.ldc + (g.kperm ? x6_kperm_src(n0 + col, g.N) : n0 + col), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The same workgroup / wave decomposition with the rows staged through LDS: each wave copies the next 16 rows of its own K
// range (64 columns of dY, 64 of X: 8 KiB) global -> a private 4-stage LDS ring by LDS-DMA — eight 16-byte-per-lane
// instructions per step where the register form above needs 32 dword loads (16 address cycles per 256 B each: measured
// texture-address-bound) — and reads the column fragments back with ds_read_b32.  No barriers: a wave only ever reads what
// it copied itself.  Requires every row of the launch to be inside the tensor (the host gives the ragged remainder to the
// register-load kernel) and 16-byte aligned rows; columns past M / N are read from a clamped address and zeroed after the
// LDS read.  Step s issues the DMA of step s + 4, waits until step s + 2 has landed (counted vmcnt) and reads it.
// (split, tile) of workgroup `lin` of a weight-gradient launch.  Default map: slab s runs with ALL its tiles on XCD s % 8 (lin & 7
// is the XCD the hardware dispatches workgroup lin to).  g.share = t > 1 (batched launches of few-slab problems, splits * t == 8):
// t XCDs share a slab and take every t-th tile of it each, so that a problem of 1 / 2 / 4 slabs still spreads over all 8 XCDs
// with no idle workgroup ids.  `rot` rotates the XCD assignment (batched launches: consecutive problems start on different XCDs).
__device__ __forceinline__ bool x6w_map(const X6WgradArgs& g, int lin, int& split, int& tile) {
  const int ntile = g.mtiles * g.ntiles;
  const int xcd = (lin + g.rot) & 7, slot = lin >> 3;
  if (g.share > 1) {
    split = xcd / g.share;
    tile = slot * g.share + xcd % g.share;
    return split < g.splits && tile < ntile;
  }
  split = (slot / ntile) * 8 + xcd;
  tile = slot % ntile;
  return split < g.splits;
}

__device__ __forceinline__ void wgrad_x6_dma_body(const X6WgradArgs& g, const int lin, const int z) {
  extern __shared__ float wlds[];               // [4 waves][4 stages][2 operands][16 rows][64 cols], then reused for the reduce
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int split, tile;
  if (!x6w_map(g, lin, split, tile)) return;
  const int m0 = (tile / g.ntiles) * 64, n0 = (tile % g.ntiles) * 64;
  const int r = lane & 31, h = lane >> 5;
  // the last slab may be shorter (whole 128-row units: 4 waves x 2 steps x 16 rows)
  // balanced slabs of whole 128-row units: the first g.urem slabs hold one unit more (a fixed slab length left the last slab
  // — and, with slab s on XCD s % 8, one XCD — with as little as a third of the others' rows)
  const long long slab0 = 128ll * ((long long)split * g.ubase + min(split, g.urem));
  const int rows_here = 128 * (g.ubase + (split < g.urem ? 1 : 0));
  const int rows_per_wave = rows_here >> 2;
  const long long k_begin = slab0 + (long long)w * rows_per_wave;
  const int T = rows_per_wave >> 4;
  const char* Az = (const char*)(g.A + (size_t)z * g.a_bstride);
  const char* Bz = (const char*)(g.B + (size_t)z * g.b_bstride);
  const unsigned lds0 = (unsigned)(size_t)wlds + w * (4 * 8192);

  // DMA piece p (4 per operand): rows 4p .. 4p+3; lane -> row 4p + (lane >> 4), 16-byte column chunk lane & 15
  const int prow = lane >> 4, pch = lane & 15;
  const int ca = min(m0 + 4 * pch, g.M - 4), cb = min(n0 + 4 * pch, g.N - 4);      // clamped: masked after the LDS read
  const unsigned dA = (unsigned)(((size_t)prow * g.lda + ca) * 4), dB = (unsigned)(((size_t)prow * g.ldb + cb) * 4);
  const unsigned strideA = (unsigned)g.lda * 16u, strideB = (unsigned)g.ldb * 16u;            // 4 rows, bytes
  auto dma = [&](int s) __attribute__((always_inline)) {
    const long long row0 = k_begin + 16ll * s;
    const char* ab = Az + row0 * g.lda * 4;
    const char* bb = Bz + row0 * g.ldb * 4;
    const unsigned dst = lds0 + (s & 3) * 8192;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      x6_dma16(ab, dA + p * strideA, dst + p * 1024);
      x6_dma16(bb, dB + p * strideB, dst + 4096 + p * 1024);
    }
  };
  // fragment (block i of operand o) of stage st: lane (r, h) reads rows 8h .. 8h+7 at column 32 i + r
  const bool va0 = m0 + r < g.M, va1 = m0 + 32 + r < g.M, vb0 = n0 + r < g.N, vb1 = n0 + 32 + r < g.N;
  const bool tail = m0 + 64 > g.M || n0 + 64 > g.N;                                           // uniform
  auto readfr = [&](float (&raw)[4][8], int s) __attribute__((always_inline)) {
    const float* base = wlds + w * (4 * 2048) + (s & 3) * 2048 + (8 * h) * 64 + r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      raw[0][j] = base[j * 64];
      raw[1][j] = base[j * 64 + 32];
      raw[2][j] = base[1024 + j * 64];
      raw[3][j] = base[1024 + j * 64 + 32];
    }
    if (tail) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        raw[0][j] = va0 ? raw[0][j] : 0.f; raw[1][j] = va1 ? raw[1][j] : 0.f;
        raw[2][j] = vb0 ? raw[2][j] : 0.f; raw[3][j] = vb1 ? raw[3][j] : 0.f;
      }
    }
  };
  float raw0[4][8], raw1[4][8];
  X6WFrag fr0, fr1;
  unsigned t0[4], t1[4];
  x6_f32x16 acc[2][2], accs[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) { acc[i][j][q] = 0.f; accs[i][j][q] = 0.f; }

  // prologue: DMA steps 0..3; read steps 0 and 1; split step 0
  dma(0);
  if (T > 1) dma(1);
  if (T > 2) dma(2);
  if (T > 3) dma(3);
  if (T > 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  readfr(raw0, 0);
  if (T > 1) readfr(raw1, 1);
  x6w_ops<0, 176>(raw0, fr0, t0, t1);
  // step s (fr[s&1]): splits raw[(s+1)&1] (step s+1); then DMA step s+4, wait for step s+2, read it into raw[s&1]
  auto tailwork = [&](float (&rawn)[4][8], int s) __attribute__((always_inline)) {
    if (s + 4 < T) dma(s + 4);
    if (s + 2 < T) {
      if (s + 4 < T) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // steps s+3, s+4 may stay in flight
      else if (s + 3 < T) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      readfr(rawn, s + 2);
    }
  };
  for (int s = 0; s + 2 < T; s += 2) {
    x6w_slots<0, true>(fr0, fr1, acc, accs, raw1, t0, t1);     // step s: splits step s+1 (raw1) into fr1
    tailwork(raw0, s);                                            // raw0 <- step s+2
    x6w_slots<0, true>(fr1, fr0, acc, accs, raw0, t0, t1);     // step s+1: splits step s+2 (raw0) into fr0
    tailwork(raw1, s + 1);                                        // raw1 <- step s+3
  }
  x6w_slots<0, true>(fr0, fr1, acc, accs, raw1, t0, t1);
  x6w_slots<0, false>(fr1, fr0, acc, accs, raw0, t0, t1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  __syncthreads();                              // every wave is done with its ring: reuse the LDS for the reduction
  float* red = wlds;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h, col = 32 * j + r;
        red[w * 4096 + row * 64 + col] = acc[i][j][q] + accs[i][j][q];
      }
  __syncthreads();
  float* C = g.C + (size_t)z * g.c_bstride;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = tid + 256 * u, row = e >> 6, col = e & 63;
    const float v = (red[e] + red[4096 + e]) + (red[8192 + e] + red[12288 + e]);
    if (m0 + row < g.M && n0 + col < g.N)
      __hip_atomic_fetch_add(C + (size_t)(m0 + row) * g.ldc + (g.kperm ? x6_kperm_src(n0 + col, g.N) : n0 + col), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(256) void wgrad_x6_dma_kernel(X6WgradArgs g) { wgrad_x6_dma_body(g, blockIdx.x, blockIdx.y); }

// ---- wide form: 64 x 128 output tile per workgroup --------------------------------------------------------------------
// Same decomposition (a workgroup = one tile of dW x one slab of rows, its 4 waves a quarter of the slab each, private LDS
// rings filled by DMA, no barriers in the loop), but a wave owns 2 x 4 blocks of 32 x 32: six fragments per 16-row step feed
// 48 MFMAs, so the split is 264 / 48 = 5.5 vector instructions per MFMA — inside what a bf16 MFMA's shadow hides
// (tools/ubench/mfma_bf16_fill) — where the 64 x 64 form needs 176 / 24 = 7.3 and runs at ~2.6x its matrix time.  The
// products are issued product-major over the eight accumulators (a dependent MFMA is 8 slots away), small terms first, all
// into ONE accumulator per block: 128 accumulator registers + 2 x 48 raw + 2 x 72 split fragments fit the 512 of a
// one-wave-per-SIMD kernel.  Ring: 3 stages x (16 x 64 of dY | 16 x 128 of X) x 4 B = 36 KiB per wave, 144 KiB per workgroup;
// step s issues the DMA of step s + 3 into the stage whose rows were read two steps ago, waits (counted vmcnt) for step
// s + 2 and reads its fragments, which are split during step s + 1.
typedef x6_u32x4 X6WFragW[6][3];              // [A0, A1, B0, B1, B2, B3][hi, mid, lo]

template <int I>
__device__ __forceinline__ void x6ww_op(float (&raw)[6][8], X6WFragW& fn, unsigned (&t0)[4], unsigned (&t1)[4]) {
  constexpr int blk = I / 44, u = I % 44, p = u & 3, o = u >> 2;
  float& x0 = raw[blk][2 * p]; float& x1 = raw[blk][2 * p + 1];
  unsigned w;
  if constexpr (o == 0) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(x0), "v"(x1)); fn[blk][0][p] = w; }
  if constexpr (o == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(t0[p]) : "v"(fn[blk][0][p]));
  if constexpr (o == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(t1[p]) : "v"(fn[blk][0][p]));
  if constexpr (o == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x0) : "v"(t0[p]));
  if constexpr (o == 4) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x1) : "v"(t1[p]));
  if constexpr (o == 5) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(x0), "v"(x1)); fn[blk][1][p] = w; }
  if constexpr (o == 6) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(t0[p]) : "v"(fn[blk][1][p]));
  if constexpr (o == 7) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(t1[p]) : "v"(fn[blk][1][p]));
  if constexpr (o == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x0) : "v"(t0[p]));
  if constexpr (o == 9) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x1) : "v"(t1[p]));
  if constexpr (o == 10) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(x0), "v"(x1)); fn[blk][2][p] = w; }
}
template <int I, int END>
__device__ __forceinline__ void x6ww_ops(float (&raw)[6][8], X6WFragW& fn, unsigned (&t0)[4], unsigned (&t1)[4]) {
  if constexpr (I < END && I < 264) { x6ww_op<I>(raw, fn, t0, t1); x6ww_ops<I + 1, END>(raw, fn, t0, t1); }
}
// MFMA slot SI (0..47) of a 16-row step: product t = SI / 8 (small terms first, hi x hi last), block (i, j) = SI % 8;
// then its share (5 or 6) of the next step's 264 split instructions
template <int SI, bool SPLIT>
__device__ __forceinline__ void x6ww_slots(const X6WFragW& fc, X6WFragW& fn, x6_f32x16 (&acc)[2][4], float (&raw)[6][8],
                                           unsigned (&t0)[4], unsigned (&t1)[4]) {
  if constexpr (SI < 48) {
    constexpr int t = SI / 8, i = (SI % 8) / 4, j = SI % 4;
    constexpr int pa = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;
    constexpr int pb = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x6w_operand(fc[i][pa]), x6w_operand(fc[2 + j][pb]), acc[i][j], 0, 0, 0);
    if constexpr (SPLIT) x6ww_ops<(SI * 11) / 2, ((SI + 1) * 11) / 2>(raw, fn, t0, t1);
    __builtin_amdgcn_sched_barrier(0);
    x6ww_slots<SI + 1, SPLIT>(fc, fn, acc, raw, t0, t1);
  }
}

__device__ __forceinline__ void wgrad_x6_wide_body(const X6WgradArgs& g, const int lin, const int z) {
  extern __shared__ float wlds[];               // [4 waves][3 stages][16 x 64 | 16 x 128], then reused for the reduce
  constexpr int STAGE_BYTES = 16 * 192 * 4, STAGE_F = 16 * 192;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int split, tile;
  if (!x6w_map(g, lin, split, tile)) return;
  const int m0 = (tile / g.ntiles) * 64, n0 = (tile % g.ntiles) * 128;
  const int r = lane & 31, h = lane >> 5;
  // balanced slabs of whole 128-row units: the first g.urem slabs hold one unit more (a fixed slab length left the last slab
  // — and, with slab s on XCD s % 8, one XCD — with as little as a third of the others' rows)
  const long long slab0 = 128ll * ((long long)split * g.ubase + min(split, g.urem));
  const int rows_here = 128 * (g.ubase + (split < g.urem ? 1 : 0));
  const int rows_per_wave = rows_here >> 2;
  const long long k_begin = slab0 + (long long)w * rows_per_wave;
  const int T = rows_per_wave >> 4;              // even, >= 2
  const char* Az = (const char*)(g.A + (size_t)z * g.a_bstride);
  const char* Bz = (const char*)(g.B + (size_t)z * g.b_bstride);
  const unsigned lds0 = (unsigned)(size_t)wlds + w * (3 * STAGE_BYTES);

  // DMA pieces of 1 KiB: dY 4 x (4 rows x 256 B), X 8 x (2 rows x 512 B); columns past M / N come from a clamped address
  // and are zeroed after the LDS read
  const int ca = min(m0 + 4 * (lane & 15), g.M - 4), cb = min(n0 + 4 * (lane & 31), g.N - 4);
  const unsigned dA = (unsigned)(((size_t)(lane >> 4) * g.lda + ca) * 4), dB = (unsigned)(((size_t)(lane >> 5) * g.ldb + cb) * 4);
  const unsigned strideA = (unsigned)g.lda * 16u, strideB = (unsigned)g.ldb * 8u;
  auto dma = [&](int s) __attribute__((always_inline)) {
    const long long row0 = k_begin + 16ll * s;
    const char* ab = Az + row0 * g.lda * 4;
    const char* bb = Bz + row0 * g.ldb * 4;
    const unsigned dst = lds0 + (s % 3) * STAGE_BYTES;
#pragma unroll
    for (int p = 0; p < 4; ++p) x6_dma16(ab, dA + p * strideA, dst + p * 1024);
#pragma unroll
    for (int p = 0; p < 8; ++p) x6_dma16(bb, dB + p * strideB, dst + 4096 + p * 1024);
  };
  bool va[2], vb[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) va[i] = m0 + 32 * i + r < g.M;
#pragma unroll
  for (int j = 0; j < 4; ++j) vb[j] = n0 + 32 * j + r < g.N;
  const bool tail = m0 + 64 > g.M || n0 + 128 > g.N;                                          // uniform
  auto readfr = [&](float (&raw)[6][8], int s) __attribute__((always_inline)) {
    const float* ba = wlds + w * (3 * STAGE_F) + (s % 3) * STAGE_F + (8 * h) * 64 + r;
    const float* bb = wlds + w * (3 * STAGE_F) + (s % 3) * STAGE_F + 1024 + (8 * h) * 128 + r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      raw[0][j] = ba[j * 64];
      raw[1][j] = ba[j * 64 + 32];
#pragma unroll
      for (int q = 0; q < 4; ++q) raw[2 + q][j] = bb[j * 128 + 32 * q];
    }
    if (tail) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        raw[0][j] = va[0] ? raw[0][j] : 0.f; raw[1][j] = va[1] ? raw[1][j] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) raw[2 + q][j] = vb[q] ? raw[2 + q][j] : 0.f;
      }
    }
  };
  float raw0[6][8], raw1[6][8];
  X6WFragW fr0, fr1;
  unsigned t0[4], t1[4];
  x6_f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // prologue: DMA steps 0..2; read steps 0 and 1; split step 0
  dma(0);
  dma(1);
  if (T > 2) dma(2);
  if (T > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  readfr(raw0, 0);
  readfr(raw1, 1);
  x6ww_ops<0, 264>(raw0, fr0, t0, t1);
  // after step s: DMA step s + 3 (its stage held step s, read two steps ago), wait for step s + 2, read it
  auto tailwork = [&](float (&rawn)[6][8], int s) __attribute__((always_inline)) {
    if (s + 3 < T) dma(s + 3);
    if (s + 2 < T) {
      if (s + 3 < T) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      readfr(rawn, s + 2);
    }
  };
  for (int s = 0; s + 2 < T; s += 2) {
    x6ww_slots<0, true>(fr0, fr1, acc, raw1, t0, t1);          // step s: splits step s+1 (raw1) into fr1
    tailwork(raw0, s);                                            // raw0 <- step s+2
    x6ww_slots<0, true>(fr1, fr0, acc, raw0, t0, t1);          // step s+1: splits step s+2 (raw0) into fr0
    tailwork(raw1, s + 1);                                        // raw1 <- step s+3
  }
  x6ww_slots<0, true>(fr0, fr1, acc, raw1, t0, t1);
  x6ww_slots<0, false>(fr1, fr0, acc, raw0, t0, t1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  __syncthreads();                              // every wave is done with its ring: reuse the LDS for the reduction
  float* red = wlds;                            // [4][64][129]: odd pitch, so that the transposed walk below is conflict-free
  constexpr int RP = 129, RW = 64 * RP;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h, col = 32 * j + r;
        red[w * RW + row * RP + col] = acc[i][j][q];
      }
  __syncthreads();
  float* C = g.C + (size_t)z * g.c_bstride;
#pragma unroll 4
  for (int u = 0; u < 32; ++u) {
    // consecutive threads walk the dimension that is contiguous in dW: the tile's columns, or (roles exchanged) its rows
    const int e = tid + 256 * u;
    const int row = g.swap ? (e & 63) : (e >> 7), col = g.swap ? (e >> 6) : (e & 127);
    const int o = row * RP + col;
    const float v = (red[o] + red[RW + o]) + (red[2 * RW + o] + red[3 * RW + o]);
    if (m0 + row < g.M && n0 + col < g.N) {
      // (roles exchanged: the kernel's M side is cin)
      const int wr = g.swap ? n0 + col : m0 + row, wc0 = g.swap ? m0 + row : n0 + col;
      const int wc = g.kperm ? x6_kperm_src(wc0, g.swap ? g.M : g.N) : wc0;
      float* dst = C + (size_t)wr * g.ldc + wc;
      __hip_atomic_fetch_add(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_x6_wide_kernel(X6WgradArgs g) { wgrad_x6_wide_body(g, blockIdx.x, blockIdx.y); }

// ---- batched form: the weight gradients of SEVERAL layers in one launch ------------------------------------------------
// A weight gradient is off the backward's critical path (nothing downstream reads dW), and at this path's sizes each one is
// a launch of 50-400 workgroups on 256 CUs that lives 12-40 us: ten of them per Grapher + GrapherLabel step, a fifth of the
// step.  The host side queues them while the backward runs and issues ONE launch at its end: workgroup ids [begin[i],
// begin[i+1]) belong to problem i (begin[] in multiples of 8, so lin & 7 stays the XCD inside a problem), which runs the
// 64 x 64 or the 64 x 128 body above exactly as its own launch would.  The descriptors travel in the kernel arguments (a
// hipGraph node keeps them by value).
constexpr int X6W_BATCH = 16;
struct X6WgradBatch {
  int n;
  int begin[X6W_BATCH + 1];
  int gridx[X6W_BATCH];                         // workgroups per group z of the problem (a multiple of 8)
  int wide[X6W_BATCH];
  X6WgradArgs p[X6W_BATCH];
};

__global__ __launch_bounds__(256) void wgrad_x6_batch_kernel(X6WgradBatch b) {
  const int bid = blockIdx.x;
  int i = 0;
  while (i + 1 < b.n && bid >= b.begin[i + 1]) ++i;
  const int local = bid - b.begin[i], gx = b.gridx[i];
  const int z = local / gx, lin = local - z * gx;
  const X6WgradArgs g = b.p[i];
#ifdef X6_TIMELINE      // tools/ubench/wgrad_batch_timeline.py: every workgroup's start / end / placement / problem
  if (x6_tl_buf && threadIdx.x == 0 && bid < 4096) {
    long long* t = x6_tl_buf + 4096 * 8 + bid * 4;
    t[0] = (long long)__builtin_readcyclecounter();
    t[2] = ((long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) << 8) | (__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xf);
    t[3] = i * 2 + (b.wide[i] ? 1 : 0);
  }
#endif
  if (b.wide[i]) wgrad_x6_wide_body(g, lin, z);
  else wgrad_x6_dma_body(g, lin, z);
#ifdef X6_TIMELINE
  if (x6_tl_buf && threadIdx.x == 0 && bid < 4096) x6_tl_buf[4096 * 8 + bid * 4 + 1] = (long long)__builtin_readcyclecounter();
#endif
}

inline bool x6_bad_dim(int v) { return v <= 0 || (v & 3) != 0; }
inline size_t x6_plane_units(int n, int k, int nb) {       // uint4 units of one orientation: B[n][k]
  return (size_t)nb * 3 * ((k + 31) / 32 * 4) * ((n + X6_NPAD - 1) / X6_NPAD * X6_NPAD);
}

}  // namespace gkg
using namespace gkg;

#ifdef X6_TIMELINE
extern "C" int gkg_debug_set_x6_timeline(void* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(x6_tl_buf), &buf, sizeof(buf));
}
#endif

namespace gkg {
// Train-mode BN parameters from the fp64 column sums of Y (which EXCLUDES the conv bias, folded here); re-zeroes the sums
// so the scratch buffer is clean for the next projection (it is shared, stream-ordered, by all layers).
//   mean = S/R, var = Q/R - mean^2 (biased), invstd = rsqrt(var + eps), a = gamma*invstd, c = beta - a*mean
//   running_mean <- (1-mom)*rm + mom*(mean + bias), running_var <- (1-mom)*rv + mom*var*R/(R-1)
__global__ __launch_bounds__(256) void bn_sums_finalize_kernel(double* __restrict__ sums, int R, int C,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ bias, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float* __restrict__ a,
                                                               float* __restrict__ cs, float* __restrict__ mean,
                                                               float* __restrict__ invstd, float momentum, float eps,
                                                               long long* __restrict__ nbt, int nslots, int nb) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  const int q = blockIdx.y;
  if (nbt && ch == 0 && q == 0) *nbt += 1;
  if (ch >= C) return;
  // `nslots` copies of the sums ([slot][nb][2][C]): producers with thousands of row tiles spread their atomics over the
  // copies (same-address fp64 atomics serialise); added here in slot order
  double S = 0.0, Q = 0.0;
  for (int sl = 0; sl < nslots; ++sl) {
    double* sz = sums + ((size_t)sl * nb + q) * 2 * C + ch;
    S += sz[0]; Q += sz[C];
    sz[0] = 0.0; sz[C] = 0.0;
  }
  const double m = S / R;
  double var = Q / R - m * m;
  if (var < 0.0) var = 0.0;
  const size_t o = (size_t)q * C + ch;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float av = gamma[o] * is;
  a[o] = av;
  cs[o] = beta[o] - av * (float)m;
  mean[o] = (float)m;
  invstd[o] = is;
  if (running_mean) {
    const float bv = bias ? bias[o] : 0.f;
    running_mean[o] = (1.f - momentum) * running_mean[o] + momentum * ((float)m + bv);
    const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
    running_var[o] = (1.f - momentum) * running_var[o] + momentum * (float)unb;
  }
}

hipError_t launch_bn_sums_finalize(double* stats, int R, int cout, int nb, const float* gamma, const float* beta,
                                   const float* bias, float* running_mean, float* running_var, float* bn_a, float* bn_c,
                                   float* bn_mean, float* bn_invstd, float momentum, float eps, long long* nbt, hipStream_t st,
                                   int nslots) {
  hipLaunchKernelGGL(bn_sums_finalize_kernel, dim3((cout + 255) / 256, nb), dim3(256), 0, st, stats, R, cout, gamma, beta,
                     bias, running_mean, running_var, bn_a, bn_c, bn_mean, bn_invstd, momentum, eps, nbt, nslots, nb);
  return hipGetLastError();
}

}  // namespace gkg

// fp64 column-sum scratch of the projections' BN-statistics epilogue: doubles per buffer (4 groups x 2 x 4096 channels)
extern "C" int gkg_linear_stats_doubles() { return 2 * 4096 * 4; }

extern "C" size_t gkg_x6_planes_bytes(int cin, int cout, int nb, int dgrad) {
  if (x6_bad_dim(cin) || x6_bad_dim(cout) || nb <= 0) return 0;
  return 16 * (dgrad ? x6_plane_units(cin, cout, nb) : x6_plane_units(cout, cin, nb));
}

extern "C" int gkg_x6_prep_desc_bytes(void) { return (int)sizeof(X6PrepDesc); }

// Fills descriptor `index` of a HOST array laid out as gkg_x6_prep_desc_bytes() per entry (one of the two plane pointers may
// be null: that orientation is not produced) and returns the unit count after
// it (pass it as unit_begin of the next entry; the last return value is total_units of gkg_x6_prep_weights).
extern "C" long long gkg_x6_prep_desc_fill(void* host_descs, int index, const float* w, void* planes_fwd, void* planes_dgrad,
                                           int cin, int cout, int nb, long long unit_begin, int kperm) {
  if (!host_descs || !w || (!planes_fwd && !planes_dgrad) || x6_bad_dim(cin) || x6_bad_dim(cout) || nb <= 0 || index < 0) return -1;
  if (kperm && (cin & 1)) return -1;
  X6PrepDesc* d = reinterpret_cast<X6PrepDesc*>(host_descs) + index;
  d->w = w; d->pf = (uint4*)planes_fwd; d->pd = (uint4*)planes_dgrad;
  d->nb = nb; d->N = cout; d->K = cin; d->unit_begin = (int)unit_begin; d->kperm = kperm ? 1 : 0;
  const long long units = (long long)((planes_fwd ? x6_plane_units(cout, cin, nb) : 0) + (planes_dgrad ? x6_plane_units(cin, cout, nb) : 0)) / 3;
  if (unit_begin + units > 0x7fffffffLL) return -1;
  return unit_begin + units;
}

// One launch splits every described weight into its forward and dgrad planes.  `descs_dev`: the descriptor array copied to
// the device (it holds device pointers only).
static int x6_prep_impl(const void* descs_dev, int ndesc, long long total_units, void* z0, size_t z0_bytes, void* z1,
                        size_t z1_bytes, void* stream) {
  if (!descs_dev) return gkg_fail(GKG_ERR_NULL, "gkg_x6_prep_weights: null descriptor array");
  if (ndesc <= 0 || ndesc > 256 || total_units <= 0 || total_units > 0x7fffffffLL) return gkg_fail(GKG_ERR_SHAPE, "gkg_x6_prep_weights: need 1 <= ndesc <= 256 and 0 < total_units < 2^31");
  if (((size_t)z0 & 15) || ((size_t)z1 & 15) || (z0_bytes & 15) || (z1_bytes & 15) || (!z0 && z0_bytes) || (!z1 && z1_bytes))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_x6_prep_weights_zero: buffers to clear must be 16-byte aligned multiples of 16 bytes");
  X6ZeroJobs zj{};
  zj.p[0] = (uint4*)z0; zj.n16[0] = z0_bytes / 16;
  zj.p[1] = (uint4*)z1; zj.n16[1] = z1_bytes / 16;
  zj.unit_blocks = (int)((total_units + 255) / 256);
  const unsigned long long zmax = zj.n16[0] > zj.n16[1] ? zj.n16[0] : zj.n16[1];
  unsigned zblocks = (unsigned)((zmax + 256 * 16 - 1) / (256 * 16));          // ~16 stores per thread
  if (zblocks > 1024) zblocks = 1024;
  hipLaunchKernelGGL(x6_prep_kernel, dim3((unsigned)zj.unit_blocks + zblocks), dim3(256), 0, (hipStream_t)stream,
                     (const X6PrepDesc*)descs_dev, ndesc, (int)total_units, zj);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "x6_prep_kernel");
}

extern "C" int gkg_x6_prep_weights(const void* descs_dev, int ndesc, long long total_units, void* stream) {
  return x6_prep_impl(descs_dev, ndesc, total_units, nullptr, 0, nullptr, 0, stream);
}

// gkg_x6_prep_weights that also CLEARS up to two buffers (zero0 / zero1, 16-byte aligned, byte counts multiples of 16; NULL / 0:
// none) in the same launch: what a training step clears before its first projection anyway — the flat gradient buffer the
// weight-gradient kernels accumulate into and the fp64 BN scratch.
extern "C" int gkg_x6_prep_weights_zero(const void* descs_dev, int ndesc, long long total_units, void* zero0, size_t zero0_bytes,
                                        void* zero1, size_t zero1_bytes, void* stream) {
  return x6_prep_impl(descs_dev, ndesc, total_units, zero0, zero0_bytes, zero1, zero1_bytes, stream);
}

// gkg_linear_bn_fwd with the weights given as forward planes (gkg_x6_prep_weights).  x (nb, R, cin) with row pitch ldx and
// batch stride x_bstride (floats); y (nb, R, cout) contiguous.  Same `train` modes and outputs as gkg_linear_bn_fwd.
extern "C" size_t gkg_x6_splitk_workspace_bytes(void) { return X6_SK_BYTES; }

static int x6_fwd_impl(const float* x, int ldx, size_t x_bstride, const void* planes_fwd, float* y, int R,
                       int cin, int cout, int nb, int train, const float* gamma, const float* beta,
                       const float* bias, float* running_mean, float* running_var,
                       long long* num_batches_tracked, float* bn_a, float* bn_c, float* bn_mean,
                       float* bn_invstd, float momentum, float eps, double* stats, void* sk_ws, size_t sk_bytes, void* stream,
                       unsigned flags = 0) {
  if (!x || !planes_fwd || !y) return gkg_fail(GKG_ERR_NULL, "gkg_linear_bn_fwd_x6: null pointer");
  if (R <= 0 || x6_bad_dim(cin) || x6_bad_dim(cout) || nb <= 0 || nb > 64 || ldx < cin || (ldx & 3) || (x_bstride & 3) ||
      ((size_t)x & 15))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_linear_bn_fwd_x6: need R > 0, cin % 4 == 0, cout % 4 == 0, 1 <= nb <= 64, 16-byte aligned rows");
  if ((size_t)R * ldx * 4 > 0xffffffffull) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_linear_bn_fwd_x6: operand larger than 4 GiB per batch");
  X6Args a{};
  a.A = x; a.a_bstride = x_bstride; a.lda = ldx;
  a.NP = (cout + X6_NPAD - 1) / X6_NPAD * X6_NPAD; a.KC = (cin + 31) / 32 * 4;
  a.P = (const uint4*)planes_fwd; a.p_bstride = (size_t)3 * a.KC * a.NP;
  a.C = y; a.c_bstride = (size_t)R * cout; a.ldc = cout;
  a.M = R; a.N = cout; a.K = cin;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (train) {
    if (!stats) return gkg_fail(GKG_ERR_NULL, "gkg_linear_bn_fwd_x6: training needs the stats scratch");
    if ((size_t)nb * 2 * cout > (size_t)gkg_linear_stats_doubles()) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_linear_bn_fwd_x6: nb * cout too large for the stats scratch");
    if (train != 2) {
      if (!gamma || !beta || !bn_a || !bn_c || !bn_mean || !bn_invstd) return gkg_fail(GKG_ERR_NULL, "gkg_linear_bn_fwd_x6: training needs gamma, beta and the four outputs");
      if ((running_mean == nullptr) != (running_var == nullptr)) return gkg_fail(GKG_ERR_NULL, "gkg_linear_bn_fwd_x6: running stats come in pairs");
    }
    a.sums = stats; a.nbatch = nb;
    // copies of the sums to spread the row tiles' atomics over (train == 2 leaves the sums for a consumer that knows one copy)
    const int fit = gkg_linear_stats_doubles() / (nb * 2 * cout), want = (R + 127) / 128 / 64;
    a.nslots = train == 2 ? 1 : (want < 1 ? 1 : (want > 16 ? 16 : want));
    if (a.nslots > fit) a.nslots = fit;
    e = x6_launch<X6_BNSTATS>(a, nb, st, sk_ws, sk_bytes, flags);
    if (e == hipSuccess && train != 2) {
      e = launch_bn_sums_finalize(stats, R, cout, nb, gamma, beta, bias, running_mean, running_var, bn_a, bn_c, bn_mean,
                                  bn_invstd, momentum, eps, num_batches_tracked, st, a.nslots);
    }
  } else {
    e = x6_launch<X6_STORE>(a, nb, st, sk_ws, sk_bytes, flags);
  }
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "gemm_x6_kernel (forward)");
}

extern "C" int gkg_linear_bn_fwd_x6(const float* x, int ldx, size_t x_bstride, const void* planes_fwd, float* y, int R,
                                    int cin, int cout, int nb, int train, const float* gamma, const float* beta,
                                    const float* bias, float* running_mean, float* running_var,
                                    long long* num_batches_tracked, float* bn_a, float* bn_c, float* bn_mean,
                                    float* bn_invstd, float momentum, float eps, double* stats, void* stream) {
  return x6_fwd_impl(x, ldx, x_bstride, planes_fwd, y, R, cin, cout, nb, train, gamma, beta, bias, running_mean, running_var,
                     num_batches_tracked, bn_a, bn_c, bn_mean, bn_invstd, momentum, eps, stats, nullptr, 0, stream);
}

// The same call with a split-K workspace (gkg_x6_splitk_workspace_bytes() bytes, its first 4 KiB ZERO before the first use;
// every launch leaves them zero again): few-row / long-contraction shapes then run as several K ranges per tile.
extern "C" int gkg_linear_bn_fwd_x6_sk(const float* x, int ldx, size_t x_bstride, const void* planes_fwd, float* y, int R,
                                       int cin, int cout, int nb, int train, const float* gamma, const float* beta,
                                       const float* bias, float* running_mean, float* running_var,
                                       long long* num_batches_tracked, float* bn_a, float* bn_c, float* bn_mean,
                                       float* bn_invstd, float momentum, float eps, double* stats, void* splitk_ws,
                                       size_t splitk_bytes, unsigned flags, void* stream) {
  if (splitk_ws && (splitk_bytes < X6_SK_BYTES || ((size_t)splitk_ws & 15)))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_linear_bn_fwd_x6_sk: need a 16-byte aligned workspace of gkg_x6_splitk_workspace_bytes() bytes (or NULL: no split)");
  return x6_fwd_impl(x, ldx, x_bstride, planes_fwd, y, R, cin, cout, nb, train, gamma, beta, bias, running_mean, running_var,
                     num_batches_tracked, bn_a, bn_c, bn_mean, bn_invstd, momentum, eps, stats, splitk_ws, splitk_bytes, stream, flags);
}

// dx (nb, R, cin) = dy (nb, R, cout; row pitch ldg, batch stride g_bstride) * w, the weights given as dgrad planes.
static int x6_dgrad_impl(const float* dy, int ldg, size_t g_bstride, const void* planes_dgrad, float* dx, int R,
                         int cin, int cout, int nb, const float* residual, void* sk_ws, size_t sk_bytes, void* stream,
                         int ldx = 0, size_t x_bstride = 0, unsigned flags = 0) {
  if (!dy || !planes_dgrad || !dx) return gkg_fail(GKG_ERR_NULL, "gkg_linear_dgrad_x6: null pointer");
  if (ldx == 0) { ldx = cin; x_bstride = (size_t)R * cin; }
  if (ldx < cin || (ldx & 3) || (x_bstride & 3)) return gkg_fail(GKG_ERR_SHAPE, "gkg_linear_dgrad_x6: bad dx pitch / batch stride");
  if (R <= 0 || x6_bad_dim(cin) || x6_bad_dim(cout) || nb <= 0 || nb > 64 || ldg < cout || (ldg & 3) || (g_bstride & 3) ||
      ((size_t)dy & 15))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_linear_dgrad_x6: need R > 0, cin % 4 == 0, cout % 4 == 0, 1 <= nb <= 64, 16-byte aligned rows");
  if ((size_t)R * ldg * 4 > 0xffffffffull) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_linear_dgrad_x6: operand larger than 4 GiB per batch");
  X6Args a{};
  a.A = dy; a.a_bstride = g_bstride; a.lda = ldg;
  a.NP = (cin + X6_NPAD - 1) / X6_NPAD * X6_NPAD; a.KC = (cout + 31) / 32 * 4;
  a.P = (const uint4*)planes_dgrad; a.p_bstride = (size_t)3 * a.KC * a.NP;
  a.C = dx; a.c_bstride = x_bstride; a.ldc = ldx;
  a.M = R; a.N = cin; a.K = cout;
  a.add = residual;
  hipError_t e = x6_launch<X6_STORE>(a, nb, (hipStream_t)stream, sk_ws, sk_bytes, flags);
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "gemm_x6_kernel (dgrad)");
}

extern "C" int gkg_linear_dgrad_x6(const float* dy, int ldg, size_t g_bstride, const void* planes_dgrad, float* dx, int R,
                                   int cin, int cout, int nb, void* stream) {
  return x6_dgrad_impl(dy, ldg, g_bstride, planes_dgrad, dx, R, cin, cout, nb, nullptr, nullptr, 0, stream);
}

// `residual` (nb, R, cin) contiguous or null: dx = dy w + residual — the gradient that reaches the layer's input along a
// skip connection (reference torch_vertex.py:331,354,402 `+ _tmp`), added in the epilogue instead of by a stand-alone kernel.
extern "C" int gkg_linear_dgrad_x6_sk(const float* dy, int ldg, size_t g_bstride, const void* planes_dgrad, float* dx, int R,
                                      int cin, int cout, int nb, const float* residual, void* splitk_ws, size_t splitk_bytes,
                                      int ldx, size_t x_bstride, unsigned flags, void* stream) {
  if (splitk_ws && (splitk_bytes < X6_SK_BYTES || ((size_t)splitk_ws & 15)))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_linear_dgrad_x6_sk: need a 16-byte aligned workspace of gkg_x6_splitk_workspace_bytes() bytes (or NULL: no split)");
  return x6_dgrad_impl(dy, ldg, g_bstride, planes_dgrad, dx, R, cin, cout, nb, residual, splitk_ws, splitk_bytes, stream, ldx, x_bstride, flags);
}

// The same input gradient with the BACKWARD statistics of the producer's BN in the epilogue (X6_BNBWD): dx is the upstream
// gradient of the layer  h = act(BN(py))  that made this projection's input; psums [pnb][2][pco] fp64 receives (atomics)
// sum dz and sum dz * yhat per channel — what gkg_bn_bwd_atomic's statistics pass would compute from dx and py.  Un-grouped
// projections only (nb == 1): dx (R, cin), cin == pnb * pco, py (pnb, R, pco).
extern "C" int gkg_linear_dgrad_x6_bnbwd(const float* dy, int ldg, const void* planes_dgrad, float* dx, int R, int cin, int cout,
                                         const float* py, const float* pa, const float* pc, const float* pmean,
                                         const float* pinvstd, double* psums, int pnb, int pco, int pact, void* stream) {
  if (!dy || !planes_dgrad || !dx || !py || !pa || !pc || !pmean || !pinvstd || !psums)
    return gkg_fail(GKG_ERR_NULL, "gkg_linear_dgrad_x6_bnbwd: null pointer");
  if (R <= 0 || x6_bad_dim(cin) || x6_bad_dim(cout) || ldg < cout || (ldg & 3) || ((size_t)dy & 15) || pnb <= 0 || pco <= 0 ||
      (long long)pnb * pco != cin || (pact != 0 && pact != 1))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_linear_dgrad_x6_bnbwd: need R > 0, cin % 4 == 0, cout % 4 == 0, cin == pnb * pco, 16-byte aligned rows");
  if ((size_t)R * ldg * 4 > 0xffffffffull) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_linear_dgrad_x6_bnbwd: operand larger than 4 GiB");
  X6Args a{};
  a.A = dy; a.a_bstride = 0; a.lda = ldg;
  a.NP = (cin + X6_NPAD - 1) / X6_NPAD * X6_NPAD; a.KC = (cout + 31) / 32 * 4;
  a.P = (const uint4*)planes_dgrad; a.p_bstride = (size_t)3 * a.KC * a.NP;
  a.C = dx; a.c_bstride = (size_t)R * cin; a.ldc = cin;
  a.M = R; a.N = cin; a.K = cout;
  a.py = py; a.pa = pa; a.pc = pc; a.pmean = pmean; a.pinvstd = pinvstd; a.psums = psums; a.pco = pco; a.pact = pact;
  hipError_t e = x6_launch<X6_BNBWD>(a, 1, (hipStream_t)stream);
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "gemm_x6_kernel (dgrad + BN backward statistics)");
}

// dw (nb, cout, cin) += dy^T x over the R rows; dw must be ZERO on entry (the slabs of rows are added with fp32 atomics).
// dy (nb, R, cout) row pitch ldg / batch stride g_bstride, x (nb, R, cin) row pitch ldx / batch stride x_bstride (floats).
namespace gkg {
struct X6WgradPlan {
  X6WgradArgs main;              // the LDS-DMA launch over the whole 128-row units (main.K rows; 0: none)
  bool wide;
  int grid_x;                    // workgroups per group of the main launch
  X6WgradArgs rest;              // the register-load launch over the ragged remainder (rest.K rows; 0: none)
  int rest_grid_x;
};

static int x6_wgrad_set_attr(bool wide) {
  // the attribute is per DEVICE: the guard is indexed by the current device (a per-process guard left a second GPU of the
  // same process without it — ADVICE r3)
  static bool once[3][64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t sh_w = (size_t)4 * 3 * 16 * 192 * 4, sh_d = (size_t)4 * 4 * 8192;
  if (dev < 0 || dev >= 64 || !once[wide][dev]) {
    hipError_t e = hipFuncSetAttribute(wide ? (const void*)wgrad_x6_wide_kernel : (const void*)wgrad_x6_dma_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(wide ? sh_w : sh_d));
    if (e != hipSuccess) return gkg_fail_hip(e, "wgrad_x6 DMA kernel (attribute)");
    if (dev >= 0 && dev < 64) once[wide][dev] = true;
  }
  if (dev < 0 || dev >= 64 || !once[2][dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad_x6_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_w);
    if (e != hipSuccess) return gkg_fail_hip(e, "wgrad_x6 batch kernel (attribute)");
    if (dev >= 0 && dev < 64) once[2][dev] = true;
  }
  return 0;
}

// Tile shape, operand roles and row slabs of one weight gradient.  `batched` > 0: the launch shares the chip with other
// problems (gkg_linear_wgrad_x6_batch) — slabs of ~`batched` units whatever the tile count, few-slab problems spread over the
// XCDs by tiles.
static int x6_wgrad_plan(const float* dy, int ldg, size_t g_bstride, const float* x, int ldx, size_t x_bstride, float* dw, int R,
                         int cin, int cout, int nb, int batched, X6WgradPlan& pl, int kperm = 0) {
  if (!dy || !x || !dw) return gkg_fail(GKG_ERR_NULL, "gkg_linear_wgrad_x6: null pointer");
  if (kperm && (cin & 1)) return gkg_fail(GKG_ERR_SHAPE, "gkg_linear_wgrad_x6: kperm needs an even cin");
  if (R <= 0 || cin <= 0 || cout <= 0 || nb <= 0 || nb > 64 || ldg < cout || ldx < cin)
    return gkg_fail(GKG_ERR_SHAPE, "gkg_linear_wgrad_x6: need R, cin, cout > 0, 1 <= nb <= 64, pitches >= widths");
  if ((size_t)ldg * 4 * 16 > 0x7fffffffull || (size_t)ldx * 4 * 16 > 0x7fffffffull) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_linear_wgrad_x6: row pitch too large");
  X6WgradArgs a{};
  a.A = dy; a.a_bstride = g_bstride; a.lda = ldg;
  a.B = x; a.b_bstride = x_bstride; a.ldb = ldx;
  a.C = dw; a.c_bstride = (size_t)cout * cin; a.ldc = cin;
  a.M = cout; a.N = cin; a.K = R;
  a.kperm = kperm ? 1 : 0;
  a.mtiles = (cout + 63) / 64; a.ntiles = (cin + 63) / 64;
  // whole 128-row units go through an LDS-DMA kernel when the rows are 16-byte aligned; the ragged rest (and everything,
  // when they are not) through the register-load kernel
  const bool aligned = (cin & 3) == 0 && (cout & 3) == 0 && (ldg & 3) == 0 && (ldx & 3) == 0 && (g_bstride & 3) == 0 &&
                       (x_bstride & 3) == 0 && ((size_t)dy & 15) == 0 && ((size_t)x & 15) == 0 &&
                       (size_t)R * ldg * 4 < 0xffffffffull && (size_t)R * ldx * 4 < 0xffffffffull;
  // 64 x 128 tiles (wgrad_x6_wide_kernel) when one of the two widths pads to 128 at no more than a tenth of extra work
  // (measured in the cfg2 step, wide vs 64 x 64: 640 -> 320 42.6 -> 35.6 us, 1280 -> 320 37.9 -> 30.9, but 320 -> 320 at
  // 20 % padding 22.3 -> 24.3); the 128-wide side is cin, or — roles of the operands exchanged, dW written transposed — cout
  const long long padn = (long long)((cin + 127) / 128) * 128 * 100 / ((long long)a.ntiles * 64);
  const long long padm = (long long)((cout + 127) / 128) * 128 * 100 / ((long long)a.mtiles * 64);
  bool wide = aligned && (padn <= 110 || padm <= 110);
  bool swap = wide && padm < padn;
  if (batched > 0 && aligned) {
    // inside a batched launch the tile count of ONE problem no longer decides how well the chip is filled, only the matrix
    // time does: the 64 x 128 body runs at about its MFMA time, the 64 x 64 body at ~1.5x (7.3 vs 5.5 split instructions per
    // MFMA) — padding up to 128 pays until it costs that much (160 x 160: 3 x 2 wide tiles against 3 x 3; 320 x 320: 5 x 3
    // against 5 x 5)
    const long long c64 = (long long)a.mtiles * a.ntiles * 36;
    const long long cwn = (long long)a.mtiles * ((cin + 127) / 128) * 48, cwm = (long long)a.ntiles * ((cout + 127) / 128) * 48;
    wide = (cwn < cwm ? cwn : cwm) <= c64;
    swap = wide && cwm < cwn;
  }
  const int main_mtiles = !wide ? a.mtiles : (swap ? a.ntiles : a.mtiles);
  const int main_ntiles = !wide ? a.ntiles : (swap ? (cout + 127) / 128 : (cin + 127) / 128);
  const int tiles = main_mtiles * main_ntiles * nb, units = R / 128;
  int splits = 1, share = 0;
  // the stand-alone rule (below): the slab count that minimises rounds x (units per slab + fixed cost) over an XCD's 32 CUs
  int sp_alone = 1;
  {
    long long best = -1;
    for (int sp = 1; sp <= units && sp <= 4096; ++sp) {
      const long long per_xcd = (long long)tiles * ((sp + 7) / 8);
      const long long rounds = (per_xcd + 31) / 32;
      const long long cost = rounds * ((units + sp - 1) / sp + 6);
      if (best < 0 || cost < best) { best = cost; sp_alone = sp; }
    }
  }
  // a problem that fills the chip on its own (GKGNet-576's stage-1 / stage-2 weight gradients: thousands of 128-row units) keeps
  // its stand-alone slabs inside a batch too — ~20-unit slabs doubled its workgroup count (cfg4: 12.6 -> 13.3 ms of weight
  // gradients per step)
  if (batched && (long long)tiles * sp_alone >= 512) batched = 0;
  if (batched) {
    // ~`batched` (default 20) units of 128 rows per workgroup (its fixed cost — ring fill, LDS reduction, the tile's atomics — is worth about 6):
    // 1 / 2 / 4 slabs shared by 8 / 4 / 2 XCDs each, or a multiple of 8 slabs on one XCD each
    const int want = (units + batched - 1) / batched;
    if (want <= 1) splits = 1;
    else if (want <= 2) splits = 2;
    else if (want <= 5) splits = 4;
    else splits = (want + 7) / 8 * 8;
    if (splits > units) splits = units > 0 ? units : 1;
    if (splits < 8) { while (8 % splits) --splits; share = 8 / splits; }
  } else {
    // Slabs of whole 128-row units (4 waves x 2 steps x 16 rows).  One workgroup per CU fits (128 / 144 KiB of LDS rings) and
    // slab s runs with all its tiles on XCD s % 8 (its rows are fetched into that L2 once), so an XCD's 32 CUs work in rounds
    // over tiles x ceil(slabs / 8) workgroups: pick the slab count that minimises rounds x (units per slab + fixed cost), the
    // fixed cost (ring fill, LDS reduction, the tile's atomics) being worth about 6 units of streaming.  (Counting rounds over
    // the whole chip instead put 36 workgroups on two XCDs at 6 tiles x 42 slabs: 245 -> 393 us at 663 552 x 160 -> 80.)
    splits = sp_alone;
  }
  a.rows_per_split = units > 0 ? (units + splits - 1) / splits * 128 : 128;
  const int main_rows = aligned ? units * 128 : 0;
  pl.wide = wide;
  pl.main = a;
  pl.main.K = 0;
  pl.grid_x = 0;
  if (main_rows > 0) {
    X6WgradArgs m = a;
    m.mtiles = main_mtiles; m.ntiles = main_ntiles;
    if (swap) {                                    // the kernel's "A" (64-wide side) is x, its "B" (128-wide side) dy
      m.A = x; m.a_bstride = x_bstride; m.lda = ldx; m.M = cin;
      m.B = dy; m.b_bstride = g_bstride; m.ldb = ldg; m.N = cout;
      m.swap = 1;
    }
    m.K = main_rows; m.splits = splits < units ? splits : units;
    m.ubase = units / m.splits; m.urem = units % m.splits;
    if (share > 1 && m.splits * share == 8) {
      m.share = share;
      pl.grid_x = (m.mtiles * m.ntiles + share - 1) / share * 8;
    } else {
      pl.grid_x = (m.splits + 7) / 8 * 8 * m.mtiles * m.ntiles;
    }
    pl.main = m;
  }
  const int done = main_rows, rest = R - done;
  pl.rest = a;
  pl.rest.K = 0;
  pl.rest_grid_x = 0;
  if (rest > 0) {
    X6WgradArgs t = a;
    t.A = dy + (size_t)done * ldg; t.B = x + (size_t)done * ldx; t.K = rest;
    const int tunits = (rest + 127) / 128;
    int ts = splits > tunits ? tunits : splits;
    t.rows_per_split = (tunits + ts - 1) / ts * 128;
    t.splits = (rest + t.rows_per_split - 1) / t.rows_per_split;
    pl.rest = t;
    pl.rest_grid_x = (t.splits + 7) / 8 * 8 * a.mtiles * a.ntiles;
  }
  return 0;
}
}  // namespace gkg

extern "C" int gkg_linear_wgrad_x6(const float* dy, int ldg, size_t g_bstride, const float* x, int ldx, size_t x_bstride,
                                   float* dw, int R, int cin, int cout, int nb, int kperm, void* stream) {
  X6WgradPlan pl;
  int rc = x6_wgrad_plan(dy, ldg, g_bstride, x, ldx, x_bstride, dw, R, cin, cout, nb, 0, pl, kperm);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  if (pl.main.K > 0) {
    if ((rc = x6_wgrad_set_attr(pl.wide))) return rc;
    const size_t sh = pl.wide ? (size_t)4 * 3 * 16 * 192 * 4 : (size_t)4 * 4 * 8192;
    const dim3 grid(pl.grid_x, nb);
    if (pl.wide) hipLaunchKernelGGL(wgrad_x6_wide_kernel, grid, dim3(256), sh, st, pl.main);
    else hipLaunchKernelGGL(wgrad_x6_dma_kernel, grid, dim3(256), sh, st, pl.main);
    e = hipGetLastError();
    if (e != hipSuccess) return gkg_fail_hip(e, "wgrad_x6 DMA kernel");
  }
  if (pl.rest.K > 0) {
    hipLaunchKernelGGL(wgrad_x6_kernel, dim3(pl.rest_grid_x, nb), dim3(256), 0, st, pl.rest);
    e = hipGetLastError();
  }
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "wgrad_x6_kernel");
}

// The weight gradients of n layers in ONE launch per 16 problems (wgrad_x6_batch_kernel): problem i is what
// gkg_linear_wgrad_x6(p[i].dy, ..., p[i].nb) would compute — every dw ZERO on entry.  Problems whose rows are not whole
// 128-row units / 16-byte aligned fall back to their own launches.  Largest problems first (the tail of the launch is made of
// the short ones).
extern "C" int gkg_linear_wgrad_x6_batch(const GkgWgradProblem* p, int n, int units_per_slab, void* stream) {
  if (!p || n <= 0) return gkg_fail(GKG_ERR_NULL, "gkg_linear_wgrad_x6_batch: no problems");
  if (units_per_slab < 0 || units_per_slab > 4096) return gkg_fail(GKG_ERR_SHAPE, "gkg_linear_wgrad_x6_batch: units_per_slab out of range");
  if (units_per_slab == 0) units_per_slab = 20;
  hipStream_t st = (hipStream_t)stream;
  int order[256];
  if (n > 256) return gkg_fail(GKG_ERR_SHAPE, "gkg_linear_wgrad_x6_batch: at most 256 problems per call");
  for (int i = 0; i < n; ++i) order[i] = i;
  auto work = [&](int i) { return (double)p[i].R * p[i].cin * p[i].cout * p[i].nb; };
  for (int i = 1; i < n; ++i)                     // insertion sort, descending work
    for (int j = i; j > 0 && work(order[j]) > work(order[j - 1]); --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  X6WgradBatch b{};
  int rot = 0;
  auto flush = [&]() -> int {
    if (b.n == 0) return 0;
    int rc = x6_wgrad_set_attr(true);
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_x6_batch_kernel, dim3(b.begin[b.n]), dim3(256), (size_t)4 * 3 * 16 * 192 * 4, st, b);
    hipError_t e = hipGetLastError();
    b.n = 0;
    return e == hipSuccess ? 0 : gkg_fail_hip(e, "wgrad_x6_batch_kernel");
  };
  for (int oi = 0; oi < n; ++oi) {
    const GkgWgradProblem& q = p[order[oi]];
    X6WgradPlan pl;
    int rc = x6_wgrad_plan(q.dy, q.ldg, q.g_bstride, q.x, q.ldx, q.x_bstride, q.dw, q.R, q.cin, q.cout, q.nb, units_per_slab, pl, q.kperm);
    if (rc) return rc;
    if (pl.main.K > 0) {
      const int i = b.n;
      pl.main.rot = rot;
      rot = (rot + (pl.main.share > 1 ? 0 : pl.main.splits)) & 7;
      b.p[i] = pl.main; b.wide[i] = pl.wide ? 1 : 0; b.gridx[i] = pl.grid_x;
      b.begin[i + 1] = b.begin[i] + pl.grid_x * q.nb;
      b.n = i + 1;
      if (b.n == X6W_BATCH && (rc = flush())) return rc;
    }
    if (pl.rest.K > 0) {
      hipLaunchKernelGGL(wgrad_x6_kernel, dim3(pl.rest_grid_x, q.nb), dim3(256), 0, st, pl.rest);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return gkg_fail_hip(e, "wgrad_x6_kernel");
    }
  }
  return flush();
}
