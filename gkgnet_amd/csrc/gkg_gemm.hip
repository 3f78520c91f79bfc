// gkg_gemm.hip — fp32 matrix-core (v_mfma_f32_32x32x2_f32) projection kernels of the Grapher block with the
// batch-norm passes fused in (gfx950).
//
// Replaces, for the dense 1x1 projections of the block (reference torch_vertex.py:290-306 fc1/fc2, :57-62 + torch_nn.py:57-69
// BasicConv groups=4, torch_vertex.py:334-360 FFNLabel):
//   forward   Y = X W^T                      + train-mode BN statistics in the epilogue (per-tile (mean, M2), merged by the
//                                              LAST-arriving workgroup of each column tile, which also emits the BN scale /
//                                              shift, the saved mean / invstd and the running-stat update)
//   dgrad     dX = dY W,   dY = alpha*dz + beta*y + gamma   (BN backward-apply as the A-operand PROLOGUE: dY is never stored)
//   wgrad     dW = dY^T X  (same prologue), split over the token axis; the last-arriving split of an output tile adds the
//                                              partial tiles in a fixed order (deterministic)
//
// One kernel template.  C[m][n] = sum_k A(m,k) B(n,k); an operand tile is staged in LDS in one of two images:
//   KQ  global layout k-contiguous (X[m][k], W[n][k]):  LDS [BK/4][BT+1] float4 — one ds_write_b128 per loaded float4,
//       one ds_read_b128 per lane feeds FOUR MFMAs (lane half h reads k-quad 2*ko+h; MFMA j consumes element j)
//   KM  global layout m-contiguous (W[k][n] for dgrad, dY[t][m] / X[t][n] for wgrad): LDS [BK][BT] floats, float4 writes,
//       ds_read_b32 fragments at row k = 8*ko + 4*h + j  (the same k <-> (h, j) map as KQ, so the images mix freely)
// fp32 MFMA is an exact fp32 fma chain; the contraction order inside a k-octet is permuted relative to 0..K-1, which the
// projections' 1e-3 contract allows (the graph kernels' bit-exact contract is untouched: they do not use this file).
#include "gkg_common.h"

namespace gkg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBK = 32;                  // default k per LDS stage
enum { LAY_KQ = 0, LAY_KM = 1 };
enum { EPI_STORE = 0, EPI_BNSTATS = 1, EPI_SPLITK = 2, EPI_SPLITK_ATOMIC = 3 };

struct GemmArgs {
  // operands (batched over blockIdx.z / nb when nb > 1; split-K uses blockIdx.z for the split instead)
  const float* A;  const float* A2;  size_t a_bstride;  int lda;     // A2: second tensor of the dual prologue (y)
  const float* B;  size_t b_bstride;  int ldb;
  float* C;  size_t c_bstride;  int ldc;
  int M, N, K;
  int mtiles, ntiles, nbatch;  // tile grid (the launch grid is 1-D: see the XCD-aware map in the kernel)
  // dual prologue  A' = coef[0][ch]*A + coef[1][ch]*A2 + coef[2][ch]   (ch = k for LAY_KQ, m for LAY_KM)
  const float* coef;  int coef_stride;  size_t coef_bstride;          // [3][coef_stride] per batch
  // EPI_BNSTATS
  double* sums;                // [nb][2][N]  column sum / sum of squares, accumulated with fp64 atomics (zero on entry)
  unsigned* counters;          // EPI_SPLITK: [output tiles], zero between launches
  // EPI_SPLITK
  int splits, k_per_split;     // k_per_split % GBK == 0
  float* kpart;                // [tiles][splits][BM*BN]
};

template <int BT, int LAY, int BK> struct TileGeom;
template <int BT, int BK> struct TileGeom<BT, LAY_KQ, BK> {
  static constexpr int F4 = (BK / 4) * (BT + 1);          // float4 elements
  static constexpr int PER_THREAD = BT * (BK / 4) / 256;  // float4 loads per thread per stage
};
template <int BT, int BK> struct TileGeom<BT, LAY_KM, BK> {
  static constexpr int F4 = BK * BT / 4;
  static constexpr int PER_THREAD = BK * BT / 4 / 256;
};

// ---- global -> registers -> LDS staging of one operand tile ------------------------------------------------------------
// fp32 MFMA shares the vector ALU's datapath on gfx950 (measured: they do not overlap), so every address instruction in
// the k-loop is paid in matrix time.  The loads are therefore raw BUFFER loads: each thread's byte offsets are computed
// ONCE (rows / columns outside the tensor get an offset past the end), a stage adds only a scalar offset, and the
// hardware range check of the buffer descriptor returns zeros for everything outside — no compares, no 64-bit address
// arithmetic, no branches in the loop.  Requires the (batch slice of the) tensor to be < 2 GB.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB_OFF = 0x80000000u;

__device__ __forceinline__ float4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// DUAL: A' = c0*v + c1*y + c2 (BN backward-apply), zeros outside the tensor.
template <int BT, int LAY, bool DUAL, int BK>
struct Stage {
  static constexpr int P = TileGeom<BT, LAY, BK>::PER_THREAD;
  static constexpr int KC = BK / 4;                  // float4 per tile row (LAY_KQ)
  static constexpr int RP = 256 / KC;                // tile rows per pass (LAY_KQ)
  float4 v[P];
  float4 y[DUAL ? P : 1];
  float4 c0, c1, c2;
  unsigned voff[P];                                  // per-thread byte offsets (OOB_OFF: outside)
  unsigned okmask;                                   // DUAL: bit p = element p of the CURRENT stage lies inside the tensor
  __amdgpu_buffer_rsrc_t r1, r2;
  const float* coef; int cstride; int ld, kend, kq;  // kq: this thread's k offset inside a stage (LAY_KQ)

  // g / g2: batch slice base; the tensor is (rows_total x ld) for LAY_KQ, (ktotal x ld) with rows_total columns for LAY_KM
  __device__ __forceinline__ void init(const float* g, const float* g2, int ld_, int t0, int rows_total, int ktotal,
                                       int kend_, const float* coef_, int cstride_) {
    const int tid = threadIdx.x;
    ld = ld_; kend = kend_; coef = coef_; cstride = cstride_;
    const size_t bytes = (size_t)(LAY == LAY_KQ ? rows_total : ktotal) * ld * sizeof(float);
    r1 = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (unsigned)bytes, 0x00020000);
    if (DUAL) r2 = __builtin_amdgcn_make_buffer_rsrc((void*)g2, 0, (unsigned)bytes, 0x00020000);
    kq = 4 * (tid % KC);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (LAY == LAY_KQ) {
        const int row = t0 + tid / KC + RP * p;
        voff[p] = row < rows_total ? (unsigned)(((size_t)row * ld + kq) * 4) : OOB_OFF;
      } else {
        constexpr int C4 = BT / 4;
        const int idx = tid + 256 * p;
        const int col = t0 + 4 * (idx % C4);
        voff[p] = col < rows_total ? (unsigned)(((size_t)(idx / C4) * ld + col) * 4) : OOB_OFF;
      }
    }
    if (DUAL && LAY == LAY_KM) {                     // coefficient channel = tile column: fixed per thread
      int ch = t0 + 4 * (tid % (BT / 4));
      if (ch >= rows_total) ch = 0;
      c0 = *reinterpret_cast<const float4*>(coef + ch);
      c1 = *reinterpret_cast<const float4*>(coef + cstride + ch);
      c2 = *reinterpret_cast<const float4*>(coef + 2 * cstride + ch);
    }
  }
  // Issues the loads of the stage starting at k0 (nothing here depends on a loaded value, so the wave does not wait
  // before the MFMA block that follows); the prologue arithmetic runs in store(), after that block.
  __device__ __forceinline__ void load(int k0) {
    // LAY_KQ: a k-quad past kend (K % BK != 0: last stage only) must read zeros although it lies inside the next row
    const bool kin = LAY == LAY_KQ ? (k0 + kq < kend) : true;
    const unsigned soff = LAY == LAY_KQ ? (unsigned)k0 * 4u : (unsigned)k0 * (unsigned)ld * 4u;
    okmask = 0u;
    if (DUAL && LAY == LAY_KQ) {                     // coefficient channel = k: changes per stage
      const int ch = kin ? k0 + kq : 0;
      c0 = *reinterpret_cast<const float4*>(coef + ch);
      c1 = *reinterpret_cast<const float4*>(coef + cstride + ch);
      c2 = *reinterpret_cast<const float4*>(coef + 2 * cstride + ch);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const unsigned vo = kin ? voff[p] : OOB_OFF;
      v[p] = bload(r1, vo, soff);
      if (DUAL) {
        y[p] = bload(r2, vo, soff);
        // inside <=> the offset is valid and (LAY_KM) the stage row is below kend
        const bool in = vo != OOB_OFF && (LAY == LAY_KQ || k0 + (int)((threadIdx.x + 256 * p) / (BT / 4)) < kend);
        okmask |= in ? (1u << p) : 0u;
      }
    }
  }
  __device__ __forceinline__ void store(float4* __restrict__ lds) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float4 r = v[p];
      if (DUAL) {
        const bool outside = ((okmask >> p) & 1u) == 0u;
        float4 t;
        t.x = __builtin_fmaf(c0.x, r.x, __builtin_fmaf(c1.x, y[p].x, c2.x));
        t.y = __builtin_fmaf(c0.y, r.y, __builtin_fmaf(c1.y, y[p].y, c2.y));
        t.z = __builtin_fmaf(c0.z, r.z, __builtin_fmaf(c1.z, y[p].z, c2.z));
        t.w = __builtin_fmaf(c0.w, r.w, __builtin_fmaf(c1.w, y[p].w, c2.w));
        r = outside ? make_float4(0.f, 0.f, 0.f, 0.f) : t;
      }
      if (LAY == LAY_KQ) lds[(tid % KC) * (BT + 1) + tid / KC + RP * p] = r;
      else lds[tid + 256 * p] = r;                       // [k][BT] floats == [k][BT/4] float4, idx-linear
    }
  }
};

// ---- MFMA shape traits.  MI = 32: v_mfma_f32_32x32x2_f32 (16 accumulator registers, 2 k per instruction: one LDS
//      fragment read feeds a "k-group" of 8); MI = 16: v_mfma_f32_16x16x4_f32 (4 registers, 4 k per instruction, k-group
//      of 16).  The 16x16 form quarters the output granule a wave owns — the projections of this path are small enough
//      (3 240 tiles of 32x32 over 1 024 SIMDs at cfg2) that the granule, not the pipe rate, sets the time.
template <int MI> struct Mfma;
template <> struct Mfma<32> {
  typedef float acc_t __attribute__((ext_vector_type(16)));
  static constexpr int NR = 16, KG = 8, SUBS = 2;         // registers, k per group, lane sub-groups along k
  static __device__ __forceinline__ int sub(int lane) { return lane >> 5; }
  static __device__ __forceinline__ int lrow(int lane) { return lane & 31; }
  static __device__ __forceinline__ int acc_row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ float colsum(float v) { return v + __shfl_xor(v, 32, 64); }
};
template <> struct Mfma<16> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static constexpr int NR = 4, KG = 16, SUBS = 4;
  static __device__ __forceinline__ int sub(int lane) { return lane >> 4; }
  static __device__ __forceinline__ int lrow(int lane) { return lane & 15; }
  static __device__ __forceinline__ int acc_row(int lane, int r) { return 4 * (lane >> 4) + r; }
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ float colsum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
};

// four MFMA operands (j = 0..3) of k-group kg for the MI rows starting at `row0` of the tile: lane sub-group `sub` reads
// k-quad SUBS*kg + sub (KQ) / rows k = KG*kg + 4*sub + j (KM) — the same k <-> (sub, j) map in both images
template <int BT, int LAY, int MI>
__device__ __forceinline__ void frag(const float4* __restrict__ lds, int kg, int row, int sub, float (&f)[4]) {
  if (LAY == LAY_KQ) {
    const float4 v = lds[(Mfma<MI>::SUBS * kg + sub) * (BT + 1) + row];
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
    const float* l = reinterpret_cast<const float*>(lds) + (size_t)(Mfma<MI>::KG * kg + 4 * sub) * BT + row;
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = l[j * BT];
  }
}

// (n, mean, M2) merge (Chan et al.), double precision
__device__ __forceinline__ void chan_merge(double& n, double& mean, double& m2, double nb, double mb, double m2b) {
  if (nb <= 0.0) return;
  const double tot = n + nb;
  const double delta = mb - mean;
  mean += delta * (nb / tot);
  m2 += m2b + delta * delta * (n * nb / tot);
  n = tot;
}

// Write-through (sc1) stores / agent-scope loads for data another workgroup of the SAME launch reads (per-tile statistics,
// split-K partial tiles).  With write-through stores the producer needs no release fence — an agent-scope release
// (buffer_wbl2) writes back every dirty line of the XCD's L2, i.e. the output tiles all co-resident workgroups have just
// stored: measured 13-36 us per projection at cfg2.  Protocol (MI355X guide, "publish/consume recipe" R1): sc1 stores ->
// every storing wave s_waitcnt vmcnt(0) -> workgroup barrier -> ONE lane takes a ticket (agent-scope atomic add); the
// workgroup whose ticket is last does ONE agent-scope acquire (L1 invalidate) -> vmcnt(0) -> barrier -> plain loads.
__device__ __forceinline__ void store_wt(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns true in every thread of the LAST workgroup to arrive at `ctr` (of `expected`); resets the counter for the next
// launch.  `flag`: one LDS word not otherwise in use until the second barrier inside.
__device__ __forceinline__ bool last_arriver(unsigned* ctr, unsigned expected, unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = (t == expected - 1u) ? 1u : 0u;
    if (last) {
      __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last;
  }
  __syncthreads();
  const bool last = *flag != 0u;
  __syncthreads();                                            // the flag word may be reused after this point
  return last;
}

template <int MI, int BM, int BN, int WM, int WN, int ALAY, int BLAY, bool DUAL, int EPI, int BK = 32>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  typedef Mfma<MI> MF;
  typedef typename MF::acc_t acc_t;
  constexpr int TM = BM / (MI * WM), TN = BN / (MI * WN);
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && BM % (MI * WM) == 0 && BN % (MI * WN) == 0, "4 waves");
  static_assert(BK % MF::KG == 0, "stage depth is a multiple of the k-group");
  constexpr int AF4 = TileGeom<BM, ALAY, BK>::F4, BF4 = TileGeom<BN, BLAY, BK>::F4;
  extern __shared__ float4 smem4[];                 // [2][AF4 + BF4]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w / WN, wn = w % WN;
  const int lrow = MF::lrow(lane), sub = MF::sub(lane);
  // XCD-aware workgroup -> tile map.  Workgroups are dealt round-robin over the 8 XCDs, each with its own 4 MB L2.  All
  // tiles that re-read the same operand rows — the column tiles of one row block (forward / dgrad: the A rows), all
  // output tiles of one token split (wgrad: the dz / y / x rows) — get linear ids that are congruent mod 8 and adjacent
  // in dispatch order, so those rows are fetched into ONE L2 once instead of once per tile from the Infinity Cache
  // (measured before the map: wgrad 41 us at 199 MB of re-reads, i.e. bound by the ~5 TB/s fabric, not by the MFMAs).
  // Placement only affects speed.
  const int lin = blockIdx.x;
  const int xcd = lin & 7, seq = lin >> 3;
  constexpr bool SPLIT = EPI == EPI_SPLITK || EPI == EPI_SPLITK_ATOMIC;
  const int inner = SPLIT ? g.mtiles * g.ntiles : g.ntiles;
  const int outer = (seq / inner) * 8 + xcd, in = seq - (seq / inner) * inner;
  const int outer_count = SPLIT ? g.nbatch * g.splits : g.nbatch * g.mtiles;
  if (outer >= outer_count) return;                 // the grid is padded to a multiple of 8 outer units (uniform exit)
  int kbeg = 0, kend = g.K;
  int zq, split = 0, mt, nt;                        // batch (group) index, split index, tile coordinates
  if (SPLIT) {
    zq = outer / g.splits;
    split = outer - zq * g.splits;
    mt = in / g.ntiles; nt = in - mt * g.ntiles;
    kbeg = split * g.k_per_split;
    kend = min(g.K, kbeg + g.k_per_split);
  } else {
    zq = outer / g.mtiles;
    mt = outer - zq * g.mtiles;
    nt = in;
  }
  const int z = zq;
  const int m0 = mt * BM, n0 = nt * BN;
  const float* A = g.A + (size_t)zq * g.a_bstride;
  const float* A2 = DUAL ? g.A2 + (size_t)zq * g.a_bstride : nullptr;
  const float* B = g.B + (size_t)zq * g.b_bstride;
  const float* coef = DUAL ? g.coef + (size_t)zq * g.coef_bstride : nullptr;
  const int nk = (kend - kbeg + BK - 1) / BK;

  Stage<BM, ALAY, DUAL, BK> sa;
  Stage<BN, BLAY, false, BK> sb;
  acc_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < MF::NR; ++r) acc[i][j][r] = 0.f;

  sa.init(A, A2, g.lda, m0, g.M, g.K, kend, coef, g.coef_stride);
  sb.init(B, nullptr, g.ldb, n0, g.N, g.K, kend, nullptr, 0);
  sa.load(kbeg);
  sb.load(kbeg);
  sa.store(smem4);
  sb.store(smem4 + AF4);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const float4* la = smem4 + (kt & 1) * (AF4 + BF4);
    const float4* lb = la + AF4;
    if (kt + 1 < nk) {
      sa.load(kbeg + (kt + 1) * BK);
      sb.load(kbeg + (kt + 1) * BK);
    }
    // k-groups, software-pipelined: the fragments of group kg+1 are read from LDS before the MFMAs of group kg issue
    constexpr int NKG = BK / MF::KG;
    float fa[2][TM][4], fb[2][TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) frag<BM, ALAY, MI>(la, 0, (wm * TM + i) * MI + lrow, sub, fa[0][i]);
#pragma unroll
    for (int j = 0; j < TN; ++j) frag<BN, BLAY, MI>(lb, 0, (wn * TN + j) * MI + lrow, sub, fb[0][j]);
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) {
      if (kg + 1 < NKG) {
#pragma unroll
        for (int i = 0; i < TM; ++i) frag<BM, ALAY, MI>(la, kg + 1, (wm * TM + i) * MI + lrow, sub, fa[(kg + 1) & 1][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) frag<BN, BLAY, MI>(lb, kg + 1, (wn * TN + j) * MI + lrow, sub, fb[(kg + 1) & 1][j]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(fa[kg & 1][i][q], fb[kg & 1][j][q], acc[i][j]);
    }
    if (kt + 1 < nk) {
      float4* na = smem4 + ((kt + 1) & 1) * (AF4 + BF4);
      sa.store(na);
      sb.store(na + AF4);
    }
    __syncthreads();
  }

  // ---- epilogue.  acc[i][j][r]: tile row (wm*TM+i)*MI + acc_row(lane, r), tile column (wn*TN+j)*MI + lrow
  unsigned* flag = reinterpret_cast<unsigned*>(smem4) + 8192;          // an LDS word no epilogue array reaches
  if (EPI == EPI_SPLITK_ATOMIC) {
    // fp32 hardware atomics straight into the (pre-zeroed) output: one register of a 32x32 accumulator is two 128-byte
    // row segments per wave instruction, the shape the memory-side atomic units take at full rate; nothing waits on
    // another workgroup.  The summation order over the splits is run-dependent (last-bit differences, like the vendor
    // library's split-K); the ordered variant below is the deterministic alternative.
    float* Cz = g.C + (size_t)zq * g.c_bstride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * MI + lrow;
#pragma unroll
        for (int r = 0; r < MF::NR; ++r) {
          const int row = m0 + (wm * TM + i) * MI + MF::acc_row(lane, r);
          if (row < g.M && col < g.N)
            __hip_atomic_fetch_add(Cz + (size_t)row * g.ldc + col, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    return;
  }
  if (EPI == EPI_SPLITK) {
    // partial tile -> workspace (write-through); the last-arriving split of this output tile adds the partials in split order
    const int tile = (zq * g.mtiles + mt) * g.ntiles + nt;
    float* mine = g.kpart + ((size_t)tile * g.splits + split) * (BM * BN);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < MF::NR; ++r)
          store_wt(mine + ((wm * TM + i) * MI + MF::acc_row(lane, r)) * BN + (wn * TN + j) * MI + lrow, acc[i][j][r]);
    if (!last_arriver(g.counters + tile, (unsigned)g.splits, flag)) return;
    const float* base = g.kpart + (size_t)tile * g.splits * (BM * BN);
    float* Cz = g.C + (size_t)zq * g.c_bstride;
    for (int e = tid; e < BM * BN / 4; e += 256) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      int sp = 0;
      for (; sp + 4 <= g.splits; sp += 4) {                     // 4 partial tiles in flight, added in split order
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(base + (size_t)(sp + u) * (BM * BN) + 4 * e);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
      }
      for (; sp < g.splits; ++sp) {
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)sp * (BM * BN) + 4 * e);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int row = m0 + (4 * e) / BN, col = n0 + (4 * e) % BN;
      if (row < g.M && col < g.N) *reinterpret_cast<float4*>(Cz + (size_t)row * g.ldc + col) = s;
    }
    return;
  }

  float* C = g.C + (size_t)z * g.c_bstride;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * MI + lrow;
#pragma unroll
      for (int r = 0; r < MF::NR; ++r) {
        const int row = m0 + (wm * TM + i) * MI + MF::acc_row(lane, r);
        if (row < g.M && col < g.N) C[(size_t)row * g.ldc + col] = acc[i][j][r];
      }
    }
  if (EPI != EPI_BNSTATS) return;

  // ---- train-mode BN statistics of this tile's columns: per MI-row group (count, mean, M2), Chan-merged over the
  //      tile's row groups, stored per row block; the last-arriving row block of the column tile merges all of them.
  __syncthreads();                                 // LDS is free again
  float* red = reinterpret_cast<float*>(smem4);     // [WM*TM][BN][2]
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rbase = m0 + (wm * TM + i) * MI;
    const int cnt = max(0, min(MI, g.M - rbase));
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < MF::NR; ++r) s += rbase + MF::acc_row(lane, r) < g.M ? acc[i][j][r] : 0.f;
      s = MF::colsum(s);
      const float mean = cnt > 0 ? s / (float)cnt : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int r = 0; r < MF::NR; ++r) {
        const float d = acc[i][j][r] - mean;
        m2 += rbase + MF::acc_row(lane, r) < g.M ? d * d : 0.f;
      }
      m2 = MF::colsum(m2);
      if (sub == 0) {
        float* o = red + (((wm * TM + i) * BN) + (wn * TN + j) * MI + lrow) * 2;
        o[0] = mean; o[1] = m2;
      }
    }
  }
  __syncthreads();
  // One fp64 atomic pair per column and tile: S += n*mean, Q += M2 + n*mean^2.  The tile statistics are CENTRED (mean, M2
  // from registers) and everything after them is double precision, so Q/R - (S/R)^2 in bn_sums_finalize_kernel carries no
  // fp32 cancellation however large |mean|/std is.  No workgroup waits for another: no fences, no tail.
  if (tid < BN && n0 + tid < g.N) {
    double n = 0.0, mean = 0.0, m2 = 0.0;
#pragma unroll
    for (int rg = 0; rg < WM * TM; ++rg) {
      const int cnt = max(0, min(MI, g.M - (m0 + rg * MI)));
      chan_merge(n, mean, m2, (double)cnt, (double)red[(rg * BN + tid) * 2], (double)red[(rg * BN + tid) * 2 + 1]);
    }
    double* sz = g.sums + (size_t)z * 2 * g.N + n0 + tid;
    __hip_atomic_fetch_add(sz, n * mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(sz + g.N, m2 + n * mean * mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

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

using namespace gkg;

namespace {

template <int MI, int BM, int BN, int WM, int WN, int ALAY, int BLAY, bool DUAL, int EPI, int BK = 32>
hipError_t launch(const GemmArgs& a, int nbatch, hipStream_t st) {
  constexpr size_t stage = (size_t)(TileGeom<BM, ALAY, BK>::F4 + TileGeom<BN, BLAY, BK>::F4) * sizeof(float4);
  size_t lds = 2 * stage;
  const size_t epi = 8193 * sizeof(unsigned);            // epilogue scratch + the last-arriver flag word
  if (lds < epi) lds = epi;
  auto kern = gemm_f32_kernel<MI, BM, BN, WM, WN, ALAY, BLAY, DUAL, EPI, BK>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  GemmArgs b = a;
  b.mtiles = (a.M + BM - 1) / BM; b.ntiles = (a.N + BN - 1) / BN; b.nbatch = nbatch;
  constexpr bool SPLIT = EPI == EPI_SPLITK || EPI == EPI_SPLITK_ATOMIC;
  const long inner = SPLIT ? (long)b.mtiles * b.ntiles : b.ntiles;
  const long outer = SPLIT ? (long)nbatch * a.splits : (long)nbatch * b.mtiles;
  dim3 grid((unsigned)(((outer + 7) / 8) * 8 * inner));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, b);
  return hipGetLastError();
}

inline bool bad_dim(int v) { return v <= 0 || (v & 3) != 0; }

// Tile choice (measured on MI355X, tools/ubench/gemm_bench.hip): the forward runs fastest on 64x64 workgroup tiles
// (one 32x32 MFMA tile per wave) at every shape of this path — 10 368- and 41 472-token stages alike; 128x64 only ties.

}  // namespace

extern "C" int gkg_linear_stats_doubles() { return 2 * 4096 * 4; }

// Forward projection + (train) BN statistics.   x (nb, R, cin) row-major, w (nb, cout, cin), y (nb, R, cout).
// train != 0: writes bn_a / bn_c / bn_mean / bn_invstd [nb][cout] and updates the running statistics (bias folded in);
//             `stats`: gkg_linear_stats_doubles() doubles, zero on entry, zero again on exit.
// train == 0: plain projection (the caller folds eval-mode BN with gkg_bn_eval_affine).
extern "C" int gkg_linear_bn_fwd(const float* x, const float* w, float* y, int R, int cin, int cout, int nb, int train,
                                 const float* gamma, const float* beta, const float* bias, float* running_mean,
                                 float* running_var, long long* num_batches_tracked, float* bn_a, float* bn_c,
                                 float* bn_mean, float* bn_invstd, float momentum, float eps, double* stats, void* stream) {
  if (!x || !w || !y) return gkg_fail(GKG_ERR_NULL, "gkg_linear_bn_fwd: null pointer");
  if (R <= 0 || bad_dim(cin) || bad_dim(cout) || nb <= 0 || nb > 64) return gkg_fail(GKG_ERR_SHAPE, "gkg_linear_bn_fwd: need R > 0, cin % 4 == 0, cout % 4 == 0, 1 <= nb <= 64");
  GemmArgs a{};
  a.A = x; a.a_bstride = (size_t)R * cin; a.lda = cin;
  a.B = w; a.b_bstride = (size_t)cout * cin; a.ldb = cin;
  a.C = y; a.c_bstride = (size_t)R * cout; a.ldc = cout;
  a.M = R; a.N = cout; a.K = cin;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (train == 2) {                                // statistics only: the sums stay in `stats` for gkg_bn_apply_train
    if (!stats) return gkg_fail(GKG_ERR_NULL, "gkg_linear_bn_fwd: train == 2 needs the stats scratch");
    if ((size_t)nb * 2 * cout > (size_t)gkg_linear_stats_doubles()) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_linear_bn_fwd: nb * cout too large for the stats scratch");
    a.sums = stats;
    e = launch<32, 64, 64, 2, 2, LAY_KQ, LAY_KQ, false, EPI_BNSTATS>(a, nb, st);
  } else if (train) {
    if (!gamma || !beta || !bn_a || !bn_c || !bn_mean || !bn_invstd || !stats)
      return gkg_fail(GKG_ERR_NULL, "gkg_linear_bn_fwd: training needs gamma, beta, the four outputs and the stats scratch");
    if ((running_mean == nullptr) != (running_var == nullptr)) return gkg_fail(GKG_ERR_NULL, "gkg_linear_bn_fwd: running stats come in pairs");
    if ((size_t)nb * 2 * cout > (size_t)gkg_linear_stats_doubles()) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_linear_bn_fwd: nb * cout too large for the stats scratch");
    a.sums = stats;
    e = launch<32, 64, 64, 2, 2, LAY_KQ, LAY_KQ, false, EPI_BNSTATS>(a, nb, st);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(bn_sums_finalize_kernel, dim3((cout + 255) / 256, nb), dim3(256), 0, st, stats, R, cout, gamma, beta,
                         bias, running_mean, running_var, bn_a, bn_c, bn_mean, bn_invstd, momentum, eps, num_batches_tracked, 1, nb);
      e = hipGetLastError();
    }
  } else {
    e = launch<32, 64, 64, 2, 2, LAY_KQ, LAY_KQ, false, EPI_STORE>(a, nb, st);
  }
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "gemm_f32_kernel (forward)");
}
