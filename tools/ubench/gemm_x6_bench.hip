// Prototype: fp32 GEMM C[m][n] = sum_k A[m][k] * B[n][k] computed on the bf16 matrix cores with a 3-way operand split
// (x = hi + mid + lo, each bf16; six of the nine cross products kept -> per-product error ~2^-23, the fp32 level).
// Kernel-only timing with HIP events + error against an fp64 host evaluation, next to the fp32-MFMA kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I gkgnet_amd/csrc tools/ubench/gemm_x6_bench.hip
//         gkgnet_amd/csrc/gkg_api.hip -o tools/ubench/gemm_x6_bench
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../../gkgnet_amd/csrc/gkg_gemm.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef gkg::f32x16 acc16;

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// two floats -> three packed bf16 pairs (hi, mid, lo), round-to-nearest at every level; residuals are exact
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16(s0, s1);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_x6_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                       int M, int N, int K, int lda, int ldb, int ldc, int ntiles) {
  constexpr int BK = 32, ROWS = BM + BN, TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int UNITS = ROWS * 4, UPT = UNITS / 256;            // (row, k8) units of 8 floats per thread
  static_assert(UNITS % 256 == 0 && WM * WN == 4, "tile");
  extern __shared__ uint4 lds[];                                 // [2 buffers][3 planes][4 k8][ROWS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
  const int tm = blockIdx.x / ntiles, tn = blockIdx.x % ntiles, m0 = tm * BM, n0 = tn * BN;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)min((size_t)M * lda * 4, (size_t)0x7fffffff), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)min((size_t)N * ldb * 4, (size_t)0x7fffffff), 0x00020000);
  int voff[UPT]; bool isb[UPT];
  #pragma unroll
  for (int u = 0; u < UPT; ++u) {
    const int unit = tid + u * 256, row = unit >> 2, k8 = unit & 3;
    isb[u] = row >= BM;
    const int g = isb[u] ? n0 + row - BM : m0 + row, lim = isb[u] ? N : M, ld = isb[u] ? ldb : lda;
    voff[u] = g < lim ? (g * ld + k8 * 8) * 4 : 0x7ffffff0;     // rows past the end read as zeros
  }
  uint4 raw[UPT][2];
  auto load = [&](int k0) {
    #pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int unit = tid + u * 256, k8 = unit & 3;
      const int off = (k0 + k8 * 8 < K) ? voff[u] + k0 * 4 : 0x7ffffff0;
      if (unit >= BM * 4) {
        raw[u][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, 0));
        raw[u][1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rb, off + 16, 0, 0));
      } else {
        raw[u][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, 0));
        raw[u][1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(ra, off + 16, 0, 0));
      }
    }
  };
  auto store = [&](int buf) {
    #pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int unit = tid + u * 256, row = unit >> 2, k8 = unit & 3;
      uint4 h, m, l;
      const float* f0 = (const float*)&raw[u][0];
      const float* f1 = (const float*)&raw[u][1];
      split2(f0[0], f0[1], h.x, m.x, l.x); split2(f0[2], f0[3], h.y, m.y, l.y);
      split2(f1[0], f1[1], h.z, m.z, l.z); split2(f1[2], f1[3], h.w, m.w, l.w);
      uint4* base = lds + buf * (12 * ROWS) + k8 * ROWS + row;
      base[0] = h; base[4 * ROWS] = m; base[8 * ROWS] = l;
    }
  };

  acc16 acc[MI][NI], accs[MI][NI];
  #pragma unroll
  for (int i = 0; i < MI; ++i)
    #pragma unroll
    for (int j = 0; j < NI; ++j)
      #pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accs[i][j][r] = 0.f; }

  const int nk = (K + BK - 1) / BK;
  load(0); store(0);
  if (nk > 1) load(BK);
  __syncthreads();
  const int r = lane & 31, h = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const uint4* buf = lds + (kt & 1) * (12 * ROWS);
    #pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[MI][3], b[NI][3];
      #pragma unroll
      for (int p = 0; p < 3; ++p) {
        #pragma unroll
        for (int i = 0; i < MI; ++i) a[i][p] = __builtin_bit_cast(bf16x8, buf[(p * 4 + 2 * s + h) * ROWS + wm * TM + i * 32 + r]);
        #pragma unroll
        for (int j = 0; j < NI; ++j) b[j][p] = __builtin_bit_cast(bf16x8, buf[(p * 4 + 2 * s + h) * ROWS + BM + wn * TN + j * 32 + r]);
      }
      #pragma unroll
      for (int i = 0; i < MI; ++i)
        #pragma unroll
        for (int j = 0; j < NI; ++j) {
          accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], accs[i][j], 0, 0, 0);
          accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], accs[i][j], 0, 0, 0);
          accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], accs[i][j], 0, 0, 0);
          accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], accs[i][j], 0, 0, 0);
          accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], accs[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
      if (s == 0 && kt + 1 < nk) store((kt + 1) & 1);
    }
    if (kt + 2 < nk) load((kt + 2) * BK);
    __syncthreads();
  }
  #pragma unroll
  for (int i = 0; i < MI; ++i)
    #pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * TN + j * 32 + r;
      #pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = m0 + wm * TM + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (m < M && n < N) C[(size_t)m * ldc + n] = acc[i][j][q] + accs[i][j][q];
      }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// v2: LDS-DMA pipeline.  4 waves as 4(M) x 1(N): wave w owns rows [32w, 32w+32) of the 128-row tile, so its A rows go
// global -> its PRIVATE LDS ring (no workgroup barrier on that path) -> registers, where the wave splits them to bf16
// hi/mid/lo between MFMAs.  The weights arrive pre-split (prep kernel below) as [plane][k/8][n][8 bf16], so a wave
// DMA instruction copies 64 n x 16 B = 1 KiB contiguous and the LDS image is conflict-free for the fragment read.
__global__ void prep_w_kernel(const float* __restrict__ W, uint4* __restrict__ P, int N, int K, int Npad, int KC, int ldw, int transposed) {
  const int n = blockIdx.x * 256 + threadIdx.x, kc = blockIdx.y;
  if (n >= Npad) return;
  float v[8];
  #pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kc * 8 + j;
    v[j] = (n < N && k < K) ? (transposed ? W[(size_t)k * ldw + n] : W[(size_t)n * ldw + k]) : 0.f;
  }
  uint4 h, m, l;
  split2(v[0], v[1], h.x, m.x, l.x); split2(v[2], v[3], h.y, m.y, l.y); split2(v[4], v[5], h.z, m.z, l.z); split2(v[6], v[7], h.w, m.w, l.w);
  const size_t plane = (size_t)KC * Npad;
  P[(size_t)kc * Npad + n] = h; P[plane + (size_t)kc * Npad + n] = m; P[2 * plane + (size_t)kc * Npad + n] = l;
}

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int NI, int SA, int MODE = 0>
__global__ __launch_bounds__(256) void gemm_x6v2_kernel(const float* __restrict__ A, const uint4* __restrict__ P, float* __restrict__ C,
                                                         int M, int N, int K, int lda, int ldc, int Npad, int KC, int mtiles, int ntiles) {
  constexpr int BM = 128, BN = 32 * NI, BK = 32;
  constexpr int A_STAGE = 4 * 4096, B_STAGE = 12 * BN * 16, B_BASE = SA * A_STAGE;
  constexpr int BI = 12 * BN / 64 / 4;                           // B DMA instructions per wave per K-step
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware map: the column tiles of one row block run back to back on one XCD (A row block fetched into that L2 once)
  const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
  const int tn = slot % ntiles, tm = (slot / ntiles) * 8 + xcd;
  if (tm >= mtiles) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const unsigned lds0 = (unsigned)(size_t)lds;                   // LDS byte address of the dynamic segment

  unsigned aoff[4];                                              // per-lane byte offsets of the 4 A DMA pieces (k0 = 0)
  #pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int L = 64 * i + lane, row = L >> 3, sl = L & 7, chunk = sl ^ ((row >> 1) & 7);
    const int g = min(m0 + 32 * w + row, M - 1);
    aoff[i] = (unsigned)(g * lda + chunk * 4) * 4u;
  }
  const size_t plane = (size_t)KC * Npad * 16;
  unsigned boff[BI]; unsigned bdst[BI];
  #pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int piece = w + 4 * i;                                 // 1-KiB piece of the 12*BN*16-B stage image
    const int rowid = piece * 64 / BN, p = rowid >> 2, chunk = rowid & 3, nb = (piece * 64) % BN;
    boff[i] = (unsigned)(p * plane + ((size_t)chunk * Npad + n0 + nb + lane) * 16);
    bdst[i] = piece * 1024;
  }
  auto dma_a = [&](int kt) {
    const char* base = (const char*)A + (size_t)kt * (BK * 4);
    const unsigned dst = lds0 + (kt % SA) * A_STAGE + w * 4096;
    #pragma unroll
    for (int i = 0; i < 4; ++i) glds16(base, aoff[i], dst + i * 1024);
  };
  auto dma_b = [&](int kt) {
    const char* base = (const char*)P + (size_t)kt * 4 * Npad * 16;
    const unsigned dst = lds0 + B_BASE + (kt & 1) * B_STAGE;
    #pragma unroll
    for (int i = 0; i < BI; ++i) glds16(base, boff[i], dst + bdst[i]);
  };

  acc16 acc[NI], accs[NI];
  #pragma unroll
  for (int j = 0; j < NI; ++j)
    #pragma unroll
    for (int q = 0; q < 16; ++q) { acc[j][q] = 0.f; accs[j][q] = 0.f; }

  const int nk = (K + BK - 1) / BK;
  const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
  dma_a(0); dma_b(0);
  if (nk > 1) dma_a(1);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (MODE != 2) {
      if (kt + 1 < nk) dma_b(kt + 1);
      if (kt + 2 < nk) dma_a(kt + 2);
    }
    if (MODE == 1) continue;
    const char* abase = (const char*)lds + (kt % SA) * A_STAGE + w * 4096 + r * 128;
    const uint4* bbase = lds + (B_BASE + (kt & 1) * B_STAGE) / 16 + r;
    #pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c0 = 4 * s + 2 * h;
      const float4 f0 = *(const float4*)(abase + (((c0) ^ sw) << 4));
      const float4 f1 = *(const float4*)(abase + (((c0 + 1) ^ sw) << 4));
      uint4 ah, am, al;
      split2(f0.x, f0.y, ah.x, am.x, al.x); split2(f0.z, f0.w, ah.y, am.y, al.y);
      split2(f1.x, f1.y, ah.z, am.z, al.z); split2(f1.z, f1.w, ah.w, am.w, al.w);
      const bf16x8 a0 = __builtin_bit_cast(bf16x8, ah), a1 = __builtin_bit_cast(bf16x8, am), a2 = __builtin_bit_cast(bf16x8, al);
      #pragma unroll
      for (int j = 0; j < NI; ++j) {
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, bbase[(0 * 4 + 2 * s + h) * BN + j * 32]);
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, bbase[(1 * 4 + 2 * s + h) * BN + j * 32]);
        const bf16x8 b2 = __builtin_bit_cast(bf16x8, bbase[(2 * 4 + 2 * s + h) * BN + j * 32]);
        accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, accs[j], 0, 0, 0);
        accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, accs[j], 0, 0, 0);
        accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, accs[j], 0, 0, 0);
        accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, accs[j], 0, 0, 0);
        accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, accs[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
      }
    }
  }
  #pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + j * 32 + r;
    #pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + 32 * w + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (m < M && n < N) C[(size_t)m * ldc + n] = acc[j][q] + accs[j][q];
    }
  }
}

template <int NI, int SA, int MODE = 0>
static void run_x6v2(const float* A, const uint4* P, float* C, int M, int N, int K, int Npad, int KC) {
  constexpr int BN = 32 * NI;
  const int mt = (M + 127) / 128, nt = (N + BN - 1) / BN;
  const size_t sh = SA * 4 * 4096 + 2 * 12 * BN * 16;
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_x6v2_kernel<NI, SA, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); once = true; }
  const int groups = (mt + 7) / 8;
  gemm_x6v2_kernel<NI, SA, MODE><<<groups * 8 * nt, 256, sh, 0>>>(A, P, C, M, N, K, K, N, Npad, KC, mt, nt);
}


// ---------------------------------------------------------------------------------------------------------------
// v3: v2's data movement with a software-pipelined inner loop.  Each half K-step (16 deep) first issues the LDS reads
// of the NEXT half step's operands, then runs its MFMAs with the bf16 split of the next A fragment cut into eight
// slices placed one per MFMA gap.  One workgroup barrier per K-step, in the middle of the step: it certifies stage
// kt+1 (whose first fragments are read right after it) and frees stage kt's B image and the wave's A slot.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt2(float a, float b) {
  f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

template <int NI, int SA, int MODE = 0>
__global__ __launch_bounds__(256) void gemm_x6v3_kernel(const float* __restrict__ A, const uint4* __restrict__ P, float* __restrict__ C,
                                                         int M, int N, int K, int lda, int ldc, int Npad, int KC, int mtiles, int ntiles) {
  constexpr int BM = 128, BN = 32 * NI, BK = 32;
  constexpr int A_STAGE = 4 * 4096, B_STAGE = 12 * BN * 16, B_BASE = SA * A_STAGE;
  constexpr int BI = 12 * BN / 64 / 4;
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
  const int tn = slot % ntiles, tm = (slot / ntiles) * 8 + xcd;
  if (tm >= mtiles) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const unsigned lds0 = (unsigned)(size_t)lds;

  unsigned aoff[4];
  #pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int L = 64 * i + lane, row = L >> 3, sl = L & 7, chunk = sl ^ ((row >> 1) & 7);
    const int g = min(m0 + 32 * w + row, M - 1);
    aoff[i] = (unsigned)(g * lda + chunk * 4) * 4u;
  }
  const size_t plane = (size_t)KC * Npad * 16;
  unsigned boff[BI]; unsigned bdst[BI];
  #pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int piece = w + 4 * i;
    const int rowid = piece * 64 / BN, p = rowid >> 2, chunk = rowid & 3, nb = (piece * 64) % BN;
    boff[i] = (unsigned)(p * plane + ((size_t)chunk * Npad + n0 + nb + lane) * 16);
    bdst[i] = piece * 1024;
  }
  auto dma_a = [&](int kt) {
    const char* base = (const char*)A + (size_t)kt * (BK * 4);
    const unsigned dst = lds0 + (kt % SA) * A_STAGE + w * 4096;
    #pragma unroll
    for (int i = 0; i < 4; ++i) glds16(base, aoff[i], dst + i * 1024);
  };
  auto dma_b = [&](int kt) {
    const char* base = (const char*)P + (size_t)kt * 4 * Npad * 16;
    const unsigned dst = lds0 + B_BASE + (kt & 1) * B_STAGE;
    #pragma unroll
    for (int i = 0; i < BI; ++i) glds16(base, boff[i], dst + bdst[i]);
  };

  acc16 acc[NI], accs[NI];
  #pragma unroll
  for (int j = 0; j < NI; ++j)
    #pragma unroll
    for (int q = 0; q < 16; ++q) { acc[j][q] = 0.f; accs[j][q] = 0.f; }

  const int nk = (K + BK - 1) / BK;
  const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
  // per-lane byte offsets of the two 16-B chunks of the raw A fragment, for s = 0 / 1
  int ac[2][2];
  #pragma unroll
  for (int s = 0; s < 2; ++s) { ac[s][0] = r * 128 + (((4 * s + 2 * h) ^ sw) << 4); ac[s][1] = r * 128 + (((4 * s + 2 * h + 1) ^ sw) << 4); }

  float f[8];                       // raw A fragment of the next half step (split in place)
  unsigned sh_[4], sm_[4], sl_[4];  // its hi / mid / lo pairs as they are produced
  bf16x8 a[2][3], b[2][NI][3];

  auto read_raw = [&](int kt, int s) {
    const char* ab = (const char*)lds + (kt % SA) * A_STAGE + w * 4096;
    const float4 f0 = *(const float4*)(ab + ac[s][0]);
    const float4 f1 = *(const float4*)(ab + ac[s][1]);
    f[0] = f0.x; f[1] = f0.y; f[2] = f0.z; f[3] = f0.w; f[4] = f1.x; f[5] = f1.y; f[6] = f1.z; f[7] = f1.w;
  };
  auto read_b = [&](int kt, int s, int buf) {
    const uint4* bb = lds + (B_BASE + (kt & 1) * B_STAGE) / 16 + r;
    #pragma unroll
    for (int j = 0; j < NI; ++j)
      #pragma unroll
      for (int p = 0; p < 3; ++p) b[buf][j][p] = __builtin_bit_cast(bf16x8, bb[(p * 4 + 2 * s + h) * BN + j * 32]);
  };
  // the split of the next raw fragment as 44 single-instruction steps (asm volatile: fixed order, never sunk into
  // another block); step i works on pair i % 4 so that consecutive steps are independent
  unsigned t0_[4], t1_[4];
  auto op = [&](int i) {
    const int p = i & 3, o = i >> 2;
    float& x0 = f[2 * p]; float& x1 = f[2 * p + 1];
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
    a[buf][0] = __builtin_bit_cast(bf16x8, uint4{sh_[0], sh_[1], sh_[2], sh_[3]});
    a[buf][1] = __builtin_bit_cast(bf16x8, uint4{sm_[0], sm_[1], sm_[2], sm_[3]});
    a[buf][2] = __builtin_bit_cast(bf16x8, uint4{sl_[0], sl_[1], sl_[2], sl_[3]});
  };
  // one half step: MFMAs on buffer `cur`; in the gaps the next half step's fragments are read and split
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
        if (!(MODE & 1)) {
          if (t == 0) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][2], b[cur][j][0], accs[j], 0, 0, 0);
          if (t == 1) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], b[cur][j][2], accs[j], 0, 0, 0);
          if (t == 2) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1], b[cur][j][1], accs[j], 0, 0, 0);
          if (t == 3) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1], b[cur][j][0], accs[j], 0, 0, 0);
          if (t == 4) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], b[cur][j][1], accs[j], 0, 0, 0);
          if (t == 5) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], b[cur][j][0], acc[j], 0, 0, 0);
        }
        if (has_next) {
          if (si < 3 * NI && !(MODE & 8)) {                       // one B fragment read per gap, first half of the gaps
            const int jj = si / 3, pp = si % 3;
            b[cur ^ 1][jj][pp] = __builtin_bit_cast(bf16x8, bb[(pp * 4 + 2 * ns + h) * BN + jj * 32]);
          }
          if (si >= 1 && !(MODE & 4)) {
            #pragma unroll
            for (int o = 0; o < OPS_PER; ++o) { const int i = (si - 1) * OPS_PER + o; if (i < 44) op(i); }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (has_next) commit(cur ^ 1);
  };

  dma_a(0); dma_b(0);
  if (nk > 1) { dma_a(1); dma_b(1); }
  if (nk > 2) dma_a(2);
  if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + BI) : "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 + BI) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_raw(0, 0); read_b(0, 0, 0);
  #pragma unroll
  for (int i = 0; i < 44; ++i) op(i);
  commit(0);
  for (int kt = 0; kt + 1 < nk; ++kt) {
    half(0, true, kt, 1);                          // s = 0 on buffer 0; fetches and splits (kt, s = 1) into buffer 1
    // ---- middle of the step: stage kt+1 certified, stage kt released
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!(MODE & 2)) {
      if (kt + 2 < nk) dma_b(kt + 2);
      if (kt + 3 < nk) dma_a(kt + 3);
    }
    half(1, true, kt + 1, 0);                      // s = 1 on buffer 1; fetches and splits (kt+1, s = 0) into buffer 0
  }
  half(0, true, nk - 1, 1);
  half(1, false, 0, 0);
  #pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + j * 32 + r;
    #pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + 32 * w + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (m < M && n < N) C[(size_t)m * ldc + n] = acc[j][q] + accs[j][q];
    }
  }
}

template <int NI, int SA, int MODE = 0>
static void run_x6v3(const float* A, const uint4* P, float* C, int M, int N, int K, int Npad, int KC) {
  constexpr int BN = 32 * NI;
  const int mt = (M + 127) / 128, nt = (N + BN - 1) / BN;
  const size_t sh = SA * 4 * 4096 + 2 * 12 * BN * 16;
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_x6v3_kernel<NI, SA, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); once = true; }
  const int groups = (mt + 7) / 8;
  gemm_x6v3_kernel<NI, SA, MODE><<<groups * 8 * nt, 256, sh, 0>>>(A, P, C, M, N, K, K, N, Npad, KC, mt, nt);
}

template <int NI, int SA, int MODE = 0>
__global__ __launch_bounds__(256) void gemm_x6v4_kernel(const float* __restrict__ A, const uint4* __restrict__ P, float* __restrict__ C,
                                                         int M, int N, int K, int lda, int ldc, int Npad, int KC, int mtiles, int ntiles) {
  constexpr int BM = 128, BN = 32 * NI, BK = 32;
  constexpr int A_STAGE = 4 * 4096, B_STAGE = 12 * BN * 16, B_BASE = SA * A_STAGE;
  constexpr int BI = 12 * BN / 64 / 4;
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
  const int tn = slot % ntiles, tm = (slot / ntiles) * 8 + xcd;
  if (tm >= mtiles) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const unsigned lds0 = (unsigned)(size_t)lds;

  unsigned aoff[4];
  #pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int L = 64 * i + lane, row = L >> 3, sl = L & 7, chunk = sl ^ ((row >> 1) & 7);
    const int g = min(m0 + 32 * w + row, M - 1);
    aoff[i] = (unsigned)(g * lda + chunk * 4) * 4u;
  }
  const size_t plane = (size_t)KC * Npad * 16;
  unsigned boff[BI]; unsigned bdst[BI];
  #pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int piece = w + 4 * i;
    const int rowid = piece * 64 / BN, p = rowid >> 2, chunk = rowid & 3, nb = (piece * 64) % BN;
    boff[i] = (unsigned)(p * plane + ((size_t)chunk * Npad + n0 + nb + lane) * 16);
    bdst[i] = piece * 1024;
  }
  auto dma_a = [&](int kt) {
    const char* base = (const char*)A + (size_t)kt * (BK * 4);
    const unsigned dst = lds0 + (kt % SA) * A_STAGE + w * 4096;
    #pragma unroll
    for (int i = 0; i < 4; ++i) glds16(base, aoff[i], dst + i * 1024);
  };
  auto dma_b = [&](int kt) {
    const char* base = (const char*)P + (size_t)kt * 4 * Npad * 16;
    const unsigned dst = lds0 + B_BASE + (kt & 1) * B_STAGE;
    #pragma unroll
    for (int i = 0; i < BI; ++i) glds16(base, boff[i], dst + bdst[i]);
  };

  auto dma_a1 = [&](int kt, int i) {
    const char* base = (const char*)A + (size_t)kt * (BK * 4);
    const unsigned dst = lds0 + (kt % SA) * A_STAGE + w * 4096;
    #pragma unroll
    for (int q = 0; q < 4; ++q) if (q == i) glds16(base, aoff[q], dst + q * 1024);
  };
  auto dma_b1 = [&](int kt, int i) {
    const char* base = (const char*)P + (size_t)kt * 4 * Npad * 16;
    const unsigned dst = lds0 + B_BASE + (kt & 1) * B_STAGE;
    #pragma unroll
    for (int q = 0; q < BI; ++q) if (q == i) glds16(base, boff[q], dst + bdst[q]);
  };
  acc16 acc[NI], accs[NI];
  #pragma unroll
  for (int j = 0; j < NI; ++j)
    #pragma unroll
    for (int q = 0; q < 16; ++q) { acc[j][q] = 0.f; accs[j][q] = 0.f; }

  const int nk = (K + BK - 1) / BK;
  const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
  // per-lane byte offsets of the two 16-B chunks of the raw A fragment, for s = 0 / 1
  int ac[2][2];
  #pragma unroll
  for (int s = 0; s < 2; ++s) { ac[s][0] = r * 128 + (((4 * s + 2 * h) ^ sw) << 4); ac[s][1] = r * 128 + (((4 * s + 2 * h + 1) ^ sw) << 4); }

  float F[2][8];                    // raw A fragments by half-step parity (split in place, one half step after the read)
  unsigned sh_[4], sm_[4], sl_[4];  // its hi / mid / lo pairs as they are produced
  bf16x8 a[2][3], b[2][NI][3];

  auto read_raw = [&](int kt, int s) {
    const char* ab = (const char*)lds + (kt % SA) * A_STAGE + w * 4096;
    const float4 f0 = *(const float4*)(ab + ac[s][0]);
    const float4 f1 = *(const float4*)(ab + ac[s][1]);
    float* f = F[s];
    f[0] = f0.x; f[1] = f0.y; f[2] = f0.z; f[3] = f0.w; f[4] = f1.x; f[5] = f1.y; f[6] = f1.z; f[7] = f1.w;
  };
  auto read_b = [&](int kt, int s, int buf) {
    const uint4* bb = lds + (B_BASE + (kt & 1) * B_STAGE) / 16 + r;
    #pragma unroll
    for (int j = 0; j < NI; ++j)
      #pragma unroll
      for (int p = 0; p < 3; ++p) b[buf][j][p] = __builtin_bit_cast(bf16x8, bb[(p * 4 + 2 * s + h) * BN + j * 32]);
  };
  // the split of the next raw fragment as 44 single-instruction steps (asm volatile: fixed order, never sunk into
  // another block); step i works on pair i % 4 so that consecutive steps are independent
  unsigned t0_[4], t1_[4];
  auto op = [&](int fs, int i) {
    const int p = i & 3, o = i >> 2;
    float& x0 = F[fs][2 * p]; float& x1 = F[fs][2 * p + 1];
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
    a[buf][0] = __builtin_bit_cast(bf16x8, uint4{sh_[0], sh_[1], sh_[2], sh_[3]});
    a[buf][1] = __builtin_bit_cast(bf16x8, uint4{sm_[0], sm_[1], sm_[2], sm_[3]});
    a[buf][2] = __builtin_bit_cast(bf16x8, uint4{sl_[0], sl_[1], sl_[2], sl_[3]});
  };
  // one half step: MFMAs on buffer `cur`; in the gaps the next half step's fragments are read and split
  // half step `cur` (= s): MFMAs on buffer cur; in the gaps: B fragments of the next half step (nkt, cur^1), the split
  // of its raw A fragment F[cur^1] (read one half step ago), the raw read two half steps ahead (rkt, cur) into F[cur],
  // and (dma_kt >= 0) this wave's share of the DMA for stages dma_kt+2 (B) / dma_kt+3 (A), one instruction per gap
  constexpr int SLOTS = 6 * NI, FIRST = 2, OPS_PER = (44 + SLOTS - FIRST - 1) / (SLOTS - FIRST);
  auto half = [&](int cur, bool has_next, int nkt, bool has_raw, int rkt, int dma_kt) {
    const uint4* bb = lds + (B_BASE + (nkt & 1) * B_STAGE) / 16 + r;
    #pragma unroll
    for (int j = 0; j < NI; ++j) {
      #pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int si = j * 6 + t;
        if (!(MODE & 1)) {
          if (t == 0) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][2], b[cur][j][0], accs[j], 0, 0, 0);
          if (t == 1) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], b[cur][j][2], accs[j], 0, 0, 0);
          if (t == 2) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1], b[cur][j][1], accs[j], 0, 0, 0);
          if (t == 3) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1], b[cur][j][0], accs[j], 0, 0, 0);
          if (t == 4) accs[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], b[cur][j][1], accs[j], 0, 0, 0);
          if (t == 5) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0], b[cur][j][0], acc[j], 0, 0, 0);
        }
        if (has_next) {
          if (si < 3 * NI && !(MODE & 8)) {
            const int jj = si / 3, pp = si % 3;
            b[cur ^ 1][jj][pp] = __builtin_bit_cast(bf16x8, bb[(pp * 4 + 2 * (cur ^ 1) + h) * BN + jj * 32]);
          }
          if (si >= FIRST && !(MODE & 4)) {
            #pragma unroll
            for (int o = 0; o < OPS_PER; ++o) { const int i = (si - FIRST) * OPS_PER + o; if (i < 44) op(cur ^ 1, i); }
          }
        }
        if (dma_kt >= 0 && !(MODE & 2)) {
          if (si < BI) { if (dma_kt + 2 < nk) dma_b1(dma_kt + 2, si); }
          else if (si < BI + 4) { if (dma_kt + 3 < nk) dma_a1(dma_kt + 3, si - BI); }
        }
        if (si == SLOTS - 1 && has_raw) read_raw(rkt, cur);   // F[cur]'s previous content was consumed a half step ago
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (has_next) commit(cur ^ 1);
  };

  dma_a(0); dma_b(0);
  if (nk > 1) { dma_a(1); dma_b(1); }
  if (nk > 2) dma_a(2);
  if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + BI) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_raw(0, 0); read_b(0, 0, 0);
  #pragma unroll
  for (int i = 0; i < 44; ++i) op(0, i);
  commit(0);
  read_raw(0, 1);
  for (int kt = 0; kt + 1 < nk; ++kt) {
    // the wave's own A stage kt+1 must have landed before its first raw read at the end of this half step
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(BI + 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    half(0, true, kt, true, kt + 1, -1);
    // ---- middle of the step: stage kt+1's B image certified, stage kt released
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    half(1, true, kt + 1, true, kt + 1, kt);
  }
  half(0, true, nk - 1, false, 0, -1);
  half(1, false, 0, false, 0, -1);
  #pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + j * 32 + r;
    #pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + 32 * w + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (m < M && n < N) C[(size_t)m * ldc + n] = acc[j][q] + accs[j][q];
    }
  }
}

template <int NI, int SA, int MODE = 0>
static void run_x6v4(const float* A, const uint4* P, float* C, int M, int N, int K, int Npad, int KC) {
  constexpr int BN = 32 * NI;
  const int mt = (M + 127) / 128, nt = (N + BN - 1) / BN;
  const size_t sh = SA * 4 * 4096 + 2 * 12 * BN * 16;
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_x6v4_kernel<NI, SA, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); once = true; }
  const int groups = (mt + 7) / 8;
  gemm_x6v4_kernel<NI, SA, MODE><<<groups * 8 * nt, 256, sh, 0>>>(A, P, C, M, N, K, K, N, Npad, KC, mt, nt);
}

__global__ void fill(float* p, size_t n, unsigned seed, float scale) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
               p[i] = scale * ((h & 0xffffff) / 8388608.0f - 1.0f); }
}

template <typename F>
static float timeit(F f, int n = 40) {
  for (int i = 0; i < 5; ++i) f();
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < n; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return 1e3f * ms / n;
}

template <int BM, int BN, int WM, int WN>
static void run_x6(const float* A, const float* B, float* C, int M, int N, int K) {
  const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
  const size_t sh = 2 * 12 * (BM + BN) * 16;
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_x6_kernel<BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); once = true; }
  gemm_x6_kernel<BM, BN, WM, WN><<<mt * nt, 256, sh, 0>>>(A, B, C, M, N, K, K, K, N, nt);
}

static void errors(const char* name, const std::vector<float>& a, const std::vector<float>& w, const float* y_dev, int R, int K, int N) {
  std::vector<float> y((size_t)R * N);
  CK(hipMemcpy(y.data(), y_dev, y.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, sum = 0; int cnt = 0;
  for (int t = 0; t < 4000; ++t) {
    const int m = (int)((t * 2654435761u) % (unsigned)R), n = (int)((t * 40503u + 7) % (unsigned)N);
    double ref = 0, mag = 0;
    for (int k = 0; k < K; ++k) { const double p = (double)a[(size_t)m * K + k] * (double)w[(size_t)n * K + k]; ref += p; mag += fabs(p); }
    const double e = fabs((double)y[(size_t)m * N + n] - ref) / mag;      // error relative to sum |a b| (the fp32 bound's scale)
    worst = e > worst ? e : worst; sum += e; ++cnt;
  }
  printf("   %-34s err/sum|ab|: max %.3e  mean %.3e\n", name, worst, sum / cnt);
}

int main() {
  struct Shape { int R, cin, cout; };
  std::vector<Shape> shapes = {{10368, 320, 320}, {10368, 640, 320}, {2560, 320, 320}, {2560, 320, 1280}, {2560, 1280, 320}, {41472, 400, 400}, {41472, 800, 400}};
  for (auto s : shapes) {
    const size_t na = (size_t)s.R * s.cin, nw = (size_t)s.cout * s.cin, nc = (size_t)s.R * s.cout;
    float *x, *w, *y; double* part; unsigned* ctr;
    CK(hipMalloc(&x, na * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&y, nc * 4));
    CK(hipMalloc(&part, 2 * s.cout * 8)); CK(hipMemset(part, 0, 2 * s.cout * 8)); CK(hipMalloc(&ctr, 4096 * 4)); CK(hipMemset(ctr, 0, 4096 * 4));
    fill<<<(na + 255) / 256, 256>>>(x, na, 1, 2.0f); fill<<<(nw + 255) / 256, 256>>>(w, nw, 2, 0.1f);
    std::vector<float> hx(na), hw(nw);
    CK(hipMemcpy(hx.data(), x, na * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hw.data(), w, nw * 4, hipMemcpyDeviceToHost));
    GemmArgs a{};
    a.A = x; a.a_bstride = na; a.lda = s.cin; a.B = w; a.b_bstride = nw; a.ldb = s.cin; a.C = y; a.c_bstride = nc; a.ldc = s.cout;
    a.M = s.R; a.N = s.cout; a.K = s.cin; a.sums = part; a.counters = ctr;
    const double fl = 2.0 * s.R * s.cin * s.cout;
    printf("R=%d cin=%d cout=%d:\n", s.R, s.cin, s.cout);
    auto rep = [&](const char* name, float us) { printf("   %-34s %8.1f us  %6.1f TF\n", name, us, fl / us / 1e6); };
    rep("f32 mfma 64x64 2x2", timeit([&] { launch<32, 64, 64, 2, 2, LAY_KQ, LAY_KQ, false, EPI_STORE, 32>(a, 1, 0); }));
    errors("f32 mfma", hx, hw, y, s.R, s.cin, s.cout);
    CK(hipMemset(y, 0, nc * 4));
    rep("x6 128x64 2x2", timeit([&] { run_x6<128, 64, 2, 2>(x, w, y, s.R, s.cout, s.cin); }));
    errors("x6 128x64", hx, hw, y, s.R, s.cin, s.cout);
    CK(hipMemset(y, 0, nc * 4));
    rep("x6 64x64 2x2", timeit([&] { run_x6<64, 64, 2, 2>(x, w, y, s.R, s.cout, s.cin); }));
    errors("x6 64x64", hx, hw, y, s.R, s.cin, s.cout);
    {
      const int Npad = (s.cout + 127) / 128 * 128, KC = (s.cin + 31) / 32 * 4;
      uint4* P; CK(hipMalloc(&P, (size_t)3 * KC * Npad * 16));
      rep("prep W planes", timeit([&] { prep_w_kernel<<<dim3((Npad + 255) / 256, KC), 256>>>(w, P, s.cout, s.cin, Npad, KC, s.cin, 0); }));
      CK(hipMemset(y, 0, nc * 4));
      rep("x6v2 128x64 SA3", timeit([&] { run_x6v2<2, 3>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      errors("x6v2 128x64", hx, hw, y, s.R, s.cin, s.cout);
      CK(hipMemset(y, 0, nc * 4));
      rep("x6v2 128x128 SA3", timeit([&] { run_x6v2<4, 3>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      errors("x6v2 128x128", hx, hw, y, s.R, s.cin, s.cout);
      CK(hipMemset(y, 0, nc * 4));
      rep("x6v3 128x64 SA3", timeit([&] { run_x6v3<2, 3>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      errors("x6v3 128x64", hx, hw, y, s.R, s.cin, s.cout);
      CK(hipMemset(y, 0, nc * 4));
      rep("x6v3 128x128 SA3", timeit([&] { run_x6v3<4, 3>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      errors("x6v3 128x128", hx, hw, y, s.R, s.cin, s.cout);
      CK(hipMemset(y, 0, nc * 4));
      rep("x6v4 128x64 SA3", timeit([&] { run_x6v4<2, 3>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      errors("x6v4 128x64", hx, hw, y, s.R, s.cin, s.cout);
      CK(hipMemset(y, 0, nc * 4));
      rep("x6v4 128x128 SA3", timeit([&] { run_x6v4<4, 3>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      errors("x6v4 128x128", hx, hw, y, s.R, s.cin, s.cout);
      rep("x6v4 128x128 no dma (2)", timeit([&] { run_x6v4<4, 3, 2>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v3 128x128 no dma (2)", timeit([&] { run_x6v3<4, 3, 2>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v3 128x128 no mfma (1)", timeit([&] { run_x6v3<4, 3, 1>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v3 128x128 no mfma,dma (3)", timeit([&] { run_x6v3<4, 3, 3>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v3 128x128 no mfma,dma,split (7)", timeit([&] { run_x6v3<4, 3, 7>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v3 128x128 no mfma,dma,split,bread (15)", timeit([&] { run_x6v3<4, 3, 15>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v3 128x128 no dma,split (6)", timeit([&] { run_x6v3<4, 3, 6>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v3 128x128 no dma,split,bread (14)", timeit([&] { run_x6v3<4, 3, 14>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v3 128x128 no dma,bread (10)", timeit([&] { run_x6v3<4, 3, 10>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v2 128x64 SA3 DMA only", timeit([&] { run_x6v2<2, 3, 1>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v2 128x64 SA3 compute only", timeit([&] { run_x6v2<2, 3, 2>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v2 128x128 SA3 DMA only", timeit([&] { run_x6v2<4, 3, 1>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v2 128x128 SA3 compute only", timeit([&] { run_x6v2<4, 3, 2>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      rep("x6v2 128x64 SA4", timeit([&] { run_x6v2<2, 4>(x, P, y, s.R, s.cout, s.cin, Npad, KC); }));
      CK(hipFree(P));
    }
    rep("x6 128x128 2x2", timeit([&] { run_x6<128, 128, 2, 2>(x, w, y, s.R, s.cout, s.cin); }));
    rep("x6 64x128 2x2", timeit([&] { run_x6<64, 128, 2, 2>(x, w, y, s.R, s.cout, s.cin); }));
    CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(part)); CK(hipFree(ctr));
  }
  return 0;
}
