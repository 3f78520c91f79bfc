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
#include "gkg_common.h"

namespace gkg {

constexpr int MR_LDS_BUDGET = 96 * 1024;   // bytes of LDS for source rows / accumulators

// ------------------------------------------------------------------------------------------ forward
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

  const T* sp = src + ((size_t)bg * c + ch0) * M;
  for (int i = tid; i < nch * M; i += 256) rows[i] = ldf(sp + i);
  {
    // indices of this tile: (256, k) int64 contiguous -> idx_s[j][q]
    const int64_t* ip = nn_idx + ((size_t)bg * N + n0) * k;
    const int cnt = min(256, N - n0) * k;
    for (int i = tid; i < cnt; i += 256) {
      const int q = i / k, j = i - q * k;
      idx_s[j * 256 + q] = (int)ip[i];
    }
  }
  __syncthreads();
  if (n >= N) return;
  for (int ch = 0; ch < nch; ++ch) {
    const size_t o = ((size_t)bg * c + ch0 + ch) * N + n;
    const float xi = ldf(x + o);
    const float* r = rows + (size_t)ch * M;
    float best = r[idx_s[tid]] - xi;
    int arg = 0;
    for (int j = 1; j < k; ++j) {
      const float v = r[idx_s[j * 256 + tid]] - xi;
      if (v > best) { best = v; arg = j; }
    }
    stf(m_out + o, best);
    if (argmax) argmax[o] = (uint8_t)arg;
  }
}

// ------------------------------------------------------------------------------------------ backward
template <typename T, bool SELF>
__global__ __launch_bounds__(256) void mr_bwd_kernel(const T* __restrict__ g, const int64_t* __restrict__ nn_idx,
                                                     const uint8_t* __restrict__ argmax, T* __restrict__ gx,
                                                     T* __restrict__ gsrc, int c, int N, int M, int k, int CH) {
  extern __shared__ float smem[];
  float* acc = smem;                                    // [CH][M]
  int* idx_s = reinterpret_cast<int*>(smem + (size_t)CH * M);   // [256][k+?] as [q*k + j]
  const int tid = threadIdx.x;
  const int bg = blockIdx.y;
  const int ch0 = blockIdx.x * CH;
  const int nch = min(CH, c - ch0);
  const size_t gbase = ((size_t)bg * c + ch0) * N;

  if (SELF) {
    for (int i = tid; i < nch * M; i += 256) acc[i] = -ldf(g + gbase + i);      // centre term (N == M)
  } else {
    for (int i = tid; i < nch * M; i += 256) acc[i] = 0.0f;
    for (int i = tid; i < nch * N; i += 256) stf(gx + gbase + i, -ldf(g + gbase + i));
  }
  for (int n0 = 0; n0 < N; n0 += 256) {
    __syncthreads();                                    // acc init done / previous tile's idx consumed
    const int64_t* ip = nn_idx + ((size_t)bg * N + n0) * k;
    const int cnt = min(256, N - n0) * k;
    for (int i = tid; i < cnt; i += 256) idx_s[i] = (int)ip[i];
    __syncthreads();
    const int n = n0 + tid;
    if (n < N) {
      for (int ch = 0; ch < nch; ++ch) {
        const size_t o = gbase + (size_t)ch * N + n;
        const int j = idx_s[tid * k + argmax[o]];
        atomicAdd(&acc[(size_t)ch * M + j], ldf(g + o));
      }
    }
  }
  __syncthreads();
  T* out = SELF ? gx : gsrc;
  const size_t obase = ((size_t)bg * c + ch0) * M;
  for (int i = tid; i < nch * M; i += 256) stf(out + obase + i, acc[i]);
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
  if (dtype != GKG_F32 && dtype != GKG_BF16) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_fwd: dtype");
  if (BG > 65535) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_fwd: BG <= 65535");
  const int extra = k * 256 * 4;
  // channels per workgroup: as many source rows as fit, but keep enough workgroups to fill the chip
  int CH = pick_ch(c, M, extra);
  if (CH < 1) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_fwd: M too large for the LDS-resident source row");
  const long ntiles = (N + 255) / 256;
  while (CH > 4 && ntiles * ((c + CH - 1) / CH) * BG < 1024) CH = (CH + 1) / 2;
  hipError_t e = dtype == GKG_F32 ? mr_fwd_launch<float>(x, src, nn_idx, m_out, argmax, BG, c, N, M, k, CH, (hipStream_t)stream)
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
  if (dtype != GKG_F32 && dtype != GKG_BF16) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_bwd: dtype");
  if (BG > 65535) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_bwd: BG <= 65535");
  const int extra = k * 256 * 4;
  int CH = pick_ch(c, M, extra);
  if (CH < 1) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_mr_bwd: M too large for the LDS-resident accumulator row");
  while (CH > 2 && (long)((c + CH - 1) / CH) * BG < 1024) CH = (CH + 1) / 2;
  hipError_t e = dtype == GKG_F32 ? mr_bwd_launch<float>(g, nn_idx, argmax, gx, gsrc, BG, c, N, M, k, CH, (hipStream_t)stream)
                                  : mr_bwd_launch<uint16_t>(g, nn_idx, argmax, gx, gsrc, BG, c, N, M, k, CH, (hipStream_t)stream);
  if (e != hipSuccess) return gkg_fail_hip(e, "mr_bwd_kernel");
  return 0;
}
