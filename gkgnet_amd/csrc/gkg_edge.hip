// gkg_edge.hip — EdgeConv aggregation (reference vig_model/torch_vertex.py:82-101), the graph convolution GKGNet's configs
// do not select ('mr' is) but Grapher(conv='edge') offers.
//
// Reference:  out = max_k  act(norm(Conv2d_1x1,groups=4( cat[x_i, x_j - x_i] )))      on a (B, 2C, N, k) tensor.
// With groups = 4 (hard-coded in BasicConv, torch_nn.py:61) the output channels of groups 2 and 3 see only the (x_j - x_i)
// half, and a 1x1 convolution is linear, so for those channels
//     z[b][o][n][k] = Q[b][o][idx[b][n][k]] - Qc[b][o][n] + bias[o],      Q = W (src half), Qc = W (x half)
// — two ordinary per-NODE projections (k times less GEMM work than the reference and no (B, 2C, N, k) tensor) followed by
// the gather this file implements.  (Groups 0 and 1 see only x_i: no neighbour term, handled by the caller.)
//   gkg_edge_stats      per-channel sums of (Q[j] - Qc) and its square over all (b, n, k): train-mode BN statistics
//   gkg_edge_fwd        out = max_k act(a z' + c), z' = Q[j] - Qc, first-maximum argmax saved for the backward
//   gkg_edge_bwd_stats  sum g and sum g zhat over the argmax elements (g = dout * act'): dbeta, dgamma and the two means of
//                       the BN backward
//   gkg_edge_bwd        dz[n][k] = a ( g [k == argmax] - mg - zhat[n][k] mgz ) for EVERY edge (train-mode BN spreads the
//                       gradient over the batch): dQ[idx] += dz (atomics, dQ zero on entry), dQc[n] = - sum_k dz
// Layout: channel-major (B, O, N) / (B, O, M) fp32, nn_idx (B, N, k) int64.  act: 0 none, 1 GELU (erf), 2 ReLU.
#include "gkg_common.h"

namespace gkg {

// out-of-range neighbour indices (a caller-supplied edge_index, or a k-NN over non-finite inputs) are clamped into the row,
// like the max-relative kernels do (gkg_mr.hip clamp_idx): never an out-of-bounds read or atomic
__device__ __forceinline__ int edge_idx(int64_t v, int M) { return (int)(v < 0 ? 0 : (v >= M ? M - 1 : v)); }

__device__ __forceinline__ float edge_act(float u, int act) {
  if (act == 1) return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f));
  if (act == 2) return u > 0.f ? u : 0.f;
  return u;
}
__device__ __forceinline__ float edge_act_grad(float u, int act) {
  if (act == 1) return 0.5f * (1.0f + erff(u * 0.70710678118654752440f)) + u * 0.39894228040143267794f * __expf(-0.5f * u * u);
  if (act == 2) return u > 0.f ? 1.f : 0.f;
  return 1.f;
}

__device__ __forceinline__ double block_sum(double v, double* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void edge_stats_kernel(const float* __restrict__ qs, const float* __restrict__ qc,
                                                         const int64_t* __restrict__ idx, double* __restrict__ sums,
                                                         int O, int N, int M, int k) {
  __shared__ double sm[4];
  const int n = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
  double s1 = 0.0, s2 = 0.0;
  if (n < N) {
    const float* q = qs + ((size_t)b * O + o) * M;
    const float c0 = qc[((size_t)b * O + o) * N + n];
    const int64_t* ip = idx + ((size_t)b * N + n) * k;
    for (int kk = 0; kk < k; ++kk) {
      const double v = (double)(q[edge_idx(ip[kk], M)] - c0);
      s1 += v; s2 += v * v;
    }
  }
  s1 = block_sum(s1, sm);
  s2 = block_sum(s2, sm);
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(sums + o, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(sums + O + o, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(256) void edge_fwd_kernel(const float* __restrict__ qs, const float* __restrict__ qc,
                                                       const int64_t* __restrict__ idx, const float* __restrict__ a,
                                                       const float* __restrict__ c, float* __restrict__ out,
                                                       uint8_t* __restrict__ argmax, int O, int N, int M, int k, int act) {
  const int n = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
  if (n >= N) return;
  const float* q = qs + ((size_t)b * O + o) * M;
  const size_t at = ((size_t)b * O + o) * N + n;
  const float c0 = qc[at], av = a[o], cv = c[o];
  const int64_t* ip = idx + ((size_t)b * N + n) * k;
  float best = 0.f;
  int bk = 0;
  for (int kk = 0; kk < k; ++kk) {
    const float v = edge_act(__builtin_fmaf(av, q[edge_idx(ip[kk], M)] - c0, cv), act);
    if (kk == 0 || v > best || (v != v && best == best)) { best = v; bk = kk; }      // first maximum; NaN propagates
  }
  out[at] = best;
  if (argmax) argmax[at] = (uint8_t)bk;
}

__global__ __launch_bounds__(256) void edge_bwd_stats_kernel(const float* __restrict__ g, const float* __restrict__ qs,
                                                             const float* __restrict__ qc, const int64_t* __restrict__ idx,
                                                             const uint8_t* __restrict__ argmax, const float* __restrict__ a,
                                                             const float* __restrict__ c, const float* __restrict__ mean0,
                                                             const float* __restrict__ invstd, double* __restrict__ sums,
                                                             int O, int N, int M, int k, int act) {
  __shared__ double sm[4];
  const int n = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
  double t1 = 0.0, t2 = 0.0;
  if (n < N) {
    const size_t at = ((size_t)b * O + o) * N + n;
    const float z = qs[((size_t)b * O + o) * M + edge_idx(idx[((size_t)b * N + n) * k + argmax[at]], M)] - qc[at];
    const float gv = g[at] * edge_act_grad(__builtin_fmaf(a[o], z, c[o]), act);
    t1 = gv;
    t2 = (double)gv * (double)((z - mean0[o]) * invstd[o]);
  }
  t1 = block_sum(t1, sm);
  t2 = block_sum(t2, sm);
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(sums + o, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(sums + O + o, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <bool DENSE>
__global__ __launch_bounds__(256) void edge_bwd_kernel(const float* __restrict__ g, const float* __restrict__ qs,
                                                       const float* __restrict__ qc, const int64_t* __restrict__ idx,
                                                       const uint8_t* __restrict__ argmax, const float* __restrict__ a,
                                                       const float* __restrict__ c, const float* __restrict__ mean0,
                                                       const float* __restrict__ invstd, const float* __restrict__ mg,
                                                       const float* __restrict__ mgz, float* __restrict__ dqs,
                                                       float* __restrict__ dqc, int O, int N, int M, int k, int act) {
  const int n = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
  if (n >= N) return;
  const size_t row = (size_t)b * O + o, at = row * N + n;
  const float* q = qs + row * M;
  float* dq = dqs + row * M;
  const int64_t* ip = idx + ((size_t)b * N + n) * k;
  const float c0 = qc[at], av = a[o], cv = c[o];
  const int ka = argmax[at];
  if (!DENSE) {                                   // no batch statistics in the way: only the winning edge carries gradient
    const int j = edge_idx(ip[ka], M);
    const float dz = av * g[at] * edge_act_grad(__builtin_fmaf(av, q[j] - c0, cv), act);
    atomicAdd(dq + j, dz);
    dqc[at] = -dz;
    return;
  }
  const float m0 = mean0[o], is = invstd[o], gm = mg[o], gz = mgz[o];
  float acc = 0.f;
  for (int kk = 0; kk < k; ++kk) {
    const int j = edge_idx(ip[kk], M);
    const float z = q[j] - c0;
    const float gv = kk == ka ? g[at] * edge_act_grad(__builtin_fmaf(av, z, cv), act) : 0.f;
    const float dz = av * (gv - gm - (z - m0) * is * gz);
    atomicAdd(dq + j, dz);
    acc += dz;
  }
  dqc[at] = -acc;
}

static int edge_check(const void* p0, const void* p1, const void* p2, int B, int O, int N, int M, int k, const char* who) {
  if (!p0 || !p1 || !p2) return gkg_fail(GKG_ERR_NULL, who);
  if (B <= 0 || O <= 0 || N <= 0 || M <= 0 || k <= 0 || k > 255 || O > 65535 || B > 65535) return gkg_fail(GKG_ERR_SHAPE, who);
  return 0;
}

}  // namespace gkg
using namespace gkg;

extern "C" int gkg_edge_stats(const float* qs, const float* qc, const int64_t* nn_idx, double* sums, int B, int O, int N,
                              int M, int k, void* stream) {
  if (int rc = edge_check(qs, qc, nn_idx, B, O, N, M, k, "gkg_edge_stats: bad pointer / size")) return rc;
  if (!sums) return gkg_fail(GKG_ERR_NULL, "gkg_edge_stats: sums is null");
  hipLaunchKernelGGL(edge_stats_kernel, dim3((N + 255) / 256, O, B), dim3(256), 0, (hipStream_t)stream, qs, qc, nn_idx, sums, O, N, M, k);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "edge_stats_kernel");
}

extern "C" int gkg_edge_fwd(const float* qs, const float* qc, const int64_t* nn_idx, const float* a, const float* c,
                            float* out, uint8_t* argmax, int B, int O, int N, int M, int k, int act, void* stream) {
  if (int rc = edge_check(qs, qc, nn_idx, B, O, N, M, k, "gkg_edge_fwd: bad pointer / size")) return rc;
  if (!a || !c || !out || act < 0 || act > 2) return gkg_fail(GKG_ERR_NULL, "gkg_edge_fwd: a, c, out required; act in 0..2");
  hipLaunchKernelGGL(edge_fwd_kernel, dim3((N + 255) / 256, O, B), dim3(256), 0, (hipStream_t)stream, qs, qc, nn_idx, a, c, out,
                     argmax, O, N, M, k, act);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "edge_fwd_kernel");
}

extern "C" int gkg_edge_bwd_stats(const float* g, const float* qs, const float* qc, const int64_t* nn_idx,
                                  const uint8_t* argmax, const float* a, const float* c, const float* mean0,
                                  const float* invstd, double* sums, int B, int O, int N, int M, int k, int act, void* stream) {
  if (int rc = edge_check(qs, qc, nn_idx, B, O, N, M, k, "gkg_edge_bwd_stats: bad pointer / size")) return rc;
  if (!g || !argmax || !a || !c || !mean0 || !invstd || !sums) return gkg_fail(GKG_ERR_NULL, "gkg_edge_bwd_stats: null pointer");
  hipLaunchKernelGGL(edge_bwd_stats_kernel, dim3((N + 255) / 256, O, B), dim3(256), 0, (hipStream_t)stream, g, qs, qc, nn_idx, argmax,
                     a, c, mean0, invstd, sums, O, N, M, k, act);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "edge_bwd_stats_kernel");
}

// mg / mgz NULL: the normalisation (if any) uses fixed statistics — only the winning edge of each (b, o, n) carries gradient.
extern "C" int gkg_edge_bwd(const float* g, const float* qs, const float* qc, const int64_t* nn_idx, const uint8_t* argmax,
                            const float* a, const float* c, const float* mean0, const float* invstd, const float* mg,
                            const float* mgz, float* dqs, float* dqc, int B, int O, int N, int M, int k, int act, void* stream) {
  if (int rc = edge_check(qs, qc, nn_idx, B, O, N, M, k, "gkg_edge_bwd: bad pointer / size")) return rc;
  if (!g || !argmax || !a || !c || !dqs || !dqc) return gkg_fail(GKG_ERR_NULL, "gkg_edge_bwd: null pointer");
  const bool dense = mg != nullptr;
  if (dense && (!mgz || !mean0 || !invstd)) return gkg_fail(GKG_ERR_NULL, "gkg_edge_bwd: batch-statistics backward needs mean0, invstd, mg, mgz");
  dim3 grid((N + 255) / 256, O, B);
  if (dense) hipLaunchKernelGGL((edge_bwd_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, g, qs, qc, nn_idx, argmax, a, c,
                                mean0, invstd, mg, mgz, dqs, dqc, O, N, M, k, act);
  else hipLaunchKernelGGL((edge_bwd_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, g, qs, qc, nn_idx, argmax, a, c,
                          mean0, invstd, mg, mgz, dqs, dqc, O, N, M, k, act);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "edge_bwd_kernel");
}
