// gkg_dense.hip — token-major (rows = tokens, columns = channels) helpers around the dense 1x1 projections
// of the Grapher block (reference torch_vertex.py:290-306, torch_nn.py:57-69, :334-360).
//
// The projections themselves are plain fp32 GEMMs  Y(T,Cout) = X(T,Cin) W^T  and run in the vendor GEMM
// library on the MFMA units (one large GEMM per layer instead of 32 per-image ones).  Everything between two
// GEMMs is bandwidth work on L2/MALL-resident activations and lives here, fused as far as train-mode
// batch-norm allows (its batch statistics are a global reduction, so every BN costs one reduce pass and one
// apply pass in each direction):
//
//   nchw_to_tm / tm_affine_to_nchw   layout change at the module boundary, the latter fused with BN-apply + residual
//   col_stats + bn_finalize          per-channel sum / sum-of-squares (deterministic two-stage) -> scale/shift,
//                                    saved mean/invstd, running-stat update (momentum, unbiased var)
//   affine_act                       out = act(a*y + c) (+ residual); act = none | GELU(erf)
//   bn_bwd_stats / bn_bwd_apply      dz = dout * act'(z);  sum dz, sum dz*yhat;  dy = a*(dz - mean(dz) - yhat*mean(dz*yhat))
#include "gkg_common.h"

namespace gkg {

// ------------------------------------------------------------------------------------------ layout
// (B, C, N) channel-major -> (B*N, C) token-major through a padded 32x32 LDS tile (coalesced both sides).
template <typename OutT>
__global__ __launch_bounds__(256) void nchw_to_tm_kernel(const float* __restrict__ x, OutT* __restrict__ out,
                                                         int C, int N, const float* __restrict__ img_scale,
                                                         const float* __restrict__ add_tm = nullptr) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = c0 + ty + 8 * i, n = n0 + tx;
    tile[ty + 8 * i][tx] = (ch < C && n < N) ? x[((size_t)b * C + ch) * N + n] : 0.f;
  }
  const float sc = img_scale ? img_scale[b] : 1.0f;       // stochastic depth: per-image keep / (1 - p) factor
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, ch = c0 + tx;
    if (ch < C && n < N) {
      const size_t o = ((size_t)b * N + n) * C + ch;
      float v = tile[tx][ty + 8 * i];
      if (add_tm) v += add_tm[o];                 // a second gradient of the same tensor that arrives token-major
      stf(out + o, img_scale ? v * sc : v);
    }
  }
}

// out(B,C,N) = act(a[ch]*y_tm[t][ch] + c[ch]) + res(B,C,N)     (a/c/res optional)
__global__ __launch_bounds__(256) void tm_affine_to_nchw_kernel(const float* __restrict__ y, const float* __restrict__ a,
                                                                const float* __restrict__ cs, const float* __restrict__ res,
                                                                float* __restrict__ out, int C, int N,
                                                                const float* __restrict__ img_scale, BnDerive d,
                                                                const float* __restrict__ res_tm = nullptr,
                                                                float* __restrict__ out_tm = nullptr) {
  // res_tm / out_tm (round 5): the residual given token-major (the block's own token-major copy of its input) and the result
  // written token-major as well — what a GrapherLabel behind this block reads as keys / values (torch_vertex.py:392-403) —
  // from the pass that writes the NCHW tensor anyway.
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  float av = 1.f, cv = 0.f;
  const bool affine = a != nullptr || d.sums != nullptr;
  if (c0 + tx < C) {
    if (d.sums) {                                  // every thread derives the pair of its channel (8 rows share it)
      const bool side = blockIdx.x == 0 && b == 0 && ty == 0;
      bn_derive_channel(d, c0 + tx, d.sums[c0 + tx], d.sums[C + c0 + tx], side, av, cv);
      if (side && c0 + tx == 0 && d.nbt) *d.nbt += 1;
    } else if (a) {
      av = a[c0 + tx]; cv = cs[c0 + tx];
    }
  }
  if (d.sums && blockIdx.x == 0 && blockIdx.y == 0 && b == 0)
    for (size_t i = threadIdx.x; i < d.zero_doubles; i += 256) d.zero_buf[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, ch = c0 + tx;
    float v = 0.f;
    if (ch < C && n < N) {
      const size_t ot = ((size_t)b * N + n) * C + ch;
      v = y[ot];
      if (affine) v = __builtin_fmaf(av, v, cv);
      if (res_tm) {
        if (img_scale) v *= img_scale[b];
        v += res_tm[ot];
        if (out_tm) out_tm[ot] = v;
      }
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = c0 + ty + 8 * i, n = n0 + tx;
    if (ch < C && n < N) {
      const size_t o = ((size_t)b * C + ch) * N + n;
      float v = tile[tx][ty + 8 * i];
      if (!res_tm) {
        if (img_scale) v *= img_scale[b];           // DropPath (reference torch_vertex.py:332): branch * mask / keep
        if (res) v += res[o];
      }
      out[o] = v;
    }
  }
}

// Pooled key set of a Grapher with r > 1 (reference torch_vertex.py:194-196: F.avg_pool2d(x, r, r) of the (B, C, H, W) map) on a
// token-major map x (B, H, W, C) given as a view (row pitch ldx, chunk: gkg_common.h "XM layout") -> out (B, H/r, W/r, C) plain.
// Floor mode like the reference's pooling: rows / columns past the last full window are ignored.  Window sum in (h, w) order,
// one division by r^2.  One thread = one pooled token x 4 channels.
__global__ __launch_bounds__(256) void avgpool_tm_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int C,
                                                         int r, int ldx, int xchunk, size_t total4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int C4 = C >> 2, Hr = H / r, Wr = W / r;
  const int cg = (int)(i % C4);
  size_t t = i / C4;
  const int wo = (int)(t % Wr); t /= Wr;
  const int ho = (int)(t % Hr);
  const size_t b = t / Hr;
  const float* p = x + ((b * H + (size_t)ho * r) * W + (size_t)wo * r) * ldx + xm_col(4 * cg, xchunk);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int dh = 0; dh < r; ++dh)
    for (int dw = 0; dw < r; ++dw) {
      const float4 v = *reinterpret_cast<const float4*>(p + ((size_t)dh * W + dw) * ldx);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  const float d = (float)(r * r);
  *reinterpret_cast<float4*>(out + 4 * i) = make_float4(s.x / d, s.y / d, s.z / d, s.w / d);
}

// ------------------------------------------------------------------------------------------ column statistics
// part[q][chunk][0][c] = sum_r y[q][r][c], part[q][chunk][1][c] = sum_r y^2 over the chunk's row range.
// 2-D decomposition: a workgroup owns one 64-channel column tile (16 float4 groups) x one row chunk, its 256
// threads are 16 column groups x 16 row lanes; grid = (row chunks, column tiles, nb) where q = blockIdx.z selects
// one of `nb` stacked (R, C) matrices (the 4 groups of the grouped projection).  Wide-and-short matrices (the
// label path's (B*L, 4C)) therefore still fill the chip, and the number of partial rows the second stage has
// to walk stays <= 256.
constexpr int ST_CG = 16;                           // float4 column groups per tile (64 channels)
constexpr int ST_RL = 16;                           // row lanes

__device__ __forceinline__ void stats_block_reduce(float (*red)[ST_RL][4 * ST_CG], const float4& s, const float4& sq,
                                                   int cg, int rl, float* __restrict__ part, int C, int tile,
                                                   double* __restrict__ asums = nullptr) {
  *reinterpret_cast<float4*>(&red[0][rl][4 * cg]) = s;
  *reinterpret_cast<float4*>(&red[1][rl][4 * cg]) = sq;
  __syncthreads();
  const int t = threadIdx.x;
  if (t < 8 * ST_CG) {
    const int which = t >> 6, col = t & 63;
    float acc = 0.f;
#pragma unroll
    for (int l = 0; l < ST_RL; ++l) acc += red[which][l][col];
    const int ch = tile * 4 * ST_CG + col;
    if (ch < C) {
      if (asums) __hip_atomic_fetch_add(asums + (size_t)which * C + ch, (double)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else part[(size_t)which * C + ch] = acc;
    }
  }
}

// SHIFT: sums are taken of (y - y[row 0]) instead of y: any sample of the column is within a few standard deviations of
// its mean, so the second moment no longer cancels catastrophically when |mean| >> std (E[y^2] - E[y]^2 in fp32 loses
// ~1e-7 * mean^2 / var of the variance); reduce_finalize_kernel adds the shift back.  The cross-rank (SyncBN) form keeps
// plain sums: its partial sums travel through an all-reduce and every rank would shift by a different row.
template <bool SHIFT>
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ y, float* __restrict__ part,
                                                        int R, int C, int rows_per_block) {
  __shared__ float red[2][ST_RL][4 * ST_CG];
  const int tid = threadIdx.x;
  const int cg = tid & (ST_CG - 1), rl = tid >> 4;
  const int cgi = blockIdx.y * ST_CG + cg;
  const int q = blockIdx.z;
  y += (size_t)q * R * C;
  part += ((size_t)q * gridDim.x + blockIdx.x) * 2 * C;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(R, r0 + rows_per_block);
  float4 s = make_float4(0, 0, 0, 0), sq = make_float4(0, 0, 0, 0);
  if (cgi < (C >> 2)) {
    const float* p = y + (size_t)4 * cgi;
    const float4 sh = SHIFT ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    auto ld = [&](int row) {
      float4 v = *reinterpret_cast<const float4*>(p + (size_t)row * C);
      if (SHIFT) { v.x -= sh.x; v.y -= sh.y; v.z -= sh.z; v.w -= sh.w; }
      return v;
    };
    int r = r0 + rl;
    for (; r + 3 * ST_RL < r1; r += 4 * ST_RL) {        // 4 independent loads in flight
      const float4 v0 = ld(r);
      const float4 v1 = ld(r + ST_RL);
      const float4 v2 = ld(r + 2 * ST_RL);
      const float4 v3 = ld(r + 3 * ST_RL);
      s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
      s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
      sq.x += (v0.x * v0.x + v1.x * v1.x) + (v2.x * v2.x + v3.x * v3.x);
      sq.y += (v0.y * v0.y + v1.y * v1.y) + (v2.y * v2.y + v3.y * v3.y);
      sq.z += (v0.z * v0.z + v1.z * v1.z) + (v2.z * v2.z + v3.z * v3.z);
      sq.w += (v0.w * v0.w + v1.w * v1.w) + (v2.w * v2.w + v3.w * v3.w);
    }
    for (; r < r1; r += ST_RL) {
      const float4 v = ld(r);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w;
    }
  }
  stats_block_reduce(red, s, sq, cg, rl, part, C, blockIdx.y);
}

// Same two-stage scheme for the backward statistics: sum dz and sum dz*yhat, dz = dout * act'(a*y + c),
// yhat = (y - mean) * invstd.  dout[q] may be a column slice of a wider matrix (row pitch ldg).
template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                           const float* __restrict__ a, const float* __restrict__ cs,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           float* __restrict__ part, int R, int C, int rows_per_block,
                                                           int ldg, size_t g_bstride, float* __restrict__ dz_out,
                                                           double* __restrict__ asums = nullptr,
                                                           const float* __restrict__ row_scale = nullptr, int rows_per_scale = 1) {
  __shared__ float red[2][ST_RL][4 * ST_CG];
  const int tid = threadIdx.x;
  const int cg = tid & (ST_CG - 1), rl = tid >> 4;
  const int cgi = blockIdx.y * ST_CG + cg;
  const int q = blockIdx.z;
  y += (size_t)q * R * C; dout += (size_t)q * g_bstride;
  if (dz_out) dz_out += (size_t)q * R * C;
  a += (size_t)q * C; cs += (size_t)q * C; mean += (size_t)q * C; invstd += (size_t)q * C;
  if (asums) asums += (size_t)q * 2 * C;                // fp64 column sums accumulated with atomics (no partials, no second stage)
  else part += ((size_t)q * gridDim.x + blockIdx.x) * 2 * C;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(R, r0 + rows_per_block);
  float4 s = make_float4(0, 0, 0, 0), sq = make_float4(0, 0, 0, 0);
  if (cgi < (C >> 2)) {
    const float4 a4 = *reinterpret_cast<const float4*>(a + 4 * cgi);
    const float4 c4 = *reinterpret_cast<const float4*>(cs + 4 * cgi);
    const float4 m4 = *reinterpret_cast<const float4*>(mean + 4 * cgi);
    const float4 i4 = *reinterpret_cast<const float4*>(invstd + 4 * cgi);
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += ST_RL) {
      const float4 g = *reinterpret_cast<const float4*>(dout + (size_t)r * ldg + 4 * cgi);
      const float4 v = *reinterpret_cast<const float4*>(y + (size_t)r * C + 4 * cgi);
      float4 dz = g;
      if (row_scale) {                                // DropPath: the branch output was scaled per image, so is its gradient
        const float sc = row_scale[r / rows_per_scale];
        dz.x *= sc; dz.y *= sc; dz.z *= sc; dz.w *= sc;
      }
      if (ACT == 1) {
        dz.x *= gelu_grad_f(__builtin_fmaf(a4.x, v.x, c4.x)); dz.y *= gelu_grad_f(__builtin_fmaf(a4.y, v.y, c4.y));
        dz.z *= gelu_grad_f(__builtin_fmaf(a4.z, v.z, c4.z)); dz.w *= gelu_grad_f(__builtin_fmaf(a4.w, v.w, c4.w));
        if (dz_out) *reinterpret_cast<float4*>(dz_out + (size_t)r * C + 4 * cgi) = dz;
      }
      s.x += dz.x; s.y += dz.y; s.z += dz.z; s.w += dz.w;
      sq.x += dz.x * ((v.x - m4.x) * i4.x); sq.y += dz.y * ((v.y - m4.y) * i4.y);
      sq.z += dz.z * ((v.z - m4.z) * i4.z); sq.w += dz.w * ((v.w - m4.w) * i4.w);
    }
  }
  stats_block_reduce(red, s, sq, cg, rl, part, C, blockIdx.y, asums);
}

// sums[q][0][c], sums[q][1][c] = fixed-order (deterministic) double-precision reduction of the chunk partials.
// 32 columns x 8 partial-lanes per workgroup; each lane sums every 8th partial with 8 loads in flight, the 8
// lane sums are combined in a fixed order.  Optionally also scatters the two halves to out0/out1 ([nb][C]),
// which is how the BN backward gets dbeta / dgamma.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ sums,
                                                              int nblk, int C2, float* __restrict__ out0,
                                                              float* __restrict__ out1) {
  __shared__ double lane_sum[8][32];
  const int t = threadIdx.x & 31;
  const int col = blockIdx.x * 32 + t;
  const int ln = threadIdx.x >> 5;
  const int q = blockIdx.y;
  double acc = 0.0;
  if (col < C2) {
    const float* p = part + (size_t)q * nblk * C2 + col;
    int b = ln;
    for (; b + 56 < nblk; b += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(b + 8 * u) * C2];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; b < nblk; b += 8) acc += (double)p[(size_t)b * C2];
  }
  lane_sum[ln][t] = acc;
  __syncthreads();
  if (ln == 0 && col < C2) {
    double tot = lane_sum[0][t];
#pragma unroll
    for (int l = 1; l < 8; ++l) tot += lane_sum[l][t];
    const float r = (float)tot;
    sums[(size_t)q * C2 + col] = r;
    const int C = C2 >> 1;
    if (out0) { if (col < C) out0[(size_t)q * C + col] = r; else out1[(size_t)q * C + col - C] = r; }
  }
}

// Forward statistics: fixed-order reduction of the partials (sum and sum-of-squares of 32 channels per
// workgroup, 4 lanes each) fused with the BN parameter computation of bn_finalize_kernel below.
__global__ __launch_bounds__(256) void reduce_finalize_kernel(const float* __restrict__ part, int nblk,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ bias, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, float* __restrict__ a,
                                                              float* __restrict__ cs, float* __restrict__ mean,
                                                              float* __restrict__ invstd, int R, int C, float momentum, float eps,
                                                              long long* __restrict__ num_batches_tracked,
                                                              const float* __restrict__ y_shift_row /* y: row 0 is the shift */) {
  __shared__ double lane_sum[8][32];
  if (num_batches_tracked && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
  const int t = threadIdx.x & 31;
  const int ch = blockIdx.x * 32 + t;
  const int ln = threadIdx.x >> 5;                 // lanes 0-3: sum (every 4th partial); lanes 4-7: sum of squares
  const int q = blockIdx.y;
  const int C2 = 2 * C;
  double acc = 0.0;
  if (ch < C) {
    const float* p = part + (size_t)q * nblk * C2 + (ln >> 2) * C + ch;
    int b = ln & 3;
    for (; b + 28 < nblk; b += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(b + 4 * u) * C2];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; b < nblk; b += 4) acc += (double)p[(size_t)b * C2];
  }
  lane_sum[ln][t] = acc;
  __syncthreads();
  if (ln != 0 || ch >= C) return;
  const double S = (lane_sum[0][t] + lane_sum[1][t]) + (lane_sum[2][t] + lane_sum[3][t]);
  const double Q = (lane_sum[4][t] + lane_sum[5][t]) + (lane_sum[6][t] + lane_sum[7][t]);
  const size_t o = (size_t)q * C + ch;
  const double ms = S / R;                         // mean of the shifted values
  double var = Q / R - ms * ms;
  if (var < 0.0) var = 0.0;
  const double m = ms + (y_shift_row ? (double)y_shift_row[(size_t)q * R * C + ch] : 0.0);
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float av = gamma[o] * is;
  a[o] = av;
  cs[o] = (float)((double)beta[o] - (double)av * m);
  mean[o] = (float)m;
  invstd[o] = is;
  if (running_mean) {
    const float bv = bias ? bias[o] : 0.f;
    running_mean[o] = (1.f - momentum) * running_mean[o] + momentum * ((float)m + bv);
    const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
    running_var[o] = (1.f - momentum) * running_var[o] + momentum * (float)unb;
  }
}

// Train-mode BN parameters from the column sums of Y (which EXCLUDES the conv bias `bias`, folded here):
//   mean_y = S/R, var = Q/R - mean_y^2 (biased), invstd = rsqrt(var + eps)
//   a = gamma*invstd, c = beta - a*mean_y             (bias cancels in train mode)
//   running_mean <- (1-mom)*rm + mom*(mean_y + bias), running_var <- (1-mom)*rv + mom*var*R/(R-1)
__global__ void bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, const float* __restrict__ bias,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ a, float* __restrict__ cs, float* __restrict__ mean,
                                   float* __restrict__ invstd, const float* __restrict__ count, int C, float momentum,
                                   float eps, long long* __restrict__ num_batches_tracked) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (num_batches_tracked && ch == 0 && blockIdx.y == 0) *num_batches_tracked += 1;
  if (ch >= C) return;
  const double R = (double)count[0];               // rows over ALL ranks (device scalar: it rode the all-reduce)
  const int q = blockIdx.y;
  sums += (size_t)q * 2 * C; gamma += (size_t)q * C; beta += (size_t)q * C; a += (size_t)q * C; cs += (size_t)q * C;
  mean += (size_t)q * C; invstd += (size_t)q * C;
  if (bias) bias += (size_t)q * C;
  if (running_mean) { running_mean += (size_t)q * C; running_var += (size_t)q * C; }
  const double m = (double)sums[ch] / R;
  double var = (double)sums[C + ch] / R - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float av = gamma[ch] * is;
  a[ch] = av;
  cs[ch] = beta[ch] - av * (float)m;
  mean[ch] = (float)m;
  invstd[ch] = is;
  if (running_mean) {
    const float b = bias ? bias[ch] : 0.f;
    running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * ((float)m + b);
    const double unb = R > 1.0 ? var * R / (R - 1.0) : var;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unb;
  }
}

// Eval-mode BN folded to an affine: a = gamma/sqrt(rv+eps), c = beta + a*(bias - rm)
__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ bias, const float* __restrict__ running_mean,
                                      const float* __restrict__ running_var, float* __restrict__ a,
                                      float* __restrict__ cs, int C, float eps) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= C) return;
  const float av = gamma[ch] / sqrtf(running_var[ch] + eps);
  a[ch] = av;
  cs[ch] = beta[ch] + av * ((bias ? bias[ch] : 0.f) - running_mean[ch]);
}

// ------------------------------------------------------------------------------------------ apply
// out[r][c] = act(a[c]*y[r][c] + cs[c]) (+ res[r][c]); out may have a different row pitch / column offset
// (ldo, used to write the grouped conv's output straight into the next layer's (T, 2C) input).
template <int ACT, typename OutT>
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ y, const float* __restrict__ a,
                                                         const float* __restrict__ cs, const float* __restrict__ res,
                                                         OutT* __restrict__ out, size_t total4, int C, int ldo,
                                                         size_t o_bstride, const float* __restrict__ row_scale,
                                                         int rows_per_scale, uint16_t* __restrict__ out2, BnDerive d,
                                                         int ochunk = 0) {
  extern __shared__ float ac_tab[];                 // derive mode: [2][C] scale / shift of this group
  const int C4 = C >> 2;
  const int q = blockIdx.y;
  y += (size_t)q * total4 * 4; out += (size_t)q * o_bstride;
  if (res) res += (size_t)q * total4 * 4;
  if (d.sums) {
    const bool side = blockIdx.x == 0;
    for (int ch = threadIdx.x; ch < C; ch += 256) {
      float av, cv;
      bn_derive_channel(d, (size_t)q * C + ch, d.sums[(size_t)q * 2 * C + ch], d.sums[(size_t)q * 2 * C + C + ch], side, av, cv);
      ac_tab[ch] = av; ac_tab[C + ch] = cv;
    }
    if (side && q == 0) {
      if (threadIdx.x == 0 && d.nbt) *d.nbt += 1;
      for (size_t i = threadIdx.x; i < d.zero_doubles; i += 256) d.zero_buf[i] = 0.0;
    }
    __syncthreads();
    a = ac_tab; cs = ac_tab + C;
  } else {
    a += (size_t)q * C; cs += (size_t)q * C;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    const size_t r = i / C4;
    const int cg = (int)(i - r * C4);
    const float4 v = *reinterpret_cast<const float4*>(y + 4 * i);
    const float4 a4 = *reinterpret_cast<const float4*>(a + 4 * cg);
    const float4 c4 = *reinterpret_cast<const float4*>(cs + 4 * cg);
    float4 o;
    o.x = __builtin_fmaf(a4.x, v.x, c4.x); o.y = __builtin_fmaf(a4.y, v.y, c4.y);
    o.z = __builtin_fmaf(a4.z, v.z, c4.z); o.w = __builtin_fmaf(a4.w, v.w, c4.w);
    if (ACT == 1) { o.x = gelu_f(o.x); o.y = gelu_f(o.y); o.z = gelu_f(o.z); o.w = gelu_f(o.w); }
    if (row_scale) {                                // DropPath: one factor per image = rows_per_scale consecutive rows
      const float sc = row_scale[r / rows_per_scale];
      o.x *= sc; o.y *= sc; o.z *= sc; o.w *= sc;
    }
    if (res) {
      const float4 rv = *reinterpret_cast<const float4*>(res + 4 * i);
      o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    // ochunk > 0: column ch lands at ch + (ch / ochunk) * ochunk — the x half of the grouped projection's [x | m] operand
    // buffer (gkg_hip.h "XM layout"): the Grapher's fc1 writes its result where BasicConv reads it
    const int oc = ochunk > 0 ? 4 * cg + (4 * cg / ochunk) * ochunk : 4 * cg;
    stf4(out + r * (size_t)ldo + oc, o);
    if (out2) stf4(out2 + r * (size_t)ldo + oc, o);           // second, bf16 copy of the same values (gkg_affine_act_dual)
  }
}

// The same apply pass for a bf16 input matrix (the output of a library convolution under bf16 autocast, channels-last =
// token-major): out32 (fp32) and / or out16 (its bf16 rounding), 8 values per thread and step.
template <int ACT>
__global__ __launch_bounds__(256) void affine_act_bf16in_kernel(const uint16_t* __restrict__ y, const float* __restrict__ a,
                                                                const float* __restrict__ cs, float* __restrict__ out32,
                                                                uint16_t* __restrict__ out16, size_t total8, int C) {
  const int C8 = C >> 3;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % C8);
    const uint4 v = *reinterpret_cast<const uint4*>(y + 8 * i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float o[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      o[2 * u] = __uint_as_float(w[u] << 16);
      o[2 * u + 1] = __uint_as_float(w[u] & 0xffff0000u);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      o[u] = __builtin_fmaf(a[8 * cg + u], o[u], cs[8 * cg + u]);
      if (ACT == 1) o[u] = gelu_f(o[u]);
    }
    if (out32) {
      stf4(out32 + 8 * i, make_float4(o[0], o[1], o[2], o[3]));
      stf4(out32 + 8 * i + 4, make_float4(o[4], o[5], o[6], o[7]));
    }
    if (out16)
      *reinterpret_cast<uint4*>(out16 + 8 * i) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]),
                                                             pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
  }
}

// dy[r][c] = a[c] * (dz - sdz[c]/R - yhat*sdzy[c]/R), dz = dout*act'(a*y+c).  dout may be strided (ldg).
// fp64-sums form of the apply pass (gkg_bn_bwd_atomic): reads the column sums the statistics pass accumulated with atomics,
// its first workgroup of every group also emits dbeta / dgamma and clears `zero_buf` (the OTHER scratch buffer: what the
// previous call accumulated into — nobody reads it any more; this call's own buffer is cleared by the next call).
template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply_d_kernel(const float* dout, const float* __restrict__ y,
                                                             const float* __restrict__ a, const float* __restrict__ cs,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const double* __restrict__ dsums, float* dy,
                                                             size_t total4, int C, int R, int ldg, size_t g_bstride,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             double* __restrict__ zero_buf, size_t zero_doubles,
                                                             const float* __restrict__ row_scale = nullptr, int rows_per_scale = 1) {
  const int C4 = C >> 2;
  const float invR = 1.0f / (float)R;
  const int q = blockIdx.y;
  y += (size_t)q * total4 * 4; dy += (size_t)q * total4 * 4; dout += (size_t)q * g_bstride;
  a += (size_t)q * C; cs += (size_t)q * C; mean += (size_t)q * C; invstd += (size_t)q * C; dsums += (size_t)q * 2 * C;
  if (blockIdx.x == 0) {
    for (int ch = threadIdx.x; ch < C; ch += 256) {
      dbeta[(size_t)q * C + ch] = (float)dsums[ch];
      dgamma[(size_t)q * C + ch] = (float)dsums[C + ch];
    }
    if (q == 0)
      for (size_t i = threadIdx.x; i < zero_doubles; i += 256) zero_buf[i] = 0.0;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    const size_t r = i / C4;
    const int cg = (int)(i - r * C4);
    const float4 g = *reinterpret_cast<const float4*>(dout + r * (size_t)ldg + 4 * cg);
    const float4 v = *reinterpret_cast<const float4*>(y + 4 * i);
    const float4 a4 = *reinterpret_cast<const float4*>(a + 4 * cg);
    const float4 c4 = *reinterpret_cast<const float4*>(cs + 4 * cg);
    const float4 m4 = *reinterpret_cast<const float4*>(mean + 4 * cg);
    const float4 i4 = *reinterpret_cast<const float4*>(invstd + 4 * cg);
    const double2 sa = *reinterpret_cast<const double2*>(dsums + 4 * cg), sb = *reinterpret_cast<const double2*>(dsums + 4 * cg + 2);
    const double2 qa = *reinterpret_cast<const double2*>(dsums + C + 4 * cg), qb = *reinterpret_cast<const double2*>(dsums + C + 4 * cg + 2);
    const float4 s4 = make_float4((float)sa.x, (float)sa.y, (float)sb.x, (float)sb.y);
    const float4 q4 = make_float4((float)qa.x, (float)qa.y, (float)qb.x, (float)qb.y);
    float4 dz = g;
    if (row_scale) {
      const float sc = row_scale[r / (size_t)rows_per_scale];
      dz.x *= sc; dz.y *= sc; dz.z *= sc; dz.w *= sc;
    }
    if (ACT == 1) {
      dz.x *= gelu_grad_f(__builtin_fmaf(a4.x, v.x, c4.x)); dz.y *= gelu_grad_f(__builtin_fmaf(a4.y, v.y, c4.y));
      dz.z *= gelu_grad_f(__builtin_fmaf(a4.z, v.z, c4.z)); dz.w *= gelu_grad_f(__builtin_fmaf(a4.w, v.w, c4.w));
    }
    float4 o;
    o.x = a4.x * (dz.x - s4.x * invR - ((v.x - m4.x) * i4.x) * (q4.x * invR));
    o.y = a4.y * (dz.y - s4.y * invR - ((v.y - m4.y) * i4.y) * (q4.y * invR));
    o.z = a4.z * (dz.z - s4.z * invR - ((v.z - m4.z) * i4.z) * (q4.z * invR));
    o.w = a4.w * (dz.w - s4.w * invR - ((v.w - m4.w) * i4.w) * (q4.w * invR));
    *reinterpret_cast<float4*>(dy + 4 * i) = o;
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dout, const float* __restrict__ y,
                                                           const float* __restrict__ a, const float* __restrict__ cs,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ sums, float* dy,
                                                           size_t total4, int C, int R, int ldg, size_t g_bstride,
                                                           const float* __restrict__ count) {
  const int C4 = C >> 2;
  const float invR = 1.0f / (count ? count[0] : (float)R);   // count: rows over all ranks (SyncBN)
  const int q = blockIdx.y;
  y += (size_t)q * total4 * 4; dy += (size_t)q * total4 * 4; dout += (size_t)q * g_bstride;
  a += (size_t)q * C; cs += (size_t)q * C; mean += (size_t)q * C; invstd += (size_t)q * C; sums += (size_t)q * 2 * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    const size_t r = i / C4;
    const int cg = (int)(i - r * C4);
    const float4 g = *reinterpret_cast<const float4*>(dout + r * (size_t)ldg + 4 * cg);
    const float4 v = *reinterpret_cast<const float4*>(y + 4 * i);
    const float4 a4 = *reinterpret_cast<const float4*>(a + 4 * cg);
    const float4 c4 = *reinterpret_cast<const float4*>(cs + 4 * cg);
    const float4 m4 = *reinterpret_cast<const float4*>(mean + 4 * cg);
    const float4 i4 = *reinterpret_cast<const float4*>(invstd + 4 * cg);
    const float4 s4 = *reinterpret_cast<const float4*>(sums + 4 * cg);
    const float4 q4 = *reinterpret_cast<const float4*>(sums + C + 4 * cg);
    float4 dz = g;
    if (ACT == 1) {
      dz.x *= gelu_grad_f(__builtin_fmaf(a4.x, v.x, c4.x)); dz.y *= gelu_grad_f(__builtin_fmaf(a4.y, v.y, c4.y));
      dz.z *= gelu_grad_f(__builtin_fmaf(a4.z, v.z, c4.z)); dz.w *= gelu_grad_f(__builtin_fmaf(a4.w, v.w, c4.w));
    }
    float4 o;
    o.x = a4.x * (dz.x - s4.x * invR - ((v.x - m4.x) * i4.x) * (q4.x * invR));
    o.y = a4.y * (dz.y - s4.y * invR - ((v.y - m4.y) * i4.y) * (q4.y * invR));
    o.z = a4.z * (dz.z - s4.z * invR - ((v.z - m4.z) * i4.z) * (q4.z * invR));
    o.w = a4.w * (dz.w - s4.w * invR - ((v.w - m4.w) * i4.w) * (q4.w * invR));
    *reinterpret_cast<float4*>(dy + 4 * i) = o;
  }
}

}  // namespace gkg

using namespace gkg;

static int stats_tiles(int C) { return ((C >> 2) + ST_CG - 1) / ST_CG; }

// Row chunks: enough workgroups (chunks x column tiles x nb) for ~3 per CU, at least ST_RL rows each, and at
// most 128 (256 for narrow matrices) partial rows for the second stage to reduce.
static int stats_blocks(int R, int C, int nb, int* rows_per_block) {
  const int cols = stats_tiles(C) * nb;
  int nblk = (768 + cols - 1) / cols;
  const int cap = cols >= 4 ? 128 : 256;
  if (nblk > cap) nblk = cap;
  const int by_rows = (R + ST_RL - 1) / ST_RL;
  if (nblk > by_rows) nblk = by_rows;
  if (nblk < 1) nblk = 1;
  *rows_per_block = (R + nblk - 1) / nblk;
  return (R + *rows_per_block - 1) / *rows_per_block;
}

static bool bad_c(int C) { return C <= 0 || (C & 3) != 0 || C > 4096; }

extern "C" size_t gkg_bn_workspace_bytes(int R, int C, int nb) {
  if (R <= 0 || bad_c(C) || nb <= 0) return 0;
  int rpb;
  const int nblk = stats_blocks(R, C, nb, &rpb);
  return (size_t)nb * (nblk + 1) * 2 * C * sizeof(float);
}

extern "C" int gkg_nchw_to_tm(const float* x, void* out, int B, int C, int N, int out_dtype, const float* img_scale,
                              void* stream) {
  if (!x || !out) return gkg_fail(GKG_ERR_NULL, "gkg_nchw_to_tm: null pointer");
  if (B <= 0 || C <= 0 || N <= 0 || B > 65535) return gkg_fail(GKG_ERR_SHAPE, "gkg_nchw_to_tm: bad sizes");
  if (out_dtype != GKG_F32 && out_dtype != GKG_BF16) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_nchw_to_tm: out_dtype is GKG_F32 or GKG_BF16");
  dim3 grid((N + 31) / 32, (C + 31) / 32, B);
  if (out_dtype == GKG_BF16) hipLaunchKernelGGL(nchw_to_tm_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)out, C, N, img_scale);
  else hipLaunchKernelGGL(nchw_to_tm_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, (float*)out, C, N, img_scale);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "nchw_to_tm_kernel");
}

// (B,C,N) + token-major addend -> token-major fp32: the upstream gradient of a block output that was handed out in BOTH
// layouts (gkg_tm_affine_to_nchw_dual), summed while it is re-laid out.
extern "C" int gkg_nchw_to_tm_add(const float* x, const float* add_tm, float* out, int B, int C, int N, void* stream) {
  if (!x || !add_tm || !out) return gkg_fail(GKG_ERR_NULL, "gkg_nchw_to_tm_add: null pointer");
  if (B <= 0 || C <= 0 || N <= 0 || B > 65535) return gkg_fail(GKG_ERR_SHAPE, "gkg_nchw_to_tm_add: bad sizes");
  dim3 grid((N + 31) / 32, (C + 31) / 32, B);
  hipLaunchKernelGGL(nchw_to_tm_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, out, C, N, (const float*)nullptr, add_tm);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "nchw_to_tm_kernel (add)");
}

// gkg_tm_affine_to_nchw with the residual given token-major and the result written in both layouts.
extern "C" int gkg_tm_affine_to_nchw_dual(const float* y, const float* a, const float* c, const float* res_tm, float* out,
                                          float* out_tm, int B, int C, int N, void* stream) {
  if (!y || !out || !res_tm || !out_tm || ((a == nullptr) != (c == nullptr))) return gkg_fail(GKG_ERR_NULL, "gkg_tm_affine_to_nchw_dual: null pointer");
  if (B <= 0 || C <= 0 || N <= 0 || B > 65535) return gkg_fail(GKG_ERR_SHAPE, "gkg_tm_affine_to_nchw_dual: bad sizes");
  dim3 grid((N + 31) / 32, (C + 31) / 32, B);
  hipLaunchKernelGGL(tm_affine_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, a, c, (const float*)nullptr, out, C, N,
                     (const float*)nullptr, BnDerive{}, res_tm, out_tm);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "tm_affine_to_nchw_kernel (dual)");
}

extern "C" int gkg_tm_affine_to_nchw(const float* y, const float* a, const float* c, const float* res, float* out,
                                     int B, int C, int N, const float* img_scale, void* stream) {
  if (!y || !out || ((a == nullptr) != (c == nullptr))) return gkg_fail(GKG_ERR_NULL, "gkg_tm_affine_to_nchw: null pointer");
  if (B <= 0 || C <= 0 || N <= 0 || B > 65535) return gkg_fail(GKG_ERR_SHAPE, "gkg_tm_affine_to_nchw: bad sizes");
  dim3 grid((N + 31) / 32, (C + 31) / 32, B);
  hipLaunchKernelGGL(tm_affine_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, a, c, res, out, C, N, img_scale, BnDerive{});
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "tm_affine_to_nchw_kernel");
}

extern "C" int gkg_bn_train_stats(const float* y, const float* gamma, const float* beta, const float* bias,
                                  float* running_mean, float* running_var, float* a, float* c, float* mean,
                                  float* invstd, int R, int C, int nb, float momentum, float eps,
                                  long long* num_batches_tracked, void* workspace, size_t workspace_bytes, void* stream) {
  if (!y || !gamma || !beta || !a || !c || !mean || !invstd || !workspace)
    return gkg_fail(GKG_ERR_NULL, "gkg_bn_train_stats: null pointer");
  if (R <= 0 || bad_c(C) || nb <= 0 || nb > 64) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_train_stats: need R > 0, C % 4 == 0, C <= 4096, 1 <= nb <= 64");
  if ((running_mean == nullptr) != (running_var == nullptr)) return gkg_fail(GKG_ERR_NULL, "gkg_bn_train_stats: running stats come in pairs");
  int rpb;
  const int nblk = stats_blocks(R, C, nb, &rpb);
  if (workspace_bytes < (size_t)nb * (nblk + 1) * 2 * C * sizeof(float))
    return gkg_fail(GKG_ERR_WORKSPACE, "gkg_bn_train_stats: workspace too small (gkg_bn_workspace_bytes)");
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* sums = part + (size_t)nb * nblk * 2 * C;
  hipLaunchKernelGGL(col_stats_kernel<true>, dim3(nblk, stats_tiles(C), nb), dim3(256), 0, st, y, part, R, C, rpb);
  hipLaunchKernelGGL(reduce_finalize_kernel, dim3((C + 31) / 32, nb), dim3(256), 0, st, part, nblk, gamma, beta, bias,
                     running_mean, running_var, a, c, mean, invstd, R, C, momentum, eps, num_batches_tracked, y);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "bn_train_stats");
}

extern "C" int gkg_bn_eval_affine(const float* gamma, const float* beta, const float* bias, const float* running_mean,
                                  const float* running_var, float* a, float* c, int C, float eps, void* stream) {
  if (!gamma || !beta || !running_mean || !running_var || !a || !c) return gkg_fail(GKG_ERR_NULL, "gkg_bn_eval_affine: null pointer");
  if (C <= 0) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_eval_affine: bad C");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, bias,
                     running_mean, running_var, a, c, C, eps);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "bn_eval_affine_kernel");
}

extern "C" int gkg_avgpool_tm(const float* x, int ldx, int xchunk, float* out, int B, int H, int W, int C, int r, void* stream) {
  if (!x || !out) return gkg_fail(GKG_ERR_NULL, "gkg_avgpool_tm: null pointer");
  if (ldx == 0) ldx = C;
  if (B <= 0 || H <= 0 || W <= 0 || bad_c(C) || r <= 0 || H / r <= 0 || W / r <= 0 || ldx < C || (ldx & 3) || xchunk < 0 || (xchunk & 3) ||
      (xchunk > 0 && (C % xchunk || ldx < 2 * C)) || ((size_t)x & 15) || ((size_t)out & 15))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_avgpool_tm: bad sizes (C % 4 == 0, r <= H, W; 16-byte aligned rows)");
  const size_t total4 = (size_t)B * (H / r) * (W / r) * (C >> 2);
  if ((total4 + 255) / 256 > 0x7fffffffull) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_avgpool_tm: too large for one launch");
  hipLaunchKernelGGL(avgpool_tm_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, H, W, C, r, ldx,
                     xchunk, total4);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "avgpool_tm_kernel");
}

extern "C" int gkg_affine_act(const float* y, const float* a, const float* c, const float* res, void* out, int R, int C,
                              int nb, int ldo, size_t out_bstride, int ochunk, int act, int out_dtype, const float* row_scale,
                              int rows_per_scale, void* stream) {
  if (row_scale && rows_per_scale <= 0) return gkg_fail(GKG_ERR_SHAPE, "gkg_affine_act: rows_per_scale must be positive");
  if (ochunk < 0 || (ochunk & 3) || (ochunk > 0 && (C % ochunk || ldo < 2 * C)))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_affine_act: ochunk must be a multiple of 4 dividing C, with ldo >= 2 C");
  if (!y || !a || !c || !out) return gkg_fail(GKG_ERR_NULL, "gkg_affine_act: null pointer");
  if (R <= 0 || bad_c(C) || nb <= 0 || nb > 64 || ldo < C || (ldo & 3) || (out_bstride & 3) || (act != 0 && act != 1))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_affine_act: bad sizes");
  if (out_dtype != GKG_F32 && out_dtype != GKG_BF16) return gkg_fail(GKG_ERR_UNSUPPORTED, "gkg_affine_act: out_dtype is GKG_F32 or GKG_BF16");
  const size_t total4 = (size_t)R * (C >> 2);
  const int blocks = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
  const dim3 grid(blocks, nb);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == GKG_BF16) {
    uint16_t* o = (uint16_t*)out;
    if (act == 1) hipLaunchKernelGGL((affine_act_kernel<1, uint16_t>), grid, dim3(256), 0, st, y, a, c, res, o, total4, C, ldo, out_bstride, row_scale, rows_per_scale, (uint16_t*)nullptr, BnDerive{}, ochunk);
    else hipLaunchKernelGGL((affine_act_kernel<0, uint16_t>), grid, dim3(256), 0, st, y, a, c, res, o, total4, C, ldo, out_bstride, row_scale, rows_per_scale, (uint16_t*)nullptr, BnDerive{}, ochunk);
  } else {
    float* o = (float*)out;
    if (act == 1) hipLaunchKernelGGL((affine_act_kernel<1, float>), grid, dim3(256), 0, st, y, a, c, res, o, total4, C, ldo, out_bstride, row_scale, rows_per_scale, (uint16_t*)nullptr, BnDerive{}, ochunk);
    else hipLaunchKernelGGL((affine_act_kernel<0, float>), grid, dim3(256), 0, st, y, a, c, res, o, total4, C, ldo, out_bstride, row_scale, rows_per_scale, (uint16_t*)nullptr, BnDerive{}, ochunk);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "affine_act_kernel");
}

// out = act(a * y + c) for a BF16 matrix y (R, C) with C % 8 == 0 — the output of a library convolution under bf16 autocast
// viewed token-major (channels-last) with the eval-mode BN (+ conv bias) folded into a / c (reference gkgnet.py:79-118 in
// eval mode): out_f32 (fp32, the residual stream) and / or out_bf16 (the next GEMM / convolution operand); at least one.
extern "C" int gkg_affine_act_bf16in(const void* y_bf16, const float* a, const float* c, float* out_f32, void* out_bf16, int R,
                                     int C, int act, void* stream) {
  if (!y_bf16 || !a || !c || (!out_f32 && !out_bf16)) return gkg_fail(GKG_ERR_NULL, "gkg_affine_act_bf16in: null pointer");
  if (R <= 0 || C <= 0 || (C & 7) || (act != 0 && act != 1)) return gkg_fail(GKG_ERR_SHAPE, "gkg_affine_act_bf16in: bad sizes (C % 8 == 0)");
  const size_t total8 = (size_t)R * (C >> 3);
  const int blocks = (int)((total8 + 255) / 256 > 4096 ? 4096 : (total8 + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (act == 1) hipLaunchKernelGGL((affine_act_bf16in_kernel<1>), dim3(blocks), dim3(256), 0, st, (const uint16_t*)y_bf16, a, c, out_f32, (uint16_t*)out_bf16, total8, C);
  else hipLaunchKernelGGL((affine_act_bf16in_kernel<0>), dim3(blocks), dim3(256), 0, st, (const uint16_t*)y_bf16, a, c, out_f32, (uint16_t*)out_bf16, total8, C);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "affine_act_bf16in_kernel");
}

// gkg_affine_act (one batch, contiguous rows) writing the result twice: fp32 (the residual stream) and its bf16 rounding
// (the next projection's GEMM operand under bf16 autocast inference) — saves the stand-alone cast pass between blocks.
extern "C" int gkg_affine_act_dual(const float* y, const float* a, const float* c, const float* res, float* out_f32,
                                   void* out_bf16, int R, int C, int act, const float* row_scale, int rows_per_scale,
                                   void* stream) {
  if (row_scale && rows_per_scale <= 0) return gkg_fail(GKG_ERR_SHAPE, "gkg_affine_act_dual: rows_per_scale must be positive");
  if (!y || !a || !c || !out_f32 || !out_bf16) return gkg_fail(GKG_ERR_NULL, "gkg_affine_act_dual: null pointer");
  if (R <= 0 || bad_c(C) || (act != 0 && act != 1)) return gkg_fail(GKG_ERR_SHAPE, "gkg_affine_act_dual: bad sizes");
  const size_t total4 = (size_t)R * (C >> 2);
  const int blocks = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (act == 1) hipLaunchKernelGGL((affine_act_kernel<1, float>), dim3(blocks, 1), dim3(256), 0, st, y, a, c, res, out_f32, total4, C, C, (size_t)0, row_scale, rows_per_scale, (uint16_t*)out_bf16, BnDerive{});
  else hipLaunchKernelGGL((affine_act_kernel<0, float>), dim3(blocks, 1), dim3(256), 0, st, y, a, c, res, out_f32, total4, C, C, (size_t)0, row_scale, rows_per_scale, (uint16_t*)out_bf16, BnDerive{});
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "affine_act_kernel (dual)");
}

extern "C" int gkg_bn_bwd(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                          const float* invstd, float* dy, float* dgamma, float* dbeta, int R, int C, int nb, int ldg,
                          size_t dout_bstride, int act, void* workspace, size_t workspace_bytes, void* stream) {
  if (!dout || !y || !a || !c || !mean || !invstd || !dy || !dgamma || !dbeta || !workspace)
    return gkg_fail(GKG_ERR_NULL, "gkg_bn_bwd: null pointer");
  if (R <= 0 || bad_c(C) || nb <= 0 || nb > 64 || ldg < C || (ldg & 3) || (dout_bstride & 3) || (act != 0 && act != 1))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_bwd: bad sizes");
  int rpb;
  const int nblk = stats_blocks(R, C, nb, &rpb);
  if (workspace_bytes < (size_t)nb * (nblk + 1) * 2 * C * sizeof(float)) return gkg_fail(GKG_ERR_WORKSPACE, "gkg_bn_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* sums = part + (size_t)nb * nblk * 2 * C;
  // With an activation both passes evaluate dz = dout*act'(z) themselves: since GELU' costs ~20 vector instructions (A&S
  // erf sharing its exponential with the Gaussian) recomputing it in the memory-bound apply pass is cheaper than parking
  // dz (a full R x C store in the statistics pass); rounds 1-2 stored it, erff being twice as expensive.
  if (act == 1) hipLaunchKernelGGL((bn_bwd_stats_kernel<1>), dim3(nblk, stats_tiles(C), nb), dim3(256), 0, st, dout, y, a, c, mean, invstd, part, R, C, rpb, ldg, dout_bstride, (float*)nullptr);
  else hipLaunchKernelGGL((bn_bwd_stats_kernel<0>), dim3(nblk, stats_tiles(C), nb), dim3(256), 0, st, dout, y, a, c, mean, invstd, part, R, C, rpb, ldg, dout_bstride, (float*)nullptr);
  // dbeta[q] = sum dz ; dgamma[q] = sum dz*yhat
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * C + 31) / 32, nb), dim3(256), 0, st, part, sums, nblk, 2 * C, dbeta, dgamma);
  const size_t total4 = (size_t)R * (C >> 2);
  const int blocks = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
  if (act == 1) hipLaunchKernelGGL((bn_bwd_apply_kernel<1>), dim3(blocks, nb), dim3(256), 0, st, dout, y, a, c, mean, invstd, sums, dy, total4, C, R, ldg, dout_bstride, (const float*)nullptr);
  else hipLaunchKernelGGL((bn_bwd_apply_kernel<0>), dim3(blocks, nb), dim3(256), 0, st, dout, y, a, c, mean, invstd, sums, dy, total4, C, R, ldg, dout_bstride, (const float*)nullptr);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "bn_bwd");
}

// gkg_bn_bwd in TWO launches: the statistics pass adds its per-workgroup column sums to `sums` (fp64, 2 * nb * C doubles,
// ZERO on entry) with atomics — no partial rows, no second-stage reduction kernel — and the apply pass reads them, emits
// dgamma / dbeta and clears `zero_buf` (zero_doubles doubles; may be NULL / 0).  The caller alternates between two scratch
// buffers and hands each call the region the PREVIOUS call used as `zero_buf`: every buffer is clean again before its next
// use without a memset launch or a last-arriver ticket.  fp64 atomics: the sums are run-dependent in their last fp64 bits
// (deterministic callers keep gkg_bn_bwd).
static int bn_bwd_atomic_impl(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                              const float* invstd, float* dy, float* dgamma, float* dbeta, int R, int C, int nb, int ldg,
                              size_t dout_bstride, int act, double* sums, double* zero_buf, size_t zero_doubles, void* stream,
                              bool stats_pass, const float* row_scale = nullptr, int rows_per_scale = 1);

extern "C" int gkg_bn_bwd_atomic(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                                 const float* invstd, float* dy, float* dgamma, float* dbeta, int R, int C, int nb, int ldg,
                                 size_t dout_bstride, int act, double* sums, double* zero_buf, size_t zero_doubles, void* stream) {
  if (!dout || !y || !a || !c || !mean || !invstd || !dy || !dgamma || !dbeta || !sums)
    return gkg_fail(GKG_ERR_NULL, "gkg_bn_bwd_atomic: null pointer");
  if (R <= 0 || bad_c(C) || nb <= 0 || nb > 64 || ldg < C || (ldg & 3) || (dout_bstride & 3) || (act != 0 && act != 1) ||
      (zero_doubles && !zero_buf))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_bwd_atomic: bad sizes");
  return bn_bwd_atomic_impl(dout, y, a, c, mean, invstd, dy, dgamma, dbeta, R, C, nb, ldg, dout_bstride, act, sums, zero_buf,
                            zero_doubles, stream, true);
}

// gkg_bn_bwd_atomic for a branch whose OUTPUT was scaled per image (DropPath, torch_vertex.py:332,355,402): the incoming
// gradient is multiplied by row_scale[row / rows_per_scale] inside both passes instead of by a separate elementwise launch.
extern "C" int gkg_bn_bwd_atomic_scaled(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                                        const float* invstd, float* dy, float* dgamma, float* dbeta, int R, int C, int nb,
                                        int ldg, size_t dout_bstride, int act, double* sums, double* zero_buf,
                                        size_t zero_doubles, const float* row_scale, int rows_per_scale, void* stream) {
  if (!dout || !y || !a || !c || !mean || !invstd || !dy || !dgamma || !dbeta || !sums || !row_scale)
    return gkg_fail(GKG_ERR_NULL, "gkg_bn_bwd_atomic_scaled: null pointer");
  if (R <= 0 || bad_c(C) || nb <= 0 || nb > 64 || ldg < C || (ldg & 3) || (dout_bstride & 3) || (act != 0 && act != 1) ||
      (zero_doubles && !zero_buf) || rows_per_scale <= 0)
    return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_bwd_atomic_scaled: bad sizes");
  return bn_bwd_atomic_impl(dout, y, a, c, mean, invstd, dy, dgamma, dbeta, R, C, nb, ldg, dout_bstride, act, sums, zero_buf,
                            zero_doubles, stream, true, row_scale, rows_per_scale);
}

// The same backward with the statistics ALREADY in `sums` (accumulated by the epilogue of the GEMM that produced dout:
// gkg_linear_dgrad_x6_bnbwd): only the apply pass runs.
extern "C" int gkg_bn_bwd_apply_from_sums(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                                          const float* invstd, float* dy, float* dgamma, float* dbeta, int R, int C, int nb,
                                          int ldg, size_t dout_bstride, int act, const double* sums, double* zero_buf,
                                          size_t zero_doubles, void* stream) {
  if (!dout || !y || !a || !c || !mean || !invstd || !dy || !dgamma || !dbeta || !sums)
    return gkg_fail(GKG_ERR_NULL, "gkg_bn_bwd_apply_from_sums: null pointer");
  if (R <= 0 || bad_c(C) || nb <= 0 || nb > 64 || ldg < C || (ldg & 3) || (dout_bstride & 3) || (act != 0 && act != 1) ||
      (zero_doubles && !zero_buf))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_bwd_apply_from_sums: bad sizes");
  return bn_bwd_atomic_impl(dout, y, a, c, mean, invstd, dy, dgamma, dbeta, R, C, nb, ldg, dout_bstride, act,
                            const_cast<double*>(sums), zero_buf, zero_doubles, stream, false);
}


static int bn_bwd_atomic_impl(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                              const float* invstd, float* dy, float* dgamma, float* dbeta, int R, int C, int nb, int ldg,
                              size_t dout_bstride, int act, double* sums, double* zero_buf, size_t zero_doubles, void* stream,
                              bool stats_pass, const float* row_scale, int rows_per_scale) {
  // (A one-launch form with a grid barrier was built and measured in round 5 — a barrier of a few hundred workgroups costs
  // more than the kernel boundary it replaces, EXPERIMENTS.md — and removed in round 6.)
  int rpb;
  const int nblk = stats_blocks(R, C, nb, &rpb);
  hipStream_t st = (hipStream_t)stream;
  if (stats_pass) {
    if (act == 1) hipLaunchKernelGGL((bn_bwd_stats_kernel<1>), dim3(nblk, stats_tiles(C), nb), dim3(256), 0, st, dout, y, a, c, mean, invstd, (float*)nullptr, R, C, rpb, ldg, dout_bstride, (float*)nullptr, sums, row_scale, rows_per_scale);
    else hipLaunchKernelGGL((bn_bwd_stats_kernel<0>), dim3(nblk, stats_tiles(C), nb), dim3(256), 0, st, dout, y, a, c, mean, invstd, (float*)nullptr, R, C, rpb, ldg, dout_bstride, (float*)nullptr, sums, row_scale, rows_per_scale);
  }
  const size_t total4 = (size_t)R * (C >> 2);
  const int blocks = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
  if (act == 1) hipLaunchKernelGGL((bn_bwd_apply_d_kernel<1>), dim3(blocks, nb), dim3(256), 0, st, dout, y, a, c, mean, invstd, sums, dy, total4, C, R, ldg, dout_bstride, dgamma, dbeta, zero_buf, zero_doubles, row_scale, rows_per_scale);
  else hipLaunchKernelGGL((bn_bwd_apply_d_kernel<0>), dim3(blocks, nb), dim3(256), 0, st, dout, y, a, c, mean, invstd, sums, dy, total4, C, R, ldg, dout_bstride, dgamma, dbeta, zero_buf, zero_doubles, row_scale, rows_per_scale);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "bn_bwd_atomic");
}

// Train-mode BN-apply straight from the projection's fp64 column sums (gkg_linear_bn_fwd / _x6 with train == 2): replaces
// the finalize launch + gkg_affine_act / gkg_tm_affine_to_nchw.  `sums` [nb][2][C] (this call's scratch buffer, read only),
// `zero_buf` / `zero_doubles`: the OTHER scratch buffer's dirty region, cleared here (see gkg_bn_bwd_atomic).  Writes the
// saved a / c / mean / invstd [nb][C], updates running_mean / running_var (conv `bias` folded in) and num_batches_tracked.
//   nchw_B == 0: out (nb, R, ...) token-major like gkg_affine_act (fp32 out; act, res, row_scale as there)
//   nchw_B  > 0: out / res are (nchw_B, C, R / nchw_B) channel-major like gkg_tm_affine_to_nchw (nb == 1, act == 0,
//                row_scale = one factor per image)
extern "C" int gkg_bn_apply_train(const float* y, const double* sums, const float* gamma, const float* beta, const float* bias,
                                  float* running_mean, float* running_var, long long* num_batches_tracked, float* a, float* c,
                                  float* mean, float* invstd, const float* res, float* out, int R, int C, int nb, int ldo,
                                  size_t out_bstride, int ochunk, int act, int nchw_B, const float* row_scale, int rows_per_scale,
                                  float momentum, float eps, double* zero_buf, size_t zero_doubles, void* stream) {
  if (!y || !sums || !gamma || !beta || !a || !c || !mean || !invstd || !out)
    return gkg_fail(GKG_ERR_NULL, "gkg_bn_apply_train: null pointer");
  if ((running_mean == nullptr) != (running_var == nullptr)) return gkg_fail(GKG_ERR_NULL, "gkg_bn_apply_train: running stats come in pairs");
  if (R <= 0 || bad_c(C) || nb <= 0 || nb > 64 || (act != 0 && act != 1) || (zero_doubles && !zero_buf) || nchw_B < 0)
    return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_apply_train: bad sizes");
  BnDerive d{sums, gamma, beta, bias, running_mean, running_var, num_batches_tracked, a, c, mean, invstd, R, momentum, eps,
             zero_buf, zero_doubles};
  hipStream_t st = (hipStream_t)stream;
  if (nchw_B > 0) {
    if (nb != 1 || act != 0 || R % nchw_B) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_apply_train: channel-major output needs nb == 1, act == 0, R % B == 0");
    const int N = R / nchw_B;
    dim3 grid((N + 31) / 32, (C + 31) / 32, nchw_B);
    hipLaunchKernelGGL(tm_affine_to_nchw_kernel, grid, dim3(256), 0, st, y, (const float*)nullptr, (const float*)nullptr, res, out, C, N,
                       row_scale, d);
  } else {
    if (row_scale && rows_per_scale <= 0) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_apply_train: rows_per_scale must be positive");
    if (ldo < C || (ldo & 3) || (out_bstride & 3)) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_apply_train: bad output pitch");
    if (ochunk < 0 || (ochunk & 3) || (ochunk > 0 && (C % ochunk || ldo < 2 * C)))
      return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_apply_train: ochunk must be a multiple of 4 dividing C, with ldo >= 2 C");
    const size_t total4 = (size_t)R * (C >> 2);
    const int blocks = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
    const dim3 grid(blocks, nb);
    const size_t lds = (size_t)2 * C * sizeof(float);
    if (act == 1) hipLaunchKernelGGL((affine_act_kernel<1, float>), grid, dim3(256), lds, st, y, (const float*)nullptr, (const float*)nullptr, res, out, total4, C, ldo, out_bstride, row_scale, rows_per_scale, (uint16_t*)nullptr, d, ochunk);
    else hipLaunchKernelGGL((affine_act_kernel<0, float>), grid, dim3(256), lds, st, y, (const float*)nullptr, (const float*)nullptr, res, out, total4, C, ldo, out_bstride, row_scale, rows_per_scale, (uint16_t*)nullptr, d, ochunk);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "bn_apply_train");
}

// gkg_bn_apply_train's channel-major form with the residual given token-major and the result written in both layouts
// (gkg_tm_affine_to_nchw_dual with the coefficients derived from the projection's column sums).
extern "C" int gkg_bn_apply_train_dual(const float* y, const double* sums, const float* gamma, const float* beta, const float* bias,
                                       float* running_mean, float* running_var, long long* num_batches_tracked, float* a,
                                       float* c, float* mean, float* invstd, const float* res_tm, float* out, float* out_tm,
                                       int B, int C, int N, float momentum, float eps, double* zero_buf, size_t zero_doubles,
                                       void* stream) {
  if (!y || !sums || !gamma || !beta || !a || !c || !mean || !invstd || !out || !res_tm || !out_tm)
    return gkg_fail(GKG_ERR_NULL, "gkg_bn_apply_train_dual: null pointer");
  if ((running_mean == nullptr) != (running_var == nullptr)) return gkg_fail(GKG_ERR_NULL, "gkg_bn_apply_train_dual: running stats come in pairs");
  if (B <= 0 || B > 65535 || N <= 0 || bad_c(C) || (zero_doubles && !zero_buf)) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_apply_train_dual: bad sizes");
  BnDerive d{sums, gamma, beta, bias, running_mean, running_var, num_batches_tracked, a, c, mean, invstd, B * N, momentum, eps,
             zero_buf, zero_doubles};
  dim3 grid((N + 31) / 32, (C + 31) / 32, B);
  hipLaunchKernelGGL(tm_affine_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, out, C, N, (const float*)nullptr, d, res_tm, out_tm);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "bn_apply_train (dual)");
}

// ------------------------------------------------------------------------------------------ cross-rank (SyncBN) halves
// The same passes split where the reference's SyncBatchNorm exchanges statistics: the caller all-reduces `sums`
// (and the row count) over the ranks between the two calls.
extern "C" int gkg_bn_stats_sums(const float* y, float* sums, int R, int C, int nb, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (!y || !sums || !workspace) return gkg_fail(GKG_ERR_NULL, "gkg_bn_stats_sums: null pointer");
  if (R <= 0 || bad_c(C) || nb <= 0 || nb > 64) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_stats_sums: bad sizes");
  int rpb;
  const int nblk = stats_blocks(R, C, nb, &rpb);
  if (workspace_bytes < (size_t)nb * (nblk + 1) * 2 * C * sizeof(float))
    return gkg_fail(GKG_ERR_WORKSPACE, "gkg_bn_stats_sums: workspace too small (gkg_bn_workspace_bytes)");
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  hipLaunchKernelGGL(col_stats_kernel<false>, dim3(nblk, stats_tiles(C), nb), dim3(256), 0, st, y, part, R, C, rpb);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * C + 31) / 32, nb), dim3(256), 0, st, part, sums, nblk, 2 * C,
                     (float*)nullptr, (float*)nullptr);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "bn_stats_sums");
}

extern "C" int gkg_bn_finalize(const float* sums, const float* count, const float* gamma, const float* beta,
                               const float* bias, float* running_mean, float* running_var, float* a, float* c,
                               float* mean, float* invstd, int C, int nb, float momentum, float eps,
                               long long* num_batches_tracked, void* stream) {
  if (!sums || !count || !gamma || !beta || !a || !c || !mean || !invstd) return gkg_fail(GKG_ERR_NULL, "gkg_bn_finalize: null pointer");
  if (bad_c(C) || nb <= 0 || nb > 64) return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_finalize: bad sizes");
  if ((running_mean == nullptr) != (running_var == nullptr)) return gkg_fail(GKG_ERR_NULL, "gkg_bn_finalize: running stats come in pairs");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256, nb), dim3(256), 0, (hipStream_t)stream, sums, gamma, beta,
                     bias, running_mean, running_var, a, c, mean, invstd, count, C, momentum, eps, num_batches_tracked);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "bn_finalize_kernel");
}

extern "C" int gkg_bn_bwd_sums(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                               const float* invstd, float* dy, float* sums, float* dgamma, float* dbeta, int R, int C,
                               int nb, int ldg, size_t dout_bstride, int act, void* workspace, size_t workspace_bytes,
                               void* stream) {
  if (!dout || !y || !a || !c || !mean || !invstd || !dy || !sums || !dgamma || !dbeta || !workspace)
    return gkg_fail(GKG_ERR_NULL, "gkg_bn_bwd_sums: null pointer");
  if (R <= 0 || bad_c(C) || nb <= 0 || nb > 64 || ldg < C || (ldg & 3) || (dout_bstride & 3) || (act != 0 && act != 1))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_bwd_sums: bad sizes");
  int rpb;
  const int nblk = stats_blocks(R, C, nb, &rpb);
  if (workspace_bytes < (size_t)nb * (nblk + 1) * 2 * C * sizeof(float)) return gkg_fail(GKG_ERR_WORKSPACE, "gkg_bn_bwd_sums: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (act == 1) hipLaunchKernelGGL((bn_bwd_stats_kernel<1>), dim3(nblk, stats_tiles(C), nb), dim3(256), 0, st, dout, y, a, c, mean, invstd, part, R, C, rpb, ldg, dout_bstride, dy);
  else hipLaunchKernelGGL((bn_bwd_stats_kernel<0>), dim3(nblk, stats_tiles(C), nb), dim3(256), 0, st, dout, y, a, c, mean, invstd, part, R, C, rpb, ldg, dout_bstride, (float*)nullptr);
  // the LOCAL sums are the parameter gradients (the data-parallel gradient exchange averages them later)
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * C + 31) / 32, nb), dim3(256), 0, st, part, sums, nblk, 2 * C, dbeta, dgamma);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "bn_bwd_sums");
}

extern "C" int gkg_bn_bwd_apply(const float* dout, const float* y, const float* a, const float* c, const float* mean,
                                const float* invstd, const float* sums, const float* count, float* dy, int R, int C,
                                int nb, int ldg, size_t dout_bstride, int act, void* stream) {
  if (!dout || !y || !a || !c || !mean || !invstd || !sums || !count || !dy) return gkg_fail(GKG_ERR_NULL, "gkg_bn_bwd_apply: null pointer");
  if (R <= 0 || bad_c(C) || nb <= 0 || nb > 64 || ldg < C || (ldg & 3) || (dout_bstride & 3) || (act != 0 && act != 1))
    return gkg_fail(GKG_ERR_SHAPE, "gkg_bn_bwd_apply: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const size_t total4 = (size_t)R * (C >> 2);
  const int blocks = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
  // act == 1: gkg_bn_bwd_sums parked dz = dout*act'(z) in dy; apply in place
  if (act == 1) hipLaunchKernelGGL((bn_bwd_apply_kernel<0>), dim3(blocks, nb), dim3(256), 0, st, dy, y, a, c, mean, invstd, sums, dy, total4, C, R, C, (size_t)R * C, count);
  else hipLaunchKernelGGL((bn_bwd_apply_kernel<0>), dim3(blocks, nb), dim3(256), 0, st, dout, y, a, c, mean, invstd, sums, dy, total4, C, R, ldg, dout_bstride, count);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gkg_fail_hip(e, "bn_bwd_apply_kernel");
}
