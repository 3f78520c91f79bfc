/*
 * TEST INFRASTRUCTURE — CPU oracle (plain C, scalar) for libgkg_hip.so.
 *
 * Bit-for-bit restatement of the arithmetic contract in include/gkg_hip.h, i.e. of the reference's
 * algorithm with a *defined* fp32 evaluation order:
 *   - L2 normalisation over the group's channels, eps 1e-12   (reference torch_edge.py:167-168,173)
 *   - dist = (|x|^2 + (-2 x.y)) + |y|^2, then += relative_pos  (torch_edge.py:17-20 / 47-51, :82 / :103)
 *   - k*d smallest per query, sorted ascending, every d-th kept (torch_edge.py:83,104,146-148)
 *   - m = max_j(src[idx_j] - x), first-j argmax                (torch_vertex.py:49-54, torch_nn.py:84-105)
 *   - backward scatter of the max                               (autograd of the above)
 * The only freedom the reference leaves (BLAS accumulation order of the dot product, topk tie order)
 * is fixed here as: ordered fmaf chain over channels 0..c-1; ties -> smaller key index first.
 * This oracle is itself checked against the reference's golden vectors (tests/test_c_oracle.py)
 * with the near-tie protocol, and the HIP kernels must match THIS file exactly (indices, m, argmax).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp; fmaf must be a true fused multiply-add).
 * OpenMP only spreads independent queries / (problem, channel) rows over threads: every output element is
 * still computed by one thread in the scalar order above, so results do not depend on the thread count.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_NORMALIZE 1u

/* th (c,T) <- normalised copy of t (c,T); sq (T) <- sum th^2.  Both via ordered fmaf chains. */
static void token_prep(const float* t, float* th, float* sq, int c, int T, unsigned flags) {
  for (int n = 0; n < T; ++n) {
    float den = 1.0f;
    if (flags & ORACLE_NORMALIZE) {
      float s = 0.0f;
      for (int ch = 0; ch < c; ++ch) s = fmaf(t[(size_t)ch * T + n], t[(size_t)ch * T + n], s);
      den = fmaxf(sqrtf(s), 1e-12f);
    }
    float q = 0.0f;
    for (int ch = 0; ch < c; ++ch) {
      float v = t[(size_t)ch * T + n];
      if (flags & ORACLE_NORMALIZE) v = v / den;
      th[(size_t)ch * T + n] = v;
      q = fmaf(v, v, q);
    }
    sq[n] = q;
  }
}

/* Full distance row for query n of one (bg) problem. */
static void dist_row(const float* xh, const float* yh, const float* sqx, const float* sqy,
                     const float* relpos, int c, int N, int M, int n, float* out) {
  /* channel-outer loop: out[m] carries key m's ordered fmaf chain over ch = 0..c-1 (same arithmetic as a
     per-key inner loop, contiguous reads of yh) */
  for (int m = 0; m < M; ++m) out[m] = 0.0f;
  for (int ch = 0; ch < c; ++ch) {
    const float xv = xh[(size_t)ch * N + n];
    const float* yr = yh + (size_t)ch * M;
    for (int m = 0; m < M; ++m) out[m] = fmaf(yr[m], xv, out[m]);
  }
  for (int m = 0; m < M; ++m) {
    float d = (sqx[n] + (-2.0f * out[m])) + sqy[m];
    if (relpos) d = d + relpos[(size_t)n * M + m];
    out[m] = d;
  }
}

/* Returns 0 on success, negative on bad arguments.  dist_out (BG,N,M) optional. */
int oracle_knn_fwd(const float* x, const float* y, const float* relpos, int64_t* nn_idx, int64_t* center,
                   int BG, int c, int N, int M, int k, int dilation, unsigned flags, float* dist_out) {
  if (!x || !nn_idx) return -1;
  if (BG <= 0 || c <= 0 || N <= 0 || M <= 0 || k <= 0 || dilation <= 0) return -2;
  if (!y && M != N) return -2;
  const int kd = k * dilation;
  if (kd > M) return -2;
  float* xh = (float*)malloc(sizeof(float) * (size_t)c * N);
  float* yh = y ? (float*)malloc(sizeof(float) * (size_t)c * M) : xh;
  float* sqx = (float*)malloc(sizeof(float) * N);
  float* sqy = y ? (float*)malloc(sizeof(float) * M) : sqx;
  for (int bg = 0; bg < BG; ++bg) {
    token_prep(x + (size_t)bg * c * N, xh, sqx, c, N, flags);
    if (y) token_prep(y + (size_t)bg * c * M, yh, sqy, c, M, flags);
    /* queries are independent: OpenMP over n with thread-private scratch (results do not depend on the thread count) */
#pragma omp parallel
    {
      float* row = (float*)malloc(sizeof(float) * M);
      float* bv = (float*)malloc(sizeof(float) * kd);
      int* bi = (int*)malloc(sizeof(int) * kd);
#pragma omp for schedule(static)
      for (int n = 0; n < N; ++n) {
        dist_row(xh, yh, sqx, sqy, relpos, c, N, M, n, row);
        if (dist_out) memcpy(dist_out + ((size_t)bg * N + n) * M, row, sizeof(float) * M);
        /* stable insertion of keys in increasing m: strict '<' keeps the smaller index first on ties */
        int cnt = 0;
        for (int m = 0; m < M; ++m) {
          float d = row[m];
          if (cnt == kd && !(d < bv[kd - 1])) continue;
          int p = cnt < kd ? cnt : kd - 1;
          while (p > 0 && d < bv[p - 1]) { bv[p] = bv[p - 1]; bi[p] = bi[p - 1]; --p; }
          bv[p] = d; bi[p] = m;
          if (cnt < kd) ++cnt;
        }
        for (int j = 0; j < k; ++j) {
          nn_idx[((size_t)bg * N + n) * k + j] = bi[j * dilation];
          if (center) center[((size_t)bg * N + n) * k + j] = n;
        }
      }
      free(bi); free(bv); free(row);
    }
  }
  if (y) { free(sqy); free(yh); }
  free(sqx); free(xh);
  return 0;
}

int oracle_mr_fwd(const float* x, const float* src, const int64_t* nn_idx, float* m_out, uint8_t* argmax,
                  int BG, int c, int N, int M, int k) {
  if (!x || !nn_idx || !m_out) return -1;
  if (!src) { src = x; if (M != N) return -2; }
#pragma omp parallel for collapse(2) schedule(static)
  for (int bg = 0; bg < BG; ++bg)
    for (int ch = 0; ch < c; ++ch) {
      const float* xr = x + ((size_t)bg * c + ch) * N;
      const float* sr = src + ((size_t)bg * c + ch) * M;
      for (int n = 0; n < N; ++n) {
        const int64_t* id = nn_idx + ((size_t)bg * N + n) * k;
        float best = sr[id[0]] - xr[n];
        int arg = 0;
        for (int j = 1; j < k; ++j) {
          float v = sr[id[j]] - xr[n];
          if (v > best || (v != v && best == best)) { best = v; arg = j; }   /* torch.max: NaN propagates, first NaN wins */
        }
        m_out[((size_t)bg * c + ch) * N + n] = best;
        if (argmax) argmax[((size_t)bg * c + ch) * N + n] = (uint8_t)arg;
      }
    }
  return 0;
}

int oracle_mr_bwd(const float* g, const int64_t* nn_idx, const uint8_t* argmax, float* gx, float* gsrc,
                  int BG, int c, int N, int M, int k) {
  if (!g || !nn_idx || !argmax || !gx) return -1;
  if (!gsrc && M != N) return -2;
#pragma omp parallel for collapse(2) schedule(static)
  for (int bg = 0; bg < BG; ++bg)
    for (int ch = 0; ch < c; ++ch) {
      const size_t ro = ((size_t)bg * c + ch) * N;
      float* gxr = gx + ro;
      float* gsr = gsrc ? gsrc + ((size_t)bg * c + ch) * M : gxr;
      for (int n = 0; n < N; ++n) gxr[n] = -g[ro + n];
      if (gsrc) for (int m = 0; m < M; ++m) gsr[m] = 0.0f;
      for (int n = 0; n < N; ++n) {
        const int64_t j = nn_idx[((size_t)bg * N + n) * k + argmax[ro + n]];
        gsr[j] += g[ro + n];
      }
    }
  return 0;
}
