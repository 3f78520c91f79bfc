// Micro-benchmark: can one SIMD overlap fp32 MFMA (v_mfma_f32_32x32x2_f32) of one wave with fp32 VALU of another?
// 512-thread workgroups: waves 0-3 and 4-7 share SIMDs 0-3.  mode: 0 = all MFMA, 1 = all VALU, 2 = half/half,
// 3 = one wave per SIMD MFMA only, 4 = one wave per SIMD VALU only, 5 = each wave alternates MFMA and VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int w = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  float a = lane * 0.001f, b = 1.0f + lane * 1e-4f;
  f32x16 acc0 = {0}, acc1 = {0};
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  bool do_m, do_v;
  if (mode == 0) { do_m = true; do_v = false; }
  else if (mode == 1) { do_m = false; do_v = true; }
  else if (mode == 2) { do_m = w < 4; do_v = w >= 4; }
  else if (mode == 3) { do_m = w < 4; do_v = false; }
  else if (mode == 4) { do_m = false; do_v = w < 4; }
  else { do_m = true; do_v = true; }
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
      }
    }
    if (do_v) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {   // 32 x 8 = 256 dependent-chain-free fmas
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], b, a);
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* out; hipMalloc(&out, 256 * 4 * 512 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int mode = 0; mode < 6; ++mode) {
    k<<<256, 512>>>(out, 10, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<256, 512>>>(out, iters, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per iteration per wave: 16 MFMA (= 1024 pipe cycles), 256 VALU
    printf("mode %d: %.3f ms  -> %.1f ns / iteration\n", mode, ms, ms * 1e6 / iters);
  }
  return 0;
}
