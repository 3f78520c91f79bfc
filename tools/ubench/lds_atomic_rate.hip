// LDS atomic throughput on gfx950: ds_add_f32 vs ds_add_u32 vs ds_add_u64 vs ds_add_f64 (no-return forms), addresses spread over an image
// like the aggregation backward's (M rows x CW floats).  hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int KIND>
__global__ __launch_bounds__(256) void k(const int* __restrict__ idx, float* out, int rounds, int n_idx, int words) {
  extern __shared__ unsigned char lds[];
  float* f = reinterpret_cast<float*>(lds);
  unsigned* u = reinterpret_cast<unsigned*>(lds);
  unsigned long long* q = reinterpret_cast<unsigned long long*>(lds);
  for (int i = threadIdx.x; i < words; i += 256) u[i] = 0;
  __syncthreads();
  int p = (blockIdx.x * 256 + threadIdx.x) % n_idx;
  float v = 1.0f + threadIdx.x * 1e-3f;
  for (int r = 0; r < rounds; ++r) {
    const int a = idx[p];
    p += 256; if (p >= n_idx) p -= n_idx;
    if (KIND == 0) atomicAdd(f + a, v);
    if (KIND == 1) atomicAdd(u + a, (unsigned)(v * 1024.f));
    if (KIND == 2) atomicAdd(q + (a >> 1), (unsigned long long)(v * 1024.f));
    if (KIND == 3) f[a] += v;                       // plain read-modify-write (racy: rate reference only)
    if (KIND == 4) unsafeAtomicAdd(reinterpret_cast<double*>(lds) + (a >> 1), (double)v);   // ds_add_f64
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = f[blockIdx.x % words];
}

int main() {
  const int words = 324 * 16, n_idx = 1 << 20, rounds = 2048, blocks = 256 * 4;
  int* h = (int*)malloc(n_idx * sizeof(int));
  for (int mode = 0; mode < 2; ++mode) {                 // 0: random addresses, 1: runs of 16 equal rows (adjacent queries share a key)
    unsigned s = 12345;
    int row = 0;
    for (int i = 0; i < n_idx; ++i) {
      s = s * 1664525u + 1013904223u;
      if (mode == 0 || (i & 15) == 0) row = (s >> 8) % 324;
      h[i] = row * 16 + ((i >> 4) & 3) * 4 + (s >> 28) % 4;
    }
    int* d; float* o;
    hipMalloc(&d, n_idx * sizeof(int)); hipMalloc(&o, blocks * sizeof(float));
    hipMemcpy(d, h, n_idx * sizeof(int), hipMemcpyHostToDevice);
    const char* names[5] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain rmw (racy)", "ds_add_f64"};
    for (int kind = 0; kind < 5; ++kind) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), words * 4, 0, d, o, rounds, n_idx, words);
        if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), words * 4, 0, d, o, rounds, n_idx, words);
        if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), words * 4, 0, d, o, rounds, n_idx, words);
        if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), words * 4, 0, d, o, rounds, n_idx, words);
        if (kind == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), words * 4, 0, d, o, rounds, n_idx, words);
        hipEventRecord(b); hipEventSynchronize(b);
      }
      float ms; hipEventElapsedTime(&ms, a, b);
      const double lane_ops = (double)blocks * 256 * rounds;
      // 256 CUs; cycles per wave instruction per CU at 2.1 GHz
      const double wave_instr_per_cu = lane_ops / 64 / 256;
      printf("%s addresses, %-18s %8.3f ms  %7.2f G lane-atomics/s  ~%6.1f cycles per wave instruction per CU (2.1 GHz)\n",
             mode ? "clustered" : "random   ", names[kind], ms, lane_ops / ms / 1e6, ms * 1e-3 * 2.1e9 / wave_instr_per_cu);
    }
    hipFree(d); hipFree(o);
  }
  return 0;
}
