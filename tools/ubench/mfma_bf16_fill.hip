// Micro-benchmark: which VALU / LDS instructions hide in the shadow of v_mfma_f32_32x32x16_bf16 (one wave per SIMD)?
// Each variant runs 12 x { MFMA ; fillers } per iteration from one asm block, so the order is exactly as written.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF "v_mfma_f32_32x32x16_bf16 %[c0], %[a], %[b], %[c0]\n\t"
#define MF2 "v_mfma_f32_32x32x16_bf16 %[c1], %[a], %[b], %[c1]\n\t"
#define ADD4 "v_add_f32 %[v0], %[v0], %[c]\n\tv_add_f32 %[v1], %[v1], %[c]\n\tv_add_f32 %[v2], %[v2], %[c]\n\tv_add_f32 %[v3], %[v3], %[c]\n\t"
#define ADD2 "v_add_f32 %[v0], %[v0], %[c]\n\tv_add_f32 %[v1], %[v1], %[c]\n\t"
// the split chain on (%4,%5) -> packed in %6, residuals back in %4,%5
#define CHAIN "v_cvt_pk_bf16_f32 %[v2], %[v0], %[v1]\n\tv_lshlrev_b32 %[v3], 16, %[v2]\n\tv_and_b32 %[t], 0xffff0000, %[v2]\n\tv_sub_f32 %[v0], %[v0], %[v3]\n\tv_sub_f32 %[v1], %[v1], %[t]\n\t"
//#define CHAINPK "v_cvt_pk_bf16_f32 %[v2], %[v0], %[v1]\n\tv_lshlrev_b32 %[v3], 16, %[v2]\n\tv_and_b32 %[t], 0xffff0000, %[v2]\n\tv_pk_add_f32 %[p], %[p], %[q] neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define DSR "ds_read_b128 %[d], %[addr]\n\t"
#define R6(x) x x x x x x
#define R12(x) R6(x) R6(x)

template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float4 sm[1024];
  const int lane = threadIdx.x & 63;
  sm[threadIdx.x] = float4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x16 acc0 = {0}, acc1 = {0};
  f32x4 a = {lane * 0.001f, 1.f, 2.f, 3.f}, b = {1.0f, lane * 1e-4f, 0.5f, 0.25f};
  float v0 = lane, v1 = lane + 1.f, v2 = lane + 2.f, v3 = lane + 3.f, c = 1e-3f, t = 0.f;
  f32x4 d = {0, 0, 0, 0};
  const unsigned addr = (threadIdx.x & 255) * 16;
  for (int it = 0; it < iters; ++it) {
    if (V == 0) asm volatile(R12(MF) : [c0] "+v"(acc0), [c1] "+v"(acc1), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [t] "+v"(t), [d] "+v"(d) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [addr] "v"(addr));
    if (V == 1) asm volatile(R12(MF ADD4) : [c0] "+v"(acc0), [c1] "+v"(acc1), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [t] "+v"(t), [d] "+v"(d) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [addr] "v"(addr));
    if (V == 2) asm volatile(R12(MF ADD4 ADD2) : [c0] "+v"(acc0), [c1] "+v"(acc1), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [t] "+v"(t), [d] "+v"(d) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [addr] "v"(addr));
    if (V == 3) asm volatile(R12(MF CHAIN) : [c0] "+v"(acc0), [c1] "+v"(acc1), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [t] "+v"(t), [d] "+v"(d) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [addr] "v"(addr));
    if (V == 4) asm volatile(R12(MF ADD4 ADD4) : [c0] "+v"(acc0), [c1] "+v"(acc1), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [t] "+v"(t), [d] "+v"(d) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [addr] "v"(addr));
    if (V == 5) asm volatile(R12(MF DSR) "s_waitcnt lgkmcnt(0)\n\t" : [c0] "+v"(acc0), [c1] "+v"(acc1), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [t] "+v"(t), [d] "+v"(d) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [addr] "v"(addr));
    if (V == 6) asm volatile(R12(MF CHAIN DSR) "s_waitcnt lgkmcnt(0)\n\t" : [c0] "+v"(acc0), [c1] "+v"(acc1), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [t] "+v"(t), [d] "+v"(d) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [addr] "v"(addr));
    if (V == 7) asm volatile(R6(MF MF2) : [c0] "+v"(acc0), [c1] "+v"(acc1), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [t] "+v"(t), [d] "+v"(d) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [addr] "v"(addr));
    if (V == 8) asm volatile(R12(CHAIN) : [c0] "+v"(acc0), [c1] "+v"(acc1), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [t] "+v"(t), [d] "+v"(d) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [addr] "v"(addr));
    if (V == 9) asm volatile(R12(MF "s_nop 7\n\ts_nop 7\n\t") : [c0] "+v"(acc0), [c1] "+v"(acc1), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [t] "+v"(t), [d] "+v"(d) : [a] "v"(a), [b] "v"(b), [c] "v"(c), [addr] "v"(addr));
  }
  float s = v0 + v1 + v2 + v3 + d.x + t;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
void run(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  k<V><<<256, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<V><<<256, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %8.1f ns / 12 slots  = %6.1f ns per slot\n", name, ms * 1e6 / iters, ms * 1e6 / iters / 12);
}

int main() {
  float* out; hipMalloc(&out, 256 * 256 * sizeof(float));
  run<0>("12 MFMA (one chain)", out);
  run<7>("12 MFMA (two chains)", out);
  run<9>("MFMA + 2 s_nop 7", out);
  run<1>("MFMA + 4 v_add_f32", out);
  run<2>("MFMA + 6 v_add_f32", out);
  run<4>("MFMA + 8 v_add_f32", out);
  run<3>("MFMA + split chain (5 dependent VALU)", out);
  run<8>("split chain only (5 VALU)", out);
  run<5>("MFMA + ds_read_b128", out);
  run<6>("MFMA + split chain + ds_read_b128", out);
  return 0;
}
