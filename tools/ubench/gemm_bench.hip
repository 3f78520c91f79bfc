// Kernel-only timing of gemm_f32_kernel tile variants (HIP events around back-to-back launches, no host framework).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I gkgnet_amd/csrc tools/ubench/gemm_bench.hip
//         gkgnet_amd/csrc/gkg_api.hip -o tools/ubench/gemm_bench
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../gkgnet_amd/csrc/gkg_gemm.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill(float* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; p[i] = ((h & 0xffff) / 32768.0f - 1.0f); }
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

int main() {
  struct Shape { int R, cin, cout, nb; };
  std::vector<Shape> shapes = {{10368, 320, 320, 1}, {10368, 160, 160, 4}, {10368, 640, 320, 1}, {2560, 320, 320, 1},
                               {2560, 320, 1280, 1}, {2560, 1280, 320, 1}, {41472, 400, 400, 1}, {41472, 800, 400, 1}};
  for (auto s : shapes) {
    size_t na = (size_t)s.nb * s.R * s.cin, nw = (size_t)s.nb * s.cout * s.cin, nc = (size_t)s.nb * s.R * s.cout;
    float *x, *w, *y, *bn; double* part; unsigned* ctr;
    CK(hipMalloc(&x, na * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&y, nc * 4));
    CK(hipMalloc(&part, (size_t)s.nb * 2 * s.cout * 8)); CK(hipMemset(part, 0, (size_t)s.nb * 2 * s.cout * 8)); CK(hipMalloc(&bn, (size_t)s.nb * s.cout * 8 * 4));
    CK(hipMalloc(&ctr, 4096 * 4)); CK(hipMemset(ctr, 0, 4096 * 4));
    fill<<<(na + 255) / 256, 256>>>(x, na, 1); fill<<<(nw + 255) / 256, 256>>>(w, nw, 2); fill<<<(s.nb * s.cout * 8 + 255) / 256, 256>>>(bn, (size_t)s.nb * s.cout * 8, 3);
    GemmArgs a{};
    a.A = x; a.a_bstride = (size_t)s.R * s.cin; a.lda = s.cin;
    a.B = w; a.b_bstride = (size_t)s.cout * s.cin; a.ldb = s.cin;
    a.C = y; a.c_bstride = (size_t)s.R * s.cout; a.ldc = s.cout;
    a.M = s.R; a.N = s.cout; a.K = s.cin;
    a.sums = part; a.counters = ctr;
    const double fl = 2.0 * s.R * s.cin * s.cout * s.nb;
    printf("R=%d cin=%d cout=%d nb=%d:\n", s.R, s.cin, s.cout, s.nb);
    auto rep = [&](const char* name, float us) { printf("   %-34s %8.1f us  %6.1f TF\n", name, us, fl / us / 1e6); };
    rep("mi32 128x64 4x1 BK32 store", timeit([&] { launch<32, 128, 64, 4, 1, LAY_KQ, LAY_KQ, false, EPI_STORE, 32>(a, s.nb, 0); }));
    rep("mi32 128x64 4x1 BK32 bnstats", timeit([&] { launch<32, 128, 64, 4, 1, LAY_KQ, LAY_KQ, false, EPI_BNSTATS, 32>(a, s.nb, 0); }));
    rep("mi32 64x64 2x2 BK32 store", timeit([&] { launch<32, 64, 64, 2, 2, LAY_KQ, LAY_KQ, false, EPI_STORE, 32>(a, s.nb, 0); }));
    rep("mi32 64x64 2x2 BK32 bnstats", timeit([&] { launch<32, 64, 64, 2, 2, LAY_KQ, LAY_KQ, false, EPI_BNSTATS, 32>(a, s.nb, 0); }));
    rep("mi16 64x32 4x1 BK32 store", timeit([&] { launch<16, 64, 32, 4, 1, LAY_KQ, LAY_KQ, false, EPI_STORE, 32>(a, s.nb, 0); }));
    rep("mi16 32x64 2x2 BK32 store", timeit([&] { launch<16, 32, 64, 2, 2, LAY_KQ, LAY_KQ, false, EPI_STORE, 32>(a, s.nb, 0); }));
    {   // backward: dgrad / wgrad with the BN backward-apply prologue (through the C ABI)
      float *dz, *coef, *dx, *dw; void* ws;
      size_t wsb = gkg_linear_workspace_bytes(s.R, s.cin, s.cout, s.nb);
      CK(hipMalloc(&dz, nc * 4)); CK(hipMalloc(&coef, (size_t)s.nb * 3 * s.cout * 4)); CK(hipMalloc(&dx, na * 4)); CK(hipMalloc(&dw, nw * 4));
      CK(hipMalloc(&ws, wsb));
      fill<<<(nc + 255) / 256, 256>>>(dz, nc, 5); fill<<<(s.nb * 3 * s.cout + 255) / 256, 256>>>(coef, (size_t)s.nb * 3 * s.cout, 6);
      rep("dgrad (dual prologue)", timeit([&] { gkg_linear_bn_bwd(dz, s.cout, (size_t)s.R * s.cout, y, coef, x, w, dx, nullptr, s.R, s.cin, s.cout, s.nb, 0u, ws, wsb, ctr, 0); }));
      rep("wgrad atomic (memset incl.)", timeit([&] { gkg_linear_bn_bwd(dz, s.cout, (size_t)s.R * s.cout, y, coef, x, w, nullptr, dw, s.R, s.cin, s.cout, s.nb, 0u, ws, wsb, ctr, 0); }));
      rep("wgrad ordered (last arriver)", timeit([&] { gkg_linear_bn_bwd(dz, s.cout, (size_t)s.R * s.cout, y, coef, x, w, nullptr, dw, s.R, s.cin, s.cout, s.nb, GKG_LINEAR_DETERMINISTIC, ws, wsb, ctr, 0); }));
      {   // wgrad variants, launched directly
        const WgradPlan wp = wgrad_plan(s.R, s.cin, s.cout, s.nb);
        GemmArgs wa{};
        wa.A = dz; wa.A2 = y; wa.a_bstride = (size_t)s.R * s.cout; wa.lda = s.cout;
        wa.B = x; wa.b_bstride = (size_t)s.R * s.cin; wa.ldb = s.cin;
        wa.C = dw; wa.c_bstride = (size_t)s.cout * s.cin; wa.ldc = s.cin;
        wa.M = s.cout; wa.N = s.cin; wa.K = s.R;
        wa.coef = coef; wa.coef_stride = s.cout; wa.coef_bstride = (size_t)3 * s.cout;
        wa.splits = wp.splits; wa.k_per_split = wp.kper;
        printf("   (wgrad plan: %d tiles x %d splits, %d rows per split)\n", wp.tiles, wp.splits, wp.kper);
        rep("wgrad 64x64 2x2 dual BK32", timeit([&] { launch<32, 64, 64, 2, 2, LAY_KM, LAY_KM, true, EPI_SPLITK_ATOMIC, 32>(wa, s.nb, 0); }));
        rep("wgrad 64x64 2x2 single BK32", timeit([&] { launch<32, 64, 64, 2, 2, LAY_KM, LAY_KM, false, EPI_SPLITK_ATOMIC, 32>(wa, s.nb, 0); }));
        rep("wgrad 64x64 2x2 dual BK64", timeit([&] { launch<32, 64, 64, 2, 2, LAY_KM, LAY_KM, true, EPI_SPLITK_ATOMIC, 64>(wa, s.nb, 0); }));
        rep("wgrad 128x64 4x1 dual BK32", timeit([&] { launch<32, 128, 64, 4, 1, LAY_KM, LAY_KM, true, EPI_SPLITK_ATOMIC, 32>(wa, s.nb, 0); }));
        rep("wgrad 64x128 2x2 dual BK32", timeit([&] { launch<32, 64, 128, 2, 2, LAY_KM, LAY_KM, true, EPI_SPLITK_ATOMIC, 32>(wa, s.nb, 0); }));
        rep("wgrad 128x128 2x2 dual BK32", timeit([&] { launch<32, 128, 128, 2, 2, LAY_KM, LAY_KM, true, EPI_SPLITK_ATOMIC, 32>(wa, s.nb, 0); }));
        rep("wgrad mi16 64x64 2x2 dual BK32", timeit([&] { launch<16, 64, 64, 2, 2, LAY_KM, LAY_KM, true, EPI_SPLITK_ATOMIC, 32>(wa, s.nb, 0); }));
        wa.splits = wp.splits * 2 > 64 ? 64 : wp.splits * 2; wa.k_per_split = ((s.R + wa.splits - 1) / wa.splits + 31) / 32 * 32; wa.splits = (s.R + wa.k_per_split - 1) / wa.k_per_split;
        rep("wgrad 128x128 2x2 dual, 2x splits", timeit([&] { launch<32, 128, 128, 2, 2, LAY_KM, LAY_KM, true, EPI_SPLITK_ATOMIC, 32>(wa, s.nb, 0); }));
        rep("wgrad 128x64 4x1 dual, 2x splits", timeit([&] { launch<32, 128, 64, 4, 1, LAY_KM, LAY_KM, true, EPI_SPLITK_ATOMIC, 32>(wa, s.nb, 0); }));
      }
      hipFree(dz); hipFree(coef); hipFree(dx); hipFree(dw); hipFree(ws);
    }
    CK(hipDeviceSynchronize());
    hipFree(x); hipFree(w); hipFree(y); hipFree(part); hipFree(bn); hipFree(ctr);
  }
  return 0;
}
