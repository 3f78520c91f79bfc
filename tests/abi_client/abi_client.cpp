// TEST INFRASTRUCTURE — a torch-free client of the C ABI (include/gkg_hip.h).
//
// What a maintainer of the reference would link against is libgkg_hip.so and nothing else: this program takes device memory
// from the HIP runtime, calls the drop-in entry points in the reference's own (B*G, c, N) layout —
//     gkg_knn_fwd   for DenseDilatedKnnGraph.forward   (vig_model/torch_edge.py:164-176)
//     gkg_mr_fwd    for MRConv2d's two batched_index_select + max   (vig_model/torch_vertex.py:49-54, torch_nn.py:84-105)
//     gkg_mr_bwd    for their autograd (SURVEY.md §8a backward contract)
// — and checks every output bit for bit against the CPU oracle (oracle/gkg_oracle.c, linked as libgkg_oracle.so; the checker,
// never the product).  Cases: self graph with a positional bias and dilation, bipartite graph (label queries over image keys),
// a ragged size, exact duplicate tokens (ties decided by the key index).  Built by __graft_entry__.build(); run by
// tests/test_hip_abi_client.py.  Exit code 0 = every case bit-exact.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gkg_hip.h"

extern "C" {
int oracle_knn_fwd(const float* x, const float* y, const float* relpos, int64_t* nn_idx, int64_t* center, int BG, int c, int N, int M,
                   int k, int dilation, unsigned flags, float* dist_out);
int oracle_mr_fwd(const float* x, const float* src, const int64_t* nn_idx, float* m_out, uint8_t* argmax, int BG, int c, int N, int M,
                  int k);
int oracle_mr_bwd(const float* g, const int64_t* nn_idx, const uint8_t* argmax, float* gx, float* gsrc, int BG, int c, int N, int M,
                  int k);
}

#define HIP_OK(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      std::exit(2);                                                                               \
    }                                                                                             \
  } while (0)

namespace {

struct Rng {                                     // xorshift + Box-Muller: the same inputs on every run, no library involved
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
  double uni() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (double)(s >> 11) * (1.0 / 9007199254740992.0);
  }
  float normal() {
    const double u = uni() + 1e-300, v = uni();
    return (float)(std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v));
  }
};

template <typename T>
T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  HIP_OK(hipMalloc(&d, h.size() * sizeof(T) + 16));
  HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}
template <typename T>
T* device_buffer(size_t n) {
  T* d = nullptr;
  HIP_OK(hipMalloc(&d, n * sizeof(T) + 16));
  HIP_OK(hipMemset(d, 0xff, n * sizeof(T)));    // poison: an output the library forgot to write would show
  return d;
}
template <typename T>
std::vector<T> to_host(const T* d, size_t n) {
  std::vector<T> h(n);
  HIP_OK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
  return h;
}
template <typename T>
size_t mismatches(const std::vector<T>& a, const std::vector<T>& b) {
  size_t bad = 0;
  for (size_t i = 0; i < a.size(); ++i) bad += std::memcmp(&a[i], &b[i], sizeof(T)) != 0;
  return bad;
}

struct Case {
  const char* name;
  int BG, c, N, M, k, d;
  bool bipartite, relpos, duplicates;
};

int run(const Case& cs, hipStream_t st) {
  const int BG = cs.BG, c = cs.c, N = cs.N, M = cs.bipartite ? cs.M : cs.N, k = cs.k, d = cs.d;
  Rng rng(1234 + (uint64_t)BG * 7 + (uint64_t)N);
  std::vector<float> x((size_t)BG * c * N), y, rp, g((size_t)BG * c * N);
  for (auto& v : x) v = rng.normal();
  for (auto& v : g) v = rng.normal();
  if (cs.bipartite) {
    y.resize((size_t)BG * c * M);
    for (auto& v : y) v = rng.normal();
  }
  if (cs.duplicates) {                           // a third of the key tokens are exact copies of another third: distance ties
    std::vector<float>& t = cs.bipartite ? y : x;
    const int T = (cs.bipartite ? M : N), third = T / 3;
    for (int bg = 0; bg < BG; ++bg)
      for (int ch = 0; ch < c; ++ch)
        for (int i = 0; i < third; ++i) t[((size_t)bg * c + ch) * T + third + i] = t[((size_t)bg * c + ch) * T + i];
  }
  if (cs.relpos) {                               // multiples of 1/8 in [-8, 0]: exactly representable, like the tests' biases
    rp.resize((size_t)N * M);
    for (auto& v : rp) v = -(float)std::floor(rng.uni() * 64.0) / 8.0f;
  }
  // ---- the oracle
  std::vector<int64_t> want_idx((size_t)BG * N * k), want_ctr((size_t)BG * N * k);
  std::vector<float> want_m((size_t)BG * c * N), want_gx((size_t)BG * c * N), want_gs;
  std::vector<uint8_t> want_arg((size_t)BG * c * N);
  if (oracle_knn_fwd(x.data(), cs.bipartite ? y.data() : nullptr, cs.relpos ? rp.data() : nullptr, want_idx.data(), want_ctr.data(), BG,
                     c, N, M, k, d, GKG_KNN_NORMALIZE, nullptr) != 0 ||
      oracle_mr_fwd(x.data(), cs.bipartite ? y.data() : nullptr, want_idx.data(), want_m.data(), want_arg.data(), BG, c, N, M, k) != 0)
    return std::fprintf(stderr, "%s: oracle refused the case\n", cs.name), 1;
  if (cs.bipartite) want_gs.resize((size_t)BG * c * M);
  if (oracle_mr_bwd(g.data(), want_idx.data(), want_arg.data(), want_gx.data(), cs.bipartite ? want_gs.data() : nullptr, BG, c, N, M, k) != 0)
    return std::fprintf(stderr, "%s: oracle refused the backward\n", cs.name), 1;
  // ---- the library, through the header only
  float* dx = to_device(x);
  float* dy = cs.bipartite ? to_device(y) : nullptr;
  float* drp = cs.relpos ? to_device(rp) : nullptr;
  float* dg = to_device(g);
  int64_t* didx = device_buffer<int64_t>((size_t)BG * N * k);
  int64_t* dctr = device_buffer<int64_t>((size_t)BG * N * k);
  float* dm = device_buffer<float>((size_t)BG * c * N);
  uint8_t* darg = device_buffer<uint8_t>((size_t)BG * c * N);
  float* dgx = device_buffer<float>((size_t)BG * c * N);
  float* dgs = cs.bipartite ? device_buffer<float>((size_t)BG * c * M) : nullptr;
  const size_t wsb = gkg_knn_workspace_bytes(BG, c, N, M, k, d, GKG_F32, GKG_KNN_NORMALIZE);
  void* ws = nullptr;
  HIP_OK(hipMalloc(&ws, wsb + 16));
  int rc = gkg_knn_fwd(dx, dy, drp, didx, dctr, BG, c, N, M, k, d, GKG_F32, GKG_KNN_NORMALIZE, ws, wsb, st);
  if (rc == 0) rc = gkg_mr_fwd(dx, dy, didx, dm, darg, BG, c, N, M, k, GKG_F32, st);
  if (rc == 0) rc = gkg_mr_bwd(dg, didx, darg, dgx, dgs, BG, c, N, M, k, GKG_F32, st);
  if (rc != 0) return std::fprintf(stderr, "%s: library returned %d (%s)\n", cs.name, rc, gkg_last_error_string()), 1;
  HIP_OK(hipStreamSynchronize(st));
  const size_t b_idx = mismatches(to_host(didx, want_idx.size()), want_idx), b_ctr = mismatches(to_host(dctr, want_ctr.size()), want_ctr);
  const size_t b_m = mismatches(to_host(dm, want_m.size()), want_m), b_arg = mismatches(to_host(darg, want_arg.size()), want_arg);
  // backward: -g and sums of g; the library's scatter is exact fixed point rounded once, the oracle adds in index order
  // (fp32, order-dependent): compared to 1e-5 relative, as tests/test_hip_ops.py does
  size_t b_g = 0;
  auto close = [&](const std::vector<float>& got, const std::vector<float>& want) {
    for (size_t i = 0; i < got.size(); ++i) b_g += !(std::fabs(got[i] - want[i]) <= 1e-5f * std::fabs(want[i]) + 1e-5f);
  };
  close(to_host(dgx, want_gx.size()), want_gx);
  if (cs.bipartite) close(to_host(dgs, want_gs.size()), want_gs);
  std::printf("%-34s BG %3d c %3d N %5d M %5d k %2d d %d  workspace %8zu B  mismatches: idx %zu centre %zu m %zu argmax %zu grad %zu\n",
              cs.name, BG, c, N, M, k, d, wsb, b_idx, b_ctr, b_m, b_arg, b_g);
  for (void* p : {(void*)dx, (void*)dy, (void*)drp, (void*)dg, (void*)didx, (void*)dctr, (void*)dm, (void*)darg, (void*)dgx, (void*)dgs, ws})
    if (p) HIP_OK(hipFree(p));
  return (b_idx || b_ctr || b_m || b_arg || b_g) ? 1 : 0;
}

}  // namespace

int main() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return std::fprintf(stderr, "no HIP device\n"), 3;
  if (gkg_version() != GKG_ABI_VERSION) return std::fprintf(stderr, "library ABI %d, header %d\n", gkg_version(), GKG_ABI_VERSION), 3;
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  const Case cases[] = {
      {"self graph, bias, dilation 2", 8, 32, 196, 196, 9, 2, false, true, false},     // BASELINE cfg1-like (14 x 14 tokens)
      {"label queries over image keys", 8, 80, 80, 324, 9, 1, true, false, false},     // GrapherLabel at 18 x 18
      {"ragged sizes", 3, 20, 77, 131, 5, 3, true, true, false},
      {"exact duplicates (index ties)", 4, 16, 150, 150, 9, 1, false, false, true},
      {"pooled keys, long stream", 2, 40, 2304, 576, 9, 1, true, true, false},         // stage-1-like: queries over r = 2 pooled keys
  };
  int bad = 0;
  for (const Case& cs : cases) bad += run(cs, st);
  HIP_OK(hipStreamDestroy(st));
  std::printf(bad ? "FAILED: %d case(s)\n" : "all %d cases bit-exact against the oracle\n", bad ? bad : (int)(sizeof(cases) / sizeof(cases[0])));
  return bad ? 1 : 0;
}
