// bf16-contraction instantiations of the k-NN tile kernel (GKG_KNN_BF16_CONTRACT: the bf16-autocast inference form) —
// their own translation unit so that they compile beside the fp32 forms of gkg_knn.hip.
#include "gkg_knn_tile.h"

using namespace gkg;

// bf16 matrix-core contraction (GKG_KNN_BF16_CONTRACT): direct or buffered selection, guarded insert
// HAS_RP is a compile-time parameter of the launcher: the relative_pos forms are instantiated in this translation unit, the
// others in gkg_knn_bf_norp.hip (the same source with GKG_KNN_NORP_PART defined).
template <int KD, bool HAS_RP>
static hipError_t launch_tile_bf(const KnnArgs& a, dim3 grid, size_t lds, hipStream_t st, int wbuf, bool solo) {
  const bool buffered = wbuf > 0 && KD <= 36;   // 64-entry lists: direct insert only (the buffered forms spilled to scratch)
  GkgProfScope prof(GKG_PROF_KNN_TILE, st);
  if constexpr (KD <= 36) {
  if (buffered && solo) {                       // one wave per 64-query tile, all keys (see the kernel's NWV)
    // 9-entry lists: 12-entry candidate buffers (flush when a lane holds more than 4) — with one short list per query the
    // fresher threshold is worth more than fuller flushes (cfg3 forward, k-NN kernels 5.08 -> 4.90 ms); longer lists keep
    // 16 (cfg5, k*d = 18 / 36: 29.8 -> 33.2 ms with 12).  Measured on the models' own activations: random tokens
    // (tools/bench_knn_bf.py) rank the two the other way round for k*d = 18.
    constexpr int SBUF = KD <= 12 ? 12 : KNN_BUF;
    return launch_tile_v<KD, HAS_RP, 4, false, SBUF, true, 1>(a, grid, lds, st);
  }
  if (buffered) {
    // 9-entry lists: 12 entries per lane (fresher thresholds); longer lists: 16, or 12 where 16 would cost a workgroup per CU
    if (KD <= 12 || wbuf < 16) {
      return launch_tile_v<KD, HAS_RP, 4, false, 12, true>(a, grid, lds, st);
    }
    if constexpr (KD > 12) {
      return launch_tile_v<KD, HAS_RP, 4, false, KNN_BUF, true>(a, grid, lds, st);
    }
  }
  }   // KD <= 36
  return launch_tile_v<KD, HAS_RP, 4, true, 0, true>(a, grid, lds, st);
}

namespace gkg {
#ifdef GKG_KNN_NORP_PART
hipError_t launch_knn_tile_bf_norp(const KnnArgs& a, dim3 grid, size_t lds, int KD, int wbuf, bool solo, hipStream_t st) {
  switch (KD) {
    case 9: return launch_tile_bf<9, false>(a, grid, lds, st, wbuf, solo);
    case 18: return launch_tile_bf<18, false>(a, grid, lds, st, wbuf, solo);
    case 27: return launch_tile_bf<27, false>(a, grid, lds, st, wbuf, solo);
    case 36: return launch_tile_bf<36, false>(a, grid, lds, st, wbuf, solo);
    default: return launch_tile_bf<64, false>(a, grid, lds, st, wbuf, solo);
  }
}
#else
hipError_t launch_knn_tile_bf_norp(const KnnArgs& a, dim3 grid, size_t lds, int KD, int wbuf, bool solo, hipStream_t st);
hipError_t launch_knn_tile_bf(const KnnArgs& a, dim3 grid, size_t lds, int KD, int wbuf, bool solo, hipStream_t st) {
  if (!a.relpos) return launch_knn_tile_bf_norp(a, grid, lds, KD, wbuf, solo, st);
  switch (KD) {
    case 9: return launch_tile_bf<9, true>(a, grid, lds, st, wbuf, solo);
    case 18: return launch_tile_bf<18, true>(a, grid, lds, st, wbuf, solo);
    case 27: return launch_tile_bf<27, true>(a, grid, lds, st, wbuf, solo);
    case 36: return launch_tile_bf<36, true>(a, grid, lds, st, wbuf, solo);
    default: return launch_tile_bf<64, true>(a, grid, lds, st, wbuf, solo);
  }
}
#endif
}  // namespace gkg
