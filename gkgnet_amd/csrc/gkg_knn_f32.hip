// fp32-contract instantiations of the k-NN tile kernel (direct / guard-less / buffered selection) — their own translation
// unit, compiled twice: with the positional bias here, without it in gkg_knn_f32_norp.hip (GKG_KNN_NORP_PART).
#include "gkg_knn_tile.h"

using namespace gkg;

// GKG_KNN_MR_PART (gkg_knn_f32_mr.hip / gkg_knn_f32_mr_norp.hip): the same forms with the max-relative aggregation in the
// epilogue (knn_tile_kernel<..., MRF = true>), modes 0-2.
#ifdef GKG_KNN_MR_PART
constexpr bool kMRF = true;
#else
constexpr bool kMRF = false;
#endif

// mode 0: direct selection with the ballot guard; 1: without it — short key streams (< 10 key tiles per wave: the 18x18
// stage, label graphs over it), only for the 9-entry list: with 18 or 27 entries the insert is expensive enough that
// skipping it wins again (measured: k*d = 27 at 18x18 is 19 % slower without the guard); 2: buffered selection (long key
// streams per wave and / or long lists, see the kernel's comment)
template <int KD, bool HAS_RP>
static hipError_t launch_tile_f32(const KnnArgs& a, dim3 grid, size_t lds, int mode, hipStream_t st) {
  GkgProfScope prof(GKG_PROF_KNN_TILE, st);
  const bool deep = (a.cpad % 16) == 0;      // 8 k-pairs per register batch when the channel count allows
  // 6 per batch for padded widths of 12, 24, 36, 60 ... (the plan pads to a multiple of 12 only for the buffered forms below)
  const bool six = !deep && (a.cpad % 12) == 0;
  // the catch-all 64-entry list has no buffered form: its list (128 registers) + the 16-entry batch + the merge network's
  // temporaries spilled 100-400 VGPRs to scratch in every such instantiation; the guarded direct insert fits
  if constexpr (KD <= 36) {
    // mode 5: buffered selection with ONE wave per workgroup (the wave streams all keys of its 64 queries: one list per query
    // instead of four quarter-stream lists, no merge) — launches with enough query tiles to fill the chip that way
    if constexpr (!kMRF) {
      if (mode == 5 && six) return launch_tile_v<KD, HAS_RP, 6, false, KNN_BUF, false, 1>(a, grid, lds, st);
      if (mode == 5) return deep ? launch_tile_v<KD, HAS_RP, 8, false, KNN_BUF, false, 1>(a, grid, lds, st) : launch_tile_v<KD, HAS_RP, 4, false, KNN_BUF, false, 1>(a, grid, lds, st);
      if (mode == 2 && six) return launch_tile_v<KD, HAS_RP, 6, false, KNN_BUF, false, NW, false>(a, grid, lds, st);
    }
    if (mode == 2) return deep ? launch_tile_v<KD, HAS_RP, 8, false, KNN_BUF, false, NW, kMRF>(a, grid, lds, st) : launch_tile_v<KD, HAS_RP, 4, false, KNN_BUF, false, NW, kMRF>(a, grid, lds, st);
  }
  if constexpr (KD == 9) {
    if (mode == 1) return deep ? launch_tile_v<KD, HAS_RP, 8, false, 0, false, NW, kMRF>(a, grid, lds, st) : launch_tile_v<KD, HAS_RP, 4, false, 0, false, NW, kMRF>(a, grid, lds, st);
  }
  return deep ? launch_tile_v<KD, HAS_RP, 8, true, 0, false, NW, kMRF>(a, grid, lds, st) : launch_tile_v<KD, HAS_RP, 4, true, 0, false, NW, kMRF>(a, grid, lds, st);
}

template <bool HAS_RP>
static hipError_t launch_tile_f32_kd(const KnnArgs& a, dim3 grid, size_t lds, int KD, int mode, hipStream_t st) {
  switch (KD) {
    case 9: return launch_tile_f32<9, HAS_RP>(a, grid, lds, mode, st);
    case 18: return launch_tile_f32<18, HAS_RP>(a, grid, lds, mode, st);
    case 27: return launch_tile_f32<27, HAS_RP>(a, grid, lds, mode, st);
    case 36: return launch_tile_f32<36, HAS_RP>(a, grid, lds, mode, st);
    default:
      if constexpr (kMRF) return hipErrorInvalidValue;          // the fused forms stop at 36-entry lists (the caller checks)
      else return launch_tile_f32<64, HAS_RP>(a, grid, lds, mode, st);
  }
}

namespace gkg {
#ifdef GKG_KNN_MR_PART
#ifdef GKG_KNN_NORP_PART
hipError_t launch_knn_tile_f32_mr_norp(const KnnArgs& a, dim3 grid, size_t lds, int KD, int mode, hipStream_t st) {
  return launch_tile_f32_kd<false>(a, grid, lds, KD, mode, st);
}
#else
hipError_t launch_knn_tile_f32_mr_norp(const KnnArgs& a, dim3 grid, size_t lds, int KD, int mode, hipStream_t st);
hipError_t launch_knn_tile_f32_mr(const KnnArgs& a, dim3 grid, size_t lds, int KD, int mode, hipStream_t st) {
  if (!a.relpos) return launch_knn_tile_f32_mr_norp(a, grid, lds, KD, mode, st);
  return launch_tile_f32_kd<true>(a, grid, lds, KD, mode, st);
}
#endif
#else
#ifdef GKG_KNN_NORP_PART
hipError_t launch_knn_tile_f32_norp(const KnnArgs& a, dim3 grid, size_t lds, int KD, int mode, hipStream_t st) {
  return launch_tile_f32_kd<false>(a, grid, lds, KD, mode, st);
}
#else
hipError_t launch_knn_tile_f32_norp(const KnnArgs& a, dim3 grid, size_t lds, int KD, int mode, hipStream_t st);
hipError_t launch_knn_tile_f32(const KnnArgs& a, dim3 grid, size_t lds, int KD, int mode, hipStream_t st) {
  if (!a.relpos) return launch_knn_tile_f32_norp(a, grid, lds, KD, mode, st);
  return launch_tile_f32_kd<true>(a, grid, lds, KD, mode, st);
}
#endif
#endif
}  // namespace gkg
