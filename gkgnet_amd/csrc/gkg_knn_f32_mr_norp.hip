// fp32-contract k-NN tile kernel with the max-relative aggregation in its epilogue (row g2), forms without a positional
// bias: a parallel-build unit of gkg_knn_f32.hip.
#define GKG_KNN_MR_PART 1
#define GKG_KNN_NORP_PART 1
#include "gkg_knn_f32.hip"
