// The fp32-contract k-NN tile kernel without a positional bias: the same source as gkg_knn_f32.hip, compiled beside it.
#define GKG_KNN_NORP_PART 1
#include "gkg_knn_f32.hip"
