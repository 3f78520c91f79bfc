// The prefilter k-NN kernel without a positional bias: the same source as gkg_knn_pf.hip, compiled beside it.
#define GKG_KNN_NORP_PART 1
#include "gkg_knn_pf.hip"
