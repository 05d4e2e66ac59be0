// Graph-mode kernels (HNSW-style level-0 best-first search over an HBM-resident graph).
// Placeholder translation unit: the flat (exhaustive) path is built first; see DESIGN.md.
#include "ehx_kernels.h"
namespace ehx {}
