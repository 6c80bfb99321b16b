// EXACT-mode instantiations of the fused step kernel (warp_step_kernel, remap_rows_kernel.hpp): resample of batch k +
// map construction of batch k+1 + attention reduce of batch k+2 in one launch (attwarp_warp_step_fused).
#include "remap_rows_kernel.hpp"

namespace attwarp {

int launch_step_exact(const RowsParams& p, int tile_ko, hipStream_t st, const StepExtra* ex) {
  return launch_rows_mode<ATTWARP_EXACT, false, 1, 4, true>(p, tile_ko, st, ex);
}

}  // namespace attwarp
