// CV2-mode instantiations of the fused step kernel (warp_step_kernel, remap_rows_kernel.hpp): resample of batch k +
// map construction of batch k+1 + attention reduce of batch k+2 in one launch (attwarp_warp_step_fused).
#include "remap_rows_kernel.hpp"

namespace attwarp {

int launch_step_cv2(const RowsParams& p, int tile_ko, hipStream_t st, const StepExtra* ex) {
  const int ki = (p.VLV + NT_BIG - 1) / NT_BIG;
  if (ki >= 3) return launch_rows_mode<ATTWARP_CV2, true, 3, 4, true>(p, tile_ko, st, ex);     // as launch_rows_cv2
  return launch_rows_mode<ATTWARP_CV2, false, 1, 2, true>(p, tile_ko, st, ex);
}

}  // namespace attwarp
