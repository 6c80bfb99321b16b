// CV2-mode instantiations of the float32 staged resample (kernel: remap_rows_kernel.hpp): OpenCV's
// 1/32-pixel coordinate quantisation and 4-weight sum, AGW/new_method.py:268-271, MN/checkpoint_utils.py:195-198.
#include "remap_rows_kernel.hpp"

namespace attwarp {

int launch_rows_cv2(const RowsParams& p, int tile_ko, hipStream_t st, const StepExtra* ex) {
  if (ex) return launch_step_cv2(p, tile_ko, st, ex);
  const int ki = tile_ko == 12 ? 4 : tile_ko == 8 ? 3 : (p.VLV + NT_BIG - 1) / NT_BIG;
  // Rows of <= 8 KB: two [top | bottom] buffers, one barrier per row (<= 32 KB of LDS).  Wider rows: ONE buffer and
  // two barriers per row -- measured on MI355X, 1024x1024x3 float32 B=256: 1.157 ms against 1.181 ms for the
  // double-buffered form, whose 48 KB of LDS leaves 3 workgroups per CU; at 16 KB rows two buffers would not
  // fit the 64 KB a launch gets by default.  With one buffer CV2 runs at the speed of EXACT mode.
  if (ki >= 3) return launch_rows_mode<ATTWARP_CV2, true, 3, 4, false>(p, tile_ko, st, nullptr);
  return launch_rows_mode<ATTWARP_CV2, false, 1, 2, false>(p, tile_ko, st, nullptr);
}

}  // namespace attwarp
