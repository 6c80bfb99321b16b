// CV2-mode instantiations of the float32 staged resample (kernel: remap_rows_kernel.hpp): OpenCV's
// 1/32-pixel coordinate quantisation and 4-weight sum, AGW/new_method.py:268-271, MN/checkpoint_utils.py:195-198.
#include "remap_rows_kernel.hpp"

namespace attwarp {

int launch_rows_cv2(const RowsParams& p, int tile_ko, hipStream_t st, const StepExtra* ex) {
  if (ex) return launch_step_cv2(p, tile_ko, st, ex);
  const int ki = tile_ko == 12 ? 4 : tile_ko == 8 ? 3 : tile_ko == 4 ? 2 : (p.VLV + NT_BIG - 1) / NT_BIG;
  // Rows of <= 12 KB: two [top | bottom] buffers, one barrier per row (<= 48 KB of LDS).  Wider rows: ONE buffer and two
  // barriers per row (two buffers would not fit the 64 KB a launch gets by default).  History of the 8-12 KB class
  // (1024x1024x3, B=256): with the row loop of rounds 1-3 the one-buffer form won (1.157 against 1.181 ms: 48 KB leave 3
  // workgroups per CU); with the loop whose look-ahead overlaps the gather (round 4) a workgroup hides its own load latency
  // and the barrier is what it waits at: five leases, uniform maps 1.111-1.114 ms against 1.17-1.18 (and 1.115 for the best
  // one-buffer order, which pays 0.98 instead of 0.93 ms on peaked maps): tools/attic/lease_orders.py, docs/experiments.md.
  if (ki == 3 && tile_ko == 0 && tune(TUNE_REMAP_CV2_DOUBLE) != 0) return launch_rows_mode<ATTWARP_CV2, false, 3, 3, false>(p, tile_ko, st, nullptr);
  if (ki >= 3) return launch_rows_mode<ATTWARP_CV2, true, 3, 4, false>(p, tile_ko, st, nullptr);
  return launch_rows_mode<ATTWARP_CV2, false, 1, 2, false>(p, tile_ko, st, nullptr);
}

}  // namespace attwarp
