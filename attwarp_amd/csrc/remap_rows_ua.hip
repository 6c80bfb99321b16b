// The "unaligned" instantiations of the float32 staged resample (kernel: remap_rows_kernel.hpp, UA = true): interleaved
// rows whose length is not a multiple of 4 floats (683 x 3), or images that do not start on a 16-byte boundary --
// cv2.remap of AGW/new_method.py:268-271 / MN/checkpoint_utils.py:195-198 on whatever size the image has.  Both
// arithmetic modes; planar images come here plane by plane (remap_rows.hip).  A translation unit of its own so that
// the aligned kernels' compile time does not grow.
#include "remap_rows_kernel.hpp"

namespace attwarp {

template <int KI, int KO, int MODE, bool SINGLE>
static int launch_ua_t(const RowsParams& p, hipStream_t st) {
  const size_t lds = rows_lds_bytes<MODE, SINGLE>(KI, NT_BIG) + (size_t)p.lds_pad;
  hipLaunchKernelGGL((remap_rows_kernel<NT_BIG, KI, KO, true, false, false, MODE, SINGLE, true>), dim3(p.nblocks), dim3(NT_BIG), lds, st, p);
  return check_launch("remap_rows_kernel (unaligned rows)");
}
template <int KI, int MODE, bool SINGLE>
static int launch_ua_ki(const RowsParams& p, int ko, hipStream_t st) {
  if (ko <= 4) return launch_ua_t<KI, 4, MODE, SINGLE>(p, st);
  if (ko <= 8) return launch_ua_t<KI, 8, MODE, SINGLE>(p, st);
  if (ko <= 12) return launch_ua_t<KI, 12, MODE, SINGLE>(p, st);
  return launch_ua_t<KI, 16, MODE, SINGLE>(p, st);
}
// cv2 rows wider than 8 KB: one [top | bottom] LDS buffer (SINGLE), as the aligned kernels' 12-16 KB class
template <int MODE>
static int launch_ua_mode(const RowsParams& p, hipStream_t st) {
  constexpr bool CV = MODE == ATTWARP_CV2;
  const int ki = (p.VLV + NT_BIG - 1) / NT_BIG, ko = (p.OVL + NT_BIG - 1) / NT_BIG;
  if (ki <= 1) return launch_ua_ki<1, MODE, false>(p, ko, st);
  if (ki == 2) return launch_ua_ki<2, MODE, false>(p, ko, st);
  if (ki == 3) return launch_ua_ki<3, MODE, CV>(p, ko, st);
  return launch_ua_ki<4, MODE, CV>(p, ko, st);
}

int launch_rows_ua(const RowsParams& p, int mode, hipStream_t st) {
  return mode == ATTWARP_CV2 ? launch_ua_mode<ATTWARP_CV2>(p, st) : launch_ua_mode<ATTWARP_EXACT>(p, st);
}

}  // namespace attwarp
