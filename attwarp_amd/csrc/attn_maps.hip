// Two independent pieces of a batch stream's step in ONE launch: the attention reduce of batch k+2 (A1,
// AGW/attention_extraction/llava.py:385-396) and the map construction of batch k+1 (A2 + A6 + A8 + A9 + A11) -- block
// ranges of one grid, bodies in attn_f32v.hpp / axis_blocks.hpp.  For large images, where the resample is 92 % of a step
// and keeps its own launch (attwarp_warp_step_fused loses there), this hides the latency-bound 25 us map kernel and
// one launch boundary behind the reduce.  Same arithmetic, bit for bit, as attwarp_attn_reduce_step +
// attwarp_axis_maps_from_steps_t.
#include "common.hpp"
#include "axis_blocks.hpp"
#include "attn_f32v.hpp"

#include <algorithm>
#include <cstring>

namespace attwarp {

// Register budget: both bodies share one allocation, and the reduce -- 5120 of the 5632 blocks -- is the one that needs
// the occupancy.  With 24 step maps per request the map body took 138 VGPRs (3 waves per SIMD for every block; the
// reduce alone runs at 82 VGPRs = 5 waves): 8 per request and a 5-wave bound keep the launch at the reduce's occupancy.
template <typename T>
__global__ __launch_bounds__(ATTN_NT, 5) void attn_maps_kernel(const AttnStepArgsT<T> a, const StepsMapsArgs m, int n_maps8) {
  extern __shared__ __attribute__((aligned(16))) float am_smem[];
  __shared__ float s_tmp[64], s_pm[64];
  const int blk = blockIdx.x;
  if (blk < n_maps8) {          // map blocks first: the longest dependent chain starts at once
    if (blk < 2 * m.B) axis_maps_from_steps_block<8, T>(m, blk >> 1, blk & 1, reinterpret_cast<double*>(am_smem), s_tmp, s_pm);
    return;
  }
  attn_reduce_v4_block<T, 3, 4>(a, blk - n_maps8, am_smem);
}

template <typename T>
static int launch_attn_maps(const void* rows, int n_rows, int heads, int kv_len, const int32_t* starts, int starts_mod,
                            int ntok, void* steps_out, const StepsMapsArgs& m, hipStream_t st) {
  AttnStepArgsT<T> a;
  a.attn = static_cast<const T*>(rows); a.heads = heads; a.sb = (int64_t)heads * kv_len; a.sh = kv_len; a.row_off = 0;
  a.starts = starts; a.starts_mod = starts_mod; a.max_start = kv_len - ntok; a.ntok = ntok; a.out = static_cast<T*>(steps_out);
  const int n_maps8 = ((2 * m.B + 7) / 8) * 8;
  const size_t lds = std::max(attn_v4_lds_bytes<3>(), steps_maps_lds_bytes(std::max(m.W, m.H), m.g));
  if (const int rc = grant_dynamic_lds(attn_maps_kernel<T>, lds, "attn_reduce_and_maps")) return rc;
  hipLaunchKernelGGL((attn_maps_kernel<T>), dim3(n_maps8 + n_rows), dim3(ATTN_NT), lds, st, a, m, n_maps8);
  return check_launch("attn_maps_kernel");
}

}  // namespace attwarp

using namespace attwarp;

extern "C" int attwarp_attn_reduce_and_maps(int attn_dtype, const void* rows, int n_rows, int heads, int kv_len,
                                            const int32_t* starts, int starts_mod, int ntok, void* steps_out,
                                            const void* steps_in, int T, int B, int g, int W, int H, int W_out, int H_out,
                                            const double* inv_x, const double* inv_y, float* map_x, float* map_y,
                                            void* stream) {
  ATTWARP_REQUIRE(rows && starts && steps_out && steps_in && inv_x && inv_y && map_x && map_y,
                  "attn_reduce_and_maps: null pointer");
  ATTWARP_REQUIRE(n_rows > 0 && heads > 0 && kv_len > 0 && ntok > 0 && starts_mod > 0 && T > 0 && B > 0 && g > 0 && W > 0 &&
                  H > 0 && W_out > 0 && H_out > 0, "attn_reduce_and_maps: non-positive size");
  ATTWARP_REQUIRE(attn_dtype == ATTWARP_F32 || attn_dtype == ATTWARP_F16 || attn_dtype == ATTWARP_BF16,
                  "attn_reduce_and_maps: attn_dtype must be F32, F16 or BF16 (got %d)", attn_dtype);
  ATTWARP_REQUIRE(ntok <= kv_len, "attn_reduce_and_maps: ntok=%d > kv_len=%d", ntok, kv_len);
  ATTWARP_REQUIRE(steps_out != steps_in, "attn_reduce_and_maps: steps_out must not alias steps_in");
  if (ntok % 4 != 0 || ntok > 3 * 4 * WAVE || ntok != g * g)
    return fail(ATTWARP_E_UNSUPPORTED, "attn_reduce_and_maps: ntok must equal g*g, be a multiple of 4 and <= 768");
  if (g > 32 || std::max(W, H) > 8192 || B > 65535)
    return fail(ATTWARP_E_UNSUPPORTED, "attn_reduce_and_maps: g > 32, max(W,H) > 8192 or B > 65535");
  StepsMapsArgs m;
  m.steps = steps_in; m.step_dtype = attn_dtype; m.T = T; m.B = B; m.g = g; m.W = W; m.H = H; m.W_out = W_out; m.H_out = H_out;
  m.inv_x = inv_x; m.inv_y = inv_y; m.map_x = map_x; m.map_y = map_y; m.att_out = nullptr;
  hipStream_t st = as_stream(stream);
  if (attn_dtype == ATTWARP_F32) return launch_attn_maps<float>(rows, n_rows, heads, kv_len, starts, starts_mod, ntok, steps_out, m, st);
  if (attn_dtype == ATTWARP_F16) return launch_attn_maps<__half>(rows, n_rows, heads, kv_len, starts, starts_mod, ntok, steps_out, m, st);
  return launch_attn_maps<__hip_bfloat16>(rows, n_rows, heads, kv_len, starts, starts_mod, ntok, steps_out, m, st);
}
