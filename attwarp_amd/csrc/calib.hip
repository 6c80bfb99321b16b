// Measurement yardstick, TUNING FLAVOUR ONLY (libattwarp_hip_tuning.so; the product library gets an empty object file):
// ONE launch that streams `read_bytes` in and `write_bytes` out -- what a plain copy kernel reaches for the bytes of a step of
// the main_batched chain at the same batch size from the same ring (bench.py: `calibration` / `step_over_copy` of
// also_main_batched and also_main_batched_ragged).  At 18 + 24 MB per step (B=32, 336 -> 500) the fraction of the 8 TB/s HBM
// peak is the wrong ruler: launch ramp and tail are a large part of any kernel of that size.
#include "common.hpp"

#ifdef ATTWARP_TUNING
namespace attwarp {

__global__ __launch_bounds__(256) void debug_stream_copy_kernel(const uint4* __restrict__ src, size_t nr, uint4* __restrict__ dst,
                                                                size_t nw) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  uint4 acc = {0u, 0u, 0u, 0u};
  size_t k = i;
  for (; k + 3 * stride < nr; k += 4 * stride) {            // four 16-byte loads in flight per thread
    const uint4 a = src[k], b = src[k + stride], c = src[k + 2 * stride], d = src[k + 3 * stride];
    acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y;
    acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
  }
  for (; k < nr; k += stride) { const uint4 a = src[k]; acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w; }
  for (k = i; k < nw; k += stride) dst[k] = acc;            // (the value written depends on everything read: nothing is elided)
}

}  // namespace attwarp

extern "C" int attwarp_debug_stream_copy(const void* src, size_t read_bytes, void* dst, size_t write_bytes, void* stream) {
  using namespace attwarp;
  ATTWARP_REQUIRE(src && dst, "debug_stream_copy: null pointer");
  ATTWARP_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0, "debug_stream_copy: 16-byte aligned buffers");
  const size_t nr = read_bytes / 16, nw = write_bytes / 16, nmax = nr > nw ? nr : nw;
  if (nmax == 0) return ATTWARP_OK;
  size_t blocks = (nmax + 1023) / 1024;                      // ~4 vectors per thread on the longer side
  if (blocks > 1u << 20) blocks = 1u << 20;
  hipLaunchKernelGGL(debug_stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                     static_cast<const uint4*>(src), nr, static_cast<uint4*>(dst), nw);
  return check_launch("debug_stream_copy_kernel");
}
#endif
