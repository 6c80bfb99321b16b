// Do unaligned dword accesses work on this box?  (buffer loads / stores with a byte offset that is not a multiple of 4,
// global loads / stores likewise; what a raw buffer load returns for a dword that straddles num_records.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void probe(const uint8_t* src, uint8_t* dst, uint32_t* res, int n) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, n, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, n, 0x00020000);
  const int t = threadIdx.x;
  for (int sh = 0; sh < 4; ++sh) {
    res[sh * 256 + t] = __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * t + sh, 0, 0);                // vector offset unaligned
    res[1024 + sh * 256 + t] = __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * t, sh, 0);             // scalar offset unaligned
    res[2048 + sh * 256 + t] = *reinterpret_cast<const uint32_t*>(src + 4 * t + sh);              // global
  }
  // stores: dst + 2048*sh + 4t + sh
  for (int sh = 0; sh < 4; ++sh) {
    const uint32_t v = 0x03020100u + 0x04040404u * (uint32_t)t;
    __builtin_amdgcn_raw_buffer_store_b32(v, rd, 4 * t + sh, 2048 * sh, 2);
    *reinterpret_cast<uint32_t*>(dst + 8192 + 2048 * sh + 4 * t + sh) = v;
  }
  // straddling num_records: dword at n-2 (2 bytes in range)
  if (t == 0) {
    res[3072] = __builtin_amdgcn_raw_buffer_load_b32(rs, n - 2, 0, 0);
    res[3073] = __builtin_amdgcn_raw_buffer_load_b32(rs, n - 4, 0, 0);
    res[3074] = __builtin_amdgcn_raw_buffer_load_b32(rs, n - 3, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0xab, rd, n - 1, 0, 0);
  }
}
int main() {
  const int n = 16384 + 3;
  std::vector<uint8_t> h(n + 64);
  for (int i = 0; i < n + 64; ++i) h[i] = (uint8_t)(i * 7 + 1);
  uint8_t *s, *d; uint32_t* r;
  hipMalloc(&s, n + 64); hipMalloc(&d, n + 64); hipMalloc(&r, 4096 * 4);
  hipMemcpy(s, h.data(), n + 64, hipMemcpyHostToDevice);
  hipMemset(d, 0xee, n + 64);
  hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, s, d, r, n);
  std::vector<uint32_t> res(4096); std::vector<uint8_t> out(n + 64);
  if (hipMemcpy(res.data(), r, 4096 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }
  hipMemcpy(out.data(), d, n + 64, hipMemcpyDeviceToHost);
  int bad[3] = {0, 0, 0};
  for (int sh = 0; sh < 4; ++sh)
    for (int t = 0; t < 256; ++t) {
      uint32_t e = 0;
      for (int j = 0; j < 4; ++j) e |= (uint32_t)h[4 * t + sh + j] << (8 * j);
      bad[0] += res[sh * 256 + t] != e; bad[1] += res[1024 + sh * 256 + t] != e; bad[2] += res[2048 + sh * 256 + t] != e;
    }
  int sbad[2] = {0, 0};
  for (int sh = 0; sh < 4; ++sh)
    for (int t = 0; t < 256; ++t)
      for (int j = 0; j < 4; ++j) {
        const uint8_t e = (uint8_t)(j + 4 * t);
        sbad[0] += out[2048 * sh + 4 * t + sh + j] != e;
        sbad[1] += out[8192 + 2048 * sh + 4 * t + sh + j] != e;
      }
  for (int sh = 0; sh < 4; ++sh) {
    printf("store sh=%d buffer:", sh);
    for (int i = 0; i < 24; ++i) printf(" %02x", out[2048 * sh + i]);
    printf("  ... tail:");
    for (int i = 1016; i < 1032; ++i) printf(" %02x", out[2048 * sh + i]);
    printf("\nstore sh=%d global:", sh);
    for (int i = 0; i < 24; ++i) printf(" %02x", out[8192 + 2048 * sh + i]);
    printf("\n");
  }
  uint32_t e2 = 0, e4 = 0, e3 = 0;
  for (int j = 0; j < 4; ++j) { e2 |= (uint32_t)h[n - 2 + j] << (8 * j); e4 |= (uint32_t)h[n - 4 + j] << (8 * j); e3 |= (uint32_t)h[n - 3 + j] << (8 * j); }
  printf("unaligned loads: buffer voffset bad=%d, buffer soffset bad=%d, global bad=%d\n", bad[0], bad[1], bad[2]);
  printf("unaligned stores: buffer bad=%d, global bad=%d\n", sbad[0], sbad[1]);
  printf("straddle n-2: got %08x (in-memory %08x); n-4: got %08x (exp %08x); n-3: got %08x (in-memory %08x); byte store at n-1: %02x\n",
         res[3072], e2, res[3073], e4, res[3074], e3, out[n - 1]);
  return 0;
}
