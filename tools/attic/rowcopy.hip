// Dev tool: is remap_rows_kernel limited by its access PATTERN or by its own overheads?
// Copies [rows x 3072] float32 with the same work decomposition (one 256-thread workgroup per R consecutive
// 12 KB rows, 3 x 16-byte loads and stores per thread per row) but no LDS, no barrier, no arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
template <int MODE>
__global__ __launch_bounds__(256) void rowcopy(const float4* __restrict__ src, float4* __restrict__ dst, int nrows, int R) {
  const int tid = threadIdx.x;
  const long long row0 = (long long)blockIdx.x * R;
  for (int r = 0; r < R; ++r) {
    const long long row = row0 + r;
    if (row >= nrows) break;
    const float4* s = src + row * 768;
    float4* d = dst + row * 768;
    float4 a, b, c;
    if (MODE >= 2) {   // nontemporal loads
      typedef float v4f __attribute__((ext_vector_type(4)));
      const v4f* sv = reinterpret_cast<const v4f*>(s);
      v4f ta = __builtin_nontemporal_load(sv + tid), tb = __builtin_nontemporal_load(sv + tid + 256), tc = __builtin_nontemporal_load(sv + tid + 512);
      a = make_float4(ta.x, ta.y, ta.z, ta.w); b = make_float4(tb.x, tb.y, tb.z, tb.w); c = make_float4(tc.x, tc.y, tc.z, tc.w);
    } else {
      a = s[tid]; b = s[tid + 256]; c = s[tid + 512];
    }
    if (MODE == 1) {  // dword stores, 256 B per wave instruction, like the resample kernel
      float* df = reinterpret_cast<float*>(d);
      const float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
      for (int k = 0; k < 12; ++k) df[tid + 256 * k] = v[k];
    } else if (MODE == 3) {   // nontemporal loads AND stores
      typedef float v4f __attribute__((ext_vector_type(4)));
      v4f* dv = reinterpret_cast<v4f*>(d);
      v4f ta = {a.x, a.y, a.z, a.w}, tb = {b.x, b.y, b.z, b.w}, tc = {c.x, c.y, c.z, c.w};
      __builtin_nontemporal_store(ta, dv + tid); __builtin_nontemporal_store(tb, dv + tid + 256); __builtin_nontemporal_store(tc, dv + tid + 512);
    } else {
      d[tid] = a; d[tid + 256] = b; d[tid + 512] = c;
    }
  }
}
// the resample kernel's block orders: mode 0 = contiguous range of row blocks per XCD, g >= 2 = XCDs interleaved in groups of g
__global__ __launch_bounds__(256) void rowcopy_swz(const float4* __restrict__ src, float4* __restrict__ dst, int nrows, int R, int nblocks, int g) {
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  {
    const int n = nblocks, q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    if (g == 0) bid = (xcd < r) ? xcd * (q + 1) + idx : r * (q + 1) + (xcd - r) * q + idx;
    else if (g >= 2) { const int per = 8 * g, grp = idx / g, within = idx - grp * g, cand = grp * per + xcd * g + within; bid = cand < (n / per) * per ? cand : bid; }
  }
  const long long row0 = (long long)bid * R;
  for (int r = 0; r < R; ++r) {
    const long long row = row0 + r;
    if (row >= nrows) break;
    const float4* s = src + row * 768;
    float4* d = dst + row * 768;
    const float4 a = s[tid], b = s[tid + 256], c = s[tid + 512];
    d[tid] = a; d[tid + 256] = b; d[tid + 512] = c;
  }
}
template <int VPT, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void flatcopy(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f* s = reinterpret_cast<const v4f*>(src);
  v4f* d = reinterpret_cast<v4f*>(dst);
  const long long base = (long long)blockIdx.x * (256 * VPT) + threadIdx.x;
  v4f v[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) v[i] = NTL ? __builtin_nontemporal_load(s + base + 256 * i) : s[base + 256 * i];
#pragma unroll
  for (int i = 0; i < VPT; ++i) { if (NTS) __builtin_nontemporal_store(v[i], d + base + 256 * i); else d[base + 256 * i] = v[i]; }
}
// Locality probe: the 4 KB-per-workgroup nontemporal copy (the fastest, state-immune form) with the chunk a workgroup
// copies displaced from its dispatch slot.  mode 0: chunk = (slot + shift) mod n (shift 8 keeps "chunk mod 8 == XCD",
// shifts 1..7 break it); mode 1: XCD k owns the k-th contiguous eighth of the buffer; mode 2: source chunk displaced
// by `shift` from the destination chunk (reads and writes of a workgroup land on different 4 KB slots).
__global__ __launch_bounds__(256) void flatcopy_map(const float4* __restrict__ src, float4* __restrict__ dst, int nchunks,
                                                    int mode, int shift) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f* s = reinterpret_cast<const v4f*>(src);
  v4f* d = reinterpret_cast<v4f*>(dst);
  int slot = blockIdx.x, cs, cd;
  if (mode == 0) { cd = slot + shift; if (cd >= nchunks) cd -= nchunks; cs = cd; }
  else if (mode == 1) { const int q = nchunks >> 3; cd = (slot & 7) * q + (slot >> 3); cs = cd; }
  else { cd = slot; cs = slot + shift; if (cs >= nchunks) cs -= nchunks; }
  const v4f v = __builtin_nontemporal_load(s + (long long)cs * 256 + threadIdx.x);
  __builtin_nontemporal_store(v, d + (long long)cd * 256 + threadIdx.x);
}
static void run_map(const float4* src, float4* dst, size_t bytes, int mode, int shift, hipEvent_t e0, hipEvent_t e1) {
  std::vector<float> ts;
  const int nchunks = (int)(bytes / 4096);
  for (int it = 0; it < 12; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(flatcopy_map, dim3(nchunks), dim3(256), 0, 0, src, dst, nchunks, mode, shift);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("flat 4KB copy mode=%d shift=%2d: %.4f ms  %.2f TB/s\n", mode, shift, ts[ts.size() / 2], 2.0 * bytes / ts[ts.size() / 2] / 1e9);
}
// Tile probe: the traffic of a resample with ONE output row third (4 KB) per workgroup -- two 16-byte loads per thread
// (the same column third of source rows y and y+1: the second is the halo the next row's workgroup reads again) and
// one 16-byte store.  order 0: row-major linear; 1: a contiguous range of that order per XCD; 2: per XCD a contiguous
// range of rows, walked column third by column third in chunks of `chunk` rows (consecutive workgroups of an XCD read
// consecutive rows of one third: the halo comes from that XCD's L2).
__global__ __launch_bounds__(256) void tilecopy(const float4* __restrict__ src, float4* __restrict__ dst, int nrows,
                                                int order, int chunk, int nt) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int n = nrows * 3;
  int bid = blockIdx.x;
  long long y; int j;
  if (order >= 1) {
    const int q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r) ? xcd * (q + 1) + idx : r * (q + 1) + (xcd - r) * q + idx;
  }
  if (order <= 1) { y = bid / 3; j = bid - 3 * (int)y; }
  else {
    const int per = 3 * chunk, c = bid / per, w = bid - c * per;
    j = w / chunk; y = (long long)c * chunk + (w - j * chunk);
    if (y >= nrows) return;
  }
  const long long y1 = min(y + 1, (long long)nrows - 1);
  const v4f* s0 = reinterpret_cast<const v4f*>(src) + y * 768 + 256 * j + threadIdx.x;
  const v4f* s1 = reinterpret_cast<const v4f*>(src) + y1 * 768 + 256 * j + threadIdx.x;
  v4f a = (nt & 1) ? __builtin_nontemporal_load(s0) : *s0;
  const v4f b = *s1;
  if (b.x == 12345.678f) a.x += 1.0f;
  v4f* d = reinterpret_cast<v4f*>(dst) + y * 768 + 256 * j + threadIdx.x;
  if (nt & 2) __builtin_nontemporal_store(a, d); else *d = a;
}
static void run_tile(const float4* src, float4* dst, int nrows, int order, int chunk, int nt, hipEvent_t e0, hipEvent_t e1) {
  std::vector<float> ts;
  const int grid = order == 2 ? ((nrows + chunk - 1) / chunk) * chunk * 3 : nrows * 3;
  for (int it = 0; it < 12; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(tilecopy, dim3(grid), dim3(256), 0, 0, src, dst, nrows, order, chunk, nt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("tile copy (4 KB out + 2 x 4 KB in per workgroup) order=%d chunk=%3d nt=%d: %.4f ms  %.2f TB/s algorithmic\n", order, chunk, nt,
         ts[ts.size() / 2], 2.0 * nrows * 12288.0 / ts[ts.size() / 2] / 1e9);
}
// Occupancy probe: the nontemporal flat copy with a dynamic LDS allocation that limits the workgroups per CU
// (bytes in flight per CU = workgroups x 256 x VPT x 16).
template <int VPT>
__global__ __launch_bounds__(256) void flatcopy_occ(const float4* __restrict__ src, float4* __restrict__ dst) {
  extern __shared__ float4 pad_lds[];
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f* s = reinterpret_cast<const v4f*>(src);
  v4f* d = reinterpret_cast<v4f*>(dst);
  const long long base = (long long)blockIdx.x * (256 * VPT) + threadIdx.x;
  v4f v[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) v[i] = __builtin_nontemporal_load(s + base + 256 * i);
#pragma unroll
  for (int i = 0; i < VPT; ++i) __builtin_nontemporal_store(v[i], d + base + 256 * i);
}
template <int VPT>
static void run_occ(const float4* src, float4* dst, size_t bytes, int wgs_per_cu, hipEvent_t e0, hipEvent_t e1) {
  std::vector<float> ts;
  const long long n4 = (long long)bytes / 16;
  const size_t lds = wgs_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / wgs_per_cu) - 512;
  if (lds > 65536) hipFuncSetAttribute((const void*)flatcopy_occ<VPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int it = 0; it < 12; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((flatcopy_occ<VPT>), dim3((unsigned)(n4 / (256 * VPT))), dim3(256), lds, 0, src, dst);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("flat nt copy %d float4/thread, <= %d workgroups/CU (%3d KB in flight/CU): %.4f ms  %.2f TB/s\n", VPT, wgs_per_cu,
         wgs_per_cu * 4 * VPT, ts[ts.size() / 2], 2.0 * bytes / ts[ts.size() / 2] / 1e9);
}
template <int VPT, bool NTL, bool NTS>
static void run_flat(const float4* src, float4* dst, size_t bytes, hipEvent_t e0, hipEvent_t e1) {
  std::vector<float> ts;
  const long long n4 = (long long)bytes / 16;
  for (int it = 0; it < 12; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((flatcopy<VPT, NTL, NTS>), dim3((unsigned)(n4 / (256 * VPT))), dim3(256), 0, 0, src, dst, n4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("flat copy %2d float4/thread ntload=%d ntstore=%d: %.4f ms  %.2f TB/s\n", VPT, (int)NTL, (int)NTS, ts[ts.size() / 2], 2.0 * bytes / ts[ts.size() / 2] / 1e9);
}
int main() {
  const int B = 256, S = 1024;
  const long long nrows = (long long)B * S;
  const size_t bytes = (size_t)nrows * 3072 * 4;
  float4 *src, *dst;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
  hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 4; ++mode)
    for (int R : {1, 4, 32}) {
      std::vector<float> ts;
      for (int it = 0; it < 12; ++it) {
        hipEventRecord(e0);
        const int grid = (int)((nrows + R - 1) / R);
        if (mode == 0) hipLaunchKernelGGL(rowcopy<0>, dim3(grid), dim3(256), 0, 0, src, dst, (int)nrows, R);
        else if (mode == 1) hipLaunchKernelGGL(rowcopy<1>, dim3(grid), dim3(256), 0, 0, src, dst, (int)nrows, R);
        else if (mode == 2) hipLaunchKernelGGL(rowcopy<2>, dim3(grid), dim3(256), 0, 0, src, dst, (int)nrows, R);
        else hipLaunchKernelGGL(rowcopy<3>, dim3(grid), dim3(256), 0, 0, src, dst, (int)nrows, R);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 2) ts.push_back(ms);
      }
      std::sort(ts.begin(), ts.end());
      const char* names[4] = {"plain float4", "dword stores", "nt loads    ", "nt ld + st  "};
      printf("mode=%s R=%3d: %.4f ms  %.2f TB/s\n", names[mode], R, ts[ts.size() / 2], 2.0 * bytes / ts[ts.size() / 2] / 1e9);
    }
  for (int g : {0, 1, 4})
    for (int R : {3, 4, 8}) {
      std::vector<float> ts;
      const int grid = (int)((nrows + R - 1) / R);
      for (int it = 0; it < 12; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(rowcopy_swz, dim3(grid), dim3(256), 0, 0, src, dst, (int)nrows, R, grid, g);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 2) ts.push_back(ms);
      }
      std::sort(ts.begin(), ts.end());
      printf("rowcopy order=%s R=%d: %.4f ms  %.2f TB/s\n", g == 0 ? "xcd-contiguous" : (g == 1 ? "plain" : "groups-of-4"), R, ts[ts.size() / 2], 2.0 * bytes / ts[ts.size() / 2] / 1e9);
    }
  run_flat<4, false, false>(src, dst, bytes, e0, e1);
  run_flat<4, true, false>(src, dst, bytes, e0, e1);
  run_flat<4, false, true>(src, dst, bytes, e0, e1);
  run_flat<4, true, true>(src, dst, bytes, e0, e1);
  run_flat<1, true, true>(src, dst, bytes, e0, e1);
  run_flat<2, true, true>(src, dst, bytes, e0, e1);
  run_flat<8, true, true>(src, dst, bytes, e0, e1);
  run_flat<3, true, true>(src, dst, bytes, e0, e1);
  for (int nt : {0, 2, 3}) {
    run_tile(src, dst, (int)nrows, 0, 0, nt, e0, e1);
    run_tile(src, dst, (int)nrows, 1, 0, nt, e0, e1);
    for (int chunk : {8, 32, 128}) run_tile(src, dst, (int)nrows, 2, chunk, nt, e0, e1);
  }
  for (int w : {8, 6, 4, 3, 2, 1}) run_occ<1>(src, dst, bytes, w, e0, e1);
  for (int w : {8, 4, 3, 2, 1}) run_occ<2>(src, dst, bytes, w, e0, e1);
  for (int w : {8, 4, 2, 1}) run_occ<3>(src, dst, bytes, w, e0, e1);
  for (int w : {8, 4, 2, 1}) run_occ<4>(src, dst, bytes, w, e0, e1);
  for (int shift : {0, 8, 1}) run_map(src, dst, bytes, 0, shift, e0, e1);
  run_map(src, dst, bytes, 1, 0, e0, e1);
  for (int shift : {1, 24}) run_map(src, dst, bytes, 2, shift, e0, e1);
  return 0;
}
