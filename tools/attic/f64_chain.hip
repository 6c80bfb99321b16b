// Dev tool: latency of a dependent chain of v_add_f64 / v_add_f32 in ONE wave that has its SIMD to itself, and with
// 1..7 other waves of the same workgroup spinning on independent VALU work on the same SIMDs (what a one-lane cumsum sees
// inside the one-launch steps).  hipcc --offload-arch=gfx950 -O3 -o tools/f64_chain tools/f64_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ void chain(T* out, unsigned long long* cyc, int n, int busy_waves) {
  const int wave = threadIdx.x / 64;
  if (wave == 0) {
    T c = (T)threadIdx.x, x = (T)1.000001;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 8) {
      c = c + x; c = c + x; c = c + x; c = c + x; c = c + x; c = c + x; c = c + x; c = c + x;
      asm volatile("" : "+v"(c));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  } else if (wave <= busy_waves) {
    float a = threadIdx.x, b = 1.0001f, d = 0.5f, e = 0.25f;
    for (int i = 0; i < n * 6; ++i) { a = a * b + d; d = d * b + e; e = e * b + a; asm volatile("" : "+v"(a), "+v"(d), "+v"(e)); }
    out[threadIdx.x] = (T)(a + d + e);
  }
}
template <typename T>
void run(const char* name) {
  T* out; unsigned long long* cyc;
  hipMalloc(&out, 2048 * sizeof(T)); hipMalloc(&cyc, 8);
  const int n = 1 << 16;
  for (int waves : {1, 4, 8, 16, 32}) {       // waves per workgroup: 1 = the chain alone; 32 = 8 per SIMD
    const int nt = waves * 64 > 1024 ? 1024 : waves * 64;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(chain<T>, dim3(1), dim3(nt), 0, 0, out, cyc, n, nt / 64 - 1);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s dependent adds, %2d waves in the workgroup (%d on the chain's SIMD): %.1f cycles per add\n", name, nt / 64, (nt / 64 + 3) / 4, (double)h / n);
  }
}
int main() { run<double>("v_add_f64"); run<float>("v_add_f32"); return 0; }
