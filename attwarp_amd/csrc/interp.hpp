// np.interp restated for the device (shared by axis.hip and profiles.hip).
#pragma once
#include "common.hpp"

namespace attwarp {

// ---- np.interp, exact restatement --------------------------------------------------
// numpy/_core/src/multiarray/compiled_base.c : binary_search_with_guess + arr_interp.
// xp: knots (double, LDS), fp[j] = j (the reference's *_orig_map_fwd = [0,1,..,L]).
constexpr int LIKELY_IN_CACHE_SIZE = 8;

__device__ inline int np_search_with_guess(double key, const double* arr, int len, int guess) {
  int imin = 0, imax = len;
  if (key > arr[len - 1]) return len;
  if (key < arr[0]) return -1;
  if (len <= 4) {
    int i;
    for (i = 1; i < len && key >= arr[i]; ++i) {}
    return i - 1;
  }
  if (guess > len - 3) guess = len - 3;
  if (guess < 1) guess = 1;
  if (key < arr[guess]) {
    if (key < arr[guess - 1]) {
      imax = guess - 1;
      if (guess > LIKELY_IN_CACHE_SIZE && key >= arr[guess - LIKELY_IN_CACHE_SIZE]) imin = guess - LIKELY_IN_CACHE_SIZE;
    } else {
      return guess - 1;
    }
  } else {
    if (key < arr[guess + 1]) return guess;
    if (key < arr[guess + 2]) return guess + 1;
    imin = guess + 2;
    if (guess < len - LIKELY_IN_CACHE_SIZE - 1 && key < arr[guess + LIKELY_IN_CACHE_SIZE]) imax = guess + LIKELY_IN_CACHE_SIZE;
  }
  while (imin < imax) {
    const int imid = imin + ((imax - imin) >> 1);
    if (key >= arr[imid]) imin = imid + 1; else imax = imid;
  }
  return imin - 1;
}

__device__ __forceinline__ double np_interp_eval(double x, int j, const double* xp, int len) {
  // fp[j] = j
  if (j == -1) return 0.0;
  if (j == len) return (double)(len - 1);
  if (j == len - 1) return (double)j;
  if (xp[j] == x) return (double)j;
  const double slope = 1.0 / (xp[j + 1] - xp[j]);      // (fp[j+1]-fp[j]) / (xp[j+1]-xp[j])
  double r = slope * (x - xp[j]) + (double)j;
  if (isnan(r)) {
    r = slope * (x - xp[j + 1]) + (double)(j + 1);
    // fp[j] != fp[j+1] always here, so numpy's last fallback never applies
  }
  return r;
}

// All threads of the block: map[i] = float(np.interp(i, xp, arange(len))) for i < n_out.
// xp must be in LDS and visible (caller synchronised).  `mono` = xp is non-decreasing (block uniform).
__device__ inline void np_interp_block(const double* xp, int len, int n_out, float* map, bool mono) {
  if (mono) {
    // for a sorted array every path of the guess search returns (number of knots <= x) - 1
    for (int i = threadIdx.x; i < n_out; i += blockDim.x) {
      const double x = (double)i;
      int j;
      if (x > xp[len - 1]) j = len;
      else if (x < xp[0]) j = -1;
      else {
        int lo = 0, hi = len;
        while (lo < hi) {
          const int mid = lo + ((hi - lo) >> 1);
          if (xp[mid] <= x) lo = mid + 1; else hi = mid;
        }
        j = lo - 1;
      }
      map[i] = (float)np_interp_eval(x, j, xp, len);
    }
  } else if (threadIdx.x == 0) {
    // unsorted / NaN knots: numpy's result depends on the search path, replay it exactly
    int j = 0;
    for (int i = 0; i < n_out; ++i) {
      const double x = (double)i;
      j = np_search_with_guess(x, xp, len, j);
      map[i] = (float)np_interp_eval(x, j, xp, len);
    }
  }
}

__device__ __forceinline__ bool block_is_sorted(const double* xp, int len) {
  int ok = 1;
  for (int k = threadIdx.x; k + 1 < len; k += blockDim.x) ok &= (xp[k + 1] >= xp[k]);
  return __syncthreads_and(ok) != 0;
}


}  // namespace attwarp
