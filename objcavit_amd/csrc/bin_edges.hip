// Bin-width normalisation, bin edges and bin centres of one batch in ONE launch: the tail of mViT / ObjCAViT.forward
// (modules/miniViT.py:33-42 == modules/ObjCAViT.py:378-388: relu + 0.1 | sigmoid, divide by the row sum) and the glue of
// AdaBins / GraphBins.forward (modules/AdaBins.py:79-83 == modules/GraphBins.py:111-115: scale by (max - min), prepend min_depth,
// cumsum, centres = mean of neighbouring edges).  257 floats per image: as ~10 ATen launches (relu, add, sum, div, mul, pad,
// cumsum, two slices + add + mul) this was 0.05 ms of a 4 ms bs-1 forward; one workgroup per image does it in one.
// Fixed order: the row sum is a sequential sum per thread chunk + a tree over 256 partials, the cumsum a sequential scan per
// chunk + a scan over the chunk totals -- both carried in DOUBLE (257 values: free) and rounded once, so edges and centres are
// within half an ulp of the exact prefix sums of the fp32 widths (a 256-term fp32 chain drifts by ~1e-6 of max_depth).
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

constexpr int BE_MAX_BINS = 4096;

__global__ __launch_bounds__(256) void bin_edges_kernel(const float* __restrict__ raw, int mode, float min_depth, float max_depth,
                                                        float* __restrict__ widths, float* __restrict__ edges,
                                                        float* __restrict__ centers, int n) {
  __shared__ float v[BE_MAX_BINS];
  __shared__ double part[256];
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  const int per = (n + 255) / 256, lo = tid * per, hi = min(n, lo + per);
  double s = 0.0;
  for (int i = lo; i < hi; ++i) {
    float y = raw[b * n + i];
    if (mode == OCV_BINNORM_LINEAR) y = fmaxf(y, 0.f) + 0.1f;
    else if (mode == OCV_BINNORM_SIGMOID) y = 1.0f / (1.0f + expf(-y));
    v[i] = y;
    s += y;
  }
  part[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) part[tid] += part[tid + o];
    __syncthreads();
  }
  const float total = (float)part[0];
  __syncthreads();
  const float scale = max_depth - min_depth;
  double run = 0.0;
  for (int i = lo; i < hi; ++i) {
    const float w = mode == OCV_BINNORM_NONE ? v[i] : v[i] / total;
    widths[b * n + i] = w;
    const float e = scale * w;
    v[i] = e;
    run += (double)e;
  }
  part[tid] = run;
  __syncthreads();
  if (tid == 0) {                                     // exclusive scan of the 256 chunk totals, in order
    double acc = (double)min_depth;
    for (int t = 0; t < 256; ++t) {
      const double c = part[t];
      part[t] = acc;
      acc += c;
    }
  }
  __syncthreads();
  double e0 = part[tid];                              // edge in front of this thread's first bin
  if (tid == 0) edges[b * (n + 1)] = min_depth;
  for (int i = lo; i < hi; ++i) {
    const double e1 = e0 + (double)v[i];
    edges[b * (n + 1) + i + 1] = (float)e1;
    centers[b * n + i] = 0.5f * ((float)e0 + (float)e1);
    e0 = e1;
  }
}

}  // namespace

extern "C" int ocv_bin_edges_fwd(const float* raw, int mode, float min_depth, float max_depth, float* widths_normed, float* edges,
                                 float* centers, int B, int n_bins, ocv_stream_t stream) {
  OCV_CHECK_ARG(raw && widths_normed && edges && centers, "ocv_bin_edges_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && n_bins >= 1 && n_bins <= BE_MAX_BINS, "ocv_bin_edges_fwd: bad sizes (n_bins <= %d, got %d)", BE_MAX_BINS, n_bins);
  OCV_CHECK_ARG(mode == OCV_BINNORM_LINEAR || mode == OCV_BINNORM_SIGMOID || mode == OCV_BINNORM_NONE, "ocv_bin_edges_fwd: unknown mode %d", mode);
  hipLaunchKernelGGL(bin_edges_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, raw, mode, min_depth, max_depth,
                     widths_normed, edges, centers, n_bins);
  OCV_CHECK_LAUNCH("ocv_bin_edges_fwd");
  return 0;
}
