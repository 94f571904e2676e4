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

// ---------------------------------------------------------------------------
// The bin regressor AND the launch above in one: token 0 of an image -> Linear + LeakyReLU -> Linear + LeakyReLU -> Linear -> normalise ->
// widths, edges, centres (modules/miniViT.py:33-42 == modules/ObjCAViT.py:373-388, then AdaBins.py:79-83 == GraphBins.py:111-115).  One
// row per image: as three GEMM launches of TWO workgroups each (16 rows x 256 columns) + the launch above these were 4 x 20 - 40 us at
// the END of the token chain, i.e. on the forward's critical path whatever the batch.  One workgroup per image, one output unit per
// thread, plain fp32 FMA chains over contiguous weight rows (640 KB per workgroup from L2), activations in LDS.
// ---------------------------------------------------------------------------
constexpr int RB_MAX_DIM = 1024;

struct RBArgs {
  const float* x;          // token rows: image b's row at x + b * x_stride
  long x_stride;
  const float *w1, *b1, *w2, *b2, *w3, *b3;      // row-major [out][in]
  int E, H1, H2, n, mode;
  float min_depth, max_depth, slope;
  float *widths, *edges, *centers;
};

__device__ __forceinline__ float rb_dot(const float* __restrict__ w, const float* xs, int K, float acc) {
  for (int k = 0; k < K; k += 4) {
    const float4 q = ld4(w + k);
    acc = fmaf(q.x, xs[k], acc);
    acc = fmaf(q.y, xs[k + 1], acc);
    acc = fmaf(q.z, xs[k + 2], acc);
    acc = fmaf(q.w, xs[k + 3], acc);
  }
  return acc;
}

__global__ __launch_bounds__(256) void regressor_bins_kernel(RBArgs p) {
  __shared__ __attribute__((aligned(16))) float xa[RB_MAX_DIM], xb[RB_MAX_DIM];
  __shared__ float v[BE_MAX_BINS];
  __shared__ double part[256];
  const int tid = threadIdx.x, n = p.n;
  const long b = blockIdx.x;
  for (int k = tid; k < p.E; k += 256) xa[k] = p.x[b * p.x_stride + k];
  __syncthreads();
  for (int o = tid; o < p.H1; o += 256) {
    const float a = rb_dot(p.w1 + (long)o * p.E, xa, p.E, p.b1[o]);
    xb[o] = a > 0.f ? a : p.slope * a;
  }
  __syncthreads();
  for (int o = tid; o < p.H2; o += 256) {
    const float a = rb_dot(p.w2 + (long)o * p.H1, xb, p.H1, p.b2[o]);
    xa[o] = a > 0.f ? a : p.slope * a;
  }
  __syncthreads();
  for (int o = tid; o < n; o += 256) v[o] = rb_dot(p.w3 + (long)o * p.H2, xa, p.H2, p.b3[o]);
  __syncthreads();
  // from here on: bin_edges_kernel on the row in v
  const int per = (n + 255) / 256, lo = tid * per, hi = min(n, lo + per);
  double s = 0.0;
  for (int i = lo; i < hi; ++i) {
    float y = v[i];
    if (p.mode == OCV_BINNORM_LINEAR) y = fmaxf(y, 0.f) + 0.1f;
    else if (p.mode == OCV_BINNORM_SIGMOID) y = 1.0f / (1.0f + expf(-y));
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
  const float scale = p.max_depth - p.min_depth;
  double run = 0.0;
  for (int i = lo; i < hi; ++i) {
    const float w = p.mode == OCV_BINNORM_NONE ? v[i] : v[i] / total;
    p.widths[b * n + i] = w;
    const float e = scale * w;
    v[i] = e;
    run += (double)e;
  }
  part[tid] = run;
  __syncthreads();
  if (tid == 0) {
    double acc = (double)p.min_depth;
    for (int t = 0; t < 256; ++t) {
      const double c = part[t];
      part[t] = acc;
      acc += c;
    }
  }
  __syncthreads();
  double e0 = part[tid];
  if (tid == 0) p.edges[b * (n + 1)] = p.min_depth;
  for (int i = lo; i < hi; ++i) {
    const double e1 = e0 + (double)v[i];
    p.edges[b * (n + 1) + i + 1] = (float)e1;
    p.centers[b * n + i] = 0.5f * ((float)e0 + (float)e1);
    e0 = e1;
  }
}

}  // namespace

extern "C" int ocv_regressor_bins_fwd(const float* x, long x_stride, const float* w1, const float* b1, const float* w2, const float* b2,
                                      const float* w3, const float* b3, int E, int H1, int H2, int n_bins, float leaky_slope, int mode,
                                      float min_depth, float max_depth, float* widths_normed, float* edges, float* centers, int B,
                                      ocv_stream_t stream) {
  OCV_CHECK_ARG(x && w1 && b1 && w2 && b2 && w3 && b3 && widths_normed && edges && centers, "ocv_regressor_bins_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && n_bins >= 1 && n_bins <= BE_MAX_BINS, "ocv_regressor_bins_fwd: bad sizes (n_bins <= %d, got %d)", BE_MAX_BINS, n_bins);
  OCV_CHECK_ARG(E >= 4 && H1 >= 4 && H2 >= 4 && E <= RB_MAX_DIM && H1 <= RB_MAX_DIM && H2 <= RB_MAX_DIM && E % 4 == 0 && H1 % 4 == 0 && H2 % 4 == 0,
                "ocv_regressor_bins_fwd: E, H1, H2 must be multiples of 4, at most %d (got %d, %d, %d)", RB_MAX_DIM, E, H1, H2);
  OCV_CHECK_ARG(x_stride >= E, "ocv_regressor_bins_fwd: x_stride below E");
  OCV_CHECK_ARG(ocv_aligned16(w1) && ocv_aligned16(w2) && ocv_aligned16(w3), "ocv_regressor_bins_fwd: weights must be 16-byte aligned");
  OCV_CHECK_ARG(mode == OCV_BINNORM_LINEAR || mode == OCV_BINNORM_SIGMOID || mode == OCV_BINNORM_NONE, "ocv_regressor_bins_fwd: unknown mode %d", mode);
  RBArgs a{x, x_stride, w1, b1, w2, b2, w3, b3, E, H1, H2, n_bins, mode, min_depth, max_depth, leaky_slope, widths_normed, edges, centers};
  hipLaunchKernelGGL(regressor_bins_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
  OCV_CHECK_LAUNCH("ocv_regressor_bins_fwd");
  return 0;
}

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
