// Validation-step arithmetic on the device, one pass (row N2 of SURVEY.md section 8f): what the reference does between
// the model output and its eight logged numbers --
//   flip-TTA average 0.5 (clamp(d) + clamp(flip(d_mirror)))                  modules/GraphBinsLM.py:159-181
//   bilinear align_corners resize to the ground-truth size, nan -> min_depth,
//   +-inf -> max_depth, validity mask min < gt <= max, Garg / Eigen crop      metrics/MetricsPreprocess.py:14-45
//   abs_rel, sq_rel, rmse, rmse_log, log10, delta 1.25 / 1.25^2 / 1.25^3      metrics/AbsRel.py:44-52, SqRel.py:45-52,
//                                                                            RMSE.py:48-55, RMSELog.py:45-52,
//                                                                            Log10.py:52-61, AccThresh.py:59-66
// -- as ONE record of 10 floats per image (objcavit_amd/dp.py RECORD_FIELDS), so that a data-parallel job needs a
// single all-gather (the reference: ~10 element-wise passes over B x H x W, a boolean gather, 16 torchmetrics states
// and 32 scalar collectives).  The resized prediction is never materialised: a thread evaluates the four low-resolution
// taps of its pixel (clamped, averaged with the mirrored tap) straight from the two model outputs.
// Two stages, fixed order, no float atomics: (tiles, B) workgroups reduce 9 double-precision sums each, a second tiny
// launch adds the tiles in order and finishes the means / square roots.  HBM-bound: 4 B of ground truth per pixel.
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

constexpr int NSUM = 9;           // abs_rel, sq_rel, sq, sq_log, log10, d1, d2, d3, count

struct MetArgs {
  const float *pred, *mirror, *gt;
  double* part;                   // [B][tiles][NSUM]
  int h, w, H, W, y0, y1, x0, x1, tiles;
  float sh, sw, dmin, dmax;
};

// torch.clamp semantics: NaN stays NaN (fminf / fmaxf would drop it)
__device__ __forceinline__ float clamp_keep_nan(float v, float lo, float hi) { return v != v ? v : fminf(fmaxf(v, lo), hi); }

__device__ __forceinline__ float tap(const MetArgs& p, const float* pb, const float* mb, int y, int x) {
  const float a = clamp_keep_nan(pb[y * p.w + x], p.dmin, p.dmax);
  if (mb == nullptr) return a;
  return 0.5f * (a + clamp_keep_nan(mb[y * p.w + (p.w - 1 - x)], p.dmin, p.dmax));
}

__global__ __launch_bounds__(256) void depth_metrics_partial_kernel(MetArgs p) {
  __shared__ double red[NSUM][4];
  const int tid = threadIdx.x, tile = blockIdx.x;
  const long b = blockIdx.y;
  const long P = (long)p.H * p.W;
  const long per = (P + p.tiles - 1) / p.tiles;
  const long lo = tile * per, hi = min(P, lo + per);
  const float* pb = p.pred + b * (long)p.h * p.w;
  const float* mb = p.mirror != nullptr ? p.mirror + b * (long)p.h * p.w : nullptr;
  const float* gb = p.gt + b * P;
  float s[NSUM];
#pragma unroll
  for (int i = 0; i < NSUM; ++i) s[i] = 0.f;
  double acc[NSUM];
#pragma unroll
  for (int i = 0; i < NSUM; ++i) acc[i] = 0.0;
  int pending = 0;
#pragma unroll 4
  for (long pix = lo + tid; pix < hi; pix += 256) {
    const float g = gb[pix];
    const int Y = (int)(pix / p.W), X = (int)(pix - (long)Y * p.W);
    const bool valid = g > p.dmin && g <= p.dmax && Y >= p.y0 && Y < p.y1 && X >= p.x0 && X < p.x1;
    if (valid) {
      // ATen upsample_bilinear2d, align_corners = True
      const float sy = p.sh * Y, sx = p.sw * X;
      const int ya = (int)sy, xa = (int)sx;
      const int yb = ya + (ya < p.h - 1 ? 1 : 0), xb = xa + (xa < p.w - 1 ? 1 : 0);
      const float h1 = sy - (float)ya, h0 = 1.0f - h1, w1 = sx - (float)xa, w0 = 1.0f - w1;
      // (all four terms always, so a NaN tap reaches its neighbours through a zero weight exactly as in ATen; equal
      // sizes are ATen's identity short-cut, where it does not)
      float v = (p.h == p.H && p.w == p.W)
                    ? tap(p, pb, mb, Y, X)
                    : h0 * (w0 * tap(p, pb, mb, ya, xa) + w1 * tap(p, pb, mb, ya, xb)) +
                          h1 * (w0 * tap(p, pb, mb, yb, xa) + w1 * tap(p, pb, mb, yb, xb));
      if (v != v) v = p.dmin;                                   // nan_to_num(nan = min, posinf = neginf = max)
      else if (__builtin_isinf(v)) v = p.dmax;
      const float d = g - v, ratio = fmaxf(g / v, v / g);
      const float dl = logf(g) - logf(v);
      s[0] += fabsf(d) / g;
      s[1] += d * d / g;
      s[2] += d * d;
      s[3] += dl * dl;
      s[4] += fabsf(log10f(g) - log10f(v));
      s[5] += ratio < 1.25f ? 1.f : 0.f;
      s[6] += ratio < 1.25f * 1.25f ? 1.f : 0.f;
      s[7] += ratio < 1.25f * 1.25f * 1.25f ? 1.f : 0.f;
      s[8] += 1.f;
      if (++pending == 64) {                                    // short fp32 runs, double-precision totals
#pragma unroll
        for (int i = 0; i < NSUM; ++i) { acc[i] += (double)s[i]; s[i] = 0.f; }
        pending = 0;
      }
    }
  }
  // workgroup totals: a fixed xor-tree over each wavefront's 64 lanes, then the four wavefronts in order (one thread walking 256 LDS
  // doubles per sum was a third of the launch)
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < NSUM; ++i) {
    double t = acc[i] + (double)s[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0) red[i][wave] = t;
  }
  __syncthreads();
  if (tid < NSUM) p.part[((b * p.tiles) + tile) * NSUM + tid] = ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3];
}

// one workgroup per image, one wavefront per sum: lanes stride over the tiles, then a fixed xor-tree adds the 64 lanes
__global__ __launch_bounds__(64 * NSUM) void depth_metrics_finish_kernel(const double* __restrict__ part, int tiles,
                                                                       float* __restrict__ rec, int B, long first_id) {
  __shared__ double tot[NSUM];
  const int b = blockIdx.x, i = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.0;
  for (int t = lane; t < tiles; t += 64) s += part[((long)b * tiles + t) * NSUM + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) tot[i] = s;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double n = tot[8] > 0.0 ? tot[8] : 1.0;
  float* r = rec + (long)b * 10;
  r[0] = (float)(tot[0] / n);
  r[1] = (float)(tot[1] / n);
  r[2] = (float)sqrt(tot[2] / n);
  r[3] = (float)sqrt(tot[3] / n);
  r[4] = (float)(tot[4] / n);
  r[5] = (float)(tot[5] / n);
  r[6] = (float)(tot[6] / n);
  r[7] = (float)(tot[7] / n);
  r[8] = (float)tot[8];
  r[9] = (float)(first_id + b);
}

int metric_tiles(int B, long P) {
  long t = (2048 + B - 1) / B;                         // ~2048 workgroups per launch
  const long maxt = (P + 4095) / 4096;                 // at least 16 pixels per thread
  if (t > maxt) t = maxt;
  return (int)(t < 1 ? 1 : t);
}

}  // namespace

extern "C" size_t ocv_depth_metrics_workspace_bytes(int B, int H, int W) {
  if (B < 1 || H < 1 || W < 1) return 0;
  return (size_t)B * metric_tiles(B, (long)H * W) * NSUM * sizeof(double);
}

extern "C" int ocv_depth_metrics_fwd(const float* pred, const float* pred_mirror, int h, int w, const float* gt, int H, int W,
                                     float min_depth, float max_depth, int crop_y0, int crop_y1, int crop_x0, int crop_x1,
                                     long first_image_id, float* records, int B, void* workspace, size_t workspace_bytes,
                                     ocv_stream_t stream) {
  OCV_CHECK_ARG(pred && gt && records && workspace, "ocv_depth_metrics_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && B <= 65535 && h >= 1 && w >= 1 && H >= 1 && W >= 1, "ocv_depth_metrics_fwd: bad sizes");
  OCV_CHECK_ARG(min_depth < max_depth, "ocv_depth_metrics_fwd: min_depth must be below max_depth");
  OCV_CHECK_ARG(crop_y0 >= 0 && crop_y0 <= crop_y1 && crop_y1 <= H && crop_x0 >= 0 && crop_x0 <= crop_x1 && crop_x1 <= W,
                "ocv_depth_metrics_fwd: crop box outside the ground-truth map (pass 0, H, 0, W for no crop)");
  OCV_CHECK_ARG(workspace_bytes >= ocv_depth_metrics_workspace_bytes(B, H, W) && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
                "ocv_depth_metrics_fwd: workspace too small or misaligned");
  const int tiles = metric_tiles(B, (long)H * W);
  MetArgs a{pred, pred_mirror, gt, (double*)workspace, h, w, H, W, crop_y0, crop_y1, crop_x0, crop_x1, tiles,
            H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f, min_depth, max_depth};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(depth_metrics_partial_kernel, dim3(tiles, B), dim3(256), 0, st, a);
  OCV_CHECK_LAUNCH("ocv_depth_metrics_fwd(partial)");
  hipLaunchKernelGGL(depth_metrics_finish_kernel, dim3(B), dim3(64 * NSUM), 0, st, (const double*)workspace, tiles, records, B,
                     first_image_id);
  OCV_CHECK_LAUNCH("ocv_depth_metrics_fwd(finish)");
  return 0;
}
