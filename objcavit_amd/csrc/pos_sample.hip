// Positional-embedding samplers of GridRandomPositionalEmbeddings (reference modules/ObjCAViT.py:50-147) on the
// learnable table viewed as a gh x gw grid of E-vectors (row s = y * gw + x, reference :82-83 -- the table's own
// row-major [S][E] storage IS the channels-last grid, so a sample's E channels are one contiguous, coalesced run).
//
//   OCV_POS_CENTRE_OBJ  F.grid_sample(bilinear, zeros, align_corners=False) at (x / p0 * 2 - 1, y / p1 * 2 - 1)   :102-110
//   OCV_POS_CENTRE_IMG  the same sampler with the reference's token-indexed normalisation         :93-100
//   OCV_POS_ROI         torchvision.ops.ps_roi_align(output_size=[1,1], sampling_ratio=-1) of xywh boxes          :111-145
//
// One workgroup per coordinate row, one lane per channel (channels strided by the workgroup size).  The adaptive
// sample grid of the RoI mode, ceil(roi_h) x ceil(roi_w), is unbounded in the box size, but only samples inside
// [-1, H] x [-1, W] contribute: the loops run over that index window only (<= 2 (H + 1) + 3 by 2 (W + 1) + 3
// iterations whatever the box), the divisor stays the full count.  No host round trip, no data-dependent launch
// shape: capturable into a hipGraph.
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

// F.grid_sample, bilinear, padding_mode = zeros, align_corners = False, one (x, y) in normalised [-1, 1] coordinates
__device__ __forceinline__ void grid_sample_row(const float* __restrict__ table, int gh, int gw, int E, float xn, float yn,
                                                const float* __restrict__ add, float* __restrict__ out) {
  const float ix = ((xn + 1.0f) * (float)gw - 1.0f) * 0.5f;
  const float iy = ((yn + 1.0f) * (float)gh - 1.0f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  // corner weights as ATen computes them: (ix_se - ix) * (iy_se - iy) etc.
  const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix;
  const float wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
  // a NaN / huge coordinate fails every bounds test below -> 0, like ATen's within_bounds_2d on the converted index
  const bool okx0 = fx >= 0.0f && fx <= (float)(gw - 1), okx1 = fx + 1.0f >= 0.0f && fx + 1.0f <= (float)(gw - 1);
  const bool oky0 = fy >= 0.0f && fy <= (float)(gh - 1), oky1 = fy + 1.0f >= 0.0f && fy + 1.0f <= (float)(gh - 1);
  const int x0 = okx0 ? (int)fx : 0, x1 = okx1 ? (int)fx + 1 : 0;
  const int y0 = oky0 ? (int)fy : 0, y1 = oky1 ? (int)fy + 1 : 0;
  const float* r00 = table + ((size_t)y0 * gw + x0) * E;
  const float* r01 = table + ((size_t)y0 * gw + x1) * E;
  const float* r10 = table + ((size_t)y1 * gw + x0) * E;
  const float* r11 = table + ((size_t)y1 * gw + x1) * E;
  const float w00 = (oky0 && okx0) ? wx0 * wy0 : 0.0f, w01 = (oky0 && okx1) ? wx1 * wy0 : 0.0f;
  const float w10 = (oky1 && okx0) ? wx0 * wy1 : 0.0f, w11 = (oky1 && okx1) ? wx1 * wy1 : 0.0f;
  for (int c = threadIdx.x; c < E; c += blockDim.x) {
    float v = 0.0f;                      // nw, ne, sw, se: ATen's accumulation order
    if (oky0 && okx0) v += r00[c] * w00;
    if (oky0 && okx1) v += r01[c] * w01;
    if (oky1 && okx0) v += r10[c] * w10;
    if (oky1 && okx1) v += r11[c] * w11;
    out[c] = add ? add[c] + v : v;
  }
}

// torchvision bilinear_interpolate's per-axis preparation: sample coordinate -> (low index, high index, low weight
// complement).  Returns false when the coordinate is outside [-1, size] (the sample then contributes 0).
__device__ __forceinline__ bool roi_axis(float v, int size, int& lo, int& hi, float& frac) {
  if (v < -1.0f || v > (float)size) return false;
  if (v <= 0.0f) v = 0.0f;
  lo = (int)v;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    v = (float)lo;
  } else {
    hi = lo + 1;
  }
  frac = v - (float)lo;
  return true;
}

// first / last sample index whose coordinate start + (i + .5) * step can lie inside [-1, size], widened by one on
// each side (the exact test is repeated per sample with the reference's own expression)
__device__ __forceinline__ void roi_window(float start, float step, int n, int size, int& lo, int& hi) {
  float a = (-1.0f - start) / step - 0.5f, b = ((float)size - start) / step - 0.5f;
  a = fminf(fmaxf(floorf(a) - 1.0f, 0.0f), (float)n);
  b = fminf(fmaxf(ceilf(b) + 1.0f, -1.0f), (float)(n - 1));
  lo = (int)a;
  hi = (int)b;
}

__global__ void __launch_bounds__(128) pos_sample_kernel(const float* __restrict__ table, int gh, int gw, int E,
                                                         const float* __restrict__ coords, int coord_ld, int mode, float p0,
                                                         float p1, int rows_per_image, const float* __restrict__ addend,
                                                         float* __restrict__ out) {
  const int r = blockIdx.x;
  const float* cr = coords + (size_t)r * coord_ld;
  const float* add = addend ? addend + (size_t)r * E : nullptr;
  float* o = out + (size_t)r * E;
  if (mode == OCV_POS_CENTRE_OBJ) {
    // x over the image HEIGHT, y over the image WIDTH -- the reference's own normalisation (:104-105)
    grid_sample_row(table, gh, gw, E, ((cr[0] / p0) * 2.0f) - 1.0f, ((cr[1] / p1) * 2.0f) - 1.0f, add, o);
    return;
  }
  if (mode == OCV_POS_CENTRE_IMG) {
    // `norm_coords[:, 0]` / `[:, 1]` index dim 1 of a B x S x 2 tensor: TOKEN 0 has both components normalised by
    // the grid height, TOKEN 1 by the grid width, every other token keeps its raw coordinates (:95-96)
    const int s = r % rows_per_image;
    float x = cr[0], y = cr[1];
    if (s == 0) {
      x = ((x / p0) * 2.0f) - 1.0f;
      y = ((y / p0) * 2.0f) - 1.0f;
    } else if (s == 1) {
      x = ((x / p1) * 2.0f) - 1.0f;
      y = ((y / p1) * 2.0f) - 1.0f;
    }
    grid_sample_row(table, gh, gw, E, x, y, add, o);
    return;
  }
  // OCV_POS_ROI: xywh -> x1 y1 x2 y2, clamped at 0 from below (:115-124 / :134-143), then PS-RoI-align 1x1
  const float hw = cr[2] / 2.0f, hh = cr[3] / 2.0f;
  const float bx1 = fmaxf(cr[0] - hw, 0.0f), by1 = fmaxf(cr[1] - hh, 0.0f);
  const float bx2 = fmaxf(cr[0] + hw, 0.0f), by2 = fmaxf(cr[1] + hh, 0.0f);
  const float x1 = bx1 * p0 - 0.5f, y1 = by1 * p0 - 0.5f;
  const float rw = (bx2 * p0 - 0.5f) - x1, rh = (by2 * p0 - 0.5f) - y1;      // pooled 1x1: bin == roi
  const float cap = 16777216.0f;
  const float fnx = ceilf(rw), fny = ceilf(rh);
  const bool finite = fabsf(x1) < cap && fabsf(y1) < cap && fnx < cap && fny < cap && fnx == fnx && fny == fny;
  if (!finite) {
    for (int c = threadIdx.x; c < E; c += blockDim.x) o[c] = __builtin_nanf("");
    return;
  }
  const int nx = fnx > 0.0f ? (int)fnx : 0, ny = fny > 0.0f ? (int)fny : 0;
  const float count = (float)ny * (float)nx;      // 0 for a box without extent: 0 / 0 = NaN, like the published kernel
  int ylo = 0, yhi = -1, xlo = 0, xhi = -1;
  if (nx > 0 && ny > 0) {
    roi_window(y1, rh / (float)ny, ny, gh, ylo, yhi);
    roi_window(x1, rw / (float)nx, nx, gw, xlo, xhi);
  }
  for (int c = threadIdx.x; c < E; c += blockDim.x) {
    float acc = 0.0f;
    for (int iy = ylo; iy <= yhi; ++iy) {
      const float y = y1 + ((float)iy + 0.5f) * rh / (float)ny;
      int yl, yh;
      float ly;
      if (!roi_axis(y, gh, yl, yh, ly)) continue;
      const float hy = 1.0f - ly;
      for (int ix = xlo; ix <= xhi; ++ix) {
        const float x = x1 + ((float)ix + 0.5f) * rw / (float)nx;
        int xl, xh;
        float lx;
        if (!roi_axis(x, gw, xl, xh, lx)) continue;
        const float hx = 1.0f - lx;
        const float v1 = table[((size_t)yl * gw + xl) * E + c], v2 = table[((size_t)yl * gw + xh) * E + c];
        const float v3 = table[((size_t)yh * gw + xl) * E + c], v4 = table[((size_t)yh * gw + xh) * E + c];
        acc += hy * hx * v1 + hy * lx * v2 + ly * hx * v3 + ly * lx * v4;
      }
    }
    const float v = acc / count;
    o[c] = add ? add[c] + v : v;
  }
}

}  // namespace

extern "C" int ocv_pos_grid_sample_fwd(const float* table, int gh, int gw, int E, const float* coords, int coord_ld,
                                       int n_rows, int mode, float p0, float p1, int rows_per_image, const float* addend,
                                       float* out, ocv_stream_t stream) {
  OCV_CHECK_ARG(table && coords && out, "ocv_pos_grid_sample_fwd: null pointer");
  OCV_CHECK_ARG(gh >= 1 && gw >= 1 && E >= 1 && n_rows >= 0, "ocv_pos_grid_sample_fwd: bad sizes (gh=%d gw=%d E=%d rows=%d)", gh,
                gw, E, n_rows);
  OCV_CHECK_ARG(mode == OCV_POS_CENTRE_OBJ || mode == OCV_POS_CENTRE_IMG || mode == OCV_POS_ROI,
                "ocv_pos_grid_sample_fwd: unknown mode %d", mode);
  OCV_CHECK_ARG(coord_ld >= (mode == OCV_POS_ROI ? 4 : 2), "ocv_pos_grid_sample_fwd: coord_ld %d too small for mode %d", coord_ld,
                mode);
  OCV_CHECK_ARG(mode != OCV_POS_CENTRE_IMG || rows_per_image >= 1, "ocv_pos_grid_sample_fwd: rows_per_image must be >= 1");
  OCV_CHECK_ARG(mode == OCV_POS_CENTRE_IMG || (p0 == p0 && p0 != 0.0f), "ocv_pos_grid_sample_fwd: p0 must be non-zero");
  if (n_rows == 0) return 0;
  hipLaunchKernelGGL(pos_sample_kernel, dim3(n_rows), dim3(E >= 128 ? 128 : 64), 0, (hipStream_t)stream, table, gh, gw, E, coords,
                     coord_ld, mode, p0, p1, rows_per_image > 0 ? rows_per_image : 1, addend, out);
  OCV_CHECK_LAUNCH("ocv_pos_grid_sample_fwd");
  return 0;
}
