// 3 x 3 convolution (stride 1, zero padding 1) of an image with AT MOST FOUR channels into Cout channels, exact fp32 FMA,
// NHWC fp32 out, no bias / activation: the skip part of the LAST decoder stage of do_final_upscale models, whose skip tensor
// is the 3-channel input image (reference modules/DenseFeatureExtractor.py:99-101,116-117 -> :37-47; the other input-channel
// half of that convolution runs at the low resolution, csrc/tap_interp.hip, which adds this result, the bias and the
// activation).  On the matrix-core convolution kernel this layer is K = 9 x 4 (padded) on a pipeline built for K >= 288:
// 1.42 ms at 16 x 480 x 640 x 128, most of it prologue / epilogue.  Here it is what it is -- 27 multiply-adds per output and a
// 2.5 GB result to write: a thread owns 4 output channels (their 27 x 4 weights in registers) of one image column segment
// and slides a 3 x 3 x C window of scalars down the rows of the workgroup's image patch (staged once in LDS, zero padded); the
// 32 threads of a pixel broadcast-read the same few floats and write 512 contiguous bytes.  Bound by the write.
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

constexpr int FX = 8;      // pixels of a row per workgroup (8 x 32 channel groups = 256 threads)
constexpr int FY = 32;     // rows per workgroup

struct FewArgs {
  const float* x;          // element strides sb, sc, sy, sx
  const float* w;          // [9][NC][Cout]
  float* y;                // [B][H][W][Cout]
  long sb, sc, sy, sx;
  int H, W, Cout, ncb;     // ncb = 128-channel blocks of Cout
};

template <int NC>
__global__ __launch_bounds__(256) void conv_few_kernel(FewArgs p) {
  // the image patch under the workgroup's FX x FY outputs with its one-pixel halo, zero padded: the 32 threads of a pixel read
  // the same scalars -- from LDS a broadcast; as global loads they were one vector-memory instruction per scalar and wavefront
  // and the kernel ran at the texture addresser's pace (1.06 ms against 0.55 for the write alone)
  __shared__ float tile[FY + 2][NC][FX + 2];
  const int tid = threadIdx.x;
  const int cb = blockIdx.y % p.ncb, yb = blockIdx.y / p.ncb;
  const int co = cb * 128 + (tid & 31) * 4;
  const int px = tid >> 5;
  const int X0 = blockIdx.x * FX, X = X0 + px;
  const int Y0 = yb * FY;
  const long b = blockIdx.z;
  const float* xb = p.x + b * p.sb;
  for (int i = tid; i < (FY + 2) * NC * (FX + 2); i += 256) {
    const int j = i % (FX + 2), c = (i / (FX + 2)) % NC, r = i / ((FX + 2) * NC);
    const int yy = Y0 - 1 + r, xx = X0 - 1 + j;
    tile[r][c][j] = ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) ? xb[c * p.sc + (long)yy * p.sy + (long)xx * p.sx] : 0.f;
  }
  const bool live = co < p.Cout && X < p.W;
  float4 wv[9][NC];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int c = 0; c < NC; ++c) wv[t][c] = live ? ld4(p.w + ((long)t * NC + c) * p.Cout + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  if (!live) return;
  float win[3][3][NC];                                        // rows Y - 1, Y, Y + 1 of columns X - 1 .. X + 1
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) win[r][kx][c] = tile[r][c][px + kx];
    }
  }
  const int Y1 = min(Y0 + FY, p.H);
  for (int Y = Y0; Y < Y1; ++Y) {
    const int r2 = Y - Y0 + 2;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) win[2][kx][c] = tile[r2][c][px + kx];
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float v = win[ky][kx][c];
          const float4 wq = wv[3 * ky + kx][c];
          acc.x = fmaf(v, wq.x, acc.x);
          acc.y = fmaf(v, wq.y, acc.y);
          acc.z = fmaf(v, wq.z, acc.z);
          acc.w = fmaf(v, wq.w, acc.w);
        }
      }
    }
    *reinterpret_cast<float4*>(p.y + ((b * p.H + Y) * p.W + X) * p.Cout + co) = acc;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        win[0][kx][c] = win[1][kx][c];
        win[1][kx][c] = win[2][kx][c];
      }
    }
  }
}

}  // namespace

extern "C" int ocv_conv3x3_few_channels_fwd(const float* x, long stride_b, long stride_c, long stride_y, long stride_x,
                                            const float* w_taps, float* y, int B, int C, int H, int W, int Cout,
                                            ocv_stream_t stream) {
  OCV_CHECK_ARG(x && w_taps && y, "ocv_conv3x3_few_channels_fwd: null pointer");
  OCV_CHECK_ARG(C >= 1 && C <= 4, "ocv_conv3x3_few_channels_fwd: 1 <= C <= 4 input channels (got %d)", C);
  OCV_CHECK_ARG(B >= 1 && B <= 65535 && H >= 1 && W >= 1 && Cout >= 4 && Cout % 4 == 0,
                "ocv_conv3x3_few_channels_fwd: bad sizes (Cout must be a multiple of 4, got %d)", Cout);
  OCV_CHECK_ARG(ocv_aligned16(w_taps) && ocv_aligned16(y), "ocv_conv3x3_few_channels_fwd: w_taps / y must be 16-byte aligned");
  const int ncb = ocv_cdiv(Cout, 128);
  const long gy = (long)ocv_cdiv(H, FY) * ncb;
  OCV_CHECK_ARG(gy <= 65535, "ocv_conv3x3_few_channels_fwd: grid too large");
  FewArgs a{x, w_taps, y, stride_b, stride_c, stride_y, stride_x, H, W, Cout, ncb};
  const dim3 grid(ocv_cdiv(W, FX), (unsigned)gy, B), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (C) {
    case 1: hipLaunchKernelGGL(conv_few_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(conv_few_kernel<2>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(conv_few_kernel<3>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(conv_few_kernel<4>, grid, block, 0, st, a); break;
  }
  OCV_CHECK_LAUNCH("ocv_conv3x3_few_channels_fwd");
  return 0;
}
