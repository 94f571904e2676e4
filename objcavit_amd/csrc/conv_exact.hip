// Exact-fp32 convolution (k x k, k odd, stride 1, zero "same" padding) on NHWC activations for gfx950: the implicit GEMM
// of csrc/conv_igemm.hip on v_mfma_f32_32x32x2_f32 -- bit-for-bit a k-ordered fp32 fma chain per output, no split
// operands -- for callers that want the reference's own arithmetic class from a hand-written kernel: the A/B numerics
// route of the decoder / head convolutions (OCV_CONV=exact; round 1 used MIOpen for that) and the convolution shapes
// the split-bf16 kernels do not take (channel counts that are not multiples of 4).  5x slower than the split-bf16
// kernel by construction (64 instead of 3 x 4 matrix-pipe cycles per 32 x 32 x 2 block); not on any default path.
//
// GEMM view as in conv_igemm.hip: M = B*H*W pixels, N = Cout, K = taps x (C1 + C2) with x2 a virtual channel concat.
// Tile 32 pixels x 128 channels per workgroup (4 wavefronts x 32 channels), K advances 32 channels of one tap per step
// through LDS tiles with rows padded to 33 floats (conflict-free per-lane ds_read_b32 of the MFMA operands).
// Weights: fp32 [taps][Cout][Cin] (tap-major re-layout of nn.Conv2d's [Cout][Cin][k][k], done once by the caller).
// Replaces the same nn.Conv2d (+ folded BatchNorm + activation) as ocv_conv_nhwc_fwd
// (modules/DenseFeatureExtractor.py:37-42,97; modules/ObjCAViT.py:298,374; modules/miniViT.py:15,25).
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

constexpr int XM = 32, XN = 128, XK = 32, XLD = XK + 1;

struct ExArgs {
  const float *x1, *x2, *w, *bias, *res;
  float* y;
  int C1, C2, Cin, Cout, H, W, ks, act;
  long M;
};

__device__ __forceinline__ float ex_act(float v, int act) {
  if (act == OCV_ACT_LEAKY_RELU) return v > 0.f ? v : 0.01f * v;
  if (act == OCV_ACT_SILU) return fast_silu(v);
  if (act == OCV_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

__global__ __launch_bounds__(256) void conv_exact_kernel(ExArgs p) {
  __shared__ float As[XM][XLD];
  __shared__ float Ws[XN][XLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const long m0 = (long)blockIdx.x * XM;
  const int n0 = blockIdx.y * XN;
  const int taps = p.ks * p.ks, pad = p.ks >> 1;
  const int nchunk = (p.Cin + XK - 1) / XK;

  // A role: thread -> (pixel tid / 8, 4 consecutive channels of the chunk)
  const int ar = tid >> 3, ak = (tid & 7) * 4;
  const long am = m0 + ar;
  const bool aok = am < p.M;
  int ay = 0, ax = 0;
  long apix = 0;
  if (aok) {
    const long hw = (long)p.H * p.W, rem = am % hw;
    ay = (int)(rem / p.W);
    ax = (int)(rem - (long)ay * p.W);
    apix = am;
  }

  f32x16 acc = {0};
  for (int c0 = 0; c0 < nchunk * XK; c0 += XK) {
    for (int t = 0; t < taps; ++t) {
      const int dy = t / p.ks - pad, dx = t % p.ks - pad;
      {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const bool inb = aok && (unsigned)(ay + dy) < (unsigned)p.H && (unsigned)(ax + dx) < (unsigned)p.W;
        if (inb) {
          const long px = apix + (long)dy * p.W + dx;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = c0 + ak + e;
            if (c < p.C1) v[e] = p.x1[px * p.C1 + c];
            else if (c < p.Cin) v[e] = p.x2[px * p.C2 + (c - p.C1)];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) As[ar][ak + e] = v[e];
      }
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int r = pass * 32 + (tid >> 3), n = n0 + r;
        const float* src = p.w + ((long)t * p.Cout + (n < p.Cout ? n : 0)) * p.Cin;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = c0 + ak + e;
          Ws[r][ak + e] = (n < p.Cout && c < p.Cin) ? src[c] : 0.f;
        }
      }
      __syncthreads();
      const float* arow = &As[l31][hh];
      const float* wrow = &Ws[wave * 32 + l31][hh];
#pragma unroll
      for (int s = 0; s < XK / 2; ++s) acc = mfma_32x32x2(arow[2 * s], wrow[2 * s], acc);
      __syncthreads();
    }
  }
  const int n = n0 + wave * 32 + l31;
  if (n >= p.Cout) return;
  const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long m = m0 + acc_row(r, hh);
    if (m < p.M) {
      float v = ex_act(acc[r] + bv, p.act);
      if (p.res != nullptr) v += p.res[m * p.Cout + n];
      p.y[m * p.Cout + n] = v;
    }
  }
}

}  // namespace

extern "C" int ocv_conv_nhwc_exact_fwd(const float* x1, int C1, const float* x2, int C2, const float* w_tap_major,
                                       const float* bias, const float* residual, float* y, int B, int H, int W, int Cout,
                                       int ksize, int act, ocv_stream_t stream) {
  OCV_CHECK_ARG(x1 && w_tap_major && y, "ocv_conv_nhwc_exact_fwd: null pointer");
  OCV_CHECK_ARG(ksize >= 1 && ksize <= 7 && (ksize & 1) == 1, "ocv_conv_nhwc_exact_fwd: kernel size must be odd and <= 7 (got %d)", ksize);
  OCV_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && Cout >= 1 && C1 >= 1 && (x2 == nullptr || C2 >= 1), "ocv_conv_nhwc_exact_fwd: bad sizes");
  OCV_CHECK_ARG(act >= 0 && act <= 3, "ocv_conv_nhwc_exact_fwd: unknown activation %d", act);
  ExArgs a{x1, x2, w_tap_major, bias, residual, y, C1, x2 ? C2 : 0, C1 + (x2 ? C2 : 0), Cout, H, W, ksize, act, (long)B * H * W};
  OCV_CHECK_ARG(ocv_cdiv(Cout, XN) <= 65535, "ocv_conv_nhwc_exact_fwd: too many output channels");
  hipLaunchKernelGGL(conv_exact_kernel, dim3((unsigned)ocv_cdiv(a.M, XM), ocv_cdiv(Cout, XN)), dim3(256), 0, (hipStream_t)stream, a);
  OCV_CHECK_LAUNCH("ocv_conv_nhwc_exact_fwd");
  return 0;
}
