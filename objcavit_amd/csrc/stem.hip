// Stem convolution of the EfficientNet encoder: dense 3x3, stride 2, TensorFlow "SAME" padding, 3 -> 48 channels,
// + folded BatchNorm + SiLU, reading the NCHW image and writing the NHWC activation the MBConv kernels consume
// (row N1 of SURVEY.md section 8; conv_stem / bn1 / act1 of the hub backbone the reference walks in
// modules/DenseFeatureExtractor.py:18-27).
//
// K = Cin k k = 27 is far too short for an implicit-GEMM pipeline (MIOpen's fp32 NHWC igemm needs 0.75 ms for it, plus
// separate BatchNorm and SiLU passes over the 236 MB output); the layer is a pure output-write stream.  A wavefront
// owns 32 consecutive output pixels: lane (pixel l & 31, tap parity l >> 5) gathers its ceil(K / 2) input taps straight
// from the three image planes (neighbouring lanes read neighbouring pixels, stride-2 floats), which is exactly the A
// operand of v_mfma_f32_32x32x2_f32; the whole weight matrix (Cout x K, 5 KB) sits in VGPRs as the B operand for the
// lifetime of the wavefront, which walks pixel tiles grid-stride.  Exact fp32.  Output rows are 128-byte (+ 64-byte)
// contiguous runs per pixel.
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

struct StemArgs {
  const float *x, *w, *bias;
  float* y;
  int Cin, H, W, Cout, stride, pad_t, pad_l, Ho, Wo, act;
  long M;          // B * Ho * Wo
  long tiles;      // ceil(M / 32)
};

constexpr int STEM_KT = 16;      // K <= 32

template <int KS, int NT>
__global__ __launch_bounds__(256) void stem_conv_kernel(StemArgs p) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, hh = lane >> 5;
  const int K = p.Cin * KS * KS;
  float wreg[NT][STEM_KT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = 32 * j + l31;
#pragma unroll
    for (int t = 0; t < STEM_KT; ++t) {
      const int k = 2 * t + hh;
      wreg[j][t] = (n < p.Cout && k < K) ? p.w[(long)n * K + k] : 0.f;
    }
  }
  float bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bv[j] = (p.bias != nullptr && 32 * j + l31 < p.Cout) ? p.bias[32 * j + l31] : 0.f;

  const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const long plane = (long)p.H * p.W;
  for (long tile = wave0; tile < p.tiles; tile += nwaves) {
    const long m = tile * 32 + l31;
    const bool ok = m < p.M;
    const long mm = ok ? m : 0;
    const int ox = (int)(mm % p.Wo);
    const long t1 = mm / p.Wo;
    const int oy = (int)(t1 % p.Ho);
    const long b = t1 / p.Ho;
    const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
    const float* img = p.x + b * p.Cin * plane;
    float a[STEM_KT];
#pragma unroll
    for (int t = 0; t < STEM_KT; ++t) {
      // taps k = 2t (hh = 0) and 2t + 1 (hh = 1): channel / row / column are compile-time for each, selected by hh
      constexpr int dummy = 0; (void)dummy;
      const int k0 = 2 * t, k1 = 2 * t + 1;
      const int c = hh ? k1 / (KS * KS) : k0 / (KS * KS);
      const int ky = hh ? (k1 % (KS * KS)) / KS : (k0 % (KS * KS)) / KS;
      const int kx = hh ? k1 % KS : k0 % KS;
      const int iy = iy0 + ky, ix = ix0 + kx;
      const bool in = ok && (2 * t + hh) < K && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      a[t] = in ? img[c * plane + (long)iy * p.W + ix] : 0.f;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x16{0};
#pragma unroll
    for (int t = 0; t < STEM_KT; ++t) {
      if (2 * t < K) {
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = mfma_32x32x2(a[t], wreg[j][t], acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = 32 * j + l31;
      if (n < p.Cout) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long mo = tile * 32 + acc_row(r, hh);
          if (mo < p.M) {
            float v = acc[j][r] + bv[j];
            if (p.act == OCV_ACT_SILU) v = fast_silu(v);
            else if (p.act == OCV_ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == OCV_ACT_LEAKY_RELU) v = v > 0.f ? v : 0.01f * v;
            p.y[mo * p.Cout + n] = v;
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" int ocv_stem_conv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                                 int Cout, int k, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                                 ocv_stream_t stream) {
  OCV_CHECK_ARG(x && w && y, "ocv_stem_conv_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && Cin >= 1 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1 && Cout >= 1, "ocv_stem_conv_fwd: bad sizes");
  OCV_CHECK_ARG(k == 3 && Cin * k * k <= 2 * STEM_KT, "ocv_stem_conv_fwd: kernel must be 3x3 with Cin * 9 <= %d (got k=%d Cin=%d)", 2 * STEM_KT, k, Cin);
  OCV_CHECK_ARG(Cout <= 64, "ocv_stem_conv_fwd: at most 64 output channels (got %d)", Cout);
  OCV_CHECK_ARG(stride >= 1 && pad_t >= 0 && pad_l >= 0 && pad_t < k && pad_l < k, "ocv_stem_conv_fwd: bad stride / padding");
  OCV_CHECK_ARG((Ho - 1) * stride - pad_t < H && (Wo - 1) * stride - pad_l < W, "ocv_stem_conv_fwd: output larger than the padded input allows");
  OCV_CHECK_ARG(act >= OCV_ACT_NONE && act <= OCV_ACT_SILU, "ocv_stem_conv_fwd: activation must be none / ReLU / LeakyReLU / SiLU");
  StemArgs a{x, w, bias, y, Cin, H, W, Cout, stride, pad_t, pad_l, Ho, Wo, act, (long)B * Ho * Wo, 0};
  a.tiles = (a.M + 31) / 32;
  long blocks = (a.tiles + 3) / 4;
  if (blocks > 256L * 8) blocks = 256L * 8;
  hipStream_t st = (hipStream_t)stream;
  if (Cout <= 32) hipLaunchKernelGGL((stem_conv_kernel<3, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((stem_conv_kernel<3, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  OCV_CHECK_LAUNCH("ocv_stem_conv_fwd");
  return 0;
}
