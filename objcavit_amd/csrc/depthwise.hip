// Depthwise KxK convolution (K = 3 or 5, stride 1 or 2, arbitrary zero padding
// incl. TensorFlow "SAME" asymmetric padding) with fused per-channel bias
// (= folded BatchNorm) and SiLU, NCHW fp32, for the EfficientNet-B5 encoder
// (row N1 of SURVEY.md section 8: the MBConv depthwise stages; MIOpen has no tuned
// fp32 solver for them on gfx950 and falls back to naive_conv, 12.6 ms of the
// 30 ms encoder at bs = 16).
//
// HBM-bound by construction: 9 or 25 MACs per output against 8 bytes of traffic.
// Each lane produces 4 consecutive outputs of one row (one 16-byte store) from a
// (3S + K) x K input window read through L1/L2 (neighbouring lanes share all
// but 4S columns, neighbouring rows share K - S rows); no LDS, no cross-lane
// traffic, planes (b, c) are independent so the grid is flat over
// B * C * Ho * ceil(Wo / 4) work items.
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

struct DWArgs {
  const float* in; const float* w; const float* bias; float* out;
  int C, H, W, Ho, Wo, pad_t, pad_l, act;
  long total;      // B * C * Ho * ceil(Wo / 4)
};

template <int K, int S>
__global__ __launch_bounds__(256) void depthwise_kernel(DWArgs p) {
  constexpr int NIN = 3 * S + K;
  const int wo4 = (p.Wo + 3) >> 2;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < p.total; idx += (long)gridDim.x * 256) {
    const int ox4 = (int)(idx % wo4);
    long t = idx / wo4;
    const int oy = (int)(t % p.Ho);
    const long plane = t / p.Ho;                 // b * C + c
    const int c = (int)(plane % p.C);
    const float* ip = p.in + plane * (long)p.H * p.W;
    const float* wp = p.w + (long)c * K * K;
    const int ox = ox4 * 4;
    const int ix0 = ox * S - p.pad_l;
    const float b0 = p.bias ? p.bias[c] : 0.f;
    float acc[4] = {b0, b0, b0, b0};
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int iy = oy * S - p.pad_t + i;
      if (iy < 0 || iy >= p.H) continue;
      const float* row = ip + (long)iy * p.W;
      float v[NIN];
#pragma unroll
      for (int j = 0; j < NIN; ++j) {
        const int ix = ix0 + j;
        v[j] = (ix >= 0 && ix < p.W) ? row[ix] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float wv = wp[i * K + j];
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] = fmaf(wv, v[o * S + j], acc[o]);
      }
    }
    if (p.act == 3) {
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] = fast_silu(acc[o]);
    }
    float* op = p.out + (plane * p.Ho + oy) * (long)p.Wo + ox;
    if (ox + 3 < p.Wo && (p.Wo & 3) == 0) {
      *reinterpret_cast<float4*>(op) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (ox + o < p.Wo) op[o] = acc[o];
    }
  }
}

template <int K, int S>
int launch(const DWArgs& a, hipStream_t st) {
  long blocks = (a.total + 255) / 256;
  if (blocks > 256L * 64) blocks = 256L * 64;          // grid-stride beyond ~64 workgroups per CU
  hipLaunchKernelGGL((depthwise_kernel<K, S>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  OCV_CHECK_LAUNCH("ocv_depthwise_conv_fwd");
  return 0;
}

}  // namespace

extern "C" int ocv_depthwise_conv_fwd(const float* in, const float* w, const float* bias, float* out, int B, int C,
                                      int H, int W, int k, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                                      ocv_stream_t stream) {
  OCV_CHECK_ARG(in && w && out, "ocv_depthwise_conv_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && C >= 1 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1, "ocv_depthwise_conv_fwd: bad sizes");
  OCV_CHECK_ARG((k == 3 || k == 5) && (stride == 1 || stride == 2), "ocv_depthwise_conv_fwd: k must be 3 or 5 and stride 1 or 2 (got k=%d s=%d)", k, stride);
  OCV_CHECK_ARG(pad_t >= 0 && pad_l >= 0 && pad_t < k && pad_l < k, "ocv_depthwise_conv_fwd: bad padding");
  OCV_CHECK_ARG((Ho - 1) * stride - pad_t < H && (Wo - 1) * stride - pad_l < W, "ocv_depthwise_conv_fwd: output larger than the padded input allows");
  OCV_CHECK_ARG(act == OCV_ACT_NONE || act == OCV_ACT_SILU, "ocv_depthwise_conv_fwd: activation must be none or SiLU");
  OCV_CHECK_ARG(ocv_aligned16(out), "ocv_depthwise_conv_fwd: out must be 16-byte aligned");
  DWArgs a{in, w, bias, out, C, H, W, Ho, Wo, pad_t, pad_l, act, (long)B * C * Ho * ((Wo + 3) / 4)};
  hipStream_t st = (hipStream_t)stream;
  if (k == 3 && stride == 1) return launch<3, 1>(a, st);
  if (k == 3 && stride == 2) return launch<3, 2>(a, st);
  if (k == 5 && stride == 1) return launch<5, 1>(a, st);
  return launch<5, 2>(a, st);
}
