// NHWC building blocks of the EfficientNet-B5 encoder's MBConv stages for
// gfx950 (row N1 of SURVEY.md section 8; the reference runs these through its hub
// backbone, modules/DenseFeatureExtractor.py:18-27,149):
//
//   ocv_pointwise_conv_nhwc_fwd   1x1 convolution = row-major GEMM on [B*H*W, Cin] with fused per-(image, channel)
//                                 squeeze-excite gate on the INPUT, bias (folded BatchNorm), SiLU / sigmoid and
//                                 residual add -- exact fp32 on v_mfma_f32_32x32x2_f32
//   ocv_depthwise_conv_nhwc_fwd   k x k depthwise convolution (k 3/5, stride 1/2, TF "SAME"), + bias + SiLU
//   ocv_channel_mean_nhwc_fwd     squeeze: mean over H*W per (image, channel), two-stage and deterministic
//
// Why NHWC and why fused: the expand / depthwise activations are the largest tensors of the whole network
// (144 ch x 240 x 320 x 16 images = 708 MB) and every un-fused element-wise pass (bias add, BatchNorm, SiLU,
// gate multiply, residual add) re-reads and re-writes them; the 1x1 convolutions themselves are HBM-bound at
// these channel counts (2 Cin Cout / 4 (Cin + Cout) = 10 flop/B for 24 -> 144).  With channels innermost a pixel
// row is the GEMM's A row as it lies in memory, the depthwise kernel vectorises over channels (float4), and the
// decoder consumes the skip activations without a layout change.
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

constexpr int KC = 128, XLD = KC + 4;      // K chunk and padded LDS row (floats): conflict-free ds_read_b128

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case OCV_ACT_RELU: return fmaxf(v, 0.f);
    case OCV_ACT_LEAKY_RELU: return v > 0.f ? v : 0.01f * v;
    case OCV_ACT_SILU: return fast_silu(v);
    case OCV_ACT_SIGMOID: return fast_sigmoid(v);
    default: return v;
  }
}

// ---------------------------------------------------------------------------
// pointwise convolution
// ---------------------------------------------------------------------------
struct PWArgs {
  const float *x, *gate, *W, *bias, *res;
  float* y;
  long M;
  int K, N, rows_per_image, act;
};

// WN = wavefronts along the output channels: tile = (128 / WN) rows x (32 WN) channels per workgroup.
// Weights stream from L2 straight into VGPRs (each element feeds exactly one wavefront), activation rows are
// staged in LDS; K order inside a chunk is permuted so both operands are 16-byte vectors (see csrc/linear.hip).
template <int WN>
__global__ __launch_bounds__(256) void pointwise_kernel(PWArgs p) {
  constexpr int RM = 128 / WN;
  extern __shared__ __attribute__((aligned(16))) float Xs[];      // [RM][XLD]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wr = wave / WN, wc = wave % WN;
  const long m0 = (long)blockIdx.x * RM;
  const int n = blockIdx.y * (32 * WN) + wc * 32 + l31;
  const float* wrow = p.W + (long)(n < p.N ? n : p.N - 1) * p.K + 4 * hh;

  f32x16 acc = {0};
  for (int k0 = 0; k0 < p.K; k0 += KC) {
    const int kc = min(KC, p.K - k0);
    float4 w[KC / 8];
#pragma unroll
    for (int t = 0; t < KC / 8; ++t) w[t] = (8 * t < kc) ? ld4(wrow + k0 + 8 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // stage RM rows x kc floats (x gate): thread -> (row = tid / 8 + 32 i, 4 floats at (tid % 8) * 4 + 32 j)
#pragma unroll
    for (int i = 0; i < RM / 32; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const long m = m0 + row;
      const bool ok = m < p.M;
      const float* src = p.x + m * p.K + k0;
      const float* gsrc = (p.gate != nullptr && ok) ? p.gate + (m / p.rows_per_image) * p.K + k0 : nullptr;
#pragma unroll
      for (int j = 0; j < KC / 32; ++j) {
        const int c4 = (tid & 7) * 4 + 32 * j;
        if (c4 < kc) {
          float4 t = ok ? ld4(src + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
          if (gsrc != nullptr) {
            const float4 g = ld4(gsrc + c4);
            t.x *= g.x; t.y *= g.y; t.z *= g.z; t.w *= g.w;
          }
          *reinterpret_cast<float4*>(&Xs[row * XLD + c4]) = t;
        }
      }
    }
    __syncthreads();
    const float* xrow = Xs + (wr * 32 + l31) * XLD + 4 * hh;
#pragma unroll
    for (int t = 0; t < KC / 8; ++t) {
      if (8 * t < kc) {
        const float4 a = *reinterpret_cast<const float4*>(xrow + 8 * t);
        acc = mfma_32x32x2(a.x, w[t].x, acc);
        acc = mfma_32x32x2(a.y, w[t].y, acc);
        acc = mfma_32x32x2(a.z, w[t].z, acc);
        acc = mfma_32x32x2(a.w, w[t].w, acc);
      }
    }
  }

  if (n < p.N) {
    const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long m = m0 + wr * 32 + acc_row(r, hh);
      if (m < p.M) {
        float v = apply_act(acc[r] + bv, p.act);
        if (p.res != nullptr) v += p.res[m * p.N + n];
        p.y[m * p.N + n] = v;
      }
    }
  }
}

// Small-K variant (K <= 128, the HBM-bound expand / DS-project layers: 24 -> 144 at 240 x 320 moves 826 MB for
// 8 GFLOP).  Each wavefront owns 32 rows, keeps them (x gate) in VGPRs in MFMA A-operand order -- lane (l31, hh)
// holds floats 8t + 4hh .. +3 of row l31, one 16-byte global load per t, every 32-byte sector fetched exactly once
// -- and walks ALL column tiles itself (weights re-fetched per tile from L2, K/8 float4 per lane), so activation
// rows are read from HBM exactly once whatever Cout is.  No LDS and no barrier: occupancy is bound by VGPRs only
// (the earlier LDS-staged form held 67 KB per workgroup = 2 wavefronts per SIMD and ran at 1.5 TB/s).
// KT = ceil(K / 8) rounded up to 4 / 8 / 16 sizes the register array.
template <int KT>
__global__ __launch_bounds__(256) void pointwise_smallk_kernel(PWArgs p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const long m0 = (long)blockIdx.x * 128;
  const int K = p.K;
  float4 a[KT];
  {
    const long m = m0 + wave * 32 + l31;
    const bool ok = m < p.M;
    const float* src = p.x + (ok ? m : 0) * K + 4 * hh;
    const float* gsrc = p.gate != nullptr ? p.gate + ((ok ? m : 0) / p.rows_per_image) * K + 4 * hh : nullptr;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      a[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (8 * t < K && ok) {
        a[t] = ld4(src + 8 * t);
        if (gsrc != nullptr) {
          const float4 g = ld4(gsrc + 8 * t);
          a[t].x *= g.x; a[t].y *= g.y; a[t].z *= g.z; a[t].w *= g.w;
        }
      }
    }
  }

  // blockIdx.y splits the column tiles when there are too few 128-row workgroups to fill the chip (small M, wide N)
  const int ntiles_all = (p.N + 31) >> 5;
  const int per_y = (ntiles_all + gridDim.y - 1) / gridDim.y;
  const int nt_lo = blockIdx.y * per_y, ntiles = min(ntiles_all, nt_lo + per_y);
  for (int nt = nt_lo; nt < ntiles; ++nt) {
    const int n = nt * 32 + l31;
    const float* wrow = p.W + (long)(n < p.N ? n : p.N - 1) * K + 4 * hh;
    f32x16 acc = {0};
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if (8 * t < K) {
        const float4 w = ld4(wrow + 8 * t);
        acc = mfma_32x32x2(a[t].x, w.x, acc);
        acc = mfma_32x32x2(a[t].y, w.y, acc);
        acc = mfma_32x32x2(a[t].z, w.z, acc);
        acc = mfma_32x32x2(a[t].w, w.w, acc);
      }
    }
    if (n < p.N) {
      const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wave * 32 + acc_row(r, hh);
        if (m < p.M) {
          float v = apply_act(acc[r] + bv, p.act);
          if (p.res != nullptr) v += p.res[m * p.N + n];
          p.y[m * p.N + n] = v;
        }
      }
    }
  }
}

template <int WN>
int launch_pw(const PWArgs& a, hipStream_t st) {
  constexpr int RM = 128 / WN;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)pointwise_kernel<WN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  dim3 grid((unsigned)((a.M + RM - 1) / RM), ocv_cdiv(a.N, 32 * WN));
  hipLaunchKernelGGL((pointwise_kernel<WN>), grid, dim3(256), (size_t)RM * XLD * sizeof(float), st, a);
  OCV_CHECK_LAUNCH("ocv_pointwise_conv_nhwc_fwd");
  return 0;
}

// ---------------------------------------------------------------------------
// depthwise convolution, NHWC: thread -> 4 channels x PX consecutive output pixels of one row
// ---------------------------------------------------------------------------
struct DWNArgs {
  const float *in, *w, *bias;      // w: [k*k][C]
  float* out;
  int C, H, W, Ho, Wo, pad_t, pad_l, act;
  long total;                      // B * Ho * ceil(Wo / PX) * C / 4
};

template <int K, int S, int PX>
__global__ __launch_bounds__(256) void depthwise_nhwc_kernel(DWNArgs p) {
  constexpr int NIN = (PX - 1) * S + K;
  const int c4n = p.C >> 2, wox = (p.Wo + PX - 1) / PX;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < p.total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % c4n) * 4;
    long t = idx / c4n;
    const int ox = (int)(t % wox) * PX;
    t /= wox;
    const int oy = (int)(t % p.Ho);
    const long b = t / p.Ho;
    const float* ib = p.in + b * (long)p.H * p.W * p.C + c;
    const int ix0 = ox * S - p.pad_l;
    float4 acc[PX];
    const float4 bv = p.bias ? ld4(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int o = 0; o < PX; ++o) acc[o] = bv;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int iy = oy * S - p.pad_t + i;
      if (iy < 0 || iy >= p.H) continue;
      const float* row = ib + (long)iy * p.W * p.C;
      float4 v[NIN];
#pragma unroll
      for (int j = 0; j < NIN; ++j) {
        const int ix = ix0 + j;
        v[j] = (ix >= 0 && ix < p.W) ? ld4(row + (long)ix * p.C) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float4 wv = ld4(p.w + (long)(i * K + j) * p.C + c);
#pragma unroll
        for (int o = 0; o < PX; ++o) {
          const float4 x = v[o * S + j];
          acc[o].x = fmaf(wv.x, x.x, acc[o].x); acc[o].y = fmaf(wv.y, x.y, acc[o].y);
          acc[o].z = fmaf(wv.z, x.z, acc[o].z); acc[o].w = fmaf(wv.w, x.w, acc[o].w);
        }
      }
    }
    float* ob = p.out + ((b * p.Ho + oy) * (long)p.Wo + ox) * p.C + c;
#pragma unroll
    for (int o = 0; o < PX; ++o) {
      if (ox + o < p.Wo) {
        float4 r = acc[o];
        r.x = apply_act(r.x, p.act); r.y = apply_act(r.y, p.act); r.z = apply_act(r.z, p.act); r.w = apply_act(r.w, p.act);
        *reinterpret_cast<float4*>(ob + (long)o * p.C) = r;
      }
    }
  }
}

template <int K, int S, int PX>
int launch_dwn(DWNArgs a, int B, hipStream_t st) {
  a.total = (long)B * a.Ho * ((a.Wo + PX - 1) / PX) * (a.C / 4);
  long blocks = (a.total + 255) / 256;
  if (blocks > 256L * 64) blocks = 256L * 64;
  hipLaunchKernelGGL((depthwise_nhwc_kernel<K, S, PX>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  OCV_CHECK_LAUNCH("ocv_depthwise_conv_nhwc_fwd");
  return 0;
}

// ---------------------------------------------------------------------------
// channel mean over H*W (squeeze), NHWC.  Stage 1: grid (C/64 chunks, splits, B), 256 threads = 16 pixel lanes x
// 16 channel quads; partial sums in fixed order.  Stage 2: add the splits in order and scale.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ part, long P,
                                                          int C, int splits) {
  __shared__ float4 red[16][16];
  const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = (blockIdx.x * 16 + q) * 4;
  const int sp = blockIdx.y;
  const long b = blockIdx.z;
  const long per = (P + splits - 1) / splits;
  const long p0 = sp * per, p1 = min(P, p0 + per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    const float* src = x + b * P * C + c;
    for (long pix = p0 + pl; pix < p1; pix += 16) {
      const float4 v = ld4(src + pix * C);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[pl][q] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    float4 t = red[0][q];
#pragma unroll
    for (int i = 1; i < 16; ++i) { t.x += red[i][q].x; t.y += red[i][q].y; t.z += red[i][q].z; t.w += red[i][q].w; }
    *reinterpret_cast<float4*>(part + ((b * splits + sp) * (long)C) + c) = t;
  }
}

__global__ __launch_bounds__(256) void channel_mean_finish_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                  int C, int splits, float inv, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;      // over B * C
  if (i >= total) return;
  const long b = i / C;
  const int c = (int)(i - b * C);
  float s = 0.f;
  for (int sp = 0; sp < splits; ++sp) s += part[(b * splits + sp) * (long)C + c];
  out[i] = s * inv;
}

// squeeze-excite gate, two tiny launches (both latency-bound; every loop is unrolled for independent loads):
//   hid[b][r]  = silu( b1[r] + sum_c W1[r][c] * mean[b][c] )          one wavefront per (r, b), coalesced over c
//   gate[b][c] = sigmoid( b2[c] + sum_r W2t[r][c] * hid[b][r] )       one lane per (c, b), coalesced over c
__global__ __launch_bounds__(256) void se_hidden_kernel(const float* __restrict__ mean, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, float* __restrict__ hid, int C,
                                                        int R) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const long b = blockIdx.y;
  const float* mb = mean + b * C;
  const float* wr = w1 + (long)r * C;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = lane;
  for (; c + 192 < C; c += 256) {
    s0 = fmaf(wr[c], mb[c], s0);
    s1 = fmaf(wr[c + 64], mb[c + 64], s1);
    s2 = fmaf(wr[c + 128], mb[c + 128], s2);
    s3 = fmaf(wr[c + 192], mb[c + 192], s3);
  }
  for (; c < C; c += 64) s0 = fmaf(wr[c], mb[c], s0);
  const float s = wave_sum((s0 + s1) + (s2 + s3));
  if (lane == 0) {
    const float v = s + b1[r];
    hid[b * R + r] = fast_silu(v);
  }
}

__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ hid, const float* __restrict__ w2t,
                                                      const float* __restrict__ b2, float* __restrict__ gate, int C,
                                                      int R) {
  __shared__ float hs[256];
  const long b = blockIdx.y;
  if (threadIdx.x < R) hs[threadIdx.x] = hid[b * R + threadIdx.x];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float* w = w2t + c;
  float s0 = b2[c], s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = 0;
  for (; r + 3 < R; r += 4) {
    s0 = fmaf(w[(long)r * C], hs[r], s0);
    s1 = fmaf(w[(long)(r + 1) * C], hs[r + 1], s1);
    s2 = fmaf(w[(long)(r + 2) * C], hs[r + 2], s2);
    s3 = fmaf(w[(long)(r + 3) * C], hs[r + 3], s3);
  }
  for (; r < R; ++r) s0 = fmaf(w[(long)r * C], hs[r], s0);
  const float s = (s0 + s1) + (s2 + s3);
  gate[b * C + c] = fast_sigmoid(s);
}

int mean_splits(int B, int C, long P) {
  const int chunks = (C + 63) / 64;
  int s = 1;
  while (s < 64 && (long)B * chunks * s < 1024 && P / (s * 2) >= 256) s *= 2;
  return s;
}

}  // namespace

extern "C" int ocv_pointwise_conv_nhwc_fwd(const float* x, const float* gate, int rows_per_image, const float* W,
                                           const float* bias, const float* residual, float* y, long M, int Cin,
                                           int Cout, int act, ocv_stream_t stream) {
  OCV_CHECK_ARG(x && W && y, "ocv_pointwise_conv_nhwc_fwd: null pointer");
  OCV_CHECK_ARG(M >= 0 && Cin >= 8 && Cin % 8 == 0 && Cout >= 1, "ocv_pointwise_conv_nhwc_fwd: Cin must be a positive multiple of 8 (got M=%ld Cin=%d Cout=%d)", M, Cin, Cout);
  OCV_CHECK_ARG(gate == nullptr || rows_per_image >= 1, "ocv_pointwise_conv_nhwc_fwd: gate needs rows_per_image");
  OCV_CHECK_ARG(act >= 0 && act <= OCV_ACT_SIGMOID, "ocv_pointwise_conv_nhwc_fwd: unknown activation %d", act);
  OCV_CHECK_ARG(ocv_aligned16(x) && ocv_aligned16(W) && ocv_aligned16(gate), "ocv_pointwise_conv_nhwc_fwd: x / W / gate must be 16-byte aligned");
  if (M == 0) return 0;
  PWArgs a{x, gate, W, bias, residual, y, M, Cin, Cout, rows_per_image > 0 ? rows_per_image : 1, act};
  hipStream_t st = (hipStream_t)stream;
  if (Cin <= KC && M >= 4096) {
    const long mblocks = (M + 127) / 128;
    int ysplit = 1;
    const int ntl = (Cout + 31) / 32;
    while (mblocks * ysplit < 768 && ysplit * 2 <= ntl) ysplit *= 2;
    const dim3 grid((unsigned)mblocks, ysplit);
    if (Cin <= 32) hipLaunchKernelGGL(pointwise_smallk_kernel<4>, grid, dim3(256), 0, st, a);
    else if (Cin <= 64) hipLaunchKernelGGL(pointwise_smallk_kernel<8>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(pointwise_smallk_kernel<16>, grid, dim3(256), 0, st, a);
    OCV_CHECK_LAUNCH("ocv_pointwise_conv_nhwc_fwd(small K)");
    return 0;
  }
  if (Cout <= 32) return launch_pw<1>(a, st);
  if (Cout <= 64) return launch_pw<2>(a, st);
  return launch_pw<4>(a, st);
}

extern "C" int ocv_depthwise_conv_nhwc_fwd(const float* in, const float* w, const float* bias, float* out, int B, int C,
                                           int H, int W, int k, int stride, int pad_t, int pad_l, int Ho, int Wo,
                                           int act, ocv_stream_t stream) {
  OCV_CHECK_ARG(in && w && out, "ocv_depthwise_conv_nhwc_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && C >= 4 && C % 4 == 0 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1, "ocv_depthwise_conv_nhwc_fwd: bad sizes (C must be a multiple of 4)");
  OCV_CHECK_ARG((k == 3 || k == 5) && (stride == 1 || stride == 2), "ocv_depthwise_conv_nhwc_fwd: k must be 3 or 5 and stride 1 or 2");
  OCV_CHECK_ARG(pad_t >= 0 && pad_l >= 0 && pad_t < k && pad_l < k, "ocv_depthwise_conv_nhwc_fwd: bad padding");
  OCV_CHECK_ARG((Ho - 1) * stride - pad_t < H && (Wo - 1) * stride - pad_l < W, "ocv_depthwise_conv_nhwc_fwd: output larger than the padded input allows");
  OCV_CHECK_ARG(act == OCV_ACT_NONE || act == OCV_ACT_SILU, "ocv_depthwise_conv_nhwc_fwd: activation must be none or SiLU");
  OCV_CHECK_ARG(ocv_aligned16(in) && ocv_aligned16(w) && ocv_aligned16(out) && ocv_aligned16(bias), "ocv_depthwise_conv_nhwc_fwd: operands must be 16-byte aligned");
  DWNArgs a{in, w, bias, out, C, H, W, Ho, Wo, pad_t, pad_l, act, 0};
  hipStream_t st = (hipStream_t)stream;
  if (k == 3 && stride == 1) return launch_dwn<3, 1, 4>(a, B, st);
  if (k == 3 && stride == 2) return launch_dwn<3, 2, 2>(a, B, st);
  if (k == 5 && stride == 1) return launch_dwn<5, 1, 4>(a, B, st);
  return launch_dwn<5, 2, 2>(a, B, st);
}

extern "C" size_t ocv_channel_mean_workspace_bytes(int B, int C, long P) {
  if (B < 1 || C < 4 || P < 1) return 0;
  return (size_t)B * mean_splits(B, C, P) * C * sizeof(float);
}

extern "C" int ocv_channel_mean_nhwc_fwd(const float* x, float* out, int B, int C, long P, void* workspace,
                                         size_t workspace_bytes, ocv_stream_t stream) {
  OCV_CHECK_ARG(x && out && workspace, "ocv_channel_mean_nhwc_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && B <= 65535 && C >= 4 && C % 4 == 0 && P >= 1, "ocv_channel_mean_nhwc_fwd: bad sizes (C must be a multiple of 4)");
  OCV_CHECK_ARG(workspace_bytes >= ocv_channel_mean_workspace_bytes(B, C, P) && ocv_aligned16(workspace) && ocv_aligned16(x),
                "ocv_channel_mean_nhwc_fwd: workspace too small or operands misaligned");
  const int splits = mean_splits(B, C, P);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(channel_sum_kernel, dim3((C + 63) / 64, splits, B), dim3(256), 0, st, x, (float*)workspace, P, C, splits);
  OCV_CHECK_LAUNCH("ocv_channel_mean_nhwc_fwd(sum)");
  const long total = (long)B * C;
  hipLaunchKernelGGL(channel_mean_finish_kernel, dim3(ocv_cdiv(total, 256)), dim3(256), 0, st, (const float*)workspace, out, C,
                     splits, 1.0f / (float)P, total);
  OCV_CHECK_LAUNCH("ocv_channel_mean_nhwc_fwd(finish)");
  return 0;
}

extern "C" int ocv_se_gate_fwd(const float* mean, const float* w1, const float* b1, const float* w2t, const float* b2,
                               float* gate, float* hidden_ws, int B, int C, int R, ocv_stream_t stream) {
  OCV_CHECK_ARG(mean && w1 && b1 && w2t && b2 && gate && hidden_ws, "ocv_se_gate_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && B <= 65535 && C >= 1 && R >= 1 && R <= 256, "ocv_se_gate_fwd: bad sizes (R <= 256)");
  hipLaunchKernelGGL(se_hidden_kernel, dim3(ocv_cdiv(R, 4), B), dim3(256), 0, (hipStream_t)stream, mean, w1, b1, hidden_ws, C, R);
  OCV_CHECK_LAUNCH("ocv_se_gate_fwd(hidden)");
  hipLaunchKernelGGL(se_gate_kernel, dim3(ocv_cdiv(C, 256), B), dim3(256), 0, (hipStream_t)stream, hidden_ws, w2t, b2,
                     gate, C, R);
  OCV_CHECK_LAUNCH("ocv_se_gate_fwd");
  return 0;
}
