// Bilinear resize (align_corners = True) + channel concat with the skip connection + fp32 -> split-bf16, one pass,
// NHWC, for the UNet decoder on gfx950 (UpSampleWithSkip.forward, modules/DenseFeatureExtractor.py:44-47:
// F.interpolate(x, size=skip.size, mode='bilinear', align_corners=True); torch.cat([up_x, skip], dim=1)).
//
// Output format = what the split-bf16 convolution consumes directly (csrc/conv_igemm.hip): the "hl32" layout of
// include/objcavit_hip.h -- per pixel and per block of 32 channels of the concatenated activation [B, H, W, C1 + C2],
// 32 hi = bf16(v) values followed by the 32 lo = bf16(v - hi) values (one 128-byte line per convolution K step; pad
// channels up to the next multiple of 32 are written as zeros).  Producing the split ONCE here
// (and in the convolution epilogues) instead of inside every convolution removes 5 VALU operations per element from
// each of the 9 taps x N-tiles that re-read the element -- the limiter of the first convolution kernel -- and the
// resized tensor and the concatenated tensor are never materialised in fp32.
// HBM-bound: reads C1 x 4 B x (h w / H W, through L2) + C2 x 4 B, writes (C1 + C2) x 4 B per output pixel; a lane owns
// 4 consecutive channels of one output pixel (16-byte loads, 8-byte stores).
// Arithmetic follows ATen's upsample_bilinear2d (scale = (in-1)/(out-1), src = scale*dst, lambda1 = src - floor(src),
// out = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)) so the fp32 value before splitting matches torch to rounding.
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

struct UpArgs {
  const float *x, *skip;
  unsigned short* hl;           // bf16 bit patterns, hl32 layout (csrc/conv_igemm.hip): per pixel and 32-channel block, 32 hi then 32 lo
  int Cp;                       // channels rounded up to 32 (pad channels are written as zeros)
  int h, w, H, W, C1, C2;
  float sh, sw;
  long total;                   // B * H * W * Cp / 4
};

// 4 consecutive channels c .. c + 3 (c % 4 == 0: inside one 32-block) of pixel pix
__device__ __forceinline__ void store_split4(unsigned short* hl, long pix, int c, int Cp, float4 v) {
  unsigned short* hi = hl + pix * 2 * Cp + (c >> 5) * 64 + (c & 31);
  unsigned short* lo = hi + 32;
  const float f[4] = {v.x, v.y, v.z, v.w};
  unsigned short h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 hb = (__bf16)f[i];
    const __bf16 lb = (__bf16)(f[i] - (float)hb);
    h[i] = __builtin_bit_cast(unsigned short, hb);
    l[i] = __builtin_bit_cast(unsigned short, lb);
  }
  *reinterpret_cast<uint2*>(hi) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
  *reinterpret_cast<uint2*>(lo) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
}

constexpr int UP_ITEMS = 8;

__global__ __launch_bounds__(256) void upsample_concat_split_kernel(UpArgs p) {
  const int C = p.C1 + p.C2, c4n = p.Cp >> 2;
  // XCD-aware, bijective workgroup -> work map: consecutive workgroup ids go round-robin to the 8 XCDs, and the four
  // bilinear taps of neighbouring output pixels re-read the same low-resolution rows -- through their XCD's own L2.
  // Each XCD gets a contiguous band of output rows (FETCH_SIZE of the 240 x 320 launch was 3x the algorithmic bytes
  // with the plain grid-stride order); a workgroup walks UP_ITEMS consecutive 256-element groups.
  long wg = blockIdx.x;
  {
    const long nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, i = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const long idx0 = wg * (256L * UP_ITEMS) + threadIdx.x;
#pragma unroll 1
  for (int it = 0; it < UP_ITEMS; ++it) {
    const long idx = idx0 + it * 256L;
    if (idx >= p.total) break;
    const int c = (int)(idx % c4n) * 4;
    long t = idx / c4n;
    const int X = (int)(t % p.W);
    t /= p.W;
    const int Y = (int)(t % p.H);
    const long b = t / p.H;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c >= C) {
      // pad channel quad: zeros
    } else if (c < p.C1) {
      const float sy = p.sh * Y, sx = p.sw * X;
      const int y0 = (int)sy, x0 = (int)sx;
      const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
      const float h1 = sy - (float)y0, h0 = 1.0f - h1, w1 = sx - (float)x0, w0 = 1.0f - w1;
      const float* base = p.x + b * (long)p.h * p.w * p.C1 + c;
      const float4 v00 = ld4(base + ((long)y0 * p.w + x0) * p.C1), v01 = ld4(base + ((long)y0 * p.w + x1) * p.C1);
      const float4 v10 = ld4(base + ((long)y1 * p.w + x0) * p.C1), v11 = ld4(base + ((long)y1 * p.w + x1) * p.C1);
      v.x = h0 * (w0 * v00.x + w1 * v01.x) + h1 * (w0 * v10.x + w1 * v11.x);
      v.y = h0 * (w0 * v00.y + w1 * v01.y) + h1 * (w0 * v10.y + w1 * v11.y);
      v.z = h0 * (w0 * v00.z + w1 * v01.z) + h1 * (w0 * v10.z + w1 * v11.z);
      v.w = h0 * (w0 * v00.w + w1 * v01.w) + h1 * (w0 * v10.w + w1 * v11.w);
    } else {
      v = ld4(p.skip + ((b * p.H + Y) * (long)p.W + X) * p.C2 + (c - p.C1));
    }
    store_split4(p.hl, (b * p.H + Y) * (long)p.W + X, c, p.Cp, v);
  }
}

}  // namespace

extern "C" int ocv_upsample_concat_split_fwd(const float* x, int h, int w, int C1, const float* skip, int C2, void* out_hl,
                                             int B, int H, int W, ocv_stream_t stream) {
  OCV_CHECK_ARG(x && out_hl, "ocv_upsample_concat_split_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && h >= 1 && w >= 1 && H >= 1 && W >= 1 && C1 >= 4 && C1 % 4 == 0, "ocv_upsample_concat_split_fwd: bad sizes (C1 must be a multiple of 4)");
  OCV_CHECK_ARG(skip == nullptr ? C2 == 0 : (C2 >= 4 && C2 % 4 == 0), "ocv_upsample_concat_split_fwd: C2 must be a multiple of 4 (0 without a skip tensor)");
  OCV_CHECK_ARG(ocv_aligned16(x) && ocv_aligned16(skip) && ocv_aligned16(out_hl), "ocv_upsample_concat_split_fwd: operands must be 16-byte aligned");
  const int Cp = (C1 + C2 + 31) / 32 * 32;
  UpArgs a{x, skip, (unsigned short*)out_hl, Cp, h, w, H, W, C1, C2,
           H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f,
           (long)B * H * W * (Cp / 4)};
  const long blocks = (a.total + 256L * UP_ITEMS - 1) / (256L * UP_ITEMS);
  OCV_CHECK_ARG(blocks < (1L << 31), "ocv_upsample_concat_split_fwd: tensor too large");
  hipLaunchKernelGGL(upsample_concat_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  OCV_CHECK_LAUNCH("ocv_upsample_concat_split_fwd");
  return 0;
}
