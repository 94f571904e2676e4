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
// Algorithmic traffic: reads C1 x 4 B x (h w / H W) + C2 x 4 B, writes (C1 + C2) x 4 B per output pixel.  Four kernels,
// chosen per launch (ocv_upsample_concat_split_fwd at the end of this file):
//   upsample_concat_split_lds_kernel   >= 2x up-scaling, C1 % 32 == 0: source tile through LDS (the decoder's case; see there)
//   upsample_concat_split_2x2_kernel   channel counts multiples of 8, not shrinking: item = 8 channels of a 2 x 2 output
//                                      block, 9 shared taps instead of 16; every load of the item ahead of its stores
//   upsample_concat_split8_kernel      multiples of 8, any scale: item = 8 channels of one pixel, 4 items per thread, all
//                                      loads ahead of all stores
//   upsample_concat_split_kernel       multiples of 4: item = 4 channels of one pixel, flat index decode
// What bounded them, in the order found (B = 16, the four decoder launches together): 1.30 ms for the last kernel --
// neither its index arithmetic (octet items with incremental indices: 1.33 ms) nor the shape of its stores (full
// 128-byte lines per lane group: 1.44 ms) but the ORDER of loads and stores: a store counts in vmcnt like a load and
// vmcnt retires in order, so each wait for an item's loads also waited for the previous item's stores (loads first:
// 1.14 ms); then the four-taps-per-output L1 traffic, 8 TB/s (2 x 2 blocks: 0.96 ms; source tile in LDS: 0.74 ms).
// torch's fill of the 240 x 320 output alone takes 0.21 ms and a copy of it 0.55 ms (tools/membw.py); that launch is
// now at 0.38 ms.
// Arithmetic follows ATen's upsample_bilinear2d (scale = (in-1)/(out-1), src = scale*dst, lambda1 = src - floor(src),
// out = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)) so the fp32 value before splitting matches torch to rounding.
#include <stdlib.h>

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
  unsigned* range_flag;         // nullable: armed range-guard word, fp16 pairs only (common.hpp ocv_range_note)
};

// 4 consecutive channels c .. c + 3 (c % 4 == 0: inside one 32-block) of pixel pix
template <bool F16>
__device__ __forceinline__ void store_split4(unsigned short* hl, long pix, int c, int Cp, float4 v, unsigned* range_flag) {
  if constexpr (F16) ocv_range_note(range_flag, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  unsigned short* hi = hl + pix * 2 * Cp + (c >> 5) * 64 + (c & 31);
  unsigned short* lo = hi + 32;
  const float f[4] = {v.x, v.y, v.z, v.w};
  unsigned short h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ocv_split1<F16>(f[i], h[i], l[i]);
  *reinterpret_cast<uint2*>(hi) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
  *reinterpret_cast<uint2*>(lo) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
}

__device__ __forceinline__ float up_amax8(const float (&f)[8]) {
  return fmaxf(fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[2]), fabsf(f[3]))),
               fmaxf(fmaxf(fabsf(f[4]), fabsf(f[5])), fmaxf(fabsf(f[6]), fabsf(f[7]))));
}

constexpr int UP_ITEMS = 8;

template <bool F16>
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
    store_split4<F16>(p.hl, (b * p.H + Y) * (long)p.W + X, c, p.Cp, v, p.range_flag);
  }
}

// Octet form (C1 and C2 multiples of 8): a work item is 8 consecutive channels of one output pixel (two 16-byte loads per
// bilinear tap, one 16-byte hi store and one 16-byte lo store); a workgroup owns PB consecutive output pixels and its
// threads walk the (pixel, octet) items in order, so a wavefront's loads are 2 KB runs of one pixel's channels and the
// pixel's row / column / interpolation weights are derived from the workgroup's first pixel by increments.  The quad
// kernel above decodes (b, Y, X, c) from a flat index for every 16 bytes it moves -- six 64-bit divisions, ~250 VALU
// instructions per item -- and sat at 2.8 TB/s with the vector ALU 60 % busy; this form spends ~70 per 32 bytes.
typedef __bf16 up_bf16x8 __attribute__((ext_vector_type(8)));

struct Up8Args {
  const float *x, *skip;
  __bf16* hl;
  int Cp, h, w, H, W, C1, C2;
  float sh, sw, inv_noct;
  int noct, PB;                  // octets per pixel (Cp / 8), pixels per workgroup
  long npix;                     // B * H * W
  unsigned* range_flag;          // nullable (UpArgs::range_flag)
};

__device__ __forceinline__ float4 up_lerp(const float4 a, const float4 b, const float4 c, const float4 d, float h0, float h1,
                                          float w0, float w1) {
  float4 v;
  v.x = h0 * (w0 * a.x + w1 * b.x) + h1 * (w0 * c.x + w1 * d.x);
  v.y = h0 * (w0 * a.y + w1 * b.y) + h1 * (w0 * c.y + w1 * d.y);
  v.z = h0 * (w0 * a.z + w1 * b.z) + h1 * (w0 * c.z + w1 * d.z);
  v.w = h0 * (w0 * a.w + w1 * b.w) + h1 * (w0 * c.w + w1 * d.w);
  return v;
}

constexpr int UP8_ITEMS = 4;          // items per thread, all loaded before the first store

template <bool F16>
__global__ __launch_bounds__(256) void upsample_concat_split8_kernel(Up8Args p) {
  const int C = p.C1 + p.C2;
  long wg = blockIdx.x;
  {
    const long nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, i = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const long pix0 = wg * p.PB;                                   // wave-uniform: scalar divisions, once per workgroup
  const int X0 = (int)(pix0 % p.W);
  const long t0 = pix0 / p.W;
  const int Y0 = (int)(t0 % p.H);
  const long b0 = t0 / p.H;
  const int nitems = (int)(min((long)p.PB, p.npix - pix0)) * p.noct;      // <= 256 * UP8_ITEMS
  // Branch-free body, loads of all items ahead of the first store: on gfx9 a store counts in vmcnt like a load and
  // vmcnt retires in order, so a loop of (loads, wait, stores) makes every wait for an item's loads also wait for the
  // previous item's stores to be acknowledged.
  up_bf16x8 hi[UP8_ITEMS], lo[UP8_ITEMS];
  __bf16* dst[UP8_ITEMS];
#pragma unroll
  for (int u = 0; u < UP8_ITEMS; ++u) {
    const int item = min((int)threadIdx.x + 256 * u, nitems - 1);          // clamped: stores are predicated below
    const int dp = (int)(((float)item + 0.5f) * p.inv_noct);               // item / noct (exact: item < 2^16, noct <= 2^12)
    const int c = (item - dp * p.noct) * 8;
    int X = X0 + dp, Y = Y0;
    long b = b0;
    while (X >= p.W) { X -= p.W; ++Y; }
    while (Y >= p.H) { Y -= p.H; ++b; }
    const bool is_up = c < p.C1, is_skip = !is_up && c < C;
    const float sy = p.sh * Y, sx = p.sw * X;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
    const float h1 = sy - (float)y0, h0 = 1.0f - h1, w1 = sx - (float)x0, w0 = 1.0f - w1;
    const float* base = p.x + b * (long)p.h * p.w * p.C1 + (is_up ? c : 0);
    const float* sk = is_skip ? p.skip + ((b * p.H + Y) * (long)p.W + X) * p.C2 + (c - p.C1) : base;
    const float* r00 = is_up ? base + ((long)y0 * p.w + x0) * p.C1 : sk;    // skip / pad octets: four (L1-resident) copies
    const float* r01 = is_up ? base + ((long)y0 * p.w + x1) * p.C1 : sk;
    const float* r10 = is_up ? base + ((long)y1 * p.w + x0) * p.C1 : sk;
    const float* r11 = is_up ? base + ((long)y1 * p.w + x1) * p.C1 : sk;
    const float4 a0 = ld4(r00), a1 = ld4(r00 + 4), b0v = ld4(r01), b1v = ld4(r01 + 4);
    const float4 c0 = ld4(r10), c1 = ld4(r10 + 4), d0 = ld4(r11), d1 = ld4(r11 + 4);
    const float4 la = up_lerp(a0, b0v, c0, d0, h0, h1, w0, w1), lb = up_lerp(a1, b1v, c1, d1, h0, h1, w0, w1);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 va = is_up ? la : (is_skip ? a0 : z), vb = is_up ? lb : (is_skip ? a1 : z);
    const float f[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
    if constexpr (F16) ocv_range_note(p.range_flag, up_amax8(f));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned short hb_, lb_;
      ocv_split1<F16>(f[i], hb_, lb_);
      hi[u][i] = __builtin_bit_cast(__bf16, hb_);
      lo[u][i] = __builtin_bit_cast(__bf16, lb_);
    }
    dst[u] = p.hl + ((b * p.H + Y) * (long)p.W + X) * 2 * p.Cp + (c >> 5) * 64 + (c & 31);
  }
#pragma unroll
  for (int u = 0; u < UP8_ITEMS; ++u) {
    if ((int)threadIdx.x + 256 * u < nitems) {
      *reinterpret_cast<up_bf16x8*>(dst[u]) = hi[u];
      *reinterpret_cast<up_bf16x8*>(dst[u] + 32) = lo[u];
    }
  }
}

// 2 x 2 form (octets, and a resize that does not shrink: sh, sw <= 1): a work item is 8 channels of a 2 x 2 block of
// output pixels.  Neighbouring outputs of an up-scaling share taps -- the block's sixteen taps lie in a 3 x 3
// neighbourhood of the source -- so an item issues 9 x 2 loads instead of 16 x 2.  The octet kernel moved 8 TB/s through
// L1 (four taps of 32 B per output octet), which is where every re-reading kernel of this project has topped out.
struct UpBArgs {
  const float *x, *skip;
  __bf16* hl;
  int Cp, h, w, H, W, C1, C2;
  float sh, sw, inv_noct;
  int noct, PB;                  // octets per pixel, blocks per workgroup
  int BW, BH;                    // blocks per row / column of one image
  long nblk;                     // B * BH * BW
  unsigned* range_flag;          // nullable (UpArgs::range_flag)
};

__device__ __forceinline__ float4 up_sel(bool second, const float4 a, const float4 b) {
  return make_float4(second ? b.x : a.x, second ? b.y : a.y, second ? b.z : a.z, second ? b.w : a.w);
}

template <bool F16>
__global__ __launch_bounds__(256) void upsample_concat_split_2x2_kernel(UpBArgs p) {
  const int C = p.C1 + p.C2;
  long wg = blockIdx.x;
  {
    const long nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, i = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const long blk0 = wg * p.PB;                                   // wave-uniform
  const int bx0 = (int)(blk0 % p.BW);
  const long t0 = blk0 / p.BW;
  const int by0 = (int)(t0 % p.BH);
  const long b0 = t0 / p.BH;
  const int nitems = (int)(min((long)p.PB, p.nblk - blk0)) * p.noct;       // <= blockDim.x
  const bool live = (int)threadIdx.x < nitems;
  const int item = live ? (int)threadIdx.x : nitems - 1;
  const int dp = (int)(((float)item + 0.5f) * p.inv_noct);
  const int c = (item - dp * p.noct) * 8;
  int bx = bx0 + dp, by = by0;
  long b = b0;
  while (bx >= p.BW) { bx -= p.BW; ++by; }
  while (by >= p.BH) { by -= p.BH; ++b; }
  const int X = 2 * bx, Y = 2 * by;
  const bool is_up = c < p.C1, is_skip = !is_up && c < C;

  // source neighbourhood: rows yc0 .. yc0 + 2, columns xc0 .. xc0 + 2 (clamped); output (Y + j, X + i) uses rows
  // (dy_j, dy_j + 1) and columns (dx_i, dx_i + 1) of it, dy_0 = dx_0 = 0, dy_1, dx_1 in {0, 1}
  const float sy0 = p.sh * Y, sy1 = p.sh * (Y + 1), sx0 = p.sw * X, sx1 = p.sw * (X + 1);
  const int yc0 = (int)sy0, xc0 = (int)sx0;
  const int dy1 = min(max((int)sy1 - yc0, 0), 1), dx1 = min(max((int)sx1 - xc0, 0), 1);
  const float hy1[2] = {sy0 - (float)yc0, sy1 - (float)(yc0 + dy1)};
  const float wx1[2] = {sx0 - (float)xc0, sx1 - (float)(xc0 + dx1)};
  float4 n0[3][3], n1[3][3];
  if (is_up) {
    const float* base = p.x + b * (long)p.h * p.w * p.C1 + c;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int yy = min(yc0 + j, p.h - 1);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int xx = min(xc0 + i, p.w - 1);
        const float* r = base + ((long)yy * p.w + xx) * p.C1;
        n0[j][i] = ld4(r);
        n1[j][i] = ld4(r + 4);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int yy = min(Y + j, p.H - 1), xx = min(X + i, p.W - 1);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* r = p.skip + ((b * p.H + yy) * (long)p.W + xx) * p.C2 + (c - p.C1);
        n0[j][i] = is_skip ? ld4(r) : z;
        n1[j][i] = is_skip ? ld4(r + 4) : z;
      }
  }
  up_bf16x8 hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 va, vb;
      if (is_up) {
        const bool ry = j == 1 && dy1 == 1, rx = i == 1 && dx1 == 1;      // second row / column pair of the neighbourhood
        const float h1 = hy1[j], h0 = 1.0f - h1, w1 = wx1[i], w0 = 1.0f - w1;
        // rows (t, u) = (0, 1) or (1, 2); columns (l, r) = (0, 1) or (1, 2)
        const float4 tl0 = up_sel(rx, up_sel(ry, n0[0][0], n0[1][0]), up_sel(ry, n0[0][1], n0[1][1]));
        const float4 tr0 = up_sel(rx, up_sel(ry, n0[0][1], n0[1][1]), up_sel(ry, n0[0][2], n0[1][2]));
        const float4 bl0 = up_sel(rx, up_sel(ry, n0[1][0], n0[2][0]), up_sel(ry, n0[1][1], n0[2][1]));
        const float4 br0 = up_sel(rx, up_sel(ry, n0[1][1], n0[2][1]), up_sel(ry, n0[1][2], n0[2][2]));
        const float4 tl1 = up_sel(rx, up_sel(ry, n1[0][0], n1[1][0]), up_sel(ry, n1[0][1], n1[1][1]));
        const float4 tr1 = up_sel(rx, up_sel(ry, n1[0][1], n1[1][1]), up_sel(ry, n1[0][2], n1[1][2]));
        const float4 bl1 = up_sel(rx, up_sel(ry, n1[1][0], n1[2][0]), up_sel(ry, n1[1][1], n1[2][1]));
        const float4 br1 = up_sel(rx, up_sel(ry, n1[1][1], n1[2][1]), up_sel(ry, n1[1][2], n1[2][2]));
        va = up_lerp(tl0, tr0, bl0, br0, h0, h1, w0, w1);
        vb = up_lerp(tl1, tr1, bl1, br1, h0, h1, w0, w1);
      } else {
        va = n0[j][i];
        vb = n1[j][i];
      }
      const float f[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
    if constexpr (F16) ocv_range_note(p.range_flag, up_amax8(f));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        unsigned short hb_, lb_;
        ocv_split1<F16>(f[e], hb_, lb_);
        hi[2 * j + i][e] = __builtin_bit_cast(__bf16, hb_);
        lo[2 * j + i][e] = __builtin_bit_cast(__bf16, lb_);
      }
    }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (live && Y + j < p.H && X + i < p.W) {
        __bf16* dst = p.hl + ((b * p.H + Y + j) * (long)p.W + X + i) * 2 * p.Cp + (c >> 5) * 64 + (c & 31);
        *reinterpret_cast<up_bf16x8*>(dst) = hi[2 * j + i];
        *reinterpret_cast<up_bf16x8*>(dst + 32) = lo[2 * j + i];
      }
    }
}

// LDS form (at least 2x up-scaling, C1 a multiple of 32, C2 of 8): a workgroup owns a 16 x 32 tile of output pixels and
// 32 channels of the resized tensor; the 10 x 18 source pixels under the tile are loaded ONCE into LDS (23 KB, one
// 128-byte line per pixel) and every bilinear tap is an LDS read.  The 2 x 2 kernel still moved 2.25 taps of 32 B per
// output octet through L1 -- 6.3 TB/s, the level at which every re-reading kernel of this project has stalled; here L1
// carries the source once.  LDS reads are counted in lgkmcnt, not vmcnt, so the stores of one item never hold up the
// next item's taps.  blockIdx.y >= C1 / 32 are the skip-connection / pad channels: no resize, straight split.
// Measured (B = 16, four decoder launches): 0.74 ms against 0.96 for the 2 x 2 kernel; 240 x 320: 383 us = 4.8 TB/s of
// algorithmic traffic (torch's copy of the output alone: 5.1 TB/s).  64 channels per workgroup (46 KB, 3 workgroups per
// CU): 0.86 ms; 8- or 32-row tiles: 0.74 / 0.81 ms.
#ifndef OCV_ULC
#define OCV_ULC 32
#endif
#ifndef OCV_ULT_H
#define OCV_ULT_H 16
#endif
constexpr int ULT_H = OCV_ULT_H, ULT_W = 32, ULS_H = ULT_H / 2 + 2, ULS_W = 18, ULC = OCV_ULC;

struct UpLArgs {
  const float *x, *skip;
  __bf16* hl;
  int Cp, h, w, H, W, C1, C2;
  float sh, sw;
  int tiles_x, tiles_per_image, nup;     // nup = C1 / 64 resize chunks; further chunks: 64 skip / pad channels each
  unsigned* range_flag;                  // nullable (UpArgs::range_flag)
};

template <bool F16>
__global__ __launch_bounds__(256) void upsample_concat_split_lds_kernel(UpLArgs p) {
  __shared__ __attribute__((aligned(16))) float4 src[ULS_H * ULS_W][ULC / 4];
  const int tid = threadIdx.x;
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, i = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const long b = wg / p.tiles_per_image;
  const int t2 = wg - (int)b * p.tiles_per_image;
  const int ty = t2 / p.tiles_x, tx = t2 - ty * p.tiles_x;
  const int Y0 = ty * ULT_H, X0 = tx * ULT_W;
  const int chunk = blockIdx.y;
  if (chunk < p.nup) {
    const int ys0 = (int)(p.sh * Y0), xs0 = (int)(p.sw * X0);
    const float* xb = p.x + b * (long)p.h * p.w * p.C1 + chunk * ULC;
    for (int idx = tid; idx < ULS_H * ULS_W * (ULC / 4); idx += 256) {
      const int px = idx / (ULC / 4), f = idx % (ULC / 4);
      const int sy = min(ys0 + px / ULS_W, p.h - 1), sx = min(xs0 + px % ULS_W, p.w - 1);
      src[px][f] = ld4(xb + ((long)sy * p.w + sx) * p.C1 + 4 * f);
    }
    __syncthreads();
#pragma unroll 2
    for (int item = tid; item < ULT_H * ULT_W * (ULC / 8); item += 256) {
      const int oct = item % (ULC / 8), pix = item / (ULC / 8);
      const int X = X0 + (pix & (ULT_W - 1)), Y = Y0 + (pix >> 5);
      if (Y >= p.H || X >= p.W) continue;
      const float sy = p.sh * Y, sx = p.sw * X;
      const int y0 = (int)sy, x0 = (int)sx;
      const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
      const float h1 = sy - (float)y0, h0 = 1.0f - h1, w1 = sx - (float)x0, w0 = 1.0f - w1;
      const int r0 = (y0 - ys0) * ULS_W, r1 = (y1 - ys0) * ULS_W, c0 = x0 - xs0, c1 = x1 - xs0;
      const float4 va = up_lerp(src[r0 + c0][2 * oct], src[r0 + c1][2 * oct], src[r1 + c0][2 * oct], src[r1 + c1][2 * oct], h0, h1, w0, w1);
      const float4 vb = up_lerp(src[r0 + c0][2 * oct + 1], src[r0 + c1][2 * oct + 1], src[r1 + c0][2 * oct + 1], src[r1 + c1][2 * oct + 1], h0, h1, w0, w1);
      const float f[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
    if constexpr (F16) ocv_range_note(p.range_flag, up_amax8(f));
      up_bf16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        unsigned short hb_, lb_;
        ocv_split1<F16>(f[e], hb_, lb_);
        hi[e] = __builtin_bit_cast(__bf16, hb_);
        lo[e] = __builtin_bit_cast(__bf16, lb_);
      }
      const int c = chunk * ULC + oct * 8;
      __bf16* dst = p.hl + ((b * p.H + Y) * (long)p.W + X) * 2 * p.Cp + (c >> 5) * 64 + (c & 31);
      *reinterpret_cast<up_bf16x8*>(dst) = hi;
      *reinterpret_cast<up_bf16x8*>(dst + 32) = lo;
    }
    return;
  }
  // skip-connection and pad channels [cbase, cbase + 64) of the tile's pixels
  const int cbase = p.C1 + (chunk - p.nup) * ULC;
  const int noct = min(ULC, p.Cp - cbase) >> 3;                  // octets of this chunk (Cp % 32 == 0)
  const int C = p.C1 + p.C2;
  for (int item = tid; item < ULT_H * ULT_W * noct; item += 256) {
    const int pix = item / noct, oct = item - pix * noct;
    const int X = X0 + (pix & (ULT_W - 1)), Y = Y0 + (pix >> 5);
    if (Y >= p.H || X >= p.W) continue;
    const int c = cbase + oct * 8;
    float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
    if (c < C) {
      const float* sp = p.skip + ((b * p.H + Y) * (long)p.W + X) * p.C2 + (c - p.C1);
      va = ld4(sp);
      vb = ld4(sp + 4);
    }
    const float f[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
    if constexpr (F16) ocv_range_note(p.range_flag, up_amax8(f));
    up_bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned short hb_, lb_;
      ocv_split1<F16>(f[e], hb_, lb_);
      hi[e] = __builtin_bit_cast(__bf16, hb_);
      lo[e] = __builtin_bit_cast(__bf16, lb_);
    }
    __bf16* dst = p.hl + ((b * p.H + Y) * (long)p.W + X) * 2 * p.Cp + (c >> 5) * 64 + (c & 31);
    *reinterpret_cast<up_bf16x8*>(dst) = hi;
    *reinterpret_cast<up_bf16x8*>(dst + 32) = lo;
  }
}

}  // namespace

extern "C" int ocv_upsample_concat_split_x_fwd(const float* x, int h, int w, int C1, const float* skip, int C2, void* out_hl,
                                               int f16, int B, int H, int W, ocv_stream_t stream) {
  OCV_CHECK_ARG(f16 == 0 || f16 == 1, "ocv_upsample_concat_split_fwd: f16 must be 0 (bf16 pairs) or 1 (fp16 pairs)");
  OCV_CHECK_ARG(x && out_hl, "ocv_upsample_concat_split_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && h >= 1 && w >= 1 && H >= 1 && W >= 1 && C1 >= 4 && C1 % 4 == 0, "ocv_upsample_concat_split_fwd: bad sizes (C1 must be a multiple of 4)");
  OCV_CHECK_ARG(skip == nullptr ? C2 == 0 : (C2 >= 4 && C2 % 4 == 0), "ocv_upsample_concat_split_fwd: C2 must be a multiple of 4 (0 without a skip tensor)");
  OCV_CHECK_ARG(ocv_aligned16(x) && ocv_aligned16(skip) && ocv_aligned16(out_hl), "ocv_upsample_concat_split_fwd: operands must be 16-byte aligned");
  const int Cp = (C1 + C2 + 31) / 32 * 32;
  UpArgs a{x, skip, (unsigned short*)out_hl, Cp, h, w, H, W, C1, C2,
           H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f,
           (long)B * H * W * (Cp / 4), f16 ? ocv_range_flag_current() : nullptr};
  if (C1 % ULC == 0 && C2 % 8 == 0 && a.sh <= 0.5f && a.sw <= 0.5f) {
    UpLArgs g{x, skip, (__bf16*)out_hl, Cp, h, w, H, W, C1, C2, a.sh, a.sw, (W + ULT_W - 1) / ULT_W, 0, C1 / ULC, a.range_flag};
    g.tiles_per_image = ((H + ULT_H - 1) / ULT_H) * g.tiles_x;
    const long nb = (long)B * g.tiles_per_image;
    const int nchunk = g.nup + (Cp - C1 + ULC - 1) / ULC;
    OCV_CHECK_ARG(nb < (1L << 31) && nchunk < 65536, "ocv_upsample_concat_split_fwd: tensor too large");
    if (f16) hipLaunchKernelGGL(upsample_concat_split_lds_kernel<true>, dim3((unsigned)nb, (unsigned)nchunk), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(upsample_concat_split_lds_kernel<false>, dim3((unsigned)nb, (unsigned)nchunk), dim3(256), 0, (hipStream_t)stream, g);
    OCV_CHECK_LAUNCH("ocv_upsample_concat_split_fwd");
    return 0;
  }
  if (C1 % 8 == 0 && C2 % 8 == 0 && Cp / 8 <= 256 && a.sh <= 1.0f && a.sw <= 1.0f) {
    UpBArgs g{x, skip, (__bf16*)out_hl, Cp, h, w, H, W, C1, C2, a.sh, a.sw, 0.f, Cp / 8, 1, (W + 1) / 2, (H + 1) / 2, 0, a.range_flag};
    g.inv_noct = 1.0f / (float)g.noct;
    g.nblk = (long)B * g.BH * g.BW;
    int threads = 256;                                            // workgroup size with the fewest idle lanes
    for (int t = 64; t <= 256; t += 64)
      if (t >= g.noct && (t / g.noct) * g.noct * threads > (threads / g.noct) * g.noct * t) threads = t;
    g.PB = threads / g.noct;
    const long nb = (g.nblk + g.PB - 1) / g.PB;
    OCV_CHECK_ARG(nb < (1L << 31), "ocv_upsample_concat_split_fwd: tensor too large");
    if (f16) hipLaunchKernelGGL(upsample_concat_split_2x2_kernel<true>, dim3((unsigned)nb), dim3(threads), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(upsample_concat_split_2x2_kernel<false>, dim3((unsigned)nb), dim3(threads), 0, (hipStream_t)stream, g);
    OCV_CHECK_LAUNCH("ocv_upsample_concat_split_fwd");
    return 0;
  }
  if (C1 % 8 == 0 && C2 % 8 == 0 && Cp / 8 <= 256 * UP8_ITEMS) {
    Up8Args g{x, skip, (__bf16*)out_hl, Cp, h, w, H, W, C1, C2, a.sh, a.sw, 0.f, Cp / 8, 1, (long)B * H * W, a.range_flag};
    g.inv_noct = 1.0f / (float)g.noct;
    g.PB = (256 * UP8_ITEMS) / g.noct;                           // a workgroup's items fit one pass of its threads
    const long nb = (g.npix + g.PB - 1) / g.PB;
    OCV_CHECK_ARG(nb < (1L << 31), "ocv_upsample_concat_split_fwd: tensor too large");
    if (f16) hipLaunchKernelGGL(upsample_concat_split8_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(upsample_concat_split8_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g);
    OCV_CHECK_LAUNCH("ocv_upsample_concat_split_fwd");
    return 0;
  }
  const long blocks = (a.total + 256L * UP_ITEMS - 1) / (256L * UP_ITEMS);
  OCV_CHECK_ARG(blocks < (1L << 31), "ocv_upsample_concat_split_fwd: tensor too large");
  if (f16) hipLaunchKernelGGL(upsample_concat_split_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(upsample_concat_split_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  OCV_CHECK_LAUNCH("ocv_upsample_concat_split_fwd");
  return 0;
}

extern "C" int ocv_upsample_concat_split_fwd(const float* x, int h, int w, int C1, const float* skip, int C2, void* out_hl,
                                             int B, int H, int W, ocv_stream_t stream) {
  return ocv_upsample_concat_split_x_fwd(x, h, w, C1, skip, C2, out_hl, 0, B, H, W, stream);
}
