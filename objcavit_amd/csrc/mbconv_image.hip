// Expand 1x1 (+BN+SiLU) and depthwise k x k (+BN+SiLU, + the squeeze-excite pooling sums) of a LATE EfficientNet MBConv block
// in ONE launch (round 5; row N1 of SURVEY.md section 8: conv_pw/bn1/act1 -> conv_dw/bn2/act2 of the
// InvertedResidual blocks the reference runs through its hub backbone, modules/DenseFeatureExtractor.py:18-27,149).
//
// Where csrc/mbconv_fused.hip (stages 2 - 4: large maps, Cin <= 64) tiles the map and recomputes the expand GEMM on every tile's
// halo, the late stages are small maps with wide layers -- 30 x 40 and 15 x 20 at 768 ... 3072 expanded channels, B5's stages 4 - 7,
// stride 1 -- where the two-launch form moved the expanded tensor through HBM twice (stage 6 at bs 16: expand 30 us + depthwise
// 34 us + pooling 9 us for 35 MB written and read back) in launches too short for their own load -> multiply -> store chain.
// Here a workgroup owns (image, band of BR output rows -- the WHOLE 15 x 20 image, 8 rows of a 30 x 40 one --, chunk of 32
// expanded channels):
//   phase A  [pixels of the band + halo rows] x [Cin] x [Cin x 32] on v_mfma_f32_32x32x16_bf16 with the numerics of
//            csrc/pointwise_split.hip (activation rows split hi / lo on the fly, packed two-term weights, hi*hi + hi*lo + lo*hi,
//            fp32 accumulate from the bias).  The band's pixels are CONTIGUOUS rows of the NHWC map (full-width bands), so the A
//            operand is a plain [M x Cin] matrix: staged through LDS 32 channels at a time (coalesced rows, split once per workgroup);
//            wavefront w takes M tiles w, w + 5, ...; the chunk's weight fragments come from L1 / L2 (4 KB per slab, the same bytes
//            for all five wavefronts).  bias + SiLU -> LDS as [pixel][32 channels] fp32 (the staging area's memory); halo rows
//            outside the image are ZERO (the depthwise convolution pads the EXPANDED tensor).
//   phase B  thread = (channel quad, output column, row group) slides down its columns' rows: K ds_read_b128 per input row (taps left
//            or right of the image read a zero pixel: one address select per tap, no column padding in LDS), K x K x 4 FMAs per row
//            into the running outputs; depthwise weights of the quad in VGPRs (fetched behind phase A).  bias + SiLU, 16-byte
//            stores, and the band's per-channel sums for the squeeze-excite mean through LDS in a fixed order: part[b][band][c]
//            (ocv_se_gate_partials_fwd reads them with tiles = bands per image: 1 - 4 rows instead of the depthwise kernel's 10 - 75).
// Halo recompute of the GEMM: none at 15 x 20 (one band), 1.25x (k = 3) / 1.5x (k = 5) with bands of 8 rows at 30 x 40.
// All chunks of a band run back to back on one XCD: its rows come from HBM once and from that XCD's L2 after.
#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct MIArgs {
  const float *x, *be, *wd, *bd;
  const __bf16* wp;                 // packed expand weights (csrc/pointwise_split.hip, w_frag)
  float *y, *part;
  int Cin, mid, nchunks, nsteps;    // nsteps = Cin / 16
};

template <int K, int H, int W, int BR>
struct MIGeom {
  static constexpr int PAD = K / 2, RB = BR + 2 * PAD, NB = (H + BR - 1) / BR;
  static constexpr int PIX = RB * W, MT = (PIX + 31) / 32;
  static constexpr int NW = 5, NT = 64 * NW;                 // 320 threads = 8 channel quads x 40 (column, row group) slots
  static constexpr int NMW = (MT + NW - 1) / NW;             // M tiles per wavefront
  static constexpr int RG = 40 / W, NR = (BR + RG - 1) / RG; // row groups, output rows per thread
  static constexpr int NLI = NR + K - 1;                     // input rows a thread walks
  static constexpr int ZPIX = PIX;                           // the zero pixel behind the tile
  static constexpr int APITCH = 144;                         // bytes per STAGED row of phase A: 32 channels hi | 32 lo | 16 pad (conflict-free b128 reads)
  static constexpr int NP = (PIX + 39) / 40;                 // staging passes: 320 threads move 40 rows x 128 B per pass
  static constexpr int TILE_BYTES = (PIX + 1) * 128, STAGE_BYTES = MT * 32 * APITCH;
  static constexpr int LDS_FLOATS = (TILE_BYTES > STAGE_BYTES ? TILE_BYTES : STAGE_BYTES) / 4;
  static_assert(40 % W == 0 && RG * W == 40, "phase B maps 40 (column, row group) slots");
  static_assert(LDS_FLOATS >= 40 * 32, "the pooling reduction reuses the tile");
};

__device__ __forceinline__ void mi_split8(const float4 u, const float4 v, bf16x8& hi, bf16x8& lo) {
  const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)f[i];
    hi[i] = h;
    lo[i] = (__bf16)(f[i] - (float)h);
  }
}

__device__ __forceinline__ float4 mi_fma4(const float4 w, const float4 v, float4 a) {
  a.x = fmaf(w.x, v.x, a.x); a.y = fmaf(w.y, v.y, a.y); a.z = fmaf(w.z, v.z, a.z); a.w = fmaf(w.w, v.w, a.w);
  return a;
}

template <int K, int H, int W, int BR>
__global__ __launch_bounds__(320, 3) void mbconv_image_kernel(MIArgs p) {
  using G = MIGeom<K, H, W, BR>;
  extern __shared__ __attribute__((aligned(16))) float e[];        // [PIX + 1 pixels][32 channels]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int t = wg / p.nchunks, chunk = wg - t * p.nchunks;
  const int b = t / G::NB, band = t - b * G::NB;
  const int y0 = band * BR, y1 = min(H, y0 + BR);
  const int ya = max(0, y0 - G::PAD), yb = min(H, y1 + G::PAD);     // image rows the band reads
  const int rtop = ya - (y0 - G::PAD);                              // their first row inside the LDS tile
  const int Mb = (yb - ya) * W;                                     // GEMM rows of this band
  const int n0 = chunk * 32;

  // ---------------- phase A: expand GEMM of the band's pixels, K slab by K slab of 32 channels
  // The band's rows are staged through LDS once per workgroup and slab -- coalesced 128-byte row segments (8 lanes per row), split
  // to bf16 hi | lo while they are stored -- and every wavefront reads the A fragments of its M tiles from there.  (The first
  // version read A straight from global memory in operand order, 32 rows of 32 bytes per instruction: correct, and 5 % SLOWER
  // end to end than the two-launch form -- per-line tag work, the cost the packed weight layout of pointwise_split.hip removed
  // for B.)  The staging area IS the depthwise tile's memory: nothing is written there before the K loop has ended.
  {
    const float bev = p.be != nullptr ? p.be[n0 + l31] : 0.f;
    const float* xb = p.x + ((long)b * H + ya) * W * p.Cin;
    const __bf16* wf = p.wp + ((long)chunk * p.nsteps * 2) * 512 + lane * 8;       // + (s * 2 + part) * 512
    char* stage = reinterpret_cast<char*>(e);
    f32x16 acc[G::NMW];
#pragma unroll
    for (int i = 0; i < G::NMW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = bev;
    const int srow = tid >> 3, c4 = tid & 7;                      // staging role: row srow + 40 pass, channels 4 c4 .. + 3 of the slab
    const float* src = xb + (long)srow * p.Cin + 4 * c4;
    char* sdst = stage + srow * G::APITCH + c4 * 8;
    float4 ra[G::NP];
    auto fetch = [&](int slab) {
      const bool cok = slab * 32 + 4 * c4 < p.Cin;                 // the last slab of an odd step count is half a slab
#pragma unroll
      for (int j = 0; j < G::NP; ++j)
        ra[j] = (cok && srow + 40 * j < Mb) ? ld4(src + (long)(40 * j) * p.Cin + slab * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store = [&]() {
#pragma unroll
      for (int j = 0; j < G::NP; ++j) {
        if (srow + 40 * j < G::MT * 32) {
          typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
          const float f[4] = {ra[j].x, ra[j].y, ra[j].z, ra[j].w};
          bf16x4 h, l;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const __bf16 hb = (__bf16)f[q4];
            h[q4] = hb;
            l[q4] = (__bf16)(f[q4] - (float)hb);
          }
          *reinterpret_cast<bf16x4*>(sdst + (40 * j) * G::APITCH) = h;
          *reinterpret_cast<bf16x4*>(sdst + (40 * j) * G::APITCH + 64) = l;
        }
      }
    };
    const int nslab = (p.nsteps + 1) >> 1;
    fetch(0);
    for (int slab = 0; slab < nslab; ++slab) {
      const int s0 = 2 * slab;
      const bool two = s0 + 1 < p.nsteps;                           // wave-uniform
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int ss = min(s0 + s, p.nsteps - 1);
        bh[s] = *reinterpret_cast<const bf16x8*>(wf + (long)ss * 1024);
        bl[s] = *reinterpret_cast<const bf16x8*>(wf + (long)ss * 1024 + 512);
      }
      if (slab > 0) __syncthreads();                                // every wavefront is past the previous slab's fragment reads
      store();
      __syncthreads();
      if (slab + 1 < nslab) fetch(slab + 1);                        // in flight under this slab's MFMAs
#pragma unroll
      for (int i = 0; i < G::NMW; ++i) {
        if ((wave + G::NW * i) * 32 < Mb) {                         // wave-uniform
          const char* ap = stage + ((wave + G::NW * i) * 32 + l31) * G::APITCH + hh * 16;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (s == 0 || two) {
              const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap + s * 32);
              const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + s * 32 + 64);
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s], acc[i], 0, 0, 0);
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[s], acc[i], 0, 0, 0);
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[s], acc[i], 0, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();                                                // the staging area becomes the depthwise tile
    // rows of the tile that lie outside the image, and the zero pixel
    for (int i = tid; i < (G::PIX + 1) * 8; i += G::NT) {
      const int pix = i >> 3;
      const int r = pix / W;
      if (pix == G::ZPIX || r < rtop || r >= rtop + (yb - ya)) *reinterpret_cast<float4*>(e + pix * 32 + 4 * (i & 7)) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < G::NMW; ++i) {
      const int m0 = (wave + G::NW * i) * 32;
      if (m0 < Mb) {
        float* dst = e + (rtop * W + m0 + 4 * hh) * 32 + l31;        // accumulator row r -> band pixel m0 + acc_row(r, hh)
        if (m0 + 32 <= Mb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[acc_row(r, 0) * 32] = fast_silu(acc[i][r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (m0 + acc_row(r, hh) < Mb) dst[acc_row(r, 0) * 32] = fast_silu(acc[i][r]);
        }
      }
    }
  }

  // ---------------- phase B: depthwise over the LDS tile
  const int q = tid & 7, slot = tid >> 3;                             // slot 0 .. 39
  const int col = slot % W, rg = slot / W;
  const int o0 = rg * G::NR;                                          // first output row of this thread (band-relative)
  const int cq = n0 + 4 * q;
  const float4 bdw = p.bd != nullptr ? ld4(p.bd + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  float4 acc[G::NR];
#pragma unroll
  for (int o = 0; o < G::NR; ++o) acc[o] = bdw;
  // The taps are walked in column groups of three (k = 5: columns 0 - 2, then 3 - 5, the sixth a phantom that reads the zero pixel):
  // a group's K x 3 weight quads are the only ones in VGPRs -- 60 registers instead of 100; with all 25 live the kernel spilled ~100
  // registers and fell to two wavefronts per SIMD.  The group loop is a RUNTIME loop (one copy of the row code; nothing of the
  // next group can be hoisted into this one); every tile element is still read once per output column it feeds.
  constexpr int KG = 3;
#pragma unroll 1
  for (int kx0 = 0; kx0 < K; kx0 += KG) {
    float4 wdw[K][KG];
    int coff[KG];                                                      // float offset of tap kx0 + j inside a tile row, or -1
#pragma unroll
    for (int j = 0; j < KG; ++j) {
      const int kx = kx0 + j, xx = col + kx - G::PAD;
      coff[j] = (kx < K && xx >= 0 && xx < W) ? xx * 32 + 4 * q : -1;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) wdw[ky][j] = ld4(p.wd + (unsigned)((ky * K + min(kx, K - 1)) * p.mid + cq));
    }
    // input rows one at a time, the next row's reads in flight while this row's K x KG x 4 FMAs issue; a scheduling fence per row
    // keeps the unrolled loop from hoisting every row's reads to the top
    float4 v[2][KG];
    auto fetch = [&](int set, int li) {
      const int row = o0 + li;                                         // tile row (rows >= RB feed outputs that are never stored)
      const int rbase = row < G::RB ? row * (W * 32) : -1;
#pragma unroll
      for (int j = 0; j < KG; ++j) {
        const int off = (rbase >= 0 && coff[j] >= 0) ? rbase + coff[j] : G::ZPIX * 32 + 4 * q;
        v[set][j] = *reinterpret_cast<const float4*>(e + off);
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int li = 0; li < G::NLI; ++li) {
      if (li + 1 < G::NLI) fetch((li + 1) & 1, li + 1);
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int d = li - ky;
        if (d >= 0 && d < G::NR) {
#pragma unroll
          for (int j = 0; j < KG; ++j) acc[d] = mi_fma4(wdw[ky][j], v[li & 1][j], acc[d]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
  float* orow = p.y + (((long)b * H + y0 + o0) * W + col) * p.mid + cq;
  const long ostep = (long)W * p.mid;
#pragma unroll
  for (int o = 0; o < G::NR; ++o, orow += ostep) {
    if (o0 + o < BR && y0 + o0 + o < H) {
      float4 r = acc[o];
      r.x = fast_silu(r.x); r.y = fast_silu(r.y); r.z = fast_silu(r.z); r.w = fast_silu(r.w);
      *reinterpret_cast<float4*>(orow) = r;
      psum.x += r.x; psum.y += r.y; psum.z += r.z; psum.w += r.w;
    }
  }
  // per-channel sum of the band (squeeze-excite pooling partial), fixed order: slots ascending
  __syncthreads();
  *reinterpret_cast<float4*>(e + slot * 32 + 4 * q) = psum;
  __syncthreads();
  if (tid < 32) {
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < 40; ++j) s += e[j * 32 + tid];
    p.part[((long)b * G::NB + band) * p.mid + n0 + tid] = s;
  }
}

template <int K, int H, int W, int BR>
int mi_launch(const MIArgs& a, int B, hipStream_t st) {
  using G = MIGeom<K, H, W, BR>;
  constexpr size_t LDS = (size_t)G::LDS_FLOATS * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)mbconv_image_kernel<K, H, W, BR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    attr = true;
  }
  const long grid = (long)B * G::NB * a.nchunks;
  hipLaunchKernelGGL((mbconv_image_kernel<K, H, W, BR>), dim3((unsigned)grid), dim3(G::NT), LDS, st, a);
  OCV_CHECK_LAUNCH("ocv_mbconv_image_fwd");
  return 0;
}

}  // namespace

// bands per image of the shapes this kernel is built for (= the `tiles` of its pooling partials), 0 = shape not covered
extern "C" int ocv_mbconv_image_tiles(int H, int W, int k) {
  if (H == 15 && W == 20 && (k == 3 || k == 5)) return 1;
  if (H == 30 && W == 40 && k == 3) return 4;
  if (H == 30 && W == 40 && k == 5) return 4;
  return 0;
}

extern "C" int ocv_mbconv_image_fwd(const float* x, const void* w_packed, const float* bias_expand, const float* w_dw,
                                    const float* bias_dw, float* y, float* part, int B, int H, int W, int Cin, int mid, int k,
                                    ocv_stream_t stream) {
  OCV_CHECK_ARG(x && w_packed && w_dw && y && part, "ocv_mbconv_image_fwd: null pointer");
  OCV_CHECK_ARG(ocv_mbconv_image_tiles(H, W, k) > 0, "ocv_mbconv_image_fwd: built for 15 x 20 and 30 x 40 maps with k = 3 or 5 (got %d x %d, k = %d)", H, W, k);
  OCV_CHECK_ARG(B >= 1 && Cin >= 16 && Cin % 16 == 0, "ocv_mbconv_image_fwd: Cin must be a positive multiple of 16 (got %d)", Cin);
  OCV_CHECK_ARG(mid >= 32 && mid % 32 == 0, "ocv_mbconv_image_fwd: expanded channels must be a multiple of 32 (got %d)", mid);
  OCV_CHECK_ARG(ocv_aligned16(x) && ocv_aligned16(w_packed) && ocv_aligned16(w_dw) && ocv_aligned16(bias_dw) && ocv_aligned16(y),
                "ocv_mbconv_image_fwd: operands must be 16-byte aligned");
  OCV_CHECK_ARG((long)B * 4 * (mid / 32) < (1L << 31) && (long)H * W * Cin * 4 < (1L << 31), "ocv_mbconv_image_fwd: too many work items");
  MIArgs a{x, bias_expand, w_dw, bias_dw, (const __bf16*)w_packed, y, part, Cin, mid, mid / 32, Cin / 16};
  hipStream_t st = (hipStream_t)stream;
  if (H == 15 && k == 5) return mi_launch<5, 15, 20, 15>(a, B, st);
  if (H == 15) return mi_launch<3, 15, 20, 15>(a, B, st);
  if (k == 5) return mi_launch<5, 30, 40, 8>(a, B, st);
  return mi_launch<3, 30, 40, 8>(a, B, st);
}
