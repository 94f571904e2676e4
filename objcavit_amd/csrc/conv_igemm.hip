// Implicit-GEMM convolution (k = 1 or 3, stride 1, "same" zero padding) on
// NHWC fp32 activations for gfx950, computed on the bf16 matrix cores with
// SPLIT-bf16 operands and fp32 accumulation:
//
//     a = a_hi + a_lo,  b = b_hi + b_lo   (each part bf16, a_lo = bf16(a - a_hi))
//     a * b  ~=  a_hi b_hi + a_hi b_lo + a_lo b_hi            (3 MFMAs, fp32 accumulate)
//
// hi + lo carries 16 significant bits, the dropped a_lo b_lo term is 2^-18
// relative, so a product is good to ~1e-5 relative and a K-long sum to ~1e-6 of
// its magnitude -- two orders inside the 1e-3 depth tolerance of the path and
// measured end to end in the parity tests -- while the three
// v_mfma_f32_32x32x16_bf16 issues cost 96 cycles per 32x32x16 block against
// 512 cycles for the eight exact v_mfma_f32_32x32x2_f32 issues: 5.3x the fp32
// matrix rate (gfx950 has no xf32/TF32 path).  This is row N1 of SURVEY.md section 8:
// the UNet decoder's 3x3 convolutions are 83 % of the forward's FLOPs
// (modules/DenseFeatureExtractor.py:37-42,97,104-116) and MIOpen's fp32
// implicit GEMM already sits at 85 % of the fp32 matrix peak there.
//
// GEMM view: M = B*H*W output pixels, N = Cout, K = taps x Cin.  The A operand
// is gathered on the fly (tap offset, zero outside the image) from one or two
// NHWC tensors -- a second tensor acts as a virtual channel concat, which is how
// UpSampleWithSkip feeds [upsampled, skip] without materialising the cat --
// and split into hi/lo in registers on its way to LDS.  Weights are split once
// on the host into two bf16 arrays laid out [tap][Cout][Cin rounded up to 32].
//
// Tile: 256 pixels x 128 channels per workgroup; K advances 32 channels of one tap per step (channel chunk outer,
// tap inner: the nine taps of a chunk re-read the same L2-resident lines) through a double-buffered LDS image
// (A_hi, A_lo, B_hi, B_lo; rows padded to 80 bytes so the 16-byte fragment reads of 16 consecutive rows land in
// 16 different bank slots).
//
// The 8 wavefronts are SPECIALISED (first version: every wavefront did everything; PMC: 8.5 VALU instructions per
// MFMA, matrix pipe 32 % busy -- the address / bounds / fp32->hi,lo conversion work of a step ran in lock-step on
// all waves between two barriers and could not overlap the MFMAs):
//   waves 0-3  CONSUMERS: 2 x 2 over the tile, 128 x 64 outputs each (8 accumulators); per step 24 ds_read_b128 and
//              48 MFMAs, nothing else.
//   waves 4-7  PRODUCERS: gather the next A chunk (4 rows x 32 B per lane, out-of-image taps read a zero page, no
//              selects), split it into hi/lo, fetch the pre-split weight chunk, write the other LDS buffer.
// One consumer and one producer wave share each SIMD: matrix pipe and VALU run side by side; one barrier per step.
// Workgroup ids are remapped so that the N-tiles of one pixel tile run on the same XCD and share its L2.
#include <stdlib.h>

#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cv_h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cv_h16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 ti4 __attribute__((ext_vector_type(4)));

constexpr int CBM = 256, CBN = 128, CBK = 32;
constexpr int ROWB = 80;                                  // bytes per LDS row (64 data + 16 pad)
constexpr int A_BYTES = CBM * ROWB, B_BYTES = CBN * ROWB;
constexpr int BUF_BYTES = 2 * A_BYTES + 2 * B_BYTES;      // 61440

struct ConvArgs {
  const float* x1; const float* x2;     // NHWC fp32; x2 (nullable) is concatenated after x1's channels
  const __bf16* xhl;                    // OR the input already split, "hl32" layout (see hl_index)   (DMA kernel)
  __bf16* yhl;                          // optional split copy of the output in the same layout (for the next convolution)
  int Cpo;                              // Cout rounded up to 32 (channel blocks of the split output)
  const __bf16* whi; const __bf16* wlo; // [taps][Cout][Cp]
  const float* bias; const float* res; float* y;
  int C1, C2, Cin, Cp, Cout, H, W, ks, act;
  long M;
  int mtiles, ntiles;
  int ksplit;                           // conv_split_dma_kernel: 1, or 2 = the channel chunks are halved between two workgroups
                                        // per tile, each writing raw fp32 partial sums to y + khalf * M * Cout (bias / act /
                                        // residual / split output then belong to conv_splitk_finish_kernel)
  int zbatch;                           // conv_split_dma_kernel: > 1 = that many independent GEMMs in one launch (the 16
  long xz_bytes, wz_bytes;              // Winograd positions): operand z at xhl + z xz_bytes / whi, wlo + z wz_bytes, raw fp32
                                        // result at y + (z ksplit + khalf) * M * Cout
  int zflat;                            // conv_split_dma_kernel, 1 x 1 only: > 0 = the zbatch GEMMs are the consecutive pieces of ONE
                                        // K axis (the patch embedding's 16 patch rows): its zbatch * Cp / 32 steps are cut in zflat
                                        // equal parts, one workgroup per (tile, part), raw fp32 result of part k at y + k * M * Cout
  int f16;                              // conv_split_dma_kernel: != 0 = the operands are fp16 (hi, lo) pairs (conv_split_dma_kernel<true>)
                                        // and so is the split copy of the output (yhl)
  const float* oscale;                  // nullable [Cout]: raw accumulators are multiplied by it before bias / activation (the
                                        // per-output-channel power of two the fp16 weights were scaled by, inverted)
  unsigned* range_flag;                 // nullable: the armed range-guard word (common.hpp ocv_range_note); read only where yhl holds fp16 pairs
  unsigned rowpitch;                    // conv_split_dma_kernel: bytes between consecutive rows of the (B H) x W pixel grid of the
                                        // split input when they are not dense (0 = dense, W * 4 Cp); the patch embedding reads
                                        // every 16th image row of the feature map as one GEMM row grid this way
  int gpt, gpt_inv;                     // conv_split_dma_kernel, 3 x 3 only: > 0 = PACKED TAPS (round 6).  The K axis is the nine taps'
  int Kp;                               // REAL channel granules (8 channels = 16 bytes of hi, 16 of lo) laid end to end, gpt = ceil(Cin / 8)
                                        // per tap, instead of nine chunks of Cp channels: a 32-channel K step holds four granules that may
                                        // belong to two taps (two pixels).  24 input channels: 27 granules = 7 steps instead of 9; 40: 45
                                        // granules = 12 steps instead of 18 (their Cp = 32 / 64 multiply 25 % / 37 % zeros).  The weights
                                        // are ONE [Cout][Kp] matrix in the same granule order, Kp = steps x 32; gpt_inv = a 16.16
                                        // reciprocal of gpt, exact for every granule index of the launch (checked on the host).
};

// 16-byte global load (compiler-visible: hipcc tracks it and inserts the s_waitcnt before the first use).
// An earlier version issued these from inline asm with hand-counted waits to keep two register stages in flight;
// the register allocator is free to COPY such a destination register before the load has landed (it did, at the loop
// back-edge), which reads stale data -- a silent, history-dependent corruption.  Never hide an in-flight load.
__device__ __forceinline__ f32x4 gload16(const void* p) { return *reinterpret_cast<const f32x4*>(p); }

__device__ __forceinline__ void split4(const f32x4 v, __bf16* hi, __bf16* lo) {
  const float f[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 h = (__bf16)f[i];
    hi[i] = h;
    lo[i] = (__bf16)(f[i] - (float)h);
  }
}

// The same split with FP16 terms (round 4: the decoder's convolutions as two-term fp16 products, 2^-22 instead of 2^-17): hi =
// fp16(v), lo = fp16(v - hi), UNSCALED low term (one accumulator in the GEMM kernel): for |v| below ~0.1 the low term is an fp16
// subnormal (honoured by the MFMA), i.e. the pair's ABSOLUTE error has a floor of 2^-25 ~ 3e-8 -- harmless next to O(1)
// activations, and the reason the weights are scaled per output channel (oscale) while activations are range-checked, not scaled.
// Values beyond +-65504 become inf (loud), never clipped.
__device__ __forceinline__ void split4h(const f32x4 v, _Float16* hi, _Float16* lo) {
  const float f[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const _Float16 h = (_Float16)f[i];
    hi[i] = h;
    lo[i] = (_Float16)(f[i] - (float)h);
  }
}
// two-byte (hi, lo) parts of four values as raw bits, either element type
template <bool F16>
__device__ __forceinline__ void split4_bits(const f32x4 v, unsigned short* hi, unsigned short* lo) {
  if constexpr (F16) split4h(v, reinterpret_cast<_Float16*>(hi), reinterpret_cast<_Float16*>(lo));
  else split4(v, reinterpret_cast<__bf16*>(hi), reinterpret_cast<__bf16*>(lo));
}
__device__ __forceinline__ void split4_bits(const f32x4 v, unsigned short* hi, unsigned short* lo, bool f16) {
  if (f16) split4_bits<true>(v, hi, lo);
  else split4_bits<false>(v, hi, lo);
}
// hi + lo of four consecutive (hi at p, lo at p + lo_off) two-byte pairs, either element type.  The two 8-byte loads are issued
// UNCONDITIONALLY and only the conversion depends on the type: a branch around a load makes hipcc wait for every load on its own
// (the first version of this function did, and the Winograd input transform went from 95 to 120 us per launch).
__device__ __forceinline__ f32x4 join4_bits(const void* p, int lo_off, bool f16) {
  const uint2 hb = *reinterpret_cast<const uint2*>(p);
  const uint2 lb = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + lo_off);
  f32x4 r;
  if (f16) {
    const cv_h16x4 h = __builtin_bit_cast(cv_h16x4, hb), l = __builtin_bit_cast(cv_h16x4, lb);
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (float)h[e] + (float)l[e];
  } else {
    r[0] = __uint_as_float(hb.x << 16) + __uint_as_float(lb.x << 16);
    r[1] = __uint_as_float(hb.x & 0xffff0000u) + __uint_as_float(lb.x & 0xffff0000u);
    r[2] = __uint_as_float(hb.y << 16) + __uint_as_float(lb.y << 16);
    r[3] = __uint_as_float(hb.y & 0xffff0000u) + __uint_as_float(lb.y & 0xffff0000u);
  }
  return r;
}

// "hl32" layout of a split activation: per pixel, per block of 32 channels, 32 hi values followed by their 32 lo values
// (bf16; hi = bf16(v), lo = bf16(v - hi)); channels padded with zeros to a multiple of 32:
//   element (pixel m, channel c, part) at  m * 2 Cp + (c >> 5) * 64 + part * 32 + (c & 31)
// One (pixel, 32-channel) chunk -- what a K step of the implicit GEMM consumes -- is ONE 128-byte line holding both
// parts.  With hi and lo in two separate NHWC tensors the same chunk was two half lines in two places, whose other
// halves (the next K chunk) were needed nine taps later, after the 4 MB L2 had lost them: every line came in from
// beyond L2 twice (FETCH_SIZE 2.2x the algorithmic bytes) and the LDS-DMA pieces touched twice as many lines.
__device__ __forceinline__ long hl_index(long m, int c, int Cp) { return m * 2 * Cp + (c >> 5) * 64 + (c & 31); }

__device__ __attribute__((aligned(256))) float ocv_zero_page[64];      // zero-initialised: source of padded taps


template <bool IN_SPLIT>   /* always false: pre-split inputs take conv_split_dma_kernel */
__global__ __launch_bounds__(512) void conv_igemm_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;

  // ---- XCD-aware, bijective workgroup -> tile map: consecutive tiles (N fastest) share an XCD
  const int nwg = p.mtiles * p.ntiles;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = wg / p.ntiles, nt = wg - mt * p.ntiles;
  const long m0 = (long)mt * CBM;
  const int n0 = nt * CBN;
  const int taps = p.ks * p.ks, pad = p.ks >> 1;
  const int nsteps = taps * (p.Cp / CBK);

  if (wave < 4) {
    // =========================== CONSUMERS ===========================
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};

    __syncthreads();                                   // buffer 0 written by the producers' prologue
    for (int step = 0; step < nsteps; ++step) {
      const unsigned char* base = lds + (step & 1) * BUF_BYTES;
      const unsigned char* pa = base + (wm * 128 + l31) * ROWB + hh * 16;
      const unsigned char* pb = base + 2 * A_BYTES + (wn * 64 + l31) * ROWB + hh * 16;
      // all 24 fragment reads of the step are issued up front: the second k-slice's reads return while the first
      // slice's 24 MFMAs run (the only matrix-pipe wave of this SIMD would otherwise sit through their latency)
      bf16x8 ah[2][4], al[2][4], bh[2][2], bl[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bh[kk][j] = *reinterpret_cast<const bf16x8*>(pb + j * 32 * ROWB + kk * 32);
          bl[kk][j] = *reinterpret_cast<const bf16x8*>(pb + B_BYTES + j * 32 * ROWB + kk * 32);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ah[kk][i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * ROWB + kk * 32);
          al[kk][i] = *reinterpret_cast<const bf16x8*>(pa + A_BYTES + i * 32 * ROWB + kk * 32);
        }
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk][i], bh[kk][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk][i], bl[kk][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kk][i], bh[kk][j], acc[i][j], 0, 0, 0);
          }
      __syncthreads();
    }

    // ---- epilogue: bias, activation, optional residual, NHWC store (128-byte runs per half-wave)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + l31;
      const bool nok = n < p.Cout;
      const float bv = (p.bias != nullptr && nok) ? p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long m = m0 + wm * 128 + i * 32 + acc_row(r, hh);
          if (nok && m < p.M) {
            float v = acc[i][j][r] + bv;
            if (p.act == OCV_ACT_LEAKY_RELU) v = v > 0.f ? v : 0.01f * v;
            else if (p.act == OCV_ACT_SILU) v = fast_silu(v);
            else if (p.act == OCV_ACT_RELU) v = fmaxf(v, 0.f);
            if (p.res != nullptr) v += p.res[m * p.Cout + n];
            if (p.y != nullptr) p.y[m * p.Cout + n] = v;
            if (p.yhl != nullptr) {                      // pre-split copy for the next convolution's A operand
              const __bf16 hb = (__bf16)v;
              p.yhl[hl_index(m, n, p.Cpo)] = hb;
              p.yhl[hl_index(m, n, p.Cpo) + 32] = (__bf16)(v - (float)hb);
            }
          }
        }
    }
    return;
  }

  // =========================== PRODUCERS ===========================
  // Two producer groups (waves 4-5 and 6-7) ALTERNATE: group g owns the K steps t = g, g+2, g+4, ...  A step's data
  // may only be written to LDS during the interval before it is consumed (its buffer is being read in the one before
  // that), so with a single group the global-load latency sat on the critical path of every step (1.8 us per step
  // against 0.8 us of MFMAs).  Each group now has TWO intervals per step it owns:
  //     interval t-1 (after writing step t):  issue the loads of step t+2          (>= one interval to land)
  //     interval t   ("convert" interval)   :  wait, split fp32 -> bf16 hi/lo into registers
  //     interval t+1 ("write" interval)     :  ds_write step t+2, then issue step t+4 ...
  // while the other group does the same shifted by one interval.  Every wave has at most one stage of loads in
  // flight and all loads are ordinary, compiler-visible loads (see gload16): exact s_waitcnt, no stale registers.
  //
  // All per-lane address work is hoisted out of the K loop (s_memtime stamps showed 2200 of 3600 producer cycles per
  // step in 64-bit index arithmetic, divisions and bounds tests): per row the byte offsets of the pixel in both source
  // tensors (32-bit; operands < 4 GiB, checked on the host) and a 9-bit mask of in-image taps; per step one scalar
  // byte offset for (tap shift, channel chunk); tap / chunk counters advance incrementally.
  const int g = (wave - 4) >> 1;                       // producer group
  const int gt = tid - 256 - 128 * g;                  // 0..127 inside the group
  // A role: 4 lanes per row (32 B = 8 channels each); rows (gt >> 2) + 32 i, i = 0..7
  const int apart = (gt & 3) * 8;
  unsigned rb1[8], rb2[8], tapmask[8];
  {
    const long hw = (long)p.H * p.W;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long am = m0 + (gt >> 2) + 32 * i;
      const bool valid = am < p.M;
      const long rem = valid ? am % hw : 0;
      const int y = (int)(rem / p.W), x = (int)(rem - (long)y * p.W);
      unsigned mask = 0;
      for (int t = 0; t < taps; ++t) {
        const int dy = t / p.ks - pad, dx = t % p.ks - pad;
        if (valid && (unsigned)(y + dy) < (unsigned)p.H && (unsigned)(x + dx) < (unsigned)p.W) mask |= 1u << t;
      }
      tapmask[i] = mask;
      const long pix = valid ? am : 0;
      rb1[i] = IN_SPLIT ? (unsigned)((pix * 2 * p.Cp + apart) * 2) : (unsigned)((pix * p.C1 + apart) * 4);
      rb2[i] = (unsigned)((pix * p.C2 + apart) * 4);
    }
  }
  // B role: one weight row per lane (32 channels: 64 B of hi, 64 B of lo)
  const int bn = min(n0 + gt, p.Cout - 1);
  const unsigned wrow = (unsigned)((long)bn * p.Cp * 2);                 // byte offset inside one tap's [Cout][Cp] slab
  const unsigned wtap = (unsigned)((long)p.Cout * p.Cp * 2);             // bytes per tap slab

  struct Raw { f32x4 a[16]; f32x4 bh[4], bl[4]; };                       // one step as loaded
  struct Cvt { bf16x8 ahi[8], alo[8]; };                                 // its A part, split

  int nx_tap = g % taps, nx_c0 = (g / taps) * CBK;                       // (tap, chunk) of this group's next step
  auto advance = [&]() {                                                 // += 2 steps
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (++nx_tap == taps) { nx_tap = 0; nx_c0 += CBK; }
  };
  auto issue_loads = [&](Raw& st) {
    const int tap = nx_tap, c0 = nx_c0;
    advance();
    const int ky = tap / p.ks, kx = tap - ky * p.ks;                     // scalar, ks in {1, 3}
    if (IN_SPLIT) {
      // pre-split input: 8 channels = 16 B of hi and 16 B of lo per lane and row, no conversion later
      const int soff = (((ky - pad) * p.W + (kx - pad)) * 2 * p.Cp + 2 * c0) * 2;      // hl32: chunk c0 starts at 2 c0
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool ok = ((tapmask[i] >> tap) & 1u);
        const unsigned off = rb1[i] + (unsigned)soff;
        st.a[2 * i + 0] = gload16(ok ? (const void*)((const char*)p.xhl + off) : (const void*)ocv_zero_page);
        st.a[2 * i + 1] = gload16(ok ? (const void*)((const char*)p.xhl + off + 64) : (const void*)ocv_zero_page);
      }
    } else {
    const bool first = c0 < p.C1;
    const char* tbase = (const char*)(first ? p.x1 : p.x2);
    const int tc = first ? p.C1 : p.C2;
    const int cin = first ? c0 : c0 - p.C1;                              // chunk start inside the source tensor
    const int soff = (((ky - pad) * p.W + (kx - pad)) * tc + cin) * 4;   // scalar byte offset of (tap, chunk)
    const bool cok0 = cin + apart + 4 <= tc, cok1 = cin + apart + 8 <= tc;   // channel tail of a partial chunk
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool inb = (tapmask[i] >> tap) & 1u;
      const unsigned off = (first ? rb1[i] : rb2[i]) + (unsigned)soff;
      const char* src = tbase + off;
      st.a[2 * i + 0] = gload16((inb && cok0) ? (const void*)src : (const void*)ocv_zero_page);
      st.a[2 * i + 1] = gload16((inb && cok1) ? (const void*)(src + 16) : (const void*)ocv_zero_page);
    }
    }
    const unsigned woff = (unsigned)tap * wtap + wrow + (unsigned)c0 * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      st.bh[e] = gload16((const char*)p.whi + woff + 16 * e);
      st.bl[e] = gload16((const char*)p.wlo + woff + 16 * e);
    }
  };
  auto convert = [&](const Raw& st, Cvt& cv) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (IN_SPLIT) {
        cv.ahi[i] = __builtin_bit_cast(bf16x8, st.a[2 * i + 0]);
        cv.alo[i] = __builtin_bit_cast(bf16x8, st.a[2 * i + 1]);
        continue;
      }
      __bf16 hi[8], lo[8];
      split4(st.a[2 * i + 0], hi, lo);
      split4(st.a[2 * i + 1], hi + 4, lo + 4);
      cv.ahi[i] = *reinterpret_cast<bf16x8*>(hi);
      cv.alo[i] = *reinterpret_cast<bf16x8*>(lo);
    }
  };
  auto write_lds = [&](int buf, const Raw& st, const Cvt& cv) {
    unsigned char* base = lds + buf * BUF_BYTES;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned char* ah = base + ((gt >> 2) + 32 * i) * ROWB + apart * 2;
      *reinterpret_cast<bf16x8*>(ah) = cv.ahi[i];
      *reinterpret_cast<bf16x8*>(ah + A_BYTES) = cv.alo[i];
    }
    unsigned char* bh = base + 2 * A_BYTES + gt * ROWB;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      *reinterpret_cast<f32x4*>(bh + 16 * e) = st.bh[e];
      *reinterpret_cast<f32x4*>(bh + B_BYTES + 16 * e) = st.bl[e];
    }
  };

  Raw raw;
  Cvt cvt;
  // prologue: group g fills buffer g with step g (nobody reads yet); group 0 also starts the loads of step 2
  if (g < nsteps) {
    issue_loads(raw);
    convert(raw, cvt);
    write_lds(g, raw, cvt);
  }
  if (g == 0 && 2 < nsteps) issue_loads(raw);
  __syncthreads();
  for (int t = 0; t < nsteps; ++t) {                   // interval t: the consumers multiply buffer t & 1
    if ((t & 1) == g) {
      // convert interval: the loads of step t+2 (issued one interval ago) land and are split; B stays as loaded
      if (t + 2 < nsteps) convert(raw, cvt);
      __syncthreads();
    } else {
      // write interval: step t+1 goes to buffer (t+1) & 1 (free since the last barrier), then fetch step t+3
      if (t + 1 >= 2 && t + 1 < nsteps) write_lds((t + 1) & 1, raw, cvt);
      if (t + 3 < nsteps) issue_loads(raw);
      __syncthreads();
    }
  }
}

}  // namespace


namespace {

// ---------------------------------------------------------------------------
// Pre-split input, LDS-DMA variant.  With the activation already stored as (hi, lo) bf16 the A and B tiles are pure
// copies, so the producers move them global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write: the
// ablation of the register-staged kernel put 25 % of its time in the ds_write_b128 path).  An LDS-DMA instruction
// writes wave-uniform base + lane x 16 B, i.e. rows cannot be padded; bank conflicts are avoided instead by an XOR
// swizzle of the 16-byte chunk index with (row >> 1) & 3, applied on the SOURCE side (each lane fetches the logical
// chunk that belongs at its linear LDS slot) and on the fragment reads.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int DROW = 64;                                        // bytes per LDS row (32 bf16), unpadded
constexpr int DA = CBM * DROW, DB = CBN * DROW;                 // 16384, 8192
constexpr int DBUF = 2 * DA + 2 * DB;                           // 49152 per buffer
constexpr int DNBUF = 3;                                         // 3 x 48 KiB: the DMA runs two K steps ahead

// Second half of the epilogue of conv_split_dma_kernel, run by ALL EIGHT wavefronts: the four consumers have parked
// their raw accumulators in LDS (one 128 x 64 fp32 image of ERS-float rows each); wavefront w turns rows
// [64 (w >> 2), +64) of consumer (w & 3)'s image into bias + activation (+ residual) and 16-byte stores, lane = (row
// it * 8 + (lane >> 3), channel octet lane & 7).  Stamps (tools/run_conv_split.py, -DOCV_STAMPS) had put the epilogue
// at 20 K cycles per tile when the four consumer wavefronts did all of it while the producers idled -- instruction
// issue on one wavefront per SIMD, not memory: 19 % of a 128 -> 128 tile at 240 x 320.
constexpr int ERS = 68;                                       // floats per LDS row: 64 + 4 -> conflict-free writes
__device__ __forceinline__ float conv_act(float v, int act) {
  if (act == OCV_ACT_LEAKY_RELU) return v > 0.f ? v : 0.01f * v;
  if (act == OCV_ACT_SILU) return fast_silu(v);
  if (act == OCV_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}
template <bool F16>
__device__ __forceinline__ void conv_store_rows(const ConvArgs& p, const unsigned char* lds, int wave, int lane, long m0, int n0,
                                                long yoff) {
  const int cw = wave & 3, half = wave >> 2;
  const int wm = cw >> 1, wn = cw & 1;
  const float* tile = reinterpret_cast<const float*>(lds) + cw * (128 * ERS);
  const int oct = lane & 7, rsub = lane >> 3;
  const int ncol = n0 + wn * 64 + oct * 8;
  if (ncol >= p.Cout) return;
  float bv[8], sv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bv[e] = p.bias != nullptr ? p.bias[ncol + e] : 0.f;
    sv[e] = p.oscale != nullptr ? p.oscale[ncol + e] : 1.f;
  }
  float amax = 0.f;                               // largest magnitude written as an fp16 pair (range guard, common.hpp)
#pragma unroll 4
  for (int it = 0; it < 8; ++it) {
    const int row = half * 64 + it * 8 + rsub;
    const long m = m0 + wm * 128 + row;
    if (m >= p.M) continue;
    f32x4 a = *reinterpret_cast<const f32x4*>(tile + row * ERS + oct * 8);
    f32x4 c = *reinterpret_cast<const f32x4*>(tile + row * ERS + oct * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = conv_act(fmaf(a[e], sv[e], bv[e]), p.act);            // (sv = 1: exact, the product of round 3)
      c[e] = conv_act(fmaf(c[e], sv[4 + e], bv[4 + e]), p.act);
    }
    const long o = m * p.Cout + ncol;
    if (p.res != nullptr) {
      a += *reinterpret_cast<const f32x4*>(p.res + o);
      c += *reinterpret_cast<const f32x4*>(p.res + o + 4);
    }
    if (p.y != nullptr) {
      *reinterpret_cast<f32x4*>(p.y + yoff + o) = a;
      *reinterpret_cast<f32x4*>(p.y + yoff + o + 4) = c;
    }
    if (p.yhl != nullptr) {
      __attribute__((aligned(16))) unsigned short hi[8], lo[8];
      split4_bits<F16>(a, hi, lo);
      split4_bits<F16>(c, hi + 4, lo + 4);
      if constexpr (F16) amax = ocv_amax4(ocv_amax4(amax, a), c);
      const long oh = hl_index(m, ncol, p.Cpo);          // ncol % 8 == 0: the octet stays inside one 32-block
      *reinterpret_cast<bf16x8*>(p.yhl + oh) = *reinterpret_cast<bf16x8*>(hi);
      *reinterpret_cast<bf16x8*>(p.yhl + oh + 32) = *reinterpret_cast<bf16x8*>(lo);
    }
  }
  if constexpr (F16) ocv_range_note(p.range_flag, amax);
}

// F16: the operands are fp16 (hi, lo) pairs instead of bf16 ones -- the same bytes through the same LDS-DMA pipeline, the
// MFMA of the other 2-byte type (the Winograd F(4x4, 3x3) GEMM batch below: 22-bit products).
template <bool F16>
__global__ __launch_bounds__(512) void conv_split_dma_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;

  const int ntile = p.mtiles * p.ntiles, nwg = ntile * (p.zflat > 0 ? p.zflat : p.ksplit * p.zbatch);
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int kz = wg / ntile;                                     // (GEMM of the batch, K part) of this workgroup
  wg -= kz * ntile;
  const int taps = p.ks * p.ks, pad = p.ks >> 1;
  const int nchunk = p.Cp / CBK;
  int zb, ch0, nsteps;
  if (p.zflat > 0) {                                             // steps [s0, s1) of the zbatch GEMMs' chunks laid end to end
    const int S = p.zbatch * nchunk;
    const int s0 = (int)((long)S * kz / p.zflat), s1 = (int)((long)S * (kz + 1) / p.zflat);
    zb = s0 / nchunk;
    ch0 = s0 - zb * nchunk;
    nsteps = s1 - s0;
  } else {
    zb = kz / p.ksplit;                                          // zb: 0 unless zbatch > 1
    const int kh = kz - zb * p.ksplit;                           // kh: 0 unless ksplit == 2
    ch0 = nchunk * kh / p.ksplit;                                // channel chunks [ch0, ch1)
    nsteps = p.gpt > 0 ? p.Kp / CBK : taps * (nchunk * (kh + 1) / p.ksplit - ch0);
  }
  const char* xz = (const char*)p.xhl + (long)zb * p.xz_bytes;   // (advanced by the producers in zflat mode)
  const char* whz = (const char*)p.whi + (long)zb * p.wz_bytes;
  const char* wlz = (const char*)p.wlo + (long)zb * p.wz_bytes;
  const int mt = wg / p.ntiles, nt = wg - mt * p.ntiles;
  const long m0 = (long)mt * CBM;
  const int n0 = nt * CBN;
  const long yoff = (long)kz * p.M * p.Cout;

  if (wave < 4) {
    // =========================== CONSUMERS, v_mfma_f32_16x16x32_bf16 ===========================
    // 2 x 2 over the tile, 128 x 64 outputs each (32 accumulators of 16 x 16); per step 24 ds_read_b128 and 96 MFMAs.
    // The 16 x 16 x 32 shape costs the same matrix-pipe cycles as 48 of 32 x 32 x 16 over the same LDS image and the
    // same reads, but the chip sustains a higher clock under it: A/B in one process on one device, 3-9 % faster per
    // launch (2224 -> 1024 at 30 x 40: 2.55 -> 2.33 ms).  Operand lanes: row / column l & 15, K octet l >> 4 (one
    // 16-byte chunk of the 64-byte row).  Swizzle key (row >> 1) & 3: any 8 consecutive rows of one K octet land in
    // 8 different 16-byte bank groups of a 128-byte LDS cycle.  (With the key (row >> 2) & 3 of the 32 x 32 consumers
    // this read pattern counted 1.4e8 SQ_LDS_BANK_CONFLICT cycles per launch and ran 1-3.5 % slower.)
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, q4 = lane >> 4;
    const int co = ((q4 ^ ((l15 >> 1) & 3)) * 16);
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
      const unsigned char* base = lds + (step % DNBUF) * DBUF;
      const unsigned char* pa = base + (wm * 128 + l15) * DROW + co;
      const unsigned char* pb = base + 2 * DA + (wn * 64 + l15) * DROW + co;
      bf16x8 ah[8], al[8], bh[4], bl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(pb + j * 16 * DROW);
        bl[j] = *reinterpret_cast<const bf16x8*>(pb + DB + j * 16 * DROW);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(pa + i * 16 * DROW);
        al[i] = *reinterpret_cast<const bf16x8*>(pa + DA + i * 16 * DROW);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (F16) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cv_h16x8, ah[i]), __builtin_bit_cast(cv_h16x8, bh[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cv_h16x8, ah[i]), __builtin_bit_cast(cv_h16x8, bl[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cv_h16x8, al[i]), __builtin_bit_cast(cv_h16x8, bh[j]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          }
        }
      __syncthreads();
    }

    // ---- epilogue.  accumulator (i, j): register r of lane l is output row 16 i + 4 (l >> 4) + r, column 16 j + (l & 15)
    // Storing straight from that layout costs 384 two- and four-byte store instructions per wavefront in 32- / 64-byte
    // runs; it was store-ISSUE-bound and, with one workgroup per CU, fully exposed: 23 % of the 128 -> 128 launches
    // (ablation: 1.63 -> 1.26 ms without the stores).  Instead the raw tile goes through this wavefront's own 34 KB of
    // the (now idle) LDS and leaves, from all eight wavefronts (conv_store_rows), as 16-byte-per-lane stores.
    if ((p.Cout & 7) == 0) {
      float* tile = reinterpret_cast<float*>(lds) + wave * (128 * ERS);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) tile[(16 * i + 4 * q4 + r) * ERS + 16 * j + l15] = acc[i][j][r];
      __syncthreads();
      conv_store_rows<F16>(p, lds, wave, lane, m0, n0, yoff);
      return;
    }
    // Cout not a multiple of 8: element-wise stores
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + l15;
      const bool nok = n < p.Cout;
      const float bv = (p.bias != nullptr && nok) ? p.bias[n] : 0.f;
      const float sv = (p.oscale != nullptr && nok) ? p.oscale[n] : 1.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long m = m0 + wm * 128 + i * 16 + 4 * q4 + r;
          if (nok && m < p.M) {
            float v = fmaf(acc[i][j][r], sv, bv);
            if (p.act == OCV_ACT_LEAKY_RELU) v = v > 0.f ? v : 0.01f * v;
            else if (p.act == OCV_ACT_SILU) v = fast_silu(v);
            else if (p.act == OCV_ACT_RELU) v = fmaxf(v, 0.f);
            if (p.res != nullptr) v += p.res[m * p.Cout + n];
            if (p.y != nullptr) p.y[yoff + m * p.Cout + n] = v;
            if (p.yhl != nullptr) {
              if constexpr (F16) {
                ocv_range_note(p.range_flag, fabsf(v));
                _Float16* yh = reinterpret_cast<_Float16*>(p.yhl);
                const _Float16 hb = (_Float16)v;
                yh[hl_index(m, n, p.Cpo)] = hb;
                yh[hl_index(m, n, p.Cpo) + 32] = (_Float16)(v - (float)hb);
              } else {
                const __bf16 hb = (__bf16)v;
                p.yhl[hl_index(m, n, p.Cpo)] = hb;
                p.yhl[hl_index(m, n, p.Cpo) + 32] = (__bf16)(v - (float)hb);
              }
            }
          }
        }
    }
    return;
  }

  // =========================== PRODUCERS (LDS-DMA issuers) ===========================
  // wave pw moves A rows [64 pw, 64 pw + 64) (hi and lo: 8 x 1 KiB pieces) and B rows [32 pw, 32 pw + 32) (4 pieces).
  // Lane L of a piece lands at piece base + 16 L  =  row L >> 2, stored chunk L & 3, which must hold LOGICAL chunk
  // (L & 3) ^ ((row >> 1) & 3) = (L & 3) ^ ((L >> 3) & 3)  (piece bases are multiples of 16 rows).
  const int pw = wave - 4;
  const int lrow = lane >> 2;
  const int lchunk = (lane & 3) ^ ((lane >> 3) & 3);               // logical 16-byte chunk (8 channels) this lane fetches
  // (32-bit arithmetic and no division by the runtime kernel size: the 64-bit % and / of a first version, eight per
  // lane plus two per tap, were most of a 14 K-cycle prologue in front of every tile's first load)
  unsigned rbA[4], tapmask[4];
  {
    const unsigned hw = (unsigned)(p.H * p.W);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned am = (unsigned)m0 + 64 * pw + 16 * i + lrow;           // M < 2^30 (4 GiB operand limit)
      const bool valid = am < (unsigned)p.M;
      const unsigned img = valid ? am / hw : 0u;
      const unsigned rem = valid ? am - img * hw : 0u;
      const int y = (int)(rem / (unsigned)p.W), x = (int)(rem - (unsigned)y * (unsigned)p.W);
      unsigned mask = 0;
      int t = 0;
      for (int dy = -pad; dy <= pad; ++dy)
        for (int dx = -pad; dx <= pad; ++dx, ++t)
          if (valid && (unsigned)(y + dy) < (unsigned)p.H && (unsigned)(x + dx) < (unsigned)p.W) mask |= 1u << t;
      tapmask[i] = mask;
      rbA[i] = (p.rowpitch == 0 ? (valid ? am : 0u) * (unsigned)(4 * p.Cp)
                                : (img * (unsigned)p.H + (unsigned)y) * p.rowpitch + (unsigned)x * (unsigned)(4 * p.Cp)) +
               (unsigned)(lchunk * 16);
    }
  }
  unsigned rbB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int bn = min(n0 + 32 * pw + 16 * i + lrow, p.Cout - 1);
    rbB[i] = (unsigned)(((long)bn * (p.gpt > 0 ? p.Kp : p.Cp) + lchunk * 8) * 2);
  }
  const unsigned wtap = (unsigned)((long)p.Cout * p.Cp * 2);

  // PACKED TAPS: this lane's granule of step s is G = 4 s + lchunk = (tap, granule cg of the tap's pixel); tap and cg differ between
  // the four lanes of a row, so the tap shift and the in-image test are per lane (a dozen VALU instructions per step on wavefronts
  // that otherwise only issue DMA).  Granules beyond the ninth tap (the K padding of the last step) read the zero page; the
  // weights hold zeros there.
  int nx_pk = 0;
  auto issue_dma_packed = [&](int buf) {
    const int G = 4 * nx_pk + lchunk;
    const int tap_raw = (int)(((unsigned)G * (unsigned)p.gpt_inv) >> 16);
    const int cg = G - tap_raw * p.gpt;
    const bool in_k = tap_raw < 9;
    const int tap = in_k ? tap_raw : 8;
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;                  // tap / 3 for tap <= 8
    // rbA carries lchunk * 16 (the dense layout's granule of this lane): replaced by the packed granule's place in its pixel
    const int soff = ((ky - 1) * p.W + (kx - 1)) * 4 * p.Cp + (cg >> 2) * 128 + (cg & 3) * 16 - lchunk * 16;
    unsigned char* base = lds + buf * DBUF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = in_k && ((tapmask[i] >> tap) & 1u);
      const unsigned off = rbA[i] + (unsigned)soff;
      const void* sh = ok ? (const void*)(xz + off) : (const void*)ocv_zero_page;
      const void* sl = ok ? (const void*)(xz + off + 64) : (const void*)ocv_zero_page;
      unsigned char* dst = base + (64 * pw + 16 * i) * DROW;
      __builtin_amdgcn_global_load_lds((gptr_t)sh, (lptr_t)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)sl, (lptr_t)(dst + DA), 16, 0, 0);
    }
    const unsigned woff = (unsigned)nx_pk * (CBK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned char* dst = base + 2 * DA + (32 * pw + 16 * i) * DROW;
      __builtin_amdgcn_global_load_lds((gptr_t)(whz + woff + rbB[i]), (lptr_t)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(wlz + woff + rbB[i]), (lptr_t)(dst + DB), 16, 0, 0);
    }
    ++nx_pk;
  };

  int nx_tap = 0, nx_c0 = ch0 * CBK, nx_ky = 0, nx_kx = 0;
  auto issue_dma_dense = [&](int buf) {
    const int tap = nx_tap, c0 = nx_c0, ky = nx_ky, kx = nx_kx;
    const char* const xz_ = xz;
    const char* const whz_ = whz;
    const char* const wlz_ = wlz;
    if (++nx_kx == p.ks) { nx_kx = 0; ++nx_ky; }
    if (++nx_tap == taps) { nx_tap = 0; nx_ky = 0; nx_c0 += CBK; }
    if (p.zflat > 0 && nx_c0 == p.Cp) {                          // next step: first chunk of the next GEMM of the batch
      nx_c0 = 0;
      xz += p.xz_bytes; whz += p.wz_bytes; wlz += p.wz_bytes;
    }
    const int soff = (((ky - pad) * p.W + (kx - pad)) * 2 * p.Cp + 2 * c0) * 2;       // hl32: chunk c0 starts at 2 c0
    unsigned char* base = lds + buf * DBUF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = ((tapmask[i] >> tap) & 1u);               // pad channels are zeros in memory: no channel check
      const unsigned off = rbA[i] + (unsigned)soff;
      const void* sh = ok ? (const void*)(xz_ + off) : (const void*)ocv_zero_page;
      const void* sl = ok ? (const void*)(xz_ + off + 64) : (const void*)ocv_zero_page;   // same 128-B line
      unsigned char* dst = base + (64 * pw + 16 * i) * DROW;
      __builtin_amdgcn_global_load_lds((gptr_t)sh, (lptr_t)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)sl, (lptr_t)(dst + DA), 16, 0, 0);
    }
    const unsigned woff = (unsigned)tap * wtap + (unsigned)c0 * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned char* dst = base + 2 * DA + (32 * pw + 16 * i) * DROW;
      __builtin_amdgcn_global_load_lds((gptr_t)(whz_ + woff + rbB[i]), (lptr_t)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(wlz_ + woff + rbB[i]), (lptr_t)(dst + DB), 16, 0, 0);
    }
  };

  auto issue_dma = [&](int buf) {
    if (p.gpt > 0) issue_dma_packed(buf);                         // (uniform: a scalar branch)
    else issue_dma_dense(buf);
  };

  // Three LDS buffers, DMA two K steps ahead: in interval t (consumers on buffer t % 3) the producers issue step t+2
  // into buffer (t+2) % 3 -- last read in interval t-1, free since the previous barrier -- and then only wait for step
  // t+1, issued a whole interval earlier: a COUNTED s_waitcnt vmcnt(12) (this wave's 12 newest pieces may stay in
  // flight) followed by a raw s_barrier; __syncthreads() would emit vmcnt(0) and drain the young pieces too.  An
  // LDS-DMA has no register destination, so hand-counting it carries none of the register-copy hazard described at
  // gload16().
#define OCV_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
  issue_dma(0);
  if (nsteps > 1) {
    issue_dma(1);
    OCV_WAIT_VM(12);
  } else {
    OCV_WAIT_VM(0);
  }
  __builtin_amdgcn_s_barrier();
  for (int t = 0; t < nsteps; ++t) {
    if (t + 2 < nsteps) {
      issue_dma((t + 2) % DNBUF);
      OCV_WAIT_VM(12);                                 // step t+1 has landed, step t+2 may still be in flight
    } else {
      OCV_WAIT_VM(0);
    }
    __builtin_amdgcn_s_barrier();
  }
#undef OCV_WAIT_VM
  if ((p.Cout & 7) == 0) {                             // the consumers park their accumulators in LDS; all eight waves store
    __syncthreads();
    conv_store_rows<F16>(p, lds, wave, lane, m0, n0, yoff);
  }
}


// Second pass of a split-K convolution: y = act(part0 + part1 + bias) (+ residual), fp32 and / or hl32 split output.
// One thread per (pixel, channel octet); fixed summation order.
struct FinArgs {
  const float *part, *bias, *res;
  float* y;
  __bf16* yhl;
  long M, items;         // items = M * Cout / 8
  int Cout, Cpo, act, ksplit;
  const float* oscale;   // nullable [Cout] (ConvArgs::oscale)
  int f16;               // element type of yhl
  unsigned* range_flag;  // nullable (ConvArgs::range_flag)
};

__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(FinArgs p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.items) return;
  const int noct = p.Cout >> 3;
  const long m = i / noct;
  const int n = (int)(i - m * noct) * 8;
  const long o = m * p.Cout + n;
  f32x4 a = *reinterpret_cast<const f32x4*>(p.part + o), c = *reinterpret_cast<const f32x4*>(p.part + o + 4);
  for (int k = 1; k < p.ksplit; ++k) {
    a += *reinterpret_cast<const f32x4*>(p.part + k * p.M * p.Cout + o);
    c += *reinterpret_cast<const f32x4*>(p.part + k * p.M * p.Cout + o + 4);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = conv_act(fmaf(a[e], p.oscale != nullptr ? p.oscale[n + e] : 1.f, p.bias != nullptr ? p.bias[n + e] : 0.f), p.act);
    c[e] = conv_act(fmaf(c[e], p.oscale != nullptr ? p.oscale[n + 4 + e] : 1.f, p.bias != nullptr ? p.bias[n + 4 + e] : 0.f), p.act);
  }
  if (p.res != nullptr) {
    a += *reinterpret_cast<const f32x4*>(p.res + o);
    c += *reinterpret_cast<const f32x4*>(p.res + o + 4);
  }
  if (p.y != nullptr) {
    *reinterpret_cast<f32x4*>(p.y + o) = a;
    *reinterpret_cast<f32x4*>(p.y + o + 4) = c;
  }
  if (p.yhl != nullptr) {
    __attribute__((aligned(16))) unsigned short hi[8], lo[8];
    split4_bits(a, hi, lo, p.f16 != 0);
    split4_bits(c, hi + 4, lo + 4, p.f16 != 0);
    if (p.f16 != 0) ocv_range_note(p.range_flag, ocv_amax4(ocv_amax4(0.f, a), c));
    const long oh = hl_index(m, n, p.Cpo);
    *reinterpret_cast<bf16x8*>(p.yhl + oh) = *reinterpret_cast<bf16x8*>(hi);
    *reinterpret_cast<bf16x8*>(p.yhl + oh + 32) = *reinterpret_cast<bf16x8*>(lo);
  }
}

}  // namespace

namespace {
// Split-K pays when the tile count leaves the last round of workgroups mostly empty: 600 tiles on 256 CUs run three
// rounds for 2.34 rounds of work; as 1200 half-K workgroups they run five half-rounds = 2.5.  Returns 1 or 2.
int conv_ksplit(long M, int Cout, int Cin, int ksize) {
  const long tiles = (long)ocv_cdiv(M, CBM) * ocv_cdiv(Cout, CBN);
  const int nsteps = ksize * ksize * ((Cin + CBK - 1) / CBK);
  if ((Cout & 7) != 0 || nsteps < 128 || tiles <= 256) return 1;
  const double c1 = (double)((tiles + 255) / 256), c2 = (double)((2 * tiles + 255) / 256) / 2.0;
  return c2 <= 0.9 * c1 ? 2 : 1;
}
int launch_conv(ConvArgs& a, int B, bool in_split, hipStream_t st) {
  a.Cp = (a.Cin + CBK - 1) / CBK * CBK;
  a.M = (long)B * a.H * a.W;
  a.mtiles = ocv_cdiv(a.M, CBM); a.ntiles = ocv_cdiv(a.Cout, CBN);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (in_split) {
    static bool attr2 = false;
    if (!attr2) {
      (void)hipFuncSetAttribute((const void*)conv_split_dma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_split_dma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr2 = true;
    }
    if (a.ksplit < 1) a.ksplit = 1;
    if (a.zbatch < 1) a.zbatch = 1;
    const unsigned parts = a.zflat > 0 ? a.zflat : a.ksplit * a.zbatch;
    if (a.f16)
      hipLaunchKernelGGL(conv_split_dma_kernel<true>, dim3(a.mtiles * a.ntiles * parts), dim3(512), DNBUF * DBUF, st, a);
    else
      hipLaunchKernelGGL(conv_split_dma_kernel<false>, dim3(a.mtiles * a.ntiles * parts), dim3(512), DNBUF * DBUF, st, a);
  } else hipLaunchKernelGGL(conv_igemm_kernel<false>, dim3(a.mtiles * a.ntiles), dim3(512), 2 * BUF_BYTES, st, a);
  OCV_CHECK_LAUNCH("ocv_conv_nhwc");
  return 0;
}
}  // namespace

extern "C" size_t ocv_split_act_elems(int B, int H, int W, int C) {
  if (B < 1 || H < 1 || W < 1 || C < 1) return 0;
  return (size_t)B * H * W * 2 * ((C + 31) / 32 * 32);
}

extern "C" size_t ocv_conv_nhwc_split_workspace_bytes(int B, int H, int W, int Cin, int Cout, int ksize) {
  if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || (ksize != 1 && ksize != 3)) return 0;
  const long M = (long)B * H * W;
  const int ks = conv_ksplit(M, Cout, Cin, ksize);
  return ks > 1 ? (size_t)ks * M * Cout * sizeof(float) : 0;
}

extern "C" int ocv_conv_nhwc_split_x_fwd(const void* x_hl, int Cin, const void* w_hi, const void* w_lo, const float* oscale,
                                         int f16, const float* bias, const float* residual, float* y, void* y_hl, int B, int H,
                                         int W, int Cout, int ksize, int act, void* workspace, size_t workspace_bytes,
                                         ocv_stream_t stream) {
  OCV_CHECK_ARG(x_hl && w_hi && w_lo && (y || y_hl), "ocv_conv_nhwc_split_fwd: null pointer");
  OCV_CHECK_ARG(ocv_aligned16(y) && ocv_aligned16(y_hl) && ocv_aligned16(residual) && ocv_aligned16(workspace), "ocv_conv_nhwc_split_fwd: outputs, residual and workspace must be 16-byte aligned");
  OCV_CHECK_ARG(ksize == 1 || ksize == 3, "ocv_conv_nhwc_split_fwd: kernel size must be 1 or 3 (got %d)", ksize);
  OCV_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && Cout >= 1 && Cin >= 1, "ocv_conv_nhwc_split_fwd: bad sizes");
  OCV_CHECK_ARG(act >= 0 && act <= 3, "ocv_conv_nhwc_split_fwd: unknown activation %d", act);
  OCV_CHECK_ARG((reinterpret_cast<uintptr_t>(x_hl) & 127) == 0 && ocv_aligned16(w_hi) && ocv_aligned16(w_lo), "ocv_conv_nhwc_split_fwd: x_hl must be 128-byte aligned, weights 16-byte aligned");
  OCV_CHECK_ARG((long)B * H * W * (Cin + 32) * 4 < (1L << 32) && 9L * Cout * (Cin + 32) * 2 < (1L << 32), "ocv_conv_nhwc_split_fwd: each operand must be smaller than 4 GiB");
  OCV_CHECK_ARG(f16 == 0 || f16 == 1, "ocv_conv_nhwc_split_fwd: f16 must be 0 (bf16 pairs) or 1 (fp16 pairs)");
  ConvArgs a{};
  a.xhl = (const __bf16*)x_hl; a.yhl = (__bf16*)y_hl; a.Cpo = (Cout + 31) / 32 * 32;
  a.whi = (const __bf16*)w_hi; a.wlo = (const __bf16*)w_lo; a.bias = bias; a.res = residual; a.y = y;
  a.C1 = Cin; a.C2 = 0; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.ks = ksize; a.act = act; a.ksplit = 1;
  a.f16 = f16; a.oscale = oscale;
  a.range_flag = (f16 && y_hl != nullptr) ? ocv_range_flag_current() : nullptr;
  if (y_hl != nullptr && Cout % 32 != 0) {       // the kernels write channels < Cout only: pad channels must read as zero
    const int zrc = ocv_zero_async(y_hl, ocv_split_act_elems(B, H, W, Cout) * sizeof(__bf16), (hipStream_t)stream);          // (a launch, not a memset node: common.hpp)
    if (zrc != 0) return zrc;
  }
  const long M = (long)B * H * W;
  const int ks = conv_ksplit(M, Cout, Cin, ksize);
  if (ks > 1 && workspace != nullptr && workspace_bytes >= (size_t)ks * M * Cout * sizeof(float)) {
    // two workgroups per tile, each over half of the channel chunks -> raw partial sums -> finish pass
    ConvArgs h = a;
    h.bias = nullptr; h.res = nullptr; h.yhl = nullptr; h.act = OCV_ACT_NONE; h.y = (float*)workspace; h.ksplit = ks; h.oscale = nullptr;
    const int rc = launch_conv(h, B, true, (hipStream_t)stream);
    if (rc != 0) return rc;
    FinArgs f{(const float*)workspace, bias, residual, y, (__bf16*)y_hl, M, M * Cout / 8, Cout, a.Cpo, act, ks, oscale, f16, a.range_flag};
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)((f.items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f);
    OCV_CHECK_LAUNCH("ocv_conv_nhwc_split_fwd(finish)");
    return 0;
  }
  return launch_conv(a, B, true, (hipStream_t)stream);
}

// PACKED TAPS (round 6; ConvArgs::gpt): the same 3 x 3 convolution with the K axis = the nine taps' REAL 8-channel granules laid end
// to end.  Pays where Cin is far from a multiple of 32 -- the decoder's skip parts over 24 and 40 encoder channels (7 K steps instead
// of 9, 12 instead of 18): the matrix cores no longer multiply the zero pad channels of every tap.  w_hi / w_lo: ONE [Cout][Kp]
// matrix each, Kp = ocv_conv3x3_packed_taps_k(Cin), element (co, 8 (gpt t + g) + e) = weight[co][8 g + e][tap t] (zero where
// 8 g + e >= Cin, and beyond the ninth tap), split into pairs like the tap-major weights (hip_ops.prep_conv_weight_packed_taps).
extern "C" int ocv_conv3x3_packed_taps_k(int Cin) {
  if (Cin < 1 || Cin > 2040) return 0;
  const int gpt = (Cin + 7) / 8;
  return (9 * gpt + 3) / 4 * CBK;
}

extern "C" int ocv_conv3x3_split_packed_taps_fwd(const void* x_hl, int Cin, const void* w_hi, const void* w_lo, const float* oscale,
                                                 int f16, const float* bias, const float* residual, float* y, void* y_hl, int B,
                                                 int H, int W, int Cout, int act, ocv_stream_t stream) {
  OCV_CHECK_ARG(x_hl && w_hi && w_lo && (y || y_hl), "ocv_conv3x3_split_packed_taps_fwd: null pointer");
  OCV_CHECK_ARG(ocv_aligned16(y) && ocv_aligned16(y_hl) && ocv_aligned16(residual), "ocv_conv3x3_split_packed_taps_fwd: outputs and residual must be 16-byte aligned");
  OCV_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && Cout >= 1 && Cin >= 8 && Cin % 8 == 0 && Cin <= 2040,
                "ocv_conv3x3_split_packed_taps_fwd: bad sizes (Cin must be a multiple of 8 in [8, 2040], got %d)", Cin);
  OCV_CHECK_ARG(act >= 0 && act <= 3, "ocv_conv3x3_split_packed_taps_fwd: unknown activation %d", act);
  OCV_CHECK_ARG(f16 == 0 || f16 == 1, "ocv_conv3x3_split_packed_taps_fwd: f16 must be 0 (bf16 pairs) or 1 (fp16 pairs)");
  OCV_CHECK_ARG((reinterpret_cast<uintptr_t>(x_hl) & 127) == 0 && ocv_aligned16(w_hi) && ocv_aligned16(w_lo),
                "ocv_conv3x3_split_packed_taps_fwd: x_hl must be 128-byte aligned, weights 16-byte aligned");
  const int Kp = ocv_conv3x3_packed_taps_k(Cin), gpt = Cin / 8;
  OCV_CHECK_ARG((long)B * H * W * (Cin + 32) * 4 < (1L << 32) && (long)Cout * Kp * 2 < (1L << 32),
                "ocv_conv3x3_split_packed_taps_fwd: each operand must be smaller than 4 GiB");
  const int inv = (65536 + gpt - 1) / gpt;
  for (int G = 0; G < Kp / 8; ++G)
    OCV_CHECK_ARG((int)(((unsigned)G * (unsigned)inv) >> 16) == G / gpt, "ocv_conv3x3_split_packed_taps_fwd: internal (reciprocal of %d inexact at %d)", gpt, G);
  ConvArgs a{};
  a.xhl = (const __bf16*)x_hl; a.yhl = (__bf16*)y_hl; a.Cpo = (Cout + 31) / 32 * 32;
  a.whi = (const __bf16*)w_hi; a.wlo = (const __bf16*)w_lo; a.bias = bias; a.res = residual; a.y = y;
  a.C1 = Cin; a.C2 = 0; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.ks = 3; a.act = act; a.ksplit = 1;
  a.f16 = f16; a.oscale = oscale;
  a.gpt = gpt; a.gpt_inv = inv; a.Kp = Kp;
  a.range_flag = (f16 && y_hl != nullptr) ? ocv_range_flag_current() : nullptr;
  if (y_hl != nullptr && Cout % 32 != 0) {       // the kernels write channels < Cout only: pad channels must read as zero
    const int zrc = ocv_zero_async(y_hl, ocv_split_act_elems(B, H, W, Cout) * sizeof(__bf16), (hipStream_t)stream);
    if (zrc != 0) return zrc;
  }
  return launch_conv(a, B, true, (hipStream_t)stream);
}

extern "C" int ocv_conv_nhwc_split_ws_fwd(const void* x_hl, int Cin, const void* w_hi, const void* w_lo, const float* bias,
                                          const float* residual, float* y, void* y_hl, int B, int H, int W, int Cout,
                                          int ksize, int act, void* workspace, size_t workspace_bytes, ocv_stream_t stream) {
  return ocv_conv_nhwc_split_x_fwd(x_hl, Cin, w_hi, w_lo, nullptr, 0, bias, residual, y, y_hl, B, H, W, Cout, ksize, act, workspace,
                                   workspace_bytes, stream);
}

extern "C" int ocv_conv_nhwc_split_fwd(const void* x_hl, int Cin, const void* w_hi, const void* w_lo, const float* bias,
                                       const float* residual, float* y, void* y_hl, int B, int H, int W, int Cout,
                                       int ksize, int act, ocv_stream_t stream) {
  return ocv_conv_nhwc_split_ws_fwd(x_hl, Cin, w_hi, w_lo, bias, residual, y, y_hl, B, H, W, Cout, ksize, act, nullptr, 0, stream);
}

extern "C" int ocv_conv_nhwc_fwd(const float* x1, int C1, const float* x2, int C2, const void* w_hi, const void* w_lo,
                                 const float* bias, const float* residual, float* y, int B, int H, int W, int Cout,
                                 int ksize, int act, ocv_stream_t stream) {
  OCV_CHECK_ARG(x1 && w_hi && w_lo && y, "ocv_conv_nhwc_fwd: null pointer");
  OCV_CHECK_ARG(ksize == 1 || ksize == 3, "ocv_conv_nhwc_fwd: kernel size must be 1 or 3 (got %d)", ksize);
  OCV_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && Cout >= 1 && C1 >= 4, "ocv_conv_nhwc_fwd: bad sizes");
  OCV_CHECK_ARG(C1 % 4 == 0 && (x2 == nullptr || (C2 >= 4 && C2 % 4 == 0 && C1 % CBK == 0)),
                "ocv_conv_nhwc_fwd: channel counts must be multiples of 4 (and C1 a multiple of %d when a second tensor is concatenated)", CBK);
  OCV_CHECK_ARG(act >= 0 && act <= 3, "ocv_conv_nhwc_fwd: unknown activation %d", act);
  OCV_CHECK_ARG(ocv_aligned16(x1) && ocv_aligned16(x2) && ocv_aligned16(w_hi) && ocv_aligned16(w_lo),
                "ocv_conv_nhwc_fwd: operands must be 16-byte aligned");
  OCV_CHECK_ARG((long)B * H * W * C1 * 4 < (1L << 32) && (long)B * H * W * (x2 ? C2 : 0) * 4 < (1L << 32) &&
                    9L * Cout * (C1 + C2 + 32) * 2 < (1L << 32),
                "ocv_conv_nhwc_fwd: each operand must be smaller than 4 GiB (32-bit byte offsets inside the kernel)");
  ConvArgs a{};
  a.x1 = x1; a.x2 = x2; a.whi = (const __bf16*)w_hi; a.wlo = (const __bf16*)w_lo;
  a.bias = bias; a.res = residual; a.y = y;
  a.C1 = C1; a.C2 = x2 ? C2 : 0; a.Cin = a.C1 + a.C2;
  a.Cout = Cout; a.H = H; a.W = W; a.ks = ksize; a.act = act;
  return launch_conv(a, B, false, (hipStream_t)stream);
}

namespace {
inline size_t wino_align(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

// (Round 2's Winograd F(2x2, 3x3) on bf16 pairs -- 2.25x fewer matrix operations, the 30 x 40 stage only -- was superseded by the
// F(4x4, 3x3) form below in round 3 and left the product in round 5: tools/diag/winograd_f22.patch.txt.)

// =====================================================================================================================
// Winograd F(4x4, 3x3) on the two-term FP16 split (round 3).  36 multiplies per 16 outputs instead of 144: 4x fewer matrix-core
// operations than the direct form (F(2x2, 3x3): 2.25x), and a transformed input of 2.25x the activation instead of 4x.  Its
// transforms amplify rounding ~100x, which is why round 2 stopped at F(2x2): with two bf16 terms (2^-17 products) the result
// is 1e-4 of max |y| from fp64, five times the kernels' bar.  Two FP16 terms carry 22 bits: 2 - 3.5e-6 (CPU experiment on
// 256 .. 1024 input channels, the figure of a three-term bf16 split) -- with ONE accumulator, i.e. on conv_split_dma_kernel
// as it is (the MFMA of the other 2-byte type), provided no term falls into fp16's subnormals: the transformed filter
// U = G g G^T has entries down to 1/576 of the filter's, so every position's U is scaled by a power of two that puts its
// largest entry near 2^8 (undone on the raw GEMM result by the output transform).  The transformed INPUT is up to 49x the
// activation (7x per 1-D pass with the interpolation points 0, 1, -1, 2, -1/2, inf) and its low term is unscaled, so as it stands
// it would overflow from activations of ~1.3e3 and lose its low terms to fp16's subnormals below ~0.1: every TILE's
// transformed input is therefore scaled by a power of two taken from the tile's own largest input (49 amax 2^s in [2^14, 2^15):
// a row of the GEMM, so it factors out and is undone exactly by the output transform), and every input CHANNEL by a static power
// of two that equalises the filters' columns (a channel with tiny weights and huge activations -- or the reverse -- has BOTH
// factors moved to the middle of fp16's range; hip_ops.prep_winograd43_weight).  All scalings are exact.
//   1. wino43_input_kernel   V[xi][tile][c] = (B^T d B)[xi] in fp32 from the re-joined bf16 split input, written as fp16 (hi, lo)
//                            rows, xi = 6 i + j, T = B ceil(H/4) ceil(W/4) tiles (6 x 6 input patch, stride 4, zero padded)
//   2. conv_split_dma_kernel<true>, a batch of 36 GEMMs M[xi] = V[xi] . U'[xi]^T, raw fp32
//   3. wino43_output_kernel  Y = A^T (M (.) 2^-k) A + bias, activation, fp32 and / or bf16 hl32 split stores
// =====================================================================================================================
namespace {

struct Wino43InArgs {
  const __bf16* xhl;      // [B][H][W][2 Cp] hl32 (bf16 hi | lo; fp16 pairs with in_f16)
  _Float16* v;            // [36][T][2 Cp] hl32 rows (fp16 hi | lo)
  const float* cscale;    // [Cp] per-input-channel power of two 2^a_c (1 for pad channels): the filters carry 2^-a_c
  float* tinv;            // [T] out: 2^-s_t, the inverse of the power of two this tile's transformed input was scaled by
  int B, H, W, Cp, th, tw;
  long T;
  int in_f16;             // element type of xhl
};

// y = B^T x for one column of six values.  Interpolation points 0, 1, -1, 2, -1/2, inf: against the textbook set (0, +-1, +-2) the
// transformed input is at most 49x the activation (7x per 1-D pass: the largest absolute row sum of B^T) instead of 27^2 x, and the
// result 2.6x closer to fp64 (CPU experiment with fp32-accumulating GEMMs, 1024 channels: 1.35e-6 against 3.46e-6 of max |y|);
// every coefficient is a dyadic rational, i.e. exact.
//   B^T = [1 3/2 -2 -3/2 1 0;  0 -1 -5/2 -1/2 1 0;  0 1 1/2 -5/2 1 0;  0 -1/2 -1 1/2 1 0;  0 2 -1 -2 1 0;  0 1 3/2 -2 -3/2 1]
__device__ __forceinline__ void w43_bt(const float (&x)[6], float (&y)[6]) {
  y[0] = x[0] + 1.5f * (x[1] - x[3]) - 2.f * x[2] + x[4];
  y[1] = -x[1] - 2.5f * x[2] - 0.5f * x[3] + x[4];
  y[2] = x[1] + 0.5f * x[2] - 2.5f * x[3] + x[4];
  y[3] = 0.5f * (x[3] - x[1]) - x[2] + x[4];
  y[4] = 2.f * (x[1] - x[3]) - x[2] + x[4];
  y[5] = x[1] + 1.5f * (x[2] - x[4]) - 2.f * x[3] + x[5];
}

// The 6 x 6 input patch of tile t, channel quad c (re-joined hi + lo, times the channels' static power of two), after the COLUMN
// pass of the transform: w[ii][xx][e] = (B^T d)[ii][xx].  Returns the largest |d| of the quad's 36 x 4 inputs.
__device__ __forceinline__ float w43_load_patch(const Wino43InArgs& p, long t, int c, float (&w)[6][6][4]) {
  const int tx = (int)(t % p.tw);
  const long r = t / p.tw;
  const int ty = (int)(r % p.th), b = (int)(r / p.th);
  const long coff = (long)(c >> 5) * 64 + (c & 31);                       // hi quad; lo quad at + 32
  const f32x4 cs = p.cscale != nullptr ? *reinterpret_cast<const f32x4*>(p.cscale + c) : f32x4{1.f, 1.f, 1.f, 1.f};
  float amax = 0.f;
#pragma unroll
  for (int xx = 0; xx < 6; ++xx) {
    const int x = 4 * tx - 1 + xx;
    float col[4][6];
#pragma unroll
    for (int yy = 0; yy < 6; ++yy) {
      const int y = 4 * ty - 1 + yy;
      const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const __bf16* src = ok ? p.xhl + (((long)b * p.H + y) * p.W + x) * 2 * p.Cp + coff
                             : reinterpret_cast<const __bf16*>(ocv_zero_page);
      const f32x4 d4 = join4_bits(src, ok ? 32 : 4, p.in_f16 != 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        col[e][yy] = d4[e] * cs[e];
        amax = fmaxf(amax, fabsf(col[e][yy]));                             // (a NaN input is dropped here and still reaches V below)
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float o[6];
      w43_bt(col[e], o);
#pragma unroll
      for (int ii = 0; ii < 6; ++ii) w[ii][xx][e] = o[ii];
    }
  }
  return amax;
}

// the largest |d| alone (first round of the many-channel path)
__device__ __forceinline__ float w43_patch_amax(const Wino43InArgs& p, long t, int c) {
  const int tx = (int)(t % p.tw);
  const long r = t / p.tw;
  const int ty = (int)(r % p.th), b = (int)(r / p.th);
  const long coff = (long)(c >> 5) * 64 + (c & 31);
  const f32x4 cs = p.cscale != nullptr ? *reinterpret_cast<const f32x4*>(p.cscale + c) : f32x4{1.f, 1.f, 1.f, 1.f};
  float amax = 0.f;
  for (int yy = 0; yy < 6; ++yy) {
    const int y = 4 * ty - 1 + yy;
    if ((unsigned)y >= (unsigned)p.H) continue;
#pragma unroll
    for (int xx = 0; xx < 6; ++xx) {
      const int x = 4 * tx - 1 + xx;
      if ((unsigned)x >= (unsigned)p.W) continue;
      const __bf16* src = p.xhl + (((long)b * p.H + y) * p.W + x) * 2 * p.Cp + coff;
      const f32x4 d4 = join4_bits(src, 32, p.in_f16 != 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(d4[e] * cs[e]));
    }
  }
  return amax;
}

// row pass of the transform, scaled by the tile's power of two, split into fp16 (hi, lo) and stored
__device__ __forceinline__ void w43_store_patch(const Wino43InArgs& p, long t, int c, const float (&w)[6][6][4], float sc) {
  const long coff = (long)(c >> 5) * 64 + (c & 31);
#pragma unroll
  for (int ii = 0; ii < 6; ++ii) {
    float v6[4][6];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float rowv[6];
#pragma unroll
      for (int xx = 0; xx < 6; ++xx) rowv[xx] = w[ii][xx][e];
      w43_bt(rowv, v6[e]);
    }
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
      cv_h16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = v6[e][jj] * sc;
        const _Float16 hb = (_Float16)v;
        hi[e] = hb;
        lo[e] = (_Float16)(v - (float)hb);
      }
      _Float16* dst = p.v + (((long)(6 * ii + jj) * p.T + t) * 2 * p.Cp) + coff;
      *reinterpret_cast<cv_h16x4*>(dst) = hi;
      *reinterpret_cast<cv_h16x4*>(dst + 32) = lo;
    }
  }
}

// The power of two a tile's transformed input is multiplied by: |V| <= 49 amax (two passes of B^T, row sums <= 7), scaled so that
// 49 amax 2^s lands in [2^14, 2^15) -- under fp16's 65504 whatever the activations' magnitude, and with the low terms of everything
// within 2^-13 of the tile's maximum outside fp16's subnormals.  amax = 0 (an all-zero tile) and non-finite amax (an inf input: the
// result must be non-finite, not silently clipped) leave the data unscaled.
__device__ __forceinline__ float w43_tile_scale(float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(49.f * amax, &e);                                           // 49 amax = m 2^e, m in [0.5, 1)
  e = 15 - e;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return ldexpf(1.f, e);
}

// Workgroup of 256 threads; thread = (tile, channel quad).  Cp <= 1024 (the shapes the dispatcher sends here): a workgroup owns
// G = 256 / (Cp / 4) whole tiles, every thread holds its quad's patch in registers while the tile's largest input is found (LDS
// atomic max on the bit pattern: exact and order-independent).  Cp > 1024: one tile per workgroup, the quads in rounds of 256 --
// one round for the maximum, a second (re-reading the patch from L2) for the transform.
template <bool ROUNDS>
// (two wavefronts per SIMD: forcing the three of round 3's kernel spills 18 registers and measured 1.5 - 2.5 % slower per convolution)
__global__ __launch_bounds__(256, ROUNDS ? 1 : 2) void wino43_input_kernel(Wino43InArgs p) {
  __shared__ unsigned gmax[32];
  const int tid = threadIdx.x;
  const int nq = p.Cp >> 2;
  if (tid < 32) gmax[tid] = 0u;
  __syncthreads();
  float w[6][6][4];
  if (!ROUNDS) {
    // (the workgroup is only as large as its tiles need -- 64, 128 or 256 threads: the barrier below then holds back two or four
    // wavefronts, not always four, and more workgroups interleave their load and store phases on a CU)
    const int nt = blockDim.x;
    const int G = nt / nq;
    const int g = tid / nq, q = tid - g * nq;
    const long t = (long)blockIdx.x * G + g;
    const bool on = g < G && t < p.T;
    float amax = on ? w43_load_patch(p, t, 4 * q, w) : 0.f;
    if ((nq & 63) == 0) {                                                  // whole wavefronts per tile: one LDS atomic per wavefront
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
      if (on && (tid & 63) == 0) atomicMax(&gmax[g], __float_as_uint(amax));
    } else if (on) {
      atomicMax(&gmax[g], __float_as_uint(amax));
    }
    __syncthreads();
    if (!on) return;
    const float sc = w43_tile_scale(__uint_as_float(gmax[g]));
    if (q == 0) p.tinv[t] = 1.f / sc;                                      // a power of two: exact
    w43_store_patch(p, t, 4 * q, w, sc);
  } else {
    const long t = blockIdx.x;
    float amax = 0.f;
    for (int q = tid; q < nq; q += 256) amax = fmaxf(amax, w43_patch_amax(p, t, 4 * q));
    atomicMax(&gmax[0], __float_as_uint(amax));
    __syncthreads();
    const float sc = w43_tile_scale(__uint_as_float(gmax[0]));
    if (tid == 0) p.tinv[t] = 1.f / sc;
    for (int q = tid; q < nq; q += 256) {
      (void)w43_load_patch(p, t, 4 * q, w);
      w43_store_patch(p, t, 4 * q, w, sc);
    }
  }
}

struct Wino43OutArgs {
  const float* m;         // [36][T][Cout] raw GEMM results
  const float* fscale;    // [36]: 2^-k of each position's filter scaling
  const float* tinv;      // [T]: 2^-s of each tile's input scaling (wino43_input_kernel)
  const float* bias;
  float* y;               // [B][H][W][Cout] fp32 (nullable)
  __bf16* yhl;            // hl32 split copy (nullable): bf16 pairs, fp16 pairs with out_f16
  int B, H, W, Cout, Cpo, th, tw, act;
  long T, items;          // items = T * Cout / 4
  int out_f16;
  unsigned* range_flag;   // nullable (ConvArgs::range_flag)
};

// one thread = (tile, 4 channels): Y = A^T M A (A^T = [1 1 1 1 1 0; 0 1 -1 2 -1/2 0; 0 1 1 4 1/4 0; 0 1 -1 8 -1/8 1]), bias,
// activation, up to 4 x 4 pixel stores
__global__ __launch_bounds__(256) void wino43_output_kernel(Wino43OutArgs p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.items) return;
  const int nq = p.Cout >> 2;
  const long t = i / nq;
  const int n = (int)(i - t * nq) * 4;
  const int tx = (int)(t % p.tw);
  const long r = t / p.tw;
  const int ty = (int)(r % p.th), b = (int)(r / p.th);
  f32x4 s[4][6];                                                           // A^T M: rows dy, columns jj
  const float ti = p.tinv[t];
#pragma unroll
  for (int jj = 0; jj < 6; ++jj) {
    f32x4 m[6];
#pragma unroll
    for (int ii = 0; ii < 6; ++ii)
      m[ii] = *reinterpret_cast<const f32x4*>(p.m + ((long)(6 * ii + jj) * p.T + t) * p.Cout + n) * (p.fscale[6 * ii + jj] * ti);
    s[0][jj] = m[0] + m[1] + m[2] + m[3] + m[4];
    s[1][jj] = m[1] - m[2] + 2.f * m[3] - 0.5f * m[4];
    s[2][jj] = m[1] + m[2] + 4.f * m[3] + 0.25f * m[4];
    s[3][jj] = m[1] - m[2] + 8.f * m[3] - 0.125f * m[4] + m[5];
  }
  const f32x4 bv = p.bias != nullptr ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  float amax = 0.f;
#pragma unroll
  for (int dy = 0; dy < 4; ++dy) {
    const int y = 4 * ty + dy;
    if (y >= p.H) continue;
    f32x4 o[4];
    o[0] = s[dy][0] + s[dy][1] + s[dy][2] + s[dy][3] + s[dy][4] + bv;
    o[1] = s[dy][1] - s[dy][2] + 2.f * s[dy][3] - 0.5f * s[dy][4] + bv;
    o[2] = s[dy][1] + s[dy][2] + 4.f * s[dy][3] + 0.25f * s[dy][4] + bv;
    o[3] = s[dy][1] - s[dy][2] + 8.f * s[dy][3] - 0.125f * s[dy][4] + s[dy][5] + bv;
#pragma unroll
    for (int dx = 0; dx < 4; ++dx) {
      const int x = 4 * tx + dx;
      if (x >= p.W) continue;
      f32x4 v = o[dx];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = conv_act(v[e], p.act);
      const long px = ((long)b * p.H + y) * p.W + x;
      if (p.y != nullptr) *reinterpret_cast<f32x4*>(p.y + px * p.Cout + n) = v;
      if (p.yhl != nullptr) {
        __attribute__((aligned(8))) unsigned short hi[4], lo[4];
        split4_bits(v, hi, lo, p.out_f16 != 0);
        amax = ocv_amax4(amax, v);
        const long oh = hl_index(px, n, p.Cpo);
        *reinterpret_cast<uint2*>(p.yhl + oh) = *reinterpret_cast<uint2*>(hi);
        *reinterpret_cast<uint2*>(p.yhl + oh + 32) = *reinterpret_cast<uint2*>(lo);
      }
    }
  }
  if (p.out_f16 != 0 && p.yhl != nullptr) ocv_range_note(p.range_flag, amax);
}

}  // namespace

extern "C" size_t ocv_conv3x3_winograd43_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
  const long T = (long)B * ((H + 3) / 4) * ((W + 3) / 4);
  const int Cp = (Cin + 31) / 32 * 32;
  return wino_align((size_t)36 * T * 2 * Cp * sizeof(_Float16)) + wino_align((size_t)36 * T * Cout * sizeof(float)) +
         wino_align((size_t)T * sizeof(float));
}

extern "C" int ocv_conv3x3_winograd43_split_fwd(const void* x_hl, int Cin, const void* u_hi, const void* u_lo, const float* fscale,
                                                const float* cscale, const float* bias, float* y, void* y_hl, int B, int H, int W,
                                                int Cout, int act, int hl_f16, void* workspace, size_t workspace_bytes,
                                                ocv_stream_t stream) {
  OCV_CHECK_ARG(hl_f16 == 0 || hl_f16 == 1, "ocv_conv3x3_winograd43_split_fwd: hl_f16 must be 0 (bf16 pairs) or 1 (fp16 pairs)");
  OCV_CHECK_ARG(x_hl && u_hi && u_lo && fscale && (y || y_hl) && workspace, "ocv_conv3x3_winograd43_split_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && Cin >= 1 && Cout >= 8 && Cout % 8 == 0,
                "ocv_conv3x3_winograd43_split_fwd: bad sizes (Cout must be a multiple of 8, got %d)", Cout);
  OCV_CHECK_ARG(act >= 0 && act <= 3, "ocv_conv3x3_winograd43_split_fwd: unknown activation %d", act);
  OCV_CHECK_ARG((reinterpret_cast<uintptr_t>(x_hl) & 127) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0 &&
                    ocv_aligned16(u_hi) && ocv_aligned16(u_lo) && ocv_aligned16(y) && ocv_aligned16(y_hl) && ocv_aligned16(bias) &&
                    ocv_aligned16(cscale),
                "ocv_conv3x3_winograd43_split_fwd: x_hl must be 128-byte, workspace 256-byte, the rest 16-byte aligned");
  OCV_CHECK_ARG(workspace_bytes >= ocv_conv3x3_winograd43_workspace_bytes(B, H, W, Cin, Cout),
                "ocv_conv3x3_winograd43_split_fwd: workspace too small");
  const int th = (H + 3) / 4, tw = (W + 3) / 4, Cp = (Cin + 31) / 32 * 32;
  const long T = (long)B * th * tw;
  OCV_CHECK_ARG(T * (Cin + 32) * 4 < (1L << 32) && (long)Cout * (Cin + 32) * 2 < (1L << 32),
                "ocv_conv3x3_winograd43_split_fwd: one transformed operand must stay below 4 GiB");
  hipStream_t st = (hipStream_t)stream;
  _Float16* v = (_Float16*)workspace;
  const size_t v_bytes = wino_align((size_t)36 * T * 2 * Cp * sizeof(_Float16));
  const size_t m_bytes = wino_align((size_t)36 * T * Cout * sizeof(float));
  float* m = (float*)((char*)workspace + v_bytes);
  float* tinv = (float*)((char*)workspace + v_bytes + m_bytes);
  if (y_hl != nullptr && Cout % 32 != 0) {
    const int zrc = ocv_zero_async(y_hl, ocv_split_act_elems(B, H, W, Cout) * sizeof(__bf16), st);          // (a launch, not a memset node: common.hpp)
    if (zrc != 0) return zrc;
  }
  Wino43InArgs wi{(const __bf16*)x_hl, v, cscale, tinv, B, H, W, Cp, th, tw, T, hl_f16};
  const int nq = Cp / 4;
  const int in_threads = nq <= 64 ? 64 : (nq <= 128 ? 128 : 256);          // tiles of <= 64 / <= 128 / more channel quads
  const long in_wgs = nq <= 256 ? (T + in_threads / nq - 1) / (in_threads / nq) : T;
  OCV_CHECK_ARG(in_wgs < (1L << 31), "ocv_conv3x3_winograd43_split_fwd: too many tiles");
  if (nq <= 256) hipLaunchKernelGGL(wino43_input_kernel<false>, dim3((unsigned)in_wgs), dim3(in_threads), 0, st, wi);
  else hipLaunchKernelGGL(wino43_input_kernel<true>, dim3((unsigned)in_wgs), dim3(256), 0, st, wi);
  OCV_CHECK_LAUNCH("ocv_conv3x3_winograd43_split_fwd(input transform)");
  ConvArgs a{};
  a.xhl = (const __bf16*)v; a.whi = (const __bf16*)u_hi; a.wlo = (const __bf16*)u_lo; a.y = m;
  a.C1 = Cin; a.Cin = Cin; a.Cout = Cout; a.H = 1; a.W = (int)T; a.ks = 1; a.act = OCV_ACT_NONE; a.ksplit = 1;
  a.zbatch = 36; a.xz_bytes = T * 2 * Cp * (long)sizeof(_Float16); a.wz_bytes = (long)Cout * Cp * (long)sizeof(_Float16);
  a.Cpo = (Cout + 31) / 32 * 32;
  a.f16 = 1;
  const int rc = launch_conv(a, 1, true, st);
  if (rc != 0) return rc;
  Wino43OutArgs wo{m, fscale, tinv, bias, y, (__bf16*)y_hl, B, H, W, Cout, a.Cpo, th, tw, act, T, T * (Cout / 4), hl_f16,
                   (hl_f16 && y_hl != nullptr) ? ocv_range_flag_current() : nullptr};
  hipLaunchKernelGGL(wino43_output_kernel, dim3((unsigned)((wo.items + 255) / 256)), dim3(256), 0, st, wo);
  OCV_CHECK_LAUNCH("ocv_conv3x3_winograd43_split_fwd(output transform)");
  return 0;
}


// =====================================================================================================================
// Patch embedding (Conv2d(C -> E, kernel = stride = 16) + flatten + bias + positional embedding; modules/ObjCAViT.py:287-288,
// 333,362-364, modules/layers.py:11-12,17-22) on a PRE-SPLIT feature map, as 16 split-bf16 GEMMs in ONE launch of the DMA
// kernel.  In the hl32 layout the 16 pixels x C channels of one patch row are 16 C consecutive "channels" (C a multiple of
// 32: whole hi|lo blocks), and the patches of one image row follow each other without a gap: for a fixed ky the feature map
// IS the row-major operand of a 1 x 1 GEMM over a (B gh) x gw pixel grid with 16 C channels whose grid rows lie 16 image rows
// apart (ConvArgs::rowpitch) and whose first row is image row ky (the batch offset xz_bytes).  The 16 raw results
// [ky][B S][E] are summed in a fixed order with bias and positional embedding by patch_sum_kernel.
// Products as everywhere in the convolutions: hi*hi + hi*lo + lo*hi, error <= 2^-17 per product, fp32 accumulation.
// =====================================================================================================================
namespace {
__global__ __launch_bounds__(256) void patch_sum_kernel(const float* __restrict__ part, long M, int E, int S,
                                                        const float* __restrict__ bias, const float* __restrict__ pos,
                                                        long pos_bs, float* __restrict__ out, int nslab) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // float4 index
  const int e4 = E / 4;
  if (idx >= M * e4) return;
  const long m = idx / e4;
  const int e = (int)(idx - m * e4) * 4;
  // K pieces in a FIXED order: four interleaved chains (pieces k = j mod 4), eight loads in flight, then (a0 + a1) + (a2 + a3)
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  const float* src = part + m * E + e;
  const long slab = M * E;
  int k = 0;
  for (; k + 8 <= nslab; k += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (k + j) * slab);
    a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
    a0 += v[4]; a1 += v[5]; a2 += v[6]; a3 += v[7];
  }
  for (; k < nslab; ++k) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + k * slab);
    if ((k & 3) == 0) a0 += v; else if ((k & 3) == 1) a1 += v; else if ((k & 3) == 2) a2 += v; else a3 += v;
  }
  f32x4 a = (a0 + a1) + (a2 + a3);
  if (bias != nullptr) a += *reinterpret_cast<const f32x4*>(bias + e);
  if (pos != nullptr) {
    const long b = m / S, s = m - b * S;
    a += *reinterpret_cast<const f32x4*>(pos + b * pos_bs + s * E + e);
  }
  *reinterpret_cast<f32x4*>(out + m * E + e) = a;
}
}  // namespace

namespace {
// K parts of the patch embedding.  Per patch the contraction is over (ky, kx, c): 16 C / 32 K steps for each of the 16 patch rows,
// 1024 steps at C = 128, which the kernel walks as ONE axis (ConvArgs::zflat) cut in `parts` equal pieces -- one workgroup per (row
// tile, piece), raw partial slabs summed in a fixed order by patch_sum_kernel.  With the pieces = the 16 patch rows (rounds 2 - 4) bs
// 16 was 19 x 16 = 304 workgroups: two rounds of the 256 CUs for 1.19 rounds of work (203 us); 13 pieces are 247 workgroups, ONE round
// of 79 steps.  The part count is the one with the least modelled time: rounds x (steps x ~1.4 us + ~6 us of prologue and stores) + the
// slabs written and read back at ~3 TB/s; at least 16 steps per piece.
int patch_parts(int B, int C, int h, int w, int E) {
  const long M = (long)B * (h / 16) * (w / 16);
  const long tiles = (long)ocv_cdiv(M, CBM) * ocv_cdiv(E, CBN);
  const int S = 16 * (16 * C / CBK);
  int best = 1;
  double cost = 0.0;
  for (int n = 1; n <= 256 && S / n >= 16; ++n) {
    const double c = (double)((tiles * n + 255) / 256) * (1.4 * ((S + n - 1) / n) + 6.0) + (double)n * (double)M * E * 8.0 / 3.0e6;
    if (n == 1 || c < cost) { cost = c; best = n; }
  }
  return best;
}
}  // namespace

extern "C" size_t ocv_patch_embed_split_workspace_bytes(int B, int C, int h, int w, int E) {
  if (B < 1 || C < 32 || C % 32 != 0 || h < 16 || w < 16 || E < 8 || E % 8 != 0) return 0;
  return (size_t)patch_parts(B, C, h, w, E) * B * (h / 16) * (w / 16) * E * sizeof(float);
}

extern "C" int ocv_patch_embed_split_fwd(const void* x_hl, int C, const void* w_hi, const void* w_lo, const float* oscale, int f16,
                                         const float* bias, const float* pos, long pos_bs, float* out, int B, int h, int w, int E,
                                         void* workspace, size_t workspace_bytes, ocv_stream_t stream) {
  OCV_CHECK_ARG(x_hl && w_hi && w_lo && out && workspace, "ocv_patch_embed_split_fwd: null pointer");
  OCV_CHECK_ARG(f16 == 0 || f16 == 1, "ocv_patch_embed_split_fwd: f16 must be 0 (bf16 pairs) or 1 (fp16 pairs)");
  const size_t need = ocv_patch_embed_split_workspace_bytes(B, C, h, w, E);
  OCV_CHECK_ARG(need != 0, "ocv_patch_embed_split_fwd: needs C a multiple of 32 (got %d), E a multiple of 8 (got %d), a map of at "
                "least one 16 x 16 patch (got %d x %d)", C, E, h, w);
  OCV_CHECK_ARG(workspace_bytes >= need, "ocv_patch_embed_split_fwd: workspace too small");
  OCV_CHECK_ARG((reinterpret_cast<uintptr_t>(x_hl) & 127) == 0 && ocv_aligned16(w_hi) && ocv_aligned16(w_lo) && ocv_aligned16(bias) &&
                    ocv_aligned16(pos) && ocv_aligned16(out) && ocv_aligned16(workspace) && (pos == nullptr || pos_bs % 4 == 0),
                "ocv_patch_embed_split_fwd: x_hl must be 128-byte aligned, the rest 16-byte aligned");
  OCV_CHECK_ARG((long)B * h * w * C * 4 < (1L << 32) && (long)E * 16 * C * 2 < (1L << 32),
                "ocv_patch_embed_split_fwd: each operand must be smaller than 4 GiB");
  const int gh = h / 16, gw = w / 16;
  ConvArgs a{};
  a.xhl = (const __bf16*)x_hl; a.whi = (const __bf16*)w_hi; a.wlo = (const __bf16*)w_lo; a.y = (float*)workspace;
  a.C1 = 16 * C; a.Cin = 16 * C; a.Cout = E; a.H = B * gh; a.W = gw; a.ks = 1; a.act = OCV_ACT_NONE;
  a.ksplit = 1;
  a.zflat = patch_parts(B, C, h, w, E);                                       // pieces of the (ky, kx, c) axis: slab k at workspace + k M E
  a.Cpo = (E + 31) / 32 * 32;
  a.f16 = f16; a.oscale = oscale;                                             // (the 16 GEMMs share their output channels' scales)
  a.zbatch = 16;
  a.xz_bytes = (long)w * 2 * C * (long)sizeof(__bf16);                        // one image row down
  a.wz_bytes = (long)E * 16 * C * (long)sizeof(__bf16);
  a.rowpitch = (unsigned)(16L * w * 2 * C * (long)sizeof(__bf16));           // grid rows: 16 image rows apart
  // image b's rows start at image row b h, not b gh 16, when h is not a multiple of 16: only then is the grid not uniform
  OCV_CHECK_ARG(h % 16 == 0 || B == 1, "ocv_patch_embed_split_fwd: the map height must be a multiple of 16 for B > 1 (got %d)", h);
  const int rc = launch_conv(a, 1, true, (hipStream_t)stream);
  if (rc != 0) return rc;
  const long M = (long)B * gh * gw;
  hipLaunchKernelGGL(patch_sum_kernel, dim3((unsigned)((M * (E / 4) + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, M, E, gh * gw, bias, pos, pos_bs, out, a.zflat);
  OCV_CHECK_LAUNCH("ocv_patch_embed_split_fwd(sum)");
  return 0;
}
