// Pointwise (1x1) convolution of the late EfficientNet-B5 stages on a PRE-SPLIT input -- the "hl32" layout the producers
// of the row operand now write (depthwise epilogue, project epilogue) -- with the squeeze-excite gate folded into
// per-image weights.  Row N1 of SURVEY.md section 8; the reference runs conv_pw / conv_pwl of the hub backbone's
// InvertedResidual blocks, walked child by child at modules/DenseFeatureExtractor.py:18-27.
//
//   y[m][co] = act( bias[co] + sum_ci x[m][ci] * Wimg(m)[co][ci] ) + residual[m][co]            (fp32 and / or hl32 out)
//
// Why a second pointwise kernel (csrc/pointwise_split.hip stays for fp32 rows): profiles/r02e_sq.json put the 32-row tile
// kernel at 9.5 - 28 VALU instructions per MFMA and 11 - 24 % matrix-pipe busy on stages 4 - 7.  Three causes, all removed
// by where the operands now come from:
//   * every (row tile, channel tile) workgroup converted its rows fp32 -> (hi, lo) and multiplied them by the gate -- N / 128
//     times redundantly.  Rows arrive split (hi = bf16(v), lo = bf16(v - hi), one 128-byte line per pixel and 32 channels),
//     the gate lives in the weights (ocv_se_gate_weights_fwd packs W * diag(gate[b]) per image): the A path is a pure copy.
//   * a pure copy can be an LDS-DMA (global_load_lds_dwordx4): a FIFTH wavefront issues it NBUF - 1 K-slabs ahead with a
//     counted s_waitcnt vmcnt + raw s_barrier (the pattern of conv_split_dma_kernel); no VGPR staging, no ds_write, and the
//     one-slab register prefetch whose L2 round trip bounded the old kernel is gone.
//   * per wavefront and 16-wide K step the old tile (32 rows x 32 channels) moved 2 KB of weights L2 -> VGPR for three MFMAs:
//     85 B / clk / CU against the 64 the vector L1 delivers.  The wavefront tile is now RT x TN blocks of 32 x 32
//     (64 x 64 by default: half the weight bytes and half the LDS reads per MFMA), weights for three K steps in flight in a
//     static four-slot register ring.
// Tiles never span images when the weights are per image (rows_per_image); rows / channels / K are ragged-safe: rows past
// the image read a zero page, channels past N are not stored, K past Cin multiplies the zero pad channels of the hl32 row.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(256))) float ocv_pwhl_zero_page[64];       // zero-initialised: source of out-of-range rows

struct PHArgs {
  const __bf16* xhl;            // [M][2 Cp] hl32
  const __bf16* wp;             // packed B-operand fragments (hip_ops.SplitWeight order); + image * w_img_elems when per image
  long w_img_elems;
  const float *bias, *res;
  float* y;                     // nullable
  __bf16* yhl;                  // nullable: hl32 copy of the output, Cpo = ceil32(N) channels, pad channels written as zero
  long M;
  int K, Cp, Kp, N, Cpo, act;
  int rows_per_image, tiles_per_image;     // shared weights: one "image" of M rows
  int nbx, nby, col_major;
};

__device__ __forceinline__ float ph_act(float v, int act) {
  switch (act) {
    case OCV_ACT_RELU: return fmaxf(v, 0.f);
    case OCV_ACT_LEAKY_RELU: return v > 0.f ? v : 0.01f * v;
    case OCV_ACT_SILU: return fast_silu(v);
    case OCV_ACT_SIGMOID: return fast_sigmoid(v);
    default: return v;
  }
}

__device__ __forceinline__ bf16x8 ph_ldb8(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

__device__ __forceinline__ f32x16 ph_mfma3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  return acc;
}

constexpr int PH_TS = 32;                       // floats per LDS row of the epilogue's transpose scratch: UNPADDED -- the four
                                                // 16-lane groups of its ds_read_b128 each cover whole 256-byte bank rows (a padded
                                                // row of 40 floats costs 3 LDS cycles per group: MI355X_MICROARCH.md, LDS table)
constexpr int PH_TSCRATCH = 32 * PH_TS;         // floats per wavefront
constexpr int PH_SLABB = 256;                   // bytes per row and K slab: 64 channels = two hl32 blocks (hi | lo | hi | lo)

// bias + activation (+ residual) of one 32 x 32 accumulator tile, transposed through the wavefront's own LDS scratch so
// that it leaves as 16-byte stores of whole 128-byte row segments (pointwise_split.hip, store_tile_lds); optionally also
// as the hl32 split of the same values (two 8-byte stores per lane: hi and lo quads of the channel block).
// rows_left = rows of the tile that exist (image / tensor end).
__device__ __forceinline__ void ph_store_tile(const PHArgs& p, const f32x16& acc, float* scratch, long m_base, int rows_left,
                                              int n0, int lane) {
  const int l31 = lane & 31, hh = lane >> 5;
  const int n = n0 + l31;
  const float bv = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
  // (one uniform branch per tile, not a switch per element)
  if (p.act == OCV_ACT_SILU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) scratch[acc_row(r, hh) * PH_TS + l31] = fast_silu(acc[r] + bv);
  } else if (p.act == OCV_ACT_NONE) {
#pragma unroll
    for (int r = 0; r < 16; ++r) scratch[acc_row(r, hh) * PH_TS + l31] = acc[r] + bv;
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) scratch[acc_row(r, hh) * PH_TS + l31] = ph_act(acc[r] + bv, p.act);
  }
  const int c4 = lane & 7, ncol = n0 + 4 * c4;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = (lane >> 3) + 8 * it;
    const long m = m_base + row;
    float4 v = *reinterpret_cast<const float4*>(scratch + row * PH_TS + 4 * c4);
    if (row < rows_left) {
      if (ncol < p.N) {
        if (p.res != nullptr) {
          const float4 q = ld4(p.res + m * p.N + ncol);
          v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        if (p.y != nullptr) *reinterpret_cast<float4*>(p.y + m * p.N + ncol) = v;
      } else {
        v = make_float4(0.f, 0.f, 0.f, 0.f);       // pad channels of the split copy
      }
      if (p.yhl != nullptr && ncol < p.Cpo) {
        const float f[4] = {v.x, v.y, v.z, v.w};
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const __bf16 hb = (__bf16)f[e];
          h[e] = hb;
          l[e] = (__bf16)(f[e] - (float)hb);
        }
        __bf16* d = p.yhl + m * 2 * p.Cpo + (ncol >> 5) * 64 + (ncol & 31);
        *reinterpret_cast<bf16x4*>(d) = h;
        *reinterpret_cast<bf16x4*>(d + 32) = l;
      }
    }
  }
}

// RT row blocks x TN channel blocks of 32 x 32 per consumer wavefront; four consumer wavefronts side by side over the
// channels (workgroup tile 32 RT rows x 128 TN channels) + one DMA wavefront; NBUF LDS slabs of 32 RT rows x 256 bytes.
template <int RT, int TN, int NBUF>
__global__ __launch_bounds__(320) void pw_hl_kernel(PHArgs p) {
  constexpr int ROWS = 32 * RT;
  constexpr int BUFB = ROWS * PH_SLABB;
  constexpr int PIECES = ROWS / 4;               // LDS-DMA instructions per slab (1 KB each: 4 rows x 16 chunks)
  static_assert(NBUF >= 2 && (NBUF - 2) * PIECES <= 63, "counted vmcnt must fit its 6-bit field");
  static_assert(NBUF * BUFB >= 4 * PH_TSCRATCH * (int)sizeof(float), "epilogue scratch aliases the slab ring");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware, bijective workgroup -> tile map (pointwise_split.hip): every XCD gets a contiguous run of tiles in the
  // order the host chose (channel-block major when the weights would not fit an L2: an XCD keeps one weight slice
  // resident and streams rows; row-block major otherwise: rows read once, all of W resident)
  int bx, by;
  {
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (p.col_major) { by = wg / p.nbx; bx = wg % p.nbx; }
    else { bx = wg / p.nby; by = wg % p.nby; }
  }
  const int img = bx / p.tiles_per_image, tl = bx - img * p.tiles_per_image;
  const int r0 = tl * ROWS;                                          // first row of the tile inside its image
  const int rows_left = min(ROWS, p.rows_per_image - r0);             // >= 1
  const long m0 = (long)img * p.rows_per_image + r0;
  const int nslab = (p.Cp + 63) >> 6;

  if (wave == 4) {
    // =========================== PRODUCER (LDS-DMA issuer) ===========================
    // piece i moves rows 4 i .. 4 i + 3 of the slab; lane L lands at piece base + 16 L = row 4 i + (L >> 4), stored chunk
    // L & 15, which must hold LOGICAL chunk (L & 15) ^ (row & 15): a ds_read_b128 is served in four groups of 16 lanes
    // (lanes {0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS), one 256-byte bank row = 16 slots per cycle; the 16 rows of
    // a group are distinct modulo 16, so with the 4-bit key their reads of one K octet land on 16 different slots
    // (unpadded 256-byte rows would put all of them on one; a 3-bit key -- the first version of this kernel -- on 8: PMC
    // showed SQ_LDS_BANK_CONFLICT = SQ_BUSY_CYCLES, every A-fragment read took two cycles per group).
    const int lrow = lane >> 4, cst = lane & 15;
    const unsigned rowb = (unsigned)p.Cp * 4u;                        // bytes per hl32 row
    const char* xb = (const char*)p.xhl + m0 * (long)rowb;
    auto issue = [&](int it) {
      unsigned char* base = lds + (it % NBUF) * BUFB;
      const unsigned kb = (unsigned)it * PH_SLABB;                    // byte offset of the slab inside a row
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int row = 4 * i + lrow;
        const unsigned cb = kb + (unsigned)((cst ^ (row & 15)) * 16);
        const bool ok = row < rows_left && cb < rowb;
        const void* src = ok ? (const void*)(xb + (long)row * rowb + cb) : (const void*)ocv_pwhl_zero_page;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + i * 1024), 16, 0, 0);
      }
    };
#define OCV_PH_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
    // slabs 0 .. NBUF - 2 up front; in interval it (consumers on slab it) slab it + NBUF - 1 goes into the buffer slab
    // it - 1 has just left, then only slab it + 1 is waited for: (NBUF - 2) slabs of pieces may stay in flight
    const int pro = min(NBUF - 1, nslab);
    for (int it = 0; it < pro; ++it) issue(it);
    if (pro == NBUF - 1) OCV_PH_WAIT_VM((NBUF - 2) * PIECES);
    else OCV_PH_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    for (int it = 0; it < nslab; ++it) {
      if (it + NBUF - 1 < nslab) {
        issue(it + NBUF - 1);
        OCV_PH_WAIT_VM((NBUF - 2) * PIECES);
      } else {
        OCV_PH_WAIT_VM(0);
      }
      __builtin_amdgcn_s_barrier();
    }
#undef OCV_PH_WAIT_VM
    return;
  }

  // =========================== CONSUMERS ===========================
  const int l31 = lane & 31, hh = lane >> 5;
  const int nsteps = p.Kp >> 4;                                      // 16-wide K steps that have weights
  const int ntl_all = (p.N + 31) >> 5;
  const __bf16* wimg = p.wp + (long)img * p.w_img_elems;
  const __bf16* wf[TN];
  int jt[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    jt[j] = (by * 4 + wave) * TN + j;
    wf[j] = wimg + ((long)min(jt[j], ntl_all - 1) * nsteps * 2) * 512 + lane * 8;       // + step * 1024 (+ 512: lo)
  }
  const bool any_cols = jt[0] < ntl_all;                              // a wavefront past the last channel block only keeps the barriers

  f32x16 acc[RT][TN];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[rt][j] = f32x16{0};

  // weights: a static ring of four K steps, three in flight ahead of the MFMAs that use them (straight from L2 into the
  // B operand: a wavefront's load is one contiguous 1 KB fragment per channel block and part)
  bf16x8 bh[4][TN], bl[4][TN];
  auto load_b = [&](int gs, bf16x8 (&h)[TN], bf16x8 (&l)[TN]) {
    const long o = (long)min(gs, nsteps - 1) * 1024;                 // past the end: re-read the last step (multiplies zeros)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      h[j] = ph_ldb8(wf[j] + o);
      l[j] = ph_ldb8(wf[j] + o + 512);
    }
  };
  load_b(0, bh[0], bl[0]);
  load_b(1, bh[1], bl[1]);
  load_b(2, bh[2], bl[2]);

  // A fragments: row 32 rt + l31 of the slab, logical chunk (16-byte unit of the 256-byte slab row)
  //   c(s, part) = (s >> 1) * 8 + part * 4 + (s & 1) * 2 + hh,   stored at c ^ (row & 15)  (the producer's swizzle)
  const int key = l31 & 15;
  const unsigned char* arow = lds + l31 * PH_SLABB;
  int coff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) coff[s] = ((((s >> 1) * 8 + (s & 1) * 2 + hh) ^ key) * 16);
  const int lo_x = (4 ^ 0) * 16;                                      // part toggles bit 2 of the chunk index: XOR 64 bytes

  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();                                       // slab 0 has landed (producer waited for it)
  asm volatile("" ::: "memory");
  for (int it = 0; it < nslab; ++it) {
    const unsigned char* base = arow + (it % NBUF) * BUFB;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int gs = it * 4 + s;
      load_b(gs + 3, bh[(s + 3) & 3], bl[(s + 3) & 3]);
      if (any_cols) {
        bf16x8 ah[RT], al[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          ah[rt] = *reinterpret_cast<const bf16x8*>(base + rt * 32 * PH_SLABB + coff[s]);
          al[rt] = *reinterpret_cast<const bf16x8*>(base + rt * 32 * PH_SLABB + (coff[s] ^ lo_x));
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[rt][j] = ph_mfma3(ah[rt], al[rt], bh[s][j], bl[s][j], acc[rt][j]);
      }
    }
    // this wavefront's reads of the slab have returned (the MFMAs consumed them); release the buffer
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (!any_cols) return;
  // epilogue: the ring is idle (every consumer is past the last barrier, the producer has nothing in flight)
  float* scratch = reinterpret_cast<float*>(lds) + wave * PH_TSCRATCH;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      if (jt[j] < ntl_all && rt * 32 < rows_left)
        ph_store_tile(p, acc[rt][j], scratch, m0 + rt * 32, rows_left - rt * 32, jt[j] * 32, lane);
}

template <int RT, int TN, int NBUF>
int launch_ph(PHArgs a, hipStream_t st) {
  constexpr int ROWS = 32 * RT;
  constexpr size_t LDS = (size_t)NBUF * ROWS * PH_SLABB;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)pw_hl_kernel<RT, TN, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  a.tiles_per_image = ocv_cdiv(a.rows_per_image, ROWS);
  a.nbx = (int)(a.M / a.rows_per_image) * a.tiles_per_image;
  a.nby = ocv_cdiv(a.N, 128 * TN);
  a.col_major = (long)a.N * a.Kp * 4 > (3L << 20);
  hipLaunchKernelGGL((pw_hl_kernel<RT, TN, NBUF>), dim3((unsigned)((long)a.nbx * a.nby)), dim3(320), LDS, st, a);
  OCV_CHECK_LAUNCH("ocv_pointwise_hl_fwd");
  return 0;
}

struct PhCfg { int rt = 0, tn = 0; };
PhCfg& ph_cfg() {
  static PhCfg c;                                                     // (0, 0) = automatic; ocv_pointwise_hl_set_dispatch overrides (tests, tools)
  return c;
}

}  // namespace

extern "C" int ocv_pointwise_hl_set_dispatch(int rt, int tn) {
  OCV_CHECK_ARG((rt == 0 && tn == 0) || ((rt == 1 || rt == 2 || rt == 4) && (tn == 1 || tn == 2)),
                "ocv_pointwise_hl_set_dispatch: (rt, tn) must be (0, 0) = automatic or rt in {1, 2, 4}, tn in {1, 2}");
  ph_cfg().rt = rt;
  ph_cfg().tn = tn;
  return 0;
}

extern "C" int ocv_pointwise_hl_fwd(const void* x_hl, int Cin, const void* w_packed, long w_image_elems, int rows_per_image,
                                    const float* bias, const float* residual, float* y, void* y_hl, long M, int Cout, int act,
                                    ocv_stream_t stream) {
  OCV_CHECK_ARG(x_hl && w_packed && (y || y_hl), "ocv_pointwise_hl_fwd: null pointer");
  OCV_CHECK_ARG(M >= 0 && Cin >= 8 && Cin % 8 == 0 && Cout >= 1, "ocv_pointwise_hl_fwd: Cin must be a positive multiple of 8 (got M=%ld Cin=%d Cout=%d)", M, Cin, Cout);
  OCV_CHECK_ARG(act >= 0 && act <= OCV_ACT_SIGMOID, "ocv_pointwise_hl_fwd: unknown activation %d", act);
  OCV_CHECK_ARG(Cout % 4 == 0, "ocv_pointwise_hl_fwd: Cout must be a multiple of 4 (16-byte row segments), got %d", Cout);
  OCV_CHECK_ARG(y_hl == nullptr || Cout % 8 == 0, "ocv_pointwise_hl_fwd: the split output needs Cout to be a multiple of 8");
  OCV_CHECK_ARG(w_image_elems >= 0 && (w_image_elems == 0 || rows_per_image >= 1), "ocv_pointwise_hl_fwd: per-image weights need rows_per_image");
  OCV_CHECK_ARG(w_image_elems == 0 || M % rows_per_image == 0, "ocv_pointwise_hl_fwd: M must be a whole number of images");
  OCV_CHECK_ARG((reinterpret_cast<uintptr_t>(x_hl) & 127) == 0 && ocv_aligned16(w_packed) && ocv_aligned16(y) && ocv_aligned16(y_hl) &&
                    ocv_aligned16(residual) && (w_image_elems & 7) == 0,
                "ocv_pointwise_hl_fwd: x_hl must be 128-byte aligned, the other operands 16-byte aligned");
  const int Cp = (Cin + 31) / 32 * 32;
  OCV_CHECK_ARG(M * (long)Cp * 4 < (1L << 32), "ocv_pointwise_hl_fwd: the split input must be smaller than 4 GiB");
  if (M == 0) return 0;
  PHArgs a{};
  a.xhl = (const __bf16*)x_hl; a.wp = (const __bf16*)w_packed; a.w_img_elems = w_image_elems;
  a.bias = bias; a.res = residual; a.y = y; a.yhl = (__bf16*)y_hl;
  a.M = M; a.K = Cin; a.Cp = Cp; a.Kp = (Cin + 15) / 16 * 16; a.N = Cout; a.Cpo = (Cout + 31) / 32 * 32; a.act = act;
  a.rows_per_image = w_image_elems ? rows_per_image : (int)M;
  OCV_CHECK_ARG(w_image_elems != 0 || M < (1L << 31), "ocv_pointwise_hl_fwd: M too large");
  hipStream_t st = (hipStream_t)stream;
  // Tile choice (measured on MI355X at bs = 16, tools/history/run_pw_hl.py -> profiles/r03_pointwise_hl_sweep.txt;
  // ocv_pointwise_hl_set_dispatch force one): long K (the project layers, 512 -> 3072) 64 rows per wavefront, two channel
  // blocks when the output is wide as well; short K (the expand layers, K <= 304: 2 - 5 slabs) the smallest tile -- those
  // launches are bound by the latency of a workgroup's single pass load -> multiply -> store, and the most workgroups in
  // flight win, as they did for the fp32-row kernel.
  int rt = ph_cfg().rt, tn = ph_cfg().tn;
  if (rt == 0) {
    if (Cin >= 512) { rt = 2; tn = Cout >= 512 ? 2 : 1; }
    else { rt = 1; tn = 1; }
  }
  if (rt == 1 && tn == 1) return launch_ph<1, 1, 4>(a, st);
  if (rt == 1 && tn == 2) return launch_ph<1, 2, 4>(a, st);
  if (rt == 2 && tn == 1) return launch_ph<2, 1, 4>(a, st);
  if (rt == 2 && tn == 2) return launch_ph<2, 2, 4>(a, st);
  if (rt == 4 && tn == 1) return launch_ph<4, 1, 3>(a, st);
  return launch_ph<4, 2, 3>(a, st);
}
