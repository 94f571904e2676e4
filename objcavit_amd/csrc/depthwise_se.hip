// Depthwise k x k convolution (+ folded BatchNorm + SiLU) that also produces the squeeze-excite pooling sums, and the
// squeeze-excite gate computed from them -- 3 launches per MBConv block where csrc/encoder_nhwc.hip needs 5
// (depthwise, channel sum, mean, hidden layer, gate), and no second pass over the depthwise output (the largest
// activations of the network: 295 MB at 120 x 160 x 240 x 16 images).
// Row N1 of SURVEY.md section 8; the reference runs conv_dw / bn / act / se of the hub backbone's blocks
// (modules/DenseFeatureExtractor.py:18-27,149).
//
// dw_slide_kernel<K,S,PX>: a workgroup owns (image, chunk of <= 64 channel quads, PL work items); thread = (channel
//   quad q, work item pl); a work item is a PX-wide, RY-tall strip of outputs walked top to bottom with the K input
//   rows under it held in registers (see the kernel).  Every thread keeps the running sum of its own outputs; the
//   work items of a quad are added through LDS in lane order and the workgroup writes ONE partial per channel:
//   part[b][tile][c].  Fixed order everywhere -> bit-reproducible.
// se_hidden_partials_kernel + se_gate_hid_kernel: pooling mean from the partials, hidden layer, gate (see there).
#include <stdlib.h>

#include "common.hpp"
#include "../../include/objcavit_hip.h"

namespace {

typedef __bf16 dw_bf16x4 __attribute__((ext_vector_type(4)));

struct DWSArgs {
  const float *in, *w, *bias;
  float *out, *part;               // out nullable when out_hl is given
  __bf16* out_hl;                  // nullable: the output in the "hl32" split layout (C a multiple of 32), what the project
                                   // 1x1 convolution reads by LDS-DMA (csrc/pointwise_hl.hip): split ONCE, where it is produced
  int C, H, W, Ho, Wo, pad_t, pad_l;
  int QL, PL, RY;                  // quads per chunk, pixel lanes, output rows per work item
  int wox, nwork, tiles, chunks;   // x-blocks per output row, work items per image, workgroups per (image, chunk), chunks
};

// Sliding window: a work item = (x-block of PX output columns, run of RY output rows) of one channel quad.  The K input
// rows under the current output row live in VGPRs as a ring (static indices: the row loop is unrolled K times); moving
// down one output row loads only the S NEW input rows.  Every input element is therefore fetched (PX - 1 + K) / PX / S
// times per run (+ K - S rows of run-in) instead of K * (PX - 1 + K) / PX times: 1.5x instead of 4.5x at k = 3,
// 3x instead of 15x at k = 5 (PX = 2).  The L1 <- L2 path, not HBM, is what bounded the plain kernel (its 18 / 40
// loads per x-block moved 6.5 TB/s out of L2 for 2.9 TB/s of useful traffic).
// Weights: K = 3 in VGPRs (36); K = 5 in LDS (the 100 registers go to the window instead).
// (Round 4 built the squeeze-excite gate INTO this launch -- the image's last workgroup to arrive formed it: correct under load, 50
// launches fewer, not faster (profiles/r04_se_tail.txt); removed in round 5, source in tools/diag/se_tail.patch.txt.)
template <int K, int S, int PX>
__global__ __launch_bounds__(256, 2) void dw_slide_kernel(DWSArgs p) {
  constexpr int NIN = (PX - 1) * S + K;
  constexpr bool WLDS = K > 3;
  __shared__ float4 red[256];
  __shared__ float4 wsh[WLDS ? K * K * 64 : 1];
  const int tid = threadIdx.x;
  const int q = tid % p.QL, pl = tid / p.QL;
  const int c4n = p.C >> 2;
  // XCD-aware, bijective workgroup -> (image, chunk, tile) map: consecutive workgroup ids go round-robin to the 8 XCDs,
  // and neighbouring work items share halo rows / columns through their XCD's own L2.  Give every XCD a contiguous run
  // of tiles (whole images at bs = 16).
  int tile, chunk;
  long b;
  {
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    const int qq = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    tile = wg % p.tiles;
    const int rest = wg / p.tiles;
    chunk = rest % p.chunks;
    b = rest / p.chunks;
  }
  const int cq = chunk * p.QL + q;
  const int wi = tile * p.PL + pl;
  const bool active = pl < p.PL && cq < c4n && wi < p.nwork;
  const int c = (cq < c4n ? cq : 0) * 4;

  float4 wv[WLDS ? 1 : K * K];
  if (WLDS) {
    for (int t = pl; t < K * K; t += 256 / p.QL) wsh[t * 64 + q] = ld4(p.w + (long)t * p.C + c);
    __syncthreads();
  } else {
#pragma unroll
    for (int t = 0; t < K * K; ++t) wv[t] = ld4(p.w + (long)t * p.C + c);
  }
  const float4 bv = p.bias ? ld4(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);

  if (active) {
    const int run = wi / p.wox, xb = wi - run * p.wox;
    const int y0 = run * p.RY, y1 = min(p.Ho, y0 + p.RY);
    const int ox = xb * PX, ix0 = ox * S - p.pad_l;
    const int iyb = y0 * S - p.pad_t;                           // input row of window-relative row 0
    const bool allc = ix0 >= 0 && ix0 + NIN <= p.W;
    const float* ib = p.in + b * (long)p.H * p.W * p.C + c;
    const long opix = (b * p.Ho + y0) * (long)p.Wo + ox;
    float* ob = p.out + opix * p.C + c;
    __bf16* ohb = p.out_hl + opix * 2 * p.C + (c >> 5) * 64 + (c & 31);       // hl32: hi quad here, lo quad 32 elements on
    float4 win[K][NIN];
    auto load_row = [&](int rel, float4 (&dst)[NIN]) {
      const int iy = iyb + rel;
      const bool rok = iy >= 0 && iy < p.H;
      const float* row = ib + ((long)(rok ? iy : 0) * p.W + ix0) * p.C;
      if (rok && allc) {
#pragma unroll
        for (int j = 0; j < NIN; ++j) dst[j] = ld4(row + (long)j * p.C);
      } else {
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
          const int ix = ix0 + j;
          dst[j] = (rok && ix >= 0 && ix < p.W) ? ld4(row + (long)j * p.C) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    };
    // PRE (k = 3): the S input rows the NEXT output row adds are requested BEFORE this row's stores, into the ring slots
    // this row has just finished with.  A store counts in vmcnt like a load and vmcnt retires in order, so loads issued
    // behind the stores cannot be waited for without waiting for the stores' acknowledgement as well (240 x 320 x 24:
    // 79 -> 66 us).  It keeps the accumulators alive across the load issue, +16 VGPRs: at k = 5 that costs the second
    // wavefront per SIMD and loses (56 -> 80 us at 30 x 40 x 1056), so k = 5 loads at the top of the row.
    constexpr bool PRE = K == 3;
#pragma unroll
    for (int i = 0; i < (PRE ? K : K - S); ++i) load_row(i, win[i % K]);
    for (int y = y0; y < y1; y += K) {
#pragma unroll
      for (int u = 0; u < K; ++u) {
        if (y + u < y1) {
          if (!PRE) {
#pragma unroll
            for (int sidx = 0; sidx < S; ++sidx) load_row((y + u - y0) * S + K - S + sidx, win[(S * u + K - S + sidx) % K]);
          }
          float4 acc[PX];
#pragma unroll
          for (int o = 0; o < PX; ++o) acc[o] = bv;
#pragma unroll
          for (int r = 0; r < K; ++r) {
            // LDS-resident weights: re-read one row of taps at a time (the fence keeps the compiler from hoisting all
            // K x K x 4 weights of all K unrolled bodies into registers, which costs the occupancy the window needs)
            if (WLDS) asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < K; ++j) {
              const float4 w4 = WLDS ? wsh[(r * K + j) * 64 + q] : wv[WLDS ? 0 : r * K + j];
#pragma unroll
              for (int o = 0; o < PX; ++o) {
                const float4 x = win[(S * u + r) % K][o * S + j];
                acc[o].x = fmaf(w4.x, x.x, acc[o].x); acc[o].y = fmaf(w4.y, x.y, acc[o].y);
                acc[o].z = fmaf(w4.z, x.z, acc[o].z); acc[o].w = fmaf(w4.w, x.w, acc[o].w);
              }
            }
          }
          if (PRE && y + u + 1 < y1) {
#pragma unroll
            for (int sidx = 0; sidx < S; ++sidx) load_row((y + u + 1 - y0) * S + K - S + sidx, win[(S * (u + 1) + K - S + sidx) % K]);
          }
          const long roff = (long)(y + u - y0) * p.Wo;
#pragma unroll
          for (int o = 0; o < PX; ++o) {
            if (ox + o < p.Wo) {
              float4 r4 = acc[o];
              r4.x = fast_silu(r4.x); r4.y = fast_silu(r4.y); r4.z = fast_silu(r4.z); r4.w = fast_silu(r4.w);
              if (p.out != nullptr) *reinterpret_cast<float4*>(ob + (roff + o) * p.C) = r4;
              if (p.out_hl != nullptr) {
                const float f[4] = {r4.x, r4.y, r4.z, r4.w};
                dw_bf16x4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const __bf16 hb = (__bf16)f[e];
                  h[e] = hb;
                  l[e] = (__bf16)(f[e] - (float)hb);
                }
                __bf16* d = ohb + (roff + o) * 2 * p.C;
                *reinterpret_cast<dw_bf16x4*>(d) = h;
                *reinterpret_cast<dw_bf16x4*>(d + 32) = l;
              }
              sum.x += r4.x; sum.y += r4.y; sum.z += r4.z; sum.w += r4.w;
            }
          }
        }
      }
    }
  }
  red[tid] = sum;
  __syncthreads();
  if (pl == 0 && cq < c4n) {
    float4 t = red[q];
    for (int l = 1; l < p.PL; ++l) {
      const float4 u = red[l * p.QL + q];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    float* dst = p.part + ((b * p.tiles + tile) * (long)p.C) + c;
    *reinterpret_cast<float4*>(dst) = t;
  }
}

// workgroup geometry shared by the size query and the launch
struct DWSGeom { int QL, PL, RY, PX, chunks, wox, nruns, nwork, tiles; };
DWSGeom dws_geom(int B, int C, int Ho, int Wo, int k, int stride) {
  DWSGeom g;
  g.PX = (stride == 1 && k == 3) ? 4 : ((stride == 2 && k == 5) ? 1 : 2);
  const int c4n = C / 4;
  g.chunks = (c4n + 63) / 64;
  g.QL = (c4n + g.chunks - 1) / g.chunks;
  g.PL = 256 / g.QL;
  g.wox = (Wo + g.PX - 1) / g.PX;
  // rows per work item: up to 8 (measured: 16 was slower on the 120 x 160 and 240 x 320 stages, fewer workgroups in
  // flight), shorter while the launch would have fewer than ~800 workgroups (a run re-reads k - stride rows)
  int ry = 8;
  while (ry > 2 && (long)g.wox * ((Ho + ry - 1) / ry) * g.chunks * B < 800L * g.PL) ry >>= 1;
  g.RY = ry;
  g.nruns = (Ho + ry - 1) / ry;
  g.nwork = g.wox * g.nruns;
  g.tiles = (g.nwork + g.PL - 1) / g.PL;
  return g;
}

// squeeze-excite gate from the pooling partials, two small launches (both latency-bound, so both are spread wide):
//   se_hidden_partials_kernel  grid (ceil(R / 16), B): mean[c] = (sum_tile part) / P into LDS (every workgroup of an
//                              image repeats this: tiles x C floats from L2), then one wavefront per hidden unit:
//                              hid[b][r] = silu(b1[r] + W1[r] . mean)
//   se_gate_hid_kernel         grid (ceil(C / 256), B): gate[b][c] = sigmoid(b2[c] + sum_r W2t[r][c] hid[b][r])
__global__ __launch_bounds__(1024) void se_hidden_partials_kernel(const float* __restrict__ part, int tiles, float inv,
                                                                 const float* __restrict__ w1, const float* __restrict__ b1,
                                                                 float* __restrict__ hid, int C, int R) {
  extern __shared__ float sm[];            // mean[C] | red[TG][C]
  float* mean = sm;
  float* red = sm + C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long b = blockIdx.y;
  const float* pb = part + b * tiles * (long)C;
  // pooling: item = (channel QUAD, row group); every thread keeps eight independent 16-byte loads in flight.  (Round 4: the first
  // version summed one float per load, four in flight -- at bs 1, where a depthwise launch leaves up to 600 partial rows per
  // image, that was 38 dependent round trips and 10.7 us per launch, 0.42 ms of a 4 ms forward; profiles/r04a_*.)  Out-of-range
  // rows are clamped to a valid address and selected away: no branch around a load.  Fixed order: rows ascending within a group,
  // groups ascending.
  const int nq = C >> 2;                    // C % 4 == 0 (the depthwise kernels' contract)
  const int TG = nq >= 1024 ? 1 : min(tiles, 1024 / nq);
  for (int idx = tid; idx < TG * nq; idx += 1024) {
    const int q = idx % nq, tg = idx / nq;
    float4 a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = pb + 4 * q;
    for (int t0 = tg; t0 < tiles; t0 += 8 * TG) {
      float4 u[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) u[j] = ld4(src + (long)min(t0 + j * TG, tiles - 1) * C);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = t0 + j * TG < tiles;
        a[j].x += ok ? u[j].x : 0.f; a[j].y += ok ? u[j].y : 0.f; a[j].z += ok ? u[j].z : 0.f; a[j].w += ok ? u[j].w : 0.f;
      }
    }
    float4 r;
    r.x = ((a[0].x + a[1].x) + (a[2].x + a[3].x)) + ((a[4].x + a[5].x) + (a[6].x + a[7].x));
    r.y = ((a[0].y + a[1].y) + (a[2].y + a[3].y)) + ((a[4].y + a[5].y) + (a[6].y + a[7].y));
    r.z = ((a[0].z + a[1].z) + (a[2].z + a[3].z)) + ((a[4].z + a[5].z) + (a[6].z + a[7].z));
    r.w = ((a[0].w + a[1].w) + (a[2].w + a[3].w)) + ((a[4].w + a[5].w) + (a[6].w + a[7].w));
    *reinterpret_cast<float4*>(red + tg * C + 4 * q) = r;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 1024) {
    float s = red[c];
    for (int tg = 1; tg < TG; ++tg) s += red[tg * C + c];
    mean[c] = s * inv;
  }
  __syncthreads();
  const int r = blockIdx.x * 16 + wave;
  if (r >= R) return;
  // one wavefront per hidden unit: a lane takes channel QUADS lane, lane + 64, ... -- 16-byte loads, four of them in flight
  // (C <= 1024: the whole row of W1 in one round trip; one float per load took C / 256 dependent rounds)
  const float* wr = w1 + (long)r * C;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int q = lane;
  for (; q + 192 < nq; q += 256) {
    const float4 u0 = ld4(wr + 4 * q), u1 = ld4(wr + 4 * (q + 64)), u2 = ld4(wr + 4 * (q + 128)), u3 = ld4(wr + 4 * (q + 192));
    const float4 m0 = *reinterpret_cast<const float4*>(mean + 4 * q), m1 = *reinterpret_cast<const float4*>(mean + 4 * (q + 64));
    const float4 m2 = *reinterpret_cast<const float4*>(mean + 4 * (q + 128)), m3 = *reinterpret_cast<const float4*>(mean + 4 * (q + 192));
    s0 = fmaf(u0.x, m0.x, fmaf(u0.y, m0.y, fmaf(u0.z, m0.z, fmaf(u0.w, m0.w, s0))));
    s1 = fmaf(u1.x, m1.x, fmaf(u1.y, m1.y, fmaf(u1.z, m1.z, fmaf(u1.w, m1.w, s1))));
    s2 = fmaf(u2.x, m2.x, fmaf(u2.y, m2.y, fmaf(u2.z, m2.z, fmaf(u2.w, m2.w, s2))));
    s3 = fmaf(u3.x, m3.x, fmaf(u3.y, m3.y, fmaf(u3.z, m3.z, fmaf(u3.w, m3.w, s3))));
  }
  for (; q < nq; q += 64) {
    const float4 u0 = ld4(wr + 4 * q);
    const float4 m0 = *reinterpret_cast<const float4*>(mean + 4 * q);
    s0 = fmaf(u0.x, m0.x, fmaf(u0.y, m0.y, fmaf(u0.z, m0.z, fmaf(u0.w, m0.w, s0))));
  }
  const float s = wave_sum((s0 + s1) + (s2 + s3));
  if (lane == 0) hid[b * R + r] = fast_silu(s + b1[r]);
}

__global__ __launch_bounds__(256) void se_gate_hid_kernel(const float* __restrict__ hid, const float* __restrict__ w2t,
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
  for (; r + 15 < R; r += 16) {                       // sixteen loads in flight (four were: R / 4 dependent round trips)
    float u[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) u[j] = w[(long)(r + j) * C];
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
      s0 = fmaf(u[j], hs[r + j], s0);
      s1 = fmaf(u[j + 1], hs[r + j + 1], s1);
      s2 = fmaf(u[j + 2], hs[r + j + 2], s2);
      s3 = fmaf(u[j + 3], hs[r + j + 3], s3);
    }
  }
  for (; r + 3 < R; r += 4) {
    s0 = fmaf(w[(long)r * C], hs[r], s0);
    s1 = fmaf(w[(long)(r + 1) * C], hs[r + 1], s1);
    s2 = fmaf(w[(long)(r + 2) * C], hs[r + 2], s2);
    s3 = fmaf(w[(long)(r + 3) * C], hs[r + 3], s3);
  }
  for (; r < R; ++r) s0 = fmaf(w[(long)r * C], hs[r], s0);
  gate[b * C + c] = fast_sigmoid((s0 + s1) + (s2 + s3));
}

// Both launches above as ONE, for the blocks whose squeeze-excite weights are small (round 4: B5's stages 1 - 5, 27 of 39 blocks):
// grid (ceil(C / 256), B), 1024 threads; every workgroup of an image repeats the pooling and ALL hidden units (<= 64 of them: W1 is
// <= 186 KB, L2-resident) and then forms the gate of its own 256 channels.  The arithmetic is the two kernels' own, statement for
// statement (same thread -> element maps, same summation orders), so the gate is bit-identical to theirs; what goes away is one
// launch floor (~4.5 us) and the hidden units' round trip through memory: 7.3 + 5.1 -> ~8.5 us at bs 1.
__global__ __launch_bounds__(1024) void se_fused_small_kernel(const float* __restrict__ part, int tiles, float inv,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2t, const float* __restrict__ b2,
                                                             float* __restrict__ gate, int C, int R) {
  extern __shared__ float sm[];            // mean[C] | red[TG][C] | hs[128]
  float* mean = sm;
  float* red = sm + C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long b = blockIdx.y;
  const float* pb = part + b * tiles * (long)C;
  const int nq = C >> 2;
  const int TG = nq >= 1024 ? 1 : min(tiles, 1024 / nq);
  float* hs = red + (size_t)TG * C;
  for (int idx = tid; idx < TG * nq; idx += 1024) {
    const int q = idx % nq, tg = idx / nq;
    float4 a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = pb + 4 * q;
    for (int t0 = tg; t0 < tiles; t0 += 8 * TG) {
      float4 u[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) u[j] = ld4(src + (long)min(t0 + j * TG, tiles - 1) * C);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = t0 + j * TG < tiles;
        a[j].x += ok ? u[j].x : 0.f; a[j].y += ok ? u[j].y : 0.f; a[j].z += ok ? u[j].z : 0.f; a[j].w += ok ? u[j].w : 0.f;
      }
    }
    float4 r;
    r.x = ((a[0].x + a[1].x) + (a[2].x + a[3].x)) + ((a[4].x + a[5].x) + (a[6].x + a[7].x));
    r.y = ((a[0].y + a[1].y) + (a[2].y + a[3].y)) + ((a[4].y + a[5].y) + (a[6].y + a[7].y));
    r.z = ((a[0].z + a[1].z) + (a[2].z + a[3].z)) + ((a[4].z + a[5].z) + (a[6].z + a[7].z));
    r.w = ((a[0].w + a[1].w) + (a[2].w + a[3].w)) + ((a[4].w + a[5].w) + (a[6].w + a[7].w));
    *reinterpret_cast<float4*>(red + tg * C + 4 * q) = r;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 1024) {
    float s = red[c];
    for (int tg = 1; tg < TG; ++tg) s += red[tg * C + c];
    mean[c] = s * inv;
  }
  __syncthreads();
  for (int r = wave; r < R; r += 16) {                 // every hidden unit, one wavefront each (se_hidden_partials_kernel's dot)
    const float* wr = w1 + (long)r * C;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int q = lane;
    for (; q + 192 < nq; q += 256) {
      const float4 u0 = ld4(wr + 4 * q), u1 = ld4(wr + 4 * (q + 64)), u2 = ld4(wr + 4 * (q + 128)), u3 = ld4(wr + 4 * (q + 192));
      const float4 m0 = *reinterpret_cast<const float4*>(mean + 4 * q), m1 = *reinterpret_cast<const float4*>(mean + 4 * (q + 64));
      const float4 m2 = *reinterpret_cast<const float4*>(mean + 4 * (q + 128)), m3 = *reinterpret_cast<const float4*>(mean + 4 * (q + 192));
      s0 = fmaf(u0.x, m0.x, fmaf(u0.y, m0.y, fmaf(u0.z, m0.z, fmaf(u0.w, m0.w, s0))));
      s1 = fmaf(u1.x, m1.x, fmaf(u1.y, m1.y, fmaf(u1.z, m1.z, fmaf(u1.w, m1.w, s1))));
      s2 = fmaf(u2.x, m2.x, fmaf(u2.y, m2.y, fmaf(u2.z, m2.z, fmaf(u2.w, m2.w, s2))));
      s3 = fmaf(u3.x, m3.x, fmaf(u3.y, m3.y, fmaf(u3.z, m3.z, fmaf(u3.w, m3.w, s3))));
    }
    for (; q < nq; q += 64) {
      const float4 u0 = ld4(wr + 4 * q);
      const float4 m0 = *reinterpret_cast<const float4*>(mean + 4 * q);
      s0 = fmaf(u0.x, m0.x, fmaf(u0.y, m0.y, fmaf(u0.z, m0.z, fmaf(u0.w, m0.w, s0))));
    }
    const float s = wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) hs[r] = fast_silu(s + b1[r]);
  }
  __syncthreads();
  const int c = blockIdx.x * 256 + tid;
  if (tid >= 256 || c >= C) return;
  const float* w = w2t + c;
  float s0 = b2[c], s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = 0;
  for (; r + 15 < R; r += 16) {
    float u[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) u[j] = w[(long)(r + j) * C];
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
      s0 = fmaf(u[j], hs[r + j], s0);
      s1 = fmaf(u[j + 1], hs[r + j + 1], s1);
      s2 = fmaf(u[j + 2], hs[r + j + 2], s2);
      s3 = fmaf(u[j + 3], hs[r + j + 3], s3);
    }
  }
  for (; r + 3 < R; r += 4) {
    s0 = fmaf(w[(long)r * C], hs[r], s0);
    s1 = fmaf(w[(long)(r + 1) * C], hs[r + 1], s1);
    s2 = fmaf(w[(long)(r + 2) * C], hs[r + 2], s2);
    s3 = fmaf(w[(long)(r + 3) * C], hs[r + 3], s3);
  }
  for (; r < R; ++r) s0 = fmaf(w[(long)r * C], hs[r], s0);
  gate[b * C + c] = fast_sigmoid((s0 + s1) + (s2 + s3));
}

// The gate FOLDED INTO THE PROJECT WEIGHTS, per image: Wg[b][n][k] = W[n][k] * gate[b][k], split (hi = bf16, lo = bf16 of the
// rest) and packed in the B-operand order of the pointwise kernels ([jt][s][part][lane][8], hip_ops.SplitWeight) -- so that
// the project convolution's row operand is the depthwise output AS STORED (hl32, LDS-DMA) instead of rows re-gated and
// re-split by every workgroup that touches them.  Grid (ceil(K / 64), B): a workgroup forms the gate of its 64 input
// channels (same summation order as se_gate_hid_kernel) and scales / splits / packs their column block of W for all N
// rows; thread = (octet of 8 inputs, one of 32 consecutive output rows): a wavefront half writes 512 contiguous bytes.
__global__ __launch_bounds__(256) void se_gate_weights_kernel(const float* __restrict__ hid, const float* __restrict__ w2t,
                                                              const float* __restrict__ b2, const float* __restrict__ W,
                                                              __bf16* __restrict__ wpk, long img_elems, float* __restrict__ gate,
                                                              int C, int R, int N) {
  __shared__ float hs[256];
  __shared__ float gs[64];
  const long b = blockIdx.y;
  const int tid = threadIdx.x;
  const int k0 = blockIdx.x * 64;
  if (tid < R) hs[tid] = hid[b * R + tid];
  __syncthreads();
  if (tid < 64) {
    const int c = k0 + tid;
    float g = 0.f;                                   // inputs past C (K padded to 16): weight columns of zeros
    if (c < C) {
      const float* w = w2t + c;
      float s0 = b2[c], s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int r = 0;
      for (; r + 15 < R; r += 16) {                   // sixteen loads in flight, the summation order of se_gate_hid_kernel
        float u[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) u[j] = w[(long)(r + j) * C];
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
          s0 = fmaf(u[j], hs[r + j], s0);
          s1 = fmaf(u[j + 1], hs[r + j + 1], s1);
          s2 = fmaf(u[j + 2], hs[r + j + 2], s2);
          s3 = fmaf(u[j + 3], hs[r + j + 3], s3);
        }
      }
      for (; r + 3 < R; r += 4) {
        s0 = fmaf(w[(long)r * C], hs[r], s0);
        s1 = fmaf(w[(long)(r + 1) * C], hs[r + 1], s1);
        s2 = fmaf(w[(long)(r + 2) * C], hs[r + 2], s2);
        s3 = fmaf(w[(long)(r + 3) * C], hs[r + 3], s3);
      }
      for (; r < R; ++r) s0 = fmaf(w[(long)r * C], hs[r], s0);
      g = fast_sigmoid((s0 + s1) + (s2 + s3));
      if (gate != nullptr) gate[b * C + c] = g;
    }
    gs[tid] = g;
  }
  __syncthreads();
  const int o = tid >> 5, nl = tid & 31;             // octet of the 64-channel block, row inside a 32-row channel tile
  const int k = k0 + 8 * o;
  const int Kp = (C + 15) & ~15;
  if (k >= Kp) return;
  const int nsteps = Kp >> 4, s = k >> 4, hh = (k >> 3) & 1;
  float g8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) g8[e] = gs[8 * o + e];
  __bf16* ob = wpk + b * img_elems;
  const int ntl = (N + 31) >> 5;
  for (int jt = 0; jt < ntl; ++jt) {
    const int n = jt * 32 + nl;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0.f;
    if (n < N && k < C) {                            // C is a multiple of 8: an octet is inside or outside as a whole
      const float4 u = ld4(W + (long)n * C + k), v = ld4(W + (long)n * C + k + 4);
      f[0] = u.x * g8[0]; f[1] = u.y * g8[1]; f[2] = u.z * g8[2]; f[3] = u.w * g8[3];
      f[4] = v.x * g8[4]; f[5] = v.y * g8[5]; f[6] = v.z * g8[6]; f[7] = v.w * g8[7];
    }
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    bf16x8_t h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 hb = (__bf16)f[e];
      h[e] = hb;
      l[e] = (__bf16)(f[e] - (float)hb);
    }
    __bf16* d = ob + (((long)jt * nsteps + s) * 2) * 512 + (hh * 32 + nl) * 8;
    *reinterpret_cast<bf16x8_t*>(d) = h;
    *reinterpret_cast<bf16x8_t*>(d + 512) = l;
  }
}

template <int K, int S, int PX>
int launch_dws(const DWSArgs& a, const DWSGeom& g, int B, hipStream_t st) {
  hipLaunchKernelGGL((dw_slide_kernel<K, S, PX>), dim3((unsigned)((long)g.tiles * g.chunks * B)), dim3(256), 0, st, a);
  OCV_CHECK_LAUNCH("ocv_depthwise_conv_nhwc_sum_fwd");
  return 0;
}

}  // namespace

extern "C" int ocv_depthwise_sum_tiles(int B, int C, int Ho, int Wo, int k, int stride) {
  if (B < 1 || C < 4 || C % 4 != 0 || Ho < 1 || Wo < 1 || (k != 3 && k != 5) || (stride != 1 && stride != 2)) return 0;
  return dws_geom(B, C, Ho, Wo, k, stride).tiles;
}

extern "C" int ocv_depthwise_conv_nhwc_sum_fwd(const float* in, const float* w, const float* bias, float* out, float* part,
                                               int B, int C, int H, int W, int k, int stride, int pad_t, int pad_l,
                                               int Ho, int Wo, ocv_stream_t stream) {
  return ocv_depthwise_conv_nhwc_sum_hl_fwd(in, w, bias, out, nullptr, part, B, C, H, W, k, stride, pad_t, pad_l, Ho, Wo, stream);
}

namespace {
int dws_run(const float* in, const float* w, const float* bias, float* out, void* out_hl, float* part,
            int B, int C, int H, int W, int k, int stride, int pad_t, int pad_l, int Ho, int Wo, ocv_stream_t stream) {
  OCV_CHECK_ARG(in && w && (out || out_hl) && part, "ocv_depthwise_conv_nhwc_sum_fwd: null pointer");
  OCV_CHECK_ARG(out_hl == nullptr || (C % 32 == 0 && ocv_aligned16(out_hl)), "ocv_depthwise_conv_nhwc_sum_hl_fwd: the split output needs C to be a multiple of 32 (got %d) and 16-byte alignment", C);
  OCV_CHECK_ARG(B >= 1 && C >= 4 && C % 4 == 0 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1, "ocv_depthwise_conv_nhwc_sum_fwd: bad sizes (C must be a multiple of 4)");
  OCV_CHECK_ARG((k == 3 || k == 5) && (stride == 1 || stride == 2), "ocv_depthwise_conv_nhwc_sum_fwd: k must be 3 or 5 and stride 1 or 2");
  OCV_CHECK_ARG(pad_t >= 0 && pad_l >= 0 && pad_t < k && pad_l < k, "ocv_depthwise_conv_nhwc_sum_fwd: bad padding");
  OCV_CHECK_ARG((Ho - 1) * stride - pad_t < H && (Wo - 1) * stride - pad_l < W, "ocv_depthwise_conv_nhwc_sum_fwd: output larger than the padded input allows");
  OCV_CHECK_ARG(ocv_aligned16(in) && ocv_aligned16(w) && ocv_aligned16(out) && ocv_aligned16(bias) && ocv_aligned16(part), "ocv_depthwise_conv_nhwc_sum_fwd: operands must be 16-byte aligned");
  const DWSGeom g = dws_geom(B, C, Ho, Wo, k, stride);
  OCV_CHECK_ARG((long)g.tiles * g.chunks * B < (1L << 31), "ocv_depthwise_conv_nhwc_sum_fwd: too many workgroups");
  DWSArgs a{in, w, bias, out, part, (__bf16*)out_hl, C, H, W, Ho, Wo, pad_t, pad_l, g.QL, g.PL, g.RY, g.wox, g.nwork, g.tiles, g.chunks};
  hipStream_t st = (hipStream_t)stream;
  if (k == 3 && stride == 1) return launch_dws<3, 1, 4>(a, g, B, st);
  if (k == 3 && stride == 2) return launch_dws<3, 2, 2>(a, g, B, st);
  if (k == 5 && stride == 1) return launch_dws<5, 1, 2>(a, g, B, st);
  return launch_dws<5, 2, 1>(a, g, B, st);
}
}  // namespace

extern "C" int ocv_depthwise_conv_nhwc_sum_hl_fwd(const float* in, const float* w, const float* bias, float* out, void* out_hl,
                                                  float* part, int B, int C, int H, int W, int k, int stride, int pad_t,
                                                  int pad_l, int Ho, int Wo, ocv_stream_t stream) {
  return dws_run(in, w, bias, out, out_hl, part, B, C, H, W, k, stride, pad_t, pad_l, Ho, Wo, stream);
}

extern "C" int ocv_se_gate_partials_fwd(const float* part, int tiles, long pixels_per_image, const float* w1,
                                        const float* b1, const float* w2t, const float* b2, float* gate, float* hidden_ws,
                                        int B, int C, int R, ocv_stream_t stream) {
  OCV_CHECK_ARG(part && w1 && b1 && w2t && b2 && gate && hidden_ws, "ocv_se_gate_partials_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && B <= 65535 && C >= 1 && R >= 1 && R <= 256 && tiles >= 1 && pixels_per_image >= 1, "ocv_se_gate_partials_fwd: bad sizes (R <= 256)");
  OCV_CHECK_ARG(C <= 8192 && C % 4 == 0, "ocv_se_gate_partials_fwd: C must be a multiple of 4, at most 8192 (got %d)", C);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(C + (C >= 4096 ? C : 4096)) * sizeof(float);      // mean[C] | red[TG][C], TG * C <= 4096
  // (A/B, round 4: limits of 1824 / 76 and 3072 / 128 instead of the two below take stages 6 and 7 in -- measured SLOWER there, every workgroup of an
  //  image re-reading 0.55 / 1.5 MB of W1: bs 16 1000 -> 996 -> 994 img/s one at a time, bs 1 293 -> 286 -> 284)
  constexpr int max_c = 1536, max_r = 64;
  if (C <= max_c && R <= max_r && R <= 128) {                       // small squeeze-excite weights: ONE launch
    hipLaunchKernelGGL(se_fused_small_kernel, dim3((C + 255) / 256, B), dim3(1024), lds + 128 * sizeof(float), st, part, tiles,
                       1.0f / (float)pixels_per_image, w1, b1, w2t, b2, gate, C, R);
    OCV_CHECK_LAUNCH("ocv_se_gate_partials_fwd(fused)");
    return 0;
  }
  hipLaunchKernelGGL(se_hidden_partials_kernel, dim3((R + 15) / 16, B), dim3(1024), lds, st, part, tiles,
                     1.0f / (float)pixels_per_image, w1, b1, hidden_ws, C, R);
  OCV_CHECK_LAUNCH("ocv_se_gate_partials_fwd(hidden)");
  hipLaunchKernelGGL(se_gate_hid_kernel, dim3((C + 255) / 256, B), dim3(256), 0, st, (const float*)hidden_ws, w2t, b2, gate, C, R);
  OCV_CHECK_LAUNCH("ocv_se_gate_partials_fwd(gate)");
  return 0;
}

extern "C" int ocv_se_gate_weights_fwd(const float* part, int tiles, long pixels_per_image, const float* w1, const float* b1,
                                       const float* w2t, const float* b2, const float* W, void* w_packed, long w_image_elems,
                                       float* gate, float* hidden_ws, int B, int C, int R, int N, ocv_stream_t stream) {
  OCV_CHECK_ARG(part && w1 && b1 && w2t && b2 && W && w_packed && hidden_ws, "ocv_se_gate_weights_fwd: null pointer");
  OCV_CHECK_ARG(B >= 1 && B <= 65535 && C >= 8 && C % 8 == 0 && R >= 1 && R <= 256 && N >= 1 && tiles >= 1 && pixels_per_image >= 1,
                "ocv_se_gate_weights_fwd: bad sizes (C a multiple of 8, R <= 256)");
  OCV_CHECK_ARG(C <= 8192, "ocv_se_gate_weights_fwd: C too large (%d)", C);
  OCV_CHECK_ARG(w_image_elems >= (long)ocv_pointwise_packed_weight_elems(C, N) && (w_image_elems & 7) == 0 && ocv_aligned16(w_packed) && ocv_aligned16(W),
                "ocv_se_gate_weights_fwd: w_image_elems must hold one packed matrix (ocv_pointwise_packed_weight_elems) and keep 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(C + (C >= 4096 ? C : 4096)) * sizeof(float);      // mean[C] | red[TG][C], TG * C <= 4096
  hipLaunchKernelGGL(se_hidden_partials_kernel, dim3((R + 15) / 16, B), dim3(1024), lds, st, part, tiles,
                     1.0f / (float)pixels_per_image, w1, b1, hidden_ws, C, R);
  OCV_CHECK_LAUNCH("ocv_se_gate_weights_fwd(hidden)");
  const int Kp = (C + 15) / 16 * 16;
  hipLaunchKernelGGL(se_gate_weights_kernel, dim3((Kp + 63) / 64, B), dim3(256), 0, st, (const float*)hidden_ws, w2t, b2, W,
                     (__bf16*)w_packed, w_image_elems, gate, C, R, N);
  OCV_CHECK_LAUNCH("ocv_se_gate_weights_fwd(gate + weights)");
  return 0;
}
