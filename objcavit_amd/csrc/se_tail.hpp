// The squeeze-excite gate of an MBConv block computed INSIDE the launch that produces the pooled tensor (depthwise /
// fused expand + depthwise), by the LAST workgroup of each image to finish -- instead of two latency-bound launches behind it
// (se_hidden_partials_kernel + se_gate_hid_kernel: 72 of the 271 launches of a forward, 0.48 ms at bs 16, 0.61 of 4.05 ms at bs 1;
// profiles/r04a_*).  Reference arithmetic: SqueezeExcite of the hub backbone's blocks (x.mean((2, 3)) -> conv_reduce -> SiLU ->
// conv_expand -> sigmoid), reached through modules/DenseFeatureExtractor.py:18-27,149.
//
// In-launch hand-off between workgroups, counter form (cdna_hip_programming.md section 6 Guideline 16 / section 5 "In-launch
// split-K reduction"; MI355X_MICROARCH.md "inter-workgroup visibility"): per-XCD L2s are not coherent and a CU's L1 is never
// refreshed by other CUs' stores, so
//   producer (every workgroup): its pooling partial goes out WRITE-THROUGH (sc1: agent-scope atomic stores of <= 8 bytes), the
//     storing wave drains them (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane draws a ticket with an agent-scope atomic add;
//   consumer (the workgroup whose ticket is the last of its image): that lane issues ONE agent-scope acquire (invalidates this
//     CU's L1) + s_waitcnt vmcnt(0), workgroup barrier, then every wave reads the partials with ordinary vector loads.
// No workgroup ever WAITS for another (no spin, no residency assumption): whoever arrives last does the tail.  The counter is
// zero when the launch starts (zero-initialised once by the caller) and the last arriver puts it back to zero, so a captured
// graph replays without a memset node.  Sums run in a fixed order that depends on the launch geometry only: bit-reproducible.
#pragma once
#include "common.hpp"

typedef __attribute__((address_space(1))) unsigned se_gu32;
typedef __attribute__((address_space(1))) unsigned long long se_gu64;

constexpr int SE_TAIL_MAX_C = 1536;          // widest pooled tensor the tail takes (LDS: mean[C] + group sums + hidden units)
constexpr int SE_TAIL_MAX_R = 128;
constexpr int SE_TAIL_RED = 4096;             // floats of row-group sums ([TG][C])
constexpr int SE_TAIL_LDS_FLOATS = SE_TAIL_MAX_C + SE_TAIL_RED + SE_TAIL_MAX_R + 4;

struct SETail {
  const float *w1, *b1, *w2t, *b2;   // conv_reduce [R][C], [R]; conv_expand TRANSPOSED [R][C], [C]
  float* gate;                       // [B][C] out
  unsigned* cnt;                     // [B] arrival tickets: zero at launch, zero again when the launch has finished
  int R, total;                      // hidden units; workgroups per image
  float inv;                         // 1 / pixels per image
};

__device__ __forceinline__ void se_store_sc1(float* p, float v) {
  __hip_atomic_store((se_gu32*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void se_store_sc1(float* p, float4 v) {      // p 16-byte aligned
  typedef unsigned long long u64;
  __hip_atomic_store((se_gu64*)p, ((u64)__float_as_uint(v.y) << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((se_gu64*)p + 1, ((u64)__float_as_uint(v.w) << 32) | __float_as_uint(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by ALL 256 threads of the workgroup, after the waves that stored pooling partials have drained them
// (asm volatile("s_waitcnt vmcnt(0)" ::: "memory") behind the stores).  Returns true in the image's last workgroup, whose L1
// has then been invalidated: its loads of the partials see every other workgroup's stores.
__device__ __forceinline__ bool se_arrive(const SETail& s, long b, float* sm) {
  __syncthreads();
  unsigned* flag = reinterpret_cast<unsigned*>(sm + SE_TAIL_LDS_FLOATS - 1);
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add((se_gu32*)(s.cnt + b), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = old == (unsigned)(s.total - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store((se_gu32*)(s.cnt + b), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // nobody else touches it any more
    }
    *flag = last ? 1u : 0u;
  }
  __syncthreads();
  return *flag != 0u;
}

// mean -> hidden (SiLU) -> gate (sigmoid) of image b from its pooling partials part_img [rows][C]; 256 threads.
// ONE workgroup does what two wide launches did, so everything here is shaped for memory-level parallelism: a first version with
// four loads in flight per thread took 13 - 62 us per block (every round trip ~1 us: the partials were stored write-through and
// come from memory, the weights from L2); here every thread keeps up to 32 independent 16-byte loads in flight (the depthwise
// kernels around it own 195 - 256 registers anyway), three to nine round trips in all.  Out-of-range loads are CLAMPED to a valid
// address and their values selected away (a branch around a load makes the compiler wait for every load on its own).
__device__ __forceinline__ float4 se_sel(const float4 u, const bool ok) {
  return make_float4(ok ? u.x : 0.f, ok ? u.y : 0.f, ok ? u.z : 0.f, ok ? u.w : 0.f);
}
__device__ __forceinline__ void se_add(float4& a, const float4 u) { a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
__device__ __forceinline__ float se_dot(const float4 a, const float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }

__device__ __forceinline__ void se_gate_from_partials(const SETail& s, const float* __restrict__ part_img, int rows, int C, long b,
                                                      float* sm) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* mean = sm;                             // [C]
  float* red = sm + SE_TAIL_MAX_C;              // [TG][C], TG * C <= SE_TAIL_RED
  float* hid = red + SE_TAIL_RED;               // [R]
  const int nq = C >> 2;
  // --- pooled mean: item = (channel quad, row group); up to 4 items per thread, 8 rows of each in flight -----------------------
  const int TG = min(rows, max(1, SE_TAIL_RED / C));   // row groups per quad: <= 1024 items (their sums meet in LDS, in group order)
  const int items = TG * nq;
  {
    int q[4], tg[4];
    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = min(tid + 256 * k, items - 1);
      q[k] = idx % nq; tg[k] = idx / nq;
      acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int nit = (rows + TG - 1) / TG;                        // rows of the longest group
    for (int i0 = 0; i0 < nit; i0 += 8) {
      float4 u[4][8];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int t = tg[k] + (i0 + j) * TG;
          u[k][j] = ld4(part_img + (long)min(t, rows - 1) * C + 4 * q[k]);
        }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) se_add(acc[k], se_sel(u[k][j], tg[k] + (i0 + j) * TG < rows));     // fixed order: rows ascending
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (tid + 256 * k < items) *reinterpret_cast<float4*>(red + tg[k] * C + 4 * q[k]) = acc[k];
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float v = red[c];
    for (int g = 1; g < TG; ++g) v += red[g * C + c];
    mean[c] = v * s.inv;
  }
  __syncthreads();
  // --- hidden units: wavefront w owns units w, w + 4, ...; four units x all column blocks (<= 6 of 64 quads) in flight --------
  {
    const int M = (nq + 63) >> 6;
    float4 mv[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) mv[m] = se_sel(*reinterpret_cast<const float4*>(mean + 4 * min(lane + 64 * m, nq - 1)), m < M && lane + 64 * m < nq);
    for (int r0 = wave; r0 < s.R; r0 += 16) {
      float4 u[4][6];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* wr = s.w1 + (long)min(r0 + 4 * k, s.R - 1) * C;
#pragma unroll
        for (int m = 0; m < 6; ++m) u[k][m] = ld4(wr + 4 * min(lane + 64 * m, nq - 1));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float d = 0.f;
#pragma unroll
        for (int m = 0; m < 6; ++m) d += se_dot(u[k][m], mv[m]);              // masked columns meet zeros of mv
        d = wave_sum(d);
        const int r = r0 + 4 * k;
        if (lane == 0 && r < s.R) hid[r] = fast_silu(d + s.b1[r]);
      }
    }
  }
  __syncthreads();
  // --- gate: thread = channel quad (two per thread beyond 1024 channels); 16 hidden units of each in flight --------------------
  {
    const int q0 = min(tid, nq - 1), q1 = min(tid + 256, nq - 1);
    const bool two = nq > 256;
    float4 g0 = ld4(s.b2 + 4 * q0), g1 = ld4(s.b2 + 4 * q1);
    for (int r0 = 0; r0 < s.R; r0 += 16) {
      float4 u[2][16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float* wr = s.w2t + (long)min(r0 + j, s.R - 1) * C;
        u[0][j] = ld4(wr + 4 * q0);
        u[1][j] = two ? ld4(wr + 4 * q1) : make_float4(0.f, 0.f, 0.f, 0.f);    // (uniform: one branch for the workgroup)
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float h = r0 + j < s.R ? hid[min(r0 + j, s.R - 1)] : 0.f;
        g0.x = fmaf(u[0][j].x, h, g0.x); g0.y = fmaf(u[0][j].y, h, g0.y); g0.z = fmaf(u[0][j].z, h, g0.z); g0.w = fmaf(u[0][j].w, h, g0.w);
        g1.x = fmaf(u[1][j].x, h, g1.x); g1.y = fmaf(u[1][j].y, h, g1.y); g1.z = fmaf(u[1][j].z, h, g1.z); g1.w = fmaf(u[1][j].w, h, g1.w);
      }
    }
    float* go = s.gate + b * C;
    if (tid < nq)
      *reinterpret_cast<float4*>(go + 4 * tid) = make_float4(fast_sigmoid(g0.x), fast_sigmoid(g0.y), fast_sigmoid(g0.z), fast_sigmoid(g0.w));
    if (two && tid + 256 < nq)
      *reinterpret_cast<float4*>(go + 4 * (tid + 256)) = make_float4(fast_sigmoid(g1.x), fast_sigmoid(g1.y), fast_sigmoid(g1.z), fast_sigmoid(g1.w));
  }
}
