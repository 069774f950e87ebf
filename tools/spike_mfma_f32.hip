// spike_mfma_f32.hip -- what a v_mfma_f32_32x32x2_f32 stream sustains on gfx950 (developer probe, not part of the library):
//   mode 0: MFMAs only, 4 independent accumulators, operands in registers
//   mode 1: as the conv kernel's inner loop -- B operands re-read from LDS (ds_read_b32, one group ahead), A operands from registers
//   mode 2: mode 1 + the A operands refreshed from L2 (one float4 per 4 k-steps) as the conv kernel does
// hipcc --offload-arch=gfx950 -O3 tools/spike_mfma_f32.hip -o /tmp/spike_mfma && /tmp/spike_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void k(float* out, const float4* __restrict__ w, int iters) {
  __shared__ float xs[32 * 257];
  const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
  for (int i = threadIdx.x; i < 32 * 257; i += 256) xs[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 a[4][2];
  for (int g = 0; g < 4; ++g)
    for (int i = 0; i < 2; ++i) a[g][i] = w[(g * 2 + i) * 64 + lane];
  const float* xrow = xs + lh * 257 + l31;
  float bq[2][4][2];
  for (int kk = 0; kk < 4; ++kk)
    for (int j = 0; j < 2; ++j) bq[0][kk][j] = xrow[(2 * kk) * 257 + j * 32];
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2)
      for (int g = 0; g < 4; ++g)
        for (int i = 0; i < 2; ++i) a[g][i] = w[((it & 63) * 8 + g * 2 + i) * 64 + lane];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      if (MODE >= 1) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int j = 0; j < 2; ++j) bq[(gg + 1) & 1][kk][j] = xrow[(((gg + 1) & 3) * 8 + 2 * kk) * 257 + j * 32 + (it & 7)];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float av = kk == 0 ? a[gg][i].x : kk == 1 ? a[gg][i].y : kk == 2 ? a[gg][i].z : a[gg][i].w;
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[MODE >= 1 ? (gg & 1) : 0][kk][j], acc[i][j], 0, 0, 0);
        }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// modes 3 / 4: the production loop shape -- chunks of `ntap` taps (runtime), A fragments of the next tap requested at the tap's
// entry into a_nxt and copied to a_cur at its end, B fragments of the next tap's first group carried; mode 4 adds the per-chunk
// x-tile copy by LDS-DMA into the other buffer + wait + barrier.
template <int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void k2(float* out, const float4* __restrict__ w, const float* __restrict__ xg, int nchunks, int ntap, int adil) {
  extern __shared__ float xs2[];  // 2 x [32][XWP]
  const int XW = 128 + (ntap - 1) * adil, XWP = XW | 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5, wn = wave & 1;
  for (int i = threadIdx.x; i < 2 * 32 * XWP; i += 256) xs2[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 a_cur[4][2], a_nxt[4][2];
  const float4* wq = w + lane;
  auto load_a = [&](float4 (&a)[4][2], int idx) {
    for (int g = 0; g < 4; ++g)
      for (int i = 0; i < 2; ++i) a[g][i] = wq[((idx & 63) * 8 + g * 2 + i) * 64];
  };
  auto load_b = [&](float (&bq)[4][2], const float* xrow, int gg) {
    for (int kk = 0; kk < 4; ++kk)
      for (int j = 0; j < 2; ++j) bq[kk][j] = xrow[(gg * 8 + 2 * kk) * XWP + j * 32];
  };
  float bcarry[4][2];
  auto issue_dma = [&](float* dst) {
    const int np = (XW + 63) >> 6;
    float* dst0 = dst + (wave * 8) * XWP;
    for (int pc = 0; pc < np; ++pc) {
      const int cs = (pc < np - 1) ? pc * 64 : XW - 64;
      for (int r = 0; r < 8; ++r)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xg + (size_t)(blockIdx.x % 64) * 4096 + r * 512 + cs + lane),
                                         (__attribute__((address_space(3))) void*)(dst0 + r * XWP + cs), 4, 0, 0);
    }
  };
  auto run_tap = [&](const float* Xc, int g, int tap) {
    load_a(a_nxt, g * ntap + tap + 1);
    const float* xrow = Xc + lh * XWP + wn * 64 + l31 + tap * adil;
    const float* xrow_n = Xc + lh * XWP + wn * 64 + l31 + (tap + 1 < ntap ? tap + 1 : tap) * adil;
    float bq[2][4][2];
    if (tap == 0) load_b(bq[0], xrow, 0);
    else
      for (int kk = 0; kk < 4; ++kk)
        for (int j = 0; j < 2; ++j) bq[0][kk][j] = bcarry[kk][j];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      if (gg + 1 < 4) load_b(bq[(gg + 1) & 1], xrow, gg + 1);
      else load_b(bcarry, xrow_n, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float av = kk == 0 ? a_cur[gg][i].x : kk == 1 ? a_cur[gg][i].y : kk == 2 ? a_cur[gg][i].z : a_cur[gg][i].w;
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[gg & 1][kk][j], acc[i][j], 0, 0, 0);
        }
    }
    for (int gg = 0; gg < 4; ++gg)
      for (int i = 0; i < 2; ++i) a_cur[gg][i] = a_nxt[gg][i];
  };
  const int bufsz = 32 * XWP;
  if (MODE == 4) issue_dma(xs2);
  load_a(a_cur, 0);
  auto chunk_step = [&](int g, const float* __restrict__ cur, float* __restrict__ nxt) {
    if (MODE == 4) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    run_tap(cur, g, 0);
    if (MODE == 4) issue_dma(nxt);
    for (int tap = 1; tap < ntap; ++tap) run_tap(cur, g, tap);
  };
  for (int g = 0; g < nchunks; ++g) chunk_step(g, xs2 + (g & 1) * bufsz, xs2 + ((g + 1) & 1) * bufsz);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int OCC>
void run2(const char* name, float* out, float4* w, float* xg, int ntap, int adil) {
  const int nchunks = 24, grid = 256 * OCC * 4;
  const int XWP = (128 + (ntap - 1) * adil) | 1;
  const size_t smem = (size_t)2 * 32 * XWP * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k2<MODE, OCC>), dim3(grid), dim3(256), smem, 0, out, w, xg, 2, ntap, adil);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k2<MODE, OCC>), dim3(grid), dim3(256), smem, 0, out, w, xg, nchunks, ntap, adil);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * 4 * nchunks * ntap * 64 * 4096.0;
  printf("%-44s k=%2d d=%d %8.3f ms  %7.1f TFLOP/s\n", name, ntap, adil, ms, fl / ms / 1e9);
}

template <int MODE, int OCC>
void run(const char* name, float* out, float4* w) {
  const int iters = 2000, grid = 256 * OCC * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, OCC>), dim3(grid), dim3(256), 0, 0, out, w, 100);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE, OCC>), dim3(grid), dim3(256), 0, 0, out, w, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * 4 * iters * 64 * 4096.0;
  printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
}

int main() {
  float* out;
  float4* w;
  hipMalloc(&out, 256 * 4096 * 4 * sizeof(float));
  hipMalloc(&w, 64 * 8 * 64 * sizeof(float4));
  hipMemset(w, 0, 64 * 8 * 64 * sizeof(float4));
  run<0, 1>("mfma only, 1 wave/SIMD", out, w);
  run<0, 2>("mfma only, 2 waves/SIMD", out, w);
  run<1, 1>("+ B from LDS, 1 wave/SIMD", out, w);
  run<1, 2>("+ B from LDS, 2 waves/SIMD", out, w);
  run<2, 2>("+ B from LDS + A from L2, 2 waves/SIMD", out, w);
  float* xg;
  hipMalloc(&xg, 64 * 4096 * sizeof(float) + 4096);
  hipMemset(xg, 0, 64 * 4096 * sizeof(float) + 4096);
  for (int ntap : {3, 11}) {
    run2<3, 2>("production loop shape, no staging, occ 2", out, w, xg, ntap, 1);
    run2<4, 2>("production loop shape + DMA + barrier, occ 2", out, w, xg, ntap, 1);
    run2<4, 2>("   same, dilation 5", out, w, xg, ntap, 5);
  }
  return 0;
}
