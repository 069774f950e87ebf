// spike_mfma_bf16x6.hip -- developer probe: fp32-accurate products as six bf16 MFMAs (x = h + m + l in bf16, products hh hm mh mm hl lh,
// fp32 accumulate) in the conv kernel's loop shape.  Per 16-channel group and wave (2 x 2 tiles of 32 x 32): A = 3 planes x 2 row
// tiles x 16 B from L2, B = 3 planes x 2 column tiles x ds_read_b128 from a channel-octet-interleaved LDS tile, 24 MFMAs
// (v_mfma_f32_32x32x16_bf16).  Reports fp32-EQUIVALENT TFLOP/s (2 * M * N * K, not the 6x bf16 work).
//   mode 0: MFMAs only; 1: + B from LDS; 2: + A from L2; 3: + per-chunk 16-byte LDS-DMA of the next x tile + barrier;
//   4: as 3 but A arrives as fp32 (32 B per fragment instead of 48) and is split into its three planes in registers
// hipcc --offload-arch=gfx950 -O3 tools/spike_mfma_bf16x6.hip -o /tmp/spike6 && /tmp/spike6
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int OCC, int MT = 2>
__global__ __launch_bounds__(256, OCC) void k6(float* out, const uint4* __restrict__ w, const uint4* __restrict__ xg, int nchunks, int ntap, int adil) {
  extern __shared__ uint4 xs[];  // 2 buffers x [plane 3][group 2][kh 2][XW] 16-byte units (8 channels of one column)
  const int XW = 128 + (ntap - 1) * adil;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5, wn = wave & 1;
  const int bufsz = 3 * 2 * 2 * XW;
  for (int i = threadIdx.x; i < 2 * bufsz; i += 256) xs[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  uint4 a_cur[2][3][2], a_nxt[2][3][2];  // [group][plane][row tile]
  const uint4* wq = w + lane;
  auto split8 = [](const uint4& x0, const uint4& x1, uint4& ph, uint4& pm, uint4& pl) {
    const unsigned int v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    unsigned int h[8], m[8], l[8];
    for (int e = 0; e < 8; ++e) {
      h[e] = v[e] + 0x8000u;
      const float r = __uint_as_float(v[e]) - __uint_as_float(h[e] & 0xffff0000u);
      m[e] = __float_as_uint(r) + 0x8000u;
      l[e] = __float_as_uint(r - __uint_as_float(m[e] & 0xffff0000u));
    }
    auto pk = [](unsigned int hi, unsigned int lo) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); };
    ph = make_uint4(pk(h[1], h[0]), pk(h[3], h[2]), pk(h[5], h[4]), pk(h[7], h[6]));
    pm = make_uint4(pk(m[1], m[0]), pk(m[3], m[2]), pk(m[5], m[4]), pk(m[7], m[6]));
    pl = make_uint4(pk(l[1], l[0]), pk(l[3], l[2]), pk(l[5], l[4]), pk(l[7], l[6]));
  };
  auto load_a = [&](uint4 (&a)[2][3][2], int idx) {
    for (int g = 0; g < 2; ++g)
      for (int i = 0; i < 2; ++i) {
        if (MODE == 4) {  // planes 0, 1 hold the raw fp32 halves until split_a() turns them into the three planes
          a[g][0][i] = wq[((idx & 31) * 12 + (g * 3 + 0) * 2 + i) * 64];
          a[g][1][i] = wq[((idx & 31) * 12 + (g * 3 + 1) * 2 + i) * 64];
        } else {
          for (int p = 0; p < 3; ++p) a[g][p][i] = wq[((idx & 31) * 12 + (g * 3 + p) * 2 + i) * 64];
        }
      }
  };
  auto split_a = [&](uint4 (&a)[2][3][2]) {
    for (int g = 0; g < 2; ++g)
      for (int i = 0; i < 2; ++i) {
        const uint4 x0 = a[g][0][i], x1 = a[g][1][i];
        split8(x0, x1, a[g][0][i], a[g][1][i], a[g][2][i]);
      }
  };
  auto load_b = [&](uint4 (&bq)[3][2], const uint4* xrow, int g) {
    for (int p = 0; p < 3; ++p)
      for (int j = 0; j < 2; ++j) bq[p][j] = xrow[((p * 2 + g) * 2) * XW + j * 32];
  };
  auto issue_dma = [&](uint4* dst) {  // 12 rows of XW units, 3 per wave, 64 units per instruction
    const int np = (XW + 63) >> 6;
    for (int r = 0; r < 3; ++r) {
      uint4* drow = dst + (wave * 3 + r) * XW;
      for (int pc = 0; pc < np; ++pc) {
        const int cs = (pc < np - 1) ? pc * 64 : XW - 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xg + (size_t)(blockIdx.x % 64) * 4096 + r * 512 + cs + lane),
                                         (__attribute__((address_space(3))) void*)(drow + cs), 16, 0, 0);
      }
    }
  };
  uint4 bcarry[3][2];
  auto run_tap = [&](const uint4* Xc, int gch, int tap) {
    if (MODE >= 2) load_a(a_nxt, gch * ntap + tap + 1);
    if (MODE == 4) split_a(a_cur);
    const uint4* xrow = Xc + lh * XW + wn * 64 + l31 + tap * adil;
    const uint4* xrow_n = Xc + lh * XW + wn * 64 + l31 + (tap + 1 < ntap ? tap + 1 : tap) * adil;
    uint4 bq[2][3][2];
    if (tap == 0) load_b(bq[0], xrow, 0);
    else
      for (int p = 0; p < 3; ++p)
        for (int j = 0; j < 2; ++j) bq[0][p][j] = bcarry[p][j];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (MODE >= 1) {
        if (g + 1 < 2) load_b(bq[(g + 1) & 1], xrow, g + 1);
        else load_b(bcarry, xrow_n, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int bb = MODE >= 1 ? (g & 1) : 0;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // smallest terms first: l*h, h*l, m*m, m*h, h*m, h*h
          const bf16x8 ah = __builtin_bit_cast(bf16x8, a_cur[g][0][i]), am = __builtin_bit_cast(bf16x8, a_cur[g][1][i]), al = __builtin_bit_cast(bf16x8, a_cur[g][2][i]);
          const bf16x8 bh = __builtin_bit_cast(bf16x8, bq[bb][0][j]), bm = __builtin_bit_cast(bf16x8, bq[bb][1][j]), bl = __builtin_bit_cast(bf16x8, bq[bb][2][j]);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
        }
    }
    if (MODE >= 2)
      for (int g = 0; g < 2; ++g)
        for (int p = 0; p < 3; ++p)
          for (int i = 0; i < 2; ++i) a_cur[g][p][i] = a_nxt[g][p][i];
  };
  if (MODE >= 3) issue_dma(xs);
  load_a(a_cur, 0);
  auto chunk_step = [&](int g, const uint4* __restrict__ cur, uint4* __restrict__ nxt) {
    if (MODE >= 3) {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    run_tap(cur, g, 0);
    if (MODE >= 3) issue_dma(nxt);
    for (int tap = 1; tap < ntap; ++tap) run_tap(cur, g, tap);
  };
  for (int g = 0; g < nchunks; ++g) chunk_step(g, xs + (g & 1) * bufsz, xs + ((g + 1) & 1) * bufsz);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int OCC, int MT = 2>
void run(const char* name, float* out, uint4* w, uint4* xg, int ntap, int adil) {
  const int nchunks = 24, grid = 256 * OCC * 4;
  const int XW = 128 + (ntap - 1) * adil;
  const size_t smem = (size_t)2 * 12 * XW * 16;
  hipFuncSetAttribute((const void*)k6<MODE, OCC, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k6<MODE, OCC, MT>), dim3(grid), dim3(256), smem, 0, out, w, xg, 2, ntap, adil);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k6<MODE, OCC, MT>), dim3(grid), dim3(256), smem, 0, out, w, xg, nchunks, ntap, adil);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per wave and (chunk, tap): 2 groups x 4 tiles x (32 x 32 x 16) MACs, fp32-equivalent
  const double fl = (double)grid * 4 * nchunks * ntap * 2 * (2 * MT) * 2.0 * 32 * 32 * 16;
  printf("%-52s k=%2d d=%d smem %3zu KB %8.3f ms  %7.1f TFLOP/s fp32-equivalent (%6.0f bf16)\n", name, ntap, adil, smem >> 10, ms, fl / ms / 1e9, 6 * fl / ms / 1e9);
}

int main() {
  float* out;
  uint4 *w, *xg;
  hipMalloc(&out, 256 * 4096 * 4 * sizeof(float));
  hipMalloc(&w, 32 * 12 * 64 * sizeof(uint4) + 4096);
  hipMemset(w, 0, 32 * 12 * 64 * sizeof(uint4) + 4096);
  hipMalloc(&xg, 64 * 4096 * sizeof(uint4) + 65536);
  hipMemset(xg, 0, 64 * 4096 * sizeof(uint4) + 65536);
  for (int ntap : {3, 11}) {
    run<0, 2>("six bf16 MFMAs per tile, operands in registers, occ 2", out, w, xg, ntap, 1);
    run<1, 2>("+ B (3 planes) from LDS, occ 2", out, w, xg, ntap, 1);
    run<2, 2>("+ A (3 planes) from L2, occ 2", out, w, xg, ntap, 1);
    run<3, 2>("+ 16-byte LDS-DMA of the next x tile + barrier, occ 2", out, w, xg, ntap, 1);
    run<3, 2>("   same, dilation 5", out, w, xg, ntap, 5);
    run<4, 2>("   A as fp32 from L2 (32 B per fragment), split in registers", out, w, xg, ntap, 1);
    run<0, 1>("2 x 2 tiles, operands in registers, ONE wave per SIMD", out, w, xg, ntap, 1);
    run<1, 1>("   + B from LDS, one wave per SIMD", out, w, xg, ntap, 1);
    run<0, 1, 1>("1 x 2 tiles, operands in registers, one wave per SIMD", out, w, xg, ntap, 1);
    run<1, 1, 1>("   + B from LDS, one wave per SIMD", out, w, xg, ntap, 1);
    run<0, 2, 1>("1 x 2 tiles per wave: operands in registers", out, w, xg, ntap, 1);
    run<1, 2, 1>("1 x 2 tiles per wave: + B from LDS (6 reads per 12 MFMAs)", out, w, xg, ntap, 1);
    run<1, 3, 1>("   same, 3 waves per SIMD", out, w, xg, ntap, 1);
  }
  return 0;
}
