// attn_full_x3.hip -- the s2mel DiT attention (attn_full.hip's operator, row N1) on the bf16 matrix cores, fp32-accurate.
//
// Reference op: F.scaled_dot_product_attention(q, k, v) in indextts/s2mel/modules/gpt_fast/model.py:303, run in fp32 by the
// reference (infer_v2.py:710-711).
//
// Both products of flash attention take their operands as three bf16 pieces per fp32 value (x = h + m + l exactly; see
// conv1d_x3.hip for the argument and the error bound) and six v_mfma_f32_32x32x16_bf16 partial products per 16-deep step with
// fp32 accumulation: 96 such MFMAs (3 072 matrix-pipe cycles) per 64-key tile and wave where the fp32-MFMA kernel issues 128 of
// 64 cycles (8 192).  Same flash structure as attn_full.hip, and the same trick that avoids any transpose between the two
// products:
//   S^T = K Q^T   rows = keys, cols = queries; accumulator register r of lane l = score of query (l & 31) against key
//                 (r&3) + 8(r>>2) + 4(l>>5) of the 32-key half.
//   O^T += V^T P^T  contracts over keys; the contraction order inside a 16-deep step is free, so step (j, u) takes for lane
//                 half kh the eight keys whose weights that lane already holds in registers 8u .. 8u+7 of half j -- P^T never
//                 moves; V^T is stored in exactly that key order by the split pass.
// A split pass per call writes K and V as planes in tile-major blocks ([tile][plane][8 units][64] 16-byte units: 24 KB each), so
// the attention kernel stages a tile as two contiguous blocks with no ALU work; K(t+1) is fetched under the PV MFMAs of tile t,
// V(t) under the S^T MFMAs and the softmax of tile t.  Softmax weights are split in registers (6 ALU ops).
#include "attn_full.h"

namespace ixtts {

typedef float ax_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ax_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int AX_TILE_UNITS = 3 * 8 * 64;  // one K (or V) tile: 3 planes x 8 units x 64 = 1536 16-byte units
constexpr int AX_WAVES = 4;

size_t attn_full_x3_plane_bytes(int B, int H, int T) {
  const size_t tiles = (size_t)(T + AF_KT - 1) / AF_KT;
  return (size_t)B * H * tiles * 2 * AX_TILE_UNITS * 16;
}


// ---- split pass: one workgroup per (64-key tile, batch*head).
//   K block  [plane][oct 8][key 64]: unit = d 8*oct .. 8*oct+7 of one key (the A operand of S^T for lane (key, kh = oct & 1))
//   V block  [plane][unit 8 = (j, u, kh)][d 64]: element e = key 32j + 16u + 8(e>>2) + 4kh + (e&3) of one d
// Keys at or beyond T are written as zeros (their scores are masked in the attention kernel).
__global__ __launch_bounds__(256) void attn_kv_planes_kernel(AttnFullArgs a, uint4* __restrict__ planes) {
  const int tile = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int n_tiles = (a.T + AF_KT - 1) / AF_KT;
  const float* kb = a.k + b * a.sb + h * a.sh;
  const float* vb = a.v + b * a.sb + h * a.sh;
  uint4* kdst = planes + ((size_t)bh * n_tiles + tile) * 2 * AX_TILE_UNITS;
  uint4* vdst = kdst + AX_TILE_UNITS;
  const int t0 = tile * AF_KT;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int uid = threadIdx.x + 256 * i;  // (key, oct): consecutive threads read consecutive 32-byte pieces of a key's row
    const int key = uid >> 3, oct = uid & 7;
    const int t = t0 + key;
    float v[8];
    const float4* src = reinterpret_cast<const float4*>(kb + (long)min(t, a.T - 1) * a.st + 8 * oct);
    const float4 x0 = src[0], x1 = src[1];
    const bool live = t < a.T;
    v[0] = live ? x0.x : 0.f; v[1] = live ? x0.y : 0.f; v[2] = live ? x0.z : 0.f; v[3] = live ? x0.w : 0.f;
    v[4] = live ? x1.x : 0.f; v[5] = live ? x1.y : 0.f; v[6] = live ? x1.z : 0.f; v[7] = live ? x1.w : 0.f;
    uint4 ph, pm, pl;
    split8_bf16x3(v, ph, pm, pl);
    kdst[(0 * 8 + oct) * 64 + key] = ph;
    kdst[(1 * 8 + oct) * 64 + key] = pm;
    kdst[(2 * 8 + oct) * 64 + key] = pl;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int d = threadIdx.x & 63, un = (threadIdx.x >> 6) + 4 * i;  // unit un = 4j + 2u + kh
    const int j = un >> 2, u = (un >> 1) & 1, kh = un & 1;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t = t0 + 32 * j + 16 * u + 8 * (e >> 2) + 4 * kh + (e & 3);
      const float x = vb[(long)min(t, a.T - 1) * a.st + d];
      v[e] = t < a.T ? x : 0.f;
    }
    uint4 ph, pm, pl;
    split8_bf16x3(v, ph, pm, pl);
    vdst[(0 * 8 + un) * 64 + d] = ph;
    vdst[(1 * 8 + un) * 64 + d] = pm;
    vdst[(2 * 8 + un) * 64 + d] = pl;
  }
}

template <int KSPLIT>
__global__ __launch_bounds__(AX_WAVES * 64, 3) void attn_full_x3_kernel(AttnFullArgs a, const uint4* __restrict__ planes) {
  __shared__ uint4 Ks[AX_TILE_UNITS];  // [plane][oct][key]
  __shared__ uint4 Vs[AX_TILE_UNITS];  // [plane][unit][d]
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int q0 = blockIdx.x * (AX_WAVES * AF_QW) + wave * AF_QW;
  const float* qb = a.q + b * a.sb + h * a.sh;

  // Q^T as the B operand of S^T = K Q^T: for step s the lane holds d = 16s + 8lh .. +7 of query l31, pre-scaled into the log2
  // domain (softmax weights are one v_exp_f32 each), split once
  uint4 qp[4][3];
  {
    const int qi = min(q0 + l31, a.T - 1);
    const float* qrow = qb + (long)qi * a.st + 8 * lh;
    const float qs = a.scale * 1.4426950408889634f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 x0 = *reinterpret_cast<const float4*>(qrow + 16 * s), x1 = *reinterpret_cast<const float4*>(qrow + 16 * s + 4);
      const float v[8] = {x0.x * qs, x0.y * qs, x0.z * qs, x0.w * qs, x1.x * qs, x1.y * qs, x1.z * qs, x1.w * qs};
      split8_bf16x3(v, qp[s][0], qp[s][1], qp[s][2]);
    }
  }
  ax_f32x16 ot[2];  // O^T tiles: d 0..31, 32..63 (rows d in registers, column = this lane's query)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[j][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int n_tiles = (a.T + AF_KT - 1) / AF_KT;
  const int tiles_per = (n_tiles + KSPLIT - 1) / KSPLIT;
  const int tile_lo = KSPLIT > 1 ? blockIdx.z * tiles_per : 0;
  const int tile_hi = KSPLIT > 1 ? min(n_tiles, tile_lo + tiles_per) : n_tiles;
  const uint4* pb = planes + (size_t)bh * n_tiles * 2 * AX_TILE_UNITS;
  // a tile block is 1536 contiguous units: six 16-byte LDS-DMA copies per thread
  auto issue_tile = [&](const uint4* __restrict__ src, uint4* __restrict__ dst) {
#pragma unroll
    for (int i = 0; i < AX_TILE_UNITS / (AX_WAVES * 64); ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * (AX_WAVES * 64) + threadIdx.x),
                                       (__attribute__((address_space(3))) void*)(dst + i * (AX_WAVES * 64) + wave * 64), 16, 0, 0);
  };
  // six partial products into each of two accumulators, alternating between them (smallest products first)
  auto mfma6x2 = [](ax_f32x16& c0, ax_f32x16& c1, const uint4 (&A0)[3], const uint4 (&A1)[3], const uint4 (&Bq)[3]) {
    const ax_bf16x8 a0h = __builtin_bit_cast(ax_bf16x8, A0[0]), a0m = __builtin_bit_cast(ax_bf16x8, A0[1]), a0l = __builtin_bit_cast(ax_bf16x8, A0[2]);
    const ax_bf16x8 a1h = __builtin_bit_cast(ax_bf16x8, A1[0]), a1m = __builtin_bit_cast(ax_bf16x8, A1[1]), a1l = __builtin_bit_cast(ax_bf16x8, A1[2]);
    const ax_bf16x8 bh_ = __builtin_bit_cast(ax_bf16x8, Bq[0]), bm = __builtin_bit_cast(ax_bf16x8, Bq[1]), bl = __builtin_bit_cast(ax_bf16x8, Bq[2]);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, bh_, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, bh_, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bl, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bl, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0m, bm, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1m, bm, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0m, bh_, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1m, bh_, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bm, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bm, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bh_, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bh_, c1, 0, 0, 0);
  };
  // ---- the tile loop: two barriers per 64-key tile; V(t) is copied under the S^T MFMAs and the softmax of tile t, K(t+1) under
  // the PV MFMAs.  Three workgroups per CU: the A fragments of a step are read right before its MFMAs (no second register set
  // for a step-ahead prefetch: 165 VGPRs instead of 205) and the third resident wave covers their LDS latency: 157 us against
  // 172 with two.  Tried and measured on (B=2, H=8, T=2322), whole call, at two per CU: this form 172 us; K double-buffered a
  // tile ahead 172; tile copies through registers + ds_write 236; 8-wave workgroups (256 queries per staged tile) 194; 8 waves as
  // two anti-phase halves (four barriers per tile, one half in an MFMA segment while the other does its softmax) 196 -- a 48-MFMA
  // segment is short against a barrier (0.15-0.3 us each) and the clock sits near 1.9 GHz under this load; work ids grouped per
  // XCD 180.  In-kernel timestamps of this form: S^T 1.4, softmax 1.4, PV + split 1.7, the two waits 0.35 us each per tile.
  // (fp32-MFMA kernel: 216-243 us.)
  if (tile_lo < tile_hi) issue_tile(pb + (size_t)tile_lo * 2 * AX_TILE_UNITS, Ks);
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int t0 = tile * AF_KT;
    // K(tile) has landed (the only copies in flight), for everybody; every wave has left the PV phase of the previous tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_tile(pb + ((size_t)tile * 2 + 1) * AX_TILE_UNITS, Vs);
    // ---- S^T = K Q^T for the two 32-key halves: A = K units (plane, oct 2s + lh, key)
    ax_f32x16 st[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[j][r] = 0.f;
    const uint4* kp = Ks + lh * 64 + l31;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 ka[2][3];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ka[j][pl] = kp[(pl * 8 + 2 * s) * 64 + 32 * j];
      mfma6x2(st[0], st[1], ka[0], ka[1], qp[s]);
    }
    // ---- online softmax for this lane's query; keys beyond T are masked (only the last tile has any)
    if (t0 + AF_KT > a.T) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= a.T) st[j][r] = -INFINITY;
        }
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, st[j][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mn = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - mn);  // 0 on the first tile (m = -inf)
    m_run = mn;
    if (alpha != 1.0f) {  // (per lane; a stable running maximum leaves O^T untouched)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[j][r] *= alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[j][r] = __builtin_amdgcn_exp2f(st[j][r] - mn);
        psum += st[j][r];
      }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    // V(tile) has landed, for everybody; every wave has left the S^T phase, so K(tile + 1) may overwrite Ks
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tile + 1 < tile_hi) issue_tile(pb + (size_t)(tile + 1) * 2 * AX_TILE_UNITS, Ks);
    // ---- O^T += V^T P^T: step (j, u) contracts the eight keys of registers 8u .. 8u+7 of half j (per lane half);
    //      A = V units (plane, unit 4j + 2u + lh, d); the weights are split right before their step
    const uint4* vp = Vs + lh * 64 + l31;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int j = s >> 1, u = s & 1;
      uint4 va[2][3];
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) va[dh][pl] = vp[(pl * 8 + 2 * s) * 64 + 32 * dh];
      uint4 pp[3];
      {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = st[j][8 * u + e];
        split8_bf16x3(pv, pp[0], pp[1], pp[2]);
      }
      mfma6x2(ot[0], ot[1], va[0], va[1], pp);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int qi = q0 + l31;
  if (qi < a.T) {
    if constexpr (KSPLIT == 1) {
      const float inv = 1.0f / l_run;
      float* op = a.o + b * a.osb + (long)qi * a.ost + h * a.osh;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) op[j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = ot[j][r] * inv;
    } else {
      const long row = ((long)blockIdx.z * a.B * a.H + bh) * a.T + qi;
      float* op = a.ws_o + row * AF_D;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) op[j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = ot[j][r];
      if (lh == 0) *reinterpret_cast<float2*>(a.ws_ml + row * 2) = make_float2(m_run, l_run);
    }
  }
}

int launch_attn_full_x3(const AttnFullArgs& a, void* planes, bool split, hipStream_t st) {
  const int n_tiles = ceil_div(a.T, AF_KT);
  const int qblocks = ceil_div(a.T, AX_WAVES * AF_QW);
  uint4* pl = reinterpret_cast<uint4*>(planes);
  hipLaunchKernelGGL(attn_kv_planes_kernel, dim3(n_tiles, a.B * a.H), dim3(256), 0, st, a, pl);
  if (split) hipLaunchKernelGGL(attn_full_x3_kernel<AX_KSPLIT>, dim3(qblocks, a.B * a.H, AX_KSPLIT), dim3(AX_WAVES * 64), 0, st, a, (const uint4*)pl);
  else hipLaunchKernelGGL(attn_full_x3_kernel<1>, dim3(qblocks, a.B * a.H), dim3(AX_WAVES * 64), 0, st, a, (const uint4*)pl);
  IX_HIP(hipGetLastError());
  return IXTTS_OK;
}

}  // namespace ixtts
