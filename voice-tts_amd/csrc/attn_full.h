// attn_full.h -- arguments shared by the two builds of the DiT attention (attn_full.hip: fp32 MFMA; attn_full_x3.hip: bf16 MFMA
// on three-way split operands).
#pragma once
#include "common.h"

namespace ixtts {

constexpr int AF_D = 64;        // head dim
constexpr int AF_KT = 64;       // keys per tile
constexpr int AF_QW = 32;       // queries per wave
constexpr int AF_WAVES = 4;
constexpr int AF_KSPLIT = 4;    // key-range splits when the grid is too small (see attn_full_f32_kernel)
constexpr int AX_KSPLIT = 5;    // the same for attn_full_x3_kernel: 19 x 16 x 5 workgroups of <= 8 tiles = 2 rounds of 768 slots (4 splits: 2 rounds of 10)

struct AttnFullArgs {
  const float* q;  // element (b, t, h, d) at q[b*sb + t*st + h*sh + d]
  const float* k;
  const float* v;
  float* o;        // same strides as q (separate base)
  long sb, st, sh;       // q/k/v strides in floats
  long osb, ost, osh;    // output strides
  int B, H, T;
  float scale;
  float* ws_o;   // KSPLIT > 1: un-normalised partial outputs [split][B][H][T][64]
  float* ws_ml;  //             and their (running max in log2 units, sum) [split][B][H][T][2]
};

// bytes of K / V planes the x3 build keeps per call: [B*H][tiles][2 (K, V)][3 planes][512 units] x 16 B
size_t attn_full_x3_plane_bytes(int B, int H, int T);
// split pass + attention (+ partials for the caller's merge when a.ws_o is set); planes = attn_full_x3_plane_bytes() of scratch
int launch_attn_full_x3(const AttnFullArgs& a, void* planes, bool split, hipStream_t st);

}  // namespace ixtts
