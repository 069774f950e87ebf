// conv.h -- parameters of the implicit-GEMM conv kernel (conv1d.hip), shared with bigvgan.hip.
#pragma once
#include "common.h"

namespace ixtts {

struct ConvParams {
  const float* x;      // [B][Cin][Tin]
  const float* wp;     // [nphase][ntap][Cin_pad][Cout_pad]
  const float* bias;   // [Cout] or null
  const float* res;    // [B][Cout][Tout] or null  (residual)
  const float* accum;  // [B][Cout][Tout] or null  (running sum of the stage's resblocks)
  const float* accum2; // [B][Cout][Tout] or null  (a second resblock result: y = ((accum + accum2) + conv) / 3)
  float* y;            // [B][Cout][Tout]
  const float* zeros;  // >= 16 floats of zeros in device memory: the target of loads that must not happen (absent operand, padding)
  int B, Cin, Cin_pad, Cout, Cout_pad, Tin, Tout;
  int ntap, dil, off0;  // input index = q + off0 + tap*dil   (dil = -1 for a transposed-conv phase)
  int os, oo;           // output index = q*os + oo (+ phase when nphase > 1)
  int Nq;               // GEMM N extent: q in [0, Nq)
  int nphase;           // 1 for Conv1d, stride for ConvTranspose1d
  int div3;             // divide by 3 (mean of the stage's 3 resblocks, bigvgan.py:375)
  int n_tiles, m_tiles; // filled by the launcher
};

// ---- epilogue shared by conv1d.hip and conv1d_x3.hip: a wave's MT x NT accumulator tiles of 32 x 32 (register r of lane l =
// output channel mrow0 + 32i + (r&3) + 8(r>>2) + 4(l>>5), column qcol0 + 32j + (l&31)) -> y, with bias, residual, the 3-way resblock
// accumulate and /3 fused
typedef float conv_f32x16 __attribute__((ext_vector_type(16)));
template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const conv_f32x16 (&acc)[MT][NT], int mrow0, int qcol0, int b, int phase, int l31, int lh) {
  // ---- epilogue: bias, residual, 3-way accumulate, /3.  Every operand load is unconditional and issued before the first
  // store of its 32x32 block (absent operands and out-of-range elements read the zero page / a clamped element): loads
  // under the bounds branch were waited for one by one, 64+ dependent round trips per lane.
  const size_t ob = (size_t)b * p.Cout * p.Tout;
  const int ophase = p.oo + (p.nphase > 1 ? phase : 0);
  float bv[MT][16];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = min(mrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, p.Cout - 1);
      bv[i][r] = *(p.bias ? p.bias + m : p.zeros);
    }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int q = qcol0 + j * 32 + l31;
      const int t = q * p.os + ophase;
      const bool tv = (q < p.Nq) && (t >= 0) && (t < p.Tout);
      const int tc = min(max(t, 0), p.Tout - 1);
      float rv[16], av[16], av2[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = min(mrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, p.Cout - 1);
        const size_t o = ob + (size_t)m * p.Tout + tc;
        rv[r] = *(p.res ? p.res + o : p.zeros);
        av[r] = *(p.accum ? p.accum + o : p.zeros);
        av2[r] = *(p.accum2 ? p.accum2 + o : p.zeros);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[i][j][r];
        if (p.bias) v += bv[i][r];
        if (p.res) v += rv[r];
        if (p.accum2) v = (av[r] + av2[r]) + v;  // xs = r0; xs += r1; xs += r2 (bigvgan.py:369-375): same order
        else if (p.accum) v = av[r] + v;
        if (p.div3) v = v / 3.0f;
        if (tv && m < p.Cout) p.y[ob + (size_t)m * p.Tout + t] = v;
      }
    }
  }
}

int launch_conv1d(const ConvParams& p, hipStream_t st);
// the same operator with x and the weights as three bf16 planes each (conv1d_x3.hip): p.x = x planes, p.wp = weight planes
int launch_conv1d_x3(const ConvParams& p, hipStream_t st);
int conv_x3_cout_pad(int Cout);
int launch_split_planes(const float* x, void* xplanes, int B, int C, int T, hipStream_t st);
int launch_conv_post(const float* x, const float* w, const float* bias, float* y, int B, int C, int T, hipStream_t st);
int conv_tile_bm(int Cout);
int launch_aa_snake(const float* x, float* y, const float* up12, const float* down12, const float* la,
                    const float* lb, int B, int C, int T, bool fast_sin, hipStream_t st);
// the same activation, written as the x planes of conv1d_x3.hip
int launch_aa_snake_planes(const float* x, void* xplanes, const float* up12, const float* down12, const float* la, const float* lb, int B, int C, int T,
                           bool fast_sin, hipStream_t st);

}  // namespace ixtts
