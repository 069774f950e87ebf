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
