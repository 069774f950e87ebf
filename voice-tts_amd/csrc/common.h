// common.h -- shared host/device helpers for libixtts_hip.so (gfx950 only).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>

#include "../../include/ixtts_hip.h"

namespace ixtts {

// ---- error plumbing: no exceptions across the ABI ---------------------------------
void set_error(const char* fmt, ...);

#define IX_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      ::ixtts::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return IXTTS_ERR_HIP;                                                            \
    }                                                                                  \
  } while (0)

#define IX_ARG(cond, ...)                  \
  do {                                     \
    if (!(cond)) {                         \
      ::ixtts::set_error(__VA_ARGS__);     \
      return IXTTS_ERR_ARG;                \
    }                                      \
  } while (0)

#define IX_TRY(expr)            \
  do {                          \
    int _r = (expr);            \
    if (_r != IXTTS_OK) return _r; \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device helpers -------------------------------------------------------------
typedef __hip_bfloat16 bf16;

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}
__device__ __forceinline__ float lo_bf16(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf16(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

// ---- wavefront reductions on DPP (no LDS crossbar): fixed order -> bit-reproducible.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_take(float v) {
  // lanes disabled by the masks / reading out of range receive 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, true));
}
// total lands in lane 63
__device__ __forceinline__ float wave_sum63(float v) {
  float s = v + dpp_take<0x111, 0xf, 0xf>(v);  // row_shr:1
  s += dpp_take<0x112, 0xf, 0xf>(v);           // row_shr:2
  s += dpp_take<0x113, 0xf, 0xf>(v);           // row_shr:3
  s += dpp_take<0x114, 0xf, 0xe>(s);           // row_shr:4
  s += dpp_take<0x118, 0xf, 0xc>(s);           // row_shr:8
  s += dpp_take<0x142, 0xa, 0xf>(s);           // row_bcast:15
  s += dpp_take<0x143, 0xc, 0xf>(s);           // row_bcast:31
  return s;
}
// every lane ends with the total
__device__ __forceinline__ float wave_sum(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum63(v)), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// the same on DPP (lanes without a source keep their own value): the maximum lands in lane 63, then goes to every lane
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_take_or_self(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  float s = fmaxf(v, dpp_take_or_self<0x111, 0xf, 0xf>(v));  // row_shr:1
  s = fmaxf(s, dpp_take_or_self<0x112, 0xf, 0xf>(v));        // row_shr:2
  s = fmaxf(s, dpp_take_or_self<0x113, 0xf, 0xf>(v));        // row_shr:3
  s = fmaxf(s, dpp_take_or_self<0x114, 0xf, 0xe>(s));        // row_shr:4
  s = fmaxf(s, dpp_take_or_self<0x118, 0xf, 0xc>(s));        // row_shr:8
  s = fmaxf(s, dpp_take_or_self<0x142, 0xa, 0xf>(s));        // row_bcast:15
  s = fmaxf(s, dpp_take_or_self<0x143, 0xc, 0xf>(s));        // row_bcast:31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 63));
}
// inclusive prefix sums over the wave's lanes (the scan wave_sum63 is built on), unsigned
__device__ __forceinline__ unsigned int wave_scan_u32(unsigned int v) {
  auto take = [](unsigned int x, auto ctrl, auto rm, auto bm) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl)::value, decltype(rm)::value, decltype(bm)::value, true);
  };
  using std::integral_constant;
  unsigned int s = v + take(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});
  s += take(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});
  s += take(v, integral_constant<int, 0x113>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{});
  s += take(s, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xe>{});
  s += take(s, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xc>{});
  s += take(s, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}, integral_constant<int, 0xf>{});
  s += take(s, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}, integral_constant<int, 0xf>{});
  return s;
}

// minimum of an int over the wave (DPP), to every lane
__device__ __forceinline__ int wave_min_i32_dpp(int v) {
  auto take = [](int x, auto ctrl, auto rm, auto bm) {
    return __builtin_amdgcn_update_dpp(x, x, decltype(ctrl)::value, decltype(rm)::value, decltype(bm)::value, false);
  };
  using std::integral_constant;
  int s = min(v, take(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{}));
  s = min(s, take(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{}));
  s = min(s, take(v, integral_constant<int, 0x113>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xf>{}));
  s = min(s, take(s, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xe>{}));
  s = min(s, take(s, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{}, integral_constant<int, 0xc>{}));
  s = min(s, take(s, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}, integral_constant<int, 0xf>{}));
  s = min(s, take(s, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}, integral_constant<int, 0xf>{}));
  return __builtin_amdgcn_readlane(s, 63);
}

// ---- fp32 -> three bf16 pieces (the split-product kernels: conv1d_x3.hip, attn_full_x3.hip, aa_snake.hip)
// v = h + m + l EXACTLY: each piece is the remainder before it rounded to 8 significant bits (to nearest, ties away: truncated
// pieces would all err the same way in the products that are dropped); at most 8 significant bits are left for l.  Eight values
// become three 16-byte units of eight bf16: word w of a unit = values (2w, 2w+1), value 2w in the low 16 bits.
__device__ __forceinline__ void split8_bf16x3(const float (&v)[8], uint4& ph, uint4& pm, uint4& pl) {
  unsigned int h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = __float_as_uint(v[e]) + 0x8000u;
    const float r = v[e] - __uint_as_float(h[e] & 0xffff0000u);
    m[e] = __float_as_uint(r) + 0x8000u;
    l[e] = __float_as_uint(r - __uint_as_float(m[e] & 0xffff0000u));
  }
  auto pk = [](unsigned int hi, unsigned int lo) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); };  // the high halves of both
  ph = make_uint4(pk(h[1], h[0]), pk(h[3], h[2]), pk(h[5], h[4]), pk(h[7], h[6]));
  pm = make_uint4(pk(m[1], m[0]), pk(m[3], m[2]), pk(m[5], m[4]), pk(m[7], m[6]));
  pl = make_uint4(pk(l[1], l[0]), pk(l[3], l[2]), pk(l[5], l[4]), pk(l[7], l[6]));
}

}  // namespace ixtts
