// common.h -- shared host/device helpers for libixtts_hip.so (gfx950 only).
#pragma once
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

// 64-lane butterfly sum; every lane ends with the total (fixed order -> deterministic)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace ixtts
