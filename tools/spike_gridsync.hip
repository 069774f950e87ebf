// spike_gridsync.hip -- research spike (not part of the library): what does one stage of a persistent
// whole-step decode kernel cost on MI355X?  Measures, per stage, with G workgroups of 256 threads:
//   mode 0: grid barrier only (monotonic agent-scope counter, bounded spin)
//   mode 1: barrier + exchange (each WG writes 16 floats before, every wave reads 2x1280 floats after)
//   mode 2: mode 1 + a 13 MB weight stream per stage, loads issued BEFORE the barrier (prefetch), used after
//   mode 3: like 2 but loads issued AFTER the barrier (what separate kernels effectively do)
// build: hipcc -O3 --offload-arch=gfx950 tools/spike_gridsync.hip -o build/spike_gridsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Args {
  unsigned* ctr;      // [0] barrier counter, [1] abort flag
  float* xbuf;        // [2][2560] ping-pong exchange buffers
  const uint4* w;     // weight stream: stages x (G*256 lanes x NLOAD uint4)
  float* sink;
  int iters, mode, nstage_w;
};

constexpr int NLOAD = 10;
constexpr unsigned SPIN_MAX = 1u << 22;

// barrier variants: 0 = single counter, sleep; 1 = single counter, no sleep; 2 = counter + separate flag line;
// 3 = per-XCD counters -> global counter -> flag; 4 = like 2 but no fences (cost of wbl2/inv)
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
template <int BT>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned gen, unsigned G) {
  // layout (uints): [0] counter, [1] abort, [32] flag, [64 + 32*x] per-XCD counters, [64+32*8] global xcd counter
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    unsigned* flag = ctr + 32;
    if constexpr (BT == 0 || BT == 1) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      unsigned n = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * G) {
        if (++n > SPIN_MAX) { __hip_atomic_store(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        if (BT == 0) __builtin_amdgcn_s_sleep(1);
      }
    } else {
      bool last;
      if constexpr (BT == 3) {
        const unsigned x = xcc_id();
        const unsigned per = G / 8;  // WGs per XCD (G multiple of 8, round-robin dispatch)
        const unsigned o = __hip_atomic_fetch_add(ctr + 64 + 32 * x, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        last = false;
        if (o == gen * per - 1) {
          const unsigned o2 = __hip_atomic_fetch_add(ctr + 64 + 32 * 8, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
          last = (o2 == gen * 8 - 1);
        }
      } else if constexpr (BT == 4) {
        const unsigned o = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (o == gen * G - 1);
      } else {
        const unsigned o = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (o == gen * G - 1);
      }
      if (last) {
        if (BT == 4) __hip_atomic_store(flag, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(flag, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        unsigned n = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
          if (++n > SPIN_MAX) { __hip_atomic_store(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        }
      }
    }
    if (BT != 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

template <int BT>
__global__ __launch_bounds__(256) void stage_kernel(Args a) {
  const int G = gridDim.x;
  const int lane_g = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  uint4 wr[NLOAD];
  for (int it = 0; it < a.iters; ++it) {
    const float* xin = a.xbuf + (it & 1) * 2560;
    float* xout = a.xbuf + ((it + 1) & 1) * 2560;
    if (a.mode == 2) {
      const uint4* wp = a.w + ((size_t)(it % a.nstage_w) * G * 256 + lane_g) * NLOAD;
#pragma unroll
      for (int j = 0; j < NLOAD; ++j) wr[j] = wp[j];
    }
    if (a.mode >= 1) {
      // this WG's outputs of the previous stage: 16 floats (8 rows x 2 slots)
      if (threadIdx.x < 16 && blockIdx.x * 16 + threadIdx.x < 2560) xout[blockIdx.x * 16 + threadIdx.x] = acc + it;
    }
    // (the barrier is entered by every thread of every WG: exit condition = iteration count or abort flag)
    if (!grid_barrier<BT>(a.ctr, (unsigned)(it + 1), (unsigned)G)) break;
    if (a.mode == 3) {
      const uint4* wp = a.w + ((size_t)(it % a.nstage_w) * G * 256 + lane_g) * NLOAD;
#pragma unroll
      for (int j = 0; j < NLOAD; ++j) wr[j] = wp[j];
    }
    if (a.mode >= 1) {
      // every wave reads the whole exchanged vector (2 x 1280 floats = 640 float4: 10 per lane)
      const float4* xp = reinterpret_cast<const float4*>(xout);
      const int lane = threadIdx.x & 63;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const float4 t = xp[j * 64 + lane];
        s += t.x + t.y + t.z + t.w;
      }
      acc += s * 1e-9f;
    }
    if (a.mode >= 2) {
#pragma unroll
      for (int j = 0; j < NLOAD; ++j) acc += __uint_as_float((wr[j].x ^ wr[j].y ^ wr[j].z ^ wr[j].w) & 0x007fffffu) * 1e-30f;
    }
  }
  if (acc == 12345.678f) a.sink[lane_g] = acc;
}

int main(int argc, char** argv) {
  const int iters = 1000;
  unsigned* ctr;
  float *xbuf, *sink;
  uint4* w;
  const int GMAX = 1024;
  const int NSTAGE = 40;  // 40 x 13 MB = 524 MB of distinct weights: no cache reuse between stages
  CK(hipMalloc(&ctr, 4096));
  CK(hipMalloc(&xbuf, 2 * 2560 * 4));
  CK(hipMalloc(&sink, GMAX * 256 * 4));
  const size_t wbytes = (size_t)NSTAGE * GMAX * 256 * NLOAD * 16;
  CK(hipMalloc(&w, wbytes));
  CK(hipMemset(w, 1, wbytes));
  CK(hipMemset(xbuf, 0, 2 * 2560 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grids[] = {256, 320, 512};
  for (int gi = 0; gi < 3; ++gi) {
    const int G = grids[gi];
    for (int bt = 0; bt < 5; ++bt)
    for (int mode = 0; mode < 4; ++mode) {
      if (bt == 3 && G % 8) continue;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctr, 0, 4096));
        Args a{ctr, xbuf, w, sink, iters, mode, NSTAGE};
        CK(hipEventRecord(e0, 0));
        switch (bt) {
          case 0: hipLaunchKernelGGL(stage_kernel<0>, dim3(G), dim3(256), 0, 0, a); break;
          case 1: hipLaunchKernelGGL(stage_kernel<1>, dim3(G), dim3(256), 0, 0, a); break;
          case 2: hipLaunchKernelGGL(stage_kernel<2>, dim3(G), dim3(256), 0, 0, a); break;
          case 3: hipLaunchKernelGGL(stage_kernel<3>, dim3(G), dim3(256), 0, 0, a); break;
          default: hipLaunchKernelGGL(stage_kernel<4>, dim3(G), dim3(256), 0, 0, a); break;
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      unsigned h[2];
      CK(hipMemcpy(h, ctr, 8, hipMemcpyDeviceToHost));
      const double us = best * 1e3 / iters;
      const double mb = (double)G * 256 * NLOAD * 16 / 1e6;
      printf("G=%4d bt=%d mode=%d  %.3f us/stage  (stream %.1f MB/stage -> %.0f GB/s)  abort=%u\n", G, bt, mode, us, mode >= 2 ? mb : 0.0,
             mode >= 2 ? mb * 1e6 / (us * 1e-6) / 1e9 : 0.0, h[1]);
      fflush(stdout);
      if (h[1]) { printf("aborted (barrier timeout)\n"); return 2; }
    }
  }
  return 0;
}
