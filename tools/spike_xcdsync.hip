// spike_xcdsync.hip -- research spike (not part of the library): can the workgroups that share an XCD (one L2) hand
// data to each other INSIDE a kernel through that L2, and what does such a hand-off cost?
//
// One launch = one "stage" of 256 workgroups x 320 threads (the decode GEMV grid).  Workgroup i is assumed to run on XCD
// i % 8 (checked against HW_REG_XCC_ID).  Each workgroup writes 40 floats of its XCD's 1280-float vector, waits for its
// stores, arrives at a per-XCD counter with a WORKGROUP-scope atomic (executes in the XCD's L2, no cross-XCD coherence
// action), polls the counter with a never-succeeding compare-exchange (an RMW always executes at L2, so it cannot hit a stale L1 line), and
// then reads the whole vector with plain loads (first touch in this launch: L1 was invalidated at kernel start) and
// checks every value.  Mode 1 does the same with AGENT-scope atomics + fences for comparison.
// build: hipcc -O3 --offload-arch=gfx950 tools/spike_xcdsync.hip -o build/spike_xcdsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int G = 256, NT = 320, PER_XCD = G / 8, VEC = 1280, SLICE = VEC / PER_XCD;  // 40 floats per workgroup
constexpr unsigned SPIN_MAX = 1u << 20;

struct Args {
  float* data;        // [8][VEC]
  unsigned* ctr;      // [8][32] (one line per XCD)
  unsigned* stats;    // [0] mismatches, [1] timeouts, [2] xcc mapping mismatches
  unsigned long long* lat;  // [G] barrier latency of this launch in 100 MHz ticks
  float* sink;
  int launch, mode;
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__device__ __forceinline__ float expected(int launch, int xcd, int k) { return (float)(launch * 7 + xcd * 3) + 0.001f * (float)k; }

__global__ __launch_bounds__(NT) void stage_kernel(Args a) {
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  if (threadIdx.x == 0 && xcc_id() != (unsigned)xcd) atomicAdd(a.stats + 2, 1u);
  float* vec = a.data + xcd * VEC;
  if (threadIdx.x < SLICE) vec[j * SLICE + threadIdx.x] = expected(a.launch, xcd, j * SLICE + threadIdx.x);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores are in L2
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  if (threadIdx.x == 0) {
    unsigned* c = a.ctr + xcd * 32;
    t0 = wall_clock64();
    unsigned n = 0;
    if (a.mode == 0) {
      __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // (a compare-exchange that never succeeds: a real RMW at L2.  fetch_or(c, 0) is folded into a plain load by the
      // compiler; that happened to work in this small kernel and spun on a stale L1 line in the real one)
      for (;;) {
        unsigned expect = 0xffffffffu;
        __hip_atomic_compare_exchange_strong(c, &expect, 0xffffffffu, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (expect >= (unsigned)PER_XCD) break;
        if (++n > SPIN_MAX) { atomicAdd(a.stats + 1, 1u); break; }
      }
    } else {
      __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)PER_XCD) {
        if (++n > SPIN_MAX) { atomicAdd(a.stats + 1, 1u); break; }
      }
    }
    t1 = wall_clock64();
    a.lat[blockIdx.x] = t1 - t0;
  }
  __syncthreads();
  asm volatile("" ::: "memory");
  unsigned bad = 0;
  float s = 0.f;
  for (int k = threadIdx.x; k < VEC; k += NT) {
    float v;
    if (a.mode == 0) v = vec[k];
    else v = __hip_atomic_load(vec + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v != expected(a.launch, xcd, k)) ++bad;
    s += v;
  }
  if (bad) atomicAdd(a.stats, bad);
  if (s == 12345.678f) a.sink[0] = s;
}

__global__ void reset_kernel(unsigned* ctr) {
  if (threadIdx.x < 8) ctr[threadIdx.x * 32] = 0u;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 2000;
  const int lds_bytes = argc > 2 ? atoi(argv[2]) : 0;  // dynamic LDS per workgroup: > 80 KB forces one workgroup per CU
  if (lds_bytes > 0) CK(hipFuncSetAttribute((const void*)stage_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  Args a;
  CK(hipMalloc(&a.data, 8 * VEC * sizeof(float)));
  CK(hipMalloc(&a.ctr, 8 * 32 * sizeof(unsigned)));
  CK(hipMalloc(&a.stats, 4 * sizeof(unsigned)));
  CK(hipMalloc(&a.lat, G * sizeof(unsigned long long)));
  CK(hipMalloc(&a.sink, 16));
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(a.data, 0, 8 * VEC * sizeof(float)));
    CK(hipMemset(a.stats, 0, 4 * sizeof(unsigned)));
    a.mode = mode;
    std::vector<unsigned long long> lat(G);
    std::vector<double> maxlat, medlat;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float total_ms = 0.f;
    for (int l = 0; l < launches; ++l) {
      a.launch = l;
      hipLaunchKernelGGL(reset_kernel, dim3(1), dim3(64), 0, 0, a.ctr);
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(stage_kernel, dim3(G), dim3(NT), lds_bytes, 0, a);
      CK(hipEventRecord(e1, 0));
      if (l % 50 == 0) {
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        CK(hipMemcpy(lat.data(), a.lat, G * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::sort(lat.begin(), lat.end());
        maxlat.push_back(lat[G - 1] * 0.01);
        medlat.push_back(lat[G / 2] * 0.01);
      }
    }
    CK(hipDeviceSynchronize());
    unsigned st[4];
    CK(hipMemcpy(st, a.stats, sizeof(st), hipMemcpyDeviceToHost));
    std::sort(maxlat.begin(), maxlat.end());
    std::sort(medlat.begin(), medlat.end());
    if (st[1]) {
      unsigned c[8 * 32];
      CK(hipMemcpy(c, a.ctr, sizeof(c), hipMemcpyDeviceToHost));
      printf("  counters after the last launch:");
      for (int x = 0; x < 8; ++x) printf(" %u", c[x * 32]);
      printf("\n");
    }
    printf("mode %d (%s): launches %d  mismatches %u  timeouts %u  xcc-mapping mismatches %u | arrive->released us: median-of-medians %.2f, "
           "median-of-max %.2f | kernel (event) %.2f us\n",
           mode, mode == 0 ? "workgroup-scope atomics through the XCD's L2" : "agent-scope atomics + acquire/release", launches, st[0], st[1], st[2],
           medlat[medlat.size() / 2], maxlat[maxlat.size() / 2], total_ms / maxlat.size() * 1e3);
  }
  return 0;
}
