// gather2.hip — vector-L1 (TCP) request-rate study for per-lane 64-byte record fetches on gfx950.
// Modes (one random 64-B record per "ray", table L1- or L2-resident):
//   own4   : every lane fetches its own record with 4 x dwordx4 (the traversal kernel's pattern)
//   quad   : lanes 4q..4q+3 fetch the 4 pieces of ONE record with one dwordx4 each (16 records / instruction)
//   actN   : own4 with only N of 64 lanes active
//   same   : all lanes fetch the same record (own4)
//   pair   : lanes 2i, 2i+1 fetch the same record (own4)
//   dma    : quad pattern through global_load_lds_dwordx4 + ds_read_b128 back
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

enum { OWN4 = 0, QUAD = 1, ACT = 2, SAME = 3, PAIR = 4, DMA = 5, COAL = 6, QUADN = 7, OCT = 8, DMAN = 9 };

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4 *__restrict__ tab, uint32_t mask, int iters, int nact, float *out) {
  __shared__ float4 land[4][64];
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t st = mix(tid + 1u);
  float acc = 0.0f;
  if (MODE == ACT && lane >= nact) return;
  for (int i = 0; i < iters; ++i) {
    st = mix(st + 0x9e3779b9u);
    if (MODE == OWN4 || MODE == ACT) {
      const float4 *n = tab + (size_t)(st & mask) * 4;
      float4 a = n[0], b = n[1], c = n[2], d = n[3];
      acc += a.x + b.y + c.z + d.w;
    } else if (MODE == SAME) {
      uint32_t s = __builtin_amdgcn_readfirstlane(st);
      const float4 *n = tab + (size_t)(s & mask) * 4;
      float4 a = n[0], b = n[1], c = n[2], d = n[3];
      acc += a.x + b.y + c.z + d.w;
    } else if (MODE == PAIR) {
      uint32_t s = __shfl(st, lane & ~1, 64);
      const float4 *n = tab + (size_t)(s & mask) * 4;
      float4 a = n[0], b = n[1], c = n[2], d = n[3];
      acc += a.x + b.y + c.z + d.w;
    } else if (MODE == QUAD) {
      // 4 instructions serve the wave's 64 records: instruction k serves records of lanes 16k..16k+15
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        uint32_t s = __shfl(st, 16 * kk + (lane >> 2), 64);
        const float4 *n = tab + (size_t)(s & mask) * 4 + (lane & 3);
        float4 a = n[0];
        acc += a.x;
      }
    } else if (MODE == COAL) {
      // 4 fully coalesced dwordx4 (1 KB each) at wave-uniform random 1-KB-aligned places
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        uint32_t s = __builtin_amdgcn_readfirstlane(mix(st + kk));
        const float4 *n = tab + ((size_t)(s & mask) & ~(size_t)15) * 4 + lane;
        float4 a = n[0];
        acc += a.x;
      }
    } else if (MODE == QUADN) {
      // quad pattern without the shuffles: record index derived from (tid >> 2)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        uint32_t s = mix((tid >> 2) * 4u + kk + (uint32_t)i * 0x9e3779b9u);
        const float4 *n = tab + (size_t)(s & mask) * 4 + (lane & 3);
        float4 a = n[0];
        acc += a.x;
      }
    } else if (MODE == OCT) {
      // 8 lanes share a 128-B aligned pair of records
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        uint32_t s = mix((tid >> 3) * 4u + kk + (uint32_t)i * 0x9e3779b9u);
        const float4 *n = tab + ((size_t)(s & mask) & ~(size_t)1) * 4 + (lane & 7);
        float4 a = n[0];
        acc += a.x;
      }
    } else if (MODE == DMAN) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        uint32_t s = mix((tid >> 2) * 4u + kk + (uint32_t)i * 0x9e3779b9u);
        const float4 *n = tab + (size_t)(s & mask) * 4 + (lane & 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)n,
                                         (__attribute__((address_space(3))) void *)&land[wave][0], 16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
      float4 a = land[wave][lane];
      acc += a.x;
    } else if (MODE == DMA) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        uint32_t s = __shfl(st, 16 * kk + (lane >> 2), 64);
        const float4 *n = tab + (size_t)(s & mask) * 4 + (lane & 3);
        // 16 B per lane -> LDS at base + lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)n,
                                         (__attribute__((address_space(3))) void *)&land[wave][0], 16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
      float4 a = land[wave][lane];
      acc += a.x;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int MODE>
static void run(const float4 *tab, uint32_t n_nodes, int wps, int nact, float *out, const char *label) {
  int blocks = 256 * wps, iters = 1000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(tab, n_nodes - 1, 50, nact, out);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(tab, n_nodes - 1, iters, nact, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  double recs = (double)blocks * 4 * (MODE == ACT ? nact : 64) * iters;
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-6s n=%2d table %8.0f KB waves/SIMD=%d : %7.3f ms  %6.3f records/cycle/CU  (%5.1f cycles per wave-fetch of 64 lanes' records)\n",
         label, nact, n_nodes * 64.0 / 1024, wps, ms, recs / cyc / 256.0, cyc * 256.0 / ((double)blocks * 4 * iters));
}

int main() {
  const uint32_t max_nodes = 1u << 18;
  std::vector<float> h((size_t)max_nodes * 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(rand() & 0xffffff);
  float4 *tab; float *out;
  hipMalloc(&tab, h.size() * 4); hipMalloc(&out, 4);
  hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (uint32_t n : {256u, 1u << 16}) {
    for (int w : {4, 7}) {
      run<OWN4>(tab, n, w, 64, out, "own4");
      run<QUAD>(tab, n, w, 64, out, "quad");
      run<QUADN>(tab, n, w, 64, out, "quadn");
      run<OCT>(tab, n, w, 64, out, "oct");
      run<COAL>(tab, n, w, 64, out, "coal");
      run<DMAN>(tab, n, w, 64, out, "dman");
      run<DMA>(tab, n, w, 64, out, "dma");
      run<SAME>(tab, n, w, 64, out, "same");
      run<PAIR>(tab, n, w, 64, out, "pair");
      for (int a : {8, 16, 32, 48}) run<ACT>(tab, n, w, a, out, "act");
    }
  }
  return 0;
}
