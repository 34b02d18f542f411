// l1_peak.hip — the request-rate peak bench.py prices k_wf_trace against, measured on the box the
// bench runs on: per-lane gathers of 64-byte records (4 x global_load_dwordx4 per lane and record, the traversal's node
// fetch) from an L2-resident 4 MB table, every lane a different random record, no arithmetic in between, at the
// occupancy k_wf_trace runs at (6 waves/SIMD).  Also the same with all 64 lanes reading consecutive 16-byte pieces
// (fully coalesced) for comparison, and a float4 copy (HBM).  Prints one JSON line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
template <bool COAL>
__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ tab, uint32_t mask, int iters, float *out) {
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t st = mix(tid + 1u);
  float acc = 0.0f;
  for (int i = 0; i < iters; ++i) {
    st = mix(st + 0x9e3779b9u);
    if (COAL) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t s = __builtin_amdgcn_readfirstlane(mix(st + k));
        const float4 a = tab[((size_t)(s & mask) & ~(size_t)15) * 4 + lane];
        acc += a.x;
      }
    } else {
      const float4 *n = tab + (size_t)(st & mask) * 4;
      const float4 a = n[0], b = n[1], c = n[2], d = n[3];
      acc += a.x + b.y + c.z + d.w;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) b[i] = a[i];
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("{\"error\": \"no device\"}\n"); return 1; }
  const int cus = prop.multiProcessorCount;
  const uint32_t n_nodes = 1u << 16; // 4 MB of 64-byte records
  std::vector<float> h((size_t)n_nodes * 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(rand() & 0xffffff);
  float4 *tab; float *out;
  hipMalloc(&tab, h.size() * 4); hipMalloc(&out, 4);
  hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = cus * 6, iters = 1000; // 6 blocks x 4 waves per CU = 6 waves/SIMD
  double req[2];
  for (int coal = 0; coal < 2; ++coal) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      if (coal) k_gather<true><<<blocks, 256>>>(tab, n_nodes - 1, 50, out); else k_gather<false><<<blocks, 256>>>(tab, n_nodes - 1, 50, out);
      hipEventRecord(e0);
      if (coal) k_gather<true><<<blocks, 256>>>(tab, n_nodes - 1, iters, out); else k_gather<false><<<blocks, 256>>>(tab, n_nodes - 1, iters, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    req[coal] = (double)blocks * 256 * iters * 4.0 / (best * 1e-3);
  }
  const size_t n = (size_t)1 << 27; // 2 GiB in, 2 GiB out
  float4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
  hipMemset(a, 1, n * 16); hipMemset(b, 0, n * 16);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); k_copy<<<cus * 16, 256>>>(a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("{\"cus\": %d, \"gather_lane_requests_per_s\": %.4g, \"coalesced_lane_requests_per_s\": %.4g, \"copy_GBps\": %.1f, "
         "\"what\": \"16-byte lane-requests/s, 64-byte records (4 x dwordx4 per lane) from a 4 MB table, 6 waves/SIMD; float4 copy read+write\"}\n",
         cus, req[0], req[1], 2.0 * n * 16 / (best * 1e-3) / 1e9);
  return 0;
}
