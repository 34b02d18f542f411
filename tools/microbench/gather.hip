// gather.hip — what can one CU's vector L1 deliver to divergent per-lane 64-byte node fetches?
// Every lane walks its own pseudo-random sequence of 64-B records in a table and loads K x 16 B of each.
//   dep=0: the next index does not depend on the loaded data (throughput);
//   dep=1: it does (one traversal step's load -> use -> next address chain).
// Build: hipcc --offload-arch=gfx950 -O3 -o gather gather.hip ; run: ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int K, bool DEP>
__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ tab, uint32_t mask, int iters, float *out) {
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  float acc = 0.0f;
  for (int i = 0; i < iters; ++i) {
    const float4 *n = tab + (size_t)(idx & mask) * 4;
    float4 a = n[0];
    float s = a.x;
    if (K > 1) { float4 b = n[1]; s += b.y; }
    if (K > 2) { float4 c = n[2]; s += c.z; }
    if (K > 3) { float4 d = n[3]; s += d.w; }
    acc += s;
    idx = idx * 1664525u + 1013904223u;
    if (DEP) idx ^= __float_as_uint(s) >> 9;
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int K, bool DEP>
static void run(const float4 *tab, uint32_t n_nodes, int waves_per_simd, float *out, const char *label) {
  int cus = 256;
  int blocks = cus * waves_per_simd; // 4 waves per block, 4 SIMDs per CU
  int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k_gather<K, DEP><<<blocks, 256>>>(tab, n_nodes - 1, 100, out);
  hipEventRecord(e0);
  k_gather<K, DEP><<<blocks, 256>>>(tab, n_nodes - 1, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  double fetches = (double)blocks * 256 * iters;
  double per_cu_cycle = fetches / (ms * 1e-3) / 256.0 / 2.4e9;
  printf("%-6s table %8.1f KB K=%d dep=%d waves/SIMD=%d : %7.3f ms  %6.3f lane-fetch/cycle/CU  %7.1f GB/s (loaded)  %5.2f lane-loads/cycle/CU\n",
         label, n_nodes * 64.0 / 1024, K, (int)DEP, waves_per_simd, ms, per_cu_cycle, fetches * K * 16 / (ms * 1e-3) / 1e9,
         per_cu_cycle * K);
}

int main() {
  const uint32_t max_nodes = 1u << 22; // 256 MB
  std::vector<float> h((size_t)max_nodes * 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(rand() & 0xffffff);
  float4 *tab; float *out;
  hipMalloc(&tab, h.size() * 4); hipMalloc(&out, 4);
  hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (uint32_t n : {256u, 1u << 16, 1u << 18, 1u << 22}) { // 16 KB (L1), 4 MB (L2), 16 MB (MALL), 256 MB (HBM)
    for (int w : {4, 7, 8}) {
      run<1, false>(tab, n, w, out, "thr");
      run<2, false>(tab, n, w, out, "thr");
      run<4, false>(tab, n, w, out, "thr");
      run<1, true>(tab, n, w, out, "chain");
      run<4, true>(tab, n, w, out, "chain");
    }
  }
  return 0;
}
