// gather4.hip — VERDICT r2 item 3: does a QUAD-COOPERATIVE fetch of the traversal's 64-byte nodes relieve the vector-memory
// pipeline?  A dependent chain of per-lane record fetches (15/16 of them from a hot set, like the top of a BVH), with the
// traversal's ~40 VALU instructions per step, in three forms:
//   own4   every lane fetches its own record: 4 x global_load_dwordx4 (what k_wf_trace does)
//   qdma   the 4 lanes of a quad fetch ONE lane's record per instruction, one 16-byte piece each (a whole 64-byte line
//          per quad and instruction), through LDS-DMA (global_load_lds_dwordx4); 4 instructions cover the wave's 64
//          records; every lane then reads its own record back with 4 x ds_read_b128.  Addresses exchanged with ds_bpermute.
//   qreg   the same quad-cooperative loads into registers, pieces handed to their owners with 12 ds_bpermute
// `act` = lanes that still traverse (the others idle, as in the while-while loop); helper lanes of idle owners are masked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
enum { OWN4 = 0, QDMA = 1, QREG = 2 };

template <int MODE, int WAVES, int VALU>
__global__ __launch_bounds__(WAVES * 64) void k(const float4 *__restrict__ tab, uint32_t mask, uint32_t hot_mask, int iters, int act, float *out) {
  __shared__ float4 land[WAVES][4][64];
  const uint32_t tid = blockIdx.x * (WAVES * 64) + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t st = mix(tid + 1u);
  float acc = 0.0f;
  const bool active = lane < act;
  for (int i = 0; i < iters; ++i) {
    uint32_t h = mix(st);
    uint32_t idx = (h & 15u) ? ((h >> 4) & hot_mask) : ((h >> 4) & mask);
    float4 a = {0, 0, 0, 0}, b = a, c = a, d = a;
    if (MODE == OWN4) {
      if (active) { const float4 *n = tab + (size_t)idx * 4; a = n[0]; b = n[1]; c = n[2]; d = n[3]; }
    } else if (MODE == QDMA) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int owner = 16 * kk + (lane >> 2);
        const uint32_t s = __shfl(idx, owner, 64);
        const bool need = owner < act;
        if (need) __builtin_amdgcn_global_load_lds((glb_void *)(tab + (size_t)s * 4 + (lane & 3)), (lds_void *)&land[wave][kk][0], 16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
      if (active) {
        const float4 *r = &land[wave][lane >> 4][(lane & 15) * 4];
        a = r[0]; b = r[1]; c = r[2]; d = r[3];
      }
    } else {
      float4 p[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int owner = 16 * kk + (lane >> 2);
        const uint32_t s = __shfl(idx, owner, 64);
        p[kk] = make_float4(0, 0, 0, 0);
        if (owner < act) p[kk] = tab[(size_t)s * 4 + (lane & 3)];
      }
      // lane L owns record (L>>4 = instruction, (L&15) = quad): piece q sits in lane ((L&15)*4 + q), register p[L>>4]
      float4 mine[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int src = (lane & 15) * 4 + q;
        float4 v0, v1, v2, v3;
        v0 = make_float4(__shfl(p[0].x, src, 64), __shfl(p[0].y, src, 64), __shfl(p[0].z, src, 64), __shfl(p[0].w, src, 64));
        v1 = make_float4(__shfl(p[1].x, src, 64), __shfl(p[1].y, src, 64), __shfl(p[1].z, src, 64), __shfl(p[1].w, src, 64));
        v2 = make_float4(__shfl(p[2].x, src, 64), __shfl(p[2].y, src, 64), __shfl(p[2].z, src, 64), __shfl(p[2].w, src, 64));
        v3 = make_float4(__shfl(p[3].x, src, 64), __shfl(p[3].y, src, 64), __shfl(p[3].z, src, 64), __shfl(p[3].w, src, 64));
        const int g = lane >> 4;
        mine[q] = g == 0 ? v0 : (g == 1 ? v1 : (g == 2 ? v2 : v3));
      }
      a = mine[0]; b = mine[1]; c = mine[2]; d = mine[3];
    }
    float s = a.x + b.y + c.z + d.w;
#pragma unroll
    for (int v = 0; v < VALU; ++v) s = __builtin_fmaf(s, 1.0001f, a.y);
    acc += s;
    st = st * 1664525u + 1013904223u + (__float_as_uint(s) & 0xffu);
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int WAVES, int VALU>
static void run(const char *label, const float4 *tab, uint32_t n_nodes, uint32_t n_hot, int blocks_per_cu, int act, float *out) {
  int blocks = 256 * blocks_per_cu, iters = 600;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, WAVES, VALU><<<blocks, WAVES * 64>>>(tab, n_nodes - 1, n_hot - 1, 30, act, out);
  hipEventRecord(e0);
  k<MODE, WAVES, VALU><<<blocks, WAVES * 64>>>(tab, n_nodes - 1, n_hot - 1, iters, act, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  double recs = (double)blocks * WAVES * act * iters;
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-5s valu=%2d waves/SIMD=%4.1f act=%2d hot %5.0f KB of %6.0f KB : %7.3f ms  %6.3f records/cycle/CU  %5.1f cycles per wave step\n", label, VALU,
         blocks_per_cu * WAVES / 4.0, act, n_hot * 64.0 / 1024, n_nodes * 64.0 / 1024, ms, recs / cyc / 256.0, cyc * 256.0 / ((double)blocks * WAVES * iters));
}

int main() {
  const uint32_t max_nodes = 1u << 16;
  std::vector<float> h((size_t)max_nodes * 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(rand() & 0xffffff);
  float4 *tab; float *out;
  hipMalloc(&tab, h.size() * 4); hipMalloc(&out, 4);
  hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (uint32_t hot : {256u, 2048u, 1u << 16})
    for (int act : {64, 40, 20})
      for (int bpc : {6}) {
        run<OWN4, 4, 40>("own4", tab, max_nodes, hot, bpc, act, out);
        run<QDMA, 4, 40>("qdma", tab, max_nodes, hot, bpc, act, out);
        run<QREG, 4, 40>("qreg", tab, max_nodes, hot, bpc, act, out);
      }
  return 0;
}
