// fetch_calib.hip — what rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ* / WRITE_SIZE report on gfx950 for KNOWN byte counts
// in the access patterns the path-trace kernels use: MI355X_MICROARCH.md documents the
// factor 2 (FETCH_SIZE = half the bytes) for wide coalesced streaming reads only.
//
// One kernel per pattern (the profiler reports per kernel name); every kernel reads exactly `reads` records:
//   k_stream16           64 lanes x 16 B coalesced (the documented case)
//   k_stream4            64 lanes x  4 B coalesced
//   k_gather<B>          one random B-byte record per lane from a table (B = 4, 8, 16, 64; 64 = 4 x dwordx4), each
//                        128-byte line of the table is hit at most ~once: record index = lane-unique random permutation
//   k_gather_pair8       two 8-byte loads 32 bytes apart in one random 128-byte line (the environment lookup's shape)
//   k_hitrec192          one random 192-byte, 64-byte-aligned record per lane (12 x dwordx4: the hit record's shape)
//   k_write16 / k_write12 / k_write_scatter8   stores: coalesced 16 B, 12-byte elements, random 8-byte records
// Table: 2 GiB (8 x the 256 MiB Infinity Cache), so reads come from DRAM; a second set of runs uses a 64 MiB table
// (Infinity-Cache resident after the first pass) to show that cache hits ARE counted by FETCH_SIZE.
// Usage: fetch_calib [table_MiB=2048] [reads_M=64]      (prints the algorithmic bytes per kernel; counters come from
//        rocprofv3 --pmc ... -- ./fetch_calib, see tools/fetch_calib.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// a bijection on [0, 2^bits): odd multiplier + xorshift + odd multiplier, all mod 2^bits
__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t bits) {
  const uint32_t m = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
  uint32_t x = (i * 0x9e3779b1u) & m;
  x ^= x >> (bits / 2 + 1);
  x = (x * 0x85ebca6bu) & m;
  x ^= x >> (bits / 2 + 1);
  return x & m;
}

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f3 __attribute__((ext_vector_type(3), aligned(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_stream16(const f4 *__restrict__ t, size_t n, float *out) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) { f4 v = t[i]; acc += v.x + v.w; }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream4(const float *__restrict__ t, size_t n, float *out) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) acc += t[i];
  if (acc == 123.456f) out[0] = acc;
}
// B-byte records, one per 128-byte line slot: record r lives at line perm(r) (lines = 2^bits), offset (r % (128/B)) * B
template <int B>
__global__ __launch_bounds__(256) void k_gather(const char *__restrict__ t, uint32_t bits, uint32_t reads, float *out) {
  float acc = 0.f;
  for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < reads; r += gridDim.x * 256u) {
    const size_t line = perm(r, bits);
    const char *p = t + line * 128u + (size_t)((mix(r) % (128u / (B > 64 ? 64 : B))) * (B > 64 ? 64 : B));
    if (B == 4) acc += *reinterpret_cast<const float *>(p);
    else if (B == 8) { f2 v = *reinterpret_cast<const f2 *>(p); acc += v.x + v.y; }
    else if (B == 16) { f4 v = *reinterpret_cast<const f4 *>(p); acc += v.x + v.w; }
    else { const f4 *q = reinterpret_cast<const f4 *>(p); f4 a = q[0], b = q[1], c = q[2], d = q[3]; acc += a.x + b.y + c.z + d.w; }
  }
  if (acc == 123.456f) out[0] = acc;
}
// Round 6: the same 16-byte random read with the load's cache-policy bits set (gfx940+ sc0 / sc1 / nt): does any policy
// make the L2 ask the fabric for LESS than a 128-byte line (TCC_EA0_RDREQ_32B / _64B)?  The environment lookups of the
// logic kernel use 16 bytes of every line they fetch.
template <int POLICY> // 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc1 nt, 6 sc0 sc1 nt
__global__ __launch_bounds__(256) void k_gather16_pol(const char *__restrict__ t, uint32_t bits, uint32_t reads, float *out) {
  float acc = 0.f;
  for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < reads; r += gridDim.x * 256u) {
    const size_t line = perm(r, bits);
    const char *p = t + line * 128u + (size_t)((mix(r) % 8u) * 16u);
    f4 v;
    if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POLICY == 5) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    acc += v.x + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_gather_pair8(const char *__restrict__ t, uint32_t bits, uint32_t reads, float *out) {
  float acc = 0.f;
  for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < reads; r += gridDim.x * 256u) {
    const size_t line = perm(r, bits);
    const char *p = t + line * 128u + (size_t)((mix(r) % 7u) * 4u + (mix(r + 77u) % 3u) * 32u);
    f2 a = *reinterpret_cast<const f2 *>(p), b = *reinterpret_cast<const f2 *>(p + 32);
    acc += a.x + a.y + b.x + b.y;
  }
  if (acc == 123.456f) out[0] = acc;
}
// 192-byte records at 64-byte alignment: record r at (perm(r) * 192) - consecutive records tile the table without gaps
__global__ __launch_bounds__(256) void k_hitrec192(const char *__restrict__ t, uint32_t bits, uint32_t reads, float *out) {
  float acc = 0.f;
  for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < reads; r += gridDim.x * 256u) {
    const f4 *q = reinterpret_cast<const f4 *>(t + (size_t)perm(r, bits) * 192u);
    f4 s = q[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) s += q[i];
    acc += s.x + s.y + s.z + s.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write16(f4 *t, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) t[i] = (f4){1.f, 2.f, 3.f, (float)i};
}
__global__ __launch_bounds__(256) void k_write12(float *t, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) *reinterpret_cast<f3 *>(t + 3 * i) = (f3){1.f, 2.f, (float)i};
}
__global__ __launch_bounds__(256) void k_write_scatter8(char *t, uint32_t bits, uint32_t writes) {
  for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < writes; r += gridDim.x * 256u)
    *reinterpret_cast<f2 *>(t + (size_t)perm(r, bits) * 128u + (mix(r) % 16u) * 8u) = (f2){1.f, (float)r};
}

static uint32_t log2u(size_t x) { uint32_t b = 0; while ((1ull << (b + 1)) <= x) ++b; return b; }

int main(int argc, char **argv) {
  const size_t mib = argc > 1 ? strtoull(argv[1], 0, 10) : 2048;
  const uint32_t reads = (uint32_t)((argc > 2 ? strtoull(argv[2], 0, 10) : 64) << 20);
  const size_t bytes = mib << 20;
  char *t; float *out;
  if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(t, 1, bytes);
  hipDeviceSynchronize();
  const uint32_t line_bits = log2u(bytes / 128);        // lines the gathers spread over (power of two)
  const uint32_t rec_bits = log2u(bytes / 192);
  const uint32_t rd = reads < (1u << line_bits) ? reads : (1u << line_bits);  // at most one record per line
  const uint32_t rd192 = reads / 4 < (1u << rec_bits) ? reads / 4 : (1u << rec_bits);
  const int grid = 256 * 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timed = [&](const char *name, double alg_bytes, auto launch) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("{\"kernel\": \"%s\", \"table_MiB\": %zu, \"alg_bytes\": %.0f, \"ms\": %.3f, \"alg_GBps\": %.1f}\n", name, mib, alg_bytes, ms,
           alg_bytes / (ms * 1e-3) / 1e9);
  };
  const size_t n16 = bytes / 16, n4 = bytes / 16; // the 4-byte stream reads a quarter of the table
  timed("k_stream16", 16.0 * n16, [&] { k_stream16<<<grid, 256>>>((const f4 *)t, n16, out); });
  timed("k_stream4", 4.0 * n4, [&] { k_stream4<<<grid, 256>>>((const float *)t, n4, out); });
  timed("k_gather<4>", 4.0 * rd, [&] { k_gather<4><<<grid, 256>>>(t, line_bits, rd, out); });
  timed("k_gather<8>", 8.0 * rd, [&] { k_gather<8><<<grid, 256>>>(t, line_bits, rd, out); });
  timed("k_gather<16>", 16.0 * rd, [&] { k_gather<16><<<grid, 256>>>(t, line_bits, rd, out); });
  timed("k_gather16_pol<1>", 16.0 * rd, [&] { k_gather16_pol<1><<<grid, 256>>>(t, line_bits, rd, out); });
  timed("k_gather16_pol<2>", 16.0 * rd, [&] { k_gather16_pol<2><<<grid, 256>>>(t, line_bits, rd, out); });
  timed("k_gather16_pol<3>", 16.0 * rd, [&] { k_gather16_pol<3><<<grid, 256>>>(t, line_bits, rd, out); });
  timed("k_gather16_pol<4>", 16.0 * rd, [&] { k_gather16_pol<4><<<grid, 256>>>(t, line_bits, rd, out); });
  timed("k_gather16_pol<5>", 16.0 * rd, [&] { k_gather16_pol<5><<<grid, 256>>>(t, line_bits, rd, out); });
  timed("k_gather16_pol<6>", 16.0 * rd, [&] { k_gather16_pol<6><<<grid, 256>>>(t, line_bits, rd, out); });
  timed("k_gather<64>", 64.0 * rd, [&] { k_gather<64><<<grid, 256>>>(t, line_bits, rd, out); });
  timed("k_gather_pair8", 16.0 * rd, [&] { k_gather_pair8<<<grid, 256>>>(t, line_bits, rd, out); });
  timed("k_hitrec192", 192.0 * rd192, [&] { k_hitrec192<<<grid, 256>>>(t, rec_bits, rd192, out); });
  timed("k_write16", 16.0 * n16, [&] { k_write16<<<grid, 256>>>((f4 *)t, n16); });
  timed("k_write12", 12.0 * (bytes / 12), [&] { k_write12<<<grid, 256>>>((float *)t, bytes / 12); });
  timed("k_write_scatter8", 8.0 * rd, [&] { k_write_scatter8<<<grid, 256>>>(t, line_bits, rd); });
  return 0;
}
