// Micro-benchmark: what a streaming kernel achieves on this chip (SURVEY section 8d "Roofline": "confirm on the box with a device
// copy / triad microbench; report both").  The headline tile pass is WRITE-ONLY (tiles start from a clear, the window is a forwarded
// second store), so the ceiling that matters is a pure 16 B/lane store of the same size IN ONE LAUNCH -- launch ramp and drain
// included, as the raster launch pays them -- next to the usual copy and triad figures.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream.hip -o tools/ubench/stream && tools/ubench/stream [json-path]
// Shapes:
//   store      grid-stride 16 B/lane stores (blocks = 8 per CU)
//   store_wg   one 256-thread workgroup per 16 KB (a 64 x 64 BGRA8 bin: the raster kernel's shape), rows of 256 B per 16 lanes
//   store_wg2  the same + the second (forwarded) store of the bin at another address: the tile pass with forwarding
//   copy       16 B/lane load + store,  triad  a = b + s * c on float4
// Sizes: 75.8 MB (the cfg2 tile pass: 41.9 MB of tiles + 33.2 MB of window), 311 MB (the cfg5 pass), 1 GiB (beyond the 256 MiB
// Infinity Cache).  Reported: best and median of 20 single launches, each bracketed by its own hipEvent pair.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <string>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_store(uint4* __restrict__ dst, size_t n16, uint32_t v) {
  const uint4 val = make_uint4(v, v + 1, v + 2, v + 3);
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += size_t(gridDim.x) * 256) dst[i] = val;
}

// one workgroup per 16 KB bin; lane l of wave w writes rows (16 w + (l >> 4) + 4 j), 16 bytes at column 4 (l & 15): the raster kernel's store pattern
__global__ void __launch_bounds__(256) k_store_wg(uint4* __restrict__ dst, uint4* __restrict__ dst2, int bins_x, int pitch16, uint32_t v) {
  const int bx = blockIdx.x % bins_x, by = blockIdx.x / bins_x;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint4 val = make_uint4(v, v + 1, v + 2, v + 3);
  uint4* p = dst + size_t(by * 64 + 16 * w + (l >> 4)) * pitch16 + bx * 16 + (l & 15);
#pragma unroll
  for (int j = 0; j < 4; j++) p[size_t(4 * j) * pitch16] = val;
  if (dst2) {
    uint4* q = dst2 + size_t(by * 64 + 16 * w + (l >> 4)) * pitch16 + bx * 16 + (l & 15);
#pragma unroll
    for (int j = 0; j < 4; j++) q[size_t(4 * j) * pitch16] = val;
  }
}

__global__ void __launch_bounds__(256) k_copy(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += size_t(gridDim.x) * 256) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) k_triad(float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c, size_t n16, float s) {
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += size_t(gridDim.x) * 256) {
    const float4 x = b[i], y = c[i];
    a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
  }
}

__global__ void k_empty() {}

struct Res { std::string name; double bytes, best_us, med_us; };
static std::vector<Res> results;

template <class F>
static void timeit(const char* name, double bytes, F launch) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  std::vector<float> us;
  for (int it = 0; it < 24; it++) {
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    if (it >= 4) us.push_back(ms * 1e3f);
  }
  std::sort(us.begin(), us.end());
  Res r{name, bytes, us.front(), us[us.size() / 2]};
  results.push_back(r);
  printf("%-22s %9.1f MB  best %8.2f us = %7.1f GB/s   median %8.2f us = %7.1f GB/s\n", name, bytes / 1e6, r.best_us, bytes / r.best_us / 1e3,
         r.med_us, bytes / r.med_us / 1e3);
}

int main(int argc, char** argv) {
  const size_t GiB = size_t(1) << 30;
  uint4 *A, *B, *C;
  CHECK(hipMalloc(&A, GiB)); CHECK(hipMalloc(&B, GiB)); CHECK(hipMalloc(&C, GiB));
  CHECK(hipMemset(A, 1, GiB)); CHECK(hipMemset(B, 2, GiB)); CHECK(hipMemset(C, 3, GiB));
  CHECK(hipDeviceSynchronize());
  timeit("empty_launch", 0, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); });
  const size_t sizes[3] = {75774976, 310900000, GiB};
  const char* tags[3] = {"75.8MB", "311MB", "1GiB"};
  for (int s = 0; s < 3; s++) {
    const size_t n16 = sizes[s] / 16;
    char nm[64];
    for (int bpc : {8, 16, 32}) {
      snprintf(nm, sizeof nm, "store_%s_b%d", tags[s], bpc);
      timeit(nm, double(n16) * 16, [&] { hipLaunchKernelGGL(k_store, dim3(256 * bpc), dim3(256), 0, 0, A, n16, 7u); });
    }
    // bin-shaped stores: a surface of 4096-byte rows (1024 px) by as many 64-row bands as the size gives
    const int pitch16 = 4096 / 16, bins_x = 16;
    const int bands = int(sizes[s] / (size_t(4096) * 64));
    snprintf(nm, sizeof nm, "store_wg_%s", tags[s]);
    timeit(nm, double(bands) * 64 * 4096, [&] { hipLaunchKernelGGL(k_store_wg, dim3(bins_x * bands), dim3(256), 0, 0, A, (uint4*)nullptr, bins_x, pitch16, 7u); });
    snprintf(nm, sizeof nm, "store_wg2_%s", tags[s]);
    timeit(nm, double(bands / 2) * 64 * 4096 * 2, [&] { hipLaunchKernelGGL(k_store_wg, dim3(bins_x * (bands / 2)), dim3(256), 0, 0, A, B, bins_x, pitch16, 7u); });
    snprintf(nm, sizeof nm, "copy_%s", tags[s]);
    timeit(nm, double(n16) * 32, [&] { hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, A, B, n16); });
    snprintf(nm, sizeof nm, "triad_%s", tags[s]);
    timeit(nm, double(n16) * 48, [&] { hipLaunchKernelGGL(k_triad, dim3(256 * 16), dim3(256), 0, 0, (float4*)A, (const float4*)B, (const float4*)C, n16, 1.5f); });
  }
  if (argc > 1) {
    FILE* f = fopen(argv[1], "w");
    if (f) {
      fprintf(f, "{\"what\": \"tools/ubench/stream.hip on one MI355X: single launches, hipEvent-bracketed, best / median of 20\", \"results\": [\n");
      for (size_t i = 0; i < results.size(); i++) {
        const Res& r = results[i];
        fprintf(f, "  {\"name\": \"%s\", \"bytes\": %.0f, \"best_us\": %.2f, \"median_us\": %.2f, \"best_GBps\": %.1f, \"median_GBps\": %.1f}%s\n", r.name.c_str(), r.bytes,
                r.best_us, r.med_us, r.bytes ? r.bytes / r.best_us / 1e3 : 0.0, r.bytes ? r.bytes / r.med_us / 1e3 : 0.0, i + 1 < results.size() ? "," : "");
      }
      fprintf(f, "]}\n");
      fclose(f);
    }
  }
  return 0;
}
