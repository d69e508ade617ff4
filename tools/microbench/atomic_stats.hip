// tools/microbench/atomic_stats.hip - what does it cost to accumulate per-tile BatchNorm partial sums with integer atomics?
// (DESIGN.md 7 open item 2: the 90 BatchNorm finalisation launches of a step exist only to sum per-tile partials in a fixed order; a
// FIXED-POINT integer accumulation is order-independent, i.e. deterministic without the second stage.)  Each of `tiles` workgroups adds
// 2 x C partials as one or two 64-bit limbs to C x 2 (x limbs) addresses; variants: all tiles at once (worst-case burst), no-return atomics.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/atomic_stats.hip -o tools/microbench/atomic_stats
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LIMBS>
__global__ __launch_bounds__(512) void k_acc(unsigned long long* __restrict__ acc, int c, const double* __restrict__ vals) {
  const int t = threadIdx.x;
  if (t >= 2 * c) return;
  const double a = vals[(size_t)blockIdx.x * 2 * c + t];
  // fixed point, scale 2^-48: I = H * 2^40 + L, 0 <= L < 2^40
  const double s = a * 256.0;                       // a * 2^8
  const double h = floor(s);
  const long long H = (long long)h;
  const unsigned long long L = (unsigned long long)((s - h) * 1099511627776.0);     // * 2^40
  if (LIMBS == 2) {
    __hip_atomic_fetch_add(&acc[(size_t)t * 2 + 0], L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&acc[(size_t)t * 2 + 1], (unsigned long long)H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    __hip_atomic_fetch_add(&acc[t], (unsigned long long)(long long)(a * 16777216.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ __launch_bounds__(512) void k_store(double* __restrict__ part, int c, const double* __restrict__ vals) {
  const int t = threadIdx.x;
  if (t >= 2 * c) return;
  part[(size_t)blockIdx.x * 2 * c + t] = vals[(size_t)blockIdx.x * 2 * c + t];
}
int main() {
  struct Case { int tiles, c; } cases[] = {{750, 256}, {750, 128}, {188, 256}, {47, 512}, {1620, 64}, {2750, 256}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# tiles C | partial store (today) us | 1-limb atomics us | 2-limb atomics us | exact sum check\n");
  for (auto cs : cases) {
    const size_t n = (size_t)cs.tiles * 2 * cs.c;
    std::vector<double> h(n);
    unsigned s = 7u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((double)(s >> 8) / 16777216.0 - 0.5) * 1000.0; }
    double *d_vals, *d_part; unsigned long long* d_acc;
    CK(hipMalloc(&d_vals, n * 8)); CK(hipMalloc(&d_part, n * 8)); CK(hipMalloc(&d_acc, (size_t)4 * cs.c * 8));
    CK(hipMemcpy(d_vals, h.data(), n * 8, hipMemcpyHostToDevice));
    float t[3];
    for (int v = 0; v < 3; ++v) {
      std::vector<float> ts;
      for (int it = 0; it < 25; ++it) {
        CK(hipMemset(d_acc, 0, (size_t)4 * cs.c * 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        if (v == 0) hipLaunchKernelGGL(k_store, dim3(cs.tiles), dim3(512), 0, 0, d_part, cs.c, d_vals);
        if (v == 1) hipLaunchKernelGGL(k_acc<1>, dim3(cs.tiles), dim3(512), 0, 0, d_acc, cs.c, d_vals);
        if (v == 2) hipLaunchKernelGGL(k_acc<2>, dim3(cs.tiles), dim3(512), 0, 0, d_acc, cs.c, d_vals);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it >= 5) ts.push_back(ms * 1e3f);
      }
      std::sort(ts.begin(), ts.end()); t[v] = ts[ts.size() / 2];
    }
    // exactness of the 2-limb sum against a long-double host sum of the same fixed-point roundings
    std::vector<unsigned long long> acc((size_t)4 * cs.c);
    CK(hipMemcpy(acc.data(), d_acc, acc.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int col = 0; col < 2 * cs.c; ++col) {
      long double ref = 0;
      for (int tl = 0; tl < cs.tiles; ++tl) ref += (long double)h[(size_t)tl * 2 * cs.c + col];
      const long double got = ((long double)(long long)acc[(size_t)col * 2 + 1] * 1099511627776.0L + (long double)acc[(size_t)col * 2]) / 281474976710656.0L;
      worst = std::max(worst, (double)fabsl(got - ref));
    }
    printf("  %5d %4d | %8.1f | %8.1f | %8.1f | max abs err %.3g\n", cs.tiles, cs.c, t[0], t[1], t[2], worst);
    CK(hipFree(d_vals)); CK(hipFree(d_part)); CK(hipFree(d_acc));
  }
  return 0;
}
