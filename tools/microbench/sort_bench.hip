// Microbenchmark of gts_prims.hpp's radix sort (test tool): correctness against
// std::stable_sort on a small input, then timing on n pairs.
//   hipcc -O3 --offload-arch=gfx950 -I../../gt-scaffold_amd/csrc sort_bench.hip -o sort_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "gts_prims.hpp"

template <typename K>
static int run(uint64_t n, const int *shifts, int np, int reps, bool check, uint64_t keymask)
{
  std::vector<K> hk(n);
  std::vector<uint32_t> hv(n);
  std::mt19937_64 rng(7);
  for (uint64_t i = 0; i < n; ++i) { hk[i] = (K)(rng() & keymask); hv[i] = (uint32_t)i; }
  K *k0, *k1; uint32_t *v0, *v1, *tmp;
  hipMalloc(&k0, n * sizeof(K)); hipMalloc(&k1, n * sizeof(K));
  hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
  hipMalloc(&tmp, gts_sort_tmp_elems(n) * 4);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f;
  int where = 0;
  for (int r = 0; r < reps; ++r) {
    hipMemcpy(k0, hk.data(), n * sizeof(K), hipMemcpyHostToDevice);
    hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
    hipEventRecord(a, st);
    where = gts_radix_sort<K>(k0, v0, k1, v1, n, shifts, np, tmp, st);
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  int bad = 0;
  if (check) {
    std::vector<K> ok(n); std::vector<uint32_t> ov(n);
    hipMemcpy(ok.data(), where ? k1 : k0, n * sizeof(K), hipMemcpyDeviceToHost);
    hipMemcpy(ov.data(), where ? v1 : v0, n * 4, hipMemcpyDeviceToHost);
    std::vector<uint32_t> idx(n);
    for (uint64_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    uint64_t bits = 0;
    for (int p = 0; p < np; ++p) bits |= 0xFFull << shifts[p];
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) {
      return ((uint64_t)hk[x] & bits) < ((uint64_t)hk[y] & bits); });
    for (uint64_t i = 0; i < n; ++i)
      if (ov[i] != idx[i] || ok[i] != hk[idx[i]]) { if (++bad < 5) printf("mismatch at %llu\n", (unsigned long long)i); }
  }
  const double bytes = (double)n * (sizeof(K) + np * 2.0 * (sizeof(K) + 4));
  printf("K=%zu B n=%llu passes=%d block=%d items=%d: %.3f ms  %.2f TB/s moved%s\n", sizeof(K),
         (unsigned long long)n, np, GTS_SB, GtsSortItems<K>::value, best, bytes / best / 1e9,
         check ? (bad ? "  WRONG" : "  correct+stable") : "");
  hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(tmp);
  return bad != 0;
}

int main(int argc, char **argv)
{
  const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 100000000ull;
  int s64[6] = {0, 8, 16, 32, 40, 48}, s32[3] = {0, 8, 16};
  int rc = 0;
  // few distinct keys: long runs of equal keys across many tiles (stability + look-back)
  rc |= run<uint64_t>(300007, s64, 6, 1, true, 0x0000000300000007ull | (1ull << 63));
  rc |= run<uint64_t>(1000003, s64, 6, 1, true, 0x00FFFFFF00FFFFFFull | (1ull << 63));
  rc |= run<uint32_t>(1000003, s32, 3, 1, true, 0x00FFFFFFull);
  rc |= run<uint32_t>(5, s32, 3, 1, true, 0xFFFFull);
  rc |= run<uint64_t>(n, s64, 6, 3, false, 0x00FFFFFF00FFFFFFull | (1ull << 63));
  rc |= run<uint32_t>(n, s32, 3, 3, false, 0x00FFFFFFull);
  return rc;
}
