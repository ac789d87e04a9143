/*
  gts_prims.hpp -- device primitives of the engine, written for gfx950:
  64-lane wavefronts, 256-thread workgroups, LDS-staged digit counters.

    * exclusive prefix sums (u32 / u64), reduce-then-scan over 2048-element tiles
    * stable LSD radix sort of (key, u32 value) pairs, 8-bit digits: per-tile
      digit histograms -> one scan over [digit][tile] -> stable scatter.  The
      rank of a key inside its wavefront comes from eight __ballot()s (the
      lanes holding the same digit) and a popcount of the lower lanes; each
      wavefront owns a contiguous slice of the tile so ranks follow index order.

  These are HBM-bound streaming passes: per 8-bit pass a pair is read twice
  (histogram, scatter) and written once.
*/
#ifndef GTS_PRIMS_HPP
#define GTS_PRIMS_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#define GTS_BLOCK 256
#define GTS_WAVE 64
#define GTS_SCAN_ITEMS 8
#define GTS_SCAN_TILE (GTS_BLOCK * GTS_SCAN_ITEMS)
#define GTS_SORT_ITEMS 16
#define GTS_SORT_TILE (GTS_BLOCK * GTS_SORT_ITEMS)

__device__ __forceinline__ uint32_t gts_lane() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t gts_lanemask_lt()
{
  return (1ull << gts_lane()) - 1ull;
}

/* ------------------------------------------------------------------ */
/* block-wide exclusive scan of one value per thread; returns the exclusive
   prefix and the block total */
template <typename T>
__device__ __forceinline__ T gts_block_exscan(T v, T &total)
{
  __shared__ T wsum[GTS_BLOCK / GTS_WAVE];
  const uint32_t lane = gts_lane(), w = threadIdx.x >> 6;
  T inc = v;
#pragma unroll
  for (int off = 1; off < GTS_WAVE; off <<= 1) {
    T o = __shfl_up(inc, off);
    if (lane >= (uint32_t)off) inc += o;
  }
  if (lane == GTS_WAVE - 1) wsum[w] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < GTS_BLOCK / GTS_WAVE; ++i) {
    if (i < (int)w) base += wsum[i];
    tot += wsum[i];
  }
  __syncthreads();
  total = tot;
  return base + inc - v;
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(GTS_BLOCK)
k_scan_tile_sums(const TI *in, TO *sums, uint64_t n)
{
  const uint64_t base = (uint64_t)blockIdx.x * GTS_SCAN_TILE +
                        (uint64_t)threadIdx.x * GTS_SCAN_ITEMS;
  TO s = 0;
#pragma unroll
  for (int i = 0; i < GTS_SCAN_ITEMS; ++i)
    if (base + i < n) s += (TO)in[base + i];
  TO tot;
  gts_block_exscan<TO>(s, tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

/* out[i] = offs[tile] + exclusive prefix inside the tile; in may alias out */
template <typename TI, typename TO>
__global__ void __launch_bounds__(GTS_BLOCK)
k_scan_tile_apply(const TI *in, TO *out, const TO *offs, uint64_t n)
{
  const uint64_t base = (uint64_t)blockIdx.x * GTS_SCAN_TILE +
                        (uint64_t)threadIdx.x * GTS_SCAN_ITEMS;
  TO v[GTS_SCAN_ITEMS];
  TO s = 0;
#pragma unroll
  for (int i = 0; i < GTS_SCAN_ITEMS; ++i) {
    v[i] = base + i < n ? (TO)in[base + i] : (TO)0;
    s += v[i];
  }
  TO tot;
  TO ex = gts_block_exscan<TO>(s, tot) + (offs ? offs[blockIdx.x] : (TO)0);
#pragma unroll
  for (int i = 0; i < GTS_SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += v[i];
  }
}

/* Exclusive scan of n elements; out may alias in.  tmp must hold
   gts_scan_tmp_elems(n) elements of TO.  If total is not null, the grand total
   is written to *total (device pointer). */
static inline uint64_t gts_scan_tmp_elems(uint64_t n)
{
  uint64_t t = 0;
  while (n > GTS_SCAN_TILE) {
    n = (n + GTS_SCAN_TILE - 1) / GTS_SCAN_TILE;
    t += n + 1;
  }
  return t + 2;
}

template <typename TO>
__global__ void k_store_total(const TO *sums, TO *total) { *total = sums[0]; }

template <typename TI, typename TO>
static void gts_exscan(const TI *in, TO *out, uint64_t n, TO *tmp, TO *total,
                       hipStream_t st)
{
  if (n == 0) {
    if (total) hipMemsetAsync(total, 0, sizeof(TO), st);
    return;
  }
  const uint64_t tiles = (n + GTS_SCAN_TILE - 1) / GTS_SCAN_TILE;
  if (tiles == 1) {
    if (total) {
      k_scan_tile_sums<TI, TO><<<1, GTS_BLOCK, 0, st>>>(in, tmp, n);
      k_store_total<TO><<<1, 1, 0, st>>>(tmp, total);
    }
    k_scan_tile_apply<TI, TO><<<1, GTS_BLOCK, 0, st>>>(in, out, (const TO *)nullptr, n);
    return;
  }
  TO *sums = tmp;
  k_scan_tile_sums<TI, TO><<<(uint32_t)tiles, GTS_BLOCK, 0, st>>>(in, sums, n);
  gts_exscan<TO, TO>(sums, sums, tiles, tmp + tiles + 1, total, st);
  k_scan_tile_apply<TI, TO><<<(uint32_t)tiles, GTS_BLOCK, 0, st>>>(in, out, sums, n);
}

/* ------------------------------------------------------------------ */
/* radix sort */

/* 8-bit digit of a key.  Bit 63 of a 64-bit key is payload that rides along
   (gts_engine.hip, k_pair_keys) and never takes part in the order. */
__device__ __forceinline__ uint32_t gts_digit(uint32_t key, int shift) { return (key >> shift) & 255u; }
__device__ __forceinline__ uint32_t gts_digit(uint64_t key, int shift)
{
  return (uint32_t)((key & ~(1ull << 63)) >> shift) & 255u;
}

template <typename K>
__global__ void __launch_bounds__(GTS_BLOCK)
k_radix_hist(const K *keys, uint64_t n, int shift, uint32_t *hist,
             uint32_t ntiles)
{
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * GTS_SORT_TILE;
#pragma unroll
  for (int i = 0; i < GTS_SORT_ITEMS; ++i) {
    const uint64_t idx = base + (uint64_t)i * GTS_BLOCK + threadIdx.x;
    if (idx < n) atomicAdd(&h[gts_digit(keys[idx], shift)], 1u);
  }
  __syncthreads();
  hist[(uint64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

template <typename K>
__global__ void __launch_bounds__(GTS_BLOCK)
k_radix_scatter(const K *keys, const uint32_t *vals, K *okeys, uint32_t *ovals,
                uint64_t n, int shift, const uint32_t *offs, uint32_t ntiles)
{
  __shared__ uint32_t cnt[GTS_BLOCK / GTS_WAVE][256];
  const uint32_t lane = gts_lane(), w = threadIdx.x >> 6;
  for (int i = 0; i < GTS_BLOCK / GTS_WAVE; ++i) cnt[i][threadIdx.x] = 0;
  __syncthreads();
  const uint64_t wbase = (uint64_t)blockIdx.x * GTS_SORT_TILE +
                         (uint64_t)w * (GTS_SORT_ITEMS * GTS_WAVE);
  K key[GTS_SORT_ITEMS];
  uint32_t val[GTS_SORT_ITEMS], rank[GTS_SORT_ITEMS];
  const uint64_t lt = gts_lanemask_lt();
#pragma unroll
  for (int i = 0; i < GTS_SORT_ITEMS; ++i) {
    const uint64_t idx = wbase + (uint64_t)i * GTS_WAVE + lane;
    const bool valid = idx < n;
    key[i] = valid ? keys[idx] : (K)0;
    val[i] = valid ? vals[idx] : 0u;
    const uint32_t d = gts_digit(key[i], shift);
    uint64_t peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bm = __builtin_amdgcn_ballot_w64(valid && ((d >> b) & 1u));
      peers &= ((d >> b) & 1u) ? bm : ~bm;
    }
    uint32_t prev = 0;
    if (valid) prev = cnt[w][d];
    /* all peers have read the counter before the first of them updates it:
       the wavefront executes the read and the write as separate instructions */
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (valid && (peers & lt) == 0) cnt[w][d] = prev + (uint32_t)__popcll(peers);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    rank[i] = prev + (uint32_t)__popcll(peers & lt);
  }
  __syncthreads();
  if constexpr (sizeof(K) > 4) {
    /* 64-bit keys: as below, but keys and values take turns in the same
       32 KB of LDS (both at once would cost a third of the resident waves) */
    __shared__ K s_key[GTS_SORT_TILE];
    __shared__ uint32_t s_goff[256];
    uint32_t *s_val = (uint32_t *)s_key;
    {
      const uint32_t d = threadIdx.x;
      uint32_t run = 0;
#pragma unroll
      for (int i = 0; i < GTS_BLOCK / GTS_WAVE; ++i) {
        const uint32_t c = cnt[i][d];
        cnt[i][d] = run;
        run += c;
      }
      uint32_t total;
      const uint32_t lbase = gts_block_exscan<uint32_t>(run, total);
#pragma unroll
      for (int i = 0; i < GTS_BLOCK / GTS_WAVE; ++i) cnt[i][d] += lbase;
      s_goff[d] = offs[(uint64_t)d * ntiles + blockIdx.x] - lbase;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < GTS_SORT_ITEMS; ++i) {
      const uint64_t idx = wbase + (uint64_t)i * GTS_WAVE + lane;
      if (idx < n) s_key[cnt[w][gts_digit(key[i], shift)] + rank[i]] = key[i];
    }
    __syncthreads();
    const uint64_t tbase = (uint64_t)blockIdx.x * GTS_SORT_TILE;
    const uint32_t tcount = n - tbase < GTS_SORT_TILE ? (uint32_t)(n - tbase) : (uint32_t)GTS_SORT_TILE;
    uint32_t dst[GTS_SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < GTS_SORT_ITEMS; ++i) {
      const uint32_t t = threadIdx.x + (uint32_t)i * GTS_BLOCK;
      if (t < tcount) {
        const K k = s_key[t];
        dst[i] = s_goff[gts_digit(k, shift)] + t;
        okeys[dst[i]] = k;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < GTS_SORT_ITEMS; ++i) {
      const uint64_t idx = wbase + (uint64_t)i * GTS_WAVE + lane;
      if (idx < n) s_val[cnt[w][gts_digit(key[i], shift)] + rank[i]] = val[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < GTS_SORT_ITEMS; ++i) {
      const uint32_t t = threadIdx.x + (uint32_t)i * GTS_BLOCK;
      if (t < tcount) ovals[dst[i]] = s_val[t];
    }
  } else {
    /* 32-bit keys: the tile is first put in digit order in LDS, then written
       out by consecutive threads: elements of one digit go to consecutive
       addresses, so every run of a digit is written as whole lines instead of
       one store per lane into 64 different lines. */
    __shared__ K s_key[GTS_SORT_TILE];
    __shared__ uint32_t s_val[GTS_SORT_TILE];
    __shared__ uint32_t s_goff[256];
    {
      /* thread d: per-wave counts of digit d -> starts inside the digit's run;
         digit totals -> start of the run in the tile (block scan) */
      const uint32_t d = threadIdx.x;
      uint32_t run = 0;
#pragma unroll
      for (int i = 0; i < GTS_BLOCK / GTS_WAVE; ++i) {
        const uint32_t c = cnt[i][d];
        cnt[i][d] = run;
        run += c;
      }
      uint32_t total;
      const uint32_t lbase = gts_block_exscan<uint32_t>(run, total);
#pragma unroll
      for (int i = 0; i < GTS_BLOCK / GTS_WAVE; ++i) cnt[i][d] += lbase;
      s_goff[d] = offs[(uint64_t)d * ntiles + blockIdx.x] - lbase;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < GTS_SORT_ITEMS; ++i) {
      const uint64_t idx = wbase + (uint64_t)i * GTS_WAVE + lane;
      if (idx < n) {
        const uint32_t d = gts_digit(key[i], shift);
        const uint32_t pos = cnt[w][d] + rank[i];
        s_key[pos] = key[i];
        s_val[pos] = val[i];
      }
    }
    __syncthreads();
    const uint64_t tbase = (uint64_t)blockIdx.x * GTS_SORT_TILE;
    const uint32_t tcount = n - tbase < GTS_SORT_TILE ? (uint32_t)(n - tbase) : (uint32_t)GTS_SORT_TILE;
    for (uint32_t t = threadIdx.x; t < tcount; t += GTS_BLOCK) {
      const K k = s_key[t];
      const uint32_t dst = s_goff[gts_digit(k, shift)] + t;
      okeys[dst] = k;
      ovals[dst] = s_val[t];
    }
  }
}

static inline uint64_t gts_sort_tiles(uint64_t n)
{
  return (n + GTS_SORT_TILE - 1) / GTS_SORT_TILE;
}
/* u32 elements of scratch needed by gts_radix_sort (histogram + scan tmp) */
static inline uint64_t gts_sort_tmp_elems(uint64_t n)
{
  const uint64_t h = 256 * gts_sort_tiles(n);
  return h + gts_scan_tmp_elems(h) + 8;
}

/* Stable sort of n pairs on key bits [shift0 + 8*i) for the given digit
   shifts.  Ping-pongs between (k0, v0) and (k1, v1); returns 0 if the result
   is in (k0, v0), 1 if in (k1, v1). */
template <typename K>
static int gts_radix_sort(K *k0, uint32_t *v0, K *k1, uint32_t *v1, uint64_t n,
                          const int *shifts, int npasses, uint32_t *tmp,
                          hipStream_t st)
{
  if (n == 0) return 0;
  const uint32_t ntiles = (uint32_t)gts_sort_tiles(n);
  uint32_t *hist = tmp;
  uint32_t *scan_tmp = tmp + 256ull * ntiles;
  int cur = 0;
  for (int p = 0; p < npasses; ++p) {
    K *ki = cur ? k1 : k0, *ko = cur ? k0 : k1;
    uint32_t *vi = cur ? v1 : v0, *vo = cur ? v0 : v1;
    k_radix_hist<K><<<ntiles, GTS_BLOCK, 0, st>>>(ki, n, shifts[p], hist, ntiles);
    gts_exscan<uint32_t, uint32_t>(hist, hist, 256ull * ntiles, scan_tmp,
                                   (uint32_t *)nullptr, st);
    k_radix_scatter<K><<<ntiles, GTS_BLOCK, 0, st>>>(ki, vi, ko, vo, n, shifts[p],
                                                     hist, ntiles);
    cur ^= 1;
  }
  return cur;
}

#endif
