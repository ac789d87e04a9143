/*
  gts_prims.hpp -- device primitives of the engine, written for gfx950:
  64-lane wavefronts, 256-thread workgroups, LDS-staged digit counters.

    * exclusive prefix sums (u32 / u64), reduce-then-scan over 2048-element tiles
    * stable LSD radix sort of (key, u32 value) pairs, 8-bit digits, "one
      sweep": ONE up-front pass histograms every digit of every pass; each
      pass is then a single kernel that reads a pair once and writes it once.
      A workgroup takes the next tile (atomic ticket, so every earlier tile is
      already running), ranks its keys, publishes its per-digit counts and
      finds its global offsets by looking back over the earlier tiles' entries
      (decoupled look-back: "aggregate" until a tile knows its prefix, then
      "inclusive prefix"), then scatters through LDS.  The rank of a key
      inside its wavefront comes from eight __ballot()s (the lanes holding the
      same digit) and a popcount of the lower lanes; each wavefront owns a
      contiguous slice of the tile so ranks follow index order: the sort is
      stable.

  HBM traffic per pair: 8 (or 4) bytes once for the histograms, then per pass
  one read and one write (2 x 12 B for 64-bit keys) -- the three-kernel
  version (histogram, scan, scatter per pass) read the keys twice per pass.
*/
#ifndef GTS_PRIMS_HPP
#define GTS_PRIMS_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#define GTS_BLOCK 256
#define GTS_WAVE 64
#define GTS_SCAN_ITEMS 8
#define GTS_SCAN_TILE (GTS_BLOCK * GTS_SCAN_ITEMS)

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

/* look-back state of (tile, digit): flag in the two top bits, count below */
#define GTS_LB_AGG 0x40000000u      /* the tile's own count */
#define GTS_LB_PREFIX 0x80000000u   /* count of this and all earlier tiles */
#define GTS_LB_VALUE 0x3FFFFFFFu
/* n below 2^30 so that every count fits under the flags */
#define GTS_ONESWEEP_MAX_N (1ull << 30)

/* global start of digit d's run minus the position of its first element in
   this tile (lbase), for digit d = threadIdx.x.  L2 caches are per XCD and not
   coherent with each other: the entries are written and polled with
   agent-scope atomics, which go past them; an entry carries its data and its
   flag in one word, so no further ordering is needed. */
__device__ __forceinline__ uint32_t gts_lookback(uint32_t *status, uint32_t tile, uint32_t d,
                                                 uint32_t count, uint32_t gbase)
{
  uint32_t *mine = status + (uint64_t)tile * 256 + d;
  if (tile == 0) {
    __hip_atomic_store(mine, GTS_LB_PREFIX | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return gbase;
  }
  __hip_atomic_store(mine, GTS_LB_AGG | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t excl = 0;
  for (uint32_t t = tile; t-- > 0;) {
    const uint32_t *p = status + (uint64_t)t * 256 + d;
    uint32_t v;
    /* tile t holds an earlier ticket: its workgroup is running and publishes
       without waiting for anything later */
    while (((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) &
            (GTS_LB_AGG | GTS_LB_PREFIX)) == 0)
      __builtin_amdgcn_s_sleep(1);
    excl += v & GTS_LB_VALUE;
    if (v & GTS_LB_PREFIX) break;
  }
  __hip_atomic_store(mine, GTS_LB_PREFIX | (excl + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return gbase + excl;
}

/* Where the first pass (and the histogram pass) takes its pairs from.
   GTS_SRC_PLAIN: (keys, vals) arrays.  GTS_SRC_IOTA: keys array, value = index
   (no value array is read).  GTS_SRC_RECORDS (64-bit keys only): the key is made
   from the record's two contig ids a[i], b[i] -- (max << 32) | min, bit 63 = "listed
   from the smaller contig" -- and the value is the record number; ids of nvert
   or more raise *bad (the caller looks at it before anything is indexed with
   them). */
enum { GTS_SRC_PLAIN = 0, GTS_SRC_IOTA = 1, GTS_SRC_RECORDS = 2 };
struct GtsSortSrc {
  int kind;
  const uint32_t *a, *b;
  uint32_t nvert;
  uint32_t *bad;
};
__device__ __forceinline__ uint64_t gts_pair_key(uint32_t a, uint32_t b)
{
  const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
  return ((uint64_t)hi << 32) | lo | (a <= b ? 1ull << 63 : 0ull);
}
template <typename K, int SRC>
__device__ __forceinline__ K gts_sort_key(const K *keys, const GtsSortSrc &src, uint64_t idx)
{
  if constexpr (SRC == GTS_SRC_RECORDS && sizeof(K) == 8) {
    const uint32_t a = src.a[idx], b = src.b[idx];
    if (a >= src.nvert || b >= src.nvert) *src.bad = 1;
    return (K)gts_pair_key(a, b);
  } else
    return keys[idx];
}

/* digit histograms of every pass in one read of the keys: ghist[pass][256] */
template <typename K, int SRC>
__global__ void __launch_bounds__(GTS_BLOCK)
k_onesweep_hist(const K *keys, GtsSortSrc src, uint64_t n, uint32_t *ghist, int npasses, int s0, int s1, int s2,
                int s3, int s4, int s5, int s6, int s7)
{
  __shared__ uint32_t h[8][256];
  const int sh[8] = {s0, s1, s2, s3, s4, s5, s6, s7};
  for (int p = 0; p < npasses; ++p) h[p][threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t idx = (uint64_t)blockIdx.x * GTS_BLOCK + threadIdx.x; idx < n;
       idx += (uint64_t)gridDim.x * GTS_BLOCK) {
    const K k = gts_sort_key<K, SRC>(keys, src, idx);
#pragma unroll
    for (int p = 0; p < 8; ++p)
      if (p < npasses) atomicAdd(&h[p][gts_digit(k, sh[p])], 1u);
  }
  __syncthreads();
  for (int p = 0; p < npasses; ++p)
    if (h[p][threadIdx.x]) atomicAdd(&ghist[p * 256 + threadIdx.x], h[p][threadIdx.x]);
}
/* ghist[pass][*] -> exclusive prefix, one workgroup per pass */
static __global__ void __launch_bounds__(GTS_BLOCK)
k_onesweep_bases(uint32_t *ghist)
{
  uint32_t total;
  uint32_t *h = ghist + blockIdx.x * 256;
  const uint32_t ex = gts_block_exscan<uint32_t>(h[threadIdx.x], total);
  h[threadIdx.x] = ex;
}

/* One pass of the sort: a workgroup of GTS_SB threads takes the next tile of
   GTS_SB x ITEMS pairs.  Each wavefront owns a contiguous slice of the
   tile and ranks its keys item by item (all lanes at once): the lanes holding
   the same digit are found with eight ballots, the rank is the wave's running
   count of the digit plus the number of lower lanes among them.  Threads
   0..255 then turn the per-wave counts into positions inside the tile and look
   the global offsets up (gts_lookback); the tile is put in digit order in LDS
   -- keys, then values through the same buffer -- and written out by
   consecutive threads, so a digit's run goes out as whole lines. */
#ifndef GTS_SB
#define GTS_SB 512
#endif
#define GTS_SW (GTS_SB / GTS_WAVE)
/* pairs per thread: 12 with 64-bit keys, 16 with 32-bit keys (measured on
   100 M pairs, tools/microbench/sort_bench.hip: 64-bit 4.9 ms for six passes
   with 12, 5.7 with 16; 32-bit 1.96 ms for three passes with 16, 2.2 with 12) */
#ifndef GTS_SORT_ITEMS64
#define GTS_SORT_ITEMS64 12
#endif
#ifndef GTS_SORT_ITEMS32
#define GTS_SORT_ITEMS32 16
#endif
template <typename K> struct GtsSortItems { static const int value = sizeof(K) > 4 ? GTS_SORT_ITEMS64 : GTS_SORT_ITEMS32; };

template <typename K, int ITEMS, int SRC>
__global__ void __launch_bounds__(GTS_SB)
k_radix_scatter(const K *keys, const uint32_t *vals, GtsSortSrc src, K *okeys, uint32_t *ovals,
                uint64_t n, int shift, const uint32_t *gbase, uint32_t *status,
                uint32_t *ticket)
{
  __shared__ uint32_t cnt[GTS_SW][256];
  __shared__ K s_key[(GTS_SB * ITEMS)];
  __shared__ uint32_t s_goff[256];
  __shared__ uint32_t s_wsum[4];
  __shared__ uint32_t s_tile;
  const uint32_t lane = gts_lane(), w = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  for (uint32_t i = threadIdx.x; i < GTS_SW * 256; i += GTS_SB) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint64_t tbase = (uint64_t)tile * (GTS_SB * ITEMS);
  const uint64_t wbase = tbase + (uint64_t)w * (ITEMS * GTS_WAVE);
  K key[ITEMS];
  uint32_t val[ITEMS], rank[ITEMS];
  const uint64_t lt = gts_lanemask_lt();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint64_t idx = wbase + (uint64_t)i * GTS_WAVE + lane;
    const bool valid = idx < n;
    key[i] = valid ? gts_sort_key<K, SRC>(keys, src, idx) : (K)0;
    if constexpr (SRC == GTS_SRC_PLAIN) val[i] = valid ? vals[idx] : 0u;
    else val[i] = (uint32_t)idx;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint64_t idx = wbase + (uint64_t)i * GTS_WAVE + lane;
    const bool valid = idx < n;
    const uint32_t d = gts_digit(key[i], shift);
    uint64_t peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bm = __builtin_amdgcn_ballot_w64((d >> b) & 1u);
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
  {
    /* thread d < 256: per-wave counts of digit d -> starts inside the digit's
       run; digit totals -> start of the run in the tile (scan over 256 digits
       by the first four wavefronts) */
    const uint32_t d = threadIdx.x;
    uint32_t run = 0;
    if (d < 256) {
#pragma unroll
      for (int i = 0; i < GTS_SW; ++i) {
        const uint32_t c = cnt[i][d];
        cnt[i][d] = run;
        run += c;
      }
    }
    uint32_t inc = run;
#pragma unroll
    for (int off = 1; off < GTS_WAVE; off <<= 1) {
      const uint32_t o = __shfl_up(inc, off);
      if (lane >= (uint32_t)off) inc += o;
    }
    if (d < 256 && lane == GTS_WAVE - 1) s_wsum[w] = inc;
    __syncthreads();
    if (d < 256) {
      uint32_t lbase = inc - run;
      for (uint32_t i = 0; i < w; ++i) lbase += s_wsum[i];
#pragma unroll
      for (int i = 0; i < GTS_SW; ++i) cnt[i][d] += lbase;
      s_goff[d] = gts_lookback(status, tile, d, run, gbase[d]) - lbase;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint64_t idx = wbase + (uint64_t)i * GTS_WAVE + lane;
    if (idx < n) {
      rank[i] += cnt[w][gts_digit(key[i], shift)];   /* position inside the tile */
      s_key[rank[i]] = key[i];
    }
  }
  __syncthreads();
  const uint32_t tcount = n - tbase < (GTS_SB * ITEMS) ? (uint32_t)(n - tbase) : (uint32_t)(GTS_SB * ITEMS);
  uint32_t dst[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t t = threadIdx.x + (uint32_t)i * GTS_SB;
    if (t < tcount) {
      const K k = s_key[t];
      dst[i] = s_goff[gts_digit(k, shift)] + t;
      okeys[dst[i]] = k;
    }
  }
  __syncthreads();
  uint32_t *s_val = (uint32_t *)s_key;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint64_t idx = wbase + (uint64_t)i * GTS_WAVE + lane;
    if (idx < n) s_val[rank[i]] = val[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t t = threadIdx.x + (uint32_t)i * GTS_SB;
    if (t < tcount) ovals[dst[i]] = s_val[t];
  }
}

static inline uint64_t gts_sort_tiles(uint64_t n, int items = 12)
{
  const uint64_t tile = (uint64_t)GTS_SB * (uint64_t)items;
  return (n + tile - 1) / tile;
}
#define GTS_SORT_MAX_PASSES 8
/* u32 elements of scratch needed by gts_radix_sort: per pass the look-back
   entries of every tile, the digit histograms of all passes, a ticket per pass */
static inline uint64_t gts_sort_tmp_elems(uint64_t n)
{
  /* (tiles of the smaller of the two tile sizes: enough for either key width) */
  const int items = GTS_SORT_ITEMS64 < GTS_SORT_ITEMS32 ? GTS_SORT_ITEMS64 : GTS_SORT_ITEMS32;
  return (256 * gts_sort_tiles(n, items) + 256 + 1) * GTS_SORT_MAX_PASSES + 64;
}

/* Stable sort of n (< 2^30) pairs on key bits [shift0 + 8*i) for the given
   digit shifts (at most GTS_SORT_MAX_PASSES).  Ping-pongs between (k0, v0) and
   (k1, v1); returns 0 if the result is in (k0, v0), 1 if in (k1, v1), -1 if n
   is too large. */
template <typename K, int SRC>
static void gts_sort_first(const K *ki, const uint32_t *vi, const GtsSortSrc &src, K *ko, uint32_t *vo,
                           uint64_t n, const int *sh, int npasses, uint32_t ntiles, uint32_t hgrid,
                           uint32_t *ghist, uint32_t *status, uint32_t *ticket, hipStream_t st)
{
  k_onesweep_hist<K, SRC><<<hgrid, GTS_BLOCK, 0, st>>>(ki, src, n, ghist, npasses, sh[0], sh[1], sh[2],
                                                       sh[3], sh[4], sh[5], sh[6], sh[7]);
  k_onesweep_bases<<<npasses, GTS_BLOCK, 0, st>>>(ghist);
  k_radix_scatter<K, GtsSortItems<K>::value, SRC><<<ntiles, GTS_SB, 0, st>>>(ki, vi, src, ko, vo, n, sh[0],
                                                                            ghist, status, ticket);
}

/* src (optional) describes where the first pass reads from: with
   GTS_SRC_IOTA v0 is never read, with GTS_SRC_RECORDS neither k0 nor v0 is;
   the result still ends in (k0, v0) or (k1, v1) as the return value says. */
template <typename K>
static int gts_radix_sort(K *k0, uint32_t *v0, K *k1, uint32_t *v1, uint64_t n,
                          const int *shifts, int npasses, uint32_t *tmp,
                          hipStream_t st, const GtsSortSrc *srcp = nullptr)
{
  if (n == 0) return 0;
  if (n >= GTS_ONESWEEP_MAX_N || npasses > GTS_SORT_MAX_PASSES) return -1;
  const uint32_t ntiles = (uint32_t)gts_sort_tiles(n, GtsSortItems<K>::value);
  uint32_t *ghist = tmp;                                   /* [npasses][256] */
  uint32_t *ticket = ghist + 256 * GTS_SORT_MAX_PASSES;    /* [npasses] */
  uint32_t *status = ticket + 64;                          /* [npasses][ntiles][256] */
  hipMemsetAsync(tmp, 0, (256ull * GTS_SORT_MAX_PASSES + 64 + 256ull * ntiles * npasses) * 4, st);
  int sh[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = 0; p < npasses; ++p) sh[p] = shifts[p];
  const uint32_t hgrid = ntiles < 4096 ? ntiles : 4096;
  GtsSortSrc src = {GTS_SRC_PLAIN, nullptr, nullptr, 0, nullptr};
  if (srcp) src = *srcp;
  if (npasses == 0) return src.kind == GTS_SRC_PLAIN ? 0 : -1;
  if (src.kind == GTS_SRC_RECORDS && sizeof(K) != 8) return -1;
  if (src.kind == GTS_SRC_RECORDS)
    gts_sort_first<K, GTS_SRC_RECORDS>(k0, v0, src, k1, v1, n, sh, npasses, ntiles, hgrid, ghist, status, ticket, st);
  else if (src.kind == GTS_SRC_IOTA)
    gts_sort_first<K, GTS_SRC_IOTA>(k0, v0, src, k1, v1, n, sh, npasses, ntiles, hgrid, ghist, status, ticket, st);
  else
    gts_sort_first<K, GTS_SRC_PLAIN>(k0, v0, src, k1, v1, n, sh, npasses, ntiles, hgrid, ghist, status, ticket, st);
  int cur = 1;
  for (int p = 1; p < npasses; ++p) {
    K *ki = cur ? k1 : k0, *ko = cur ? k0 : k1;
    uint32_t *vi = cur ? v1 : v0, *vo = cur ? v0 : v1;
    k_radix_scatter<K, GtsSortItems<K>::value, GTS_SRC_PLAIN><<<ntiles, GTS_SB, 0, st>>>(
        ki, vi, src, ko, vo, n, shifts[p], ghist + 256 * p, status + 256ull * ntiles * p, ticket + p);
    cur ^= 1;
  }
  return cur;
}

#endif
