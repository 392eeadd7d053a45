// psk_lookup.hpp -- partitioned LOOKUPS for the counter structures (CountMinSketch.check, CountingBloomFilter.check).
//
// The direct lookup kernels fetch one 64-byte line across the fabric per 4-byte counter (266-454 B per key where 40-48
// are algorithmic, rocprof r01) and sit on the fabric's request ceiling (~64 G gathers/s).  A lookup needs a RETURN trip
// as well: the k counters of one key live in k different slices, i.e. they are read by k different workgroups.  Three
// passes, every byte moved in coalesced 16-byte pieces:
//
//   pass 1  k_part_scatter (psk_partition.hpp), payload-free probes (8 x 16-bit slice-local cells per group, the unit
//           counter-add format) + two by-products of its LDS counting sort:
//             perm[key]      the position of each of the key's k probes inside the tile's sorted stage (16 bits each,
//                            ceil(k / 2) dwords per key: PermRec, psk_partition.hpp)
//             runinfo[tile][slice] = (first group of the tile's run inside its (slice, workgroup) segment,
//                                     stage offset << 16 | probe count)
//   pass 2  k_counter_gather: one workgroup per slice keeps the slice (2^15 counters = 128 KiB) in LDS, streams the slice's
//           probe groups and writes the counter VALUES to a buffer shaped like the probe buffer (same group, same slot).
//   pass 3  k_lookup_collect: one workgroup per tile copies the tile's runs of values back into the tile's sorted order in
//           LDS (runinfo), then every key picks its k values through perm[] and applies the query -- min / mean /
//           mean-min (countminsketch.py:429-453) or the CBF min (countingbloom.py:166-174) -- and stores its result.
//
// Probes carry no key id and results need no atomics.  A (slice, workgroup) segment that overflows (adversarial /
// duplicate-heavy batches) raises a device flag; the host then enqueues the flag-guarded direct kernel over the round
// (k_apply_if), so the result is exact for any input.
#pragma once
#include "psk_partition.hpp"

namespace psk {

struct PayUnitLookup {  // as PayUnit (8 x 16-bit cells per group) + the by-products pass 3 needs
    static constexpr int mode = kModePlain;
    static constexpr int group = 8;
    static constexpr bool lookup = true;
    uint32_t *perm;    // [round keys][PermRec<KT>::PD dwords]
    uint2 *runinfo;    // [tiles][slices]
    __device__ __forceinline__ uint32_t operator()(uint64_t, uint64_t) const { return 0; }
};

struct PayBloomLookup {  // as PayNone (6 x 20-bit bit indices per group) + the by-products (Bloom lookups, below)
    static constexpr int mode = kModePlain;
    static constexpr int group = 6;
    static constexpr bool lookup = true;
    // 4096-key tiles (32 probes per thread) for tables of ~900 slices and more -- the 1 GiB CountingBloomFilter's 1024 nibble slices, Bloom tables of
    // 2^30 bits and more: a 2048-key tile brings 14 probes = 2.3 groups per slice there, and pass 3 copies every (tile, slice) run of values on
    // its own (launch_scatter_nt keeps 2048-key tiles below that: kLookupFatSlices)
    static constexpr bool fat1024 = true;
    uint32_t *perm;
    uint2 *runinfo;
    __device__ __forceinline__ uint32_t operator()(uint64_t, uint64_t) const { return 0; }
};

struct SpillRaiseFlag {  // a lookup probe that found its segment full: the round is redone by the direct kernel
    uint32_t *flag;
    __device__ __forceinline__ void operator()(uint32_t, uint32_t) const { *flag = 1u; }
};

// ------------------------------------------------------------------------------------ pass 2
constexpr int kGatherDepth = 8;

// The counter behind every probe of every group, in the probe buffer's shape (pads: unspecified).  Per SLICE the values are
// written either as 8 x uint32 (32 bytes per group) or -- when every counter of the slice is below 2^16 (the usual case: a
// pass over the slice while it is loaded tells) -- as 8 x uint16 (16 bytes per group, dense); fmt[slice] says which.
// Grid: nbuckets * max(1, g.split) workgroups; with g.split > 1 the workgroups of a slice share its segments
// (slice counts that do not fill the CUs -- CMS 5 x 2^20: 320 slices on 256 CUs -- left pass 2 unbalanced).
// HALF (tables of more than 2048 x 2^15 counters, up to 2^27): slices of 2^16 counters whose LDS image holds 16-BIT values
// (128 KiB); the values always leave in the 16-bit format.  A slice with a counter at or above 2^16 (or a negative CMS bin)
// cannot be held that way: it raises *flag and the host's flag-guarded direct kernel redoes the batch (exact, slow, rare --
// the counters of a CountingBloomFilter are small; a CBF for 10 M elements at 1 % already has 9.6e7 of them).
template <bool HALF>
__global__ __launch_bounds__(kApplyThreads) void k_counter_gather(const uint32_t *tab, uint64_t tab_cells, PartGeom g,
                                                                  const uint32_t *segcnt, const uint4 *buckets, uint4 *vals, uint8_t *fmt, uint32_t *flag)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t wave_max[kApplyWaves];
    const uint32_t S = g.split > 1 ? g.split : 1;
    const uint32_t b = blockIdx.x / S;
    g.split_idx = blockIdx.x % S;
    const uint32_t slice_cells = 1u << g.shift;
    const uint32_t mask = slice_cells - 1;
    const uint64_t c0 = (uint64_t)b * slice_cells;
    const uint16_t *smem16 = reinterpret_cast<const uint16_t *>(smem);
    const uint32_t mycnt = lane_segment_count(segcnt, g, b);  // (requested before the slice: see lane_segment_count)
    uint32_t mx = 0;
    for (uint32_t wb = threadIdx.x * 4; wb < slice_cells; wb += kApplyThreads * 4 * kSliceLoads) {  // kSliceLoads pieces in flight per lane
        uint4 tt[kSliceLoads];
#pragma unroll
        for (int u = 0; u < kSliceLoads; ++u) {
            const uint32_t w = wb + (uint32_t)u * kApplyThreads * 4;
            tt[u] = w < slice_cells ? slice_piece(tab, tab_cells, c0, w) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kSliceLoads; ++u) {
            const uint32_t w = wb + (uint32_t)u * kApplyThreads * 4;
            const uint4 t = tt[u];
            if (w < slice_cells) {
                if (HALF) *reinterpret_cast<uint2 *>(smem + w / 2) = make_uint2((t.x & 0xFFFFu) | (t.y << 16), (t.z & 0xFFFFu) | (t.w << 16));
                else *reinterpret_cast<uint4 *>(smem + w) = t;
            }
            mx |= t.x | t.y | t.z | t.w;  // (an OR is enough to tell whether any counter has a bit at or above 2^16; negative CMS bins do)
        }
    }
    for (int o = 32; o > 0; o >>= 1) mx |= __shfl_down(mx, o);
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = mx;
    __syncthreads();
    uint32_t all = 0;
#pragma unroll
    for (int w = 0; w < kApplyWaves; ++w) all |= wave_max[w];
    const bool narrow = HALF || all < 65536u;
    if (threadIdx.x == 0) {
        fmt[b] = narrow ? 1 : 0;  // (every workgroup of the slice writes the same value)
        if (HALF && all >= 65536u) *flag = 1u;
    }
    auto cell = [&](uint32_t i) -> uint32_t { return HALF ? (uint32_t)smem16[i & mask] : smem[i & mask]; };
    for_each_batch_at<kGatherDepth>(buckets, segcnt, g, b, make_uint4(0, 0, 0, 0), [&](const uint4 (&q)[kGatherDepth], const uint64_t (&at)[kGatherDepth],
                                                                                      const uint32_t (&wg)[kGatherDepth]) {
        uint4 lo[kGatherDepth], hi[kGatherDepth];
#pragma unroll
        for (int d = 0; d < kGatherDepth; ++d) {  // the 8 LDS reads of every group first (a pad cell 0xFFFF reads a harmless in-slice word)
            lo[d] = make_uint4(cell(q[d].x & 0xFFFFu), cell(q[d].x >> 16), cell(q[d].y & 0xFFFFu), cell(q[d].y >> 16));
            hi[d] = make_uint4(cell(q[d].z & 0xFFFFu), cell(q[d].z >> 16), cell(q[d].w & 0xFFFFu), cell(q[d].w >> 16));
        }
#pragma unroll
        for (int d = 0; d < kGatherDepth; ++d) {
            if (at[d] != ~0ULL) {
                if (narrow) {  // the segment's groups packed at 16 bytes each from the segment's start
                    const uint64_t seg4 = seg_index(g, b, wg[d]) * g.segcap;  // == at[d] - slot
                    vals[2 * seg4 + (at[d] - seg4)] = make_uint4(lo[d].x | (lo[d].y << 16), lo[d].z | (lo[d].w << 16), hi[d].x | (hi[d].y << 16), hi[d].z | (hi[d].w << 16));
                } else {
                    vals[2 * at[d]] = lo[d];
                    vals[2 * at[d] + 1] = hi[d];
                }
            }
        }
    }, mycnt);
}

// ------------------------------------------------------------------------------------ pass 3
// queries: v[0..k) are the key's counters in hash order
struct QueryCmsMin {   // countminsketch.py:429-432
    using Out = int32_t;
    template <int KT>
    __device__ __forceinline__ Out operator()(const uint32_t (&v)[KT], uint32_t k) const
    {
        int32_t mn = INT32_MAX;
#pragma unroll
        for (int j = 0; j < KT; ++j)
            if ((uint32_t)j < k) mn = (int32_t)v[j] < mn ? (int32_t)v[j] : mn;
        return mn;
    }
};
struct QueryCmsMean {  // countminsketch.py:434-436
    using Out = int32_t;
    template <int KT>
    __device__ __forceinline__ Out operator()(const uint32_t (&v)[KT], uint32_t k) const
    {
        int64_t sum = 0;
#pragma unroll
        for (int j = 0; j < KT; ++j)
            if ((uint32_t)j < k) sum += (int32_t)v[j];
        return (int32_t)floordiv(sum, (int64_t)k);
    }
};
struct QueryCmsMeanMin {  // countminsketch.py:438-453
    using Out = int64_t;
    int64_t els_added, width;
    template <int KT>
    __device__ __forceinline__ Out operator()(const uint32_t (&v)[KT], uint32_t k) const
    {
        int64_t x[KT];
        bool all_zero = true;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            x[j] = (uint32_t)j < k ? (int64_t)(int32_t)v[j] : INT64_MAX;  // unused slots sort to the end
            if ((uint32_t)j < k) all_zero &= x[j] == 0;
        }
        if (all_zero) return 0;                                                    // :440-441
#pragma unroll
        for (int j = 0; j < KT; ++j)
            if ((uint32_t)j < k) x[j] = x[j] - floordiv(els_added - x[j], width - 1);  // :442-446
        sort_small(x, (uint32_t)KT);
        return (k % 2 == 0) ? floordiv(x[k / 2] + x[k / 2 - 1], 2) : x[k / 2];     // :448-452
    }
};
struct QueryCbfMin {   // countingbloom.py:166-174
    using Out = uint32_t;
    template <int KT>
    __device__ __forceinline__ Out operator()(const uint32_t (&v)[KT], uint32_t k) const
    {
        uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < KT; ++j)
            if ((uint32_t)j < k) mn = v[j] < mn ? v[j] : mn;
        return mn;
    }
};

// dynamic LDS: runinfo[B] (uint2) | stage[stage_cap] (values in the tile's sorted order) | fmt[B] bytes
// kCollectThreads: 1024 = two workgroups (tiles in flight) per CU, 512 = four (round 3 A/B: option "lookup_collect_threads")
template <class Query, int KT, int kCollectThreads>
__global__ __launch_bounds__(kCollectThreads) void k_lookup_collect(Query query, PartGeom g, uint64_t n, const uint32_t *perm, const uint2 *runinfo,
                                                                    const uint32_t *vals, const uint8_t *fmt, uint32_t stage_cap, uint32_t run_lanes,
                                                                    typename Query::Out *out)
{
    constexpr int GS = 8;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint2 *info = reinterpret_cast<uint2 *>(smem);
    uint32_t *stage = smem + 2 * g.nbuckets;
    uint8_t *fmt_lds = reinterpret_cast<uint8_t *>(stage + stage_cap);  // the per-slice value format, read once (a global read per
    const uint32_t B = g.nbuckets, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // run put a second round trip into the copy loop)
    for (uint32_t b = threadIdx.x; b < B; b += kCollectThreads) fmt_lds[b] = fmt[b];
    const uint32_t k = g.k < (uint32_t)KT ? g.k : (uint32_t)KT;
    const uint64_t ntiles = (n + g.tile - 1) / g.tile;
    // Software pipeline over tiles (a workgroup walks ~10 tiles; each tile used to be three dependent global round trips:
    // runinfo -> runs of values -> perm): the NEXT tile's runinfo is fetched into registers while this tile's runs are
    // copied, and this tile's perm[] entries are requested before the copy starts.
    constexpr int kInfoRegs = kPartMaxBuckets / kCollectThreads;  // slices per thread (<= 2048 slices)
    constexpr int kPre = 2048 / kCollectThreads;                   // keys per thread whose perm[] is prefetched (tiles <= 2048 keys)
    uint2 nxt[kInfoRegs];
#pragma unroll
    for (int r = 0; r < kInfoRegs; ++r) {
        const uint32_t b = threadIdx.x + (uint32_t)r * kCollectThreads;
        nxt[r] = (blockIdx.x < ntiles && b < B) ? runinfo[(uint64_t)blockIdx.x * B + b] : make_uint2(0, 0);
    }
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t wg = (uint32_t)(tile % g.nwg);  // the pass-1 workgroup that owned this tile names the segments
        const uint64_t base = tile * g.tile;
        const uint64_t end = base + g.tile < n ? base + g.tile : n;
        // Run descriptors (round 3: the kernel was instruction bound -- 27 M VALU + 29 M SALU wave-instructions per 10 M keys, most of them
        // the per-lane address arithmetic of the run copies: every one of a run's 16-32 lanes redid the 64-bit segment base, the room
        // clamp and the format test).  The thread that holds a slice's runinfo turns it, once, into what the copy loop needs:
        //   .x = first 16-byte unit of the run's values (16-bit format: 2 * segment base + group; 32-bit: segment base + group, in 32-byte
        //        groups) -- below 2^32, the host checks;  .y = stage offset (a multiple of 8; bit 0 = 16-bit format) << 16 | dwords to copy
#pragma unroll
        for (int r = 0; r < kInfoRegs; ++r) {
            const uint32_t b = threadIdx.x + (uint32_t)r * kCollectThreads;
            if (b < B) {
                const uint2 ri = nxt[r];
                const uint32_t cnt = ri.y & 0xFFFFu, off = ri.y >> 16;
                const uint64_t seg = seg_index(g, b, wg) * g.segcap;
                const uint32_t room = ri.x < g.segcap ? (g.segcap - ri.x) * GS : 0;  // (an overflowed run: the flag is up, the redo overwrites out[])
                const uint32_t lim = cnt < room ? cnt : room;
                const bool two = fmt_lds[b] != 0;
                info[b] = two ? make_uint2((uint32_t)(2 * seg + ri.x), ((off | 1u) << 16) | ((lim + 1) / 2))
                              : make_uint2((uint32_t)(seg + ri.x), (off << 16) | lim);
            }
        }
        PermRec<KT> pw[kPre];
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const uint64_t i = base + threadIdx.x + (uint64_t)q * kCollectThreads;
            pw[q] = perm_load<KT>(perm, i < end ? i : base);  // clamped, never branched around
        }
        __syncthreads();
        {
            const uint64_t nt = tile + gridDim.x;
#pragma unroll
            for (int r = 0; r < kInfoRegs; ++r) {
                const uint32_t b = threadIdx.x + (uint32_t)r * kCollectThreads;
                nxt[r] = (nt < ntiles && b < B) ? runinfo[nt * B + b] : make_uint2(0, 0);
            }
        }
        // ---- the tile's runs of values, back into the sorted order of pass 1's LDS stage.  A run is ~tile*k/B values, often
        // far fewer than 64: `run_lanes` (a power of two, host's choice from that mean) lanes take one run, 64 / run_lanes
        // runs ride one wave-instruction, four instructions are in flight per lane before LDS is written.  (Measured on
        // MI355X: dword copies beat 16-byte pieces here -- 49 vs 64 us per 16.7 M values -- and a layout that makes pass 2's
        // stores lane-contiguous costs pass 3 more than it saves pass 2.)
        const uint32_t rl = run_lanes, per_wave = 64u / rl, sub = lane / rl, e0 = lane % rl;
        const uint32_t stride = (kCollectThreads / 64) * per_wave;
        for (uint32_t b0 = wave * per_wave + sub; b0 < B; b0 += 4 * stride) {
            uint32_t v[4], at[4];
            bool live[4], two[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t b = b0 + (uint32_t)u * stride;
                live[u] = two[u] = false;
                if (b < B) {
                    const uint2 rd = info[b];
                    const uint32_t cnt = rd.y & 0xFFFFu, off = (rd.y >> 16) & ~1u;
                    two[u] = (rd.y >> 16) & 1u;
                    live[u] = e0 < cnt;
                    if (two[u]) {  // 16-bit values, the segment's groups packed at 16 bytes each: a lane moves TWO values (one dword);
                        // the odd tail lands on the run's pad slots
                        const uint32_t *src = vals + (uint64_t)rd.x * 4;
                        at[u] = off + 2 * e0;
                        if (live[u]) v[u] = src[e0];
                        for (uint32_t e = e0 + rl; e < cnt; e += rl) {  // longer runs
                            const uint32_t x = src[e];
                            *reinterpret_cast<uint2 *>(stage + off + 2 * e) = make_uint2(x & 0xFFFFu, x >> 16);
                        }
                    } else {
                        const uint32_t *src = vals + (uint64_t)rd.x * 8;
                        at[u] = off + e0;
                        if (live[u]) v[u] = src[e0];
                        for (uint32_t e = e0 + rl; e < cnt; e += rl) stage[off + e] = src[e];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (live[u]) {
                    if (two[u]) *reinterpret_cast<uint2 *>(stage + at[u]) = make_uint2(v[u] & 0xFFFFu, v[u] >> 16);
                    else stage[at[u]] = v[u];
                }
            }
        }
        __syncthreads();
        // ---- every key picks its k values through perm[]
        auto finish_key = [&](uint64_t i, const PermRec<KT> &rec) {
            uint32_t v[KT];
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const uint32_t pj = rec.pos(j);
                v[j] = (uint32_t)j < k ? stage[pj < stage_cap ? pj : 0] : 0u;
            }
            out[i] = query.template operator()<KT>(v, k);
        };
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const uint64_t i = base + threadIdx.x + (uint64_t)q * kCollectThreads;
            if (i < end) finish_key(i, pw[q]);
        }
        for (uint64_t i = base + threadIdx.x + (uint64_t)kPre * kCollectThreads; i < end; i += kCollectThreads)  // tiles beyond 2048 keys (k <= 5)
            finish_key(i, perm_load<KT>(perm, i));
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------ Bloom lookups, same three passes
// The keyed lookup of psk_partition.hpp (k_bloom_test) pays one scattered byte store per CLEAR bit it meets: batches of keys
// that were never inserted -- the common case of a Bloom lookup -- ran at half the rate of all-hit batches (21 vs 41 G
// keys/s).  With the return trip of the counter lookups the cost does not depend on the answers: pass 2 writes ONE byte per
// group of six probes (bit e = probe e's bit), pass 3 copies the tile's bytes (a few KB) into LDS and every key ANDs its k
// bits (bloom.py:261-272).

// bits[group] = the six tested bits of the group (slots past the run's end: unspecified)
static __global__ __launch_bounds__(kApplyThreads) void k_bloom_gather(const uint32_t *tab, uint64_t tab_words, PartGeom g,
                                                                       const uint32_t *segcnt, const uint4 *buckets, uint8_t *bits)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t b = blockIdx.x;
    const uint32_t slice_words = 1u << (g.shift - 5);
    const uint64_t w0 = (uint64_t)b * slice_words;
    const uint32_t mycnt = lane_segment_count(segcnt, g, b);
    load_slice(smem, tab, tab_words, w0, slice_words, (g.dbg & kGeomNtBit) != 0);
    __syncthreads();
    // (8 groups in flight per lane: 48 LDS words + the 8 groups stay inside the 128 VGPRs of a 1024-thread workgroup; with 12
    // the kernel spilled and ran 3x slower)
    constexpr int D = 8;
    auto field = [](const uint4 &q, int e) -> uint32_t {  // slice-local bit index e of a group (two 64-bit halves of 3 x 20 bits)
        const unsigned long long h = e < 3 ? (((unsigned long long)q.y << 32) | q.x) : (((unsigned long long)q.w << 32) | q.z);
        return (uint32_t)(h >> (20 * (e % 3))) & 0xFFFFFu;
    };
    for_each_batch_at<D>(buckets, segcnt, g, b, make_uint4(0, 0, 0, 0), [&](const uint4 (&q)[D], const uint64_t (&at)[D], const uint32_t (&)[D]) {
        uint32_t w[D][6];
#pragma unroll
        for (int d = 0; d < D; ++d)  // the LDS reads of the whole batch first (pads read a harmless in-slice word)
#pragma unroll
            for (int e = 0; e < 6; ++e) w[d][e] = smem[field(q[d], e) >> 5];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (at[d] != ~0ULL) {
                uint32_t r = 0;
#pragma unroll
                for (int e = 0; e < 6; ++e) r |= ((w[d][e] >> (field(q[d], e) & 31)) & 1u) << e;
                bits[at[d]] = (uint8_t)r;
            }
        }
    }, mycnt);
}

constexpr int kBloomCollectThreads = 1024;
// Keys per collect thread whose perm[] record is requested before the run copies: ceil(largest pass-1 tile / workgroup), at most 8.
// (Rounds 1-3 asked for 8 per thread whatever k was: for k = 7 -- 2048-key tiles, two keys per thread -- six of the eight were clamped
// re-reads of the tile's first record and 24 VGPRs of ballast.)
template <int KT>
__host__ __device__ constexpr int collect_prefetch_keys()
{
    constexpr int a = PartTile<PayBloomLookup, KT, 1024>::TILE, b = PartTile<PayBloomLookup, KT, kPartThreads>::TILE;
    constexpr int per = ((a > b ? a : b) + kBloomCollectThreads - 1) / kBloomCollectThreads;
    return per < 1 ? 1 : (per > 8 ? 8 : per);
}

// dynamic LDS: runinfo[B] (uint2) | stage bytes (one per group of the tile's sorted stage)
// (1024-thread workgroups: 256-thread ones -- more tiles in flight per CU -- measured slower, 84 vs 64 us per 10 M keys: the
// kernel is bound by the 16 bytes of perm[] per key, not by per-tile latency)

template <int KT>
__global__ __launch_bounds__(kBloomCollectThreads) void k_bloom_collect(PartGeom g, uint64_t n, const uint32_t *perm, const uint2 *runinfo, const uint8_t *bits,
                                                                   uint32_t stage_groups, uint32_t run_lanes, uint8_t *out, unsigned long long *miss_ctr)
{
    uint32_t nmiss = 0;  // keys answered "absent" (feeds the host's choice of lookup scheme)
    constexpr int GS = 6;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint2 *info = reinterpret_cast<uint2 *>(smem);
    uint8_t *stage = reinterpret_cast<uint8_t *>(smem + 2 * g.nbuckets);
    const uint32_t B = g.nbuckets, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t k = g.k < (uint32_t)KT ? g.k : (uint32_t)KT;
    const uint64_t ntiles = (n + g.tile - 1) / g.tile;
    constexpr int kInfoRegs = kPartMaxBuckets / kBloomCollectThreads;
    constexpr int kPre = collect_prefetch_keys<KT>();  // keys per thread whose perm[] is prefetched: the whole tile
    uint2 nxt[kInfoRegs];
#pragma unroll
    for (int r = 0; r < kInfoRegs; ++r) {
        const uint32_t b = threadIdx.x + (uint32_t)r * kBloomCollectThreads;
        nxt[r] = (blockIdx.x < ntiles && b < B) ? runinfo[(uint64_t)blockIdx.x * B + b] : make_uint2(0, 0);
    }
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t wg = (uint32_t)(tile % g.nwg);
        const uint64_t base = tile * g.tile;
        const uint64_t end = base + g.tile < n ? base + g.tile : n;
#pragma unroll
        for (int r = 0; r < kInfoRegs; ++r) {
            const uint32_t b = threadIdx.x + (uint32_t)r * kBloomCollectThreads;
            if (b < B) info[b] = nxt[r];
        }
        PermRec<KT> pw[kPre];
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const uint64_t i = base + threadIdx.x + (uint64_t)q * kBloomCollectThreads;
            pw[q] = perm_load<KT>(perm, i < end ? i : base);
        }
        __syncthreads();
        {
            const uint64_t nt = tile + gridDim.x;
#pragma unroll
            for (int r = 0; r < kInfoRegs; ++r) {
                const uint32_t b = threadIdx.x + (uint32_t)r * kBloomCollectThreads;
                nxt[r] = (nt < ntiles && b < B) ? runinfo[nt * B + b] : make_uint2(0, 0);
            }
        }
        // ---- the tile's result bytes (one per group of six probes) into the order of pass 1's sorted stage
        const uint32_t rl = run_lanes, per_wave = 64u / rl, sub = lane / rl, e0 = lane % rl;
        const uint32_t stride = (kBloomCollectThreads / 64) * per_wave;
        for (uint32_t b = wave * per_wave + sub; b < B; b += stride) {
            const uint2 ri = info[b];
            const uint32_t groups = ((ri.y & 0xFFFFu) + GS - 1) / GS, off_g = (ri.y >> 16) / GS;
            const uint64_t src = seg_index(g, b, wg) * g.segcap + ri.x;
            const uint32_t room = ri.x < g.segcap ? g.segcap - ri.x : 0;  // (an overflowed run: the flag is up, the redo overwrites out[])
            const uint32_t lim = groups < room ? groups : room;
            for (uint32_t e = e0; e < lim; e += rl) stage[off_g + e] = bits[src + e];
        }
        __syncthreads();
        // ---- every key ANDs its k bits (bloom.py:269-271)
        auto finish_key = [&](uint64_t i, const PermRec<KT> &rec) {
            uint32_t ok = 1;
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                if ((uint32_t)j < k) {
                    const uint32_t pj = rec.pos(j);
                    const uint32_t gi = pj / GS, e = pj - gi * GS;
                    ok &= ((uint32_t)stage[gi < stage_groups ? gi : 0] >> e) & 1u;
                }
            }
            out[i] = (uint8_t)ok;
            nmiss += ok ^ 1u;
        };
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const uint64_t i = base + threadIdx.x + (uint64_t)q * kBloomCollectThreads;
            if (i < end) finish_key(i, pw[q]);
        }
        for (uint64_t i = base + threadIdx.x + (uint64_t)kPre * kBloomCollectThreads; i < end; i += kBloomCollectThreads)
            finish_key(i, perm_load<KT>(perm, i));
        __syncthreads();
    }
    if (miss_ctr) {  // ONE atomic per workgroup: same-address device atomics serialise (~11 ns each; one per wave cost 90 us here)
        for (int o = 32; o > 0; o >>= 1) nmiss += __shfl_down(nmiss, o);
        uint32_t *part = smem;  // (the tile loop is over: LDS is free)
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = nmiss;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < kBloomCollectThreads / 64; ++w) t += part[w];
            if (t) atomicAdd(miss_ctr, t);
        }
    }
}

// the tally of a finished lookup -> the pinned host page the next call's choice of scheme reads (one consistent triple)
// (and zeroes the tally for the next call: no per-call memset)
static __global__ void k_lookup_publish(unsigned long long *tally, volatile unsigned long long *pin, unsigned long long units, unsigned long long scheme)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        pin[1] = units;
        pin[2] = scheme;
        pin[0] = tally[0];
        pin[3] = pin[3] + 1;
        tally[0] = 0;
    }
}

}  // namespace psk
