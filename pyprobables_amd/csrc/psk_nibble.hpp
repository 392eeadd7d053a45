// psk_nibble.hpp -- CountingBloomFilter tables beyond one level of 32-bit LDS slices: 4-bit slice images (round 3).
#pragma once
#include "psk_lookup.hpp"

namespace psk {

// ------------------------------------------------------------------------------------ CountingBloomFilter, NIBBLE slices
// Counter tables beyond 2048 x 2^15 cells (the 1 GiB table of BASELINE cfg 4: 2^28 counters) cannot be cut into 32-bit LDS
// slices in one level, and the direct lookup fetches one 64-byte line per 4-byte counter (1.3 ms per 10 M keys, 503 B/key).
// The counters of a CountingBloomFilter are tiny by construction (k * n / m per cell: 0.7 at capacity), and the lookup is a
// MIN (countingbloom.py:166-174): min_j min(c_j, 15) == min(min_j c_j, 15), so a 4-bit saturating image answers exactly
// whenever the answer is below 15.  A slice is 2^18 counters held as nibbles (128 KiB): 1024 slices for 2^28 counters, one
// level, the 6 x 20-bit probe groups of the Bloom lookups.  A key whose min comes out as 15 raises the redo flag and the
// k_cbf_recheck15 answers exactly those keys from the table itself (exact for any table).
//   pass 2  k_nib_gather   loads the slice (coalesced, 1 MiB), packs it to nibbles, streams the slice's probe groups and
//                          writes ONE dword per group: nibble e = min(counter of probe e, 15)
//   pass 3  k_nib_collect  as k_bloom_collect with dword runs: every key takes the min of its k nibbles


// slice-local index e of a 6 x 20-bit probe group (two 64-bit halves of 3 x 20 bits; see PayNone)
__device__ __forceinline__ uint32_t group_field(const uint4 &q, int e)
{
    const unsigned long long h = e < 3 ? (((unsigned long long)q.y << 32) | q.x) : (((unsigned long long)q.w << 32) | q.z);
    return (uint32_t)(h >> (20 * (e % 3))) & 0xFFFFFu;
}

// Image layout: word w holds the counters 8w .. 8w+7 of the slice, nibble e = counter 8w + e; as 16-bit halves it is one half
// per 16-byte PIECE of the table (4 counters), so lane t of a load / store instruction handles piece p0 + t: every
// wave-instruction covers 1 KiB of contiguous table.  (Measured on the 1 GiB table, per pass over it: a lane packing two
// neighbouring pieces itself -- 32 B per lane, every instruction touching half of each 64-byte line -- 322 us for the lookups'
// pass 2 and 697 us for the adds' fold; 8192-counter blocks with the two pieces of a lane 16 KiB apart 361 / 613 us.)
__device__ __forceinline__ uint32_t nib_word(uint32_t cell) { return cell >> 3; }
__device__ __forceinline__ uint32_t nib_bit(uint32_t cell) { return (cell & 7u) << 2; }  // bit offset of the nibble

// 4 counters of the table (one 16-byte piece) -> one 16-bit half of an image word (saturating at 15)
__device__ __forceinline__ uint32_t nib_pack4(const uint4 &a)
{
    auto n = [](uint32_t c) -> uint32_t { return c < 15u ? c : 15u; };
    return n(a.x) | (n(a.y) << 4) | (n(a.z) << 8) | (n(a.w) << 12);
}

typedef unsigned int psk_u32x4 __attribute__((ext_vector_type(4)));

// the 16-byte piece of table cells [gc, gc + 4); cells at or beyond tab_cells read as 0 (the table is padded to whole pieces)
// nt: nontemporal load -- the pass over a table far larger than the 256 MB Infinity Cache should not push the probe lists out of it
__device__ __forceinline__ uint4 nib_load_piece(const uint32_t *tab, uint64_t tab_cells, uint64_t gc, bool nt = false)
{
    if (gc + 3 < tab_cells) {
        if (nt) {
            const psk_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const psk_u32x4 *>(tab + gc));
            return make_uint4(v.x, v.y, v.z, v.w);
        }
        return *reinterpret_cast<const uint4 *>(tab + gc);
    }
    uint4 t = make_uint4(0, 0, 0, 0);
    if (gc + 0 < tab_cells) t.x = tab[gc + 0];
    if (gc + 1 < tab_cells) t.y = tab[gc + 1];
    if (gc + 2 < tab_cells) t.z = tab[gc + 2];
    return t;
}

// loads slice b of the table into the LDS nibble image (2^(shift-3) words)
__device__ __forceinline__ void nib_load_slice(uint32_t *smem, const uint32_t *tab, uint64_t tab_cells, uint32_t shift, uint32_t b, bool nt)
{
    const uint32_t pieces = 1u << (shift - 2);
    const uint64_t c0 = (uint64_t)b << shift;
    uint16_t *half = reinterpret_cast<uint16_t *>(smem);
    constexpr int U = 8;  // 16-byte loads in flight per lane
    for (uint32_t p0 = threadIdx.x; p0 < pieces; p0 += kApplyThreads * U) {
        uint4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t pc = p0 + (uint32_t)u * kApplyThreads;
            t[u] = pc < pieces ? nib_load_piece(tab, tab_cells, c0 + 4ULL * pc, nt) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t pc = p0 + (uint32_t)u * kApplyThreads;
            if (pc < pieces) half[pc] = (uint16_t)nib_pack4(t[u]);
        }
    }
}

// shadow_in: the slice images as an earlier launch left them (psk_sketch::shadow: a linear 4-bit copy of the table, valid while the
// table is unchanged) -- 1/8 of the bytes; shadow_out: leave them behind for the next lookup.  Both null: build and forget.
static __global__ __launch_bounds__(kApplyThreads) void k_nib_gather(const uint32_t *tab, uint64_t tab_cells, PartGeom g, const uint32_t *segcnt,
                                                                     const uint4 *buckets, uint32_t *vals, uint32_t nt, const uint32_t *shadow_in,
                                                                     uint32_t *shadow_out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t b = blockIdx.x;
    const uint32_t mycnt = lane_segment_count(segcnt, g, b);
    const uint32_t vecs = 1u << (g.shift - 5);  // 16-byte pieces of one image (2^(shift-3) words)
    if (shadow_in) {
        const uint4 *src = reinterpret_cast<const uint4 *>(shadow_in) + (uint64_t)b * vecs;
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        constexpr int U = 8;
        for (uint32_t p0 = threadIdx.x; p0 < vecs; p0 += kApplyThreads * U) {
            uint4 t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pc = p0 + (uint32_t)u * kApplyThreads;
                t[u] = pc < vecs ? src[pc] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pc = p0 + (uint32_t)u * kApplyThreads;
                if (pc < vecs) dst[pc] = t[u];
            }
        }
    } else {
        nib_load_slice(smem, tab, tab_cells, g.shift, b, nt != 0);
    }
    __syncthreads();
    if (shadow_out) {  // (the stores drain under the probe stream below)
        uint4 *dst = reinterpret_cast<uint4 *>(shadow_out) + (uint64_t)b * vecs;
        const uint4 *src = reinterpret_cast<const uint4 *>(smem);
        for (uint32_t pc = threadIdx.x; pc < vecs; pc += kApplyThreads) dst[pc] = src[pc];
    }
    constexpr int D = 8;  // (48 LDS words + 8 groups per lane stay inside the 128 VGPRs of a 1024-thread workgroup, see k_bloom_gather)
    for_each_batch_at<D>(buckets, segcnt, g, b, make_uint4(0, 0, 0, 0), [&](const uint4 (&q)[D], const uint64_t (&at)[D], const uint32_t (&)[D]) {
        uint32_t w[D][6];
#pragma unroll
        for (int d = 0; d < D; ++d)  // the LDS reads of the whole batch first (slots past a run's end read counter 0: harmless)
#pragma unroll
            for (int e = 0; e < 6; ++e) w[d][e] = smem[nib_word(group_field(q[d], e))];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (at[d] != ~0ULL) {
                uint32_t r = 0;
#pragma unroll
                for (int e = 0; e < 6; ++e) r |= ((w[d][e] >> nib_bit(group_field(q[d], e))) & 15u) << (4 * e);
                vals[at[d]] = r;
            }
        }
    }, mycnt);
}

// dynamic LDS: runinfo[B] (uint2) | stage words (one per group of the tile's sorted stage)
template <int KT>
__global__ __launch_bounds__(kBloomCollectThreads) void k_nib_collect(PartGeom g, uint64_t n, const uint32_t *perm, const uint2 *runinfo, const uint32_t *vals,
                                                                 uint32_t stage_groups, uint32_t run_lanes, uint32_t *out, uint32_t *flag)
{
    constexpr int GS = 6;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint2 *info = reinterpret_cast<uint2 *>(smem);
    uint32_t *stage = smem + 2 * g.nbuckets;
    const uint32_t B = g.nbuckets, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t k = g.k < (uint32_t)KT ? g.k : (uint32_t)KT;
    const uint64_t ntiles = (n + g.tile - 1) / g.tile;
    constexpr int kInfoRegs = kPartMaxBuckets / kBloomCollectThreads;
    constexpr int kPre = collect_prefetch_keys<KT>();  // keys per thread whose perm[] is prefetched (see k_bloom_collect)
    uint2 nxt[kInfoRegs];
#pragma unroll
    for (int r = 0; r < kInfoRegs; ++r) {
        const uint32_t b = threadIdx.x + (uint32_t)r * kBloomCollectThreads;
        nxt[r] = (blockIdx.x < ntiles && b < B) ? runinfo[(uint64_t)blockIdx.x * B + b] : make_uint2(0, 0);
    }
    bool ambiguous = false;
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
        // ---- the tile's value words (one per group of six probes) into the order of pass 1's sorted stage
        const uint32_t rl = run_lanes, per_wave = 64u / rl, sub = lane / rl, e0 = lane % rl;
        const uint32_t stride = (kBloomCollectThreads / 64) * per_wave;
        for (uint32_t b = wave * per_wave + sub; b < B; b += stride) {
            const uint2 ri = info[b];
            const uint32_t groups = ((ri.y & 0xFFFFu) + GS - 1) / GS, off_g = (ri.y >> 16) / GS;
            const uint64_t src = seg_index(g, b, wg) * g.segcap + ri.x;
            const uint32_t room = ri.x < g.segcap ? g.segcap - ri.x : 0;  // (an overflowed run: the flag is up, the redo overwrites out[])
            const uint32_t lim = groups < room ? groups : room;
            for (uint32_t e = e0; e < lim; e += rl) stage[off_g + e] = vals[src + e];
        }
        __syncthreads();
        // ---- every key takes the min of its k nibbles (countingbloom.py:174)
        auto finish_key = [&](uint64_t i, const PermRec<KT> &rec) {
            uint32_t mn = 15;
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                if ((uint32_t)j < k) {
                    const uint32_t pj = rec.pos(j);
                    const uint32_t gi = pj / GS, e = pj - gi * GS;
                    const uint32_t v = (stage[gi < stage_groups ? gi : 0] >> (4 * e)) & 15u;
                    mn = v < mn ? v : mn;
                }
            }
            out[i] = mn;
            ambiguous |= mn == 15u;  // every counter of the key is 15 or more: the 4-bit image cannot tell -- exact redo
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
    if (ambiguous) *flag = 1u;
}

// The keys k_nib_collect could not answer (result 15: every counter of the key is 15 or more), one by one from the table itself; runs
// only when the ambiguity flag is up.  A batch with one heavy hitter then costs one sweep over out[] and k gathers for that key -- not
// a second, direct lookup of the whole batch (round 3, first version: a 10 M-key lookup into 9.6e7 counters whose keys had been added
// six times each fell from 13.4 to 5.4 G keys/s because 35 of the keys had a min of 18).
template <class Src, bool POW2>
__global__ __launch_bounds__(kBlock) void k_cbf_recheck15(const uint32_t *amb, Src src, const uint32_t *tab, Mod md, uint32_t kk, uint64_t n, uint32_t *out)
{
    if (*amb == 0) return;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        if (out[i] != 15u) continue;
        const typename Src::Key key = src.load(i);
        uint32_t mn = 0xFFFFFFFFu;
        for_each_hash(src, key, i, kk, [&](uint32_t, uint64_t h) {
            const uint32_t v = tab[reduce<POW2>(md, h)];
            mn = v < mn ? v : mn;
        });
        out[i] = mn;
    }
}

// ------------------------------------------------------------------------------------ validated remove, OPTIMISTIC decrement
// countingbloom.py:186-208 removes a key only if the min of its k counters is non-zero (and not frozen).  As a batch that is a
// lookup of every key (a return trip: three passes) followed by the decrement of the keys that passed.  For the batches one
// meets in practice -- every key present -- a much cheaper test suffices: with R[c] = how often the batch's probes hit counter c,
//      T[c] >= R[c] for every counter c (none of them frozen)   ==>   every key of the batch is removed, in any order
// (whatever was removed before key X, each of X's counters still holds T[c] - (R[c] - 1) >= 1 just before X's own decrement, so
// its min is >= 1 and to_remove = num_els = 1), and the result is T - R.  The fold of k_nib_apply sees T[c] and R[c] side by
// side, so it checks the condition WHILE it decrements (OPT = 1: wrapping subtraction, a device flag for t < d or a frozen
// counter).  Flag clear: done -- one pass 1 and one pass over the table.  Flag up: the same probe groups are added back (OPT = 2:
// wrapping addition, the exact inverse) and the exact path (lookup, amounts, masked decrement) takes the batch.  A dry run that
// only checked (a saturating image of the table + one returning ds_sub per probe) measured 403 us per 10 M keys on top of the
// 509 us decrement; the optimistic form has no extra pass at all.
// ------------------------------------------------------------------------------------ unit adds / decrements, NIBBLE deltas
// The same slices for UPDATES with unit weights (add_many(keys) / the validated remove's decrement, countingbloom.py:135-155,
// :203-206): the LDS image holds 4-bit DELTAS (how often the round hits each counter), the fold adds them to the table with
// the reference's saturation.  One level and no k_part_split for 2^28 counters (the 32-bit images needed 8192 slices: coarse
// buckets + a second split, 150 us per 10 M keys).  A counter hit 16 times or more in one round would carry into its
// neighbour: every hit is a RETURNING ds_add whose old nibble tells (15 -> overflow); the workgroup then drops its image and
// applies its probe groups with exact saturating atomics on the table -- the slice is its alone -- so the result is exact for
// any batch (duplicate-heavy ones just run that slice at the direct rate).
// The DELTA image of the update kernel comes in two layouts (bench A/B, option "nibble_update_layout"): 0 = as above (one 16-bit
// half per 16-byte piece); 1 = blocks of 8192 counters, word (block, t) = the pieces t and 1024 + t of the block, so that a lane
// folds two pieces 16 KiB apart per image word.
template <bool BLOCKS>
__device__ __forceinline__ uint32_t dlt_word(uint32_t cell) { return BLOCKS ? (((cell >> 13) << 10) | ((cell >> 2) & 1023u)) : (cell >> 3); }
template <bool BLOCKS>
__device__ __forceinline__ uint32_t dlt_bit(uint32_t cell) { return BLOCKS ? (((cell >> 8) & 16u) | ((cell & 3u) << 2)) : ((cell & 7u) << 2); }

// one list (segcnt, buckets) into slice b; `direct`: skip the image and apply every probe with an atomic on the table (few probes:
// a pass over the whole slice would cost more)
// OPT 0: the reference's saturating add / checked decrement; 1 (NEG): optimistic decrement, see above; 2 (!NEG): its inverse
// Round 4: a slice may be shared by 2^lgp workgroups (PARTS): workgroup h of a slice owns the counters [h << pshift, (h + 1) << pshift), pshift =
// shift - lgp -- it streams ALL probe groups of the slice, applies those of its part and folds its part of the table.  With 2^17-counter
// parts the delta image is 64 KiB, two workgroups fit one CU, and one of them streams probes (LDS-atomic bound) while the other folds
// (HBM bound): the pass over a 1 GiB table was 3.0-3.7 TB/s with one 128 KiB workgroup per CU, whose phases only follow each other; the
// probe groups are a tenth of the bytes, reading them twice costs little.
constexpr int kNibDepth = 6;  // probe groups in flight per lane in k_nib_apply (two workgroups per CU share the latency hiding)
template <bool NEG, bool BLOCKS, int OPT = 0>
__device__ __forceinline__ void nib_apply_list(uint32_t *smem, uint32_t *carried, uint32_t *tab, uint64_t tab_cells, const PartGeom &g, const uint32_t *segcnt,
                                               const uint4 *buckets, unsigned long long *sat_ctr, uint32_t b, bool direct, bool nt, uint32_t *flag = nullptr,
                                               uint32_t lgp = 0, uint32_t h = 0)
{
    uint32_t bad = 0;
    const uint32_t pshift = g.shift - lgp, pmask = (1u << pshift) - 1;
    const uint32_t pieces = 1u << (pshift - 2);
    const uint64_t c0 = ((uint64_t)b << g.shift) + ((uint64_t)h << pshift);
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    if (!direct) {
        for (uint32_t w = threadIdx.x; w < (1u << (pshift - 3)); w += kApplyThreads) smem[w] = 0;
        if (threadIdx.x == 0) *carried = 0;
        __syncthreads();
        uint32_t over = 0;
        const uint32_t hpart = h;
        auto half = [&](uint32_t lo, uint32_t hi) {  // 3 x 20-bit slice-local indices, valid count in bits 60..63
            const unsigned long long h = ((unsigned long long)hi << 32) | lo;
            const uint32_t nv = hi >> 28;
            const uint32_t xx[3] = {(uint32_t)h & 0xFFFFFu, (uint32_t)(h >> 20) & 0xFFFFFu, (uint32_t)(h >> 40) & 0xFFFFFu};
            uint32_t old[3] = {0, 0, 0}, x[3];
            bool mine[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                mine[e] = (uint32_t)e < nv && (xx[e] >> pshift) == hpart;
                x[e] = xx[e] & pmask;
            }
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if (mine[e]) old[e] = atomicAdd(&smem[dlt_word<BLOCKS>(x[e])], 1u << dlt_bit<BLOCKS>(x[e]));  // ds_add_rtn_u32
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if (mine[e]) over |= (uint32_t)(((old[e] >> dlt_bit<BLOCKS>(x[e])) & 15u) == 15u);
        };
        for_each_group<kNibDepth>(buckets, segcnt, g, b, zero4, [&](const uint4 q) { half(q.x, q.y); half(q.z, q.w); });
        if (over) *carried = 1u;
        __syncthreads();
    }
    if (direct || *carried) {  // uniform: exact atomics for this slice, straight from its probe groups
        const uint32_t hpart2 = h;
        auto slow = [&](uint32_t lo, uint32_t hi) {
            const unsigned long long h = ((unsigned long long)hi << 32) | lo;
            const uint32_t nv = hi >> 28;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const uint32_t xe = (uint32_t)(h >> (20 * e)) & 0xFFFFFu;
                if ((uint32_t)e < nv && (xe >> pshift) == hpart2) {
                    const uint64_t cell = c0 + (xe & pmask);
                    if (OPT == 1) {
                        const uint32_t old = atomicSub(tab + cell, 1u);
                        bad |= (uint32_t)(old == 0u) | (uint32_t)(old == 0xFFFFFFFFu);
                    } else if (OPT == 2) {
                        atomicAdd(tab + cell, 1u);
                    } else if (NEG) cbf_sat_sub(tab + cell, 1u, sat_ctr - 1);
                    else cbf_sat_add(tab + cell, 1u, sat_ctr);
                }
            }
        };
        for_each_group<kNibDepth>(buckets, segcnt, g, b, zero4, [&](const uint4 q) { slow(q.x, q.y); slow(q.z, q.w); });
        if (OPT == 1 && bad) *flag = 1u;
        return;
    }
    unsigned long long sat = 0, viol = 0;
    auto fold = [&](uint32_t t, uint32_t d) -> uint32_t {
        if (d == 0) return t;
        if (OPT == 1) {
            bad |= (uint32_t)(t < d) | (uint32_t)(t == 0xFFFFFFFFu);
            return t - d;
        }
        if (OPT == 2) return t + d;
        if (NEG) {  // countingbloom.py:203-206: a counter frozen at 2^32-1 stays; below zero = the stream was not well-formed
            if (t == 0xFFFFFFFFu) return t;
            if (t < d) { ++viol; return 0u; }
            return t - d;
        }
        const uint64_t v = (uint64_t)t + d;  // countingbloom.py:149-153
        if (v > 0xFFFFFFFFULL) { ++sat; return 0xFFFFFFFFu; }
        return (uint32_t)v;
    };
    auto fold4 = [&](uint4 t, uint32_t d16) -> uint4 {
        return make_uint4(fold(t.x, d16 & 15u), fold(t.y, (d16 >> 4) & 15u), fold(t.z, (d16 >> 8) & 15u), fold(t.w, (d16 >> 12) & 15u));
    };
    auto store_piece = [&](uint64_t gc, const uint4 &o) {
        if (gc + 3 < tab_cells) { *reinterpret_cast<uint4 *>(tab + gc) = o; return; }
        if (gc + 0 < tab_cells) tab[gc + 0] = o.x;  // the table ends inside this piece
        if (gc + 1 < tab_cells) tab[gc + 1] = o.y;
        if (gc + 2 < tab_cells) tab[gc + 2] = o.z;
    };
    if constexpr (BLOCKS) {
        const uint32_t blocks = 1u << (pshift - 13);
        constexpr int U = 2;  // blocks (two 16-byte pieces per lane each) in flight (two workgroups per CU: 64 VGPRs per lane)
        for (uint32_t k0 = 0; k0 < blocks; k0 += U) {
            uint32_t d[U];
            uint4 t[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {  // (slices of 2^15 counters and more: blocks is a multiple of U = 4)
                const uint64_t gc = c0 + (uint64_t)(k0 + u) * 8192u + 4u * threadIdx.x;
                d[u] = smem[(k0 + u) * 1024u + threadIdx.x];
                t[u][0] = t[u][1] = zero4;
                if (d[u] & 0xFFFFu) t[u][0] = nib_load_piece(tab, tab_cells, gc, nt);
                if (d[u] >> 16) t[u][1] = nib_load_piece(tab, tab_cells, gc + 4096u, nt);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t gc = c0 + (uint64_t)(k0 + u) * 8192u + 4u * threadIdx.x;
                if (d[u] & 0xFFFFu) store_piece(gc, fold4(t[u][0], d[u] & 0xFFFFu));
                if (d[u] >> 16) store_piece(gc + 4096u, fold4(t[u][1], d[u] >> 16));
            }
        }
    } else {
        const uint16_t *half16 = reinterpret_cast<const uint16_t *>(smem);
        constexpr int U = 2;  // 16-byte pieces in flight per lane
        for (uint32_t p0 = threadIdx.x; p0 < pieces; p0 += kApplyThreads * U) {
            uint32_t d[U];
            uint4 t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pc = p0 + (uint32_t)u * kApplyThreads;
                d[u] = pc < pieces ? (uint32_t)half16[pc] : 0u;
                t[u] = zero4;
                if (d[u]) t[u] = nib_load_piece(tab, tab_cells, c0 + 4ULL * pc, nt);  // untouched pieces are neither read nor written
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pc = p0 + (uint32_t)u * kApplyThreads;
                if (d[u]) store_piece(c0 + 4ULL * pc, fold4(t[u], d[u]));
            }
        }
    }
    if (sat) atomicAdd(sat_ctr, sat);
    if (viol) atomicAdd(sat_ctr - 1, viol);
    if (OPT == 1 && bad) *flag = 1u;
}

// MODE 0: the list is adds; 1: decrements; 2: list A adds, THEN list B decrements (one launch for a write-combined flush: the
// slice a workgroup has just folded is still on-die when it folds it again); 3: optimistic decrement (flag), 4: its inverse.
// direct: see nib_apply_list.
template <int MODE, bool BLOCKS>
__global__ __launch_bounds__(kApplyThreads, 8) void k_nib_apply(uint32_t *tab, uint64_t tab_cells, PartGeom g, const uint32_t *segcnt_a, const uint4 *buckets_a,
                                                             const uint32_t *segcnt_b, const uint4 *buckets_b, unsigned long long *sat_ctr, uint32_t direct,
                                                             uint32_t *flag, PartGeom gb)
{
    // gb: geometry of list B (MODE 2; same slices as g, its own workgroup count / segment capacity)
    // direct: bit 0 = atomics instead of the image, bit 1 = nontemporal table loads, bits 8.. = log2(workgroups per slice), see nib_apply_list
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t carried;
    const uint32_t lgp = direct >> 8;
    const uint32_t b = blockIdx.x >> lgp, h = blockIdx.x & ((1u << lgp) - 1u);
    if (((uint64_t)b << g.shift) + ((uint64_t)h << (g.shift - lgp)) >= tab_cells) return;  // (a part past the table's end)
    if (MODE == 0 || MODE == 2) nib_apply_list<false, BLOCKS>(smem, &carried, tab, tab_cells, g, segcnt_a, buckets_a, sat_ctr, b, (direct & 1u) != 0, (direct & 2u) != 0, nullptr, lgp, h);
    if (MODE == 2) {
        __threadfence();   // my stores to the slice are visible to my loads below (same CU, but through L2: not the L1)
        __syncthreads();
    }
    if (MODE == 1) nib_apply_list<true, BLOCKS>(smem, &carried, tab, tab_cells, g, segcnt_a, buckets_a, sat_ctr, b, (direct & 1u) != 0, (direct & 2u) != 0, nullptr, lgp, h);
    if (MODE == 2) nib_apply_list<true, BLOCKS>(smem, &carried, tab, tab_cells, gb, segcnt_b, buckets_b, sat_ctr, b, (direct & 1u) != 0, (direct & 2u) != 0, nullptr, lgp, h);
    if (MODE == 3) nib_apply_list<true, BLOCKS, 1>(smem, &carried, tab, tab_cells, g, segcnt_a, buckets_a, sat_ctr, b, (direct & 1u) != 0, (direct & 2u) != 0, flag, lgp, h);
    if (MODE == 4) nib_apply_list<false, BLOCKS, 2>(smem, &carried, tab, tab_cells, g, segcnt_a, buckets_a, sat_ctr, b, (direct & 1u) != 0, (direct & 2u) != 0, flag, lgp, h);
}


// segcnt[] of a persistent list back to zero after a flush (one launch for both lists)
static __global__ __launch_bounds__(256) void k_zero_u32(uint32_t *a, uint64_t na, uint32_t *b, uint64_t nb)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += stride) {
        if (i < na) a[i] = 0;
        else b[i - na] = 0;
    }
}

}  // namespace psk
