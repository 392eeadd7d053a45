// psk_nibble_pipe.hpp -- the pass over a big CountingBloomFilter table as a software pipeline (round 4): k_nib_apply's successor for the
// default delta-image layout.  Included by the update launchers only (psk_part_counter.hpp).
#pragma once
#include "psk_nibble.hpp"

#include <type_traits>

namespace psk {

template <int I, int N, int S, class F>
__device__ __forceinline__ void pipe_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pipe_static_for<I + S, N, S>(f);
    }
}

// ------------------------------------------------------------------------------------ round 4: the table pass, pipelined
// k_nib_apply runs one workgroup per CU (the 128 KiB delta image), so a slice's three phases follow each other: zero the image, stream the
// slice's probe groups into it (LDS atomics, ~10 % of the bytes), fold 1 MiB of table (HBM: read + write).  scripts/ubench/tabpass.hip puts
// the fold's floor -- 2 GiB read and written in place in this very shape -- at 370-400 us; the kernel took 582 us per 10 M keys because the
// probe phases (~150 us in all) ran with the memory pipe idle.  Here a PERSISTENT workgroup walks its slices and hands the finished delta
// image over to REGISTERS (2^15 words / 1024 threads = 32 VGPRs per lane): the fold of slice s then runs out of registers while the probe
// groups of the workgroup's next slice already go into the (zeroed) image -- both in one instruction stream per wave, fold loads and probe
// group loads in flight together, the returning LDS atomics issued under the fold's memory round trips.

// A slice's probe groups, D per lane and step, resumable (the walk of for_each_batch as an iterator: same chunked / dense numbering).
template <int D>
struct SegWalk {
    const uint4 *buckets;
    uint32_t b, nwg, segcap, nbuckets, dense;
    uint32_t mycnt, incl, excl, total, pos;
    __device__ __forceinline__ void init(const uint4 *bk, const uint32_t *segcnt, const PartGeom &g, uint32_t bb)
    {
        const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        buckets = bk; b = bb; nwg = g.nwg; segcap = g.segcap; nbuckets = g.nbuckets; dense = g.dense;
        const uint32_t nseg = g.nwg > wave ? (g.nwg - wave + kApplyWaves - 1) / kApplyWaves : 0;  // <= 64
        mycnt = lane < nseg ? segcnt[(uint64_t)b * g.nwg + wave + kApplyWaves * lane] : 0u;
        const uint32_t unit = dense ? mycnt : (mycnt + 63) >> 6;
        incl = wave_inclusive_scan(unit);
        excl = incl - unit;
        total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        pos = 0;
    }
    __device__ __forceinline__ void none() { total = pos = 0; }
    __device__ __forceinline__ bool more() const { return pos < total; }  // wave-uniform
    __device__ __forceinline__ uint64_t seg_base(uint32_t seg) const { return ((uint64_t)(threadIdx.x >> 6) + (uint64_t)kApplyWaves * seg) * nbuckets + b; }
    __device__ __forceinline__ void next(uint4 (&q)[D], const uint4 pad)
    {
        uint32_t at[D];
        next_at(q, at, pad);
    }
    // at[d]: the group's flat index in the bucket buffer (all ones = no group: q[d] holds `pad`); 32 bits -- a round's groups are far
    // below 2^32 (psk_part_lookup.hpp lookup_round_keys; the update rounds are smaller still)
    __device__ __forceinline__ void next_at(uint4 (&q)[D], uint32_t (&at)[D], const uint4 pad)
    {
        const uint32_t lane = threadIdx.x & 63;
        if (dense) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const uint32_t row = pos + 64u * (uint32_t)d;
                const uint32_t G = row + lane;
                uint32_t t = (uint32_t)__builtin_popcountll(__ballot(incl <= row));
                const uint32_t sl = t < 64 ? t : 63;
                uint32_t seg = sl;
                uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)excl, sl);
                for (; t < 64; ++t) {
                    const uint32_t end_t = (uint32_t)__builtin_amdgcn_readlane((int)incl, t);
                    if (end_t > row + 63u) break;
                    const bool past = G >= end_t;
                    seg = past ? t + 1 : seg;
                    first = past ? end_t : first;
                }
                seg = seg < 64 ? seg : 63;
                q[d] = pad;
                at[d] = 0xFFFFFFFFu;
                if (G < total) {
                    at[d] = (uint32_t)seg_base(seg) * segcap + (G - first);
                    q[d] = buckets[at[d]];
                }
            }
            pos += 64u * D;
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const uint32_t cc = pos + (uint32_t)d;
                const uint32_t seg = (uint32_t)__builtin_popcountll(__ballot(incl <= cc));  // uniform; == 64 past the end
                const uint32_t sl = seg < 64 ? seg : 63;
                const uint32_t v = (cc - (uint32_t)__builtin_amdgcn_readlane((int)excl, sl)) * 64 + lane;
                const uint32_t cnt = seg < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)mycnt, sl) : 0;
                q[d] = pad;
                at[d] = 0xFFFFFFFFu;
                if (v < cnt) {
                    at[d] = (uint32_t)seg_base(sl) * segcap + v;
                    q[d] = buckets[at[d]];
                }
            }
            pos += D;
        }
    }
};

constexpr int kPipeDepth = 4;     // probe groups in flight per lane
constexpr int kPipeWords = 32;    // image words per lane at most: slices of 2^18 counters
// MODE 0: adds; 1: decrements; 3: optimistic decrement (flag); 4: its inverse -- as k_nib_apply's.  BLOCKS layout of the delta image
// (dlt_word<true> / dlt_bit<true>): word (block k, lane t) = the pieces t and 1024 + t of block k (8192 counters).
// Grid: min(slices, CUs) persistent workgroups; slice b, b + gridDim.x, ...
template <int MODE>
__global__ __launch_bounds__(kApplyThreads, 4) void k_nib_apply_pipe(uint32_t *tab, uint64_t tab_cells, PartGeom g, const uint32_t *segcnt, const uint4 *buckets,
                                                                    unsigned long long *sat_ctr, uint32_t nt, uint32_t *flag)
{
    constexpr bool NEG = MODE == 1 || MODE == 3;
    constexpr int OPT = MODE == 3 ? 1 : (MODE == 4 ? 2 : 0);
    constexpr int D = kPipeDepth;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t carried[2];
    const uint32_t words = 1u << (g.shift - 3);
    const uint32_t nk = words / kApplyThreads;  // image words per lane = blocks of 8192 counters per slice (4 .. 32)
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    uint32_t bad = 0, over = 0;
    unsigned long long sat = 0, viol = 0;

    // one probe group into the image: returning LDS atomics, an old nibble of 15 = the counter is hit a 16th time this round
    auto half = [&](uint32_t lo, uint32_t hi) {
        const unsigned long long h = ((unsigned long long)hi << 32) | lo;
        const uint32_t nv = hi >> 28;
        const uint32_t x[3] = {(uint32_t)h & 0xFFFFFu, (uint32_t)(h >> 20) & 0xFFFFFu, (uint32_t)(h >> 40) & 0xFFFFFu};
        uint32_t old[3] = {0, 0, 0};
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if ((uint32_t)e < nv) old[e] = atomicAdd(&smem[dlt_word<true>(x[e])], 1u << dlt_bit<true>(x[e]));  // ds_add_rtn_u32
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if ((uint32_t)e < nv) over |= (uint32_t)(((old[e] >> dlt_bit<true>(x[e])) & 15u) == 15u);
    };
    auto apply = [&](const uint4 (&q)[D]) {
#pragma unroll
        for (int d = 0; d < D; ++d) { half(q[d].x, q[d].y); half(q[d].z, q[d].w); }
    };
    // the slice's probes straight onto the table (a counter of the slice was hit 16 times or more: the image is void)
    auto slow_slice = [&](uint32_t b) {
        const uint64_t c0 = (uint64_t)b << g.shift;
        auto slow = [&](uint32_t lo, uint32_t hi) {
            const unsigned long long h = ((unsigned long long)hi << 32) | lo;
            const uint32_t nv = hi >> 28;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const uint32_t xe = (uint32_t)(h >> (20 * e)) & 0xFFFFFu;
                if ((uint32_t)e < nv) {
                    const uint64_t cell = c0 + xe;
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
    };
    auto probe_all = [&](uint32_t b) {  // unpipelined probe phase (the workgroup's first slice; the slice behind an overflowed one)
        SegWalk<D> w;
        w.init(buckets, segcnt, g, b);
        while (w.more()) {
            uint4 q[D];
            w.next(q, zero4);
            apply(q);
        }
    };
    auto fold = [&](uint32_t t, uint32_t d) -> uint32_t {
        if (d == 0) return t;
        if (OPT == 1) {
            bad |= (uint32_t)(t < d) | (uint32_t)(t == 0xFFFFFFFFu);
            return t - d;
        }
        if (OPT == 2) return t + d;
        if (NEG) {  // countingbloom.py:203-206
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
        if (gc + 3 < tab_cells) {
            if (nt) {
                psk_u32x4 v;
                v.x = o.x; v.y = o.y; v.z = o.z; v.w = o.w;
                __builtin_nontemporal_store(v, reinterpret_cast<psk_u32x4 *>(tab + gc));
            } else {
                *reinterpret_cast<uint4 *>(tab + gc) = o;
            }
            return;
        }
        if (gc + 0 < tab_cells) tab[gc + 0] = o.x;  // the table ends inside this piece
        if (gc + 1 < tab_cells) tab[gc + 1] = o.y;
        if (gc + 2 < tab_cells) tab[gc + 2] = o.z;
    };

    for (uint32_t w = threadIdx.x; w < words; w += kApplyThreads) smem[w] = 0;
    if (threadIdx.x < 2) carried[threadIdx.x] = 0;
    __syncthreads();
    uint32_t par = 0;
    if (blockIdx.x < g.nbuckets && ((uint64_t)blockIdx.x << g.shift) < tab_cells) {
        probe_all(blockIdx.x);
        if (over) carried[0] = 1u;
        over = 0;
    }
    __syncthreads();
    for (uint32_t b = blockIdx.x; b < g.nbuckets; b += gridDim.x) {
        const uint64_t c0 = (uint64_t)b << g.shift;
        if (c0 >= tab_cells) break;  // (slices past the table's end)
        const uint32_t nb = b + gridDim.x;
        const bool have_next = nb < g.nbuckets && ((uint64_t)nb << g.shift) < tab_cells;
        if (carried[par]) {  // uniform: exact atomics for this slice; its successor's probes then go in unpipelined
            slow_slice(b);
            __syncthreads();
            for (uint32_t w = threadIdx.x; w < words; w += kApplyThreads) smem[w] = 0;
            if (threadIdx.x == 0) carried[par] = 0;
            __syncthreads();
            if (have_next) {
                probe_all(nb);
                if (over) carried[par ^ 1u] = 1u;
                over = 0;
            }
            par ^= 1u;
            __syncthreads();
            continue;
        }
        // ---- the image moves into registers; every lane zeroes the words it took
        uint32_t dl[kPipeWords];
#pragma unroll
        for (int k = 0; k < kPipeWords; ++k) {
            dl[k] = 0;
            if ((uint32_t)k < nk) {
                dl[k] = smem[(uint32_t)k * kApplyThreads + threadIdx.x];
                smem[(uint32_t)k * kApplyThreads + threadIdx.x] = 0;
            }
        }
        __syncthreads();  // (all of carried[par] read, all words zeroed)
        if (threadIdx.x == 0) carried[par] = 0;
        SegWalk<D> walk;
        if (have_next) walk.init(buckets, segcnt, g, nb);
        else walk.none();
        // ---- fold slice b out of the registers, two blocks (four 16-byte pieces per lane) per step, under the probe groups of slice nb
        // (the image words are registers, so the loop cannot index them: every step folds dl[0 .. U) and then ROTATES the words down by U --
        // 30 moves per step; unrolling the 16 steps instead made 22 K instructions of it, the 64-bit saturation logic 256 times over)
        constexpr int U = 2;
        for (uint32_t k0 = 0; k0 < nk; k0 += U) {
            uint4 t[U][2];
            bool touch[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t gc = c0 + (uint64_t)(k0 + u) * 8192u + 4u * threadIdx.x;
                t[u][0] = t[u][1] = zero4;
                // A piece is read and re-written when ANY lane of the wave-instruction has a delta for its piece: the 64 lanes cover 1 KiB of
                // contiguous table, and whole lines out and back run at the copy rate (scripts/ubench/tabpass.hip: 5.3 TB/s) where the
                // per-lane test -- 65 % of the pieces of a 10 M-key batch, scattered 16-byte writes into lines read in part -- reached 3.8.
                touch[u][0] = __any((dl[u] & 0xFFFFu) != 0u);
                touch[u][1] = __any((dl[u] >> 16) != 0u);
                if (touch[u][0]) t[u][0] = nib_load_piece(tab, tab_cells, gc, nt != 0);
                if (touch[u][1]) t[u][1] = nib_load_piece(tab, tab_cells, gc + 4096u, nt != 0);
            }
            // a probe step of the next slice under every other fold step (a slice brings 2-3 steps per wave at 10 M keys)
            uint4 q[D];
            const bool had = (k0 & U) == 0 && walk.more();  // (uniform)
            if (had) walk.next(q, zero4);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t gc = c0 + (uint64_t)(k0 + u) * 8192u + 4u * threadIdx.x;
                if (touch[u][0]) store_piece(gc, fold4(t[u][0], dl[u] & 0xFFFFu));
                if (touch[u][1]) store_piece(gc + 4096u, fold4(t[u][1], dl[u] >> 16));
            }
            if (had) apply(q);
#pragma unroll
            for (int k = 0; k + U < kPipeWords; ++k) dl[k] = dl[k + U];
        }
        while (walk.more()) {  // more probe steps than fold steps (heavy slices)
            uint4 q[D];
            walk.next(q, zero4);
            apply(q);
        }
        if (over) carried[par ^ 1u] = 1u;
        over = 0;
        par ^= 1u;
        __syncthreads();
    }
    if (sat) atomicAdd(sat_ctr, sat);
    if (viol) atomicAdd(sat_ctr - 1, viol);
    if (OPT == 1 && bad) *flag = 1u;
}

// ------------------------------------------------------------------------------------ lookups: the slice load under the probe walk
// k_nib_gather's two phases -- load 1 MiB of table and pack it to nibbles; answer the slice's probe groups from the image -- follow each
// other in a workgroup that holds the whole CU (128 KiB image): 292 us per 10 M keys where the table read alone takes 155-173 us
// (scripts/ubench/tabpass.hip).  Here the NEXT slice of a persistent workgroup is loaded and packed into registers (64 pieces per lane ->
// 64 halves = 32 VGPRs) while the current slice's probe groups are answered; two barriers hand the registers over to the image.
// Image layout and kept images (shadow_out) exactly as k_nib_gather; the host takes this kernel when no kept images exist to load.
template <int DUMMY = 0>
__global__ __launch_bounds__(kApplyThreads, 4) void k_nib_gather_pipe(const uint32_t *tab, uint64_t tab_cells, PartGeom g, const uint32_t *segcnt, const uint4 *buckets,
                                                                     uint32_t *vals, uint32_t nt, uint32_t *shadow_out)
{
    constexpr int D = 3;   // probe groups in flight per lane
    constexpr int LP = 4;  // table pieces in flight per lane (one load step).  Registers bound both: 32 words of the next image + LP x 4 + D x (4 + 6 LDS words);
                           // LP = 8 with D = 2 spilled and ran at 366 us, two steps of four in flight (double-buffered) spilled as well
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint16_t *half16 = reinterpret_cast<uint16_t *>(smem);
    const uint32_t pieces = 1u << (g.shift - 2);           // 16-byte pieces of a slice = 16-bit halves of its image
    const uint32_t nsteps = pieces / (kApplyThreads * LP);  // load steps per slice (1 .. 8)
    const uint32_t vecs = 1u << (g.shift - 5);              // 16-byte units of an image
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    uint32_t r[kPipeWords];  // the next image, two halves per word: word k = pieces t + 1024 * 2k (low) and t + 1024 * (2k + 1) (high) of lane t
#pragma unroll
    for (int k = 0; k < kPipeWords; ++k) r[k] = 0;

    auto slice_ok = [&](uint32_t b) { return b < g.nbuckets && ((uint64_t)b << g.shift) < tab_cells; };
    // one load step of slice b: pieces t + 1024 * (LP j + 0 .. LP - 1)
    // `whole`: the slice lies inside the table (all but possibly the last): a uniform base + the lane's offset, no per-piece bound test
    auto whole = [&](uint32_t b) { return (((uint64_t)b + 1) << g.shift) <= tab_cells; };
    auto issue = [&](uint4 (&t)[LP], uint32_t b, uint32_t j) {
        const psk_u32x4 *sb = reinterpret_cast<const psk_u32x4 *>(tab + ((uint64_t)b << g.shift)) + (size_t)kApplyThreads * LP * j;  // uniform
#pragma unroll
        for (int u = 0; u < LP; ++u) {
            const psk_u32x4 v = nt ? __builtin_nontemporal_load(sb + kApplyThreads * u + threadIdx.x) : sb[kApplyThreads * u + threadIdx.x];
            t[u] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    auto issue_tail = [&](uint4 (&t)[LP], uint32_t b, uint32_t j) {  // the slice the table ends in
        const uint64_t c0 = (uint64_t)b << g.shift;
#pragma unroll
        for (int u = 0; u < LP; ++u) t[u] = nib_load_piece(tab, tab_cells, c0 + 4ULL * (threadIdx.x + kApplyThreads * ((uint32_t)LP * j + (uint32_t)u)), nt != 0);
    };
    auto pack = [&](const uint4 (&t)[LP]) {  // the words rotate down by LP / 2; the step's words come in at the top
#pragma unroll
        for (int k = 0; k + LP / 2 < kPipeWords; ++k) r[k] = r[k + LP / 2];
#pragma unroll
        for (int u = 0; u < LP / 2; ++u) r[kPipeWords - LP / 2 + u] = nib_pack4(t[2 * u]) | (nib_pack4(t[2 * u + 1]) << 16);
    };
    auto load_plain = [&](uint32_t b) {  // a slice into the registers, nothing overlapped
        for (uint32_t j = 0; j < nsteps; ++j) {
            uint4 t[LP];
            if (whole(b)) issue(t, b, j);
            else issue_tail(t, b, j);
            pack(t);
        }
    };
    // registers -> image: after `nsteps` steps the slice's words sit in r[32 - (LP / 2) nsteps .. 32)
    auto hand_over = [&]() {
#pragma unroll
        for (int k = 0; k < kPipeWords; ++k) {
            const int first = kPipeWords - (LP / 2) * (int)nsteps;  // (uniform)
            if (k >= first) {
                const uint32_t u2 = (uint32_t)(k - first);
                half16[threadIdx.x + kApplyThreads * (2u * u2)] = (uint16_t)(r[k] & 0xFFFFu);
                half16[threadIdx.x + kApplyThreads * (2u * u2 + 1u)] = (uint16_t)(r[k] >> 16);
            }
        }
    };
    auto answer = [&](const uint4 (&q)[D], const uint32_t (&at)[D]) {
        uint32_t w[D][6];
#pragma unroll
        for (int d = 0; d < D; ++d)  // the LDS reads of the whole step first (slots past a run's end read counter 0: harmless)
#pragma unroll
            for (int e = 0; e < 6; ++e) w[d][e] = smem[nib_word(group_field(q[d], e))];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (at[d] != 0xFFFFFFFFu) {
                uint32_t rr = 0;
#pragma unroll
                for (int e = 0; e < 6; ++e) rr |= ((w[d][e] >> nib_bit(group_field(q[d], e))) & 15u) << (4 * e);
                vals[at[d]] = rr;
            }
        }
    };

    // ---- my first slice: load, pack, hand over (nothing to overlap with yet)
    if (!slice_ok(blockIdx.x)) return;
    load_plain(blockIdx.x);
    hand_over();
    __syncthreads();
    for (uint32_t b = blockIdx.x; slice_ok(b); b += gridDim.x) {
        if (shadow_out) {  // (the stores drain under the probe walk below)
            uint4 *dst = reinterpret_cast<uint4 *>(shadow_out) + (uint64_t)b * vecs;
            const uint4 *src = reinterpret_cast<const uint4 *>(smem);
            for (uint32_t pc = threadIdx.x; pc < vecs; pc += kApplyThreads) dst[pc] = src[pc];
        }
        const uint32_t nb = b + gridDim.x;
        const bool next_any = slice_ok(nb);
        const bool have_next = next_any && whole(nb);  // (the slice the table ends in is loaded behind the walk, unpipelined)
        SegWalk<D> walk;
        walk.init(buckets, segcnt, g, b);
        // every load step of the next slice is followed by probe steps of this one: the pieces land while the groups are answered
        const uint32_t per = 1;  // probe steps per load step (a slice brings ~4 steps of three groups per wave at 10 M keys)
        if (have_next) {
            for (uint32_t j = 0; j < nsteps; ++j) {
                // (the probe groups are requested BEFORE the pieces: vmcnt counts in order, so waiting for the groups leaves the pieces in flight)
                uint4 q[D];
                uint32_t at[D];
                const bool had = walk.more();  // (uniform)
                if (had) walk.next_at(q, at, zero4);
                uint4 t[LP];
                issue(t, nb, j);
                if (had) answer(q, at);
                for (uint32_t e = 1; e < per && walk.more(); ++e) {
                    uint4 q2[D];
                    uint32_t at2[D];
                    walk.next_at(q2, at2, zero4);
                    answer(q2, at2);
                }
                pack(t);
            }
        }
        while (walk.more()) {
            uint4 q[D];
            uint32_t at[D];
            walk.next_at(q, at, zero4);
            answer(q, at);
        }
        if (next_any && !have_next) load_plain(nb);
        __syncthreads();  // every wave is done with this slice's image
        if (next_any) hand_over();
        __syncthreads();
    }
}

}  // namespace psk
