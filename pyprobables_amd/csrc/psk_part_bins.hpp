// psk_part_bins.hpp -- pass 1 without the sort: the probes of a tile go STRAIGHT into fixed-capacity LDS bins, one per slice (round 6).
//
// k_part_scatter (psk_partition.hpp) ranks every probe in an LDS histogram, waits at a barrier, lets ONE wave scan the histogram
// (the other waves idle at a second barrier), walks all probes a second time to copy them into a sorted stage (an off[] read and a
// stage write per probe, idx[] / rank[] held in 2 x 28 registers across both barriers), fills the runs' pads, waits at a third barrier and
// only then writes groups out.  The counting sort buys compact runs; what pass 2 needs is only "the probes of slice b, whole 16-byte groups".
//
// Here a slice's bin has room for `cap` probes of a tile (mean + ~3.5 sigma of the tile's Poisson load):
//   hash phase   rank4 = ds_add_rtn(cnt4[b], 4) hands the probe its slot; the 20-bit slice-local value is stored at bins[b][rank] at once --
//                nothing of a key outlives its own chains (no idx[] / rank[] arrays: ~55 VGPRs instead of 113), a probe whose bin is full
//                takes the exact fallback (Spill) -- ~1e-3 of the bins of a tile at 3.5 sigma;
//   barrier
//   write-out    slice b belongs to the SAME lanes in every tile (NT / B lanes per slice, or several slices per lane): they turn the bin
//                into 16-byte groups (the probe format of PayNone / PayTileTag: 6 x 20 bits, counted halves), append them at a segment
//                cursor kept in a REGISTER, and zero the bin's count;
//   barrier
// -- two barriers per tile, no scan, no second pass over the probes, no cursor / offset / delta arrays.  Same segments, same groups, same
// segment counts as k_part_scatter: pass 2 (k_bloom_apply, k_bloom_test_flag, the nibble folds) does not know which pass 1 ran.
// Bins are `stride` words apart (cap rounded up to an even number + 2): consecutive bins start two banks apart, so that slot r of all
// bins -- what the lanes of a wave write at about the same time -- does not sit in ONE bank.
//
// Eligible (launch_scatter_nt): the 6 x 20-bit probe groups -- PayNone, PayTileTag (Bloom inserts / tile-flag lookups, CBF unit updates) and
// PayWeightSmall (weighted CountMinSketch adds: field = weight << 15 | cell, a pad is the all-zero field) --, k <= 8, the 16- and 8-byte key
// layouts, bins that fit the LDS twice per CU, not the append mode of the write-combined lists.  Everything else keeps k_part_scatter.
#pragma once
#include "psk_partition.hpp"

namespace psk {

// threads per workgroup (A/B: -DPSK_BINS_NT=n, a multiple of 64).  768 threads x 2 keys = 1536-key tiles with SIX waves per SIMD (two workgroups per
// CU, 57 VGPRs) measured 1 % ahead of 512 x 3 with four (profiles/r06_ab_pass1.txt)
#ifndef PSK_BINS_NT
#define PSK_BINS_NT 768
#endif
constexpr int kBinThreads = PSK_BINS_NT;
constexpr int kBinMinWaves = kBinThreads <= 512 ? 4 : (kBinThreads <= 768 ? 6 : 8);  // waves per SIMD the register budget must allow: two workgroups per CU
// keys per thread and tile (A/B: scripts/build_variant.sh -DPSK_BINS_KPT=n): 1536-key tiles = a bin of 66 probes at 256 slices, k = 7 (72 KB of LDS:
// two workgroups per CU).  1024-key tiles (three workgroups per CU) lose more to their shorter runs -- more pad slots, and a 10 M-key tile-flag
// lookup no longer fits ONE round of 16 tiles per workgroup -- than they win: step 392 against 354 us; 2048-key tiles do not fit twice.
#ifndef PSK_BINS_KPT
#define PSK_BINS_KPT 2
#endif
constexpr int kBinKpt = PSK_BINS_KPT;
constexpr int kBinSlicesPerLane = 2;  // slices a write-out lane may own: at most kBinThreads * 2 slices

template <class Pay>
struct pay_bins_ok { static constexpr bool value = false; };
template <>
struct pay_bins_ok<PayNone> { static constexpr bool value = true; };
template <>
struct pay_bins_ok<PayTileTag> { static constexpr bool value = true; };
template <>
struct pay_bins_ok<PayWeightSmall> { static constexpr bool value = true; };  // weighted CountMinSketch adds, weights 0 .. 15 (countminsketch.py:267-288)

// dynamic LDS: B bins of `stride` words: word 1 = 4 x (2 + probes in the bin) (the byte offset of the next free slot), words 2 .. 2 + cap - 1 the probes
// (8-byte aligned: the write-out reads pairs), word 2 + cap = the slot the stores of a full bin land on; + 8 words of slack behind the last bin.
// g.tile keys per tile (<= kBinThreads * KPT); rounds hold fewer than 2^32 keys (32-bit key indices: launch_scatter_bins checks).
// The kernel is VALU bound (SQ counters, profiles/r06_sq_pass1.txt: every SIMD issues a VALU instruction in ~96 % of its cycles, 2 of 3 of
// them the FNV chains), so everything around the chains is counted in instructions: one v_bfe + one 24-bit multiply turn a hash into its bin's
// byte offset, which is the address of the bin's counter AND (plus what the counter returns) of the probe's slot.
constexpr uint32_t kBinHead = 2;  // words in front of a bin's probes
template <class Src, class IdxFn, class Pay, class Spill, int KT, int KPT>
__global__ __launch_bounds__(kBinThreads, kBinMinWaves) void k_part_bins(Src src, IdxFn idxfn, Pay pay, Spill spill, PartGeom g, uint32_t n, uint32_t cap,
                                                              uint32_t stride, uint32_t *segcnt, uint4 *buckets)
{
    constexpr int NT = kBinThreads;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    char *lds = reinterpret_cast<char *>(smem);
    const uint32_t B = g.nbuckets;
    constexpr bool kExactK = KT != 8;
    const uint32_t k = kExactK ? (uint32_t)KT : g.k;
    const uint32_t mask = (1u << g.shift) - 1;
    const uint32_t tk = g.tile;
    const uint32_t ntiles = (n + tk - 1) / tk;
    const uint32_t last = n ? n - 1 : 0;
    const uint32_t stride4 = stride * 4u;
    const uint32_t lg_slices = 31u - (uint32_t)__builtin_clz(B);  // (power-of-two tables: B is a power of two)
    const uint32_t full4 = (cap + kBinHead) * 4u;  // the counter of a full bin = the byte offset of the slot behind its probes
    constexpr uint32_t kCnt = 4;                   // byte offset of a bin's counter (word 1)

    // my slices in the write-out: L lanes share a slice (B <= NT), or a lane owns several (B > NT)
    // (L a power of two: the lanes of a slice must be neighbours inside ONE wave -- see the reset of the counters below)
    uint32_t L = 1;
    while (2 * L * B <= (uint32_t)NT) L *= 2;
    constexpr int kSlicesPerLane = kBinSlicesPerLane;
    const uint32_t sub = B <= (uint32_t)NT ? threadIdx.x % L : 0u;
    uint32_t myb[kSlicesPerLane], cur[kSlicesPerLane];
    uint4 *seg[kSlicesPerLane];  // my segment of the slice (slot 0)
#pragma unroll
    for (int s = 0; s < kSlicesPerLane; ++s) {
        myb[s] = B <= (uint32_t)NT ? (s == 0 && threadIdx.x / L < B ? threadIdx.x / L : ~0u) : threadIdx.x + (uint32_t)s * NT;
        if (myb[s] >= B) myb[s] = ~0u;
        cur[s] = 0;
        seg[s] = buckets + (seg_index(g, 0, blockIdx.x) + (myb[s] == ~0u ? 0u : myb[s])) * g.segcap;
    }
    for (uint32_t b = threadIdx.x; b < B; b += NT) *reinterpret_cast<uint32_t *>(lds + b * stride4 + kCnt) = kBinHead * 4u;

    // bench-only (a -DPSK_BENCH_KNOBS=1 build, scripts/ablate.py; every test below folds to false in the shipped library): PartGeom::dbg bit 1 =
    // no segment stores, 2 = hashing only, 4 = no hashing (a cheap stand-in for the chains), 32 = phase profile -- lane 0 of the FIRST and of the
    // LAST wave accumulate cycle-counter deltas per phase (hash + slots, wait at barrier 1, write-out, wait at barrier 2) in the LDS slack
    const uint32_t dbg = kBenchKnobs ? g.dbg : 0u;
    unsigned long long *t_acc = reinterpret_cast<unsigned long long *>(lds + (size_t)B * stride4 + 32);  // 8 slots (host: kBinSlackWords)
    unsigned long long t_prev = 0;
    const bool t_lane = (dbg & 32) && (threadIdx.x == 0 || threadIdx.x == NT - 64);
    const uint32_t t_base = threadIdx.x == 0 ? 0u : 4u;
    if ((dbg & 32) && threadIdx.x < 8) t_acc[threadIdx.x] = 0;
#define PSK_BIN_TICK(ph)                                                 \
    if (t_lane) {                                                        \
        const unsigned long long t_now = __builtin_readcyclecounter();   \
        t_acc[t_base + (ph)] += t_now - t_prev;                          \
        t_prev = t_now;                                                  \
    }
    uint32_t fold = 0;  // (dbg & 2: keeps the hashes alive)
    constexpr bool WP = pay_weighted_plain<Pay>::value;  // PayWeightSmall: every probe carries its key's weight (prefetched with the key)
    typename Src::Key kcur[KPT];
    uint32_t wcur[WP ? KPT : 1];
    long long tally_s = 0;            // fused weight accounting (PayWeightSmall::tally, see k_part_scatter)
    unsigned long long tally_a = 0;
    uint32_t tally_b = 0;
    {
        const uint32_t b0 = blockIdx.x * tk;
#pragma unroll
        for (int q = 0; q < KPT; ++q) {
            const uint32_t i = b0 + (uint32_t)q * NT + threadIdx.x;
            kcur[q] = src.load(i < last ? i : last);  // clamped, never branched around
            if constexpr (WP) wcur[q] = pay(i < last ? i : last, 0);
        }
    }
    lds_barrier();
    if (t_lane) t_prev = __builtin_readcyclecounter();

    uint32_t ordinal = ~0u;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        ++ordinal;
        const uint32_t base = tile * tk;
        const uint32_t tile_end = base + tk < n ? base + tk : n;
        if constexpr (PSK_EXP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        // ---- hash, take a slot, store: key after key
#pragma unroll
        for (int q = 0; q < KPT; ++q) {
            const uint32_t i = base + (uint32_t)q * NT + threadIdx.x;
            if ((uint32_t)q * NT < tk && i < tile_end) {
                // val[j] = the probe's slice-local value, bin[j] = its slice.  Power-of-two tables (32-bit chains): both are bit fields of the
                // hash itself -- h & mask and v_bfe(h, shift, lg B): no index h & (m - 1) in between
                uint32_t val[KT], bin[KT];
                uint32_t wfield = 0, wbig = 0, wq = 0;  // the weight as the probes carry it (<< shift), / a weight outside 0 .. 15: straight to the table
                if constexpr (WP) {
                    wq = wcur[q];
                    if (pay.tally) {
                        const long long v = pay.weights_signed ? (long long)(int32_t)wq : (long long)wq;
                        tally_s += v;
                        tally_a += (unsigned long long)(v < 0 ? -v : v);
                        tally_b += (uint32_t)(wq >= (1u << kSmallWeightBits));
                    }
                    wbig = wq >= (1u << kSmallWeightBits) ? 1u : 0u;
                    wfield = wbig ? 0u : wq << 15;   // (field = weight << 15 | 15-bit cell; a no-op field stays behind a weight that went to the table)
                }
                if constexpr (IdxFn::lo32) {
                    uint32_t h[KT];
                    if (dbg & 4) {
#pragma unroll
                        for (int j = 0; j < KT; ++j) h[j] = (uint32_t)(((uint64_t)(i * 2654435761u + (uint32_t)j * 40503u) * 0x9E3779B97F4A7C15ULL) >> 13);
                    } else {
                        src.template hash32<KT>(kcur[q], i, 0u, h);
                    }
#pragma unroll
                    for (int j = 0; j < KT; ++j) {
                        if constexpr (std::is_same<IdxFn, IdxBloom<true>>::value || std::is_same<IdxFn, IdxBloomWide<true>>::value) {
                            val[j] = h[j] & mask;                                            // (index = h & (m - 1), m = B << shift)
                            bin[j] = __builtin_amdgcn_ubfe(h[j], g.shift, lg_slices);
                        } else {
                            const uint32_t x = idxfn.from32((uint32_t)j, h[j]);
                            val[j] = x & mask;
                            bin[j] = x >> g.shift;
                        }
                    }
                } else {
                    uint64_t h[KT];
                    src.template hash<KT>(kcur[q], i, 0u, h);
#pragma unroll
                    for (int j = 0; j < KT; ++j) {
                        const uint32_t x = idxfn((uint32_t)j, h[j]);
                        val[j] = x & mask;
                        bin[j] = x >> g.shift;
                    }
                }
                if (dbg & 2) {  // bench-only: hashing alone
#pragma unroll
                    for (int j = 0; j < KT; ++j) fold ^= val[j] + bin[j];
                    continue;
                }
                uint32_t at[KT], r4[KT], worst = 0;
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    r4[j] = 0;
                    if ((uint32_t)j < k) {
                        at[j] = __umul24(bin[j], stride4);  // the bin's byte offset
                        r4[j] = atomicAdd(reinterpret_cast<uint32_t *>(lds + at[j] + kCnt), 4u);  // ds_add_rtn_u32: the byte offset of my slot in the bin
                    }
                }
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    if ((uint32_t)j < k) {
                        worst = r4[j] > worst ? r4[j] : worst;
                        // (a full bin: the store lands on the slot behind the bin's probes, which holds nothing, and the probe takes the exact fallback below)
                        *reinterpret_cast<uint32_t *>(lds + at[j] + (r4[j] < full4 ? r4[j] : full4)) = WP ? (val[j] | wfield) : val[j];
                    }
                }
                if constexpr (WP) {
                    if (wbig) {  // big / negative weight: exact saturating add on the table (rare: the host picks this format while there were none)
#pragma unroll
                        for (int j = 0; j < KT; ++j)
                            if ((uint32_t)j < k) spill((bin[j] << g.shift) | val[j], wq);
                    }
                }
                if (worst >= full4) {  // rare: this key met a full bin
#pragma unroll
                    for (int j = 0; j < KT; ++j)
                        if ((uint32_t)j < k && r4[j] >= full4) {
                            if constexpr (pay_tile_tag<Pay>::value) spill((bin[j] << g.shift) | val[j], tile);
                            else if constexpr (WP) { if (!wbig) spill((bin[j] << g.shift) | val[j], wq); }
                            else spill((bin[j] << g.shift) | val[j], 0u);
                        }
                }
            }
        }
        // ---- the next tile's keys (they land under the write-out)
        {
            const uint32_t nb = (tile + gridDim.x) * tk;
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const uint32_t i = nb + (uint32_t)q * NT + threadIdx.x;
                kcur[q] = src.load(i < last ? i : last);
                if constexpr (WP) wcur[q] = pay(i < last ? i : last, 0);
            }
        }
        if (dbg & 2) {  // bench-only: hashing alone -- no barrier, no write-out (uniform)
            if (fold == 0x12345u) segcnt[0] = fold;
            continue;
        }
        PSK_BIN_TICK(0);
        lds_barrier();
        PSK_BIN_TICK(1);
        if constexpr (PSK_EXP_PRIO == 1) __builtin_amdgcn_s_setprio(3);

        // ---- write-out: my slices' bins as 16-byte groups behind my segments' cursors
        const uint32_t t0 = pay_tile_tag<Pay>::value ? (ordinal & 3u) << 2 : 0u, t1 = pay_tile_tag<Pay>::value ? ordinal & 12u : 0u;
#pragma unroll
        for (int s = 0; s < kSlicesPerLane; ++s) {
            const uint32_t b = myb[s];
            if (b == ~0u) continue;
            const char *bin = lds + b * stride4;
            const uint32_t c4 = *reinterpret_cast<const uint32_t *>(bin + kCnt);
            const uint32_t c = ((c4 < full4 ? c4 : full4) >> 2) - kBinHead;  // probes in the bin
            const uint32_t whole = c / 6u, rem = c - 6u * whole, ngroups = whole + (rem ? 1u : 0u);
            const uint2 *e = reinterpret_cast<const uint2 *>(bin + kBinHead * 4u);
            uint4 *out = seg[s] + cur[s];
            if (cur[s] + ngroups <= g.segcap) {  // (all but never)
                const uint32_t hi0 = (3u | t0) << 28, hi1 = (3u | t1) << 28;
                for (uint32_t gq = sub; gq < whole; gq += L) {  // whole groups: three probes in either half
                    const uint2 a0 = e[3 * gq], a1 = e[3 * gq + 1], a2 = e[3 * gq + 2];
                    uint4 o;
                    o.x = a0.x | (a0.y << 20);
                    o.y = (a0.y >> 12) | (a1.x << 8) | hi0;
                    o.z = a1.y | (a2.x << 20);
                    o.w = (a2.x >> 12) | (a2.y << 8) | hi1;
                    if (!(dbg & 1)) out[gq] = o;
                }
                if (rem && sub == (whole & (L - 1))) {  // the run's last group: the slots past the count read as pads (all ones in the field, as k_part_scatter leaves them)
                    const uint2 a0 = e[3 * whole], a1 = e[3 * whole + 1], a2 = e[3 * whole + 2];
                    uint32_t f[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
#pragma unroll
                    for (int x = 1; x < 6; ++x)
                        if ((uint32_t)x >= rem) f[x] = WP ? 0u : mask;
                    const uint32_t n0 = rem < 3u ? rem : 3u, n1 = rem - n0;
                    uint4 o;
                    o.x = f[0] | (f[1] << 20);
                    o.y = (f[1] >> 12) | (f[2] << 8) | ((n0 | t0) << 28);
                    o.z = f[3] | (f[4] << 20);
                    o.w = (f[4] >> 12) | (f[5] << 8) | ((n1 | t1) << 28);
                    if (!(dbg & 1)) out[whole] = o;
                }
            } else {  // the segment fills up: group by group, what does not fit takes the exact fallback probe by probe
                for (uint32_t gq = sub; gq < ngroups; gq += L) {
                    const uint2 a0 = e[3 * gq], a1 = e[3 * gq + 1], a2 = e[3 * gq + 2];
                    uint32_t f[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
                    const uint32_t left = c - 6u * gq, nv = left < 6u ? left : 6u;
#pragma unroll
                    for (int x = 1; x < 6; ++x)
                        if ((uint32_t)x >= nv) f[x] = WP ? 0u : mask;
                    const uint32_t n0 = nv < 3u ? nv : 3u, n1 = nv - n0;
                    if (cur[s] + gq < g.segcap) {
                        uint4 o;
                        o.x = f[0] | (f[1] << 20);
                        o.y = (f[1] >> 12) | (f[2] << 8) | ((n0 | t0) << 28);
                        o.z = f[3] | (f[4] << 20);
                        o.w = (f[4] >> 12) | (f[5] << 8) | ((n1 | t1) << 28);
                        out[gq] = o;
                    } else {
#pragma unroll
                        for (int x = 0; x < 6; ++x)
                            if ((uint32_t)x < nv) {
                                if constexpr (pay_tile_tag<Pay>::value) spill((b << g.shift) | f[x], tile);
                                else if constexpr (WP) spill((b << g.shift) | (f[x] & mask), f[x] >> 15);
                                else spill((b << g.shift) | f[x], 0u);
                            }
                    }
                }
            }
            cur[s] += ngroups;
        }
        // (the lanes that share a slice are neighbours in ONE wave, and a wave's LDS operations complete in order: the counter is reset behind
        // every read of it above)
#pragma unroll
        for (int s = 0; s < kSlicesPerLane; ++s)
            if (myb[s] != ~0u && sub == 0) *reinterpret_cast<uint32_t *>(lds + myb[s] * stride4 + kCnt) = kBinHead * 4u;
#pragma unroll
        for (int q = 0; q < KPT; ++q) {
            Src::pin(kcur[q]);
            if constexpr (WP) asm volatile("" : "+v"(wcur[q]));
        }
        PSK_BIN_TICK(2);
        lds_barrier();
        PSK_BIN_TICK(3);
    }
    if (t_lane) {  // (bench-only) behind the segment counts, where scripts/ablate.py reads them (psk_debug_phase_profile)
        unsigned long long *prof = reinterpret_cast<unsigned long long *>(segcnt + (size_t)g.nbuckets * g.nwg);
        for (int ph = 0; ph < 4; ++ph) atomicAdd(prof + 1 + t_base + ph, t_acc[t_base + ph]);
        if (threadIdx.x == 0) atomicAdd(prof, 1ULL);
    }
#undef PSK_BIN_TICK
    if constexpr (WP) {  // (sum w, sum |w|, weights outside 0 .. 15) of my keys -> my slot: plain stores, folded by pass 2 / k_tally_fold
        if (pay.tally) {
            for (int o = 32; o > 0; o >>= 1) {
                tally_s += __shfl_down(tally_s, o);
                tally_a += __shfl_down(tally_a, o);
                tally_b += __shfl_down(tally_b, o);
            }
            unsigned long long *red = reinterpret_cast<unsigned long long *>(smem);  // (the tile loop is over: the bins are free)
            if ((threadIdx.x & 63) == 0) {
                red[3 * (threadIdx.x >> 6)] = (unsigned long long)tally_s;
                red[3 * (threadIdx.x >> 6) + 1] = tally_a;
                red[3 * (threadIdx.x >> 6) + 2] = tally_b;
            }
            lds_barrier();
            if (threadIdx.x == 0) {
                unsigned long long ss = 0, aa = 0, bb = 0;
                for (int w = 0; w < NT / 64; ++w) { ss += red[3 * w]; aa += red[3 * w + 1]; bb += red[3 * w + 2]; }
                pay.tally[blockIdx.x] = make_ulonglong4(ss, aa, bb, 0ULL);
            }
        }
    }
    // publish how many groups of each of my segments are valid (the kernel boundary orders it before pass 2)
#pragma unroll
    for (int s = 0; s < kSlicesPerLane; ++s)
        if (myb[s] != ~0u && sub == 0) segcnt[(uint64_t)myb[s] * g.nwg + blockIdx.x] = cur[s] < g.segcap ? cur[s] : g.segcap;
}

}  // namespace psk
