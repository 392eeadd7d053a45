// psk_part_lookup.hpp -- launcher template of the partitioned counter lookups (psk_lookup.hpp: pass 1 with perm / runinfo
// by-products, pass 2 k_counter_gather, pass 3 k_lookup_collect), shared by the CMS and CBF translation units.
#pragma once
#include "psk_host.hpp"
#include "psk_lookup.hpp"
#include "psk_nibble.hpp"
#include "psk_nibble_pipe.hpp"

extern PSK_HIDDEN int64_t g_nib_gather_pipe;  // psk_capi.hip: option "nibble_lookup_pipe"
extern PSK_HIDDEN int64_t g_cbf_shadow_hits;  // nibble-slice lookups that loaded kept images (psk_sketch::shadow)

// Keys per round.  Measured on MI355X (10 M CMS lookups): one round of 10 M keys 432 us, two 446, three cache-sized ones 464
// -- the three kernels of a round stream ~50 B per key once and every round re-reads the table, so unlike the update
// paths (part_round_keys) rounds are only cut at `partition_max_keys`.
static inline uint64_t lookup_round_keys(uint64_t n, uint32_t k)
{
    uint64_t rk = (uint64_t)g_part_max_keys < n ? (uint64_t)g_part_max_keys : n;
    rk = cap_round_by_budget(rk, (double)k * (2.0 + 4.0) * 1.5 + 4.0 * ((k + 1) / 2) + 8.0);  // probes + values + perm + runinfo
    // pass 3 addresses a run's values by a 32-bit unit index (k_lookup_collect's run descriptors): at most 2^31 probes per round keeps the
    // round's groups, pads included, far below 2^31
    const uint64_t by_probes = (1ULL << 31) / (k ? k : 1);
    if (rk > by_probes) rk = by_probes;
    return rk ? rk : 1;
}

// IDX: index functor family (IdxCms / IdxBloom); `query` is the pass-3 epilogue; `redo(flag, st)` enqueues the flag-guarded
// direct kernel over the whole batch (exactness when a segment overflowed).  *done = false: nothing was launched.
template <template <bool> class IDX, class Query, class Redo>
static inline int counter_check_partitioned(psk_sketch *s, const Batch &b, uint32_t kk, uint64_t cells, const Query &query,
                                            typename Query::Out *out_dev, hipStream_t st, bool *done, Redo &&redo)
{
    *done = false;
    if (!part_wanted(b.n, kk, 4)) return PSK_OK;
    PartGeom g;
    // 2^15 counters = 128 KiB per slice.  Beyond 2048 such slices (2^26 counters: a CBF for 7 M elements at 1 %) and up to 2^27
    // counters: slices of 2^16 counters held as 16-bit values (k_counter_gather<true>; a counter at or above 2^16 raises the redo
    // flag).  Larger tables: direct kernels.
    bool half = false;
    if (!part_slices(cells, 15, 5, &g, kPartMaxBuckets, 7)) {
        if (g_lookup_half == 0 || !part_slices(cells, 16, 16, &g, kPartMaxBuckets, 7)) return PSK_OK;
        half = true;
    }
    g.k = kk;
    const uint64_t round_keys = lookup_round_keys(b.n, kk);
    PSK_TRY(ensure(s->s_flag, 8));
    uint32_t *flag = (uint32_t *)s->s_flag.p;
    HIP_TRY(hipMemsetAsync(flag, 0, 4, st));
    for (uint64_t start = 0; start < b.n; start += round_keys) {
        const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
        const Batch sub = sub_batch(b, start, cnt);
        bool handled = false, fits = true;
        PSK_TRY(with_part_source(sub, &handled, [&](auto src) {
            using Src = decltype(src);
            return with_kt<Src>(kk, [&](auto kt) {
                constexpr int KT = decltype(kt)::value;
                constexpr int P4 = (KT + 7) / 8;
                using TileSmall = PartTile<PayUnitLookup, KT, kPartThreads>;  // the smaller of the two tile sizes bounds the tile count
                // 16-bit stage positions (perm[]): the largest tile pass 1 may choose must fit -- else not eligible (direct kernels),
                // decided BEFORE anything is enqueued
                {
                    const size_t tile_max = PartTile<PayUnitLookup, KT, 1024>::TILE > TileSmall::TILE ? PartTile<PayUnitLookup, KT, 1024>::TILE : TileSmall::TILE;
                    const uint32_t kq0 = g.k < (uint32_t)KT ? g.k : (uint32_t)KT;
                    if (tile_max * kq0 + (size_t)7 * g.nbuckets + 3 > 0xFFFFu) { fits = false; return (int)PSK_OK; }
                }
                const uint64_t max_tiles = (cnt + TileSmall::TILE - 1) / TileSmall::TILE + 1024;  // (+ workgroups: evened tiles, launch_scatter_nt)
                PSK_TRY(ensure(s->s_perm, cnt * (uint64_t)PermRec<KT>::PD * 4 + 16));
                PSK_TRY(ensure(s->s_run, max_tiles * g.nbuckets * 8));
                PayUnitLookup pay{(uint32_t *)s->s_perm.p, (uint2 *)s->s_run.p};
                SpillRaiseFlag spill{flag};
                PSK_TRY((launch_scatter<Src, IDX<kTuPow2>, PayUnitLookup, SpillRaiseFlag, KT>(s, src, IDX<kTuPow2>{s->md}, pay, spill, &g, cnt, st)));
                // pass 2: the counters behind every probe, in the probe buffer's shape
                const uint64_t vals_bytes = (uint64_t)g.nbuckets * g.nwg * g.segcap * 32;
                PSK_TRY(ensure(s->s_vals, vals_bytes + g.nbuckets + 256));  // values + one format byte per slice
                uint8_t *fmt = (uint8_t *)s->s_vals.p + vals_bytes;
                const size_t lds2 = (size_t)(half ? 2 : 4) << g.shift;
                auto gather = half ? k_counter_gather<true> : k_counter_gather<false>;
                PSK_TRY(set_dyn_lds(gather, lds2));
                // slice counts that do not fill the 256 CUs evenly: two workgroups share a slice's segments (each loads the slice)
                PartGeom g2 = g;
                // Round 4 (same-box A/B over four CMS geometries, scripts/ab_cms_check_lib.py): up to 256 slices the split is the largest power
                // of two that still fits ONE wave of workgroups (32 slices: 8 per slice; 160 slices, BASELINE cfg 3: none -- two per slice
                // were 320 workgroups, a second wave of 64); between 256 and 1024 slices two per slice as before.
                g2.split = 1;
                if (g_lookup_split != 0) {
                    if (g.nbuckets <= 256) {
                        while (g2.split < 8 && g.nbuckets * g2.split * 2 <= 256) g2.split *= 2;
                    } else if (g.nbuckets % 256 != 0 && g.nbuckets < 1024) {
                        g2.split = 2;
                    }
                }
                hipLaunchKernelGGL(gather, dim3(g.nbuckets * g2.split), dim3(kApplyThreads), lds2, st, (const uint32_t *)s->table, cells, g2,
                                   (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p, (uint4 *)s->s_vals.p, fmt, flag);
                HIP_TRY(hipGetLastError());
                // pass 3: back to key order, query epilogue
                const uint32_t kq = g.k < (uint32_t)KT ? g.k : (uint32_t)KT;
                const uint32_t stage_cap = (uint32_t)(((size_t)g.tile * kq + (size_t)7 * g.nbuckets + 3) & ~(size_t)3);
                const size_t lds3 = ((size_t)2 * g.nbuckets + stage_cap) * 4 + ((g.nbuckets + 15) & ~(size_t)15);
                const uint64_t ntiles = (cnt + g.tile - 1) / g.tile;
                // pass 3 as 512-thread workgroups (four tiles in flight per CU) pays up to 256 slices (cfg 3: 270 -> 259 us per 10 M lookups) and
                // costs 15 % at 1024 (twice the runinfo reads per key): option "lookup_collect_threads" 0 = this rule, 512 / 1024 = forced
                const bool narrow_wg = g_lookup_collect_threads == 512 || (g_lookup_collect_threads == 0 && g.nbuckets <= 256);
                auto kern = narrow_wg ? k_lookup_collect<Query, KT, 512> : k_lookup_collect<Query, KT, 1024>;
                PSK_TRY(set_dyn_lds(kern, lds3));
                // lanes that copy one (tile, slice) run of values: the power of two at or above HALF the mean run -- a lane moves two
                // 16-bit values at a time, the usual format; runs of 32-bit values take a second trip through the loop
                uint32_t run_lanes = 4;
                while (run_lanes < 64 && (uint64_t)run_lanes * 2 * g.nbuckets < (uint64_t)g.tile * kq) run_lanes *= 2;
                if (g_lookup_run_lanes > 0) run_lanes = (uint32_t)g_lookup_run_lanes;
                const uint64_t grid3 = narrow_wg ? 1024 : 512;
                hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < grid3 ? ntiles : grid3)), dim3(narrow_wg ? 512 : 1024), lds3, st, query, g, cnt,
                                   (const uint32_t *)s->s_perm.p, (const uint2 *)s->s_run.p, (const uint32_t *)s->s_vals.p, (const uint8_t *)fmt, stage_cap, run_lanes,
                                   out_dev + start);
                HIP_TRY(hipGetLastError());
                return (int)PSK_OK;
            });
        }));
        if (!handled || !fits) return PSK_OK;  // layout without a partitioned instantiation / tile too large: nothing was launched (first round)
    }
    PSK_TRY(redo(flag, st));  // runs only if a segment overflowed (device-side flag): exact for any input
    *done = true;
    return PSK_OK;
}

// CountingBloomFilter lookups through NIBBLE slices (psk_nibble.hpp): tables of 2^25 .. 2^29 counters -- beyond 1024 of the 32-bit
// slices above the three passes below win (fewer, longer runs in pass 1 and pass 3), and beyond 2^27 counters they are the only
// partitioned form (BASELINE cfg 4's 1 GiB table: 1024 slices of 2^18 counters).  countingbloom.py:166-174.
// redo(flag, st): the whole batch through the direct kernel if a pass-1 segment overflowed; recheck(amb, st): the keys whose nibble answer is
// 15 from the table itself (k_cbf_recheck15)
template <class Redo, class Recheck>
static inline int cbf_check_nibble(psk_sketch *s, const Batch &b, uint32_t kk, uint32_t *out_dev, hipStream_t st, bool *done, Redo &&redo, Recheck &&recheck)
{
    *done = false;
    const uint64_t cells = s->m;
    if (g_lookup_nibble == 0 || !part_wanted(b.n, kk, 4)) return PSK_OK;
    // pass 2 reads the WHOLE table (4 B per counter at ~4.4 TB/s: 0.25 ms for 2^28 counters) where the direct kernel fetches one
    // 64-byte line per probe (52 G probes/s): worth it from about cells / 16 probes on (measured on the 1 GiB table: a 0.5 M-key
    // lookup 67 us direct, 0.4 ms through the slices)
    PartGeom g;
    if (!nib_geometry(cells, false, &g)) return PSK_OK;
    g.k = kk;
    const uint64_t shadow_words = (uint64_t)g.nbuckets << (g.shift - 3);
    // (with the table's 4-bit images at hand -- psk_sketch::shadow, below -- the pass reads 1/8 of that: from cells / 64 probes on;
    // measured on the 1 GiB table: 1 M keys 134 us direct, ~75 us through kept images)
    const bool shadow_ready = g_cbf_shadow != 0 && s->shadow.allow && s->shadow.built == s->table_version && s->shadow.words == shadow_words &&
                              s->shadow.stream == st && s->shadow.img.p != nullptr;
    const bool may_shadow = g_cbf_shadow != 0 && s->shadow.allow;
    if (may_shadow) {  // lookups in a row that found this version of the table (this one included)
        if (s->shadow.seen == s->table_version) ++s->shadow.seen_count;
        else s->shadow.seen = s->table_version, s->shadow.seen_count = 1;
    }
    const uint64_t probes = b.n * (uint64_t)kk;
    if (g_lookup_nibble != 2) {  // (2: always -- tests)
        // below the crossover of the plain pass a batch takes the slices only through kept images -- or, the third such lookup in a
        // row of an unchanged table, to leave them behind (one pass over the table, repaid by the lookups that follow)
        if (probes < cells / 64) return PSK_OK;
        if (probes < cells / 16 && !shadow_ready && !(may_shadow && s->shadow.seen_count >= 3)) return PSK_OK;
    }
    const uint64_t round_keys = lookup_round_keys(b.n, kk);
    PSK_TRY(ensure(s->s_flag, 8));
    uint32_t *flag = (uint32_t *)s->s_flag.p, *amb = flag + 1;  // [0] a segment overflowed, [1] a key's answer is ambiguous (15)
    HIP_TRY(hipMemsetAsync(flag, 0, 8, st));
    // The 4-bit images of an unchanged table (psk_sketch::shadow): loaded when they mirror this version of the table; left behind
    // when they will be read again -- a second round of this call, or the second lookup in a row that finds the table unchanged.
    bool shadow_used = false;
    const uint32_t *shadow_in = nullptr;
    uint32_t *shadow_out = nullptr;
    if (g_cbf_shadow != 0 && s->shadow.allow) {
        if (shadow_ready) {
            shadow_in = (const uint32_t *)s->shadow.img.p;
        } else if (b.n > round_keys || s->shadow.seen_count >= 2) {
            // (an allocation failure only costs the shortcut)
            if (s->shadow.img.cap >= shadow_words * 4 || ensure(s->shadow.img, shadow_words * 4) == PSK_OK) shadow_out = (uint32_t *)s->shadow.img.p;
            s->shadow.built = ~0ULL;  // until the first round below has been enqueued
        }
    }
    for (uint64_t start = 0; start < b.n; start += round_keys) {
        const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
        const Batch sub = sub_batch(b, start, cnt);
        bool handled = false, fits = true;
        PSK_TRY(PSK_VARIANT(bloomidx_lookup_scatter)(s, sub, cnt, kk, &g, flag, st, &handled, &fits));  // pass 1 (perm[] / runinfo[] by-products)
        if (!handled || !fits) return PSK_OK;  // (only ever on the first round: nothing was launched)
        PSK_TRY(with_part_source(sub, &handled, [&](auto src) {
            using Src = decltype(src);
            return with_kt<Src>(kk, [&](auto kt) {
                constexpr int KT = decltype(kt)::value;
                const uint32_t kq = g.k < (uint32_t)KT ? g.k : (uint32_t)KT;
                PSK_TRY(ensure(s->s_vals, (uint64_t)g.nbuckets * g.nwg * g.segcap * 4 + 256));  // one dword (six nibbles) per group
                const size_t lds2 = (size_t)1 << (g.shift - 1);
                if (g_nib_gather_pipe != 0 && shadow_in == nullptr && g.shift >= 15) {
                    // round 4: no kept images to load -- the pipelined pass (psk_nibble_pipe.hpp: the next slice's table load under this
                    // slice's probe walk); option "nibble_lookup_pipe" (0 = k_nib_gather, the A/B partner)
                    static int ncu = 0;
                    if (ncu == 0) {
                        int dev = 0, v = 0;
                        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
                        ncu = v;
                    }
                    auto kp = k_nib_gather_pipe<0>;
                    PSK_TRY(set_dyn_lds(kp, lds2));
                    const uint32_t grid = g.nbuckets < (uint32_t)ncu ? g.nbuckets : (uint32_t)ncu;
                    hipLaunchKernelGGL(kp, dim3(grid), dim3(kApplyThreads), lds2, st, (const uint32_t *)s->table, cells, g, (const uint32_t *)s->s_cnt.p,
                                       (const uint4 *)s->s_part.p, (uint32_t *)s->s_vals.p, (uint32_t)(g_nib_nt != 0), shadow_out);
                } else {
                    PSK_TRY(set_dyn_lds(k_nib_gather, lds2));
                    hipLaunchKernelGGL(k_nib_gather, dim3(g.nbuckets), dim3(kApplyThreads), lds2, st, (const uint32_t *)s->table, cells, g,
                                       (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p, (uint32_t *)s->s_vals.p, (uint32_t)(g_nib_nt != 0), shadow_in,
                                       shadow_out);
                }
                HIP_TRY(hipGetLastError());
                shadow_used = shadow_used || shadow_in != nullptr;
                if (shadow_out) {  // the images are complete behind this launch: the next round / lookup on this stream loads them
                    s->shadow.built = s->table_version;
                    s->shadow.words = shadow_words;
                    s->shadow.stream = st;
                    shadow_in = shadow_out;
                    shadow_out = nullptr;
                }
                const uint32_t stage_cap = (uint32_t)(((size_t)g.tile * kq + (size_t)5 * g.nbuckets + 3) & ~(size_t)3);
                const uint32_t stage_groups = stage_cap / 6 + 1;
                const size_t lds3 = (size_t)8 * g.nbuckets + (size_t)4 * ((stage_groups + 3) & ~3u);
                const uint64_t ntiles = (cnt + g.tile - 1) / g.tile;
                auto kern = k_nib_collect<KT>;
                PSK_TRY(set_dyn_lds(kern, lds3));
                uint32_t run_lanes = 2;  // lanes (one dword = one group of 6 probes each) per (tile, slice) run
                while (run_lanes < 64 && (uint64_t)run_lanes * 6 * g.nbuckets < (uint64_t)g.tile * kq + 6ULL * g.nbuckets) run_lanes *= 2;
                if (g_lookup_run_lanes > 0) run_lanes = (uint32_t)g_lookup_run_lanes;
                hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < 512 ? ntiles : 512)), dim3(kBloomCollectThreads), lds3, st, g, cnt, (const uint32_t *)s->s_perm.p,
                                   (const uint2 *)s->s_run.p, (const uint32_t *)s->s_vals.p, stage_groups, run_lanes, out_dev + start, amb);
                HIP_TRY(hipGetLastError());
                return (int)PSK_OK;
            });
        }));
        if (!handled || !fits) return PSK_OK;  // (only ever on the first round: nothing was launched)
    }
    if (shadow_used) ++g_cbf_shadow_hits;  // (calls that loaded the images; option "cbf_lookup_shadow_hits": tests)
    PSK_TRY(recheck(amb, st));  // keys whose counters are all 15 or more: answered from the table itself (flag-guarded)
    PSK_TRY(redo(flag, st));     // a segment overflowed: exact redo of the batch by the direct kernel (flag-guarded)
    *done = true;
    return PSK_OK;
}
