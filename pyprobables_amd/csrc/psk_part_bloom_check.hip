// partitioned Bloom lookup launcher (own translation unit: parallel build)
#include "psk_host.hpp"
#include "psk_lookup.hpp"

// pass 1 of one round: keyed probes of keys [0, cnt) of `sub` into the bucket buffer.  defer != nullptr: split lookup
// (the table is not consulted; an overflowing segment raises *defer instead of testing its probes directly)
// Pass 1 workgroups of the keyed lookups.  A round holds 16 tiles per workgroup (PayKeyId) and pass 2 reads the whole table
// once per round: with 2048 slices (m = 2^31: 256 MiB per round) four times the workgroups -- 33.5 M keys per round -- is worth
// +13 % (256 workgroups 21.0, 512: 22.7, 1024: 23.8 G keys/s; scripts/ab_wgs.py, scripts/ab_2p31_lookup.py); at 1024 slices and
// below it measures the same or slightly worse.
static inline uint32_t keyed_wgs(const PartGeom &g) { return g.nbuckets >= 2048 ? 1024u : 0u; }  // 0: launch_scatter's default

static int check_round_scatter(psk_sketch *s, const Batch &sub, uint64_t cnt, uint8_t *out, uint32_t *defer, PartGeom *g,
                               hipStream_t st, bool *handled)
{
    return with_part_source(sub, handled, [&](auto src) {
        using Src = decltype(src);
        return with_kt<Src>(s->k, [&](auto kt) {
            constexpr int KT = decltype(kt)::value;
            SpillBloomTest spill{(const uint32_t *)s->table, out, defer};
            return launch_scatter<Src, IdxBloom<kTuPow2>, PayKeyId, SpillBloomTest, KT>(s, src, IdxBloom<kTuPow2>{s->md}, PayKeyId{},
                                                                                         spill, g, cnt, st, keyed_wgs(*g));
        });
    });
}

// pass 2 of one round: any probe that finds its bit clear stores a 0 (out[] pre-set to 1 by the caller)
static int check_round_test(psk_sketch *s, const PartGeom &g, uint8_t *out, hipStream_t st, unsigned long long *miss_ctr = nullptr)
{
    const size_t lds = (size_t)1 << (g.shift - 3);
    PSK_TRY(set_dyn_lds(k_bloom_test, lds));
    hipLaunchKernelGGL(k_bloom_test, dim3(g.nbuckets), dim3(kApplyThreads), lds, st, (const uint32_t *)s->table, s->padded_bytes / 4, g,
                       (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p, out, miss_ctr);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

static bool check_geometry(psk_sketch *s, const Batch &b, PartGeom *g, uint64_t *round_keys)
{
    const uint64_t n = b.n;
    if (!part_wanted(n, s->k, 4)) return false;
    if (!part_slices(s->m, 20, 7, g)) return false;
    g->k = s->k;
    uint64_t rk = part_round_keys_big_table(n, s->k, PayKeyId::group, s->padded_bytes);
    // a keyed group spells the tile's ordinal inside its workgroup in 4 bits: at most 16 tiles per workgroup and round
    // (256 workgroups x 16 x 2048-key tiles for k <= 8; 512-key tiles beyond)
    uint64_t cap = (uint64_t)PayKeyId::max_tiles_per_wg * (g_part_wgs > 0 ? (uint64_t)(g_part_wgs < 1024 ? g_part_wgs : 1024) : (keyed_wgs(*g) ? keyed_wgs(*g) : 256u)) *
                   (s->k <= 8 ? (g_part_tile_threads == 512 ? 1024 : 2048) : 512);  // (forced 512-thread tiles hold 1024 keys at k = 7, 8)
    // ... of the tile launch_scatter_nt will really run: it cuts the tile where the LDS stage -- plus the per-tile length sort of ragged keys --
    // would not fit (e.g. 1280 keys at 2048 slices), and a round sized for 2048-key tiles would then need more than 16 tiles per workgroup
    bool handled = false;
    uint64_t cap_layout = cap;
    (void)with_part_source(b, &handled, [&](auto src) {
        using Src = decltype(src);
        return with_kt<Src>(s->k, [&](auto kt) {
            constexpr int KT = decltype(kt)::value;
            cap_layout = scatter_round_cap<PayKeyId, KT, src_fat512<Src>::value, src_sorted<Src>::value>(g, keyed_wgs(*g), PayKeyId::max_tiles_per_wg);
            return (int)PSK_OK;
        });
    });
    if (handled && cap_layout < cap) cap = cap_layout;
    if (rk > cap) rk = cap;
    *round_keys = rk;
    return true;
}

// Lookups with a return trip (psk_lookup.hpp): pass 1 with perm / runinfo, k_bloom_gather, k_bloom_collect.  The cost does not
// depend on how many probes miss; option "bloom_lookup" = 0 selects the keyed kernels below instead (A/B, split lookups).
static int bloom_check_return_trip(psk_sketch *s, const Batch &b, uint8_t *out_dev, hipStream_t st, bool *done)
{
    *done = false;
    if (!part_wanted(b.n, s->k, 4)) return PSK_OK;
    PartGeom g;
    if (!part_slices(s->m, 20, 7, &g)) return PSK_OK;
    g.k = s->k;
    const uint64_t round_keys = (uint64_t)g_part_max_keys < b.n ? (uint64_t)g_part_max_keys : b.n;
    PSK_TRY(ensure(s->s_flag, 8));
    uint32_t *flag = (uint32_t *)s->s_flag.p;
    HIP_TRY(hipMemsetAsync(flag, 0, 4, st));
    for (uint64_t start = 0; start < b.n; start += round_keys) {
        const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
        const Batch sub = sub_batch(b, start, cnt);
        bool handled = false, fits = true;
        // pass 1 with the perm[] / runinfo[] by-products (shared with the CountingBloomFilter's 4-bit-slice lookups); 16-bit stage positions:
        // decided before anything is enqueued (else: not eligible, the keyed kernels take the batch)
        PSK_TRY(PSK_VARIANT(bloomidx_lookup_scatter)(s, sub, cnt, s->k, &g, flag, st, &handled, &fits));
        if (!handled || !fits) return PSK_OK;
        PSK_TRY(with_part_source(sub, &handled, [&](auto src) {
            using Src = decltype(src);
            return with_kt<Src>(s->k, [&](auto kt) {
                constexpr int KT = decltype(kt)::value;
                PSK_TRY(ensure(s->s_vals, (uint64_t)g.nbuckets * g.nwg * g.segcap + 256));  // one result byte per group
                const size_t lds2 = (size_t)1 << (g.shift - 3);
                PSK_TRY(set_dyn_lds(k_bloom_gather, lds2));
                hipLaunchKernelGGL(k_bloom_gather, dim3(g.nbuckets), dim3(kApplyThreads), lds2, st, (const uint32_t *)s->table, s->padded_bytes / 4, g,
                                   (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p, (uint8_t *)s->s_vals.p);
                HIP_TRY(hipGetLastError());
                const uint32_t kq = g.k < (uint32_t)KT ? g.k : (uint32_t)KT;
                const uint32_t stage_cap = (uint32_t)(((size_t)g.tile * kq + (size_t)5 * g.nbuckets + 3) & ~(size_t)3);
                const uint32_t stage_groups = stage_cap / 6 + 1;
                const size_t lds3 = (size_t)8 * g.nbuckets + ((stage_groups + 15) & ~(size_t)15);
                const uint64_t ntiles = (cnt + g.tile - 1) / g.tile;
                auto kern = k_bloom_collect<KT>;
                PSK_TRY(set_dyn_lds(kern, lds3));
                uint32_t run_lanes = 2;  // lanes (one byte = one group of 6 probes each) per (tile, slice) run
                while (run_lanes < 64 && (uint64_t)run_lanes * 6 * g.nbuckets < (uint64_t)g.tile * kq + 6ULL * g.nbuckets) run_lanes *= 2;
                hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < 512 ? ntiles : 512)), dim3(kBloomCollectThreads), lds3, st, g, cnt, (const uint32_t *)s->s_perm.p,
                                   (const uint2 *)s->s_run.p, (const uint8_t *)s->s_vals.p, stage_groups, run_lanes, out_dev + start, s->lk.dev);
                HIP_TRY(hipGetLastError());
                return (int)PSK_OK;
            });
        }));
        if (!handled || !fits) return PSK_OK;
    }
    // exact redo through the direct kernel, taken on the device only if a segment overflowed
    bool handled = false;
    PSK_TRY(with_part_source(b, &handled, [&](auto src) {
        using Src = decltype(src);
        using Op = BloomCheck<kTuPow2>;
        hipLaunchKernelGGL((k_apply_if<Src, Op>), dim3(grid_for_keys(b.n)), dim3(kBlock), 0, st, (const uint32_t *)flag, src,
                           Op{(const uint32_t *)s->table, s->md, s->k, out_dev}, b.n);
        HIP_TRY(hipGetLastError());
        return (int)PSK_OK;
    }));
    *done = true;
    return PSK_OK;
}

// Tile-flag lookups (round 5; PayTileTag, k_bloom_test_flag, k_bloom_flag_resolve): for batches whose keys are (nearly) all present.  Pass 1 and
// the probe stream are the insert's (2.67-byte probes, two 512-thread workgroups per CU, one round of up to 16 tiles per workgroup); pass 2
// answers every key "present" and raises a flag per TILE that met a clear bit; the resolving kernel re-checks the keys of flagged tiles
// against the table.  Exact for any batch; a batch with absent keys in most tiles costs the direct kernel's gathers on top -- the automatic choice
// (choose_scheme) only comes here while the previous lookups on the handle missed (almost) nothing.
// geometry and round size of a tile-flag lookup of b; false: not eligible
static bool tile_flag_geometry(psk_sketch *s, const Batch &b, PartGeom *g, uint64_t *round_keys_out)
{
    if (!part_wanted(b.n, s->k, 4)) return false;
    if (!part_slices(s->m, 20, 7, g)) return false;
    g->k = s->k;
    uint64_t round_keys = part_round_keys_big_table(b.n, s->k, PayTileTag::group, s->padded_bytes);
    // 4 bits of tile ordinal per group: at most 16 tiles per pass-1 workgroup and round
    bool handled = false;
    uint64_t cap = 0;
    if (with_part_source(b, &handled, [&](auto src) {
            using Src = decltype(src);
            return with_kt<Src>(s->k, [&](auto kt) {
                constexpr int KT = decltype(kt)::value;
                // More pass-1 workgroups than the chip runs at once (they follow one another) where 16 tiles each would cut the batch into
                // more rounds: layouts without the two-per-CU shape get 512 (a 10 M-key batch stays ONE round as it is for the 16-byte
                // layout -- two rounds of 5 M cost ragged keys 30 us per lookup), and big tables, whose every round sweeps the WHOLE table in
                // pass 2, up to 1024 (m = 2^31: a 2^25-key lookup was four rounds, four 256 MiB sweeps of 127 us)
                uint32_t want = src_fat512<Src>::value ? 0u : 512u;
                cap = scatter_round_cap<PayTileTag, KT, src_fat512<Src>::value, src_sorted<Src>::value>(g, want, PayTileTag::max_tiles_per_wg);
                if (s->padded_bytes >= (64ULL << 20) && g_part_wgs <= 0) {
                    while (round_keys > cap && want < 1024u) {
                        want = want ? want * 2 : 512u;
                        const uint64_t c2 = scatter_round_cap<PayTileTag, KT, src_fat512<Src>::value, src_sorted<Src>::value>(g, want, PayTileTag::max_tiles_per_wg);
                        if (c2 <= cap) break;
                        cap = c2;
                    }
                }
                s->tflag_wgs = want;
                return (int)PSK_OK;
            });
        }) != PSK_OK || !handled || cap == 0) return false;
    if (round_keys > cap) {  // equal rounds
        const uint64_t rounds = (b.n + cap - 1) / cap;
        round_keys = ((b.n + rounds - 1) / rounds + 4095) & ~4095ULL;
        if (round_keys > cap) round_keys = cap;
    }
    *round_keys_out = round_keys;
    return true;
}

// pass 1 of one round (keys [0, cnt) of `sub`); defer: split lookup, the table is not consulted
static int tile_flag_scatter(psk_sketch *s, const Batch &sub, uint64_t cnt, bool defer, PartGeom *g, uint32_t *gen_out, hipStream_t st, bool *handled)
{
    return with_part_source(sub, handled, [&](auto src) {
        using Src = decltype(src);
        return with_kt<Src>(s->k, [&](auto kt) {
            constexpr int KT = decltype(kt)::value;
            // one flag per tile; tiles are at least 64 keys (evened tiles are multiples of 64)
            const uint64_t flag_bytes = (cnt / 64 + 2048) * 4;
            if (flag_bytes > s->s_tflag.cap || s->tflag_gen == 0xFFFFFFFFu) {
                PSK_TRY(ensure(s->s_tflag, flag_bytes));
                HIP_TRY(hipMemsetAsync(s->s_tflag.p, 0, s->s_tflag.cap, st));  // once (and when the generation number wraps)
                s->tflag_gen = 0;
            }
            const uint32_t gen = ++s->tflag_gen;
            *gen_out = gen;
            SpillBloomFlag spill{(const uint32_t *)s->table, (uint32_t *)s->s_tflag.p, gen, defer ? 1u : 0u};
            return launch_scatter<Src, IdxBloom<kTuPow2>, PayTileTag, SpillBloomFlag, KT>(s, src, IdxBloom<kTuPow2>{s->md}, PayTileTag{}, spill, g, cnt, st,
                                                                                          s->tflag_wgs);
        });
    });
}

// pass 2 + the re-check of flagged tiles for one scattered round; publish_units != 0: last round of a call, the tally goes to the pinned page
static int tile_flag_test(psk_sketch *s, const Batch &sub, uint64_t cnt, const PartGeom &g, uint32_t gen, uint8_t *out, unsigned long long publish_units, hipStream_t st)
{
    bool handled = false;
    PSK_TRY(with_part_source(sub, &handled, [&](auto src) {
        using Src = decltype(src);
        uint32_t *tflag = (uint32_t *)s->s_tflag.p;
        const size_t lds = (size_t)1 << (g.shift - 3);
        PSK_TRY(set_dyn_lds(k_bloom_test_flag, lds));
        hipLaunchKernelGGL(k_bloom_test_flag, dim3(g.nbuckets), dim3(kApplyThreads), lds, st, (const uint32_t *)s->table, s->padded_bytes / 4, g,
                           (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p, tflag, gen, s->lk.dev, out, cnt);
        HIP_TRY(hipGetLastError());
        const uint64_t ntiles = (cnt + g.tile - 1) / g.tile;
        LookupPublish pub;
        if (publish_units && g_bloom_lookup == 2 && s->lk.dev) pub = LookupPublish{s->lk.dev, s->lk.pin, publish_units, 3};
        hipLaunchKernelGGL((k_bloom_flag_resolve<Src, kTuPow2>), dim3((unsigned)(ntiles < 1024 ? ntiles : 1024)), dim3(kResolveThreads), 0, st, src,
                           (const uint32_t *)s->table, s->md, s->k, (const uint32_t *)tflag, gen, g.tile, cnt, out, pub);
        HIP_TRY(hipGetLastError());
        return (int)PSK_OK;
    }));
    return handled ? (int)PSK_OK : fail(PSK_EHIP, "tile-flag lookup: layout lost its partitioned instantiation");
}

// Tile-flag lookups (round 5; PayTileTag, k_bloom_test_flag, k_bloom_flag_resolve): for batches whose keys are (nearly) all present.  Pass 1 and
// the probe stream are the insert's (2.67-byte probes, two 512-thread workgroups per CU, one round of up to 16 tiles per workgroup); pass 2
// answers every key "present" and raises a flag per TILE that met a clear bit; the resolving kernel re-checks the keys of flagged tiles
// against the table.  Exact for any batch; a batch with absent keys in most tiles costs the direct kernel's gathers on top -- the automatic
// choice (choose_scheme) only comes here while the previous lookups on the handle missed (almost) nothing.
static int bloom_check_tile_flags(psk_sketch *s, const Batch &b, uint8_t *out_dev, hipStream_t st, bool *done)
{
    *done = false;
    PartGeom g;
    uint64_t round_keys = 0;
    if (!tile_flag_geometry(s, b, &g, &round_keys)) return PSK_OK;
    for (uint64_t start = 0; start < b.n; start += round_keys) {
        const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
        const Batch sub = sub_batch(b, start, cnt);
        bool handled = false;
        uint32_t gen = 0;
        PSK_TRY(tile_flag_scatter(s, sub, cnt, false, &g, &gen, st, &handled));
        if (!handled) return start == 0 ? (int)PSK_OK : fail(PSK_EHIP, "tile-flag lookup: layout lost its partitioned instantiation");
        PSK_TRY(tile_flag_test(s, sub, cnt, g, gen, out_dev + start, start + cnt == b.n ? b.n * (uint64_t)s->k : 0ULL, st));
    }
    *done = true;
    return PSK_OK;
}

// Lazy gathers (scheme 4, round 5): lookups of keys that are (nearly) all ABSENT -- what a Bloom filter is asked most often (a deduplicating
// caller, a cache in front of a store).  bloom.py:261-272 stops at the first clear bit, and so does a lane here: one key per lane, the word of
// probe j is requested only by the lanes still undecided, chain j + 1 is hashed while that gather is in flight, and a wave leaves the key
// when none of its lanes is left.  A key absent from a table with a share f of its bits set costs 1 / (1 - f) gathers (1.3 at cfg 2's 23 %)
// instead of the k probes + the way back of the partitioned schemes; every gather is a 64-byte line across the fabric (~63 G/s), so the
// scheme only pays while the gathers per key stay below ~1.7 -- the kernel tallies the gathers it issued and choose_scheme goes by them.
// H32: power-of-two tables up to 2^32 bits, the 32-bit chains.
template <class Src, bool POW2, bool H32>
__global__ __launch_bounds__(kBlock) void k_bloom_check_lazy(Src src, const uint32_t *tab, Mod md, uint32_t k, uint64_t n, uint8_t *out,
                                                             unsigned long long *gather_ctr)
{
    auto bit_of = [&](const typename Src::Key &key, uint64_t i, uint32_t j) -> uint64_t {
        if constexpr (H32) {
            uint32_t h[1];
            src.template hash32<1>(key, i, j, h);
            return (uint64_t)(h[0] & (uint32_t)md.mask);
        } else {
            uint64_t h[1];
            src.template hash<1>(key, i, j, h);
            return reduce<POW2>(md, h[0]);
        }
    };
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t nround = (n + 63) & ~63ULL;  // wave-uniform trip count
    uint32_t ngather = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nround; i += stride) {
        const bool mine = i < n;
        const uint64_t ii = mine ? i : n - 1;
        const typename Src::Key key = src.load(ii);
        bool alive = mine;
        uint64_t bit = bit_of(key, ii, 0);
        for (uint32_t j = 0; j < k; ++j) {
            uint32_t word = 0;
            if (alive) {
                word = tab[bit >> 5];
                ++ngather;
            }
            const uint32_t sh = (uint32_t)bit & 31u;
            if (j + 1 < k) bit = bit_of(key, ii, j + 1);  // (under the gather)
            alive = alive && ((word >> sh) & 1u) != 0;
            if (__ballot(alive) == 0) break;               // (uniform)
        }
        if (mine) out[i] = alive ? 1 : 0;
    }
    if (gather_ctr) {  // one atomic per workgroup
        __shared__ uint32_t wsum[kBlock / 64];
        for (int o = 32; o > 0; o >>= 1) ngather += __shfl_down(ngather, o);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ngather;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < kBlock / 64; ++w) t += wsum[w];
            if (t) atomicAdd(gather_ctr, t);
        }
    }
}

static int bloom_check_lazy(psk_sketch *s, const Batch &b, uint8_t *out_dev, hipStream_t st, bool *done)
{
    *done = false;
    if (b.n == 0) return PSK_OK;
    bool handled = false;
    PSK_TRY(with_part_source(b, &handled, [&](auto src) {
        using Src = decltype(src);
        const dim3 grid((unsigned)grid_for_keys(b.n)), block(kBlock);
        unsigned long long *ctr = g_bloom_lookup == 2 ? s->lk.dev : nullptr;
        if constexpr (kTuPow2) {
            if (s->m <= (1ULL << 32))
                hipLaunchKernelGGL((k_bloom_check_lazy<Src, true, true>), grid, block, 0, st, src, (const uint32_t *)s->table, s->md, s->k, b.n, out_dev, ctr);
            else
                hipLaunchKernelGGL((k_bloom_check_lazy<Src, true, false>), grid, block, 0, st, src, (const uint32_t *)s->table, s->md, s->k, b.n, out_dev, ctr);
        } else {
            hipLaunchKernelGGL((k_bloom_check_lazy<Src, false, false>), grid, block, 0, st, src, (const uint32_t *)s->table, s->md, s->k, b.n, out_dev, ctr);
        }
        HIP_TRY(hipGetLastError());
        return (int)PSK_OK;
    }));
    *done = handled;
    return PSK_OK;
}

// Which scheme?  Keyed probes cost ~240 us per 10 M all-hit keys but one scattered byte store per probe that misses (~450 us
// when every key is absent); the return trip costs ~300 us whatever the answers.  Mode 2 follows what the previous large
// lookups on this handle saw: the tally of the last finished call sits in a pinned page (no synchronisation: it may be one
// call late, and the very first call is keyed).
static int choose_scheme(psk_sketch *s, hipStream_t st)
{
    if (g_bloom_lookup != 2) return (int)g_bloom_lookup;   // forced: 0 keyed, 1 return trip, 3 tile flags, 4 lazy gathers
    if (!s->lk.dev) {
        HIP_TRY(hipMalloc((void **)&s->lk.dev, 8));
        void *pin = nullptr;
        HIP_TRY(hipHostMalloc(&pin, 64, hipHostMallocDefault));  // [0..3] written by the device (psk_sketch::lk), [4] [5] host-only (lazy gathers, below)
        s->lk.pin = (volatile unsigned long long *)pin;
        for (int e = 0; e < 8; ++e) s->lk.pin[e] = 0;
        HIP_TRY(hipMemsetAsync(s->lk.dev, 0, 8, st));  // once: k_lookup_publish re-zeroes the tally at the end of every call
    }
    const unsigned long long miss = s->lk.pin[0], units = s->lk.pin[1], by = s->lk.pin[2], seq = s->lk.pin[3];
    if (units) {
        const double f = (double)miss / (double)units;
        // consecutive finished calls that missed (almost) nothing; an alternating hit / miss workload never gets to the tile flags
        const bool clean_call = by == 1 ? f == 0.0 : f <= 1e-6;
        if (seq != s->lk.seen_seq) {
            s->lk.seen_seq = seq;
            s->lk.clean = clean_call ? s->lk.clean + 1 : 0;
        }
        const bool flags_ok = clean_call && s->lk.clean >= 2;
        // keyed (0) and tile flags (3) tally PROBES that missed, the return trip (1) KEYS answered absent (each misses one probe or more).
        // Tile flags pay a direct re-check of every ~2000-key tile that holds one miss: only while (almost) nothing misses -- one probe
        // in a million flags ~1 % of the tiles.  Keyed probes pay a scattered byte store per miss: beyond ~1/4 of the probes the return
        // trip, whose cost does not depend on the answers, is cheaper.
        // Lazy gathers (4) tally the GATHERS they issued (per key: 1 / (1 - fill) for an absent key, k for a present one): they stay while a
        // key costs at most kLazyStay gathers (what the return trip costs at this table's slice count) and are entered
        // from the return trip, the one scheme that counts absent KEYS, when (nearly) every key was absent; a handle whose table is too full
        // for them (gathers per key above the bound although every key is absent) is not tried again for kLazyBackoff calls.
        // (scripts/ab_bloom_lookup_r05.py, profiles/r05_ab_bloom_lookup_lazy.txt: a gather costs ~17.4 ns of kernel time at any table
        // size, the return trip 28.5 ns per key at 256 slices and 54.7 at 2048 -- the bound is their ratio)
        const double kLazyStay = s->m <= (1ULL << 29) ? 1.6 : (s->m <= (1ULL << 30) ? 2.2 : 3.0);
        const double kLazyEnter = s->m <= (1ULL << 29) ? 0.93 : (s->m <= (1ULL << 30) ? 0.85 : 0.72);
        constexpr uint32_t kLazyBackoff = 32;
        volatile unsigned long long &lazy_seq = s->lk.pin[4], &lazy_wait = s->lk.pin[5];  // (host-only words of the pinned page)
        if (seq != lazy_seq) {
            lazy_seq = seq;
            if (lazy_wait) lazy_wait = lazy_wait - 1;
            if (by == 4 && f > kLazyStay) lazy_wait = kLazyBackoff;
        }
        // (keyed probes against the return trip: at 2048 slices the trip costs twice as much per key, a miss store the same -- m = 2^31, 25 % of
        // the keys absent: keyed 1683 us per 2^25 keys, return trip 1817)
        const bool big = s->m > (1ULL << 30);
        if (by == 4) s->lk.mode = f <= kLazyStay ? 4 : 1;
        else if (by == 1) s->lk.mode = flags_ok ? 3 : (f >= kLazyEnter && !lazy_wait ? 4 : (f < (big ? 0.28 : 0.12) ? 0 : 1));
        else s->lk.mode = flags_ok ? 3 : (f > (big ? 0.30 : 0.22) ? 1 : 0);
    }
    return s->lk.mode;
}

// the tally of this call -> the pinned page, by a one-thread kernel at the end of the call's work (stream-ordered, no host wait)
static int publish_tally(psk_sketch *s, unsigned long long units, int scheme, hipStream_t st)
{
    // (every call: the one-thread launch is hidden behind the call's last kernel -- publishing only every fourth call measured the same
    // step time, 46.0-46.5 G key-ops/s either way, as did folding it into the last workgroup of k_bloom_test in round 2)
    if (g_bloom_lookup != 2 || !s->lk.dev) return PSK_OK;
    hipLaunchKernelGGL(k_lookup_publish, dim3(1), dim3(1), 0, st, s->lk.dev, s->lk.pin, units, (unsigned long long)scheme);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

int PSK_VARIANT(bloom_check_partitioned)(psk_sketch *s, const Batch &b, uint8_t *out_dev, hipStream_t st, bool *done)
{
    *done = false;
    if (!part_wanted(b.n, s->k, 4)) return PSK_OK;
    const int scheme = choose_scheme(s, st);
    if (scheme < 0) return scheme;
    if (scheme == 1) {
        PSK_TRY(bloom_check_return_trip(s, b, out_dev, st, done));
        if (*done) return publish_tally(s, b.n, 1, st);
        // (not eligible -- e.g. a tile too large for 16-bit stage positions: the keyed kernels below take the batch)
    }
    if (scheme == 3) {
        PSK_TRY(bloom_check_tile_flags(s, b, out_dev, st, done));
        if (*done) return PSK_OK;  // (the finishing kernel of the last round has published the tally)
    }
    if (scheme == 4) {
        PSK_TRY(bloom_check_lazy(s, b, out_dev, st, done));
        if (*done) return publish_tally(s, b.n, 4, st);
    }
    *done = false;
    PartGeom g;
    uint64_t round_keys;
    if (!check_geometry(s, b, &g, &round_keys)) return PSK_OK;
    HIP_TRY(hipMemsetAsync(out_dev, 1, b.n, st));  // once for every round (a fill launch is ~5 us whatever it fills)
    for (uint64_t start = 0; start < b.n; start += round_keys) {
        const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
        const Batch sub = sub_batch(b, start, cnt);
        uint8_t *out = out_dev + start;
        bool handled = false;
        PSK_TRY(check_round_scatter(s, sub, cnt, out, nullptr, &g, st, &handled));
        if (!handled) return PSK_OK;
        PSK_TRY(check_round_test(s, g, out, st, s->lk.dev));
    }
    PSK_TRY(publish_tally(s, b.n * (uint64_t)s->k, 0, st));
    *done = true;
    return PSK_OK;
}

// Split lookup, first half: hash + partition the first round now (this never reads the table), e.g. while a
// multi-GPU merge of the table is still in flight on another stream.
int PSK_VARIANT(bloom_check_begin_partitioned)(psk_sketch *s, const Batch &b, hipStream_t st)
{
    s->pend.active = true;
    s->pend.scattered = false;
    s->pend.b = b;
    s->pend.scheme = 0;
    if (!part_wanted(b.n, s->k, 4)) return PSK_OK;
    const int scheme = choose_scheme(s, st);
    if (scheme < 0) return scheme;
    if (scheme == 3 && tile_flag_geometry(s, b, &s->pend.g, &s->pend.round_keys)) {
        // tile flags: a probe of an overflowing segment cannot be tested yet -- it flags its tile, whose keys the finish re-checks
        const uint64_t cnt = b.n < s->pend.round_keys ? b.n : s->pend.round_keys;
        bool handled = false;
        PSK_TRY(tile_flag_scatter(s, sub_batch(b, 0, cnt), cnt, true, &s->pend.g, &s->pend.gen, st, &handled));
        if (handled) {
            s->pend.scattered = true;
            s->pend.scheme = 3;
            return PSK_OK;
        }
    }
    if (!check_geometry(s, b, &s->pend.g, &s->pend.round_keys)) return PSK_OK;
    PSK_TRY(ensure(s->s_flag, 8));
    uint32_t *flag = (uint32_t *)s->s_flag.p;
    HIP_TRY(hipMemsetAsync(flag, 0, 4, st));
    const uint64_t cnt = b.n < s->pend.round_keys ? b.n : s->pend.round_keys;
    bool handled = false;
    PSK_TRY(check_round_scatter(s, sub_batch(b, 0, cnt), cnt, nullptr, flag, &s->pend.g, st, &handled));
    s->pend.scattered = handled;
    return PSK_OK;
}

// second half: pass 2 of the first round, then the remaining rounds in full.  *redo_flag_possible tells the caller
// to follow up with the flag-guarded direct check of the first round (an overflowed segment dropped probes).
int PSK_VARIANT(bloom_check_finish_partitioned)(psk_sketch *s, uint8_t *out_dev, hipStream_t st, bool *redo_flag_possible)
{
    *redo_flag_possible = false;
    const Batch &b = s->pend.b;
    if (!s->pend.scattered) return PSK_OK;
    const uint64_t round_keys = s->pend.round_keys;
    const uint64_t cnt0 = b.n < round_keys ? b.n : round_keys;
    if (s->pend.scheme == 3) {
        PartGeom g = s->pend.g;
        PSK_TRY(tile_flag_test(s, sub_batch(b, 0, cnt0), cnt0, g, s->pend.gen, out_dev, cnt0 == b.n ? b.n * (uint64_t)s->k : 0ULL, st));
        for (uint64_t start = cnt0; start < b.n; start += round_keys) {
            const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
            const Batch sub = sub_batch(b, start, cnt);
            bool handled = false;
            uint32_t gen = 0;
            PSK_TRY(tile_flag_scatter(s, sub, cnt, false, &g, &gen, st, &handled));
            if (!handled) return fail(PSK_EHIP, "split lookup: layout lost its partitioned instantiation");
            PSK_TRY(tile_flag_test(s, sub, cnt, g, gen, out_dev + start, start + cnt == b.n ? b.n * (uint64_t)s->k : 0ULL, st));
        }
        return PSK_OK;
    }
    HIP_TRY(hipMemsetAsync(out_dev, 1, cnt0, st));
    PSK_TRY(check_round_test(s, s->pend.g, out_dev, st, s->lk.dev));
    *redo_flag_possible = true;
    PartGeom g = s->pend.g;
    for (uint64_t start = cnt0; start < b.n; start += round_keys) {
        const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
        const Batch sub = sub_batch(b, start, cnt);
        uint8_t *out = out_dev + start;
        bool handled = false;
        HIP_TRY(hipMemsetAsync(out, 1, cnt, st));
        PSK_TRY(check_round_scatter(s, sub, cnt, out, nullptr, &g, st, &handled));
        if (!handled) return fail(PSK_EHIP, "split lookup: layout lost its partitioned instantiation");
        PSK_TRY(check_round_test(s, g, out, st, s->lk.dev));
    }
    return publish_tally(s, b.n * (uint64_t)s->k, 0, st);
}
