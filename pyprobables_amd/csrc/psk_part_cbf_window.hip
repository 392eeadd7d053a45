// update windows of the CountingBloomFilter (psk_window.hpp): phase-aware pass 1 + the phase-by-phase fold (own translation unit)
#include "psk_part_counter.hpp"
#include "psk_window.hpp"

extern PSK_HIDDEN int64_t g_window_nt;     // psk_capi.hip: option "update_window_nt"
extern PSK_HIDDEN int64_t g_window_image;  // psk_capi.hip: option "update_window_image"
extern PSK_HIDDEN int64_t g_window_wide;   // option "update_window_wide"
extern PSK_HIDDEN int64_t g_window_tile;   // option "update_window_tile": 0 = by the rule in window_scatter, 2048 / 4096 = forced (A/B)
extern PSK_HIDDEN int64_t g_window_shadow, g_window_shadow_writes;  // options "update_window_shadow" / "update_window_shadow_writes" (read-only tally)

// Tables of the window's pass 1 and fold (pinned staging + device copy, one contiguous upload): [0 .. kWinMaxPhases] the fold's phases,
// behind them the pieces of pass 1 (PhaseDesc in psk_partition.hpp).
constexpr size_t kWinMaxPieces = 4096 + 2 * kWinMaxPhases;   // (kWinMaxBatches waiting batches, each cut at most once more per phase end)
constexpr size_t kWinTableEntries = (kWinMaxPhases + 1) + (kWinMaxPieces + 1);

template <int KT, int NT>
static int window_scatter(psk_sketch *s, const WinBatchHost *wb, uint32_t nb, PartGeom *g, uint32_t *flag, hipStream_t st, uint32_t *nph_out)
{
    using Tile = PartTile<PayNonePhased, KT, NT>;
    // keys per tile: the shape's full tile (4096 keys for k <= 8) where it brings at most ~6 probe groups per slice and its stage fits the
    // LDS -- tables of ~900 slices and more, BASELINE cfg 4's 1024 --, half of it otherwise (the fold holds 12 groups per segment and phase)
    uint64_t tk = Tile::TILE;
    {
        const uint32_t kk0 = g->k < (uint32_t)KT ? g->k : (uint32_t)KT;
        const bool roomy = (double)tk * kk0 / (double)g->nbuckets / 6.0 + 0.5 <= 6.0 && scatter_lds_bytes<PayNonePhased, KT, NT>(g, tk) <= kScatterLdsBudget;
        if (!roomy && tk >= 2048 && g_window_tile != 4096) tk /= 2;
        if (g_window_tile == 2048 && tk > 2048) tk = 2048;  // option "update_window_tile" (A/B)
    }
    if (!s->win.pin) HIP_TRY(hipHostMalloc(&s->win.pin, kWinTableEntries * sizeof(PhaseDesc), hipHostMallocDefault));
    PhaseDesc *fold = (PhaseDesc *)s->win.pin;            // [nph + 1]
    PhaseDesc *piece = fold + (kWinMaxPhases + 1);         // [npc + 1]
    const uint32_t nwg = 256;  // one 1024-thread workgroup per CU (k <= 8), every one of them writes its snapshots
    // A phase of the fold is at most two tiles per pass-1 workgroup (longer runs of same-type batches are cut: k_win_fold holds a
    // phase's probe groups of a segment in a fixed number of registers).
    // (tables of few slices bring long runs per tile -- 2048 keys x k / B probes: 6.5 groups at 366 slices -- so there one tile per workgroup
    // and phase: with two, nearly every slice of a 9.6e7-counter table overflowed the fold's 12 groups per segment and phase and took the atomics)
    const double groups_per_tile = (double)tk * (g->k < (uint32_t)KT ? g->k : (uint32_t)KT) / (double)g->nbuckets / 6.0 + 0.5;
    const uint64_t max_tiles = (groups_per_tile > 4.0 ? 1ULL : 2ULL) * nwg;  // tiles per phase
    // Every piece -- a stretch of keys contiguous in memory -- starts on a tile boundary and its last tile is short; a batch that follows its
    // predecessor in memory (copies in the window's list) continues that piece when the piece ends on a whole tile.
    uint64_t tiles = 0, phase_tiles = 0;
    uint32_t nph = 0, npc = 0;
    *nph_out = 0;
    for (uint32_t bi = 0; bi < nb; ++bi) {
        if (((uintptr_t)wb[bi].keys & 15) != 0) return fail(PSK_EINVAL, "update window: key batches must be 16-byte aligned");
        for (uint64_t off = 0; off < wb[bi].n;) {
            if (nph == 0 || (fold[nph - 1].remove != wb[bi].remove) || phase_tiles == max_tiles) {
                if (nph >= (uint32_t)kWinMaxPhases) return PSK_OK;  // (the caller replays such a window batch by batch)
                if (npc) piece[npc - 1].remove |= kPieceEndsPhase;
                fold[nph++] = PhaseDesc{(uint32_t)tiles, wb[bi].remove, 0LL, 0ULL};
                phase_tiles = 0;
            }
            const uint64_t room = (max_tiles - phase_tiles) * tk;
            const uint64_t cnt = wb[bi].n - off < room ? wb[bi].n - off : room;
            const uint64_t first = (uint64_t)((uintptr_t)wb[bi].keys >> 4) + off;  // in 16-byte units from address 0
            PhaseDesc *pv = npc ? &piece[npc - 1] : nullptr;
            const bool joins = pv && !(pv->remove & kPieceEndsPhase) && (pv->remove >> 8) == nph - 1 && pv->nkeys % tk == 0 &&
                               (uint64_t)(pv->key_off + (long long)((uint64_t)pv->tile0 * tk)) + pv->nkeys == first;
            if (joins) {
                pv->nkeys += cnt;
            } else {
                if (npc >= kWinMaxPieces) return PSK_OK;
                piece[npc++] = PhaseDesc{(uint32_t)tiles, (wb[bi].remove & 1u) | ((nph - 1) << 8), (long long)first - (long long)(tiles * tk), cnt};
            }
            const uint64_t t = (cnt + tk - 1) / tk;
            tiles += t;
            phase_tiles += t;
            off += cnt;
        }
    }
    if (npc == 0) return PSK_OK;
    piece[npc - 1].remove |= kPieceEndsPhase;
    if (tiles >= (1ULL << 31)) return fail(PSK_EINVAL, "update window of %llu tiles", (unsigned long long)tiles);
    fold[nph] = PhaseDesc{(uint32_t)tiles, 0u, 0LL, 0ULL};
    piece[npc] = PhaseDesc{(uint32_t)tiles, 0u, 0LL, 0ULL};
    PSK_TRY(ensure(s->s_phase, kWinTableEntries * sizeof(PhaseDesc)));
    // (pinned: consumed before the flush's sync; the unused tail of the fold's table travels along -- one copy)
    HIP_TRY(hipMemcpyAsync(s->s_phase.p, fold, ((kWinMaxPhases + 1) + npc + 1) * sizeof(PhaseDesc), hipMemcpyHostToDevice, st));
    const uint64_t tiles_per_wg = (tiles + nwg - 1) / nwg;
    const uint32_t kk = g->k < (uint32_t)KT ? g->k : (uint32_t)KT;
    const double mean = (double)tiles_per_wg * (double)tk * kk / (double)g->nbuckets;
    const uint64_t segcap = (uint64_t)(mean / Tile::GS + 0.5 * (double)tiles_per_wg + 8.0 * __builtin_sqrt(mean) / Tile::GS + 16.0);
    if (segcap >= (1u << 24) || (uint64_t)g->nbuckets * segcap >= (1ULL << 32)) return fail(PSK_EINVAL, "update window too large (segments of %llu groups)", (unsigned long long)segcap);
    g->nwg = nwg;
    g->segcap = (uint32_t)segcap;
    g->tile = (uint32_t)tk;
    g->dense = 0;
    g->append = 0;
    PSK_TRY(ensure(s->s_part, (uint64_t)g->nbuckets * nwg * segcap * 16 + 256));
    PSK_TRY(ensure(s->s_cnt, (uint64_t)g->nbuckets * nwg * 4 + 128));
    PSK_TRY(ensure(s->s_snap, (uint64_t)nph * g->nbuckets * nwg * 4));
    const PhaseDesc *piece_dev = (const PhaseDesc *)s->s_phase.p + (kWinMaxPhases + 1);
    const PayNonePhased pay{piece_dev, npc, (uint32_t *)s->s_snap.p, (uint64_t)(piece[0].key_off + (long long)((uint64_t)piece[0].tile0 * tk))};
    // (an overflowing segment would need the reference's clamp, which the undo could not invert: it raises the flag instead)
    const SpillRaiseFlagCounter spill{flag};
    const size_t lds = scatter_lds_bytes<PayNonePhased, KT, NT>(g, tk);
    auto kern = k_part_scatter<KeysFixed16, IdxBloom<kTuPow2>, PayNonePhased, SpillRaiseFlagCounter, KT, NT>;
    PSK_TRY(set_dyn_lds(kern, lds));
    // (the keys are addressed from 0 in 16-byte units: a piece's key_off is where its memory is)
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT), lds, st, KeysFixed16{(const uint4 *)nullptr}, IdxBloom<kTuPow2>{s->md}, pay, spill, *g, tiles * tk,
                       (uint32_t *)s->s_cnt.p, (uint4 *)s->s_part.p);
    HIP_TRY(hipGetLastError());
    *nph_out = nph;
    return PSK_OK;
}

int PSK_VARIANT(cbf_window_fold)(psk_sketch *s, const WinBatchHost *wb, uint32_t nb, hipStream_t st, bool *launched, bool *ok)
{
    *launched = false;
    *ok = false;
    PartGeom g;
    if (g_update_nibble == 0 || nb == 0 || s->k > 32 || !nib_geometry(s->m, true, &g)) return PSK_OK;
    g.k = s->k;
    PSK_TRY(ensure(s->s_flag, 8));
    uint32_t *flag = (uint32_t *)s->s_flag.p;  // [0] a remove met a zero (undo + replay), [1] a slice took the atomics (its 4-bit image is void)
    HIP_TRY(hipMemsetAsync(flag, 0, 8, st));
    uint32_t nph_dev = 0;  // phases of the fold (0: the window does not fit -- nothing was launched)
    PSK_TRY(with_kt<KeysFixed16>(s->k, [&](auto kt) {
        constexpr int KT = decltype(kt)::value;
        if constexpr (KT <= 8) {
            if (scatter_lds_bytes<PayNonePhased, KT, 1024>(&g, PartTile<PayNonePhased, KT, 1024>::TILE / 2) <= kScatterLdsBudget) return window_scatter<KT, 1024>(s, wb, nb, &g, flag, st, &nph_dev);
        }
        if (scatter_lds_bytes<PayNonePhased, KT, kPartThreads>(&g) <= kScatterLdsBudget) return window_scatter<KT, kPartThreads>(s, wb, nb, &g, flag, st, &nph_dev);
        return (int)PSK_OK;
    }));
    if (nph_dev == 0) return PSK_OK;
    *launched = true;
    const WinPhases wp{nph_dev, (const PhaseDesc *)s->s_phase.p};
    const bool nib = g_window_image != 8;  // option "update_window_image": 4 (default) = nibble images, one workgroup per slice; 8 = byte images, two
    const uint32_t pshift = win_part_shift(g.shift, nib);
    const uint32_t parts = g.nbuckets << (g.shift - pshift);
    // tables of few slices (a 2048-key tile brings more than 4 probe groups per slice: below ~600 slices): the WIDE fold -- five groups per lane
    // and phase, byte-wide group counts -- when its count table still fits the LDS next to the image
    const double groups_per_tile = 2048.0 * (double)(s->k < 8 ? s->k : 8) / (double)g.nbuckets / 6.0 + 0.5;
    // (per segment and phase the narrow fold holds 12 groups: one tile of more than 4 groups per slice, or two tiles of more than 3.25 -- 732 slices:
    // 2 x 3.8, most slices overflowed, 9.4 -> 15.9 G ops/s with the wide fold; 1024 slices, BASELINE cfg 4: 2 x 2.8 fit, and the wide fold costs 5 %)
    const bool wide = nib && g_window_wide != 0 && (groups_per_tile > 3.25 || g_window_wide == 2) && win_fold_lds(g, nph_dev, nib, true) <= 160 * 1024;  // (2: wherever it fits, A/B)
    const size_t lds = win_fold_lds(g, nph_dev, nib, wide);
    if (lds > 160 * 1024) return fail(PSK_EINVAL, "update window: %u phases of %u segments do not fit the fold's LDS", nph_dev, g.nwg);
    PSK_TRY(ensure(s->s_wstat, (uint64_t)parts * 4));
    // Kept 4-bit images (psk_sketch::shadow): when the lookups already keep them, the fold -- which ends with the very image of every slice
    // in LDS -- leaves them up to date instead of stale: the lookup behind a flush loads 128 MiB instead of reading the 1 GiB table again.
    const uint64_t shadow_words = (uint64_t)g.nbuckets << (g.shift - 3);
    uint32_t *shadow_out = nullptr;
    if (nib && pshift == g.shift && g_cbf_shadow != 0 && g_window_shadow != 0 && !s->shadow.exposed && s->shadow.img.p && s->shadow.words == shadow_words &&
        s->shadow.img.cap >= shadow_words * 4)
        shadow_out = (uint32_t *)s->shadow.img.p;
    if (shadow_out) s->shadow.built = ~0ULL;  // (being overwritten: valid again only once the verdict is in)
    {
        auto kern = wide ? k_win_fold<false, true, 5> : (nib ? k_win_fold<false, true> : k_win_fold<false, false>);
        PSK_TRY(set_dyn_lds(kern, lds));
        hipLaunchKernelGGL(kern, dim3(parts), dim3(kApplyThreads), lds, st, (uint32_t *)s->table, s->m, g, (const uint4 *)s->s_part.p, (const uint32_t *)s->s_snap.p,
                           wp, (uint32_t *)s->s_wstat.p, flag, (uint32_t)(g_window_nt != 0), shadow_out);
        HIP_TRY(hipGetLastError());
    }
    if (g_window_force_fail) HIP_TRY(hipMemsetAsync(flag, 1, 4, st));  // (tests: the undo + replay path on a well-formed stream)
    uint32_t verdict[2] = {1, 1};
    HIP_TRY(hipMemcpyAsync(verdict, flag, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (verdict[0] == 0) {
        *ok = true;
        if (shadow_out) {  // every slice wrote its image (from LDS, or from the table behind its atomics): the kept images mirror the table as it is now
            s->shadow.built = s->table_version;
            s->shadow.stream = st;
            ++g_window_shadow_writes;
        }
        return PSK_OK;
    }
    auto kern = nib ? k_win_fold<true, true> : k_win_fold<true, false>;  // the proof failed: put every part back where it was
    PSK_TRY(set_dyn_lds(kern, lds));
    hipLaunchKernelGGL(kern, dim3(parts), dim3(kApplyThreads), lds, st, (uint32_t *)s->table, s->m, g, (const uint4 *)s->s_part.p, (const uint32_t *)s->s_snap.p, wp,
                       (uint32_t *)s->s_wstat.p, flag, (uint32_t)(g_window_nt != 0), (uint32_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}
