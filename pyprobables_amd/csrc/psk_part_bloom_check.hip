// partitioned Bloom lookup launcher (own translation unit: parallel build)
#include "psk_host.hpp"

// pass 1 of one round: keyed probes of keys [0, cnt) of `sub` into the bucket buffer.  defer != nullptr: split lookup
// (the table is not consulted; an overflowing segment raises *defer instead of testing its probes directly)
static int check_round_scatter(psk_sketch *s, const Batch &sub, uint64_t cnt, uint8_t *out, uint32_t *defer, PartGeom *g,
                               hipStream_t st, bool *handled)
{
    return with_part_source(sub, handled, [&](auto src) {
        using Src = decltype(src);
        return with_kt<Src>(s->k, [&](auto kt) {
            constexpr int KT = decltype(kt)::value;
            SpillBloomTest spill{(const uint32_t *)s->table, out, defer};
            return launch_scatter<Src, IdxBloom<kTuPow2>, PayKeyId, SpillBloomTest, KT>(s, src, IdxBloom<kTuPow2>{s->md}, PayKeyId{},
                                                                                         spill, g, cnt, st);
        });
    });
}

// pass 2 of one round: any probe that finds its bit clear stores a 0 (out[] pre-set to 1 by the caller)
static int check_round_test(psk_sketch *s, const PartGeom &g, uint8_t *out, hipStream_t st)
{
    const size_t lds = (size_t)1 << (g.shift - 3);
    PSK_TRY(set_dyn_lds(k_bloom_test, lds));
    hipLaunchKernelGGL(k_bloom_test, dim3(g.nbuckets), dim3(kApplyThreads), lds, st, (const uint32_t *)s->table, s->padded_bytes / 4, g,
                       (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p, out);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

static bool check_geometry(psk_sketch *s, uint64_t n, PartGeom *g, uint64_t *round_keys)
{
    if (!part_wanted(n, s->k, 4)) return false;
    if (!part_slices(s->m, 20, 7, g)) return false;
    g->k = s->k;
    uint64_t rk = part_round_keys_big_table(n, s->k, PayKeyId::group, s->padded_bytes);
    // a keyed group spells the tile's ordinal inside its workgroup in 4 bits: at most 16 tiles per workgroup and round
    // (256 workgroups x 16 x 2048-key tiles for k <= 8; 512-key tiles beyond)
    const uint64_t cap = (uint64_t)PayKeyId::max_tiles_per_wg * 256 * (s->k <= 8 ? 2048 : 512);
    if (rk > cap) rk = cap;
    *round_keys = rk;
    return true;
}

int PSK_VARIANT(bloom_check_partitioned)(psk_sketch *s, const Batch &b, uint8_t *out_dev, hipStream_t st, bool *done)
{
    *done = false;
    PartGeom g;
    uint64_t round_keys;
    if (!check_geometry(s, b.n, &g, &round_keys)) return PSK_OK;
    for (uint64_t start = 0; start < b.n; start += round_keys) {
        const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
        const Batch sub = sub_batch(b, start, cnt);
        uint8_t *out = out_dev + start;
        bool handled = false;
        HIP_TRY(hipMemsetAsync(out, 1, cnt, st));
        PSK_TRY(check_round_scatter(s, sub, cnt, out, nullptr, &g, st, &handled));
        if (!handled) return PSK_OK;
        PSK_TRY(check_round_test(s, g, out, st));
    }
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
    if (!check_geometry(s, b.n, &s->pend.g, &s->pend.round_keys)) return PSK_OK;
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
    HIP_TRY(hipMemsetAsync(out_dev, 1, cnt0, st));
    PSK_TRY(check_round_test(s, s->pend.g, out_dev, st));
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
        PSK_TRY(check_round_test(s, g, out, st));
    }
    return PSK_OK;
}
