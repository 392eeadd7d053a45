// partitioned Bloom insert launcher (own translation unit: parallel build)
#include "psk_host.hpp"

// Bloom insert through the partitioned path; *done = false when this batch/table is not eligible
int PSK_VARIANT(bloom_add_partitioned)(psk_sketch *s, const Batch &b, hipStream_t st, bool *done)
{
    *done = false;
    if (!part_wanted(b.n, s->k)) return PSK_OK;
    PartGeom g;
    if (!part_slices(s->m, 20, 7, &g, 16384)) return PSK_OK;
    g.k = s->k;
    PartGeom g1;
    uint32_t sub_bits = 0;
    if (two_level_geometry(g, &g1, &sub_bits)) {
        // more slices than one pass can bin well: coarse buckets first (inline 32-bit probes), then k_part_split
        const uint64_t round_keys = part_round_keys_two_level(b.n, s->k);
        for (uint64_t start = 0; start < b.n; start += round_keys) {
            const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
            const Batch sub = sub_batch(b, start, cnt);
            bool handled = false;
            SpillBloomOr spill{(uint32_t *)s->table};
            PSK_TRY(with_part_source(sub, &handled, [&](auto src) {
                using Src = decltype(src);
                return with_kt<Src>(s->k, [&](auto kt) {
                    constexpr int KT = decltype(kt)::value;
                    return launch_scatter<Src, IdxBloomWide<kTuPow2>, PayZero, SpillBloomOr, KT>(s, src, IdxBloomWide<kTuPow2>{s->md}, PayZero{},
                                                                                              spill, &g1, cnt, st);
                });
            }));
            if (!handled) return PSK_OK;
            PartGeom g2 = g;
            PSK_TRY((split_level2<0, SpillBloomOr>(s, g1, &g2, sub_bits, cnt * (uint64_t)s->k, spill, st)));
            const size_t lds = (size_t)1 << (g2.shift - 3);
            PSK_TRY(set_dyn_lds(k_bloom_apply, lds));
            hipLaunchKernelGGL(k_bloom_apply, dim3(g2.nbuckets), dim3(kApplyThreads), lds, st, (uint32_t *)s->table, s->padded_bytes / 4,
                               g2, (const uint32_t *)s->s_cnt2.p, (const uint4 *)s->s_part2.p);
            HIP_TRY(hipGetLastError());
        }
        *done = true;
        return PSK_OK;
    }
    if (g.nbuckets > (uint32_t)kPartMaxBuckets) return PSK_OK;
    const uint64_t round_keys = part_round_keys_big_table(b.n, s->k, PayNone::group, s->padded_bytes);
    for (uint64_t start = 0; start < b.n; start += round_keys) {
        const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
        const Batch sub = sub_batch(b, start, cnt);
        bool handled = false;
        PSK_TRY(with_part_source(sub, &handled, [&](auto src) {
            using Src = decltype(src);
            return with_kt<Src>(s->k, [&](auto kt) {
                constexpr int KT = decltype(kt)::value;
                SpillBloomOr spill{(uint32_t *)s->table};
                return launch_scatter<Src, IdxBloom<kTuPow2>, PayNone, SpillBloomOr, KT>(s, src, IdxBloom<kTuPow2>{s->md}, PayNone{},
                                                                                          spill, &g, cnt, st);
            });
        }));
        if (!handled) return PSK_OK;  // layout without a partitioned instantiation: nothing was launched
        const size_t lds = (size_t)1 << (g.shift - 3);
        PSK_TRY(set_dyn_lds(k_bloom_apply, lds));
        hipLaunchKernelGGL(k_bloom_apply, dim3(g.nbuckets), dim3(kApplyThreads), lds, st, (uint32_t *)s->table,
                           s->padded_bytes / 4, g, (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p);
        HIP_TRY(hipGetLastError());
    }
    *done = true;
    return PSK_OK;
}

// Bloom lookup through the partitioned path: probes carry their key's index; out[] starts at 1 and