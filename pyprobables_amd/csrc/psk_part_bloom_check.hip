// partitioned Bloom lookup launcher (own translation unit: parallel build)
#include "psk_host.hpp"

// any probe that finds its bit clear stores a 0
int bloom_check_partitioned(psk_sketch *s, const Batch &b, uint8_t *out_dev, hipStream_t st, bool *done)
{
    *done = false;
    if (!part_wanted(b.n, s->k, 4)) return PSK_OK;
    PartGeom g;
    if (!part_slices(s->m, 20, 7, &g)) return PSK_OK;
    g.k = s->k;
    uint64_t round_keys = part_round_keys(b.n, s->k, PayKeyId::group);
    if (round_keys > 0xFFFFFFFFULL) round_keys = 0xFFFFFFFFULL;  // 32-bit key ids inside a round
    for (uint64_t start = 0; start < b.n; start += round_keys) {
        const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
        const Batch sub = sub_batch(b, start, cnt);
        uint8_t *out = out_dev + start;
        bool handled = false;
        PSK_TRY(with_part_source(sub, &handled, [&](auto src) {
            using Src = decltype(src);
            return with_kt<Src>(s->k, [&](auto kt) {
                constexpr int KT = decltype(kt)::value;
                HIP_TRY(hipMemsetAsync(out, 1, cnt, st));
                SpillBloomTest spill{(const uint32_t *)s->table, out};
                if (s->pow2)
                    return launch_scatter<Src, IdxBloom<true>, PayKeyId, SpillBloomTest, KT>(s, src, IdxBloom<true>{s->md},
                                                                                             PayKeyId{}, spill, &g, cnt, st);
                return launch_scatter<Src, IdxBloom<false>, PayKeyId, SpillBloomTest, KT>(s, src, IdxBloom<false>{s->md},
                                                                                          PayKeyId{}, spill, &g, cnt, st);
            });
        }));
        if (!handled) return PSK_OK;
        const size_t lds = (size_t)1 << (g.shift - 3);
        PSK_TRY(set_dyn_lds(k_bloom_test, lds));
        hipLaunchKernelGGL(k_bloom_test, dim3(g.nbuckets), dim3(kApplyThreads), lds, st, (const uint32_t *)s->table,
                           s->padded_bytes / 4, g, (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p, out);
        HIP_TRY(hipGetLastError());
    }
    *done = true;
    return PSK_OK;
}

// Counter add (CMS add / remove, CBF add) through the partitioned path.  IDX = IdxCms / IdxBloom;
// w_dev = per-key weights (uint32 bit patterns) or nullptr for unit weights.  The caller has already run