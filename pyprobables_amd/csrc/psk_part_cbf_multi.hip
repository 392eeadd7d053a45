// borrowed-batch flush of the write-combined CountingBloomFilter updates (own translation unit: parallel build)
#include "psk_part_counter.hpp"

// All borrowed batches of one list as ONE pass 1 (KeysFixed16Multi: the keys are hashed where they lie -- no copy into a list, one
// read) + one fold over the table (psk_nibble.hpp).  countingbloom.py:135-155 (adds) / :203-206 with to_remove == num_els (decrements).
int PSK_VARIANT(cbf_unit_multi_partitioned)(psk_sketch *s, const void *const *base_dev, const uint64_t *start_dev, uint32_t nb, uint64_t n, int neg,
                                            hipStream_t st, bool *done)
{
    *done = false;
    const uint64_t cells = s->m;
    if (g_update_nibble == 0 || n == 0 || n > part_round_keys_two_level(n, s->k) || !nib_load_ok(n, s->k, cells)) return PSK_OK;
    PartGeom g;
    if (!nib_geometry(cells, true, &g)) return PSK_OK;
    g.k = s->k;
    SpillCounter<false> spill{(uint32_t *)s->table, true, neg != 0, (unsigned long long *)(s->ctr + PSK_CTR_SATURATED)};
    const KeysFixed16Multi src{(const uint4 *const *)base_dev, start_dev, nb, (uint64_t)((((unsigned __int128)nb) << 64) / n)};
    PSK_TRY(with_kt<KeysFixed16Multi>(s->k, [&](auto kt) {
        constexpr int KT = decltype(kt)::value;
        return launch_scatter<KeysFixed16Multi, IdxBloom<kTuPow2>, PayNone, SpillCounter<false>, KT>(s, src, IdxBloom<kTuPow2>{s->md}, PayNone{}, spill, &g, n, st);
    }));
    // few probes: drain with atomics instead of a pass over the table (k_nib_apply's direct mode)
    const uint32_t lgp = nib_update_lgparts(g);
    const uint32_t direct = (n * (uint64_t)s->k < cells / 8 ? 1u : 0u) | (lgp << 8);
    const size_t lds = (size_t)1 << (g.shift - 1 - lgp);
    auto launch = [&](auto kern) {
        PSK_TRY(set_dyn_lds(kern, lds));
        hipLaunchKernelGGL(kern, dim3(g.nbuckets << lgp), dim3(kApplyThreads), lds, st, (uint32_t *)s->table, s->m, g, (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p,
                           (const uint32_t *)nullptr, (const uint4 *)nullptr, (unsigned long long *)(s->ctr + PSK_CTR_SATURATED), direct, (uint32_t *)nullptr, g);
        HIP_TRY(hipGetLastError());
        return (int)PSK_OK;
    };
    if (g_nib_update_layout) PSK_TRY(neg ? launch(k_nib_apply<1, true>) : launch(k_nib_apply<0, true>));
    else PSK_TRY(neg ? launch(k_nib_apply<1, false>) : launch(k_nib_apply<0, false>));
    *done = true;
    return PSK_OK;
}
