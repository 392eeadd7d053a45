// partitioned CountingBloomFilter lookup launcher (own translation unit: parallel build)
#include "psk_part_lookup.hpp"

int PSK_VARIANT(bloomidx_lookup_scatter)(psk_sketch *s, const Batch &sub, uint64_t cnt, uint32_t kk, PartGeom *g, uint32_t *flag, hipStream_t st, bool *handled, bool *fits)
{
    *fits = true;
    return with_part_source(sub, handled, [&](auto src) {
        using Src = decltype(src);
        return with_kt<Src>(kk, [&](auto kt) {
            constexpr int KT = decltype(kt)::value;
            using TileSmall = PartTile<PayBloomLookup, KT, kPartThreads>;
            using TileBig = PartTile<PayBloomLookup, KT, 1024>;
            const uint32_t kq = g->k < (uint32_t)KT ? g->k : (uint32_t)KT;
            // 16-bit stage positions (perm[]): the largest tile pass 1 may choose must fit -- else not eligible, nothing launched
            const size_t tile_max = TileBig::TILE > TileSmall::TILE ? TileBig::TILE : TileSmall::TILE;
            if (tile_max * kq + (size_t)5 * g->nbuckets + 3 > 0xFFFFu) { *fits = false; return (int)PSK_OK; }
            const uint64_t max_tiles = (cnt + TileSmall::TILE - 1) / TileSmall::TILE + 1024;  // (+ workgroups: evened tiles, launch_scatter_nt)
            PSK_TRY(ensure(s->s_perm, cnt * (uint64_t)PermRec<KT>::PD * 4 + 16));
            PSK_TRY(ensure(s->s_run, max_tiles * g->nbuckets * 8));
            PayBloomLookup pay{(uint32_t *)s->s_perm.p, (uint2 *)s->s_run.p};
            SpillRaiseFlag spill{flag};
            return launch_scatter<Src, IdxBloom<kTuPow2>, PayBloomLookup, SpillRaiseFlag, KT>(s, src, IdxBloom<kTuPow2>{s->md}, pay, spill, g, cnt, st);
        });
    });
}

// countingbloom.py:166-174 check_alt: min over the kk supplied hashes (kk = k for key layouts)
int PSK_VARIANT(cbf_check_partitioned)(psk_sketch *s, const Batch &b, uint32_t kk, uint32_t *out_dev, hipStream_t st, bool *done)
{
    auto redo = [&](const uint32_t *flag, hipStream_t st2) {
        bool handled = false;
        return with_part_source(b, &handled, [&](auto src) {
            using Src = decltype(src);
            using Op = CbfCheck<kTuPow2>;
            hipLaunchKernelGGL((k_apply_if<Src, Op>), dim3(grid_for_keys(b.n)), dim3(kBlock), 0, st2, flag, src,
                               Op{(const uint32_t *)s->table, s->md, kk, out_dev}, b.n);
            HIP_TRY(hipGetLastError());
            return (int)PSK_OK;
        });
    };
    auto recheck = [&](const uint32_t *amb, hipStream_t st2) {
        bool handled = false;
        return with_part_source(b, &handled, [&](auto src) {
            using Src = decltype(src);
            hipLaunchKernelGGL((k_cbf_recheck15<Src, kTuPow2>), dim3(grid_for_keys(b.n)), dim3(kBlock), 0, st2, amb, src, (const uint32_t *)s->table, s->md, kk, b.n, out_dev);
            HIP_TRY(hipGetLastError());
            return (int)PSK_OK;
        });
    };
    PSK_TRY(cbf_check_nibble(s, b, kk, out_dev, st, done, redo, recheck));  // big tables: 4-bit slice images (psk_nibble.hpp)
    if (*done) return PSK_OK;
    return counter_check_partitioned<IdxBloom>(s, b, kk, s->m, QueryCbfMin{}, out_dev, st, done, redo);
}
