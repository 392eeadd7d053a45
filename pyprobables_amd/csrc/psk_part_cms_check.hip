// partitioned CountMinSketch lookup launcher (own translation unit: parallel build)
#include "psk_part_lookup.hpp"

template <class Op>
static int redo_direct(const Batch &b, const Op &op, const uint32_t *flag, hipStream_t st)
{
    bool handled = false;
    return with_part_source(b, &handled, [&](auto src) {
        using Src = decltype(src);
        hipLaunchKernelGGL((k_apply_if<Src, Op>), dim3(grid_for_keys(b.n)), dim3(kBlock), 0, st, flag, src, op, b.n);
        HIP_TRY(hipGetLastError());
        return (int)PSK_OK;
    });
}

// countminsketch.py:332-340 check_alt under the min / mean (int32 out) or mean-min (int64 out) query
int PSK_VARIANT(cms_check_partitioned)(psk_sketch *s, const Batch &b, int query, int64_t els_added, void *out_dev, hipStream_t st, bool *done)
{
    const uint64_t cells = s->m * (uint64_t)s->k;
    if (query == PSK_Q_MEANMIN) {
        auto redo = [&](const uint32_t *flag, hipStream_t st2) {
            return redo_direct(b, CmsCheckMeanMin<kTuPow2>{(const int32_t *)s->table, s->md, s->k, els_added, (int64_t *)out_dev}, flag, st2);
        };
        return counter_check_partitioned<IdxCms>(s, b, s->k, cells, QueryCmsMeanMin{els_added, (int64_t)s->m}, (int64_t *)out_dev, st, done, redo);
    }
    const bool mean = query == PSK_Q_MEAN;
    auto redo = [&](const uint32_t *flag, hipStream_t st2) {
        return redo_direct(b, CmsCheck<kTuPow2>{(const int32_t *)s->table, s->md, s->k, (int32_t *)out_dev, mean}, flag, st2);
    };
    if (mean) return counter_check_partitioned<IdxCms>(s, b, s->k, cells, QueryCmsMean{}, (int32_t *)out_dev, st, done, redo);
    return counter_check_partitioned<IdxCms>(s, b, s->k, cells, QueryCmsMin{}, (int32_t *)out_dev, st, done, redo);
}
