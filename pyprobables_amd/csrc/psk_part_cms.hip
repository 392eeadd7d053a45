// partitioned CountMinSketch add / remove launchers (ONE translation unit: both share every pass-1 instantiation -- the sign of an update is a
// run-time field of the spill functor and a template argument of the small pass-2 kernel only)
#include "psk_part_counter.hpp"

int PSK_VARIANT(cms_add_partitioned)(psk_sketch *s, const Batch &b, const uint32_t *w_dev, hipStream_t st, bool *done)
{
    return counter_add_partitioned<IdxCms, true, false>(s, b, w_dev, s->m * (uint64_t)s->k, st, done);
}

int PSK_VARIANT(cms_remove_partitioned)(psk_sketch *s, const Batch &b, const uint32_t *w_dev, hipStream_t st, bool *done)
{
    return counter_add_partitioned<IdxCms, true, true>(s, b, w_dev, s->m * (uint64_t)s->k, st, done);
}
