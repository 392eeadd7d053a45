// partitioned CountingBloomFilter add launcher (own translation unit: parallel build)
#include "psk_part_counter.hpp"

int PSK_VARIANT(cbf_add_partitioned)(psk_sketch *s, const Batch &b, const uint32_t *w_dev, hipStream_t st, bool *done)
{
    return counter_add_partitioned<IdxBloom, false, false>(s, b, w_dev, s->m, st, done);
}
