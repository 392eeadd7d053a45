// partitioned CountingBloomFilter decrement launcher (own translation unit: parallel build)
#include "psk_part_counter.hpp"

// The unchecked decrement of every index by the key's weight: what countingbloom.py:186-208 does for a well-formed stream
// (min_val >= num_els, so to_remove == num_els); frozen counters stay, a counter that would go below zero is tallied as a
// contract violation (k_counter_apply's fold).  Used by the write-combined update path.
int PSK_VARIANT(cbf_remove_partitioned)(psk_sketch *s, const Batch &b, const uint32_t *w_dev, hipStream_t st, bool *done)
{
    return counter_add_partitioned<IdxBloom, false, true>(s, b, w_dev, s->m, st, done);
}
