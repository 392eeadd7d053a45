// partitioned CountingBloomFilter add launcher (own translation unit: parallel build)
#include "psk_part_counter.hpp"

int PSK_VARIANT(cbf_add_partitioned)(psk_sketch *s, const Batch &b, const uint32_t *w_dev, hipStream_t st, bool *done)
{
    return counter_add_partitioned<IdxBloom, false, false>(s, b, w_dev, s->m, st, done);
}

// Write-combined unit-weight updates as scattered probes (psk_sketch::scat): pass 1 of this batch, appended to the add / decrement
// list.  The caller (psk_capi.hip) has set up the geometry and the buffers and flushes a full list first.
int PSK_VARIANT(cbf_scat_append)(psk_sketch *s, const Batch &b, int neg, hipStream_t st, bool *done)
{
    *done = false;
    psk_sketch::ScatList &l = neg ? s->scat.rem : s->scat.add;
    const ScatterTarget target{(uint32_t *)l.cnt.p, (uint4 *)l.part.p};
    PartGeom g = s->scat.g;
    bool handled = false;
    PSK_TRY(nib_scatter<false>(s, b, nullptr, neg != 0, &g, st, &handled, &target));
    *done = handled;
    return PSK_OK;
}

// pass 1 alone (fused flush of the write-combined lists: psk_capi.hip flush_combined)
int PSK_VARIANT(cbf_nib_scatter)(psk_sketch *s, const Batch &b, int neg, int second, PartGeom *g_out, hipStream_t st, bool *done)
{
    return cbf_nib_scatter_only(s, b, neg != 0, second != 0, g_out, st, done);
}
