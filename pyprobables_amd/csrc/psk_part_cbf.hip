// partitioned CountingBloomFilter add / decrement launchers (ONE translation unit: adds and decrements share every pass-1 instantiation)
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

// The unchecked decrement of every index by the key's weight: what countingbloom.py:186-208 does for a well-formed stream
// (min_val >= num_els, so to_remove == num_els); frozen counters stay, a counter that would go below zero is tallied as a
// contract violation (k_counter_apply's fold).  Used by the write-combined update path.
// opt 1: the transactional form (wrapping subtraction, `flag` raised where the reference's result would depend on the order inside the
// batch); opt 2: its inverse.  The same batch and amounts must be passed to both.
int PSK_VARIANT(cbf_remove_partitioned)(psk_sketch *s, const Batch &b, const uint32_t *w_dev, hipStream_t st, bool *done, int opt, uint32_t *flag)
{
    return counter_add_partitioned<IdxBloom, false, true>(s, b, w_dev, s->m, st, done, opt, flag);
}

// Validated remove of a unit-weight batch, fast path (psk_nibble.hpp, "OPTIMISTIC decrement"): pass 1 of all keys + the decrement that
// checks T[c] >= R[c] while it subtracts; *launched = false: not eligible (nothing enqueued).  The caller reads the flag (s_flag): clear
// = done; up = cbf_remove_fast_undo puts the counters back and the exact path takes the batch.
int PSK_VARIANT(cbf_remove_fast_begin)(psk_sketch *s, const Batch &b, hipStream_t st, bool *launched)
{
    *launched = false;
    const uint64_t cells = s->m;
    if (g_update_nibble == 0 || g_remove_dryrun == 0 || !part_wanted(b.n, s->k, 4)) return PSK_OK;
    if (b.n * (uint64_t)s->k < cells / 8 || b.n > part_round_keys_two_level(b.n, s->k) || !nib_load_ok(b.n, s->k, cells)) return PSK_OK;  // (one round only)
    PartGeom g;
    if (!nib_geometry(cells, true, &g)) return PSK_OK;
    g.k = s->k;
    PSK_TRY(ensure(s->s_flag, 8));
    uint32_t *flag = (uint32_t *)s->s_flag.p;
    HIP_TRY(hipMemsetAsync(flag, 0, 4, st));
    bool handled = false;
    // (the spill of an overflowing segment would decrement with the reference's clamp, which the undo could not invert: such a
    // batch -- hundreds of thousands of copies of one key -- raises the flag instead and the exact path handles it)
    SpillRaiseFlagCounter spill{flag};
    PSK_TRY(with_part_source(b, &handled, [&](auto src) {
        using Src = decltype(src);
        return with_kt<Src>(s->k, [&](auto kt) {
            constexpr int KT = decltype(kt)::value;
            return launch_scatter<Src, IdxBloom<kTuPow2>, PayNone, SpillRaiseFlagCounter, KT>(s, src, IdxBloom<kTuPow2>{s->md}, PayNone{}, spill, &g, b.n, st);
        });
    }));
    if (!handled) return PSK_OK;
    PSK_TRY(nib_apply_mode<3>(s, g, s->s_cnt.p, s->s_part.p, st, flag));
    s->rm_g = g;
    *launched = true;
    return PSK_OK;
}

int PSK_VARIANT(cbf_remove_fast_undo)(psk_sketch *s, hipStream_t st)
{
    return nib_apply_mode<4>(s, s->rm_g, s->s_cnt.p, s->s_part.p, st);
}
