// psk_part_counter.hpp -- launcher template of the partitioned counter adds (CMS add / remove, CBF add)
#pragma once
#include "psk_host.hpp"
#include "psk_nibble.hpp"
#include "psk_nibble_pipe.hpp"

#include <utility>

// account_weights(), so ctr[6] holds this batch's sum|w| for the wrap check inside pass 2.
// the weights of keys [start, ...) + the fused accounting request posted by the caller (psk_sketch::acct), if any
static inline int pay_weights(psk_sketch *s, const uint32_t *w_dev, uint64_t start, PayWeight *pay)
{
    *pay = PayWeight{w_dev ? w_dev + start : nullptr};
    if (w_dev && s->acct.pending) {
        PSK_TRY(ensure(s->s_tally, 1024 * sizeof(ulonglong4)));  // one slot per pass-1 workgroup (<= 512)
        pay->tally = (ulonglong4 *)s->s_tally.p;
        pay->weights_signed = s->acct.weights_signed ? 1 : 0;
    }
    return PSK_OK;
}

// after a weighted pass 1 of `nwg` workgroups: its slots -> the device counters
static inline int fold_tally(psk_sketch *s, const PayWeight &pay, uint32_t nwg, hipStream_t st)
{
    if (!pay.tally) return PSK_OK;
    if (!s->wt.pin) {  // the page the next batch's choice of probe format reads (psk_sketch::wt)
        void *pin = nullptr;
        HIP_TRY(hipHostMalloc(&pin, 32, hipHostMallocDefault));
        s->wt.pin = (volatile unsigned long long *)pin;
        s->wt.pin[0] = s->wt.pin[1] = 0;
    }
    hipLaunchKernelGGL(k_tally_fold, dim3(1), dim3(256), 0, st, (const ulonglong4 *)pay.tally, nwg, s->ctr, s->acct.which, s->acct.bound_mult,
                       s->acct.grow_bound ? 1 : 0, s->wt.pin, ++s->wt.issued);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

// the same accounting folded into pass 2 (k_counter_apply's TallyArgs) instead of the k_tally_fold launch between the passes
static inline int tally_args(psk_sketch *s, const PayWeight &pay, uint32_t nwg, TallyArgs *ta)
{
    *ta = TallyArgs{};
    if (!pay.tally) return PSK_OK;
    if (!s->wt.pin) {
        void *pin = nullptr;
        HIP_TRY(hipHostMalloc(&pin, 32, hipHostMallocDefault));
        s->wt.pin = (volatile unsigned long long *)pin;
        s->wt.pin[0] = s->wt.pin[1] = 0;
    }
    *ta = TallyArgs{(const ulonglong4 *)pay.tally, nwg, s->acct.which, s->acct.bound_mult, s->acct.grow_bound ? 1 : 0, s->wt.pin, ++s->wt.issued};
    return PSK_OK;
}

extern PSK_HIDDEN int64_t g_small_weights_used;

// the compact probe format for this weighted batch?  (exact either way: a weight outside 0 .. 15 goes to the table directly -- at the atomics'
// rate, hence the hint: the count of such weights pass 1 of the previous batches saw)
static inline bool small_weights_wanted(psk_sketch *s)
{
    if (g_small_weights != 1) return g_small_weights == 2;
    if (!s->wt.pin) return false;
    const unsigned long long seq = s->wt.pin[1], big = s->wt.pin[0], seq2 = s->wt.pin[1];  // (the device writes the count, then the number)
    if (seq == 0) return false;  // nothing published yet
    if (seq == seq2 && seq != s->wt.seen) {
        s->wt.seen = seq;
        if (big) s->wt.backoff = 64;
        else if (s->wt.backoff) --s->wt.backoff;
    }
    return s->wt.backoff == 0;
}

// One round of pass 1 for the NIBBLE update path (psk_nibble.hpp): 6 x 20-bit probe groups of the keys of `sub` (all of them, or -- mask
// != nullptr, decrements only -- those with mask[i] != 0) into the handle's bucket buffer, or (fixed) appended to persistent segments.
template <bool MASKABLE>
static inline int nib_scatter(psk_sketch *s, const Batch &sub, const uint32_t *mask, bool neg, PartGeom *g, hipStream_t st, bool *handled,
                              const ScatterTarget *fixed = nullptr, int opt = 0, uint32_t *flag = nullptr)
{
    SpillCounter<false> spill{(uint32_t *)s->table, true, neg, (unsigned long long *)(s->ctr + PSK_CTR_SATURATED), flag, opt};
    return with_part_source(sub, handled, [&](auto src) {
        using Src = decltype(src);
        return with_kt<Src>(s->k, [&](auto kt) {
            constexpr int KT = decltype(kt)::value;
            if constexpr (MASKABLE) {  // (masked batches only exist for decrements: instantiated in that translation unit only)
                if (mask) return launch_scatter<Src, IdxBloom<kTuPow2>, PayUnitMasked, SpillCounter<false>, KT>(s, src, IdxBloom<kTuPow2>{s->md}, PayUnitMasked{mask}, spill, g, sub.n, st, 0, fixed);
            }
            return launch_scatter<Src, IdxBloom<kTuPow2>, PayNone, SpillCounter<false>, KT>(s, src, IdxBloom<kTuPow2>{s->md}, PayNone{}, spill, g, sub.n, st, 0, fixed);
        });
    });
}

extern PSK_HIDDEN int64_t g_nib_update_pipe;  // psk_capi.hip: option "nibble_update_pipe"
// MODE: k_nib_apply's (0 adds, 1 decrements, 3 optimistic decrement with `flag`, 4 its inverse)
template <int MODE>
static inline int nib_apply_mode(psk_sketch *s, const PartGeom &g, const void *cnt, const void *part, hipStream_t st, uint32_t *flag = nullptr)
{
    const uint32_t lgp = nib_update_lgparts(g);
    // round 4: the pipelined pass (psk_nibble_pipe.hpp) -- persistent workgroups, the fold of one slice under the probe groups of the next;
    // the blocks layout of the delta image, one workgroup per slice; option "nibble_update_pipe" (0 = k_nib_apply, the A/B partner)
    if (g_nib_update_pipe != 0 && g_nib_update_layout != 0 && lgp == 0 && g.shift >= 15) {
        static int ncu = 0;
        if (ncu == 0) {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
            ncu = v;
        }
        const size_t lds_p = (size_t)1 << (g.shift - 1);
        auto kp = k_nib_apply_pipe<MODE>;
        PSK_TRY(set_dyn_lds(kp, lds_p));
        const uint32_t grid = g.nbuckets < (uint32_t)ncu ? g.nbuckets : (uint32_t)ncu;
        hipLaunchKernelGGL(kp, dim3(grid), dim3(kApplyThreads), lds_p, st, (uint32_t *)s->table, s->m, g, (const uint32_t *)cnt, (const uint4 *)part,
                           (unsigned long long *)(s->ctr + PSK_CTR_SATURATED), (uint32_t)(g_nib_update_pipe != 3), flag);  // (3: plain instead of nontemporal table accesses, bench A/B)
        HIP_TRY(hipGetLastError());
        return PSK_OK;
    }
    const size_t lds = (size_t)1 << (g.shift - 1 - lgp);
    auto kern = g_nib_update_layout ? k_nib_apply<MODE, true> : k_nib_apply<MODE, false>;
    PSK_TRY(set_dyn_lds(kern, lds));
    hipLaunchKernelGGL(kern, dim3(g.nbuckets << lgp), dim3(kApplyThreads), lds, st, (uint32_t *)s->table, s->m, g, (const uint32_t *)cnt, (const uint4 *)part,
                       (const uint32_t *)nullptr, (const uint4 *)nullptr, (unsigned long long *)(s->ctr + PSK_CTR_SATURATED), lgp << 8, flag, g);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}
template <bool NEG>
static inline int nib_apply(psk_sketch *s, const PartGeom &g, const void *cnt, const void *part, hipStream_t st)
{
    return nib_apply_mode<NEG ? 1 : 0>(s, g, cnt, part, st);
}

// Pass 1 alone of a unit-weight batch (one round) into the handle's first (second = false) or second bucket buffer: the fused flush of
// the write-combined lists scatters both lists, then folds them in ONE launch (k_nib_apply<2>).  *done = false: not eligible.
static inline int cbf_nib_scatter_only(psk_sketch *s, const Batch &b, bool neg, bool second, PartGeom *g_out, hipStream_t st, bool *done)
{
    *done = false;
    const uint64_t cells = s->m;
    if (g_update_nibble == 0 || !part_wanted(b.n, s->k, 4) || b.n > part_round_keys_two_level(b.n, s->k) || !nib_load_ok(b.n, s->k, cells)) return PSK_OK;
    PartGeom g;
    if (!nib_geometry(cells, true, &g)) return PSK_OK;
    g.k = s->k;
    if (second) { std::swap(s->s_part, s->s_part2); std::swap(s->s_cnt, s->s_cnt2); }  // (launch_scatter fills s_part / s_cnt)
    bool handled = false;
    const int rc = nib_scatter<false>(s, b, nullptr, neg, &g, st, &handled);
    if (second) { std::swap(s->s_part, s->s_part2); std::swap(s->s_cnt, s->s_cnt2); }
    PSK_TRY(rc);
    *g_out = g;
    *done = handled;
    return PSK_OK;
}

// CountingBloomFilter unit-weight adds / decrements into 2^26 .. 2^29 counters: ONE level of 2^18-counter slices with 4-bit delta
// images (the 32-bit images need the two-level path there).  w01: the per-key weights are known to be 0 or 1 (the amounts of the
// validated remove): keys with 0 send no probes.  Eligible when the batch brings enough probes to pay for the pass over the table.
// opt / flag (decrements): the transactional remove, see SpillCounter
template <bool NEG>
static inline int cbf_unit_nibble(psk_sketch *s, const Batch &b, const uint32_t *w01, hipStream_t st, bool *done, int opt = 0, uint32_t *flag = nullptr)
{
    *done = false;
    const uint64_t cells = s->m;
    if (g_update_nibble == 0 || !part_wanted(b.n, s->k, 4)) return PSK_OK;
    if (b.n * (uint64_t)s->k < cells / 8) return PSK_OK;
    PartGeom g;
    if (!nib_geometry(cells, true, &g)) return PSK_OK;
    g.k = s->k;
    if (!nib_load_ok(b.n, s->k, cells)) return PSK_OK;  // (more than ~2.5 probes per counter: the 32-bit slices take the batch)
    const uint64_t round_keys = part_round_keys_two_level(b.n, s->k);
    for (uint64_t start = 0; start < b.n; start += round_keys) {
        const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
        bool handled = false;
        PSK_TRY(nib_scatter<NEG>(s, sub_batch(b, start, cnt), w01 ? w01 + start : nullptr, NEG, &g, st, &handled, nullptr, opt, flag));
        if (!handled) return PSK_OK;  // (first round: nothing was launched)
        if (NEG && opt == 1) PSK_TRY(nib_apply_mode<3>(s, g, s->s_cnt.p, s->s_part.p, st, flag));
        else if (NEG && opt == 2) PSK_TRY(nib_apply_mode<4>(s, g, s->s_cnt.p, s->s_part.p, st));
        else PSK_TRY(nib_apply<NEG>(s, g, s->s_cnt.p, s->s_part.p, st));
    }
    *done = true;
    return PSK_OK;
}

template <template <bool> class IDX, bool SIGNED, bool NEG>
static inline int counter_add_partitioned(psk_sketch *s, const Batch &b, const uint32_t *w_dev, uint64_t cells, hipStream_t st,
                                   bool *done, int opt = 0, uint32_t *flag = nullptr)
{
    *done = false;
    if (!part_wanted(b.n, s->k, 4)) return PSK_OK;
    if constexpr (!SIGNED) {  // CountingBloomFilter: unit weights (or 0 / 1 amounts) into a big table -> nibble deltas, one level
        if (!w_dev || s->acct.weights01) {
            PSK_TRY(cbf_unit_nibble<NEG>(s, b, w_dev, st, done, opt, flag));
            if (*done) return PSK_OK;  // (weights, if any, stay to be accounted by the caller's stand-alone pass: acct.pending is untouched)
        }
    }
    // unit-weight batches cannot wrap a 32-bit partial sum when n*k < 2^31 (weighted ones are checked on the device)
    if (!w_dev && b.n * (uint64_t)s->k >= (1ULL << 31)) return PSK_OK;
    PartGeom g;
    if (!part_slices(cells, 15, 5, &g, 16384, 7)) return PSK_OK;  // 2^15 counters = 128 KiB per slice
    g.k = s->k;
    unsigned long long *sat2 = (unsigned long long *)(s->ctr + PSK_CTR_SATURATED);
    PartGeom g1;
    uint32_t sub_bits = 0;
    if (two_level_geometry(g, &g1, &sub_bits)) {
        // Pass 2 read-modify-writes every slice of the table: only worth it when the batch brings enough probes
        // (direct atomics into a 1 GiB table run at ~20 G/s; the table RMW at ~4 TB/s)
        if (b.n * (uint64_t)s->k < cells / 8) return PSK_OK;
        const uint64_t round_keys = part_round_keys_two_level(b.n, s->k);
        SpillCounter<SIGNED> spill{(uint32_t *)s->table, w_dev == nullptr, NEG, sat2, flag, opt};
        TallyArgs ta2{};
        ta2.opt = (uint32_t)opt;
        ta2.flag = flag;
        for (uint64_t start = 0; start < b.n; start += round_keys) {
            const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
            const Batch sub = sub_batch(b, start, cnt);
            bool handled = false;
            PayWeight pay;  // level 1 is always inline: weight (or 1) << shift1 | index in the coarse bucket
            PSK_TRY(pay_weights(s, w_dev, start, &pay));
            PSK_TRY(with_part_source(sub, &handled, [&](auto src) {
                using Src = decltype(src);
                return with_kt<Src>(s->k, [&](auto kt) {
                    constexpr int KT = decltype(kt)::value;
                    return launch_scatter<Src, IDX<kTuPow2>, PayWeight, SpillCounter<SIGNED>, KT>(s, src, IDX<kTuPow2>{s->md}, pay, spill, &g1, cnt, st);
                });
            }));
            if (!handled) return PSK_OK;
            PSK_TRY(fold_tally(s, pay, g1.nwg, st));
            PartGeom g2 = g;
            const size_t lds = (size_t)4 << g2.shift;
            if (w_dev) {
                PSK_TRY((split_level2<2, SpillCounter<SIGNED>>(s, g1, &g2, sub_bits, cnt * (uint64_t)s->k, spill, st)));
                auto kern = k_counter_apply<SIGNED, true, NEG>;
                PSK_TRY(set_dyn_lds(kern, lds));
                hipLaunchKernelGGL(kern, dim3(g2.nbuckets), dim3(kApplyThreads), lds, st, (uint32_t *)s->table, cells, g2,
                                   (const uint32_t *)s->s_cnt2.p, (const uint4 *)s->s_part2.p, (long long *)s->ctr, sat2, ta2);
            } else {
                PSK_TRY((split_level2<1, SpillCounter<SIGNED>>(s, g1, &g2, sub_bits, cnt * (uint64_t)s->k, spill, st)));
                auto kern = k_counter_apply<SIGNED, false, NEG>;
                PSK_TRY(set_dyn_lds(kern, lds));
                hipLaunchKernelGGL(kern, dim3(g2.nbuckets), dim3(kApplyThreads), lds, st, (uint32_t *)s->table, cells, g2,
                                   (const uint32_t *)s->s_cnt2.p, (const uint4 *)s->s_part2.p, (long long *)s->ctr, sat2, ta2);
            }
            HIP_TRY(hipGetLastError());
        }
        if (w_dev) s->acct.pending = false;  // pass 1 summed the weights
        *done = true;
        return PSK_OK;
    }
    if (g.nbuckets > (uint32_t)kPartMaxBuckets) return PSK_OK;
    // weighted CountMinSketch adds: the compact probe format (PayWeightSmall) when the table allows it (the stage holds weight << 27 | cell)
    // and the previous weighted batches brought no weight outside 0 .. 15
    bool small_fmt = false;
    if constexpr (SIGNED && !NEG) small_fmt = w_dev != nullptr && cells < (1ULL << kSmallWeightShift) && g.shift <= 15 && small_weights_wanted(s);
    const bool small_asked = small_fmt;
    const uint64_t round_keys = part_round_keys_big_table(b.n, s->k, w_dev ? (small_fmt ? PayWeightSmall::group : PayWeight::group) : PayUnit::group, s->padded_bytes);
    unsigned long long *sat = (unsigned long long *)(s->ctr + PSK_CTR_SATURATED);
    for (uint64_t start = 0; start < b.n; start += round_keys) {
        const uint64_t cnt = b.n - start < round_keys ? b.n - start : round_keys;
        const Batch sub = sub_batch(b, start, cnt);
        bool handled = false;
        PayWeight payw;
        PSK_TRY(pay_weights(s, w_dev, start, &payw));
        PSK_TRY(with_part_source(sub, &handled, [&](auto src) {
            using Src = decltype(src);
            return with_kt<Src>(s->k, [&](auto kt) {
                constexpr int KT = decltype(kt)::value;
                SpillCounter<SIGNED> spill{(uint32_t *)s->table, w_dev == nullptr, NEG, sat, flag, opt};
                if (w_dev) {
                    if constexpr (SIGNED && !NEG && std::is_same<Src, KeysFixed16>::value) {  // (the fast key layout only: instantiations)
                        if (small_fmt) {
                            const PayWeightSmall pay{payw.w, payw.tally, payw.weights_signed};
                            return launch_scatter<Src, IDX<kTuPow2>, PayWeightSmall, SpillCounter<SIGNED>, KT>(s, src, IDX<kTuPow2>{s->md}, pay, spill, &g, cnt, st);
                        }
                    } else {
                        small_fmt = false;
                    }
                    const PayWeight pay = payw;
                    return launch_scatter<Src, IDX<kTuPow2>, PayWeight, SpillCounter<SIGNED>, KT>(s, src, IDX<kTuPow2>{s->md}, pay, spill, &g, cnt, st);
                }
                return launch_scatter<Src, IDX<kTuPow2>, PayUnit, SpillCounter<SIGNED>, KT>(s, src, IDX<kTuPow2>{s->md}, PayUnit{}, spill, &g, cnt, st);
            });
        }));
        if (!handled) return PSK_OK;
        TallyArgs ta;  // pass 1's weight sums are booked by pass 2 (no launch between the passes)
        PSK_TRY(tally_args(s, payw, g.nwg, &ta));
        ta.opt = (uint32_t)opt;
        ta.flag = flag;
        const size_t lds = (size_t)4 << g.shift;
        if (w_dev && small_fmt) {
            if constexpr (SIGNED && !NEG) {
                auto kern = k_counter_apply<SIGNED, 2, NEG>;
                PSK_TRY(set_dyn_lds(kern, lds));
                hipLaunchKernelGGL(kern, dim3(g.nbuckets), dim3(kApplyThreads), lds, st, (uint32_t *)s->table, cells, g,
                                   (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p, (long long *)s->ctr, sat, ta);
            }
        } else if (w_dev) {
            auto kern = k_counter_apply<SIGNED, true, NEG>;
            PSK_TRY(set_dyn_lds(kern, lds));
            hipLaunchKernelGGL(kern, dim3(g.nbuckets), dim3(kApplyThreads), lds, st, (uint32_t *)s->table, cells, g,
                               (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p, (long long *)s->ctr, sat, ta);
        } else {
            auto kern = k_counter_apply<SIGNED, false, NEG>;
            PSK_TRY(set_dyn_lds(kern, lds));
            hipLaunchKernelGGL(kern, dim3(g.nbuckets), dim3(kApplyThreads), lds, st, (uint32_t *)s->table, cells, g,
                               (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p, (long long *)s->ctr, sat, ta);
        }
        HIP_TRY(hipGetLastError());
    }
    if (w_dev) s->acct.pending = false;  // pass 1 summed the weights
    if (small_asked && small_fmt) ++g_small_weights_used;  // (calls that travelled in the compact format; option "cms_small_weights_used": tests)
    *done = true;
    return PSK_OK;
}
