// partitioned CountingBloomFilter lookup launcher (own translation unit: parallel build)
#include "psk_part_lookup.hpp"

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
