// psk_part_dispatch.hip -- picks the power-of-two / Barrett build of every partitioned launcher (psk_host.hpp: each launcher
// translation unit is compiled twice, -DPSK_TU_POW2=1 and =0, so that the instantiations build in parallel)
#include "psk_host.hpp"

#define PSK_DISPATCH(name, params, call)          \
    int name params                               \
    {                                             \
        return s->pow2 ? name##_v1 call : name##_v0 call; \
    }

PSK_DISPATCH(bloom_add_partitioned, (psk_sketch *s, const Batch &b, hipStream_t st, bool *done), (s, b, st, done))
PSK_DISPATCH(bloom_check_partitioned, (psk_sketch *s, const Batch &b, uint8_t *out_dev, hipStream_t st, bool *done), (s, b, out_dev, st, done))
PSK_DISPATCH(bloom_check_begin_partitioned, (psk_sketch *s, const Batch &b, hipStream_t st), (s, b, st))
PSK_DISPATCH(bloom_check_finish_partitioned, (psk_sketch *s, uint8_t *out_dev, hipStream_t st, bool *redo), (s, out_dev, st, redo))
PSK_DISPATCH(cms_add_partitioned, (psk_sketch *s, const Batch &b, const uint32_t *w, hipStream_t st, bool *done), (s, b, w, st, done))
PSK_DISPATCH(cms_remove_partitioned, (psk_sketch *s, const Batch &b, const uint32_t *w, hipStream_t st, bool *done), (s, b, w, st, done))
PSK_DISPATCH(cbf_add_partitioned, (psk_sketch *s, const Batch &b, const uint32_t *w, hipStream_t st, bool *done), (s, b, w, st, done))
PSK_DISPATCH(cbf_remove_partitioned, (psk_sketch *s, const Batch &b, const uint32_t *w, hipStream_t st, bool *done, int opt, uint32_t *flag), (s, b, w, st, done, opt, flag))
PSK_DISPATCH(cms_check_partitioned, (psk_sketch *s, const Batch &b, int query, int64_t els, void *out_dev, hipStream_t st, bool *done),
             (s, b, query, els, out_dev, st, done))
PSK_DISPATCH(cbf_check_partitioned, (psk_sketch *s, const Batch &b, uint32_t kk, uint32_t *out_dev, hipStream_t st, bool *done), (s, b, kk, out_dev, st, done))
PSK_DISPATCH(cbf_scat_append, (psk_sketch *s, const Batch &b, int neg, hipStream_t st, bool *done), (s, b, neg, st, done))
PSK_DISPATCH(cbf_remove_fast_begin, (psk_sketch *s, const Batch &b, hipStream_t st, bool *launched), (s, b, st, launched))
PSK_DISPATCH(cbf_remove_fast_undo, (psk_sketch *s, hipStream_t st), (s, st))
PSK_DISPATCH(cbf_unit_multi_partitioned, (psk_sketch *s, const void *const *base, const uint64_t *start, uint32_t nb, uint64_t n, int neg, hipStream_t st, bool *done),
             (s, base, start, nb, n, neg, st, done))
PSK_DISPATCH(cbf_window_fold, (psk_sketch *s, const WinBatchHost *wb, uint32_t nb, hipStream_t st, bool *launched, bool *ok), (s, wb, nb, st, launched, ok))
PSK_DISPATCH(cbf_nib_scatter, (psk_sketch *s, const Batch &b, int neg, int second, PartGeom *g_out, hipStream_t st, bool *done), (s, b, neg, second, g_out, st, done))
