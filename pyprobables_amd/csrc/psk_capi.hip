// psk_capi.hip -- the extern "C" boundary (include/psk.h) over the gfx950 kernels in psk_device.hpp.
// Host side: handle bookkeeping, host<->device staging for PSK_HOST buffers, launch geometry.
#include "psk_host.hpp"
#include "psk_digest.hpp"
#include "psk_nibble.hpp"
#include "psk_window.hpp"

#include <chrono>
#include <map>
#include <mutex>
#include <utility>

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}


extern "C" const char *psk_last_error(void) { return g_err; }
extern "C" int psk_version(void) { return 100; }

extern "C" int psk_device_count(int *count)
{
    if (!count) return fail(PSK_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(PSK_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return PSK_OK;
}

static uint64_t round16(uint64_t b) { return (b + 15) & ~15ULL; }
extern "C" uint64_t psk_bloom_table_bytes(uint64_t m_bits) { return round16((m_bits + 7) / 8); }
extern "C" uint64_t psk_cbf_table_bytes(uint64_t m) { return round16(4 * m); }
extern "C" uint64_t psk_cms_table_bytes(uint64_t width, uint32_t depth) { return round16(4 * width * (uint64_t)depth); }

static Mod make_mod(uint64_t m, bool *pow2)
{
    Mod md;
    md.m = m;
    *pow2 = (m & (m - 1)) == 0;
    md.mask = m - 1;
    md.magic = *pow2 ? 0 : (uint64_t)((((unsigned __int128)1) << 64) / m);
    return md;
}

int ensure(DevBuf &b, uint64_t bytes)
{
    if (bytes <= b.cap) return PSK_OK;
    if (b.p) HIP_TRY(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    uint64_t cap = bytes + bytes / 4 + 256;
    HIP_TRY(hipMalloc(&b.p, cap));
    b.cap = cap;
    return PSK_OK;
}


static int grid_for(uint64_t n)
{
    uint64_t g = (n + kBlock - 1) / kBlock;
    const uint64_t cap = 256ULL * 16;  // 256 CUs x 16 blocks: grid-stride beyond that
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return (int)g;
}

static int create_common(int kind, uint64_t m, uint32_t k, uint64_t padded, uint64_t logical, int device,
                         void *ext_table, psk_sketch **out)
{
    if (!out) return fail(PSK_EINVAL, "out handle pointer is NULL");
    *out = nullptr;
    if (m == 0 || k == 0) return fail(PSK_EINVAL, "table dimensions must be > 0 (m=%llu k=%u)", (unsigned long long)m, k);
    if (m >> 62) return fail(PSK_EINVAL, "table too large");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(PSK_ENODEV, "no HIP device available (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(PSK_EINVAL, "device %d out of range [0,%d)", device, ndev);
    PSK_USE_DEVICE(device);
    psk_sketch *s = new (std::nothrow) psk_sketch();
    if (!s) return fail(PSK_ENOMEM, "host allocation failed");
    s->kind = kind;
    s->device = device;
    s->m = m;
    s->k = k;
    s->md = make_mod(m, &s->pow2);
    s->padded_bytes = padded;
    s->logical_bytes = logical;
    s->owns_table = ext_table == nullptr;
    s->shadow.exposed = ext_table != nullptr;  // (until the caller declares the table private to this handle: psk_sketch_set_option)
    s->table = ext_table;
    s->ctr = nullptr;
    if (s->owns_table) {
        hipError_t e2 = hipMalloc(&s->table, padded);
        if (e2 != hipSuccess) {
            delete s;
            return fail(PSK_ENOMEM, "hipMalloc(%llu bytes) for the table failed: %s", (unsigned long long)padded,
                        hipGetErrorString(e2));
        }
        e2 = hipMemset(s->table, 0, padded);
        if (e2 != hipSuccess) {
            hipFree(s->table);
            delete s;
            return fail(PSK_EHIP, "hipMemset failed: %s", hipGetErrorString(e2));
        }
    }
    hipError_t e3 = hipMalloc((void **)&s->ctr, sizeof(long long) * PSK_CTR_COUNT);
    if (e3 == hipSuccess) e3 = hipMemset(s->ctr, 0, sizeof(long long) * PSK_CTR_COUNT);
    if (e3 != hipSuccess) {
        if (s->owns_table) hipFree(s->table);
        delete s;
        return fail(PSK_ENOMEM, "counter allocation failed: %s", hipGetErrorString(e3));
    }
    *out = s;
    return PSK_OK;
}

extern "C" int psk_bloom_create(uint64_t m_bits, uint32_t k, int device, void *ext_table, psk_sketch **out)
{
    return create_common(PSK_KIND_BLOOM, m_bits, k, psk_bloom_table_bytes(m_bits), (m_bits + 7) / 8, device, ext_table, out);
}

extern "C" int psk_cbf_create(uint64_t m, uint32_t k, int device, void *ext_table, psk_sketch **out)
{
    return create_common(PSK_KIND_CBF, m, k, psk_cbf_table_bytes(m), 4 * m, device, ext_table, out);
}

extern "C" int psk_cms_create(uint64_t width, uint32_t depth, int device, void *ext_table, psk_sketch **out)
{
    return create_common(PSK_KIND_CMS, width, depth, psk_cms_table_bytes(width, depth), 4 * width * (uint64_t)depth,
                         device, ext_table, out);
}

static inline void ho_apply(const psk_sketch *s);  // (the handle's option overrides -> this thread's effective values; defined with the options below)
extern "C" int psk_destroy(psk_sketch *s)
{
    if (!s) return PSK_OK;
    DeviceScope scope;
    (void)scope.enter(s->device);
    if (!s->owns_table && s->table) {  // the caller's table outlives the handle: write-combined updates still waiting must reach it
        ho_apply(s);  // (under THIS sketch's options, not those of whatever handle the thread used last: remove_exact, update_window ...)
        if (flush_combined(s, nullptr) == PSK_OK) (void)hipStreamSynchronize(nullptr);
    }
    if (s->owns_table && s->table) hipFree(s->table);
    if (s->ctr) hipFree(s->ctr);
    if (s->lk.dev) hipFree(s->lk.dev);
    if (s->lk.pin) hipHostFree((void *)s->lk.pin);
    if (s->wt.pin) hipHostFree((void *)s->wt.pin);
    for (DevBuf *b : {&s->s_keys, &s->s_offs, &s->s_w, &s->s_out, &s->s_aux, &s->s_part, &s->s_cnt, &s->s_flag, &s->s_tflag, &s->s_part2, &s->s_cnt2, &s->s_merge, &s->s_vals, &s->s_perm, &s->s_run, &s->s_tally,
                      &s->comb.add.keys, &s->comb.add.w, &s->comb.rem.keys, &s->comb.rem.w, &s->scat.add.part, &s->scat.add.cnt, &s->scat.rem.part, &s->scat.rem.cnt, &s->s_brw, &s->shadow.img,
                      &s->win.keys, &s->s_snap, &s->s_wstat, &s->s_phase}) {
        if (b->p) hipFree(b->p);
        if (b->pin) hipHostFree(b->pin);
    }
    if (s->win.pin) hipHostFree(s->win.pin);
    if (s->mbox) hipHostFree((void *)s->mbox);
    if (s->scat.ev) hipEventDestroy(s->scat.ev);
    delete s;
    return PSK_OK;
}

// ---- per-sketch options (psk_host.hpp HandleOpt): process defaults, names, and the refresh every handle entry point does
static const char *const kHoNames[HO_COUNT] = {"partition_min_keys", "cbf_lookup_shadow", "auto_combine", "update_window", "update_window_keys",
                                                "scratch_budget_bytes", "remove_exact", "bloom_lookup"};
static int64_t d_opt[HO_COUNT] = {1 << 16, 1, 1, 1, 1 << 27, 0, 1, 2};  // (== the thread-local variables' initial values)
static int64_t *ho_var(int i)
{
    switch (i) {
        case HO_PART_MIN_KEYS: return &g_part_min_keys;
        case HO_CBF_SHADOW: return &g_cbf_shadow;
        case HO_AUTO_COMBINE: return &g_auto_combine;
        case HO_WINDOW: return &g_window;
        case HO_WINDOW_KEYS: return &g_window_keys;
        case HO_SCRATCH_BUDGET: return &g_scratch_budget;
        case HO_REMOVE_EXACT: return &g_remove_exact;
        default: return &g_bloom_lookup;
    }
}
static int ho_index(const char *name)
{
    for (int i = 0; i < HO_COUNT; ++i)
        if (!strcmp(name, kHoNames[i])) return i;
    return -1;
}
static inline void ho_apply(const psk_sketch *s)
{
    for (int i = 0; i < HO_COUNT; ++i) *ho_var(i) = (s && s->opt[i] != kHoUnset) ? s->opt[i] : __atomic_load_n(&d_opt[i], __ATOMIC_RELAXED);
}

#define CHECK_HANDLE_RO(s, want_kind)                                                    \
    do {                                                                                 \
        if (!(s)) return fail(PSK_EINVAL, "sketch handle is NULL");                      \
        if ((want_kind) >= 0 && (s)->kind != (want_kind))                                \
            return fail(PSK_EINVAL, "wrong sketch kind %d for this call", (s)->kind);    \
        ho_apply(s);                                                                     \
    } while (0);                                                                         \
    PSK_USE_DEVICE((s)->device)
// every entry point that may change the table (or hands its pointer out) moves the table's version on: what was derived from the
// table -- psk_sketch::shadow, the 4-bit images of the nibble-slice lookup -- is stale from here on.  Read-only entries: _RO.
#define CHECK_HANDLE(s, want_kind)                                                       \
    CHECK_HANDLE_RO(s, want_kind);                                                       \
    ++(s)->table_version

// table (padded to 16 bytes) and the handle's counter block, zeroed by one kernel
// (round 4: nontemporal stores measured SLOWER for this write-only sweep -- 247 vs 215 us for the 1 GiB table of cfg 4 -- unlike the
// read-modify-write passes; plain stores stay)
static __global__ __launch_bounds__(kBlock) void k_clear(uint4 *tab, uint64_t nvec, long long *ctr)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint4 z = make_uint4(0, 0, 0, 0);
    uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {  // four 16-byte stores in flight per lane
        tab[i] = z;
        tab[i + stride] = z;
        tab[i + 2 * stride] = z;
        tab[i + 3 * stride] = z;
    }
    for (; i < nvec; i += stride) tab[i] = z;
    if (blockIdx.x == 0 && threadIdx.x < PSK_CTR_COUNT) ctr[threadIdx.x] = 0;
}

static int scat_drop(psk_sketch *s, hipStream_t st);  // forget the scattered write-combined updates (defined with them below)

extern "C" int psk_clear(psk_sketch *s, void *stream)
{
    CHECK_HANDLE(s, -1);
    hipStream_t st = (hipStream_t)stream;
    s->comb.add.n = s->comb.rem.n = 0;  // write-combined updates that have not reached the table are cleared with it
    s->comb.add.unit = s->comb.rem.unit = true;
    s->comb.badd.clear();
    s->comb.brem.clear();
    s->win.n = s->win.copied = 0;  // (the update window too; a window's back-off is a property of the stream and stays)
    s->win.batches.clear();
    PSK_TRY(scat_drop(s, st));
    // one launch for the table AND the counter block (two fills are two ~5 us launches; clear sits in every bench step)
    const uint64_t nvec = s->padded_bytes / 16;
    uint64_t grid = (nvec + kBlock * 4 - 1) / (kBlock * 4);
    if (grid > 2048) grid = 2048;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(k_clear, dim3((unsigned)grid), dim3(kBlock), 0, st, (uint4 *)s->table, nvec, s->ctr);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

extern "C" int psk_synchronize(psk_sketch *s, void *stream)
{
    CHECK_HANDLE_RO(s, -1);
    PSK_TRY(flush_combined(s, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return PSK_OK;
}

extern "C" int psk_table_info(psk_sketch *s, void **dev_ptr, uint64_t *padded_bytes, uint64_t *logical_bytes)
{
    if (!s) return fail(PSK_EINVAL, "sketch handle is NULL");
    ++s->table_version;  // the pointer leaves the engine: whoever holds it may write
    if (dev_ptr) s->shadow.exposed = true;  // ... and later, too: no kept images until the holder says it is done (psk_rescan_bound)
    if (dev_ptr) *dev_ptr = s->table;
    if (padded_bytes) *padded_bytes = s->padded_bytes;
    if (logical_bytes) *logical_bytes = s->logical_bytes;
    return PSK_OK;
}

extern "C" int psk_read_table(psk_sketch *s, void *dst_host, uint64_t nbytes, void *stream)
{
    CHECK_HANDLE_RO(s, -1);
    if (!dst_host || nbytes > s->padded_bytes) return fail(PSK_EINVAL, "bad read_table arguments");
    hipStream_t st = (hipStream_t)stream;
    PSK_TRY(flush_combined(s, st));
    HIP_TRY(hipMemcpyAsync(dst_host, s->table, nbytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PSK_OK;
}

extern "C" int psk_write_table(psk_sketch *s, const void *src_host, uint64_t nbytes, void *stream)
{
    CHECK_HANDLE(s, -1);
    if (!src_host || nbytes > s->padded_bytes) return fail(PSK_EINVAL, "bad write_table arguments");
    hipStream_t st = (hipStream_t)stream;
    s->comb.add.n = s->comb.rem.n = 0;  // the table is replaced: pending updates go with the old contents
    s->comb.add.unit = s->comb.rem.unit = true;
    s->comb.badd.clear();
    s->comb.brem.clear();
    s->win.n = s->win.copied = 0;
    s->win.batches.clear();
    PSK_TRY(scat_drop(s, st));
    HIP_TRY(hipMemsetAsync(s->table, 0, s->padded_bytes, st));
    HIP_TRY(hipMemcpyAsync(s->table, src_host, nbytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(s->ctr, 0, sizeof(long long) * PSK_CTR_COUNT, st));
    if (s->kind != PSK_KIND_BLOOM) {
        // re-seed the wrap-free bound from the loaded counters
        const uint64_t nel = s->logical_bytes / 4;
        hipLaunchKernelGGL(k_absmax, dim3(grid_for(nel)), dim3(kBlock), 0, st, (const uint32_t *)s->table, nel,
                           s->kind == PSK_KIND_CMS ? 1 : 0, s->ctr);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(st));
    return PSK_OK;
}

extern "C" int psk_rescan_bound(psk_sketch *s, void *stream)
{
    CHECK_HANDLE(s, -1);
    // "I wrote through the pointer I hold, and I am done": kept images may be built again (a holder of a caller-owned table must have
    // promised to announce every outside write: option "table_private")
    if (s->owns_table || s->table_private) s->shadow.exposed = false;
    if (s->kind == PSK_KIND_BLOOM) return PSK_OK;
    hipStream_t st = (hipStream_t)stream;
    PSK_TRY(flush_combined(s, st));
    HIP_TRY(hipMemsetAsync(s->ctr + PSK_CTR_ABS_BOUND, 0, sizeof(long long), st));
    const uint64_t nel = s->logical_bytes / 4;
    hipLaunchKernelGGL(k_absmax, dim3(grid_for(nel)), dim3(kBlock), 0, st, (const uint32_t *)s->table, nel,
                       s->kind == PSK_KIND_CMS ? 1 : 0, s->ctr);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

extern "C" int psk_get_counters(psk_sketch *s, int64_t out[PSK_CTR_COUNT], void *stream)
{
    CHECK_HANDLE_RO(s, -1);
    if (!out) return fail(PSK_EINVAL, "out is NULL");
    hipStream_t st = (hipStream_t)stream;
    PSK_TRY(flush_combined(s, st));  // the tallies describe every update handed over so far
    HIP_TRY(hipMemcpyAsync(out, s->ctr, sizeof(long long) * PSK_CTR_COUNT, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PSK_OK;
}

extern "C" int psk_reset_counters(psk_sketch *s, void *stream)
{
    CHECK_HANDLE(s, -1);
    // keep the wrap-free bound: it describes the table, not the batch history
    HIP_TRY(hipMemsetAsync(s->ctr, 0, sizeof(long long) * PSK_CTR_ABS_BOUND, (hipStream_t)stream));
    return PSK_OK;
}

int raise_dyn_lds(const void *kernel, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> granted;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = granted[{kernel, dev}];
    if (bytes > have) {
        HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    return PSK_OK;
}

// ---- completion mailbox of tiny PSK_HOST batches (mailbox_post, psk_device.hpp): the kernel stores the call's sequence number into a
// pinned word behind its results (which it wrote into a pinned page) and the host polls that word -- a value-returning single-key call
// ends when its answer is in host memory; the stream wait (a barrier packet, its signal, the runtime's bookkeeping: ~5 us of a ~15 us
// call) is what the reference's per-key callers would otherwise pay on every `key in blm`.  The poll gives up after host_poll_us
// microseconds (option; 0 = never poll) and falls back to the stream wait, which also reports a kernel that died.
int64_t g_host_poll_us = 200;
struct Mailbox {
    volatile uint32_t *word = nullptr;  // nullptr: not armed -- finish() waits for the stream
    uint32_t seq = 0;
    uint32_t *timeouts = nullptr;       // the handle's count of polls in a row that gave up (mailbox_arm)
    uint32_t *dev() const { return const_cast<uint32_t *>(word); }
};
static void mailbox_disarm(Mailbox *mb) { mb->word = nullptr; }
static int mailbox_arm(psk_sketch *s, int where, uint64_t n, bool out_pinned, Mailbox *mb)
{
    mb->word = nullptr;
    if (where != PSK_HOST || n == 0 || n > kBlock || !out_pinned || g_host_poll_us <= 0) return PSK_OK;
    // Eight polls in a row that gave up -- a stream that always has long work queued in front of the call, or pinned memory the host does not
    // see device stores to while the kernel runs -- and the handle stops paying host_poll_us per call for nothing: stream waits, one more try
    // every 1024 calls
    if (s->mbox_timeouts >= 8 && (++s->mbox_skipped & 1023u) != 0) return PSK_OK;
    if (!s->mbox) {
        void *pp = nullptr;
        HIP_TRY(hipHostMalloc(&pp, 64, hipHostMallocDefault));
        *(volatile uint32_t *)pp = 0;
        s->mbox = (volatile uint32_t *)pp;
    }
    if (++s->mbox_seq == 0) ++s->mbox_seq;  // (never the word's initial 0)
    mb->word = s->mbox;
    mb->seq = s->mbox_seq;
    mb->timeouts = &s->mbox_timeouts;
    return PSK_OK;
}
static bool mailbox_wait(const Mailbox *mb)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 1;; ++spins) {
        if (__atomic_load_n(mb->word, __ATOMIC_ACQUIRE) == mb->seq) return true;
        __builtin_ia32_pause();
        if ((spins & 1023) == 0 &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > g_host_poll_us)
            return false;
    }
}

// small-transfer fast path: lazily allocated pinned (host-coherent, device-visible) page per scratch buffer
static int pinned(DevBuf &b, void **out)
{
    if (!b.pin) HIP_TRY(hipHostMalloc(&b.pin, kPinBytes, hipHostMallocDefault));
    *out = b.pin;
    return PSK_OK;
}

// ---------------------------------------------------------- key batches
static int elem_bytes(int layout) { return layout == PSK_KEYS_VARLEN32 ? 4 : (layout == PSK_KEYS_HASHES ? 8 : 1); }

// Validate, and for PSK_HOST stage the batch into the handle's device scratch.
static int stage_batch(DevBuf &kbuf, DevBuf &obuf, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                       uint32_t key_len, int where, hipStream_t st, Batch *b)
{
    if (layout < PSK_KEYS_FIXED || layout > PSK_KEYS_HASHES) return fail(PSK_EINVAL, "unknown key layout %d", layout);
    if (where != PSK_HOST && where != PSK_DEVICE) return fail(PSK_EINVAL, "`where` must be PSK_HOST or PSK_DEVICE");
    const bool varlen = layout == PSK_KEYS_VARLEN8 || layout == PSK_KEYS_VARLEN32;
    if (n && !data && !(layout == PSK_KEYS_FIXED && key_len == 0)) return fail(PSK_EINVAL, "key data pointer is NULL");
    if (n && varlen && !offsets) return fail(PSK_EINVAL, "variable-length layout needs offsets[n+1]");
    b->layout = layout;
    b->n = n;
    b->key_len = key_len;
    b->data = data;
    b->offs = offsets;
    if (where == PSK_DEVICE || n == 0) return PSK_OK;
    uint64_t nbytes;
    if (varlen) {
        const uint64_t total = offsets[n] - offsets[0];
        if (offsets[0] != 0) return fail(PSK_EINVAL, "host offsets must start at 0");
        nbytes = total * (uint64_t)elem_bytes(layout);
        if ((n + 1) * 8 <= kPinBytes) {
            void *pp;
            PSK_TRY(pinned(obuf, &pp));
            memcpy(pp, offsets, (n + 1) * 8);
            b->offs = (const uint64_t *)pp;
        } else {
            PSK_TRY(ensure(obuf, (n + 1) * 8));
            HIP_TRY(hipMemcpyAsync(obuf.p, offsets, (n + 1) * 8, hipMemcpyHostToDevice, st));
            b->offs = (const uint64_t *)obuf.p;
        }
    } else {
        nbytes = n * (uint64_t)key_len * (uint64_t)elem_bytes(layout);
    }
    if (nbytes <= kPinBytes) {  // tiny batch (single-key API): the kernel reads the keys straight from pinned host memory
        void *pp;
        PSK_TRY(pinned(kbuf, &pp));
        if (nbytes) memcpy(pp, data, nbytes);
        b->data = pp;
        return PSK_OK;
    }
    PSK_TRY(ensure(kbuf, nbytes));
    HIP_TRY(hipMemcpyAsync(kbuf.p, data, nbytes, hipMemcpyHostToDevice, st));
    b->data = kbuf.p;
    return PSK_OK;
}

// Dispatch a functor over the concrete key-source type of a batch.
template <class F>
static int with_source(const Batch &b, F &&f)
{
    switch (b.layout) {
        case PSK_KEYS_FIXED:
            if (b.key_len == 16 && ((uintptr_t)b.data & 15) == 0) return f(KeysFixed16{(const uint4 *)b.data});
            if (b.key_len == 8 && ((uintptr_t)b.data & 7) == 0) return f(KeysFixed8{(const uint2 *)b.data});
            if (b.key_len == 32 && ((uintptr_t)b.data & 15) == 0) return f(KeysFixed32{(const uint4 *)b.data});
            if (b.key_len % 4 == 0 && ((uintptr_t)b.data & 3) == 0) return f(KeysFixed<true>{(const uint8_t *)b.data, b.key_len, b.n});
            return f(KeysFixed<false>{(const uint8_t *)b.data, b.key_len, b.n});
        case PSK_KEYS_VARLEN8: return f(KeysVarlen<uint8_t>{(const uint8_t *)b.data, b.offs, b.n});
        case PSK_KEYS_VARLEN32: return f(KeysVarlen<uint32_t>{(const uint32_t *)b.data, b.offs, b.n});
        case PSK_KEYS_HASHES: return f(KeysHashes{(const uint64_t *)b.data, b.key_len});
    }
    return fail(PSK_EINVAL, "unknown key layout");
}

// ONE fixed-layout key of a PSK_HOST call that ends on the mailbox travels in the kernel arguments (KeysInline64, psk_device.hpp), not
// through the pinned page stage_batch filled; `data` is the caller's pointer.  -> the source to launch with, or nullptr (the batch's own)
static const KeysInline64 *inline_key(int layout, const void *data, uint64_t n, uint32_t key_len, const Mailbox &mb, KeysInline64 *k)
{
    if (!mb.word || n != 1 || layout != PSK_KEYS_FIXED || key_len > sizeof k->w) return nullptr;
    memset(k->w, 0, sizeof k->w);
    if (key_len) memcpy(k->w, data, key_len);
    k->L = key_len;
    return k;
}
template <class F>
static int with_source_one(const Batch &b, const KeysInline64 *one, F &&f)
{
    if (one) return f(*one);
    return with_source(b, f);
}

template <class Src, class Op>
static int launch_apply(const Src &src, const Op &op, uint64_t n, hipStream_t st, Mailbox *mb = nullptr)
{
    if (n == 0) return PSK_OK;
    const uint32_t grid = grid_for(n);
    if (mb && grid != 1) mailbox_disarm(mb);  // (the kernel's one workgroup posts it: see mailbox_post, psk_device.hpp)
    // (thread t of workgroup 0 takes keys t, t + kBlock ...: a batch of up to 64 keys needs one wave)
    hipLaunchKernelGGL((k_apply<Src, Op>), dim3(grid), dim3(n <= 64 ? 64 : kBlock), 0, st, src, op, n, mb ? mb->dev() : nullptr, mb ? mb->seq : 0u);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

static int check_hashes_width(const psk_sketch *s, int layout, uint32_t key_len)
{
    if (layout == PSK_KEYS_HASHES && key_len < s->k)
        return fail(PSK_EINVAL, "pre-hashed batch carries %u hashes per key, the sketch needs %u", key_len, s->k);
    return PSK_OK;
}

// stage an optional per-key vector (weights); returns device pointer or nullptr
template <class T>
static int stage_vec(DevBuf &buf, const T *v, uint64_t n, int where, hipStream_t st, const T **dev)
{
    *dev = v;
    if (!v || where == PSK_DEVICE || n == 0) return PSK_OK;
    if (n * sizeof(T) <= kPinBytes) {
        void *pp;
        PSK_TRY(pinned(buf, &pp));
        memcpy(pp, v, n * sizeof(T));
        *dev = (const T *)pp;
        return PSK_OK;
    }
    PSK_TRY(ensure(buf, n * sizeof(T)));
    HIP_TRY(hipMemcpyAsync(buf.p, v, n * sizeof(T), hipMemcpyHostToDevice, st));
    *dev = (const T *)buf.p;
    return PSK_OK;
}

// output buffer: device pointer to write into (+ copy-back for PSK_HOST)
struct OutBuf {
    void *dev = nullptr;
    void *host = nullptr;
    uint64_t bytes = 0;
    bool is_pinned = false;  // dev is pinned host memory: no device-to-host copy, just a sync and a memcpy
};

static int stage_out(DevBuf &buf, void *out, uint64_t bytes, int where, OutBuf *o)
{
    o->bytes = bytes;
    if (where == PSK_DEVICE || bytes == 0) {
        o->dev = out;
        return PSK_OK;
    }
    o->host = out;
    if (bytes <= kPinBytes) {
        PSK_TRY(pinned(buf, &o->dev));
        o->is_pinned = true;
        return PSK_OK;
    }
    PSK_TRY(ensure(buf, bytes));
    o->dev = buf.p;
    return PSK_OK;
}

static int finish(int where, const OutBuf *o, hipStream_t st, const Mailbox *mb = nullptr)
{
    if (where == PSK_HOST) {
        const bool copy = o && o->host && o->bytes;
        if (copy && !o->is_pinned) HIP_TRY(hipMemcpyAsync(o->host, o->dev, o->bytes, hipMemcpyDeviceToHost, st));
        bool posted = false;
        if (mb && mb->word) {
            posted = mailbox_wait(mb);
            if (mb->timeouts) *mb->timeouts = posted ? 0u : *mb->timeouts + 1u;
        }
        if (!posted) HIP_TRY(hipStreamSynchronize(st));
        if (copy && o->is_pinned) memcpy(o->host, o->dev, o->bytes);
    }
    return PSK_OK;
}

// ------------------------------------------------- partitioned (large-batch) path: options
int64_t g_part_mode = 1;
__thread int64_t g_part_min_keys = 1 << 16;   // x1 for Bloom inserts, x4 for lookups / counter adds (part_wanted)
int64_t g_part_max_keys = 1 << 26;   // keys per partition round (bounds the bucket buffer: ~2 GB of scratch at k = 7; sized for 288 GB of HBM --
                                     // every round into a big table ends in a pass over the whole table, so fewer, larger rounds)
int64_t g_part_cache_bytes = 240 << 20;  // bucket-buffer budget per round: the part of the 256 MB MALL we count on
int64_t g_part_two_level_slices = 2048;     // tables cut into more slices than this take the two-level path (0 = never)
int64_t g_part_debug = 0;            // ablation bits for bench runs (see PartGeom::dbg); 0 in production
__thread int64_t g_bloom_lookup = 2;
int64_t g_lookup_run_lanes = 0, g_lookup_split = 1, g_part_tile_threads = 0, g_part_slice_bias = 0, g_part_wgs = 0, g_part_even_tiles = 1;
int64_t g_lookup_half = 1;
int64_t g_lookup_collect_threads = 0;   // pass 3 of the counter lookups: 0 = 512-thread workgroups up to 256 slices, 1024 beyond (psk_part_lookup.hpp); 512 / 1024 = forced
int64_t g_remove_dryrun = 1;   // validated unit-weight CBF removes into big tables: optimistic decrement first (psk_nibble.hpp), option "remove_optimistic"
__thread int64_t g_scratch_budget = 0;  // psk_set_option("scratch_budget_bytes"): cap on a handle's partition scratch (more, smaller rounds); 0 = none
int64_t g_lookup_nibble = 1;   // CBF lookups into 2^25 .. 2^29 counters: 4-bit slice images (psk_nibble.hpp) from cells / 16 probes on; 2 = always; 0 = the 32-bit / 16-bit slices or direct
int64_t g_small_weights_used = 0;
int64_t g_small_weights = 1;   // PayWeightSmall for weighted CountMinSketch adds (psk_sketch::wt)
int64_t g_cbf_shadow_hits = 0;
__thread int64_t g_cbf_shadow = 1;  // nibble-slice lookups keep their 4-bit images while the table is unchanged (psk_sketch::shadow; cells / 2 bytes)
int64_t g_nib_nt = 1;   // nontemporal table loads in k_nib_gather (1 GiB lookups 710 -> 656 us per 10 M keys); the fold of k_nib_apply re-writes what it
                        // reads and measured slower with them (795 -> 984 us): never there
int64_t g_nib_min_lg_lookup = 23, g_nib_min_lg_update = 24;  // see nib_geometry (psk_host.hpp); measured crossovers: scripts/ab_nib_threshold.py
int64_t g_nib_update_parts = 1;   // 1 = one workgroup per slice (default: two measured the same, 0.78-0.82 ms per 10 M adds either way), 2 = two, 0 = by slice size; see nib_update_lgparts
int64_t g_nib_update_layout = 1;   // see psk_nibble.hpp (bench A/B)
int64_t g_ragged_sort = 1;    // pass 1's per-tile length sort of ragged keys (A/B: 0 = batch order); option "ragged_sort"
int64_t g_big_table_nt = 1;   // nontemporal table sweeps in the Bloom pass-2 kernels for tables of 128 MiB and more (BASELINE cfg 5); option "big_table_nt"
int64_t g_window_shadow = 0;   // 1 = a successful window fold leaves the lookups' kept 4-bit images up to date (when they exist) instead of stale.  Measured on the
                               // 1 GiB table (scripts/ab_window_shadow.py: rounds of 15 M window updates + a 10 M-key lookup): 2.99 vs 3.00 ms per round -- the
                               // 0.1 ms the lookup saves is within the noise of the round and is partly paid by the fold's 128 MiB of image stores: off
int64_t g_window_shadow_writes = 0;  // folds that did (tests)
int64_t g_window_wide = 1;    // update windows on tables of few slices: the fold with five probe groups per lane and phase and byte-wide group counts (0: three, nibbles)
int64_t g_window_tile = 0;    // keys per pass-1 tile of an update window: 0 = rule (4096 for tables of many slices), 2048 / 4096 forced; option "update_window_tile"
int64_t g_window_image = 4;   // update windows' fold: 4 = nibble images, one workgroup per 2^18-counter slice; 8 = byte images, two per slice (round 4 A/B)
int64_t g_window_nt = 1;   // nontemporal table loads / stores in the update windows' fold (k_win_fold); option "update_window_nt"
int64_t g_nib_gather_pipe = 0;   // 1 = k_nib_gather_pipe (psk_nibble_pipe.hpp: the next slice's table load under this slice's probe walk) when no kept images exist.
                                 // Measured on MI355X: 260 vs 257 us per 10 M keys -- no gain (the register budget allows one 4-piece load step in flight, which
                                 // cannot keep the table stream busy; deeper variants spill: 366 us), so it stays off: k_nib_gather
int64_t g_nib_update_pipe = 1;   // 1 = k_nib_apply_pipe (psk_nibble_pipe.hpp: persistent workgroups, fold of slice s under the probes of slice s + 1; nontemporal
                                 // table accesses: 551 -> 514 us per 10 M adds), 3 = the same with plain accesses (A/B), 0 = k_nib_apply
int64_t g_update_nibble = 1;   // CBF unit-weight adds / decrements into 2^26 .. 2^29 counters: 4-bit delta images, one level; 0 = two-level 32-bit path
int64_t g_part_bins = 1;   // pass 1 through fixed-capacity bins wherever eligible (psk_part_bins.hpp); option "pass1_bins"
int64_t g_part_dense_groups = 40;   // pass 2: segments of fewer groups (mean) are walked end to end (for_each_batch_at); 0 = never
extern PSK_HIDDEN int64_t g_merge_single_rank;  // psk_merge.hip

// ---- process-wide options: ONE table.  Three classes (include/psk.h lists the first by name):
//   supported   tunables of the shipped library a caller may have a reason to touch;
//   threshold   where the engine switches between its kernel families, and test hooks -- tests steer small inputs onto the big-table paths with them;
//   knob        A/B switches of experiments that were measured and dropped: compiled in only with -DPSK_BENCH_KNOBS=1
//               (python -m pyprobables_amd.build --knobs -> libpsk_hip_knobs.so), the shipped library answers "unknown option";
//   read-only   counters tests read back.
// (the per-sketch options -- kHoNames -- are handled in front of the table: their process defaults live in d_opt)
namespace {
enum OptClass { kOptSupported, kOptThreshold, kOptKnob, kOptReadOnly };
struct OptDesc {
    const char *name;
    int64_t *var;
    OptClass cls;
    int64_t lo;  // smallest value accepted (smaller ones are raised to it)
};
constexpr int64_t kAny = INT64_MIN;
const OptDesc kOptions[] = {
    // supported
    {"partition", &g_part_mode, kOptSupported, kAny},
    {"partition_max_keys", &g_part_max_keys, kOptSupported, 1024},
    {"partition_cache_bytes", &g_part_cache_bytes, kOptSupported, kAny},
    {"combine_keys", &g_combine_keys, kOptSupported, kAny},
    {"cms_small_weights", &g_small_weights, kOptSupported, kAny},
    {"pass1_bins", &g_part_bins, kOptSupported, kAny},
    {"merge_single_rank", &g_merge_single_rank, kOptSupported, kAny},
    // thresholds of the path choice, test hooks
    {"partition_two_level_slices", &g_part_two_level_slices, kOptThreshold, kAny},
    {"auto_combine_keys", &g_auto_combine_keys, kOptThreshold, kAny},
    {"tile_threads", &g_part_tile_threads, kOptThreshold, kAny},
    {"even_tiles", &g_part_even_tiles, kOptThreshold, kAny},
    {"dense_walk_groups", &g_part_dense_groups, kOptThreshold, kAny},
    {"lookup_half_slices", &g_lookup_half, kOptThreshold, kAny},
    {"remove_optimistic", &g_remove_dryrun, kOptThreshold, kAny},
    {"lookup_nibble_slices", &g_lookup_nibble, kOptThreshold, kAny},
    {"update_nibble_slices", &g_update_nibble, kOptThreshold, kAny},
    {"nibble_min_lg_lookup", &g_nib_min_lg_lookup, kOptThreshold, 20},
    {"nibble_min_lg_update", &g_nib_min_lg_update, kOptThreshold, 20},
    {"update_window_tile", &g_window_tile, kOptThreshold, kAny},
    {"update_window_wide", &g_window_wide, kOptThreshold, kAny},
    {"update_window_force_fail", &g_window_force_fail, kOptThreshold, kAny},
    {"ragged_sort", &g_ragged_sort, kOptThreshold, kAny},
    {"host_poll_us", &g_host_poll_us, kOptThreshold, kAny},
    // read-only counters
    {"cbf_ordered_replays", &g_cbf_ordered_replays, kOptReadOnly, kAny},
    {"update_window_folds", &g_window_folds, kOptReadOnly, kAny},
    {"update_window_replays", &g_window_replays, kOptReadOnly, kAny},
    {"cms_small_weights_used", &g_small_weights_used, kOptReadOnly, kAny},
    {"cbf_lookup_shadow_hits", &g_cbf_shadow_hits, kOptReadOnly, kAny},
    // retired experiments (bench builds only)
    {"part_debug", &g_part_debug, kOptKnob, kAny},
    {"combine_scatter", &g_combine_scatter, kOptKnob, kAny},
    {"combine_fused_flush", &g_fused_flush, kOptKnob, kAny},
    {"lookup_run_lanes", &g_lookup_run_lanes, kOptKnob, kAny},
    {"lookup_split", &g_lookup_split, kOptKnob, kAny},
    {"lookup_collect_threads", &g_lookup_collect_threads, kOptKnob, kAny},
    {"slice_bias", &g_part_slice_bias, kOptKnob, kAny},
    {"scatter_workgroups", &g_part_wgs, kOptKnob, kAny},
    {"nibble_update_layout", &g_nib_update_layout, kOptKnob, kAny},
    {"nibble_update_parts", &g_nib_update_parts, kOptKnob, kAny},
    {"nibble_update_pipe", &g_nib_update_pipe, kOptKnob, kAny},
    {"nibble_lookup_pipe", &g_nib_gather_pipe, kOptKnob, kAny},
    {"nibble_nt_loads", &g_nib_nt, kOptKnob, kAny},
    {"update_window_nt", &g_window_nt, kOptKnob, kAny},
    {"update_window_image", &g_window_image, kOptKnob, kAny},
    {"update_window_shadow", &g_window_shadow, kOptKnob, kAny},
    {"update_window_shadow_writes", &g_window_shadow_writes, kOptKnob, kAny},
    {"big_table_nt", &g_big_table_nt, kOptKnob, kAny},
};
const OptDesc *opt_find(const char *name)
{
    for (const OptDesc &d : kOptions)
        if (!strcmp(name, d.name)) return (d.cls == kOptKnob && !kBenchKnobs) ? nullptr : &d;
    return nullptr;
}
}  // namespace

extern "C" int psk_set_option(const char *name, int64_t value)
{
    if (!name) return fail(PSK_EINVAL, "option name is NULL");
    if (const int i = ho_index(name); i >= 0) {  // a default of the per-sketch options (sketches without an override follow it)
        if (i == HO_PART_MIN_KEYS && value < 1) value = 1;
        __atomic_store_n(&d_opt[i], value, __ATOMIC_RELAXED);
        *ho_var(i) = value;
        return PSK_OK;
    }
    const OptDesc *d = opt_find(name);
    if (!d || d->cls == kOptReadOnly) return fail(PSK_EINVAL, d ? "option %s is read-only" : "unknown option %s", name);
    if (value < d->lo) value = d->lo;
    *d->var = value;
    return PSK_OK;
}

// bench-only: phase cycle totals of the last pass-1 launch (valid when part_debug & 32); zeroes them afterwards
extern "C" int psk_debug_phase_profile(psk_sketch *s, uint32_t nbuckets, uint32_t nwg, uint64_t out[12])
{
    if (!s || !s->s_cnt.p) return fail(PSK_EINVAL, "no partition scratch yet");
    PSK_USE_DEVICE(s->device);
    HIP_TRY(hipDeviceSynchronize());
    char *p = (char *)s->s_cnt.p + (size_t)nbuckets * nwg * 4;
    HIP_TRY(hipMemcpy(out, p, 96, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(p, 0, 96));
    return PSK_OK;
}

extern "C" int psk_sketch_set_option(psk_sketch *s, const char *name, int64_t value)
{
    if (!s || !name) return fail(PSK_EINVAL, "NULL argument");
    if (!strcmp(name, "table_private")) {
        s->table_private = value != 0;
        if (s->table_private) s->shadow.exposed = false;  // (from here on the holder announces its writes)
        else if (!s->owns_table) s->shadow.exposed = true;
        ++s->table_version;
        return PSK_OK;
    }
    const int i = ho_index(name);
    if (i < 0) return fail(PSK_EINVAL, "%s is not a per-sketch option", name);
    if (i == HO_PART_MIN_KEYS && value != kHoUnset && value < 1) value = 1;
    s->opt[i] = value;  // (kHoUnset = INT64_MIN: follow the process default again)
    return PSK_OK;
}

extern "C" int psk_sketch_get_option(psk_sketch *s, const char *name, int64_t *value)
{
    if (!s || !name || !value) return fail(PSK_EINVAL, "NULL argument");
    if (!strcmp(name, "table_private")) {
        *value = s->table_private ? 1 : 0;
        return PSK_OK;
    }
    if (!strcmp(name, "window_pending_batches")) {  // read-only: batches the update window still holds (borrowed ones among them must stay as they are)
        *value = s->win.n ? (int64_t)s->win.batches.size() : 0;
        return PSK_OK;
    }
    const int i = ho_index(name);
    if (i < 0) return fail(PSK_EINVAL, "%s is not a per-sketch option", name);
    *value = s->opt[i] != kHoUnset ? s->opt[i] : __atomic_load_n(&d_opt[i], __ATOMIC_RELAXED);
    return PSK_OK;
}

extern "C" int psk_get_option(const char *name, int64_t *value)
{
    if (!name || !value) return fail(PSK_EINVAL, "NULL argument");
    if (const int i = ho_index(name); i >= 0) {
        *value = __atomic_load_n(&d_opt[i], __ATOMIC_RELAXED);
        return PSK_OK;
    }
    const OptDesc *d = opt_find(name);
    if (!d) return fail(PSK_EINVAL, "unknown option %s", name);
    *value = *d->var;
    return PSK_OK;
}

// ------------------------------------------------------------- BloomFilter
extern "C" int psk_bloom_add(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                             uint32_t key_len, int where, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_BLOOM);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    hipStream_t st = (hipStream_t)stream;
    Batch b;
    PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
    bool done = false;
    if (!s->pend.active) PSK_TRY(bloom_add_partitioned(s, b, st, &done));  // (a pending split lookup owns the bucket buffer)
    if (done) return finish(where, nullptr, st);
    Mailbox mb;  // (an update returns nothing, but a PSK_HOST call ends when the kernel has read the caller's keys: the same mailbox says so)
    PSK_TRY(mailbox_arm(s, where, n, true, &mb));
    KeysInline64 ik;
    PSK_TRY(with_source_one(b, inline_key(layout, data, n, key_len, mb, &ik), [&](auto src) {
        if (s->pow2) return launch_apply(src, BloomAdd<true>{(uint32_t *)s->table, s->md, s->k}, n, st, &mb);
        return launch_apply(src, BloomAdd<false>{(uint32_t *)s->table, s->md, s->k}, n, st, &mb);
    }));
    return finish(where, nullptr, st, &mb);
}

extern "C" int psk_bloom_check(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                               uint32_t key_len, int where, uint8_t *out, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_BLOOM);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    if (n && !out) return fail(PSK_EINVAL, "out is NULL");
    hipStream_t st = (hipStream_t)stream;
    Batch b;
    PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
    OutBuf o;
    PSK_TRY(stage_out(s->s_out, out, n, where, &o));
    {
        bool done = false;
        if (!s->pend.active) PSK_TRY(bloom_check_partitioned(s, b, (uint8_t *)o.dev, st, &done));
        if (done) return finish(where, &o, st);
    }
    Mailbox mb;
    PSK_TRY(mailbox_arm(s, where, n, o.is_pinned, &mb));
    KeysInline64 ik;
    PSK_TRY(with_source_one(b, inline_key(layout, data, n, key_len, mb, &ik), [&](auto src) {
        if (s->pow2) return launch_apply(src, BloomCheck<true>{(const uint32_t *)s->table, s->md, s->k, (uint8_t *)o.dev}, n, st, &mb);
        return launch_apply(src, BloomCheck<false>{(const uint32_t *)s->table, s->md, s->k, (uint8_t *)o.dev}, n, st, &mb);
    }));
    return finish(where, &o, st, &mb);
}

extern "C" int psk_bloom_indices(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                                uint32_t key_len, int where, uint32_t *out_idx_dev, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_BLOOM);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    if (n && !out_idx_dev) return fail(PSK_EINVAL, "out_idx_dev is NULL");
    if (s->m > (1ULL << 32)) return fail(PSK_EINVAL, "bit indices are 32-bit: m must be <= 2^32");
    hipStream_t st = (hipStream_t)stream;
    Batch b;
    PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
    PSK_TRY(with_source(b, [&](auto src) {
        if (s->pow2) return launch_apply(src, BloomIndexOut<true>{out_idx_dev, s->md, s->k}, n, st);
        return launch_apply(src, BloomIndexOut<false>{out_idx_dev, s->md, s->k}, n, st);
    }));
    return finish(where, nullptr, st);
}

// Split lookup (see include/psk.h): begin = hash + partition (never reads the table), finish = probe.
extern "C" int psk_bloom_check_begin(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                                     uint32_t key_len, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_BLOOM);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    if (s->pend.active) return fail(PSK_EINVAL, "a split lookup is already pending on this handle");
    Batch b;
    b.layout = layout;
    b.data = data;
    b.offs = offsets;
    b.n = n;
    b.key_len = key_len;
    if (n && !data) return fail(PSK_EINVAL, "keys are NULL");
    return bloom_check_begin_partitioned(s, b, (hipStream_t)stream);
}

extern "C" int psk_bloom_check_finish(psk_sketch *s, uint8_t *out_dev, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_BLOOM);
    if (!s->pend.active) return fail(PSK_EINVAL, "no split lookup pending on this handle");
    s->pend.active = false;
    const Batch b = s->pend.b;
    if (b.n && !out_dev) return fail(PSK_EINVAL, "out is NULL");
    hipStream_t st = (hipStream_t)stream;
    bool redo = false;
    PSK_TRY(bloom_check_finish_partitioned(s, out_dev, st, &redo));
    if (!s->pend.scattered) {  // batch / table not eligible for the partitioned path: plain direct lookup now
        return with_source(b, [&](auto src) {
            if (s->pow2) return launch_apply(src, BloomCheck<true>{(const uint32_t *)s->table, s->md, s->k, out_dev}, b.n, st);
            return launch_apply(src, BloomCheck<false>{(const uint32_t *)s->table, s->md, s->k, out_dev}, b.n, st);
        });
    }
    if (redo) {  // exact redo of the first round, taken on the device only if a segment overflowed during begin
        const uint64_t cnt0 = b.n < s->pend.round_keys ? b.n : s->pend.round_keys;
        const uint32_t *flag = (const uint32_t *)s->s_flag.p;
        PSK_TRY(with_source(sub_batch(b, 0, cnt0), [&](auto src) {
            using Src = decltype(src);
            if (s->pow2) {
                using Op = BloomCheck<true>;
                hipLaunchKernelGGL((k_apply_if<Src, Op>), dim3(grid_for(cnt0)), dim3(kBlock), 0, st, flag, src,
                                   Op{(const uint32_t *)s->table, s->md, s->k, out_dev}, cnt0);
            } else {
                using Op = BloomCheck<false>;
                hipLaunchKernelGGL((k_apply_if<Src, Op>), dim3(grid_for(cnt0)), dim3(kBlock), 0, st, flag, src,
                                   Op{(const uint32_t *)s->table, s->md, s->k, out_dev}, cnt0);
            }
            HIP_TRY(hipGetLastError());
            return (int)PSK_OK;
        }));
    }
    return PSK_OK;
}

// Large batches of psk_bloom_check_bits: the partitioned lookup answers a byte per key (whichever scheme the batch calls for: tile flags,
// keyed probes, return trip, lazy gathers), and this one streaming pass turns the bytes into the ballot words and counts the hits -- the
// direct kernel pays k 64-byte gathers per key (~9 G keys/s at k = 7 against the partitioned lookups' 30-50).  One atomic per workgroup.
static __global__ __launch_bounds__(kBlock) void k_pack_answer_bits(const uint8_t *ans, uint64_t n, unsigned long long *out_bits, unsigned long long *hits)
{
    __shared__ unsigned long long wsum[kBlock / 64];
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t nround = (n + 63) & ~63ULL;  // wave-uniform trip count
    unsigned long long my_hits = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nround; i += stride) {
        const unsigned long long bal = __ballot(i < n && ans[i < n ? i : 0] != 0);
        if ((threadIdx.x & 63) == 0) {
            out_bits[i >> 6] = bal;
            my_hits += (unsigned long long)__popcll(bal);
        }
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = my_hits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kBlock / 64; ++w) t += wsum[w];
        if (t) atomicAdd(hits, t);
    }
}

extern "C" int psk_bloom_check_bits(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                                    uint32_t key_len, int where, uint64_t *out_bits, uint64_t *hits, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_BLOOM);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    if (n && (!out_bits || !hits)) return fail(PSK_EINVAL, "out_bits / hits is NULL");
    hipStream_t st = (hipStream_t)stream;
    Batch b;
    PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
    const uint64_t nwords = (n + 63) / 64;
    OutBuf o;
    PSK_TRY(stage_out(s->s_out, out_bits, nwords * 8, where, &o));
    unsigned long long *hits_dev = (unsigned long long *)hits;
    const bool big = n && !s->pend.active && part_wanted(n, s->k, 4);
    if (where == PSK_HOST || big) PSK_TRY(ensure(s->s_aux, 16 + (big ? n : 0)));  // hits (staged for host callers) | a byte per key
    if (where == PSK_HOST) {
        hits_dev = (unsigned long long *)s->s_aux.p;
        HIP_TRY(hipMemcpyAsync(hits_dev, hits, 8, hipMemcpyHostToDevice, st));
    }
    bool packed = false;
    if (big) {
        uint8_t *ans = (uint8_t *)s->s_aux.p + 16;
        PSK_TRY(bloom_check_partitioned(s, b, ans, st, &packed));
        if (packed) {
            const uint64_t blocks = (n + kBlock - 1) / kBlock;
            hipLaunchKernelGGL(k_pack_answer_bits, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(kBlock), 0, st, (const uint8_t *)ans, n,
                               (unsigned long long *)o.dev, hits_dev);
            HIP_TRY(hipGetLastError());
        }
    }
    if (n && !packed) {
        PSK_TRY(with_source(b, [&](auto src) {
            using Src = decltype(src);
            if (s->pow2)
                hipLaunchKernelGGL((k_bloom_check_bits<Src, true>), dim3(grid_for(n)), dim3(kBlock), 0, st, src,
                                   (const uint32_t *)s->table, s->md, s->k, n, (unsigned long long *)o.dev, hits_dev);
            else
                hipLaunchKernelGGL((k_bloom_check_bits<Src, false>), dim3(grid_for(n)), dim3(kBlock), 0, st, src,
                                   (const uint32_t *)s->table, s->md, s->k, n, (unsigned long long *)o.dev, hits_dev);
            HIP_TRY(hipGetLastError());
            return (int)PSK_OK;
        }));
    }
    if (where == PSK_HOST) HIP_TRY(hipMemcpyAsync(hits, hits_dev, 8, hipMemcpyDeviceToHost, st));
    return finish(where, &o, st);
}

// ------------------------------------------------------ counters / weights
// grow_bound = false: the batch only lowers counters (CBF removes) -- the wrap-free bound on |counter| stays as it is
template <class W>
static int account_weights(psk_sketch *s, const W *w_dev, uint64_t n, int which, long long bound_mult, hipStream_t st, bool grow_bound = true)
{
    if (n == 0) return PSK_OK;
    if (w_dev) {
        HIP_TRY(hipMemsetAsync(s->ctr + 6, 0, sizeof(long long), st));  // per-batch sum|w| (partitioned path wrap check)
        hipLaunchKernelGGL((k_weight_sum<W>), dim3(grid_for(n) > 256 ? 256 : grid_for(n)), dim3(kBlock), 0, st, w_dev, n,
                           s->ctr, which, bound_mult, (int)grow_bound);
    } else {
        hipLaunchKernelGGL(k_ctr_add, dim3(1), dim3(1), 0, st, s->ctr, which, (long long)n, grow_bound ? (long long)n * bound_mult : 0LL);
    }
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

// Accounting of a weighted batch, fused into pass 1 when the partitioned path takes it (PayWeight::tally): post the request,
// try the partitioned launcher, settle what it did not take over with the stand-alone pass over the weights.
template <class W>
static int post_acct(psk_sketch *s, const W *w_dev, uint64_t n, int which, long long bound_mult, hipStream_t st, bool grow_bound, bool weights_signed)
{
    s->acct.pending = false;
    if (!w_dev || n == 0) return account_weights(s, w_dev, n, which, bound_mult, st, grow_bound);  // unit weights: a one-thread kernel
    s->acct.pending = true;
    s->acct.which = which;
    s->acct.bound_mult = bound_mult;
    s->acct.grow_bound = grow_bound;
    s->acct.weights_signed = weights_signed;
    s->acct.weights01 = false;
    return PSK_OK;
}

template <class W>
static int settle_acct(psk_sketch *s, const W *w_dev, uint64_t n, hipStream_t st)
{
    if (!s->acct.pending) return PSK_OK;
    s->acct.pending = false;
    return account_weights(s, w_dev, n, s->acct.which, s->acct.bound_mult, st, s->acct.grow_bound);
}

// ----------------------------------------------------- CountingBloomFilter
int64_t g_combine_keys = 1 << 26;  // keys per write-combining list (psk_set_option "combine_keys"): 1 GiB of 16-byte keys per list

// one unordered CBF update over a DEVICE-resident batch: add (countingbloom.py:135-155) or the unchecked decrement
static int cbf_apply_device(psk_sketch *s, const Batch &b, const uint32_t *w, bool remove, hipStream_t st)
{
    if (b.n == 0) return PSK_OK;
    PSK_TRY(post_acct(s, w, b.n, remove ? PSK_CTR_REMOVED : PSK_CTR_ADDED, (long long)s->k, st, !remove, false));
    unsigned long long *sat = (unsigned long long *)(s->ctr + PSK_CTR_SATURATED);
    bool done = false;
    PSK_TRY(remove ? cbf_remove_partitioned(s, b, w, st, &done, 0, nullptr) : cbf_add_partitioned(s, b, w, st, &done));
    PSK_TRY(settle_acct(s, w, b.n, st));
    if (done) return PSK_OK;
    return with_source(b, [&](auto src) {
        if (remove) {
            if (s->pow2) return launch_apply(src, CbfSub<true>{(uint32_t *)s->table, s->md, s->k, w, sat - 1}, b.n, st);
            return launch_apply(src, CbfSub<false>{(uint32_t *)s->table, s->md, s->k, w, sat - 1}, b.n, st);
        }
        if (s->pow2) return launch_apply(src, CbfAdd<true>{(uint32_t *)s->table, s->md, s->k, w, s->ctr, sat, false}, b.n, st);
        return launch_apply(src, CbfAdd<false>{(uint32_t *)s->table, s->md, s->k, w, s->ctr, sat, false}, b.n, st);
    });
}

// ---- write-combined updates as scattered probes (psk_sketch::scat, psk_nibble.hpp)
int64_t g_fused_flush = 0;             // flush of both write-combined key lists as two pass 1s + ONE fold (adds, then decrements, per slice): measured
                                       // SLOWER on BASELINE cfg 4 (4.50 vs 3.80 ms per step: one workgroup per slice streams both lists and folds twice
                                       // back to back, nothing overlaps) -- off; option "combine_fused_flush"
int64_t g_combine_scatter = 0;         // psk_cbf_update_combined: 1 = unit-weight batches wait as scattered probes instead of key lists (see there)
__thread int64_t g_auto_combine = 1;            // psk_cbf_add: small unit-weight batches into big tables wait as scattered probes (adds commute: exact)
int64_t g_auto_combine_keys = 1 << 24; // keys per list in that mode (~0.8 GB of segments for k = 7, allocated on first use)

// a flush (or drop) on another stream than the last append must not overtake it
// (the event is recorded only when a second stream shows up: one per append put a barrier packet -- ~5 us of dispatch bubble --
// behind every 1 M-key batch of BASELINE cfg 4, 0.45 ms per step)
static int comb_order(psk_sketch *s, hipStream_t st)
{
    if (!s->scat.appended || st == s->scat.last) return PSK_OK;
    if (!s->scat.ev) HIP_TRY(hipEventCreateWithFlags(&s->scat.ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(s->scat.ev, s->scat.last));  // the tail of the stream that appended last: behind all of its appends
    HIP_TRY(hipStreamWaitEvent(st, s->scat.ev, 0));
    s->scat.last = st;  // (what follows on `st` is ordered behind it)
    return PSK_OK;
}

static int comb_appended(psk_sketch *s, hipStream_t st)
{
    s->scat.last = st;
    s->scat.appended = true;
    return PSK_OK;
}

static int scat_zero(psk_sketch *s, bool add, bool rem, hipStream_t st)
{
    const uint64_t nseg = (uint64_t)s->scat.g.nbuckets * s->scat.g.nwg;
    uint32_t *a = add ? (uint32_t *)s->scat.add.cnt.p : nullptr, *b = rem ? (uint32_t *)s->scat.rem.cnt.p : nullptr;
    if (!a && !b) return PSK_OK;
    hipLaunchKernelGGL(k_zero_u32, dim3(256), dim3(256), 0, st, a ? a : b, nseg, (a && b) ? b : nullptr, (a && b) ? nseg : 0ULL);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

static int scat_drop(psk_sketch *s, hipStream_t st)
{
    if (!s->scat.ready || (s->scat.add.n == 0 && s->scat.rem.n == 0)) return PSK_OK;
    PSK_TRY(comb_order(s, st));
    PSK_TRY(scat_zero(s, s->scat.add.n != 0, s->scat.rem.n != 0, st));
    s->scat.add.n = s->scat.rem.n = 0;
    return PSK_OK;
}

// apply the scattered lists: adds, then decrements (a remove whose add waits in the same window must find it applied).
// Enough probes: one pass over the table (k_nib_apply; both lists in ONE launch when both are due); few: a drain with atomics.
static int scat_flush(psk_sketch *s, hipStream_t st)
{
    if (!s->scat.ready || (s->scat.add.n == 0 && s->scat.rem.n == 0)) return PSK_OK;
    PSK_TRY(comb_order(s, st));
    const uint64_t na = s->scat.add.n, nr = s->scat.rem.n;
    s->scat.add.n = s->scat.rem.n = 0;  // (cleared first: a failure must not re-apply the lists on the next call)
    PartGeom g = s->scat.g;
    const uint64_t per_seg = (uint64_t)g.nbuckets * g.nwg * 6;
    g.dense = ((na > nr ? na : nr) * s->k / per_seg) < (uint64_t)g_part_dense_groups ? 1u : 0u;
    const uint32_t lgp = nib_update_lgparts(g);
    const size_t lds = (size_t)1 << (g.shift - 1 - lgp);
    unsigned long long *sat = (unsigned long long *)(s->ctr + PSK_CTR_SATURATED);
    const bool pass_a = na * s->k >= s->m / 8, pass_r = nr * s->k >= s->m / 8;
    auto launch = [&](auto kern, const psk_sketch::ScatList *la, const psk_sketch::ScatList *lb, uint32_t direct) {
        PSK_TRY(set_dyn_lds(kern, lds));
        hipLaunchKernelGGL(kern, dim3(g.nbuckets << lgp), dim3(kApplyThreads), lds, st, (uint32_t *)s->table, s->m, g, (const uint32_t *)la->cnt.p, (const uint4 *)la->part.p,
                           (const uint32_t *)(lb ? lb->cnt.p : nullptr), (const uint4 *)(lb ? lb->part.p : nullptr), sat, direct | (lgp << 8), (uint32_t *)nullptr, g);
        HIP_TRY(hipGetLastError());
        return (int)PSK_OK;
    };
    const bool blocks = g_nib_update_layout != 0;
    if (na && nr && pass_a && pass_r) {
        PSK_TRY(blocks ? launch(k_nib_apply<2, true>, &s->scat.add, &s->scat.rem, 0u) : launch(k_nib_apply<2, false>, &s->scat.add, &s->scat.rem, 0u));
    } else {
        if (na) PSK_TRY(blocks ? launch(k_nib_apply<0, true>, &s->scat.add, nullptr, pass_a ? 0u : 1u) : launch(k_nib_apply<0, false>, &s->scat.add, nullptr, pass_a ? 0u : 1u));
        if (nr) PSK_TRY(blocks ? launch(k_nib_apply<1, true>, &s->scat.rem, nullptr, pass_r ? 0u : 1u) : launch(k_nib_apply<1, false>, &s->scat.rem, nullptr, pass_r ? 0u : 1u));
    }
    return scat_zero(s, na != 0, nr != 0, st);
}

// Hand a unit-weight batch over as scattered probes.  cap: keys per list; *done = false: not eligible (nothing was launched / changed).
static int scat_append(psk_sketch *s, const Batch &b, bool neg, uint64_t cap, hipStream_t st, bool *done)
{
    *done = false;
    {   // a list must stay within what one fold's 4-bit deltas hold (nib_load_ok): ~2.5 probes per counter
        const uint64_t by_table = s->m * 5 / (2 * (uint64_t)(s->k ? s->k : 1));
        if (cap > by_table) cap = by_table;
    }
    if (s->kind != PSK_KIND_CBF || g_update_nibble == 0 || b.n == 0 || b.n > cap || s->k > 32) return PSK_OK;
    if (b.layout == PSK_KEYS_HASHES && b.key_len < s->k) return PSK_OK;
    if (!s->scat.ready || s->scat.cap != cap) {
        PartGeom g;
        if (!scat_geometry(s, cap, &g)) return PSK_OK;
        PSK_TRY(scat_flush(s, st));  // (a list sized for another capacity)
        s->scat.g = g;
        s->scat.cap = cap;
        s->scat.ready = true;
    }
    psk_sketch::ScatList &l = neg ? s->scat.rem : s->scat.add;
    if (l.n + b.n > cap) PSK_TRY(flush_combined(s, st));
    const PartGeom &g = s->scat.g;
    const uint64_t part_bytes = (uint64_t)g.nbuckets * g.nwg * g.segcap * 16 + 256, cnt_bytes = (uint64_t)g.nbuckets * g.nwg * 4 + 128;
    if (neg) return PSK_OK;  // (decrements stay out of the persistent segments: an overflowing segment would apply them ahead of the window's adds)
    if (l.part.cap < part_bytes || l.cnt.cap < cnt_bytes) {  // first use (or released): allocate, counts start at zero
        if (g_scratch_budget > 0 && (int64_t)(part_bytes + cnt_bytes) > g_scratch_budget) return PSK_OK;  // (not taken: the direct path serves)
        if (ensure(l.part, part_bytes) != PSK_OK || ensure(l.cnt, cnt_bytes) != PSK_OK) return PSK_OK;    // out of memory costs the shortcut, not the add
        HIP_TRY(hipMemsetAsync(l.cnt.p, 0, cnt_bytes, st));
        l.n = 0;
    }
    PSK_TRY(comb_order(s, st));  // (appends are ordered among themselves too: two streams must not race on the cursors)
    bool appended = false;
    PSK_TRY(cbf_scat_append(s, b, neg ? 1 : 0, st, &appended));
    if (!appended) return PSK_OK;  // layout without a partitioned instantiation
    l.n += b.n;
    PSK_TRY(comb_appended(s, st));
    *done = true;
    return PSK_OK;
}

// the borrowed batches of one list: their pointer / prefix tables go to the device, then ONE pass 1 over all of them + one fold
static int borrowed_flush(psk_sketch *s, psk_sketch::BorrowList &bl, bool remove, hipStream_t st)
{
    const uint64_t n = bl.n();
    if (n == 0) return PSK_OK;
    const uint32_t nb = (uint32_t)bl.base.size();
    PSK_TRY(ensure(s->s_brw, (uint64_t)(2 * nb + 2) * 8));
    const void **base_dev = (const void **)s->s_brw.p;
    uint64_t *start_dev = (uint64_t *)s->s_brw.p + nb;
    HIP_TRY(hipMemcpyAsync(base_dev, bl.base.data(), (size_t)nb * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(start_dev, bl.start.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // (the host vectors are reused from here on; a flush is milliseconds of device work anyway)
    std::vector<const void *> bases;
    std::vector<uint64_t> starts;
    bases.swap(bl.base);   // (cleared first: a failure must not re-apply the list on the next call)
    starts.swap(bl.start);
    bl.clear();
    PSK_TRY(account_weights(s, (const uint32_t *)nullptr, n, remove ? PSK_CTR_REMOVED : PSK_CTR_ADDED, (long long)s->k, st, !remove));
    bool done = false;
    PSK_TRY(cbf_unit_multi_partitioned(s, (const void *const *)base_dev, start_dev, nb, n, remove ? 1 : 0, st, &done));
    if (done) return PSK_OK;
    for (uint32_t j = 0; j < nb; ++j) {  // table not eligible after all (option changed meanwhile): batch by batch through the general path
        Batch b{PSK_KEYS_FIXED, bases[j], nullptr, starts[j + 1] - starts[j], 16};
        unsigned long long *sat = (unsigned long long *)(s->ctr + PSK_CTR_SATURATED);
        PSK_TRY(with_source(b, [&](auto src) {
            if (remove) {
                if (s->pow2) return launch_apply(src, CbfSub<true>{(uint32_t *)s->table, s->md, s->k, nullptr, sat - 1}, b.n, st);
                return launch_apply(src, CbfSub<false>{(uint32_t *)s->table, s->md, s->k, nullptr, sat - 1}, b.n, st);
            }
            if (s->pow2) return launch_apply(src, CbfAdd<true>{(uint32_t *)s->table, s->md, s->k, nullptr, s->ctr, sat, false}, b.n, st);
            return launch_apply(src, CbfAdd<false>{(uint32_t *)s->table, s->md, s->k, nullptr, s->ctr, sat, false}, b.n, st);
        }));
    }
    return PSK_OK;
}

// ---- update windows (psk_window.hpp): small unit-weight add / remove batches of 16-byte keys into big tables wait, in arrival order,
// as key copies; win_flush applies them in one pass over the table, proving the removes while it folds -- or replays them one by one
__thread int64_t g_window = 1;                // option "update_window"
__thread int64_t g_window_keys = 1 << 27;     // option "update_window_keys": most keys a window holds (16 bytes each); also cells / 2 and the scratch budget
int64_t g_window_folds = 0, g_window_replays = 0;  // windows applied by the fold / replayed batch by batch (tests, bench)
int64_t g_window_force_fail = 0;     // tests: pretend the proof failed (exercises undo + replay on a well-formed stream)
constexpr size_t kWinMaxBatches = 4096;

static uint64_t win_capacity(const psk_sketch *s)
{
    uint64_t cap = s->m / 2;  // (BASELINE cfg 4's whole 74.5 M-operation step is one window of the 2^28-counter table)
    if (cap < (1u << 20)) cap = 1u << 20;
    if (g_window_keys > 0 && cap > (uint64_t)g_window_keys) cap = (uint64_t)g_window_keys;
    cap = cap_round_by_budget(cap, 16.0 + (double)s->k * (16.0 / 6.0) * 1.5);  // key copy + probe groups with their padding
    return cap;
}

static bool win_eligible(const psk_sketch *s, int layout, const void *data, uint32_t key_len, const uint32_t *weights, uint64_t n)
{
    PartGeom g;
    return s->kind == PSK_KIND_CBF && g_window != 0 && g_update_nibble != 0 && !weights && layout == PSK_KEYS_FIXED && key_len == 16 && data && n != 0 &&
           (int64_t)n >= g_part_min_keys && n * (uint64_t)s->k < s->m / 8 && s->k <= 32 && n <= win_capacity(s) && nib_geometry(s->m, true, &g);
}

static int cbf_remove_device(psk_sketch *s, const Batch &b, const uint32_t *w, hipStream_t st);

// every waiting batch through the per-batch paths, in arrival order (windows too small for a pass over the table, tables / options that
// rule the fold out, and the windows whose proof failed)
static int win_replay(psk_sketch *s, const void *keys, const std::vector<psk_sketch::WinBatch> &batches, hipStream_t st)
{
    ++s->win.replays;
    ++g_window_replays;
    for (const auto &wb : batches) {
        const Batch b{PSK_KEYS_FIXED, wb.ext ? wb.ext : (const void *)((const uint8_t *)keys + wb.start * 16), nullptr, wb.n, 16};
        if (wb.remove) PSK_TRY(cbf_remove_device(s, b, nullptr, st));
        else PSK_TRY(cbf_apply_device(s, b, nullptr, false, st));
    }
    return PSK_OK;
}

static int win_flush(psk_sketch *s, hipStream_t st)
{
    if (s->win.n == 0) return PSK_OK;
    PSK_TRY(comb_order(s, st));
    std::vector<psk_sketch::WinBatch> batches;
    batches.swap(s->win.batches);  // (cleared first: a failure must not re-apply the window on the next call)
    const uint64_t n = s->win.n;
    s->win.n = 0;
    s->win.copied = 0;
    const void *keys = s->win.keys.p;
    // runs of same-type batches are the fold's phases
    std::vector<WinBatchHost> wbh;
    wbh.reserve(batches.size());
    uint64_t n_add = 0, n_rem = 0;
    size_t phases = 0;
    bool borrowed = false;
    for (const auto &wb : batches) {
        phases += wbh.empty() || wbh.back().remove != wb.remove;
        wbh.push_back(WinBatchHost{wb.ext ? wb.ext : (const void *)((const uint8_t *)keys + wb.start * 16), wb.n, wb.remove});
        borrowed = borrowed || wb.ext != nullptr;
        (wb.remove ? n_rem : n_add) += wb.n;
    }
    // One phase of keys that lie end to end (copies in the list, or one borrowed batch): a plain batch (the partitioned add / the validated
    // remove take it as a whole).  Too few probes for a pass over the table, or a recent window whose proof failed: batch by batch.
    if (phases == 1 && (!borrowed || batches.size() == 1)) {
        const Batch b{PSK_KEYS_FIXED, wbh[0].keys, nullptr, n, 16};
        return wbh[0].remove ? cbf_remove_device(s, b, nullptr, st) : cbf_apply_device(s, b, nullptr, false, st);
    }
    const bool worth = n * (uint64_t)s->k >= s->m / 8 && phases <= (size_t)kWinMaxPhases;
    if (worth && s->win.backoff == 0) {
        bool launched = false, ok = false;
        PSK_TRY(cbf_window_fold(s, wbh.data(), (uint32_t)wbh.size(), st, &launched, &ok));
        if (launched && ok) {
            ++s->win.folds;
            ++g_window_folds;
            PSK_TRY(account_weights(s, (const uint32_t *)nullptr, n_add, PSK_CTR_ADDED, (long long)s->k, st, true));
            return account_weights(s, (const uint32_t *)nullptr, n_rem, PSK_CTR_REMOVED, (long long)s->k, st, false);
        }
        if (launched) s->win.backoff = 8;  // this stream removes keys that are not there: stop paying for fold + undo for a while
    } else if (s->win.backoff) {
        --s->win.backoff;
    }
    return win_replay(s, keys, batches, st);
}

// Room for `want` keys in the window's key list (16 bytes each).  The list grows in steps -- 2^22 keys (64 MiB) first, then doubling up to
// the window's capacity, the waiting keys copied over -- instead of cap x 16 bytes (2 GiB for a 2^28-counter table) on the first small
// batch.  *ok = false: the memory is not there (the HIP error is cleared): the caller takes the paths that need no list.
static int win_reserve(psk_sketch *s, uint64_t want, uint64_t cap, hipStream_t st, bool *ok)
{
    *ok = true;
    if (want * 16 <= s->win.keys.cap) return PSK_OK;
    uint64_t keys = s->win.keys.cap / 16 ? s->win.keys.cap / 16 : (1ULL << 22);
    while (keys < want) keys *= 2;
    if (keys > cap) keys = cap;
    void *p = nullptr;
    if (hipMalloc(&p, keys * 16) != hipSuccess) {
        (void)hipGetLastError();
        *ok = false;
        return PSK_OK;
    }
    if (s->win.copied) {
        // (the new list is ours until it is stored below: every failure on the way frees it)
        auto moved = [&]() -> int {
            PSK_TRY(comb_order(s, st));
            HIP_TRY(hipMemcpyAsync(p, s->win.keys.p, s->win.copied * 16, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));  // (the old list is freed below)
            return PSK_OK;
        };
        const int rc = moved();
        if (rc != PSK_OK) {
            (void)hipFree(p);
            return rc;
        }
    }
    if (s->win.keys.p) {
        const hipError_t e = hipFree(s->win.keys.p);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return fail(PSK_EHIP, "hipFree of the window's key list failed: %s", hipGetErrorString(e));
        }
    }
    s->win.keys.p = p;
    s->win.keys.cap = keys * 16;
    return PSK_OK;
}

// hand a batch over to the window (eligible: win_eligible); host batches are copied straight from the caller's buffer, PSK_DEVICE batches
// device to device, PSK_DEVICE_BORROWED ones stay where they are (the caller keeps them unchanged until the window has been applied:
// psk_flush_combined, any entry point that reads the table, or psk_sketch_get_option "window_pending_batches" back at 0).
// *taken = false: no memory for the key list -- nothing was appended, what waited has been applied, the caller applies this batch itself.
static int win_append(psk_sketch *s, const void *data, uint64_t n, bool remove, int where, hipStream_t st, bool *taken)
{
    *taken = true;
    const uint64_t cap = win_capacity(s);
    if (s->win.cap != cap && s->win.n) PSK_TRY(win_flush(s, st));
    if (s->win.n + n > cap || s->win.batches.size() >= kWinMaxBatches) PSK_TRY(win_flush(s, st));
    if (s->win.n && !s->win.batches.empty() && s->win.batches.back().remove != (remove ? 1u : 0u)) {
        size_t phases = 1;  // (a new phase: the fold holds at most kWinMaxPhases of them)
        for (size_t i = 1; i < s->win.batches.size(); ++i) phases += s->win.batches[i].remove != s->win.batches[i - 1].remove;
        if (phases >= (size_t)kWinMaxPhases) PSK_TRY(win_flush(s, st));
    }
    s->win.cap = cap;
    if (where == PSK_DEVICE_BORROWED) {  // (16-byte aligned: win_eligible's caller checked)
        s->win.batches.push_back(psk_sketch::WinBatch{0, n, remove ? 1u : 0u, data});
        s->win.n += n;
        return comb_appended(s, st);  // (a flush on another stream waits for this one: the keys may still be in the making on it)
    }
    bool room = false;
    PSK_TRY(win_reserve(s, s->win.copied + n, cap, st, &room));
    if (!room && s->win.n) {  // what waits fits what there is: apply it, then this batch may fit too
        PSK_TRY(win_flush(s, st));
        room = n * 16 <= s->win.keys.cap;
    }
    if (!room) {
        *taken = false;
        return PSK_OK;
    }
    PSK_TRY(comb_order(s, st));
    // (round 4: an own copy kernel with nontemporal loads / stores measured slower than the runtime's blit: 3.60 vs 3.55 ms per cfg-4 step)
    HIP_TRY(hipMemcpyAsync((uint8_t *)s->win.keys.p + s->win.copied * 16, data, n * 16, where == PSK_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    s->win.batches.push_back(psk_sketch::WinBatch{s->win.copied, n, remove ? 1u : 0u, nullptr});
    s->win.copied += n;
    s->win.n += n;
    PSK_TRY(comb_appended(s, st));
    if (where == PSK_HOST) HIP_TRY(hipStreamSynchronize(st));  // the caller may reuse its buffer on return
    return PSK_OK;
}

int flush_combined(psk_sketch *s, hipStream_t st)
{
    if (s->kind != PSK_KIND_CBF) return PSK_OK;
    if (s->win.n) {  // (the window holds what arrived AFTER anything the older mechanisms below hold: see win_append's callers)
        ++s->table_version;
        if (s->comb.add.n || s->comb.rem.n || s->comb.badd.n() || s->comb.brem.n() || (s->scat.ready && (s->scat.add.n || s->scat.rem.n))) {
            std::vector<psk_sketch::WinBatch> keep;
            keep.swap(s->win.batches);
            const uint64_t wn = s->win.n, wc = s->win.copied;
            s->win.n = 0;
            PSK_TRY(flush_combined(s, st));
            s->win.batches.swap(keep);
            s->win.n = wn;
            s->win.copied = wc;
        }
        PSK_TRY(win_flush(s, st));
    }
    const bool keys_pending = s->comb.add.n != 0 || s->comb.rem.n != 0 || s->comb.badd.n() != 0 || s->comb.brem.n() != 0;
    const bool scat_pending = s->scat.ready && (s->scat.add.n != 0 || s->scat.rem.n != 0);
    if (!keys_pending && !scat_pending) return PSK_OK;
    ++s->table_version;  // (also from the read-only entry points: what waited reaches the table now)
    PSK_TRY(comb_order(s, st));
    // adds first: a remove whose add waits in the same window must find it applied.  Two mechanisms may hold updates -- key lists
    // (weighted batches, tables below the nibble geometry) and scattered probes: all adds of both, then all removes of both.
    auto key_list = [&](int pass) {
        psk_sketch::PendList &l = pass == 0 ? s->comb.add : s->comb.rem;
        if (l.n == 0) return (int)PSK_OK;
        Batch b{PSK_KEYS_FIXED, l.keys.p, nullptr, l.n, s->comb.key_len};
        l.n = 0;  // (cleared first: a failure must not re-apply the list on the next call)
        const bool unit = l.unit;
        l.unit = true;
        return cbf_apply_device(s, b, unit ? nullptr : (const uint32_t *)l.w.p, pass == 1, st);
    };
    // Both key lists due, unit weights, big table: scatter both (two bucket buffers), then ONE fold -- adds, then decrements -- per slice
    // (k_nib_apply<2>): the slice a workgroup has just written is still on-die when it reads it back for the decrements.
    // (only when no OTHER mechanism holds adds of this window: they would have to land before these removes)
    if (s->comb.add.n && s->comb.rem.n && s->comb.add.unit && s->comb.rem.unit && g_fused_flush != 0 && s->comb.add.n * s->k >= s->m / 8 &&
        s->comb.rem.n * s->k >= s->m / 8 && s->comb.badd.n() == 0 && !(s->scat.ready && s->scat.add.n != 0)) {
        Batch ba{PSK_KEYS_FIXED, s->comb.add.keys.p, nullptr, s->comb.add.n, s->comb.key_len};
        Batch br{PSK_KEYS_FIXED, s->comb.rem.keys.p, nullptr, s->comb.rem.n, s->comb.key_len};
        PartGeom ga, gr;
        bool oka = false, okr = false;
        PSK_TRY(cbf_nib_scatter(s, ba, 0, 0, &ga, st, &oka));
        if (oka) PSK_TRY(cbf_nib_scatter(s, br, 1, 1, &gr, st, &okr));
        if (oka && okr) {
            const uint64_t na = s->comb.add.n, nr = s->comb.rem.n;
            s->comb.add.n = s->comb.rem.n = 0;
            PSK_TRY(account_weights(s, (const uint32_t *)nullptr, na, PSK_CTR_ADDED, (long long)s->k, st, true));
            PSK_TRY(account_weights(s, (const uint32_t *)nullptr, nr, PSK_CTR_REMOVED, (long long)s->k, st, false));
            const uint32_t lgp = nib_update_lgparts(ga);
            const size_t lds = (size_t)1 << (ga.shift - 1 - lgp);
            auto launch2 = [&](auto kern) {
                PSK_TRY(set_dyn_lds(kern, lds));
                hipLaunchKernelGGL(kern, dim3(ga.nbuckets << lgp), dim3(kApplyThreads), lds, st, (uint32_t *)s->table, s->m, ga, (const uint32_t *)s->s_cnt.p, (const uint4 *)s->s_part.p,
                                   (const uint32_t *)s->s_cnt2.p, (const uint4 *)s->s_part2.p, (unsigned long long *)(s->ctr + PSK_CTR_SATURATED), lgp << 8, (uint32_t *)nullptr, gr);
                HIP_TRY(hipGetLastError());
                return (int)PSK_OK;
            };
            PSK_TRY(g_nib_update_layout ? launch2(k_nib_apply<2, true>) : launch2(k_nib_apply<2, false>));
        }
        // (oka && !okr: the adds sit scattered in s_part but nothing was applied -- the lists are untouched, the general path below redoes them)
    }
    PSK_TRY(key_list(0));
    PSK_TRY(borrowed_flush(s, s->comb.badd, false, st));
    if (scat_pending && (s->comb.rem.n != 0 || s->comb.brem.n() != 0) && s->scat.add.n != 0) {  // key-list removes wait: the scattered adds must land before them
        const uint64_t nr = s->scat.rem.n;
        s->scat.rem.n = 0;
        PSK_TRY(scat_flush(s, st));
        s->scat.rem.n = nr;
    }
    PSK_TRY(key_list(1));
    PSK_TRY(borrowed_flush(s, s->comb.brem, true, st));
    return scat_flush(s, st);
}

extern "C" int psk_flush(psk_sketch *s, void *stream)
{
    CHECK_HANDLE_RO(s, -1);
    return flush_combined(s, (hipStream_t)stream);
}

extern "C" int psk_cbf_update_combined(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                                       uint32_t key_len, const uint32_t *weights, int remove, int where, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_CBF);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    if (where != PSK_HOST && where != PSK_DEVICE && where != PSK_DEVICE_BORROWED) return fail(PSK_EINVAL, "`where` must be PSK_HOST, PSK_DEVICE or PSK_DEVICE_BORROWED");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return PSK_OK;
    // Batches that wait in the update window (psk_cbf_add / psk_cbf_remove on this handle) arrived EARLIER than this one: they reach the
    // table first.  (flush_combined applies the lists below before the window: the window may only ever hold what came after them.)
    if (s->win.n) PSK_TRY(flush_combined(s, st));
    const uint64_t cap = g_combine_keys > 0 ? (uint64_t)g_combine_keys : 0;
    if (where == PSK_DEVICE_BORROWED) {
        // 16-byte unit-weight keys into a table with the nibble geometry: remember WHERE they are, nothing else.  The flush hashes all
        // borrowed batches of a list in one pass 1 where they lie (KeysFixed16Multi): no copy into a list, the keys are read once.
        PartGeom probe;
        const bool borrowable = layout == PSK_KEYS_FIXED && key_len == 16 && data && ((uintptr_t)data & 15) == 0 && !weights && cap && n <= cap &&
                                g_update_nibble != 0 && s->k <= 32 && scat_geometry(s, cap, &probe);
        if (borrowable) {
            psk_sketch::BorrowList &bl = remove ? s->comb.brem : s->comb.badd;
            const uint64_t by_table = s->m * 5 / (2 * (uint64_t)s->k);  // (what one fold's 4-bit deltas hold: nib_load_ok)
            const uint64_t capb = cap < by_table ? cap : by_table;
            if (bl.n() + n > capb || bl.base.size() >= 4096) PSK_TRY(flush_combined(s, st));
            bl.base.push_back(data);
            bl.start.push_back(bl.start.back() + n);
            return comb_appended(s, st);  // (a flush on another stream waits for this one: the keys may still be in the making on it)
        }
        where = PSK_DEVICE;  // anything else is copied as usual
    }
    // Scattered probes instead of key lists (option "combine_scatter", off): pass 1 per batch saves the key copy and the second read
    // of the keys, but measured on BASELINE cfg 4 (99 batches of 0.5-1 M keys) it costs more than it saves -- a 1 M-key pass 1 runs
    // two tiles per workgroup and pays its fixed costs (1024 cursors read and written per workgroup, pipeline fill) every time:
    // 45 us per batch against 7 us for the copy plus 21 us per 1 M keys of a 50 M-key pass 1; the step took 6.7 ms instead of 3.9.
    if (!weights && cap && g_combine_scatter != 0) {
        Batch b;
        PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
        bool taken = false;
        PSK_TRY(scat_append(s, b, remove != 0, cap, st, &taken));
        if (taken) {
            PSK_TRY(account_weights(s, (const uint32_t *)nullptr, n, remove ? PSK_CTR_REMOVED : PSK_CTR_ADDED, (long long)s->k, st, !remove));
            return finish(where, nullptr, st);
        }
    }
    const bool combinable = layout == PSK_KEYS_FIXED && key_len > 0 && data && n < cap;
    if (!combinable || (s->comb.key_len && s->comb.key_len != key_len) || (s->comb.cap && s->comb.cap != cap)) {
        PSK_TRY(flush_combined(s, st));
        if (!combinable) {  // other layouts, empty keys, batches as large as a list: applied at once (same semantics)
            Batch b;
            PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
            const uint32_t *w;
            PSK_TRY(stage_vec(s->s_w, weights, n, where, st, &w));
            PSK_TRY(cbf_apply_device(s, b, w, remove != 0, st));
            return finish(where, nullptr, st);
        }
    }
    psk_sketch::PendList &l = remove ? s->comb.rem : s->comb.add;
    if (l.n + n > cap) PSK_TRY(flush_combined(s, st));
    s->comb.key_len = key_len;
    s->comb.cap = cap;
    PSK_TRY(ensure(l.keys, cap * (uint64_t)key_len));  // full capacity at once: growing would drop the pending keys
    if (weights || !l.unit) PSK_TRY(ensure(l.w, cap * 4));  // the weight list only once a non-unit batch has arrived
    PSK_TRY(comb_order(s, st));
    const hipMemcpyKind kind = where == PSK_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    HIP_TRY(hipMemcpyAsync((uint8_t *)l.keys.p + l.n * (uint64_t)key_len, data, n * (uint64_t)key_len, kind, st));
    uint32_t *wdst = (uint32_t *)l.w.p + l.n;
    if (weights) {
        if (l.unit && l.n) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)l.w.p, 1, l.n, st));  // earlier unit batches get their 1s now
        HIP_TRY(hipMemcpyAsync(wdst, weights, n * 4, kind, st));
        l.unit = false;
    } else if (!l.unit) {
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)wdst, 1, n, st));
    }
    l.n += n;
    PSK_TRY(comb_appended(s, st));
    if (where == PSK_HOST) HIP_TRY(hipStreamSynchronize(st));  // the caller may reuse its buffers on return
    return PSK_OK;
}

extern "C" int psk_cbf_add(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                           uint32_t key_len, const uint32_t *weights, int where, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_CBF);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    hipStream_t st = (hipStream_t)stream;
    if (where != PSK_HOST && where != PSK_DEVICE && where != PSK_DEVICE_BORROWED) return fail(PSK_EINVAL, "`where` must be PSK_HOST, PSK_DEVICE or PSK_DEVICE_BORROWED");
    if (win_eligible(s, layout, data, key_len, weights, n)) {
        // (what the older write-combining mechanisms hold arrived earlier: it goes first)
        if (s->comb.add.n || s->comb.rem.n || s->comb.badd.n() || s->comb.brem.n() || (s->scat.ready && (s->scat.add.n || s->scat.rem.n))) PSK_TRY(flush_combined(s, st));
        bool taken = false;
        PSK_TRY(win_append(s, data, n, false, where == PSK_DEVICE_BORROWED && ((uintptr_t)data & 15) ? PSK_DEVICE : where, st, &taken));
        if (taken) return PSK_OK;  // (else: no memory for the window's key list -- the batch goes on below like any other)
    }
    if (where == PSK_DEVICE_BORROWED) where = PSK_DEVICE;  // (applied before this call returns: nothing is kept)
    Batch b;
    PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
    // Automatic write-combining (no opt-in): a unit-weight batch too small to pay for a pass over a big table would take one
    // fabric atomic per probe.  Adds commute (countingbloom.py:135-155; the clamp at 2^32-1 is applied by the fold just the
    // same), so the batch is scattered now and folded with its successors; every entry point that reads or removes flushes first.
    // (16-byte keys wait in the update window above; the other layouts here.  An append launches min(256, tiles) workgroups and workgroup i
    // always fills segment column i: batches of fewer than 256 tiles would pile the whole list into a few columns, which overflow long before
    // the list is full -- such batches take the direct kernel, as before round 3.)
    if (!weights && g_auto_combine != 0 && s->win.n == 0 && s->comb.rem.n == 0 && s->comb.brem.n() == 0 && s->scat.rem.n == 0 && (int64_t)n >= g_part_min_keys &&
        n >= (256u * 2048u * 7u) / (s->k ? s->k : 1u) && n * (uint64_t)s->k < s->m / 8 && g_auto_combine_keys > 0) {
        bool taken = false;
        PSK_TRY(scat_append(s, b, false, (uint64_t)g_auto_combine_keys, st, &taken));
        if (taken) {
            PSK_TRY(account_weights(s, (const uint32_t *)nullptr, n, PSK_CTR_ADDED, (long long)s->k, st, true));
            return finish(where, nullptr, st);
        }
    }
    PSK_TRY(flush_combined(s, st));  // write-combined updates reach the table before anything else touches it
    const uint32_t *w;
    PSK_TRY(stage_vec(s->s_w, weights, n, where, st, &w));
    PSK_TRY(post_acct(s, w, n, PSK_CTR_ADDED, (long long)s->k, st, true, false));
    unsigned long long *sat = (unsigned long long *)(s->ctr + PSK_CTR_SATURATED);
    {
        bool done = false;
        PSK_TRY(cbf_add_partitioned(s, b, w, st, &done));
        PSK_TRY(settle_acct(s, w, n, st));
        if (done) return finish(where, nullptr, st);
    }
    Mailbox mb;
    PSK_TRY(mailbox_arm(s, where, n, true, &mb));
    KeysInline64 ik;
    PSK_TRY(with_source_one(b, inline_key(layout, data, n, key_len, mb, &ik), [&](auto src) {
        if (s->pow2) return launch_apply(src, CbfAdd<true>{(uint32_t *)s->table, s->md, s->k, w, s->ctr, sat, false}, n, st, &mb);
        return launch_apply(src, CbfAdd<false>{(uint32_t *)s->table, s->md, s->k, w, s->ctr, sat, false}, n, st, &mb);
    }));
    return finish(where, nullptr, st, &mb);
}

// countingbloom.py:198-203 for a whole batch: from the min over the key's counters (a lookup) to the amount actually removed
//   mn == 0 (absent) or mn == 2^32-1 (frozen): nothing;  else to_remove = min(mn, num_els)
// a partial removal (mn < num_els) makes the result depend on the order inside the batch: `dep` is raised
static __global__ __launch_bounds__(kBlock) void k_cbf_to_remove(const uint32_t *mins, const uint32_t *weights, uint64_t n, uint32_t *to_remove, uint32_t *dep)
{
    uint32_t partial = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t mn = mins[i], w = weights ? weights[i] : 1u;
        uint32_t tr = 0;
        if (mn != 0 && mn != 0xFFFFFFFFu) {
            tr = mn > w ? w : mn;
            partial |= (uint32_t)(tr != w);
        }
        to_remove[i] = tr;
    }
    if (partial) *dep = 1u;
}

static __global__ void k_widen_u32(const uint32_t *w, uint64_t n, int64_t *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (int64_t)w[i];
}

// tmp[1] (sum of the amounts, k_weight_sum's booking slot) -> ctr[PSK_CTR_REMOVED]
static __global__ void k_book_removed(long long *ctr, const long long *tmp)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr[PSK_CTR_REMOVED] += tmp[PSK_CTR_REMOVED];
}

int64_t g_cbf_ordered_replays = 0;  // remove batches whose result depended on the order inside them: undone, replayed in order (tests)
__thread int64_t g_remove_exact = 1;         // option "remove_exact": 0 = round 3's composition (clamps and tallies instead of replaying)

// The validated remove (countingbloom.py:186-208) of a device-resident batch, as a TRANSACTION.  Unordered execution gives the
// reference's table whenever the result does not depend on the order inside the batch; here that is CHECKED and, where it fails, the
// batch is put back and executed in order:
//   0. unit weights into a big table: the optimistic decrement (psk_nibble.hpp) -- every key present: done in one pass over the table;
//   1. mins <- lookup of every key's k counters (the state BEFORE the batch);
//   2. amounts <- min(mn, num_els), 0 for absent / frozen keys; a partial removal (mn < num_els) raises the flag;
//   3. decrement by the amounts, wrapping, flag raised wherever a counter would go below zero or is frozen.  With T[c] >= the batch's
//      total on c for every counter, every key that step 2 found present is still present when its turn comes, whatever the order --
//      and a key found absent stays absent (removes only lower counters): the unordered result IS the sequential one;
//   4. flag up: the same amounts are added back (wrapping: the exact inverse) and the batch runs through k_cbf_ordered, one key after
//      the other -- the reference literally, for any batch (duplicates beyond their count, keys running a shared counter dry ...).
static int cbf_remove_exact(psk_sketch *s, const Batch &b, const uint32_t *w, hipStream_t st)
{
    if (b.n == 0) return PSK_OK;
    const bool big = part_wanted(b.n, s->k, 4);
    if (!w && big) {
        // Unit weights into a big table: decrement optimistically (psk_nibble.hpp) -- if every counter holds at least as much as the
        // batch takes from it, every key is removed and one pass 1 + ONE pass over the table did it.  The verdict is one 4-byte
        // read-back.  Otherwise the decrement is undone (exactly: wrapping arithmetic both ways) and the steps below take the batch.
        bool launched = false;
        PSK_TRY(cbf_remove_fast_begin(s, b, st, &launched));
        if (launched) {
            uint32_t flag = 1;
            HIP_TRY(hipMemcpyAsync(&flag, s->s_flag.p, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (flag == 0) return account_weights(s, (const uint32_t *)nullptr, b.n, PSK_CTR_REMOVED, (long long)s->k, st, false);
            PSK_TRY(cbf_remove_fast_undo(s, st));
        }
    }
    // s_aux: mins[n] | amounts[n] | scratch counter block | (ordered replay of a weighted batch: int64 weights[n])
    const uint64_t n4 = (b.n + 3) & ~3ULL;
    PSK_TRY(ensure(s->s_aux, n4 * 8 + 128 + (w ? b.n * 8 : 0)));
    uint32_t *mins = (uint32_t *)s->s_aux.p, *amount = mins + n4;
    long long *tmp = (long long *)(amount + n4);
    bool looked = false;
    if (big) PSK_TRY(cbf_check_partitioned(s, b, s->k, mins, st, &looked));
    if (!looked) {
        PSK_TRY(with_source(b, [&](auto src) {
            if (s->pow2) return launch_apply(src, CbfCheck<true>{(const uint32_t *)s->table, s->md, s->k, mins}, b.n, st);
            return launch_apply(src, CbfCheck<false>{(const uint32_t *)s->table, s->md, s->k, mins}, b.n, st);
        }));
    }
    PSK_TRY(ensure(s->s_flag, 8));
    uint32_t *flag = (uint32_t *)s->s_flag.p;
    HIP_TRY(hipMemsetAsync(flag, 0, 4, st));
    hipLaunchKernelGGL(k_cbf_to_remove, dim3(grid_for(b.n) > 1024 ? 1024 : grid_for(b.n)), dim3(kBlock), 0, st, (const uint32_t *)mins, w, b.n, amount, flag);
    HIP_TRY(hipGetLastError());
    // the amounts' sum: into a scratch block (booked once the verdict is in), and as this round's sum |w| for pass 2's wrap check
    HIP_TRY(hipMemsetAsync(tmp, 0, sizeof(long long) * PSK_CTR_COUNT, st));
    hipLaunchKernelGGL((k_weight_sum<uint32_t>), dim3(grid_for(b.n) > 256 ? 256 : grid_for(b.n)), dim3(kBlock), 0, st, (const uint32_t *)amount, b.n, tmp,
                       (int)PSK_CTR_REMOVED, (long long)s->k, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(s->ctr + 6, tmp + 6, sizeof(long long), hipMemcpyDeviceToDevice, st));
    s->acct.pending = false;
    auto decrement = [&](int opt, bool *partitioned) {  // opt 1: checked, 2: inverse.  The same path both times (same batch, same options)
        *partitioned = false;
        if (big) {
            s->acct.weights01 = w == nullptr;  // unit removes: every amount is 0 or 1 (masked unit probes may serve)
            const int rc = cbf_remove_partitioned(s, b, amount, st, partitioned, opt, flag);
            s->acct.weights01 = false;
            PSK_TRY(rc);
        }
        if (*partitioned) return (int)PSK_OK;
        return with_source(b, [&](auto src) {
            if (s->pow2) return launch_apply(src, CbfSubChecked<true>{(uint32_t *)s->table, s->md, s->k, amount, flag, opt == 2}, b.n, st);
            return launch_apply(src, CbfSubChecked<false>{(uint32_t *)s->table, s->md, s->k, amount, flag, opt == 2}, b.n, st);
        });
    };
    bool part1 = false, part2 = false;
    PSK_TRY(decrement(1, &part1));
    uint32_t verdict = 1;
    HIP_TRY(hipMemcpyAsync(&verdict, flag, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (verdict == 0) {
        hipLaunchKernelGGL(k_book_removed, dim3(1), dim3(1), 0, st, s->ctr, (const long long *)tmp);
        HIP_TRY(hipGetLastError());
        return PSK_OK;
    }
    PSK_TRY(decrement(2, &part2));
    if (part1 != part2) return fail(PSK_EHIP, "transactional remove: the undo took another path than the decrement");
    ++g_cbf_ordered_replays;
    const int64_t *w64 = nullptr;
    if (w) {
        int64_t *dst = (int64_t *)(tmp + PSK_CTR_COUNT + 2);
        hipLaunchKernelGGL(k_widen_u32, dim3(grid_for(b.n) > 1024 ? 1024 : grid_for(b.n)), dim3(kBlock), 0, st, w, b.n, dst);
        HIP_TRY(hipGetLastError());
        w64 = dst;
    }
    uint64_t *wide = nullptr;  // (k beyond the ordered kernel's register arrays: index / value lists in device scratch)
    if (s->k > (uint32_t)kMaxKOrdered) {
        PSK_TRY(ensure(s->s_out, 16ULL * s->k));
        wide = (uint64_t *)s->s_out.p;
    }
    return with_source(b, [&](auto src) {
        using Src = decltype(src);
        if (s->pow2)
            hipLaunchKernelGGL((k_cbf_ordered<Src, true>), dim3(1), dim3(64), 0, st, src, (uint32_t *)s->table, s->md, s->k, w64, (int)PSK_OP_REMOVE, b.n,
                               (uint32_t *)nullptr, (unsigned long long *)s->ctr, wide, (uint32_t *)nullptr, 0u);
        else
            hipLaunchKernelGGL((k_cbf_ordered<Src, false>), dim3(1), dim3(64), 0, st, src, (uint32_t *)s->table, s->md, s->k, w64, (int)PSK_OP_REMOVE, b.n,
                               (uint32_t *)nullptr, (unsigned long long *)s->ctr, wide, (uint32_t *)nullptr, 0u);
        HIP_TRY(hipGetLastError());
        return (int)PSK_OK;
    });
}

// the validated remove (countingbloom.py:186-208) of a device-resident batch: composed from the partitioned pipelines when the batch is
// large enough, else the direct kernel
static int cbf_remove_device(psk_sketch *s, const Batch &b, const uint32_t *w, hipStream_t st)
{
    if (b.n == 0) return PSK_OK;
    if (g_remove_exact != 0) return cbf_remove_exact(s, b, w, st);
    // (option "remove_exact" = 0, bench A/B: the one-kernel form -- per key: read the k counters, decide, subtract; exact for
    // well-formed batches, deviations tallied in PSK_CTR_VIOLATIONS)
    return with_source(b, [&](auto src) {
        using Src = decltype(src);
        if (s->pow2)
            hipLaunchKernelGGL((k_cbf_remove<Src, true>), dim3(grid_for(b.n)), dim3(kBlock), 0, st, src, (uint32_t *)s->table, s->md, s->k, w, b.n,
                               (unsigned long long *)s->ctr);
        else
            hipLaunchKernelGGL((k_cbf_remove<Src, false>), dim3(grid_for(b.n)), dim3(kBlock), 0, st, src, (uint32_t *)s->table, s->md, s->k, w, b.n,
                               (unsigned long long *)s->ctr);
        HIP_TRY(hipGetLastError());
        return (int)PSK_OK;
    });
}

extern "C" int psk_cbf_remove(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                              uint32_t key_len, const uint32_t *weights, int where, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_CBF);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    hipStream_t st = (hipStream_t)stream;
    if (where != PSK_HOST && where != PSK_DEVICE && where != PSK_DEVICE_BORROWED) return fail(PSK_EINVAL, "`where` must be PSK_HOST, PSK_DEVICE or PSK_DEVICE_BORROWED");
    if (win_eligible(s, layout, data, key_len, weights, n)) {
        // A small batch into a big table: it waits in the update window (with the adds around it, in order) for a shared pass over
        // the table; the flush proves that it would have removed every key at this point of the stream, or replays it right here.
        if (s->comb.add.n || s->comb.rem.n || s->comb.badd.n() || s->comb.brem.n() || (s->scat.ready && (s->scat.add.n || s->scat.rem.n))) PSK_TRY(flush_combined(s, st));
        bool taken = false;
        PSK_TRY(win_append(s, data, n, true, where == PSK_DEVICE_BORROWED && ((uintptr_t)data & 15) ? PSK_DEVICE : where, st, &taken));
        if (taken) return PSK_OK;
    }
    if (where == PSK_DEVICE_BORROWED) where = PSK_DEVICE;
    PSK_TRY(flush_combined(s, st));  // write-combined updates reach the table before anything else touches it
    Batch b;
    PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
    const uint32_t *w;
    PSK_TRY(stage_vec(s->s_w, weights, n, where, st, &w));
    PSK_TRY(cbf_remove_device(s, b, w, st));
    return finish(where, nullptr, st);
}

extern "C" int psk_cbf_check(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                             uint32_t key_len, int where, uint32_t *out, void *stream)
{
    CHECK_HANDLE_RO(s, PSK_KIND_CBF);
    if (n && !out) return fail(PSK_EINVAL, "out is NULL");
    if (layout == PSK_KEYS_HASHES && key_len == 0) return fail(PSK_EINVAL, "check needs at least one hash per key");
    hipStream_t st = (hipStream_t)stream;
    PSK_TRY(flush_combined(s, st));  // write-combined updates reach the table before anything else touches it
    Batch b;
    PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
    OutBuf o;
    PSK_TRY(stage_out(s->s_out, out, n * 4, where, &o));
    // countingbloom.py:174 takes the min over ALL supplied hashes (not just the first k)
    const uint32_t kk = layout == PSK_KEYS_HASHES ? key_len : s->k;
    {
        bool done = false;
        s->shadow.allow = !s->shadow.exposed;  // (only here: a lookup INSIDE an updating entry point is followed by writes at the same table version)
        const int rc = cbf_check_partitioned(s, b, kk, (uint32_t *)o.dev, st, &done);
        s->shadow.allow = false;
        PSK_TRY(rc);
        if (done) return finish(where, &o, st);
    }
    Mailbox mb;
    PSK_TRY(mailbox_arm(s, where, n, o.is_pinned, &mb));
    KeysInline64 ik;
    PSK_TRY(with_source_one(b, inline_key(layout, data, n, key_len, mb, &ik), [&](auto src) {
        if (s->pow2) return launch_apply(src, CbfCheck<true>{(const uint32_t *)s->table, s->md, kk, (uint32_t *)o.dev}, n, st, &mb);
        return launch_apply(src, CbfCheck<false>{(const uint32_t *)s->table, s->md, kk, (uint32_t *)o.dev}, n, st, &mb);
    }));
    return finish(where, &o, st, &mb);
}

extern "C" int psk_cbf_update_ordered(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                                      uint32_t key_len, const int64_t *weights, int opmode, int where, uint32_t *out,
                                      void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_CBF);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    if (opmode < PSK_OP_ADD || opmode > PSK_OP_SIGNED) return fail(PSK_EINVAL, "bad opmode %d", opmode);
    hipStream_t st = (hipStream_t)stream;
    PSK_TRY(flush_combined(s, st));  // write-combined updates reach the table before anything else touches it
    Batch b;
    PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
    const int64_t *w;
    PSK_TRY(stage_vec(s->s_w, weights, n, where, st, &w));
    OutBuf o;
    PSK_TRY(stage_out(s->s_out, out, out ? n * 4 : 0, where, &o));
    uint64_t *wide = nullptr;  // k beyond the register arrays (fpr below ~1e-20): index / value lists live in device scratch
    if (s->k > (uint32_t)kMaxKOrdered) {
        PSK_TRY(ensure(s->s_aux, 16ULL * s->k));
        wide = (uint64_t *)s->s_aux.p;
    }
    Mailbox mb;
    PSK_TRY(mailbox_arm(s, where, n, out && o.is_pinned, &mb));
    if (mb.word && n == 1 && weights && weights[0] == 1) w = nullptr;  // (a null weight list means 1: no read of the pinned page for `cbf.add(key)`)
    KeysInline64 ik;
    if (n) {
        PSK_TRY(with_source_one(b, inline_key(layout, data, n, key_len, mb, &ik), [&](auto src) {
            using Src = decltype(src);
            if (s->pow2)
                hipLaunchKernelGGL((k_cbf_ordered<Src, true>), dim3(1), dim3(64), 0, st, src, (uint32_t *)s->table, s->md,
                                   s->k, w, opmode, n, (uint32_t *)(out ? o.dev : nullptr), (unsigned long long *)s->ctr, wide, mb.dev(), mb.seq);
            else
                hipLaunchKernelGGL((k_cbf_ordered<Src, false>), dim3(1), dim3(64), 0, st, src, (uint32_t *)s->table, s->md,
                                   s->k, w, opmode, n, (uint32_t *)(out ? o.dev : nullptr), (unsigned long long *)s->ctr, wide, mb.dev(), mb.seq);
            HIP_TRY(hipGetLastError());
            return (int)PSK_OK;
        }));
    }
    return finish(where, &o, st, &mb);
}

// ---------------------------------------------------------- CountMinSketch
template <bool NEG>
static int cms_update(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n, uint32_t key_len,
                      const int32_t *weights, int where, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_CMS);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    hipStream_t st = (hipStream_t)stream;
    Batch b;
    PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
    const int32_t *w;
    PSK_TRY(stage_vec(s->s_w, weights, n, where, st, &w));
    PSK_TRY(post_acct(s, w, n, NEG ? PSK_CTR_REMOVED : PSK_CTR_ADDED, 1LL, st, true, true));
    unsigned long long *sat = (unsigned long long *)(s->ctr + PSK_CTR_SATURATED);
    {
        bool done = false;
        PSK_TRY(NEG ? cms_remove_partitioned(s, b, (const uint32_t *)w, st, &done) : cms_add_partitioned(s, b, (const uint32_t *)w, st, &done));
        PSK_TRY(settle_acct(s, w, n, st));
        if (done) return finish(where, nullptr, st);
    }
    Mailbox mb;
    PSK_TRY(mailbox_arm(s, where, n, true, &mb));
    KeysInline64 ik;
    PSK_TRY(with_source_one(b, inline_key(layout, data, n, key_len, mb, &ik), [&](auto src) {
        if (s->pow2)
            return launch_apply(src, CmsAdd<true, NEG>{(int32_t *)s->table, s->md, s->k, w, s->ctr, sat, false}, n, st, &mb);
        return launch_apply(src, CmsAdd<false, NEG>{(int32_t *)s->table, s->md, s->k, w, s->ctr, sat, false}, n, st, &mb);
    }));
    return finish(where, nullptr, st, &mb);
}

extern "C" int psk_cms_add(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                           uint32_t key_len, const int32_t *weights, int where, void *stream)
{
    return cms_update<false>(s, layout, data, offsets, n, key_len, weights, where, stream);
}

extern "C" int psk_cms_remove(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                              uint32_t key_len, const int32_t *weights, int where, void *stream)
{
    return cms_update<true>(s, layout, data, offsets, n, key_len, weights, where, stream);
}

extern "C" int psk_cms_check(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                             uint32_t key_len, int where, int query, int32_t *out, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_CMS);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    if (query != PSK_Q_MIN && query != PSK_Q_MEAN) return fail(PSK_EINVAL, "psk_cms_check handles MIN and MEAN queries");
    if (n && !out) return fail(PSK_EINVAL, "out is NULL");
    hipStream_t st = (hipStream_t)stream;
    Batch b;
    PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
    OutBuf o;
    PSK_TRY(stage_out(s->s_out, out, n * 4, where, &o));
    const bool mean = query == PSK_Q_MEAN;
    {
        bool done = false;
        PSK_TRY(cms_check_partitioned(s, b, query, 0, o.dev, st, &done));
        if (done) return finish(where, &o, st);
    }
    Mailbox mb;
    PSK_TRY(mailbox_arm(s, where, n, o.is_pinned, &mb));
    KeysInline64 ik;
    PSK_TRY(with_source_one(b, inline_key(layout, data, n, key_len, mb, &ik), [&](auto src) {
        if (s->pow2) return launch_apply(src, CmsCheck<true>{(const int32_t *)s->table, s->md, s->k, (int32_t *)o.dev, mean}, n, st, &mb);
        return launch_apply(src, CmsCheck<false>{(const int32_t *)s->table, s->md, s->k, (int32_t *)o.dev, mean}, n, st, &mb);
    }));
    return finish(where, &o, st, &mb);
}

extern "C" int psk_cms_check_meanmin(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                                     uint32_t key_len, int where, int64_t elements_added, int64_t *out, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_CMS);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    if (s->m < 2) return fail(PSK_EINVAL, "mean-min query needs width >= 2 (divides by width-1)");
    if (n && !out) return fail(PSK_EINVAL, "out is NULL");
    hipStream_t st = (hipStream_t)stream;
    Batch b;
    PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
    OutBuf o;
    PSK_TRY(stage_out(s->s_out, out, n * 8, where, &o));
    if (s->k > (uint32_t)kMaxDepthMeanMin) {
        // deeper than the per-lane register array: the ordered kernel in query-only mode (one lane, scratch list)
        PSK_TRY(ensure(s->s_aux, 8ULL * s->k));
        if (n) {
            PSK_TRY(with_source(b, [&](auto src) {
                using Src = decltype(src);
                if (s->pow2)
                    hipLaunchKernelGGL((k_cms_ordered<Src, true>), dim3(1), dim3(64), 0, st, src, (int32_t *)s->table, s->md, s->k,
                                       (const int64_t *)nullptr, 3, (int)PSK_Q_MEANMIN, elements_added, n, (int64_t *)o.dev, s->ctr, (int64_t *)s->s_aux.p,
                                       (uint32_t *)nullptr, 0u);
                else
                    hipLaunchKernelGGL((k_cms_ordered<Src, false>), dim3(1), dim3(64), 0, st, src, (int32_t *)s->table, s->md, s->k,
                                       (const int64_t *)nullptr, 3, (int)PSK_Q_MEANMIN, elements_added, n, (int64_t *)o.dev, s->ctr, (int64_t *)s->s_aux.p,
                                       (uint32_t *)nullptr, 0u);
                HIP_TRY(hipGetLastError());
                return (int)PSK_OK;
            }));
        }
        return finish(where, &o, st);
    }
    {
        bool done = false;
        PSK_TRY(cms_check_partitioned(s, b, PSK_Q_MEANMIN, elements_added, o.dev, st, &done));
        if (done) return finish(where, &o, st);
    }
    Mailbox mb;
    PSK_TRY(mailbox_arm(s, where, n, o.is_pinned, &mb));
    KeysInline64 ik;
    PSK_TRY(with_source_one(b, inline_key(layout, data, n, key_len, mb, &ik), [&](auto src) {
        if (s->pow2)
            return launch_apply(src, CmsCheckMeanMin<true>{(const int32_t *)s->table, s->md, s->k, elements_added, (int64_t *)o.dev}, n, st, &mb);
        return launch_apply(src, CmsCheckMeanMin<false>{(const int32_t *)s->table, s->md, s->k, elements_added, (int64_t *)o.dev}, n, st, &mb);
    }));
    return finish(where, &o, st, &mb);
}

extern "C" int psk_cms_update_ordered(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                                      uint32_t key_len, const int64_t *weights, int opmode, int query,
                                      int64_t elements_added_in, int where, int64_t *out, void *stream)
{
    CHECK_HANDLE(s, PSK_KIND_CMS);
    PSK_TRY(check_hashes_width(s, layout, key_len));
    if (opmode < PSK_OP_ADD || opmode > PSK_OP_SIGNED) return fail(PSK_EINVAL, "bad opmode %d", opmode);
    if (query < PSK_Q_MIN || query > PSK_Q_MEANMIN) return fail(PSK_EINVAL, "bad query %d", query);
    if (query == PSK_Q_MEANMIN && s->m < 2) return fail(PSK_EINVAL, "mean-min query needs width >= 2");
    hipStream_t st = (hipStream_t)stream;
    Batch b;
    PSK_TRY(stage_batch(s->s_keys, s->s_offs, layout, data, offsets, n, key_len, where, st, &b));
    const int64_t *w;
    PSK_TRY(stage_vec(s->s_w, weights, n, where, st, &w));
    OutBuf o;
    PSK_TRY(stage_out(s->s_out, out, out ? (n + 1) * 8 : 0, where, &o));  // out[n] = elements_added after the batch
    int64_t *wide = nullptr;  // depth beyond the register array: the per-op value list lives in device scratch
    if (s->k > (uint32_t)kMaxDepthMeanMin) {
        PSK_TRY(ensure(s->s_aux, 8ULL * s->k));
        wide = (int64_t *)s->s_aux.p;
    }
    Mailbox mb;
    PSK_TRY(mailbox_arm(s, where, n, out && o.is_pinned, &mb));
    if (mb.word && n == 1 && weights && weights[0] == 1) w = nullptr;  // (a null weight list means 1: no read of the pinned page for `cms.add(key)`)
    KeysInline64 ik;
    PSK_TRY(with_source_one(b, inline_key(layout, data, n, key_len, mb, &ik), [&](auto src) {
        using Src = decltype(src);
        if (s->pow2)
            hipLaunchKernelGGL((k_cms_ordered<Src, true>), dim3(1), dim3(64), 0, st, src, (int32_t *)s->table, s->md, s->k, w,
                               opmode, query, elements_added_in, n, (int64_t *)(out ? o.dev : nullptr), s->ctr, wide, mb.dev(), mb.seq);
        else
            hipLaunchKernelGGL((k_cms_ordered<Src, false>), dim3(1), dim3(64), 0, st, src, (int32_t *)s->table, s->md, s->k, w,
                               opmode, query, elements_added_in, n, (int64_t *)(out ? o.dev : nullptr), s->ctr, wide, mb.dev(), mb.seq);
        HIP_TRY(hipGetLastError());
        return (int)PSK_OK;
    }));
    return finish(where, &o, st, &mb);
}

// ------------------------------------------------------------------ hashing
static thread_local DevBuf g_hkeys, g_hoffs, g_hout;  // staging for the handle-less calls

extern "C" int psk_fnv1a_hash(int layout, const void *data, const uint64_t *offsets, uint64_t n, uint32_t key_len,
                              uint32_t depth, int where, uint64_t *out, int device, void *stream)
{
    if (layout == PSK_KEYS_HASHES) return fail(PSK_EINVAL, "psk_fnv1a_hash needs a key layout");
    if (n && depth && !out) return fail(PSK_EINVAL, "out is NULL");
    PSK_USE_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    Batch b;
    PSK_TRY(stage_batch(g_hkeys, g_hoffs, layout, data, offsets, n, key_len, where, st, &b));
    OutBuf o;
    PSK_TRY(stage_out(g_hout, out, n * depth * 8, where, &o));
    if (n && depth) {
        PSK_TRY(with_source(b, [&](auto src) {
            using Src = decltype(src);
            hipLaunchKernelGGL((k_hash<Src>), dim3(grid_for(n)), dim3(kBlock), 0, st, src, (uint64_t *)o.dev, depth, n);
            HIP_TRY(hipGetLastError());
            return (int)PSK_OK;
        }));
    }
    return finish(where, &o, st);
}

// hashes.py:125-150 default_md5 / default_sha256 as digest chains (psk_digest.hpp); byte keys only
extern "C" int psk_digest_chain(int algo, int layout, const void *data, const uint64_t *offsets, uint64_t n, uint32_t key_len,
                                uint32_t depth, int where, uint64_t *out, int device, void *stream)
{
    if (algo != PSK_DIGEST_MD5 && algo != PSK_DIGEST_SHA256) return fail(PSK_EINVAL, "unknown digest %d", algo);
    if (layout != PSK_KEYS_FIXED && layout != PSK_KEYS_VARLEN8)
        return fail(PSK_EINVAL, "digest chains hash bytes: use PSK_KEYS_FIXED or PSK_KEYS_VARLEN8 (a str is UTF-8 encoded by the caller)");
    if (n && depth && !out) return fail(PSK_EINVAL, "out is NULL");
    PSK_USE_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    Batch b;
    PSK_TRY(stage_batch(g_hkeys, g_hoffs, layout, data, offsets, n, key_len, where, st, &b));
    OutBuf o;
    PSK_TRY(stage_out(g_hout, out, n * depth * 8, where, &o));
    if (n && depth) {
        const uint64_t *offs = layout == PSK_KEYS_VARLEN8 ? b.offs : nullptr;
        const dim3 grid((unsigned)((n + 255) / 256 > 65535 ? 65535 : (n + 255) / 256));
        if (algo == PSK_DIGEST_MD5)
            hipLaunchKernelGGL((k_digest_chain<Md5>), grid, dim3(256), 0, st, (const uint8_t *)b.data, offs, key_len, n, depth, (uint64_t *)o.dev);
        else
            hipLaunchKernelGGL((k_digest_chain<Sha256>), grid, dim3(256), 0, st, (const uint8_t *)b.data, offs, key_len, n, depth, (uint64_t *)o.dev);
        HIP_TRY(hipGetLastError());
    }
    return finish(where, &o, st);
}

// ------------------------------------------------------------ table algebra
static int check_vec(const void *a, const void *b, uint64_t nwords32)
{
    if (!a || !b) return fail(PSK_EINVAL, "table pointer is NULL");
    if (nwords32 % 4 || ((uintptr_t)a & 15) || ((uintptr_t)b & 15))
        return fail(PSK_EINVAL, "tables must be 16-byte aligned and a multiple of 16 bytes long");
    return PSK_OK;
}

extern "C" int psk_table_or(void *dst, const void *src, uint64_t nwords32, int device, void *stream)
{
    PSK_TRY(check_vec(dst, src, nwords32));
    PSK_USE_DEVICE(device);
    if (!nwords32) return PSK_OK;
    hipLaunchKernelGGL((k_table_binop<OpOr>), dim3(grid_for(nwords32 / 4)), dim3(kBlock), 0, (hipStream_t)stream, (uint4 *)dst,
                       (const uint4 *)src, nwords32 / 4, OpOr{});
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

extern "C" int psk_table_and(void *dst, const void *src, uint64_t nwords32, int device, void *stream)
{
    PSK_TRY(check_vec(dst, src, nwords32));
    PSK_USE_DEVICE(device);
    if (!nwords32) return PSK_OK;
    hipLaunchKernelGGL((k_table_binop<OpAnd>), dim3(grid_for(nwords32 / 4)), dim3(kBlock), 0, (hipStream_t)stream, (uint4 *)dst,
                       (const uint4 *)src, nwords32 / 4, OpAnd{});
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

static int table_count(const void *tab, uint64_t nwords32, int mode, uint64_t *out_host, int device, void *stream)
{
    PSK_TRY(check_vec(tab, tab, nwords32));
    if (!out_host) return fail(PSK_EINVAL, "out is NULL");
    PSK_USE_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 8));
    hipError_t e = hipMemsetAsync(d, 0, 8, st);
    if (e == hipSuccess && nwords32) {
        const int g = grid_for(nwords32 / 4) > 1024 ? 1024 : grid_for(nwords32 / 4);
        hipLaunchKernelGGL(k_table_count, dim3(g), dim3(kBlock), 0, st, (const uint4 *)tab, nwords32 / 4, mode, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, d, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d);
    if (e != hipSuccess) return fail(PSK_EHIP, "table count failed: %s", hipGetErrorString(e));
    return PSK_OK;
}

extern "C" int psk_table_popcount(const void *tab, uint64_t nwords32, uint64_t *out_host, int device, void *stream)
{
    return table_count(tab, nwords32, 0, out_host, device, stream);
}

extern "C" int psk_table_nonzero_u32(const void *tab, uint64_t nwords32, uint64_t *out_host, int device, void *stream)
{
    return table_count(tab, nwords32, 1, out_host, device, stream);
}

extern "C" int psk_table_add_sat_i32(void *dst, const void *src, uint64_t n, int device, void *stream)
{
    if (!dst || !src) return fail(PSK_EINVAL, "table pointer is NULL");
    PSK_USE_DEVICE(device);
    if (!n) return PSK_OK;
    hipLaunchKernelGGL(k_add_sat_i32, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, (int32_t *)dst, (const int32_t *)src, n);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

extern "C" int psk_table_add_u32(void *dst, const void *src, uint64_t n, uint64_t *overflowed_host, int device, void *stream)
{
    if (!dst || !src) return fail(PSK_EINVAL, "table pointer is NULL");
    PSK_USE_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 8));
    hipError_t e = hipMemsetAsync(d, 0, 8, st);
    if (e == hipSuccess && n) {
        hipLaunchKernelGGL(k_add_u32, dim3(grid_for(n)), dim3(kBlock), 0, st, (uint32_t *)dst, (const uint32_t *)src, n, d);
        e = hipGetLastError();
    }
    uint64_t ov = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&ov, d, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d);
    if (e != hipSuccess) return fail(PSK_EHIP, "table add failed: %s", hipGetErrorString(e));
    if (overflowed_host) *overflowed_host = ov;
    return PSK_OK;
}

extern "C" int psk_cbf_intersect(void *dst, const void *a, const void *b, uint64_t n, uint64_t *overflowed_host, int device, void *stream)
{
    if (!dst || !a || !b) return fail(PSK_EINVAL, "table pointer is NULL");
    PSK_USE_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 8));
    hipError_t e = hipMemsetAsync(d, 0, 8, st);
    if (e == hipSuccess && n) {
        const int g = grid_for(n) > 2048 ? 2048 : grid_for(n);
        hipLaunchKernelGGL(k_cbf_intersect, dim3(g), dim3(kBlock), 0, st, (uint32_t *)dst, (const uint32_t *)a, (const uint32_t *)b, n, d);
        e = hipGetLastError();
    }
    uint64_t ov = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&ov, d, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d);
    if (e != hipSuccess) return fail(PSK_EHIP, "cbf intersect failed: %s", hipGetErrorString(e));
    if (overflowed_host) *overflowed_host = ov;
    return PSK_OK;
}

extern "C" int psk_cbf_jaccard_counts(const void *a, const void *b, uint64_t n, uint64_t out_host[2], int device, void *stream)
{
    if (!a || !b || !out_host) return fail(PSK_EINVAL, "NULL argument");
    PSK_USE_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 16));
    hipError_t e = hipMemsetAsync(d, 0, 16, st);
    if (e == hipSuccess && n) {
        const int g = grid_for(n) > 1024 ? 1024 : grid_for(n);
        hipLaunchKernelGGL(k_cbf_jaccard, dim3(g), dim3(kBlock), 0, st, (const uint32_t *)a, (const uint32_t *)b, n, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, d, 16, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d);
    if (e != hipSuccess) return fail(PSK_EHIP, "cbf jaccard failed: %s", hipGetErrorString(e));
    return PSK_OK;
}

// per-handle accounting of the device scratch (the write-combining / window lists are large and allocated on first use)
extern "C" int psk_scratch_bytes(psk_sketch *s, uint64_t bytes[3])
{
    if (!s || !bytes) return fail(PSK_EINVAL, "psk_scratch_bytes: NULL argument");
    uint64_t all = 0, waiting = 0;
    for (const DevBuf *b : {&s->s_keys, &s->s_offs, &s->s_w, &s->s_out, &s->s_aux, &s->s_part, &s->s_cnt, &s->s_flag, &s->s_tflag, &s->s_part2, &s->s_cnt2, &s->s_merge, &s->s_vals, &s->s_perm, &s->s_run, &s->s_tally,
                            &s->comb.add.keys, &s->comb.add.w, &s->comb.rem.keys, &s->comb.rem.w, &s->scat.add.part, &s->scat.add.cnt, &s->scat.rem.part, &s->scat.rem.cnt, &s->s_brw, &s->shadow.img,
                            &s->win.keys, &s->s_snap, &s->s_wstat, &s->s_phase})
        if (b->p) all += b->cap;
    for (const DevBuf *b : {&s->comb.add.keys, &s->comb.add.w, &s->comb.rem.keys, &s->comb.rem.w, &s->scat.add.part, &s->scat.add.cnt, &s->scat.rem.part, &s->scat.rem.cnt, &s->win.keys, &s->s_snap})
        if (b->p) waiting += b->cap;
    bytes[0] = all;
    bytes[1] = waiting;
    bytes[2] = s->shadow.img.p ? s->shadow.img.cap : 0;
    return PSK_OK;
}

// release the partition / staging scratch of a handle (hundreds of MB after a large batch); it regrows on demand
extern "C" int psk_release_scratch(psk_sketch *s)
{
    if (!s) return fail(PSK_EINVAL, "sketch handle is NULL");
    PSK_USE_DEVICE(s->device);
    if (s->pend.active) return fail(PSK_EINVAL, "a split lookup is pending: finish it before releasing the scratch buffers");
    ho_apply(s);  // (what is waiting is applied under this sketch's own options)
    PSK_TRY(flush_combined(s, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    for (DevBuf *b : {&s->s_keys, &s->s_offs, &s->s_w, &s->s_out, &s->s_aux, &s->s_part, &s->s_cnt, &s->s_flag, &s->s_tflag, &s->s_part2, &s->s_cnt2, &s->s_merge, &s->s_vals, &s->s_perm, &s->s_run, &s->s_tally,
                      &s->comb.add.keys, &s->comb.add.w, &s->comb.rem.keys, &s->comb.rem.w, &s->scat.add.part, &s->scat.add.cnt, &s->scat.rem.part, &s->scat.rem.cnt, &s->s_brw, &s->shadow.img,
                      &s->win.keys, &s->s_snap, &s->s_wstat, &s->s_phase}) {
        if (b->p) HIP_TRY(hipFree(b->p));
        b->p = nullptr;
        b->cap = 0;
    }
    s->scat.ready = false;
    s->shadow.built = s->shadow.seen = ~0ULL;
    s->shadow.seen_count = 0;
    return PSK_OK;
}

extern "C" int psk_or_reduce_slices(void *dst, const void *src, uint32_t nslices, uint64_t slice_words32, int device, void *stream)
{
    PSK_TRY(check_vec(dst, src, slice_words32));
    if (nslices == 0) return fail(PSK_EINVAL, "nslices must be > 0");
    PSK_USE_DEVICE(device);
    if (!slice_words32) return PSK_OK;
    hipLaunchKernelGGL(k_or_reduce, dim3(grid_for(slice_words32 / 4)), dim3(kBlock), 0, (hipStream_t)stream, (uint4 *)dst,
                       (const uint4 *)src, nslices, slice_words32 / 4);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

// ------------------------------------------------------- synthetic streams
extern "C" int psk_gen_keys16(void *dst_dev, uint64_t start, uint64_t n, uint64_t seed, int device, void *stream)
{
    if (n && (!dst_dev || ((uintptr_t)dst_dev & 15))) return fail(PSK_EINVAL, "dst must be a 16-byte aligned device pointer");
    PSK_USE_DEVICE(device);
    if (!n) return PSK_OK;
    hipLaunchKernelGGL(k_gen_keys16, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, (ulonglong2 *)dst_dev, start, n, seed);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

extern "C" int psk_gen_weights(void *dst_dev, uint64_t start, uint64_t n, uint64_t seed, int device, void *stream)
{
    if (n && !dst_dev) return fail(PSK_EINVAL, "dst is NULL");
    PSK_USE_DEVICE(device);
    if (!n) return PSK_OK;
    hipLaunchKernelGGL(k_gen_weights, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, (int32_t *)dst_dev, start, n, seed);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

extern "C" int psk_gups(void *table_dev, uint64_t nwords32, uint64_t n, int op, uint64_t seed, uint64_t *sink_dev, int device,
                        void *stream)
{
    if (!table_dev || !nwords32) return fail(PSK_EINVAL, "bad table");
    if (op == 2 && !sink_dev) return fail(PSK_EINVAL, "load mode needs a sink");
    PSK_USE_DEVICE(device);
    if (!n) return PSK_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 g(grid_for(n)), blk(kBlock);
    if (op == 0) hipLaunchKernelGGL((k_gups<0>), g, blk, 0, st, (uint32_t *)table_dev, nwords32, n, seed, (unsigned long long *)sink_dev);
    else if (op == 1) hipLaunchKernelGGL((k_gups<1>), g, blk, 0, st, (uint32_t *)table_dev, nwords32, n, seed, (unsigned long long *)sink_dev);
    else if (op == 2) hipLaunchKernelGGL((k_gups<2>), g, blk, 0, st, (uint32_t *)table_dev, nwords32, n, seed, (unsigned long long *)sink_dev);
    else return fail(PSK_EINVAL, "bad gups op %d", op);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}
