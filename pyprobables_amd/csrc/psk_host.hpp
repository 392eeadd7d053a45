// psk_host.hpp -- host-side declarations shared by the translation units of libpsk_hip.so
// (psk_capi.hip = C ABI + direct kernels; psk_part_*.hip = the partitioned-path launchers, split so that
// hipcc can build the ~300 k_part_scatter instantiations in parallel).
#pragma once
#include "psk_device.hpp"
#include "psk_partition.hpp"
#include "psk_part_bins.hpp"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/psk.h"

using namespace psk;

#define PSK_HIDDEN __attribute__((visibility("hidden")))

// ------------------------------------------------------------------ errors
PSK_HIDDEN int fail(int code, const char *fmt, ...);  // records the thread-local message, returns `code`

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(e__ == hipErrorOutOfMemory ? PSK_ENOMEM : PSK_EHIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e__), __FILE__, __LINE__);                                    \
    } while (0)

#define PSK_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__ != PSK_OK)      \
            return rc__;         \
    } while (0)

// Every entry point works on the device its handle / `device` argument names and puts the caller's current device back
// before it returns: the library must not move the calling thread (torch, another runtime) to a different GPU.
struct DeviceScope {
    int prev = -1;
    bool moved = false;
    int enter(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == dev) return hipSuccess;
        const hipError_t e = hipSetDevice(dev);
        moved = e == hipSuccess && prev >= 0;
        return (int)e;
    }
    ~DeviceScope()
    {
        if (moved) (void)hipSetDevice(prev);
    }
};
#define PSK_USE_DEVICE(dev)                                                                                   \
    DeviceScope psk_device_scope__;                                                                           \
    do {                                                                                                      \
        const hipError_t e__ = (hipError_t)psk_device_scope__.enter(dev);                                     \
        if (e__ != hipSuccess) return fail(PSK_EHIP, "hipSetDevice(%d) failed: %s", (int)(dev), hipGetErrorString(e__)); \
    } while (0)

// ---------------------------------------------------------- key batches
struct Batch {  // device-resident view of one key batch
    int layout;
    const void *data;
    const uint64_t *offs;
    uint64_t n;
    uint32_t key_len;
};

// ------------------------------------------------------------------ handle
struct DevBuf {
    void *p = nullptr;    // device scratch, grown on demand
    uint64_t cap = 0;
    void *pin = nullptr;  // kPinBytes of pinned, device-visible host memory for tiny PSK_HOST batches: the kernel reads /
                          // writes it in place, so a single-key call is one launch + one stream sync, no staging copies
};
constexpr uint64_t kPinBytes = 4096;

// Per-sketch tunables (psk_sketch_set_option): the options below can differ between two sketches of one process.  The variables the
// launchers read (g_part_min_keys ...) are THREAD-LOCAL effective values: every entry point that takes a handle sets them from the handle's
// overrides, falling back to the process-wide defaults psk_set_option maintains (kHoUnset = inherit the default).
enum HandleOpt { HO_PART_MIN_KEYS, HO_CBF_SHADOW, HO_AUTO_COMBINE, HO_WINDOW, HO_WINDOW_KEYS, HO_SCRATCH_BUDGET, HO_REMOVE_EXACT, HO_BLOOM_LOOKUP, HO_COUNT };
constexpr int64_t kHoUnset = INT64_MIN;

struct psk_sketch {
    int64_t opt[HO_COUNT] = {kHoUnset, kHoUnset, kHoUnset, kHoUnset, kHoUnset, kHoUnset, kHoUnset, kHoUnset};
    bool table_private = false;  // option "table_private": the holder of a caller-owned table announces every outside write (psk_table_info / psk_rescan_bound)
    int kind;
    int device;
    uint64_t m;        // bits (bloom), counters (cbf), width (cms)
    uint32_t k;        // hashes (bloom/cbf), depth (cms)
    Mod md;
    bool pow2;
    void *table;
    bool owns_table;
    uint64_t padded_bytes, logical_bytes;
    long long *ctr;    // device int64[PSK_CTR_COUNT]
    DevBuf s_keys, s_offs, s_w, s_out, s_aux;  // staging for PSK_HOST buffers
    volatile uint32_t *mbox = nullptr;         // completion mailbox of tiny PSK_HOST batches (psk_capi.hip Mailbox): a pinned word the
    uint32_t mbox_seq = 0;                     // kernel stores the call's sequence number into, behind its results
    uint32_t mbox_timeouts = 0, mbox_skipped = 0;  // polls in a row that gave up / calls since the handle stopped polling (mailbox_arm)
    DevBuf s_part, s_cnt;                      // partitioned path: bucket buffer + per-bucket fill counts
    DevBuf s_flag;                             // split lookup: "a segment overflowed" flag
    DevBuf s_tflag;                            // tile-flag Bloom lookups: one uint32 per pass-1 tile of a round; "flagged" = holds the round's
    uint32_t tflag_gen = 0;                    // generation number (never 0), so the flags are never reset
    uint32_t tflag_wgs = 0;  // pass-1 workgroups of the tile-flag lookup in progress (0: the shape's own count; tile_flag_geometry)
    DevBuf s_part2, s_cnt2;                    // two-level path: bucket buffer + fill counts after the second split
    DevBuf s_merge;                            // multi-GPU merge (psk_merge_or / _sum): exchange buffers
    DevBuf s_tally;                            // weighted pass 1: (sum w, sum |w|) per workgroup, folded by k_tally_fold
    DevBuf s_brw;                              // borrowed write-combined batches: pointer / prefix tables of a flush
    DevBuf s_vals, s_perm, s_run;              // partitioned counter lookups: values, per-key stage positions, per-(tile, slice) runs
    // Weighted counter updates: the caller (psk_capi.hip) posts what has to be accounted for the batch; a partitioned launcher
    // that scatters the weights takes the request over (PayWeight::tally sums them inside pass 1) and clears `pending`;
    // otherwise the caller runs the stand-alone pass over the weights (k_weight_sum).
    struct {
        bool pending = false;
        int which = -1;          // PSK_CTR_ADDED / PSK_CTR_REMOVED
        long long bound_mult = 1;
        bool grow_bound = true, weights_signed = false;
        bool weights01 = false;  // the weights are 0 / 1 flags (amounts of a validated unit-weight CBF remove): masked unit probes may serve
    } acct;
    // Bloom lookups: which scheme the next large batch takes (g_bloom_lookup = 2, auto).  The kernels tally what they see
    // (keyed: probes that missed; return trip: keys answered absent) into lk_dev; the tally is copied to a pinned host page
    // when the call ends and read -- without any synchronisation, so possibly one call late -- when the next one starts.
    struct {
        unsigned long long *dev = nullptr;           // [0] misses of the call in flight
        volatile unsigned long long *pin = nullptr;  // [0] misses, [1] units (probes or keys), [2] scheme that produced them, [3] calls published
        unsigned long long seen_seq = 0;             // pin[3] when the choice last looked
        uint32_t clean = 0;                          // finished calls in a row that missed (almost) nothing
        int mode = 0;                                // 0 keyed probes + miss stores, 1 return trip, 3 tile flags (PayTileTag)
    } lk;
    // write-combined CBF updates (psk_cbf_update_combined): key batches wait here until a list is full, then each list is
    // applied as ONE partitioned update (the fold of a big table read-modify-writes the whole table whatever the batch size)
    struct PendList {
        DevBuf keys, w;      // uint8[cap][key_len], uint32[cap]
        uint64_t n = 0;
        bool unit = true;    // every batch so far had unit weights
    };
    struct BorrowList {      // PSK_DEVICE_BORROWED batches (16-byte keys): pointers only, hashed where they lie at the flush
        std::vector<const void *> base;
        std::vector<uint64_t> start{0};   // prefix of the batch sizes
        uint64_t n() const { return start.back(); }
        void clear() { base.clear(); start.assign(1, 0); }
    };
    struct {
        uint32_t key_len = 0;
        uint64_t cap = 0;    // keys per list
        PendList add, rem;
        BorrowList badd, brem;
    } comb;
    // Write-combined unit-weight CBF updates kept as SCATTERED PROBES (round 3): pass 1 runs when a batch is handed over and
    // appends to persistent (slice, workgroup) segments; a flush is pass 2 alone (psk_nibble.hpp: one level of 2^18-counter
    // slices), or -- few probes -- a drain with atomics.  No key copies, the keys are read once.
    struct ScatList {
        DevBuf part, cnt;    // bucket buffer (nbuckets x nwg segments of segcap 16-byte groups) + groups per segment
        uint64_t n = 0;      // keys behind the probes
    };
    struct {
        bool ready = false;
        PartGeom g{};        // the fixed geometry of both lists (nbuckets, shift, nwg, segcap, k; append = 1)
        uint64_t cap = 0;    // keys per list the segments were sized for
        ScatList add, rem;
        hipEvent_t ev = nullptr;       // recorded behind the last append (either mechanism): a flush on ANOTHER stream waits for it
        hipStream_t last = nullptr;
        bool appended = false;
    } scat;
    // Probe format of the next weighted counter add (PayWeight: 4 bytes per probe, any weight; PayWeightSmall: 2.7, weights 0 .. 15 in the
    // group, others straight to the table): follows what pass 1 of the previous weighted batches counted -- k_tally_fold leaves
    // (weights outside 0 .. 15, batch number) on a pinned page, read without synchronisation.  One such weight keeps the wide format
    // for the next 64 batches.
    struct {
        volatile unsigned long long *pin = nullptr;
        unsigned long long issued = 0, seen = 0;
        uint32_t backoff = 0;
    } wt;
    // Update WINDOW (psk_window.hpp, round 4): small unit-weight add / remove batches of 16-byte keys into a big table wait here, in
    // arrival order, as key copies; the flush proves while it folds that every remove would have succeeded at its own position of the
    // stream (else: undo + batch-by-batch replay), so the result is the reference's for ANY stream -- no opt-in, no contract.
    struct WinBatch {
        uint64_t start, n;   // keys [start, start + n) of the list ...
        uint32_t remove;
        const void *ext;     // ... or n keys the caller keeps where they are until the window is applied (PSK_DEVICE_BORROWED); null: copied
    };
    struct {
        DevBuf keys;         // uint8[cap][16]
        uint64_t n = 0, cap = 0;   // keys waiting (copied + borrowed), the window's capacity
        uint64_t copied = 0;       // keys of the list in use
        std::vector<WinBatch> batches;
        uint32_t backoff = 0;      // windows left that are replayed batch by batch without trying the fold (after a failed proof)
        void *pin = nullptr;       // pinned staging of the phase table
        uint64_t folds = 0, replays = 0;   // statistics (psk_get_option "update_window_folds" / "_replays" report the globals)
    } win;
    DevBuf s_snap, s_wstat, s_phase;           // window fold: per-phase segment fill counts, per-part status, phase table
    PartGeom rm_g{};     // geometry of the validated remove's fast path between its optimistic decrement and a possible undo
    // Read-mostly CountingBloomFilter tables (round 3): the nibble-slice lookup reads the whole 32-bit table to build its 4-bit
    // images (1 GiB for BASELINE cfg 4).  When the table has not changed since the previous lookup the images are kept -- a linear
    // 4-bit saturating copy of the table, cells / 2 bytes -- and later lookups load them instead (k_nib_gather).  `table_version`
    // counts the entry points that may change the table (CHECK_HANDLE; the read-only ones use CHECK_HANDLE_RO), `built` is the
    // version the copy mirrors, `seen` the version of the previous nibble lookup; the copy is used on the stream that built it.
    uint64_t table_version = 0;
    struct {
        DevBuf img;
        uint64_t built = ~0ULL, seen = ~0ULL, words = 0;
        uint32_t seen_count = 0;  // nibble-eligible lookups in a row that found version `seen`
        hipStream_t stream = nullptr;
        bool allow = false;  // set by psk_cbf_check around its lookup
        // the table's pointer is in somebody's hands who has not said that every outside write will be announced: no kept images (they
        // would go stale silently).  Set by psk_table_info and for caller-owned tables; cleared by option "table_private" / psk_rescan_bound.
        bool exposed = false;
    } shadow;
    // split lookup (psk_bloom_check_begin / _finish): pass 1 of the first round has run, the rest waits for the table
    struct {
        bool active = false, scattered = false;
        Batch b{};            // device-resident batch (the caller keeps it alive until finish)
        PartGeom g{};
        uint64_t round_keys = 0;
        int scheme = 0;       // 0 keyed probes, 3 tile flags
        uint32_t gen = 0;     // tile flags: generation number of the scattered round
    } pend;
};

PSK_HIDDEN int ensure(DevBuf &b, uint64_t bytes);  // grow a scratch buffer

static inline int grid_for_keys(uint64_t n)  // direct kernels: 256 CUs x 16 blocks, grid-stride beyond that
{
    uint64_t g = (n + kBlock - 1) / kBlock;
    const uint64_t cap = 256ULL * 16;
    if (g > cap) g = cap;
    return (int)(g ? g : 1);
}


// ------------------------------------------------- partitioned (large-batch) path
// Tunables (psk_set_option): the partitioned path is taken when the batch has at least g_part_min_keys keys and
// the table geometry allows it; g_part_mode 0 = never, 1 = auto.
extern PSK_HIDDEN __thread int64_t g_part_min_keys;
extern PSK_HIDDEN int64_t g_part_mode, g_part_max_keys, g_part_cache_bytes, g_part_two_level_slices, g_part_debug;
extern PSK_HIDDEN __thread int64_t g_bloom_lookup;      // Bloom lookups: 0 keyed probes + miss stores, 1 return trip (psk_lookup.hpp), 2 (default) by the observed miss rate
extern PSK_HIDDEN int64_t g_part_slice_bias;     // bench knob: added to log2(cells per slice)
extern PSK_HIDDEN int64_t g_part_tile_threads;   // pass 1 workgroup shape for k <= 8: 0 = auto (launch_scatter), 512 / 1024 = forced
extern PSK_HIDDEN int64_t g_part_even_tiles;     // 1 (default): pass 1 evens the tile size out over the workgroups
extern PSK_HIDDEN int64_t g_lookup_half;           // 1 (default): counter lookups into 2^26 .. 2^27 counters use 2^16-counter slices of 16-bit values
extern PSK_HIDDEN int64_t g_small_weights;     // weighted CMS adds: 0 never the compact probe format, 1 by the hint, 2 always (tests)
extern PSK_HIDDEN __thread int64_t g_cbf_shadow;         // keep the nibble-slice lookup's images of an unchanged table
extern PSK_HIDDEN int64_t g_nib_nt;             // nontemporal table loads in the nibble-slice kernels (bench A/B)
extern PSK_HIDDEN int64_t g_nib_update_layout;  // delta-image layout of k_nib_apply: 0 pieces, 1 blocks (psk_nibble.hpp)
extern PSK_HIDDEN int64_t g_lookup_nibble, g_update_nibble;  // CBF tables beyond one level of 32-bit slices: 4-bit slice images (psk_nibble.hpp)
extern PSK_HIDDEN int64_t g_part_dense_groups;   // pass 2 walks a wave's segments end to end when a segment holds fewer groups than this on average (0 = never)
extern PSK_HIDDEN int64_t g_ragged_sort;         // option "ragged_sort": pass 1 hands keys of different lengths to its lanes in order of length (psk_partition.hpp sort_tile)
extern PSK_HIDDEN int64_t g_big_table_nt;        // option "big_table_nt": nontemporal sweeps of Bloom tables of 128 MiB and more (psk_partition.hpp slice_piece)
extern PSK_HIDDEN int64_t g_part_wgs;            // bench knob: pass 1 workgroups (0 = auto: one or two per CU)
extern PSK_HIDDEN int64_t g_lookup_split;        // bench knob: 0 = never share a slice between two pass-2 workgroups
extern PSK_HIDDEN int64_t g_lookup_collect_threads;  // pass 3 of the counter lookups: 1024 (two tiles in flight per CU) or 512 (four)
extern PSK_HIDDEN int64_t g_lookup_run_lanes;  // bench knob of the counter lookups (lanes per run in pass 3; 0 = auto)

// slices of a table of `cells` cells; max_shift = log2(cells one LDS slice may hold)
// target_lg: aim at 2^target_lg .. 2^(target_lg+1)-1 slices.  8 (one slice per CU or more) for the Bloom tables; the counter
// tables take 7: measured on MI355X (scripts/ab_slices.py, CMS 2^20 x 5) 160 slices of 2^15 counters beat 320 of 2^14 --
// pass 1 sorts into half as many bins with runs twice as long (weighted add 211 -> 193 us, lookups 314 -> 302 us), which
// outweighs pass 2 running on 160 of the 256 CUs.
static inline bool part_slices(uint64_t cells, uint32_t max_shift, uint32_t min_shift, PartGeom *g,
                               uint64_t max_buckets = kPartMaxBuckets, int target_lg = 8)
{
    if (cells >= (1ULL << 32) || cells < (1ULL << 16)) return false;  // cell index 0xFFFFFFFF is the pad marker
    const uint32_t lg = 63 - __builtin_clzll(cells);  // floor(log2 cells)
    int shift = (int)lg - target_lg + (int)g_part_slice_bias;  // (bias: bench knob)
    if (shift > (int)max_shift) shift = max_shift;
    if (shift < (int)min_shift) shift = min_shift;
    const uint64_t B = (cells + (1ULL << shift) - 1) >> shift;
    if (B > max_buckets) return false;
    g->nbuckets = (uint32_t)B;
    g->shift = (uint32_t)shift;
    g->dbg = (uint32_t)g_part_debug | (g_big_table_nt != 0 ? kGeomNtBit : 0u) | (g_ragged_sort == 0 ? kGeomNoSortBit : 0u);
    g->split = g->split_idx = 0;
    g->dense = 0;
    g->append = 0;
    return true;
}

// Geometry of the 4-bit slice images (psk_nibble.hpp): slices of 2^15 .. 2^18 counters, at least 256 of them when the table allows
// (2^18 from 2^26 counters on: 1024 slices for BASELINE cfg 4's 2^28).  update = false: lookups (tables of 2^nibble_min_lg_lookup
// counters and more), true: unit adds / decrements / write-combining segments (more than 2^nibble_min_lg_update counters).
extern PSK_HIDDEN int64_t g_nib_min_lg_lookup, g_nib_min_lg_update;
static inline bool nib_geometry(uint64_t cells, bool update, PartGeom *g)
{
    if (update ? cells <= (1ULL << g_nib_min_lg_update) : cells < (1ULL << g_nib_min_lg_lookup)) return false;
    return part_slices(cells, kNibShift, 15, g, kPartMaxBuckets, 8);
}

// workgroups per slice of k_nib_apply (log2): 2 for the 2^18-counter slices (two 64 KiB delta images per CU: one streams probes while the
// other folds -- psk_nibble.hpp nib_apply_list); option "nibble_update_parts": 0 = this rule, 1 / 2 = forced (bench A/B)
extern PSK_HIDDEN int64_t g_nib_update_parts;
static inline uint32_t nib_update_lgparts(const PartGeom &g)
{
    if (g_nib_update_parts > 0) return g_nib_update_parts >= 2 && g.shift >= 16 ? 1u : 0u;
    return g.shift >= 18 ? 1u : 0u;
}

// A 4-bit DELTA image holds at most 15 hits per counter and round; a round that brings more than ~2.5 probes per counter on average
// overflows some counter of nearly every slice and would run at the atomics' rate (measured: 10 M keys into 1.7e7 counters, 4.2 probes
// per counter: 1.8 ms against 0.31 ms through the 32-bit slices).  At 2.5 the tail P(Poisson >= 16) is ~3e-9 per counter.
static inline bool nib_load_ok(uint64_t n, uint32_t k, uint64_t cells) { return n * (uint64_t)k * 2 <= cells * 5; }

// The fixed geometry of a handle's persistent lists for `cap` keys per list (psk_sketch::scat); false: the table is not eligible.
static inline bool scat_geometry(const psk_sketch *s, uint64_t cap, PartGeom *g)
{
    if (!nib_geometry(s->m, true, g)) return false;
    g->k = s->k;
    g->nwg = 256;
    g->append = 1;
    g->tile = 0;
    const double tile_keys = 2048.0 * 7.0 / (double)(s->k < 1 ? 1 : s->k);             // keys per pass-1 tile, roughly (14 K probes)
    const double mean = (double)cap * s->k / ((double)g->nbuckets * g->nwg);           // probes per segment when the list is full
    const double runs = 2.0 * (double)cap / (tile_keys * g->nwg) + 64.0;               // (tile, slice) runs per segment: small batches bring short tiles
    const double segcap = mean / 6.0 + 0.5 * runs + 8.0 * __builtin_sqrt(mean) / 6.0 + 16.0;
    if (segcap >= (double)(1u << 24) || (double)g->nbuckets * segcap >= 4294967296.0) return false;
    g->segcap = (uint32_t)segcap;
    return true;
}

// raise a kernel's dynamic-LDS limit (needed above 64 KiB); remembered per (kernel, device) so the driver call is
// paid once, not on every launch
PSK_HIDDEN int raise_dyn_lds(const void *kernel, size_t bytes);

template <class K>
static int set_dyn_lds(K kernel, size_t bytes)
{
    return raise_dyn_lds((const void *)kernel, bytes);
}

// Pass 1 for one concrete (Src, IdxFn, Pay, Spill, KT, NT): sizes the (slice, workgroup) segments for `n` keys,
// grows the handle's bucket buffer, launches.  g->nwg / g->segcap are filled in for pass 2.
template <class Pay, int KT, int NT>
static size_t scatter_lds_bytes(const PartGeom *g, uint64_t tile = 0, bool sorted = false)  // tile: keys per tile (0 = the shape's full tile)
{
    if (sorted) return scatter_lds_bytes<Pay, KT, NT>(g, tile, false) + 16 + 8 * (size_t)kSortBins + 18 * (size_t)PartTile<Pay, KT, NT>::TILE;  // length sort (src_sorted): counts, descriptors, slots
    using Tile = PartTile<Pay, KT, NT>;
    const uint32_t kk = g->k < (uint32_t)KT ? g->k : (uint32_t)KT;
    const size_t tk = tile ? (size_t)tile : (size_t)Tile::TILE;
    // stage (one word per probe in every mode) + one slice id per group for the payload modes
    const size_t stage_cap = ((tk * kk + (size_t)(Tile::GS - 1) * g->nbuckets) + 3) & ~(size_t)3;
    const size_t stage_words = stage_cap + (Tile::pair ? stage_cap / Tile::GS + 4 : 0);
    return (5 * (size_t)g->nbuckets + 16 + 24 + stage_words) * 4;
}

// append mode (PartGeom::append): the caller's persistent segments; g->nwg / g->segcap are the caller's and stay as they are
struct ScatterTarget {
    uint32_t *cnt;
    uint4 *part;
};

#ifndef PSK_LOOKUP_FAT_SLICES
#define PSK_LOOKUP_FAT_SLICES 900
#endif
constexpr uint32_t kLookupFatSlices = PSK_LOOKUP_FAT_SLICES;  // return-trip lookups: 4096-key pass-1 tiles from this many slices on
constexpr size_t kScatterLdsBudget = 160 * 1024;
constexpr size_t kScatterLdsTwoPerCu = 78 * 1024;  // two workgroups' dynamic LDS per CU

template <class Src>
struct src_fat512 { static constexpr bool value = std::is_same<Src, KeysFixed16>::value || std::is_same<Src, KeysFixed16Multi>::value || std::is_same<Src, KeysFixed8>::value; };

// ---- pass 1 through fixed-capacity bins (psk_part_bins.hpp, round 6): which (layout, payload, k) may take it, and its geometry
// option "pass1_bins": 1 (default) = wherever eligible, 0 = k_part_scatter everywhere (A/B, tests)
extern PSK_HIDDEN int64_t g_part_bins;
template <class Src, class Pay, int KT>
struct bins_eligible {
    static constexpr bool value = pay_bins_ok<Pay>::value && KT <= 8 && src_fat512<Src>::value;  // (the 16- and 8-byte layouts)
};
constexpr size_t kBinSlackWords = 8 + 16;  // behind the last bin: what the write-out's last group may read past a bin's end + the bench build's phase-profile slots
struct BinsPlan {
    uint32_t tile, cap, stride, per_cu;
    size_t lds;
};
// false: the geometry does not fit (too many slices for the write-out's lanes, or a bin table beyond the LDS)
template <int KT>
static bool bins_plan(const PartGeom *g, BinsPlan *p)
{
    if (g_part_bins == 0 || g->nbuckets > (uint32_t)kBinThreads * kBinSlicesPerLane || g->nbuckets < 2) return false;
    constexpr double kSigmas = 3.5;  // bin capacity = mean + 3.5 sigma of a tile's load: ~1e-3 of the bins of a tile fill up (2.8 measured the same)
    const uint32_t kk = g->k < (uint32_t)KT ? g->k : (uint32_t)KT;
    p->tile = (uint32_t)kBinThreads * kBinKpt;
    const double mean = (double)p->tile * kk / (double)g->nbuckets;
    uint32_t cap = (uint32_t)(mean + kSigmas * __builtin_sqrt(mean) + 2.0);
    cap = (cap + 5) / 6 * 6;            // whole groups
    p->cap = cap;
    // words between bins: 2 in front of the probes (word 1 = the counter), the probes, the slot a full bin's stores land on -- rounded up to
    // 2 (mod 4), so that neighbouring bins start two banks apart (slot r of ALL bins is what the lanes of a wave write at about the same time)
    uint32_t stride = kBinHead + cap + 1;
    while (stride % 4 != 2) ++stride;
    p->stride = stride;
    p->lds = ((size_t)g->nbuckets * stride + kBinSlackWords) * 4;
    if (p->lds > kScatterLdsBudget / 2) return false;   // (at least two workgroups per CU, or the old shape does better)
    uint32_t per_cu = (uint32_t)(kScatterLdsBudget / p->lds);
    const uint32_t by_waves = 2048u / (uint32_t)kBinThreads;   // (32 wave slots per CU)
    p->per_cu = per_cu > by_waves ? by_waves : per_cu;
    return true;
}

template <class Src, class IdxFn, class Pay, class Spill, int KT>
static int launch_scatter_bins(psk_sketch *s, const Src &src, const IdxFn &idxfn, const Pay &pay, const Spill &spill, PartGeom *g, uint64_t n,
                               hipStream_t st, uint32_t want_wgs, const BinsPlan &bp)
{
    const uint32_t kk = g->k < (uint32_t)KT ? g->k : (uint32_t)KT;
    const uint64_t tile_full = bp.tile;
    const uint64_t ntiles = (n + tile_full - 1) / tile_full;
    // (32-bit key indices inside the kernel: n < 2^31, the caller checked; the prefetch of the tile behind the last one stays below 2^32)
    uint64_t nwg = 256ULL * bp.per_cu;
    if (want_wgs) nwg = want_wgs;
    if (g_part_wgs > 0) nwg = (uint64_t)g_part_wgs;
    if (nwg > 64u * kApplyWaves) nwg = 64u * kApplyWaves;  // pass 2: a wave walks at most one segment per lane (for_each_batch_at)
    if (nwg > ntiles) nwg = ntiles;
    if (nwg == 0) nwg = 1;
    const uint64_t tiles_per_wg = (ntiles + nwg - 1) / nwg;
    uint64_t tk = tile_full;
    if (g_part_even_tiles != 0 && nwg * tiles_per_wg > ntiles) {  // (see launch_scatter_nt)
        tk = ((n + nwg * tiles_per_wg - 1) / (nwg * tiles_per_wg) + 63) & ~63ULL;
        if (tk > tile_full) tk = tile_full;
    }
    const double mean = (double)tiles_per_wg * (double)tk * kk / (double)g->nbuckets;  // probes per segment
    uint64_t segcap = (uint64_t)(mean / 6.0 + 0.5 * (double)tiles_per_wg + 8.0 * __builtin_sqrt(mean) / 6.0 + 16.0);
    if constexpr (pay_tile_tag<Pay>::value) {
        if (tiles_per_wg > 16) return fail(PSK_EINVAL, "lookup round of %llu keys needs %llu tiles per workgroup (max 16)",
                                           (unsigned long long)n, (unsigned long long)tiles_per_wg);
    }
    if (segcap >= (1u << 24) || (uint64_t)g->nbuckets * segcap >= (1ULL << 32))
        return fail(PSK_EINVAL, "partition round of %llu keys is too large (segments of %llu groups)", (unsigned long long)n, (unsigned long long)segcap);
    g->nwg = (uint32_t)nwg;
    g->segcap = (uint32_t)segcap;
    g->tile = (uint32_t)tk;
    g->dense = (mean / 6.0 + 0.5 * (double)tiles_per_wg) < (double)g_part_dense_groups ? 1u : 0u;
    PSK_TRY(ensure(s->s_part, (uint64_t)g->nbuckets * nwg * segcap * 16 + 256));
    PSK_TRY(ensure(s->s_cnt, (uint64_t)g->nbuckets * nwg * 4 + 128));
    auto kern = k_part_bins<Src, IdxFn, Pay, Spill, KT, kBinKpt>;
    PSK_TRY(set_dyn_lds(kern, bp.lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(kBinThreads), bp.lds, st, src, idxfn, pay, spill, *g, (uint32_t)n, bp.cap, bp.stride,
                       (uint32_t *)s->s_cnt.p, (uint4 *)s->s_part.p);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}
template <class Src, class IdxFn, class Pay, class Spill, int KT, int NT>
static int launch_scatter_nt(psk_sketch *s, const Src &src, const IdxFn &idxfn, const Pay &pay, const Spill &spill, PartGeom *g,
                             uint64_t n, hipStream_t st, uint32_t want_wgs, const ScatterTarget *fixed = nullptr)
{
    if constexpr (bins_eligible<Src, Pay, KT>::value) {  // pass 1 through fixed-capacity bins where the geometry admits it (psk_part_bins.hpp)
        BinsPlan bp;
        if (!fixed && n < (1ULL << 31) && bins_plan<KT>(g, &bp)) return launch_scatter_bins<Src, IdxFn, Pay, Spill, KT>(s, src, idxfn, pay, spill, g, n, st, want_wgs, bp);
    }
    using Tile = PartTile<Pay, KT, NT>;
    constexpr bool kSorted = src_sorted<Src>::value;  // (keys of different lengths: + the LDS of the tile's length sort)
    const uint32_t kk = g->k < (uint32_t)KT ? g->k : (uint32_t)KT;
    // keys per tile: the shape's, cut down (whole waves) where its LDS stage would not fit -- 8-probe groups at 2048 slices carry 7 pad
    // slots per slice: 1984-key tiles instead of 2048 (the kernel sizes its stage by PartGeom::tile)
    uint64_t tile_full = Tile::TILE;
    // lookups with the 4096-key shape (PayBloomLookup): only where the slices are many (pass 3's run copies halve); 2048 keys below that
    if constexpr (pay_is_lookup<Pay>::value && pay_fat1024<Pay>::value) {
        if (g->nbuckets < kLookupFatSlices && tile_full > 2048) tile_full = 2048;
    }
    while (tile_full > 64 && scatter_lds_bytes<Pay, KT, NT>(g, tile_full, kSorted) > kScatterLdsBudget) tile_full -= 64;
    if (scatter_lds_bytes<Pay, KT, NT>(g, tile_full, kSorted) > kScatterLdsBudget) return fail(PSK_EINVAL, "pass 1: %u slices do not fit the LDS stage", g->nbuckets);
    const uint64_t ntiles = (n + tile_full - 1) / tile_full;
    if (fixed) {  // append behind what the segments already hold: the first min(nwg, tiles) workgroups each take their share of tiles
        if (Pay::mode != kModePlain) return fail(PSK_EINVAL, "persistent segments carry payload-free probes");
        const uint64_t grid = g->nwg < ntiles ? g->nwg : ntiles;
        if (grid == 0) return PSK_OK;
        const uint64_t per_wg = (ntiles + grid - 1) / grid;
        uint64_t tk = tile_full;
        if (g_part_even_tiles != 0 && grid * per_wg > ntiles) {
            tk = ((n + grid * per_wg - 1) / (grid * per_wg) + 63) & ~63ULL;
            if (tk > tile_full) tk = tile_full;
        }
        g->tile = (uint32_t)tk;
        const size_t lds = scatter_lds_bytes<Pay, KT, NT>(g, tk, kSorted);
        auto kern = k_part_scatter<Src, IdxFn, Pay, Spill, KT, NT>;
        PSK_TRY(set_dyn_lds(kern, lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, st, src, idxfn, pay, spill, *g, n, fixed->cnt, fixed->part);
        HIP_TRY(hipGetLastError());
        return PSK_OK;
    }
    uint64_t per_cu = NT > 512 ? 1 : (scatter_lds_bytes<Pay, KT, NT>(g, tile_full, kSorted) > kScatterLdsTwoPerCu ? 1 : 2);
    if (kBenchKnobs && (g->dbg & 8)) per_cu = 1;  // ablation: one workgroup per CU
    uint64_t nwg = 256 * per_cu;
    if (want_wgs) nwg = want_wgs;  // caller's choice (keyed lookups into big tables: twice the keys per round)
    if (g_part_wgs > 0) nwg = (uint64_t)g_part_wgs;
    if (nwg > 64u * kApplyWaves) nwg = 64u * kApplyWaves;  // pass 2: a wave walks at most one segment per lane (for_each_batch_at)
    if (nwg > ntiles) nwg = ntiles;
    const uint64_t tiles_per_wg = (ntiles + nwg - 1) / nwg;
    // Even tiles: 10 M keys are 4883 tiles of 2048, i.e. 19 rounds of all 256 workgroups and a 20th of only 19 of them --
    // every workgroup takes ceil(tiles / workgroups) tiles of the same, slightly smaller size instead (a multiple of 64 keys:
    // whole waves), the last one short.  Pass 2 and the lookups' pass 3 read the tile size from the geometry.
    uint64_t tk = tile_full;
    if (g_part_even_tiles != 0 && nwg * tiles_per_wg > ntiles) {
        tk = ((n + nwg * tiles_per_wg - 1) / (nwg * tiles_per_wg) + 63) & ~63ULL;
        if (tk > tile_full) tk = tile_full;
    }
    const size_t lds = scatter_lds_bytes<Pay, KT, NT>(g, tk, kSorted);
    const double mean = (double)tiles_per_wg * (double)tk * kk / (double)g->nbuckets;  // probes per segment
    // 16-byte groups per segment: mean/GS, + ~half a group of padding per (tile, slice) run, + 8 sigma
    uint64_t segcap = (uint64_t)(mean / Tile::GS + 0.5 * (double)tiles_per_wg + 8.0 * __builtin_sqrt(mean) / Tile::GS + 16.0);
    if constexpr (Pay::mode == kModeKeyed || pay_tile_tag<Pay>::value) {
        if (tiles_per_wg > 16) return fail(PSK_EINVAL, "lookup round of %llu keys needs %llu tiles per workgroup (max 16)",
                                           (unsigned long long)n, (unsigned long long)tiles_per_wg);
    }
    // (pass 1 addresses a group as wg_base + slice * segcap + slot with a 24-bit multiply and a 32-bit sum)
    if (segcap >= (1u << 24) || (uint64_t)g->nbuckets * segcap >= (1ULL << 32))
        return fail(PSK_EINVAL, "partition round of %llu keys is too large (segments of %llu groups)", (unsigned long long)n, (unsigned long long)segcap);
    g->nwg = (uint32_t)nwg;
    g->segcap = (uint32_t)segcap;
    g->tile = (uint32_t)tk;
    // pass 2's walk (for_each_batch_at): segments much shorter than a 64-lane load are walked end to end
    g->dense = (mean / Tile::GS + 0.5 * (double)tiles_per_wg) < (double)g_part_dense_groups ? 1u : 0u;
    PSK_TRY(ensure(s->s_part, (uint64_t)g->nbuckets * nwg * segcap * 16 + 256));
    PSK_TRY(ensure(s->s_cnt, (uint64_t)g->nbuckets * nwg * 4 + 128));  // + 12 x u64 of phase profile (dbg & 32)
    auto kern = k_part_scatter<Src, IdxFn, Pay, Spill, KT, NT>;
    PSK_TRY(set_dyn_lds(kern, lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NT), lds, st, src, idxfn, pay, spill, *g, n, (uint32_t *)s->s_cnt.p,
                       (uint4 *)s->s_part.p);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

// The workgroup shape launch_scatter picks for (Pay, KT) on geometry g: threads per workgroup, and the pass-1 workgroups
// launch_scatter_nt starts by default (before it clamps them to the number of tiles)
// the two-per-CU twin (Pay::fat512) exists for the 16-byte fast layouts only: the other layouts take the one 1024-thread shape (keys of
// different lengths hold a descriptor and a window per key in registers -- four keys per thread spill --, and the rare layouts are not worth
// a second copy of every kernel)

template <class Pay, int KT, bool FAT = true>
static int scatter_threads(const PartGeom *g)
{
    if constexpr (!FAT && (pay_fat512<Pay>::value || pay_fat1024<Pay>::value)) return scatter_threads<PaySlim<Pay>, KT, false>(g);
    // k <= 8 without the two-per-CU shape (every payload but the Bloom insert's / the tile-flag lookup's): 1024 threads, always -- its stage fits the
    // LDS for every geometry the single-level path takes (at most 2048 slices, at most 16 K probes per tile: <= 154 KB), so the 512-thread
    // form of these kernels is not even instantiated (round 5: it was a third of the library's device code and never selected)
    if constexpr (KT <= 8 && !(pay_fat512<Pay>::value && FAT)) return 1024;
    if constexpr (KT <= 8) {
        // (a 4096-key tile is cut to what fits -- launch_scatter_nt --: the shape is taken as long as 2048 keys do)
        const bool fits1024 = scatter_lds_bytes<Pay, KT, 1024>(g, PartTile<Pay, KT, 1024>::TILE > 2048 ? 2048 : 0) <= kScatterLdsBudget;
        const bool two_per_cu = pay_fat512<Pay>::value && scatter_lds_bytes<Pay, KT, kPartThreads>(g) <= kScatterLdsTwoPerCu;
        bool use1024 = fits1024 && !two_per_cu;
        if (g_part_tile_threads == 1024) use1024 = fits1024;
        if (g_part_tile_threads == 512 || (kBenchKnobs && (g->dbg & 16))) use1024 = false;
        if (use1024) return 1024;
    }
    return kPartThreads;
}
// largest round (keys) whose tiles number at most `max_tiles` per pass-1 workgroup (probes that spell the tile's ordinal in 4 bits)
template <class Pay, int KT, bool FAT = true, bool SORTED = false>
static uint64_t scatter_round_cap(const PartGeom *g, uint32_t want_wgs, uint32_t max_tiles)
{
    if constexpr (!FAT && (pay_fat512<Pay>::value || pay_fat1024<Pay>::value)) return scatter_round_cap<PaySlim<Pay>, KT, false, SORTED>(g, want_wgs, max_tiles);
    if constexpr (pay_bins_ok<Pay>::value && KT <= 8 && FAT) {  // (FAT = src_fat512: the layouts whose pass 1 goes through the bins, launch_scatter_bins)
        BinsPlan bp;
        if (bins_plan<KT>(g, &bp)) {
            uint64_t nwg = 256ULL * bp.per_cu;
            if (want_wgs) nwg = want_wgs;
            if (g_part_wgs > 0) nwg = (uint64_t)g_part_wgs;
            if (nwg > 64u * kApplyWaves) nwg = 64u * kApplyWaves;
            return nwg * max_tiles * bp.tile;
        }
    }
    const bool big = scatter_threads<Pay, KT, FAT>(g) == 1024;
    const size_t lds = big ? scatter_lds_bytes<Pay, KT, 1024>(g) : scatter_lds_bytes<Pay, KT, kPartThreads>(g);
    uint64_t tile = big ? (uint64_t)PartTile<Pay, KT, 1024>::TILE : (uint64_t)PartTile<Pay, KT, kPartThreads>::TILE;
    // (the tile launch_scatter_nt will really use: cut where the LDS stage -- with the length sort's share for ragged keys -- would not fit)
    if (big) while (tile > 64 && scatter_lds_bytes<Pay, KT, 1024>(g, tile, SORTED) > kScatterLdsBudget) tile -= 64;
    else while (tile > 64 && scatter_lds_bytes<Pay, KT, kPartThreads>(g, tile, SORTED) > kScatterLdsBudget) tile -= 64;
    uint64_t nwg = 256 * (uint64_t)(big ? 1 : (lds > kScatterLdsTwoPerCu ? 1 : 2));
    if (kBenchKnobs && (g->dbg & 8)) nwg = 256;
    if (want_wgs) nwg = want_wgs;
    if (g_part_wgs > 0) nwg = (uint64_t)g_part_wgs;
    if (nwg > 64u * kApplyWaves) nwg = 64u * kApplyWaves;
    return nwg * max_tiles * tile;
}

// Workgroup shape (PartTile): k <= 8 takes one 1024-thread workgroup per CU; the Bloom insert (Pay::fat512) two 512-thread
// workgroups per CU with 32 probes per thread when two LDS stages fit (tables of up to ~512 slices); larger k runs 512 threads
// with one key per thread.
// Option "tile_threads": 0 = this rule, 512 / 1024 = force the shape (A/B).
template <class Src, class IdxFn, class Pay, class Spill, int KT>
static int launch_scatter(psk_sketch *s, const Src &src, const IdxFn &idxfn, const Pay &pay, const Spill &spill, PartGeom *g,
                          uint64_t n, hipStream_t st, uint32_t want_wgs = 0, const ScatterTarget *fixed = nullptr)
{
    // (the fat shapes -- 32 probes per thread -- are for the 16- and 8-byte layouts: every other layout runs the payload's plain twin)
    if constexpr (!src_fat512<Src>::value && (pay_fat512<Pay>::value || pay_fat1024<Pay>::value))
        return launch_scatter<Src, IdxFn, PaySlim<Pay>, Spill, KT>(s, src, idxfn, PaySlim<Pay>(pay), spill, g, n, st, want_wgs, fixed);
    if constexpr (Pay::mode == kModeKeyed)
        static_assert(((uint64_t)PartTile<Pay, KT, kPartThreads>::TILE << Pay::slice_shift) <= (1ULL << 31),
                      "512-thread tiles must keep keyed probes inside 31 bits for the largest slice");
    if constexpr (KT <= 8) {
        // keyed probes carry (key index in tile << shift | bit in slice) in 31 bits (the top bit spells the tile ordinal): the
        // tile must stay within 2^(31 - shift) keys (PayKeyId::max_tile caps it at 2048 keys)
        if constexpr (Pay::mode == kModeKeyed)
            static_assert(((uint64_t)PartTile<Pay, KT, 1024>::TILE << Pay::slice_shift) <= (1ULL << 31), "keyed tile too large");
        if constexpr (!(pay_fat512<Pay>::value && src_fat512<Src>::value)) {  // one shape only (scatter_threads; launch_scatter_nt cuts the tile where the stage would not fit)
            return launch_scatter_nt<Src, IdxFn, Pay, Spill, KT, 1024>(s, src, idxfn, pay, spill, g, n, st, want_wgs, fixed);
        } else {
            if (scatter_threads<Pay, KT, true>(g) == 1024) return launch_scatter_nt<Src, IdxFn, Pay, Spill, KT, 1024>(s, src, idxfn, pay, spill, g, n, st, want_wgs, fixed);
            return launch_scatter_nt<Src, IdxFn, Pay, Spill, KT, kPartThreads>(s, src, idxfn, pay, spill, g, n, st, want_wgs, fixed);
        }
    } else {
        return launch_scatter_nt<Src, IdxFn, Pay, Spill, KT, kPartThreads>(s, src, idxfn, pay, spill, g, n, st, want_wgs, fixed);
    }
}

// compile-time hash count: exact for the common small k on the 16-byte fast layout, rounded up otherwise
template <class Src, class F>
static int with_kt(uint32_t k, F &&f)
{
    // (`if constexpr`: a plain `if` instantiated -- and emitted -- the exact-k kernels of EVERY layout, though the rare ones could never be
    // selected: 6 dead kernels per layout, payload and launcher.  Ragged byte keys -- the reference's native key type -- get the exact sizes too.)
    // The 16-byte layouts get every common k exactly; 8-byte keys and ragged byte keys (the reference's native key type) the two that
    // matter most -- 7 (fpr 0.01) and 5 (the CountMinSketch's default depth) --; everything else the round-ups, whose chains go four at a
    // time and stop at k.  (Exact sizes for all three layouts were 60 % of the library's device code and of its build time.)
    constexpr bool fast16 = std::is_same<Src, KeysFixed16>::value || std::is_same<Src, KeysFixed16Multi>::value;
    constexpr bool fast2 = std::is_same<Src, KeysFixed8>::value || std::is_same<Src, KeysFixed32>::value || std::is_same<Src, KeysVarlen<uint8_t>>::value;
    if constexpr (fast16) {
        switch (k) {
            case 3: return f(std::integral_constant<int, 3>{});
            case 4: return f(std::integral_constant<int, 4>{});
            case 5: return f(std::integral_constant<int, 5>{});
            case 6: return f(std::integral_constant<int, 6>{});
            case 7: return f(std::integral_constant<int, 7>{});
            case 10: return f(std::integral_constant<int, 10>{});
            default: break;
        }
    } else if constexpr (fast2) {
        switch (k) {
            case 5: return f(std::integral_constant<int, 5>{});
            case 7: return f(std::integral_constant<int, 7>{});
            default: break;
        }
    }
    if (k <= 8) return f(std::integral_constant<int, 8>{});
    // (k = 9 .. 16 on the other layouts runs the 32-chain instantiation, whose chains go four at a time and stop at k: both forms hash one key
    // per thread and tile, so a separate 16-chain kernel per layout and payload bought nothing but device code)
    if constexpr (fast16) {
        if (k <= 16) return f(std::integral_constant<int, 16>{});
    }
    return f(std::integral_constant<int, 32>{});
}

// sources the partitioned path is instantiated for (the rest use the direct kernels)
template <class F>
static int with_part_source(const Batch &b, bool *handled, F &&f)
{
    *handled = true;
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
        default: break;
    }
    *handled = false;
    return PSK_OK;
}

// `scale`: measured crossover vs the direct kernels (scripts/crossover.py, m = 2^28 / 2^20 x 5): Bloom insert wins
// from ~64 K keys (scale 1), Bloom lookups and counter adds from ~256 K keys (scale 4)
static inline bool part_wanted(uint64_t n, uint32_t k, int64_t scale = 1)
{
    return g_part_mode != 0 && (int64_t)n >= g_part_min_keys * scale && k <= 32;
}

// Option "scratch_budget_bytes" (0 = none): caps the partition scratch of a handle by cutting a batch into more rounds.  per_key:
// scratch bytes one key of a round occupies (bucket buffer incl. padding and slack, plus values / perm for lookups).
extern PSK_HIDDEN __thread int64_t g_scratch_budget;
static inline uint64_t cap_round_by_budget(uint64_t rk, double per_key)
{
    if (g_scratch_budget <= 0 || per_key <= 0) return rk;
    uint64_t cap = (uint64_t)((double)g_scratch_budget / per_key);
    if (cap < (1u << 18)) cap = 1u << 18;  // (below ~256 K keys the per-round fixed costs dominate: the floor of the cap)
    return rk < cap ? rk : cap;
}

// Keys per partition round.  Pass 2 reads back what pass 1 has just written: while a round's bucket buffer fits the
// 256 MB Infinity Cache (MALL) most of that read never reaches HBM (measured, 10 M lookups: 475 MB in one round
// 321 us, two rounds of 237 MB 271 us; inserts at 300 MB are still best in one round).  So a batch whose buffer
// would exceed 1.5 x `partition_cache_bytes` is cut into equal rounds of at most that size.
// group = probes per 16-byte group of the encoding in use.
static inline uint64_t part_round_keys(uint64_t n, uint32_t k, int group)
{
    uint64_t rk = (uint64_t)g_part_max_keys < n ? (uint64_t)g_part_max_keys : n;
    if (g_part_cache_bytes > 0 && n) {
        const double per_key = (double)k * 16.0 / group * 1.3;  // + run padding and partly filled 64-byte lines
        const double total = per_key * (double)n;
        if (total > 1.5 * (double)g_part_cache_bytes) {
            const uint64_t rounds = (uint64_t)(total / (double)g_part_cache_bytes) + 1;
            uint64_t per = ((n + rounds - 1) / rounds + 4095) & ~4095ULL;
            if (per < 1u << 20) per = 1u << 20;
            if (per < rk) rk = per;
        }
    }
    rk = cap_round_by_budget(rk, (double)k * 16.0 / group * 1.5);
    return rk ? rk : 1;
}

// Big tables (>= 64 MiB): pass 2 reads (lookups) or read-modify-writes (updates) the WHOLE table once per round, which costs
// more than what a cache-sized bucket buffer saves -- rounds as large as `partition_max_keys` allows
static inline uint64_t part_round_keys_big_table(uint64_t n, uint32_t k, int group, uint64_t table_bytes)
{
    const uint64_t rk = part_round_keys(n, k, group);
    if (table_bytes < (64ULL << 20)) return rk;
    uint64_t big = (uint64_t)g_part_max_keys < n ? (uint64_t)g_part_max_keys : n;
    big = cap_round_by_budget(big, (double)k * 16.0 / group * 1.5);
    return big > rk ? big : rk;
}

// Rounds of the two-level path: every round ends in a fold that read-modify-writes the WHOLE table (0.5 ms for 1 GiB),
// which dwarfs what a cache-sized bucket buffer saves -- as few rounds as `partition_max_keys` allows
static inline uint64_t part_round_keys_two_level(uint64_t n, uint32_t k = 7)
{
    uint64_t rk = (uint64_t)g_part_max_keys < n ? (uint64_t)g_part_max_keys : n;
    rk = cap_round_by_budget(rk, (double)k * (4.0 + 4.0) * 1.5);  // two bucket buffers (level 1: 4 B per probe, level 2: up to 4)
    return rk ? rk : 1;
}

// Two-level path (k_part_scatter by coarse bucket, then k_part_split by slice) for tables cut into more than
// `partition_two_level_slices` slices.  Fills level 1 (g1: coarse buckets) from the final geometry g2; the caller runs
// launch_scatter(..., g1, ...) with an inline-mode payload, then split_level2<OUT>(), then pass 2 on s_part2 / s_cnt2.
constexpr uint32_t kSplitParts = 16;  // level-2 segments per slice = waves of a pass-2 workgroup
static inline bool two_level_geometry(const PartGeom &g2, PartGeom *g1, uint32_t *sub_bits)
{
    if (g_part_two_level_slices <= 0 || (int64_t)g2.nbuckets <= g_part_two_level_slices) return false;
    uint32_t sb = 1;
    while (((g2.nbuckets + (1u << sb) - 1) >> sb) > 256) ++sb;
    if ((1u << sb) > (uint32_t)kSplitMaxSub) return false;
    *g1 = g2;
    g1->nbuckets = (g2.nbuckets + (1u << sb) - 1) >> sb;
    g1->shift = g2.shift + sb;
    *sub_bits = sb;
    return g1->shift <= 31;
}

template <int OUT, class Spill>
static int split_level2(psk_sketch *s, const PartGeom &g1, PartGeom *g2, uint32_t sub_bits, uint64_t probes, const Spill &spill,
                        hipStream_t st)
{
    constexpr int GS = OUT == 0 ? 6 : (OUT == 1 ? 8 : 4);
    const uint32_t P = kSplitParts;
    const uint32_t per = (g1.nwg + P - 1) / P;
    if (per > (uint32_t)kSplitMaxSegs) return fail(PSK_EINVAL, "two-level split: too many level-1 segments per part");
    const double mean = (double)probes / ((double)g2->nbuckets * P);  // probes per (slice, part) segment
    // groups: mean/GS + half a group of padding per split tile that touches the slice + 8 sigma
    const double tiles = (double)probes / ((double)g1.nbuckets * P) / (kSplitThreads * 3.0) + 1.0;
    g2->nwg = P;
    g2->segcap = (uint32_t)(mean / GS + tiles + 8.0 * __builtin_sqrt(mean) / GS + 16.0);
    PSK_TRY(ensure(s->s_part2, (uint64_t)g2->nbuckets * P * g2->segcap * 16 + 256));
    PSK_TRY(ensure(s->s_cnt2, (uint64_t)g2->nbuckets * P * 4 + 64));
    hipLaunchKernelGGL((k_part_split<OUT, Spill>), dim3(P, g1.nbuckets), dim3(kSplitThreads), 0, st, g1, (const uint32_t *)s->s_cnt.p,
                       (const uint4 *)s->s_part.p, *g2, sub_bits, (uint32_t *)s->s_cnt2.p, (uint4 *)s->s_part2.p, spill);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

// view of keys [start, start+cnt) of a device batch
static inline Batch sub_batch(const Batch &b, uint64_t start, uint64_t cnt)
{
    Batch sub = b;
    sub.n = cnt;
    if (b.layout == PSK_KEYS_VARLEN8 || b.layout == PSK_KEYS_VARLEN32) sub.offs = b.offs + start;
    else sub.data = (const uint8_t *)b.data + start * (uint64_t)b.key_len * (b.layout == PSK_KEYS_HASHES ? 8 : 1);
    return sub;
}

// The launchers (one translation unit each, and each of those built TWICE: -DPSK_TU_POW2=1 holds the instantiations for
// power-of-two tables, =0 the Barrett ones -- the ~1000 k_part_scatter instantiations then build in parallel and the
// critical path of a full build halves); *done = false when the batch / table is not eligible.
// psk_part_dispatch.hip picks the variant by s->pow2.
#ifdef PSK_TU_POW2
constexpr bool kTuPow2 = PSK_TU_POW2 != 0;
#define PSK_VARIANT_CAT2(name, v) name##_v##v
#define PSK_VARIANT_CAT(name, v) PSK_VARIANT_CAT2(name, v)
#define PSK_VARIANT(name) PSK_VARIANT_CAT(name, PSK_TU_POW2)
#endif
#define PSK_DECLARE_VARIANTS(ret, name, args) \
    PSK_HIDDEN ret name args;                 \
    PSK_HIDDEN ret name##_v0 args;            \
    PSK_HIDDEN ret name##_v1 args;
PSK_DECLARE_VARIANTS(int, bloom_add_partitioned, (psk_sketch *s, const Batch &b, hipStream_t st, bool *done))
PSK_DECLARE_VARIANTS(int, bloom_check_partitioned, (psk_sketch *s, const Batch &b, uint8_t *out_dev, hipStream_t st, bool *done))
PSK_DECLARE_VARIANTS(int, bloom_check_begin_partitioned, (psk_sketch *s, const Batch &b, hipStream_t st))
PSK_DECLARE_VARIANTS(int, bloom_check_finish_partitioned, (psk_sketch *s, uint8_t *out_dev, hipStream_t st, bool *redo_flag_possible))
PSK_DECLARE_VARIANTS(int, cms_add_partitioned, (psk_sketch *s, const Batch &b, const uint32_t *w_dev, hipStream_t st, bool *done))
PSK_DECLARE_VARIANTS(int, cms_remove_partitioned, (psk_sketch *s, const Batch &b, const uint32_t *w_dev, hipStream_t st, bool *done))
PSK_DECLARE_VARIANTS(int, cbf_add_partitioned, (psk_sketch *s, const Batch &b, const uint32_t *w_dev, hipStream_t st, bool *done))
PSK_DECLARE_VARIANTS(int, cbf_remove_partitioned, (psk_sketch *s, const Batch &b, const uint32_t *w_dev, hipStream_t st, bool *done, int opt, uint32_t *flag))  // decrement by w (opt: SpillCounter)
// lookups (psk_lookup.hpp): query = psk_query; out_dev int32 (min / mean) or int64 (mean-min); kk = hashes per key
PSK_DECLARE_VARIANTS(int, cms_check_partitioned, (psk_sketch *s, const Batch &b, int query, int64_t els_added, void *out_dev, hipStream_t st, bool *done))
PSK_DECLARE_VARIANTS(int, cbf_check_partitioned, (psk_sketch *s, const Batch &b, uint32_t kk, uint32_t *out_dev, hipStream_t st, bool *done))
// pass 1 of the return-trip lookups over a BLOOM-indexed table (IdxBloom, PayBloomLookup: perm[] / runinfo[] in s_perm / s_run, probes in
// s_part / s_cnt; an overflowing segment raises *flag): shared by the Bloom return trip and the CountingBloomFilter's 4-bit-slice lookups --
// ONE set of instantiations (psk_part_cbf_check.hip) instead of one per caller.  *fits = false: a tile too large for 16-bit stage positions.
PSK_DECLARE_VARIANTS(int, bloomidx_lookup_scatter, (psk_sketch *s, const Batch &sub, uint64_t cnt, uint32_t kk, PartGeom *g, uint32_t *flag, hipStream_t st, bool *handled, bool *fits))
PSK_HIDDEN int flush_combined(psk_sketch *s, hipStream_t st);  // apply the write-combined CBF updates, if any (psk_capi.hip)
// pass 1 of a unit-weight CBF batch, appended to the handle's persistent add (neg = 0) / decrement (neg = 1) list; *done = false:
// the batch / table is not eligible (nothing was launched)
PSK_DECLARE_VARIANTS(int, cbf_scat_append, (psk_sketch *s, const Batch &b, int neg, hipStream_t st, bool *done))
// unit-weight add (neg = 0) / unchecked decrement (neg = 1) of `n` borrowed 16-byte keys (device tables base[nb], start[nb + 1]) through the
// nibble path; *done = false: table not eligible (nothing launched)
PSK_DECLARE_VARIANTS(int, cbf_unit_multi_partitioned, (psk_sketch *s, const void *const *base_dev, const uint64_t *start_dev, uint32_t nb, uint64_t n, int neg, hipStream_t st, bool *done))
// pass 1 alone of a unit-weight batch into the handle's first / second bucket buffer (fused flush of the write-combined lists)
PSK_DECLARE_VARIANTS(int, cbf_nib_scatter, (psk_sketch *s, const Batch &b, int neg, int second, PartGeom *g_out, hipStream_t st, bool *done))
// validated unit-weight remove, fast path: pass 1 + the optimistic decrement (flag in s_flag); flag up: _undo adds the probe groups back
PSK_DECLARE_VARIANTS(int, cbf_remove_fast_begin, (psk_sketch *s, const Batch &b, hipStream_t st, bool *launched))
PSK_DECLARE_VARIANTS(int, cbf_remove_fast_undo, (psk_sketch *s, hipStream_t st))
// The fold of an update window (psk_window.hpp): phase-aware pass 1 over the window's key list + k_win_fold; *launched = false: table /
// window not eligible (nothing changed); *ok = false: the proof failed -- the fold has been undone, the caller replays batch by batch.
struct WinBatchHost {   // one waiting batch, in arrival order: n 16-byte keys at `keys` (the window's list, or the caller's own buffer)
    const void *keys;
    uint64_t n;
    uint32_t remove;
};
PSK_DECLARE_VARIANTS(int, cbf_window_fold, (psk_sketch *s, const WinBatchHost *wb, uint32_t nb, hipStream_t st, bool *launched, bool *ok))
extern PSK_HIDDEN __thread int64_t g_remove_exact, g_window, g_window_keys;
extern PSK_HIDDEN int64_t g_cbf_ordered_replays;
extern PSK_HIDDEN int64_t g_window_folds, g_window_replays, g_window_force_fail;
extern PSK_HIDDEN int64_t g_remove_dryrun;
extern PSK_HIDDEN __thread int64_t g_auto_combine;
extern PSK_HIDDEN int64_t g_auto_combine_keys, g_combine_keys, g_combine_scatter, g_fused_flush;
