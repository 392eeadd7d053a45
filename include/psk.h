/*
 * psk.h -- C ABI of the MI355X-native probabilistic-sketch engine (libpsk_hip.so).
 *
 * This is the drop-in boundary for ONE path of barrust/pyprobables (v0.7.0): bulk
 * insert / lookup of BloomFilter, CountingBloomFilter and CountMinSketch with the
 * default_fnv_1a hash family.  The reference is pure Python and has no FFI of its
 * own; each entry point below names the reference method(s) whose per-key loop it
 * replaces (paths relative to the reference checkout).  INTEGRATION.md shows the
 * ctypes stub a pyprobables maintainer would add to bind them.
 *
 * Conventions
 *   - threading: psk_last_error() is thread-local; a sketch handle owns scratch buffers, so one handle must not be
 *     used from two threads / two streams at the same time (different handles are independent).  Every call leaves
 *     the calling thread's current HIP device as it found it.
 *   - plain pointers and sizes only; every function returns a psk status code
 *     (PSK_OK == 0, negative on error; text via psk_last_error()); no exceptions
 *     cross the boundary.
 *   - `where` says where the caller's buffers (keys, offsets, weights, outputs) live:
 *     PSK_HOST  -> pageable/pinned host memory; the call stages through device
 *                  scratch and returns after the result is back on the host.
 *                  (That is what it promises -- not hipStreamSynchronize: a tiny batch, e.g. the
 *                  one-key calls of a per-key loop, ends as soon as its kernel has posted a
 *                  completion word in pinned memory, a few microseconds before the runtime sees
 *                  the stream idle.  Work the caller queued on `stream` EARLIER has completed by
 *                  then: the kernel ran behind it.  Option "host_poll_us" = 0 restores the wait.)
 *     PSK_DEVICE-> device memory of the sketch's GPU; the call only enqueues work on
 *                  `stream` (a hipStream_t passed as void*, NULL = default stream)
 *                  and returns immediately.
 *   - a sketch handle is externally synchronised: one in-flight batch per handle.
 *   - key batches come in four layouts (`layout` argument):
 *       PSK_KEYS_FIXED    data = uint8[n][key_len]            equal-length byte keys
 *       PSK_KEYS_VARLEN8  data = uint8 blob, offsets=uint64[n+1]   bytes / latin-1 str keys
 *       PSK_KEYS_VARLEN32 data = uint32 code points, offsets=uint64[n+1]  str keys with
 *                         code points > 255 (hashes.py:98 hashes ord(c), not UTF-8 bytes)
 *       PSK_KEYS_HASHES   data = uint64[n][key_len] pre-computed hashes (key_len = hashes
 *                         per key, >= k); the add_alt/check_alt path and the route for a
 *                         user-supplied hash_function (hashes.py:10-15 HashFuncT)
 *     For the three key layouts the engine computes default_fnv_1a(key, k)
 *     (hashes.py:71-103) inside the kernel.
 */
#ifndef PSK_H
#define PSK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psk_sketch psk_sketch; /* opaque handle: owns (or borrows) one device-resident table */

enum psk_status {
    PSK_OK = 0,
    PSK_EINVAL = -1,    /* bad argument */
    PSK_ENODEV = -2,    /* no usable HIP device */
    PSK_ENOMEM = -3,    /* device allocation failed */
    PSK_EHIP = -4,      /* a HIP runtime call failed (see psk_last_error) */
    PSK_ECONTRACT = -5  /* batch-stream contract violated (see psk_get_counters) */
};

enum psk_where {
    PSK_HOST = 0,
    PSK_DEVICE = 1,
    /* psk_cbf_add / psk_cbf_remove / psk_cbf_update_combined: a device buffer that stays valid AND UNCHANGED until the handle's next
     * flush (any entry point that applies the waiting updates, psk_flush, psk_clear; psk_sketch_get_option "window_pending_batches"
     * tells how many of the latest batches still wait).  16-byte-aligned 16-byte fixed-length unit-weight keys that the engine defers
     * (update windows, write-combining lists) are then not copied: the flush hashes them where they lie, all waiting batches in one
     * pass (no key copy, one key read; round 5: BASELINE cfg 4 spent 22 % of its step copying keys).  Everything else -- other layouts,
     * batches applied at once -- is treated as PSK_DEVICE: nothing is kept. */
    PSK_DEVICE_BORROWED = 2
};

enum psk_layout { PSK_KEYS_FIXED = 0, PSK_KEYS_VARLEN8 = 1, PSK_KEYS_VARLEN32 = 2, PSK_KEYS_HASHES = 3 };

enum psk_query { PSK_Q_MIN = 0, PSK_Q_MEAN = 1, PSK_Q_MEANMIN = 2 }; /* countminsketch.py:429-453 */

enum psk_kind { PSK_KIND_BLOOM = 0, PSK_KIND_CBF = 1, PSK_KIND_CMS = 2 };

/* ordered-update opcode source (psk_*_update_ordered) */
enum psk_opmode { PSK_OP_ADD = 0, PSK_OP_REMOVE = 1, PSK_OP_SIGNED = 2 /* w>=0 add(w), w<0 remove(-w) */ };

/* counters kept on the device per sketch (psk_get_counters) */
enum psk_counter {
    PSK_CTR_ADDED = 0,      /* sum of weights applied by add batches */
    PSK_CTR_REMOVED = 1,    /* sum of weights actually removed (CBF: to_remove, countingbloom.py:203-207) */
    PSK_CTR_VIOLATIONS = 2, /* unordered batch ops whose result depended on order (CBF partial/underflowing remove) */
    PSK_CTR_SATURATED = 3,  /* counter updates that hit a rail (UINT32_MAX / INT32_MAX / INT32_MIN) */
    PSK_CTR_ABS_BOUND = 4,  /* upper bound on |any counter| (selects the wrap-free fast path) */
    PSK_CTR_COUNT = 8
};

/* ------------------------------------------------------------------ misc */
const char *psk_last_error(void);         /* thread-local text of the last failure */
int psk_version(void);
int psk_device_count(int *count);
/* Options (tunables of this engine: no reference counterpart).  psk_set_option keeps the PROCESS value; the names marked (s) can also be
 * overridden for one handle with psk_sketch_set_option (value INT64_MIN: follow the process value again).
 *
 *   name                        default  meaning
 *   "partition"                 1        0 = direct kernels only (one fabric transaction per probe), 1 = large batches take the partitioned passes
 *   "partition_min_keys" (s)    65536    Bloom inserts of at least this many keys are partitioned (4 x as many for lookups / counter updates)
 *   "partition_max_keys"        2^26     keys per partition round (bounds the scratch: ~2 GB per round at k = 7)
 *   "partition_cache_bytes"     240 MiB  cache-sized tables: a batch whose probe buffer exceeds 1.5 x this is cut into equal rounds (0 = never)
 *   "scratch_budget_bytes" (s)  0        cap on a handle's partition scratch (more, smaller rounds); 0 = none
 *   "pass1_bins"                1        pass 1 of Bloom inserts / tile-flag lookups, CBF unit updates and weighted CMS adds through fixed-capacity
 *                                        LDS bins where they fit (psk_part_bins.hpp); 0 = the counting sort everywhere (same probes, same results)
 *   "bloom_lookup" (s)          2        large Bloom lookups: 0 keyed probes, 1 return trip, 3 tile flags (batches of present keys), 4 lazy gathers
 *                                        (batches of absent keys), 2 = chosen per call from the previous lookups' miss tally
 *   "cms_small_weights"         1        weighted psk_cms_add: weights 0 .. 15 travel as 20-bit fields once the previous batches brought no
 *                                        other weight; 0 = never, 2 = always (exact either way)
 *   "cbf_lookup_shadow" (s)     1        psk_cbf_check keeps 4-bit images of an unchanged big table between lookups (cells / 2 bytes)
 *   "remove_exact" (s)          1        psk_cbf_remove replays order-dependent batches in order; 0 = tallies them as violations instead
 *   "update_window" (s)         1        small unit psk_cbf_add / _remove batches into big tables share one proven pass (psk_cbf_add below)
 *   "update_window_keys" (s)    2^27     most keys such a window holds
 *   "auto_combine" (s)          1        (update windows off) small unit add batches wait as scattered probes
 *   "combine_keys"              2^26     keys per list of psk_cbf_update_combined
 *   "merge_single_rank"         0        1: psk_merge_* run the collective path on a one-rank communicator (tests)
 *   per sketch only: "table_private" (below), read-only "window_pending_batches" (batches an update window still holds: PSK_DEVICE_BORROWED
 *   buffers among them must stay as they are).
 *
 * Also accepted -- where the engine switches between its kernel families, and test hooks (the tests steer small inputs onto the big-table paths
 * with them; not for callers): "partition_two_level_slices", "auto_combine_keys", "tile_threads", "even_tiles", "dense_walk_groups",
 * "lookup_half_slices", "remove_optimistic", "lookup_nibble_slices", "update_nibble_slices", "nibble_min_lg_lookup", "nibble_min_lg_update",
 * "update_window_tile", "update_window_wide", "update_window_force_fail", "ragged_sort", "host_poll_us" (how long a tiny PSK_HOST call
 * polls its completion mailbox before it waits for the stream; 0 = never poll); read-only counters "cbf_ordered_replays",
 * "update_window_folds", "update_window_replays", "cms_small_weights_used", "cbf_lookup_shadow_hits".
 * The A/B switches of experiments that were measured and dropped (NOTES.md) exist only in the bench build (-DPSK_BENCH_KNOBS=1,
 * libpsk_hip_knobs.so); this library answers "unknown option" to them. */
int psk_set_option(const char *name, int64_t value);
int psk_get_option(const char *name, int64_t *value);
/* Per-sketch options (round 4): "partition_min_keys", "cbf_lookup_shadow", "auto_combine", "update_window", "update_window_keys",
 * "scratch_budget_bytes", "remove_exact", "bloom_lookup" can differ between the sketches of one process -- psk_set_option keeps the
 * DEFAULT of each, psk_sketch_set_option overrides it for one handle (value INT64_MIN: follow the default again).  One more name exists
 * per sketch only: "table_private" = 1 declares that whoever holds the pointer of a caller-owned table (ext_table) announces EVERY write it
 * makes behind the engine's back (psk_table_info before, psk_rescan_bound after); without that promise -- and between psk_table_info(&ptr)
 * and the next psk_rescan_bound -- the engine keeps nothing derived from the table (the 4-bit images of repeated psk_cbf_check calls).
 * No reference counterpart (tunables of this engine). */
int psk_sketch_set_option(psk_sketch *s, const char *name, int64_t value);
int psk_sketch_get_option(psk_sketch *s, const char *name, int64_t *value);
/* bench-only: s_memtime totals per phase of the last partition pass 1 (option part_debug & 32) */
int psk_debug_phase_profile(psk_sketch *s, uint32_t nbuckets, uint32_t nwg, uint64_t out[12]);

/* -------------------------------------------------------------- lifecycle
 * ext_table: NULL -> the library hipMallocs (and zeroes) the table;
 *            else -> device pointer to caller-owned, zero-initialised memory of at least
 *                    psk_table_bytes(...) bytes (lets the caller keep the table in a tensor it
 *                    can hand to RCCL).
 * Replaces the array allocation of bloom.py:105, countingbloom.py:78, countminsketch.py:115. */
uint64_t psk_bloom_table_bytes(uint64_t m_bits);            /* ceil(m/8) rounded up to 16 B */
uint64_t psk_cbf_table_bytes(uint64_t m);                   /* 4*m rounded up to 16 B */
uint64_t psk_cms_table_bytes(uint64_t width, uint32_t depth); /* 4*width*depth rounded up to 16 B */
int psk_bloom_create(uint64_t m_bits, uint32_t k, int device, void *ext_table, psk_sketch **out);
int psk_cbf_create(uint64_t m, uint32_t k, int device, void *ext_table, psk_sketch **out);
int psk_cms_create(uint64_t width, uint32_t depth, int device, void *ext_table, psk_sketch **out);
int psk_destroy(psk_sketch *s);                             /* ext_table: write-combined updates still waiting are applied first (NULL stream + sync) */
int psk_clear(psk_sketch *s, void *stream);                 /* bloom.py:217-221, countminsketch.py:240-244 */
int psk_synchronize(psk_sketch *s, void *stream);
int psk_release_scratch(psk_sketch *s);                     /* free staging + partition buffers (regrow on demand) */
/* what the handle holds besides its table right now (round 4): bytes[0] = device scratch in all (staging, bucket buffers, values / perm,
 * write-combining segments, update-window key list and snapshots, kept 4-bit images), bytes[1] = of which the update window / write-combining
 * lists (they hold waiting updates: psk_flush applies them), bytes[2] = of which the kept 4-bit images; no stream work, no synchronisation.
 * (The scratch is an implementation detail of the path behind countingbloom.py:135-208 / bloom.py:234-272: the reference has none.) */
int psk_scratch_bytes(psk_sketch *s, uint64_t bytes[3]);
/* device pointer + padded size + logical size (the reference's array byte length) */
int psk_table_info(psk_sketch *s, void **dev_ptr, uint64_t *padded_bytes, uint64_t *logical_bytes);
/* copy the first nbytes of the table to / from host memory in the reference's byte layout
 * (array('B') LSB-first bits, array('I') uint32 LE, array('i') int32 LE row-major by depth):
 * what export()/bytes()/frombytes() read and write (bloom.py:287-304,548-550; countminsketch.py:342-354,417-427) */
int psk_read_table(psk_sketch *s, void *dst_host, uint64_t nbytes, void *stream);
int psk_write_table(psk_sketch *s, const void *src_host, uint64_t nbytes, void *stream);
int psk_get_counters(psk_sketch *s, int64_t out[PSK_CTR_COUNT], void *stream); /* synchronises */
int psk_reset_counters(psk_sketch *s, void *stream);
/* recompute PSK_CTR_ABS_BOUND = max |counter| after the table was modified from outside the engine
 * (e.g. the RCCL all-reduce of the multi-GPU merge wrote into it) */
int psk_rescan_bound(psk_sketch *s, void *stream);

/* ------------------------------------------------------------- BloomFilter
 * add:   for each key: for i<k: bit = h_i % m; table |= bit      (bloom.py:234-250 add/add_alt)
 * check: for each key: out = AND_i bit(h_i % m)                  (bloom.py:252-272 check/check_alt)
 * check_bits: same result ballot-packed, bit (i & 63) of out_bits[i >> 6]; *hits += popcount (large batches: the partitioned lookup's
 *             answers -- n bytes of the handle's scratch -- packed by one streaming pass) */
int psk_bloom_add(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                  uint32_t key_len, int where, void *stream);
int psk_bloom_check(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                    uint32_t key_len, int where, uint8_t *out, void *stream);
int psk_bloom_check_bits(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                         uint32_t key_len, int where, uint64_t *out_bits, uint64_t *hits, void *stream);
/* Split lookup for DEVICE-resident keys: _begin hashes and partitions the batch without ever reading the table, _finish
 * probes it and writes out_dev[n] (0/1).  Everything between the two calls may still change the table -- the point is
 * to run pass 1 under a multi-GPU merge (pyprobables_amd/parallel.py) that is in flight on another stream.  The key
 * buffers must stay valid and unchanged until _finish; one pending lookup per handle.  Result == psk_bloom_check at
 * _finish time (bit-exact; an overflowing bucket segment is re-checked on the device). */
int psk_bloom_check_begin(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n, uint32_t key_len,
                          void *stream);
int psk_bloom_check_finish(psk_sketch *s, uint8_t *out_dev, void *stream);

/* ----------------------------------------------------- CountingBloomFilter
 * Batches (weights: uint32[n] or NULL = all 1).  Round 4: the result of every batch equals the reference's per-key loop over it for ANY
 * stream -- adds commute (the clamp at 2^32-1 included); psk_cbf_remove is a TRANSACTION: it runs unordered (optimistic decrement, or lookup
 * -> amounts -> checked decrement) and proves on the way that the result does not depend on the order inside the batch; where it does
 * (duplicates of a key with too few inserts, a false positive among absent keys, a frozen counter) the batch is undone and replayed in
 * order on the device (k_cbf_ordered).  Option "remove_exact" = 0 restores the round-3 contract: exact for well-formed streams (no counter
 * saturates; every remove targets a key with >= num_els live inserts), anything else tallied in PSK_CTR_VIOLATIONS / PSK_CTR_SATURATED.
 * Update WINDOWS (round 4; tables of more than 2^24 counters, 16-byte keys, unit weights; options "update_window" = 0: off,
 * "update_window_keys": capacity): add / remove batches too small to pay for a pass over the table wait on the device, in arrival order, as
 * key copies (the caller's buffer is free when the call returns); the window reaches the table in ONE pass that walks it batch run by batch
 * run and proves every deferred remove (it must meet a non-zero counter at its own position of the stream: countingbloom.py:198-201);
 * a window that fails the proof is undone and replayed batch by batch through the transaction above.  Every entry point that reads the
 * table applies what is waiting first (psk_flush before using psk_table_info's pointer).
 * add:    counters[h_i % m] = min(c + w, 2^32-1) for i<k        (countingbloom.py:125-155)
 * remove: conditional decrement                                 (countingbloom.py:176-208)
 * check:  out = min_i counters[h_i % m]                         (countingbloom.py:157-174)
 * Big tables (more than 2^24 counters -- option "nibble_min_lg_update"; round 3):
 *   - unit-weight add batches too small to pay for a pass over the table (n * k < m / 8) are write-combined automatically: the
 *     batch is hashed and partitioned when it is handed over, its probes wait in persistent per-slice segments and reach the
 *     table together with their successors (adds commute, the clamp at 2^32-1 is applied all the same: exact, no opt-in;
 *     option "auto_combine" = 0 turns it off, "auto_combine_keys" sizes the segments).  Every entry point that reads the
 *     table, removes or hands out its pointer applies what is waiting first (psk_flush before using psk_table_info's pointer).
 *   - a unit-weight psk_cbf_remove of at least m / 8 probes decrements optimistically (one pass; exact whenever every counter
 *     holds what the batch takes from it, i.e. every key is present) and otherwise undoes that and takes the lookup + masked
 *     decrement path; it reads ONE 4-byte verdict back, i.e. synchronises `stream` once (option "remove_optimistic" = 0: never).
 *   - lookups of a table that is not changing (from 2^23 counters on): the partitioned psk_cbf_check squeezes the 32-bit table
 *     into 4-bit slice images on every call; from the second such call in a row on it keeps them (cells / 2 bytes, freed by
 *     psk_release_scratch) and later calls on the same stream load them instead, until any entry point that may write the
 *     table -- or psk_table_info, which hands its pointer out -- is called on the handle (option "cbf_lookup_shadow" = 0: never).
 *     Whoever writes to the table behind the engine's back must say so afterwards: psk_rescan_bound (as for the wrap-free bound). */
int psk_cbf_add(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                uint32_t key_len, const uint32_t *weights, int where, void *stream);
int psk_cbf_remove(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                   uint32_t key_len, const uint32_t *weights, int where, void *stream);
int psk_cbf_check(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                  uint32_t key_len, int where, uint32_t *out, void *stream);
/* Write-combined updates incl. REMOVES (opt-in).  The fold of a big counter table read-modify-writes the whole table whatever the batch
 * brings (1 GiB at BASELINE config 4), so small batches -- the config's 1M-key add / remove batches -- are collected on the
 * device and applied as ONE partitioned update per list once "combine_keys" keys (psk_set_option, default 2^26) are waiting:
 * first the adds (countingbloom.py:135-155), then the removes as plain decrements of every index by the key's weight
 * (countingbloom.py:203-206 with to_remove == num_els).  Exact for well-formed streams (every remove targets a key with at
 * least num_els live inserts at that point of the stream; nothing saturates) -- the contract of the unordered batch ops
 * above; a decrement below zero is tallied in PSK_CTR_VIOLATIONS.  Unlike psk_cbf_remove no per-key min is read first, so a
 * remove of an ABSENT key is a violation here, not a no-op.  Every other entry point on the handle (check, read_table,
 * get_counters, synchronize, merge ...) applies what is waiting first; psk_clear / psk_write_table drop it.  Before handing
 * the raw table pointer (psk_table_info) to anything else call psk_flush.  remove = 0: add, 1: remove. */
int psk_cbf_update_combined(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                            uint32_t key_len, const uint32_t *weights, int remove, int where, void *stream);
int psk_flush(psk_sketch *s, void *stream);
/* Ordered (one-at-a-time, in sequence) execution of the reference semantics on the GPU, including
 * each op's return value: exact for ANY stream.  weights int64[n] or NULL (=1); opmode psk_opmode. */
int psk_cbf_update_ordered(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                           uint32_t key_len, const int64_t *weights, int opmode, int where, uint32_t *out,
                           void *stream);

/* ---------------------------------------------------------- CountMinSketch
 * bins row-major by depth: bin = (h_i % width) + i*width, i < depth (countminsketch.py:275)
 * add:    saturating at INT32_MAX    (countminsketch.py:257-288)
 * remove: saturating at INT32_MIN    (countminsketch.py:290-321)
 * check:  min | mean (int32 out);  mean-min needs elements_added (int64 out)  (countminsketch.py:323-340,429-453)
 * Unordered batches are bit-exact with the reference whenever the final table does not depend on
 * order (same-sign weights, or no bin touching a rail). */
int psk_cms_add(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                uint32_t key_len, const int32_t *weights, int where, void *stream);
int psk_cms_remove(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                   uint32_t key_len, const int32_t *weights, int where, void *stream);
int psk_cms_check(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                  uint32_t key_len, int where, int query /* MIN or MEAN */, int32_t *out, void *stream);
int psk_cms_check_meanmin(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                          uint32_t key_len, int where, int64_t elements_added, int64_t *out, void *stream);
/* out (optional): int64[n + 1] -- the n return values, then elements_added after the batch (exact int64 clamps) */
int psk_cms_update_ordered(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n,
                           uint32_t key_len, const int64_t *weights, int opmode, int query,
                           int64_t elements_added_in, int where, int64_t *out, void *stream);

/* ------------------------------------------------------------------ hashing
 * out[i*depth + j] = fnv_1a(key_i, seed=j)  (hashes.py:71-103); layout != PSK_KEYS_HASHES */
int psk_fnv1a_hash(int layout, const void *data, const uint64_t *offsets, uint64_t n, uint32_t key_len,
                   uint32_t depth, int where, uint64_t *out, int device, void *stream);
/* the digest families (hashes.py:17-40,125-150 default_md5 / default_sha256): a CHAIN, tmp = H(tmp).digest() per depth
 * step, out[i*depth + j] = LE64 of the first 8 bytes of link j.  Byte layouts only (PSK_KEYS_FIXED / PSK_KEYS_VARLEN8);
 * the reference UTF-8-encodes a str for these families (hashes.py:34), which is the caller's packing job. */
enum psk_digest { PSK_DIGEST_MD5 = 0, PSK_DIGEST_SHA256 = 1 };
int psk_digest_chain(int algo, int layout, const void *data, const uint64_t *offsets, uint64_t n, uint32_t key_len,
                     uint32_t depth, int where, uint64_t *out, int device, void *stream);

/* ------------------------------------------------------ table algebra (device pointers)
 * Streaming kernels over whole tables; also the local half of the multi-GPU merge.
 * or/and: bloom.py:371-428 union/intersection;  popcount: bloom.py:552-557;
 * add_sat_i32: countminsketch.py:380-391 join;  add_u32: countingbloom.py:296-298 union;
 * or_reduce: dst[w] = OR_j src[j*slice_words + w] (the reduce step of allreduce(OR): RCCL has no OR op) */
int psk_table_or(void *dst, const void *src, uint64_t nwords32, int device, void *stream);
int psk_table_and(void *dst, const void *src, uint64_t nwords32, int device, void *stream);
int psk_table_popcount(const void *tab, uint64_t nwords32, uint64_t *out_host, int device, void *stream);
int psk_table_nonzero_u32(const void *tab, uint64_t nwords32, uint64_t *out_host, int device, void *stream);
int psk_table_add_sat_i32(void *dst, const void *src, uint64_t n, int device, void *stream);
int psk_table_add_u32(void *dst, const void *src, uint64_t n, uint64_t *overflowed_host, int device, void *stream);
/* countingbloom.py:210-269: intersection (sum where both non-zero) and the two counts of jaccard_index */
int psk_cbf_intersect(void *dst, const void *a, const void *b, uint64_t n, uint64_t *overflowed_host, int device, void *stream);
int psk_cbf_jaccard_counts(const void *a, const void *b, uint64_t n, uint64_t out_host[2] /* union, intersection */,
                           int device, void *stream);
int psk_or_reduce_slices(void *dst, const void *src, uint32_t nslices, uint64_t slice_words32, int device,
                         void *stream);

/* ------------------------------------------------------ multi-GPU merge (SURVEY.md 8b "psk_merge_allreduce", 8e)
 * One rank per GPU holds a full-size replica fed with its own range of the key stream; ONE collective makes every replica
 * the table a single sketch fed the whole stream would hold.  `nccl_comm` is the caller's ncclComm_t (RCCL) for this rank,
 * passed as void* (ncclCommInitRank / ncclCommInitAll: one communicator per device); the handle's table is merged in
 * place on `stream`.  RCCL is looked up in the host process (no link-time dependency of libpsk_hip.so).
 *   psk_merge_or   Bloom: allreduce(OR) = grouped ncclSend/ncclRecv of the bit-range slices (slice j goes straight to rank
 *                  j), the engine's OR-reduce kernel, ncclAllGather.  bloom.py:401-428 (union is a bytewise OR).
 *                  elements_added of the merged filter is the SUM over ranks (the caller's scalar).
 *   psk_merge_sum  CMS / CBF: allreduce(SUM) of the counters -- 32-bit while the summed per-rank bounds prove that no counter
 *                  reaches a rail, else 64-bit and clamped like join (countminsketch.py:380-391; CBF at 2^32-1,
 *                  countingbloom.py:149-151); the tallies of psk_get_counters (added / removed / violations / saturated)
 *                  become global totals (cells the merge itself clamps are counted once, not once per rank).  ONE-SHOT per
 *                  stream segment: replicas and tallies are per-rank deltas going in, global totals coming out -- merging
 *                  the same replicas twice counts everything again.  Synchronises `stream` once (8-byte read-back of the
 *                  summed bound; each rank's bound is capped at 2^40 first so the sum cannot wrap).
 * One-rank communicators return at once (option "merge_single_rank" = 1 runs the collective path anyway: tests). */
int psk_merge_or(psk_sketch *s, void *nccl_comm, void *stream);
int psk_merge_sum(psk_sketch *s, void *nccl_comm, void *stream);

/* --------------------------------------------- filters whose insert depends on a lookup
 * ExpandingBloomFilter / RotatingBloomFilter (expandingbloom.py:140-170, :320-331): a key goes into the newest filter
 * unless ANY filter of the stack already reports it.  All filters share (m, k, hash): hash once, then index kernels.
 *   psk_bloom_indices        out_idx_dev[i*k + j] = hash_j(key_i) % m  (bloom.py:247; device buffer; m <= 2^32)
 *   psk_idx_test             out[i] = all k bits set (bloom.py:269-271); accumulate != 0: out[i] |= ...  (the `any`
 *                            over the filters of expandingbloom.py:147)
 *   psk_idx_resolve_ordered  flag[i] = 1 iff the sequential loop "if key not in table: table.add(key)" over the ordered
 *                            batch would insert key i (expandingbloom.py:166-170), else 0; present[i] != 0 marks keys
 *                            another filter already holds (nullable); table_dev is NOT modified.  first_dev: uint32[m]
 *                            scratch, all-ones on entry and again on exit; count_dev: uint64 scratch; *inserted_host
 *                            receives the number of inserts (NULL: no host sync)
 *   psk_idx_insert           table |= bits of the keys with flag[i] == 1 (flag NULL: all keys)  (bloom.py:247-249)
 *   psk_bytes_or             dst[i] |= src[i] */
int psk_bloom_indices(psk_sketch *s, int layout, const void *data, const uint64_t *offsets, uint64_t n, uint32_t key_len,
                      int where, uint32_t *out_idx_dev, void *stream);
int psk_idx_test(const void *table_dev, const uint32_t *idx_dev, uint64_t n, uint32_t k, uint8_t *out_dev, int accumulate,
                 int device, void *stream);
int psk_idx_resolve_ordered(const void *table_dev, const uint32_t *idx_dev, const uint8_t *present_dev, uint64_t n, uint32_t k,
                            uint32_t *first_dev, uint8_t *flag_dev, uint64_t *count_dev, uint64_t *inserted_host, int device,
                            void *stream);
/* the same resolution with the per-bit scratch replaced by a bounded hash map (round 4): slots_dev = uint32[2][2^lg_slots], all-ones on entry
 * and on exit, 2^lg_slots >= 2 * n * k.  The caller walks an ordered chunk in sub-chunks of n keys -- resolve, insert, next -- so the scratch
 * is a few MB whatever the filter's size (psk_idx_resolve_ordered needs 4 bytes per filter BIT).  count_dev[0] += keys to insert,
 * count_dev[1] != 0: the map overflowed (cannot happen with the sizing above).  Enqueue only: no host synchronisation. */
int psk_idx_resolve_ordered_hashed(const void *table_dev, const uint32_t *idx_dev, const uint8_t *present_dev, uint64_t n, uint32_t k,
                                   void *slots_dev, uint32_t lg_slots, uint8_t *flag_dev, uint64_t *count_dev, int device, void *stream);
int psk_idx_insert(void *table_dev, const uint32_t *idx_dev, const uint8_t *flag_dev, uint64_t n, uint32_t k, int device,
                   void *stream);
int psk_bytes_or(void *dst_dev, const void *src_dev, uint64_t n, int device, void *stream);

/* --------------------------------------------- synthetic streams (bench / tests)
 * key_i = LE64(sm(seed+2i)) || LE64(sm(seed+2i+1)); w_i = 1 + sm((seed^0xC0FFEE)+i) % 7 */
int psk_gen_keys16(void *dst_dev, uint64_t start, uint64_t n, uint64_t seed, int device, void *stream);
int psk_gen_weights(void *dst_dev, uint64_t start, uint64_t n, uint64_t seed, int device, void *stream);
/* GUPS-style ceiling: n uniformly random 4-byte atomic ORs (op=0), atomic adds (op=1) or loads (op=2)
 * into a table of nwords32 words; the denominator of the "random-access roofline" */
int psk_gups(void *table_dev, uint64_t nwords32, uint64_t n, int op, uint64_t seed, uint64_t *sink_dev,
             int device, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PSK_H */
