// psk_partition.hpp -- the large-batch path: radix-partition the probes by table slice, then apply
// each slice inside LDS.
//
// Why: the direct kernels issue one device-scope atomic (a 32 B fabric write, ~27 G/s chip-wide) or
// one 64 B line fetch per 4-byte probe; rocprof shows 3-9x the algorithmic bytes crossing the fabric
// and the kernels pinned to that transaction ceiling.  Here the table is cut into slices that fit
// one CU's LDS (<= 128 KiB of the 160 KiB):
//   pass 1  k_part_scatter : hash the keys, bin every probe by slice (LDS histogram + LDS counting
//                            sort per tile of 1024..4096 keys) and append each bin as a coalesced run of
//                            16-byte GROUPS to the (slice, workgroup) SEGMENT of the bucket buffer in HBM.
//                            Segments are private to one workgroup, so the append cursor lives in LDS: no
//                            global atomics, no cross-workgroup line sharing (a first version reserved
//                            space with one returning device atomic per (tile, slice): 1.25 M atomics =
//                            300 of its 400 us).
//   pass 2  k_*_apply      : one workgroup per slice keeps the slice in LDS, streams the slice's segments
//                            in (one wave per segment, dwordx4), does the random bit / counter updates
//                            with LDS atomics (ds_or / ds_add) and merges the slice back with one
//                            coalesced read-modify-write.  No atomics on the table, no random HBM access.
// Probe encodings (one uint4 group each, pads = 0xFFFFFFFF):
//   packed  6 x 20-bit bit-in-slice (+ 2 x 4-bit counts)     Bloom insert            (2.67 B / probe)
//           8 x 16-bit cell-in-slice (0xFFFF = pad)          unit-weight counter adds (2 B / probe)
//   inline  4 x (weight << shift | cell index within slice)  weighted counter adds, weight < 2^(31-shift)
//   keyed   4 x (tile bit << 31 | key index within tile << shift | bit index within slice)   Bloom lookups (4 B / probe)
// Anything that does not fit (segment overflow on adversarial / duplicate-heavy batches, weights too
// large to inline) falls back to an exact direct atomic on the table, so the result is always exact.
#pragma once
#include <type_traits>
#include "psk_device.hpp"

// Bench-only ablation / phase-profile knobs (PartGeom::dbg) exist only in a -DPSK_BENCH_KNOBS=1 build
// (python -m pyprobables_amd.build --knobs -> csrc/libpsk_hip_knobs.so, loaded through PSK_LIB_PATH by scripts/ablate.py);
// in the shipped library every `dbg & bit` test below is a compile-time false.
#ifndef PSK_BENCH_KNOBS
#define PSK_BENCH_KNOBS 0
#endif

namespace psk {

constexpr bool kBenchKnobs = PSK_BENCH_KNOBS != 0;
constexpr int kPartThreads = 512;       // 8 wavefronts per workgroup (k > 8); small k uses 16, see PartTile::NT
constexpr int kPartProbes = 32;        // PartTile::PP = kPartProbes / 2 = 16 probes per thread per tile (<= 128 VGPRs)
constexpr int kPartMaxBuckets = 2048;
constexpr int kPartScanPerThread = 4;   // slices per lane of a scanning thread: 64 x 4 = one wave covers 256 slices
constexpr bool kPartPipeline = true;   // prefetch the next tile's keys under the current tile (see k_part_scatter)
constexpr bool kPartHash32 = true;     // explicit 32-bit FNV chains for power-of-two tables
constexpr uint32_t kNibShift = 18;           // log2(counters per 4-bit slice image): psk_nibble.hpp (CountingBloomFilter tables beyond 2^26 cells)
constexpr uint32_t kGeomNoSortBit = 0x40000000u;  // PartGeom::dbg bit 30 (a production bit): option "ragged_sort" is off -- ragged keys go to the lanes in batch order
constexpr int kSortSub = 16;                 // pass 1's length sort of ragged keys: counters per length class (one per lane mod 16)
constexpr int kSortBins = 64 * kSortSub;
constexpr uint32_t kPadProbe = 0xFFFFFFFFu;  // filler that pads every run to whole groups; pass 2 skips it
// Wave priorities inside pass 1 (s_setprio; -DPSK_EXP_PRIO=n for A/B builds, scripts/build_variant.sh): 1 (default) = the hash phase runs at
// priority 0 and everything behind barrier 1 (scan, sort, write-out: short dependent VALU sequences between LDS round trips) at priority 3,
// so that a workgroup in its LDS phases is never queued behind the other workgroup's hash chains (same-box A/B, profiles/r06_ab_pass1.txt:
// cfg 2 step 369 -> 362-366 us with k_part_scatter, 350 -> 346 with k_part_bins); 0 = all waves equal, 2 = the reverse (no gain), 3 = only the
// scanning wave raised (no gain)
#ifndef PSK_EXP_PRIO
#define PSK_EXP_PRIO 1
#endif

enum PartMode { kModePlain = 0, kModeInline = 1, kModeKeyed = 2 };

struct PartGeom {
    uint32_t nbuckets;      // B = ceil(cells / 2^shift) slices
    uint32_t shift;         // log2(cells per slice)
    uint32_t nwg;           // workgroups of pass 1 (segments per slice)
    uint32_t segcap;        // 16-byte groups per (slice, workgroup) segment
    uint32_t k;             // hashes per key
    uint32_t tile;          // keys per pass-1 tile (keyed probes: key = tile id * tile + local index)
    uint32_t dbg;           // bench-only bits: 1 skip stores, 4 skip hashing, 8 one workgroup per CU, 32 phase profile
    uint32_t split;         // read-only pass-2 kernels: workgroups per slice (0 / 1 = one); workgroup `split_idx` of a slice
    uint32_t split_idx;     // walks segments split_idx, split_idx + split, ... (balances slice counts that do not fill the CUs)
    uint32_t dense;         // pass 2: 1 = short segments, a wave walks its segments end to end (for_each_batch_at); set by pass 1's launcher
    uint32_t append;        // pass 1: 1 = my segments already hold segcnt[] groups from earlier launches with the SAME geometry -- append behind
                            // them (write-combined updates: every batch is scattered when it is handed over, pass 2 runs once per flush)
};

// Where segment (slice b, workgroup wg) lives in the bucket buffer.  Workgroup-major: the B runs a workgroup
// emits per tile go to B neighbouring segments (a few KiB apart) instead of nwg * segcap * 16 B apart
// (a multiple of 8 KiB).  Measured neutral on MI355X (197.8 vs 196 us); kept because it is the safer layout.
__device__ __forceinline__ uint64_t seg_index(const PartGeom &g, uint32_t b, uint32_t wg) { return (uint64_t)wg * g.nbuckets + b; }

// idx functors: which table cell does hash j of a key address?
// lo32: the index needs only the low 32 bits of the hash (power-of-two modulus <= 2^32)
// The modulus of IdxBloom / IdxCms is at most 2^31 (single-level geometry: <= 2048 slices of <= 2^20 cells; counter tables
// <= 2^29 cells): non-power-of-two ones reduce with reduce_small.  IdxBloomWide serves the two-level Bloom insert (m < 2^32).
template <bool POW2>
struct IdxBloom {  // bloom.py:247 / countingbloom.py:145:  h % m
    static constexpr bool lo32 = POW2 && kPartHash32;
    Mod md;
    __device__ __forceinline__ uint32_t operator()(uint32_t, uint64_t h) const
    {
        return POW2 ? (uint32_t)reduce<true>(md, h) : reduce_small(md, h);
    }
    __device__ __forceinline__ uint32_t from32(uint32_t, uint32_t h) const { return h & (uint32_t)md.mask; }
};
template <bool POW2>
struct IdxBloomWide {
    static constexpr bool lo32 = POW2 && kPartHash32;
    Mod md;
    __device__ __forceinline__ uint32_t operator()(uint32_t, uint64_t h) const { return (uint32_t)reduce<POW2>(md, h); }
    __device__ __forceinline__ uint32_t from32(uint32_t, uint32_t h) const { return h & (uint32_t)md.mask; }
};
template <bool POW2>
struct IdxCms {  // countminsketch.py:275:  (h % width) + i*width
    static constexpr bool lo32 = POW2 && kPartHash32;
    Mod md;
    __device__ __forceinline__ uint32_t operator()(uint32_t j, uint64_t h) const
    {
        return (POW2 ? (uint32_t)reduce<true>(md, h) : reduce_small(md, h)) + j * (uint32_t)md.m;
    }
    __device__ __forceinline__ uint32_t from32(uint32_t j, uint32_t h) const
    {
        return (h & (uint32_t)md.mask) + j * (uint32_t)md.m;
    }
};

// (a payload functor with `static constexpr bool lookup = true` asks pass 1 for the by-products the partitioned
// lookups need -- psk_lookup.hpp: perm[] and runinfo[])
template <class Pay, class = void>
struct pay_max_tile { static constexpr int value = 1 << 30; };
template <class Pay>
struct pay_max_tile<Pay, decltype((void)Pay::max_tile)> { static constexpr int value = Pay::max_tile; };
template <class Pay, class = void>
struct pay_fat512 { static constexpr bool value = false; };
template <class Pay>
struct pay_fat512<Pay, decltype((void)Pay::fat512)> { static constexpr bool value = Pay::fat512; };
// 32 probes per thread in the 1024-thread shape as well: 4096-key tiles for tables of ~900 slices and more, where a 2048-key tile brings
// only 2-3 probe groups per slice and the per-tile, per-slice work (scan, pads, separately addressed group stores) dominates pass 1
template <class Pay, class = void>
struct pay_fat1024 { static constexpr bool value = false; };
template <class Pay>
struct pay_fat1024<Pay, decltype((void)Pay::fat1024)> { static constexpr bool value = Pay::fat1024; };
// the same payload in the plain shapes (16 probes per thread): for the key layouts that hold more per key in registers than the 16-byte
// ones (launch_scatter; the probe format, and with it pass 2, is the payload's)
template <class Pay>
struct PaySlim : Pay {
    static constexpr bool fat512 = false;
    static constexpr bool fat1024 = false;
    __host__ __device__ PaySlim(const Pay &p) : Pay(p) {}
};
template <class Pay, class = void>
struct pay_has_tally { static constexpr bool value = false; };
template <class Pay>
struct pay_has_tally<Pay, decltype((void)&Pay::tally)> { static constexpr bool value = true; };
template <class Pay, class = void>
struct pay_has_keep { static constexpr bool value = false; };
template <class Pay>
struct pay_has_keep<Pay, decltype((void)&Pay::keep)> { static constexpr bool value = true; };
// key sources whose keys differ in length (KeysVarlen): pass 1 hands the keys of a tile to its lanes in order of length (below)
template <class Src, class = void>
struct src_sorted { static constexpr bool value = false; };
template <class Src>
struct src_sorted<Src, decltype((void)Src::sorted)> { static constexpr bool value = Src::sorted; };
template <class Pay, class = void>
struct pay_weighted_plain { static constexpr bool value = false; };
template <class Pay>
struct pay_weighted_plain<Pay, decltype((void)Pay::weighted_plain)> { static constexpr bool value = Pay::weighted_plain; };
template <class Pay, class = void>
struct pay_is_lookup { static constexpr bool value = false; };
template <class Pay>
struct pay_is_lookup<Pay, decltype((void)Pay::lookup)> { static constexpr bool value = Pay::lookup; };

template <class Pay, class = void>
struct pay_tile_tag { static constexpr bool value = false; };
template <class Pay>
struct pay_tile_tag<Pay, decltype((void)Pay::tile_tag)> { static constexpr bool value = Pay::tile_tag; };

template <class Pay, class = void>
struct pay_is_phased { static constexpr bool value = false; };
template <class Pay>
struct pay_is_phased<Pay, decltype((void)Pay::phased)> { static constexpr bool value = Pay::phased; };

// perm[] record of one key (the partitioned lookups, psk_lookup.hpp): the position of each of its KT probes inside the tile's sorted
// stage, 16 bits each, in PD = ceil(KT / 2) dwords -- 12 bytes for k = 5 / 6, 8 for k <= 4 (round 4; rounds 1-3 wrote whole 16-byte
// units: 16 bytes for every k <= 8, written by pass 1 and read back by pass 3).  Records are dword aligned; one wide access per <= 4 dwords.
typedef uint32_t perm_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t perm_u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t perm_u32x4 __attribute__((ext_vector_type(4)));
typedef perm_u32x2 perm_u32x2a __attribute__((aligned(4)));
typedef perm_u32x3 perm_u32x3a __attribute__((aligned(4)));
typedef perm_u32x4 perm_u32x4a __attribute__((aligned(4)));
template <int KT>
struct PermRec {
    static constexpr int PD = (KT + 1) / 2;
    uint32_t w[PD];
    __device__ __forceinline__ uint32_t pos(int j) const { return (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu; }
};
template <int PD, int C = 0>
__device__ __forceinline__ void perm_load_chunks(uint32_t (&w)[PD], const uint32_t *p)
{
    if constexpr (PD - C >= 4) { const perm_u32x4 v = *reinterpret_cast<const perm_u32x4a *>(p + C); w[C] = v.x; w[C + 1] = v.y; w[C + 2] = v.z; w[C + 3] = v.w; }
    else if constexpr (PD - C == 3) { const perm_u32x3 v = *reinterpret_cast<const perm_u32x3a *>(p + C); w[C] = v.x; w[C + 1] = v.y; w[C + 2] = v.z; }
    else if constexpr (PD - C == 2) { const perm_u32x2 v = *reinterpret_cast<const perm_u32x2a *>(p + C); w[C] = v.x; w[C + 1] = v.y; }
    else if constexpr (PD - C == 1) w[C] = p[C];
    if constexpr (PD - C > 4) perm_load_chunks<PD, C + 4>(w, p);
}
template <int PD, int C = 0>
__device__ __forceinline__ void perm_store_chunks(const uint32_t (&w)[PD], uint32_t *p)
{
    if constexpr (PD - C >= 4) { perm_u32x4 v; v.x = w[C]; v.y = w[C + 1]; v.z = w[C + 2]; v.w = w[C + 3]; *reinterpret_cast<perm_u32x4a *>(p + C) = v; }
    else if constexpr (PD - C == 3) { perm_u32x3 v; v.x = w[C]; v.y = w[C + 1]; v.z = w[C + 2]; *reinterpret_cast<perm_u32x3a *>(p + C) = v; }
    else if constexpr (PD - C == 2) { perm_u32x2 v; v.x = w[C]; v.y = w[C + 1]; *reinterpret_cast<perm_u32x2a *>(p + C) = v; }
    else if constexpr (PD - C == 1) p[C] = w[C];
    if constexpr (PD - C > 4) perm_store_chunks<PD, C + 4>(w, p);
}
template <int KT>
__device__ __forceinline__ PermRec<KT> perm_load(const uint32_t *perm, uint64_t i)
{
    PermRec<KT> r;
    perm_load_chunks<PermRec<KT>::PD>(r.w, perm + i * PermRec<KT>::PD);
    return r;
}
template <int KT>
__device__ __forceinline__ void perm_store(uint32_t *perm, uint64_t i, const PermRec<KT> &r)
{
    perm_store_chunks<PermRec<KT>::PD>(r.w, perm + i * PermRec<KT>::PD);
}

// payload functors: the second word a probe carries through the LDS sort (key i of a tile starting at base)
struct PayNone {   // Bloom insert: 6 probes per group, 20-bit slice-local bit indices
    static constexpr int mode = kModePlain;
    static constexpr int group = 6;
    static constexpr bool fat512 = true;  // PartTile: two 512-thread workgroups per CU with 32 probes per thread (measured: -3.5 %)
    static constexpr bool fat1024 = true; // ... and 32 per thread in the 1024-thread shape too (tables of more than ~512 slices: 4096-key tiles)
    __device__ __forceinline__ uint32_t operator()(uint64_t, uint64_t) const { return 0; }
};
// Bloom lookups of batches whose keys are (nearly) all present (round 5): PayNone's 2.67-byte probes -- the lookup's pass 1 is then the
// insert's, not the keyed one with its 4-byte probes -- and the two spare bits of each half's count nibble spell the tile's ORDINAL inside
// its workgroup's sequence (tile = ordinal * workgroups + the segment's workgroup, as for PayKeyId: at most 16 tiles per workgroup and
// round).  A hit needs no way back to its key: pass 2 (k_bloom_test_flag) only raises tileflag[tile] when a probe of the tile finds its bit
// clear, and k_bloom_flag_resolve re-checks the keys of the flagged tiles directly.
struct PayTileTag {
    static constexpr int mode = kModePlain;
    static constexpr int group = 6;
    static constexpr bool fat512 = true;
    static constexpr bool fat1024 = true;
    static constexpr bool tile_tag = true;
    static constexpr uint32_t max_tiles_per_wg = 16;
    __device__ __forceinline__ uint32_t operator()(uint64_t, uint64_t) const { return 0; }
};
struct PayUnitMasked {  // PayNone's probes, for the keys with amount[i] != 0 only: the decrement of the validated CBF remove, whose
    static constexpr int mode = kModePlain;  // per-key amounts are 0 (absent / frozen: countingbloom.py:198-201) or 1
    static constexpr int group = 6;
    const uint32_t *amount;
    __device__ __forceinline__ uint32_t operator()(uint64_t, uint64_t) const { return 0; }
    __device__ __forceinline__ uint32_t keep(uint64_t i) const { return amount[i]; }
};
// One PHASE of a CountingBloomFilter update window (psk_window.hpp): a run of same-type batches (all adds, or all removes).
// Pass 1 runs ONCE over all waiting keys, but a tile never straddles two phases: phase p owns the tiles [tile0, next phase's tile0).
// The fold reads a table of its phases (`remove` = 0 / 1); pass 1 reads a table of PIECES of them -- one per stretch of keys that is
// contiguous in memory (a batch, or part of one: batches wait where the caller left them, or as copies in the window's list) -- whose
// `remove` also says whether the piece ends its phase (kPieceEndsPhase: the workgroups leave their segment counts) and which one (bits 8..).
// Key j of tile t of a piece is key_off + t * tile + j in units of 16 bytes from address 0: the piece's last tile is short.
struct PhaseDesc {
    uint32_t tile0;            // first pass-1 tile of the phase / piece (entry [n] closes the table: tile0 = number of tiles)
    uint32_t remove;           // bit 0: 0 adds, 1 removes (countingbloom.py:135-155 / :186-208)
    long long key_off;
    unsigned long long nkeys;
};
constexpr uint32_t kPieceEndsPhase = 2u;
// PayNone's probes for such a list.  A (slice, workgroup) segment receives its tiles in tile order, i.e. in phase order, so the
// segment's fill count at the END of every phase cuts it into per-phase pieces: the workgroup leaves these counts in
// snap[phase][slice][workgroup] (the fold walks a slice phase by phase: adds, barrier, removes that must not meet a zero, ...).
struct PayNonePhased {
    static constexpr int mode = kModePlain;
    static constexpr int group = 6;
    static constexpr bool phased = true;
    static constexpr bool fat1024 = true;  // (update windows exist for big tables: 4096-key tiles where the slices are many, window_scatter)
    const PhaseDesc *ph;       // device array [nph + 1]: the PIECES
    uint32_t nph;
    uint32_t *snap;            // [phases][nbuckets][nwg]
    uint64_t first;            // the first piece's first key (what the prefetches of a workgroup without tiles fall back to)
    __device__ __forceinline__ uint32_t operator()(uint64_t, uint64_t) const { return 0; }
};
struct PayUnit {   // unit-weight counter adds: 8 probes per group, 16-bit slice-local cell indices (slices <= 2^15 cells)
    static constexpr int mode = kModePlain;
    static constexpr int group = 8;
    __device__ __forceinline__ uint32_t operator()(uint64_t, uint64_t) const { return 0; }
};
struct PayWeight {
    static constexpr int group = 4;
    static constexpr int mode = kModeInline;
    const uint32_t *w;  // int32 / uint32 bit patterns; null = unit weights (level 1 of the two-level path)
    // Fused accounting (round 2; it used to be a separate pass over the weights, k_weight_sum: 18 us per 10 M): pass 1 reads
    // every weight anyway, so every workgroup also leaves (sum w, sum |w|) of its keys in tally[blockIdx.x] -- plain stores,
    // no atomics: 768 same-address atomics at the kernel's end cost as much as the pass they replaced -- and the one-block
    // k_tally_fold between pass 1 and pass 2 adds the slots into the handle's device counters.  null: nothing to account.
    ulonglong4 *tally = nullptr;   // (sum w, sum |w|, weights outside 0 .. 15, -)
    int weights_signed = 0;
    __device__ __forceinline__ uint32_t operator()(uint64_t i, uint64_t) const { return w ? w[i] : 1u; }
};
// Weighted counter adds whose weights fit four bits (1 .. 15: what a count-min sketch is fed in practice; round 3): PayNone's 6 x 20-bit
// groups with field = weight << 15 | 15-bit slice-local cell -- 2.7 bytes per probe instead of PayWeight's 4, and the plain stage (no
// per-group slice ids).  The LDS stage holds weight << 27 | cell (tables below 2^27 cells; bit 31 stays the pad marker).  A weight of 0
// travels as a field that adds nothing; one of 16 or more (any negative one) goes to the table directly -- exact saturating add -- and
// leaves such a field behind.  The host picks this format when the previous weighted batches had no such weight (PayWeight::tally's
// third count, published to a pinned page by k_tally_fold).
constexpr uint32_t kSmallWeightBits = 4, kSmallWeightShift = 27, kSmallCellMask = (1u << kSmallWeightShift) - 1;
struct PayWeightSmall {
    static constexpr int mode = kModePlain;
    static constexpr int group = 6;
    static constexpr bool weighted_plain = true;
    const uint32_t *w;
    ulonglong4 *tally = nullptr;
    int weights_signed = 0;
    __device__ __forceinline__ uint32_t operator()(uint64_t i, uint64_t) const { return w[i]; }
};
struct PayZero {   // level 1 of the two-level Bloom insert: 4 x 32-bit (0 << shift | bit index inside the coarse bucket)
    static constexpr int group = 4;
    static constexpr int mode = kModeInline;
    __device__ __forceinline__ uint32_t operator()(uint64_t, uint64_t) const { return 0u; }
};
// Bloom lookups: 4 probes per group, word = tile bit << 31 | key index within the tile << shift | bit index within the slice.
// The four top bits of a group spell the tile's ordinal inside its workgroup's sequence (tile = ordinal * nwg + wg, wg = the
// segment's workgroup), so a round holds at most 16 tiles per workgroup; the tile is at most 2^(31 - shift) keys
// (max_tile = 2048 keys: slices of 2^20 bits).  Trailing slots of a run's last group repeat its first probe (a lookup
// probe may be tested twice), so there is no pad marker in HBM.
struct PayKeyId {
    static constexpr int group = 4;
    static constexpr int mode = kModeKeyed;
    static constexpr int max_tile = 2048;
    static constexpr int slice_shift = 20;  // largest slice (log2 bits) these probes address: max_tile << slice_shift <= 2^31
    static constexpr uint32_t max_tiles_per_wg = 16;
    __device__ __forceinline__ uint32_t operator()(uint64_t i, uint64_t base) const { return (uint32_t)(i - base); }
};

// fallback for a probe that cannot go through the bucket buffer: apply it straight to the table
struct SpillBloomOr {
    uint32_t *tab;
    __device__ __forceinline__ void operator()(uint32_t idx, uint32_t) const { atomicOr(tab + (idx >> 5), 1u << (idx & 31)); }
};
template <bool SIGNED>
struct SpillCounter {  // saturating CAS add (countminsketch.py:280-284,312-316 / countingbloom.py:149-153)
    uint32_t *tab;
    bool unit, neg;
    unsigned long long *sat_ctr;
    // The transactional CBF decrement (psk_capi.hip cbf_remove_exact): opt 1 = wrapping subtraction that raises `flag` when a counter
    // would go below zero or is frozen (the batch is then order-dependent: undone and replayed in order), opt 2 = its inverse.
    uint32_t *flag = nullptr;
    int opt = 0;
    __device__ __forceinline__ void operator()(uint32_t idx, uint32_t w) const
    {
        const uint32_t v = unit ? 1u : w;
        if (!SIGNED && opt == 1) {
            const uint32_t old = atomicSub(tab + idx, v);
            if (old < v || old == 0xFFFFFFFFu) *flag = 1u;
            return;
        }
        if (!SIGNED && opt == 2) {
            atomicAdd(tab + idx, v);
            return;
        }
        if (SIGNED) cms_sat_add((int32_t *)tab + idx, neg ? -(int64_t)(int32_t)v : (int64_t)(int32_t)v, sat_ctr);
        else if (neg) cbf_sat_sub(tab + idx, v, sat_ctr - 1);  // (the violations tally sits right before the saturation tally)
        else cbf_sat_add(tab + idx, v, sat_ctr);
    }
};
struct SpillRaiseFlagCounter {  // optimistic CBF decrement (psk_nibble.hpp): a segment that overflows makes the whole batch take the exact path
    uint32_t *flag;
    __device__ __forceinline__ void operator()(uint32_t, uint32_t) const { *flag = 1u; }
};
struct SpillBloomTest {  // lookup probe: test it directly (bloom.py:269-271); `key` is the index inside this round
    const uint32_t *tab;
    uint8_t *out;
    uint32_t *defer;  // split lookup (psk_bloom_check_begin): the table is not final yet -- just raise the flag, the
                      // finish step then re-checks the whole round with the direct kernel
    __device__ __forceinline__ void operator()(uint32_t idx, uint32_t key) const
    {
        if (defer) { *defer = 1u; return; }
        if (((tab[idx >> 5] >> (idx & 31)) & 1u) == 0) out[key] = 0;
    }
};

struct SpillBloomFlag {  // PayTileTag probe of an overflowing segment: test it directly (bloom.py:269-271), a clear bit flags the probe's tile
    const uint32_t *tab;
    uint32_t *tileflag;
    uint32_t gen;  // the round's generation number: "flagged" = holds this value (k_bloom_test_flag)
    uint32_t defer;  // split lookup (psk_bloom_check_begin): the table is not final yet -- flag the tile whatever the table says (it is then
                     // re-checked key by key at the finish: exact)
    __device__ __forceinline__ void operator()(uint32_t idx, uint32_t tile) const
    {
        if (defer || ((tab[idx >> 5] >> (idx & 31)) & 1u) == 0) tileflag[tile] = gen;
    }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a fence, for which hipcc emits
// s_waitcnt vmcnt(0): every barrier of pass 1 would then drain the key prefetch (a full HBM latency per tile)
// and the previous tile's write-out stores.  Pass 1 exchanges data between waves through LDS only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wave64 inclusive prefix sum on the DPP network (row_shr 1/2/4/8 inside each row of 16, then row_bcast 15 / 31 across
// rows): 6 dependent VALU adds instead of 6 ds_bpermute round trips through the LDS pipe
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return x;
}

// block-wide exclusive scan of one uint32 per thread (NT threads = NT/64 waves)
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wave_tot /*LDS[16]*/, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) wave_tot[wid] = inc;
    lds_barrier();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const uint32_t t = wave_tot[w];
        if (w < wid) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

// ------------------------------------------------------------------------------------ pass 1
// KT = hashes computed per key (>= k, compile time so the probes stay in registers).
// dynamic LDS: hist[2][B] | off[B] | delta[B] | cur[B] | wave_tot[16] | profile[24] | stage (1 or 2 words per probe)
template <class Pay, int KT, int NT_ = kPartThreads>
struct PartTile {
    static constexpr bool pair = Pay::mode != kModePlain;           // probes carry a payload (weight / key id): the stage holds
                                                                    // the final 32-bit word and a per-group slice id (gb[])
    static constexpr int GS = Pay::group;                           // probes per 16-byte output group
    // Probes per thread and tile.  k <= 8 comes in two shapes with the same tile (2048 keys for k = 7: 14 K probes, so the
    // per-tile fixed costs -- scan, barriers -- are paid equally often): 1024 threads x 16 probes, one workgroup per CU, and
    // 512 threads x 32 probes (113 VGPRs), TWO workgroups per CU when two LDS stages fit (the host's choice, launch_scatter).
    // With one workgroup per CU the VALU-bound hash phase and the LDS / latency-bound scan, sort and write-out phases run
    // strictly one after the other (ablation: hashing alone 94 us, everything else 138 us, both 203 us per 10 M keys); two
    // workgroups drift out of phase and overlap them: insert 185 -> 178 us.  (Two 512-thread workgroups with 16 probes per
    // thread -- 1024-key tiles, twice the tiles -- gave that gain back in fixed costs: round 1.)  Large k: 512 threads, 16
    // or 32 probes, one key per thread.
    // (Pay::fat512: the Bloom insert only.  Keyed lookups and weighted adds measured the same either way, unit counter adds
    // 2 % and the counter lookups -- more registers per key: perm[] positions -- 5 % worse: they keep 16 probes per thread;
    // scripts/ab_shape.py.)
    static constexpr int PP = (KT <= 8 && ((NT_ == 512 && pay_fat512<Pay>::value) || (NT_ == 1024 && pay_fat1024<Pay>::value))) ? kPartProbes : kPartProbes / 2;
    static constexpr int KPT_CAP = (NT_ == 512 || PP == kPartProbes) ? 6 : 1 << 20;  // registers: 2 words per probe + 4 per prefetched key
    static constexpr int KPT1 = PP / KT >= 1 ? PP / KT : 1;
    static constexpr int KPT0 = KPT1 < KPT_CAP ? KPT1 : KPT_CAP;
    static constexpr int KPT_PAY = pay_max_tile<Pay>::value / NT_;  // (keyed probes: the key index inside the tile has 11 bits)
    static constexpr int KPT = KPT0 < KPT_PAY ? KPT0 : KPT_PAY;     // keys per thread per tile
    static constexpr int NT = NT_;
    static constexpr int TILE = NT * KPT;                           // keys per tile
};

// Write-out of ONE 16-byte group of the sorted LDS stage: lane = group gi of the tile; its first probe is always
// real (pads trail), so it names the slice.  delta[b] turns the stage group index into the slot of my segment.
template <class Pay, class Spill>
__device__ __forceinline__ void emit_group(const uint32_t *stage, const uint32_t *gb, const uint32_t *delta, const PartGeom &g, uint32_t mask,
                                           uint32_t gi, uint64_t tile, uint64_t base, const Spill &spill, uint4 *wg_buckets)
{
    // wg_buckets: this workgroup's nbuckets segments (seg_index(g, 0, blockIdx.x) * segcap groups into the buffer); slice and
    // segment capacity are both < 2^24 and a workgroup's share of the buffer is far below 2^32 groups: 24-bit multiply-add,
    // 32-bit group index (the 64-bit (wg * B + b) * segcap + slot cost nine VALU instructions per group)
    constexpr int GS = Pay::group;
    if constexpr (Pay::mode == kModePlain) {
        // the LDS stage holds full cell indices (the first one names the slice); HBM gets them packed
        uint32_t c[GS];
        if constexpr (GS == 6) {
            const uint2 a0 = reinterpret_cast<const uint2 *>(stage)[3 * gi];
            const uint2 a1 = reinterpret_cast<const uint2 *>(stage)[3 * gi + 1];
            const uint2 a2 = reinterpret_cast<const uint2 *>(stage)[3 * gi + 2];
            c[0] = a0.x; c[1] = a0.y; c[2] = a1.x; c[3] = a1.y; c[4] = a2.x; c[5] = a2.y;
        } else {
            const uint4 a0 = reinterpret_cast<const uint4 *>(stage)[2 * gi];
            const uint4 a1 = reinterpret_cast<const uint4 *>(stage)[2 * gi + 1];
            c[0] = a0.x; c[1] = a0.y; c[2] = a0.z; c[3] = a0.w; c[4] = a1.x; c[5] = a1.y; c[6] = a1.z; c[7] = a1.w;
        }
        constexpr bool WP = pay_weighted_plain<Pay>::value;  // stage word = weight << 27 | cell
        const uint32_t b = (WP ? c[0] & kSmallCellMask : c[0]) >> g.shift;
        const uint32_t slot = delta[b] + gi;
        if (slot < g.segcap) {
            uint4 o;
            if constexpr (GS == 6) {
                // two 64-bit halves: 3 x 20-bit local indices + the number of valid ones in bits 60..63
                // (cells are < 2^31 and the pad is all ones: the sign bits of c[1..5] count the pads -- no compare / carry chains)
                const uint32_t nv = 6u - ((c[1] >> 31) + (c[2] >> 31) + (c[3] >> 31) + (c[4] >> 31) + (c[5] >> 31));
                const uint32_t n0 = nv < 3 ? nv : 3, n1 = nv - n0;
                // (weighted: field = weight << 15 | cell in the slice; a pad's field is never read -- n0 / n1 say how many are valid)
                // (a pad -- all ones -- becomes the all-zero field: weight 0, so pass 2 needs no count test per field)
                auto fld = [&](uint32_t x) -> unsigned long long {
                    return WP ? (unsigned long long)(((x & mask) | ((x >> kSmallWeightShift) << 15)) & ~(uint32_t)((int32_t)x >> 31)) : (unsigned long long)(x & mask);
                };
                // (PayTileTag: `tile` is the tile's ordinal inside this workgroup's sequence, two bits of it above each half's count)
                const uint32_t t0 = pay_tile_tag<Pay>::value ? ((uint32_t)tile & 3u) << 2 : 0u, t1 = pay_tile_tag<Pay>::value ? (uint32_t)tile & 12u : 0u;
                const unsigned long long h0 = fld(c[0]) | (fld(c[1]) << 20) | ((fld(c[2]) & 0xFFFFFull) << 40) | ((unsigned long long)(n0 | t0) << 60);
                const unsigned long long h1 = fld(c[3]) | (fld(c[4]) << 20) | ((fld(c[5]) & 0xFFFFFull) << 40) | ((unsigned long long)(n1 | t1) << 60);
                o = make_uint4((uint32_t)h0, (uint32_t)(h0 >> 32), (uint32_t)h1, (uint32_t)(h1 >> 32));
            } else {
                auto h16 = [&](uint32_t x) -> uint32_t { return (x & mask) | ((uint32_t)((int32_t)x >> 31) & 0xFFFFu); };  // pad (all ones) -> 0xFFFF
                o = make_uint4(h16(c[0]) | (h16(c[1]) << 16), h16(c[2]) | (h16(c[3]) << 16),
                               h16(c[4]) | (h16(c[5]) << 16), h16(c[6]) | (h16(c[7]) << 16));
            }
            wg_buckets[__umul24(b, g.segcap) + slot] = o;
        } else {  // segment full: exact fallback, probe by probe
#pragma unroll
            for (int e = 0; e < GS; ++e)
                if (c[e] != kPadProbe) {
                    if constexpr (WP) spill(c[e] & kSmallCellMask, c[e] >> kSmallWeightShift);
                    else if constexpr (pay_tile_tag<Pay>::value) spill(c[e], (uint32_t)tile * gridDim.x + blockIdx.x);
                    else spill(c[e], 0u);
                }
        }
    } else if constexpr (Pay::mode == kModeInline) {
        const uint4 e = reinterpret_cast<const uint4 *>(stage)[gi];  // four final words (weight << shift | cell in slice)
        const uint32_t b = gb[gi];
        const uint32_t slot = delta[b] + gi;
        if (slot < g.segcap) {
            wg_buckets[__umul24(b, g.segcap) + slot] = e;
        } else {  // segment full: exact saturating add on the table, probe by probe
            const uint32_t w[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
            for (int x = 0; x < 4; ++x)
                if (w[x] != kPadProbe) spill((b << g.shift) | (w[x] & mask), w[x] >> g.shift);
        }
    } else {  // keyed: `tile` is the tile's ordinal inside this workgroup's sequence (< 16), spelled by the four top bits
        const uint4 e = reinterpret_cast<const uint4 *>(stage)[gi];
        const uint32_t b = gb[gi];
        const uint32_t slot = delta[b] + gi;
        const uint32_t w[4] = {e.x, e.y, e.z, e.w};  // LDS words are 31 bits; kPadProbe marks the unused trailing slots
        if (slot < g.segcap) {
            uint32_t o[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) o[x] = (w[x] == kPadProbe ? w[0] : w[x]) | ((((uint32_t)tile >> x) & 1u) << 31);
            // (v_cmp + v_cndmask + v_or per word; the branch-free sign-mask form compiles to max / ashr / and / or3 and measured slower)
            wg_buckets[__umul24(b, g.segcap) + slot] = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int x = 0; x < 4; ++x)
                if (w[x] != kPadProbe) spill((b << g.shift) | (w[x] & mask), (uint32_t)base + (w[x] >> g.shift));
        }
    }
}

// (the second launch bound is hipcc's "min waves per SIMD": 4 = two workgroups per CU = at most 128 VGPRs; without it
// small source changes tip the keyed instantiation to 133 VGPRs and one workgroup per CU, 15 % slower)
template <class Src, class IdxFn, class Pay, class Spill, int KT, int NTHREADS>
__global__ __launch_bounds__(NTHREADS, (KT <= 8 ? 4 : 1)) void k_part_scatter(Src src, IdxFn idxfn, Pay pay, Spill spill, PartGeom g,
                                                               uint64_t n, uint32_t *segcnt, uint4 *buckets)
{
    using T = PartTile<Pay, KT, NTHREADS>;
    constexpr int KPT = T::KPT, TILE = T::TILE, GS = T::GS, NT = T::NT;
    constexpr bool PAIR = T::pair;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t dbg = kBenchKnobs ? g.dbg : 0u;  // folds to 0 in the shipped build
    const uint32_t B = g.nbuckets;
    constexpr bool PHASED = pay_is_phased<Pay>::value;  // n = tiles x tile size (virtual: every phase is padded to whole tiles)
    const uint64_t nlim = n;                   // keys the source holds (phased lists: tiles x tile size, see tile_base)
    const uint64_t last = nlim ? nlim - 1 : 0; // index the clamped prefetches fall back to (the key buffer holds at least one key)
    uint32_t *hist0 = smem;  // two copies: tile t counts in one while the scan phase of tile t zeroes the other
    uint32_t *off = hist0 + 2 * B;
    uint32_t *delta = off + B;
    uint32_t *cur = delta + B;  // groups already appended to my segment of every slice, across all my tiles
    uint32_t *wave_tot = cur + B;
    unsigned long long *t_acc = reinterpret_cast<unsigned long long *>(wave_tot + 16);  // phase profile (dbg & 32), 12 slots
    uint32_t *stage = wave_tot + 16 + 24;
    // slice id of every stage group (payload modes); sits behind the stage: TILE * k probes + (GS - 1) pads per slice
    // (sized by the tile the host chose -- g.tile <= TILE: evened tiles, or tiles cut down so that the stage fits the LDS at 2048 slices)
    const uint32_t stage_cap = (g.tile * (g.k < (uint32_t)KT ? g.k : (uint32_t)KT) + (uint32_t)(GS - 1) * B + 3u) & ~3u;
    uint32_t *gb = stage + stage_cap;
    // keys of different lengths (src_sorted): class counts, class offsets and the slot order of the tile's length sort, behind everything else
    // (+ the keys' descriptors in slot order, 16 bytes each, 16-byte aligned)
    // (aligned by index arithmetic on the LDS base: an integer round trip would make these FLAT accesses)
    uint32_t *sort_hist = smem + ((uint32_t)((gb + (PAIR ? stage_cap / GS + 4 : 0)) - smem) + 3u & ~3u);
    uint32_t *sort_off = sort_hist + kSortBins;
    uint4 *sort_keys = reinterpret_cast<uint4 *>(sort_off + kSortBins);
    uint16_t *sort_order = reinterpret_cast<uint16_t *>(sort_keys + T::TILE);
    // KT other than the round-up sizes 8 / 16 / 32 is an exact instantiation (with_kt): k == KT, and every per-probe
    // "j < k" test below folds away (28 exec-mask branch sequences per tile for k = 7)
    constexpr bool kExactK = KT != 8 && KT != 16 && KT != 32;
    const uint32_t k = kExactK ? (uint32_t)KT : g.k;
    const uint32_t mask = (1u << g.shift) - 1;
    // keys per tile: TILE, or fewer when the host evened the tiles out over the workgroups (launch_scatter_nt); a multiple of 64
    const uint32_t tk = g.tile;
    const uint64_t ntiles = (n + tk - 1) / tk;
    uint4 *wg_buckets = buckets + seg_index(g, 0, blockIdx.x) * g.segcap;  // my segment of slice 0; slice b: + b * segcap

    for (uint32_t b = threadIdx.x; b < B; b += NT) cur[b] = g.append ? segcnt[(uint64_t)b * g.nwg + blockIdx.x] : 0u;
    for (uint32_t b = threadIdx.x; b < 2 * B; b += NT) hist0[b] = 0;
    if constexpr (src_sorted<Src>::value) {
        for (uint32_t b = threadIdx.x; b < (uint32_t)kSortBins; b += NT) sort_hist[b] = 0;
    }
    uint32_t parity = 0;
    lds_barrier();

    // Software pipeline over tiles: the NEXT tile's keys are loaded right after this tile's hash phase and
    // pinned before this tile's write-out stores are issued (vmcnt counts loads and stores in order on CDNA4:
    // a key load waited for AFTER the stores would also wait for ~300 KB of stores to drain).
    // list index of tile t's first key and of its last one (`lastk`: what the tile's loads are clamped to); phased lists: p (a piece at
    // or before t's) moves on to t's piece -- uniform, scalar loads
    auto tile_base = [&](uint64_t t, uint32_t &p, uint64_t &lastk) -> uint64_t {
        if constexpr (PHASED) {
            if (t >= ntiles) {  // (past the end: every lane clamps to a key that exists)
                lastk = pay.first;
                return pay.first;
            }
            while (p + 1 < pay.nph && t >= (uint64_t)pay.ph[p + 1].tile0) ++p;
            const uint64_t b = (uint64_t)(pay.ph[p].key_off + (long long)(t * tk));
            const uint64_t left = pay.ph[p].nkeys - (t - pay.ph[p].tile0) * tk;  // keys of the piece from this tile on
            lastk = b + (left < tk ? left : tk) - 1;
            return b;
        } else {
            lastk = last;
            return t * tk;
        }
    };
    uint32_t ph_cur = 0;  // phased lists: first phase whose end-of-phase counts I have not written yet
    auto write_snapshot = [&](uint32_t p) {
        if constexpr (PHASED) {
            // (bit 31: the phase removes -- the fold reads the phase's type off the counts it loads anyway, phases ahead of their use)
            const uint32_t f = pay.ph[p].remove;
            if (!(f & kPieceEndsPhase)) return;  // (uniform)
            const uint32_t type_bit = (f & 1u) ? 0x80000000u : 0u;
            for (uint32_t b = threadIdx.x; b < B; b += NT) pay.snap[((uint64_t)(f >> 8) * B + b) * g.nwg + blockIdx.x] = cur[b] | type_bit;
        }
    };
    typename Src::Key kcur[KPT];
    uint64_t b0 = 0;
    if (kPartPipeline) {
        uint32_t p0 = 0;
        uint64_t l0;
        b0 = tile_base(blockIdx.x, p0, l0);
#pragma unroll
        for (int q = 0; q < KPT; ++q) {
            const uint64_t i = b0 + (uint64_t)q * NT + threadIdx.x;
            kcur[q] = src.load(i < l0 ? i : l0);  // coalesced; clamped, never branched around (a conditional load
        }                                              // makes hipcc wait vmcnt(0) per element: serial round trips)
    }
    // Keys of different lengths (KeysVarlen): a wave hashes for as long as its LONGEST key lasts, every shorter key's lane idling -- with
    // lengths 4 .. 40 (mean 16) the hash phase took 4 x the time of 16-byte keys.  So a tile's keys are handed to the lanes in order of
    // length: a counting sort of the prefetched lengths (64 classes: exact up to 47 elements; LDS atomics, one wave scans) gives every key a
    // slot, the keys' descriptors go through LDS into slot order, and lane (q, thread) takes the key of slot q * NT + thread -- for odd q
    // with the waves in reverse order, so that every wave gets short AND long keys.  Everything behind the hash phase names a key by its
    // index i.  The sort of tile t + 1 runs in front of tile t's write-out, on the prefetched descriptors, and ends with the requests for
    // the keys' first windows: they land under the write-out (as dependent loads in front of every key's chains -- offsets, then windows,
    // key after key -- they cost a tile ~10 us, as much as the rest of it).
    constexpr bool SORTED = src_sorted<Src>::value;
    static_assert(!(SORTED && PHASED), "phased lists hold 16-byte keys");
    static_assert(!SORTED || sizeof(typename Src::Key) == 16, "the length sort moves 16-byte key descriptors");
    uint32_t slot[KPT], slot_n[SORTED ? KPT : 1];  // of this tile / of the next one (sorted in the middle of this tile)
#pragma unroll
    for (int q = 0; q < KPT; ++q) slot[q] = (uint32_t)q * NT + threadIdx.x;
    typename Src::Key ks[SORTED ? KPT : 1];
    uint4 kfirst[SORTED ? KPT : 1];
    auto sort_tile = [&](uint64_t nb, uint64_t ne) {  // keys [nb, ne) in kcur (natural order) -> slot_n / ks / kfirst
        if constexpr (SORTED) {
            // (every class has kSortSub counters, one per lane mod kSortSub: the lanes of a wave -- neighbours in the batch, often of ONE length --
            // would otherwise queue at one LDS address, and a returning atomic there takes its ~16 cycles per lane, one after the other:
            // 64 classes x 1 counter cost a 2048-key tile ~12 us, as much as everything else in it)
            uint32_t cls[KPT], crank[KPT];
            const bool nosort = (g.dbg & kGeomNoSortBit) != 0;  // (uniform; option "ragged_sort" = 0: batch order -- the descriptors still travel)
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const uint64_t i = nb + (uint32_t)q * NT + threadIdx.x;
                cls[q] = (i < ne ? Src::len_class(kcur[q]) : 63u) * (uint32_t)kSortSub + (threadIdx.x & (uint32_t)(kSortSub - 1));  // past the tile's end: behind all others
                if (!nosort) crank[q] = atomicAdd(&sort_hist[cls[q]], 1u);
            }
            if (!nosort) {
            lds_barrier();
            {   // exclusive scan over the counters in class order (a thread takes kSortBins / NT neighbours)
                constexpr int BPT = kSortBins / NT > 0 ? kSortBins / NT : 1;
                uint32_t v[BPT], sum = 0;
#pragma unroll
                for (int c = 0; c < BPT; ++c) {
                    const uint32_t b = threadIdx.x * BPT + c;
                    v[c] = b < (uint32_t)kSortBins ? sort_hist[b] : 0u;
                    sum += v[c];
                }
                uint32_t total;
                uint32_t run = block_exclusive_scan<NT>(sum, wave_tot, &total);
#pragma unroll
                for (int c = 0; c < BPT; ++c) {
                    const uint32_t b = threadIdx.x * BPT + c;
                    if (b < (uint32_t)kSortBins) {
                        sort_off[b] = run;
                        sort_hist[b] = 0;  // (the next tile counts behind more barriers)
                    }
                    run += v[c];
                }
            }
            lds_barrier();
            }
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const uint32_t pos = nosort ? (uint32_t)q * NT + threadIdx.x : sort_off[cls[q]] + crank[q];
                uint4 d;
                __builtin_memcpy(&d, &kcur[q], 16);
                sort_keys[pos] = d;
                sort_order[pos] = (uint16_t)((uint32_t)q * NT + threadIdx.x);
            }
            lds_barrier();
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const uint32_t pos = (uint32_t)q * NT + ((q & 1) ? (uint32_t)(NT - 64) - (threadIdx.x & ~63u) + (threadIdx.x & 63u) : threadIdx.x);
                const uint4 d = sort_keys[pos];
                __builtin_memcpy(&ks[q], &d, 16);
                slot_n[q] = sort_order[pos];
            }
#pragma unroll
            for (int q = 0; q < KPT; ++q) kfirst[q] = src.first(ks[q]);
        }
    };
    if constexpr (SORTED) {
        const uint64_t e0 = blockIdx.x < ntiles ? (b0 + tk < nlim ? b0 + tk : nlim) : b0;
        sort_tile(b0, e0);
#pragma unroll
        for (int q = 0; q < KPT; ++q) slot[q] = slot_n[q];
    }

    // phase profile (dbg & 32): lane 0 of wave 0 accumulates s_memtime deltas per phase; bench-only
    unsigned long long t_prev = 0;  // (the accumulators live in LDS: 12 x 64-bit in registers cost 24 VGPRs on every lane)
    if ((dbg & 32) && threadIdx.x < 12) t_acc[threadIdx.x] = 0;
#define PSK_TICK(ph)                                                                   \
    if ((dbg & 32) && threadIdx.x == 0) {                                            \
        const unsigned long long t_now = __builtin_readcyclecounter();                 \
        t_acc[ph] += t_now - t_prev;                                                   \
        t_prev = t_now;                                                                \
    }
    if ((dbg & 32) && threadIdx.x == 0) t_prev = __builtin_readcyclecounter();

    long long tally_s = 0;            // fused weight accounting (PayWeight::tally)
    unsigned long long tally_a = 0;
    uint32_t tally_b = 0;             // weights outside 0 .. 15 (the next batch's choice of probe format, PayWeightSmall)
    uint32_t ordinal = ~0u;  // of the tile inside this workgroup's sequence (keyed probes carry it)
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        ++ordinal;
        uint32_t *hist = hist0 + (size_t)(parity ? B : 0);
        uint32_t *hist_next = hist0 + (size_t)(parity ? 0 : B);
        parity ^= 1u;
        PSK_TICK(1);
        if constexpr (PSK_EXP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if constexpr (PSK_EXP_PRIO == 2) __builtin_amdgcn_s_setprio(3);

        // ---- hash + histogram: rank = my position among this tile's probes of the same slice
        uint32_t idx[KPT][KT], rank[KPT][KT], payload[KPT];
        uint32_t fold = 0;
        uint32_t ph_tile = ph_cur;
        uint64_t tile_last;
        const uint64_t base = tile_base(tile, ph_tile, tile_last);
        uint64_t tile_end = base + tk < nlim ? base + tk : nlim;
        if constexpr (PHASED) {
            // cur[] is what my segments held after my last tile, i.e. at the end of every piece before this tile's
            for (; ph_cur < ph_tile; ++ph_cur) write_snapshot(ph_cur);
            tile_end = tile_last + 1;
        }
        // Pay::keep (masked batches): a key whose flag is 0 sends no probes.  The flags are requested up front and first consumed
        // behind the key's hash chains, which hides the load.
        constexpr bool KEEP = pay_has_keep<Pay>::value;
        uint32_t kw[KEEP ? KPT : 1];
        if constexpr (KEEP) {
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const uint64_t i = base + slot[q];
                kw[q] = pay.keep(i < nlim ? i : last);
            }
        }
        auto kept = [&](int q) -> bool { if constexpr (KEEP) return kw[q] != 0; else return true; };
#pragma unroll
        for (int q = 0; q < KPT; ++q) {
            const uint64_t i = base + slot[q];
            if (i < tile_end) {
                typename Src::Key key;
                if constexpr (SORTED) key = ks[q];
                else key = kPartPipeline ? kcur[q] : src.load(i);
                // hash<G> / hash32<G> of the key; sources with a first window (SORTED) take it along
                auto hash32_of = [&](auto gtag, uint32_t s0, uint32_t (&hh)[decltype(gtag)::value]) {
                    if constexpr (SORTED) src.template hash32_first<decltype(gtag)::value>(key, kfirst[q], i, s0, hh);
                    else src.template hash32<decltype(gtag)::value>(key, i, s0, hh);
                };
                auto hash64_of = [&](auto gtag, uint32_t s0, uint64_t (&hh)[decltype(gtag)::value]) {
                    if constexpr (SORTED) src.template hash_first<decltype(gtag)::value>(key, kfirst[q], i, s0, hh);
                    else src.template hash<decltype(gtag)::value>(key, i, s0, hh);
                };
                if (PAIR || pay_weighted_plain<Pay>::value) payload[q] = pay(i, base);
                if constexpr (IdxFn::lo32) {  // 32-bit chains (power-of-two table: the upper hash halves are dead)
                    uint32_t h[KT];
                    if (dbg & 4) {
                        for (int j = 0; j < KT; ++j) h[j] = (uint32_t)(((uint64_t)(i * 2654435761u + j * 40503u) * 0x9E3779B97F4A7C15ULL) >> 13);
                    } else if constexpr (KT != 8 && (KT <= 8 || KT % 4 != 0)) {  // exact k: every chain is live
                        hash32_of(std::integral_constant<int, KT>{}, 0u, h);
                    } else if (KT == 8 && k > 4) {  // (k = 5 .. 8 through the round-up kernel: ONE walk of the key -- two walks of four chains cost the
                        hash32_of(std::integral_constant<int, KT>{}, 0u, h);  // layouts that re-read their windows 20 %)
                    } else {  // KT is k rounded up: run the chains four at a time and skip the groups past k
#pragma unroll
                        for (int s0 = 0; s0 < KT; s0 += 4) {
                            uint32_t hh[4] = {0, 0, 0, 0};
                            if ((uint32_t)s0 < k) hash32_of(std::integral_constant<int, 4>{}, (uint32_t)s0, hh);
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[s0 + e] = hh[e];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < KT; ++j) {
                        if ((uint32_t)j < k) {
                            idx[q][j] = idxfn.from32((uint32_t)j, h[j]);
                            if (dbg & 2) { fold ^= idx[q][j]; continue; }  // bench-only: hashing alone
                            if (kept(q)) rank[q][j] = atomicAdd(&hist[idx[q][j] >> g.shift], 1u);  // ds_add_rtn_u32
                        }
                    }
                } else {
                    uint64_t h[KT];
                    if (dbg & 4) {
                        for (int j = 0; j < KT; ++j) h[j] = ((uint64_t)(i * 2654435761u + j * 40503u) * 0x9E3779B97F4A7C15ULL) >> 13;
                    } else if constexpr (KT != 8 && (KT <= 8 || KT % 4 != 0)) {  // exact k: every chain is live
                        hash64_of(std::integral_constant<int, KT>{}, 0u, h);
                    } else if (KT == 8 && k > 4) {
                        hash64_of(std::integral_constant<int, KT>{}, 0u, h);
                    } else {
#pragma unroll
                        for (int s0 = 0; s0 < KT; s0 += 4) {
                            uint64_t hh[4] = {0, 0, 0, 0};
                            if ((uint32_t)s0 < k) hash64_of(std::integral_constant<int, 4>{}, (uint32_t)s0, hh);
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[s0 + e] = hh[e];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < KT; ++j) {
                        if ((uint32_t)j < k) {
                            idx[q][j] = idxfn((uint32_t)j, h[j]);
                            if (kept(q)) rank[q][j] = atomicAdd(&hist[idx[q][j] >> g.shift], 1u);  // ds_add_rtn_u32
                        }
                    }
                }
            }
        }
        if (dbg & 2) {  // bench-only: keep the hashes alive, skip the rest of the tile (uniform)
            if (fold == 0x12345u) segcnt[0] = fold;
            if (kPartPipeline) {
                uint32_t pn = ph_tile;
                uint64_t ln;
                const uint64_t nbase = tile_base(tile + gridDim.x, pn, ln);
#pragma unroll
                for (int q = 0; q < KPT; ++q) {
                    const uint64_t i = nbase + (uint64_t)q * NT + threadIdx.x;
                    kcur[q] = src.load(i < ln ? i : ln);
                }
                if constexpr (SORTED) {
                    const uint64_t tn = tile + gridDim.x;
                    sort_tile(nbase, tn < ntiles ? (nbase + tk < nlim ? nbase + tk : nlim) : nbase);
#pragma unroll
                    for (int q = 0; q < KPT; ++q) slot[q] = slot_n[q];
                }
            }
            continue;
        }
        PSK_TICK(9);
        lds_barrier();
        PSK_TICK(2);
        if constexpr (PSK_EXP_PRIO == 1) __builtin_amdgcn_s_setprio(3);
        if constexpr (PSK_EXP_PRIO == 2) __builtin_amdgcn_s_setprio(0);
        if constexpr (PSK_EXP_PRIO == 3) { if (threadIdx.x < 64) __builtin_amdgcn_s_setprio(3); }

        // ---- prefetch the next tile's keys (consumed -- pinned -- before the write-out below)
        uint64_t nbase = 0;
        if (kPartPipeline) {
            uint32_t pn = ph_tile;
            uint64_t ln;
            nbase = tile_base(tile + gridDim.x, pn, ln);
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const uint64_t i = nbase + (uint64_t)q * NT + threadIdx.x;
                kcur[q] = src.load(i < ln ? i : ln);  // unconditional (clamped) on purpose, see above
            }
        }

        // ---- exclusive scan of the histogram; advance my segment cursors (LDS only).
        // Every run is padded to whole groups of GS probes with kPadProbe, so one lane moves one 16-byte
        // group from an aligned stage position to an aligned segment slot (dword stores were store-issue bound).
        uint32_t mine[kPartScanPerThread], s = 0;
#pragma unroll
        for (int c = 0; c < kPartScanPerThread; ++c) {
            const uint32_t b = threadIdx.x * kPartScanPerThread + c;
            mine[c] = b < B ? hist[b] : 0;
            s += (mine[c] + GS - 1) / GS * GS;
        }
        for (uint32_t b = threadIdx.x; b < B; b += NT) hist_next[b] = 0;  // ready for the next tile
        PSK_TICK(6);
        uint32_t tile_probes;  // padded
        uint32_t run;
        if (B <= 64 * kPartScanPerThread) {
            // all slices live in wave 0 (4 per lane): a wave scan, no cross-wave step, one barrier less
            if (threadIdx.x < 64) {
                const uint32_t inc = wave_inclusive_scan(s);
                run = inc - s;
                if (threadIdx.x == 63) wave_tot[0] = inc;
            } else {
                run = 0;
            }
        } else {
            run = block_exclusive_scan<NT>(s, wave_tot, &tile_probes);
        }
        PSK_TICK(7);
#pragma unroll
        for (int c = 0; c < kPartScanPerThread; ++c) {
            const uint32_t b = threadIdx.x * kPartScanPerThread + c;
            if (b < B) {
                const uint32_t padded = (mine[c] + GS - 1) / GS * GS;
                off[b] = run;               // stage position of the run, in probes (a multiple of GS)
                const uint32_t c0 = cur[b];
                delta[b] = c0 - run / GS;   // segment group = delta[b] + stage group
                cur[b] = c0 + padded / GS;
                run += padded;
            }
        }
        PSK_TICK(8);
        if constexpr (PSK_EXP_PRIO == 3) { if (threadIdx.x < 64) __builtin_amdgcn_s_setprio(0); }
        lds_barrier();
        if (B <= 64 * kPartScanPerThread) tile_probes = wave_tot[0];
        PSK_TICK(3);
        if constexpr (SORTED) {  // the NEXT tile's length sort: its descriptors have been on their way since the barrier before the scan, and the
            const uint64_t tn = tile + gridDim.x;  // first windows it requests land under the stage sort below
            sort_tile(nbase, tn < ntiles ? (nbase + tk < nlim ? nbase + tk : nlim) : nbase);
            PSK_TICK(10);
        }

        // ---- counting-sort the probes into the LDS stage; one thread per slice also fills its run's trailing pads
        // (in the scan phase that was 4 slices x up to GS-1 serial stores on the single scanning wave: 18 % of pass 1)
        constexpr bool LOOKUP = pay_is_lookup<Pay>::value;
#pragma unroll
        for (int q = 0; q < KPT; ++q) {
            const uint64_t i = base + slot[q];
            if (i < tile_end && kept(q)) {
                uint32_t pos[LOOKUP ? 2 * ((KT + 1) / 2) : 1] = {};  // lookups: where each of my probes sits in the sorted stage
                if constexpr (pay_has_tally<Pay>::value) {
                    // (here, not where the weight is loaded: the sum would pin the load's latency into the hash phase --
                    // measured +19 us per 10 M keys -- while this phase consumes the weight anyway)
                    if (pay.tally) {
                        const long long v = pay.weights_signed ? (long long)(int32_t)payload[q] : (long long)payload[q];
                        tally_s += v;
                        tally_a += (unsigned long long)(v < 0 ? -v : v);
                        tally_b += (uint32_t)(payload[q] >= (1u << kSmallWeightBits));
                    }
                }
                // the k offsets first, then the k stores: written as one loop hipcc waits for every off[] read before the stage[]
                // store in front of the next one (they may alias for all it knows) -- k dependent LDS round trips per key
                uint32_t pp[KT];
#pragma unroll
                for (int j = 0; j < KT; ++j) pp[j] = (uint32_t)j < k ? off[idx[q][j] >> g.shift] + rank[q][j] : 0u;
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    if ((uint32_t)j < k) {
                        const uint32_t p = pp[j];
                        if constexpr (LOOKUP) pos[j] = p;
                        if constexpr (PAIR) {
                            // final word (payload << shift | index in the slice); the slice of every group is kept
                            // aside by whoever fills the group's first slot
                            const uint32_t cell = idx[q][j];
                            uint32_t enc = (payload[q] << g.shift) | (cell & mask);
                            if (Pay::mode == kModeInline && payload[q] >= (1u << (31 - g.shift))) {
                                spill(cell, payload[q]);  // big / negative weight: exact saturating add on the table
                                enc = kPadProbe;
                            }
                            stage[p] = enc;
                            if (p % GS == 0) gb[p / GS] = cell >> g.shift;
                        } else if constexpr (pay_weighted_plain<Pay>::value) {
                            uint32_t wq = payload[q];
                            if (wq >= (1u << kSmallWeightBits)) {  // big / negative weight: exact saturating add on the table, a no-op field stays
                                spill(idx[q][j], wq);
                                wq = 0;
                            }
                            stage[p] = (wq << kSmallWeightShift) | idx[q][j];
                        } else {
                            stage[p] = idx[q][j];
                        }
                    }
                }
                if constexpr (LOOKUP) {  // perm[key]: 16-bit stage positions, ceil(k / 2) dwords per key (PermRec)
                    PermRec<KT> rec;
#pragma unroll
                    for (int c = 0; c < PermRec<KT>::PD; ++c) rec.w[c] = pos[2 * c] | (pos[2 * c + 1] << 16);
                    perm_store<KT>(pay.perm, i, rec);
                }
            }
        }
        // (after the sort stores: idx / rank are dead by now, so this costs no registers)
        for (uint32_t b = threadIdx.x; b < B; b += NT) {
            const uint32_t cnt = hist[b], padded = (cnt + GS - 1) / GS * GS, at = off[b];
            for (uint32_t e = cnt; e < padded; ++e) {
                stage[at + e] = kPadProbe;
            }
            // runinfo[tile][slice]: first group of this tile's run inside my segment of the slice, stage offset << 16 | count
            if constexpr (LOOKUP) pay.runinfo[tile * B + b] = make_uint2(delta[b] + at / GS, (at << 16) | cnt);
        }
        lds_barrier();
        PSK_TICK(4);

        // ---- write out: one lane = one group of GS probes of ONE run; its first probe is always real (pads
        // trail), so it names the slice
#pragma unroll
        for (int q = 0; q < KPT; ++q)
            if (kPartPipeline) Src::pin(kcur[q]);  // next tile's keys have landed: nothing to wait for later
        if constexpr (SORTED) {  // the next tile's first windows as well; its slots take over (this tile's were last used by the stage sort)
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                asm volatile("" : "+v"(kfirst[q].x), "+v"(kfirst[q].y), "+v"(kfirst[q].z), "+v"(kfirst[q].w));
                slot[q] = slot_n[q];
            }
        }
        if (!(dbg & 1)) {
            const uint32_t ngroups = tile_probes / GS;
            for (uint32_t gi = threadIdx.x; gi < ngroups; gi += NT) {
                emit_group<Pay, Spill>(stage, gb, delta, g, mask, gi, (Pay::mode == kModeKeyed || pay_tile_tag<Pay>::value) ? (uint64_t)ordinal : tile, base, spill, wg_buckets);
            }
        }
        // (pipelined form) no barrier needed here: the next iteration touches only hist (last read two barriers
        // ago) before its own barriers, and writes stage only after the barrier that follows its hash phase
        if (!kPartPipeline) lds_barrier();
        PSK_TICK(5);
    }
    lds_barrier();
    if constexpr (PHASED) {  // the phases behind my last tile (all of them for a workgroup without tiles)
        for (; ph_cur < pay.nph; ++ph_cur) write_snapshot(ph_cur);
    }
    if ((dbg & 32) && threadIdx.x == 0) {  // counts land behind the segment counts (host reserves the room)
        unsigned long long *prof = reinterpret_cast<unsigned long long *>(segcnt + (size_t)g.nbuckets * g.nwg);
        for (int ph = 1; ph < 12; ++ph) atomicAdd(prof + ph, t_acc[ph]);
        atomicAdd(prof, 1ULL);
    }
#undef PSK_TICK
    if constexpr (pay_has_tally<Pay>::value) {
        if (pay.tally) {  // block reduction through the (now idle) stage, one 16-byte store per workgroup
            for (int o = 32; o > 0; o >>= 1) {
                tally_s += __shfl_down(tally_s, o);
                tally_a += __shfl_down(tally_a, o);
                tally_b += __shfl_down(tally_b, o);
            }
            unsigned long long *red = reinterpret_cast<unsigned long long *>(stage);
            if ((threadIdx.x & 63) == 0) {
                red[3 * (threadIdx.x >> 6)] = (unsigned long long)tally_s;
                red[3 * (threadIdx.x >> 6) + 1] = tally_a;
                red[3 * (threadIdx.x >> 6) + 2] = tally_b;
            }
            lds_barrier();
            if (threadIdx.x == 0) {
                unsigned long long ss = 0, aa = 0, bb = 0;
                for (int w = 0; w < NT / 64; ++w) { ss += red[3 * w]; aa += red[3 * w + 1]; bb += red[3 * w + 2]; }
                pay.tally[blockIdx.x] = make_ulonglong4(ss, aa, bb, 0ULL);
            }
        }
    }
    // publish how many groups of each of my segments are valid (the kernel boundary orders it before pass 2)
    for (uint32_t b = threadIdx.x; b < B; b += NT) {
        const uint32_t c = cur[b];
        segcnt[(uint64_t)b * g.nwg + blockIdx.x] = c < g.segcap ? c : g.segcap;
    }
}

// The per-workgroup (sum w, sum |w|) slots of a weighted pass 1 -> the handle's device counters: ctr[which] += sum w
// (elements_added terms), ctr[6] = sum |w| * bound_mult of THIS round (pass 2's wrap check), and, when grow_bound, the
// saturating bound on |counter| ctr[4].  One block; the stream orders it between pass 1 and pass 2.
// big_pin (pinned host page, may be null): [0] = how many weights of this batch lay outside 0 .. 15, [1] = the batch's number -- read
// by the host WITHOUT synchronisation when it picks the next weighted batch's probe format (a stale answer only costs speed: PayWeightSmall
// is exact for any weight)
static __global__ __launch_bounds__(256) void k_tally_fold(const ulonglong4 *slots, uint32_t nslots, long long *ctr, int which, long long bound_mult,
                                                           int grow_bound, volatile unsigned long long *big_pin, unsigned long long seq)
{
    __shared__ unsigned long long ps[4], pa[4], pb[4];
    unsigned long long ss = 0, aa = 0, bb = 0;
    for (uint32_t i = threadIdx.x; i < nslots; i += 256) {
        const ulonglong4 v = slots[i];
        ss += v.x;
        aa += v.y;
        bb += v.z;
    }
    for (int o = 32; o > 0; o >>= 1) {
        ss += __shfl_down(ss, o);
        aa += __shfl_down(aa, o);
        bb += __shfl_down(bb, o);
    }
    if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = ss; pa[threadIdx.x >> 6] = aa; pb[threadIdx.x >> 6] = bb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ss = ps[0] + ps[1] + ps[2] + ps[3];
        aa = pa[0] + pa[1] + pa[2] + pa[3];
        if (big_pin) {
            big_pin[0] = pb[0] + pb[1] + pb[2] + pb[3];
            big_pin[1] = seq;  // (which batch the count belongs to: the host numbers them)
        }
        if (which >= 0) ctr[which] += (long long)ss;
        const unsigned long long add = aa * (unsigned long long)bound_mult;
        ctr[6] = (long long)(add >> 63 ? (1ULL << 62) : add);
        if (grow_bound) {
            const unsigned long long nb = (unsigned long long)ctr[4] + add;
            ctr[4] = (nb < add || nb > (1ULL << 62)) ? (1LL << 62) : (long long)nb;
        }
    }
}

// ------------------------------------------------------------------------------------ pass 1b: second-level split
// Tables with more slices than pass 1 can bin at once (kPartMaxBuckets, and long before that the runs per tile get
// too short to fill 16-byte groups) go through two levels: k_part_scatter bins by COARSE bucket (2^sub_bits slices,
// inline encoding: payload << shift1 | index inside the coarse bucket), then this kernel re-bins every coarse
// bucket's probe stream by slice and writes the format pass 2 expects.  No hashing here: the kernel is LDS / copy work.
//   grid (P, B1): workgroup (p, c) takes the p-th share of coarse bucket c's level-1 segments and owns segment p of each
//   of c's slices.  g1 describes level 1 (nbuckets = B1, shift = shift1, nwg, segcap), g2 the output
//   (nbuckets = all slices, shift, nwg = P, segcap).
// OUT: 0 = Bloom insert (6 x 20-bit packed), 1 = unit counter adds (8 x 16-bit), 2 = weighted counter adds (4 x 32-bit)
constexpr int kSplitThreads = 1024;
constexpr int kSplitMaxSub = 64;   // slices per coarse bucket
constexpr int kSplitMaxSegs = 512; // level-1 segments one workgroup walks

template <int OUT, class Spill>
__global__ __launch_bounds__(kSplitThreads) void k_part_split(PartGeom g1, const uint32_t *segcnt1, const uint4 *buckets1, PartGeom g2,
                                                              uint32_t sub_bits, uint32_t *segcnt2, uint4 *buckets2, Spill spill)
{
    constexpr int GS = OUT == 0 ? 6 : (OUT == 1 ? 8 : 4);
    constexpr int NT = kSplitThreads;
    __shared__ uint32_t pre[kSplitMaxSegs + 1];   // exclusive prefix of my segments' group counts
    __shared__ uint32_t hist[kSplitMaxSub], off[kSplitMaxSub], cur[kSplitMaxSub], delta[kSplitMaxSub];
    __shared__ uint32_t total_probes;
    __shared__ __attribute__((aligned(16))) uint32_t stage[NT * 4 + kSplitMaxSub * (GS - 1)];
    const uint32_t c = blockIdx.y, p = blockIdx.x, P = gridDim.x;
    const uint32_t S = 1u << sub_bits;
    const uint32_t per = (g1.nwg + P - 1) / P;
    const uint32_t seg_lo = p * per, seg_hi = seg_lo + per < g1.nwg ? seg_lo + per : g1.nwg;
    const uint32_t nseg = seg_hi > seg_lo ? seg_hi - seg_lo : 0;
    const uint32_t mask1 = (1u << g1.shift) - 1, mask = (1u << g2.shift) - 1;

    // prefix of group counts over my segments (nseg <= kSplitMaxSegs; one pass of wave 0, 8 segments per lane)
    if (threadIdx.x < 64) {
        uint32_t mine[kSplitMaxSegs / 64], s = 0;
#pragma unroll
        for (int q = 0; q < kSplitMaxSegs / 64; ++q) {
            const uint32_t sl = threadIdx.x * (kSplitMaxSegs / 64) + q;
            mine[q] = sl < nseg ? segcnt1[(uint64_t)c * g1.nwg + seg_lo + sl] : 0;
            s += mine[q];
        }
        const uint32_t inc = wave_inclusive_scan(s);
        uint32_t run = inc - s;
#pragma unroll
        for (int q = 0; q < kSplitMaxSegs / 64; ++q) {
            const uint32_t sl = threadIdx.x * (kSplitMaxSegs / 64) + q;
            if (sl <= nseg) pre[sl] = run;
            run += mine[q];
        }
    }
    if (threadIdx.x < S) cur[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t ngroups = pre[nseg];

    for (uint32_t g0 = 0; g0 < ngroups; g0 += NT) {
        if (threadIdx.x < S) hist[threadIdx.x] = 0;
        __syncthreads();
        // ---- my group: find its segment (binary search in the prefix), load, rank each probe inside its slice
        const uint32_t f = g0 + threadIdx.x;
        uint4 q = make_uint4(kPadProbe, kPadProbe, kPadProbe, kPadProbe);
        if (f < ngroups) {
            uint32_t lo = 0, hi = nseg;  // pre[lo] <= f < pre[hi]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (pre[mid] <= f) lo = mid;
                else hi = mid;
            }
            q = buckets1[seg_index(g1, c, seg_lo + lo) * g1.segcap + (f - pre[lo])];
        }
        const uint32_t e[4] = {q.x, q.y, q.z, q.w};
        uint32_t rank[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (e[j] != kPadProbe) rank[j] = atomicAdd(&hist[(e[j] & mask1) >> g2.shift], 1u);
        __syncthreads();
        // ---- scan the (<= 64) slice counts: run starts padded to whole output groups; advance my segment cursors
        if (threadIdx.x < 64) {
            const uint32_t cnt = threadIdx.x < S ? hist[threadIdx.x] : 0;
            const uint32_t padded = (cnt + GS - 1) / GS * GS;
            const uint32_t inc = wave_inclusive_scan(padded);
            if (threadIdx.x < S) {
                const uint32_t at = inc - padded;
                off[threadIdx.x] = at;
                const uint32_t c0 = cur[threadIdx.x];
                delta[threadIdx.x] = c0 - at / GS;
                cur[threadIdx.x] = c0 + padded / GS;
                for (uint32_t x = cnt; x < padded; ++x) stage[at + x] = kPadProbe;
            }
            if (threadIdx.x == 63) total_probes = inc;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (e[j] != kPadProbe) stage[off[(e[j] & mask1) >> g2.shift] + rank[j]] = e[j];
        __syncthreads();
        // ---- write-out: one lane = one output group of one slice (its first probe is real: pads trail)
        const uint32_t out_groups = total_probes / GS;
        for (uint32_t gi = threadIdx.x; gi < out_groups; gi += NT) {
            uint32_t v[GS];
#pragma unroll
            for (int x = 0; x < GS; ++x) v[x] = stage[gi * GS + x];
            const uint32_t sub = (v[0] & mask1) >> g2.shift;
            const uint32_t slice = (c << sub_bits) + sub;
            const uint32_t slot = delta[sub] + gi;
            if (slot < g2.segcap) {
                uint4 o;
                if constexpr (OUT == 0) {
                    uint32_t nv = 0;
#pragma unroll
                    for (int x = 0; x < 6; ++x) nv += v[x] != kPadProbe;
                    const uint32_t n0 = nv < 3 ? nv : 3, n1 = nv - n0;
                    auto l = [&](int x) -> unsigned long long { return (uint32_t)x < nv ? (unsigned long long)(v[x] & mask) : 0ULL; };
                    const unsigned long long h0 = l(0) | (l(1) << 20) | (l(2) << 40) | ((unsigned long long)n0 << 60);
                    const unsigned long long h1 = l(3) | (l(4) << 20) | (l(5) << 40) | ((unsigned long long)n1 << 60);
                    o = make_uint4((uint32_t)h0, (uint32_t)(h0 >> 32), (uint32_t)h1, (uint32_t)(h1 >> 32));
                } else if constexpr (OUT == 1) {
                    auto h16 = [&](int x) -> uint32_t { return v[x] == kPadProbe ? 0xFFFFu : (v[x] & mask); };
                    o = make_uint4(h16(0) | (h16(1) << 16), h16(2) | (h16(3) << 16), h16(4) | (h16(5) << 16), h16(6) | (h16(7) << 16));
                } else {
                    auto w32 = [&](int x) -> uint32_t { return v[x] == kPadProbe ? kPadProbe : (((v[x] >> g1.shift) << g2.shift) | (v[x] & mask)); };
                    o = make_uint4(w32(0), w32(1), w32(2), w32(3));
                }
                buckets2[seg_index(g2, slice, p) * g2.segcap + slot] = o;
            } else {  // level-2 segment full: exact fallback on the table
#pragma unroll
                for (int x = 0; x < GS; ++x)
                    if (v[x] != kPadProbe) spill((slice << g2.shift) | (v[x] & mask), v[x] >> g1.shift);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < S) {
        const uint32_t slice = (c << sub_bits) + threadIdx.x;
        if (slice < g2.nbuckets) segcnt2[(uint64_t)slice * g2.nwg + p] = cur[threadIdx.x] < g2.segcap ? cur[threadIdx.x] : g2.segcap;
    }
}

// ------------------------------------------------------------------------------------ pass 2
constexpr int kApplyThreads = 1024;
constexpr int kApplyWaves = kApplyThreads / 64;

// Walk the groups of slice `b`: wave w takes segments w, w+16, ... (<= 64 of them: one count per lane).
// A segment is a few hundred probes, so walking segment by segment is a chain of dependent HBM latencies
// (count -> data -> next count ...).  Two ways of keeping D 16-byte loads per lane in flight whatever segments they fall in:
//   chunked (g.dense == 0, long segments)  every segment is cut into 64-group chunks, the chunks are numbered across the
//       wave's segments (DPP prefix sum of the per-segment chunk counts); chunk -> (segment, offset) is scalar work: ballot +
//       popcount + two readlanes, the lanes of a load share one segment base.  The last chunk of a segment fills only
//       (groups mod 64) lanes: nothing for the 178-group segments of the headline configuration, but the 9-group segments
//       of a 10 M-key lookup into 2048 slices used 14 % of the lanes of every load.
//   dense (g.dense == 1, short segments)  the wave's segments are laid end to end: a prefix sum of the per-segment GROUP
//       counts numbers the wave's groups 0 .. T-1 and lane l of load d takes group G0 + 64 d + l.  The first segment of a row
//       of 64 groups is scalar work as before; the segment boundaries INSIDE the row are walked with one readlane + compare +
//       two selects each, and every lane computes its own segment base.  m = 2^31: lookups of 10 M keys 665 -> 559 us, of
//       2^25 keys 1415 -> 1382 us; on segments of ~40 groups and more it costs 1-2 % (178-group segments of the headline
//       configuration: check 41.5 -> 40.7 G keys/s), hence the switch (launch_scatter_nt: mean groups per segment below
//       the option `dense_walk_groups`, default 40).
// body(q, at, wg): the D groups, their flat index in the bucket buffer (~0 = absent) and the pass-1 workgroup of their
// segment (per lane in the dense walk); absent ones hold `pad`.
// the group count of the segment lane `lane` of my wave walks (for_each_batch_at); a kernel may ask for it BEFORE it loads its
// slice, so that the count -> addresses -> groups chain starts under the slice load instead of behind it
__device__ __forceinline__ uint32_t lane_segment_count(const uint32_t *segcnt, const PartGeom &g, uint32_t b)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t S = g.split > 1 ? g.split : 1, s0 = g.split > 1 ? g.split_idx : 0;
    const uint32_t mine_total = g.nwg > s0 ? (g.nwg - s0 + S - 1) / S : 0;                     // segments this workgroup walks
    const uint32_t nseg = mine_total > wave ? (mine_total - wave + kApplyWaves - 1) / kApplyWaves : 0;  // <= 64 per wave
    return lane < nseg ? segcnt[(uint64_t)b * g.nwg + s0 + S * (wave + kApplyWaves * lane)] : 0u;
}

template <int D, class Body>
__device__ __forceinline__ void for_each_batch_at(const uint4 *buckets, const uint32_t *segcnt, const PartGeom &g, uint32_t b,
                                                  const uint4 pad, Body body, uint32_t mycnt_pre = ~0u)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t S = g.split > 1 ? g.split : 1, s0 = g.split > 1 ? g.split_idx : 0;
    const uint32_t mycnt = mycnt_pre != ~0u ? mycnt_pre : lane_segment_count(segcnt, g, b);
    if (g.dense) {
        const uint32_t incl = wave_inclusive_scan(mycnt), excl = incl - mycnt;
        const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        for (uint32_t G0 = 0; G0 < T; G0 += 64u * D) {
            uint4 q[D];
            uint64_t at[D];
            uint32_t wg[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const uint32_t row = G0 + 64u * (uint32_t)d;  // uniform: first group of this load
                const uint32_t G = row + lane;
                uint32_t t = (uint32_t)__builtin_popcountll(__ballot(incl <= row));  // uniform: segments that end at or before `row`
                const uint32_t sl = t < 64 ? t : 63;
                uint32_t seg = sl;                                                      // per lane: my group's segment ...
                uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)excl, sl);    // ... and the number of its first group
                for (; t < 64; ++t) {  // the boundaries inside this row (uniform trip count)
                    const uint32_t end_t = (uint32_t)__builtin_amdgcn_readlane((int)incl, t);
                    if (end_t > row + 63u) break;
                    const bool past = G >= end_t;
                    seg = past ? t + 1 : seg;
                    first = past ? end_t : first;
                }
                seg = seg < 64 ? seg : 63;
                wg[d] = s0 + S * (wave + kApplyWaves * seg);
                q[d] = pad;
                at[d] = ~0ULL;
                if (G < T) {
                    const uint64_t idx = seg_index(g, b, wg[d]) * g.segcap + (G - first);
                    q[d] = buckets[idx];
                    at[d] = idx;
                }
            }
            body(q, at, wg);
        }
        return;
    }
    const uint32_t chunks = (mycnt + 63) >> 6;
    const uint32_t incl = wave_inclusive_scan(chunks), excl = incl - chunks;
    const uint32_t C = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    for (uint32_t c0 = 0; c0 < C; c0 += D) {
        uint4 q[D];
        uint64_t at[D];
        uint32_t wg[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint32_t cc = c0 + d;
            const uint32_t seg = (uint32_t)__builtin_popcountll(__ballot(incl <= cc));  // uniform; == 64 past the end
            const uint32_t sl = seg < 64 ? seg : 63;
            const uint32_t v = (cc - (uint32_t)__builtin_amdgcn_readlane((int)excl, sl)) * 64 + lane;
            const uint32_t cnt = seg < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)mycnt, sl) : 0;
            wg[d] = s0 + S * (wave + kApplyWaves * sl);
            const uint64_t base = seg_index(g, b, wg[d]) * g.segcap;
            q[d] = pad;
            at[d] = ~0ULL;
            if (v < cnt) {
                q[d] = buckets[base + v];
                at[d] = base + v;
            }
        }
        body(q, at, wg);
    }
}

template <int D, class Body>
__device__ __forceinline__ void for_each_batch(const uint4 *buckets, const uint32_t *segcnt, const PartGeom &g, uint32_t b,
                                               const uint4 pad, Body body)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nseg = g.nwg > wave ? (g.nwg - wave + kApplyWaves - 1) / kApplyWaves : 0;  // <= 64
    uint32_t mycnt = 0;
    if (lane < nseg) mycnt = segcnt[(uint64_t)b * g.nwg + wave + kApplyWaves * lane];
    if (g.dense) {  // see for_each_batch_at
        const uint32_t incl = wave_inclusive_scan(mycnt), excl = incl - mycnt;
        const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        for (uint32_t G0 = 0; G0 < T; G0 += 64u * D) {
            uint4 q[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const uint32_t row = G0 + 64u * (uint32_t)d;
                const uint32_t G = row + lane;
                uint32_t t = (uint32_t)__builtin_popcountll(__ballot(incl <= row));
                const uint32_t sl = t < 64 ? t : 63;
                uint32_t seg = sl;
                uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)excl, sl);
                for (; t < 64; ++t) {
                    const uint32_t end_t = (uint32_t)__builtin_amdgcn_readlane((int)incl, t);
                    if (end_t > row + 63u) break;
                    const bool past = G >= end_t;
                    seg = past ? t + 1 : seg;
                    first = past ? end_t : first;
                }
                seg = seg < 64 ? seg : 63;
                q[d] = pad;
                if (G < T) q[d] = buckets[seg_index(g, b, wave + kApplyWaves * seg) * g.segcap + (G - first)];
            }
            body(q);
        }
        return;
    }
    const uint32_t chunks = (mycnt + 63) >> 6;
    const uint32_t incl = wave_inclusive_scan(chunks), excl = incl - chunks;
    const uint32_t C = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    for (uint32_t c0 = 0; c0 < C; c0 += D) {
        uint4 q[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint32_t cc = c0 + d;
            const uint32_t seg = (uint32_t)__builtin_popcountll(__ballot(incl <= cc));  // uniform; == 64 past the end
            const uint32_t sl = seg < 64 ? seg : 63;
            const uint32_t v = (cc - (uint32_t)__builtin_amdgcn_readlane((int)excl, sl)) * 64 + lane;
            const uint32_t cnt = seg < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)mycnt, sl) : 0;
            const uint4 *src = buckets + seg_index(g, b, wave + kApplyWaves * sl) * g.segcap;
            q[d] = pad;
            if (v < cnt) q[d] = src[v];
        }
        body(q);
    }
}

constexpr int kApplyDepth = 12;  // 16-byte loads in flight per lane

// The 16-byte piece `w` (in words) of a table slice that starts at word w0; words at or beyond tab_words read as 0.
// Tables far beyond the 256 MB Infinity Cache (BASELINE cfg 5: 256 MiB of bits per replica) are swept once per call: nontemporal accesses
// keep the sweep from pushing the probe stream out of the cache (scripts/ubench/tabpass.hip: + 10 % on such a pass; round 4)
constexpr uint64_t kNtTableWords = 1ULL << 25;  // 128 MiB
constexpr uint32_t kGeomNtBit = 0x80000000u;    // PartGeom::dbg bit 31 (a production bit, not a bench knob): option "big_table_nt" is on
__device__ __forceinline__ uint4 slice_piece(const uint32_t *tab, uint64_t tab_words, uint64_t w0, uint32_t w, bool nt = false)
{
    const uint64_t gw = w0 + w;
    if (gw + 3 < tab_words) {
        if (nt) {
            typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4)));
            const nt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4 *>(tab + gw));
            return make_uint4(v.x, v.y, v.z, v.w);
        }
        return *reinterpret_cast<const uint4 *>(tab + gw);
    }
    uint4 t = make_uint4(0, 0, 0, 0);
    if (gw + 0 < tab_words) t.x = tab[gw + 0];
    if (gw + 1 < tab_words) t.y = tab[gw + 1];
    if (gw + 2 < tab_words) t.z = tab[gw + 2];
    return t;
}

// A table slice of `slice_words` words into LDS, kSliceLoads 16-byte loads in flight per lane (written as a plain loop the 8
// pieces a lane moves for a 128 KiB slice were 8 DEPENDENT round trips -- load, wait, LDS store -- in front of every pass-2
// workgroup's work: round 3)
constexpr int kSliceLoads = 8;
__device__ __forceinline__ void load_slice(uint32_t *smem, const uint32_t *tab, uint64_t tab_words, uint64_t w0, uint32_t slice_words, bool nt_on = false)
{
    const bool nt = nt_on && tab_words >= kNtTableWords;
    for (uint32_t wb = threadIdx.x * 4; wb < slice_words; wb += kApplyThreads * 4 * kSliceLoads) {
        uint4 t[kSliceLoads];
#pragma unroll
        for (int u = 0; u < kSliceLoads; ++u) {
            const uint32_t w = wb + (uint32_t)u * kApplyThreads * 4;
            t[u] = w < slice_words ? slice_piece(tab, tab_words, w0, w, nt) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kSliceLoads; ++u) {
            const uint32_t w = wb + (uint32_t)u * kApplyThreads * 4;
            if (w < slice_words) *reinterpret_cast<uint4 *>(smem + w) = t[u];
        }
    }
}

// consumers that only issue LDS atomics: one callback per group
template <int DEPTH = kApplyDepth, class F4>
__device__ __forceinline__ void for_each_group(const uint4 *buckets, const uint32_t *segcnt, const PartGeom &g, uint32_t b,
                                               const uint4 pad, F4 f4)
{
    for_each_batch<DEPTH>(buckets, segcnt, g, b, pad, [&](const uint4 (&q)[DEPTH]) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) f4(q[d]);
    });
}

// consumers that READ LDS per probe: `pre` (the reads) runs on the whole batch before `post` (the tests), so the
// reads carry no control dependence and issue back to back instead of one read -> wait -> branch trip per probe
template <class Pre, class Post>
__device__ __forceinline__ void for_each_group_padded(const uint4 *buckets, const uint32_t *segcnt, const PartGeom &g, uint32_t b,
                                                      const uint4 pad, Pre pre, Post post)
{
    for_each_batch<kApplyDepth>(buckets, segcnt, g, b, pad, [&](const uint4 (&q)[kApplyDepth]) {
        uint4 w[kApplyDepth];
#pragma unroll
        for (int d = 0; d < kApplyDepth; ++d) w[d] = pre(q[d]);
#pragma unroll
        for (int d = 0; d < kApplyDepth; ++d) post(q[d], w[d]);
    });
}

// Bloom insert: OR the slice's probes into an LDS image of the slice, then OR the image into the table.
// dynamic LDS: slice image, 2^shift bits
static __global__ __launch_bounds__(kApplyThreads) void k_bloom_apply(uint32_t *tab, uint64_t tab_words, PartGeom g,
                                                               const uint32_t *segcnt, const uint4 *buckets)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t b = blockIdx.x;
    const uint32_t slice_words = 1u << (g.shift - 5);
    for (uint32_t w = threadIdx.x; w < slice_words; w += kApplyThreads) smem[w] = 0;
    __syncthreads();
    // group = two 64-bit halves of 3 x 20-bit slice-local bit indices, valid count in bits 60..63
    auto half = [&](uint32_t lo, uint32_t hi) {
        const unsigned long long h = ((unsigned long long)hi << 32) | lo;
        const uint32_t nv = hi >> 28;
        if (nv > 0) { const uint32_t x = (uint32_t)h & 0xFFFFFu; atomicOr(&smem[x >> 5], 1u << (x & 31)); }  // ds_or_b32
        if (nv > 1) { const uint32_t x = (uint32_t)(h >> 20) & 0xFFFFFu; atomicOr(&smem[x >> 5], 1u << (x & 31)); }
        if (nv > 2) { const uint32_t x = (uint32_t)(h >> 40) & 0xFFFFFu; atomicOr(&smem[x >> 5], 1u << (x & 31)); }
    };
    for_each_group(buckets, segcnt, g, b, make_uint4(0, 0, 0, 0), [&](const uint4 q) { half(q.x, q.y); half(q.z, q.w); });
    __syncthreads();
    // merge: this workgroup is the only writer of its slice
    const uint64_t w0 = (uint64_t)b * slice_words;
    const bool nt = (g.dbg & kGeomNtBit) != 0 && tab_words >= kNtTableWords;  // (see slice_piece)
    typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4)));
    constexpr int kFold = 4;  // 16-byte pieces in flight per lane (one at a time was a chain of HBM round trips)
    for (uint32_t wb = threadIdx.x * 4; wb < slice_words; wb += kApplyThreads * 4 * kFold) {
        uint4 t[kFold], add[kFold];
        bool full[kFold];
#pragma unroll
        for (int u = 0; u < kFold; ++u) {
            const uint32_t w = wb + (uint32_t)u * kApplyThreads * 4;
            const uint64_t gw = w0 + w;
            full[u] = w < slice_words && gw + 3 < tab_words;
            add[u] = w < slice_words ? *reinterpret_cast<const uint4 *>(smem + w) : make_uint4(0, 0, 0, 0);
            t[u] = make_uint4(0, 0, 0, 0);
            if (full[u] && (add[u].x | add[u].y | add[u].z | add[u].w)) {
                if (nt) {
                    const nt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4 *>(tab + gw));
                    t[u] = make_uint4(v.x, v.y, v.z, v.w);
                } else {
                    t[u] = *reinterpret_cast<const uint4 *>(tab + gw);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kFold; ++u) {
            const uint32_t w = wb + (uint32_t)u * kApplyThreads * 4;
            const uint64_t gw = w0 + w;
            if (!(add[u].x | add[u].y | add[u].z | add[u].w)) continue;
            if (full[u]) {
                t[u].x |= add[u].x; t[u].y |= add[u].y; t[u].z |= add[u].z; t[u].w |= add[u].w;
                if (nt) {
                    nt_u32x4 v;
                    v.x = t[u].x; v.y = t[u].y; v.z = t[u].z; v.w = t[u].w;
                    __builtin_nontemporal_store(v, reinterpret_cast<nt_u32x4 *>(tab + gw));
                } else {
                    *reinterpret_cast<uint4 *>(tab + gw) = t[u];
                }
            } else if (w < slice_words) {
                const uint32_t aa[4] = {add[u].x, add[u].y, add[u].z, add[u].w};
                for (uint32_t e = 0; e < 4; ++e)
                    if (gw + e < tab_words && aa[e]) tab[gw + e] |= aa[e];
            }
        }
    }
}

// Bloom lookup: the slice is loaded into LDS; a probe whose bit is clear zeroes its key's result byte
// (out[] is pre-set to 1; every writer stores the same 0, so plain byte stores suffice).
// group = 4 x (tile bit << 31 | key index in tile << shift | bit index in slice), see PayKeyId
// miss_ctr (nullable): += the number of probes that found their bit clear (feeds the host's choice of lookup scheme)
static __global__ __launch_bounds__(kApplyThreads) void k_bloom_test(const uint32_t *tab, uint64_t tab_words, PartGeom g,
                                                              const uint32_t *segcnt, const uint4 *buckets, uint8_t *out,
                                                              unsigned long long *miss_ctr)
{
    uint32_t nmiss = 0;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t b = blockIdx.x;
    const uint32_t slice_words = 1u << (g.shift - 5);
    const uint32_t mask = (1u << g.shift) - 1;
    const uint64_t w0 = (uint64_t)b * slice_words;
    const uint32_t dbg = kBenchKnobs ? g.dbg : 0u;
    const uint32_t mycnt = lane_segment_count(segcnt, g, b);  // (requested before the slice: see lane_segment_count)
    if (!(dbg & 128)) load_slice(smem, tab, tab_words, w0, slice_words, (g.dbg & kGeomNtBit) != 0);
    __syncthreads();
    const uint32_t kmask = (1u << (31 - g.shift)) - 1;
    for_each_batch_at<kApplyDepth>(buckets, segcnt, g, b, make_uint4(0, 0, 0, 0), [&](const uint4 (&q)[kApplyDepth], const uint64_t (&at)[kApplyDepth],
                                                                                     const uint32_t (&wg)[kApplyDepth]) {
        // the LDS reads of the whole batch first (no control dependence: they issue back to back), then the tests
        uint4 w[kApplyDepth];
#pragma unroll
        for (int d = 0; d < kApplyDepth; ++d) {
            if (dbg & 64) { w[d] = make_uint4(~0u, ~0u, ~0u, ~0u); continue; }
            w[d] = make_uint4(smem[(q[d].x & mask) >> 5], smem[(q[d].y & mask) >> 5], smem[(q[d].z & mask) >> 5], smem[(q[d].w & mask) >> 5]);
        }
#pragma unroll
        for (int d = 0; d < kApplyDepth; ++d) {  // the rare miss stores
            const bool live = at[d] != ~0ULL;
            const bool mx = live && ((w[d].x >> (q[d].x & 31)) & 1u) == 0;
            const bool my = live && ((w[d].y >> (q[d].y & 31)) & 1u) == 0;
            const bool mz = live && ((w[d].z >> (q[d].z & 31)) & 1u) == 0;
            const bool mw = live && ((w[d].w >> (q[d].w & 31)) & 1u) == 0;
            nmiss += (uint32_t)mx + (uint32_t)my + (uint32_t)mz + (uint32_t)mw;
            if (mx | my | mz | mw) {
                const uint32_t ordinal = (q[d].x >> 31) | ((q[d].y >> 31) << 1) | ((q[d].z >> 31) << 2) | ((q[d].w >> 31) << 3);
                const uint32_t kbase = (ordinal * g.nwg + wg[d]) * g.tile;
                if (mx) out[kbase + ((q[d].x >> g.shift) & kmask)] = 0;
                if (my) out[kbase + ((q[d].y >> g.shift) & kmask)] = 0;
                if (mz) out[kbase + ((q[d].z >> g.shift) & kmask)] = 0;
                if (mw) out[kbase + ((q[d].w >> g.shift) & kmask)] = 0;
            }
        }
    }, mycnt);
    if (miss_ctr) {  // one atomic per workgroup (same-address device atomics serialise at ~11 ns each)
        for (int o = 32; o > 0; o >>= 1) nmiss += __shfl_down(nmiss, o);
        __syncthreads();  // every wave is done with the slice image: reuse its first words
        if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = nmiss;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < kApplyWaves; ++w) t += smem[w];
            if (t) atomicAdd(miss_ctr, t);
        }
    }
}

// Bloom lookup, tile-flag scheme (round 5; PayTileTag): the probes are the insert's 6 x 20-bit groups, so pass 1 and the probe stream cost what
// the insert's do (18.7 instead of 28 bytes per key).  A probe that finds its bit SET needs no way back to its key; one that finds it clear
// flags its tile: tileflag[tile] = gen, the round's generation number (plain stores of the same value; a flag of an older round is not a
// flag, so nothing ever resets them; the tile = ordinal in the groups' spare bits x workgroups + the segment's workgroup).
// k_bloom_flag_resolve then re-checks the flagged tiles.  miss_ctr as in k_bloom_test (probes that found their bit clear).
constexpr int kFlagDepth = 8;  // 16-byte groups in flight per lane: 8 x (4 + 6 LDS words) registers
// out[0 .. n): the round's answers; every workgroup first presets its share to 1 ("present") -- fire-and-forget stores under its slice
// load, instead of a fill launch -- and k_bloom_flag_resolve, the next kernel, overwrites the answers of the flagged tiles' keys.
static __global__ __launch_bounds__(kApplyThreads) void k_bloom_test_flag(const uint32_t *tab, uint64_t tab_words, PartGeom g,
                                                                   const uint32_t *segcnt, const uint4 *buckets, uint32_t *tileflag, uint32_t gen,
                                                                   unsigned long long *miss_ctr, uint8_t *out, uint64_t n)
{
    uint32_t nmiss = 0;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t b = blockIdx.x;
    {   // my share of out[]: 16-byte pieces between the 16-byte boundaries of the buffer, single bytes at its two ends
        const uintptr_t a0 = (uintptr_t)out, a1 = a0 + n;
        const uintptr_t m0 = (a0 + 15) & ~(uintptr_t)15, m1 = a1 & ~(uintptr_t)15;
        if (m0 < m1) {
            const uint64_t pieces = (m1 - m0) >> 4, per = (pieces + gridDim.x - 1) / gridDim.x;
            const uint64_t lo = (uint64_t)b * per, hi = lo + per < pieces ? lo + per : pieces;
            const uint4 ones = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
            uint4 *mid = reinterpret_cast<uint4 *>(out + (m0 - a0));  // (pointer arithmetic: through the integer these were FLAT stores)
            for (uint64_t i = lo + threadIdx.x; i < hi; i += kApplyThreads) mid[i] = ones;
            if (b == 0) {
                if (threadIdx.x < (uint32_t)(m0 - a0)) out[threadIdx.x] = 1;
                if (threadIdx.x < (uint32_t)(a1 - m1)) out[(m1 - a0) + threadIdx.x] = 1;
            }
        } else if (b == 0) {
            for (uint64_t i = threadIdx.x; i < n; i += kApplyThreads) out[i] = 1;
        }
    }
    const uint32_t slice_words = 1u << (g.shift - 5);
    const uint64_t w0 = (uint64_t)b * slice_words;
    const uint32_t mycnt = lane_segment_count(segcnt, g, b);  // (requested before the slice: see lane_segment_count)
    load_slice(smem, tab, tab_words, w0, slice_words, (g.dbg & kGeomNtBit) != 0);
    __syncthreads();
    for_each_batch_at<kFlagDepth>(buckets, segcnt, g, b, make_uint4(0, 0, 0, 0), [&](const uint4 (&q)[kFlagDepth], const uint64_t (&)[kFlagDepth],
                                                                                   const uint32_t (&wg)[kFlagDepth]) {
        // a pad field (all ones inside the slice) and the fields of an absent group (all zero, counts 0) address the slice: every read is safe
        uint32_t f[kFlagDepth][6], w[kFlagDepth][6];
#pragma unroll
        for (int d = 0; d < kFlagDepth; ++d) {
            f[d][0] = q[d].x & 0xFFFFFu;
            f[d][1] = __builtin_amdgcn_alignbit(q[d].y, q[d].x, 20) & 0xFFFFFu;
            f[d][2] = (q[d].y >> 8) & 0xFFFFFu;
            f[d][3] = q[d].z & 0xFFFFFu;
            f[d][4] = __builtin_amdgcn_alignbit(q[d].w, q[d].z, 20) & 0xFFFFFu;
            f[d][5] = (q[d].w >> 8) & 0xFFFFFu;
#pragma unroll
            for (int e = 0; e < 6; ++e) w[d][e] = smem[f[d][e] >> 5];
        }
#pragma unroll
        for (int d = 0; d < kFlagDepth; ++d) {
            uint32_t clear = 0;  // bit e: probe e's table bit is clear
#pragma unroll
            for (int e = 0; e < 6; ++e) clear |= (((w[d][e] >> (f[d][e] & 31)) & 1u) ^ 1u) << e;
            const uint32_t n0 = (q[d].y >> 28) & 3u, n1 = (q[d].w >> 28) & 3u;
            const uint32_t valid = ((1u << n0) - 1u) | (((1u << n1) - 1u) << 3);
            clear &= valid;
            if (clear) {  // rare on the batches this scheme is chosen for
                nmiss += (uint32_t)__builtin_popcount(clear);
                const uint32_t ordinal = (q[d].y >> 30) | ((q[d].w >> 30) << 2);
                tileflag[ordinal * g.nwg + wg[d]] = gen;
            }
        }
    }, mycnt);
    if (miss_ctr) {  // one atomic per workgroup
        for (int o = 32; o > 0; o >>= 1) nmiss += __shfl_down(nmiss, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = nmiss;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < kApplyWaves; ++w) t += smem[w];
            if (t) atomicAdd(miss_ctr, t);
        }
    }
}

// Last step of the tile-flag lookup: pass 2 has answered every key "present"; the keys of a FLAGGED tile are checked one by one against
// the table here (bloom.py:261-272: the direct kernel's loop) -- exact whatever the batch holds, next to nothing when no tile is flagged
// (every workgroup reads its few flags and leaves).  Workgroup w owns the tiles w, w + gridDim.x, ...
// publish (last round of a call under the automatic scheme choice): workgroup 0 copies the call's miss tally to the pinned page the next
// call's choice reads (what the one-thread k_lookup_publish launch does for the other schemes).
constexpr int kResolveThreads = 1024;  // (a flagged tile is re-checked by ONE workgroup: two keys per thread, latency-bound gathers)
struct LookupPublish {
    unsigned long long *tally = nullptr;         // device: [0] misses of the call
    volatile unsigned long long *pin = nullptr;  // pinned host page: see psk_sketch::lk
    unsigned long long units = 0, scheme = 0;
};
template <class Src, bool POW2>
__global__ __launch_bounds__(kResolveThreads) void k_bloom_flag_resolve(Src src, const uint32_t *tab, Mod md, uint32_t k, const uint32_t *tileflag, uint32_t gen,
                                                                       uint32_t tile, uint64_t n, uint8_t *out, LookupPublish pub)
{
    if (pub.tally && blockIdx.x == 0 && threadIdx.x == 0) {
        pub.pin[1] = pub.units;
        pub.pin[2] = pub.scheme;
        pub.pin[0] = pub.tally[0];
        pub.pin[3] = pub.pin[3] + 1;
        pub.tally[0] = 0;
    }
    const uint64_t ntiles = (n + tile - 1) / tile;
    BloomCheck<POW2> op{tab, md, k, out};
    // 64 of my tiles at a time: lane l of every wave reads the flag of tile (t0 + l) * gridDim.x + blockIdx.x
    for (uint64_t t0 = 0; t0 * gridDim.x + blockIdx.x < ntiles; t0 += 64) {
        const uint64_t mine = (t0 + (threadIdx.x & 63)) * gridDim.x + blockIdx.x;
        unsigned long long todo = __ballot(mine < ntiles && tileflag[mine < ntiles ? mine : 0] == gen);
        while (todo) {  // (uniform)
            const uint32_t l = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            const uint64_t t = (t0 + l) * gridDim.x + blockIdx.x, base = t * tile;
            const uint32_t cnt = (uint32_t)(n - base < tile ? n - base : tile);
            for (uint32_t j = threadIdx.x; j < cnt; j += kResolveThreads) {
                const uint64_t i = base + j;
                const typename Src::Key key = src.load(i);
                typename BloomCheck<POW2>::State st = op.begin(i);
                for_each_hash(src, key, i, k, [&](uint32_t jj, uint64_t h) { op.apply(st, jj, h); });
                op.end(st, i);
            }
        }
    }
}

// Counter add (CMS / CBF fast path): accumulate the slice's weights into an LDS image with ds_add, then
// fold the image into the table with the reference's saturating add.
// SIGNED: int32 bins clamped at both rails (countminsketch.py:280-284 / :312-316);
// else uint32 counters clamped at 2^32-1 (countingbloom.py:149-153).
// Exact for any order as long as the per-cell partial sums do not wrap 32 bits: unit weights -- the host
// checks n*k < 2^31; weighted -- ctr[6] = sum|w| of the batch, else every probe takes the saturating CAS.
// The accounting of a weighted pass 1 (PayWeight::tally), folded into pass 2 (round 3; it used to be the one-block k_tally_fold between the
// passes, ~5 us of stream time per round): every workgroup sums the slots for its own wrap check, workgroup 0 also books the sums.
struct TallyArgs {
    const ulonglong4 *slots = nullptr;  // null: nothing to fold (ctr[6] holds the round's sum |w| already)
    uint32_t nslots = 0;
    int which = -1;                     // PSK_CTR_ADDED / PSK_CTR_REMOVED (or -1)
    long long bound_mult = 1;
    int grow_bound = 0;
    volatile unsigned long long *big_pin = nullptr;  // see k_tally_fold
    unsigned long long seq = 0;
    // CBF decrements (NEG, !SIGNED) only -- the transactional remove: opt 1 = wrapping subtraction, `flag` raised when a counter would
    // go below zero or is frozen (countingbloom.py:198-206 then depends on the order inside the batch); opt 2 = the inverse (adds back)
    uint32_t opt = 0;
    uint32_t *flag = nullptr;
};

// FMT: the probe format -- 0 unit adds (8 x 16-bit cells), 1 weighted (4 x 32-bit: weight << shift | cell), 2 small weights (PayWeightSmall: 6 x
// 20-bit fields weight << 15 | cell in two counted halves)
template <bool SIGNED, int FMT, bool NEG>
__global__ __launch_bounds__(kApplyThreads) void k_counter_apply(uint32_t *tab, uint64_t tab_cells, PartGeom g,
                                                                 const uint32_t *segcnt, const uint4 *buckets,
                                                                 long long *ctr, unsigned long long *sat_ctr, TallyArgs ta)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t b = blockIdx.x;
    long long round_abs = 0;  // sum |w| * bound_mult of this round (the wrap check below)
    if (FMT != 0) {
        if (ta.slots) {
            __shared__ unsigned long long red[3 * (kApplyThreads / 64)];
            unsigned long long ss = 0, aa = 0, bb = 0;
            for (uint32_t i = threadIdx.x; i < ta.nslots; i += kApplyThreads) {
                const ulonglong4 v = ta.slots[i];
                ss += v.x;
                aa += v.y;
                bb += v.z;
            }
            for (int o = 32; o > 0; o >>= 1) {
                ss += __shfl_down(ss, o);
                aa += __shfl_down(aa, o);
                bb += __shfl_down(bb, o);
            }
            if ((threadIdx.x & 63) == 0) {
                red[3 * (threadIdx.x >> 6)] = ss;
                red[3 * (threadIdx.x >> 6) + 1] = aa;
                red[3 * (threadIdx.x >> 6) + 2] = bb;
            }
            __syncthreads();
            ss = aa = bb = 0;
            for (int w = 0; w < kApplyThreads / 64; ++w) { ss += red[3 * w]; aa += red[3 * w + 1]; bb += red[3 * w + 2]; }
            const unsigned long long add = aa * (unsigned long long)ta.bound_mult;
            round_abs = (long long)(add >> 63 ? (1ULL << 62) : add);
            if (b == 0 && threadIdx.x == 0) {  // the books (what k_tally_fold does)
                if (ta.which >= 0) ctr[ta.which] += (long long)ss;
                ctr[6] = round_abs;
                if (ta.grow_bound) {
                    const unsigned long long nb = (unsigned long long)ctr[4] + add;
                    ctr[4] = (nb < add || nb > (1ULL << 62)) ? (1LL << 62) : (long long)nb;
                }
                if (ta.big_pin) {
                    ta.big_pin[0] = bb;
                    ta.big_pin[1] = ta.seq;
                }
            }
        } else {
            round_abs = ctr[6];
        }
    }
    const uint32_t slice_cells = 1u << g.shift;
    const uint32_t mask = slice_cells - 1;
    const uint64_t c0 = (uint64_t)b * slice_cells;
    const uint4 pad4 = make_uint4(kPadProbe, kPadProbe, kPadProbe, kPadProbe);  // (unit adds: 0xFFFF halves)
    constexpr bool WEIGHTED = FMT != 0;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    // a half of a PayWeightSmall group: 3 fields (weight << 15 | cell), valid count in bits 60..63
    // (32-bit field extraction; slots past a run's end and the slots of weights that went to the table directly carry weight 0. zero4 -- what
    // for_each_group hands the lanes past a segment's end -- is three such fields per half)
    auto small_half = [&](uint32_t lo, uint32_t hi, auto &&one) {
        const uint32_t f[3] = {lo & 0xFFFFFu, __builtin_amdgcn_alignbit(hi, lo, 20) & 0xFFFFFu, (hi >> 8) & 0xFFFFFu};
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (f[e] >> 15) one(f[e] & 0x7FFFu, f[e] >> 15);
    };
    if (WEIGHTED && round_abs >= (1LL << 31)) {
        auto slow1 = [&](uint32_t cell_in_slice, uint32_t w) {
            const uint64_t cell = c0 + cell_in_slice;
            if (!SIGNED && NEG && ta.opt == 1) {
                const uint32_t old = atomicSub(tab + cell, w);
                if (old < w || old == 0xFFFFFFFFu) *ta.flag = 1u;
                return;
            }
            if (!SIGNED && NEG && ta.opt == 2) {
                atomicAdd(tab + cell, w);
                return;
            }
            if (SIGNED) cms_sat_add((int32_t *)tab + cell, NEG ? -(int64_t)w : (int64_t)w, sat_ctr);
            else if (NEG) cbf_sat_sub(tab + cell, w, sat_ctr - 1);
            else cbf_sat_add(tab + cell, w, sat_ctr);
        };
        if constexpr (FMT == 2) {
            for_each_group(buckets, segcnt, g, b, zero4, [&](const uint4 q) { small_half(q.x, q.y, slow1); small_half(q.z, q.w, slow1); });
        } else {
            auto slow = [&](uint32_t x) {
                if (x != kPadProbe) slow1(x & mask, x >> g.shift);
            };
            for_each_group(buckets, segcnt, g, b, pad4, [&](const uint4 q) { slow(q.x); slow(q.y); slow(q.z); slow(q.w); });
        }
        return;
    }
    for (uint32_t w = threadIdx.x; w < slice_cells; w += kApplyThreads) smem[w] = 0;
    __syncthreads();
    if constexpr (FMT == 2) {
        auto add1 = [&](uint32_t cell_in_slice, uint32_t w) { atomicAdd(&smem[cell_in_slice], NEG ? 0u - w : w); };  // ds_add_u32
        for_each_group(buckets, segcnt, g, b, zero4, [&](const uint4 q) { small_half(q.x, q.y, add1); small_half(q.z, q.w, add1); });
    } else if (WEIGHTED) {
        auto add = [&](uint32_t x) {
            if (x != kPadProbe) atomicAdd(&smem[x & mask], NEG ? 0u - (x >> g.shift) : (x >> g.shift));  // ds_add_u32
        };
        for_each_group(buckets, segcnt, g, b, pad4, [&](const uint4 q) { add(q.x); add(q.y); add(q.z); add(q.w); });
    } else {
        const uint32_t one = NEG ? 0xFFFFFFFFu : 1u;
        auto add2 = [&](uint32_t w) {  // two 16-bit slice-local cell indices, 0xFFFF = pad
            if ((w & 0xFFFFu) != 0xFFFFu) atomicAdd(&smem[w & 0xFFFFu], one);
            if ((w >> 16) != 0xFFFFu) atomicAdd(&smem[w >> 16], one);
        };
        for_each_group(buckets, segcnt, g, b, pad4, [&](const uint4 q) { add2(q.x); add2(q.y); add2(q.z); add2(q.w); });
    }
    __syncthreads();
    unsigned long long sat = 0, viol = 0;
    uint32_t bad = 0;
    auto fold = [&](uint32_t t, uint32_t d) -> uint32_t {  // the reference's saturating add
        if (!SIGNED && NEG) {
            // CBF decrement (countingbloom.py:203-206 with to_remove == num_els, i.e. a well-formed stream): a counter frozen
            // at 2^32-1 stays; one that would go below zero means the stream was not well-formed -- tallied, clamped at 0
            const uint32_t amount = 0u - d;
            if (ta.opt == 1) {
                if (t < amount || t == 0xFFFFFFFFu) bad = 1u;
                return t - amount;
            }
            if (ta.opt == 2) return t + amount;
            if (t == 0xFFFFFFFFu) return t;
            if (t < amount) { ++viol; return 0u; }
            return t - amount;
        }
        if (SIGNED) {
            int64_t v = (int64_t)(int32_t)t + (int64_t)(int32_t)d;
            if (v > INT32_MAX) { v = INT32_MAX; ++sat; }
            if (v < INT32_MIN) { v = INT32_MIN; ++sat; }
            return (uint32_t)(int32_t)v;
        }
        uint64_t v = (uint64_t)t + (uint64_t)d;
        if (v > 0xFFFFFFFFULL) { v = 0xFFFFFFFFULL; ++sat; }
        return (uint32_t)v;
    };
    // 16 bytes per lane, kFold of them in flight (a cell-by-cell loop was 32 dependent HBM round trips per workgroup:
    // 931 us to fold a 1 GiB table); untouched 16-byte pieces are not written back
    constexpr int kFold = 4;
    for (uint32_t w0 = threadIdx.x * 4; w0 < slice_cells; w0 += kApplyThreads * 4 * kFold) {
        uint4 t[kFold], d[kFold];
        bool full[kFold];
#pragma unroll
        for (int u = 0; u < kFold; ++u) {
            const uint32_t w = w0 + (uint32_t)u * kApplyThreads * 4;
            const uint64_t gc = c0 + w;
            full[u] = w < slice_cells && gc + 3 < tab_cells;
            d[u] = w < slice_cells ? *reinterpret_cast<const uint4 *>(smem + w) : make_uint4(0, 0, 0, 0);
            t[u] = make_uint4(0, 0, 0, 0);
            if (full[u] && (d[u].x | d[u].y | d[u].z | d[u].w)) t[u] = *reinterpret_cast<const uint4 *>(tab + gc);
        }
#pragma unroll
        for (int u = 0; u < kFold; ++u) {
            const uint32_t w = w0 + (uint32_t)u * kApplyThreads * 4;
            const uint64_t gc = c0 + w;
            if (!(d[u].x | d[u].y | d[u].z | d[u].w)) continue;
            if (full[u]) {
                uint4 o;
                o.x = d[u].x ? fold(t[u].x, d[u].x) : t[u].x;
                o.y = d[u].y ? fold(t[u].y, d[u].y) : t[u].y;
                o.z = d[u].z ? fold(t[u].z, d[u].z) : t[u].z;
                o.w = d[u].w ? fold(t[u].w, d[u].w) : t[u].w;
                *reinterpret_cast<uint4 *>(tab + gc) = o;
            } else if (w < slice_cells) {  // the table ends inside this piece
                const uint32_t dd[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
                for (uint32_t e = 0; e < 4; ++e)
                    if (gc + e < tab_cells && dd[e]) tab[gc + e] = fold(tab[gc + e], dd[e]);
            }
        }
    }
    if (sat) atomicAdd(sat_ctr, sat);
    if (viol) atomicAdd(sat_ctr - 1, viol);
    if (bad) *ta.flag = 1u;
}

}  // namespace psk
