// psk_window.hpp -- CountingBloomFilter update WINDOWS: small add / remove batches wait together and reach a big table in one pass
// over it, with the reference's per-batch semantics (round 4).
//
// Why: folding a big table (BASELINE cfg 4: 2^28 counters, 1 GiB) costs a pass over the WHOLE table whatever the batch brings, so
// a stream of 1 M-key batches only runs fast when many batches share one pass.  Adds commute; removes do not: countingbloom.py:186-208
// removes a key only if the min of its counters is non-zero (and not frozen at 2^32-1) AT THAT POINT of the stream.  Deferring a
// remove past later adds is therefore exact only if it would have succeeded at its own position.  Here that is PROVEN while folding:
//   * the window keeps its batches in arrival order; runs of same-type batches are PHASES (adds, removes, adds, ...);
//   * pass 1 (k_part_scatter<..., PayNonePhased>) runs once over the window's key list; a tile never straddles two phases, and a
//     (slice, workgroup) segment receives its tiles in order, so the end-of-phase fill counts (snap[phase][slice][workgroup]) cut
//     every segment into per-phase pieces;
//   * the fold (k_win_fold) keeps an image of the real counters of its table part in LDS -- four bits per counter (min(counter, 15)), a
//     whole 2^18-counter slice per workgroup (default), or a byte per counter (min(counter, 255)), 2^17 counters, two workgroups per slice
//     each applying the probes of its half -- and walks the slice's probes phase by phase: adds with a returning ds_add, barrier, removes
//     with a returning ds_sub whose old value must be 1 .. 14 (1 .. 254).
//     With T[c] the counter before a remove phase and R[c] the phase's probes on it:  T[c] >= R[c] for every c, none frozen  ==>
//     every key of the phase is removed whatever the order inside it (each of its counters holds >= 1 just before its own
//     decrement), and the result is T - R.  By induction over the phases the window's result is the sequential one.
//   * a remove that meets a zero (or a frozen / saturating counter) raises the window's flag: the host then UNDOES the fold
//     (k_win_fold<true>: the inverse net delta, wrapping arithmetic, exact) and replays the window batch by batch through the
//     validated per-batch path -- the reference's semantics for any stream, automatically.
// Counters of 14 (254) and more do not fit the image's arithmetic: a part that touches one drops its image and applies its probes, phase
// by phase, with wrapping atomics on the table itself (the part is its alone) -- same checks, exact.  The same for a (segment, phase) piece
// of more probe groups than the walk keeps in registers (12; 20 in the wide form that tables of few slices take).
#pragma once
#include "psk_nibble.hpp"

#include <type_traits>

namespace psk {

constexpr uint32_t kWinPartShift = 17;  // log2(counters of one workgroup's BYTE image): 128 KiB
constexpr uint32_t kWinNibShift = 18;   // ... of its NIBBLE image (round 4, second form): a whole 2^18-counter slice in 128 KiB
__host__ __device__ __forceinline__ uint32_t win_part_shift(uint32_t slice_shift, bool nib)
{
    const uint32_t cap = nib ? kWinNibShift : kWinPartShift;
    return slice_shift < cap ? slice_shift : cap;
}
constexpr int kWinMaxPhases = 192;   // (their 4-bit group counts sit next to the image: 192 x 128 bytes for 256 pass-1 workgroups)
struct WinPhases {
    uint32_t nph;
    const PhaseDesc *ph;                // device table of pass 1 (ph[p].remove; uniform reads)
};
// what a fold workgroup did to its table part (status[blockIdx.x]); the undo inverts exactly that
constexpr uint32_t kWinWritten = 0;   // image written back
constexpr uint32_t kWinAborted = 1;   // a remove met a zero: nothing written
constexpr uint32_t kWinAtomics = 2;   // applied with wrapping atomics on the table (a counter >= 254 in play)

// the six slice-local indices of a probe group that are valid (PayNone: two halves of 3 x 20 bits, valid count in bits 60..63)
template <class F>
__device__ __forceinline__ void win_each_probe(const uint4 &q, F &&f)
{
    const uint32_t n0 = q.y >> 28, n1 = q.w >> 28;
    const unsigned long long h0 = ((unsigned long long)q.y << 32) | q.x, h1 = ((unsigned long long)q.w << 32) | q.z;
    if (n0 > 0) f((uint32_t)h0 & 0xFFFFFu);
    if (n0 > 1) f((uint32_t)(h0 >> 20) & 0xFFFFFu);
    if (n0 > 2) f((uint32_t)(h0 >> 40) & 0xFFFFFu);
    if (n1 > 0) f((uint32_t)h1 & 0xFFFFFu);
    if (n1 > 1) f((uint32_t)(h1 >> 20) & 0xFFFFFu);
    if (n1 > 2) f((uint32_t)(h1 >> 40) & 0xFFFFFu);
}

// The fold's probe-group prefetch is issued behind hipcc's back (inline asm) and waited for with explicit counts.  hipcc inserts its
// own s_waitcnt in front of every use of a loaded register, and across a loop's back edge its count is conservative: it drained ALL loads
// in flight at every phase, or every K-th phase, whatever the source looked like (five formulations measured: 3.1-3.4 ms per fold of
// BASELINE cfg 4's step).  vmcnt counts in order, so "the oldest R loads have landed" is s_waitcnt vmcnt(loads issued since).
// Rules that keep this safe: the registers are touched by nothing between win_gld4 and win_wait (checked in the ISA: no v_mov of them),
// win_wait names them as read-write operands so that no use can be scheduled in front of it, and every path out of the loop waits
// for vmcnt(0) before the registers can be given to anything else.
typedef uint32_t win_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void win_gld4(win_u32x4 &dst, const uint4 *p)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void win_wait(win_u32x4 &a, win_u32x4 &b, win_u32x4 &c)
{
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void win_wait(win_u32x4 &a, win_u32x4 &b, win_u32x4 &c, win_u32x4 &d, win_u32x4 &e)
{
    asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N) : "memory");
}
template <int N, int R>
__device__ __forceinline__ void win_wait_all(win_u32x4 (&q)[R])
{
    static_assert(R == 3 || R == 5, "win_wait names three or five registers");
    if constexpr (R == 3) win_wait<N>(q[0], q[1], q[2]);
    else win_wait<N>(q[0], q[1], q[2], q[3], q[4]);
}

// Which segment (pass-1 workgroup) of the slice a lane walks: wave w owns segments [w * spw, (w + 1) * spw), L lanes each.
struct WinLane {
    uint32_t seg, sub, L;
    bool active;
};
__device__ __forceinline__ WinLane win_lane(const PartGeom &g)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t spw = (g.nwg + kApplyWaves - 1) / kApplyWaves;  // <= 64 (pass 1 runs at most 64 * kApplyWaves workgroups)
    uint32_t L = 1;
    while (L * 2 * spw <= 64) L *= 2;
    WinLane w;
    w.L = L;
    w.sub = lane % L;
    w.seg = wave * spw + lane / L;
    w.active = lane / L < spw && w.seg < g.nwg;
    return w;
}

// plain walk, phase by phase (the rare paths: atomics on the table, undo).  op(index in slice, phase removes?)
template <class Op>
__device__ __forceinline__ void win_walk_simple(const PartGeom &g, const uint4 *buckets, const uint32_t *snap, const WinPhases &wp, uint32_t b,
                                                bool barriers, Op &&op)
{
    const WinLane wl = win_lane(g);
    const uint4 *src = buckets + seg_index(g, b, wl.active ? wl.seg : 0) * g.segcap;
    uint32_t lo = 0;
    for (uint32_t p = 0; p < wp.nph; ++p) {
        uint32_t hi = 0;
        if (wl.active) {
            hi = snap[((uint64_t)p * g.nbuckets + b) * g.nwg + wl.seg] & 0x7FFFFFFFu;  // (bit 31: the phase's type)
            hi = hi < g.segcap ? hi : g.segcap;
        }
        const bool rem = wp.ph[p].remove != 0;
        for (uint32_t gi = lo + wl.sub; gi < hi; gi += wl.L) {
            const uint4 q = src[gi];
            win_each_probe(q, [&](uint32_t x) { op(x, rem); });
        }
        lo = hi;
        if (barriers) {
            __threadfence();
            __syncthreads();
        }
    }
}

// wide: five instead of three probe groups per lane and phase, whole bytes instead of nibbles for the per-(phase, segment) group counts --
// tables of few slices, whose tiles bring long runs per slice (host's choice, psk_part_cbf_window.hip)
static inline size_t win_fold_lds(const PartGeom &g, uint32_t nph, bool nib, bool wide = false)
{
    const uint32_t pshift = win_part_shift(g.shift, nib);
    const size_t row = wide ? g.nwg : (g.nwg + 1) / 2;
    return ((size_t)1 << (nib ? pshift - 1 : pshift)) + (((size_t)nph * row + 3) & ~(size_t)3) + (((size_t)nph + 31) / 32) * 4 + 16;
}

// UNDO = false: apply the window to my table part (blockIdx.x = slice * parts + part); flag: a remove met a zero / a counter
// would freeze -- the window has to be undone and replayed.  UNDO = true: the exact inverse of what the forward launch did.
// dynamic LDS: the byte image, 2^min(shift, 17) bytes | 4-bit group counts [phases][ceil(nwg / 2)] | phase types (win_fold_lds)
// NIB (round 4): the image holds min(counter, 15) in four bits, so ONE workgroup takes a whole 2^18-counter slice and every probe group is
// decoded once -- with byte images two workgroups per slice each decoded all of the slice's groups and applied half (the walk is bound by
// that decoding).  Same proof: adds must meet 0 .. 13, removes 1 .. 14 (15 = a counter the image cannot follow: atomics on the table); a
// nibble that carries or borrows into its neighbour has raised the taint / violation flag first, and either flag discards the image.
template <bool UNDO, bool NIB, int RG = 3>
__global__ __launch_bounds__(kApplyThreads) void k_win_fold(uint32_t *tab, uint64_t tab_cells, PartGeom g, const uint4 *buckets, const uint32_t *snap,
                                                            WinPhases wp, uint32_t *status, uint32_t *flag, uint32_t nt, uint32_t *shadow_out)
{
    // shadow_out (forward fold with nibble images only; may be null): the kept 4-bit images of the lookups (psk_sketch::shadow) -- the fold
    // ends with exactly that image of every slice in LDS, so the lookups that follow a flush need not read the table again.  flag[1] tells
    // the host when a slice took the atomics instead (its image is void: the kept images are dropped).
    // nt: nontemporal loads / stores of the table part (round 4; scripts/ubench/tabpass.hip: a pass over a table far larger than the
    // Infinity Cache runs 8-12 % faster with them, and the window's probe groups keep the cache)
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t s_viol, s_taint;
    const uint32_t pshift = win_part_shift(g.shift, NIB);
    const uint32_t nparts = 1u << (g.shift - pshift);
    const uint32_t b = blockIdx.x / nparts, h = blockIdx.x % nparts;
    const uint32_t pieces = 1u << (pshift - 2);  // 16-byte pieces of my table part = words (bytes) / 16-bit halves (nibbles) of the image
    const uint32_t img_words = NIB ? pieces / 2 : pieces;
    constexpr uint32_t kTop = NIB ? 15u : 255u;   // the marker: a counter the image cannot follow
    uint16_t *half16 = reinterpret_cast<uint16_t *>(smem);
    const uint32_t pmask = (1u << pshift) - 1;
    const uint64_t c0 = ((uint64_t)b << g.shift) + ((uint64_t)h << pshift);
    if (c0 >= tab_cells) {
        if (!UNDO && threadIdx.x == 0) status[blockIdx.x] = kWinAborted;  // (a part past the table's end: nothing to do, nothing to undo)
        return;
    }
    uint32_t st_fwd = kWinWritten;
    if (UNDO) {
        st_fwd = status[blockIdx.x];
        if (st_fwd == kWinAborted) return;
    }
    // ---- applied with atomics on the table itself (forward: after the image gave up, below; undo: the inverse, any order)
    auto atomics_pass = [&](bool inverse) {
        uint32_t bad = 0;
        win_walk_simple(g, buckets, snap, wp, b, !inverse, [&](uint32_t x, bool rem) {
            if ((x >> pshift) != h) return;
            uint32_t *cell = tab + c0 + (x & pmask);
            if (inverse) {
                if (rem) atomicAdd(cell, 1u);
                else atomicSub(cell, 1u);
            } else if (rem) {
                const uint32_t old = atomicSub(cell, 1u);  // countingbloom.py:198-206: zero -> not removed, 2^32-1 -> frozen
                bad |= (uint32_t)(old == 0u) | (uint32_t)(old == 0xFFFFFFFFu);
            } else {
                const uint32_t old = atomicAdd(cell, 1u);  // countingbloom.py:149-153 clamps at 2^32-1: the replay does that
                bad |= (uint32_t)(old >= 0xFFFFFFFEu);
            }
        });
        if (bad) *flag = 1u;
    };
    if (UNDO && st_fwd == kWinAtomics) {
        atomics_pass(true);
        return;
    }
    // ---- my part of the table -> byte image (min(counter, 255); 255 marks a counter the image cannot follow)
    if (threadIdx.x == 0) s_viol = s_taint = 0;
    {
        constexpr int U = 8;
        for (uint32_t p0 = threadIdx.x; p0 < pieces; p0 += kApplyThreads * U) {
            uint4 t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pc = p0 + (uint32_t)u * kApplyThreads;
                t[u] = pc < pieces ? nib_load_piece(tab, tab_cells, c0 + 4ULL * pc, nt != 0) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pc = p0 + (uint32_t)u * kApplyThreads;
                auto by = [](uint32_t c) -> uint32_t { return c < 255u ? c : 255u; };
                if (pc < pieces) {
                    if (NIB) half16[pc] = (uint16_t)nib_pack4(t[u]);
                    else smem[pc] = by(t[u].x) | (by(t[u].y) << 8) | (by(t[u].z) << 16) | (by(t[u].w) << 24);
                }
            }
        }
    }
    __syncthreads();
    if (UNDO) {
        // the inverse net delta, in any order: word arithmetic modulo 2^32 -- carries between the bytes cancel, every byte ends
        // where it started (all of them were below 255 when the forward launch wrote them)
        win_walk_simple(g, buckets, snap, wp, b, false, [&](uint32_t x, bool rem) {
            if ((x >> pshift) != h) return;
            const uint32_t c = x & pmask, one = NIB ? 1u << ((c & 7u) * 4u) : 1u << ((c & 3u) * 8u);
            uint32_t *word = &smem[NIB ? c >> 3 : c >> 2];
            if (rem) atomicAdd(word, one);
            else atomicSub(word, one);
        });
        __syncthreads();
    } else {
        // ---- the phases, in order.  A phase is at most ~2 pass-1 tiles per workgroup (the host cuts longer ones): ~1.4 probe groups per
        // lane, far too little to hide an HBM round trip behind, so the groups are requested K phases ahead.  Their addresses follow from
        // the snapshots alone: those are turned into per-(phase, segment) group COUNTS first (4 bits each, next to the image in LDS --
        // a segment-phase of more than R * L groups does not fit the registers anyway and sends the part to the atomics), so that the
        // walk's only global loads are the groups themselves, R per lane and phase, unconditional (clamped address; whether the lane
        // has a group is applied where the group is used): hipcc can then count the loads in flight and wait for the oldest only
        // (s_waitcnt vmcnt(n) counts in order; with loads under branches, or loaded registers copied / selected before their phase,
        // it drains every load at every phase -- the first version: 3.3 us per phase).
        constexpr int K = 3, R = RG;  // phases in flight; groups per lane and phase
        constexpr bool WIDE = RG > 3;  // (byte-wide group counts: up to R * L = 20 groups per segment and phase)
        const WinLane wl = win_lane(g);
        const uint4 *src = buckets + seg_index(g, b, wl.active ? wl.seg : 0) * g.segcap;
        const uint32_t nph = wp.nph;
        uint8_t *cnt4 = reinterpret_cast<uint8_t *>(smem + img_words);     // [nph][ceil(nwg / 2)]: two segments per byte (WIDE: [nph][nwg], one each)
        const uint32_t row = WIDE ? g.nwg : (g.nwg + 1) / 2;
        uint32_t *types = smem + img_words + ((nph * row + 3) / 4);  // bit p: phase p removes
        for (uint32_t i = threadIdx.x; i < (nph + 31) / 32; i += kApplyThreads) types[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nph * row; i += kApplyThreads) {
            const uint32_t p = i / row, s2 = (i - p * row) * (WIDE ? 1u : 2u);
            uint32_t byte = 0, ty = 0;
#pragma unroll
            for (uint32_t e = 0; e < (WIDE ? 1u : 2u); ++e) {
                if (s2 + e < g.nwg) {
                    const uint32_t cur = snap[((uint64_t)p * g.nbuckets + b) * g.nwg + s2 + e];
                    const uint32_t prev = p ? snap[((uint64_t)(p - 1) * g.nbuckets + b) * g.nwg + s2 + e] & 0x7FFFFFFFu : 0u;
                    ty = cur >> 31;
                    const uint32_t c1 = cur & 0x7FFFFFFFu, hi = c1 < g.segcap ? c1 : g.segcap, lo = prev < g.segcap ? prev : g.segcap;
                    const uint32_t d = hi - lo;
                    if (WIDE) byte = d < 255u ? d : 255u;
                    else byte |= (d < 15u ? d : 15u) << (4 * e);
                }
            }
            cnt4[i] = (uint8_t)byte;
            if (s2 == 0 && ty) atomicOr(&types[p >> 5], 1u << (p & 31));
        }
        __syncthreads();
        uint32_t viol = 0, taint = 0;
        auto count = [&](uint32_t p) -> uint32_t {  // groups of my segment in phase p (0 past the end / for an idle lane)
            if (p >= nph || !wl.active) return 0u;
            const uint32_t d = WIDE ? (uint32_t)cnt4[p * row + wl.seg] : (cnt4[p * row + (wl.seg >> 1)] >> (4 * (wl.seg & 1))) & 15u;
            return d;
        };
        // one probe group: the returning LDS atomics of its probes (mine: valid slot, my half of the slice) issue back to back, then the
        // old bytes are judged.  rm: all ones in a remove phase, else 0.  The atomics run under the lane's own predicate: the LDS pipe
        // is what bounds a phase (~3 lanes per clock and CU for random returning atomics), so a probe of the other half must not cost
        // a slot -- a fully branch-free version that added 0 for those took 3.3 us per phase, as long as the waits it replaced.
        // (per group: ~12 VALU per probe -- the fold is bound by VALU issue, 4 cycles per wave64 instruction with four waves per SIMD,
        // not by its waits: three applies per lane and phase at ~130 instructions each were 2.7 us per phase)
        const uint32_t amask = pmask & ~3u;          // byte address of the counter's word in the byte image
        const uint32_t hmask = (1u << (20 - pshift)) - 1u;  // which part of the slice a 20-bit index belongs to
        auto apply = [&](const win_u32x4 &q, uint32_t rm) {
            const uint32_t n0 = q.y >> 28, n1 = q.w >> 28;
            const uint32_t x[6] = {q.x, __builtin_amdgcn_alignbit(q.y, q.x, 20), q.y >> 8, q.z, __builtin_amdgcn_alignbit(q.w, q.z, 20), q.w >> 8};
            const uint32_t pm = rm | 1u;             // +1 (adds) or -1 (removes): shifted into the counter's byte lane
            uint32_t ob[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const bool mine = (uint32_t)(e % 3) < (e < 3 ? n0 : n1) && ((x[e] >> pshift) & hmask) == h;
                const uint32_t sh = NIB ? (x[e] & 7u) * 4u : (x[e] & 3u) * 8u;
                const uint32_t waddr = NIB ? ((x[e] & pmask) >> 1) & ~3u : x[e] & amask;  // byte address of the counter's image word
                uint32_t old = NIB ? 0x11111111u : 0x01010101u;  // (a probe that is not mine: an old value nobody objects to)
                if (mine) old = atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(smem) + waddr), pm << sh);  // ds_add_rtn_u32
                ob[e] = (old >> sh) & kTop;
            }
            // removes: every old byte must be 1 .. 254 (0: the key is not there -- countingbloom.py:200-201; 255: beyond the image);
            // adds: 0 .. 253 (254 would become the marker)
            const uint32_t mn = min(min(min(ob[0], ob[1]), min(ob[2], ob[3])), min(ob[4], ob[5]));
            const uint32_t mx = max(max(max(ob[0], ob[1]), max(ob[2], ob[3])), max(ob[4], ob[5]));
            viol |= rm & (uint32_t)(mn == 0u);
            taint |= (uint32_t)(mx >= (kTop - 1u) + (rm & 1u));
        };
        // the R groups of a phase that brings `d` groups from `lo` on in my segment; a lane without a group re-reads the phase's
        // first one and remembers that it has none (qv: applied where the group is used)
        auto fetch = [&](win_u32x4 (&q)[R], uint32_t (&qv)[R], uint32_t lo, uint32_t d) {
            const uint32_t safe = lo < g.segcap ? lo : g.segcap - 1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t o = wl.sub + (uint32_t)r * wl.L;
                win_gld4(q[r], src + (o < d ? lo + o : safe));
                qv[r] = o < d ? 0xFFFFFFFFu : 0u;
            }
            taint |= (uint32_t)(d > (uint32_t)R * wl.L);  // more groups than the registers take (or the 4-bit count's marker): atomics
        };
        win_u32x4 Q[K][R];
        uint32_t QV[K][R];
        uint32_t lo_ahead = 0;  // first group of the next phase to be fetched
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t d = count((uint32_t)j);
            fetch(Q[j], QV[j], lo_ahead, d);
            lo_ahead += d;
        }
        for (uint32_t p0 = 0; p0 < nph; p0 += K) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint32_t p = p0 + (uint32_t)j;
                if (p < nph) {  // (uniform)
                    const uint32_t rm = 0u - ((types[p >> 5] >> (p & 31)) & 1u);
                    const uint32_t d = count(p + K);
                    win_wait_all<(K - 1) * R, R>(Q[j]);  // K * R loads in flight: this phase's are the oldest R
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (QV[j][r]) apply(Q[j][r], rm);  // (a lane without an r-th group sits it out; a wave without one skips it)
                    fetch(Q[j], QV[j], lo_ahead, d);  // phase p + K takes the registers over
                    lo_ahead += d;
                    lds_barrier();  // the next phase reads what this one left in the image (LDS only: the loads stay in flight)
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the prefetches past the last phase: their registers are still theirs
        if (viol) s_viol = 1u;
        if (taint) s_taint = 1u;
        __syncthreads();
        if (s_taint) {  // a counter of 254 or more in play: the image is void (bytes may have carried); the table part is still untouched
            if (threadIdx.x == 0) {
                status[blockIdx.x] = kWinAtomics;
                flag[1] = 1u;  // (a tally for tests: slices that took the atomics)
            }
            atomics_pass(false);
            if constexpr (NIB) {
                if (shadow_out) {
                    // the kept image of this slice from the table itself, now that the atomics are through: every lane's atomics have
                    // returned (they are returning ones), the barrier collects the workgroup, and the loads go past this CU's L1, which
                    // still holds the lines of the first load
                    __threadfence();
                    __syncthreads();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    uint16_t *dst = reinterpret_cast<uint16_t *>(shadow_out + (uint64_t)blockIdx.x * img_words);
                    constexpr int U = 8;
                    for (uint32_t p0 = threadIdx.x; p0 < pieces; p0 += kApplyThreads * U) {
                        uint4 t[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const uint32_t pc = p0 + (uint32_t)u * kApplyThreads;
                            t[u] = pc < pieces ? nib_load_piece(tab, tab_cells, c0 + 4ULL * pc, true) : make_uint4(0, 0, 0, 0);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const uint32_t pc = p0 + (uint32_t)u * kApplyThreads;
                            if (pc < pieces) dst[pc] = (uint16_t)nib_pack4(t[u]);
                        }
                    }
                }
            }
            return;
        }
        if (s_viol) {  // a remove met a zero: nothing of this part reaches the table; the host undoes the others and replays the window
            if (threadIdx.x == 0) {
                status[blockIdx.x] = kWinAborted;
                *flag = 1u;
            }
            return;
        }
        if (threadIdx.x == 0) status[blockIdx.x] = kWinWritten;
    }
    // ---- image -> table (a value of 255 / 15 is a counter nobody touched: it keeps its value)
    for (uint32_t pc = threadIdx.x; pc < pieces; pc += kApplyThreads) {
        const uint64_t gc = c0 + 4ULL * pc;
        uint32_t v[4];
        if (NIB) {
            const uint32_t w = half16[pc];
            v[0] = w & 15u; v[1] = (w >> 4) & 15u; v[2] = (w >> 8) & 15u; v[3] = w >> 12;
        } else {
            const uint32_t w = smem[pc];
            v[0] = w & 255u; v[1] = (w >> 8) & 255u; v[2] = (w >> 16) & 255u; v[3] = w >> 24;
        }
        const bool marked = v[0] == kTop || v[1] == kTop || v[2] == kTop || v[3] == kTop;
        if (!marked && gc + 3 < tab_cells) {
            if (nt) {
                psk_u32x4 o;
                o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
                __builtin_nontemporal_store(o, reinterpret_cast<psk_u32x4 *>(tab + gc));
            } else {
                *reinterpret_cast<uint4 *>(tab + gc) = make_uint4(v[0], v[1], v[2], v[3]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (v[e] != kTop && gc + e < tab_cells) tab[gc + e] = v[e];
        }
    }
    if constexpr (NIB && !UNDO) {
        if (shadow_out) {  // (nparts == 1: blockIdx.x is the slice; the image is read-only from here on)
            uint4 *dst = reinterpret_cast<uint4 *>(shadow_out + (uint64_t)blockIdx.x * img_words);
            const uint4 *src = reinterpret_cast<const uint4 *>(smem);
            for (uint32_t v = threadIdx.x; v < img_words / 4; v += kApplyThreads) dst[v] = src[v];
        }
    }
}

}  // namespace psk
