// psk_device.hpp -- gfx950 (CDNA4) device code of the sketch engine.
//
// Everything here is integer hashing + random 4-byte read/modify/write on an HBM-resident table:
// HBM/atomic bound, no MFMA.  One lane owns one key: it streams the key in (16 B/lane coalesced for
// the fixed-16 layout), runs the k independent FNV-1a chains interleaved for ILP, reduces each
// 64-bit hash mod m (mask for power-of-two m, exact Barrett otherwise) and issues the k table
// accesses back-to-back (fire-and-forget atomics for updates, k loads in flight for lookups).
//
// Reference semantics (pyprobables v0.7.0) are cited per function as file:line.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace psk {

constexpr uint64_t kFnvBasis = 14695981039346656037ULL;  // hashes.py:96
constexpr uint64_t kFnvPrime = 1099511628211ULL;         // hashes.py:97  (2^40 + 0x1B3)
constexpr int kBlock = 256;                              // 4 wavefronts of 64
constexpr int kGroup = 8;                                // hash chains interleaved per pass

// ------------------------------------------------------------------ hashing
// hashes.py:99-102  hval ^= e; hval *= prime (mod 2^64)
// prime = 2^40 + 0x1B3 and e < 2^32, so with x = lo ^ e:  lo' = low32(x * 0x1B3),  hi' = hi * 0x1B3 + high32(x * 0x1B3) + (x << 8).
// Four VALU instructions: v_xor_b32, v_mul_lo_u32 (hi * 0x1B3), v_lshl_add_u32 ((x << 8) + that), and ONE v_mad_u64_u32 that yields
// lo' and hi' at once: x * 0x1B3 + {0, addend} -- the carry of the low product lands in the high word by itself.
// scripts/ubench/alu.hip (profiles/r04_ubench_alu.txt; ns per step, wave64, 4 waves per SIMD): this order 7.40, rounds 1-3's order (two
// v_mad_u64_u32: x * 0x1B3, then hi * 0x1B3 + addend) 7.99, a split low / high state (v_mul_hi_u32 + v_mul_lo_u32 + shift-add + add on the
// high word) 10.85, hipcc's own expansion of the 64-bit multiply 9.4; the 32-bit chain of power-of-two tables 3.23.
// The chains of a non-power-of-two table are VALU bound, and four instructions of which three are full-rate VOP3 is the floor.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
// `pair`: the chain's own {0, addend} register pair.  Its low word is zero and stays zero -- the step writes the high word only -- but the
// asm names the pair as read-WRITE, so hipcc cannot know that and keeps the pair in place; told the truth it built a fresh {0, b} pair per
// step and re-materialised the zero every time (a v_mov per step: five instructions, 8.5 ns).
__device__ __forceinline__ uint64_t fnv_step(uint64_t h, uint32_t e, uint64_t &pair)
{
    const uint32_t x = (uint32_t)h ^ e;
    const uint32_t a = (uint32_t)(h >> 32) * 0x1B3u;
    pair = (pair & 0xFFFFFFFFull) | ((uint64_t)((x << 8) + a) << 32);
    uint64_t u;
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %1" : "=&v"(u), "+v"(pair) : "v"(x), "s"(0x1B3u) : "vcc");
    return u;
}
// the {0, addend} pairs of G interleaved chains: declared once per key, handed to every step
template <int G>
struct FnvPairs {
    uint64_t p[G];
    __device__ __forceinline__ FnvPairs()
    {
#pragma unroll
        for (int g = 0; g < G; ++g) p[g] = 0;
    }
};

// hashes.py:96  seeded offset basis
__device__ __forceinline__ uint64_t fnv_seed(uint32_t seed) { return kFnvBasis + 31ULL * (uint64_t)seed; }

template <int G>
__device__ __forceinline__ void fnv_init(uint64_t (&h)[G], uint32_t s0)
{
#pragma unroll
    for (int g = 0; g < G; ++g) h[g] = fnv_seed(s0 + g);
}

template <int G>
__device__ __forceinline__ void fnv_word(uint64_t (&h)[G], FnvPairs<G> &pr, uint32_t w)
{
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        uint32_t e = (w >> (8 * b)) & 0xFFu;
        asm volatile("" : "+v"(e));  // the byte once per key byte in a VGPR (see fnv_word32)
#pragma unroll
        for (int g = 0; g < G; ++g) h[g] = fnv_step(h[g], e, pr.p[g]);
    }
}

// Low 32 bits only.  prime = 2^40 + 0x1B3, so bits 0..31 of every state depend only on bits 0..31 of the
// previous one: when the table index is h mod 2^p with p <= 32 the upper halves are dead.
// (scripts/ubench/alu.hip: v_mul_lo_u32, v_lshl_add_u32 and SDWA forms all issue at 4 cycles per wave64, plain
// VOP2 v_xor_b32 at 2 -- so one multiply beats a 3-instruction shift-add chain, and the byte is extracted once
// per key byte (shared by the k chains) instead of through an SDWA operand on every xor.)
__device__ __forceinline__ uint32_t fnv_step32(uint32_t h, uint32_t e) { return (h ^ e) * 0x1B3u; }

template <int G>
__device__ __forceinline__ void fnv_init32(uint32_t (&h)[G], uint32_t s0)
{
#pragma unroll
    for (int g = 0; g < G; ++g) h[g] = (uint32_t)fnv_seed(s0 + g);
}

template <int G>
__device__ __forceinline__ void fnv_word32(uint32_t (&h)[G], uint32_t w)
{
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        uint32_t e = (w >> (8 * b)) & 0xFFu;
        asm volatile("" : "+v"(e));  // keep the extracted byte in a VGPR (no per-chain SDWA byte select)
#pragma unroll
        for (int g = 0; g < G; ++g) h[g] = fnv_step32(h[g], e);
    }
}

// ------------------------------------------------------------ key sources
// A source turns key i into `G` hashes for seeds s0..s0+G-1 (hashes.py:71-83 default_fnv_1a).

struct KeysFixed16 {  // uint8[n][16], 16-byte aligned: one global_load_dwordx4 per lane
    const uint4 *p;
    struct Key { uint4 w; };
    __device__ __forceinline__ Key load(uint64_t i) const { return Key{p[i]}; }
    // force the load to have landed here (lets a caller drain it BEFORE it issues unrelated stores)
    static __device__ __forceinline__ void pin(Key &k) { asm volatile("" : "+v"(k.w.x), "+v"(k.w.y), "+v"(k.w.z), "+v"(k.w.w)); }
    template <int G>
    __device__ __forceinline__ void hash(const Key &k, uint64_t, uint32_t s0, uint64_t (&h)[G]) const
    {
        fnv_init<G>(h, s0);
        FnvPairs<G> pr;
        fnv_word<G>(h, pr, k.w.x);
        fnv_word<G>(h, pr, k.w.y);
        fnv_word<G>(h, pr, k.w.z);
        fnv_word<G>(h, pr, k.w.w);
    }
    template <int G>
    __device__ __forceinline__ void hash32(const Key &k, uint64_t, uint32_t s0, uint32_t (&h)[G]) const
    {
        fnv_init32<G>(h, s0);
        fnv_word32<G>(h, k.w.x);
        fnv_word32<G>(h, k.w.y);
        fnv_word32<G>(h, k.w.z);
        fnv_word32<G>(h, k.w.w);
    }
};

// uint8[n][8], 8-byte aligned (64-bit integer ids, the other common fixed layout; round 5): one global_load_dwordx2 per lane, prefetched and
// hashed from registers like the 16-byte layout -- through the generic dword loop (loads in front of the chains, k rounded up to 8) 8-byte
// keys ran SLOWER than 16-byte ones (48.6 against 55.7 G inserts/s)
struct KeysFixed8 {
    const uint2 *p;
    struct Key { uint2 w; };
    __device__ __forceinline__ Key load(uint64_t i) const { return Key{p[i]}; }
    static __device__ __forceinline__ void pin(Key &k) { asm volatile("" : "+v"(k.w.x), "+v"(k.w.y)); }
    template <int G>
    __device__ __forceinline__ void hash(const Key &k, uint64_t, uint32_t s0, uint64_t (&h)[G]) const
    {
        fnv_init<G>(h, s0);
        FnvPairs<G> pr;
        fnv_word<G>(h, pr, k.w.x);
        fnv_word<G>(h, pr, k.w.y);
    }
    template <int G>
    __device__ __forceinline__ void hash32(const Key &k, uint64_t, uint32_t s0, uint32_t (&h)[G]) const
    {
        fnv_init32<G>(h, s0);
        fnv_word32<G>(h, k.w.x);
        fnv_word32<G>(h, k.w.y);
    }
};

// uint8[n][32], 16-byte aligned (SHA-256-sized digests as keys): two dwordx4 per lane, prefetched; through the generic dword loop the
// loads sat in front of the chains (27.9 G inserts/s for twice the hashing of a 16-byte key, i.e. 0.5 x its rate at best: 27.7)
struct KeysFixed32 {
    const uint4 *p;
    struct Key { uint4 a, b; };
    __device__ __forceinline__ Key load(uint64_t i) const { return Key{p[2 * i], p[2 * i + 1]}; }
    static __device__ __forceinline__ void pin(Key &k)
    {
        asm volatile("" : "+v"(k.a.x), "+v"(k.a.y), "+v"(k.a.z), "+v"(k.a.w), "+v"(k.b.x), "+v"(k.b.y), "+v"(k.b.z), "+v"(k.b.w));
    }
    template <int G>
    __device__ __forceinline__ void hash(const Key &k, uint64_t, uint32_t s0, uint64_t (&h)[G]) const
    {
        fnv_init<G>(h, s0);
        FnvPairs<G> pr;
        fnv_word<G>(h, pr, k.a.x); fnv_word<G>(h, pr, k.a.y); fnv_word<G>(h, pr, k.a.z); fnv_word<G>(h, pr, k.a.w);
        fnv_word<G>(h, pr, k.b.x); fnv_word<G>(h, pr, k.b.y); fnv_word<G>(h, pr, k.b.z); fnv_word<G>(h, pr, k.b.w);
    }
    template <int G>
    __device__ __forceinline__ void hash32(const Key &k, uint64_t, uint32_t s0, uint32_t (&h)[G]) const
    {
        fnv_init32<G>(h, s0);
        fnv_word32<G>(h, k.a.x); fnv_word32<G>(h, k.a.y); fnv_word32<G>(h, k.a.z); fnv_word32<G>(h, k.a.w);
        fnv_word32<G>(h, k.b.x); fnv_word32<G>(h, k.b.y); fnv_word32<G>(h, k.b.z); fnv_word32<G>(h, k.b.w);
    }
};

// Several 16-byte-key batches laid end to end WITHOUT being copied together (borrowed batches of the write-combined CBF updates):
// key i lives in batch j with start[j] <= i < start[j + 1].  j is guessed as floor(i * nb / n) -- exact for equal-sized batches, the
// usual stream -- and corrected by walking start[] (a few hundred bytes, cached); a binary search per key (7 dependent loads for 50
// batches) made pass 1 30 % slower (25 vs 19 us per 1 M keys).  The lookup sits in the key prefetch, one tile ahead of its use.
struct KeysFixed16Multi {
    const uint4 *const *base;   // device array [nb]
    const uint64_t *start;      // device array [nb + 1], start[0] = 0, start[nb] = n
    uint32_t nb;
    uint64_t inv;               // floor(2^64 * nb / n)
    using Key = KeysFixed16::Key;
    __device__ __forceinline__ Key load(uint64_t i) const
    {
        uint32_t j = (uint32_t)__umul64hi(i, inv);  // <= nb - 1 for i < n
        uint64_t lo = start[j];
        while (lo > i) lo = start[--j];
        while (j + 1 < nb && start[j + 1] <= i) lo = start[++j];
        return Key{base[j][i - lo]};
    }
    static __device__ __forceinline__ void pin(Key &k) { KeysFixed16::pin(k); }
    template <int G>
    __device__ __forceinline__ void hash(const Key &k, uint64_t i, uint32_t s0, uint64_t (&h)[G]) const { KeysFixed16{nullptr}.template hash<G>(k, i, s0, h); }
    template <int G>
    __device__ __forceinline__ void hash32(const Key &k, uint64_t i, uint32_t s0, uint32_t (&h)[G]) const { KeysFixed16{nullptr}.template hash32<G>(k, i, s0, h); }
};

// ---- keys that start anywhere: 16-byte windows of the blob per lane
// A key of `len` bytes at any address is read as 4-byte-aligned global_load_dwordx4 windows (adjacent lanes hold adjacent keys, so a wave's
// windows cover one contiguous stretch of the blob just as the fixed-16 layout's loads do), the next window requested before the current
// one is hashed, and the key's words are cut out of neighbouring dwords with v_alignbyte_b32 -- instead of one dependent
// global_load_ubyte in front of every step of the k chains (rounds 1-4).
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// The window [p, p + 16), p 4-byte aligned and inside a mapped page (its first dword holds a byte the caller owns), of which the caller
// uses the first `need` bytes.  ONE load, NO branch: a window that would reach into the next 4 KiB page although the caller needs nothing
// there -- the page may not exist -- is requested `sh` dwords further down, where it ends at the page boundary, and a two-stage select moves
// its dwords back into place.  (Round 5's first form chose between the dwordx4 and four clamped dword loads with an if: the two paths
// delivered the window in different registers, hipcc joined them with moves -- and therefore waited vmcnt(0) right behind every load; the
// windows of a 2048-key tile cost ~10 us of exposed latency, as much as everything else in the tile.)
__device__ __forceinline__ uint4 load_window16(const uint32_t *p, uint32_t need)
{
    const uint32_t in_page = (uint32_t)(uintptr_t)p & 0xFFFu;
    const uint32_t over = in_page > 0xFF0u ? in_page - 0xFF0u : 0u;          // bytes of the window beyond the page: 0, 4, 8, 12
    const uint32_t sh = (need + over <= 16u) ? over >> 2 : 0u;                // (need beyond the boundary: the next page holds key bytes, it exists)
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(p - sh);
    const bool s1 = (sh & 1u) != 0, s2 = (sh & 2u) != 0;
    const uint32_t x1 = s1 ? v.y : v.x, y1 = s1 ? v.z : v.y, z1 = s1 ? v.w : v.z, w1 = v.w;
    return make_uint4(s2 ? z1 : x1, s2 ? w1 : y1, z1, w1);
}

// word(w): four key bytes (little endian); byte(e): one.  len < 2^31.  `cur`: the key's first window (first_window16; a caller that
// hashes several keys requests all their first windows before it walks the first key)
__device__ __forceinline__ uint4 first_window16(const uint8_t *q, uint32_t len)
{
    const uint32_t a = (uint32_t)(uintptr_t)q & 3u;
    // (pointer arithmetic, no round trip through an integer: that would turn the window loads into FLAT instructions -- which also count
    // against lgkmcnt, so that every LDS barrier of pass 1 waited for the windows in flight)
    return load_window16(reinterpret_cast<const uint32_t *>(q - a), a + len);
}
template <class WordFn, class ByteFn>
__device__ __forceinline__ void walk_key_bytes(const uint8_t *q, uint4 cur, uint32_t len, WordFn &&word, ByteFn &&byte)
{
    const uint32_t a = (uint32_t)(uintptr_t)q & 3u;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(q - a);
    const uint32_t span = a + len;
    const uint32_t nwin = len ? (span + 15u) >> 4 : 0u;
    uint32_t left = len;
    for (uint32_t t = 0; t < nwin; ++t) {
        const uint32_t tn = t + 1 < nwin ? t + 1 : t;  // (the last window once more rather than a branch around the load)
        const uint4 nxt = load_window16(w + 4u * tn, span - 16u * tn);
        const uint32_t d[5] = {cur.x, cur.y, cur.z, cur.w, nxt.x};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t kw = __builtin_amdgcn_alignbyte(d[e + 1], d[e], a);
            if (left >= 4) {
                word(kw);
                left -= 4;
            } else {
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    if ((uint32_t)b < left) byte((kw >> (8 * b)) & 0xFFu);
                left = 0;
            }
        }
        cur = nxt;
    }
}
template <class WordFn, class ByteFn>
__device__ __forceinline__ void walk_key_bytes(const uint8_t *q, uint32_t len, WordFn &&word, ByteFn &&byte)
{
    if (len == 0) return;
    walk_key_bytes(q, first_window16(q, len), len, word, byte);
}

// 4-byte elements (code points of str keys): windows of four elements
template <class ElemFn>
__device__ __forceinline__ void walk_key_elems(const uint32_t *q, uint4 cur, uint32_t len, ElemFn &&elem)
{
    const uint32_t nwin = (len + 3u) >> 2;
    for (uint32_t t = 0; t < nwin; ++t) {
        const uint32_t tn = t + 1 < nwin ? t + 1 : t;
        const uint32_t left_n = len - 4u * tn;  // (>= 1; capped before the multiply: a key of 2^30 code points and more must not wrap)
        const uint4 nxt = load_window16(q + 4u * tn, 4u * (left_n < 4u ? left_n : 4u));
        const uint32_t d[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4u * t + (uint32_t)e < len) elem(d[e]);
        cur = nxt;
    }
}

constexpr uint32_t kKeyLenBig = 0xFFFFFFFFu;  // Key::len of a key of 2^31 elements or more: walked element by element from off[]

template <bool DWORDS>
struct KeysFixed {  // uint8[n][L]; DWORDS: L % 4 == 0 and base 4-byte aligned
    const uint8_t *p;
    uint32_t L;
    uint64_t n;  // keys the matrix holds
    struct Key { const uint8_t *q; };
    __device__ __forceinline__ Key load(uint64_t i) const { return Key{p + i * (uint64_t)L}; }
    static __device__ __forceinline__ void pin(Key &) {}
    template <int G>
    __device__ __forceinline__ void hash(const Key &k, uint64_t, uint32_t s0, uint64_t (&h)[G]) const
    {
        fnv_init<G>(h, s0);
        FnvPairs<G> pr;
        if (DWORDS) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(k.q);
            for (uint32_t j = 0; j < L / 4; ++j) fnv_word<G>(h, pr, q[j]);
        } else {
            walk_key_bytes(k.q, L, [&](uint32_t w) { fnv_word<G>(h, pr, w); },
                           [&](uint32_t e) {
#pragma unroll
                               for (int g = 0; g < G; ++g) h[g] = fnv_step(h[g], e, pr.p[g]);
                           });
        }
    }
    template <int G>
    __device__ __forceinline__ void hash32(const Key &k, uint64_t, uint32_t s0, uint32_t (&h)[G]) const
    {
        fnv_init32<G>(h, s0);
        if (DWORDS) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(k.q);
            for (uint32_t j = 0; j < L / 4; ++j) fnv_word32<G>(h, q[j]);
        } else {
            walk_key_bytes(k.q, L, [&](uint32_t w) { fnv_word32<G>(h, w); },
                           [&](uint32_t e) {
#pragma unroll
                               for (int g = 0; g < G; ++g) h[g] = fnv_step32(h[g], e);
                           });
        }
    }
};

// ONE key of at most 64 bytes passed BY VALUE in the kernel arguments: the value-returning single-key calls (`key in blm`, `cms.add(key)`;
// psk_capi.hip inline_key).  A kernel that reads its key from the pinned staging page pays a PCIe round trip before it can hash (~1.7 us of
// a ~10 us call, scripts/ubench/latency.hip); the kernel arguments are where the kernel's other operands already come from.  Every lane
// sees the same key (scalar loads); never part of with_source's layouts -- only the one-key launches are built for it.
struct KeysInline64 {
    uint32_t w[16];
    uint32_t L;  // bytes, <= 64
    struct Key {};
    __device__ __forceinline__ Key load(uint64_t) const { return {}; }
    static __device__ __forceinline__ void pin(Key &) {}
    template <int G>
    __device__ __forceinline__ void hash(const Key &, uint64_t, uint32_t s0, uint64_t (&h)[G]) const
    {
        fnv_init<G>(h, s0);
        FnvPairs<G> pr;
        const uint32_t nw = L >> 2;
        for (uint32_t j = 0; j < nw; ++j) fnv_word<G>(h, pr, w[j]);
        const uint32_t tail = w[nw < 16u ? nw : 15u];
        for (uint32_t b = 0; b < (L & 3u); ++b) {
            const uint32_t e = (tail >> (8u * b)) & 0xFFu;
#pragma unroll
            for (int g = 0; g < G; ++g) h[g] = fnv_step(h[g], e, pr.p[g]);
        }
    }
    template <int G>
    __device__ __forceinline__ void hash32(const Key &, uint64_t, uint32_t s0, uint32_t (&h)[G]) const
    {
        fnv_init32<G>(h, s0);
        const uint32_t nw = L >> 2;
        for (uint32_t j = 0; j < nw; ++j) fnv_word32<G>(h, w[j]);
        const uint32_t tail = w[nw < 16u ? nw : 15u];
        for (uint32_t b = 0; b < (L & 3u); ++b) {
            const uint32_t e = (tail >> (8u * b)) & 0xFFu;
#pragma unroll
            for (int g = 0; g < G; ++g) h[g] = fnv_step32(h[g], e);
        }
    }
};

template <class T>
struct KeysVarlen {  // elements T (uint8 bytes, or uint32 code points for str keys: hashes.py:98)
    const T *p;
    const uint64_t *off;
    uint64_t n;  // keys of the batch
    // (the key's POSITION in the blob, not its address: descriptors travel through LDS in pass 1's length sort, and a pointer read back
    // from there has no address space -- its loads would be FLAT instructions, which count against lgkmcnt as well and make every LDS
    // barrier wait for the windows in flight)
    struct Key { uint64_t at; uint32_t len; uint32_t pad; };
    __device__ __forceinline__ const T *ptr(const Key &k) const { return p + k.at; }
    static constexpr bool sorted = true;  // pass 1 hands a tile's keys to its lanes in order of length (psk_partition.hpp, src_sorted)
    // length class of the sort (0 .. 62): the exact length below 48 elements -- the lanes of a wave then agree on every step of the walk --,
    // steps of 16 up to 272, one class for the rest
    static __device__ __forceinline__ uint32_t len_class(const Key &k)
    {
        return k.len < 48u ? k.len : (k.len < 48u + 14u * 16u ? 48u + ((k.len - 48u) >> 4) : 62u);
    }
    __device__ __forceinline__ Key load(uint64_t i) const
    {
        const uint64_t a = off[i], len = off[i + 1] - a;
        return Key{a, len < 0x80000000ull ? (uint32_t)len : kKeyLenBig, 0u};
    }
    static __device__ __forceinline__ void pin(Key &) {}
    // the key's first window (requested for all keys of a thread before the first one is walked).  An EMPTY key owns no byte -- it may sit
    // at the very end of the blob, past the last mapped page: its window (never looked at) is taken at the blob's first element instead.
    __device__ __forceinline__ uint4 first(const Key &k) const
    {
        const T *q = k.len ? p + k.at : p;
        if constexpr (sizeof(T) == 1) return first_window16(reinterpret_cast<const uint8_t *>(q), k.len);
        else return load_window16(reinterpret_cast<const uint32_t *>(q), 4u * (k.len < 4u ? k.len : 4u));
    }
    template <int G>
    __device__ __forceinline__ void hash_first(const Key &k, const uint4 &w0, uint64_t i, uint32_t s0, uint64_t (&h)[G]) const
    {
        fnv_init<G>(h, s0);
        FnvPairs<G> pr;
        auto step = [&](uint32_t e) {
#pragma unroll
            for (int g = 0; g < G; ++g) h[g] = fnv_step(h[g], e, pr.p[g]);
        };
        if (k.len == kKeyLenBig) {  // 2 G elements in one key: the plain walk
            const uint64_t len = off[i + 1] - off[i];
            for (uint64_t j = 0; j < len; ++j) step((uint32_t)ptr(k)[j]);
            return;
        }
        if constexpr (sizeof(T) == 1) walk_key_bytes(reinterpret_cast<const uint8_t *>(ptr(k)), w0, k.len, [&](uint32_t w) { fnv_word<G>(h, pr, w); }, step);
        else walk_key_elems(reinterpret_cast<const uint32_t *>(ptr(k)), w0, k.len, step);
    }
    template <int G>
    __device__ __forceinline__ void hash32_first(const Key &k, const uint4 &w0, uint64_t i, uint32_t s0, uint32_t (&h)[G]) const
    {
        fnv_init32<G>(h, s0);
        auto step = [&](uint32_t e) {  // code points > 255 XOR whole into the low word as well
#pragma unroll
            for (int g = 0; g < G; ++g) h[g] = fnv_step32(h[g], e);
        };
        if (k.len == kKeyLenBig) {
            const uint64_t len = off[i + 1] - off[i];
            for (uint64_t j = 0; j < len; ++j) step((uint32_t)ptr(k)[j]);
            return;
        }
        if constexpr (sizeof(T) == 1) walk_key_bytes(reinterpret_cast<const uint8_t *>(ptr(k)), w0, k.len, [&](uint32_t w) { fnv_word32<G>(h, w); }, step);
        else walk_key_elems(reinterpret_cast<const uint32_t *>(ptr(k)), w0, k.len, step);
    }
    template <int G>
    __device__ __forceinline__ void hash(const Key &k, uint64_t i, uint32_t s0, uint64_t (&h)[G]) const { hash_first<G>(k, first(k), i, s0, h); }
    template <int G>
    __device__ __forceinline__ void hash32(const Key &k, uint64_t i, uint32_t s0, uint32_t (&h)[G]) const { hash32_first<G>(k, first(k), i, s0, h); }
};

struct KeysHashes {  // uint64[n][stride] pre-computed hashes (add_alt / check_alt, custom hash_function)
    const uint64_t *p;
    uint32_t stride;
    struct Key { const uint64_t *q; };
    __device__ __forceinline__ Key load(uint64_t i) const { return Key{p + i * (uint64_t)stride}; }
    static __device__ __forceinline__ void pin(Key &) {}
    template <int G>
    __device__ __forceinline__ void hash(const Key &k, uint64_t, uint32_t s0, uint64_t (&h)[G]) const
    {
#pragma unroll
        for (int g = 0; g < G; ++g) h[g] = (s0 + g < stride) ? k.q[s0 + g] : 0;  // callers may round G up past k
    }
    template <int G>
    __device__ __forceinline__ void hash32(const Key &k, uint64_t, uint32_t s0, uint32_t (&h)[G]) const
    {
#pragma unroll
        for (int g = 0; g < G; ++g) h[g] = (s0 + g < stride) ? (uint32_t)k.q[s0 + g] : 0;
    }
};

// ------------------------------------------------------------------ h % m
// bloom.py:247 / countingbloom.py:145 / countminsketch.py:275 use an exact 64-bit modulo.
struct Mod {
    uint64_t m;      // divisor, 1 <= m < 2^63
    uint64_t magic;  // floor(2^64 / m) (non power-of-two m)
    uint64_t mask;   // m - 1 (power-of-two m)
};

template <bool POW2>
__device__ __forceinline__ uint64_t reduce(const Mod &md, uint64_t h)
{
    if (POW2) return h & md.mask;
    // q in {floor(h/m) - 1, floor(h/m)}  =>  one conditional correction makes it exact
    const uint64_t q = __umul64hi(h, md.magic);
    const uint64_t r = h - q * md.m;
    return r >= md.m ? r - md.m : r;
}

// h mod m for a non-power-of-two m <= 2^31 (every table the single-level partitioned path takes: at most 2048 slices of 2^20
// cells).  q = floor(h * magic / 2^64) is floor(h/m) or one less, so r = h - q*m < 2m <= 2^32 lives in the low word and only the
// low word of q is needed: 10 VALU instructions against the 21 of reduce<false> (whose 64-bit q*m and compare are dead here).
__device__ __forceinline__ uint32_t reduce_small(const Mod &md, uint64_t h)
{
    const uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32);
    const uint32_t ml = (uint32_t)md.magic, mh = (uint32_t)(md.magic >> 32), m = (uint32_t)md.m;
    const uint64_t p1 = (uint64_t)hi * ml + __umulhi(lo, ml);
    const uint64_t p2 = (uint64_t)lo * mh + (uint32_t)p1;
    const uint32_t q = hi * mh + (uint32_t)(p1 >> 32) + (uint32_t)(p2 >> 32);  // low word of the 128-bit product's high half
    const uint32_t r = lo - q * m;
    const uint32_t r2 = r - m;  // wraps to something huge when r < m
    return r2 < r ? r2 : r;
}

// ------------------------------------------------------- the generic kernel
// for_each_hash: f(j, hash_j) for j < k, hashing kGroup seeds at a time so the independent
// xor-multiply chains interleave (ILP) and the k table accesses issue back-to-back (MLP).
template <int G, class Src, class F>
__device__ __forceinline__ void hash_group(const Src &src, const typename Src::Key &key, uint64_t i, uint32_t s0, F &f)
{
    uint64_t h[G];
    src.template hash<G>(key, i, s0, h);
#pragma unroll
    for (int g = 0; g < G; ++g) f(s0 + g, h[g]);
}

template <class Src, class F>
__device__ __forceinline__ void for_each_hash(const Src &src, const typename Src::Key &key, uint64_t i, uint32_t k, F f)
{
    uint32_t s = 0;
    for (; s + kGroup <= k; s += kGroup) hash_group<kGroup>(src, key, i, s, f);
    switch (k - s) {  // wave-uniform
        case 7: hash_group<7>(src, key, i, s, f); break;
        case 6: hash_group<6>(src, key, i, s, f); break;
        case 5: hash_group<5>(src, key, i, s, f); break;
        case 4: hash_group<4>(src, key, i, s, f); break;
        case 3: hash_group<3>(src, key, i, s, f); break;
        case 2: hash_group<2>(src, key, i, s, f); break;
        case 1: hash_group<1>(src, key, i, s, f); break;
        default: break;
    }
}

// Completion mailbox of the value-returning single-key calls (`key in blm`, `cms.add(key)`: the reference's whole interface is per key).
// The results of a tiny PSK_HOST batch are written straight into a pinned host page; behind them the ONE workgroup of the launch stores the
// call's sequence number into a pinned word, and the host polls that word instead of waiting for the stream (psk_capi.hip finish()): the
// call ends when the answer is in host memory, not when the command processor has retired the kernel and a barrier packet behind it.
__device__ __forceinline__ void mailbox_post(uint32_t *mbox, uint32_t seq)  // by ONE lane, behind a system-scope fence of every writer
{
    __hip_atomic_store(mbox, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Op interface:  uint32_t k;  void prepare();  State begin(i);  void apply(State&, j, hash);  void end(State&, i)
//                optional: void merge(State &a, const State &b) -- folds what two lanes gathered for the SAME key (begin() is its identity)
// `mbox`: only with a one-workgroup grid (see mailbox_post), else nullptr
template <class Op, class = void>
struct op_has_merge : std::false_type {};
template <class Op>
struct op_has_merge<Op, std::void_t<decltype(&Op::merge)>> : std::true_type {};
template <class T>
__device__ __forceinline__ T shfl_xor_words(const T &v, int o)  // a State of 32-bit words from lane ^ o
{
    static_assert(sizeof(T) % 4 == 0, "State: whole 32-bit words");
    T r;
    uint32_t w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (unsigned e = 0; e < sizeof(T) / 4; ++e) w[e] = (uint32_t)__shfl_xor((int)w[e], o);
    __builtin_memcpy(&r, w, sizeof(T));
    return r;
}
template <class Src, class Op>
__global__ __launch_bounds__(kBlock) void k_apply(Src src, Op op, uint64_t n, uint32_t *mbox, uint32_t seq)
{
    op.prepare();
    const uint32_t k = op.k;
    bool one_done = false;
    if constexpr (op_has_merge<Op>::value) {
        // ONE key on one wave (the reference's per-key calls: `key in blm`): lane j runs hash chain j and its probe -- the k chains of a key
        // on one lane are ~450 dependent-issue VALU instructions for a 16-byte key, 1.5 us of a 3.6 us kernel -- and a butterfly folds the answers
        if (n == 1 && gridDim.x == 1 && blockDim.x == 64 && k <= 64) {
            const typename Src::Key key = src.load(0);
            typename Op::State st = op.begin(0);
            if (threadIdx.x < k) {
                uint64_t h[1];
                src.template hash<1>(key, 0, threadIdx.x, h);
                op.apply(st, threadIdx.x, h[0]);
            }
            if constexpr (!std::is_empty<typename Op::State>::value) {
                for (int o = 32; o > 0; o >>= 1) {
                    const typename Op::State other = shfl_xor_words(st, o);
                    op.merge(st, other);
                }
            }
            if (threadIdx.x == 0) op.end(st, 0);
            one_done = true;
        }
    }
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < (one_done ? 0 : n); i += stride) {
        const typename Src::Key key = src.load(i);
        typename Op::State st = op.begin(i);
        for_each_hash(src, key, i, k, [&](uint32_t j, uint64_t h) { op.apply(st, j, h); });
        op.end(st, i);
    }
    if (mbox) {  // (uniform)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) mailbox_post(mbox, seq);
    }
}

// k_apply that only runs when a device-side flag is raised (the rare exact redo of a split lookup)
template <class Src, class Op>
__global__ __launch_bounds__(kBlock) void k_apply_if(const uint32_t *flag, Src src, Op op, uint64_t n)
{
    if (*flag == 0) return;
    op.prepare();
    const uint32_t k = op.k;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const typename Src::Key key = src.load(i);
        typename Op::State st = op.begin(i);
        for_each_hash(src, key, i, k, [&](uint32_t j, uint64_t h) { op.apply(st, j, h); });
        op.end(st, i);
    }
}

// ---------------------------------------------------------------- Bloom ops
struct Empty {};

template <bool POW2>
struct BloomAdd {  // bloom.py:241-250 add_alt: bloom[k//8] |= 1 << (k%8)  == uint32 word k>>5, bit k&31 (LE)
    uint32_t *tab;
    Mod md;
    uint32_t k;
    using State = Empty;
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ State begin(uint64_t) const { return {}; }
    __device__ __forceinline__ void apply(State &, uint32_t, uint64_t h) const
    {
        const uint64_t b = reduce<POW2>(md, h);
        atomicOr(tab + (b >> 5), 1u << (uint32_t)(b & 31));  // result unused -> no-return global_atomic_or
    }
    __device__ __forceinline__ void end(State &, uint64_t) const {}
    __device__ __forceinline__ void merge(State &, const State &) const {}
};

template <bool POW2>
struct BloomCheck {  // bloom.py:261-272 check_alt (AND of the k bits; early exit does not change the result)
    const uint32_t *tab;
    Mod md;
    uint32_t k;
    uint8_t *out;
    struct State { uint32_t ok; };
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ State begin(uint64_t) const { return State{1u}; }
    __device__ __forceinline__ void apply(State &st, uint32_t, uint64_t h) const
    {
        const uint64_t b = reduce<POW2>(md, h);
        st.ok &= (tab[b >> 5] >> (uint32_t)(b & 31)) & 1u;
    }
    __device__ __forceinline__ void end(State &st, uint64_t i) const { out[i] = (uint8_t)st.ok; }
    __device__ __forceinline__ void merge(State &a, const State &b) const { a.ok &= b.ok; }
};

// the k bit positions of every key, for index-only follow-up kernels (psk_index_ops.hip); m <= 2^32
template <bool POW2>
struct BloomIndexOut {
    uint32_t *out;  // [n][k]
    Mod md;
    uint32_t k;
    struct State { uint64_t base; };
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ State begin(uint64_t i) const { return State{i * (uint64_t)k}; }
    __device__ __forceinline__ void apply(State &st, uint32_t j, uint64_t h) const { out[st.base + j] = (uint32_t)reduce<POW2>(md, h); }
    __device__ __forceinline__ void end(State &, uint64_t) const {}
};

// membership as a ballot bitmap + popcount of hits: one uint64 store per wavefront of 64 keys
template <class Src, bool POW2>
__global__ __launch_bounds__(kBlock) void k_bloom_check_bits(Src src, const uint32_t *tab, Mod md, uint32_t k,
                                                             uint64_t n, unsigned long long *out_bits,
                                                             unsigned long long *hits)
{
    BloomCheck<POW2> op{tab, md, k, nullptr};
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t nround = (n + 63) & ~63ULL;  // wave-uniform trip count
    unsigned long long my_hits = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nround; i += stride) {
        uint32_t ok = 0;
        if (i < n) {
            const typename Src::Key key = src.load(i);
            typename BloomCheck<POW2>::State st = op.begin(i);
            for_each_hash(src, key, i, k, [&](uint32_t j, uint64_t h) { op.apply(st, j, h); });
            ok = st.ok;
        }
        const unsigned long long bal = __ballot(ok != 0);
        if ((threadIdx.x & 63) == 0) {
            out_bits[i >> 6] = bal;
            my_hits += (unsigned long long)__popcll(bal);
        }
    }
    if ((threadIdx.x & 63) == 0 && my_hits) atomicAdd(hits, my_hits);
}

// ------------------------------------------------------------------ CMS ops
// saturating signed add through CAS (only taken when the wrap-free bound does not hold)
__device__ __forceinline__ void cms_sat_add(int32_t *p, int64_t w, unsigned long long *sat_ctr)
{
    int32_t old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        int64_t v = (int64_t)old + w;
        bool sat = false;
        if (v > INT32_MAX) { v = INT32_MAX; sat = true; }   // countminsketch.py:280-282
        if (v < INT32_MIN) { v = INT32_MIN; sat = true; }   // countminsketch.py:314-316
        if (__hip_atomic_compare_exchange_strong(p, &old, (int32_t)v, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)) {
            if (sat) atomicAdd(sat_ctr, 1ULL);
            return;
        }
    }
}

template <bool POW2, bool NEG>
struct CmsAdd {  // countminsketch.py:267-288 add_alt / :300-321 remove_alt (NEG): bin = h%width + i*width
    int32_t *bins;
    Mod md;  // m = width
    uint32_t k;  // depth
    const int32_t *weights;  // nullptr -> 1
    const long long *ctr;    // device counters; ctr[PSK_CTR_ABS_BOUND] chooses the path
    unsigned long long *sat_ctr;
    bool fast;
    struct State { int32_t w; };
    __device__ __forceinline__ void prepare() { fast = ctr[4] <= (long long)INT32_MAX; }
    __device__ __forceinline__ State begin(uint64_t i) const
    {
        int32_t w = weights ? weights[i] : 1;
        return State{w};
    }
    __device__ __forceinline__ void apply(State &st, uint32_t j, uint64_t h) const
    {
        const uint64_t bin = reduce<POW2>(md, h) + (uint64_t)j * md.m;
        if (fast) {
            // |bin| can never reach a rail: plain wrap-free atomic add, order-free, fire-and-forget
            atomicAdd(bins + bin, NEG ? (int32_t)(0u - (uint32_t)st.w) : st.w);
        } else {
            cms_sat_add(bins + bin, NEG ? -(int64_t)st.w : (int64_t)st.w, sat_ctr);
        }
    }
    __device__ __forceinline__ void end(State &, uint64_t) const {}
};

__device__ __forceinline__ void sort_small(int64_t *v, uint32_t n)
{
    for (uint32_t a = 1; a < n; ++a) {
        const int64_t x = v[a];
        uint32_t b = a;
        while (b > 0 && v[b - 1] > x) { v[b] = v[b - 1]; --b; }
        v[b] = x;
    }
}

__device__ __forceinline__ int64_t floordiv(int64_t a, int64_t b)  // Python's //
{
    int64_t q = a / b;
    const int64_t r = a % b;
    if (r != 0 && ((r < 0) != (b < 0))) --q;
    return q;
}

template <bool POW2>
struct CmsCheck {  // countminsketch.py:332-340 check_alt with the min (:429-432) or mean (:434-436) query
    const int32_t *bins;
    Mod md;
    uint32_t k;
    int32_t *out;
    bool mean;
    struct State { int32_t mn; int64_t sum; };
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ State begin(uint64_t) const { return State{INT32_MAX, 0}; }
    __device__ __forceinline__ void apply(State &st, uint32_t j, uint64_t h) const
    {
        const int32_t v = bins[reduce<POW2>(md, h) + (uint64_t)j * md.m];
        st.mn = v < st.mn ? v : st.mn;
        st.sum += v;
    }
    __device__ __forceinline__ void end(State &st, uint64_t i) const
    {
        out[i] = mean ? (int32_t)floordiv(st.sum, (int64_t)k) : st.mn;
    }
    __device__ __forceinline__ void merge(State &a, const State &b) const
    {
        a.mn = b.mn < a.mn ? b.mn : a.mn;
        a.sum += b.sum;
    }
};

constexpr int kMaxDepthMeanMin = 64;

template <bool POW2>
struct CmsCheckMeanMin {  // countminsketch.py:438-453 mean-min query
    const int32_t *bins;
    Mod md;
    uint32_t k;
    int64_t els_added;
    int64_t *out;
    struct State { int64_t v[kMaxDepthMeanMin]; };
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ State begin(uint64_t) const { return State{}; }
    __device__ __forceinline__ void apply(State &st, uint32_t j, uint64_t h) const
    {
        st.v[j] = bins[reduce<POW2>(md, h) + (uint64_t)j * md.m];
    }
    __device__ __forceinline__ void end(State &st, uint64_t i) const
    {
        bool all_zero = true;
        for (uint32_t j = 0; j < k; ++j) all_zero &= (st.v[j] == 0);
        if (all_zero) { out[i] = 0; return; }  // :440-441 (sorted: first and last zero <=> all zero)
        for (uint32_t j = 0; j < k; ++j) {
            const int64_t diff = els_added - st.v[j];
            st.v[j] = st.v[j] - floordiv(diff, (int64_t)md.m - 1);
        }
        sort_small(st.v, k);
        out[i] = (k % 2 == 0) ? floordiv(st.v[k / 2] + st.v[k / 2 - 1], 2) : st.v[k / 2];
    }
};

// ------------------------------------------------------------------ CBF ops
__device__ __forceinline__ void cbf_sat_add(uint32_t *p, uint32_t w, unsigned long long *sat_ctr)
{
    uint32_t old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        uint64_t v = (uint64_t)old + w;
        const bool sat = v > 0xFFFFFFFFULL;  // countingbloom.py:149-151
        if (sat) v = 0xFFFFFFFFULL;
        if (__hip_atomic_compare_exchange_strong(p, &old, (uint32_t)v, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)) {
            if (sat) atomicAdd(sat_ctr, 1ULL);
            return;
        }
    }
}

// the decrement of a well-formed remove (countingbloom.py:203-206 with to_remove == num_els): frozen counters stay, a
// counter that would go below zero is left alone and tallied as a contract violation
__device__ __forceinline__ void cbf_sat_sub(uint32_t *p, uint32_t w, unsigned long long *viol_ctr)
{
    uint32_t old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (old == 0xFFFFFFFFu) return;
        if (old < w) { atomicAdd(viol_ctr, 1ULL); return; }
        if (__hip_atomic_compare_exchange_strong(p, &old, old - w, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
}

template <bool POW2>
struct CbfSub {  // unordered decrement of every index (no min pre-check): the write-combined remove of well-formed streams
    uint32_t *tab;
    Mod md;
    uint32_t k;
    const uint32_t *weights;
    unsigned long long *viol_ctr;
    struct State { uint32_t w; };
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ State begin(uint64_t i) const { return State{weights ? weights[i] : 1u}; }
    __device__ __forceinline__ void apply(State &st, uint32_t, uint64_t h) const { cbf_sat_sub(tab + reduce<POW2>(md, h), st.w, viol_ctr); }
    __device__ __forceinline__ void end(State &, uint64_t) const {}
};

// The transactional decrement by per-key amounts (psk_capi.hip cbf_remove_exact): wrapping subtraction that raises `flag` when a counter
// would go below zero or is frozen at 2^32-1 -- the reference's result (countingbloom.py:198-206) then depends on the order inside the
// batch; `inverse`: adds the same amounts back (exact: wrapping both ways)
template <bool POW2>
struct CbfSubChecked {
    uint32_t *tab;
    Mod md;
    uint32_t k;
    const uint32_t *amounts;
    uint32_t *flag;
    bool inverse;
    struct State { uint32_t w; };
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ State begin(uint64_t i) const { return State{amounts ? amounts[i] : 1u}; }
    __device__ __forceinline__ void apply(State &st, uint32_t, uint64_t h) const
    {
        if (st.w == 0) return;
        uint32_t *p = tab + reduce<POW2>(md, h);
        if (inverse) {
            atomicAdd(p, st.w);
        } else {
            const uint32_t old = atomicSub(p, st.w);
            if (old < st.w || old == 0xFFFFFFFFu) *flag = 1u;
        }
    }
    __device__ __forceinline__ void end(State &, uint64_t) const {}
};

template <bool POW2>
struct CbfAdd {  // countingbloom.py:135-155 add_alt (k independent increments; duplicates add twice)
    uint32_t *tab;
    Mod md;
    uint32_t k;
    const uint32_t *weights;
    const long long *ctr;
    unsigned long long *sat_ctr;
    bool fast;
    struct State { uint32_t w; };
    __device__ __forceinline__ void prepare() { fast = ctr[4] <= 0xFFFFFFFFLL; }
    __device__ __forceinline__ State begin(uint64_t i) const { return State{weights ? weights[i] : 1u}; }
    __device__ __forceinline__ void apply(State &st, uint32_t, uint64_t h) const
    {
        uint32_t *p = tab + reduce<POW2>(md, h);
        if (fast) atomicAdd(p, st.w);
        else cbf_sat_add(p, st.w, sat_ctr);
    }
    __device__ __forceinline__ void end(State &, uint64_t) const {}
};

template <bool POW2>
struct CbfCheck {  // countingbloom.py:166-174 check_alt: min over the hashes
    const uint32_t *tab;
    Mod md;
    uint32_t k;
    uint32_t *out;
    struct State { uint32_t mn; };
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ State begin(uint64_t) const { return State{0xFFFFFFFFu}; }
    __device__ __forceinline__ void apply(State &st, uint32_t, uint64_t h) const
    {
        const uint32_t v = tab[reduce<POW2>(md, h)];
        st.mn = v < st.mn ? v : st.mn;
    }
    __device__ __forceinline__ void end(State &st, uint64_t i) const { out[i] = st.mn; }
    __device__ __forceinline__ void merge(State &a, const State &b) const { a.mn = b.mn < a.mn ? b.mn : a.mn; }
};

constexpr int kMaxKOrdered = 64;  // ordered kernels keep the k indices of one key in registers

// countingbloom.py:186-208 remove_alt, unordered batch form.
// Phase A (gather): min over the k counters.  Phase B: conditional decrement of every index.
// Well-formed streams (every removed key has >= num_els live inserts, nothing saturated) make
// min_val >= num_els at every interleaving, so to_remove == num_els and the result is order-free.
// Anything else is tallied in ctr[VIOLATIONS]; underflowing decrements are rolled back.
template <class Src, bool POW2>
__global__ __launch_bounds__(kBlock) void k_cbf_remove(Src src, uint32_t *tab, Mod md, uint32_t k,
                                                       const uint32_t *weights, uint64_t n,
                                                       unsigned long long *ctr)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    unsigned long long removed = 0, viol = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const typename Src::Key key = src.load(i);
        const uint32_t w = weights ? weights[i] : 1u;
        uint32_t mn = 0xFFFFFFFFu;
        for_each_hash(src, key, i, k, [&](uint32_t, uint64_t h) {
            const uint32_t v = __hip_atomic_load(tab + reduce<POW2>(md, h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mn = v < mn ? v : mn;
        });
        if (mn == 0xFFFFFFFFu || mn == 0) continue;          // :198-201 frozen / absent: no-op
        const uint32_t to_remove = mn > w ? w : mn;          // :203
        if (to_remove != w) ++viol;                          // partial removal: order-dependent
        for_each_hash(src, key, i, k, [&](uint32_t, uint64_t h) {
            uint32_t *p = tab + reduce<POW2>(md, h);
            const uint32_t old = atomicSub(p, to_remove);
            if (old == 0xFFFFFFFFu || old < to_remove) {     // :205 frozen counter, or underflow
                atomicAdd(p, to_remove);                     // roll back
                ++viol;
            }
        });
        removed += to_remove;                                // :207
    }
    // wavefront reduction, one atomic per wave
    for (int o = 32; o > 0; o >>= 1) {
        removed += __shfl_down(removed, o);
        viol += __shfl_down(viol, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (removed) atomicAdd(ctr + 1, removed);
        if (viol) atomicAdd(ctr + 2, viol);
    }
}

// ------------------------------------------------- ordered (sequential) execution on the device
// One lane walks the batch in order and applies the reference semantics literally, including every
// op's return value: exact for ANY stream (ill-formed removes, saturation, mixed signs).
// `wide`: device scratch of 2 * k uint64 for k > kMaxKOrdered (the reference has no limit on k), else nullptr
// ONE op (the per-key API) with k <= 64: lane s takes probe s, so the k hash chains and the k table round trips overlap instead of
// queueing up behind one lane; a key whose probes collide (two probes, one counter -- the sequential semantics matter) and every longer
// batch walk the literal loop below.  `mbox`: see mailbox_post.
template <class Src, bool POW2>
__global__ void k_cbf_ordered(Src src, uint32_t *tab, Mod md, uint32_t k, const int64_t *weights, int opmode,
                              uint64_t n, uint32_t *out, unsigned long long *ctr, uint64_t *wide, uint32_t *mbox, uint32_t seq)
{
    if (blockIdx.x != 0) return;
    unsigned long long added = 0, removed = 0, sat = 0, abs_sum = 0;
    bool one_done = false;
    if (n == 1 && !wide && blockDim.x == 64) {
        const uint32_t s = threadIdx.x;
        const typename Src::Key key = src.load(0);
        int64_t w = weights ? weights[0] : 1;
        bool rem = opmode == 1;
        if (opmode == 2 && w < 0) { rem = true; w = -w; }
        uint64_t my = ~0ULL;  // (no counter)
        if (s < k) {
            uint64_t h[1];
            src.template hash<1>(key, 0, s, h);
            my = reduce<POW2>(md, h[0]);
        }
        bool dup = false;
        for (uint32_t q = 0; q < k; ++q) {
            const uint64_t other = __shfl((unsigned long long)my, (int)q);
            dup |= s < k && q != s && other == my;
        }
        if (__ballot(dup) == 0) {  // k different counters: the k read-modify-writes are independent
            uint32_t ret;
            if (!rem) {  // countingbloom.py:135-155
                uint64_t v = ~0ULL;
                bool s1 = false;
                if (s < k) {
                    v = (uint64_t)tab[my] + (uint64_t)w;
                    if (v > 0xFFFFFFFFULL) { v = 0xFFFFFFFFULL; s1 = true; }
                    tab[my] = (uint32_t)v;
                }
                for (int o = 32; o > 0; o >>= 1) {
                    const uint64_t u = __shfl_xor((unsigned long long)v, o);
                    v = u < v ? u : v;
                }
                sat = (unsigned long long)__popcll(__ballot(s1));
                added = (uint64_t)w;
                ret = (uint32_t)v;
            } else {  // countingbloom.py:186-208
                const uint32_t t = s < k ? tab[my] : 0xFFFFFFFFu;
                uint32_t mn = t;
                for (int o = 32; o > 0; o >>= 1) {
                    const uint32_t u = __shfl_xor(mn, o);
                    mn = u < mn ? u : mn;
                }
                if (mn == 0xFFFFFFFFu) ret = 0xFFFFFFFFu;
                else if (mn == 0) ret = 0;
                else {
                    const uint32_t tr = (uint64_t)mn > (uint64_t)w ? (uint32_t)w : mn;
                    if (s < k && t < 0xFFFFFFFFu) tab[my] = t - tr;
                    removed = tr;
                    ret = mn - tr;
                }
            }
            abs_sum = (unsigned long long)(w < 0 ? -w : w);
            if (s == 0 && out) out[0] = ret;
            one_done = true;
        }
    }
    if (threadIdx.x != 0) return;  // (out[0] is lane 0's own store: the fence in front of the mailbox below covers it)
    uint64_t idx_r[kMaxKOrdered], vals_r[kMaxKOrdered];
    uint64_t *idx = wide ? wide : idx_r, *vals = wide ? wide + k : vals_r;
    for (uint64_t i = 0; i < (one_done ? 0 : n); ++i) {
        const typename Src::Key key = src.load(i);
        int64_t w = weights ? weights[i] : 1;
        bool rem = opmode == 1;
        if (opmode == 2 && w < 0) { rem = true; w = -w; }
        abs_sum += (unsigned long long)(w < 0 ? -w : w);
        for (uint32_t s = 0; s < k; ++s) {
            uint64_t h[1];
            src.template hash<1>(key, i, s, h);
            idx[s] = reduce<POW2>(md, h[0]);
        }
        uint32_t ret;
        if (!rem) {  // countingbloom.py:135-155
            uint64_t mn = ~0ULL;
            for (uint32_t s = 0; s < k; ++s) vals[s] = (uint64_t)tab[idx[s]] + (uint64_t)w;  // :146 pre-read
            for (uint32_t s = 0; s < k; ++s) {
                if (vals[s] > 0xFFFFFFFFULL) { tab[idx[s]] = 0xFFFFFFFFu; vals[s] = 0xFFFFFFFFULL; ++sat; }
                else tab[idx[s]] += (uint32_t)w;                                            // :153
                mn = vals[s] < mn ? vals[s] : mn;
            }
            added += (uint64_t)w;
            ret = (uint32_t)mn;
        } else {  // countingbloom.py:186-208
            uint32_t mn = 0xFFFFFFFFu;
            for (uint32_t s = 0; s < k; ++s) mn = tab[idx[s]] < mn ? tab[idx[s]] : mn;
            if (mn == 0xFFFFFFFFu) ret = 0xFFFFFFFFu;
            else if (mn == 0) ret = 0;
            else {
                const uint32_t tr = (uint64_t)mn > (uint64_t)w ? (uint32_t)w : mn;
                for (uint32_t s = 0; s < k; ++s)
                    if (tab[idx[s]] < 0xFFFFFFFFu) tab[idx[s]] -= tr;
                removed += tr;
                ret = mn - tr;
            }
        }
        if (out) out[i] = ret;
    }
    ctr[0] += added;
    ctr[1] += removed;
    ctr[3] += sat;
    // every op moved a counter by at most k*|w|: keep the wrap-free bound an upper bound
    const unsigned long long nb = ctr[4] + abs_sum * (unsigned long long)k;
    ctr[4] = (nb < ctr[4] || nb > (1ULL << 62)) ? (1ULL << 62) : nb;
    if (mbox) {
        __threadfence_system();
        mailbox_post(mbox, seq);
    }
}

// opmode 3 (internal): query only -- every op adds 0, i.e. returns check()'s value under `query` and changes nothing
// `wide`: device scratch of `depth` int64 for depth > kMaxDepthMeanMin, else nullptr
// ONE op (the per-key API) with depth <= 64: lane s takes row s -- the rows are disjoint, so the depth hash chains and read-modify-writes
// overlap -- and lane 0 finishes the op (query, elements_added, tallies) from the values it collects.  `mbox`: see mailbox_post.
template <class Src, bool POW2>
__global__ void k_cms_ordered(Src src, int32_t *bins, Mod md, uint32_t depth, const int64_t *weights, int opmode,
                              int query, int64_t els, uint64_t n, int64_t *out, long long *ctr, int64_t *wide, uint32_t *mbox, uint32_t seq)
{
    if (blockIdx.x != 0) return;
    const bool lanes = n == 1 && !wide && blockDim.x == 64;  // (!wide: depth <= kMaxDepthMeanMin = 64)
    if (!lanes && threadIdx.x != 0) return;
    unsigned long long sat = 0, abs_sum = 0;
    int64_t vals_r[kMaxDepthMeanMin];
    int64_t *vals = wide ? wide : vals_r;
    for (uint64_t i = 0; i < n; ++i) {
        const typename Src::Key key = src.load(i);
        int64_t w = opmode == 3 ? 0 : (weights ? weights[i] : 1);
        bool rem = opmode == 1;
        if (opmode == 2 && w < 0) { rem = true; w = -w; }
        abs_sum += (unsigned long long)(w < 0 ? -w : w);
        if (lanes) {
            const uint32_t s = threadIdx.x;
            int64_t v = 0;
            bool s1 = false;
            if (s < depth) {
                uint64_t h[1];
                src.template hash<1>(key, i, s, h);
                int32_t *p = bins + reduce<POW2>(md, h[0]) + (uint64_t)s * md.m;
                v = (int64_t)*p + (rem ? -w : w);   // :276 / :309
                if (v > INT32_MAX) { v = INT32_MAX; s1 = true; }
                if (v < INT32_MIN) { v = INT32_MIN; s1 = true; }
                if (opmode != 3) *p = (int32_t)v;
            }
            sat += (unsigned long long)__popcll(__ballot(s1));
            for (uint32_t q = 0; q < depth; ++q) vals[q] = (int64_t)__shfl((long long)v, (int)q);
            if (threadIdx.x != 0) return;
        } else
        for (uint32_t s = 0; s < depth; ++s) {
            uint64_t h[1];
            src.template hash<1>(key, i, s, h);
            int32_t *p = bins + reduce<POW2>(md, h[0]) + (uint64_t)s * md.m;
            int64_t v = (int64_t)*p + (rem ? -w : w);   // :276 / :309
            if (v > INT32_MAX) { v = INT32_MAX; ++sat; }
            if (v < INT32_MIN) { v = INT32_MIN; ++sat; }
            *p = (int32_t)v;
            vals[s] = v;
        }
        // countminsketch.py:285-287 / :317-319  exact add then clamp at the int64 rails
        long long e2;
        if (__builtin_saddll_overflow((long long)els, (long long)(rem ? -w : w), &e2))
            e2 = (rem ? -w : w) > 0 ? INT64_MAX : INT64_MIN;
        els = e2;
        int64_t r;
        sort_small(vals, depth);
        if (query == 1) {  // mean :434-436
            int64_t sum = 0;
            for (uint32_t s = 0; s < depth; ++s) sum += vals[s];
            r = floordiv(sum, (int64_t)depth);
        } else if (query == 2) {  // mean-min :438-453
            if (vals[0] == 0 && vals[depth - 1] == 0) r = 0;
            else {
                for (uint32_t s = 0; s < depth; ++s)
                    vals[s] = vals[s] - floordiv(els - vals[s], (int64_t)md.m - 1);
                sort_small(vals, depth);
                r = (depth % 2 == 0) ? floordiv(vals[depth / 2] + vals[depth / 2 - 1], 2) : vals[depth / 2];
            }
        } else r = vals[0];
        if (out) out[i] = r;
    }
    if (opmode != 3) {
        ctr[5] = els;
        if (out) out[n] = els;  // out is int64[n + 1]: the caller gets elements_added with the results, no second read-back
        ctr[3] += (long long)sat;
        const unsigned long long nb = (unsigned long long)ctr[4] + abs_sum;
        ctr[4] = (nb < (unsigned long long)ctr[4] || nb > (1ULL << 62)) ? (1LL << 62) : (long long)nb;
    }
    if (mbox) {
        __threadfence_system();
        mailbox_post(mbox, seq);
    }
}

// ------------------------------------------------------------- hash only
struct StoreHashes {  // hashes.py:71-83: out[i][j] = fnv_1a(key_i, j)
    uint64_t *out;
    uint32_t k;
    using State = Empty;
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ State begin(uint64_t) const { return {}; }
    __device__ __forceinline__ void apply(State &, uint32_t, uint64_t) const {}
    __device__ __forceinline__ void end(State &, uint64_t) const {}
};

template <class Src>
__global__ __launch_bounds__(kBlock) void k_hash(Src src, uint64_t *out, uint32_t depth, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const typename Src::Key key = src.load(i);
        uint32_t s = 0;
        for (; s + kGroup <= depth; s += kGroup) {
            uint64_t h[kGroup];
            src.template hash<kGroup>(key, i, s, h);
#pragma unroll
            for (int g = 0; g < kGroup; ++g) out[i * depth + s + g] = h[g];
        }
        for (; s < depth; ++s) {
            uint64_t h[1];
            src.template hash<1>(key, i, s, h);
            out[i * depth + s] = h[0];
        }
    }
}

// --------------------------------------------------------- counter helpers
// sum of a weight vector into the per-sketch device counters (selects fast/saturating path, feeds
// elements_added).  which: 0 = ADDED, 1 = REMOVED, -1 = neither;  bound_mult: k for CBF, 1 for CMS.
// grow_bound = 0: the batch only lowers counters (CBF removes): ctr[4] is left alone, ctr[6] still gets the batch's sum
template <class W>
__global__ __launch_bounds__(kBlock) void k_weight_sum(const W *w, uint64_t n, long long *ctr, int which,
                                                       long long bound_mult, int grow_bound = 1)
{
    long long s = 0;
    unsigned long long a = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    auto acc = [&](W x) {
        const long long v = (long long)x;
        s += v;
        a += (unsigned long long)(v < 0 ? -v : v);
    };
    uint64_t done = 0;
    if (sizeof(W) == 4 && ((uintptr_t)w & 15) == 0) {  // 16-byte loads, two in flight per lane (a dword loop was latency bound)
        struct alignas(16) W4 { W x, y, z, t; };
        const W4 *w4 = reinterpret_cast<const W4 *>(w);
        const uint64_t n4 = n / 4;
        uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {  // four 16-byte loads in flight per lane
            const W4 p = w4[i], q = w4[i + stride], r = w4[i + 2 * stride], u = w4[i + 3 * stride];
            acc(p.x); acc(p.y); acc(p.z); acc(p.t);
            acc(q.x); acc(q.y); acc(q.z); acc(q.t);
            acc(r.x); acc(r.y); acc(r.z); acc(r.t);
            acc(u.x); acc(u.y); acc(u.z); acc(u.t);
        }
        for (; i < n4; i += stride) {
            const W4 p = w4[i];
            acc(p.x); acc(p.y); acc(p.z); acc(p.t);
        }
        done = n4 * 4;
    }
    for (uint64_t i = done + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc(w[i]);
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o);
        a += __shfl_down(a, o);
    }
    // one set of atomics per workgroup: same-address device atomics serialise (~11 ns each)
    __shared__ long long ps[kBlock / 64];
    __shared__ unsigned long long pa[kBlock / 64];
    if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = s; pa[threadIdx.x >> 6] = a; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = 0; a = 0;
        for (int w = 0; w < kBlock / 64; ++w) { s += ps[w]; a += pa[w]; }
        if (which >= 0 && s) atomicAdd((unsigned long long *)(ctr + which), (unsigned long long)s);
        if (a) {
            atomicAdd((unsigned long long *)(ctr + 6), a * (unsigned long long)bound_mult);  // this batch only (zeroed by the host before)
            // saturating bound: never wraps back below the threshold
            unsigned long long add = a * (unsigned long long)bound_mult;
            if (grow_bound) {
                unsigned long long old = atomicAdd((unsigned long long *)(ctr + 4), add);
                if (old + add < old || (long long)(old + add) < 0) atomicExch((unsigned long long *)(ctr + 4), 1ULL << 62);
            }
        }
    }
}

static __global__ void k_ctr_add(long long *ctr, int which, long long v, long long bound_add)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (which >= 0) ctr[which] += v;
        long long b = ctr[4] + bound_add;
        ctr[4] = (b < 0 || b > (1LL << 62)) ? (1LL << 62) : b;
    }
}

// max |counter| of a freshly loaded table re-seeds the wrap-free bound
static __global__ __launch_bounds__(kBlock) void k_absmax(const uint32_t *tab, uint64_t n, int is_signed, long long *ctr)
{
    unsigned long long mx = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t v = tab[i];
        unsigned long long a;
        if (is_signed) { const long long sv = (long long)(int32_t)v; a = (unsigned long long)(sv < 0 ? -sv : sv); }
        else a = v;
        mx = a > mx ? a : mx;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_down(mx, o);
        mx = t > mx ? t : mx;
    }
    if ((threadIdx.x & 63) == 0 && mx) atomicMax((unsigned long long *)(ctr + 4), mx);
}

// -------------------------------------------------------- table algebra
// All tables are padded to 16 B, so whole-table passes run on uint4 (global_load/store_dwordx4).
struct OpOr  { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a | b; } };
struct OpAnd { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a & b; } };

template <class F>
__global__ __launch_bounds__(kBlock) void k_table_binop(uint4 *dst, const uint4 *src, uint64_t nvec, F f)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
        uint4 a = dst[i];
        const uint4 b = src[i];
        a.x = f(a.x, b.x); a.y = f(a.y, b.y); a.z = f(a.z, b.z); a.w = f(a.w, b.w);
        dst[i] = a;
    }
}

// dst[w] = OR_j src[j*slice + w]: the reduce step of allreduce(OR) (bloom.py:425-426 union is a bytewise OR)
static __global__ __launch_bounds__(kBlock) void k_or_reduce(uint4 *dst, const uint4 *src, uint32_t nslices,
                                                      uint64_t slice_vec)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < slice_vec; i += stride) {
        uint4 a = src[i];
        for (uint32_t j = 1; j < nslices; ++j) {
            const uint4 b = src[(uint64_t)j * slice_vec + i];
            a.x |= b.x; a.y |= b.y; a.z |= b.z; a.w |= b.w;
        }
        dst[i] = a;
    }
}

// mode 0: popcount of bits (bloom.py:552-557); mode 1: number of non-zero uint32 (countingbloom.py:302-304)
static __global__ __launch_bounds__(kBlock) void k_table_count(const uint4 *tab, uint64_t nvec, int mode,
                                                        unsigned long long *out)
{
    unsigned long long c = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
        const uint4 a = tab[i];
        if (mode == 0) c += __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w);
        else c += (a.x != 0) + (a.y != 0) + (a.z != 0) + (a.w != 0);
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    // one atomic per workgroup (same-address device atomics serialise at ~11 ns each)
    __shared__ unsigned long long part[kBlock / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kBlock / 64; ++w) t += part[w];
        if (t) atomicAdd(out, t);
    }
}

// countminsketch.py:380-391 join: clamp-add, bins already on a rail stay frozen
static __global__ __launch_bounds__(kBlock) void k_add_sat_i32(int32_t *dst, const int32_t *src, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const int32_t a = dst[i];
        if (a == INT32_MIN || a == INT32_MAX) continue;
        int64_t t = (int64_t)a + (int64_t)src[i];
        t = t > INT32_MAX ? INT32_MAX : (t < INT32_MIN ? INT32_MIN : t);
        dst[i] = (int32_t)t;
    }
}

// countingbloom.py:296-298 union: plain element-wise sum (the reference raises on uint32 overflow;
// here overflowing elements are clamped and counted)
static __global__ __launch_bounds__(kBlock) void k_add_u32(uint32_t *dst, const uint32_t *src, uint64_t n,
                                                    unsigned long long *overflowed)
{
    unsigned long long ov = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t t = (uint64_t)dst[i] + (uint64_t)src[i];
        if (t > 0xFFFFFFFFULL) { dst[i] = 0xFFFFFFFFu; ++ov; }
        else dst[i] = (uint32_t)t;
    }
    for (int o = 32; o > 0; o >>= 1) ov += __shfl_down(ov, o);
    if ((threadIdx.x & 63) == 0 && ov) atomicAdd(overflowed, ov);
}

// countingbloom.py:235-238 intersection: dst[i] = (a[i] > 0 && b[i] > 0) ? a[i] + b[i] : 0  (overflow clamped + counted)
static __global__ __launch_bounds__(kBlock) void k_cbf_intersect(uint32_t *dst, const uint32_t *a, const uint32_t *b, uint64_t n,
                                                                unsigned long long *overflowed)
{
    unsigned long long ov = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t x = a[i], y = b[i];
        uint64_t t = (x > 0 && y > 0) ? (uint64_t)x + (uint64_t)y : 0;
        if (t > 0xFFFFFFFFULL) { t = 0xFFFFFFFFULL; ++ov; }
        dst[i] = (uint32_t)t;
    }
    for (int o = 32; o > 0; o >>= 1) ov += __shfl_down(ov, o);
    if ((threadIdx.x & 63) == 0 && ov) atomicAdd(overflowed, ov);
}

// countingbloom.py:262-266 jaccard: out[0] += #(a>0 || b>0), out[1] += #(a>0 && b>0)
static __global__ __launch_bounds__(kBlock) void k_cbf_jaccard(const uint32_t *a, const uint32_t *b, uint64_t n,
                                                              unsigned long long *out)
{
    unsigned long long cu = 0, ci = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const bool x = a[i] > 0, y = b[i] > 0;
        cu += x || y;
        ci += x && y;
    }
    for (int o = 32; o > 0; o >>= 1) {
        cu += __shfl_down(cu, o);
        ci += __shfl_down(ci, o);
    }
    __shared__ unsigned long long pu[kBlock / 64], pi[kBlock / 64];
    if ((threadIdx.x & 63) == 0) { pu[threadIdx.x >> 6] = cu; pi[threadIdx.x >> 6] = ci; }
    __syncthreads();
    if (threadIdx.x == 0) {
        cu = 0; ci = 0;
        for (int w = 0; w < kBlock / 64; ++w) { cu += pu[w]; ci += pi[w]; }
        if (cu) atomicAdd(out, cu);
        if (ci) atomicAdd(out + 1, ci);
    }
}

// --------------------------------------------------- synthetic streams
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static __global__ __launch_bounds__(kBlock) void k_gen_keys16(ulonglong2 *dst, uint64_t start, uint64_t n, uint64_t seed)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const uint64_t i = start + j;
        dst[j] = make_ulonglong2(splitmix64(seed + 2 * i), splitmix64(seed + 2 * i + 1));
    }
}

static __global__ __launch_bounds__(kBlock) void k_gen_weights(int32_t *dst, uint64_t start, uint64_t n, uint64_t seed)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride)
        dst[j] = (int32_t)(1 + splitmix64((seed ^ 0xC0FFEEULL) + start + j) % 7);
}

// GUPS-style random-access ceiling: the same access shape as the sketch kernels (one random 4-byte
// atomic / load per probe, 7 independent probes in flight per lane) with the hashing stripped away.
template <int OP>
__global__ __launch_bounds__(kBlock) void k_gups(uint32_t *tab, uint64_t nwords, uint64_t n, uint64_t seed,
                                                 unsigned long long *sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t r = splitmix64(seed + i);
        const uint64_t idx = __umul64hi(r, nwords);  // uniform in [0, nwords)
        if (OP == 0) atomicOr(tab + idx, 1u << (uint32_t)(r & 31));
        else if (OP == 1) atomicAdd(tab + idx, 1u);
        else acc += tab[idx];
    }
    if (OP == 2 && acc == 0x12345678u) atomicAdd(sink, 1ULL);
}

}  // namespace psk
