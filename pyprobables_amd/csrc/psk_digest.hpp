// psk_digest.hpp -- the reference's digest-based hash families on the GPU (hashes.py:17-40, :125-150):
//     default_md5(key, depth) / default_sha256(key, depth):
//         tmp = key bytes (a str is UTF-8 encoded);  for idx in range(depth): tmp = H(tmp).digest();  res += LE64(tmp[:8])
// i.e. a CHAIN of digests -- the first over the key, every later one over the previous digest (16 / 32 bytes, one
// block).  One lane per key; the message schedule lives in registers, rounds fully unrolled.  RFC 1321 / FIPS 180-4.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace psk {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int s) { return (x >> s) | (x << (32 - s)); }
__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// ------------------------------------------------------------------ MD5
__device__ __constant__ const uint32_t kMd5K[64] = {
    0xd76aa478u, 0xe8c7b756u, 0x242070dbu, 0xc1bdceeeu, 0xf57c0fafu, 0x4787c62au, 0xa8304613u, 0xfd469501u,
    0x698098d8u, 0x8b44f7afu, 0xffff5bb1u, 0x895cd7beu, 0x6b901122u, 0xfd987193u, 0xa679438eu, 0x49b40821u,
    0xf61e2562u, 0xc040b340u, 0x265e5a51u, 0xe9b6c7aau, 0xd62f105du, 0x02441453u, 0xd8a1e681u, 0xe7d3fbc8u,
    0x21e1cde6u, 0xc33707d6u, 0xf4d50d87u, 0x455a14edu, 0xa9e3e905u, 0xfcefa3f8u, 0x676f02d9u, 0x8d2a4c8au,
    0xfffa3942u, 0x8771f681u, 0x6d9d6122u, 0xfde5380cu, 0xa4beea44u, 0x4bdecfa9u, 0xf6bb4b60u, 0xbebfbc70u,
    0x289b7ec6u, 0xeaa127fau, 0xd4ef3085u, 0x04881d05u, 0xd9d4d039u, 0xe6db99e5u, 0x1fa27cf8u, 0xc4ac5665u,
    0xf4292244u, 0x432aff97u, 0xab9423a7u, 0xfc93a039u, 0x655b59c3u, 0x8f0ccc92u, 0xffeff47du, 0x85845dd1u,
    0x6fa87e4fu, 0xfe2ce6e0u, 0xa3014314u, 0x4e0811a1u, 0xf7537e82u, 0xbd3af235u, 0x2ad7d2bbu, 0xeb86d391u,
};

struct Md5 {
    static constexpr int kDigestWords = 4;
    static constexpr bool kBigEndian = false;
    uint32_t h[4];
    __device__ __forceinline__ void init() { h[0] = 0x67452301u; h[1] = 0xefcdab89u; h[2] = 0x98badcfeu; h[3] = 0x10325476u; }
    // one 64-byte block, m = 16 little-endian words
    __device__ __forceinline__ void block(const uint32_t (&m)[16])
    {
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            uint32_t f;
            int g, s;
            if (i < 16) {
                f = (b & c) | (~b & d); g = i;
                constexpr int sh[4] = {7, 12, 17, 22}; s = sh[i & 3];
            } else if (i < 32) {
                f = (d & b) | (~d & c); g = (5 * i + 1) & 15;
                constexpr int sh[4] = {5, 9, 14, 20}; s = sh[i & 3];
            } else if (i < 48) {
                f = b ^ c ^ d; g = (3 * i + 5) & 15;
                constexpr int sh[4] = {4, 11, 16, 23}; s = sh[i & 3];
            } else {
                f = c ^ (b | ~d); g = (7 * i) & 15;
                constexpr int sh[4] = {6, 10, 15, 21}; s = sh[i & 3];
            }
            const uint32_t t = a + f + kMd5K[i] + m[g];
            a = d; d = c; c = b;
            b = b + rotl32(t, s);
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d;
    }
    // LE64 of the first 8 digest bytes (the digest is the little-endian image of h[0..3])
    __device__ __forceinline__ uint64_t first8() const { return (uint64_t)h[0] | ((uint64_t)h[1] << 32); }
    // message words of the digest itself, for the next link of the chain
    __device__ __forceinline__ uint32_t digest_word(int w) const { return h[w]; }
};

// ------------------------------------------------------------------ SHA-256
__device__ __constant__ const uint32_t kShaK[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u,
};

struct Sha256 {
    static constexpr int kDigestWords = 8;
    static constexpr bool kBigEndian = true;
    uint32_t h[8];
    __device__ __forceinline__ void init()
    {
        h[0] = 0x6a09e667u; h[1] = 0xbb67ae85u; h[2] = 0x3c6ef372u; h[3] = 0xa54ff53au;
        h[4] = 0x510e527fu; h[5] = 0x9b05688cu; h[6] = 0x1f83d9abu; h[7] = 0x5be0cd19u;
    }
    // one 64-byte block, m = 16 big-endian words
    __device__ __forceinline__ void block(const uint32_t (&m)[16])
    {
        uint32_t w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = m[i];
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (i >= 16) {  // rolling 16-word schedule
                const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
                const uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
                const uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
                w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
            }
            const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
            const uint32_t ch = (e & f) ^ (~e & g);
            const uint32_t t1 = hh + S1 + ch + kShaK[i] + w[i & 15];
            const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
            const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
            const uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    // the digest is the big-endian image of h[0..7]; LE64 of its first 8 bytes:
    __device__ __forceinline__ uint64_t first8() const { return (uint64_t)bswap32(h[0]) | ((uint64_t)bswap32(h[1]) << 32); }
    __device__ __forceinline__ uint32_t digest_word(int w) const { return h[w]; }
};

// digest of `len` bytes at p (any alignment): words are assembled byte by byte, padding per RFC 1321 / FIPS 180-4
template <class H>
__device__ __forceinline__ void digest_bytes(H &st, const uint8_t *p, uint64_t len)
{
    st.init();
    const uint64_t nblocks = (len + 8) / 64 + 1;
    const uint64_t bits = len * 8;
    for (uint64_t blk = 0; blk < nblocks; ++blk) {
        uint32_t m[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            uint32_t word = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint64_t pos = blk * 64 + (uint64_t)w * 4 + c;
                const uint32_t byte = pos < len ? (uint32_t)p[pos] : (pos == len ? 0x80u : 0u);
                word |= H::kBigEndian ? byte << (24 - 8 * c) : byte << (8 * c);
            }
            m[w] = word;
        }
        if (blk == nblocks - 1) {  // message length in bits closes the last block
            if (H::kBigEndian) { m[14] = (uint32_t)(bits >> 32); m[15] = (uint32_t)bits; }
            else { m[14] = (uint32_t)bits; m[15] = (uint32_t)(bits >> 32); }
        }
        st.block(m);
    }
}

// next link: digest of the previous digest (16 or 32 bytes -> a single block)
template <class H>
__device__ __forceinline__ void digest_of_digest(H &st)
{
    uint32_t m[16];
#pragma unroll
    for (int w = 0; w < 16; ++w) m[w] = w < H::kDigestWords ? st.digest_word(w) : 0u;
    m[H::kDigestWords] = H::kBigEndian ? 0x80000000u : 0x80u;
    const uint32_t bits = 32u * H::kDigestWords;
    if (H::kBigEndian) m[15] = bits;
    else m[14] = bits;
    st.init();
    st.block(m);
}

// out[i*depth + d] = LE64 of the first 8 bytes of the (d+1)-th digest of the chain started at key i
// fixed: key i = data[i*key_len .. +key_len);  else key i = data[offs[i] .. offs[i+1])
template <class H>
__global__ __launch_bounds__(256) void k_digest_chain(const uint8_t *data, const uint64_t *offs, uint32_t key_len, uint64_t n,
                                                      uint32_t depth, uint64_t *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint64_t lo = offs ? offs[i] : i * (uint64_t)key_len;
        const uint64_t len = offs ? offs[i + 1] - lo : (uint64_t)key_len;
        H st;
        digest_bytes(st, data + lo, len);
        for (uint32_t d = 0; d < depth; ++d) {
            if (d) digest_of_digest(st);
            out[i * depth + d] = st.first8();
        }
    }
}

}  // namespace psk
