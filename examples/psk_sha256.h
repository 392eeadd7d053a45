/* psk_sha256.h -- a small SHA-256 (FIPS 180-4) for the example programs: they print the digest of a sketch's table bytes so that a C caller
 * (and tests/test_gpu_c_abi.py) can compare it with the digests the reference produced (tests/golden/golden_cdemo.json).  Host code only. */
#ifndef PSK_SHA256_H
#define PSK_SHA256_H
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t len; } psk_sha256_ctx;

static const uint32_t psk_sha256_k[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
    0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
    0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
    0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
    0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t psk_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void psk_sha256_block(psk_sha256_ctx *c, const uint8_t *p)
{
    uint32_t w[64], s[8];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
        const uint32_t a = psk_rotr(w[i - 15], 7) ^ psk_rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t b = psk_rotr(w[i - 2], 17) ^ psk_rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + a + w[i - 7] + b;
    }
    memcpy(s, c->h, sizeof s);
    for (int i = 0; i < 64; ++i) {
        const uint32_t t1 = s[7] + (psk_rotr(s[4], 6) ^ psk_rotr(s[4], 11) ^ psk_rotr(s[4], 25)) + ((s[4] & s[5]) ^ (~s[4] & s[6])) + psk_sha256_k[i] + w[i];
        const uint32_t t2 = (psk_rotr(s[0], 2) ^ psk_rotr(s[0], 13) ^ psk_rotr(s[0], 22)) + ((s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]));
        s[7] = s[6]; s[6] = s[5]; s[5] = s[4]; s[4] = s[3] + t1; s[3] = s[2]; s[2] = s[1]; s[1] = s[0]; s[0] = t1 + t2;
    }
    for (int i = 0; i < 8; ++i) c->h[i] += s[i];
}

/* sha256 of `n` bytes at `data` -> 64 lowercase hex digits + NUL in `hex` */
static void psk_sha256_hex(const void *data, uint64_t n, char hex[65])
{
    psk_sha256_ctx c = {{0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19}, {0}, 0};
    const uint8_t *p = (const uint8_t *)data;
    uint64_t left = n;
    for (; left >= 64; left -= 64, p += 64) psk_sha256_block(&c, p);
    uint8_t tail[128] = {0};
    memcpy(tail, p, left);
    tail[left] = 0x80;
    const uint64_t total = left + 1 + 8 <= 64 ? 64 : 128, bits = n * 8;
    for (int i = 0; i < 8; ++i) tail[total - 1 - i] = (uint8_t)(bits >> (8 * i));
    psk_sha256_block(&c, tail);
    if (total == 128) psk_sha256_block(&c, tail + 64);
    for (int i = 0; i < 8; ++i) sprintf(hex + 8 * i, "%08x", c.h[i]);
}
#endif
