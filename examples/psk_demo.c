/* psk_demo.c -- the C ABI of libpsk_hip.so used from plain C: no Python, no torch.
 *
 *   hipcc (or gcc) examples/psk_demo.c -I include -L pyprobables_amd/csrc -lpsk_hip -L /opt/rocm/lib -lamdhip64 -o psk_demo
 *
 * Builds a BloomFilter(est_elements=100000, fpr=0.01)-sized filter (m = 958506 bits, k = 7: bloom.py:463-502), inserts
 * 50 000 synthetic 16-byte keys from a HOST buffer, looks all 100 000 up, and cross-checks three things a C caller of
 * the reference's author's C library would expect: no false negatives, a false-positive rate near the design point, and
 * the known 64-bit FNV-1a of "this is a test" (tests/hashes_test.py:29). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "psk.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc__ = (call);                                                 \
        if (rc__ != PSK_OK) {                                              \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc__, psk_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

int main(void)
{
    int ndev = 0;
    if (psk_device_count(&ndev) != PSK_OK || ndev <= 0) {
        fprintf(stderr, "no HIP device: %s\n", psk_last_error());
        return 2;
    }
    const uint64_t n = 100000, m_bits = 958506;
    const uint32_t k = 7;
    uint8_t *keys = malloc(n * 16), *hits = malloc(n);
    for (uint64_t i = 0; i < 2 * n; ++i) {
        const uint64_t w = splitmix64(0x5EED + i);
        memcpy(keys + 8 * i, &w, 8);
    }
    psk_sketch *blm = NULL;
    CHECK(psk_bloom_create(m_bits, k, 0, NULL, &blm));
    CHECK(psk_bloom_add(blm, PSK_KEYS_FIXED, keys, NULL, n / 2, 16, PSK_HOST, NULL));
    CHECK(psk_bloom_check(blm, PSK_KEYS_FIXED, keys, NULL, n, 16, PSK_HOST, hits, NULL));
    uint64_t found = 0, false_pos = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (i < n / 2) found += hits[i];
        else false_pos += hits[i];
    }
    uint64_t h[2];
    CHECK(psk_fnv1a_hash(PSK_KEYS_FIXED, "this is a test", NULL, 1, 14, 2, PSK_HOST, h, 0, NULL));
    CHECK(psk_destroy(blm));
    printf("inserted %llu, found %llu, false positives %llu of %llu (%.4f), fnv1a=%llu\n", (unsigned long long)(n / 2),
           (unsigned long long)found, (unsigned long long)false_pos, (unsigned long long)(n / 2), (double)false_pos / (double)(n / 2),
           (unsigned long long)h[0]);
    if (found != n / 2) return 3;                                   /* a Bloom filter has no false negatives */
    if (false_pos > n / 2 / 20) return 4;                           /* half full: far below the 1 % design point x 5 */
    if (h[0] != 4040040117721899264ULL) return 5;                   /* hashes_test.py:29 */
    free(keys);
    free(hits);
    puts("PSK C ABI OK");
    return 0;
}
