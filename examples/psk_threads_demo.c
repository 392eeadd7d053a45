/* psk_threads_demo.c -- two host threads, two handles, two HIP streams, ONE device, at the same time (include/psk.h "Threading").
 *
 *   gcc -O2 -std=gnu11 -pthread -D__HIP_PLATFORM_AMD__ examples/psk_threads_demo.c -I include -I /opt/rocm/include \
 *       -L pyprobables_amd/csrc -lpsk_hip -L /opt/rocm/lib -lamdhip64 -o psk_threads_demo
 *
 * Thread A: BloomFilter m = 2^26, k = 7 -- rounds of insert + lookup of 300 000 host-resident 16-byte keys (bloom.py:234-272), its own
 *           per-sketch options (partition_min_keys = 1, bloom_lookup = 3).
 * Thread B: CountingBloomFilter m = 2^25 + 12345, k = 5 -- rounds of add + remove of the first half (countingbloom.py:135-208), other
 *           per-sketch options (partition_min_keys = 65536, remove_exact = 1).
 * Each thread also provokes an error of its own every round and must read ITS message back from psk_last_error().
 * Afterwards the main thread repeats both workloads one after the other on fresh handles: the tables must be byte-identical. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "psk.h"

#define CHECK(call)                                                                       \
    do {                                                                                  \
        int rc__ = (call);                                                                \
        if (rc__ != PSK_OK) {                                                             \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc__, psk_last_error());              \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

enum { ROUNDS = 5, NKEYS = 300000 };

static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static void gen_keys(uint8_t *keys, uint64_t n, uint64_t seed)
{
    for (uint64_t i = 0; i < 2 * n; ++i) {
        const uint64_t w = splitmix64(seed + i);
        memcpy(keys + 8 * i, &w, 8);
    }
}

typedef struct {
    int kind;          /* 0 Bloom, 1 CountingBloomFilter */
    int use_stream;    /* own HIP stream (threads) or the NULL stream (the sequential repeat) */
    uint8_t *table;    /* out: the table's bytes after the last round */
    uint64_t table_bytes;
    int rc;
} Job;

static int run_job(Job *j)
{
    hipStream_t st = NULL;
    if (j->use_stream && hipStreamCreate(&st) != hipSuccess) return 1;
    uint8_t *keys = malloc((size_t)NKEYS * 16), *hits = malloc(NKEYS);
    uint32_t *mins = malloc((size_t)NKEYS * 4);
    psk_sketch *s = NULL;
    if (j->kind == 0) {
        CHECK(psk_bloom_create(1ULL << 26, 7, 0, NULL, &s));
        CHECK(psk_sketch_set_option(s, "partition_min_keys", 1));
        CHECK(psk_sketch_set_option(s, "bloom_lookup", 3));
    } else {
        CHECK(psk_cbf_create((1ULL << 25) + 12345, 5, 0, NULL, &s));
        CHECK(psk_sketch_set_option(s, "partition_min_keys", 65536));
        CHECK(psk_sketch_set_option(s, "remove_exact", 1));
    }
    for (int r = 0; r < ROUNDS; ++r) {
        gen_keys(keys, NKEYS, (j->kind ? 0xC0FFEEULL : 0x5EEDULL) + (uint64_t)r * 10 * NKEYS);
        if (j->kind == 0) {
            CHECK(psk_bloom_add(s, PSK_KEYS_FIXED, keys, NULL, NKEYS, 16, PSK_HOST, st));
            CHECK(psk_bloom_check(s, PSK_KEYS_FIXED, keys, NULL, NKEYS, 16, PSK_HOST, hits, st));
            for (int i = 0; i < NKEYS; ++i)
                if (!hits[i]) { fprintf(stderr, "bloom: inserted key %d not found in round %d\n", i, r); return 1; }
            /* an error of THIS thread's own: a lookup without an output buffer */
            if (psk_bloom_check(s, PSK_KEYS_FIXED, keys, NULL, NKEYS, 16, PSK_HOST, NULL, st) == PSK_OK) { fprintf(stderr, "bloom: NULL out accepted\n"); return 1; }
            if (!strstr(psk_last_error(), "out")) { fprintf(stderr, "bloom thread read a foreign error: %s\n", psk_last_error()); return 1; }
        } else {
            CHECK(psk_cbf_add(s, PSK_KEYS_FIXED, keys, NULL, NKEYS, 16, NULL, PSK_HOST, st));
            CHECK(psk_cbf_remove(s, PSK_KEYS_FIXED, keys, NULL, NKEYS / 2, 16, NULL, PSK_HOST, st));
            CHECK(psk_cbf_check(s, PSK_KEYS_FIXED, keys, NULL, NKEYS, 16, PSK_HOST, mins, st));
            for (int i = NKEYS / 2; i < NKEYS; ++i)
                if (!mins[i]) { fprintf(stderr, "cbf: live key %d has min 0 in round %d\n", i, r); return 1; }
            if (psk_sketch_set_option(s, "no_such_option", 1) == PSK_OK) { fprintf(stderr, "cbf: unknown option accepted\n"); return 1; }
            if (!strstr(psk_last_error(), "no_such_option")) { fprintf(stderr, "cbf thread read a foreign error: %s\n", psk_last_error()); return 1; }
        }
    }
    int64_t v = 0;
    CHECK(psk_sketch_get_option(s, "partition_min_keys", &v));
    if (v != (j->kind ? 65536 : 1)) { fprintf(stderr, "per-sketch option leaked between handles: %lld\n", (long long)v); return 1; }
    uint64_t padded = 0, logical = 0;
    void *ptr = NULL;
    CHECK(psk_table_info(s, &ptr, &padded, &logical));
    j->table_bytes = logical;
    j->table = malloc(logical);
    CHECK(psk_read_table(s, j->table, logical, st));
    CHECK(psk_destroy(s));
    if (st) (void)hipStreamDestroy(st);
    free(keys); free(hits); free(mins);
    return 0;
}

static void *thread_main(void *arg)
{
    Job *j = (Job *)arg;
    j->rc = run_job(j);
    return NULL;
}

int main(void)
{
    int ndev = 0;
    if (psk_device_count(&ndev) != PSK_OK || ndev <= 0) {
        fprintf(stderr, "no HIP device: %s\n", psk_last_error());
        return 2;
    }
    Job conc[2] = {{0, 1, NULL, 0, 0}, {1, 1, NULL, 0, 0}}, seq[2] = {{0, 0, NULL, 0, 0}, {1, 0, NULL, 0, 0}};
    pthread_t t[2];
    for (int i = 0; i < 2; ++i) pthread_create(&t[i], NULL, thread_main, &conc[i]);
    for (int i = 0; i < 2; ++i) pthread_join(t[i], NULL);
    if (conc[0].rc || conc[1].rc) return 1;
    for (int i = 0; i < 2; ++i) {
        if (run_job(&seq[i])) return 1;
        if (seq[i].table_bytes != conc[i].table_bytes || memcmp(seq[i].table, conc[i].table, seq[i].table_bytes)) {
            fprintf(stderr, "%s table differs between the concurrent and the sequential run\n", i ? "cbf" : "bloom");
            return 1;
        }
    }
    printf("PSK THREADS OK: bloom %llu bytes, cbf %llu bytes identical under concurrency\n", (unsigned long long)conc[0].table_bytes,
           (unsigned long long)conc[1].table_bytes);
    return 0;
}
