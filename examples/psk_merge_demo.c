/* psk_merge_demo.c -- the multi-GPU merge of the C ABI from plain C: one THREAD per GPU of this node, one full-size
 * replica per GPU, ONE collective per structure (psk_merge_or for the Bloom filter, psk_merge_sum for the Count-Min
 * sketch), then every replica is checked against the stream as a whole.  No Python, no torch.
 *
 *   gcc -O2 -pthread examples/psk_merge_demo.c -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       -L pyprobables_amd/csrc -lpsk_hip -L /opt/rocm/lib -lamdhip64 -lrccl -o psk_merge_demo
 *
 * Rank r inserts keys [r*N/R, (r+1)*N/R) of the synthetic stream into its own replica -- no communication
 * -- and after the merge looks up ALL N keys: a key inserted on another GPU is found only through the merged table
 * (bloom.py:401-428: the union is a bytewise OR).  Count-Min: every rank adds weight 1 per key; after psk_merge_sum every
 * key's estimate is >= 1 on every rank and psk_get_counters reports elements_added == N (countminsketch.py:380-391).
 * With a single GPU the communicator has one rank; the option "merge_single_rank" still drives the whole collective
 * path (ncclSend / ncclRecv to self, the OR-reduce kernel, ncclAllGather).
 * Wrap-and-clamp: a third, small sketch whose per-rank bounds on |counter| sum past the int32 rail takes the widened
 * (64-bit) reduction: every rank holds 2^30 in the three bins of one key (add, remove, add: the bound is 3 * 2^30), so
 * with R >= 2 ranks the summed bins clamp at INT32_MAX exactly like join (countminsketch.py:386-391) and the merge's
 * saturation tally is `depth` on EVERY rank -- not R x depth.
 * The merge entry points are called from one thread per rank and never inside an outer ncclGroupStart / ncclGroupEnd. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "psk.h"

#define N_KEYS 4000000ULL
#define M_BITS 268435456ULL /* BloomFilter(28005615, 0.01): m = 2^28, k = 7 */
#define MAX_DEV 16

typedef struct {
    int rank, nranks;
    ncclComm_t comm;
    int ok;
    char msg[512];
} job;

static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define TRY(call)                                                                                  \
    do {                                                                                           \
        int rc__ = (call);                                                                         \
        if (rc__ != PSK_OK) {                                                                      \
            snprintf(j->msg, sizeof j->msg, "%s -> %d: %s", #call, rc__, psk_last_error());        \
            return NULL;                                                                           \
        }                                                                                          \
    } while (0)

static void *rank_main(void *arg)
{
    job *j = (job *)arg;
    const uint64_t lo = N_KEYS * (uint64_t)j->rank / (uint64_t)j->nranks, hi = N_KEYS * (uint64_t)(j->rank + 1) / (uint64_t)j->nranks;
    if (hipSetDevice(j->rank) != hipSuccess) { snprintf(j->msg, sizeof j->msg, "hipSetDevice(%d) failed", j->rank); return NULL; }
    uint8_t *keys = (uint8_t *)malloc(N_KEYS * 16), *hits = (uint8_t *)malloc(N_KEYS);
    int32_t *est = (int32_t *)malloc(N_KEYS * 4);
    for (uint64_t i = 0; i < 2 * N_KEYS; ++i) {
        const uint64_t w = splitmix64(0x5EED + i);
        memcpy(keys + 8 * i, &w, 8);
    }
    psk_sketch *blm = NULL, *cms = NULL;
    TRY(psk_bloom_create(M_BITS, 7, j->rank, NULL, &blm));
    TRY(psk_cms_create(1u << 20, 5, j->rank, NULL, &cms));
    /* the rank's own key range, no communication */
    TRY(psk_bloom_add(blm, PSK_KEYS_FIXED, keys + 16 * lo, NULL, hi - lo, 16, PSK_HOST, NULL));
    TRY(psk_cms_add(cms, PSK_KEYS_FIXED, keys + 16 * lo, NULL, hi - lo, 16, NULL, PSK_HOST, NULL));
    /* ONE collective per structure */
    TRY(psk_merge_or(blm, j->comm, NULL));
    TRY(psk_merge_sum(cms, j->comm, NULL));
    TRY(psk_synchronize(blm, NULL));
    /* every replica now answers for the WHOLE stream */
    TRY(psk_bloom_check(blm, PSK_KEYS_FIXED, keys, NULL, N_KEYS, 16, PSK_HOST, hits, NULL));
    TRY(psk_cms_check(cms, PSK_KEYS_FIXED, keys, NULL, N_KEYS, 16, PSK_HOST, PSK_Q_MIN, est, NULL));
    uint64_t found = 0, counted = 0;
    for (uint64_t i = 0; i < N_KEYS; ++i) {
        found += hits[i];
        counted += est[i] >= 1;
    }
    int64_t ctr[PSK_CTR_COUNT];
    TRY(psk_get_counters(cms, ctr, NULL));
    /* wrap-and-clamp: bins of key 0 hold 2^30 on every rank, the bound on |counter| is 3 * 2^30 > INT32_MAX */
    psk_sketch *big = NULL;
    const int32_t w30 = 1 << 30;
    int32_t merged = 0;
    int64_t bctr[PSK_CTR_COUNT];
    TRY(psk_cms_create(4096, 3, j->rank, NULL, &big));
    TRY(psk_cms_add(big, PSK_KEYS_FIXED, keys, NULL, 1, 16, &w30, PSK_HOST, NULL));
    TRY(psk_cms_remove(big, PSK_KEYS_FIXED, keys, NULL, 1, 16, &w30, PSK_HOST, NULL));
    TRY(psk_cms_add(big, PSK_KEYS_FIXED, keys, NULL, 1, 16, &w30, PSK_HOST, NULL));
    TRY(psk_merge_sum(big, j->comm, NULL));
    TRY(psk_cms_check(big, PSK_KEYS_FIXED, keys, NULL, 1, 16, PSK_HOST, PSK_Q_MIN, &merged, NULL));
    TRY(psk_get_counters(big, bctr, NULL));
    const int64_t sum30 = (int64_t)j->nranks << 30;
    const int32_t want = sum30 > 2147483647LL ? 2147483647 : (int32_t)sum30;
    const int64_t want_sat = sum30 > 2147483647LL ? 3 : 0;  /* one clamped cell per row, counted once */
    const int clamp_ok = merged == want && bctr[PSK_CTR_SATURATED] == want_sat && bctr[PSK_CTR_ADDED] == 2 * sum30 &&
                         bctr[PSK_CTR_REMOVED] == sum30;
    TRY(psk_destroy(big));
    TRY(psk_destroy(blm));
    TRY(psk_destroy(cms));
    free(keys); free(hits); free(est);
    j->ok = found == N_KEYS && counted == N_KEYS && ctr[PSK_CTR_ADDED] == (int64_t)N_KEYS && clamp_ok;
    snprintf(j->msg, sizeof j->msg, "rank %d/%d: inserted [%llu, %llu), found %llu of %llu, cms >= 1 for %llu, elements_added %lld; "
             "clamped merge: bin %d (want %d), saturated %lld (want %lld)",
             j->rank, j->nranks, (unsigned long long)lo, (unsigned long long)hi, (unsigned long long)found, (unsigned long long)N_KEYS,
             (unsigned long long)counted, (long long)ctr[PSK_CTR_ADDED], merged, want, (long long)bctr[PSK_CTR_SATURATED], (long long)want_sat);
    return NULL;
}

int main(void)
{
    int ndev = 0;
    if (psk_device_count(&ndev) != PSK_OK || ndev <= 0) {
        fprintf(stderr, "no HIP device: %s\n", psk_last_error());
        return 2;
    }
    if (ndev > MAX_DEV) ndev = MAX_DEV;
    ncclComm_t comms[MAX_DEV];
    if (ncclCommInitAll(comms, ndev, NULL) != ncclSuccess) { /* one communicator per device of this node */
        fprintf(stderr, "ncclCommInitAll failed\n");
        return 2;
    }
    if (ndev == 1) psk_set_option("merge_single_rank", 1); /* still drive the collective path */
    job jobs[MAX_DEV];
    pthread_t tid[MAX_DEV];
    for (int r = 0; r < ndev; ++r) {
        jobs[r] = (job){r, ndev, comms[r], 0, ""};
        pthread_create(&tid[r], NULL, rank_main, &jobs[r]);
    }
    int ok = 1;
    for (int r = 0; r < ndev; ++r) {
        pthread_join(tid[r], NULL);
        printf("%s\n", jobs[r].msg);
        ok &= jobs[r].ok;
    }
    for (int r = 0; r < ndev; ++r) ncclCommDestroy(comms[r]);
    if (!ok) { printf("PSK MERGE FAILED\n"); return 1; }
    printf("PSK MERGE OK (%d rank%s)\n", ndev, ndev == 1 ? "" : "s");
    return 0;
}
