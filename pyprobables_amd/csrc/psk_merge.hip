// psk_merge.hip -- the multi-GPU merge behind the C ABI (SURVEY.md 8b "psk_merge_allreduce", 8e).
//
// One process (or thread) per GPU holds a full-size replica fed with its own range of the key stream; ONE collective
// then makes every replica the table a single sketch fed the whole stream would hold:
//   Bloom     allreduce(OR)  -- RCCL has no bitwise-OR reduction (rccl.h ncclRedOp_t: sum / prod / max / min / avg), so it
//             is composed: every rank sends bit-range slice j straight to its owner j (grouped ncclSend / ncclRecv: all
//             xGMI links busy at once, no ring), k_or_reduce ORs the R received slices, ncclAllGather returns the result.
//             Reference semantics: bloom.py:401-428 (union is a bytewise OR).
//   CMS / CBF allreduce(SUM) on the counters, 32-bit while the summed per-rank bounds prove that no counter can reach a
//             rail, else widened to 64 bits and clamped the way join does (countminsketch.py:380-391; CBF: 2^32-1,
//             countingbloom.py:149-151).
// RCCL is not a link-time dependency of libpsk_hip.so (like the HIP runtime, see _native.py): the entry points are looked
// up in the RCCL the host process already carries (the communicator handed in belongs to that library).
#include "psk_host.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enumerators only

namespace {

struct Rccl {
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;

template <class F>
bool sym(void *lib, const char *name, F *out)
{
    *out = reinterpret_cast<F>(dlsym(lib, name));
    return *out != nullptr;
}

bool bind_from(void *lib, Rccl *r)
{
    return sym(lib, "ncclCommCount", &r->CommCount) && sym(lib, "ncclCommUserRank", &r->CommUserRank) &&
           sym(lib, "ncclGroupStart", &r->GroupStart) && sym(lib, "ncclGroupEnd", &r->GroupEnd) && sym(lib, "ncclSend", &r->Send) &&
           sym(lib, "ncclRecv", &r->Recv) && sym(lib, "ncclAllGather", &r->AllGather) && sym(lib, "ncclAllReduce", &r->AllReduce) &&
           sym(lib, "ncclGetErrorString", &r->GetErrorString);
}

int rccl(Rccl **out)
{
    if (!g_rccl.ok) {
        Rccl r;
        // 1. symbols the process already exports globally (a C/C++ host linked with -lrccl; python: ctypes RTLD_GLOBAL)
        bool ok = bind_from(RTLD_DEFAULT, &r);
        // 2. an RCCL that is loaded but not global (torch's bundled copy), 3. the system one
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (int pass = 0; !ok && pass < 2; ++pass)
            for (const char *n : names) {
                void *lib = dlopen(n, RTLD_LAZY | (pass == 0 ? RTLD_NOLOAD : 0));
                if (lib && (ok = bind_from(lib, &r))) break;
            }
        if (!ok) {
            const char *why = dlerror();
            return fail(PSK_ENODEV, "RCCL (librccl.so) is not available in this process: %s", why ? why : "symbols missing");
        }
        r.ok = true;
        g_rccl = r;
    }
    *out = &g_rccl;
    return PSK_OK;
}

#define NCCL_TRY(R, expr)                                                                                   \
    do {                                                                                                    \
        ncclResult_t e__ = (expr);                                                                          \
        if (e__ != ncclSuccess) return fail(PSK_EHIP, "%s failed: %s", #expr, (R)->GetErrorString(e__));      \
    } while (0)

// int32 / uint32 counters -> int64 (sign- or zero-extended), and back with the reference's clamp
__global__ __launch_bounds__(kBlock) void k_widen(const uint32_t *src, long long *dst, uint64_t n, int is_signed)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        dst[i] = is_signed ? (long long)(int32_t)src[i] : (long long)src[i];
}

__global__ __launch_bounds__(kBlock) void k_narrow_clamp(const long long *src, uint32_t *dst, uint64_t n, int is_signed, unsigned long long *sat_ctr)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    unsigned long long sat = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        long long v = src[i];
        if (is_signed) {  // countminsketch.py:386-391
            if (v > INT32_MAX) { v = INT32_MAX; ++sat; }
            if (v < INT32_MIN) { v = INT32_MIN; ++sat; }
            dst[i] = (uint32_t)(int32_t)v;
        } else {          // countingbloom.py:149-151
            if (v > 0xFFFFFFFFLL) { v = 0xFFFFFFFFLL; ++sat; }
            dst[i] = (uint32_t)v;
        }
    }
    for (int o = 32; o > 0; o >>= 1) sat += __shfl_down(sat, o);
    if ((threadIdx.x & 63) == 0 && sat) atomicAdd(sat_ctr, sat);
}

// this rank's bound on |counter|, capped so that the SUM over any number of ranks cannot wrap 64 bits (the engine saturates
// the bound at 2^62: four such ranks sum to 0 mod 2^64 and the plain 32-bit reduction would be chosen)
__global__ void k_capped_bound(const long long *ctr_bound, long long *out)
{
    const long long b = *ctr_bound;
    *out = (b < 0 || b > (1LL << 40)) ? (1LL << 40) : b;
}

int launch_grid(uint64_t n)
{
    uint64_t g = (n + kBlock - 1) / kBlock;
    return (int)(g > 4096 ? 4096 : (g ? g : 1));
}

}  // namespace

extern PSK_HIDDEN int64_t g_merge_single_rank;
int64_t g_merge_single_rank = 0;  // psk_set_option("merge_single_rank", 1): run the collective path even with one rank (tests)

extern "C" int psk_merge_or(psk_sketch *s, void *nccl_comm, void *stream)
{
    if (!s) return fail(PSK_EINVAL, "sketch handle is NULL");
    if (s->kind != PSK_KIND_BLOOM) return fail(PSK_EINVAL, "psk_merge_or merges Bloom filters (counters: psk_merge_sum)");
    if (!nccl_comm) return fail(PSK_EINVAL, "communicator is NULL");
    PSK_USE_DEVICE(s->device);
    ++s->table_version;  // the merge rewrites the table (psk_sketch::shadow is stale from here on)
    Rccl *R;
    PSK_TRY(rccl(&R));
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    hipStream_t st = (hipStream_t)stream;
    int nranks = 0;
    NCCL_TRY(R, R->CommCount(comm, &nranks));
    if (nranks < 1) return fail(PSK_EINVAL, "communicator reports %d ranks", nranks);
    if (nranks == 1 && !g_merge_single_rank) return PSK_OK;
    const uint64_t words = s->padded_bytes / 4;
    uint64_t slice_words = (words + (uint64_t)nranks - 1) / (uint64_t)nranks;
    slice_words = (slice_words + 3) & ~3ULL;  // 16-byte slices for the uint4 reduce kernel
    const uint64_t padded = slice_words * (uint64_t)nranks;
    const bool staged = padded != words;      // table not a multiple of 16 * nranks bytes: exchange a zero-padded copy
    PSK_TRY(ensure(s->s_merge, (padded + slice_words + (staged ? padded : 0)) * 4));
    uint32_t *recv = (uint32_t *)s->s_merge.p, *mine = recv + padded;
    uint32_t *work = staged ? mine + slice_words : (uint32_t *)s->table;
    if (staged) {
        HIP_TRY(hipMemsetAsync(work + words, 0, (padded - words) * 4, st));
        HIP_TRY(hipMemcpyAsync(work, s->table, words * 4, hipMemcpyDeviceToDevice, st));
    }
    NCCL_TRY(R, R->GroupStart());  // slice j of every rank lands on rank j
    for (int r = 0; r < nranks; ++r) {
        NCCL_TRY(R, R->Send(work + (uint64_t)r * slice_words, slice_words, ncclInt32, r, comm, st));
        NCCL_TRY(R, R->Recv(recv + (uint64_t)r * slice_words, slice_words, ncclInt32, r, comm, st));
    }
    NCCL_TRY(R, R->GroupEnd());
    PSK_TRY(psk_or_reduce_slices(mine, recv, (uint32_t)nranks, slice_words, s->device, stream));
    NCCL_TRY(R, R->AllGather(mine, work, slice_words, ncclInt32, comm, st));
    if (staged) HIP_TRY(hipMemcpyAsync(s->table, work, words * 4, hipMemcpyDeviceToDevice, st));
    return PSK_OK;
}

extern "C" int psk_merge_sum(psk_sketch *s, void *nccl_comm, void *stream)
{
    if (!s) return fail(PSK_EINVAL, "sketch handle is NULL");
    if (s->kind == PSK_KIND_BLOOM) return fail(PSK_EINVAL, "psk_merge_sum merges counter tables (Bloom: psk_merge_or)");
    if (!nccl_comm) return fail(PSK_EINVAL, "communicator is NULL");
    PSK_USE_DEVICE(s->device);
    ++s->table_version;  // the merge rewrites the table (psk_sketch::shadow is stale from here on)
    Rccl *R;
    PSK_TRY(rccl(&R));
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    hipStream_t st = (hipStream_t)stream;
    int nranks = 0;
    NCCL_TRY(R, R->CommCount(comm, &nranks));
    if (nranks < 1) return fail(PSK_EINVAL, "communicator reports %d ranks", nranks);
    if (nranks == 1 && !g_merge_single_rank) return PSK_OK;
    const bool is_signed = s->kind == PSK_KIND_CMS;
    const uint64_t cells = s->logical_bytes / 4;
    PSK_TRY(flush_combined(s, st));  // write-combined CBF updates belong to this replica's table
    // 1. the ranks agree on the SUM of their bounds on |counter| (a 32-bit SUM wraps silently): one 8-byte read-back
    PSK_TRY(ensure(s->s_aux, 16));
    long long *bsum = (long long *)s->s_aux.p;
    hipLaunchKernelGGL(k_capped_bound, dim3(1), dim3(1), 0, st, (const long long *)(s->ctr + PSK_CTR_ABS_BOUND), bsum);
    HIP_TRY(hipGetLastError());
    NCCL_TRY(R, R->AllReduce(bsum, bsum, 1, ncclInt64, ncclSum, comm, st));
    long long bound = 0;
    HIP_TRY(hipMemcpyAsync(&bound, bsum, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const long long rail = is_signed ? (long long)INT32_MAX : 0xFFFFFFFFLL;
    // 2. the tallies (added / removed / violations / saturated) become global totals: psk_get_counters then reports the
    //    merged sketch's elements_added terms on every rank.  BEFORE the clamp below adds its count: every rank clamps the
    //    same summed table, so that count is already global (summing it would report nranks x the true number).
    //    The merge is ONE-SHOT per stream segment: tables and tallies are per-rank deltas going in and global totals coming
    //    out; merging the same replicas twice would count everything again.
    NCCL_TRY(R, R->AllReduce(s->ctr, s->ctr, 4, ncclInt64, ncclSum, comm, st));
    if (bound >= 0 && bound <= rail) {
        // 3a. no global counter can reach a rail: the plain 32-bit reduction is exact
        NCCL_TRY(R, R->AllReduce(s->table, s->table, cells, is_signed ? ncclInt32 : ncclUint32, ncclSum, comm, st));
    } else {
        // 3b. widen, sum in 64 bits, clamp like join (countminsketch.py:380-391 / countingbloom.py:149-151)
        PSK_TRY(ensure(s->s_merge, cells * 8));
        long long *wide = (long long *)s->s_merge.p;
        hipLaunchKernelGGL(k_widen, dim3(launch_grid(cells)), dim3(kBlock), 0, st, (const uint32_t *)s->table, wide, cells, (int)is_signed);
        HIP_TRY(hipGetLastError());
        NCCL_TRY(R, R->AllReduce(wide, wide, cells, ncclInt64, ncclSum, comm, st));
        hipLaunchKernelGGL(k_narrow_clamp, dim3(launch_grid(cells)), dim3(kBlock), 0, st, (const long long *)wide, (uint32_t *)s->table, cells,
                           (int)is_signed, (unsigned long long *)(s->ctr + PSK_CTR_SATURATED));
        HIP_TRY(hipGetLastError());
    }
    // 4. the wrap-free bound describes the table: re-derive it from the merged counters
    return psk_rescan_bound(s, stream);
}
