// psk_index_ops.hip -- kernels over pre-computed Bloom bit indices (uint32 idx[n][k]).
//
// They serve the filters whose insert is conditional on a lookup -- ExpandingBloomFilter / RotatingBloomFilter
// (expandingbloom.py:149-170, :320-331: "add the key to the newest filter unless ANY filter already reports it").
// All filters of such a stack share (m, k, hash), so a batch is hashed once (psk_bloom_indices) and every later
// step -- lookups in each filter, the sequential-semantics resolution, the masked insert -- is index work.
//
// psk_idx_resolve_ordered reproduces, for a whole ordered batch at once, which keys a sequential loop
//     for key in batch:  if key not in table: table.add(key)
// would insert (S = the table before the batch; all other filters are constant and enter as `present`).
// Call a key that is not reported before the batch a candidate, and let
//     first[b] = min position over the candidates that touch bit b, b clear in S        (one atomicMin pass).
// Claim: candidate i is inserted  <=>  first[b] == i for one of its clear bits b.
//   (<=) nobody before i touches b and non-candidates insert nothing, so b is still clear when i runs.
//   (=>) otherwise every clear bit b of i has an earlier candidate j = first[b].  When the loop reached j, either
//        j was inserted (setting b) or j was skipped because all of its bits -- b included -- were already set.
//        Either way b is set before i runs; that holds for every bit of i, so i is reported and skipped.
// So the fate of every key follows from first[] alone: no iteration, no ordered walk.
// first[] must be all-ones on entry and is all-ones again on exit (only touched entries are rewritten).
#include "psk_host.hpp"

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
enum : uint8_t { kSkip = 0, kInsert = 1 };

__device__ __forceinline__ bool bit_set(const uint32_t *tab, uint32_t b) { return (tab[b >> 5] >> (b & 31)) & 1u; }

// out[i] = (accumulate ? out[i] : 0) | all k bits of key i are set      (bloom.py:269-271)
__global__ __launch_bounds__(kBlock) void k_idx_test(const uint32_t *tab, const uint32_t *idx, uint64_t n, uint32_t k,
                                                     uint8_t *out, int accumulate)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        if (accumulate && out[i]) continue;  // already found in an earlier filter
        uint32_t ok = 1;
        for (uint32_t j = 0; j < k; ++j) ok &= (uint32_t)bit_set(tab, idx[i * k + j]);
        out[i] = (uint8_t)ok;
    }
}

// tab |= bits of every key with flag[i] == kInsert (flag == nullptr: every key)      (bloom.py:247-249)
__global__ __launch_bounds__(kBlock) void k_idx_insert(uint32_t *tab, const uint32_t *idx, const uint8_t *flag, uint64_t n,
                                                       uint32_t k)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        if (flag && flag[i] != kInsert) continue;
        for (uint32_t j = 0; j < k; ++j) {
            const uint32_t b = idx[i * k + j];
            atomicOr(tab + (b >> 5), 1u << (b & 31));
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_res_first(const uint32_t *tab, const uint32_t *idx, const uint8_t *present,
                                                      uint64_t n, uint32_t k, uint32_t *first)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        if (present && present[i]) continue;
        for (uint32_t j = 0; j < k; ++j) {
            const uint32_t b = idx[i * k + j];
            if (!bit_set(tab, b)) atomicMin(first + b, (uint32_t)i);
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_res_classify(const uint32_t *tab, const uint32_t *idx, const uint8_t *present,
                                                         uint64_t n, uint32_t k, const uint32_t *first, uint8_t *flag)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint8_t f = kSkip;
        if (!(present && present[i])) {
            for (uint32_t j = 0; j < k; ++j) {
                const uint32_t b = idx[i * k + j];
                if (!bit_set(tab, b) && first[b] == (uint32_t)i) f = kInsert;
            }
        }
        flag[i] = f;
    }
}

// first[b] = all-ones for every bit a candidate touched (entries of bits set in S were never written: harmless)
__global__ __launch_bounds__(kBlock) void k_res_reset(const uint32_t *idx, const uint8_t *present, uint64_t n, uint32_t k,
                                                      uint32_t *first)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        if (present && present[i]) continue;
        for (uint32_t j = 0; j < k; ++j) first[idx[i * k + j]] = kNone;
    }
}

// ---- the same resolution with `first` as a bounded HASH MAP (round 4): a uint32 per bit POSITION is 4 bytes per filter bit -- 1 GiB of
// scratch for a 32 MiB filter of 2^28 bits.  The host resolves an ordered chunk in sub-chunks (each sees the table as the previous one
// left it: the claim above holds for any ordered piece), so the map only has to hold the clear bits ONE sub-chunk touches:
// slots[2^lg] = (bit position, smallest candidate), open addressing, all-ones = empty; reset with one fill.
__device__ __forceinline__ uint32_t slot_of(uint32_t b, uint32_t lg) { return (b * 0x9E3779B1u) >> (32 - lg); }

__global__ __launch_bounds__(kBlock) void k_res_first_h(const uint32_t *tab, const uint32_t *idx, const uint8_t *present, uint64_t n, uint32_t k,
                                                        uint2 *slots, uint32_t lg, uint32_t *full)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint32_t mask = (1u << lg) - 1;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        if (present && present[i]) continue;
        for (uint32_t j = 0; j < k; ++j) {
            const uint32_t b = idx[i * k + j];
            if (bit_set(tab, b)) continue;
            uint32_t s = slot_of(b, lg), tries = 0;
            for (;; s = (s + 1) & mask) {
                const uint32_t old = atomicCAS(&slots[s].x, kNone, b);
                if (old == kNone || old == b) {
                    atomicMin(&slots[s].y, (uint32_t)i);
                    break;
                }
                if (++tries > mask) {  // (cannot happen with the host's sizing; never spin for ever)
                    *full = 1u;
                    break;
                }
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_res_classify_h(const uint32_t *tab, const uint32_t *idx, const uint8_t *present, uint64_t n, uint32_t k,
                                                           const uint2 *slots, uint32_t lg, uint8_t *flag)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint32_t mask = (1u << lg) - 1;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint8_t f = kSkip;
        if (!(present && present[i])) {
            for (uint32_t j = 0; j < k; ++j) {
                const uint32_t b = idx[i * k + j];
                if (bit_set(tab, b)) continue;
                for (uint32_t s = slot_of(b, lg), tries = 0; tries <= mask; s = (s + 1) & mask, ++tries) {
                    const uint2 e = slots[s];
                    if (e.x == b) {
                        if (e.y == (uint32_t)i) f = kInsert;
                        break;
                    }
                    if (e.x == kNone) break;
                }
            }
        }
        flag[i] = f;
    }
}

__global__ __launch_bounds__(kBlock) void k_count_flag(const uint8_t *flag, uint64_t n, uint8_t value, unsigned long long *out)
{
    __shared__ unsigned long long part[kBlock / 64];
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) mine += flag[i] == value;
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (int w = 0; w < kBlock / 64; ++w) s += part[w];
        if (s) atomicAdd(out, s);
    }
}

__global__ __launch_bounds__(kBlock) void k_bytes_or(uint8_t *dst, const uint8_t *src, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) dst[i] |= src[i];
}

unsigned grid_keys(uint64_t n)
{
    const uint64_t g = (n + kBlock - 1) / kBlock;
    return (unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

}  // namespace

extern "C" int psk_idx_test(const void *table_dev, const uint32_t *idx_dev, uint64_t n, uint32_t k, uint8_t *out_dev,
                            int accumulate, int device, void *stream)
{
    if (n && (!table_dev || !idx_dev || !out_dev)) return fail(PSK_EINVAL, "NULL argument");
    if (k == 0) return fail(PSK_EINVAL, "k must be > 0");
    PSK_USE_DEVICE(device);
    if (!n) return PSK_OK;
    hipLaunchKernelGGL(k_idx_test, dim3(grid_keys(n)), dim3(kBlock), 0, (hipStream_t)stream, (const uint32_t *)table_dev, idx_dev, n,
                       k, out_dev, accumulate);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

extern "C" int psk_idx_insert(void *table_dev, const uint32_t *idx_dev, const uint8_t *flag_dev, uint64_t n, uint32_t k,
                              int device, void *stream)
{
    if (n && (!table_dev || !idx_dev)) return fail(PSK_EINVAL, "NULL argument");
    if (k == 0) return fail(PSK_EINVAL, "k must be > 0");
    PSK_USE_DEVICE(device);
    if (!n) return PSK_OK;
    hipLaunchKernelGGL(k_idx_insert, dim3(grid_keys(n)), dim3(kBlock), 0, (hipStream_t)stream, (uint32_t *)table_dev, idx_dev,
                       flag_dev, n, k);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

extern "C" int psk_idx_resolve_ordered(const void *table_dev, const uint32_t *idx_dev, const uint8_t *present_dev, uint64_t n,
                                       uint32_t k, uint32_t *first_dev, uint8_t *flag_dev, uint64_t *count_dev,
                                       uint64_t *inserted_host, int device, void *stream)
{
    if (n && (!table_dev || !idx_dev || !first_dev || !flag_dev || !count_dev)) return fail(PSK_EINVAL, "NULL argument");
    if (k == 0) return fail(PSK_EINVAL, "k must be > 0");
    if (n >= 0xFFFFFFFFULL) return fail(PSK_EINVAL, "at most 2^32-2 keys per ordered batch");
    PSK_USE_DEVICE(device);
    if (inserted_host) *inserted_host = 0;
    if (!n) return PSK_OK;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t *tab = (const uint32_t *)table_dev;
    const dim3 grid(grid_keys(n)), block(kBlock);
    unsigned long long *cnt = (unsigned long long *)count_dev;
    HIP_TRY(hipMemsetAsync(cnt, 0, 8, st));
    hipLaunchKernelGGL(k_res_first, grid, block, 0, st, tab, idx_dev, present_dev, n, k, first_dev);
    hipLaunchKernelGGL(k_res_classify, grid, block, 0, st, tab, idx_dev, present_dev, n, k, (const uint32_t *)first_dev, flag_dev);
    hipLaunchKernelGGL(k_res_reset, grid, block, 0, st, idx_dev, present_dev, n, k, first_dev);
    hipLaunchKernelGGL(k_count_flag, dim3(grid_keys(n) > 1024 ? 1024 : grid_keys(n)), block, 0, st, (const uint8_t *)flag_dev, n,
                       (uint8_t)kInsert, cnt);
    HIP_TRY(hipGetLastError());
    if (inserted_host) {
        unsigned long long host_cnt = 0;
        HIP_TRY(hipMemcpyAsync(&host_cnt, cnt, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        *inserted_host = host_cnt;
    }
    return PSK_OK;
}

extern "C" int psk_idx_resolve_ordered_hashed(const void *table_dev, const uint32_t *idx_dev, const uint8_t *present_dev, uint64_t n, uint32_t k,
                                              void *slots_dev, uint32_t lg_slots, uint8_t *flag_dev, uint64_t *count_dev, int device, void *stream)
{
    if (n && (!table_dev || !idx_dev || !slots_dev || !flag_dev || !count_dev)) return fail(PSK_EINVAL, "NULL argument");
    if (k == 0) return fail(PSK_EINVAL, "k must be > 0");
    if (lg_slots < 4 || lg_slots > 31 || n * (uint64_t)k * 2 > (1ULL << lg_slots)) return fail(PSK_EINVAL, "the slot table must hold at least 2 * n * k entries");
    PSK_USE_DEVICE(device);
    if (!n) return PSK_OK;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t *tab = (const uint32_t *)table_dev;
    const dim3 grid(grid_keys(n)), block(kBlock);
    // count_dev[0] += keys to insert, count_dev[1] = "the table was full" (checked by the caller with its one read-back)
    hipLaunchKernelGGL(k_res_first_h, grid, block, 0, st, tab, idx_dev, present_dev, n, k, (uint2 *)slots_dev, lg_slots, (uint32_t *)(count_dev + 1));
    hipLaunchKernelGGL(k_res_classify_h, grid, block, 0, st, tab, idx_dev, present_dev, n, k, (const uint2 *)slots_dev, lg_slots, flag_dev);
    HIP_TRY(hipMemsetAsync(slots_dev, 0xFF, (size_t)8 << lg_slots, st));  // empty again for the next sub-chunk
    hipLaunchKernelGGL(k_count_flag, dim3(grid_keys(n) > 1024 ? 1024 : grid_keys(n)), block, 0, st, (const uint8_t *)flag_dev, n, (uint8_t)kInsert,
                       (unsigned long long *)count_dev);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}

extern "C" int psk_bytes_or(void *dst_dev, const void *src_dev, uint64_t n, int device, void *stream)
{
    if (n && (!dst_dev || !src_dev)) return fail(PSK_EINVAL, "NULL argument");
    PSK_USE_DEVICE(device);
    if (!n) return PSK_OK;
    hipLaunchKernelGGL(k_bytes_or, dim3(grid_keys(n)), dim3(kBlock), 0, (hipStream_t)stream, (uint8_t *)dst_dev,
                       (const uint8_t *)src_dev, n);
    HIP_TRY(hipGetLastError());
    return PSK_OK;
}
