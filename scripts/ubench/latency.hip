// Round-trip latency floor of a value-returning single-key call (NOTES.md 3.5): one launch, the answer polled from pinned memory.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/latency.hip -o ab/latency && ab/latency
// variants: the kernel posts only / reads 16 bytes from pinned memory first / takes the 16 bytes as kernel arguments / + one dependent HBM load;
// and the same launches ended by hipStreamSynchronize instead of the poll.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

struct Key16 { uint32_t w[4]; };

__global__ void k_post(uint32_t *mbox, uint32_t seq) { __hip_atomic_store(mbox, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_read_pinned(const uint4 *key, uint32_t *out, uint32_t *mbox, uint32_t seq)
{
    const uint4 k = *key;
    out[0] = k.x ^ k.y ^ k.z ^ k.w;
    __threadfence_system();
    __hip_atomic_store(mbox, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_inline(Key16 k, uint32_t *out, uint32_t *mbox, uint32_t seq)
{
    out[0] = k.w[0] ^ k.w[1] ^ k.w[2] ^ k.w[3];
    __threadfence_system();
    __hip_atomic_store(mbox, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_inline_table(Key16 k, const uint32_t *tab, uint32_t mask, uint32_t *out, uint32_t *mbox, uint32_t seq)
{
    const uint32_t h = (k.w[0] ^ k.w[1] ^ k.w[2] ^ k.w[3]) * 2654435761u;
    out[0] = tab[h & mask] + tab[(h >> 7) & mask] + tab[(h >> 13) & mask];
    __threadfence_system();
    __hip_atomic_store(mbox, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <class F>
static double loop(F f, int n = 20000)
{
    for (int i = 0; i < 2000; ++i) f();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main()
{
    uint32_t *mbox, *out, *tab;
    uint4 *key;
    hipHostMalloc((void **)&mbox, 64, hipHostMallocDefault);
    hipHostMalloc((void **)&out, 64, hipHostMallocDefault);
    hipHostMalloc((void **)&key, 64, hipHostMallocDefault);
    hipMalloc((void **)&tab, 32u << 20);
    hipMemset(tab, 1, 32u << 20);
    *mbox = 0;
    hipStream_t st;
    hipStreamCreate(&st);
    uint32_t seq = 0;
    volatile uint32_t *vm = mbox;
    auto wait = [&]() { while (__atomic_load_n(vm, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause(); };
    Key16 kk{{1, 2, 3, 4}};
    printf("post only, polled                 %6.2f us\n", loop([&] { ++seq; hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, st, mbox, seq); wait(); }));
    printf("post only, hipStreamSynchronize   %6.2f us\n", loop([&] { ++seq; hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, st, mbox, seq); hipStreamSynchronize(st); }));
    printf("key read from pinned, polled      %6.2f us\n", loop([&] { ++seq; key->x = seq; hipLaunchKernelGGL(k_read_pinned, dim3(1), dim3(64), 0, st, key, out, mbox, seq); wait(); }));
    printf("key in the kernel args, polled    %6.2f us\n", loop([&] { ++seq; kk.w[0] = seq; hipLaunchKernelGGL(k_inline, dim3(1), dim3(64), 0, st, kk, out, mbox, seq); wait(); }));
    printf("  ... + 3 table loads             %6.2f us\n", loop([&] { ++seq; kk.w[0] = seq; hipLaunchKernelGGL(k_inline_table, dim3(1), dim3(64), 0, st, kk, tab, (8u << 20) - 1, out, mbox, seq); wait(); }));
    printf("key in the kernel args, 256 thr   %6.2f us\n", loop([&] { ++seq; kk.w[0] = seq; hipLaunchKernelGGL(k_inline, dim3(1), dim3(256), 0, st, kk, out, mbox, seq); wait(); }));
    printf("launch only (no wait, 1 in 64 synced) %6.2f us\n", loop([&] { ++seq; hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, st, mbox, seq); if ((seq & 63) == 0) hipStreamSynchronize(st); }));
    hipStreamSynchronize(st);
    return 0;
}
