// micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the integer ops the FNV chain can be built from
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 4096
template <int OP>
__global__ void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 7 + i;
    uint32_t e = seed * 3 + 1;
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 2) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 3) asm volatile("v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "v"(e));
            if (OP == 4) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(e));
            if (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 7) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 8) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
            if (OP == 9) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 10) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x78" : "+v"(a[i]) : "v"(e));
            if (OP == 11) asm volatile("v_lshlrev_b32 %0, 8, %0" : "+v"(a[i]));
            if (OP == 12) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 13) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(e));
            if (OP == 15) asm volatile("v_cmp_ne_u32 vcc, %0, %1" : : "v"(a[i]), "v"(e) : "vcc");
            if (OP == 16) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 17) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 18) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[i]));
            if (OP == 19) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
            if (OP == 20) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 21) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 22) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 23) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 24) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 25) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(e) : "vcc");
            if (OP == 26) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(e) : "vcc");
            if (OP == 27) asm volatile("v_xor_b32 %0, 0x1b3, %0" : "+v"(a[i]));
            if (OP == 28) asm volatile("v_mul_lo_u32 %0, %0, s4" : "+v"(a[i]) : : "s4");
            if (OP == 29) asm volatile("v_and_b32 %0, 0xff, %0" : "+v"(a[i]));
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
__global__ void k64(uint64_t *out, uint32_t seed)
{
    uint64_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 7 + i;
    uint32_t e = seed * 3 + 1;
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(e), "v"((uint32_t)a[i]) : "vcc");
            if (OP == 1) asm volatile("v_lshl_add_u64 %0, %0, 3, %0" : "+v"(a[i]));
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- composite 64-bit FNV-1a steps (hval ^= e; hval *= 2^40 + 0x1B3), 8 independent chains per lane as in the kernels:
//   V 0  the engine's step (psk_device.hpp fnv_step): v_xor, v_mad_u64_u32 (x * 0x1B3 -> lo' and carry), v_lshl_add_u32, v_mad_u64_u32 (H * 0x1B3 + addend)
//   V 1  re-ordered: v_xor, v_mul_lo_u32 (H * 0x1B3), v_lshl_add_u32 ((x << 8) + that), v_mad_u64_u32 (x * 0x1B3 + {0, addend}) -> lo' and H' at once
//   V 2  split state: low chain v_xor + v_mul_lo_u32; high word H' = H * 0x1B3 + mulhi(x, 0x1B3) + (x << 8) as
//        v_mul_hi_u32, v_mul_lo_u32, v_lshl_add_u32, v_add_u32
//   V 3  the 32-bit chain of power-of-two tables (v_xor + v_mul_lo_u32), for scale
template <int V>
__global__ void kstep(uint64_t *out, uint32_t seed)
{
    uint32_t lo[8], hi[8];
    for (int i = 0; i < 8; ++i) { lo[i] = seed + threadIdx.x * 7 + i; hi[i] = seed * 5 + threadIdx.x + i; }
    uint32_t e = (seed * 3 + 1) & 255u;
    const uint32_t P = 0x1B3u;
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (V == 0) {
                uint32_t x, a; uint64_t t, u;
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(lo[i]), "v"(e));
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"(x), "s"(P) : "vcc");
                asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(a) : "v"(x), "v"((uint32_t)(t >> 32)));
                uint64_t add = a;  // (only the low word of the sum is used)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(u) : "v"(hi[i]), "s"(P), "v"(add) : "vcc");
                lo[i] = (uint32_t)t; hi[i] = (uint32_t)u;
            } else if (V == 1) {
                uint32_t x, a, b2; uint64_t u;
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(lo[i]), "v"(e));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a) : "v"(hi[i]), "s"(P));
                asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(b2) : "v"(x), "v"(a));
                uint64_t add = (uint64_t)b2 << 32;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(u) : "v"(x), "s"(P), "v"(add) : "vcc");
                lo[i] = (uint32_t)u; hi[i] = (uint32_t)(u >> 32);
            } else if (V == 2) {
                uint32_t x, l2, mh, a, b2, h2;
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(lo[i]), "v"(e));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(l2) : "v"(x), "s"(P));
                asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(mh) : "v"(x), "s"(P));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a) : "v"(hi[i]), "s"(P));
                asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(b2) : "v"(x), "v"(mh));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(h2) : "v"(a), "v"(b2));
                lo[i] = l2; hi[i] = h2;
            } else {
                uint32_t x, l2;
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(lo[i]), "v"(e));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(l2) : "v"(x), "s"(P));
                lo[i] = l2;
            }
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; ++i) s += ((uint64_t)hi[i] << 32) | lo[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// host check of the three 64-bit formulations against plain 64-bit arithmetic (one lane's chains are compared on the host)
static uint64_t ref_chain(uint64_t h, uint32_t e, int steps)
{
    for (int r = 0; r < steps; ++r) h = (h ^ e) * 1099511628211ULL;
    return h;
}
template <class F>
static void run(const char *name, F launch, int waves_per_simd)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: waves_per_simd waves * REP * 8 instructions
    double instr = (double)waves_per_simd * REP * 8;
    printf("%-22s waves/SIMD=%d  %8.3f ms  -> %6.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr);
}
int main()
{
    uint32_t *o; hipMalloc(&o, 256 * 4 * 8 * 1024 * 8);
    const char *names[] = {"v_mul_lo_u32", "v_xor_b32", "v_lshl_add_u32", "v_xor_b32_sdwa", "v_mul_u32_u24", "v_mad_u32_u24", "v_add_u32", "v_mul_hi_u32", "v_bfe_u32", "v_add3_u32", "v_bitop3_b32", "v_lshlrev_b32", "v_and_b32", "v_or_b32", "v_cndmask_b32", "v_cmp_ne_u32", "v_sub_u32", "v_min_u32", "v_ashrrev_i32", "v_lshrrev_b32", "v_mov_b32", "v_bfi_b32", "v_and_or_b32", "v_lshl_or_b32", "v_or3_b32", "v_addc_co_u32", "v_add_co_u32", "v_xor_b32 literal", "v_mul_lo_u32 sgpr", "v_and_b32 literal"};
    for (int wps : {4}) {
        dim3 grid(256 * wps), block(256);  // 256 CUs x wps blocks of 4 waves = wps waves per SIMD
        run(names[0], [&] { k<0><<<grid, block>>>(o, 1); }, wps);
        run(names[1], [&] { k<1><<<grid, block>>>(o, 1); }, wps);
        run(names[2], [&] { k<2><<<grid, block>>>(o, 1); }, wps);
        run(names[3], [&] { k<3><<<grid, block>>>(o, 1); }, wps);
        run(names[4], [&] { k<4><<<grid, block>>>(o, 1); }, wps);
        run(names[5], [&] { k<5><<<grid, block>>>(o, 1); }, wps);
        run(names[6], [&] { k<6><<<grid, block>>>(o, 1); }, wps);
        run(names[7], [&] { k<7><<<grid, block>>>(o, 1); }, wps);
        run(names[8], [&] { k<8><<<grid, block>>>(o, 1); }, wps);
        run(names[9], [&] { k<9><<<grid, block>>>(o, 1); }, wps);
        run(names[10], [&] { k<10><<<grid, block>>>(o, 1); }, wps);
        run(names[11], [&] { k<11><<<grid, block>>>(o, 1); }, wps);
#define RUN(i) run(names[i], [&] { k<i><<<grid, block>>>(o, 1); }, wps)
        RUN(12); RUN(13); RUN(14); RUN(15); RUN(16); RUN(17); RUN(18); RUN(19); RUN(20); RUN(21); RUN(22); RUN(23); RUN(24); RUN(25); RUN(26); RUN(27); RUN(28); RUN(29);
        run("v_mad_u64_u32", [&] { k64<0><<<grid, block>>>((uint64_t *)o, 1); }, wps);
        run("v_lshl_add_u64", [&] { k64<1><<<grid, block>>>((uint64_t *)o, 1); }, wps);

        // composite steps: ns per FNV step (one byte of one chain) per wave64 and SIMD
        {
            const char *vn[] = {"fnv64 step (engine)", "fnv64 step (reordered)", "fnv64 step (split state)", "fnv32 step"};
            uint64_t *o64 = (uint64_t *)o;
            auto chk = [&](int v) {  // thread 0 of block 0: sum of its 8 chains
                uint64_t got; hipMemcpy(&got, o64, 8, hipMemcpyDeviceToHost);
                uint64_t want = 0;
                for (int i = 0; i < 8; ++i) {
                    const uint64_t h0 = ((uint64_t)(uint32_t)(1 * 5 + 0 + i) << 32) | (uint32_t)(1 + 0 * 7 + i);
                    const uint64_t h = ref_chain(h0, (1 * 3 + 1) & 255u, REP);
                    want += v == 3 ? (((uint64_t)(uint32_t)(1 * 5 + 0 + i) << 32) | (uint32_t)h) : h;
                }
                printf("    %-26s %s\n", vn[v], got == want ? "matches 64-bit arithmetic" : "MISMATCH");
            };
            run(vn[0], [&] { kstep<0><<<grid, block>>>(o64, 1); }, wps); chk(0);
            run(vn[1], [&] { kstep<1><<<grid, block>>>(o64, 1); }, wps); chk(1);
            run(vn[2], [&] { kstep<2><<<grid, block>>>(o64, 1); }, wps); chk(2);
            run(vn[3], [&] { kstep<3><<<grid, block>>>(o64, 1); }, wps); chk(3);
        }
    }
    return 0;
}
