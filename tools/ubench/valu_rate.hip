// Micro-benchmark: issue rate of the VALU ops the PSA kernels are built from (gfx950).
// For each op: N independent instructions per loop iteration, timed with s_memtime on one
// workgroup per CU, for 1/2/4 waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void k(unsigned long long* out, unsigned seed, int iters)
{
    unsigned a0 = threadIdx.x ^ seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    unsigned b0 = a0 + 11, b1 = a1 + 13, b2 = a2 + 17, b3 = a3 + 19;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP64(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %6\n v_xor_b32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 1) { REP64(asm volatile("v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %5, %1, %6\n v_bfi_b32 %2, %6, %2, %7\n v_bfi_b32 %3, %7, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 2) { REP64(asm volatile("v_alignbit_b32 %0, %4, %0, 7\n v_alignbit_b32 %1, %5, %1, 9\n v_alignbit_b32 %2, %6, %2, 11\n v_alignbit_b32 %3, %7, %3, 13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 3) { REP64(asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0xe8\n v_bitop3_b32 %1, %1, %5, %6 bitop3:0xe8\n v_bitop3_b32 %2, %2, %6, %7 bitop3:0x96\n v_bitop3_b32 %3, %3, %7, %4 bitop3:0x96" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 4) { REP64(asm volatile("v_fma_f32 %0, %0, %0, %4\n v_fma_f32 %1, %1, %1, %5\n v_fma_f32 %2, %2, %2, %6\n v_fma_f32 %3, %3, %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 5) { REP64(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0));) }
        if (OP == 6) { REP64(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(d1));) }   // dependent chain
        if (OP == 8) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
        if (OP == 9) { REP64(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %2\n v_xor_b32 %0, %0, %3\n v_xor_b32 %0, %0, %4" : "+v"(a0) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }   // dependent chain
        if (OP == 10) { REP64(asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %5\n v_pk_add_u16 %2, %2, %6\n v_pk_add_u16 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a0 + a1 + a2 + a3 + (unsigned)(d0 + d1 + d2 + d3) + (unsigned)(f0 + f1 + f2 + f3) == 0x12345u) out[0] = 0;
}

template <int OP>
void run(const char* name)
{
    unsigned long long* d;
    hipMalloc(&d, 1024 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves_per_simd : {1, 2, 4, 8}) {
        // 8 waves/SIMD = two 1024-thread workgroups per CU
        const int blocks = waves_per_simd == 8 ? 512 : 256;
        const int threads = waves_per_simd == 8 ? 1024 : 256 * waves_per_simd;
        const int iters = 2000;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1u, iters);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1u, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += v;
        avg /= blocks;
        const double insts_per_simd = (double)iters * 64 * 4 * waves_per_simd;
        printf("%-22s waves/SIMD=%d  %.2f memtime-ticks, %.3f ns per wave-instruction per SIMD (kernel %.3f ms)\n", name,
               waves_per_simd, avg / insts_per_simd, 1e6 * ms / insts_per_simd, ms);
    }
    hipFree(d);
}

int main()
{
    run<0>("v_xor_b32");
    run<9>("v_xor_b32 dependent");
    run<8>("v_add_u32");
    run<1>("v_bfi_b32");
    run<2>("v_alignbit_b32");
    run<3>("v_bitop3_b32");
    run<10>("v_pk_add_u16");
    run<4>("v_fma_f32");
    run<5>("v_add_f64");
    run<6>("v_add_f64 dependent");
    return 0;
}
