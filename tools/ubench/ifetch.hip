// Is the count kernel's fully unrolled round (~650 VALU instructions, ~5 KB) limited by instruction
// fetch?  Same instruction mix (13 bitop3 + 4 alignbit + 2 and + 1 add per "char"), body unrolled
// U chars, 16 waves per CU on every CU; time per char for U = 1, 4, 32.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAR_BODY \
    "v_alignbit_b32 v20, v9, v8, 7\n v_alignbit_b32 v21, v11, v10, 7\n v_alignbit_b32 v22, v13, v12, 7\n v_alignbit_b32 v23, v15, v14, 7\n" \
    "v_and_b32 v24, v20, v21\n v_and_b32 v25, v22, v23\n" \
    "v_bitop3_b32 v26, v16, v20, v24 bitop3:0xe8\n v_bitop3_b32 v16, v16, v20, v24 bitop3:0x96\n" \
    "v_bitop3_b32 v27, v17, v21, v25 bitop3:0xe8\n v_bitop3_b32 v17, v17, v21, v25 bitop3:0x96\n" \
    "v_bitop3_b32 v28, v18, v22, v26 bitop3:0xe8\n v_bitop3_b32 v18, v18, v22, v26 bitop3:0x96\n" \
    "v_bitop3_b32 v29, v19, v23, v27 bitop3:0xe8\n v_bitop3_b32 v19, v19, v23, v27 bitop3:0x96\n" \
    "v_bitop3_b32 v30, v16, v28, v29 bitop3:0xe8\n v_bitop3_b32 v16, v16, v28, v29 bitop3:0x96\n" \
    "v_bitop3_b32 v31, v17, v30, v24 bitop3:0xe8\n v_bitop3_b32 v17, v17, v30, v24 bitop3:0x96\n" \
    "v_bitop3_b32 v18, v18, v31, v25 bitop3:0x96\n" \
    "v_add_u32 v8, v8, v16\n"
#define R2(x) x x
#define R4(x) R2(R2(x))
#define R32(x) R4(R4(R2(x)))
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"
template <int U>
__global__ __launch_bounds__(1024) void k(unsigned* out, int chars)
{
    asm volatile("v_mov_b32 v8, v0\n v_mov_b32 v9, v0\n v_mov_b32 v10, v0\n v_mov_b32 v11, v0\n v_mov_b32 v12, v0\n v_mov_b32 v13, v0\n v_mov_b32 v14, v0\n v_mov_b32 v15, v0\n"
                 "v_mov_b32 v16, v0\n v_mov_b32 v17, v0\n v_mov_b32 v18, v0\n v_mov_b32 v19, v0" ::: CLOB);
    for (int i = 0; i < chars; i += U) {
        if (U == 1) asm volatile(CHAR_BODY ::: CLOB);
        if (U == 4) asm volatile(R4(CHAR_BODY) ::: CLOB);
        if (U == 32) asm volatile(R32(CHAR_BODY) ::: CLOB);
    }
    unsigned r;
    asm volatile("v_xor_b32 %0, v16, v17" : "=v"(r));
    if (r == 0x12345u) out[0] = r;
}
template <int U> void run_occ(int blocks, int threads, int waves_per_simd)
{
    unsigned* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int chars = 32 * 400;
    hipLaunchKernelGGL(k<U>, dim3(blocks), dim3(threads), 0, 0, d, chars);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<U>, dim3(blocks), dim3(threads), 0, 0, d, chars);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("unroll %2d, %d waves/SIMD: %.2f ns per wave-char per SIMD\n", U, waves_per_simd, 1e6 * ms / ((double)waves_per_simd * chars));
    hipFree(d);
}
template <int U> void run()
{
    unsigned* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int chars = 32 * 400;
    hipLaunchKernelGGL(k<U>, dim3(256), dim3(1024), 0, 0, d, chars);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<U>, dim3(256), dim3(1024), 0, 0, d, chars);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x chars
    printf("unroll %2d chars: %.2f ns per wave-char per SIMD (20 VALU instr each; ideal ~ 13*1.0+4*1.75+3*1.0 = 23 ns)\n", U, 1e6 * ms / (4.0 * chars));
    hipFree(d);
}
int main()
{
    run<1>(); run<4>(); run<32>();
    run_occ<32>(256, 512, 2);
    run_occ<32>(256, 768, 3);
    run_occ<32>(256, 1024, 4);
    run_occ<32>(512, 768, 6);
    run_occ<32>(512, 1024, 8);
    return 0;
}
