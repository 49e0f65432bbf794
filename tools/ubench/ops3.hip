// Which part of the count kernel's VALU mix runs below its isolated rate?  4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R2(x) x x
#define R4(x) R2(R2(x))
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters)
{
    asm volatile("v_mov_b32 v8, v0\n v_mov_b32 v9, v0\n v_mov_b32 v10, v0\n v_mov_b32 v11, v0\n v_mov_b32 v12, v0\n v_mov_b32 v13, v0\n v_mov_b32 v14, v0\n v_mov_b32 v15, v0\n"
                 "v_mov_b32 v16, v0\n v_mov_b32 v17, v0\n v_mov_b32 v18, v0\n v_mov_b32 v19, v0" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
        // 0: bitop3, dest == src0 (accumulate)           4 instr
        if (OP == 0) asm volatile(R64("v_bitop3_b32 v16, v16, v9, v10 bitop3:0x96\n v_bitop3_b32 v17, v17, v10, v11 bitop3:0x96\n v_bitop3_b32 v18, v18, v11, v12 bitop3:0xe8\n v_bitop3_b32 v19, v19, v12, v13 bitop3:0xe8\n") ::: CLOB);
        // 1: bitop3, dest distinct from its three sources
        if (OP == 1) asm volatile(R64("v_bitop3_b32 v20, v16, v9, v10 bitop3:0x96\n v_bitop3_b32 v21, v17, v10, v11 bitop3:0x96\n v_bitop3_b32 v22, v18, v11, v12 bitop3:0xe8\n v_bitop3_b32 v23, v19, v12, v13 bitop3:0xe8\n") ::: CLOB);
        // 2: carry-save pairs exactly as the kernel emits them: maj into a new register, xor3 in place
        if (OP == 2) asm volatile(R64("v_bitop3_b32 v20, v16, v9, v10 bitop3:0xe8\n v_bitop3_b32 v16, v16, v9, v10 bitop3:0x96\n v_bitop3_b32 v21, v17, v11, v12 bitop3:0xe8\n v_bitop3_b32 v17, v17, v11, v12 bitop3:0x96\n") ::: CLOB);
        // 3: dependent carry-save chain (level 0 -> level 1)
        if (OP == 3) asm volatile(R64("v_bitop3_b32 v20, v16, v9, v10 bitop3:0xe8\n v_bitop3_b32 v16, v16, v9, v10 bitop3:0x96\n v_bitop3_b32 v21, v17, v20, v12 bitop3:0xe8\n v_bitop3_b32 v17, v17, v20, v12 bitop3:0x96\n") ::: CLOB);
        // 4: alignbit x4 (literal shift)
        if (OP == 4) asm volatile(R64("v_alignbit_b32 v20, v9, v8, 7\n v_alignbit_b32 v21, v11, v10, 7\n v_alignbit_b32 v22, v13, v12, 7\n v_alignbit_b32 v23, v15, v14, 7\n") ::: CLOB);
        // 5: alignbit feeding and feeding bitop3 (the kernel's per-word head)
        if (OP == 5) asm volatile(R64("v_alignbit_b32 v20, v9, v8, 7\n v_alignbit_b32 v21, v11, v10, 7\n v_and_b32 v24, v20, v21\n v_bitop3_b32 v16, v16, v20, v24 bitop3:0x96\n") ::: CLOB);
        // 6: v_xor_b32 e32 accumulate
        if (OP == 6) asm volatile(R64("v_xor_b32 v16, v16, v9\n v_xor_b32 v17, v17, v10\n v_xor_b32 v18, v18, v11\n v_xor_b32 v19, v19, v12\n") ::: CLOB);
        // 7: the same v_xor in its 8-byte VOP3 encoding: does the rate follow the code bytes (instruction fetch)?
        if (OP == 7) asm volatile(R64("v_xor_b32_e64 v16, v16, v9\n v_xor_b32_e64 v17, v17, v10\n v_xor_b32_e64 v18, v18, v11\n v_xor_b32_e64 v19, v19, v12\n") ::: CLOB);
        // 8: half 4-byte, half 8-byte
        if (OP == 8) asm volatile(R64("v_xor_b32 v16, v16, v9\n v_bitop3_b32 v17, v17, v10, v11 bitop3:0x96\n v_xor_b32 v18, v18, v11\n v_bitop3_b32 v19, v19, v12, v13 bitop3:0xe8\n") ::: CLOB);
    }
    unsigned r;
    asm volatile("v_xor_b32 %0, v16, v17" : "=v"(r));
    if (r == 0x12345u) out[0] = r;
}
template <int OP> void run(const char* name)
{
    unsigned* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %.3f ns per wave-instruction per SIMD (4 waves/SIMD)\n", name, 1e6 * ms / ((double)iters * 64 * 4 * 4));
    hipFree(d);
}
int main()
{
    run<0>("bitop3, dest == src0");
    run<1>("bitop3, dest distinct from sources");
    run<2>("carry-save pair (maj new reg, xor3 in place)");
    run<3>("dependent carry-save chain");
    run<4>("alignbit x4");
    run<5>("alignbit, alignbit, and, bitop3 (dependent)");
    run<6>("v_xor_b32 accumulate");
    run<7>("v_xor_b32_e64 (8-byte encoding) accumulate");
    run<8>("v_xor_b32 e32 / v_bitop3_b32 alternating");
    return 0;
}
