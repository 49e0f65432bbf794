// Does a 3-source VOP3 op pay for VGPR bank conflicts on gfx950?  (bank = register index mod 4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ void k(unsigned* out, int iters)
{
    // pin registers: v8..v31
    asm volatile("v_mov_b32 v8, v0\n v_mov_b32 v9, v0\n v_mov_b32 v10, v0\n v_mov_b32 v11, v0\n"
                 "v_mov_b32 v12, v0\n v_mov_b32 v13, v0\n v_mov_b32 v14, v0\n v_mov_b32 v15, v0\n"
                 "v_mov_b32 v16, v0\n v_mov_b32 v17, v0\n v_mov_b32 v18, v0\n v_mov_b32 v19, v0\n"
                 "v_mov_b32 v20, v0\n v_mov_b32 v21, v0\n v_mov_b32 v22, v0\n v_mov_b32 v23, v0\n"
                 ::: "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23");
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP64(asm volatile("v_bitop3_b32 v8, v8, v9, v10 bitop3:0x96\n v_bitop3_b32 v12, v12, v13, v14 bitop3:0x96\n v_bitop3_b32 v16, v16, v17, v18 bitop3:0xe8\n v_bitop3_b32 v20, v20, v21, v22 bitop3:0xe8" ::: "v8","v12","v16","v20");) }   // sources in 3 different banks
        if (OP == 1) { REP64(asm volatile("v_bitop3_b32 v8, v8, v12, v16 bitop3:0x96\n v_bitop3_b32 v9, v9, v13, v17 bitop3:0x96\n v_bitop3_b32 v10, v10, v14, v18 bitop3:0xe8\n v_bitop3_b32 v11, v11, v15, v19 bitop3:0xe8" ::: "v8","v9","v10","v11");) }   // all 3 sources in the same bank
        if (OP == 2) { REP64(asm volatile("v_bitop3_b32 v8, v8, v12, v13 bitop3:0x96\n v_bitop3_b32 v9, v9, v13, v14 bitop3:0x96\n v_bitop3_b32 v10, v10, v14, v15 bitop3:0xe8\n v_bitop3_b32 v11, v11, v15, v12 bitop3:0xe8" ::: "v8","v9","v10","v11");) }   // two sources share a bank
        if (OP == 3) { REP64(asm volatile("v_alignbit_b32 v8, v12, v16, 7\n v_alignbit_b32 v9, v13, v17, 7\n v_alignbit_b32 v10, v14, v18, 9\n v_alignbit_b32 v11, v15, v19, 9" ::: "v8","v9","v10","v11");) }
        if (OP == 4) { REP64(asm volatile("v_xor_b32 v8, v12, v16\n v_xor_b32 v9, v13, v17\n v_xor_b32 v10, v14, v18\n v_xor_b32 v11, v15, v19" ::: "v8","v9","v10","v11");) }   // 2 sources same bank
        if (OP == 5) { REP64(asm volatile("v_lshl_or_b32 v8, v12, 3, v17\n v_lshl_or_b32 v9, v13, 3, v18\n v_lshl_or_b32 v10, v14, 3, v19\n v_lshl_or_b32 v11, v15, 3, v16" ::: "v8","v9","v10","v11");) }
        if (OP == 6) { REP64(asm volatile("v_lshrrev_b32 v8, 5, v12\n v_lshrrev_b32 v9, 5, v13\n v_lshrrev_b32 v10, 5, v14\n v_lshrrev_b32 v11, 5, v15" ::: "v8","v9","v10","v11");) }
        if (OP == 7) { REP64(asm volatile("v_perm_b32 v8, v12, v17, v22\n v_perm_b32 v9, v13, v18, v23\n v_perm_b32 v10, v14, v19, v20\n v_perm_b32 v11, v15, v16, v21" ::: "v8","v9","v10","v11");) }
        if (OP == 8) { REP64(asm volatile("v_and_or_b32 v8, v12, v17, v22\n v_and_or_b32 v9, v13, v18, v23\n v_and_or_b32 v10, v14, v19, v20\n v_and_or_b32 v11, v15, v16, v21" ::: "v8","v9","v10","v11");) }
        if (OP == 9) { REP64(asm volatile("v_lshrrev_b64 v[8:9], 5, v[12:13]\n v_lshrrev_b64 v[10:11], 5, v[14:15]\n v_lshrrev_b64 v[16:17], 7, v[18:19]\n v_lshrrev_b64 v[20:21], 7, v[22:23]" ::: "v8","v9","v10","v11","v16","v17","v20","v21");) }
    }
    unsigned r;
    asm volatile("v_xor_b32 %0, v8, v9\n v_xor_b32 %0, %0, v10\n v_xor_b32 %0, %0, v11" : "=v"(r));
    if (r == 0x12345678u) out[0] = r;
}
template <int OP> void run(const char* name)
{
    unsigned* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(512), dim3(1024), 0, 0, d, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(512), dim3(1024), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ns per wave-instruction per SIMD (8 waves/SIMD)\n", name, 1e6 * ms / ((double)iters * 64 * 4 * 8));
    hipFree(d);
}
int main()
{
    run<0>("v_bitop3_b32, sources in 3 banks");
    run<1>("v_bitop3_b32, all 3 sources in one bank");
    run<2>("v_bitop3_b32, 2 sources share a bank");
    run<3>("v_alignbit_b32 (2 VGPR sources same bank)");
    run<4>("v_xor_b32 (2 sources same bank)");
    run<5>("v_lshl_or_b32");
    run<6>("v_lshrrev_b32");
    run<7>("v_perm_b32");
    run<8>("v_and_or_b32");
    run<9>("v_lshrrev_b64");
    return 0;
}
