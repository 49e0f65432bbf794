// ds_read_addtid_b32 on gfx950: which dword does lane l of wave w get?  LDS_ADDR = M0[15:0] + offset + TID*4 — TID inside
// the wave or inside the workgroup, and is the sum taken beyond 16 bits?  (hipcc --offload-arch=gfx950 -O3 addtid.hip -o addtid)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p, unsigned off)
{
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 36864; i += blockDim.x) lds[i] = i;              // 144 KB, dword i holds i
    __syncthreads();
    unsigned a, b, c;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)lds;
    const unsigned m = __builtin_amdgcn_readfirstlane(off + base);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tds_read_addtid_b32 %0 offset:0\n\tds_read_addtid_b32 %1 offset:4096\n\t"
                 "ds_read_addtid_b32 %2 offset:60000\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "s"(m) : "memory");
    p[3 * threadIdx.x] = a;
    p[3 * threadIdx.x + 1] = b;
    p[3 * threadIdx.x + 2] = c;
}
int main()
{
    unsigned* d;
    (void)hipMalloc(&d, 3 * 256 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    for (unsigned off : {0u, 40u, 20000u, 56000u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 147456, 0, d, off);
        unsigned h[768];
        (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("M0 = %5u bytes: thread 0 -> dwords %u, %u, %u; thread 1 -> %u; thread 63 -> %u; thread 64 -> %u, %u, %u; thread 255 -> %u\n", off,
               h[0], h[1], h[2], h[3], h[189], h[192], h[193], h[194], h[765]);
    }
    printf("(expected for TID = lane in wave, full-width sum: thread 0 -> M0/4, M0/4 + 1024, M0/4 + 15000; thread 64 the same)\n");
    return 0;
}
