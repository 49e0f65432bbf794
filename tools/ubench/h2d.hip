// How long does 1 MB of pageable host memory take to reach the device?  (gpu_run_program's input path)
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/h2d tools/ubench/h2d.hip && tools/ubench/h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F> static double med(F f, int reps = 40) {
    std::vector<double> t;
    for (int k = 0; k < reps + 5; ++k) { const double a = now(); f(); const double b = now(); if (k >= 5) t.push_back(b - a); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}
int main() {
    const size_t len = 1010001;
    char* pageable = (char*)malloc(len);
    memset(pageable, 'A', len);
    char *pinned, *dev;
    hipHostMalloc((void**)&pinned, len, hipHostMallocDefault);
    hipMalloc((void**)&dev, len);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    printf("pageable hipMemcpyAsync + sync           : %6.1f us\n", med([&] { hipMemcpyAsync(dev, pageable, len, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }));
    printf("pageable hipMemcpy (blocking)            : %6.1f us\n", med([&] { hipMemcpy(dev, pageable, len, hipMemcpyHostToDevice); }));
    printf("pinned   hipMemcpyAsync + sync           : %6.1f us\n", med([&] { hipMemcpyAsync(dev, pinned, len, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }));
    printf("memcpy to pinned                         : %6.1f us\n", med([&] { memcpy(pinned, pageable, len); }));
    printf("memcpy to pinned + async + sync          : %6.1f us\n", med([&] { memcpy(pinned, pageable, len); hipMemcpyAsync(dev, pinned, len, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }));
    for (int chunks : {2, 4, 8}) {
        const size_t c = (len + chunks - 1) / chunks;
        printf("%d chunks: memcpy k+1 while chunk k flies : %6.1f us\n", chunks, med([&] {
            for (int k = 0; k < chunks; ++k) {
                const size_t off = k * c, n = std::min(c, len - off);
                memcpy(pinned + off, pageable + off, n);
                hipMemcpyAsync(dev + off, pinned + off, n, hipMemcpyHostToDevice, s);
            }
            hipStreamSynchronize(s);
        }));
    }
    printf("pageable hipMemcpyAsync, enqueue only     : %6.1f us (the call returns)\n", med([&] { hipMemcpyAsync(dev, pageable, len, hipMemcpyHostToDevice, s); }));
    hipStreamSynchronize(s);
    return 0;
}
