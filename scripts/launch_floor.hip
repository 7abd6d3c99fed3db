// GPU box micro-benchmark: cost of a chain of dependent kernel launches on one stream (the LSTM sweep's skeleton).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(float* p) {}
__global__ void k_touch(float* p, const int* idx) {            // one dependent global load + store per thread
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    p[i] = p[idx[i & 1023] + i] + 1.0f;
}
__global__ void k_lds(float* p) {
    __shared__ float s[256];
    s[threadIdx.x] = p[blockIdx.x * 256 + threadIdx.x];
    __syncthreads();
    p[blockIdx.x * 256 + threadIdx.x] = s[255 - threadIdx.x];
}
template <typename F> double run(F f, hipStream_t s, int n) {
    for (int i = 0; i < 50; ++i) f();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < n; ++i) f();
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / n;
}
int main() {
    float* p; int* idx;
    hipMalloc(&p, 1 << 24); hipMalloc(&idx, 4096); hipMemset(p, 0, 1 << 24); hipMemset(idx, 0, 4096);
    hipStream_t s; hipStreamCreate(&s);
    for (int wgs : {64, 128, 256, 512}) {
        printf("WGs %3d: empty %.2f us  touch %.2f us  lds %.2f us per dependent launch\n", wgs,
               run([&] { k_empty<<<wgs, 256, 0, s>>>(p); }, s, 2000), run([&] { k_touch<<<wgs, 256, 0, s>>>(p, idx); }, s, 2000),
               run([&] { k_lds<<<wgs, 256, 0, s>>>(p); }, s, 2000));
    }
    return 0;
}
