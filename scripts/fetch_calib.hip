// Calibration of rocprofv3's FETCH_SIZE for the access widths the conv loaders use (MI355X guide: 16 B/lane streaming
// reads report 1/2 of their bytes; other widths are uncalibrated).  Each kernel streams the same 1 GiB buffer once
// (larger than the 256 MiB Infinity Cache), with 4-, 8- and 16-byte loads per lane.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib scripts/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T> __global__ void stream_read(const T* __restrict__ x, size_t n, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) {
        T v = x[i];
        const float* f = reinterpret_cast<const float*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc += f[k];
    }
    if (acc == 12345.678f) *out = acc;
}
// the conv halo pattern: a wave reads 34 consecutive floats of a row (lanes 0..33), rows 600 floats apart
__global__ void halo_read(const float* __restrict__ x, size_t rows, int pitch, float* out) {
    int lane = threadIdx.x & 63;
    size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    for (size_t r = wave; r < rows; r += nw)
        for (int s = 0; s + 34 <= pitch; s += 32)
            if (lane < 34) acc += x[r * pitch + s + lane];
    if (acc == 12345.678f) *out = acc;
}
int main() {
    size_t bytes = 1ull << 30;
    float *x, *out;
    hipMalloc(&x, bytes + 4096); hipMalloc(&out, 4);
    hipMemset(x, 0, bytes + 4096);
    for (int rep = 0; rep < 2; ++rep) {
        stream_read<float><<<4096, 256>>>(x, bytes / 4, out);
        stream_read<float2><<<4096, 256>>>((const float2*)x, bytes / 8, out);
        stream_read<float4><<<4096, 256>>>((const float4*)x, bytes / 16, out);
        halo_read<<<4096, 256>>>(x, bytes / 4 / 608, 608, out);
    }
    hipDeviceSynchronize();
    printf("streamed %zu bytes per kernel; halo kernel touches %zu rows of 608 floats\n", bytes, bytes / 4 / 608);
    return 0;
}
