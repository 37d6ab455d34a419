// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths k_tile uses (development aid):
// each kernel streams a known number of bytes once; compare with the counter (MI355X_MICROARCH.md:
// gfx950 reports half the bytes of 16-B-per-lane reads; other widths must be calibrated).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/micro/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <class T>
__global__ void k_read(const T *__restrict__ src, size_t n, unsigned long long *sink) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = src[i];
        const uint32_t *w = (const uint32_t *)&v;
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc += w[k];
    }
    if (acc == 0x123456789abcull) *sink = acc;   // keeps the loads alive, practically never true
}

__global__ void k_write4(uint32_t *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint32_t)i;
}
__global__ void k_write16(uint4 *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4(i, i, i, i);
}

int main() {
    const size_t bytes = (size_t)1 << 30;   // 1 GiB, well past the 256 MiB Infinity Cache
    void *buf; unsigned long long *sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 8);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    k_read<uint32_t><<<4096, 256>>>((const uint32_t *)buf, bytes / 4, sink);
    k_read<uint2><<<4096, 256>>>((const uint2 *)buf, bytes / 8, sink);
    k_read<uint4><<<4096, 256>>>((const uint4 *)buf, bytes / 16, sink);
    k_write4<<<4096, 256>>>((uint32_t *)buf, bytes / 4);
    k_write16<<<4096, 256>>>((uint4 *)buf, bytes / 16);
    hipDeviceSynchronize();
    printf("each kernel moved %zu bytes\n", bytes);
    return 0;
}
