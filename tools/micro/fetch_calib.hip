// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this library uses
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel streams a KNOWN number
// of bytes from a 1 GiB buffer (4x the 256 MiB Infinity Cache) exactly once:
//   read_b32_buffer   raw_buffer_load_b32, 4 B per lane, coalesced  (the conv / wgrad kernels' staging loads)
//   read_b32_rows     raw_buffer_load_b32 in rows of 34 floats out of 240-float lines (the conv kernel's halo-row shape)
//   read_b128_buffer  raw_buffer_load_b128, 16 B per lane           (GroupNorm slices)
//   read_b128_global  global_load_dwordx4                           (cost volume, elementwise kernels)
//   write_b128        16 B per lane stores
// Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- ./fetch_calib   (and WRITE_SIZE)
// tools/pmc_traffic.py turns the counter files into correction factors (profiles/r02_pmc_traffic.json).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void read_b32_buffer(const float* p, size_t n, float* sink) {
    float acc = 0.f;
    const size_t chunk = (size_t)1 << 28;                 // 1 GiB buffer = 4 descriptors of 2^28 floats... keep offsets < 2^31
    for (size_t base = 0; base < n; base += chunk) {
        const size_t len = n - base < chunk ? n - base : chunk;
        auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p + base), 0, (unsigned)(len * 4), 0x00020000);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
            acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)(i * 4), 0, 0));
    }
    if (acc == 123.456f) sink[0] = acc;
}

// rows of 34 consecutive floats at the start of every 240-float line: 34/240 of the buffer is touched
__global__ void read_b32_rows(const float* p, size_t n, float* sink) {
    float acc = 0.f;
    const size_t nrows = n / 240, chunk_rows = ((size_t)1 << 28) / 240;
    for (size_t r0 = 0; r0 < nrows; r0 += chunk_rows) {
        const size_t rows = nrows - r0 < chunk_rows ? nrows - r0 : chunk_rows;
        auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p + r0 * 240), 0, (unsigned)(rows * 240 * 4), 0x00020000);
        const size_t total = rows * 34;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const size_t row = i / 34, col = i % 34;
            acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)((row * 240 + col) * 4), 0, 0));
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void read_b128_buffer(const float* p, size_t n, float* sink) {
    float acc = 0.f;
    const size_t chunk = (size_t)1 << 28;
    for (size_t base = 0; base < n; base += chunk) {
        const size_t len = (n - base < chunk ? n - base : chunk) / 4;
        auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p + base), 0, (unsigned)(len * 16), 0x00020000);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(i * 16), 0, 0);
            acc += __builtin_bit_cast(float, v.x) + __builtin_bit_cast(float, v.w);
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void read_b128_global(const float4* p, size_t n4, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void write_b128(float4* p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const size_t n = (size_t)1 << 28;                      // floats = 1 GiB
    float *buf = nullptr, *sink = nullptr;
    if (hipMalloc(&buf, n * 4) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    if (hipMemset(buf, 0, n * 4) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 1;
    const dim3 grid(2048), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_b32_buffer, grid, block, 0, 0, buf, n, sink);
        hipLaunchKernelGGL(read_b32_rows, grid, block, 0, 0, buf, n, sink);
        hipLaunchKernelGGL(read_b128_buffer, grid, block, 0, 0, buf, n, sink);
        hipLaunchKernelGGL(read_b128_global, grid, block, 0, 0, reinterpret_cast<const float4*>(buf), n / 4, sink);
        hipLaunchKernelGGL(write_b128, grid, block, 0, 0, reinterpret_cast<float4*>(buf), n / 4);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    const size_t rows = n / 240;
    printf("bytes read_b32_buffer %zu\nbytes read_b32_rows %zu (34 of every 240 floats; lines touched: %zu B if whole 128-B lines count)\n"
           "bytes read_b128_buffer %zu\nbytes read_b128_global %zu\nbytes write_b128 %zu\n",
           n * 4, rows * 34 * 4, rows * 256, n * 4, n * 4, n * 4);
    return 0;
}
