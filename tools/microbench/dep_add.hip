// How long is a link of a dependent v_add_f32 chain on gfx950, for one wave
// alone on its SIMD?  (a) operands in registers, (b) operands arriving from
// LDS as 16 uniform ds_read_b128 per 64 links (strip_total's shape), (c) the
// same with the next chunk's reads issued before this chunk's adds.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off dep_add.hip -o dep_add
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_regs(float * out, long long * cyc, int reps, float a, float b) {
    float t = a;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t) : "v"(b));
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = t; cyc[blockIdx.x] = t1 - t0; }
}

template <int AHEAD>
__global__ void k_lds(float * out, long long * cyc, int n, int reps) {
    extern __shared__ float strip[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) strip[i] = 1e-3f * (i & 7);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    float total = 0.f;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if (AHEAD == 0) {
            for (int k0 = 0; k0 < n; k0 += 64) {
                float4 v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const float4 *>(strip + k0 + 4 * q);
#pragma unroll
                for (int q = 0; q < 16; ++q) { total += v[q].x; total += v[q].y; total += v[q].z; total += v[q].w; }
            }
        } else {
            float4 v[16], w[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const float4 *>(strip + 4 * q);
            for (int k0 = 0; k0 < n; k0 += 64) {
                const int k1 = k0 + 64 < n ? k0 + 64 : 0;
#pragma unroll
                for (int q = 0; q < 16; ++q) w[q] = *reinterpret_cast<const float4 *>(strip + k1 + 4 * q);
#pragma unroll
                for (int q = 0; q < 16; ++q) { total += v[q].x; total += v[q].y; total += v[q].z; total += v[q].w; }
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = w[q];
            }
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = total; cyc[blockIdx.x] = t1 - t0; }
}

int main() {
    float * out; long long * cyc;
    CHECK(hipMalloc(&out, 4096)); CHECK(hipMalloc(&cyc, 8192));
    const int n = 1024, reps = 200;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int blocks : {1, 256, 512}) {
        for (int variant = 0; variant < 3; ++variant) {
            for (int warm = 0; warm < 2; ++warm) {
                CHECK(hipEventRecord(e0, 0));
                if (variant == 0) hipLaunchKernelGGL(k_regs, dim3(blocks), dim3(64), 0, 0, out, cyc, reps * n / 64, 0.f, 1e-3f);
                else if (variant == 1) hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(256), n * 4, 0, out, cyc, n, reps);
                else hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(256), n * 4, 0, out, cyc, n, reps);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipDeviceSynchronize());
            }
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<long long> h(blocks);
            CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
            double links = (double)reps * n;
            printf("%3d workgroups, %-28s s_memtime ticks/link %6.2f   wall ns/link %6.2f (= %5.2f cycles at 2.4 GHz)\n",
                   blocks, variant == 0 ? "registers" : variant == 1 ? "LDS, 16 x b128 then 64 adds" : "LDS, next chunk read ahead",
                   h[0] / links, ms * 1e6 / links, ms * 1e6 / links * 2.4);
        }
    }
    return 0;
}
