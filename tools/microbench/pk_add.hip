// Microbenchmark: does v_pk_add_f32 (two rows per lane) run the dependent
// subtract chain of the value-sorted scan at twice the row rate of v_sub_f32?
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o pk_add pk_add.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(4))) * uniform_fp;

constexpr int K = 1024, CH = 32;

__global__ __launch_bounds__(256) void k_scalar(const float * l, const float * t0,
                                                float * out, int iters) {
    uniform_fp lp = (uniform_fp)(uintptr_t)l;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float t = t0[i];
    for (int it = 0; it < iters; ++it)
        for (int k0 = 0; k0 < K; k0 += CH) {
            float c[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) c[j] = lp[k0 + j];
#pragma unroll
            for (int j = 0; j < CH; ++j) t -= c[j];
        }
    out[i] = t;
}

// compiler-generated packed: two rows per lane
__global__ __launch_bounds__(256) void k_pk_cxx(const float * l, const float * t0,
                                                float * out, int iters) {
    uniform_fp lp = (uniform_fp)(uintptr_t)l;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    v2f t = {t0[2 * i], t0[2 * i + 1]};
    for (int it = 0; it < iters; ++it)
        for (int k0 = 0; k0 < K; k0 += CH) {
            float c[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) c[j] = lp[k0 + j];
#pragma unroll
            for (int j = 0; j < CH; ++j) t -= (v2f){c[j], c[j]};
        }
    out[2 * i] = t.x;
    out[2 * i + 1] = t.y;
}

// hand-placed: the scalar pair is the source, op_sel picks the half
__global__ __launch_bounds__(256) void k_pk_asm(const float * l, const float * t0,
                                                float * out, int iters) {
    uniform_fp lp = (uniform_fp)(uintptr_t)l;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    v2f t = {t0[2 * i], t0[2 * i + 1]};
    for (int it = 0; it < iters; ++it)
        for (int k0 = 0; k0 < K; k0 += CH) {
            v2f c[CH / 2];
#pragma unroll
            for (int j = 0; j < CH / 2; ++j)
                c[j] = (v2f){lp[k0 + 2 * j], lp[k0 + 2 * j + 1]};
#pragma unroll
            for (int j = 0; j < CH / 2; ++j) {
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
                             : "=v"(t) : "v"(t), "s"(c[j]));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]"
                             : "=v"(t) : "v"(t), "s"(c[j]));
            }
        }
    out[2 * i] = t.x;
    out[2 * i + 1] = t.y;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
    printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
    const size_t rows = 1 << 22;
    const int iters = 4;
    std::vector<float> l(K), t0(rows);
    srand(1);
    for (auto & x : l) x = (float)rand() / RAND_MAX * 1e-3f;
    l[5] = 1e-41f;   // a denormal entry: flushed alike?
    for (auto & x : t0) x = (float)rand() / RAND_MAX * 3.f;
    float *dl, *dt, *d0, *d1, *d2;
    CK(hipMalloc(&dl, K * 4)); CK(hipMalloc(&dt, rows * 4));
    CK(hipMalloc(&d0, rows * 4)); CK(hipMalloc(&d1, rows * 4)); CK(hipMalloc(&d2, rows * 4));
    CK(hipMemcpy(dl, l.data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, t0.data(), rows * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms[3];
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        k_scalar<<<rows / 256, 256>>>(dl, dt, d0, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[0], a, b));
        CK(hipEventRecord(a));
        k_pk_cxx<<<rows / 512, 256>>>(dl, dt, d1, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[1], a, b));
        CK(hipEventRecord(a));
        k_pk_asm<<<rows / 512, 256>>>(dl, dt, d2, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[2], a, b));
    }
    std::vector<float> h0(rows), h1(rows), h2(rows);
    CK(hipMemcpy(h0.data(), d0, rows * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), d1, rows * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), d2, rows * 4, hipMemcpyDeviceToHost));
    printf("scalar %.3f ms  pk_cxx %.3f ms  pk_asm %.3f ms  (rows %zu x %d entries)\n",
           ms[0], ms[1], ms[2], rows, K * iters);
    // one wave alone on its SIMD: the issue interval of a dependent chain
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        k_scalar<<<1, 64>>>(dl, dt, d0, 64);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[0], a, b));
        CK(hipEventRecord(a));
        k_pk_cxx<<<1, 64>>>(dl, dt, d1, 64);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[1], a, b));
    }
    printf("single wave, %d dependent ops: v_sub_f32 %.1f us  v_pk_add_f32 %.1f us\n",
           K * 64, ms[0] * 1e3, ms[1] * 1e3);
    printf("bit-identical: cxx %d asm %d\n",
           !memcmp(h0.data(), h1.data(), rows * 4), !memcmp(h0.data(), h2.data(), rows * 4));
    return 0;
}
