// Microbenchmark: what one SIMD of gfx950 issues per cycle of plain f32 VALU
// work -- the roof bench.py's `roofline` prices the VALU-bound kernels against.
//
// Every wave runs `ITERS` rounds over ACC independent accumulators (no
// dependency stall: ACC = 8 chains per lane), W waves share a SIMD
// (W = 1, 2, 4, 8), every CU is busy.  Cycles come from s_memtime inside the
// kernel (shader cycles: free of launch overhead and of the clock the chip
// happens to hold), the clock from s_memrealtime (100 MHz).  Three streams:
//   v_add_f32            one row per lane
//   v_pk_add_f32         two rows per lane (what k_vs_sample's recurrences use)
//   dependent v_add_f32  ONE chain per lane (ACC = 1)
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2048;

struct Stamp { unsigned long long cycles, real; };

template <int ACC, int PACKED>
__global__ __launch_bounds__(1024) void k_stream(const float * in, float * out,
                                                 Stamp * stamps) {
    const float c = in[0];
    v2f a[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i)
        a[i] = (v2f){in[1 + i] + (float)threadIdx.x, in[9 + i]};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) {
            if (PACKED) {
                a[i] += (v2f){c, c};
            } else {
                a[i].x += c;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += a[i].x + a[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0)
        stamps[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] =
            Stamp{t1 - t0, r1 - r0};
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
    printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int ACC, int PACKED>
static void run(const char * name, const float * din, float * dout,
                Stamp * dst, int cus) {
    for (int W : {1, 2, 4, 8}) {
        // W waves per SIMD: one block of 256*W threads per CU (two of 1024
        // for W = 8)
        const int threads = W <= 4 ? 256 * W : 1024;
        const int blocks = cus * (W <= 4 ? 1 : 2);
        const size_t waves = (size_t)blocks * threads / 64;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            k_stream<ACC, PACKED><<<blocks, threads>>>(din, dout, dst);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<Stamp> st(waves);
        CK(hipMemcpy(st.data(), dst, waves * sizeof(Stamp),
                     hipMemcpyDeviceToHost));
        std::vector<double> cyc, clk;
        for (auto & s : st) {
            cyc.push_back((double)s.cycles);
            clk.push_back((double)s.cycles / (double)s.real * 100.0);
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(clk.begin(), clk.end());
        const double instr = (double)ITERS * ACC;       // per wave
        printf("%-22s W=%d  cycles/instr/SIMD %.2f (median wave %.0f cycles "
               "for %.0f instr x %d waves)  clock %.0f MHz  kernel %.1f us\n",
               name, W, cyc[cyc.size() / 2] / (instr * W), cyc[cyc.size() / 2],
               instr, W, clk[clk.size() / 2], ms * 1e3);
    }
}

int main() {
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    std::vector<float> in(32);
    for (size_t i = 0; i < in.size(); ++i) in[i] = 1e-3f * (float)(i + 1);
    float *din, *dout;
    Stamp * dst;
    const size_t max_threads = (size_t)cus * 2 * 1024;
    CK(hipMalloc(&din, in.size() * 4));
    CK(hipMalloc(&dout, max_threads * 4));
    CK(hipMalloc(&dst, max_threads / 64 * sizeof(Stamp)));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    printf("%d CUs\n", cus);
    run<8, 0>("v_add_f32 x8 chains", din, dout, dst, cus);
    run<8, 1>("v_pk_add_f32 x8 chains", din, dout, dst, cus);
    run<1, 0>("v_add_f32 dependent", din, dout, dst, cus);
    run<1, 1>("v_pk_add_f32 dependent", din, dout, dst, cus);
    return 0;
}
