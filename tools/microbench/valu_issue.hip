// Microbenchmark: how many cycles one SIMD of gfx950 needs per vector
// instruction -- the roof bench.py's `roofline` prices the VALU-bound kernels
// against, instruction class by instruction class.
//
// Every wave issues ITERS x 8 copies of ONE instruction (inline asm, so it is
// the instruction named), over eight independent registers ("x8") or as one
// dependent chain ("dep", the shape of k_vs_sample's recurrences).  Workgroups
// of 256 threads, W of them per CU.  Where the dispatcher really put the
// waves is read back (HW_ID), and the figure is computed PER SIMD: (last end -
// first start of the waves that ran on it) / (instructions they issued), in
// s_memtime cycles (shader clock; its rate against the 100 MHz s_memrealtime
// is printed); the median over SIMDs and the resident waves per SIMD are
// reported.  (The round's first version divided the median wave's time by the
// waves it ASSUMED shared a SIMD; two 1024-thread workgroups per CU ran one
// after the other, so its eight-wave lines were four-wave lines.)
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

constexpr int ITERS = 1024;

struct Stamp { unsigned long long t0, t1, r0, r1, hw; };

// eight independent instructions / eight links of one chain
#define X8(T)  T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)

enum Op { ADD_F32, PK_ADD_F32, FMA_F32, PK_FMA_F32, MUL_F32, MOV_B32,
          CNDMASK, CMP_EQ, ADD_U32, AND_B32, LSHL, EXP_F32, LOG_F32, RCP_F32,
          CVT_F32_U32, MAD_U32_U24, ADD_F32_DEP, PK_ADD_F32_DEP, READLANE,
          PK_ADD_SGPR, CNDMASK_E64, CMP_CNDMASK, ADDC, CNDMASK_OTHER_DST,
          MAX_F32, CMP_GT_F32, EXEC_ADD, BFE_U32, CMP_CND3, CND_VALU_VCC, CMP_MOV_CND, CMP64_CND64, CMP_CND_ADD_CND, CMP_CND_ADDC, CMP64_CND64x3, N_OPS };
static const char * kNames[N_OPS] = {
    "v_add_f32 x8", "v_pk_add_f32 x8", "v_fma_f32 x8", "v_pk_fma_f32 x8",
    "v_mul_f32 x8", "v_mov_b32 x8", "v_cndmask_b32 x8", "v_cmp_eq_u32 x8",
    "v_add_u32 x8", "v_and_b32 x8", "v_lshlrev_b32 x8", "v_exp_f32 x8",
    "v_log_f32 x8", "v_rcp_f32 x8", "v_cvt_f32_u32 x8", "v_mad_u32_u24 x8",
    "v_add_f32 dep", "v_pk_add_f32 dep", "v_readlane_b32 x8",
    "v_pk_add_f32 x8 (sgpr)", "v_cndmask_b32_e64 x8", "v_cmp+v_cndmask x4",
    "v_addc_co_u32 x8", "v_cndmask_b32 d!=s x8", "v_max_f32 x8",
    "v_cmp_gt_f32 x8", "s_mov exec + v_add_f32 x4", "v_bfe_u32 x8",
    "v_cmp + 3 v_cndmask x2", "v_cndmask (vcc by v_cmp) x8",
    "[cmp mov nop cnd] x4 /8", "[cmp_e64 cnd_e64] x4", "[cmp cnd add cnd] x2",
    "[cmp cnd addc nop] x2 /8", "[cmp_e64 3 cnd_e64] x2"};

template <int OP>
__global__ __launch_bounds__(256) void k_stream(const float * in, float * out,
                                                Stamp * stamps) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const float c = in[0];
    float a[8];
    v2f p[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = in[1 + i] + (float)threadIdx.x;
        p[i] = (v2f){a[i], in[9 + i]};
        u[i] = threadIdx.x * 8 + i;
    }
    const v2f cc = {c, c};
    unsigned s = 0;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(in[17 + (threadIdx.x & 7)] > 0.012f);
    unsigned long long m2 = mask;   // (an SGPR pair the e64 compares write)
    asm volatile("s_mov_b64 vcc, %0" :: "s"(mask) : "vcc");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
        if (OP == ADD_F32) {
#define T(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            X8(T)
#undef T
        } else if (OP == PK_ADD_F32) {
#define T(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cc));
            X8(T)
#undef T
        } else if (OP == PK_ADD_SGPR) {
#define T(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" \
                          : "+v"(p[i]) : "s"(cc));
            X8(T)
#undef T
        } else if (OP == FMA_F32) {
#define T(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            X8(T)
#undef T
        } else if (OP == PK_FMA_F32) {
#define T(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(cc));
            X8(T)
#undef T
        } else if (OP == MUL_F32) {
#define T(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            X8(T)
#undef T
        } else if (OP == MOV_B32) {
#define T(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            X8(T)
#undef T
        } else if (OP == CNDMASK) {
#define T(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : );
            X8(T)
#undef T
        } else if (OP == CMP_EQ) {
#define T(i) asm volatile("v_cmp_eq_u32 vcc, %0, %1" :: "v"(u[i]), "v"(u[(i + 1) & 7]) : "vcc");
            X8(T)
#undef T
        } else if (OP == ADD_U32) {
#define T(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            X8(T)
#undef T
        } else if (OP == AND_B32) {
#define T(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            X8(T)
#undef T
        } else if (OP == LSHL) {
#define T(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
            X8(T)
#undef T
        } else if (OP == EXP_F32) {
#define T(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            X8(T)
#undef T
        } else if (OP == LOG_F32) {
#define T(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            X8(T)
#undef T
        } else if (OP == RCP_F32) {
#define T(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            X8(T)
#undef T
        } else if (OP == CVT_F32_U32) {
#define T(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
            X8(T)
#undef T
        } else if (OP == MAD_U32_U24) {
#define T(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            X8(T)
#undef T
        } else if (OP == ADD_F32_DEP) {
#define T(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
            X8(T)
#undef T
        } else if (OP == PK_ADD_F32_DEP) {
            // (a dependent packed add needs one wait state: kernels.h)
#define T(i) asm volatile("v_pk_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(p[0]) : "v"(cc));
            X8(T)
#undef T
        } else if (OP == CNDMASK_E64) {
#define T(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "s"(mask));
            X8(T)
#undef T
        } else if (OP == CMP_CNDMASK) {
#define T(i) if (i < 4) asm volatile("v_cmp_eq_u32 vcc, %2, %3\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" \
                          : "+v"(a[i]) : "v"(c), "v"(u[i]), "v"(u[i + 1]) : "vcc");
            X8(T)
#undef T
        } else if (OP == ADDC) {
#define T(i) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(u[i]) :: "vcc");
            X8(T)
#undef T
        } else if (OP == CNDMASK_OTHER_DST) {
#define T(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(c), "v"(p[i].y));
            X8(T)
#undef T
        } else if (OP == MAX_F32) {
#define T(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            X8(T)
#undef T
        } else if (OP == CMP_GT_F32) {
#define T(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(c) : "vcc");
            X8(T)
#undef T
        } else if (OP == EXEC_ADD) {
#define T(i) if (i < 4) asm volatile("s_mov_b64 exec, %1\n\tv_add_f32 %0, %0, %2\n\ts_mov_b64 exec, -1" \
                          : "+v"(a[i]) : "s"(mask), "v"(c));
            X8(T)
#undef T
        } else if (OP == BFE_U32) {
#define T(i) asm volatile("v_bfe_u32 %0, %0, 1, 5" : "+v"(u[i]));
            X8(T)
#undef T
        } else if (OP == CMP_CND3) {
#define T(i) if (i < 2) asm volatile("v_cmp_eq_u32 vcc, %4, %5\n\ts_nop 1\n\t" \
        "v_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc\n\t" \
        "v_cndmask_b32 %2, %2, %3, vcc" \
        : "+v"(a[3 * i]), "+v"(a[3 * i + 1]), "+v"(a[3 * i + 2]) \
        : "v"(c), "v"(u[i]), "v"(u[i + 1]) : "vcc");
            X8(T)
#undef T
        } else if (OP == CND_VALU_VCC) {
            if (it == 0)
                asm volatile("v_cmp_eq_u32 vcc, %0, %1\n\ts_nop 1" :: "v"(u[0]), "v"(u[1]) : "vcc");
#define T(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : );
            X8(T)
#undef T
        } else if (OP == CMP_MOV_CND) {
            // the compiler's own-slot entry: 12 vector instructions per round
#define T(i) if (i < 4) asm volatile("v_cmp_eq_u32 vcc, %2, %3\n\tv_mov_b32 %4, %1\n\ts_nop 0\n\t" \
        "v_cndmask_b32 %0, %0, %4, vcc" \
        : "+v"(a[i]) : "v"(c), "v"(u[i]), "v"(u[i + 1]), "v"(a[i + 4]) : "vcc");
            X8(T)
#undef T
        } else if (OP == CMP64_CND64) {
#define T(i) if (i < 4) asm volatile("v_cmp_eq_u32_e64 %[m], %[x], %[y]\n\ts_nop 1\n\t" \
        "v_cndmask_b32_e64 %[a], %[a], %[c], %[m]" \
        : [a] "+v"(a[i]), [m] "+s"(m2) : [c] "v"(c), [x] "v"(u[i]), [y] "v"(u[i + 1]));
            X8(T)
#undef T
        } else if (OP == CMP_CND_ADD_CND) {
#define T(i) if (i < 2) asm volatile("v_cmp_eq_u32 vcc, %3, %4\n\ts_nop 1\n\t" \
        "v_cndmask_b32 %0, %0, %2, vcc\n\tv_add_f32 %5, %5, %2\n\t" \
        "v_cndmask_b32 %1, %1, %2, vcc" \
        : "+v"(a[2 * i]), "+v"(a[2 * i + 1]) \
        : "v"(c), "v"(u[i]), "v"(u[i + 1]), "v"(a[i + 4]) : "vcc");
            X8(T)
#undef T
        } else if (OP == CMP_CND_ADDC) {
            // the scan's bookkeeping: compare, select, count (6 per round)
#define T(i) if (i < 2) asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\ts_nop 1\n\t" \
        "v_cndmask_b32 %0, %0, %1, vcc\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc" \
        : "+v"(a[i]), "+v"(a[i + 2]), "+v"(u[i]) :: "vcc");
            X8(T)
#undef T
        } else if (OP == CMP64_CND64x3) {
#define T(i) if (i < 2) asm volatile("v_cmp_eq_u32_e64 %[m], %[x], %[y]\n\ts_nop 1\n\t" \
        "v_cndmask_b32_e64 %[a], %[a], %[c], %[m]\n\tv_cndmask_b32_e64 %[b], %[b], %[c], %[m]\n\t" \
        "v_cndmask_b32_e64 %[d], %[d], %[c], %[m]" \
        : [a] "+v"(a[3 * i]), [b] "+v"(a[3 * i + 1]), [d] "+v"(a[3 * i + 2]), [m] "+s"(m2) \
        : [c] "v"(c), [x] "v"(u[i]), [y] "v"(u[i + 1]));
            X8(T)
#undef T
        } else if (OP == READLANE) {
#define T(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(a[i]));
            X8(T)
#undef T
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float r = (float)s;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y + (float)u[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        // HW_ID: wave, SIMD, CU, SH, SE and (XCC_ID) the XCD
        const unsigned long long hw =
            (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (31 << 11))
            | ((unsigned long long)__builtin_amdgcn_s_getreg(
                   (20 << 0) | (3 << 11)) << 32);
        stamps[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] =
            Stamp{t0, t1, r0, r1, hw};
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
    printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Simd { unsigned long long first = ~0ull, last = 0; int waves = 0; };

template <int OP>
static void run(const float * din, float * dout, Stamp * dst, int cus) {
    for (int W : {1, 2, 4, 8}) {
        const int blocks = cus * W;
        const size_t waves = (size_t)blocks * 4;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            k_stream<OP><<<blocks, 256>>>(din, dout, dst);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<Stamp> st(waves);
        CK(hipMemcpy(st.data(), dst, waves * sizeof(Stamp),
                     hipMemcpyDeviceToHost));
        std::map<unsigned long long, Simd> simds;
        std::vector<double> clk;
        for (auto & s : st) {
            const unsigned hw = (unsigned)s.hw;
            // SIMD (5:4), CU (11:8), SH (12), SE (15:13), XCD
            const unsigned long long key =
                ((s.hw >> 32) << 16) | (hw & 0xff30u);
            Simd & m = simds[key];
            m.first = std::min(m.first, s.t0);
            m.last = std::max(m.last, s.t1);
            m.waves += 1;
            clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 100.0);
        }
        std::vector<double> per;
        std::vector<int> res;
        for (auto & kv : simds) {
            // (the "x4" streams are four pairs: eight instructions as well)
            per.push_back((double)(kv.second.last - kv.second.first)
                          / ((double)ITERS * 8 * kv.second.waves));
            res.push_back(kv.second.waves);
        }
        std::sort(per.begin(), per.end());
        std::sort(res.begin(), res.end());
        std::sort(clk.begin(), clk.end());
        printf("%-24s W=%d  cycles/instr/SIMD %5.2f (p10 %5.2f p90 %5.2f; "
               "%zu SIMDs busy, waves per SIMD %d..%d, median %d)  "
               "clock %.0f MHz  kernel %.1f us\n",
               kNames[OP], W, per[per.size() / 2], per[per.size() / 10],
               per[per.size() * 9 / 10], simds.size(), res.front(), res.back(),
               res[res.size() / 2], clk[clk.size() / 2], ms * 1e3);
    }
}

template <int OP>
static void run_all(const float * din, float * dout, Stamp * dst, int cus) {
    run<OP>(din, dout, dst, cus);
    if constexpr (OP + 1 < N_OPS) run_all<OP + 1>(din, dout, dst, cus);
}

int main() {
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    std::vector<float> in(32);
    for (size_t i = 0; i < in.size(); ++i) in[i] = 1e-3f * (float)(i + 1);
    float *din, *dout;
    Stamp * dst;
    const size_t max_threads = (size_t)cus * 8 * 256;
    CK(hipMalloc(&din, in.size() * 4));
    CK(hipMalloc(&dout, max_threads * 4));
    CK(hipMalloc(&dst, max_threads / 64 * sizeof(Stamp)));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    printf("%d CUs\n", cus);
    run_all<0>(din, dout, dst, cus);
    return 0;
}
