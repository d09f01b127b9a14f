// Microbenchmark: what a dependent launch costs on this box, against the
// in-launch alternatives (a last-block ticket, a hand-rolled grid barrier).
// hipcc -O3 --offload-arch=gfx950 -o launch_cost launch_cost.hip
//   chain of N dependent launches on one stream, HIP events around the chain:
//   a) one thread storing a word         b) 256 x 256 threads, a store each
//   c) one workgroup chasing 4 pointers  d) the same with a 2 KB argument block
//   e) a) replayed from a hipGraph of 8 nodes
//   f) ticket: 256 workgroups, the last one to arrive runs a serial tail
//   g) persistent kernel, grid barrier (one counter / per-XCD counters)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
    exit(1); } } while (0)

__global__ void k_one(uint32_t * p, uint32_t v) { *p = v; }
__global__ void k_wide(uint32_t * p, uint32_t v) {
    p[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void k_chase(const uint32_t * next, uint32_t * out, uint32_t start) {
    if (threadIdx.x != 0) return;
    uint32_t i = start;
    for (int s = 0; s < 4; ++s) i = next[i];
    *out = i;
}
struct Big { uint32_t pad[500]; const uint32_t * next; uint32_t * out; };
__global__ void k_chase_big(Big b, uint32_t start) {
    if (threadIdx.x != 0) return;
    uint32_t i = start + b.pad[start & 255];
    for (int s = 0; s < 4; ++s) i = b.next[i];
    *b.out = i;
}
// every workgroup writes a slice, the last to arrive sums the slices
__global__ void k_ticket(uint32_t * slices, unsigned * ticket, uint32_t * out,
                         int tail) {
    __shared__ bool last;
    slices[blockIdx.x * 256 + threadIdx.x] = blockIdx.x + threadIdx.x;
    if (!tail) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
        last = (t % gridDim.x) == gridDim.x - 1;
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!last) return;
    uint32_t s = 0;
    for (unsigned b = 0; b < gridDim.x; ++b) s += slices[b * 256 + threadIdx.x];
    out[threadIdx.x] = s;
}
// grid barrier on one monotonic counter
__device__ __forceinline__ void grid_barrier(unsigned * counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // (bounded: a grid that is not resident at once must not hang the box)
        for (int spin = 0; spin < (1 << 22)
             && __hip_atomic_load(counter, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT) < target; ++spin)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
__global__ void k_barriers(unsigned * counter, uint32_t * data, int rounds) {
    for (int r = 0; r < rounds; ++r) {
        data[(size_t)blockIdx.x * blockDim.x + threadIdx.x] += 1;
        grid_barrier(counter, (unsigned)(r + 1) * gridDim.x);
    }
}
// per-XCD counters: the last arriver of an XCD goes to the top counter
__global__ void k_barriers_xcd(unsigned * xcd_counter, unsigned * top,
                               unsigned * gen, uint32_t * data, int rounds) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;
    __shared__ unsigned s_xcd_blocks;
    for (int r = 0; r < rounds; ++r) {
        data[(size_t)blockIdx.x * blockDim.x + threadIdx.x] += 1;
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // (round-robin placement: gridDim.x / 8 blocks per XCD, checked by
            // the host with a census before this kernel is trusted)
            const unsigned per = gridDim.x / 8;
            const unsigned a = __hip_atomic_fetch_add(
                xcd_counter + xcc * 32, 1u, __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
            if (a % per == per - 1) {
                const unsigned t = __hip_atomic_fetch_add(
                    top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t % 8 == 7)
                    __hip_atomic_store(gen, (unsigned)(r + 1), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int spin = 0; spin < (1 << 22)
                 && __hip_atomic_load(gen, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1);
                 ++spin)
                __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    (void)s_xcd_blocks;
}
__global__ void k_census(unsigned * per_xcd) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;
        atomicAdd(per_xcd + xcc, 1u);
    }
}

template <class F>
static void timed(const char * name, int n, hipStream_t st, F && launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 50; ++i) launch(i);
    CHECK(hipStreamSynchronize(st));
    auto w0 = std::chrono::steady_clock::now();
    CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) launch(i);
    CHECK(hipEventRecord(e1, st));
    auto w1 = std::chrono::steady_clock::now();
    CHECK(hipStreamSynchronize(st));
    auto w2 = std::chrono::steady_clock::now();
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-46s %7.2f us/launch on the device, host enqueue %6.2f, wall %6.2f\n",
           name, ms * 1e3 / n,
           std::chrono::duration<double, std::micro>(w1 - w0).count() / n,
           std::chrono::duration<double, std::micro>(w2 - w0).count() / n);
}

int main() {
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    uint32_t * buf;
    CHECK(hipMalloc(&buf, 64 << 20));
    CHECK(hipMemset(buf, 0, 64 << 20));
    std::vector<uint32_t> next(1 << 20);
    for (size_t i = 0; i < next.size(); ++i)
        next[i] = (uint32_t)((i * 7919u + 12345u) & (next.size() - 1));
    uint32_t * d_next;
    CHECK(hipMalloc(&d_next, next.size() * 4));
    CHECK(hipMemcpy(d_next, next.data(), next.size() * 4, hipMemcpyHostToDevice));
    unsigned * ctr;
    CHECK(hipMalloc(&ctr, 4096));
    CHECK(hipMemset(ctr, 0, 4096));
    const int N = 2000;
    timed("a) <<<1,1>>> one store", N, st, [&](int i) {
        hipLaunchKernelGGL(k_one, dim3(1), dim3(1), 0, st, buf, (uint32_t)i); });
    timed("b) <<<256,256>>> a store per thread", N, st, [&](int i) {
        hipLaunchKernelGGL(k_wide, dim3(256), dim3(256), 0, st, buf, (uint32_t)i); });
    timed("b2) <<<1024,1024>>> a store per thread", N, st, [&](int i) {
        hipLaunchKernelGGL(k_wide, dim3(1024), dim3(1024), 0, st, buf, (uint32_t)i); });
    timed("c) <<<1,64>>> four dependent loads", N, st, [&](int i) {
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, st, d_next, buf,
                           (uint32_t)i * 977u & 0xFFFFFu); });
    Big big;
    for (int i = 0; i < 500; ++i) big.pad[i] = i;
    big.next = d_next;
    big.out = buf;
    timed("d) the same, 2 KB of arguments", N, st, [&](int i) {
        hipLaunchKernelGGL(k_chase_big, dim3(1), dim3(64), 0, st, big,
                           (uint32_t)i * 977u & 0xFFFFu); });
    {   // e) graph of 8 dependent k_one nodes
        hipGraph_t graph;
        hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 8; ++i)
            hipLaunchKernelGGL(k_one, dim3(1), dim3(1), 0, st, buf, (uint32_t)i);
        CHECK(hipStreamEndCapture(st, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        timed("e) graph of 8 x a) (per graph launch)", 250, st, [&](int) {
            CHECK(hipGraphLaunch(exec, st)); });
    }
    timed("f0) <<<256,256>>> slices, no tail", N, st, [&](int) {
        hipLaunchKernelGGL(k_ticket, dim3(256), dim3(256), 0, st, buf, ctr,
                           buf + (1 << 20), 0); });
    timed("f1) the same + ticket, the last block sums", N, st, [&](int) {
        hipLaunchKernelGGL(k_ticket, dim3(256), dim3(256), 0, st, buf, ctr,
                           buf + (1 << 20), 1); });
    timed("f2) f0 then a <<<1,256>>> launch for the sum", N, st, [&](int) {
        hipLaunchKernelGGL(k_ticket, dim3(256), dim3(256), 0, st, buf, ctr,
                           buf + (1 << 20), 0);
        hipLaunchKernelGGL(k_wide, dim3(1), dim3(256), 0, st, buf + (2 << 20), 1u); });
    for (int wgs : {256, 512, 1024}) {
        for (int threads : {256, 1024}) {
            if (wgs * threads > 256 * 2048) continue;
            const int rounds = 200;
            CHECK(hipMemset(ctr, 0, 4096));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(k_barriers, dim3(wgs), dim3(threads), 0, st, ctr,
                               buf, 1);
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemset(ctr, 0, 4096));
            CHECK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_barriers, dim3(wgs), dim3(threads), 0, st, ctr,
                               buf, rounds);
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("g) grid barrier, one counter, %4d x %4d: %6.2f us per barrier\n",
                   wgs, threads, ms * 1e3 / rounds);
            // per-XCD form (only when the census says round-robin placement)
            unsigned census[8] = {0};
            CHECK(hipMemset(ctr, 0, 4096));
            hipLaunchKernelGGL(k_census, dim3(wgs), dim3(threads), 0, st, ctr);
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(census, ctr, 32, hipMemcpyDeviceToHost));
            bool even = true;
            for (int x = 0; x < 8; ++x) even = even && census[x] == (unsigned)wgs / 8;
            if (!even) { printf("   (census uneven: per-XCD form skipped)\n"); continue; }
            CHECK(hipMemset(ctr, 0, 4096));
            CHECK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_barriers_xcd, dim3(wgs), dim3(threads), 0, st,
                               ctr, ctr + 512, ctr + 768, buf, rounds);
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("g) grid barrier, per-XCD counters, %4d x %4d: %6.2f us per barrier\n",
                   wgs, threads, ms * 1e3 / rounds);
        }
    }
    return 0;
}
