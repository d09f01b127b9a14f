// After the draw: applying a batch's moves (k_vs_apply, k_vs_reduce, the
// generic apply kernels), merged float statistics, the ordered replay with its
// counting sort, the initial load, validate.  Part of kernels.h.
#pragma once

namespace dist {

// ---------------------------------------------------------------------------
// applying a batch of moves

__global__ void k_apply_moves(SweepParams P, StatImage img,
                              const uint32_t * __restrict__ p2g,
                              uint32_t * __restrict__ assign) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = P.row_end - P.row_begin;
    if (b >= n) return;
    const size_t row = P.row_begin + b;
    const uint32_t go = P.old_packed[b], gn = P.new_packed[b];
    if (assign) assign[row] = p2g[gn];
    if (go == gn) return;
    atomicAdd(&img.counts[go], -1);
    atomicAdd(&img.counts[gn], 1);
    for (int f = 0; f < P.F; ++f) {
        const SlaveView & s = P.feat[f];
        const uint32_t x = P.values[f][row];
        switch (s.kind) {
        case DIST_DD:
        case DIST_DPD:
            atomicAdd(&img.i0[f][go], -1);
            atomicAdd(&img.i0[f][gn], 1);
            if (x != DIST_DPD_OTHER) {
                atomicAdd(&img.cnt[f][(size_t)go * s.dim + x], -1);
                atomicAdd(&img.cnt[f][(size_t)gn * s.dim + x], 1);
            }
            break;
        case DIST_BB:
            atomicAdd(x ? &img.i0[f][go] : &img.i1[f][go], -1);
            atomicAdd(x ? &img.i0[f][gn] : &img.i1[f][gn], 1);
            break;
        case DIST_GP:
        case DIST_BNB:
            atomicAdd(&img.i0[f][go], -1);
            atomicAdd(&img.i0[f][gn], 1);
            atomicAdd(&img.i1[f][go], -(int32_t)x);
            atomicAdd(&img.i1[f][gn], (int32_t)x);
            break;
        default:
            break;
        }
    }
}

// The same with the per-group totals (sizes, and each feature's two integer
// statistics) summed in LDS first: a workgroup takes kApplyLdsRows rows and
// leaves with one global atomic per total it changed, instead of six per
// moved row all aimed at the same K addresses (C3: 443 us per 10^6 rows).
// Categorical cells (k, x) are sparse and keep their direct atomics.
// Integer additions: the result does not depend on the order.
constexpr int kApplyLdsBlock = 1024;
constexpr int kApplyLdsRows = 8192;
__global__ __launch_bounds__(kApplyLdsBlock) void k_apply_moves_lds(
        SweepParams P, StatImage img, const uint32_t * __restrict__ p2g,
        uint32_t * __restrict__ assign) {
    extern __shared__ int am_lds[];   // [1 + 2 F][K]
    const int K = sweep_K(P);
    const int words = (1 + 2 * P.F) * K;
    for (int i = threadIdx.x; i < words; i += kApplyLdsBlock) am_lds[i] = 0;
    __syncthreads();
    const size_t n = P.row_end - P.row_begin;
    const size_t begin = (size_t)blockIdx.x * kApplyLdsRows;
    const size_t end = begin + kApplyLdsRows < n ? begin + kApplyLdsRows : n;
    for (size_t b = begin + threadIdx.x; b < end; b += kApplyLdsBlock) {
        const size_t row = P.row_begin + b;
        const uint32_t go = P.old_packed[b], gn = P.new_packed[b];
        if (assign) assign[row] = p2g[gn];
        if (go == gn) continue;
        atomicAdd(&am_lds[go], -1);
        atomicAdd(&am_lds[gn], 1);
        for (int f = 0; f < P.F; ++f) {
            const SlaveView & s = P.feat[f];
            const uint32_t x = P.values[f][row];
            int * t0 = am_lds + (1 + 2 * f) * K;
            int * t1 = t0 + K;
            switch (s.kind) {
            case DIST_DD:
            case DIST_DPD:
                atomicAdd(&t0[go], -1);
                atomicAdd(&t0[gn], 1);
                if (x != DIST_DPD_OTHER) {
                    atomicAdd(&img.cnt[f][(size_t)go * s.dim + x], -1);
                    atomicAdd(&img.cnt[f][(size_t)gn * s.dim + x], 1);
                }
                break;
            case DIST_BB:
                atomicAdd(x ? &t0[go] : &t1[go], -1);
                atomicAdd(x ? &t0[gn] : &t1[gn], 1);
                break;
            case DIST_GP:
            case DIST_BNB:
                atomicAdd(&t0[go], -1);
                atomicAdd(&t0[gn], 1);
                atomicAdd(&t1[go], -(int32_t)x);
                atomicAdd(&t1[gn], (int32_t)x);
                break;
            default:
                break;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += kApplyLdsBlock) {
        const int v = am_lds[i];
        if (v == 0) continue;
        const int which = i / K, k = i - which * K;
        if (which == 0) {
            atomicAdd(&img.counts[k], v);
        } else {
            const int f = (which - 1) >> 1;
            atomicAdd(((which - 1) & 1) ? &img.i1[f][k] : &img.i0[f][k], v);
        }
    }
}

// The same again for feature lists whose WHOLE integer image -- group sizes,
// per-group totals AND categorical cells, in the stat-word layout
//   counts[K] | per feature: i0[K] i1[K] cnt[K][dim]
// -- fits a workgroup's LDS (mixed rows with small categoricals: DD(16) +
// DD(4) + BB + GP + NICH at K = 1024 is 31 K words): no global atomic at
// all.  A workgroup sums its rows' image in LDS and leaves it as one row of a
// staging matrix (plain coalesced stores); k_stage_reduce adds the rows up
// per word.  (Before: four global atomics per moved row on the cells, 220 us
// per 10^6 mixed rows.)
struct StageLayout {
    int K;
    int off_i0[kMaxF], off_i1[kMaxF], off_cnt[kMaxF];   // word offsets
    int dim[kMaxF];
    int words;
};
__global__ __launch_bounds__(kApplyLdsBlock) void k_apply_moves_stage(
        SweepParams P, StageLayout L, int32_t * __restrict__ stage,
        const uint32_t * __restrict__ p2g, uint32_t * __restrict__ assign) {
    extern __shared__ int am_lds[];   // [L.words]
    for (int i = threadIdx.x; i < L.words; i += kApplyLdsBlock) am_lds[i] = 0;
    __syncthreads();
    const size_t n = P.row_end - P.row_begin;
    const size_t begin = (size_t)blockIdx.x * kApplyLdsRows;
    const size_t end = begin + kApplyLdsRows < n ? begin + kApplyLdsRows : n;
    for (size_t b = begin + threadIdx.x; b < end; b += kApplyLdsBlock) {
        const size_t row = P.row_begin + b;
        const uint32_t go = P.old_packed[b], gn = P.new_packed[b];
        if (assign) assign[row] = p2g[gn];
        if (go == gn) continue;
        atomicAdd(&am_lds[go], -1);
        atomicAdd(&am_lds[gn], 1);
        for (int f = 0; f < P.F; ++f) {
            const int kind = P.feat[f].kind;
            const uint32_t x = P.values[f][row];
            int * t0 = am_lds + L.off_i0[f];
            int * t1 = am_lds + L.off_i1[f];
            switch (kind) {
            case DIST_DD:
            case DIST_DPD:
                atomicAdd(&t0[go], -1);
                atomicAdd(&t0[gn], 1);
                if (x != DIST_DPD_OTHER) {
                    int * cnt = am_lds + L.off_cnt[f];
                    atomicAdd(&cnt[(size_t)go * L.dim[f] + x], -1);
                    atomicAdd(&cnt[(size_t)gn * L.dim[f] + x], 1);
                }
                break;
            case DIST_BB:
                atomicAdd(x ? &t0[go] : &t1[go], -1);
                atomicAdd(x ? &t0[gn] : &t1[gn], 1);
                break;
            case DIST_GP:
            case DIST_BNB:
                atomicAdd(&t0[go], -1);
                atomicAdd(&t0[gn], 1);
                atomicAdd(&t1[go], -(int32_t)x);
                atomicAdd(&t1[gn], (int32_t)x);
                break;
            default:
                break;
            }
        }
    }
    __syncthreads();
    int32_t * out = stage + (size_t)blockIdx.x * L.words;
    for (int i = threadIdx.x; i < L.words; i += kApplyLdsBlock)
        out[i] = am_lds[i];
}
// stats += delta (after the all-reduce): the delta image is contiguous, the
// live statistics are separate arrays; one launch walks all segments
struct WordSegments {
    int n;
    int32_t * dst[1 + 3 * kMaxF];
    unsigned long long end[1 + 3 * kMaxF];   // running end offset in the image
};
// The header of the ranks' exchange (dist_gibbs_sweep_sharded).  In front of
// its delta image every rank puts two 12-bit signatures of where it believes
// the ranks' run stands -- run serial, batch index, groups exchanged, tiling --
// each as (x, x * x).  After the sum over W ranks, W * sum(x^2) == (sum x)^2
// holds if and only if every rank put the same x (Cauchy-Schwarz; exact:
// 64 * 4095^2 < 2^31).  A rank that changed something between two passes, or
// whose run stands elsewhere, therefore shows up in the very collective its
// peers issue, without a word of its own and without a host round trip: the
// kernel that consumes the sum raises `fault` in pinned host memory, and the
// engine's next entry point fails with "ranks diverged".  (Collectives of
// DIFFERENT sizes cannot be told this way -- RCCL hangs on them; the host
// transport, comm.h, checks sizes too.)
constexpr int kCommHeaderWords = 4;
struct CommCheck {
    int32_t * header;      // null: no check
    int world;
    unsigned * fault;      // pinned host word
    unsigned tag;          // what to raise
    int32_t next[kCommHeaderWords];   // left behind for the next batch
};
__global__ void k_comm_header(int32_t * header, int32_t a, int32_t b) {
    header[0] = a;
    header[1] = a * a;
    header[2] = b;
    header[3] = b * b;
}
// clear: leave the image zeroed for the next batch (the library's own
// exchange buffer is never memset again).  host_pairs: segment 0 is the group
// sizes; their new values go to pinned host memory with the batch's ticket
// (see k_vs_reduce).
__global__ void k_add_words(WordSegments seg, int32_t * __restrict__ src,
                            size_t total, int clear,
                            unsigned long long * host_pairs,
                            unsigned int seq, CommCheck chk) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && chk.header) {
        const long long w = chk.world;
        const long long s1 = chk.header[0], q1 = chk.header[1],
                        s2 = chk.header[2], q2 = chk.header[3];
        if (w * q1 != s1 * s1 || w * q2 != s2 * s2)
            __hip_atomic_store(chk.fault, chk.tag, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int j = 0; j < kCommHeaderWords; ++j) chk.header[j] = chk.next[j];
    }
    if (i >= total) return;
    int j = 0;
    while (i >= seg.end[j]) ++j;
    const size_t begin = j ? seg.end[j - 1] : 0;
    const int32_t d = src[i];
    if (j == 0 && host_pairs) {
        const int32_t now = seg.dst[0][i] + d;
        if (d) seg.dst[0][i] = now;
        host_pairs[i] = ((unsigned long long)seq << 32) | (uint32_t)now;
    } else if (d) {
        seg.dst[j][i - begin] += d;
    }
    if (clear && d) src[i] = 0;
}

// value-partitioned ranks (dist_gibbs_partition_by_value): which values have
// rows here, and the cells of the values a rank owns (zero elsewhere) -- the
// summands of dist_gibbs_gather_cells
__global__ void k_value_presence(const uint32_t * __restrict__ values,
                                 size_t n, int dim, int32_t * has) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t x = values[i];
        if (x < (uint32_t)dim && has[x] == 0) has[x] = 1;   // (benign race)
    }
}
__global__ void k_owned_cells(const int32_t * __restrict__ cnt,
                              const int32_t * __restrict__ owned, size_t cells,
                              int dim, int32_t * out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    out[i] = owned[i % (size_t)dim] ? cnt[i] : 0;
}

// dst += the staged rows of k_apply_moves_stage, summed per word
__global__ void k_stage_reduce(WordSegments seg,
                               const int32_t * __restrict__ stage, int rows,
                               int words) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words) return;
    int32_t d = 0;
    int r = 0;
    for (; r + 8 <= rows; r += 8) {   // eight loads in flight
        int32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = stage[(size_t)(r + q) * words + i];
#pragma unroll
        for (int q = 0; q < 8; ++q) d += v[q];
    }
    for (; r < rows; ++r) d += stage[(size_t)r * words + i];
    if (d == 0) return;
    int j = 0;
    while ((unsigned long long)i >= seg.end[j]) ++j;
    const size_t begin = j ? seg.end[j - 1] : 0;
    seg.dst[j][i - begin] += d;
}

// ---- merged float statistics (option "float_stats" = 1: opt-in,
// tolerance-level; the ordered replay below stays the default) --------------
// The order-dependent statistics of a batch as SUMS in binary64 -- per group
// NICH: the change of the count, of sum x and of sum x^2; GP: of log_prod --
// which do add over rows, workgroups and ranks.  The group's new (count,
// mean, count_times_variance) follows from its old ones and the sums by the
// textbook identities; it equals what nich.hpp:125-165's running updates give
// to binary32 rounding (the tests bound the difference), not bit for bit.
// Worth it where the ordered replay costs too much: a chain of ~2 B / K
// dependent Welford steps per group and batch, and in a multi-rank run every
// rank replaying every rank's rows.
struct MergeLayout {
    int F;
    int kind[kMaxF];          // DIST_NICH, DIST_GP or -1
    int off[kMaxF];           // first double of the feature's block
    int words;                // doubles per image
    int K;
};
__global__ __launch_bounds__(kApplyLdsBlock) void k_merge_float_moves(
        SweepParams P, MergeLayout L, const uint32_t * __restrict__ old_slot,
        const uint32_t * __restrict__ new_slot, double * __restrict__ stage) {
    extern __shared__ double mf_lds[];   // [L.words]
    for (int i = threadIdx.x; i < L.words; i += kApplyLdsBlock) mf_lds[i] = 0.0;
    __syncthreads();
    const size_t n = P.row_end - P.row_begin;
    const size_t begin = (size_t)blockIdx.x * kApplyLdsRows;
    const size_t end = begin + kApplyLdsRows < n ? begin + kApplyLdsRows : n;
    for (size_t b = begin + threadIdx.x; b < end; b += kApplyLdsBlock) {
        const uint32_t go = old_slot[b], gn = new_slot[b];
        if (go == gn || go == 0xFFFFFFFFu) continue;   // (or padding)
        const size_t row = P.row_begin + b;
        for (int f = 0; f < L.F; ++f) {
            if (L.kind[f] < 0) continue;
            double * d = mf_lds + L.off[f];
            const uint32_t w = P.values[f][row];
            if (L.kind[f] == DIST_NICH) {
                const double x = (double)u2f(w);
                atomicAdd(&d[go], -1.0);
                atomicAdd(&d[gn], 1.0);
                atomicAdd(&d[L.K + go], -x);
                atomicAdd(&d[L.K + gn], x);
                atomicAdd(&d[2 * L.K + go], -x * x);
                atomicAdd(&d[2 * L.K + gn], x * x);
            } else {   // GammaPoisson's log_prod (gp.hpp:115,134)
                const double lf = (double)fast_log_factorial(w);
                atomicAdd(&d[go], -lf);
                atomicAdd(&d[gn], lf);
            }
        }
    }
    __syncthreads();
    double * out = stage + (size_t)blockIdx.x * L.words;
    for (int i = threadIdx.x; i < L.words; i += kApplyLdsBlock)
        out[i] = mf_lds[i];
}
// the staged rows summed per word, in row order (a fixed order: the same
// partial sums give the same image)
__global__ void k_merge_float_reduce(const double * __restrict__ stage,
                                     int rows, int words,
                                     double * __restrict__ image) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words) return;
    double d = 0.0;
    int r = 0;
    for (; r + 8 <= rows; r += 8) {   // eight loads in flight, added in order
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = stage[(size_t)(r + q) * words + i];
#pragma unroll
        for (int q = 0; q < 8; ++q) d += v[q];
    }
    for (; r < rows; ++r) d += stage[(size_t)r * words + i];
    image[i] = d;
}
// a replica's float statistics AS such an image (count, sum x, sum x^2;
// log_prod): the all-reduce of the ranks' images, applied with `reset` (the
// old statistics taken as zero), is the statistics of all rows
__global__ void k_merge_float_export(SweepParams P, MergeLayout L,
                                     double * __restrict__ image) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= L.K) return;
    for (int f = 0; f < L.F; ++f) {
        if (L.kind[f] < 0) continue;
        const SlaveView & v = P.feat[f];
        double * d = image + L.off[f];
        if (L.kind[f] == DIST_NICH) {
            const double n = (double)v.i0[k], mean = (double)v.f0[k];
            d[k] = n;
            d[L.K + k] = n * mean;
            d[2 * L.K + k] = (double)v.f1[k] + n * mean * mean;
        } else {
            d[k] = (double)v.f0[k];
        }
    }
}
// the groups' statistics from their old ones and the (all-reduced) image
__global__ void k_merge_float_apply(SweepParams P, MergeLayout L,
                                    const double * __restrict__ image,
                                    int reset) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= L.K) return;
    for (int f = 0; f < L.F; ++f) {
        if (L.kind[f] < 0) continue;
        const SlaveView & v = P.feat[f];
        const double * d = image + L.off[f];
        if (L.kind[f] == DIST_NICH) {
            const double dn = d[k], dx = d[L.K + k], dxx = d[2 * L.K + k];
            if (!reset && dn == 0.0 && dx == 0.0 && dxx == 0.0) continue;
            const double n0 = reset ? 0.0 : (double)v.i0[k];
            const double mean0 = reset ? 0.0 : (double)v.f0[k];
            const double ctv0 = reset ? 0.0 : (double)v.f1[k];
            const double n1 = n0 + dn;
            const double s1 = n0 * mean0 + dx;
            const double s2 = ctv0 + n0 * mean0 * mean0 + dxx;
            double mean1 = 0.0, ctv1 = 0.0;
            if (n1 >= 1.0) mean1 = s1 / n1;
            if (n1 >= 2.0) {   // nich.hpp:159-163: no variance below two
                ctv1 = s2 - n1 * mean1 * mean1;
                if (ctv1 < 0.0) ctv1 = 0.0;
            }
            v.i0[k] = (int32_t)n1;
            v.f0[k] = (float)mean1;
            v.f1[k] = (float)ctv1;
        } else if (reset || d[k] != 0.0) {
            v.f0[k] = (float)((reset ? 0.0 : (double)v.f0[k]) + d[k]);
        }
    }
}

// Float statistics (NICH count/mean/ctv, GP log_prod) depend on update order
// (nich.hpp:125-165 is a running Welford update), so they are replayed per
// group in row order -- the order the sequential chain would apply them in.
// ---- ordered replay through a stable sort of the events by group ----------
// events of batch row b: 2b = "remove from old[b]", 2b+1 = "add to new[b]";
// sorted stably by group they are, per group, in row order with the removal
// of a row ahead of its own addition.
// A slot of 0xFFFFFFFF marks padding (ragged gathers of the multi-rank
// exchange): its events get key `n_groups`, a segment nobody replays.
__global__ void k_replay_events(const uint32_t * __restrict__ old_packed,
                                const uint32_t * __restrict__ new_packed,
                                size_t n_rows, uint32_t n_groups,
                                uint32_t * __restrict__ keys,
                                uint32_t * __restrict__ vals) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_rows) return;
    const uint32_t g2 = new_packed[b];
    const bool pad = g2 == 0xFFFFFFFFu;
    if (old_packed) {
        keys[2 * b] = pad ? n_groups : old_packed[b];
        vals[2 * b] = (uint32_t)(2 * b);
        keys[2 * b + 1] = pad ? n_groups : g2;
        vals[2 * b + 1] = (uint32_t)(2 * b + 1);
    } else {   // initial load: additions only
        keys[b] = pad ? n_groups : g2;
        vals[b] = (uint32_t)(2 * b + 1);
    }
}

// the order-dependent statistics back to Group::init (before a replay of
// the whole data set): all of NICH's, GP's log_prod
__global__ void k_zero_ordered_stats(SlaveView s, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    s.f0[k] = 0.f;
    if (s.kind == DIST_NICH) { s.i0[k] = 0; s.f1[k] = 0.f; }
}

// ---------------------------------------------------------------------------
// The batch's statistic events sorted STABLY by group for the ordered replay
// (k_replay_sorted): event 2b removes row b from its old group, event 2b + 1
// adds it to its new one (additions only: one event per row), and every
// group's events must come out in row order (nich.hpp:125-165, gp.hpp:109-135
// are order-dependent).  A counting sort on the group key in three launches --
// histogram, scan, scatter -- that reads the moves directly (no key / value
// arrays, no segment-bound pass: a group's events are [base[k], base[k + 1]));
// it replaced a library radix sort (five launches and eight fills).
// Keys: 0 .. n_keys - 2 the groups, n_keys - 1 the padding rows.
constexpr int kCsBlock = 256;            // threads per workgroup
constexpr int kCsEvents = 4096;          // events per workgroup: 1024 per wave
constexpr int kCsMaxKeys = 7000;         // (the scatter keeps 5 x n_keys in LDS)
__device__ __forceinline__ uint32_t cs_event_key(
        const uint32_t * __restrict__ old_packed,
        const uint32_t * __restrict__ new_packed, size_t e, uint32_t pad_key) {
    if (old_packed == nullptr) {   // additions only: event e adds row e
        const uint32_t g = new_packed[e];
        return g == 0xFFFFFFFFu ? pad_key : g;
    }
    const uint32_t g2 = new_packed[e >> 1];
    if (g2 == 0xFFFFFFFFu) return pad_key;
    return (e & 1) ? g2 : old_packed[e >> 1];
}
__global__ __launch_bounds__(kCsBlock) void k_cs_hist(
        const uint32_t * __restrict__ old_packed,
        const uint32_t * __restrict__ new_packed, size_t n_ev, int n_keys,
        uint32_t * __restrict__ hist) {
    extern __shared__ uint32_t cs_lds[];   // [n_keys]
    for (int k = threadIdx.x; k < n_keys; k += kCsBlock) cs_lds[k] = 0;
    __syncthreads();
    const size_t begin = (size_t)blockIdx.x * kCsEvents;
    for (int i = threadIdx.x; i < kCsEvents; i += kCsBlock) {
        const size_t e = begin + i;
        if (e < n_ev)
            atomicAdd(&cs_lds[cs_event_key(old_packed, new_packed, e,
                                           (uint32_t)n_keys - 1)], 1u);
    }
    __syncthreads();
    uint32_t * row = hist + (size_t)blockIdx.x * n_keys;
    for (int k = threadIdx.x; k < n_keys; k += kCsBlock) row[k] = cs_lds[k];
}
// hist[b][k] becomes the events of key k in workgroups before b, total[k]
// their number in all: a thread per key walks its column (rows coalesce
// across the threads)
__global__ __launch_bounds__(kCsBlock) void k_cs_scan(
        uint32_t * __restrict__ hist, int blocks, int n_keys,
        uint32_t * __restrict__ total) {
    const int k = blockIdx.x * kCsBlock + threadIdx.x;
    if (k >= n_keys) return;
    uint32_t run = 0;
    constexpr int U = 8;   // loads in flight per thread
    int b = 0;
    for (; b + U <= blocks; b += U) {
        uint32_t v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = hist[(size_t)(b + q) * n_keys + k];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            hist[(size_t)(b + q) * n_keys + k] = run;
            run += v[q];
        }
    }
    for (; b < blocks; ++b) {
        const uint32_t v = hist[(size_t)b * n_keys + k];
        hist[(size_t)b * n_keys + k] = run;
        run += v;
    }
    total[k] = run;
}
// every event's id to its place.  base[k] = the events of keys before k
// (every workgroup scans the totals for itself; workgroup 0 leaves base[] for
// k_replay_sorted: a group's events are [base[k], base[k + 1])).  A wave
// walks its 1024 events in order, 64 at a time: a lane takes its place with
// an LDS atomic on its key's counter; where several lanes of the 64 share a
// key -- the lane that drew the lowest place sees the counter move by more
// than one -- that key's lanes take consecutive places in LANE order instead
// (the sort must be stable: a group's events replay in row order).
__global__ __launch_bounds__(kCsBlock) void k_cs_scatter(
        const uint32_t * __restrict__ old_packed,
        const uint32_t * __restrict__ new_packed, size_t n_ev, int n_keys,
        const uint32_t * __restrict__ hist,
        const uint32_t * __restrict__ total, uint32_t * __restrict__ base_out,
        uint32_t * __restrict__ events_out) {
    extern __shared__ uint32_t cs_lds[];   // [n_keys] base | [waves][n_keys]
    __shared__ uint32_t s_part[kCsBlock / 64];
    __shared__ uint32_t s_carry;
    constexpr int kWaves = kCsBlock / 64;
    constexpr int kPerWave = kCsEvents / kWaves;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t pad_key = (uint32_t)n_keys - 1;
    uint32_t * base = cs_lds;
    uint32_t * places = cs_lds + n_keys;
    // base[]: exclusive scan of the totals, a stretch of kCsBlock keys at a time
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n_keys; k0 += kCsBlock) {
        const int k = k0 + threadIdx.x;
        const uint32_t t = k < n_keys ? total[k] : 0u;
        uint32_t incl = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        uint32_t run = s_carry + incl - t;
        for (int w = 0; w < wave; ++w) run += s_part[w];
        if (k < n_keys) base[k] = run;
        __syncthreads();
        if (threadIdx.x == kCsBlock - 1) s_carry = run + t;
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (int k = threadIdx.x; k < n_keys; k += kCsBlock) base_out[k] = base[k];
        if (threadIdx.x == 0) base_out[n_keys] = s_carry;
    }
    // the waves' own counts, then their first places: the keys before, the
    // workgroups before, the waves before
    for (int i = threadIdx.x; i < kWaves * n_keys; i += kCsBlock) places[i] = 0;
    __syncthreads();
    const size_t begin = (size_t)blockIdx.x * kCsEvents + (size_t)wave * kPerWave;
    uint32_t * mine = places + (size_t)wave * n_keys;
    for (int i = lane; i < kPerWave; i += 64) {
        const size_t e = begin + i;
        if (e < n_ev)
            atomicAdd(&mine[cs_event_key(old_packed, new_packed, e, pad_key)], 1u);
    }
    __syncthreads();
    const uint32_t * row = hist + (size_t)blockIdx.x * n_keys;
    for (int k = threadIdx.x; k < n_keys; k += kCsBlock) {
        uint32_t run = base[k] + row[k];
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const uint32_t c = places[(size_t)w * n_keys + k];
            places[(size_t)w * n_keys + k] = run;
            run += c;
        }
    }
    __syncthreads();
    for (int i0 = 0; i0 < kPerWave; i0 += 64) {
        const size_t e = begin + i0 + lane;
        const bool active = e < n_ev;
        const uint32_t key =
            active ? cs_event_key(old_packed, new_packed, e, pad_key) : 0u;
        // (additions only: the event id of row e is 2 e + 1, k_replay_sorted's
        // convention)
        const uint32_t id = old_packed ? (uint32_t)e : (uint32_t)(2 * e + 1);
        uint32_t place = 0, after = 0;
        if (active) place = atomicAdd(&mine[key], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (active) after = mine[key];
        // keys that several of the 64 lanes hold: their lane of lowest place
        // sees the counter two or more ahead of it
        unsigned long long todo =
            __builtin_amdgcn_ballot_w64(active && after - place >= 2u);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t kl =
                (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
            const bool same = active && key == kl;
            const unsigned long long group = __builtin_amdgcn_ballot_w64(same);
            if (same) {
                const unsigned long long lower = group & ((1ull << lane) - 1ull);
                place = after - (uint32_t)__popcll(group)
                        + (uint32_t)__popcll(lower);
            }
            todo &= ~group;
        }
        if (active) events_out[place] = id;
        __builtin_amdgcn_wave_barrier();
    }
}

// first/one-past-last position of every group's events in the sorted list
__global__ void k_replay_bounds(const uint32_t * __restrict__ keys_sorted,
                                size_t n, uint32_t * __restrict__ seg_begin,
                                uint32_t * __restrict__ seg_end) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys_sorted[i];
    if (i == 0 || keys_sorted[i - 1] != k) seg_begin[k] = (uint32_t)i;
    if (i + 1 == n || keys_sorted[i + 1] != k) seg_end[k] = (uint32_t)(i + 1);
}

// one wave per group: 64 events are fetched at a time (coalesced ids, gathered
// values) and then applied one after the other, every lane computing the same
// scalar update (nich.hpp:125-165 / gp.hpp:109-135)
// (blockIdx.y: the ordered feature -- they replay side by side, the longest
// chain sets the launch's time)
struct ReplayFeatures {
    int n;
    SlaveView s[kMaxF];
    const uint32_t * values[kMaxF];
};
__global__ __launch_bounds__(64) void k_replay_sorted(
        ReplayFeatures R, size_t row_begin,
        const uint32_t * __restrict__ vals_sorted,
        const uint32_t * __restrict__ seg_begin,
        const uint32_t * __restrict__ seg_end) {
    const SlaveView & s = R.s[blockIdx.y];
    const uint32_t * __restrict__ values = R.values[blockIdx.y];
    const int k = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t lo = seg_begin[k], hi = seg_end[k];
    if (lo >= hi) return;
    Stats fl = load_stats(s, k);
    for (uint32_t base = lo; base < hi; base += 64) {
        const uint32_t i = base + lane;
        uint32_t e = 0, x = 0;
        if (i < hi) {
            e = vals_sorted[i];
            x = values[row_begin + (e >> 1)];
        }
        const int cnt = (int)min(64u, hi - base);
        if (s.kind == DIST_GP) {
            // only log_prod is order-dependent (gp.hpp:115,134): the terms are
            // looked up by all lanes at once, the running sum stays in order
            // (x - t == x + (-t) exactly: the sign goes into the term, and a
            // full block of 64 events is 64 lane reads and 64 adds, no loop)
            const float lf = fast_log_factorial(x);
            const float term = (e & 1u) ? lf : -lf;
            if (cnt == 64) {
#pragma unroll
                for (int j = 0; j < 64; ++j)
                    fl.f0 += u2f((uint32_t)__builtin_amdgcn_readlane(
                        (int)f2u(term), j));
            } else {
                for (int j = 0; j < cnt; ++j)
                    fl.f0 += u2f((uint32_t)__builtin_amdgcn_readlane(
                        (int)f2u(term), j));
            }
            continue;
        }
        if (s.kind == DIST_NICH) {
            // (the kind spelled out: a switch on it per event is a dozen
            // branches for a wave that runs alone)
            int j = 0;
            for (; j + 8 <= cnt; j += 8) {
                uint32_t ej[8], xj[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ej[u] = __builtin_amdgcn_readlane((int)e, j + u);
                    xj[u] = __builtin_amdgcn_readlane((int)x, j + u);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (ej[u] & 1u) stats_add(DIST_NICH, fl, xj[u]);
                    else stats_remove(DIST_NICH, fl, xj[u]);
                }
            }
            for (; j < cnt; ++j) {
                const uint32_t ej = __builtin_amdgcn_readlane((int)e, j);
                const uint32_t xj = __builtin_amdgcn_readlane((int)x, j);
                if (ej & 1u) stats_add(DIST_NICH, fl, xj);
                else stats_remove(DIST_NICH, fl, xj);
            }
            continue;
        }
        for (int j = 0; j < cnt; ++j) {
            const uint32_t ej = __builtin_amdgcn_readlane((int)e, j);
            const uint32_t xj = __builtin_amdgcn_readlane((int)x, j);
            if (ej & 1u) stats_add(s.kind, fl, xj);
            else stats_remove(s.kind, fl, xj);
        }
    }
    if (lane == 0) {
        s.f0[k] = fl.f0;
        s.f1[k] = fl.f1;
        if (s.kind == DIST_NICH) s.i0[k] = fl.i0;
    }
}

// initial load: integer statistics of all rows by atomics
__global__ void k_load_counts(SweepParams P, StatImage img,
                              const uint32_t * __restrict__ assign_packed) {
    const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= P.row_end) return;
    const uint32_t g = assign_packed[row];
    atomicAdd(&img.counts[g], 1);
    for (int f = 0; f < P.F; ++f) {
        const SlaveView & s = P.feat[f];
        const uint32_t x = P.values[f][row];
        switch (s.kind) {
        case DIST_DD:
        case DIST_DPD:
            atomicAdd(&img.i0[f][g], 1);
            atomicAdd(&img.cnt[f][(size_t)g * s.dim + x], 1);
            break;
        case DIST_BB:
            atomicAdd(x ? &img.i0[f][g] : &img.i1[f][g], 1);
            break;
        case DIST_GP:
        case DIST_BNB:
            atomicAdd(&img.i0[f][g], 1);
            atomicAdd(&img.i1[f][g], (int32_t)x);
            break;
        default:
            break;
        }
    }
}

__global__ void k_packed_to_global(const uint32_t * __restrict__ packed,
                                   const uint32_t * __restrict__ p2g,
                                   uint32_t * __restrict__ global, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) global[i] = p2g[packed[i]];
}

// ---------------------------------------------------------------------------
// validate (mixture.hpp:152-163,440-442 and what those asserts stand for):
// the statistics recounted from the rows' assignments against the live ones.
// k_validate_rows turns every row's global id into its packed index (an id
// that is not live is reported, lowest row first) and counts the row into a
// recount image; k_validate_compare reports the lowest (feature, group, cell)
// at which the live image differs.

enum ValidateCode {
    VALIDATE_OK = 0,
    VALIDATE_DEAD_ID = 1,       // group = row, detail = the id it carries
    VALIDATE_VALUE_RANGE = 2,   // group = row, detail = the value
    VALIDATE_GROUP_SIZE = 3,    // counts[k] != rows assigned to k
    VALIDATE_STAT0 = 4,         // i0[k]: count_sum / heads / count
    VALIDATE_STAT1 = 5,         // i1[k]: tails / sum
    VALIDATE_CELL = 6,          // cnt[k][detail]
    VALIDATE_HOST = 7           // the host's mirror of the group set
};

// code:4 | feature:4 | group:28 | detail:28 -- the lowest key wins.  (A ROW
// does not fit 28 bits in a data set of 2^28 rows or more: the lowest
// offending row of either row check also goes, whole, to first_bad[2] /
// first_bad[3], and the host reports that one.)
__device__ __forceinline__ unsigned long long validate_key(
        int code, int feature, unsigned long long group,
        unsigned long long detail) {
    return ((unsigned long long)code << 60) | ((unsigned long long)feature << 56)
         | ((group & 0xFFFFFFFull) << 28) | (detail & 0xFFFFFFFull);
}

__global__ void k_validate_rows(SweepParams P, StatImage img,
                                const int32_t * __restrict__ g2p,
                                uint32_t n_global, size_t n_rows,
                                uint32_t * __restrict__ packed_out,
                                unsigned long long * __restrict__ first_bad,
                                unsigned long long * __restrict__ n_assigned) {
    const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const uint32_t id = P.assign[row];
    packed_out[row] = 0xFFFFFFFFu;
    if (id == 0xFFFFFFFFu) return;          // not assigned yet (init path)
    const int32_t g = id < n_global ? g2p[id] : -1;
    if (g < 0 || g >= P.K) {
        atomicMin(first_bad, validate_key(VALIDATE_DEAD_ID, 0, row, id));
        atomicMin(first_bad + 2, (unsigned long long)row);
        return;
    }
    packed_out[row] = (uint32_t)g;
    atomicAdd(n_assigned, 1ull);
    atomicAdd(&img.counts[g], 1);
    for (int f = 0; f < P.F; ++f) {
        const SlaveView & s = P.feat[f];
        const uint32_t x = P.values[f][row];
        switch (s.kind) {
        case DIST_DD:
        case DIST_DPD:
            if (x >= (uint32_t)s.dim) {
                atomicMin(first_bad,
                          validate_key(VALIDATE_VALUE_RANGE, f, row, x));
                atomicMin(first_bad + 3, (unsigned long long)row);
                break;
            }
            atomicAdd(&img.i0[f][g], 1);
            atomicAdd(&img.cnt[f][(size_t)g * s.dim + x], 1);
            break;
        case DIST_BB:
            atomicAdd(x ? &img.i0[f][g] : &img.i1[f][g], 1);
            break;
        case DIST_GP:
        case DIST_BNB:
            atomicAdd(&img.i0[f][g], 1);
            atomicAdd(&img.i1[f][g], (int32_t)x);
            break;
        default:    // NormalInverseChiSq: its count is the group's size
            atomicAdd(&img.i0[f][g], 1);
            break;
        }
    }
}

// item i: group i / width, column i % width of [size | i0 | i1 | cnt[dim]]
// of feature f (f = -1: the group sizes)
__global__ void k_validate_compare(SweepParams P, StatImage live,
                                   StatImage recount, int f, size_t items,
                                   unsigned long long * __restrict__ first_bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= items) return;
    if (f < 0) {
        if (live.counts[i] != recount.counts[i])
            atomicMin(first_bad, validate_key(VALIDATE_GROUP_SIZE, 0, i,
                                              (uint32_t)recount.counts[i]));
        return;
    }
    const SlaveView & s = P.feat[f];
    const bool cat = s.kind == DIST_DD || s.kind == DIST_DPD;
    const size_t width = cat ? 2 + (size_t)s.dim : 2;
    const size_t k = i / width, c = i % width;
    if (c == 0) {
        if (live.i0[f][k] != recount.i0[f][k])
            atomicMin(first_bad, validate_key(VALIDATE_STAT0, f, k,
                                              (uint32_t)recount.i0[f][k]));
    } else if (c == 1) {
        if (s.kind != DIST_NICH && !cat
            && live.i1[f][k] != recount.i1[f][k])
            atomicMin(first_bad, validate_key(VALIDATE_STAT1, f, k,
                                              (uint32_t)recount.i1[f][k]));
    } else {
        const size_t cell = k * s.dim + (c - 2);
        if (live.cnt[f][cell] != recount.cnt[f][cell])
            atomicMin(first_bad, validate_key(VALIDATE_CELL, f, k, c - 2));
    }
}

}  // namespace dist
