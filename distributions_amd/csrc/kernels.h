// HIP kernels of libdistributions_hip (gfx950).  Included once, by
// dist_hip.hip.  Layout and roofline notes per kernel are in DESIGN.md.
#pragma once

#include <type_traits>

#include "models.h"

namespace dist {

constexpr int kBlock = 256;
constexpr int kMaxF = DIST_MAX_FEATURES;

// ---------------------------------------------------------------------------
// elementwise special functions (vector_math.cc:190-291)

enum VecOp { VEC_LOG, VEC_EXP, VEC_LGAMMA, VEC_LGAMMA_NU, VEC_LOG_FACTORIAL };

__global__ void k_vector_op(int op, size_t n, const float * __restrict__ in,
                            float * __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    float y;
    switch (op) {
    case VEC_LOG: y = fast_log(x); break;
    case VEC_EXP: y = fast_exp(x); break;
    case VEC_LGAMMA: y = fast_lgamma(x); break;
    case VEC_LGAMMA_NU: y = fast_lgamma_nu(x); break;
    default: y = fast_log_factorial(f2u(x)); break;
    }
    out[i] = y;
}

// ---------------------------------------------------------------------------
// sampling from a score vector, the scalar algorithm of random.cc:94-106 and
// random.hpp:316-333 run by one lane (API path; the sweep kernel below runs
// the same recurrence once per lane)

struct SampleOut {
    float total;
    float log_sum_exp;
    int sample;
};

// mode 0: scores_to_likelihoods; 1: + sample (u given); 2: log_sum_exp only;
// 3: sample from given likelihoods/total
__global__ void k_sample_scalar(int mode, int n, float * __restrict__ scores,
                                float total_in, float u, SampleOut * out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float total = total_in;
    if (mode != 3) {
        float m = scores[0];
        for (int i = 0; i < n; ++i) {
            const float x = scores[i];
            m = x > m ? x : m;
        }
        total = 0.f;
        for (int i = 0; i < n; ++i) {
            const float l = fast_exp(scores[i] - m);
            if (mode != 2) scores[i] = l;
            total += l;
        }
        out->log_sum_exp = n ? fast_log(total) + m : 0.f;
    }
    out->total = total;
    int sample = n - 1;
    if (mode == 1 || mode == 3) {
        float t = total * u;
        for (int i = 0; i < n; ++i) {
            t -= scores[i];
            if (t <= 0.f) { sample = i; break; }
        }
    }
    out->sample = sample;
}

// ---------------------------------------------------------------------------
// PitmanYor cached mixture (clustering.hpp:151-230)

__global__ void k_py_rebuild(const int32_t * __restrict__ counts,
                             float * __restrict__ shifted, int K, float alpha,
                             float d, int nonempty, int empty) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int n = counts[k];
    shifted[k] = n ? py_nonempty_score(n, d)
                   : py_empty_score(alpha, d, nonempty, empty);
}

__global__ void k_py_set_count(int32_t * counts, float * shifted, int k,
                               int n, float d) {
    counts[k] = n;
    if (n) shifted[k] = py_nonempty_score(n, d);
}

__global__ void k_py_move(int32_t * counts, float * shifted, int dst, int src) {
    counts[dst] = counts[src];
    shifted[dst] = shifted[src];
}

__global__ void k_py_score(const float * __restrict__ shifted,
                           float * __restrict__ out, int K,
                           long long sample_size, float alpha) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    out[k] = shifted[k] + py_shift(sample_size, alpha);
}

// MixtureDriver<LowEntropy>::score_value (mixture.hpp:124-141)
__global__ void k_le_score(const int32_t * __restrict__ counts,
                           float * __restrict__ out, int K, int dataset_size,
                           int sample_size, int empty) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    out[k] = le_score_add_value(dataset_size, counts[k], sample_size, empty);
}
__global__ void k_le_score_add_value(int dataset_size, int group_size,
                                     int sample_size, int empty, float * out) {
    *out = le_score_add_value(dataset_size, group_size, sample_size, empty);
}
// LowEntropy::score_counts (clustering.cc:229-238): sum of n log n, in
// binary64 (the reference accumulates in float)
__global__ void k_le_count_terms(const int32_t * __restrict__ counts, int K,
                                 double * out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (k < K && counts[k] > 1)
        acc = (double)((float)counts[k] * fast_log((float)counts[k]));
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) atomicAdd(out, acc);
}

__global__ void k_py_score_add_value(float alpha, float d, int group_size,
                                     int nonempty, int sample_size, int empty,
                                     float * out) {
    *out = py_score_add_value(alpha, d, group_size, nonempty, sample_size,
                              empty);
}

// ---------------------------------------------------------------------------
// feature slaves (mixture.hpp:340-450 + the per-model value scorers)

// DPD prior mass alpha * beta_v (dpd.hpp:424) and the OTHER score
__global__ void k_dpd_prior(float alpha, const float * __restrict__ betas,
                            float * __restrict__ prior, int dim) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < dim) prior[v] = alpha * betas[v];
}
__global__ void k_dpd_other(float alpha, float beta0, float * out) {
    *out = fast_log(alpha * beta0);
}

// Group::init for groups [k0, k1)
__global__ void k_slave_zero_groups(SlaveView s, int k0, int k1) {
    const size_t width = is_cat(s.kind) ? (size_t)s.dim : 1;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)(k1 - k0) * width) return;
    const int k = k0 + (int)(i / width);
    const int v = (int)(i % width);
    if (is_cat(s.kind)) s.cnt[(size_t)k * s.dim + v] = 0;
    if (v == 0) {
        s.i0[k] = 0; s.i1[k] = 0; s.f0[k] = 0.f; s.f1[k] = 0.f;
    }
}

// MixtureValueScorer::update_group for groups [k0, k1) (update_all when the
// range is everything): dd.hpp:369-379,399-421 etc.  One thread per
// (value, group) cell, group fastest so that S[v][k] stores coalesce.
__global__ void k_slave_update(SlaveView s, int k0, int k1) {
    const size_t nk = (size_t)(k1 - k0);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (is_cat(s.kind)) {
        if (i >= nk * (size_t)s.dim) return;
        const int v = (int)(i / nk);
        const int k = k0 + (int)(i % nk);
        refresh_cat_cell(s, k, v);
        if (v == 0) refresh_shift(s, k);
    } else {
        if (i >= nk) return;
        refresh_scalar_entry(s, k0 + (int)i);
    }
}

// MixtureSlave::add_value / remove_value for one row (API path)
__global__ void k_slave_value_op(SlaveView s, int k, uint32_t value, int add) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Stats st = load_stats(s, k);
    if (add) stats_add(s.kind, st, value); else stats_remove(s.kind, st, value);
    store_stats(s, k, st);
    if (is_cat(s.kind)) {
        s.cnt[(size_t)k * s.dim + value] += add ? 1 : -1;
        refresh_cat_cell(s, k, (int)value);   // dd.hpp:458-467
        refresh_shift(s, k);
    } else {
        refresh_scalar_entry(s, k);
    }
}

// Packed_::packed_remove (vector.hpp:47-51): group `src` moves into `dst`
__global__ void k_slave_move_group(SlaveView s, int dst, int src) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (is_cat(s.kind)) {
        if (v < s.dim) {
            s.cnt[(size_t)dst * s.dim + v] = s.cnt[(size_t)src * s.dim + v];
            s.S[(size_t)v * s.cap + dst] = s.S[(size_t)v * s.cap + src];
        }
    }
    if (v == 0) {
        s.i0[dst] = s.i0[src]; s.i1[dst] = s.i1[src];
        s.f0[dst] = s.f0[src]; s.f1[dst] = s.f1[src];
        s.c0[dst] = s.c0[src]; s.c1[dst] = s.c1[src];
        s.c2[dst] = s.c2[src]; s.c3[dst] = s.c3[src];
    }
}

// What the device knows about the group set when it normalises the set
// itself between batches (k_normalise): the host then queues whole sweeps
// without looking at the group sizes, its mirrors follow afterwards.
struct DevState {
    int K;                   // groups after the last normalisation
    int k_new;               // first slot that normalisation appended
    int created;             // slots it appended: [k_new, K)
    int removed;             // groups it swap-removed
    uint32_t global_size;    // ids handed out so far (MixtureIdTracker)
    uint32_t first_new_global;   // id of slot k_new
    int nonempty;            // K - (empty groups)
    int pad;
};

struct SweepScalars {
    float shift;         // -fast_log(float(N - 1) + alpha)   (row removed)
    float empty_single;  // empty-group score with one non-empty group fewer
    float shift_full;    // -fast_log(float(N) + alpha)       (no removal)
};

// The driver's contribution to a row's scores in batch semantics (one row
// taken out): base[k] for rows that leave their group non-empty, base_single[k]
// for a row that was alone in its group (one non-empty group fewer in the
// empty groups' prior, clustering.hpp:221-230), and the scalars.
struct DriverPrep {
    float alpha, d;
    int cluster, dataset_size;   // see SweepParams::cluster
    long long sample_size;
    int K, n_empty;
    float * base;
    float * base_single;
    SweepScalars * scalars;
};
__device__ __forceinline__ void driver_prepare_slot(const DriverPrep & P,
                                                    size_t i, int count,
                                                    float shifted) {
    if (P.cluster == 1) {
        // MixtureDriver<LowEntropy>::score_value with the row removed:
        // sample_size - 1 rows; the score of a slot depends on its own size
        // only, so a vanished singleton changes nothing else
        if (i == 0) {
            P.scalars->shift = 0.f;
            P.scalars->shift_full = 0.f;
            P.scalars->empty_single = le_score_add_value(
                P.dataset_size, 0, (int)P.sample_size - 1, P.n_empty);
        }
        if (i >= (size_t)P.K) return;
        const float s = le_score_add_value(P.dataset_size, count,
                                           (int)P.sample_size - 1, P.n_empty);
        P.base[i] = s;
        P.base_single[i] = s;
        return;
    }
    const float shift = py_shift(P.sample_size - 1, P.alpha);
    const float empty_single =
        py_empty_score(P.alpha, P.d, P.K - P.n_empty - 1, P.n_empty);
    if (i == 0) {
        P.scalars->shift = shift;
        P.scalars->shift_full = py_shift(P.sample_size, P.alpha);
        P.scalars->empty_single = empty_single;
    }
    if (i >= (size_t)P.K) return;
    P.base[i] = shifted + shift;
    P.base_single[i] = (count == 0 ? empty_single : shifted) + shift;
}

// The tail of a batch's normalisation in ONE launch (it sits between the
// host's look at the group sizes and the next batch's first kernel, so every
// launch here is idle time on the device): groups [k_new, K) are appended
// empty (Group::init), every group's cache entries are rebuilt from its
// statistics (update_all, dd.hpp:399-421 etc.), and the driver's shifted
// scores are rebuilt (clustering.hpp:151-161).  blockIdx.y = feature, the
// last y-slice is the driver.
struct FinishParams {
    int F;
    SlaveView feat[kMaxF];
    int32_t * counts;      // driver
    float * shifted;
    int K, k_new;
    int cells_fresh;       // categorical cells of old groups are current
    float alpha, d;
    int nonempty, empty;
    DriverPrep prep;       // the next batch's base scores, while we are here
    // id maps of the appended groups (MixtureIdTracker::add_group,
    // mixture.hpp:474-479): slot k gets global id first_new_global + k - k_new;
    // nullptr when the host uploads the maps itself
    uint32_t * p2g;
    int32_t * g2p;
    uint32_t first_new_global;
    // the device normalised the group set (k_normalise): K, k_new, nonempty
    // and first_new_global are read from *dev instead of the fields above
    const DevState * dev;
    int32_t * snap;        // (optional) receives the new group sizes
};
__global__ void k_batch_finish(FinishParams P) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int K = P.K, k_new = P.k_new, nonempty = P.nonempty;
    uint32_t first_new_global = P.first_new_global;
    DriverPrep prep = P.prep;
    if (P.dev) {   // the device normalised the group set itself (k_normalise)
        K = P.dev->K;
        k_new = P.dev->k_new;
        nonempty = P.dev->nonempty;
        first_new_global = P.dev->first_new_global;
        prep.K = K;
    }
    if ((int)blockIdx.y == P.F) {
        if (i >= (size_t)K) return;
        const int k = (int)i;
        int n = P.counts[k];
        if (k >= k_new) {
            n = 0;
            P.counts[k] = 0;
            if (P.p2g) {
                const uint32_t global = first_new_global + (uint32_t)(k - k_new);
                P.p2g[k] = global;
                P.g2p[global] = k;
            }
        }
        if (P.snap) P.snap[k] = n;   // the sizes the next batch starts from
        const float shifted =
            n ? py_nonempty_score(n, P.d)
              : py_empty_score(P.alpha, P.d, nonempty, P.empty);
        P.shifted[k] = shifted;
        driver_prepare_slot(prep, i, n, shifted);
        return;
    }
    const SlaveView & s = P.feat[blockIdx.y];
    if (is_cat(s.kind)) {
        int v, k;
        if (P.cells_fresh) {
            // only the appended groups' cells and every group's shift
            const int n_new = K - k_new;
            if (i < (size_t)K) {
                if ((int)i < k_new) refresh_shift(s, (int)i);
            }
            if (i >= (size_t)n_new * s.dim) return;
            v = (int)(i / n_new);
            k = k_new + (int)(i % n_new);
        } else {
            if (i >= (size_t)K * s.dim) return;
            v = (int)(i / K);
            k = (int)(i % K);     // group fastest: S[v][k] coalesces
        }
        if (k >= k_new) {
            s.cnt[(size_t)k * s.dim + v] = 0;
            if (v == 0) { s.i0[k] = 0; s.i1[k] = 0; s.f0[k] = 0.f; s.f1[k] = 0.f; }
            s.S[(size_t)v * s.cap + k] = fast_log(s.prior[v] + 0.f);
            if (v == 0) s.c0[k] = fast_log(s.alpha_sum + 0.f);
            return;
        }
        refresh_cat_cell(s, k, v);
        if (v == 0) refresh_shift(s, k);
    } else {
        if (i >= (size_t)K) return;
        const int k = (int)i;
        if (k >= k_new) {
            const Stats zero = {0, 0, 0.f, 0.f};
            store_stats(s, k, zero);
        }
        refresh_scalar_entry(s, k);
    }
}

// MixtureDriver's group-set normalisation after a batch (mixture.hpp:84-89,
// 108-119; what Gibbs::batch_finish works out on the host), on the device:
// ONE workgroup compares the group sizes with those at batch entry (`snap`).
// Groups that lost their last member are swap-removed in descending slot
// order -- which comes to: the survivors behind the new end, in descending
// slot order, fill the vacated slots in front of it, in descending slot order
// -- with their statistics, cache entries and ids; every previously empty
// group that gained members is replaced by a new empty one at the end, whose
// statistics, cache entries and ids k_batch_finish writes (slots >= k_new).
constexpr int kNormaliseBlock = 1024;
struct NormaliseParams {
    int F;
    SlaveView feat[kMaxF];
    int32_t * counts;
    const int32_t * snap;
    uint32_t * p2g;
    int32_t * g2p;
    DevState * dev;
    int n_empty;         // invariant of the chain
};
__global__ __launch_bounds__(kNormaliseBlock) void k_normalise(
        NormaliseParams P) {
    // [K + 2] emptied-before (padded to 8 bytes) | [K / 2 + 1] {dst, src}
    extern __shared__ int nm_lds[];
    __shared__ int s_part[kNormaliseBlock / 64];
    __shared__ int s_created, s_moves;
    const int K = P.dev->K;
    int * before = nm_lds;            // before[k] = emptied groups in [0, k)
    int2 * moves = reinterpret_cast<int2 *>(nm_lds + ((K + 2) & ~1));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_created = 0; s_moves = 0; }
    __syncthreads();
    // each thread owns a contiguous slice of the slots
    const int per = (K + kNormaliseBlock - 1) / kNormaliseBlock;
    const int lo = min(K, tid * per), hi = min(K, lo + per);
    int mine = 0, created = 0;
    for (int k = lo; k < hi; ++k) {
        const int was = P.snap[k], now = P.counts[k];
        mine += (was > 0 && now == 0);
        created += (was == 0 && now > 0);
    }
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_part[wave] = incl;
    if (created) atomicAdd(&s_created, created);
    __syncthreads();
    int run = incl - mine;
    for (int w = 0; w < wave; ++w) run += s_part[w];
    for (int k = lo; k < hi; ++k) {
        before[k] = run;
        run += (P.snap[k] > 0 && P.counts[k] == 0);
    }
    if (hi == K && lo < K) before[K] = run;   // (the owner of the last slot)
    __syncthreads();
    const int removed = K > 0 ? before[K] : 0;
    const int size = K - removed;
    // the ids of the vanished groups retire (mixture.hpp:481-497) before any
    // slot is overwritten
    for (int k = lo; k < hi; ++k)
        if (before[k + 1] != before[k]) P.g2p[P.p2g[k]] = -1;
    // The i-th removal (descending slots, i = vanished groups behind it)
    // pulls in whatever sits in slot K - 1 - i at that time: that slot's own
    // group if it survives, else what THAT slot pulled in at its own, earlier
    // removal.  A vacated slot in front of the new end follows this chain to
    // the survivor it ends up with.
    for (int k = lo; k < min(hi, size); ++k)
        if (before[k + 1] != before[k]) {
            int t = k;
            do {
                t = K - 1 - (removed - before[t + 1]);
            } while (before[t + 1] != before[t]);
            moves[atomicAdd(&s_moves, 1)] = int2{k, t};
        }
    __syncthreads();
    const int n_moves = s_moves;
    // Packed_::packed_remove for every such pair, all objects
    for (int m = tid; m < n_moves; m += kNormaliseBlock) {
        const int dst = moves[m].x, src = moves[m].y;
        P.counts[dst] = P.counts[src];
        const uint32_t gid = P.p2g[src];
        P.p2g[dst] = gid;
        P.g2p[gid] = dst;
    }
    for (int f = 0; f < P.F; ++f) {
        const SlaveView & s = P.feat[f];
        const int width = is_cat(s.kind) ? s.dim : 1;
        for (int e = tid; e < n_moves * width; e += kNormaliseBlock) {
            const int dst = moves[e / width].x, src = moves[e / width].y;
            const int v = e % width;
            if (is_cat(s.kind)) {
                s.cnt[(size_t)dst * s.dim + v] = s.cnt[(size_t)src * s.dim + v];
                s.S[(size_t)v * s.cap + dst] = s.S[(size_t)v * s.cap + src];
            }
            if (v == 0) {
                s.i0[dst] = s.i0[src]; s.i1[dst] = s.i1[src];
                s.f0[dst] = s.f0[src]; s.f1[dst] = s.f1[src];
                s.c0[dst] = s.c0[src]; s.c1[dst] = s.c1[src];
                s.c2[dst] = s.c2[src]; s.c3[dst] = s.c3[src];
            }
        }
    }
    if (tid == 0) {
        const int n_created = s_created;
        P.dev->k_new = size;
        P.dev->created = n_created;
        P.dev->removed = removed;
        P.dev->K = size + n_created;
        P.dev->first_new_global = P.dev->global_size;
        P.dev->global_size += (uint32_t)n_created;
        P.dev->nonempty = size + n_created - P.n_empty;
        // packed indices mean something else now: the removal epoch moves on
        // (VsOffsets).  No entry goes into k_vs_tables' log of moves for it,
        // so offsets recorded before this launch are not translated across
        // it -- the chunks they belong to go without a band until their next
        // sort.  (Without this a fused batch whose group set was closed HERE,
        // on the host's demand, left the next run trusting offsets under
        // indices that no longer held: rows in neither tile nor band, moves
        // applied twice -- tools/fuzz.py seed 501609.)
        if (removed > 0) P.dev->pad += 1;
    }
}

// The group sizes, straight into pinned host memory, then a sequence number:
// the host polls the number instead of paying a copy engine round trip and a
// stream-synchronise wake-up on the critical path of every batch.
__global__ void k_publish_counts(const int32_t * __restrict__ counts, int K,
                                 int * host_counts,
                                 volatile unsigned int * host_seq,
                                 unsigned int seq) {
    // launched as ONE block: its barrier orders every store before the ticket
    for (int k = threadIdx.x; k < K; k += blockDim.x)
        host_counts[k] = counts[k];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *host_seq = seq;
}

// Many packed_remove steps at once: after a batch the host works out which
// original group ends up in which slot (sources lie beyond the new end,
// destinations inside it, so the copies are independent) and one launch per
// object performs them.  moves[i] = {dst, src}.
__global__ void k_slave_move_groups(SlaveView s, const int2 * __restrict__ moves,
                                    int n_moves) {
    const int width = is_cat(s.kind) ? s.dim : 1;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_moves * width) return;
    const int dst = moves[i / width].x, src = moves[i / width].y;
    const int v = (int)(i % width);
    if (is_cat(s.kind)) {
        s.cnt[(size_t)dst * s.dim + v] = s.cnt[(size_t)src * s.dim + v];
        s.S[(size_t)v * s.cap + dst] = s.S[(size_t)v * s.cap + src];
    }
    if (v == 0) {
        s.i0[dst] = s.i0[src]; s.i1[dst] = s.i1[src];
        s.f0[dst] = s.f0[src]; s.f1[dst] = s.f1[src];
        s.c0[dst] = s.c0[src]; s.c1[dst] = s.c1[src];
        s.c2[dst] = s.c2[src]; s.c3[dst] = s.c3[src];
    }
}
__global__ void k_py_move_groups(int32_t * counts, float * shifted,
                                 const int2 * __restrict__ moves, int n_moves) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_moves) return;
    counts[moves[i].x] = counts[moves[i].y];
    shifted[moves[i].x] = shifted[moves[i].y];
}

// MixtureSlave::score_value (accumulates) and score_value_group
__global__ void k_slave_score_value(SlaveView s, uint32_t value,
                                    float * __restrict__ acc, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float lf = s.kind == DIST_GP ? fast_log_factorial(value) : 0.f;
    acc[k] = accumulate(s.kind, acc[k], load_entry(s, k, value), value, lf,
                        s.p);
}
// the same for a batch of values: acc[r * ld + k] accumulates the score of
// values[r] in group k (one launch instead of one per value; per element the
// very operations of k_slave_score_value)
__global__ void k_slave_score_values(SlaveView s,
                                     const uint32_t * __restrict__ values,
                                     size_t n, float * __restrict__ acc,
                                     size_t ld, int K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * (size_t)K) return;
    const size_t r = i / (size_t)K;
    const int k = (int)(i % (size_t)K);
    const uint32_t value = values[r];
    const float lf = s.kind == DIST_GP ? fast_log_factorial(value) : 0.f;
    float * cell = acc + r * ld + k;
    *cell = accumulate(s.kind, *cell, load_entry(s, k, value), value, lf, s.p);
}
__global__ void k_slave_score_group(SlaveView s, int k, uint32_t value,
                                    float * out) {
    const float lf = s.kind == DIST_GP ? fast_log_factorial(value) : 0.f;
    *out = score_group(s.kind, load_entry(s, k, value), value, lf, s.p);
}

// ---------------------------------------------------------------------------
// MixtureDataScorer::score_data (dd.hpp:250-256,287-318; dpd.hpp:344-374;
// bb.hpp:207-229; gp.hpp:220-241; nich.hpp:262-288): every float term is the
// reference's; the terms are summed in binary64 (the reference accumulates in
// float, DD through the re-associated vector_sum) -- stated tolerance 1e-5
// relative against a float restatement of the reference's loops.

__device__ __forceinline__ void block_sum_to(double v, double * out) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v != 0.0) atomicAdd(out, v);
}

// the terms of cell i (categorical: one (group, value); scalar: one group)
__device__ __forceinline__ double score_data_cell(const SlaveView & s,
                                                  size_t i) {
    double acc = 0.0;
    if (is_cat(s.kind)) {
        const size_t n = (size_t)s.K * s.dim;
        if (i < n) {
            const int k = (int)(i / s.dim);
            const int v = (int)(i % s.dim);
            if (s.i0[k] != 0) {
                const float prior = s.prior[v];
                acc += (double)(fast_lgamma(
                                    prior + (float)s.cnt[(size_t)k * s.dim + v])
                                - fast_lgamma(prior));
                if (v == 0)
                    acc += (double)(fast_lgamma(s.alpha_sum)
                                    - fast_lgamma(s.alpha_sum + (float)s.i0[k]));
            }
        }
    } else if (i < (size_t)s.K) {
        float t[4];
        const int nt = scalar_mixture_score_terms(s.kind, s.p,
                                                  load_stats(s, (int)i), t);
        for (int j = 0; j < nt; ++j) acc += (double)t[j];
    }
    return acc;
}

// score_data_grid (mixture.hpp:238-247, dd.hpp:259-284): blockIdx.y = the
// candidate Shared; the groups' statistics are read once per candidate, the
// hyper-parameters come from the candidate arrays
__global__ void k_score_data_grid(SlaveView s, const float * __restrict__ cand_p,
                                  const float * __restrict__ cand_prior,
                                  const float * __restrict__ cand_alpha_sum,
                                  double * out) {
    const int c = blockIdx.y;
    for (int j = 0; j < 4; ++j) s.p[j] = cand_p[4 * c + j];
    if (is_cat(s.kind)) {
        s.prior = cand_prior + (size_t)c * s.dim;
        s.alpha_sum = cand_alpha_sum[c];
    }
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    block_sum_to(score_data_cell(s, i), out + c);
}

// MixtureDataScorer::score_data in the reference's own float accumulation
// order (bit-exact against a float restatement of its loops):
//   DirichletDiscrete (dd.hpp:287-318): one accumulator per value plus one for
//   the shift, each fed group by group, closed by vector_sum -- one thread per
//   accumulator walks the groups; blockIdx.x = candidate Shared
__global__ __launch_bounds__(512) void k_score_data_dd(
        SlaveView s, const float * __restrict__ cand_prior,
        const float * __restrict__ cand_alpha_sum, float * out) {
    __shared__ float chain[DIST_DD_MAX_DIM + 1];
    const int c = blockIdx.x;
    const float * prior = cand_prior + (size_t)c * s.dim;
    const float alpha_sum = cand_alpha_sum[c];
    const int v = threadIdx.x;
    if (v <= s.dim) {
        float acc = 0.f;
        if (v < s.dim) {
            const float a = prior[v];
            const float shared_part = fast_lgamma(a);
            for (int k = 0; k < s.K; ++k)
                if (s.i0[k])
                    acc += fast_lgamma(a + (float)s.cnt[(size_t)k * s.dim + v])
                         - shared_part;
        } else {
            const float shared_part = fast_lgamma(alpha_sum);
            for (int k = 0; k < s.K; ++k)
                if (s.i0[k])
                    acc += shared_part
                         - fast_lgamma(alpha_sum + (float)s.i0[k]);
        }
        chain[v] = acc;
    }
    __syncthreads();
    if (v == 0) out[c] = vector_sum_as_built((size_t)s.dim + 1, chain);
}
//   scalar kinds (bb.hpp:207-229, gp.hpp:220-241, nich.hpp:262-288,
//   bnb.hpp:226-245): ONE accumulator, every group adds its terms in order.
//   The terms are computed in parallel (absent ones as +0, which leaves the
//   accumulator unchanged) ...
__global__ void k_score_data_terms(SlaveView s, const float * __restrict__ cand_p,
                                   float * __restrict__ terms) {
    const int c = blockIdx.y;
    for (int j = 0; j < 4; ++j) s.p[j] = cand_p[4 * c + j];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= s.K) return;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    const int nt = scalar_mixture_score_terms(s.kind, s.p, load_stats(s, k), t);
    float * dst = terms + ((size_t)c * s.K + k) * 4;
    for (int j = 0; j < 4; ++j) dst[j] = j < nt ? t[j] : 0.f;
}
//   ... and summed by one wave per candidate in index order
__global__ __launch_bounds__(64) void k_score_data_serial(
        const float * __restrict__ terms, size_t n_terms, float * out) {
    const float * src = terms + (size_t)blockIdx.x * n_terms;
    const int lane = threadIdx.x;
    float total = 0.f;
    for (size_t i0 = 0; i0 < n_terms; i0 += 64) {
        const float mine = (i0 + lane < n_terms) ? src[i0 + lane] : 0.f;
#pragma unroll
        for (int j = 0; j < 64; ++j)
            total += u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(mine), j));
    }
    if (lane == 0) out[blockIdx.x] = total;
}

// PitmanYor::score_counts: before[k] = (non-empty groups, rows) ahead of k
__global__ void k_py_score_counts(const int32_t * __restrict__ counts,
                                  const unsigned long long * __restrict__ before,
                                  int K, float alpha, float d, double * out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (k < K && counts[k] > 0)
        acc = py_score_counts_term(alpha, d, counts[k], before[2 * k],
                                   before[2 * k + 1]);
    block_sum_to(acc, out);
}

// ---------------------------------------------------------------------------
// the batched row update


struct SweepParams {
    int F;
    SlaveView feat[kMaxF];
    const uint32_t * values[kMaxF];
    const int32_t * counts;    // driver counts[K] at batch entry
    const float * shifted;     // clustering.hpp shifted_scores_[K]
    const float * base;        // shifted[k] + shift
    const float * base_single; // the same for a row that was alone in its
                               // group: empty slots score with one non-empty
                               // group fewer (clustering.hpp:221-230)
    // per feature, optional k-major gather table [K][nv] rebuilt per batch:
    //   GP:      the whole additive term for value v at group k
    //   DD/DPD:  S[v][k] transposed (lanes of a wave then gather inside one
    //            short row instead of striding over the value-major cache)
    const float * ktab[kMaxF];
    int ktab_nv[kMaxF];
    const SweepScalars * scalars;
    int K;
    int n_empty;
    float alpha, d;
    // the clustering model: 0 = PitmanYor(alpha, d) through the cached
    // driver (clustering.hpp:126-234); 1 = LowEntropy(dataset_size) through
    // the generic MixtureDriver (mixture.hpp:124-141)
    int cluster;
    int dataset_size;
    long long sample_size;
    const uint32_t * assign;   // global group id per local row
    const int32_t * g2p;       // global -> packed at batch entry
    uint32_t * old_packed;     // per batch row
    uint32_t * new_packed;
    size_t row_begin, row_end;
    unsigned long long row_offset;   // global index of local row 0
    unsigned long long draw_base;
    uint32_t seed_state;
    // entropy of the open batch: row (row_begin + b) draws with engine state
    //   seed_batch * 16807^b  =  seed_state * 16807^(draw_base+row_offset+row+1)
    // 16807^b = pow_lo[b & 4095] * pow_hi[b >> 12]   (mod 2^31-1)
    uint32_t seed_batch;
    const uint32_t * pow_lo;   // [4096]  16807^i
    const uint32_t * pow_hi;   // [..]    16807^(4096 i)
    // when set, the generic kernel scores the listed items instead of the
    // whole range: the rows the value-sorted kernel handed over, as POSITIONS
    // in the batch's value-sorted order (row = row_begin + sorted_rows[pos])
    const uint32_t * row_list;
    const uint32_t * row_list_count;
    // value-sorted batches keep their per-row arrays in sorted-position order
    // (coalesced for the kernels that walk tiles): the current assignment as
    // global id, and old_packed / new_packed of the open batch
    const uint32_t * sorted_rows;
    const uint32_t * assign_pos;
    // non-null: the group count of record is dev->K (K above is then only
    // an upper bound the host sized its launches and buffers with)
    const DevState * dev;
};
__device__ __forceinline__ int sweep_K(const SweepParams & P) {
    return P.dev ? P.dev->K : P.K;
}

// the clustering model's score of the row's own group, which keeps
// `remaining` >= 1 members once the row is out
__device__ __forceinline__ float cluster_own_score(const SweepParams & P,
                                                   int remaining,
                                                   float shift) {
    if (P.cluster == 1)
        return le_score_add_value(P.dataset_size, remaining,
                                  (int)P.sample_size - 1, P.n_empty);
    return py_nonempty_score(remaining, P.d) + shift;
}

// Integer statistics are exact under atomics.  `stats` is either the live
// state or a zeroed delta image in the stat-word layout:
//   counts[K] | per feature: i0[K] i1[K] (categorical: cnt[K][dim])
// NormalInverseChiSq's count moves with its float statistics in k_replay.
struct StatImage {
    int32_t * counts;
    int32_t * i0[kMaxF];
    int32_t * i1[kMaxF];
    int32_t * cnt[kMaxF];
};

// wave-uniform read-only data: loads through the constant address space are
// issued as scalar loads (s_load_dwordx8/x16) when the address is uniform
typedef const float __attribute__((address_space(4))) * uniform_fp;
__device__ __forceinline__ uniform_fp as_uniform(const float * p) {
    return (uniform_fp)(unsigned long long)p;
}

// sample_unif01 of batch row b (random.hpp:47-50): one engine step per row,
// the step the sequential chain would have used for it
__device__ __forceinline__ float batch_row_unif01(const SweepParams & P,
                                                  size_t row) {
    const size_t b = row - P.row_begin;
    uint32_t xs = lcg_mulmod(P.seed_batch, P.pow_lo[b & 4095]);
    xs = lcg_mulmod(xs, P.pow_hi[b >> 12]);
    return lcg_unif01(xs);
}

__global__ void k_pow_tables(uint32_t * pow_lo, uint32_t * pow_hi,
                             uint32_t n_hi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4096) pow_lo[i] = lcg_jump(1u, i);
    if (i < n_hi) pow_hi[i] = lcg_jump(1u, 4096ull * i);
}

// base[k], base_single[k] and the scalars of a batch
__global__ void k_sweep_prepare(SweepParams P, float * __restrict__ base,
                                float * __restrict__ base_single,
                                SweepScalars * scalars) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const DriverPrep D = {P.alpha, P.d, P.cluster, P.dataset_size,
                          P.sample_size, P.K, P.n_empty, base, base_single,
                          scalars};
    const bool in = i < (size_t)P.K;
    driver_prepare_slot(D, i, in ? P.counts[i] : 0, in ? P.shifted[i] : 0.f);
}

// k-major gather table of one feature (see SweepParams::ktab)
__global__ void k_build_ktab(SlaveView v, float * __restrict__ tab, int nv,
                             int K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)K * nv) return;
    const int k = (int)(i / nv);
    const uint32_t x = (uint32_t)(i % nv);
    if (is_cat(v.kind)) {
        tab[i] = v.S[(size_t)x * v.cap + k];
    } else {   // GP: gp.cc:62-65, the term added to the accumulator
        const Entry e = {v.c0[k], v.c1[k], v.c2[k], v.c3[k]};
        tab[i] = score_group(v.kind, e, x, fast_log_factorial(x), v.p);
    }
}

// Scores of one row in batch semantics: state at batch entry minus the row.
//   count(g) >= 2: group order unchanged; slot g scored from (stats - row).
//   count(g) == 1: the group vanishes as MixtureDriver::remove_value does it
//     (mixture.hpp:108-119): the last group moves into slot g, one slot fewer,
//     and the empty groups' prior loses one non-empty group
//     (clustering.hpp:221-230).
// KIND0/KIND1 >= 0 pin the kind of features 0/1 at compile time and NF > 0
// the feature count; NF == 0 is the run-time generic form (any feature list).
template <int KIND0, int KIND1, int NF>
struct RowScorer {
    static constexpr int kUnroll = NF > 0 ? NF : 1;
    const SweepParams & P;
    uint32_t x[kMaxF];
    float lf[kMaxF];
    int g;
    int singleton;
    int Kl;
    float s_own;

    __device__ __forceinline__ int nf() const { return NF > 0 ? NF : P.F; }
    __device__ __forceinline__ int kind_of(int f) const {
        if (f == 0 && KIND0 >= 0) return KIND0;
        if (f == 1 && KIND1 >= 0) return KIND1;
        return P.feat[f].kind;
    }

    // the cache entry of slot k (wave-uniform k): scalar loads for the
    // per-group parameters, a per-lane gather only for a categorical table
    __device__ __forceinline__ Entry entry_at(const SlaveView & v, int kind,
                                              int k, uint32_t xv) const {
        Entry e;
        e.c0 = as_uniform(v.c0)[k];
        if (is_cat(kind)) {
            e.c1 = (kind == DIST_DPD && xv == DIST_DPD_OTHER)
                       ? v.other
                       : v.S[(size_t)xv * v.cap + k];
            e.c2 = 0.f;
            e.c3 = 0.f;
        } else {
            e.c1 = as_uniform(v.c1)[k];
            e.c2 = as_uniform(v.c2)[k];
            e.c3 = as_uniform(v.c3)[k];
        }
        return e;
    }

    // score of slot k from the caches (k wave-uniform).  PLAIN: no lane of
    // the wave holds a row that is alone in its group (the usual case): the
    // driver's score is one scalar operand instead of a per-lane select
    template <bool PLAIN = false>
    __device__ __forceinline__ float cached(int k) const {
        const float b = as_uniform(P.base)[k];
        float s = b;
        if (!PLAIN) {
            const float bs = as_uniform(P.base_single)[k];
            s = singleton ? bs : b;
        }
#pragma unroll kUnroll
        for (int f = 0; f < nf(); ++f) {
            const int kind = kind_of(f);
            const float * tab = P.ktab[f];
            const int nv = P.ktab_nv[f];
            if (tab != nullptr && (kind == DIST_GP || kind == DIST_BNB)) {
                // acc += term (gp.cc:62-65, bnb.hpp:316-327); values beyond
                // the table compute it
                const float term =
                    x[f] < (uint32_t)nv
                        ? tab[(size_t)k * nv + x[f]]
                        : score_group(kind,
                                      entry_at(P.feat[f], kind, k, x[f]),
                                      x[f], lf[f], P.feat[f].p);
                s = s + term;
            } else if (tab != nullptr && is_cat(kind)
                       && x[f] < (uint32_t)nv) {
                Entry e;
                e.c0 = as_uniform(P.feat[f].c0)[k];
                e.c1 = tab[(size_t)k * nv + x[f]];
                e.c2 = 0.f;
                e.c3 = 0.f;
                s = accumulate(kind, s, e, x[f], lf[f], P.feat[f].p);
            } else {
                s = accumulate(kind, s, entry_at(P.feat[f], kind, k, x[f]),
                               x[f], lf[f], P.feat[f].p);
            }
        }
        return s;
    }

    __device__ __forceinline__ RowScorer(const SweepParams & P_, size_t row,
                                         uint32_t global_id)
        : P(P_) {
        const float shift = P.scalars->shift;
        g = P.g2p[global_id];
        const int n_g = P.counts[g];
        singleton = (n_g == 1);
        Kl = sweep_K(P) - singleton;
#pragma unroll kUnroll
        for (int f = 0; f < nf(); ++f) {
            x[f] = P.values[f][row];
            lf[f] = kind_of(f) == DIST_GP ? fast_log_factorial(x[f]) : 0.f;
        }
        if (!singleton) {
            float s = cluster_own_score(P, n_g - 1, shift);
#pragma unroll kUnroll
            for (int f = 0; f < nf(); ++f) {
                SlaveView v = P.feat[f];
                v.kind = kind_of(f);
                s = accumulate(v.kind, s, entry_after_remove(v, g, x[f]),
                               x[f], lf[f], v.p);
            }
            s_own = s;
        } else {
            // slot g holds what was the last group (per-lane index: plain loads)
            const int src = sweep_K(P) - 1;
            float s = P.base_single[src];
#pragma unroll kUnroll
            for (int f = 0; f < nf(); ++f) {
                SlaveView v = P.feat[f];
                v.kind = kind_of(f);
                s = accumulate(v.kind, s, load_entry(v, src, x[f]), x[f],
                               lf[f], v.p);
            }
            s_own = s;
        }
    }

    // score of local slot k (k < K, wave-uniform; slots >= Kl are not part of
    // the row's view and are masked by the caller)
    template <bool PLAIN = false>
    __device__ __forceinline__ float at(int k) const {
        const float s = cached<PLAIN>(k);
        return k == g ? s_own : s;
    }

    // the same score with a per-lane slot index (lanes of a wave score 64
    // slots of ONE row at once): plain loads, identical arithmetic
    __device__ __forceinline__ float at_lane(int k) const {
        float s = singleton ? P.base_single[k] : P.base[k];
#pragma unroll kUnroll
        for (int f = 0; f < nf(); ++f) {
            SlaveView v = P.feat[f];
            v.kind = kind_of(f);
            s = accumulate(v.kind, s, load_entry(v, k, x[f]), x[f], lf[f],
                           v.p);
        }
        return k == g ? s_own : s;
    }
};

// One lane = one row: three passes over the groups in index order, exactly
// the scalar recurrences of scores_to_likelihoods (random.cc:94-106) and
// sample_from_likelihoods (random.hpp:316-333).  Rows are independent, so the
// float sums keep the reference's association while 64 rows run per wave.
// Per-group parameters arrive by scalar loads; the loops are unrolled so that
// those loads are issued ahead of the arithmetic that consumes them.
constexpr int kSweepUnroll = 4;

template <int KIND0, int KIND1, int NF>
__global__ __launch_bounds__(kBlock) void k_sweep_sample(SweepParams P) {
    __shared__ uint32_t s_exp[1024];
    for (int i = threadIdx.x; i < 1024; i += kBlock)
        s_exp[i] = g_tables_dev.exp_table[i];
    __syncthreads();
    const float ea = u2f(g_tables_dev.exp_ab[0]);
    const float eb = u2f(g_tables_dev.exp_ab[1]);
    const int K = sweep_K(P);

    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t n_items = P.row_list ? (size_t)*P.row_list_count
                                      : P.row_end - P.row_begin;
    // whole waves iterate together (inactive lanes idle) so that the
    // wave-level votes below see every lane
    const size_t n_round = (n_items + 63) / 64 * 64;
    for (size_t item = (size_t)blockIdx.x * kBlock + threadIdx.x;
         item < n_round; item += stride) {
        const bool live = item < n_items;
        // where the row's results go: batch-relative row index, or (list
        // mode) its position in the value-sorted order
        size_t out = live ? item : 0;
        size_t row = P.row_begin + out;
        uint32_t global_id;
        if (P.row_list && P.sorted_rows) {
            out = live ? (size_t)P.row_list[item] : 0;
            row = P.row_begin + P.sorted_rows[out];
            global_id = P.assign_pos[out];
        } else if (P.row_list) {   // a list of batch rows, in row order
            out = live ? (size_t)P.row_list[item] : 0;
            row = P.row_begin + out;
            global_id = P.assign[row];
        } else {
            global_id = P.assign[row];
        }
        const RowScorer<KIND0, KIND1, NF> rs(P, row, global_id);
        int Kl = rs.Kl;
        int steps = 0;
        // the three passes; PLAIN: no row of this wave is alone in its group,
        // so every lane sees all K slots and the same driver scores (no
        // per-lane select of the base score, no k < Kl masks)
        auto passes = [&](auto plain_tag) {
            constexpr bool PLAIN = decltype(plain_tag)::value;
            const int Kv = PLAIN ? K : Kl;
            // vector_max (vector_math.cc:74-83)
            float m = rs.template at<PLAIN>(0);
#pragma unroll kSweepUnroll
            for (int k = 1; k < K; ++k) {
                const float s = rs.template at<PLAIN>(k);
                m = (k < Kv && s > m) ? s : m;
            }
            // scores_to_likelihoods: total in index order
            float total = 0.f;
#pragma unroll kSweepUnroll
            for (int k = 0; k < K; ++k) {
                const float l = fast_exp_nonpos(rs.template at<PLAIN>(k) - m,
                                                s_exp, ea, eb);
                total += (k < Kv) ? l : 0.f;
            }
            // sample_from_likelihoods: subtracting non-negative terms never
            // increases t, so the first index with t <= 0 is the number of
            // steps after which t is still positive
            float t = total * batch_row_unif01(P, row);
            for (int k0 = 0; k0 < K; k0 += kSweepUnroll) {
#pragma unroll
                for (int j = 0; j < kSweepUnroll; ++j) {
                    const int k = k0 + j;
                    if (k < K) {
                        const float l = fast_exp_nonpos(
                            rs.template at<PLAIN>(k) - m, s_exp, ea, eb);
                        t -= (k < Kv) ? l : 0.f;
                        steps += (k < Kv && t > 0.f) ? 1 : 0;
                    }
                }
                if (!__any(live && t > 0.f)) break;
            }
        };
        // (the second copy of the loops only where they stay small: with the
        // count-valued kinds' out-of-line lgamma paths it costs the loops
        // their registers -- GP+NICH: 88 -> 175 and spills -- and so does
        // the lambda itself: those kinds keep the plain three loops)
        constexpr bool kTwoCopies =
            KIND0 >= 0 && KIND0 != DIST_GP && KIND0 != DIST_BNB
            && KIND1 != DIST_GP && KIND1 != DIST_BNB;
        if constexpr (kTwoCopies) {
            if (__any(rs.singleton != 0))
                passes(std::integral_constant<bool, false>{});
            else
                passes(std::integral_constant<bool, true>{});
        } else {
            float m = rs.at(0);
#pragma unroll kSweepUnroll
            for (int k = 1; k < K; ++k) {
                const float s = rs.at(k);
                m = (k < Kl && s > m) ? s : m;
            }
            float total = 0.f;
#pragma unroll kSweepUnroll
            for (int k = 0; k < K; ++k) {
                const float l = fast_exp_nonpos(rs.at(k) - m, s_exp, ea, eb);
                total += (k < Kl) ? l : 0.f;
            }
            float t = total * batch_row_unif01(P, row);
            for (int k0 = 0; k0 < K; k0 += kSweepUnroll) {
#pragma unroll
                for (int j = 0; j < kSweepUnroll; ++j) {
                    const int k = k0 + j;
                    if (k < K) {
                        const float l =
                            fast_exp_nonpos(rs.at(k) - m, s_exp, ea, eb);
                        t -= (k < Kl) ? l : 0.f;
                        steps += (k < Kl && t > 0.f) ? 1 : 0;
                    }
                }
                if (!__any(live && t > 0.f)) break;
            }
        }
        int g2 = steps < Kl - 1 ? steps : Kl - 1;
        if (rs.singleton && g2 == rs.g) g2 = K - 1;   // slot g held group K-1
        if (live) {
            P.old_packed[out] = (uint32_t)rs.g;
            P.new_packed[out] = (uint32_t)g2;
        }
    }
}

// ---------------------------------------------------------------------------
// Rows of mixed type (any feature list).  Scoring a row against a group is a
// short PROGRAM over per-batch tables, so the loop over groups has no
// model-specific code:
//   OP_GATHER_ADD  s += tab[k][x]     DD/DPD: the transposed cache column
//                                     (dd.hpp:433-445, first half); BB: the
//                                     head/tail score; GP/BNB: the whole
//                                     additive term (gp.cc:62-65)
//   OP_VEC_SUB     s -= vec[k]        DD/DPD shift (second half of the above)
//   OP_NICH        s += c0[k] + c1[k] * fast_log(1 + c2[k] * (x - c3[k])^2)
// in feature order, which is the reference's order of float operations.  A
// row's own slot takes a precomputed score (k_row_prepass: the statistics
// minus the row, by the model code); rows alone in their group and rows with a
// value outside a table are handed to the wave-per-row kernel.
enum { OP_GATHER_ADD = 0, OP_VEC_SUB = 1, OP_NICH = 2 };
constexpr int kMaxOps = 2 * kMaxF;
struct ScoreOp {
    int type;
    int f;               // feature whose value the op reads
    uint32_t nv;         // OP_GATHER_ADD: table width
    const float * p0;    // table / vector / NICH c0
    const float * p1;    // NICH c1..c3
    const float * p2;
    const float * p3;
};
struct ScoreProgram {
    int n;
    ScoreOp op[kMaxOps];
};

// own-slot score and hand-over flag of every batch row, by the model code
__global__ void k_row_prepass(SweepParams P, ScoreProgram prog,
                              float * __restrict__ own,
                              uint32_t * __restrict__ handed,
                              uint32_t * handed_count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.row_end - P.row_begin) return;
    const size_t row = P.row_begin + i;
    const int g = P.g2p[P.assign[row]];
    const int n_g = P.counts[g];
    bool hand = n_g == 1;   // the group would vanish: wave-per-row kernel
#pragma unroll
    for (int j = 0; j < kMaxOps; ++j) {
        if (j >= prog.n) break;
        if (prog.op[j].type == OP_GATHER_ADD
            && P.values[prog.op[j].f][row] >= prog.op[j].nv)
            hand = true;
    }
    // the own slot as remove_value + the cache refresh would leave it
    // (RowScorer's own-slot score; feature loop unrolled so that the row's
    // values and the feature views stay in registers)
    float s_own = 0.f;
    if (!hand) {
        s_own = cluster_own_score(P, n_g - 1, P.scalars->shift);
#pragma unroll
        for (int f = 0; f < kMaxF; ++f) {
            if (f >= P.F) break;
            const SlaveView & v = P.feat[f];
            const uint32_t x = P.values[f][row];
            const float lf = v.kind == DIST_GP ? fast_log_factorial(x) : 0.f;
            s_own = accumulate(v.kind, s_own, entry_after_remove(v, g, x), x,
                               lf, v.p);
        }
    }
    own[i] = s_own;
    if (hand) {
        handed[atomicAdd(handed_count, 1u)] = (uint32_t)i;
        P.old_packed[i] = 0xFFFFFFFFu;   // mark: not ours
    } else {
        P.old_packed[i] = (uint32_t)g;
    }
}

// kProgramBlock consecutive groups are scored at a time into registers: an
// op's parameters are fetched once per block and feature, not once per group.
constexpr int kProgramBlock = 16;

__device__ __forceinline__ void program_score_block(
        const SweepParams & P, const ScoreProgram & prog,
        const uint32_t (&xv)[kMaxOps], int k0, int g, float s_own,
        float (&s)[kProgramBlock]) {
    const int K = sweep_K(P);
#pragma unroll
    for (int j = 0; j < kProgramBlock; ++j) s[j] = as_uniform(P.base)[k0 + j];
#pragma unroll
    for (int o = 0; o < kMaxOps; ++o) {
        if (o >= prog.n) break;
        const int type = prog.op[o].type;
        if (type == OP_GATHER_ADD) {
            const uint32_t nv = prog.op[o].nv;
            const float * tab = prog.op[o].p0 + xv[o];
#pragma unroll
            for (int j = 0; j < kProgramBlock; ++j) {
                const int k = k0 + j < K ? k0 + j : K - 1;   // stay in the table
                s[j] = s[j] + tab[(size_t)k * nv];
            }
        } else if (type == OP_VEC_SUB) {
            uniform_fp vec = as_uniform(prog.op[o].p0);
#pragma unroll
            for (int j = 0; j < kProgramBlock; ++j) s[j] = s[j] - vec[k0 + j];
        } else {
            uniform_fp c0 = as_uniform(prog.op[o].p0);
            uniform_fp c1 = as_uniform(prog.op[o].p1);
            uniform_fp c2 = as_uniform(prog.op[o].p2);
            uniform_fp c3 = as_uniform(prog.op[o].p3);
            const float x = u2f(xv[o]);
#pragma unroll
            for (int j = 0; j < kProgramBlock; ++j) {
                const float d = x - c3[k0 + j];
                const float temp = 1.f + c2[k0 + j] * (d * d);
                s[j] = s[j] + (c0[k0 + j] + c1[k0 + j] * fast_log(temp));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kProgramBlock; ++j)
        if (k0 + j == g) s[j] = s_own;
}

__global__ __launch_bounds__(kBlock) void k_sweep_program(
        SweepParams P, ScoreProgram prog, const float * __restrict__ own) {
    __shared__ uint32_t s_exp[1024];
    for (int i = threadIdx.x; i < 1024; i += kBlock)
        s_exp[i] = g_tables_dev.exp_table[i];
    __syncthreads();
    const float ea = u2f(g_tables_dev.exp_ab[0]);
    const float eb = u2f(g_tables_dev.exp_ab[1]);
    const int K = sweep_K(P);
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t n_items = P.row_end - P.row_begin;
    const size_t n_round = (n_items + 63) / 64 * 64;
    for (size_t item = (size_t)blockIdx.x * kBlock + threadIdx.x;
         item < n_round; item += stride) {
        const bool in = item < n_items;
        const size_t out = in ? item : 0;
        const size_t row = P.row_begin + out;
        const uint32_t slot = P.old_packed[out];   // k_row_prepass
        const bool live = in && slot != 0xFFFFFFFFu;
        const int g = live ? (int)slot : -1;
        const float s_own = own[out];
        // (a handed-over row idles along on value 0: its own values may lie
        // outside the tables)
        uint32_t xv[kMaxOps];
#pragma unroll
        for (int o = 0; o < kMaxOps; ++o) {
            xv[o] = 0;
            if (o < prog.n && prog.op[o].type != OP_VEC_SUB && live)
                xv[o] = P.values[prog.op[o].f][row];
        }
        float s[kProgramBlock];
        // vector_max (vector_math.cc:74-83)
        float m = -INFINITY;
        for (int k0 = 0; k0 < K; k0 += kProgramBlock) {
            program_score_block(P, prog, xv, k0, g, s_own, s);
#pragma unroll
            for (int j = 0; j < kProgramBlock; ++j)
                m = (k0 + j < K && s[j] > m) ? s[j] : m;
        }
        // scores_to_likelihoods: total in index order (random.cc:100-103)
        float total = 0.f;
        for (int k0 = 0; k0 < K; k0 += kProgramBlock) {
            program_score_block(P, prog, xv, k0, g, s_own, s);
#pragma unroll
            for (int j = 0; j < kProgramBlock; ++j) {
                const float l = fast_exp_nonpos(s[j] - m, s_exp, ea, eb);
                total += k0 + j < K ? l : 0.f;
            }
        }
        // sample_from_likelihoods (random.hpp:316-333): t never increases
        float t = total * batch_row_unif01(P, row);
        int steps = 0;
        for (int k0 = 0; k0 < K; k0 += kProgramBlock) {
            program_score_block(P, prog, xv, k0, g, s_own, s);
#pragma unroll
            for (int j = 0; j < kProgramBlock; ++j) {
                if (k0 + j < K) {
                    t -= fast_exp_nonpos(s[j] - m, s_exp, ea, eb);
                    steps += t > 0.f ? 1 : 0;
                }
            }
            if (!__any(live && t > 0.f)) break;
        }
        if (live) P.new_packed[out] = (uint32_t)(steps < K - 1 ? steps : K - 1);
    }
}

// ---------------------------------------------------------------------------
// k_rows_scratch: general rows (any feature list), a lane per row.
//
// The three recurrences of a row (max, in-order total, subtractive scan:
// random.cc:94-106, random.hpp:316-333) each need every group's score; the
// exact mode evaluates score and exponential again in each pass, eight groups
// at a time in registers.  (The name is history: round 3 built and measured
// variants that kept the likelihoods -- or the scores as well -- in an HBM
// scratch column between the passes; they lost and are gone, see the kernel.)
// Same float operations in the same order as k_sweep_program: bit-identical.
//
// The per-group parameters of the whole program sit in one per-batch table
// (gtab[slot][Kpad]: slot 0 the driver's score, then each op's cache entries,
// one slot each), so a block of eight groups costs one 32-byte scalar load
// per slot, consecutive groups land in adjacent scalar registers (the
// operands of the packed instructions the compiler forms over groups 2p,
// 2p + 1), and the kernel takes a lean argument block instead of SweepParams
// (whose pointers alone exceed the scalar registers).  Table gathers are
// buffer loads: a scalar row offset plus the lane's value, no address
// arithmetic, and reads beyond the table (the padding groups of the last
// block) return zero.
// LDSLOG: FastLog's 64 KiB table is copied into LDS (the per-lane gather of
// nich.cc:60-66 then leaves the vector-memory path to the table gathers).
constexpr int kScratchMaxBlock = 1024;
constexpr int kRowsBlock = 8;       // groups scored at a time, in registers
constexpr int kRowsMaxW = 64;       // floats per gtab row
constexpr int kRowsMaxOps = 8;      // = kMaxF: one op per feature
// scan sampling: groups per snapshot of the running (sum, max)
constexpr int kRowsSuper = 32;

enum { ROP_GATHER = 0,   // s += tab[k][x]           BB, GP, BNB
       ROP_CAT = 1,      // s = (s + tab[k][x]) - shift[k]   DD, DPD
       ROP_NICH = 2 };
struct RowsOp {
    int type;
    int slot;                  // first float of the op's parameters in a row
    uint32_t tab_bytes;        // ROP_GATHER / ROP_CAT: K * nv * 4
    uint32_t row_bytes;        // nv * 4
    const float * tab;         // [K][nv]
    const uint32_t * values;   // the feature's column
};
struct RowsArgs {
    int n_ops;
    int W;                     // slots of gtab
    int K;                     // groups (an upper bound when dev != null)
    int Kpad;                  // row stride of gtab / fold / snap
    const DevState * dev;
    const float * gtab;        // [W][Kpad]
    const uint32_t * slot;     // k_row_prepass: own slot or 0xFFFFFFFF
    const float * own;         // k_row_prepass: own-slot score
    uint32_t * new_packed;
    size_t row_begin;
    size_t n_items;
    uint32_t seed_batch;
    int pad;
    const uint32_t * pow_lo;
    const uint32_t * pow_hi;
    float2 * snap;                 // scan: [waves][Kpad / kRowsSuper][64]
    // folded leading ops (see FoldSpec): the wave's rows share one joint
    // value `code`, their score before the first remaining op is
    // fold[code][k]; work items are tiles of the code-sorted row list
    const float * fold;            // [J][Kpad], null: no folding
    const uint32_t * sorted_rows;  // batch-relative row indices by code
    const uint4 * tiles;           // {code, first position, rows, 0}
    uint32_t n_tiles;
    uint32_t fold_codes;           // J
    RowsOp op[kRowsMaxOps];
};

// Folding.  The ops of a program before its first ROP_NICH read only small
// tables: for a row they depend on the row's discrete values alone.  Rows of
// a batch range are sorted once by the joint value of those features (values
// never change), a wave takes <= 64 rows of ONE joint value, and the score up
// to the first remaining op comes from a per-batch table fold[code][k] built
// with the very float operations, in the same order, that the unfolded ops
// perform -- by scalar loads, contiguous in k, instead of one gather per
// feature, row, group and pass.
struct FoldSpec {
    int n;                               // folded ops
    uint32_t nv[kRowsMaxOps];            // table widths
    uint32_t stride[kRowsMaxOps];        // code = sum x_f * stride_f
    const uint32_t * values[kRowsMaxOps];
    const float * tab[kRowsMaxOps];      // [K][nv]
    const float * shift[kRowsMaxOps];    // ROP_CAT: shift[k]; else null
};
// joint value of every row of [row_begin, row_begin + n): J for a row with a
// value outside a table (such rows are handed to the wave-per-row kernel)
__global__ void k_fold_codes(FoldSpec F, size_t row_begin, size_t n,
                             uint32_t J, uint32_t * __restrict__ codes,
                             uint32_t * __restrict__ index) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t code = 0;
    bool inside = true;
    for (int o = 0; o < F.n; ++o) {
        const uint32_t x = F.values[o][row_begin + i];
        inside = inside && x < F.nv[o];
        code += x * F.stride[o];
    }
    codes[i] = inside ? code : J;
    index[i] = (uint32_t)i;
}
// tiles of <= 64 equal-coded positions of the sorted list, in any order
__global__ void k_fold_tiles(const uint32_t * __restrict__ keys, size_t n,
                             uint4 * __restrict__ tiles,
                             uint32_t * tile_count) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t key = keys[p];
    // first position of the key's run (the keys are sorted)
    size_t lo = 0, hi = p;
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    if ((p - lo) % 64 != 0) return;
    size_t a = p, b = n;   // one past the run's last position
    while (a < b) {
        const size_t mid = (a + b) >> 1;
        if (keys[mid] <= key) a = mid + 1; else b = mid;
    }
    const uint32_t rows = (uint32_t)(a - p < 64 ? a - p : 64);
    tiles[atomicAdd(tile_count, 1u)] =
        make_uint4(key, (uint32_t)p, rows, 0u);
}
// fold[code][k]: the folded ops applied to base[k] in program order
__global__ void k_rows_fold(FoldSpec F, const float * __restrict__ base,
                            float * __restrict__ fold, uint32_t J, int Kpad,
                            int K_bound, const DevState * dev) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)J * Kpad) return;
    const int K = dev ? dev->K : K_bound;
    const int k = (int)(i % Kpad);
    uint32_t code = (uint32_t)(i / Kpad);
    float s = 0.f;
    if (k < K) {
        s = base[k];
        for (int o = 0; o < F.n; ++o) {
            const uint32_t x = code / F.stride[o];
            code -= x * F.stride[o];
            s = s + F.tab[o][(size_t)k * F.nv[o] + x];
            if (F.shift[o]) s = s - F.shift[o][k];   // dd.hpp:433-445
        }
    }
    fold[i] = s;
}

// gtab slot layout: { base, (per op in order) ROP_CAT: shift;
//                     ROP_NICH: c0, c1, c2, c3 }; groups beyond the group
// count are zero
struct GtabSource {
    int n;                          // slots after the first
    const float * p[kRowsMaxW];
};
__global__ void k_rows_gtab(const float * __restrict__ base, GtabSource src,
                            float * __restrict__ gtab, int Kpad, int K_bound,
                            const DevState * dev) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Kpad) return;
    const int K = dev ? dev->K : K_bound;
    const bool in = k < K;
    gtab[k] = in ? base[k] : 0.f;
    for (int i = 0; i < src.n; ++i)
        gtab[(size_t)(i + 1) * Kpad + k] = in ? src.p[i][k] : 0.f;
}

// fast_exp of a non-positive argument with the table in LDS, its entries
// already carrying the exponent bias: ((u + 127) << 23) | tbl[v] ==
// (u << 23) + (tbl[v] | 127 << 23).  The argument is in [-88, 0], so the
// nearest integer of x * a is exactly representable and float(r) is the
// rounded product itself (fmath.hpp:438-459, release-build order).
__device__ __forceinline__ float fast_exp_biased(float x,
                                                 const uint32_t * tab_biased,
                                                 float a, float b) {
    x = fmaxf(x, -88.0f);
    const float rf = __builtin_rintf(x * a);
    const int32_t r = (int32_t)rf;
    const uint32_t bits =
        ((uint32_t)(r >> 10) << 23) + tab_biased[(uint32_t)r & 1023u];
    return ((x + 1.0f) - rf * b) * u2f(bits);
}

// tab[k][x] for the block's groups: buffer loads with the row's byte offset
// as the scalar offset and the lane's value (times four) as the vector offset
__device__ __forceinline__ void rows_gather(const RowsOp & op, uint32_t xoff,
                                            int k0, float (&gv)[kRowsBlock]) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(op.tab), 0, (int)op.tab_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < kRowsBlock; ++j)
        gv[j] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(
                       rsrc, (int)xoff, (int)((uint32_t)(k0 + j) * op.row_bytes),
                       0));
}

// one op over kRowsBlock consecutive groups; gt = gtab + k0 (slot i of group
// k0 + j at gt[i * Kpad + j]).  xv: the row's value (ROP_NICH: its float
// bits; the gathers: the value times four)
template <int TYPE, bool LDSLOG>
__device__ __forceinline__ void rows_op(const RowsOp & op, int slot,
                                        uint32_t xv, uniform_fp gt, int Kpad,
                                        int k0, const uint32_t * log_tab,
                                        float (&s)[kRowsBlock]) {
#define GT(j, i) gt[(size_t)(i) * Kpad + (j)]
    if (TYPE == ROP_GATHER || TYPE == ROP_CAT) {
        float gv[kRowsBlock];
        rows_gather(op, xv, k0, gv);
#pragma unroll
        for (int j = 0; j < kRowsBlock; ++j) s[j] = s[j] + gv[j];
        if (TYPE == ROP_CAT) {   // dd.hpp:433-445: (acc + S) - shift
#pragma unroll
            for (int j = 0; j < kRowsBlock; ++j) s[j] = s[j] - GT(j, slot);
        }
    } else {
        const float x = u2f(xv);
        float temp[kRowsBlock], tl[kRowsBlock];
#pragma unroll
        for (int j = 0; j < kRowsBlock; ++j) {
            const float d = x - GT(j, slot + 3);
            temp[j] = 1.f + GT(j, slot + 2) * (d * d);
        }
        // FastLog::log (special.hpp:57-67): the table reads of the block
        // issued together
#pragma unroll
        for (int j = 0; j < kRowsBlock; ++j) {
            const uint32_t man = (f2u(temp[j]) >> 9) & 0x3FFFu;
            tl[j] = u2f(LDSLOG ? log_tab[man]
                               : g_tables_dev.log_table[man]);
        }
        // float(exponent - 127) in two instructions: the biased exponent is
        // shifted into the mantissa of 2^23 (temp >= 1: no sign bit), and
        // 2^23 + 127 comes off exactly
#pragma unroll
        for (int j = 0; j < kRowsBlock; ++j) {
            const float e = u2f(__builtin_amdgcn_alignbit(
                                0x258000u, f2u(temp[j]), 23)) - 8388735.0f;
            const float lg = (e + tl[j]) * 0.69314718055994529f;
            s[j] = s[j] + (GT(j, slot) + GT(j, slot + 1) * lg);
        }
    }
#undef GT
}

// SHAPE: the program's op types at compile time, base-4 digits from the first
// op (1 + type each; 0 ends the list); 0 = any program of up to kRowsMaxOps
// ops, their types tested at run time (wave-uniform branches)
constexpr int rows_shape_digit(int shape, int i) {
    return i == 0 ? shape % 4 : rows_shape_digit(shape / 4, i - 1);
}
constexpr int rows_shape_len(int shape) {
    return shape == 0 ? 0 : 1 + rows_shape_len(shape / 4);
}
// a program's gtab layout: slot 0 the driver's score, then per op ROP_CAT one
// slot, ROP_NICH four
constexpr int rows_shape_slot(int shape, int i) {   // first slot of op i
    int next = 1;
    for (int o = 0; o < i; ++o) {
        const int t = rows_shape_digit(shape, o) - 1;
        if (t == ROP_CAT) next += 1;
        if (t == ROP_NICH) next += 4;
    }
    return next;
}
constexpr int kShapeN = 1 + ROP_NICH;                      // one real
constexpr int kShapeG = 1 + ROP_GATHER;                    // GP / BB / BNB
constexpr int kShapeC = 1 + ROP_CAT;                       // DD / DPD
constexpr int kShapeGN = kShapeG + 4 * (1 + ROP_NICH);     // GP + NICH
constexpr int kShapeNN = kShapeN + 4 * (1 + ROP_NICH);     // two reals
constexpr int kRowsXv = 8;   // a row's values in registers

template <int SHAPE, int I, bool LDSLOG>
__device__ __forceinline__ void rows_shape_ops(
        const RowsArgs & A, const uint32_t (&xv)[kRowsXv], uniform_fp gt,
        int k0, const uint32_t * log_tab, float (&s)[kRowsBlock]) {
    if constexpr (I < rows_shape_len(SHAPE)) {
        rows_op<rows_shape_digit(SHAPE, I) - 1, LDSLOG>(
            A.op[I], rows_shape_slot(SHAPE, I), xv[I], gt, A.Kpad, k0,
            log_tab, s);
        rows_shape_ops<SHAPE, I + 1, LDSLOG>(A, xv, gt, k0, log_tab, s);
    }
}

// scores of groups [k0, k0 + kRowsBlock) for one row per lane; groups beyond
// the last take whatever their zeroed parameters give (the callers mask)
template <int SHAPE, bool LDSLOG>
__device__ __forceinline__ void rows_score_block(
        const RowsArgs & A, uniform_fp basep, const uint32_t (&xv)[kRowsXv],
        int k0, int g, float s_own, const uint32_t * log_tab,
        float (&s)[kRowsBlock]) {
    uniform_fp gt = as_uniform(A.gtab) + k0;
    const int W = A.Kpad;   // (the slot stride, as rows_op's GT wants it)
#pragma unroll
    for (int j = 0; j < kRowsBlock; ++j) s[j] = basep[k0 + j];
    if constexpr (SHAPE != 0) {
        rows_shape_ops<SHAPE, 0, LDSLOG>(A, xv, gt, k0, log_tab, s);
    } else {
        // (a rolled loop: the row's values are picked from their registers
        // by the wave-uniform op index, the op bodies exist once)
        for (int o = 0; o < A.n_ops; ++o) {
            const RowsOp & op = A.op[o];
            const uint32_t x = xv[o];
            if (op.type == ROP_GATHER)
                rows_op<ROP_GATHER, LDSLOG>(op, 0, x, gt, W, k0, log_tab, s);
            else if (op.type == ROP_CAT)
                rows_op<ROP_CAT, LDSLOG>(op, op.slot, x, gt, W, k0, log_tab,
                                         s);
            else
                rows_op<ROP_NICH, LDSLOG>(op, op.slot, x, gt, W, k0, log_tab,
                                          s);
        }
    }
    // the row's own slot (wave-uniform test first: most blocks hold no lane's)
    const int gl = g - k0;
    if (__any((unsigned)gl < (unsigned)kRowsBlock)) {
#pragma unroll
        for (int j = 0; j < kRowsBlock; ++j) s[j] = gl == j ? s_own : s[j];
    }
}

// the score of ONE group with a per-lane group index (vector loads; the same
// float operations as rows_score_block): the scan mode's second look at the
// kRowsSuper groups around a row's draw
template <bool LDSLOG>
__device__ __forceinline__ float rows_score_lane(
        const RowsArgs & A, const float * basep, const uint32_t (&xv)[kRowsXv],
        int k, int g, float s_own, const uint32_t * log_tab) {
    float s = basep[k];
#pragma unroll
    for (int o = 0; o < kRowsMaxOps; ++o) {
        if (o >= A.n_ops) break;
        const RowsOp & op = A.op[o];
        if (op.type == ROP_NICH) {
            const float * p = A.gtab + (size_t)op.slot * A.Kpad + k;
            const float x = u2f(xv[o]);
            const float d = x - p[3 * (size_t)A.Kpad];
            const float temp = 1.f + p[2 * (size_t)A.Kpad] * (d * d);
            const float lg = LDSLOG ? fast_log_t(temp, log_tab)
                                    : fast_log(temp);
            s = s + (p[0] + p[(size_t)A.Kpad] * lg);
        } else {
            s = s + *reinterpret_cast<const float *>(
                        reinterpret_cast<const char *>(op.tab)
                        + (size_t)k * op.row_bytes + xv[o]);
            if (op.type == ROP_CAT)
                s = s - A.gtab[(size_t)op.slot * A.Kpad + k];
        }
    }
    return k == g ? s_own : s;
}

// what a lane keeps of its row between the passes
struct RowsRow {
    uint32_t xv[kRowsXv];
    uint32_t code;     // the tile's joint value (folding), wave-uniform
    size_t out;        // batch-relative index (results, entropy)
    int g;             // own slot, -1 for a lane without a live row
    float s_own;
    bool live;
};

// SCAN: SCAN SAMPLING, tolerance-level and opt-in (option "sampling" = 1;
//         never the default).  One pass: every score is evaluated once (the
//         same float operations: the scores are the exact modes' bit for bit)
//         into a running log-sum-exp -- running maximum m, running sum S of
//         exp(s - m) rescaled whenever m grows, hardware exp2 -- with a
//         snapshot of (S, m) every kRowsSuper groups; the row's draw u (the
//         very engine step the exact modes use) is then located among the
//         snapshots and only the kRowsSuper groups around it are scored
//         again.  Same distribution as random.hpp:316-333 (first k with
//         cumulative likelihood >= u * total), different float summation
//         order: the index can differ from the exact modes' where u * total
//         falls within rounding of a boundary.
template <bool SCAN, bool LDSLOG, int SHAPE>
__global__ __launch_bounds__(kScratchMaxBlock) void k_rows_scratch(RowsArgs A) {
    __shared__ uint32_t s_exp[1024];                  // biased, see above
    __shared__ uint32_t s_log[LDSLOG ? 16384 : 1];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x)
        s_exp[i] = g_tables_dev.exp_table[i] | 0x3F800000u;
    if (LDSLOG)
        for (int i = threadIdx.x; i < 16384; i += blockDim.x)
            s_log[i] = g_tables_dev.log_table[i];
    __syncthreads();
    const float ea = u2f(g_tables_dev.exp_ab[0]);
    const float eb = u2f(g_tables_dev.exp_ab[1]);
    // (wave-uniform values the compiler cannot see as such are pinned to
    // scalar registers: loop control and table offsets stay on the scalar unit)
    const int K = __builtin_amdgcn_readfirstlane(A.dev ? A.dev->K : A.K);
    const int lane = threadIdx.x & 63;
    const size_t wave_slot =
        (size_t)blockIdx.x * (blockDim.x >> 6)
        + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K8 = (K + kRowsBlock - 1) & ~(kRowsBlock - 1);

    // work item w: 64 consecutive rows, or (folding) a tile of the
    // code-sorted row list
    auto load_row = [&](size_t w, RowsRow & r) {
        bool in;
        r.code = 0;
        if (A.fold) {
            const uint4 tile = A.tiles[w];
            r.code = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile.x);
            const uint32_t pos = tile.y, rows = tile.z;
            in = (uint32_t)lane < rows && r.code < A.fold_codes;
            r.out = in ? A.sorted_rows[pos + lane] : 0;
        } else {
            const size_t item = w * 64 + lane;
            in = item < A.n_items;
            r.out = in ? item : 0;
        }
        const size_t row = A.row_begin + r.out;
        const uint32_t slot = A.slot[r.out];   // k_row_prepass
        // (a handed-over row idles along on value 0: its own values may lie
        // outside the tables)
        r.live = in && slot != 0xFFFFFFFFu;
        r.g = r.live ? (int)slot : -1;
        r.s_own = A.own[r.out];
#pragma unroll
        for (int o = 0; o < kRowsXv; ++o) {
            r.xv[o] = 0;
            const bool used = SHAPE != 0 ? o < rows_shape_len(SHAPE)
                                         : o < A.n_ops;
            if (used && r.live) {
                const bool nich =
                    SHAPE != 0 ? rows_shape_digit(SHAPE, o) - 1 == ROP_NICH
                               : A.op[o].type == ROP_NICH;
                r.xv[o] = A.op[o].values[row] * (nich ? 1u : 4u);
            }
        }
    };
    // where a row's score starts: the driver's scores, or (folding) the
    // folded ops' scores of the wave's joint value
    auto base_of = [&](const RowsRow & r) -> uniform_fp {
        if (!A.fold) return as_uniform(A.gtab);
        const uint32_t code = r.code < A.fold_codes ? r.code : 0u;
        return as_uniform(A.fold) + (size_t)code * A.Kpad;
    };
    // one block of the max pass (vector_max, vector_math.cc:74-83; max is
    // order-free); groups beyond the last score -inf
    auto max_block = [&](const RowsRow & r, int k0, float & m) {
        float s[kRowsBlock];
        rows_score_block<SHAPE, LDSLOG>(A, base_of(r), r.xv, k0, r.g, r.s_own,
                                        s_log, s);
        if (k0 + kRowsBlock > K) {
#pragma unroll
            for (int j = 0; j < kRowsBlock; ++j)
                s[j] = k0 + j < K ? s[j] : -INFINITY;
        }
#pragma unroll
        for (int j = 0; j < kRowsBlock; ++j) m = fmaxf(m, s[j]);
    };
    // the likelihoods of one block (scores_to_likelihoods, random.cc:94-106),
    // +0 beyond the last group
    auto like_block = [&](const RowsRow & r, int k0, float m,
                          float (&s)[kRowsBlock]) {
        rows_score_block<SHAPE, LDSLOG>(A, base_of(r), r.xv, k0, r.g, r.s_own,
                                        s_log, s);
#pragma unroll
        for (int j = 0; j < kRowsBlock; ++j)
            s[j] = fast_exp_biased(s[j] - m, s_exp, ea, eb);
        if (k0 + kRowsBlock > K) {
#pragma unroll
            for (int j = 0; j < kRowsBlock; ++j)
                s[j] = k0 + j < K ? s[j] : 0.f;
        }
    };
    auto draw = [&](const RowsRow & r) {
        uint32_t xs = lcg_mulmod(A.seed_batch, A.pow_lo[r.out & 4095]);
        xs = lcg_mulmod(xs, A.pow_hi[r.out >> 12]);
        return lcg_unif01(xs);
    };

    const size_t stride = (size_t)gridDim.x * (blockDim.x >> 6);
    const size_t n_work = A.fold ? (size_t)A.n_tiles : (A.n_items + 63) / 64;
    size_t tile = wave_slot;   // the wave's work item (uniform)
    if (tile >= n_work) return;
    RowsRow cur;
    if constexpr (SCAN) {
        constexpr float kLog2e = 1.44269504088896341f;
        const int n_super = (K + kRowsSuper - 1) / kRowsSuper;
        float2 * snap =
            A.snap + wave_slot * (size_t)(A.Kpad / kRowsSuper) * 64 + lane;
        for (; tile < n_work; tile += stride) {
            load_row(tile, cur);
            float m = -INFINITY, S = 0.f;
            for (int k0 = 0; k0 < K8; k0 += kRowsBlock) {
                float s[kRowsBlock];
                rows_score_block<SHAPE, LDSLOG>(A, base_of(cur), cur.xv, k0,
                                                cur.g, cur.s_own, s_log, s);
                if (k0 + kRowsBlock > K) {
#pragma unroll
                    for (int j = 0; j < kRowsBlock; ++j)
                        s[j] = k0 + j < K ? s[j] : -INFINITY;
                }
                float bm = s[0];
#pragma unroll
                for (int j = 1; j < kRowsBlock; ++j) bm = fmaxf(bm, s[j]);
                const float m_new = fmaxf(m, bm);
                // (the first block: m = -inf, S = 0: exp2(-inf) = 0)
                S = S * __builtin_amdgcn_exp2f((m - m_new) * kLog2e);
                m = m_new;
                const float mc = -m * kLog2e;
#pragma unroll
                for (int j = 0; j < kRowsBlock; ++j)
                    S += __builtin_amdgcn_exp2f(
                        __builtin_fmaf(s[j], kLog2e, mc));
                if (((k0 + kRowsBlock) & (kRowsSuper - 1)) == 0
                    || k0 + kRowsBlock >= K8)
                    snap[(size_t)(k0 / kRowsSuper) * 64] = make_float2(S, m);
            }
            // locate the draw among the snapshots
            const float target = S * draw(cur);
            int b_sel = n_super - 1;
            float cum_before = 0.f, prev = 0.f;
            bool found = false;
            for (int b = 0; b < n_super; ++b) {
                const float2 v = snap[(size_t)b * 64];
                const float cum =
                    v.x * __builtin_amdgcn_exp2f((v.y - m) * kLog2e);
                if (!found && (cum >= target || b == n_super - 1)) {
                    found = true;
                    b_sel = b;
                    cum_before = prev;
                }
                prev = cum;
            }
            // ... and score its kRowsSuper groups again, lane by lane
            const float * basep = A.gtab;
            if (A.fold)
                basep = A.fold
                        + (size_t)(cur.code < A.fold_codes ? cur.code : 0u)
                              * A.Kpad;
            const float mc = -m * kLog2e;
            float cum = cum_before;
            int k_sel = -1;
            for (int j = 0; j < kRowsSuper; ++j) {
                const int k = b_sel * kRowsSuper + j;
                const int kc = k < K ? k : K - 1;
                const float sc = rows_score_lane<LDSLOG>(
                    A, basep, cur.xv, kc, cur.g, cur.s_own, s_log);
                if (k < K)
                    cum += __builtin_amdgcn_exp2f(
                        __builtin_fmaf(sc, kLog2e, mc));
                if (k_sel < 0 && k < K && cum >= target) k_sel = k;
            }
            if (k_sel < 0) {   // rounding left the block just short
                const int last = b_sel * kRowsSuper + kRowsSuper - 1;
                k_sel = last < K - 1 ? last : K - 1;
            }
            if (cur.live) A.new_packed[cur.out] = (uint32_t)k_sel;
        }
        return;
    }
    // the exact mode: three passes over the groups, every score evaluated in
    // each (the variants that kept the likelihoods, or the scores as well, in
    // an HBM scratch column between the passes were measured and lost: 42
    // instead of 60 instructions per (row, group), but 8 / 16 B of private
    // write-then-read traffic that tops out at 0.45 of the HBM roof:
    // profiles/r3_pmc_rows_scratch_mode1.txt)
    for (; tile < n_work; tile += stride) {
        load_row(tile, cur);
        float m = -INFINITY;
        for (int k0 = 0; k0 < K8; k0 += kRowsBlock) max_block(cur, k0, m);
        // total in index order (random.cc:100-103)
        float total = 0.f;
        for (int k0 = 0; k0 < K8; k0 += kRowsBlock) {
            float l[kRowsBlock];
            like_block(cur, k0, m, l);
#pragma unroll
            for (int j = 0; j < kRowsBlock; ++j) total += l[j];
        }
        // sample_from_likelihoods (random.hpp:316-333): t never increases, so
        // the index is the number of steps after which t is still positive
        // (entries beyond K are +0: they count only once t stayed positive
        // through K - 1, which the final clamp maps to K - 1 as well)
        float t = total * draw(cur);
        int steps = 0;
        for (int k0 = 0; k0 < K8; k0 += kRowsBlock) {
            float l[kRowsBlock];
            like_block(cur, k0, m, l);
#pragma unroll
            for (int j = 0; j < kRowsBlock; ++j) {
                t -= l[j];
                steps += t > 0.f ? 1 : 0;
            }
            if (!__any(cur.live && t > 0.f)) break;
        }
        if (cur.live)
            A.new_packed[cur.out] = (uint32_t)(steps < K - 1 ? steps : K - 1);
    }
}

// The two order-sensitive recurrences over a likelihood strip in LDS, computed
// redundantly by every lane of a wave (uniform-address LDS reads broadcast):
//   total = ((l_0 + l_1) + l_2) + ...              random.cc:100-103
//   t = total * u; t -= l_k until t <= 0           random.hpp:316-333
// The strip holds `n` entries followed by zeros up to a multiple of 64 (adding
// or subtracting +0 is exact).  64 entries arrive as 16 ds_read_b128, so the
// dependent chain is the VALU add alone.  t never increases: the scan walks
// whole chunks and replays only the chunk in which t crosses zero.
__device__ __forceinline__ float strip_total(const float * strip, int n) {
    float total = 0.f;
    for (int k0 = 0; k0 < n; k0 += 64) {
        float4 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            v[q] = *reinterpret_cast<const float4 *>(strip + k0 + 4 * q);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            total += v[q].x; total += v[q].y; total += v[q].z; total += v[q].w;
        }
    }
    return total;
}
__device__ __forceinline__ int strip_sample(const float * strip, int n,
                                            float t) {
    for (int k0 = 0; k0 < n; k0 += 64) {
        float4 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            v[q] = *reinterpret_cast<const float4 *>(strip + k0 + 4 * q);
        const float t0 = t;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            t -= v[q].x; t -= v[q].y; t -= v[q].z; t -= v[q].w;
        }
        if (!(t > 0.f)) {   // crossed inside this chunk: replay it, counting
            float tt = t0;
            int steps = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                tt -= v[q].x; steps += tt > 0.f ? 1 : 0;
                tt -= v[q].y; steps += tt > 0.f ? 1 : 0;
                tt -= v[q].z; steps += tt > 0.f ? 1 : 0;
                tt -= v[q].w; steps += tt > 0.f ? 1 : 0;
            }
            const int k = k0 + steps;
            return k < n - 1 ? k : n - 1;
        }
    }
    return n - 1;
}

// One row by one wave (k_rows_wave's body; k_vs_apply runs it for the rows
// its chunk was handed): `sl` = the wave's strip of LDS (K floats padded to a
// multiple of 64), `s_exp` = fmath's table in LDS, `out` = where in
// old_packed / new_packed the move is left.
template <int KIND0, int KIND1, int NF>
__device__ __forceinline__ void wave_row_update(
        const SweepParams & P, float * sl, const uint32_t * s_exp, float ea,
        float eb, int K, int lane, size_t row, uint32_t global_id,
        size_t out) {
    const RowScorer<KIND0, KIND1, NF> rs(P, row, global_id);
    const int Kl = rs.Kl;
    // scores and vector_max (vector_math.cc:74-83; max is order-free); four
    // slots per lane and round, so that their gathers are in flight together
    // (a round is a trip to memory: the row's latency is the rounds')
    float m = -INFINITY;
    constexpr int U = 4;
    for (int k0 = lane; k0 < Kl; k0 += 64 * U) {
        float s[U];
#pragma unroll
        for (int q = 0; q < U; ++q)
            s[q] = k0 + 64 * q < Kl ? rs.at_lane(k0 + 64 * q) : -INFINITY;
#pragma unroll
        for (int q = 0; q < U; ++q)
            if (k0 + 64 * q < Kl) {
                sl[k0 + 64 * q] = s[q];
                m = s[q] > m ? s[q] : m;
            }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    // scores_to_likelihoods: the exponentials in parallel ...
    for (int k = lane; k < ((Kl + 63) & ~63); k += 64)
        sl[k] = k < Kl ? fast_exp_nonpos(sl[k] - m, s_exp, ea, eb) : 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ... their total in index order, then the scan (strip_total /
    // strip_sample: every lane computes the same)
    const float total = strip_total(sl, Kl);
    int g2 = strip_sample(sl, Kl, total * batch_row_unif01(P, row));
    if (rs.singleton && g2 == rs.g) g2 = K - 1;   // slot g held group K-1
    if (lane == 0) {
        P.old_packed[out] = (uint32_t)rs.g;
        P.new_packed[out] = (uint32_t)g2;
    }
    __builtin_amdgcn_wave_barrier();   // before the strip is reused
}

// One WAVE per row, for the rows that come one at a time: the hand-overs of
// the value-sorted kernel, tiny batches, the sequential chain.  Lanes score 64
// slots at once (coalesced cache reads) and exponentiate them in parallel into
// the wave's LDS strip; only the two order-sensitive recurrences run serially
// (every lane computes the same sum over LDS broadcasts).  Same float
// operations as the lane-per-row kernel, a row's latency drops from ~3K
// dependent gather round trips to ~2K LDS-fed adds.
template <int KIND0, int KIND1, int NF>
__global__ __launch_bounds__(kBlock) void k_rows_wave(SweepParams P) {
    extern __shared__ float wave_lds[];
    __shared__ uint32_t s_exp[1024];
    {   // most launches find few rows or none: workgroups without one leave
        const size_t n = P.row_list ? (size_t)*P.row_list_count
                                    : P.row_end - P.row_begin;
        if ((size_t)blockIdx.x * (kBlock / 64) >= n) return;
    }
    for (int i = threadIdx.x; i < 1024; i += kBlock)
        s_exp[i] = g_tables_dev.exp_table[i];
    __syncthreads();
    const float ea = u2f(g_tables_dev.exp_ab[0]);
    const float eb = u2f(g_tables_dev.exp_ab[1]);
    const int K = sweep_K(P);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float * sl = wave_lds + (size_t)wave * ((K + 63) & ~63);

    const size_t n_items = P.row_list ? (size_t)*P.row_list_count
                                      : P.row_end - P.row_begin;
    const size_t stride = (size_t)gridDim.x * (kBlock / 64);
    for (size_t item = (size_t)blockIdx.x * (kBlock / 64) + wave;
         item < n_items; item += stride) {
        size_t out = item;
        size_t row = P.row_begin + item;
        uint32_t global_id;
        if (P.row_list && P.sorted_rows) {
            out = (size_t)P.row_list[item];
            row = P.row_begin + P.sorted_rows[out];
            global_id = P.assign_pos[out];
        } else if (P.row_list) {   // a list of batch rows, in row order
            out = (size_t)P.row_list[item];
            row = P.row_begin + out;
            global_id = P.assign[row];
        } else {
            global_id = P.assign[row];
        }
        wave_row_update<KIND0, KIND1, NF>(P, sl, s_exp, ea, eb, K, lane, row,
                                          global_id, out);
    }
}

// The reference's sequential chain, resident on the device: ONE workgroup
// walks rows [row_begin, row_end) one after the other -- remove the row from
// its group, score every group against the updated state, sample, add
// (examples/mixture/main.py:236-244 over mixture.hpp:376-425) -- so a row costs
// a few barriers instead of a dozen launches and a host round trip.  The kernel
// handles the rows that leave the group set alone and returns to the host at
// the first structural step, which the host performs with the batch code:
//   event 1: the next row is alone in its group (the group would vanish);
//            nothing has been done for it;
//   event 2: the last processed row filled an empty group (a new empty group
//            must be appended, clustering.hpp:163-176 / mixture.hpp:361-368).
// base[k] is the driver's score with the row taken out (k_sweep_prepare);
// the kernel keeps it, the group sizes, the statistics and the caches current.
struct ChainResult {
    uint32_t rng_state;
    uint32_t rows_done;
    int event;
    int pad;
};

// Group::add_value / remove_value plus the cache refresh of that group
// (k_slave_value_op as a device function).  Categorical kinds take the loads
// up front and the logarithms from the LDS copy of the table, so the update
// is one memory round trip, not five dependent ones.
__device__ __forceinline__ void chain_value_op(const SlaveView & s, int k,
                                               uint32_t value, bool add,
                                               const uint32_t * log_tab) {
    if (is_cat(s.kind)) {
        const size_t cell = (size_t)k * s.dim + value;
        const int c2 = s.cnt[cell] + (add ? 1 : -1);
        const int n2 = s.i0[k] + (add ? 1 : -1);
        const float prior = s.prior[value];
        s.cnt[cell] = c2;
        s.i0[k] = n2;
        // dd.hpp:458-467 / dpd.hpp:458-470
        s.S[(size_t)value * s.cap + k] = fast_log_t(prior + (float)c2, log_tab);
        s.c0[k] = fast_log_t(s.alpha_sum + (float)n2, log_tab);
        return;
    }
    Stats st = load_stats(s, k);
    if (add) stats_add(s.kind, st, value); else stats_remove(s.kind, st, value);
    store_stats(s, k, st);
    refresh_scalar_entry(s, k);
}

// INIT (the initialisation loops of examples/mixture/main.py:227-232 and
// 265-270): rows that have no group yet are ADDED one at a time -- score,
// sample, add; nothing is removed, the sample size grows with every row (so
// the driver's score is shifted[k] - fast_log(sample_size + alpha) afresh per
// row, clustering.hpp:195-208); 2: with the clustering model's score alone.
template <int KIND0, int KIND1, int NF, int INIT = 0>
__global__ __launch_bounds__(kBlock) void k_chain_rows(
        SweepParams P, float * __restrict__ base, int32_t * counts,
        uint32_t * assign, const uint32_t * __restrict__ p2g,
        uint32_t rng_state, ChainResult * result) {
    extern __shared__ float chain_lds[];   // [K] scores, then likelihoods
    __shared__ uint32_t s_exp[1024];
    __shared__ uint32_t s_log[16384];      // FastLog table: the per-row cache
    __shared__ float s_red[kBlock / 64];   // refreshes run on one thread
    __shared__ int s_g2, s_n2;
    for (int i = threadIdx.x; i < 1024; i += kBlock)
        s_exp[i] = g_tables_dev.exp_table[i];
    for (int i = threadIdx.x; i < 16384; i += kBlock)
        s_log[i] = g_tables_dev.log_table[i];
    __syncthreads();
    const float ea = u2f(g_tables_dev.exp_ab[0]);
    const float eb = u2f(g_tables_dev.exp_ab[1]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = sweep_K(P);
    const int nf = NF > 0 ? NF : P.F;
    const float shift = P.scalars->shift;
    float * sc = chain_lds;
    uint32_t done = 0;
    int event = 0;
    float * shifted = const_cast<float *>(P.shifted);
    for (size_t row = P.row_begin; row < P.row_end; ++row) {
        const int g = INIT ? 0 : P.g2p[assign[row]];
        const int n_g = INIT ? 0 : counts[g];
        if (!INIT && n_g == 1) { event = 1; break; }
        // INIT: the sample size this row is scored with
        const long long size_now = P.sample_size + (long long)done;
        const float shift_row = INIT ? py_shift(size_now, P.alpha) : 0.f;
        uint32_t x[kMaxF];
        float lf[kMaxF];
        int kind[kMaxF];
#pragma unroll
        for (int f = 0; f < (NF > 0 ? NF : kMaxF); ++f) {
            if (f >= nf) break;
            kind[f] = f == 0 && KIND0 >= 0 ? KIND0
                    : f == 1 && KIND1 >= 0 ? KIND1 : P.feat[f].kind;
            x[f] = P.values[f][row];
            lf[f] = kind[f] == DIST_GP ? fast_log_factorial(x[f]) : 0.f;
        }
        // remove_value (mixture.hpp:94-122,386-398; clustering.hpp:178-193)
        if (!INIT && tid == 0) {
            counts[g] = n_g - 1;
            base[g] = P.cluster == 0
                ? fast_log_t((float)(n_g - 1) - P.d, s_log) + shift
                : cluster_own_score(P, n_g - 1, shift);
            for (int f = 0; f < nf; ++f) {
                SlaveView v = P.feat[f];
                v.kind = kind[f];
                chain_value_op(v, g, x[f], false, s_log);
            }
        }
        __threadfence_block();
        __syncthreads();
        // score_value: driver, then every feature accumulates
        float m = -INFINITY;
        for (int k = tid; k < K; k += kBlock) {
            float s = base[k];
            if (INIT)   // clustering.hpp:195-208 / mixture.hpp:124-141
                s = P.cluster == 0
                    ? shifted[k] + shift_row
                    : le_score_add_value(P.dataset_size, counts[k],
                                         (int)size_now, P.n_empty);
#pragma unroll
            for (int f = 0; f < (NF > 0 ? NF : kMaxF); ++f) {
                if (f >= nf || INIT == 2) break;
                SlaveView v = P.feat[f];
                v.kind = kind[f];
                s = accumulate(kind[f], s, load_entry(v, k, x[f]), x[f],
                               lf[f], v.p);
            }
            sc[k] = s;
            m = s > m ? s : m;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(m, off);
            m = o > m ? o : m;
        }
        if (lane == 0) s_red[wave] = m;
        __syncthreads();
        m = s_red[0];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) m = s_red[w] > m ? s_red[w] : m;
        // scores_to_likelihoods (random.cc:94-106): exponentials in parallel
        for (int k = tid; k < ((K + 63) & ~63); k += kBlock)
            sc[k] = k < K ? fast_exp_nonpos(sc[k] - m, s_exp, ea, eb) : 0.f;
        __syncthreads();
        if (wave == 0) {
            const float total = strip_total(sc, K);
            rng_state = lcg_mulmod(rng_state, 16807u);
            const int g2 = strip_sample(sc, K, total * lcg_unif01(rng_state));
            if (lane == 0) {
                s_g2 = g2;
                s_n2 = counts[g2];
            }
        }
        __syncthreads();
        const int g2 = s_g2, n2 = s_n2;
        // add_value (mixture.hpp:73-92,376-384; clustering.hpp:163-176)
        if (tid == 0) {
            counts[g2] = n2 + 1;
            if (INIT)   // clustering.hpp:163-176, _update_nonempty_group
                shifted[g2] = fast_log_t((float)(n2 + 1) - P.d, s_log);
            base[g2] = P.cluster == 0
                ? fast_log_t((float)(n2 + 1) - P.d, s_log) + shift
                : cluster_own_score(P, n2 + 1, shift);
            for (int f = 0; f < nf; ++f) {
                SlaveView v = P.feat[f];
                v.kind = kind[f];
                chain_value_op(v, g2, x[f], true, s_log);
            }
            assign[row] = p2g[g2];
        }
        __threadfence_block();
        __syncthreads();
        done += 1;
        if (n2 == 0) { event = 2; break; }
    }
    if (tid == 0) {
        result->rng_state = rng_state;
        result->rows_done = done;
        result->event = event;
    }
}

// batch-semantics scores of one row, for tolerance tests of the scores
template <int KIND0, int KIND1, int NF>
__global__ void k_row_scores(SweepParams P, size_t row, float * out,
                             int * size_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const RowScorer<KIND0, KIND1, NF> rs(P, row, P.assign[row]);
    for (int k = 0; k < rs.Kl; ++k) out[k] = rs.at(k);
    *size_out = rs.Kl;
}

// score_values extension: out[r][k] against the current state, no removal
__global__ void k_score_rows(SweepParams P, float * __restrict__ out,
                             size_t ld) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (P.row_end - P.row_begin) * (size_t)P.K;
    if (i >= n) return;
    const size_t r = i / P.K;
    const int k = (int)(i % P.K);
    const size_t row = P.row_begin + r;
    float s = P.cluster == 1
        ? le_score_add_value(P.dataset_size, P.counts[k], (int)P.sample_size,
                             P.n_empty)
        : P.shifted[k] + P.scalars->shift_full;
    for (int f = 0; f < P.F; ++f) {
        const SlaveView & v = P.feat[f];
        const uint32_t x = P.values[f][row];
        const float lf = v.kind == DIST_GP ? fast_log_factorial(x) : 0.f;
        s = accumulate(v.kind, s, load_entry(v, k, x), x, lf, v.p);
    }
    out[r * ld + k] = s;
}

// ---------------------------------------------------------------------------
// The value-sorted row update (single feature with a small value domain:
// DD, DPD, BB).
//
// Rows with the same value x see the same score vector s_x[k] except in their
// own slot, and the own-slot score after self-removal never exceeds the
// unpatched one in exact arithmetic, so the softmax shift m of a row is
//   class A (own group is not the arg-max of s_x):  M[x]  = max_k s_x[k]
//   class B (own group is the arg-max of s_x):      mB[x] = max(s_own, M2[x])
// both functions of x alone.  The likelihood vectors
//   LA[x][k] = fast_exp(s_x[k] - M[x]),  LB[x][k] = fast_exp(s_x[k] - mB[x])
// are therefore computed once per value and batch (k_vs_prepare), and the
// per-row work shrinks to the two order-sensitive recurrences (running sum,
// subtractive scan) over wave-uniform inputs, with one per-lane exp for the
// own slot.  Rows are pre-sorted by value (static: values never change), one
// wave = one tile of <= 64 * kVsR rows of one value.  Every float operation a row
// performs is the one the generic kernel performs, in the same order; rows the
// shortcut does not cover exactly (group of one member; own-slot score above
// M[x] through table rounding; DPD OTHER) are handed to the generic kernel.

constexpr int kVsUnroll = 32;   // entries per scalar-loaded chunk
// rows per lane (a tile = 64 * kVsR rows of one value).  Two: the lane's two
// running values advance as one v_pk_add_f32 per entry, the entry selected
// into both halves from its scalar register (tools/microbench/pk_add.hip:
// 1.75x the rows per second of v_sub_f32, bit-identical)
constexpr int kVsR = 2;
// rows per apply work item (k_vs_apply), all of one value: a multiple of the
// tile sizes, so a tile's rows lie in one chunk
constexpr int kVsApplyRows = 4096;
struct VsTile {
    uint32_t x;      // the tile's value
    uint32_t pos;    // first position in the sorted row list
    uint32_t n;      // rows in the tile (<= 64 * kVsR)
    uint32_t chunk;  // the apply chunk (k_vs_apply work item) the rows lie in
};
// Where a tile leaves the rows its shortcut does not cover.  Either ONE list
// for the launch (`list`, `count`: a wave-per-row launch follows), or -- when
// `chunk_counts` is set -- a list per apply chunk, kept in the chunk's own
// stretch of `list` (positions chunks[c].pos ...): k_vs_apply then samples
// the handed-over rows of its chunk itself, before it adds up the moves, and
// no launch sits between the two kernels.
struct VsDefer {
    uint32_t * list;
    uint32_t * count;
    uint32_t * chunk_counts;
    const VsTile * chunks;
};
__device__ __forceinline__ void vs_hand_over(const VsDefer & D, uint32_t chunk,
                                             uint32_t at) {
    if (D.chunk_counts)
        D.list[D.chunks[chunk].pos + atomicAdd(&D.chunk_counts[chunk], 1u)] = at;
    else
        D.list[atomicAdd(D.count, 1u)] = at;
}
struct VsTables {
    float * LA;      // [nvals][Kpad]
    float * LB;
    float * M;       // [nvals]
    float * mB;
    int * argmax;    // [nvals], first index attaining the maximum
    int Kpad;
    // running sums of LA / LB at the chunk boundaries, in index order:
    // P[x][c] = ((l_0 + l_1) + ...) + l_{32c-1}; null = not built
    float * PA;      // [nvals][Kpad / kVsUnroll]
    float * PB;
    // Rows that sit in their value's arg-max group use LB.  A tile that holds
    // some next to others runs both passes -- one or two tiles per value, and
    // the SIMD that holds one sets the kernel's time.  In a group-sorted range
    // those rows are one contiguous band, so k_vs_prepare looks for it and, if
    // it is a band of at most one tile, gives it a tile of its own
    // (band_tile[x], band_mode[x] = 1): the value's regular tiles then skip
    // the band's rows and nobody runs two passes.  Otherwise band_mode[x] = 0
    // and the tiles do as before.  Null: not used for this launch.
    int * band_mode;              // [nvals]
    VsTile * band_tile;           // [nvals]
    const uint32_t * val_start;   // [nvals + 1] positions of each value's rows
    uint32_t n_values;
    // diagnostics (a -DDIST_VS_STAMPS build, `make stamps`, run with
    // DIST_VS_STAMPS=<file>; tools/vs_stamps.py): per wave of k_vs_sample
    // five s_memtime stamps and HW_ID; null otherwise
    unsigned long long * stamps;
    // [nvals + 1] index of the first apply chunk of each value (chunks of
    // one value each, kVsApplyRows rows apart: a band tile's rows find theirs)
    const uint32_t * chunk_first;
    // [nvals][Kpad] (k_vs_tables; null otherwise) the own-slot likelihood of a
    // row of value x that sits in group k, taken out of it -- what the tiles'
    // set-up computes per row from three gathers and three logarithms -- or
    // -1: the row is handed over
    float * own;
    // what the launch walks (speculative loads, k_vs_narrow's copies in LDS):
    // the bound on the group count at THIS batch, a multiple of kVsUnroll,
    // <= Kpad (which stays the run's row stride)
    int Kuse;
    // band_mode / band_tile entries: one per VALUE (k_vs_prepare's walk), or --
    // band_by_chunk, k_vs_tables -- one per apply CHUNK of the values inside
    // the tables (used by the values whose rows fit ONE chunk)
    uint32_t band_count;
    int band_by_chunk;
};
constexpr uint32_t kVsBandWalkRows = 8192;

// score of a row with value x at its own slot g after removing itself
__device__ __forceinline__ float vs_own_score(const SweepParams & P,
                                              const SlaveView & v, int g,
                                              int n_g, uint32_t x, float lf,
                                              float shift) {
    const float s = cluster_own_score(P, n_g - 1, shift);
    return accumulate(v.kind, s, entry_after_remove(v, g, x), x, lf, v.p);
}

// could group g hold a row with value x?  (only then is the own-slot score of
// (x, g) meaningful; a false positive is harmless: no lane uses the result)
__device__ __forceinline__ bool vs_group_has_value(const SlaveView & v, int g,
                                                   uint32_t x) {
    if (is_cat(v.kind)) return v.cnt[(size_t)g * v.dim + x] >= 1;
    if (v.kind == DIST_GP || v.kind == DIST_BNB)
        return (uint32_t)v.i0[g] >= 1u && (uint32_t)v.i1[g] >= x;
    return (x ? v.i0[g] : v.i1[g]) >= 1;   // BB: heads / tails
}

template <int KIND>
__global__ __launch_bounds__(kBlock) void k_vs_prepare(
        SweepParams P, VsTables T, uint32_t * deferred_count,
        uint32_t deferred_initial) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *deferred_count = deferred_initial;
    __shared__ float r_m1[kBlock / 64], r_m2[kBlock / 64];
    __shared__ int r_i1[kBlock / 64];
    __shared__ float sh_M, sh_mB;
    __shared__ uint32_t sh_lo, sh_hi, sh_n;
    __shared__ int sh_amax;
    extern __shared__ float s_l[];   // [2][Kpad] when the running sums are built
    const uint32_t x = blockIdx.x;
    SlaveView v = P.feat[0];
    v.kind = KIND;
    const int K = sweep_K(P);
    const float lf = KIND == DIST_GP ? fast_log_factorial(x) : 0.f;
    float * la = T.LA + (size_t)x * T.Kpad;
    float * lb = T.LB + (size_t)x * T.Kpad;
    // pass 1: scores, local (max, first arg-max, max of the rest)
    float m1 = -INFINITY, m2 = -INFINITY;
    int i1 = 0x7fffffff;
    for (int k = threadIdx.x; k < K; k += kBlock) {
        const float s =
            accumulate(KIND, P.base[k], load_entry(v, k, x), x, lf, v.p);
        la[k] = s;
        if (s > m1) { m2 = m1; m1 = s; i1 = k; }
        else if (s > m2) m2 = s;
    }
    // (max, first arg-max, max of the rest): shuffles within the wave, then
    // the first lane folds the waves' results
    auto fold = [](float & a1, float & a2, int & ai, float b1, float b2,
                   int bi) {
        if (a1 > b1 || (a1 == b1 && ai < bi)) {
            a2 = fmaxf(a2, b1);
        } else {
            a2 = fmaxf(b2, a1);
            a1 = b1;
            ai = bi;
        }
    };
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float b1 = __shfl_xor(m1, off), b2 = __shfl_xor(m2, off);
        const int bi = __shfl_xor(i1, off);
        fold(m1, m2, i1, b1, b2, bi);
    }
    if ((threadIdx.x & 63) == 0) {
        r_m1[threadIdx.x >> 6] = m1;
        r_m2[threadIdx.x >> 6] = m2;
        r_i1[threadIdx.x >> 6] = i1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w)
            fold(m1, m2, i1, r_m1[w], r_m2[w], r_i1[w]);
        r_m1[0] = m1; r_m2[0] = m2; r_i1[0] = i1;
    }
    if (threadIdx.x == 0) {
        const float M = r_m1[0];
        const int g = r_i1[0];
        float mB = M;
        const int n_g = P.counts[g];
        if (n_g >= 2 && vs_group_has_value(v, g, x)) {
            const float s_own =
                vs_own_score(P, v, g, n_g, x, lf, P.scalars->shift);
            mB = fmaxf(s_own, r_m2[0]);
        }
        T.M[x] = M; T.mB[x] = mB; T.argmax[x] = g;
        sh_M = M; sh_mB = mB;
        sh_amax = g;
        sh_lo = 0xFFFFFFFFu; sh_hi = 0u; sh_n = 0u;
    }
    __syncthreads();
    const float M = sh_M, mB = sh_mB;
    const float ea = u2f(g_tables_dev.exp_ab[0]);
    const float eb = u2f(g_tables_dev.exp_ab[1]);
    for (int k = threadIdx.x; k < T.Kpad; k += kBlock) {
        float a = 0.f, b = 0.f;
        if (k < K) {
            const float s = la[k];
            a = fast_exp_nonpos(s - M, g_tables_dev.exp_table, ea, eb);
            b = fast_exp_nonpos(s - mB, g_tables_dev.exp_table, ea, eb);
        }
        la[k] = a;
        lb[k] = b;
        if (T.PA) {
            s_l[k] = a;
            s_l[T.Kpad + k] = b;
        }
    }
    // Two jobs are left, and they run side by side:
    //  * waves 0 and 1, one lane each: the running sums.  The likelihood total
    //    of a row is the index-order sum with the row's own slot replaced
    //    (random.cc:100-103), so up to the first own slot of a tile it is the
    //    same number for every row of the value: the lane walks the vector
    //    once (a dependent chain of Kpad adds, fed from the copy in LDS one
    //    chunk ahead) and leaves the running sum at each chunk boundary;
    //    k_vs_sample starts there.
    //  * the other waves (all of them without running sums): the positions of
    //    this value's rows in the arg-max group (VsTables::band_tile) --
    //    first, last, how many; four loads in flight per thread.
    if (T.PA == nullptr && T.band_mode == nullptr) return;
    __syncthreads();   // s_l is complete
    const int wave = threadIdx.x >> 6;
    const bool chains = T.PA != nullptr;
    bool walk = false;
    if (chains && wave < 2) {
        if ((threadIdx.x & 63) == 0) {
            const float4 * src =
                reinterpret_cast<const float4 *>(s_l + wave * T.Kpad);
            const int nchunks = T.Kpad / kVsUnroll;
            float * dst = (wave ? T.PB : T.PA) + (size_t)x * nchunks;
            constexpr int Q = kVsUnroll / 4;
            float4 even[Q], odd[Q];   // ping-pong: no register copies
#pragma unroll
            for (int q = 0; q < Q; ++q) even[q] = src[q];
            float run = 0.f;
            auto add_chunk = [&run](const float4 (&v)[Q]) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    run += v[q].x;
                    run += v[q].y;
                    run += v[q].z;
                    run += v[q].w;
                }
            };
            for (int c = 0; c < nchunks; c += 2) {
                const int c1 = c + 1 < nchunks ? c + 1 : c;
#pragma unroll
                for (int q = 0; q < Q; ++q) odd[q] = src[c1 * Q + q];
                __builtin_amdgcn_sched_barrier(0);   // loads first
                dst[c] = run;
                add_chunk(even);
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 >= nchunks) break;
                const int c2 = c + 2 < nchunks ? c + 2 : c;
#pragma unroll
                for (int q = 0; q < Q; ++q) even[q] = src[c2 * Q + q];
                __builtin_amdgcn_sched_barrier(0);
                dst[c + 1] = run;
                add_chunk(odd);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (T.band_mode) {
        const uint32_t begin = T.val_start[x];
        // (a value with very many rows has more than a tile of them in any
        // group, and walking them here would cost more than it can save)
        walk = T.val_start[x + 1] - begin <= kVsBandWalkRows;
        const uint32_t end = walk ? T.val_start[x + 1] : begin;
        const uint32_t amax = (uint32_t)sh_amax;
        const uint32_t first = chains ? 128u : 0u;   // walking threads
        const uint32_t step = kBlock - first;
        constexpr int U = 4;
        for (uint32_t base = begin + (threadIdx.x - first); base < end;
             base += U * step) {
            uint32_t gid[U], slot[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const uint32_t i = base + q * step;
                gid[q] = i < end ? P.assign_pos[i] : 0u;
            }
#pragma unroll
            for (int q = 0; q < U; ++q) slot[q] = (uint32_t)P.g2p[gid[q]];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const uint32_t i = base + q * step;
                if (i < end && slot[q] == amax) {
                    atomicMin(&sh_lo, i);
                    atomicMax(&sh_hi, i);
                    atomicAdd(&sh_n, 1u);
                }
            }
        }
    }
    if (T.band_mode == nullptr) return;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) {   // (a walking thread: it knows `walk`)
        const uint32_t n = sh_n;
        const bool band = n > 0 && sh_hi - sh_lo + 1u == n
                          && n <= 64u * kVsR;
        T.band_mode[x] = (walk && (band || n == 0)) ? 1 : 0;
        T.band_tile[x] = VsTile{x, band ? sh_lo : 0u, band ? n : 0u};
    }
}

// ---------------------------------------------------------------------------
// ONE launch between a batch's statistics and the next batch's sampling
// (device-normalised runs of the value-sorted path, integer statistics):
// k_normalise, k_batch_finish and k_vs_prepare in one kernel, so that a
// sub-sweep is tables -> sample -> apply -> reduce.  A launch costs this chip
// 2.4 us and every dependent trip to memory inside one about a microsecond
// (profiles/r4_launch_cost.txt: a grid barrier costs 7-12 us, a last-block
// ticket no less than the launch it saves), so the three kernels' work is
// done by the workgroups of the per-value tables REDUNDANTLY where it is
// cheap, and nobody waits for anybody:
//  * every workgroup compares the group sizes with those at batch entry and
//    derives the normalisation of the group set for itself (mixture.hpp:84-89,
//    108-119, as k_normalise does): which groups vanish, which survivor fills
//    which vacated slot, how many empty groups are appended;
//  * the per-group statistics are read through that plan from the IN buffers
//    (counts, i0, i1: what the last batch left) and never written there;
//    workgroup 0 writes them, normalised, to the OUT buffers, which the
//    batch's other kernels use (the host swaps the two after the launch), with
//    the driver's scores (clustering.hpp:151-161, 215-230), the cache entries,
//    the id maps (mixture.hpp:474-497) and the new DevState;
//  * workgroup x owns column x of the categorical counts and of the cache:
//    it moves / clears the cells of moved / appended groups in place and
//    writes S[x][.] (dd.hpp:399-421);
//  * then the value's tables as k_vs_prepare builds them, from the scores it
//    has in LDS: the same float operations in the same order.
// Where k_vs_apply (sorting form) leaves, per chunk, the position at which
// each group's rows begin after its sort -- off[c * stride + k], k <= the
// host's bound on the group count -- stamped with the run's removal epoch
// (DevState::pad: packed indices mean the same as long as no group was
// swap-removed).  k_vs_tables reads the arg-max group's band of rows from it
// instead of walking the value's rows.
struct VsOffsets {
    int * off;
    uint32_t * epoch;   // [chunks]; 0 = no offsets
    int stride;
};
// A batch that swap-removes groups changes what packed indices mean
// (Packed_::packed_remove, vector.hpp:47-51: the last group moves into the
// vacated slot).  Offsets recorded under an older epoch stay usable through
// the log of those moves: one entry per epoch -- {epoch, groups left after the
// removals, moves, (dst, src) pairs} -- in a ring; a reader walks it backwards
// from the current index to the index the group had when the chunk was
// sorted.  More epochs back than the ring holds, or more moves in one batch
// than an entry does: no band for that chunk this time.
constexpr int kRemapEpochs = 64;
constexpr int kRemapPairs = 4;
constexpr int kRemapEntry = 4 + 2 * kRemapPairs;   // ints per entry
struct TablesParams {
    SlaveView feat;              // i0 / i1: the OUT buffers
    const int32_t * i0_in;
    const int32_t * i1_in;
    const int32_t * counts_in;
    int32_t * counts_out;
    const int32_t * snap_in;     // group sizes at the last batch's entry
    int32_t * snap_out;
    const DevState * dev_in;
    DevState * dev_out;
    float * shifted;
    float * base;
    float * base_single;
    SweepScalars * scalars;
    uint32_t * p2g;
    int32_t * g2p;
    float alpha, d;
    int n_empty;                 // invariant of the chain
    long long sample_size;       // rows in the mixture (invariant)
    VsOffsets offsets;           // (off == nullptr: none recorded)
    int * remap_log;             // [kRemapEpochs][kRemapEntry]
    // what the group count can be at most at THIS launch (the run's bound,
    // T.Kpad, sizes the buffers; a run that stays open for many sweeps would
    // otherwise have every launch walk the whole bound)
    int k_limit;
};
constexpr int kTablesBlock = 1024;
constexpr int kTablesPer = 8;            // groups per thread
constexpr int kTablesMaxK = kTablesBlock * kTablesPer;
template <int KIND>
__global__ __launch_bounds__(kTablesBlock) void k_vs_tables(TablesParams A,
                                                            VsTables T) {
    // [Kpad] LA | [Kpad] LB for the running sums | the plan of a batch that
    // swap-removes groups: [Kpad + 2] vanished-before | [Kpad] the slot each
    // slot's group comes from
    extern __shared__ float tb_lds[];
    constexpr int kWaves = kTablesBlock / 64;
    __shared__ float r_m1[kWaves], r_m2[kWaves];
    __shared__ int r_i1[kWaves];
    __shared__ int s_sum[2][kWaves];
    __shared__ float sh_so;
    __shared__ int s_log[kRemapEpochs * kRemapEntry];
    const int Kpad = T.Kpad;
    const uint32_t x = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SlaveView v = A.feat;
    v.kind = KIND;
    // ---- everything the kernel reads per group, in ONE trip to memory: the
    // loads assume that no group vanished (slot k's group stays in slot k),
    // the usual case; a batch that swap-removed groups reads again below
    int was[kTablesPer], now[kTablesPer], st0[kTablesPer], st1[kTablesPer],
        cell[kTablesPer];
#pragma unroll
    for (int e = 0; e < kTablesPer; ++e) {
        const int k = tid + e * kTablesBlock;
        was[e] = now[e] = st0[e] = st1[e] = cell[e] = 0;
        if (k < A.k_limit) {   // (<= Kpad: the buffers are that large)
            was[e] = A.snap_in[k];
            now[e] = A.counts_in[k];
            st0[e] = A.i0_in[k];
            st1[e] = A.i1_in[k];
            if (is_cat(KIND)) cell[e] = v.cnt[(size_t)k * v.dim + x];
        }
    }
    // (the log of earlier batches' moves, for the bands at the end)
    if (T.band_mode)
        for (int i = tid; i < kRemapEpochs * kRemapEntry; i += kTablesBlock)
            s_log[i] = A.remap_log[i];
    const float prior_x = is_cat(KIND) ? v.prior[x] : 0.f;
    const int K0 = A.dev_in->K;
    const uint32_t global_size0 = A.dev_in->global_size;
    const uint32_t epoch0 = (uint32_t)A.dev_in->pad;
    // ---- the plan: vanished and filled groups since the last batch's entry
    int removed = 0, n_created = 0;
    {
        int e_sum = 0, c_sum = 0;
#pragma unroll
        for (int e = 0; e < kTablesPer; ++e) {
            const int k = tid + e * kTablesBlock;
            if (k < K0) {
                e_sum += (was[e] > 0 && now[e] == 0);
                c_sum += (was[e] == 0 && now[e] > 0);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            e_sum += __shfl_xor(e_sum, off);
            c_sum += __shfl_xor(c_sum, off);
        }
        if (lane == 0) { s_sum[0][wave] = e_sum; s_sum[1][wave] = c_sum; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            removed += s_sum[0][w];
            n_created += s_sum[1][w];
        }
    }
    const int size = K0 - removed;
    const int k_new = size;
    const int K1 = size + n_created;
    const int nonempty = K1 - A.n_empty;
    int * before = reinterpret_cast<int *>(tb_lds + 2 * (size_t)Kpad);
    int * src_of = before + Kpad + 2;
    auto emptied_at = [&](int k) {
        return A.snap_in[k] > 0 && A.counts_in[k] == 0;
    };
    if (removed > 0) {
        // before[k] = vanished groups in [0, k); then, as k_normalise: the i-th
        // removal (descending slots) pulls in what sits in slot K0 - 1 - i at
        // that time, so a vacated slot in front of the new end follows that
        // chain to the survivor it ends up with
        __syncthreads();   // (s_sum is reused)
        const int per = (K0 + kTablesBlock - 1) / kTablesBlock;
        const int lo = min(K0, tid * per), hi = min(K0, lo + per);
        int mine = 0;
        for (int k = lo; k < hi; ++k) mine += emptied_at(k);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) s_sum[0][wave] = incl;
        __syncthreads();
        int run = incl - mine;
        for (int w = 0; w < wave; ++w) run += s_sum[0][w];
        for (int k = lo; k < hi; ++k) {
            before[k] = run;
            run += emptied_at(k);
        }
        if (hi == K0 && lo < K0) before[K0] = run;
        __syncthreads();
        for (int k = tid; k < size; k += kTablesBlock) {
            int t = k;
            if (before[k + 1] != before[k]) {
                do {
                    t = K0 - 1 - (removed - before[t + 1]);
                } while (before[t + 1] != before[t]);
            }
            src_of[k] = t;
        }
        __syncthreads();
        // the statistics again, through the plan; this value's column of the
        // categorical counts follows the moved groups in place (all reads
        // before any write: a source slot may be cleared below)
#pragma unroll
        for (int e = 0; e < kTablesPer; ++e) {
            const int k = tid + e * kTablesBlock;
            if (k < size) {
                const int t = src_of[k];
                now[e] = A.counts_in[t];
                st0[e] = A.i0_in[t];
                st1[e] = A.i1_in[t];
                if (is_cat(KIND)) cell[e] = v.cnt[(size_t)t * v.dim + x];
            }
        }
        __syncthreads();
        if (is_cat(KIND)) {
#pragma unroll
            for (int e = 0; e < kTablesPer; ++e) {
                const int k = tid + e * kTablesBlock;
                if (k < size && src_of[k] != k)
                    v.cnt[(size_t)k * v.dim + x] = cell[e];
            }
        }
    }
    // ---- every group's cache entry, its score for this value (k_vs_prepare's
    // pass 1), its own-slot score; appended groups are empty (Group::init,
    // dd.hpp:113-121)
    const bool owner = x == 0;
    const float shift = py_shift(A.sample_size - 1, A.alpha);
    const float empty_score = py_empty_score(A.alpha, A.d, nonempty, A.n_empty);
    const float empty_single =
        py_empty_score(A.alpha, A.d, nonempty - 1, A.n_empty);
    const float lf = KIND == DIST_GP ? fast_log_factorial(x) : 0.f;
    float sc[kTablesPer], so[kTablesPer];
    float m1 = -INFINITY, m2 = -INFINITY;
    int i1 = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < kTablesPer; ++e) {
        const int k = tid + e * kTablesBlock;
        sc[e] = 0.f;
        so[e] = INFINITY;
        if (k >= K1) continue;
        const bool fresh = k >= k_new;
        if (fresh) {
            now[e] = st0[e] = st1[e] = cell[e] = 0;
            if (is_cat(KIND)) v.cnt[(size_t)k * v.dim + x] = 0;
        }
        const int n = now[e];
        const Stats st = {st0[e], st1[e], 0.f, 0.f};
        const int c = cell[e];
        Entry en = {0.f, 0.f, 0.f, 0.f};
        if (is_cat(KIND)) {
            en.c0 = fast_log(v.alpha_sum + (float)st.i0);
            en.c1 = fast_log(prior_x + (float)c);
            v.S[(size_t)x * v.cap + k] = en.c1;
        } else {
            en = scorer_init(KIND, v.p, st);
        }
        const float shifted = n ? py_nonempty_score(n, A.d) : empty_score;
        const float base = shifted + shift;
        const float s = accumulate(KIND, base, en, x, lf, v.p);
        sc[e] = s;
        if (s > m1) { m2 = m1; m1 = s; i1 = k; }
        else if (s > m2) m2 = s;
        // the score a row of this value sees in its own slot k once it is
        // taken out (vs_own_score); +inf: no such row or score, -inf: the row
        // would be alone (handed over)
        bool has;
        if (is_cat(KIND)) has = c >= 1;
        else if (KIND == DIST_GP || KIND == DIST_BNB)
            has = (uint32_t)st.i0 >= 1u && (uint32_t)st.i1 >= x;
        else has = (x ? st.i0 : st.i1) >= 1;
        if (n == 1) {
            so[e] = -INFINITY;
        } else if (n >= 2 && has) {
            Entry er = {0.f, 0.f, 0.f, 0.f};
            if (is_cat(KIND)) {
                er.c0 = fast_log(v.alpha_sum + (float)(st.i0 - 1));
                er.c1 = fast_log(prior_x + (float)(c - 1));
            } else {
                Stats s2 = st;
                stats_remove(KIND, s2, x);
                er = scorer_init(KIND, v.p, s2);
            }
            so[e] = accumulate(KIND, py_nonempty_score(n - 1, A.d) + shift,
                               er, x, lf, v.p);
        }
        if (owner) {
            A.counts_out[k] = n;
            A.snap_out[k] = n;
            v.i0[k] = st.i0;
            v.i1[k] = st.i1;
            if (fresh) { v.f0[k] = 0.f; v.f1[k] = 0.f; }
            A.shifted[k] = shifted;
            A.base[k] = base;
            A.base_single[k] = (n == 0 ? empty_single : shifted) + shift;
            v.c0[k] = en.c0;
            if (!is_cat(KIND)) {
                v.c1[k] = en.c1; v.c2[k] = en.c2; v.c3[k] = en.c3;
            }
        }
    }
    // (max, first arg-max, max of the rest) over the workgroup: within the
    // wave by shuffles, the waves' results folded by every thread for itself
    auto fold = [](float & a1, float & a2, int & ai, float b1, float b2,
                   int bi) {
        if (a1 > b1 || (a1 == b1 && ai < bi)) {
            a2 = fmaxf(a2, b1);
        } else {
            a2 = fmaxf(b2, a1);
            a1 = b1;
            ai = bi;
        }
    };
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float b1 = __shfl_xor(m1, off), b2 = __shfl_xor(m2, off);
        const int bi = __shfl_xor(i1, off);
        fold(m1, m2, i1, b1, b2, bi);
    }
    if (lane == 0) { r_m1[wave] = m1; r_m2[wave] = m2; r_i1[wave] = i1; }
    __syncthreads();
    m1 = r_m1[0]; m2 = r_m2[0]; i1 = r_i1[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) fold(m1, m2, i1, r_m1[w], r_m2[w], r_i1[w]);
    const float M = m1;
    const int amax = i1;
    // (the arg-max group's rows: own score against the rest's maximum)
    if (tid == (amax & (kTablesBlock - 1))) {
        float own_g = INFINITY;
#pragma unroll
        for (int e = 0; e < kTablesPer; ++e)
            if (e == amax / kTablesBlock) own_g = so[e];
        sh_so = own_g;
    }
    __syncthreads();
    const float so_g = sh_so;
    const float mB = (so_g != INFINITY && so_g != -INFINITY) ? fmaxf(so_g, m2)
                                                             : M;
    if (tid == 0) { T.M[x] = M; T.mB[x] = mB; T.argmax[x] = amax; }
    const float ea = u2f(g_tables_dev.exp_ab[0]);
    const float eb = u2f(g_tables_dev.exp_ab[1]);
    float * la = T.LA + (size_t)x * Kpad;
    float * lb = T.LB + (size_t)x * Kpad;
    float * lds_a = tb_lds;
    float * lds_b = tb_lds + Kpad;
    const bool chains = T.PA != nullptr;
    // (the tiles read whole chunks of kVsUnroll entries up to the group count:
    // that far the vectors are written, zeros behind the last group)
    const int Kw = min(Kpad, (K1 + kVsUnroll - 1) / kVsUnroll * kVsUnroll);
#pragma unroll
    for (int e = 0; e < kTablesPer; ++e) {
        const int k = tid + e * kTablesBlock;
        if (k >= Kw) continue;
        float a = 0.f, b = 0.f, o = -1.f;
        if (k < K1) {
            a = fast_exp_nonpos(sc[e] - M, g_tables_dev.exp_table, ea, eb);
            b = fast_exp_nonpos(sc[e] - mB, g_tables_dev.exp_table, ea, eb);
            // the row's own-slot likelihood (k_vs_sample's set-up): -1 = the
            // row is handed over (alone in its group, or its own score above
            // the value's maximum through table rounding)
            const float s_own = so[e];
            if (s_own != INFINITY && s_own != -INFINITY) {
                const bool class_b = k == amax;
                if (class_b || !(s_own > M))
                    o = fast_exp_nonpos(s_own - (class_b ? mB : M),
                                        g_tables_dev.exp_table, ea, eb);
            }
        }
        la[k] = a;
        lb[k] = b;
        if (T.own) T.own[(size_t)x * Kpad + k] = o;
        if (chains) { lds_a[k] = a; lds_b[k] = b; }
    }
    // ---- workgroup 0: the id maps (mixture.hpp:474-497), the scalars, the
    // new state
    if (owner) {
        if (removed > 0) {
            // the ids of the vanished groups retire before any slot is
            // overwritten
            for (int k = tid; k < K0; k += kTablesBlock)
                if (emptied_at(k)) A.g2p[A.p2g[k]] = -1;
            __syncthreads();
            for (int k = tid; k < size; k += kTablesBlock) {
                const int t = src_of[k];
                if (t != k) {
                    const uint32_t gid = A.p2g[t];
                    A.p2g[k] = gid;
                    A.g2p[gid] = k;
                }
            }
            __syncthreads();
        }
        for (int k = k_new + tid; k < K1; k += kTablesBlock) {
            const uint32_t gid = global_size0 + (uint32_t)(k - k_new);
            A.p2g[k] = gid;
            A.g2p[gid] = k;
        }
        if (removed > 0) {
            // this batch's moves into the log, under the epoch it begins
            int * entry = A.remap_log
                          + (size_t)((epoch0 + 1u) % kRemapEpochs) * kRemapEntry;
            if (tid == 0) s_sum[1][0] = 0;
            __syncthreads();
            for (int k = tid; k < size; k += kTablesBlock)
                if (src_of[k] != k) {
                    const int j = atomicAdd(&s_sum[1][0], 1);
                    if (j < kRemapPairs) {
                        entry[4 + 2 * j] = k;
                        entry[5 + 2 * j] = src_of[k];
                    }
                }
            __syncthreads();
            if (tid == 0) {
                entry[0] = (int)(epoch0 + 1u);
                entry[1] = size;
                entry[2] = s_sum[1][0];
            }
        }
        if (tid == 0) {
            DevState st;
            st.K = K1;
            st.k_new = k_new;
            st.created = n_created;
            st.removed = removed;
            st.global_size = global_size0 + (uint32_t)n_created;
            st.first_new_global = global_size0;
            st.nonempty = nonempty;
            st.pad = (int)(epoch0 + (removed > 0 ? 1u : 0u));
            *A.dev_out = st;
            A.scalars->shift = shift;
            A.scalars->shift_full = py_shift(A.sample_size, A.alpha);
            A.scalars->empty_single = empty_single;
        }
    }
    // ---- the arg-max group's band of rows in each of the value's chunks
    // (VsTables::band_tile), from the offsets the chunk's last sort left,
    // under the index the group had then (the moves since: this batch's plan,
    // then the log, newest first)
    if (T.band_mode) {
        __syncthreads();   // (s_log)
        const uint32_t c0 = T.chunk_first[x], c1 = T.chunk_first[x + 1];
        for (uint32_t c = c0 + tid; c < c1; c += kTablesBlock) {
            const uint32_t pos = T.val_start[x] + (c - c0) * (uint32_t)kVsApplyRows;
            int mode = 0;
            VsTile band = VsTile{x, 0u, 0u, c};
            const uint32_t then = A.offsets.off ? A.offsets.epoch[c] : 0u;
            // (values of several chunks -- Zipf's head -- keep to their
            // regular tiles: a band tile per chunk of theirs was measured,
            // k_vs_sample 105 against 86 us on Zipf(1.1) values: forty more
            // tiles of full chain length for a handful of rows each, ahead
            // of everything else in the launch)
            if (then != 0u && epoch0 - then < (uint32_t)kRemapEpochs
                && c1 - c0 == 1) {
                // the arg-max group's index when the chunk was sorted; -1: it
                // did not exist then (no rows of it here)
                bool known = true;
                int a = amax;
                if (removed > 0)
                    a = a < size ? src_of[a] : -1;   // (>= size: appended now)
                for (uint32_t e = epoch0; known && a >= 0 && e != then; --e) {
                    const int * entry = s_log + (e % kRemapEpochs) * kRemapEntry;
                    if ((uint32_t)entry[0] != e || entry[2] > kRemapPairs) {
                        known = false;
                    } else if (a >= entry[1]) {
                        a = -1;   // appended by that batch, or later
                    } else {
                        for (int j = 0; j < entry[2]; ++j)
                            if (a == entry[4 + 2 * j]) {
                                a = entry[5 + 2 * j];
                                break;
                            }
                    }
                }
                if (known) {
                    const int * off = A.offsets.off + (size_t)c * A.offsets.stride;
                    const int k_then = off[A.offsets.stride - 1];
                    uint32_t lo = 0u, hi = 0u;
                    if (a >= 0 && a < k_then) {
                        lo = (uint32_t)off[a];
                        hi = (uint32_t)off[a + 1];
                    }
                    if (hi - lo <= 64u * kVsR) {
                        mode = 1;
                        band = VsTile{x, pos + lo, hi - lo, c};
                    }
                }
            }
            T.band_mode[c] = mode;
            T.band_tile[c] = band;
        }
    }
    // ---- the running sums at the chunk boundaries (see k_vs_prepare): two
    // lanes walk the copies in LDS
    if (!chains) return;
    __syncthreads();
    if (wave < 2 && lane == 0) {
        const float4 * src =
            reinterpret_cast<const float4 *>(wave ? lds_b : lds_a);
        const int nchunks = Kw / kVsUnroll;
        float * dst = (wave ? T.PB : T.PA) + (size_t)x * (Kpad / kVsUnroll);
        constexpr int Q = kVsUnroll / 4;
        float4 even[Q], odd[Q];   // ping-pong: no register copies
#pragma unroll
        for (int q = 0; q < Q; ++q) even[q] = src[q];
        float run = 0.f;
        auto add_chunk = [&run](const float4 (&w)[Q]) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                run += w[q].x;
                run += w[q].y;
                run += w[q].z;
                run += w[q].w;
            }
        };
        for (int c = 0; c < nchunks; c += 2) {
            const int c1 = c + 1 < nchunks ? c + 1 : c;
#pragma unroll
            for (int q = 0; q < Q; ++q) odd[q] = src[c1 * Q + q];
            __builtin_amdgcn_sched_barrier(0);   // loads first
            dst[c] = run;
            add_chunk(even);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 >= nchunks) break;
            const int c2 = c + 2 < nchunks ? c + 2 : c;
#pragma unroll
            for (int q = 0; q < Q; ++q) even[q] = src[c2 * Q + q];
            __builtin_amdgcn_sched_barrier(0);
            dst[c + 1] = run;
            add_chunk(odd);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// The two order-sensitive recurrences for the lanes whose likelihood vector
// is `lp` (wave-uniform), own slot replaced by the lane's l_own:
//   total = ((l_0 + l_1) + l_2) + ...                  random.cc:100-103
//   t = total*u; t -= l_k until t <= 0                 random.hpp:323-330
// The vector is consumed in chunks of kVsUnroll scalar-loaded entries.  A
// chunk into which no lane's own slot falls is pure uniform arithmetic (one
// packed VALU op per entry and pass); otherwise its eight-entry pieces that
// hold an own slot take the per-lane select.  Subtracting non-negative terms
// never increases t, so each lane crosses zero in exactly one chunk; the scan
// only records that chunk and the value of t on entry, and the lane then
// replays its kVsUnroll subtractions to get the exact index.  With `prefix`
// (the value's running sums at the chunk boundaries, k_vs_prepare) the total
// starts at the tile's first own chunk.
// Tables far larger than the scalar cache (C5: 328 MB) stream through the same
// scalar loads: a coalesced-vector-load + v_readlane variant measured 1.2-1.7x
// slower at every table size and was dropped.
__device__ __forceinline__ void vs_fetch_chunk(uniform_fp lp, int k0,
                                               float (&l)[kVsUnroll]) {
#pragma unroll
    for (int j = 0; j < kVsUnroll; ++j) l[j] = lp[k0 + j];
}
typedef float v2f __attribute__((ext_vector_type(2)));
static_assert(kVsR == 2, "the recurrences below are written for two rows per "
                         "lane (one v_pk_add_f32 per entry)");
__device__ __forceinline__ v2f vs_splat(float x) { return (v2f){x, x}; }
// One eight-entry piece of the likelihood vector into which own slots fall
// (entries k0 .. k0+7, `l` wave-uniform): acc (+/-)= the entry, a lane's own
// slot replaced by its l_own.  (A form that looks for the one entry in
// question first -- ballots, readlane, a wave-uniform index -- was tried and
// measured slower at every batch size: the compiler turns the uniform
// branches back into selects, two per entry as here, and the search is
// extra.)
template <bool SUB>
__device__ __forceinline__ void vs_own_piece(
        v2f & acc, const float (&l)[8], int k0, const int (&g)[kVsR],
        const float (&l_own)[kVsR]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const v2f e = {(k0 + j == g[0]) ? l_own[0] : l[j],
                       (k0 + j == g[1]) ? l_own[1] : l[j]};
        acc = SUB ? acc - e : acc + e;
    }
}
__device__ __forceinline__ void vs_sum_and_scan(
        uniform_fp lp, const float * lp_vec, uniform_fp prefix, int K,
        const int (&g)[kVsR],
        const float (&l_own)[kVsR], const float (&u)[kVsR],
        const bool (&active)[kVsR], int (&found)[kVsR]
#ifdef DIST_VS_STAMPS
        , int & chunks_done
#endif
        ) {
    int gchunk[kVsR], gpiece[kVsR];
#pragma unroll
    for (int r = 0; r < kVsR; ++r) {
        gchunk[r] = active[r] ? (g[r] / kVsUnroll) : -1;
        gpiece[r] = active[r] ? (g[r] >> 3) : -1;
    }
    // a lane's two rows advance together: .x is tile row 2*lane, .y the next
    // one (neighbours in the group-sorted tile, so they share own-slot pieces)
    // no own slot before the tile's first own chunk: start from the value's
    // running sum at that boundary (k_vs_prepare)
    int c_first = 0;
    float start = 0.f;
    if (prefix) {
        const int nchunks = (K + kVsUnroll - 1) / kVsUnroll;
        int m = min(active[0] ? gchunk[0] : nchunks,
                    active[1] ? gchunk[1] : nchunks);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = min(m, __shfl_xor(m, off));
        c_first = __builtin_amdgcn_readfirstlane(m);
        if (c_first >= nchunks) c_first = 0;   // (no active lane)
        start = prefix[c_first];
    }
    v2f total = {start, start};
    for (int c = c_first, k0 = c_first * kVsUnroll; k0 < K;
         ++c, k0 += kVsUnroll) {
        float l[kVsUnroll];
        vs_fetch_chunk(lp, k0, l);
#ifdef DIST_VS_STAMPS
        ++chunks_done;
#endif
        if (__any(gchunk[0] == c || gchunk[1] == c)) {
            // own slots of a group-sorted tile are neighbours: only the
            // eight-entry pieces that hold one take the per-lane select
#pragma unroll
            for (int b = 0; b < kVsUnroll / 8; ++b) {
                const int piece = (k0 >> 3) + b;
                if (__any(gpiece[0] == piece || gpiece[1] == piece)) {
                    const float l8[8] = {l[8 * b], l[8 * b + 1], l[8 * b + 2],
                                         l[8 * b + 3], l[8 * b + 4],
                                         l[8 * b + 5], l[8 * b + 6],
                                         l[8 * b + 7]};
                    vs_own_piece<false>(total, l8, k0 + 8 * b, g, l_own);
                } else {
#pragma unroll
                    for (int j = 8 * b; j < 8 * b + 8; ++j)
                        total += vs_splat(l[j]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kVsUnroll; ++j) total += vs_splat(l[j]);
        }
    }
    // t never increases, so the number of chunks that END with t > 0 is the
    // chunk in which the lane crosses zero, and the last such end value is
    // the value it enters that chunk with: three operations per chunk
    v2f t = total * (v2f){u[0], u[1]};
    float t_start[kVsR] = {t.x, t.y};
    int npos[kVsR] = {0, 0};
    const int nchunks = (K + kVsUnroll - 1) / kVsUnroll;
    for (int c = 0, k0 = 0; k0 < K; ++c, k0 += kVsUnroll) {
        float l[kVsUnroll];
        vs_fetch_chunk(lp, k0, l);
#ifdef DIST_VS_STAMPS
        ++chunks_done;
#endif
        if (__any(gchunk[0] == c || gchunk[1] == c)) {
#pragma unroll
            for (int b = 0; b < kVsUnroll / 8; ++b) {
                const int piece = (k0 >> 3) + b;
                if (__any(gpiece[0] == piece || gpiece[1] == piece)) {
                    const float l8[8] = {l[8 * b], l[8 * b + 1], l[8 * b + 2],
                                         l[8 * b + 3], l[8 * b + 4],
                                         l[8 * b + 5], l[8 * b + 6],
                                         l[8 * b + 7]};
                    vs_own_piece<true>(t, l8, k0 + 8 * b, g, l_own);
                } else {
#pragma unroll
                    for (int j = 8 * b; j < 8 * b + 8; ++j)
                        t -= vs_splat(l[j]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kVsUnroll; ++j) t -= vs_splat(l[j]);
        }
        const float tr[kVsR] = {t.x, t.y};
        bool more = false;
#pragma unroll
        for (int r = 0; r < kVsR; ++r) {
            const bool pos = tr[r] > 0.f;
            t_start[r] = pos ? tr[r] : t_start[r];
            npos[r] += pos ? 1 : 0;
            more = more || (active[r] && pos);
        }
        // (the ballot of the predicate itself: __any() goes through an int)
        if (__builtin_amdgcn_ballot_w64(more) == 0) break;
    }
#pragma unroll
    for (int r = 0; r < kVsR; ++r) {
        int f = K - 1;
        if (active[r] && npos[r] < nchunks) {
            // replay the crossing chunk: a first t <= 0 at its entry j is
            // index k0 + j (random.hpp:326-329)
            // (the chunk is 128 contiguous, aligned bytes of the padded
            // vector: eight 16-byte loads, then selects -- no branches)
            const float4 * chunk = reinterpret_cast<const float4 *>(
                lp_vec + npos[r] * kVsUnroll);
            float4 v[kVsUnroll / 4];
#pragma unroll
            for (int q = 0; q < kVsUnroll / 4; ++q) v[q] = chunk[q];
            const int own = g[r] - npos[r] * kVsUnroll;   // in 0..31 or not
            float tt = t_start[r];
            int steps = 0;
#pragma unroll
            for (int q = 0; q < kVsUnroll / 4; ++q) {
                const float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    tt -= (own == 4 * q + i) ? l_own[r] : e[i];
                    steps += (tt > 0.f) ? 1 : 0;
                }
            }
            f = npos[r] * kVsUnroll + steps;
        }
        found[r] = f < K - 1 ? f : K - 1;
    }
}

// BLOCK = kVsSampleBlock for launches that fill the chip (8 tiles per
// workgroup, mostly of one value: they share their scalar-cache lines); 64
// for small ones -- a 65 536-row batch is 512 tiles, which 1024-thread
// workgroups would pile onto 32 of the 256 CUs, four waves to a SIMD.
// (512: eight tiles per workgroup.  Measured round 4 against 1024, one box:
// DD-256 9.18 against 9.05 G row-updates/s, Zipf values 7.62 / 7.37, GP 2.71 /
// 2.67, K = 512 12.2 / 11.8, 786 k rows per launch 7.21 / 7.08; BB 9.09 /
// 9.22 and DD-16 8.36 / 8.43 -- few values, whose tiles share more of the
// scalar cache in the larger workgroup.  128: DD-16 8.18, BB 8.65.)
constexpr int kVsSampleBlock = 512;
template <int KIND, int BLOCK>
__global__ __launch_bounds__(BLOCK)
__attribute__((amdgpu_waves_per_eu(8, 8)))
void k_vs_sample(
        SweepParams P, VsTables T, const VsTile * __restrict__ tiles,
        uint32_t n_tiles, uint32_t n_band_ids,
        const uint32_t * __restrict__ sorted_rows, VsDefer D) {
    const int lane = threadIdx.x & 63;
    const uint32_t id = __builtin_amdgcn_readfirstlane(
        blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
    // the first n_band_ids ids (a whole number of workgroups, resident from
    // the launch's first cycle) are the values' band tiles (VsTables); a band
    // tile samples the arg-max group's rows only, a regular tile of a value
    // with a band tile everything else
#ifdef DIST_VS_STAMPS   // diagnostic build only (make stamps): costs 3 us
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;
    int chunks_done = 0;   // chunks of both recurrences, both vectors
    if (T.stamps) st0 = __builtin_amdgcn_s_memtime();
#endif
    const bool band = id < n_band_ids;
    const VsTile * mine = band ? T.band_tile + id : tiles + (id - n_band_ids);
    if (band ? id >= T.band_count : id - n_band_ids >= n_tiles) return;
    const uint32_t x = __builtin_amdgcn_readfirstlane(mine->x);
    const uint32_t pos = __builtin_amdgcn_readfirstlane(mine->pos);
    const uint32_t n = __builtin_amdgcn_readfirstlane(mine->n);
    if (n == 0) return;
    const bool skip_a = band;
    const bool skip_b =
        !band && n_band_ids != 0
        && T.band_mode[T.band_by_chunk ? mine->chunk : x] != 0;
    SlaveView v = P.feat[0];
    v.kind = KIND;
    const int K = sweep_K(P);
    const float shift = P.scalars->shift;
    const float M = T.M[x], mB = T.mB[x];
    const int amax = T.argmax[x];
    const float ea = u2f(g_tables_dev.exp_ab[0]);
    const float eb = u2f(g_tables_dev.exp_ab[1]);
    const float lf = KIND == DIST_GP ? fast_log_factorial(x) : 0.f;

    bool valid[kVsR], inA[kVsR], inB[kVsR];
    size_t row[kVsR];
    int g[kVsR], g2[kVsR];
    float l_own[kVsR], u[kVsR];
    bool anyA = false, anyB = false;
#pragma unroll
    for (int r = 0; r < kVsR; ++r) {
        valid[r] = (uint32_t)(kVsR * lane + r) < n;
        row[r] = 0;
        g[r] = -1;
        g2[r] = 0;
        l_own[r] = 0.f;
        u[r] = 0.f;
        bool classB = false;
        if (valid[r]) {
            const uint32_t at = pos + kVsR * lane + r;
            row[r] = P.row_begin + sorted_rows[at];
            g[r] = P.g2p[P.assign_pos[at]];
            classB = (g[r] == amax);
            if (classB ? skip_b : skip_a) valid[r] = false;
        }
        if (valid[r]) {
            float s_own = 0.f, l_tab = 0.f;
            bool defer;
            if (T.own) {   // (k_vs_tables did this per (value, group))
                l_tab = T.own[(size_t)x * T.Kpad + g[r]];
                defer = l_tab < 0.f;
            } else {
                const int n_g = P.counts[g[r]];
                defer = (n_g == 1);
                if (!defer) {
                    s_own = vs_own_score(P, v, g[r], n_g, x, lf, shift);
                    defer = !classB && s_own > M;   // table rounding lifted it
                }
            }
            const float m = classB ? mB : M;
            if (defer) {
                const uint32_t at = pos + kVsR * lane + r;
                // (a band tile's rows may straddle two chunks of its value)
                const uint32_t chunk =
                    !D.chunk_counts ? 0u
                    : (band && !T.band_by_chunk)
                        ? T.chunk_first[x]
                              + (at - T.val_start[x]) / (uint32_t)kVsApplyRows
                        : mine->chunk;
                vs_hand_over(D, chunk, at);
                valid[r] = false;
            } else {
                l_own[r] = T.own ? l_tab
                                 : fast_exp_nonpos(s_own - m,
                                                   g_tables_dev.exp_table, ea,
                                                   eb);
                u[r] = batch_row_unif01(P, row[r]);
            }
        }
        inA[r] = valid[r] && !classB;
        inB[r] = valid[r] && classB;
        anyA = anyA || inA[r];
        anyB = anyB || inB[r];
    }
    // A tile that holds rows of the value's arg-max group next to others runs
    // both passes.  Left at equal priority it finishes them alone on its SIMD,
    // one dependent add at a time (measured: +30 % on that SIMD's time, and
    // the slowest SIMD is the kernel's time); ahead of its neighbours it ends
    // with them.
    if (__any(anyA) && __any(anyB)) __builtin_amdgcn_s_setprio(3);
#ifdef DIST_VS_STAMPS
    if (T.stamps) st1 = __builtin_amdgcn_s_memtime();
#endif
    if (__any(anyA)) {
        const float * vec = T.LA + (size_t)x * T.Kpad;
        int f[kVsR];
        vs_sum_and_scan(as_uniform(vec), vec,
                        T.PA ? as_uniform(T.PA + (size_t)x
                                          * (T.Kpad / kVsUnroll)) : nullptr,
                        K, g, l_own, u, inA, f
#ifdef DIST_VS_STAMPS
                        , chunks_done
#endif
                        );
#pragma unroll
        for (int r = 0; r < kVsR; ++r) g2[r] = inA[r] ? f[r] : g2[r];
    }
#ifdef DIST_VS_STAMPS
    if (T.stamps) st2 = __builtin_amdgcn_s_memtime();
#endif
    if (__any(anyB)) {
        const float * vec = T.LB + (size_t)x * T.Kpad;
        int f[kVsR];
        vs_sum_and_scan(as_uniform(vec), vec,
                        T.PB ? as_uniform(T.PB + (size_t)x
                                          * (T.Kpad / kVsUnroll)) : nullptr,
                        K, g, l_own, u, inB, f
#ifdef DIST_VS_STAMPS
                        , chunks_done
#endif
                        );
#pragma unroll
        for (int r = 0; r < kVsR; ++r) g2[r] = inB[r] ? f[r] : g2[r];
    }
#ifdef DIST_VS_STAMPS
    if (T.stamps) st3 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int r = 0; r < kVsR; ++r) {
        if (valid[r]) {
            const uint32_t at = pos + kVsR * lane + r;
            P.old_packed[at] = (uint32_t)g[r];
            P.new_packed[at] = (uint32_t)g2[r];
        }
    }
#ifdef DIST_VS_STAMPS
    if (T.stamps && lane == 0) {
        unsigned long long * out = T.stamps + (size_t)id * 6;
        out[0] = st0; out[1] = st1; out[2] = st2; out[3] = st3;
        out[4] = __builtin_amdgcn_s_memtime();
        // HW_ID: wave, SIMD, CU, SH, SE and (XCC_ID) the XCD
        out[5] = (unsigned long long)__builtin_amdgcn_s_getreg(
                     (4 << 0) | (0 << 6) | (31 << 11))
               | ((unsigned long long)__builtin_amdgcn_s_getreg(
                     (20 << 0) | (0 << 6) | (3 << 11)) << 32)
               | ((unsigned long long)chunks_done << 40);
    }
#endif
}

// ---------------------------------------------------------------------------
// Scan sampling on the value-sorted path (option "sampling" = 1: opt-in,
// tolerance-level; the exact kernels above stay the line of record).
//
// Rows with the same value x share their score vector except in their own
// slot, so the softmax and its cumulative sums are a property of the VALUE:
// k_vs_scan_prepare computes, per value, the scores (the exact kernels' float
// operations: bit-identical scores), their maximum M[x], the likelihoods
// exp2((s - M) log2 e) and their inclusive prefix sums C[x][k] by a parallel
// scan.  A row then needs its own slot's two likelihoods -- with the row
// removed (l_own) and as tabulated (l_g) -- and a binary search for the first
// k with  C[x][k] + (k >= g ? l_own - l_g : 0)  >=  u * (C[x][K-1] + l_own -
// l_g): about log2 K dependent loads per row instead of 2 K dependent adds
// (random.hpp:316-333 in distribution; the same engine step per row).
// What a sub-sweep then costs is the table pass -- V x K entries read and
// written once, the HBM stream SURVEY 8d prices for C5 -- and the launches
// around it.  Rows alone in their group are handed to the wave-per-row kernel
// as on the exact path.
struct VsScanTables {
    float * C;        // [nvals][Kpad] inclusive prefix sums of the likelihoods
    float * coarse;   // [nvals][Kpad / kVsScanCoarse]: C[x][64 j + 63]
    float * M;        // [nvals] maxima
    float * total;    // [nvals] C[x][K - 1]
    int Kpad;         // a multiple of kVsScanCoarse
    uint32_t n_values;
    int lds_scores;
};
constexpr int kVsScanBlock = 256;
constexpr int kVsScanCoarse = 64;

template <int KIND>
__global__ __launch_bounds__(kVsScanBlock) void k_vs_scan_prepare(
        SweepParams P, VsScanTables T, uint32_t * deferred_count,
        uint32_t deferred_initial) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *deferred_count = deferred_initial;
    __shared__ float r_m[kVsScanBlock / 64];
    __shared__ float r_sum[kVsScanBlock / 64];
    __shared__ float sh_M, sh_carry;
    extern __shared__ float s_scores[];   // [Kpad] when T.lds_scores
    constexpr float kLog2e = 1.44269504088896341f;
    const uint32_t x = blockIdx.x;
    SlaveView v = P.feat[0];
    v.kind = KIND;
    const int K = sweep_K(P);
    const float lf = KIND == DIST_GP ? fast_log_factorial(x) : 0.f;
    float * c = T.C + (size_t)x * T.Kpad;
    // the scores wait for pass 2 in LDS (or, too many for it, in the prefix
    // row itself: one more trip of the row through memory)
    float * sc = T.lds_scores ? s_scores : c;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // pass 1: scores and their maximum
    float m = -INFINITY;
    {   // (four groups per thread in flight: the pass is a stream of loads)
        int k = threadIdx.x;
        for (; k + 3 * kVsScanBlock < K; k += 4 * kVsScanBlock) {
            Entry e[4];
            float b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                e[i] = load_entry(v, k + i * kVsScanBlock, x);
                b[i] = P.base[k + i * kVsScanBlock];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s = accumulate(KIND, b[i], e[i], x, lf, v.p);
                sc[k + i * kVsScanBlock] = s;
                m = fmaxf(m, s);
            }
        }
        for (; k < K; k += kVsScanBlock) {
            const float s =
                accumulate(KIND, P.base[k], load_entry(v, k, x), x, lf, v.p);
            sc[k] = s;
            m = fmaxf(m, s);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) r_m[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mm = r_m[0];
        for (int w = 1; w < kVsScanBlock / 64; ++w) mm = fmaxf(mm, r_m[w]);
        sh_M = mm;
        sh_carry = 0.f;
        T.M[x] = mm;
    }
    __syncthreads();
    const float mc = -sh_M * kLog2e;
    // pass 2: likelihoods and their inclusive prefix sums, kVsScanBlock x 4
    // entries a round (each thread four consecutive ones), the rounds chained
    // through sh_carry
    for (int k0 = 0; k0 < T.Kpad; k0 += 4 * kVsScanBlock) {
        const int k = k0 + 4 * threadIdx.x;
        float l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            l[i] = k + i < K ? __builtin_amdgcn_exp2f(
                                   __builtin_fmaf(sc[k + i], kLog2e, mc))
                             : 0.f;
        l[1] += l[0]; l[2] += l[1]; l[3] += l[2];
        // inclusive scan of the threads' sums over the wave, then the waves
        float run = l[3];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float up = __shfl_up(run, off);
            if (lane >= off) run += up;
        }
        if (lane == 63) r_sum[wave] = run;
        __syncthreads();
        float before = sh_carry;
        for (int w = 0; w < wave; ++w) before += r_sum[w];
        before += run - l[3];   // the wave's threads before this one
        if (k < T.Kpad) {   // (Kpad is a multiple of four: whole float4s)
            *reinterpret_cast<float4 *>(c + k) = make_float4(
                before + l[0], before + l[1], before + l[2], before + l[3]);
            // every kVsScanCoarse-th prefix again, close together: the rows'
            // search starts there
            if ((k & (kVsScanCoarse - 1)) == kVsScanCoarse - 4)
                T.coarse[(size_t)x * (T.Kpad / kVsScanCoarse)
                         + k / kVsScanCoarse] = before + l[3];
        }
        __syncthreads();
        if (threadIdx.x == kVsScanBlock - 1) sh_carry = before + l[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) T.total[x] = sh_carry;
}

// one thread per position of the value-sorted row list
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_vs_scan_rows(
        SweepParams P, VsScanTables T,
        const uint32_t * __restrict__ sorted_rows, size_t n,
        uint32_t * __restrict__ deferred, uint32_t * deferred_count) {
    const size_t at = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (at >= n) return;
    constexpr float kLog2e = 1.44269504088896341f;
    const size_t row = P.row_begin + sorted_rows[at];
    const uint32_t x = P.values[0][row];
    if (x >= T.n_values) return;   // (beyond the table: listed by the host)
    SlaveView v = P.feat[0];
    v.kind = KIND;
    const int K = sweep_K(P);
    const int g = P.g2p[P.assign_pos[at]];
    const int n_g = P.counts[g];
    if (n_g == 1) {   // the group would vanish: the wave-per-row kernel
        deferred[atomicAdd(deferred_count, 1u)] = (uint32_t)at;
        return;
    }
    const float lf = KIND == DIST_GP ? fast_log_factorial(x) : 0.f;
    const float s_own = vs_own_score(P, v, g, n_g, x, lf, P.scalars->shift);
    const float s_g =
        accumulate(KIND, P.base[g], load_entry(v, g, x), x, lf, v.p);
    const float mc = -T.M[x] * kLog2e;
    const float delta =
        __builtin_amdgcn_exp2f(__builtin_fmaf(s_own, kLog2e, mc))
        - __builtin_amdgcn_exp2f(__builtin_fmaf(s_g, kLog2e, mc));
    const float target = (T.total[x] + delta) * batch_row_unif01(P, row);
    // first among the block ends C[x][64 j + 63] (a few cache lines per
    // value, shared by its rows), then inside the block found
    const float * coarse = T.coarse + (size_t)x * (T.Kpad / kVsScanCoarse);
    int lo = 0, hi = (K - 1) / kVsScanCoarse;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int k = mid * kVsScanCoarse + kVsScanCoarse - 1;
        const float cum = coarse[mid] + (k >= g ? delta : 0.f);
        if (cum >= target) hi = mid; else lo = mid + 1;
    }
    const float * c = T.C + (size_t)x * T.Kpad;
    hi = min(K - 1, lo * kVsScanCoarse + kVsScanCoarse - 1);
    lo = lo * kVsScanCoarse;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const float cum = c[mid] + (mid >= g ? delta : 0.f);
        if (cum >= target) hi = mid; else lo = mid + 1;
    }
    P.old_packed[at] = (uint32_t)g;
    P.new_packed[at] = (uint32_t)lo;
}

// ---------------------------------------------------------------------------
// k_vs_narrow: the value-sorted row update for launches that cannot fill the
// chip (sub-sweeps of some 10^4..10^5 rows).  There a tile's wave is alone on
// its SIMD, and what the launch takes is ONE wave's latency: every scalar load
// of the likelihood vector a round trip to L2 that nothing hides, a dependent
// packed add every 12.5 cycles, two full passes wherever rows of the value's
// arg-max group sit next to others.  So: tiles of 64 rows, one per lane (twice
// the waves; a plain dependent v_add_f32 comes back after 8.5 cycles); the
// tile's vector(s) copied into LDS once, by coalesced loads issued before the
// rows' gathers, and read from there a chunk AHEAD of its use into registers
// (a wave alone has hundreds); rows of the arg-max group read the second
// vector, through the lane's own base address, in the same pass.  The float
// operations per row and their order are k_vs_sample's: bit-identical.
constexpr int kVsNarrowMaxK = 4096;   // two vectors of Kpad + 64 floats in LDS
// (Handing a tile's few rows of the arg-max group to the wave-per-row kernel
// instead -- the tile then keeps to one vector -- was measured: this kernel
// 32 -> 27 us at 65 536 rows, the sub-sweep as a whole 10 % slower: a row
// costs the wave-per-row kernel what a tile costs here.)

// 4 * HQ entries (a chunk, or half of one) of the recurrences, one row per lane
template <bool SCAN, int HQ>
__device__ __forceinline__ void vs_narrow_part(
        float & acc, const float4 (&a)[HQ], bool own, int k0, int g,
        float l_own) {
    if (own) {   // (wave-uniform) a lane's own slot falls into this chunk
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const float ea[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e = (k0 + 4 * q + i == g) ? l_own : ea[i];
                acc = SCAN ? acc - e : acc + e;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const float ea[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = SCAN ? acc - ea[i] : acc + ea[i];
        }
    }
}

// the two recurrences of vs_sum_and_scan for one row per lane; va / vb: the
// vectors in LDS, slack behind each.  A lane holds ONE row, so it reads the
// vector of its row's class through its own base address (the rows of the
// arg-max group vb, the others va): one LDS read serves both classes and no
// entry is selected per lane -- a tile that holds both costs what any tile
// costs (it used to read both vectors and select: 45 k against 32 k cycles,
// and such tiles ended the launch).  The vector is read one part ahead of the
// part in use: a whole chunk (HQ = 8: 32 registers per buffer -- a wave alone
// needs that distance to hide the read) or half of one (HQ = 4).
template <int HQ>
__device__ __forceinline__ int vs_narrow_row(
        const float * va, const float * vb, bool is_b, int K, int g,
        float l_own, float u, bool active) {
    constexpr int parts = kVsUnroll / 4 / HQ;   // per chunk: 1 or 2
    const int nchunks = (K + kVsUnroll - 1) / kVsUnroll;
    const int nsteps = nchunks * parts;
    const int gchunk = active ? g / kVsUnroll : -1;
    const float * mine = is_b ? vb : va;
    const float4 * m4 = reinterpret_cast<const float4 *>(mine);
    float4 a0[HQ], a1[HQ];
    auto fetch = [&](int s, float4 (&a)[HQ]) {
#pragma unroll
        for (int q = 0; q < HQ; ++q) a[q] = m4[s * HQ + q];
    };
    // total = ((l_0 + l_1) + l_2) + ...                  random.cc:100-103
    float acc = 0.f;
    fetch(0, a0);
    for (int s = 0; s < nsteps; s += 2) {
        fetch(s + 1, a1);
        vs_narrow_part<false, HQ>(acc, a0, __any(gchunk == s / parts),
                                  s * 4 * HQ, g, l_own);
        fetch(s + 2, a0);
        if (s + 1 < nsteps)
            vs_narrow_part<false, HQ>(acc, a1,
                                      __any(gchunk == (s + 1) / parts),
                                      (s + 1) * 4 * HQ, g, l_own);
    }
    // t = total*u; t -= l_k until t <= 0                 random.hpp:323-330
    float t = acc * u;
    float t_start = t;
    int npos = 0;
    auto book = [&]() {   // at the end of a chunk
        const bool pos = t > 0.f;
        t_start = pos ? t : t_start;
        npos += pos ? 1 : 0;
        return __builtin_amdgcn_ballot_w64(active && pos) != 0;
    };
    fetch(0, a0);
    for (int s = 0; s < nsteps; s += 2) {
        fetch(s + 1, a1);
        vs_narrow_part<true, HQ>(t, a0, __any(gchunk == s / parts),
                                 s * 4 * HQ, g, l_own);
        if (parts == 1 && !book()) break;
        fetch(s + 2, a0);
        if (s + 1 < nsteps) {
            vs_narrow_part<true, HQ>(t, a1, __any(gchunk == (s + 1) / parts),
                                     (s + 1) * 4 * HQ, g, l_own);
            if (!book()) break;
        }
    }
    int f = K - 1;
    if (active && npos < nchunks) {
        // replay the crossing chunk (as vs_sum_and_scan does)
        const float4 * chunk =
            reinterpret_cast<const float4 *>(mine + npos * kVsUnroll);
        const int own = g - npos * kVsUnroll;   // in 0..31 or not
        float tt = t_start;
        int steps = 0;
#pragma unroll
        for (int q = 0; q < kVsUnroll / 4; ++q) {
            const float4 v = chunk[q];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                tt -= (own == 4 * q + i) ? l_own : e[i];
                steps += (tt > 0.f) ? 1 : 0;
            }
        }
        f = npos * kVsUnroll + steps;
    }
    return f < K - 1 ? f : K - 1;
}

// HQ: float4s read ahead per vector (vs_narrow_row)
template <int KIND, int HQ>
__global__ __launch_bounds__(64) void k_vs_narrow(
        SweepParams P, VsTables T, const VsTile * __restrict__ tiles,
        uint32_t n_tiles, const uint32_t * __restrict__ sorted_rows,
        VsDefer D) {
    extern __shared__ float4 s_narrow[];   // [2][(Kpad + 2 * kVsUnroll) / 4]
    const int lane = threadIdx.x;
    const uint32_t id = blockIdx.x;
    if (id >= n_tiles) return;
    const uint32_t x = __builtin_amdgcn_readfirstlane(tiles[id].x);
    const uint32_t pos = __builtin_amdgcn_readfirstlane(tiles[id].pos);
    const uint32_t n = __builtin_amdgcn_readfirstlane(tiles[id].n);
    if (n == 0) return;
#ifdef DIST_VS_STAMPS   // diagnostic build only (make stamps)
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;
    if (T.stamps) st0 = __builtin_amdgcn_s_memtime();
#endif
    const int Kpad = T.Kpad;
    const int quads = T.Kuse / 4;
    const int stride = quads + 2 * kVsUnroll / 4;   // float4s per vector
    // the vector of the rows outside the arg-max group: on its way before
    // the rows' own gathers start
    constexpr int kQ = kVsNarrowMaxK / 4 / 64;
    const float4 * ga =
        reinterpret_cast<const float4 *>(T.LA + (size_t)x * Kpad);
    float4 stage[kQ];
#pragma unroll
    for (int q = 0; q < kQ; ++q)
        if (64 * q < quads)   // (uniform; lanes past the end re-read its last)
            stage[q] = ga[min(lane + 64 * q, quads - 1)];
        else
            stage[q] = float4{0.f, 0.f, 0.f, 0.f};

    SlaveView v = P.feat[0];
    v.kind = KIND;
    const int K = sweep_K(P);
    const float shift = P.scalars->shift;
    const float M = T.M[x], mB = T.mB[x];
    const int amax = T.argmax[x];
    const float ea = u2f(g_tables_dev.exp_ab[0]);
    const float eb = u2f(g_tables_dev.exp_ab[1]);
    const float lf = KIND == DIST_GP ? fast_log_factorial(x) : 0.f;

    bool valid = (uint32_t)lane < n;
    const uint32_t at = pos + lane;
    int g = -1;
    float l_own = 0.f, u = 0.f;
    bool is_b = false;
    if (valid) {
        const size_t row = P.row_begin + sorted_rows[at];
        g = P.g2p[P.assign_pos[at]];
        is_b = (g == amax);
        const float m = is_b ? mB : M;
        float s_own = 0.f, l_tab = 0.f;
        bool defer;
        if (T.own) {   // (k_vs_tables did this per (value, group))
            l_tab = T.own[(size_t)x * Kpad + g];
            defer = l_tab < 0.f;
        } else {
            const int n_g = P.counts[g];
            defer = (n_g == 1);
            if (!defer) {
                s_own = vs_own_score(P, v, g, n_g, x, lf, shift);
                defer = !is_b && s_own > M;   // table rounding lifted it
            }
        }
        if (defer) {
            vs_hand_over(D, tiles[id].chunk, at);
            valid = false;
        } else {
            l_own = T.own ? l_tab
                          : fast_exp_nonpos(s_own - m, g_tables_dev.exp_table,
                                            ea, eb);
            u = batch_row_unif01(P, row);
        }
    }
    const bool any_a = __any(valid && !is_b), any_b = __any(valid && is_b);
    if (!any_a && !any_b) return;
#ifdef DIST_VS_STAMPS
    if (T.stamps) st1 = __builtin_amdgcn_s_memtime();
#endif
    float4 * sa = s_narrow;
    float4 * sb = s_narrow + stride;
#pragma unroll
    for (int q = 0; q < kQ; ++q)
        if (lane + 64 * q < quads) sa[lane + 64 * q] = stage[q];
    if (any_b) {
        const float4 * gb =
            reinterpret_cast<const float4 *>(T.LB + (size_t)x * Kpad);
#pragma unroll
        for (int q = 0; q < kQ; ++q)
            if (64 * q < quads) stage[q] = gb[min(lane + 64 * q, quads - 1)];
#pragma unroll
        for (int q = 0; q < kQ; ++q)
            if (lane + 64 * q < quads) sb[lane + 64 * q] = stage[q];
    }
    // (the slack behind each vector is read ahead, never used)
    if (lane < 2 * kVsUnroll / 4) {
        sa[quads + lane] = float4{0.f, 0.f, 0.f, 0.f};
        sb[quads + lane] = float4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef DIST_VS_STAMPS
    if (T.stamps) st2 = __builtin_amdgcn_s_memtime();
#endif
    const float * fa = reinterpret_cast<const float *>(sa);
    const float * fb = reinterpret_cast<const float *>(sb);
    // (lanes without a row read va: is_b is false there)
    const int g2 = vs_narrow_row<HQ>(fa, fb, is_b && valid, K, g, l_own, u,
                                     valid);
#ifdef DIST_VS_STAMPS
    if (T.stamps) st3 = __builtin_amdgcn_s_memtime();
#endif
    if (valid) {
        P.old_packed[at] = (uint32_t)g;
        P.new_packed[at] = (uint32_t)g2;
    }
#ifdef DIST_VS_STAMPS
    // phases: rows' set-up | vectors into LDS | the recurrences | write back
    if (T.stamps && lane == 0) {
        unsigned long long * out = T.stamps + (size_t)id * 6;
        out[0] = st0; out[1] = st1; out[2] = st2; out[3] = st3;
        out[4] = __builtin_amdgcn_s_memtime();
        out[5] = (unsigned long long)__builtin_amdgcn_s_getreg(
                     (4 << 0) | (0 << 6) | (31 << 11))
               | ((unsigned long long)__builtin_amdgcn_s_getreg(
                     (20 << 0) | (0 << 6) | (3 << 11)) << 32);
    }
#endif
}

// ---------------------------------------------------------------------------
// The value-sorted row update WITHOUT per-value tables (k_vs_stream).
//
// The tables of k_vs_prepare pay when many tiles share a value's likelihood
// vector.  Where a value has a tile or two per batch (C5: V = 10 000 values,
// K = 8192 groups, 100 rows per value and sub-sweep -- the 2 x 328 MB of LA /
// LB would be written to HBM and read back exactly once each) one wave per
// tile builds the vector itself: scores and (max, arg-max, second max) in a
// first pass over the value's cache row S[x][.] (the only HBM stream), then
// per pass of the two recurrences the exponentials of kVsStreamChunk entries
// at a time into the wave's strip of LDS, consumed from there by uniform
// ds_read_b128 exactly as vs_sum_and_scan consumes its scalar loads (same
// float operations in the same order: bit-identical to k_vs_sample).  Rows of
// the arg-max group (shift mB instead of M) sit in lanes of their own and read
// a second strip in the same loop; rows the shortcut does not cover are handed
// over as before.
constexpr int kVsStreamChunk = 256;
constexpr int kVsStreamBlock = 256;

__global__ void k_set_u32(uint32_t * p, uint32_t value) { *p = value; }

// acc (+/-)= splat(w.x), then w.y, w.z, w.w: four dependent v_pk_add_f32 whose
// second operand is ONE dword of a register pair taken into both halves by
// op_sel (the compiler moves the odd dwords into place first, a VALU move per
// entry).  A packed add that consumes the previous one's result needs one
// wait state (the compiler puts s_nop 0 / a scalar move there itself); inside
// an asm block nobody does, so they are spelled out, also ahead of the first
// add and after the last.  x - y == x + (-y) exactly (neg_lo / neg_hi).
template <bool SUB>
__device__ __forceinline__ void vs_pk_chain4(v2f & acc, const float4 & w) {
    const v2f lo = {w.x, w.y}, hi = {w.z, w.w};
    if (SUB) {
        asm("s_nop 0\n\t"
            "v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "s_nop 0\n\t"
            "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "s_nop 0\n\t"
            "v_pk_add_f32 %0, %0, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "s_nop 0\n\t"
            "v_pk_add_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "s_nop 0"
            : "+v"(acc) : "v"(lo), "v"(hi));
    } else {
        asm("s_nop 0\n\t"
            "v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]\n\t"
            "s_nop 0\n\t"
            "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
            "s_nop 0\n\t"
            "v_pk_add_f32 %0, %0, %2 op_sel_hi:[1,0]\n\t"
            "s_nop 0\n\t"
            "v_pk_add_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
            "s_nop 0"
            : "+v"(acc) : "v"(lo), "v"(hi));
    }
}

// four entries (k0 .. k0+3) into which own slots fall: vs_own_piece's select
template <bool SUB>
__device__ __forceinline__ void vs_own_quad(
        v2f & acc, const float4 & w, int k0, const int (&g)[kVsR],
        const float (&l_own)[kVsR]) {
    const float l[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const v2f e = {(k0 + j == g[0]) ? l_own[0] : l[j],
                       (k0 + j == g[1]) ? l_own[1] : l[j]};
        acc = SUB ? acc - e : acc + e;
    }
}

template <int KIND>
__device__ __forceinline__ float vs_stream_score(const SweepParams & P,
                                                 const SlaveView & v, int k,
                                                 uint32_t x, float lf) {
    return accumulate(KIND, P.base[k], load_entry(v, k, x), x, lf, v.p);
}

// (five waves to a SIMD: 96 registers hold the prefetched inputs without
// spilling; measured 1.00 ms per C5 launch against 1.10 at six, 1.12 at four
// and 1.97 at eight -- profiles/r4_experiments.txt)
template <int KIND>
__global__ __launch_bounds__(kVsStreamBlock)
__attribute__((amdgpu_waves_per_eu(5, 5)))
void k_vs_stream(
        SweepParams P, const VsTile * __restrict__ tiles, uint32_t n_tiles,
        const uint32_t * __restrict__ sorted_rows,
        uint32_t * __restrict__ deferred, uint32_t * deferred_count,
        float * __restrict__ scratch, uint32_t scratch_stride) {
    __shared__ uint32_t s_exp[1024];
    __shared__ float s_strip[kVsStreamBlock / 64][2][kVsStreamChunk];
    for (int i = threadIdx.x; i < 1024; i += kVsStreamBlock)
        s_exp[i] = g_tables_dev.exp_table[i];
    __syncthreads();
    const float ea = u2f(g_tables_dev.exp_ab[0]);
    const float eb = u2f(g_tables_dev.exp_ab[1]);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float * stripA = s_strip[wave][0];
    float * stripB = s_strip[wave][1];
    const uint32_t id = __builtin_amdgcn_readfirstlane(
        blockIdx.x * (kVsStreamBlock / 64) + wave);
    if (id >= n_tiles) return;
    const uint32_t x = __builtin_amdgcn_readfirstlane(tiles[id].x);
    const uint32_t pos = __builtin_amdgcn_readfirstlane(tiles[id].pos);
    const uint32_t n = __builtin_amdgcn_readfirstlane(tiles[id].n);
    if (n == 0) return;
    const SlaveView & v = P.feat[0];   // (read in place: the argument block)
    const int K = sweep_K(P);
    const float shift = P.scalars->shift;
    const float lf = KIND == DIST_GP ? fast_log_factorial(x) : 0.f;
    // the tile's row of the scratch (null: none): the total's pass leaves
    // its likelihoods there, the scan and the replay read them back instead
    // of evaluating score and exponential a second and a third time
    float * keep =
        scratch ? scratch + (size_t)id * 2 * scratch_stride : nullptr;

    // A wave on its own is a chain of dependent steps, and seven neighbours do
    // not hide a memory round trip per step: every loop below has the inputs
    // of its NEXT step in flight while it works on this one.
    constexpr int J = kVsStreamChunk / 64;   // entries per lane and chunk
    struct Raw { float base[J]; Entry e[J]; };
    // (load_entry, spelled out on plain pointers: the value is the wave's,
    // so the row of S -- or OTHER's scalar, dpd.hpp:534-542 -- is chosen once)
    const float * const par = P.feat[0].p;   // (the launch's argument block)
    const float * const base_p = P.base;
    const float * const c0_p = v.c0;
    const float * const c1_p = v.c1;
    const float * const c2_p = v.c2;
    const float * const c3_p = v.c3;
    const bool is_other = KIND == DIST_DPD && x == DIST_DPD_OTHER;
    const float other_score = v.other;
    const float * const s_row =
        (!is_cat(KIND) || is_other) ? v.c0 : v.S + (size_t)x * v.cap;
    auto entry_at = [&](int k) {
        Entry e;
        e.c0 = c0_p[k];
        if (is_cat(KIND)) {
            const float t = s_row[k];
            e.c1 = is_other ? other_score : t;
            e.c2 = 0.f;
            e.c3 = 0.f;
        } else {
            e.c1 = c1_p[k];
            e.c2 = c2_p[k];
            e.c3 = c3_p[k];
        }
        return e;
    };
    auto fetch = [&](Raw & r, int k0) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int k = min(k0 + lane + 64 * j, K - 1);
            r.base[j] = base_p[k];
            r.e[j] = entry_at(k);
        }
    };
    // pass 0: (max, first arg-max, max of the rest) of the value's scores
    float m1 = -INFINITY, m2 = -INFINITY;
    int i1 = 0x7fffffff;
    {
        Raw even, odd;
        auto fold = [&](const Raw & r, int k0) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int k = k0 + lane + 64 * j;
                const float sc = accumulate(KIND, r.base[j], r.e[j], x, lf, par);
                if (k < K) {
                    if (sc > m1) { m2 = m1; m1 = sc; i1 = k; }
                    else if (sc > m2) m2 = sc;
                }
            }
        };
        fetch(even, 0);
        for (int k0 = 0; k0 < K; k0 += 2 * kVsStreamChunk) {
            fetch(odd, k0 + kVsStreamChunk);
            __builtin_amdgcn_sched_barrier(0);
            fold(even, k0);
            __builtin_amdgcn_sched_barrier(0);
            fetch(even, k0 + 2 * kVsStreamChunk);
            __builtin_amdgcn_sched_barrier(0);
            fold(odd, k0 + kVsStreamChunk);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float b1 = __shfl_xor(m1, off), b2 = __shfl_xor(m2, off);
        const int bi = __shfl_xor(i1, off);
        if (m1 > b1 || (m1 == b1 && i1 < bi)) {
            m2 = fmaxf(m2, b1);
        } else {
            m2 = fmaxf(b2, m1);
            m1 = b1;
            i1 = bi;
        }
    }
    const float M = m1, M2 = m2;
    const int amax = i1;

    // Rows of the value's arg-max group (class B) want the other shift and
    // have ONE vector between them (own slot included: the same group).  The
    // tile's rows are dealt to the lanes anew -- the others in tile order
    // from lane 0, the arg-max group's from the next free lane -- so that a
    // lane holds rows of one class, and each lane reads its operands from its
    // class's strip: both classes run the recurrences in the same loop.
    uint32_t * order = reinterpret_cast<uint32_t *>(stripA);   // [128], before the loop
    int slot_of[kVsR];
    {
        bool nat_valid[kVsR], nat_b[kVsR];
#pragma unroll
        for (int r = 0; r < kVsR; ++r) {
            order[kVsR * lane + r] = 0xFFu;
            nat_valid[r] = (uint32_t)(kVsR * lane + r) < n;
            nat_b[r] = false;
            if (nat_valid[r]) {
                const uint32_t at = pos + kVsR * lane + r;
                nat_b[r] = (int)P.g2p[P.assign_pos[at]] == amax;
            }
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned long long a0 =
            __builtin_amdgcn_ballot_w64(nat_valid[0] && !nat_b[0]);
        const unsigned long long a1 =
            __builtin_amdgcn_ballot_w64(nat_valid[1] && !nat_b[1]);
        const unsigned long long b0 =
            __builtin_amdgcn_ballot_w64(nat_valid[0] && nat_b[0]);
        const unsigned long long b1 =
            __builtin_amdgcn_ballot_w64(nat_valid[1] && nat_b[1]);
        const int n_a = __builtin_popcountll(a0) + __builtin_popcountll(a1);
        const int b_first = (n_a + kVsR - 1) / kVsR * kVsR;
        const int rank_a = __builtin_popcountll(a0 & below)
                           + __builtin_popcountll(a1 & below);
        const int rank_b = __builtin_popcountll(b0 & below)
                           + __builtin_popcountll(b1 & below);
        slot_of[0] = nat_b[0] ? b_first + rank_b : rank_a;
        slot_of[1] = nat_b[1] ? b_first + rank_b + (nat_b[0] ? 1 : 0)
                              : rank_a + ((nat_valid[0] && !nat_b[0]) ? 1 : 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < kVsR; ++r) {
            if (!nat_valid[r]) continue;
            if (slot_of[r] < kVsR * 64)
                order[slot_of[r]] = kVsR * lane + r;
            else   // (a full tile whose split costs a slot: one row goes on)
                deferred[atomicAdd(deferred_count, 1u)] =
                    pos + kVsR * lane + r;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        slot_of[0] = b_first;   // (kept: where class B begins)
    }
    const bool lane_b = kVsR * lane >= slot_of[0];
    bool valid[kVsR];
    uint32_t at_of[kVsR];
    size_t row[kVsR];
    int g[kVsR], g2[kVsR];
    float l_own[kVsR], u[kVsR];
    float s_own_b = 0.f;
#pragma unroll
    for (int r = 0; r < kVsR; ++r) {
        const uint32_t idx = order[kVsR * lane + r];
        valid[r] = idx != 0xFFu;
        at_of[r] = pos + idx;
        row[r] = 0;
        g[r] = -1;
        g2[r] = 0;
        l_own[r] = 0.f;
        u[r] = 0.f;
        if (valid[r]) {
            const uint32_t at = at_of[r];
            row[r] = P.row_begin + sorted_rows[at];
            g[r] = P.g2p[P.assign_pos[at]];
            const int n_g = P.counts[g[r]];
            float s_own = 0.f;
            bool defer = (n_g == 1);
            if (!defer) {
                s_own = accumulate(KIND, cluster_own_score(P, n_g - 1, shift),
                                   entry_after_remove(v, g[r], x, KIND), x, lf,
                                   par);
                defer = !lane_b && s_own > M;   // table rounding lifted it
            }
            if (defer) {
                deferred[atomicAdd(deferred_count, 1u)] = at;
                valid[r] = false;
            } else {
                if (lane_b) s_own_b = s_own;
                else l_own[r] = fast_exp_nonpos(s_own - M, s_exp, ea, eb);
                u[r] = batch_row_unif01(P, row[r]);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();   // `order` is read: the strip is free
    // class B's shift and own-slot likelihood: one row's, the same for all
    const unsigned long long who_b =
        __builtin_amdgcn_ballot_w64(lane_b && (valid[0] || valid[1]));
    const bool has_b = who_b != 0;
    float mB = M, l_own_b = 0.f;
    if (has_b) {
        const int src = __builtin_ctzll(who_b);
        const float so = u2f((uint32_t)__builtin_amdgcn_readlane(
            (int)f2u(s_own_b), src));
        mB = fmaxf(so, M2);
        l_own_b = fast_exp_nonpos(so - mB, s_exp, ea, eb);
#pragma unroll
        for (int r = 0; r < kVsR; ++r)
            if (lane_b && valid[r]) l_own[r] = l_own_b;
    }
    const float m_mine = lane_b ? mB : M;
    const float * mine = lane_b ? stripB : stripA;
    float * keepB = keep ? keep + scratch_stride : nullptr;
    const float * keep_mine = lane_b ? keepB : keep;
    const int nchunks32 = (K + kVsUnroll - 1) / kVsUnroll;
    // own slots: class B's sits in its strip already
    int gchunk[kVsR], gpiece[kVsR];
#pragma unroll
    for (int r = 0; r < kVsR; ++r) {
        const bool own = valid[r] && !lane_b;
        gchunk[r] = own ? (g[r] / kVsUnroll) : -1;
        gpiece[r] = own ? (g[r] >> 3) : -1;
    }
    v2f acc = {0.f, 0.f};              // the total, then t
    float t_start[kVsR] = {0.f, 0.f};
    int npos[kVsR] = {0, 0};
    const float4 * src = reinterpret_cast<const float4 *>(mine);
    auto run_pass = [&](auto pass_tag) {
        constexpr int pass = decltype(pass_tag)::value;
        if (pass == 1) {
            acc = acc * (v2f){u[0], u[1]};
            t_start[0] = acc.x;
            t_start[1] = acc.y;
        }
        const bool kept = pass == 1 && keep;
        // the chunk's inputs: the cache entries, or what the total's pass kept
        Raw raw;
        float ka[J], kb[J];
        auto fetch_chunk = [&](int k0) {
            if (!kept) return fetch(raw, k0);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int k = min(k0 + lane + 64 * j, (int)scratch_stride - 1);
                ka[j] = keep[k];
                kb[j] = has_b ? keepB[k] : 0.f;
            }
        };
        // sixteen entries of the lane's strip; acc (+/-)= them in order
        auto load16 = [&](float4 (&w)[4], int off) {
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = src[off / 4 + q];
        };
        auto chain16 = [&](const float4 (&w)[4], int kk, bool own_here) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int piece = (kk >> 3) + b;
                if (own_here && __any(gpiece[0] == piece
                                      || gpiece[1] == piece)) {
                    // own slots in these eight entries: the quad that holds
                    // one by per-lane select (vs_sum_and_scan's form), the
                    // other as it is
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const bool mine0 = gpiece[0] == piece
                                           && ((g[0] >> 2) & 1) == h;
                        const bool mine1 = gpiece[1] == piece
                                           && ((g[1] >> 2) & 1) == h;
                        if (__any(mine0 || mine1))
                            vs_own_quad<pass == 1>(acc, w[2 * b + h],
                                                   kk + 8 * b + 4 * h, g,
                                                   l_own);
                        else
                            vs_pk_chain4<pass == 1>(acc, w[2 * b + h]);
                    }
                } else {
                    vs_pk_chain4<pass == 1>(acc, w[2 * b]);
                    vs_pk_chain4<pass == 1>(acc, w[2 * b + 1]);
                }
            }
        };
        fetch_chunk(0);
        bool done = false;
        for (int k0 = 0; k0 < K && !done; k0 += kVsStreamChunk) {
            // the chunk's likelihoods, 64 at a time, into the strips
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int k = k0 + lane + 64 * j;
                float la = 0.f, lb = 0.f;
                if (kept) {
                    la = k < (int)scratch_stride ? ka[j] : 0.f;   // (zeros
                    lb = k < (int)scratch_stride ? kb[j] : 0.f;   // beyond K)
                } else {
                    if (k < K) {
                        const float sc = accumulate(KIND, raw.base[j],
                                                    raw.e[j], x, lf, par);
                        la = fast_exp_nonpos(sc - M, s_exp, ea, eb);
                        if (has_b)
                            lb = k == amax ? l_own_b
                                           : fast_exp_nonpos(sc - mB, s_exp,
                                                             ea, eb);
                    }
                    if (keep && k < (int)scratch_stride) {
                        keep[k] = la;
                        if (has_b) keepB[k] = lb;
                    }
                }
                stripA[lane + 64 * j] = la;
                if (has_b) stripB[lane + 64 * j] = lb;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (k0 + kVsStreamChunk < K) fetch_chunk(k0 + kVsStreamChunk);
            const int sub_end = min(kVsStreamChunk, K - k0);
            float4 w0[4], w1[4];
            load16(w0, 0);
            for (int off = 0; off < sub_end; off += kVsUnroll) {
                const int c = (k0 + off) / kVsUnroll;
                const bool own_here =
                    __any(gchunk[0] == c || gchunk[1] == c);
                load16(w1, off + 16);
                __builtin_amdgcn_sched_barrier(0);
                chain16(w0, k0 + off, own_here);
                __builtin_amdgcn_sched_barrier(0);
                // (the last one stays inside the strip and is not used)
                load16(w0, off + kVsUnroll < kVsStreamChunk ? off + kVsUnroll
                                                            : off);
                __builtin_amdgcn_sched_barrier(0);
                chain16(w1, k0 + off + 16, own_here);
                __builtin_amdgcn_sched_barrier(0);
                if (pass == 1) {
                    const float tr[kVsR] = {acc.x, acc.y};
                    bool more = false;
#pragma unroll
                    for (int r = 0; r < kVsR; ++r) {
                        const bool p = tr[r] > 0.f;
                        t_start[r] = p ? tr[r] : t_start[r];
                        npos[r] += p ? 1 : 0;
                        more = more || (valid[r] && p);
                    }
                    if (__builtin_amdgcn_ballot_w64(more) == 0) {
                        done = true;
                        break;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();   // before the strips are refilled
        }
    };
    run_pass(std::integral_constant<int, 0>{});   // the total
    run_pass(std::integral_constant<int, 1>{});   // the scan
    // replay the chunk in which a row crosses zero (random.hpp:326-329): its
    // likelihoods from what the total's pass kept, or computed once more --
    // the same operations as above
#pragma unroll
    for (int r = 0; r < kVsR; ++r) {
        if (!valid[r]) continue;
        int f = K - 1;
        if (npos[r] < nchunks32) {
            const int base_k = npos[r] * kVsUnroll;
            float tt = t_start[r];
            int steps = 0;
            if (keep) {   // (rows of the scratch are 256-byte aligned)
                const float4 * kept4 =
                    reinterpret_cast<const float4 *>(keep_mine + base_k);
                float4 l4[kVsUnroll / 4];
#pragma unroll
                for (int q = 0; q < kVsUnroll / 4; ++q) l4[q] = kept4[q];
#pragma unroll
                for (int q = 0; q < kVsUnroll / 4; ++q) {
                    const float l[4] = {l4[q].x, l4[q].y, l4[q].z, l4[q].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        tt -= (base_k + 4 * q + i == g[r]) ? l_own[r] : l[i];
                        steps += (tt > 0.f) ? 1 : 0;
                    }
                }
            } else {
                for (int j = 0; j < kVsUnroll; ++j) {
                    const int k = base_k + j;
                    float l = 0.f;
                    if (k == g[r])
                        l = l_own[r];
                    else if (k < K)
                        l = fast_exp_nonpos(
                            accumulate(KIND, base_p[k], entry_at(k), x, lf,
                                       par) - m_mine, s_exp, ea, eb);
                    tt -= l;
                    steps += (tt > 0.f) ? 1 : 0;
                }
            }
            f = base_k + steps;
        }
        g2[r] = f < K - 1 ? f : K - 1;
    }
#pragma unroll
    for (int r = 0; r < kVsR; ++r) {
        if (valid[r]) {
            P.old_packed[at_of[r]] = (uint32_t)g[r];
            P.new_packed[at_of[r]] = (uint32_t)g2[r];
        }
    }
}

// Applying a batch's moves in value-sorted order: one workgroup takes up to
// kVsApplyRows rows of ONE value x and accumulates the per-group change d[k]
// in LDS.  What every chunk changes alike -- counts[k], and the per-group
// totals of the feature -- is NOT added with atomics (every workgroup on every
// XCD would hit the same K addresses; such device-scope atomics serialise at
// the memory side): the chunk leaves its d[] as one row of a staging matrix
// and k_vs_reduce sums the rows per group.  What only this chunk touches --
// the categorical cell (k, x) -- is updated in place:
//   DD/DPD: cnt[k][x] += d[k]            (reduce: counts, count_sum += sum_c d)
//   BB:     reduce: counts += sum_c d, (x ? heads : tails) += sum_{c: x} d
//   GP/BNB: reduce: counts, count += sum_c d, sum += sum_c x_c d
// `stage` null (matrix too large: wide value tables): the atomics as before.
constexpr int kVsApplyBlock = 1024;   // one workgroup per chunk: keep the CU busy

// SORT: also reorder the chunk's rows by their NEW group (counting sort in
// LDS, written out coalesced), in place in sorted_rows.  Next time this batch
// range is sampled, the rows of a tile then sit in a narrow band of groups,
// so almost every chunk of the likelihood vector is free of own slots (see
// vs_sum_and_scan).  The order is a performance hint only: results do not
// depend on it.
template <int KIND, bool SORT>
__global__ __launch_bounds__(kVsApplyBlock) void k_vs_apply(
        SweepParams P, StatImage img, const VsTile * __restrict__ chunks,
        uint32_t * __restrict__ sorted_rows,
        const uint32_t * __restrict__ p2g, uint32_t * __restrict__ assign_pos,
        uint32_t nvals, int refresh_cells, int sole_owner,
        int32_t * __restrict__ stage, VsDefer D, VsOffsets O) {
    extern __shared__ int vs_lds[];
    const int K = sweep_K(P);
    int * delta = vs_lds;                 // [K]
    int * hist = vs_lds + K;              // [K]           (SORT)
    int * part = hist + K;                // [kVsApplyBlock / 64]  (SORT)
    uint32_t * rows_l = (uint32_t *)(part + kVsApplyBlock / 64);  // [kVsApplyRows]
    uint32_t * gn_l = rows_l + kVsApplyRows;               // [kVsApplyRows]
    uint32_t * rows_s = gn_l + kVsApplyRows;               // sorted copies
    uint32_t * gid_s = rows_s + kVsApplyRows;
    const uint32_t x = chunks[blockIdx.x].x;
    const uint32_t pos = chunks[blockIdx.x].pos;
    const uint32_t n = chunks[blockIdx.x].n;
    if (x == 0xFFFFFFFEu) return;   // several values: k_vs_apply_mixed's
    // (the offsets of the groups' rows after the sort, for k_vs_tables: the
    // sorting form of a device-normalised run stamps them, anything else
    // that changes the rows' groups leaves the stamp at 0)
    if (O.epoch && threadIdx.x == 0)
        O.epoch[blockIdx.x] =
            (SORT && O.off && P.dev) ? (uint32_t)P.dev->pad : 0u;
    // The rows of this chunk that the tiles handed over (VsDefer: alone in
    // their group, own score above the value's maximum) -- or the whole chunk
    // when its values lie beyond the tables -- are sampled here, a wave per
    // row as k_rows_wave does it, before the moves are added up: the strips
    // lie where the sort keeps its copies later on.  (GP's float statistics
    // want the moves before this kernel runs: the launch in between stays.)
    if (SORT && KIND != DIST_GP && D.chunk_counts) {
        const uint32_t n_def = x >= nvals ? n : D.chunk_counts[blockIdx.x];
        if (n_def) {   // (uniform over the workgroup)
            __shared__ uint32_t s_exp[1024];
            for (int i = threadIdx.x; i < 1024; i += kVsApplyBlock)
                s_exp[i] = g_tables_dev.exp_table[i];
            __syncthreads();
            const float ea = u2f(g_tables_dev.exp_ab[0]);
            const float eb = u2f(g_tables_dev.exp_ab[1]);
            const int strip = (K + 63) & ~63;
            const int waves = min(kVsApplyBlock / 64, 4 * kVsApplyRows / strip);
            const int wave = threadIdx.x >> 6;
            // (16-byte aligned: the recurrences read them as float4; the host
            // leaves four words of slack behind the sort's buffers)
            float * sl = reinterpret_cast<float *>(
                             ((unsigned long long)rows_l + 15ull) & ~15ull)
                         + (size_t)wave * strip;
            if (wave < waves)
                for (uint32_t item = wave; item < n_def; item += waves) {
                    const uint32_t at = x >= nvals ? pos + item
                                                   : D.list[pos + item];
                    wave_row_update<KIND, -1, 1>(
                        P, sl, s_exp, ea, eb, K, threadIdx.x & 63,
                        P.row_begin + sorted_rows[at], assign_pos[at], at);
                }
            __syncthreads();   // (their moves are read below)
            if (threadIdx.x == 0 && x < nvals) D.chunk_counts[blockIdx.x] = 0;
        }
    }
    for (int k = threadIdx.x; k < K; k += kVsApplyBlock) {
        delta[k] = 0;
        if (SORT) hist[k] = 0;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += kVsApplyBlock) {
        const uint32_t go = P.old_packed[pos + i], gn = P.new_packed[pos + i];
        if (go != gn) {
            atomicAdd(&delta[go], -1);
            atomicAdd(&delta[gn], 1);
            if ((KIND == DIST_GP || KIND == DIST_BNB) && x >= nvals) {
                // the chunk of counts beyond the value table: every row
                // brings its own value to the sums
                const int32_t v = (int32_t)P.values[0][P.row_begin
                                                      + sorted_rows[pos + i]];
                atomicAdd(&img.i1[0][go], -v);
                atomicAdd(&img.i1[0][gn], v);
            }
        }
        if (SORT) {
            rows_l[i] = sorted_rows[pos + i];
            gn_l[i] = gn;
            atomicAdd(&hist[gn], 1);
        } else {
            assign_pos[pos + i] = p2g[gn];
        }
    }
    __syncthreads();
    const int dim = P.feat[0].dim;
    // A fused batch's chunks sample the rows they were handed while their
    // siblings are already here: a handed-over row that is NOT alone in its
    // group reads the cell (its group, x) as the batch found it
    // (entry_after_remove), so the chunks of a value that has several must
    // not change that cell under it -- k_vs_reduce adds their staged deltas
    // to it after this launch (VsTile::chunk of a chunk: how many chunks its
    // value has).
    const bool defer_cells = SORT && stage && D.chunk_counts
                             && chunks[blockIdx.x].chunk > 1u;
    for (int k = threadIdx.x; k < K; k += kVsApplyBlock) {
        const int dlt = delta[k];
        if (stage) stage[(size_t)blockIdx.x * P.K + k] = dlt;
        if (dlt == 0) continue;
        if (!stage) {
            atomicAdd(&img.counts[k], dlt);
            if (KIND == DIST_BB) {
                atomicAdd(x ? &img.i0[0][k] : &img.i1[0][k], dlt);
            } else {
                atomicAdd(&img.i0[0][k], dlt);     // count_sum / count
                if ((KIND == DIST_GP || KIND == DIST_BNB) && x < nvals)
                    atomicAdd(&img.i1[0][k], dlt * (int32_t)x);   // sum
            }
        }
        if ((KIND == DIST_DD || KIND == DIST_DPD) && !defer_cells) {
            int32_t * cell = &img.cnt[0][(size_t)k * dim + x];
            int before;
            if (sole_owner) {   // one chunk per value: nobody else is here
                before = *cell;
                *cell = before + dlt;
            } else {
                before = atomicAdd(cell, dlt);
            }
            if (refresh_cells) {
                // this workgroup is the only one that touches cell (k, x)
                // (one chunk per value, live statistics): leave its cache
                // entry current (dd.hpp:458-467) and spare the batch's tail
                // a rebuild of all K * dim cells
                const SlaveView & s = P.feat[0];
                s.S[(size_t)x * s.cap + k] =
                    fast_log(s.prior[x] + (float)(before + dlt));
            }
        }
    }
    if (!SORT) return;
    // exclusive scan of hist over k: each thread owns a contiguous slice,
    // the slices are scanned within the wave by shuffles and the 16 wave
    // totals by every thread for itself (two barriers in all)
    const int per = (K + kVsApplyBlock - 1) / kVsApplyBlock;
    const int lo = threadIdx.x * per;
    const int hi = lo + per < K ? lo + per : K;
    int sum = 0;
    for (int k = lo; k < hi; ++k) sum += hist[k];
    int incl = sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; ++w) run += part[w];
    int * off = (O.off && P.dev)
                    ? O.off + (size_t)blockIdx.x * O.stride : nullptr;
    for (int k = lo; k < hi; ++k) {
        const int c = hist[k];
        hist[k] = run;
        if (off) off[k] = run;
        run += c;
    }
    // (off[K] = the end; groups that do not exist yet have no rows: the
    // reader is told how many there were, in the row's last word)
    if (off && threadIdx.x == 0) {
        off[K] = (int)n;
        off[O.stride - 1] = K;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += kVsApplyBlock) {
        const uint32_t gn = gn_l[i];
        const int p = atomicAdd(&hist[gn], 1);
        rows_s[p] = rows_l[i];
        gid_s[p] = p2g[gn];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += kVsApplyBlock) {
        sorted_rows[pos + i] = rows_s[i];
        assign_pos[pos + i] = gid_s[i];
    }
}

// k_vs_apply for a chunk that holds the rows of SEVERAL values (categorical
// kinds; VsTile::x == kVsMixedChunk): where a value has only a few rows per
// batch (C5: 100), one workgroup per value would spend its time on O(K) LDS
// passes.  The chunk covers whole values, so this workgroup is still the only
// one that touches their cells (k, x): per-group changes go to LDS and the
// staging matrix as before, the cells take one atomic per moved row and end,
// and (refresh_cells) the touched cache entries are rewritten from the final
// counts.  Rows keep their order (tiles hold one value each).
constexpr uint32_t kVsMixedChunk = 0xFFFFFFFEu;
template <int KIND>
__global__ __launch_bounds__(kVsApplyBlock) void k_vs_apply_mixed(
        SweepParams P, StatImage img, const VsTile * __restrict__ chunks,
        const uint32_t * __restrict__ sorted_rows,
        const uint32_t * __restrict__ p2g, uint32_t * __restrict__ assign_pos,
        int refresh_cells, int32_t * __restrict__ stage) {
    extern __shared__ int vs_lds[];
    const int K = sweep_K(P);
    int * delta = vs_lds;                 // [K]
    if (chunks[blockIdx.x].x != kVsMixedChunk) return;   // k_vs_apply's
    const uint32_t pos = chunks[blockIdx.x].pos;
    const uint32_t n = chunks[blockIdx.x].n;
    const int dim = P.feat[0].dim;
    for (int k = threadIdx.x; k < K; k += kVsApplyBlock) delta[k] = 0;
    // a thread's rows side by side: every step below is a round trip to a
    // matrix far larger than the caches, and the steps of one row depend on
    // each other -- the rows' do not
    constexpr int R = kVsApplyRows / kVsApplyBlock;
    uint32_t go[R], gn[R], x[R];
    bool moved[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = threadIdx.x + r * kVsApplyBlock;
        moved[r] = false;
        go[r] = gn[r] = x[r] = 0;
        if (i < n) {
            go[r] = P.old_packed[pos + i];
            gn[r] = P.new_packed[pos + i];
            x[r] = sorted_rows[pos + i];
            moved[r] = go[r] != gn[r];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (moved[r]) x[r] = P.values[0][P.row_begin + x[r]];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = threadIdx.x + r * kVsApplyBlock;
        if (i < n) assign_pos[pos + i] = p2g[gn[r]];
        if (!moved[r]) continue;
        atomicAdd(&delta[go[r]], -1);
        atomicAdd(&delta[gn[r]], 1);
        // (workgroup scope: no other workgroup touches these cells in this
        // launch, and an agent-scope atomic is performed at the memory side
        // of the eight XCDs' L2s -- measured 208 us per launch against ...)
        __hip_atomic_fetch_add(&img.cnt[0][(size_t)go[r] * dim + x[r]], -1,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&img.cnt[0][(size_t)gn[r] * dim + x[r]], 1,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // the cell updates are complete before the refresh
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int k = threadIdx.x; k < K; k += kVsApplyBlock) {
        const int dlt = delta[k];
        if (stage) {
            stage[(size_t)blockIdx.x * P.K + k] = dlt;
        } else if (dlt != 0) {
            atomicAdd(&img.counts[k], dlt);
            atomicAdd(&img.i0[0][k], dlt);     // count_sum
        }
    }
    if (!refresh_cells) return;
    // dd.hpp:458-467 for every touched cell, from the counts as they now stand
    // (a cell moved by several rows is rewritten by each of them, alike)
    const SlaveView & s = P.feat[0];
    int c_old[R], c_new[R];
    float prior[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        c_old[r] = c_new[r] = 0;
        prior[r] = 0.f;
        if (!moved[r]) continue;
        prior[r] = s.prior[x[r]];
        c_old[r] = __hip_atomic_load(
            &img.cnt[0][(size_t)go[r] * dim + x[r]], __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_WORKGROUP);
        c_new[r] = __hip_atomic_load(
            &img.cnt[0][(size_t)gn[r] * dim + x[r]], __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!moved[r]) continue;
        s.S[(size_t)x[r] * s.cap + go[r]] = fast_log(prior[r] + (float)c_old[r]);
        s.S[(size_t)x[r] * s.cap + gn[r]] = fast_log(prior[r] + (float)c_new[r]);
    }
}

// The per-group sums of the staged chunk deltas (see k_vs_apply): thread
// (k, slice) adds up a slice of the chunks, the slices meet in LDS, and the
// owner of k updates the statistics without atomics.
constexpr int kVsReduceGroups = 16;    // groups per workgroup
constexpr int kVsReduceSlices = 32;
template <int KIND>
__global__ __launch_bounds__(kVsReduceGroups * kVsReduceSlices)
void k_vs_reduce(StatImage img, const int32_t * __restrict__ stage,
                 const VsTile * __restrict__ chunks, uint32_t n_chunks, int K,
                 uint32_t nvals, unsigned long long * host_pairs,
                 unsigned int seq, const DevState * dev, int k_limit,
                 const uint32_t * __restrict__ multi, uint32_t n_multi,
                 int dim) {
    // (the rows of the staging matrix are the host's bound apart: their
    // addresses do not wait for the group count of record; k_limit: what
    // that count can be at most at this batch)
    const int stride = K;
    if (dev) K = dev->K;   // (see SweepParams::dev)
    __shared__ int s_a[kVsReduceSlices][kVsReduceGroups];
    __shared__ int s_b[kVsReduceSlices][kVsReduceGroups];
    const int kk = threadIdx.x % kVsReduceGroups;
    const int slice = threadIdx.x / kVsReduceGroups;
    const int k = blockIdx.x * kVsReduceGroups + kk;
    int a = 0, b = 0;   // a: plain sum; b: BB heads part / GP value-weighted
    if (k < k_limit) {   // (slots past the group count: nothing is used)
        for (uint32_t c = slice; c < n_chunks; c += kVsReduceSlices) {
            const int d = stage[(size_t)c * stride + k];
            const uint32_t x = chunks[c].x;
            a += d;
            if (KIND == DIST_BB) b += x ? d : 0;
            if (KIND == DIST_GP || KIND == DIST_BNB)
                b += x < nvals ? d * (int32_t)x : 0;
        }
    }
    // the cells k_vs_apply left to this kernel (a fused batch's values with
    // several chunks; multi[] = {value, first chunk, chunks} each): thread
    // (k, slice) owns cell (k, x) of the slice's values
    if ((KIND == DIST_DD || KIND == DIST_DPD) && k < k_limit)
        for (uint32_t m = slice; m < n_multi; m += kVsReduceSlices) {
            const uint32_t x = multi[3 * m], c0 = multi[3 * m + 1],
                           nc = multi[3 * m + 2];
            int d = 0;
            for (uint32_t c = c0; c < c0 + nc; ++c)
                d += stage[(size_t)c * stride + k];
            if (d) img.cnt[0][(size_t)k * dim + x] += d;
        }
    s_a[slice][kk] = a;
    s_b[slice][kk] = b;
    __syncthreads();
    if (slice != 0 || k >= K) return;
    for (int q = 1; q < kVsReduceSlices; ++q) {
        a += s_a[q][kk];
        b += s_b[q][kk];
    }
    // host_pairs: the new group sizes go straight into pinned host memory,
    // each with the batch's ticket in the upper half of ONE 8-byte store; the
    // host polls until every slot carries the ticket (k_publish_counts and
    // its launch are not needed on this path)
    const int32_t size_now = img.counts[k] + a;
    if (host_pairs)
        host_pairs[k] = ((unsigned long long)seq << 32) | (uint32_t)size_now;
    if (a == 0 && b == 0) return;
    img.counts[k] = size_now;
    if (KIND == DIST_BB) {
        img.i0[0][k] += b;        // heads
        img.i1[0][k] += a - b;    // tails
    } else {
        img.i0[0][k] += a;        // count_sum / count
        if (KIND == DIST_GP || KIND == DIST_BNB) img.i1[0][k] += b;   // sum
    }
}

// (grid-stride: a bounded number of atomics on the one result word)
__global__ void k_max_value(const uint32_t * __restrict__ values, size_t n,
                            uint32_t * out) {
    uint32_t v = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        v = max(v, values[i]);
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}

// row order <-> value-sorted position order
__global__ void k_pos_gather(const uint32_t * __restrict__ by_row,
                             const uint32_t * __restrict__ sorted_rows,
                             uint32_t * __restrict__ by_pos, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) by_pos[i] = by_row[sorted_rows[i]];
}
__global__ void k_pos_scatter(const uint32_t * __restrict__ by_pos,
                              const uint32_t * __restrict__ sorted_rows,
                              uint32_t * __restrict__ by_row, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) by_row[sorted_rows[i]] = by_pos[i];
}

// counting sort of a batch's rows by value (one-time per batch range).  A
// workgroup counts its rows in LDS first, so a small value domain does not
// serialise on a handful of global counters.
constexpr int kVsSortBins = 4096;   // LDS bins; larger domains go global
constexpr int kVsSortRows = 4096;   // rows per workgroup

__global__ __launch_bounds__(kBlock) void k_vs_hist(
        const uint32_t * __restrict__ values, size_t row_begin, size_t n,
        uint32_t nvals, uint32_t * __restrict__ hist) {
    __shared__ uint32_t bins[kVsSortBins];
    const bool local = nvals + 1 <= kVsSortBins;
    if (local) {
        for (uint32_t i = threadIdx.x; i <= nvals; i += kBlock) bins[i] = 0;
        __syncthreads();
    }
    const size_t lo = (size_t)blockIdx.x * kVsSortRows;
    const size_t hi = lo + kVsSortRows < n ? lo + kVsSortRows : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += kBlock) {
        const uint32_t x = values[row_begin + i];
        const uint32_t b = x < nvals ? x : nvals;   // last bin: outside the table
        if (local) atomicAdd(&bins[b], 1u); else atomicAdd(&hist[b], 1u);
    }
    if (local) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i <= nvals; i += kBlock)
            if (bins[i]) atomicAdd(&hist[i], bins[i]);
    }
}
__global__ __launch_bounds__(kBlock) void k_vs_scatter(
        const uint32_t * __restrict__ values, size_t row_begin, size_t n,
        uint32_t nvals, uint32_t * __restrict__ cursor,
        uint32_t * __restrict__ sorted_rows) {
    __shared__ uint32_t bins[kVsSortBins];
    const bool local = nvals + 1 <= kVsSortBins;
    const size_t lo = (size_t)blockIdx.x * kVsSortRows;
    const size_t hi = lo + kVsSortRows < n ? lo + kVsSortRows : n;
    if (!local) {
        for (size_t i = lo + threadIdx.x; i < hi; i += kBlock) {
            const uint32_t x = values[row_begin + i];
            sorted_rows[atomicAdd(&cursor[x < nvals ? x : nvals], 1u)] =
                (uint32_t)i;
        }
        return;
    }
    // count, reserve one range per value for the whole workgroup, then place
    for (uint32_t i = threadIdx.x; i <= nvals; i += kBlock) bins[i] = 0;
    __syncthreads();
    for (size_t i = lo + threadIdx.x; i < hi; i += kBlock) {
        const uint32_t x = values[row_begin + i];
        atomicAdd(&bins[x < nvals ? x : nvals], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= nvals; i += kBlock)
        bins[i] = bins[i] ? atomicAdd(&cursor[i], bins[i]) : 0u;
    __syncthreads();
    for (size_t i = lo + threadIdx.x; i < hi; i += kBlock) {
        const uint32_t x = values[row_begin + i];
        sorted_rows[atomicAdd(&bins[x < nvals ? x : nvals], 1u)] = (uint32_t)i;
    }
}

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
// clear: leave the image zeroed for the next batch (the library's own
// exchange buffer is never memset again).  host_pairs: segment 0 is the group
// sizes; their new values go to pinned host memory with the batch's ticket
// (see k_vs_reduce).
__global__ void k_add_words(WordSegments seg, int32_t * __restrict__ src,
                            size_t total, int clear,
                            unsigned long long * host_pairs,
                            unsigned int seq) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
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

// code:4 | feature:4 | group:28 | detail:28 -- the lowest key wins
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
