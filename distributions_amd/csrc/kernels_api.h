// The object API's kernels: elementwise special functions, the scalar
// sampler, the PitmanYor driver's cache, the feature slaves (MixtureSlave and
// the per-model value scorers), score_data.  Part of kernels.h.
#pragma once

namespace dist {

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

}  // namespace dist
