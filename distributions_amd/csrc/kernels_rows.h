// The batched row update for ANY feature list: the generic kernels, the
// per-batch score program, k_rows_scratch (general rows, a lane per row), the
// device-resident sequential chain.  Part of kernels.h.
#pragma once

namespace dist {

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
    // (A.pad: wave priorities by pass -- a wave that is behind goes first, so
    // that a SIMD's waves end together instead of one after the other)
    const int prio = A.pad;   // 0x10000 | load << 12 | max << 8 | total << 4 | scan
    auto set_prio = [](int p) {
        switch (p & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
    };
    for (; tile < n_work; tile += stride) {
        if (prio) set_prio(prio >> 12);
        load_row(tile, cur);
        float m = -INFINITY;
        if (prio) set_prio(prio >> 8);
        for (int k0 = 0; k0 < K8; k0 += kRowsBlock) max_block(cur, k0, m);
        if (prio) set_prio(prio >> 4);
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
        if (prio) set_prio(prio);
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
    extern __shared__ __attribute__((aligned(16))) float wave_lds[];
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
    extern __shared__ __attribute__((aligned(16))) float chain_lds[];   // [K] scores, then likelihoods
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

// ---------------------------------------------------------------------------
// M exact chains in one launch (BASELINE configs[3] read literally: "8
// independent chains"; examples/mixture/main.py:236-244 per chain).  Workgroup
// m runs the reference's sequential chain of engine m -- its own rows,
// statistics, caches, id maps and entropy -- from the arguments in all[m].
// Unlike k_chain_rows the kernel takes the STRUCTURAL steps itself, so a chain
// never goes back to the host inside its range:
//   a row that is alone in its group: the group vanishes as
//     MixtureDriver::remove_value does it (mixture.hpp:108-119: the last
//     group moves into its slot), MixtureSlave::remove_group for every feature
//     (mixture.hpp:370-375), MixtureIdTracker::remove_group (:489-499), and
//     the empty groups' prior follows the count of non-empty ones
//     (clustering.hpp:221-230);
//   a row that fills an empty group: a fresh empty group is appended
//     (mixture.hpp:84-89, 361-368, 481-487; clustering.hpp:163-176).
// Per row the critical path is the two order-sensitive recurrences (total in
// index order, subtractive scan: 2 K dependent adds in the worst case, on one
// wave); everything around them is kept off it: the group sizes live in LDS,
// the next row's words are fetched under this row's recurrences, and this
// row's add_value runs beside the next row's remove_value on two waves (they
// touch different groups; the same group: one thread, in order).
struct ChainArgs {
    SweepParams P;         // the engine's views; rows [row_begin, row_end)
    float * base;          // the driver's score with one row out, per slot
    int32_t * counts;
    uint32_t * assign;
    uint32_t * p2g;
    int32_t * g2p;
    DevState * dev;        // in/out: K, nonempty, global_size
    ChainResult * result;
    uint32_t rng_state;
    int k_room;            // slots the arrays and the LDS strips hold
    uint32_t g_room;       // global ids the map holds
    // DIST_CHAIN_STAMPS: wave 0's cycles by phase (scores + max | exp |
    // recurrences | update | slow path), summed over the rows; else null
    unsigned long long * stamps;
};

template <class T>
__device__ __forceinline__ T chain_peek(const T * p) {
    // (written by this workgroup's own vector stores: never a scalar load)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// chain_value_op that also hands back the group's NEW cache entry as a row of
// value `next_value` will see it (the next row's score of this slot is
// patched from it, see k_chains): one round trip to memory for everything.
__device__ __forceinline__ Entry chain_value_op_next(
        const SlaveView & s, int k, uint32_t value, bool add,
        const uint32_t * log_tab, uint32_t next_value) {
    Entry e = {0.f, 0.f, 0.f, 0.f};
    if (is_cat(s.kind)) {
        const size_t cell = (size_t)k * s.dim + value;
        const bool same = next_value == value;
        const bool other = s.kind == DIST_DPD && next_value == DIST_DPD_OTHER;
        const int c_in = s.cnt[cell];
        const int n_in = s.i0[k];
        const float prior = s.prior[value];
        const float s_next =
            (same || other) ? 0.f : s.S[(size_t)next_value * s.cap + k];
        const int c2 = c_in + (add ? 1 : -1);
        const int n2 = n_in + (add ? 1 : -1);
        s.cnt[cell] = c2;
        s.i0[k] = n2;
        // dd.hpp:458-467 / dpd.hpp:458-470
        const float s_new = fast_log_t(prior + (float)c2, log_tab);
        e.c0 = fast_log_t(s.alpha_sum + (float)n2, log_tab);
        s.S[(size_t)value * s.cap + k] = s_new;
        s.c0[k] = e.c0;
        e.c1 = other ? s.other : same ? s_new : s_next;
        return e;
    }
    Stats st = load_stats(s, k);
    if (add) stats_add(s.kind, st, value); else stats_remove(s.kind, st, value);
    store_stats(s, k, st);
    e = scorer_init(s.kind, s.p, st);
    s.c0[k] = e.c0;
    s.c1[k] = e.c1;
    s.c2[k] = e.c2;
    s.c3[k] = e.c3;
    return e;
}

// a barrier for data that lives in LDS only: __syncthreads() also waits for
// the wave's outstanding global loads and stores (vmcnt), which here would
// put a trip to memory on the chain's critical path at every phase
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LOGL: FastLog's 64 KiB table in LDS (two chains fit a compute unit: the
// fastest single chain) or read where it lies (L2; the three logarithms of
// an update then cost it a second trip to memory, but four chains fit a CU:
// launches of more than two chains per CU take this instance)
template <int KIND0, int KIND1, int NF, bool LOGL>
__global__ __launch_bounds__(kBlock)
__attribute__((amdgpu_waves_per_eu(LOGL ? 2 : 4, LOGL ? 2 : 4)))
void k_chains(const ChainArgs * __restrict__ all) {
    const ChainArgs & A = all[blockIdx.x];
    const SweepParams & P = A.P;
    extern __shared__ __attribute__((aligned(16))) float chain_lds[];   // [room] scores, [room] group sizes
    __shared__ uint32_t s_exp[1024];
    __shared__ uint32_t s_log_lds[LOGL ? 16384 : 1];   // FastLog's table
    __shared__ float s_red[kBlock / 64];
    __shared__ int s_g2;
    __shared__ int s_patch_slot[2];
    __shared__ float s_patch_score[2];
    for (int i = threadIdx.x; i < 1024; i += kBlock)
        s_exp[i] = g_tables_dev.exp_table[i];
    if (LOGL)
        for (int i = threadIdx.x; i < 16384; i += kBlock)
            s_log_lds[i] = g_tables_dev.log_table[i];
    const uint32_t * const s_log =
        LOGL ? s_log_lds : g_tables_dev.log_table;
    const float ea = u2f(g_tables_dev.exp_ab[0]);
    const float eb = u2f(g_tables_dev.exp_ab[1]);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int room = (A.k_room + 63) & ~63;
    float * sc = chain_lds;
    int * size_l = reinterpret_cast<int *>(chain_lds + room);
    int K = A.dev->K, nonempty = A.dev->nonempty;
    uint32_t gsz = A.dev->global_size;
    const int nf = NF > 0 ? NF : P.F;
    const float shift = P.scalars->shift;
    float * base = A.base;
    uint32_t rng_state = A.rng_state;
    for (int k = tid; k < K; k += kBlock) size_l[k] = A.counts[k];
    __syncthreads();

    int kind[kMaxF];
#pragma unroll
    for (int f = 0; f < (NF > 0 ? NF : kMaxF); ++f) {
        if (f >= nf) break;
        kind[f] = f == 0 && KIND0 >= 0 ? KIND0
                : f == 1 && KIND1 >= 0 ? KIND1 : P.feat[f].kind;
    }
    // the driver's score of a group that keeps n >= 1 members
    auto own_score = [&](int n) {
        return P.cluster == 0 ? fast_log_t((float)n - P.d, s_log) + shift
                              : cluster_own_score(P, n, shift);
    };
    // clustering.hpp:221-230 with the row out: every empty group's score
    auto rescore_empties = [&]() {
        if (P.cluster != 0) return;
        const float es =
            py_empty_score(P.alpha, P.d, nonempty, P.n_empty) + shift;
        for (int k = tid; k < K; k += kBlock)
            if (size_l[k] == 0) base[k] = es;
    };
    // score_value of slot k for a row (driver, then every feature accumulates)
    auto score_slot = [&](int k, const uint32_t * xx, const float * lff) {
        float sk = base[k];
#pragma unroll
        for (int f = 0; f < (NF > 0 ? NF : kMaxF); ++f) {
            if (f >= nf) break;
            SlaveView v = P.feat[f];
            v.kind = kind[f];
            sk = accumulate(kind[f], sk, load_entry(v, k, xx[f]), xx[f],
                            lff[f], v.p);
        }
        return sk;
    };
    // add_value / remove_value on slot g for a group that stays (one thread);
    // returns the slot's score for the NEXT row (values xq, log-factorials lq)
    auto change = [&](int g, const uint32_t * xv, bool add,
                      const uint32_t * xq, const float * lq) {
        const int n = size_l[g] + (add ? 1 : -1);
        size_l[g] = n;
        float sk = own_score(n);
        base[g] = sk;
        for (int f = 0; f < nf; ++f) {
            SlaveView v = P.feat[f];
            v.kind = kind[f];
            const Entry e = chain_value_op_next(v, g, xv[f], add, s_log, xq[f]);
            sk = accumulate(kind[f], sk, e, xq[f], lq[f], v.p);
        }
        return sk;
    };
    // remove_value of the next row, by all threads (between two barriers):
    // the group stays, or vanishes with its last member
    auto remove_row = [&](int g, const uint32_t * xv) {
        const int members = size_l[g];
        __syncthreads();   // (everybody has read it before anybody writes)
        if (members != 1) {
            if (tid == 0) {
                const float none[kMaxF] = {};
                (void)change(g, xv, false, xv, none);
            }
            return;
        }
        const int last = K - 1;
        if (tid == 0) {
            const uint32_t gid = chain_peek(A.p2g + g);
            A.g2p[gid] = -1;
            if (g != last) {
                const uint32_t moved = chain_peek(A.p2g + last);
                A.p2g[g] = moved;
                A.g2p[moved] = g;
                size_l[g] = size_l[last];
                base[g] = base[last];
            }
        }
        if (g != last)
            for (int f = 0; f < nf; ++f) {
                const SlaveView & v = P.feat[f];
                if (tid == 0) {
                    v.i0[g] = v.i0[last]; v.i1[g] = v.i1[last];
                    v.f0[g] = v.f0[last]; v.f1[g] = v.f1[last];
                    v.c0[g] = v.c0[last]; v.c1[g] = v.c1[last];
                    v.c2[g] = v.c2[last]; v.c3[g] = v.c3[last];
                }
                if (is_cat(kind[f]))
                    for (int vv = tid; vv < v.dim; vv += kBlock) {
                        v.cnt[(size_t)g * v.dim + vv] =
                            v.cnt[(size_t)last * v.dim + vv];
                        v.S[(size_t)vv * v.cap + g] =
                            v.S[(size_t)vv * v.cap + last];
                    }
            }
        K = last;
        nonempty -= 1;
        __threadfence_block();
        __syncthreads();
        rescore_empties();
    };
    // a fresh empty group behind the others, by all threads
    auto append_group = [&]() {
        const int kn = K;
        for (int f = 0; f < nf; ++f) {
            SlaveView v = P.feat[f];
            v.kind = kind[f];
            if (is_cat(kind[f])) {
                for (int vv = tid; vv < v.dim; vv += kBlock) {
                    v.cnt[(size_t)kn * v.dim + vv] = 0;
                    v.S[(size_t)vv * v.cap + kn] =
                        fast_log_t(v.prior[vv] + 0.f, s_log);
                }
                if (tid == 0) {
                    v.i0[kn] = 0; v.i1[kn] = 0; v.f0[kn] = 0.f; v.f1[kn] = 0.f;
                    v.c0[kn] = fast_log_t(v.alpha_sum + 0.f, s_log);
                }
            } else if (tid == 0) {
                const Stats zero = {0, 0, 0.f, 0.f};
                store_stats(v, kn, zero);
                refresh_scalar_entry(v, kn);
            }
        }
        if (tid == 0) {
            size_l[kn] = 0;
            A.p2g[kn] = gsz;
            A.g2p[gsz] = kn;
            if (P.cluster != 0)
                base[kn] = le_score_add_value(P.dataset_size, 0,
                                              (int)P.sample_size - 1,
                                              P.n_empty);
        }
        K += 1;
        nonempty += 1;
        gsz += 1;
        __threadfence_block();
        __syncthreads();
        rescore_empties();
    };

    // The NEXT row's scores are made by waves 1..3 while wave 0 runs this
    // row's recurrences (the trips to memory of a row's K gathers are then
    // nobody's wait), kept in registers -- slot (tid - 64) + 192 j -- and
    // written to the strip once the recurrences are done; the two slots this
    // row's add_value and the next row's remove_value change are patched from
    // the values the updating threads hold anyway.
    constexpr int kPre = 12;
    constexpr int kPreLanes = kBlock - 64;
    float sn[kPre];
#pragma unroll
    for (int j = 0; j < kPre; ++j) sn[j] = 0.f;
    bool pre_ok = false;   // (uniform) sn[] holds this row's scores

    uint32_t done = 0;
    int event = 0;
    size_t row = P.row_begin;
    uint32_t x[kMaxF], xn[kMaxF];
    float lf[kMaxF], lfn[kMaxF];
    for (int f = 0; f < kMaxF; ++f) { x[f] = xn[f] = 0; lf[f] = lfn[f] = 0.f; }
    if (row < P.row_end) {
        if (K + 1 > A.k_room || gsz + 1 > A.g_room) {
            event = 3;
        } else {
            for (int f = 0; f < nf; ++f) {
                x[f] = P.values[f][row];
                lf[f] = kind[f] == DIST_GP ? fast_log_factorial(x[f]) : 0.f;
            }
            remove_row(chain_peek(A.g2p + A.assign[row]), x);
        }
    }
    unsigned long long acc[5] = {0, 0, 0, 0, 0};
    while (event == 0 && row < P.row_end) {
        const unsigned long long t_a = A.stamps ? clock64() : 0;
        // the next row's words, under this row's work
        const bool has_next = row + 1 < P.row_end;
        uint32_t a_next = 0;
        if (has_next) {
            a_next = A.assign[row + 1];
            for (int f = 0; f < nf; ++f) xn[f] = P.values[f][row + 1];
        }
        float m = -INFINITY;
        if (pre_ok) {
            if (wave != 0) {
                const int s0 = s_patch_slot[0], s1 = s_patch_slot[1];
#pragma unroll
                for (int j = 0; j < kPre; ++j) {
                    const int k = (tid - 64) + kPreLanes * j;
                    if (k < K) {
                        float sk = sn[j];
                        if (k == s0) sk = s_patch_score[0];
                        if (k == s1) sk = s_patch_score[1];
                        sc[k] = sk;
                        m = sk > m ? sk : m;
                    }
                }
            }
        } else {
            __threadfence_block();
            __syncthreads();
            for (int k = tid; k < K; k += kBlock) {
                const float sk = score_slot(k, x, lf);
                sc[k] = sk;
                m = sk > m ? sk : m;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(m, off);
            m = o > m ? o : m;
        }
        if (lane == 0) s_red[wave] = m;
        lds_barrier();
        const unsigned long long t_b = A.stamps ? clock64() : 0;
        m = s_red[0];
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) m = s_red[w] > m ? s_red[w] : m;
        // scores_to_likelihoods (random.cc:94-106): exponentials in parallel
        for (int k = tid; k < ((K + 63) & ~63); k += kBlock)
            sc[k] = k < K ? fast_exp_nonpos(sc[k] - m, s_exp, ea, eb) : 0.f;
        // (the statistics the last update wrote are everybody's from here on:
        // the stores are a phase old, waiting for them costs nothing)
        __threadfence_block();
        __syncthreads();
        const unsigned long long t_c = A.stamps ? clock64() : 0;
        // (no step of this row moves a group: the map is stable until then)
        const int gn = has_next ? chain_peek(A.g2p + a_next) : -1;
        const bool can_pre = has_next && K <= kPreLanes * kPre;
        if (has_next)
            for (int f = 0; f < nf; ++f)
                lfn[f] = kind[f] == DIST_GP ? fast_log_factorial(xn[f]) : 0.f;
        if (wave == 0) {
            const float total = strip_total(sc, K);
            rng_state = lcg_mulmod(rng_state, 16807u);
            const int g2 = strip_sample(sc, K, total * lcg_unif01(rng_state));
            if (lane == 0) s_g2 = g2;
        } else if (can_pre) {
#pragma unroll
            for (int j = 0; j < kPre; ++j) {
                const int k = (tid - 64) + kPreLanes * j;
                if (k < K) sn[j] = score_slot(k, xn, lfn);
            }
        }
        lds_barrier();
        const unsigned long long t_d = A.stamps ? clock64() : 0;
        const int g2 = s_g2;
        const int n2 = size_l[g2];
        const bool room_ok = K + 1 <= A.k_room && gsz + 1 <= A.g_room;
        const bool fast = n2 > 0 && has_next && room_ok
                          && (gn == g2 || size_l[gn] > 1);
        lds_barrier();   // (size_l is read above, written below)
        if (fast) {
            // add_value beside the next row's remove_value: different
            // groups on two waves, the same group in order on one thread
            if (tid == 0) {
                // (asked for first: one trip to memory for all of it)
                const uint32_t gid = chain_peek(A.p2g + g2);
                float sk = change(g2, x, true, xn, lfn);
                if (gn == g2) sk = change(gn, xn, false, xn, lfn);
                A.assign[row] = gid;
                s_patch_slot[0] = g2;
                s_patch_score[0] = sk;
                if (gn == g2) s_patch_slot[1] = -1;
            } else if (tid == 64 && gn != g2) {
                s_patch_score[1] = change(gn, xn, false, xn, lfn);
                s_patch_slot[1] = gn;
            }
            pre_ok = can_pre;
            lds_barrier();
        } else {
            if (tid == 0) {
                (void)change(g2, x, true, x, lf);
                A.assign[row] = chain_peek(A.p2g + g2);
            }
            __threadfence_block();
            __syncthreads();
            if (n2 == 0) append_group();
            if (has_next) {
                if (K + 1 > A.k_room || gsz + 1 > A.g_room) {
                    event = 3;   // (the next row is untouched)
                } else {
                    __syncthreads();
                    remove_row(gn, xn);
                }
            }
            pre_ok = false;
        }
        if (A.stamps) {
            const unsigned long long t_e = clock64();
            acc[0] += t_b - t_a;
            acc[1] += t_c - t_b;
            acc[2] += t_d - t_c;
            acc[fast ? 3 : 4] += t_e - t_d;
        }
        done += 1;
        row += 1;
        for (int f = 0; f < nf; ++f) { x[f] = xn[f]; lf[f] = lfn[f]; }
    }
    __threadfence_block();
    __syncthreads();
    for (int k = tid; k < K; k += kBlock) A.counts[k] = size_l[k];
    if (tid == 0 && A.stamps)
        for (int i = 0; i < 5; ++i) A.stamps[i] = acc[i];
    if (tid == 0) {
        A.dev->K = K;
        A.dev->nonempty = nonempty;
        A.dev->global_size = gsz;
        A.result->rng_state = rng_state;
        A.result->rows_done = done;
        A.result->event = event;
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

}  // namespace dist
