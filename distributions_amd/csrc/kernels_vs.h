// The value-sorted row update (one feature with a small value domain): per-
// value tables (k_vs_prepare / k_vs_tables), k_vs_sample, k_vs_narrow,
// k_vs_stream, the scan-sampling variants.  Part of kernels.h.
#pragma once

namespace dist {

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
// tile sizes, so a tile's rows lie in one chunk.  (5120, not 4096: the 3906
// +- 62 rows a value has in the headline's 10^6-row batches went over 4096 for
// a value or two in two batches out of ten -- 258 chunks for 256 CUs that take
// one workgroup each: a second round, k_vs_apply 23 us instead of 14.5.)
constexpr int kVsApplyRows = 5120;
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
    // k_vs_apply: its dynamic LDS in bytes (0: not told).  What lies behind
    // the sort buffers -- sized by the host's BOUND on the group count, laid
    // out by the true one -- holds strips of K floats for the rows sampled
    // WHILE the chunk's other waves add up the moves
    int lds_bytes;
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
    // wave priorities by phase (0: none; 0x10000 | set-up << 12 | total <<
    // 8 | the scan's first half << 4 | its second half): see k_vs_sample
    int prio_mode;
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
    extern __shared__ __attribute__((aligned(16))) float s_l[];   // [2][Kpad] when the running sums are built
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
    extern __shared__ __attribute__((aligned(16))) float tb_lds[];
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
// s_setprio takes an immediate
__device__ __forceinline__ void vs_set_prio(int p) {
    switch (p & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
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
        const bool (&active)[kVsR], int (&found)[kVsR], int prio_steps
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
    if (prio_steps) vs_set_prio(prio_steps >> 4);
    for (int c = 0, k0 = 0; k0 < K; ++c, k0 += kVsUnroll) {
        float l[kVsUnroll];
        vs_fetch_chunk(lp, k0, l);
        if (prio_steps && c == nchunks / 2) vs_set_prio(prio_steps);
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
    // A SIMD issues from its oldest ready wave: at equal priority the waves
    // still gathering their rows get few slots beside the ones that run
    // their chains, reach their own chains late and finish them alone, one
    // dependent add at a time (tools/vs_stamps.py: the slowest waves spent
    // 65 k cycles in a set-up that takes the median wave 19 k; the launch
    // ended at 157 k, the median SIMD at 130 k).  With the set-up ahead of the
    // total, the total ahead of the scan and the scan's first half ahead of
    // its second, whoever is behind goes first and a SIMD's waves end
    // together: 69 -> 64 us per launch at C2 (profiles/r5_wave_priorities.txt).
    if (T.prio_mode) vs_set_prio(T.prio_mode >> 12);
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
    const bool both = __any(anyA) && __any(anyB);
    if (both) __builtin_amdgcn_s_setprio(3);
    else if (T.prio_mode) vs_set_prio(T.prio_mode >> 8);
#ifdef DIST_VS_STAMPS
    if (T.stamps) st1 = __builtin_amdgcn_s_memtime();
#endif
    if (__any(anyA)) {
        const float * vec = T.LA + (size_t)x * T.Kpad;
        int f[kVsR];
        vs_sum_and_scan(as_uniform(vec), vec,
                        T.PA ? as_uniform(T.PA + (size_t)x
                                          * (T.Kpad / kVsUnroll)) : nullptr,
                        K, g, l_own, u, inA, f, both ? 0 : T.prio_mode
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
                        K, g, l_own, u, inB, f, both ? 0 : T.prio_mode
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
    extern __shared__ __attribute__((aligned(16))) float s_scores[];   // [Kpad] when T.lds_scores
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
        float * __restrict__ scratch, uint32_t scratch_stride, int prio_mode) {
    // (wave priorities by phase, as in k_vs_sample: whoever is behind goes
    // first)
    if (prio_mode) vs_set_prio(prio_mode >> 12);
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
        if (prio_mode) vs_set_prio(pass == 0 ? prio_mode >> 8 : prio_mode >> 4);
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
            if (pass == 1 && prio_mode && k0 >= K / 2
                && k0 - kVsStreamChunk < K / 2)
                vs_set_prio(prio_mode);
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

// A handed-over row of a chunk whose value x is known (k_vs_apply), by one
// wave: RowScorer + wave_row_update's arithmetic to the letter, the loads in a
// different order.  A row that comes alone pays for latency, not throughput:
// every slot's cache entry and both driver vectors are requested BEFORE the
// chain g2p -> counts -> own entry that tells what the row's view is, so the
// row costs two trips to memory instead of the chain's three plus one per 256
// slots.  Leaves the move in old_packed / new_packed and returns it.
template <int KIND>
__device__ __forceinline__ void vs_deferred_row(
        const SweepParams & P, float * sl, const uint32_t * s_exp, float ea,
        float eb, int K, int lane, uint32_t x, size_t row,
        uint32_t global_id, size_t out, int & g_old, int & g_new) {
    SlaveView v = P.feat[0];
    v.kind = KIND;
    const float lf = KIND == DIST_GP ? fast_log_factorial(x) : 0.f;
    constexpr int U = 16;
    Entry e[U];
    float b[U], bs[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
        // (slots beyond K read the last one's: loaded, never used)
        const int k = min(lane + 64 * q, K - 1);
        e[q] = load_entry(v, k, x);
        b[q] = P.base[k];
        bs[q] = P.base_single[k];
    }
    // (RowScorer's constructor)
    const float shift = P.scalars->shift;
    const int g = P.g2p[global_id];
    const int n_g = P.counts[g];
    const bool singleton = (n_g == 1);
    const int Kl = K - (singleton ? 1 : 0);
    float s_own;
    if (!singleton) {
        s_own = accumulate(KIND, cluster_own_score(P, n_g - 1, shift),
                           entry_after_remove(v, g, x), x, lf, v.p);
    } else {   // slot g holds what was the last group
        const int src = K - 1;
        s_own = accumulate(KIND, P.base_single[src], load_entry(v, src, x), x,
                           lf, v.p);
    }
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < U; ++q) {
        const int k = lane + 64 * q;
        if (k < Kl) {
            const float one = bs[q], many = b[q];   // (by value)
            float s = accumulate(KIND, singleton ? one : many, e[q], x, lf,
                                 v.p);
            s = k == g ? s_own : s;
            sl[k] = s;
            m = s > m ? s : m;
        }
    }
    for (int k = lane + 64 * U; k < Kl; k += 64) {   // (K > 1024)
        float s = accumulate(KIND, singleton ? P.base_single[k] : P.base[k],
                             load_entry(v, k, x), x, lf, v.p);
        s = k == g ? s_own : s;
        sl[k] = s;
        m = s > m ? s : m;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    for (int k = lane; k < ((Kl + 63) & ~63); k += 64)
        sl[k] = k < Kl ? fast_exp_nonpos(sl[k] - m, s_exp, ea, eb) : 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float total = strip_total(sl, Kl);
    int g2 = strip_sample(sl, Kl, total * batch_row_unif01(P, row));
    if (singleton && g2 == g) g2 = K - 1;   // slot g held group K-1
    if (lane == 0) {
        P.old_packed[out] = (uint32_t)g;
        P.new_packed[out] = (uint32_t)g2;
    }
    g_old = g;
    g_new = g2;
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
    // row.  (GP's float statistics want the moves before this kernel runs:
    // the launch in between stays.)
    uint32_t n_def = 0;
    if (SORT && KIND != DIST_GP && D.chunk_counts)
        n_def = x >= nvals ? n : D.chunk_counts[blockIdx.x];
    __shared__ uint32_t s_exp[1024];
    if (n_def)   // (uniform over the workgroup)
        for (int i = threadIdx.x; i < 1024; i += kVsApplyBlock)
            s_exp[i] = g_tables_dev.exp_table[i];
    for (int k = threadIdx.x; k < K; k += kVsApplyBlock) {
        delta[k] = 0;
        if (SORT) hist[k] = 0;
    }
    __syncthreads();
    const float ea = u2f(g_tables_dev.exp_ab[0]);
    const float eb = u2f(g_tables_dev.exp_ab[1]);
    const int strip = (K + 63) & ~63;
    const int wave = threadIdx.x >> 6;
    // A few of them (the usual case: one): the LAST waves take a row each,
    // in strips of their own, while the others add up the moves of the rest
    // -- the row's two recurrences (some 1.5 K dependent adds) are longer
    // than everything else this workgroup does, and used to stand in front
    // of it.
    constexpr int kMaxOverlap = 4;
    const int strips_free =
        (D.lds_bytes - (int)((2 * K + kVsApplyBlock / 64 + 4 * kVsApplyRows + 4)
                             * sizeof(int))) / (strip * (int)sizeof(float));
    const int n_over = (SORT && KIND != DIST_GP && x < nvals
                        && (int)n_def <= min(strips_free, kMaxOverlap))
                           ? (int)n_def : 0;
    if (n_def && !n_over) {
        // many, or no room: every wave samples, in the strips where the sort
        // keeps its copies later on, before the moves are added up
        const int waves = min(kVsApplyBlock / 64, 4 * kVsApplyRows / strip);
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
    }
    if (n_def && threadIdx.x == 0 && x < nvals) D.chunk_counts[blockIdx.x] = 0;
    uint32_t skip[kMaxOverlap];
#pragma unroll
    for (int j = 0; j < kMaxOverlap; ++j)
        skip[j] = j < n_over ? D.list[pos + j] - pos : 0xFFFFFFFFu;
    const int first_over = kVsApplyBlock / 64 - n_over;
    if (SORT && KIND != DIST_GP && wave >= first_over) {
        float * sl = reinterpret_cast<float *>(
                         ((unsigned long long)(gid_s + kVsApplyRows) + 15ull)
                         & ~15ull)
                     + (size_t)(wave - first_over) * strip;
        // (whoever is behind goes first: these waves' recurrences are the
        // workgroup's critical path)
        __builtin_amdgcn_s_setprio(3);
        uint32_t i = 0;
#pragma unroll
        for (int j = 0; j < kMaxOverlap; ++j)
            if (wave == first_over + j) i = skip[j];
        const uint32_t at = pos + i;
        const uint32_t srow = sorted_rows[at];
        int go, gn;
        vs_deferred_row<KIND>(P, sl, s_exp, ea, eb, K, threadIdx.x & 63, x,
                              P.row_begin + srow, assign_pos[at], at, go, gn);
        if ((threadIdx.x & 63) == 0) {
            if (go != gn) {
                atomicAdd(&delta[go], -1);
                atomicAdd(&delta[gn], 1);
            }
            rows_l[i] = srow;
            gn_l[i] = (uint32_t)gn;
            atomicAdd(&hist[gn], 1);
        }
        __builtin_amdgcn_s_setprio(0);
    } else {
        const uint32_t step = (uint32_t)first_over * 64u;
        for (uint32_t i = threadIdx.x; i < n; i += step) {
            bool handed = false;
#pragma unroll
            for (int j = 0; j < kMaxOverlap; ++j) handed = handed || i == skip[j];
            if (handed) continue;
            const uint32_t go = P.old_packed[pos + i],
                           gn = P.new_packed[pos + i];
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
    const int lane = threadIdx.x & 63;
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
    // (slots past the group count of record: k_vs_apply staged nothing there
    // -- the words are stale -- and a delta image laid out for the live part
    // of the group set, Gibbs::exchange_K, has no room for them)
    (void)k_limit;
    if (k < K) {
        auto take = [&](int d, uint32_t x) {
            a += d;
            if (KIND == DIST_BB) b += x ? d : 0;
            if (KIND == DIST_GP || KIND == DIST_BNB)
                b += x < nvals ? d * (int32_t)x : 0;
        };
        // (four chunks' words in flight per step)
        uint32_t c = slice;
        for (; c + 3 * kVsReduceSlices < n_chunks; c += 4 * kVsReduceSlices) {
            int d[4];
            uint32_t x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d[q] = stage[(size_t)(c + q * kVsReduceSlices) * stride + k];
                x[q] = chunks[c + q * kVsReduceSlices].x;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) take(d[q], x[q]);
        }
        for (; c < n_chunks; c += kVsReduceSlices)
            take(stage[(size_t)c * stride + k], chunks[c].x);
    }
    // the cells k_vs_apply left to this kernel (a fused batch's values with
    // several chunks; multi[] = {value, first chunk, chunks} each): thread
    // (k, slice) owns cell (k, x) of the slice's values
    if ((KIND == DIST_DD || KIND == DIST_DPD) && k < K)
        for (uint32_t m = slice; m < n_multi; m += kVsReduceSlices) {
            const uint32_t x = multi[3 * m], c0 = multi[3 * m + 1],
                           nc = multi[3 * m + 2];
            // (eight loads in flight: a head value of Zipf data has some
            // thirty chunks, and one dependent trip per chunk made this
            // kernel 9.4 us where uniform values take 5.0)
            int d = 0;
            uint32_t c = c0;
            for (; c + 8 <= c0 + nc; c += 8) {
                int v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    v[q] = stage[(size_t)(c + q) * stride + k];
#pragma unroll
                for (int q = 0; q < 8; ++q) d += v[q];
            }
            for (; c < c0 + nc; ++c) d += stage[(size_t)c * stride + k];
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

}  // namespace dist
