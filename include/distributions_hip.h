/* distributions_hip.h -- C ABI of libdistributions_hip.so
 *
 * MI355X (gfx950) implementation of the collapsed-Gibbs mixture hot path of
 * forcedotcom/distributions v2.0.28.  The reference has no C ABI: its boundary
 * is C++ templates (include/distributions/{mixture,clustering}.hpp,
 * the models headers) wrapped by Cython (distributions/lp/).  Each entry point below
 * names the reference member it stands in for (paths relative to the
 * reference root).  include/distributions_hip.hpp re-presents these as the
 * reference's C++ classes; distributions_amd/lp re-presents them as the
 * reference's Python classes.
 *
 * Conventions
 *   - every call returns 0 on success, non-zero on error; dist_last_error()
 *     then holds the message (the reference throws std::runtime_error under
 *     DIST_THROW_ON_ERROR, common.hpp:49-57).  Bounds are always checked.
 *   - numeric state (sufficient statistics, score caches) lives in HBM; host
 *     pointers are caller-owned and copied in/out, like the numpy<->VectorFloat
 *     copies of lp/vector.pyx:33-47.  Pointers named *_dev are device memory.
 *   - values cross the ABI as 32-bit words: int for DirichletDiscrete /
 *     BetaBernoulli / GammaPoisson / DirichletProcessDiscrete, IEEE float bits
 *     for NormalInverseChiSq.
 *   - group ids are "packed" ids (mixture.hpp:41-46) unless named global.
 */
#ifndef DISTRIBUTIONS_HIP_H
#define DISTRIBUTIONS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIST_ABI_VERSION 1

enum dist_kind {
    DIST_DD = 0,   /* models/dd.hpp   DirichletDiscrete<256>          */
    DIST_BB = 1,   /* models/bb.hpp   BetaBernoulli                   */
    DIST_GP = 2,   /* models/gp.hpp   GammaPoisson                    */
    DIST_NICH = 3, /* models/nich.hpp NormalInverseChiSq              */
    DIST_DPD = 4,  /* models/dpd.hpp  DirichletProcessDiscrete (dense
                      value remap: value v in [0,dim), OTHER = 2^32-1) */
    DIST_BNB = 5   /* models/bnb.hpp  BetaNegativeBinomial            */
};

#define DIST_DD_MAX_DIM 256
#define DIST_DPD_OTHER 0xFFFFFFFFu
#define DIST_MAX_FEATURES 8

/* Model::Shared (dd.hpp:57-86, bb.hpp:54-76, gp.hpp:52-81, nich.hpp:52-95,
 * dpd.hpp:59-153) */
typedef struct dist_shared {
    int kind;
    int dim;             /* DD: dim; DPD: number of known values            */
    float p[4];          /* BB: alpha,beta | GP: alpha,inv_beta |
                            NICH: mu,kappa,sigmasq,nu | DPD: alpha,beta0 |
                            BNB: alpha,beta,r (r a positive integer)        */
    float alphas[DIST_DD_MAX_DIM]; /* DD                                    */
    const float * betas; /* DPD: betas[dim], host pointer, copied           */
} dist_shared_t;

/* Model::Group as 32-bit words (dd.hpp:89-93, bb.hpp:79-82, gp.hpp:84-88,
 * nich.hpp:98-102, dpd.hpp:157-158):
 *   DD / DPD: { count_sum, counts[dim] }     BB:   { heads, tails }
 *   GP:       { count, sum, log_prod(f32) }  NICH: { count, mean(f32),
 *                                                    count_times_variance(f32) }
 *   BNB:      { count, sum }                                  (bnb.hpp:87-89) */
size_t dist_group_words(const dist_shared_t * shared);

int dist_abi_version(void);
const char * dist_last_error(void);
int dist_device_count(int * count);
int dist_set_device(int device);
int dist_synchronize(void);
/* The HIP stream (hipStream_t, NULL = the default stream) that carries every
 * launch and copy the CALLING THREAD makes from now on.  Independent engines
 * driven from different host threads on different streams overlap on the
 * device; objects must be used on the stream they were created on (or after
 * the caller has synchronised the two). */
int dist_set_stream(void * hip_stream);

/* ---- entropy: rng_t = std::default_random_engine (random_fwd.hpp:34) ----
 * The engine state is one word; sample_unif01 (random.hpp:47-50) consumes
 * exactly one step. */
uint32_t dist_rng_seed(uint64_t seed);            /* engine.seed(seed)      */
uint32_t dist_rng_next(uint32_t * state);         /* engine()               */
float dist_rng_unif01(uint32_t * state);          /* sample_unif01          */
uint32_t dist_rng_jump(uint32_t state, uint64_t steps);

/* ---- special.hpp / vector_math.hpp on device (host arrays in/out) -------- */
int dist_vector_log(size_t n, const float * in, float * out);      /* vector_math.cc:224-255 */
int dist_vector_exp(size_t n, const float * in, float * out);      /* :190-221 */
int dist_vector_lgamma(size_t n, const float * in, float * out);   /* :257-272 */
int dist_vector_lgamma_nu(size_t n, const float * in, float * out);/* :275-291 */
int dist_vector_log_factorial(size_t n, const uint32_t * in, float * out); /* special.hpp:208-214 */

/* ---- discrete sampling (random.hpp:316-392, random.cc:77-128) ------------ */
/* sample_from_scores_overwrite: scores[n] (host) become likelihoods */
int dist_sample_from_scores_overwrite(uint32_t * rng_state, size_t n,
                                      float * scores, size_t * sample_out);
int dist_scores_to_likelihoods(size_t n, float * scores, float * total_out);
int dist_sample_from_likelihoods(uint32_t * rng_state, size_t n,
                                 const float * likelihoods, float total,
                                 size_t * sample_out);
int dist_log_sum_exp(size_t n, const float * scores, float * out);

/* ---- Clustering<int>::PitmanYor (clustering.hpp:58-123) ------------------ */
int dist_py_score_add_value(float alpha, float d, int group_size,
                            int nonempty_group_count, int sample_size,
                            int empty_group_count, float * out);
int dist_py_score_remove_value(float alpha, float d, int group_size,
                               int nonempty_group_count, int sample_size,
                               int empty_group_count, float * out);

/* sample_assignments(size, rng): sequential CRP/Pitman-Yor draw of `size`
 * group ids (clustering.cc:67-142); host-side like the reference */
int dist_py_sample_assignments(float alpha, float d, int size,
                               uint32_t * rng_state, int * assignments_out);
/* score_counts(counts): log probability of a partition (clustering.cc:152-183) */
int dist_py_score_counts(float alpha, float d, const int * counts,
                         size_t group_count, float * out);

/* ---- Clustering<int>::LowEntropy (clustering.hpp:245-331) ----------------
 * dataset_size is the model; nonempty_group_count is accepted and unused, as
 * in the reference.  score_counts: clustering.cc:229-248 (n log n terms summed
 * in binary64, 1e-6 relative); log_partition_function: clustering.cc:204-215.
 * sample_assignments: clustering.cc:250-283, host-side like the reference
 * (its totals follow the release build's vector_sum association). */
int dist_le_score_add_value(int dataset_size, int group_size,
                            int nonempty_group_count, int sample_size,
                            int empty_group_count, float * out);
int dist_le_score_remove_value(int dataset_size, int group_size,
                               int nonempty_group_count, int sample_size,
                               int empty_group_count, float * out);
int dist_le_log_partition_function(int sample_size, float * out);
int dist_le_sample_assignments(int dataset_size, int sample_size,
                               uint32_t * rng_state, int * assignments_out);
int dist_le_score_counts(int dataset_size, const int * counts,
                         size_t group_count, float * out);
/* LowEntropy::Mixture = MixtureDriver<LowEntropy, int> (mixture.hpp:48-163) */
typedef struct dist_le_mixture dist_le_mixture_t;
dist_le_mixture_t * dist_le_mixture_create(void);
void dist_le_mixture_destroy(dist_le_mixture_t * m);
int dist_le_mixture_init(dist_le_mixture_t * m, const int * counts,
                         size_t group_count);                 /* mixture.hpp:59-71   */
int dist_le_mixture_add_value(dist_le_mixture_t * m, size_t groupid,
                              int * added_out);               /* mixture.hpp:73-92   */
int dist_le_mixture_remove_value(dist_le_mixture_t * m, size_t groupid,
                                 int * removed_out);          /* mixture.hpp:94-122  */
int dist_le_mixture_score_value(const dist_le_mixture_t * m, int dataset_size,
                                float * scores, size_t size); /* mixture.hpp:124-141 */
int dist_le_mixture_score_data(const dist_le_mixture_t * m, int dataset_size,
                               float * out);                  /* mixture.hpp:143-145 */
size_t dist_le_mixture_size(const dist_le_mixture_t * m);
size_t dist_le_mixture_sample_size(const dist_le_mixture_t * m);
int dist_le_mixture_counts(const dist_le_mixture_t * m, int * out);

/* ---- PitmanYor::Mixture = CachedMixture (clustering.hpp:126-234) --------- */
typedef struct dist_py_mixture dist_py_mixture_t;
dist_py_mixture_t * dist_py_mixture_create(void);
void dist_py_mixture_destroy(dist_py_mixture_t * m);
/* counts() = counts; init(model)                       clustering.hpp:151-161 */
int dist_py_mixture_init(dist_py_mixture_t * m, float alpha, float d,
                         const int * counts, size_t group_count);
/* add_value(model, groupid) -> group was empty         clustering.hpp:163-176 */
int dist_py_mixture_add_value(dist_py_mixture_t * m, float alpha, float d,
                              size_t groupid, int * added_out);
/* remove_value(model, groupid) -> group became empty   clustering.hpp:178-193 */
int dist_py_mixture_remove_value(dist_py_mixture_t * m, float alpha, float d,
                                 size_t groupid, int * removed_out);
/* score_value(model, scores): OVERWRITES scores[size]  clustering.hpp:195-208 */
int dist_py_mixture_score_value(const dist_py_mixture_t * m, float alpha,
                                float d, float * scores, size_t size);
/* score_data(model) = model.score_counts(counts())      clustering.hpp:210-212 */
int dist_py_mixture_score_data(const dist_py_mixture_t * m, float alpha,
                               float d, float * out);
size_t dist_py_mixture_size(const dist_py_mixture_t * m);           /* counts().size() */
size_t dist_py_mixture_sample_size(const dist_py_mixture_t * m);    /* sample_size()   */
int dist_py_mixture_counts(const dist_py_mixture_t * m, int * out); /* counts()        */
/* empty_groupids(): writes at most cap ids, returns how many exist */
size_t dist_py_mixture_empty_groupids(const dist_py_mixture_t * m,
                                      size_t * out, size_t cap);

/* ---- Model::Mixture = MixtureSlave<Model, ...> (mixture.hpp:340-450) ----- */
typedef struct dist_mixture dist_mixture_t;
dist_mixture_t * dist_mixture_create(const dist_shared_t * shared);
void dist_mixture_destroy(dist_mixture_t * m);
int dist_mixture_clear(dist_mixture_t * m);                        /* groups().clear()      */
int dist_mixture_append(dist_mixture_t * m, const uint32_t * group);/* groups().push_back(g) */
int dist_mixture_get_group(const dist_mixture_t * m, size_t groupid,
                           uint32_t * group_out);                  /* groups(i) (copy)      */
size_t dist_mixture_size(const dist_mixture_t * m);                /* groups().size()       */
int dist_mixture_init(dist_mixture_t * m);                         /* init        :354-359  */
int dist_mixture_add_group(dist_mixture_t * m);                    /* add_group   :361-368  */
int dist_mixture_remove_group(dist_mixture_t * m, size_t groupid); /* remove_group:370-375  */
int dist_mixture_add_value(dist_mixture_t * m, size_t groupid,
                           uint32_t value);                        /* add_value   :377-384  */
int dist_mixture_remove_value(dist_mixture_t * m, size_t groupid,
                              uint32_t value);                     /* remove_value:386-398  */
int dist_mixture_score_value_group(const dist_mixture_t * m, size_t groupid,
                                   uint32_t value, float * out);   /* :400-414 */
/* score_value: ACCUMULATES into scores_accum[size]                   :416-425 */
int dist_mixture_score_value(const dist_mixture_t * m, uint32_t value,
                             float * scores_accum, size_t size);

/* score_data: log marginal likelihood of all groups' data    mixture.hpp:427-431
 * Accumulated in the reference's own float order (per-value chains closed by
 * vector_sum for DirichletDiscrete, one chain over the groups otherwise):
 * bit-exact against a float restatement of its loops.  DirichletProcessDiscrete
 * walks a hash map in the reference: its terms are summed in binary64,
 * 1e-5 relative. */
int dist_mixture_score_data(const dist_mixture_t * m, float * out);
/* score_data_grid: scores_out[i] = score_data under candidate shareds[i]
 * (hyper-parameter grid, mixture.hpp:433-438 / 238-247; DirichletDiscrete
 * carries alpha_sum from candidate to candidate like dd.hpp:259-284,320-344).
 * Every candidate is scored in the same launches; same contract. */
int dist_mixture_score_data_grid(const dist_mixture_t * m,
                                 const dist_shared_t * shareds, size_t n,
                                 float * scores_out);

/* score_value for n values in ONE launch (extension; mixture.hpp:416-425 per
 * value): scores_accum[r * ld + k] accumulates the score of values[r] in group
 * k, bit for bit what n calls of dist_mixture_score_value leave.  The per-value
 * call costs a launch and a round trip (measured with the reference's own
 * harness: 0.03 cells/us at every K); this is the member to
 * use between one value and a whole dist_gibbs_t. */
int dist_mixture_score_values(const dist_mixture_t * m,
                              const uint32_t * values, size_t n,
                              float * scores_accum, size_t ld);
/* Mixture::validate (mixture.hpp:440-444; dd.hpp:447-455 and the other value
 * scorers'): group counts agree, count_sum == sum of counts, and the value
 * scorer's cache equals Scorer::init of the statistics bit for bit
 * (recomputed on the device).  Non-zero + dist_last_error() otherwise. */
int dist_mixture_validate(const dist_mixture_t * m);

/* ---- Model::Group scalar API (host side, O(1); dd.hpp:113-199 etc.) ------ */
int dist_group_init(const dist_shared_t * shared, uint32_t * group);
int dist_group_add_value(const dist_shared_t * shared, uint32_t * group,
                         uint32_t value);
int dist_group_remove_value(const dist_shared_t * shared, uint32_t * group,
                            uint32_t value);
int dist_group_score_value(const dist_shared_t * shared,
                           const uint32_t * group, uint32_t value,
                           float * out);
int dist_group_score_data(const dist_shared_t * shared, const uint32_t * group,
                          float * out);
/* Model::Scorer (dd.hpp:222-245, bb.hpp:185-205, gp.hpp:198-217,
 * nich.hpp:239-259, bnb.hpp:195-223, dpd.hpp:309-341): init caches what eval
 * needs of ONE group's statistics, eval scores a value from the cache -- the
 * per-group ("naive") counterpart of the Mixture's vectorised score_value,
 * the other half of benchmarks/mixture.cc:41-74,119-135.  Host side, O(dim)
 * / O(1); the state is dist_scorer_words(shared) floats owned by the caller:
 * alpha_sum and alphas[dim] for DirichletDiscrete, the logarithm for OTHER
 * and for each of the dim values for DirichletProcessDiscrete, the scalar
 * members of the reference's Scorer otherwise. */
size_t dist_scorer_words(const dist_shared_t * shared);
int dist_scorer_init(const dist_shared_t * shared, const uint32_t * group,
                     float * state);
int dist_scorer_eval(const dist_shared_t * shared, const float * state,
                     uint32_t value, float * out);

/* ---- DirichletProcessDiscrete::Shared, the stick-breaking side -------------
 * (dpd.hpp:59-101; lp/models/_dpd.pyx:31-45).  Host side like the reference's:
 * gamma, alpha, beta0, the betas of the values that exist and how many rows
 * carry each.  A value owns a DENSE SLOT for life (the index groups and the
 * kernels count it under); a value whose last row leaves gives its beta back
 * to beta0 and its slot to the next new value.
 *   add_value     dpd.hpp:66-74  a first row of a new value breaks
 *                 beta0 * sample_beta_safe(rng, 1, gamma, MIN_BETA) off the stick
 *   remove_value  dpd.hpp:76-83
 *   realize       dpd.hpp:85-101 new values until beta0 <= 1e-4 or 9 999 exist,
 *                 the rest of the stick to one last value; beta0 = 0
 *   load          protobuf_load dpd.hpp:103-124 (beta0 = max(0, 1 - sum betas))
 *   view          the dist_shared_t the mixtures take: dim = slots, betas by
 *                 slot (0 for a free slot); the pointer is the object's own
 *                 storage, valid until its next mutation (`version` moves)
 *   slot          value -> dense slot (OTHER -> OTHER); unknown values fail
 *   dump          by slot: value (0xFFFFFFFF = free), beta, count
 * Entropy: rng_t's one word of state, advanced as libstdc++'s
 * std::gamma_distribution<double> over std::default_random_engine advances it
 * (random.hpp:87-119). */
typedef struct dist_dpd_shared dist_dpd_shared_t;
dist_dpd_shared_t * dist_dpd_shared_create(void);
void dist_dpd_shared_destroy(dist_dpd_shared_t * s);
int dist_dpd_shared_copy(dist_dpd_shared_t * dst,
                         const dist_dpd_shared_t * src);  /* Shared's copy  */
int dist_dpd_shared_load(dist_dpd_shared_t * s, float gamma, float alpha,
                         const uint32_t * values, const float * betas,
                         const int * counts, size_t n);
int dist_dpd_shared_add_value(dist_dpd_shared_t * s, uint32_t value,
                              uint32_t * rng_state);
int dist_dpd_shared_remove_value(dist_dpd_shared_t * s, uint32_t value);
int dist_dpd_shared_realize(dist_dpd_shared_t * s, uint32_t * rng_state);
size_t dist_dpd_shared_slots(const dist_dpd_shared_t * s);
size_t dist_dpd_shared_size(const dist_dpd_shared_t * s);
uint64_t dist_dpd_shared_version(const dist_dpd_shared_t * s);
int dist_dpd_shared_params(const dist_dpd_shared_t * s, float * gamma,
                           float * alpha, float * beta0);
int dist_dpd_shared_view(const dist_dpd_shared_t * s, dist_shared_t * out);
int dist_dpd_shared_slot(const dist_dpd_shared_t * s, uint32_t value,
                         uint32_t * slot_out);
int dist_dpd_shared_dump(const dist_dpd_shared_t * s, uint32_t * values,
                         float * betas, int * counts);
/* sample_gamma / sample_beta_safe (random.hpp:87-97,110-119) over rng_t */
int dist_sample_gamma(uint32_t * rng_state, float alpha, float beta,
                      float * out);
int dist_sample_beta_safe(uint32_t * rng_state, float alpha, float beta,
                          float min_value, float * out);

/* ---- protobuf wire format of Shared / Group -------------------------------
 * The messages of distributions/io/schema.proto (package
 * protobuf.distributions) as bytes, what Group::protobuf_dump / protobuf_load
 * exchange through generated classes in the reference (dd.hpp:94-111,
 * bb.hpp:84-93, gp.hpp:90-101, nich.hpp:104-115, dpd.hpp:161-180) -- here
 * without libprotobuf.  dump: buf == NULL asks for the size (*len_out).
 * keys (DirichletProcessDiscrete only): keys[i] = the value that dense index
 * i stands for; NULL = identity. */
int dist_group_protobuf_dump(const dist_shared_t * shared,
                             const uint32_t * group, const uint32_t * keys,
                             uint8_t * buf, size_t cap, size_t * len_out);
int dist_group_protobuf_load(const dist_shared_t * shared,
                             const uint32_t * keys, const uint8_t * data,
                             size_t len, uint32_t * group_out);
/* Shared messages of DD / BB / GP / NICH (DPD's Shared carries the
 * stick-breaking state, which lives in the lp layer) */
int dist_shared_protobuf_dump(const dist_shared_t * shared, uint8_t * buf,
                              size_t cap, size_t * len_out);
int dist_shared_protobuf_load(int kind, const uint8_t * data, size_t len,
                              dist_shared_t * shared_out);

/* ---- MixtureIdTracker (mixture.hpp:460-521) ------------------------------ */
typedef struct dist_id_tracker dist_id_tracker_t;
dist_id_tracker_t * dist_id_tracker_create(void);
void dist_id_tracker_destroy(dist_id_tracker_t * t);
int dist_id_tracker_init(dist_id_tracker_t * t, size_t group_count);
int dist_id_tracker_add_group(dist_id_tracker_t * t);
int dist_id_tracker_remove_group(dist_id_tracker_t * t, uint32_t packed);
int dist_id_tracker_packed_to_global(const dist_id_tracker_t * t,
                                     uint32_t packed, uint32_t * global_out);
int dist_id_tracker_global_to_packed(const dist_id_tracker_t * t,
                                     uint32_t global, uint32_t * packed_out);
size_t dist_id_tracker_packed_size(const dist_id_tracker_t * t);
size_t dist_id_tracker_global_size(const dist_id_tracker_t * t);

/* ==== batched row engine (extension; SURVEY 8b "new batched entry points") ==
 * One PitmanYor driver + n_features slaves + an id tracker over a resident
 * table of rows -- the loop of examples/mixture/main.py:236-244 /
 * benchmarks/mixture.cc:104-115 with sampling, run for many rows per launch.
 *
 * Batch semantics: every row of a batch is scored against the
 * state at batch entry minus itself (exactly what remove_value leaves behind,
 * including group removal when the row was alone), draws with engine step
 * (draw_base + global row index + 1) of `seed_state`, and all moves are then
 * applied in row order.  A batch of one row is the reference's sequential
 * update. */
typedef struct dist_gibbs dist_gibbs_t;
dist_gibbs_t * dist_gibbs_create(float alpha, float d, int n_features,
                                 const dist_shared_t * shareds);
/* the same engine under the LowEntropy clustering model (generic driver
 * scores, mixture.hpp:124-141); dataset_size >= the total number of rows */
dist_gibbs_t * dist_gibbs_create_low_entropy(int dataset_size, int n_features,
                                             const dist_shared_t * shareds);
void dist_gibbs_destroy(dist_gibbs_t * g);

/* Rows that have no group yet: `empty_groups` empty groups, every row
 * unassigned (mixture.init(model) of a fresh mixture, examples/mixture/
 * main.py:222-224).  dist_gibbs_init_sequential then assigns rows
 * [row_begin, row_end) in order -- it must start at the first unassigned row
 * -- by the initialisation loop of examples/mixture/main.py: score every
 * group for the row (prior_only: with the clustering model alone, main.py:
 * 227-232; else with all features, main.py:265-270), sample, add; nothing is
 * removed; one engine step per row from *rng_state, which is advanced. */
int dist_gibbs_load_rows_unassigned(dist_gibbs_t * g, size_t n_rows,
                                    const uint32_t * const * values,
                                    int empty_groups, uint64_t row_offset);
int dist_gibbs_init_sequential(dist_gibbs_t * g, size_t row_begin,
                               size_t row_end, uint32_t * rng_state,
                               int prior_only);
/* Make n_rows rows resident.  values[f] -> n_rows words; assign_packed[i] in
 * [0, nonempty_groups); empty_groups >= 1 empty groups are appended
 * (mixture.hpp:152-162).  row_offset = global index of local row 0 (multi-GPU
 * shards).  The *_dev form takes device pointers and keeps using them. */
int dist_gibbs_load_rows(dist_gibbs_t * g, size_t n_rows,
                         const uint32_t * const * values,
                         const uint32_t * assign_packed, int nonempty_groups,
                         int empty_groups, uint64_t row_offset);
int dist_gibbs_load_rows_dev(dist_gibbs_t * g, size_t n_rows,
                             const uint32_t * const * values_dev,
                             uint32_t * assign_packed_dev, int nonempty_groups,
                             int empty_groups, uint64_t row_offset);
/* multi-GPU: the statistics were built from local rows only; add the other
 * shards' (dist_gibbs_stat_words() int32 words, all-reduced by the caller) */
size_t dist_gibbs_stat_words(const dist_gibbs_t * g);   /* (size_t)-1: failed, see dist_last_error */
int dist_gibbs_export_stats_dev(const dist_gibbs_t * g, int32_t * stats_dev);
int dist_gibbs_import_stats_dev(dist_gibbs_t * g, const int32_t * stats_dev);

/* one Gibbs pass over local rows [row_begin,row_end) in batches of
 * batch_rows, single GPU */
int dist_gibbs_sweep(dist_gibbs_t * g, size_t row_begin, size_t row_end,
                     size_t batch_rows, uint32_t seed_state,
                     uint64_t draw_base);
/* the reference's sequential chain (examples/mixture/main.py:236-244 over
 * mixture.hpp:73-122, 361-398: remove the row, score, sample, add -- a batch
 * of one row), advancing *rng_state one step per row.  Device-resident: one
 * workgroup walks the range, group creation and removal included. */
int dist_gibbs_sweep_sequential(dist_gibbs_t * g, size_t row_begin,
                                size_t row_end, uint32_t * rng_state);
/* M independent exact chains in ONE launch (BASELINE configs[3]: "8
 * independent chains"): engine i -- its own rows, statistics, id maps --
 * runs dist_gibbs_sweep_sequential over ITS rows [row_begin, row_end) with
 * rng_states[i], one workgroup per chain, side by side on the GPU; each
 * chain's result is exactly that of its own sequential call.  The engines
 * share one feature list (same kinds, any hyper-parameters) and are distinct;
 * up to two chains fit a compute unit (512 on an MI355X run concurrently). */
int dist_gibbs_sweep_sequential_many(dist_gibbs_t * const * engines, size_t m,
                                     size_t row_begin, size_t row_end,
                                     uint32_t * rng_states);
/* the same pass in phases, for callers that exchange statistics between
 * GPUs: sample -> delta -> [all-reduce delta] -> apply_delta ->
 * (ordered statistics: moves -> [all-gather] -> replay_ordered) -> finish.
 * delta is dist_gibbs_stat_words() int32 words of device memory and carries
 * the statistics that add over ranks (group sizes, DD/DPD/BB counts, GP count
 * and sum). */
int dist_gibbs_batch_sample(dist_gibbs_t * g, size_t row_begin, size_t row_end,
                            uint32_t seed_state, uint64_t draw_base);
int dist_gibbs_batch_delta_dev(dist_gibbs_t * g, int32_t * delta_dev);
int dist_gibbs_batch_apply_delta_dev(dist_gibbs_t * g,
                                     const int32_t * delta_dev);
int dist_gibbs_batch_apply_local(dist_gibbs_t * g);
/* "float_stats" = 1 (merged; opt-in, tolerance-level): the order-dependent
 * statistics -- NormalInverseChiSq's count / mean / count_times_variance
 * (nich.hpp:125-165), GammaPoisson's log_prod (gp.hpp:115,134) -- are updated
 * from binary64 SUMS per group (change of the count, of sum x, of sum x^2; of
 * log_prod) instead of being replayed row by row: equal to the running
 * updates to binary32 rounding, not bit for bit.  The sums add over ranks:
 * dist_gibbs_float_delta_words() doubles per image (0: not in this mode);
 * between dist_gibbs_batch_apply_delta_dev and dist_gibbs_batch_finish a
 * multi-rank caller takes its open batch's image, all-reduces it and hands it
 * back; dist_gibbs_sweep_sharded does so itself.  After every rank loaded
 * ITS rows: export, all-reduce, import (the import replaces the rank's own). */
size_t dist_gibbs_float_delta_words(const dist_gibbs_t * g);
int dist_gibbs_batch_float_delta_dev(dist_gibbs_t * g, double * delta_dev);
int dist_gibbs_batch_apply_float_delta_dev(dist_gibbs_t * g,
                                           const double * delta_dev);
int dist_gibbs_export_float_moments_dev(dist_gibbs_t * g, double * out_dev);
int dist_gibbs_import_float_moments_dev(dist_gibbs_t * g,
                                        const double * image_dev);
int dist_gibbs_batch_finish(dist_gibbs_t * g);
/* Order-dependent statistics -- all of NormalInverseChiSq's (Welford updates,
 * nich.hpp:125-165) and GammaPoisson's log_prod (gp.hpp:115,134) -- do not add
 * over ranks: they are replayed in global row order on every replica.
 * ordered_features: how many features carry such statistics.
 * batch_moves_dev: the open batch's moves in row order (slot indices of the
 * batch snapshot, identical on every replica); call after batch_delta_dev.
 * replay_ordered_dev: row i of the gathered list (rank order) leaves slot
 * old_slot[i] (NULL: additions only) and joins new_slot[i]; 0xFFFFFFFF in
 * new_slot marks padding; values_dev[f] -> n_rows words for every ordered
 * feature f (other entries may be NULL).  reset != 0 returns those statistics
 * to Group::init first (replay of the whole data set after load_rows). */
int dist_gibbs_ordered_features(const dist_gibbs_t * g, int * count_out);
int dist_gibbs_batch_moves_dev(dist_gibbs_t * g, uint32_t * old_slot_dev,
                               uint32_t * new_slot_dev);
int dist_gibbs_replay_ordered_dev(dist_gibbs_t * g,
                                  const uint32_t * old_slot_dev,
                                  const uint32_t * new_slot_dev,
                                  const uint32_t * const * values_dev,
                                  size_t n_rows, int reset);

/* ---- the library's own communicator ----------------------------------------
 * Optional: with it the whole multi-GPU sweep (sample -> delta -> all-reduce
 * -> apply -> finish, per sub-sweep) runs inside the library, the all-reduce
 * on the engine's own stream.  Two transports, chosen by the id:
 *   RCCL (dist_comm_unique_id): bound at run time (the librccl.so.1 already
 *     in the process, e.g. PyTorch's, else ROCm's); one process per GPU.
 *   host (dist_comm_unique_id_host): ranks that SHARE a GPU -- RCCL refuses
 *     two ranks on one device -- meet in a POSIX shared-memory segment and
 *     the all-reduce is staged through the host (drains the stream: for tests
 *     and single-GPU rehearsals of the multi-rank protocol).  It checks that
 *     all ranks issue the same collective and fails the call on every rank
 *     ("ranks diverged") instead of hanging when they do not, or when a peer
 *     does not arrive within DIST_COMM_TIMEOUT_S seconds (default 120).
 * Rank 0 makes the id and hands the 128 bytes to the other ranks by whatever
 * channel the application has (torch.distributed broadcast, MPI, a file);
 * every rank then calls dist_comm_create (collective).  Engines with
 * order-dependent statistics (NormalInverseChiSq, GammaPoisson's log_prod) are
 * refused by dist_gibbs_sweep_sharded: they exchange rows (see above), or
 * sums with "float_stats" = 1. */
typedef struct dist_comm dist_comm_t;
int dist_comm_available(void);                       /* 1 if RCCL can be bound */
int dist_comm_unique_id(uint8_t id_out[128]);
int dist_comm_unique_id_host(uint8_t id_out[128]);
dist_comm_t * dist_comm_create(const uint8_t id[128], int rank, int world);
void dist_comm_destroy(dist_comm_t * c);
int dist_comm_size(const dist_comm_t * c, int * rank_out, int * world_out);
/* in-place all-reduce of device memory on the calling thread's stream, waited
 * for: type 0 = int32, 1 = binary64; op 0 = sum, 1 = min (what the engines'
 * own exchanges use; for the application's set-up steps, e.g. the statistics
 * after every rank loaded its rows) */
int dist_comm_all_reduce_dev(dist_comm_t * c, void * data_dev, size_t count,
                             int type, int op);
/* n_batches sub-sweeps over the local rows in batches of batch_rows (ranks
 * whose shard is exhausted take part with empty batches: pass the same
 * n_batches on every rank).  Per sub-sweep ONE all-reduce of
 *   4 + min(bound, K0 + j * empty_groups) * (3 + dim)   int32 words
 * (K0: the group count the ranks' run began with, j: its batches so far --
 * the live part of the group set, not the bound buffers are sized by), or of
 *   4 + 3 * min(bound, K0 + j * empty_groups)
 * on value-partitioned ranks (below).  The 4 header words let the ranks tell
 * that one of them left the common run (dist_last_error: "ranks diverged";
 * the engine is unusable then) -- between two passes a rank may LOOK at its
 * engine (counts, assignments, groups, validate, statistics export, timers)
 * but not change it. */
int dist_gibbs_sweep_sharded(dist_gibbs_t * g, dist_comm_t * c,
                             size_t n_batches, size_t batch_rows,
                             uint32_t seed_state, uint64_t draw_base);
/* Value-partitioned ranks (one categorical feature: DirichletDiscrete,
 * DirichletProcessDiscrete): when the rows are placed so that no value has
 * rows on two ranks, a rank's rows only ever touch the cells counts[.][x] of
 * ITS values (dd.hpp:123-149, dpd.hpp:430-469) -- those cells never travel,
 * and what the ranks exchange per sub-sweep is the group sizes and the
 * per-group totals, 3 words per group (C2: 12 KB instead of 1.1 MB; C5: 98 KB
 * instead of 328 MB).  partition_by_value (collective, after load_rows and
 * the initial statistics exchange) checks the placement -- a value with rows
 * on two ranks fails the call on every rank -- and switches
 * dist_gibbs_sweep_sharded to that exchange.  From then on the cells of OTHER
 * ranks' values are stale here: entry points that read whole groups
 * (get_group, validate, export_stats, score_data) fail until gather_cells
 * (collective: one all-reduce of the owned cells) has made the replicas
 * whole again.  Results are those of the unpartitioned exchange, bit for
 * bit. */
int dist_gibbs_partition_by_value(dist_gibbs_t * g, dist_comm_t * c);
int dist_gibbs_gather_cells(dist_gibbs_t * g, dist_comm_t * c);
/* the exchanges of dist_gibbs_sweep_sharded since the last reset: out[0] =
 * all-reduces issued, out[1] = int32 words sent in all of them, out[2] = the
 * largest, out[3] = the last one (header included; does not close a run) */
int dist_gibbs_comm_volume(dist_gibbs_t * g, uint64_t out[4], int reset);

/* batch-semantics scores of one resident row (length written to *size_out;
 * scores_out needs dist_gibbs_group_count() floats) */
int dist_gibbs_row_scores(dist_gibbs_t * g, size_t row, float * scores_out,
                          size_t * size_out);
/* score_values extension: scores[r * ld + k] for rows [row_begin,row_end)
 * against the current state (no self-removal), device output */
int dist_gibbs_score_rows_dev(dist_gibbs_t * g, size_t row_begin,
                              size_t row_end, float * scores_dev, size_t ld);

size_t dist_gibbs_group_count(const dist_gibbs_t * g);     /* counts().size() */
size_t dist_gibbs_row_count(const dist_gibbs_t * g);
int dist_gibbs_counts(const dist_gibbs_t * g, int * out);  /* driver counts() */
int dist_gibbs_assignments(const dist_gibbs_t * g, uint32_t * global_out);
int dist_gibbs_get_group(const dist_gibbs_t * g, int feature, size_t groupid,
                         uint32_t * group_out);
int dist_gibbs_packed_to_global(const dist_gibbs_t * g, uint32_t packed,
                                uint32_t * global_out);
int dist_gibbs_global_to_packed(const dist_gibbs_t * g, uint32_t global,
                                uint32_t * packed_out);
/* MixtureIdTracker::global_size (mixture.hpp:517): ids handed out so far */
size_t dist_gibbs_global_size(const dist_gibbs_t * g);
/* Mixture::validate for the whole engine (mixture.hpp:152-163 the driver's
 * _validate, :440-444 the slaves'), as far as the rows themselves can vouch:
 * every row's group id is live; group sizes, and every integer statistic of
 * every feature (count_sum and cells, heads/tails, count/sum; the count of a
 * NormalInverseChiSq group), equal a recount from the rows on the device; the
 * host's mirror of the group set (sizes, empty groups, id maps) equals the
 * device's; there is an empty group; sizes sum to the rows assigned.  Closes
 * an open device-normalised run first; fails while a batch is open.
 * Returns 0 when consistent; 2 with dist_last_error() and *report (may be
 * NULL) naming the first inconsistency; 1 when the check itself failed.
 * Cost: one pass of atomics over the rows + O(K * dim); a diagnostic. */
typedef struct dist_validate_report {
    int code;            /* 0 ok | 1 dead id (group = row, detail = id) |
                            2 value out of range (group = row) | 3 group size |
                            4 statistic 0 | 5 statistic 1 | 6 count cell
                            (detail = value) | 7 host mirror               */
    int feature;         /* -1: the driver                                  */
    long long group;     /* packed group index (or the row, codes 1-2)      */
    long long detail;
    long long expected;  /* the recount                                     */
    long long found;     /* the live statistic                              */
    long long rows_assigned;
    char what[96];
} dist_validate_report_t;
int dist_gibbs_validate(dist_gibbs_t * g, dist_validate_report_t * report);

/* whether THIS rank could run a sharded pass of n_batches batches of
 * batch_rows rows with the group set normalised on the device (a diagnostic:
 * dist_gibbs_sweep_sharded asks every rank itself and takes the device path
 * only when all of them can; "sharded_device_normalise", default 1, set to 0
 * on a rank keeps all of them on the host-normalised loop) */
int dist_gibbs_sharded_device_normalise_ok(const dist_gibbs_t * g,
                                           size_t n_batches,
                                           size_t batch_rows, int * ok_out);
/* Options.  Ten are public; each names the test that exercises it.  None but
 * the last two changes a result.
 *
 *  "value_sorted"  0 generic kernels only | 1 auto (default) | 2 the
 *      value-sorted kernels whenever the feature list allows (ONE feature with
 *      a small value domain).  tests/test_gpu_sweep.py::test_batch_sweeps_bit_exact
 *  "value_stream"  0 per-value tables always | 1 auto (default: the table-free
 *      kernel k_vs_stream where a value has about one tile per batch -- C5) |
 *      2 always.  test_gpu_sweep.py::test_stream_kernel_on_full_tiles,
 *      test_gpu_fullsize.py::test_c5_stream_kernel_where_the_bench_runs_it
 *  "narrow_tiles"  0 never | 1 auto (default: launches too small to fill the
 *      chip take tiles of 64 rows, vectors from LDS: k_vs_narrow) | 2 whenever
 *      the vectors fit.  test_gpu_sweep.py::test_batch_sweeps_bit_exact (modes 4, 5)
 *  "device_normalise"  1 (default; 2 accepted) sweeps that stay on the
 *      value-sorted path with integer statistics normalise the group set on
 *      the device, no host round trip per batch; such a run stays OPEN when
 *      dist_gibbs_sweep / dist_gibbs_sweep_sharded return (the next sweep of
 *      the same tiling goes on with it, any other call pulls the host's
 *      mirrors first), so they return before the device has finished.  An
 *      open run is queued on the stream of the thread that swept; a call from
 *      another thread drains that stream before it reads the state.  0 never.
 *      test_gpu_sweep.py::test_device_side_normalisation_under_group_churn
 *  "sharded_device_normalise"  1 (default) lets dist_gibbs_sweep_sharded do the
 *      same: when a run is opened the ranks agree among themselves (one
 *      all-reduce of a flag) whether every one of them can; 0 keeps this
 *      rank, and so all of them, on the host-normalised loop.  Whether an
 *      open run GOES ON needs no word between them: it follows from the
 *      call's tiling and the batches the run has left, the same on every
 *      rank, and a rank that closed its run between two passes (any look at
 *      its state does) takes it up again with the same bound on the group
 *      count and the same batches left -- the collectives its peers issue.
 *      So READ-ONLY calls between two passes need not be the same on every
 *      rank; calls that change the state must be, as for any replicated state.
 *      tests/test_gpu_native_comm.py (a peek between passes: resumed_runs)
 *  "fused_tables"  1 (default) a device-normalised run spends ONE launch between
 *      a batch's statistics and the next batch's sampling (k_vs_tables: group
 *      set, caches, per-value tables) and the rows a tile hands over are
 *      sampled by k_vs_apply | 0 k_normalise, k_batch_finish, k_vs_prepare and
 *      k_rows_wave as launches of their own -- the path every batch takes
 *      that is NOT part of a device-normalised run of integer statistics
 *      (float statistics: GammaPoisson, NormalInverseChiSq; LowEntropy; more
 *      than 8128 groups; apply chunks of several values; the table-free
 *      kernel).  test_device_side_normalisation_under_group_churn,
 *      test_fused_batches_with_values_of_several_apply_chunks
 *  "kernel_timing"  n: HIP events around the score+sample kernel of every n-th
 *      batch feed dist_gibbs_kernel_stats (1, the default: every batch; 0
 *      none; two events cost a batch some 8 us).
 *      test_gpu_sweep.py::test_kernel_timing_samples_every_nth_batch
 *  "phase_timing"  1: events at a sub-sweep's phase boundaries feed
 *      dist_gibbs_phase_stats (bench.py's step_breakdown); default 0.
 *      tests/test_bench_cli.py
 *  "float_stats"  0 (default) float statistics by the ordered replay | 1 merged
 *      sums (dist_gibbs_float_delta_words): TOLERANCE-LEVEL.
 *      tests/test_gpu_scan.py, tests/test_gpu_fullsize_modes.py
 *  "sampling"  0 (default, the line of record: the reference's float
 *      operations in the reference's order, bit-identical assignments) | 1 SCAN
 *      SAMPLING, TOLERANCE-LEVEL: the same scores bit for bit and the same
 *      engine step per row, but the softmax and its inverse CDF by a running
 *      log-sum-exp and cumulative sums (random.hpp:316-333 in distribution;
 *      the index can differ where u * total falls within float rounding of a
 *      boundary between two groups).  tests/test_gpu_scan.py,
 *      tests/test_gpu_fullsize_modes.py
 *
 * Test hooks, spelled "debug.<name>": each forces a kernel variant the library
 * otherwise picks by itself, so that the differential fuzz (tools/fuzz.py,
 * tests/test_gpu_fuzz.py) reaches every variant at every size; not for
 * callers, and none changes a result.  sequential_chain (1 the device-resident
 * chain kernel | 0 rows as batches of one), running_sums_min_tiles (launches of
 * at least this many tiles use the per-value running sums and band tiles:
 * 2048), narrow_read_ahead (k_vs_narrow's instance: 0 by launch size | 4 | 8
 * float4s), stream_scratch (k_vs_stream keeps a tile's likelihoods between its
 * passes: 1 | 0 computes them again), rows_scratch (general rows: 3
 * k_rows_scratch | 0 k_sweep_program, which feature lists beyond 64 parameter
 * slots take anyway), rows_scratch_lds_log (FastLog's table in LDS: 1 | 0),
 * rows_scratch_block (threads per workgroup of that kernel: 512), rows_fold
 * (leading discrete features folded into a per-(joint value, group) table:
 * 1 where 128 rows share a value | 2 whenever the joint domain fits | 0),
 * apply_stage (general rows' integer statistics summed in LDS: 1 | 0),
 * program_all (every batch outside the value-sorted path through the
 * per-batch score program: 1 | 0), sample_prio / rows_prio (wave priorities
 * by phase in k_vs_sample + k_vs_stream / k_rows_scratch: 0x13210 set-up,
 * first pass, ..., last pass | 0 none -- the A/B of
 * profiles/r5_wave_priorities.txt), apply_overlap (k_vs_apply samples a
 * chunk's few handed-over rows while its other waves add up the moves: 1 | 0
 * before they do), run_batches_cap (a device-normalised run covers at most
 * this many batches: 0 as many as fit -- tests/test_gpu_native_ranks.py sees a
 * run used up and the ranks agree on the next one).
 */
int dist_gibbs_set_option(dist_gibbs_t * g, const char * name, int value);
/* how many batches each score+sample kernel has served */
int dist_gibbs_path_counts(const dist_gibbs_t * g, uint64_t * value_sorted,
                           uint64_t * generic);
/* diagnostics for the tests: out[0..15] = batches through the value-sorted
 * kernel, through the other kernels, launches with band tiles on, launches
 * with running sums on, values whose arg-max rows had their own tile in the
 * last value-sorted launch, rows that launch handed to the wave-per-row
 * kernel, value-sorted batches that took the table-free kernel, batches
 * whose group set the device normalised itself, value-sorted batches that
 * took the small-launch kernel, batches through k_rows_scratch, batches with
 * folded leading features, batches sampled in scan mode, batches with merged
 * float statistics, batches whose group set, caches and tables came from the
 * one fused launch, sharded runs this rank closed between two passes and took
 * up again, launches of the exact-chain kernel k_chains this engine issued
 * (first min(n, 16) entries are written) */
int dist_gibbs_debug_counts(dist_gibbs_t * g, uint64_t * out, size_t n);
/* "phase_timing" = 1 (a diagnostic: six events per sub-sweep): HIP-event time
 * (ms, summed) of the five phases of the device-normalised value-sorted
 * sub-sweeps since the last reset -- the per-value tables, the score+sample
 * kernel, the handed-over rows, the statistics, the group set and caches --
 * and how many sub-sweeps were timed */
int dist_gibbs_phase_stats(dist_gibbs_t * g, double ms_out[5],
                           uint64_t * batches_out, int reset);
/* HIP-event time (ms) and count of the all-reduces dist_gibbs_sweep_sharded
 * timed (every "kernel_timing"-th sub-sweep) since the last reset */
int dist_gibbs_comm_stats(dist_gibbs_t * g, double * ms_out,
                          uint64_t * launches_out, int reset);
/* HIP-event time (ms) and launch count of the score+sample kernel since the
 * last reset, measured on the engine's stream */
int dist_gibbs_kernel_stats(dist_gibbs_t * g, double * ms_out,
                            uint64_t * launches_out, uint64_t * rows_out,
                            int reset);

#ifdef __cplusplus
}
#endif
#endif /* DISTRIBUTIONS_HIP_H */
