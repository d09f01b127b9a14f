// libdistributions_hip.so -- host side: device-resident mixture objects and
// the C ABI declared in include/distributions_hip.h.  gfx950 only.
//
// Structure mirrors the reference's objects, not its code:
//   PyDriver  ~ Clustering<int>::PitmanYor::CachedMixture (clustering.hpp:126-234)
//   Slave     ~ MixtureSlave<Model,...>                    (mixture.hpp:340-450)
//   Tracker   ~ MixtureIdTracker                           (mixture.hpp:460-521)
//   Gibbs     = the three together over a resident row table (extension)
// All numeric state is in HBM; the host keeps structure (group count, the
// empty-group set, id maps) and a mirror of the integer group sizes.
#include <algorithm>
#include <cstring>
#include <memory>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <set>

#include "common.h"
#include "comm.h"

#include "kernels.h"
#include "le_table.h"

namespace dist {

// sort.hip
size_t sort_pairs_temp_bytes(size_t n, int bits);
void sort_pairs(void * temp, size_t temp_bytes, const uint32_t * keys_in,
                uint32_t * keys_out, const uint32_t * vals_in,
                uint32_t * vals_out, size_t n, int bits, hipStream_t stream);

__device__ Tables g_tables_dev;
__device__ LgammaLut g_lgamma_lut[2];
__device__ int g_lgamma_cur;
static Tables g_tables_host_storage;
const Tables * g_tables_host = nullptr;

static thread_local std::string t_last_error;
void set_last_error(const std::string & what) { t_last_error = what; }

static std::mutex g_init_mutex;
static std::set<int> g_ready_devices;

static void fill_host_tables() {
    Tables & t = g_tables_host_storage;
    memcpy(t.log_table, DIST_REF_LOG_TABLE, sizeof(t.log_table));
    memcpy(t.exp_table, DIST_REF_EXP_TABLE, sizeof(t.exp_table));
    memcpy(t.lgamma_coeff5, DIST_REF_LGAMMA_COEFF5, sizeof(t.lgamma_coeff5));
    memcpy(t.lgamma_nu_coeff3, DIST_REF_LGAMMA_NU_COEFF3,
           sizeof(t.lgamma_nu_coeff3));
    memcpy(t.log_factorial, DIST_REF_LOG_FACTORIAL, sizeof(t.log_factorial));
    memcpy(t.exp_ab, DIST_REF_EXP_AB, sizeof(t.exp_ab));
    g_tables_host = &g_tables_host_storage;
}

static void ensure_host_tables() {
    std::lock_guard<std::mutex> lock(g_init_mutex);
    if (!g_tables_host) fill_host_tables();
}

void ensure_device_ready() {
    std::lock_guard<std::mutex> lock(g_init_mutex);
    if (!g_tables_host) fill_host_tables();
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess || n == 0)
        throw Error("no HIP device: libdistributions_hip has no CPU path");
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (g_ready_devices.count(dev)) return;
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        throw Error(std::string("built for gfx950, found ") + prop.gcnArchName);
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_tables_dev),
                                &g_tables_host_storage, sizeof(Tables)));
    HIP_CHECK(hipDeviceSynchronize());
    g_ready_devices.insert(dev);
}

// Every launch and copy of the calling thread goes to this stream: the
// default (null) stream unless dist_set_stream gave the thread its own, which
// lets independent engines -- one per host thread -- overlap on the device.
static thread_local hipStream_t t_stream = nullptr;
hipStream_t stream() { return t_stream; }

// the pinned ring behind DeviceBuf::upload (common.h): per host thread and
// device
namespace {
struct UploadRing {
    static constexpr size_t kBytes = 8u << 20, kHalf = kBytes / 2;
    static constexpr size_t kMaxUpload = 256u << 10;   // larger: as before
    char * base = nullptr;
    size_t head = 0;
    hipEvent_t left[2] = {nullptr, nullptr};   // recorded on leaving a half
    bool recorded[2] = {false, false};
    hipStream_t on = nullptr;   // the stream the recorded events are of
    ~UploadRing() {
        // (process teardown: the runtime may be gone; nothing to wait for)
    }
    const void * stage(const void * host, size_t bytes) {
        if (bytes > kMaxUpload) return host;
        if (!base) {
            if (hipHostMalloc((void **)&base, kBytes, hipHostMallocDefault)
                    != hipSuccess) {
                base = nullptr;
                return host;   // (no pinned memory: the synchronous copy)
            }
            for (int i = 0; i < 2; ++i)
                HIP_CHECK(hipEventCreateWithFlags(&left[i],
                                                  hipEventDisableTiming));
        }
        hipStream_t now = stream();
        if (now != on) {
            // copies of another stream may still read the ring: drain it.
            // By an event of its own recorded on that stream -- or, when the
            // caller has destroyed the stream meanwhile (dist_set_stream with
            // a temporary one: the handle is invalid, and whatever it queued
            // went with it), by the device -- never by a throw that would
            // leave `on` stale and fail every later upload of this thread.
            const hipStream_t before = on;
            on = now;
            if (recorded[0] || recorded[1] || head) {
                bool drained = hipEventRecord(left[0], before) == hipSuccess
                               && hipEventSynchronize(left[0]) == hipSuccess;
                if (!drained) {
                    (void)hipGetLastError();
                    drained = hipDeviceSynchronize() == hipSuccess;
                }
                if (!drained) {
                    (void)hipGetLastError();
                    return host;   // (the synchronous copy, this once)
                }
            }
            recorded[0] = recorded[1] = false;
            head = 0;
        }
        const size_t need = (bytes + 255) & ~(size_t)255;
        // the half written last, and what is left of it
        const int half_now = head == 0 ? 0 : (int)((head - 1) / kHalf);
        size_t at = head;
        if (head + need > (size_t)(half_now + 1) * kHalf) {
            const int half_next = half_now ^ 1;
            // leaving half_now: everything queued from it so far ...
            HIP_CHECK(hipEventRecord(left[half_now], now));
            recorded[half_now] = true;
            // ... entering half_next: its last lap's copies must be done
            if (recorded[half_next]) {
                HIP_CHECK(hipEventSynchronize(left[half_next]));
                recorded[half_next] = false;
            }
            at = (size_t)half_next * kHalf;
        }
        memcpy(base + at, host, bytes);
        head = at + need;
        return base + at;
    }
};
}  // namespace
const void * staged_for_upload(const void * host, size_t bytes) {
    // (a ring per device: its events belong to the device that was current
    // when they were made, and a thread may drive several)
    static thread_local UploadRing rings[16];
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 16)
        return host;
    return rings[device].stage(host, bytes);
}

// (y, glibc lgammaf(y)) pairs for the current device's table, sorted by y
// (y -> glibc lgammaf(y), users) per device.  lgammaf is a pure function, so
// an entry is right whoever put it there; `users` only decides what may be
// dropped when the table is full (entries of features that are gone).
struct LgammaEntry {
    float value;
    int users;
};
static std::map<int, std::map<float, LgammaEntry>> g_lgamma_registered;
static std::map<int, int> g_lgamma_current;   // per device: the copy in use

// (under g_init_mutex)  The new table goes into the copy no kernel reads:
// kernels that other host threads launch meanwhile keep searching the current
// one, which stays untouched until the flip below; whatever still ran on the
// spare copy from before the previous flip is drained first.
static void upload_lgamma_table(int dev,
                                const std::map<float, LgammaEntry> & table) {
    static LgammaLut staging;
    staging.n = 0;
    for (auto & kv : table) {
        staging.y[staging.n] = kv.first;
        staging.v[staging.n] = kv.second.value;
        staging.n += 1;
    }
    const int spare = 1 - g_lgamma_current[dev];
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_lgamma_lut), &staging,
                                sizeof(LgammaLut),
                                (size_t)spare * sizeof(LgammaLut)));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_lgamma_cur), &spare,
                                sizeof(int)));
    HIP_CHECK(hipDeviceSynchronize());
    g_lgamma_current[dev] = spare;
}
// Adds the arguments below 2.5 among `ys` to the device's table (see
// special.h: the reference evaluates libm's lgammaf there).  `hold`: the
// caller keeps them in use until release_small_lgamma(ys).
static void register_small_lgamma(const std::vector<float> & ys,
                                  bool hold = false) {
    std::lock_guard<std::mutex> lock(g_init_mutex);
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::map<float, LgammaEntry> & table = g_lgamma_registered[dev];
    bool changed = false;
    for (float y : ys) {
        if (!(y > 0.f && y < 2.5f)) continue;
        auto it = table.find(y);
        if (it == table.end()) {
            if ((int)table.size() >= kLgammaLutCap) {
                // full: entries nobody holds any more make room
                for (auto d = table.begin(); d != table.end();)
                    d = d->second.users == 0 ? table.erase(d) : std::next(d);
                changed = true;
                if ((int)table.size() >= kLgammaLutCap) continue;
            }
            it = table.emplace(y, LgammaEntry{::lgammaf(y), 0}).first;
            changed = true;
        }
        if (hold) it->second.users += 1;
    }
    if (changed) upload_lgamma_table(dev, table);
}
static void release_small_lgamma(const std::vector<float> & ys) {
    std::lock_guard<std::mutex> lock(g_init_mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    auto t = g_lgamma_registered.find(dev);
    if (t == g_lgamma_registered.end()) return;
    for (float y : ys) {
        auto it = t->second.find(y);
        if (it != t->second.end() && it->second.users > 0)
            it->second.users -= 1;
    }
}

// the arguments below 2.5 that MixtureDataScorer::score_data hands to
// fast_lgamma for groups of at most two members under `sh` (dd.hpp:287-318,
// bb.hpp:207-229, gp.hpp:220-241, nich.hpp:262-288, bnb.hpp:226-245)
static void score_data_small_args(const dist_shared_t & sh,
                                  std::vector<float> & ys) {
    switch (sh.kind) {
    case DIST_DD: {
        float alpha_sum = 0.f;
        for (int v = 0; v < sh.dim; ++v) {
            alpha_sum += sh.alphas[v];
            for (int c = 0; c <= 2; ++c) ys.push_back(sh.alphas[v] + (float)c);
        }
        for (int n = 0; n <= 2; ++n) ys.push_back(alpha_sum + (float)n);
        break;
    }
    case DIST_BB:
        ys.push_back(sh.p[0] + sh.p[1]);
        for (int h = 0; h <= 2; ++h)
            for (int t = 0; t <= 2; ++t) {
                const float a = sh.p[0] + (float)h, b = sh.p[1] + (float)t;
                ys.push_back(a);
                ys.push_back(b);
                ys.push_back(a + b);
            }
        break;
    case DIST_GP:
        for (int sum = 0; sum <= 2; ++sum) ys.push_back(sh.p[0] + (float)sum);
        break;
    case DIST_NICH:
        for (int n = 0; n <= 5; ++n) ys.push_back(0.5f * (sh.p[3] + (float)n));
        break;
    case DIST_BNB:
        ys.push_back(sh.p[0] + sh.p[1]);
        for (int n = 0; n <= 2; ++n)
            for (int sum = 0; sum <= 2; ++sum) {
                const float pa = sh.p[0] + sh.p[2] * (float)n;
                const float pb = sh.p[1] + (float)sum;
                ys.push_back(pa);
                ys.push_back(pb);
                ys.push_back(pa + pb);
            }
        break;
    default:
        break;
    }
}

static inline dim3 grid_for(size_t n, int block = kBlock) {
    return dim3((unsigned)std::max<size_t>(1, (n + block - 1) / block));
}
#define LAUNCH(kernel, n, ...)                                               \
    do {                                                                     \
        hipLaunchKernelGGL(kernel, grid_for(n), dim3(kBlock), 0, stream(),   \
                           __VA_ARGS__);                                     \
        HIP_CHECK(hipGetLastError());                                        \
    } while (0)
#define LAUNCH1(kernel, ...)                                                 \
    do {                                                                     \
        hipLaunchKernelGGL(kernel, dim3(1), dim3(1), 0, stream(),            \
                           __VA_ARGS__);                                     \
        HIP_CHECK(hipGetLastError());                                        \
    } while (0)

static void sync() { HIP_CHECK(hipStreamSynchronize(stream())); }

static size_t group_words(const dist_shared_t & sh) {
    return is_cat(sh.kind) ? 1 + (size_t)sh.dim
           : (sh.kind == DIST_BB || sh.kind == DIST_BNB ? 2 : 3);
}

static void check_shared(const dist_shared_t & sh) {
    DIST_REQUIRE(sh.kind >= DIST_DD && sh.kind <= DIST_BNB, "bad model kind");
    if (sh.kind == DIST_BNB)
        DIST_REQUIRE(sh.p[2] >= 1.f && sh.p[2] == (float)(uint32_t)sh.p[2],
                     "BetaNegativeBinomial: r must be a positive integer");
    if (sh.kind == DIST_DD)
        DIST_REQUIRE(sh.dim >= 1 && sh.dim <= DIST_DD_MAX_DIM,
                     "expected 1 <= dim <= 256");
    if (sh.kind == DIST_DPD)
        DIST_REQUIRE(sh.dim >= 1 && sh.betas != nullptr, "DPD needs betas");
}

// a reusable scratch of device floats for the per-call API paths
// DIST_HOST_PROFILE=1: where the host's time per batch goes (nanoseconds per
// named section, printed when the process ends).  A diagnostic; off, a probe is
// one load and a branch.
struct HostProf {
    static constexpr int kSlots = 24;
    const char * name[kSlots] = {nullptr};
    unsigned long long ns[kSlots] = {0}, hits[kSlots] = {0};
    bool on = getenv("DIST_HOST_PROFILE") != nullptr;
    // (the first calls of every section -- the ranges' one-time sorts, the
    // first allocations -- are left out: DIST_HOST_PROFILE=<calls to skip>)
    unsigned long long skip =
        getenv("DIST_HOST_PROFILE") ? strtoull(getenv("DIST_HOST_PROFILE"),
                                               nullptr, 10) : 0;
    unsigned long long seen[kSlots] = {0};
    ~HostProf() {
        if (!on) return;
        for (int i = 0; i < kSlots; ++i)
            if (name[i])
                fprintf(stderr, "[dist host] %-22s %10.3f ms  %8llu calls  %8.2f us each\n",
                        name[i], ns[i] * 1e-6, hits[i],
                        hits[i] ? ns[i] * 1e-3 / hits[i] : 0.0);
    }
};
static HostProf & host_prof() {
    static HostProf p;
    return p;
}
struct HostProbe {
    int slot;
    std::chrono::steady_clock::time_point t0;
    HostProbe(int s, const char * n) : slot(s) {
        HostProf & p = host_prof();
        if (!p.on) { slot = -1; return; }
        p.name[s] = n;
        t0 = std::chrono::steady_clock::now();
    }
    ~HostProbe() {
        if (slot < 0) return;
        HostProf & p = host_prof();
        if (p.seen[slot]++ < p.skip) return;
        p.ns[slot] += (unsigned long long)std::chrono::duration_cast<
            std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        p.hits[slot] += 1;
    }
};
#define HOST_PROBE(slot, name) HostProbe host_probe_##slot(slot, name)

struct Scratch {
    DeviceBuf<float> f;
    DeviceBuf<uint32_t> u;
    DeviceBuf<SampleOut> s;
};
static Scratch & scratch() {
    static thread_local Scratch sc;
    return sc;
}

// ---------------------------------------------------------------------------
// Slave: one feature's groups in HBM

struct Slave {
    dist_shared_t sh;
    std::vector<float> betas;
    float alpha_sum = 0.f;
    int K = 0;
    int cap = 0;
    DeviceBuf<int32_t> i0, i1, cnt;
    DeviceBuf<float> f0, f1, c0, c1, c2, c3, S, prior, other;
    float other_host = 0.f;
    float gp_lut[12] = {0};   // see gp_lgamma (models.h)
    std::vector<float> held_lgamma;   // see register_small_lgamma
    ~Slave() { release_small_lgamma(held_lgamma); }
    Slave(const Slave &) = delete;
    Slave & operator=(const Slave &) = delete;

    explicit Slave(const dist_shared_t & shared) : sh(shared) {
        check_shared(sh);
        ensure_device_ready();
        if (sh.kind == DIST_GP) {
            // every float (alpha + s) + x with s + x <= 2, and glibc's lgammaf
            // of it (the call the reference makes, special.hpp:121-123)
            int n = 0;
            for (int s = 0; s <= 2; ++s)
                for (int x = 0; s + x <= 2; ++x) {
                    const float y = (sh.p[0] + (float)s) + (float)x;
                    gp_lut[n] = y;
                    gp_lut[6 + n] = ::lgammaf(y);
                    n += 1;
                }
        }
        if (sh.kind == DIST_BNB) {
            // every argument below 2.5 that Scorer::init / eval can hand to
            // fast_lgamma (bnb.hpp:200-223): post.alpha = alpha + r*n,
            // post.beta = beta + s, their sum, post.alpha + r, post.beta + x
            // and its sum with (post.alpha + r), for n, s, x <= 2
            std::vector<float> ys;
            const float r = sh.p[2];
            for (int n = 0; n <= 2; ++n) {
                const float pa = sh.p[0] + r * (float)n;
                const float a2 = pa + r;
                ys.push_back(pa);
                ys.push_back(a2);
                for (int s = 0; s <= 2; ++s) {
                    const float pb = sh.p[1] + (float)s;
                    ys.push_back(pb);
                    ys.push_back(pa + pb);
                    for (int x = 0; x <= 2; ++x) {
                        const float b2 = pb + (float)x;
                        ys.push_back(b2);
                        ys.push_back(b2 + a2);
                    }
                }
            }
            ys.push_back(sh.p[0] + sh.p[1]);
            held_lgamma.insert(held_lgamma.end(), ys.begin(), ys.end());
        }
        if (sh.kind == DIST_NICH && sh.p[3] < 0.0625f) {
            // fast_lgamma_nu's libm branch (special.hpp:226-229), reached
            // by a group without members when nu itself is below 1/16
            held_lgamma.push_back((sh.p[3] + 1.0f) * 0.5f);
            held_lgamma.push_back(sh.p[3] * 0.5f);
        }
        // ... and what score_data reaches under this Shared; held in the
        // device's table for as long as this feature lives
        score_data_small_args(sh, held_lgamma);
        register_small_lgamma(held_lgamma, true);
        if (sh.kind == DIST_DD) {
            // dd.hpp:403-406: alpha_sum_ accumulates in index order
            alpha_sum = 0.f;
            for (int v = 0; v < sh.dim; ++v) alpha_sum += sh.alphas[v];
            prior.upload(sh.alphas, sh.dim);
        } else if (sh.kind == DIST_DPD) {
            betas.assign(sh.betas, sh.betas + sh.dim);
            sh.betas = betas.data();
            alpha_sum = sh.p[0];
            DeviceBuf<float> b;
            b.upload(betas.data(), sh.dim);
            prior.reserve(sh.dim, 0);
            LAUNCH(k_dpd_prior, sh.dim, sh.p[0], b.p, prior.p, sh.dim);
            other.reserve(1, 0);
            LAUNCH1(k_dpd_other, sh.p[0], sh.p[1], other.p);
            other.download(&other_host, 1);   // also drains the stream
        }
        sync();
    }

    int dim() const { return is_cat(sh.kind) ? sh.dim : 0; }

    SlaveView view() const {
        SlaveView v;
        v.kind = sh.kind;
        v.dim = sh.dim;
        for (int i = 0; i < 4; ++i) v.p[i] = sh.p[i];
        for (int i = 0; i < 12; ++i) v.p[4 + i] = gp_lut[i];
        v.alpha_sum = alpha_sum;
        v.other = other_host;
        v.K = K;
        v.cap = cap;
        v.i0 = i0.p; v.i1 = i1.p; v.f0 = f0.p; v.f1 = f1.p; v.cnt = cnt.p;
        v.c0 = c0.p; v.c1 = c1.p; v.c2 = c2.p; v.c3 = c3.p; v.S = S.p;
        v.prior = prior.p;
        return v;
    }

    void reserve(int need) {
        if (need <= cap) return;
        const int ncap = (int)grow_capacity((size_t)need);
        i0.reserve(ncap, K); i1.reserve(ncap, K);
        f0.reserve(ncap, K); f1.reserve(ncap, K);
        c0.reserve(ncap, K); c1.reserve(ncap, K);
        c2.reserve(ncap, K); c3.reserve(ncap, K);
        if (is_cat(sh.kind)) {
            cnt.reserve((size_t)ncap * sh.dim, (size_t)K * sh.dim);
            // S is [dim][cap]: re-pitch the rows
            DeviceBuf<float> nS;
            nS.reserve((size_t)ncap * sh.dim, 0);
            if (S.p && K)
                HIP_CHECK(hipMemcpy2DAsync(
                    nS.p, (size_t)ncap * sizeof(float), S.p,
                    (size_t)cap * sizeof(float), (size_t)K * sizeof(float),
                    sh.dim, hipMemcpyDeviceToDevice, stream()));
            sync();
            std::swap(S.p, nS.p);
            std::swap(S.cap, nS.cap);
        }
        cap = ncap;
    }

    void check_group(size_t g) const {
        DIST_REQUIRE(g < (size_t)K, "bad groupid: " + std::to_string(g));
    }
    void check_value(uint32_t value) const {
        if (sh.kind == DIST_DD)
            DIST_REQUIRE(value < (uint32_t)sh.dim,
                         "value out of bounds: " + std::to_string(value));
        if (sh.kind == DIST_DPD)
            DIST_REQUIRE(value < (uint32_t)sh.dim || value == DIST_DPD_OTHER,
                         "unknown value: " + std::to_string(value));
    }

    void clear() { K = 0; }

    void append_zero(int n) {
        reserve(K + n);
        const size_t width = is_cat(sh.kind) ? sh.dim : 1;
        SlaveView v = view();
        LAUNCH(k_slave_zero_groups, (size_t)n * width, v, K, K + n);
        K += n;
    }

    // groups().push_back(group): statistics only; init() builds the caches
    void append(const uint32_t * g) {
        reserve(K + 1);
        const int k = K;
        int32_t a = (int32_t)g[0], b = 0;
        float x = 0.f, y = 0.f;
        switch (sh.kind) {
        case DIST_DD:
        case DIST_DPD:
            HIP_CHECK(hipMemcpyAsync(cnt.p + (size_t)k * sh.dim, g + 1,
                                     sizeof(int32_t) * sh.dim,
                                     hipMemcpyHostToDevice, stream()));
            break;
        case DIST_BB:
        case DIST_BNB:
            b = (int32_t)g[1];
            break;
        case DIST_GP:
            b = (int32_t)g[1];
            memcpy(&x, &g[2], 4);
            break;
        default:
            memcpy(&x, &g[1], 4);
            memcpy(&y, &g[2], 4);
            break;
        }
        HIP_CHECK(hipMemcpyAsync(i0.p + k, &a, 4, hipMemcpyHostToDevice, stream()));
        HIP_CHECK(hipMemcpyAsync(i1.p + k, &b, 4, hipMemcpyHostToDevice, stream()));
        HIP_CHECK(hipMemcpyAsync(f0.p + k, &x, 4, hipMemcpyHostToDevice, stream()));
        HIP_CHECK(hipMemcpyAsync(f1.p + k, &y, 4, hipMemcpyHostToDevice, stream()));
        sync();
        K += 1;
    }

    void get_group(size_t g, uint32_t * out) const {
        check_group(g);
        int32_t a, b;
        float x, y;
        HIP_CHECK(hipMemcpyAsync(&a, i0.p + g, 4, hipMemcpyDeviceToHost, stream()));
        HIP_CHECK(hipMemcpyAsync(&b, i1.p + g, 4, hipMemcpyDeviceToHost, stream()));
        HIP_CHECK(hipMemcpyAsync(&x, f0.p + g, 4, hipMemcpyDeviceToHost, stream()));
        HIP_CHECK(hipMemcpyAsync(&y, f1.p + g, 4, hipMemcpyDeviceToHost, stream()));
        HIP_CHECK(hipStreamSynchronize(stream()));
        out[0] = (uint32_t)a;
        switch (sh.kind) {
        case DIST_DD:
        case DIST_DPD:
            HIP_CHECK(hipMemcpyAsync(out + 1, cnt.p + g * sh.dim,
                                     sizeof(int32_t) * sh.dim,
                                     hipMemcpyDeviceToHost, stream()));
            HIP_CHECK(hipStreamSynchronize(stream()));
            break;
        case DIST_BB:
        case DIST_BNB:
            out[1] = (uint32_t)b;
            break;
        case DIST_GP:
            out[1] = (uint32_t)b;
            memcpy(&out[2], &x, 4);
            break;
        default:
            memcpy(&out[1], &x, 4);
            memcpy(&out[2], &y, 4);
            break;
        }
    }

    // MixtureSlave::validate (mixture.hpp:440-444): the groups and both
    // scorers agree on the group count (dd.hpp:447-455 ...), and -- what those
    // size checks stand for -- the value scorer's cache IS Scorer::init of
    // the statistics: recomputed in place and compared bit for bit.
    void validate() {
        DIST_REQUIRE(K >= 0 && K <= cap, "validate: group count beyond capacity");
        const size_t width = is_cat(sh.kind) ? (size_t)sh.dim : 0;
        auto grab = [&](const DeviceBuf<float> & b, size_t n) {
            std::vector<float> v(b.p ? n : 0);
            if (b.p && n) b.download(v.data(), n);
            return v;
        };
        const size_t Kn = (size_t)K;
        std::vector<float> before[5] = {grab(c0, Kn), grab(c1, Kn), grab(c2, Kn),
                                        grab(c3, Kn), grab(S, width * cap)};
        std::vector<int32_t> sizes(Kn);
        if (Kn) i0.download(sizes.data(), Kn);
        if (is_cat(sh.kind) && Kn) {
            std::vector<int32_t> cells(Kn * width);
            cnt.download(cells.data(), cells.size());
            for (size_t k = 0; k < Kn; ++k) {
                long long sum = 0;
                for (size_t v = 0; v < width; ++v) sum += cells[k * width + v];
                DIST_REQUIRE(sum == sizes[k],
                             "validate: count_sum != sum of counts in group "
                                 + std::to_string(k));
            }
        }
        init();
        std::vector<float> after[5] = {grab(c0, Kn), grab(c1, Kn), grab(c2, Kn),
                                       grab(c3, Kn), grab(S, width * cap)};
        static const char * names[5] = {"c0", "c1", "c2", "c3", "S"};
        DeviceBuf<float> * const bufs[5] = {&c0, &c1, &c2, &c3, &S};
        for (int b = 0; b < 5; ++b)
            for (size_t i = 0; i < before[b].size(); ++i) {
                if (b == 4 && i % cap >= Kn) continue;   // slots beyond K
                uint32_t x, y;
                memcpy(&x, &before[b][i], 4);
                memcpy(&y, &after[b][i], 4);
                if (x == y) continue;
                // (a check must not repair what it reports: the cache goes
                // back to what it was, so that the failure can be seen again)
                for (int r = 0; r < 5; ++r)
                    if (!before[r].empty())
                        bufs[r]->upload(before[r].data(), before[r].size());
                dist::sync();
                throw Error(std::string("ERROR validate: the value scorer's "
                                        "cache ")
                            + names[b] + " is stale at "
                            + std::to_string(b == 4 ? i % cap : i));
            }
    }

    void update(int k0, int k1) {
        if (k1 <= k0) return;
        const size_t width = is_cat(sh.kind) ? sh.dim : 1;
        SlaveView v = view();
        LAUNCH(k_slave_update, (size_t)(k1 - k0) * width, v, k0, k1);
    }
    void init() { update(0, K); }            // mixture.hpp:354-359

    void add_group() {                        // mixture.hpp:361-368
        append_zero(1);
        update(K - 1, K);
    }
    void remove_group(size_t g) {             // mixture.hpp:370-375
        check_group(g);
        const int last = K - 1;
        if ((int)g != last) {
            SlaveView v = view();
            LAUNCH(k_slave_move_group, (size_t)std::max(1, dim()), v, (int)g,
                   last);
        }
        K = last;
    }
    void value_op(size_t g, uint32_t value, int add) {
        check_group(g);
        check_value(value);
        DIST_REQUIRE(!(sh.kind == DIST_DPD && value == DIST_DPD_OTHER),
                     "cannot add or remove OTHER");
        SlaveView v = view();
        LAUNCH1(k_slave_value_op, v, (int)g, value, add);
    }
    float score_value_group(size_t g, uint32_t value) const {
        check_group(g);
        check_value(value);
        Scratch & sc = scratch();
        sc.f.reserve(1, 0);
        SlaveView v = view();
        LAUNCH1(k_slave_score_group, v, (int)g, value, sc.f.p);
        float out;
        sc.f.download(&out, 1);
        return out;
    }
    float score_data() const {                // mixture.hpp:427-431
        float out = 0.f;
        score_data_grid(&sh, 1, &out);
        return out;
    }
    // mixture.hpp:433-438 over MixtureSlaveDataScorerMixin::score_data_grid
    // (mixture.hpp:238-247) / DirichletDiscrete's incremental form
    // (dd.hpp:259-284, whose alpha_sum is carried in binary64 from one
    // candidate to the next)
    void score_data_grid(const dist_shared_t * shareds, size_t n,
                         float * scores_out) const {
        if (!n) return;
        const size_t dimc = is_cat(sh.kind) ? (size_t)sh.dim : 0;
        std::vector<float> cp(4 * n), cprior(std::max<size_t>(1, dimc * n)),
            csum(n, 0.f);
        double alpha_sum_d = 0.0;
        for (size_t c = 0; c < n; ++c) {
            const dist_shared_t & cs = shareds[c];
            DIST_REQUIRE(cs.kind == sh.kind, "score_data_grid: model mismatch");
            DIST_REQUIRE(!dimc || cs.dim == sh.dim,
                         "score_data_grid: dim mismatch");
            check_shared(cs);
            for (int j = 0; j < 4; ++j) cp[4 * c + j] = cs.p[j];
            if (sh.kind == DIST_DD) {
                if (c == 0) {
                    float a = 0.f;   // _init: float, index order
                    for (int v = 0; v < sh.dim; ++v) a += cs.alphas[v];
                    alpha_sum_d = a;
                } else {             // _update: only the changed entries
                    for (int v = 0; v < sh.dim; ++v)
                        if (cs.alphas[v] != shareds[c - 1].alphas[v])
                            alpha_sum_d += (double)cs.alphas[v]
                                         - (double)shareds[c - 1].alphas[v];
                }
                csum[c] = (float)alpha_sum_d;
                for (int v = 0; v < sh.dim; ++v)
                    cprior[c * dimc + v] = cs.alphas[v];
            } else if (sh.kind == DIST_DPD) {
                csum[c] = cs.p[0];
                for (int v = 0; v < sh.dim; ++v)   // dpd.hpp:362
                    cprior[c * dimc + v] = cs.betas[v] * cs.p[0];
            }
        }
        if (!K) {
            for (size_t c = 0; c < n; ++c) scores_out[c] = 0.f;
            return;
        }
        // glibc's lgammaf for the small arguments these candidates reach,
        // held for the duration of the call
        std::vector<float> grid_args;
        for (size_t c = 0; c < n; ++c)
            score_data_small_args(shareds[c], grid_args);
        register_small_lgamma(grid_args, true);
        struct Release {
            const std::vector<float> & ys;
            ~Release() { release_small_lgamma(ys); }
        } release_at_exit{grid_args};
        DeviceBuf<float> dp, dprior, dsum;
        dp.upload(cp.data(), cp.size());
        dprior.upload(cprior.data(), cprior.size());
        dsum.upload(csum.data(), csum.size());
        SlaveView v = view();
        if (sh.kind == DIST_DPD) {
            // sparse-counter iteration order is not defined in the reference
            // (dpd.hpp:344-374): terms summed in binary64, 1e-5 relative
            DeviceBuf<double> out;
            out.reserve(n, 0);   // zero-filled
            const size_t cells = (size_t)K * dimc;
            hipLaunchKernelGGL(k_score_data_grid,
                               dim3((unsigned)((cells + kBlock - 1) / kBlock),
                                    (unsigned)n),
                               dim3(kBlock), 0, stream(), v, dp.p, dprior.p,
                               dsum.p, out.p);
            HIP_CHECK(hipGetLastError());
            std::vector<double> totals(n);
            out.download(totals.data(), n);
            for (size_t c = 0; c < n; ++c) scores_out[c] = (float)totals[c];
            return;
        }
        // the reference's float accumulation order, bit for bit
        DeviceBuf<float> out;
        out.reserve(n, 0);
        if (sh.kind == DIST_DD) {
            hipLaunchKernelGGL(k_score_data_dd, dim3((unsigned)n), dim3(512), 0,
                               stream(), v, dprior.p, dsum.p, out.p);
            HIP_CHECK(hipGetLastError());
        } else {
            DeviceBuf<float> terms;
            terms.reserve(n * (size_t)K * 4, 0);
            hipLaunchKernelGGL(k_score_data_terms,
                               dim3((unsigned)((K + kBlock - 1) / kBlock),
                                    (unsigned)n),
                               dim3(kBlock), 0, stream(), v, dp.p, terms.p);
            HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL(k_score_data_serial, dim3((unsigned)n), dim3(64),
                               0, stream(), terms.p, (size_t)K * 4, out.p);
            HIP_CHECK(hipGetLastError());
        }
        out.download(scores_out, n);
    }
    // score_value for n values at once: acc[r * ld + k] accumulates
    void score_values(const uint32_t * vals, size_t n, float * acc,
                      size_t ld) const {
        DIST_REQUIRE(ld >= (size_t)K, "row stride below len(mixture)");
        for (size_t r = 0; r < n; ++r) check_value(vals[r]);
        if (!K || !n) return;
        Scratch & sc = scratch();
        sc.u.upload(vals, n);
        sc.f.upload(acc, n * ld);
        SlaveView v = view();
        LAUNCH(k_slave_score_values, n * (size_t)K, v, sc.u.p, n, sc.f.p, ld, K);
        sc.f.download(acc, n * ld);
    }
    void score_value(uint32_t value, float * acc, size_t size) const {
        DIST_REQUIRE(size == (size_t)K, "scores_accum != len(mixture)");
        check_value(value);
        if (!K) return;
        Scratch & sc = scratch();
        sc.f.upload(acc, K);
        SlaveView v = view();
        LAUNCH(k_slave_score_value, (size_t)K, v, value, sc.f.p, K);
        sc.f.download(acc, K);
    }
};

// ---------------------------------------------------------------------------
// PyDriver: PitmanYor::CachedMixture over MixtureDriver

struct PyDriver {
    std::vector<int> counts;   // host mirror of the device counts
    int n_empty = 0;
    long long sample_size = 0;
    DeviceBuf<int32_t> d_counts;
    DeviceBuf<float> d_shifted;

    int K() const { return (int)counts.size(); }

    void reserve(int need) {
        if ((size_t)need <= d_counts.cap) return;
        const size_t ncap = grow_capacity((size_t)need);
        d_counts.reserve(ncap, counts.size());
        d_shifted.reserve(ncap, counts.size());
    }
    void rebuild(float alpha, float d) {      // clustering.hpp:151-161
        n_empty = 0;
        sample_size = 0;
        for (int c : counts) {
            sample_size += c;
            n_empty += (c == 0);
        }
        if (!counts.empty())
            LAUNCH(k_py_rebuild, counts.size(), d_counts.p, d_shifted.p, K(),
                   alpha, d, K() - n_empty, n_empty);
    }
    void init(float alpha, float d, const int * c, size_t n) {
        ensure_device_ready();
        for (size_t i = 0; i < n; ++i)
            DIST_REQUIRE(c[i] >= 0, "negative group size");
        counts.assign(c, c + n);
        reserve((int)n);
        d_counts.upload(counts.data(), n);
        rebuild(alpha, d);
        DIST_REQUIRE(n_empty > 0, "missing empty groups");  // mixture.hpp:153
    }
    void update_empties(float alpha, float d) {   // clustering.hpp:221-230
        LAUNCH(k_py_rebuild, counts.size(), d_counts.p, d_shifted.p, K(),
               alpha, d, K() - n_empty, n_empty);
    }
    bool add_value(float alpha, float d, size_t g) {   // clustering.hpp:163-176
        DIST_REQUIRE(g < counts.size(), "bad groupid: " + std::to_string(g));
        const bool add_group = (counts[g] == 0);
        counts[g] += 1;
        sample_size += 1;
        if (add_group) {
            // mixture.hpp:84-89: the filled group leaves the empty set and a
            // fresh empty group is appended, so the set's size is unchanged
            counts.push_back(0);
            reserve(K());
            LAUNCH1(k_py_set_count, d_counts.p, d_shifted.p, K() - 1, 0, d);
            LAUNCH1(k_py_set_count, d_counts.p, d_shifted.p, (int)g, counts[g], d);
            update_empties(alpha, d);
        } else {
            LAUNCH1(k_py_set_count, d_counts.p, d_shifted.p, (int)g, counts[g], d);
        }
        return add_group;
    }
    bool remove_value(float alpha, float d, size_t g) {  // clustering.hpp:178-193
        DIST_REQUIRE(g < counts.size(), "bad groupid: " + std::to_string(g));
        DIST_REQUIRE(counts[g] > 0, "cannot remove value from empty group");
        counts[g] -= 1;
        sample_size -= 1;
        const bool remove_group = (counts[g] == 0);
        if (remove_group) {
            const int last = K() - 1;                   // mixture.hpp:108-119
            if ((int)g != last) {
                counts[g] = counts[last];
                LAUNCH1(k_py_move, d_counts.p, d_shifted.p, (int)g, last);
            }
            counts.pop_back();
            update_empties(alpha, d);
        } else {
            LAUNCH1(k_py_set_count, d_counts.p, d_shifted.p, (int)g, counts[g], d);
        }
        return remove_group;
    }
    void score_value(float alpha, float * scores, size_t size) const {
        DIST_REQUIRE(size == counts.size(), "scores.size() != counts().size()");
        Scratch & sc = scratch();
        sc.f.reserve(size, 0);
        LAUNCH(k_py_score, size, d_shifted.p, sc.f.p, K(), sample_size, alpha);
        sc.f.download(scores, size);
    }
};

// ---------------------------------------------------------------------------
// Tracker: MixtureIdTracker (host; mirrors the swap-remove of the groups)

struct Tracker {
    std::vector<uint32_t> p2g;
    std::vector<int32_t> g2p;   // -1 = retired
    // how often the host changed what a packed index means (a group
    // swap-removed, the set rebuilt): see Gibbs::run_epoch
    uint64_t repacked = 0;

    void init(size_t n) {
        repacked += 1;
        p2g.clear();
        g2p.clear();
        for (size_t i = 0; i < n; ++i) add_group();
    }
    void add_group() {
        const uint32_t packed = (uint32_t)p2g.size();
        const uint32_t global = (uint32_t)g2p.size();
        p2g.push_back(global);
        g2p.push_back((int32_t)packed);
    }
    void remove_group(uint32_t packed) {
        DIST_REQUIRE(packed < p2g.size(),
                     "bad packed id: " + std::to_string(packed));
        repacked += 1;
        g2p[p2g[packed]] = -1;
        p2g[packed] = p2g.back();
        p2g.pop_back();
        if (packed != p2g.size()) g2p[p2g[packed]] = (int32_t)packed;
    }
    uint32_t packed_to_global(uint32_t packed) const {
        DIST_REQUIRE(packed < p2g.size(),
                     "bad packed id: " + std::to_string(packed));
        return p2g[packed];
    }
    uint32_t global_to_packed(uint32_t global) const {
        DIST_REQUIRE(global < g2p.size(),
                     "bad global id: " + std::to_string(global));
        DIST_REQUIRE(g2p[global] >= 0,
                     "stale global id: " + std::to_string(global));
        return (uint32_t)g2p[global];
    }
};

// ---------------------------------------------------------------------------
// Gibbs: the batched row engine

struct Gibbs {
    float alpha, d;
    int cluster = 0;        // 0 PitmanYor(alpha, d), 1 LowEntropy(dataset_size)
    int dataset_size = 0;
    PyDriver py;
    std::vector<std::unique_ptr<Slave>> feats;
    Tracker tracker;

    size_t n_rows = 0;
    uint64_t row_offset = 0;
    std::vector<DeviceBuf<uint32_t>> own_values;
    std::vector<const uint32_t *> values;   // device pointers
    DeviceBuf<uint32_t> own_assign;
    uint32_t * assign = nullptr;             // device, global ids
    DeviceBuf<uint32_t> d_maps;             // see upload_maps
    std::vector<uint32_t> maps_host;
    const uint32_t * d_p2g_ptr = nullptr;
    const int32_t * d_g2p_ptr = nullptr;
    size_t maps_pcap = 0;                   // slots reserved for p2g
    bool maps_dirty = true;

    DeviceBuf<uint32_t> old_packed, new_packed;
    DeviceBuf<uint32_t> old_row, new_row;   // row-ordered copies (see replay)
    DeviceBuf<int2> struct_moves;           // {dst, src} slot copies
    // ordered replay of float statistics: events sorted stably by group
    DeviceBuf<uint32_t> ev_keys, ev_vals, ev_keys_sorted, ev_vals_sorted;
    DeviceBuf<uint32_t> cs_hist, cs_total, cs_base;   // the counting sort's
    DeviceBuf<uint32_t> seg_begin;   // [begin | end] of every group's events
    DeviceBuf<unsigned char> sort_temp;
    DeviceBuf<uint32_t> pow_lo, pow_hi;   // 16807^i, 16807^(4096 i) mod 2^31-1
    uint32_t n_pow_hi = 0;
    DeviceBuf<float> base, base_single;
    std::vector<DeviceBuf<float>> ktab;     // per feature, see SweepParams
    std::vector<uint32_t> max_value;        // per feature (discrete kinds)
    DeviceBuf<SweepScalars> scalars;
    DeviceBuf<float> row_scores;
    DeviceBuf<int> row_size;

    size_t batch_begin = 0, batch_end = 0;   // rows of the open batch
    uint32_t batch_seed = 0;                 // ... and its entropy
    uint64_t batch_draw_base = 0;
    bool batch_open = false;
    bool batch_value_sorted = false;
    bool moves_in_row_order = false;   // old_row/new_row hold the open batch
    bool base_valid = false;           // base[], base_single[], scalars current
    bool cells_fresh = false;          // set by apply_ints, used by batch_finish
    bool timing_pending = false;
    int * pinned_counts = nullptr;
    size_t pinned_cap = 0;
    unsigned * pinned_seq = nullptr;   // see k_publish_counts
    unsigned publish_ticket = 0;
    // (size, ticket) pairs written by k_vs_reduce; pairs_ticket != 0: the
    // sizes of the open batch are on their way there
    unsigned long long * pinned_pairs = nullptr;
    size_t pinned_pairs_cap = 0;
    unsigned pairs_ticket = 0;

    // value-sorted path (single small-domain feature): rows of a batch range
    // sorted by value once, tiles of <= 64 equal-valued rows
    struct VsCache {
        size_t r0 = 0, r1 = 0;
        DeviceBuf<uint32_t> sorted_rows;
        DeviceBuf<VsTile> tiles;
        uint32_t n_tiles = 0;
        DeviceBuf<VsTile> narrow_tiles;    // <= 64 rows each (k_vs_narrow)
        uint32_t n_narrow_tiles = 0;
        DeviceBuf<VsTile> chunks;          // apply work items, one value each
        uint32_t n_chunks = 0;
        bool one_chunk_per_value = false;  // every value's rows in ONE chunk
        bool mixed_chunks = false;         // some chunks hold several values
        // the chunks of the values inside the tables come first, each of one
        // value, kVsApplyRows rows apart within a value (bands per chunk)
        bool one_value_chunks = false;
        uint32_t n_table_chunks = 0;
        DeviceBuf<uint32_t> val_start;    // [nvals + 1] first position per value
        DeviceBuf<uint32_t> chunk_first;  // [nvals + 1] first chunk per value
        // {value, first chunk, chunks} of every table value with several
        DeviceBuf<uint32_t> multi;
        uint32_t n_multi = 0;
        // rows handed over per chunk (VsDefer): zero between batches
        DeviceBuf<uint32_t> def_counts;
        // where each group's rows begin in each chunk after its last sort
        // (VsOffsets), the stamps that say whether that still holds
        DeviceBuf<int> grp_off;
        DeviceBuf<uint32_t> off_epoch;
        int off_stride = 0;
        DeviceBuf<uint32_t> other_pos;    // positions the tiles do not cover
        uint32_t n_other = 0;
        uint32_t n_values_present = 0;    // values with at least one row
        // the rows' current assignment (global ids) in sorted-position
        // order; while `dirty`, assign[] (row order) is stale for this range
        DeviceBuf<uint32_t> assign_pos;
        bool dirty = false;
    };
    // value-sorted batches keep assignments by position; anything that reads
    // assign[] in row order calls this first
    void flush_assign_pos() {
        for (auto & c : vs_cache) {
            if (!c->dirty) continue;
            const size_t n = c->r1 - c->r0;
            LAUNCH(k_pos_scatter, n, c->assign_pos.p, c->sorted_rows.p,
                   assign + c->r0, n);
            c->dirty = false;
        }
    }
    // A cached range's position-ordered assignments go stale when rows of the
    // range are re-assigned through any other path (another batch tiling, the
    // sequential chain): such caches are written back and dropped.
    void drop_overlapping_caches(size_t r0, size_t r1, bool keep_exact) {
        for (size_t i = 0; i < vs_cache.size();) {
            // (the compact copy of the ranges: a pass at 65 536 rows per batch
            // has 153 of them, and walking the objects themselves cost the
            // host 15 us per look)
            const bool overlap = vs_ranges[i].first < r1 && r0 < vs_ranges[i].second;
            const bool exact = vs_ranges[i].first == r0 && vs_ranges[i].second == r1;
            if (!overlap || (exact && keep_exact)) { ++i; continue; }
            VsCache & c = *vs_cache[i];
            if (c.dirty) {
                const size_t n = c.r1 - c.r0;
                LAUNCH(k_pos_scatter, n, c.assign_pos.p, c.sorted_rows.p,
                       assign + c.r0, n);
            }
            sync();   // the cache's buffers are freed below
            vs_cache.erase(vs_cache.begin() + (long)i);
            vs_ranges.erase(vs_ranges.begin() + (long)i);
            vs_last = 0;
        }
    }
    std::vector<std::unique_ptr<VsCache>> vs_cache;
    std::vector<std::pair<size_t, size_t>> vs_ranges;   // [r0, r1) of each
    size_t vs_last = 0;   // where the last look-up found its range
    DeviceBuf<float> vsLA, vsLB, vsM, vsmB, vsPA, vsPB, vsOwn;
    DeviceBuf<int> remap_log;   // the batches' swap-removals (kernels.h)
    // the removal epoch (DevState::pad) the groups' recorded offsets are
    // stamped with (VsOffsets): carried from one device-normalised run to the
    // next as long as the host did not re-pack the groups in between
    uint32_t run_epoch = 0;
    uint64_t repacked_seen = ~0ull;
    DeviceBuf<int32_t> vs_stage;   // [chunks][K] deltas of the open batch
    DeviceBuf<int> vsBandMode;     // VsTables::band_mode
    DeviceBuf<VsTile> vsBandTile;  // VsTables::band_tile
    DeviceBuf<unsigned long long> vsStamps;   // diagnostics, see VsTables
    DeviceBuf<float> vsScratch;    // k_vs_stream: [tiles][K] likelihoods
    int stream_scratch_mode = 1;   // 0: recompute them in the scan instead
    DeviceBuf<ChainResult> chain_result;
    DeviceBuf<int32_t> delta_image;         // dist_gibbs_sweep_sharded
    // (what its header words hold: sharded_header's tag, 0 = nothing)
    uint64_t delta_header_tag = 0;
    unsigned * comm_fault = nullptr;        // pinned: raised by k_add_words
    uint64_t comm_words_total = 0, comm_collectives = 0, comm_words_last = 0,
             comm_words_max = 0;
    // dist_gibbs_partition_by_value: this rank's rows carry values no other
    // rank's rows do, so the cells cnt[.][x] of its values are its own and
    // never travel; cells_partial: the OTHER values' cells here are stale
    // (dist_gibbs_gather_cells makes the replicas whole again)
    bool value_partitioned = false, cells_partial = false;
    size_t present_values = 0;              // values with rows on this rank
    DeviceBuf<int32_t> owned_values;        // [dim] 1: this rank's
    void require_whole(const char * what) const {
        DIST_REQUIRE(!cells_partial,
                     std::string(what) + ": the cells of other ranks' values "
                     "are stale on a value-partitioned rank "
                     "(dist_gibbs_gather_cells first)");
    }
    DeviceBuf<float> own_score;             // k_row_prepass
    // general rows: 3 = k_rows_scratch (default), 0 = k_sweep_program, the
    // kernel that stays for feature lists k_rows_scratch's table does not
    // hold (more than 64 parameter slots), kept selectable so that the tests
    // reach it
    int rows_scratch_mode = 3;
    int rows_scratch_lds_log = 1;     // FastLog's table in LDS where NICH scores
    int rows_scratch_block = 512;     // threads per workgroup
    // 1: every batch the value-sorted kernels do not take goes through the
    // score program when its tables exist (0: only feature lists without a
    // compile-time instance of k_sweep_sample, as in round 2)
    int program_all = 1;
    DeviceBuf<float> rows_gtab;             // [W][Kpad], see k_rows_gtab
    DeviceBuf<float2> rows_snap;            // k_rows_scratch, scan mode
    // 0 exact (the reference's float operations in the reference's order:
    // bit-identical assignments; the default and the line of record);
    // 1 scan: tolerance-level sampling (same scores, same draw per row, the
    // softmax and its inverse CDF by parallel-friendly float arithmetic)
    int sampling_mode = 0;
    // folding (kernels.h, FoldSpec): 0 off, 1 where the joint domain of the
    // leading discrete features leaves >= kFoldRowsPerCode rows per value
    int rows_fold_mode = 1;
    static constexpr size_t kFoldRowsPerCode = 128;
    struct FoldCache {
        size_t r0 = 0, r1 = 0;
        int n_fold = 0;
        uint32_t J = 0;
        DeviceBuf<uint32_t> sorted_rows;
        DeviceBuf<uint4> tiles;
        uint32_t n_tiles = 0;
    };
    std::vector<std::unique_ptr<FoldCache>> fold_cache;
    DeviceBuf<float> rows_fold;             // [J][Kpad]
    uint64_t fold_batches = 0;
    FoldCache & fold_get(size_t r0, size_t r1, const FoldSpec & F, uint32_t J) {
        for (auto & c : fold_cache)
            if (c->r0 == r0 && c->r1 == r1 && c->n_fold == F.n && c->J == J)
                return *c;
        std::unique_ptr<FoldCache> c(new FoldCache());
        c->r0 = r0; c->r1 = r1; c->n_fold = F.n; c->J = J;
        const size_t n = r1 - r0;
        DeviceBuf<uint32_t> codes, index, codes_sorted;
        codes.reserve(n, 0); index.reserve(n, 0); codes_sorted.reserve(n, 0);
        c->sorted_rows.reserve(n, 0);
        LAUNCH(k_fold_codes, n, F, r0, n, J, codes.p, index.p);
        int bits = 1;
        while ((1ull << bits) < (unsigned long long)J + 1) bits += 1;
        const size_t tb = sort_pairs_temp_bytes(n, bits);
        DeviceBuf<unsigned char> temp;
        temp.reserve(tb + 256, 0);
        sort_pairs(temp.p, tb, codes.p, codes_sorted.p, index.p,
                   c->sorted_rows.p, n, bits, stream());
        c->tiles.reserve(n / 64 + (size_t)J + 2, 0);
        DeviceBuf<uint32_t> count;
        count.reserve(1, 0);   // zero-filled
        LAUNCH(k_fold_tiles, n, codes_sorted.p, n, c->tiles.p, count.p);
        count.download(&c->n_tiles, 1);   // (also drains the stream)
        // (room for every range of a pass at this batch size: a pass that
        // evicts its own ranges would sort and host-sync on every batch)
        const size_t keep = std::min<size_t>(
            std::max<size_t>(64, (n_rows + n - 1) / std::max<size_t>(n, 1) + 2),
            4096);
        if (fold_cache.size() >= keep) fold_cache.erase(fold_cache.begin());
        fold_cache.push_back(std::move(c));
        return *fold_cache.back();
    }
    uint64_t scratch_batches = 0;
    // 2: k_chains (structural steps on the device); 1: round 3's kernel, back
    // to the host at every structural step; 0: every row as a batch of one
    int sequential_mode = 2;
    DeviceBuf<int> vsArg;
    DeviceBuf<uint32_t> deferred, deferred_count;
    int value_sorted_mode = 1;   // 0 off, 1 auto, 2 always (when eligible)
    // the table-free form of the value-sorted kernel (k_vs_stream): 0 never,
    // 1 where a value has about a tile per batch, 2 always
    int value_stream_mode = 1;
    uint64_t stream_batches = 0;
    // launches too small to fill the chip take tiles of 64 rows, one per lane,
    // and their vectors from LDS (k_vs_narrow): 0 never, 1 below
    // kVsNarrowBelowTiles regular tiles, 2 whenever the vectors fit
    int narrow_mode = 1;
    int narrow_read_ahead = 0;   // float4s per vector: 0 auto, 4 or 8
    // Wave priorities by phase (k_vs_sample, k_vs_stream, k_rows_scratch):
    // 0x10000 | set-up << 12 | first pass << 8 | ... | last pass.  A SIMD
    // issues from its oldest ready wave, so at equal priority its waves
    // finish one after the other and the last ones run their dependent adds
    // alone; with the earlier phase ahead of the later one whoever is behind
    // goes first and they end together (profiles/r5_wave_priorities.txt).
    int sample_prio_mode = 0x13210;   // VsTables::prio_mode, k_vs_stream
    int rows_prio_mode = 0x13210;     // RowsArgs::pad
    int apply_overlap_mode = 1;       // k_vs_apply: hand-overs beside the adds
    uint64_t narrow_batches = 0;
    // (measured, round 4, K = 1024, rows per launch: 400 k 49.5 us against
    // k_vs_sample's 51.3, 524 k 61.2 / 68.5, 655 k 66.7 / 69.0, 786 k 77.1 /
    // 68.9, 10^6 93.8 / 67.6; Zipf values 524 k 63.4 / 101.9, 786 k 77.7 /
    // 92.1, 10^6 97.5 / 88.2.  At larger K the vectors leave fewer waves a
    // place in LDS: round 2's bound stays there.)
    static constexpr uint32_t kVsNarrowBelowTiles = 5600;
    bool use_narrow(const VsCache & c, int Kpad) const {
        if (narrow_mode == 0 || Kpad > kVsNarrowMaxK || !c.n_narrow_tiles)
            return false;
        return narrow_mode == 2
               || c.n_tiles < (Kpad <= 1152 ? kVsNarrowBelowTiles : 4100u);
    }
    int running_sums_min_tiles = 2048;   // see sample_value_sorted
    int cu_count_cached = 0;
    int cu_count() {
        if (!cu_count_cached) {
            int dev = 0;
            HIP_CHECK(hipGetDevice(&dev));
            HIP_CHECK(hipDeviceGetAttribute(
                &cu_count_cached, hipDeviceAttributeMultiprocessorCount, dev));
        }
        return cu_count_cached;
    }
    uint64_t vs_batches = 0, generic_batches = 0;
    // diagnostics (dist_gibbs_debug_counts): launches that had band tiles /
    // running sums switched on, and whether the last one did
    uint64_t band_batches = 0, prefix_batches = 0;
    bool last_bands = false, last_prefix = false;

    // Device-side normalisation of the group set (k_normalise): a whole sweep
    // is queued without the host looking at the group sizes in between.
    // While `async_active`, K() is an UPPER BOUND (the host sizes launches and
    // buffers with it; the kernels read the group count from dev_state) and
    // the host mirrors (py.counts, tracker) are stale; sweep_async pulls the
    // state back before it returns.
    // (two slots: k_vs_tables reads one and writes the other, dev_cur says
    // which one holds the state of record)
    DeviceBuf<DevState> dev_state;
    int dev_cur = 0;
    DevState * dev_ptr() const { return dev_state.p + dev_cur; }
    DeviceBuf<int32_t> snap_counts;   // group sizes at batch entry
    // The fused launch between two batches (k_vs_tables: group set, caches
    // and per-value tables in one kernel) reads the per-group statistics from
    // one set of buffers and writes them, normalised, to another: these are
    // the other set (swapped with the live ones after every such launch).
    DeviceBuf<int32_t> alt_counts, alt_i0, alt_i1, alt_snap;
    // 0: k_normalise + k_batch_finish + k_vs_prepare as three launches (and
    // the handed-over rows as a fourth); 1 (default): k_vs_tables, the
    // handed-over rows inside k_vs_apply
    int fused_tables_mode = 1;
    uint64_t fused_batches = 0;
    // the open device-normalised run's last batch left its group set to the
    // next k_vs_tables (or to finish_pending())
    bool finish_pending = false;
    bool batch_fused = false;   // the open batch went through k_vs_tables
    bool async_active = false;
    // 0 never, 1 where it applies (2: the same; default)
    int device_normalise_mode = 2;
    // dist_gibbs_sweep_sharded may normalise on the device (the ranks agree
    // on it among themselves when a run is opened)
    bool sharded_device_normalise = true;
    uint64_t async_batches = 0;
    std::vector<hipEvent_t> ev_pool;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // The score+sample kernel of a batch sits between two events
    // (dist_gibbs_kernel_stats).  An event is a packet of its own on the
    // queue: around every batch they cost the headline workload 8 us of a
    // sub-sweep's 134 (events attached to the dispatch itself,
    // hipExtLaunchKernelGGL, cost the same).  "kernel_timing" = n times every
    // n-th batch (1: all, the default; 0: none).
    int kernel_timing = 1;
    uint64_t timing_tick = 0;
    bool timing_this_batch = true;
    void mark(hipEvent_t e) {
        if (timing_this_batch) HIP_CHECK(hipEventRecord(e, stream()));
    }
    double kernel_ms = 0.0;
    uint64_t kernel_launches = 0, kernel_rows = 0;
    // the all-reduce of dist_gibbs_sweep_sharded between two events, every
    // kernel_timing-th sub-sweep (dist_gibbs_comm_stats)
    double comm_ms = 0.0;
    uint64_t comm_launches = 0, comm_tick = 0;
    std::vector<hipEvent_t> comm_ev_free;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> comm_ev_pending;
    hipEvent_t comm_event() {
        if (comm_ev_free.empty()) {
            hipEvent_t e = nullptr;
            HIP_CHECK(hipEventCreate(&e));
            return e;
        }
        hipEvent_t e = comm_ev_free.back();
        comm_ev_free.pop_back();
        return e;
    }
    // "phase_timing" = 1: HIP events at the phase boundaries of every
    // device-normalised sub-sweep (tables | score+sample | handed-over rows |
    // statistics | group set + caches): a diagnostic -- six events cost a
    // sub-sweep some 20 us -- read by dist_gibbs_phase_stats
    static constexpr int kPhases = 5;
    int phase_timing = 0;
    double phase_ms[kPhases] = {0, 0, 0, 0, 0};
    uint64_t phase_batches = 0;
    std::vector<std::vector<hipEvent_t>> phase_pending;
    void phase_mark(int i) {
        if (!phase_timing) return;
        if (i == 0) phase_pending.emplace_back();
        if (phase_pending.empty() || (int)phase_pending.back().size() != i)
            return;   // (a path without all the marks: dropped below)
        hipEvent_t e = comm_event();
        HIP_CHECK(hipEventRecord(e, stream()));
        phase_pending.back().push_back(e);
    }
    // (call with the stream drained)
    void collect_comm_timing() {
        for (auto & ev : phase_pending) {
            if ((int)ev.size() == kPhases + 1) {
                bool ok = true;
                float ms[kPhases];
                for (int i = 0; i < kPhases; ++i)
                    ok = ok && hipEventElapsedTime(&ms[i], ev[i], ev[i + 1])
                                   == hipSuccess;
                if (ok) {
                    for (int i = 0; i < kPhases; ++i) phase_ms[i] += ms[i];
                    phase_batches += 1;
                }
            }
            for (hipEvent_t e : ev) comm_ev_free.push_back(e);
        }
        phase_pending.clear();
        for (auto & pr : comm_ev_pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                comm_ms += ms;
                comm_launches += 1;
            }
            comm_ev_free.push_back(pr.first);
            comm_ev_free.push_back(pr.second);
        }
        comm_ev_pending.clear();
    }

    Gibbs(float alpha_, float d_, int F, const dist_shared_t * shareds)
        : alpha(alpha_), d(d_) {
        ensure_device_ready();
        DIST_REQUIRE(F >= 0 && F <= kMaxF, "too many features");
        DIST_REQUIRE(alpha > 0.f && d >= 0.f && d < 1.f,
                     "expected alpha > 0, 0 <= d < 1");
        for (int f = 0; f < F; ++f)
            feats.emplace_back(new Slave(shareds[f]));
        HIP_CHECK(hipEventCreate(&ev0));
        HIP_CHECK(hipEventCreate(&ev1));
    }
    ~Gibbs() {
        if (publish_ev) (void)hipEventDestroy(publish_ev);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        for (hipEvent_t e : comm_ev_free) (void)hipEventDestroy(e);
        for (auto & pr : comm_ev_pending) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        if (peek_pending) (void)hipEventSynchronize(peek_event);
        if (peek_event) (void)hipEventDestroy(peek_event);
        if (pinned_state) (void)hipHostFree(pinned_state);
        if (pinned_counts) (void)hipHostFree(pinned_counts);
        if (pinned_seq) (void)hipHostFree(pinned_seq);
        if (pinned_pairs) (void)hipHostFree(pinned_pairs);
        if (comm_fault) (void)hipHostFree(comm_fault);
    }

    int F() const { return (int)feats.size(); }
    int K() const { return py.K(); }

    // both id maps travel in one copy: [p2g (capacity slots) | g2p]
    // min_p / min_g: room the device needs for the slots / ids it appends
    // itself (k_batch_finish)
    void upload_maps(size_t min_p = 0, size_t min_g = 0) {
        const size_t np = tracker.p2g.size(), ng = tracker.g2p.size();
        const bool roomy = maps_pcap >= std::max(np, min_p)
                           && d_maps.cap >= maps_pcap + std::max(ng, min_g);
        if (!maps_dirty && roomy) return;
        const size_t pcap = grow_capacity(std::max(np, min_p));
        maps_host.resize(pcap + ng);
        std::copy(tracker.p2g.begin(), tracker.p2g.end(), maps_host.begin());
        for (size_t i = 0; i < ng; ++i)
            maps_host[pcap + i] = (uint32_t)tracker.g2p[i];
        // headroom for groups the device appends itself (k_batch_finish)
        const size_t want = pcap + std::max(ng + 1024, min_g);
        if (want > d_maps.cap) d_maps.reserve(grow_capacity(want), 0);
        d_maps.upload(maps_host.data(), maps_host.size());
        d_p2g_ptr = d_maps.p;
        d_g2p_ptr = reinterpret_cast<const int32_t *>(d_maps.p + pcap);
        maps_pcap = pcap;
        maps_dirty = false;
    }

    SweepParams params(size_t r0, size_t r1, uint32_t seed, uint64_t draw_base) {
        SweepParams P;
        memset(&P, 0, sizeof(P));
        P.F = F();
        for (int f = 0; f < F(); ++f) {
            P.feat[f] = feats[f]->view();
            P.values[f] = values[f];
        }
        P.counts = py.d_counts.p;
        P.shifted = py.d_shifted.p;
        P.base = base.p;
        P.base_single = base_single.p;
        P.scalars = scalars.p;
        P.K = K();
        P.n_empty = py.n_empty;
        P.alpha = alpha;
        P.d = d;
        P.cluster = cluster;
        P.dataset_size = dataset_size;
        P.sample_size = py.sample_size;
        P.assign = assign;
        P.g2p = d_g2p_ptr;
        P.old_packed = old_packed.p;
        P.new_packed = new_packed.p;
        P.row_begin = r0;
        P.row_end = r1;
        P.row_offset = row_offset;
        P.draw_base = draw_base;
        P.seed_state = seed;
        // the state one step before the first row's draw, then per-row powers
        P.seed_batch = lcg_jump(seed, draw_base + row_offset + r0 + 1ull);
        ensure_pow_tables(r1 - r0);
        P.pow_lo = pow_lo.p;
        P.pow_hi = pow_hi.p;
        P.dev = async_active ? dev_ptr() : nullptr;
        return P;
    }

    void ensure_pow_tables(size_t batch_rows) {
        const uint32_t need = (uint32_t)(batch_rows / 4096 + 1);
        if (pow_lo.p && need <= n_pow_hi) return;
        n_pow_hi = std::max<uint32_t>(4096, need * 2);
        pow_lo.reserve(4096, 0);
        pow_hi.reserve(n_pow_hi, 0);
        LAUNCH(k_pow_tables, (size_t)std::max<uint32_t>(4096, n_pow_hi),
               pow_lo.p, pow_hi.p, n_pow_hi);
    }

    StatImage live_image() {
        StatImage img;
        memset(&img, 0, sizeof(img));
        img.counts = py.d_counts.p;
        for (int f = 0; f < F(); ++f) {
            img.i0[f] = feats[f]->i0.p;
            img.i1[f] = feats[f]->i1.p;
            img.cnt[f] = feats[f]->cnt.p;
        }
        return img;
    }
    // Mixture::validate / MixtureDriver::_validate (mixture.hpp:152-163,
    // 440-442) and what the reference's asserts stand for: every row's group
    // is live, the group sizes and every integer statistic equal a recount
    // from the rows, the host's mirror of the group set equals the device's,
    // there is an empty group, sizes sum to the rows assigned.
    void validate(dist_validate_report_t * rep) {
        DIST_REQUIRE(!batch_open, "validate: a batch is open");
        memset(rep, 0, sizeof(*rep));
        rep->feature = -1;
        flush_assign_pos();
        upload_maps();
        const size_t Kn = (size_t)K();
        // the host's mirror (MixtureDriver::_validate)
        std::vector<int> dev(Kn);
        py.d_counts.download(dev.data(), Kn);
        long long total = 0;
        int empties = 0;
        auto host_bad = [&](long long group, long long expected,
                            long long found, const char * what) {
            rep->code = VALIDATE_HOST;
            rep->group = group;
            rep->expected = expected;
            rep->found = found;
            snprintf(rep->what, sizeof(rep->what), "%s", what);
        };
        for (size_t k = 0; k < Kn && !rep->code; ++k) {
            if (dev[k] != py.counts[k])
                host_bad((long long)k, dev[k], py.counts[k],
                         "host mirror of a group size differs from the device");
            else if (dev[k] < 0)
                host_bad((long long)k, 0, dev[k], "negative group size");
            total += dev[k];
            empties += dev[k] == 0;
        }
        if (!rep->code && empties < 1)
            host_bad(-1, 1, 0, "missing empty groups");
        if (!rep->code && empties != py.n_empty)
            host_bad(-1, empties, py.n_empty, "empty-group count");
        if (!rep->code && total != py.sample_size)
            host_bad(-1, total, py.sample_size, "sample_size");
        if (!rep->code && tracker.p2g.size() != Kn)
            host_bad(-1, (long long)Kn, (long long)tracker.p2g.size(),
                     "id tracker's packed size");
        for (size_t k = 0; k < Kn && !rep->code; ++k) {
            const uint32_t id = tracker.p2g[k];
            if (id >= tracker.g2p.size() || tracker.g2p[id] != (int32_t)k)
                host_bad((long long)k, (long long)k,
                         id < tracker.g2p.size() ? tracker.g2p[id] : -2,
                         "id tracker's maps are not inverse");
        }
        for (auto & s : feats)
            if (!rep->code && (size_t)s->K != Kn)
                host_bad(-1, (long long)Kn, s->K, "a feature's group count");
        if (rep->code) return;
        // the recount on the device
        DeviceBuf<int32_t> words;
        words.reserve(std::max<size_t>(stat_words(), 1), 0);
        HIP_CHECK(hipMemsetAsync(words.p, 0, stat_words() * sizeof(int32_t),
                                 stream()));
        // (rows beyond assigned_rows have no group yet: load_rows_unassigned
        // / init_sequential, examples/mixture/main.py:227-232)
        const size_t rows = assigned_rows;
        DeviceBuf<uint32_t> packed;
        packed.reserve(std::max<size_t>(rows, 1), 0);
        DeviceBuf<unsigned long long> out;
        out.reserve(4, 0);
        const unsigned long long init[4] = {~0ull, 0ull, ~0ull, ~0ull};
        out.upload(init, 4);
        SweepParams P = params(0, rows, 1, 0);
        StatImage recount = word_image(words.p);
        if (rows)
            LAUNCH(k_validate_rows, rows, P, recount, d_g2p_ptr,
                   (uint32_t)tracker.g2p.size(), rows, packed.p, out.p,
                   out.p + 1);
        StatImage live = live_image();
        if (Kn) LAUNCH(k_validate_compare, Kn, P, live, recount, -1, Kn, out.p);
        for (int f = 0; f < F(); ++f) {
            const size_t width = 2 + (size_t)feats[f]->dim();
            LAUNCH(k_validate_compare, Kn * width, P, live, recount, f,
                   Kn * width, out.p);
        }
        unsigned long long got[4];
        out.download(got, 4);
        rep->rows_assigned = (long long)got[1];
        if (got[0] == ~0ull) {
            if ((long long)got[1] != total)
                host_bad(-1, (long long)got[1], total,
                         "group sizes do not sum to the rows assigned");
            return;
        }
        rep->code = (int)(got[0] >> 60);
        rep->feature = (int)((got[0] >> 56) & 15);
        rep->group = (long long)((got[0] >> 28) & 0xFFFFFFFull);
        rep->detail = (long long)(got[0] & 0xFFFFFFFull);
        // (a row check: the offending row whole, not its low 28 bits, and
        // what it carries read back from the row itself)
        if (rep->code == VALIDATE_DEAD_ID && got[2] != ~0ull) {
            rep->group = (long long)got[2];
            uint32_t id = 0;
            HIP_CHECK(hipMemcpyAsync(&id, assign + got[2], 4,
                                     hipMemcpyDeviceToHost, stream()));
            sync();
            rep->detail = (long long)id;
        } else if (rep->code == VALIDATE_VALUE_RANGE && got[3] != ~0ull
                   && got[3] >= (1ull << 28)) {
            rep->group = (long long)got[3];
            for (int f = 0; f < F(); ++f) {
                if (!is_cat(feats[f]->sh.kind)) continue;
                uint32_t x = 0;
                HIP_CHECK(hipMemcpyAsync(&x, values[f] + got[3], 4,
                                         hipMemcpyDeviceToHost, stream()));
                sync();
                if (x >= (uint32_t)feats[f]->dim()) {
                    rep->feature = f;
                    rep->detail = (long long)x;
                    break;
                }
            }
        }
        static const char * names[] = {
            "", "a row carries a group id that is not live",
            "a row's value is outside the feature's domain",
            "group size != rows assigned to the group",
            "statistic 0 (count_sum / heads / count) != recount",
            "statistic 1 (tails / sum) != recount",
            "categorical count cell != recount", ""};
        snprintf(rep->what, sizeof(rep->what), "%s", names[rep->code & 7]);
        // expected (recount) and found (live) for the report
        auto word_at = [&](const int32_t * p) {
            int32_t v = 0;
            HIP_CHECK(hipMemcpyAsync(&v, p, 4, hipMemcpyDeviceToHost, stream()));
            sync();
            return (long long)v;
        };
        const size_t k = (size_t)rep->group;
        const int f = rep->feature;
        switch (rep->code) {
        case VALIDATE_GROUP_SIZE:
            rep->feature = -1;
            rep->expected = word_at(recount.counts + k);
            rep->found = word_at(live.counts + k);
            break;
        case VALIDATE_STAT0:
            rep->expected = word_at(recount.i0[f] + k);
            rep->found = word_at(live.i0[f] + k);
            break;
        case VALIDATE_STAT1:
            rep->expected = word_at(recount.i1[f] + k);
            rep->found = word_at(live.i1[f] + k);
            break;
        case VALIDATE_CELL: {
            const size_t cell = k * feats[f]->dim() + (size_t)rep->detail;
            rep->expected = word_at(recount.cnt[f] + cell);
            rep->found = word_at(live.cnt[f] + cell);
            break;
        }
        default:
            break;
        }
    }

    // the integer statistics as one image of words, laid out for `k` groups
    // (group sizes | per feature: i0, i1, cells [k][dim])
    size_t stat_words(size_t k) const {
        size_t n = k;
        for (auto & s : feats) n += 2 * k + k * s->dim();
        return n;
    }
    size_t stat_words() const { return stat_words((size_t)K()); }
    StatImage word_image(int32_t * w) { return word_image(w, (size_t)K()); }
    StatImage word_image(int32_t * w, size_t k) {
        StatImage img;
        memset(&img, 0, sizeof(img));
        img.counts = w;
        w += k;
        for (int f = 0; f < F(); ++f) {
            img.i0[f] = w; w += k;
            img.i1[f] = w; w += k;
            img.cnt[f] = w; w += k * feats[f]->dim();
        }
        return img;
    }
    // copy the integer statistics between the live arrays and a word image
    void copy_stats(int32_t * words, bool to_words) {
        StatImage a = live_image(), b = word_image(words);
        const size_t k = (size_t)K();
        auto cp = [&](int32_t * live, int32_t * w, size_t n) {
            if (!n) return;
            HIP_CHECK(hipMemcpyAsync(to_words ? w : live, to_words ? live : w,
                                     n * 4, hipMemcpyDeviceToDevice, stream()));
        };
        cp(a.counts, b.counts, k);
        for (int f = 0; f < F(); ++f) {
            cp(a.i0[f], b.i0[f], k);
            cp(a.i1[f], b.i1[f], k);
            cp(a.cnt[f], b.cnt[f], k * feats[f]->dim());
        }
    }

    void rebuild_caches() {
        base_valid = false;
        py.rebuild(alpha, d);
        for (auto & s : feats) s->init();
    }
    // The group sizes are final once a batch's integer statistics are applied;
    // the ordered replay of its float statistics (C3: 0.26 of 2.2 ms) does
    // not touch them.  Published BEFORE the replay, they are on the host when
    // batch_finish asks, the host works out the normalisation while the
    // replay runs, and the kernels that follow start the moment it ends (they
    // used to start 25-35 us after it: the wait, the host's work, the
    // launches -- profiles/r5_timeline_c3.txt).
    unsigned early_ticket = 0;
    size_t early_n = 0;
    uint64_t batch_serial = 0, early_serial = 0;   // (which batch it is of)
    unsigned launch_publish(size_t n) {
        if (n > pinned_cap) {
            sync();   // (an earlier publish may still write the old buffer)
            if (pinned_counts) (void)hipHostFree(pinned_counts);
            pinned_cap = grow_capacity(n);
            HIP_CHECK(hipHostMalloc((void **)&pinned_counts,
                                    pinned_cap * sizeof(int),
                                    hipHostMallocCoherent));
        }
        if (!pinned_seq) {
            HIP_CHECK(hipHostMalloc((void **)&pinned_seq, sizeof(unsigned),
                                    hipHostMallocCoherent));
            *pinned_seq = 0;
        }
        const unsigned ticket = ++publish_ticket;
        hipLaunchKernelGGL(k_publish_counts, dim3(1), dim3(1024), 0, stream(),
                           py.d_counts.p, (int)n, pinned_counts,
                           (volatile unsigned *)pinned_seq, ticket);
        HIP_CHECK(hipGetLastError());
        // (what a waiter that has spun long enough sleeps on: THIS kernel,
        // not the stream -- the replay queued behind it is not its business)
        if (!publish_ev)
            HIP_CHECK(hipEventCreateWithFlags(
                &publish_ev, hipEventDisableTiming | hipEventBlockingSync));
        HIP_CHECK(hipEventRecord(publish_ev, stream()));
        return ticket;
    }
    hipEvent_t publish_ev = nullptr;
    void publish_counts_early() {
        // (k_vs_reduce publishes the sizes of a value-sorted batch itself;
        // an open device-normalised run never asks)
        if (async_active || (pairs_ticket != 0 && pairs_ticket == publish_ticket))
            return;
        early_n = (size_t)K();
        early_serial = batch_serial;
        early_ticket = launch_publish(early_n);
    }
    void refresh_host_counts() {
        const size_t n = (size_t)K();
        if (n > pinned_cap) {
            if (pinned_counts) (void)hipHostFree(pinned_counts);
            pinned_cap = grow_capacity(n);
            HIP_CHECK(hipHostMalloc((void **)&pinned_counts,
                                    pinned_cap * sizeof(int),
                                    hipHostMallocCoherent));
        }
        py.counts.resize(n);
        if (pairs_ticket != 0 && pairs_ticket == publish_ticket) {
            // k_vs_reduce is publishing them: wait for the ticket in every slot
            const unsigned long long want = pairs_ticket;
            pairs_ticket = 0;
            const volatile unsigned long long * pairs = pinned_pairs;
            bool seen = false;
            size_t k = 0;
            auto scan = [&] {
                while (k < n && (pairs[k] >> 32) == want) ++k;
                return k == n;
            };
            // a short poll of the pinned pairs (the publishing kernel is a
            // few microseconds away in the usual case), then the stream: a
            // kernel queued behind a long one or behind a collective that
            // waits for a late peer must not cost this rank a busy core
            const auto t_poll = std::chrono::steady_clock::now();
            while (!seen) {
                for (int spin = 0; spin < 256 && !seen; ++spin) seen = scan();
                if (std::chrono::steady_clock::now() - t_poll
                    > std::chrono::microseconds(200))
                    break;
            }
            if (!seen) {
                // Not there after the bounded spin: the publishing kernel may
                // simply be queued behind a long kernel or a collective that
                // waits for a late peer.  Drain the stream (which also
                // surfaces the error of a failed kernel) and look again;
                // only slots still without the ticket after that are a bug.
                HIP_CHECK(hipStreamSynchronize(stream()));
                std::atomic_thread_fence(std::memory_order_acquire);
                seen = scan();
                DIST_REQUIRE(seen, "group sizes were not published");
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            for (size_t i = 0; i < n; ++i)
                py.counts[i] = (int32_t)(uint32_t)pairs[i];
            return;
        }
        pairs_ticket = 0;
        // one block writes the sizes into the pinned buffer, then the ticket
        // (or did already: publish_counts_early)
        unsigned ticket = early_ticket;
        early_ticket = 0;
        if (!ticket || early_n != n || early_serial != batch_serial)
            ticket = launch_publish(n);
        bool seen = false;
        const auto t_poll = std::chrono::steady_clock::now();
        while (!seen) {
            for (int spin = 0; spin < 256 && !seen; ++spin)
                seen = *(volatile unsigned *)pinned_seq == ticket;
            if (std::chrono::steady_clock::now() - t_poll
                > std::chrono::microseconds(200))
                break;
        }
        // (not yet: sleep until the publishing kernel is done, which also
        // surfaces the error of a failed kernel -- that one never writes the
        // ticket)
        if (!seen) {
            HIP_CHECK(hipEventSynchronize(publish_ev));
            DIST_REQUIRE(*(volatile unsigned *)pinned_seq == ticket,
                         "group sizes were not published");
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        std::copy(pinned_counts, pinned_counts + n, py.counts.begin());
    }

    void load(size_t n, const uint32_t * const * vals, bool vals_on_device,
              const uint32_t * assign_packed, uint32_t * assign_dev,
              int nonempty, int empty, uint64_t offset) {
        DIST_REQUIRE(nonempty >= 0 && empty >= 1,
                     "missing empty groups");   // mixture.hpp:153
        n_rows = n;
        row_offset = offset;
        const int Kt = nonempty + empty;
        values.assign(F(), nullptr);
        own_values.clear();
        own_values.resize(F());
        DeviceBuf<uint32_t> packed_tmp;
        const uint32_t * packed_dev = nullptr;
        if (vals_on_device) {
            for (int f = 0; f < F(); ++f) values[f] = vals[f];
            packed_dev = assign_dev;
            assign = assign_dev;       // rewritten to global ids below
        } else if (assign_packed == nullptr) {
            // rows without a group yet (init_sequential assigns them)
            DIST_REQUIRE(nonempty == 0, "unassigned rows: no group has members");
            for (int f = 0; f < F(); ++f) {
                own_values[f].upload(vals[f], n);
                values[f] = own_values[f].p;
            }
            own_assign.reserve(std::max<size_t>(n, 1), 0);
            HIP_CHECK(hipMemsetAsync(own_assign.p, 0xFF,
                                     std::max<size_t>(n, 1) * 4, stream()));
            assign = own_assign.p;
        } else {
            std::vector<uint32_t> chk(assign_packed, assign_packed + n);
            for (size_t i = 0; i < n; ++i)
                DIST_REQUIRE(chk[i] < (uint32_t)nonempty, "bad groupid in assignments");
            for (int f = 0; f < F(); ++f) {
                own_values[f].upload(vals[f], n);
                values[f] = own_values[f].p;
            }
            own_assign.upload(assign_packed, n);
            packed_dev = own_assign.p;
            assign = own_assign.p;
        }
        const bool unassigned = !vals_on_device && assign_packed == nullptr;
        assigned_rows = unassigned ? 0 : n;
        if (vals_on_device && n) {   // the host-pointer form checked above
            DeviceBuf<uint32_t> mx;
            mx.reserve(1, 0);
            LAUNCH(k_max_value, std::min<size_t>(n, (size_t)2048 * kBlock), packed_dev, n, mx.p);
            uint32_t top = 0;
            mx.download(&top, 1);
            DIST_REQUIRE(top < (uint32_t)nonempty, "bad groupid in assignments");
        }
        // largest value of each discrete feature (validation; sizes the
        // value tables of the count-valued ones)
        max_value.assign((size_t)F(), 0);
        resume_bound = resume_left = 0;
        value_partitioned = cells_partial = false;
        vs_cache.clear();
        vs_ranges.clear();
        vs_last = 0;
        fold_cache.clear();
        for (int f = 0; f < F(); ++f) {
            const int kind = feats[f]->sh.kind;
            if (kind == DIST_NICH || !n) continue;
            DeviceBuf<uint32_t> mx;
            mx.reserve(1, 0);
            LAUNCH(k_max_value, std::min<size_t>(n, (size_t)2048 * kBlock), values[f], n, mx.p);
            mx.download(&max_value[f], 1);
            // the reference asserts these in its debug builds (dd.hpp:125,
            // bb.hpp value is a bool); here an out-of-range value would
            // index outside the statistics
            if (is_cat(kind))
                DIST_REQUIRE(max_value[f] < (uint32_t)feats[f]->sh.dim,
                             "value out of bounds in feature "
                                 + std::to_string(f) + ": "
                                 + std::to_string(max_value[f]));
            if (kind == DIST_BB)
                DIST_REQUIRE(max_value[f] <= 1u,
                             "BetaBernoulli value is not 0/1 in feature "
                                 + std::to_string(f));
        }
        // empty statistics for all groups
        py.counts.assign((size_t)Kt, 0);
        py.reserve(Kt);
        HIP_CHECK(hipMemsetAsync(py.d_counts.p, 0, sizeof(int32_t) * Kt, stream()));
        for (auto & s : feats) {
            s->clear();
            s->append_zero(Kt);
        }
        tracker.init((size_t)Kt);
        maps_dirty = true;
        upload_maps();
        old_packed.reserve(std::max<size_t>(n, 1), 0);
        new_packed.reserve(std::max<size_t>(n, 1), 0);
        base.reserve((size_t)grow_capacity(Kt), 0);
        base_single.reserve((size_t)grow_capacity(Kt), 0);
        scalars.reserve(1, 0);
        // integer statistics by atomics, float statistics replayed in row
        // order (Group::add_value per row, like GroupIoMixin.from_values,
        // distributions/mixins.py:83-90)
        SweepParams P = params(0, n, 0, 0);
        if (n && !unassigned) {
            LAUNCH(k_load_counts, n, P, live_image(), packed_dev);
            replay_sorted(nullptr, packed_dev, 0, n);
            // assignments become global ids (identity map right after init)
            LAUNCH(k_packed_to_global, n, packed_dev, d_p2g_ptr, assign, n);
        }
        refresh_host_counts();
        rebuild_caches();
        sync();
    }

    // which template instance scores this feature list
    template <class Fn>
    void dispatch(Fn && fn) {
        const int k0 = F() > 0 ? feats[0]->sh.kind : -1;
        const int k1 = F() > 1 ? feats[1]->sh.kind : -1;
        if (F() == 1 && k0 == DIST_DD) return fn.template run<DIST_DD, -1, 1>();
        if (F() == 1 && k0 == DIST_DPD) return fn.template run<DIST_DPD, -1, 1>();
        if (F() == 1 && k0 == DIST_BB) return fn.template run<DIST_BB, -1, 1>();
        if (F() == 1 && k0 == DIST_GP) return fn.template run<DIST_GP, -1, 1>();
        if (F() == 1 && k0 == DIST_NICH) return fn.template run<DIST_NICH, -1, 1>();
        if (F() == 1 && k0 == DIST_BNB) return fn.template run<DIST_BNB, -1, 1>();
        if (F() == 2 && k0 == DIST_GP && k1 == DIST_NICH)
            return fn.template run<DIST_GP, DIST_NICH, 2>();
        return fn.template run<-1, -1, 0>();
    }

    // size of feature f's k-major gather table (0 = none)
    int ktab_values(int f) const {
        const dist_shared_t & sh = feats[f]->sh;
        size_t nv = 0;
        if (sh.kind == DIST_GP || sh.kind == DIST_BNB)
            nv = std::min<size_t>(max_value[f] + 1, 64);
        if (sh.kind == DIST_BB) nv = 2;
        if (is_cat(sh.kind)) nv = (size_t)sh.dim;
        if (nv * (size_t)K() > ((size_t)256 << 20)) return 0;   // > 1 GiB
        return (int)nv;
    }
    // with_ktab = false: only tables that cost next to nothing are built
    void prepare(SweepParams & P, bool with_ktab = true) {
        base.reserve(grow_capacity((size_t)K()), 0);
        base_single.reserve(grow_capacity((size_t)K()), 0);
        P.base = base.p;
        P.base_single = base_single.p;
        if (!base_valid) {   // batch_finish leaves them ready
            LAUNCH(k_sweep_prepare, (size_t)K(), P, base.p, base_single.p,
                   scalars.p);
            base_valid = true;
        }
        ktab.resize((size_t)F());
        for (int f = 0; f < F(); ++f) {
            int nv = ktab_values(f);
            if (!with_ktab && (size_t)nv * K() > ((size_t)1 << 18)) nv = 0;
            P.ktab[f] = nullptr;
            P.ktab_nv[f] = 0;
            if (!nv) continue;
            const size_t n = (size_t)K() * nv;
            // (+ a block of padding groups: k_rows_scratch scores whole
            // blocks of kRowsBlock groups and masks afterwards)
            ktab[f].reserve(grow_capacity(n + (size_t)kRowsBlock * nv), 0);
            LAUNCH(k_build_ktab, n, feats[f]->view(), ktab[f].p, nv, K());
            P.ktab[f] = ktab[f].p;
            P.ktab_nv[f] = nv;
        }
    }

    // feature lists without a compile-time instance of the sampling kernel
    bool uses_runtime_kernel() const {
        const int k0 = F() > 0 ? feats[0]->sh.kind : -1;
        const int k1 = F() > 1 ? feats[1]->sh.kind : -1;
        if (F() == 1) return false;
        if (F() == 2 && k0 == DIST_GP && k1 == DIST_NICH) return false;
        return F() >= 2;
    }
    // k_rows_scratch over the open batch: a grid of exactly the workgroups
    // that are resident at once (each wave keeps a block of the scratch and
    // walks the rows grid-stride).  false: program or scratch do not fit.
    bool launch_rows_scratch(SweepParams & P, const ScoreProgram & prog,
                             size_t n) {
        RowsArgs A;
        memset(&A, 0, sizeof(A));
        A.pad = rows_prio_mode;   // (wave priorities by pass)
        GtabSource src;
        memset(&src, 0, sizeof(src));
        int n_ops = 0;
        const float * par[kRowsMaxOps][4];   // an op's per-group parameters
        memset(par, 0, sizeof(par));
        const int K8 = (K() + kRowsBlock - 1) / kRowsBlock * kRowsBlock;
        for (int o = 0; o < prog.n; ++o) {
            const ScoreOp & op = prog.op[o];
            if (n_ops == kRowsMaxOps) return false;
            RowsOp & r = A.op[n_ops];
            r.values = P.values[op.f];
            if (op.type == OP_GATHER_ADD) {
                r.type = ROP_GATHER;
                r.tab = op.p0;
                r.row_bytes = op.nv * 4u;
                // (prepare() allocates the padding groups' rows)
                const size_t bytes = (size_t)K8 * op.nv * 4;
                if (bytes >= ((size_t)1 << 31)) return false;
                r.tab_bytes = (uint32_t)bytes;
                if (o + 1 < prog.n && prog.op[o + 1].type == OP_VEC_SUB
                    && prog.op[o + 1].f == op.f) {
                    // a categorical feature: (acc + S[x][k]) - shift[k]
                    r.type = ROP_CAT;
                    par[n_ops][0] = prog.op[o + 1].p0;
                    o += 1;
                }
            } else if (op.type == OP_NICH) {
                r.type = ROP_NICH;
                par[n_ops][0] = op.p0; par[n_ops][1] = op.p1;
                par[n_ops][2] = op.p2; par[n_ops][3] = op.p3;
            } else {
                return false;   // (a shift without its table: not a program
                                //  sample_by_program builds)
            }
            n_ops += 1;
        }
        // fold the leading table ops into a per-(joint value, group) base
        FoldSpec F;
        memset(&F, 0, sizeof(F));
        uint32_t J = 1;
        // (a multiple of the snapshot distance, a block of padding groups
        // behind the last: whole blocks are scored, the surplus is masked)
        const int Kpad = (K() + kRowsSuper - 1) / kRowsSuper * kRowsSuper
                         + kRowsSuper;
        if (rows_fold_mode == 2 || (rows_fold_mode == 1 && n >= 8192)) {
            const size_t per_code = rows_fold_mode == 2 ? 1 : kFoldRowsPerCode;
            for (int o = 0; o < n_ops; ++o) {
                const RowsOp & r = A.op[o];
                if (r.type == ROP_NICH) break;
                const uint32_t nv = r.row_bytes / 4u;
                const size_t Jn = (size_t)J * nv;
                if (Jn * per_code > n || Jn > 65536
                    || Jn * Kpad * 4 > ((size_t)256 << 20))
                    break;
                F.nv[F.n] = nv;
                F.values[F.n] = r.values;
                F.tab[F.n] = r.tab;
                F.shift[F.n] = r.type == ROP_CAT ? par[o][0] : nullptr;
                F.n += 1;
                J = (uint32_t)Jn;
            }
            // the first folded feature is the most significant digit
            uint32_t stride = J;
            for (int o = 0; o < F.n; ++o) {
                stride /= F.nv[o];
                F.stride[o] = stride;
            }
        }
        if (F.n) {   // the remaining ops move to the front
            for (int o = F.n; o < n_ops; ++o) {
                A.op[o - F.n] = A.op[o];
                for (int i = 0; i < 4; ++i) par[o - F.n][i] = par[o][i];
            }
            n_ops -= F.n;
            for (int o = n_ops; o < kRowsMaxOps; ++o)
                memset(&A.op[o], 0, sizeof(RowsOp));
        }
        // the gtab slots and the shape of what remains
        bool nich = false;
        int next = 1;   // slot 0: the driver's score
        int shape = 0;
        for (int o = 0; o < n_ops; ++o) {
            RowsOp & r = A.op[o];
            if (r.type == ROP_CAT) {
                r.slot = next;
                src.p[src.n++] = par[o][0];
                next += 1;
            } else if (r.type == ROP_NICH) {
                nich = true;
                r.slot = next;
                for (int i = 0; i < 4; ++i) src.p[src.n++] = par[o][i];
                next += 4;
            }
            if (o < 4) {
                int mul = 1;
                for (int i = 0; i < o; ++i) mul *= 4;
                shape += (1 + r.type) * mul;
            }
        }
        const int W = next;
        if (W > kRowsMaxW) return false;
        const bool lds_log = nich && rows_scratch_lds_log != 0;
        // the exact loops, or scan sampling (tolerance-level, option
        // "sampling")
        const bool scan = sampling_mode == 1;
        const int block = rows_scratch_block;
        // the program's shape at compile time where an instance exists
        if (n_ops > 4) shape = 0;
        const void * fn = nullptr;
        int shape_id = 0;
#define ROWS_SCRATCH_M(M, SHAPE)                                             \
        do {                                                                 \
            if (lds_log)                                                     \
                fn = (const void *)&k_rows_scratch<M, true, SHAPE>;          \
            else                                                             \
                fn = (const void *)&k_rows_scratch<M, false, SHAPE>;         \
        } while (0)
#define ROWS_SCRATCH(ID, SHAPE)                                              \
        do {                                                                 \
            shape_id = ID;                                                   \
            if (scan) ROWS_SCRATCH_M(true, SHAPE);                           \
            else ROWS_SCRATCH_M(false, SHAPE);                               \
        } while (0)
        if (shape == kShapeGN) ROWS_SCRATCH(1, kShapeGN);
        else if (shape == kShapeN) ROWS_SCRATCH(2, kShapeN);
        else if (shape == kShapeNN) ROWS_SCRATCH(3, kShapeNN);
        else if (shape == kShapeG) ROWS_SCRATCH(4, kShapeG);
        else if (shape == kShapeC) ROWS_SCRATCH(5, kShapeC);
        else ROWS_SCRATCH(0, 0);
#undef ROWS_SCRATCH
#undef ROWS_SCRATCH_M
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        // workgroups of `block` threads resident per CU, per instance
        static std::atomic<int> per_cu[2][2][6][64];
        std::atomic<int> & cached =
            per_cu[scan ? 1 : 0][lds_log ? 1 : 0][shape_id][dev & 63];
        int resident = cached.load(std::memory_order_relaxed);
        if (resident == 0 || resident / 4096 != block) {
            int nb = 0;
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(
                &nb, fn, block, 0));
            DIST_REQUIRE(nb >= 1, "k_rows_scratch does not fit a CU");
            resident = block * 4096 + nb;
            cached.store(resident, std::memory_order_relaxed);
        }
        const int wgs_per_cu = resident % 4096;
        size_t blocks = (size_t)wgs_per_cu * cu_count();
        const size_t blocks_cap = blocks;
        if (scan)
            rows_snap.reserve(blocks * (size_t)(block / 64)
                                  * (Kpad / kRowsSuper) * 64, 0);
        rows_gtab.reserve(grow_capacity((size_t)Kpad * W), 0);
        LAUNCH(k_rows_gtab, (size_t)Kpad, P.base, src, rows_gtab.p, Kpad, K(),
               P.dev);
        size_t n_work = (n + 63) / 64;
        if (F.n) {
            FoldCache & fc = fold_get(P.row_begin, P.row_end, F, J);
            rows_fold.reserve(grow_capacity((size_t)J * Kpad), 0);
            LAUNCH(k_rows_fold, (size_t)J * Kpad, F, P.base, rows_fold.p, J,
                   Kpad, K(), P.dev);
            A.fold = rows_fold.p;
            A.sorted_rows = fc.sorted_rows.p;
            A.tiles = fc.tiles.p;
            A.n_tiles = fc.n_tiles;
            A.fold_codes = J;
            n_work = fc.n_tiles;
            fold_batches += 1;
        }
        A.n_ops = n_ops;
        A.W = W;
        A.K = K();
        A.Kpad = Kpad;
        A.dev = P.dev;
        A.gtab = rows_gtab.p;
        A.slot = P.old_packed;
        A.own = own_score.p;
        A.new_packed = P.new_packed;
        A.row_begin = P.row_begin;
        A.n_items = n;
        A.seed_batch = P.seed_batch;
        A.pow_lo = P.pow_lo;
        A.pow_hi = P.pow_hi;
        A.snap = rows_snap.p;
        blocks = std::max<size_t>(
            1, std::min(blocks_cap,
                        (n_work + block / 64 - 1) / (size_t)(block / 64)));
        void * args[] = {&A};
        HIP_CHECK(hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(block),
                                  args, 0, stream()));
        scratch_batches += 1;
        return true;
    }

    // Rows of mixed type: compile the feature list into a ScoreProgram over
    // this batch's tables (kernels.h) and sample with k_sweep_program; rows it
    // hands over go to the wave-per-row kernel.  false: a table is missing
    // (a categorical feature too wide to tabulate), take the model-code kernel.
    bool sample_by_program(SweepParams & P) {
        prepare(P);   // base[], and every feature's k-major table
        ScoreProgram prog;
        memset(&prog, 0, sizeof(prog));
        for (int f = 0; f < F(); ++f) {
            const int kind = feats[f]->sh.kind;
            const SlaveView v = feats[f]->view();
            ScoreOp & op = prog.op[prog.n];
            op.f = f;
            if (kind == DIST_NICH) {
                op.type = OP_NICH;
                op.p0 = v.c0; op.p1 = v.c1; op.p2 = v.c2; op.p3 = v.c3;
                prog.n += 1;
                continue;
            }
            if (!P.ktab[f]) return false;
            op.type = OP_GATHER_ADD;
            op.nv = (uint32_t)P.ktab_nv[f];
            op.p0 = P.ktab[f];
            prog.n += 1;
            if (is_cat(kind)) {
                ScoreOp & sub = prog.op[prog.n];
                sub.type = OP_VEC_SUB;
                sub.f = f;
                sub.p0 = v.c0;
                prog.n += 1;
            }
        }
        const size_t n = P.row_end - P.row_begin;
        own_score.reserve(std::max<size_t>(n, 1), 0);
        deferred.reserve(std::max<size_t>(n, 1), 0);
        deferred_count.reserve(1, 0);
        HIP_CHECK(hipMemsetAsync(deferred_count.p, 0, 4, stream()));
        mark(ev0);
        LAUNCH(k_row_prepass, n, P, prog, own_score.p, deferred.p,
               deferred_count.p);
        if ((rows_scratch_mode == 0 && sampling_mode == 0)
            || !launch_rows_scratch(P, prog, n)) {
            const unsigned blocks = (unsigned)std::min<size_t>(
                (n + kBlock - 1) / kBlock, 256 * 16);
            hipLaunchKernelGGL(k_sweep_program, dim3(blocks), dim3(kBlock), 0,
                               stream(), P, prog, own_score.p);
            HIP_CHECK(hipGetLastError());
        }
        // the handed-over rows (listed by batch row, in row order)
        SweepParams Q = P;
        Q.row_list = deferred.p;
        Q.row_list_count = deferred_count.p;
        Q.sorted_rows = nullptr;
        if (wave_rows_fit()) {
            WaveRowsLaunch D{&Q, K(), 256};
            dispatch(D);
        } else {
            DeferredLaunch D{&Q};
            dispatch(D);
        }
        mark(ev1);
        return true;
    }
    struct SampleLaunch {
        Gibbs * self;
        SweepParams * P;
        template <int A, int B, int NF>
        void run() {
            const size_t n = P->row_end - P->row_begin;
            // >> 256 workgroups to fill 256 CUs; grid-stride beyond that
            const unsigned blocks = (unsigned)std::min<size_t>(
                (n + kBlock - 1) / kBlock, 256 * 16);
            self->mark(self->ev0);
            hipLaunchKernelGGL((k_sweep_sample<A, B, NF>), dim3(blocks),
                               dim3(kBlock), 0, stream(), *P);
            HIP_CHECK(hipGetLastError());
            self->mark(self->ev1);
        }
    };
    struct RowScoreLaunch {
        Gibbs * self;
        SweepParams * P;
        size_t row;
        template <int A, int B, int NF>
        void run() {
            hipLaunchKernelGGL((k_row_scores<A, B, NF>), dim3(1), dim3(1), 0,
                               stream(), *P, row, self->row_scores.p,
                               self->row_size.p);
            HIP_CHECK(hipGetLastError());
        }
    };

    int vs_nvals() const {
        const dist_shared_t & sh = feats[0]->sh;
        if (sh.kind == DIST_BB) return 2;
        if (sh.kind == DIST_GP || sh.kind == DIST_BNB)
            // counts beyond the table: generic kernel
            return (int)std::min<uint32_t>(max_value[0] + 1, 256);
        return sh.dim;
    }
    bool use_value_sorted(size_t rows) const {
        if (value_sorted_mode == 0 || F() != 1) return false;
        const int kind = feats[0]->sh.kind;
        if (kind != DIST_DD && kind != DIST_DPD && kind != DIST_BB
            && kind != DIST_GP && kind != DIST_BNB)
            return false;
        if (value_sorted_mode == 2) return true;
        // (a value-partitioned rank meets only ITS values: what counts is
        // the rows per value that is there, not per value of the domain)
        const size_t nv = value_partitioned && present_values
                              ? std::min<size_t>(present_values, vs_nvals())
                              : (size_t)vs_nvals();
        return rows >= (size_t)16 * nv && rows >= 4096;
    }

    VsCache & vs_get(size_t r0, size_t r1) {
        // (a pass visits its ranges in order: the one after the last hit first)
        const std::pair<size_t, size_t> want(r0, r1);
        for (size_t probe : {vs_last, vs_last + 1})
            if (probe < vs_ranges.size() && vs_ranges[probe] == want) {
                vs_last = probe;
                return *vs_cache[probe];
            }
        for (size_t i = 0; i < vs_ranges.size(); ++i)
            if (vs_ranges[i] == want) {
                vs_last = i;
                return *vs_cache[i];
            }
        std::unique_ptr<VsCache> c(new VsCache());
        c->r0 = r0;
        c->r1 = r1;
        const size_t n = r1 - r0;
        const uint32_t nv = (uint32_t)vs_nvals();
        DeviceBuf<uint32_t> hist;
        hist.reserve(nv + 1, 0);   // zero-filled
        const dim3 sort_grid((unsigned)((n + kVsSortRows - 1) / kVsSortRows));
        hipLaunchKernelGGL(k_vs_hist, sort_grid, dim3(kBlock), 0, stream(),
                           values[0], r0, n, nv, hist.p);
        HIP_CHECK(hipGetLastError());
        std::vector<uint32_t> h(nv + 1), start(nv + 2, 0);
        hist.download(h.data(), nv + 1);
        for (uint32_t x = 0; x <= nv; ++x) start[x + 1] = start[x] + h[x];
        DeviceBuf<uint32_t> cursor;
        cursor.upload(start.data(), nv + 1);
        c->val_start.upload(start.data(), nv + 1);
        c->sorted_rows.reserve(std::max<size_t>(n, 1), 0);
        hipLaunchKernelGGL(k_vs_scatter, sort_grid, dim3(kBlock), 0, stream(),
                           values[0], r0, n, nv, cursor.p, c->sorted_rows.p);
        HIP_CHECK(hipGetLastError());
        // (Listing a value's tiles so that every SIMD gets tiles with long
        // and with short first passes alike -- four from the front, four from
        // the back -- was measured: 70.9 against 69.7 us per launch on C2.
        // Consecutive tiles of one value share their scalar-cache lines.)
        std::vector<VsTile> tiles;
        auto cut_tiles = [&](uint32_t rows) {
            tiles.clear();
            for (uint32_t x = 0; x < nv; ++x)
                for (uint32_t off = 0; off < h[x]; off += rows)
                    tiles.push_back(VsTile{x, start[x] + off,
                                           std::min<uint32_t>(rows,
                                                              h[x] - off)});
            c->n_tiles = (uint32_t)tiles.size();
        };
        cut_tiles(64 * kVsR);
        for (uint32_t x = 0; x < nv; ++x) c->n_values_present += h[x] != 0;
        // (k_vs_stream deals a tile's rows to lanes of one class each: a
        // spare lane keeps a full tile from handing a row over)
        if (use_stream(*c)) cut_tiles(64 * kVsR - kVsR);
        std::vector<VsTile> narrow;
        if (c->n_tiles < kVsNarrowBelowTiles || narrow_mode == 2) {
            for (uint32_t x = 0; x < nv; ++x)
                for (uint32_t off = 0; off < h[x]; off += 64)
                    narrow.push_back(VsTile{x, start[x] + off,
                                            std::min<uint32_t>(64, h[x] - off)});
            c->n_narrow_tiles = (uint32_t)narrow.size();
        }
        // apply work items: up to kVsApplyRows rows of one value -- or, where
        // values have few rows each (the table-free kernel's case) and the
        // kind is categorical, of several WHOLE values (k_vs_apply_mixed)
        std::vector<VsTile> chunks;
        const bool pack = is_cat(feats[0]->sh.kind) && use_stream(*c);
        bool open = false;   // the last chunk can still take whole values
        for (uint32_t x = 0; x <= nv; ++x) {
            if (pack && x < nv && h[x] && h[x] <= (uint32_t)kVsApplyRows / 2) {
                if (open && chunks.back().n + h[x] <= (uint32_t)kVsApplyRows) {
                    chunks.back().x = kVsMixedChunk;
                    chunks.back().n += h[x];
                } else {
                    chunks.push_back(VsTile{x, start[x], h[x]});
                    open = true;
                }
                continue;
            }
            for (uint32_t off = 0; off < h[x]; off += kVsApplyRows)
                chunks.push_back(VsTile{x, start[x] + off,
                                        std::min<uint32_t>(kVsApplyRows,
                                                           h[x] - off)});
            if (h[x]) open = false;
        }
        // (a chunk's `chunk` word: how many chunks its value has -- the
        // cells of a value with several are k_vs_reduce's in a fused batch)
        {
            std::vector<uint32_t> multi;
            for (size_t i = 0; i < chunks.size();) {
                size_t j = i + 1;
                while (j < chunks.size() && chunks[j].x == chunks[i].x
                       && chunks[i].x != kVsMixedChunk) ++j;
                const bool in_table = chunks[i].x < nv;
                for (size_t q = i; q < j; ++q)
                    chunks[q].chunk = in_table ? (uint32_t)(j - i) : 1u;
                if (j - i > 1 && in_table) {
                    multi.push_back(chunks[i].x);
                    multi.push_back((uint32_t)i);
                    multi.push_back((uint32_t)(j - i));
                }
                i = j;
            }
            c->n_multi = (uint32_t)(multi.size() / 3);
            if (c->n_multi) c->multi.upload(multi.data(), multi.size());
        }
        c->n_chunks = (uint32_t)chunks.size();
        c->one_chunk_per_value = true;
        c->mixed_chunks = false;
        for (uint32_t x = 0; x <= nv; ++x)
            if (h[x] > (uint32_t)kVsApplyRows) c->one_chunk_per_value = false;
        for (auto & ch : chunks) c->mixed_chunks |= ch.x == kVsMixedChunk;
        c->chunks.upload(chunks.data(), chunks.size());
        // every tile's rows lie in one chunk (both lists ascend in position)
        auto set_chunks = [&](std::vector<VsTile> & list) {
            uint32_t ci = 0;
            for (auto & t : list) {
                while (ci + 1 < chunks.size()
                       && chunks[ci].pos + chunks[ci].n <= t.pos) ++ci;
                t.chunk = ci;
            }
        };
        set_chunks(tiles);
        set_chunks(narrow);
        c->tiles.upload(tiles.data(), tiles.size());
        if (!narrow.empty())
            c->narrow_tiles.upload(narrow.data(), narrow.size());
        {
            std::vector<uint32_t> first(nv + 1, 0);
            uint32_t ci = 0;
            for (uint32_t x = 0; x <= nv; ++x) {
                while (ci < chunks.size()
                       && chunks[ci].pos + chunks[ci].n <= start[x]) ++ci;
                first[x] = ci;   // (the chunk that holds the value's first row)
            }
            c->chunk_first.upload(first.data(), first.size());
            c->n_table_chunks = first[nv];
            c->one_value_chunks = !c->mixed_chunks;
        }
        c->def_counts.reserve(std::max<size_t>(chunks.size(), 1), 0);   // zeros
        // rows whose value is outside the table: generic kernel, by position
        c->n_other = h[nv];
        if (c->n_other) {
            std::vector<uint32_t> idx(c->n_other);
            for (uint32_t i = 0; i < c->n_other; ++i) idx[i] = start[nv] + i;
            c->other_pos.upload(idx.data(), idx.size());
        }
        // current assignments in position order
        flush_assign_pos();
        c->assign_pos.reserve(std::max<size_t>(n, 1), 0);
        LAUNCH(k_pos_gather, n, assign + r0, c->sorted_rows.p,
               c->assign_pos.p, n);
        sync();
        // room for every range of a pass over the rows at this batch size (a
        // pass that evicts its own ranges pays the sort, two host round
        // trips and the loss of the group-sorted order on every batch); the
        // ranges together hold 8 B per row plus the tile lists
        const size_t keep = std::min<size_t>(
            std::max<size_t>(64, (n_rows + n - 1) / std::max<size_t>(n, 1) + 2),
            1u << 16);
        if (vs_cache.size() >= keep) {
            flush_assign_pos();
            sync();   // the evicted range's buffers are freed below
            vs_cache.erase(vs_cache.begin());
            vs_ranges.erase(vs_ranges.begin());
        }
        vs_cache.push_back(std::move(c));
        vs_ranges.push_back(want);
        vs_last = vs_cache.size() - 1;
        return *vs_cache.back();
    }

    struct VsLaunch {
        Gibbs * self;
        SweepParams * P;
        VsCache * c;
        VsTables T;
        bool narrow;
        bool fused;
        template <int KIND>
        void go() {
            const uint32_t nv = (uint32_t)self->vs_nvals();
            if (fused) {
                HOST_PROBE(8, "      launch_tables");
                self->launch_tables<KIND>(*P, T, *c);
            } else
                hipLaunchKernelGGL((k_vs_prepare<KIND>), dim3(nv), dim3(kBlock),
                                   T.PA ? (size_t)T.Kpad * 8 : 0,
                                   stream(), *P, T, self->deferred_count.p,
                                   c->n_other);
            HIP_CHECK(hipGetLastError());
            // (A variant in which the tile's own wave sampled such rows on a
            // strip of LDS once its tile was done -- off k_vs_apply's path --
            // was measured: k_vs_apply 17.4 -> 14.8 us on average, k_vs_sample
            // 79 -> 86: its 64 registers spill.)
            const VsDefer D{self->deferred.p, self->deferred_count.p,
                            fused ? c->def_counts.p : nullptr, c->chunks.p,
                            0};
            self->phase_mark(1);
            {
                HOST_PROBE(10, "      mark(ev0)");
                self->mark(self->ev0);
            }
            HOST_PROBE(11, "      sample launch+mark");
            // a launch that cannot fill the chip spreads out: a wave per
            // workgroup (no band tiles on such launches)
            if (narrow) {
                // (waves alone or in pairs on their SIMDs read a whole chunk
                // ahead; more of them half a chunk, and four fit)
                const size_t lds =
                    2 * ((size_t)T.Kuse + 2 * kVsUnroll) * sizeof(float);
#define VS_NARROW(HQ)                                                        \
                hipLaunchKernelGGL(                                          \
                    (k_vs_narrow<KIND, HQ>), dim3(c->n_narrow_tiles),        \
                    dim3(64), lds, stream(), *P, T, c->narrow_tiles.p,       \
                    c->n_narrow_tiles, c->sorted_rows.p, D)
                const bool whole =
                    self->narrow_read_ahead
                        ? self->narrow_read_ahead == 8
                        : c->n_narrow_tiles <= 8u * (uint32_t)self->cu_count();
                if (whole)
                    VS_NARROW(8);
                else
                    VS_NARROW(4);
#undef VS_NARROW
                HIP_CHECK(hipGetLastError());
                self->mark(self->ev1);
                return;
            }
            const bool small = !T.band_mode && c->n_tiles < 4096;
            const uint32_t per =
                small ? 1 : kVsSampleBlock / 64;   // tiles per workgroup
            // (band tiles first, a whole number of workgroups)
            const uint32_t band_ids =
                T.band_mode ? (T.band_count + per - 1) / per * per : 0;
            const dim3 grid((band_ids + c->n_tiles + per - 1) / per);
            if (c->n_tiles && small)
                hipLaunchKernelGGL((k_vs_sample<KIND, 64>), grid, dim3(64), 0,
                                   stream(), *P, T, c->tiles.p, c->n_tiles,
                                   band_ids, c->sorted_rows.p, D);
            else if (c->n_tiles)   // else every value lies beyond the table
                hipLaunchKernelGGL((k_vs_sample<KIND, kVsSampleBlock>), grid,
                                   dim3(kVsSampleBlock), 0,
                                   stream(), *P, T, c->tiles.p, c->n_tiles,
                                   band_ids, c->sorted_rows.p, D);
            HIP_CHECK(hipGetLastError());
            self->mark(self->ev1);
        }
    };
    // May this batch of the open device-normalised run take the fused launch
    // (k_vs_tables; the handed-over rows inside k_vs_apply)?  Its LDS holds
    // four words per group; k_vs_apply must be the sorting form with a staging
    // matrix (its sort buffers are the handed-over rows' strips, and
    // k_vs_reduce writes what the fused launch reads).
    bool fused_ok(const VsCache & c, int Kpad) const {
        if (!fused_tables_mode || !async_active || sampling_mode != 0)
            return false;
        if (Kpad > kTablesMaxK || c.mixed_chunks || any_float_stats())
            return false;
        // rows beyond the value table bring their own value to the sums,
        // unstaged (k_vs_apply), while the chunks' handed-over rows read
        // those sums: not in one launch
        const int kind0 = feats[0]->sh.kind;
        if ((kind0 == DIST_BNB || kind0 == DIST_GP) && c.n_other) return false;
        // (k_vs_apply's LDS follows the batch's own bound on the group
        // count, not the run's: the device lays it out by the true count)
        const size_t lds_sort =
            ((size_t)batch_k_limit * 2 + kVsApplyBlock / 64 + 4 * kVsApplyRows
             + 4) * 4;
        if (lds_sort > 144 * 1024) return false;
        return (size_t)c.n_chunks * K() <= ((size_t)1 << 22);
    }
    // k_vs_tables for the open batch; afterwards the OUT buffers are the live
    // ones and P points at them
    template <int KIND>
    void launch_tables(SweepParams & P, const VsTables & T, VsCache & c) {
        Slave & f = *feats[0];
        TablesParams A;
        memset(&A, 0, sizeof(A));
        A.i0_in = f.i0.p;
        A.i1_in = f.i1.p;
        A.counts_in = py.d_counts.p;
        A.snap_in = snap_counts.p;
        A.dev_in = dev_ptr();
        std::swap(f.i0.p, alt_i0.p);   std::swap(f.i0.cap, alt_i0.cap);
        std::swap(f.i1.p, alt_i1.p);   std::swap(f.i1.cap, alt_i1.cap);
        std::swap(py.d_counts.p, alt_counts.p);
        std::swap(py.d_counts.cap, alt_counts.cap);
        std::swap(snap_counts.p, alt_snap.p);
        std::swap(snap_counts.cap, alt_snap.cap);
        dev_cur ^= 1;
        A.feat = f.view();
        A.counts_out = py.d_counts.p;
        A.snap_out = snap_counts.p;
        A.dev_out = dev_ptr();
        A.shifted = py.d_shifted.p;
        A.base = base.p;
        A.base_single = base_single.p;
        A.scalars = scalars.p;
        A.p2g = d_maps.p;
        A.g2p = reinterpret_cast<int32_t *>(d_maps.p + maps_pcap);
        A.alpha = alpha;
        A.d = d;
        A.n_empty = py.n_empty;
        A.sample_size = py.sample_size;
        A.offsets = VsOffsets{c.grp_off.p, c.off_epoch.p, c.off_stride};
        remap_log.reserve((size_t)kRemapEpochs * kRemapEntry, 0);   // zeros
        A.remap_log = remap_log.p;
        A.k_limit = T.Kuse;
        const size_t lds = ((size_t)T.Kpad * 4 + 2) * 4;
        int device = 0;
        HIP_CHECK(hipGetDevice(&device));
        static std::atomic<size_t> opted_in[64];
        std::atomic<size_t> & have = opted_in[device & 63];
        if (lds > 64 * 1024 && lds > have.load(std::memory_order_relaxed)) {
            HIP_CHECK(hipFuncSetAttribute(
                reinterpret_cast<const void *>(&k_vs_tables<KIND>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            have.store(lds, std::memory_order_relaxed);
        }
        hipLaunchKernelGGL((k_vs_tables<KIND>), dim3(T.n_values),
                           dim3(kTablesBlock), lds, stream(), A, T);
        HIP_CHECK(hipGetLastError());
        finish_pending = false;   // (this launch normalised the group set)
        base_valid = true;
        P.feat[0] = f.view();
        P.counts = py.d_counts.p;
        P.dev = dev_ptr();
        fused_batches += 1;
    }

    // Tables (k_vs_prepare) pay when several tiles share a value's vector;
    // with a tile or so per value the tile builds the vector itself.
    bool use_stream(const VsCache & c) const {
        if (value_stream_mode == 0) return false;
        if (value_stream_mode == 2) return true;
        // ... and where the tables are large: small ones are built in a few
        // microseconds and serve every wave from the scalar cache or LDS,
        // while a streaming tile evaluates K scores and exponentials per pass
        // (the group count the run began with: K() is an inflated bound while
        // a device-normalised run is open)
        const size_t groups = async_active ? (size_t)async_K0 : (size_t)K();
        if ((size_t)vs_nvals() * groups < ((size_t)1 << 21)) return false;
        return (size_t)c.n_tiles * 2 <= (size_t)c.n_values_present * 3;
    }
    struct VsStreamLaunch {
        Gibbs * self;
        SweepParams * P;
        VsCache * c;
        template <int KIND>
        void go() {
            const uint32_t per = kVsStreamBlock / 64;
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, stream(),
                               self->deferred_count.p, c->n_other);
            self->mark(self->ev0);
            // two rows of K likelihoods per tile (either class's vector)
            // between the total's pass and the scan (up to 2 GiB of scratch; beyond that they are computed
            // again, as before)
            const size_t stride = ((size_t)self->K() + 63) & ~(size_t)63;
            float * scratch = nullptr;
            if (self->stream_scratch_mode
                && (size_t)c->n_tiles * 2 * stride <= ((size_t)1 << 29)) {
                self->vsScratch.reserve(
                    grow_capacity((size_t)c->n_tiles * 2 * stride), 0);
                scratch = self->vsScratch.p;
            }
            if (c->n_tiles)
                hipLaunchKernelGGL((k_vs_stream<KIND>),
                                   dim3((c->n_tiles + per - 1) / per),
                                   dim3(kVsStreamBlock), 0, stream(), *P,
                                   c->tiles.p, c->n_tiles, c->sorted_rows.p,
                                   self->deferred.p, self->deferred_count.p,
                                   scratch, (uint32_t)stride,
                                   self->sample_prio_mode);
            HIP_CHECK(hipGetLastError());
            self->mark(self->ev1);
        }
    };
    void sample_value_stream(SweepParams & P, VsCache & c) {
        const size_t n = P.row_end - P.row_begin;
        deferred.reserve(std::max<size_t>(n, 1), 0);
        deferred_count.reserve(1, 0);
        prepare(P, false);
        if (c.n_other)
            HIP_CHECK(hipMemcpyAsync(deferred.p, c.other_pos.p,
                                     4 * (size_t)c.n_other,
                                     hipMemcpyDeviceToDevice, stream()));
        P.sorted_rows = c.sorted_rows.p;
        P.assign_pos = c.assign_pos.p;
        last_bands = last_prefix = false;
        stream_batches += 1;
        VsStreamLaunch L{this, &P, &c};
        {
        HOST_PROBE(7, "    launches (L.go)");
        switch (feats[0]->sh.kind) {
        case DIST_DD: L.go<DIST_DD>(); break;
        case DIST_DPD: L.go<DIST_DPD>(); break;
        case DIST_GP: L.go<DIST_GP>(); break;
        case DIST_BNB: L.go<DIST_BNB>(); break;
        default: L.go<DIST_BB>(); break;
        }
        }
        launch_deferred(P);
    }
    // the handed-over rows, by the wave-per-row kernel (or, when its strip of
    // LDS cannot hold K entries, the lane-per-row kernel) in list mode
    void launch_deferred(SweepParams & P) {
        SweepParams Q = P;
        Q.row_list = deferred.p;
        Q.row_list_count = deferred_count.p;
        if (wave_rows_fit()) {
            WaveRowsLaunch D{&Q, K(), 256};
            dispatch(D);
        } else {
            DeferredLaunch D{&Q};
            dispatch(D);
        }
    }

    // scan sampling on the value-sorted path (kernels.h, k_vs_scan_*)
    struct VsScanLaunch {
        Gibbs * self;
        SweepParams * P;
        VsCache * c;
        VsScanTables T;
        template <int KIND>
        void go() {
            const size_t n = P->row_end - P->row_begin;
            self->mark(self->ev0);
            hipLaunchKernelGGL((k_vs_scan_prepare<KIND>), dim3(T.n_values),
                               dim3(kVsScanBlock),
                               T.lds_scores ? (size_t)T.Kpad * 4 : 0,
                               stream(), *P, T, self->deferred_count.p,
                               c->n_other);
            self->phase_mark(1);
            hipLaunchKernelGGL((k_vs_scan_rows<KIND>), grid_for(n),
                               dim3(kBlock), 0, stream(), *P, T,
                               c->sorted_rows.p, n, self->deferred.p,
                               self->deferred_count.p);
            HIP_CHECK(hipGetLastError());
            self->mark(self->ev1);
        }
    };
    uint64_t scan_batches = 0;
    void sample_value_scan(SweepParams & P, VsCache & c) {
        const size_t n = P.row_end - P.row_begin;
        const uint32_t nv = (uint32_t)vs_nvals();
        const int Kpad = (K() + kVsScanCoarse - 1) / kVsScanCoarse
                         * kVsScanCoarse;
        vsLA.reserve(grow_capacity((size_t)nv * Kpad), 0);
        vsPA.reserve(grow_capacity((size_t)nv * (Kpad / kVsScanCoarse)), 0);
        vsM.reserve(nv, 0);
        vsmB.reserve(nv, 0);
        deferred.reserve(std::max<size_t>(n, 1), 0);
        deferred_count.reserve(1, 0);
        prepare(P, false);
        if (c.n_other)
            HIP_CHECK(hipMemcpyAsync(deferred.p, c.other_pos.p,
                                     4 * (size_t)c.n_other,
                                     hipMemcpyDeviceToDevice, stream()));
        P.sorted_rows = c.sorted_rows.p;
        P.assign_pos = c.assign_pos.p;
        VsScanLaunch L{this, &P, &c,
                       VsScanTables{vsLA.p, vsPA.p, vsM.p, vsmB.p, Kpad, nv,
                                    (size_t)Kpad * 4 <= 48 * 1024 ? 1 : 0}};
        switch (feats[0]->sh.kind) {
        case DIST_DD: L.go<DIST_DD>(); break;
        case DIST_DPD: L.go<DIST_DPD>(); break;
        case DIST_GP: L.go<DIST_GP>(); break;
        case DIST_BNB: L.go<DIST_BNB>(); break;
        default: L.go<DIST_BB>(); break;
        }
        scan_batches += 1;
        phase_mark(2);
        launch_deferred(P);
        phase_mark(3);
    }

    void sample_value_sorted(SweepParams & P) {
        VsCache * cp;
        {
            HOST_PROBE(6, "    vs_get");
            cp = &vs_get(P.row_begin, P.row_end);
        }
        VsCache & c = *cp;
        phase_mark(0);
        const int Kpad = (K() + kVsUnroll - 1) / kVsUnroll * kVsUnroll;
        batch_k_limit = k_limit();
        const bool fused = sampling_mode != 1 && !use_stream(c)
                           && fused_ok(c, Kpad);
        batch_fused = fused;
        // the last batch left its group set to a fused launch: any other
        // path wants it normalised and the caches rebuilt first
        if (!fused) run_pending_finish();
        if (sampling_mode == 1) return sample_value_scan(P, c);
        if (use_stream(c)) return sample_value_stream(P, c);
        const size_t n = P.row_end - P.row_begin;
        const uint32_t nv = (uint32_t)vs_nvals();
        HOST_PROBE(13, "    after vs_get");
        // (headroom: K creeps up by a group per batch; no realloc per step)
        vsLA.reserve(grow_capacity((size_t)nv * Kpad), 0);
        vsLB.reserve(grow_capacity((size_t)nv * Kpad), 0);
        vsM.reserve(nv, 0);
        vsmB.reserve(nv, 0);
        vsArg.reserve(nv, 0);
        // (what this launch walks and keeps in LDS: the group count's bound
        // at this batch; Kpad, the run's, is the tables' row stride)
        const int Kuse =
            fused ? std::min(Kpad, (k_limit() + kVsUnroll - 1) / kVsUnroll
                                       * kVsUnroll)
                  : Kpad;
        // chunk-boundary running sums: worth their serial pass in
        // k_vs_prepare once the sampling kernel is throughput-bound
        const bool narrow = use_narrow(c, Kuse);
        narrow_batches += narrow ? 1 : 0;
        const bool large = !narrow
                           && c.n_tiles >= (uint32_t)running_sums_min_tiles;
        const bool prefix = large && Kpad <= 8192;
        // the arg-max group's rows get a tile of their own per value when the
        // launch is large (and group-sorted: k_vs_apply's LDS sort fits)
        // ... and their workgroups do not push the launch past what is
        // resident at once (two 1024-thread workgroups per CU at 8 waves/SIMD;
        // a band workgroup is gone in about half the time of a regular one)
        // (fused: a band per apply CHUNK, read from the offsets the chunk's
        // sort left -- no walk; else a band per value, found by
        // k_vs_prepare's walk over the value's rows)
        const uint32_t band_count = fused ? c.n_table_chunks : nv;
        const uint32_t per_wg = kVsSampleBlock / 64;
        const uint32_t tile_wgs = (c.n_tiles + per_wg - 1) / per_wg;
        const uint32_t band_wgs = (band_count + per_wg - 1) / per_wg;
        const uint32_t slots = 2u * (uint32_t)cu_count();
        const bool bands = large && (fused || n / nv <= kVsBandWalkRows)
                           && !(fused && !c.one_value_chunks)
                           && (tile_wgs + band_wgs / 2 <= slots
                               || tile_wgs > slots);
        if (bands) {
            vsBandMode.reserve(band_count, 0);
            vsBandTile.reserve(band_count, 0);
        }
        last_bands = bands;
        last_prefix = prefix;
        band_batches += bands ? 1 : 0;
        prefix_batches += prefix ? 1 : 0;
        if (prefix) {
            vsPA.reserve(grow_capacity((size_t)nv * (Kpad / kVsUnroll)), 0);
            vsPB.reserve(grow_capacity((size_t)nv * (Kpad / kVsUnroll)), 0);
        }
        if (fused) vsOwn.reserve(grow_capacity((size_t)nv * Kpad), 0);
        deferred.reserve(std::max<size_t>(n, 1), 0);
        deferred_count.reserve(1, 0);
        // base[], base_single[] and the scalars; the few handed-over rows
        // do not pay for a gather table (fused: k_vs_tables writes them, and
        // rows beyond the tables are their chunk's to sample, k_vs_apply)
        if (!fused) {
            prepare(P, false);
            if (c.n_other)
                HIP_CHECK(hipMemcpyAsync(deferred.p, c.other_pos.p,
                                         4 * (size_t)c.n_other,
                                         hipMemcpyDeviceToDevice, stream()));
        }
        P.sorted_rows = c.sorted_rows.p;
        P.assign_pos = c.assign_pos.p;
        VsLaunch L{this, &P, &c,
                   VsTables{vsLA.p, vsLB.p, vsM.p, vsmB.p, vsArg.p, Kpad,
                            prefix ? vsPA.p : nullptr,
                            prefix ? vsPB.p : nullptr,
                            bands ? vsBandMode.p : nullptr,
                            bands ? vsBandTile.p : nullptr, c.val_start.p,
                            nv, nullptr, c.chunk_first.p,
                            fused ? vsOwn.p : nullptr, Kuse, band_count,
                            fused ? 1 : 0, sample_prio_mode}, narrow, fused};
        // DIST_VS_STAMPS=<file>: per-wave phase stamps of every launch (the
        // last one stays in the file): tools/vs_stamps.py
        static const char * stamps_path = getenv("DIST_VS_STAMPS");
        if (stamps_path) {
            vsStamps.reserve(((size_t)c.n_tiles * 2 + 4096) * 6, 0);
            L.T.stamps = vsStamps.p;
        }
        switch (feats[0]->sh.kind) {
        case DIST_DD: L.go<DIST_DD>(); break;
        case DIST_DPD: L.go<DIST_DPD>(); break;
        case DIST_GP: L.go<DIST_GP>(); break;
        case DIST_BNB: L.go<DIST_BNB>(); break;
        default: L.go<DIST_BB>(); break;
        }
        if (stamps_path) {
            sync();
            std::vector<unsigned long long> h(((size_t)c.n_tiles * 2 + 4096) * 6);
            vsStamps.download(h.data(), h.size());
            if (FILE * f = fopen(stamps_path, "wb")) {
                fwrite(h.data(), 8, h.size(), f);
                fclose(f);
            }
        }
        phase_mark(2);
        if (!fused) launch_deferred(P);
        phase_mark(3);
    }
    struct DeferredLaunch {
        SweepParams * P;
        template <int A, int B, int NF>
        void run() {
            hipLaunchKernelGGL((k_sweep_sample<A, B, NF>), dim3(16),
                               dim3(kBlock), 0, stream(), *P);
            HIP_CHECK(hipGetLastError());
        }
    };
    // one wave per row (list mode or a small range)
    struct WaveRowsLaunch {
        SweepParams * P;
        int K;
        unsigned blocks;
        template <int A, int B, int NF>
        void run() {
            hipLaunchKernelGGL((k_rows_wave<A, B, NF>), dim3(blocks),
                               dim3(kBlock),
                               (size_t)(kBlock / 64) * ((K + 63) & ~63)
                                   * sizeof(float),
                               stream(), *P);
            HIP_CHECK(hipGetLastError());
        }
    };
    // the wave-per-row kernel keeps K floats per wave in LDS
    bool wave_rows_fit() const {
        return (size_t)(kBlock / 64) * ((K() + 63) & ~63) * sizeof(float)
               <= 60 * 1024;
    }

    void batch_sample(size_t r0, size_t r1, uint32_t seed, uint64_t draw_base) {
        DIST_REQUIRE(!batch_open, "previous batch not finished");
        DIST_REQUIRE(r0 <= r1 && r1 <= n_rows, "bad row range");
        DIST_REQUIRE(r1 <= assigned_rows || r0 == r1,
                     "rows without a group yet: init_sequential first");
        batch_begin = r0;
        batch_end = r1;
        batch_serial += 1;
        batch_seed = seed;
        batch_draw_base = draw_base;
        batch_open = true;
        batch_value_sorted = false;
        moves_in_row_order = false;
        if (r0 == r1) {
            // A rank whose shard is exhausted takes part with empty batches:
            // its peers' deltas arrive in slots numbered AFTER the last
            // batch's normalisation, so whatever a fused batch left to "the
            // next k_vs_tables" is normalised here and now -- there is no
            // such launch for an empty batch -- and this batch closes with
            // the separate kernels (found by tools/fuzz_ranks.py: ragged
            // value-partitioned shards under group churn diverged).
            run_pending_finish();
            batch_fused = false;
            return;
        }
        timing_this_batch =
            kernel_timing > 0 && timing_tick++ % (uint64_t)kernel_timing == 0;
        // (a host-driven batch: whatever sharded run was closed before it is
        // not taken up again)
        if (!async_active) resume_bound = resume_left = 0;
        upload_maps();
        SweepParams P;
        {
            HOST_PROBE(3, "  params()");
            P = params(r0, r1, seed, draw_base);
        }
        batch_value_sorted = use_value_sorted(r1 - r0);
        batch_fused = false;
        if (!batch_value_sorted) run_pending_finish();
        {
            HOST_PROBE(4, "  drop_overlapping");
            drop_overlapping_caches(r0, r1, batch_value_sorted);
        }
        if (!batch_value_sorted) flush_assign_pos();
        if (batch_value_sorted) {
            HOST_PROBE(5, "  sample_value_sorted");
            sample_value_sorted(P);
            vs_batches += 1;
        } else if (r1 - r0 <= 2048 && wave_rows_fit()) {
            // a handful of rows (the sequential chain is one): a wave each
            prepare(P, false);
            mark(ev0);
            WaveRowsLaunch L{&P, K(),
                             (unsigned)((r1 - r0 + kBlock / 64 - 1)
                                        / (kBlock / 64))};
            dispatch(L);
            mark(ev1);
            generic_batches += 1;
        } else if ((uses_runtime_kernel() || (program_all && F() >= 1))
                   && sample_by_program(P)) {
            generic_batches += 1;
        } else {
            prepare(P);
            SampleLaunch L{this, &P};
            dispatch(L);
            generic_batches += 1;
        }
        timing_pending = true;
    }
    // the events of the open batch are read once the batch has been drained
    // (batch_finish has synchronised by then): no extra host sync per batch
    void collect_timing() {
        if (!timing_pending) return;
        timing_pending = false;
        if (!timing_this_batch) return;
        HIP_CHECK(hipEventSynchronize(ev1));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
        kernel_ms += ms;
        kernel_launches += 1;
        kernel_rows += batch_end - batch_begin;
        static const bool trace_deferred = getenv("DIST_TRACE_DEFERRED");
        if (batch_value_sorted && trace_deferred) {
            uint32_t n = 0;
            deferred_count.download(&n, 1);
            fprintf(stderr, "[dist] batch [%zu, %zu): %u rows handed over\n",
                    batch_begin, batch_end, n);
        }
    }

    bool any_float_stats() const {
        for (auto & f : feats)
            if (has_float_stats(f->sh.kind)) return true;
        return false;
    }
    // Float statistics (NICH count/mean/ctv, GP log_prod) depend on update
    // order, so they are replayed per group in row order: the batch's events
    // are sorted stably by group, then one wave per group walks its segment.
    // old_dev == nullptr: rows are only added (initial load).
    void replay_sorted(const uint32_t * old_dev, const uint32_t * new_dev,
                       size_t row_begin, size_t n_rows,
                       const uint32_t * const * vals = nullptr) {
        if (!any_float_stats() || !n_rows) return;
        if (!vals) vals = values.data();
        const size_t n_ev = old_dev ? 2 * n_rows : n_rows;
        const size_t Kn = (size_t)K();
        ReplayFeatures R;
        R.n = 0;
        for (int f = 0; f < F(); ++f) {
            if (!has_float_stats(feats[f]->sh.kind)) continue;
            DIST_REQUIRE(vals[f], "replay: no values for an ordered feature");
            R.s[R.n] = feats[f]->view();
            R.values[R.n] = vals[f];
            R.n += 1;
        }
        if (Kn + 1 <= (size_t)kCsMaxKeys && n_ev < ((size_t)1 << 31)) {
            // the events sorted stably by group: histogram, scan, scatter
            // (kernels.h, k_cs_*), straight from the moves
            const int n_keys = (int)Kn + 1;
            const int blocks = (int)((n_ev + kCsEvents - 1) / kCsEvents);
            cs_hist.reserve(grow_capacity((size_t)blocks * n_keys), 0);
            cs_base.reserve(grow_capacity((size_t)n_keys + 1), 0);
            cs_total.reserve(grow_capacity((size_t)n_keys), 0);
            ev_vals_sorted.reserve(n_ev, 0);
            hipLaunchKernelGGL(k_cs_hist, dim3(blocks), dim3(kCsBlock),
                               (size_t)n_keys * 4, stream(), old_dev, new_dev,
                               n_ev, n_keys, cs_hist.p);
            hipLaunchKernelGGL(k_cs_scan,
                               dim3((n_keys + kCsBlock - 1) / kCsBlock),
                               dim3(kCsBlock), 0, stream(), cs_hist.p, blocks,
                               n_keys, cs_total.p);
            const size_t lds = (size_t)(kCsBlock / 64 + 1) * n_keys * 4;
            int device = 0;
            HIP_CHECK(hipGetDevice(&device));
            static std::atomic<size_t> opted_in[64];
            std::atomic<size_t> & have = opted_in[device & 63];
            if (lds > 64 * 1024 && lds > have.load(std::memory_order_relaxed)) {
                HIP_CHECK(hipFuncSetAttribute(
                    reinterpret_cast<const void *>(&k_cs_scatter),
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                have.store(lds, std::memory_order_relaxed);
            }
            hipLaunchKernelGGL(k_cs_scatter, dim3(blocks), dim3(kCsBlock), lds,
                               stream(), old_dev, new_dev, n_ev, n_keys,
                               cs_hist.p, cs_total.p, cs_base.p,
                               ev_vals_sorted.p);
            HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL(k_replay_sorted,
                               dim3((unsigned)Kn, (unsigned)R.n), dim3(64), 0,
                               stream(), R, row_begin, ev_vals_sorted.p,
                               cs_base.p, cs_base.p + 1);
            HIP_CHECK(hipGetLastError());
            return;
        }
        // (more groups than the scatter's LDS holds: the library radix sort)
        int bits = 1;   // keys 0..Kn (Kn = padding)
        while ((1ull << bits) < Kn + 1) bits += 1;
        ev_keys.reserve(n_ev, 0); ev_vals.reserve(n_ev, 0);
        ev_keys_sorted.reserve(n_ev, 0); ev_vals_sorted.reserve(n_ev, 0);
        // (begin | end in one buffer: one fill)
        seg_begin.reserve(grow_capacity(2 * (Kn + 1)), 0);
        uint32_t * const seg_end_p = seg_begin.p + (Kn + 1);
        const size_t tb = sort_pairs_temp_bytes(n_ev, bits);
        sort_temp.reserve(tb + 256, 0);
        LAUNCH(k_replay_events, n_rows, old_dev, new_dev, n_rows,
               (uint32_t)Kn, ev_keys.p, ev_vals.p);
        sort_pairs(sort_temp.p, tb, ev_keys.p, ev_keys_sorted.p, ev_vals.p,
                   ev_vals_sorted.p, n_ev, bits, stream());
        HIP_CHECK(hipMemsetAsync(seg_begin.p, 0, 2 * (Kn + 1) * 4, stream()));
        LAUNCH(k_replay_bounds, n_ev, ev_keys_sorted.p, n_ev, seg_begin.p,
               seg_end_p);
        hipLaunchKernelGGL(k_replay_sorted, dim3((unsigned)Kn, (unsigned)R.n),
                           dim3(64), 0, stream(), R, row_begin,
                           ev_vals_sorted.p, seg_begin.p, seg_end_p);
        HIP_CHECK(hipGetLastError());
    }
    // Multi-rank exchange of the order-dependent statistics: the moves of the
    // open batch in row order (slot indices of the batch snapshot) ...
    void batch_moves(uint32_t * old_dev, uint32_t * new_dev) {
        DIST_REQUIRE(batch_open, "no open batch");
        const size_t n = batch_end - batch_begin;
        if (!n) return;
        const uint32_t * o = old_packed.p, * w = new_packed.p;
        if (batch_value_sorted) {
            DIST_REQUIRE(any_float_stats() && moves_in_row_order,
                         "batch_moves: call after batch_delta, on a mixture "
                         "with order-dependent statistics");
            o = old_row.p; w = new_row.p;
        }
        HIP_CHECK(hipMemcpyAsync(old_dev, o, n * 4, hipMemcpyDeviceToDevice,
                                 stream()));
        HIP_CHECK(hipMemcpyAsync(new_dev, w, n * 4, hipMemcpyDeviceToDevice,
                                 stream()));
    }
    // ... and the replay of an event list gathered from all ranks in rank
    // (= global row) order.  old_dev == nullptr: additions only.
    void replay_ordered(const uint32_t * old_dev, const uint32_t * new_dev,
                        const uint32_t * const * vals, size_t n, bool reset) {
        if (reset) {
            for (int f = 0; f < F(); ++f) {
                if (!has_float_stats(feats[f]->sh.kind)) continue;
                LAUNCH(k_zero_ordered_stats, (size_t)K(), feats[f]->view(),
                       K());
            }
        }
        replay_sorted(old_dev, new_dev, 0, n, vals);
        if (reset && !batch_open) rebuild_caches();
    }
    // "float_stats": 0 the ordered replay (bit-identical to the sequential
    // chain's running updates; default), 1 merged: binary64 sums per group
    // (kernels.h, k_merge_float_*; tolerance-level)
    int float_stats_mode = 0;
    DeviceBuf<double> merge_stage, merge_image;
    uint64_t merged_batches = 0;
    MergeLayout merge_layout() const {
        MergeLayout L;
        memset(&L, 0, sizeof(L));
        L.F = F();
        L.K = K();
        int off = 0;
        for (int f = 0; f < F(); ++f) {
            const int kind = feats[f]->sh.kind;
            L.kind[f] = has_float_stats(kind) ? kind : -1;
            L.off[f] = off;
            if (kind == DIST_NICH) off += 3 * K();
            if (kind == DIST_GP) off += K();
        }
        L.words = off;
        return L;
    }
    bool merged_floats() const {
        return float_stats_mode == 1 && any_float_stats()
               && (size_t)merge_layout().words * 8 <= 144 * 1024;
    }
    // the open batch's float-statistic sums into merge_image (not applied)
    void merge_float_delta() {
        const size_t n = batch_end - batch_begin;
        const MergeLayout L = merge_layout();
        merge_image.reserve(grow_capacity((size_t)L.words), 0);
        if (!n) {
            HIP_CHECK(hipMemsetAsync(merge_image.p, 0, (size_t)L.words * 8,
                                     stream()));
            return;
        }
        const bool by_pos = batch_value_sorted;
        SweepParams P = params(batch_begin, batch_end, 0, 0);
        const unsigned blocks =
            (unsigned)((n + kApplyLdsRows - 1) / kApplyLdsRows);
        merge_stage.reserve(grow_capacity((size_t)blocks * L.words), 0);
        const size_t lds = (size_t)L.words * 8;
        int device = 0;
        HIP_CHECK(hipGetDevice(&device));
        static std::atomic<size_t> opted_in[64];
        std::atomic<size_t> & have = opted_in[device & 63];
        if (lds > 64 * 1024 && lds > have.load(std::memory_order_relaxed)) {
            HIP_CHECK(hipFuncSetAttribute(
                reinterpret_cast<const void *>(&k_merge_float_moves),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            have.store(lds, std::memory_order_relaxed);
        }
        hipLaunchKernelGGL(k_merge_float_moves, dim3(blocks),
                           dim3(kApplyLdsBlock), lds, stream(), P, L,
                           by_pos ? old_row.p : old_packed.p,
                           by_pos ? new_row.p : new_packed.p, merge_stage.p);
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(k_merge_float_reduce,
                           dim3((unsigned)((L.words + 63) / 64)), dim3(64), 0,
                           stream(), merge_stage.p, (int)blocks, L.words,
                           merge_image.p);
        HIP_CHECK(hipGetLastError());
    }
    void merge_float_apply(const double * image, bool reset = false) {
        const MergeLayout L = merge_layout();
        SweepParams P = params(batch_begin, batch_end, 0, 0);
        LAUNCH(k_merge_float_apply, (size_t)K(), P, L, image, reset ? 1 : 0);
        merged_batches += reset ? 0 : 1;
    }
    void replay_floats() {
        if (merged_floats()) {
            merge_float_delta();
            merge_float_apply(merge_image.p);
            return;
        }
        // value-sorted batches hold the moves by position; apply_ints left a
        // row-ordered copy for the replay
        const bool by_pos = batch_value_sorted;
        replay_sorted(by_pos ? old_row.p : old_packed.p,
                      by_pos ? new_row.p : new_packed.p, batch_begin,
                      batch_end - batch_begin);
    }

    // integer statistics of the open batch into `img` (live arrays or a
    // zeroed delta image)
    void apply_ints(StatImage img) {
        const size_t n = batch_end - batch_begin;
        // (the batch's entropy too: a fused batch's chunks sample the rows
        // they were handed)
        SweepParams P;
        {
            HOST_PROBE(9, "  params() in apply");
            P = params(batch_begin, batch_end, batch_seed, batch_draw_base);
        }
        // (a fused batch: the bound its launches were sized with, see
        // fused_ok; the device's layout follows the true group count)
        const size_t k_lds = batch_fused ? (size_t)batch_k_limit : (size_t)K();
        const size_t lds_sort =
            (k_lds * 2 + kVsApplyBlock / 64 + 4 * kVsApplyRows + 4) * 4;
        const size_t lds_plain = (size_t)K() * 4;
        // a 1024-thread workgroup may take most of the CU's 160 KiB of LDS
        const size_t lds_limit = 144 * 1024;
        if (batch_value_sorted && lds_plain <= lds_limit) {
            VsCache & c = vs_get(batch_begin, batch_end);
            const bool sort = lds_sort <= lds_limit;
            const bool bb = feats[0]->sh.kind == DIST_BB;
            const bool gp = feats[0]->sh.kind == DIST_GP;
            const bool bnb = feats[0]->sh.kind == DIST_BNB;
            const dim3 grid(c.n_chunks), block(kVsApplyBlock);
            // one chunk per value and live statistics: the kernel keeps the
            // touched cache cells current and batch_finish skips the rebuild
            // (worth it when the full rebuild is the bigger cost: wide tables)
            const int refresh = (c.one_chunk_per_value && is_cat(
                feats[0]->sh.kind) && img.counts == py.d_counts.p
                && (size_t)K() * feats[0]->dim() > ((size_t)1 << 22)) ? 1 : 0;
            cells_fresh = refresh != 0;
            // float statistics replay in ROW order: un-sort the moves first
            // (before the kernel below permutes sorted_rows)
            if (any_float_stats()) {
                old_row.reserve(std::max<size_t>(n, 1), 0);
                new_row.reserve(std::max<size_t>(n, 1), 0);
                LAUNCH(k_pos_scatter, n, old_packed.p, c.sorted_rows.p,
                       old_row.p, n);
                LAUNCH(k_pos_scatter, n, new_packed.p, c.sorted_rows.p,
                       new_row.p, n);
                moves_in_row_order = true;
            }
            c.dirty = true;
            // the chunks' per-group deltas meet in a staging matrix instead
            // of in contended atomics (unless that matrix would be huge)
            const size_t stage_words = (size_t)c.n_chunks * K();
            int32_t * stage = nullptr;
            if (stage_words <= ((size_t)1 << 22)) {
                vs_stage.reserve(grow_capacity(stage_words), 0);
                stage = vs_stage.p;
            }
            const int sole = c.one_chunk_per_value ? 1 : 0;
            int apply_device = 0;
            HIP_CHECK(hipGetDevice(&apply_device));
            unsigned long long * pairs = nullptr;
            unsigned pairs_seq = 0;
            if (stage && img.counts == py.d_counts.p && !async_active) {
                pairs = pairs_buffer();   // live statistics: publish sizes
                pairs_seq = pairs_ticket = ++publish_ticket;
            }
            const dim3 rgrid((K() + kVsReduceGroups - 1) / kVsReduceGroups),
                rblock(kVsReduceGroups * kVsReduceSlices);
            // (fused batches: the chunk samples its handed-over rows itself)
            DIST_REQUIRE(!batch_fused || (sort && stage),
                         "internal: fused batch without the sorting apply");
            // (and, LDS permitting, a few of them in strips of their own
            // while the chunk's other waves add up the moves)
            int lds_told = 0;
            size_t lds_sort_used = lds_sort;
            if (batch_fused && sort && !gp && apply_overlap_mode) {
                const size_t strip_bytes = ((k_lds + 63) & ~(size_t)63) * 4;
                lds_sort_used = std::min(lds_limit, lds_sort + 4 * strip_bytes);
                lds_told = (int)lds_sort_used;
            }
            const VsDefer D{deferred.p, deferred_count.p,
                            batch_fused ? c.def_counts.p : nullptr,
                            c.chunks.p, lds_told};
            static const bool trace_deferred = getenv("DIST_TRACE_DEFERRED");
            if (trace_deferred && batch_fused) {   // (diagnostic: drains)
                std::vector<uint32_t> h(c.n_chunks);
                HIP_CHECK(hipStreamSynchronize(stream()));
                HIP_CHECK(hipMemcpy(h.data(), c.def_counts.p,
                                    h.size() * sizeof(uint32_t),
                                    hipMemcpyDeviceToHost));
                uint32_t total = 0, most = 0, chunks_with = 0;
                for (uint32_t v : h) {
                    total += v;
                    most = std::max(most, v);
                    chunks_with += v != 0;
                }
                fprintf(stderr, "[dist] batch [%zu, %zu): %u rows handed over "
                        "in %u of %u chunks, at most %u in one; %d bytes of "
                        "LDS at a bound of %d groups\n",
                        batch_begin, batch_end, total, chunks_with,
                        (unsigned)c.n_chunks, most, lds_told, (int)k_lds);
            }
            // the sorting form of a device-normalised run leaves the groups'
            // offsets per chunk (bands without a walk, k_vs_tables); any
            // other form clears the stamps of a range that has some
            VsOffsets O{nullptr, c.off_epoch.p, 0};
            if (sort && async_active && c.one_value_chunks) {
                if (c.off_stride < K() + 2) {
                    c.off_stride = (int)grow_capacity((size_t)K() + 2);
                    c.grp_off.release();
                    c.grp_off.reserve((size_t)c.n_chunks * c.off_stride, 0);
                    c.off_epoch.release();
                    c.off_epoch.reserve(std::max<size_t>(c.n_chunks, 1), 0);
                }
                O = VsOffsets{c.grp_off.p, c.off_epoch.p, c.off_stride};
            }
#define VS_APPLY(KIND, SORT, LDS)                                            \
            do {                                                             \
                /* beyond the default opt-in: raised (never lowered) once  \
                 * per size, kernel instance and device */                   \
                static std::atomic<size_t> opted_in[64];                     \
                std::atomic<size_t> & have = opted_in[apply_device & 63];    \
                if ((LDS) > 64 * 1024                                        \
                    && (LDS) > have.load(std::memory_order_relaxed)) {       \
                    HIP_CHECK(hipFuncSetAttribute(                           \
                        reinterpret_cast<const void *>(                      \
                            &k_vs_apply<KIND, SORT>),                        \
                        hipFuncAttributeMaxDynamicSharedMemorySize,          \
                        (int)(LDS)));                                        \
                    have.store((LDS), std::memory_order_relaxed);            \
                }                                                            \
                hipLaunchKernelGGL((k_vs_apply<KIND, SORT>), grid, block,    \
                                   LDS, stream(), P, img, c.chunks.p,        \
                                   c.sorted_rows.p, d_p2g_ptr,               \
                                   c.assign_pos.p, (uint32_t)vs_nvals(),     \
                                   refresh, sole, stage, D, O);              \
                if (stage)                                                   \
                    hipLaunchKernelGGL((k_vs_reduce<KIND>), rgrid, rblock,   \
                                       0, stream(), img, stage, c.chunks.p,  \
                                       c.n_chunks, K(),                      \
                                       (uint32_t)vs_nvals(), pairs,          \
                                       pairs_seq,                            \
                                       async_active ? dev_ptr() : nullptr,   \
                                       k_limit(),                            \
                                       batch_fused ? c.multi.p : nullptr,    \
                                       batch_fused ? c.n_multi : 0u,         \
                                       feats[0]->dim());                     \
            } while (0)
            HOST_PROBE(12, "  apply launches");
            // chunks of several values first (their rows of the staging
            // matrix must be there when k_vs_reduce runs)
            if (c.mixed_chunks) {   // (categorical kinds only, see vs_get)
                const size_t lds = lds_plain;
                auto mixed = feats[0]->sh.kind == DIST_DPD
                                 ? &k_vs_apply_mixed<DIST_DPD>
                                 : &k_vs_apply_mixed<DIST_DD>;
                static std::atomic<size_t> opted_in[2][64];
                std::atomic<size_t> & have =
                    opted_in[feats[0]->sh.kind == DIST_DPD][apply_device & 63];
                if (lds > 64 * 1024
                    && lds > have.load(std::memory_order_relaxed)) {
                    HIP_CHECK(hipFuncSetAttribute(
                        reinterpret_cast<const void *>(mixed),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    have.store(lds, std::memory_order_relaxed);
                }
                hipLaunchKernelGGL(mixed, grid, block, lds, stream(), P, img,
                                   c.chunks.p, c.sorted_rows.p, d_p2g_ptr,
                                   c.assign_pos.p, refresh, stage);
                HIP_CHECK(hipGetLastError());
            }
            if (bb && sort) VS_APPLY(DIST_BB, true, lds_sort_used);
            else if (bb) VS_APPLY(DIST_BB, false, lds_plain);
            else if (gp && sort) VS_APPLY(DIST_GP, true, lds_sort_used);
            else if (gp) VS_APPLY(DIST_GP, false, lds_plain);
            else if (bnb && sort) VS_APPLY(DIST_BNB, true, lds_sort_used);
            else if (bnb) VS_APPLY(DIST_BNB, false, lds_plain);
            else if (sort && feats[0]->sh.kind == DIST_DPD)
                VS_APPLY(DIST_DPD, true, lds_sort_used);   // (its rows' scorer)
            else if (sort) VS_APPLY(DIST_DD, true, lds_sort_used);
            else VS_APPLY(DIST_DD, false, lds_plain);
#undef VS_APPLY
            HIP_CHECK(hipGetLastError());
        } else if (batch_value_sorted) {
            // group count too large for the LDS-aggregated kernel: un-sort
            // the moves and take the direct-atomics kernel
            VsCache & c = vs_get(batch_begin, batch_end);
            flush_assign_pos();
            old_row.reserve(std::max<size_t>(n, 1), 0);
            new_row.reserve(std::max<size_t>(n, 1), 0);
            LAUNCH(k_pos_scatter, n, old_packed.p, c.sorted_rows.p, old_row.p, n);
            LAUNCH(k_pos_scatter, n, new_packed.p, c.sorted_rows.p, new_row.p, n);
            moves_in_row_order = true;
            P.old_packed = old_row.p;
            P.new_packed = new_row.p;
            apply_moves(n, P, img, d_p2g_ptr, assign);
            // assign[] is now current: refresh the position copy
            LAUNCH(k_pos_gather, n, assign + batch_begin, c.sorted_rows.p,
                   c.assign_pos.p, n);
        } else {
            apply_moves(n, P, img, d_p2g_ptr, assign);
        }
    }
    // the general rows' integer statistics: per-group totals through LDS
    // where (1 + 2 F) K integers fit a workgroup's, else direct atomics
    uint64_t staged_applies = 0;
    int apply_stage_mode = 1;   // 0: never the all-in-LDS form
    void apply_moves(size_t n, const SweepParams & P, const StatImage & img,
                     const uint32_t * p2g, uint32_t * assign_out) {
        // the whole integer image in LDS (small categoricals): no global
        // atomics, the workgroups' images meet in a staging matrix
        const size_t all_words = stat_words();
        if (apply_stage_mode && all_words * sizeof(int) <= 144 * 1024
            && n >= (size_t)4 * K()) {
            bool cat = false;
            for (auto & f : feats) cat = cat || is_cat(f->sh.kind);
            if (cat) {
                StageLayout L;
                memset(&L, 0, sizeof(L));
                L.K = K();
                int off = K();
                for (int f = 0; f < F(); ++f) {
                    L.off_i0[f] = off; off += K();
                    L.off_i1[f] = off; off += K();
                    L.off_cnt[f] = off; off += K() * feats[f]->dim();
                    L.dim[f] = feats[f]->dim();
                }
                L.words = off;
                const unsigned blocks =
                    (unsigned)((n + kApplyLdsRows - 1) / kApplyLdsRows);
                vs_stage.reserve(grow_capacity((size_t)blocks * L.words), 0);
                const size_t lds_all = (size_t)L.words * sizeof(int);
                int device = 0;
                HIP_CHECK(hipGetDevice(&device));
                static std::atomic<size_t> opted_in[64];
                std::atomic<size_t> & have = opted_in[device & 63];
                if (lds_all > 64 * 1024
                    && lds_all > have.load(std::memory_order_relaxed)) {
                    HIP_CHECK(hipFuncSetAttribute(
                        reinterpret_cast<const void *>(&k_apply_moves_stage),
                        hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds_all));
                    have.store(lds_all, std::memory_order_relaxed);
                }
                hipLaunchKernelGGL(k_apply_moves_stage, dim3(blocks),
                                   dim3(kApplyLdsBlock), lds_all, stream(), P,
                                   L, vs_stage.p, p2g, assign_out);
                HIP_CHECK(hipGetLastError());
                WordSegments seg;
                memset(&seg, 0, sizeof(seg));
                size_t end = 0;
                auto push = [&](int32_t * dst, size_t words) {
                    end += words;
                    seg.dst[seg.n] = dst;
                    seg.end[seg.n] = end;
                    seg.n += 1;
                };
                push(img.counts, (size_t)K());
                for (int f = 0; f < F(); ++f) {
                    push(img.i0[f], (size_t)K());
                    push(img.i1[f], (size_t)K());
                    // (a zero-length segment never matches a word)
                    push(img.cnt[f], (size_t)K() * feats[f]->dim());
                }
                hipLaunchKernelGGL(k_stage_reduce,
                                   dim3((unsigned)((L.words + 63) / 64)),
                                   dim3(64), 0, stream(), seg, vs_stage.p,
                                   (int)blocks, L.words);
                HIP_CHECK(hipGetLastError());
                staged_applies += 1;
                return;
            }
        }
        const size_t lds = (size_t)(1 + 2 * F()) * K() * sizeof(int);
        if (lds > 64 * 1024 || n < (size_t)4 * K()) {   // (too few to pay)
            LAUNCH(k_apply_moves, n, P, img, p2g, assign_out);
            return;
        }
        hipLaunchKernelGGL(
            k_apply_moves_lds,
            dim3((unsigned)((n + kApplyLdsRows - 1) / kApplyLdsRows)),
            dim3(kApplyLdsBlock), lds, stream(), P, img, p2g, assign_out);
        HIP_CHECK(hipGetLastError());
    }
    void batch_apply_local() {
        DIST_REQUIRE(batch_open, "no open batch");
        if (batch_end == batch_begin) return;
        apply_ints(live_image());
        if (any_float_stats()) publish_counts_early();
        replay_floats();
    }
    // zeroed = the caller's image is all zero already (see clear below)
    void batch_delta(int32_t * delta_dev, bool zeroed = false) {
        DIST_REQUIRE(batch_open, "no open batch");
        if (!zeroed)
            HIP_CHECK(hipMemsetAsync(delta_dev, 0, stat_words() * 4, stream()));
        if (batch_end == batch_begin) return;
        apply_ints(word_image(delta_dev));
    }
    // the sharded loop's exchange image: the words of `k` groups (exchange_K)
    // behind the header; value-partitioned ranks leave the cells out -- they
    // change their own, in place
    StatImage exchange_image(int32_t * words, size_t k) {
        StatImage img = word_image(words, k);
        if (value_partitioned) img.cnt[0] = feats[0]->cnt.p;
        return img;
    }
    size_t exchange_words(size_t k) const {
        return value_partitioned ? 3 * k : stat_words(k);
    }
    void batch_delta_exchange(int32_t * words, size_t k) {
        DIST_REQUIRE(batch_open, "no open batch");
        if (value_partitioned) cells_partial = true;
        if (batch_end == batch_begin) return;
        apply_ints(exchange_image(words, k));
    }
    unsigned long long * pairs_buffer() {
        if ((size_t)K() > pinned_pairs_cap) {
            if (pinned_pairs) (void)hipHostFree(pinned_pairs);
            pinned_pairs_cap = grow_capacity((size_t)K());
            HIP_CHECK(hipHostMalloc((void **)&pinned_pairs,
                                    pinned_pairs_cap * 8,
                                    hipHostMallocCoherent));
            memset(pinned_pairs, 0, pinned_pairs_cap * 8);
        }
        return pinned_pairs;
    }
    // clear = leave the image zeroed behind (it is not const then)
    void batch_apply_delta(int32_t * delta_dev, bool clear = false) {
        CommCheck none;
        memset(&none, 0, sizeof(none));
        batch_apply_words(delta_dev, clear, (size_t)K(), true, none);
    }
    // k: the group count the image is laid out for; cells: whether it holds
    // the categorical cells
    void batch_apply_words(int32_t * delta_dev, bool clear, size_t k,
                           bool cells, const CommCheck & chk) {
        DIST_REQUIRE(batch_open, "no open batch");
        StatImage a = live_image();
        WordSegments seg;
        memset(&seg, 0, sizeof(seg));
        size_t off = 0;
        auto push = [&](int32_t * dst, size_t n) {
            if (!n) return;
            off += n;
            seg.dst[seg.n] = dst;
            seg.end[seg.n] = off;
            seg.n += 1;
        };
        push(a.counts, k);
        for (int f = 0; f < F(); ++f) {
            push(a.i0[f], k);
            push(a.i1[f], k);
            if (cells) push(a.cnt[f], k * feats[f]->dim());
        }
        unsigned long long * pairs = nullptr;
        if (!async_active) {   // the host will want the new group sizes
            pairs = pairs_buffer();
            pairs_ticket = ++publish_ticket;
        }
        early_ticket = 0;   // (sizes published before this are stale now)
        LAUNCH(k_add_words, off, seg, delta_dev, off, clear ? 1 : 0, pairs,
               pairs_ticket, chk);
        // the order-dependent statistics (NICH, GP log_prod) are not in the
        // image: the caller gathers the moves and calls replay_ordered
    }

    // Normalise the group set after a batch (DESIGN.md "Batch semantics"):
    // groups that lost their last member are swap-removed in descending slot
    // order; one empty group is appended per previously empty group that
    // gained members; caches are rebuilt from the statistics.
    void batch_finish() {
        DIST_REQUIRE(batch_open, "no open batch");
        batch_open = false;
        const std::vector<int> snap = py.counts;
        const int K0 = K();
        {
            HOST_PROBE(17, "  finish: refresh_host_counts");
            refresh_host_counts();
        }
        int created = 0;
        for (int k = 0; k < K0; ++k)
            if (snap[k] == 0 && py.counts[k] > 0) created += 1;
        bool structural = created > 0;
        // swap-removals in descending slot order, simulated on the host:
        // content[i] = original slot of the group that ends up in slot i
        std::vector<int> content((size_t)K0);
        for (int k = 0; k < K0; ++k) content[k] = k;
        int size = K0;
        for (int k = K0 - 1; k >= 0; --k) {
            if (snap[k] > 0 && py.counts[k] == 0) {
                structural = true;
                const int last = size - 1;
                if (k != last) {
                    content[k] = content[last];
                    py.counts[k] = py.counts[last];
                }
                size -= 1;
                py.counts.pop_back();
                tracker.remove_group((uint32_t)k);
            }
        }
        std::vector<int2> moves;
        for (int i = 0; i < size; ++i)
            if (content[i] != i) moves.push_back(int2{i, content[i]});
        if (!moves.empty()) {
            // sources lie in [size, K0), destinations in [0, size): one
            // launch per object copies them all
            struct_moves.upload(moves.data(), moves.size());
            const int nm = (int)moves.size();
            LAUNCH(k_py_move_groups, (size_t)nm, py.d_counts.p, py.d_shifted.p,
                   struct_moves.p, nm);
            for (auto & s : feats) {
                const size_t width = std::max(1, s->dim());
                LAUNCH(k_slave_move_groups, (size_t)nm * width, s->view(),
                       struct_moves.p, nm);
            }
        }
        for (auto & s : feats) s->K = size;
        const int k_new = size;
        const uint32_t first_new_global = (uint32_t)tracker.g2p.size();
        // groups that are only appended extend the device's id maps in place
        // (k_batch_finish); anything else re-uploads them
        const bool maps_on_device =
            !maps_dirty && size == K0 && created > 0
            && (size_t)(size + created) <= maps_pcap
            && maps_pcap + tracker.g2p.size() + (size_t)created <= d_maps.cap;
        if (created) {
            py.counts.resize((size_t)(size + created), 0);
            py.reserve(K());
            for (auto & s : feats) {
                s->reserve(s->K + created);
                s->K += created;
            }
            for (int c = 0; c < created; ++c) tracker.add_group();
        }
        if (structural && !maps_on_device) maps_dirty = true;
        // appended groups zeroed, caches rebuilt, driver scores rebuilt: one
        // launch (k_batch_finish)
        py.n_empty = 0;
        py.sample_size = 0;
        for (int c : py.counts) {
            py.sample_size += c;
            py.n_empty += (c == 0);
        }
        FinishParams Q;
        memset(&Q, 0, sizeof(Q));
        Q.F = F();
        size_t cells = (size_t)K();
        const bool fresh = cells_fresh;
        for (int f = 0; f < F(); ++f) {
            Q.feat[f] = feats[f]->view();
            // with current cells only the appended groups need a full column
            const size_t groups = fresh ? (size_t)(K() - k_new) : (size_t)K();
            cells = std::max(cells, groups * std::max(1, feats[f]->dim()));
        }
        Q.counts = py.d_counts.p;
        Q.shifted = py.d_shifted.p;
        Q.K = K();
        Q.k_new = k_new;
        // (moved groups carry their cache entries with them)
        Q.cells_fresh = cells_fresh ? 1 : 0;
        cells_fresh = false;
        Q.alpha = alpha;
        Q.d = d;
        Q.nonempty = K() - py.n_empty;
        Q.empty = py.n_empty;
        base.reserve(grow_capacity((size_t)K()), 0);
        base_single.reserve(grow_capacity((size_t)K()), 0);
        Q.prep = DriverPrep{alpha, d, cluster, dataset_size, py.sample_size,
                            K(), py.n_empty, base.p, base_single.p, scalars.p};
        base_valid = true;
        if (maps_on_device) {
            Q.p2g = d_maps.p;
            Q.g2p = reinterpret_cast<int32_t *>(d_maps.p + maps_pcap);
            Q.first_new_global = first_new_global;
        }
        hipLaunchKernelGGL(k_batch_finish,
                           dim3((unsigned)((cells + kBlock - 1) / kBlock),
                                (unsigned)(F() + 1)),
                           dim3(kBlock), 0, stream(), Q);
        HIP_CHECK(hipGetLastError());
        // (reading the batch's kernel timer waits on nothing by now, and the
        // device already has its next launch)
        collect_timing();
    }

    // ---- sweeps with the group set normalised on the device ---------------
    // LDS of k_normalise: K + 2 ints and K / 2 + 1 slot pairs
    static size_t normalise_lds(int K) {
        return ((size_t)K + 2) * 4 + ((size_t)K / 2 + 1) * 8;
    }
    // Every batch of the sweep takes the value-sorted path, the statistics
    // are integers, and the bound on the group count fits the kernels' LDS.
    // (measured, C2: with the runs left open across sweeps -- settle() -- the
    // host's wait for each sub-sweep's group sizes costs 6 % at 10^6 rows per
    // sub-sweep and 18 % at 65 536; while every sweep pulled the state back it
    // only paid below some 500 000 rows)
    bool async_eligible(size_t r0, size_t r1, size_t batch) const {
        if (device_normalise_mode == 0 || cluster != 0 || F() != 1) return false;
        if (r1 <= r0 || any_float_stats() || py.n_empty < 1) return false;
        const size_t last = (r1 - r0) % batch;
        if (!use_value_sorted(std::min(batch, r1 - r0))) return false;
        if (last && !use_value_sorted(last)) return false;
        const size_t n_batches = (r1 - r0 + batch - 1) / batch;
        return async_bound_fits((size_t)K() + n_batches * (size_t)py.n_empty);
    }
    static bool async_bound_fits(size_t bound) {
        if (normalise_lds((int)bound) > 150 * 1024) return false;
        // k_vs_apply's plain form must fit (see apply_ints)
        return bound * 4 <= 144 * 1024;
    }
    void launch_normalise() {
        NormaliseParams N;
        memset(&N, 0, sizeof(N));
        N.F = F();
        for (int f = 0; f < F(); ++f) N.feat[f] = feats[f]->view();
        N.counts = py.d_counts.p;
        N.snap = snap_counts.p;
        N.p2g = d_maps.p;
        N.g2p = reinterpret_cast<int32_t *>(d_maps.p + maps_pcap);
        N.dev = dev_ptr();
        N.n_empty = py.n_empty;
        const size_t lds = normalise_lds(K());
        int device = 0;
        HIP_CHECK(hipGetDevice(&device));
        static std::atomic<size_t> opted_in[64];
        std::atomic<size_t> & have = opted_in[device & 63];
        if (lds > 64 * 1024 && lds > have.load(std::memory_order_relaxed)) {
            HIP_CHECK(hipFuncSetAttribute(
                reinterpret_cast<const void *>(&k_normalise),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            have.store(lds, std::memory_order_relaxed);
        }
        hipLaunchKernelGGL(k_normalise, dim3(1), dim3(kNormaliseBlock), lds,
                           stream(), N);
        HIP_CHECK(hipGetLastError());
    }
    // batch_finish with the group set normalised by the device
    void batch_finish_device() {
        DIST_REQUIRE(batch_open, "no open batch");
        batch_open = false;
        if (batch_fused) {
            // the next k_vs_tables normalises the group set and rebuilds the
            // caches on its way (or run_pending_finish, whoever comes first)
            finish_pending = true;
            return;
        }
        finish_kernels();
    }
    void run_pending_finish() {
        if (!finish_pending) return;
        finish_pending = false;
        finish_kernels();
    }
    // k_normalise + k_batch_finish on the state the last batch left
    void finish_kernels() {
        launch_normalise();
        FinishParams Q;
        memset(&Q, 0, sizeof(Q));
        Q.F = F();
        size_t cells = (size_t)K();
        for (int f = 0; f < F(); ++f) {
            Q.feat[f] = feats[f]->view();
            // (with current cells: the appended groups' columns only, at
            // most one per empty group)
            const size_t groups =
                cells_fresh ? (size_t)py.n_empty : (size_t)K();
            cells = std::max(cells, groups * std::max(1, feats[f]->dim()));
        }
        Q.counts = py.d_counts.p;
        Q.shifted = py.d_shifted.p;
        Q.cells_fresh = cells_fresh ? 1 : 0;
        cells_fresh = false;
        Q.alpha = alpha;
        Q.d = d;
        Q.empty = py.n_empty;
        Q.prep = DriverPrep{alpha, d, cluster, dataset_size, py.sample_size,
                            K(), py.n_empty, base.p, base_single.p, scalars.p};
        base_valid = true;
        Q.p2g = d_maps.p;
        Q.g2p = reinterpret_cast<int32_t *>(d_maps.p + maps_pcap);
        Q.dev = dev_ptr();
        Q.snap = snap_counts.p;
        hipLaunchKernelGGL(k_batch_finish,
                           dim3((unsigned)((cells + kBlock - 1) / kBlock),
                                (unsigned)(F() + 1)),
                           dim3(kBlock), 0, stream(), Q);
        HIP_CHECK(hipGetLastError());
    }
    // the host mirrors (group sizes, group count, id maps) from the device
    void pull_host_state() {
        // the run is queued on the stream of the thread that swept; another
        // thread closing it must drain THAT stream, not its own
        if (async_stream && async_stream != stream())
            HIP_CHECK(hipStreamSynchronize(async_stream));
        run_pending_finish();   // (a fused batch left its group set open)
        sync();
        DevState both[2];
        dev_state.download(both, 2);
        const DevState st = both[dev_cur];
        run_epoch = (uint32_t)st.pad;   // (where the next run goes on)
        const size_t Kn = (size_t)st.K;
        py.counts.resize(Kn);
        py.d_counts.download(py.counts.data(), Kn);
        py.n_empty = 0;
        py.sample_size = 0;
        for (int c : py.counts) {
            py.sample_size += c;
            py.n_empty += (c == 0);
        }
        for (auto & s : feats) s->K = (int)Kn;
        std::vector<uint32_t> maps(maps_pcap + st.global_size);
        d_maps.download(maps.data(), maps.size());
        tracker.p2g.assign(maps.begin(), maps.begin() + (long)Kn);
        tracker.g2p.resize(st.global_size);
        for (size_t i = 0; i < st.global_size; ++i)
            tracker.g2p[i] = (int32_t)maps[maps_pcap + i];
        maps_dirty = false;   // the device's copy IS the state
    }
    // Everything a device-normalised run of `n_batches` batches may grow
    // into is reserved up front (growing a buffer synchronises), the state
    // the kernels read is put on the device, and K() becomes the bound.
    hipEvent_t async_own0 = nullptr, async_own1 = nullptr;
    std::vector<size_t> async_rows;   // rows of each TIMED batch of the run
    // A run stays open when its sweep returns (the host's mirrors are pulled
    // by the next call that is not another such sweep: settle()), so that
    // consecutive sweeps pay for the hand-over of the state once: room for
    // kAsyncSweeps sweeps like the first is reserved where the kernels' LDS
    // allows, and async_left counts the batches still covered.
    static constexpr size_t kAsyncSweeps = 8;
    // ... and with the fused launch (whose work follows the per-launch bound
    // on the group count, k_limit(), not the run's) as many sweeps as its
    // LDS allows, up to this many batches
    static constexpr size_t kAsyncMaxBatches = 16384;
    // The group count is at most this when batch `index` of the open run is
    // sampled: every batch can fill each empty group once.  K_seen is the
    // device's own count, copied back without waiting for it (async_peek).
    size_t run_batches = 0;          // batches sampled in the open run
    int K_seen = 0;
    size_t K_seen_batch = 0;
    int batch_k_limit = 0;           // k_limit() when the open batch was sampled
    DevState * pinned_state = nullptr;
    hipEvent_t peek_event = nullptr;
    bool peek_pending = false;
    size_t peek_batch = 0;
    int k_limit() const {
        if (!async_active) return K();
        const size_t grown = (size_t)K_seen
            + (run_batches - K_seen_batch + 1) * (size_t)py.n_empty;
        return (int)std::min<size_t>((size_t)K(), grown);
    }
    // the device's state on its way to pinned memory (no wait); the next
    // sweep of the run picks it up if it has arrived
    void async_peek() {
        if (!pinned_state) {
            HIP_CHECK(hipHostMalloc((void **)&pinned_state, sizeof(DevState),
                                    hipHostMallocDefault));
            HIP_CHECK(hipEventCreateWithFlags(&peek_event,
                                              hipEventDisableTiming));
        }
        if (peek_pending) return;   // (one in flight)
        HIP_CHECK(hipMemcpyAsync(pinned_state, dev_ptr(), sizeof(DevState),
                                 hipMemcpyDeviceToHost, stream()));
        HIP_CHECK(hipEventRecord(peek_event, stream()));
        peek_pending = true;
        peek_batch = run_batches;
    }
    void async_peek_collect() {
        if (!peek_pending || hipEventQuery(peek_event) != hipSuccess) return;
        peek_pending = false;
        // (the state before the last batch's own moves were normalised)
        K_seen = pinned_state->K + py.n_empty;
        K_seen_batch = peek_batch;
    }
    size_t async_left = 0;
    int async_K0 = 0;                    // the group count the run began with
    hipStream_t async_stream = nullptr;  // the stream the run is queued on
    // a sharded run (dist_gibbs_sweep_sharded) goes on only with the tiling
    // all ranks agreed on when it was opened
    size_t sharded_batches = 0, sharded_batch_rows = 0;
    // A sharded run this rank closed ITSELF between two passes (any look at
    // the state does: settle) can be taken up again with the bound on the
    // group count and the batches it had left -- to its peers, who kept
    // theirs open, it then looks like a run that went on: the same collectives
    // of the same size (stat_words() follows the bound), and the same results
    // (closing a run never changes one).  So whether open runs go on needs
    // no word between the ranks, and no pass begins with a host round trip.
    size_t resume_bound = 0, resume_left = 0;   // 0: nothing to take up
    // What the ranks exchange per sub-sweep is sized by the LIVE part of the
    // group set, not by the run's bound (round 5 sent the bound's image: 8.4
    // MB where 1.1 MB were in use): a batch's deltas touch the groups that
    // existed when it was sampled, and every batch can fill each empty group
    // once, so batch j of a run that began with K0 groups touches slots below
    // K0 + j * n_empty -- a number every rank knows without asking (the
    // device's own count, K_seen, arrives asynchronously and is not).  A rank
    // that closed its run and takes it up again keeps the origin it shared
    // with its peers.
    size_t run_origin_K0 = 0, run_origin_done = 0;
    size_t resume_K0 = 0, resume_done = 0;
    uint64_t sharded_run_serial = 0;   // runs the ranks agreed to open
    size_t exchange_K() const {
        if (!async_active) return (size_t)K();
        return std::min<size_t>((size_t)K(), run_origin_K0
                                + run_origin_done * (size_t)py.n_empty);
    }
    void forget_resume() { resume_bound = resume_left = 0; }
    // the exchange's header did not add up (k_add_words, CommCheck): some
    // rank's run stood elsewhere, the sums applied since are not to be trusted
    void check_comm_fault() {
        if (!comm_fault || *comm_fault == 0) return;
        const unsigned at = *comm_fault & 0x7FFFFFFFu;
        *comm_fault = 0;
        poisoned = true;
        throw Error("ERROR ranks diverged: the header of exchange #"
                    + std::to_string(at) + " did not add up -- a rank changed "
                    "its engine between two passes of dist_gibbs_sweep_sharded "
                    "(or passes a different tiling); the statistics are lost");
    }
    uint64_t resumed_runs = 0;
    DeviceBuf<int> agree_flag;
    // Is [r0, r1) a range whose batches the fused launch takes, as far as
    // that is known before a run is open?  (ADVICE: a long run sizes every
    // OTHER path by its inflated bound on the group count -- k_vs_prepare's
    // tables, k_vs_apply's LDS, the staging matrix -- so it is only opened
    // where the first range has been seen to take the fused launch: a range
    // that was never sorted yet, i.e. a first sweep, gets a short run.)
    bool vs_expect_fused(size_t r0, size_t r1) const {
        if (!fused_tables_mode || sampling_mode != 0 || any_float_stats())
            return false;
        for (size_t i = 0; i < vs_ranges.size(); ++i) {
            if (vs_ranges[i].first != r0 || vs_ranges[i].second != r1) continue;
            const VsCache & c = *vs_cache[i];
            const int kind0 = feats[0]->sh.kind;
            if (c.mixed_chunks || use_stream(c)) return false;
            if ((kind0 == DIST_BNB || kind0 == DIST_GP) && c.n_other)
                return false;
            return (size_t)c.n_chunks * kTablesMaxK <= ((size_t)1 << 22);
        }
        return false;
    }
    int async_bound_K = 0;   // the open run's bound on the group count
    int run_batches_cap = 0; // debug.run_batches_cap (0: none)
    void async_begin(size_t n_first, bool expect_fused = true,
                     size_t forced_bound = 0, size_t forced_left = 0) {
        const int K0 = K();
        const size_t ne = (size_t)py.n_empty;
        size_t n_batches = kAsyncSweeps * n_first;
        // (only where the fused launch will run: every other path sizes its
        // work by the run's bound -- the scan mode's prefix tables, for one)
        if (expect_fused && fused_tables_mode && sampling_mode == 0
            && (size_t)K0 + n_first * ne + 64 <= kTablesMaxK) {
            const size_t room = ((size_t)kTablesMaxK - 64 - (size_t)K0) / ne;
            const size_t sweeps =
                std::min(room, kAsyncMaxBatches) / std::max<size_t>(n_first, 1);
            n_batches = std::max<size_t>(sweeps, 1) * n_first;
        }
        if (!async_bound_fits((size_t)K0 + n_batches * ne)
            || n_batches > kAsyncMaxBatches)
            n_batches = n_first;
        // (debug.run_batches_cap: short runs, so that a test sees one used up)
        if (run_batches_cap && n_first)
            n_batches = std::min(n_batches, std::max<size_t>(
                n_first, (size_t)run_batches_cap / n_first * n_first));
        if (forced_bound) n_batches = forced_left;   // (a run taken up again)
        async_left = n_batches;
        run_batches = 0;
        K_seen = K0;
        K_seen_batch = 0;
        if (peek_pending) {   // (a copy of another run's state: not wanted)
            (void)hipEventSynchronize(peek_event);
            peek_pending = false;
        }
        const int bound = forced_bound ? (int)forced_bound
                                       : K0 + (int)n_batches * py.n_empty;
        DIST_REQUIRE((size_t)K0 + n_batches * ne <= (size_t)bound,
                     "internal: a run's bound below what its batches can grow");
        async_bound_K = bound;
        if (forced_bound) {   // (taken up again: the peers' origin)
            run_origin_K0 = resume_K0;
            run_origin_done = resume_done;
        } else {
            run_origin_K0 = (size_t)K0;
            run_origin_done = 0;
        }
        resume_bound = resume_left = 0;
        py.reserve(bound);
        for (auto & s : feats) s->reserve(bound);
        if ((size_t)bound > base.cap || (size_t)bound > base_single.cap) {
            base.reserve(grow_capacity((size_t)bound), 0);
            base_single.reserve(grow_capacity((size_t)bound), 0);
            base_valid = false;   // (the new buffers are empty)
        }
        upload_maps((size_t)bound,
                    tracker.g2p.size() + n_batches * (size_t)py.n_empty);
        DevState st;
        memset(&st, 0, sizeof(st));
        st.K = K0;
        st.k_new = K0;
        st.global_size = (uint32_t)tracker.g2p.size();
        st.first_new_global = st.global_size;
        st.nonempty = K0 - py.n_empty;
        // (DevState::pad: the removal epoch the groups' recorded offsets are
        // stamped with, VsOffsets; it moves on once per batch at most)
        if (tracker.repacked != repacked_seen || run_epoch == 0u)
            run_epoch = (run_epoch + (1u << 20)) & ~((1u << 20) - 1u);
        repacked_seen = tracker.repacked;
        if (run_epoch == 0u) {   // wrapped: no old stamp may match again
            for (auto & c : vs_cache)
                if (c->off_epoch.p)
                    HIP_CHECK(hipMemsetAsync(c->off_epoch.p, 0,
                                             c->off_epoch.cap * 4, stream()));
            // ... nor an old entry of the swap-removal log answer to a new
            // epoch of the same number
            if (remap_log.p)
                HIP_CHECK(hipMemsetAsync(remap_log.p, 0, remap_log.cap * 4,
                                         stream()));
            run_epoch = 1u << 20;
        }
        st.pad = (int)run_epoch;
        const DevState both[2] = {st, st};
        dev_state.upload(both, 2);
        dev_cur = 0;
        finish_pending = false;
        snap_counts.reserve(grow_capacity((size_t)bound), 0);
        HIP_CHECK(hipMemcpyAsync(snap_counts.p, py.d_counts.p,
                                 (size_t)K0 * 4, hipMemcpyDeviceToDevice,
                                 stream()));
        // the other set of per-group buffers (k_vs_tables): as large as the
        // live ones, whichever way round they are swapped by now
        alt_counts.reserve(std::max(py.d_counts.cap, alt_counts.cap), 0);
        py.d_counts.reserve(alt_counts.cap, (size_t)K0);
        alt_snap.reserve(std::max(snap_counts.cap, alt_snap.cap), 0);
        snap_counts.reserve(alt_snap.cap, (size_t)K0);
        if (F() == 1) {
            Slave & f = *feats[0];
            const size_t cap = std::max<size_t>(
                {f.i0.cap, f.i1.cap, alt_i0.cap, alt_i1.cap, (size_t)f.cap});
            alt_i0.reserve(cap, 0);
            alt_i1.reserve(cap, 0);
            f.i0.reserve(cap, (size_t)K0);
            f.i1.reserve(cap, (size_t)K0);
        }
        if (!base_valid) {   // (with the true group count, before K() bounds)
            SweepParams P0 = params(0, 0, 0, 0);
            prepare(P0, false);
        }
        async_own0 = ev0;
        async_own1 = ev1;
        async_rows.clear();
        py.counts.resize((size_t)bound, 0);   // from here on K() is the bound
        async_K0 = K0;
        async_stream = stream();
        async_active = true;
        pairs_ticket = 0;
    }
    // sample one batch of a device-normalised run (its events from the pool)
    void async_sample(size_t b, size_t e, uint32_t seed, uint64_t draw_base) {
        // (events only for the batches kernel_timing picks: batch_sample's
        // own rule, looked at before it counts the batch)
        const bool will_time =
            e > b && kernel_timing > 0
            && timing_tick % (uint64_t)kernel_timing == 0;
        if (will_time) {
            const size_t i = async_rows.size();
            while (ev_pool.size() < 2 * (i + 1)) {
                hipEvent_t ev = nullptr;
                HIP_CHECK(hipEventCreate(&ev));
                ev_pool.push_back(ev);
            }
            ev0 = ev_pool[2 * i];
            ev1 = ev_pool[2 * i + 1];
        }
        batch_sample(b, e, seed, draw_base);
        DIST_REQUIRE(e == b || batch_value_sorted,
                     "internal: device-normalised run left its path");
        if (will_time) async_rows.push_back(e - b);
        run_batches += 1;
        async_batches += 1;
    }
    // back to host-driven operation; `failed`: on the way out of an error
    void async_end(bool failed) {
        ev0 = async_own0;
        ev1 = async_own1;
        timing_pending = false;
        if (failed) {
            batch_open = false;
            (void)hipStreamSynchronize(stream());
            async_active = false;
            tracker.repacked += 1;   // (recorded offsets: not to be trusted)
            try { pull_host_state(); } catch (...) {}
            return;
        }
        try {
            pull_host_state();   // (reads dev_state: K() is still the bound)
        } catch (...) {
            // the mirrors are unknown: every later call on this engine fails
            poisoned = true;
            async_active = false;
            throw;
        }
        // (a sharded run: remember what it had left, see resume_bound)
        resume_bound = sharded_batches ? (size_t)async_bound_K : 0;
        resume_left = sharded_batches ? async_left : 0;
        resume_K0 = run_origin_K0;
        resume_done = run_origin_done;
        async_active = false;
        collect_comm_timing();
        for (size_t i = 0; i < async_rows.size(); ++i) {
            if (!async_rows[i]) continue;
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, ev_pool[2 * i],
                                          ev_pool[2 * i + 1]));
            kernel_ms += ms;
            kernel_launches += 1;
            kernel_rows += async_rows[i];
        }
    }
    void sweep_async(size_t r0, size_t r1, size_t batch, uint32_t seed,
                     uint64_t draw_base) {
        const size_t n_batches = (r1 - r0 + batch - 1) / batch;
        if (!async_active) {   // else: it goes on
            // (the first range sorted now rather than by its first batch:
            // what it looks like decides how long a run is opened)
            const size_t e0 = std::min(r1, r0 + batch);
            drop_overlapping_caches(r0, e0, true);
            vs_get(r0, e0);
            async_begin(n_batches, vs_expect_fused(r0, e0));
            sharded_batches = sharded_batch_rows = 0;   // (not a sharded run)
        }
        async_left -= n_batches;
        async_peek_collect();
        try {
            for (size_t b = r0; b < r1; b += batch) {
                {
                    HOST_PROBE(0, "async_sample");
                    async_sample(b, std::min(r1, b + batch), seed, draw_base);
                }
                {
                    HOST_PROBE(1, "apply_ints");
                    apply_ints(live_image());
                }
                phase_mark(4);
                batch_finish_device();
                phase_mark(5);
            }
            {
                HOST_PROBE(2, "async_peek");
                async_peek();
            }
        } catch (...) {
            async_end(true);
            throw;
        }
        // (left open: settle())
    }
    // may this sweep go on with the open device-normalised run?
    bool async_continues(size_t r0, size_t r1, size_t batch) const {
        if (device_normalise_mode == 0 || r1 <= r0) return false;
        const size_t last = (r1 - r0) % batch;
        if (!use_value_sorted(std::min(batch, r1 - r0))) return false;
        if (last && !use_value_sorted(last)) return false;
        return (r1 - r0 + batch - 1) / batch <= async_left;
    }
    // the host's mirrors are current again (every entry point but sweep()
    // comes through here: dist_gibbs::impl)
    bool poisoned = false;   // a run could not be closed: state unknown
    void settle() {
        DIST_REQUIRE(!poisoned, "engine state lost: an open device-normalised "
                                "run failed to close, or the ranks diverged "
                                "(see the earlier error)");
        if (async_active && !batch_open) async_end(false);
        check_comm_fault();
    }
    // the sharded loop (dist_gibbs_sweep_sharded): rank-local conditions; the
    // ranks must agree before they rely on it (engine.ShardedGibbs)
    bool async_eligible_sharded(size_t n_batches, size_t batch) const {
        if (device_normalise_mode == 0 || cluster != 0 || F() != 1) return false;
        if (!n_batches || any_float_stats() || py.n_empty < 1) return false;
        for (size_t b = 0; b < n_batches; ++b) {
            const size_t r0 = std::min(n_rows, b * batch);
            const size_t r1 = std::min(n_rows, r0 + batch);
            if (r1 > r0 && !use_value_sorted(r1 - r0)) return false;
            if (r1 - r0 < batch) break;   // (the rest are empty)
        }
        const size_t bound = (size_t)K() + n_batches * (size_t)py.n_empty;
        if (normalise_lds((int)bound) > 150 * 1024) return false;
        return bound * 4 <= 144 * 1024;
    }

    void sweep(size_t r0, size_t r1, size_t batch, uint32_t seed,
               uint64_t draw_base) {
        DIST_REQUIRE(batch > 0, "batch_rows must be positive");
        DIST_REQUIRE(r0 <= r1 && r1 <= n_rows, "bad row range");
        // (a rank-local sweep is not a pass of the ranks' run: the run is
        // closed and not taken up again -- this rank's next sharded pass asks
        // its peers for a new one, and if they did not sweep likewise the
        // exchange's header tells, sharded_header)
        if (async_active && sharded_batches) settle();
        forget_resume();
        if (async_active) {   // an open run: go on with it, or close it
            if (async_continues(r0, r1, batch)) {
                sweep_async(r0, r1, batch, seed, draw_base);
                return;
            }
            settle();
        }
        if (async_eligible(r0, r1, batch)) {
            DIST_REQUIRE(!batch_open, "previous batch not finished");
            sweep_async(r0, r1, batch, seed, draw_base);
            return;
        }
        for (size_t b = r0; b < r1; b += batch) {
            const size_t e = std::min(r1, b + batch);
            {
                HOST_PROBE(14, "sweep: batch_sample");
                batch_sample(b, e, seed, draw_base);
            }
            {
                HOST_PROBE(15, "sweep: batch_apply_local");
                batch_apply_local();
            }
            {
                HOST_PROBE(16, "sweep: batch_finish");
                batch_finish();
            }
        }
        sync();
    }
    struct ChainLaunch {
        SweepParams * P;
        float * base;
        int32_t * counts;
        uint32_t * assign;
        const uint32_t * p2g;
        uint32_t rng;
        ChainResult * res;
        int K;
        int init = 0;   // 1 / 2: the initialisation loops (k_chain_rows)
        template <int A, int B, int NF>
        void run() {
            const size_t lds = (size_t)((K + 63) & ~63) * sizeof(float);
            if (init == 1)
                hipLaunchKernelGGL((k_chain_rows<A, B, NF, 1>), dim3(1),
                                   dim3(kBlock), lds, stream(), *P, base,
                                   counts, assign, p2g, rng, res);
            else if (init == 2)
                hipLaunchKernelGGL((k_chain_rows<A, B, NF, 2>), dim3(1),
                                   dim3(kBlock), lds, stream(), *P, base,
                                   counts, assign, p2g, rng, res);
            else
                hipLaunchKernelGGL((k_chain_rows<A, B, NF>), dim3(1),
                                   dim3(kBlock), lds, stream(), *P, base,
                                   counts, assign, p2g, rng, res);
            HIP_CHECK(hipGetLastError());
        }
    };
    // one row as a batch of one: the structural steps of the chain (a group
    // vanishing with its last member) take this path
    void sequential_row_as_batch(size_t r, uint32_t * rng_state) {
        // draw index (draw_base + row_offset + r) must be 0 for this row
        const uint64_t draw_base = (uint64_t)0 - (row_offset + r);
        batch_sample(r, r + 1, *rng_state, draw_base);
        batch_apply_local();
        batch_finish();
        *rng_state = lcg_mulmod(*rng_state, 16807u);
    }
    // ---- the exact chain, device-resident with its structural steps -----
    // (k_chains: one workgroup per chain; dist_gibbs_sweep_sequential_many
    // launches M engines' chains together)
    static constexpr int kChainRoom = 256;   // groups a launch may found
    static constexpr uint32_t kChainIds = 4096;   // ... and ids it may issue
    bool chain_fits() const {
        return sequential_mode == 2 && (size_t)K() + kChainRoom <= 8192;
    }
    size_t chain_lds_bytes() const {
        return (size_t)((K() + kChainRoom + 63) & ~63) * 8;
    }
    // everything the launch may grow into is reserved, the state the kernel
    // reads is put on the device
    ChainArgs chain_prepare(size_t r0, size_t r1, uint32_t rng) {
        DIST_REQUIRE(!batch_open, "previous batch not finished");
        DIST_REQUIRE(r0 <= r1 && r1 <= n_rows, "bad row range");
        DIST_REQUIRE(r1 <= assigned_rows || r0 == r1,
                     "rows without a group yet: init_sequential first");
        drop_overlapping_caches(r0, r1, false);
        flush_assign_pos();
        const int room = K() + kChainRoom;
        py.reserve(room);
        for (auto & s : feats) s->reserve(room);
        if ((size_t)room > base.cap || (size_t)room > base_single.cap) {
            base.reserve(grow_capacity((size_t)room), 0);
            base_single.reserve(grow_capacity((size_t)room), 0);
            base_valid = false;   // (the new buffers are empty)
        }
        upload_maps((size_t)room, tracker.g2p.size() + kChainIds);
        SweepParams P = params(r0, r1, 0, 0);
        prepare(P, false);
        DevState st;
        memset(&st, 0, sizeof(st));
        st.K = K();
        st.k_new = K();
        st.global_size = (uint32_t)tracker.g2p.size();
        st.first_new_global = st.global_size;
        st.nonempty = K() - py.n_empty;
        dev_state.upload(&st, 1);
        chain_result.reserve(1, 0);
        ChainArgs A;
        memset(&A, 0, sizeof(A));
        A.P = P;
        A.base = base.p;
        A.counts = py.d_counts.p;
        A.assign = assign;
        A.p2g = d_maps.p;
        A.g2p = reinterpret_cast<int32_t *>(d_maps.p + maps_pcap);
        A.dev = dev_state.p;
        A.result = chain_result.p;
        A.rng_state = rng;
        A.k_room = room;
        A.g_room = (uint32_t)std::min<size_t>(d_maps.cap - maps_pcap,
                                              0x7FFFFFFFu);
        static const bool stamps = getenv("DIST_CHAIN_STAMPS");
        if (stamps) {
            chain_stamps.reserve(8, 0);
            A.stamps = chain_stamps.p;
        }
        return A;
    }
    // the host's mirrors after a launch: group sizes, group count, id maps
    DeviceBuf<unsigned long long> chain_stamps;
    // (the three small downloads of an engine go to ONE pinned slot, queued
    // for all engines of a launch before the host waits once: a blocking copy
    // apiece cost a 1024-chain call 45 of its 95 ms)
    static constexpr size_t kChainSlotHead = 64;   // result + device state
    size_t chain_slot_bytes() const {
        return kChainSlotHead + (size_t)(K() + kChainRoom) * sizeof(int32_t);
    }
    void chain_collect_enqueue(char * slot, int room) {
        static_assert(sizeof(ChainResult) + sizeof(DevState) <= kChainSlotHead,
                      "slot head too small");
        HIP_CHECK(hipMemcpyAsync(slot, chain_result.p, sizeof(ChainResult),
                                 hipMemcpyDeviceToHost, stream()));
        HIP_CHECK(hipMemcpyAsync(slot + sizeof(ChainResult), dev_state.p,
                                 sizeof(DevState), hipMemcpyDeviceToHost,
                                 stream()));
        HIP_CHECK(hipMemcpyAsync(slot + kChainSlotHead, py.d_counts.p,
                                 (size_t)room * sizeof(int32_t),
                                 hipMemcpyDeviceToHost, stream()));
    }
    ChainResult chain_collect_finish(const char * slot) {
        ChainResult res;
        memcpy(&res, slot, sizeof(res));
        if (chain_stamps.p) {   // (diagnostic: DIST_CHAIN_STAMPS)
            unsigned long long t[5];
            chain_stamps.download(t, 5);
            const double n = res.rows_done ? (double)res.rows_done : 1.0;
            fprintf(stderr, "[dist] chain: %u rows; cycles per row: scores+max "
                    "%.0f, exp %.0f, recurrences %.0f, update %.0f; slow "
                    "path total %.0f\n", res.rows_done, t[0] / n, t[1] / n,
                    t[2] / n, t[3] / n, (double)t[4]);
        }
        DevState st;
        memcpy(&st, slot + sizeof(ChainResult), sizeof(st));
        const size_t Kn = (size_t)st.K;
        const bool structural = Kn != (size_t)K()
                                || st.global_size != tracker.g2p.size();
        py.counts.resize(Kn);
        memcpy(py.counts.data(), slot + kChainSlotHead, Kn * sizeof(int32_t));
        for (auto & s : feats) s->K = (int)Kn;
        if (structural) {
            std::vector<uint32_t> maps(maps_pcap + st.global_size);
            d_maps.download(maps.data(), maps.size());
            tracker.p2g.assign(maps.begin(), maps.begin() + (long)Kn);
            tracker.g2p.resize(st.global_size);
            for (size_t i = 0; i < st.global_size; ++i)
                tracker.g2p[i] = (int32_t)maps[maps_pcap + i];
            tracker.repacked += 1;   // (recorded offsets: not to be trusted)
            maps_dirty = false;      // the device's copy IS the state
        }
        py.rebuild(alpha, d);        // sample size, empty groups, shifted[]
        base_valid = false;
        cells_fresh = false;
        return res;
    }
    // pinned memory for the slots of a launch's engines (per thread, grown)
    static char * chain_pinned(size_t bytes) {
        static thread_local char * p = nullptr;
        static thread_local size_t cap = 0;
        if (bytes > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr;
            cap = 0;
            const size_t want = grow_capacity(bytes);
            HIP_CHECK(hipHostMalloc((void **)&p, want, hipHostMallocDefault));
            cap = want;
        }
        return p;
    }
    ChainResult chain_collect() {
        const int room = K() + kChainRoom;
        char * slot = chain_pinned(chain_slot_bytes());
        chain_collect_enqueue(slot, room);
        HIP_CHECK(hipStreamSynchronize(stream()));
        return chain_collect_finish(slot);
    }
    struct ChainsLaunch {
        const ChainArgs * args;
        unsigned chains;
        size_t lds;
        template <int A, int B, int NF, bool LOGL>
        void go() {
            (void)hipFuncSetAttribute(
                reinterpret_cast<const void *>(&k_chains<A, B, NF, LOGL>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipGetLastError();
            hipLaunchKernelGGL((k_chains<A, B, NF, LOGL>), dim3(chains),
                               dim3(kBlock), lds, stream(), args);
            HIP_CHECK(hipGetLastError());
        }
        template <int A, int B, int NF>
        void run() {
            // more than two chains per compute unit: the instance that
            // leaves FastLog's table where it lies (four fit a CU)
            int device = 0, cus = 256;
            if (hipGetDevice(&device) == hipSuccess)
                (void)hipDeviceGetAttribute(
                    &cus, hipDeviceAttributeMultiprocessorCount, device);
            static const char * force = getenv("DIST_CHAIN_LOG_TABLE");
            const bool in_lds = force ? atoi(force) != 0
                                      : chains <= 2u * (unsigned)cus;
            if (in_lds) go<A, B, NF, true>();
            else go<A, B, NF, false>();
        }
    };
    DeviceBuf<ChainArgs> chain_args;
    uint64_t chain_launches = 0;
    void sweep_sequential(size_t r0, size_t r1, uint32_t * rng_state) {
        DIST_REQUIRE(!batch_open, "previous batch not finished");
        resume_bound = resume_left = 0;
        DIST_REQUIRE(r0 <= r1 && r1 <= n_rows, "bad row range");
        DIST_REQUIRE(r1 <= assigned_rows || r0 == r1,
                     "rows without a group yet: init_sequential first");
        size_t r = r0;
        while (r < r1) {
            if (chain_fits()) {
                // the whole range on the device, structural steps included
                const ChainArgs A = chain_prepare(r, r1, *rng_state);
                chain_args.upload(&A, 1);
                ChainsLaunch L{chain_args.p, 1u, chain_lds_bytes()};
                dispatch(L);
                chain_launches += 1;
                const ChainResult res = chain_collect();
                DIST_REQUIRE(res.rows_done > 0 || res.event != 3,
                             "internal: the chain kernel found no room");
                *rng_state = res.rng_state;
                r += res.rows_done;
                continue;
            }
            if ((size_t)((K() + 63) & ~63) * sizeof(float) > 60 * 1024
                || sequential_mode == 0) {   // scores do not fit one LDS strip
                sequential_row_as_batch(r, rng_state);
                r += 1;
                continue;
            }
            // round 3's chain kernel: up to its first structural step
            drop_overlapping_caches(r, r1, false);
            flush_assign_pos();
            upload_maps();
            SweepParams P = params(r, r1, 0, 0);
            prepare(P, false);
            chain_result.reserve(1, 0);
            ChainLaunch L{&P, base.p, py.d_counts.p, assign, d_p2g_ptr,
                          *rng_state, chain_result.p, K()};
            dispatch(L);
            ChainResult res;
            chain_result.download(&res, 1);
            // bookkeeping as after a batch: host mirrors, appended groups,
            // caches rebuilt from the statistics
            batch_begin = batch_end = r;
            batch_value_sorted = false;
            batch_open = true;
            batch_finish();
            *rng_state = res.rng_state;
            r += res.rows_done;
            if (res.event == 1) {
                sequential_row_as_batch(r, rng_state);
                r += 1;
            }
        }
        sync();
    }

    // rows [assigned_rows, n_rows) have no group yet (load without
    // assignments); init_sequential gives them one, in order
    size_t assigned_rows = 0;
    void init_sequential(size_t r0, size_t r1, uint32_t * rng_state,
                         bool prior_only) {
        DIST_REQUIRE(!batch_open, "previous batch not finished");
        DIST_REQUIRE(r0 == assigned_rows && r0 <= r1 && r1 <= n_rows,
                     "init_sequential: rows are assigned in order, from the "
                     "first unassigned one");
        size_t r = r0;
        while (r < r1) {
            DIST_REQUIRE((size_t)((K() + 63) & ~63) * sizeof(float)
                             <= 60 * 1024,
                         "init_sequential: too many groups for the chain "
                         "kernel's strip of LDS");
            upload_maps();
            SweepParams P = params(r, r1, 0, 0);
            prepare(P, false);
            chain_result.reserve(1, 0);
            ChainLaunch L{&P, base.p, py.d_counts.p, assign, d_p2g_ptr,
                          *rng_state, chain_result.p, K(),
                          prior_only ? 2 : 1};
            dispatch(L);
            ChainResult res;
            chain_result.download(&res, 1);
            // host mirrors, the appended empty group, caches from the
            // statistics (the empty groups' prior follows the number of
            // non-empty ones, clustering.hpp:221-230)
            batch_begin = batch_end = r;
            batch_value_sorted = false;
            batch_open = true;
            batch_finish();
            *rng_state = res.rng_state;
            r += res.rows_done;
            assigned_rows = r;
        }
        sync();
    }

    void get_row_scores(size_t row, float * out, size_t * size_out) {
        DIST_REQUIRE(row < n_rows, "bad row");
        DIST_REQUIRE(row < assigned_rows,
                     "row without a group yet: init_sequential first");
        DIST_REQUIRE(!batch_open, "batch open");
        flush_assign_pos();
        upload_maps();
        row_scores.reserve(grow_capacity((size_t)K()), 0);
        row_size.reserve(1, 0);
        SweepParams P = params(row, row + 1, 0, 0);
        prepare(P);
        RowScoreLaunch L{this, &P, row};
        dispatch(L);
        int n = 0;
        row_size.download(&n, 1);
        row_scores.download(out, (size_t)n);
        *size_out = (size_t)n;
    }
    void score_rows(size_t r0, size_t r1, float * out_dev, size_t ld) {
        DIST_REQUIRE(r0 <= r1 && r1 <= n_rows && ld >= (size_t)K(),
                     "bad row range or leading dimension");
        DIST_REQUIRE(r1 <= assigned_rows || r0 == r1,
                     "rows without a group yet: init_sequential first");
        if (r0 == r1) return;
        flush_assign_pos();
        SweepParams P = params(r0, r1, 0, 0);
        prepare(P);
        LAUNCH(k_score_rows, (r1 - r0) * (size_t)K(), P, out_dev, ld);
    }
};

}  // namespace dist

// ===========================================================================
// C ABI

using namespace dist;

struct dist_py_mixture { PyDriver impl; };
struct dist_mixture { std::unique_ptr<Slave> impl; };
struct dist_id_tracker { Tracker impl; };
// Every entry point reaches the engine through `impl->`, which first closes a
// device-normalised run that a sweep left open (Gibbs::settle: the host's
// mirrors of the group set are pulled from the device); dist_gibbs_sweep alone
// takes `impl.open()` and may go on with it.
struct GibbsRef {
    std::unique_ptr<Gibbs> p;
    // (a failure while pulling is recorded like any entry point's, and the
    // call that follows meets the same broken device)
    // (every entry point dereferences inside its own guarded(): a failure
    // while closing an open run is that call's failure, not a later one's)
    // (... and, unless the entry point is known to leave the chain's state
    // alone -- read() --, forgets the sharded run this rank could have taken
    // up again: a rank that CHANGED something between two passes must not
    // pass for one that went on)
    Gibbs * operator->() const {
        p->settle();
        p->forget_resume();
        return p.get();
    }
    Gibbs & operator*() const {
        p->settle();
        p->forget_resume();
        return *p;
    }
    Gibbs * read() const {
        p->settle();
        return p.get();
    }
    Gibbs * open() const { return p.get(); }
    void reset(Gibbs * q) { p.reset(q); }
};
struct dist_gibbs { GibbsRef impl; };

extern "C" {

int dist_abi_version(void) { return DIST_ABI_VERSION; }
const char * dist_last_error(void) { return t_last_error.c_str(); }
int dist_device_count(int * count) {
    return guarded([&] { HIP_CHECK(hipGetDeviceCount(count)); });
}
int dist_set_device(int device) {
    return guarded([&] { HIP_CHECK(hipSetDevice(device)); });
}
int dist_synchronize(void) {
    return guarded([&] { HIP_CHECK(hipDeviceSynchronize()); });
}
int dist_set_stream(void * hip_stream) {
    return guarded([&] { t_stream = static_cast<hipStream_t>(hip_stream); });
}
size_t dist_group_words(const dist_shared_t * shared) {
    return group_words(*shared);
}

// ---- entropy ----------------------------------------------------------------
uint32_t dist_rng_seed(uint64_t seed) {
    const uint64_t s = seed % 2147483647ull;
    return (uint32_t)(s == 0 ? 1 : s);
}
uint32_t dist_rng_next(uint32_t * state) {
    *state = lcg_mulmod(*state, 16807u);
    return *state;
}
float dist_rng_unif01(uint32_t * state) {
    ensure_host_tables();
    return lcg_unif01(dist_rng_next(state));
}
uint32_t dist_rng_jump(uint32_t state, uint64_t steps) {
    return lcg_jump(state, steps);
}

// ---- vector_math ------------------------------------------------------------
static int vector_op(int op, size_t n, const void * in, float * out) {
    return guarded([&] {
        ensure_device_ready();
        if (!n) return;
        DeviceBuf<float> a, b;
        a.upload(static_cast<const float *>(in), n);
        b.reserve(n, 0);
        LAUNCH(k_vector_op, n, op, n, a.p, b.p);
        b.download(out, n);
    });
}
int dist_vector_log(size_t n, const float * in, float * out) {
    return vector_op(VEC_LOG, n, in, out);
}
int dist_vector_exp(size_t n, const float * in, float * out) {
    return vector_op(VEC_EXP, n, in, out);
}
int dist_vector_lgamma(size_t n, const float * in, float * out) {
    return vector_op(VEC_LGAMMA, n, in, out);
}
int dist_vector_lgamma_nu(size_t n, const float * in, float * out) {
    return vector_op(VEC_LGAMMA_NU, n, in, out);
}
int dist_vector_log_factorial(size_t n, const uint32_t * in, float * out) {
    return vector_op(VEC_LOG_FACTORIAL, n, in, out);
}

// ---- sampling ---------------------------------------------------------------
static void sample_kernel(int mode, size_t n, float * scores, bool copy_back,
                          float total, float u, SampleOut * out) {
    ensure_device_ready();
    DIST_REQUIRE(n > 0 || mode == 2, "expected 0 < size");
    Scratch & sc = scratch();
    sc.f.upload(scores, std::max<size_t>(n, 1));
    sc.s.reserve(1, 0);
    LAUNCH1(k_sample_scalar, mode, (int)n, sc.f.p, total, u, sc.s.p);
    sc.s.download(out, 1);
    if (copy_back) sc.f.download(scores, n);
}
int dist_sample_from_scores_overwrite(uint32_t * rng_state, size_t n,
                                      float * scores, size_t * sample_out) {
    return guarded([&] {
        ensure_host_tables();
        const float u = lcg_unif01(dist_rng_next(rng_state));
        SampleOut out;
        sample_kernel(1, n, scores, true, 0.f, u, &out);
        *sample_out = (size_t)out.sample;
    });
}
int dist_scores_to_likelihoods(size_t n, float * scores, float * total_out) {
    return guarded([&] {
        SampleOut out;
        sample_kernel(0, n, scores, true, 0.f, 0.f, &out);
        *total_out = out.total;
    });
}
int dist_sample_from_likelihoods(uint32_t * rng_state, size_t n,
                                 const float * likelihoods, float total,
                                 size_t * sample_out) {
    return guarded([&] {
        ensure_host_tables();
        const float u = lcg_unif01(dist_rng_next(rng_state));
        SampleOut out;
        sample_kernel(3, n, const_cast<float *>(likelihoods), false, total, u,
                      &out);
        *sample_out = (size_t)out.sample;
    });
}
int dist_log_sum_exp(size_t n, const float * scores, float * out_value) {
    return guarded([&] {
        if (n == 0) { *out_value = 0.f; return; }   // random.cc:80-82
        SampleOut out;
        sample_kernel(2, n, const_cast<float *>(scores), false, 0.f, 0.f, &out);
        *out_value = out.log_sum_exp;
    });
}

// PitmanYor::score_counts (src/clustering.cc:152-183) on device: the integer
// prefixes are formed on the host, the float terms on the GPU
static float py_score_counts(float alpha, float d, const int * counts,
                             size_t n) {
    ensure_device_ready();
    if (!n) return 0.f;
    std::vector<unsigned long long> before(2 * n);
    unsigned long long ne = 0, rows = 0;
    for (size_t k = 0; k < n; ++k) {
        DIST_REQUIRE(counts[k] >= 0, "negative group size");
        before[2 * k] = ne;
        before[2 * k + 1] = rows;
        if (counts[k]) { ne += 1; rows += (unsigned long long)counts[k]; }
    }
    DeviceBuf<int32_t> c;
    DeviceBuf<unsigned long long> b;
    DeviceBuf<double> out;
    c.upload(counts, n);
    b.upload(before.data(), 2 * n);
    out.reserve(1, 0);
    LAUNCH(k_py_score_counts, n, c.p, b.p, (int)n, alpha, d, out.p);
    double total = 0.0;
    out.download(&total, 1);
    return (float)total;
}

// ---- PitmanYor --------------------------------------------------------------
int dist_py_score_add_value(float alpha, float d, int group_size,
                            int nonempty, int sample_size, int empty,
                            float * out) {
    return guarded([&] {
        ensure_device_ready();
        Scratch & sc = scratch();
        sc.f.reserve(1, 0);
        LAUNCH1(k_py_score_add_value, alpha, d, group_size, nonempty,
                sample_size, empty, sc.f.p);
        sc.f.download(out, 1);
    });
}
int dist_py_score_remove_value(float alpha, float d, int group_size,
                               int nonempty, int sample_size, int empty,
                               float * out) {
    // clustering.hpp:106-123
    group_size -= 1;
    if (group_size == 0) nonempty -= 1;
    sample_size -= 1;
    int rc = dist_py_score_add_value(alpha, d, group_size, nonempty,
                                     sample_size, empty, out);
    if (rc == 0) *out = -*out;
    return rc;
}

// PitmanYor::sample_assignments (src/clustering.cc:67-142): an inherently
// sequential draw (each row sees the tables of the rows before it) made of
// float adds/compares and engine steps only -- host code, like the reference
int dist_py_sample_assignments(float alpha, float d, int size,
                               uint32_t * rng_state, int * assignments) {
    return guarded([&] {
        DIST_REQUIRE(size >= 0, "negative size");
        DIST_REQUIRE((float)size + 1.f > (float)size, "underflow expected");
        ensure_host_tables();
        std::vector<float> likelihoods;
        likelihoods.reserve(100);
        int table_count = 0;
        const float py_likelihood_new = 1 - d;
        likelihoods.push_back(alpha);
        if (size) {
            assignments[0] = 0;
            table_count = 1;
            likelihoods.push_back(alpha + d * table_count);
            likelihoods[0] = py_likelihood_new;
        }
        for (int i = 1; i < size; ++i) {
            const float total = i + alpha;
            // sample_from_likelihoods (random.hpp:316-333)
            float t = total * lcg_unif01(dist_rng_next(rng_state));
            int assign = (int)likelihoods.size() - 1;
            for (size_t k = 0; k < likelihoods.size(); ++k) {
                t -= likelihoods[k];
                if (t <= 0) { assign = (int)k; break; }
            }
            assignments[i] = assign;
            if (assign == table_count) {
                table_count += 1;
                likelihoods.push_back(alpha + d * table_count);
                likelihoods[assign] = py_likelihood_new;
            } else {
                likelihoods[assign] += 1.0f;
            }
        }
    });
}
int dist_py_score_counts(float alpha, float d, const int * counts, size_t n,
                         float * out) {
    return guarded([&] { *out = py_score_counts(alpha, d, counts, n); });
}
int dist_py_mixture_score_data(const dist_py_mixture_t * m, float alpha,
                               float d, float * out) {
    return guarded([&] {
        *out = py_score_counts(alpha, d, m->impl.counts.data(),
                               m->impl.counts.size());
    });
}

// ---- Clustering<int>::LowEntropy ------------------------------------------
static float le_log_partition_function(int n) {   // clustering.cc:204-215
    ensure_host_tables();
    DIST_REQUIRE(n >= 0, "negative sample size");
    if (n < 48) return u2f(DIST_LE_LOG_PARTITION[n]);
    const float coeff = 0.28269584f;
    const float log_z_max = (float)n * fast_log((float)n);
    return log_z_max * (1.f + coeff * powf((float)n, -0.75f));
}
int dist_le_score_add_value(int dataset_size, int group_size,
                            int nonempty_group_count, int sample_size,
                            int empty_group_count, float * out) {
    (void)nonempty_group_count;
    return guarded([&] {
        ensure_device_ready();
        Scratch & sc = scratch();
        sc.f.reserve(1, 0);
        LAUNCH1(k_le_score_add_value, dataset_size, group_size, sample_size,
                empty_group_count, sc.f.p);
        sc.f.download(out, 1);
    });
}
int dist_le_score_remove_value(int dataset_size, int group_size,
                               int nonempty_group_count, int sample_size,
                               int empty_group_count, float * out) {
    // clustering.hpp:294-309
    const int rc = dist_le_score_add_value(dataset_size, group_size - 1,
                                           nonempty_group_count, sample_size,
                                           empty_group_count, out);
    if (rc == 0) *out = -*out;
    return rc;
}
int dist_le_log_partition_function(int sample_size, float * out) {
    return guarded([&] { *out = le_log_partition_function(sample_size); });
}
// clustering.cc:229-248; the sum of n log n terms in binary64 on the device,
// the closing scalar arithmetic in float like the reference
static float le_score_counts(int dataset_size, const int * counts, size_t n) {
    ensure_device_ready();
    long long sample_size = 0;
    for (size_t i = 0; i < n; ++i) {
        DIST_REQUIRE(counts[i] >= 0, "negative group size");
        sample_size += counts[i];
    }
    DIST_REQUIRE(sample_size <= dataset_size, "sample_size > dataset_size");
    double terms = 0.0;
    if (n) {
        DeviceBuf<int32_t> dc;
        DeviceBuf<double> out;
        dc.upload(counts, n);
        out.reserve(1, 0);
        LAUNCH(k_le_count_terms, n, dc.p, (int)n, out.p);
        out.download(&terms, 1);
    }
    float score = (float)terms;
    if (sample_size != dataset_size) {
        const float log_factor =
            le_postpred_correction((float)sample_size, dataset_size);
        score += log_factor * (float)(n - 1);
        const float ln = fast_log((float)sample_size);
        const float lN = fast_log((float)dataset_size);
        score += 0.061f * ln * (ln - lN) * powf(ln + lN, 0.75f);
    }
    score -= le_log_partition_function((int)sample_size);
    return score;
}
// LowEntropy::sample_assignments (clustering.cc:250-283): a sequential draw
// over a growing likelihood vector -- host code, like the reference's and
// like dist_py_sample_assignments
int dist_le_sample_assignments(int dataset_size, int sample_size,
                               uint32_t * rng_state, int * assignments) {
    return guarded([&] {
        DIST_REQUIRE(sample_size >= 0 && sample_size <= dataset_size,
                     "expected 0 <= sample_size <= dataset_size");
        ensure_host_tables();
        std::vector<int> counts;
        std::vector<float> likelihoods;
        counts.reserve(100);
        likelihoods.reserve(100);
        int size = 0;
        for (int i = 0; i < sample_size; ++i) {
            const float likelihood_empty =
                fast_exp(le_score_add_value(dataset_size, 0, size, 1));
            if (counts.empty() || counts.back()) {
                counts.push_back(0);
                likelihoods.push_back(likelihood_empty);
            } else {
                likelihoods.back() = likelihood_empty;
            }
            // sample_from_likelihoods(rng, likelihoods), random.hpp:316-341
            const float total =
                vector_sum_as_built(likelihoods.size(), likelihoods.data());
            float t = total * lcg_unif01(dist_rng_next(rng_state));
            int assign = (int)likelihoods.size() - 1;
            for (size_t k = 0; k < likelihoods.size(); ++k) {
                t -= likelihoods[k];
                if (t <= 0) { assign = (int)k; break; }
            }
            assignments[i] = assign;
            counts[assign] += 1;
            size += 1;
            likelihoods[assign] =
                fast_exp(le_score_add_value(dataset_size, counts[assign], 0, 1));
        }
    });
}
int dist_le_score_counts(int dataset_size, const int * counts,
                         size_t group_count, float * out) {
    return guarded(
        [&] { *out = le_score_counts(dataset_size, counts, group_count); });
}

// LowEntropy::Mixture = MixtureDriver<LowEntropy, int> (mixture.hpp:48-163):
// the counts bookkeeping is the cached driver's, the scores are computed
// afresh per call as the generic driver does
struct dist_le_mixture {
    PyDriver impl;
};
dist_le_mixture_t * dist_le_mixture_create(void) {
    dist_le_mixture_t * m = nullptr;
    guarded([&] { m = new dist_le_mixture(); });
    return m;
}
void dist_le_mixture_destroy(dist_le_mixture_t * m) { delete m; }
int dist_le_mixture_init(dist_le_mixture_t * m, const int * counts, size_t n) {
    return guarded([&] { m->impl.init(0.f, 0.f, counts, n); dist::sync(); });
}
int dist_le_mixture_add_value(dist_le_mixture_t * m, size_t groupid,
                              int * added_out) {
    return guarded([&] { *added_out = m->impl.add_value(0.f, 0.f, groupid); });
}
int dist_le_mixture_remove_value(dist_le_mixture_t * m, size_t groupid,
                                 int * removed_out) {
    return guarded(
        [&] { *removed_out = m->impl.remove_value(0.f, 0.f, groupid); });
}
int dist_le_mixture_score_value(const dist_le_mixture_t * m, int dataset_size,
                                float * scores, size_t size) {
    return guarded([&] {
        const PyDriver & d = m->impl;
        DIST_REQUIRE(size == d.counts.size(),
                     "scores.size() != counts().size()");
        if (!size) return;
        Scratch & sc = scratch();
        sc.f.reserve(size, 0);
        LAUNCH(k_le_score, size, d.d_counts.p, sc.f.p, d.K(), dataset_size,
               (int)d.sample_size, d.n_empty);
        sc.f.download(scores, size);
    });
}
int dist_le_mixture_score_data(const dist_le_mixture_t * m, int dataset_size,
                               float * out) {
    return guarded([&] {
        *out = le_score_counts(dataset_size, m->impl.counts.data(),
                               m->impl.counts.size());
    });
}
size_t dist_le_mixture_size(const dist_le_mixture_t * m) {
    return m->impl.counts.size();
}
size_t dist_le_mixture_sample_size(const dist_le_mixture_t * m) {
    return (size_t)m->impl.sample_size;
}
int dist_le_mixture_counts(const dist_le_mixture_t * m, int * out) {
    return guarded([&] {
        std::copy(m->impl.counts.begin(), m->impl.counts.end(), out);
    });
}

dist_py_mixture_t * dist_py_mixture_create(void) {
    dist_py_mixture_t * m = nullptr;
    guarded([&] { m = new dist_py_mixture(); });
    return m;
}
void dist_py_mixture_destroy(dist_py_mixture_t * m) { delete m; }
int dist_py_mixture_init(dist_py_mixture_t * m, float alpha, float d,
                         const int * counts, size_t n) {
    return guarded([&] { m->impl.init(alpha, d, counts, n); dist::sync(); });
}
int dist_py_mixture_add_value(dist_py_mixture_t * m, float alpha, float d,
                              size_t groupid, int * added_out) {
    return guarded([&] { *added_out = m->impl.add_value(alpha, d, groupid); });
}
int dist_py_mixture_remove_value(dist_py_mixture_t * m, float alpha, float d,
                                 size_t groupid, int * removed_out) {
    return guarded(
        [&] { *removed_out = m->impl.remove_value(alpha, d, groupid); });
}
int dist_py_mixture_score_value(const dist_py_mixture_t * m, float alpha,
                                float d, float * scores, size_t size) {
    (void)d;
    return guarded([&] { m->impl.score_value(alpha, scores, size); });
}
size_t dist_py_mixture_size(const dist_py_mixture_t * m) {
    return m->impl.counts.size();
}
size_t dist_py_mixture_sample_size(const dist_py_mixture_t * m) {
    return (size_t)m->impl.sample_size;
}
int dist_py_mixture_counts(const dist_py_mixture_t * m, int * out) {
    return guarded([&] {
        // the device copy is the authority; the mirror must agree
        std::vector<int> dev(m->impl.counts.size());
        m->impl.d_counts.download(dev.data(), dev.size());
        DIST_REQUIRE(dev == m->impl.counts, "count mirror out of sync");
        std::copy(dev.begin(), dev.end(), out);
    });
}
size_t dist_py_mixture_empty_groupids(const dist_py_mixture_t * m,
                                      size_t * out, size_t cap) {
    size_t n = 0;
    for (size_t i = 0; i < m->impl.counts.size(); ++i) {
        if (m->impl.counts[i] == 0) {
            if (n < cap) out[n] = i;
            n += 1;
        }
    }
    return n;
}

// ---- Model::Mixture ---------------------------------------------------------
dist_mixture_t * dist_mixture_create(const dist_shared_t * shared) {
    dist_mixture_t * m = nullptr;
    guarded([&] {
        std::unique_ptr<dist_mixture> p(new dist_mixture());
        p->impl.reset(new Slave(*shared));
        m = p.release();
    });
    return m;
}
void dist_mixture_destroy(dist_mixture_t * m) { delete m; }
int dist_mixture_clear(dist_mixture_t * m) {
    return guarded([&] { m->impl->clear(); });
}
int dist_mixture_append(dist_mixture_t * m, const uint32_t * group) {
    return guarded([&] { m->impl->append(group); });
}
int dist_mixture_get_group(const dist_mixture_t * m, size_t groupid,
                           uint32_t * out) {
    return guarded([&] { dist::sync(); m->impl->get_group(groupid, out); });
}
size_t dist_mixture_size(const dist_mixture_t * m) { return (size_t)m->impl->K; }
int dist_mixture_init(dist_mixture_t * m) {
    return guarded([&] { m->impl->init(); dist::sync(); });
}
int dist_mixture_add_group(dist_mixture_t * m) {
    return guarded([&] { m->impl->add_group(); });
}
int dist_mixture_remove_group(dist_mixture_t * m, size_t groupid) {
    return guarded([&] { m->impl->remove_group(groupid); });
}
int dist_mixture_add_value(dist_mixture_t * m, size_t groupid, uint32_t value) {
    return guarded([&] { m->impl->value_op(groupid, value, 1); });
}
int dist_mixture_remove_value(dist_mixture_t * m, size_t groupid,
                              uint32_t value) {
    return guarded([&] { m->impl->value_op(groupid, value, 0); });
}
int dist_mixture_score_value_group(const dist_mixture_t * m, size_t groupid,
                                   uint32_t value, float * out) {
    return guarded([&] { *out = m->impl->score_value_group(groupid, value); });
}
int dist_mixture_score_value(const dist_mixture_t * m, uint32_t value,
                             float * scores_accum, size_t size) {
    return guarded([&] { m->impl->score_value(value, scores_accum, size); });
}

int dist_mixture_score_values(const dist_mixture_t * m,
                              const uint32_t * values, size_t n,
                              float * scores_accum, size_t ld) {
    return guarded([&] { m->impl->score_values(values, n, scores_accum, ld); });
}
int dist_mixture_score_data(const dist_mixture_t * m, float * out) {
    return guarded([&] { *out = m->impl->score_data(); });
}
int dist_mixture_score_data_grid(const dist_mixture_t * m,
                                 const dist_shared_t * shareds, size_t n,
                                 float * scores_out) {
    return guarded([&] { m->impl->score_data_grid(shareds, n, scores_out); });
}

// ---- Model::Group (host scalars over the same inline model code) -----------
int dist_group_init(const dist_shared_t * shared, uint32_t * group) {
    return guarded([&] {
        check_shared(*shared);
        memset(group, 0, 4 * group_words(*shared));
    });
}
static Stats group_to_stats(const dist_shared_t & sh, const uint32_t * g) {
    Stats st = {0, 0, 0.f, 0.f};
    st.i0 = (int32_t)g[0];
    if (sh.kind == DIST_BB || sh.kind == DIST_BNB) st.i1 = (int32_t)g[1];
    if (sh.kind == DIST_GP) { st.i1 = (int32_t)g[1]; st.f0 = u2f(g[2]); }
    if (sh.kind == DIST_NICH) { st.f0 = u2f(g[1]); st.f1 = u2f(g[2]); }
    return st;
}
static void stats_to_group(const dist_shared_t & sh, const Stats & st,
                           uint32_t * g) {
    g[0] = (uint32_t)st.i0;
    if (sh.kind == DIST_BB || sh.kind == DIST_BNB) g[1] = (uint32_t)st.i1;
    if (sh.kind == DIST_GP) { g[1] = (uint32_t)st.i1; g[2] = f2u(st.f0); }
    if (sh.kind == DIST_NICH) { g[1] = f2u(st.f0); g[2] = f2u(st.f1); }
}
static void group_value_op(const dist_shared_t * shared, uint32_t * group,
                           uint32_t value, bool add) {
    ensure_host_tables();
    check_shared(*shared);
    if (is_cat(shared->kind))
        DIST_REQUIRE(value < (uint32_t)shared->dim, "value out of bounds");
    Stats st = group_to_stats(*shared, group);
    if (add) stats_add(shared->kind, st, value);
    else stats_remove(shared->kind, st, value);
    stats_to_group(*shared, st, group);
    if (is_cat(shared->kind)) group[1 + value] += add ? 1u : 0xFFFFFFFFu;
}
int dist_group_add_value(const dist_shared_t * shared, uint32_t * group,
                         uint32_t value) {
    return guarded([&] { group_value_op(shared, group, value, true); });
}
int dist_group_remove_value(const dist_shared_t * shared, uint32_t * group,
                            uint32_t value) {
    return guarded([&] { group_value_op(shared, group, value, false); });
}
int dist_group_score_value(const dist_shared_t * sh, const uint32_t * group,
                           uint32_t value, float * out) {
    return guarded([&] {
        ensure_host_tables();
        check_shared(*sh);
        if (sh->kind == DIST_DD) {   // dd.hpp:222-245
            DIST_REQUIRE(value < (uint32_t)sh->dim, "value out of bounds");
            float alpha_sum = 0.f, mine = 0.f;
            for (int v = 0; v < sh->dim; ++v) {
                const float a = sh->alphas[v] + (float)(int32_t)group[1 + v];
                if ((uint32_t)v == value) mine = a;
                alpha_sum += a;
            }
            *out = fast_log(mine / alpha_sum);
            return;
        }
        if (sh->kind == DIST_DPD) {  // dpd.hpp:223-232
            const float alpha = sh->p[0];
            const float numer = value == DIST_DPD_OTHER
                ? alpha * sh->p[1]
                : alpha * sh->betas[value] + (float)(int32_t)group[1 + value];
            const float denom = alpha + (float)(int32_t)group[0];
            *out = fast_log(numer / denom);
            return;
        }
        const Stats st = group_to_stats(*sh, group);
        const Entry e = scorer_init(sh->kind, sh->p, st);
        const float lf = sh->kind == DIST_GP ? fast_log_factorial(value) : 0.f;
        *out = score_group(sh->kind, e, value, lf, sh->p);
    });
}

size_t dist_scorer_words(const dist_shared_t * sh) {
    return is_cat(sh->kind) ? (size_t)sh->dim + 1 : 4;
}
int dist_scorer_init(const dist_shared_t * sh, const uint32_t * group,
                     float * state) {
    return guarded([&] {
        ensure_host_tables();
        check_shared(*sh);
        if (sh->kind == DIST_DD) {   // dd.hpp:226-236
            float alpha_sum = 0.f;
            for (int v = 0; v < sh->dim; ++v) {
                const float alpha = sh->alphas[v] + (float)(int32_t)group[1 + v];
                state[1 + v] = alpha;
                alpha_sum += alpha;
            }
            state[0] = alpha_sum;
            return;
        }
        if (sh->kind == DIST_DPD) {  // dpd.hpp:312-333
            const float alpha = sh->p[0];
            const float total = (float)(size_t)(int32_t)group[0];
            const float beta_scale = alpha / (alpha + total);
            state[0] = fast_log(beta_scale * sh->p[1]);
            const float counts_scale = 1.0f / (alpha + total);
            for (int v = 0; v < sh->dim; ++v) {
                float score = sh->betas[v] * beta_scale;
                const int32_t count = (int32_t)group[1 + v];
                // (the reference adds the counts it HAS: absent ones add nothing)
                if (count) score += counts_scale * (float)count;
                state[1 + v] = fast_log(score);
            }
            return;
        }
        const Entry e = scorer_init(sh->kind, sh->p, group_to_stats(*sh, group));
        state[0] = e.c0; state[1] = e.c1; state[2] = e.c2; state[3] = e.c3;
    });
}
int dist_scorer_eval(const dist_shared_t * sh, const float * state,
                     uint32_t value, float * out) {
    return guarded([&] {
        ensure_host_tables();
        if (sh->kind == DIST_DD) {   // dd.hpp:238-244
            DIST_REQUIRE(value < (uint32_t)sh->dim, "value out of bounds");
            *out = fast_log(state[1 + value] / state[0]);
            return;
        }
        if (sh->kind == DIST_DPD) {  // dpd.hpp:335-340
            DIST_REQUIRE(value == DIST_DPD_OTHER || value < (uint32_t)sh->dim,
                         "value out of bounds");
            *out = value == DIST_DPD_OTHER ? state[0] : state[1 + value];
            return;
        }
        const Entry e = {state[0], state[1], state[2], state[3]};
        const float lf = sh->kind == DIST_GP ? fast_log_factorial(value) : 0.f;
        *out = score_group(sh->kind, e, value, lf, sh->p);
    });
}

int dist_group_score_data(const dist_shared_t * sh, const uint32_t * group,
                          float * out) {
    return guarded([&] {
        ensure_host_tables();
        check_shared(*sh);
        if (is_cat(sh->kind)) {   // dd.hpp:160-177, dpd.hpp:234-250
            const float alpha_sum = [&] {
                if (sh->kind == DIST_DPD) return sh->p[0];
                float a = 0.f;
                for (int v = 0; v < sh->dim; ++v) a += sh->alphas[v];
                return a;
            }();
            float score = 0.f;
            for (int v = 0; v < sh->dim; ++v) {
                const float prior = sh->kind == DIST_DD
                    ? sh->alphas[v] : sh->p[0] * sh->betas[v];
                const int32_t c = (int32_t)group[1 + v];
                if (sh->kind == DIST_DPD && c == 0) continue;   // sparse counts
                score += fast_lgamma(prior + (float)c) - fast_lgamma(prior);
            }
            score += fast_lgamma(alpha_sum)
                   - fast_lgamma(alpha_sum + (float)(int32_t)group[0]);
            *out = score;
            return;
        }
        *out = scalar_group_score_data(sh->kind, sh->p,
                                       group_to_stats(*sh, group));
    });
}

// ---- MixtureIdTracker -------------------------------------------------------
dist_id_tracker_t * dist_id_tracker_create(void) { return new dist_id_tracker(); }
void dist_id_tracker_destroy(dist_id_tracker_t * t) { delete t; }
int dist_id_tracker_init(dist_id_tracker_t * t, size_t n) {
    return guarded([&] { t->impl.init(n); });
}
int dist_id_tracker_add_group(dist_id_tracker_t * t) {
    return guarded([&] { t->impl.add_group(); });
}
int dist_id_tracker_remove_group(dist_id_tracker_t * t, uint32_t packed) {
    return guarded([&] { t->impl.remove_group(packed); });
}
int dist_id_tracker_packed_to_global(const dist_id_tracker_t * t,
                                     uint32_t packed, uint32_t * out) {
    return guarded([&] { *out = t->impl.packed_to_global(packed); });
}
int dist_id_tracker_global_to_packed(const dist_id_tracker_t * t,
                                     uint32_t global, uint32_t * out) {
    return guarded([&] { *out = t->impl.global_to_packed(global); });
}
size_t dist_id_tracker_packed_size(const dist_id_tracker_t * t) {
    return t->impl.p2g.size();
}
size_t dist_id_tracker_global_size(const dist_id_tracker_t * t) {
    return t->impl.g2p.size();
}

// ---- batched engine ---------------------------------------------------------
dist_gibbs_t * dist_gibbs_create(float alpha, float d, int n_features,
                                 const dist_shared_t * shareds) {
    dist_gibbs_t * g = nullptr;
    guarded([&] {
        std::unique_ptr<dist_gibbs> p(new dist_gibbs());
        p->impl.reset(new Gibbs(alpha, d, n_features, shareds));
        g = p.release();
    });
    return g;
}
dist_gibbs_t * dist_gibbs_create_low_entropy(int dataset_size, int n_features,
                                             const dist_shared_t * shareds) {
    dist_gibbs_t * g = nullptr;
    guarded([&] {
        DIST_REQUIRE(dataset_size > 0, "LowEntropy: dataset_size must be > 0");
        std::unique_ptr<dist_gibbs> p(new dist_gibbs());
        // (alpha, d) only feed the cached PitmanYor scores, which the
        // LowEntropy kernels never read
        p->impl.reset(new Gibbs(1.f, 0.f, n_features, shareds));
        p->impl->cluster = 1;
        p->impl->dataset_size = dataset_size;
        g = p.release();
    });
    return g;
}
void dist_gibbs_destroy(dist_gibbs_t * g) { delete g; }
int dist_gibbs_load_rows(dist_gibbs_t * g, size_t n_rows,
                         const uint32_t * const * values,
                         const uint32_t * assign_packed, int nonempty,
                         int empty, uint64_t row_offset) {
    return guarded([&] {
        g->impl->load(n_rows, values, false, assign_packed, nullptr, nonempty,
                      empty, row_offset);
    });
}
int dist_gibbs_load_rows_unassigned(dist_gibbs_t * g, size_t n_rows,
                                    const uint32_t * const * values,
                                    int empty, uint64_t row_offset) {
    return guarded([&] {
        g->impl->load(n_rows, values, false, nullptr, nullptr, 0, empty,
                      row_offset);
    });
}
int dist_gibbs_init_sequential(dist_gibbs_t * g, size_t row_begin,
                               size_t row_end, uint32_t * rng_state,
                               int prior_only) {
    return guarded([&] {
        g->impl->init_sequential(row_begin, row_end, rng_state,
                                 prior_only != 0);
    });
}
int dist_gibbs_load_rows_dev(dist_gibbs_t * g, size_t n_rows,
                             const uint32_t * const * values_dev,
                             uint32_t * assign_packed_dev, int nonempty,
                             int empty, uint64_t row_offset) {
    return guarded([&] {
        g->impl->load(n_rows, values_dev, true, nullptr, assign_packed_dev,
                      nonempty, empty, row_offset);
    });
}
// (the size_t getters have no status channel: (size_t)-1 reports a failure,
// with its message in dist_last_error)
size_t dist_gibbs_stat_words(const dist_gibbs_t * g) {
    size_t n = (size_t)-1;
    (void)guarded([&] { n = g->impl.read()->stat_words(); });
    return n;
}
int dist_gibbs_export_stats_dev(const dist_gibbs_t * g, int32_t * stats_dev) {
    return guarded([&] {
        g->impl.read()->require_whole("export_stats");
        g->impl.read()->copy_stats(stats_dev, true);
        dist::sync();
    });
}
int dist_gibbs_import_stats_dev(dist_gibbs_t * g, const int32_t * stats_dev) {
    return guarded([&] {
        g->impl->copy_stats(const_cast<int32_t *>(stats_dev), false);
        g->impl->resume_bound = g->impl->resume_left = 0;
        g->impl->pairs_ticket = 0;   // whatever a batch published is stale now
        g->impl->early_ticket = 0;
        g->impl->cells_partial = false;   // (whole again: the caller's image)
        g->impl->refresh_host_counts();
        g->impl->rebuild_caches();
        dist::sync();
    });
}
int dist_gibbs_sweep(dist_gibbs_t * g, size_t row_begin, size_t row_end,
                     size_t batch_rows, uint32_t seed_state,
                     uint64_t draw_base) {
    return guarded([&] {
        g->impl.open()->sweep(row_begin, row_end, batch_rows, seed_state,
                              draw_base);
    });
}
int dist_comm_available(void) { return rccl().ok ? 1 : 0; }
int dist_comm_unique_id(uint8_t id_out[128]) {
    return guarded([&] {
        DIST_REQUIRE(rccl().ok, "RCCL could not be bound");
        static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
        ncclUniqueId id;
        RCCL_CHECK(rccl().get_unique_id(&id));
        memcpy(id_out, &id, 128);
    });
}
int dist_comm_unique_id_host(uint8_t id_out[128]) {
    return guarded([&] {
        memset(id_out, 0, 128);
        memcpy(id_out, kHostIdMagic, 8);
        // the segment's name: 8 random bytes + the pid, as hex
        unsigned char rnd[8] = {0};
        FILE * f = fopen("/dev/urandom", "rb");
        const size_t got = f ? fread(rnd, 1, sizeof(rnd), f) : 0;
        if (f) fclose(f);
        if (got != sizeof(rnd)) {
            const uint64_t t = (uint64_t)std::chrono::steady_clock::now()
                                   .time_since_epoch().count();
            memcpy(rnd, &t, sizeof(rnd));
        }
        char hex[25];
        snprintf(hex, sizeof(hex), "%02x%02x%02x%02x%02x%02x%02x%02x%08x",
                 rnd[0], rnd[1], rnd[2], rnd[3], rnd[4], rnd[5], rnd[6],
                 rnd[7], (unsigned)getpid());
        memcpy(id_out + 8, hex, 24);
    });
}
dist_comm_t * dist_comm_create(const uint8_t id[128], int rank, int world) {
    dist_comm_t * c = nullptr;
    guarded([&] {
        DIST_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank/world");
        ensure_device_ready();
        std::unique_ptr<dist_comm> p(new dist_comm());
        p->rank = rank;
        p->world = world;
        if (memcmp(id, kHostIdMagic, 8) == 0) {   // dist_comm_unique_id_host
            p->host.reset(new HostComm(id, rank, world));
        } else {
            DIST_REQUIRE(rccl().ok, "RCCL could not be bound");
            ncclUniqueId uid;
            memcpy(&uid, id, 128);
            RCCL_CHECK(rccl().comm_init_rank(&p->comm, world, uid, rank));
        }
        c = p.release();
    });
    return c;
}
void dist_comm_destroy(dist_comm_t * c) {
    if (!c) return;
    if (c->comm && rccl().ok) (void)rccl().comm_destroy(c->comm);
    delete c;
}
int dist_comm_size(const dist_comm_t * c, int * rank_out, int * world_out) {
    return guarded([&] {
        DIST_REQUIRE(c && c->valid(), "no communicator");
        if (rank_out) *rank_out = c->rank;
        if (world_out) *world_out = c->world;
    });
}
int dist_comm_all_reduce_dev(dist_comm_t * c, void * data_dev, size_t count,
                             int type, int op) {
    return guarded([&] {
        DIST_REQUIRE(c && c->valid(), "no communicator");
        DIST_REQUIRE((type == COMM_I32 || type == COMM_F64)
                         && (op == COMM_SUM || op == COMM_MIN),
                     "unknown element type or operation");
        c->all_reduce(data_dev, count, (CommType)type, (CommOp)op, stream());
        HIP_CHECK(hipStreamSynchronize(stream()));
    });
}
// (splitmix64's finaliser: the header's signatures)
static uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
int dist_gibbs_sweep_sharded(dist_gibbs_t * g, dist_comm_t * c,
                             size_t n_batches, size_t batch_rows,
                             uint32_t seed_state, uint64_t draw_base) {
    return guarded([&] {
        DIST_REQUIRE(c && c->valid(), "no communicator");
        DIST_REQUIRE(batch_rows > 0, "batch_rows must be positive");
        DIST_REQUIRE(!g->impl.open()->any_float_stats()
                         || g->impl.open()->merged_floats(),
                     "order-dependent statistics are exchanged as rows "
                     "(dist_gibbs_batch_moves_dev / replay_ordered_dev), or "
                     "as sums with float_stats = 1");
        // The group set is normalised on the device -- no host round trip per
        // sub-sweep on any rank -- when EVERY rank can (one all-reduce of a
        // flag when a run is opened: the layout of the delta image depends
        // on it).  Like the single-engine sweep, the run then stays open
        // across passes of the same tiling; any other call settles it.
        Gibbs * open = g->impl.open();
        open->check_comm_fault();
        auto agree = [&](Gibbs & e, int mine) {
            e.agree_flag.reserve(1, 0);
            HIP_CHECK(hipMemcpyAsync(e.agree_flag.p, &mine, sizeof(int),
                                     hipMemcpyHostToDevice, stream()));
            c->all_reduce(e.agree_flag.p, 1, COMM_I32, COMM_MIN, stream());
            e.agree_flag.download(&mine, 1);
            return mine != 0;
        };
        // Does the ranks' run go on?  Decided by every rank for itself, from
        // what is the same on all of them: the tiling of this call and of the
        // run, and the batches the run has left.  A rank that kept its run
        // open goes on with it; a rank that closed it between the passes by
        // LOOKING at its state (the entry points that read: GibbsRef::read)
        // takes it up again with the same bound, the same batches left and
        // the same origin (resume_bound): its collectives are the ones its
        // peers issue.  No word between the ranks, no host round trip at the
        // start of a pass (round 4 agreed on every call: 50 us per pass,
        // profiles/r5_collective_pass.txt).  Only when the run is used up, on
        // every rank at the same call, do they agree on a new one.  A rank
        // that CHANGED something in between (any other entry point: options,
        // rows, statistics, a sweep of its own) has forgotten the run and asks
        // for a new one here while its peers go on: a caller's error, told by
        // the host transport at once (collectives of different sizes) and,
        // where the sizes happen to agree, by the exchange's header.
        const bool tiling_same = open->sharded_batches == n_batches
                                 && open->sharded_batch_rows == batch_rows;
        bool on_device = false;
        if (open->async_active && !open->batch_open && tiling_same
            && n_batches <= open->async_left) {
            on_device = true;
        } else if (!open->async_active && !open->batch_open && tiling_same
                   && open->resume_bound && n_batches <= open->resume_left
                   && (size_t)open->K() + open->resume_left
                              * (size_t)open->py.n_empty <= open->resume_bound
                   && open->sharded_device_normalise
                   && open->async_eligible_sharded(n_batches, batch_rows)) {
            const size_t bound = open->resume_bound, left = open->resume_left;
            open->async_begin(n_batches, true, bound, left);
            open->resumed_runs += 1;
            on_device = true;
        }
        if (!on_device) {
            Gibbs & s = *g->impl;   // (settles an open run, forgets it)
            on_device = agree(
                s, s.sharded_device_normalise
                       && s.async_eligible_sharded(n_batches, batch_rows)
                   ? 1 : 0);
            s.sharded_run_serial += 1;
            if (on_device) {
                // (the run's bound on the group count sizes the buffers: the
                // same on every rank, so not a function of what this rank's
                // ranges look like)
                s.async_begin(n_batches);
                s.sharded_batches = n_batches;
                s.sharded_batch_rows = batch_rows;
            } else {
                s.sharded_batches = s.sharded_batch_rows = 0;
            }
        }
        Gibbs & e = *open;
        if (on_device) {
            e.async_left -= std::min(e.async_left, n_batches);
            e.async_peek_collect();
        }
        if (!e.comm_fault) {
            HIP_CHECK(hipHostMalloc((void **)&e.comm_fault, sizeof(unsigned),
                                    hipHostMallocDefault));
            *e.comm_fault = 0;
        }
        const int ne = e.py.n_empty;
        // where this rank believes the ranks' run stands at batch `done` of
        // it, exchanging `kx` groups: two 12-bit signatures
        auto signature = [&](size_t done, size_t kx, uint64_t & tag) {
            uint64_t h = mix64(e.sharded_run_serial);
            h = mix64(h ^ (uint64_t)done);
            h = mix64(h ^ (uint64_t)kx);
            h = mix64(h ^ (uint64_t)n_batches);
            h = mix64(h ^ (uint64_t)batch_rows);
            h = mix64(h ^ (uint64_t)(e.value_partitioned ? 1 : 0));
            h = mix64(h ^ (uint64_t)(on_device ? 1 : 0));
            tag = h | 1ull;
            return std::make_pair((int32_t)(h & 0xFFF),
                                  (int32_t)((h >> 12) & 0xFFF));
        };
        try {
            for (size_t b = 0; b < n_batches; ++b) {
                const size_t r0 = std::min(e.n_rows, b * batch_rows);
                const size_t r1 = std::min(e.n_rows, r0 + batch_rows);
                // the groups this batch's deltas can touch: rank-independent
                // (the run's bound only sizes buffers, see exchange_K)
                const size_t kx = e.exchange_K();
                const size_t done = on_device ? e.run_origin_done : b;
                if (on_device) e.async_sample(r0, r1, seed_state, draw_base);
                else e.batch_sample(r0, r1, seed_state, draw_base);
                const size_t words = e.exchange_words(kx);
                const size_t total = kCommHeaderWords + words;
                // the exchange buffer is zeroed once; k_add_words clears what
                // it consumes, so it is all zero again before every batch
                if (total > e.delta_image.cap || !e.delta_image.p) {
                    e.delta_image.reserve(grow_capacity(total), 0);
                    HIP_CHECK(hipMemsetAsync(e.delta_image.p, 0,
                                             e.delta_image.cap * 4, stream()));
                    e.delta_header_tag = 0;
                }
                int32_t * image = e.delta_image.p + kCommHeaderWords;
                uint64_t tag = 0;
                const auto sig = signature(done, kx, tag);
                if (e.delta_header_tag != tag)   // (not left by the last batch)
                    hipLaunchKernelGGL(k_comm_header, dim3(1), dim3(1), 0,
                                       stream(), e.delta_image.p, sig.first,
                                       sig.second);
                e.batch_delta_exchange(image, kx);
                // in place, on the engine's stream: no hop to another stream
                const bool timed = e.kernel_timing > 0
                                   && e.comm_tick++ % (uint64_t)e.kernel_timing
                                          == 0;
                hipEvent_t t0 = nullptr, t1 = nullptr;
                if (timed) {
                    t0 = e.comm_event();
                    t1 = e.comm_event();
                    HIP_CHECK(hipEventRecord(t0, stream()));
                }
                c->all_reduce(e.delta_image.p, total, COMM_I32, COMM_SUM,
                              stream());
                e.comm_collectives += 1;
                e.comm_words_total += total;
                e.comm_words_last = total;
                e.comm_words_max = std::max<uint64_t>(e.comm_words_max, total);
                if (timed) {
                    HIP_CHECK(hipEventRecord(t1, stream()));
                    e.comm_ev_pending.emplace_back(t0, t1);
                }
                // the kernel that consumes the sum checks the header and
                // leaves the NEXT batch's behind (no launch of its own then)
                CommCheck chk;
                memset(&chk, 0, sizeof(chk));
                chk.header = e.delta_image.p;
                chk.world = c->world;
                chk.fault = e.comm_fault;
                chk.tag = (unsigned)(e.comm_collectives & 0x7FFFFFFFu)
                          | 0x80000000u;
                e.delta_header_tag = 0;
                if (b + 1 < n_batches) {
                    const size_t kx_next =
                        on_device ? std::min<size_t>(
                                        (size_t)e.K(),
                                        e.run_origin_K0
                                            + (e.run_origin_done + 1) * (size_t)ne)
                                  : 0;
                    if (on_device) {
                        uint64_t tag_next = 0;
                        const auto nx = signature(done + 1, kx_next, tag_next);
                        chk.next[0] = nx.first;
                        chk.next[1] = nx.first * nx.first;
                        chk.next[2] = nx.second;
                        chk.next[3] = nx.second * nx.second;
                        e.delta_header_tag = tag_next;
                    }
                }
                e.batch_apply_words(image, true, kx, !e.value_partitioned, chk);
                if (on_device) e.run_origin_done += 1;
                if (e.merged_floats()) {   // the float statistics as sums
                    e.merge_float_delta();
                    c->all_reduce(e.merge_image.p,
                                  (size_t)e.merge_layout().words, COMM_F64,
                                  COMM_SUM, stream());
                    e.merge_float_apply(e.merge_image.p);
                }
                if (on_device) e.batch_finish_device();
                else e.batch_finish();
            }
            if (on_device) e.async_peek();
        } catch (...) {
            if (on_device) e.async_end(true);
            throw;
        }
        if (!on_device) {
            dist::sync();
            e.collect_comm_timing();
            e.check_comm_fault();
        }
        // (an on-device run stays open: Gibbs::settle)
    });
}

// ---- value-partitioned ranks ------------------------------------------------
int dist_gibbs_partition_by_value(dist_gibbs_t * g, dist_comm_t * c) {
    return guarded([&] {
        DIST_REQUIRE(c && c->valid(), "no communicator");
        Gibbs & e = *g->impl;
        DIST_REQUIRE(e.F() == 1 && is_cat(e.feats[0]->sh.kind)
                         && e.feats[0]->dim() > 0,
                     "value partitioning takes engines with one categorical "
                     "feature (DirichletDiscrete, DirichletProcessDiscrete)");
        DIST_REQUIRE(!e.batch_open, "a batch is open");
        const int dim = e.feats[0]->dim();
        DeviceBuf<int32_t> has;
        has.reserve((size_t)dim * 2, 0);   // [mine | everybody's]
        if (e.n_rows)
            LAUNCH(k_value_presence,
                   std::min<size_t>(e.n_rows, (size_t)2048 * kBlock),
                   e.values[0], e.n_rows, dim, has.p);
        HIP_CHECK(hipMemcpyAsync(has.p + dim, has.p, (size_t)dim * 4,
                                 hipMemcpyDeviceToDevice, stream()));
        c->all_reduce(has.p + dim, (size_t)dim, COMM_I32, COMM_SUM, stream());
        std::vector<int32_t> h((size_t)dim * 2);
        has.download(h.data(), h.size());
        std::vector<int32_t> owned((size_t)dim);
        e.present_values = 0;
        for (int x = 0; x < dim; ++x) {
            e.present_values += h[x] ? 1 : 0;
            DIST_REQUIRE(h[(size_t)dim + x] <= 1,
                         "value " + std::to_string(x) + " has rows on "
                         + std::to_string(h[(size_t)dim + x])
                         + " ranks: not a partition by value");
            // (a value nobody has rows of: its cells never change; rank 0's)
            owned[x] = h[x] || (h[(size_t)dim + x] == 0 && c->rank == 0);
        }
        e.owned_values.upload(owned.data(), owned.size());
        e.value_partitioned = true;
        dist::sync();
    });
}
int dist_gibbs_gather_cells(dist_gibbs_t * g, dist_comm_t * c) {
    return guarded([&] {
        DIST_REQUIRE(c && c->valid(), "no communicator");
        Gibbs & e = *g->impl.read();
        DIST_REQUIRE(e.value_partitioned, "not a value-partitioned engine");
        DIST_REQUIRE(!e.batch_open, "a batch is open");
        const int dim = e.feats[0]->dim();
        const size_t cells = (size_t)e.K() * dim;
        DeviceBuf<int32_t> image;
        image.reserve(std::max<size_t>(cells, 1), 0);
        if (cells)
            LAUNCH(k_owned_cells, cells, e.feats[0]->cnt.p, e.owned_values.p,
                   cells, dim, image.p);
        c->all_reduce(image.p, cells, COMM_I32, COMM_SUM, stream());
        if (cells)
            HIP_CHECK(hipMemcpyAsync(e.feats[0]->cnt.p, image.p, cells * 4,
                                     hipMemcpyDeviceToDevice, stream()));
        e.cells_partial = false;
        e.rebuild_caches();
        dist::sync();
    });
}
int dist_gibbs_comm_volume(dist_gibbs_t * g, uint64_t out[4], int reset) {
    return guarded([&] {
        Gibbs * e = g->impl.open();   // (counters only: the run stays open)
        out[0] = e->comm_collectives;
        out[1] = e->comm_words_total;
        out[2] = e->comm_words_max;
        out[3] = e->comm_words_last;
        if (reset)
            e->comm_collectives = e->comm_words_total = e->comm_words_max =
                e->comm_words_last = 0;
    });
}

int dist_gibbs_sweep_sequential(dist_gibbs_t * g, size_t row_begin,
                                size_t row_end, uint32_t * rng_state) {
    return guarded(
        [&] { g->impl->sweep_sequential(row_begin, row_end, rng_state); });
}
int dist_gibbs_sweep_sequential_many(dist_gibbs_t * const * engines, size_t m,
                                     size_t row_begin, size_t row_end,
                                     uint32_t * rng_states) {
    return guarded([&] {
        DIST_REQUIRE(m == 0 || (engines && rng_states), "null argument");
        if (!m) return;
        std::vector<Gibbs *> e(m);
        for (size_t i = 0; i < m; ++i) {
            DIST_REQUIRE(engines[i], "null engine");
            e[i] = &*engines[i]->impl;   // (settles, forgets a sharded run)
            for (size_t j = 0; j < i; ++j)
                DIST_REQUIRE(e[j] != e[i], "the same engine twice");
            DIST_REQUIRE(e[i]->F() == e[0]->F(), "engines of one feature list");
            for (int f = 0; f < e[0]->F(); ++f)
                DIST_REQUIRE(e[i]->feats[f]->sh.kind == e[0]->feats[f]->sh.kind,
                             "engines of one feature list");
        }
        // chains whose group count is beyond the kernel's strips (or with the
        // kernel switched off) take their own path, one after the other
        std::vector<size_t> at(m, row_begin);
        std::vector<size_t> todo;
        for (size_t i = 0; i < m; ++i) {
            if (e[i]->chain_fits()) {
                todo.push_back(i);
            } else {
                e[i]->sweep_sequential(row_begin, row_end, &rng_states[i]);
                at[i] = row_end;
            }
        }
        std::vector<ChainArgs> args;
        while (!todo.empty()) {
            args.clear();
            size_t lds = 0;
            for (size_t i : todo) {
                args.push_back(e[i]->chain_prepare(at[i], row_end,
                                                   rng_states[i]));
                lds = std::max(lds, e[i]->chain_lds_bytes());
            }
            Gibbs & first = *e[todo[0]];
            first.chain_args.upload(args.data(), args.size());
            Gibbs::ChainsLaunch L{first.chain_args.p, (unsigned)todo.size(),
                                  lds};
            first.dispatch(L);
            first.chain_launches += 1;
            // every engine's small downloads queued, then ONE wait
            std::vector<size_t> slot_at(todo.size());
            size_t slots = 0;
            for (size_t j = 0; j < todo.size(); ++j) {
                slot_at[j] = slots;
                slots += (e[todo[j]]->chain_slot_bytes() + 63) & ~(size_t)63;
            }
            char * pinned = Gibbs::chain_pinned(std::max<size_t>(slots, 64));
            for (size_t j = 0; j < todo.size(); ++j)
                e[todo[j]]->chain_collect_enqueue(
                    pinned + slot_at[j],
                    e[todo[j]]->K() + Gibbs::kChainRoom);
            HIP_CHECK(hipStreamSynchronize(stream()));
            std::vector<size_t> again;
            for (size_t j = 0; j < todo.size(); ++j) {
                const size_t i = todo[j];
                const ChainResult res =
                    e[i]->chain_collect_finish(pinned + slot_at[j]);
                DIST_REQUIRE(res.rows_done > 0 || res.event != 3
                                 || at[i] >= row_end,
                             "internal: the chain kernel found no room");
                rng_states[i] = res.rng_state;
                at[i] += res.rows_done;
                // (out of room, or a group count beyond the strips now)
                if (at[i] < row_end) {
                    if (e[i]->chain_fits()) {
                        again.push_back(i);
                    } else {
                        e[i]->sweep_sequential(at[i], row_end, &rng_states[i]);
                        at[i] = row_end;
                    }
                }
            }
            todo.swap(again);
        }
        dist::sync();
    });
}
int dist_gibbs_batch_sample(dist_gibbs_t * g, size_t row_begin, size_t row_end,
                            uint32_t seed_state, uint64_t draw_base) {
    return guarded([&] {
        g->impl->batch_sample(row_begin, row_end, seed_state, draw_base);
    });
}
int dist_gibbs_batch_delta_dev(dist_gibbs_t * g, int32_t * delta_dev) {
    // no host sync: the collective is ordered after these kernels on the
    // same (default) stream
    return guarded([&] { g->impl->batch_delta(delta_dev); });
}
int dist_gibbs_batch_apply_delta_dev(dist_gibbs_t * g,
                                     const int32_t * delta_dev) {
    // (clear = false: the caller's image is only read)
    return guarded([&] {
        g->impl->batch_apply_delta(const_cast<int32_t *>(delta_dev), false);
    });
}
int dist_gibbs_batch_apply_local(dist_gibbs_t * g) {
    return guarded([&] { g->impl->batch_apply_local(); });
}
int dist_gibbs_ordered_features(const dist_gibbs_t * g, int * count_out) {
    return guarded([&] {
        int n = 0;
        for (auto & f : g->impl.read()->feats)
            if (has_float_stats(f->sh.kind)) n += 1;
        *count_out = n;
    });
}
int dist_gibbs_batch_moves_dev(dist_gibbs_t * g, uint32_t * old_slot_dev,
                               uint32_t * new_slot_dev) {
    return guarded([&] { g->impl->batch_moves(old_slot_dev, new_slot_dev); });
}
int dist_gibbs_replay_ordered_dev(dist_gibbs_t * g,
                                  const uint32_t * old_slot_dev,
                                  const uint32_t * new_slot_dev,
                                  const uint32_t * const * values_dev,
                                  size_t n_rows, int reset) {
    return guarded([&] {
        g->impl->replay_ordered(old_slot_dev, new_slot_dev, values_dev,
                                n_rows, reset != 0);
    });
}
size_t dist_gibbs_float_delta_words(const dist_gibbs_t * g) {
    size_t n = (size_t)-1;
    (void)guarded([&] {
        Gibbs * e = g->impl.open();
        n = e->merged_floats() ? (size_t)e->merge_layout().words : 0;
    });
    return n;
}
int dist_gibbs_export_float_moments_dev(dist_gibbs_t * g, double * out_dev) {
    return guarded([&] {
        Gibbs & e = *g->impl.read();
        DIST_REQUIRE(!e.batch_open, "a batch is open");
        DIST_REQUIRE(e.merged_floats(), "float_stats is not 1 (merged)");
        const MergeLayout L = e.merge_layout();
        SweepParams P = e.params(0, 0, 0, 0);
        LAUNCH(k_merge_float_export, (size_t)e.K(), P, L, out_dev);
    });
}
int dist_gibbs_import_float_moments_dev(dist_gibbs_t * g,
                                        const double * image_dev) {
    return guarded([&] {
        Gibbs & e = *g->impl;
        DIST_REQUIRE(!e.batch_open, "a batch is open");
        DIST_REQUIRE(e.merged_floats(), "float_stats is not 1 (merged)");
        e.merge_float_apply(image_dev, true);
        e.rebuild_caches();
        dist::sync();
    });
}
int dist_gibbs_batch_float_delta_dev(dist_gibbs_t * g, double * delta_dev) {
    return guarded([&] {
        Gibbs & e = *g->impl;
        DIST_REQUIRE(e.batch_open, "no open batch");
        DIST_REQUIRE(e.merged_floats(), "float_stats is not 1 (merged)");
        e.merge_float_delta();
        HIP_CHECK(hipMemcpyAsync(delta_dev, e.merge_image.p,
                                 (size_t)e.merge_layout().words * 8,
                                 hipMemcpyDeviceToDevice, stream()));
    });
}
int dist_gibbs_batch_apply_float_delta_dev(dist_gibbs_t * g,
                                           const double * delta_dev) {
    return guarded([&] {
        Gibbs & e = *g->impl;
        DIST_REQUIRE(e.batch_open, "no open batch");
        DIST_REQUIRE(e.merged_floats(), "float_stats is not 1 (merged)");
        e.merge_float_apply(delta_dev);
    });
}
int dist_gibbs_batch_finish(dist_gibbs_t * g) {
    return guarded([&] { g->impl->batch_finish(); });
}
int dist_gibbs_row_scores(dist_gibbs_t * g, size_t row, float * scores_out,
                          size_t * size_out) {
    return guarded([&] { g->impl.read()->get_row_scores(row, scores_out, size_out); });
}
int dist_gibbs_score_rows_dev(dist_gibbs_t * g, size_t row_begin,
                              size_t row_end, float * scores_dev, size_t ld) {
    return guarded([&] {
        g->impl.read()->score_rows(row_begin, row_end, scores_dev, ld);
        dist::sync();
    });
}
size_t dist_gibbs_group_count(const dist_gibbs_t * g) {
    size_t n = (size_t)-1;
    (void)guarded([&] { n = (size_t)g->impl.read()->K(); });
    return n;
}
size_t dist_gibbs_row_count(const dist_gibbs_t * g) {
    return g->impl.open()->n_rows;
}
int dist_gibbs_counts(const dist_gibbs_t * g, int * out) {
    return guarded([&] {
        std::vector<int> dev((size_t)g->impl.read()->K());
        g->impl.read()->py.d_counts.download(dev.data(), dev.size());
        std::copy(dev.begin(), dev.end(), out);
    });
}
int dist_gibbs_assignments(const dist_gibbs_t * g, uint32_t * global_out) {
    return guarded([&] {
        g->impl.read()->flush_assign_pos();
        dist::sync();
        if (g->impl.read()->n_rows) {
            HIP_CHECK(hipMemcpyAsync(global_out, g->impl.read()->assign,
                                     g->impl.read()->n_rows * 4,
                                     hipMemcpyDeviceToHost, stream()));
            dist::sync();
        }
    });
}
int dist_gibbs_get_group(const dist_gibbs_t * g, int feature, size_t groupid,
                         uint32_t * group_out) {
    return guarded([&] {
        DIST_REQUIRE(feature >= 0 && feature < g->impl.read()->F(), "bad feature");
        g->impl.read()->require_whole("get_group");
        dist::sync();
        g->impl.read()->feats[feature]->get_group(groupid, group_out);
    });
}
int dist_gibbs_packed_to_global(const dist_gibbs_t * g, uint32_t packed,
                                uint32_t * out) {
    return guarded([&] { *out = g->impl.read()->tracker.packed_to_global(packed); });
}
int dist_gibbs_global_to_packed(const dist_gibbs_t * g, uint32_t global,
                                uint32_t * out) {
    return guarded([&] { *out = g->impl.read()->tracker.global_to_packed(global); });
}
size_t dist_gibbs_global_size(const dist_gibbs_t * g) {
    size_t n = (size_t)-1;
    (void)guarded([&] { n = g->impl.read()->tracker.g2p.size(); });
    return n;
}
int dist_gibbs_validate(dist_gibbs_t * g, dist_validate_report_t * report) {
    dist_validate_report_t local;
    dist_validate_report_t * rep = report ? report : &local;
    const int rc = guarded([&] { g->impl.read()->validate(rep); });
    if (rc) return rc;
    if (rep->code == 0) return 0;
    char msg[256];
    snprintf(msg, sizeof(msg),
             "validate: %s (code %d, feature %d, group/row %lld, detail %lld, "
             "expected %lld, found %lld)", rep->what, rep->code, rep->feature,
             rep->group, rep->detail, rep->expected, rep->found);
    set_last_error(msg);
    return 2;
}
int dist_mixture_validate(const dist_mixture_t * m) {
    return guarded([&] { m->impl->validate(); });
}
int dist_gibbs_sharded_device_normalise_ok(const dist_gibbs_t * g,
                                           size_t n_batches,
                                           size_t batch_rows, int * ok_out) {
    return guarded([&] {
        *ok_out = g->impl.read()->async_eligible_sharded(n_batches, batch_rows) ? 1 : 0;
    });
}
int dist_gibbs_set_option(dist_gibbs_t * g, const char * name, int value) {
    return guarded([&] {
        // The public options first (include/distributions_hip.h lists them);
        // "debug.<name>" are the tests' hooks: each forces a kernel variant
        // the library otherwise picks by itself, and none changes a result.
        std::string key(name);
        const bool hook = key.compare(0, 6, "debug.") == 0;
        if (hook) key = key.substr(6);
        static const char * const hooks[] = {
            "sequential_chain", "running_sums_min_tiles", "narrow_read_ahead",
            "stream_scratch", "rows_scratch", "rows_scratch_lds_log",
            "rows_scratch_block", "rows_fold", "apply_stage", "program_all",
            "sample_prio", "rows_prio", "apply_overlap", "run_batches_cap"};
        bool is_hook = false;
        for (const char * h : hooks) is_hook = is_hook || key == h;
        DIST_REQUIRE(hook == is_hook,
                     is_hook ? "a test hook: spell it debug." + key
                             : "unknown option debug." + key);
        if (key == "sample_prio" || key == "rows_prio") {
            // wave priorities by phase: 0 none, else 0x10000 | four levels
            DIST_REQUIRE(value == 0 || (value >> 16) == 1,
                         "sample_prio / rows_prio: 0 or 0x1abcd");
            (key == "sample_prio" ? g->impl->sample_prio_mode
                                  : g->impl->rows_prio_mode) = value;
        } else if (key == "run_batches_cap") {
            // a device-normalised run covers at most this many batches (a
            // whole number of passes, at least one); 0: as many as fit
            DIST_REQUIRE(value >= 0, "run_batches_cap: >= 0");
            g->impl->run_batches_cap = value;
        } else if (key == "apply_overlap") {
            // k_vs_apply samples a chunk's few handed-over rows while its
            // other waves add up the moves (1, default) or before (0)
            DIST_REQUIRE(value == 0 || value == 1, "apply_overlap: 0 or 1");
            g->impl->apply_overlap_mode = value;
        } else if (key == "value_sorted") {
            DIST_REQUIRE(value >= 0 && value <= 2, "value_sorted: 0, 1 or 2");
            g->impl->value_sorted_mode = value;
        } else if (key == "value_stream") {
            // the table-free value-sorted kernel: 0 never, 1 auto, 2 always
            DIST_REQUIRE(value >= 0 && value <= 2, "value_stream: 0, 1 or 2");
            g->impl->value_stream_mode = value;
            // (cached ranges chose their apply chunks by it)
            g->impl->drop_overlapping_caches(0, g->impl->n_rows, false);
        } else if (key == "narrow_tiles") {
            // k_vs_narrow for launches that cannot fill the chip: 0 never,
            // 1 auto, 2 whenever the vectors fit its LDS
            DIST_REQUIRE(value >= 0 && value <= 2, "narrow_tiles: 0, 1 or 2");
            g->impl->narrow_mode = value;
            // (cached ranges carry their tile lists)
            g->impl->drop_overlapping_caches(0, g->impl->n_rows, false);
        } else if (key == "stream_scratch") {
            // k_vs_stream keeps the first pass's likelihoods for the second
            // in a scratch row per tile (1, default) or computes them again
            DIST_REQUIRE(value == 0 || value == 1, "stream_scratch: 0 or 1");
            g->impl->stream_scratch_mode = value;
        } else if (key == "kernel_timing") {
            // HIP events around the score+sample kernel of every n-th batch
            // feed dist_gibbs_kernel_stats: 1 (default) all, 0 none
            DIST_REQUIRE(value >= 0, "kernel_timing: >= 0");
            g->impl->kernel_timing = value;
        } else if (key == "narrow_read_ahead") {
            // k_vs_narrow's instance: 0 by launch size, 4 or 8 float4s
            DIST_REQUIRE(value == 0 || value == 4 || value == 8,
                         "narrow_read_ahead: 0, 4 or 8");
            g->impl->narrow_read_ahead = value;
        } else if (key == "device_normalise") {
            // sweeps whose batches all take the value-sorted path normalise
            // the group set on the device (no host round trip per batch):
            // 0 never, 1 where it applies (default; 2 is accepted as 1)
            DIST_REQUIRE(value >= 0 && value <= 2, "device_normalise: 0, 1 or 2");
            g->impl->device_normalise_mode = value;
        } else if (key == "sharded_device_normalise") {
            // dist_gibbs_sweep_sharded may normalise on the device: set on
            // EVERY rank or on none (engine.ShardedGibbs agrees on it)
            DIST_REQUIRE(value == 0 || value == 1,
                         "sharded_device_normalise: 0 or 1");
            g->impl->sharded_device_normalise = value != 0;
        } else if (key == "running_sums_min_tiles") {
            // launches of at least this many value tiles start each tile's
            // total from the per-value running sums (a tuning knob: results
            // do not depend on it)
            DIST_REQUIRE(value >= 0, "running_sums_min_tiles: >= 0");
            g->impl->running_sums_min_tiles = value;
        } else if (key == "rows_scratch") {
            // general rows: 3 k_rows_scratch (default), 0 k_sweep_program
            // (what feature lists beyond k_rows_scratch's table take anyway)
            DIST_REQUIRE(value == 0 || value == 3, "rows_scratch: 0 or 3");
            g->impl->rows_scratch_mode = value;
        } else if (key == "rows_scratch_lds_log") {
            DIST_REQUIRE(value == 0 || value == 1,
                         "rows_scratch_lds_log: 0 or 1");
            g->impl->rows_scratch_lds_log = value;
        } else if (key == "rows_scratch_block") {
            DIST_REQUIRE(value >= 64 && value <= kScratchMaxBlock
                             && value % 64 == 0,
                         "rows_scratch_block: a multiple of 64 up to 1024");
            g->impl->rows_scratch_block = value;
        } else if (key == "sampling") {
            DIST_REQUIRE(value == 0 || value == 1, "sampling: 0 exact, 1 scan");
            g->impl->sampling_mode = value;
        } else if (key == "phase_timing") {
            DIST_REQUIRE(value == 0 || value == 1, "phase_timing: 0 or 1");
            g->impl->phase_timing = value;
        } else if (key == "float_stats") {
            DIST_REQUIRE(value == 0 || value == 1,
                         "float_stats: 0 ordered, 1 merged");
            g->impl->float_stats_mode = value;
        } else if (key == "apply_stage") {
            // general rows' integer statistics: 1 (default) the whole image
            // in LDS and a staging matrix where it fits, 0 global atomics on
            // the categorical cells
            DIST_REQUIRE(value == 0 || value == 1, "apply_stage: 0 or 1");
            g->impl->apply_stage_mode = value;
        } else if (key == "rows_fold") {
            // general rows: the leading discrete features' scores from a
            // per-(joint value, group) table, rows sorted by joint value
            // (2: whenever the joint domain is no larger than the batch)
            DIST_REQUIRE(value >= 0 && value <= 2, "rows_fold: 0, 1 or 2");
            g->impl->rows_fold_mode = value;
        } else if (key == "program_all") {
            // 1 (default): every batch outside the value-sorted path is
            // scored by the program kernels when its tables exist
            DIST_REQUIRE(value == 0 || value == 1, "program_all: 0 or 1");
            g->impl->program_all = value;
        } else if (key == "fused_tables") {
            // device-normalised runs of the value-sorted path: 1 (default)
            // group set, caches and per-value tables in ONE launch between
            // two batches (k_vs_tables), the handed-over rows inside
            // k_vs_apply; 0 the separate launches
            DIST_REQUIRE(value == 0 || value == 1, "fused_tables: 0 or 1");
            g->impl->fused_tables_mode = value;
        } else if (key == "sequential_chain") {
            // 2 (default): k_chains, structural steps on the device; 1:
            // round 3's kernel (back to the host at every structural step);
            // 0: every row as a batch of one
            DIST_REQUIRE(value >= 0 && value <= 2, "sequential_chain: 0, 1 or 2");
            g->impl->sequential_mode = value;
        } else {
            throw Error("unknown option: " + key);
        }
    });
}
int dist_gibbs_path_counts(const dist_gibbs_t * g, uint64_t * value_sorted,
                           uint64_t * generic) {
    return guarded([&] {
        *value_sorted = g->impl.read()->vs_batches;
        *generic = g->impl.read()->generic_batches;
    });
}
int dist_gibbs_debug_counts(dist_gibbs_t * g, uint64_t * out, size_t n) {
    return guarded([&] {
        Gibbs & e = *g->impl.read();
        uint64_t v[16] = {e.vs_batches, e.generic_batches, e.band_batches,
                          e.prefix_batches, 0, 0, e.stream_batches,
                          e.async_batches, e.narrow_batches,
                          e.scratch_batches, e.fold_batches, e.scan_batches,
                          e.merged_batches, e.fused_batches, e.resumed_runs,
                          e.chain_launches};
        if (e.last_bands && !e.batch_open && e.vsBandMode.p) {
            // values whose arg-max group's rows had a tile of their own in
            // the last value-sorted launch
            std::vector<int> mode((size_t)e.vs_nvals());
            e.vsBandMode.download(mode.data(), mode.size());
            for (int m : mode) v[4] += m != 0;
        }
        if (e.deferred_count.p) {
            uint32_t d = 0;
            e.deferred_count.download(&d, 1);
            v[5] = d;
        }
        for (size_t i = 0; i < n && i < 16; ++i) out[i] = v[i];
    });
}
int dist_gibbs_phase_stats(dist_gibbs_t * g, double ms_out[5],
                           uint64_t * batches_out, int reset) {
    return guarded([&] {
        Gibbs & e = *g->impl.read();   // (settles an open run: its events are read)
        e.collect_comm_timing();
        for (int i = 0; i < Gibbs::kPhases; ++i) ms_out[i] = e.phase_ms[i];
        *batches_out = e.phase_batches;
        if (reset) {
            for (int i = 0; i < Gibbs::kPhases; ++i) e.phase_ms[i] = 0.0;
            e.phase_batches = 0;
        }
    });
}
int dist_gibbs_comm_stats(dist_gibbs_t * g, double * ms_out,
                          uint64_t * launches_out, int reset) {
    return guarded([&] {
        Gibbs & e = *g->impl.read();   // (settles an open run: its events are read)
        e.collect_comm_timing();
        if (ms_out) *ms_out = e.comm_ms;
        if (launches_out) *launches_out = e.comm_launches;
        if (reset) {
            e.comm_ms = 0.0;
            e.comm_launches = 0;
        }
    });
}
int dist_gibbs_kernel_stats(dist_gibbs_t * g, double * ms_out,
                            uint64_t * launches_out, uint64_t * rows_out,
                            int reset) {
    return guarded([&] {
        if (ms_out) *ms_out = g->impl.read()->kernel_ms;
        if (launches_out) *launches_out = g->impl.read()->kernel_launches;
        if (rows_out) *rows_out = g->impl.read()->kernel_rows;
        if (reset) {
            g->impl.read()->kernel_ms = 0.0;
            g->impl.read()->kernel_launches = 0;
            g->impl.read()->kernel_rows = 0;
        }
    });
}

}  // extern "C"
