// Protobuf wire codec for the Model::Shared / Model::Group messages of the
// reference's schema (distributions/io/schema.proto, package
// protobuf.distributions; proto2: repeated scalars are written unpacked, both
// encodings are accepted on input, unknown fields are skipped).  Host code
// only: lets a C/C++ caller checkpoint groups in the reference's on-wire
// format without libprotobuf.  Field numbers per message are cited below.
#include "common.h"

#include <cstring>

namespace {

using dist::Error;

struct Writer {
    uint8_t * buf;
    size_t cap;
    size_t len = 0;
    void byte(uint8_t b) {
        if (buf && len < cap) buf[len] = b;
        len += 1;
    }
    void varint(uint64_t v) {
        while (v >= 0x80) { byte((uint8_t)(v | 0x80)); v >>= 7; }
        byte((uint8_t)v);
    }
    void field_varint(int number, uint64_t v) {
        varint((uint64_t)number << 3 | 0);
        varint(v);
    }
    void field_float(int number, float f) {
        varint((uint64_t)number << 3 | 5);
        uint32_t u;
        memcpy(&u, &f, 4);
        for (int i = 0; i < 4; ++i) byte((uint8_t)(u >> (8 * i)));
    }
};

struct Reader {
    const uint8_t * p;
    const uint8_t * end;
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            DIST_REQUIRE(p < end, "protobuf: truncated varint");
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        throw Error("ERROR protobuf: varint too long");
    }
    float fixed32() {
        DIST_REQUIRE(end - p >= 4, "protobuf: truncated fixed32");
        uint32_t u = 0;
        for (int i = 0; i < 4; ++i) u |= (uint32_t)p[i] << (8 * i);
        p += 4;
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
    Reader sub() {
        const uint64_t n = varint();
        DIST_REQUIRE((uint64_t)(end - p) >= n, "protobuf: truncated field");
        Reader r{p, p + n};
        p += n;
        return r;
    }
    void skip(int wire_type) {
        switch (wire_type) {
        case 0: varint(); break;
        case 1: DIST_REQUIRE(end - p >= 8, "protobuf: truncated"); p += 8; break;
        case 2: sub(); break;
        case 5: DIST_REQUIRE(end - p >= 4, "protobuf: truncated"); p += 4; break;
        default: throw Error("ERROR protobuf: unsupported wire type");
        }
    }
};

// calls on_varint(number, value) / on_float(number, value) for every scalar,
// expanding packed repeated fields
template <class V, class F>
void parse(const uint8_t * data, size_t len, uint32_t float_fields,
           V && on_varint, F && on_float) {
    Reader r{data, data + len};
    while (!r.done()) {
        const uint64_t tag = r.varint();
        const int number = (int)(tag >> 3);
        const int wt = (int)(tag & 7);
        const bool is_float = number < 32 && (float_fields >> number & 1u);
        if (wt == 0 && !is_float) on_varint(number, r.varint());
        else if (wt == 5 && is_float) on_float(number, r.fixed32());
        else if (wt == 2) {   // packed
            Reader s = r.sub();
            while (!s.done()) {
                if (is_float) on_float(number, s.fixed32());
                else on_varint(number, s.varint());
            }
        } else r.skip(wt);
    }
}

float word_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
uint32_t float_word(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

void finish(const Writer & w, size_t * len_out) {
    *len_out = w.len;
    DIST_REQUIRE(!w.buf || w.len <= w.cap, "protobuf: buffer too small");
}

}  // namespace

extern "C" {

int dist_group_protobuf_dump(const dist_shared_t * shared,
                             const uint32_t * group, const uint32_t * keys,
                             uint8_t * buf, size_t cap, size_t * len_out) {
    return dist::guarded([&] {
        Writer w{buf, cap};
        switch (shared->kind) {
        case DIST_DD:     // DirichletDiscrete.Group: repeated uint64 counts = 1
            for (int v = 0; v < shared->dim; ++v)      // dd.hpp:104-111
                w.field_varint(1, group[1 + v]);
            break;
        case DIST_DPD:    // ...ProcessDiscrete.Group: keys = 1, values = 2
            for (int v = 0; v < shared->dim; ++v)      // dpd.hpp:171-180
                if (group[1 + v])
                    w.field_varint(1, keys ? keys[v] : (uint32_t)v);
            for (int v = 0; v < shared->dim; ++v)
                if (group[1 + v]) w.field_varint(2, group[1 + v]);
            break;
        case DIST_BB:     // BetaBernoulli.Group: heads = 1, tails = 2
        case DIST_BNB:    // BetaNegativeBinomial.Group: count = 1, sum = 2
            w.field_varint(1, group[0]);
            w.field_varint(2, group[1]);
            break;
        case DIST_GP:     // GammaPoisson.Group: count = 1, sum = 2, log_prod = 3
            w.field_varint(1, group[0]);
            w.field_varint(2, group[1]);
            w.field_float(3, word_float(group[2]));
            break;
        case DIST_NICH:   // NormalInverseChiSq.Group: count, mean, c_t_v
            w.field_varint(1, group[0]);
            w.field_float(2, word_float(group[1]));
            w.field_float(3, word_float(group[2]));
            break;
        default:
            throw Error("ERROR bad model kind");
        }
        finish(w, len_out);
    });
}

int dist_group_protobuf_load(const dist_shared_t * shared,
                             const uint32_t * keys, const uint8_t * data,
                             size_t len, uint32_t * group_out) {
    return dist::guarded([&] {
        const int kind = shared->kind;
        const size_t words = dist_group_words(shared);
        DIST_REQUIRE(words > 0, "bad model kind");
        memset(group_out, 0, 4 * words);
        int n_counts = 0;
        std::vector<uint32_t> dpd_keys;
        std::vector<uint64_t> dpd_values;
        const uint32_t float_fields =
            kind == DIST_GP ? 1u << 3 : kind == DIST_NICH ? 3u << 2 : 0u;
        parse(data, len, float_fields,
              [&](int number, uint64_t v) {
                  if (kind == DIST_DD && number == 1) {   // dd.hpp:94-102
                      DIST_REQUIRE(n_counts < shared->dim,
                                   "protobuf: more counts than dim");
                      group_out[1 + n_counts] = (uint32_t)v;
                      group_out[0] += (uint32_t)v;
                      n_counts += 1;
                  } else if (kind == DIST_DPD && number == 1) {
                      dpd_keys.push_back((uint32_t)v);
                  } else if (kind == DIST_DPD && number == 2) {
                      dpd_values.push_back(v);
                  } else if (kind != DIST_DD && kind != DIST_DPD
                             && number >= 1 && number <= 2) {
                      // BB heads/tails, GP count/sum, NICH count
                      if (!(kind == DIST_NICH && number == 2))
                          group_out[number - 1] = (uint32_t)v;
                  }
              },
              [&](int number, float f) {
                  if (kind == DIST_GP && number == 3) group_out[2] = float_word(f);
                  if (kind == DIST_NICH) group_out[number - 1] = float_word(f);
              });
        if (kind == DIST_DD)
            DIST_REQUIRE(n_counts == shared->dim, "protobuf: counts != dim");
        if (kind == DIST_DPD) {                            // dpd.hpp:161-169
            DIST_REQUIRE(dpd_keys.size() == dpd_values.size(),
                         "protobuf: keys != values");
            for (size_t i = 0; i < dpd_keys.size(); ++i) {
                int idx = -1;
                if (!keys) idx = (int)dpd_keys[i];
                else
                    for (int v = 0; v < shared->dim; ++v)
                        if (keys[v] == dpd_keys[i]) { idx = v; break; }
                DIST_REQUIRE(idx >= 0 && idx < shared->dim,
                             "protobuf: key is not a value of this Shared");
                group_out[1 + idx] += (uint32_t)dpd_values[i];
                group_out[0] += (uint32_t)dpd_values[i];
            }
        }
    });
}

int dist_shared_protobuf_dump(const dist_shared_t * shared, uint8_t * buf,
                              size_t cap, size_t * len_out) {
    return dist::guarded([&] {
        Writer w{buf, cap};
        switch (shared->kind) {
        case DIST_DD:     // Shared: repeated float alphas = 1
            for (int v = 0; v < shared->dim; ++v)
                w.field_float(1, shared->alphas[v]);
            break;
        case DIST_BB:     // alpha = 1, beta = 2
        case DIST_GP:     // alpha = 1, inv_beta = 2
            w.field_float(1, shared->p[0]);
            w.field_float(2, shared->p[1]);
            break;
        case DIST_NICH:   // mu = 1, kappa = 2, sigmasq = 3, nu = 4
            for (int i = 0; i < 4; ++i) w.field_float(1 + i, shared->p[i]);
            break;
        case DIST_BNB:    // alpha = 1, beta = 2, uint64 r = 3
            w.field_float(1, shared->p[0]);
            w.field_float(2, shared->p[1]);
            w.field_varint(3, (uint64_t)shared->p[2]);
            break;
        default:
            throw Error("ERROR Shared message of this model carries state the "
                        "dense remap does not hold (use the lp layer)");
        }
        finish(w, len_out);
    });
}

int dist_shared_protobuf_load(int kind, const uint8_t * data, size_t len,
                              dist_shared_t * shared_out) {
    return dist::guarded([&] {
        DIST_REQUIRE(kind == DIST_DD || kind == DIST_BB || kind == DIST_GP
                         || kind == DIST_NICH || kind == DIST_BNB,
                     "Shared message of this model is not supported here");
        memset(shared_out, 0, sizeof(*shared_out));
        shared_out->kind = kind;
        parse(data, len, kind == DIST_BNB ? 0x6u : 0x1eu,
              [&](int number, uint64_t v) {
                  if (kind == DIST_BNB && number == 3)
                      shared_out->p[2] = (float)v;
              },
              [&](int number, float f) {
                  if (kind == DIST_DD) {
                      if (number != 1) return;
                      DIST_REQUIRE(shared_out->dim < DIST_DD_MAX_DIM,
                                   "protobuf: more than 256 alphas");
                      shared_out->alphas[shared_out->dim++] = f;
                  } else if (number >= 1 && number <= 4) {
                      shared_out->p[number - 1] = f;
                  }
              });
    });
}

}  // extern "C"
