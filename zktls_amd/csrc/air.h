// air.h -- constraint programs: an AIR supplied as DATA.  Host side: validation, the
// program digest the transcript binds, the verifier's evaluation at zeta, and the device image the quotient kernel interprets
// (stark.hip, quotient_air_kernel).
//
// Upstream a constraint is a Rust `Air::eval` body that p3-uni-stark drives through its prover / verifier constraint folders
// (p3-air, p3-uni-stark 0.2.1-succinct: reference Cargo.lock:3835, 4055; sp1-stark :6172; behind sp1.rs:116): a polynomial in
// the local / next row, the public values and the selectors is_first_row / is_last_row / is_transition, folded as
// acc = acc * alpha + constraint.  A program is that polynomial written out in sum-of-products form, so any AIR of degree <= 5
// (log_quotient_degree 1 for degree <= 3 -- the bound of SP1's core machine --, 2 for degree 4 and 5) is proven without touching a kernel.
//
// Program (u32 words, canonical residues; format declared in include/zkhip.h):
//   [0] 0x50524941 "AIRP"  [1] 1  [2] width  [3] constraints K  [4] n_public  [5] total words
//   K x { selector (0 every row, 1 first row, 2 last row, 3 transition), n_terms, n_terms x { coeff, degree d <= 5, d variables } }
//   variable = kind << 30 | index;  kind 0 local row, 1 next row, 2 public value.  A selector counts one degree.
#pragma once
#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "babybear.cuh"
#include "kernels.h"
#include "poseidon2.cuh"

namespace zk {

constexpr uint32_t AIR_MAGIC = 0x50524941u;

struct AirView {
    const uint32_t* w = nullptr;
    size_t words = 0;
    uint32_t width = 0, K = 0, n_public = 0;
    int lqd = 1;             // log2 of the number of quotient chunks: degree <= 3 -> 1, degree 4 or 5 -> 2 (needs log_blowup >= 2)
};

inline bool air_validate(const uint32_t* prog, size_t words, uint32_t width, size_t n_public, AirView* out) {
    if (!prog || words < 6 || prog[0] != AIR_MAGIC || prog[1] != 1 || prog[2] != width || prog[4] != n_public || prog[5] != words) return false;
    if (prog[3] == 0 || prog[3] > (1u << 20)) return false;
    size_t p = 6;
    uint32_t maxd = 0;
    for (uint32_t k = 0; k < prog[3]; k++) {
        if (p + 2 > words) return false;
        const uint32_t sel = prog[p++], nt = prog[p++];
        if (sel > 3 || nt == 0) return false;
        for (uint32_t t = 0; t < nt; t++) {
            if (p + 2 > words) return false;
            const uint32_t coeff = prog[p++], d = prog[p++];
            if (coeff >= P || d > 5 || d + (sel ? 1u : 0u) > 5 || p + d > words) return false;
            if (d + (sel ? 1u : 0u) > maxd) maxd = d + (sel ? 1u : 0u);
            for (uint32_t j = 0; j < d; j++) {
                const uint32_t v = prog[p++], kind = v >> 30, idx = v & 0xFFFFu;
                if ((v & 0x3FFF0000u) || kind > 2) return false;
                if (kind == 2 ? idx >= n_public : idx >= width) return false;
            }
        }
    }
    if (p != words) return false;
    if (out) { out->w = prog; out->words = words; out->width = width; out->K = prog[3]; out->n_public = (uint32_t)n_public; out->lqd = maxd <= 3 ? 1 : 2; }
    return true;
}

// digest: the width-16 overwrite-mode sponge over the 16-bit halves of every word (halves are field elements whatever the word); canonical out
inline void air_digest(const AirView& a, uint32_t out[8]) {
    uint32_t st[16] = {0};
    int pos = 0;
    auto absorb = [&](uint32_t v) { st[pos++] = to_monty(v); if (pos == 8) { p2_permute(st); pos = 0; } };
    for (size_t i = 0; i < a.words; i++) { absorb(a.w[i] & 0xFFFFu); absorb(a.w[i] >> 16); }
    if (pos) p2_permute(st);
    for (int i = 0; i < 8; i++) out[i] = from_monty(st[i]);
}

// verifier side: the fold on opened (extension, Montgomery) values; pub = canonical public values
inline Ext air_fold_ext(const AirView& a, const Ext* local, const Ext* next, const uint32_t* pub, const Ext& sel_first, const Ext& sel_last,
                        const Ext& sel_trans, const Ext& alpha) {
    Ext acc = ext_zero();
    size_t p = 6;
    for (uint32_t k = 0; k < a.K; k++) {
        const uint32_t sel = a.w[p++], nt = a.w[p++];
        Ext c = ext_zero();
        for (uint32_t t = 0; t < nt; t++) {
            Ext prod = ext_from_base(to_monty(a.w[p++]));
            const uint32_t d = a.w[p++];
            for (uint32_t j = 0; j < d; j++) {
                const uint32_t v = a.w[p++], kind = v >> 30, idx = v & 0xFFFFu;
                prod = ext_mul(prod, kind == 0 ? local[idx] : (kind == 1 ? next[idx] : ext_from_base(to_monty(pub[idx]))));
            }
            c = ext_add(c, prod);
        }
        if (sel == 1) c = ext_mul(c, sel_first); else if (sel == 2) c = ext_mul(c, sel_last); else if (sel == 3) c = ext_mul(c, sel_trans);
        acc = ext_add(ext_mul(acc, alpha), c);
    }
    return acc;
}

// device image: the program with Montgomery coefficients and the public values resolved into constants is not possible for
// products (a public value is a factor), so variables keep their kinds; layout = the program body (from word 6) with
// coefficients converted to Montgomery form.  weights[k] = alpha^(K-1-k) (extension, Montgomery).
inline void air_device_image(const AirView& a, const Ext& alpha, std::vector<uint32_t>& body, std::vector<uint32_t>& weights, const Ext& scale = ext_one()) {
    body.assign(a.w + 6, a.w + a.words);
    size_t p = 0;
    for (uint32_t k = 0; k < a.K; k++) {
        p++;                                   // selector
        const uint32_t nt = body[p++];
        for (uint32_t t = 0; t < nt; t++) { body[p] = to_monty(body[p]); p++; const uint32_t d = body[p++]; p += d; }
    }
    weights.resize(4 * (size_t)a.K);
    Ext w = scale;                             // alpha^(number of constraints folded after the program's: a chip's lookups)
    for (size_t k = a.K; k-- > 0;) { for (int i = 0; i < 4; i++) weights[4 * k + i] = w.c[i]; w = ext_mul(w, alpha); }
}

// Term records for the term-parallel quotient kernel (stark.hip, quotient_air_terms_kernel): the program flattened into one
// record per TERM, the constraint's weight alpha^(K-1-k) folded into the coefficient (extension, Montgomery) and the constraint's
// selector appended as one more factor, so that the quotient numerator is one flat sum  sum_t coeff_t * prod_j slot[off_tj]  over
// a per-point array of slots: [0, W) local row, [W, 2W) next row, 2W is_first, 2W+1 is_last, 2W+2 is_transition, 2W+3 the constant 1,
// 2W+4+i public value i.  Record = 8 words: coefficient (4), off0 | off1 << 16, off2 | off3 << 16, off4 | nvars << 16, 0; nvars in
// [1, 5] (a constant term reads the slot of 1); off = the slot's LDS word for point 0 (kernels.h, air_lds_base).  Sorted by nvars, so that the lanes of a wavefront run the same number of products.
inline uint32_t air_slots(const AirView& a) { return 2 * a.width + AIR_SLOT_EXTRA + a.n_public; }
// Terms with the SAME monomial (the same factors, selector included) are merged: their weighted coefficients add up, the flat sum is
// the same field element.  Real chips repeat monomials across constraints -- the S-box output x3 x3 y of the Poseidon2 chip enters all
// sixteen elements of the next state: 6 130 terms, 1 008 monomials; the SHA-256 chip: 5 900 terms, 3 366 -- and a record costs the same
// whatever its coefficient.  Which terms share a monomial depends on the program alone, so that plan is built once per program
// (kept by content) and a proof only adds up weighted coefficients.
struct AirTermPlan {
    std::vector<std::array<uint32_t, 6>> monomials;    // n, then the n slot offsets in ascending order; sorted (by n first)
    struct Src { uint32_t monomial, constraint, coeff_monty; uint16_t n_pub, pub[5]; };      // (pub: the term's public-value factors -- constants of a proof, folded into its coefficient)
    std::vector<Src> terms;                            // every term of the program, in program order
};
inline std::shared_ptr<const AirTermPlan> air_term_plan(const AirView& a) {
    struct Entry { uint64_t hash; size_t words; std::vector<uint32_t> prog; std::shared_ptr<const AirTermPlan> plan; };
    static std::mutex mu;
    static std::vector<Entry> cache;                   // a handful of programs per process; newest last
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < a.words; i++) h = (h ^ a.w[i]) * 1099511628211ull;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Entry& e : cache)
            if (e.hash == h && e.words == a.words && std::memcmp(e.prog.data(), a.w, a.words * 4) == 0) return e.plan;
    }
    auto plan = std::make_shared<AirTermPlan>();
    std::map<std::array<uint32_t, 6>, uint32_t> index;
    std::vector<std::array<uint32_t, 6>> keys;
    size_t p = 6;
    const uint32_t W = a.width;
    for (uint32_t k = 0; k < a.K; k++) {
        const uint32_t sel = a.w[p++], nt = a.w[p++];
        for (uint32_t t = 0; t < nt; t++) {
            const uint32_t coeff = to_monty(a.w[p++]), d = a.w[p++];
            std::array<uint32_t, 6> key = {0, 0, 0, 0, 0, 0};
            uint32_t n = 0;
            AirTermPlan::Src src{0, k, coeff, 0, {0, 0, 0, 0, 0}};
            for (uint32_t j = 0; j < d; j++) {
                const uint32_t v = a.w[p++], kind = v >> 30, idx = v & 0xFFFFu;
                if (kind == 2) { src.pub[src.n_pub++] = (uint16_t)idx; continue; }      // a public value is no slot: it multiplies the coefficient (air_term_records)
                key[1 + n++] = kind == 0 ? idx : W + idx;
            }
            if (sel) key[1 + n++] = 2 * W + (sel - 1);
            if (n == 0) key[1 + n++] = 2 * W + 3;
            std::sort(key.begin() + 1, key.begin() + 1 + n);
            key[0] = n;
            auto it = index.find(key);
            if (it == index.end()) { it = index.emplace(key, (uint32_t)keys.size()).first; keys.push_back(key); }
            src.monomial = it->second;
            plan->terms.push_back(src);
        }
    }
    // records in sorted key order (by n first: the lanes of a wavefront run the same number of products)
    std::vector<uint32_t> order(keys.size()), rank(keys.size());
    for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return keys[x] < keys[y]; });
    for (uint32_t r = 0; r < order.size(); r++) { rank[order[r]] = r; plan->monomials.push_back(keys[order[r]]); }
    for (AirTermPlan::Src& t : plan->terms) t.monomial = rank[t.monomial];
    std::lock_guard<std::mutex> lk(mu);
    if (cache.size() >= 32) cache.erase(cache.begin());
    cache.push_back(Entry{h, a.words, std::vector<uint32_t>(a.w, a.w + a.words), plan});
    return plan;
}
// public values a program READS: 1 + the largest index among its public-value factors (0: none).  A machine's chips all declare the
// machine's public values, most read none -- the term kernels stage only what is read.
inline uint32_t air_public_used(const AirView& a) {
    uint32_t used = 0;
    size_t p = 6;
    for (uint32_t k = 0; k < a.K; k++) {
        const uint32_t nt = a.w[p + 1];
        p += 2;
        for (uint32_t t = 0; t < nt; t++) {
            const uint32_t d = a.w[p + 1];
            p += 2;
            for (uint32_t j = 0; j < d; j++, p++) if ((a.w[p] >> 30) == 2u && (a.w[p] & 0xFFFFu) + 1u > used) used = (a.w[p] & 0xFFFFu) + 1u;
        }
    }
    return used;
}
inline size_t air_term_count(const AirView& a) { const size_t nm = air_term_plan(a)->monomials.size(); return nm + (nm & 1); }
// (pub_monty: the proof's public values, Montgomery form -- every record's coefficient carries its terms' public-value factors, so
// that the kernels stage no public value at all: the shard verifier's transcript table reads 64 x 91 of them)
inline uint32_t air_term_coeff(const AirTermPlan::Src& t, const uint32_t* pub_monty) {
    uint32_t c = t.coeff_monty;
    for (uint32_t j = 0; j < t.n_pub; j++) c = fmul(c, pub_monty[t.pub[j]]);
    return c;
}
inline void air_term_records(const AirView& a, const Ext& alpha, const uint32_t* pub_monty, std::vector<uint32_t>& recs, const Ext& scale = ext_one()) {
    std::vector<Ext> wts(a.K);
    Ext w = scale;
    for (size_t k = a.K; k-- > 0;) { wts[k] = w; w = ext_mul(w, alpha); }
    const std::shared_ptr<const AirTermPlan> plan = air_term_plan(a);
    const uint32_t W = a.width;
    const size_t nm = plan->monomials.size();
    recs.assign((nm + (nm & 1)) * 8, 0u);
    for (size_t m = 0; m < nm; m++) {                // a factor travels as its LDS word for point 0 (kernels.h, air_lds_base)
        const std::array<uint32_t, 6>& k = plan->monomials[m];
        uint32_t* r = recs.data() + 8 * m;
        uint32_t o[5] = {0, 0, 0, 0, 0};
        for (uint32_t j = 0; j < k[0]; j++) o[j] = air_lds_base(k[1 + j], W);
        r[4] = o[0] | o[1] << 16; r[5] = o[2] | o[3] << 16; r[6] = o[4] | k[0] << 16;
    }
    for (const AirTermPlan::Src& t : plan->terms) {
        uint32_t* r = recs.data() + 8 * (size_t)t.monomial;
        const Ext c = ext_mul_base(wts[t.constraint], air_term_coeff(t, pub_monty));
        for (int i = 0; i < 4; i++) r[i] = fadd(r[i], c.c[i]);
    }
    if (nm & 1) {                                // the kernel takes terms in pairs: pad with 0 * (the constant 1)
        uint32_t* r = recs.data() + 8 * nm;
        r[4] = air_lds_base(2 * W + 3, W); r[6] = 1u << 16;
    }
}

// The records of the WIDE kernel (stark.hip, quotient_air_wide_kernel: a lane per point, the wavefront walks one record): the same
// monomials, grouped by their number of factors -- cls[n - 1] .. cls[n] are the records of n factors, every class padded to an even
// count (the kernel takes records in pairs) with a zero-weight record -- so that the kernel runs one branch-free loop per class.  A
// factor travels as the LDS word of its column in the kernel's column-major tile (kernels.h, air_wide_word).
inline void air_term_records_wide(const AirView& a, const Ext& alpha, const uint32_t* pub_monty, std::vector<uint32_t>& recs, uint32_t cls[6], const Ext& scale = ext_one()) {
    std::vector<Ext> wts(a.K);
    Ext w = scale;
    for (size_t k = a.K; k-- > 0;) { wts[k] = w; w = ext_mul(w, alpha); }
    const std::shared_ptr<const AirTermPlan> plan = air_term_plan(a);
    const uint32_t W = a.width, one = air_wide_word(2 * W + 3, W);
    const size_t nm = plan->monomials.size();
    std::vector<uint32_t> pos(nm);
    recs.clear();
    recs.reserve((nm + 5) * 8);
    uint32_t cur = 1;
    cls[0] = 0;
    auto pad_to_even = [&](uint32_t n) {
        if ((recs.size() / 8) & 1) {
            uint32_t r[8] = {0, 0, 0, 0, one | one << 16, one | one << 16, one | n << 16, 0};
            recs.insert(recs.end(), r, r + 8);
        }
    };
    for (size_t m = 0; m < nm; m++) {                // monomials come sorted by their factor count
        const std::array<uint32_t, 6>& k = plan->monomials[m];
        while (cur < k[0]) { pad_to_even(cur); cls[cur] = (uint32_t)(recs.size() / 8); cur++; }
        uint32_t o[5] = {one, one, one, one, one};
        for (uint32_t j = 0; j < k[0]; j++) o[j] = air_wide_word(k[1 + j], W);
        pos[m] = (uint32_t)(recs.size() / 8);
        const uint32_t r[8] = {0, 0, 0, 0, o[0] | o[1] << 16, o[2] | o[3] << 16, o[4] | k[0] << 16, 0};
        recs.insert(recs.end(), r, r + 8);
    }
    while (cur <= 5) { pad_to_even(cur); cls[cur] = (uint32_t)(recs.size() / 8); cur++; }
    for (const AirTermPlan::Src& t : plan->terms) {
        uint32_t* r = recs.data() + 8 * (size_t)pos[t.monomial];
        const Ext c = ext_mul_base(wts[t.constraint], air_term_coeff(t, pub_monty));
        for (int i = 0; i < 4; i++) r[i] = fadd(r[i], c.c[i]);
    }
}

// ---- lookups as data: a chip's interaction table (machine mode; format declared in include/zkhip.h) ------------------------------
//   [0] 0x50554B4C "LKUP"  [1] interactions I (1..64)  [2] total words
//   I x { sign (0 send, 1 receive), multiplicity (0xFFFFFFFF: the constant 1, else a column), bus (canonical), values V (1..8), V columns }
// Fingerprint d = gamma + bus + sum_t beta^(t+1) v_t; one extension column phi_j per pair of interactions (2j, 2j+1):
// phi_j = s_a m_a / d_a + s_b m_b / d_b; then the running sum.  Constraints (folded after the program's): per column
// phi_j d_a d_b - (s_a m_a d_b + s_b m_b d_a); is_first (S - sum phi); is_transition (S' - S - sum phi'); is_last (S - C).
// Upstream: sp1-stark's generate_permutation_trace / eval_permutation_constraints with batch size 2 (reference Cargo.lock:6172).
constexpr uint32_t LOOKUP_MAGIC = 0x50554B4Cu;
struct LookupView {
    const uint32_t* w = nullptr;
    size_t words = 0;
    uint32_t ni = 0, cols = 0;                 // interactions, extension columns = ceil(ni / 2)
};
inline bool lookup_validate(const uint32_t* t, size_t words, uint32_t width, LookupView* out) {
    if (!t || words < 3 || t[0] != LOOKUP_MAGIC || t[1] < 1 || t[1] > 64 || t[2] != words) return false;
    size_t p = 3;
    for (uint32_t i = 0; i < t[1]; i++) {
        if (p + 4 > words) return false;
        const uint32_t sign = t[p], mult = t[p + 1], bus = t[p + 2], nv = t[p + 3];
        p += 4;
        if (sign > 1 || (mult != 0xFFFFFFFFu && mult >= width) || bus >= P || nv < 1 || nv > 8 || p + nv > words) return false;
        for (uint32_t v = 0; v < nv; v++) if (t[p + v] >= width) return false;
        p += nv;
    }
    if (p != words) return false;
    if (out) { out->w = t; out->words = words; out->ni = t[1]; out->cols = (t[1] + 1) / 2; }
    return true;
}
// device records (kernels.h LOOKUP_REC_WORDS): bus in Montgomery form, columns padded to 8
inline void lookup_device_records(const LookupView& v, std::vector<uint32_t>& recs) {
    recs.assign((size_t)v.ni * LOOKUP_REC_WORDS, 0u);
    size_t p = 3;
    for (uint32_t i = 0; i < v.ni; i++) {
        uint32_t* r = recs.data() + (size_t)i * LOOKUP_REC_WORDS;
        r[0] = v.w[p]; r[1] = v.w[p + 1]; r[2] = to_monty(v.w[p + 2]); r[3] = v.w[p + 3];
        for (uint32_t k = 0; k < r[3]; k++) r[4 + k] = v.w[p + 4 + k];
        p += 4 + r[3];
    }
}
// verifier side: the lookup constraints on opened (extension) values, folded onto acc
inline Ext lookup_fold_ext(Ext acc, const LookupView& v, const Ext* row, const Ext* pl, const Ext* pn, const Ext& gamma, const Ext& beta,
                           const Ext& sel_first, const Ext& sel_trans, const Ext& sel_last, const Ext& alpha, const Ext& cumsum) {
    Ext bpow[10];
    bpow[0] = ext_one();
    for (int t = 1; t < 10; t++) bpow[t] = ext_mul(bpow[t - 1], beta);
    auto finger = [&](size_t p) {
        Ext d = ext_add_base(gamma, to_monty(v.w[p + 2]));
        for (uint32_t k = 0; k < v.w[p + 3]; k++) d = ext_add(d, ext_mul(bpow[k + 1], row[v.w[p + 4 + k]]));
        return d;
    };
    auto mult = [&](size_t p) {
        const Ext m = v.w[p + 1] == 0xFFFFFFFFu ? ext_one() : row[v.w[p + 1]];
        return v.w[p] ? ext_neg(m) : m;
    };
    Ext sum_l = ext_zero(), sum_n = ext_zero();
    size_t p = 3;
    for (uint32_t j = 0; j < v.cols; j++) {
        const Ext da = finger(p), ma = mult(p);
        p += 4 + v.w[p + 3];
        Ext c;
        if (2 * j + 1 < v.ni) {
            const Ext db = finger(p), mb = mult(p);
            p += 4 + v.w[p + 3];
            c = ext_sub(ext_mul(ext_mul(pl[j], da), db), ext_add(ext_mul(ma, db), ext_mul(mb, da)));
        } else c = ext_sub(ext_mul(pl[j], da), ma);
        acc = ext_add(ext_mul(acc, alpha), c);
        sum_l = ext_add(sum_l, pl[j]);
        sum_n = ext_add(sum_n, pn[j]);
    }
    const Ext S = pl[v.cols], Sn = pn[v.cols];
    acc = ext_add(ext_mul(acc, alpha), ext_mul(sel_first, ext_sub(S, sum_l)));
    acc = ext_add(ext_mul(acc, alpha), ext_mul(sel_trans, ext_sub(ext_sub(Sn, S), sum_n)));
    acc = ext_add(ext_mul(acc, alpha), ext_mul(sel_last, ext_sub(S, cumsum)));
    return acc;
}

}  // namespace zk
