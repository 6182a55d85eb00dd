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
#include <vector>

#include "babybear.cuh"
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
inline void air_device_image(const AirView& a, const Ext& alpha, std::vector<uint32_t>& body, std::vector<uint32_t>& weights) {
    body.assign(a.w + 6, a.w + a.words);
    size_t p = 0;
    for (uint32_t k = 0; k < a.K; k++) {
        p++;                                   // selector
        const uint32_t nt = body[p++];
        for (uint32_t t = 0; t < nt; t++) { body[p] = to_monty(body[p]); p++; const uint32_t d = body[p++]; p += d; }
    }
    weights.resize(4 * (size_t)a.K);
    Ext w = ext_one();
    for (size_t k = a.K; k-- > 0;) { for (int i = 0; i < 4; i++) weights[4 * k + i] = w.c[i]; w = ext_mul(w, alpha); }
}

}  // namespace zk
