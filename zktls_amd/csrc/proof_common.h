// proof_common.h -- what the provers (prover.cpp) and the host verifiers (verifier.cpp) share: the Fiat-Shamir transcript, the shape
// parameters, proof sizes and header checks of the single-matrix proofs and of the multi-chip / machine / keyed-machine proofs
// (DESIGN.md sections 3 and 6).  Host code only.
#pragma once
#include <atomic>
#include <cstring>
#include <string>
#include <vector>

#include "context.h"
#include "poseidon2.cuh"
#include "air.h"
#include "batch.h"
#include "p2_x16.h"

#define CHECK_CTX(ctx)                                                  \
    do {                                                                \
        if (!(ctx)) return fail(ZKHIP_ERR_INVALID, "null context");     \
        ZK_HIP(hipSetDevice((ctx)->device));                            \
    } while (0)

namespace zk {

// ---------------------------------------------------------------- transcript (host)
// p3-challenger DuplexChallenger<16, 8> over Poseidon2; values in Montgomery form.
struct Challenger {
    uint32_t state[16] = {0};
    uint32_t in[8] = {0};
    int n_in = 0;
    uint32_t out[8] = {0};
    int n_out = 0;
    void duplex() {
        for (int i = 0; i < n_in; i++) state[i] = in[i];
        n_in = 0;
        p2_permute(state);
        for (int i = 0; i < 8; i++) out[i] = state[i];
        n_out = 8;
    }
    void observe(uint32_t m) {
        n_out = 0;
        in[n_in++] = m;
        if (n_in == 8) duplex();
    }
    void observe_canonical(uint32_t c) { observe(to_monty(c)); }
    void observe_ext(const Ext& e) { for (int i = 0; i < 4; i++) observe(e.c[i]); }
    uint32_t sample() {
        if (n_in != 0 || n_out == 0) duplex();
        return out[--n_out];
    }
    Ext sample_ext() { Ext e; for (int i = 0; i < 4; i++) e.c[i] = sample(); return e; }
    uint32_t sample_bits(int bits) { return from_monty(sample()) & ((1u << bits) - 1u); }
};

// n words into the transcript -- what `for (i) ch.observe(w[i])` does.  Inside a lock-step batch (batch.h) the members reach this point together, each with the
// opened values of ITS proof (5 500 words for the SHA-256 machine: 690 permutations, 0.8 ms of a core per member, one member after the other on the lane's
// thread): the lane absorbs them side by side, sixteen sponges per AVX-512 permutation (p2_x16.h).  Same states, same bytes.
static inline void challenger_observe_merged(void* const* objs, const uint32_t* const* words, size_t n, int cnt) {
    bool same = cnt > 1 && p2x16_available();
    for (int m = 1; same && m < cnt; m++) same = ((Challenger*)objs[m])->n_in == ((Challenger*)objs[0])->n_in;
    if (!same) {
        for (int m = 0; m < cnt; m++) { Challenger& ch = *(Challenger*)objs[m]; for (size_t i = 0; i < n; i++) ch.observe(words[m][i]); }
        return;
    }
    for (int g = 0; g < cnt; g += 16) {
        const int k = cnt - g < 16 ? cnt - g : 16;
        Challenger* ch[16];
        for (int m = 0; m < 16; m++) ch[m] = (Challenger*)objs[g + (m < k ? m : 0)];
        uint32_t st[16][16];
        for (int e = 0; e < 16; e++) for (int m = 0; m < 16; m++) st[e][m] = ch[m]->state[e];
        int n_in = ch[0]->n_in;
        bool fresh = false;                                  // the last word completed a block: out = the new state's first half
        for (size_t i = 0; i < n; i++) {
            for (int m = 0; m < k; m++) ch[m]->in[n_in] = words[g + m][i];
            n_in++;
            fresh = false;
            if (n_in == 8) {
                for (int e = 0; e < 8; e++) for (int m = 0; m < 16; m++) st[e][m] = ch[m]->in[e];
                p2x16_permute(st);
                n_in = 0;
                fresh = true;
            }
        }
        for (int m = 0; m < k; m++) {
            Challenger& c = *ch[m];
            for (int e = 0; e < 16; e++) c.state[e] = st[e][m];
            c.n_in = n_in;
            c.n_out = 0;
            if (fresh) { for (int e = 0; e < 8; e++) c.out[e] = c.state[e]; c.n_out = 8; }
        }
    }
}
static inline void observe_words(Challenger& ch, const uint32_t* w, size_t n) {
    if (t_batcher && n >= 64) { t_batcher->host_merge(&challenger_observe_merged, &ch, w, n); return; }
    for (size_t i = 0; i < n; i++) ch.observe(w[i]);
}

// shape parameters with their defaults resolved (0 = the SP1 shape)
struct Shape { int b = 1, K = 1, F = 0, hw = 16, R = 0; bool ext = false; uint32_t cw = 0; };
static bool shape_of(int log_n, const zkhip_params* prm, Shape& sh) {
    sh.b = prm->log_blowup;
    sh.K = prm->log_fold ? prm->log_fold : 1;
    sh.F = prm->log_final;
    sh.hw = prm->hash_width ? prm->hash_width : 16;
    sh.cw = (uint32_t)prm->code_width;           // code / data group split (callers check it against the width)
    sh.ext = !(sh.b == 1 && sh.K == 1 && sh.F == 0 && sh.hw == 16) || sh.cw != 0;
    if (prm->code_width < 0 || prm->code_width % 4 != 0) return false;
    if (sh.b < 1 || sh.b > 3 || sh.K < 1 || sh.K > 5 || sh.F < 0 || sh.F > 10 || sh.F > log_n || (log_n - sh.F) % sh.K != 0) return false;
    if (sh.hw != 16 && sh.hw != 24) return false;
    sh.R = (log_n - sh.F) / sh.K;
    return true;
}

// The program digest is a sponge over every 16-bit half of the program (4 500 permutations for the 18 000-word SHA-256 chip: 6.5 ms
// on the host), needed by the header and the transcript of every proof and verification: the last few programs' digests are kept,
// keyed by the program's full contents (an exact comparison, ~5 us for that program).
void air_digest_cached(const AirView& a, uint32_t out[8]);      // prover.cpp: the program digest, the last few programs kept


// `air`: the constraint program in effect (air.h) or null for the built-in synthetic AIR.  With a program the header always has
// the extended form and the 8-word program digest follows it (proof version 7), all of it observed.
static void transcript_init(Challenger& ch, int log_n, uint32_t width, const zkhip_params* prm, size_t n_public, const Shape& sh,
                            const AirView* air = nullptr) {
    ch.observe_canonical((uint32_t)log_n);
    ch.observe_canonical(width);
    ch.observe_canonical((uint32_t)prm->log_blowup);
    ch.observe_canonical((uint32_t)prm->num_queries);
    ch.observe_canonical((uint32_t)prm->pow_bits);
    ch.observe_canonical((uint32_t)n_public);
    if (air) {
        ch.observe_canonical((uint32_t)prm->logup_pairs);
        ch.observe_canonical((uint32_t)sh.K);
        ch.observe_canonical((uint32_t)sh.F);
        ch.observe_canonical((uint32_t)sh.hw);
        uint32_t dg[8];
        air_digest_cached(*air, dg);
        for (int i = 0; i < 8; i++) ch.observe_canonical(dg[i]);
        return;
    }
    if (sh.ext) {
        ch.observe_canonical((uint32_t)prm->logup_pairs);
        ch.observe_canonical((uint32_t)sh.K);
        ch.observe_canonical((uint32_t)sh.F);
        ch.observe_canonical((uint32_t)sh.hw);
    } else if (prm->logup_pairs) ch.observe_canonical((uint32_t)prm->logup_pairs);
    if (sh.cw) ch.observe_canonical(sh.cw);
}

constexpr uint32_t PROOF_MAGIC = 0x41544B5Au;   // "ZKTA"
constexpr uint32_t PROOF_VERSION = 1u;

// workspace roles: enum Slot in context.h

static int pow2ceil(int v) { int r = 1; while (r < v) r <<= 1; return r; }

static size_t proof_words(int log_n, uint32_t width, const zkhip_params* prm, bool air = false, int lqd = 1) {
    Shape sh;
    if (!shape_of(log_n, prm, sh)) return 0;
    const size_t H = (size_t)(log_n + sh.b);
    const size_t Q = (size_t)prm->logup_pairs, wp = Q ? 4 * (Q + 1) : 0;
    const size_t QW = (size_t)4 << lqd;          // width of the quotient matrix: 4 base columns per chunk
    const size_t CW = air ? 0 : sh.cw;          // code / data split: one more header word, root, and path per query
    size_t words = (air ? 20 : (sh.ext ? 12 : (Q ? 9 : 8))) + (CW ? 9 : 0) + 16 + 8 * (size_t)width + 4 * QW + 8 * (size_t)sh.R + 4 * ((size_t)1 << sh.F) + 1;
    size_t perq = width + QW + 16 * H + (CW ? 8 * H : 0);
    if (Q) { words += 8 + 8 * wp; perq += wp + 8 * H; }
    for (int l = 0; l < sh.R; l++) perq += 4 * (((size_t)1 << sh.K) - 1) + 8 * (H - (size_t)sh.K * (l + 1));
    return words + (size_t)prm->num_queries * perq;
}

static int check_shape(int log_n, uint32_t width, const zkhip_params* prm) {
    if (!prm) return fail(ZKHIP_ERR_INVALID, "null params");
    if (log_n < 5 || log_n > MAX_LOG_ROWS) return fail(ZKHIP_ERR_INVALID, "log_n must be in [5, 22]");
    if (width == 0 || width % 4 != 0 || width > 1024) return fail(ZKHIP_ERR_INVALID, "width must be a positive multiple of 4, at most 1024");
    Shape sh;
    if (!shape_of(log_n, prm, sh))
        return fail(ZKHIP_ERR_INVALID, "shape: log_blowup in [1,3], log_fold in [1,5] dividing log_n - log_final, log_final in [0,10], hash_width 16 or 24");
    if (prm->num_queries < 1 || prm->num_queries > 4096) return fail(ZKHIP_ERR_INVALID, "num_queries out of range");
    if (prm->pow_bits < 0 || prm->pow_bits > 28) return fail(ZKHIP_ERR_INVALID, "pow_bits out of range");
    if (prm->logup_pairs < 0 || prm->logup_pairs > 64 || (uint32_t)prm->logup_pairs * 8 > width)
        return fail(ZKHIP_ERR_INVALID, "logup_pairs out of range (each pair needs two column groups, at most 64 pairs)");
    if (prm->code_width && (uint32_t)prm->code_width >= width) return fail(ZKHIP_ERR_INVALID, "code_width must be a multiple of 4 below the width");
    return ZKHIP_OK;
}

// in-place inverse DFT of 2^log extension elements (natural order in and out); host, tiny sizes
static void host_intt_ext(std::vector<Ext>& a, int log) {
    const size_t nn = (size_t)1 << log;
    for (size_t i = 0; i < nn; i++) { size_t j = reverse_bits((uint32_t)i, log); if (i < j) std::swap(a[i], a[j]); }
    for (int s = 1; s <= log; s++) {
        const size_t half = (size_t)1 << (s - 1);
        const uint32_t wl = finv(two_adic_generator(s));
        for (size_t base = 0; base < nn; base += 2 * half) {
            uint32_t w = MONTY_R1;
            for (size_t j = 0; j < half; j++) {
                const Ext u = a[base + j], v = ext_mul_base(a[base + j + half], w);
                a[base + j] = ext_add(u, v);
                a[base + j + half] = ext_sub(u, v);
                w = fmul(w, wl);
            }
        }
    }
    const uint32_t ninv = finv(to_monty((uint32_t)nn));
    for (size_t i = 0; i < nn; i++) a[i] = ext_mul_base(a[i], ninv);
}


// ================================================================ shards of several chips with different heights
// The structure of an SP1 shard (sp1-stark 4.1.4 ShardProof, reference Cargo.lock:6172, behind sp1.rs:116): one Merkle
// commitment per phase over matrices of different heights (p3-merkle-tree injection rule), one zeta, one reduced-opening
// vector per height that joins the FRI vector when folding reaches that height (p3-fri 0.2.1 TwoAdicFriPcs), one query
// index with chip c opened at index >> (Hmax - h_c).  Byte layout: DESIGN.md section 6.
constexpr uint32_t CHIPS_VERSION = 4u, CHIPS_VERSION_LOGUP = 5u, CHIPS_VERSION_CROSS = 6u, CHIPS_VERSION_AIR = 9u, CHIPS_VERSION_MACHINE = 10u,
                   CHIPS_VERSION_KEYED = 11u;
constexpr int MAX_CHIPS = 32;

// What a multi-chip call proves beside the plain chips, handed down explicitly from the entry point that parsed it:
//   air      the chips' constraint programs (nullptr, or nullptr per chip: the built-in synthetic AIR).  Version 9: each chip's header
//            entry gains a has-program flag, the programs' digests follow the entries.
//   machine  machine mode (zkhip_*_machine, proof version 10): every chip runs through a program (its own, or the synthetic AIR
//            written as one: has_prog says which, for the header) and may bring an interaction table (air.h, LookupView).  The
//            number of extension columns of its permutation trace travels in the `pairs` slot of the chip arrays, so the layout
//            code of versions 5 / 6 serves unchanged.
//   key      keyed machine (zkhip_*_machine_keyed, proof version 11): chips with PREPROCESSED columns, committed once by
//            zkhip_machine_setup -- sp1-stark's StarkMachine::setup, which the reference calls before every prove
//            (crates/guest-prover-sp1/src/sp1.rs:113).  pw[c] is chip c's preprocessed width (0: none); its program and interaction
//            table address the combined row [preprocessed | main].  The prover side carries the key's device data, the verifier
//            side only the widths and the root.
struct MachineTables { const LookupView* lk[32]; bool has_prog[32]; };
struct KeyView {
    uint32_t pw[32];
    uint32_t root_m[8];                 // the key's commitment, Montgomery
    const uint32_t* d_trace[32];        // prover: preprocessed traces [2^log_n][pw], their LDEs [2^(log_n + b)][pw], the mixed-height tree
    const uint32_t* d_lde[32];
    const uint32_t* d_tree;
    int He;                             // height of the tallest preprocessed LDE = height of the key's tree
};
struct ChipSet {
    const AirView* const* air = nullptr;
    const MachineTables* machine = nullptr;
    const KeyView* key = nullptr;
};
static bool any_prog(const ChipSet& cs, int n) { if (cs.air) for (int c = 0; c < n; c++) if (cs.air[c]) return true; return false; }
static const AirView* prog_of(const ChipSet& cs, int c) { return cs.air ? cs.air[c] : nullptr; }
static const LookupView* lookup_of(const ChipSet& cs, int c) { return cs.machine ? cs.machine->lk[c] : nullptr; }
static bool header_has_prog(const ChipSet& cs, int c) { return cs.machine ? cs.machine->has_prog[c] : prog_of(cs, c) != nullptr; }
// log2 of chip c's number of quotient chunks: 2 for a program of degree 4 or 5 (needs log_blowup >= 2), else 1; the chip's quotient matrix
// has 4 * 2^lq columns.  The header's has-program word carries it: 0 = no program, else the program's log_quotient_degree.
static int lq_of(const ChipSet& cs, int c) { return prog_of(cs, c) ? prog_of(cs, c)->lqd : 1; }
static size_t qw_of(const ChipSet& cs, int c) { return (size_t)4 << lq_of(cs, c); }
static uint32_t header_prog_word(const ChipSet& cs, int c) { return header_has_prog(cs, c) ? (uint32_t)lq_of(cs, c) : 0u; }
static uint32_t pre_w(const ChipSet& cs, int c) { return cs.key ? cs.key->pw[c] : 0u; }
static void lookup_digest(const LookupView& v, uint32_t out[8]) {      // the program-digest sponge over the table's words (cached alike)
    AirView a;
    a.w = v.w; a.words = v.words;
    air_digest_cached(a, out);
}

static bool any_pairs(const int32_t* pairs, int n) { if (pairs) for (int c = 0; c < n; c++) if (pairs[c]) return true; return false; }
static size_t perm_width(const int32_t* pairs, int c) { return (pairs && pairs[c]) ? 4 * ((size_t)pairs[c] + 1) : 0; }
static bool any_cross(const ChipSet& cs, const int32_t* partners, int n) {
    if (cs.machine) { for (int c = 0; c < n; c++) if (cs.machine->lk[c]) return true; return false; }     // machine mode: the sums are always exposed
    if (partners) for (int c = 0; c < n; c++) if (partners[c] >= 0) return true;
    return false;
}
static uint32_t chips_version(const ChipSet& cs, const int32_t* pairs, const int32_t* partners, int n) {
    if (cs.machine) return cs.key ? CHIPS_VERSION_KEYED : CHIPS_VERSION_MACHINE;
    if (any_prog(cs, n)) return CHIPS_VERSION_AIR;
    return any_cross(cs, partners, n) ? CHIPS_VERSION_CROSS : (any_pairs(pairs, n) ? CHIPS_VERSION_LOGUP : CHIPS_VERSION);
}
static int check_chips(const ChipSet& cs, const int32_t* log_ns, const uint32_t* widths, const int32_t* pairs, const int32_t* partners, int n, const zkhip_params* prm) {
    if (!prm || !log_ns || !widths) return fail(ZKHIP_ERR_INVALID, "chips: null argument");
    if (n < 1 || n > MAX_CHIPS) return fail(ZKHIP_ERR_INVALID, "chips: 1..32 chips");
    if (prm->log_blowup < 1 || prm->log_blowup > 3) return fail(ZKHIP_ERR_INVALID, "chips: log_blowup in [1,3]");
    if ((prm->log_fold != 0 && prm->log_fold != 1) || prm->log_final != 0 || (prm->hash_width != 0 && prm->hash_width != 16) || prm->logup_pairs != 0 || prm->code_width != 0)
        return fail(ZKHIP_ERR_INVALID, "chips: the multi-chip prover uses the SP1 FRI shape (fold by 2, constant final polynomial, width-16 hash) without lookups");
    if (prm->num_queries < 1 || prm->num_queries > 4096 || prm->pow_bits < 0 || prm->pow_bits > 28) return fail(ZKHIP_ERR_INVALID, "chips: queries / pow_bits out of range");
    if (!cs.machine && any_prog(cs, n) && (any_pairs(pairs, n) || any_cross(cs, partners, n))) return fail(ZKHIP_ERR_INVALID, "chips: no lookups next to constraint programs (use the machine entries)");
    for (int c = 0; c < n; c++) {
        if (log_ns[c] < 5 || log_ns[c] > MAX_LOG_ROWS || widths[c] == 0 || widths[c] % 4 != 0 || widths[c] > 1024)
            return fail(ZKHIP_ERR_INVALID, "chips: log_n in [5,22], width a multiple of 4 up to 1024");
        if (c && log_ns[c] > log_ns[c - 1]) return fail(ZKHIP_ERR_INVALID, "chips: tallest first");
        if (pairs && (pairs[c] < 0 || pairs[c] > 64 || (!cs.machine && (uint32_t)pairs[c] * 8 > widths[c]))) return fail(ZKHIP_ERR_INVALID, "chips: logup_pairs out of range");
        if (partners && partners[c] >= 0) {
            const int d = partners[c];
            if (!pairs || d >= n || d == c || partners[d] != c || pairs[c] == 0 || pairs[d] != pairs[c] || log_ns[d] != log_ns[c])
                return fail(ZKHIP_ERR_INVALID, "chips: partners must be mutual, of equal height and pair count");
        } else if (partners && partners[c] < -1) return fail(ZKHIP_ERR_INVALID, "chips: bad partner index");
        int same = 0;
        for (int d = 0; d < n; d++) same += log_ns[d] == log_ns[c];
        if (same > MAX_LEAF_MATS) return fail(ZKHIP_ERR_INVALID, "chips: at most 8 chips per height");
    }
    for (int c = 0; c < n; c++)
        if (lq_of(cs, c) > prm->log_blowup) return fail(ZKHIP_ERR_INVALID, "chips: a program of degree 4 or 5 needs log_blowup >= 2 (its quotient domain must lie inside the committed LDE domain)");
    if (cs.key) {
        bool some = false;
        for (int c = 0; c < n; c++) {
            const uint32_t pw = pre_w(cs, c);
            if (pw % 4 != 0 || pw + widths[c] > 1024) return fail(ZKHIP_ERR_INVALID, "keyed machine: preprocessed width a multiple of 4, preprocessed + main columns at most 1024");
            if (pw && !header_has_prog(cs, c)) return fail(ZKHIP_ERR_INVALID, "keyed machine: a chip with preprocessed columns brings its own program");
            some = some || pw != 0;
        }
        if (!some) return fail(ZKHIP_ERR_INVALID, "keyed machine: no chip has preprocessed columns (use the plain machine entries)");
    }
    return ZKHIP_OK;
}
static size_t chips_proof_words(const ChipSet& cs, const int32_t* log_ns, const uint32_t* widths, const int32_t* pairs, const int32_t* partners, int n, const zkhip_params* prm) {
    const bool lk = any_pairs(pairs, n), cross = any_cross(cs, partners, n);
    const size_t b = (size_t)prm->log_blowup, Hmax = (size_t)log_ns[0] + b, L = (size_t)log_ns[0];
    size_t words = 8 + (cross ? 4 : (lk ? 3 : 2)) * (size_t)n + 16 + (lk ? 8 : 0) + 8 * L + 4 + 1, perq = 16 * Hmax, hp = 0;
    if (cs.machine) {
        words = 8 + (cs.key ? 5 : 4) * (size_t)n + 16 + (lk ? 8 : 0) + 8 * L + 4 + 1 + (cs.key ? 8 : 0);
        for (int c = 0; c < n; c++) words += (header_has_prog(cs, c) ? 8 : 0) + (lookup_of(cs, c) ? 8 : 0);
    } else if (any_prog(cs, n)) { words += (size_t)n; for (int c = 0; c < n; c++) if (prog_of(cs, c)) words += 8; }
    size_t he = 0;
    for (int c = 0; c < n; c++) {
        const size_t wp = perm_width(pairs, c);
        words += 8 * (size_t)widths[c] + 8 * wp + 4 * qw_of(cs, c) + ((cross && wp) ? 4 : 0) + 8 * (size_t)pre_w(cs, c);
        perq += widths[c] + wp + qw_of(cs, c) + pre_w(cs, c);
        if (wp && (size_t)log_ns[c] + b > hp) hp = (size_t)log_ns[c] + b;
        if (pre_w(cs, c) && (size_t)log_ns[c] + b > he) he = (size_t)log_ns[c] + b;
    }
    perq += 8 * hp + 8 * he;
    for (size_t l = 0; l < L; l++) perq += 4 + 8 * (Hmax - 1 - l);
    return words + (size_t)prm->num_queries * perq;
}
static void chips_transcript_init(const ChipSet& cs, Challenger& ch, const int32_t* log_ns, const uint32_t* widths, const int32_t* pairs, const int32_t* partners, int n,
                                  const zkhip_params* prm, size_t n_public) {
    const bool lk = any_pairs(pairs, n), cross = any_cross(cs, partners, n);
    ch.observe_canonical(chips_version(cs, pairs, partners, n));
    ch.observe_canonical((uint32_t)n);
    ch.observe_canonical((uint32_t)prm->log_blowup);
    ch.observe_canonical((uint32_t)prm->num_queries);
    ch.observe_canonical((uint32_t)prm->pow_bits);
    ch.observe_canonical((uint32_t)n_public);
    for (int c = 0; c < n; c++) {
        ch.observe_canonical((uint32_t)log_ns[c]); ch.observe_canonical(widths[c]);
        if (cs.machine) {
            ch.observe_canonical(header_prog_word(cs, c)); ch.observe_canonical(lookup_of(cs, c) ? lookup_of(cs, c)->ni : 0u);
            if (cs.key) ch.observe_canonical(pre_w(cs, c));
            continue;
        }
        if (lk) ch.observe_canonical((uint32_t)pairs[c]);
        if (cross) ch.observe_canonical((uint32_t)(partners[c] + 1));
        if (any_prog(cs, n)) ch.observe_canonical(header_prog_word(cs, c));
    }
    for (int c = 0; c < n; c++)
        if (header_has_prog(cs, c)) {
            uint32_t dg[8];
            air_digest_cached(*prog_of(cs, c), dg);
            for (int i = 0; i < 8; i++) ch.observe_canonical(dg[i]);
        }
    for (int c = 0; c < n; c++)
        if (lookup_of(cs, c)) {
            uint32_t dg[8];
            lookup_digest(*lookup_of(cs, c), dg);
            for (int i = 0; i < 8; i++) ch.observe_canonical(dg[i]);
        }
    if (cs.key) for (int i = 0; i < 8; i++) ch.observe(cs.key->root_m[i]);
}
// alpha-power offset of chip c inside the reduced-opening vector of its height
static uint64_t height_offset(const ChipSet& cs, const int32_t* log_ns, const uint32_t* widths, const int32_t* pairs, int c) {
    uint64_t off = 0;
    for (int d = 0; d < c; d++) if (log_ns[d] == log_ns[c]) off += 2 * (uint64_t)pre_w(cs, d) + 2 * (uint64_t)widths[d] + 2 * perm_width(pairs, d) + qw_of(cs, d);
    return off;
}

// ---- the programs / tables / key of a multi-chip call, parsed once by its entry point
static int chip_programs(const uint32_t* const* programs, const size_t* program_words, const uint32_t* widths, int n, size_t n_public,
                         AirView* views, const AirView** table) {
    if (!programs || !program_words || !widths || n < 1 || n > MAX_CHIPS) return fail(ZKHIP_ERR_INVALID, "chips_air: bad arguments");
    for (int c = 0; c < n; c++) {
        table[c] = nullptr;
        if (!programs[c]) continue;
        if (!air_validate(programs[c], program_words[c], widths[c], n_public, &views[c])) return fail(ZKHIP_ERR_INVALID, "chips_air: malformed constraint program (or its n_public differs from the shard's)");
        table[c] = &views[c];
    }
    return ZKHIP_OK;
}
struct MachineSetup {
    AirView views[MAX_CHIPS];
    const AirView* table[MAX_CHIPS];
    LookupView lks[MAX_CHIPS];
    MachineTables mt;
    std::vector<uint32_t> synthetic[MAX_CHIPS];       // the built-in AIR written as a program, for chips that bring none
    int32_t cols[MAX_CHIPS];
};
static int machine_setup(const uint32_t* const* programs, const size_t* program_words, const uint32_t* const* tables, const size_t* table_words,
                         const uint32_t* widths, int n, size_t n_public, MachineSetup& m) {
    if (!programs || !program_words || !tables || !table_words || !widths || n < 1 || n > MAX_CHIPS) return fail(ZKHIP_ERR_INVALID, "machine: bad arguments");
    for (int c = 0; c < n; c++) {
        const uint32_t* prog = programs[c];
        size_t words = program_words[c];
        m.mt.has_prog[c] = prog != nullptr;
        if (!prog) {
            size_t need = 0;
            if (zkhip_air_synthetic(widths[c], n_public, nullptr, 0, &need) != ZKHIP_OK) return ZKHIP_ERR_INVALID;
            m.synthetic[c].resize(need);
            ZK_TRY(zkhip_air_synthetic(widths[c], n_public, m.synthetic[c].data(), need, &need));
            prog = m.synthetic[c].data(); words = need;
        }
        if (!air_validate(prog, words, widths[c], n_public, &m.views[c])) return fail(ZKHIP_ERR_INVALID, "machine: malformed constraint program (or its n_public differs from the shard's)");
        m.table[c] = &m.views[c];
        m.mt.lk[c] = nullptr;
        m.cols[c] = 0;
        if (tables[c]) {
            if (!lookup_validate(tables[c], table_words[c], widths[c], &m.lks[c])) return fail(ZKHIP_ERR_INVALID, "machine: malformed interaction table");
            m.mt.lk[c] = &m.lks[c];
            m.cols[c] = (int32_t)m.lks[c].cols;
        }
    }
    return ZKHIP_OK;
}
static int keyed_widths(const uint32_t* widths, const uint32_t* pre_widths, int n, uint32_t* combined) {
    if (!widths || !pre_widths || n < 1 || n > MAX_CHIPS) return fail(ZKHIP_ERR_INVALID, "keyed machine: bad arguments");
    for (int c = 0; c < n; c++) {
        if (widths[c] > 1024 || pre_widths[c] > 1024) return fail(ZKHIP_ERR_INVALID, "keyed machine: widths up to 1024");
        combined[c] = widths[c] + pre_widths[c];
    }
    return ZKHIP_OK;
}

}  // namespace zk
