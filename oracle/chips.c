/*
 * oracle/chips.c -- a shard made of several chips (AIR tables) of DIFFERENT heights, proven together, CPU restatement.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle/oracle.h).
 *
 * This is the structure of an SP1 shard (sp1-stark 4.1.4 `ShardProof`, reference Cargo.lock:6172; call site
 * crates/guest-prover-sp1/src/sp1.rs:116): every chip has its own trace height, but there is ONE commitment per phase
 * and ONE FRI proof.  Restated from the published construction of p3-fri 0.2.1-succinct TwoAdicFriPcs + p3-merkle-tree
 * FieldMerkleTreeMmcs (Cargo.lock:3930, :4013):
 *   commit      all LDEs go into one Merkle tree: the tallest matrices form the leaves, a shorter matrix is injected at
 *               the level that has as many nodes as it has rows (oracle/merkle.c);
 *   quotient    per chip, on that chip's own coset g <w_2N_c>, same constraint challenge alpha; the chunk LDEs of all
 *               chips form the second tree;
 *   open        every chip at the same zeta, and at zeta * g_c (g_c generates the chip's trace domain);
 *   FRI input   one reduced-opening vector PER HEIGHT (alpha powers run on across the chips of that height);
 *   FRI         starts from the tallest vector; after each fold, the vector of the height just reached is added in;
 *   query       one index; a chip of height 2^h is opened at index >> (Hmax - h); one Merkle path per tree.
 * Chips use the synthetic AIR of stark.c at their own width.  Shape: the SP1 FRI shape (fold by 2, constant final
 * polynomial, Poseidon2 width 16) at any log_blowup in [1, 3].
 * Lookups: a chip may carry in-table LogUp pairs (stark.c: pairs[c] > 0); the permutation traces of those chips form a
 * third mixed-height tree between the trace and the quotient commitments, as sp1-stark commits the permutation traces
 * of a shard together (proof version 5).  Lookups BETWEEN chips (partners[]): two chips of equal height hold each
 * other's sender columns; every chip with pairs then exposes the final value C of its running sum, the last-row constraint
 * becomes S = C and the verifier checks sum C = 0 -- sp1-stark's local cumulative sums (proof version 6).
 * Chips with their own AIR (orc_prove_chips_air): any chip may bring a constraint program (oracle/air.c; degree 4 / 5: four quotient chunks) instead of
 * the synthetic AIR -- a machine of different tables, as an SP1 shard is.  Proof version 9: every chip's header entry gains a
 * has-program flag and the 8-word digests of the programs follow the entries, all observed; no lookups in this version.
 */
#include "oracle.h"
#include "stark_internal.h"
#include <stdlib.h>
#include <string.h>

#define ld4 orc__ld4
#define st4 orc__st4
#define CHIPS_MAGIC 0x41544B5Au
#define CHIPS_VERSION 4u
#define CHIPS_VERSION_LOGUP 5u
#define CHIPS_VERSION_CROSS 6u     /* some chips look each other up: cumulative sums are part of the proof */
#define CHIPS_VERSION_AIR 9u       /* some chips carry a constraint program */
#define MAX_CHIPS 32

/* the programs in effect for the running orc_*_chips_air call (NULL: every chip uses the synthetic AIR) */
static _Thread_local const uint32_t* const* g_progs = NULL;
static _Thread_local const size_t* g_prog_words = NULL;
/* log2 of chip c's number of quotient chunks: 2 for a program of degree 4 or 5 (needs log_blowup >= 2), else 1; its quotient matrix has
 * 4 * 2^lq columns.  The header's has-program word carries it (0: no program, 1 or 2: the program's log_quotient_degree). */
static int lq_of(int c);
static size_t qw_of(int c) { return (size_t)4 << lq_of(c); }
static int any_prog(int n) { if (g_progs) for (int c = 0; c < n; c++) if (g_progs[c]) return 1; return 0; }
static const uint32_t* prog_of(int c) { return g_progs ? g_progs[c] : NULL; }
static int lq_of(int c) { return prog_of(c) ? orc_air_log_quotient_degree(prog_of(c)) : 1; }

/* ---- machine mode (orc_*_machine): lookups as DATA.  Every chip may bring an interaction table next to its program:
 *   [0] "LKUP" 0x50554B4C  [1] interactions I (1..64)  [2] total words
 *   I x { sign (0 send, 1 receive), multiplicity (0xFFFFFFFF: the constant 1, else a column), bus (a field element), values V (1..8), V columns }
 * Fingerprint of a tuple: d = gamma + bus + sum_t beta^(t+1) v_t.  The permutation trace has one extension column per PAIR of
 * interactions (2j, 2j+1): phi_j = s_a m_a / d_a + s_b m_b / d_b (s = +1 send, -1 receive; 1/0 = 0), then the running sum S of the row
 * sums.  Constraints, folded after the program's: phi_j d_a d_b - (s_a m_a d_b + s_b m_b d_a) per column; is_first (S - sum phi);
 * is_transition (S' - S - sum phi'); is_last (S - C), C = the chip's exposed cumulative sum; the verifier checks sum_chips C = 0.
 * This is sp1-stark's permutation argument (generate_permutation_trace / eval_permutation_constraints, batch size 2) with the
 * interactions written out as data.  Proof version 10: header entry (log_n, width, has_program, interactions), then the programs'
 * digests, then the tables' digests. ---- */
#define LKUP_MAGIC 0x50554B4Cu
#define CHIPS_VERSION_MACHINE 10u
typedef struct { uint32_t sign, mult, bus, nv; const uint32_t* cols; } inter_t;
static _Thread_local const uint32_t* const* g_tables = NULL;
static _Thread_local const size_t* g_table_words = NULL;
static _Thread_local int g_machine = 0;
static const uint32_t* table_of(int c) { return g_tables ? g_tables[c] : NULL; }
/* ---- keyed machine (orc_*_machine_keyed): PREPROCESSED columns.  sp1-stark's setup (StarkMachine::setup, called by the reference at
 * crates/guest-prover-sp1/src/sp1.rs:113) commits the chips' preprocessed traces -- program ROM, byte-operation tables -- once; the
 * root is the verifying key's commitment, and every proof opens those columns beside the main ones.  Here: chip c has pre_widths[c]
 * preprocessed columns (0: none; a chip that has some brings its own program).  Programs and interaction tables address the
 * COMBINED row [preprocessed | main].  The preprocessed LDEs of all chips form one mixed-height tree (the key); its root is observed
 * right after the header.  Proof version 11: header entry (log_n, width, has_program, interactions, pre_width), digests as in 10, then
 * the key's root; every chip's openings start with its preprocessed columns at zeta and zeta g; every query starts with the
 * preprocessed rows and their path.  In a height's reduced opening a chip contributes pre@zeta, pre@zeta g, then as before. ---- */
#define CHIPS_VERSION_KEYED 11u
static _Thread_local const size_t* g_pre_widths = NULL;
static _Thread_local const uint32_t* const* g_pre_traces = NULL;     /* prover only */
static _Thread_local const uint32_t* g_pre_root = NULL;              /* 8 canonical words */
static int keyed(void) { return g_pre_widths != NULL; }
static size_t pre_w(int c) { return g_pre_widths ? g_pre_widths[c] : 0; }
static int table_parse(const uint32_t* t, size_t words, size_t width, inter_t* out, int* n_out) {
    if (!t || words < 3 || t[0] != LKUP_MAGIC || t[1] < 1 || t[1] > 64 || t[2] != words) return 0;
    size_t p = 3;
    for (uint32_t i = 0; i < t[1]; i++) {
        if (p + 4 > words) return 0;
        inter_t it = {t[p], t[p + 1], t[p + 2], t[p + 3], t + p + 4};
        p += 4;
        if (it.sign > 1 || (it.mult != 0xFFFFFFFFu && it.mult >= width) || it.bus >= BB_P || it.nv < 1 || it.nv > 8 || p + it.nv > words) return 0;
        for (uint32_t v = 0; v < it.nv; v++) if (t[p + v] >= width) return 0;
        p += it.nv;
        if (out) out[i] = it;
    }
    if (p != words) return 0;
    if (n_out) *n_out = (int)t[1];
    return 1;
}
static bb4_t fingerprint(const inter_t* it, const bb4_t* row, bb4_t gamma, const bb4_t* bpow) {
    bb4_t d = bb4_add_base(gamma, it->bus);
    for (uint32_t v = 0; v < it->nv; v++) d = bb4_add(d, bb4_mul(bpow[v + 1], row[it->cols[v]]));
    return d;
}
static bb4_t signed_mult(const inter_t* it, const bb4_t* row) {
    bb4_t m = it->mult == 0xFFFFFFFFu ? bb4_from_base(1) : row[it->mult];
    return it->sign ? bb4_sub(bb4_zero(), m) : m;
}
/* permutation trace of a chip with an interaction table: [n][4 (cols + 1)], extension columns as 4 base words */
static void perm_trace_machine(const uint32_t* trace, int log_n, size_t width, const inter_t* its, int ni, bb4_t gamma, bb4_t beta, uint32_t* out) {
    const size_t n = (size_t)1 << log_n, cols = ((size_t)ni + 1) / 2, wp = 4 * (cols + 1);
    bb4_t bpow[10];
    bpow[0] = bb4_from_base(1);
    for (int t = 1; t < 10; t++) bpow[t] = bb4_mul(bpow[t - 1], beta);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        bb4_t* row = (bb4_t*)malloc(width * sizeof(bb4_t));
        for (size_t j = 0; j < width; j++) row[j] = bb4_from_base(trace[i * width + j]);
        bb4_t sum = bb4_zero();
        for (size_t j = 0; j < cols; j++) {
            bb4_t phi = bb4_mul(signed_mult(&its[2 * j], row), bb4_inv(fingerprint(&its[2 * j], row, gamma, bpow)));
            if (2 * j + 1 < (size_t)ni) phi = bb4_add(phi, bb4_mul(signed_mult(&its[2 * j + 1], row), bb4_inv(fingerprint(&its[2 * j + 1], row, gamma, bpow))));
            st4(out + i * wp + 4 * j, phi);
            sum = bb4_add(sum, phi);
        }
        st4(out + i * wp + 4 * cols, sum);
        free(row);
    }
    bb4_t run = bb4_zero();
    for (size_t i = 0; i < n; i++) { run = bb4_add(run, ld4(out + i * wp + 4 * cols)); st4(out + i * wp + 4 * cols, run); }
}
/* the lookup constraints folded onto acc (prover on the quotient domain and verifier at zeta alike: extension arithmetic) */
static bb4_t fold_interactions(bb4_t acc, const inter_t* its, int ni, const bb4_t* row, const bb4_t* pl, const bb4_t* pn, bb4_t gamma, bb4_t beta,
                               bb4_t sel_first, bb4_t sel_trans, bb4_t sel_last, bb4_t alpha, bb4_t cumsum) {
    const size_t cols = ((size_t)ni + 1) / 2;
    bb4_t bpow[10];
    bpow[0] = bb4_from_base(1);
    for (int t = 1; t < 10; t++) bpow[t] = bb4_mul(bpow[t - 1], beta);
    bb4_t sum_l = bb4_zero(), sum_n = bb4_zero();
    for (size_t j = 0; j < cols; j++) {
        const bb4_t da = fingerprint(&its[2 * j], row, gamma, bpow), ma = signed_mult(&its[2 * j], row);
        bb4_t c;
        if (2 * j + 1 < (size_t)ni) {
            const bb4_t db = fingerprint(&its[2 * j + 1], row, gamma, bpow), mb = signed_mult(&its[2 * j + 1], row);
            c = bb4_sub(bb4_mul(bb4_mul(pl[j], da), db), bb4_add(bb4_mul(ma, db), bb4_mul(mb, da)));
        } else c = bb4_sub(bb4_mul(pl[j], da), ma);
        acc = bb4_add(bb4_mul(acc, alpha), c);
        sum_l = bb4_add(sum_l, pl[j]);
        sum_n = bb4_add(sum_n, pn[j]);
    }
    const bb4_t S = pl[cols], Sn = pn[cols];
    acc = bb4_add(bb4_mul(acc, alpha), bb4_mul(sel_first, bb4_sub(S, sum_l)));
    acc = bb4_add(bb4_mul(acc, alpha), bb4_mul(sel_trans, bb4_sub(bb4_sub(Sn, S), sum_n)));
    acc = bb4_add(bb4_mul(acc, alpha), bb4_mul(sel_last, bb4_sub(S, cumsum)));
    return acc;
}

static bb4_t sample_ext(orc_challenger_t* ch) { bb4_t r; orc_chal_sample_ext(ch, r.c); return r; }

static int any_pairs(const int* pairs, int n) { if (pairs) for (int c = 0; c < n; c++) if (pairs[c]) return 1; return 0; }
static int any_cross(const int* partners, int n) {
    if (g_machine) { for (int c = 0; c < n; c++) if (table_of(c)) return 1; return 0; }       /* machine mode: sums are always exposed */
    if (partners) for (int c = 0; c < n; c++) if (partners[c] >= 0) return 1;
    return 0;
}
static uint32_t chips_version(const int* pairs, const int* partners, int n) {
    if (g_machine) return keyed() ? CHIPS_VERSION_KEYED : CHIPS_VERSION_MACHINE;
    if (any_prog(n)) return CHIPS_VERSION_AIR;
    return any_cross(partners, n) ? CHIPS_VERSION_CROSS : (any_pairs(pairs, n) ? CHIPS_VERSION_LOGUP : CHIPS_VERSION);
}
static int chips_ok(const int* log_ns, const size_t* widths, const int* pairs, const int* partners, int n, const orc_params_t* prm) {
    if (n < 1 || n > MAX_CHIPS) return 0;
    if (prm->log_blowup < 1 || prm->log_blowup > 3) return 0;
    if ((prm->log_fold != 0 && prm->log_fold != 1) || prm->log_final != 0 || (prm->hash_width != 0 && prm->hash_width != 16)) return 0;
    if (prm->logup_pairs != 0) return 0;
    if (!g_machine && any_prog(n) && (any_pairs(pairs, n) || any_cross(partners, n))) return 0;      /* version 9: no lookups next to programs */
    for (int c = 0; c < n; c++) {
        if (log_ns[c] < 5 || log_ns[c] > 22 || widths[c] == 0 || widths[c] % 4 != 0 || widths[c] > 1024) return 0;
        if (c && log_ns[c] > log_ns[c - 1]) return 0;             /* tallest first */
        if (pairs && (pairs[c] < 0 || pairs[c] > 64 || (!g_machine && (size_t)pairs[c] * 8 > widths[c]))) return 0;
        if (partners && partners[c] >= 0) {                        /* mutual, equal heights and pair counts */
            int d = partners[c];
            if (!pairs || d >= n || d == c || partners[d] != c || pairs[c] == 0 || pairs[d] != pairs[c] || log_ns[d] != log_ns[c]) return 0;
        } else if (partners && partners[c] < -1) return 0;
    }
    for (int c = 0; c < n; c++) if (lq_of(c) > prm->log_blowup) return 0;      /* the quotient domain must lie inside the committed LDE domain */
    if (keyed()) {
        int some = 0;
        for (int c = 0; c < n; c++) {
            const size_t pw = pre_w(c);
            if (pw % 4 != 0 || pw + widths[c] > 1024 || (pw && !prog_of(c))) return 0;
            if (pw) some = 1;
        }
        if (!some) return 0;
    }
    for (int c = 0; c < n; c++) {                                  /* at most 8 chips share a height (one leaf hash) */
        int same = 0;
        for (int d = 0; d < n; d++) if (log_ns[d] == log_ns[c]) same++;
        if (same > 8) return 0;
    }
    return 1;
}

size_t orc_chips_proof_size(const int* log_ns, const size_t* widths, const int* pairs, const int* partners, int n, const orc_params_t* prm, size_t n_public) {
    (void)n_public;
    if (!chips_ok(log_ns, widths, pairs, partners, n, prm)) return 0;
    const int lk = any_pairs(pairs, n), cross = any_cross(partners, n);
    size_t b = (size_t)prm->log_blowup, Hmax = (size_t)log_ns[0] + b, L = (size_t)log_ns[0];
    size_t words = 8 + (cross ? 4 : (lk ? 3 : 2)) * (size_t)n + 16 + (lk ? 8 : 0) + 8 * L + 4 + 1;
    if (g_machine) {
        words = 8 + (keyed() ? 5 : 4) * (size_t)n + 16 + (lk ? 8 : 0) + 8 * L + 4 + 1 + (keyed() ? 8 : 0);
        for (int c = 0; c < n; c++) words += (prog_of(c) ? 8 : 0) + (table_of(c) ? 8 : 0);
    } else if (any_prog(n)) { words += (size_t)n; for (int c = 0; c < n; c++) if (prog_of(c)) words += 8; }
    size_t perq = 16 * Hmax, hp = 0, he = 0;
    for (int c = 0; c < n; c++) {
        size_t wp = (pairs && pairs[c]) ? 4 * ((size_t)pairs[c] + 1) : 0;
        words += 8 * widths[c] + 8 * wp + 4 * qw_of(c) + ((cross && wp) ? 4 : 0) + 8 * pre_w(c);      /* + the chip's cumulative sum */
        perq += widths[c] + wp + qw_of(c) + pre_w(c);
        if (wp && (size_t)log_ns[c] + b > hp) hp = (size_t)log_ns[c] + b;
        if (pre_w(c) && (size_t)log_ns[c] + b > he) he = (size_t)log_ns[c] + b;
    }
    perq += 8 * hp + 8 * he;                           /* paths of the permutation tree and of the key's tree */
    for (size_t l = 0; l < L; l++) perq += 4 + 8 * (Hmax - 1 - l);
    return (words + (size_t)prm->num_queries * perq) * 4;
}

static void transcript_init(orc_challenger_t* ch, const int* log_ns, const size_t* widths, const int* pairs, const int* partners, int n, const orc_params_t* prm, size_t n_public) {
    const int lk = any_pairs(pairs, n), cross = any_cross(partners, n);
    orc_chal_init(ch);
    orc_chal_observe(ch, chips_version(pairs, partners, n));
    orc_chal_observe(ch, (uint32_t)n);
    orc_chal_observe(ch, (uint32_t)prm->log_blowup);
    orc_chal_observe(ch, (uint32_t)prm->num_queries);
    orc_chal_observe(ch, (uint32_t)prm->pow_bits);
    orc_chal_observe(ch, (uint32_t)n_public);
    for (int c = 0; c < n; c++) {
        orc_chal_observe(ch, (uint32_t)log_ns[c]); orc_chal_observe(ch, (uint32_t)widths[c]);
        if (g_machine) {
            orc_chal_observe(ch, prog_of(c) ? (uint32_t)lq_of(c) : 0u); orc_chal_observe(ch, table_of(c) ? table_of(c)[1] : 0u);
            if (keyed()) orc_chal_observe(ch, (uint32_t)pre_w(c));
            continue;
        }
        if (lk) orc_chal_observe(ch, (uint32_t)pairs[c]);
        if (cross) orc_chal_observe(ch, (uint32_t)(partners[c] + 1));
        if (any_prog(n)) orc_chal_observe(ch, prog_of(c) ? (uint32_t)lq_of(c) : 0u);
    }
    for (int c = 0; c < n; c++)
        if (prog_of(c)) {
            uint32_t dg[8];
            orc_air_digest(prog_of(c), g_prog_words[c], dg);
            orc_chal_observe_slice(ch, dg, 8);
        }
    for (int c = 0; c < n; c++)
        if (table_of(c)) {
            uint32_t dg[8];
            orc_air_digest(table_of(c), g_table_words[c], dg);          /* the same sponge over 16-bit halves */
            orc_chal_observe_slice(ch, dg, 8);
        }
    if (keyed()) orc_chal_observe_slice(ch, g_pre_root, 8);
}

/* alpha-power offset of chip c inside the reduced-opening vector of its height: the chips of one height share one power
 * sequence, each contributing [trace@zeta (W), trace@zeta*g (W), quotient chunks@zeta (8)] */
static size_t perm_width(const int* pairs, int c) { return (pairs && pairs[c]) ? 4 * ((size_t)pairs[c] + 1) : 0; }
static size_t height_offset(const int* log_ns, const size_t* widths, const int* pairs, int c) {
    size_t off = 0;
    for (int d = 0; d < c; d++) if (log_ns[d] == log_ns[c]) off += 2 * pre_w(d) + 2 * widths[d] + 2 * perm_width(pairs, d) + qw_of(d);
    return off;
}

/* quotient values of chip c in machine mode: its program (the synthetic AIR written as a program when it has none), then its lookups */
static void quotient_values_machine(int c, const uint32_t* lde, int log_n, size_t width, const uint32_t* plde, bb4_t gamma, bb4_t beta,
                                    bb4_t alpha, bb4_t cumsum, const uint32_t* pub, size_t n_public, uint32_t* out) {
    const uint32_t* prog = prog_of(c);
    uint32_t* synth = NULL;
    if (!prog) {
        const size_t cap = 6 + (width / 4) * 33;
        synth = (uint32_t*)malloc(cap * 4);
        orc_air_synthetic(width, n_public, synth, cap);
        prog = synth;
    }
    inter_t its[64]; int ni = 0;
    if (table_of(c)) table_parse(table_of(c), g_table_words[c], width, its, &ni);
    const size_t cols = ((size_t)ni + 1) / 2, wp = ni ? 4 * (cols + 1) : 0;
    const int lq = lq_of(c), log_m = log_n + lq;
    const size_t m = (size_t)1 << log_m, n = (size_t)1 << log_n, step = (size_t)1 << lq;
    const bb_t w = bb_two_adic_generator(log_m), wn_inv = bb_inv(bb_two_adic_generator(log_n));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; i++) {
        bb_t x = bb_mul(BB_GEN, bb_pow(w, i));
        bb_t zh = bb_sub(bb_pow(x, n), 1);
        bb_t sel_first = bb_mul(zh, bb_inv(bb_sub(x, 1)));
        bb_t sel_last = bb_mul(zh, bb_inv(bb_sub(x, wn_inv)));
        bb_t sel_trans = bb_sub(x, wn_inv);
        size_t p = bb_reverse_bits((uint32_t)i, log_m), pn = bb_reverse_bits((uint32_t)((i + step) & (m - 1)), log_m);
        bb4_t acc = orc__air_fold_base(prog, lde + p * width, lde + pn * width, pub, sel_first, sel_last, sel_trans, alpha);
        if (ni) {
            bb4_t* row = (bb4_t*)malloc(width * sizeof(bb4_t));
            for (size_t j = 0; j < width; j++) row[j] = bb4_from_base(lde[p * width + j]);
            bb4_t pl[33], pnx[33];
            for (size_t q = 0; q <= cols; q++) { pl[q] = ld4(plde + p * wp + 4 * q); pnx[q] = ld4(plde + pn * wp + 4 * q); }
            acc = fold_interactions(acc, its, ni, row, pl, pnx, gamma, beta, bb4_from_base(sel_first), bb4_from_base(sel_trans),
                                    bb4_from_base(sel_last), alpha, cumsum);
            free(row);
        }
        st4(out + 4 * p, bb4_mul_base(acc, bb_inv(zh)));
    }
    free(synth);
}

size_t orc_prove_chips(const uint32_t* const* traces, const int* log_ns, const size_t* widths, const int* pairs, const int* partners, int n,
                       const uint32_t* public_values, size_t n_public, const orc_params_t* prm,
                       uint8_t* proof_bytes, size_t cap) {
    size_t need = orc_chips_proof_size(log_ns, widths, pairs, partners, n, prm, n_public);
    if (need == 0 || cap < need) return 0;
    uint32_t* pf = (uint32_t*)proof_bytes;
    size_t pos = 0;
    const int b = prm->log_blowup, Hmax = log_ns[0] + b, L = log_ns[0], lk = any_pairs(pairs, n), cross = any_cross(partners, n);
    /* 0. keyed machine: the preprocessed LDEs and their tree (what setup computes once), combined traces [pre | main] */
    uint32_t* elde[MAX_CHIPS]; uint32_t* ctrace[MAX_CHIPS]; uint32_t* clde[MAX_CHIPS];
    uint32_t* etree = NULL; uint32_t eroot[8];
    const uint32_t* em[MAX_CHIPS]; size_t ew[MAX_CHIPS]; int elh[MAX_CHIPS]; int ne = 0, He = 0;
    for (int c = 0; c < n; c++) { elde[c] = NULL; ctrace[c] = NULL; clde[c] = NULL; }
    if (keyed()) {
        for (int c = 0; c < n; c++) {
            const size_t pw = pre_w(c), W = widths[c], nc = (size_t)1 << log_ns[c];
            if (!pw) continue;
            elde[c] = (uint32_t*)malloc(((size_t)1 << (log_ns[c] + b)) * pw * 4);
            orc_coset_lde(g_pre_traces[c], elde[c], log_ns[c], pw, b, BB_GEN);
            em[ne] = elde[c]; ew[ne] = pw; elh[ne] = log_ns[c] + b; ne++;
            if (log_ns[c] + b > He) He = log_ns[c] + b;
            ctrace[c] = (uint32_t*)malloc(nc * (pw + W) * 4);
            for (size_t i = 0; i < nc; i++) {
                memcpy(ctrace[c] + i * (pw + W), g_pre_traces[c] + i * pw, pw * 4);
                memcpy(ctrace[c] + i * (pw + W) + pw, traces[c] + i * W, W * 4);
            }
        }
        etree = (uint32_t*)malloc((2 * ((size_t)1 << He) - 1) * 32);
        orc_merkle_tree_mixed(em, ew, elh, ne, etree);
        memcpy(eroot, etree + (2 * ((size_t)1 << He) - 2) * 8, 32);
        g_pre_root = eroot;
    }
    pf[pos++] = CHIPS_MAGIC; pf[pos++] = chips_version(pairs, partners, n); pf[pos++] = (uint32_t)n; pf[pos++] = (uint32_t)b;
    pf[pos++] = (uint32_t)prm->num_queries; pf[pos++] = (uint32_t)prm->pow_bits; pf[pos++] = (uint32_t)n_public; pf[pos++] = 16u;
    for (int c = 0; c < n; c++) {
        pf[pos++] = (uint32_t)log_ns[c]; pf[pos++] = (uint32_t)widths[c];
        if (g_machine) { pf[pos++] = prog_of(c) ? (uint32_t)lq_of(c) : 0u; pf[pos++] = table_of(c) ? table_of(c)[1] : 0u; if (keyed()) pf[pos++] = (uint32_t)pre_w(c); continue; }
        if (lk) pf[pos++] = (uint32_t)pairs[c];
        if (cross) pf[pos++] = (uint32_t)(partners[c] + 1);
        if (any_prog(n)) pf[pos++] = prog_of(c) ? (uint32_t)lq_of(c) : 0u;
    }
    for (int c = 0; c < n; c++) if (prog_of(c)) { orc_air_digest(prog_of(c), g_prog_words[c], pf + pos); pos += 8; }
    for (int c = 0; c < n; c++) if (table_of(c)) { orc_air_digest(table_of(c), g_table_words[c], pf + pos); pos += 8; }
    if (keyed()) { memcpy(pf + pos, eroot, 32); pos += 8; }
    orc_challenger_t ch;
    transcript_init(&ch, log_ns, widths, pairs, partners, n, prm, n_public);
    bb4_t cumsum[MAX_CHIPS];
    for (int c = 0; c < n; c++) cumsum[c] = bb4_zero();

    /* 1. trace LDEs, one mixed-height tree */
    uint32_t* tlde[MAX_CHIPS]; uint32_t* qlde[MAX_CHIPS]; uint32_t* plde[MAX_CHIPS];
    int lh[MAX_CHIPS]; size_t w8[MAX_CHIPS], wp[MAX_CHIPS];
    for (int c = 0; c < n; c++) {
        lh[c] = log_ns[c] + b; w8[c] = qw_of(c); wp[c] = perm_width(pairs, c); plde[c] = NULL;
        tlde[c] = (uint32_t*)malloc(((size_t)1 << lh[c]) * widths[c] * 4);
        orc_coset_lde(traces[c], tlde[c], log_ns[c], widths[c], b, BB_GEN);
        if (elde[c]) {                                 /* what the chip's program and interactions read: [pre | main] rows of the LDE */
            const size_t pw = pre_w(c), W = widths[c], mc = (size_t)1 << lh[c];
            clde[c] = (uint32_t*)malloc(mc * (pw + W) * 4);
            for (size_t i = 0; i < mc; i++) {
                memcpy(clde[c] + i * (pw + W), elde[c] + i * pw, pw * 4);
                memcpy(clde[c] + i * (pw + W) + pw, tlde[c] + i * W, W * 4);
            }
        }
    }
    const size_t mmax = (size_t)1 << Hmax;
    uint32_t* ttree = (uint32_t*)malloc((2 * mmax - 1) * 32);
    orc_merkle_tree_mixed((const uint32_t* const*)tlde, widths, lh, n, ttree);
    const uint32_t* troot = ttree + (2 * mmax - 2) * 8;
    memcpy(pf + pos, troot, 32); pos += 8;
    orc_chal_observe_slice(&ch, troot, 8);
    orc_chal_observe_slice(&ch, public_values, n_public);

    /* 1b. lookups: one (gamma, beta) for the shard; the permutation traces of the chips that have pairs form a third tree */
    bb4_t gamma = bb4_zero(), beta_l = bb4_zero();
    uint32_t* ptree = NULL;
    const uint32_t* pm[MAX_CHIPS]; size_t pw[MAX_CHIPS]; int plh[MAX_CHIPS]; int np = 0, Hp = 0;
    if (lk) {
        gamma = sample_ext(&ch);
        beta_l = sample_ext(&ch);
        for (int c = 0; c < n; c++) {
            if (!wp[c]) continue;
            const size_t nc = (size_t)1 << log_ns[c], mc = (size_t)1 << lh[c];
            uint32_t* perm = (uint32_t*)malloc(nc * wp[c] * 4);
            if (g_machine) {
                inter_t its[64]; int ni = 0;
                table_parse(table_of(c), g_table_words[c], pre_w(c) + widths[c], its, &ni);
                perm_trace_machine(ctrace[c] ? ctrace[c] : traces[c], log_ns[c], pre_w(c) + widths[c], its, ni, gamma, beta_l, perm);
            } else orc_perm_trace(traces[c], log_ns[c], widths[c], pairs[c], gamma.c, beta_l.c, perm);
            if (cross) cumsum[c] = ld4(perm + (nc - 1) * wp[c] + 4 * (size_t)pairs[c]);       /* the running sum's last value */
            plde[c] = (uint32_t*)malloc(mc * wp[c] * 4);
            orc_coset_lde(perm, plde[c], log_ns[c], wp[c], b, BB_GEN);
            free(perm);
            pm[np] = plde[c]; pw[np] = wp[c]; plh[np] = lh[c]; np++;
            if (lh[c] > Hp) Hp = lh[c];
        }
        const size_t mp = (size_t)1 << Hp;
        ptree = (uint32_t*)malloc((2 * mp - 1) * 32);
        orc_merkle_tree_mixed(pm, pw, plh, np, ptree);
        const uint32_t* proot = ptree + (2 * mp - 2) * 8;
        memcpy(pf + pos, proot, 32); pos += 8;
        orc_chal_observe_slice(&ch, proot, 8);
        if (cross)
            for (int c = 0; c < n; c++) if (wp[c]) { st4(pf + pos, cumsum[c]); pos += 4; orc_chal_observe_slice(&ch, cumsum[c].c, 4); }
    }

    /* 2. quotients, per chip on its own 2N_c coset (= the first 2N_c rows of its LDE), chunk LDEs, quotient tree */
    bb4_t alpha = sample_ext(&ch);
    for (int c = 0; c < n; c++) {
        const int ln = log_ns[c], lq = lq_of(c), Hq = ln + lq;
        const size_t nc = (size_t)1 << ln, mc = (size_t)1 << lh[c], mq = (size_t)1 << Hq, NQ = (size_t)1 << lq, QW = 4 * NQ;
        uint32_t* qv = (uint32_t*)malloc(mq * 16);
        if (g_machine) quotient_values_machine(c, clde[c] ? clde[c] : tlde[c], ln, pre_w(c) + widths[c], plde[c], gamma, beta_l, alpha, cumsum[c], public_values, n_public, qv);
        else if (prog_of(c)) orc_quotient_values_air(prog_of(c), tlde[c], ln, widths[c], public_values, alpha.c, lq, qv);
        else orc_quotient_values_logup_c(tlde[c], ln, widths[c], plde[c], wp[c] ? pairs[c] : 0, gamma.c, beta_l.c, alpha.c, cumsum[c].c, qv);
        qlde[c] = (uint32_t*)malloc(mc * QW * 4);
        uint32_t* chunk = (uint32_t*)malloc(nc * 16);
        uint32_t* clde = (uint32_t*)malloc(mc * 16);
        bb_t w2n = bb_two_adic_generator(Hq);
        for (size_t k = 0; k < NQ; k++) {
            for (size_t j = 0; j < nc; j++) memcpy(chunk + 4 * j, qv + 4 * bb_reverse_bits((uint32_t)(NQ * j + k), Hq), 16);
            orc_coset_lde(chunk, clde, ln, 4, b, bb_inv(bb_pow(w2n, (uint64_t)k)));
            for (size_t r = 0; r < mc; r++) memcpy(qlde[c] + r * QW + 4 * k, clde + r * 4, 16);
        }
        free(qv); free(chunk); free(clde);
    }
    uint32_t* qtree = (uint32_t*)malloc((2 * mmax - 1) * 32);
    orc_merkle_tree_mixed((const uint32_t* const*)qlde, w8, lh, n, qtree);
    const uint32_t* qroot = qtree + (2 * mmax - 2) * 8;
    memcpy(pf + pos, qroot, 32); pos += 8;
    orc_chal_observe_slice(&ch, qroot, 8);

    /* 3. openings: one zeta for every chip, "next" point zeta * g_c; per chip: trace local | next | [perm local | next] | quotient */
    bb4_t zeta = sample_ext(&ch);
    uint32_t* op[MAX_CHIPS]; uint32_t* opre[MAX_CHIPS];      /* op[c]: the chip's main openings; opre[c]: its preprocessed ones, just before */
    size_t oplen[MAX_CHIPS];
    for (int c = 0; c < n; c++) {
        const size_t W = widths[c], pw = pre_w(c);
        oplen[c] = 8 * pw + 8 * W + 8 * wp[c] + 4 * qw_of(c);
        bb4_t zn = bb4_mul_base(zeta, bb_two_adic_generator(log_ns[c]));
        if (pw) {
            orc_open_at(elde[c], log_ns[c], pw, zeta.c, pf + pos);
            orc_open_at(elde[c], log_ns[c], pw, zn.c, pf + pos + 4 * pw);
        }
        opre[c] = pf + pos;
        op[c] = pf + pos + 8 * pw; pos += oplen[c];
        orc_open_at(tlde[c], log_ns[c], W, zeta.c, op[c]);
        orc_open_at(tlde[c], log_ns[c], W, zn.c, op[c] + 4 * W);
        if (wp[c]) {
            orc_open_at(plde[c], log_ns[c], wp[c], zeta.c, op[c] + 8 * W);
            orc_open_at(plde[c], log_ns[c], wp[c], zn.c, op[c] + 8 * W + 4 * wp[c]);
        }
        orc_open_at(qlde[c], log_ns[c], qw_of(c), zeta.c, op[c] + 8 * W + 8 * wp[c]);
    }
    for (int c = 0; c < n; c++) orc_chal_observe_slice(&ch, opre[c], oplen[c]);

    /* 4. one reduced-opening vector per height; per chip the batching order is
     * [pre@zeta, pre@zeta*g (2 Pw powers)], trace@zeta 0, trace@zeta*g W, [perm@zeta 2W, perm@zeta*g 2W+Wp], quotient@zeta 2W+2Wp */
    bb4_t fa = sample_ext(&ch);
    bb4_t* ro[32];
    for (int h = 0; h < 32; h++) ro[h] = NULL;
    for (int c = 0; c < n; c++) {
        const size_t W = widths[c], Wp = wp[c], mc = (size_t)1 << lh[c], Pw = pre_w(c);
        const size_t QW = qw_of(c);
        size_t npw = W > QW ? W : QW;
        if (Wp > npw) npw = Wp;
        if (Pw > npw) npw = Pw;
        bb4_t* fapow = (bb4_t*)malloc(npw * sizeof(bb4_t));
        fapow[0] = bb4_one();
        for (size_t j = 1; j < npw; j++) fapow[j] = bb4_mul(fapow[j - 1], fa);
        const uint32_t *o_loc = op[c], *o_nxt = op[c] + 4 * W, *o_pl = op[c] + 8 * W, *o_pn = op[c] + 8 * W + 4 * Wp, *o_q = op[c] + 8 * W + 8 * Wp;
        bb4_t y_loc = bb4_zero(), y_nxt = bb4_zero(), y_pl = bb4_zero(), y_pn = bb4_zero(), y_q = bb4_zero();
        for (size_t j = 0; j < W; j++) {
            y_loc = bb4_add(y_loc, bb4_mul(fapow[j], ld4(o_loc + 4 * j)));
            y_nxt = bb4_add(y_nxt, bb4_mul(fapow[j], ld4(o_nxt + 4 * j)));
        }
        for (size_t j = 0; j < Wp; j++) {
            y_pl = bb4_add(y_pl, bb4_mul(fapow[j], ld4(o_pl + 4 * j)));
            y_pn = bb4_add(y_pn, bb4_mul(fapow[j], ld4(o_pn + 4 * j)));
        }
        for (size_t j = 0; j < QW; j++) y_q = bb4_add(y_q, bb4_mul(fapow[j], ld4(o_q + 4 * j)));
        bb4_t y_el = bb4_zero(), y_en = bb4_zero();
        for (size_t j = 0; j < Pw; j++) {
            y_el = bb4_add(y_el, bb4_mul(fapow[j], ld4(opre[c] + 4 * j)));
            y_en = bb4_add(y_en, bb4_mul(fapow[j], ld4(opre[c] + 4 * Pw + 4 * j)));
        }
        const size_t off0 = height_offset(log_ns, widths, pairs, c), off = off0 + 2 * Pw;
        bb4_t s_el = bb4_pow(fa, off0), s_en = bb4_pow(fa, off0 + Pw);
        bb4_t s_loc = bb4_pow(fa, off), s_nxt = bb4_pow(fa, off + W), s_pl = bb4_pow(fa, off + 2 * W), s_pn = bb4_pow(fa, off + 2 * W + Wp),
              s_q = bb4_pow(fa, off + 2 * W + 2 * Wp);
        bb4_t zn = bb4_mul_base(zeta, bb_two_adic_generator(log_ns[c]));
        if (!ro[lh[c]]) ro[lh[c]] = (bb4_t*)calloc(mc, sizeof(bb4_t));
        bb4_t* dst = ro[lh[c]];
        bb_t wm = bb_two_adic_generator(lh[c]);
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < mc; p++) {
            bb_t x = bb_mul(BB_GEN, bb_pow(wm, bb_reverse_bits((uint32_t)p, lh[c])));
            bb4_t d1 = bb4_inv(bb4_neg(bb4_sub_base(zeta, x)));
            bb4_t d2 = bb4_inv(bb4_neg(bb4_sub_base(zn, x)));
            bb4_t at = orc__row_dot(fapow, tlde[c] + p * W, W), aq = orc__row_dot(fapow, qlde[c] + p * QW, QW);
            bb4_t r = bb4_mul(s_loc, bb4_mul(bb4_sub(at, y_loc), d1));
            r = bb4_add(r, bb4_mul(s_nxt, bb4_mul(bb4_sub(at, y_nxt), d2)));
            if (Wp) {
                bb4_t ap = orc__row_dot(fapow, plde[c] + p * Wp, Wp);
                r = bb4_add(r, bb4_mul(s_pl, bb4_mul(bb4_sub(ap, y_pl), d1)));
                r = bb4_add(r, bb4_mul(s_pn, bb4_mul(bb4_sub(ap, y_pn), d2)));
            }
            r = bb4_add(r, bb4_mul(s_q, bb4_mul(bb4_sub(aq, y_q), d1)));
            if (Pw) {
                bb4_t ae = orc__row_dot(fapow, elde[c] + p * Pw, Pw);
                r = bb4_add(r, bb4_mul(s_el, bb4_mul(bb4_sub(ae, y_el), d1)));
                r = bb4_add(r, bb4_mul(s_en, bb4_mul(bb4_sub(ae, y_en), d2)));
            }
            dst[p] = bb4_add(dst[p], r);
        }
        free(fapow);
    }

    /* 5. FRI commit phase with the shorter vectors joining at their height */
    bb4_t** layers = (bb4_t**)malloc(L * sizeof(bb4_t*));
    uint32_t** ltrees = (uint32_t**)malloc(L * sizeof(uint32_t*));
    uint32_t* commits = pf + pos; pos += 8 * (size_t)L;
    bb4_t* cur = ro[Hmax];
    ro[Hmax] = NULL;
    for (int l = 0; l < L; l++) {
        const int rows_log = Hmax - 1 - l;
        const size_t rows = (size_t)1 << rows_log;
        layers[l] = cur;
        ltrees[l] = (uint32_t*)malloc((2 * rows - 1) * 32);
        orc_merkle_tree_hw((const uint32_t*)cur, 8, rows_log, ltrees[l], 16);
        const uint32_t* root = ltrees[l] + (2 * rows - 2) * 8;
        memcpy(commits + 8 * l, root, 32);
        orc_chal_observe_slice(&ch, root, 8);
        bb4_t beta = sample_ext(&ch);
        bb4_t* nxt = (bb4_t*)malloc(rows * sizeof(bb4_t));
        orc_fri_fold((const uint32_t*)cur, rows_log + 1, beta.c, (uint32_t*)nxt);
        if (ro[rows_log]) {
            for (size_t i = 0; i < rows; i++) nxt[i] = bb4_add(nxt[i], ro[rows_log][i]);
            free(ro[rows_log]); ro[rows_log] = NULL;
        }
        cur = nxt;
    }
    int const_ok = 1;
    for (size_t i = 1; i < ((size_t)1 << b); i++) if (!bb4_eq(cur[0], cur[i])) const_ok = 0;
    st4(pf + pos, cur[0]); pos += 4;
    orc_chal_observe_slice(&ch, cur[0].c, 4);
    free(cur);

    /* 6. proof of work, queries */
    uint32_t witness = orc_chal_grind(&ch, prm->pow_bits);
    pf[pos++] = witness;
    for (int q = 0; q < prm->num_queries; q++) {
        size_t index = orc_chal_sample_bits(&ch, Hmax);
        if (keyed()) {
            for (int c = 0; c < n; c++) if (elde[c]) { memcpy(pf + pos, elde[c] + (index >> (Hmax - lh[c])) * pre_w(c), pre_w(c) * 4); pos += pre_w(c); }
            orc__copy_path(pf, &pos, etree, (size_t)1 << He, index >> (Hmax - He), He);
        }
        for (int c = 0; c < n; c++) { memcpy(pf + pos, tlde[c] + (index >> (Hmax - lh[c])) * widths[c], widths[c] * 4); pos += widths[c]; }
        orc__copy_path(pf, &pos, ttree, mmax, index, Hmax);
        if (lk) {
            for (int c = 0; c < n; c++) if (wp[c]) { memcpy(pf + pos, plde[c] + (index >> (Hmax - lh[c])) * wp[c], wp[c] * 4); pos += wp[c]; }
            orc__copy_path(pf, &pos, ptree, (size_t)1 << Hp, index >> (Hmax - Hp), Hp);
        }
        for (int c = 0; c < n; c++) { memcpy(pf + pos, qlde[c] + (index >> (Hmax - lh[c])) * qw_of(c), qw_of(c) * 4); pos += qw_of(c); }
        orc__copy_path(pf, &pos, qtree, mmax, index, Hmax);
        size_t idx = index;
        for (int l = 0; l < L; l++) {
            const int rows_log = Hmax - 1 - l;
            st4(pf + pos, layers[l][idx ^ 1]); pos += 4;
            orc__copy_path(pf, &pos, ltrees[l], (size_t)1 << rows_log, idx >> 1, rows_log);
            idx >>= 1;
        }
    }
    for (int l = 0; l < L; l++) { free(layers[l]); free(ltrees[l]); }
    free(layers); free(ltrees); free(ttree); free(qtree); free(ptree);
    for (int c = 0; c < n; c++) { free(tlde[c]); free(qlde[c]); free(plde[c]); free(elde[c]); free(ctrace[c]); free(clde[c]); }
    free(etree);
    if (keyed()) g_pre_root = NULL;
    for (int h = 0; h < 32; h++) free(ro[h]);
    if (!const_ok) return 0;
    return pos * 4 == need ? need : 0;
}

/* opening of a mixed-height tree: rows[c] is matrix c's row at index >> (Hmax - lh[c]); Hmax = the tree's own height */
static int verify_mixed(const uint32_t root[8], int Hmax, size_t index, const uint32_t* const* rows, const size_t* widths,
                        const int* lh, int n, const uint32_t* sibs) {
    uint32_t buf[8 * 1024 + 8], cur[8];
    size_t len = 0;
    for (int c = 0; c < n; c++) if (lh[c] == Hmax) { memcpy(buf + len, rows[c], widths[c] * 4); len += widths[c]; }
    orc_sponge_hash(buf, len, cur);
    for (int lvl = 0; lvl < Hmax; lvl++) {
        const uint32_t* sib = sibs + 8 * lvl;
        if ((index >> lvl) & 1) orc_compress(sib, cur, cur);
        else orc_compress(cur, sib, cur);
        const int h = Hmax - lvl - 1;
        len = 0;
        for (int c = 0; c < n; c++) if (lh[c] == h) { memcpy(buf + len, rows[c], widths[c] * 4); len += widths[c]; }
        if (len) {
            uint32_t rh[8];
            orc_sponge_hash(buf, len, rh);
            orc_compress(cur, rh, cur);
        }
    }
    return memcmp(cur, root, 32) == 0 ? 0 : 1;
}

int orc_verify_chips(const uint8_t* proof_bytes, size_t len, const int* log_ns, const size_t* widths, const int* pairs, const int* partners, int n,
                     const uint32_t* public_values, size_t n_public, const orc_params_t* prm) {
    if (!chips_ok(log_ns, widths, pairs, partners, n, prm)) return 1;
    if (len != orc_chips_proof_size(log_ns, widths, pairs, partners, n, prm, n_public)) return 2;
    const uint32_t* pf = (const uint32_t*)proof_bytes;
    const int b = prm->log_blowup, Hmax = log_ns[0] + b, L = log_ns[0], lk = any_pairs(pairs, n), cross = any_cross(partners, n);
    if (pf[0] != CHIPS_MAGIC || pf[1] != chips_version(pairs, partners, n) || pf[2] != (uint32_t)n || pf[3] != (uint32_t)b ||
        pf[4] != (uint32_t)prm->num_queries || pf[5] != (uint32_t)prm->pow_bits || pf[6] != (uint32_t)n_public || pf[7] != 16u) return 3;
    size_t pos = 8;
    for (int c = 0; c < n; c++) {
        if (pf[pos] != (uint32_t)log_ns[c] || pf[pos + 1] != (uint32_t)widths[c]) return 3;
        pos += 2;
        if (g_machine) {
            if (pf[pos] != (prog_of(c) ? (uint32_t)lq_of(c) : 0u) || pf[pos + 1] != (table_of(c) ? table_of(c)[1] : 0u)) return 3;
            pos += 2;
            if (keyed()) { if (pf[pos] != (uint32_t)pre_w(c)) return 3; pos++; }
            continue;
        }
        if (lk) { if (pf[pos] != (uint32_t)pairs[c]) return 3; pos++; }
        if (cross) { if (pf[pos] != (uint32_t)(partners[c] + 1)) return 3; pos++; }
        if (any_prog(n)) { if (pf[pos] != (prog_of(c) ? (uint32_t)lq_of(c) : 0u)) return 3; pos++; }
    }
    for (int c = 0; c < n; c++)
        if (prog_of(c)) {
            uint32_t dg[8];
            orc_air_digest(prog_of(c), g_prog_words[c], dg);
            if (memcmp(pf + pos, dg, 32) != 0) return 3;
            pos += 8;
        }
    for (int c = 0; c < n; c++)
        if (table_of(c)) {
            uint32_t dg[8];
            orc_air_digest(table_of(c), g_table_words[c], dg);
            if (memcmp(pf + pos, dg, 32) != 0) return 3;
            pos += 8;
        }
    if (keyed()) { if (memcmp(pf + pos, g_pre_root, 32) != 0) return 3; pos += 8; }       /* a proof under another key */
    for (size_t i = pos; i < len / 4; i++) if (pf[i] >= BB_P) return 4;
    int lh[MAX_CHIPS]; size_t w8[MAX_CHIPS], wp[MAX_CHIPS];
    for (int c = 0; c < n; c++) { lh[c] = log_ns[c] + b; w8[c] = qw_of(c); wp[c] = perm_width(pairs, c); }

    orc_challenger_t ch;
    transcript_init(&ch, log_ns, widths, pairs, partners, n, prm, n_public);
    bb4_t cumsum[MAX_CHIPS];
    for (int c = 0; c < n; c++) cumsum[c] = bb4_zero();
    const uint32_t* troot = pf + pos; pos += 8;
    orc_chal_observe_slice(&ch, troot, 8);
    orc_chal_observe_slice(&ch, public_values, n_public);
    bb4_t gamma = bb4_zero(), beta_l = bb4_zero();
    const uint32_t* proot = NULL;
    size_t pw[MAX_CHIPS]; int plh[MAX_CHIPS], pchip[MAX_CHIPS]; int np = 0, Hp = 0;
    if (lk) {
        gamma = sample_ext(&ch);
        beta_l = sample_ext(&ch);
        proot = pf + pos; pos += 8;
        orc_chal_observe_slice(&ch, proot, 8);
        for (int c = 0; c < n; c++) if (wp[c]) { pw[np] = wp[c]; plh[np] = lh[c]; pchip[np] = c; np++; if (lh[c] > Hp) Hp = lh[c]; }
        if (cross) {
            bb4_t total = bb4_zero();
            for (int c = 0; c < n; c++) if (wp[c]) { cumsum[c] = ld4(pf + pos); pos += 4; orc_chal_observe_slice(&ch, cumsum[c].c, 4); total = bb4_add(total, cumsum[c]); }
            if (!bb4_eq(total, bb4_zero())) return 11;            /* the lookups of the shard do not balance */
        }
    }
    bb4_t alpha = sample_ext(&ch);
    const uint32_t* qroot = pf + pos; pos += 8;
    orc_chal_observe_slice(&ch, qroot, 8);
    bb4_t zeta = sample_ext(&ch);
    const uint32_t* op[MAX_CHIPS]; const uint32_t* opre[MAX_CHIPS]; size_t oplen[MAX_CHIPS];
    for (int c = 0; c < n; c++) { oplen[c] = 8 * pre_w(c) + 8 * widths[c] + 8 * wp[c] + 4 * qw_of(c); opre[c] = pf + pos; op[c] = pf + pos + 8 * pre_w(c); pos += oplen[c]; }
    for (int c = 0; c < n; c++) orc_chal_observe_slice(&ch, opre[c], oplen[c]);
    size_t ew[MAX_CHIPS]; int elh[MAX_CHIPS], echip[MAX_CHIPS]; int ne = 0, He = 0;
    for (int c = 0; c < n; c++) if (pre_w(c)) { ew[ne] = pre_w(c); elh[ne] = log_ns[c] + b; echip[ne] = c; ne++; if (log_ns[c] + b > He) He = log_ns[c] + b; }

    /* (a) every chip's AIR identity at zeta */
    for (int c = 0; c < n; c++) {
        const size_t W0 = widths[c], Pw = pre_w(c), W = Pw + W0, Wp = wp[c], nc = (size_t)1 << log_ns[c];
        bb4_t* loc = (bb4_t*)malloc(W * sizeof(bb4_t));          /* the combined row [pre | main] at zeta, and at zeta g */
        bb4_t* nxt = (bb4_t*)malloc(W * sizeof(bb4_t));
        for (size_t j = 0; j < Pw; j++) { loc[j] = ld4(opre[c] + 4 * j); nxt[j] = ld4(opre[c] + 4 * Pw + 4 * j); }
        for (size_t j = 0; j < W0; j++) { loc[Pw + j] = ld4(op[c] + 4 * j); nxt[Pw + j] = ld4(op[c] + 4 * W0 + 4 * j); }
        bb_t gn = bb_two_adic_generator(log_ns[c]);
        bb4_t zn = bb4_pow(zeta, nc), zh = bb4_sub_base(zn, 1);
        bb4_t sel_first = bb4_mul(zh, bb4_inv(bb4_sub_base(zeta, 1)));
        bb4_t sel_trans = bb4_sub_base(zeta, bb_inv(gn));
        bb4_t folded = prog_of(c) ? orc__air_fold_ext(prog_of(c), loc, nxt, public_values, sel_first,
                                                        bb4_mul(zh, bb4_inv(bb4_sub_base(zeta, bb_inv(gn)))), sel_trans, alpha)
                                  : orc__fold_constraints_ext(loc, nxt, W, sel_first, sel_trans, alpha);
        if (Wp && g_machine) {
            inter_t its[64]; int ni = 0;
            table_parse(table_of(c), g_table_words[c], W, its, &ni);
            const size_t cols = ((size_t)ni + 1) / 2;
            bb4_t sel_last = bb4_mul(zh, bb4_inv(bb4_sub_base(zeta, bb_inv(gn))));
            bb4_t pl[33], pn[33];
            const uint32_t *o_pl = op[c] + 8 * W0, *o_pn = o_pl + 4 * Wp;
            for (size_t q = 0; q <= cols; q++) { pl[q] = orc__recombine(o_pl + 16 * q); pn[q] = orc__recombine(o_pn + 16 * q); }
            folded = fold_interactions(folded, its, ni, loc, pl, pn, gamma, beta_l, sel_first, sel_trans, sel_last, alpha, cumsum[c]);
        } else if (Wp) {
            const int Q = pairs[c];
            bb4_t sel_last = bb4_mul(zh, bb4_inv(bb4_sub_base(zeta, bb_inv(gn))));
            bb4_t as[64], bs[64], ar[64], br[64], pl[65], pn[65];
            const uint32_t *o_pl = op[c] + 8 * W0, *o_pn = o_pl + 4 * Wp;
            for (int q = 0; q < Q; q++) { as[q] = loc[8 * q]; bs[q] = loc[8 * q + 1]; ar[q] = loc[8 * q + 4]; br[q] = loc[8 * q + 5]; }
            for (int q = 0; q <= Q; q++) { pl[q] = orc__recombine(o_pl + 16 * q); pn[q] = orc__recombine(o_pn + 16 * q); }
            folded = orc__fold_logup(folded, Q, as, bs, ar, br, pl, pn, gamma, beta_l, sel_first, sel_trans, sel_last, alpha, cumsum[c]);
        }
        free(loc); free(nxt);
        /* quotient(zeta) = sum_k zps_k(zeta) q_k(zeta), zps_k = prod_{j != k} ((zeta / s_j)^N - 1) / ((s_k / s_j)^N - 1), s_k = g w^k on the
         * chip's own quotient domain of 2^lq cosets */
        const size_t NQ = (size_t)1 << lq_of(c);
        bb_t wq = bb_two_adic_generator(log_ns[c] + lq_of(c));
        bb_t sN[4];
        for (size_t k = 0; k < NQ; k++) sN[k] = bb_pow(bb_mul(BB_GEN, bb_pow(wq, k)), nc);
        bb4_t quot = bb4_zero();
        const uint32_t* o_q = op[c] + 8 * W0 + 8 * Wp;
        for (size_t k = 0; k < NQ; k++) {
            bb4_t zps = bb4_one();
            for (size_t j = 0; j < NQ; j++) {
                if (j == k) continue;
                bb_t sjn_inv = bb_inv(sN[j]);
                bb4_t num = bb4_sub_base(bb4_mul_base(zn, sjn_inv), 1);
                bb_t den = bb_sub(bb_mul(sN[k], sjn_inv), 1);
                zps = bb4_mul(zps, bb4_mul_base(num, bb_inv(den)));
            }
            quot = bb4_add(quot, bb4_mul(zps, orc__recombine(o_q + 16 * k)));
        }
        if (!bb4_eq(bb4_mul(folded, bb4_inv(zh)), quot)) return 10;
    }

    /* (b) FRI */
    bb4_t fa = sample_ext(&ch);
    size_t npmax = 8;
    for (int c = 0; c < n; c++) { if (widths[c] > npmax) npmax = widths[c]; if (wp[c] > npmax) npmax = wp[c]; if (pre_w(c) > npmax) npmax = pre_w(c); if (qw_of(c) > npmax) npmax = qw_of(c); }
    bb4_t* fapow = (bb4_t*)malloc(npmax * sizeof(bb4_t));
    fapow[0] = bb4_one();
    for (size_t j = 1; j < npmax; j++) fapow[j] = bb4_mul(fapow[j - 1], fa);
    bb4_t y_loc[MAX_CHIPS], y_nxt[MAX_CHIPS], y_pl[MAX_CHIPS], y_pn[MAX_CHIPS], y_q[MAX_CHIPS];
    bb4_t s_loc[MAX_CHIPS], s_nxt[MAX_CHIPS], s_pl[MAX_CHIPS], s_pn[MAX_CHIPS], s_q[MAX_CHIPS], zn_c[MAX_CHIPS];
    bb4_t y_el[MAX_CHIPS], y_en[MAX_CHIPS], s_el[MAX_CHIPS], s_en[MAX_CHIPS];
    for (int c = 0; c < n; c++) {
        const size_t W = widths[c], Wp = wp[c], Pw = pre_w(c);
        y_el[c] = y_en[c] = bb4_zero();
        for (size_t j = 0; j < Pw; j++) {
            y_el[c] = bb4_add(y_el[c], bb4_mul(fapow[j], ld4(opre[c] + 4 * j)));
            y_en[c] = bb4_add(y_en[c], bb4_mul(fapow[j], ld4(opre[c] + 4 * Pw + 4 * j)));
        }
        const uint32_t *o_pl = op[c] + 8 * W, *o_pn = o_pl + 4 * Wp, *o_q = op[c] + 8 * W + 8 * Wp;
        y_loc[c] = y_nxt[c] = y_pl[c] = y_pn[c] = y_q[c] = bb4_zero();
        for (size_t j = 0; j < W; j++) {
            y_loc[c] = bb4_add(y_loc[c], bb4_mul(fapow[j], ld4(op[c] + 4 * j)));
            y_nxt[c] = bb4_add(y_nxt[c], bb4_mul(fapow[j], ld4(op[c] + 4 * W + 4 * j)));
        }
        for (size_t j = 0; j < Wp; j++) {
            y_pl[c] = bb4_add(y_pl[c], bb4_mul(fapow[j], ld4(o_pl + 4 * j)));
            y_pn[c] = bb4_add(y_pn[c], bb4_mul(fapow[j], ld4(o_pn + 4 * j)));
        }
        for (size_t j = 0; j < qw_of(c); j++) y_q[c] = bb4_add(y_q[c], bb4_mul(fapow[j], ld4(o_q + 4 * j)));
        const size_t off0 = height_offset(log_ns, widths, pairs, c), off = off0 + 2 * Pw;
        s_el[c] = bb4_pow(fa, off0); s_en[c] = bb4_pow(fa, off0 + Pw);
        s_loc[c] = bb4_pow(fa, off); s_nxt[c] = bb4_pow(fa, off + W); s_pl[c] = bb4_pow(fa, off + 2 * W);
        s_pn[c] = bb4_pow(fa, off + 2 * W + Wp); s_q[c] = bb4_pow(fa, off + 2 * W + 2 * Wp);
        zn_c[c] = bb4_mul_base(zeta, bb_two_adic_generator(log_ns[c]));
    }
    const uint32_t* commits = pf + pos; pos += 8 * (size_t)L;
    bb4_t* betas = (bb4_t*)malloc(L * sizeof(bb4_t));
    for (int l = 0; l < L; l++) { orc_chal_observe_slice(&ch, commits + 8 * l, 8); betas[l] = sample_ext(&ch); }
    bb4_t final_poly = ld4(pf + pos); pos += 4;
    orc_chal_observe_slice(&ch, final_poly.c, 4);
    uint32_t witness = pf[pos++];
    int rc = 0;
    if (!orc_chal_check_witness(&ch, prm->pow_bits, witness)) rc = 20;
    for (int q = 0; q < prm->num_queries && rc == 0; q++) {
        size_t index = orc_chal_sample_bits(&ch, Hmax);
        const uint32_t* trow[MAX_CHIPS]; const uint32_t* qrow[MAX_CHIPS]; const uint32_t* prow[MAX_CHIPS]; const uint32_t* prow_all[MAX_CHIPS];
        const uint32_t* erow[MAX_CHIPS]; const uint32_t* erow_all[MAX_CHIPS];
        for (int c = 0; c < n; c++) erow_all[c] = NULL;
        if (keyed()) {
            for (int k = 0; k < ne; k++) { erow[k] = pf + pos; erow_all[echip[k]] = erow[k]; pos += ew[k]; }
            const uint32_t* epath = pf + pos; pos += 8 * (size_t)He;
            if (verify_mixed(g_pre_root, He, index >> (Hmax - He), erow, ew, elh, ne, epath)) { rc = 33; break; }
        }
        for (int c = 0; c < n; c++) { trow[c] = pf + pos; pos += widths[c]; prow_all[c] = NULL; }
        const uint32_t* tpath = pf + pos; pos += 8 * (size_t)Hmax;
        const uint32_t* ppath = NULL;
        if (lk) {
            for (int k = 0; k < np; k++) { prow[k] = pf + pos; prow_all[pchip[k]] = prow[k]; pos += pw[k]; }
            ppath = pf + pos; pos += 8 * (size_t)Hp;
        }
        for (int c = 0; c < n; c++) { qrow[c] = pf + pos; pos += qw_of(c); }
        const uint32_t* qpath = pf + pos; pos += 8 * (size_t)Hmax;
        if (verify_mixed(troot, Hmax, index, trow, widths, lh, n, tpath)) { rc = 30; break; }
        if (lk && verify_mixed(proot, Hp, index >> (Hmax - Hp), prow, pw, plh, np, ppath)) { rc = 32; break; }
        if (verify_mixed(qroot, Hmax, index, qrow, w8, lh, n, qpath)) { rc = 31; break; }
        /* reduced opening of every height at this query's point */
        bb4_t roh[32];
        for (int h = 0; h < 32; h++) roh[h] = bb4_zero();
        for (int c = 0; c < n; c++) {
            size_t ic = index >> (Hmax - lh[c]);
            bb_t x = bb_mul(BB_GEN, bb_pow(bb_two_adic_generator(lh[c]), bb_reverse_bits((uint32_t)ic, lh[c])));
            bb4_t d1 = bb4_inv(bb4_neg(bb4_sub_base(zeta, x)));
            bb4_t d2 = bb4_inv(bb4_neg(bb4_sub_base(zn_c[c], x)));
            bb4_t at = orc__row_dot(fapow, trow[c], widths[c]), aq = orc__row_dot(fapow, qrow[c], qw_of(c));
            bb4_t r = bb4_mul(s_loc[c], bb4_mul(bb4_sub(at, y_loc[c]), d1));
            r = bb4_add(r, bb4_mul(s_nxt[c], bb4_mul(bb4_sub(at, y_nxt[c]), d2)));
            if (wp[c]) {
                bb4_t ap = orc__row_dot(fapow, prow_all[c], wp[c]);
                r = bb4_add(r, bb4_mul(s_pl[c], bb4_mul(bb4_sub(ap, y_pl[c]), d1)));
                r = bb4_add(r, bb4_mul(s_pn[c], bb4_mul(bb4_sub(ap, y_pn[c]), d2)));
            }
            r = bb4_add(r, bb4_mul(s_q[c], bb4_mul(bb4_sub(aq, y_q[c]), d1)));
            if (pre_w(c)) {
                bb4_t ae = orc__row_dot(fapow, erow_all[c], pre_w(c));
                r = bb4_add(r, bb4_mul(s_el[c], bb4_mul(bb4_sub(ae, y_el[c]), d1)));
                r = bb4_add(r, bb4_mul(s_en[c], bb4_mul(bb4_sub(ae, y_en[c]), d2)));
            }
            roh[lh[c]] = bb4_add(roh[lh[c]], r);
        }
        bb4_t folded = roh[Hmax];
        size_t idx = index;
        for (int l = 0; l < L; l++) {
            const int rows_log = Hmax - 1 - l;
            bb4_t sib = ld4(pf + pos); pos += 4;
            const uint32_t* path = pf + pos; pos += 8 * (size_t)rows_log;
            bb4_t ev[2];
            ev[idx & 1] = folded; ev[(idx & 1) ^ 1] = sib;
            uint32_t rowbuf[8];
            memcpy(rowbuf, ev[0].c, 16); memcpy(rowbuf + 4, ev[1].c, 16);
            if (orc_merkle_verify_hw(commits + 8 * l, rows_log, idx >> 1, rowbuf, 8, path, 16)) { rc = 40 + (l < 50 ? l : 50); break; }
            folded = orc__fri_fold_row(idx >> 1, rows_log, betas[l], ev[0], ev[1]);
            idx >>= 1;
            folded = bb4_add(folded, roh[rows_log]);          /* zero unless some chip has this height */
        }
        if (rc) break;
        if (!bb4_eq(folded, final_poly)) { rc = 100; break; }
    }
    free(fapow); free(betas);
    if (rc == 0 && pos * 4 != len) rc = 5;
    return rc;
}


/* ---- chips with their own constraint programs (progs[c] NULL: the synthetic AIR); a program of degree 4 / 5 needs log_blowup >= 2 ---- */
static int progs_ok(const uint32_t* const* progs, const size_t* prog_words, const size_t* widths, int n, size_t n_public) {
    if (!progs || !prog_words || n < 1 || n > MAX_CHIPS) return 0;
    for (int c = 0; c < n; c++)
        if (progs[c] && !orc_air_validate(progs[c], prog_words[c], widths[c], n_public)) return 0;
    return 1;
}
size_t orc_chips_proof_size_air(const int* log_ns, const size_t* widths, const uint32_t* const* progs, const size_t* prog_words, int n,
                                const orc_params_t* prm, size_t n_public) {
    if (!progs_ok(progs, prog_words, widths, n, n_public)) return 0;
    g_progs = progs; g_prog_words = prog_words;
    size_t r = orc_chips_proof_size(log_ns, widths, NULL, NULL, n, prm, n_public);
    g_progs = NULL; g_prog_words = NULL;
    return r;
}
size_t orc_prove_chips_air(const uint32_t* const* traces, const int* log_ns, const size_t* widths, const uint32_t* const* progs,
                           const size_t* prog_words, int n, const uint32_t* public_values, size_t n_public, const orc_params_t* prm,
                           uint8_t* proof_bytes, size_t cap) {
    if (!progs_ok(progs, prog_words, widths, n, n_public)) return 0;
    g_progs = progs; g_prog_words = prog_words;
    size_t r = orc_prove_chips(traces, log_ns, widths, NULL, NULL, n, public_values, n_public, prm, proof_bytes, cap);
    g_progs = NULL; g_prog_words = NULL;
    return r;
}
int orc_verify_chips_air(const uint8_t* proof_bytes, size_t len, const int* log_ns, const size_t* widths, const uint32_t* const* progs,
                         const size_t* prog_words, int n, const uint32_t* public_values, size_t n_public, const orc_params_t* prm) {
    if (!progs_ok(progs, prog_words, widths, n, n_public)) return 1;
    g_progs = progs; g_prog_words = prog_words;
    int r = orc_verify_chips(proof_bytes, len, log_ns, widths, NULL, NULL, n, public_values, n_public, prm);
    g_progs = NULL; g_prog_words = NULL;
    return r;
}


/* ---- the machine: every chip with its program (or the synthetic AIR) AND its interaction table (or none); proof version 10 ---- */
static int machine_ok(const uint32_t* const* progs, const size_t* prog_words, const uint32_t* const* tables, const size_t* table_words,
                      const size_t* widths, int n, size_t n_public, int* cols) {
    if (!tables || !table_words || !progs_ok(progs, prog_words, widths, n, n_public)) return 0;
    for (int c = 0; c < n; c++) {
        cols[c] = 0;
        if (!tables[c]) continue;
        int ni = 0;
        if (!table_parse(tables[c], table_words[c], widths[c], NULL, &ni)) return 0;
        cols[c] = (ni + 1) / 2;
    }
    return 1;
}
#define MACHINE_ENTER g_progs = progs; g_prog_words = prog_words; g_tables = tables; g_table_words = table_words; g_machine = 1
#define MACHINE_LEAVE g_progs = NULL; g_prog_words = NULL; g_tables = NULL; g_table_words = NULL; g_machine = 0
size_t orc_machine_proof_size(const int* log_ns, const size_t* widths, const uint32_t* const* progs, const size_t* prog_words,
                              const uint32_t* const* tables, const size_t* table_words, int n, const orc_params_t* prm, size_t n_public) {
    int cols[MAX_CHIPS];
    if (n < 1 || n > MAX_CHIPS || !machine_ok(progs, prog_words, tables, table_words, widths, n, n_public, cols)) return 0;
    MACHINE_ENTER;
    size_t r = orc_chips_proof_size(log_ns, widths, cols, NULL, n, prm, n_public);
    MACHINE_LEAVE;
    return r;
}
size_t orc_prove_machine(const uint32_t* const* traces, const int* log_ns, const size_t* widths, const uint32_t* const* progs,
                         const size_t* prog_words, const uint32_t* const* tables, const size_t* table_words, int n,
                         const uint32_t* public_values, size_t n_public, const orc_params_t* prm, uint8_t* proof_bytes, size_t cap) {
    int cols[MAX_CHIPS];
    if (n < 1 || n > MAX_CHIPS || !machine_ok(progs, prog_words, tables, table_words, widths, n, n_public, cols)) return 0;
    MACHINE_ENTER;
    size_t r = orc_prove_chips(traces, log_ns, widths, cols, NULL, n, public_values, n_public, prm, proof_bytes, cap);
    MACHINE_LEAVE;
    return r;
}
int orc_verify_machine(const uint8_t* proof_bytes, size_t len, const int* log_ns, const size_t* widths, const uint32_t* const* progs,
                       const size_t* prog_words, const uint32_t* const* tables, const size_t* table_words, int n,
                       const uint32_t* public_values, size_t n_public, const orc_params_t* prm) {
    int cols[MAX_CHIPS];
    if (n < 1 || n > MAX_CHIPS || !machine_ok(progs, prog_words, tables, table_words, widths, n, n_public, cols)) return 1;
    MACHINE_ENTER;
    int r = orc_verify_chips(proof_bytes, len, log_ns, widths, cols, NULL, n, public_values, n_public, prm);
    MACHINE_LEAVE;
    return r;
}


/* ---- the keyed machine: preprocessed columns committed once (setup), proof version 11 ---- */
int orc_machine_setup(const uint32_t* const* pre_traces, const int* log_ns, const size_t* pre_widths, int n, const orc_params_t* prm, uint32_t root[8]) {
    if (!pre_traces || !log_ns || !pre_widths || n < 1 || n > MAX_CHIPS || prm->log_blowup < 1 || prm->log_blowup > 3) return 1;
    const int b = prm->log_blowup;
    uint32_t* elde[MAX_CHIPS]; const uint32_t* em[MAX_CHIPS]; size_t ew[MAX_CHIPS]; int elh[MAX_CHIPS]; int ne = 0, He = 0;
    for (int c = 0; c < n; c++) {
        if (log_ns[c] < 5 || log_ns[c] > 22 || (c && log_ns[c] > log_ns[c - 1]) || pre_widths[c] % 4 != 0 || pre_widths[c] > 1024) return 1;
        if (!pre_widths[c]) continue;
        if (!pre_traces[c]) return 1;
        elde[ne] = (uint32_t*)malloc(((size_t)1 << (log_ns[c] + b)) * pre_widths[c] * 4);
        orc_coset_lde(pre_traces[c], elde[ne], log_ns[c], pre_widths[c], b, BB_GEN);
        em[ne] = elde[ne]; ew[ne] = pre_widths[c]; elh[ne] = log_ns[c] + b; ne++;
        if (log_ns[c] + b > He) He = log_ns[c] + b;
    }
    if (!ne) return 1;
    uint32_t* etree = (uint32_t*)malloc((2 * ((size_t)1 << He) - 1) * 32);
    orc_merkle_tree_mixed(em, ew, elh, ne, etree);
    memcpy(root, etree + (2 * ((size_t)1 << He) - 2) * 8, 32);
    free(etree);
    for (int k = 0; k < ne; k++) free(elde[k]);
    return 0;
}
static int keyed_ok(const size_t* widths, const size_t* pre_widths, int n, size_t* combined) {
    if (!pre_widths || n < 1 || n > MAX_CHIPS) return 0;
    for (int c = 0; c < n; c++) { if (pre_widths[c] > 1024 || widths[c] > 1024) return 0; combined[c] = pre_widths[c] + widths[c]; }
    return 1;
}
#define KEYED_LEAVE MACHINE_LEAVE; g_pre_widths = NULL; g_pre_traces = NULL; g_pre_root = NULL
size_t orc_machine_proof_size_keyed(const int* log_ns, const size_t* widths, const size_t* pre_widths, const uint32_t* const* progs, const size_t* prog_words,
                                    const uint32_t* const* tables, const size_t* table_words, int n, const orc_params_t* prm, size_t n_public) {
    int cols[MAX_CHIPS]; size_t cw[MAX_CHIPS];
    if (!keyed_ok(widths, pre_widths, n, cw) || !machine_ok(progs, prog_words, tables, table_words, cw, n, n_public, cols)) return 0;
    MACHINE_ENTER; g_pre_widths = pre_widths;
    size_t r = orc_chips_proof_size(log_ns, widths, cols, NULL, n, prm, n_public);
    KEYED_LEAVE;
    return r;
}
size_t orc_prove_machine_keyed(const uint32_t* const* traces, const uint32_t* const* pre_traces, const int* log_ns, const size_t* widths, const size_t* pre_widths,
                               const uint32_t* const* progs, const size_t* prog_words, const uint32_t* const* tables, const size_t* table_words, int n,
                               const uint32_t* public_values, size_t n_public, const orc_params_t* prm, uint8_t* proof_bytes, size_t cap) {
    int cols[MAX_CHIPS]; size_t cw[MAX_CHIPS];
    if (!pre_traces || !keyed_ok(widths, pre_widths, n, cw) || !machine_ok(progs, prog_words, tables, table_words, cw, n, n_public, cols)) return 0;
    for (int c = 0; c < n; c++) if (pre_widths[c] && !pre_traces[c]) return 0;
    MACHINE_ENTER; g_pre_widths = pre_widths; g_pre_traces = pre_traces;
    size_t r = orc_prove_chips(traces, log_ns, widths, cols, NULL, n, public_values, n_public, prm, proof_bytes, cap);
    KEYED_LEAVE;
    return r;
}
int orc_verify_machine_keyed(const uint8_t* proof_bytes, size_t len, const int* log_ns, const size_t* widths, const size_t* pre_widths, const uint32_t pre_root[8],
                             const uint32_t* const* progs, const size_t* prog_words, const uint32_t* const* tables, const size_t* table_words, int n,
                             const uint32_t* public_values, size_t n_public, const orc_params_t* prm) {
    int cols[MAX_CHIPS]; size_t cw[MAX_CHIPS];
    if (!pre_root || !keyed_ok(widths, pre_widths, n, cw) || !machine_ok(progs, prog_words, tables, table_words, cw, n, n_public, cols)) return 1;
    for (int i = 0; i < 8; i++) if (pre_root[i] >= BB_P) return 1;
    MACHINE_ENTER; g_pre_widths = pre_widths; g_pre_root = pre_root;
    int r = orc_verify_chips(proof_bytes, len, log_ns, widths, cols, NULL, n, public_values, n_public, prm);
    KEYED_LEAVE;
    return r;
}
