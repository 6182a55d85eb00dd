// fri_chip.hip -- the first piece of a verifier inside a proof (second half): the FRI-fold chip.
//
// The reference's one hot call is `client.prove(&pk, &stdin, SP1ProofMode::Groth16)` (crates/guest-prover-sp1/src/sp1.rs:116):
// core -> compress -> shrink -> wrap, and compress is a machine that VERIFIES shard proofs (sp1-recursion, reference Cargo.lock:6172 ff.;
// RISC Zero: lift -> join behind prover.rs:90).  A FRI verifier spends its rows on two things: Poseidon2 (Merkle paths -- the
// Poseidon2 chip, poseidon2_chip.cpp) and extension-field folding -- this chip.  The statement proven here, about ONE shard proof of
// this library (fold by 2, constant final value):
//     "every query's chain of layer pairs folds, under the challenges beta_l, from its reduced opening to the final value, and the
//      pairs are exactly the ones listed in the key"
// as a keyed machine of two chips:
//   FRI   one row per (query, layer).  Columns: E0 E1 (the pair at positions 2k, 2k + 1 of the layer's vector, extension elements),
//         BIT (own position's parity), K (the pair's index), X = w^(bitrev K) and XI = 1/X, S = X^2, T = 1 + BIT (c_l - 1) and
//         B = suffix product of the T's (binds X of the first layer to the index bits), BETA, FOLD = (E0 + E1)/2 + BETA (E0 - E1)/(2 X),
//         ACTIVE, LN (layer number), G = ACTIVE - END with G S, G T and OWN (the row's own entry) as helper columns, L (one-hot
//         layer).  Every constraint has degree <= 3, its selector included.  The row sends (LN, K, E0) and (LN, K, E1)
//         on two buses.
//   OPENINGS   a PREPROCESSED table (setup commits it; its root is the verifying key): one row per distinct (layer, pair) with both
//         entries and the number of queries that read it.  It receives the tuples with those FIXED multiplicities, so the FRI chip
//         must send every listed pair exactly as often as the proof's queries read it: no query can be left out, none invented.
// Public values: beta_l (4 words each), then the final value.  What is NOT yet in-circuit: the Merkle paths of the pairs (the
// Poseidon2 chip proves such paths, but is not wired to this bus yet), the reduced openings themselves, the transcript.  A verifier of
// this machine proof recomputes the key from the inner proof's openings (zkhip_fri_queries_key) -- it still reads them, but no longer
// folds them.  tests/fri_air.py writes the same program, trace and table independently; the words must be equal.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "air.h"
#include "context.h"
#include "batch.h"
#include "p2chip.h"
#include "p2_x16.h"

namespace zk {
namespace frichip {

constexpr uint32_t E0 = 0, E1 = 4, BIT = 8, K = 9, X = 10, XI = 11, S = 12, T = 13, B = 14, BETA = 15, FOLD = 19, ACTIVE = 23, LN = 24, G = 25, GS = 26, GT = 27,
                   OWN = 28, L = 32;
// the wired form (zkhip_prove_fri_layers: the Poseidon2 chip authenticates every pair by its Merkle path) has two more columns in front
// of the layer selectors: K2 = 2 K (the bus key the Poseidon2 chip's index walk produces on a leaf row) and IDX = K2 + BIT (the query
// index on a query's first row)
constexpr uint32_t K2 = 32, IDX = 33, L_WIRED = 34;
// the shard verifier's form (shard_verifier.inl): one more column, XS = X (1 - 2 BIT) -- on a query's first row the query's evaluation point
// divided by the coset shift; no public final value (END rows send their folded value to the transcript table instead)
// PT = (number of the inner proof) x (its trees), constant along a chain; LNX = PT + LN names the layer's tree on the buses (the join: several
// inner proofs in one machine)
constexpr uint32_t XS = 34, PT = 35, LNX = 36, L_REC = 37, BUS_FIN = 60;
constexpr uint32_t BUS_E0 = 40, BUS_E1 = 41, BUS_R0 = 42, BUS_R1 = 43, BUS_Q = 44;
constexpr uint32_t BUS_B = 45, BUS_BF = 46;                  // transcript machine: (layer, beta) from the Poseidon2 chip's transcript rows to the ROOTS table, and from there to the fold rows
constexpr uint32_t BUS_S0 = 47, BUS_S1 = 48, BUS_I = 49;     // query-phase machine: a sponge row's sampled words (two halves) to the SAMPLES chip, (query, index) from there to QUERIES
// SAMPLES chip, one row per query-phase sponge row.  Preprocessed: C (the sponge row's number), ROW, ACT[8] (word j is a query index), POW (row 0: word 0
// is the proof-of-work sample), KQ[8] (the query's number).  Main: W[8] the words, IDX[8] their low layers + 1 bits, H1 H2 HH [8] (canonical-form
// helpers), 8 x 31 bits.
constexpr uint32_t S_PRE = 20, S_C = 0, S_ROW = 1, S_ACT = 2, S_POW = 10, S_KQ = 11;
constexpr uint32_t S_W = 0, S_IDX = 8, S_H1 = 16, S_H2 = 24, S_HH = 32, S_BITS = 40, S_MAIN = 288;
constexpr uint32_t ROOTS_MAIN_T = 8;                         // ... whose MAIN row is then (paths + 1, beta[4], queries, 0, 0); preprocessed (layer, root[8], 1, 0, 0)
constexpr uint32_t N_PUBLIC_T = 12;                          // ... and whose public values are the final value and the challenger's capacity
constexpr uint32_t QUERIES_PRE = 8, ROOTS_PRE = 12;          // QUERIES: (index, value[4], 1, 0, 0); ROOTS: (layer, root[8], 0, 0, 0) + main (count, 0, 0, 0)
constexpr uint32_t OPEN_PRE = 12, OPEN_MAIN = 4;             // OPENINGS: preprocessed (ln, k, e0[4], e1[4], m, 0), main 4 unused columns
constexpr int MIN_LAYERS = 2, MAX_LAYERS = 22;
inline uint32_t width_of(int layers, bool wired = false, bool rec = false) { return ((rec ? L_REC : wired ? L_WIRED : L) + (uint32_t)layers + 3u) & ~3u; }
inline uint32_t n_public_of(int layers, bool transcript = false) { return transcript ? N_PUBLIC_T : 4u * (uint32_t)layers + 4u; }   // betas, final value -- or final value, capacity

namespace {
inline Ext ext_from_canon(const uint32_t* p);
struct Term { uint32_t coeff; std::vector<uint32_t> vars; };
typedef std::vector<Term> Terms;
inline uint32_t var(uint32_t col, bool next = false) { return next ? ((1u << 30) | col) : col; }
inline uint32_t pub(uint32_t idx) { return (2u << 30) | idx; }
inline uint32_t neg(uint64_t c) { c %= P; return c ? (uint32_t)(P - c) : 0u; }
inline uint32_t mulm(uint64_t a, uint64_t b) { return (uint32_t)((a % P) * (b % P) % P); }
enum : uint32_t { ALL = 0, FIRST = 1, LAST = 2, TRANSITION = 3 };
struct Builder {
    std::vector<uint32_t> body;
    uint32_t count = 0;
    void add(uint32_t selector, const Terms& terms) {
        Terms kept;
        for (const Term& t : terms) if (t.coeff % P) kept.push_back(t);
        body.push_back(selector);
        body.push_back((uint32_t)kept.size());
        for (const Term& t : kept) {
            body.push_back(t.coeff % P);
            body.push_back((uint32_t)t.vars.size());
            for (uint32_t v : t.vars) body.push_back(v);
        }
        count++;
    }
};
// c_l = w_{2^(l+1)} canonical: the factor bit l of a query index contributes to its evaluation point
inline uint32_t root_const(int l) { return from_monty(two_adic_generator(l + 1)); }

// rec_public >= 0: the shard verifier's form, a program with that many public values (none of them its own)
// (inject: machine mode of the shard verifier -- machine_verifier.inl: the layers >= 1 at whose row the reduced opening of the height just reached
// joins the folded value; two more column groups behind the recursion form's: INJ [4], INJF)
std::vector<uint32_t> build_program(int RL, bool wired, bool transcript = false, int rec_public = -1, const std::vector<int>* inject = nullptr) {
    const bool rec = rec_public >= 0;
    const uint32_t L = rec ? L_REC : wired ? L_WIRED : frichip::L;       // first layer-selector column of this form
    const uint32_t W = width_of(RL, wired, rec) + (inject ? 8u : 0u), NP = rec ? (uint32_t)rec_public : n_public_of(RL, transcript), END = L + (uint32_t)RL - 1u;
    const uint32_t INJ = width_of(RL, wired, rec), INJF = INJ + 4;
    const uint32_t inv2 = (P + 1) / 2;
    Builder b;
    // G * t for the gate G = ACTIVE - END ("an active row that is not the last of its query").  A selector counts one degree, so a
    // transition constraint may multiply two columns only: G, G S, G T and the row's own entry OWN are columns of their own.
    auto gated = [&](const Terms& ts) {
        Terms out;
        for (const Term& t : ts) {
            Term a = t;
            a.vars.insert(a.vars.begin(), var(G));
            out.push_back(a);
        }
        return out;
    };
    {   // ACTIVE = sum L_l, a bit; LN = sum l L_l
        Terms t{{1u, {var(ACTIVE)}}}, n{{1u, {var(LN)}}};
        for (int l = 0; l < RL; l++) { t.push_back(Term{P - 1, {var(L + l)}}); n.push_back(Term{neg((uint64_t)l), {var(L + l)}}); }
        b.add(ALL, t);
        b.add(ALL, n);
    }
    b.add(ALL, Terms{{1u, {var(ACTIVE), var(ACTIVE)}}, {P - 1, {var(ACTIVE)}}});
    b.add(ALL, Terms{{1u, {var(BIT), var(BIT)}}, {P - 1, {var(BIT)}}});
    for (int l = 0; l < RL; l++) b.add(ALL, Terms{{1u, {var(L + l), var(L + l)}}, {P - 1, {var(L + l)}}});
    if (!transcript)                        // (transcript machine: the rows receive (layer, BETA) on a bus from the ROOTS table instead)
    for (uint32_t j = 0; j < 4; j++) {      // BETA = the layer's public challenge
        Terms t{{1u, {var(BETA + j)}}};
        for (int l = 0; l < RL; l++) t.push_back(Term{P - 1, {var(L + l), pub(4u * (uint32_t)l + j)}});
        b.add(ALL, t);
    }
    b.add(ALL, Terms{{1u, {var(G)}}, {P - 1, {var(ACTIVE)}}, {1u, {var(END)}}});
    b.add(ALL, Terms{{1u, {var(S)}}, {P - 1, {var(X), var(X)}}});
    b.add(ALL, Terms{{1u, {var(GS)}}, {P - 1, {var(G), var(S)}}});
    b.add(ALL, Terms{{1u, {var(GT)}}, {P - 1, {var(G), var(T)}}});
    for (uint32_t j = 0; j < 4; j++)        // OWN = E0 (1 - BIT) + E1 BIT
        b.add(ALL, Terms{{1u, {var(OWN + j)}}, {P - 1, {var(E0 + j)}}, {1u, {var(BIT), var(E0 + j)}}, {P - 1, {var(BIT), var(E1 + j)}}});
    b.add(ALL, Terms{{1u, {var(ACTIVE), var(X), var(XI)}}, {P - 1, {var(ACTIVE)}}});
    {   // T = 1 - BIT + BIT c_l
        Terms t{{1u, {var(T)}}, {P - 1, {}}, {1u, {var(BIT)}}};
        for (int l = 0; l < RL; l++) t.push_back(Term{neg(root_const(l)), {var(BIT), var(L + l)}});
        b.add(ALL, t);
    }
    for (uint32_t j = 0; j < 4; j++) {      // FOLD = (E0 + E1)/2 + BETA (E0 - E1) XI / 2 in F_p[x] / (x^4 - 11)
        Terms t{{1u, {var(FOLD + j)}}, {neg(inv2), {var(E0 + j)}}, {neg(inv2), {var(E1 + j)}}};
        for (uint32_t a = 0; a < 4; a++)
            for (uint32_t d = 0; d < 4; d++) {
                if ((a + d) % 4 != j) continue;
                const uint32_t w = a + d >= 4 ? mulm(inv2, EXT_W) : inv2;
                t.push_back(Term{neg(w), {var(BETA + a), var(E0 + d), var(XI)}});
                t.push_back(Term{w, {var(BETA + a), var(E1 + d), var(XI)}});
            }
        b.add(ALL, t);
    }
    for (int l = 0; l + 1 < RL; l++) b.add(TRANSITION, Terms{{1u, {var(L + l + 1, true)}}, {P - 1, {var(L + l)}}});
    b.add(TRANSITION, gated(Terms{{1u, {var(K)}}, {P - 2, {var(K, true)}}, {P - 1, {var(BIT, true)}}}));           // K = 2 K' + BIT'
    b.add(ALL, Terms{{1u, {var(END), var(K), var(K)}}, {P - 1, {var(END), var(K)}}});                              // the bit above the folded ones
    b.add(TRANSITION, Terms{{1u, {var(G), var(X, true)}}, {P - 1, {var(GS)}}, {2u, {var(GS), var(BIT, true)}}});           // X' = X^2 (1 - 2 BIT')
    b.add(TRANSITION, Terms{{1u, {var(G), var(B)}}, {P - 1, {var(GT), var(B, true)}}});                                    // B = B' T
    b.add(ALL, Terms{{1u, {var(END), var(B)}}, {P - 1, {var(END), var(T)}}, {neg((uint64_t)root_const(RL) + P - 1), {var(END), var(T), var(K)}}});
    b.add(TRANSITION, Terms{{1u, {var(L), var(X)}}, {P - 1, {var(L), var(B, true)}}});                             // first layer: X = product over the higher bits
    if (!inject)
    for (uint32_t j = 0; j < 4; j++)        // the folded value is the next layer's own entry
        b.add(TRANSITION, gated(Terms{{1u, {var(FOLD + j)}}, {P - 1, {var(OWN + j, true)}}}));
    else {                                  // ... plus what joins there
        for (uint32_t j = 0; j < 4; j++) b.add(TRANSITION, gated(Terms{{1u, {var(FOLD + j)}}, {1u, {var(INJ + j, true)}}, {P - 1, {var(OWN + j, true)}}}));
        Terms f{{1u, {var(INJF)}}};
        for (int l : *inject) f.push_back(Term{P - 1, {var(L + (uint32_t)l)}});
        b.add(ALL, f);
        for (uint32_t j = 0; j < 4; j++) b.add(ALL, Terms{{1u, {var(INJ + j)}}, {P - 1, {var(INJF), var(INJ + j)}}});
        const uint32_t IDX0 = INJF + 1;     // the query's index, constant along its chain: what joins is named by the query it joins
        b.add(ALL, Terms{{1u, {var(L), var(IDX0)}}, {P - 1, {var(L), var(IDX)}}});
        b.add(TRANSITION, gated(Terms{{1u, {var(IDX0)}}, {P - 1, {var(IDX0, true)}}}));
    }
    if (!rec) for (uint32_t j = 0; j < 4; j++) b.add(ALL, Terms{{1u, {var(END), var(FOLD + j)}}, {P - 1, {var(END), pub((transcript ? 0u : 4u * (uint32_t)RL) + j)}}});
    if (wired) {
        b.add(ALL, Terms{{1u, {var(K2)}}, {P - 2, {var(K)}}});
        b.add(ALL, Terms{{1u, {var(IDX)}}, {P - 1, {var(K2)}}, {P - 1, {var(BIT)}}});
    }
    if (rec) {
        b.add(ALL, Terms{{1u, {var(XS)}}, {P - 1, {var(X)}}, {2u, {var(X), var(BIT)}}});
        b.add(ALL, Terms{{1u, {var(LNX)}}, {P - 1, {var(PT)}}, {P - 1, {var(LN)}}});
        b.add(TRANSITION, gated(Terms{{1u, {var(PT)}}, {P - 1, {var(PT, true)}}}));
    }
    std::vector<uint32_t> p{AIR_MAGIC, 1u, W, b.count, NP, (uint32_t)(6 + b.body.size())};
    p.insert(p.end(), b.body.begin(), b.body.end());
    return p;
}
std::shared_ptr<const std::vector<uint32_t>> program(int RL, bool wired = false, bool transcript = false, int rec_public = -1) {
    static std::mutex mu;
    static std::map<int, std::shared_ptr<const std::vector<uint32_t>>> cache;
    std::lock_guard<std::mutex> lk(mu);
    const int key = 4 * RL + (wired ? 1 : 0) + (transcript ? 2 : 0) + 1024 * (rec_public + 1);
    auto it = cache.find(key);
    if (it == cache.end()) it = cache.emplace(key, std::make_shared<const std::vector<uint32_t>>(build_program(RL, wired, transcript, rec_public))).first;
    return it->second;
}
// a table whose contents are fixed by the KEY: combined row [pre | 4 main columns], one harmless first-row identity on the last main column
std::shared_ptr<const std::vector<uint32_t>> table_program(int RL, uint32_t pre_width, bool transcript = false, uint32_t main_width = 4) {
    static std::mutex mu;
    static std::map<uint64_t, std::shared_ptr<const std::vector<uint32_t>>> cache;
    std::lock_guard<std::mutex> lk(mu);
    const uint64_t key = ((uint64_t)RL << 32) | ((uint64_t)(transcript ? 1 : 0) << 31) | ((uint64_t)main_width << 16) | pre_width;
    auto it = cache.find(key);
    if (it == cache.end())
        it = cache.emplace(key, std::make_shared<const std::vector<uint32_t>>(std::vector<uint32_t>{AIR_MAGIC, 1u, pre_width + main_width, 1u, n_public_of(RL, transcript), 6u + 5u,
                                                                                                     FIRST, 1u, 1u, 1u, var(pre_width + main_width - 1u)})).first;
    return it->second;
}
// OPENINGS: combined row [ln k e0 e1 m 0 | 0 0 0 0]; its program is one harmless identity (the contents are fixed by the KEY)
std::shared_ptr<const std::vector<uint32_t>> openings_program(int RL) {
    static std::mutex mu;
    static std::map<int, std::shared_ptr<const std::vector<uint32_t>>> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(RL);
    if (it == cache.end())
        it = cache.emplace(RL, std::make_shared<const std::vector<uint32_t>>(std::vector<uint32_t>{AIR_MAGIC, 1u, OPEN_PRE + OPEN_MAIN, 1u, n_public_of(RL), 6u + 5u,
                                                                                                    FIRST, 1u, 1u, 1u, var(OPEN_PRE + 3)})).first;
    return it->second;
}
const std::vector<uint32_t>& fri_interactions() {       // two sends with multiplicity ACTIVE
    static const std::vector<uint32_t> t{LOOKUP_MAGIC, 2u, 3u + 2u * 10u,
                                         0u, ACTIVE, BUS_E0, 6u, LN, K, E0, E0 + 1, E0 + 2, E0 + 3,
                                         0u, ACTIVE, BUS_E1, 6u, LN, K, E1, E1 + 1, E1 + 2, E1 + 3};
    return t;
}
const std::vector<uint32_t>& openings_interactions() {  // two receives with the preprocessed multiplicity (column 10)
    static const std::vector<uint32_t> t{LOOKUP_MAGIC, 2u, 3u + 2u * 10u,
                                         1u, 10u, BUS_E0, 6u, 0u, 1u, 2u, 3u, 4u, 5u,
                                         1u, 10u, BUS_E1, 6u, 0u, 1u, 6u, 7u, 8u, 9u};
    return t;
}

std::vector<uint32_t> fri_interactions_wired(int RL, bool transcript = false) {     // pairs to the Poseidon2 chip (key = 2 K), the query's start to the QUERIES table
    (void)RL;
    if (transcript)                                         // ... and every active row receives its layer's challenge from the ROOTS table
        return std::vector<uint32_t>{LOOKUP_MAGIC, 4u, 3u + 2u * 10u + 2u * 9u,
                                     0u, ACTIVE, BUS_E0, 6u, LN, K2, E0, E0 + 1, E0 + 2, E0 + 3,
                                     0u, ACTIVE, BUS_E1, 6u, LN, K2, E1, E1 + 1, E1 + 2, E1 + 3,
                                     0u, L_WIRED, BUS_Q, 5u, IDX, OWN, OWN + 1, OWN + 2, OWN + 3,
                                     1u, ACTIVE, BUS_BF, 5u, LN, BETA, BETA + 1, BETA + 2, BETA + 3};
    return std::vector<uint32_t>{LOOKUP_MAGIC, 3u, 3u + 2u * 10u + 9u,
                                 0u, ACTIVE, BUS_E0, 6u, LN, K2, E0, E0 + 1, E0 + 2, E0 + 3,
                                 0u, ACTIVE, BUS_E1, 6u, LN, K2, E1, E1 + 1, E1 + 2, E1 + 3,
                                 0u, L_WIRED, BUS_Q, 5u, IDX, OWN, OWN + 1, OWN + 2, OWN + 3};
}
const std::vector<uint32_t>& p2_interactions() {          // leaf rows receive the pairs; END rows send (layer, digest) in two halves
    using namespace p2chip;
    static const std::vector<uint32_t> t{LOOKUP_MAGIC, 4u, 3u + 2u * 10u + 2u * 9u,
                                         1u, M, BUS_E0, 6u, LNP, KP, IN, IN + 1, IN + 2, IN + 3,
                                         1u, M, BUS_E1, 6u, LNP, KP, IN + 4, IN + 5, IN + 6, IN + 7,
                                         0u, END, BUS_R0, 5u, LNP, oute(7), oute(7) + 1, oute(7) + 2, oute(7) + 3,
                                         0u, END, BUS_R1, 5u, LNP, oute(7) + 4, oute(7) + 5, oute(7) + 6, oute(7) + 7};
    return t;
}
const std::vector<uint32_t>& p2_interactions_transcript() {   // ... and the transcript rows send the root they absorb and the challenge they produce
    using namespace p2chip;
    static const std::vector<uint32_t> t{LOOKUP_MAGIC, 7u, 3u + 2u * 10u + 5u * 9u,
                                         1u, M, BUS_E0, 6u, LNP, KP, IN, IN + 1, IN + 2, IN + 3,
                                         1u, M, BUS_E1, 6u, LNP, KP, IN + 4, IN + 5, IN + 6, IN + 7,
                                         0u, END, BUS_R0, 5u, LNP, oute(7), oute(7) + 1, oute(7) + 2, oute(7) + 3,
                                         0u, END, BUS_R1, 5u, LNP, oute(7) + 4, oute(7) + 5, oute(7) + 6, oute(7) + 7,
                                         0u, TRS, BUS_R0, 5u, LNP, IN, IN + 1, IN + 2, IN + 3,
                                         0u, TRS, BUS_R1, 5u, LNP, IN + 4, IN + 5, IN + 6, IN + 7,
                                         0u, TRS, BUS_B, 5u, LNP, oute(7) + 7, oute(7) + 6, oute(7) + 5, oute(7) + 4};
    return t;
}
const std::vector<uint32_t>& queries_interactions() {     // receive (preprocessed multiplicity column 5, [index, value])
    static const std::vector<uint32_t> t{LOOKUP_MAGIC, 1u, 3u + 9u, 1u, 5u, BUS_Q, 5u, 0u, 1u, 2u, 3u, 4u};
    return t;
}
const std::vector<uint32_t>& roots_interactions() {       // receive (main count column = combined column 12, [layer, half of the root])
    static const std::vector<uint32_t> t{LOOKUP_MAGIC, 2u, 3u + 2u * 9u,
                                         1u, ROOTS_PRE, BUS_R0, 5u, 0u, 1u, 2u, 3u, 4u,
                                         1u, ROOTS_PRE, BUS_R1, 5u, 0u, 5u, 6u, 7u, 8u};
    return t;
}

const std::vector<uint32_t>& roots_interactions_transcript() {   // ... the challenge (MAIN columns 13 .. 16: the prover's) received once from the transcript
    static const std::vector<uint32_t> t{LOOKUP_MAGIC, 4u, 3u + 4u * 9u,                                // row (preprocessed multiplicity 1, column 9) and sent
                                         1u, ROOTS_PRE, BUS_R0, 5u, 0u, 1u, 2u, 3u, 4u,                 // to the fold rows (main multiplicity column 17 = queries)
                                         1u, ROOTS_PRE, BUS_R1, 5u, 0u, 5u, 6u, 7u, 8u,
                                         1u, 9u, BUS_B, 5u, 0u, ROOTS_PRE + 1, ROOTS_PRE + 2, ROOTS_PRE + 3, ROOTS_PRE + 4,
                                         0u, ROOTS_PRE + 5, BUS_BF, 5u, 0u, ROOTS_PRE + 1, ROOTS_PRE + 2, ROOTS_PRE + 3, ROOTS_PRE + 4};
    return t;
}

// ---- the query-phase machine's tables
const std::vector<uint32_t>& p2_interactions_indices() {      // ... and the query-phase rows send the words they hand out, out[7] first
    using namespace p2chip;
    static const std::vector<uint32_t> t = [] {
        std::vector<uint32_t> v = p2_interactions_transcript();
        const uint32_t o = oute(7);
        const uint32_t more[18] = {0u, QP, BUS_S0, 5u, LNP, o + 7, o + 6, o + 5, o + 4, 0u, QP, BUS_S1, 5u, LNP, o + 3, o + 2, o + 1, o};
        v.insert(v.end(), more, more + 18);
        v[1] += 2u; v[2] = (uint32_t)v.size();
        return v;
    }();
    return t;
}
const std::vector<uint32_t>& queries_interactions_indices() {  // the start of query k: (main index, preprocessed value); and (preprocessed k, main index) from the SAMPLES chip
    static const std::vector<uint32_t> t{LOOKUP_MAGIC, 2u, 3u + 9u + 6u, 1u, 5u, BUS_Q, 5u, QUERIES_PRE, 1u, 2u, 3u, 4u, 1u, 5u, BUS_I, 2u, 0u, QUERIES_PRE};
    return t;
}
const std::vector<uint32_t>& samples_interactions() {
    static const std::vector<uint32_t> t = [] {
        std::vector<uint32_t> v{LOOKUP_MAGIC, 10u, 0u,
                                1u, S_ROW, BUS_S0, 5u, S_C, S_PRE + S_W, S_PRE + S_W + 1, S_PRE + S_W + 2, S_PRE + S_W + 3,
                                1u, S_ROW, BUS_S1, 5u, S_C, S_PRE + S_W + 4, S_PRE + S_W + 5, S_PRE + S_W + 6, S_PRE + S_W + 7};
        for (uint32_t j = 0; j < 8; j++) { const uint32_t e[6] = {0u, S_ACT + j, BUS_I, 2u, S_KQ + j, S_PRE + S_IDX + j}; v.insert(v.end(), e, e + 6); }
        v[2] = (uint32_t)v.size();
        return v;
    }();
    return t;
}
inline size_t sample_rows(size_t nq) { return (1 + nq + 7) / 8; }
// every word = sum of its 31 bits, in canonical form (P = 2^31 - 2^27 + 1: bits 27..30 all set -> bits 0..26 clear); IDX = the low layers + 1 bits;
// the proof-of-work word's low pow_bits bits are zero
std::shared_ptr<const std::vector<uint32_t>> samples_program(int RL, int pow_bits, uint32_t n_public = N_PUBLIC_T) {
    static std::mutex mu;
    static std::map<uint64_t, std::shared_ptr<const std::vector<uint32_t>>> cache;
    std::lock_guard<std::mutex> lk(mu);
    const uint64_t key = ((uint64_t)n_public << 32) | ((uint32_t)RL << 8) | (uint32_t)pow_bits;
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    Builder b;
    const uint32_t M0 = S_PRE;
    for (uint32_t j = 0; j < 8; j++) {
        auto bit = [&](uint32_t i) { return var(M0 + S_BITS + 31u * j + i); };
        for (uint32_t i = 0; i < 31; i++) b.add(ALL, Terms{{1u, {bit(i), bit(i)}}, {P - 1, {bit(i)}}});
        Terms w{{1u, {var(M0 + S_W + j)}}};
        for (uint32_t i = 0; i < 31; i++) w.push_back(Term{neg(1ull << i), {bit(i)}});
        b.add(ALL, w);
        b.add(ALL, Terms{{1u, {var(M0 + S_H1 + j)}}, {P - 1, {bit(30), bit(29)}}});
        b.add(ALL, Terms{{1u, {var(M0 + S_H2 + j)}}, {P - 1, {bit(28), bit(27)}}});
        b.add(ALL, Terms{{1u, {var(M0 + S_HH + j)}}, {P - 1, {var(M0 + S_H1 + j), var(M0 + S_H2 + j)}}});
        Terms c;
        for (uint32_t i = 0; i < 27; i++) c.push_back(Term{1u, {var(M0 + S_HH + j), bit(i)}});
        b.add(ALL, c);
        Terms x{{1u, {var(M0 + S_IDX + j)}}};
        for (uint32_t i = 0; i < (uint32_t)RL + 1u; i++) x.push_back(Term{neg(1ull << i), {bit(i)}});
        b.add(ALL, x);
    }
    if (pow_bits) {
        Terms c;
        for (uint32_t i = 0; i < (uint32_t)pow_bits; i++) c.push_back(Term{1u, {var(S_POW), var(M0 + S_BITS + i)}});
        b.add(ALL, c);
    }
    std::vector<uint32_t> p{AIR_MAGIC, 1u, S_PRE + S_MAIN, b.count, n_public, (uint32_t)(6 + b.body.size())};
    p.insert(p.end(), b.body.begin(), b.body.end());
    return cache.emplace(key, std::make_shared<const std::vector<uint32_t>>(std::move(p))).first->second;
}
// the chip's preprocessed rows (Montgomery): fixed by the shape alone
void samples_pre(int RL, size_t nq, int log_rows, std::vector<uint32_t>& t, int base = -1) {      // base: the number of the first query-phase sponge row (default: RL)
    t.assign(((size_t)S_PRE) << log_rows, 0u);
    for (size_t r = 0; r < sample_rows(nq); r++) {
        uint32_t* row = t.data() + S_PRE * r;
        row[S_C] = to_monty((uint32_t)(base < 0 ? RL : base) + (uint32_t)r); row[S_ROW] = MONTY_R1;
        for (uint32_t j = 0; j < 8; j++) {
            const size_t slot = 8 * r + j;
            if (slot == 0) row[S_POW] = MONTY_R1;
            else if (slot <= nq) { row[S_ACT + j] = MONTY_R1; row[S_KQ + j] = to_monty((uint32_t)(slot - 1)); }
        }
    }
}
// ... and its main rows from the sampled words (canonical); the low bits of the query words -> drawn
void samples_main(int RL, size_t nq, int log_rows, const uint32_t* words, std::vector<uint32_t>& t, std::vector<uint32_t>& drawn) {
    t.assign(((size_t)S_MAIN) << log_rows, 0u);
    drawn.clear();
    const uint32_t mask = (1u << (RL + 1)) - 1u;
    for (size_t r = 0; r < sample_rows(nq); r++) {
        uint32_t* row = t.data() + S_MAIN * r;
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t w = words[8 * r + j];
            row[S_W + j] = to_monty(w); row[S_IDX + j] = to_monty(w & mask);
            for (uint32_t i = 0; i < 31; i++) row[S_BITS + 31 * j + i] = (w >> i) & 1u ? MONTY_R1 : 0u;
            const uint32_t h1 = (w >> 30) & (w >> 29) & 1u, h2 = (w >> 28) & (w >> 27) & 1u;
            row[S_H1 + j] = h1 ? MONTY_R1 : 0u; row[S_H2 + j] = h2 ? MONTY_R1 : 0u; row[S_HH + j] = h1 & h2 ? MONTY_R1 : 0u;
            const size_t slot = 8 * r + j;
            if (slot >= 1 && slot <= nq) drawn.push_back(w & mask);
        }
    }
}

struct TraceArgs {
    const uint32_t *betas, *indices, *values, *siblings;    // canonical, on the device
    uint32_t n_queries, layers, width, log_h, wired;        // log_h = layers + 1: bits of a query index
    uint64_t rows;
    uint32_t* trace; uint64_t ld;                           // Montgomery
    uint32_t* finals;                                       // [n_queries][4] canonical: the value every chain ends in
    uint32_t pt;                                            // rec form: the PT column's value (canonical)
    uint32_t per_proof, pt_stride;                          // several inner proofs in one launch (per_proof > 0): query q belongs to proof q / per_proof -- its betas are
                                                            // betas + 4 layers proof, its PT is pt + pt_stride proof
    uint64_t row_base, pad_from;                            // the chains' rows start at row_base; rows pad_from .. rows - 1 are padding (pad_from = rows: none)
};
// one thread per query walks its layers (the folded value of a layer is the next layer's own entry); threads past the queries fill
// the padding rows: zeros with T = 1
__device__ __forceinline__ void fri_trace_kernel_body(const TraceArgs& a) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t RL = a.layers, W = a.width, L = a.wired == 2u ? L_REC : a.wired ? L_WIRED : frichip::L;
    if (q >= a.n_queries) {
        for (uint64_t r = a.pad_from + (q - a.n_queries); r < a.rows; r += (uint64_t)gridDim.x * blockDim.x - a.n_queries) {
            uint32_t* row = a.trace + r * a.ld;
            for (uint32_t c = 0; c < W; c++) row[c] = c == T ? MONTY_R1 : 0u;
        }
        return;
    }
    uint32_t idx = a.indices[q];
    const uint32_t proof = a.per_proof ? q / a.per_proof : 0u, pt = a.pt + a.pt_stride * proof;
    const uint32_t* betas = a.betas + 4u * RL * proof;
    Ext own;
    for (int i = 0; i < 4; i++) own.c[i] = to_monty(a.values[4 * q + i]);
    uint32_t tcol[MAX_LAYERS];
    for (uint32_t l = 0; l < RL; l++) {
        uint32_t* row = a.trace + (a.row_base + (uint64_t)q * RL + l) * a.ld;
        const uint32_t bit = idx & 1u, k = idx >> 1;
        Ext sib, beta;
        for (int i = 0; i < 4; i++) { sib.c[i] = to_monty(a.siblings[4 * ((uint64_t)q * RL + l) + i]); beta.c[i] = to_monty(betas[4 * l + i]); }
        const Ext e0 = bit ? sib : own, e1 = bit ? own : sib;
        const int lh = (int)a.log_h - (int)(l + 1);                       // the layer's vector has 2^(lh + 1) entries
        const uint32_t x = fpow(two_adic_generator(lh + 1), reverse_bits(k, lh));
        const uint32_t xi = finv(x);
        const Ext even = ext_mul_base(ext_add(e0, e1), MONTY_INV2);
        const Ext odd = ext_mul_base(ext_sub(e0, e1), fmul(MONTY_INV2, xi));
        const Ext fold = ext_add(even, ext_mul(beta, odd));
        for (uint32_t c = 0; c < W; c++) row[c] = 0u;
        for (int i = 0; i < 4; i++) { row[E0 + i] = e0.c[i]; row[E1 + i] = e1.c[i]; row[BETA + i] = beta.c[i]; row[FOLD + i] = fold.c[i]; }
        row[BIT] = bit ? MONTY_R1 : 0u;
        row[K] = to_monty(k);
        row[X] = x; row[XI] = xi; row[S] = fmul(x, x);
        tcol[l] = bit ? two_adic_generator((int)l + 1) : MONTY_R1;
        row[T] = tcol[l];
        row[ACTIVE] = MONTY_R1;
        row[LN] = to_monty(l);
        row[L + l] = MONTY_R1;
        if (l + 1 < RL) { row[G] = MONTY_R1; row[GS] = row[S]; row[GT] = tcol[l]; }
        for (int i = 0; i < 4; i++) row[OWN + i] = own.c[i];
        if (a.wired) { row[K2] = to_monty(2u * k); row[IDX] = to_monty(2u * k + bit); }
        if (a.wired == 2u) { row[XS] = bit ? fsub(0u, x) : x; row[PT] = to_monty(pt); row[LNX] = to_monty(pt + l); }
        own = fold;
        idx = k;
    }
    // B: suffix products of T, the last row's including the factor of the one index bit above the folded ones (idx is that bit now)
    uint32_t bacc = idx ? two_adic_generator((int)RL + 1) : MONTY_R1;
    for (int l = (int)RL - 1; l >= 0; l--) {
        bacc = fmul(bacc, tcol[l]);
        a.trace[(a.row_base + (uint64_t)q * RL + (uint32_t)l) * a.ld + B] = bacc;
    }
    for (int i = 0; i < 4; i++) a.finals[4 * q + i] = from_monty(own.c[i]);
}
__global__ void __launch_bounds__(64) fri_trace_kernel(TraceArgs a) { fri_trace_kernel_body(a); }
struct fri_trace_kernel_bargs { TraceArgs a; static fri_trace_kernel_bargs make(TraceArgs a) { return fri_trace_kernel_bargs{a}; } };
__global__ void __launch_bounds__(64) fri_trace_kernel_batch(const fri_trace_kernel_bargs* __restrict__ zk_arr) { const fri_trace_kernel_bargs& zk_b = zk_arr[blockIdx.z]; fri_trace_kernel_body(zk_b.a); }


int shape_ok(int layers, size_t n_queries, int* log_rows) {
    if (layers < MIN_LAYERS || layers > MAX_LAYERS || n_queries < 1 || n_queries > ((size_t)1 << 16))
        return fail(ZKHIP_ERR_INVALID, "fri queries: 2..22 layers, 1..65536 queries");
    int lr = 5;
    while (((size_t)1 << lr) < n_queries * (size_t)layers) lr++;
    *log_rows = lr;
    return ZKHIP_OK;
}
struct Machine {
    int32_t log_ns[2]; uint32_t widths[2], pre_widths[2];
    const uint32_t* progs[2]; size_t prog_words[2]; const uint32_t* tabs[2]; size_t tab_words[2];
    std::shared_ptr<const std::vector<uint32_t>> p0, p1;
};
Machine machine_of(int layers, int log_rows) {
    Machine m;
    m.p0 = program(layers); m.p1 = openings_program(layers);
    m.log_ns[0] = m.log_ns[1] = log_rows;
    m.widths[0] = width_of(layers); m.widths[1] = OPEN_MAIN;
    m.pre_widths[0] = 0; m.pre_widths[1] = OPEN_PRE;
    m.progs[0] = m.p0->data(); m.prog_words[0] = m.p0->size(); m.progs[1] = m.p1->data(); m.prog_words[1] = m.p1->size();
    m.tabs[0] = fri_interactions().data(); m.tab_words[0] = fri_interactions().size();
    m.tabs[1] = openings_interactions().data(); m.tab_words[1] = openings_interactions().size();
    return m;
}
// the OPENINGS table of a view: one row per distinct (layer, pair) in ascending (layer, pair) order, the pair's two entries, and how
// many queries read it; the remaining rows are zero (multiplicity 0).  Host, canonical -> Montgomery.  Also checks that the chains are
// consistent (the same pair read by two queries holds the same entries): a view taken from an accepted proof always is.
int build_openings(int layers, size_t n_queries, const uint32_t* betas, const uint32_t* indices, const uint32_t* values, const uint32_t* siblings,
                   int log_rows, std::vector<uint32_t>& table, uint32_t final_value[4]) {
    const int H = layers + 1;
    std::map<uint64_t, std::vector<uint32_t>> rows;          // key = layer << 32 | pair -> e0[4] e1[4] count
    bool have_final = false;
    for (size_t q = 0; q < n_queries; q++) {
        uint32_t idx = indices[q];
        if (idx >> H) return fail(ZKHIP_ERR_INVALID, "fri queries: a query index has more than layers + 1 bits");
        Ext own = ext_from_canon(values + 4 * q);
        for (int l = 0; l < layers; l++) {
            const uint32_t bit = idx & 1u, k = idx >> 1;
            const Ext sib = ext_from_canon(siblings + 4 * (q * (size_t)layers + l));
            const Ext e0 = bit ? sib : own, e1 = bit ? own : sib;
            std::vector<uint32_t> r(9);
            for (int i = 0; i < 4; i++) { r[i] = e0.c[i]; r[4 + i] = e1.c[i]; }
            auto it = rows.find(((uint64_t)l << 32) | k);
            if (it == rows.end()) { r[8] = 1; rows.emplace(((uint64_t)l << 32) | k, r); }
            else {
                for (int i = 0; i < 8; i++) if (it->second[i] != r[i]) return fail(ZKHIP_ERR_INVALID, "fri queries: two queries disagree about a layer pair");
                it->second[8]++;
            }
            const int lh = H - (l + 1);
            const uint32_t xi = finv(fpow(two_adic_generator(lh + 1), reverse_bits(k, lh)));
            const Ext beta = ext_from_canon(betas + 4 * l);
            own = ext_add(ext_mul_base(ext_add(e0, e1), MONTY_INV2), ext_mul(beta, ext_mul_base(ext_sub(e0, e1), fmul(MONTY_INV2, xi))));
            idx = k;
        }
        uint32_t fv[4];
        for (int i = 0; i < 4; i++) fv[i] = from_monty(own.c[i]);
        if (!have_final) { std::memcpy(final_value, fv, 16); have_final = true; }
        else if (std::memcmp(final_value, fv, 16) != 0) return fail(ZKHIP_ERR_INVALID, "fri queries: the chains do not end in one value");
    }
    if (rows.size() > ((size_t)1 << log_rows)) return fail(ZKHIP_ERR_INTERNAL, "fri queries: more pairs than rows");
    table.assign(((size_t)OPEN_PRE) << log_rows, 0u);
    size_t r = 0;
    for (const auto& e : rows) {
        uint32_t* row = table.data() + OPEN_PRE * r++;
        row[0] = to_monty((uint32_t)(e.first >> 32));
        row[1] = to_monty((uint32_t)e.first);
        for (int i = 0; i < 8; i++) row[2 + i] = e.second[i];
        row[10] = to_monty(e.second[8]);
    }
    return ZKHIP_OK;
}
inline Ext ext_from_canon(const uint32_t* p) { return Ext{{to_monty(p[0]), to_monty(p[1]), to_monty(p[2]), to_monty(p[3])}}; }
bool canonical(const uint32_t* v, size_t n) { for (size_t i = 0; i < n; i++) if (v[i] >= P) return false; return true; }

}  // namespace
}  // namespace frichip
}  // namespace zk

using namespace zk;

#define CHECK_CTX(ctx)                                                  \
    do {                                                                \
        if (!(ctx)) return fail(ZKHIP_ERR_INVALID, "null context");     \
        ZK_HIP(hipSetDevice((ctx)->device));                            \
    } while (0)

extern "C" {

uint32_t zkhip_fri_chip_width(int layers) { return layers >= frichip::MIN_LAYERS && layers <= frichip::MAX_LAYERS ? frichip::width_of(layers) : 0u; }

size_t zkhip_fri_chip_air(int layers, uint32_t* program, size_t cap_words) {
    if (layers < frichip::MIN_LAYERS || layers > frichip::MAX_LAYERS) return 0;
    const auto p = frichip::program(layers);
    if (program && cap_words >= p->size()) std::memcpy(program, p->data(), p->size() * 4);
    return p->size();
}

}  // extern "C"
static int fri_gen_trace(zkhip_ctx* ctx, int layers, size_t n_queries, const uint32_t* betas, const uint32_t* indices, const uint32_t* values,
                         const uint32_t* siblings, int log_rows, uint32_t* d_trace, size_t ld, uint32_t* finals, bool wired, bool rec = false, uint32_t pt = 0,
                         uint64_t row_base = 0, int64_t pad_from = -1, size_t n_proofs = 1, uint32_t pt_stride = 0) {
    // n_proofs > 1 (the shard verifier's join): n_queries counts ALL proofs' queries (n_queries / n_proofs each), betas holds every proof's challenges
    CHECK_CTX(ctx);
    int need;
    if (n_proofs < 1 || n_queries % n_proofs) return fail(ZKHIP_ERR_INVALID, "fri_chip_gen_trace: bad arguments");
    ZK_TRY(frichip::shape_ok(layers, n_queries / n_proofs, &need));
    if (n_proofs > 1) { need = 5; while (((size_t)1 << need) < n_queries * (size_t)layers) need++; }
    const uint32_t W = frichip::width_of(layers, wired, rec);
    if (!betas || !indices || !values || !siblings || !d_trace || !finals || ld < W || log_rows < need || log_rows > MAX_LOG_ROWS)
        return fail(ZKHIP_ERR_INVALID, "fri_chip_gen_trace: bad arguments");
    const size_t nb = 4 * (size_t)layers * n_proofs, nv = 4 * n_queries, ns = nv * (size_t)layers;
    if (!frichip::canonical(betas, nb) || !frichip::canonical(values, nv) || !frichip::canonical(siblings, ns)) return fail(ZKHIP_ERR_INVALID, "fri_chip_gen_trace: values must be canonical");
    for (size_t q = 0; q < n_queries; q++) if (indices[q] >> (layers + 1)) return fail(ZKHIP_ERR_INVALID, "fri_chip_gen_trace: a query index has more than layers + 1 bits");
    void* stage;
    ZK_TRY(ctx_reserve(ctx, S_STAGE, (nb + n_queries + nv + ns + nv) * 4, &stage));
    uint32_t* d = (uint32_t*)stage;
    {   // one upload: betas | indices | values | siblings
        std::vector<uint32_t> up(nb + n_queries + nv + ns);
        std::memcpy(up.data(), betas, nb * 4);
        std::memcpy(up.data() + nb, indices, n_queries * 4);
        std::memcpy(up.data() + nb + n_queries, values, nv * 4);
        std::memcpy(up.data() + nb + n_queries + nv, siblings, ns * 4);
        ZK_TRY(dev_h2d(ctx, d, up.data(), up.size() * 4));
    }
    frichip::TraceArgs a{};
    a.betas = d; a.indices = d + nb; a.values = d + nb + n_queries; a.siblings = d + nb + n_queries + nv;
    a.n_queries = (uint32_t)n_queries; a.layers = (uint32_t)layers; a.width = W; a.log_h = (uint32_t)layers + 1u; a.wired = rec ? 2u : wired ? 1u : 0u;
    a.rows = (uint64_t)1 << log_rows; a.trace = d_trace; a.ld = ld; a.finals = d + nb + n_queries + nv + ns;
    a.per_proof = n_proofs > 1 ? (uint32_t)(n_queries / n_proofs) : 0u; a.pt_stride = pt_stride;
    a.pt = pt; a.row_base = row_base; a.pad_from = pad_from < 0 ? (uint64_t)n_queries * (uint64_t)layers : (uint64_t)pad_from;
    const size_t pad = a.rows - a.pad_from;
    const size_t threads = n_queries + (pad < 4096 ? pad : 4096);            // the padding rows are shared among up to 4096 extra threads
    ZK_LAUNCH(frichip::fri_trace_kernel, frichip::fri_trace_kernel_batch, frichip::fri_trace_kernel_bargs, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, ctx->stream, a);
    ZK_HIP(hipGetLastError());
    return dev_d2h(ctx, finals, a.finals, nv * 4);
}
// the same launch with every input ALREADY on the device (round 6: the shard verifier's reduced openings and siblings are produced there): the rec form,
// n_proofs proofs of n_queries / n_proofs queries each; d_finals [n_queries][4] stays on the device
static int fri_gen_trace_dev(zkhip_ctx* ctx, int layers, size_t n_queries, const uint32_t* d_betas, const uint32_t* d_indices, const uint32_t* d_values, const uint32_t* d_siblings,
                             int log_rows, uint32_t* d_trace, size_t ld, uint32_t* d_finals, size_t n_proofs, uint32_t pt_stride) {
    const uint32_t W = frichip::width_of(layers, true, true);
    if (n_proofs < 1 || n_queries % n_proofs || ld < W || log_rows > MAX_LOG_ROWS || ((size_t)1 << log_rows) < n_queries * (size_t)layers) return fail(ZKHIP_ERR_INVALID, "fri_chip_gen_trace: bad arguments");
    frichip::TraceArgs a{};
    a.betas = d_betas; a.indices = d_indices; a.values = d_values; a.siblings = d_siblings;
    a.n_queries = (uint32_t)n_queries; a.layers = (uint32_t)layers; a.width = W; a.log_h = (uint32_t)layers + 1u; a.wired = 2u;
    a.rows = (uint64_t)1 << log_rows; a.trace = d_trace; a.ld = ld; a.finals = d_finals;
    a.per_proof = n_proofs > 1 ? (uint32_t)(n_queries / n_proofs) : 0u; a.pt_stride = pt_stride;
    a.pt = 0u; a.row_base = 0; a.pad_from = (uint64_t)n_queries * (uint64_t)layers;
    const size_t pad = a.rows - a.pad_from;
    const size_t threads = n_queries + (pad < 4096 ? pad : 4096);
    ZK_LAUNCH(frichip::fri_trace_kernel, frichip::fri_trace_kernel_batch, frichip::fri_trace_kernel_bargs, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, ctx->stream, a);
    ZK_HIP(hipGetLastError());
    return ZKHIP_OK;
}
// (round 6: the entries of the first FRI-only machine -- zkhip_fri_chip_gen_trace, zkhip_fri_queries_key / _proof_size, zkhip_prove_fri_queries,
// zkhip_verify_fri_queries -- are gone with the two generations after it; the chip, its program and its trace kernel stay: they are the FOLD chip of the
// shard verifier machines.)

// ================================================================ the wired machine: Merkle paths of the pairs in-circuit
// Four chips, tallest first:
//   P2F      the Poseidon2 chip's FRI-layers variant (poseidon2_chip.cpp): one path per (query, layer) -- a leaf row hashing the pair, then
//            layers - l compression rows up to the layer's root.  Leaf rows RECEIVE (layer, 2 pair-index, E0) and (.., E1); END rows SEND
//            (layer, root half) twice.
//   FRI      the fold chip in its wired form: sends the pairs, and on a query's first row (index, reduced opening).
//   QUERIES  preprocessed (index, reduced opening, 1): receives the starts -- what the verifier computed itself for every query.
//   ROOTS    preprocessed (layer, root): receives the path ends with a main count column (= queries per layer).
// Statement: "for the FRI layer commitments and the (index, reduced opening) pairs in the key, every query's chain opens those
// commitments layer by layer and folds, under the public challenges, to the public final value."  The key no longer holds any FRI
// layer VALUE: a verifier needs the layer roots from the inner proof and its own reduced openings.
namespace zk {
namespace p2chip {
std::shared_ptr<const std::vector<uint32_t>> program_fri_layers(uint32_t n_public);
std::shared_ptr<const std::vector<uint32_t>> program_fri_transcript(uint32_t n_public, uint32_t cap_pub);
std::shared_ptr<const std::vector<uint32_t>> program_fri_indices(uint32_t n_public, uint32_t cap_pub, uint32_t final_pub);
std::vector<uint32_t> permutation_body(uint32_t col_offset, uint32_t* count);      // (shard_verifier.inl: the permutation's constraints behind preprocessed columns)
}
extern std::atomic<uint64_t> g_p2_generation;      // params.cpp
namespace frichip {
namespace {
struct WiredMachine {
    int n = 4;                                                 // chips: 5 with the SAMPLES chip of the query-phase machine
    int32_t log_ns[5]; uint32_t widths[5], pre_widths[5];
    const uint32_t* progs[5]; size_t prog_words[5]; const uint32_t* tabs[5]; size_t tab_words[5];
    std::shared_ptr<const std::vector<uint32_t>> p[5];
    std::vector<uint32_t> fri_tab;
};
inline int log2_ceil(size_t n, int lo) { int l = lo; while (((size_t)1 << l) < n) l++; return l; }
inline size_t p2_rows(int layers, size_t nq, bool transcript = false, bool indices = false) {
    return nq * ((size_t)layers + (size_t)layers * ((size_t)layers + 1) / 2) + (transcript ? (size_t)layers : 0) + (indices ? sample_rows(nq) : 0);
}
// transcript: the TRANSCRIPT machine (zkhip_prove_fri_transcript) -- the same four chips; the Poseidon2 chip in its transcript variant
// (its trace starts with a sponge chain over the layer roots); the ROOTS table with the challenges in its MAIN columns, received once
// from the transcript row of the layer and handed to the layer's fold rows; the fold chip without public challenges.  Public values:
// the final value and the challenger's capacity.  Neither the key nor the verifier holds a challenge: "... under the challenges the
// transcript derives from these layer roots, starting from this challenger state".
// pow_bits >= 0: the QUERY-PHASE machine (zkhip_prove_fri_indices) -- the transcript machine whose sponge chain goes on through the final value and the
// proof-of-work witness of the inner proof (pow_bits: ITS grinding bits); a fifth chip, SAMPLES, takes the bits of the words the chain then hands out:
// the first one's low bits are zero, the others' low bits are the query indices, which the QUERIES table holds in a MAIN column (received from SAMPLES
// by query number, handed to the query's first fold row).  The key holds no index any more: "... at the indices the transcript draws".
void wired_machine(int layers, size_t nq, WiredMachine& m, bool transcript = false, int pow_bits = -1) {
    const uint32_t NP = n_public_of(layers, transcript);
    const bool QM = pow_bits >= 0;
    m.p[0] = QM ? p2chip::program_fri_indices(NP, 4u, 0u) : transcript ? p2chip::program_fri_transcript(NP, 4u) : p2chip::program_fri_layers(NP);       // (the capacity follows the final value)
    m.p[1] = program(layers, true, transcript);
    m.p[2] = table_program(layers, QUERIES_PRE, transcript);
    m.p[3] = table_program(layers, ROOTS_PRE, transcript, transcript ? ROOTS_MAIN_T : 4u);
    m.fri_tab = fri_interactions_wired(layers, transcript);
    m.log_ns[0] = log2_ceil(p2_rows(layers, nq, transcript, QM), 5); m.log_ns[1] = log2_ceil(nq * (size_t)layers, 5);
    m.log_ns[2] = log2_ceil(nq, 5); m.log_ns[3] = log2_ceil((size_t)layers, 5);
    if (QM) { m.log_ns[4] = log2_ceil(sample_rows(nq), 5); if (m.log_ns[4] > m.log_ns[3]) m.log_ns[3] = m.log_ns[4]; }
    if (m.log_ns[3] > m.log_ns[2]) m.log_ns[2] = m.log_ns[3];                // tallest first also for few queries
    m.widths[0] = transcript ? p2chip::WIDTH_T : p2chip::WIDTH; m.widths[1] = width_of(layers, true); m.widths[2] = 4; m.widths[3] = transcript ? ROOTS_MAIN_T : 4u;
    m.pre_widths[0] = 0; m.pre_widths[1] = 0; m.pre_widths[2] = QUERIES_PRE; m.pre_widths[3] = ROOTS_PRE;
    const std::vector<uint32_t>* tabs[5] = {QM ? &p2_interactions_indices() : transcript ? &p2_interactions_transcript() : &p2_interactions(), &m.fri_tab,
                                            QM ? &queries_interactions_indices() : &queries_interactions(),
                                            transcript ? &roots_interactions_transcript() : &roots_interactions(), &samples_interactions()};
    m.n = QM ? 5 : 4;
    if (QM) { m.p[4] = samples_program(layers, pow_bits); m.widths[4] = S_MAIN; m.pre_widths[4] = S_PRE; }
    for (int c = 0; c < m.n; c++) { m.progs[c] = m.p[c]->data(); m.prog_words[c] = m.p[c]->size(); m.tabs[c] = tabs[c]->data(); m.tab_words[c] = tabs[c]->size(); }
}
}  // namespace
}  // namespace frichip
}  // namespace zk

extern "C" {

size_t zkhip_fri_layers_chip_air(int layers, uint32_t* program, size_t cap_words) {
    if (layers < frichip::MIN_LAYERS || layers > frichip::MAX_LAYERS) return 0;
    const auto p = frichip::program(layers, true);
    if (program && cap_words >= p->size()) std::memcpy(program, p->data(), p->size() * 4);
    return p->size();
}
size_t zkhip_p2chip_air_fri_layers(int layers, uint32_t* program, size_t cap_words) {
    if (layers < frichip::MIN_LAYERS || layers > frichip::MAX_LAYERS) return 0;
    const auto p = p2chip::program_fri_layers(frichip::n_public_of(layers));
    if (program && cap_words >= p->size()) std::memcpy(program, p->data(), p->size() * 4);
    return p->size();
}

// the key: QUERIES (index, reduced opening, 1) and ROOTS (layer, root) committed by zkhip_machine_setup
// T: the key of the TRANSCRIPT machine (the ROOTS rows carry the multiplicity 1 of their challenge, which itself is not in the key)
// pow_bits >= 0: the key of the QUERY-PHASE machine: QUERIES lists (query number, reduced opening) -- no index -- and the SAMPLES chip's fixed columns join
static int fri_layers_key_impl(zkhip_ctx* ctx, int layers, size_t n_queries, const uint32_t* indices, const uint32_t* values, const uint32_t* roots,
                               bool T, const zkhip_params* prm, zkhip_machine_key** key, uint32_t vk[8], int pow_bits = -1) {
    CHECK_CTX(ctx);
    const bool QM = pow_bits >= 0;
    if ((!indices && !QM) || !values || !roots || !prm || !key || !vk) return fail(ZKHIP_ERR_INVALID, "fri_layers_key: null argument");
    if (pow_bits > 30) return fail(ZKHIP_ERR_INVALID, "fri_indices_key: 0..30 proof-of-work bits");
    int lr;
    ZK_TRY(frichip::shape_ok(layers, n_queries, &lr));
    if (!frichip::canonical(values, 4 * n_queries) || !frichip::canonical(roots, 8 * (size_t)layers))
        return fail(ZKHIP_ERR_INVALID, "fri_layers_key: values must be canonical");
    const uint32_t RP = frichip::ROOTS_PRE;
    frichip::WiredMachine m;
    frichip::wired_machine(layers, n_queries, m, T, pow_bits);
    std::vector<uint32_t> qt(((size_t)frichip::QUERIES_PRE) << m.log_ns[2], 0u), rt(((size_t)RP) << m.log_ns[3], 0u), st;
    for (size_t q = 0; q < n_queries; q++) {
        if (!QM && indices[q] >> (layers + 1)) return fail(ZKHIP_ERR_INVALID, "fri_layers_key: a query index has more than layers + 1 bits");
        uint32_t* r = qt.data() + frichip::QUERIES_PRE * q;
        r[0] = to_monty(QM ? (uint32_t)q : indices[q]);
        for (int i = 0; i < 4; i++) r[1 + i] = to_monty(values[4 * q + i]);
        r[5] = MONTY_R1;
    }
    for (int l = 0; l < layers; l++) {
        uint32_t* r = rt.data() + RP * (size_t)l;
        r[0] = to_monty((uint32_t)l);
        for (int i = 0; i < 8; i++) r[1 + i] = to_monty(roots[8 * l + i]);
        if (T) r[9] = MONTY_R1;
    }
    void *dq, *dr;
    ZK_TRY(ctx_reserve(ctx, S_REC_C, qt.size() * 4, &dq));
    ZK_TRY(ctx_reserve(ctx, S_REC_D, rt.size() * 4, &dr));
    void* ds = nullptr;
    if (QM) {
        frichip::samples_pre(layers, n_queries, m.log_ns[4], st);
        ZK_TRY(ctx_reserve(ctx, S_REC_E, st.size() * 4, &ds));
        ZK_TRY(dev_h2d(ctx, ds, st.data(), st.size() * 4));
    }
    ZK_TRY(dev_h2d(ctx, dq, qt.data(), qt.size() * 4));
    ZK_TRY(dev_h2d(ctx, dr, rt.data(), rt.size() * 4));
    zkhip_chip pre[5]{};
    for (int c = 0; c < m.n; c++) { pre[c].log_n = m.log_ns[c]; pre[c].width = m.pre_widths[c]; pre[c].ld = m.pre_widths[c]; pre[c].partner = -1; }
    pre[2].d_trace = (const uint32_t*)dq; pre[3].d_trace = (const uint32_t*)dr; pre[4].d_trace = (const uint32_t*)ds;
    return zkhip_machine_setup(ctx, pre, (size_t)m.n, prm, key, vk);
}
int zkhip_fri_indices_key(zkhip_ctx* ctx, int layers, size_t n_queries, int inner_pow_bits, const uint32_t* values, const uint32_t* roots,
                          const zkhip_params* prm, zkhip_machine_key** key, uint32_t vk[8]) {
    if (inner_pow_bits < 0) return fail(ZKHIP_ERR_INVALID, "fri_indices_key: 0..30 proof-of-work bits");
    return fri_layers_key_impl(ctx, layers, n_queries, nullptr, values, roots, true, prm, key, vk, inner_pow_bits);
}

static size_t fri_layers_proof_size_impl(int layers, size_t n_queries, const zkhip_params* prm, bool transcript, int pow_bits = -1) {
    int lr;
    if (!prm || pow_bits > 30 || frichip::shape_ok(layers, n_queries, &lr) != ZKHIP_OK) return 0;
    frichip::WiredMachine m;
    frichip::wired_machine(layers, n_queries, m, transcript, pow_bits);
    return zkhip_machine_proof_size_keyed(m.log_ns, m.widths, m.pre_widths, m.progs, m.prog_words, m.tabs, m.tab_words, (size_t)m.n, prm, frichip::n_public_of(layers, transcript));
}
size_t zkhip_fri_indices_proof_size(int layers, size_t n_queries, int inner_pow_bits, const zkhip_params* prm) {
    return inner_pow_bits < 0 ? 0 : fri_layers_proof_size_impl(layers, n_queries, prm, true, inner_pow_bits);
}
// the query-phase machine's programs that differ from the transcript machine's: which = 0 the Poseidon2 chip (query-phase rows), 1 the SAMPLES chip
size_t zkhip_fri_indices_program(int which, int layers, int inner_pow_bits, uint32_t* program, size_t cap_words) {
    if (layers < frichip::MIN_LAYERS || layers > frichip::MAX_LAYERS || inner_pow_bits < 0 || inner_pow_bits > 30 || which < 0 || which > 1) return 0;
    const auto p = which == 0 ? p2chip::program_fri_indices(frichip::N_PUBLIC_T, 4u, 0u) : frichip::samples_program(layers, inner_pow_bits);
    if (program && cap_words >= p->size()) std::memcpy(program, p->data(), p->size() * 4);
    return p->size();
}
// the transcript machine's two chip programs (the tables' are one identity each)
size_t zkhip_fri_transcript_chip_air(int layers, uint32_t* program, size_t cap_words) {
    if (layers < frichip::MIN_LAYERS || layers > frichip::MAX_LAYERS) return 0;
    const auto p = frichip::program(layers, true, true);
    if (program && cap_words >= p->size()) std::memcpy(program, p->data(), p->size() * 4);
    return p->size();
}
size_t zkhip_p2chip_air_fri_transcript(int layers, uint32_t* program, size_t cap_words) {
    if (layers < frichip::MIN_LAYERS || layers > frichip::MAX_LAYERS) return 0;
    const auto p = p2chip::program_fri_transcript(frichip::n_public_of(layers, true), 4u);
    if (program && cap_words >= p->size()) std::memcpy(program, p->data(), p->size() * 4);
    return p->size();
}

// the Poseidon2 chip's trace for the layer paths of a view, generated on the device (d_trace: [2^log_rows][360], Montgomery)
// capacity != NULL: the transcript variant -- rows 0 .. layers - 1 are the sponge chain over `roots` from `capacity` (ld >= 364), and the
// chain must produce `betas`
static int fri_layers_gen_paths_trace_impl(zkhip_ctx* ctx, int layers, size_t n_queries, const uint32_t* betas, const uint32_t* indices, const uint32_t* values,
                                           const uint32_t* siblings, const uint32_t* roots, const uint32_t* paths, const uint32_t* capacity, int log_rows,
                                           uint32_t* d_trace, size_t ld, const uint32_t* final_witness = nullptr, std::vector<uint32_t>* samples = nullptr) {
    CHECK_CTX(ctx);
    int lr;
    ZK_TRY(frichip::shape_ok(layers, n_queries, &lr));
    const bool T = capacity != nullptr, QM = final_witness != nullptr;           // QM: query-phase rows behind the chain (final value [4], witness -> the sampled words)
    if (QM && (!T || !samples || !frichip::canonical(final_witness, 5))) return fail(ZKHIP_ERR_INVALID, "fri_indices: bad arguments");
    const size_t NQR = QM ? frichip::sample_rows(n_queries) : 0;
    if (!betas || !indices || !values || !siblings || !roots || !paths || !d_trace || ld < (T ? p2chip::WIDTH_T : p2chip::WIDTH)) return fail(ZKHIP_ERR_INVALID, "fri_layers_gen_paths_trace: bad arguments");
    if (T && (!frichip::canonical(capacity, 8) || !frichip::canonical(roots, 8 * (size_t)layers))) return fail(ZKHIP_ERR_INVALID, "fri_transcript: values must be canonical");
    const size_t R = (size_t)layers, np = n_queries * R, used = frichip::p2_rows(layers, n_queries, T, QM), per_q = 4 * R * (R + 1);
    if (log_rows > MAX_LOG_ROWS || ((size_t)1 << log_rows) < used) return fail(ZKHIP_ERR_INVALID, "fri_layers_gen_paths_trace: 2^log_rows rows do not hold the paths");
    if (!frichip::canonical(betas, 4 * R) || !frichip::canonical(values, 4 * n_queries) || !frichip::canonical(siblings, 4 * np) || !frichip::canonical(paths, per_q * n_queries))
        return fail(ZKHIP_ERR_INVALID, "fri_layers_gen_paths_trace: values must be canonical");
    // the pairs of every (query, layer): fold the chain on the host (canonical words), as build_openings does
    std::vector<uint32_t> leaves(8 * np), sib_off(np), idx(np), depths(np), lay(np), mults(np, 1u), starts(np);
    const int H = layers + 1;
    size_t row = T ? R + NQR : 0;                        // the paths lie behind the transcript rows
    for (size_t q = 0; q < n_queries; q++) {
        uint32_t i = indices[q];
        if (i >> H) return fail(ZKHIP_ERR_INVALID, "fri_layers_gen_paths_trace: a query index has more than layers + 1 bits");
        Ext own = frichip::ext_from_canon(values + 4 * q);
        for (int l = 0; l < layers; l++) {
            const size_t p = q * R + (size_t)l;
            const uint32_t bit = i & 1u, k = i >> 1;
            const Ext sib = frichip::ext_from_canon(siblings + 4 * p);
            const Ext e0 = bit ? sib : own, e1 = bit ? own : sib;
            for (int c = 0; c < 4; c++) { leaves[8 * p + c] = from_monty(e0.c[c]); leaves[8 * p + 4 + c] = from_monty(e1.c[c]); }
            const int lh = H - (l + 1);
            sib_off[p] = (uint32_t)(q * per_q + 8 * ((size_t)l * R - (size_t)l * ((size_t)l - 1) / 2));
            idx[p] = k; depths[p] = (uint32_t)lh; lay[p] = (uint32_t)l; starts[p] = (uint32_t)row;
            row += 1 + (size_t)lh;
            const uint32_t xi = finv(fpow(two_adic_generator(lh + 1), reverse_bits(k, lh)));
            const Ext beta = frichip::ext_from_canon(betas + 4 * l);
            own = ext_add(ext_mul_base(ext_add(e0, e1), MONTY_INV2), ext_mul(beta, ext_mul_base(ext_sub(e0, e1), fmul(MONTY_INV2, xi))));
            i = k;
        }
    }
    const size_t np8 = 8 * np, npaths_words = per_q * n_queries;
    // the sponge chain of the transcript rows, walked on the host (layers + query rows permutations): the input state of every row, the
    // challenges and the sampled words it produces
    const size_t n_chain = T ? R + NQR : 0;
    std::vector<uint32_t> chain_in(16 * n_chain);
    if (T) {
        uint32_t st[16];
        for (int j = 0; j < 8; j++) st[8 + j] = to_monty(capacity[j]);
        for (size_t l = 0; l < R; l++) {
            for (int j = 0; j < 8; j++) st[j] = to_monty(roots[8 * l + j]);
            for (int j = 0; j < 16; j++) chain_in[16 * l + j] = from_monty(st[j]);
            p2_permute(st);
            for (int j = 0; j < 4; j++)
                if (from_monty(st[7 - j]) != betas[4 * l + j])
                    return fail(ZKHIP_ERR_INVALID, "fri_transcript: the challenges are not the ones the transcript derives from these roots and this capacity");
        }
        if (QM) samples->resize(8 * NQR);
        for (size_t i = 0; i < NQR; i++) {
            if (i == 0) for (int j = 0; j < 5; j++) st[j] = to_monty(final_witness[j]);
            for (int j = 0; j < 16; j++) chain_in[16 * (R + i) + j] = from_monty(st[j]);
            p2_permute(st);
            for (int j = 0; j < 8; j++) (*samples)[8 * i + j] = from_monty(st[7 - j]);
        }
    }
    // staging, one upload and one download: [leaves | path siblings | 6 x meta | chain inputs] [roots]
    const size_t up_words = np8 + npaths_words + 6 * np + 16 * n_chain, down_words = np8;
    void* stage;
    ZK_TRY(ctx_reserve(ctx, S_STAGE, (up_words + down_words) * 4, &stage));
    uint32_t* d = (uint32_t*)stage;
    uint32_t *d_leaves = d, *d_sibs = d + np8, *d_meta = d_sibs + npaths_words, *d_chain = d_meta + 6 * np, *d_roots = d + up_words;
    {
        std::vector<uint32_t> up(up_words, 0u);
        std::memcpy(up.data(), leaves.data(), np8 * 4);
        std::memcpy(up.data() + np8, paths, npaths_words * 4);
        const std::vector<uint32_t>* meta[6] = {&sib_off, &idx, &depths, &lay, &mults, &starts};
        for (int k = 0; k < 6; k++) std::memcpy(up.data() + np8 + npaths_words + (size_t)k * np, meta[k]->data(), np * 4);
        if (n_chain) std::memcpy(up.data() + np8 + npaths_words + 6 * np, chain_in.data(), 64 * n_chain);
        ZK_TRY(dev_h2d(ctx, d, up.data(), up_words * 4));
    }
    p2chip::LayerPathsArgs a{};
    a.leaves = d_leaves; a.siblings = d_sibs; a.sib_off = d_meta; a.indices = d_meta + np; a.depths = d_meta + 2 * np; a.layers = d_meta + 3 * np;
    a.mults = d_meta + 4 * np; a.starts = d_meta + 5 * np; a.n_paths = np; a.rows = (uint64_t)1 << log_rows; a.used_rows = used;
    a.trace = d_trace; a.ld = ld; a.roots = d_roots;
    if (T) { a.n_transcript = (uint32_t)layers; a.n_query_rows = (uint32_t)NQR; a.chain_inputs = d_chain; }
    ZK_HIP(launch_p2chip_layer_paths(a, ctx->stream));
    std::vector<uint32_t> down(down_words);
    ZK_TRY(dev_d2h(ctx, down.data(), d_roots, down_words * 4));
    const uint32_t* got = down.data();
    for (size_t p = 0; p < np; p++)
        if (std::memcmp(got + 8 * p, roots + 8 * (p % R), 32) != 0)
            return fail(ZKHIP_ERR_INVALID, "fri_layers: the path of query " + std::to_string(p / R) + ", layer " + std::to_string(p % R) + " does not end in the layer's root");
    return ZKHIP_OK;
}

static int prove_fri_layers_impl(zkhip_ctx* ctx, const zkhip_machine_key* key, int layers, size_t n_queries, const uint32_t* betas, const uint32_t* indices,
                                 const uint32_t* values, const uint32_t* siblings, const uint32_t* roots, const uint32_t* paths, const uint32_t* capacity,
                                 const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len, const uint32_t* witness = nullptr, int pow_bits = -1) {
    CHECK_CTX(ctx);
    if (!key || !prm || !proof || !len) return fail(ZKHIP_ERR_INVALID, "prove_fri_layers: null argument");
    int lr;
    ZK_TRY(frichip::shape_ok(layers, n_queries, &lr));
    const bool T = capacity != nullptr, QM = pow_bits >= 0;
    if (QM && (!T || !witness || pow_bits > 30)) return fail(ZKHIP_ERR_INVALID, "prove_fri_indices: bad arguments");
    frichip::WiredMachine m;
    frichip::wired_machine(layers, n_queries, m, T, pow_bits);
    void *t_p2, *t_fri, *t_q, *t_r, *t_s = nullptr;
    ZK_TRY(ctx_reserve(ctx, S_REC_A, ((size_t)m.widths[0] << m.log_ns[0]) * 4, &t_p2));
    ZK_TRY(ctx_reserve(ctx, S_REC_B, ((size_t)m.widths[1] << m.log_ns[1]) * 4, &t_fri));
    ZK_TRY(ctx_reserve(ctx, S_CHIP, ((size_t)4 << m.log_ns[2]) * 4, &t_q));
    ZK_TRY(ctx_reserve(ctx, S_CHIP_B, ((size_t)m.widths[3] << m.log_ns[3]) * 4, &t_r));
    std::vector<uint32_t> finals(4 * n_queries);
    ZK_TRY(fri_gen_trace(ctx, layers, n_queries, betas, indices, values, siblings, m.log_ns[1], (uint32_t*)t_fri, m.widths[1], finals.data(), true));
    for (size_t q = 1; q < n_queries; q++)
        if (std::memcmp(finals.data(), finals.data() + 4 * q, 16) != 0) return fail(ZKHIP_ERR_INVALID, "prove_fri_layers: the chains do not end in one value");
    std::vector<uint32_t> words, smain, drawn;
    uint32_t fw[5];
    if (QM) { std::memcpy(fw, finals.data(), 16); fw[4] = *witness; }
    ZK_TRY(fri_layers_gen_paths_trace_impl(ctx, layers, n_queries, betas, indices, values, siblings, roots, paths, capacity, m.log_ns[0], (uint32_t*)t_p2, m.widths[0],
                                           QM ? fw : nullptr, QM ? &words : nullptr));
    if (QM) {                                            // the SAMPLES chip's rows; the words must be the ones the inner proof's verifier drew
        frichip::samples_main(layers, n_queries, m.log_ns[4], words.data(), smain, drawn);
        if (pow_bits && (words[0] & ((1u << pow_bits) - 1u))) return fail(ZKHIP_ERR_INVALID, "prove_fri_indices: the witness does not satisfy the proof of work");
        if (std::memcmp(drawn.data(), indices, 4 * n_queries) != 0) return fail(ZKHIP_ERR_INVALID, "prove_fri_indices: the query indices are not the ones the transcript draws");
        ZK_TRY(ctx_reserve(ctx, S_REC_C, smain.size() * 4, &t_s));
        ZK_TRY(dev_h2d(ctx, t_s, smain.data(), smain.size() * 4));
    }
    // main columns of the tables: QUERIES none (zeros), ROOTS the number of paths per layer
    std::vector<uint32_t> rmain((size_t)m.widths[3] << m.log_ns[3], 0u);
    for (int l = 0; l < layers; l++) {
        uint32_t* r = rmain.data() + (size_t)m.widths[3] * (size_t)l;
        r[0] = to_monty((uint32_t)n_queries + (T ? 1u : 0u));     // the paths, and the transcript row that absorbs the root
        if (T) { for (int i = 0; i < 4; i++) r[1 + i] = to_monty(betas[4 * l + i]); r[5] = to_monty((uint32_t)n_queries); }
    }
    std::vector<uint32_t> qmain;
    if (QM) {                                            // QUERIES main column 0: the index of query q
        qmain.assign((size_t)4 << m.log_ns[2], 0u);
        for (size_t q = 0; q < n_queries; q++) qmain[4 * q] = to_monty(indices[q]);
        ZK_TRY(dev_h2d(ctx, t_q, qmain.data(), qmain.size() * 4));
    } else
        ZK_TRY(dev_memset(ctx, t_q, 0, ((size_t)4 << m.log_ns[2]) * 4));
    ZK_TRY(dev_h2d(ctx, t_r, rmain.data(), rmain.size() * 4));
    std::vector<uint32_t> pv(frichip::n_public_of(layers, T));
    if (T) { std::memcpy(pv.data(), finals.data(), 16); std::memcpy(pv.data() + 4, capacity, 32); }
    else { std::memcpy(pv.data(), betas, 16 * (size_t)layers); std::memcpy(pv.data() + 4 * (size_t)layers, finals.data(), 16); }
    zkhip_chip chips[5]{};
    void* tr[5] = {t_p2, t_fri, t_q, t_r, t_s};
    for (int c = 0; c < m.n; c++) { chips[c].d_trace = (const uint32_t*)tr[c]; chips[c].ld = m.widths[c]; chips[c].log_n = m.log_ns[c]; chips[c].width = m.widths[c]; chips[c].partner = -1; }
    return zkhip_prove_machine_keyed(ctx, key, chips, m.progs, m.prog_words, m.tabs, m.tab_words, (size_t)m.n, pv.data(), pv.size(), prm, proof, cap, len);
}
int zkhip_prove_fri_indices(zkhip_ctx* ctx, const zkhip_machine_key* key, int layers, size_t n_queries, int inner_pow_bits, const uint32_t* betas, const uint32_t* indices,
                            const uint32_t* values, const uint32_t* siblings, const uint32_t* roots, const uint32_t* paths, const uint32_t capacity[8], uint32_t witness,
                            const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len) {
    if (!capacity || inner_pow_bits < 0) return fail(ZKHIP_ERR_INVALID, "prove_fri_indices: bad arguments");
    return prove_fri_layers_impl(ctx, key, layers, n_queries, betas, indices, values, siblings, roots, paths, capacity, prm, proof, cap, len, &witness, inner_pow_bits);
}

static int verify_fri_layers_impl(const uint8_t* proof, size_t len, int layers, size_t n_queries, const uint32_t* betas, const uint32_t final_value[4],
                                  const uint32_t* capacity, const uint32_t vk[8], const zkhip_params* prm, int* reason, int pow_bits = -1) {
    int lr;
    if (!proof || (!betas && !capacity) || !final_value || !vk || !prm || pow_bits > 30 || frichip::shape_ok(layers, n_queries, &lr) != ZKHIP_OK) {
        if (reason) *reason = 1;
        return fail(ZKHIP_ERR_VERIFY, "verify_fri_layers: bad arguments");
    }
    const bool T = capacity != nullptr;
    std::vector<uint32_t> pv(frichip::n_public_of(layers, T));
    if (T) { std::memcpy(pv.data(), final_value, 16); std::memcpy(pv.data() + 4, capacity, 32); }
    else { std::memcpy(pv.data(), betas, 16 * (size_t)layers); std::memcpy(pv.data() + 4 * (size_t)layers, final_value, 16); }
    frichip::WiredMachine m;
    frichip::wired_machine(layers, n_queries, m, T, pow_bits);
    return zkhip_verify_machine_keyed(proof, len, m.log_ns, m.widths, m.pre_widths, vk, m.progs, m.prog_words, m.tabs, m.tab_words, (size_t)m.n, pv.data(), pv.size(), prm, reason);
}
int zkhip_verify_fri_indices(const uint8_t* proof, size_t len, int layers, size_t n_queries, int inner_pow_bits, const uint32_t final_value[4], const uint32_t capacity[8],
                             const uint32_t vk[8], const zkhip_params* prm, int* reason) {
    if (!capacity || inner_pow_bits < 0) { if (reason) *reason = 1; return fail(ZKHIP_ERR_VERIFY, "verify_fri_indices: bad arguments"); }
    return verify_fri_layers_impl(proof, len, layers, n_queries, nullptr, final_value, capacity, vk, prm, reason, inner_pow_bits);
}


// ---- many shard proofs in one call: the compress-like step of the path (sp1.rs:116: core -> COMPRESS verifies the shard proofs; BASELINE
// configs[3]: a proof of many shards).  Per job: the FRI view of the shard proof (host, on a small pool before any device work), the key of
// its query-phase machine and the machine's proof -- dealt over the devices like every batch of this library: lock-step lanes (batch.h) for
// the launch-bound sizes, one context per worker otherwise.  All jobs share (log_n, width, inner): one machine shape.
int zkhip_prove_fri_indices_batch(const int* devices, int n_devices, zkhip_fri_job* jobs, int n_jobs, int log_n, uint32_t width,
                                  const zkhip_params* inner, const zkhip_params* outer, int in_flight_per_device, int verify) {
    if (!jobs || n_jobs < 0 || !inner || !outer) return fail(ZKHIP_ERR_INVALID, "prove_fri_indices_batch: bad arguments");
    for (int i = 0; i < n_jobs; i++) { jobs[i].status = ZKHIP_ERR_INVALID; jobs[i].proof_len = 0; }
    const int layers = log_n;
    const size_t nq = (size_t)inner->num_queries;
    int lr;
    if (inner->log_blowup != 1 || inner->pow_bits < 0 || inner->pow_bits > 30 || frichip::shape_ok(layers, nq, &lr) != ZKHIP_OK)
        return fail(ZKHIP_ERR_INVALID, "prove_fri_indices_batch: fold-by-2, blowup-2 shard proofs of 2^2 .. 2^22 rows");
    std::vector<int> devs;
    const int rc = resolve_devices(devices, n_devices, "prove_fri_indices_batch", devs);
    if (rc == ZKHIP_ERR_NO_DEVICE) {
        for (int i = 0; i < n_jobs; i++) jobs[i].status = ZKHIP_ERR_NO_DEVICE;
        return n_jobs == 0 ? ZKHIP_OK : rc;
    }
    if (rc != ZKHIP_OK) return rc;
    // (a) the views: one host pass over every shard proof, on a small pool that runs AHEAD of the device work (job i waits for view i only)
    struct View {
        std::vector<uint32_t> betas, indices, values, siblings, roots, paths; uint32_t fin[4], tr[10]; int rc = ZKHIP_OK; std::string msg;
        std::mutex mu; std::condition_variable cv; bool done = false;
    };
    std::vector<View> views((size_t)n_jobs);
    const size_t R = (size_t)layers;
    HostPool viewers(8);
    for (int i = 0; i < n_jobs; i++)
        viewers.submit([&, i] {
            View& v = views[(size_t)i];
            const zkhip_fri_job& j = jobs[i];
            v.betas.resize(4 * R); v.indices.resize(nq); v.values.resize(4 * nq); v.siblings.resize(4 * nq * R); v.roots.resize(8 * R);
            v.paths.resize(zkhip_fri_view_path_words(layers) * nq);
            v.rc = zkhip_fri_view_all(j.shard_proof, j.shard_proof_len, log_n, width, j.public_values, j.n_public, inner, v.betas.data(), v.fin,
                                      v.indices.data(), v.values.data(), v.siblings.data(), v.roots.data(), v.paths.data(), v.tr);
            if (v.rc != ZKHIP_OK) v.msg = zkhip_last_error();
            { std::lock_guard<std::mutex> lk(v.mu); v.done = true; }
            v.cv.notify_all();
        });
    // (b) key + proof per job on the devices
    std::unique_ptr<HostPool> checkers;
    std::vector<std::string> check_msg((size_t)n_jobs);
    std::vector<char> ran;
    const size_t need = zkhip_fri_indices_proof_size(layers, nq, inner->pow_bits, outer);
    auto run = [&](zkhip_ctx* ctx, int i) {
        zkhip_fri_job& j = jobs[i];
        View& v = views[(size_t)i];
        { std::unique_lock<std::mutex> lk(v.mu); v.cv.wait(lk, [&] { return v.done; }); }
        if (v.rc != ZKHIP_OK) { batch_leave(); j.status = v.rc; set_error(v.msg); return v.rc; }
        int r = j.proof && j.proof_cap >= need ? ZKHIP_OK : fail(ZKHIP_ERR_INVALID, "prove_fri_indices_batch: proof buffer too small (zkhip_fri_indices_proof_size)");
        zkhip_machine_key* key = nullptr;
        size_t len = 0;
        if (r == ZKHIP_OK) r = zkhip_fri_indices_key(ctx, layers, nq, inner->pow_bits, v.values.data(), v.roots.data(), outer, &key, j.vk);
        if (r == ZKHIP_OK)
            r = zkhip_prove_fri_indices(ctx, key, layers, nq, inner->pow_bits, v.betas.data(), v.indices.data(), v.values.data(), v.siblings.data(), v.roots.data(),
                                        v.paths.data(), v.tr, v.tr[9], outer, j.proof, j.proof_cap, &len);
        if (key) { (void)dev_sync(ctx); zkhip_machine_key_destroy(key); }
        batch_leave();                                           // (lock-step batch: the rest is host work)
        std::memcpy(j.final_value, v.fin, 16);
        std::memcpy(j.capacity, v.tr, 32);
        j.status = r;
        j.proof_len = r == ZKHIP_OK ? len : 0;
        if (r == ZKHIP_OK && verify) {
            auto check = [&jobs, &check_msg, i, len, layers, nq, ipow = inner->pow_bits, p = *outer] {
                zkhip_fri_job& jj = jobs[i];
                const int c = zkhip_verify_fri_indices(jj.proof, len, layers, nq, ipow, jj.final_value, jj.capacity, jj.vk, &p, nullptr);
                if (c != ZKHIP_OK) { jj.status = c; jj.proof_len = 0; check_msg[(size_t)i] = zkhip_last_error(); }
            };
            if (checkers && t_batcher) checkers->submit(check);
            else { check(); r = j.status; if (r != ZKHIP_OK) set_error(check_msg[(size_t)i]); }
        }
        return r;
    };
    frichip::WiredMachine m;
    frichip::wired_machine(layers, nq, m, true, inner->pow_bits);
    uint64_t cells = 0;
    for (int c = 0; c < m.n; c++) cells += ((uint64_t)(m.widths[c] + m.pre_widths[c])) << m.log_ns[c];
    const int nd = (int)devs.size(), max_batch = lockstep_batch();
    int rcj;
    if (max_batch > 1 && cells <= LOCKSTEP_MAX_CELLS && n_jobs >= 2 * nd) {
        if (verify) checkers.reset(new HostPool(8));
        std::vector<int> shape((size_t)n_jobs, 0);
        rcj = deal_jobs_lockstep(devs.data(), nd, n_jobs, shape.data(), max_batch, lockstep_lanes(), run, ran,
                                 cells << (outer->log_blowup > 1 ? outer->log_blowup - 1 : 0));
    } else
        rcj = deal_jobs(devs.data(), nd, n_jobs, in_flight_per_device, run, ran);
    std::string msg = rcj != ZKHIP_OK ? zkhip_last_error() : "";
    viewers.wait();                                              // (a failing call may return before every view was looked at)
    if (checkers) {
        checkers->wait();
        for (int i = 0; i < n_jobs && rcj == ZKHIP_OK; i++)
            if (!check_msg[(size_t)i].empty()) { rcj = jobs[i].status; msg = check_msg[(size_t)i]; }
    }
    if (rcj != ZKHIP_OK) set_error(msg);
    return rcj;
}

}  // extern "C"

#include "shard_verifier.inl"
#include "machine_verifier.inl"
