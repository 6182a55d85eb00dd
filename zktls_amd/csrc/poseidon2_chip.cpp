// poseidon2_chip.cpp -- a second real chip on the constraint-program path: the width-16 Poseidon2 permutation
// with Merkle-path chaining.  A STARK verifier inside a STARK -- the recursion the reference asks for behind SP1ProofMode::Groth16
// (crates/guest-prover-sp1/src/sp1.rs:116: core -> compress -> shrink -> wrap) -- spends its rows on Poseidon2: the Merkle paths of every
// FRI query and the transcript; upstream these are sp1-recursion's Poseidon2 chips (reference Cargo.lock:6172 ff.).  This is the chip and
// the first thing it is used for, Merkle openings in-circuit -- not a recursion machine.
//
// One row = one permutation of THIS library's parameter set (whatever zkhip_load_poseidon2_params left in effect: the round constants
// are coefficients of the program, so the program -- and its digest in every proof -- follows the tables).  360 columns, every
// constraint of degree <= 3 with its selector:
//   IN 16 | S0 16 (after the initial external layer) | per external round r: X3E[r] 16 = (y + rc)^3, OUTE[r] 16 = the state after the round
//   (x^7 = x^3 x^3 x, then the external matrix) | per internal round r: S0P[r], X3P[r], SBP[r] = element 0 before the S-box, its cube, its
//   seventh power -- the other fifteen elements stay linear forms over OUTE[3] and the SBP columns so far | SP 16 (after the internal rounds)
//   | D 8 = the digest-carrying half of the input, IN[j] (1 - BIT) + IN[8 + j] BIT | BIT CH END booleans: right child, continues the
//   previous row's digest (D = previous OUTE[7][0..8]), ends a path (OUTE[7][0..8] = the public root) | CNT running count of END rows,
//   the last row's is the public count | SPG SS booleans for LEAF HASHING, the overwrite-mode sponge over an opened row of 8 k values:
//   SS = the row absorbs a leaf's first 8 values (capacity half IN[8..16] zero), SPG = it absorbs the next 8 (capacity half = the previous
//   row's OUTE[7][8..16]); the row after the last sponge row starts the path with CH = 1, its D being the leaf digest | three unused.
// Public values: root[8], count.  A proof says: "I know `count` openings that end in `root`" (sponge + truncated-permutation compression:
// the commitments of this library and of p3-merkle-tree); rows or leaf digests, siblings and positions are the prover's.  tests/poseidon2_air.py writes the
// same program and trace independently (on tests/pyref.py's Poseidon2); the words must be equal.
#include <atomic>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "air.h"
#include "context.h"
#include "p2chip.h"

namespace zk {
extern std::atomic<uint64_t> g_p2_generation;      // params.cpp
namespace p2chip {
namespace {

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
    void add(uint32_t selector, const Terms& terms) {          // terms with coefficient 0 are omitted
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
// x3 - (c + k)^3
Terms cube_def(uint32_t x3, uint32_t c, uint32_t k) {
    return Terms{{1u, {var(x3)}}, {P - 1, {var(c), var(c), var(c)}}, {neg(3ull * k), {var(c), var(c)}}, {neg(3ull * mulm(k, k)), {var(c)}}, {neg(mulm(mulm(k, k), k)), {}}};
}
typedef std::map<uint32_t, uint32_t> Form;                      // column -> coefficient (canonical), ascending columns
Terms linear_def(uint32_t col, const Form& f) {
    Terms t{{1u, {var(col)}}};
    for (const auto& e : f) t.push_back(Term{neg(e.second), {var(e.first)}});
    return t;
}

// the constraints that make a row ONE permutation out = poseidon2(in): columns IN .. SP, shared by every variant of the chip
void permutation_part(Builder& b) {
    // the matrices and constants of the tables in effect, canonical
    uint32_t ME[16][16], rc_e[8][16], rc_i[13], diag[16];
    for (int j = 0; j < 16; j++) {
        uint32_t e[16] = {0};
        e[j] = MONTY_R1;
        p2_external_linear(e);
        for (int i = 0; i < 16; i++) ME[i][j] = from_monty(e[i]);
    }
    for (int r = 0; r < 8; r++) for (int i = 0; i < 16; i++) rc_e[r][i] = from_monty(g_p2_tables.k16.ext_rc[r][i]);
    for (int r = 0; r < 13; r++) rc_i[r] = from_monty(g_p2_tables.k16.int_rc[r]);
    for (int i = 0; i < 16; i++) diag[i] = from_monty(g_p2_tables.k16.diag[i]);
    for (uint32_t i = 0; i < 16; i++) {
        Form f;
        for (uint32_t j = 0; j < 16; j++) f[IN + j] = ME[i][j];
        b.add(ALL, linear_def(S0 + i, f));
    }
    auto external_round = [&](uint32_t r) {
        const uint32_t c0 = ext_input(r);
        for (uint32_t i = 0; i < 16; i++) b.add(ALL, cube_def(x3e(r) + i, c0 + i, rc_e[r][i]));
        for (uint32_t i = 0; i < 16; i++) {
            Terms t{{1u, {var(oute(r) + i)}}};
            for (uint32_t j = 0; j < 16; j++) {
                const uint32_t x3 = var(x3e(r) + j), c = var(c0 + j);
                t.push_back(Term{neg(ME[i][j]), {x3, x3, c}});
                t.push_back(Term{neg(mulm(ME[i][j], rc_e[r][j])), {x3, x3}});
            }
            b.add(ALL, t);
        }
    };
    for (uint32_t r = 0; r < 4; r++) external_round(r);
    Form lin[16];
    for (uint32_t i = 0; i < 16; i++) lin[i][oute(3) + i] = 1u;
    for (uint32_t r = 0; r < 13; r++) {
        const uint32_t k = rc_i[r];
        b.add(ALL, linear_def(s0p(r), lin[0]));
        b.add(ALL, cube_def(x3p(r), s0p(r), k));
        b.add(ALL, Terms{{1u, {var(sbp(r))}}, {P - 1, {var(x3p(r)), var(x3p(r)), var(s0p(r))}}, {neg(k), {var(x3p(r)), var(x3p(r))}}});
        lin[0].clear();
        lin[0][sbp(r)] = 1u;
        Form total;
        for (const Form& f : lin) for (const auto& e : f) total[e.first] = (uint32_t)(((uint64_t)total[e.first] + e.second) % P);
        for (uint32_t i = 0; i < 16; i++) {                      // M_I = J + diag(d): s_i' = d_i s_i + sum
            Form nf;
            for (const auto& e : total) {
                const auto it = lin[i].find(e.first);
                nf[e.first] = (uint32_t)(((uint64_t)mulm(diag[i], it == lin[i].end() ? 0u : it->second) + e.second) % P);
            }
            lin[i] = nf;
        }
    }
    for (uint32_t i = 0; i < 16; i++) b.add(ALL, linear_def(SP + i, lin[i]));
    for (uint32_t r = 4; r < 8; r++) external_round(r);
}

std::vector<uint32_t> build_program(bool fri_layers = false, uint32_t n_public = N_PUBLIC, int transcript = -1, int queries = -1) {
    Builder b;
    permutation_part(b);
    for (uint32_t j = 0; j < 8; j++)
        b.add(ALL, Terms{{1u, {var(D + j)}}, {P - 1, {var(IN + j)}}, {1u, {var(BIT), var(IN + j)}}, {P - 1, {var(BIT), var(IN + 8 + j)}}});
    for (uint32_t f : {BIT, CH, END, SPG, SS}) b.add(ALL, Terms{{1u, {var(f), var(f)}}, {P - 1, {var(f)}}});
    b.add(FIRST, Terms{{1u, {var(CH)}}});
    b.add(FIRST, Terms{{1u, {var(SPG)}}});
    for (uint32_t j = 0; j < 8; j++) b.add(TRANSITION, Terms{{1u, {var(SPG, true), var(IN + 8 + j, true)}}, {P - 1, {var(SPG, true), var(oute(7) + 8 + j)}}});
    for (uint32_t j = 0; j < 8; j++) b.add(ALL, Terms{{1u, {var(SS), var(IN + 8 + j)}}});
    for (uint32_t j = 0; j < 8; j++) b.add(TRANSITION, Terms{{1u, {var(CH, true), var(D + j, true)}}, {P - 1, {var(CH, true), var(oute(7) + j)}}});
    if (!fri_layers) for (uint32_t j = 0; j < 8; j++) b.add(ALL, Terms{{1u, {var(END), var(oute(7) + j)}}, {P - 1, {var(END), pub(j)}}});
    b.add(FIRST, Terms{{1u, {var(CNT)}}, {P - 1, {var(END)}}});
    b.add(TRANSITION, Terms{{1u, {var(CNT, true)}}, {P - 1, {var(CNT)}}, {P - 1, {var(END, true)}}});
    if (!fri_layers) b.add(LAST, Terms{{1u, {var(CNT)}}, {P - 1, {pub(8)}}});
    if (fri_layers) {
        // a row inside a path (leaf or compression, not the last) is followed by a row that continues it; the trace does not end inside one
        b.add(TRANSITION, Terms{{1u, {var(SS)}}, {P - 1, {var(SS), var(CH, true)}}, {1u, {var(CH)}}, {P - 1, {var(CH), var(CH, true)}},
                                {P - 1, {var(END)}}, {1u, {var(END), var(CH, true)}}});
        b.add(LAST, Terms{{1u, {var(SS)}}, {1u, {var(CH)}}, {P - 1, {var(END)}}});
        b.add(ALL, Terms{{1u, {var(END)}}, {P - 1, {var(END), var(CH)}}});                                   // a path ends on a compression row
        b.add(ALL, Terms{{1u, {var(SS), var(BIT)}}});
        b.add(ALL, Terms{{1u, {var(SS), var(CH)}}});
        b.add(TRANSITION, Terms{{1u, {var(CH, true), var(LNP, true)}}, {P - 1, {var(CH, true), var(LNP)}}});   // one layer per path
        b.add(TRANSITION, Terms{{1u, {var(CH, true), var(KP)}}, {P - 2, {var(CH, true), var(KP, true)}}, {P - 1, {var(CH, true), var(BIT)}}});   // KP = 2 KP' + BIT
        b.add(ALL, Terms{{1u, {var(END), var(KP)}}, {P - 1, {var(END), var(BIT)}}});                           // exactly depth bits
        b.add(ALL, Terms{{1u, {var(M)}}, {P - 1, {var(M), var(SS)}}});                                         // tuples are received on leaf rows only
    }
    if (transcript >= 0) {                                  // the transcript variant (p2chip.h): public values transcript .. transcript + 7 = the capacity
        b.add(ALL, Terms{{1u, {var(TRS), var(TRS)}}, {P - 1, {var(TRS)}}});
        b.add(FIRST, Terms{{1u, {var(TRS)}}, {P - 1, {}}});                                                    // the trace starts with the transcript
        b.add(FIRST, Terms{{1u, {var(LNP)}}});
        for (uint32_t j = 0; j < 8; j++) b.add(FIRST, Terms{{1u, {var(IN + 8 + j)}}, {P - 1, {pub((uint32_t)transcript + j)}}});
        b.add(TRANSITION, Terms{{1u, {var(TRS, true)}}, {P - 1, {var(TRS), var(TRS, true)}}});                 // transcript rows are a prefix
        if (queries < 0) b.add(TRANSITION, Terms{{1u, {var(TRS, true), var(LNP, true)}}, {P - 1, {var(TRS, true), var(LNP)}}, {P - 1, {var(TRS, true)}}});
        b.add(TRANSITION, Terms{{1u, {var(TRS, true)}}, {P - 1, {var(TRS, true), var(SPG, true)}}});           // ... chained through the capacity
        if (queries < 0) b.add(ALL, Terms{{1u, {var(SPG)}}, {P - 1, {var(SPG), var(TRS)}}});                   // and nothing else is
        for (uint32_t f : {CH, END, SS, BIT, M}) b.add(ALL, Terms{{1u, {var(TRS), var(f)}}});
    }
    if (queries >= 0) {                                     // the query-phase rows (p2chip.h): public values queries .. queries + 3 = the final value
        const uint32_t o7 = oute(7);
        for (uint32_t f : {QP, QF}) b.add(ALL, Terms{{1u, {var(f), var(f)}}, {P - 1, {var(f)}}});
        b.add(ALL, Terms{{1u, {var(QF)}}, {P - 1, {var(QF), var(QP)}}});                                       // the first query row is one
        b.add(ALL, Terms{{1u, {var(QP), var(TRS)}}});
        b.add(FIRST, Terms{{1u, {var(QF)}}});
        b.add(TRANSITION, Terms{{1u, {var(QF, true)}}, {P - 1, {var(TRS)}}, {1u, {var(TRS), var(TRS, true)}}});                   // QF' = TRS (1 - TRS'): right behind the chain
        b.add(TRANSITION, Terms{{1u, {var(QP, true)}}, {P - 1, {var(QP, true), var(QP)}}, {P - 1, {var(QF, true)}}});             // a query row is the first or follows one
        b.add(TRANSITION, Terms{{1u, {var(TRS, true), var(LNP, true)}}, {P - 1, {var(TRS, true), var(LNP)}}, {P - 1, {var(TRS, true)}},    // the row counter runs on
                                {1u, {var(QP, true), var(LNP, true)}}, {P - 1, {var(QP, true), var(LNP)}}, {P - 1, {var(QP, true)}}});
        b.add(TRANSITION, Terms{{1u, {var(QP, true)}}, {P - 1, {var(QP, true), var(SPG, true)}}});             // the capacity is kept
        b.add(ALL, Terms{{1u, {var(SPG)}}, {P - 1, {var(SPG), var(TRS)}}, {P - 1, {var(SPG), var(QP)}}});
        for (uint32_t j = 5; j < 8; j++)                                                                       // rate words the inputs do not reach
            b.add(TRANSITION, Terms{{1u, {var(QP, true), var(IN + j, true)}}, {P - 1, {var(QP, true), var(o7 + j)}}});
        for (uint32_t j = 0; j < 5; j++)                                                                       // later rows: nothing absorbed
            b.add(TRANSITION, Terms{{1u, {var(QP, true), var(IN + j, true)}}, {P - 1, {var(QP, true), var(o7 + j)}},
                                    {P - 1, {var(QF, true), var(IN + j, true)}}, {1u, {var(QF, true), var(o7 + j)}}});
        for (uint32_t j = 0; j < 4; j++)                                                                       // the first absorbs the final value (and a free witness)
            b.add(ALL, Terms{{1u, {var(QF), var(IN + j)}}, {P - 1, {var(QF), pub((uint32_t)queries + j)}}});
        for (uint32_t f : {CH, END, SS, BIT, M}) b.add(ALL, Terms{{1u, {var(QP), var(f)}}});
    }
    std::vector<uint32_t> p{AIR_MAGIC, 1u, transcript >= 0 ? WIDTH_T : WIDTH, b.count, n_public, (uint32_t)(6 + b.body.size())};
    p.insert(p.end(), b.body.begin(), b.body.end());
    return p;
}

// the program of the Poseidon2 tables in effect (rebuilt when zkhip_load_poseidon2_params / _reset_ changes them)
std::shared_ptr<const std::vector<uint32_t>> program() {
    static std::mutex mu;
    static std::shared_ptr<const std::vector<uint32_t>> cached;
    static uint64_t cached_gen = ~0ull;
    std::lock_guard<std::mutex> lk(mu);
    const uint64_t gen = g_p2_generation.load();
    if (!cached || cached_gen != gen) { cached = std::make_shared<const std::vector<uint32_t>>(build_program()); cached_gen = gen; }
    return cached;
}

}  // namespace
// the permutation's constraints alone, as program body words with every column moved up by `col_offset` (the recursion machine's chip has
// preprocessed columns in front, fri_chip.hip / shard_verifier.inl); *count = the number of constraints
std::vector<uint32_t> permutation_body(uint32_t col_offset, uint32_t* count) {
    Builder b;
    permutation_part(b);
    std::vector<uint32_t>& w = b.body;
    size_t p = 0;
    for (uint32_t k = 0; k < b.count; k++) {
        const uint32_t nt = w[p + 1];
        p += 2;
        for (uint32_t t = 0; t < nt; t++) {
            const uint32_t d = w[p + 1];
            p += 2;
            for (uint32_t j = 0; j < d; j++, p++) if ((w[p] >> 30) < 2u) w[p] += col_offset;
        }
    }
    *count = b.count;
    return w;
}
// the FRI-layers variant for a machine with n_public public values (fri_chip.hip); follows the Poseidon2 tables in effect like program()
std::shared_ptr<const std::vector<uint32_t>> program_fri_layers(uint32_t n_public) {
    static std::mutex mu;
    static std::map<uint32_t, std::shared_ptr<const std::vector<uint32_t>>> cache;
    static uint64_t cached_gen = ~0ull;
    std::lock_guard<std::mutex> lk(mu);
    const uint64_t gen = g_p2_generation.load();
    if (cached_gen != gen) { cache.clear(); cached_gen = gen; }
    auto it = cache.find(n_public);
    if (it == cache.end()) it = cache.emplace(n_public, std::make_shared<const std::vector<uint32_t>>(build_program(true, n_public))).first;
    return it->second;
}
// ... and its transcript variant: the capacity = public values cap_pub .. cap_pub + 7 of n_public
std::shared_ptr<const std::vector<uint32_t>> program_fri_transcript(uint32_t n_public, uint32_t cap_pub) {
    static std::mutex mu;
    static std::map<uint64_t, std::shared_ptr<const std::vector<uint32_t>>> cache;
    static uint64_t cached_gen = ~0ull;
    std::lock_guard<std::mutex> lk(mu);
    const uint64_t gen = g_p2_generation.load();
    if (cached_gen != gen) { cache.clear(); cached_gen = gen; }
    const uint64_t key = ((uint64_t)n_public << 32) | cap_pub;
    auto it = cache.find(key);
    if (it == cache.end()) it = cache.emplace(key, std::make_shared<const std::vector<uint32_t>>(build_program(true, n_public, (int)cap_pub))).first;
    return it->second;
}
// ... and the query-phase variant of that: the final value = public values final_pub .. final_pub + 3
std::shared_ptr<const std::vector<uint32_t>> program_fri_indices(uint32_t n_public, uint32_t cap_pub, uint32_t final_pub) {
    static std::mutex mu;
    static std::map<uint64_t, std::shared_ptr<const std::vector<uint32_t>>> cache;
    static uint64_t cached_gen = ~0ull;
    std::lock_guard<std::mutex> lk(mu);
    const uint64_t gen = g_p2_generation.load();
    if (cached_gen != gen) { cache.clear(); cached_gen = gen; }
    const uint64_t key = ((uint64_t)n_public << 32) | (cap_pub << 16) | final_pub;
    auto it = cache.find(key);
    if (it == cache.end()) it = cache.emplace(key, std::make_shared<const std::vector<uint32_t>>(build_program(true, n_public, (int)cap_pub, (int)final_pub))).first;
    return it->second;
}
namespace {

int paths_shape(size_t n_paths, int depth, uint32_t row_width, int* log_n) {
    if (row_width % 8 != 0 || row_width > 1024) return fail(ZKHIP_ERR_INVALID, "merkle paths: the opened row width is 0 (leaf digests are given) or a multiple of 8 up to 1024");
    const size_t per = (size_t)row_width / 8 + (size_t)(depth > 0 ? depth : 0);
    if (n_paths < 1 || depth < 1 || depth > 32 || n_paths > ((size_t)1 << MAX_LOG_ROWS) / per) return fail(ZKHIP_ERR_INVALID, "merkle paths: 1..2^22 rows of paths, depth 1..32");
    int ln = 5;
    while (((size_t)1 << ln) < n_paths * per) ln++;
    *log_n = ln;
    return ln <= MAX_LOG_ROWS ? ZKHIP_OK : fail(ZKHIP_ERR_INVALID, "merkle paths: more than 2^22 rows");
}

}  // namespace
}  // namespace p2chip
}  // namespace zk

using namespace zk;

#define CHECK_CTX(ctx)                                                  \
    do {                                                                \
        if (!(ctx)) return fail(ZKHIP_ERR_INVALID, "null context");     \
        ZK_HIP(hipSetDevice((ctx)->device));                            \
    } while (0)

extern "C" {

size_t zkhip_p2chip_air(uint32_t* program, size_t cap_words) {
    const auto p = p2chip::program();
    if (program && cap_words >= p->size()) std::memcpy(program, p->data(), p->size() * 4);
    return p->size();
}

int zkhip_p2chip_gen_merkle_trace(zkhip_ctx* ctx, const uint32_t* leaves, uint32_t row_width, const uint32_t* siblings, const uint32_t* indices, size_t n_paths,
                                  int depth, int log_n, uint32_t* d_trace, size_t ld, uint32_t* roots) {
    CHECK_CTX(ctx);
    if (!leaves || !siblings || !indices || !d_trace || !roots || ld < p2chip::WIDTH) return fail(ZKHIP_ERR_INVALID, "p2chip_gen_merkle_trace: bad arguments");
    int need;
    ZK_TRY(p2chip::paths_shape(n_paths, depth, row_width, &need));
    if (log_n < need || log_n > MAX_LOG_ROWS) return fail(ZKHIP_ERR_INVALID, "p2chip_gen_merkle_trace: 2^log_n rows do not hold the paths");
    const size_t nl = n_paths * (row_width ? row_width : 8), ns = n_paths * (size_t)depth * 8, nr = n_paths * 8;
    for (size_t i = 0; i < nl; i++) if (leaves[i] >= P) return fail(ZKHIP_ERR_INVALID, "p2chip_gen_merkle_trace: leaves must be canonical");
    for (size_t i = 0; i < ns; i++) if (siblings[i] >= P) return fail(ZKHIP_ERR_INVALID, "p2chip_gen_merkle_trace: siblings must be canonical");
    void* stage;
    ZK_TRY(ctx_reserve(ctx, S_STAGE, (nl + ns + n_paths + nr) * 4, &stage));
    uint32_t* d = (uint32_t*)stage;
    ZK_HIP(hipMemcpyAsync(d, leaves, nl * 4, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(d + nl, siblings, ns * 4, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(d + nl + ns, indices, n_paths * 4, hipMemcpyHostToDevice, ctx->stream));
    p2chip::MerkleTraceArgs a{};
    a.leaves = d; a.row_width = row_width; a.siblings = d + nl; a.indices = d + nl + ns; a.n_paths = n_paths; a.rows = (uint64_t)1 << log_n; a.depth = (uint32_t)depth;
    a.trace = d_trace; a.ld = ld; a.roots = d + nl + ns + n_paths;
    ZK_HIP(launch_p2chip_merkle(a, ctx->stream));
    ZK_HIP(hipMemcpyAsync(roots, a.roots, nr * 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

size_t zkhip_merkle_paths_proof_size(size_t n_paths, int depth, uint32_t row_width, const zkhip_params* prm) {
    int log_n;
    if (p2chip::paths_shape(n_paths, depth, row_width, &log_n) != ZKHIP_OK) return 0;
    const auto p = p2chip::program();
    return zkhip_proof_size_air(p->data(), p->size(), log_n, p2chip::WIDTH, prm, p2chip::N_PUBLIC);
}

int zkhip_prove_merkle_paths(zkhip_ctx* ctx, const uint32_t* leaves, uint32_t row_width, const uint32_t* siblings, const uint32_t* indices, size_t n_paths, int depth,
                             const uint32_t root[8], const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len) {
    CHECK_CTX(ctx);
    if (!root || !prm || !proof || !len) return fail(ZKHIP_ERR_INVALID, "prove_merkle_paths: null argument");
    int log_n;
    ZK_TRY(p2chip::paths_shape(n_paths, depth, row_width, &log_n));
    void* trace;
    ZK_TRY(ctx_reserve(ctx, S_CHIP, ((size_t)p2chip::WIDTH << log_n) * 4, &trace));
    std::vector<uint32_t> roots(n_paths * 8);
    ZK_TRY(zkhip_p2chip_gen_merkle_trace(ctx, leaves, row_width, siblings, indices, n_paths, depth, log_n, (uint32_t*)trace, p2chip::WIDTH, roots.data()));
    for (size_t p = 0; p < n_paths; p++)
        if (std::memcmp(roots.data() + 8 * p, root, 32) != 0) return fail(ZKHIP_ERR_INVALID, "prove_merkle_paths: path " + std::to_string(p) + " does not end in the root");
    uint32_t pv[p2chip::N_PUBLIC];
    std::memcpy(pv, root, 32);
    pv[8] = (uint32_t)n_paths;
    const auto prog = p2chip::program();
    return zkhip_prove_shard_air(ctx, prog->data(), prog->size(), (const uint32_t*)trace, p2chip::WIDTH, log_n, p2chip::WIDTH, pv, p2chip::N_PUBLIC, prm, proof, cap, len);
}

int zkhip_verify_merkle_paths(const uint8_t* proof, size_t len, const uint32_t root[8], size_t n_paths, const zkhip_params* prm, int* reason) {
    if (!proof || !root || !prm || len < 16) return fail(ZKHIP_ERR_INVALID, "verify_merkle_paths: null argument");
    uint32_t head[4];
    std::memcpy(head, proof, 16);
    const int log_n = (int)head[2];                                    // the trace height is read from the proof and bound by its transcript
    if (log_n < 5 || log_n > MAX_LOG_ROWS || n_paths >= P) { if (reason) *reason = 1; return fail(ZKHIP_ERR_VERIFY, "verify_merkle_paths: not a proof of the Poseidon2 chip"); }
    uint32_t pv[p2chip::N_PUBLIC];
    std::memcpy(pv, root, 32);
    pv[8] = (uint32_t)n_paths;
    const auto prog = p2chip::program();
    return zkhip_verify_shard_air(prog->data(), prog->size(), proof, len, log_n, p2chip::WIDTH, pv, p2chip::N_PUBLIC, prm, reason);
}

}  // extern "C"
