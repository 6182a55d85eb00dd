// shard_verifier.inl -- included at the end of fri_chip.hip (it builds on that file's fold chip, SAMPLES chip and trace kernel).
//
// THE SHARD VERIFIER AS A MACHINE (second half).  The reference's one hot call is
// `client.prove(&pk, &stdin, SP1ProofMode::Groth16)` (crates/guest-prover-sp1/src/sp1.rs:116): core -> COMPRESS -> shrink -> wrap, where
// compress is a machine that verifies shard proofs (sp1-recursion, reference Cargo.lock:6172 ff.; RISC Zero: lift -> join behind
// crates/guest-prover-r0/src/prover.rs:90).  The machines above this line check the FRI part of a shard proof and leave the rest to a
// verifier that still reads the inner proof.  This one checks ALL of it in-circuit -- a version-1 shard proof of this library (the synthetic
// AIR, SP1 shape: blowup 2, fold by 2, constant final value, Poseidon2 width 16):
//     the transcript from the header words to the last query index (every challenge a sponge output), the AIR identity at zeta on the
//     opened values, every Merkle opening of trace, quotient and FRI layers, the reduced openings, the folds, the proof of work.
// Every structural fact -- which sponge row absorbs what, which path belongs to which query and tree, which words are constants of the
// shape -- is a PREPROCESSED column.  The key therefore depends on the inner proof's SHAPE (log_n, width, queries, proof-of-work bits,
// number of public values) and on nothing else, and zkhip_verify_shard_recursive is handed the key, the shape and the inner proof's
// PUBLIC VALUES: no byte of the inner proof.
//
// Eight chips (tests/recursion_air.py writes the same programs, tables and traces independently; the words must be equal):
//   P2R      one Poseidon2 permutation per row (p2chip.h, P2RArgs): the transcript's sponge rows, then per query the FRI layer paths, the
//            trace opening (the opened row hashed by sponge rows, then its path) and the quotient opening.  24 preprocessed flag / tag columns.
//   ROWSUM   one row per 8 values of an opened row: hands them to the P2R sponge rows, accumulates sum_j fa^j row[j] (Horner in fa).
//   FOLD     the fold chip (above) in its `rec` form: sends its folded END value to the transcript table, its first rows (index, point, value).
//   TS       the transcript as a table: one row per absorbing sponge row -- the observed words (header words fixed by the key, public values
//            tied to the outer proof's, roots received from the paths' ends), the challenge sampled behind the row (handed to its users).
//   QUERY    one row per query: index (from SAMPLES), point (from FOLD), the reduced opening
//            (at - y_loc)/(x - zeta) + fa^W (at - y_nxt)/(x - zeta g) + fa^2W (aq - y_q)/(x - zeta).
//   OPENED   one row per column group (a, b, c, d) of the synthetic AIR: the opened values at zeta and zeta g (from TS), their fa-weighted
//            sums, the AIR's three constraints per group folded with alpha.
//   SAMPLES  the SAMPLES chip (above): bits of the sampled words -- proof of work, query indices.
//   SCALARS  the verifier's scalars in one row: zeta^N, selectors, powers of fa, the quotient recombination, and the identity
//            fold(zeta) = quotient(zeta) Z_H(zeta).
namespace zk {
namespace rec {
namespace {
using frichip::Builder;
using frichip::ALL; using frichip::FIRST; using frichip::LAST; using frichip::TRANSITION;

// ---- polynomials over columns: a list of (coefficient, variables); an extension expression = four of them (x^4 = 11).  No merging of
// like terms: the order in which terms are produced IS the program (tests/recursion_air.py produces them in the same order).
struct PT { uint32_t c; std::vector<uint32_t> v; };
typedef std::vector<PT> Poly;
typedef std::array<Poly, 4> EE;
inline uint32_t mulp(uint64_t a, uint64_t b) { return (uint32_t)((a % P) * (b % P) % P); }
inline Poly pc(uint64_t c) { c %= P; return c ? Poly{PT{(uint32_t)c, {}}} : Poly{}; }
inline Poly pv(uint32_t col, bool nxt = false) { return Poly{PT{1u, {nxt ? ((1u << 30) | col) : col}}}; }
inline Poly ppub(uint32_t i) { return Poly{PT{1u, {(2u << 30) | i}}}; }
inline Poly padd(const Poly& a, const Poly& b) { Poly o = a; o.insert(o.end(), b.begin(), b.end()); return o; }
inline Poly pscale(const Poly& a, uint64_t k) { Poly o; for (const PT& t : a) { const uint32_t c = mulp(t.c, k); if (c) o.push_back(PT{c, t.v}); } return o; }
inline Poly pneg(const Poly& a) { return pscale(a, P - 1); }
inline Poly pmul(const Poly& a, const Poly& b) {
    Poly o;
    for (const PT& x : a) for (const PT& y : b) { const uint32_t c = mulp(x.c, y.c); if (!c) continue; PT t{c, x.v}; t.v.insert(t.v.end(), y.v.begin(), y.v.end()); o.push_back(t); }
    return o;
}
inline EE ev(uint32_t col, bool nxt = false) { return EE{pv(col, nxt), pv(col + 1, nxt), pv(col + 2, nxt), pv(col + 3, nxt)}; }
inline EE ec(uint64_t c0, uint64_t c1 = 0, uint64_t c2 = 0, uint64_t c3 = 0) { return EE{pc(c0), pc(c1), pc(c2), pc(c3)}; }
inline EE eb(const Poly& p) { return EE{p, Poly{}, Poly{}, Poly{}}; }
inline EE eadd(const EE& a, const EE& b) { return EE{padd(a[0], b[0]), padd(a[1], b[1]), padd(a[2], b[2]), padd(a[3], b[3])}; }
inline EE eadd(const EE& a, const EE& b, const EE& c) { return eadd(eadd(a, b), c); }
inline EE esub(const EE& a, const EE& b) { return EE{padd(a[0], pneg(b[0])), padd(a[1], pneg(b[1])), padd(a[2], pneg(b[2])), padd(a[3], pneg(b[3]))}; }
inline EE escale(const EE& a, uint64_t k) { return EE{pscale(a[0], k), pscale(a[1], k), pscale(a[2], k), pscale(a[3], k)}; }
inline EE emul(const EE& a, const EE& b) {
    EE o;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 4; k++) {
                if ((i + k) % 4 != j) continue;
                Poly t = pmul(a[i], b[k]);
                if (i + k >= 4) t = pscale(t, EXT_W);
                o[j].insert(o[j].end(), t.begin(), t.end());
            }
    return o;
}
inline EE egate(const Poly& f, const EE& e) { return EE{pmul(f, e[0]), pmul(f, e[1]), pmul(f, e[2]), pmul(f, e[3])}; }
struct Cons {
    Builder b;
    void add(uint32_t sel, const Poly& p) {
        frichip::Terms ts;
        for (const PT& t : p) ts.push_back(frichip::Term{t.c, t.v});
        b.add(sel, ts);
    }
    void ext(uint32_t sel, const EE& e) { for (int i = 0; i < 4; i++) add(sel, e[i]); }
    std::vector<uint32_t> program(uint32_t width, uint32_t n_public) const {
        std::vector<uint32_t> p{AIR_MAGIC, 1u, width, b.count, n_public, (uint32_t)(6 + b.body.size())};
        p.insert(p.end(), b.body.begin(), b.body.end());
        return p;
    }
};
inline uint32_t rup4(uint32_t n) { return (n + 3u) & ~3u; }
inline int lg(size_t n, int lo = 5) { int l = lo; while (((size_t)1 << l) < n) l++; return l; }

// ---- buses (BUS_E0 / E1 / R0 / R1 / Q / S0 / S1 / I keep their meaning from the machines above)
constexpr uint32_t BUS_IN0 = 61, BUS_IN1 = 62, BUS_TC = 63, BUS_BETA = 64, BUS_SC = 65, BUS_QI = 66, BUS_AT = 67, BUS_AQ = 68;
// K0 .. K0 + 6: the QUERY chip's seven constants; KO0 .. KO0 + 4: the OPENED chip's five; OY, OY + 1: its two sums (every tuple keyed by the proof's number)
constexpr uint32_t BUS_K0 = 70, BUS_KFA = 77, BUS_KO0 = 78, BUS_OY = 83, BUS_OA = 85;
// the program evaluator (air mode): VAL carries (key, value at zeta) to the EVAL chip's factor slots; EA hands it the proof's alpha
constexpr uint32_t BUS_VAL = 86, BUS_EA = 87;
using frichip::BUS_FIN; using frichip::BUS_E0; using frichip::BUS_E1; using frichip::BUS_R0; using frichip::BUS_R1; using frichip::BUS_Q;
using frichip::BUS_S0; using frichip::BUS_S1; using frichip::BUS_I;

// ---- the shape of an inner proof: everything the machine's structure depends on
struct Shape {
    int n, W, Q, PB, NPUB, R, H, G, WB, NP;          // NP: inner proofs verified by ONE outer proof (the join); every chip holds proof 0's rows, then proof 1's, ...
    int TAGSPAN, TREES;                               // tags / trees of one proof: tags, tree numbers and query numbers carry the proof's number
    // AIR MODE (round 5): the inner proofs are version-7 proofs of a constraint PROGRAM (zkhip_prove_shard_air: the SHA-256 chip ...) instead of
    // version-1 proofs of the synthetic AIR.  The transcript then starts from 18 constant words (6 shape words, logup_pairs / K / F / hash width,
    // the program's 8-word digest: proof_common.h transcript_init), and the fold of the AIR at zeta is the EVAL chip's: one row per TERM.
    bool air = false;
    int HL = 6;                                       // header words the transcript starts from
    struct ETerm { uint32_t coeff; uint32_t key[3]; uint32_t first; };      // factor keys inside one proof's key space (0 = the constant one)
    std::vector<ETerm> terms;                         // the program flattened: constraint by constraint, a selector = one more factor
    std::vector<uint32_t> mult;                       // per key: how many factor slots read it (the senders' preprocessed multiplicities)
    uint32_t KSPAN = 0;                               // keys per proof: 1 + 2 W + NPUB + 3
    std::vector<uint64_t> prog_id;                    // the program's digest (the machine cache's key)
    uint32_t key_local(uint32_t c) const { return 1u + c; }
    uint32_t key_next(uint32_t c) const { return 1u + (uint32_t)W + c; }
    uint32_t key_pub(uint32_t i) const { return 1u + 2u * (uint32_t)W + i; }
    uint32_t key_sel(uint32_t which) const { return 1u + 2u * (uint32_t)W + (uint32_t)NPUB + which; }     // 0 first row, 1 last row, 2 transition
    uint32_t head[18];
    int f0, r0, TA, TQ, TO0, TF, TL0, TP, NS, NT, NTS;
    std::vector<int> pub_rows;
    size_t fri_rows, p2_fri0, p2_tr0, p2_q0, p2_rows;
    int tag0;
    uint32_t ttag(int p, int T) const { return (uint32_t)(p * TAGSPAN + T); }
    uint32_t row_tag(int p, int q, int b) const { return (uint32_t)(p * TAGSPAN + tag0 + q * (WB + 1) + b); }
    uint32_t npub_total() const { return (uint32_t)(NP * NPUB); }
    int absorbed(int T) const {
        if (T < f0 || (TQ <= T && T < TP)) return 8;
        if (T == f0 && r0) return r0;
        if (T == TP) return 5;
        return 0;
    }
    bool has_challenge(int T) const { return T == TA || T == TQ || T == TF || (TL0 <= T && T < TP); }
};
constexpr size_t MAX_JOIN = 1024;       // proofs per join: as many as fit the Poseidon2 chip (zkhip_shard_verifier_max_proofs: 136 at the headline shape under an outer proof at blowup 2, 68 otherwise), at most this
// the Poseidon2 chip's rows are 384 words apart with the key's columns.  An outer proof at blowup 2 extends 2^22-row matrices of any pitch (its
// tile passes run on dense 2^20-row classes: context.cpp, coset_lde_big); other blowups write every 4th row of the LDE and take a pitch of 256 words
constexpr int P2R_MAX_LOG_ROWS = 22;
inline int p2r_max_log_rows(const zkhip_params* outer) { return !outer || outer->log_blowup == 1 ? 22 : 21; }
int make_shape(int log_n, uint32_t width, size_t n_queries, int pow_bits, size_t n_public, size_t n_proofs, Shape& s, const uint32_t* program = nullptr, size_t program_words = 0) {
    // (the inner proofs are zkhip_prove_shard's: its bounds on rows and proof-of-work bits -- proof_common.h, check_shape -- are this machine's)
    if (log_n < 5 || log_n > MAX_LOG_ROWS || width < 8 || width > 1024 || width % 8 || n_queries < 1 || n_queries > 1024 || pow_bits < 0 || pow_bits > 28 ||
        n_public > (program ? 128u : 64u) || n_proofs < 1 || n_proofs > MAX_JOIN)
        return fail(ZKHIP_ERR_INVALID, "shard verifier: 2^5 .. 2^22 rows, a width of 8 .. 1024 in multiples of 8, 1 .. 1024 queries, 0 .. 28 proof-of-work bits, at most 64 public values (128 with a program), 1 .. 1024 proofs");
    s.NP = (int)n_proofs;
    s.n = log_n; s.W = (int)width; s.Q = (int)n_queries; s.PB = pow_bits; s.NPUB = (int)n_public;
    s.R = log_n; s.H = log_n + 1; s.G = s.W / 4; s.WB = s.W / 8;
    const uint32_t head[6] = {(uint32_t)log_n, width, 1u, (uint32_t)n_queries, (uint32_t)pow_bits, (uint32_t)n_public};
    std::memcpy(s.head, head, sizeof head);
    s.air = program != nullptr; s.HL = 6;
    s.terms.clear(); s.mult.clear(); s.prog_id.clear(); s.KSPAN = 0;
    if (program) {
        // (what follows depends on (program, width, public values) alone and costs 4 ms for the SHA-256 chip's program -- its digest is a
        // sponge over 20 000 words --: the last few programs' results are kept)
        struct AirPart { std::vector<uint32_t> prog; uint32_t width, npub, dg[8]; std::vector<Shape::ETerm> terms; std::vector<uint32_t> mult; };
        static std::mutex mu;
        static std::vector<std::shared_ptr<const AirPart>> kept;
        std::shared_ptr<const AirPart> part;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (const auto& k : kept)
                if (k->width == width && k->npub == (uint32_t)n_public && k->prog.size() == program_words && std::memcmp(k->prog.data(), program, program_words * 4) == 0) { part = k; break; }
        }
        s.KSPAN = 1u + 2u * width + (uint32_t)n_public + 3u;
        if (!part) {
            AirView av;
            if (!air_validate(program, program_words, width, n_public, &av) || av.lqd != 1)
                return fail(ZKHIP_ERR_INVALID, "shard verifier: the inner program must be a valid constraint program of this width and public-value count with log_quotient_degree 1");
            auto np = std::make_shared<AirPart>();
            np->prog.assign(program, program + program_words); np->width = width; np->npub = (uint32_t)n_public;
            air_digest(av, np->dg);
            np->mult.assign(s.KSPAN, 0u);
            size_t p = 6;
            for (uint32_t k = 0; k < program[3]; k++) {
                const uint32_t sel = program[p++], nt = program[p++];
                for (uint32_t t = 0; t < nt; t++) {
                    Shape::ETerm e{program[p], {0u, 0u, 0u}, t == 0 ? 1u : 0u};
                    const uint32_t d = program[p + 1];
                    p += 2;
                    uint32_t nf = 0;
                    if (d + (sel ? 1u : 0u) > 3) return fail(ZKHIP_ERR_INVALID, "shard verifier: a term of the inner program has more than three factors");
                    for (uint32_t j = 0; j < d; j++) {
                        const uint32_t v = program[p++], kind = v >> 30, idx = v & 0xFFFFu;
                        e.key[nf++] = kind == 0 ? s.key_local(idx) : (kind == 1 ? s.key_next(idx) : s.key_pub(idx));
                    }
                    if (sel) e.key[nf++] = s.key_sel(sel - 1u);        // program selectors: 1 first row, 2 last row, 3 transition
                    for (int j = 0; j < 3; j++) np->mult[e.key[j]]++;
                    np->terms.push_back(e);
                }
            }
            if (np->terms.empty() || np->terms.size() > ((size_t)1 << 20)) return fail(ZKHIP_ERR_INVALID, "shard verifier: the inner program has no terms, or more than 2^20");
            part = np;
            std::lock_guard<std::mutex> lk(mu);
            if (kept.size() >= 4) kept.erase(kept.begin());
            kept.push_back(part);
        }
        // the header of a version-7 proof of the SP1 shape: + logup_pairs 0, fold by 2^1, constant final value, Poseidon2 width 16, the program's digest
        s.HL = 18;
        s.head[6] = 0u; s.head[7] = 1u; s.head[8] = 0u; s.head[9] = 16u;
        for (int i = 0; i < 8; i++) { s.head[10 + i] = part->dg[i]; s.prog_id.push_back(part->dg[i]); }
        s.terms = part->terms; s.mult = part->mult;
    }
    const int n0 = s.HL + 8 + s.NPUB;
    s.f0 = n0 / 8; s.r0 = n0 % 8;
    s.TA = s.r0 ? s.f0 : s.f0 - 1; s.TQ = s.TA + 1; s.TO0 = s.TQ + 1; s.TF = s.TO0 + s.W + 3; s.TL0 = s.TF + 1; s.TP = s.TL0 + s.R;
    s.NS = (int)frichip::sample_rows(n_queries); s.NT = s.TP + s.NS; s.NTS = s.TP + 1;
    s.pub_rows.clear();
    for (int i = 0; i < s.NPUB; i++) { const int r = (s.HL + 8 + i) / 8; if (s.pub_rows.empty() || s.pub_rows.back() != r) s.pub_rows.push_back(r); }
    s.fri_rows = (size_t)s.R + (size_t)s.R * (size_t)(s.R + 1) / 2;
    s.p2_fri0 = (size_t)s.NT; s.p2_tr0 = s.p2_fri0 + (size_t)s.Q * s.fri_rows; s.p2_q0 = s.p2_tr0 + (size_t)s.Q * (size_t)(s.WB + s.H);
    s.p2_rows = s.p2_q0 + (size_t)s.Q * (size_t)(1 + s.H);
    s.tag0 = s.NT;
    s.TAGSPAN = s.NT + s.Q * (s.WB + 1); s.TREES = s.R + 2;
    if (lg((size_t)s.NP * s.p2_rows) > P2R_MAX_LOG_ROWS) return fail(ZKHIP_ERR_INVALID, "shard verifier: the Poseidon2 chip would need more than 2^22 rows");
    // the transcript table has one preprocessed indicator column per (proof, sponge row that carries public values): at most 1024 preprocessed columns
    if ((s.air ? 46u : 29u) + (size_t)s.NP * s.pub_rows.size() > 1024) return fail(ZKHIP_ERR_INVALID, "shard verifier: too many proofs x public values for the transcript table's preprocessed columns");
    return ZKHIP_OK;
}

int check_outer(const Shape& s, const zkhip_params* outer) {
    if (lg((size_t)s.NP * s.p2_rows) > p2r_max_log_rows(outer))
        return fail(ZKHIP_ERR_INVALID, "shard verifier: a Poseidon2 chip of 2^22 rows takes an outer proof at blowup 2 (log_blowup 1) only");
    return ZKHIP_OK;
}

// ============================================================================================================ P2R
constexpr uint32_t P2_PRE = 24, P2_MAIN = p2chip::R_WIDTH;
constexpr uint32_t PP_SS = 0, PP_SPG = 1, PP_CH = 2, PP_END = 3, PP_K = 4, PP_RIN = 12, PP_TAG = 13, PP_SROOT = 14, PP_TREE = 15, PP_SCH = 16, PP_SSMP = 17, PP_QIDX = 18,
                   PP_QN = 19, PP_RPAIR = 20;
std::vector<uint32_t> p2r_program(const Shape& sh) {
    using namespace p2chip;
    const uint32_t M0 = P2_PRE, IN_ = M0 + IN, OUT = M0 + oute(7), D_ = M0 + D, BIT_ = M0 + BIT, KP_ = M0 + R_KP;
    Cons c;
    c.b.body = permutation_body(M0, &c.b.count);
    for (uint32_t j = 0; j < 8; j++) c.add(ALL, padd(padd(pv(D_ + j), pneg(pv(IN_ + j))), padd(pmul(pv(BIT_), pv(IN_ + j)), pneg(pmul(pv(BIT_), pv(IN_ + 8 + j))))));
    c.add(ALL, padd(pmul(pv(BIT_), pv(BIT_)), pneg(pv(BIT_))));
    c.add(ALL, pmul(padd(pv(PP_SS), pv(PP_SPG)), pv(BIT_)));
    for (uint32_t j = 0; j < 8; j++) c.add(ALL, pmul(pv(PP_SS), pv(IN_ + 8 + j)));
    for (uint32_t j = 0; j < 8; j++) c.add(TRANSITION, pmul(pv(PP_SPG, true), padd(pv(IN_ + 8 + j, true), pneg(pv(OUT + 8 + j)))));
    for (uint32_t j = 0; j < 8; j++) c.add(TRANSITION, pmul(pv(PP_CH, true), padd(pv(D_ + j, true), pneg(pv(OUT + j)))));
    for (uint32_t j = 0; j < 8; j++) c.add(TRANSITION, pmul(pv(PP_K + j, true), padd(pv(IN_ + j, true), pneg(pv(OUT + j)))));
    c.add(TRANSITION, pmul(pv(PP_CH, true), padd(padd(pv(KP_), pscale(pv(KP_, true), P - 2)), pneg(pv(BIT_)))));
    c.add(ALL, pmul(pv(PP_END), padd(pv(KP_), pneg(pv(BIT_)))));
    return c.program(P2_PRE + P2_MAIN, sh.npub_total());
}
// interaction tables: {sign, multiplicity column, bus, n, columns...}
struct Tab {
    std::vector<uint32_t> w{LOOKUP_MAGIC, 0u, 0u};
    void add(uint32_t sign, uint32_t mult, uint32_t bus, std::initializer_list<uint32_t> cols) {
        w.push_back(sign); w.push_back(mult); w.push_back(bus); w.push_back((uint32_t)cols.size());
        for (uint32_t c : cols) w.push_back(c);
        w[1]++; w[2] = (uint32_t)w.size();
    }
    void add8(uint32_t sign, uint32_t mult, uint32_t bus, uint32_t a, uint32_t b) { add(sign, mult, bus, {a, a + 1, a + 2, a + 3, b, b + 1, b + 2, b + 3}); }
    void add4(uint32_t sign, uint32_t mult, uint32_t bus, uint32_t a) { add(sign, mult, bus, {a, a + 1, a + 2, a + 3}); }
    void add5(uint32_t sign, uint32_t mult, uint32_t bus, uint32_t key, uint32_t a) { add(sign, mult, bus, {key, a, a + 1, a + 2, a + 3}); }
};
constexpr uint32_t SEND = 0, RECV = 1;
std::vector<uint32_t> p2r_table() {
    using namespace p2chip;
    const uint32_t M0 = P2_PRE, o = M0 + oute(7), IN_ = M0 + IN, KP_ = M0 + R_KP;
    Tab t;
    t.add5(RECV, PP_RIN, BUS_IN0, PP_TAG, IN_); t.add5(RECV, PP_RIN, BUS_IN1, PP_TAG, IN_ + 4);
    t.add(RECV, PP_RPAIR, BUS_E0, {PP_TREE, KP_, IN_, IN_ + 1, IN_ + 2, IN_ + 3}); t.add(RECV, PP_RPAIR, BUS_E1, {PP_TREE, KP_, IN_ + 4, IN_ + 5, IN_ + 6, IN_ + 7});
    t.add5(SEND, PP_SROOT, BUS_R0, PP_TREE, o); t.add5(SEND, PP_SROOT, BUS_R1, PP_TREE, o + 4);
    t.add(SEND, PP_SCH, BUS_TC, {PP_TAG, o + 7, o + 6, o + 5, o + 4});
    t.add(SEND, PP_SSMP, BUS_S0, {PP_TAG, o + 7, o + 6, o + 5, o + 4}); t.add(SEND, PP_SSMP, BUS_S1, {PP_TAG, o + 3, o + 2, o + 1, o});
    t.add(RECV, PP_QIDX, BUS_QI, {PP_QN, KP_});
    return t.w;
}
// preprocessed traces are built canonical and turned into Montgomery form in one pass (monty_all)
void monty_all(std::vector<uint32_t>& t) { for (uint32_t& v : t) v = to_monty(v); }
void p2r_pre(const Shape& sh, int log_rows, std::vector<uint32_t>& t) {
    t.assign((size_t)P2_PRE << log_rows, 0u);
    for (int p = 0; p < sh.NP; p++) {
    auto row = [&](size_t r) { return t.data() + (size_t)P2_PRE * ((size_t)p * sh.p2_rows + r); };
    for (int T = 0; T < sh.NT; T++) {
        uint32_t* r = row((size_t)T);
        const int k = sh.absorbed(T);
        if (T == 0) r[PP_SS] = 1;
        else { r[PP_SPG] = 1; for (int j = k; j < 8; j++) r[PP_K + j] = 1; }
        r[PP_TAG] = sh.ttag(p, T);
        if (k) r[PP_RIN] = 1;
        if (sh.has_challenge(T)) r[PP_SCH] = 1;
        if (T >= sh.TP) r[PP_SSMP] = 1;
    }
    size_t at = sh.p2_fri0;
    for (int q = 0; q < sh.Q; q++)
        for (int l = 0; l < sh.R; l++) {
            uint32_t* r = row(at++);
            r[PP_SS] = 1; r[PP_RPAIR] = 1; r[PP_TREE] = (uint32_t)(p * sh.TREES + l);
            const int depth = sh.H - (l + 1);
            for (int lvl = 0; lvl < depth; lvl++) {
                r = row(at++);
                r[PP_CH] = 1; r[PP_TREE] = (uint32_t)(p * sh.TREES + l);
                if (lvl == depth - 1) r[PP_END] = r[PP_SROOT] = 1;
            }
        }
    for (int which = 0; which < 2; which++) {
        const int tree = sh.R + which, blocks = which ? 1 : sh.WB;
        for (int q = 0; q < sh.Q; q++) {
            for (int b = 0; b < blocks; b++) {
                uint32_t* r = row(at++);
                r[b == 0 ? PP_SS : PP_SPG] = 1;
                r[PP_RIN] = 1; r[PP_TAG] = sh.row_tag(p, q, which ? sh.WB : b);
            }
            for (int lvl = 0; lvl < sh.H; lvl++) {
                uint32_t* r = row(at++);
                r[PP_CH] = 1; r[PP_TREE] = (uint32_t)(p * sh.TREES + tree);
                if (lvl == 0) { r[PP_QIDX] = 1; r[PP_QN] = (uint32_t)(p * sh.Q + q); }
                if (lvl == sh.H - 1) r[PP_END] = r[PP_SROOT] = 1;
            }
        }
    }
    }
    monty_all(t);
}

// ============================================================================================================ TS
struct TsCols { uint32_t T, ACT, NSEND, CF, CV, IND0, IP, NROOT, NTR, TREE, HASCH, NBETA, NSC, KIND, NFIN, PT, PK, PM, Z, pre, W, TR, CH; };
constexpr uint32_t TS_MAIN = 20;
TsCols ts_cols(const Shape& sh) {
    TsCols c{};
    uint32_t n = 0;
    auto take = [&](uint32_t w) { const uint32_t at = n; n += w; return at; };
    c.T = take(1); c.ACT = take(1); c.NSEND = take(1); c.CF = take(8); c.CV = take(8); c.IND0 = take(1); c.IP = take((uint32_t)(sh.NP * (int)sh.pub_rows.size()));
    c.NROOT = take(1); c.NTR = take(1); c.TREE = take(1); c.HASCH = take(1); c.NBETA = take(1); c.NSC = take(1); c.KIND = take(1); c.NFIN = take(1); c.PT = take(1);
    if (sh.air) { c.PK = take(8); c.PM = take(8); c.Z = take(1); }       // air mode: the public values go to the EVAL chip word by word (key, multiplicity per word; a zero column)
    c.pre = rup4(n);
    c.W = c.pre; c.TR = c.pre + 8; c.CH = c.pre + 16;
    return c;
}
int pub_row_index(const Shape& sh, int row) { for (size_t i = 0; i < sh.pub_rows.size(); i++) if (sh.pub_rows[i] == row) return (int)i; return -1; }
std::vector<uint32_t> ts_program(const Shape& sh) {
    const TsCols c = ts_cols(sh);
    Cons k;
    for (uint32_t j = 0; j < 8; j++) k.add(ALL, pmul(pv(c.CF + j), padd(pv(c.W + j), pneg(pv(c.CV + j)))));
    const int npr = (int)sh.pub_rows.size();
    for (int p = 0; p < sh.NP; p++)              // the outer proof's public values: those of proof 0, then those of proof 1, ...
        for (int i = 0; i < sh.NPUB; i++) {
            const int pos = sh.HL + 8 + i;
            k.add(ALL, pmul(pv(c.IP + (uint32_t)(p * npr + pub_row_index(sh, pos / 8))), padd(pv(c.W + (uint32_t)(pos % 8)), pneg(ppub((uint32_t)(p * sh.NPUB + i))))));
        }
    // the trace root sits behind the header: words HL .. HL + 7 of the transcript = the rest of row HL / 8 and the start of the next one
    const uint32_t o = (uint32_t)(sh.HL % 8);
    for (uint32_t j = 0; j < 8 - o; j++) k.add(ALL, pmul(pv(c.IND0), padd(pv(c.W + o + j), pneg(pv(c.TR + j)))));
    for (uint32_t j = 0; j < o; j++) k.add(TRANSITION, pmul(pv(c.IND0), padd(pv(c.W + j, true), pneg(pv(c.TR + 8 - o + j)))));
    return k.program(c.pre + TS_MAIN, sh.npub_total());
}
std::vector<uint32_t> ts_table(const Shape& sh) {
    const TsCols c = ts_cols(sh);
    Tab t;
    t.add5(SEND, c.NSEND, BUS_IN0, c.T, c.W); t.add5(SEND, c.NSEND, BUS_IN1, c.T, c.W + 4);
    t.add5(RECV, c.HASCH, BUS_TC, c.T, c.CH);
    t.add5(SEND, c.NBETA, BUS_BETA, c.TREE, c.CH);
    t.add5(SEND, c.NSC, BUS_SC, c.KIND, c.CH);
    t.add5(RECV, c.NROOT, BUS_R0, c.TREE, c.W); t.add5(RECV, c.NROOT, BUS_R1, c.TREE, c.W + 4);
    t.add5(RECV, c.NTR, BUS_R0, c.TREE, c.TR); t.add5(RECV, c.NTR, BUS_R1, c.TREE, c.TR + 4);
    t.add5(RECV, c.NFIN, BUS_FIN, c.PT, c.W);
    if (sh.air) for (uint32_t j = 0; j < 8; j++) t.add(SEND, c.PM + j, BUS_VAL, {c.PK + j, c.W + j, c.Z, c.Z, c.Z});
    return t.w;
}
void ts_pre(const Shape& sh, int log_rows, std::vector<uint32_t>& t) {
    const TsCols c = ts_cols(sh);
    t.assign((size_t)c.pre << log_rows, 0u);
    const int npr = (int)sh.pub_rows.size();
    for (int p = 0; p < sh.NP; p++)
    for (int T = 0; T < sh.NTS; T++) {
        uint32_t* r = t.data() + (size_t)c.pre * ((size_t)p * (size_t)sh.NTS + (size_t)T);
        const uint32_t tree0 = (uint32_t)(p * sh.TREES);
        r[c.T] = sh.ttag(p, T); r[c.ACT] = 1; r[c.NSEND] = (sh.TO0 <= T && T <= sh.TF) ? 2u : 1u;
        for (int j = 0; j < 8; j++) if (8 * T + j < sh.HL) { r[c.CF + j] = 1; r[c.CV + j] = sh.head[8 * T + j]; }
        const int pi = pub_row_index(sh, T);
        if (pi >= 0) r[c.IP + (uint32_t)(p * npr + pi)] = 1;
        if (sh.air)
            for (int j = 0; j < 8; j++) {
                const int i = 8 * T + j - (sh.HL + 8);
                if (i >= 0 && i < sh.NPUB) { r[c.PK + j] = (uint32_t)p * sh.KSPAN + sh.key_pub((uint32_t)i); r[c.PM + j] = sh.mult[sh.key_pub((uint32_t)i)]; }
            }
        if (T == sh.HL / 8) { r[c.IND0] = 1; r[c.NTR] = (uint32_t)sh.Q; r[c.TREE] = tree0 + (uint32_t)sh.R; }
        if (T == sh.TQ) { r[c.NROOT] = (uint32_t)sh.Q; r[c.TREE] = tree0 + (uint32_t)sh.R + 1u; }
        if (sh.TL0 <= T && T < sh.TP) { r[c.NROOT] = (uint32_t)sh.Q; r[c.TREE] = tree0 + (uint32_t)(T - sh.TL0); r[c.NBETA] = (uint32_t)sh.Q; }
        const int kinds[3] = {sh.TA, sh.TQ, sh.TF};
        for (int kd = 0; kd < 3; kd++) if (T == kinds[kd]) { r[c.NSC] = 1; r[c.KIND] = (uint32_t)(3 * p + kd); }
        if (sh.has_challenge(T)) r[c.HASCH] = 1;
        if (T == sh.TP) { r[c.NFIN] = (uint32_t)sh.Q; r[c.PT] = tree0; }
    }
    monty_all(t);
}

// ============================================================================================================ ROWSUM
constexpr uint32_t RS_PRE = 12, RP_TAG = 0, RP_ACT = 1, RP_NOTFIRST = 2, RP_LAST0 = 3, RP_LAST1 = 4, RP_QN = 5, RP_FIRST = 6, RP_PID = 7, RP_NFC = 8;
constexpr uint32_t RS_V = 0, RS_ACCIN = 8, RS_T = 12, RS_FA = 44, RS_MAIN = 48;
std::vector<uint32_t> rowsum_program(const Shape& sh) {
    const uint32_t M0 = RS_PRE;
    Cons c;
    const EE fa = ev(M0 + RS_FA);
    c.ext(TRANSITION, egate(pv(RP_NFC, true), esub(ev(M0 + RS_FA, true), fa)));
    EE prev = ev(M0 + RS_ACCIN);
    for (int s = 7; s >= 0; s--) {
        const EE cur = ev(M0 + RS_T + 4u * (uint32_t)s);
        c.ext(ALL, esub(cur, eadd(emul(prev, fa), eb(pv(M0 + RS_V + (uint32_t)s)))));
        prev = cur;
    }
    c.ext(TRANSITION, egate(pv(RP_NOTFIRST, true), esub(ev(M0 + RS_ACCIN, true), ev(M0 + RS_T))));
    c.ext(ALL, egate(padd(pv(RP_ACT), pneg(pv(RP_NOTFIRST))), ev(M0 + RS_ACCIN)));
    return c.program(RS_PRE + RS_MAIN, sh.npub_total());
}
std::vector<uint32_t> rowsum_table() {
    const uint32_t M0 = RS_PRE, v = M0 + RS_V, t0 = M0 + RS_T, fa = M0 + RS_FA;
    Tab t;
    t.add5(SEND, RP_ACT, BUS_IN0, RP_TAG, v); t.add5(SEND, RP_ACT, BUS_IN1, RP_TAG, v + 4);
    t.add5(SEND, RP_LAST0, BUS_AT, RP_QN, t0); t.add5(SEND, RP_LAST1, BUS_AQ, RP_QN, t0);
    t.add5(RECV, RP_FIRST, BUS_KFA, RP_PID, fa);
    return t.w;
}
// (proof, q, block) in trace order: a query's trace blocks from the last to the first, then its quotient block (block number WB)
inline void rowsum_row(const Shape& sh, size_t r, int* p, int* q, int* b) {
    const size_t per = (size_t)sh.WB + 1, per_proof = (size_t)sh.Q * per;
    *p = (int)(r / per_proof);
    r %= per_proof;
    *q = (int)(r / per);
    const int i = (int)(r % per);
    *b = i < sh.WB ? sh.WB - 1 - i : sh.WB;
}
void rowsum_pre(const Shape& sh, int log_rows, std::vector<uint32_t>& t) {
    t.assign((size_t)RS_PRE << log_rows, 0u);
    const size_t per_proof = (size_t)sh.Q * (size_t)(sh.WB + 1), used = (size_t)sh.NP * per_proof;
    for (size_t r = 0; r < used; r++) {
        int p, q, b;
        rowsum_row(sh, r, &p, &q, &b);
        uint32_t* row = t.data() + RS_PRE * r;
        row[RP_TAG] = sh.row_tag(p, q, b); row[RP_ACT] = 1; row[RP_QN] = (uint32_t)(p * sh.Q + q); row[RP_PID] = (uint32_t)p;
        row[RP_NOTFIRST] = (b == sh.WB - 1 || b == sh.WB) ? 0u : 1u;
        row[RP_LAST0] = b == 0 ? 1u : 0u; row[RP_LAST1] = b == sh.WB ? 1u : 0u;
        const bool first = r % per_proof == 0;
        row[RP_FIRST] = first ? 1u : 0u; row[RP_NFC] = first ? 0u : 1u;
    }
    monty_all(t);
}

// ============================================================================================================ QUERY
constexpr uint32_t Q_PRE = 8, QP_QN = 0, QP_ACT = 1, QP_ACT2 = 2, QP_FIRST = 3, QP_PID = 4, QP_NFC = 5, QP_PT = 6;
struct QCols { uint32_t IDX, XQ, RO, AT, AQ, I1, I2, P1, P2, P2O, P3, P3O, ZETA, ZNX, YL, YN, YQ, OFFN, OFFQ, end; };
constexpr QCols qcols() {
    QCols c{};
    uint32_t n = Q_PRE;
    c.IDX = n++; c.XQ = n++;
    uint32_t* f[] = {&c.RO, &c.AT, &c.AQ, &c.I1, &c.I2, &c.P1, &c.P2, &c.P2O, &c.P3, &c.P3O, &c.ZETA, &c.ZNX, &c.YL, &c.YN, &c.YQ, &c.OFFN, &c.OFFQ};
    for (uint32_t* p : f) { *p = n; n += 4; }
    c.end = n;
    return c;
}
constexpr uint32_t Q_MAIN = ((qcols().end - Q_PRE) + 3u) & ~3u;
std::vector<uint32_t> query_program(const Shape& sh) {
    constexpr QCols m = qcols();
    Cons c;
    const uint32_t consts[7] = {m.ZETA, m.ZNX, m.YL, m.YN, m.YQ, m.OFFN, m.OFFQ};
    for (uint32_t col : consts) c.ext(TRANSITION, egate(pv(QP_NFC, true), esub(ev(col, true), ev(col))));
    const EE x = eb(pscale(pv(m.XQ), GEN));
    const Poly act = pv(QP_ACT);
    c.ext(ALL, egate(act, esub(emul(esub(x, ev(m.ZETA)), ev(m.I1)), ec(1))));
    c.ext(ALL, egate(act, esub(emul(esub(x, ev(m.ZNX)), ev(m.I2)), ec(1))));
    c.ext(ALL, esub(ev(m.P1), emul(esub(ev(m.AT), ev(m.YL)), ev(m.I1))));
    c.ext(ALL, esub(ev(m.P2), emul(esub(ev(m.AT), ev(m.YN)), ev(m.I2))));
    c.ext(ALL, esub(ev(m.P2O), emul(ev(m.OFFN), ev(m.P2))));
    c.ext(ALL, esub(ev(m.P3), emul(esub(ev(m.AQ), ev(m.YQ)), ev(m.I1))));
    c.ext(ALL, esub(ev(m.P3O), emul(ev(m.OFFQ), ev(m.P3))));
    c.ext(ALL, esub(ev(m.RO), eadd(ev(m.P1), ev(m.P2O), ev(m.P3O))));
    return c.program(Q_PRE + Q_MAIN, sh.npub_total());
}
std::vector<uint32_t> query_table() {
    constexpr QCols m = qcols();
    Tab t;
    t.add(RECV, QP_ACT, BUS_I, {QP_QN, m.IDX});
    t.add(RECV, QP_ACT, BUS_Q, {QP_PT, m.IDX, m.XQ, m.RO, m.RO + 1, m.RO + 2, m.RO + 3});
    t.add5(RECV, QP_ACT, BUS_AT, QP_QN, m.AT); t.add5(RECV, QP_ACT, BUS_AQ, QP_QN, m.AQ);
    t.add(SEND, QP_ACT2, BUS_QI, {QP_QN, m.IDX});
    const uint32_t consts[7] = {m.ZETA, m.ZNX, m.YL, m.YN, m.YQ, m.OFFN, m.OFFQ};
    for (uint32_t i = 0; i < 7; i++) t.add5(RECV, QP_FIRST, BUS_K0 + i, QP_PID, consts[i]);
    return t.w;
}
void query_pre(const Shape& sh, int log_rows, std::vector<uint32_t>& t) {
    t.assign((size_t)Q_PRE << log_rows, 0u);
    for (int p = 0; p < sh.NP; p++)
        for (int q = 0; q < sh.Q; q++) {
            uint32_t* r = t.data() + Q_PRE * ((size_t)p * (size_t)sh.Q + (size_t)q);
            r[QP_QN] = (uint32_t)(p * sh.Q + q); r[QP_ACT] = 1; r[QP_ACT2] = 2; r[QP_FIRST] = q == 0 ? 1u : 0u; r[QP_PID] = (uint32_t)p; r[QP_NFC] = q == 0 ? 0u : 1u;
            r[QP_PT] = (uint32_t)(p * sh.TREES);
        }
    monty_all(t);
}

// ============================================================================================================ OPENED
constexpr uint32_t OP_PRE = 12, OP_ACT = 0, OP_FIRST = 1, OP_LASTG = 2, OP_NOTFIRST = 3, OP_K1 = 4, OP_K2 = 5, OP_K3 = 6, OP_TL0 = 7, OP_TL1 = 8, OP_TN0 = 9, OP_TN1 = 10, OP_PID = 11;
enum OpCol : uint32_t { O_A, O_B, O_C, O_D, O_AN, O_BN, O_CN, O_DN, O_FA, O_FA4, O_ALPHA, O_SELT, O_SELF, O_PW, O_PWN, O_H2, O_H1, O_IL, O_G2, O_G1, O_INX,
                        O_YLIN, O_YLO, O_YNIN, O_YNO, O_A2, O_AB, O_ACCIN, O_U1, O_U2, O_ACCO, O_COUNT };
constexpr uint32_t oc(uint32_t name) { return OP_PRE + 4u * name; }
constexpr uint32_t OP_MAIN = 4u * O_COUNT;
// air mode: eight (key, multiplicity) pairs more -- the row's eight opened values go to the EVAL chip's factor slots
constexpr uint32_t OP_PRE_AIR = 28, OP_KEY0 = 12, OP_MUL0 = 20;
inline uint32_t op_pre(const Shape& sh) { return sh.air ? OP_PRE_AIR : OP_PRE; }
std::vector<uint32_t> opened_program(const Shape& sh) {
    Cons c;
    const uint32_t B0 = op_pre(sh);
    auto e = [&](uint32_t name, bool nxt = false) { return ev(B0 + 4u * name, nxt); };
    const uint32_t consts[5] = {O_FA, O_FA4, O_ALPHA, O_SELT, O_SELF};
    const Poly first = pv(OP_FIRST), nf = pv(OP_NOTFIRST, true);
    for (uint32_t nm : consts) c.ext(TRANSITION, egate(nf, esub(e(nm, true), e(nm))));
    c.ext(ALL, egate(first, esub(e(O_PW), ec(1))));
    c.ext(ALL, esub(e(O_PWN), emul(e(O_PW), e(O_FA4))));
    c.ext(TRANSITION, egate(nf, esub(e(O_PW, true), e(O_PWN))));
    const EE fa = e(O_FA);
    const uint32_t sets[2][7] = {{O_H2, O_H1, O_IL, O_A, O_B, O_C, O_D}, {O_G2, O_G1, O_INX, O_AN, O_BN, O_CN, O_DN}};
    for (const auto& s : sets) {
        c.ext(ALL, esub(e(s[0]), eadd(e(s[5]), emul(fa, e(s[6])))));
        c.ext(ALL, esub(e(s[1]), eadd(e(s[4]), emul(fa, e(s[0])))));
        c.ext(ALL, esub(e(s[2]), eadd(e(s[3]), emul(fa, e(s[1])))));
    }
    const uint32_t ys[2][3] = {{O_YLIN, O_YLO, O_IL}, {O_YNIN, O_YNO, O_INX}};
    for (const auto& y : ys) {
        c.ext(ALL, egate(first, e(y[0])));
        c.ext(ALL, esub(e(y[1]), eadd(e(y[0]), emul(e(O_PW), e(y[2])))));
        c.ext(TRANSITION, egate(nf, esub(e(y[0], true), e(y[1]))));
    }
    if (sh.air) return c.program(B0 + OP_MAIN, sh.npub_total());       // (the AIR's fold is the EVAL chip's: the columns behind O_YNO stay zero)
    c.ext(ALL, esub(e(O_A2), emul(e(O_A), e(O_A))));
    c.ext(ALL, esub(e(O_AB), emul(e(O_A), e(O_B))));
    const EE al = e(O_ALPHA);
    c.ext(ALL, egate(first, e(O_ACCIN)));
    c.ext(ALL, esub(e(O_U1), eadd(emul(e(O_ACCIN), al), esub(esub(e(O_C), emul(e(O_A2), e(O_B))), eb(pv(OP_K1))))));
    c.ext(ALL, esub(e(O_U2), eadd(emul(e(O_U1), al), emul(e(O_SELT), esub(esub(esub(e(O_DN), e(O_AB)), e(O_C)), eb(pv(OP_K2)))))));
    c.ext(ALL, esub(e(O_ACCO), eadd(emul(e(O_U2), al), emul(e(O_SELF), esub(e(O_D), eb(pv(OP_K3)))))));
    c.ext(TRANSITION, egate(nf, esub(e(O_ACCIN, true), e(O_ACCO))));
    return c.program(OP_PRE + OP_MAIN, sh.npub_total());
}
std::vector<uint32_t> opened_table(const Shape& sh) {
    Tab t;
    const uint32_t B0 = op_pre(sh);
    auto oc_ = [&](uint32_t name) { return B0 + 4u * name; };
    const uint32_t tags[4] = {OP_TL0, OP_TL1, OP_TN0, OP_TN1}, lo[4] = {O_A, O_C, O_AN, O_CN}, hi[4] = {O_B, O_D, O_BN, O_DN};
    for (int i = 0; i < 4; i++) { t.add5(RECV, OP_ACT, BUS_IN0, tags[i], oc_(lo[i])); t.add5(RECV, OP_ACT, BUS_IN1, tags[i], oc_(hi[i])); }
    t.add5(SEND, OP_LASTG, BUS_OY, OP_PID, oc_(O_YLO)); t.add5(SEND, OP_LASTG, BUS_OY + 1, OP_PID, oc_(O_YNO));
    if (!sh.air) t.add5(SEND, OP_LASTG, BUS_OA, OP_PID, oc_(O_ACCO));
    const uint32_t consts[5] = {O_FA, O_FA4, O_ALPHA, O_SELT, O_SELF};
    for (uint32_t i = 0; i < 5; i++) t.add5(RECV, OP_FIRST, BUS_KO0 + i, OP_PID, oc_(consts[i]));
    if (sh.air) for (uint32_t i = 0; i < 8; i++) t.add5(SEND, OP_MUL0 + i, BUS_VAL, OP_KEY0 + i, oc_(O_A + i));      // O_A .. O_D at zeta, O_AN .. O_DN at zeta g
    return t.w;
}
void opened_pre(const Shape& sh, int log_rows, std::vector<uint32_t>& t) {
    const uint32_t PW = op_pre(sh);
    t.assign((size_t)PW << log_rows, 0u);
    for (int p = 0; p < sh.NP; p++)
        for (int g = 0; g < sh.G; g++) {
            uint32_t* r = t.data() + PW * ((size_t)p * (size_t)sh.G + (size_t)g);
            if (sh.air)
                for (uint32_t i = 0; i < 4; i++) {
                    const uint32_t kl = sh.key_local(4u * (uint32_t)g + i), kn = sh.key_next(4u * (uint32_t)g + i);
                    r[OP_KEY0 + i] = (uint32_t)p * sh.KSPAN + kl; r[OP_MUL0 + i] = sh.mult[kl];
                    r[OP_KEY0 + 4 + i] = (uint32_t)p * sh.KSPAN + kn; r[OP_MUL0 + 4 + i] = sh.mult[kn];
                }
            r[OP_ACT] = 1; r[OP_NOTFIRST] = g ? 1u : 0u; r[OP_PID] = (uint32_t)p;
            r[OP_K1] = (uint32_t)g + 1u; r[OP_K2] = 2u * (uint32_t)g + 3u; r[OP_K3] = 5u * (uint32_t)g + 7u;
            r[OP_TL0] = sh.ttag(p, sh.TO0 + 2 * g); r[OP_TL1] = sh.ttag(p, sh.TO0 + 2 * g + 1);
            r[OP_TN0] = sh.ttag(p, sh.TO0 + sh.W / 2 + 2 * g); r[OP_TN1] = sh.ttag(p, sh.TO0 + sh.W / 2 + 2 * g + 1);
            if (g == 0) r[OP_FIRST] = 1;
            if (g == sh.G - 1) r[OP_LASTG] = 1;
        }
    monty_all(t);
}

// ============================================================================================================ SCALARS
constexpr uint32_t SC_PRE = 12, SP_FIRST = 0, SP_KA = 1, SP_KZ = 2, SP_KF = 3, SP_TQZ = 4, SP_PID = 8;
// air mode: the three selectors and the constant one go to the EVAL chip (keys, multiplicities, a zero column)
constexpr uint32_t SC_PRE_AIR = 20, SP_KSEL = 9, SP_KONE = 12, SP_MSEL = 13, SP_MONE = 16, SP_Z = 17;
inline uint32_t sc_pre(const Shape& sh) { return sh.air ? SC_PRE_AIR : SC_PRE; }
struct ScCols {
    uint32_t ALPHA, ZETA, FA, ZP1, INVF, SELF, SELT, ZNX, FP1, PR1, OFFN, OFFQ, QZ0, HQ0, QK0, QK1, QUO, YL, YN, ACC, INVT, SELL, end;
    int mb, nbits; int bits[12];
    uint32_t zp(int i) const { return ZP1 + 4u * (uint32_t)(i - 1); }       // ZP_i = zeta^(2^i), i = 1 .. n
    uint32_t fp(int i) const { return i == 0 ? FA : FP1 + 4u * (uint32_t)(i - 1); }
    uint32_t pr(int k) const { return PR1 + 4u * (uint32_t)(k - 1); }
    uint32_t qz(int j) const { return QZ0 + 4u * (uint32_t)j; }
    uint32_t hq(int j) const { return HQ0 + 4u * (uint32_t)j; }
};
ScCols sc_cols(const Shape& sh) {
    ScCols c{};
    uint32_t n = sc_pre(sh);
    auto take = [&](uint32_t k = 1) { const uint32_t at = n; n += 4 * k; return at; };
    c.ALPHA = take(); c.ZETA = take(); c.FA = take();
    c.ZP1 = take((uint32_t)sh.n);
    c.INVF = take(); c.SELF = take(); c.SELT = take(); c.ZNX = take();
    c.mb = 31 - __builtin_clz((unsigned)sh.W);
    c.FP1 = take((uint32_t)c.mb);
    c.nbits = 0;
    for (int i = 0; i <= c.mb; i++) if ((sh.W >> i) & 1) c.bits[c.nbits++] = i;
    c.PR1 = take((uint32_t)(c.nbits - 1));
    c.OFFN = take(); c.OFFQ = take();
    c.QZ0 = take(8); c.HQ0 = take(7);
    c.QK0 = take(); c.QK1 = take(); c.QUO = take(); c.YL = take(); c.YN = take(); c.ACC = take();
    if (sh.air) { c.INVT = take(); c.SELL = take(); }      // the last-row selector Z_H(zeta) / (zeta - w^-1): a program may use it
    c.end = n;
    return c;
}
// zps_k(zeta) = a_k zeta^N + b_k for the two quotient chunks (canonical)
void zps_consts(const Shape& sh, uint32_t a[2], uint32_t b[2]) {
    const uint64_t N = (uint64_t)1 << sh.n;
    const uint32_t wq = two_adic_generator(sh.n + 1);
    uint32_t sN[2];
    for (int k = 0; k < 2; k++) sN[k] = fpow(fmul(MONTY_GEN, fpow(wq, (uint64_t)k)), N);
    for (int k = 0; k < 2; k++) {
        const int j = 1 - k;
        const uint32_t sj_inv = finv(sN[j]);
        const uint32_t den_inv = finv(fsub(fmul(sN[k], sj_inv), MONTY_R1));
        a[k] = from_monty(fmul(sj_inv, den_inv));
        b[k] = from_monty(fneg(den_inv));
    }
}
std::vector<uint32_t> scalars_program(const Shape& sh) {
    const ScCols m = sc_cols(sh);
    Cons c;
    EE prev = ev(m.ZETA);
    for (int i = 1; i <= sh.n; i++) { c.ext(ALL, esub(ev(m.zp(i)), emul(prev, prev))); prev = ev(m.zp(i)); }
    const EE znn = prev;
    const uint32_t wn = from_monty(two_adic_generator(sh.n)), wni = from_monty(finv(two_adic_generator(sh.n)));
    c.ext(ALL, esub(emul(esub(ev(m.ZETA), ec(1)), ev(m.INVF)), ec(1)));
    c.ext(ALL, esub(ev(m.SELF), emul(esub(znn, ec(1)), ev(m.INVF))));
    c.ext(ALL, esub(ev(m.SELT), esub(ev(m.ZETA), ec(wni))));
    c.ext(ALL, esub(ev(m.ZNX), escale(ev(m.ZETA), wn)));
    prev = ev(m.FA);
    for (int i = 1; i <= m.mb; i++) { c.ext(ALL, esub(ev(m.fp(i)), emul(prev, prev))); prev = ev(m.fp(i)); }
    EE acc = ev(m.fp(m.bits[0]));
    for (int k = 1; k < m.nbits; k++) { c.ext(ALL, esub(ev(m.pr(k)), emul(acc, ev(m.fp(m.bits[k]))))); acc = ev(m.pr(k)); }
    c.ext(ALL, esub(ev(m.OFFN), acc));
    c.ext(ALL, esub(ev(m.OFFQ), emul(ev(m.OFFN), ev(m.OFFN))));
    prev = ev(m.qz(7));
    for (int j = 6; j >= 0; j--) { c.ext(ALL, esub(ev(m.hq(j)), eadd(ev(m.qz(j)), emul(ev(m.FA), prev)))); prev = ev(m.hq(j)); }
    for (int k = 0; k < 2; k++) {
        EE q = ec(0);
        for (int t = 0; t < 4; t++) {
            uint64_t basis[4] = {0, 0, 0, 0};
            basis[t] = 1;
            q = eadd(q, emul(ec(basis[0], basis[1], basis[2], basis[3]), ev(m.qz(4 * k + t))));
        }
        c.ext(ALL, esub(ev(k ? m.QK1 : m.QK0), q));
    }
    uint32_t za[2], zb[2];
    zps_consts(sh, za, zb);
    const EE z0 = eadd(escale(znn, za[0]), ec(zb[0])), z1 = eadd(escale(znn, za[1]), ec(zb[1]));
    c.ext(ALL, esub(ev(m.QUO), eadd(emul(z0, ev(m.QK0)), emul(z1, ev(m.QK1)))));
    c.ext(ALL, esub(ev(m.ACC), emul(ev(m.QUO), esub(znn, ec(1)))));
    if (sh.air) {
        c.ext(ALL, esub(emul(ev(m.SELT), ev(m.INVT)), ec(1)));
        c.ext(ALL, esub(ev(m.SELL), emul(esub(znn, ec(1)), ev(m.INVT))));
    }
    return c.program(sc_pre(sh) + rup4(m.end - sc_pre(sh)), sh.npub_total());
}
std::vector<uint32_t> scalars_table(const Shape& sh) {
    const ScCols m = sc_cols(sh);
    Tab t;
    t.add5(RECV, SP_FIRST, BUS_SC, SP_KA, m.ALPHA); t.add5(RECV, SP_FIRST, BUS_SC, SP_KZ, m.ZETA); t.add5(RECV, SP_FIRST, BUS_SC, SP_KF, m.FA);
    for (int i = 0; i < 4; i++) { t.add5(RECV, SP_FIRST, BUS_IN0, SP_TQZ + (uint32_t)i, m.qz(2 * i)); t.add5(RECV, SP_FIRST, BUS_IN1, SP_TQZ + (uint32_t)i, m.qz(2 * i + 1)); }
    t.add5(RECV, SP_FIRST, BUS_OY, SP_PID, m.YL); t.add5(RECV, SP_FIRST, BUS_OY + 1, SP_PID, m.YN); t.add5(RECV, SP_FIRST, BUS_OA, SP_PID, m.ACC);
    const uint32_t qk[7] = {m.ZETA, m.ZNX, m.YL, m.YN, m.hq(0), m.OFFN, m.OFFQ};
    for (uint32_t i = 0; i < 7; i++) t.add5(SEND, SP_FIRST, BUS_K0 + i, SP_PID, qk[i]);
    t.add5(SEND, SP_FIRST, BUS_KFA, SP_PID, m.FA);
    const uint32_t ok[5] = {m.FA, m.fp(2), m.ALPHA, m.SELT, m.SELF};
    for (uint32_t i = 0; i < 5; i++) t.add5(SEND, SP_FIRST, BUS_KO0 + i, SP_PID, ok[i]);
    if (sh.air) {
        const uint32_t sels[3] = {m.SELF, m.SELL, m.SELT};
        for (uint32_t i = 0; i < 3; i++) t.add5(SEND, SP_MSEL + i, BUS_VAL, SP_KSEL + i, sels[i]);
        t.add(SEND, SP_MONE, BUS_VAL, {SP_KONE, SP_FIRST, SP_Z, SP_Z, SP_Z});          // (key of one, (1, 0, 0, 0)): SP_FIRST is 1 on the proof's row
        t.add5(SEND, SP_FIRST, BUS_EA, SP_PID, m.ALPHA);
    }
    return t.w;
}
void scalars_pre(const Shape& sh, int log_rows, std::vector<uint32_t>& t) {
    const uint32_t PW = sc_pre(sh);
    t.assign((size_t)PW << log_rows, 0u);
    for (int p = 0; p < sh.NP; p++) {
        uint32_t* row = t.data() + PW * (size_t)p;
        if (sh.air) {
            for (uint32_t i = 0; i < 3; i++) { row[SP_KSEL + i] = (uint32_t)p * sh.KSPAN + sh.key_sel(i); row[SP_MSEL + i] = sh.mult[sh.key_sel(i)]; }
            row[SP_KONE] = (uint32_t)p * sh.KSPAN; row[SP_MONE] = sh.mult[0];
        }
        row[SP_FIRST] = 1; row[SP_PID] = (uint32_t)p;
        row[SP_KA] = 3u * (uint32_t)p; row[SP_KZ] = 3u * (uint32_t)p + 1u; row[SP_KF] = 3u * (uint32_t)p + 2u;
        for (int i = 0; i < 4; i++) row[SP_TQZ + i] = sh.ttag(p, sh.TO0 + sh.W + i);
    }
    monty_all(t);
}

// ============================================================================================================ EVAL (air mode)
// One row per TERM of the inner program: coeff x F1 x F2 x F3 with the three factor VALUES received over BUS_VAL by their (preprocessed)
// keys -- an opened value at zeta or zeta g from OPENED, a public value from TS, a selector or the constant one from SCALARS -- and the
// fold of the constraints with alpha as a running sum: where a constraint's first term stands, ACC is multiplied by alpha first.
// Everything that depends on the program is preprocessed: the key of a shape AND a program.
constexpr uint32_t EV_PRE = 12, EP_COEF = 0, EP_K0 = 1, EP_FIRSTC = 4, EP_ACT = 5, EP_LAST = 6, EP_PID = 7, EP_NFC = 8, EP_PFIRST = 9;
constexpr uint32_t EV_F0 = 0, EV_M = 12, EV_TV = 16, EV_ACCIN = 20, EV_ACCO = 24, EV_ALPHA = 28, EV_MAIN = 32;
std::vector<uint32_t> eval_program(const Shape& sh) {
    const uint32_t M0 = EV_PRE;
    Cons c;
    const EE f0 = ev(M0 + EV_F0), f1 = ev(M0 + EV_F0 + 4), f2 = ev(M0 + EV_F0 + 8), mm = ev(M0 + EV_M), tv = ev(M0 + EV_TV), ai = ev(M0 + EV_ACCIN), ao = ev(M0 + EV_ACCO), al = ev(M0 + EV_ALPHA);
    c.ext(ALL, esub(mm, emul(f0, f1)));
    c.ext(ALL, esub(tv, egate(pv(EP_COEF), emul(mm, f2))));
    // ACCO = ACCIN (1 + FIRSTC (alpha - 1)) + TV
    c.ext(ALL, esub(ao, eadd(ai, egate(pv(EP_FIRSTC), esub(emul(ai, al), ai)), tv)));
    c.ext(TRANSITION, egate(pv(EP_NFC, true), esub(ev(M0 + EV_ACCIN, true), ao)));
    c.ext(ALL, egate(pv(EP_PFIRST), ai));
    c.ext(TRANSITION, egate(pv(EP_NFC, true), esub(ev(M0 + EV_ALPHA, true), al)));
    return c.program(EV_PRE + EV_MAIN, sh.npub_total());
}
std::vector<uint32_t> eval_table() {
    const uint32_t M0 = EV_PRE;
    Tab t;
    for (uint32_t j = 0; j < 3; j++) t.add5(RECV, EP_ACT, BUS_VAL, EP_K0 + j, M0 + EV_F0 + 4u * j);
    t.add5(RECV, EP_PFIRST, BUS_EA, EP_PID, M0 + EV_ALPHA);
    t.add5(SEND, EP_LAST, BUS_OA, EP_PID, M0 + EV_ACCO);
    return t.w;
}
void eval_pre(const Shape& sh, int log_rows, std::vector<uint32_t>& t) {
    t.assign((size_t)EV_PRE << log_rows, 0u);
    const size_t nt = sh.terms.size();
    for (int p = 0; p < sh.NP; p++)
        for (size_t i = 0; i < nt; i++) {
            uint32_t* r = t.data() + (size_t)EV_PRE * ((size_t)p * nt + i);
            const Shape::ETerm& e = sh.terms[i];
            r[EP_COEF] = e.coeff;
            for (int j = 0; j < 3; j++) r[EP_K0 + j] = (uint32_t)p * sh.KSPAN + e.key[j];
            r[EP_FIRSTC] = e.first; r[EP_ACT] = 1; r[EP_LAST] = i + 1 == nt ? 1u : 0u; r[EP_PID] = (uint32_t)p;
            r[EP_NFC] = i ? 1u : 0u; r[EP_PFIRST] = i ? 0u : 1u;
        }
    monty_all(t);
}

// ============================================================================================================ the machine
enum Chip : int { C_P2R, C_ROWSUM, C_FOLD, C_TS, C_QUERY, C_OPENED, C_SAMPLES, C_SCALARS, C_EVAL, N_CHIPS };      // (C_EVAL: air mode only)
struct Machine {
    Shape sh;
    int n = N_CHIPS - 1;                        // chips of this machine: eight, nine in air mode
    int order[N_CHIPS];                         // position -> chip, tallest first (equal heights in the order of the enum)
    int32_t log_ns[N_CHIPS]; uint32_t widths[N_CHIPS], pre_widths[N_CHIPS];
    std::vector<uint32_t> prog[N_CHIPS], tab[N_CHIPS];     // by position
    const uint32_t* progs[N_CHIPS]; size_t prog_words[N_CHIPS]; const uint32_t* tabs[N_CHIPS]; size_t tab_words[N_CHIPS];
    int height[N_CHIPS];                        // by chip
    int pos_of(int chip) const { for (int i = 0; i < n; i++) if (order[i] == chip) return i; return -1; }
};
std::vector<uint32_t> samples_table_words() { return frichip::samples_interactions(); }
std::vector<uint32_t> fold_table(const Shape& sh) {
    using namespace frichip;
    Tab t;
    t.add(SEND, ACTIVE, BUS_E0, {LNX, K2, E0, E0 + 1, E0 + 2, E0 + 3}); t.add(SEND, ACTIVE, BUS_E1, {LNX, K2, E1, E1 + 1, E1 + 2, E1 + 3});
    t.add(SEND, L_REC, BUS_Q, {frichip::PT, IDX, XS, OWN, OWN + 1, OWN + 2, OWN + 3});
    t.add5(RECV, ACTIVE, BUS_BETA, LNX, BETA);
    t.add5(SEND, L_REC + (uint32_t)sh.R - 1u, BUS_FIN, frichip::PT, FOLD);
    return t.w;
}
// the machine of a shape: programs, tables, heights.  Built once per (shape, Poseidon2 tables) and kept.
std::shared_ptr<const Machine> machine_of(const Shape& sh) {
    static std::mutex mu;
    static std::map<std::vector<uint64_t>, std::shared_ptr<const Machine>> cache;
    std::vector<uint64_t> key{(uint64_t)sh.n, (uint64_t)sh.W, (uint64_t)sh.Q, (uint64_t)sh.PB, (uint64_t)sh.NPUB, (uint64_t)sh.NP, g_p2_generation.load()};
    key.insert(key.end(), sh.prog_id.begin(), sh.prog_id.end());       // (air mode: the inner program's digest)
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    auto m = std::make_shared<Machine>();
    m->sh = sh;
    const size_t np = (size_t)sh.NP;
    m->n = sh.air ? N_CHIPS : N_CHIPS - 1;
    const int h[N_CHIPS] = {lg(np * sh.p2_rows), lg(np * (size_t)sh.Q * (size_t)(sh.WB + 1)), lg(np * (size_t)sh.Q * (size_t)sh.R), lg(np * (size_t)sh.NTS), lg(np * (size_t)sh.Q),
                            lg(np * (size_t)sh.G), lg(np * (size_t)sh.NS), lg(np), lg(np * (sh.air ? sh.terms.size() : (size_t)1))};
    for (int c = 0; c < N_CHIPS; c++) { m->height[c] = h[c]; m->order[c] = c; }
    std::stable_sort(m->order, m->order + m->n, [&](int a, int b) { return h[a] > h[b]; });
    const ScCols scc = sc_cols(sh);
    const TsCols tsc = ts_cols(sh);
    const uint32_t w_main[N_CHIPS] = {P2_MAIN, RS_MAIN, frichip::width_of(sh.R, true, true), TS_MAIN, Q_MAIN, OP_MAIN, frichip::S_MAIN, rup4(scc.end - sc_pre(sh)), EV_MAIN};
    const uint32_t w_pre[N_CHIPS] = {P2_PRE, RS_PRE, 0u, tsc.pre, Q_PRE, op_pre(sh), frichip::S_PRE, sc_pre(sh), EV_PRE};
    for (int i = 0; i < m->n; i++) {
        const int c = m->order[i];
        switch (c) {
            case C_P2R: m->prog[i] = p2r_program(sh); m->tab[i] = p2r_table(); break;
            case C_ROWSUM: m->prog[i] = rowsum_program(sh); m->tab[i] = rowsum_table(); break;
            case C_FOLD: m->prog[i] = *frichip::program(sh.R, true, true, (int)sh.npub_total()); m->tab[i] = fold_table(sh); break;
            case C_TS: m->prog[i] = ts_program(sh); m->tab[i] = ts_table(sh); break;
            case C_QUERY: m->prog[i] = query_program(sh); m->tab[i] = query_table(); break;
            case C_OPENED: m->prog[i] = opened_program(sh); m->tab[i] = opened_table(sh); break;
            case C_EVAL: m->prog[i] = eval_program(sh); m->tab[i] = eval_table(); break;
            case C_SAMPLES: m->prog[i] = *frichip::samples_program(sh.R, sh.PB, sh.npub_total()); m->tab[i] = samples_table_words(); break;
            default: m->prog[i] = scalars_program(sh); m->tab[i] = scalars_table(sh); break;
        }
        m->log_ns[i] = h[c]; m->widths[i] = w_main[c]; m->pre_widths[i] = w_pre[c];
    }
    for (int i = 0; i < m->n; i++) { m->progs[i] = m->prog[i].data(); m->prog_words[i] = m->prog[i].size(); m->tabs[i] = m->tab[i].data(); m->tab_words[i] = m->tab[i].size(); }
    if (cache.size() > 16) cache.clear();
    cache.emplace(key, m);
    return m;
}

// the SAMPLES chip's fixed columns for several proofs: proof p's rows behind proof p - 1's; its sponge rows are numbered from ITS tags, its queries from p Q
void samples_pre_all(const Shape& sh, int log_rows, std::vector<uint32_t>& t) {
    t.assign((size_t)frichip::S_PRE << log_rows, 0u);
    std::vector<uint32_t> one;
    for (int p = 0; p < sh.NP; p++) {
        frichip::samples_pre(sh.R, (size_t)sh.Q, lg((size_t)sh.NS), one, (int)sh.ttag(p, sh.TP));
        for (int r = 0; r < sh.NS; r++) {
            uint32_t* row = one.data() + (size_t)frichip::S_PRE * (size_t)r;
            for (uint32_t j = 0; j < 8; j++) if (row[frichip::S_ACT + j]) row[frichip::S_KQ + j] = fadd(row[frichip::S_KQ + j], to_monty((uint32_t)(p * sh.Q)));
        }
        std::memcpy(t.data() + (size_t)frichip::S_PRE * (size_t)p * (size_t)sh.NS, one.data(), (size_t)frichip::S_PRE * (size_t)sh.NS * 4);
    }
}
// every chip's preprocessed trace, by chip
void all_pre(const Shape& sh, const Machine& m, std::vector<uint32_t> pre[N_CHIPS]) {
    p2r_pre(sh, m.height[C_P2R], pre[C_P2R]); rowsum_pre(sh, m.height[C_ROWSUM], pre[C_ROWSUM]); ts_pre(sh, m.height[C_TS], pre[C_TS]);
    query_pre(sh, m.height[C_QUERY], pre[C_QUERY]); opened_pre(sh, m.height[C_OPENED], pre[C_OPENED]); scalars_pre(sh, m.height[C_SCALARS], pre[C_SCALARS]);
    samples_pre_all(sh, m.height[C_SAMPLES], pre[C_SAMPLES]);
    if (sh.air) eval_pre(sh, m.height[C_EVAL], pre[C_EVAL]);
}
// ---- the witness: everything the main columns hold, read off the inner proof (which the host verifier has accepted)
struct Witness {
    const uint32_t* w = nullptr;                // the proof's words (canonical)
    size_t o_troot, o_qroot, o_loc, o_nxt, o_qz, o_lroots, o_final, o_wit, o_queries, per_query;
    std::vector<uint32_t> betas, indices, values, siblings, lroots, paths;     // zkhip_fri_view_all
    uint32_t fin[4], tr[10];
    size_t q_trow(int q) const { return o_queries + (size_t)q * per_query; }
};
inline Ext ext_canon(const uint32_t* p) { return Ext{{to_monty(p[0]), to_monty(p[1]), to_monty(p[2]), to_monty(p[3])}}; }
inline void put_ext(uint32_t* row, uint32_t col, const Ext& e) { for (int i = 0; i < 4; i++) row[col + i] = e.c[i]; }

}  // namespace
}  // namespace rec
}  // namespace zk

// ---- main traces + proof.  Host-side tables (TS, ROWSUM, QUERY, OPENED, SAMPLES, SCALARS: a few thousand rows per inner proof) are filled on
// the host in Montgomery form and uploaded; the Poseidon2 chip's rows (2^15 per headline proof) and the fold chip's rows are filled on the device.
namespace zk {
namespace rec {
namespace {
// round 6: where the per-query witnesses of the recursion machines are made.  0 (default): device kernels; 1: the host's walk (fill_one / fill_proof in
// full) -- the fallback for a device that misbehaves, and what the tests compare the kernels with (zkhip_recursion_witnesses_on_host; the A/B build also
// reads ZKHIP_REC_HOST per call, for tools/)
static std::atomic<int> g_witnesses_on_host{0};
inline bool witnesses_on_host() {
#ifdef ZKHIP_AB_HOOKS
    if (const char* e = getenv("ZKHIP_REC_HOST")) return atoi(e) != 0;
#endif
    return g_witnesses_on_host.load() != 0;
}
// what ONE inner proof contributes: the host tables' rows of its segment, and its entries of the device work lists
struct HostSpan {                                           // a slice of the context's pinned upload block
    uint32_t* p = nullptr; size_t n = 0;
    uint32_t* data() const { return p; }
    size_t size() const { return n; }
    uint32_t& operator[](size_t i) const { return p[i]; }
};
// a table of zero words whose pages the kernel hands out on first touch (calloc): the fill threads touch them side by side -- a serial
// memset of the 170 MB a join of 64 SHA-256 proofs fills took 32 ms of its 185
// The machines' host tables are hundreds of megabytes of zeroed words per call, most of them written once.  Fresh from calloc they cost a page fault per
// 4 KB while they are filled and a munmap when they go (55 of the 216 ms the 64 keyed transcript proofs took were spent AFTER the proof was done).  So the
// tables of a finished call are zeroed again on a thread of their own, off the caller's path, and kept for the next call that asks for the same size
// (WordPool: at most POOL_CAP bytes; zkhip_release_cached_contexts empties it); a call that finds nothing takes fresh calloc pages as before.
struct WordPool {
    static constexpr size_t POOL_CAP = (size_t)6 << 30;
    std::mutex mu;
    std::vector<std::pair<uint32_t*, size_t>> idle;          // (zeroed block, words)
    size_t bytes = 0;
    uint32_t* take(size_t words) {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = idle.size(); i-- > 0;)
            if (idle[i].second == words) { uint32_t* q = idle[i].first; idle.erase(idle.begin() + (long)i); bytes -= words * 4; return q; }
        return nullptr;
    }
    void give(uint32_t* q, size_t words) {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (bytes + words * 4 <= POOL_CAP) { idle.emplace_back(q, words); bytes += words * 4; return; }
        }
        std::free(q);
    }
    void clear() {
        std::vector<std::pair<uint32_t*, size_t>> all;
        { std::lock_guard<std::mutex> lk(mu); all.swap(idle); bytes = 0; }
        for (auto& e : all) std::free(e.first);
    }
};
inline WordPool& word_pool() { static WordPool* pool = new WordPool(); return *pool; }       // (never destroyed: a recycling thread may outlive main)
struct ZeroedWords {
    uint32_t* p = nullptr; size_t n = 0;
    ZeroedWords() = default;
    ZeroedWords(const ZeroedWords&) = delete;
    ZeroedWords& operator=(const ZeroedWords&) = delete;
    ~ZeroedWords() { std::free(p); }
    bool reset(size_t words) {
        std::free(p);
        const size_t w = words ? words : 1;
        p = w >= ((size_t)1 << 18) ? word_pool().take(w) : nullptr;          // (blocks of a megabyte and more are worth keeping)
        if (!p) p = (uint32_t*)std::calloc(w, 4);
        n = p ? words : 0;
        return p != nullptr;
    }
    uint32_t* data() { return p; }
    const uint32_t* data() const { return p; }
    size_t size() const { return n; }
    std::pair<uint32_t*, size_t> release() { const std::pair<uint32_t*, size_t> r{p, n ? n : 1}; p = nullptr; n = 0; return r; }
};
// the tables of a finished call: zeroed and pooled (the large ones) or freed (the small ones) on a thread of their own
inline void recycle_later(std::vector<std::pair<uint32_t*, size_t>> blocks) {
    auto job = [blocks] {
        for (const auto& b : blocks) {
            if (!b.first) continue;
            if (b.second >= ((size_t)1 << 18)) { std::memset(b.first, 0, b.second * 4); word_pool().give(b.first, b.second); }
            else std::free(b.first);
        }
    };
    try { std::thread(job).detach(); } catch (...) { for (const auto& b : blocks) std::free(b.first); }
}
struct HostTables {
    ZeroedWords sc, op, rs, q, ts, sm, evl;       // SCALARS, OPENED, ROWSUM, QUERY, TS, SAMPLES (and, air mode, EVAL) main traces (all proofs)
    HostSpan desc, data, chain_in, trows;                   // P2R: chains, their data, the transcript rows' input states and row numbers -- consecutive slices of ONE pinned block
                                                            // (every word is written by fill_one: nothing is cleared)
    std::vector<const uint32_t*> want_roots;                // per chain: where it must end (canonical words; owned by the witnesses)
    // (desc / data / chain_in / trows / want_roots hold a fixed slice per proof)
};
// round 6: the transcripts of up to sixteen proofs side by side, one AVX-512 permutation per sponge row of the group (p2_x16.h) -- a transcript is a serial chain, so
// the vector unit is used ACROSS the proofs.  What a chain leaves behind per proof:
struct SvTranscript { std::vector<uint32_t> chain_in, samples; std::vector<Ext> chal; bool done = false; };
inline void sv_blocks(const Shape& sh, const uint32_t* pw, const uint32_t* public_values, std::vector<std::array<uint32_t, 8>>& blocks) {
    // observed words per absorbing row (canonical; absent ones zero): fill_one's section (b)
    const int W = sh.W, R = sh.R;
    const size_t n_public = (size_t)sh.NPUB;
    const size_t o_troot = sh.air ? 20 : 8, o_qroot = o_troot + 8, o_loc = o_qroot + 8, o_qz = o_loc + 8 * (size_t)W, o_lroots = o_qz + 32, o_final = o_lroots + 8 * (size_t)R, o_wit = o_final + 4;
    blocks.assign((size_t)sh.NTS, std::array<uint32_t, 8>{});
    std::vector<uint32_t> seq0(sh.head, sh.head + sh.HL);
    seq0.insert(seq0.end(), pw + o_troot, pw + o_troot + 8);
    for (size_t i = 0; i < n_public; i++) seq0.push_back(public_values[i] % P);
    for (int T = 0; T < sh.f0 + (sh.r0 ? 1 : 0); T++) for (int j = 0; j < 8 && 8 * (size_t)T + j < seq0.size(); j++) blocks[(size_t)T][j] = seq0[8 * (size_t)T + j];
    for (int j = 0; j < 8; j++) blocks[(size_t)sh.TQ][j] = pw[o_qroot + j];
    for (int i = 0; i < W + 4; i++) for (int j = 0; j < 8; j++) blocks[(size_t)(sh.TO0 + i)][j] = pw[o_loc + 8 * (size_t)i + j];
    for (int l = 0; l < R; l++) for (int j = 0; j < 8; j++) blocks[(size_t)(sh.TL0 + l)][j] = pw[o_lroots + 8 * (size_t)l + j];
    for (int j = 0; j < 4; j++) blocks[(size_t)sh.TP][j] = pw[o_final + j];
    blocks[(size_t)sh.TP][4] = pw[o_wit];
}
void sv_walk_transcripts_x16(const Shape& sh, int n, const uint32_t* const* pws, const uint32_t* const* pubss, SvTranscript* const* out) {
    std::vector<std::array<uint32_t, 8>> blocks[16];
    for (int i = 0; i < n; i++) {
        sv_blocks(sh, pws[i], pubss[i], blocks[i]);
        out[i]->chain_in.assign(16 * (size_t)sh.NT, 0u); out[i]->samples.assign(8 * (size_t)sh.NS, 0u); out[i]->chal.assign((size_t)sh.NT, ext_zero());
    }
    uint32_t st[16][16];
    for (int e = 0; e < 16; e++) for (int j = 0; j < 16; j++) st[e][j] = 0u;
    for (int T = 0; T < sh.NT; T++) {
        const int k = sh.absorbed(T);
        for (int e = 0; e < k; e++) for (int i = 0; i < 16; i++) st[e][i] = to_monty(blocks[i < n ? i : 0][(size_t)T][(size_t)e]);
        for (int i = 0; i < n; i++) for (int e = 0; e < 16; e++) out[i]->chain_in[16 * (size_t)T + (size_t)e] = from_monty(st[e][i]);
        p2x16_permute(st);
        for (int i = 0; i < n; i++) {
            out[i]->chal[(size_t)T] = Ext{{st[7][i], st[6][i], st[5][i], st[4][i]}};
            if (T >= sh.TP) for (int j = 0; j < 8; j++) out[i]->samples[8 * (size_t)(T - sh.TP) + (size_t)j] = from_monty(st[7 - j][i]);
        }
    }
    for (int i = 0; i < n; i++) out[i]->done = true;
}
// round 6 (device mode): what the device's per-query kernels read of ONE proof -- [indices Q][betas 4 R canonical][fa][zeta][znx][yl][yn][yq][offn][offq] (extension
// elements in Montgomery form)
inline size_t sv_vals_words(const Shape& sh) { return (size_t)sh.Q + 4 * (size_t)sh.R + 32; }
int fill_one(const Shape& sh, const Machine& m, int p, const uint8_t* inner, size_t inner_len, const uint32_t* public_values, const zkhip_params* inner_prm, Witness& wt,
             std::vector<uint32_t>& words, HostTables& ht, Ext* fa_out, const uint32_t* program, size_t program_words, uint32_t* dev_vals = nullptr, const SvTranscript* pre = nullptr) {
    const int R = sh.R, H = sh.H, Q = sh.Q, W = sh.W;
    const int log_n = sh.n;
    const uint32_t width = (uint32_t)W;
    const size_t n_public = (size_t)sh.NPUB;
    const bool on_device = dev_vals != nullptr;        // ROWSUM, QUERY, the fold chains and the layers' pairs are the device's (shard_verifier_prove_impl)
#ifdef ZKHIP_AB_HOOKS
    static const bool sec_timing = getenv("ZKHIP_REC_SECTIONS") != nullptr;
    auto sec_t = std::chrono::steady_clock::now();
    auto sec = [&](const char* what) {
        if (!sec_timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "      [fill_one %d] %-28s %8.1f us\n", p, what, std::chrono::duration<double, std::micro>(now - sec_t).count());
        sec_t = now;
    };
#else
    auto sec = [](const char*) {};
#endif
    wt.lroots.resize(8 * (size_t)R);
    if (!on_device) {
        // (a) one host pass: the verifier accepts the proof and hands out the FRI side; the rest is read off the words (docs/PROTOCOL.md section 6)
        wt.betas.resize(4 * (size_t)R); wt.indices.resize((size_t)Q); wt.values.resize(4 * (size_t)Q); wt.siblings.resize(4 * (size_t)Q * (size_t)R);
        wt.paths.resize(zkhip_fri_view_path_words(R) * (size_t)Q);
        // (the Merkle paths are not hashed here: the P2R rows kernel hashes every one of them for the trace, and the roots it arrives at are compared below)
        ZK_TRY(fri_view_all_unhashed(inner, inner_len, log_n, width, public_values, n_public, inner_prm, wt.betas.data(), wt.fin, wt.indices.data(), wt.values.data(),
                                     wt.siblings.data(), wt.lroots.data(), wt.paths.data(), wt.tr, program, program_words));
    } else {
        // device mode has no separate verifier pass: filling the tables IS the verification (as in machine mode).  What that pass checked of the words
        // themselves: the header names this shape, every word is a canonical residue
        const uint32_t* pf = (const uint32_t*)inner;
        auto bad = [&](const char* what) { return fail(ZKHIP_ERR_VERIFY, std::string("prove_shard_verifier: proof ") + std::to_string(p) + " rejected: " + what); };
        if (inner_len < 32 || pf[0] != 0x41544B5Au || pf[1] != (sh.air ? 7u : 1u)) return bad("not a shard proof of this kind");
        for (int i = 0; i < sh.HL; i++) if (pf[2 + i] != sh.head[i]) return bad("another shape's header");
        for (size_t i = (size_t)sh.HL + 2; i < inner_len / 4; i++) if (pf[i] >= P) return bad("a non-canonical word");
        for (size_t i = 0; i < n_public; i++) if (public_values[i] >= P) return bad("a non-canonical public value");
    }
    words.resize(inner_len / 4);
    std::memcpy(words.data(), inner, words.size() * 4);
    wt.w = words.data();
    wt.o_troot = sh.air ? 20 : 8;                       // a version-7 header: 12 shape words and the program's digest in front of the trace root
    wt.o_qroot = wt.o_troot + 8; wt.o_loc = wt.o_qroot + 8; wt.o_nxt = wt.o_loc + 4 * (size_t)W; wt.o_qz = wt.o_nxt + 4 * (size_t)W; wt.o_lroots = wt.o_qz + 32;
    wt.o_final = wt.o_lroots + 8 * (size_t)R; wt.o_wit = wt.o_final + 4; wt.o_queries = wt.o_wit + 1;
    wt.per_query = (size_t)W + 8 * (size_t)H + 8 + 8 * (size_t)H;
    for (int l = 0; l < R; l++) wt.per_query += 4 + 8 * (size_t)(H - 1 - l);
    if (words.size() != wt.o_queries + (size_t)Q * wt.per_query) return fail(ZKHIP_ERR_INTERNAL, "prove_shard_verifier: unexpected proof length");
    const uint32_t* pw = wt.w;
    sec("checks + words");
    // (b) the transcript's sponge chain on the host (NT permutations): the input state of every row, the challenges, the sampled words
    std::vector<uint32_t> chain_in(16 * (size_t)sh.NT), samples(8 * (size_t)sh.NS);
    std::vector<Ext> chal((size_t)sh.NT);
    if (pre && pre->done) {                                                // (walked beside other proofs' transcripts: sv_walk_transcripts_x16)
        chain_in = pre->chain_in; samples = pre->samples; chal = pre->chal;
        if (on_device) {
            wt.betas.resize(4 * (size_t)R);
            for (int l = 0; l < R; l++) for (int j = 0; j < 4; j++) wt.betas[4 * (size_t)l + j] = from_monty(chal[(size_t)(sh.TL0 + l)].c[j]);
            std::memcpy(wt.lroots.data(), pw + wt.o_lroots, 32 * (size_t)R);
        } else
        for (int l = 0; l < R; l++) for (int j = 0; j < 4; j++)
            if (from_monty(chal[(size_t)(sh.TL0 + l)].c[j]) != wt.betas[4 * (size_t)l + j]) return fail(ZKHIP_ERR_INTERNAL, "prove_shard_verifier: the sponge rows do not reproduce the verifier's challenges");
    } else {
        std::vector<std::array<uint32_t, 8>> blocks((size_t)sh.NTS);      // observed words per absorbing row (canonical; absent ones zero)
        std::vector<uint32_t> seq0(sh.head, sh.head + sh.HL);
        seq0.insert(seq0.end(), pw + wt.o_troot, pw + wt.o_troot + 8);
        for (size_t i = 0; i < n_public; i++) seq0.push_back(public_values[i] % P);
        for (auto& b : blocks) b.fill(0u);
        for (int T = 0; T < sh.f0 + (sh.r0 ? 1 : 0); T++) for (int j = 0; j < 8 && 8 * (size_t)T + j < seq0.size(); j++) blocks[(size_t)T][j] = seq0[8 * (size_t)T + j];
        for (int j = 0; j < 8; j++) blocks[(size_t)sh.TQ][j] = pw[wt.o_qroot + j];
        for (int i = 0; i < W + 4; i++) for (int j = 0; j < 8; j++) blocks[(size_t)(sh.TO0 + i)][j] = pw[wt.o_loc + 8 * (size_t)i + j];
        for (int l = 0; l < R; l++) for (int j = 0; j < 8; j++) blocks[(size_t)(sh.TL0 + l)][j] = pw[wt.o_lroots + 8 * (size_t)l + j];
        for (int j = 0; j < 4; j++) blocks[(size_t)sh.TP][j] = pw[wt.o_final + j];
        blocks[(size_t)sh.TP][4] = pw[wt.o_wit];
        uint32_t st[16] = {0};
        for (int T = 0; T < sh.NT; T++) {
            const int k = sh.absorbed(T);
            for (int j = 0; j < k; j++) st[j] = to_monty(blocks[(size_t)T][j]);
            for (int j = 0; j < 16; j++) chain_in[16 * (size_t)T + j] = from_monty(st[j]);
            p2_permute(st);
            chal[(size_t)T] = Ext{{st[7], st[6], st[5], st[4]}};
            if (T >= sh.TP) for (int j = 0; j < 8; j++) samples[8 * (size_t)(T - sh.TP) + j] = from_monty(st[7 - j]);
        }
        if (on_device) {
            wt.betas.resize(4 * (size_t)R);
            for (int l = 0; l < R; l++) for (int j = 0; j < 4; j++) wt.betas[4 * (size_t)l + j] = from_monty(chal[(size_t)(sh.TL0 + l)].c[j]);
            std::memcpy(wt.lroots.data(), pw + wt.o_lroots, 32 * (size_t)R);
        } else
        for (int l = 0; l < R; l++) for (int j = 0; j < 4; j++)
            if (from_monty(chal[(size_t)(sh.TL0 + l)].c[j]) != wt.betas[4 * (size_t)l + j]) return fail(ZKHIP_ERR_INTERNAL, "prove_shard_verifier: the sponge rows do not reproduce the verifier's challenges");
    }
    const Ext alpha = chal[(size_t)sh.TA], zeta = chal[(size_t)sh.TQ], fa = chal[(size_t)sh.TF];
    *fa_out = fa;
    sec("transcript");
    // (c) the scalars: row p of SCALARS
    const ScCols scc = sc_cols(sh);
    const uint32_t scp = sc_pre(sh), sc_w = rup4(scc.end - scp);
    uint32_t* sc_row = ht.sc.data() + (size_t)sc_w * (size_t)p;
    auto scput = [&](uint32_t col, const Ext& e) { put_ext(sc_row, col - scp, e); };
    Ext znn = zeta;
    scput(scc.ALPHA, alpha); scput(scc.ZETA, zeta); scput(scc.FA, fa);
    for (int i = 1; i <= sh.n; i++) { znn = ext_mul(znn, znn); scput(scc.zp(i), znn); }
    const uint32_t wn = two_adic_generator(sh.n), wni = finv(wn);
    const Ext invf = ext_inv(ext_sub_base(zeta, MONTY_R1)), self_ = ext_mul(ext_sub_base(znn, MONTY_R1), invf), selt = ext_sub_base(zeta, wni), znx = ext_mul_base(zeta, wn);
    scput(scc.INVF, invf); scput(scc.SELF, self_); scput(scc.SELT, selt); scput(scc.ZNX, znx);
    const Ext invt = sh.air ? ext_inv(selt) : ext_zero(), sell = sh.air ? ext_mul(ext_sub_base(znn, MONTY_R1), invt) : ext_zero();
    if (sh.air) { scput(scc.INVT, invt); scput(scc.SELL, sell); }
    std::vector<Ext> fp{fa};
    for (int i = 1; i <= scc.mb; i++) { fp.push_back(ext_mul(fp.back(), fp.back())); scput(scc.fp(i), fp.back()); }
    Ext offn = fp[(size_t)scc.bits[0]];
    for (int k = 1; k < scc.nbits; k++) { offn = ext_mul(offn, fp[(size_t)scc.bits[k]]); scput(scc.pr(k), offn); }
    const Ext offq = ext_mul(offn, offn), fa4 = fp[2];
    scput(scc.OFFN, offn); scput(scc.OFFQ, offq);
    Ext qz[8], hq[8];
    for (int j = 0; j < 8; j++) { qz[j] = ext_canon(pw + wt.o_qz + 4 * (size_t)j); scput(scc.qz(j), qz[j]); }
    hq[7] = qz[7];
    for (int j = 6; j >= 0; j--) { hq[j] = ext_add(qz[j], ext_mul(fa, hq[j + 1])); scput(scc.hq(j), hq[j]); }
    const Ext yq = hq[0];
    Ext qk[2];
    for (int k = 0; k < 2; k++) {
        Ext q = ext_zero();
        for (int t = 0; t < 4; t++) { Ext basis = ext_zero(); basis.c[t] = MONTY_R1; q = ext_add(q, ext_mul(basis, qz[4 * k + t])); }
        qk[k] = q;
        scput(k ? scc.QK1 : scc.QK0, q);
    }
    uint32_t za[2], zb[2];
    zps_consts(sh, za, zb);
    const Ext quo = ext_add(ext_mul(ext_add_base(ext_mul_base(znn, to_monty(za[0])), to_monty(zb[0])), qk[0]), ext_mul(ext_add_base(ext_mul_base(znn, to_monty(za[1])), to_monty(zb[1])), qk[1]));
    scput(scc.QUO, quo);
    sec("scalars");
    // (d) OPENED: the opened values, their fa-weighted sums, the AIR folded with alpha.  The rows behind the last proof carry ITS constants.
    const size_t op_rows = (size_t)1 << m.height[C_OPENED];
    const size_t op_lo = (size_t)p * (size_t)sh.G, op_hi = p + 1 == sh.NP ? op_rows : op_lo + (size_t)sh.G;
    Ext yl = ext_zero(), yn = ext_zero(), acc = ext_zero(), pwr = ext_one(), res_yl = yl, res_yn = yn, res_acc = acc;
    for (size_t row = op_lo; row < op_hi; row++) {
        const size_t g = row - op_lo;
        const bool active = g < (size_t)sh.G;
        if (!active && g % (size_t)sh.G == 0) { pwr = ext_one(); yl = yn = acc = ext_zero(); }      // (a padding "segment" starts like a proof's)
        uint32_t* r = ht.op.data() + (size_t)OP_MAIN * row;
        auto put = [&](uint32_t name, const Ext& e) { put_ext(r, 4u * name, e); };
        put(O_FA, fa); put(O_FA4, fa4); put(O_ALPHA, alpha); put(O_SELT, selt); put(O_SELF, self_);
        Ext v[8];
        uint32_t k1 = 0, k2 = 0, k3 = 0;
        if (active) {
            for (int i = 0; i < 4; i++) { v[i] = ext_canon(pw + wt.o_loc + 4 * (4 * g + (size_t)i)); v[4 + i] = ext_canon(pw + wt.o_nxt + 4 * (4 * g + (size_t)i)); }
            k1 = to_monty((uint32_t)g + 1u); k2 = to_monty(2u * (uint32_t)g + 3u); k3 = to_monty(5u * (uint32_t)g + 7u);
        } else {
            for (auto& e : v) e = ext_zero();
            pwr = yl = yn = acc = ext_zero();
        }
        for (uint32_t i = 0; i < 8; i++) put(O_A + i, v[i]);
        put(O_PW, pwr);
        const Ext pwn = ext_mul(pwr, fa4);
        put(O_PWN, pwn);
        Ext il[2];
        for (int s = 0; s < 2; s++) {
            const Ext* x = v + 4 * s;
            const Ext h2 = ext_add(x[2], ext_mul(fa, x[3])), h1 = ext_add(x[1], ext_mul(fa, h2));
            il[s] = ext_add(x[0], ext_mul(fa, h1));
            put(s ? O_G2 : O_H2, h2); put(s ? O_G1 : O_H1, h1); put(s ? O_INX : O_IL, il[s]);
        }
        put(O_YLIN, yl); put(O_YNIN, yn);
        yl = ext_add(yl, ext_mul(pwr, il[0])); yn = ext_add(yn, ext_mul(pwr, il[1]));
        put(O_YLO, yl); put(O_YNO, yn);
        if (!sh.air) {
            const Ext a2 = ext_mul(v[0], v[0]), ab = ext_mul(v[0], v[1]);
            put(O_A2, a2); put(O_AB, ab); put(O_ACCIN, acc);
            const Ext u1 = ext_add(ext_mul(acc, alpha), ext_sub_base(ext_sub(v[2], ext_mul(a2, v[1])), k1));
            const Ext u2 = ext_add(ext_mul(u1, alpha), ext_mul(selt, ext_sub_base(ext_sub(ext_sub(v[7], ab), v[2]), k2)));
            acc = ext_add(ext_mul(u2, alpha), ext_mul(self_, ext_sub_base(v[3], k3)));
            put(O_U1, u1); put(O_U2, u2); put(O_ACCO, acc);
        }
        pwr = pwn;
        if (g + 1 == (size_t)sh.G) { res_yl = yl; res_yn = yn; res_acc = acc; }
    }
    if (sh.air) {
        // EVAL: the program's terms at zeta, one row each; the fold with alpha runs down the rows.  The rows behind the last proof stay zero
        // (no receive, coefficient 0: the running sum passes through them unread).
        const size_t nt = sh.terms.size();
        auto value_of = [&](uint32_t key) -> Ext {
            if (key == 0) return ext_one();
            const uint32_t W2 = 2u * (uint32_t)W;
            if (key <= (uint32_t)W) return ext_canon(pw + wt.o_loc + 4 * (size_t)(key - 1u));
            if (key <= W2) return ext_canon(pw + wt.o_nxt + 4 * (size_t)(key - 1u - (uint32_t)W));
            if (key <= W2 + (uint32_t)sh.NPUB) return ext_from_base(to_monty(public_values[key - 1u - W2] % P));
            const uint32_t which = key - 1u - W2 - (uint32_t)sh.NPUB;
            return which == 0 ? self_ : (which == 1 ? sell : selt);
        };
        Ext run = ext_zero();
        for (size_t i = 0; i < nt; i++) {
            uint32_t* r = ht.evl.data() + (size_t)EV_MAIN * ((size_t)p * nt + i);
            const Shape::ETerm& e = sh.terms[i];
            const Ext f0 = value_of(e.key[0]), f1 = value_of(e.key[1]), f2 = value_of(e.key[2]);
            const Ext mm = ext_mul(f0, f1), tv = ext_mul_base(ext_mul(mm, f2), to_monty(e.coeff));
            put_ext(r, EV_F0, f0); put_ext(r, EV_F0 + 4, f1); put_ext(r, EV_F0 + 8, f2); put_ext(r, EV_M, mm); put_ext(r, EV_TV, tv);
            put_ext(r, EV_ACCIN, run);
            run = ext_add(e.first ? ext_mul(run, alpha) : run, tv);
            put_ext(r, EV_ACCO, run); put_ext(r, EV_ALPHA, alpha);
        }
        res_acc = run;
    }
    scput(scc.YL, res_yl); scput(scc.YN, res_yn); scput(scc.ACC, res_acc);
    if (!ext_eq(res_acc, ext_mul(quo, ext_sub_base(znn, MONTY_R1))))
        return fail(on_device ? ZKHIP_ERR_VERIFY : ZKHIP_ERR_INTERNAL, on_device ? "prove_shard_verifier: proof " + std::to_string(p) + " rejected: the constraints do not match the quotient at zeta"
                                                                       : std::string("prove_shard_verifier: the AIR identity at zeta does not hold"));
    sec("opened + eval");
    // (e) ROWSUM
    std::vector<Ext> at((size_t)Q), aq((size_t)Q);
    if (!on_device) {
        const size_t rs_rows = (size_t)1 << m.height[C_ROWSUM], per = (size_t)Q * (size_t)(sh.WB + 1);
        const size_t lo = (size_t)p * per, hi = p + 1 == sh.NP ? rs_rows : lo + per;
        Ext a = ext_zero();
        for (size_t r = lo; r < hi; r++) {
            uint32_t* row = ht.rs.data() + (size_t)RS_MAIN * r;
            put_ext(row, RS_FA, fa);
            if (r >= lo + per) continue;
            int pp, q, b;
            rowsum_row(sh, r, &pp, &q, &b);
            const uint32_t* vals = b == sh.WB ? pw + wt.q_trow(q) + (size_t)W + 8 * (size_t)H : pw + wt.q_trow(q) + 8 * (size_t)b;
            if (b == sh.WB - 1 || b == sh.WB) a = ext_zero();
            put_ext(row, RS_ACCIN, a);
            for (int s = 7; s >= 0; s--) {
                const uint32_t v = to_monty(vals[s]);
                row[RS_V + s] = v;
                a = ext_add_base(ext_mul(a, fa), v);
                put_ext(row, RS_T + 4u * (uint32_t)s, a);
            }
            if (b == 0) at[(size_t)q] = a;
            if (b == sh.WB) aq[(size_t)q] = a;
        }
    }
    // (f) QUERY
    if (!on_device) {
        constexpr QCols qc = qcols();
        const size_t q_rows = (size_t)1 << m.height[C_QUERY], lo = (size_t)p * (size_t)Q, hi = p + 1 == sh.NP ? q_rows : lo + (size_t)Q;
        for (size_t rr = lo; rr < hi; rr++) {
            uint32_t* row = ht.q.data() + (size_t)Q_MAIN * rr;
            auto put = [&](uint32_t col, const Ext& e) { put_ext(row, col - Q_PRE, e); };
            put(qc.ZETA, zeta); put(qc.ZNX, znx); put(qc.YL, res_yl); put(qc.YN, res_yn); put(qc.YQ, yq); put(qc.OFFN, offn); put(qc.OFFQ, offq);
            const size_t r = rr - lo;
            if (r >= (size_t)Q) continue;
            const uint32_t index = wt.indices[r];
            const uint32_t xq = fpow(two_adic_generator(H), reverse_bits(index, H));
            const Ext x = ext_from_base(fmul(MONTY_GEN, xq));
            const Ext i1 = ext_inv(ext_sub(x, zeta)), i2 = ext_inv(ext_sub(x, znx));
            const Ext p1 = ext_mul(ext_sub(at[r], res_yl), i1), p2 = ext_mul(ext_sub(at[r], res_yn), i2), p2o = ext_mul(offn, p2);
            const Ext p3 = ext_mul(ext_sub(aq[r], yq), i1), p3o = ext_mul(offq, p3), ro = ext_add(ext_add(p1, p2o), p3o);
            for (int i = 0; i < 4; i++) if (from_monty(ro.c[i]) != wt.values[4 * r + (size_t)i]) return fail(ZKHIP_ERR_INTERNAL, "prove_shard_verifier: a reduced opening is not the verifier's");
            row[qc.IDX - Q_PRE] = to_monty(index); row[qc.XQ - Q_PRE] = xq;
            put(qc.RO, ro); put(qc.AT, at[r]); put(qc.AQ, aq[r]); put(qc.I1, i1); put(qc.I2, i2); put(qc.P1, p1); put(qc.P2, p2); put(qc.P2O, p2o); put(qc.P3, p3); put(qc.P3O, p3o);
        }
    }
    sec("rowsum + query");
    // (g) TS
    for (int T = 0; T < sh.NTS; T++) {
        uint32_t* row = ht.ts.data() + (size_t)TS_MAIN * ((size_t)p * (size_t)sh.NTS + (size_t)T);
        for (int j = 0; j < 8; j++) row[j] = to_monty(chain_in[16 * (size_t)T + (size_t)j]);       // the absorbed words, and whatever the kept ones are
        if (sh.has_challenge(T)) put_ext(row, 16, chal[(size_t)T]);
        if (T == sh.HL / 8) for (int j = 0; j < 8; j++) row[8 + j] = to_monty(pw[wt.o_troot + j]);
    }
    sec("ts");
    // (h) SAMPLES: this proof's NS rows
    {
        std::vector<uint32_t> t_sm, drawn;
        frichip::samples_main(R, (size_t)Q, lg((size_t)sh.NS), samples.data(), t_sm, drawn);
        if (sh.PB && (samples[0] & ((1u << sh.PB) - 1u)))
            return fail(on_device ? ZKHIP_ERR_VERIFY : ZKHIP_ERR_INTERNAL, on_device ? "prove_shard_verifier: proof " + std::to_string(p) + " rejected: proof of work" : std::string("prove_shard_verifier: the witness does not satisfy the proof of work"));
        if (on_device) { if (drawn.size() != (size_t)Q) return fail(ZKHIP_ERR_INTERNAL, "prove_shard_verifier: query indices"); wt.indices = drawn; }
        else if (std::memcmp(drawn.data(), wt.indices.data(), 4 * (size_t)Q) != 0) return fail(ZKHIP_ERR_INTERNAL, "prove_shard_verifier: the query indices are not the ones the transcript draws");
        std::memcpy(ht.sm.data() + (size_t)frichip::S_MAIN * (size_t)p * (size_t)sh.NS, t_sm.data(), (size_t)frichip::S_MAIN * (size_t)sh.NS * 4);
    }
    if (on_device) {
        // what the device reads of this proof, and its entries of the P2R work lists with the pairs left to the device (the words go up with the lists; a layer's
        // path is read where it sits in them)
        uint32_t* v = dev_vals;
        size_t at_v = 0;
        for (int q = 0; q < Q; q++) v[at_v++] = wt.indices[(size_t)q];
        for (size_t i = 0; i < 4 * (size_t)R; i++) v[at_v++] = wt.betas[i];
        const Ext exts[8] = {fa, zeta, znx, res_yl, res_yn, yq, offn, offq};
        for (const Ext& e : exts) for (int j = 0; j < 4; j++) v[at_v++] = e.c[j];
        const size_t row0 = (size_t)p * sh.p2_rows;
        for (int T = 0; T < sh.NT; T++) ht.trows[(size_t)p * (size_t)sh.NT + (size_t)T] = (uint32_t)(row0 + (size_t)T);
        std::memcpy(ht.chain_in.data() + 16 * (size_t)p * (size_t)sh.NT, chain_in.data(), chain_in.size() * 4);
        const size_t seg = 8 * (size_t)Q * (size_t)R + words.size();
        const size_t off_pairs = (size_t)p * seg, off_words = off_pairs + 8 * (size_t)Q * (size_t)R;
        std::memcpy(ht.data.data() + off_words, words.data(), words.size() * 4);
        const size_t chains_per = (size_t)Q * (size_t)R + 2 * (size_t)Q;
        uint32_t* desc = ht.desc.data() + 6 * (size_t)p * chains_per;
        const uint32_t** want = ht.want_roots.data() + (size_t)p * chains_per;
        size_t row = row0 + sh.p2_fri0, ch = 0;
        for (int q = 0; q < Q; q++) {
            uint32_t idx = wt.indices[(size_t)q];
            size_t fat = wt.q_trow(q) + (size_t)W + 8 * (size_t)H + 8 + 8 * (size_t)H;       // the query's first FRI word
            for (int l = 0; l < R; l++, ch++) {
                const uint32_t k = idx >> 1;
                const int lh = H - (l + 1);
                const uint32_t d[6] = {(uint32_t)row, 1u, (uint32_t)(off_pairs + 8 * ch), (uint32_t)lh, k, (uint32_t)(off_words + fat + 4)};
                std::memcpy(desc + 6 * ch, d, sizeof d);
                want[ch] = wt.lroots.data() + 8 * (size_t)l;
                row += 1 + (size_t)lh;
                fat += 4 + 8 * (size_t)lh;
                idx = k;
            }
        }
        for (int which = 0; which < 2; which++)
            for (int q = 0; q < Q; q++, ch++) {
                const size_t trow = wt.q_trow(q), tpath = trow + (size_t)W, qrow = tpath + 8 * (size_t)H, qpath = qrow + 8;
                const uint32_t d[6] = {(uint32_t)row, which ? 1u : (uint32_t)sh.WB, (uint32_t)(off_words + (which ? qrow : trow)), (uint32_t)H, wt.indices[(size_t)q],
                                       (uint32_t)(off_words + (which ? qpath : tpath))};
                std::memcpy(desc + 6 * ch, d, sizeof d);
                want[ch] = pw + (which ? wt.o_qroot : wt.o_troot);
                row += (which ? 1 : (size_t)sh.WB) + (size_t)H;
            }
        if (row != row0 + sh.p2_rows || ch != chains_per) return fail(ZKHIP_ERR_INTERNAL, "prove_shard_verifier: row layout");
        return ZKHIP_OK;
    }
    sec("samples");
    // (i) this proof's entries of the P2R work lists: transcript rows, then Q R FRI paths (leaf block = the pair), Q trace openings, Q quotient openings
    {
        // (every proof writes ITS slices of the shared lists: the proofs are filled side by side on a few host threads)
        const size_t row0 = (size_t)p * sh.p2_rows;
        for (int T = 0; T < sh.NT; T++) ht.trows[(size_t)p * (size_t)sh.NT + (size_t)T] = (uint32_t)(row0 + (size_t)T);
        std::memcpy(ht.chain_in.data() + 16 * (size_t)p * (size_t)sh.NT, chain_in.data(), chain_in.size() * 4);
        const size_t per_q_paths = zkhip_fri_view_path_words(R);
        const size_t seg = 8 * (size_t)Q * (size_t)R + wt.paths.size() + words.size();
        const size_t off_pairs = (size_t)p * seg, off_paths = off_pairs + 8 * (size_t)Q * (size_t)R, off_words = off_paths + wt.paths.size();
        std::memcpy(ht.data.data() + off_paths, wt.paths.data(), wt.paths.size() * 4);
        std::memcpy(ht.data.data() + off_words, words.data(), words.size() * 4);
        const size_t chains_per = (size_t)Q * (size_t)R + 2 * (size_t)Q;
        uint32_t* desc = ht.desc.data() + 6 * (size_t)p * chains_per;
        const uint32_t** want = ht.want_roots.data() + (size_t)p * chains_per;
        size_t row = row0 + sh.p2_fri0, ch = 0;
        for (int q = 0; q < Q; q++) {
            uint32_t idx = wt.indices[(size_t)q];
            Ext own = ext_canon(wt.values.data() + 4 * (size_t)q);
            for (int l = 0; l < R; l++, ch++) {
                const uint32_t bit = idx & 1u, k = idx >> 1;
                const Ext sib = ext_canon(wt.siblings.data() + 4 * ((size_t)q * (size_t)R + (size_t)l));
                const Ext e0 = bit ? sib : own, e1 = bit ? own : sib;
                uint32_t* pair = ht.data.data() + off_pairs + 8 * ch;
                for (int i = 0; i < 4; i++) { pair[i] = from_monty(e0.c[i]); pair[4 + i] = from_monty(e1.c[i]); }
                const int lh = H - (l + 1);
                const uint32_t d[6] = {(uint32_t)row, 1u, (uint32_t)(off_pairs + 8 * ch), (uint32_t)lh, k,
                                       (uint32_t)(off_paths + (size_t)q * per_q_paths + 8 * ((size_t)l * (size_t)R - (size_t)l * ((size_t)l - 1) / 2))};
                std::memcpy(desc + 6 * ch, d, sizeof d);
                want[ch] = wt.lroots.data() + 8 * (size_t)l;
                row += 1 + (size_t)lh;
                const uint32_t xi = finv(fpow(two_adic_generator(lh + 1), reverse_bits(k, lh)));
                const Ext beta = ext_canon(wt.betas.data() + 4 * (size_t)l);
                own = ext_add(ext_mul_base(ext_add(e0, e1), MONTY_INV2), ext_mul(beta, ext_mul_base(ext_sub(e0, e1), fmul(MONTY_INV2, xi))));
                idx = k;
            }
        }
        for (int which = 0; which < 2; which++)
            for (int q = 0; q < Q; q++, ch++) {
                const size_t trow = wt.q_trow(q), tpath = trow + (size_t)W, qrow = tpath + 8 * (size_t)H, qpath = qrow + 8;
                const uint32_t d[6] = {(uint32_t)row, which ? 1u : (uint32_t)sh.WB, (uint32_t)(off_words + (which ? qrow : trow)), (uint32_t)H, wt.indices[(size_t)q],
                                       (uint32_t)(off_words + (which ? qpath : tpath))};
                std::memcpy(desc + 6 * ch, d, sizeof d);
                want[ch] = pw + (which ? wt.o_qroot : wt.o_troot);
                row += (which ? 1 : (size_t)sh.WB) + (size_t)H;
            }
        if (row != row0 + sh.p2_rows || ch != chains_per) return fail(ZKHIP_ERR_INTERNAL, "prove_shard_verifier: row layout");
    }
    sec("work lists");
    return ZKHIP_OK;
}
}  // namespace
}  // namespace rec
// ---- round 6: the per-query part of the shard verifier's tables ON THE DEVICE (what fill_one's sections (e), (f) and the pair arithmetic of (i) compute on the
// host): one lane per (proof, query) fills the query's ROWSUM rows and its QUERY row from the proof's words, and hands the fold kernel its inputs -- the
// reduced opening, the layers' siblings --; a second kernel walks the fold chain once more for the layers' pairs (the leaf blocks of the FRI paths' chains).
namespace rec {
struct SvWitArgs {
    uint32_t* data; uint64_t seg; uint32_t words_off;       // proof p's words: data + p seg + words_off; its pairs: data + p seg + 8 (q R + l)
    const uint32_t* vals; uint32_t vstride;                 // sv_vals_words per proof
    uint32_t NP, Q, R, H, W, WB;
    uint32_t o_queries, per_query;
    uint32_t* rs; uint64_t rs_rows; uint32_t* qt; uint64_t q_rows;
    uint32_t *f_betas, *f_indices, *f_values, *f_siblings;  // the fold kernel's inputs, canonical
};
__device__ __forceinline__ Ext sv_ld4(const uint32_t* p) { return Ext{{p[0], p[1], p[2], p[3]}}; }
__device__ __forceinline__ void sv_put(uint32_t* r, uint32_t col, const Ext& e) { r[col] = e.c[0]; r[col + 1] = e.c[1]; r[col + 2] = e.c[2]; r[col + 3] = e.c[3]; }
__global__ void __launch_bounds__(64) sv_rowsum_query_kernel(SvWitArgs a) {
    constexpr QCols qc = qcols();
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nq = (uint64_t)a.NP * a.Q;
    const uint32_t ve = a.Q + 4u * a.R;                    // where a proof's extension values start: fa zeta znx yl yn yq offn offq
    if (gid >= nq) {
        // the rows behind the last proof carry ITS constants (ROWSUM: fa; QUERY: the seven constants)
        const uint32_t* v = a.vals + (uint64_t)(a.NP - 1) * a.vstride + ve;
        const Ext fa = sv_ld4(v);
        const uint64_t k = gid - nq, stride = (uint64_t)gridDim.x * blockDim.x - nq;
        for (uint64_t r = nq * (a.WB + 1) + k; r < a.rs_rows; r += stride) sv_put(a.rs + (uint64_t)RS_MAIN * r, RS_FA, fa);
        for (uint64_t r = nq + k; r < a.q_rows; r += stride) {
            uint32_t* row = a.qt + (uint64_t)Q_MAIN * r;
            sv_put(row, qc.ZETA - Q_PRE, sv_ld4(v + 4)); sv_put(row, qc.ZNX - Q_PRE, sv_ld4(v + 8)); sv_put(row, qc.YL - Q_PRE, sv_ld4(v + 12)); sv_put(row, qc.YN - Q_PRE, sv_ld4(v + 16));
            sv_put(row, qc.YQ - Q_PRE, sv_ld4(v + 20)); sv_put(row, qc.OFFN - Q_PRE, sv_ld4(v + 24)); sv_put(row, qc.OFFQ - Q_PRE, sv_ld4(v + 28));
        }
        return;
    }
    const uint32_t q = (uint32_t)(gid % a.Q), p = (uint32_t)(gid / a.Q);
    const uint32_t* vv = a.vals + (uint64_t)p * a.vstride;
    const uint32_t* v = vv + ve;
    const Ext fa = sv_ld4(v), zeta = sv_ld4(v + 4), znx = sv_ld4(v + 8), yl = sv_ld4(v + 12), yn = sv_ld4(v + 16), yq = sv_ld4(v + 20), offn = sv_ld4(v + 24), offq = sv_ld4(v + 28);
    const uint32_t* pw = a.data + (uint64_t)p * a.seg + a.words_off;
    const uint32_t* qw = pw + a.o_queries + (uint64_t)q * a.per_query;
    // ---- ROWSUM: the trace row's blocks from the last to the first, then the quotient row
    Ext acc = ext_zero(), at = ext_zero(), aq = ext_zero();
    for (uint32_t i = 0; i <= a.WB; i++) {
        const uint32_t b = i < a.WB ? a.WB - 1 - i : a.WB;
        const uint32_t* vals = b == a.WB ? qw + a.W + 8 * a.H : qw + 8 * b;
        uint32_t* row = a.rs + (uint64_t)RS_MAIN * (gid * (a.WB + 1) + i);
        sv_put(row, RS_FA, fa);
        if (b == a.WB - 1 || b == a.WB) acc = ext_zero();
        sv_put(row, RS_ACCIN, acc);
        for (int sidx = 7; sidx >= 0; sidx--) {
            const uint32_t x = to_monty(vals[sidx]);
            row[RS_V + sidx] = x;
            acc = ext_add_base(ext_mul(acc, fa), x);
            sv_put(row, RS_T + 4u * (uint32_t)sidx, acc);
        }
        if (b == 0) at = acc;
        if (b == a.WB) aq = acc;
    }
    // ---- QUERY
    const uint32_t index = vv[q];
    const uint32_t xq = fpow(two_adic_generator((int)a.H), reverse_bits(index, (int)a.H));
    const Ext x = ext_from_base(fmul(MONTY_GEN, xq));
    const Ext i1 = ext_inv(ext_sub(x, zeta)), i2 = ext_inv(ext_sub(x, znx));
    const Ext p1 = ext_mul(ext_sub(at, yl), i1), p2 = ext_mul(ext_sub(at, yn), i2), p2o = ext_mul(offn, p2);
    const Ext p3 = ext_mul(ext_sub(aq, yq), i1), p3o = ext_mul(offq, p3), ro = ext_add(ext_add(p1, p2o), p3o);
    uint32_t* row = a.qt + (uint64_t)Q_MAIN * gid;
    sv_put(row, qc.ZETA - Q_PRE, zeta); sv_put(row, qc.ZNX - Q_PRE, znx); sv_put(row, qc.YL - Q_PRE, yl); sv_put(row, qc.YN - Q_PRE, yn);
    sv_put(row, qc.YQ - Q_PRE, yq); sv_put(row, qc.OFFN - Q_PRE, offn); sv_put(row, qc.OFFQ - Q_PRE, offq);
    row[qc.IDX - Q_PRE] = to_monty(index); row[qc.XQ - Q_PRE] = xq;
    sv_put(row, qc.RO - Q_PRE, ro); sv_put(row, qc.AT - Q_PRE, at); sv_put(row, qc.AQ - Q_PRE, aq); sv_put(row, qc.I1 - Q_PRE, i1); sv_put(row, qc.I2 - Q_PRE, i2);
    sv_put(row, qc.P1 - Q_PRE, p1); sv_put(row, qc.P2 - Q_PRE, p2); sv_put(row, qc.P2O - Q_PRE, p2o); sv_put(row, qc.P3 - Q_PRE, p3); sv_put(row, qc.P3O - Q_PRE, p3o);
    // ---- the fold kernel's inputs
    a.f_indices[gid] = index;
    for (int j = 0; j < 4; j++) a.f_values[4 * gid + (uint64_t)j] = from_monty(ro.c[j]);
    uint64_t fat = (uint64_t)a.W + 8 * (uint64_t)a.H + 8 + 8 * (uint64_t)a.H;
    for (uint32_t l = 0; l < a.R; l++) {
        for (int j = 0; j < 4; j++) a.f_siblings[4 * (gid * a.R + l) + (uint64_t)j] = qw[fat + (uint64_t)j];
        fat += 4 + 8 * (uint64_t)(a.H - 1 - l);
    }
    if (q == 0) for (uint32_t i = 0; i < 4 * a.R; i++) a.f_betas[(uint64_t)p * 4 * a.R + i] = vv[a.Q + i];
}
// the layers' pairs (canonical) into the chains' leaf blocks: the fold chain of (proof, query) once more
__global__ void __launch_bounds__(64) sv_pairs_kernel(SvWitArgs a) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)a.NP * a.Q) return;
    const uint32_t q = (uint32_t)(gid % a.Q), p = (uint32_t)(gid / a.Q);
    uint32_t idx = a.f_indices[gid];
    Ext own = Ext{{to_monty(a.f_values[4 * gid]), to_monty(a.f_values[4 * gid + 1]), to_monty(a.f_values[4 * gid + 2]), to_monty(a.f_values[4 * gid + 3])}};
    uint32_t* pairs = a.data + (uint64_t)p * a.seg + 8 * ((uint64_t)q * a.R);
    for (uint32_t l = 0; l < a.R; l++) {
        const uint32_t bit = idx & 1u, k = idx >> 1;
        const uint32_t* sp = a.f_siblings + 4 * (gid * a.R + l);
        const uint32_t* bp = a.f_betas + (uint64_t)p * 4 * a.R + 4 * l;
        const Ext sib = Ext{{to_monty(sp[0]), to_monty(sp[1]), to_monty(sp[2]), to_monty(sp[3])}}, beta = Ext{{to_monty(bp[0]), to_monty(bp[1]), to_monty(bp[2]), to_monty(bp[3])}};
        const Ext e0 = bit ? sib : own, e1 = bit ? own : sib;
        uint32_t* pair = pairs + 8 * l;
        for (int i = 0; i < 4; i++) { pair[i] = from_monty(e0.c[i]); pair[4 + i] = from_monty(e1.c[i]); }
        const int lh = (int)a.H - (int)(l + 1);
        const uint32_t xi = finv(fpow(two_adic_generator(lh + 1), reverse_bits(k, lh)));
        own = ext_add(ext_mul_base(ext_add(e0, e1), MONTY_INV2), ext_mul(beta, ext_mul_base(ext_sub(e0, e1), fmul(MONTY_INV2, xi))));
        idx = k;
    }
}
}  // namespace rec
// (zkhip_release_cached_contexts, jobs.cpp: the pooled host tables go with the pooled contexts)
void rec_release_host_tables() { rec::word_pool().clear(); }
}  // namespace zk

static int shard_verifier_prove_impl(zkhip_ctx* ctx, const zkhip_machine_key* key, const uint32_t* program, size_t program_words, const uint8_t* const* inner, const size_t* inner_len,
                                     size_t n_proofs, int log_n, uint32_t width,
                                     const uint32_t* public_values, size_t n_public, const zkhip_params* inner_prm, const zkhip_params* outer,
                                     uint8_t* proof, size_t cap, size_t* len) {
    using namespace zk::rec;
    CHECK_CTX(ctx);
    if (!key || !inner || !inner_len || !inner_prm || !outer || !proof || !len || (n_public && !public_values)) return fail(ZKHIP_ERR_INVALID, "prove_shard_verifier: null argument");
    if (inner_prm->log_blowup != 1 || inner_prm->logup_pairs != 0 || inner_prm->log_fold > 1 || inner_prm->log_final != 0 || (inner_prm->hash_width != 0 && inner_prm->hash_width != 16) ||
        inner_prm->code_width != 0)
        return fail(ZKHIP_ERR_INVALID, "prove_shard_verifier: version-1 shard proofs (SP1 shape: blowup 2, fold by 2, constant final value, no lookups)");
#ifdef ZKHIP_AB_HOOKS
    static const bool timing = getenv("ZKHIP_REC_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  [shard verifier] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
#else
    auto lap = [](const char*) {};
#endif
    Shape sh;
    ZK_TRY(make_shape(log_n, width, (size_t)inner_prm->num_queries, inner_prm->pow_bits, n_public, n_proofs, sh, program, program_words));
    ZK_TRY(check_outer(sh, outer));
    const auto mp = machine_of(sh);
    const Machine& m = *mp;
    const int R = sh.R, Q = sh.Q, NP = sh.NP;
    const ScCols scc = sc_cols(sh);
    const uint32_t sc_w = rup4(scc.end - sc_pre(sh));
    const uint32_t w_main[N_CHIPS] = {P2_MAIN, RS_MAIN, frichip::width_of(R, true, true), TS_MAIN, Q_MAIN, OP_MAIN, frichip::S_MAIN, sc_w, EV_MAIN};
    lap("shape + machine");
    // round 6: ROWSUM, QUERY, the fold chains and the layers' pairs are the device's (sv_rowsum_query_kernel, sv_pairs_kernel); ZKHIP_REC_HOST=1 (and a call
    // from inside a lock-step batch, whose launches are merged) keeps the host's walk
    const bool host_forced = witnesses_on_host();
    const bool on_device = !host_forced && !t_batcher;
    HostTables ht;
    if (!ht.sc.reset((size_t)sc_w << m.height[C_SCALARS]) || !ht.op.reset((size_t)OP_MAIN << m.height[C_OPENED]) || (!on_device && !ht.rs.reset((size_t)RS_MAIN << m.height[C_ROWSUM])) ||
        (!on_device && !ht.q.reset((size_t)Q_MAIN << m.height[C_QUERY])) || !ht.ts.reset((size_t)TS_MAIN << m.height[C_TS]) || !ht.sm.reset((size_t)frichip::S_MAIN << m.height[C_SAMPLES]) ||
        (sh.air && !ht.evl.reset((size_t)EV_MAIN << m.height[C_EVAL])))
        return fail(ZKHIP_ERR_NOMEM, "prove_shard_verifier: no host memory for the machine's tables");
    std::vector<Witness> wts((size_t)NP);
    std::vector<std::vector<uint32_t>> words((size_t)NP);
    std::vector<Ext> fas((size_t)NP);
    lap("host: tables zeroed");
    {
        // the shared lists have a fixed slice per proof (the shape fixes every size): the proofs are filled side by side
        const size_t chains_per = (size_t)Q * (size_t)R + 2 * (size_t)Q, seg = 8 * (size_t)Q * (size_t)R + (on_device ? 0 : zkhip_fri_view_path_words(R) * (size_t)Q) + inner_len[0] / 4;
        for (int p = 0; p < NP; p++) {
            if (!inner[p]) return fail(ZKHIP_ERR_INVALID, "prove_shard_verifier: null proof");
            if (inner_len[p] != inner_len[0] || inner_len[p] % 4) return fail(ZKHIP_ERR_INVALID, "prove_shard_verifier: the proofs of one call have one shape, hence one length");
        }
        if ((size_t)NP * seg >= ((size_t)1 << 32)) return fail(ZKHIP_ERR_INVALID, "prove_shard_verifier: too much witness data for one call");
        ht.want_roots.assign((size_t)NP * chains_per, nullptr);
        {   // the four P2R work lists are filled IN PLACE in the context's pinned block and go up as one DMA
            const size_t n_desc = 6 * (size_t)NP * chains_per, n_data = (size_t)NP * seg, n_cin = 16 * (size_t)NP * (size_t)sh.NT, n_trows = (size_t)NP * (size_t)sh.NT;
            void* up;
            ZK_TRY(ctx_host_pinned(ctx, (n_desc + n_data + n_cin + n_trows + (on_device ? (size_t)NP * sv_vals_words(sh) : 0)) * 4, &up));
            ht.desc = HostSpan{(uint32_t*)up, n_desc}; ht.data = HostSpan{ht.desc.p + n_desc, n_data};
            ht.chain_in = HostSpan{ht.data.p + n_data, n_cin}; ht.trows = HostSpan{ht.chain_in.p + n_cin, n_trows};
        }
        uint32_t* const vals_host = on_device ? ht.trows.p + ht.trows.size() : nullptr;      // (device mode: the per-proof values behind the lists, same DMA)
        std::vector<SvTranscript> pres;
        if (NP >= 8 && p2x16_available() && !t_batcher) {           // (fewer proofs than half the lanes: a thread per proof walks its own chain sooner)
            // every proof's transcript before the fills: sixteen chains per permutation, a few threads when there are several groups
            pres.assign((size_t)NP, SvTranscript{});
            const int ng = (NP + 15) / 16;
            auto group = [&](int g) {
                const uint32_t* pws[16]; const uint32_t* pubss[16]; SvTranscript* outp[16];
                int n = 0;
                for (int p = 16 * g; p < NP && n < 16; p++, n++) { pws[n] = (const uint32_t*)inner[p]; pubss[n] = public_values + (size_t)p * n_public; outp[n] = &pres[(size_t)p]; }
                try { sv_walk_transcripts_x16(sh, n, pws, pubss, outp); } catch (...) { for (int i = 0; i < n; i++) outp[i]->done = false; }
            };
            // (lengths were checked above: every proof has the shape's length, and make_shape / the first fill check that it IS the shape's)
            const size_t want_words = (size_t)(sh.air ? 20 : 8) + 8 + 8 + 8 * (size_t)sh.W + 32 + 8 * (size_t)R + 4 + 1;
            if (inner_len[0] / 4 > want_words) {
                if (ng == 1) group(0);
                else { HostPool tp(ng < 8 ? ng : 8); for (int g = 0; g < ng; g++) tp.submit([&group, g] { group(g); }); tp.wait(); }
            } else pres.clear();
        }
        lap("host: transcripts (sixteen per permutation)");
        std::vector<int> rcs((size_t)NP, ZKHIP_OK);
        std::vector<std::string> msgs((size_t)NP);
        auto one = [&](int p) {
            t_query_threads_cap = NP >= 16 ? 1 : 16 / NP;      // the proofs are filled side by side: a proof's host pass starts few threads of its own
            rcs[(size_t)p] = fill_one(sh, m, p, inner[p], inner_len[p], public_values + (size_t)p * n_public, inner_prm, wts[(size_t)p], words[(size_t)p], ht, &fas[(size_t)p], program, program_words,
                                      on_device ? vals_host + (size_t)p * sv_vals_words(sh) : nullptr, pres.empty() ? nullptr : &pres[(size_t)p]);
            if (rcs[(size_t)p] != ZKHIP_OK) msgs[(size_t)p] = zkhip_last_error();
            t_query_threads_cap = 0;
        };
        if (NP == 1 || t_batcher) for (int p = 0; p < NP; p++) one(p);
        else {
            HostPool pool(NP < 16 ? NP : 16);          // (a proof's pass is ~6 ms of hashing on one core; the boxes give a container 16)
            for (int p = 0; p < NP; p++) pool.submit([&one, p] { one(p); });
            pool.wait();
        }
        for (int p = 0; p < NP; p++) if (rcs[(size_t)p] != ZKHIP_OK) { set_error("proof " + std::to_string(p) + ": " + msgs[(size_t)p]); return rcs[(size_t)p]; }
    }
    lap("host: witnesses + tables");
    for (size_t r = (size_t)NP; r < ((size_t)1 << m.height[C_SCALARS]); r++) std::memcpy(ht.sc.data() + sc_w * r, ht.sc.data(), sc_w * 4);     // rows behind the proofs repeat row 0
    // device: the fold rows proof by proof (the last call fills the padding), then ONE launch for every Poseidon2 row
    void* dev[N_CHIPS] = {nullptr};
    const int slots[N_CHIPS] = {S_REC_A, S_REC_C, S_REC_B, S_REC_D, S_REC_E, S_REC_F, S_REC_G, S_REC_H, S_REC_I};
    for (int c = 0; c < m.n; c++) ZK_TRY(ctx_reserve(ctx, slots[c], ((size_t)w_main[c] << m.height[c]) * 4, &dev[c]));
    if (on_device) {
        // ---- lists, words and values up in ONE DMA; the per-query rows, the fold rows, the pairs and every Poseidon2 row on the device
        const size_t nq = (size_t)NP * (size_t)Q, vw = sv_vals_words(sh), n_chains = ht.desc.size() / 6;
        const size_t up_words = ht.desc.size() + ht.data.size() + ht.chain_in.size() + ht.trows.size() + (size_t)NP * vw, down_words = 8 * n_chains;
        void *stage, *scratch;
        ZK_TRY(ctx_reserve(ctx, S_STAGE, (up_words + down_words) * 4, &stage));
        const size_t n_fb = 4 * (size_t)R * (size_t)NP, n_fi = nq, n_fv = 4 * nq, n_fs = 4 * nq * (size_t)R;
        ZK_TRY(ctx_reserve(ctx, S_WIT_B, (n_fb + n_fi + n_fv + n_fs + n_fv) * 4, &scratch));
        uint32_t* d = (uint32_t*)stage;
        ZK_HIP(hipMemcpyAsync(d, ht.desc.data(), up_words * 4, hipMemcpyHostToDevice, ctx->stream));       // desc | data | chain_in | trows | vals: one pinned block
        ZK_HIP(hipMemsetAsync(dev[C_ROWSUM], 0, ((size_t)RS_MAIN << m.height[C_ROWSUM]) * 4, ctx->stream));
        ZK_HIP(hipMemsetAsync(dev[C_QUERY], 0, ((size_t)Q_MAIN << m.height[C_QUERY]) * 4, ctx->stream));
        SvWitArgs a{};
        a.data = d + ht.desc.size(); a.seg = ht.data.size() / (size_t)NP; a.words_off = (uint32_t)(8 * (size_t)Q * (size_t)R);
        a.vals = d + ht.desc.size() + ht.data.size() + ht.chain_in.size() + ht.trows.size(); a.vstride = (uint32_t)vw;
        a.NP = (uint32_t)NP; a.Q = (uint32_t)Q; a.R = (uint32_t)R; a.H = (uint32_t)sh.H; a.W = (uint32_t)sh.W; a.WB = (uint32_t)sh.WB;
        a.o_queries = (uint32_t)wts[0].o_queries; a.per_query = (uint32_t)wts[0].per_query;
        a.rs = (uint32_t*)dev[C_ROWSUM]; a.rs_rows = (uint64_t)1 << m.height[C_ROWSUM]; a.qt = (uint32_t*)dev[C_QUERY]; a.q_rows = (uint64_t)1 << m.height[C_QUERY];
        uint32_t* sc = (uint32_t*)scratch;
        a.f_betas = sc; a.f_indices = sc + n_fb; a.f_values = a.f_indices + n_fi; a.f_siblings = a.f_values + n_fv;
        uint32_t* d_finals = a.f_siblings + n_fs;
        hipLaunchKernelGGL(sv_rowsum_query_kernel, dim3((unsigned)((nq + 4096 + 63) / 64)), dim3(64), 0, ctx->stream, a);
        ZK_HIP(hipGetLastError());
        ZK_TRY(fri_gen_trace_dev(ctx, R, nq, a.f_betas, a.f_indices, a.f_values, a.f_siblings, m.height[C_FOLD], (uint32_t*)dev[C_FOLD], w_main[C_FOLD], d_finals, (size_t)NP, (uint32_t)sh.TREES));
        hipLaunchKernelGGL(sv_pairs_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, ctx->stream, a);
        ZK_HIP(hipGetLastError());
        lap("device: ROWSUM, QUERY, fold rows, pairs");
        p2chip::P2RArgs pa{};
        pa.desc = d; pa.data = a.data; pa.chain_inputs = pa.data + ht.data.size(); pa.trows = pa.chain_inputs + ht.chain_in.size();
        pa.n_chains = (uint32_t)n_chains; pa.n_transcript = (uint32_t)ht.trows.size(); pa.rows = (uint64_t)1 << m.height[C_P2R]; pa.used_rows = (uint64_t)NP * sh.p2_rows;
        pa.trace = (uint32_t*)dev[C_P2R]; pa.ld = P2_MAIN; pa.roots = d + up_words;
        ZK_HIP(launch_p2r_rows(pa, ctx->stream));
        std::vector<uint32_t> down(down_words), finals(n_fv);
        ZK_TRY(dev_d2h(ctx, finals.data(), d_finals, n_fv * 4));
        ZK_TRY(dev_d2h(ctx, down.data(), pa.roots, down_words * 4));
        for (int p = 0; p < NP; p++)
            for (int q = 0; q < Q; q++)
                if (std::memcmp(finals.data() + 4 * ((size_t)p * (size_t)Q + (size_t)q), wts[(size_t)p].w + wts[(size_t)p].o_final, 16) != 0)
                    return fail(ZKHIP_ERR_VERIFY, "prove_shard_verifier: proof " + std::to_string(p) + " rejected: a fold chain does not end in the final value");
        for (size_t c = 0; c < n_chains; c++)
            if (std::memcmp(down.data() + 8 * c, ht.want_roots[c], 32) != 0)
                return fail(ZKHIP_ERR_VERIFY, "prove_shard_verifier: proof " + std::to_string(c / (n_chains / (size_t)NP)) + " rejected: an opening does not end in its root");
        lap("device: P2R rows + roots back");
    } else {
    {   // the fold rows of every proof in ONE launch
        const size_t nq = (size_t)NP * (size_t)Q;
        std::vector<uint32_t> betas, indices, values, siblings, finals(4 * nq);
        for (int p = 0; p < NP; p++) {
            const Witness& wt = wts[(size_t)p];
            betas.insert(betas.end(), wt.betas.begin(), wt.betas.end()); indices.insert(indices.end(), wt.indices.begin(), wt.indices.end());
            values.insert(values.end(), wt.values.begin(), wt.values.end()); siblings.insert(siblings.end(), wt.siblings.begin(), wt.siblings.end());
        }
        ZK_TRY(fri_gen_trace(ctx, R, nq, betas.data(), indices.data(), values.data(), siblings.data(), m.height[C_FOLD], (uint32_t*)dev[C_FOLD], w_main[C_FOLD], finals.data(),
                             true, true, 0u, 0, -1, (size_t)NP, (uint32_t)sh.TREES));
        for (int p = 0; p < NP; p++)
            for (int q = 0; q < Q; q++)
                if (std::memcmp(finals.data() + 4 * ((size_t)p * (size_t)Q + (size_t)q), wts[(size_t)p].w + wts[(size_t)p].o_final, 16) != 0)
                    return fail(ZKHIP_ERR_INTERNAL, "prove_shard_verifier: a chain does not end in the final value");
    }
    lap("device: fold rows");
    {
        const size_t n_chains = ht.desc.size() / 6;
        const size_t up_words = ht.desc.size() + ht.data.size() + ht.chain_in.size() + ht.trows.size(), down_words = 8 * n_chains;
        void* stage;
        ZK_TRY(ctx_reserve(ctx, S_STAGE, (up_words + down_words) * 4, &stage));
        uint32_t* d = (uint32_t*)stage;
        ZK_TRY(dev_h2d(ctx, d, ht.desc.data(), up_words * 4));       // desc | data | chain_in | trows: one pinned block
        lap("upload: P2R work lists");
        p2chip::P2RArgs a{};
        a.desc = d; a.data = d + ht.desc.size(); a.chain_inputs = a.data + ht.data.size(); a.trows = a.chain_inputs + ht.chain_in.size();
        a.n_chains = (uint32_t)n_chains; a.n_transcript = (uint32_t)ht.trows.size(); a.rows = (uint64_t)1 << m.height[C_P2R]; a.used_rows = (uint64_t)NP * sh.p2_rows;
        a.trace = (uint32_t*)dev[C_P2R]; a.ld = P2_MAIN; a.roots = d + up_words;
        ZK_HIP(launch_p2r_rows(a, ctx->stream));
        std::vector<uint32_t> down(down_words);
        ZK_TRY(dev_d2h(ctx, down.data(), a.roots, down_words * 4));
        for (size_t c = 0; c < n_chains; c++)
            if (std::memcmp(down.data() + 8 * c, ht.want_roots[c], 32) != 0)      // the Merkle check of the inner proofs (the host pass left it to this kernel)
                return fail(ZKHIP_ERR_VERIFY, "prove_shard_verifier: proof " + std::to_string(c / (n_chains / (size_t)NP)) + " rejected: an opening does not end in its root");
    }
    lap("device: P2R rows + roots back");
    }   // (host mode)
    // the host tables up, then the machine's proof
    const ZeroedWords* host[N_CHIPS] = {nullptr, &ht.rs, nullptr, &ht.ts, &ht.q, &ht.op, &ht.sm, &ht.sc, &ht.evl};
    for (int c = 0; c < m.n; c++) if (host[c] && host[c]->size()) ZK_TRY(dev_h2d(ctx, dev[c], host[c]->data(), host[c]->size() * 4));
    zkhip_chip chips[N_CHIPS]{};
    for (int i = 0; i < m.n; i++) {
        const int c = m.order[i];
        chips[i].d_trace = (const uint32_t*)dev[c]; chips[i].ld = w_main[c]; chips[i].log_n = m.height[c]; chips[i].width = w_main[c]; chips[i].partner = -1;
    }
    std::vector<uint32_t> pv((size_t)NP * n_public);
    for (size_t i = 0; i < pv.size(); i++) pv[i] = public_values[i] % P;
    lap("upload: host tables");
    const int rc = zkhip_prove_machine_keyed(ctx, key, chips, m.progs, m.prog_words, m.tabs, m.tab_words, m.n, pv.data(), pv.size(), outer, proof, cap, len);
    lap("the machine's proof");
    recycle_later({ht.sc.release(), ht.op.release(), ht.rs.release(), ht.q.release(), ht.ts.release(), ht.sm.release(), ht.evl.release()});
    return rc;
}

// ---- the entries, with the inner AIR as the synthetic one (program == NULL: version-1 inner proofs) or as a constraint program (air mode:
// version-7 inner proofs, zkhip_prove_shard_air's)
static int sv_setup(zkhip_ctx* ctx, const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs,
                    const zkhip_params* outer, zkhip_machine_key** key, uint32_t vk[8]) {
    using namespace zk::rec;
    CHECK_CTX(ctx);
    if (!outer || !key || !vk) return fail(ZKHIP_ERR_INVALID, "shard_verifier_setup: null argument");
    Shape sh;
    ZK_TRY(make_shape(log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, sh, program, program_words));
    ZK_TRY(check_outer(sh, outer));
    const auto mp = machine_of(sh);
    const Machine& m = *mp;
    std::vector<uint32_t> pre[N_CHIPS];
    all_pre(sh, m, pre);
    size_t total = 0;
    for (int c = 0; c < N_CHIPS; c++) total += pre[c].size();
    void* d;
    ZK_TRY(ctx_reserve(ctx, S_REC_A, total * 4, &d));       // (staging: the key keeps its own copies)
    zkhip_chip chips[N_CHIPS]{};
    size_t at = 0;
    for (int i = 0; i < m.n; i++) {
        const int c = m.order[i];
        chips[i].log_n = m.height[c]; chips[i].width = m.pre_widths[i]; chips[i].ld = m.pre_widths[i]; chips[i].partner = -1;
        if (pre[c].empty()) continue;
        ZK_TRY(dev_h2d(ctx, (uint32_t*)d + at, pre[c].data(), pre[c].size() * 4));
        chips[i].d_trace = (const uint32_t*)d + at;
        at += pre[c].size();
    }
    return zkhip_machine_setup(ctx, chips, m.n, outer, key, vk);
}
// The same key WITHOUT a device (round 5; host_key.cpp): the preprocessed traces are host tables anyway -- their low-degree extensions and the
// mixed-height Poseidon2 commitment are computed on the host's cores, so that a verifier that owns no GPU derives the key of the shape it
// means by itself (the reference verifies on the CPU: sp1.rs:120).  Equal to the device's vk at every shape.
static int sv_key_host(const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs,
                       const zkhip_params* outer, uint32_t vk[8]) {
    using namespace zk::rec;
    try {
        if (!outer || !vk) return fail(ZKHIP_ERR_INVALID, "shard_verifier_key_host: null argument");
        Shape sh;
        ZK_TRY(make_shape(log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, sh, program, program_words));
        ZK_TRY(check_outer(sh, outer));
        const auto mp = machine_of(sh);
        const Machine& m = *mp;
        std::vector<uint32_t> pre[N_CHIPS];
        all_pre(sh, m, pre);
        const uint32_t* traces[N_CHIPS]; int32_t lns[N_CHIPS]; uint32_t pws[N_CHIPS];
        for (int i = 0; i < m.n; i++) {
            const int c = m.order[i];
            lns[i] = m.height[c]; pws[i] = m.pre_widths[i];
            traces[i] = pre[c].empty() ? nullptr : pre[c].data();
            if (pre[c].empty()) pws[i] = 0;
        }
        return zkhip_machine_key_host(traces, lns, pws, m.n, outer, vk);
    } catch (const std::bad_alloc&) {
        return fail(ZKHIP_ERR_NOMEM, "shard_verifier_key_host: out of host memory");
    }
}
static size_t sv_max_proofs(const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, const zkhip_params* outer) {
    using namespace zk::rec;
    Shape sh;
    if (make_shape(log_n, width, n_queries, inner_pow_bits, n_public, 1, sh, program, program_words) != ZKHIP_OK) return 0;
    size_t fit = ((size_t)1 << p2r_max_log_rows(outer)) / sh.p2_rows;
    if (fit > MAX_JOIN) fit = MAX_JOIN;
    const size_t fixed = sh.air ? 46 : 29;
    if (!sh.pub_rows.empty() && fit > (1024 - fixed) / sh.pub_rows.size()) fit = (1024 - fixed) / sh.pub_rows.size();      // (the transcript table's indicator columns)
    if (sh.air && fit > ((size_t)1 << MAX_LOG_ROWS) / sh.terms.size()) fit = ((size_t)1 << MAX_LOG_ROWS) / sh.terms.size();       // (the EVAL chip: one row per term and proof)
    return fit;
}
static size_t sv_proof_size(const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs,
                            const zkhip_params* outer) {
    using namespace zk::rec;
    Shape sh;
    if (!outer || make_shape(log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, sh, program, program_words) != ZKHIP_OK || check_outer(sh, outer) != ZKHIP_OK) return 0;
    const auto m = machine_of(sh);
    return zkhip_machine_proof_size_keyed(m->log_ns, m->widths, m->pre_widths, m->progs, m->prog_words, m->tabs, m->tab_words, m->n, outer, n_proofs * n_public);
}
static int sv_verify(const uint32_t* program, size_t program_words, const uint8_t* proof, size_t len, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits,
                     const uint32_t* public_values, size_t n_public, size_t n_proofs, const uint32_t vk[8], const zkhip_params* outer, int* reason) {
    using namespace zk::rec;
    Shape sh;
    if (!proof || !vk || !outer || (n_public && !public_values) || make_shape(log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, sh, program, program_words) != ZKHIP_OK) {
        if (reason) *reason = 1;
        return fail(ZKHIP_ERR_VERIFY, "verify_shard_recursive: bad arguments");
    }
    const auto m = machine_of(sh);
    std::vector<uint32_t> pv(n_proofs * n_public);
    for (size_t i = 0; i < pv.size(); i++) pv[i] = public_values[i];
    return zkhip_verify_machine_keyed(proof, len, m->log_ns, m->widths, m->pre_widths, vk, m->progs, m->prog_words, m->tabs, m->tab_words, m->n, pv.data(), pv.size(), outer, reason);
}
static size_t sv_describe(const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs, int which,
                          int kind, uint32_t* out, size_t cap_words, int* log_rows, uint32_t* main_width, uint32_t* pre_width) {
    using namespace zk::rec;
    Shape sh;
    if (which < 0 || kind < 0 || kind > 2 || make_shape(log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, sh, program, program_words) != ZKHIP_OK) return 0;
    const auto m = machine_of(sh);
    if (which >= m->n) return 0;
    if (log_rows) *log_rows = m->log_ns[which];
    if (main_width) *main_width = m->widths[which];
    if (pre_width) *pre_width = m->pre_widths[which];
    std::vector<uint32_t> pre;
    const std::vector<uint32_t>* src = kind == 0 ? &m->prog[which] : &m->tab[which];
    if (kind == 2) {
        const int c = m->order[which], h = m->height[c];
        switch (c) {
            case C_P2R: p2r_pre(sh, h, pre); break;
            case C_ROWSUM: rowsum_pre(sh, h, pre); break;
            case C_TS: ts_pre(sh, h, pre); break;
            case C_QUERY: query_pre(sh, h, pre); break;
            case C_OPENED: opened_pre(sh, h, pre); break;
            case C_SAMPLES: samples_pre_all(sh, h, pre); break;
            case C_SCALARS: scalars_pre(sh, h, pre); break;
            case C_EVAL: eval_pre(sh, h, pre); break;
            default: break;
        }
        for (uint32_t& v : pre) v = from_monty(v);
        src = &pre;
    }
    if (out && cap_words >= src->size()) std::memcpy(out, src->data(), src->size() * 4);
    return src->size();
}

extern "C" {

// the key of a SHAPE: the commitment to the eight chips' preprocessed columns -- no inner proof is involved
int zkhip_recursion_witnesses_on_host(int enable) { return zk::rec::g_witnesses_on_host.exchange(enable != 0 ? 1 : 0); }
int zkhip_shard_verifier_setup(zkhip_ctx* ctx, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs, const zkhip_params* outer,
                               zkhip_machine_key** key, uint32_t vk[8]) {
    return sv_setup(ctx, nullptr, 0, log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, outer, key, vk);
}
int zkhip_shard_verifier_key_host(int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs, const zkhip_params* outer, uint32_t vk[8]) {
    return sv_key_host(nullptr, 0, log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, outer, vk);
}
// the largest n_proofs one join takes for this inner shape: the Poseidon2 chip holds every permutation of every proof in at most 2^22 rows
// (2^21 when the outer proof's blowup is not 2)
size_t zkhip_shard_verifier_max_proofs(int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, const zkhip_params* outer) {
    return sv_max_proofs(nullptr, 0, log_n, width, n_queries, inner_pow_bits, n_public, outer);
}
size_t zkhip_shard_verifier_proof_size(int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs, const zkhip_params* outer) {
    return sv_proof_size(nullptr, 0, log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, outer);
}
int zkhip_prove_shard_verifier(zkhip_ctx* ctx, const zkhip_machine_key* key, const uint8_t* const* shard_proofs, const size_t* shard_proof_lens, size_t n_proofs, int log_n,
                               uint32_t width, const uint32_t* public_values, size_t n_public, const zkhip_params* inner, const zkhip_params* outer, uint8_t* proof, size_t cap,
                               size_t* len) {
    return shard_verifier_prove_impl(ctx, key, nullptr, 0, shard_proofs, shard_proof_lens, n_proofs, log_n, width, public_values, n_public, inner, outer, proof, cap, len);
}
// The compress stage over several joins: n_proofs / proofs_per_join joins of ONE shape (hence one key); join j verifies the shard proofs
// [j J, (j + 1) J) and is proven on devices[j mod n_devices], `in_flight_per_device` at a time on each -- the joins are independent units like the
// shards below them (no exchange step), and one join's host stretches (witness tables, transcript round trips) overlap another's
// kernels.  Every worker runs on a pooled context that keeps the shape's proving key (setup once per context and shape; every context arrives at
// the same vk).  Join j's proof goes to joined + j joined_stride (stride >= zkhip_shard_verifier_proof_size).  verify != 0: each join is checked by
// the host verifier on its worker's thread (sp1.rs:120), beside the other workers' kernels.
// (on_join, when given, runs on the worker's thread right after join j exists -- zkhip_prove_shard_tree fills the top's tables for it there; its
// failure is the join's)
extern "C++" int sv_prove_batch(const int* devices, int n_devices, const uint8_t* const* shard_proofs, const size_t* shard_proof_lens, size_t n_proofs,
                                size_t proofs_per_join, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public, const zkhip_params* inner,
                                const zkhip_params* outer, int in_flight_per_device, int verify, uint8_t* joined, size_t joined_stride, size_t* joined_lens,
                                uint32_t vk[8], const std::function<int(size_t, const uint8_t*, size_t)>* on_join) {
    if (!shard_proofs || !shard_proof_lens || !inner || !outer || !joined || !joined_lens || !vk || (n_public && !public_values))
        return fail(ZKHIP_ERR_INVALID, "prove_shard_verifier_batch: null argument");
    if (proofs_per_join == 0 || n_proofs == 0 || n_proofs % proofs_per_join != 0)
        return fail(ZKHIP_ERR_INVALID, "prove_shard_verifier_batch: the number of shard proofs must be a positive multiple of proofs_per_join");
    const size_t J = proofs_per_join, n_joins = n_proofs / J;
    if (n_joins > (size_t)1 << 20) return fail(ZKHIP_ERR_INVALID, "prove_shard_verifier_batch: too many joins");
    const size_t cap = zkhip_shard_verifier_proof_size(log_n, width, (size_t)inner->num_queries, inner->pow_bits, n_public, J, outer);
    if (cap == 0) return ZKHIP_ERR_INVALID;                       // (the message is the size query's)
    if (joined_stride < cap) return fail(ZKHIP_ERR_INVALID, "prove_shard_verifier_batch: joined_stride is below zkhip_shard_verifier_proof_size");
    for (size_t j = 0; j < n_joins; j++) joined_lens[j] = 0;
    std::vector<int> devs;
    ZK_TRY(resolve_devices(devices, n_devices, "prove_shard_verifier_batch", devs));
    const std::vector<uint64_t> sig = {(uint64_t)(uint32_t)log_n, width, (uint64_t)(uint32_t)inner->num_queries, (uint64_t)(uint32_t)inner->pow_bits, n_public, J,
                                       (uint64_t)(uint32_t)outer->log_blowup, (uint64_t)(uint32_t)outer->num_queries, (uint64_t)(uint32_t)outer->pow_bits,
                                       (uint64_t)(uint32_t)outer->log_fold, (uint64_t)(uint32_t)outer->log_final, (uint64_t)(uint32_t)outer->hash_width,
                                       zkhip_poseidon2_params_generation()};
    std::mutex mu;
    bool have_vk = false;
    std::memset(vk, 0, 32);
    std::vector<char> ran;
    auto run = [&](zkhip_ctx* ctx, int j) {
        int r = ZKHIP_OK;
        if (ctx->rec_key && ctx->rec_key_sig != sig) { (void)zkhip_ctx_sync(ctx); zkhip_machine_key_destroy(ctx->rec_key); ctx->rec_key = nullptr; }
        if (!ctx->rec_key) {
            r = zkhip_shard_verifier_setup(ctx, log_n, width, (size_t)inner->num_queries, inner->pow_bits, n_public, J, outer, &ctx->rec_key, ctx->rec_vk);
            if (r == ZKHIP_OK) ctx->rec_key_sig = sig;
            else ctx->rec_key = nullptr;
        }
        if (r == ZKHIP_OK) {
            std::lock_guard<std::mutex> lk(mu);
            if (!have_vk) { std::memcpy(vk, ctx->rec_vk, 32); have_vk = true; }
            else if (std::memcmp(vk, ctx->rec_vk, 32) != 0) r = fail(ZKHIP_ERR_INTERNAL, "prove_shard_verifier_batch: two contexts disagree about the key of the shape");
        }
        size_t len = 0;
        uint8_t* const out = joined + (size_t)j * joined_stride;
        const uint32_t* const pv = n_public ? public_values + (size_t)j * J * n_public : nullptr;
        if (r == ZKHIP_OK) r = zkhip_prove_shard_verifier(ctx, ctx->rec_key, shard_proofs + (size_t)j * J, shard_proof_lens + (size_t)j * J, J, log_n, width, pv, n_public,
                                                          inner, outer, out, joined_stride, &len);
        if (r == ZKHIP_OK && verify)
            r = zkhip_verify_shard_recursive(out, len, log_n, width, (size_t)inner->num_queries, inner->pow_bits, pv, n_public, J, ctx->rec_vk, outer, nullptr);
        if (r == ZKHIP_OK && on_join) r = (*on_join)((size_t)j, out, len);
        joined_lens[(size_t)j] = r == ZKHIP_OK ? len : 0;
        return r;
    };
    return deal_jobs(devs.data(), (int)devs.size(), (int)n_joins, in_flight_per_device, run, ran);          // (0: four)
}
int zkhip_prove_shard_verifier_batch(const int* devices, int n_devices, const uint8_t* const* shard_proofs, const size_t* shard_proof_lens, size_t n_proofs,
                                     size_t proofs_per_join, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public, const zkhip_params* inner,
                                     const zkhip_params* outer, int in_flight_per_device, int verify, uint8_t* joined, size_t joined_stride, size_t* joined_lens,
                                     uint32_t vk[8]) {
    return sv_prove_batch(devices, n_devices, shard_proofs, shard_proof_lens, n_proofs, proofs_per_join, log_n, width, public_values, n_public, inner, outer,
                          in_flight_per_device, verify, joined, joined_stride, joined_lens, vk, nullptr);
}
// The verifier of the outer proof: the shape of the inner proofs, THEIR public values (proof 0's, then proof 1's, ...), the key of the shape.
// No byte of an inner proof.
int zkhip_verify_shard_recursive(const uint8_t* proof, size_t len, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, const uint32_t* public_values,
                                 size_t n_public, size_t n_proofs, const uint32_t vk[8], const zkhip_params* outer, int* reason) {
    return sv_verify(nullptr, 0, proof, len, log_n, width, n_queries, inner_pow_bits, public_values, n_public, n_proofs, vk, outer, reason);
}
// the machine as data (tests compare with tests/recursion_air.py word for word): which = position (tallest chip first); kind 0 = the chip's
// program, 1 = its interaction table, 2 = its preprocessed trace (canonical words, row-major); *log_rows, *main_width, *pre_width describe the chip
size_t zkhip_shard_verifier_describe(int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs, int which, int kind, uint32_t* out,
                                     size_t cap_words, int* log_rows, uint32_t* main_width, uint32_t* pre_width) {
    return sv_describe(nullptr, 0, log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, which, kind, out, cap_words, log_rows, main_width, pre_width);
}

// ---- the same entries for inner proofs of a constraint PROGRAM (version 7: zkhip_prove_shard_air, zkhip_prove_sha256, the chained SHA-256 shards;
// SP1 shape, log_quotient_degree 1, a width that is a multiple of 8, terms of at most three factors): nine chips -- the EVAL chip evaluates the
// program at zeta, one row per term.  The key is a function of the shape AND the program.
int zkhip_shard_verifier_setup_air(zkhip_ctx* ctx, const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public,
                                   size_t n_proofs, const zkhip_params* outer, zkhip_machine_key** key, uint32_t vk[8]) {
    if (!program) return fail(ZKHIP_ERR_INVALID, "shard_verifier_setup_air: null program");
    return sv_setup(ctx, program, program_words, log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, outer, key, vk);
}
int zkhip_shard_verifier_key_host_air(const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs,
                                      const zkhip_params* outer, uint32_t vk[8]) {
    if (!program) return fail(ZKHIP_ERR_INVALID, "shard_verifier_key_host_air: null program");
    return sv_key_host(program, program_words, log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, outer, vk);
}
size_t zkhip_shard_verifier_max_proofs_air(const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public,
                                           const zkhip_params* outer) {
    return program ? sv_max_proofs(program, program_words, log_n, width, n_queries, inner_pow_bits, n_public, outer) : 0;
}
size_t zkhip_shard_verifier_proof_size_air(const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public,
                                           size_t n_proofs, const zkhip_params* outer) {
    return program ? sv_proof_size(program, program_words, log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, outer) : 0;
}
int zkhip_prove_shard_verifier_air(zkhip_ctx* ctx, const zkhip_machine_key* key, const uint32_t* program, size_t program_words, const uint8_t* const* shard_proofs,
                                   const size_t* shard_proof_lens, size_t n_proofs, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public,
                                   const zkhip_params* inner, const zkhip_params* outer, uint8_t* proof, size_t cap, size_t* len) {
    if (!program) return fail(ZKHIP_ERR_INVALID, "prove_shard_verifier_air: null program");
    return shard_verifier_prove_impl(ctx, key, program, program_words, shard_proofs, shard_proof_lens, n_proofs, log_n, width, public_values, n_public, inner, outer, proof, cap, len);
}
int zkhip_verify_shard_recursive_air(const uint32_t* program, size_t program_words, const uint8_t* proof, size_t len, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits,
                                     const uint32_t* public_values, size_t n_public, size_t n_proofs, const uint32_t vk[8], const zkhip_params* outer, int* reason) {
    if (!program) { if (reason) *reason = 1; return fail(ZKHIP_ERR_VERIFY, "verify_shard_recursive_air: null program"); }
    return sv_verify(program, program_words, proof, len, log_n, width, n_queries, inner_pow_bits, public_values, n_public, n_proofs, vk, outer, reason);
}
size_t zkhip_shard_verifier_describe_air(const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public,
                                         size_t n_proofs, int which, int kind, uint32_t* out, size_t cap_words, int* log_rows, uint32_t* main_width, uint32_t* pre_width) {
    return program ? sv_describe(program, program_words, log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, which, kind, out, cap_words, log_rows, main_width, pre_width) : 0;
}

}  // extern "C"
