// machine_verifier.inl -- included at the end of fri_chip.hip, behind shard_verifier.inl (it builds on that file's polynomial / table builders,
// on the fold chip, the SAMPLES chip and the Poseidon2 chip's row kernel).
//
// MACHINE MODE OF THE SHARD VERIFIER MACHINE (docs/RECURSION_NEXT.md): a keyed machine that checks WHOLE
// KEYED-MACHINE PROOFS (version 11: chips of mixed heights, each with its constraint program and its interaction table, most with
// preprocessed columns committed by a key) in-circuit -- everything zkhip_verify_machine_keyed checks: the transcript from the header (the
// entries, the digests, the inner key's root) on, gamma / beta, the permutation root and the cumulative sums, every chip's program and
// lookup constraints at zeta with the chip's OWN selectors and quotient, the four mixed-height commitments of every query (the key's
// tree, main, permutation, quotient: concatenated leaves, injection of the shorter matrices' rows), one reduced opening per height, FRI
// with the heights joining on the way down, proof of work.  With it the join's own output is joinable: a TREE of joins
// (crates/guest-prover-sp1/src/sp1.rs:116 core -> compress; RISC Zero lift -> join, prover.rs:90).
//
// Written AFTER its restatement: tests/recursion_machine.py is the executable design (rows checked in plain integers, buses balanced,
// proven by the oracle, a two-level tree on the CPU); this file produces the same programs, tables and preprocessed traces word for
// word (tests/test_recursion_machine_cpu.py) and the same proof bytes (tests/test_gpu_recursion_machine.py).  Ten chips; the comments
// of the restatement say what each constrains.
namespace zk {
namespace mrec {
namespace {
using namespace zk::rec;
using frichip::ALL; using frichip::TRANSITION;

constexpr uint32_t BUS_DG0 = 90, BUS_DG1 = 91, BUS_TC2 = 92, BUS_PW = 93, BUS_CS = 98, BUS_ACC = 100, BUS_KL = 101, BUS_ZH0 = 104, BUS_ZH1 = 105, BUS_KFA_M = 106,
                   BUS_YH0 = 107, BUS_YH1 = 108, BUS_AH0 = 109, BUS_AH1 = 110;
constexpr int MAX_INNER_CHIPS = 16;
constexpr uint32_t N_CHAL = 5, PWSPAN = 1u << 14;
enum Kind : int { K_EL, K_EN, K_TL, K_TN, K_PL, K_PN, K_Q, N_KINDS };
enum Tree : int { T_E, T_T, T_P, T_Q, N_TREES };
enum SrcKind : uint8_t { S_CONST, S_TROOT, S_PUB, S_PROOT, S_CUM, S_QROOT, S_OP, S_LROOT, S_FIN, S_WIT };
struct Src { uint8_t kind; uint32_t a, b; };
struct Inter { uint32_t sign; int mcol; uint32_t bus; std::vector<uint32_t> cols; };
struct Seg { int chip, kind; uint32_t pos, n, e; };
struct LeafSeg { int chip; uint32_t at, width; };
struct Leaf { std::vector<LeafSeg> segs; uint32_t words = 0; };
struct ETerm { int chip; uint32_t coeff, key[3], first; };
struct LSide { bool present = false; uint32_t sign = 0, mkey = 0, bus = 0; std::vector<uint32_t> vkeys; };
struct LRow { int chip; bool bnd; uint32_t phi[4], phin[4], sels[3]; LSide a, b; };

struct MShape {
    int C = 0, Q = 0, PB = 0, NPUB = 0, NP = 1, H = 0, R = 0;
    std::vector<int> ln, lh;
    std::vector<uint32_t> W, Pw, Wp, cols;
    std::vector<const uint32_t*> prog, tab;
    std::vector<size_t> prog_words, tab_words;
    std::vector<std::vector<Inter>> inter;
    std::vector<int> hs;                                    // distinct heights (log2 of LDE rows), tallest first
    std::vector<uint32_t> head; int HL = 0;
    uint32_t key_root[8];
    std::vector<Seg> segs; uint32_t NV = 0;
    uint32_t seg_pos[MAX_INNER_CHIPS][N_KINDS], seg_n[MAX_INNER_CHIPS][N_KINDS], seg_e[MAX_INNER_CHIPS][N_KINDS];
    std::vector<std::vector<Src>> plan;                     // rows 0 .. TP: what each absorbs
    int TG = 0, TPR = 0, TA = 0, TQ = 0, TO0 = 0, TF = 0, TL0 = 0, TP = 0, NS = 0, NT = 0, NTS = 0;
    std::vector<int> pub_rows;
    bool has_tree[N_TREES];
    std::vector<int> tree_chips[N_TREES], tree_hs[N_TREES];
    const std::vector<uint32_t>* tree_w[N_TREES];
    std::vector<uint32_t> w8;
    Leaf leaf[N_TREES][32];
    size_t fri_rows = 0, p2_fri0 = 0, p2_tree0[N_TREES], tree_rows[N_TREES], p2_rows = 0;
    uint32_t blk0[N_TREES][32], BLKSPAN = 0, TAGSPAN = 0, NTREES = 0, KSPAN = 0;
    std::vector<uint32_t> mult;
    std::vector<ETerm> terms; std::vector<size_t> term0, term1;
    std::vector<LRow> lrows; std::vector<size_t> lrow0, lrow1;
    std::vector<uint64_t> id;                               // what the machine cache is keyed by

    uint32_t npub_total() const { return (uint32_t)(NP * NPUB); }
    uint32_t ttag(int p, int T) const { return (uint32_t)p * TAGSPAN + (uint32_t)T; }
    uint32_t blk_tag(int p, int q, int t, int h, uint32_t b) const { return (uint32_t)p * TAGSPAN + (uint32_t)NT + (uint32_t)q * BLKSPAN + (blk0[t][h] - (uint32_t)NT) + b; }
    uint32_t dg_tag(int p, int q, int t, int h) const { return (uint32_t)(((p * Q + q) * 4 + t) * 32 + h); }
    uint32_t tree_id(int p, int layer) const { return (uint32_t)p * NTREES + (uint32_t)layer; }
    uint32_t tree_id_t(int p, int t) const { return (uint32_t)p * NTREES + (uint32_t)R + (uint32_t)t; }
    int absorbed(int T) const { return T <= TP ? (int)plan[(size_t)T].size() : 0; }
    bool has_h(int t, int h) const { for (int x : tree_hs[t]) if (x == h) return true; return false; }
    uint32_t key_op(int c, int kind, uint32_t col) const { return 1u + seg_pos[c][kind] + col; }
    uint32_t key_pub(uint32_t i) const { return 1u + NV + i; }
    uint32_t key_sel(int c, uint32_t which) const { return 1u + NV + (uint32_t)NPUB + 3u * (uint32_t)c + which; }
    uint32_t key_var(int c, uint32_t v) const {
        const uint32_t kind = v >> 30, idx = v & 0xFFFFu;
        if (kind == 2) return key_pub(idx);
        if (idx < Pw[(size_t)c]) return key_op(c, kind ? K_EN : K_EL, idx);
        return key_op(c, kind ? K_TN : K_TL, idx - Pw[(size_t)c]);
    }
    uint32_t acc_key(int p, int c, int stage) const { return (uint32_t)((p * C + c) * 2 + stage); }
    uint32_t pw_key(int p, int h, uint32_t e) const { return (uint32_t)(p * 32 + h) * PWSPAN + e; }
    uint32_t ah_key(int p, int q, int h) const { return (uint32_t)((p * Q + q) * 32 + h); }
    uint32_t qn_of(int p, int q, int t) const { return (uint32_t)((p * Q + q) * 32 + tree_hs[t][0]); }
};

// the duplex sponge of proof_common.h's challenger run on SOURCES: which sponge row absorbs what, behind which row every challenge is sampled
struct Plan {
    std::vector<std::vector<Src>> rows; std::vector<Src> pending; int ready = 0;
    void duplex() { rows.push_back(pending); pending.clear(); ready = 8; }
    void observe(const Src& s) { ready = 0; pending.push_back(s); if (pending.size() == 8) duplex(); }
    int sample(int* half) { if (!pending.empty() || ready < 4) duplex(); *half = ready == 8 ? 0 : 1; ready -= 4; return (int)rows.size() - 1; }
};

int make_mshape(const zkhip_machine_desc* d, size_t n_proofs, MShape& s) {
    if (!d || d->n_chips < 1 || d->n_chips > MAX_INNER_CHIPS || !d->log_ns || !d->widths || !d->pre_widths || !d->programs || !d->program_words || !d->tables || !d->table_words)
        return fail(ZKHIP_ERR_INVALID, "machine verifier: null description, or more than 16 chips");
    if (d->num_queries < 1 || d->num_queries > 1024 || d->pow_bits < 0 || d->pow_bits > 28 || d->n_public > 4096 || n_proofs < 1 || n_proofs > 64 || n_proofs * (size_t)d->n_public > 16384)
        return fail(ZKHIP_ERR_INVALID, "machine verifier: 1 .. 1024 queries, 0 .. 28 proof-of-work bits, at most 4096 public values per proof and 16384 in all, 1 .. 64 proofs");
    const int C = d->n_chips;
    s.C = C; s.Q = d->num_queries; s.PB = d->pow_bits; s.NPUB = (int)d->n_public; s.NP = (int)n_proofs;
    std::memcpy(s.key_root, d->key_root, 32);
    s.ln.resize(C); s.lh.resize(C); s.W.resize(C); s.Pw.resize(C); s.Wp.resize(C); s.cols.resize(C); s.prog.resize(C); s.tab.resize(C); s.prog_words.resize(C); s.tab_words.resize(C);
    s.inter.assign(C, {}); s.w8.assign(C, 8u);
    for (int c = 0; c < C; c++) {
        s.ln[c] = d->log_ns[c]; s.lh[c] = s.ln[c] + 1; s.W[c] = d->widths[c]; s.Pw[c] = d->pre_widths[c];
        s.prog[c] = d->programs[c]; s.prog_words[c] = d->program_words[c]; s.tab[c] = d->tables[c]; s.tab_words[c] = d->table_words[c];
        if (s.ln[c] < 5 || s.ln[c] > 21 || (c && s.ln[c] > s.ln[c - 1]) || s.W[c] == 0 || s.W[c] % 4 || s.Pw[c] % 4 || s.W[c] + s.Pw[c] > 1024 || !s.prog[c] || !s.tab[c])
            return fail(ZKHIP_ERR_INVALID, "machine verifier: chips tallest first, 2^5 .. 2^21 rows, widths in multiples of 4, every chip with a program and an interaction table");
        AirView av; LookupView lv;
        if (!air_validate(s.prog[c], s.prog_words[c], s.W[c] + s.Pw[c], (uint32_t)s.NPUB, &av) || av.lqd != 1 || !lookup_validate(s.tab[c], s.tab_words[c], s.W[c] + s.Pw[c], &lv))
            return fail(ZKHIP_ERR_INVALID, "machine verifier: a chip's program (log_quotient_degree 1) or interaction table is not valid for its combined width");
        size_t p = 3;
        for (uint32_t i = 0; i < lv.ni; i++) {
            Inter it{s.tab[c][p], s.tab[c][p + 1] == 0xFFFFFFFFu ? -1 : (int)s.tab[c][p + 1], s.tab[c][p + 2], {}};
            const uint32_t nv = s.tab[c][p + 3];
            for (uint32_t v = 0; v < nv; v++) it.cols.push_back(s.tab[c][p + 4 + v]);
            p += 4 + nv;
            s.inter[c].push_back(it);
        }
        s.cols[c] = (lv.ni + 1) / 2; s.Wp[c] = 4 * (s.cols[c] + 1);
    }
    s.H = s.lh[0]; s.R = s.ln[0];
    s.hs.clear();
    for (int c = 0; c < C; c++) if (s.hs.empty() || s.hs.back() != s.lh[c]) s.hs.push_back(s.lh[c]);
    // ---- the header (verifier.cpp, chips_transcript_init): six words, the chips' entries, the programs' digests, the tables', the inner key's root
    s.head = {11u, (uint32_t)C, 1u, (uint32_t)s.Q, (uint32_t)s.PB, (uint32_t)s.NPUB};
    for (int c = 0; c < C; c++) { s.head.push_back((uint32_t)s.ln[c]); s.head.push_back(s.W[c]); s.head.push_back(1u); s.head.push_back((uint32_t)s.inter[c].size()); s.head.push_back(s.Pw[c]); }
    for (int c = 0; c < C; c++) { AirView av; air_validate(s.prog[c], s.prog_words[c], s.W[c] + s.Pw[c], (uint32_t)s.NPUB, &av); uint32_t dg[8]; air_digest(av, dg); s.head.insert(s.head.end(), dg, dg + 8); }
    for (int c = 0; c < C; c++) { AirView tv; tv.w = s.tab[c]; tv.words = s.tab_words[c]; uint32_t dg[8]; air_digest(tv, dg); s.head.insert(s.head.end(), dg, dg + 8); }      // (the same sponge over the table's words)
    s.head.insert(s.head.end(), s.key_root, s.key_root + 8);
    s.HL = (int)s.head.size();
    // ---- the opened-value stream
    s.segs.clear();
    uint32_t pos = 0, e = 0;
    for (int c = 0; c < C; c++) {
        if (c && s.lh[c] != s.lh[c - 1]) e = 0;
        const uint32_t lens[N_KINDS] = {s.Pw[c], s.Pw[c], s.W[c], s.W[c], s.Wp[c], s.Wp[c], 8u};
        for (int k = 0; k < N_KINDS; k++) {
            s.segs.push_back(Seg{c, k, pos, lens[k], e});
            s.seg_pos[c][k] = pos; s.seg_n[c][k] = lens[k]; s.seg_e[c][k] = e;
            pos += lens[k]; e += lens[k];
        }
    }
    s.NV = pos;
    // ---- the transcript
    Plan pl;
    for (uint32_t v : s.head) pl.observe(Src{S_CONST, v, 0});
    for (uint32_t j = 0; j < 8; j++) pl.observe(Src{S_TROOT, j, 0});
    for (int i = 0; i < s.NPUB; i++) pl.observe(Src{S_PUB, (uint32_t)i, 0});
    int half;
    s.TG = pl.sample(&half); (void)pl.sample(&half);
    for (uint32_t j = 0; j < 8; j++) pl.observe(Src{S_PROOT, j, 0});
    for (int c = 0; c < C; c++) for (uint32_t j = 0; j < 4; j++) pl.observe(Src{S_CUM, (uint32_t)c, j});
    s.TA = pl.sample(&half); s.TPR = s.TG + 1;
    for (uint32_t j = 0; j < 8; j++) pl.observe(Src{S_QROOT, j, 0});
    s.TQ = pl.sample(&half);
    for (uint32_t i = 0; i < 4 * s.NV; i++) pl.observe(Src{S_OP, i, 0});
    s.TF = pl.sample(&half); s.TO0 = s.TQ + 1;
    for (int l = 0; l < s.R; l++) { for (uint32_t j = 0; j < 8; j++) pl.observe(Src{S_LROOT, (uint32_t)l, j}); (void)pl.sample(&half); }
    s.TL0 = s.TF + 1;
    for (uint32_t j = 0; j < 4; j++) pl.observe(Src{S_FIN, j, 0});
    pl.observe(Src{S_WIT, 0, 0});
    s.TP = pl.sample(&half);
    if (s.TP != s.TL0 + s.R || s.TF - s.TO0 + 1 != (int)(s.NV / 2)) return fail(ZKHIP_ERR_INTERNAL, "machine verifier: transcript plan");
    s.plan = pl.rows;
    s.NS = (int)frichip::sample_rows((size_t)s.Q); s.NT = s.TP + s.NS; s.NTS = s.TP + 1;
    s.pub_rows.clear();
    for (int i = 0; i < s.NPUB; i++) { const int r = (s.HL + 8 + i) / 8; if (s.pub_rows.empty() || s.pub_rows.back() != r) s.pub_rows.push_back(r); }
    // ---- the four commitments
    s.tree_w[T_E] = &s.Pw; s.tree_w[T_T] = &s.W; s.tree_w[T_P] = &s.Wp; s.tree_w[T_Q] = &s.w8;
    for (int t = 0; t < N_TREES; t++) {
        s.tree_chips[t].clear(); s.tree_hs[t].clear();
        for (int c = 0; c < C; c++) if ((*s.tree_w[t])[c]) { s.tree_chips[t].push_back(c); if (s.tree_hs[t].empty() || s.tree_hs[t].back() != s.lh[c]) s.tree_hs[t].push_back(s.lh[c]); }
        s.has_tree[t] = !s.tree_chips[t].empty();
        for (int h = 0; h < 32; h++) { s.leaf[t][h].segs.clear(); s.leaf[t][h].words = 0; }
        for (int c : s.tree_chips[t]) { Leaf& lf = s.leaf[t][s.lh[c]]; lf.segs.push_back(LeafSeg{c, lf.words, (*s.tree_w[t])[c]}); lf.words += (*s.tree_w[t])[c]; }
    }
    if (!s.has_tree[T_E]) return fail(ZKHIP_ERR_INVALID, "machine verifier: a keyed machine has preprocessed columns");
    // ---- P2R rows
    s.fri_rows = (size_t)s.R + (size_t)s.R * (size_t)(s.R + 1) / 2;
    s.p2_fri0 = (size_t)s.NT;
    size_t at = s.p2_fri0 + (size_t)s.Q * s.fri_rows;
    for (int t = 0; t < N_TREES; t++) {
        if (!s.has_tree[t]) continue;
        size_t per = (size_t)s.tree_hs[t][0] + (s.tree_hs[t].size() - 1);
        for (int h : s.tree_hs[t]) per += (s.leaf[t][h].words + 7) / 8;
        s.p2_tree0[t] = at; s.tree_rows[t] = per;
        at += (size_t)s.Q * per;
    }
    s.p2_rows = at;
    uint32_t n = (uint32_t)s.NT;
    for (int t = 0; t < N_TREES; t++) if (s.has_tree[t]) for (int h : s.tree_hs[t]) { s.blk0[t][h] = n; n += (s.leaf[t][h].words + 7) / 8; }
    s.BLKSPAN = n - (uint32_t)s.NT; s.TAGSPAN = (uint32_t)s.NT + (uint32_t)s.Q * s.BLKSPAN; s.NTREES = (uint32_t)s.R + 4u;
    if (lg((size_t)s.NP * s.p2_rows) > P2R_MAX_LOG_ROWS) return fail(ZKHIP_ERR_INVALID, "machine verifier: the Poseidon2 chip would need more than 2^22 rows");
    // ---- what is read where: the EVAL chip's terms, the LOGUP chip's rows, the multiplicities of the value bus
    s.KSPAN = 1u + s.NV + (uint32_t)s.NPUB + 3u * (uint32_t)C;
    s.mult.assign(s.KSPAN, 0u);
    s.terms.clear();
    for (int c = 0; c < C; c++) {
        const uint32_t* prog = s.prog[c];
        size_t p = 6;
        for (uint32_t k = 0; k < prog[3]; k++) {
            const uint32_t sel = prog[p], nt = prog[p + 1];
            p += 2;
            if (nt == 0) s.terms.push_back(ETerm{c, 0u, {0u, 0u, 0u}, 1u});
            for (uint32_t t = 0; t < nt; t++) {
                ETerm et{c, prog[p], {0u, 0u, 0u}, t == 0 ? 1u : 0u};
                const uint32_t dgr = prog[p + 1];
                p += 2;
                if (dgr + (sel ? 1u : 0u) > 3) return fail(ZKHIP_ERR_INVALID, "machine verifier: a term of a chip's program has more than three factors");
                uint32_t nf = 0;
                for (uint32_t j = 0; j < dgr; j++) et.key[nf++] = s.key_var(c, prog[p++]);
                if (sel) et.key[nf++] = s.key_sel(c, sel - 1u);
                s.terms.push_back(et);
            }
        }
    }
    s.term0.assign(C, 0); s.term1.assign(C, 0);
    for (size_t i = s.terms.size(); i-- > 0;) s.term0[s.terms[i].chip] = i;
    for (size_t i = 0; i < s.terms.size(); i++) s.term1[s.terms[i].chip] = i;
    for (const ETerm& t : s.terms) for (int j = 0; j < 3; j++) s.mult[t.key[j]]++;
    s.lrows.clear();
    for (int c = 0; c < C; c++) {
        const std::vector<Inter>& its = s.inter[c];
        for (uint32_t j = 0; j < s.cols[c]; j++) {
            LRow row{}; row.chip = c; row.bnd = false;
            for (uint32_t k = 0; k < 4; k++) { row.phi[k] = s.key_op(c, K_PL, 4 * j + k); row.phin[k] = s.key_op(c, K_PN, 4 * j + k); }
            for (int side = 0; side < 2; side++) {
                const size_t i = 2 * j + (size_t)side;
                LSide& sd = side ? row.b : row.a;
                if (i >= its.size()) continue;
                sd.present = true; sd.sign = its[i].sign ? P - 1 : 1u; sd.mkey = its[i].mcol < 0 ? 0u : s.key_var(c, (uint32_t)its[i].mcol); sd.bus = its[i].bus;
                for (uint32_t x : its[i].cols) sd.vkeys.push_back(s.key_var(c, x));
            }
            s.lrows.push_back(row);
        }
        LRow bnd{}; bnd.chip = c; bnd.bnd = true;
        for (uint32_t k = 0; k < 4; k++) { bnd.phi[k] = s.key_op(c, K_PL, 4 * s.cols[c] + k); bnd.phin[k] = s.key_op(c, K_PN, 4 * s.cols[c] + k); }
        for (uint32_t w = 0; w < 3; w++) bnd.sels[w] = s.key_sel(c, w);
        s.lrows.push_back(bnd);
    }
    for (const LRow& row : s.lrows) {
        for (int k = 0; k < 4; k++) { s.mult[row.phi[k]]++; s.mult[row.phin[k]]++; }
        if (row.bnd) for (int k = 0; k < 3; k++) s.mult[row.sels[k]]++;
        for (const LSide* sd : {&row.a, &row.b}) if (sd->present) { s.mult[sd->mkey]++; for (uint32_t k : sd->vkeys) s.mult[k]++; }
    }
    for (int c = 0; c < C; c++) for (uint32_t k = 0; k < 8; k++) s.mult[s.key_op(c, K_Q, k)]++;
    s.lrow0.assign(C, 0); s.lrow1.assign(C, 0);
    for (size_t i = s.lrows.size(); i-- > 0;) s.lrow0[s.lrows[i].chip] = i;
    for (size_t i = 0; i < s.lrows.size(); i++) s.lrow1[s.lrows[i].chip] = i;
    if (s.terms.size() > ((size_t)1 << 20) || (size_t)s.NP * s.terms.size() > ((size_t)1 << MAX_LOG_ROWS))
        return fail(ZKHIP_ERR_INVALID, "machine verifier: too many program terms for the EVAL chip");
    if ((uint64_t)s.NP * s.KSPAN >= P / 2 || (uint64_t)s.NP * 32 * PWSPAN >= P / 2 || s.NV >= PWSPAN) return fail(ZKHIP_ERR_INVALID, "machine verifier: key space");
    // the transcript table: one indicator column per (proof, sponge row with public values)
    if (64u + (size_t)s.NP * s.pub_rows.size() > 1024) return fail(ZKHIP_ERR_INVALID, "machine verifier: too many proofs x public values for the transcript table");
    s.id = {(uint64_t)C, (uint64_t)s.Q, (uint64_t)s.PB, (uint64_t)s.NPUB, (uint64_t)s.NP};
    for (uint32_t v : s.head) s.id.push_back(v);
    return ZKHIP_OK;
}

// ============================================================================================================ P2R
constexpr uint32_t P2_PRE = 28, P2_MAIN = p2chip::R_WIDTH;
constexpr uint32_t PP_SS = 0, PP_SPG = 1, PP_CH = 2, PP_END = 3, PP_K = 4, PP_RIN = 12, PP_TAG = 13, PP_SROOT = 14, PP_TREE = 15, PP_SCH = 16, PP_SSMP = 17, PP_QIDX = 18,
                   PP_QN = 19, PP_RPAIR = 20, PP_RIN1 = 21, PP_SDG = 22, PP_RDG = 23, PP_CHN = 24, PP_SCH2 = 25, PP_DTAG = 26, PP_HALF = 27;
std::vector<uint32_t> p2r_program(const MShape& sh) {
    using namespace p2chip;
    const uint32_t M0 = P2_PRE, IN_ = M0 + IN, OUT = M0 + oute(7), D_ = M0 + D, BIT_ = M0 + BIT, KP_ = M0 + R_KP;
    Cons c;
    c.b.body = permutation_body(M0, &c.b.count);
    for (uint32_t j = 0; j < 8; j++) c.add(ALL, padd(padd(pv(D_ + j), pneg(pv(IN_ + j))), padd(pmul(pv(BIT_), pv(IN_ + j)), pneg(pmul(pv(BIT_), pv(IN_ + 8 + j))))));
    c.add(ALL, padd(pmul(pv(BIT_), pv(BIT_)), pneg(pv(BIT_))));
    c.add(ALL, pmul(padd(padd(pv(PP_SS), pv(PP_SPG)), pv(PP_RDG)), pv(BIT_)));
    for (uint32_t j = 0; j < 8; j++) c.add(ALL, pmul(pv(PP_SS), pv(IN_ + 8 + j)));
    for (uint32_t j = 0; j < 4; j++) c.add(ALL, pmul(pv(PP_HALF), pv(IN_ + 4 + j)));
    for (uint32_t j = 0; j < 8; j++) c.add(TRANSITION, pmul(pv(PP_SPG, true), padd(pv(IN_ + 8 + j, true), pneg(pv(OUT + 8 + j)))));
    for (uint32_t j = 0; j < 8; j++) c.add(TRANSITION, pmul(padd(pv(PP_CH, true), pv(PP_CHN, true)), padd(pv(D_ + j, true), pneg(pv(OUT + j)))));
    for (uint32_t j = 0; j < 8; j++) c.add(TRANSITION, pmul(pv(PP_K + j, true), padd(pv(IN_ + j, true), pneg(pv(OUT + j)))));
    c.add(TRANSITION, pmul(pv(PP_CH, true), padd(padd(pv(KP_), pscale(pv(KP_, true), P - 2)), pneg(pv(BIT_)))));
    c.add(TRANSITION, pmul(pv(PP_CHN, true), padd(pv(KP_), pneg(pv(KP_, true)))));
    c.add(ALL, pmul(pv(PP_END), padd(pv(KP_), pneg(pv(BIT_)))));
    return c.program(P2_PRE + P2_MAIN, sh.npub_total());
}
std::vector<uint32_t> p2r_table() {
    using namespace p2chip;
    const uint32_t M0 = P2_PRE, o = M0 + oute(7), IN_ = M0 + IN, KP_ = M0 + R_KP;
    Tab t;
    t.add5(RECV, PP_RIN, BUS_IN0, PP_TAG, IN_); t.add5(RECV, PP_RIN1, BUS_IN1, PP_TAG, IN_ + 4);
    t.add(RECV, PP_RPAIR, BUS_E0, {PP_TREE, KP_, IN_, IN_ + 1, IN_ + 2, IN_ + 3}); t.add(RECV, PP_RPAIR, BUS_E1, {PP_TREE, KP_, IN_ + 4, IN_ + 5, IN_ + 6, IN_ + 7});
    t.add5(SEND, PP_SROOT, BUS_R0, PP_TREE, o); t.add5(SEND, PP_SROOT, BUS_R1, PP_TREE, o + 4);
    t.add(SEND, PP_SCH, BUS_TC, {PP_TAG, o + 7, o + 6, o + 5, o + 4}); t.add(SEND, PP_SCH2, BUS_TC2, {PP_TAG, o + 3, o + 2, o + 1, o});
    t.add(SEND, PP_SSMP, BUS_S0, {PP_TAG, o + 7, o + 6, o + 5, o + 4}); t.add(SEND, PP_SSMP, BUS_S1, {PP_TAG, o + 3, o + 2, o + 1, o});
    t.add(RECV, PP_QIDX, BUS_QI, {PP_QN, KP_});
    t.add5(SEND, PP_SDG, BUS_DG0, PP_DTAG, o); t.add5(SEND, PP_SDG, BUS_DG1, PP_DTAG, o + 4);
    t.add5(RECV, PP_RDG, BUS_DG0, PP_DTAG, IN_ + 8); t.add5(RECV, PP_RDG, BUS_DG1, PP_DTAG, IN_ + 12);
    return t.w;
}
bool challenge_row(const MShape& sh, int T) { return T == sh.TG || T == sh.TA || T == sh.TQ || T == sh.TF || (T >= sh.TL0 && T < sh.TP); }
// the heights of a tree in the order their sponge rows stand: the shorter ones first (their digests travel), the tallest's last
std::vector<int> sponge_order(const MShape& sh, int t) { std::vector<int> o(sh.tree_hs[t].begin() + 1, sh.tree_hs[t].end()); o.push_back(sh.tree_hs[t][0]); return o; }
void p2r_pre(const MShape& sh, int log_rows, std::vector<uint32_t>& t) {
    t.assign((size_t)P2_PRE << log_rows, 0u);
    for (int p = 0; p < sh.NP; p++) {
        auto row = [&](size_t r) { return t.data() + (size_t)P2_PRE * ((size_t)p * sh.p2_rows + r); };
        for (int T = 0; T < sh.NT; T++) {
            uint32_t* r = row((size_t)T);
            const int k = sh.absorbed(T);
            if (T == 0) r[PP_SS] = 1;
            else { r[PP_SPG] = 1; for (int j = k; j < 8; j++) r[PP_K + j] = 1; }
            r[PP_TAG] = sh.ttag(p, T);
            if (k) r[PP_RIN] = r[PP_RIN1] = 1;
            if (challenge_row(sh, T)) r[PP_SCH] = 1;
            if (T == sh.TG) r[PP_SCH2] = 1;
            if (T >= sh.TP) r[PP_SSMP] = 1;
        }
        size_t at = sh.p2_fri0;
        for (int q = 0; q < sh.Q; q++)
            for (int l = 0; l < sh.R; l++) {
                uint32_t* r = row(at++);
                r[PP_SS] = 1; r[PP_RPAIR] = 1; r[PP_TREE] = sh.tree_id(p, l);
                const int depth = sh.H - (l + 1);
                for (int lvl = 0; lvl < depth; lvl++) {
                    r = row(at++);
                    r[PP_CH] = 1; r[PP_TREE] = sh.tree_id(p, l);
                    if (lvl == depth - 1) r[PP_END] = r[PP_SROOT] = 1;
                }
            }
        for (int tr = 0; tr < N_TREES; tr++) {
            if (!sh.has_tree[tr]) continue;
            const std::vector<int>& hs = sh.tree_hs[tr];
            const std::vector<int> so = sponge_order(sh, tr);
            for (int q = 0; q < sh.Q; q++) {
                for (int h : so) {
                    const uint32_t words = sh.leaf[tr][h].words, nb = (words + 7) / 8;
                    for (uint32_t b = 0; b < nb; b++) {
                        uint32_t* r = row(at++);
                        r[b == 0 ? PP_SS : PP_SPG] = 1;
                        const uint32_t k = words - 8 * b < 8 ? words - 8 * b : 8u;
                        if (b == 0 && k < 8) r[PP_HALF] = 1;
                        else for (uint32_t j = k; j < 8; j++) r[PP_K + j] = 1;
                        r[PP_RIN] = 1; r[PP_RIN1] = k == 8 ? 1u : 0u; r[PP_TAG] = sh.blk_tag(p, q, tr, h, b);
                        if (b == nb - 1 && h != hs[0]) { r[PP_SDG] = 1; r[PP_DTAG] = sh.dg_tag(p, q, tr, h); }
                    }
                }
                bool chained = false;
                for (int lvl = 0; lvl < hs[0]; lvl++) {
                    uint32_t* r = row(at++);
                    r[chained ? PP_CHN : PP_CH] = 1; r[PP_TREE] = sh.tree_id_t(p, tr);
                    chained = false;
                    if (lvl == 0) { r[PP_QIDX] = 1; r[PP_QN] = sh.qn_of(p, q, tr); }
                    if (lvl == hs[0] - 1) r[PP_END] = r[PP_SROOT] = 1;
                    const int h = hs[0] - lvl - 1;
                    if (h != hs[0] && sh.has_h(tr, h)) {
                        r = row(at++);
                        r[PP_CH] = 1; r[PP_RDG] = 1; r[PP_DTAG] = sh.dg_tag(p, q, tr, h); r[PP_TREE] = sh.tree_id_t(p, tr);
                        chained = true;
                    }
                }
            }
        }
    }
    monty_all(t);
}

// ============================================================================================================ TS
struct TsCols { uint32_t T, ACT, NSEND, CF, CV, IND0, IP, NROOT, NTR, TREE, HASCH, NBETA, NSC, KIND, NFIN, PT, HASCH2, NSC2, KIND2, CK0, CM0, CK1, CM1, PK, PM, Z, pre, W, TR, CH, CH2; };
constexpr uint32_t TS_MAIN = 24;
TsCols ts_cols(const MShape& sh) {
    TsCols c{};
    uint32_t n = 0;
    auto take = [&](uint32_t w) { const uint32_t at = n; n += w; return at; };
    c.T = take(1); c.ACT = take(1); c.NSEND = take(1); c.CF = take(8); c.CV = take(8); c.IND0 = take(1); c.IP = take((uint32_t)(sh.NP * (int)sh.pub_rows.size()));
    c.NROOT = take(1); c.NTR = take(1); c.TREE = take(1); c.HASCH = take(1); c.NBETA = take(1); c.NSC = take(1); c.KIND = take(1); c.NFIN = take(1); c.PT = take(1);
    c.HASCH2 = take(1); c.NSC2 = take(1); c.KIND2 = take(1); c.CK0 = take(1); c.CM0 = take(1); c.CK1 = take(1); c.CM1 = take(1); c.PK = take(8); c.PM = take(8); c.Z = take(1);
    c.pre = rup4(n);
    c.W = c.pre; c.TR = c.pre + 8; c.CH = c.pre + 16; c.CH2 = c.pre + 20;
    return c;
}
int pub_row_index(const MShape& sh, int row) { for (size_t i = 0; i < sh.pub_rows.size(); i++) if (sh.pub_rows[i] == row) return (int)i; return -1; }
std::vector<uint32_t> ts_program(const MShape& sh) {
    const TsCols c = ts_cols(sh);
    Cons k;
    for (uint32_t j = 0; j < 8; j++) k.add(ALL, pmul(pv(c.CF + j), padd(pv(c.W + j), pneg(pv(c.CV + j)))));
    const int npr = (int)sh.pub_rows.size();
    for (int p = 0; p < sh.NP; p++)
        for (int i = 0; i < sh.NPUB; i++) {
            const int pos = sh.HL + 8 + i;
            k.add(ALL, pmul(pv(c.IP + (uint32_t)(p * npr + pub_row_index(sh, pos / 8))), padd(pv(c.W + (uint32_t)(pos % 8)), pneg(ppub((uint32_t)(p * sh.NPUB + i))))));
        }
    const uint32_t o = (uint32_t)(sh.HL % 8);
    for (uint32_t j = 0; j < 8 - o; j++) k.add(ALL, pmul(pv(c.IND0), padd(pv(c.W + o + j), pneg(pv(c.TR + j)))));
    for (uint32_t j = 0; j < o; j++) k.add(TRANSITION, pmul(pv(c.IND0), padd(pv(c.W + j, true), pneg(pv(c.TR + 8 - o + j)))));
    return k.program(c.pre + TS_MAIN, sh.npub_total());
}
std::vector<uint32_t> ts_table(const MShape& sh) {
    const TsCols c = ts_cols(sh);
    Tab t;
    t.add5(SEND, c.NSEND, BUS_IN0, c.T, c.W); t.add5(SEND, c.NSEND, BUS_IN1, c.T, c.W + 4);
    t.add5(RECV, c.HASCH, BUS_TC, c.T, c.CH); t.add5(RECV, c.HASCH2, BUS_TC2, c.T, c.CH2);
    t.add5(SEND, c.NBETA, BUS_BETA, c.TREE, c.CH);
    t.add5(SEND, c.NSC, BUS_SC, c.KIND, c.CH); t.add5(SEND, c.NSC2, BUS_SC, c.KIND2, c.CH2);
    t.add5(RECV, c.NROOT, BUS_R0, c.TREE, c.W); t.add5(RECV, c.NROOT, BUS_R1, c.TREE, c.W + 4);
    t.add5(RECV, c.NTR, BUS_R0, c.TREE, c.TR); t.add5(RECV, c.NTR, BUS_R1, c.TREE, c.TR + 4);
    t.add5(RECV, c.NFIN, BUS_FIN, c.PT, c.W);
    t.add5(SEND, c.CM0, BUS_CS, c.CK0, c.W); t.add5(SEND, c.CM1, BUS_CS, c.CK1, c.W + 4);
    for (uint32_t j = 0; j < 8; j++) t.add(SEND, c.PM + j, BUS_VAL, {c.PK + j, c.W + j, c.Z, c.Z, c.Z});
    return t.w;
}
void ts_pre(const MShape& sh, int log_rows, std::vector<uint32_t>& t) {
    const TsCols c = ts_cols(sh);
    t.assign((size_t)c.pre << log_rows, 0u);
    const int npr = (int)sh.pub_rows.size();
    for (int p = 0; p < sh.NP; p++)
        for (int T = 0; T < sh.NTS; T++) {
            uint32_t* r = t.data() + (size_t)c.pre * ((size_t)p * (size_t)sh.NTS + (size_t)T);
            r[c.T] = sh.ttag(p, T); r[c.ACT] = 1; r[c.NSEND] = (T >= sh.TO0 && T <= sh.TF) ? 2u : 1u;
            const std::vector<Src>& pl = sh.plan[(size_t)T];
            for (uint32_t j = 0; j < pl.size(); j++) {
                if (pl[j].kind == S_CONST) { r[c.CF + j] = 1; r[c.CV + j] = pl[j].a; }
                if (pl[j].kind == S_PUB) { r[c.PK + j] = (uint32_t)p * sh.KSPAN + sh.key_pub(pl[j].a); r[c.PM + j] = sh.mult[sh.key_pub(pl[j].a)]; }
                if (pl[j].kind == S_CUM && pl[j].b == 0) { r[j == 0 ? c.CK0 : c.CK1] = (uint32_t)(p * sh.C) + pl[j].a; r[j == 0 ? c.CM0 : c.CM1] = 2; }
            }
            const int pr = pub_row_index(sh, T);
            if (pr >= 0) r[c.IP + (uint32_t)(p * npr + pr)] = 1;
            if (T == sh.HL / 8) { r[c.IND0] = 1; r[c.NTR] = (uint32_t)sh.Q; r[c.TREE] = sh.tree_id_t(p, T_T); }
            if (T == sh.TPR) { r[c.NROOT] = (uint32_t)sh.Q; r[c.TREE] = sh.tree_id_t(p, T_P); }
            if (T == sh.TQ) { r[c.NROOT] = (uint32_t)sh.Q; r[c.TREE] = sh.tree_id_t(p, T_Q); }
            if (T >= sh.TL0 && T < sh.TP) { r[c.NROOT] = (uint32_t)sh.Q; r[c.TREE] = sh.tree_id(p, T - sh.TL0); r[c.NBETA] = (uint32_t)sh.Q; }
            const int kinds[4] = {sh.TA, sh.TQ, sh.TF, sh.TG};
            for (int kind = 0; kind < 4; kind++) if (T == kinds[kind]) { r[c.NSC] = 1; r[c.KIND] = N_CHAL * (uint32_t)p + (uint32_t)kind; }
            if (T == sh.TG) { r[c.HASCH2] = 1; r[c.NSC2] = 1; r[c.KIND2] = N_CHAL * (uint32_t)p + 4u; }
            if (challenge_row(sh, T)) r[c.HASCH] = 1;
            if (T == sh.TP) { r[c.NFIN] = (uint32_t)sh.Q; r[c.PT] = (uint32_t)p * sh.NTREES; }
        }
    monty_all(t);
}

// ============================================================================================================ EVAL
constexpr uint32_t EV_PRE = 12, EP_COEF = 0, EP_K0 = 1, EP_FIRSTC = 4, EP_ACT = 5, EP_LAST = 6, EP_PID = 7, EP_NFC = 8, EP_CFIRST = 9, EP_CKEY = 10;
constexpr uint32_t EV_F0 = 0, EV_M = 12, EV_TV = 16, EV_ACCIN = 20, EV_ACCO = 24, EV_ALPHA = 28, EV_MAIN = 32;
std::vector<uint32_t> eval_program(const MShape& sh) {
    const uint32_t M0 = EV_PRE;
    Cons c;
    const EE f0 = ev(M0 + EV_F0), f1 = ev(M0 + EV_F0 + 4), f2 = ev(M0 + EV_F0 + 8), mm = ev(M0 + EV_M), tv = ev(M0 + EV_TV), ai = ev(M0 + EV_ACCIN), ao = ev(M0 + EV_ACCO), al = ev(M0 + EV_ALPHA);
    c.ext(ALL, esub(mm, emul(f0, f1)));
    c.ext(ALL, esub(tv, egate(pv(EP_COEF), emul(mm, f2))));
    c.ext(ALL, esub(ao, eadd(ai, egate(pv(EP_FIRSTC), esub(emul(ai, al), ai)), tv)));
    c.ext(TRANSITION, egate(pv(EP_NFC, true), esub(ev(M0 + EV_ACCIN, true), ao)));
    c.ext(ALL, egate(pv(EP_CFIRST), ai));
    return c.program(EV_PRE + EV_MAIN, sh.npub_total());
}
std::vector<uint32_t> eval_table() {
    const uint32_t M0 = EV_PRE;
    Tab t;
    for (uint32_t j = 0; j < 3; j++) t.add5(RECV, EP_ACT, BUS_VAL, EP_K0 + j, M0 + EV_F0 + 4u * j);
    t.add5(RECV, EP_ACT, BUS_EA, EP_PID, M0 + EV_ALPHA);
    t.add5(SEND, EP_LAST, BUS_ACC, EP_CKEY, M0 + EV_ACCO);
    return t.w;
}
void eval_pre(const MShape& sh, int log_rows, std::vector<uint32_t>& t) {
    t.assign((size_t)EV_PRE << log_rows, 0u);
    const size_t nt = sh.terms.size();
    for (int p = 0; p < sh.NP; p++)
        for (size_t i = 0; i < nt; i++) {
            uint32_t* r = t.data() + (size_t)EV_PRE * ((size_t)p * nt + i);
            const ETerm& e = sh.terms[i];
            r[EP_COEF] = e.coeff; r[EP_FIRSTC] = e.first; r[EP_ACT] = 1; r[EP_PID] = (uint32_t)p;
            for (int j = 0; j < 3; j++) r[EP_K0 + j] = (uint32_t)p * sh.KSPAN + e.key[j];
            r[EP_LAST] = i == sh.term1[e.chip] ? 1u : 0u; r[EP_CFIRST] = i == sh.term0[e.chip] ? 1u : 0u;
            r[EP_NFC] = i != sh.term0[e.chip] ? 1u : 0u; r[EP_CKEY] = sh.acc_key(p, e.chip, 0);
        }
    monty_all(t);
}

// ============================================================================================================ LOGUP
constexpr uint32_t LG_PRE = 64;
constexpr uint32_t LP_ACT = 0, LP_ISP = 1, LP_ISB = 2, LP_NF = 3, LP_PFIRST = 4, LP_NFP = 5, LP_PID = 6, LP_KIN = 7, LP_KOUT = 8, LP_CFIRST = 9, LP_KCUM = 10, LP_BUSA = 11, LP_BUSB = 12,
                   LP_SA = 13, LP_SB = 14, LP_HASB = 15, LP_NOB = 16, LP_KMA = 17, LP_KMB = 18, LP_MB = 19, LP_KA = 20, LP_FA = 28, LP_KB = 36, LP_FB = 44, LP_KP = 52, LP_KPN = 56;
struct LgCols { uint32_t VA, VB, MA, MB, PH, PN, PHI, PHIN, DA, DB, CST, ACCIN, ACCO, U1, U2, SUML, SUMN, CUM, ALPHA, GAMMA, BP, end; };
LgCols lg_cols() {
    LgCols c{};
    uint32_t n = LG_PRE;
    auto take = [&](uint32_t w) { const uint32_t at = n; n += w; return at; };
    c.VA = take(32); c.VB = take(32); c.MA = take(4); c.MB = take(4); c.PH = take(16); c.PN = take(16); c.PHI = take(4); c.PHIN = take(4); c.DA = take(4); c.DB = take(4); c.CST = take(4);
    c.ACCIN = take(4); c.ACCO = take(4); c.U1 = take(4); c.U2 = take(4); c.SUML = take(4); c.SUMN = take(4); c.CUM = take(4); c.ALPHA = take(4); c.GAMMA = take(4); c.BP = take(32);
    c.end = n;
    return c;
}
const uint32_t LG_MAIN = lg_cols().end - LG_PRE;
EE from_columns(uint32_t first) {       // sum_k X^k column_k
    EE out;
    for (uint32_t k = 0; k < 4; k++) {
        const EE e = ev(first + 4 * k);
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t j = i + k;
            const Poly t = j >= 4 ? pscale(e[i], EXT_W) : e[i];
            out[j % 4].insert(out[j % 4].end(), t.begin(), t.end());
        }
    }
    return out;
}
std::vector<uint32_t> logup_program(const MShape& sh) {
    const LgCols m = lg_cols();
    Cons c;
    const uint32_t consts[2] = {m.ALPHA, m.GAMMA};
    for (uint32_t col : consts) c.ext(TRANSITION, egate(pv(LP_NFP, true), esub(ev(col, true), ev(col))));
    for (uint32_t k = 0; k < 8; k++) c.ext(TRANSITION, egate(pv(LP_NFP, true), esub(ev(m.BP + 4 * k, true), ev(m.BP + 4 * k))));
    for (uint32_t k = 1; k < 8; k++) c.ext(ALL, esub(ev(m.BP + 4 * k), emul(ev(m.BP + 4 * (k - 1)), ev(m.BP))));
    c.ext(ALL, esub(ev(m.PHI), from_columns(m.PH)));
    c.ext(ALL, esub(ev(m.PHIN), from_columns(m.PN)));
    for (int side = 0; side < 2; side++) {
        const uint32_t V_ = side ? m.VB : m.VA, D_ = side ? m.DB : m.DA, BUS_ = side ? LP_BUSB : LP_BUSA, F_ = side ? LP_FB : LP_FA;
        EE d = eadd(egate(side ? pv(LP_HASB) : pv(LP_ISP), ev(m.GAMMA)), eb(pv(BUS_)));
        if (side) d = eadd(d, eb(pv(LP_NOB)));
        for (uint32_t t = 0; t < 8; t++) d = eadd(d, egate(pv(F_ + t), emul(ev(m.BP + 4 * t), ev(V_ + 4 * t))));
        c.ext(ALL, esub(ev(D_), d));
    }
    const EE ma = egate(pv(LP_SA), ev(m.MA)), mb = egate(pv(LP_SB), ev(m.MB));
    c.ext(ALL, esub(ev(m.CST), esub(emul(emul(ev(m.PHI), ev(m.DA)), ev(m.DB)), eadd(emul(ma, ev(m.DB)), emul(mb, ev(m.DA))))));
    const EE al = ev(m.ALPHA);
    c.ext(ALL, esub(ev(m.U1), eadd(emul(ev(m.ACCIN), al), emul(ev(m.VA), esub(ev(m.PHI), ev(m.SUML))))));
    c.ext(ALL, esub(ev(m.U2), eadd(emul(ev(m.U1), al), emul(ev(m.VA + 8), esub(esub(ev(m.PHIN), ev(m.PHI)), ev(m.SUMN))))));
    c.ext(ALL, esub(ev(m.ACCO), eadd(egate(pv(LP_ISP), eadd(emul(ev(m.ACCIN), al), ev(m.CST))),
                                      egate(pv(LP_ISB), eadd(emul(ev(m.U2), al), emul(ev(m.VA + 4), esub(ev(m.PHI), ev(m.CUM))))))));
    c.ext(TRANSITION, egate(pv(LP_NF, true), esub(ev(m.ACCIN, true), ev(m.ACCO))));
    c.ext(TRANSITION, egate(pv(LP_NF, true), esub(ev(m.SUML, true), eadd(ev(m.SUML), ev(m.PHI)))));
    c.ext(TRANSITION, egate(pv(LP_NF, true), esub(ev(m.SUMN, true), eadd(ev(m.SUMN), ev(m.PHIN)))));
    c.ext(ALL, egate(pv(LP_CFIRST), ev(m.SUML)));
    c.ext(ALL, egate(pv(LP_CFIRST), ev(m.SUMN)));
    return c.program(LG_PRE + LG_MAIN, sh.npub_total());
}
std::vector<uint32_t> logup_table() {
    const LgCols m = lg_cols();
    Tab t;
    for (uint32_t k = 0; k < 8; k++) t.add5(RECV, LP_FA + k, BUS_VAL, LP_KA + k, m.VA + 4 * k);
    for (uint32_t k = 0; k < 8; k++) t.add5(RECV, LP_FB + k, BUS_VAL, LP_KB + k, m.VB + 4 * k);
    t.add5(RECV, LP_ISP, BUS_VAL, LP_KMA, m.MA); t.add5(RECV, LP_MB, BUS_VAL, LP_KMB, m.MB);
    for (uint32_t k = 0; k < 4; k++) t.add5(RECV, LP_ACT, BUS_VAL, LP_KP + k, m.PH + 4 * k);
    for (uint32_t k = 0; k < 4; k++) t.add5(RECV, LP_ACT, BUS_VAL, LP_KPN + k, m.PN + 4 * k);
    t.add5(RECV, LP_ISB, BUS_CS, LP_KCUM, m.CUM); t.add5(RECV, LP_CFIRST, BUS_ACC, LP_KIN, m.ACCIN); t.add5(SEND, LP_ISB, BUS_ACC, LP_KOUT, m.ACCO);
    t.add5(RECV, LP_PFIRST, BUS_KL, LP_PID, m.ALPHA); t.add5(RECV, LP_PFIRST, BUS_KL + 1, LP_PID, m.GAMMA); t.add5(RECV, LP_PFIRST, BUS_KL + 2, LP_PID, m.BP);
    return t.w;
}
void logup_pre(const MShape& sh, int log_rows, std::vector<uint32_t>& t) {
    t.assign((size_t)LG_PRE << log_rows, 0u);
    const size_t n = sh.lrows.size();
    for (int p = 0; p < sh.NP; p++) {
        const uint32_t K = (uint32_t)p * sh.KSPAN;
        for (size_t i = 0; i < n; i++) {
            uint32_t* r = t.data() + (size_t)LG_PRE * ((size_t)p * n + i);
            const LRow& row = sh.lrows[i];
            const int c = row.chip;
            r[LP_ACT] = 1; r[LP_PID] = (uint32_t)p;
            r[LP_PFIRST] = i == 0; r[LP_NFP] = i != 0;
            r[LP_CFIRST] = i == sh.lrow0[c]; r[LP_NF] = i != sh.lrow0[c];
            r[LP_KIN] = sh.acc_key(p, c, 0); r[LP_KOUT] = sh.acc_key(p, c, 1); r[LP_KCUM] = (uint32_t)(p * sh.C + c);
            for (uint32_t k = 0; k < 4; k++) { r[LP_KP + k] = K + row.phi[k]; r[LP_KPN + k] = K + row.phin[k]; }
            if (row.bnd) {
                r[LP_ISB] = 1;
                for (uint32_t k = 0; k < 3; k++) { r[LP_KA + k] = K + row.sels[k]; r[LP_FA + k] = 1; }
                continue;
            }
            r[LP_ISP] = 1;
            r[LP_BUSA] = row.a.bus; r[LP_SA] = row.a.sign; r[LP_KMA] = K + row.a.mkey;
            for (size_t k = 0; k < row.a.vkeys.size(); k++) { r[LP_KA + k] = K + row.a.vkeys[k]; r[LP_FA + k] = 1; }
            if (row.b.present) {
                r[LP_HASB] = 1; r[LP_MB] = 1; r[LP_BUSB] = row.b.bus; r[LP_SB] = row.b.sign; r[LP_KMB] = K + row.b.mkey;
                for (size_t k = 0; k < row.b.vkeys.size(); k++) { r[LP_KB + k] = K + row.b.vkeys[k]; r[LP_FB + k] = 1; }
            } else r[LP_NOB] = 1;
        }
    }
    monty_all(t);
}

// ============================================================================================================ SCALARS
struct ScCols { uint32_t ACT, PFIRST, NFP, PID, KACC, KCUM, LASTC, OH, WINV, WN, ZA0, ZB0, ZA1, ZB1, KQZ, KSEL, MSEL, KONE, MONE, Z, KZH, MZH, MKR, TREE, KIND, MEA, MFA, pre,
                  ALPHA, ZETA, FA, GAMMA, BETA, ZP, ZN, INVF, INVT, SELF, SELL, SELT, ZNX, QZ, Q0, Q1, QUO, ACC, CUM, TOTIN, TOTO, KR, end; };
ScCols sc_cols(const MShape& sh) {
    ScCols c{};
    uint32_t n = 0;
    auto take = [&](uint32_t w) { const uint32_t at = n; n += w; return at; };
    c.ACT = take(1); c.PFIRST = take(1); c.NFP = take(1); c.PID = take(1); c.KACC = take(1); c.KCUM = take(1); c.LASTC = take(1); c.OH = take((uint32_t)sh.R + 1); c.WINV = take(1); c.WN = take(1);
    c.ZA0 = take(1); c.ZB0 = take(1); c.ZA1 = take(1); c.ZB1 = take(1); c.KQZ = take(8); c.KSEL = take(3); c.MSEL = take(3); c.KONE = take(1); c.MONE = take(1); c.Z = take(1); c.KZH = take(1);
    c.MZH = take(1); c.MKR = take(1); c.TREE = take(1); c.KIND = take(N_CHAL); c.MEA = take(1); c.MFA = take(1);
    c.pre = rup4(n);
    n = c.pre;
    c.ALPHA = take(4); c.ZETA = take(4); c.FA = take(4); c.GAMMA = take(4); c.BETA = take(4); c.ZP = take(4 * (uint32_t)sh.R); c.ZN = take(4); c.INVF = take(4); c.INVT = take(4); c.SELF = take(4);
    c.SELL = take(4); c.SELT = take(4); c.ZNX = take(4); c.QZ = take(32); c.Q0 = take(4); c.Q1 = take(4); c.QUO = take(4); c.ACC = take(4); c.CUM = take(4); c.TOTIN = take(4); c.TOTO = take(4);
    c.KR = take(8);
    c.end = n;
    return c;
}
std::vector<uint32_t> scalars_program(const MShape& sh) {
    const ScCols c = sc_cols(sh);
    Cons k;
    const uint32_t consts[5] = {c.ALPHA, c.ZETA, c.FA, c.GAMMA, c.BETA};
    for (uint32_t col : consts) k.ext(TRANSITION, egate(pv(c.NFP, true), esub(ev(col, true), ev(col))));
    std::vector<EE> zp{ev(c.ZETA)};
    for (int i = 0; i < sh.R; i++) zp.push_back(ev(c.ZP + 4 * (uint32_t)i));
    for (int i = 0; i < sh.R; i++) k.ext(ALL, esub(zp[(size_t)i + 1], emul(zp[(size_t)i], zp[(size_t)i])));
    EE zn;
    for (int i = 0; i <= sh.R; i++) zn = eadd(zn, egate(pv(c.OH + (uint32_t)i), zp[(size_t)i]));
    k.ext(ALL, esub(ev(c.ZN), zn));
    const EE zh = esub(ev(c.ZN), eb(pv(c.ACT))), one = eb(pv(c.ACT));
    k.ext(ALL, esub(emul(esub(ev(c.ZETA), one), ev(c.INVF)), one));
    k.ext(ALL, esub(emul(esub(ev(c.ZETA), eb(pv(c.WINV))), ev(c.INVT)), one));
    k.ext(ALL, esub(ev(c.SELF), emul(zh, ev(c.INVF))));
    k.ext(ALL, esub(ev(c.SELL), emul(zh, ev(c.INVT))));
    k.ext(ALL, esub(ev(c.SELT), esub(ev(c.ZETA), eb(pv(c.WINV)))));
    k.ext(ALL, esub(ev(c.ZNX), egate(pv(c.WN), ev(c.ZETA))));
    for (uint32_t h = 0; h < 2; h++) {
        EE q;
        for (uint32_t j = 0; j < 4; j++) {
            const EE v = ev(c.QZ + 16 * h + 4 * j);
            for (uint32_t i = 0; i < 4; i++) { const Poly t = i + j >= 4 ? pscale(v[i], EXT_W) : v[i]; q[(i + j) % 4].insert(q[(i + j) % 4].end(), t.begin(), t.end()); }
        }
        k.ext(ALL, esub(ev(h ? c.Q1 : c.Q0), q));
    }
    const EE zps0 = eadd(egate(pv(c.ZA0), ev(c.ZN)), eb(pv(c.ZB0))), zps1 = eadd(egate(pv(c.ZA1), ev(c.ZN)), eb(pv(c.ZB1)));
    k.ext(ALL, esub(ev(c.QUO), eadd(emul(zps0, ev(c.Q0)), emul(zps1, ev(c.Q1)))));
    k.ext(ALL, esub(ev(c.ACC), emul(ev(c.QUO), zh)));
    k.ext(ALL, egate(pv(c.PFIRST), ev(c.TOTIN)));
    k.ext(ALL, esub(ev(c.TOTO), eadd(ev(c.TOTIN), ev(c.CUM))));
    k.ext(TRANSITION, egate(pv(c.NFP, true), esub(ev(c.TOTIN, true), ev(c.TOTO))));
    k.ext(ALL, egate(pv(c.LASTC), ev(c.TOTO)));
    for (uint32_t j = 0; j < 8; j++) k.add(ALL, pmul(pv(c.PFIRST), padd(pv(c.KR + j), pc(P - sh.key_root[j]))));
    return k.program(c.pre + rup4(c.end - c.pre), sh.npub_total());
}
std::vector<uint32_t> scalars_table(const MShape& sh) {
    const ScCols c = sc_cols(sh);
    Tab t;
    t.add5(RECV, c.ACT, BUS_ACC, c.KACC, c.ACC); t.add5(RECV, c.ACT, BUS_CS, c.KCUM, c.CUM);
    for (uint32_t k = 0; k < 8; k++) t.add5(RECV, c.ACT, BUS_VAL, c.KQZ + k, c.QZ + 4 * k);
    const uint32_t sels[3] = {c.SELF, c.SELL, c.SELT};
    for (uint32_t i = 0; i < 3; i++) t.add5(SEND, c.MSEL + i, BUS_VAL, c.KSEL + i, sels[i]);
    t.add(SEND, c.MONE, BUS_VAL, {c.KONE, c.PFIRST, c.Z, c.Z, c.Z});
    const uint32_t chal[N_CHAL] = {c.ALPHA, c.ZETA, c.FA, c.GAMMA, c.BETA};
    for (uint32_t k = 0; k < N_CHAL; k++) t.add5(RECV, c.PFIRST, BUS_SC, c.KIND + k, chal[k]);
    t.add5(SEND, c.MEA, BUS_EA, c.PID, c.ALPHA);
    const uint32_t kl[3] = {c.ALPHA, c.GAMMA, c.BETA};
    for (uint32_t k = 0; k < 3; k++) t.add5(SEND, c.PFIRST, BUS_KL + k, c.PID, kl[k]);
    t.add5(SEND, c.MFA, BUS_KFA_M, c.PID, c.FA);
    t.add5(SEND, c.MZH, BUS_ZH0, c.KZH, c.ZETA); t.add5(SEND, c.MZH, BUS_ZH1, c.KZH, c.ZNX);
    t.add5(RECV, c.MKR, BUS_R0, c.TREE, c.KR); t.add5(RECV, c.MKR, BUS_R1, c.TREE, c.KR + 4);
    return t.w;
}
// zps_k(zeta) = A_k zeta^N + B_k for the two quotient chunks of a trace domain of 2^ln rows (canonical)
void chunk_weights(int ln, uint32_t out[4]) {
    const uint64_t N = (uint64_t)1 << ln;
    const uint32_t wq = two_adic_generator(ln + 1);
    const uint32_t sN[2] = {fpow(MONTY_GEN, N), fpow(fmul(MONTY_GEN, wq), N)};
    for (int k = 0; k < 2; k++) {
        const uint32_t sjn_inv = finv(sN[1 - k]);
        const uint32_t den_inv = finv(fsub(fmul(sN[k], sjn_inv), MONTY_R1));
        out[2 * k] = from_monty(fmul(sjn_inv, den_inv)); out[2 * k + 1] = from_monty(fsub(0u, den_inv));
    }
}
void scalars_pre(const MShape& sh, int log_rows, std::vector<uint32_t>& t) {
    const ScCols c = sc_cols(sh);
    t.assign((size_t)c.pre << log_rows, 0u);
    for (int p = 0; p < sh.NP; p++) {
        const uint32_t K = (uint32_t)p * sh.KSPAN;
        for (int ch = 0; ch < sh.C; ch++) {
            uint32_t* r = t.data() + (size_t)c.pre * ((size_t)p * (size_t)sh.C + (size_t)ch);
            const int ln = sh.ln[ch];
            const uint32_t wN = two_adic_generator(ln);
            r[c.ACT] = 1; r[c.PID] = (uint32_t)p; r[c.PFIRST] = ch == 0; r[c.NFP] = ch != 0; r[c.LASTC] = ch == sh.C - 1;
            r[c.KACC] = sh.acc_key(p, ch, 1); r[c.KCUM] = (uint32_t)(p * sh.C + ch);
            r[c.OH + (uint32_t)ln] = 1;
            r[c.WINV] = from_monty(finv(wN)); r[c.WN] = from_monty(wN);
            uint32_t zw[4];
            chunk_weights(ln, zw);
            r[c.ZA0] = zw[0]; r[c.ZB0] = zw[1]; r[c.ZA1] = zw[2]; r[c.ZB1] = zw[3];
            for (uint32_t k = 0; k < 8; k++) r[c.KQZ + k] = K + sh.key_op(ch, K_Q, k);
            for (uint32_t i = 0; i < 3; i++) { r[c.KSEL + i] = K + sh.key_sel(ch, i); r[c.MSEL + i] = sh.mult[sh.key_sel(ch, i)]; }
            if (ch == 0) {
                r[c.KONE] = K; r[c.MONE] = sh.mult[0];
                r[c.MEA] = (uint32_t)sh.terms.size(); r[c.MFA] = 2;
                for (uint32_t k = 0; k < N_CHAL; k++) r[c.KIND + k] = N_CHAL * (uint32_t)p + k;
                r[c.MKR] = (uint32_t)sh.Q; r[c.TREE] = sh.tree_id_t(p, T_E);
            }
            if (ch == 0 || sh.lh[ch] != sh.lh[ch - 1]) { r[c.KZH] = (uint32_t)(p * 32 + sh.lh[ch]); r[c.MZH] = (uint32_t)sh.Q; }
        }
    }
    monty_all(t);
}

// ============================================================================================================ OPENED (the stream)
constexpr uint32_t OS_PRE = 24, OS_ACT = 0, OS_TAG = 1, OS_PFIRST = 2, OS_NFP = 3, OS_PID = 4, OS_ISN = 5, OS_NISN = 6, OS_RST = 7, OS_NRST = 8, OS_K0 = 9, OS_M0 = 10, OS_K1 = 11, OS_M1 = 12,
                   OS_KPW = 13, OS_MPW = 14, OS_KYH = 15, OS_MYH = 16;
constexpr uint32_t OS_W = 0, OS_FA = 8, OS_FA2 = 12, OS_PW = 16, OS_M = 20, OS_YZIN = 24, OS_YNIN = 28, OS_YZO = 32, OS_YNO = 36, OS_PWN = 40, OS_MAIN = 44;
struct StreamRow { int chip, kind; uint32_t col; int h; uint32_t e; bool first, last; };
std::vector<StreamRow> stream_rows(const MShape& sh) {
    std::vector<StreamRow> out;
    for (const Seg& s : sh.segs) for (uint32_t j = 0; j < s.n; j += 2) out.push_back(StreamRow{s.chip, s.kind, j, sh.lh[s.chip], s.e + j, false, false});
    for (size_t i = 0; i < out.size(); i++) { out[i].first = i == 0 || out[i - 1].h != out[i].h; out[i].last = i + 1 == out.size() || out[i + 1].h != out[i].h; }
    return out;
}
bool is_next_kind(int kind) { return kind == K_EN || kind == K_TN || kind == K_PN; }
std::vector<uint32_t> opened_program(const MShape& sh) {
    const uint32_t M0 = OS_PRE;
    Cons c;
    const EE v0 = ev(M0 + OS_W), v1 = ev(M0 + OS_W + 4), fa = ev(M0 + OS_FA), fa2 = ev(M0 + OS_FA2), pw = ev(M0 + OS_PW), mm = ev(M0 + OS_M);
    const uint32_t consts[2] = {OS_FA, OS_FA2};
    for (uint32_t col : consts) c.ext(TRANSITION, egate(pv(OS_NFP, true), esub(ev(M0 + col, true), ev(M0 + col))));
    c.ext(ALL, esub(fa2, emul(fa, fa)));
    c.ext(ALL, esub(mm, eadd(v0, emul(fa, v1))));
    c.ext(ALL, egate(pv(OS_RST), esub(pw, ec(1))));
    c.ext(ALL, esub(ev(M0 + OS_PWN), emul(pw, fa2)));
    c.ext(TRANSITION, egate(pv(OS_NRST, true), esub(ev(M0 + OS_PW, true), ev(M0 + OS_PWN))));
    c.ext(ALL, esub(ev(M0 + OS_YZO), eadd(ev(M0 + OS_YZIN), egate(pv(OS_NISN), emul(pw, mm)))));
    c.ext(ALL, esub(ev(M0 + OS_YNO), eadd(ev(M0 + OS_YNIN), egate(pv(OS_ISN), emul(pw, mm)))));
    c.ext(TRANSITION, egate(pv(OS_NRST, true), esub(ev(M0 + OS_YZIN, true), ev(M0 + OS_YZO))));
    c.ext(TRANSITION, egate(pv(OS_NRST, true), esub(ev(M0 + OS_YNIN, true), ev(M0 + OS_YNO))));
    c.ext(ALL, egate(pv(OS_RST), ev(M0 + OS_YZIN)));
    c.ext(ALL, egate(pv(OS_RST), ev(M0 + OS_YNIN)));
    return c.program(OS_PRE + OS_MAIN, sh.npub_total());
}
std::vector<uint32_t> opened_table() {
    const uint32_t M0 = OS_PRE, W = M0 + OS_W;
    Tab t;
    t.add5(RECV, OS_ACT, BUS_IN0, OS_TAG, W); t.add5(RECV, OS_ACT, BUS_IN1, OS_TAG, W + 4);
    t.add5(SEND, OS_M0, BUS_VAL, OS_K0, W); t.add5(SEND, OS_M1, BUS_VAL, OS_K1, W + 4);
    t.add5(SEND, OS_MPW, BUS_PW, OS_KPW, M0 + OS_PW);
    t.add5(SEND, OS_MYH, BUS_YH0, OS_KYH, M0 + OS_YZO); t.add5(SEND, OS_MYH, BUS_YH1, OS_KYH, M0 + OS_YNO);
    t.add5(RECV, OS_PFIRST, BUS_KFA_M, OS_PID, M0 + OS_FA);
    return t.w;
}
// how many ROWSUM rows (per query) weight a segment with fa^e of height h
std::map<std::pair<int, uint32_t>, uint32_t> pw_uses(const MShape& sh) {
    std::map<std::pair<int, uint32_t>, uint32_t> uses;
    const int kz[N_TREES] = {K_EL, K_TL, K_PL, K_Q}, kn[N_TREES] = {K_EN, K_TN, K_PN, -1};
    for (int c = 0; c < sh.C; c++)
        for (int tr = 0; tr < N_TREES; tr++) {
            if ((*sh.tree_w[tr])[c] == 0) continue;
            uses[{sh.lh[c], sh.seg_e[c][kz[tr]]}]++;
            if (kn[tr] >= 0) uses[{sh.lh[c], sh.seg_e[c][kn[tr]]}]++;
        }
    return uses;
}
void opened_pre(const MShape& sh, int log_rows, std::vector<uint32_t>& t) {
    t.assign((size_t)OS_PRE << log_rows, 0u);
    const std::vector<StreamRow> rows = stream_rows(sh);
    const auto uses = pw_uses(sh);
    const size_t n = rows.size();
    for (int p = 0; p < sh.NP; p++) {
        const uint32_t K = (uint32_t)p * sh.KSPAN;
        for (size_t i = 0; i < n; i++) {
            uint32_t* r = t.data() + (size_t)OS_PRE * ((size_t)p * n + i);
            const StreamRow& s = rows[i];
            const uint32_t isn = is_next_kind(s.kind) ? 1u : 0u;
            r[OS_ACT] = 1; r[OS_TAG] = sh.ttag(p, sh.TO0 + (int)i); r[OS_PFIRST] = i == 0; r[OS_NFP] = i != 0; r[OS_PID] = (uint32_t)p;
            r[OS_ISN] = isn; r[OS_NISN] = 1 - isn; r[OS_RST] = s.first; r[OS_NRST] = !s.first;
            const uint32_t k0 = sh.key_op(s.chip, s.kind, s.col);
            r[OS_K0] = K + k0; r[OS_M0] = sh.mult[k0]; r[OS_K1] = K + k0 + 1; r[OS_M1] = sh.mult[k0 + 1];
            const auto it = uses.find({s.h, s.e});
            if (it != uses.end()) { r[OS_KPW] = sh.pw_key(p, s.h, s.e); r[OS_MPW] = (uint32_t)sh.Q * it->second; }
            if (s.last) { r[OS_KYH] = (uint32_t)(p * 32 + s.h); r[OS_MYH] = (uint32_t)sh.Q; }
        }
    }
    monty_all(t);
}

// ============================================================================================================ ROWSUM
constexpr uint32_t RS_PRE = 24, RP_TAG = 0, RP_ACT = 1, RP_RIN1 = 2, RP_HALF = 3, RP_NR7 = 4, RP_NR3 = 5, RP_F0 = 6, RP_F4 = 7, RP_N0 = 8, RP_N4 = 9, RP_KZ0 = 10, RP_KN0 = 11, RP_KZ4 = 12,
                   RP_KN4 = 13, RP_GFIRST = 14, RP_NG = 15, RP_GLAST = 16, RP_KAH = 17, RP_PFIRST = 18, RP_NFP = 19, RP_PID = 20;
constexpr uint32_t RS_V = 0, RS_ACCIN = 8, RS_T = 12, RS_FA = 44, RS_KZ0 = 48, RS_KN0 = 52, RS_KZ4 = 56, RS_KN4 = 60, RS_AZIN = 64, RS_ANIN = 68, RS_AZO = 72, RS_ANO = 76, RS_MAIN = 80;
struct RsRow { int q, h, tr; uint32_t b, k; int start[2]; bool r7, r3; };      // start[0] / start[1]: the chip whose segment STARTS at word 0 / 4 (-1: none)
std::vector<RsRow> rowsum_rows(const MShape& sh) {
    std::vector<RsRow> out;
    for (int q = 0; q < sh.Q; q++)
        for (int h : sh.hs)
            for (int tr = 0; tr < N_TREES; tr++) {
                if (!sh.has_tree[tr] || !sh.has_h(tr, h)) continue;
                const Leaf& lf = sh.leaf[tr][h];
                auto starts_at = [&](uint32_t w) { for (const LeafSeg& s : lf.segs) if (s.at == w) return s.chip; return -1; };
                const uint32_t nb = (lf.words + 7) / 8;
                for (uint32_t b = nb; b-- > 0;) {
                    RsRow r{q, h, tr, b, lf.words - 8 * b < 8 ? lf.words - 8 * b : 8u, {starts_at(8 * b), starts_at(8 * b + 4)}, b == nb - 1 || starts_at(8 * b + 8) >= 0, starts_at(8 * b + 4) >= 0};
                    out.push_back(r);
                }
            }
    return out;
}
std::vector<uint32_t> rowsum_program(const MShape& sh) {
    const uint32_t M0 = RS_PRE;
    Cons c;
    const EE fa = ev(M0 + RS_FA);
    c.ext(TRANSITION, egate(pv(RP_NFP, true), esub(ev(M0 + RS_FA, true), fa)));
    EE prev = ev(M0 + RS_ACCIN);
    for (int s = 7; s >= 0; s--) {
        const EE cur = ev(M0 + RS_T + 4 * (uint32_t)s);
        EE carried = emul(prev, fa);
        if (s == 7) carried = egate(pv(RP_NR7), carried);
        if (s == 3) carried = egate(pv(RP_NR3), carried);
        c.ext(ALL, esub(cur, eadd(carried, eb(pv(M0 + RS_V + (uint32_t)s)))));
        prev = cur;
    }
    c.ext(TRANSITION, egate(pv(RP_NR7, true), esub(ev(M0 + RS_ACCIN, true), ev(M0 + RS_T))));
    for (uint32_t j = 0; j < 4; j++) c.add(ALL, pmul(pv(RP_HALF), pv(M0 + RS_V + 4 + j)));
    const EE t0 = ev(M0 + RS_T), t4 = ev(M0 + RS_T + 16);
    c.ext(ALL, esub(ev(M0 + RS_AZO), eadd(ev(M0 + RS_AZIN), egate(pv(RP_F0), emul(ev(M0 + RS_KZ0), t0)), egate(pv(RP_F4), emul(ev(M0 + RS_KZ4), t4)))));
    c.ext(ALL, esub(ev(M0 + RS_ANO), eadd(ev(M0 + RS_ANIN), egate(pv(RP_N0), emul(ev(M0 + RS_KN0), t0)), egate(pv(RP_N4), emul(ev(M0 + RS_KN4), t4)))));
    c.ext(TRANSITION, egate(pv(RP_NG, true), esub(ev(M0 + RS_AZIN, true), ev(M0 + RS_AZO))));
    c.ext(TRANSITION, egate(pv(RP_NG, true), esub(ev(M0 + RS_ANIN, true), ev(M0 + RS_ANO))));
    c.ext(ALL, egate(pv(RP_GFIRST), ev(M0 + RS_AZIN)));
    c.ext(ALL, egate(pv(RP_GFIRST), ev(M0 + RS_ANIN)));
    return c.program(RS_PRE + RS_MAIN, sh.npub_total());
}
std::vector<uint32_t> rowsum_table() {
    const uint32_t M0 = RS_PRE, v = M0 + RS_V;
    Tab t;
    t.add5(SEND, RP_ACT, BUS_IN0, RP_TAG, v); t.add5(SEND, RP_RIN1, BUS_IN1, RP_TAG, v + 4);
    t.add5(RECV, RP_F0, BUS_PW, RP_KZ0, M0 + RS_KZ0); t.add5(RECV, RP_N0, BUS_PW, RP_KN0, M0 + RS_KN0);
    t.add5(RECV, RP_F4, BUS_PW, RP_KZ4, M0 + RS_KZ4); t.add5(RECV, RP_N4, BUS_PW, RP_KN4, M0 + RS_KN4);
    t.add5(SEND, RP_GLAST, BUS_AH0, RP_KAH, M0 + RS_AZO); t.add5(SEND, RP_GLAST, BUS_AH1, RP_KAH, M0 + RS_ANO);
    t.add5(RECV, RP_PFIRST, BUS_KFA_M, RP_PID, M0 + RS_FA);
    return t.w;
}
void rowsum_pre(const MShape& sh, int log_rows, std::vector<uint32_t>& t) {
    t.assign((size_t)RS_PRE << log_rows, 0u);
    const std::vector<RsRow> rows = rowsum_rows(sh);
    const size_t n = rows.size();
    const int kz[N_TREES] = {K_EL, K_TL, K_PL, K_Q}, kn[N_TREES] = {K_EN, K_TN, K_PN, -1};
    for (int p = 0; p < sh.NP; p++)
        for (size_t i = 0; i < n; i++) {
            uint32_t* r = t.data() + (size_t)RS_PRE * ((size_t)p * n + i);
            const RsRow& w = rows[i];
            r[RP_TAG] = sh.blk_tag(p, w.q, w.tr, w.h, w.b); r[RP_ACT] = 1; r[RP_RIN1] = w.k == 8; r[RP_HALF] = w.k < 8; r[RP_PID] = (uint32_t)p;
            r[RP_NR7] = !w.r7; r[RP_NR3] = !w.r3;
            const uint32_t fl[2] = {RP_F0, RP_F4}, nl[2] = {RP_N0, RP_N4}, kzc[2] = {RP_KZ0, RP_KZ4}, knc[2] = {RP_KN0, RP_KN4};
            for (int s = 0; s < 2; s++) {
                if (w.start[s] < 0) continue;
                const int c = w.start[s];
                r[fl[s]] = 1; r[kzc[s]] = sh.pw_key(p, w.h, sh.seg_e[c][kz[w.tr]]);
                if (kn[w.tr] >= 0) { r[nl[s]] = 1; r[knc[s]] = sh.pw_key(p, w.h, sh.seg_e[c][kn[w.tr]]); }
            }
            const bool first = i == 0 || rows[i - 1].q != w.q || rows[i - 1].h != w.h, last = i + 1 == n || rows[i + 1].q != w.q || rows[i + 1].h != w.h;
            r[RP_GFIRST] = first; r[RP_NG] = !first; r[RP_GLAST] = last; r[RP_KAH] = sh.ah_key(p, w.q, w.h);
            r[RP_PFIRST] = i == 0; r[RP_NFP] = i != 0;
        }
    monty_all(t);
}

// ============================================================================================================ QUERY
constexpr uint32_t Q_PRE = 12, QP_ACT = 0, QP_KEY = 1, QP_KAH = 2, QP_KYH = 3, QP_KZH = 4, QP_TOP = 5, QP_QNS = 6, QP_QN = 7, QP_NQI = 8, QP_LOW = 9, QP_NQ = 10;
constexpr uint32_t QM_IDX = Q_PRE, QM_XQ = Q_PRE + 1, QM_IDX0 = Q_PRE + 2, QM_RO = Q_PRE + 4, QM_AZ = QM_RO + 4, QM_AN = QM_AZ + 4, QM_YZ = QM_AN + 4, QM_YN = QM_YZ + 4, QM_ZETA = QM_YN + 4, QM_ZNX = QM_ZETA + 4,
                   QM_I1 = QM_ZNX + 4, QM_I2 = QM_I1 + 4, QM_P1 = QM_I2 + 4, QM_P2 = QM_P1 + 4, QM_END = QM_P2 + 4;
constexpr uint32_t Q_MAIN = ((QM_END - Q_PRE) + 3u) & ~3u;
std::vector<uint32_t> query_program(const MShape& sh) {
    Cons c;
    const EE x = eb(pscale(pv(QM_XQ), from_monty(MONTY_GEN))), one = eb(pv(QP_ACT));
    c.ext(ALL, esub(emul(esub(x, ev(QM_ZETA)), ev(QM_I1)), one));
    c.ext(ALL, esub(emul(esub(x, ev(QM_ZNX)), ev(QM_I2)), one));
    c.ext(ALL, esub(ev(QM_P1), emul(esub(ev(QM_AZ), ev(QM_YZ)), ev(QM_I1))));
    c.ext(ALL, esub(ev(QM_P2), emul(esub(ev(QM_AN), ev(QM_YN)), ev(QM_I2))));
    c.ext(ALL, esub(ev(QM_RO), eadd(ev(QM_P1), ev(QM_P2))));
    // the rows of a query carry its index: what the fold chain hands a LOWER height is named by the query (two heights of two queries cannot be exchanged)
    c.add(ALL, pmul(pv(QP_TOP), padd(pv(QM_IDX0), pneg(pv(QM_IDX)))));
    c.add(TRANSITION, pmul(pv(QP_NQ, true), padd(pv(QM_IDX0, true), pneg(pv(QM_IDX0)))));
    return c.program(Q_PRE + Q_MAIN, sh.npub_total());
}
std::vector<uint32_t> query_table() {
    Tab t;
    t.add(RECV, QP_TOP, BUS_I, {QP_QNS, QM_IDX});
    t.add(RECV, QP_TOP, BUS_Q, {QP_KEY, QM_IDX, QM_XQ, QM_RO, QM_RO + 1, QM_RO + 2, QM_RO + 3});
    t.add(RECV, QP_LOW, BUS_Q, {QP_KEY, QM_IDX0, QM_IDX, QM_XQ, QM_RO, QM_RO + 1, QM_RO + 2, QM_RO + 3});
    t.add5(RECV, QP_ACT, BUS_AH0, QP_KAH, QM_AZ); t.add5(RECV, QP_ACT, BUS_AH1, QP_KAH, QM_AN);
    t.add5(RECV, QP_ACT, BUS_YH0, QP_KYH, QM_YZ); t.add5(RECV, QP_ACT, BUS_YH1, QP_KYH, QM_YN);
    t.add5(RECV, QP_ACT, BUS_ZH0, QP_KZH, QM_ZETA); t.add5(RECV, QP_ACT, BUS_ZH1, QP_KZH, QM_ZNX);
    t.add(SEND, QP_NQI, BUS_QI, {QP_QN, QM_IDX});
    return t.w;
}
void query_pre(const MShape& sh, int log_rows, std::vector<uint32_t>& t) {
    t.assign((size_t)Q_PRE << log_rows, 0u);
    size_t i = 0;
    for (int p = 0; p < sh.NP; p++)
        for (int q = 0; q < sh.Q; q++)
            for (int h : sh.hs) {
                uint32_t* r = t.data() + (size_t)Q_PRE * i++;
                r[QP_ACT] = 1; r[QP_KEY] = (uint32_t)p * sh.NTREES + (uint32_t)(sh.H - h); r[QP_KAH] = sh.ah_key(p, q, h); r[QP_KYH] = (uint32_t)(p * 32 + h); r[QP_KZH] = (uint32_t)(p * 32 + h);
                r[QP_TOP] = h == sh.H; r[QP_LOW] = h != sh.H; r[QP_NQ] = h != sh.H; r[QP_QNS] = (uint32_t)(p * sh.Q + q); r[QP_QN] = (uint32_t)((p * sh.Q + q) * 32 + h);
                uint32_t nqi = 0;
                for (int tr = 0; tr < N_TREES; tr++) if (sh.has_tree[tr] && sh.tree_hs[tr][0] == h) nqi++;
                r[QP_NQI] = nqi;
            }
    monty_all(t);
}

// ============================================================================================================ FOLD
std::vector<int> inject_layers(const MShape& sh) { std::vector<int> o; for (size_t i = sh.hs.size(); i-- > 0;) if (sh.hs[i] != sh.H) o.push_back(sh.H - sh.hs[i]); std::sort(o.begin(), o.end()); return o; }
std::vector<uint32_t> fold_table(const MShape& sh) {
    using namespace frichip;
    const uint32_t INJ = width_of(sh.R, true, true), INJF = INJ + 4;
    Tab t;
    t.add(SEND, ACTIVE, BUS_E0, {LNX, K2, E0, E0 + 1, E0 + 2, E0 + 3}); t.add(SEND, ACTIVE, BUS_E1, {LNX, K2, E1, E1 + 1, E1 + 2, E1 + 3});
    t.add(SEND, L_REC, BUS_Q, {frichip::PT, IDX, XS, OWN, OWN + 1, OWN + 2, OWN + 3}); t.add(SEND, INJF, BUS_Q, {LNX, INJF + 1, IDX, XS, INJ, INJ + 1, INJ + 2, INJ + 3});
    t.add5(RECV, ACTIVE, BUS_BETA, LNX, BETA);
    t.add5(SEND, L_REC + (uint32_t)sh.R - 1u, BUS_FIN, frichip::PT, FOLD);
    return t.w;
}

// ============================================================================================================ the machine
enum MChipId : int { C_P2R, C_ROWSUM, C_FOLD, C_TS, C_QUERY, C_OPENED, C_SAMPLES, C_SCALARS, C_EVAL, C_LOGUP, N_CHIPS };
struct Machine {
    MShape sh;
    int order[N_CHIPS];
    int32_t log_ns[N_CHIPS]; uint32_t widths[N_CHIPS], pre_widths[N_CHIPS];
    std::vector<uint32_t> prog[N_CHIPS], tab[N_CHIPS];                  // by position
    const uint32_t* progs[N_CHIPS]; size_t prog_words[N_CHIPS]; const uint32_t* tabs[N_CHIPS]; size_t tab_words[N_CHIPS];
    int height[N_CHIPS];                                                // by chip
    uint32_t w_main[N_CHIPS], w_pre[N_CHIPS];                           // by chip
    std::vector<std::vector<uint32_t>> programs_kept, tables_kept;      // the inner machine's own words (the description's pointers need not outlive the call)
};
void heights_of(const MShape& sh, int h[N_CHIPS]) {
    const size_t np = (size_t)sh.NP;
    h[C_P2R] = lg(np * sh.p2_rows); h[C_ROWSUM] = lg(np * rowsum_rows(sh).size()); h[C_FOLD] = lg(np * (size_t)sh.Q * (size_t)sh.R); h[C_TS] = lg(np * (size_t)sh.NTS);
    h[C_QUERY] = lg(np * (size_t)sh.Q * sh.hs.size()); h[C_OPENED] = lg(np * (size_t)(sh.NV / 2)); h[C_SAMPLES] = lg(np * (size_t)sh.NS); h[C_SCALARS] = lg(np * (size_t)sh.C);
    h[C_EVAL] = lg(np * sh.terms.size()); h[C_LOGUP] = lg(np * sh.lrows.size());
}
std::shared_ptr<const Machine> machine_of_impl(const zkhip_machine_desc* d, size_t n_proofs, int* rc) {
    static std::mutex mu;
    static std::map<std::vector<uint64_t>, std::shared_ptr<const Machine>> cache;
    struct Seen { uint64_t hash; size_t n_proofs; uint64_t gen; std::shared_ptr<const Machine> m; };
    static std::vector<Seen> seen;                                          // by the description's CONTENT: a hit skips validation and the digests (25 ms for the join machine's programs)
    auto m = std::make_shared<Machine>();
    // the description's words are copied first: the shape keeps pointers into the copies
    if (!d || d->n_chips < 1 || d->n_chips > MAX_INNER_CHIPS || !d->log_ns || !d->widths || !d->pre_widths || !d->programs || !d->program_words || !d->tables || !d->table_words) {
        *rc = fail(ZKHIP_ERR_INVALID, "machine verifier: null description");
        return nullptr;
    }
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { hsh = (hsh ^ v) * 1099511628211ull; };
    mix((uint64_t)d->n_chips); mix((uint64_t)d->num_queries); mix((uint64_t)d->pow_bits); mix(d->n_public);
    for (int i = 0; i < 8; i++) mix(d->key_root[i]);
    for (int c = 0; c < d->n_chips; c++) {
        if (!d->programs[c] || !d->tables[c] || d->program_words[c] > (1u << 24) || d->table_words[c] > (1u << 16)) { *rc = fail(ZKHIP_ERR_INVALID, "machine verifier: every chip needs a program and a table"); return nullptr; }
        mix((uint64_t)d->log_ns[c]); mix(d->widths[c]); mix(d->pre_widths[c]); mix(d->program_words[c]); mix(d->table_words[c]);
        for (size_t i = 0; i < d->program_words[c]; i++) mix(d->programs[c][i]);
        for (size_t i = 0; i < d->table_words[c]; i++) mix(d->tables[c][i]);
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Seen& sn : seen) {
            if (sn.hash != hsh || sn.n_proofs != n_proofs || sn.gen != g_p2_generation.load() || sn.m->sh.C != d->n_chips) continue;
            bool same = sn.m->sh.Q == d->num_queries && sn.m->sh.PB == d->pow_bits && sn.m->sh.NPUB == (int)d->n_public && std::memcmp(sn.m->sh.key_root, d->key_root, 32) == 0;
            for (int c = 0; same && c < d->n_chips; c++)
                same = sn.m->sh.ln[(size_t)c] == d->log_ns[c] && sn.m->sh.W[(size_t)c] == d->widths[c] && sn.m->sh.Pw[(size_t)c] == d->pre_widths[c] &&
                       sn.m->programs_kept[(size_t)c].size() == d->program_words[c] && sn.m->tables_kept[(size_t)c].size() == d->table_words[c] &&
                       std::memcmp(sn.m->programs_kept[(size_t)c].data(), d->programs[c], d->program_words[c] * 4) == 0 &&
                       std::memcmp(sn.m->tables_kept[(size_t)c].data(), d->tables[c], d->table_words[c] * 4) == 0;
            if (same) { *rc = ZKHIP_OK; return sn.m; }
        }
    }
    zkhip_machine_desc dd = *d;
    std::vector<const uint32_t*> pp((size_t)d->n_chips), tp((size_t)d->n_chips);
    for (int c = 0; c < d->n_chips; c++) {
        if (!d->programs[c] || !d->tables[c] || d->program_words[c] > (1u << 24) || d->table_words[c] > (1u << 16)) { *rc = fail(ZKHIP_ERR_INVALID, "machine verifier: every chip needs a program and a table"); return nullptr; }
        m->programs_kept.emplace_back(d->programs[c], d->programs[c] + d->program_words[c]);
        m->tables_kept.emplace_back(d->tables[c], d->tables[c] + d->table_words[c]);
    }
    for (int c = 0; c < d->n_chips; c++) { pp[(size_t)c] = m->programs_kept[(size_t)c].data(); tp[(size_t)c] = m->tables_kept[(size_t)c].data(); }
    dd.programs = pp.data(); dd.tables = tp.data();
    *rc = make_mshape(&dd, n_proofs, m->sh);
    if (*rc != ZKHIP_OK) return nullptr;
    std::vector<uint64_t> key = m->sh.id;
    key.push_back(g_p2_generation.load());
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) { if (seen.size() >= 8) seen.erase(seen.begin()); seen.push_back(Seen{hsh, n_proofs, g_p2_generation.load(), it->second}); return it->second; }
    }
    const MShape& sh = m->sh;
    int h[N_CHIPS];
    heights_of(sh, h);
    for (int c = 0; c < N_CHIPS; c++) { m->height[c] = h[c]; m->order[c] = c; }
    std::stable_sort(m->order, m->order + N_CHIPS, [&](int a, int b) { return h[a] > h[b]; });
    const std::vector<int> inj = inject_layers(sh);
    const ScCols scc = sc_cols(sh);
    const TsCols tsc = ts_cols(sh);
    const uint32_t w_main[N_CHIPS] = {P2_MAIN, RS_MAIN, frichip::width_of(sh.R, true, true) + 8u, TS_MAIN, Q_MAIN, OS_MAIN, frichip::S_MAIN, rup4(scc.end - scc.pre), EV_MAIN, LG_MAIN};
    const uint32_t w_pre[N_CHIPS] = {P2_PRE, RS_PRE, 0u, tsc.pre, Q_PRE, OS_PRE, frichip::S_PRE, scc.pre, EV_PRE, LG_PRE};
    for (int c = 0; c < N_CHIPS; c++) { m->w_main[c] = w_main[c]; m->w_pre[c] = w_pre[c]; }
    for (int i = 0; i < N_CHIPS; i++) {
        const int c = m->order[i];
        switch (c) {
            case C_P2R: m->prog[i] = p2r_program(sh); m->tab[i] = p2r_table(); break;
            case C_ROWSUM: m->prog[i] = rowsum_program(sh); m->tab[i] = rowsum_table(); break;
            case C_FOLD: m->prog[i] = frichip::build_program(sh.R, true, true, (int)sh.npub_total(), &inj); m->tab[i] = fold_table(sh); break;
            case C_TS: m->prog[i] = ts_program(sh); m->tab[i] = ts_table(sh); break;
            case C_QUERY: m->prog[i] = query_program(sh); m->tab[i] = query_table(); break;
            case C_OPENED: m->prog[i] = opened_program(sh); m->tab[i] = opened_table(); break;
            case C_SAMPLES: m->prog[i] = *frichip::samples_program(sh.R, sh.PB, sh.npub_total()); m->tab[i] = frichip::samples_interactions(); break;
            case C_SCALARS: m->prog[i] = scalars_program(sh); m->tab[i] = scalars_table(sh); break;
            case C_EVAL: m->prog[i] = eval_program(sh); m->tab[i] = eval_table(); break;
            default: m->prog[i] = logup_program(sh); m->tab[i] = logup_table(); break;
        }
        m->log_ns[i] = h[c]; m->widths[i] = w_main[c]; m->pre_widths[i] = w_pre[c];
    }
    for (int i = 0; i < N_CHIPS; i++) { m->progs[i] = m->prog[i].data(); m->prog_words[i] = m->prog[i].size(); m->tabs[i] = m->tab[i].data(); m->tab_words[i] = m->tab[i].size(); }
    std::lock_guard<std::mutex> lk(mu);
    if (cache.size() > 8) cache.clear();
    cache.emplace(key, m);
    if (seen.size() >= 8) seen.erase(seen.begin());
    seen.push_back(Seen{hsh, n_proofs, g_p2_generation.load(), m});
    return m;
}
// (nothing may unwind across the C ABI: the entries below start here)
std::shared_ptr<const Machine> machine_of(const zkhip_machine_desc* d, size_t n_proofs, int* rc) {
    try { return machine_of_impl(d, n_proofs, rc); }
    catch (const std::bad_alloc&) { *rc = fail(ZKHIP_ERR_NOMEM, "machine verifier: out of host memory"); }
    catch (const std::exception& e) { *rc = fail(ZKHIP_ERR_INTERNAL, std::string("machine verifier: ") + e.what()); }
    return nullptr;
}
void samples_pre_all(const MShape& sh, int log_rows, std::vector<uint32_t>& t) {
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
void all_pre(const Machine& m, std::vector<uint32_t> pre[N_CHIPS]) {
    const MShape& sh = m.sh;
    p2r_pre(sh, m.height[C_P2R], pre[C_P2R]); rowsum_pre(sh, m.height[C_ROWSUM], pre[C_ROWSUM]); ts_pre(sh, m.height[C_TS], pre[C_TS]); query_pre(sh, m.height[C_QUERY], pre[C_QUERY]);
    opened_pre(sh, m.height[C_OPENED], pre[C_OPENED]); samples_pre_all(sh, m.height[C_SAMPLES], pre[C_SAMPLES]); scalars_pre(sh, m.height[C_SCALARS], pre[C_SCALARS]);
    eval_pre(sh, m.height[C_EVAL], pre[C_EVAL]); logup_pre(sh, m.height[C_LOGUP], pre[C_LOGUP]);
}

// ============================================================================================================ the witness and the main traces (host)
// Everything below reads the inner proof's words by their position (docs/PROTOCOL.md section 6, version 11) and recomputes what the host
// verifier computes; every check the verifier makes shows up as a mismatch here (ZKHIP_ERR_VERIFY).  The Poseidon2 rows are handed to the
// device as (input state, direction bit, index) per row: the host walks the chains (one scalar permutation per row), the kernel fills the
// 360 columns.
struct Wit {
    const uint32_t* w = nullptr;                    // the proof's words (canonical)
    size_t o_troot, o_proot, o_cum, o_qroot, o_stream, o_lroots, o_final, o_wit, o_queries, per_query;
    Ext gamma, beta, alpha, zeta, fa;
    std::vector<Ext> betas;
    std::vector<uint32_t> samples;                  // [NS][8] canonical words
    std::vector<uint32_t> indices;
    const uint32_t* pubs = nullptr;
    std::vector<Ext> roh;                           // [Q][32]: the reduced opening of every (query, height)
    // per query: where its pieces start
    size_t q_at(int q) const { return o_queries + (size_t)q * per_query; }
};
struct HostTabs {
    ZeroedWords rs, fold, ts, q, op, sm, sc, evl, lgu;
    ZeroedWords p2_in, p2_bit, p2_kp;     // the Poseidon2 rows: input state [16] canonical, direction bit, KP (canonical), per used row (every one written)
    void release_later() {                                   // (a finished call: zeroed again and kept for the next one, off the caller's path -- rec::recycle_later)
        rec::recycle_later({rs.release(), fold.release(), ts.release(), q.release(), op.release(), sm.release(), sc.release(), evl.release(), lgu.release(),
                            p2_in.release(), p2_bit.release(), p2_kp.release()});
    }
};
inline Ext ext_at(const uint32_t* p) { return Ext{{to_monty(p[0]), to_monty(p[1]), to_monty(p[2]), to_monty(p[3])}}; }
inline Ext recombine4(const Ext* four) {            // sum_k X^k four[k]
    Ext acc = ext_zero();
    for (int k = 0; k < 4; k++) { Ext basis = ext_zero(); basis.c[k] = MONTY_R1; acc = ext_add(acc, ext_mul(basis, four[k])); }
    return acc;
}
struct ScVals { Ext zn, invf, invt, self, sell, selt, znx, q0, q1, quo, acc, cum, totin, toto; Ext qz[8]; };

// ---- the transcripts of SEVERAL proofs side by side (round 6): a transcript is one serial sponge chain (NT permutations), so the vector unit is used ACROSS
// the proofs -- sixteen chains per AVX-512 permutation (p2_x16.h), the way the host verifiers walk sixteen queries.  What a chain leaves behind per proof:
struct Transcript { std::vector<Ext> chal, chal2; std::vector<uint32_t> samples; bool done = false; };
struct ProofPos { size_t o_troot, o_proot, o_cum, o_qroot, o_stream, o_lroots, o_final, o_wit, o_queries, per_query, words; };
inline ProofPos proof_pos(const MShape& sh) {
    ProofPos q{};
    size_t pos = (size_t)sh.HL + 2;
    q.o_troot = pos; pos += 8; q.o_proot = pos; pos += 8; q.o_cum = pos; pos += 4 * (size_t)sh.C; q.o_qroot = pos; pos += 8; q.o_stream = pos; pos += 4 * (size_t)sh.NV;
    q.o_lroots = pos; pos += 8 * (size_t)sh.R; q.o_final = pos; pos += 4; q.o_wit = pos; pos += 1; q.o_queries = pos;
    size_t perq = 0;
    for (int t = 0; t < N_TREES; t++) if (sh.has_tree[t]) { for (int c : sh.tree_chips[t]) perq += (*sh.tree_w[t])[(size_t)c]; perq += 8 * (size_t)sh.tree_hs[t][0]; }
    for (int l = 0; l < sh.R; l++) perq += 4 + 8 * (size_t)(sh.H - 1 - l);
    q.per_query = perq; q.words = q.o_queries + (size_t)sh.Q * perq;
    return q;
}
// n <= 16 proofs (words ws[i], public values pubss[i], proof numbers ps[i]); writes every transcript row's input state / bit / KP where fill_proof would
// (stride p2_stride rows per proof) and the challenges into out[i].  Only called when p2x16_available(); every ws[i] has the machine's length.
void walk_transcripts_x16(const MShape& sh, int n, const uint32_t* const* ws, const uint32_t* const* pubss, const int* ps, size_t p2_stride, HostTabs& ht, Transcript* const* out) {
    const ProofPos pp = proof_pos(sh);
    auto src_val = [&](const Src& s, const uint32_t* w, const uint32_t* pubs) -> uint32_t {
        switch (s.kind) {
            case S_CONST: return s.a; case S_TROOT: return w[pp.o_troot + s.a]; case S_PUB: return pubs[s.a] % P; case S_PROOT: return w[pp.o_proot + s.a];
            case S_CUM: return w[pp.o_cum + 4 * s.a + s.b]; case S_QROOT: return w[pp.o_qroot + s.a]; case S_OP: return w[pp.o_stream + s.a];
            case S_LROOT: return w[pp.o_lroots + 8 * s.a + s.b]; case S_FIN: return w[pp.o_final + s.a]; default: return w[pp.o_wit];
        }
    };
    for (int i = 0; i < n; i++) { out[i]->chal.assign((size_t)sh.NT, ext_zero()); out[i]->chal2.assign((size_t)sh.NT, ext_zero()); out[i]->samples.assign(8 * (size_t)sh.NS, 0u); }
    uint32_t st[16][16];
    for (int e = 0; e < 16; e++) for (int j = 0; j < 16; j++) st[e][j] = 0u;
    for (int T = 0; T < sh.NT; T++) {
        if (T <= sh.TP) {
            const std::vector<Src>& pl = sh.plan[(size_t)T];
            for (size_t e = 0; e < pl.size(); e++) for (int i = 0; i < 16; i++) st[e][i] = to_monty(src_val(pl[e], ws[i < n ? i : 0], pubss[i < n ? i : 0]));
        }
        for (int i = 0; i < n; i++) {
            uint32_t* tin = ht.p2_in.data() + 16 * ((size_t)ps[i] * p2_stride + (size_t)T);
            for (int e = 0; e < 16; e++) tin[e] = from_monty(st[e][i]);
            ht.p2_bit.data()[(size_t)ps[i] * p2_stride + (size_t)T] = 0; ht.p2_kp.data()[(size_t)ps[i] * p2_stride + (size_t)T] = 0;
        }
        p2x16_permute(st);
        for (int i = 0; i < n; i++) {
            out[i]->chal[(size_t)T] = Ext{{st[7][i], st[6][i], st[5][i], st[4][i]}}; out[i]->chal2[(size_t)T] = Ext{{st[3][i], st[2][i], st[1][i], st[0][i]}};
            if (T >= sh.TP) for (int j = 0; j < 8; j++) out[i]->samples[8 * (size_t)(T - sh.TP) + (size_t)j] = from_monty(st[7 - j][i]);
        }
    }
    for (int i = 0; i < n; i++) out[i]->done = true;
}

// ---- the per-query part of a proof's tables ON THE DEVICE (round 6).  The plan is a function of the machine's shape alone: the ROWSUM
// rows of one query (where each of a row's eight words sits in the proof relative to the query's first word, which power a segment's sum is weighted with,
// where Horner restarts), the commitments' chains (p2chip.h MrecTreePlan) and the layers at which a height joins the fold chain.  What is left on the
// host per proof is its transcript (a serial sponge chain) and the tables that follow a proof's opened values once (SCALARS, EVAL, LOGUP, OPENED, TS,
// SAMPLES); it hands the device the challenges and sums the per-query rows read (DevVals).
struct WRow { int32_t src[8]; int32_t kz[2], kn[2]; uint32_t flags; };                 // flags: 1 = Horner restarts at word 7 (r7), 2 = at word 3 (r3)
struct WitPlan {
    std::vector<WRow> rows;                 // the rows of ONE query, in table order
    std::vector<uint32_t> g0;               // rows [g0[hi], g0[hi + 1]) belong to height sh.hs[hi]
    std::vector<std::pair<int, uint32_t>> pw_keys;      // (height, exponent) of every power the rows read: DevVals::pw in this order
    std::vector<p2chip::MrecTreePlan> trees;
    std::vector<int32_t> src;               // the commitments' leaf word offsets
    int32_t inj_hi[32];                     // layer -> index in sh.hs of the height that joins there, -1: none
    uint32_t fri_off = 0;                   // a query's first FRI word relative to its first word
    int32_t c0[32];                         // per height index: the first chip of that height (whose zeta g the QUERY row takes)
};
struct DevVals { std::vector<uint32_t> v; };    // [indices Q][betas 4 R][fa 4][zeta 4][per height: znx 4, yz 4, yn 4][pw 4 each], Montgomery except the indices
inline size_t dv_words(const MShape& sh, const WitPlan& pl) { return (size_t)sh.Q + 4 * (size_t)sh.R + 8 + 12 * sh.hs.size() + 4 * pl.pw_keys.size(); }
WitPlan build_wit_plan(const MShape& sh) {
    WitPlan pl;
    const int H = sh.H, R = sh.R;
    // where a query's pieces sit relative to its first word (fill_proof's QPos with q_at = 0)
    size_t row_off[N_TREES][MAX_INNER_CHIPS] = {}, path_off[N_TREES] = {}, at = 0;
    for (int t = 0; t < N_TREES; t++) {
        if (!sh.has_tree[t]) continue;
        for (int c : sh.tree_chips[t]) { row_off[t][c] = at; at += (*sh.tree_w[t])[(size_t)c]; }
        path_off[t] = at; at += 8 * (size_t)sh.tree_hs[t][0];
    }
    pl.fri_off = (uint32_t)at;
    auto leaf_off = [&](int t, int h, uint32_t i) -> int32_t {
        for (const LeafSeg& sg : sh.leaf[t][h].segs) if (i >= sg.at && i < sg.at + sg.width) return (int32_t)(row_off[t][sg.chip] + (i - sg.at));
        return -1;
    };
    auto pw_index = [&](int h, uint32_t e) -> int32_t {
        for (size_t i = 0; i < pl.pw_keys.size(); i++) if (pl.pw_keys[i].first == h && pl.pw_keys[i].second == e) return (int32_t)i;
        pl.pw_keys.emplace_back(h, e);
        return (int32_t)pl.pw_keys.size() - 1;
    };
    const int kz[N_TREES] = {K_EL, K_TL, K_PL, K_Q}, kn[N_TREES] = {K_EN, K_TN, K_PN, -1};
    const std::vector<RsRow> all = rowsum_rows(sh);
    const size_t nq = all.size() / (size_t)sh.Q;
    pl.g0.assign(sh.hs.size() + 1, 0u);
    for (size_t i = 0; i < nq; i++) {                       // query 0's rows: every query has the same
        const RsRow& rr = all[i];
        WRow wr{};
        for (uint32_t j = 0; j < 8; j++) wr.src[j] = j < rr.k ? leaf_off(rr.tr, rr.h, 8 * rr.b + j) : -1;
        for (int sidx = 0; sidx < 2; sidx++) {
            wr.kz[sidx] = wr.kn[sidx] = -1;
            if (rr.start[sidx] < 0) continue;
            wr.kz[sidx] = pw_index(rr.h, sh.seg_e[rr.start[sidx]][kz[rr.tr]]);
            if (kn[rr.tr] >= 0) wr.kn[sidx] = pw_index(rr.h, sh.seg_e[rr.start[sidx]][kn[rr.tr]]);
        }
        wr.flags = (rr.r7 ? 1u : 0u) | (rr.r3 ? 2u : 0u);
        pl.rows.push_back(wr);
        size_t hi = 0;
        while (sh.hs[hi] != rr.h) hi++;
        pl.g0[hi + 1] = (uint32_t)(i + 1);
    }
    for (size_t hi = 1; hi <= sh.hs.size(); hi++) if (pl.g0[hi] < pl.g0[hi - 1]) pl.g0[hi] = pl.g0[hi - 1];     // (a height without rows: an empty range)
    for (size_t hi = 0; hi < sh.hs.size(); hi++) { int c = 0; while (sh.lh[c] != sh.hs[hi]) c++; pl.c0[hi] = c; }
    for (int l = 0; l < 32; l++) pl.inj_hi[l] = -1;
    for (int l = 1; l < R; l++) for (size_t hi = 0; hi < sh.hs.size(); hi++) if (sh.hs[hi] == H - l && sh.hs[hi] != H) pl.inj_hi[l] = (int32_t)hi;
    for (int tr = 0; tr < N_TREES; tr++) {
        if (!sh.has_tree[tr]) continue;
        const std::vector<int>& hs = sh.tree_hs[tr];
        const std::vector<int> so = sponge_order(sh, tr);
        p2chip::MrecTreePlan tp{};
        tp.row0 = (uint32_t)sh.p2_tree0[tr]; tp.rows_per_query = (uint32_t)sh.tree_rows[tr];
        tp.n_sponges = (uint32_t)so.size(); tp.depth = (uint32_t)hs[0]; tp.shift = (uint32_t)(H - hs[0]); tp.path_off = (uint32_t)path_off[tr];
        tp.root_off = -1;
        for (size_t sidx = 0; sidx < so.size(); sidx++) {
            const uint32_t words = sh.leaf[tr][so[sidx]].words;
            tp.sp_words[sidx] = words; tp.sp_src[sidx] = (uint32_t)pl.src.size();
            for (uint32_t i = 0; i < ((words + 7) / 8) * 8; i++) pl.src.push_back(i < words ? leaf_off(tr, so[sidx], i) : -1);
        }
        for (int lvl = 0; lvl < 32; lvl++) tp.inj[lvl] = -1;
        for (int lvl = 0; lvl < hs[0]; lvl++) {
            const int h = hs[0] - lvl - 1;
            if (h != hs[0] && sh.has_h(tr, h)) for (size_t sidx = 0; sidx < so.size(); sidx++) if (so[sidx] == h) tp.inj[lvl] = (int32_t)sidx;
        }
        pl.trees.push_back(tp);
    }
    return pl;
}

int fill_proof(const Machine& m, int p, const uint8_t* inner, size_t inner_len, const uint32_t* pubs, HostTabs& ht, const WitPlan* dplan = nullptr, DevVals* dvals = nullptr,
               const Transcript* pre = nullptr) {
    const MShape& sh = m.sh;
    const bool on_device = dplan != nullptr;               // the per-query tables (ROWSUM, QUERY, FOLD, the queries' Poseidon2 rows) are the device's: see top_finish
    const int C = sh.C, Q = sh.Q, R = sh.R, H = sh.H;
    auto bad = [&](const char* what) { return fail(ZKHIP_ERR_VERIFY, std::string("prove_machine_verifier: proof ") + std::to_string(p) + " rejected: " + what); };
    Wit wt;
    wt.w = (const uint32_t*)inner; wt.pubs = pubs;
    const uint32_t* w = wt.w;
#ifdef ZKHIP_AB_HOOKS
    static const bool sec_timing = getenv("ZKHIP_REC_SECTIONS") != nullptr;
    auto sec_t = std::chrono::steady_clock::now();
    auto sec = [&](const char* what) {
        if (!sec_timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "      [fill_proof %d] %-28s %8.1f us\n", p, what, std::chrono::duration<double, std::micro>(now - sec_t).count());
        sec_t = now;
    };
#else
    auto sec = [](const char*) {};
#endif
    // ---- positions
    size_t pos = (size_t)sh.HL + 2;
    wt.o_troot = pos; pos += 8; wt.o_proot = pos; pos += 8; wt.o_cum = pos; pos += 4 * (size_t)C; wt.o_qroot = pos; pos += 8; wt.o_stream = pos; pos += 4 * (size_t)sh.NV;
    wt.o_lroots = pos; pos += 8 * (size_t)R; wt.o_final = pos; pos += 4; wt.o_wit = pos; pos += 1; wt.o_queries = pos;
    size_t perq = 0;
    for (int t = 0; t < N_TREES; t++) if (sh.has_tree[t]) { for (int c : sh.tree_chips[t]) perq += (*sh.tree_w[t])[(size_t)c]; perq += 8 * (size_t)sh.tree_hs[t][0]; }
    for (int l = 0; l < R; l++) perq += 4 + 8 * (size_t)(H - 1 - l);
    wt.per_query = perq;
    if (inner_len != (wt.o_queries + (size_t)Q * perq) * 4) return bad("its length is not that of a proof of this machine");
    if (w[0] != 0x41544B5Au || w[7] != 16u) return bad("not a proof");                 // ("ZKTA": proof_common.h PROOF_MAGIC)
    for (int i = 0; i < sh.HL; i++) if (w[(i < 6 ? 1 : 2) + i] != sh.head[(size_t)i]) return bad("another machine's header");
    // (what follows is a complete verifier of the proof -- every check of zkhip_verify_machine_keyed has its mismatch below --, and a proof that
    // passed here wrongly would still fail the outer machine's constraints: no second pass through the host verifier)
    for (size_t i = (size_t)sh.HL + 2; i < inner_len / 4; i++) if (w[i] >= P) return bad("a non-canonical word");
    for (int i = 0; i < sh.NPUB; i++) if (pubs[i] >= P) return bad("a non-canonical public value");
    // ---- the transcript: every sponge row's input state, the challenges
    const size_t p2_stride = on_device ? (size_t)sh.NT : sh.p2_rows;      // (device mode: only the transcript's rows are walked here)
    uint32_t* tin = ht.p2_in.data() + 16 * (size_t)p * p2_stride;
    uint32_t* tbit = ht.p2_bit.data() + (size_t)p * p2_stride;
    uint32_t* tkp = ht.p2_kp.data() + (size_t)p * p2_stride;
    auto src_val = [&](const Src& s) -> uint32_t {
        switch (s.kind) {
            case S_CONST: return s.a; case S_TROOT: return w[wt.o_troot + s.a]; case S_PUB: return pubs[s.a] % P; case S_PROOT: return w[wt.o_proot + s.a];
            case S_CUM: return w[wt.o_cum + 4 * s.a + s.b]; case S_QROOT: return w[wt.o_qroot + s.a]; case S_OP: return w[wt.o_stream + s.a];
            case S_LROOT: return w[wt.o_lroots + 8 * s.a + s.b]; case S_FIN: return w[wt.o_final + s.a]; default: return w[wt.o_wit];
        }
    };
    std::vector<Ext> chal((size_t)sh.NT), chal2((size_t)sh.NT);
    wt.samples.assign(8 * (size_t)sh.NS, 0u);
    if (pre && pre->done) { chal = pre->chal; chal2 = pre->chal2; wt.samples = pre->samples; }       // (walked beside other proofs' transcripts: walk_transcripts_x16)
    else {
        uint32_t st[16];
        for (int j = 0; j < 16; j++) st[j] = 0u;
        for (int T = 0; T < sh.NT; T++) {
            if (T <= sh.TP) { const std::vector<Src>& pl = sh.plan[(size_t)T]; for (size_t j = 0; j < pl.size(); j++) st[j] = to_monty(src_val(pl[j])); }
            for (int j = 0; j < 16; j++) tin[16 * (size_t)T + (size_t)j] = from_monty(st[j]);
            tbit[T] = 0; tkp[T] = 0;
            p2_permute(st);
            chal[(size_t)T] = Ext{{st[7], st[6], st[5], st[4]}}; chal2[(size_t)T] = Ext{{st[3], st[2], st[1], st[0]}};
            if (T >= sh.TP) for (int j = 0; j < 8; j++) wt.samples[8 * (size_t)(T - sh.TP) + (size_t)j] = from_monty(st[7 - j]);
        }
    }
    sec("checks + transcript");
    wt.gamma = chal[(size_t)sh.TG]; wt.beta = chal2[(size_t)sh.TG]; wt.alpha = chal[(size_t)sh.TA]; wt.zeta = chal[(size_t)sh.TQ]; wt.fa = chal[(size_t)sh.TF];
    for (int l = 0; l < R; l++) wt.betas.push_back(chal[(size_t)(sh.TL0 + l)]);
    // proof of work and the query indices (SAMPLES main)
    {
        std::vector<uint32_t> one, drawn;
        frichip::samples_main(R, (size_t)Q, lg((size_t)sh.NS), wt.samples.data(), one, drawn);
        if (sh.PB && (wt.samples[0] & ((1u << sh.PB) - 1u)) != 0) return bad("proof of work");
        std::memcpy(ht.sm.data() + (size_t)frichip::S_MAIN * (size_t)p * (size_t)sh.NS, one.data(), (size_t)frichip::S_MAIN * (size_t)sh.NS * 4);
        wt.indices = drawn;
        if ((int)wt.indices.size() != Q) return fail(ZKHIP_ERR_INTERNAL, "prove_machine_verifier: query indices");
    }
    sec("samples");
    auto stream_ext = [&](uint32_t posv) { return ext_at(w + wt.o_stream + 4 * (size_t)posv); };
    // ---- SCALARS values per chip
    std::vector<ScVals> sc((size_t)C);
    std::vector<Ext> zp{wt.zeta};
    for (int i = 0; i < R; i++) zp.push_back(ext_mul(zp.back(), zp.back()));
    {
        const ScCols cc = sc_cols(sh);
        Ext tot = ext_zero();
        for (int ch = 0; ch < C; ch++) {
            ScVals& v = sc[(size_t)ch];
            const int ln = sh.ln[ch];
            const uint32_t wN = two_adic_generator(ln), winv = finv(wN);
            v.zn = zp[(size_t)ln];
            const Ext zh = ext_sub_base(v.zn, MONTY_R1);
            v.invf = ext_inv(ext_sub_base(wt.zeta, MONTY_R1)); v.invt = ext_inv(ext_sub_base(wt.zeta, winv));
            v.self = ext_mul(zh, v.invf); v.sell = ext_mul(zh, v.invt); v.selt = ext_sub_base(wt.zeta, winv);
            v.znx = ext_mul_base(wt.zeta, wN);
            for (uint32_t k = 0; k < 8; k++) v.qz[k] = stream_ext(sh.seg_pos[ch][K_Q] + k);
            v.q0 = recombine4(v.qz); v.q1 = recombine4(v.qz + 4);
            uint32_t zw[4];
            chunk_weights(ln, zw);
            const Ext zps0 = ext_add_base(ext_mul_base(v.zn, to_monty(zw[0])), to_monty(zw[1])), zps1 = ext_add_base(ext_mul_base(v.zn, to_monty(zw[2])), to_monty(zw[3]));
            v.quo = ext_add(ext_mul(zps0, v.q0), ext_mul(zps1, v.q1));
            v.acc = ext_mul(v.quo, zh);
            v.cum = ext_at(w + wt.o_cum + 4 * (size_t)ch); v.totin = tot;
            tot = ext_add(tot, v.cum);
            v.toto = tot;
            uint32_t* r = ht.sc.data() + (size_t)m.w_main[C_SCALARS] * ((size_t)p * (size_t)C + (size_t)ch);
            auto put = [&](uint32_t col, const Ext& e) { put_ext(r, col - cc.pre, e); };
            put(cc.ALPHA, wt.alpha); put(cc.ZETA, wt.zeta); put(cc.FA, wt.fa); put(cc.GAMMA, wt.gamma); put(cc.BETA, wt.beta);
            for (int i = 0; i < R; i++) put(cc.ZP + 4 * (uint32_t)i, zp[(size_t)i + 1]);
            put(cc.ZN, v.zn); put(cc.INVF, v.invf); put(cc.INVT, v.invt); put(cc.SELF, v.self); put(cc.SELL, v.sell); put(cc.SELT, v.selt); put(cc.ZNX, v.znx);
            for (uint32_t k = 0; k < 8; k++) put(cc.QZ + 4 * k, v.qz[k]);
            put(cc.Q0, v.q0); put(cc.Q1, v.q1); put(cc.QUO, v.quo); put(cc.ACC, v.acc); put(cc.CUM, v.cum); put(cc.TOTIN, v.totin); put(cc.TOTO, v.toto);
            if (ch == 0) for (uint32_t j = 0; j < 8; j++) r[cc.KR - cc.pre + j] = to_monty(sh.key_root[j]);
        }
        if (!ext_eq(tot, ext_zero())) return bad("the lookups do not balance");
    }
    auto value_of = [&](uint32_t key) -> Ext {
        if (key == 0) return ext_one();
        if (key <= sh.NV) return stream_ext(key - 1);
        if (key <= sh.NV + (uint32_t)sh.NPUB) return ext_from_base(to_monty(pubs[key - 1 - sh.NV] % P));
        const uint32_t c = (key - 1 - sh.NV - (uint32_t)sh.NPUB) / 3, which = (key - 1 - sh.NV - (uint32_t)sh.NPUB) % 3;
        return which == 0 ? sc[c].self : (which == 1 ? sc[c].sell : sc[c].selt);
    };
    sec("scalars");
    // ---- EVAL
    std::vector<Ext> acc_eval((size_t)C);
    {
        const size_t nt = sh.terms.size();
        Ext run = ext_zero();
        for (size_t i = 0; i < nt; i++) {
            const ETerm& e = sh.terms[i];
            uint32_t* r = ht.evl.data() + (size_t)EV_MAIN * ((size_t)p * nt + i);
            if (i == sh.term0[e.chip]) run = ext_zero();
            const Ext f0 = value_of(e.key[0]), f1 = value_of(e.key[1]), f2 = value_of(e.key[2]);
            const Ext mm = ext_mul(f0, f1), tv = ext_mul_base(ext_mul(mm, f2), to_monty(e.coeff));
            put_ext(r, EV_F0, f0); put_ext(r, EV_F0 + 4, f1); put_ext(r, EV_F0 + 8, f2); put_ext(r, EV_M, mm); put_ext(r, EV_TV, tv); put_ext(r, EV_ACCIN, run);
            run = ext_add(e.first ? ext_mul(run, wt.alpha) : run, tv);
            put_ext(r, EV_ACCO, run); put_ext(r, EV_ALPHA, wt.alpha);
            acc_eval[(size_t)e.chip] = run;
        }
    }
    sec("eval");
    // ---- LOGUP
    {
        const LgCols lc = lg_cols();
        const size_t n = sh.lrows.size();
        Ext bp[8];
        bp[0] = wt.beta;
        for (int k = 1; k < 8; k++) bp[k] = ext_mul(bp[k - 1], wt.beta);
        Ext acc = ext_zero(), suml = ext_zero(), sumn = ext_zero();
        for (size_t i = 0; i < n; i++) {
            const LRow& row = sh.lrows[i];
            const int c = row.chip;
            uint32_t* r = ht.lgu.data() + (size_t)LG_MAIN * ((size_t)p * n + i);
            auto put = [&](uint32_t col, const Ext& e, uint32_t k = 0) { put_ext(r, col - LG_PRE + 4 * k, e); };
            if (i == sh.lrow0[(size_t)c]) { acc = acc_eval[(size_t)c]; suml = ext_zero(); sumn = ext_zero(); }
            put(lc.ALPHA, wt.alpha); put(lc.GAMMA, wt.gamma);
            for (uint32_t k = 0; k < 8; k++) put(lc.BP, bp[k], k);
            Ext ph[4], pn[4];
            for (uint32_t k = 0; k < 4; k++) { ph[k] = value_of(row.phi[k]); pn[k] = value_of(row.phin[k]); put(lc.PH, ph[k], k); put(lc.PN, pn[k], k); }
            const Ext phi = recombine4(ph), phin = recombine4(pn);
            put(lc.PHI, phi); put(lc.PHIN, phin); put(lc.ACCIN, acc); put(lc.SUML, suml); put(lc.SUMN, sumn);
            Ext va[8], vb[8], ma = ext_zero(), mb = ext_zero();
            for (int k = 0; k < 8; k++) va[k] = vb[k] = ext_zero();
            uint32_t na = 0, nb = 0, busa = 0, busb = 0, sa = 0, sb = 0, isp = 0, hasb = 0, nob = 0;
            if (row.bnd) { for (uint32_t k = 0; k < 3; k++) va[k] = value_of(row.sels[k]); na = 3; }
            else {
                isp = 1;
                for (size_t k = 0; k < row.a.vkeys.size(); k++) va[k] = value_of(row.a.vkeys[k]);
                na = (uint32_t)row.a.vkeys.size(); ma = value_of(row.a.mkey); busa = row.a.bus; sa = row.a.sign;
                if (row.b.present) {
                    for (size_t k = 0; k < row.b.vkeys.size(); k++) vb[k] = value_of(row.b.vkeys[k]);
                    nb = (uint32_t)row.b.vkeys.size(); mb = value_of(row.b.mkey); busb = row.b.bus; sb = row.b.sign; hasb = 1;
                } else nob = 1;
            }
            for (uint32_t k = 0; k < 8; k++) { put(lc.VA, va[k], k); put(lc.VB, vb[k], k); }
            put(lc.MA, ma); put(lc.MB, mb);
            Ext da = ext_add_base(isp ? wt.gamma : ext_zero(), to_monty(busa)), db = ext_add_base(hasb ? wt.gamma : ext_zero(), to_monty((busb + nob) % P));
            for (uint32_t k = 0; k < na; k++) da = ext_add(da, ext_mul(bp[k], va[k]));
            for (uint32_t k = 0; k < nb; k++) db = ext_add(db, ext_mul(bp[k], vb[k]));
            const Ext sma = ext_mul_base(ma, to_monty(sa)), smb = ext_mul_base(mb, to_monty(sb));
            const Ext cst = ext_sub(ext_mul(ext_mul(phi, da), db), ext_add(ext_mul(sma, db), ext_mul(smb, da)));
            const Ext u1 = ext_add(ext_mul(acc, wt.alpha), ext_mul(va[0], ext_sub(phi, suml)));
            const Ext u2 = ext_add(ext_mul(u1, wt.alpha), ext_mul(va[2], ext_sub(ext_sub(phin, phi), sumn)));
            const Ext cum = row.bnd ? ext_at(w + wt.o_cum + 4 * (size_t)c) : ext_zero();
            const Ext acco = row.bnd ? ext_add(ext_mul(u2, wt.alpha), ext_mul(va[1], ext_sub(phi, cum))) : ext_add(ext_mul(acc, wt.alpha), cst);
            put(lc.DA, da); put(lc.DB, db); put(lc.CST, cst); put(lc.U1, u1); put(lc.U2, u2); put(lc.CUM, cum); put(lc.ACCO, acco);
            acc = acco; suml = ext_add(suml, phi); sumn = ext_add(sumn, phin);
            if (i == sh.lrow1[(size_t)c] && !ext_eq(acco, sc[(size_t)c].acc)) return bad("a chip's constraints do not match its quotient at zeta");
        }
    }
    sec("logup");
    // ---- OPENED (the stream): the sums per height, the powers the row sums are weighted with
    std::map<std::pair<int, uint32_t>, Ext> pwh;
    Ext yz_h[32], yn_h[32];
    {
        const std::vector<StreamRow> rows = stream_rows(sh);
        const size_t n = rows.size();
        const Ext fa2 = ext_mul(wt.fa, wt.fa);
        Ext pw = ext_one(), yz = ext_zero(), yn = ext_zero();
        for (size_t i = 0; i < n; i++) {
            const StreamRow& s = rows[i];
            uint32_t* r = ht.op.data() + (size_t)OS_MAIN * ((size_t)p * n + i);
            if (s.first) { pw = ext_one(); yz = ext_zero(); yn = ext_zero(); }
            const uint32_t* words = w + wt.o_stream + 8 * i;
            for (int j = 0; j < 8; j++) r[OS_W + j] = to_monty(words[j]);
            const Ext v0 = ext_at(words), v1 = ext_at(words + 4), mm = ext_add(v0, ext_mul(wt.fa, v1));
            put_ext(r, OS_FA, wt.fa); put_ext(r, OS_FA2, fa2); put_ext(r, OS_PW, pw); put_ext(r, OS_M, mm); put_ext(r, OS_YZIN, yz); put_ext(r, OS_YNIN, yn);
            const Ext add = ext_mul(pw, mm);
            if (is_next_kind(s.kind)) yn = ext_add(yn, add); else yz = ext_add(yz, add);
            put_ext(r, OS_YZO, yz); put_ext(r, OS_YNO, yn);
            pwh[{s.h, s.e}] = pw;
            pw = ext_mul(pw, fa2);
            put_ext(r, OS_PWN, pw);
            if (s.last) { yz_h[s.h] = yz; yn_h[s.h] = yn; }
        }
    }
    sec("opened");
    if (on_device) {
        // ---- what the device's per-query kernels read of this proof
        std::vector<uint32_t>& v = dvals->v;
        v.assign(dv_words(sh, *dplan), 0u);
        size_t at = 0;
        for (int q = 0; q < Q; q++) v[at++] = wt.indices[(size_t)q];
        auto pute = [&](const Ext& e) { for (int j = 0; j < 4; j++) v[at++] = e.c[j]; };
        for (int l = 0; l < R; l++) pute(wt.betas[(size_t)l]);
        pute(wt.fa); pute(wt.zeta);
        for (size_t hi = 0; hi < sh.hs.size(); hi++) { pute(sc[(size_t)dplan->c0[hi]].znx); pute(yz_h[sh.hs[hi]]); pute(yn_h[sh.hs[hi]]); }
        for (const auto& k : dplan->pw_keys) pute(pwh[{k.first, k.second}]);
    }
    if (!on_device) {
    // ---- per query: where its rows and paths are
    struct QPos { size_t row[N_TREES][MAX_INNER_CHIPS]; size_t path[N_TREES]; size_t fri; };
    std::vector<QPos> qp((size_t)Q);
    for (int q = 0; q < Q; q++) {
        size_t at = wt.q_at(q);
        for (int t = 0; t < N_TREES; t++) {
            if (!sh.has_tree[t]) continue;
            for (int c : sh.tree_chips[t]) { qp[(size_t)q].row[t][c] = at; at += (*sh.tree_w[t])[(size_t)c]; }
            qp[(size_t)q].path[t] = at; at += 8 * (size_t)sh.tree_hs[t][0];
        }
        qp[(size_t)q].fri = at;
    }
    auto leaf_word = [&](int q, int t, int h, uint32_t i) -> uint32_t {        // word i of the concatenated rows of height h in tree t
        for (const LeafSeg& s : sh.leaf[t][h].segs) if (i >= s.at && i < s.at + s.width) return w[qp[(size_t)q].row[t][s.chip] + (i - s.at)];
        return 0u;
    };
    // ---- ROWSUM
    wt.roh.assign((size_t)Q * 32, ext_zero());
    std::vector<Ext> az_qh((size_t)Q * 32, ext_zero()), an_qh((size_t)Q * 32, ext_zero());
    {
        const std::vector<RsRow> rows = rowsum_rows(sh);
        const size_t n = rows.size();
        const int kz[N_TREES] = {K_EL, K_TL, K_PL, K_Q}, kn[N_TREES] = {K_EN, K_TN, K_PN, -1};
        Ext acc = ext_zero(), az = ext_zero(), an = ext_zero();
        for (size_t i = 0; i < n; i++) {
            const RsRow& rr = rows[i];
            uint32_t* r = ht.rs.data() + (size_t)RS_MAIN * ((size_t)p * n + i);
            if (i == 0 || rows[i - 1].q != rr.q || rows[i - 1].h != rr.h) { az = ext_zero(); an = ext_zero(); }
            uint32_t vals[8];
            for (uint32_t j = 0; j < 8; j++) vals[j] = j < rr.k ? leaf_word(rr.q, rr.tr, rr.h, 8 * rr.b + j) : 0u;
            for (int j = 0; j < 8; j++) r[RS_V + j] = to_monty(vals[j]);
            put_ext(r, RS_FA, wt.fa); put_ext(r, RS_ACCIN, acc); put_ext(r, RS_AZIN, az); put_ext(r, RS_ANIN, an);
            Ext steps[8], prev = acc;
            for (int s = 7; s >= 0; s--) {
                Ext carried = ((s == 7 && rr.r7) || (s == 3 && rr.r3)) ? ext_zero() : ext_mul(prev, wt.fa);
                prev = ext_add_base(carried, to_monty(vals[s]));
                steps[s] = prev;
                put_ext(r, RS_T + 4 * (uint32_t)s, prev);
            }
            acc = steps[0];
            const uint32_t kzc[2] = {RS_KZ0, RS_KZ4}, knc[2] = {RS_KN0, RS_KN4};
            for (int s = 0; s < 2; s++) {
                if (rr.start[s] < 0) continue;
                const int c = rr.start[s];
                const Ext kzv = pwh[{rr.h, sh.seg_e[c][kz[rr.tr]]}];
                put_ext(r, kzc[s], kzv);
                az = ext_add(az, ext_mul(kzv, steps[4 * s]));
                if (kn[rr.tr] >= 0) { const Ext knv = pwh[{rr.h, sh.seg_e[c][kn[rr.tr]]}]; put_ext(r, knc[s], knv); an = ext_add(an, ext_mul(knv, steps[4 * s])); }
            }
            put_ext(r, RS_AZO, az); put_ext(r, RS_ANO, an);
            az_qh[(size_t)rr.q * 32 + (size_t)rr.h] = az; an_qh[(size_t)rr.q * 32 + (size_t)rr.h] = an;
        }
    }
    sec("rowsum");
    // ---- QUERY
    {
        size_t i = (size_t)p * (size_t)Q * sh.hs.size();
        for (int q = 0; q < Q; q++)
            for (int h : sh.hs) {
                uint32_t* r = ht.q.data() + (size_t)Q_MAIN * i++;
                const uint32_t ic = wt.indices[(size_t)q] >> (H - h);
                const uint32_t xq = fpow(two_adic_generator(h), reverse_bits(ic, h));
                const Ext x = ext_from_base(fmul(MONTY_GEN, xq));
                int c0 = 0;
                while (sh.lh[c0] != h) c0++;
                const Ext zeta = wt.zeta, znx = sc[(size_t)c0].znx, az = az_qh[(size_t)q * 32 + (size_t)h], an = an_qh[(size_t)q * 32 + (size_t)h], yz = yz_h[h], yn = yn_h[h];
                const Ext i1 = ext_inv(ext_sub(x, zeta)), i2 = ext_inv(ext_sub(x, znx));
                const Ext p1 = ext_mul(ext_sub(az, yz), i1), p2 = ext_mul(ext_sub(an, yn), i2), ro = ext_add(p1, p2);
                wt.roh[(size_t)q * 32 + (size_t)h] = ro;
                r[QM_IDX - Q_PRE] = to_monty(ic); r[QM_XQ - Q_PRE] = xq; r[QM_IDX0 - Q_PRE] = to_monty(wt.indices[(size_t)q]);
                auto put = [&](uint32_t col, const Ext& e) { put_ext(r, col - Q_PRE, e); };
                put(QM_RO, ro); put(QM_AZ, az); put(QM_AN, an); put(QM_YZ, yz); put(QM_YN, yn); put(QM_ZETA, zeta); put(QM_ZNX, znx); put(QM_I1, i1); put(QM_I2, i2); put(QM_P1, p1); put(QM_P2, p2);
            }
    }
    sec("query");
    // ---- FOLD rows (host: the recursion form with what joins on the way down) and the Poseidon2 rows of the FRI layers
    {
        using namespace frichip;
        const uint32_t FW = m.w_main[C_FOLD], INJ = width_of(R, true, true), INJF = INJ + 4;
        // (the queries are independent: a few threads per proof walk them -- one scalar permutation per Poseidon2 row is most of this function's time)
        std::vector<int> qerr((size_t)Q, 0);
        const bool x16 = p2x16_available();                 // sixteen queries walked in lockstep, one per AVX-512 lane (p2_x16.h); else one scalar permutation per row
        std::vector<uint32_t> pair_words(x16 ? 8 * (size_t)Q * (size_t)R : 0), pair_k(x16 ? (size_t)Q * (size_t)R : 0);
        std::vector<size_t> pair_path(x16 ? (size_t)Q * (size_t)R : 0);
        auto fri_q = [&](int q) -> int {
            size_t prow = sh.p2_fri0 + (size_t)q * sh.fri_rows;
            uint32_t idx = wt.indices[(size_t)q];
            Ext own = wt.roh[(size_t)q * 32 + (size_t)H];
            uint32_t tcol[MAX_LAYERS];
            size_t fat = qp[(size_t)q].fri;
            for (int l = 0; l < R; l++) {
                uint32_t* row = ht.fold.data() + (size_t)FW * (((size_t)p * (size_t)Q + (size_t)q) * (size_t)R + (size_t)l);
                row[INJF + 1] = to_monty(wt.indices[(size_t)q]);
                const int hh = H - l;
                if (l > 0 && hh != H) {
                    bool inj = false;
                    for (int x : sh.hs) if (x == hh) inj = true;
                    if (inj) { const Ext v = wt.roh[(size_t)q * 32 + (size_t)hh]; put_ext(row, INJ, v); row[INJF] = MONTY_R1; own = ext_add(own, v); }
                }
                const uint32_t bit = idx & 1u, k = idx >> 1;
                const Ext sib = ext_at(w + fat), beta = wt.betas[(size_t)l];
                const uint32_t* path = w + fat + 4;
                fat += 4 + 8 * (size_t)(H - 1 - l);
                const Ext e0 = bit ? sib : own, e1 = bit ? own : sib;
                const int lhh = H - (l + 1);
                const uint32_t x = fpow(two_adic_generator(lhh + 1), reverse_bits(k, lhh)), xi = finv(x);
                const Ext even = ext_mul_base(ext_add(e0, e1), MONTY_INV2), odd = ext_mul_base(ext_sub(e0, e1), fmul(MONTY_INV2, xi)), fold = ext_add(even, ext_mul(beta, odd));
                put_ext(row, E0, e0); put_ext(row, E1, e1); put_ext(row, BETA, beta); put_ext(row, FOLD, fold);
                row[BIT] = bit ? MONTY_R1 : 0u; row[K] = to_monty(k); row[X] = x; row[XI] = xi; row[S] = fmul(x, x);
                tcol[l] = bit ? two_adic_generator(l + 1) : MONTY_R1;
                row[frichip::T] = tcol[l]; row[ACTIVE] = MONTY_R1; row[LN] = to_monty((uint32_t)l); row[L_REC + (uint32_t)l] = MONTY_R1;
                if (l + 1 < R) { row[G] = MONTY_R1; row[GS] = row[S]; row[GT] = tcol[l]; }
                put_ext(row, OWN, own);
                row[K2] = to_monty(2u * k); row[IDX] = to_monty(2u * k + bit);
                row[XS] = bit ? fsub(0u, x) : x; row[frichip::PT] = to_monty((uint32_t)p * sh.NTREES); row[LNX] = to_monty((uint32_t)p * sh.NTREES + (uint32_t)l);
                if (x16) {                                   // (hashed below, sixteen queries at a time)
                    uint32_t* pw8 = pair_words.data() + 8 * ((size_t)q * (size_t)R + (size_t)l);
                    for (int j = 0; j < 4; j++) { pw8[j] = e0.c[j]; pw8[4 + j] = e1.c[j]; }
                    pair_k[(size_t)q * (size_t)R + (size_t)l] = k; pair_path[(size_t)q * (size_t)R + (size_t)l] = (size_t)(path - w);
                    own = fold; idx = k;
                    continue;
                }
                // the layer's leaf (the pair) and its path: Poseidon2 rows
                uint32_t st[16];
                for (int j = 0; j < 4; j++) { st[j] = e0.c[j]; st[4 + j] = e1.c[j]; }
                for (int j = 8; j < 16; j++) st[j] = 0u;
                for (int j = 0; j < 16; j++) tin[16 * prow + (size_t)j] = from_monty(st[j]);
                tbit[prow] = 0; tkp[prow] = 2u * k; prow++;
                p2_permute(st);
                for (int lvl = 0; lvl < H - 1 - l; lvl++) {
                    const uint32_t b = (k >> lvl) & 1u;
                    uint32_t in[16];
                    for (int j = 0; j < 8; j++) { in[b ? 8 + j : j] = st[j]; in[b ? j : 8 + j] = to_monty(path[8 * (size_t)lvl + (size_t)j]); }
                    for (int j = 0; j < 16; j++) tin[16 * prow + (size_t)j] = from_monty(in[j]);
                    tbit[prow] = b; tkp[prow] = k >> lvl; prow++;
                    std::memcpy(st, in, sizeof st);
                    p2_permute(st);
                }
                for (int j = 0; j < 8; j++) if (from_monty(st[j]) != w[wt.o_lroots + 8 * (size_t)l + (size_t)j]) return 1;
                own = fold; idx = k;
            }
            uint32_t bacc = idx ? two_adic_generator(R + 1) : MONTY_R1;
            for (int l = R - 1; l >= 0; l--) {
                bacc = fmul(bacc, tcol[l]);
                ht.fold.data()[(size_t)FW * (((size_t)p * (size_t)Q + (size_t)q) * (size_t)R + (size_t)l) + frichip::B] = bacc;
            }
            for (int j = 0; j < 4; j++) if (from_monty(own.c[j]) != w[wt.o_final + (size_t)j]) return 2;
            return 0;
        };
        // ---- the four commitments: the shorter heights' sponges, the tallest's, the path with the injections
        auto tree_q = [&](int q) -> int {
        for (int tr = 0; tr < N_TREES; tr++) {
            if (!sh.has_tree[tr]) continue;
            const std::vector<int>& hs = sh.tree_hs[tr];
            const std::vector<int> so = sponge_order(sh, tr);
            const uint32_t* root = tr == T_E ? sh.key_root : w + (tr == T_T ? wt.o_troot : tr == T_P ? wt.o_proot : wt.o_qroot);
            size_t prow = sh.p2_tree0[tr] + (size_t)q * sh.tree_rows[tr];
            {
                const uint32_t index = wt.indices[(size_t)q] >> (H - hs[0]);
                uint32_t dg[32][8], st[16];
                for (int h : so) {
                    const uint32_t words = sh.leaf[tr][h].words, nb = (words + 7) / 8;
                    for (int j = 0; j < 16; j++) st[j] = 0u;
                    for (uint32_t b = 0; b < nb; b++) {
                        const uint32_t k = words - 8 * b < 8 ? words - 8 * b : 8u;
                        for (uint32_t j = 0; j < k; j++) st[j] = to_monty(leaf_word(q, tr, h, 8 * b + j));
                        for (int j = 0; j < 16; j++) tin[16 * prow + (size_t)j] = from_monty(st[j]);
                        tbit[prow] = 0; tkp[prow] = (b == nb - 1 && h == hs[0]) ? 2u * index : 0u; prow++;
                        p2_permute(st);
                    }
                    for (int j = 0; j < 8; j++) dg[h][j] = st[j];
                }
                uint32_t cur[8];
                for (int j = 0; j < 8; j++) cur[j] = dg[hs[0]][j];
                const uint32_t* path = w + qp[(size_t)q].path[tr];
                for (int lvl = 0; lvl < hs[0]; lvl++) {
                    const uint32_t b = (index >> lvl) & 1u;
                    uint32_t in[16];
                    for (int j = 0; j < 8; j++) { in[b ? 8 + j : j] = cur[j]; in[b ? j : 8 + j] = to_monty(path[8 * (size_t)lvl + (size_t)j]); }
                    for (int j = 0; j < 16; j++) tin[16 * prow + (size_t)j] = from_monty(in[j]);
                    tbit[prow] = b; tkp[prow] = index >> lvl; prow++;
                    p2_permute(in);
                    for (int j = 0; j < 8; j++) cur[j] = in[j];
                    const int h = hs[0] - lvl - 1;
                    if (h != hs[0] && sh.has_h(tr, h)) {
                        for (int j = 0; j < 8; j++) { in[j] = cur[j]; in[8 + j] = dg[h][j]; }
                        for (int j = 0; j < 16; j++) tin[16 * prow + (size_t)j] = from_monty(in[j]);
                        tbit[prow] = 0; tkp[prow] = index >> (lvl + 1); prow++;
                        p2_permute(in);
                        for (int j = 0; j < 8; j++) cur[j] = in[j];
                    }
                }
                for (int j = 0; j < 8; j++) if (from_monty(cur[j]) != root[j]) return 3;
                if (prow != sh.p2_tree0[tr] + (size_t)(q + 1) * sh.tree_rows[tr]) return 4;
            }
        }
        return 0;
        };
        // the same rows, sixteen queries in lockstep (the queries of a proof have one structure): st[element][lane]
        auto walk16 = [&](int q0, int cnt) -> int {
            uint32_t st[16][16], cur[8][16];
            uint32_t bits[16], kps[16];
            auto emit = [&](size_t base0, size_t stride, size_t r) {
                for (int j = 0; j < cnt; j++) {
                    const size_t row = base0 + (size_t)(q0 + j) * stride + r;
                    for (int e = 0; e < 16; e++) tin[16 * row + (size_t)e] = from_monty(st[e][j]);
                    tbit[row] = bits[j]; tkp[row] = kps[j];
                }
                p2x16_permute(st);
            };
            for (int l = 0; l < R; l++) {
                size_t r = (size_t)l + (size_t)l * (size_t)(2 * (H - 1) - (l - 1)) / 2;          // rows of the layers before: l leaves + sum_{i < l} (H - 1 - i) path rows
                for (int j = 0; j < 16; j++) {
                    const size_t at = ((size_t)(q0 + (j < cnt ? j : 0)) * (size_t)R + (size_t)l);
                    for (int e = 0; e < 8; e++) st[e][j] = pair_words[8 * at + (size_t)e];
                    for (int e = 8; e < 16; e++) st[e][j] = 0u;
                    bits[j] = 0; kps[j] = 2u * pair_k[at];
                }
                emit(sh.p2_fri0, sh.fri_rows, r++);
                for (int lvl = 0; lvl < H - 1 - l; lvl++) {
                    for (int e = 0; e < 8; e++) for (int j = 0; j < 16; j++) cur[e][j] = st[e][j];
                    for (int j = 0; j < 16; j++) {
                        const size_t at = ((size_t)(q0 + (j < cnt ? j : 0)) * (size_t)R + (size_t)l);
                        const uint32_t k = pair_k[at], b = (k >> lvl) & 1u;
                        const uint32_t* path = w + pair_path[at];
                        for (int e = 0; e < 8; e++) { const uint32_t sv = to_monty(path[8 * (size_t)lvl + (size_t)e]); st[b ? 8 + e : e][j] = cur[e][j]; st[b ? e : 8 + e][j] = sv; }
                        bits[j] = b; kps[j] = k >> lvl;
                    }
                    emit(sh.p2_fri0, sh.fri_rows, r++);
                }
                for (int j = 0; j < cnt; j++) for (int e = 0; e < 8; e++) if (from_monty(st[e][j]) != w[wt.o_lroots + 8 * (size_t)l + (size_t)e]) return 1;
            }
            for (int tr = 0; tr < N_TREES; tr++) {
                if (!sh.has_tree[tr]) continue;
                const std::vector<int>& hs = sh.tree_hs[tr];
                const std::vector<int> so = sponge_order(sh, tr);
                const uint32_t* root = tr == T_E ? sh.key_root : w + (tr == T_T ? wt.o_troot : tr == T_P ? wt.o_proot : wt.o_qroot);
                uint32_t index[16];
                for (int j = 0; j < 16; j++) index[j] = wt.indices[(size_t)(q0 + (j < cnt ? j : 0))] >> (H - hs[0]);
                static thread_local uint32_t dg[32][8][16];
                size_t r = 0;
                for (int h : so) {
                    const uint32_t words = sh.leaf[tr][h].words, nb = (words + 7) / 8;
                    for (int e = 0; e < 16; e++) for (int j = 0; j < 16; j++) st[e][j] = 0u;
                    for (uint32_t b = 0; b < nb; b++) {
                        const uint32_t k = words - 8 * b < 8 ? words - 8 * b : 8u;
                        for (int j = 0; j < 16; j++) {
                            const int q = q0 + (j < cnt ? j : 0);
                            for (uint32_t e = 0; e < k; e++) st[e][j] = to_monty(leaf_word(q, tr, h, 8 * b + e));
                            bits[j] = 0; kps[j] = (b == nb - 1 && h == hs[0]) ? 2u * index[j] : 0u;
                        }
                        emit(sh.p2_tree0[tr], sh.tree_rows[tr], r++);
                    }
                    for (int e = 0; e < 8; e++) for (int j = 0; j < 16; j++) dg[h][e][j] = st[e][j];
                }
                for (int e = 0; e < 8; e++) for (int j = 0; j < 16; j++) cur[e][j] = dg[hs[0]][e][j];
                for (int lvl = 0; lvl < hs[0]; lvl++) {
                    for (int j = 0; j < 16; j++) {
                        const uint32_t* path = w + qp[(size_t)(q0 + (j < cnt ? j : 0))].path[tr];
                        const uint32_t b = (index[j] >> lvl) & 1u;
                        for (int e = 0; e < 8; e++) { const uint32_t sv = to_monty(path[8 * (size_t)lvl + (size_t)e]); st[b ? 8 + e : e][j] = cur[e][j]; st[b ? e : 8 + e][j] = sv; }
                        bits[j] = b; kps[j] = index[j] >> lvl;
                    }
                    emit(sh.p2_tree0[tr], sh.tree_rows[tr], r++);
                    for (int e = 0; e < 8; e++) for (int j = 0; j < 16; j++) cur[e][j] = st[e][j];
                    const int h = hs[0] - lvl - 1;
                    if (h != hs[0] && sh.has_h(tr, h)) {
                        for (int e = 0; e < 8; e++) for (int j = 0; j < 16; j++) { st[e][j] = cur[e][j]; st[8 + e][j] = dg[h][e][j]; }
                        for (int j = 0; j < 16; j++) { bits[j] = 0; kps[j] = index[j] >> (lvl + 1); }
                        emit(sh.p2_tree0[tr], sh.tree_rows[tr], r++);
                        for (int e = 0; e < 8; e++) for (int j = 0; j < 16; j++) cur[e][j] = st[e][j];
                    }
                }
                if (r != sh.tree_rows[tr]) return 4;
                for (int j = 0; j < cnt; j++) for (int e = 0; e < 8; e++) if (from_monty(cur[e][j]) != root[e]) return 3;
            }
            return 0;
        };
        const int nth = sh.NP >= 16 ? 1 : 16 / sh.NP;
        if (x16) {
            for (int q = 0; q < Q; q++) qerr[(size_t)q] = fri_q(q);          // (the fold rows and the pairs: field arithmetic only)
            const int ng = (Q + 15) / 16;
            auto group = [&](int g) {
                const int q0 = 16 * g, cnt = Q - q0 < 16 ? Q - q0 : 16;
                int e;
                try { e = walk16(q0, cnt); } catch (...) { e = 4; }
                if (e) for (int j = 0; j < cnt; j++) if (!qerr[(size_t)(q0 + j)]) qerr[(size_t)(q0 + j)] = e;
            };
            if (nth == 1 || ng < 2) for (int g = 0; g < ng; g++) group(g);
            else {
                HostPool pool(nth < ng ? nth : ng);
                for (int g = 0; g < ng; g++) pool.submit([&group, g] { group(g); });
                pool.wait();
            }
        } else if (nth == 1 || Q < 4) for (int q = 0; q < Q; q++) { qerr[(size_t)q] = fri_q(q); if (!qerr[(size_t)q]) qerr[(size_t)q] = tree_q(q); }
        else {
            HostPool pool(nth);
            for (int t = 0; t < nth; t++) pool.submit([&, t] { for (int q = t; q < Q; q += nth) { qerr[(size_t)q] = fri_q(q); if (!qerr[(size_t)q]) qerr[(size_t)q] = tree_q(q); } });
            pool.wait();
        }
        for (int q = 0; q < Q; q++) {
            if (qerr[(size_t)q] == 1) return bad("a FRI layer opening does not end in the layer's root");
            if (qerr[(size_t)q] == 2) return bad("a fold chain does not end in the final value");
            if (qerr[(size_t)q] == 3) return bad("an opening does not end in its root");
            if (qerr[(size_t)q]) return fail(ZKHIP_ERR_INTERNAL, "prove_machine_verifier: row layout");
        }
    }
    sec("fold rows + Poseidon2 chains");
    }   // (!on_device)
    // ---- TS
    {
        for (int T = 0; T < sh.NTS; T++) {
            uint32_t* row = ht.ts.data() + (size_t)TS_MAIN * ((size_t)p * (size_t)sh.NTS + (size_t)T);
            for (int j = 0; j < 8; j++) row[j] = to_monty(tin[16 * (size_t)T + (size_t)j]);
            if (challenge_row(sh, T)) put_ext(row, 16, chal[(size_t)T]);
            if (T == sh.TG) put_ext(row, 20, chal2[(size_t)T]);
            if (T == sh.HL / 8) for (int j = 0; j < 8; j++) row[8 + j] = to_monty(w[wt.o_troot + (size_t)j]);
        }
    }
    sec("ts");
    return ZKHIP_OK;
}
}  // namespace
}  // namespace mrec
}  // namespace zk

// ---- the entries (their bodies live in the namespace: its names hide shard_verifier.inl's of the same spelling)
namespace zk {
namespace mrec {

// ---- the per-query tables on the device (WitPlan above): what fill_proof's ROWSUM / QUERY / FOLD sections compute on the host, word for word
struct WitArgs {
    const uint32_t* proofs; uint64_t proof_words;       // [NP][proof_words] canonical
    const uint32_t* vals; uint64_t vals_stride;         // DevVals per proof
    uint32_t v_betas, v_fa, v_zeta, v_h, v_pw;
    const WRow* rows; const uint32_t* g0; const int32_t* hs; const int32_t* inj_hi;
    uint32_t nh, nq, NP, Q, R, H, NTREES;
    uint32_t o_queries, per_query, o_final, fri_off;
    uint32_t* rs; uint32_t* qt; uint32_t* fold; uint32_t fw, inj_col;
    uint32_t* roh;                                       // [NP][Q][nh][4]
    uint32_t* pairs; uint32_t* pair_k;                   // [NP Q R][8], [NP Q R]
    uint32_t* err;
};
__device__ __forceinline__ Ext w_ld4(const uint32_t* p) { return Ext{{p[0], p[1], p[2], p[3]}}; }
__device__ __forceinline__ Ext w_ext_at(const uint32_t* p) { return Ext{{to_monty(p[0]), to_monty(p[1]), to_monty(p[2]), to_monty(p[3])}}; }
__device__ __forceinline__ void w_put(uint32_t* r, uint32_t col, const Ext& e) { r[col] = e.c[0]; r[col + 1] = e.c[1]; r[col + 2] = e.c[2]; r[col + 3] = e.c[3]; }
// one lane per (proof, query, height): the height's ROWSUM rows (Horner from the back over the row words the sponge rows hash), then its QUERY row
__global__ void __launch_bounds__(64) mrec_rowsum_query_kernel(WitArgs a) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)a.NP * a.Q * a.nh) return;
    const uint32_t hi = (uint32_t)(gid % a.nh), q = (uint32_t)((gid / a.nh) % a.Q), p = (uint32_t)(gid / ((uint64_t)a.nh * a.Q));
    const uint32_t* w = a.proofs + (uint64_t)p * a.proof_words;
    const uint32_t* qw = w + a.o_queries + (uint64_t)q * a.per_query;
    const uint32_t* v = a.vals + (uint64_t)p * a.vals_stride;
    const Ext fa = w_ld4(v + a.v_fa), zeta = w_ld4(v + a.v_zeta);
    Ext acc = ext_zero(), az = ext_zero(), an = ext_zero();
    for (uint32_t i = a.g0[hi]; i < a.g0[hi + 1]; i++) {
        const WRow& rr = a.rows[i];
        uint32_t* r = a.rs + (uint64_t)RS_MAIN * (((uint64_t)p * a.Q + q) * a.nq + i);
        uint32_t vals[8];
        for (int j = 0; j < 8; j++) { const int32_t o = rr.src[j]; vals[j] = o < 0 ? 0u : to_monty(qw[o]); r[RS_V + j] = vals[j]; }
        w_put(r, RS_FA, fa); w_put(r, RS_ACCIN, acc); w_put(r, RS_AZIN, az); w_put(r, RS_ANIN, an);
        Ext t0 = ext_zero(), t4 = ext_zero(), prev = acc;
        for (int sidx = 7; sidx >= 0; sidx--) {
            const bool restart = (sidx == 7 && (rr.flags & 1u)) || (sidx == 3 && (rr.flags & 2u));
            const Ext carried = restart ? ext_zero() : ext_mul(prev, fa);
            prev = ext_add_base(carried, vals[sidx]);
            w_put(r, RS_T + 4u * (uint32_t)sidx, prev);
            if (sidx == 4) t4 = prev;
            if (sidx == 0) t0 = prev;
        }
        acc = t0;
        const uint32_t kzc[2] = {RS_KZ0, RS_KZ4}, knc[2] = {RS_KN0, RS_KN4};
        for (int sidx = 0; sidx < 2; sidx++) {
            if (rr.kz[sidx] < 0) continue;
            const Ext st = sidx ? t4 : t0;
            const Ext kzv = w_ld4(v + a.v_pw + 4u * (uint32_t)rr.kz[sidx]);
            w_put(r, kzc[sidx], kzv);
            az = ext_add(az, ext_mul(kzv, st));
            if (rr.kn[sidx] >= 0) { const Ext knv = w_ld4(v + a.v_pw + 4u * (uint32_t)rr.kn[sidx]); w_put(r, knc[sidx], knv); an = ext_add(an, ext_mul(knv, st)); }
        }
        w_put(r, RS_AZO, az); w_put(r, RS_ANO, an);
    }
    // ---- the QUERY row of (query, height)
    const int h = a.hs[hi];
    const uint32_t idx0 = v[q], ic = idx0 >> (a.H - (uint32_t)h);
    const uint32_t xq = fpow(two_adic_generator(h), reverse_bits(ic, h));
    const Ext x = ext_from_base(fmul(MONTY_GEN, xq));
    const Ext znx = w_ld4(v + a.v_h + 12u * hi), yz = w_ld4(v + a.v_h + 12u * hi + 4), yn = w_ld4(v + a.v_h + 12u * hi + 8);
    const Ext i1 = ext_inv(ext_sub(x, zeta)), i2 = ext_inv(ext_sub(x, znx));
    const Ext p1 = ext_mul(ext_sub(az, yz), i1), p2 = ext_mul(ext_sub(an, yn), i2), ro = ext_add(p1, p2);
    uint32_t* r = a.qt + (uint64_t)Q_MAIN * (((uint64_t)p * a.Q + q) * a.nh + hi);
    r[QM_IDX - Q_PRE] = to_monty(ic); r[QM_XQ - Q_PRE] = xq; r[QM_IDX0 - Q_PRE] = to_monty(idx0);
    w_put(r, QM_RO - Q_PRE, ro); w_put(r, QM_AZ - Q_PRE, az); w_put(r, QM_AN - Q_PRE, an); w_put(r, QM_YZ - Q_PRE, yz); w_put(r, QM_YN - Q_PRE, yn);
    w_put(r, QM_ZETA - Q_PRE, zeta); w_put(r, QM_ZNX - Q_PRE, znx); w_put(r, QM_I1 - Q_PRE, i1); w_put(r, QM_I2 - Q_PRE, i2); w_put(r, QM_P1 - Q_PRE, p1); w_put(r, QM_P2 - Q_PRE, p2);
    w_put(a.roh, 4u * (uint32_t)((((uint64_t)p * a.Q + q) * a.nh + hi)), ro);
}
// the host's running value is ONE variable over all rows of a proof: the first row of a (query, height) group carries, in ACCIN, the last row of the
// group before it (a column no constraint reads there: Horner restarts) -- copied here so that the tables are the host's word for word
__global__ void __launch_bounds__(64) mrec_rowsum_accin_kernel(WitArgs a) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)a.NP * a.Q * a.nh) return;
    const uint32_t hi = (uint32_t)(gid % a.nh), q = (uint32_t)((gid / a.nh) % a.Q), p = (uint32_t)(gid / ((uint64_t)a.nh * a.Q));
    if (a.g0[hi] == a.g0[hi + 1]) return;
    const uint64_t local = (uint64_t)q * a.nq + a.g0[hi];
    if (local == 0) return;
    uint32_t* r = a.rs + (uint64_t)RS_MAIN * ((uint64_t)p * a.Q * a.nq + local);
    const uint32_t* before = r - RS_MAIN;
    for (int j = 0; j < 4; j++) r[RS_ACCIN + j] = before[RS_T + j];
}
// one lane per (proof, query): the fold chain's rows, the layers' pairs for the Poseidon2 chains, the final value
__global__ void __launch_bounds__(64) mrec_fold_kernel(WitArgs a) {
    using namespace frichip;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)a.NP * a.Q) return;
    const uint32_t q = (uint32_t)(gid % a.Q), p = (uint32_t)(gid / a.Q);
    const uint32_t* w = a.proofs + (uint64_t)p * a.proof_words;
    const uint32_t* qw = w + a.o_queries + (uint64_t)q * a.per_query;
    const uint32_t* v = a.vals + (uint64_t)p * a.vals_stride;
    const uint32_t idx0 = v[q], INJ = a.inj_col, INJF = INJ + 4;
    uint32_t idx = idx0;
    const uint32_t* rohq = a.roh + 4u * (uint32_t)(((uint64_t)p * a.Q + q) * a.nh);
    Ext own = w_ld4(rohq);                                     // the tallest height's reduced opening (sh.hs[0] = H)
    uint32_t tcol[MAX_LAYERS];
    uint64_t fat = a.fri_off;
    const int H = (int)a.H, R = (int)a.R;
    for (int l = 0; l < R; l++) {
        const uint64_t at = ((uint64_t)p * a.Q + q) * (uint64_t)R + (uint64_t)l;
        uint32_t* row = a.fold + (uint64_t)a.fw * at;
        row[INJF + 1] = to_monty(idx0);
        if (l > 0 && a.inj_hi[l] >= 0) { const Ext jv = w_ld4(rohq + 4 * a.inj_hi[l]); w_put(row, INJ, jv); row[INJF] = MONTY_R1; own = ext_add(own, jv); }
        const uint32_t bit = idx & 1u, k = idx >> 1;
        const Ext sib = w_ext_at(qw + fat), beta = w_ld4(v + a.v_betas + 4u * (uint32_t)l);
        fat += 4 + 8 * (uint64_t)(H - 1 - l);
        const Ext e0 = bit ? sib : own, e1 = bit ? own : sib;
        const int lhh = H - (l + 1);
        const uint32_t x = fpow(two_adic_generator(lhh + 1), reverse_bits(k, lhh)), xi = finv(x);
        const Ext even = ext_mul_base(ext_add(e0, e1), MONTY_INV2), odd = ext_mul_base(ext_sub(e0, e1), fmul(MONTY_INV2, xi)), fold = ext_add(even, ext_mul(beta, odd));
        w_put(row, E0, e0); w_put(row, E1, e1); w_put(row, BETA, beta); w_put(row, FOLD, fold);
        row[BIT] = bit ? MONTY_R1 : 0u; row[K] = to_monty(k); row[X] = x; row[XI] = xi; row[S] = fmul(x, x);
        tcol[l] = bit ? two_adic_generator(l + 1) : MONTY_R1;
        row[frichip::T] = tcol[l]; row[ACTIVE] = MONTY_R1; row[LN] = to_monty((uint32_t)l); row[L_REC + (uint32_t)l] = MONTY_R1;
        if (l + 1 < R) { row[G] = MONTY_R1; row[GS] = row[S]; row[GT] = tcol[l]; }
        w_put(row, OWN, own);
        row[K2] = to_monty(2u * k); row[IDX] = to_monty(2u * k + bit);
        row[XS] = bit ? fsub(0u, x) : x; row[frichip::PT] = to_monty(p * a.NTREES); row[LNX] = to_monty(p * a.NTREES + (uint32_t)l);
        uint32_t* pw8 = a.pairs + 8 * at;
        for (int j = 0; j < 4; j++) { pw8[j] = e0.c[j]; pw8[4 + j] = e1.c[j]; }
        a.pair_k[at] = k;
        own = fold; idx = k;
    }
    uint32_t bacc = idx ? two_adic_generator(R + 1) : MONTY_R1;
    for (int l = R - 1; l >= 0; l--) {
        bacc = fmul(bacc, tcol[l]);
        a.fold[(uint64_t)a.fw * (((uint64_t)p * a.Q + q) * (uint64_t)R + (uint64_t)l) + frichip::B] = bacc;
    }
    bool ok = true;
    for (int j = 0; j < 4; j++) ok = ok && from_monty(own.c[j]) == w[a.o_final + (uint32_t)j];
    if (!ok) atomicCAS(a.err + p, 0u, 2u);
}
// the fold chip's padding rows: T = 1 (what the host table starts from)
__global__ void __launch_bounds__(256) mrec_fold_pad_kernel(uint32_t* fold, uint32_t fw, uint64_t row0, uint64_t rows) {
    const uint64_t r = row0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) fold[(uint64_t)fw * r + frichip::T] = MONTY_R1;
}

int m_machine_verifier_setup(zkhip_ctx* ctx, const zkhip_machine_desc* inner, size_t n_proofs, const zkhip_params* outer, zkhip_machine_key** key, uint32_t vk[8]) {
    CHECK_CTX(ctx);
    if (!outer || !key || !vk) return fail(ZKHIP_ERR_INVALID, "machine_verifier_setup: null argument");
    int rc = ZKHIP_OK;
    const auto mp = machine_of(inner, n_proofs, &rc);
    if (!mp) return rc;
    const Machine& m = *mp;
    std::vector<uint32_t> pre[N_CHIPS];
    all_pre(m, pre);
    size_t total = 0;
    for (int c = 0; c < N_CHIPS; c++) total += pre[c].size();
    void* d;
    ZK_TRY(ctx_reserve(ctx, S_REC_A, total * 4, &d));
    zkhip_chip chips[N_CHIPS]{};
    size_t at = 0;
    for (int i = 0; i < N_CHIPS; i++) {
        const int c = m.order[i];
        chips[i].log_n = m.height[c]; chips[i].width = m.pre_widths[i]; chips[i].ld = m.pre_widths[i]; chips[i].partner = -1;
        if (pre[c].empty()) continue;
        ZK_TRY(dev_h2d(ctx, (uint32_t*)d + at, pre[c].data(), pre[c].size() * 4));
        chips[i].d_trace = (const uint32_t*)d + at;
        at += pre[c].size();
    }
    return zkhip_machine_setup(ctx, chips, N_CHIPS, outer, key, vk);
}
int m_machine_verifier_key_host(const zkhip_machine_desc* inner, size_t n_proofs, const zkhip_params* outer, uint32_t vk[8]) {
    try {
        if (!outer || !vk) return fail(ZKHIP_ERR_INVALID, "machine_verifier_key_host: null argument");
        int rc = ZKHIP_OK;
        const auto mp = machine_of(inner, n_proofs, &rc);
        if (!mp) return rc;
        const Machine& m = *mp;
        std::vector<uint32_t> pre[N_CHIPS];
        all_pre(m, pre);
        const uint32_t* traces[N_CHIPS]; int32_t lns[N_CHIPS]; uint32_t pws[N_CHIPS];
        for (int i = 0; i < N_CHIPS; i++) {
            const int c = m.order[i];
            lns[i] = m.height[c]; pws[i] = pre[c].empty() ? 0u : m.pre_widths[i];
            traces[i] = pre[c].empty() ? nullptr : pre[c].data();
        }
        return zkhip_machine_key_host(traces, lns, pws, N_CHIPS, outer, vk);
    } catch (const std::bad_alloc&) {
        return fail(ZKHIP_ERR_NOMEM, "machine_verifier_key_host: out of host memory");
    }
}
size_t m_machine_verifier_proof_size(const zkhip_machine_desc* inner, size_t n_proofs, const zkhip_params* outer) {
    int rc = ZKHIP_OK;
    const auto mp = machine_of(inner, n_proofs, &rc);
    if (!mp || !outer) return 0;
    return zkhip_machine_proof_size_keyed(mp->log_ns, mp->widths, mp->pre_widths, mp->progs, mp->prog_words, mp->tabs, mp->tab_words, N_CHIPS, outer, mp->sh.npub_total());
}
// what the machine is made of, for tests and for a verifier that wants to look: position `which` of the ten chips (tallest first): kind 0 its program,
// 1 its interaction table, 2 its preprocessed trace (canonical words).  Returns the word count; out may be null
size_t m_machine_verifier_describe(const zkhip_machine_desc* inner, size_t n_proofs, int which, int kind, uint32_t* out, size_t cap, int* log_rows, uint32_t* main_width,
                                       uint32_t* pre_width) {
    int rc = ZKHIP_OK;
    const auto mp = machine_of(inner, n_proofs, &rc);
    if (!mp || which < 0 || which >= N_CHIPS || kind < 0 || kind > 2) return 0;
    const Machine& m = *mp;
    if (log_rows) *log_rows = m.log_ns[which];
    if (main_width) *main_width = m.widths[which];
    if (pre_width) *pre_width = m.pre_widths[which];
    std::vector<uint32_t> pre;
    const std::vector<uint32_t>* src = kind == 0 ? &m.prog[which] : &m.tab[which];
    if (kind == 2) {
        const int c = m.order[which], h = m.height[c];
        const MShape& sh = m.sh;
        switch (c) {
            case C_P2R: p2r_pre(sh, h, pre); break; case C_ROWSUM: rowsum_pre(sh, h, pre); break; case C_TS: ts_pre(sh, h, pre); break; case C_QUERY: query_pre(sh, h, pre); break;
            case C_OPENED: opened_pre(sh, h, pre); break; case C_SAMPLES: samples_pre_all(sh, h, pre); break; case C_SCALARS: scalars_pre(sh, h, pre); break;
            case C_EVAL: eval_pre(sh, h, pre); break; case C_LOGUP: logup_pre(sh, h, pre); break; default: break;
        }
        for (uint32_t& v : pre) v = from_monty(v);
        src = &pre;
    }
    if (out && cap >= src->size()) std::memcpy(out, src->data(), src->size() * 4);
    return src->size();
}

// The top in three steps, so that a caller that makes the inner proofs itself (zkhip_prove_shard_tree: the joins of a tree) fills a proof's tables the
// moment it exists, beside the others still being proven: begin (the machine, zeroed tables), fill (one inner proof -- its tables ARE its verification;
// distinct proofs may be filled from distinct threads), finish (uploads, the Poseidon2 columns, the machine's proof).
struct TopSession {
    std::shared_ptr<const Machine> mp;
    HostTabs ht;
    size_t used = 0;
    // round 6: the per-query tables are the device's (WitPlan); the host keeps a proof's transcript and the tables that follow its opened values once
    bool device = true;
    WitPlan plan;
    std::vector<DevVals> vals;
    std::vector<const uint8_t*> ptrs;
    size_t proof_len = 0;
    std::vector<Transcript> transcripts;          // filled by top_transcripts (all proofs at hand: sixteen chains per permutation); empty / not done: fill_proof walks its own
};
inline bool rec_host_forced() { return zk::rec::witnesses_on_host(); }      // (zkhip_recursion_witnesses_on_host: shard_verifier.inl)
int top_begin(const zkhip_machine_desc* inner, size_t n_proofs, size_t n_public, TopSession& s) {
    int rc = ZKHIP_OK;
    s.mp = machine_of(inner, n_proofs, &rc);
    if (!s.mp) return rc;
    const Machine& m = *s.mp;
    const MShape& sh = m.sh;
    if ((int)n_public != sh.NPUB) return fail(ZKHIP_ERR_INVALID, "prove_machine_verifier: n_public is not the machine's");
    HostTabs& ht = s.ht;
    s.device = !rec_host_forced() && !t_batcher;            // (inside a lock-step batch the members' launches are merged: the witness kernels are launched directly, so there the host walks)
    if (s.device) { s.plan = build_wit_plan(sh); s.vals.assign((size_t)sh.NP, DevVals{}); s.ptrs.assign((size_t)sh.NP, nullptr); }
    ZeroedWords* tabs[N_CHIPS] = {nullptr, &ht.rs, &ht.fold, &ht.ts, &ht.q, &ht.op, &ht.sm, &ht.sc, &ht.evl, &ht.lgu};
    for (int c = 0; c < N_CHIPS; c++) {
        if (!tabs[c] || (s.device && (c == C_ROWSUM || c == C_FOLD || c == C_QUERY))) continue;       // (device mode: those three never exist on the host)
        if (!tabs[c]->reset((size_t)m.w_main[c] << m.height[c])) return fail(ZKHIP_ERR_NOMEM, "prove_machine_verifier: no host memory for the machine's tables");
    }
    if (!s.device) for (size_t r = 0; r < ((size_t)1 << m.height[C_FOLD]); r++) ht.fold.data()[(size_t)m.w_main[C_FOLD] * r + frichip::T] = MONTY_R1;      // (the fold chip's padding rows: T = 1)
    s.used = (size_t)sh.NP * sh.p2_rows;
    const size_t walked = s.device ? (size_t)sh.NP * (size_t)sh.NT : s.used;                          // Poseidon2 rows whose input states the host walks
    if (!ht.p2_in.reset(16 * walked) || !ht.p2_bit.reset(walked) || !ht.p2_kp.reset(walked)) return fail(ZKHIP_ERR_NOMEM, "prove_machine_verifier: no host memory");
    return ZKHIP_OK;
}
// (any thread, a pool's included: nothing unwinds out of it; the message stays in this thread's zkhip_last_error)
int top_fill(TopSession& s, int p, const uint8_t* proof, size_t proof_len, const uint32_t* pubs) {
    try {
        if (!proof) return fail(ZKHIP_ERR_INVALID, "prove_machine_verifier: null proof");
        if (s.device) {
            if (s.proof_len == 0) s.proof_len = proof_len;                 // (every proof of one machine has one length: fill_proof checks it)
            s.ptrs[(size_t)p] = proof;
            return fill_proof(*s.mp, p, proof, proof_len, pubs, s.ht, &s.plan, &s.vals[(size_t)p], s.transcripts.empty() ? nullptr : &s.transcripts[(size_t)p]);
        }
        return fill_proof(*s.mp, p, proof, proof_len, pubs, s.ht, nullptr, nullptr, s.transcripts.empty() ? nullptr : &s.transcripts[(size_t)p]);
    } catch (const std::bad_alloc&) { return fail(ZKHIP_ERR_NOMEM, "prove_machine_verifier: out of host memory"); }
      catch (const std::exception& e) { return fail(ZKHIP_ERR_INTERNAL, std::string("prove_machine_verifier: ") + e.what()); }
}
// every proof at hand before the fills start: their transcripts in groups of sixteen, one AVX-512 permutation per row of the group (a few threads when there are
// several groups).  Proofs of another length are left to fill_proof, which refuses them.
void top_transcripts(TopSession& s, const uint8_t* const* proofs, const size_t* proof_lens, const uint32_t* public_values, size_t n_public) {
    const MShape& sh = s.mp->sh;
    if (sh.NP < 8 || !p2x16_available()) return;               // (fewer proofs than half the lanes: a thread per proof walks its own chain sooner -- 2.2 against 2.8 ms for the tree's four joins)
    const ProofPos pp = proof_pos(sh);
    s.transcripts.assign((size_t)sh.NP, Transcript{});
    std::vector<int> good;
    for (int p = 0; p < sh.NP; p++) if (proofs[p] && proof_lens[p] == pp.words * 4) good.push_back(p);
    const size_t stride = s.device ? (size_t)sh.NT : sh.p2_rows;
    const int ng = ((int)good.size() + 15) / 16;
    auto group = [&](int g) {
        const uint32_t* ws[16]; const uint32_t* pubss[16]; int ps[16]; Transcript* out[16];
        int n = 0;
        for (int i = 16 * g; i < (int)good.size() && n < 16; i++, n++) {
            const int p = good[(size_t)i];
            ws[n] = (const uint32_t*)proofs[p]; pubss[n] = public_values + (size_t)p * n_public; ps[n] = p; out[n] = &s.transcripts[(size_t)p];
        }
        try { if (n) walk_transcripts_x16(sh, n, ws, pubss, ps, stride, s.ht, out); } catch (...) { for (int i = 0; i < n; i++) out[i]->done = false; }
    };
    if (ng <= 1) { if (ng) group(0); return; }
    HostPool pool(ng < 8 ? ng : 8);
    for (int g = 0; g < ng; g++) pool.submit([&group, g] { group(g); });
    pool.wait();
}
template <class Lap>
int top_finish(zkhip_ctx* ctx, const zkhip_machine_key* key, TopSession& s, const uint32_t* public_values, size_t n_public, const zkhip_params* outer, uint8_t* proof, size_t cap,
               size_t* len, Lap&& lap) {
    const Machine& m = *s.mp;
    const MShape& sh = m.sh;
    const int NP = sh.NP;
    HostTabs& ht = s.ht;
    const size_t used = s.used;
    ZeroedWords* tabs[N_CHIPS] = {nullptr, &ht.rs, &ht.fold, &ht.ts, &ht.q, &ht.op, &ht.sm, &ht.sc, &ht.evl, &ht.lgu};
    void* dev[N_CHIPS] = {nullptr};
    const int slots[N_CHIPS] = {S_REC_A, S_REC_C, S_REC_B, S_REC_D, S_REC_E, S_REC_F, S_REC_G, S_REC_H, S_REC_I, S_REC_J};
    for (int c = 0; c < N_CHIPS; c++) ZK_TRY(ctx_reserve(ctx, slots[c], ((size_t)m.w_main[c] << m.height[c]) * 4, &dev[c]));
    if (s.device) {
        // ---- the per-query tables on the device: the inner proofs' words up once, then four launches (ROWSUM + QUERY rows, the fold chains, the
        // Poseidon2 chains) fill ROWSUM, QUERY, FOLD and the queries' rows of the Poseidon2 chip in place
        const WitPlan& pl = s.plan;
        const size_t NPs = (size_t)NP, Q = (size_t)sh.Q, R = (size_t)sh.R, nh = sh.hs.size(), nq = pl.rows.size();
        const size_t pwords = s.proof_len / 4, vstride = dv_words(sh, pl);
        // positions in a proof (fill_proof's)
        size_t pos = (size_t)sh.HL + 2;
        const size_t o_troot = pos; pos += 8; const size_t o_proot = pos; pos += 8; pos += 4 * (size_t)sh.C; const size_t o_qroot = pos; pos += 8; pos += 4 * (size_t)sh.NV;
        const size_t o_lroots = pos; pos += 8 * R; const size_t o_final = pos; pos += 4; pos += 1; const size_t o_queries = pos;
        size_t perq = 0;
        for (int t = 0; t < N_TREES; t++) if (sh.has_tree[t]) { for (int c : sh.tree_chips[t]) perq += (*sh.tree_w[t])[(size_t)c]; perq += 8 * (size_t)sh.tree_hs[t][0]; }
        for (int l = 0; l < sh.R; l++) perq += 4 + 8 * (size_t)(sh.H - 1 - l);
        if (pwords > 0xFFFFFFFFull || o_queries + Q * perq != pwords) return fail(ZKHIP_ERR_INTERNAL, "prove_machine_verifier: proof layout");
        std::vector<p2chip::MrecTreePlan> trees = pl.trees;
        {
            size_t i = 0;
            for (int tr = 0; tr < N_TREES; tr++) { if (!sh.has_tree[tr]) continue; trees[i++].root_off = tr == T_E ? -1 : (int32_t)(tr == T_T ? o_troot : tr == T_P ? o_proot : o_qroot); }
        }
        // one block of small things: [vals NP][rows][g0][hs][inj_hi][trees][src], then scratch [roh][pairs][pair_k][err]
        auto rup = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t b_vals = rup(NPs * vstride * 4), b_rows = rup(nq * sizeof(WRow)), b_g0 = rup((nh + 1) * 4), b_hs = rup(nh * 4), b_inj = rup(32 * 4),
                     b_trees = rup(trees.size() * sizeof(p2chip::MrecTreePlan)), b_src = rup(pl.src.size() * 4 + 4);
        const size_t up_bytes = b_vals + b_rows + b_g0 + b_hs + b_inj + b_trees + b_src;
        const size_t b_roh = rup(NPs * Q * nh * 16), b_pairs = rup(NPs * Q * R * 32), b_pk = rup(NPs * Q * R * 4), b_err = rup(NPs * 4);
        void *dprf, *dblk, *hpin;
        ZK_TRY(ctx_reserve(ctx, S_WIT_A, NPs * s.proof_len, &dprf));
        ZK_TRY(ctx_reserve(ctx, S_WIT_B, up_bytes + b_roh + b_pairs + b_pk + b_err, &dblk));
        ZK_TRY(ctx_host_pinned(ctx, NPs * s.proof_len + up_bytes, &hpin));
        uint8_t* hp = (uint8_t*)hpin;
        {   // the proofs into the pinned block side by side (a few threads: 60 MB for 64 transcript proofs), the small block behind them
            const int nthr = NP < 8 ? NP : 8;
            if (nthr <= 1) for (int p = 0; p < NP; p++) std::memcpy(hp + (size_t)p * s.proof_len, s.ptrs[(size_t)p], s.proof_len);
            else {
                HostPool pool(nthr);
                for (int p = 0; p < NP; p++) pool.submit([&, p] { std::memcpy(hp + (size_t)p * s.proof_len, s.ptrs[(size_t)p], s.proof_len); });
                pool.wait();
            }
            uint8_t* u = hp + NPs * s.proof_len;
            std::memset(u, 0, up_bytes);
            for (int p = 0; p < NP; p++) std::memcpy(u + (size_t)p * vstride * 4, s.vals[(size_t)p].v.data(), vstride * 4);
            size_t at = b_vals;
            std::memcpy(u + at, pl.rows.data(), nq * sizeof(WRow)); at += b_rows;
            std::memcpy(u + at, pl.g0.data(), (nh + 1) * 4); at += b_g0;
            for (size_t i = 0; i < nh; i++) ((int32_t*)(u + at))[i] = sh.hs[i];
            at += b_hs;
            std::memcpy(u + at, pl.inj_hi, 32 * 4); at += b_inj;
            std::memcpy(u + at, trees.data(), trees.size() * sizeof(p2chip::MrecTreePlan)); at += b_trees;
            if (!pl.src.empty()) std::memcpy(u + at, pl.src.data(), pl.src.size() * 4);
        }
        uint8_t* db = (uint8_t*)dblk;
        ZK_HIP(hipMemcpyAsync(dprf, hp, NPs * s.proof_len, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipMemcpyAsync(db, hp + NPs * s.proof_len, up_bytes, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipMemsetAsync(db + up_bytes, 0, b_roh + b_pairs + b_pk + b_err, ctx->stream));
        ZK_HIP(hipMemsetAsync(dev[C_ROWSUM], 0, ((size_t)m.w_main[C_ROWSUM] << m.height[C_ROWSUM]) * 4, ctx->stream));
        ZK_HIP(hipMemsetAsync(dev[C_QUERY], 0, ((size_t)m.w_main[C_QUERY] << m.height[C_QUERY]) * 4, ctx->stream));
        ZK_HIP(hipMemsetAsync(dev[C_FOLD], 0, ((size_t)m.w_main[C_FOLD] << m.height[C_FOLD]) * 4, ctx->stream));
        WitArgs a{};
        a.proofs = (const uint32_t*)dprf; a.proof_words = pwords; a.vals = (const uint32_t*)db; a.vals_stride = vstride;
        a.v_betas = (uint32_t)Q; a.v_fa = a.v_betas + 4u * (uint32_t)R; a.v_zeta = a.v_fa + 4; a.v_h = a.v_zeta + 4; a.v_pw = a.v_h + 12u * (uint32_t)nh;
        size_t at = b_vals;
        a.rows = (const WRow*)(db + at); at += b_rows; a.g0 = (const uint32_t*)(db + at); at += b_g0; a.hs = (const int32_t*)(db + at); at += b_hs;
        a.inj_hi = (const int32_t*)(db + at); at += b_inj;
        const p2chip::MrecTreePlan* d_trees = (const p2chip::MrecTreePlan*)(db + at); at += b_trees;
        const int32_t* d_src = (const int32_t*)(db + at);
        a.nh = (uint32_t)nh; a.nq = (uint32_t)nq; a.NP = (uint32_t)NP; a.Q = (uint32_t)Q; a.R = (uint32_t)R; a.H = (uint32_t)sh.H; a.NTREES = sh.NTREES;
        a.o_queries = (uint32_t)o_queries; a.per_query = (uint32_t)perq; a.o_final = (uint32_t)o_final; a.fri_off = pl.fri_off;
        a.rs = (uint32_t*)dev[C_ROWSUM]; a.qt = (uint32_t*)dev[C_QUERY]; a.fold = (uint32_t*)dev[C_FOLD]; a.fw = m.w_main[C_FOLD]; a.inj_col = frichip::width_of(sh.R, true, true);
        a.roh = (uint32_t*)(db + up_bytes); a.pairs = (uint32_t*)(db + up_bytes + b_roh); a.pair_k = (uint32_t*)(db + up_bytes + b_roh + b_pairs);
        a.err = (uint32_t*)(db + up_bytes + b_roh + b_pairs + b_pk);
        const unsigned gq = (unsigned)((NPs * Q * nh + 63) / 64), gf = (unsigned)((NPs * Q + 63) / 64);
        hipLaunchKernelGGL(mrec_rowsum_query_kernel, dim3(gq), dim3(64), 0, ctx->stream, a);
        hipLaunchKernelGGL(mrec_rowsum_accin_kernel, dim3(gq), dim3(64), 0, ctx->stream, a);
        hipLaunchKernelGGL(mrec_fold_kernel, dim3(gf), dim3(64), 0, ctx->stream, a);
        {
            const uint64_t frows = (uint64_t)1 << m.height[C_FOLD], fused = NPs * Q * R;
            if (frows > fused) hipLaunchKernelGGL(mrec_fold_pad_kernel, dim3((unsigned)((frows - fused + 255) / 256)), dim3(256), 0, ctx->stream, a.fold, a.fw, fused, frows);
        }
        ZK_HIP(hipGetLastError());
        p2chip::MrecChainArgs ca{};
        ca.proofs = a.proofs; ca.proof_words = pwords; ca.vals = a.vals; ca.vals_stride = vstride; ca.trees = d_trees; ca.n_trees = (uint32_t)trees.size(); ca.src = d_src;
        ca.NP = a.NP; ca.Q = a.Q; ca.R = a.R; ca.H = a.H; ca.o_queries = a.o_queries; ca.per_query = a.per_query; ca.o_lroots = (uint32_t)o_lroots; ca.fri_off = pl.fri_off;
        ca.p2_rows = (uint32_t)sh.p2_rows; ca.p2_fri0 = (uint32_t)sh.p2_fri0; ca.fri_rows = (uint32_t)sh.fri_rows;
        ca.pairs = a.pairs; ca.pair_k = a.pair_k;
        for (int j = 0; j < 8; j++) ca.key_root[j] = sh.key_root[j];
        void* rowbuf;                                                     // every used row's (input state, bit, KP)
        ZK_TRY(ctx_reserve(ctx, S_WIT_C, used * 18 * 4, &rowbuf));
        ca.row_in = (uint32_t*)rowbuf; ca.row_bit = ca.row_in + 16 * used; ca.row_kp = ca.row_in + 17 * used; ca.err = a.err;
        ca.trace = (uint32_t*)dev[C_P2R]; ca.ld = P2_MAIN;              // sixteen lanes per chain write the queries' rows themselves (hash.hip mrec_chains16_kernel)
        ZK_HIP(launch_mrec_chains(ca, ctx->stream));
        if (!(ca.trace && ca.ld == p2chip::R_WIDTH)) {   // the one-lane form: 360 columns each from (state, bit, KP), one lane per row; the transcripts' rows are skipped here
            p2chip::P2RArgs qa{};
            qa.chain_inputs = ca.row_in; qa.trows = nullptr; qa.n_chains = 0; qa.n_transcript = (uint32_t)used;
            qa.rows = (uint64_t)1 << m.height[C_P2R]; qa.used_rows = qa.rows;      // (no padding rows from this launch)
            qa.trace = (uint32_t*)dev[C_P2R]; qa.ld = P2_MAIN; qa.roots = nullptr; qa.row_bits = ca.row_bit; qa.row_kps = ca.row_kp;
            qa.inputs_monty = 1; qa.seg_rows = (uint32_t)sh.p2_rows; qa.skip_first = (uint32_t)sh.NT;
            ZK_HIP(launch_p2r_rows(qa, ctx->stream));
        }
        // the transcripts' rows (walked on the host: a serial sponge chain per proof) and the padding rows
        const size_t nt = NPs * (size_t)sh.NT;
        void* stage;
        ZK_TRY(ctx_reserve(ctx, S_STAGE, (16 * nt + 3 * nt) * 4, &stage));
        uint32_t* d = (uint32_t*)stage;
        std::vector<uint32_t> trows(nt);
        for (size_t p = 0; p < NPs; p++) for (size_t T = 0; T < (size_t)sh.NT; T++) trows[p * (size_t)sh.NT + T] = (uint32_t)(p * sh.p2_rows + T);
        ZK_TRY(dev_h2d(ctx, d, ht.p2_in.data(), 16 * nt * 4));
        ZK_TRY(dev_h2d(ctx, d + 16 * nt, ht.p2_bit.data(), nt * 4));
        ZK_TRY(dev_h2d(ctx, d + 17 * nt, ht.p2_kp.data(), nt * 4));
        ZK_TRY(dev_h2d(ctx, d + 18 * nt, trows.data(), nt * 4));
        p2chip::P2RArgs pa{};
        pa.desc = nullptr; pa.data = nullptr; pa.chain_inputs = d; pa.trows = d + 18 * nt; pa.n_chains = 0; pa.n_transcript = (uint32_t)nt;
        pa.rows = (uint64_t)1 << m.height[C_P2R]; pa.used_rows = used; pa.trace = (uint32_t*)dev[C_P2R]; pa.ld = P2_MAIN; pa.roots = nullptr;
        pa.row_bits = d + 16 * nt; pa.row_kps = d + 17 * nt;
        ZK_HIP(launch_p2r_rows(pa, ctx->stream));
        std::vector<uint32_t> errs(NPs, 0u);
        ZK_TRY(dev_d2h(ctx, errs.data(), a.err, NPs * 4));
        for (size_t p = 0; p < NPs; p++) {
            if (!errs[p]) continue;
            const char* what = errs[p] == 1 ? "a FRI layer opening does not end in the layer's root" : errs[p] == 2 ? "a fold chain does not end in the final value" : "an opening does not end in its root";
            return fail(ZKHIP_ERR_VERIFY, "prove_machine_verifier: proof " + std::to_string(p) + " rejected: " + what);
        }
    } else {   // the Poseidon2 rows: input states, bits, indices up; one launch fills the 360 columns of every row
        const size_t words = 16 * used + 3 * used;
        void* stage;
        ZK_TRY(ctx_reserve(ctx, S_STAGE, words * 4, &stage));
        uint32_t* d = (uint32_t*)stage;
        std::vector<uint32_t> trows(used);
        for (size_t r = 0; r < used; r++) trows[r] = (uint32_t)r;
        ZK_TRY(dev_h2d(ctx, d, ht.p2_in.data(), 16 * used * 4));
        ZK_TRY(dev_h2d(ctx, d + 16 * used, ht.p2_bit.data(), used * 4));
        ZK_TRY(dev_h2d(ctx, d + 17 * used, ht.p2_kp.data(), used * 4));
        ZK_TRY(dev_h2d(ctx, d + 18 * used, trows.data(), used * 4));
        p2chip::P2RArgs a{};
        a.desc = nullptr; a.data = nullptr; a.chain_inputs = d; a.trows = d + 18 * used; a.n_chains = 0; a.n_transcript = (uint32_t)used;
        a.rows = (uint64_t)1 << m.height[C_P2R]; a.used_rows = used; a.trace = (uint32_t*)dev[C_P2R]; a.ld = P2_MAIN; a.roots = nullptr;
        a.row_bits = d + 16 * used; a.row_kps = d + 17 * used;
        ZK_HIP(launch_p2r_rows(a, ctx->stream));
    }
    lap("device: Poseidon2 rows");
    for (int c = 0; c < N_CHIPS; c++) if (tabs[c] && tabs[c]->size()) ZK_TRY(dev_h2d(ctx, dev[c], tabs[c]->data(), tabs[c]->size() * 4));
    lap("upload: host tables");
    zkhip_chip chips[N_CHIPS]{};
    for (int i = 0; i < N_CHIPS; i++) {
        const int c = m.order[i];
        chips[i].d_trace = (const uint32_t*)dev[c]; chips[i].ld = m.w_main[c]; chips[i].log_n = m.height[c]; chips[i].width = m.w_main[c]; chips[i].partner = -1;
    }
    std::vector<uint32_t> pv((size_t)NP * n_public);
    for (size_t i = 0; i < pv.size(); i++) pv[i] = public_values[i] % P;
    const int prc = zkhip_prove_machine_keyed(ctx, key, chips, m.progs, m.prog_words, m.tabs, m.tab_words, N_CHIPS, pv.data(), pv.size(), outer, proof, cap, len);
    lap("the machine's proof");
    return prc;
}

int m_prove_machine_verifier(zkhip_ctx* ctx, const zkhip_machine_key* key, const zkhip_machine_desc* inner, const uint8_t* const* proofs, const size_t* proof_lens, size_t n_proofs,
                                 const uint32_t* public_values, size_t n_public, const zkhip_params* outer, uint8_t* proof, size_t cap, size_t* len) {
    CHECK_CTX(ctx);
    if (!key || !inner || !proofs || !proof_lens || !outer || !proof || !len || (n_public && !public_values)) return fail(ZKHIP_ERR_INVALID, "prove_machine_verifier: null argument");
#ifdef ZKHIP_AB_HOOKS
    static const bool timing = getenv("ZKHIP_REC_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  [machine verifier] %-34s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
#else
    auto lap = [](const char*) {};
#endif
    TopSession s;
    ZK_TRY(top_begin(inner, n_proofs, n_public, s));
    const int NP = s.mp->sh.NP;
    top_transcripts(s, proofs, proof_lens, public_values, n_public);
    lap("host: transcripts (sixteen per permutation)");
    // the inner proofs side by side: each one's tables, which is its verification
    {
        std::vector<int> rcs((size_t)NP, ZKHIP_OK);
        std::vector<std::string> msgs((size_t)NP);
        auto one = [&](int p) {
#ifdef ZKHIP_AB_HOOKS
            const auto tv = std::chrono::steady_clock::now();
#endif
            const int r = top_fill(s, p, proofs[p], proof_lens[p], public_values + (size_t)p * n_public);
#ifdef ZKHIP_AB_HOOKS
            if (timing) std::fprintf(stderr, "    [machine verifier] proof %d: tables filled in %.2f ms\n", p, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv).count());
#endif
            rcs[(size_t)p] = r;
            if (r != ZKHIP_OK) msgs[(size_t)p] = zkhip_last_error();
        };
        if (NP == 1) one(0);
        else {
            HostPool pool(NP < 16 ? NP : 16);
            for (int p = 0; p < NP; p++) pool.submit([&one, p] { one(p); });
            pool.wait();
        }
        for (int p = 0; p < NP; p++) if (rcs[(size_t)p] != ZKHIP_OK) { set_error("proof " + std::to_string(p) + ": " + msgs[(size_t)p]); return rcs[(size_t)p]; }
    }
    lap("host: verify + witnesses + tables");
    const int prc = top_finish(ctx, key, s, public_values, n_public, outer, proof, cap, len, lap);
    s.ht.release_later();
    return prc;
}

// THE TREE in one call: the shard proofs of an execution -> joins of proofs_per_join (zkhip_prove_shard_verifier_batch's way: dealt over the devices,
// several in flight) -> ONE proof over the joins, on ctx.  A join's tables for the top are filled on its worker's thread the moment the join exists,
// beside the joins still being proven, so that after the last join only the uploads and the machine's proof remain.
int m_prove_shard_tree(zkhip_ctx* ctx, const zkhip_machine_key* top_key, const zkhip_machine_desc* join_machine, const int* devices, int n_devices,
                       const uint8_t* const* shard_proofs, const size_t* shard_proof_lens, size_t n_proofs, size_t proofs_per_join, int log_n, uint32_t width,
                       const uint32_t* public_values, size_t n_public, const zkhip_params* inner, const zkhip_params* join_outer, const zkhip_params* top_outer,
                       int in_flight_per_device, uint8_t* joined, size_t joined_stride, size_t* joined_lens, uint32_t join_vk[8], uint8_t* proof, size_t cap, size_t* len) {
    CHECK_CTX(ctx);
    if (!top_key || !join_machine || !top_outer || !proof || !len || !join_vk) return fail(ZKHIP_ERR_INVALID, "prove_shard_tree: null argument");
    if (proofs_per_join == 0 || n_proofs == 0 || n_proofs % proofs_per_join != 0) return fail(ZKHIP_ERR_INVALID, "prove_shard_tree: the number of shard proofs must be a positive multiple of proofs_per_join");
    if (joined_stride % 4 != 0) return fail(ZKHIP_ERR_INVALID, "prove_shard_tree: joined_stride must be a multiple of 4");
    const size_t J = proofs_per_join, n_joins = n_proofs / J, jpub = J * n_public;
    if (join_machine->n_public != jpub) return fail(ZKHIP_ERR_INVALID, "prove_shard_tree: the join machine's public values are not proofs_per_join x n_public");
    TopSession s;
    ZK_TRY(top_begin(join_machine, n_joins, jpub, s));
    const std::function<int(size_t, const uint8_t*, size_t)> on_join = [&](size_t j, const uint8_t* jp, size_t jl) {
        return top_fill(s, (int)j, jp, jl, public_values + j * jpub);
    };
    ZK_TRY(sv_prove_batch(devices, n_devices, shard_proofs, shard_proof_lens, n_proofs, J, log_n, width, public_values, n_public, inner, join_outer, in_flight_per_device, 0,
                          joined, joined_stride, joined_lens, join_vk, &on_join));
    if (std::memcmp(join_vk, join_machine->key_root, 32) != 0) return fail(ZKHIP_ERR_INVALID, "prove_shard_tree: the join machine's key is not the key of this shape");
    const int prc = top_finish(ctx, top_key, s, public_values, jpub, top_outer, proof, cap, len, [](const char*) {});
    s.ht.release_later();
    return prc;
}

// the machine's MAIN traces as the prover fills them on the host, for tests without a device: chip at position `which` (tallest first), canonical words,
// [2^log_rows][main width]; for the Poseidon2 chip the rows are (input state [16], direction bit, KP) x used rows -- its 360 columns are the device's.
// Returns the word count (0: refused -- zkhip_last_error says why, a bad inner proof included); out may be null
size_t m_machine_verifier_host_tables(const zkhip_machine_desc* inner, const uint8_t* const* proofs, const size_t* proof_lens, size_t n_proofs, const uint32_t* public_values,
                                      size_t n_public, int which, uint32_t* out, size_t cap) {
    int rc = ZKHIP_OK;
    const auto mp = machine_of(inner, n_proofs, &rc);
    if (!mp || !proofs || !proof_lens || which < 0 || which >= N_CHIPS || (int)n_public != mp->sh.NPUB || (n_public && !public_values)) { if (mp) (void)fail(ZKHIP_ERR_INVALID, "machine_verifier_host_tables: bad arguments"); return 0; }
    const Machine& m = *mp;
    const MShape& sh = m.sh;
    HostTabs ht;
    ZeroedWords* tabs[N_CHIPS] = {nullptr, &ht.rs, &ht.fold, &ht.ts, &ht.q, &ht.op, &ht.sm, &ht.sc, &ht.evl, &ht.lgu};
    for (int c = 0; c < N_CHIPS; c++) if (tabs[c] && !tabs[c]->reset((size_t)m.w_main[c] << m.height[c])) { (void)fail(ZKHIP_ERR_NOMEM, "machine_verifier_host_tables: no host memory"); return 0; }
    for (size_t r = 0; r < ((size_t)1 << m.height[C_FOLD]); r++) ht.fold.data()[(size_t)m.w_main[C_FOLD] * r + frichip::T] = MONTY_R1;
    const size_t used = (size_t)sh.NP * sh.p2_rows;
    if (!ht.p2_in.reset(16 * used) || !ht.p2_bit.reset(used) || !ht.p2_kp.reset(used)) { (void)fail(ZKHIP_ERR_NOMEM, "machine_verifier_host_tables: no host memory"); return 0; }
    try {
        for (int p = 0; p < sh.NP; p++) if (!proofs[p] || fill_proof(m, p, proofs[p], proof_lens[p], public_values + (size_t)p * n_public, ht) != ZKHIP_OK) return 0;
    } catch (const std::exception& e) { (void)fail(ZKHIP_ERR_NOMEM, std::string("machine_verifier_host_tables: ") + e.what()); return 0; }
    const int c = m.order[which];
    if (c == C_P2R) {
        const size_t n = 18 * used;
        if (out && cap >= n)
            for (size_t r = 0; r < used; r++) { std::memcpy(out + 18 * r, ht.p2_in.data() + 16 * r, 64); out[18 * r + 16] = ht.p2_bit.data()[r]; out[18 * r + 17] = ht.p2_kp.data()[r]; }
        return n;
    }
    const size_t n = tabs[c]->size();
    if (out && cap >= n) for (size_t i = 0; i < n; i++) out[i] = from_monty(tabs[c]->data()[i]);
    return n;
}

int m_verify_machine_recursive(const zkhip_machine_desc* inner, const uint8_t* proof, size_t len, const uint32_t* public_values, size_t n_public, size_t n_proofs, const uint32_t vk[8],
                                   const zkhip_params* outer, int* reason) {
    if (!proof || !vk || !outer || (n_public && !public_values)) { if (reason) *reason = 1; return fail(ZKHIP_ERR_VERIFY, "verify_machine_recursive: null argument"); }
    int rc = ZKHIP_OK;
    const auto mp = machine_of(inner, n_proofs, &rc);
    if (!mp || (int)n_public != mp->sh.NPUB) { if (reason) *reason = 1; return fail(ZKHIP_ERR_VERIFY, "verify_machine_recursive: not a machine this verifier takes (or another number of public values)"); }
    const Machine& m = *mp;
    std::vector<uint32_t> pv(n_proofs * n_public);
    for (size_t i = 0; i < pv.size(); i++) pv[i] = public_values[i] % P;
    return zkhip_verify_machine_keyed(proof, len, m.log_ns, m.widths, m.pre_widths, vk, m.progs, m.prog_words, m.tabs, m.tab_words, N_CHIPS, pv.data(), pv.size(), outer, reason);
}

}  // namespace mrec
}  // namespace zk

// (nothing unwinds across the C ABI)
#define ZK_MREC_GUARD(expr, on_error)                                                                                        \
    try { return expr; }                                                                                                     \
    catch (const std::bad_alloc&) { (void)fail(ZKHIP_ERR_NOMEM, "machine verifier: out of host memory"); return on_error; }  \
    catch (const std::exception& e) { (void)fail(ZKHIP_ERR_INTERNAL, std::string("machine verifier: ") + e.what()); return on_error; }
extern "C" {
int zkhip_machine_verifier_setup(zkhip_ctx* ctx, const zkhip_machine_desc* inner, size_t n_proofs, const zkhip_params* outer, zkhip_machine_key** key, uint32_t vk[8]) {
    ZK_MREC_GUARD(zk::mrec::m_machine_verifier_setup(ctx, inner, n_proofs, outer, key, vk), ZKHIP_ERR_NOMEM)
}
int zkhip_machine_verifier_key_host(const zkhip_machine_desc* inner, size_t n_proofs, const zkhip_params* outer, uint32_t vk[8]) { ZK_MREC_GUARD(zk::mrec::m_machine_verifier_key_host(inner, n_proofs, outer, vk), ZKHIP_ERR_NOMEM) }
size_t zkhip_machine_verifier_proof_size(const zkhip_machine_desc* inner, size_t n_proofs, const zkhip_params* outer) { ZK_MREC_GUARD(zk::mrec::m_machine_verifier_proof_size(inner, n_proofs, outer), 0) }
size_t zkhip_machine_verifier_describe(const zkhip_machine_desc* inner, size_t n_proofs, int which, int kind, uint32_t* out, size_t cap, int* log_rows, uint32_t* main_width,
                                       uint32_t* pre_width) {
    ZK_MREC_GUARD(zk::mrec::m_machine_verifier_describe(inner, n_proofs, which, kind, out, cap, log_rows, main_width, pre_width), 0)
}
int zkhip_prove_machine_verifier(zkhip_ctx* ctx, const zkhip_machine_key* key, const zkhip_machine_desc* inner, const uint8_t* const* proofs, const size_t* proof_lens, size_t n_proofs,
                                 const uint32_t* public_values, size_t n_public, const zkhip_params* outer, uint8_t* proof, size_t cap, size_t* len) {
    ZK_MREC_GUARD(zk::mrec::m_prove_machine_verifier(ctx, key, inner, proofs, proof_lens, n_proofs, public_values, n_public, outer, proof, cap, len), ZKHIP_ERR_NOMEM)
}
int zkhip_prove_shard_tree(zkhip_ctx* ctx, const zkhip_machine_key* top_key, const zkhip_machine_desc* join_machine, const int* devices, int n_devices,
                           const uint8_t* const* shard_proofs, const size_t* shard_proof_lens, size_t n_proofs, size_t proofs_per_join, int log_n, uint32_t width,
                           const uint32_t* public_values, size_t n_public, const zkhip_params* inner, const zkhip_params* join_outer, const zkhip_params* top_outer,
                           int in_flight_per_device, uint8_t* joined, size_t joined_stride, size_t* joined_lens, uint32_t join_vk[8], uint8_t* proof, size_t cap, size_t* len) {
    ZK_MREC_GUARD(zk::mrec::m_prove_shard_tree(ctx, top_key, join_machine, devices, n_devices, shard_proofs, shard_proof_lens, n_proofs, proofs_per_join, log_n, width, public_values,
                                               n_public, inner, join_outer, top_outer, in_flight_per_device, joined, joined_stride, joined_lens, join_vk, proof, cap, len), ZKHIP_ERR_NOMEM)
}
size_t zkhip_machine_verifier_host_tables(const zkhip_machine_desc* inner, const uint8_t* const* proofs, const size_t* proof_lens, size_t n_proofs, const uint32_t* public_values,
                                          size_t n_public, int which, uint32_t* out, size_t cap) {
    ZK_MREC_GUARD(zk::mrec::m_machine_verifier_host_tables(inner, proofs, proof_lens, n_proofs, public_values, n_public, which, out, cap), 0)
}
int zkhip_verify_machine_recursive(const zkhip_machine_desc* inner, const uint8_t* proof, size_t len, const uint32_t* public_values, size_t n_public, size_t n_proofs, const uint32_t vk[8],
                                   const zkhip_params* outer, int* reason) {
    ZK_MREC_GUARD(zk::mrec::m_verify_machine_recursive(inner, proof, len, public_values, n_public, n_proofs, vk, outer, reason), ZKHIP_ERR_NOMEM)
}
}  // extern "C"
