// verifier.cpp -- the host verifiers: `client.verify(&proof, &vk)` of the reference (crates/guest-prover-sp1/src/sp1.rs:120) for every
// proof version of this library (single matrix, multi-chip, constraint programs, machine, keyed machine), the FRI view the recursion
// chips are fed from, and the self-test of the batched host permutation.  No device code: a verifier runs on any host.
#include <chrono>
#include <functional>
#include <mutex>
#include <thread>

#include "proof_common.h"
#include "p2_x16.h"

using namespace zk;

extern "C" {

// ---------------------------------------------------------------- verifier (host CPU)
static bool verify_path(const uint32_t* root_m, int log_h, size_t index, const uint32_t* row_canon, size_t width,
                        const uint32_t* sibs_canon, int hw) {
    uint32_t cur[8];
    if (hw == 24) {
        // Poseidon2 width 24: sponge rate 16, compress(l, r) = permute(l || r || 0^8)[0..8]
        uint32_t s[24] = {0};
        size_t posn = 0;
        for (size_t i = 0; i < width; i++) {
            s[posn++] = to_monty(row_canon[i]);
            if (posn == 16) { p24_permute(s); posn = 0; }
        }
        if (posn) p24_permute(s);
        for (int i = 0; i < 8; i++) cur[i] = s[i];
        for (int lvl = 0; lvl < log_h; lvl++) {
            uint32_t t[24] = {0};
            const bool right = (index >> lvl) & 1;
            for (int i = 0; i < 8; i++) {
                const uint32_t sib = to_monty(sibs_canon[8 * lvl + i]);
                t[i] = right ? sib : cur[i];
                t[8 + i] = right ? cur[i] : sib;
            }
            p24_permute(t);
            for (int i = 0; i < 8; i++) cur[i] = t[i];
        }
    } else {
        uint32_t s[16] = {0};
        size_t posn = 0;
        for (size_t i = 0; i < width; i++) {
            s[posn++] = to_monty(row_canon[i]);
            if (posn == 8) { p2_permute(s); posn = 0; }
        }
        if (posn) p2_permute(s);
        for (int i = 0; i < 8; i++) cur[i] = s[i];
        for (int lvl = 0; lvl < log_h; lvl++) {
            uint32_t sib[8];
            for (int i = 0; i < 8; i++) sib[i] = to_monty(sibs_canon[8 * lvl + i]);
            if ((index >> lvl) & 1) p2_compress(sib, cur, cur);
            else p2_compress(cur, sib, cur);
        }
    }
    for (int i = 0; i < 8; i++) if (cur[i] != root_m[i]) return false;
    return true;
}

// ---- up to sixteen openings of ONE tree at once (p2_x16.cpp: one query per AVX-512 lane).  A verifier's time goes into Poseidon2
// (the leaf of the opened row, then one compression per level), and a single opening is one dependency chain -- so the queries of a
// group are hashed in lockstep.  Returns a bit mask: bit j set = opening j FAILED.  Width-24 trees and CPUs without AVX-512 take the
// scalar verify_path per opening.
struct PathBatch { int count; size_t index[16]; const uint32_t* row[16]; const uint32_t* path[16]; };
static void sponge_x16(uint32_t st[16][16], int count, const uint32_t* const* rows, size_t width, size_t& posn, size_t& total) {
    for (size_t i = 0; i < width; i++) {
        for (int j = 0; j < count; j++) st[posn][j] = rows[j][i];
        p2x16_to_monty(st[posn]);
        posn++; total++;
        if (posn == 8) { p2x16_permute(st); posn = 0; }
    }
}
static void compress_x16(uint32_t cur[8][16], uint32_t other[8][16], int count, const size_t* index, int lvl, bool other_is_sibling) {
    // other_is_sibling: a path step (the sibling goes left when bit `lvl` of the index is set); otherwise cur || other (an injected row hash)
    uint32_t st[16][16];
    for (int e = 0; e < 8; e++)
        for (int j = 0; j < 16; j++) {
            const bool right = other_is_sibling && j < count && ((index[j] >> lvl) & 1);
            st[e][j] = right ? other[e][j] : cur[e][j];
            st[8 + e][j] = right ? cur[e][j] : other[e][j];
        }
    p2x16_permute(st);
    memcpy(cur, st, 8 * 16 * 4);
}
static uint32_t verify_paths_x16(const uint32_t* root_m, int log_h, const PathBatch& b, size_t width, int hw) {
    uint32_t failed = 0;
    if (hw != 16 || !p2x16_available()) {
        for (int j = 0; j < b.count; j++) if (!verify_path(root_m, log_h, b.index[j], b.row[j], width, b.path[j], hw)) failed |= 1u << j;
        return failed;
    }
    uint32_t st[16][16] = {};
    size_t posn = 0, total = 0;
    sponge_x16(st, b.count, b.row, width, posn, total);
    if (posn) p2x16_permute(st);
    uint32_t cur[8][16], sib[8][16];
    memcpy(cur, st, sizeof(cur));
    for (int lvl = 0; lvl < log_h; lvl++) {
        for (int e = 0; e < 8; e++) {
            for (int j = 0; j < 16; j++) sib[e][j] = j < b.count ? b.path[j][8 * lvl + e] : 0u;
            p2x16_to_monty(sib[e]);
        }
        compress_x16(cur, sib, b.count, b.index, lvl, true);
    }
    for (int j = 0; j < b.count; j++)
        for (int e = 0; e < 8; e++) if (cur[e][j] != root_m[e]) { failed |= 1u << j; break; }
    return failed;
}
static Ext ext_from_canon(const uint32_t* p) { return Ext{{to_monty(p[0]), to_monty(p[1]), to_monty(p[2]), to_monty(p[3])}}; }
static Ext fri_fold_row(size_t index, int log_folded_h, const Ext& beta, const Ext& e0, const Ext& e1) {
    const uint32_t x = fpow(two_adic_generator(log_folded_h + 1), reverse_bits((uint32_t)index, log_folded_h));
    const uint32_t inv = finv(fneg(fadd(x, x)));
    return ext_add(e0, ext_mul_base(ext_mul(ext_sub_base(beta, x), ext_sub(e1, e0)), inv));
}

// a committed FRI row of 2^K adjacent entries folded K times by 2 with beta, beta^2, ...; `row_index` is the
// row's index in the layer matrix of 2^log_rows rows
static Ext fold_row_k(size_t row_index, int log_rows, int K, Ext beta, const Ext* ev) {
    Ext tmp[32];
    size_t cnt = (size_t)1 << K;
    for (size_t j = 0; j < cnt; j++) tmp[j] = ev[j];
    for (int j = 0; j < K; j++) {
        cnt >>= 1;
        const int log_folded = log_rows + (K - 1 - j);
        for (size_t t = 0; t < cnt; t++) tmp[t] = fri_fold_row(row_index * cnt + t, log_folded, beta, tmp[2 * t], tmp[2 * t + 1]);
        beta = ext_mul(beta, beta);
    }
    return tmp[0];
}

// value at zeta of an extension column committed as 4 base columns: sum_e x^e * v_e(zeta)
static Ext recombine(const Ext* opened4) {
    Ext r = ext_zero();
    for (int e = 0; e < 4; e++) {
        Ext basis = ext_zero();
        basis.c[e] = MONTY_R1;
        r = ext_add(r, ext_mul(basis, opened4[e]));
    }
    return r;
}

// runs check(q) for q in [0, n) on up to 8 host threads; returns the failure code of the LOWEST failing query (0: all passed), so the
// verdict does not depend on the thread count
static int run_queries(int n, const std::function<int(int)>& check, int min_per_thread = 4) {
    unsigned hw = std::thread::hardware_concurrency();
    int threads = (int)(hw ? (hw < 8 ? hw : 8) : 1);
    if (threads > n / min_per_thread) threads = n / min_per_thread;
    if (zk::t_query_threads_cap > 0 && threads > zk::t_query_threads_cap) threads = zk::t_query_threads_cap;      // a caller that is itself one of many workers
    if (threads <= 1) {
        for (int q = 0; q < n; q++) { const int r = check(q); if (r) return r; }
        return 0;
    }
    std::vector<int> result(n, 0);
    std::atomic<int> next{0};
    auto worker = [&]() { for (;;) { const int q = next.fetch_add(1); if (q >= n) return; result[q] = check(q); } };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) { try { pool.emplace_back(worker); } catch (...) { break; } }     // fewer threads, same verdict: this thread works too
    worker();
    for (auto& t : pool) t.join();
    for (int q = 0; q < n; q++) if (result[q]) return result[q];
    return 0;
}

// What the FRI part of a (fold-by-2, constant final value) shard proof consists of, as the verifier meets it: the folding
// challenges, the final value, and per query the index, the reduced opening it starts from and the sibling of every layer.
// zkhip_fri_view_shard points this at its caller's buffers and runs the verifier; the FRI-fold chip (fri_chip.hip) proves
// statements about exactly these values.  Canonical words.
struct FriViewSink {
    uint32_t *betas, *final_value, *indices, *values, *siblings; int layers; uint32_t *roots, *paths; uint32_t* transcript;      // roots / paths / transcript optional
    // skip_paths: the Merkle paths are NOT hashed (every other check runs).  For a caller that recomputes every opening itself and
    // compares the roots -- the shard verifier machine, whose device kernel hashes exactly these paths for its trace (shard_verifier.inl)
    bool skip_paths = false;
};

static int verify_shard_impl(const uint8_t* proof, size_t len, int log_n, uint32_t width, const uint32_t* public_values,
                             size_t n_public, const zkhip_params* prm, int* reason, const AirView* air, FriViewSink* sink = nullptr) {
    int dummy;
    if (!reason) reason = &dummy;
    *reason = 0;
    auto reject = [&](int why) { *reason = why; return fail(ZKHIP_ERR_VERIFY, "proof rejected (check " + std::to_string(why) + ")"); };
    if (check_shape(log_n, width, prm) != ZKHIP_OK) return reject(1);
    if (!proof || (n_public && !public_values)) return reject(1);
    if (air && prm->logup_pairs) return reject(1);
    const int lqd = air ? air->lqd : 1;
    if (lqd > prm->log_blowup) return reject(1);
    const size_t NQ = (size_t)1 << lqd, QW = 4 * NQ;
    if (len != proof_words(log_n, width, prm, air != nullptr, lqd) * 4) return reject(2);
    const uint32_t* pf = (const uint32_t*)proof;
    Shape sh;
    shape_of(log_n, prm, sh);
    const int H = log_n + sh.b, Hq = log_n + lqd, RL = sh.R, K = sh.K;
    const size_t n = (size_t)1 << log_n, arity = (size_t)1 << K;
    const uint32_t LQ = (uint32_t)prm->logup_pairs;
    const size_t wp = LQ ? 4 * ((size_t)LQ + 1) : 0;
    const uint32_t CW = sh.cw;
    if (air && CW) return reject(1);
    if (pf[0] != PROOF_MAGIC || pf[1] != (air ? 7u : (CW ? 8u : (sh.ext ? 3u : (LQ ? 2u : PROOF_VERSION)))) || pf[2] != (uint32_t)log_n || pf[3] != width ||
        pf[4] != (uint32_t)prm->log_blowup || pf[5] != (uint32_t)prm->num_queries || pf[6] != (uint32_t)prm->pow_bits ||
        pf[7] != (uint32_t)n_public) return reject(3);
    size_t pos = 8;
    if (sh.ext || air) {
        if (pf[8] != LQ || pf[9] != (uint32_t)sh.K || pf[10] != (uint32_t)sh.F || pf[11] != (uint32_t)sh.hw) return reject(3);
        pos = 12;
    } else if (LQ) { if (pf[8] != LQ) return reject(3); pos = 9; }
    if (air) {
        uint32_t dg[8];
        air_digest_cached(*air, dg);
        for (int i = 0; i < 8; i++) if (pf[pos + i] != dg[i]) return reject(3);
        pos += 8;
    }
    if (CW) { if (pf[pos] != CW) return reject(3); pos++; }
    for (size_t i = pos; i < len / 4; i++) if (pf[i] >= P) return reject(4);
    for (size_t i = 0; i < n_public; i++) if (public_values[i] >= P) return reject(4);
    Challenger ch;
    transcript_init(ch, log_n, width, prm, n_public, sh, air);
    uint32_t croot[8], troot[8], proot[8], qroot[8];
    if (CW) for (int i = 0; i < 8; i++) { croot[i] = to_monty(pf[pos++]); ch.observe(croot[i]); }
    for (int i = 0; i < 8; i++) { troot[i] = to_monty(pf[pos++]); }
    for (int i = 0; i < 8; i++) ch.observe(troot[i]);
    for (size_t i = 0; i < n_public; i++) ch.observe_canonical(public_values[i]);
    Ext gamma = ext_zero(), beta_l = ext_zero();
    if (LQ) {
        gamma = ch.sample_ext();
        beta_l = ch.sample_ext();
        for (int i = 0; i < 8; i++) { proot[i] = to_monty(pf[pos++]); ch.observe(proot[i]); }
    }
    for (int i = 0; i < 8; i++) { qroot[i] = to_monty(pf[pos++]); }
    const Ext alpha = ch.sample_ext();
    for (int i = 0; i < 8; i++) ch.observe(qroot[i]);
    const Ext zeta = ch.sample_ext();
    const uint32_t gn = two_adic_generator(log_n);
    const Ext zeta_next = ext_mul_base(zeta, gn);
    std::vector<Ext> loc(width), nxt(width), opl(wp), opn(wp);
    Ext opq[16];
    for (size_t j = 0; j < width; j++) loc[j] = ext_from_canon(pf + pos + 4 * j);
    pos += 4 * (size_t)width;
    for (size_t j = 0; j < width; j++) nxt[j] = ext_from_canon(pf + pos + 4 * j);
    pos += 4 * (size_t)width;
    for (size_t j = 0; j < wp; j++) opl[j] = ext_from_canon(pf + pos + 4 * j);
    pos += 4 * wp;
    for (size_t j = 0; j < wp; j++) opn[j] = ext_from_canon(pf + pos + 4 * j);
    pos += 4 * wp;
    for (size_t j = 0; j < QW; j++) opq[j] = ext_from_canon(pf + pos + 4 * j);
    pos += 4 * QW;
    for (size_t j = 0; j < width; j++) ch.observe_ext(loc[j]);
    for (size_t j = 0; j < width; j++) ch.observe_ext(nxt[j]);
    for (size_t j = 0; j < wp; j++) ch.observe_ext(opl[j]);
    for (size_t j = 0; j < wp; j++) ch.observe_ext(opn[j]);
    for (size_t j = 0; j < QW; j++) ch.observe_ext(opq[j]);

    // (a) the AIR identity at zeta: folded constraints / Z_H == sum_k zps_k * q_k
    {
        const Ext zn = ext_pow(zeta, n);
        const Ext zh = ext_sub_base(zn, MONTY_R1);
        const Ext sel_first = ext_mul(zh, ext_inv(ext_sub_base(zeta, MONTY_R1)));
        const Ext sel_trans = ext_sub_base(zeta, finv(gn));
        Ext acc = ext_zero();
        if (air) acc = air_fold_ext(*air, loc.data(), nxt.data(), public_values, sel_first, ext_mul(zh, ext_inv(ext_sub_base(zeta, finv(gn)))), sel_trans, alpha);
        else for (uint32_t g = 0; g < width / 4; g++) {
            const Ext &a = loc[4 * g], &b = loc[4 * g + 1], &c = loc[4 * g + 2], &d = loc[4 * g + 3], &dn = nxt[4 * g + 3];
            const uint32_t k1 = to_monty(g + 1), k2 = to_monty(2 * g + 3), d0 = to_monty(5 * g + 7);
            const Ext c1 = ext_sub_base(ext_sub(c, ext_mul(ext_mul(a, a), b)), k1);
            const Ext c2 = ext_mul(sel_trans, ext_sub_base(ext_sub(ext_sub(dn, ext_mul(a, b)), c), k2));
            const Ext c3 = ext_mul(sel_first, ext_sub_base(d, d0));
            acc = ext_add(ext_mul(acc, alpha), c1);
            acc = ext_add(ext_mul(acc, alpha), c2);
            acc = ext_add(ext_mul(acc, alpha), c3);
        }
        if (LQ) {
            // LogUp: L_q, then T1 (first row), T2 (transition), T3 (last row)
            const Ext sel_last = ext_mul(zh, ext_inv(ext_sub_base(zeta, finv(gn))));
            Ext sum_l = ext_zero(), sum_n = ext_zero();
            for (uint32_t q = 0; q < LQ; q++) {
                const Ext ds = ext_add(ext_add(gamma, loc[8 * q]), ext_mul(beta_l, loc[8 * q + 1]));
                const Ext dr = ext_add(ext_add(gamma, loc[8 * q + 4]), ext_mul(beta_l, loc[8 * q + 5]));
                const Ext phi = recombine(&opl[4 * q]), phin = recombine(&opn[4 * q]);
                const Ext c = ext_sub(ext_mul(ext_mul(phi, ds), dr), ext_sub(dr, ds));
                acc = ext_add(ext_mul(acc, alpha), c);
                sum_l = ext_add(sum_l, phi);
                sum_n = ext_add(sum_n, phin);
            }
            const Ext S = recombine(&opl[4 * LQ]), Sn = recombine(&opn[4 * LQ]);
            acc = ext_add(ext_mul(acc, alpha), ext_mul(sel_first, ext_sub(S, sum_l)));
            acc = ext_add(ext_mul(acc, alpha), ext_mul(sel_trans, ext_sub(ext_sub(Sn, S), sum_n)));
            acc = ext_add(ext_mul(acc, alpha), ext_mul(sel_last, S));
        }
        // quotient(zeta) = sum_k zps_k(zeta) q_k(zeta): chunk k lives on the coset s_k <w_N>, s_k = g w_{2^Hq}^k, and
        // zps_k = prod_{j != k} Z_Dj(zeta) / Z_Dj(s_k), Z_Dj(x) = (x / s_j)^N - 1, vanishes on every other chunk's coset
        const uint32_t wq = two_adic_generator(Hq);
        uint32_t sN[4];
        for (size_t k = 0; k < NQ; k++) sN[k] = fpow(fmul(MONTY_GEN, fpow(wq, (uint64_t)k)), n);
        Ext quot = ext_zero();
        for (size_t k = 0; k < NQ; k++) {
            Ext zps = ext_one();
            for (size_t j = 0; j < NQ; j++) {
                if (j == k) continue;
                const uint32_t sjn_inv = finv(sN[j]);
                const Ext num = ext_sub_base(ext_mul_base(zn, sjn_inv), MONTY_R1);
                const uint32_t den = fsub(fmul(sN[k], sjn_inv), MONTY_R1);
                zps = ext_mul(zps, ext_mul_base(num, finv(den)));
            }
            quot = ext_add(quot, ext_mul(zps, recombine(&opq[4 * k])));
        }
        if (!ext_eq(ext_mul(acc, ext_inv(zh)), quot)) return reject(10);
    }

    // (b) FRI
    const Ext fa = ch.sample_ext();
    size_t np = width > QW ? width : QW;
    if (wp > np) np = wp;
    std::vector<Ext> fapow(np);
    fapow[0] = ext_one();
    for (size_t j = 1; j < np; j++) fapow[j] = ext_mul(fapow[j - 1], fa);
    Ext y_loc = ext_zero(), y_nxt = ext_zero(), y_pl = ext_zero(), y_pn = ext_zero(), y_q = ext_zero();
    for (size_t j = 0; j < width; j++) {
        y_loc = ext_add(y_loc, ext_mul(fapow[j], loc[j]));
        y_nxt = ext_add(y_nxt, ext_mul(fapow[j], nxt[j]));
    }
    for (size_t j = 0; j < wp; j++) {
        y_pl = ext_add(y_pl, ext_mul(fapow[j], opl[j]));
        y_pn = ext_add(y_pn, ext_mul(fapow[j], opn[j]));
    }
    for (size_t j = 0; j < QW; j++) y_q = ext_add(y_q, ext_mul(fapow[j], opq[j]));
    const Ext off_next = ext_pow(fa, width), off_pl = ext_pow(fa, 2 * (uint64_t)width),
              off_pn = ext_pow(fa, 2 * (uint64_t)width + wp), off_q = ext_pow(fa, 2 * (uint64_t)width + 2 * wp);
    std::vector<uint32_t> commits((size_t)RL * 8 + 8);
    std::vector<Ext> betas(RL + 1);
    if (sink && sink->transcript) {               // the challenger as the commit phase finds it: capacity half of the state, pending inputs
        for (int i = 0; i < 8; i++) sink->transcript[i] = from_monty(ch.state[8 + i]);
        sink->transcript[8] = (uint32_t)ch.n_in;
    }
    for (int l = 0; l < RL; l++) {
        for (int i = 0; i < 8; i++) { commits[8 * l + i] = to_monty(pf[pos++]); ch.observe(commits[8 * l + i]); }
        betas[l] = ch.sample_ext();
    }
    const size_t keep = (size_t)1 << sh.F;
    std::vector<Ext> final_poly(keep);            // coefficients, lowest first
    for (size_t i = 0; i < keep; i++) { final_poly[i] = ext_from_canon(pf + pos); pos += 4; ch.observe_ext(final_poly[i]); }
    const uint32_t witness = pf[pos++];
    if (sink && sink->transcript) sink->transcript[9] = witness;
    ch.observe_canonical(witness);
    if (ch.sample_bits(prm->pow_bits) != 0) return reject(20);
    const uint32_t wm = two_adic_generator(H);
    // The query indices come out of the transcript one after the other; the checks of a query read only its own slice of the proof
    // (every query has the same length), so they run on a few host threads -- a verifier spends its time in the ~200 Poseidon2
    // permutations per query (leaf of the trace row, Merkle paths, FRI layers).
    const int NQ_ = prm->num_queries;
    std::vector<size_t> indices(NQ_);
    for (int q = 0; q < NQ_; q++) indices[q] = ch.sample_bits(H);
    if (sink) {
        if (K != 1 || sh.F != 0 || RL != sink->layers) return reject(1);
        for (int l = 0; l < RL; l++) for (int i = 0; i < 4; i++) sink->betas[4 * l + i] = from_monty(betas[l].c[i]);
        for (int i = 0; i < 4; i++) sink->final_value[i] = from_monty(final_poly[0].c[i]);
        for (int q = 0; q < NQ_; q++) sink->indices[q] = (uint32_t)indices[q];
        if (sink->roots) for (int l = 0; l < RL; l++) for (int i = 0; i < 8; i++) sink->roots[8 * l + i] = from_monty(commits[8 * l + i]);
    }
    const size_t pos0 = pos, words_total = len / 4;
    if ((words_total - pos0) % (size_t)NQ_ != 0) return reject(5);
    const size_t perq = (words_total - pos0) / (size_t)NQ_;
    // queries go in groups of 16: the Merkle openings of a group are hashed in lockstep (verify_paths_x16), the field arithmetic in
    // between stays per query.  code[j] = the first check query j fails, in the order a query-by-query verifier meets them.
    const int NG = (NQ_ + 15) / 16;
    std::vector<int> qcode(NQ_, 0);
    auto check_group = [&](int g) -> int {
        const int q0 = 16 * g, cnt = NQ_ - q0 < 16 ? NQ_ - q0 : 16;
        const uint32_t *trow[16], *cpath[16], *tpath[16], *prow[16], *ppath[16], *qrow[16], *qpath[16];
        size_t qpos[16], index[16];
        int code[16] = {0};
        auto mark = [&](uint32_t mask, int why) { for (int j = 0; j < cnt; j++) if (((mask >> j) & 1u) && !code[j]) code[j] = why; };
        for (int j = 0; j < cnt; j++) {
            size_t pos = pos0 + (size_t)(q0 + j) * perq;
            index[j] = indices[q0 + j];
            trow[j] = pf + pos; pos += width;
            cpath[j] = nullptr;
            if (CW) { cpath[j] = pf + pos; pos += 8 * (size_t)H; }
            tpath[j] = pf + pos; pos += 8 * (size_t)H;
            prow[j] = ppath[j] = nullptr;
            if (LQ) { prow[j] = pf + pos; pos += wp; ppath[j] = pf + pos; pos += 8 * (size_t)H; }
            qrow[j] = pf + pos; pos += QW;
            qpath[j] = pf + pos; pos += 8 * (size_t)H;
            qpos[j] = pos;
        }
        auto batch = [&](const uint32_t* const* rows, size_t row_off, const uint32_t* const* paths, const size_t* idx) {
            PathBatch b;
            b.count = cnt;
            for (int j = 0; j < cnt; j++) { b.index[j] = idx[j]; b.row[j] = rows[j] + row_off; b.path[j] = paths[j]; }
            return b;
        };
        const bool hash_paths = !(sink && sink->skip_paths);
        if (hash_paths) {
            if (CW) mark(verify_paths_x16(croot, H, batch(trow, 0, cpath, index), CW, sh.hw), 33);
            mark(verify_paths_x16(troot, H, batch(trow, CW, tpath, index), width - CW, sh.hw), 30);
            if (LQ) mark(verify_paths_x16(proot, H, batch(prow, 0, ppath, index), wp, sh.hw), 32);
            mark(verify_paths_x16(qroot, H, batch(qrow, 0, qpath, index), QW, sh.hw), 31);
        }
        Ext folded[16];
        size_t idx[16];
        for (int j = 0; j < cnt; j++) {
            const uint32_t x = fmul(MONTY_GEN, fpow(wm, reverse_bits((uint32_t)index[j], H)));
            const Ext d1 = ext_inv(ext_neg(ext_sub_base(zeta, x)));
            const Ext d2 = ext_inv(ext_neg(ext_sub_base(zeta_next, x)));
            Ext at = ext_zero(), ap = ext_zero(), aq = ext_zero();
            for (size_t k = 0; k < width; k++) at = ext_add(at, ext_mul_base(fapow[k], to_monty(trow[j][k])));
            for (size_t k = 0; k < wp; k++) ap = ext_add(ap, ext_mul_base(fapow[k], to_monty(prow[j][k])));
            for (size_t k = 0; k < QW; k++) aq = ext_add(aq, ext_mul_base(fapow[k], to_monty(qrow[j][k])));
            Ext f = ext_mul(ext_sub(at, y_loc), d1);
            f = ext_add(f, ext_mul(off_next, ext_mul(ext_sub(at, y_nxt), d2)));
            if (LQ) {
                f = ext_add(f, ext_mul(off_pl, ext_mul(ext_sub(ap, y_pl), d1)));
                f = ext_add(f, ext_mul(off_pn, ext_mul(ext_sub(ap, y_pn), d2)));
            }
            folded[j] = ext_add(f, ext_mul(off_q, ext_mul(ext_sub(aq, y_q), d1)));
            idx[j] = index[j];
            if (sink) for (int i = 0; i < 4; i++) sink->values[4 * (size_t)(q0 + j) + i] = from_monty(folded[j].c[i]);
        }
        std::vector<uint32_t> rowbuf((size_t)16 * 4 * arity);
        std::vector<Ext> ev((size_t)16 * arity);
        for (int l = 0; l < RL; l++) {
            const int lh = H - K * (l + 1);
            const uint32_t *rows[16], *paths[16];
            size_t rowidx[16];
            for (int j = 0; j < cnt; j++) {
                const size_t row = idx[j] >> K, own = idx[j] & (arity - 1);
                uint32_t* rb = rowbuf.data() + (size_t)j * 4 * arity;
                Ext* e = ev.data() + (size_t)j * arity;
                for (size_t k = 0; k < arity; k++) {
                    if (k == own) { e[k] = folded[j]; for (int i = 0; i < 4; i++) rb[4 * k + i] = from_monty(folded[j].c[i]); }
                    else {
                        e[k] = ext_from_canon(pf + qpos[j]);
                        for (int i = 0; i < 4; i++) rb[4 * k + i] = pf[qpos[j] + i];
                        if (sink) for (int i = 0; i < 4; i++) sink->siblings[4 * ((size_t)(q0 + j) * RL + l) + i] = pf[qpos[j] + i];
                        qpos[j] += 4;
                    }
                }
                rows[j] = rb; paths[j] = pf + qpos[j]; rowidx[j] = row;
                if (sink && sink->paths) {       // per query: the layers' paths one after the other, 8 (RL - l) words for layer l (fold by 2, H = RL + 1)
                    const size_t per_query = 4 * (size_t)RL * ((size_t)RL + 1), before = 8 * ((size_t)l * RL - (size_t)l * ((size_t)l - 1) / 2);
                    std::memcpy(sink->paths + (size_t)(q0 + j) * per_query + before, pf + qpos[j], 32 * (size_t)lh);
                }
                qpos[j] += 8 * (size_t)lh;
            }
            if (hash_paths) mark(verify_paths_x16(&commits[8 * l], lh, batch(rows, 0, paths, rowidx), 4 * arity, sh.hw), 40 + (l < 50 ? l : 50));
            for (int j = 0; j < cnt; j++) { folded[j] = fold_row_k(rowidx[j], lh, K, betas[l], ev.data() + (size_t)j * arity); idx[j] = rowidx[j]; }
        }
        // the final polynomial at every query's point of the last domain <w_{2^(F+b)}> (Horner)
        for (int j = 0; j < cnt; j++) {
            const int lf = sh.F + sh.b;
            const uint32_t xf = fpow(two_adic_generator(lf), reverse_bits((uint32_t)idx[j], lf));
            Ext v = ext_zero();
            for (size_t i = keep; i-- > 0;) v = ext_add(ext_mul_base(v, xf), final_poly[i]);
            if (!ext_eq(folded[j], v) && !code[j]) code[j] = 100;
            if (qpos[j] != pos0 + (size_t)(q0 + j + 1) * perq && !code[j]) code[j] = 5;
            qcode[q0 + j] = code[j];
        }
        return 0;
    };
    run_queries(NG, check_group, 1);
    for (int q = 0; q < NQ_; q++) if (qcode[q]) return reject(qcode[q]);
    pos = pos0 + (size_t)NQ_ * perq;
    if (pos * 4 != len) return reject(5);
    return ZKHIP_OK;
}

int zkhip_verify_shard(const uint8_t* proof, size_t len, int log_n, uint32_t width, const uint32_t* public_values,
                       size_t n_public, const zkhip_params* prm, int* reason) {
    return verify_shard_impl(proof, len, log_n, width, public_values, n_public, prm, reason, nullptr);
}
int zkhip_verify_shard_air(const uint32_t* program, size_t program_words, const uint8_t* proof, size_t len, int log_n, uint32_t width,
                           const uint32_t* public_values, size_t n_public, const zkhip_params* prm, int* reason) {
    AirView a;
    if (!air_validate(program, program_words, width, n_public, &a)) {
        if (reason) *reason = 1;
        return fail(ZKHIP_ERR_VERIFY, "verify_shard_air: malformed constraint program");
    }
    return verify_shard_impl(proof, len, log_n, width, public_values, n_public, prm, reason, &a);
}

int zkhip_fri_view_shard(const uint8_t* proof, size_t len, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public,
                         const zkhip_params* prm, uint32_t* betas, uint32_t final_value[4], uint32_t* indices, uint32_t* values, uint32_t* siblings) {
    if (!prm || !betas || !final_value || !indices || !values || !siblings) return fail(ZKHIP_ERR_INVALID, "fri_view_shard: null argument");
    Shape sh;
    if (check_shape(log_n, width, prm) != ZKHIP_OK) return ZKHIP_ERR_INVALID;
    shape_of(log_n, prm, sh);
    if (sh.K != 1 || sh.F != 0) return fail(ZKHIP_ERR_INVALID, "fri_view_shard: fold-by-2 proofs with a constant final value only");
    FriViewSink sink{betas, final_value, indices, values, siblings, sh.R, nullptr, nullptr, nullptr};
    int why = 0;
    return verify_shard_impl(proof, len, log_n, width, public_values, n_public, prm, &why, nullptr, &sink);
}
size_t zkhip_fri_view_path_words(int layers) { return layers >= 1 && layers <= 22 ? 4 * (size_t)layers * ((size_t)layers + 1) : 0; }
int zkhip_fri_view_shard_paths(const uint8_t* proof, size_t len, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public,
                               const zkhip_params* prm, uint32_t* betas, uint32_t final_value[4], uint32_t* indices, uint32_t* values, uint32_t* siblings,
                               uint32_t* roots, uint32_t* paths) {
    if (!prm || !betas || !final_value || !indices || !values || !siblings || !roots || !paths) return fail(ZKHIP_ERR_INVALID, "fri_view_shard_paths: null argument");
    Shape sh;
    if (check_shape(log_n, width, prm) != ZKHIP_OK) return ZKHIP_ERR_INVALID;
    shape_of(log_n, prm, sh);
    if (sh.K != 1 || sh.F != 0 || sh.b != 1) return fail(ZKHIP_ERR_INVALID, "fri_view_shard_paths: fold-by-2, blowup-2 proofs with a constant final value only");
    FriViewSink sink{betas, final_value, indices, values, siblings, sh.R, roots, paths, nullptr};
    int why = 0;
    return verify_shard_impl(proof, len, log_n, width, public_values, n_public, prm, &why, nullptr, &sink);
}
// The Fiat-Shamir side of the same view: the layer roots, the challenges, and the duplex challenger's state as the commit phase finds
// it -- transcript[0..8) = the capacity half of its state (canonical), transcript[8] = inputs pending (0 for every proof shape of this
// library: the step before is a sample), transcript[9] = the proof-of-work witness (absorbed behind the final value by the query phase).  With these the challenges are a SPONGE CHAIN over the roots: state <- (root_l | capacity),
// permute, beta_l = (state[7], state[6], state[5], state[4]), capacity <- state[8..16) -- the rows a Poseidon2 chip in sponge mode
// already has; what a transcript chip has to prove (docs/RECURSION_NEXT.md).
int zkhip_fri_view_transcript(const uint8_t* proof, size_t len, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public,
                              const zkhip_params* prm, uint32_t* roots, uint32_t* betas, uint32_t transcript[10]) {
    if (!prm || !roots || !betas || !transcript) return fail(ZKHIP_ERR_INVALID, "fri_view_transcript: null argument");
    Shape sh;
    if (check_shape(log_n, width, prm) != ZKHIP_OK) return ZKHIP_ERR_INVALID;
    shape_of(log_n, prm, sh);
    if (sh.K != 1 || sh.F != 0) return fail(ZKHIP_ERR_INVALID, "fri_view_transcript: fold-by-2 proofs with a constant final value only");
    std::vector<uint32_t> final_value(4), indices((size_t)prm->num_queries), values((size_t)prm->num_queries * 4),
        siblings((size_t)prm->num_queries * (size_t)sh.R * 4);
    FriViewSink sink{betas, final_value.data(), indices.data(), values.data(), siblings.data(), sh.R, roots, nullptr, transcript};
    int why = 0;
    return verify_shard_impl(proof, len, log_n, width, public_values, n_public, prm, &why, nullptr, &sink);
}

// the same view WITHOUT hashing the Merkle paths (library-internal: the caller recomputes every opening and compares the roots)
extern "C++" {
namespace zk {
thread_local int t_query_threads_cap = 0;
int fri_view_all_unhashed(const uint8_t* proof, size_t len, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public,
                          const zkhip_params* prm, uint32_t* betas, uint32_t final_value[4], uint32_t* indices, uint32_t* values, uint32_t* siblings,
                          uint32_t* roots, uint32_t* paths, uint32_t transcript[10], const uint32_t* program, size_t program_words) {
    Shape sh;
    if (!prm || check_shape(log_n, width, prm) != ZKHIP_OK) return ZKHIP_ERR_INVALID;
    shape_of(log_n, prm, sh);
    if (sh.K != 1 || sh.F != 0 || sh.b != 1) return fail(ZKHIP_ERR_INVALID, "fri_view_all: fold-by-2, blowup-2 proofs with a constant final value only");
    FriViewSink sink{betas, final_value, indices, values, siblings, sh.R, roots, paths, transcript};
    sink.skip_paths = true;
    int why = 0;
    AirView av;
    if (program && !air_validate(program, program_words, width, n_public, &av)) return fail(ZKHIP_ERR_INVALID, "fri_view_all: malformed constraint program");
    return verify_shard_impl(proof, len, log_n, width, public_values, n_public, prm, &why, program ? &av : nullptr, &sink);
}
}  // namespace zk
}  // extern "C++"

// everything the recursion machines read, from ONE pass over the proof: the view with roots and paths, and the challenger's side
int zkhip_fri_view_all(const uint8_t* proof, size_t len, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public,
                       const zkhip_params* prm, uint32_t* betas, uint32_t final_value[4], uint32_t* indices, uint32_t* values, uint32_t* siblings,
                       uint32_t* roots, uint32_t* paths, uint32_t transcript[10]) {
    if (!prm || !betas || !final_value || !indices || !values || !siblings || !roots || !paths || !transcript) return fail(ZKHIP_ERR_INVALID, "fri_view_all: null argument");
    Shape sh;
    if (check_shape(log_n, width, prm) != ZKHIP_OK) return ZKHIP_ERR_INVALID;
    shape_of(log_n, prm, sh);
    if (sh.K != 1 || sh.F != 0 || sh.b != 1) return fail(ZKHIP_ERR_INVALID, "fri_view_all: fold-by-2, blowup-2 proofs with a constant final value only");
    FriViewSink sink{betas, final_value, indices, values, siblings, sh.R, roots, paths, transcript};
    int why = 0;
    return verify_shard_impl(proof, len, log_n, width, public_values, n_public, prm, &why, nullptr, &sink);
}

// opening of a mixed-height tree (host): rows[c] = chip c's row at index >> (Hmax - lh[c]), canonical words
static bool verify_mixed(const uint32_t* root_m, int Hmax, size_t index, const uint32_t* const* rows, const uint32_t* widths,
                         const int* lh, int n, const uint32_t* sibs_canon) {
    auto hash_height = [&](int h, uint32_t out[8]) -> bool {
        uint32_t s[16] = {0};
        size_t posn = 0, total = 0;
        for (int c = 0; c < n; c++)
            if (lh[c] == h)
                for (uint32_t i = 0; i < widths[c]; i++) {
                    s[posn++] = to_monty(rows[c][i]); total++;
                    if (posn == 8) { p2_permute(s); posn = 0; }
                }
        if (!total) return false;
        if (posn) p2_permute(s);
        for (int i = 0; i < 8; i++) out[i] = s[i];
        return true;
    };
    uint32_t cur[8], rh[8];
    hash_height(Hmax, cur);
    for (int lvl = 0; lvl < Hmax; lvl++) {
        uint32_t sib[8];
        for (int i = 0; i < 8; i++) sib[i] = to_monty(sibs_canon[8 * lvl + i]);
        if ((index >> lvl) & 1) p2_compress(sib, cur, cur);
        else p2_compress(cur, sib, cur);
        if (hash_height(Hmax - lvl - 1, rh)) p2_compress(cur, rh, cur);
    }
    for (int i = 0; i < 8; i++) if (cur[i] != root_m[i]) return false;
    return true;
}

// sixteen openings of one mixed-height tree at once (as verify_paths_x16): rows[j][c] = the row of chip c in query j
static uint32_t verify_mixed_x16(const uint32_t* root_m, int Hmax, int count, const size_t* index, const uint32_t* const (*rows)[32],
                                 const uint32_t* widths, const int* lh, int n, const uint32_t* const* paths) {
    uint32_t failed = 0;
    if (!p2x16_available()) {
        for (int j = 0; j < count; j++) if (!verify_mixed(root_m, Hmax, index[j], rows[j], widths, lh, n, paths[j])) failed |= 1u << j;
        return failed;
    }
    auto hash_height = [&](int h, uint32_t out[8][16]) -> bool {
        uint32_t st[16][16] = {};
        size_t posn = 0, total = 0;
        for (int c = 0; c < n; c++)
            if (lh[c] == h) {
                const uint32_t* r[16];
                for (int j = 0; j < count; j++) r[j] = rows[j][c];
                sponge_x16(st, count, r, widths[c], posn, total);
            }
        if (!total) return false;
        if (posn) p2x16_permute(st);
        memcpy(out, st, 8 * 16 * 4);
        return true;
    };
    uint32_t cur[8][16], other[8][16];
    hash_height(Hmax, cur);
    for (int lvl = 0; lvl < Hmax; lvl++) {
        for (int e = 0; e < 8; e++) {
            for (int j = 0; j < 16; j++) other[e][j] = j < count ? paths[j][8 * lvl + e] : 0u;
            p2x16_to_monty(other[e]);
        }
        compress_x16(cur, other, count, index, lvl, true);
        if (hash_height(Hmax - lvl - 1, other)) compress_x16(cur, other, count, index, lvl, false);
    }
    for (int j = 0; j < count; j++)
        for (int e = 0; e < 8; e++) if (cur[e][j] != root_m[e]) { failed |= 1u << j; break; }
    return failed;
}

static int verify_chips_impl(const ChipSet& cs, const uint8_t* proof, size_t len, const int32_t* log_ns, const uint32_t* widths, const int32_t* pairs, const int32_t* partners, int n,
                             const uint32_t* public_values, size_t n_public, const zkhip_params* prm, int* reason) {
    int dummy;
    if (!reason) reason = &dummy;
    *reason = 0;
    auto reject = [&](int why) { *reason = why; return fail(ZKHIP_ERR_VERIFY, "proof rejected (check " + std::to_string(why) + ")"); };
    if (check_chips(cs, log_ns, widths, pairs, partners, n, prm) != ZKHIP_OK) return reject(1);
    if (!proof || (n_public && !public_values)) return reject(1);
    if (len != chips_proof_words(cs, log_ns, widths, pairs, partners, n, prm) * 4) return reject(2);
    const uint32_t* pf = (const uint32_t*)proof;
    const bool lk = any_pairs(pairs, n), cross = any_cross(cs, partners, n);
    const int b = prm->log_blowup, Hmax = log_ns[0] + b, L = log_ns[0];
    if (pf[0] != PROOF_MAGIC || pf[1] != chips_version(cs, pairs, partners, n) || pf[2] != (uint32_t)n || pf[3] != (uint32_t)b ||
        pf[4] != (uint32_t)prm->num_queries || pf[5] != (uint32_t)prm->pow_bits || pf[6] != (uint32_t)n_public || pf[7] != 16u) return reject(3);
    size_t pos = 8;
    for (int c = 0; c < n; c++) {
        if (pf[pos] != (uint32_t)log_ns[c] || pf[pos + 1] != widths[c]) return reject(3);
        pos += 2;
        if (cs.machine) {
            if (pf[pos] != header_prog_word(cs, c) || pf[pos + 1] != (lookup_of(cs, c) ? lookup_of(cs, c)->ni : 0u)) return reject(3);
            pos += 2;
            if (cs.key) { if (pf[pos] != pre_w(cs, c)) return reject(3); pos++; }
            continue;
        }
        if (lk) { if (pf[pos] != (uint32_t)pairs[c]) return reject(3); pos++; }
        if (cross) { if (pf[pos] != (uint32_t)(partners[c] + 1)) return reject(3); pos++; }
        if (any_prog(cs, n)) { if (pf[pos] != header_prog_word(cs, c)) return reject(3); pos++; }
    }
    for (int c = 0; c < n; c++)
        if (header_has_prog(cs, c)) {
            uint32_t dg[8];
            air_digest_cached(*prog_of(cs, c), dg);
            for (int i = 0; i < 8; i++) if (pf[pos + i] != dg[i]) return reject(3);
            pos += 8;
        }
    for (int c = 0; c < n; c++)
        if (lookup_of(cs, c)) {
            uint32_t dg[8];
            lookup_digest(*lookup_of(cs, c), dg);
            for (int i = 0; i < 8; i++) if (pf[pos + i] != dg[i]) return reject(3);
            pos += 8;
        }
    if (cs.key) { for (int i = 0; i < 8; i++) if (pf[pos + i] != from_monty(cs.key->root_m[i])) return reject(3); pos += 8; }     // a proof under another key
    for (size_t i = pos; i < len / 4; i++) if (pf[i] >= P) return reject(4);
    for (size_t i = 0; i < n_public; i++) if (public_values[i] >= P) return reject(4);
    int lh[MAX_CHIPS]; uint32_t w8[MAX_CHIPS]; size_t wp[MAX_CHIPS];
    for (int c = 0; c < n; c++) { lh[c] = log_ns[c] + b; w8[c] = (uint32_t)qw_of(cs, c); wp[c] = perm_width(pairs, c); }
    Challenger ch;
    chips_transcript_init(cs, ch, log_ns, widths, pairs, partners, n, prm, n_public);
    Ext cumsum[MAX_CHIPS];
    for (int c = 0; c < n; c++) cumsum[c] = ext_zero();
    uint32_t troot[8], proot[8] = {0}, qroot[8];
    for (int i = 0; i < 8; i++) { troot[i] = to_monty(pf[pos++]); ch.observe(troot[i]); }
    for (size_t i = 0; i < n_public; i++) ch.observe_canonical(public_values[i]);
    Ext gamma = ext_zero(), beta_l = ext_zero();
    uint32_t pw[MAX_CHIPS]; int plh[MAX_CHIPS], pchip[MAX_CHIPS]; int np = 0, Hp = 0;
    if (lk) {
        gamma = ch.sample_ext();
        beta_l = ch.sample_ext();
        for (int i = 0; i < 8; i++) { proot[i] = to_monty(pf[pos++]); ch.observe(proot[i]); }
        for (int c = 0; c < n; c++) if (wp[c]) { pw[np] = (uint32_t)wp[c]; plh[np] = lh[c]; pchip[np] = c; np++; if (lh[c] > Hp) Hp = lh[c]; }
        if (cross) {
            Ext total = ext_zero();
            for (int c = 0; c < n; c++)
                if (wp[c]) { cumsum[c] = ext_from_canon(pf + pos); pos += 4; ch.observe_ext(cumsum[c]); total = ext_add(total, cumsum[c]); }
            if (!ext_eq(total, ext_zero())) return reject(11);          // the lookups of the shard do not balance
        }
    }
    const Ext alpha = ch.sample_ext();
    for (int i = 0; i < 8; i++) { qroot[i] = to_monty(pf[pos++]); ch.observe(qroot[i]); }
    const Ext zeta = ch.sample_ext();
    std::vector<std::vector<Ext>> loc(n), nxt(n), opl(n), opn(n), opq(n), oel(n), oen(n);
    for (int c = 0; c < n; c++) {
        const uint32_t W = widths[c];
        auto take = [&](std::vector<Ext>& v, size_t cnt) { v.resize(cnt); for (size_t j = 0; j < cnt; j++) v[j] = ext_from_canon(pf + pos + 4 * j); pos += 4 * cnt; };
        take(oel[c], pre_w(cs, c)); take(oen[c], pre_w(cs, c));
        take(loc[c], W); take(nxt[c], W); take(opl[c], wp[c]); take(opn[c], wp[c]); take(opq[c], w8[c]);
    }
    // (the query groups run on worker threads, which do not see this thread's key: widths and root by value from here on)
    uint32_t ew[MAX_CHIPS], pwv[MAX_CHIPS]; int elh[MAX_CHIPS], echip[MAX_CHIPS]; int ne = 0, He = 0;
    const uint32_t* const eroot = cs.key ? cs.key->root_m : nullptr;
    for (int c = 0; c < n; c++) pwv[c] = pre_w(cs, c);
    for (int c = 0; c < n; c++) if (pre_w(cs, c)) { ew[ne] = pre_w(cs, c); elh[ne] = lh[c]; echip[ne] = c; ne++; if (lh[c] > He) He = lh[c]; }
    for (int c = 0; c < n; c++) {
        for (const Ext& e : oel[c]) ch.observe_ext(e);
        for (const Ext& e : oen[c]) ch.observe_ext(e);
        for (const Ext& e : loc[c]) ch.observe_ext(e);
        for (const Ext& e : nxt[c]) ch.observe_ext(e);
        for (const Ext& e : opl[c]) ch.observe_ext(e);
        for (const Ext& e : opn[c]) ch.observe_ext(e);
        for (const Ext& e : opq[c]) ch.observe_ext(e);
    }
    // (a) every chip's AIR identity at zeta
    for (int c = 0; c < n; c++) {
        const size_t nc = (size_t)1 << log_ns[c];
        const uint32_t gn = two_adic_generator(log_ns[c]);
        const Ext zn = ext_pow(zeta, nc), zh = ext_sub_base(zn, MONTY_R1);
        const Ext sel_first = ext_mul(zh, ext_inv(ext_sub_base(zeta, MONTY_R1)));
        const Ext sel_trans = ext_sub_base(zeta, finv(gn));
        Ext acc = ext_zero();
        // what the chip's program and interactions read: the combined row [preprocessed | main] at zeta, and at zeta g
        std::vector<Ext> cl_, cn_;
        const Ext *row_l = loc[c].data(), *row_n = nxt[c].data();
        if (pre_w(cs, c)) {
            cl_ = oel[c]; cl_.insert(cl_.end(), loc[c].begin(), loc[c].end());
            cn_ = oen[c]; cn_.insert(cn_.end(), nxt[c].begin(), nxt[c].end());
            row_l = cl_.data(); row_n = cn_.data();
        }
        if (prog_of(cs, c))
            acc = air_fold_ext(*prog_of(cs, c), row_l, row_n, public_values, sel_first, ext_mul(zh, ext_inv(ext_sub_base(zeta, finv(gn)))), sel_trans, alpha);
        else for (uint32_t g = 0; g < widths[c] / 4; g++) {
            const Ext &a = loc[c][4 * g], &bb = loc[c][4 * g + 1], &cc = loc[c][4 * g + 2], &d = loc[c][4 * g + 3], &dn = nxt[c][4 * g + 3];
            const uint32_t k1 = to_monty(g + 1), k2 = to_monty(2 * g + 3), d0 = to_monty(5 * g + 7);
            const Ext c1 = ext_sub_base(ext_sub(cc, ext_mul(ext_mul(a, a), bb)), k1);
            const Ext c2 = ext_mul(sel_trans, ext_sub_base(ext_sub(ext_sub(dn, ext_mul(a, bb)), cc), k2));
            const Ext c3 = ext_mul(sel_first, ext_sub_base(d, d0));
            acc = ext_add(ext_mul(acc, alpha), c1);
            acc = ext_add(ext_mul(acc, alpha), c2);
            acc = ext_add(ext_mul(acc, alpha), c3);
        }
        if (wp[c] && cs.machine) {
            const LookupView& lv = *lookup_of(cs, c);
            const Ext sel_last = ext_mul(zh, ext_inv(ext_sub_base(zeta, finv(gn))));
            std::vector<Ext> pl(lv.cols + 1), pn(lv.cols + 1);
            for (uint32_t q = 0; q <= lv.cols; q++) { pl[q] = recombine(&opl[c][4 * q]); pn[q] = recombine(&opn[c][4 * q]); }
            acc = lookup_fold_ext(acc, lv, row_l, pl.data(), pn.data(), gamma, beta_l, sel_first, sel_trans, sel_last, alpha, cumsum[c]);
        } else if (wp[c]) {
            const uint32_t LQ = (uint32_t)pairs[c];
            const Ext sel_last = ext_mul(zh, ext_inv(ext_sub_base(zeta, finv(gn))));
            Ext sum_l = ext_zero(), sum_n = ext_zero();
            for (uint32_t q = 0; q < LQ; q++) {
                const Ext ds = ext_add(ext_add(gamma, loc[c][8 * q]), ext_mul(beta_l, loc[c][8 * q + 1]));
                const Ext dr = ext_add(ext_add(gamma, loc[c][8 * q + 4]), ext_mul(beta_l, loc[c][8 * q + 5]));
                const Ext phi = recombine(&opl[c][4 * q]), phin = recombine(&opn[c][4 * q]);
                acc = ext_add(ext_mul(acc, alpha), ext_sub(ext_mul(ext_mul(phi, ds), dr), ext_sub(dr, ds)));
                sum_l = ext_add(sum_l, phi);
                sum_n = ext_add(sum_n, phin);
            }
            const Ext S = recombine(&opl[c][4 * LQ]), Sn = recombine(&opn[c][4 * LQ]);
            acc = ext_add(ext_mul(acc, alpha), ext_mul(sel_first, ext_sub(S, sum_l)));
            acc = ext_add(ext_mul(acc, alpha), ext_mul(sel_trans, ext_sub(ext_sub(Sn, S), sum_n)));
            acc = ext_add(ext_mul(acc, alpha), ext_mul(sel_last, ext_sub(S, cumsum[c])));
        }
        // quotient(zeta) = sum_k zps_k(zeta) q_k(zeta) over the chip's own 2^lq cosets s_k <w_N>, s_k = g w^k (as in the single-matrix verifier)
        const size_t NQc = w8[c] / 4;
        const uint32_t wq = two_adic_generator(log_ns[c] + (NQc == 4 ? 2 : 1));
        uint32_t sN[4];
        for (size_t k = 0; k < NQc; k++) sN[k] = fpow(fmul(MONTY_GEN, fpow(wq, (uint64_t)k)), nc);
        Ext quot = ext_zero();
        for (size_t k = 0; k < NQc; k++) {
            Ext zps = ext_one();
            for (size_t j = 0; j < NQc; j++) {
                if (j == k) continue;
                const uint32_t sjn_inv = finv(sN[j]);
                const Ext num = ext_sub_base(ext_mul_base(zn, sjn_inv), MONTY_R1);
                const uint32_t den = fsub(fmul(sN[k], sjn_inv), MONTY_R1);
                zps = ext_mul(zps, ext_mul_base(num, finv(den)));
            }
            quot = ext_add(quot, ext_mul(zps, recombine(&opq[c][4 * k])));
        }
        if (!ext_eq(ext_mul(acc, ext_inv(zh)), quot)) return reject(10);
    }
    // (b) FRI
    const Ext fa = ch.sample_ext();
    size_t npmax = 8;
    for (int c = 0; c < n; c++) { if (widths[c] > npmax) npmax = widths[c]; if (wp[c] > npmax) npmax = wp[c]; if (pre_w(cs, c) > npmax) npmax = pre_w(cs, c); if (w8[c] > npmax) npmax = w8[c]; }
    std::vector<Ext> fapow(npmax);
    fapow[0] = ext_one();
    for (size_t j = 1; j < npmax; j++) fapow[j] = ext_mul(fapow[j - 1], fa);
    Ext y_loc[MAX_CHIPS], y_nxt[MAX_CHIPS], y_pl[MAX_CHIPS], y_pn[MAX_CHIPS], y_q[MAX_CHIPS];
    Ext s_loc[MAX_CHIPS], s_nxt[MAX_CHIPS], s_pl[MAX_CHIPS], s_pn[MAX_CHIPS], s_q[MAX_CHIPS], znext[MAX_CHIPS];
    Ext y_el[MAX_CHIPS], y_en[MAX_CHIPS], s_el[MAX_CHIPS], s_en[MAX_CHIPS];
    for (int c = 0; c < n; c++) {
        const uint32_t W = widths[c], Pw = pre_w(cs, c);
        y_loc[c] = y_nxt[c] = y_pl[c] = y_pn[c] = y_q[c] = y_el[c] = y_en[c] = ext_zero();
        for (uint32_t j = 0; j < Pw; j++) {
            y_el[c] = ext_add(y_el[c], ext_mul(fapow[j], oel[c][j]));
            y_en[c] = ext_add(y_en[c], ext_mul(fapow[j], oen[c][j]));
        }
        for (uint32_t j = 0; j < W; j++) {
            y_loc[c] = ext_add(y_loc[c], ext_mul(fapow[j], loc[c][j]));
            y_nxt[c] = ext_add(y_nxt[c], ext_mul(fapow[j], nxt[c][j]));
        }
        for (size_t j = 0; j < wp[c]; j++) {
            y_pl[c] = ext_add(y_pl[c], ext_mul(fapow[j], opl[c][j]));
            y_pn[c] = ext_add(y_pn[c], ext_mul(fapow[j], opn[c][j]));
        }
        for (uint32_t j = 0; j < w8[c]; j++) y_q[c] = ext_add(y_q[c], ext_mul(fapow[j], opq[c][j]));
        const uint64_t off0 = height_offset(cs, log_ns, widths, pairs, c), off = off0 + 2 * (uint64_t)Pw;
        s_el[c] = ext_pow(fa, off0); s_en[c] = ext_pow(fa, off0 + Pw);
        s_loc[c] = ext_pow(fa, off); s_nxt[c] = ext_pow(fa, off + W); s_pl[c] = ext_pow(fa, off + 2 * (uint64_t)W);
        s_pn[c] = ext_pow(fa, off + 2 * (uint64_t)W + wp[c]); s_q[c] = ext_pow(fa, off + 2 * (uint64_t)W + 2 * wp[c]);
        znext[c] = ext_mul_base(zeta, two_adic_generator(log_ns[c]));
    }
    std::vector<uint32_t> commits((size_t)L * 8);
    std::vector<Ext> betas(L);
    for (int l = 0; l < L; l++) {
        for (int i = 0; i < 8; i++) { commits[8 * l + i] = to_monty(pf[pos++]); ch.observe(commits[8 * l + i]); }
        betas[l] = ch.sample_ext();
    }
    const Ext final_poly = ext_from_canon(pf + pos);
    pos += 4;
    ch.observe_ext(final_poly);
    const uint32_t witness = pf[pos++];
    ch.observe_canonical(witness);
    if (ch.sample_bits(prm->pow_bits) != 0) return reject(20);
    // as in the single-matrix verifier: indices from the transcript first, then the queries on a few host threads
    const int NQ_ = prm->num_queries;
    std::vector<size_t> indices(NQ_);
    for (int q = 0; q < NQ_; q++) indices[q] = ch.sample_bits(Hmax);
    const size_t pos0 = pos, words_total = len / 4;
    if ((words_total - pos0) % (size_t)NQ_ != 0) return 5;
    const size_t perq = (words_total - pos0) / (size_t)NQ_;
    // groups of 16 queries: the three mixed-height openings and every FRI layer's opening are hashed in lockstep
    const int NG = (NQ_ + 15) / 16;
    std::vector<int> qcode(NQ_, 0);
    auto check_group = [&](int g) -> int {
        const int q0 = 16 * g, cnt = NQ_ - q0 < 16 ? NQ_ - q0 : 16;
        const uint32_t *trow[16][MAX_CHIPS], *qrow[16][MAX_CHIPS], *prow[16][MAX_CHIPS], *prow_all[16][MAX_CHIPS];
        const uint32_t *erow[16][MAX_CHIPS], *erow_all[16][MAX_CHIPS];
        const uint32_t *tpath[16], *ppath[16], *qpath[16], *epath[16];
        size_t qpos[16], index[16], pindex[16], eindex[16];
        int code[16] = {0};
        auto mark = [&](uint32_t mask, int why) { for (int j = 0; j < cnt; j++) if (((mask >> j) & 1u) && !code[j]) code[j] = why; };
        for (int j = 0; j < cnt; j++) {
            size_t pos = pos0 + (size_t)(q0 + j) * perq;
            index[j] = indices[q0 + j];
            pindex[j] = lk ? index[j] >> (Hmax - Hp) : 0;
            eindex[j] = ne ? index[j] >> (Hmax - He) : 0;
            epath[j] = nullptr;
            for (int c = 0; c < n; c++) erow_all[j][c] = nullptr;
            if (ne) {
                for (int k = 0; k < ne; k++) { erow[j][k] = pf + pos; erow_all[j][echip[k]] = erow[j][k]; pos += ew[k]; }
                epath[j] = pf + pos; pos += 8 * (size_t)He;
            }
            for (int c = 0; c < n; c++) { trow[j][c] = pf + pos; pos += widths[c]; prow_all[j][c] = nullptr; }
            tpath[j] = pf + pos; pos += 8 * (size_t)Hmax;
            ppath[j] = nullptr;
            if (lk) {
                for (int k = 0; k < np; k++) { prow[j][k] = pf + pos; prow_all[j][pchip[k]] = prow[j][k]; pos += pw[k]; }
                ppath[j] = pf + pos; pos += 8 * (size_t)Hp;
            }
            for (int c = 0; c < n; c++) { qrow[j][c] = pf + pos; pos += w8[c]; }
            qpath[j] = pf + pos; pos += 8 * (size_t)Hmax;
            qpos[j] = pos;
        }
        if (ne) mark(verify_mixed_x16(eroot, He, cnt, eindex, erow, ew, elh, ne, epath), 33);
        mark(verify_mixed_x16(troot, Hmax, cnt, index, trow, widths, lh, n, tpath), 30);
        if (lk) mark(verify_mixed_x16(proot, Hp, cnt, pindex, prow, pw, plh, np, ppath), 32);
        mark(verify_mixed_x16(qroot, Hmax, cnt, index, qrow, w8, lh, n, qpath), 31);
        Ext folded[16];
        size_t idx[16];
        std::vector<Ext> roh((size_t)16 * 32);
        for (int j = 0; j < cnt; j++) {
            Ext* r_h = roh.data() + (size_t)j * 32;
            for (int h = 0; h < 32; h++) r_h[h] = ext_zero();
            for (int c = 0; c < n; c++) {
                const size_t ic = index[j] >> (Hmax - lh[c]);
                const uint32_t x = fmul(MONTY_GEN, fpow(two_adic_generator(lh[c]), reverse_bits((uint32_t)ic, lh[c])));
                const Ext d1 = ext_inv(ext_neg(ext_sub_base(zeta, x))), d2 = ext_inv(ext_neg(ext_sub_base(znext[c], x)));
                Ext at = ext_zero(), ap = ext_zero(), aq = ext_zero();
                for (uint32_t k = 0; k < widths[c]; k++) at = ext_add(at, ext_mul_base(fapow[k], to_monty(trow[j][c][k])));
                for (size_t k = 0; k < wp[c]; k++) ap = ext_add(ap, ext_mul_base(fapow[k], to_monty(prow_all[j][c][k])));
                for (uint32_t k = 0; k < w8[c]; k++) aq = ext_add(aq, ext_mul_base(fapow[k], to_monty(qrow[j][c][k])));
                Ext r = ext_mul(s_loc[c], ext_mul(ext_sub(at, y_loc[c]), d1));
                r = ext_add(r, ext_mul(s_nxt[c], ext_mul(ext_sub(at, y_nxt[c]), d2)));
                if (wp[c]) {
                    r = ext_add(r, ext_mul(s_pl[c], ext_mul(ext_sub(ap, y_pl[c]), d1)));
                    r = ext_add(r, ext_mul(s_pn[c], ext_mul(ext_sub(ap, y_pn[c]), d2)));
                }
                r = ext_add(r, ext_mul(s_q[c], ext_mul(ext_sub(aq, y_q[c]), d1)));
                if (pwv[c]) {
                    Ext ae = ext_zero();
                    for (uint32_t k = 0; k < pwv[c]; k++) ae = ext_add(ae, ext_mul_base(fapow[k], to_monty(erow_all[j][c][k])));
                    r = ext_add(r, ext_mul(s_el[c], ext_mul(ext_sub(ae, y_el[c]), d1)));
                    r = ext_add(r, ext_mul(s_en[c], ext_mul(ext_sub(ae, y_en[c]), d2)));
                }
                r_h[lh[c]] = ext_add(r_h[lh[c]], r);
            }
            folded[j] = r_h[Hmax];
            idx[j] = index[j];
        }
        uint32_t rowbuf[16][8];
        Ext ev[16][2];
        for (int l = 0; l < L; l++) {
            const int rows_log = Hmax - 1 - l;
            PathBatch b;
            b.count = cnt;
            for (int j = 0; j < cnt; j++) {
                const Ext sib = ext_from_canon(pf + qpos[j]);
                for (int i = 0; i < 4; i++) { rowbuf[j][4 * (idx[j] & 1) + i] = from_monty(folded[j].c[i]); rowbuf[j][4 * ((idx[j] & 1) ^ 1) + i] = pf[qpos[j] + i]; }
                qpos[j] += 4;
                ev[j][idx[j] & 1] = folded[j]; ev[j][(idx[j] & 1) ^ 1] = sib;
                b.index[j] = idx[j] >> 1; b.row[j] = rowbuf[j]; b.path[j] = pf + qpos[j];
                qpos[j] += 8 * (size_t)rows_log;
            }
            mark(verify_paths_x16(&commits[8 * l], rows_log, b, 8, 16), 40 + (l < 50 ? l : 50));
            for (int j = 0; j < cnt; j++) {
                folded[j] = ext_add(fri_fold_row(idx[j] >> 1, rows_log, betas[l], ev[j][0], ev[j][1]), roh[(size_t)j * 32 + rows_log]);
                idx[j] >>= 1;
            }
        }
        for (int j = 0; j < cnt; j++) {
            if (!ext_eq(folded[j], final_poly) && !code[j]) code[j] = 100;
            if (qpos[j] != pos0 + (size_t)(q0 + j + 1) * perq && !code[j]) code[j] = 5;
            qcode[q0 + j] = code[j];
        }
        return 0;
    };
    run_queries(NG, check_group, 1);
    for (int q = 0; q < NQ_; q++) if (qcode[q]) return reject(qcode[q]);
    pos = pos0 + (size_t)NQ_ * perq;
    if (pos * 4 != len) return reject(5);
    return ZKHIP_OK;
}

int zkhip_verify_chips(const uint8_t* proof, size_t len, const int32_t* log_ns, const uint32_t* widths, const int32_t* pairs, const int32_t* partners, int n,
                       const uint32_t* public_values, size_t n_public, const zkhip_params* prm, int* reason) {
    return verify_chips_impl(ChipSet{}, proof, len, log_ns, widths, pairs, partners, n, public_values, n_public, prm, reason);
}
int zkhip_verify_chips_air(const uint8_t* proof, size_t len, const int32_t* log_ns, const uint32_t* widths, const uint32_t* const* programs,
                           const size_t* program_words, int n_chips, const uint32_t* public_values, size_t n_public, const zkhip_params* prm, int* reason) {
    AirView views[MAX_CHIPS];
    const AirView* table[MAX_CHIPS];
    if (chip_programs(programs, program_words, widths, n_chips, n_public, views, table) != ZKHIP_OK) { if (reason) *reason = 1; return ZKHIP_ERR_VERIFY; }
    ChipSet cs;
    cs.air = table;
    return verify_chips_impl(cs, proof, len, log_ns, widths, nullptr, nullptr, n_chips, public_values, n_public, prm, reason);
}

int zkhip_verify_machine(const uint8_t* proof, size_t len, const int32_t* log_ns, const uint32_t* widths, const uint32_t* const* programs,
                         const size_t* program_words, const uint32_t* const* tables, const size_t* table_words, int n_chips,
                         const uint32_t* public_values, size_t n_public, const zkhip_params* prm, int* reason) {
    MachineSetup m;
    if (machine_setup(programs, program_words, tables, table_words, widths, n_chips, n_public, m) != ZKHIP_OK) { if (reason) *reason = 1; return ZKHIP_ERR_VERIFY; }
    ChipSet cs;
    cs.air = m.table; cs.machine = &m.mt;
    return verify_chips_impl(cs, proof, len, log_ns, widths, m.cols, nullptr, n_chips, public_values, n_public, prm, reason);
}

int zkhip_verify_machine_keyed(const uint8_t* proof, size_t len, const int32_t* log_ns, const uint32_t* widths, const uint32_t* pre_widths,
                               const uint32_t root[8], const uint32_t* const* programs, const size_t* program_words, const uint32_t* const* tables,
                               const size_t* table_words, int n_chips, const uint32_t* public_values, size_t n_public, const zkhip_params* prm, int* reason) {
    uint32_t cw[MAX_CHIPS];
    MachineSetup m;
    KeyView kv{};
    if (!root || keyed_widths(widths, pre_widths, n_chips, cw) != ZKHIP_OK ||
        machine_setup(programs, program_words, tables, table_words, cw, n_chips, n_public, m) != ZKHIP_OK) { if (reason) *reason = 1; return ZKHIP_ERR_VERIFY; }
    for (int i = 0; i < 8; i++) { if (root[i] >= P) { if (reason) *reason = 1; return ZKHIP_ERR_VERIFY; } kv.root_m[i] = to_monty(root[i]); }
    for (int c = 0; c < n_chips; c++) kv.pw[c] = pre_widths[c];
    ChipSet cs;
    cs.air = m.table; cs.machine = &m.mt; cs.key = &kv;
    return verify_chips_impl(cs, proof, len, log_ns, widths, m.cols, nullptr, n_chips, public_values, n_public, prm, reason);
}

// the verifier's batched host permutation (p2_x16.cpp) against the scalar one on pseudo-random states: 1 = AVX-512 in use and equal,
// 0 = this CPU lacks it (the verifiers then hash query by query), negative = mismatch.  *ns_x16 / *ns_scalar (optional): time per
// permutation of either form.
int zkhip_host_simd(int enable) { return p2x16_enable(enable != 0) ? 1 : 0; }
// nanoseconds per Poseidon2 permutation of ONE state on this host: form 0 = the scalar permutation, 1 = what the transcripts and host Merkle checks call (the state
// in one AVX-512 register when the CPU has it: p2_x16.cpp p2h_permute, else the scalar form again).  A measurement for docs and tools; no device.
double zkhip_host_permutation_ns(int form) {
    uint32_t s1[16];
    for (int e = 0; e < 16; e++) s1[e] = (uint32_t)(e + 1);
    const int reps = 20000;
    for (int k = 0; k < 200; k++) { if (form) p2_permute(s1); else p2_permute_scalar(s1); }
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; k++) { if (form) p2_permute(s1); else p2_permute_scalar(s1); }
    const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / reps;
    return ns + (s1[0] == 0xFFFFFFFFu ? 1.0 : 0.0);
}
int zkhip_selftest_host_simd(double* ns_x16, double* ns_scalar) {
    if (ns_x16) *ns_x16 = 0;
    if (ns_scalar) *ns_scalar = 0;
    if (!p2x16_available()) return 0;
    uint32_t st[16][16], ref[16][16];
    uint64_t z = 0x9E3779B97F4A7C15ull;
    for (int round = 0; round < 8; round++) {
        for (int e = 0; e < 16; e++)
            for (int j = 0; j < 16; j++) {
                z = z * 6364136223846793005ull + 1442695040888963407ull;
                st[e][j] = round == 0 && e < 2 ? (e ? P - 1 : 0u) : (uint32_t)((z >> 33) % P);     // extremes in the first round
                ref[e][j] = st[e][j];
            }
        p2x16_permute(st);
        for (int j = 0; j < 16; j++) {
            uint32_t s1[16];
            for (int e = 0; e < 16; e++) s1[e] = ref[e][j];
            uint32_t s2[16];
            for (int e = 0; e < 16; e++) s2[e] = s1[e];
            p2_permute_scalar(s1);
            for (int e = 0; e < 16; e++) if (s1[e] != st[e][j]) return fail(ZKHIP_ERR_INTERNAL, "host SIMD permutation differs from the scalar one");
            if (p2h_permute(s2)) for (int e = 0; e < 16; e++) if (s1[e] != s2[e]) return fail(ZKHIP_ERR_INTERNAL, "the one-register host permutation differs from the scalar one");
        }
    }
    const int reps = 2000;
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; k++) p2x16_permute(st);
    const double a = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / (16.0 * reps);
    uint32_t s1[16];
    for (int e = 0; e < 16; e++) s1[e] = st[e][0];
    t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; k++) p2_permute_scalar(s1);
    const double b = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / reps;

    if (ns_x16) *ns_x16 = a;
    if (ns_scalar) *ns_scalar = b + (s1[0] == 0xFFFFFFFFu ? 1 : 0);
    return 1;
}

}  // extern "C"
