// prover.cpp -- whole-shard STARK prover (host orchestration of the gfx950 kernels), the
// host-side Fiat-Shamir transcript, the CPU verifier, and the stage-level C ABI entries.
//
// Stands in for the span the reference times around `client.prove`
// (crates/guest-prover-sp1/src/sp1.rs:115-118) and for `client.verify` (:120); the
// protocol restates p3-uni-stark `prove` + p3-fri TwoAdicFriPcs over the synthetic AIR
// (DESIGN.md sections 3 and 6).  Every heavy step is a HIP kernel on the context's
// stream; the host only runs the duplex challenger (a few dozen permutations) between
// launches, exactly where the protocol forces a round trip (commit -> challenge).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "context.h"
#include "batch.h"
#include "poseidon2.cuh"
#include "air.h"
#include "p2_x16.h"

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
extern std::atomic<uint64_t> g_p2_generation;      // params.cpp: bumped when the Poseidon2 table set changes
static void air_digest_cached(const AirView& a, uint32_t out[8]) {
    struct Entry { std::vector<uint32_t> words; uint64_t generation; uint32_t dg[8]; };
    static std::mutex mu;
    static std::vector<Entry> cache;
    const uint64_t gen = g_p2_generation.load();
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Entry& e : cache)
            if (e.generation == gen && e.words.size() == a.words && memcmp(e.words.data(), a.w, a.words * 4) == 0) { memcpy(out, e.dg, 32); return; }
    }
    air_digest(a, out);
    std::lock_guard<std::mutex> lk(mu);
    if (cache.size() >= 8) cache.erase(cache.begin());
    Entry e;
    e.words.assign(a.w, a.w + a.words);
    e.generation = gen;
    memcpy(e.dg, out, 32);
    cache.push_back(std::move(e));
}

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

static int ensure_domain(zkhip_ctx* ctx, int log_n, int log_blowup = 1) {
    if (ctx->dom_log_n == log_n && ctx->dom_log_blowup == log_blowup) return ZKHIP_OK;
    for (const auto& d : ctx->domains)
        if (d.log_n == log_n && d.log_blowup == log_blowup) {
            ctx->dom_xs = d.xs; ctx->dom_sel_first = d.sel_first; ctx->dom_sel_last = d.sel_last; ctx->dom_itw = d.itw;
            ctx->dom_log_n = log_n; ctx->dom_log_blowup = log_blowup;
            return ZKHIP_OK;
        }
    zkhip_ctx::DomainSet d{log_n, log_blowup, nullptr, nullptr, nullptr, nullptr};
    const size_t m = (size_t)1 << (log_n + log_blowup), mq = (size_t)1 << (log_n + (log_blowup < 2 ? log_blowup : 2));
    ZK_HIP(hipMalloc((void**)&d.xs, m * 4));
    ZK_HIP(hipMalloc((void**)&d.sel_first, mq * 4));
    ZK_HIP(hipMalloc((void**)&d.sel_last, mq * 4));
    ZK_HIP(hipMalloc((void**)&d.itw, (m / 2) * 4));
    ZK_HIP(launch_domain_tables(d.xs, d.sel_first, d.sel_last, d.itw, log_n, log_blowup, ctx->stream));
    ctx->domains.push_back(d);
    ctx->dom_xs = d.xs; ctx->dom_sel_first = d.sel_first; ctx->dom_sel_last = d.sel_last; ctx->dom_itw = d.itw;
    ctx->dom_log_n = log_n; ctx->dom_log_blowup = log_blowup;
    return ZKHIP_OK;
}
// the FRI inverse-twiddle table of a domain is a prefix of the table of any larger one
static int ensure_fold_table(zkhip_ctx* ctx, int log_h) {
    if (ctx->dom_log_n >= 0 && ctx->dom_log_n + ctx->dom_log_blowup >= log_h) return ZKHIP_OK;
    return ensure_domain(ctx, log_h - 1 < 5 ? 5 : log_h - 1, 1);
}

static int d2h(zkhip_ctx* ctx, void* dst, const void* src, size_t bytes) { return dev_d2h(ctx, dst, src, bytes); }
static int h2d(zkhip_ctx* ctx, void* dst, const void* src, size_t bytes) { return dev_h2d(ctx, dst, src, bytes); }

struct LogupIn {
    Ext cumsum = ext_zero();              // last-row constraint S = cumsum
    uint32_t pairs = 0;
    const uint32_t* perm_lde = nullptr;   // [2N][4 (pairs + 1)]
    Ext gamma = ext_zero(), beta = ext_zero();
};
static int run_quotient(zkhip_ctx* ctx, const uint32_t* lde, size_t ld, int log_n, uint32_t width, const Ext& alpha,
                        const LogupIn& lu, uint32_t* out_chunks, uint32_t* lde_out = nullptr, size_t lde_ld = 0) {
    if (ctx->dom_log_n != log_n) ZK_TRY(ensure_domain(ctx, log_n));   // any blowup serves: the quotient domain is a prefix
    const uint32_t G = width / 4;
    // weight of constraint k is alpha^(K-1-k): filled from the last constraint backwards;
    // order: 3 per column group, then (LogUp) L_0 .. L_{Q-1}, T1, T2, T3
    const size_t K = (size_t)3 * G + (lu.pairs ? lu.pairs + 3 : 0);
    std::vector<Ext> ap(K);
    Ext w = ext_one();
    for (size_t k = K; k-- > 0;) { ap[k] = w; w = ext_mul(w, alpha); }
    // device table: per column group 16 words = three weights + the group's constants (g+1, 2g+3, 5g+7)
    // in Montgomery form; then the Q + 3 LogUp weights
    std::vector<uint32_t> tab(16 * (size_t)G + 4 * (K - 3 * (size_t)G));
    for (uint32_t g = 0; g < G; g++) {
        for (int t = 0; t < 3; t++) memcpy(&tab[16 * (size_t)g + 4 * t], ap[3 * (size_t)g + t].c, 16);
        tab[16 * (size_t)g + 12] = to_monty(g + 1);
        tab[16 * (size_t)g + 13] = to_monty(2 * g + 3);
        tab[16 * (size_t)g + 14] = to_monty(5 * g + 7);
        tab[16 * (size_t)g + 15] = 0;
    }
    for (size_t k = 3 * (size_t)G; k < K; k++) memcpy(&tab[16 * (size_t)G + 4 * (k - 3 * (size_t)G)], ap[k].c, 16);
    void* d_ap;
    ZK_TRY(ctx_reserve(ctx, S_APOW_Q, tab.size() * 4, &d_ap));
    ZK_TRY(h2d(ctx, d_ap, tab.data(), tab.size() * 4));
    QuotientArgs q{};
    q.lde = lde; q.ld = ld; q.width = width; q.log_n = log_n;
    q.lanes_per_row = pow2ceil((int)G) > 16 ? 16 : pow2ceil((int)G);
    q.xs = ctx->dom_xs; q.sel_first = ctx->dom_sel_first;
    q.wn_inv = finv(two_adic_generator(log_n));
    const uint32_t gn = fpow(MONTY_GEN, (uint64_t)1 << log_n);
    q.inv_zh_even = finv(fsub(gn, MONTY_R1));
    q.inv_zh_odd = finv(fsub(fneg(gn), MONTY_R1));
    q.alpha_pow = (const uint32_t*)d_ap;
    q.pairs = lu.pairs; q.perm = lu.perm_lde; q.perm_ld = 4 * ((uint64_t)lu.pairs + 1);
    q.gamma = lu.gamma; q.beta = lu.beta; q.cumsum = lu.cumsum; q.sel_last = ctx->dom_sel_last;
    q.out = out_chunks;
    q.lde_out = lde_out; q.lde_ld = lde_ld;
    ZK_HIP(launch_quotient(q, ctx->stream));
    return ZKHIP_OK;
}

// quotient values of a constraint program (air.h): the interpreter kernel, same outputs as run_quotient
static int run_quotient_air(zkhip_ctx* ctx, const AirView& air, const uint32_t* lde, size_t ld, int log_n, uint32_t width,
                            const uint32_t* public_values, const Ext& alpha, uint32_t* out_chunks, uint32_t* lde_out, size_t lde_ld,
                            const Ext& scale = ext_one(), const uint32_t* addend = nullptr) {
    if (ctx->dom_log_n != log_n || ctx->dom_log_blowup < air.lqd) ZK_TRY(ensure_domain(ctx, log_n, air.lqd));
    std::vector<uint32_t> body, weights;
    air_device_image(air, alpha, body, weights, scale);
    std::vector<uint32_t> pub(air.n_public ? air.n_public : 1, 0u);
    for (uint32_t i = 0; i < air.n_public; i++) pub[i] = to_monty(public_values[i]);
    // the flattened form for the term-parallel kernel (up to 64 public values: they travel in every point's LDS slots)
    std::vector<uint32_t> recs;
    if (air.n_public <= 64) air_term_records(air, alpha, recs, scale);
    // one staging buffer: body | weights (16-byte aligned) | public values | term records (16-byte aligned)
    const size_t body_w = (body.size() + 3) & ~(size_t)3, pub_w = (pub.size() + 3) & ~(size_t)3;
    std::vector<uint32_t> stage(body_w + weights.size() + pub_w + recs.size(), 0u);
    memcpy(stage.data(), body.data(), body.size() * 4);
    memcpy(stage.data() + body_w, weights.data(), weights.size() * 4);
    memcpy(stage.data() + body_w + weights.size(), pub.data(), pub.size() * 4);
    if (!recs.empty()) memcpy(stage.data() + body_w + weights.size() + pub_w, recs.data(), recs.size() * 4);
    void* d_stage;
    ZK_TRY(ctx_reserve(ctx, S_APOW_Q, stage.size() * 4, &d_stage));
    ZK_TRY(h2d(ctx, d_stage, stage.data(), stage.size() * 4));
    QuotientAirArgs q{};
    q.lde = lde; q.ld = ld; q.width = width; q.log_n = log_n;
    q.xs = ctx->dom_xs; q.sel_first = ctx->dom_sel_first; q.sel_last = ctx->dom_sel_last;
    q.wn_inv = finv(two_adic_generator(log_n));
    q.log_qd = air.lqd;
    const uint32_t gn = fpow(MONTY_GEN, (uint64_t)1 << log_n), wq = two_adic_generator(air.lqd);
    for (int j = 0; j < (1 << air.lqd); j++) q.inv_zh[j] = finv(fsub(fmul(gn, fpow(wq, (uint64_t)j)), MONTY_R1));   // x^N = g^N w_{2^lqd}^j on chunk j
    q.body = (const uint32_t*)d_stage; q.n_constraints = air.K;
    q.weights = (const uint32_t*)d_stage + body_w; q.pub = (const uint32_t*)d_stage + body_w + weights.size();
    q.out = out_chunks; q.lde_out = lde_out; q.lde_ld = lde_ld;
    q.recs = recs.empty() ? nullptr : (const uint32_t*)d_stage + body_w + weights.size() + pub_w;
    q.n_terms = (uint32_t)(recs.size() / 8); q.n_public = air.n_public;
    q.addend = addend;
    ZK_HIP(launch_quotient_air(q, ctx->stream));
    return ZKHIP_OK;
}

// opens `width` columns at npts points; xw must hold x_q/(x_q - z_k) for the first N rows, stride N
static int run_open(zkhip_ctx* ctx, const uint32_t* lde, size_t ld, int log_n, uint32_t width, const Ext* z, int npts,
                    const uint32_t* xw, uint32_t* d_out) {
    const uint64_t n = (uint64_t)1 << log_n;
    OpenArgs o{};
    o.mat = lde; o.ld = ld; o.width = width; o.rows = n; o.xw = xw; o.xw_stride = n;
    o.tx = pow2ceil((int)width) > 64 ? 64 : pow2ceil((int)width);
    const size_t nchunks = open_chunks(n, width, ld, lde);
    void* part;
    ZK_TRY(ctx_reserve(ctx, S_PARTIAL, nchunks * npts * width * 16, &part));
    o.partial = (uint32_t*)part;
    Ext scale[2] = {ext_zero(), ext_zero()};
    const uint32_t ginv = finv(MONTY_GEN), ninv = finv(to_monty((uint32_t)(n % P)));
    for (int k = 0; k < npts; k++)
        scale[k] = ext_mul_base(ext_sub_base(ext_pow(ext_mul_base(z[k], ginv), n), MONTY_R1), ninv);
    ZK_HIP(launch_open(o, npts, scale[0], scale[1], d_out, ctx->stream));
    return ZKHIP_OK;
}

// one Merkle commitment of a row-major matrix with the shape's hash
static int commit_hw(zkhip_ctx* ctx, const uint32_t* mat, size_t ld, uint32_t width, int log_h, uint32_t* tree, int hw) {
    if (hw == 24) { ZK_HIP(launch_merkle_p24_rowmajor(mat, ld, width, log_h, tree, ctx->stream)); return ZKHIP_OK; }
    MatDesc md{mat, ld, width};
    return op_merkle_commit(ctx, &md, 1, log_h, tree);
}

// FRI commit phase shared by the single-matrix and the multi-chip prover: RL committed layers (rows of 2^K adjacent
// entries): commit, transcript step, fold K times.  `inject` (multi-chip, K = 1): inject[h] is the reduced-opening vector
// of the chips whose LDE has 2^h rows; it is added to the folded vector when that reaches 2^h entries.
static int fri_commit_phase(zkhip_ctx* ctx, Challenger& ch, const Shape& sh, int H, int RL, uint32_t* layers, uint32_t* ltrees,
                            const std::vector<size_t>& layer_off, const std::vector<size_t>& tree_off, uint32_t* fold_tmp, size_t m,
                            const uint32_t* const* inject, uint32_t* pf, size_t& pos) {
    hipStream_t st = ctx->stream;
    const int K = sh.K;
    const size_t arity = (size_t)1 << K;
    uint32_t root[8];
    // The per-layer transcript step (observe the root, sample beta) runs ON THE DEVICE (fri_challenge_kernel), so the
    // whole commit loop is enqueued without a host round trip; afterwards the host replays the same steps on its own
    // challenger from the logged roots and checks that both transcripts agree.  (A/B builds: ZKHIP_FRI_HOST=1 keeps the round trips.)
#ifdef ZKHIP_AB_HOOKS
    static const bool fri_on_host = [] { const char* e = getenv("ZKHIP_FRI_HOST"); return e && atoi(e) != 0; }();
#else
    constexpr bool fri_on_host = false;
#endif
    void* v_chal = nullptr;
    uint32_t *d_betas = nullptr, *d_roots = nullptr;
    DevChallenger* d_chal = nullptr;
    if (!fri_on_host && RL > 0) {
        ZK_TRY(ctx_reserve(ctx, S_CHAL, sizeof(DevChallenger) + (size_t)RL * 12 * 4, &v_chal));
        d_chal = (DevChallenger*)v_chal;
        d_betas = (uint32_t*)((char*)v_chal + sizeof(DevChallenger));
        d_roots = d_betas + 4 * (size_t)RL;
        DevChallenger hc{};
        for (int i = 0; i < 16; i++) hc.state[i] = ch.state[i];
        for (int i = 0; i < 8; i++) { hc.in[i] = ch.in[i]; hc.out[i] = ch.out[i]; }
        hc.n_in = ch.n_in; hc.n_out = ch.n_out;
        ZK_TRY(h2d(ctx, d_chal, &hc, sizeof hc));
    }
    // (a single-workgroup kernel walking all layers of <= 512 rows was tried: no faster than these launches)
    auto enqueue_layers = [&]() -> int {
    for (int l = 0; l < RL; l++) {
        const int lh = H - K * (l + 1);
        const size_t rows = (size_t)1 << lh;
        uint32_t* cur = layers + layer_off[l];
        uint32_t* tree = ltrees + tree_off[l];
        ZK_TRY(commit_hw(ctx, cur, 4 * arity, (uint32_t)(4 * arity), lh, tree, sh.hw));
        Ext beta = ext_zero();
        if (d_chal) {
            ZK_HIP(launch_fri_challenge(d_chal, tree + (2 * rows - 2) * 8, d_betas + 4 * l, d_roots + 8 * l, st));
        } else {
            ZK_TRY(d2h(ctx, root, tree + (2 * rows - 2) * 8, 32));
            for (int i = 0; i < 8; i++) { ch.observe(root[i]); pf[pos++] = from_monty(root[i]); }
            beta = ch.sample_ext();
        }
        // fold by 2 with beta, beta^2, beta^4, ...: f = sum_j X^j f_j(X^(2^K))  ->  sum_j beta^j f_j
        const uint32_t* src = cur;
        for (int j = 0; j < K; j++) {
            const size_t out_cnt = (size_t)1 << (H - K * l - j - 1);
            uint32_t* dst = (j == K - 1) ? layers + layer_off[l + 1] : fold_tmp + ((j & 1) ? 4 * (m / 2) : 0);
            if (d_chal) ZK_HIP(launch_fri_fold_dev(src, dst, ctx->dom_itw, out_cnt, d_betas + 4 * l, j, st));
            else ZK_HIP(launch_fri_fold(src, dst, ctx->dom_itw, out_cnt, beta, st));
            src = dst;
            beta = ext_mul(beta, beta);
        }
        const int reached = H - K * (l + 1);
        if (inject && inject[reached]) ZK_HIP(launch_ext_add(layers + layer_off[l + 1], inject[reached], (uint64_t)1 << reached, st));
    }
    return ZKHIP_OK;
    };
    // With the transcript on the device the loop above is a fixed sequence of ~12 small launches per layer that depends only on
    // sizes and workspace addresses: it is captured once into a HIP graph and replayed with one launch per proof
    // (A/B builds: ZKHIP_FRI_GRAPH=0 keeps the plain launches); capture is thread-local, other contexts' threads are not affected.
#ifdef ZKHIP_AB_HOOKS
    static const bool use_graph = [] { const char* e = getenv("ZKHIP_FRI_GRAPH"); return !e || atoi(e) != 0; }();
#else
    constexpr bool use_graph = true;
#endif
    if (d_chal && use_graph && !t_batcher) {                   // lock-step members launch one by one: their launches merge across the batch
        std::vector<uint64_t> key = {(uint64_t)H, (uint64_t)K, (uint64_t)RL, (uint64_t)sh.hw, (uint64_t)m, (uint64_t)(uintptr_t)layers,
                                     (uint64_t)(uintptr_t)ltrees, (uint64_t)(uintptr_t)fold_tmp, (uint64_t)(uintptr_t)d_chal,
                                     (uint64_t)(uintptr_t)ctx->dom_itw};
        for (size_t o : layer_off) key.push_back(o);
        for (size_t o : tree_off) key.push_back(o);
        if (inject) for (int h = 0; h <= H; h++) key.push_back((uint64_t)(uintptr_t)inject[h]);
        if (!ctx->fri_graph_exec || ctx->fri_graph_key != key) {
            if (ctx->fri_graph_exec) { (void)hipGraphExecDestroy(ctx->fri_graph_exec); ctx->fri_graph_exec = nullptr; }
            hipGraph_t graph = nullptr;
            ZK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue_layers();
            const hipError_t ce = hipStreamEndCapture(st, &graph);
            if (rc != ZKHIP_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            ZK_HIP(ce);
            const hipError_t ie = hipGraphInstantiate(&ctx->fri_graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            ZK_HIP(ie);
            ctx->fri_graph_key = key;
        }
        ZK_HIP(hipGraphLaunch(ctx->fri_graph_exec, st));
    } else {
        ZK_TRY(enqueue_layers());
    }
    if (d_chal) {
        std::vector<uint32_t> log((size_t)RL * 12);
        ZK_TRY(d2h(ctx, log.data(), d_betas, log.size() * 4));
        for (int l = 0; l < RL; l++) {
            const uint32_t* r = log.data() + 4 * (size_t)RL + 8 * (size_t)l;
            for (int i = 0; i < 8; i++) { ch.observe(r[i]); pf[pos++] = from_monty(r[i]); }
            const Ext beta = ch.sample_ext();
            const uint32_t* b = log.data() + 4 * (size_t)l;
            if (beta.c[0] != b[0] || beta.c[1] != b[1] || beta.c[2] != b[2] || beta.c[3] != b[3])
                return fail(ZKHIP_ERR_INTERNAL, "prove: device and host transcripts disagree in the FRI commit phase");
        }
    }
    return ZKHIP_OK;
}

// proof of work: smallest witness, searched 2^20 candidates per launch; observes it
static int grind_witness(zkhip_ctx* ctx, Challenger& ch, int pow_bits, uint32_t* out) {
    uint32_t witness = 0xFFFFFFFFu;
    GrindArgs ga{};
    for (int i = 0; i < 16; i++) ga.state[i] = ch.state[i];
    for (int i = 0; i < ch.n_in; i++) ga.state[i] = ch.in[i];
    ga.slot = ch.n_in;
    ga.mask = (1u << pow_bits) - 1u;
    void* v_res;
    ZK_TRY(ctx_reserve(ctx, S_GATHER_OUT, 4, &v_res));
    ZK_TRY(dev_memset(ctx, v_res, 0xFF, 4));
    // candidates are scanned in order, so the first batch that contains a hit contains the smallest witness; a batch of 4 * 2^bits
    // candidates has one with probability 1 - e^-4 = 98 % (2^20 candidates = 0.12 ms of permutations would mostly be wasted)
    uint64_t want = (uint64_t)4 << pow_bits;
    if (want < (1u << 14)) want = 1u << 14;
    const uint32_t batch = (uint32_t)(want > (1u << 20) ? (1u << 20) : want);
    // (lock-step batches, batch.h: members loop until EVERY member has its witness -- later launches cannot lower a minimum found
    // among smaller candidates -- so that the launch sequences stay identical)
    for (uint64_t base = 0; base < P; base += batch) {
        ZK_HIP(launch_grind(ga, (uint32_t)base, batch, (uint32_t*)v_res, ctx->stream));
        ZK_TRY(d2h(ctx, &witness, v_res, 4));
        const bool found = witness != 0xFFFFFFFFu;
        if (t_batcher ? t_batcher->all(found) : found) break;
    }
    if (witness == 0xFFFFFFFFu) return fail(ZKHIP_ERR_INTERNAL, "prove: no proof-of-work witness found");
    ch.observe_canonical(witness);
    if (ch.sample_bits(pow_bits) != 0) return fail(ZKHIP_ERR_INTERNAL, "prove: device and host disagree on the PoW witness");
    *out = witness;
    return ZKHIP_OK;
}

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

}  // namespace zk

using namespace zk;

#define CHECK_CTX(ctx)                                                  \
    do {                                                                \
        if (!(ctx)) return fail(ZKHIP_ERR_INVALID, "null context");     \
        ZK_HIP(hipSetDevice((ctx)->device));                            \
    } while (0)

extern "C" {

int zkhip_quotient_values(zkhip_ctx* ctx, const uint32_t* d_lde, size_t ld, int log_n, uint32_t width,
                          const uint32_t alpha[4], uint32_t* d_out) {
    CHECK_CTX(ctx);
    zkhip_params prm{1, 1, 0, 0, 0, 0, 0, 0};
    ZK_TRY(check_shape(log_n, width, &prm));
    if (!d_lde || !d_out || !alpha || ld < width) return fail(ZKHIP_ERR_INVALID, "quotient_values: bad arguments");
    // kernel writes natural-order chunks; this entry point returns the bit-reversed
    // layout of the LDE (row p), so gather it back: p = bitrev(2 j + k)
    const size_t n = (size_t)1 << log_n;
    void* chunks;
    ZK_TRY(ctx_reserve(ctx, S_QCHUNK, 2 * n * 16, &chunks));
    Ext a{{alpha[0], alpha[1], alpha[2], alpha[3]}};
    ZK_TRY(run_quotient(ctx, d_lde, ld, log_n, width, a, LogupIn{}, (uint32_t*)chunks));
    // the chunk layout is [k][j]; position p holds e = bitrev(p) = 2 j + k, and bit-reversing a
    // (log_n+1)-bit index moves k to the top bit: p = k * N + bitrev_n(j).  Use the gather kernel.
    std::vector<GatherDesc> descs(2 * n);
    for (size_t k = 0; k < 2; k++)
        for (size_t j = 0; j < n; j++) {
            size_t p = k * n + reverse_bits((uint32_t)j, log_n);
            descs[p] = GatherDesc{(const uint32_t*)chunks + (k * n + j) * 4, (uint32_t)(p * 4), 4};
        }
    void* dd;
    ZK_TRY(ctx_reserve(ctx, S_GATHER_DESC, descs.size() * sizeof(GatherDesc), &dd));
    ZK_TRY(h2d(ctx, dd, descs.data(), descs.size() * sizeof(GatherDesc)));
    ZK_HIP(launch_gather((const GatherDesc*)dd, (uint32_t)descs.size(), d_out, ctx->stream));   // -> canonical
    ZK_HIP(launch_convert(d_out, d_out, 2 * n * 4, true, ctx->stream));                        // back to Montgomery
    return ZKHIP_OK;
}

static int run_perm_trace(zkhip_ctx* ctx, const uint32_t* d_trace, size_t ld, int log_n, uint32_t pairs,
                          const Ext& gamma, const Ext& beta, uint32_t* d_out) {
    PermArgs pa{};
    pa.trace = d_trace; pa.ld = ld; pa.rows = (uint64_t)1 << log_n; pa.pairs = pairs;
    pa.gamma = gamma; pa.beta = beta; pa.out = d_out; pa.out_ld = 4 * ((uint64_t)pairs + 1);
    void* scratch;
    ZK_TRY(ctx_reserve(ctx, S_GATHER_OUT, ((pa.rows + 255) / 256) * 16, &scratch));
    ZK_HIP(launch_perm_trace(pa, (uint32_t*)scratch, ctx->stream));
    return ZKHIP_OK;
}

// lookups as data (machine mode): the interaction records of one chip on the device, then the generic permutation trace
static int lookup_args(zkhip_ctx* ctx, const LookupView& lv, const Ext& gamma, const Ext& beta, const std::vector<uint32_t>& weights, LookupArgs* lk,
                       const uint32_t** d_weights) {
    std::vector<uint32_t> recs;
    lookup_device_records(lv, recs);
    const size_t rec_w = (recs.size() + 3) & ~(size_t)3;
    std::vector<uint32_t> stage(rec_w + weights.size(), 0u);
    memcpy(stage.data(), recs.data(), recs.size() * 4);
    if (!weights.empty()) memcpy(stage.data() + rec_w, weights.data(), weights.size() * 4);
    void* d_stage;
    ZK_TRY(ctx_reserve(ctx, S_LOOKUP, stage.size() * 4, &d_stage));
    ZK_TRY(h2d(ctx, d_stage, stage.data(), stage.size() * 4));
    lk->table = (const uint32_t*)d_stage; lk->ni = lv.ni; lk->cols = lv.cols; lk->gamma = gamma;
    lk->bpow[0] = ext_one();
    for (int t = 1; t < 9; t++) lk->bpow[t] = ext_mul(lk->bpow[t - 1], beta);
    if (d_weights) *d_weights = (const uint32_t*)d_stage + rec_w;
    return ZKHIP_OK;
}
static int run_lookup_perm(zkhip_ctx* ctx, const uint32_t* d_trace, size_t ld, int log_n, const LookupView& lv, const Ext& gamma, const Ext& beta,
                           uint32_t* d_out) {
    MachinePermArgs a{};
    ZK_TRY(lookup_args(ctx, lv, gamma, beta, {}, &a.lk, nullptr));
    a.trace = d_trace; a.ld = ld; a.rows = (uint64_t)1 << log_n; a.out = d_out; a.out_ld = 4 * ((uint64_t)lv.cols + 1);
    void* scratch;
    ZK_TRY(ctx_reserve(ctx, S_GATHER_OUT, ((a.rows + 255) / 256) * 16, &scratch));
    ZK_HIP(launch_perm_trace_machine(a, (uint32_t*)scratch, ctx->stream));
    return ZKHIP_OK;
}
// the chip's lookup constraints on its quotient domain, folded with the LAST cols + 3 powers of alpha (column j: alpha^(cols + 2 - j);
// is_first, is_transition, is_last rows: alpha^2, alpha, 1)
static int run_lookup_addend(zkhip_ctx* ctx, const uint32_t* lde, size_t ld, const uint32_t* perm_lde, size_t perm_ld, int log_n, int log_qd, const LookupView& lv,
                             const Ext& gamma, const Ext& beta, const Ext& alpha, const Ext& cumsum, uint32_t* d_addend) {
    if (ctx->dom_log_n != log_n || ctx->dom_log_blowup < log_qd) ZK_TRY(ensure_domain(ctx, log_n, log_qd));
    std::vector<uint32_t> weights(4 * ((size_t)lv.cols + 3));
    Ext w = ext_one();
    for (size_t k = lv.cols + 3; k-- > 0;) { for (int i = 0; i < 4; i++) weights[4 * k + i] = w.c[i]; w = ext_mul(w, alpha); }
    MachineQuotArgs a{};
    ZK_TRY(lookup_args(ctx, lv, gamma, beta, weights, &a.lk, &a.weights));
    a.lde = lde; a.ld = ld; a.perm = perm_lde; a.perm_ld = perm_ld; a.log_n = log_n; a.log_qd = log_qd;
    a.xs = ctx->dom_xs; a.sel_first = ctx->dom_sel_first; a.sel_last = ctx->dom_sel_last; a.wn_inv = finv(two_adic_generator(log_n));
    a.cumsum = cumsum; a.addend = d_addend;
    ZK_HIP(launch_lookup_addend(a, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_perm_trace(zkhip_ctx* ctx, const uint32_t* d_trace, size_t ld, int log_n, uint32_t width, int pairs,
                     const uint32_t gamma[4], const uint32_t beta[4], uint32_t* d_out) {
    CHECK_CTX(ctx);
    if (!d_trace || !d_out || !gamma || !beta || log_n < 0 || log_n > 24 || pairs < 1 || pairs > 64 ||
        (uint32_t)pairs * 8 > width || ld < width || (ld % 4) != 0)
        return fail(ZKHIP_ERR_INVALID, "perm_trace: bad arguments");
    Ext g{{gamma[0], gamma[1], gamma[2], gamma[3]}}, b{{beta[0], beta[1], beta[2], beta[3]}};
    return run_perm_trace(ctx, d_trace, ld, log_n, (uint32_t)pairs, g, b, d_out);
}

int zkhip_open_at(zkhip_ctx* ctx, const uint32_t* d_lde, size_t ld, int log_n, int log_blowup, uint32_t width,
                  const uint32_t* z, int npoints, uint32_t* h_out) {
    CHECK_CTX(ctx);
    if (log_n < 5 || log_n > MAX_LOG_ROWS || log_blowup != 1) return fail(ZKHIP_ERR_INVALID, "open_at: log_n in [5,22], log_blowup = 1");
    if (!d_lde || !z || !h_out || width == 0 || ld < width || npoints < 1 || npoints > 2)
        return fail(ZKHIP_ERR_INVALID, "open_at: bad arguments (1 or 2 points)");
    if (ctx->dom_log_n != log_n) ZK_TRY(ensure_domain(ctx, log_n));
    const uint64_t m = (uint64_t)2 << log_n, n = (uint64_t)1 << log_n;
    Ext zz[2];
    for (int k = 0; k < npoints; k++) zz[k] = Ext{{z[4 * k], z[4 * k + 1], z[4 * k + 2], z[4 * k + 3]}};
    if (npoints == 1) zz[1] = zz[0];
    void *dinv, *dout;
    ZK_TRY(ctx_reserve(ctx, S_DINV, 2 * (m + n) * 16, &dinv));
    ZK_TRY(ctx_reserve(ctx, S_OPEN_OUT, (size_t)npoints * width * 16, &dout));
    uint32_t* xw = (uint32_t*)dinv + 8 * m;
    ZK_HIP(launch_inv_denominators(ctx->dom_xs, m, zz[0], zz[1], npoints, (uint32_t*)dinv, xw, n, ctx->stream));
    ZK_TRY(run_open(ctx, d_lde, ld, log_n, width, zz, npoints, xw, (uint32_t*)dout));
    return d2h(ctx, h_out, dout, (size_t)npoints * width * 16);
}

int zkhip_fri_fold(zkhip_ctx* ctx, const uint32_t* d_in, int log_h, const uint32_t beta[4], uint32_t* d_out) {
    CHECK_CTX(ctx);
    if (!d_in || !d_out || !beta || log_h < 1 || log_h > MAX_LOG_ROWS + 3) return fail(ZKHIP_ERR_INVALID, "fri_fold: bad arguments");
    ZK_TRY(ensure_fold_table(ctx, log_h));
    Ext b{{beta[0], beta[1], beta[2], beta[3]}};
    ZK_HIP(launch_fri_fold(d_in, d_out, ctx->dom_itw, (uint64_t)1 << (log_h - 1), b, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_fri_fold_k(zkhip_ctx* ctx, const uint32_t* d_in, int log_h, int log_arity, const uint32_t beta[4], uint32_t* d_out) {
    CHECK_CTX(ctx);
    if (!d_in || !d_out || !beta || log_arity < 1 || log_arity > 6 || log_h < log_arity || log_h > MAX_LOG_ROWS + 3)
        return fail(ZKHIP_ERR_INVALID, "fri_fold_k: bad arguments");
    ZK_TRY(ensure_fold_table(ctx, log_h));
    // f = sum_j X^j f_j(X^(2^k)); folding by 2 with b, then b^2, b^4, ... leaves sum_j b^j f_j
    void* tmp;
    ZK_TRY(ctx_reserve(ctx, S_PARTIAL, ((size_t)1 << log_h) * 16, &tmp));
    uint32_t* ping = (uint32_t*)tmp;
    uint32_t* pong = ping + ((size_t)2 << log_h);          // second half of the scratch
    Ext b{{beta[0], beta[1], beta[2], beta[3]}};
    const uint32_t* src = d_in;
    for (int j = 0; j < log_arity; j++) {
        const int lh = log_h - j;
        uint32_t* dst = (j == log_arity - 1) ? d_out : ((j & 1) ? pong : ping);
        ZK_HIP(launch_fri_fold(src, dst, ctx->dom_itw, (uint64_t)1 << (lh - 1), b, ctx->stream));
        src = dst;
        b = ext_mul(b, b);
    }
    return ZKHIP_OK;
}

// 8 canonical words binding (input, program): overwrite-mode Poseidon2 sponge over 3-byte limbs, both fields length-prefixed,
// domain-separated by "ZKT".  Host only (no device needed): the glue on either side of the FFI derives the same public values.
int zkhip_request_digest(const uint8_t* input, size_t input_len, const uint8_t* program, size_t program_len, uint32_t out[8]) {
    if (!out || (input_len && !input) || (program_len && !program)) return fail(ZKHIP_ERR_INVALID, "request_digest: null pointer");
    uint32_t st[16] = {0};
    int pos = 0;
    auto absorb = [&](uint32_t canonical) {
        st[pos++] = to_monty(canonical);
        if (pos == 8) { p2_permute(st); pos = 0; }
    };
    auto absorb_bytes = [&](const uint8_t* b, size_t n) {
        absorb((uint32_t)(n & 0xFFFFFF));
        absorb((uint32_t)((uint64_t)n >> 24) & 0xFFFFFF);
        for (size_t i = 0; i < n; i += 3) {
            uint32_t v = b[i];
            if (i + 1 < n) v |= (uint32_t)b[i + 1] << 8;
            if (i + 2 < n) v |= (uint32_t)b[i + 2] << 16;
            absorb(v);
        }
    };
    absorb(0x5A4B54);   // "ZKT"
    absorb_bytes(input, input_len);
    absorb_bytes(program, program_len);
    if (pos) p2_permute(st);
    for (int i = 0; i < 8; i++) out[i] = from_monty(st[i]);
    return ZKHIP_OK;
}

size_t zkhip_proof_size(int log_n, uint32_t width, const zkhip_params* prm, size_t n_public) {
    (void)n_public;
    if (check_shape(log_n, width, prm) != ZKHIP_OK) return 0;
    return proof_words(log_n, width, prm) * 4;
}

static int prove_shard_impl(zkhip_ctx* ctx, const uint32_t* d_trace, size_t ld, int log_n, uint32_t width,
                            const uint32_t* public_values, size_t n_public, const zkhip_params* prm,
                            uint8_t* proof, size_t cap, size_t* len, const AirView* air) {
    CHECK_CTX(ctx);
    ZK_TRY(check_shape(log_n, width, prm));
    if (!d_trace || !proof || !len || ld < width || (n_public && !public_values)) return fail(ZKHIP_ERR_INVALID, "prove_shard: bad arguments");
    for (size_t i = 0; i < n_public; i++) if (public_values[i] >= P) return fail(ZKHIP_ERR_INVALID, "prove_shard: public values must be canonical");
    if (air && prm->logup_pairs) return fail(ZKHIP_ERR_INVALID, "prove_shard_air: a constraint program excludes the built-in lookup argument (logup_pairs must be 0)");
    if (air && prm->code_width) return fail(ZKHIP_ERR_INVALID, "prove_shard_air: the code / data group split belongs to the built-in prover (code_width must be 0)");
    const int lqd = air ? air->lqd : 1;                      // log2 of the number of quotient chunks
    if (lqd > prm->log_blowup) return fail(ZKHIP_ERR_INVALID, "prove_shard_air: constraints of degree 4 or 5 need log_blowup >= 2 (the quotient domain must lie inside the committed LDE domain)");
    const size_t NQ = (size_t)1 << lqd, QW = 4 * NQ;
    const size_t need = proof_words(log_n, width, prm, air != nullptr, lqd) * 4;
    if (cap < need) return fail(ZKHIP_ERR_BUFFER, "prove_shard: proof buffer too small (see zkhip_proof_size)");
    *len = 0;
    Shape sh;
    shape_of(log_n, prm, sh);
    const int H = log_n + sh.b, Hq = log_n + lqd, Q = prm->num_queries;   // LDE domain 2^H, quotient domain 2^Hq
    const int RL = sh.R, K = sh.K;                                          // committed FRI layers, folds per layer
    const size_t n = (size_t)1 << log_n, m = (size_t)1 << H, arity = (size_t)1 << K;
    hipStream_t st = ctx->stream;
    ZK_TRY(ensure_domain(ctx, log_n, sh.b));

    uint32_t* pf = (uint32_t*)proof;
    size_t pos = 0;
    const uint32_t LQ = (uint32_t)prm->logup_pairs;          // LogUp pairs (0 = none)
    const size_t wp = LQ ? 4 * ((size_t)LQ + 1) : 0;         // permutation-trace width in words
    const uint32_t CW = sh.cw;
    pf[pos++] = PROOF_MAGIC; pf[pos++] = air ? 7u : (CW ? 8u : (sh.ext ? 3u : (LQ ? 2u : PROOF_VERSION))); pf[pos++] = (uint32_t)log_n; pf[pos++] = width;
    pf[pos++] = (uint32_t)prm->log_blowup; pf[pos++] = (uint32_t)Q; pf[pos++] = (uint32_t)prm->pow_bits; pf[pos++] = (uint32_t)n_public;
    if (sh.ext || air) { pf[pos++] = LQ; pf[pos++] = (uint32_t)sh.K; pf[pos++] = (uint32_t)sh.F; pf[pos++] = (uint32_t)sh.hw; }
    else if (LQ) pf[pos++] = LQ;
    if (air) { air_digest_cached(*air, pf + pos); pos += 8; }
    if (CW) pf[pos++] = CW;

    Challenger ch;
    transcript_init(ch, log_n, width, prm, n_public, sh, air);
    uint32_t root[8];

    // ---- 1. commit the trace: LDE on g <w_2N> (bit-reversed rows) + Merkle tree
    void *v_tlde, *v_ttree;
    ZK_TRY(ctx_reserve(ctx, S_TLDE, m * width * 4, &v_tlde));
    ZK_TRY(ctx_reserve(ctx, S_TTREE, (2 * m - 1) * 32, &v_ttree));
    uint32_t* tlde = (uint32_t*)v_tlde; uint32_t* ttree = (uint32_t*)v_ttree;
    ZK_TRY(op_coset_lde(ctx, d_trace, ld, tlde, width, log_n, width, sh.b, MONTY_GEN));
    uint32_t* ctree = nullptr;
    if (CW) {
        // RISC Zero's group order: the code columns [0, CW) and the data columns [CW, width) of the SAME row-major LDE get a tree
        // each (a leaf hashes a column range of a row: row pitch `width`, no copy); the code root is committed and observed first
        void* v_ctree;
        ZK_TRY(ctx_reserve(ctx, S_CTREE, (2 * m - 1) * 32, &v_ctree));
        ctree = (uint32_t*)v_ctree;
        ZK_TRY(commit_hw(ctx, tlde, width, CW, H, ctree, sh.hw));
        ZK_TRY(d2h(ctx, root, ctree + (2 * m - 2) * 8, 32));
        for (int i = 0; i < 8; i++) { ch.observe(root[i]); pf[pos++] = from_monty(root[i]); }
        ZK_TRY(commit_hw(ctx, tlde + CW, width, width - CW, H, ttree, sh.hw));
    } else ZK_TRY(commit_hw(ctx, tlde, width, width, H, ttree, sh.hw));
    ZK_TRY(d2h(ctx, root, ttree + (2 * m - 2) * 8, 32));
    for (int i = 0; i < 8; i++) { ch.observe(root[i]); pf[pos++] = ctx->debug.trace_root[i] = from_monty(root[i]); }
    for (size_t i = 0; i < n_public; i++) ch.observe_canonical(public_values[i]);

    // ---- 1b. LogUp: lookup challenges, permutation trace (per-row inverses + prefix sum), its commitment
    LogupIn lu;
    uint32_t *plde = nullptr, *ptree = nullptr;
    if (LQ) {
        lu.pairs = LQ;
        lu.gamma = ch.sample_ext();
        lu.beta = ch.sample_ext();
        void *v_perm, *v_plde, *v_ptree;
        ZK_TRY(ctx_reserve(ctx, S_PERM, n * wp * 4, &v_perm));
        ZK_TRY(ctx_reserve(ctx, S_PLDE, m * wp * 4, &v_plde));
        ZK_TRY(ctx_reserve(ctx, S_PTREE, (2 * m - 1) * 32, &v_ptree));
        plde = (uint32_t*)v_plde; ptree = (uint32_t*)v_ptree;
        ZK_TRY(run_perm_trace(ctx, d_trace, ld, log_n, LQ, lu.gamma, lu.beta, (uint32_t*)v_perm));
        ZK_TRY(op_coset_lde(ctx, (const uint32_t*)v_perm, wp, plde, wp, log_n, (uint32_t)wp, sh.b, MONTY_GEN));
        ZK_TRY(commit_hw(ctx, plde, wp, (uint32_t)wp, H, ptree, sh.hw));
        ZK_TRY(d2h(ctx, root, ptree + (2 * m - 2) * 8, 32));
        for (int i = 0; i < 8; i++) { ch.observe(root[i]); pf[pos++] = from_monty(root[i]); }
        lu.perm_lde = plde;
    }

    // ---- 2. constraint challenge, quotient chunks, their LDE + commitment
    const Ext alpha = ch.sample_ext();
    void *v_qchunk, *v_qlde, *v_qtree;
    ZK_TRY(ctx_reserve(ctx, S_QCHUNK, NQ * n * 16, &v_qchunk));
    ZK_TRY(ctx_reserve(ctx, S_QLDE, m * QW * 4, &v_qlde));
    ZK_TRY(ctx_reserve(ctx, S_QTREE, (2 * m - 1) * 32, &v_qtree));
    uint32_t* qchunk = (uint32_t*)v_qchunk; uint32_t* qlde = (uint32_t*)v_qlde; uint32_t* qtree = (uint32_t*)v_qtree;
    // With blowup 2 the LDE domain g <w_2N> is exactly the two cosets the chunks live on: on its own coset a chunk's extension is
    // the quotient value itself (the kernel writes it straight into the LDE matrix), only the OTHER coset needs a transform.
    const bool own_coset_direct = sh.b == 1;           // (then lqd == 1 too)
    if (air) ZK_TRY(run_quotient_air(ctx, *air, tlde, width, log_n, width, public_values, alpha, qchunk, own_coset_direct ? qlde : nullptr, 8));
    else ZK_TRY(run_quotient(ctx, tlde, width, log_n, width, alpha, lu, qchunk, own_coset_direct ? qlde : nullptr, 8));
    {
        // the quotient kernel works on the first 2N rows of the LDE: they are the coset g <w_2N>, bit-reversed
        const uint32_t w2n = two_adic_generator(Hq);
        for (int k = 0; k < (int)NQ; k++) {
            if (own_coset_direct) {
                // rows [(1-k) N, (2-k) N) of the LDE = coset 1-k = (g w_2N^(1-k)) <w_N>, relative to the chunk's own coset: w_2N^(1-2k)
                ZK_TRY(op_coset_lde(ctx, qchunk + (size_t)k * n * 4, 4, qlde + (size_t)(1 - k) * n * 8 + 4 * k, 8, log_n, 4, 0, k == 0 ? w2n : finv(w2n)));
                continue;
            }
            // chunk k lives on (g w_{2^Hq}^k) <w_N>; extend it to the LDE domain g <w_M>: shift = g / (g w^k)
            const uint32_t shift = finv(fpow(w2n, (uint64_t)k));
            ZK_TRY(op_coset_lde(ctx, qchunk + (size_t)k * n * 4, 4, qlde + 4 * k, QW, log_n, 4, sh.b, shift));
        }
        ZK_TRY(commit_hw(ctx, qlde, QW, (uint32_t)QW, H, qtree, sh.hw));
    }
    ZK_TRY(d2h(ctx, root, qtree + (2 * m - 2) * 8, 32));
    for (int i = 0; i < 8; i++) { ch.observe(root[i]); pf[pos++] = ctx->debug.quotient_root[i] = from_monty(root[i]); }

    // ---- 3. out-of-domain point, openings
    const Ext zeta = ch.sample_ext();
    const Ext zpts[2] = {zeta, ext_mul_base(zeta, two_adic_generator(log_n))};
    void *v_dinv, *v_open;
    ZK_TRY(ctx_reserve(ctx, S_DINV, 2 * (m + n) * 16, &v_dinv));
    ZK_TRY(ctx_reserve(ctx, S_OPEN_OUT, (2 * (size_t)width + 2 * wp + QW) * 16, &v_open));
    uint32_t* dinv = (uint32_t*)v_dinv; uint32_t* d_open = (uint32_t*)v_open;
    uint32_t* xw = dinv + 8 * m;       // x_q / (x_q - z_k) for the N rows the openings sum over
    ZK_HIP(launch_inv_denominators(ctx->dom_xs, m, zpts[0], zpts[1], 2, dinv, xw, n, st));
    ZK_TRY(run_open(ctx, tlde, width, log_n, width, zpts, 2, xw, d_open));
    ZK_TRY(dev_sync(ctx));   // S_PARTIAL is reused by the next call
    if (LQ) {
        ZK_TRY(run_open(ctx, plde, wp, log_n, (uint32_t)wp, zpts, 2, xw, d_open + 8 * (size_t)width));
        ZK_TRY(dev_sync(ctx));
    }
    ZK_TRY(run_open(ctx, qlde, QW, log_n, (uint32_t)QW, zpts, 1, xw, d_open + 8 * (size_t)width + 8 * wp));
    std::vector<uint32_t> opened((2 * (size_t)width + 2 * wp + QW) * 4);
    ZK_TRY(d2h(ctx, opened.data(), d_open, opened.size() * 4));
    for (size_t i = 0; i < opened.size(); i++) { ch.observe(opened[i]); pf[pos++] = from_monty(opened[i]); }
    const Ext* op_loc = (const Ext*)opened.data();
    const Ext* op_nxt = op_loc + width;
    const Ext* op_pl = op_nxt + width;
    const Ext* op_pn = op_pl + wp;
    const Ext* op_q = op_pn + wp;

    // ---- 4. FRI input: alpha-batched reduced openings at every LDE point
    const Ext fa = ch.sample_ext();
    size_t np = width > QW ? width : QW;
    if (wp > np) np = wp;
    std::vector<Ext> fapow(np);
    fapow[0] = ext_one();
    for (size_t j = 1; j < np; j++) fapow[j] = ext_mul(fapow[j - 1], fa);
    ReducedArgs ra{};
    ra.y_loc = ra.y_next = ra.y_pl = ra.y_pn = ra.y_q = ext_zero();
    ra.off_loc = ext_one(); ra.accumulate = 0;
    for (size_t j = 0; j < width; j++) {
        ra.y_loc = ext_add(ra.y_loc, ext_mul(fapow[j], op_loc[j]));
        ra.y_next = ext_add(ra.y_next, ext_mul(fapow[j], op_nxt[j]));
    }
    for (size_t j = 0; j < wp; j++) {
        ra.y_pl = ext_add(ra.y_pl, ext_mul(fapow[j], op_pl[j]));
        ra.y_pn = ext_add(ra.y_pn, ext_mul(fapow[j], op_pn[j]));
    }
    for (size_t j = 0; j < QW; j++) ra.y_q = ext_add(ra.y_q, ext_mul(fapow[j], op_q[j]));
    // batching offsets: trace@zeta 0, trace@zeta_next W, [perm@zeta 2W, perm@zeta_next 2W+Wp], quotient 2W+2Wp
    ra.off_next = ext_pow(fa, width);
    ra.off_pl = ext_pow(fa, 2 * (uint64_t)width);
    ra.off_pn = ext_pow(fa, 2 * (uint64_t)width + wp);
    ra.off_q = ext_pow(fa, 2 * (uint64_t)width + 2 * wp);
    void *v_apf, *v_layers, *v_ltrees;
    ZK_TRY(ctx_reserve(ctx, S_APOW_F, np * 16, &v_apf));
    ZK_TRY(h2d(ctx, v_apf, fapow.data(), np * 16));
    ZK_TRY(ctx_reserve(ctx, S_FRI_LAYERS, 2 * m * 16, &v_layers));
    ZK_TRY(ctx_reserve(ctx, S_FRI_TREES, 2 * m * 32, &v_ltrees));
    uint32_t* layers = (uint32_t*)v_layers; uint32_t* ltrees = (uint32_t*)v_ltrees;
    ra.tlde = tlde; ra.t_ld = width; ra.width = width; ra.qlde = qlde; ra.q_ld = QW; ra.q_width = (uint32_t)QW; ra.rows = m;
    ra.plde = plde; ra.p_ld = wp; ra.p_width = (uint32_t)wp;
    ra.alpha_pow = (const uint32_t*)v_apf; ra.dinv = dinv; ra.out = layers;
    void* v_at;
    ZK_TRY(ctx_reserve(ctx, S_PARTIAL, 2 * m * 16, &v_at));   // openings are done with S_PARTIAL by now
    ZK_HIP(launch_reduced_opening(ra, (uint32_t*)v_at, st));

    // ---- 5. FRI commit phase: RL committed layers (rows of 2^K adjacent entries): commit, challenge, fold K times
    std::vector<size_t> layer_off(RL + 1), tree_off(RL + 1);
    {
        size_t lo = 0, to = 0;
        for (int l = 0; l <= RL; l++) {
            layer_off[l] = lo; tree_off[l] = to;
            lo += ((size_t)1 << (H - K * l)) * 4;
            if (l < RL) to += (2 * ((size_t)1 << (H - K * (l + 1))) - 1) * 8;
        }
    }
    ZK_TRY(fri_commit_phase(ctx, ch, sh, H, RL, layers, ltrees, layer_off, tree_off, (uint32_t*)v_at, m, nullptr, pf, pos));
    // 2^(F+b) evaluations of a polynomial of < 2^F coefficients remain: interpolate on the host, send the coefficients
    {
        const int lf = sh.F + sh.b;
        const size_t nf = (size_t)1 << lf, keep = (size_t)1 << sh.F;
        std::vector<Ext> last(nf), nat(nf);
        ZK_TRY(d2h(ctx, last.data(), layers + layer_off[RL], nf * 16));
        for (size_t i = 0; i < nf; i++) nat[i] = last[reverse_bits((uint32_t)i, lf)];
        host_intt_ext(nat, lf);
        for (size_t i = keep; i < nf; i++)
            if (!ext_eq(nat[i], ext_zero()))
                return fail(ZKHIP_ERR_INVALID, "prove_shard: final FRI layer is not low-degree (the trace violates the AIR)");
        for (size_t i = 0; i < keep; i++) {
            for (int e = 0; e < 4; e++) pf[pos++] = from_monty(nat[i].c[e]);
            ch.observe_ext(nat[i]);
        }
    }

    // ---- 6. proof of work: smallest witness, searched 2^20 candidates per launch
    uint32_t witness = 0;
    ZK_TRY(grind_witness(ctx, ch, prm->pow_bits, &witness));
    pf[pos++] = ctx->debug.pow_witness = witness;

    // ---- 7. queries: one gather launch over (row, path, sibling) descriptors
    {
        std::vector<GatherDesc> descs;
        descs.reserve((size_t)Q * (4 + 4 * H + (size_t)RL * (H + arity)));
        size_t qpos = 0;   // word offset inside the query section
        auto push = [&](const uint32_t* src, size_t nwords) { descs.push_back(GatherDesc{src, (uint32_t)qpos, (uint32_t)nwords}); qpos += nwords; };
        auto push_path = [&](const uint32_t* tree, size_t leaves, size_t index, int levels) {
            const uint32_t* lvl = tree; size_t cnt = leaves, idx = index;
            for (int k = 0; k < levels; k++) { push(lvl + 8 * (idx ^ 1), 8); lvl += 8 * cnt; cnt >>= 1; idx >>= 1; }
        };
        for (int q = 0; q < Q; q++) {
            const size_t index = ch.sample_bits(H);
            push(tlde + index * width, width);
            if (CW) push_path(ctree, m, index, H);
            push_path(ttree, m, index, H);
            if (LQ) { push(plde + index * wp, wp); push_path(ptree, m, index, H); }
            push(qlde + index * QW, QW);
            push_path(qtree, m, index, H);
            size_t idx = index;
            for (int l = 0; l < RL; l++) {
                const int lh = H - K * (l + 1);
                const size_t row = idx >> K, own = idx & (arity - 1);
                for (size_t j = 0; j < arity; j++)
                    if (j != own) push(layers + layer_off[l] + (row * arity + j) * 4, 4);
                push_path(ltrees + tree_off[l], (size_t)1 << lh, row, lh);
                idx = row;
            }
        }
        if (pos + qpos != need / 4) return fail(ZKHIP_ERR_INTERNAL, "prove_shard: proof layout mismatch");
        void *v_desc, *v_out;
        ZK_TRY(ctx_reserve(ctx, S_GATHER_DESC, descs.size() * sizeof(GatherDesc), &v_desc));
        ZK_TRY(ctx_reserve(ctx, S_GATHER_OUT, qpos * 4, &v_out));
        ZK_TRY(h2d(ctx, v_desc, descs.data(), descs.size() * sizeof(GatherDesc)));
        ZK_HIP(launch_gather((const GatherDesc*)v_desc, (uint32_t)descs.size(), (uint32_t*)v_out, st));
        ZK_TRY(d2h(ctx, pf + pos, v_out, qpos * 4));
        pos += qpos;
    }
    for (int i = 0; i < 4; i++) {
        ctx->debug.alpha[i] = from_monty(alpha.c[i]);
        ctx->debug.zeta[i] = from_monty(zeta.c[i]);
        ctx->debug.fri_alpha[i] = from_monty(fa.c[i]);
    }
    *len = pos * 4;
    return ZKHIP_OK;
}

int zkhip_prove_shard(zkhip_ctx* ctx, const uint32_t* d_trace, size_t ld, int log_n, uint32_t width,
                      const uint32_t* public_values, size_t n_public, const zkhip_params* prm,
                      uint8_t* proof, size_t cap, size_t* len) {
    return prove_shard_impl(ctx, d_trace, ld, log_n, width, public_values, n_public, prm, proof, cap, len, nullptr);
}

// ---- the AIR as data: prove / verify against a constraint program (air.h)
size_t zkhip_proof_size_air(const uint32_t* program, size_t program_words, int log_n, uint32_t width, const zkhip_params* prm, size_t n_public) {
    AirView a;
    if (check_shape(log_n, width, prm) != ZKHIP_OK || prm->logup_pairs || !air_validate(program, program_words, width, n_public, &a) || a.lqd > prm->log_blowup) return 0;
    return proof_words(log_n, width, prm, true, a.lqd) * 4;
}
int zkhip_air_validate(const uint32_t* program, size_t words, uint32_t width, size_t n_public) {
    if (!air_validate(program, words, width, n_public, nullptr))
        return fail(ZKHIP_ERR_INVALID, "air_validate: malformed constraint program (header, selector, degree > 3, or a variable out of range)");
    return ZKHIP_OK;
}
int zkhip_air_digest(const uint32_t* program, size_t words, uint32_t out[8]) {
    AirView a;
    if (!out || !program || words < 6 || !air_validate(program, words, program[2], program[4], &a)) return fail(ZKHIP_ERR_INVALID, "air_digest: malformed constraint program");
    air_digest(a, out);
    return ZKHIP_OK;
}
// the synthetic AIR of DESIGN.md section 3 (without lookups) written as a program: what zkhip_prove_shard has built in
int zkhip_air_synthetic(uint32_t width, size_t n_public, uint32_t* out, size_t cap, size_t* words) {
    if (!words || width == 0 || width % 4 != 0 || width > 1024) return fail(ZKHIP_ERR_INVALID, "air_synthetic: width must be a positive multiple of 4, at most 1024");
    const size_t G = width / 4, need = 6 + G * 33;
    *words = need;
    if (!out) return ZKHIP_OK;                                // size query
    if (cap < need) return fail(ZKHIP_ERR_BUFFER, "air_synthetic: buffer too small");
    size_t p = 0;
    out[p++] = AIR_MAGIC; out[p++] = 1; out[p++] = width; out[p++] = (uint32_t)(3 * G); out[p++] = (uint32_t)n_public; out[p++] = (uint32_t)need;
    for (uint32_t g = 0; g < G; g++) {
        const uint32_t a = 4 * g, b = a + 1, c = a + 2, d = a + 3, NEXT = 1u << 30;
        out[p++] = 0; out[p++] = 3;                           // c - a a b - (g + 1) on every row
        out[p++] = 1; out[p++] = 1; out[p++] = c;
        out[p++] = P - 1; out[p++] = 3; out[p++] = a; out[p++] = a; out[p++] = b;
        out[p++] = P - (g + 1); out[p++] = 0;
        out[p++] = 3; out[p++] = 4;                           // d' - a b - c - (2g + 3) on transitions
        out[p++] = 1; out[p++] = 1; out[p++] = NEXT | d;
        out[p++] = P - 1; out[p++] = 2; out[p++] = a; out[p++] = b;
        out[p++] = P - 1; out[p++] = 1; out[p++] = c;
        out[p++] = P - (2 * g + 3); out[p++] = 0;
        out[p++] = 1; out[p++] = 2;                           // d - (5g + 7) on the first row
        out[p++] = 1; out[p++] = 1; out[p++] = d;
        out[p++] = P - (5 * g + 7); out[p++] = 0;
    }
    return p == need ? ZKHIP_OK : fail(ZKHIP_ERR_INTERNAL, "air_synthetic: layout mismatch");
}
int zkhip_prove_shard_air(zkhip_ctx* ctx, const uint32_t* program, size_t program_words, const uint32_t* d_trace, size_t ld, int log_n,
                          uint32_t width, const uint32_t* public_values, size_t n_public, const zkhip_params* prm,
                          uint8_t* proof, size_t cap, size_t* len) {
    AirView a;
    if (!air_validate(program, program_words, width, n_public, &a)) return fail(ZKHIP_ERR_INVALID, "prove_shard_air: malformed constraint program");
    return prove_shard_impl(ctx, d_trace, ld, log_n, width, public_values, n_public, prm, proof, cap, len, &a);
}
int zkhip_quotient_values_air(zkhip_ctx* ctx, const uint32_t* program, size_t program_words, const uint32_t* d_lde, size_t ld, int log_n,
                              uint32_t width, const uint32_t* public_values, size_t n_public, const uint32_t alpha[4], uint32_t* d_out) {
    CHECK_CTX(ctx);
    zkhip_params prm{1, 1, 0, 0, 0, 0, 0, 0};
    ZK_TRY(check_shape(log_n, width, &prm));
    AirView a;
    if (!d_lde || !d_out || !alpha || ld < width || (n_public && !public_values) || !air_validate(program, program_words, width, n_public, &a))
        return fail(ZKHIP_ERR_INVALID, "quotient_values_air: bad arguments or malformed program");
    const size_t n = (size_t)1 << log_n, NQ = (size_t)1 << a.lqd;
    void* chunks;
    ZK_TRY(ctx_reserve(ctx, S_QCHUNK, NQ * n * 16, &chunks));
    ZK_TRY(run_quotient_air(ctx, a, d_lde, ld, log_n, width, public_values, Ext{{alpha[0], alpha[1], alpha[2], alpha[3]}}, (uint32_t*)chunks, nullptr, 0));
    // natural chunk order -> the bit-reversed layout of the quotient domain (the first 2^lqd N rows of the LDE): point e = NQ j + k sits at
    // row bitrev(e) = bitrev_lqd(k) N + bitrev_n(j)
    std::vector<GatherDesc> descs(NQ * n);
    for (size_t k = 0; k < NQ; k++)
        for (size_t j = 0; j < n; j++) {
            const size_t p = (size_t)reverse_bits((uint32_t)k, a.lqd) * n + reverse_bits((uint32_t)j, log_n);
            descs[p] = GatherDesc{(const uint32_t*)chunks + (k * n + j) * 4, (uint32_t)(p * 4), 4};
        }
    void* dd;
    ZK_TRY(ctx_reserve(ctx, S_GATHER_DESC, descs.size() * sizeof(GatherDesc), &dd));
    ZK_TRY(h2d(ctx, dd, descs.data(), descs.size() * sizeof(GatherDesc)));
    ZK_HIP(launch_gather((const GatherDesc*)dd, (uint32_t)descs.size(), d_out, ctx->stream));
    ZK_HIP(launch_convert(d_out, d_out, NQ * n * 4, true, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_commit(zkhip_ctx* ctx, const uint32_t* d_trace, size_t ld, int log_n, uint32_t width, int log_blowup,
                 int hash_width, uint32_t* d_lde, uint32_t* d_tree, uint32_t root[8]) {
    CHECK_CTX(ctx);
    if (!d_trace || !d_lde || !d_tree || !root || ld < width || width == 0 || log_n < 5 || log_n > MAX_LOG_ROWS || log_blowup < 0 || log_blowup > 3 ||
        (hash_width != 16 && hash_width != 24) || (hash_width == 24 && width % 4 != 0))
        return fail(ZKHIP_ERR_INVALID, "commit: bad arguments (log_n in [5,22], log_blowup in [0,3], hash_width 16 or 24; width % 4 == 0 for 24)");
    const int H = log_n + log_blowup;
    ZK_TRY(op_coset_lde(ctx, d_trace, ld, d_lde, width, log_n, width, log_blowup, MONTY_GEN));
    ZK_TRY(commit_hw(ctx, d_lde, width, width, H, d_tree, hash_width));
    uint32_t r[8];
    ZK_TRY(d2h(ctx, r, d_tree + ((size_t)2 << H) * 8 - 16, 32));
    for (int i = 0; i < 8; i++) root[i] = from_monty(r[i]);
    return ZKHIP_OK;
}

int zkhip_prove_shard_host(zkhip_ctx* ctx, const uint32_t* h_trace, int log_n, uint32_t width,
                           const uint32_t* public_values, size_t n_public, const zkhip_params* prm,
                           uint8_t* proof, size_t cap, size_t* len) {
    CHECK_CTX(ctx);
    ZK_TRY(check_shape(log_n, width, prm));
    if (!h_trace) return fail(ZKHIP_ERR_INVALID, "prove_shard_host: null trace");
    const size_t words = (size_t)width << log_n;
    void* staged;
    ZK_TRY(ctx_reserve(ctx, S_STAGE, words * 4, &staged));
    ZK_HIP(hipMemcpyAsync(staged, h_trace, words * 4, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(launch_convert((const uint32_t*)staged, (uint32_t*)staged, words, true, ctx->stream));
    return zkhip_prove_shard(ctx, (const uint32_t*)staged, width, log_n, width, public_values, n_public, prm, proof, cap, len);
}

// ---- a batch of independent shards, `in_flight` at a time (one internal context + host thread each)
// Contexts (HIP stream + multi-GiB workspaces) are expensive to create and to free (hipFree synchronises the device), so the
// batch entry keeps the ones it made in a process-wide pool per device; zkhip_release_cached_contexts() empties it.
namespace {
std::mutex g_pool_mu;
std::vector<std::pair<int, zkhip_ctx*>> g_pool;
zkhip_ctx* pool_take(int device) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++)
        if (g_pool[i].first == device) { zkhip_ctx* c = g_pool[i].second; g_pool.erase(g_pool.begin() + (long)i); return c; }
    return nullptr;
}
void pool_give(int device, zkhip_ctx* c) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.emplace_back(device, c);
}
}  // namespace
void zkhip_release_cached_contexts(void) {
    std::vector<std::pair<int, zkhip_ctx*>> all;
    { std::lock_guard<std::mutex> lk(g_pool_mu); all.swap(g_pool); }
    for (auto& e : all) zkhip_ctx_destroy(e.second);
}
// Job i of a batch goes to devices[i mod n_devices] (shard-parallel, no exchange step); every device runs up to
// `in_flight` workers (a pooled context + HIP stream + host thread each) that take that device's jobs in index order and call
// run(ctx, i) -> status.  Returns the status of the lowest failing job (its message in zkhip_last_error), and tells through `ran`
// which jobs a worker reached at all (none when every worker of a device failed to get a context).
extern "C++" {
namespace zk {
int deal_jobs(const int* devices, int n_devices, int n_jobs, int in_flight, const std::function<int(zkhip_ctx*, int)>& run, std::vector<char>& ran) {
    ran.assign((size_t)n_jobs, 0);
    if (n_jobs == 0) return ZKHIP_OK;
    if (in_flight <= 0) in_flight = 4;
    std::vector<std::atomic<int>> next(n_devices);            // per device: how many of ITS jobs were handed out
    for (auto& a : next) a.store(0);
    std::mutex mu;
    int first_rc = ZKHIP_OK, first_job = n_jobs, ctx_rc = ZKHIP_OK;
    std::string first_msg, ctx_msg;
    auto note = [&](int job, int rc) {                     // keep the failure of the lowest job index
        std::lock_guard<std::mutex> lk(mu);
        if (job < first_job) { first_job = job; first_rc = rc; first_msg = zkhip_last_error(); }
    };
    auto worker = [&](int slot) {
        const int device = devices[slot];
        zkhip_ctx* ctx = pool_take(device);
        int rc = ctx ? ZKHIP_OK : zkhip_ctx_create(device, nullptr, &ctx);
        if (rc != ZKHIP_OK) {                                // e.g. no memory for one more workspace: the other workers carry on
            std::lock_guard<std::mutex> lk(mu);
            ctx_rc = rc; ctx_msg = zkhip_last_error();
            return;
        }
        bool healthy = true;                                 // a context that saw a failing job does not go back to the pool:
        for (;;) {                                           // a sticky HIP error or a half-built key would fail unrelated jobs later
            const int k = next[slot].fetch_add(1);
            const long i = (long)slot + (long)k * n_devices;   // the k-th job of this device
            if (i >= n_jobs) break;
            rc = run(ctx, (int)i);
            ran[(size_t)i] = 1;
            if (rc != ZKHIP_OK) { note((int)i, rc); healthy = false; }
        }
        if (zkhip_ctx_sync(ctx) != ZKHIP_OK) healthy = false;
        if (healthy) pool_give(device, ctx);
        else zkhip_ctx_destroy(ctx);
    };
    std::vector<std::thread> pool;
    for (int slot = 0; slot < n_devices; slot++) {
        const int mine = (n_jobs - slot + n_devices - 1) / n_devices;       // jobs of this device
        const int workers = mine < in_flight ? mine : in_flight;
        for (int t = 0; t < workers; t++) {
            // thread creation can throw (resource limits): never let that unwind through joinable threads into the C ABI --
            // the workers already started (or, with none, this thread) take the jobs instead
            try { pool.emplace_back(worker, slot); }
            catch (...) { if (t == 0) worker(slot); break; }
        }
    }
    for (auto& t : pool) t.join();
    if (first_rc != ZKHIP_OK) { set_error(first_msg); return first_rc; }
    for (int i = 0; i < n_jobs; i++)                         // jobs nobody could take: every worker of that device failed to get a context
        if (!ran[(size_t)i]) { set_error(ctx_msg.empty() ? "prove_shards: job not run" : ctx_msg); return ctx_rc != ZKHIP_OK ? ctx_rc : ZKHIP_ERR_INVALID; }
    return ZKHIP_OK;
}
// lock-step batches (batch.h): members per batch (0 / 1: off) and batches in flight per device; process-wide
static std::atomic<int> g_lockstep_batch{16}, g_lockstep_lanes{6};
int lockstep_batch() { return g_lockstep_batch.load(); }
int lockstep_lanes() { return g_lockstep_lanes.load(); }
void lockstep_set(int max_batch, int lanes) {
    g_lockstep_batch.store(max_batch < 0 ? 0 : (max_batch > LaunchBatcher::MAX_MEMBERS ? LaunchBatcher::MAX_MEMBERS : max_batch));
    if (lanes > 0) g_lockstep_lanes.store(lanes > 32 ? 32 : lanes);
}
struct HostPool::Impl {
    std::mutex mu;
    std::condition_variable cv, idle;
    std::deque<std::function<void()>> queue;
    std::vector<std::thread> threads;
    int running = 0;
    bool stop = false;
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return stop || !queue.empty(); });
            if (queue.empty()) return;
            std::function<void()> job = std::move(queue.front());
            queue.pop_front();
            running++;
            lk.unlock();
            job();
            lk.lock();
            running--;
            if (queue.empty() && running == 0) idle.notify_all();
        }
    }
};
HostPool::HostPool(int threads) : impl_(new Impl) {
    for (int t = 0; t < threads; t++) {
        try { impl_->threads.emplace_back([this] { impl_->loop(); }); } catch (...) { break; }
    }
}
HostPool::~HostPool() {
    wait();
    { std::lock_guard<std::mutex> lk(impl_->mu); impl_->stop = true; }
    impl_->cv.notify_all();
    for (auto& t : impl_->threads) t.join();
    delete impl_;
}
void HostPool::submit(std::function<void()> job) {
    if (impl_->threads.empty()) { job(); return; }              // no thread could be made: the caller does it
    { std::lock_guard<std::mutex> lk(impl_->mu); impl_->queue.push_back(std::move(job)); }
    impl_->cv.notify_one();
}
void HostPool::wait() {
    std::unique_lock<std::mutex> lk(impl_->mu);
    impl_->idle.wait(lk, [&] { return impl_->queue.empty() && impl_->running == 0; });
}

// Lock-step variant for SMALL proofs (batch.h): the jobs of a device are grouped by `shape[i]` (jobs of one shape run the same launch
// sequence), groups are cut into batches of up to `max_batch`, and a batch is proven by that many provers on pooled contexts that
// share ONE stream and merge their kernel launches through a LaunchBatcher.  `lanes` batches are in flight per device (a host thread
// each; the members of a batch are fibers of that thread), so that the host-side work around one batch (padding, transcripts,
// verification) overlaps the other batches' kernels.  Same contract as deal_jobs.
int deal_jobs_lockstep(const int* devices, int n_devices, int n_jobs, const int* shape, int max_batch, int lanes,
                       const std::function<int(zkhip_ctx*, int)>& run, std::vector<char>& ran) {
    ran.assign((size_t)n_jobs, 0);
    if (n_jobs == 0) return ZKHIP_OK;
    if (max_batch < 1) max_batch = 1;
    if (max_batch > LaunchBatcher::MAX_MEMBERS) max_batch = LaunchBatcher::MAX_MEMBERS;
    if (lanes < 1) lanes = 1;
    std::mutex mu;
    int first_rc = ZKHIP_OK, first_job = n_jobs, ctx_rc = ZKHIP_OK;
    std::string first_msg, ctx_msg;
    auto note = [&](int job, int rc, const std::string& msg) {
        std::lock_guard<std::mutex> lk(mu);
        if (job < first_job) { first_job = job; first_rc = rc; first_msg = msg; }
    };
    // batches per device, in job order
    std::vector<std::vector<std::vector<int>>> batches((size_t)n_devices);
    for (int d = 0; d < n_devices; d++) {
        std::vector<std::pair<int, std::vector<int>>> groups;
        for (int i = d; i < n_jobs; i += n_devices) {
            size_t g = 0;
            while (g < groups.size() && groups[g].first != shape[i]) g++;
            if (g == groups.size()) groups.emplace_back(shape[i], std::vector<int>());
            groups[g].second.push_back(i);
        }
        for (auto& g : groups) {
            // equal cuts: sixty-four jobs on two lanes are two batches of thirty-two, not one of max_batch and a remainder
            const size_t n = g.second.size();
            size_t cuts = (n + (size_t)max_batch - 1) / (size_t)max_batch;
            if (cuts < (size_t)lanes && n >= 2 * (size_t)lanes) cuts = (size_t)lanes;
            for (size_t c = 0; c < cuts; c++) {
                const size_t lo = n * c / cuts, hi = n * (c + 1) / cuts;
                if (hi > lo) batches[(size_t)d].emplace_back(g.second.begin() + (long)lo, g.second.begin() + (long)hi);
            }
        }
    }
    std::vector<std::atomic<int>> next((size_t)n_devices);
    for (auto& a : next) a.store(0);
    auto lane = [&](int slot) {
        const int device = devices[slot];
        for (;;) {
            const int k = next[(size_t)slot].fetch_add(1);
            if (k >= (int)batches[(size_t)slot].size()) break;
            const std::vector<int>& jobs = batches[(size_t)slot][(size_t)k];
            const int B = (int)jobs.size();
            std::vector<zkhip_ctx*> ctxs;
            for (int b = 0; b < B; b++) {
                zkhip_ctx* c = pool_take(device);
                if (!c && zkhip_ctx_create(device, nullptr, &c) != ZKHIP_OK) {
                    std::lock_guard<std::mutex> lk(mu);
                    ctx_rc = ZKHIP_ERR_HIP; ctx_msg = zkhip_last_error();
                    break;
                }
                ctxs.push_back(c);
            }
            if (ctxs.empty()) continue;                          // no context at all: the jobs stay unrun
            // fewer contexts than jobs (memory): the batch runs in rounds of ctxs.size()
            const int W = (int)ctxs.size();
            std::vector<hipStream_t> own((size_t)W);
            for (int b = 0; b < W; b++) own[(size_t)b] = ctxs[(size_t)b]->stream;
            const hipStream_t shared = own[0];
            std::vector<char> healthy((size_t)W, 1);
            for (int at = 0; at < B; at += W) {
                const int n = B - at < W ? B - at : W;
                if (n == 1) {                                    // nothing to merge with
                    const int i = jobs[(size_t)at];
                    const int rc = run(ctxs[0], i);
                    ran[(size_t)i] = 1;
                    if (rc != ZKHIP_OK) { note(i, rc, zkhip_last_error()); healthy[0] = 0; }
                    continue;
                }
                for (int b = 0; b < n; b++) ctxs[(size_t)b]->stream = shared;
                {
                    LaunchBatcher lb(n, shared);
                    auto member = [&](int b) {
                        const int i = jobs[(size_t)(at + b)];
                        const int rc = run(ctxs[(size_t)b], i);
                        ran[(size_t)i] = 1;
                        if (rc != ZKHIP_OK) { note(i, rc, zkhip_last_error()); healthy[(size_t)b] = 0; }
                    };
                    if (lb.ok()) lb.run(member);                 // the members as fibers of this thread, their launches merged
                    else for (int b = 0; b < n; b++) member(b);  // (no pinned memory / stacks: one after the other, unmerged)
                    if (hipStreamSynchronize(shared) != hipSuccess) { (void)hipGetLastError(); for (int b = 0; b < n; b++) healthy[(size_t)b] = 0; }
                    if (lb.failed()) {                            // a merged launch failed: every proof of the batch is suspect
                        for (int b = 0; b < n; b++) {
                            healthy[(size_t)b] = 0;
                            note(jobs[(size_t)(at + b)], ZKHIP_ERR_HIP, "lock-step batch: a merged kernel launch failed");
                        }
                    }
                }
                for (int b = 0; b < n; b++) ctxs[(size_t)b]->stream = own[(size_t)b];
            }
            for (int b = 0; b < W; b++) {
                if (healthy[(size_t)b] && zkhip_ctx_sync(ctxs[(size_t)b]) == ZKHIP_OK) pool_give(device, ctxs[(size_t)b]);
                else zkhip_ctx_destroy(ctxs[(size_t)b]);
            }
        }
    };
    std::vector<std::thread> pool;
    for (int slot = 0; slot < n_devices; slot++) {
        const int nb = (int)batches[(size_t)slot].size();
        const int workers = nb < lanes ? nb : lanes;
        for (int t = 0; t < workers; t++) {
            try { pool.emplace_back(lane, slot); }
            catch (...) { if (t == 0) lane(slot); break; }
        }
    }
    for (auto& t : pool) t.join();
    if (first_rc != ZKHIP_OK) { set_error(first_msg); return first_rc; }
    for (int i = 0; i < n_jobs; i++)
        if (!ran[(size_t)i]) { set_error(ctx_msg.empty() ? "prove (lock-step): job not run" : ctx_msg); return ctx_rc != ZKHIP_OK ? ctx_rc : ZKHIP_ERR_INVALID; }
    return ZKHIP_OK;
}
// the device list of a batch entry: NULL (with n_devices == 0) = every visible device; ordinals non-negative and distinct.
// ZKHIP_ERR_NO_DEVICE when nothing is visible (the caller marks its jobs)
int resolve_devices(const int* devices, int n_devices, const char* what, std::vector<int>& devs) {
    devs.clear();
    if (!devices) {
        if (n_devices != 0) return fail(ZKHIP_ERR_INVALID, std::string(what) + ": n_devices must be 0 when devices is NULL (all visible devices)");
        const int n = zkhip_device_count();
        if (n <= 0) return fail(ZKHIP_ERR_NO_DEVICE, "no HIP device visible: libzkhip has no CPU fallback");
        for (int d = 0; d < n; d++) devs.push_back(d);
        return ZKHIP_OK;
    }
    if (n_devices < 1 || n_devices > 64) return fail(ZKHIP_ERR_INVALID, std::string(what) + ": 1..64 devices");
    for (int d = 0; d < n_devices; d++) {
        if (devices[d] < 0) return fail(ZKHIP_ERR_INVALID, std::string(what) + ": negative device ordinal");
        for (int e = 0; e < d; e++) if (devices[e] == devices[d]) return fail(ZKHIP_ERR_INVALID, std::string(what) + ": device listed twice");
        devs.push_back(devices[d]);
    }
    return ZKHIP_OK;
}
}  // namespace zk
}  // extern "C++"

// Job traces are device pointers ON THE DEVICE THE SHARD IS ASSIGNED TO, or host pointers with host_traces.
static int prove_shards_on(const int* devices, int n_devices, zkhip_shard_job* jobs, int n_jobs, const zkhip_params* prm, int in_flight,
                           int host_traces, const uint32_t* program = nullptr, size_t program_words = 0) {
    for (int i = 0; i < n_jobs; i++) { jobs[i].status = ZKHIP_ERR_INVALID; jobs[i].proof_len = 0; }
    std::vector<char> ran;
    auto run = [&](zkhip_ctx* ctx, int i) {
        zkhip_shard_job& j = jobs[i];
        size_t len = 0;
        int rc;
        if (program)             // every job of the batch is a trace of the same constraint program (device traces)
            rc = zkhip_prove_shard_air(ctx, program, program_words, j.trace, j.ld, j.log_n, j.width, j.public_values, j.n_public, prm, j.proof, j.proof_cap, &len);
        else
            rc = host_traces
                     ? zkhip_prove_shard_host(ctx, j.trace, j.log_n, j.width, j.public_values, j.n_public, prm, j.proof, j.proof_cap, &len)
                     : zkhip_prove_shard(ctx, j.trace, j.ld, j.log_n, j.width, j.public_values, j.n_public, prm, j.proof, j.proof_cap, &len);
        j.status = rc;
        j.proof_len = rc == ZKHIP_OK ? len : 0;
        return rc;
    };
    // a batch of SMALL shards is launch-bound: lock-step lanes (batch.h) when every job is small, one context and stream each otherwise
    const int max_batch = lockstep_batch();
    bool small = max_batch > 1 && n_jobs >= 2 * n_devices;
    std::vector<int> shape((size_t)n_jobs);
    for (int i = 0; i < n_jobs && small; i++) {
        const zkhip_shard_job& j = jobs[i];
        if (j.log_n < 1 || j.log_n > 24 || j.width == 0 || ((uint64_t)j.width << j.log_n) > LOCKSTEP_MAX_CELLS) small = false;
        shape[(size_t)i] = (int)(((uint32_t)j.log_n << 24) ^ j.width);
    }
    if (small) return deal_jobs_lockstep(devices, n_devices, n_jobs, shape.data(), max_batch, lockstep_lanes(), run, ran);
    return deal_jobs(devices, n_devices, n_jobs, in_flight, run, ran);
}

int zkhip_prove_shards(int device, zkhip_shard_job* jobs, int n_jobs, const zkhip_params* prm, int in_flight, int host_traces) {
    if (!jobs || n_jobs < 0 || !prm) return fail(ZKHIP_ERR_INVALID, "prove_shards: bad arguments");
    return prove_shards_on(&device, 1, jobs, n_jobs, prm, in_flight, host_traces);
}

int zkhip_shard_device(int shard_index, const int* devices, int n_devices) {
    if (shard_index < 0 || n_devices < 1) return -1;
    return devices ? devices[shard_index % n_devices] : shard_index % n_devices;
}

int zkhip_prove_shards_multi(const int* devices, int n_devices, zkhip_shard_job* jobs, int n_jobs, const zkhip_params* prm,
                             int in_flight_per_device, int host_traces) {
    if (!jobs || n_jobs < 0 || !prm) return fail(ZKHIP_ERR_INVALID, "prove_shards_multi: bad arguments");
    std::vector<int> devs;
    const int rc = resolve_devices(devices, n_devices, "prove_shards_multi", devs);
    if (rc == ZKHIP_ERR_NO_DEVICE) {
        for (int i = 0; i < n_jobs; i++) { jobs[i].status = ZKHIP_ERR_NO_DEVICE; jobs[i].proof_len = 0; }
        return n_jobs == 0 ? ZKHIP_OK : rc;
    }
    if (rc != ZKHIP_OK) return rc;
    return prove_shards_on(devs.data(), (int)devs.size(), jobs, n_jobs, prm, in_flight_per_device, host_traces);
}

// the same batch when every job is a trace of ONE constraint program (e.g. sixty-four SHA-256 chip traces: sixty-four transcripts)
int zkhip_prove_shards_air_multi(const int* devices, int n_devices, zkhip_shard_job* jobs, int n_jobs, const uint32_t* program, size_t program_words,
                                 const zkhip_params* prm, int in_flight_per_device) {
    if (!program || program_words < 6) return fail(ZKHIP_ERR_INVALID, "prove_shards_air_multi: null program");
    if (!jobs || n_jobs < 0 || !prm) return fail(ZKHIP_ERR_INVALID, "prove_shards_air_multi: bad arguments");
    std::vector<int> devs;
    const int rc = resolve_devices(devices, n_devices, "prove_shards_air_multi", devs);
    if (rc == ZKHIP_ERR_NO_DEVICE) {
        for (int i = 0; i < n_jobs; i++) { jobs[i].status = ZKHIP_ERR_NO_DEVICE; jobs[i].proof_len = 0; }
        return n_jobs == 0 ? ZKHIP_OK : rc;
    }
    if (rc != ZKHIP_OK) return rc;
    return prove_shards_on(devs.data(), (int)devs.size(), jobs, n_jobs, prm, in_flight_per_device, 0, program, program_words);
}

int zkhip_prove_segment(zkhip_ctx* ctx, const uint32_t* d_cols, int log_n, uint32_t width,
                        const uint32_t* public_values, size_t n_public, const zkhip_params* prm,
                        uint8_t* proof, size_t cap, size_t* len) {
    CHECK_CTX(ctx);
    ZK_TRY(check_shape(log_n, width, prm));
    if (!d_cols) return fail(ZKHIP_ERR_INVALID, "prove_segment: null trace");
    // the prover's kernels stream row-major rows (leaf hashing, quotient, openings all want whole rows):
    // one tiled transpose in HBM, then the common path
    const uint64_t n = (uint64_t)1 << log_n;
    void* rows;
    ZK_TRY(ctx_reserve(ctx, S_STAGE, n * width * 4, &rows));
    ZK_HIP(launch_transpose(d_cols, (uint32_t*)rows, width, n, 0, 0, ctx->stream));
    return zkhip_prove_shard(ctx, (const uint32_t*)rows, width, log_n, width, public_values, n_public, prm, proof, cap, len);
}

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
struct FriViewSink { uint32_t *betas, *final_value, *indices, *values, *siblings; int layers; uint32_t *roots, *paths; };      // roots / paths optional

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
    for (int l = 0; l < RL; l++) {
        for (int i = 0; i < 8; i++) { commits[8 * l + i] = to_monty(pf[pos++]); ch.observe(commits[8 * l + i]); }
        betas[l] = ch.sample_ext();
    }
    const size_t keep = (size_t)1 << sh.F;
    std::vector<Ext> final_poly(keep);            // coefficients, lowest first
    for (size_t i = 0; i < keep; i++) { final_poly[i] = ext_from_canon(pf + pos); pos += 4; ch.observe_ext(final_poly[i]); }
    const uint32_t witness = pf[pos++];
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
        if (CW) mark(verify_paths_x16(croot, H, batch(trow, 0, cpath, index), CW, sh.hw), 33);
        mark(verify_paths_x16(troot, H, batch(trow, CW, tpath, index), width - CW, sh.hw), 30);
        if (LQ) mark(verify_paths_x16(proot, H, batch(prow, 0, ppath, index), wp, sh.hw), 32);
        mark(verify_paths_x16(qroot, H, batch(qrow, 0, qpath, index), QW, sh.hw), 31);
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
            mark(verify_paths_x16(&commits[8 * l], lh, batch(rows, 0, paths, rowidx), 4 * arity, sh.hw), 40 + (l < 50 ? l : 50));
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
    FriViewSink sink{betas, final_value, indices, values, siblings, sh.R, nullptr, nullptr};
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
    FriViewSink sink{betas, final_value, indices, values, siblings, sh.R, roots, paths};
    int why = 0;
    return verify_shard_impl(proof, len, log_n, width, public_values, n_public, prm, &why, nullptr, &sink);
}

// ================================================================ shards of several chips with different heights
// The structure of an SP1 shard (sp1-stark 4.1.4 ShardProof, reference Cargo.lock:6172, behind sp1.rs:116): one Merkle
// commitment per phase over matrices of different heights (p3-merkle-tree injection rule), one zeta, one reduced-opening
// vector per height that joins the FRI vector when folding reaches that height (p3-fri 0.2.1 TwoAdicFriPcs), one query
// index with chip c opened at index >> (Hmax - h_c).  Byte layout: DESIGN.md section 6.
namespace zk {
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
}  // namespace zk

static size_t chips_proof_size_impl(const ChipSet& cs, const int32_t* log_ns, const uint32_t* widths, const int32_t* pairs, const int32_t* partners, int n_chips,
                                    const zkhip_params* prm, size_t n_public) {
    (void)n_public;
    if (check_chips(cs, log_ns, widths, pairs, partners, n_chips, prm) != ZKHIP_OK) return 0;
    return chips_proof_words(cs, log_ns, widths, pairs, partners, n_chips, prm) * 4;
}

size_t zkhip_chips_proof_size(const int32_t* log_ns, const uint32_t* widths, const int32_t* pairs, const int32_t* partners, int n_chips,
                              const zkhip_params* prm, size_t n_public) {
    return chips_proof_size_impl(ChipSet{}, log_ns, widths, pairs, partners, n_chips, prm, n_public);
}

static int prove_chips_impl(const ChipSet& cs, zkhip_ctx* ctx, const zkhip_chip* chips, int n, const uint32_t* public_values, size_t n_public,
                            const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len) {
    CHECK_CTX(ctx);
    if (!chips || !proof || !len || (n_public && !public_values)) return fail(ZKHIP_ERR_INVALID, "prove_chips: bad arguments");
    int32_t log_ns[MAX_CHIPS], pairs[MAX_CHIPS], partners[MAX_CHIPS]; uint32_t widths[MAX_CHIPS];
    if (n < 1 || n > MAX_CHIPS) return fail(ZKHIP_ERR_INVALID, "chips: 1..32 chips");
    for (int c = 0; c < n; c++) {
        log_ns[c] = chips[c].log_n; widths[c] = chips[c].width; pairs[c] = chips[c].logup_pairs; partners[c] = chips[c].partner;
        if (!chips[c].d_trace || chips[c].ld < chips[c].width) return fail(ZKHIP_ERR_INVALID, "prove_chips: bad chip descriptor");
    }
    ZK_TRY(check_chips(cs, log_ns, widths, pairs, partners, n, prm));
    for (size_t i = 0; i < n_public; i++) if (public_values[i] >= P) return fail(ZKHIP_ERR_INVALID, "prove_chips: public values must be canonical");
    const size_t need = chips_proof_words(cs, log_ns, widths, pairs, partners, n, prm) * 4;
    const bool lk = any_pairs(pairs, n), cross = any_cross(cs, partners, n);
    Ext cumsum[MAX_CHIPS];
    for (int c = 0; c < n; c++) cumsum[c] = ext_zero();
    if (cap < need) return fail(ZKHIP_ERR_BUFFER, "prove_chips: proof buffer too small (see zkhip_chips_proof_size)");
    *len = 0;
    hipStream_t st = ctx->stream;
    const int b = prm->log_blowup, Hmax = log_ns[0] + b, L = log_ns[0], Q = prm->num_queries;
    const size_t mmax = (size_t)1 << Hmax;
    Shape sh;                                     // SP1 FRI shape at this blowup
    sh.b = b; sh.R = L;
    int lh[MAX_CHIPS];
    size_t tl_off[MAX_CHIPS + 1], ql_off[MAX_CHIPS + 1], dv_off[MAX_CHIPS + 1], op_off[MAX_CHIPS + 1], ap_off[MAX_CHIPS + 1], pl_off[MAX_CHIPS + 1], wp[MAX_CHIPS];
    tl_off[0] = ql_off[0] = dv_off[0] = op_off[0] = ap_off[0] = pl_off[0] = 0;
    size_t nmax_chunk = 0, perm_max = 0;
    int Hp = 0;
    for (int c = 0; c < n; c++) {
        lh[c] = log_ns[c] + b;
        const size_t mc = (size_t)1 << lh[c], nc = (size_t)1 << log_ns[c];
        tl_off[c + 1] = tl_off[c] + mc * (pre_w(cs, c) + widths[c]);     // a keyed chip's LDE rows are [preprocessed | main]
        ql_off[c + 1] = ql_off[c] + mc * qw_of(cs, c);
        dv_off[c + 1] = dv_off[c] + 8 * (mc + nc);               // [2][mc] 1/(x - z) then [2][nc] x/(x - z), ext words
        wp[c] = perm_width(pairs, c);
        pl_off[c + 1] = pl_off[c] + mc * wp[c];
        op_off[c + 1] = op_off[c] + 8 * (size_t)pre_w(cs, c) + 8 * (size_t)widths[c] + 8 * wp[c] + 4 * qw_of(cs, c);
        size_t npw = widths[c] > qw_of(cs, c) ? widths[c] : qw_of(cs, c);
        if (wp[c] > npw) npw = wp[c];
        if (pre_w(cs, c) > npw) npw = pre_w(cs, c);
        ap_off[c + 1] = ap_off[c] + 4 * npw;
        if ((nc << lq_of(cs, c)) > nmax_chunk) nmax_chunk = nc << lq_of(cs, c);       // quotient values of the chip: 2^lq chunks of nc points
        if (nc * wp[c] > perm_max) perm_max = nc * wp[c];
        if (wp[c] && lh[c] > Hp) Hp = lh[c];
    }
    uint32_t* pf = (uint32_t*)proof;
    size_t pos = 0;
    pf[pos++] = PROOF_MAGIC; pf[pos++] = chips_version(cs, pairs, partners, n); pf[pos++] = (uint32_t)n; pf[pos++] = (uint32_t)b;
    pf[pos++] = (uint32_t)Q; pf[pos++] = (uint32_t)prm->pow_bits; pf[pos++] = (uint32_t)n_public; pf[pos++] = 16u;
    for (int c = 0; c < n; c++) {
        pf[pos++] = (uint32_t)log_ns[c]; pf[pos++] = widths[c];
        if (cs.machine) { pf[pos++] = header_prog_word(cs, c); pf[pos++] = lookup_of(cs, c) ? lookup_of(cs, c)->ni : 0u; if (cs.key) pf[pos++] = pre_w(cs, c); continue; }
        if (lk) pf[pos++] = (uint32_t)pairs[c];
        if (cross) pf[pos++] = (uint32_t)(partners[c] + 1);
        if (any_prog(cs, n)) pf[pos++] = header_prog_word(cs, c);
    }
    for (int c = 0; c < n; c++) if (header_has_prog(cs, c)) { air_digest_cached(*prog_of(cs, c), pf + pos); pos += 8; }
    for (int c = 0; c < n; c++) if (lookup_of(cs, c)) { lookup_digest(*lookup_of(cs, c), pf + pos); pos += 8; }
    if (cs.key) for (int i = 0; i < 8; i++) pf[pos++] = from_monty(cs.key->root_m[i]);
    Challenger ch;
    chips_transcript_init(cs, ch, log_ns, widths, pairs, partners, n, prm, n_public);
    uint32_t root[8];

    // ---- 1. every chip's LDE, one mixed-height tree
    void *v_tlde, *v_ttree, *v_qlde, *v_qtree, *v_qchunk;
    ZK_TRY(ctx_reserve(ctx, S_TLDE, tl_off[n] * 4, &v_tlde));
    ZK_TRY(ctx_reserve(ctx, S_TTREE, (2 * mmax - 1) * 32, &v_ttree));
    ZK_TRY(ctx_reserve(ctx, S_QLDE, ql_off[n] * 4, &v_qlde));
    ZK_TRY(ctx_reserve(ctx, S_QTREE, (2 * mmax - 1) * 32, &v_qtree));
    ZK_TRY(ctx_reserve(ctx, S_QCHUNK, nmax_chunk * 16, &v_qchunk));
    uint32_t *tlde = (uint32_t*)v_tlde, *ttree = (uint32_t*)v_ttree, *qlde = (uint32_t*)v_qlde, *qtree = (uint32_t*)v_qtree, *qchunk = (uint32_t*)v_qchunk;
    MatDesc tm[MAX_CHIPS], qm[MAX_CHIPS];
    size_t cw[MAX_CHIPS];                          // row pitch of a chip's LDE: preprocessed + main columns
    for (int c = 0; c < n; c++) {
        const uint32_t pw = pre_w(cs, c);
        cw[c] = (size_t)pw + widths[c];
        if (pw)                                    // the key's LDE columns beside the main ones: what the program and the interactions read
            ZK_HIP(launch_copy2d(tlde + tl_off[c], cw[c], cs.key->d_lde[c], pw, (uint32_t)pw, (uint64_t)1 << lh[c], st));
        ZK_TRY(op_coset_lde(ctx, chips[c].d_trace, chips[c].ld, tlde + tl_off[c] + pw, cw[c], log_ns[c], widths[c], b, MONTY_GEN));
        tm[c] = MatDesc{tlde + tl_off[c] + pw, cw[c], widths[c]};
        qm[c] = MatDesc{qlde + ql_off[c], qw_of(cs, c), (uint32_t)qw_of(cs, c)};
    }
    ZK_TRY(op_merkle_commit_mixed(ctx, tm, lh, n, ttree));
    ZK_TRY(d2h(ctx, root, ttree + (2 * mmax - 2) * 8, 32));
    for (int i = 0; i < 8; i++) { ch.observe(root[i]); pf[pos++] = from_monty(root[i]); }
    for (size_t i = 0; i < n_public; i++) ch.observe_canonical(public_values[i]);

    // ---- 1b. lookups: one (gamma, beta) for the shard; the permutation traces of the chips that have pairs -> third tree
    Ext gamma = ext_zero(), beta_l = ext_zero();
    uint32_t *plde = nullptr, *ptree = nullptr;
    if (lk) {
        gamma = ch.sample_ext();
        beta_l = ch.sample_ext();
        void *v_perm, *v_plde, *v_ptree;
        ZK_TRY(ctx_reserve(ctx, S_PERM, perm_max * 4, &v_perm));
        ZK_TRY(ctx_reserve(ctx, S_PLDE, pl_off[n] * 4, &v_plde));
        ZK_TRY(ctx_reserve(ctx, S_PTREE, (2 * ((size_t)1 << Hp) - 1) * 32, &v_ptree));
        plde = (uint32_t*)v_plde; ptree = (uint32_t*)v_ptree;
        MatDesc pmats[MAX_CHIPS]; int plh[MAX_CHIPS]; int np = 0;
        for (int c = 0; c < n; c++) {
            if (!wp[c]) continue;
            if (cs.machine && pre_w(cs, c)) {           // the interactions address [preprocessed | main] rows of the trace domain
                const size_t nc = (size_t)1 << log_ns[c], pw = pre_w(cs, c);
                void* v_ct;
                ZK_TRY(ctx_reserve(ctx, S_KEYTRACE, nc * cw[c] * 4, &v_ct));
                uint32_t* ct = (uint32_t*)v_ct;
                ZK_HIP(launch_copy2d(ct, cw[c], cs.key->d_trace[c], pw, (uint32_t)pw, nc, st));
                ZK_HIP(launch_copy2d(ct + pw, cw[c], chips[c].d_trace, chips[c].ld, widths[c], nc, st));
                ZK_TRY(run_lookup_perm(ctx, ct, cw[c], log_ns[c], *lookup_of(cs, c), gamma, beta_l, (uint32_t*)v_perm));
            } else if (cs.machine) ZK_TRY(run_lookup_perm(ctx, chips[c].d_trace, chips[c].ld, log_ns[c], *lookup_of(cs, c), gamma, beta_l, (uint32_t*)v_perm));
            else ZK_TRY(run_perm_trace(ctx, chips[c].d_trace, chips[c].ld, log_ns[c], (uint32_t)pairs[c], gamma, beta_l, (uint32_t*)v_perm));
            if (cross)          // the running sum's last value: row N - 1, column S
                ZK_TRY(d2h(ctx, &cumsum[c], (const uint32_t*)v_perm + (((size_t)1 << log_ns[c]) - 1) * wp[c] + 4 * (size_t)pairs[c], 16));
            ZK_TRY(op_coset_lde(ctx, (const uint32_t*)v_perm, wp[c], plde + pl_off[c], wp[c], log_ns[c], (uint32_t)wp[c], b, MONTY_GEN));
            pmats[np] = MatDesc{plde + pl_off[c], wp[c], (uint32_t)wp[c]}; plh[np] = lh[c]; np++;
        }
        ZK_TRY(op_merkle_commit_mixed(ctx, pmats, plh, np, ptree));
        ZK_TRY(d2h(ctx, root, ptree + (2 * ((size_t)1 << Hp) - 2) * 8, 32));
        for (int i = 0; i < 8; i++) { ch.observe(root[i]); pf[pos++] = from_monty(root[i]); }
        if (cross)
            for (int c = 0; c < n; c++)
                if (wp[c]) { ch.observe_ext(cumsum[c]); for (int e = 0; e < 4; e++) pf[pos++] = from_monty(cumsum[c].c[e]); }
    }

    // ---- 2. per-chip quotients on the chip's own coset, chunk LDEs, quotient tree
    const Ext alpha = ch.sample_ext();
    for (int c = 0; c < n; c++) {
        const size_t nc = (size_t)1 << log_ns[c];
        ZK_TRY(ensure_domain(ctx, log_ns[c], b));
        LogupIn lu;
        if (wp[c]) { lu.pairs = (uint32_t)pairs[c]; lu.perm_lde = plde + pl_off[c]; lu.gamma = gamma; lu.beta = beta_l; lu.cumsum = cumsum[c]; }
        const bool own_coset_direct = b == 1;             // as in the single-matrix prover
        if (cs.machine && !lookup_of(cs, c) && !header_has_prog(cs, c)) {
            // a synthetic table without lookups inside a machine: the specialised kernel (same values as its program form, bit for bit)
            ZK_TRY(run_quotient(ctx, tlde + tl_off[c], widths[c], log_ns[c], widths[c], alpha, lu, qchunk, own_coset_direct ? qlde + ql_off[c] : nullptr, 8));
        } else if (cs.machine && lookup_of(cs, c)) {
            // the chip's lookup constraints fold after its program's: the program's weights move up by alpha^(cols + 3)
            const LookupView& lv = *lookup_of(cs, c);
            void* v_add;
            ZK_TRY(ctx_reserve(ctx, S_ADDEND, ((size_t)1 << (log_ns[c] + lq_of(cs, c))) * 16, &v_add));
            ZK_TRY(run_lookup_addend(ctx, tlde + tl_off[c], cw[c], plde + pl_off[c], wp[c], log_ns[c], lq_of(cs, c), lv, gamma, beta_l, alpha, cumsum[c], (uint32_t*)v_add));
            ZK_TRY(run_quotient_air(ctx, *prog_of(cs, c), tlde + tl_off[c], cw[c], log_ns[c], (uint32_t)cw[c], public_values, alpha, qchunk,
                                    own_coset_direct ? qlde + ql_off[c] : nullptr, 8, ext_pow(alpha, lv.cols + 3), (const uint32_t*)v_add));
        } else if (prog_of(cs, c)) ZK_TRY(run_quotient_air(ctx, *prog_of(cs, c), tlde + tl_off[c], cw[c], log_ns[c], (uint32_t)cw[c], public_values, alpha, qchunk,
                                                own_coset_direct ? qlde + ql_off[c] : nullptr, 8));
        else ZK_TRY(run_quotient(ctx, tlde + tl_off[c], widths[c], log_ns[c], widths[c], alpha, lu, qchunk, own_coset_direct ? qlde + ql_off[c] : nullptr, 8));
        const uint32_t w2n = two_adic_generator(log_ns[c] + lq_of(cs, c));
        for (int k = 0; k < (1 << lq_of(cs, c)); k++) {
            if (own_coset_direct)                              // (blowup 2: every chip has two chunks)
                ZK_TRY(op_coset_lde(ctx, qchunk + (size_t)k * nc * 4, 4, qlde + ql_off[c] + (size_t)(1 - k) * nc * 8 + 4 * k, 8, log_ns[c], 4, 0,
                                    k == 0 ? w2n : finv(w2n)));
            else
                ZK_TRY(op_coset_lde(ctx, qchunk + (size_t)k * nc * 4, 4, qlde + ql_off[c] + 4 * k, qw_of(cs, c), log_ns[c], 4, b, finv(fpow(w2n, (uint64_t)k))));
        }
    }
    ZK_TRY(op_merkle_commit_mixed(ctx, qm, lh, n, qtree));
    ZK_TRY(d2h(ctx, root, qtree + (2 * mmax - 2) * 8, 32));
    for (int i = 0; i < 8; i++) { ch.observe(root[i]); pf[pos++] = from_monty(root[i]); }

    // ---- 3. openings: one zeta, per-chip "next" point zeta * g_c
    const Ext zeta = ch.sample_ext();
    void *v_dinv, *v_open;
    ZK_TRY(ctx_reserve(ctx, S_DINV, dv_off[n] * 4, &v_dinv));
    ZK_TRY(ctx_reserve(ctx, S_OPEN_OUT, op_off[n] * 4, &v_open));
    uint32_t *dinv = (uint32_t*)v_dinv, *d_open = (uint32_t*)v_open;
    Ext znext[MAX_CHIPS];
    for (int c = 0; c < n; c++) {
        const size_t mc = (size_t)1 << lh[c], nc = (size_t)1 << log_ns[c];
        znext[c] = ext_mul_base(zeta, two_adic_generator(log_ns[c]));
        const Ext zpts[2] = {zeta, znext[c]};
        ZK_TRY(ensure_domain(ctx, log_ns[c], b));
        uint32_t* dv = dinv + dv_off[c];
        uint32_t* xw = dv + 8 * mc;
        ZK_HIP(launch_inv_denominators(ctx->dom_xs, mc, zpts[0], zpts[1], 2, dv, xw, nc, st));
        const size_t pw = pre_w(cs, c);
        uint32_t* oc = d_open + op_off[c] + 8 * pw;                       // the chip's openings: [preprocessed local | next] first
        if (pw) ZK_TRY(run_open(ctx, tlde + tl_off[c], cw[c], log_ns[c], (uint32_t)pw, zpts, 2, xw, d_open + op_off[c]));
        ZK_TRY(run_open(ctx, tlde + tl_off[c] + pw, cw[c], log_ns[c], widths[c], zpts, 2, xw, oc));
        if (wp[c]) ZK_TRY(run_open(ctx, plde + pl_off[c], wp[c], log_ns[c], (uint32_t)wp[c], zpts, 2, xw, oc + 8 * (size_t)widths[c]));
        ZK_TRY(run_open(ctx, qlde + ql_off[c], qw_of(cs, c), log_ns[c], (uint32_t)qw_of(cs, c), zpts, 1, xw, oc + 8 * (size_t)widths[c] + 8 * wp[c]));
    }
    std::vector<uint32_t> opened(op_off[n]);
    ZK_TRY(d2h(ctx, opened.data(), d_open, opened.size() * 4));
    for (size_t i = 0; i < opened.size(); i++) { ch.observe(opened[i]); pf[pos++] = from_monty(opened[i]); }

    // ---- 4. one reduced-opening vector per height (alpha powers run on across the chips of a height)
    const Ext fa = ch.sample_ext();
    std::vector<size_t> layer_off(L + 1), tree_off(L + 1);
    {
        size_t lo = 0, to = 0;
        for (int l = 0; l <= L; l++) {
            layer_off[l] = lo; tree_off[l] = to;
            lo += ((size_t)1 << (Hmax - l)) * 4;
            if (l < L) to += (2 * ((size_t)1 << (Hmax - 1 - l)) - 1) * 8;
        }
    }
    void *v_apf, *v_layers, *v_ltrees, *v_ro, *v_at;
    ZK_TRY(ctx_reserve(ctx, S_APOW_F, ap_off[n] * 4, &v_apf));
    ZK_TRY(ctx_reserve(ctx, S_FRI_LAYERS, 2 * mmax * 16, &v_layers));
    ZK_TRY(ctx_reserve(ctx, S_FRI_TREES, 2 * mmax * 32, &v_ltrees));
    ZK_TRY(ctx_reserve(ctx, S_RO, mmax * 16, &v_ro));             // heights below Hmax: sum of 2^h < 2^Hmax entries
    ZK_TRY(ctx_reserve(ctx, S_PARTIAL, 2 * mmax * 16, &v_at));
    uint32_t *layers = (uint32_t*)v_layers, *ltrees = (uint32_t*)v_ltrees;
    const uint32_t* inject[32] = {nullptr};
    uint32_t* ro_of[32] = {nullptr};
    {
        size_t used = 0;
        for (int c = 0; c < n; c++)
            if (lh[c] != Hmax && !ro_of[lh[c]]) { ro_of[lh[c]] = (uint32_t*)v_ro + used; used += ((size_t)1 << lh[c]) * 4; inject[lh[c]] = ro_of[lh[c]]; }
        ro_of[Hmax] = layers;
    }
    std::vector<uint32_t> apows(ap_off[n]);
    for (int c = 0; c < n; c++) {
        const size_t np = (ap_off[c + 1] - ap_off[c]) / 4;
        Ext* fp = (Ext*)(apows.data() + ap_off[c]);
        fp[0] = ext_one();
        for (size_t j = 1; j < np; j++) fp[j] = ext_mul(fp[j - 1], fa);
    }
    ZK_TRY(h2d(ctx, v_apf, apows.data(), apows.size() * 4));
    bool started[32] = {false};
    for (int c = 0; c < n; c++) {
        const uint32_t W = widths[c];
        const Ext* fp = (const Ext*)(apows.data() + ap_off[c]);
        const uint32_t Pw = pre_w(cs, c);
        const Ext* op_el = (const Ext*)(opened.data() + op_off[c]);
        const Ext* op_en = op_el + Pw;
        const Ext* op_loc = op_en + Pw;
        const Ext* op_nxt = op_loc + W;
        const Ext* op_pl = op_nxt + W;
        const Ext* op_pn = op_pl + wp[c];
        const Ext* op_q = op_pn + wp[c];
        ReducedArgs ra{};
        ra.y_loc = ra.y_next = ra.y_pl = ra.y_pn = ra.y_q = ext_zero();
        for (uint32_t j = 0; j < W; j++) {
            ra.y_loc = ext_add(ra.y_loc, ext_mul(fp[j], op_loc[j]));
            ra.y_next = ext_add(ra.y_next, ext_mul(fp[j], op_nxt[j]));
        }
        for (size_t j = 0; j < wp[c]; j++) {
            ra.y_pl = ext_add(ra.y_pl, ext_mul(fp[j], op_pl[j]));
            ra.y_pn = ext_add(ra.y_pn, ext_mul(fp[j], op_pn[j]));
        }
        for (size_t j = 0; j < qw_of(cs, c); j++) ra.y_q = ext_add(ra.y_q, ext_mul(fp[j], op_q[j]));
        const uint64_t off0 = height_offset(cs, log_ns, widths, pairs, c), off = off0 + 2 * (uint64_t)Pw;
        ra.off_loc = ext_pow(fa, off); ra.off_next = ext_pow(fa, off + W);
        ra.off_pl = ext_pow(fa, off + 2 * (uint64_t)W); ra.off_pn = ext_pow(fa, off + 2 * (uint64_t)W + wp[c]);
        ra.off_q = ext_pow(fa, off + 2 * (uint64_t)W + 2 * wp[c]);
        ra.tlde = tlde + tl_off[c] + Pw; ra.t_ld = cw[c]; ra.width = W; ra.qlde = qlde + ql_off[c]; ra.q_ld = qw_of(cs, c); ra.q_width = (uint32_t)qw_of(cs, c); ra.rows = (uint64_t)1 << lh[c];
        ra.plde = wp[c] ? plde + pl_off[c] : nullptr; ra.p_ld = wp[c]; ra.p_width = (uint32_t)wp[c];
        ra.alpha_pow = (const uint32_t*)v_apf + ap_off[c]; ra.dinv = dinv + dv_off[c]; ra.out = ro_of[lh[c]];
        ra.accumulate = started[lh[c]] ? 1 : 0;
        started[lh[c]] = true;
        ZK_HIP(launch_reduced_opening(ra, (uint32_t*)v_at, st));
        if (Pw) {                                  // the preprocessed columns' two terms, added on: the same kernel without permutation / quotient parts
            ReducedArgs re = ra;
            re.tlde = tlde + tl_off[c]; re.width = Pw; re.p_width = 0; re.plde = nullptr; re.q_width = 0;
            re.y_loc = re.y_next = re.y_q = ext_zero();
            for (uint32_t j = 0; j < Pw; j++) {
                re.y_loc = ext_add(re.y_loc, ext_mul(fp[j], op_el[j]));
                re.y_next = ext_add(re.y_next, ext_mul(fp[j], op_en[j]));
            }
            re.off_loc = ext_pow(fa, off0); re.off_next = ext_pow(fa, off0 + Pw);
            re.accumulate = 1;
            ZK_HIP(launch_reduced_opening(re, (uint32_t*)v_at, st));
        }
    }

    // ---- 5. FRI commit phase; shorter vectors join at their height
    ZK_TRY(ensure_domain(ctx, log_ns[0], b));                   // fold twiddles of the largest domain
    ZK_TRY(fri_commit_phase(ctx, ch, sh, Hmax, L, layers, ltrees, layer_off, tree_off, (uint32_t*)v_at, mmax, inject, pf, pos));
    {
        std::vector<Ext> last((size_t)1 << b);
        ZK_TRY(d2h(ctx, last.data(), layers + layer_off[L], last.size() * 16));
        for (size_t i = 1; i < last.size(); i++)
            if (!ext_eq(last[0], last[i])) return fail(ZKHIP_ERR_INVALID, "prove_chips: final FRI layer is not constant (a trace violates its AIR)");
        for (int e = 0; e < 4; e++) pf[pos++] = from_monty(last[0].c[e]);
        ch.observe_ext(last[0]);
    }

    // ---- 6. proof of work, 7. queries
    uint32_t witness = 0;
    ZK_TRY(grind_witness(ctx, ch, prm->pow_bits, &witness));
    pf[pos++] = witness;
    {
        std::vector<GatherDesc> descs;
        size_t qpos = 0;
        auto push = [&](const uint32_t* src, size_t nwords) { descs.push_back(GatherDesc{src, (uint32_t)qpos, (uint32_t)nwords}); qpos += nwords; };
        auto push_path = [&](const uint32_t* tree, size_t leaves, size_t index, int levels) {
            const uint32_t* lvl = tree; size_t cnt = leaves, idx = index;
            for (int k = 0; k < levels; k++) { push(lvl + 8 * (idx ^ 1), 8); lvl += 8 * cnt; cnt >>= 1; idx >>= 1; }
        };
        for (int q = 0; q < Q; q++) {
            const size_t index = ch.sample_bits(Hmax);
            if (cs.key) {
                for (int c = 0; c < n; c++) if (pre_w(cs, c)) push(tlde + tl_off[c] + (index >> (Hmax - lh[c])) * cw[c], pre_w(cs, c));
                push_path(cs.key->d_tree, (size_t)1 << cs.key->He, index >> (Hmax - cs.key->He), cs.key->He);
            }
            for (int c = 0; c < n; c++) push(tlde + tl_off[c] + (index >> (Hmax - lh[c])) * cw[c] + pre_w(cs, c), widths[c]);
            push_path(ttree, mmax, index, Hmax);
            if (lk) {
                for (int c = 0; c < n; c++) if (wp[c]) push(plde + pl_off[c] + (index >> (Hmax - lh[c])) * wp[c], wp[c]);
                push_path(ptree, (size_t)1 << Hp, index >> (Hmax - Hp), Hp);
            }
            for (int c = 0; c < n; c++) push(qlde + ql_off[c] + (index >> (Hmax - lh[c])) * qw_of(cs, c), qw_of(cs, c));
            push_path(qtree, mmax, index, Hmax);
            size_t idx = index;
            for (int l = 0; l < L; l++) {
                const int rows_log = Hmax - 1 - l;
                push(layers + layer_off[l] + (idx ^ 1) * 4, 4);
                push_path(ltrees + tree_off[l], (size_t)1 << rows_log, idx >> 1, rows_log);
                idx >>= 1;
            }
        }
        if (pos + qpos != need / 4) return fail(ZKHIP_ERR_INTERNAL, "prove_chips: proof layout mismatch");
        void *v_desc, *v_out;
        ZK_TRY(ctx_reserve(ctx, S_GATHER_DESC, descs.size() * sizeof(GatherDesc), &v_desc));
        ZK_TRY(ctx_reserve(ctx, S_GATHER_OUT, qpos * 4, &v_out));
        ZK_TRY(h2d(ctx, v_desc, descs.data(), descs.size() * sizeof(GatherDesc)));
        ZK_HIP(launch_gather((const GatherDesc*)v_desc, (uint32_t)descs.size(), (uint32_t*)v_out, st));
        ZK_TRY(d2h(ctx, pf + pos, v_out, qpos * 4));
        pos += qpos;
    }
    *len = pos * 4;
    return ZKHIP_OK;
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

// ---- chips with their own constraint programs (programs[c] == NULL: the built-in synthetic AIR); degree <= 3, no lookups ----
int zkhip_prove_chips(zkhip_ctx* ctx, const zkhip_chip* chips, int n, const uint32_t* public_values, size_t n_public,
                      const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len) {
    return prove_chips_impl(ChipSet{}, ctx, chips, n, public_values, n_public, prm, proof, cap, len);
}
int zkhip_verify_chips(const uint8_t* proof, size_t len, const int32_t* log_ns, const uint32_t* widths, const int32_t* pairs, const int32_t* partners, int n,
                       const uint32_t* public_values, size_t n_public, const zkhip_params* prm, int* reason) {
    return verify_chips_impl(ChipSet{}, proof, len, log_ns, widths, pairs, partners, n, public_values, n_public, prm, reason);
}
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
size_t zkhip_chips_proof_size_air(const int32_t* log_ns, const uint32_t* widths, const uint32_t* const* programs, const size_t* program_words,
                                  int n_chips, const zkhip_params* prm, size_t n_public) {
    AirView views[MAX_CHIPS];
    const AirView* table[MAX_CHIPS];
    if (chip_programs(programs, program_words, widths, n_chips, n_public, views, table) != ZKHIP_OK) return 0;
    ChipSet cs;
    cs.air = table;
    return chips_proof_size_impl(cs, log_ns, widths, nullptr, nullptr, n_chips, prm, n_public);
}
int zkhip_prove_chips_air(zkhip_ctx* ctx, const zkhip_chip* chips, const uint32_t* const* programs, const size_t* program_words, int n_chips,
                          const uint32_t* public_values, size_t n_public, const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len) {
    if (!chips || n_chips < 1 || n_chips > MAX_CHIPS) return fail(ZKHIP_ERR_INVALID, "prove_chips_air: bad arguments");
    uint32_t widths[MAX_CHIPS];
    for (int c = 0; c < n_chips; c++) {
        widths[c] = chips[c].width;
        if (chips[c].logup_pairs || chips[c].partner >= 0) return fail(ZKHIP_ERR_INVALID, "prove_chips_air: no lookups next to constraint programs");
    }
    AirView views[MAX_CHIPS];
    const AirView* table[MAX_CHIPS];
    ZK_TRY(chip_programs(programs, program_words, widths, n_chips, n_public, views, table));
    ChipSet cs;
    cs.air = table;
    return prove_chips_impl(cs, ctx, chips, n_chips, public_values, n_public, prm, proof, cap, len);
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

// ---- the machine: chips with programs AND interaction tables (lookups as data); proof version 10 ----
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
size_t zkhip_machine_proof_size(const int32_t* log_ns, const uint32_t* widths, const uint32_t* const* programs, const size_t* program_words,
                                const uint32_t* const* tables, const size_t* table_words, int n_chips, const zkhip_params* prm, size_t n_public) {
    MachineSetup m;
    if (machine_setup(programs, program_words, tables, table_words, widths, n_chips, n_public, m) != ZKHIP_OK) return 0;
    ChipSet cs;
    cs.air = m.table; cs.machine = &m.mt;
    return chips_proof_size_impl(cs, log_ns, widths, m.cols, nullptr, n_chips, prm, n_public);
}
int zkhip_prove_machine(zkhip_ctx* ctx, const zkhip_chip* chips, const uint32_t* const* programs, const size_t* program_words,
                        const uint32_t* const* tables, const size_t* table_words, int n_chips, const uint32_t* public_values, size_t n_public,
                        const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len) {
    if (!chips || n_chips < 1 || n_chips > MAX_CHIPS) return fail(ZKHIP_ERR_INVALID, "prove_machine: bad arguments");
    uint32_t widths[MAX_CHIPS];
    for (int c = 0; c < n_chips; c++) widths[c] = chips[c].width;
    MachineSetup m;
    ZK_TRY(machine_setup(programs, program_words, tables, table_words, widths, n_chips, n_public, m));
    zkhip_chip mine[MAX_CHIPS];
    for (int c = 0; c < n_chips; c++) { mine[c] = chips[c]; mine[c].logup_pairs = m.cols[c]; mine[c].partner = -1; }
    ChipSet cs;
    cs.air = m.table; cs.machine = &m.mt;
    return prove_chips_impl(cs, ctx, mine, n_chips, public_values, n_public, prm, proof, cap, len);
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

// ---- the keyed machine: preprocessed columns committed once; proof version 11 ----
struct zkhip_machine_key {
    zkhip_ctx* ctx = nullptr;            // identity only (a key serves the context it was made with)
    int device = 0;
    int n = 0, b = 0;
    int32_t log_ns[MAX_CHIPS];
    KeyView view{};
    std::vector<void*> owned;           // device allocations of the key
};
void zkhip_machine_key_destroy(zkhip_machine_key* key) {
    if (!key) return;
    if (!key->owned.empty()) {          // the context may be gone by now: only the device is needed (hipFree waits for work in flight)
        (void)hipSetDevice(key->device);
        for (void* p : key->owned) (void)hipFree(p);
    }
    delete key;
}
int zkhip_machine_setup(zkhip_ctx* ctx, const zkhip_chip* pre, int n_chips, const zkhip_params* prm, zkhip_machine_key** key_out, uint32_t root[8]) {
    CHECK_CTX(ctx);
    if (!pre || !prm || !key_out || !root || n_chips < 1 || n_chips > MAX_CHIPS) return fail(ZKHIP_ERR_INVALID, "machine_setup: bad arguments");
    if (prm->log_blowup < 1 || prm->log_blowup > 3) return fail(ZKHIP_ERR_INVALID, "machine_setup: log_blowup in [1,3]");
    *key_out = nullptr;
    zkhip_machine_key* key = new (std::nothrow) zkhip_machine_key;
    if (!key) return fail(ZKHIP_ERR_INTERNAL, "machine_setup: out of memory");
    struct Guard { zkhip_machine_key* k; ~Guard() { if (k) zkhip_machine_key_destroy(k); } } guard{key};
    key->ctx = ctx; key->device = ctx->device; key->n = n_chips; key->b = prm->log_blowup;
    const int b = prm->log_blowup;
    MatDesc mats[MAX_CHIPS]; int lhs[MAX_CHIPS]; int ne = 0, He = 0;
    auto dev_alloc = [&](size_t bytes, uint32_t** out) -> int {
        void* p = nullptr;
        ZK_HIP(hipMalloc(&p, bytes));
        key->owned.push_back(p);
        *out = (uint32_t*)p;
        return ZKHIP_OK;
    };
    for (int c = 0; c < n_chips; c++) {
        const uint32_t pw = pre[c].width;
        key->log_ns[c] = pre[c].log_n; key->view.pw[c] = pw; key->view.d_trace[c] = nullptr; key->view.d_lde[c] = nullptr;
        if (pre[c].log_n < 5 || pre[c].log_n > MAX_LOG_ROWS || (c && pre[c].log_n > pre[c - 1].log_n)) return fail(ZKHIP_ERR_INVALID, "machine_setup: log_n in [5,22], tallest first");
        if (pw % 4 != 0 || pw > 1024) return fail(ZKHIP_ERR_INVALID, "machine_setup: preprocessed width a multiple of 4 up to 1024 (0: none)");
        if (!pw) continue;
        if (!pre[c].d_trace || pre[c].ld < pw) return fail(ZKHIP_ERR_INVALID, "machine_setup: bad preprocessed trace descriptor");
        const size_t nc = (size_t)1 << pre[c].log_n, mc = nc << b;
        uint32_t *tr, *lde;
        ZK_TRY(dev_alloc(nc * pw * 4, &tr));
        ZK_TRY(dev_alloc(mc * pw * 4, &lde));
        ZK_HIP(hipMemcpy2DAsync(tr, (size_t)pw * 4, pre[c].d_trace, pre[c].ld * 4, (size_t)pw * 4, nc, hipMemcpyDeviceToDevice, ctx->stream));
        ZK_TRY(op_coset_lde(ctx, tr, pw, lde, pw, pre[c].log_n, pw, b, MONTY_GEN));
        key->view.d_trace[c] = tr; key->view.d_lde[c] = lde;
        mats[ne] = MatDesc{lde, pw, pw}; lhs[ne] = pre[c].log_n + b; ne++;
        if (pre[c].log_n + b > He) He = pre[c].log_n + b;
    }
    if (!ne) return fail(ZKHIP_ERR_INVALID, "machine_setup: no chip has preprocessed columns");
    uint32_t* tree;
    ZK_TRY(dev_alloc((2 * ((size_t)1 << He) - 1) * 32, &tree));
    ZK_TRY(op_merkle_commit_mixed(ctx, mats, lhs, ne, tree));
    ZK_TRY(d2h(ctx, key->view.root_m, tree + (2 * ((size_t)1 << He) - 2) * 8, 32));
    key->view.d_tree = tree; key->view.He = He;
    for (int i = 0; i < 8; i++) root[i] = from_monty(key->view.root_m[i]);
    guard.k = nullptr;
    *key_out = key;
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
size_t zkhip_machine_proof_size_keyed(const int32_t* log_ns, const uint32_t* widths, const uint32_t* pre_widths, const uint32_t* const* programs,
                                      const size_t* program_words, const uint32_t* const* tables, const size_t* table_words, int n_chips,
                                      const zkhip_params* prm, size_t n_public) {
    uint32_t cw[MAX_CHIPS];
    if (keyed_widths(widths, pre_widths, n_chips, cw) != ZKHIP_OK) return 0;
    MachineSetup m;
    if (machine_setup(programs, program_words, tables, table_words, cw, n_chips, n_public, m) != ZKHIP_OK) return 0;
    KeyView kv{};
    for (int c = 0; c < n_chips; c++) kv.pw[c] = pre_widths[c];
    ChipSet cs;
    cs.air = m.table; cs.machine = &m.mt; cs.key = &kv;
    return chips_proof_size_impl(cs, log_ns, widths, m.cols, nullptr, n_chips, prm, n_public);
}
int zkhip_prove_machine_keyed_at(zkhip_ctx* ctx, const zkhip_machine_key* key, const int32_t* key_entries, const zkhip_chip* chips,
                                 const uint32_t* const* programs, const size_t* program_words, const uint32_t* const* tables, const size_t* table_words,
                                 int n_chips, const uint32_t* public_values, size_t n_public, const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len) {
    CHECK_CTX(ctx);
    if (!key || !chips || !prm || n_chips < 1 || n_chips > MAX_CHIPS) return fail(ZKHIP_ERR_INVALID, "prove_machine_keyed: bad arguments");
    if (key->ctx != ctx) return fail(ZKHIP_ERR_INVALID, "prove_machine_keyed: the key belongs to another context");
    if (key->b != prm->log_blowup || (!key_entries && key->n != n_chips))
        return fail(ZKHIP_ERR_INVALID, "prove_machine_keyed: the key was set up for another machine shape (chips / log_blowup)");
    // the machine's view of the key: chip c uses entry key_entries[c] (none: -1); every entry that has columns is used once, in the
    // key's order (the rows of equally tall entries are hashed in that order)
    KeyView kv = key->view;
    uint32_t widths[MAX_CHIPS], cw[MAX_CHIPS];
    int last = -1, used = 0, have = 0;
    for (int e = 0; e < key->n; e++) have += key->view.pw[e] != 0;
    for (int c = 0; c < n_chips; c++) {
        widths[c] = chips[c].width;
        const int e = key_entries ? key_entries[c] : c;
        kv.pw[c] = 0; kv.d_trace[c] = nullptr; kv.d_lde[c] = nullptr;
        if (e < 0 || (e < key->n && key->view.pw[e] == 0)) continue;
        if (e >= key->n || e <= last) return fail(ZKHIP_ERR_INVALID, "prove_machine_keyed: key entries must be used once each, in the key's order");
        if (chips[c].log_n != key->log_ns[e]) return fail(ZKHIP_ERR_INVALID, "prove_machine_keyed: a chip's height differs from the key's");
        kv.pw[c] = key->view.pw[e]; kv.d_trace[c] = key->view.d_trace[e]; kv.d_lde[c] = key->view.d_lde[e];
        last = e; used++;
    }
    if (used != have) return fail(ZKHIP_ERR_INVALID, "prove_machine_keyed: the machine leaves out a preprocessed table of the key");
    ZK_TRY(keyed_widths(widths, kv.pw, n_chips, cw));
    MachineSetup m;
    ZK_TRY(machine_setup(programs, program_words, tables, table_words, cw, n_chips, n_public, m));
    zkhip_chip mine[MAX_CHIPS];
    for (int c = 0; c < n_chips; c++) { mine[c] = chips[c]; mine[c].logup_pairs = m.cols[c]; mine[c].partner = -1; }
    ChipSet cs;
    cs.air = m.table; cs.machine = &m.mt; cs.key = &kv;
    return prove_chips_impl(cs, ctx, mine, n_chips, public_values, n_public, prm, proof, cap, len);
}
int zkhip_prove_machine_keyed(zkhip_ctx* ctx, const zkhip_machine_key* key, const zkhip_chip* chips, const uint32_t* const* programs,
                              const size_t* program_words, const uint32_t* const* tables, const size_t* table_words, int n_chips,
                              const uint32_t* public_values, size_t n_public, const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len) {
    return zkhip_prove_machine_keyed_at(ctx, key, nullptr, chips, programs, program_words, tables, table_words, n_chips, public_values, n_public, prm, proof, cap, len);
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
            p2_permute(s1);
            for (int e = 0; e < 16; e++) if (s1[e] != st[e][j]) return fail(ZKHIP_ERR_INTERNAL, "host SIMD permutation differs from the scalar one");
        }
    }
    const int reps = 2000;
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; k++) p2x16_permute(st);
    const double a = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / (16.0 * reps);
    uint32_t s1[16];
    for (int e = 0; e < 16; e++) s1[e] = st[e][0];
    t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; k++) p2_permute(s1);
    const double b = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / reps;
    if (ns_x16) *ns_x16 = a;
    if (ns_scalar) *ns_scalar = b + (s1[0] == 0xFFFFFFFFu ? 1 : 0);
    return 1;
}

int zkhip_last_prove_debug(zkhip_ctx* ctx, zkhip_prove_debug* out) {
    if (!ctx || !out) return fail(ZKHIP_ERR_INVALID, "null argument");
    *out = ctx->debug;
    return ZKHIP_OK;
}

}  // extern "C"
