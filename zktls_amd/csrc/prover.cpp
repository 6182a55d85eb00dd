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

#include "proof_common.h"
#include "batch.h"

namespace zk { std::atomic<bool> g_fri_graph{true}; }
namespace zk {

// The program digest is a sponge over every 16-bit half of the program (4 500 permutations for the 18 000-word SHA-256 chip: 6.5 ms
// on the host), needed by the header and the transcript of every proof and verification: the last few programs' digests are kept,
// keyed by the program's full contents (an exact comparison, ~5 us for that program).
extern std::atomic<uint64_t> g_p2_generation;      // params.cpp: bumped when the Poseidon2 table set changes
void air_digest_cached(const AirView& a, uint32_t out[8]) {
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
    if (cache.size() >= 64) cache.erase(cache.begin());      // (a five-chip machine alone has ten: programs and interaction tables)
    Entry e;
    e.words.assign(a.w, a.w + a.words);
    e.generation = gen;
    memcpy(e.dg, out, 32);
    cache.push_back(std::move(e));
}

// workspace roles: enum Slot in context.h
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
    if (lu.pairs) {                      // the LogUp constraints first, one lane per point; the chain kernel adds them in
        void* v_add;
        ZK_TRY(ctx_reserve(ctx, S_ADDEND, ((size_t)2 << log_n) * 16, &v_add));
        q.addend_out = (uint32_t*)v_add;
        ZK_HIP(launch_logup_addend(q, ctx->stream));
        q.addend = (const uint32_t*)v_add;
    }
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
    // the flattened form for the term-parallel kernels: one record per monomial of COLUMNS and selectors -- a term's public-value factors are
    // constants of the proof and multiply its coefficient here (until round 5 they travelled in every point's LDS slots, which held a
    // program to 256 of them: the shard verifier's transcript table reads every public value of a join -- 64 x 91 -- and its 1 472
    // constraints ran on the row-per-lane interpreter, 15 ms of a 156 ms compression)
    std::vector<uint32_t> recs;
    uint32_t cls[6] = {0, 0, 0, 0, 0, 0};
    // (not inside a lock-step batch: the batched twin takes its arguments from memory, its record loads then are per-lane loads of one
    // address instead of scalar loads, and that form is slower than the 8-point kernel)
    const bool wide = !t_batcher && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(lde) & 15u) == 0 &&
                      air_wide_form(width, (uint32_t)air_term_count(air), log_n, 0);
    if (wide) air_term_records_wide(air, alpha, pub.data(), recs, cls, scale);
    else air_term_records(air, alpha, pub.data(), recs, scale);
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
    q.n_terms = (uint32_t)(recs.size() / 8); q.n_public = recs.empty() ? air.n_public : 0u;
    q.wide = wide ? 1u : 0u;
    for (int i = 0; i < 6; i++) q.cls[i] = cls[i];
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
    static const bool env_graph = [] { const char* e = getenv("ZKHIP_FRI_GRAPH"); return !e || atoi(e) != 0; }();
    const bool use_graph = env_graph && g_fri_graph.load();
#else
    const bool use_graph = g_fri_graph.load();       // zkhip_set_fri_graph (a debugging switch)
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

}  // namespace zk

using namespace zk;

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
    const size_t rec_w = (recs.size() + 3) & ~(size_t)3, map_w = LOOKUP_MAX_COLS / 4;
    std::vector<uint32_t> stage(rec_w + weights.size() + map_w, 0u);
    memcpy(stage.data(), recs.data(), recs.size() * 4);
    if (!weights.empty()) memcpy(stage.data() + rec_w, weights.data(), weights.size() * 4);
    // the columns the interactions read, in the order they are first met: the kernels stage exactly these (kernels.h LookupArgs)
    uint8_t* cmap = (uint8_t*)(stage.data() + rec_w + weights.size());
    memset(cmap, 0xFF, LOOKUP_MAX_COLS);
    uint32_t n_used = 0, chunk_mask = 0;
    bool stageable = true;
    auto touch = [&](uint32_t col) {
        if (col >= LOOKUP_MAX_COLS) { stageable = false; return; }
        if (cmap[col] == 0xFF) { if (n_used >= LOOKUP_STAGE_MAX_USED) { stageable = false; return; } cmap[col] = (uint8_t)n_used++; }
        chunk_mask |= 1u << (col >> 4);
    };
    for (uint32_t i = 0; i < lv.ni && stageable; i++) {
        const uint32_t* rec = recs.data() + (size_t)i * LOOKUP_REC_WORDS;
        if (rec[1] != 0xFFFFFFFFu) touch(rec[1]);
        for (uint32_t v = 0; v < rec[3]; v++) touch(rec[4 + v]);
    }
    void* d_stage;
    ZK_TRY(ctx_reserve(ctx, S_LOOKUP, stage.size() * 4, &d_stage));
    ZK_TRY(h2d(ctx, d_stage, stage.data(), stage.size() * 4));
    lk->table = (const uint32_t*)d_stage; lk->ni = lv.ni; lk->cols = lv.cols; lk->gamma = gamma;
    lk->cmap = stageable && n_used ? (const uint8_t*)((const uint32_t*)d_stage + rec_w + weights.size()) : nullptr;
    lk->n_used = n_used; lk->chunk_mask = chunk_mask;
    lk->bpow[0] = ext_one();
    for (int t = 1; t < 9; t++) lk->bpow[t] = ext_mul(lk->bpow[t - 1], beta);
    if (d_weights) *d_weights = (const uint32_t*)d_stage + rec_w;
    return ZKHIP_OK;
}
// (pre != nullptr: the rows are [pre | trace]; *took_two = whether the kernel could read the two where they lie -- else the caller copies them side by side and calls again)
static int run_lookup_perm(zkhip_ctx* ctx, const uint32_t* d_trace, size_t ld, int log_n, const LookupView& lv, const Ext& gamma, const Ext& beta,
                           uint32_t* d_out, const uint32_t* d_pre = nullptr, size_t pre_ld = 0, uint32_t pre_w = 0, bool* took_two = nullptr) {
    MachinePermArgs a{};
    ZK_TRY(lookup_args(ctx, lv, gamma, beta, {}, &a.lk, nullptr));
    a.trace = d_trace; a.ld = ld; a.rows = (uint64_t)1 << log_n; a.out = d_out; a.out_ld = 4 * ((uint64_t)lv.cols + 1);
    if (d_pre) {
        a.pre = d_pre; a.pre_ld = pre_ld; a.pre_w = pre_w;
        const bool ok = lookup_perm_two_sources_ok(a);
        if (took_two) *took_two = ok;
        if (!ok) return ZKHIP_OK;
    }
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
    observe_words(ch, opened.data(), opened.size());          // (inside a lock-step batch: the members' sponges side by side, proof_common.h)
    for (size_t i = 0; i < opened.size(); i++) pf[pos++] = from_monty(opened[i]);
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
#ifdef ZKHIP_AB_HOOKS
    // A/B build only: ZKHIP_CHIPS_TIMING=1 prints the wall time of every phase (the stream is drained at each lap)
    static const bool timing = getenv("ZKHIP_CHIPS_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "    [chips prover] %-36s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
#else
    auto lap = [](const char*) {};
#endif

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
    lap("1. trace LDEs");
    ZK_TRY(op_merkle_commit_mixed(ctx, tm, lh, n, ttree));
    lap("1. trace tree");
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
                // the kernel stages the few columns it reads from the key's trace and the main trace where they lie (round 6); when it cannot, the two are put side by side first
                bool two = false;
                ZK_TRY(run_lookup_perm(ctx, chips[c].d_trace, chips[c].ld, log_ns[c], *lookup_of(cs, c), gamma, beta_l, (uint32_t*)v_perm, cs.key->d_trace[c], pw, (uint32_t)pw, &two));
                if (!two) {
                    void* v_ct;
                    ZK_TRY(ctx_reserve(ctx, S_KEYTRACE, nc * cw[c] * 4, &v_ct));
                    uint32_t* ct = (uint32_t*)v_ct;
                    ZK_HIP(launch_copy2d(ct, cw[c], cs.key->d_trace[c], pw, (uint32_t)pw, nc, st));
                    ZK_HIP(launch_copy2d(ct + pw, cw[c], chips[c].d_trace, chips[c].ld, widths[c], nc, st));
                    ZK_TRY(run_lookup_perm(ctx, ct, cw[c], log_ns[c], *lookup_of(cs, c), gamma, beta_l, (uint32_t*)v_perm));
                }
            } else if (cs.machine) ZK_TRY(run_lookup_perm(ctx, chips[c].d_trace, chips[c].ld, log_ns[c], *lookup_of(cs, c), gamma, beta_l, (uint32_t*)v_perm));
            else ZK_TRY(run_perm_trace(ctx, chips[c].d_trace, chips[c].ld, log_ns[c], (uint32_t)pairs[c], gamma, beta_l, (uint32_t*)v_perm));
            if (cross)          // the running sum's last value: row N - 1, column S
                ZK_TRY(d2h(ctx, &cumsum[c], (const uint32_t*)v_perm + (((size_t)1 << log_ns[c]) - 1) * wp[c] + 4 * (size_t)pairs[c], 16));
            ZK_TRY(op_coset_lde(ctx, (const uint32_t*)v_perm, wp[c], plde + pl_off[c], wp[c], log_ns[c], (uint32_t)wp[c], b, MONTY_GEN));
            pmats[np] = MatDesc{plde + pl_off[c], wp[c], (uint32_t)wp[c]}; plh[np] = lh[c]; np++;
        }
        lap("1b. permutation traces + LDEs");
        ZK_TRY(op_merkle_commit_mixed(ctx, pmats, plh, np, ptree));
        lap("1b. permutation tree");
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
    lap("2. quotients + chunk LDEs");
    ZK_TRY(op_merkle_commit_mixed(ctx, qm, lh, n, qtree));
    ZK_TRY(d2h(ctx, root, qtree + (2 * mmax - 2) * 8, 32));
    for (int i = 0; i < 8; i++) { ch.observe(root[i]); pf[pos++] = from_monty(root[i]); }
    lap("2. quotient tree");

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
    observe_words(ch, opened.data(), opened.size());          // (inside a lock-step batch: the members' sponges side by side, proof_common.h)
    for (size_t i = 0; i < opened.size(); i++) pf[pos++] = from_monty(opened[i]);
    lap("3. openings");

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

    lap("4. reduced openings");
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

    lap("5. FRI commit phase");
    // ---- 6. proof of work, 7. queries
    uint32_t witness = 0;
    ZK_TRY(grind_witness(ctx, ch, prm->pow_bits, &witness));
    pf[pos++] = witness;
    lap("6. proof of work");
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
    lap("7. queries");
    *len = pos * 4;
    return ZKHIP_OK;
}

// ---- chips with their own constraint programs (programs[c] == NULL: the built-in synthetic AIR); degree <= 3, no lookups ----
int zkhip_prove_chips(zkhip_ctx* ctx, const zkhip_chip* chips, int n, const uint32_t* public_values, size_t n_public,
                      const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len) {
    return prove_chips_impl(ChipSet{}, ctx, chips, n, public_values, n_public, prm, proof, cap, len);
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
// ---- the machine: chips with programs AND interaction tables (lookups as data); proof version 10 ----
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
int zkhip_last_prove_debug(zkhip_ctx* ctx, zkhip_prove_debug* out) {
    if (!ctx || !out) return fail(ZKHIP_ERR_INVALID, "null argument");
    *out = ctx->debug;
    return ZKHIP_OK;
}

}  // extern "C"

extern "C" void zkhip_set_fri_graph(int on) { zk::g_fri_graph.store(on != 0); }

// ---- self-check of the merged transcript absorption (zkhip_selftest_lockstep; no device): `members` fibers, each with a transcript of its own history, send word
// vectors of several lengths through observe_words at the same points of their programs -- a fifth of them with another length (its own group), one round with
// unequal buffer fills (the one-after-the-other path) --; every state, buffer and squeezed word against the scalar Challenger's.  10 + the failing member's check.
namespace zk {
int lockstep_selftest_observe(int members) {
    if (members < 1 || members > LaunchBatcher::MAX_MEMBERS) return 1;
    LaunchBatcher lb(members, nullptr);
    if (!lb.ok_scheduler()) return 2;
    std::vector<int> bad((size_t)members, 0);
    auto word = [](uint64_t b, uint64_t r, uint64_t i) { uint64_t x = (b + 1) * 0x9E3779B97F4A7C15ull ^ (r + 7) * 0xC2B2AE3D27D4EB4Full ^ (i + 3) * 0x165667B19E3779F9ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return (uint32_t)(x % P); };
    const size_t lens[5] = {64, 71, 200, 1000, 5501};
    lb.run([&](int b) {
        Challenger got, want;
        for (int r = 0; r < 5; r++) {
            const int pre = r == 3 ? b % 3 : r;                       // round 3: the members' buffers are filled unequally
            for (int i = 0; i < pre; i++) { const uint32_t w = word((uint64_t)b, 100 + (uint64_t)r, (uint64_t)i); got.observe(w); want.observe(w); }
            const size_t n = lens[r] + (b % 5 == 4 ? 8 : 0);
            std::vector<uint32_t> ws(n);
            for (size_t i = 0; i < n; i++) ws[i] = word((uint64_t)b, (uint64_t)r, i);
            observe_words(got, ws.data(), n);
            for (size_t i = 0; i < n; i++) want.observe(ws[i]);
            if (std::memcmp(got.state, want.state, sizeof got.state) != 0 || got.n_in != want.n_in || got.n_out != want.n_out || std::memcmp(got.in, want.in, 4 * (size_t)got.n_in) != 0 ||
                std::memcmp(got.out, want.out, 4 * (size_t)got.n_out) != 0) bad[(size_t)b] = 10 + r;
            if (got.sample() != want.sample()) bad[(size_t)b] = 20 + r;
        }
    });
    for (int b = 0; b < members; b++) if (bad[(size_t)b]) return bad[(size_t)b];
    return 0;
}
}  // namespace zk
