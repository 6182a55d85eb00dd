// examples/compress_tree.cpp -- a TREE of joins from plain C++ (crates/guest-prover-sp1/src/sp1.rs:116: core -> compress; sp1-recursion joins the
// joins; RISC Zero: lift -> join, prover.rs:90):
//   1. the shards of an execution proven in ONE call (zkhip_prove_shards);
//   2. level 1: the shard proofs joined `per_join` at a time (zkhip_prove_shard_verifier) -- every join has the same shape, hence the same key;
//   3. level 2: ONE proof that verifies the join proofs in-circuit (zkhip_prove_machine_verifier: machine mode -- a join proof is a
//      keyed-machine proof of eight chips of mixed heights with lookups and preprocessed columns; zkhip_shard_verifier_describe hands out the
//      join machine's description);
//   4. that proof checked on the host from (the join machine's description, the shards' public values, the two keys -- both derived on
//      the host, no device): no shard proof and no join proof is needed any more.
//
//   make -C examples && ./examples/compress_tree [shards=8] [per_join=4] [log_n=12] [width=16]
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "../include/zkhip.h"
#include "../include/zkhip_chips.h"

#define CHECK(call)                                                                 \
    do {                                                                            \
        int rc_ = (call);                                                           \
        if (rc_ != ZKHIP_OK) {                                                      \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, zkhip_last_error()); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// the join machine over `per_join` shard proofs of a shape as a zkhip_machine_desc (the description's words live in `keep`)
struct JoinMachine {
    std::vector<std::vector<uint32_t>> progs, tabs;
    std::vector<const uint32_t*> pp, tp;
    std::vector<size_t> pw, tw;
    int32_t lns[8];
    uint32_t widths[8], pres[8];
    zkhip_machine_desc desc{};
    bool build(int log_n, uint32_t width, const zkhip_params& prm, size_t n_public, size_t per_join, const uint32_t join_key[8]) {
        progs.resize(8); tabs.resize(8); pp.resize(8); tp.resize(8); pw.resize(8); tw.resize(8);
        for (int i = 0; i < 8; i++) {
            int ln = 0;
            uint32_t mw = 0, prw = 0;
            for (int kind = 0; kind < 2; kind++) {
                std::vector<uint32_t>& dst = kind ? tabs[(size_t)i] : progs[(size_t)i];
                const size_t n = zkhip_shard_verifier_describe(log_n, width, (size_t)prm.num_queries, prm.pow_bits, n_public, per_join, i, kind, nullptr, 0, &ln, &mw, &prw);
                if (n == 0) return false;
                dst.resize(n);
                zkhip_shard_verifier_describe(log_n, width, (size_t)prm.num_queries, prm.pow_bits, n_public, per_join, i, kind, dst.data(), n, &ln, &mw, &prw);
            }
            lns[i] = ln; widths[i] = mw; pres[i] = prw;
            pp[(size_t)i] = progs[(size_t)i].data(); tp[(size_t)i] = tabs[(size_t)i].data(); pw[(size_t)i] = progs[(size_t)i].size(); tw[(size_t)i] = tabs[(size_t)i].size();
        }
        desc.n_chips = 8; desc.log_ns = lns; desc.widths = widths; desc.pre_widths = pres;
        desc.programs = pp.data(); desc.program_words = pw.data(); desc.tables = tp.data(); desc.table_words = tw.data();
        for (int i = 0; i < 8; i++) desc.key_root[i] = join_key[i];
        desc.num_queries = prm.num_queries; desc.pow_bits = prm.pow_bits; desc.n_public = (uint32_t)(n_public * per_join);
        return true;
    }
};

int main(int argc, char** argv) {
    const int shards = argc > 1 ? std::atoi(argv[1]) : 8, per_join = argc > 2 ? std::atoi(argv[2]) : 4;
    const int log_n = argc > 3 ? std::atoi(argv[3]) : 12;
    const uint32_t width = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 16;
    if (zkhip_device_count() <= 0) { std::fprintf(stderr, "no gfx950 device: libzkhip has no CPU fallback\n"); return 2; }
    if (shards < 2 || per_join < 1 || shards % per_join || shards / per_join < 2 || shards / per_join > 64) { std::fprintf(stderr, "shards = joins x per_join with 2 .. 64 joins\n"); return 1; }
    const int n_joins = shards / per_join;
    const zkhip_params prm = {1, 100, 16, 0, 0, 0, 0, 0};
    const size_t n_public = 1;
    zkhip_ctx* ctx = nullptr;
    CHECK(zkhip_ctx_create(0, nullptr, &ctx));
    const size_t words = (size_t)width << log_n, cap = zkhip_proof_size(log_n, width, &prm, n_public);
    const size_t jcap = zkhip_shard_verifier_proof_size(log_n, width, (size_t)prm.num_queries, prm.pow_bits, n_public, (size_t)per_join, &prm);
    if (cap == 0 || jcap == 0) { std::fprintf(stderr, "bad shape: %s\n", zkhip_last_error()); return 1; }
    // 1. the shard proofs
    std::vector<void*> traces((size_t)shards);
    std::vector<std::vector<uint8_t>> proofs((size_t)shards, std::vector<uint8_t>(cap));
    std::vector<uint32_t> pvs((size_t)shards);
    std::vector<zkhip_shard_job> jobs((size_t)shards);
    for (int s = 0; s < shards; s++) {
        CHECK(zkhip_malloc(ctx, words * 4, &traces[(size_t)s]));
        CHECK(zkhip_gen_trace(ctx, 0x5A4B544C53ull, (uint64_t)s, log_n, width, (uint32_t*)traces[(size_t)s], width));
        pvs[(size_t)s] = 100u + (uint32_t)s;
        jobs[(size_t)s] = zkhip_shard_job{(const uint32_t*)traces[(size_t)s], width, log_n, width, &pvs[(size_t)s], n_public, proofs[(size_t)s].data(), cap, 0, 0};
    }
    CHECK(zkhip_ctx_sync(ctx));
    const double t0 = now_ms();
    CHECK(zkhip_prove_shards(0, jobs.data(), shards, &prm, 4, 0));
    const double t1 = now_ms();
    for (int s = 0; s < shards; s++) CHECK(zkhip_free(ctx, traces[(size_t)s]));
    // 2. level 1: the joins (one key for all of them)
    zkhip_machine_key* jkey = nullptr;
    uint32_t jvk[8];
    CHECK(zkhip_shard_verifier_setup(ctx, log_n, width, (size_t)prm.num_queries, prm.pow_bits, n_public, (size_t)per_join, &prm, &jkey, jvk));
    std::vector<std::vector<uint8_t>> joins((size_t)n_joins, std::vector<uint8_t>(jcap));
    std::vector<size_t> jlens((size_t)n_joins);
    size_t inner_total = 0, join_total = 0;
    const double t2 = now_ms();
    for (int j = 0; j < n_joins; j++) {
        std::vector<const uint8_t*> ptrs((size_t)per_join);
        std::vector<size_t> lens((size_t)per_join);
        for (int k = 0; k < per_join; k++) { ptrs[(size_t)k] = proofs[(size_t)(j * per_join + k)].data(); lens[(size_t)k] = jobs[(size_t)(j * per_join + k)].proof_len; inner_total += lens[(size_t)k]; }
        CHECK(zkhip_prove_shard_verifier(ctx, jkey, ptrs.data(), lens.data(), (size_t)per_join, log_n, width, pvs.data() + (size_t)j * (size_t)per_join, n_public, &prm, &prm,
                                         joins[(size_t)j].data(), jcap, &jlens[(size_t)j]));
        join_total += jlens[(size_t)j];
    }
    const double t3 = now_ms();
    // 3. level 2: ONE proof over the joins
    JoinMachine jm;
    if (!jm.build(log_n, width, prm, n_public, (size_t)per_join, jvk)) { std::fprintf(stderr, "describe: %s\n", zkhip_last_error()); return 1; }
    zkhip_machine_key* tkey = nullptr;
    uint32_t tvk[8];
    CHECK(zkhip_machine_verifier_setup(ctx, &jm.desc, (size_t)n_joins, &prm, &tkey, tvk));
    const size_t tcap = zkhip_machine_verifier_proof_size(&jm.desc, (size_t)n_joins, &prm);
    std::vector<uint8_t> top(tcap);
    std::vector<const uint8_t*> jp((size_t)n_joins);
    for (int j = 0; j < n_joins; j++) jp[(size_t)j] = joins[(size_t)j].data();
    size_t tlen = 0;
    const double t4 = now_ms();
    CHECK(zkhip_prove_machine_verifier(ctx, tkey, &jm.desc, jp.data(), jlens.data(), (size_t)n_joins, pvs.data(), n_public * (size_t)per_join, &prm, top.data(), tcap, &tlen));
    const double t5 = now_ms();
    // ... and the same tree in ONE call: the joins in flight on pooled contexts, each one's tables for the top filled the moment it exists
    {
        std::vector<const uint8_t*> ptrs((size_t)shards);
        std::vector<size_t> lens((size_t)shards), jl((size_t)n_joins);
        for (int s = 0; s < shards; s++) { ptrs[(size_t)s] = proofs[(size_t)s].data(); lens[(size_t)s] = jobs[(size_t)s].proof_len; }
        std::vector<uint8_t> jbuf((size_t)n_joins * jcap), top1(tcap);
        uint32_t vk1[8];
        size_t len1 = 0;
        const double o0 = now_ms();
        CHECK(zkhip_prove_shard_tree(ctx, tkey, &jm.desc, nullptr, 0, ptrs.data(), lens.data(), (size_t)shards, (size_t)per_join, log_n, width, pvs.data(), n_public, &prm, &prm, &prm, 0,
                                     jbuf.data(), jcap, jl.data(), vk1, top1.data(), tcap, &len1));
        const double o1 = now_ms();
        if (len1 != tlen || std::memcmp(top1.data(), top.data(), tlen) != 0 || std::memcmp(vk1, jvk, 32) != 0) { std::fprintf(stderr, "zkhip_prove_shard_tree made another proof\n"); return 3; }
        std::printf("one call (zkhip_prove_shard_tree, first call: the pooled contexts make the join key): %.1f ms, the same %zu bytes\n", o1 - o0, len1);
    }
    zkhip_machine_key_destroy(jkey);
    zkhip_machine_key_destroy(tkey);
    zkhip_ctx_destroy(ctx);
    zkhip_release_cached_contexts();
    // 4. from here on: host only; the shard proofs and the joins are gone, and both keys are derived here
    proofs.clear(); joins.clear();
    const double v0 = now_ms();
    uint32_t hjvk[8], htvk[8];
    CHECK(zkhip_shard_verifier_key_host(log_n, width, (size_t)prm.num_queries, prm.pow_bits, n_public, (size_t)per_join, &prm, hjvk));
    JoinMachine vm;
    if (!vm.build(log_n, width, prm, n_public, (size_t)per_join, hjvk)) return 1;
    CHECK(zkhip_machine_verifier_key_host(&vm.desc, (size_t)n_joins, &prm, htvk));
    const double v1 = now_ms();
    int reason = 0;
    CHECK(zkhip_verify_machine_recursive(&vm.desc, top.data(), tlen, pvs.data(), n_public * (size_t)per_join, (size_t)n_joins, htvk, &prm, &reason));
    const double v2 = now_ms();
    pvs[(size_t)shards - 1] ^= 1u;                                   // another statement must be refused
    if (zkhip_verify_machine_recursive(&vm.desc, top.data(), tlen, pvs.data(), n_public * (size_t)per_join, (size_t)n_joins, htvk, &prm, &reason) == ZKHIP_OK) {
        std::fprintf(stderr, "the tree's top was accepted for other public values\n");
        return 3;
    }
    std::printf("%d shards of 2^%d x %u: shard proofs %.1f ms (%zu bytes); %d joins of %d: %.1f ms (%zu bytes); ONE proof over the joins %.1f ms (%zu bytes = 1 / %.1f of the shard "
                "proofs); verified on the host in %.2f ms from (the join machine's description, %d public values, keys derived on the host in %.1f ms) alone\n",
                shards, log_n, width, t1 - t0, inner_total, n_joins, per_join, t3 - t2, join_total, t5 - t4, tlen, (double)inner_total / (double)tlen, v2 - v1, shards, v1 - v0);
    return 0;
}
