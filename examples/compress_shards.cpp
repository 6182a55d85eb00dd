// examples/compress_shards.cpp -- core -> compress from plain C++ (crates/guest-prover-sp1/src/sp1.rs:116: `client.prove(.., Groth16)` runs
// core, then COMPRESS verifies the shard proofs inside a proof):
//   1. the shards of an execution proven in ONE call (zkhip_prove_shards);
//   2. the key of the shard-verifier machine for that SHAPE (zkhip_shard_verifier_setup: no shard proof is involved -- the key could be
//      published once per shape);
//   3. ALL shard proofs verified -- transcript, AIR identity, every Merkle opening, reduced openings, FRI, proof of work -- inside ONE
//      outer proof (zkhip_prove_shard_verifier, the join);
//   4. that proof checked on the host with the shape, the shards' public values and the key: the shard proofs themselves are not needed
//      any more (zkhip_verify_shard_recursive takes no byte of them).
//
//   make -C examples && ./examples/compress_shards [shards=8] [log_n=16] [width=64]
#include <chrono>
#include <cstdio>
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

int main(int argc, char** argv) {
    const int shards = argc > 1 ? std::atoi(argv[1]) : 8;
    const int log_n = argc > 2 ? std::atoi(argv[2]) : 16;
    const uint32_t width = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 64;
    if (zkhip_device_count() <= 0) {
        std::fprintf(stderr, "no gfx950 device: libzkhip has no CPU fallback\n");
        return 2;
    }
    const zkhip_params shape = {1, 100, 16, 0, 0, 0, 0, 0};
    const size_t most = zkhip_shard_verifier_max_proofs(log_n, width, (size_t)shape.num_queries, shape.pow_bits, 1, &shape);
    if (shards < 1 || (size_t)shards > most) { std::fprintf(stderr, "1..%zu shards of this shape fit one join\n", most); return 1; }
    const zkhip_params prm = {1, 100, 16, 0, 0, 0, 0, 0};            // SP1-core-like shape, for the shard proofs and for the proof about them
    const size_t n_public = 1;
    zkhip_ctx* ctx = nullptr;
    CHECK(zkhip_ctx_create(0, nullptr, &ctx));
    const size_t words = (size_t)width << log_n, cap = zkhip_proof_size(log_n, width, &prm, n_public);
    const size_t jcap = zkhip_shard_verifier_proof_size(log_n, width, (size_t)prm.num_queries, prm.pow_bits, n_public, (size_t)shards, &prm);
    if (cap == 0 || jcap == 0) { std::fprintf(stderr, "bad shape: %s\n", zkhip_last_error()); return 1; }
    std::vector<void*> traces((size_t)shards);
    std::vector<std::vector<uint8_t>> proofs((size_t)shards, std::vector<uint8_t>(cap));
    std::vector<uint8_t> joined(jcap);
    std::vector<uint32_t> pvs((size_t)shards);
    std::vector<zkhip_shard_job> jobs((size_t)shards);
    for (int s = 0; s < shards; s++) {
        CHECK(zkhip_malloc(ctx, words * 4, &traces[(size_t)s]));
        CHECK(zkhip_gen_trace(ctx, 0x5A4B544C53ull, (uint64_t)s, log_n, width, (uint32_t*)traces[(size_t)s], width));
        pvs[(size_t)s] = (uint32_t)s;
        jobs[(size_t)s] = zkhip_shard_job{(const uint32_t*)traces[(size_t)s], width, log_n, width, &pvs[(size_t)s], n_public, proofs[(size_t)s].data(), cap, 0, 0};
    }
    CHECK(zkhip_ctx_sync(ctx));
    double core = 1e30, compress = 1e30;
    for (int rep = 0; rep < 3; rep++) {                              // (the first call creates the internal contexts)
        const double t0 = now_ms();
        CHECK(zkhip_prove_shards(0, jobs.data(), shards, &prm, 4, 0));
        const double t1 = now_ms();
        if (t1 - t0 < core) core = t1 - t0;
    }
    for (int s = 0; s < shards; s++) CHECK(zkhip_free(ctx, traces[(size_t)s]));
    // the key: a function of (log_n, width, queries, proof-of-work bits, public values per proof, proofs per join)
    zkhip_machine_key* key = nullptr;
    uint32_t vk[8];
    const double s0 = now_ms();
    CHECK(zkhip_shard_verifier_setup(ctx, log_n, width, (size_t)prm.num_queries, prm.pow_bits, n_public, (size_t)shards, &prm, &key, vk));
    CHECK(zkhip_ctx_sync(ctx));
    const double setup = now_ms() - s0;
    std::vector<const uint8_t*> ptrs((size_t)shards);
    std::vector<size_t> lens((size_t)shards);
    size_t inner_total = 0;
    for (int s = 0; s < shards; s++) { ptrs[(size_t)s] = proofs[(size_t)s].data(); lens[(size_t)s] = jobs[(size_t)s].proof_len; inner_total += lens[(size_t)s]; }
    size_t jlen = 0;
    for (int rep = 0; rep < 3; rep++) {
        const double t0 = now_ms();
        CHECK(zkhip_prove_shard_verifier(ctx, key, ptrs.data(), lens.data(), (size_t)shards, log_n, width, pvs.data(), n_public, &prm, &prm, joined.data(), jcap, &jlen));
        const double t1 = now_ms();
        if (t1 - t0 < compress) compress = t1 - t0;
    }
    zkhip_machine_key_destroy(key);
    zkhip_ctx_destroy(ctx);
    zkhip_release_cached_contexts();
    // from here on: host only, and the shard proofs are gone
    proofs.clear();
    const double v0 = now_ms();
    int reason = 0;
    CHECK(zkhip_verify_shard_recursive(joined.data(), jlen, log_n, width, (size_t)prm.num_queries, prm.pow_bits, pvs.data(), n_public, (size_t)shards, vk, &prm, &reason));
    const double v1 = now_ms();
    pvs[0] ^= 1u;                                                    // another statement must be refused
    if (zkhip_verify_shard_recursive(joined.data(), jlen, log_n, width, (size_t)prm.num_queries, prm.pow_bits, pvs.data(), n_public, (size_t)shards, vk, &prm, &reason) == ZKHIP_OK) {
        std::fprintf(stderr, "the joined proof was accepted for other public values\n");
        return 3;
    }
    std::printf("%d shards of 2^%d x %u: shard proofs %.1f ms (%.2f ms each, %zu bytes in all); key of the shape %.1f ms; ONE proof that verifies them all %.1f ms "
                "(%.2f ms per shard proof, %zu bytes = 1 / %.1f); verified on the host in %.2f ms from (shape, %d public values, key) alone\n",
                shards, log_n, width, core, core / shards, inner_total, setup, compress, compress / shards, jlen, (double)inner_total / (double)jlen, v1 - v0, shards);
    return 0;
}
