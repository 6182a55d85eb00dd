// examples/compress_shards.cpp -- the two batch calls of the path from plain C++: the shards of an execution proven in ONE call
// (zkhip_prove_shards: `client.prove`, crates/guest-prover-sp1/src/sp1.rs:116, core stage), then the FRI check of every shard proof
// proven in-circuit in ONE call (zkhip_prove_fri_indices_batch: the first piece of the compress stage behind the same line), every
// outer proof verified on the host with nothing but (vk, final value, challenger capacity) beside it.
//
//   make -C examples && ./examples/compress_shards [shards=8] [log_n=16] [width=64]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/zkhip.h"

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
    if (shards < 1 || shards > 4096) { std::fprintf(stderr, "1..4096 shards\n"); return 1; }
    const zkhip_params prm = {1, 100, 16, 0, 0, 0, 0, 0};            // SP1-core-like shape, for the shard proofs and for the proofs about them
    zkhip_ctx* ctx = nullptr;
    CHECK(zkhip_ctx_create(0, nullptr, &ctx));
    const size_t words = (size_t)width << log_n, cap = zkhip_proof_size(log_n, width, &prm, 1);
    const size_t rcap = zkhip_fri_indices_proof_size(log_n, (size_t)prm.num_queries, prm.pow_bits, &prm);
    if (cap == 0 || rcap == 0) { std::fprintf(stderr, "bad shape: %s\n", zkhip_last_error()); return 1; }
    std::vector<void*> traces((size_t)shards);
    std::vector<std::vector<uint8_t>> proofs((size_t)shards, std::vector<uint8_t>(cap)), outer((size_t)shards, std::vector<uint8_t>(rcap));
    std::vector<uint32_t> pvs((size_t)shards);
    std::vector<zkhip_shard_job> jobs((size_t)shards);
    for (int s = 0; s < shards; s++) {
        CHECK(zkhip_malloc(ctx, words * 4, &traces[(size_t)s]));
        CHECK(zkhip_gen_trace(ctx, 0x5A4B544C53ull, (uint64_t)s, log_n, width, (uint32_t*)traces[(size_t)s], width));
        pvs[(size_t)s] = (uint32_t)s;
        jobs[(size_t)s] = zkhip_shard_job{(const uint32_t*)traces[(size_t)s], width, log_n, width, &pvs[(size_t)s], 1, proofs[(size_t)s].data(), cap, 0, 0};
    }
    CHECK(zkhip_ctx_sync(ctx));
    double core = 1e30, compress = 1e30;
    for (int rep = 0; rep < 3; rep++) {                              // (the first call creates the internal contexts)
        const double t0 = now_ms();
        CHECK(zkhip_prove_shards(0, jobs.data(), shards, &prm, 4, 0));
        const double t1 = now_ms();
        if (t1 - t0 < core) core = t1 - t0;
    }
    std::vector<zkhip_fri_job> rjobs((size_t)shards);
    for (int s = 0; s < shards; s++) {
        zkhip_fri_job& j = rjobs[(size_t)s];
        j = zkhip_fri_job{};
        j.shard_proof = proofs[(size_t)s].data(); j.shard_proof_len = jobs[(size_t)s].proof_len;
        j.public_values = &pvs[(size_t)s]; j.n_public = 1;
        j.proof = outer[(size_t)s].data(); j.proof_cap = rcap;
    }
    const int device = 0;
    for (int rep = 0; rep < 3; rep++) {
        const double t0 = now_ms();
        CHECK(zkhip_prove_fri_indices_batch(&device, 1, rjobs.data(), shards, log_n, width, &prm, &prm, 4, 0));
        const double t1 = now_ms();
        if (t1 - t0 < compress) compress = t1 - t0;
    }
    const double v0 = now_ms();
    for (int s = 0; s < shards; s++) {
        const zkhip_fri_job& j = rjobs[(size_t)s];
        int reason = 0;
        CHECK(zkhip_verify_fri_indices(j.proof, j.proof_len, log_n, (size_t)prm.num_queries, prm.pow_bits, j.final_value, j.capacity, j.vk, &prm, &reason));
    }
    const double v1 = now_ms();
    std::printf("%d shards of 2^%d x %u: shard proofs %.1f ms (%.2f ms each, %zu bytes), their FRI checks in-circuit %.1f ms (%.2f ms each, %zu bytes), "
                "host verification of the outer proofs %.2f ms each\n",
                shards, log_n, width, core, core / shards, jobs[0].proof_len, compress, compress / shards, rjobs[0].proof_len, (v1 - v0) / shards);
    for (int s = 0; s < shards; s++) CHECK(zkhip_free(ctx, traces[(size_t)s]));
    zkhip_ctx_destroy(ctx);
    zkhip_release_cached_contexts();
    return 0;
}
