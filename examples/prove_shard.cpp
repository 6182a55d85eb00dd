// examples/prove_shard.cpp -- the C ABI (include/zkhip.h) used from plain C++, no Python, no torch:
// generate a synthetic shard on the device, prove it, verify the proof on the host, print the timing.
//
//   make -C examples && ./examples/prove_shard [log_n=16] [width=64] [proofs=5]
//
// This is what the Rust glue of INTEGRATION.md does through FFI.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
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

int main(int argc, char** argv) {
    const int log_n = argc > 1 ? std::atoi(argv[1]) : 16;
    const uint32_t width = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 64;
    const int proofs = argc > 3 ? std::atoi(argv[3]) : 5;
    if (zkhip_device_count() <= 0) {
        std::fprintf(stderr, "no gfx950 device: libzkhip has no CPU fallback\n");
        return 2;
    }
    zkhip_ctx* ctx = nullptr;
    CHECK(zkhip_ctx_create(0, nullptr, &ctx));
    void* d_trace = nullptr;
    const size_t words = (size_t)width << log_n;
    CHECK(zkhip_malloc(ctx, words * 4, &d_trace));
    const zkhip_params prm = {1, 100, 16, 0, 0, 0, 0, 0};            // SP1-core-like shape
    const size_t cap = zkhip_proof_size(log_n, width, &prm, 1);
    if (cap == 0) { std::fprintf(stderr, "bad shape: %s\n", zkhip_last_error()); return 1; }
    std::vector<uint8_t> proof(cap);
    double total_ms = 0;
    for (int s = 0; s < proofs; s++) {
        CHECK(zkhip_gen_trace(ctx, 0x5A4B544C53ull, (uint64_t)s, log_n, width, (uint32_t*)d_trace, width));
        CHECK(zkhip_ctx_sync(ctx));
        const uint32_t pv[1] = {(uint32_t)s};
        size_t len = 0;
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(zkhip_prove_shard(ctx, (const uint32_t*)d_trace, width, log_n, width, pv, 1, &prm, proof.data(), cap, &len));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        int reason = 0;
        CHECK(zkhip_verify_shard(proof.data(), len, log_n, width, pv, 1, &prm, &reason));
        std::printf("shard %d: 2^%d x %u, %zu proof bytes, %.2f ms%s\n", s, log_n, width, len, ms, s == 0 ? " (first call builds plans and workspaces)" : "");
        if (s) total_ms += ms;
    }
    if (proofs > 1) std::printf("mean %.2f ms per proof, %.2f G trace cells/s (one shard in flight)\n", total_ms / (proofs - 1),
                                (double)words / (total_ms / (proofs - 1)) / 1e6);
    // ---- the same shards as ONE call with four of them in flight (zkhip_prove_shards): argv[4] = "batch"
    if (argc > 4 && std::string(argv[4]) == "batch" && proofs > 1) {
        std::vector<void*> traces(proofs);
        std::vector<std::vector<uint8_t>> bufs(proofs, std::vector<uint8_t>(cap));
        std::vector<uint32_t> pvs(proofs);
        std::vector<zkhip_shard_job> jobs(proofs);
        for (int s = 0; s < proofs; s++) {
            CHECK(zkhip_malloc(ctx, words * 4, &traces[s]));
            CHECK(zkhip_gen_trace(ctx, 0x5A4B544C53ull, (uint64_t)s, log_n, width, (uint32_t*)traces[s], width));
            pvs[s] = (uint32_t)s;
            jobs[s] = zkhip_shard_job{(const uint32_t*)traces[s], width, log_n, width, &pvs[s], 1, bufs[s].data(), cap, 0, 0};
        }
        CHECK(zkhip_ctx_sync(ctx));
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {                 // the first call creates the internal contexts
            const auto t0 = std::chrono::steady_clock::now();
            CHECK(zkhip_prove_shards(0, jobs.data(), proofs, &prm, 4, 0));
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms < best) best = ms;
        }
        for (int s = 0; s < proofs; s++) {
            int reason = 0;
            CHECK(zkhip_verify_shard(bufs[s].data(), jobs[s].proof_len, log_n, width, &pvs[s], 1, &prm, &reason));
            CHECK(zkhip_free(ctx, traces[s]));
        }
        std::printf("batch of %d, four in flight: %.2f ms per proof, %.2f G trace cells/s\n", proofs, best / proofs, (double)words * proofs / best / 1e6);
        zkhip_release_cached_contexts();
    }
    CHECK(zkhip_free(ctx, d_trace));
    zkhip_ctx_destroy(ctx);
    return 0;
}
