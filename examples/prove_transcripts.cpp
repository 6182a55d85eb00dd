// examples/prove_transcripts.cpp -- setup -> prove -> verify as the reference calls them (crates/guest-prover-sp1/src/sp1.rs:113, :116,
// :120), for a BATCH of transcripts in one call: every file named on the command line is proven as the keyed SHA-256 machine
// (compression chip + range table with preprocessed values) on the GPUs of this node, and every proof is then checked on the host
// against the one verifying key.  Plain C++ over the C ABI (include/zkhip.h), no Python, no torch.
//
//   make -C examples && ./examples/prove_transcripts <file> [<file> ...]
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/zkhip.h"
#include "../include/zkhip_chips.h"

static bool read_file(const char* path, std::vector<uint8_t>& out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) out.insert(out.end(), buf, buf + n);
    std::fclose(f);
    return true;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <file> [<file> ...]\n", argv[0]); return 1; }
    if (zkhip_device_count() <= 0) { std::fprintf(stderr, "no gfx950 device: libzkhip has no CPU fallback\n"); return 2; }
    const zkhip_params prm = ZKHIP_PARAMS_SP1_CORE;
    const int n = argc - 1;
    std::vector<std::vector<uint8_t>> msgs(n), proofs(n);
    std::vector<zkhip_transcript_job> jobs(n);
    for (int i = 0; i < n; i++) {
        if (!read_file(argv[1 + i], msgs[i])) { std::fprintf(stderr, "cannot read %s\n", argv[1 + i]); return 1; }
        const size_t cap = zkhip_sha256_machine_proof_size(msgs[i].size(), &prm);
        if (cap == 0) { std::fprintf(stderr, "%s: too long for one proof (1 MiB)\n", argv[1 + i]); return 1; }
        proofs[i].resize(cap);
        std::memset(&jobs[i], 0, sizeof jobs[i]);
        jobs[i].message = msgs[i].data(); jobs[i].message_len = msgs[i].size();
        jobs[i].proof = proofs[i].data(); jobs[i].proof_cap = cap;
    }
    uint32_t vk[8];
    for (int round = 0; round < 2; round++) {                         // the first call creates contexts, plans and the proving keys
        const auto t0 = std::chrono::steady_clock::now();
        // NULL / 0: every visible GPU; job i runs on device i mod n_devices, four in flight per device
        if (zkhip_prove_transcripts(nullptr, 0, jobs.data(), n, &prm, 4, /*verify=*/0, vk) != ZKHIP_OK) { std::fprintf(stderr, "%s\n", zkhip_last_error()); return 1; }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%s: %d transcripts proven in %.1f ms (%.2f ms each)\n", round ? "again" : "first call", n, ms, ms / n);
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) {
        uint8_t expect[32];
        zkhip_sha256_digest(msgs[i].data(), msgs[i].size(), expect);
        int reason = 0;
        if (std::memcmp(expect, jobs[i].digest, 32) != 0 ||
            zkhip_verify_sha256_machine(jobs[i].proof, jobs[i].proof_len, jobs[i].digest, (uint64_t)jobs[i].message_len, vk, &prm, &reason) != ZKHIP_OK) {
            std::fprintf(stderr, "%s: check failed (%s)\n", argv[1 + i], zkhip_last_error());
            return 3;
        }
    }
    std::printf("all %d proofs verified against the key on the host in %.1f ms; vk =", n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    for (int i = 0; i < 8; i++) std::printf(" %08x", vk[i]);
    std::printf("\n");
    zkhip_release_cached_contexts();
    return 0;
}
