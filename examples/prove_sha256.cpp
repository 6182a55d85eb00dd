// examples/prove_sha256.cpp -- "I know a file with this SHA-256 digest", proven on the GPU through the SHA-256 compression chip
// (include/zkhip.h, zkhip_prove_sha256) and checked on the host -- plain C++ over the C ABI, no Python, no torch.
//
//   make -C examples && ./examples/prove_sha256 <file> [proof-out]
//   ./examples/prove_sha256 --verify <proof> <hex digest> <message length in bytes>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
    const zkhip_params prm = ZKHIP_PARAMS_SP1_CORE;                        // blowup 2, 100 queries, 16 PoW bits
    if (argc == 5 && !std::strcmp(argv[1], "--verify")) {            // host only: no GPU needed to check a proof
        std::vector<uint8_t> proof;
        if (!read_file(argv[2], proof)) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
        uint8_t digest[32];
        if (std::strlen(argv[3]) != 64) { std::fprintf(stderr, "digest: 64 hex characters\n"); return 1; }
        for (int i = 0; i < 32; i++) { unsigned v; if (std::sscanf(argv[3] + 2 * i, "%2x", &v) != 1) return 1; digest[i] = (uint8_t)v; }
        int reason = 0;
        const uint64_t message_len = std::strtoull(argv[4], nullptr, 10);      // the statement: digest = SHA-256 of a message of THIS many bytes
        const int rc = zkhip_verify_sha256(proof.data(), proof.size(), digest, message_len, &prm, &reason);
        std::printf("%s\n", rc == ZKHIP_OK ? "proof accepted" : zkhip_last_error());
        return rc == ZKHIP_OK ? 0 : 3;
    }
    if (argc < 2) { std::fprintf(stderr, "usage: %s <file> [proof-out] | --verify <proof> <hex digest> <message length in bytes>\n", argv[0]); return 1; }
    std::vector<uint8_t> msg;
    if (!read_file(argv[1], msg)) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    if (zkhip_device_count() <= 0) { std::fprintf(stderr, "no gfx950 device: libzkhip has no CPU fallback\n"); return 2; }
    zkhip_ctx* ctx = nullptr;
    if (zkhip_ctx_create(0, nullptr, &ctx) != ZKHIP_OK) { std::fprintf(stderr, "%s\n", zkhip_last_error()); return 1; }
    const size_t cap = zkhip_sha256_proof_size(msg.size(), &prm);
    if (cap == 0) { std::fprintf(stderr, "file too long for one proof: %s\n", zkhip_last_error()); return 1; }
    std::vector<uint8_t> proof(cap);
    uint8_t digest[32], expect[32];
    size_t len = 0;
    for (int round = 0; round < 2; round++) {                         // the first call builds plans and workspaces
        const auto t0 = std::chrono::steady_clock::now();
        if (zkhip_prove_sha256(ctx, msg.data(), msg.size(), &prm, digest, proof.data(), cap, &len) != ZKHIP_OK) { std::fprintf(stderr, "%s\n", zkhip_last_error()); return 1; }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%s: %zu bytes proven in %.1f ms, proof %zu bytes\n", round ? "again" : "first call", msg.size(), ms, len);
    }
    zkhip_sha256_digest(msg.data(), msg.size(), expect);
    int reason = 0;
    if (std::memcmp(digest, expect, 32) != 0 || zkhip_verify_sha256(proof.data(), len, digest, (uint64_t)msg.size(), &prm, &reason) != ZKHIP_OK) { std::fprintf(stderr, "self-check failed\n"); return 1; }
    std::printf("sha256 = ");
    for (int i = 0; i < 32; i++) std::printf("%02x", digest[i]);
    std::printf("  (proof verified)\n");
    if (argc > 2) { FILE* f = std::fopen(argv[2], "wb"); if (f) { std::fwrite(proof.data(), 1, len, f); std::fclose(f); } }
    zkhip_ctx_destroy(ctx);
    return 0;
}
