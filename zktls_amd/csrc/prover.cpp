// prover.cpp -- STARK stages and the whole-shard prover (host orchestration).
#include "context.h"

using namespace zk;

extern "C" {

int zkhip_quotient_values(zkhip_ctx*, const uint32_t*, size_t, int, uint32_t, const uint32_t*, uint32_t*) {
    return fail(ZKHIP_ERR_INTERNAL, "quotient_values: not implemented yet");
}
int zkhip_open_at(zkhip_ctx*, const uint32_t*, size_t, int, int, uint32_t, const uint32_t*, int, uint32_t*) {
    return fail(ZKHIP_ERR_INTERNAL, "open_at: not implemented yet");
}
int zkhip_fri_fold(zkhip_ctx*, const uint32_t*, int, const uint32_t*, uint32_t*) {
    return fail(ZKHIP_ERR_INTERNAL, "fri_fold: not implemented yet");
}
size_t zkhip_proof_size(int, uint32_t, const zkhip_params*, size_t) { return 0; }
int zkhip_prove_shard(zkhip_ctx*, const uint32_t*, size_t, int, uint32_t, const uint32_t*, size_t,
                      const zkhip_params*, uint8_t*, size_t, size_t*) {
    return fail(ZKHIP_ERR_INTERNAL, "prove_shard: not implemented yet");
}
int zkhip_verify_shard(const uint8_t*, size_t, int, uint32_t, const uint32_t*, size_t, const zkhip_params*, int*) {
    return fail(ZKHIP_ERR_INTERNAL, "verify_shard: not implemented yet");
}
int zkhip_last_prove_debug(zkhip_ctx* ctx, zkhip_prove_debug* out) {
    if (!ctx || !out) return fail(ZKHIP_ERR_INVALID, "null argument");
    *out = ctx->debug;
    return ZKHIP_OK;
}

}  // extern "C"
