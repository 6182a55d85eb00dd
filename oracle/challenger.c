/*
 * oracle/challenger.c -- duplex-sponge Fiat-Shamir challenger, CPU restatement.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle/oracle.h).
 *
 * Restates p3-challenger 0.2.1-succinct DuplexChallenger<F, Perm, 16, 8>
 * (reference Cargo.lock:3875; reached from crates/guest-prover-sp1/src/sp1.rs:116):
 *   observe(x):   clear output buffer; push x; when 8 inputs are buffered -> duplex
 *   duplex:       overwrite state[0..n_in] with the inputs, permute,
 *                 output buffer = state[0..8]
 *   sample():     if inputs pending or output empty -> duplex; pop the LAST output
 *   sample_bits:  canonical value of sample() & (2^bits - 1)
 *   grind(bits):  witness w with (observe(w); sample_bits(bits) == 0).  Upstream
 *                 searches in parallel and takes any hit; this build fixes the
 *                 SMALLEST canonical w so proofs are reproducible.
 */
#include "oracle.h"
#include <string.h>

void orc_chal_init(orc_challenger_t* c) { memset(c, 0, sizeof *c); }

static void duplexing(orc_challenger_t* c) {
    for (int i = 0; i < c->n_input; i++) c->state[i] = c->input[i];
    c->n_input = 0;
    orc_poseidon2_permute(c->state);
    memcpy(c->output, c->state, 8 * sizeof(uint32_t));
    c->n_output = 8;
}

void orc_chal_observe(orc_challenger_t* c, uint32_t v) {
    c->n_output = 0;
    c->input[c->n_input++] = v;
    if (c->n_input == 8) duplexing(c);
}

void orc_chal_observe_slice(orc_challenger_t* c, const uint32_t* v, size_t n) {
    for (size_t i = 0; i < n; i++) orc_chal_observe(c, v[i]);
}

uint32_t orc_chal_sample(orc_challenger_t* c) {
    if (c->n_input != 0 || c->n_output == 0) duplexing(c);
    return c->output[--c->n_output];
}

void orc_chal_sample_ext(orc_challenger_t* c, uint32_t out[4]) {
    for (int i = 0; i < 4; i++) out[i] = orc_chal_sample(c);
}

uint32_t orc_chal_sample_bits(orc_challenger_t* c, int bits) {
    uint32_t v = orc_chal_sample(c);
    return v & (((uint32_t)1 << bits) - 1);
}

int orc_chal_check_witness(orc_challenger_t* c, int bits, uint32_t witness) {
    orc_chal_observe(c, witness);
    return orc_chal_sample_bits(c, bits) == 0;
}

uint32_t orc_chal_grind(orc_challenger_t* c, int bits) {
    for (uint32_t w = 0; w < BB_P; w++) {
        orc_challenger_t t = *c;
        if (orc_chal_check_witness(&t, bits, w)) { *c = t; return w; }
    }
    return 0xFFFFFFFFu; /* unreachable for bits <= 30 in practice */
}
