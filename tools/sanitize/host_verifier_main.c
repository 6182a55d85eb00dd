#include "oracle.h"
#include "zkhip.h"
#include "zkhip_chips.h"
#include <stdio.h>
#include <stdlib.h>
int main(void) {
    int fails = 0;
    orc_params_t shapes[] = {{1, 10, 4, 0, 0, 0, 0}, {2, 10, 0, 0, 4, 2, 24}, {1, 6, 4, 2, 0, 0, 0}, {3, 5, 0, 1, 2, 4, 16}};
    for (int s = 0; s < 4; s++) {
        int log_n = 10; size_t w = 32;
        uint32_t* t = malloc(((size_t)1 << log_n) * w * 4);
        if (shapes[s].logup_pairs) orc_gen_trace_logup(7, 0, log_n, w, shapes[s].logup_pairs, t); else orc_gen_trace(7, 0, log_n, w, t);
        uint32_t pv[3] = {1, 2, 3};
        size_t sz = orc_proof_size(log_n, w, &shapes[s], 3);
        uint8_t* pf = malloc(sz);
        orc_prove_shard(t, log_n, w, pv, 3, &shapes[s], pf, sz);
        zkhip_params zp = {shapes[s].log_blowup, shapes[s].num_queries, shapes[s].pow_bits, shapes[s].logup_pairs, shapes[s].log_fold, shapes[s].log_final, shapes[s].hash_width};
        int reason = 0;
        int rc = zkhip_verify_shard(pf, sz, log_n, (uint32_t)w, pv, 3, &zp, &reason);
        int bad = 0;
        for (size_t off = 64; off < sz; off += sz / 23) { pf[off] ^= 1; int r2 = 0; if (zkhip_verify_shard(pf, sz, log_n, (uint32_t)w, pv, 3, &zp, &r2) == 0) bad++; pf[off] ^= 1; }
        printf("shape %d: host verifier rc %d reason %d, undetected corruptions %d\n", s, rc, reason, bad);
        if (rc != 0 || bad) fails++;
        free(t); free(pf);
    }
    int lns[4] = {10, 8, 8, 5}; size_t ws[4] = {16, 8, 12, 4}; int32_t l32[4] = {10, 8, 8, 5}; uint32_t w32[4] = {16, 8, 12, 4};
    const uint32_t* tr[4];
    for (int c = 0; c < 4; c++) { uint32_t* t = malloc(((size_t)1 << lns[c]) * ws[c] * 4); orc_gen_trace(9, c, lns[c], ws[c], t); tr[c] = t; }
    orc_params_t p = {2, 8, 4, 0, 0, 0, 0};
    zkhip_params zp = {2, 8, 4, 0, 0, 0, 0};
    size_t sz = orc_chips_proof_size(lns, ws, NULL, NULL, 4, &p, 0);
    uint8_t* pf = malloc(sz);
    orc_prove_chips(tr, lns, ws, NULL, NULL, 4, NULL, 0, &p, pf, sz);
    int reason = 0;
    int rc = zkhip_verify_chips(pf, sz, l32, w32, NULL, NULL, 4, NULL, 0, &zp, &reason);
    printf("chips: host verifier rc %d reason %d\n", rc, reason);
    if (rc) fails++;
    for (int c = 0; c < 4; c++) free((void*)tr[c]);
    free(pf);
    return fails;
}
