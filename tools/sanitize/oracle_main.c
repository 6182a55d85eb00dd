#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
int main(void) {
    int fails = 0;
    /* single matrix, several shapes */
    orc_params_t shapes[] = {{1, 10, 4, 0, 0, 0, 0}, {2, 10, 0, 0, 4, 2, 24}, {1, 6, 4, 2, 0, 0, 0}, {3, 5, 0, 1, 2, 4, 16}};
    for (int s = 0; s < 4; s++) {
        int log_n = 10; size_t w = 32;
        uint32_t* t = malloc(((size_t)1 << log_n) * w * 4);
        if (shapes[s].logup_pairs) orc_gen_trace_logup(7, 0, log_n, w, shapes[s].logup_pairs, t); else orc_gen_trace(7, 0, log_n, w, t);
        uint32_t pv[3] = {1, 2, 3};
        size_t sz = orc_proof_size(log_n, w, &shapes[s], 3);
        uint8_t* pf = malloc(sz);
        size_t got = orc_prove_shard(t, log_n, w, pv, 3, &shapes[s], pf, sz);
        int rc = orc_verify_shard(pf, got, log_n, w, pv, 3, &shapes[s]);
        pf[sz / 2] ^= 1;
        int rc2 = orc_verify_shard(pf, got, log_n, w, pv, 3, &shapes[s]);
        printf("shape %d: size %zu got %zu verify %d corrupt %d\n", s, sz, got, rc, rc2);
        if (got != sz || rc != 0 || rc2 == 0) fails++;
        free(t); free(pf);
    }
    /* chips */
    int lns[4] = {10, 8, 8, 5}; size_t ws[4] = {16, 8, 12, 4};
    const uint32_t* tr[4];
    for (int c = 0; c < 4; c++) { uint32_t* t = malloc(((size_t)1 << lns[c]) * ws[c] * 4); orc_gen_trace(9, c, lns[c], ws[c], t); tr[c] = t; }
    orc_params_t p = {2, 8, 4, 0, 0, 0, 0};
    size_t sz = orc_chips_proof_size(lns, ws, NULL, NULL, 4, &p, 0);
    uint8_t* pf = malloc(sz);
    size_t got = orc_prove_chips(tr, lns, ws, NULL, NULL, 4, NULL, 0, &p, pf, sz);
    int rc = orc_verify_chips(pf, got, lns, ws, NULL, NULL, 4, NULL, 0, &p);
    printf("chips: size %zu got %zu verify %d\n", sz, got, rc);
    if (got != sz || rc) fails++;
    for (int c = 0; c < 4; c++) free((void*)tr[c]);
    free(pf);
    return fails;
}
