/* oracle/stark_internal.h -- helpers shared by stark.c and chips.c.  TEST INFRASTRUCTURE ONLY (see oracle.h). */
#ifndef ORACLE_STARK_INTERNAL_H
#define ORACLE_STARK_INTERNAL_H
#include "oracle.h"
bb4_t orc__ld4(const uint32_t* p);
void orc__st4(uint32_t* p, bb4_t v);
/* the synthetic AIR's constraints on extension-field rows, folded with alpha (verifier side) */
bb4_t orc__fold_constraints_ext(const bb4_t* local, const bb4_t* next, size_t width, bb4_t sel_first, bb4_t sel_trans, bb4_t alpha);
bb4_t orc__fri_fold_row(size_t index, int log_folded_h, bb4_t beta, bb4_t e0, bb4_t e1);
bb4_t orc__row_dot(const bb4_t* pw, const uint32_t* row, size_t w);
void orc__copy_path(uint32_t* pf, size_t* pos, const uint32_t* tree, size_t leaves, size_t index, int levels);
bb4_t orc__recombine(const uint32_t* opened4);
bb4_t orc__fold_logup(bb4_t acc, int pairs, const bb4_t* as, const bb4_t* bs, const bb4_t* ar, const bb4_t* br,
                     const bb4_t* perm_local, const bb4_t* perm_next, bb4_t gamma, bb4_t beta,
                     bb4_t sel_first, bb4_t sel_trans, bb4_t sel_last, bb4_t alpha, bb4_t cumsum);
/* constraint programs (oracle/air.c) */
bb4_t orc__air_fold_base(const uint32_t* prog, const uint32_t* local, const uint32_t* next, const uint32_t* pub,
                         bb_t sel_first, bb_t sel_last, bb_t sel_trans, bb4_t alpha);
bb4_t orc__air_fold_ext(const uint32_t* prog, const bb4_t* local, const bb4_t* next, const uint32_t* pub,
                        bb4_t sel_first, bb4_t sel_last, bb4_t sel_trans, bb4_t alpha);
#endif
