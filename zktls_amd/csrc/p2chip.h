// p2chip.h -- column layout of the Poseidon2 permutation chip (poseidon2_chip.cpp: the constraint program and the C entries;
// hash.hip: the on-device trace generator).  One row = one width-16 permutation; see poseidon2_chip.cpp for what the columns mean.
#pragma once
#include <cstdint>

namespace zk {
namespace p2chip {

constexpr uint32_t IN = 0, S0 = 16, SP = 327, D = 343, BIT = 351, CH = 352, END = 353, CNT = 354, SPG = 355, SS = 356,
                   WIDTH = 360, N_PUBLIC = 9;
constexpr uint32_t x3e(uint32_t r) { return 32 + 32 * r; }      // external round r: cubes of the round's input + constant
constexpr uint32_t oute(uint32_t r) { return 48 + 32 * r; }     // ... and the state after the round
constexpr uint32_t s0p(uint32_t r) { return 288 + 3 * r; }      // internal round r: element 0 before the S-box,
constexpr uint32_t x3p(uint32_t r) { return 289 + 3 * r; }      // its cube,
constexpr uint32_t sbp(uint32_t r) { return 290 + 3 * r; }      // its seventh power
constexpr uint32_t ext_input(uint32_t r) { return r == 0 ? S0 : (r == 4 ? SP : oute(r - 1)); }

// paths: path p = rows [p (row_width / 8 + depth), ...): row_width / 8 sponge rows over its opened row (none when row_width = 0: the
// leaf digest is given), then depth compression rows; leaves / siblings / indices are canonical words already on the device
struct MerkleTraceArgs {
    const uint32_t* leaves;      // [n_paths][8] digests, or [n_paths][row_width] opened rows
    uint32_t row_width;          // 0, or a multiple of 8
    const uint32_t* siblings;    // [n_paths][depth][8]
    const uint32_t* indices;     // [n_paths]: bit l = the node is a right child at level l
    uint64_t n_paths, rows;
    uint32_t depth;
    uint32_t* trace; uint64_t ld;   // [rows][ld], Montgomery
    uint32_t* roots;             // [n_paths][8], canonical
};

}  // namespace p2chip
}  // namespace zk
