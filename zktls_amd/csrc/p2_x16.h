// p2_x16.h -- sixteen width-16 Poseidon2 permutations at once on the host (p2_x16.cpp: one state per AVX-512 lane)
#pragma once
#include <stdint.h>

namespace zk {
bool p2x16_available();                       // the CPU has AVX-512 F + DQ (checked at run time)
bool p2x16_enable(bool on);                   // tests: switch the batched form off / on; returns the previous setting
void p2x16_permute(uint32_t st[16][16]);      // st[element][lane]: 16 states, canonical Montgomery residues; only call when available
void p2x16_to_monty(uint32_t v[16]);          // 16 canonical words -> Montgomery form
bool p2h_permute(uint32_t st[16]);            // ONE permutation with the state in one AVX-512 register (what a serial sponge chain can use); false: not available, nothing done
}  // namespace zk
