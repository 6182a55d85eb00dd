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

// The FRI-layers variant of the chip (fri_chip.hip wires it to the FRI-fold chip): the three spare columns carry
//   LNP  the path's layer number (constant along a path),
//   KP   on the leaf row twice the leaf's index; every later row of the path: KP = 2 KP' + BIT walks the index down bit by bit,
//   M    how many times the leaf's tuple is received from the bus (non-zero on leaf rows only),
// paths of DIFFERENT depths end in different roots: an END row sends (LNP, digest) to a preprocessed table of layer roots instead
// of comparing with one public root.  Public values: those of the machine it sits in (none of its own).
constexpr uint32_t LNP = 357, KP = 358, M = 359;
// The transcript variant of that variant (one more flag column, 364 columns): the trace STARTS with transcript rows -- TRS = 1,
// LNP = 0, 1, 2, ... -- a sponge chain over the FRI layer roots: row l absorbs root_l into the rate half (and sends it to the ROOTS
// table like a path's digest), keeps the capacity of row l - 1 (row 0: eight public values, the duplex challenger's capacity as the
// commit phase finds it) and sends (l, out[7], out[6], out[5], out[4]) = the challenge beta_l on a bus of its own.
constexpr uint32_t TRS = 360, WIDTH_T = 364;
// ... and its query-phase rows (the machine of zkhip_prove_fri_indices): behind the chain the duplex challenger absorbs the final
// value and the proof-of-work witness (row QF: rate words 0..4, the rest of the previous output kept) and then only permutes (QP rows:
// the whole previous output).  The rate outputs of these rows are the sampled words, out[7] first: the proof-of-work word, then one
// word per query index; they go to the SAMPLES chip (fri_chip.hip) in two halves.
constexpr uint32_t QP = 361, QF = 362;
// variable-depth paths, one leaf row (the sponge over 8 values) + depth compression rows each; path p starts at row start[p]
struct LayerPathsArgs {
    const uint32_t* leaves;      // [n_paths][8] the opened pairs, canonical
    const uint32_t* siblings;    // concatenated per path: depth[p] x 8 words
    const uint32_t* sib_off;     // [n_paths] word offset of path p's siblings
    const uint32_t* indices;     // [n_paths] leaf index (bit l = right child at level l)
    const uint32_t* depths;      // [n_paths]
    const uint32_t* layers;      // [n_paths] layer number
    const uint32_t* mults;       // [n_paths] receive multiplicity
    const uint32_t* starts;      // [n_paths] first row
    uint64_t n_paths, rows, used_rows;
    uint32_t* trace; uint64_t ld;   // [rows][ld], Montgomery
    uint32_t* roots;             // [n_paths][8], canonical
    // transcript variant (n_transcript > 0: rows 0 .. n_transcript - 1, the paths' starts lie behind them; ld >= WIDTH_T) and its query-phase
    // rows (n_query_rows > 0: rows n_transcript .. n_transcript + n_query_rows - 1).  The sponge chain is a chain: its states are walked on
    // the host (a few dozen permutations) and handed over as the INPUT state of every row, so that the rows are filled side by side
    uint32_t n_transcript;       // layers
    uint32_t n_query_rows;
    const uint32_t* chain_inputs;   // [n_transcript + n_query_rows][16] canonical
};

// The shard verifier's chip (fri_chip.hip / shard_verifier.inl: P2R): the same 352 permutation columns (IN .. BIT), KP in column 352, nothing else
// in the main trace (every flag is a PREPROCESSED column there).  Rows are filled by CHAINS: chain c starts at row desc[6 c], hashes
// desc[6 c + 1] blocks of 8 words (data + desc[6 c + 2]) with the overwrite-mode sponge (KP = 2 index on the last of them), then walks
// desc[6 c + 3] path levels with the leaf index desc[6 c + 4] and the siblings at data + desc[6 c + 5]; n_transcript rows (trows) are the
// transcript's sponge rows, filled side by side from the input states the host walked.
constexpr uint32_t R_KP = 352, R_WIDTH = 360;
struct P2RArgs {
    const uint32_t* desc;            // [n_chains][6]
    const uint32_t* data;            // canonical words
    const uint32_t* chain_inputs;    // [n_transcript][16] canonical
    const uint32_t* trows;           // [n_transcript] the row of every transcript entry (several inner proofs: each proof's sponge rows start its segment)
    uint32_t n_chains, n_transcript;
    uint64_t rows, used_rows;
    uint32_t* trace; uint64_t ld;    // [rows][ld >= R_WIDTH], Montgomery
    uint32_t* roots;                 // [n_chains][8] canonical: where every chain ends
    const uint32_t* row_bits = nullptr;   // machine mode (machine_verifier.inl): EVERY used row arrives as a transcript-style row -- its input state, and here
    const uint32_t* row_kps = nullptr;    // its direction bit and its KP column (canonical); null: 0 / untouched
    // round 6 (machine mode, the queries' rows walked by mrec_chains_kernel): the input states arrive in Montgomery form, entry r is row r (trows null),
    // and the first skip_first rows of every segment of seg_rows rows are left alone (a proof's transcript rows)
    uint32_t inputs_monty = 0, seg_rows = 0, skip_first = 0;
};

// ---- machine mode (machine_verifier.inl), round 6: the Poseidon2 chains of the inner proofs' queries walked ON THE DEVICE -- one lane per
// (proof, query, commitment) and per (proof, query, FRI layer).  A commitment's chain: the sponges over the concatenated rows of every height
// (shorter heights first, the tallest last; the words are gathered from the proof by the plan's offsets), then the path with the
// injections (compress(node, digest of the height reached)); a layer's chain: the pair's leaf row, then its path.  A lane writes every row's
// input state, direction bit and KP column where the host's walk put them (machine_verifier.inl fill_proof) -- the chains are serial, so it runs the
// fast permutation only; the rows' 360 columns are filled by p2r_rows_kernel, one lane per row --; a chain that does not end in its root marks its
// proof in err (1: a FRI layer, 3: a commitment).
struct MrecTreePlan {
    uint32_t row0, rows_per_query;     // P2 rows: proof segment + row0 + query * rows_per_query
    uint32_t n_sponges, depth, shift;  // heights hashed; path levels; index = query index >> shift
    uint32_t path_off;                 // the path's first word relative to the query's first word
    int32_t root_off;                  // word offset of the root in the proof; -1: the inner key's root
    uint32_t sp_words[16], sp_src[16]; // per sponge: words, where its offsets start in src
    int32_t inj[32];                   // level -> the sponge whose digest joins behind that level's compression, -1: none
};
struct MrecChainArgs {
    const uint32_t* proofs; uint64_t proof_words;     // [NP][proof_words] canonical
    const uint32_t* vals; uint64_t vals_stride;       // per proof: the query indices at words [0, Q)
    const MrecTreePlan* trees; uint32_t n_trees;
    const int32_t* src;                                // leaf word offsets relative to the query's first word
    uint32_t NP, Q, R, H;
    uint32_t o_queries, per_query, o_lroots, fri_off;
    uint32_t p2_rows, p2_fri0, fri_rows;
    const uint32_t* pairs; const uint32_t* pair_k;     // [NP Q R][8] Montgomery, [NP Q R]: a layer's pair and its index
    uint32_t key_root[8];                              // canonical
    uint32_t* row_in; uint32_t* row_bit; uint32_t* row_kp;   // [used rows][16] Montgomery, [used rows], [used rows]: what every row starts from (p2r_rows_kernel fills the columns)
    uint32_t* err;                                     // [NP]
    uint32_t* trace; uint64_t ld;                      // non-null: sixteen lanes per chain fill the Poseidon2 chip's rows themselves (mrec_chains16_kernel); row_in / row_bit / row_kp are then unused
};

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
