// kernels.h -- internal launch interface between the gfx950 kernel translation units
// (ntt.hip, hash.hip, stark.hip, util.hip) and the host-side orchestration (context.cpp,
// prover.cpp).  Nothing here is part of the public C ABI (include/zkhip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace zk {

// ---------------------------------------------------------------- NTT pass (ntt.hip)
// One launch = one pass of a (up to) two-pass NTT over the columns of a row-major
// matrix: every workgroup transforms a tile of M = 2^log_m rows x C columns.
//   in row of tile element n   = tile * in_tile_mul  + n * in_stride
//   out row of tile element k  = tile * out_tile_mul + o(k) * out_stride,
//   o(k) = bitrev_m(k) if bitrev_out else k
//   value written = post[tile*M + k] * NTT_M( pre[n] * in[n] )[k]
struct NttPassArgs {
    const uint32_t* in;
    uint32_t* out;
    uint64_t in_ld, out_ld;          // row pitch, in elements
    uint32_t ncols;
    uint32_t num_tiles;
    uint32_t log_m;                  // 5..10
    uint64_t in_tile_mul, in_stride;
    uint64_t out_tile_mul, out_stride;
    uint32_t bitrev_out;
    uint32_t map_mode;               // 0: column group fastest; 1: XCD-aware
    const uint32_t* w1024;           // w_1024^e (forward) or w_1024^-e (inverse), e < 1024
    const uint32_t* pre;             // [M] or nullptr
    const uint32_t* post;            // [num_tiles * M] or nullptr
};
hipError_t launch_ntt_pass(const NttPassArgs& a, bool inverse, hipStream_t s);

// out[i] = scale * base^i, i < n   (Montgomery form in and out)
hipError_t launch_pow_table(uint32_t* out, size_t n, uint32_t base, uint32_t scale, hipStream_t s);
// out[i*cols + k] = scale * omega^(i*k) * shift^i
hipError_t launch_post_table(uint32_t* out, uint32_t rows, uint32_t cols, uint32_t omega,
                             uint32_t shift, uint32_t scale, hipStream_t s);

// ---------------------------------------------------------------- hashing (hash.hip)
struct MatDesc {
    const uint32_t* ptr;
    uint64_t ld;       // row pitch in elements
    uint32_t width;
};
constexpr int MAX_LEAF_MATS = 4;
struct LeafArgs {
    MatDesc mats[MAX_LEAF_MATS];
    int nmats;
    uint64_t height;
    uint32_t* digests;   // [height][8]
};
hipError_t launch_hash_rows(const LeafArgs& a, hipStream_t s);
// parents[i] = compress(children[2i], children[2i+1]), i < count
hipError_t launch_compress_level(const uint32_t* children, uint32_t* parents, uint64_t count,
                                 hipStream_t s);
// all remaining levels of a small subtree in one launch: `tree` points at a level with
// `count` (<= 2048, power of two) digests followed by room for the levels above it
hipError_t launch_compress_top(uint32_t* tree, uint32_t count, hipStream_t s);
// raw permutation on `count` states of 16 words (KAT / microbenchmark)
hipError_t launch_permute_states(uint32_t* states, uint64_t count, hipStream_t s);

// ---------------------------------------------------------------- utilities (util.hip)
hipError_t launch_fill_uniform(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows,
                               uint32_t width, hipStream_t s);
hipError_t launch_gen_trace(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows,
                            uint32_t width, hipStream_t s);
// element-wise Montgomery <-> canonical conversion
hipError_t launch_convert(const uint32_t* in, uint32_t* out, uint64_t n, bool to_monty_form,
                          hipStream_t s);

}  // namespace zk
