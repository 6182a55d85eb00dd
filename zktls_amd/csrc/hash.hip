// hash.hip -- Poseidon2 Merkle commitment kernels for gfx950: row (leaf) hashing with the
// overwrite-mode sponge, 2-to-1 compression of tree levels, and a single-workgroup
// kernel for the small top of the tree.
//
// Replaces p3-merkle-tree 0.2.1-succinct FieldMerkleTreeMmcs::commit and the
// PaddingFreeSponge / TruncatedPermutation of p3-symmetric (reference Cargo.lock:4013,
// 4044) on the path below crates/guest-prover-sp1/src/sp1.rs:116.
//
// One permutation state per lane, held in 16 VGPRs: the permutation is ~800 modular
// multiplications per 8 absorbed words, i.e. integer-VALU bound, not HBM bound
// (DESIGN.md section 4.2), so lanes never idle on a partial round and no cross-lane
// traffic is needed.
#include "poseidon2.cuh"
#include "kernels.h"

namespace zk {

__device__ __forceinline__ uint32_t load_virtual(const LeafArgs& a, uint64_t row, uint32_t vc) {
    // concatenation of the rows of up to MAX_LEAF_MATS matrices; control flow is uniform
    uint32_t off = vc;
#pragma unroll
    for (int m = 0; m < MAX_LEAF_MATS; m++) {
        if (m < a.nmats) {
            if (off < a.mats[m].width) return a.mats[m].ptr[row * a.mats[m].ld + off];
            off -= a.mats[m].width;
        }
    }
    return 0u;
}

__global__ void __launch_bounds__(256) hash_rows_generic_kernel(LeafArgs a, uint32_t total_w) {
    const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.height) return;
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0u;
    for (uint32_t q = 0; q < total_w; q += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (q + i < total_w) s[i] = load_virtual(a, row, q + i);
        p2_permute_dev(s);
    }
    uint4* d = reinterpret_cast<uint4*>(a.digests + row * 8);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]);
    d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// single matrix, width % 8 == 0, 16-byte aligned rows: 2 x dwordx4 per absorbed block
__global__ void __launch_bounds__(256) hash_rows_vec_kernel(const uint32_t* __restrict__ mat, uint64_t ld,
                                                            uint32_t width, uint64_t height,
                                                            uint32_t* __restrict__ digests) {
    const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= height) return;
    const uint4* rp = reinterpret_cast<const uint4*>(mat + row * ld);
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0u;
    for (uint32_t q = 0; q < width / 8; q++) {
        uint4 v0 = rp[2 * q], v1 = rp[2 * q + 1];
        s[0] = v0.x; s[1] = v0.y; s[2] = v0.z; s[3] = v0.w;
        s[4] = v1.x; s[5] = v1.y; s[6] = v1.z; s[7] = v1.w;
        p2_permute_dev(s);
    }
    uint4* d = reinterpret_cast<uint4*>(digests + row * 8);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]);
    d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

hipError_t launch_hash_rows(const LeafArgs& a, hipStream_t s) {
    if (a.height == 0) return hipSuccess;
    if (a.nmats < 1 || a.nmats > MAX_LEAF_MATS) return hipErrorInvalidValue;
    uint32_t total = 0;
    for (int m = 0; m < a.nmats; m++) total += a.mats[m].width;
    dim3 block(256), grid((unsigned)((a.height + 255) / 256));
    const MatDesc& m0 = a.mats[0];
    bool vec = a.nmats == 1 && m0.width % 8 == 0 && m0.ld % 4 == 0 &&
               (reinterpret_cast<uintptr_t>(m0.ptr) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(hash_rows_vec_kernel, grid, block, 0, s, m0.ptr, m0.ld, m0.width, a.height, a.digests);
    else
        hipLaunchKernelGGL(hash_rows_generic_kernel, grid, block, 0, s, a, total);
    return hipGetLastError();
}

__device__ __forceinline__ void compress_node(const uint32_t* children, uint32_t* parent) {
    const uint4* cp = reinterpret_cast<const uint4*>(children);
    uint4 v0 = cp[0], v1 = cp[1], v2 = cp[2], v3 = cp[3];
    uint32_t s[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w,
                      v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    p2_permute_dev(s);
    uint4* d = reinterpret_cast<uint4*>(parent);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]);
    d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

__global__ void __launch_bounds__(256) compress_level_kernel(const uint32_t* __restrict__ children,
                                                             uint32_t* __restrict__ parents, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    compress_node(children + 16 * i, parents + 8 * i);
}
hipError_t launch_compress_level(const uint32_t* children, uint32_t* parents, uint64_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(compress_level_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, children, parents, count);
    return hipGetLastError();
}

// levels count -> count/2 -> ... -> 1 inside one workgroup (count <= 2048)
__global__ void __launch_bounds__(1024) compress_top_kernel(uint32_t* tree, uint32_t count) {
    uint32_t* level = tree;
    for (uint32_t n = count; n > 1; n >>= 1) {
        uint32_t* next = level + 8 * (size_t)n;
        for (uint32_t i = threadIdx.x; i < n / 2; i += blockDim.x) compress_node(level + 16 * (size_t)i, next + 8 * (size_t)i);
        __threadfence_block();
        __syncthreads();
        level = next;
    }
}
hipError_t launch_compress_top(uint32_t* tree, uint32_t count, hipStream_t s) {
    if (count <= 1) return hipSuccess;
    if (count > 2048) return hipErrorInvalidValue;
    unsigned threads = count / 2 < 64 ? 64 : count / 2;
    hipLaunchKernelGGL(compress_top_kernel, dim3(1), dim3(threads), 0, s, tree, count);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) permute_states_kernel(uint32_t* states, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = states[16 * i + k];
    p2_permute_dev(s);
#pragma unroll
    for (int k = 0; k < 16; k++) states[16 * i + k] = s[k];
}
hipError_t launch_permute_states(uint32_t* states, uint64_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(permute_states_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, states, count);
    return hipGetLastError();
}

}  // namespace zk
