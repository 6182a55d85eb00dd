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
#include "batch.h"
#include "p2chip.h"

namespace zk {

hipError_t hash_upload_p2_tables(const P2Tables& t, hipStream_t s) { return p2_upload_tables(t, s); }

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

// 16-lanes-per-permutation kernels, defined further down
__global__ void hash_rows16_kernel(LeafArgs a, uint32_t total_w);
struct hash_rows16_kernel_bargs { LeafArgs a; uint32_t total_w; static hash_rows16_kernel_bargs make(LeafArgs a, uint32_t total_w) { return hash_rows16_kernel_bargs{a, total_w}; } };
__global__ void hash_rows16_kernel_batch(const hash_rows16_kernel_bargs* __restrict__ zk_arr);
__global__ void compress_level16_kernel(const uint32_t* __restrict__ children, uint32_t* __restrict__ parents, uint32_t count);
struct compress_level16_kernel_bargs { const uint32_t* children; uint32_t* parents; uint32_t count; static compress_level16_kernel_bargs make(const uint32_t* children, uint32_t* parents, uint32_t count) { return compress_level16_kernel_bargs{children, parents, count}; } };
__global__ void compress_level16_kernel_batch(const compress_level16_kernel_bargs* __restrict__ zk_arr);
__global__ void compress_top16_kernel(uint32_t* tree, uint32_t count);
struct compress_top16_kernel_bargs { uint32_t* tree; uint32_t count; static compress_top16_kernel_bargs make(uint32_t* tree, uint32_t count) { return compress_top16_kernel_bargs{tree, count}; } };
__global__ void compress_top16_kernel_batch(const compress_top16_kernel_bargs* __restrict__ zk_arr);
__global__ void compress_sub16_kernel(uint32_t* tree, uint32_t count, uint32_t sub);
struct compress_sub16_kernel_bargs { uint32_t* tree; uint32_t count; uint32_t sub; static compress_sub16_kernel_bargs make(uint32_t* tree, uint32_t count, uint32_t sub) { return compress_sub16_kernel_bargs{tree, count, sub}; } };
__global__ void compress_sub16_kernel_batch(const compress_sub16_kernel_bargs* __restrict__ zk_arr);
__global__ void hash_sub16_kernel(LeafArgs a, uint32_t total_w, uint32_t sub);
struct hash_sub16_kernel_bargs { LeafArgs a; uint32_t total_w; uint32_t sub; static hash_sub16_kernel_bargs make(LeafArgs a, uint32_t total_w, uint32_t sub) { return hash_sub16_kernel_bargs{a, total_w, sub}; } };
__global__ void hash_sub16_kernel_batch(const hash_sub16_kernel_bargs* __restrict__ zk_arr);

__device__ __forceinline__ void hash_rows_generic_kernel_body(const LeafArgs& a, uint32_t total_w) {
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
__global__ void __launch_bounds__(256) hash_rows_generic_kernel(LeafArgs a, uint32_t total_w) { hash_rows_generic_kernel_body(a, total_w); }
struct hash_rows_generic_kernel_bargs { LeafArgs a; uint32_t total_w; static hash_rows_generic_kernel_bargs make(LeafArgs a, uint32_t total_w) { return hash_rows_generic_kernel_bargs{a, total_w}; } };
__global__ void __launch_bounds__(256) hash_rows_generic_kernel_batch(const hash_rows_generic_kernel_bargs* __restrict__ zk_arr) { const hash_rows_generic_kernel_bargs& zk_b = zk_arr[blockIdx.z]; hash_rows_generic_kernel_body(zk_b.a, zk_b.total_w); }


// single matrix, width % 4 == 0, 16-byte aligned rows: 2 x dwordx4 per absorbed block
__device__ __forceinline__ void hash_rows_vec_kernel_body(const uint32_t* __restrict__ mat, uint64_t ld, uint32_t width, uint64_t height, uint32_t* __restrict__ digests) {
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
    if (width & 4u) {                       // a last half block: the other four rate words keep the state (overwrite mode)
        const uint4 v0 = rp[width / 4 - 1];
        s[0] = v0.x; s[1] = v0.y; s[2] = v0.z; s[3] = v0.w;
        p2_permute_dev(s);
    }
    uint4* d = reinterpret_cast<uint4*>(digests + row * 8);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]);
    d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}
__global__ void __launch_bounds__(256) hash_rows_vec_kernel(const uint32_t* __restrict__ mat, uint64_t ld, uint32_t width, uint64_t height, uint32_t* __restrict__ digests) { hash_rows_vec_kernel_body(mat, ld, width, height, digests); }
struct hash_rows_vec_kernel_bargs { const uint32_t* mat; uint64_t ld; uint32_t width; uint64_t height; uint32_t* digests; static hash_rows_vec_kernel_bargs make(const uint32_t* mat, uint64_t ld, uint32_t width, uint64_t height, uint32_t* digests) { return hash_rows_vec_kernel_bargs{mat, ld, width, height, digests}; } };
__global__ void __launch_bounds__(256) hash_rows_vec_kernel_batch(const hash_rows_vec_kernel_bargs* __restrict__ zk_arr) { const hash_rows_vec_kernel_bargs& zk_b = zk_arr[blockIdx.z]; hash_rows_vec_kernel_body(zk_b.mat, zk_b.ld, zk_b.width, zk_b.height, zk_b.digests); }


// Several matrices of one height whose widths and pitches are multiples of 4 words (16-byte aligned rows): the sponge absorbs the
// CONCATENATED row, so with quads as the unit a rate block is two consecutive quads of the virtual row, wherever the matrix boundaries
// fall.  A cursor (matrix, quad) walks the matrices -- wave-uniform, the descriptors come from the argument block through scalar loads --
// and every absorbed block is two (or, for a last half block, one) 16-byte loads, as in the single-matrix kernel above; the generic
// kernel's word-by-word load_virtual costs a quarter more per permutation (profiles/r05_multichip_*: SP1's shard shape puts two to three
// matrices on the tallest height of every commitment).
__device__ __forceinline__ void sponge_rows_vec(const LeafArgs& a, uint64_t row, uint32_t total_q, uint32_t s[16]) {
    int m = 0;
    uint32_t q = 0, nq = a.mats[0].width / 4;
    const uint4* rp = reinterpret_cast<const uint4*>(a.mats[0].ptr + row * a.mats[0].ld);
    for (uint32_t b = 0; b < total_q; b += 2) {
        while (q == nq) { m++; q = 0; nq = a.mats[m].width / 4; rp = reinterpret_cast<const uint4*>(a.mats[m].ptr + row * a.mats[m].ld); }
        const uint4 v0 = rp[q++];
        s[0] = v0.x; s[1] = v0.y; s[2] = v0.z; s[3] = v0.w;
        if (b + 1 < total_q) {                  // (a last half block keeps the other four rate words: overwrite mode)
            while (q == nq) { m++; q = 0; nq = a.mats[m].width / 4; rp = reinterpret_cast<const uint4*>(a.mats[m].ptr + row * a.mats[m].ld); }
            const uint4 v1 = rp[q++];
            s[4] = v1.x; s[5] = v1.y; s[6] = v1.z; s[7] = v1.w;
        }
        p2_permute_dev(s);
    }
}
__device__ __forceinline__ void hash_rows_mvec_kernel_body(const LeafArgs& a, uint32_t total_q) {
    const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.height) return;
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0u;
    sponge_rows_vec(a, row, total_q, s);
    uint4* d = reinterpret_cast<uint4*>(a.digests + row * 8);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]);
    d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}
__global__ void __launch_bounds__(256) hash_rows_mvec_kernel(LeafArgs a, uint32_t total_q) { hash_rows_mvec_kernel_body(a, total_q); }
struct hash_rows_mvec_kernel_bargs { LeafArgs a; uint32_t total_q; static hash_rows_mvec_kernel_bargs make(LeafArgs a, uint32_t total_q) { return hash_rows_mvec_kernel_bargs{a, total_q}; } };
__global__ void __launch_bounds__(256) hash_rows_mvec_kernel_batch(const hash_rows_mvec_kernel_bargs* __restrict__ zk_arr) { const hash_rows_mvec_kernel_bargs& zk_b = zk_arr[blockIdx.z]; hash_rows_mvec_kernel_body(zk_b.a, zk_b.total_q); }

// One level of a mixed-height tree at which matrices are injected, in ONE launch: parent = compress(children), then
// node = compress(parent, sponge(the injected matrices' row)) -- what compress_level + hash_rows (into a scratch level) + inject did in
// three launches with the row digests and the parents making a round trip through memory each.  a.digests = the parents' level.
__device__ __forceinline__ void compress_inject_mvec_kernel_body(const uint32_t* __restrict__ children, const LeafArgs& a, uint32_t total_q) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.height) return;
    const uint4* cp = reinterpret_cast<const uint4*>(children + 16 * i);
    const uint4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
    uint32_t s[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
    uint32_t t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = 0u;
    p2_permute_dev(s);
    sponge_rows_vec(a, i, total_q, t);
#pragma unroll
    for (int k = 0; k < 8; k++) s[8 + k] = t[k];
    p2_permute_dev(s);
    uint4* d = reinterpret_cast<uint4*>(a.digests + i * 8);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]);
    d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}
__global__ void __launch_bounds__(256) compress_inject_mvec_kernel(const uint32_t* __restrict__ children, LeafArgs a, uint32_t total_q) { compress_inject_mvec_kernel_body(children, a, total_q); }
struct compress_inject_mvec_kernel_bargs { const uint32_t* children; LeafArgs a; uint32_t total_q; static compress_inject_mvec_kernel_bargs make(const uint32_t* children, LeafArgs a, uint32_t total_q) { return compress_inject_mvec_kernel_bargs{children, a, total_q}; } };
__global__ void __launch_bounds__(256) compress_inject_mvec_kernel_batch(const compress_inject_mvec_kernel_bargs* __restrict__ zk_arr) { const compress_inject_mvec_kernel_bargs& zk_b = zk_arr[blockIdx.z]; compress_inject_mvec_kernel_body(zk_b.children, zk_b.a, zk_b.total_q); }

// every matrix: width > 0, width and pitch multiples of 4 words, 16-byte aligned base
static bool leaf_mats_vec(const LeafArgs& a) {
    for (int m = 0; m < a.nmats; m++) {
        const MatDesc& d = a.mats[m];
        if (d.width == 0 || d.width % 4 || d.ld % 4 || (reinterpret_cast<uintptr_t>(d.ptr) & 15)) return false;
    }
    return a.nmats >= 1;
}
hipError_t launch_compress_inject(const uint32_t* children, const LeafArgs& a, hipStream_t s) {
    if (a.height == 0) return hipSuccess;
    if (a.nmats < 1 || a.nmats > MAX_LEAF_MATS || !leaf_mats_vec(a)) return hipErrorInvalidValue;
    uint32_t total = 0;
    for (int m = 0; m < a.nmats; m++) total += a.mats[m].width;
    ZK_LAUNCH(compress_inject_mvec_kernel, compress_inject_mvec_kernel_batch, compress_inject_mvec_kernel_bargs, dim3((unsigned)((a.height + 255) / 256)), dim3(256), 0, s, children, a, total / 4);
    return hipGetLastError();
}
bool compress_inject_ok(const LeafArgs& a) { return a.height > coop_max_nodes() && leaf_mats_vec(a); }

hipError_t launch_hash_rows(const LeafArgs& a, hipStream_t s) {
    if (a.height == 0) return hipSuccess;
    if (a.nmats < 1 || a.nmats > MAX_LEAF_MATS) return hipErrorInvalidValue;
    uint32_t total = 0;
    for (int m = 0; m < a.nmats; m++) total += a.mats[m].width;
    if (a.height <= coop_max_nodes()) {
        const uint64_t threads = a.height * 16;
        ZK_LAUNCH(hash_rows16_kernel, hash_rows16_kernel_batch, hash_rows16_kernel_bargs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a, total);
        return hipGetLastError();
    }
    dim3 block(256), grid((unsigned)((a.height + 255) / 256));
    const MatDesc& m0 = a.mats[0];
    bool vec = a.nmats == 1 && m0.width % 4 == 0 && m0.ld % 4 == 0 &&
               (reinterpret_cast<uintptr_t>(m0.ptr) & 15) == 0;
    if (vec)
        ZK_LAUNCH(hash_rows_vec_kernel, hash_rows_vec_kernel_batch, hash_rows_vec_kernel_bargs, grid, block, 0, s, m0.ptr, m0.ld, m0.width, a.height, a.digests);
    else if (leaf_mats_vec(a))
        ZK_LAUNCH(hash_rows_mvec_kernel, hash_rows_mvec_kernel_batch, hash_rows_mvec_kernel_bargs, grid, block, 0, s, a, total / 4);
    else
        ZK_LAUNCH(hash_rows_generic_kernel, hash_rows_generic_kernel_batch, hash_rows_generic_kernel_bargs, grid, block, 0, s, a, total);
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

__device__ __forceinline__ void compress_level_kernel_body(const uint32_t* __restrict__ children, uint32_t* __restrict__ parents, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    compress_node(children + 16 * i, parents + 8 * i);
}
__global__ void __launch_bounds__(256) compress_level_kernel(const uint32_t* __restrict__ children, uint32_t* __restrict__ parents, uint64_t count) { compress_level_kernel_body(children, parents, count); }
struct compress_level_kernel_bargs { const uint32_t* children; uint32_t* parents; uint64_t count; static compress_level_kernel_bargs make(const uint32_t* children, uint32_t* parents, uint64_t count) { return compress_level_kernel_bargs{children, parents, count}; } };
__global__ void __launch_bounds__(256) compress_level_kernel_batch(const compress_level_kernel_bargs* __restrict__ zk_arr) { const compress_level_kernel_bargs& zk_b = zk_arr[blockIdx.z]; compress_level_kernel_body(zk_b.children, zk_b.parents, zk_b.count); }

hipError_t launch_compress_level(const uint32_t* children, uint32_t* parents, uint64_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    if (count <= coop_max_nodes()) {
        const uint64_t threads = count * 16;
        ZK_LAUNCH(compress_level16_kernel, compress_level16_kernel_batch, compress_level16_kernel_bargs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, children, parents, (uint32_t)count);
        return hipGetLastError();
    }
    ZK_LAUNCH(compress_level_kernel, compress_level_kernel_batch, compress_level_kernel_bargs, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, children, parents, count);
    return hipGetLastError();
}

// node[i] = compress(node[i], extra[i]): injection of a shorter matrix's row digests at its level
__device__ __forceinline__ void inject_kernel_body(uint32_t* __restrict__ nodes, const uint32_t* __restrict__ extra, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint4* a = reinterpret_cast<const uint4*>(nodes + 8 * i);
    const uint4* b = reinterpret_cast<const uint4*>(extra + 8 * i);
    const uint4 v0 = a[0], v1 = a[1], v2 = b[0], v3 = b[1];
    uint32_t s[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    p2_permute_dev(s);
    uint4* d = reinterpret_cast<uint4*>(nodes + 8 * i);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]);
    d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}
__global__ void __launch_bounds__(256) inject_kernel(uint32_t* __restrict__ nodes, const uint32_t* __restrict__ extra, uint64_t count) { inject_kernel_body(nodes, extra, count); }
struct inject_kernel_bargs { uint32_t* nodes; const uint32_t* extra; uint64_t count; static inject_kernel_bargs make(uint32_t* nodes, const uint32_t* extra, uint64_t count) { return inject_kernel_bargs{nodes, extra, count}; } };
__global__ void __launch_bounds__(256) inject_kernel_batch(const inject_kernel_bargs* __restrict__ zk_arr) { const inject_kernel_bargs& zk_b = zk_arr[blockIdx.z]; inject_kernel_body(zk_b.nodes, zk_b.extra, zk_b.count); }

hipError_t launch_inject(uint32_t* nodes, const uint32_t* extra, uint64_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    ZK_LAUNCH(inject_kernel, inject_kernel_batch, inject_kernel_bargs, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, nodes, extra, count);
    return hipGetLastError();
}

// levels count -> count/2 -> ... -> 1 inside one workgroup (count <= 2048)
__device__ __forceinline__ void compress_top_kernel_body(uint32_t* tree, uint32_t count) {
    uint32_t* level = tree;
    for (uint32_t n = count; n > 1; n >>= 1) {
        uint32_t* next = level + 8 * (size_t)n;
        for (uint32_t i = threadIdx.x; i < n / 2; i += blockDim.x) compress_node(level + 16 * (size_t)i, next + 8 * (size_t)i);
        __threadfence_block();
        __syncthreads();
        level = next;
    }
}
__global__ void __launch_bounds__(1024) compress_top_kernel(uint32_t* tree, uint32_t count) { compress_top_kernel_body(tree, count); }
struct compress_top_kernel_bargs { uint32_t* tree; uint32_t count; static compress_top_kernel_bargs make(uint32_t* tree, uint32_t count) { return compress_top_kernel_bargs{tree, count}; } };
__global__ void __launch_bounds__(1024) compress_top_kernel_batch(const compress_top_kernel_bargs* __restrict__ zk_arr) { const compress_top_kernel_bargs& zk_b = zk_arr[blockIdx.z]; compress_top_kernel_body(zk_b.tree, zk_b.count); }

hipError_t launch_compress_top(uint32_t* tree, uint32_t count, hipStream_t s) {
    if (count <= 1) return hipSuccess;
    if (count > COOP_TOP_NODES) return hipErrorInvalidValue;
    unsigned threads = count * 8 < 64 ? 64 : (count * 8 > 1024 ? 1024 : count * 8);
    ZK_LAUNCH(compress_top16_kernel, compress_top16_kernel_batch, compress_top16_kernel_bargs, dim3(1), dim3(threads), 0, s, tree, count);
    return hipGetLastError();
}

hipError_t launch_hash_sub(const LeafArgs& a, uint32_t sub, hipStream_t s) {
    if (sub < 2 || (sub & (sub - 1)) || a.height % sub || a.height > COOP_MAX_NODES || a.nmats < 1 || a.nmats > MAX_LEAF_MATS) return hipErrorInvalidValue;
    uint32_t total = 0;
    for (int m = 0; m < a.nmats; m++) total += a.mats[m].width;
    unsigned threads = sub * 8 < 64 ? 64 : (sub * 8 > 1024 ? 1024 : sub * 8);
    ZK_LAUNCH(hash_sub16_kernel, hash_sub16_kernel_batch, hash_sub16_kernel_bargs, dim3((unsigned)(a.height / sub)), dim3(threads), 0, s, a, total, sub);
    return hipGetLastError();
}
hipError_t launch_compress_sub(uint32_t* tree, uint32_t count, uint32_t sub, hipStream_t s) {
    if (sub < 2 || (sub & (sub - 1)) || count % sub || count > COOP_MAX_NODES) return hipErrorInvalidValue;
    unsigned threads = sub * 8 < 64 ? 64 : (sub * 8 > 1024 ? 1024 : sub * 8);
    ZK_LAUNCH(compress_sub16_kernel, compress_sub16_kernel_batch, compress_sub16_kernel_bargs, dim3(count / sub), dim3(threads), 0, s, tree, count, sub);
    return hipGetLastError();
}

// ------------------------------------------------------------------ latency-optimised form
// One permutation spread over the 16 lanes of a DPP row (state word i in lane i): an
// external round is 4 dependent products + ~13 dependent additions, an internal round one
// S-box + a 4-step rotate-and-add + one product, instead of ~10 k serial instructions.  A
// permutation finishes in ~3 us instead of ~19 us, which is what bounds the SMALL levels of
// every Merkle tree (a proof walks ~250 such levels one after the other).  Throughput per
// wave is worse (4 permutations instead of 64), so the wide levels keep the lane-per-state form.
template <int CTRL>
ZK_D uint32_t dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false); }

struct CoopConsts { uint32_t rc_ext[8]; uint32_t diag; };
ZK_D CoopConsts coop_load_consts(int lane16) {
    CoopConsts k;
#pragma unroll
    for (int r = 0; r < 8; r++) k.rc_ext[r] = P2K.ext_rcm[r][lane16];      // rc - P, see p2_sbox_rc_dev
    k.diag = P2K.diag[lane16];
    return k;
}
ZK_D uint32_t coop_external_linear(uint32_t x) {
    // M4 = circ(2,3,1,1) inside each quad: y_i = 2 x_i + 3 x_{i+1} + x_{i+2} + x_{i+3} = (quad sum) + x_i + 2 x_{i+1};
    // quad_perm [1,2,3,0] [2,3,0,1] [3,0,1,2] are the three rotations
    const uint32_t r1 = dpp<0x39>(x), r2 = dpp<0x4E>(x), r3 = dpp<0x93>(x);
    const uint32_t sum = dadd(dadd(x, r1), dadd(r2, r3));
    const uint32_t y = dadd(dadd(sum, x), ddbl(r1));
    uint32_t t = dadd(y, dpp<0x124>(y));                                  // row_ror:4
    t = dadd(t, dpp<0x128>(t));                                           // row_ror:8
    return dadd(y, t);
}
ZK_D uint32_t coop_permute(uint32_t x, int lane16, const CoopConsts& k) {
    x = coop_external_linear(x);
#pragma unroll 1
    for (int r = 0; r < 4; r++) x = coop_external_linear(p2_sbox_rc_dev(x, k.rc_ext[r]));
#pragma unroll 1
    for (int r = 0; r < 13; r++) {
        const uint32_t sb = p2_sbox_rc_dev(x, P2K.int_rcm[r]);
        x = lane16 == 0 ? sb : x;
        uint32_t t = dadd(x, dpp<0x128>(x));
        t = dadd(t, dpp<0x124>(t));
        t = dadd(t, dpp<0x122>(t));
        t = dadd(t, dpp<0x121>(t));                                       // every lane holds the sum
        x = dadd(dmul(x, k.diag), t);
    }
#pragma unroll 1
    for (int r = 4; r < 8; r++) x = coop_external_linear(p2_sbox_rc_dev(x, k.rc_ext[r]));
    return x;
}

// parents[i] = compress(children[2i], children[2i+1]); one node per 16 lanes
__device__ __forceinline__ void compress_level16_kernel_body(const uint32_t* __restrict__ children, uint32_t* __restrict__ parents, uint32_t count) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t node = gid >> 4;
    const int lane16 = threadIdx.x & 15;
    if (node >= count) return;                       // whole rows leave together
    const CoopConsts k = coop_load_consts(lane16);
    const uint32_t x = coop_permute(children[16 * (size_t)node + lane16], lane16, k);
    if (lane16 < 8) parents[8 * (size_t)node + lane16] = x;
}
__global__ void __launch_bounds__(256) compress_level16_kernel(const uint32_t* __restrict__ children, uint32_t* __restrict__ parents, uint32_t count) { compress_level16_kernel_body(children, parents, count); }
__global__ void __launch_bounds__(256) compress_level16_kernel_batch(const compress_level16_kernel_bargs* __restrict__ zk_arr) { const compress_level16_kernel_bargs& zk_b = zk_arr[blockIdx.z]; compress_level16_kernel_body(zk_b.children, zk_b.parents, zk_b.count); }

// all levels count -> 1 in one workgroup of 1024 threads (count <= 512), 64 nodes per step
__device__ __forceinline__ void compress_top16_kernel_body(uint32_t* tree, uint32_t count) {
    const int lane16 = threadIdx.x & 15;
    const uint32_t grp = threadIdx.x >> 4, ngrp = blockDim.x >> 4;
    const CoopConsts k = coop_load_consts(lane16);
    uint32_t* level = tree;
    for (uint32_t n = count; n > 1; n >>= 1) {
        uint32_t* next = level + 8 * (size_t)n;
        for (uint32_t i = grp; i < n / 2; i += ngrp) {
            const uint32_t x = coop_permute(level[16 * (size_t)i + lane16], lane16, k);
            if (lane16 < 8) next[8 * (size_t)i + lane16] = x;
        }
        __threadfence_block();
        __syncthreads();
        level = next;
    }
}
__global__ void __launch_bounds__(1024) compress_top16_kernel(uint32_t* tree, uint32_t count) { compress_top16_kernel_body(tree, count); }
__global__ void __launch_bounds__(1024) compress_top16_kernel_batch(const compress_top16_kernel_bargs* __restrict__ zk_arr) { const compress_top16_kernel_bargs& zk_b = zk_arr[blockIdx.z]; compress_top16_kernel_body(zk_b.tree, zk_b.count); }

// Several levels of a medium tree in one launch: `tree` points at a level with `count` digests (levels above follow it, as in
// compress_top16_kernel); workgroup b reduces the `sub` consecutive digests [b sub, (b + 1) sub) of that level to one node, writing
// its share of every level on the way.  count / sub nodes remain for compress_top16_kernel.
__device__ __forceinline__ void compress_sub16_kernel_body(uint32_t* tree, uint32_t count, uint32_t sub) {
    const int lane16 = threadIdx.x & 15;
    const uint32_t grp = threadIdx.x >> 4, ngrp = blockDim.x >> 4;
    const CoopConsts k = coop_load_consts(lane16);
    uint32_t* level = tree;
    uint32_t n = count, mine = sub;                       // nodes in the level, nodes of this workgroup in it
    while (mine > 1) {
        uint32_t* next = level + 8 * (size_t)n;
        const size_t in0 = (size_t)blockIdx.x * mine, out0 = (size_t)blockIdx.x * (mine / 2);
        for (uint32_t i = grp; i < mine / 2; i += ngrp) {
            const uint32_t x = coop_permute(level[16 * (in0 / 2 + i) + lane16], lane16, k);
            if (lane16 < 8) next[8 * (out0 + i) + lane16] = x;
        }
        __threadfence_block();
        __syncthreads();
        level = next; n >>= 1; mine >>= 1;
    }
}
__global__ void __launch_bounds__(1024) compress_sub16_kernel(uint32_t* tree, uint32_t count, uint32_t sub) { compress_sub16_kernel_body(tree, count, sub); }
__global__ void __launch_bounds__(1024) compress_sub16_kernel_batch(const compress_sub16_kernel_bargs* __restrict__ zk_arr) { const compress_sub16_kernel_bargs& zk_b = zk_arr[blockIdx.z]; compress_sub16_kernel_body(zk_b.tree, zk_b.count, zk_b.sub); }

// The same with the LEAVES hashed by the workgroup that owns them: a medium tree (512 < leaves <= 16 384: the FRI layers, the trees of small
// proofs) is one launch less -- workgroup b hashes rows [b sub, (b + 1) sub) into the leaf level (one row per 16 lanes, as
// hash_rows16_kernel) and walks its subtree from there.
__device__ __forceinline__ void hash_sub16_kernel_body(const LeafArgs& a, uint32_t total_w, uint32_t sub) {
    const int lane16 = threadIdx.x & 15;
    const uint32_t grp = threadIdx.x >> 4, ngrp = blockDim.x >> 4;
    const CoopConsts k = coop_load_consts(lane16);
    for (uint32_t i = grp; i < sub; i += ngrp) {
        const uint64_t row = (uint64_t)blockIdx.x * sub + i;
        uint32_t x = 0;
        for (uint32_t q = 0; q < total_w; q += 8) {
            if (lane16 < 8 && q + lane16 < total_w) x = load_virtual(a, row, q + lane16);
            x = coop_permute(x, lane16, k);
        }
        if (lane16 < 8) a.digests[row * 8 + lane16] = x;
    }
    __threadfence_block();
    __syncthreads();
    uint32_t* level = a.digests;
    uint32_t n = (uint32_t)a.height, mine = sub;
    while (mine > 1) {
        uint32_t* next = level + 8 * (size_t)n;
        const size_t in0 = (size_t)blockIdx.x * mine, out0 = (size_t)blockIdx.x * (mine / 2);
        for (uint32_t i = grp; i < mine / 2; i += ngrp) {
            const uint32_t x = coop_permute(level[16 * (in0 / 2 + i) + lane16], lane16, k);
            if (lane16 < 8) next[8 * (out0 + i) + lane16] = x;
        }
        __threadfence_block();
        __syncthreads();
        level = next; n >>= 1; mine >>= 1;
    }
}
__global__ void __launch_bounds__(1024) hash_sub16_kernel(LeafArgs a, uint32_t total_w, uint32_t sub) { hash_sub16_kernel_body(a, total_w, sub); }
__global__ void __launch_bounds__(1024) hash_sub16_kernel_batch(const hash_sub16_kernel_bargs* __restrict__ zk_arr) { const hash_sub16_kernel_bargs& zk_b = zk_arr[blockIdx.z]; hash_sub16_kernel_body(zk_b.a, zk_b.total_w, zk_b.sub); }

// leaf digests, one row per 16 lanes (small heights: FRI layers, tests)
__device__ __forceinline__ void hash_rows16_kernel_body(const LeafArgs& a, uint32_t total_w) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t row = gid >> 4;
    const int lane16 = threadIdx.x & 15;
    if (row >= a.height) return;
    const CoopConsts k = coop_load_consts(lane16);
    uint32_t x = 0;
    for (uint32_t q = 0; q < total_w; q += 8) {
        if (lane16 < 8 && q + lane16 < total_w) x = load_virtual(a, row, q + lane16);
        x = coop_permute(x, lane16, k);
    }
    if (lane16 < 8) a.digests[row * 8 + lane16] = x;
}
__global__ void __launch_bounds__(256) hash_rows16_kernel(LeafArgs a, uint32_t total_w) { hash_rows16_kernel_body(a, total_w); }
__global__ void __launch_bounds__(256) hash_rows16_kernel_batch(const hash_rows16_kernel_bargs* __restrict__ zk_arr) { const hash_rows16_kernel_bargs& zk_b = zk_arr[blockIdx.z]; hash_rows16_kernel_body(zk_b.a, zk_b.total_w); }


// ------------------------------------------------------------------ transcript step on the device
// p3-challenger DuplexChallenger<16, 8> (the host Challenger of prover.cpp, word for word): observe the 8 root words,
// sample 4.  One 16-lane row runs the permutation cooperatively, lane i holding state word i; the small input /
// output buffers live in LDS.  Lets the FRI commit loop run without a host round trip per layer.
// observe the 8 words at `root`, sample 4 into beta_out; executed by ONE 16-lane row (lane = 0..15), state word in x
ZK_D void chal_observe_root_sample_ext(uint32_t& x, int& n_in, int& n_out, uint32_t* sin, uint32_t* sout, int lane, const CoopConsts& k,
                                       const uint32_t* root, uint32_t* beta_out, uint32_t* root_log) {
    auto duplex = [&]() {
        if (lane < n_in) x = sin[lane];              // overwrite-mode absorb
        n_in = 0;
        x = coop_permute(x, lane, k);
        if (lane < 8) sout[lane] = x;
        n_out = 8;
    };
    for (int w = 0; w < 8; w++) {
        const uint32_t v = root[w];
        if (lane == 0 && root_log) root_log[w] = v;
        n_out = 0;
        if (lane == 0) sin[n_in] = v;
        n_in++;
        if (n_in == 8) duplex();
    }
    for (int e = 0; e < 4; e++) {
        if (n_in != 0 || n_out == 0) duplex();
        --n_out;
        if (lane == 0) beta_out[e] = sout[n_out];
    }
}
__device__ __forceinline__ void fri_challenge_kernel_body(DevChallenger* c, const uint32_t* __restrict__ root, uint32_t* __restrict__ beta_out, uint32_t* __restrict__ root_log) {
    __shared__ uint32_t sin[8], sout[8];
    const int lane = threadIdx.x;                    // 64 launched, lanes 0..15 work (a DPP row)
    if (lane >= 16) return;
    const CoopConsts k = coop_load_consts(lane);
    uint32_t x = c->state[lane];
    int n_in = c->n_in, n_out = c->n_out;
    if (lane < 8) { sin[lane] = c->in[lane]; sout[lane] = c->out[lane]; }
    chal_observe_root_sample_ext(x, n_in, n_out, sin, sout, lane, k, root, beta_out, root_log);
    c->state[lane] = x;
    if (lane < 8) { c->in[lane] = sin[lane]; c->out[lane] = sout[lane]; }
    if (lane == 0) { c->n_in = n_in; c->n_out = n_out; }
}
__global__ void __launch_bounds__(64) fri_challenge_kernel(DevChallenger* c, const uint32_t* __restrict__ root, uint32_t* __restrict__ beta_out, uint32_t* __restrict__ root_log) { fri_challenge_kernel_body(c, root, beta_out, root_log); }
struct fri_challenge_kernel_bargs { DevChallenger* c; const uint32_t* root; uint32_t* beta_out; uint32_t* root_log; static fri_challenge_kernel_bargs make(DevChallenger* c, const uint32_t* root, uint32_t* beta_out, uint32_t* root_log) { return fri_challenge_kernel_bargs{c, root, beta_out, root_log}; } };
__global__ void __launch_bounds__(64) fri_challenge_kernel_batch(const fri_challenge_kernel_bargs* __restrict__ zk_arr) { const fri_challenge_kernel_bargs& zk_b = zk_arr[blockIdx.z]; fri_challenge_kernel_body(zk_b.c, zk_b.root, zk_b.beta_out, zk_b.root_log); }

hipError_t launch_fri_challenge(DevChallenger* chal, const uint32_t* root, uint32_t* beta_out, uint32_t* root_log, hipStream_t s) {
    ZK_LAUNCH(fri_challenge_kernel, fri_challenge_kernel_batch, fri_challenge_kernel_bargs, dim3(1), dim3(64), 0, s, chal, root, beta_out, root_log);
    return hipGetLastError();
}

// ------------------------------------------------------------------ RISC Zero layout (row a11)
// column-major [cols][rows] polynomials, Poseidon2 width 24, rate 16: one row per lane, so a
// wave reads 64 consecutive words of every column -- the layout is coalesced as it stands.
__device__ __forceinline__ void hash_cols24_kernel_body(const uint32_t* __restrict__ mat, uint32_t cols, uint64_t rows, uint32_t* __restrict__ digests) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    uint32_t s[24];
#pragma unroll
    for (int i = 0; i < 24; i++) s[i] = 0u;
    for (uint32_t q = 0; q < cols; q += 16) {
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (q + i < cols) s[i] = mat[(uint64_t)(q + i) * rows + r];
        p24_permute_dev(s);
    }
    uint4* d = reinterpret_cast<uint4*>(digests + r * 8);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]);
    d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}
__global__ void __launch_bounds__(256) hash_cols24_kernel(const uint32_t* __restrict__ mat, uint32_t cols, uint64_t rows, uint32_t* __restrict__ digests) { hash_cols24_kernel_body(mat, cols, rows, digests); }
struct hash_cols24_kernel_bargs { const uint32_t* mat; uint32_t cols; uint64_t rows; uint32_t* digests; static hash_cols24_kernel_bargs make(const uint32_t* mat, uint32_t cols, uint64_t rows, uint32_t* digests) { return hash_cols24_kernel_bargs{mat, cols, rows, digests}; } };
__global__ void __launch_bounds__(256) hash_cols24_kernel_batch(const hash_cols24_kernel_bargs* __restrict__ zk_arr) { const hash_cols24_kernel_bargs& zk_b = zk_arr[blockIdx.z]; hash_cols24_kernel_body(zk_b.mat, zk_b.cols, zk_b.rows, zk_b.digests); }

__device__ __forceinline__ void compress24_level_kernel_body(const uint32_t* __restrict__ children, uint32_t* __restrict__ parents, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint4* cp = reinterpret_cast<const uint4*>(children + 16 * i);
    const uint4 v0 = cp[0], v1 = cp[1], v2 = cp[2], v3 = cp[3];
    uint32_t s[24] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w,
                      0, 0, 0, 0, 0, 0, 0, 0};
    p24_permute_dev(s);
    uint4* d = reinterpret_cast<uint4*>(parents + 8 * i);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]);
    d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}
__global__ void __launch_bounds__(256) compress24_level_kernel(const uint32_t* __restrict__ children, uint32_t* __restrict__ parents, uint64_t count) { compress24_level_kernel_body(children, parents, count); }
struct compress24_level_kernel_bargs { const uint32_t* children; uint32_t* parents; uint64_t count; static compress24_level_kernel_bargs make(const uint32_t* children, uint32_t* parents, uint64_t count) { return compress24_level_kernel_bargs{children, parents, count}; } };
__global__ void __launch_bounds__(256) compress24_level_kernel_batch(const compress24_level_kernel_bargs* __restrict__ zk_arr) { const compress24_level_kernel_bargs& zk_b = zk_arr[blockIdx.z]; compress24_level_kernel_body(zk_b.children, zk_b.parents, zk_b.count); }

// all levels from `count` (<= 2048) nodes down to the root in one workgroup: the small levels are pure launch latency
__device__ __forceinline__ void compress24_top_kernel_body(uint32_t* tree, uint32_t count) {
    uint32_t* level = tree;
    for (uint32_t n = count; n > 1; n >>= 1) {
        uint32_t* next = level + 8 * (size_t)n;
        for (uint32_t i = threadIdx.x; i < n / 2; i += blockDim.x) {
            const uint4* cp = reinterpret_cast<const uint4*>(level + 16 * (size_t)i);
            const uint4 v0 = cp[0], v1 = cp[1], v2 = cp[2], v3 = cp[3];
            uint32_t s[24] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w,
                              0, 0, 0, 0, 0, 0, 0, 0};
            p24_permute_dev(s);
            uint4* d = reinterpret_cast<uint4*>(next + 8 * (size_t)i);
            d[0] = make_uint4(s[0], s[1], s[2], s[3]);
            d[1] = make_uint4(s[4], s[5], s[6], s[7]);
        }
        __threadfence_block();
        __syncthreads();
        level = next;
    }
}
__global__ void __launch_bounds__(1024) compress24_top_kernel(uint32_t* tree, uint32_t count) { compress24_top_kernel_body(tree, count); }
struct compress24_top_kernel_bargs { uint32_t* tree; uint32_t count; static compress24_top_kernel_bargs make(uint32_t* tree, uint32_t count) { return compress24_top_kernel_bargs{tree, count}; } };
__global__ void __launch_bounds__(1024) compress24_top_kernel_batch(const compress24_top_kernel_bargs* __restrict__ zk_arr) { const compress24_top_kernel_bargs& zk_b = zk_arr[blockIdx.z]; compress24_top_kernel_body(zk_b.tree, zk_b.count); }

// levels above the leaf digests of a width-24 tree: wide levels one launch each, the top 2048 nodes in one
static hipError_t compress24_levels(uint32_t* tree, uint64_t rows, hipStream_t s) {
    hipError_t e = hipSuccess;
    uint32_t* level = tree;
    uint64_t cnt = rows;
    while (cnt > 2048 && e == hipSuccess) {
        uint32_t* next = level + 8 * cnt;
        ZK_LAUNCH(compress24_level_kernel, compress24_level_kernel_batch, compress24_level_kernel_bargs, dim3((unsigned)((cnt / 2 + 255) / 256)), dim3(256), 0, s, level, next, cnt / 2);
        e = hipGetLastError();
        level = next; cnt >>= 1;
    }
    if (e == hipSuccess && cnt > 1) {
        const unsigned threads = cnt / 2 < 64 ? 64 : (cnt / 2 > 1024 ? 1024 : (unsigned)(cnt / 2));
        ZK_LAUNCH(compress24_top_kernel, compress24_top_kernel_batch, compress24_top_kernel_bargs, dim3(1), dim3(threads), 0, s, level, (uint32_t)cnt);
        e = hipGetLastError();
    }
    return e;
}
hipError_t launch_merkle_p24_colmajor(const uint32_t* mat, uint32_t cols, int log_rows, uint32_t* tree, hipStream_t s) {
    const uint64_t rows = (uint64_t)1 << log_rows;
    ZK_LAUNCH(hash_cols24_kernel, hash_cols24_kernel_batch, hash_cols24_kernel_bargs, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, mat, cols, rows, tree);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return compress24_levels(tree, rows, s);
}

// row-major matrix (rows x ld words, width % 4 == 0, 16-byte aligned rows), same hash: the leaves of the
// RISC-Zero-shaped prover mode, whose committed matrices stay row-major like everything else in the prover
__device__ __forceinline__ void hash_rows24_kernel_body(const uint32_t* __restrict__ mat, uint64_t ld, uint32_t width, uint64_t rows, uint32_t* __restrict__ digests) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint4* row = reinterpret_cast<const uint4*>(mat + r * ld);
    uint32_t s[24];
#pragma unroll
    for (int i = 0; i < 24; i++) s[i] = 0u;
    const uint32_t nq = width / 4;                   // 16-byte groups in the row
    for (uint32_t q = 0; q < nq; q += 4) {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (q + i < nq) { const uint4 v = row[q + i]; s[4 * i] = v.x; s[4 * i + 1] = v.y; s[4 * i + 2] = v.z; s[4 * i + 3] = v.w; }
        p24_permute_dev(s);
    }
    uint4* d = reinterpret_cast<uint4*>(digests + r * 8);
    d[0] = make_uint4(s[0], s[1], s[2], s[3]);
    d[1] = make_uint4(s[4], s[5], s[6], s[7]);
}
__global__ void __launch_bounds__(256) hash_rows24_kernel(const uint32_t* __restrict__ mat, uint64_t ld, uint32_t width, uint64_t rows, uint32_t* __restrict__ digests) { hash_rows24_kernel_body(mat, ld, width, rows, digests); }
struct hash_rows24_kernel_bargs { const uint32_t* mat; uint64_t ld; uint32_t width; uint64_t rows; uint32_t* digests; static hash_rows24_kernel_bargs make(const uint32_t* mat, uint64_t ld, uint32_t width, uint64_t rows, uint32_t* digests) { return hash_rows24_kernel_bargs{mat, ld, width, rows, digests}; } };
__global__ void __launch_bounds__(256) hash_rows24_kernel_batch(const hash_rows24_kernel_bargs* __restrict__ zk_arr) { const hash_rows24_kernel_bargs& zk_b = zk_arr[blockIdx.z]; hash_rows24_kernel_body(zk_b.mat, zk_b.ld, zk_b.width, zk_b.rows, zk_b.digests); }

hipError_t launch_merkle_p24_rowmajor(const uint32_t* mat, uint64_t ld, uint32_t width, int log_rows, uint32_t* tree, hipStream_t s) {
    const uint64_t rows = (uint64_t)1 << log_rows;
    ZK_LAUNCH(hash_rows24_kernel, hash_rows24_kernel_batch, hash_rows24_kernel_bargs, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, mat, ld, width, rows, tree);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return compress24_levels(tree, rows, s);
}

__device__ __forceinline__ void permute_states_kernel_body(uint32_t* states, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = states[16 * i + k];
    p2_permute_dev(s);
#pragma unroll
    for (int k = 0; k < 16; k++) states[16 * i + k] = s[k];
}
__global__ void __launch_bounds__(256) permute_states_kernel(uint32_t* states, uint64_t count) { permute_states_kernel_body(states, count); }
struct permute_states_kernel_bargs { uint32_t* states; uint64_t count; static permute_states_kernel_bargs make(uint32_t* states, uint64_t count) { return permute_states_kernel_bargs{states, count}; } };
__global__ void __launch_bounds__(256) permute_states_kernel_batch(const permute_states_kernel_bargs* __restrict__ zk_arr) { const permute_states_kernel_bargs& zk_b = zk_arr[blockIdx.z]; permute_states_kernel_body(zk_b.states, zk_b.count); }

hipError_t launch_permute_states(uint32_t* states, uint64_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    ZK_LAUNCH(permute_states_kernel, permute_states_kernel_batch, permute_states_kernel_bargs, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, states, count);
    return hipGetLastError();
}

// ------------------------------------------------------------------ trace of the Poseidon2 permutation chip (p2chip.h)
// One row = one permutation with every intermediate the chip's constraints name (poseidon2_chip.cpp); Merkle paths: one lane walks one
// path from its leaf to the root, a row per level (the levels of a path depend on each other; the paths do not).
__device__ void p2chip_fill_row(uint32_t* t, const uint32_t in[16], uint32_t bit, uint32_t ch, uint32_t end, uint32_t cnt, uint32_t spg, uint32_t ss,
                                uint32_t out16[16]) {
    using namespace p2chip;
    uint32_t s[16];
    for (int i = 0; i < 16; i++) { s[i] = in[i]; t[IN + i] = in[i]; }
    p2_external_linear(s);
    for (int i = 0; i < 16; i++) t[S0 + i] = s[i];
    auto external_round = [&](int r) {
        for (int i = 0; i < 16; i++) {
            const uint32_t y = fadd(s[i], P2K.ext_rc[r][i]);
            const uint32_t x3 = fmul(fmul(y, y), y);
            t[x3e(r) + i] = x3;
            s[i] = fmul(fmul(x3, x3), y);
        }
        p2_external_linear(s);
        for (int i = 0; i < 16; i++) t[oute(r) + i] = s[i];
    };
    for (int r = 0; r < 4; r++) external_round(r);
    for (int r = 0; r < 13; r++) {
        t[s0p(r)] = s[0];
        const uint32_t y = fadd(s[0], P2K.int_rc[r]);
        const uint32_t x3 = fmul(fmul(y, y), y);
        t[x3p(r)] = x3;
        s[0] = fmul(fmul(x3, x3), y);
        t[sbp(r)] = s[0];
        p2_internal_linear(s);
    }
    for (int i = 0; i < 16; i++) t[SP + i] = s[i];
    for (int r = 4; r < 8; r++) external_round(r);
    for (int j = 0; j < 8; j++) t[D + j] = bit ? in[8 + j] : in[j];
    for (int j = 0; j < 16; j++) out16[j] = s[j];
    t[BIT] = bit ? MONTY_R1 : 0u; t[CH] = ch ? MONTY_R1 : 0u; t[END] = end ? MONTY_R1 : 0u; t[CNT] = to_monty(cnt);
    t[SPG] = spg ? MONTY_R1 : 0u; t[SS] = ss ? MONTY_R1 : 0u;
    for (uint32_t c = SS + 1; c < WIDTH; c++) t[c] = 0u;
}
__device__ __forceinline__ void p2chip_merkle_kernel_body(const p2chip::MerkleTraceArgs& a) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sponge_rows = a.row_width / 8;
    const uint64_t per_path = (uint64_t)sponge_rows + a.depth, path_rows = a.n_paths * per_path;
    if (p < a.n_paths) {
        uint32_t out[16], in[16];
        uint32_t* t = a.trace + p * per_path * a.ld;
        if (sponge_rows) {                                       // the leaf: the overwrite-mode sponge over the opened row, 8 values per row
            const uint32_t* vals = a.leaves + p * a.row_width;
            for (int j = 0; j < 16; j++) out[j] = 0u;
            for (uint32_t k = 0; k < sponge_rows; k++, t += a.ld) {
                for (int j = 0; j < 8; j++) { in[j] = to_monty(vals[8 * k + j]); in[8 + j] = out[8 + j]; }
                p2chip_fill_row(t, in, 0u, 0u, 0u, (uint32_t)p, k ? 1u : 0u, k ? 0u : 1u, out);
            }
        } else
            for (int j = 0; j < 8; j++) out[j] = to_monty(a.leaves[8 * p + j]);
        const uint32_t index = a.indices[p];
        for (uint32_t lvl = 0; lvl < a.depth; lvl++, t += a.ld) {
            const uint32_t bit = (index >> lvl) & 1u;
            const uint32_t* sib = a.siblings + 8 * (p * a.depth + lvl);
            for (int j = 0; j < 8; j++) { in[bit ? 8 + j : j] = out[j]; in[bit ? j : 8 + j] = to_monty(sib[j]); }
            const uint32_t end = lvl + 1 == a.depth ? 1u : 0u;
            p2chip_fill_row(t, in, bit, (lvl || sponge_rows) ? 1u : 0u, end, (uint32_t)p + end, 0u, 0u, out);
        }
        for (int j = 0; j < 8; j++) a.roots[8 * p + j] = from_monty(out[j]);
        return;
    }
    const uint64_t row = path_rows + (p - a.n_paths);          // the rows after the paths: permutations of the zero state, no flags
    if (row >= a.rows) return;
    uint32_t zero[16] = {0}, out[16];
    p2chip_fill_row(a.trace + row * a.ld, zero, 0u, 0u, 0u, (uint32_t)a.n_paths, 0u, 0u, out);
}
__global__ void __launch_bounds__(64) p2chip_merkle_kernel(p2chip::MerkleTraceArgs a) { p2chip_merkle_kernel_body(a); }
struct p2chip_merkle_kernel_bargs { p2chip::MerkleTraceArgs a; static p2chip_merkle_kernel_bargs make(p2chip::MerkleTraceArgs a) { return p2chip_merkle_kernel_bargs{a}; } };
__global__ void __launch_bounds__(64) p2chip_merkle_kernel_batch(const p2chip_merkle_kernel_bargs* __restrict__ zk_arr) { const p2chip_merkle_kernel_bargs& zk_b = zk_arr[blockIdx.z]; p2chip_merkle_kernel_body(zk_b.a); }

// the FRI-layers variant (p2chip.h): paths of different depths, one leaf row + depth compression rows each, with the layer number,
// the index walk and the receive multiplicity in the three spare columns
__device__ __forceinline__ void p2chip_layer_paths_kernel_body(const p2chip::LayerPathsArgs& a) {
    using namespace p2chip;
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // transcript variant: four more columns per row (TRS, QP, QF and one unused); the lanes past the paths and the padding fill the rows of
    // the sponge chain, one row each, from the input states the host walked
    const bool tv = a.n_transcript != 0;
    auto tail = [&](uint32_t* t, uint32_t trs, uint32_t qp = 0u, uint32_t qf = 0u) { if (tv) { t[TRS] = trs; t[QP] = qp; t[QF] = qf; t[TRS + 3] = 0u; } };
    const uint64_t chain0 = a.n_paths + (a.rows - a.used_rows);
    if (tv && p >= chain0) {
        const uint32_t r = (uint32_t)(p - chain0);
        if (r >= a.n_transcript + a.n_query_rows) return;
        uint32_t in[16], out[16];
        for (int j = 0; j < 16; j++) in[j] = to_monty(a.chain_inputs[16 * r + j]);
        uint32_t* t = a.trace + (uint64_t)r * a.ld;
        p2chip_fill_row(t, in, 0u, 0u, 0u, 0u, r ? 1u : 0u, 0u, out);
        t[LNP] = to_monty(r); t[KP] = 0u; t[M] = 0u;
        if (r < a.n_transcript) tail(t, MONTY_R1);
        else tail(t, 0u, MONTY_R1, r == a.n_transcript ? MONTY_R1 : 0u);
        return;
    }
    if (p < a.n_paths) {
        uint32_t out[16], in[16];
        uint32_t* t = a.trace + (uint64_t)a.starts[p] * a.ld;
        const uint32_t depth = a.depths[p], index = a.indices[p], layer = to_monty(a.layers[p]);
        for (int j = 0; j < 8; j++) { in[j] = to_monty(a.leaves[8 * p + j]); in[8 + j] = 0u; }
        p2chip_fill_row(t, in, 0u, 0u, 0u, (uint32_t)p, 0u, 1u, out);
        t[LNP] = layer; t[KP] = to_monty(2u * index); t[M] = to_monty(a.mults[p]);
        tail(t, 0u);
        t += a.ld;
        const uint32_t* sib = a.siblings + a.sib_off[p];
        for (uint32_t lvl = 0; lvl < depth; lvl++, t += a.ld) {
            const uint32_t bit = (index >> lvl) & 1u;
            for (int j = 0; j < 8; j++) { in[bit ? 8 + j : j] = out[j]; in[bit ? j : 8 + j] = to_monty(sib[8 * lvl + j]); }
            const uint32_t end = lvl + 1 == depth ? 1u : 0u;
            p2chip_fill_row(t, in, bit, 1u, end, (uint32_t)p + end, 0u, 0u, out);
            t[LNP] = layer; t[KP] = to_monty(index >> lvl); t[M] = 0u;
            tail(t, 0u);
        }
        for (int j = 0; j < 8; j++) a.roots[8 * p + j] = from_monty(out[j]);
        return;
    }
    const uint64_t row = a.used_rows + (p - a.n_paths);
    if (row >= a.rows) return;
    uint32_t zero[16] = {0}, out[16];
    uint32_t* t = a.trace + row * a.ld;
    p2chip_fill_row(t, zero, 0u, 0u, 0u, (uint32_t)a.n_paths, 0u, 0u, out);
    t[LNP] = 0u; t[KP] = 0u; t[M] = 0u;
    tail(t, 0u);
}
__global__ void __launch_bounds__(64) p2chip_layer_paths_kernel(p2chip::LayerPathsArgs a) { p2chip_layer_paths_kernel_body(a); }
struct p2chip_layer_paths_kernel_bargs { p2chip::LayerPathsArgs a; static p2chip_layer_paths_kernel_bargs make(p2chip::LayerPathsArgs a) { return p2chip_layer_paths_kernel_bargs{a}; } };
__global__ void __launch_bounds__(64) p2chip_layer_paths_kernel_batch(const p2chip_layer_paths_kernel_bargs* __restrict__ zk_arr) { const p2chip_layer_paths_kernel_bargs& zk_b = zk_arr[blockIdx.z]; p2chip_layer_paths_kernel_body(zk_b.a); }

// the shard verifier's chip (p2chip.h, P2RArgs): one lane per chain, one per transcript row, the rest share the padding rows
// ---- the same rows written COOPERATIVELY (round 6).  A lane that fills its own 360-word row stores 4 bytes at a time 1 440 bytes away from its neighbours' stores:
// 2^21 rows took 8 ms for 3 GB.  Here the wave's 64 lanes produce a group of 16 columns each (IN, S0, a round's cubes, a round's state ...: the row's natural
// units), park them in LDS (pitch 17: conflict-free both ways) and write them out as 16-byte stores, sixteen rows x 64 bytes per instruction.  One wave per
// workgroup; every lane takes part in every store (a lane without a row passes ~0 as its offset), so the callers loop wave-uniformly.
__device__ __forceinline__ void p2r_coop_put16(uint32_t* stage, const uint64_t* rowoff, uint32_t* trace, uint32_t col, const uint32_t v[16], bool vec, int n = 16) {
    const int l = (int)threadIdx.x;
#pragma unroll
    for (int j = 0; j < 16; j++) stage[l * 17 + j] = v[j];
    __syncthreads();
    if (vec) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = 16 * k + (l >> 2), q = l & 3;
            const uint64_t off = rowoff[r];
            const uint32_t* sp = stage + r * 17 + 4 * q;
            if (off != ~0ull && 4 * q < n) *reinterpret_cast<uint4*>(trace + off + col + 4 * q) = make_uint4(sp[0], sp[1], sp[2], sp[3]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int r = 4 * k + (l >> 4), j = l & 15;
            const uint64_t off = rowoff[r];
            if (off != ~0ull && j < n) trace[off + col + j] = stage[r * 17 + j];
        }
    }
    __syncthreads();
}
// one row per lane, the shard-verifier machines' flags (bit; KP in column R_KP as a Montgomery word; every other flag zero): what p2chip_fill_row(t, in, bit, 0 ...)
// followed by t[R_KP] = kp writes, word for word
__device__ void p2chip_fill_row_coop(uint32_t* stage, uint64_t* rowoff, uint32_t* trace, uint64_t my_off, bool vec, const uint32_t in[16], uint32_t bit, uint32_t kp_monty,
                                     uint32_t out16[16]) {
    using namespace p2chip;
    static_assert(R_WIDTH == 360 && SP == 327 && D == 343 && BIT == 351 && R_KP == 352 && s0p(0) == 288 && oute(7) == 272, "the cooperative writer knows the row's layout");
    __syncthreads();                                             // (the previous row's stores read rowoff)
    rowoff[threadIdx.x] = my_off;
    uint32_t s[16], x3[16], tail[72];
    for (int i = 0; i < 16; i++) s[i] = in[i];
    p2r_coop_put16(stage, rowoff, trace, IN, s, vec);
    p2_external_linear(s);
    p2r_coop_put16(stage, rowoff, trace, S0, s, vec);
    auto external_round = [&](int r) {
        for (int i = 0; i < 16; i++) {
            const uint32_t y = fadd(s[i], P2K.ext_rc[r][i]);
            x3[i] = fmul(fmul(y, y), y);
            s[i] = fmul(fmul(x3[i], x3[i]), y);
        }
        p2r_coop_put16(stage, rowoff, trace, x3e((uint32_t)r), x3, vec);
        p2_external_linear(s);
        p2r_coop_put16(stage, rowoff, trace, oute((uint32_t)r), s, vec);
    };
#pragma unroll 1
    for (int r = 0; r < 4; r++) external_round(r);
#pragma unroll
    for (int r = 0; r < 13; r++) {
        tail[3 * r] = s[0];
        const uint32_t y = fadd(s[0], P2K.int_rc[r]);
        const uint32_t c3 = fmul(fmul(y, y), y);
        tail[3 * r + 1] = c3;
        s[0] = fmul(fmul(c3, c3), y);
        tail[3 * r + 2] = s[0];
        p2_internal_linear(s);
    }
    for (int i = 0; i < 16; i++) tail[39 + i] = s[i];            // SP = 327 = 288 + 39
#pragma unroll 1
    for (int r = 4; r < 8; r++) external_round(r);
    for (int j = 0; j < 8; j++) tail[55 + j] = bit ? in[8 + j] : in[j];      // D = 343
    tail[63] = bit ? MONTY_R1 : 0u;                              // BIT = 351
    tail[64] = kp_monty;                                         // R_KP = 352
    for (int j = 65; j < 72; j++) tail[j] = 0u;
    p2r_coop_put16(stage, rowoff, trace, 288, tail, vec);
    p2r_coop_put16(stage, rowoff, trace, 304, tail + 16, vec);
    p2r_coop_put16(stage, rowoff, trace, 320, tail + 32, vec);
    p2r_coop_put16(stage, rowoff, trace, 336, tail + 48, vec);
    p2r_coop_put16(stage, rowoff, trace, 352, tail + 64, vec, 8);
    for (int j = 0; j < 16; j++) out16[j] = s[j];
}

__device__ __forceinline__ void p2r_rows_kernel_body(const p2chip::P2RArgs& a) {
    using namespace p2chip;
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t in[16], out[16];
    if (a.ld == R_WIDTH) {
        // every lane has `total` rows to fill (a chain: its leaf blocks, then its levels; a transcript or padding row: one; nothing: zero) and the wave walks them together
        __shared__ uint32_t stage[64 * 17];
        __shared__ uint64_t rowoff[64];
        const bool vec = ((uintptr_t)a.trace & 15) == 0;
        uint32_t kind = 0, total = 0, blocks = 0, index = 0;
        uint64_t row0 = 0, r = 0;
        const uint32_t *vals = nullptr, *sib = nullptr;
        if (p < a.n_chains) {
            const uint32_t* d = a.desc + 6 * p;
            kind = 1; row0 = d[0]; blocks = d[1]; index = d[4]; total = blocks + d[3];
            vals = a.data + d[2]; sib = a.data + d[5];
        } else {
            r = p - a.n_chains;
            if (r < a.n_transcript) {
                if (!(a.seg_rows && (r % a.seg_rows) < a.skip_first)) { kind = 2; total = 1; row0 = a.trows ? a.trows[r] : r; }
            } else if (a.used_rows + (r - a.n_transcript) < a.rows) { kind = 3; total = 1; row0 = a.used_rows + (r - a.n_transcript); }
        }
        for (int j = 0; j < 16; j++) out[j] = 0u;
        for (uint32_t step = 0; __any(step < total); step++) {
            const bool valid = step < total;
            uint32_t bit = 0u, kp = 0u;
            for (int j = 0; j < 16; j++) in[j] = 0u;
            if (valid && kind == 1) {
                if (step < blocks) {
                    for (int j = 0; j < 8; j++) { in[j] = to_monty(vals[8 * step + j]); in[8 + j] = out[8 + j]; }
                    kp = step + 1 == blocks ? to_monty(2u * index) : 0u;
                } else {
                    const uint32_t lvl = step - blocks;
                    bit = (index >> lvl) & 1u;
                    for (int j = 0; j < 8; j++) { in[bit ? 8 + j : j] = out[j]; in[bit ? j : 8 + j] = to_monty(sib[8 * lvl + j]); }
                    kp = to_monty(index >> lvl);
                }
            } else if (valid && kind == 2) {
                for (int j = 0; j < 16; j++) in[j] = a.inputs_monty ? a.chain_inputs[16 * r + j] : to_monty(a.chain_inputs[16 * r + j]);
                bit = a.row_bits ? a.row_bits[r] : 0u;
                kp = a.row_kps ? to_monty(a.row_kps[r]) : 0u;
            }
            uint32_t o2[16];
            p2chip_fill_row_coop(stage, rowoff, a.trace, valid ? (row0 + step) * a.ld : ~0ull, vec, in, bit, kp, o2);
            if (valid) for (int j = 0; j < 16; j++) out[j] = o2[j];        // (a lane whose rows are done keeps its last state: the chain's root)
        }
        if (kind == 1) for (int j = 0; j < 8; j++) a.roots[8 * p + j] = from_monty(out[j]);
        return;
    }
    if (p < a.n_chains) {
        const uint32_t* d = a.desc + 6 * p;
        uint32_t* t = a.trace + (uint64_t)d[0] * a.ld;
        const uint32_t blocks = d[1], depth = d[3], index = d[4];
        const uint32_t *vals = a.data + d[2], *sib = a.data + d[5];
        for (int j = 0; j < 16; j++) out[j] = 0u;
        for (uint32_t k = 0; k < blocks; k++, t += a.ld) {
            for (int j = 0; j < 8; j++) { in[j] = to_monty(vals[8 * k + j]); in[8 + j] = out[8 + j]; }
            p2chip_fill_row(t, in, 0u, 0u, 0u, 0u, 0u, 0u, out);
            t[R_KP] = k + 1 == blocks ? to_monty(2u * index) : 0u;
        }
        for (uint32_t lvl = 0; lvl < depth; lvl++, t += a.ld) {
            const uint32_t bit = (index >> lvl) & 1u;
            for (int j = 0; j < 8; j++) { in[bit ? 8 + j : j] = out[j]; in[bit ? j : 8 + j] = to_monty(sib[8 * lvl + j]); }
            p2chip_fill_row(t, in, bit, 0u, 0u, 0u, 0u, 0u, out);
            t[R_KP] = to_monty(index >> lvl);
        }
        for (int j = 0; j < 8; j++) a.roots[8 * p + j] = from_monty(out[j]);
        return;
    }
    const uint64_t r = p - a.n_chains;
    if (r < a.n_transcript) {
        if (a.seg_rows && (r % a.seg_rows) < a.skip_first) return;            // (a proof's transcript rows: filled by another launch, from the host's walk)
        for (int j = 0; j < 16; j++) in[j] = a.inputs_monty ? a.chain_inputs[16 * r + j] : to_monty(a.chain_inputs[16 * r + j]);
        uint32_t* t = a.trace + (uint64_t)(a.trows ? a.trows[r] : r) * a.ld;
        p2chip_fill_row(t, in, a.row_bits ? a.row_bits[r] : 0u, 0u, 0u, 0u, 0u, 0u, out);
        if (a.row_kps) t[R_KP] = to_monty(a.row_kps[r]);
        return;
    }
    const uint64_t row = a.used_rows + (r - a.n_transcript);
    if (row >= a.rows) return;
    for (int j = 0; j < 16; j++) in[j] = 0u;
    p2chip_fill_row(a.trace + row * a.ld, in, 0u, 0u, 0u, 0u, 0u, 0u, out);
}
__global__ void __launch_bounds__(64) p2r_rows_kernel(p2chip::P2RArgs a) { p2r_rows_kernel_body(a); }
struct p2r_rows_kernel_bargs { p2chip::P2RArgs a; static p2r_rows_kernel_bargs make(p2chip::P2RArgs a) { return p2r_rows_kernel_bargs{a}; } };
__global__ void __launch_bounds__(64) p2r_rows_kernel_batch(const p2r_rows_kernel_bargs* __restrict__ zk_arr) { const p2r_rows_kernel_bargs& zk_b = zk_arr[blockIdx.z]; p2r_rows_kernel_body(zk_b.a); }
// machine mode's chains (p2chip.h, MrecChainArgs): what machine_verifier.inl's fill_proof walks on the host (tree_q / fri_q), one lane per chain
__global__ void __launch_bounds__(64) mrec_chains_kernel(p2chip::MrecChainArgs a) {
    using namespace p2chip;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per_q = a.n_trees + a.R;
    if (gid >= (uint64_t)a.NP * a.Q * per_q) return;
    const uint32_t c = (uint32_t)(gid % per_q), q = (uint32_t)((gid / per_q) % a.Q), p = (uint32_t)(gid / ((uint64_t)per_q * a.Q));
    const uint32_t* w = a.proofs + (uint64_t)p * a.proof_words;
    const uint32_t qidx = a.vals[(uint64_t)p * a.vals_stride + q];
    const uint32_t* qw = w + a.o_queries + (uint64_t)q * a.per_query;
    uint32_t in[16], out[16];
    // a row = its input state, direction bit and KP column (the row kernel fills the 360 columns from them, one lane per row); out = the permutation
    auto emit = [&](uint64_t row, uint32_t bit, uint32_t kp) {
        uint32_t* d = a.row_in + 16 * row;
        for (int j = 0; j < 16; j++) { d[j] = in[j]; out[j] = in[j]; }
        a.row_bit[row] = bit; a.row_kp[row] = kp;
        p2_permute_dev(out);
    };
    if (c >= a.n_trees) {                                          // ---- a FRI layer: the pair's leaf, then its path
        const uint32_t l = c - a.n_trees;
        const uint64_t at = ((uint64_t)p * a.Q + q) * a.R + l;
        const uint32_t k = a.pair_k[at];
        uint64_t fat = a.fri_off;
        for (uint32_t i = 0; i < l; i++) fat += 4 + 8 * (uint64_t)(a.H - 1 - i);
        const uint32_t* path = qw + fat + 4;
        const uint64_t r0 = (uint64_t)l + (uint64_t)l * (uint64_t)(2 * (a.H - 1) - (l - 1)) / 2;      // rows of the layers before: l leaves + sum_{i < l} (H - 1 - i) path rows
        uint64_t row = (uint64_t)p * a.p2_rows + a.p2_fri0 + (uint64_t)q * a.fri_rows + r0;
        for (int j = 0; j < 8; j++) { in[j] = a.pairs[8 * at + (uint64_t)j]; in[8 + j] = 0u; }
        emit(row++, 0u, 2u * k);
        const uint32_t depth = a.H - 1 - l;
        for (uint32_t lvl = 0; lvl < depth; lvl++) {
            const uint32_t b = (k >> lvl) & 1u;
            for (int j = 0; j < 8; j++) { in[b ? 8 + j : j] = out[j]; in[b ? j : 8 + j] = to_monty(path[8 * lvl + j]); }
            emit(row++, b, k >> lvl);
        }
        bool ok = true;
        for (int j = 0; j < 8; j++) ok = ok && from_monty(out[j]) == w[a.o_lroots + 8 * l + (uint32_t)j];
        if (!ok) atomicCAS(a.err + p, 0u, 1u);
        return;
    }
    // ---- a commitment: sponges (shorter heights first, the tallest last), then the path with the injections
    const MrecTreePlan& tp = a.trees[c];
    const uint32_t index = qidx >> tp.shift;
    uint64_t row = (uint64_t)p * a.p2_rows + tp.row0 + (uint64_t)q * tp.rows_per_query;
    uint32_t dg[16][8];
    for (uint32_t sidx = 0; sidx < tp.n_sponges; sidx++) {
        const uint32_t words = tp.sp_words[sidx], nb = (words + 7) / 8;
        const int32_t* so = a.src + tp.sp_src[sidx];
        for (int j = 0; j < 16; j++) in[j] = 0u;
        for (uint32_t b = 0; b < nb; b++) {
            const uint32_t kk = words - 8 * b < 8 ? words - 8 * b : 8u;
            for (uint32_t j = 0; j < kk; j++) { const int32_t o = so[8 * b + j]; in[j] = o < 0 ? 0u : to_monty(qw[o]); }
            emit(row++, 0u, (b == nb - 1 && sidx + 1 == tp.n_sponges) ? 2u * index : 0u);
            for (int j = 0; j < 16; j++) in[j] = out[j];                                                // the sponge's state goes on (overwrite mode)
        }
        for (int j = 0; j < 8; j++) dg[sidx][j] = in[j];
    }
    uint32_t cur[8];
    for (int j = 0; j < 8; j++) cur[j] = dg[tp.n_sponges - 1][j];
    const uint32_t* path = qw + tp.path_off;
    for (uint32_t lvl = 0; lvl < tp.depth; lvl++) {
        const uint32_t b = (index >> lvl) & 1u;
        for (int j = 0; j < 8; j++) { in[b ? 8 + j : j] = cur[j]; in[b ? j : 8 + j] = to_monty(path[8 * lvl + j]); }
        emit(row++, b, index >> lvl);
        for (int j = 0; j < 8; j++) cur[j] = out[j];
        const int32_t js = tp.inj[lvl];
        if (js >= 0) {
            for (int j = 0; j < 8; j++) { in[j] = cur[j]; in[8 + j] = dg[js][j]; }
            emit(row++, 0u, index >> (lvl + 1));
            for (int j = 0; j < 8; j++) cur[j] = out[j];
        }
    }
    bool ok = true;
    for (int j = 0; j < 8; j++) ok = ok && from_monty(cur[j]) == (tp.root_off < 0 ? a.key_root[j] : w[tp.root_off + j]);
    if (!ok) atomicCAS(a.err + p, 0u, 3u);
}
static hipError_t launch_mrec_chains16(const p2chip::MrecChainArgs& a, uint64_t n, hipStream_t s);
hipError_t launch_mrec_chains(const p2chip::MrecChainArgs& a, hipStream_t s) {
    const uint64_t n = (uint64_t)a.NP * a.Q * (a.n_trees + a.R);
    if (n == 0) return hipSuccess;
    if (a.trace && a.ld == p2chip::R_WIDTH) return launch_mrec_chains16(a, n, s);
    hipLaunchKernelGGL(mrec_chains_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, a);
    return hipGetLastError();
}

// The rows after the used ones are all the permutation of the zero state: ONE lane fills the first of them, and this kernel replicates it over the
// rest -- consecutive rows are one contiguous block, so every store instruction writes 1 KB of consecutive bytes (a lane that fills its own row
// stores 16 bytes at a stride of 1 440: 640 GB/s, docs/RECURSION_NEXT.md)
__device__ __forceinline__ void p2r_pad_rows_kernel_body(uint32_t* trace, uint32_t ld, uint64_t src_row, uint64_t n_rows) {
    const uint64_t per_row = ld / 4, i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * per_row) return;
    const uint4 v = ((const uint4*)(trace + src_row * ld))[i % per_row];
    ((uint4*)(trace + (src_row + 1) * ld))[i] = v;
}
__global__ void __launch_bounds__(256) p2r_pad_rows_kernel(uint32_t* trace, uint32_t ld, uint64_t src_row, uint64_t n_rows) { p2r_pad_rows_kernel_body(trace, ld, src_row, n_rows); }
struct p2r_pad_rows_kernel_bargs { uint32_t* trace; uint32_t ld; uint64_t src_row, n_rows; static p2r_pad_rows_kernel_bargs make(uint32_t* trace, uint32_t ld, uint64_t src_row, uint64_t n_rows) { return p2r_pad_rows_kernel_bargs{trace, ld, src_row, n_rows}; } };
__global__ void __launch_bounds__(256) p2r_pad_rows_kernel_batch(const p2r_pad_rows_kernel_bargs* __restrict__ zk_arr) { const p2r_pad_rows_kernel_bargs& zk_b = zk_arr[blockIdx.z]; p2r_pad_rows_kernel_body(zk_b.trace, zk_b.ld, zk_b.src_row, zk_b.n_rows); }
// ---- the chains (a commitment opening: leaf sponge blocks, then the path's levels) with SIXTEEN lanes per chain (round 6).  A chain is ~100 dependent
// permutations; with one lane per chain a wave waits for its longest chain at ~8 us per row.  Here state word i lives in lane i of a DPP row (the latency form of the
// tree kernels above, in canonical arithmetic because the row's columns ARE the intermediates): a row costs ~1 000 instructions instead of 4 700 in sequence, and a
// round's sixteen cubes / state words go out as one 64-byte store.
ZK_D uint32_t coopc_external_linear(uint32_t x) {
    const uint32_t r1 = dpp<0x39>(x), r2 = dpp<0x4E>(x), r3 = dpp<0x93>(x);
    const uint32_t sum = fadd(fadd(x, r1), fadd(r2, r3));
    const uint32_t y = fadd(fadd(sum, x), fadd(r1, r1));
    uint32_t t = fadd(y, dpp<0x124>(y));
    t = fadd(t, dpp<0x128>(t));
    return fadd(y, t);
}
// one row of the shard-verifier machines' Poseidon2 chip from lane l's word of the input state; returns lane l's word of the output state
struct Coop16Consts { uint32_t rc[8]; uint32_t diag; };          // lane l's round constants and diagonal entry: loaded once per chain, not once per row
ZK_D Coop16Consts coop16_load_consts(int l) {
    Coop16Consts k;
#pragma unroll
    for (int r = 0; r < 8; r++) k.rc[r] = P2K.ext_rc[r][l];
    k.diag = P2K.diag[l];
    return k;
}
ZK_D uint32_t p2chip_fill_row16(uint32_t* t, uint32_t in, int l, uint32_t bit, uint32_t kp_monty, const Coop16Consts& kc) {
    using namespace p2chip;
    uint32_t x = in;
    t[IN + l] = x;
    x = coopc_external_linear(x);
    t[S0 + l] = x;
    auto external_round = [&](int r) {
        const uint32_t y = fadd(x, kc.rc[r]);
        const uint32_t c3 = fmul(fmul(y, y), y);
        t[x3e((uint32_t)r) + l] = c3;
        x = coopc_external_linear(fmul(fmul(c3, c3), y));
        t[oute((uint32_t)r) + l] = x;
    };
#pragma unroll
    for (int r = 0; r < 4; r++) external_round(r);
#pragma unroll 1
    for (int r = 0; r < 13; r++) {
        const uint32_t y = fadd(x, P2K.int_rc[r]);
        const uint32_t c3 = fmul(fmul(y, y), y), sb = fmul(fmul(c3, c3), y);
        if (l == 0) { t[s0p((uint32_t)r)] = x; t[x3p((uint32_t)r)] = c3; t[sbp((uint32_t)r)] = sb; }
        x = l == 0 ? sb : x;
        uint32_t u = fadd(x, dpp<0x128>(x));
        u = fadd(u, dpp<0x124>(u));
        u = fadd(u, dpp<0x122>(u));
        u = fadd(u, dpp<0x121>(u));                               // every lane holds the sum
        x = fadd(fmul(x, kc.diag), u);
    }
    t[SP + l] = x;
#pragma unroll
    for (int r = 4; r < 8; r++) external_round(r);
    const uint32_t in_sw = dpp<0x128>(in);                        // lane l: word (l + 8) mod 16
    if (l < 8) t[D + l] = bit ? in_sw : in;
    if (l < 9) t[BIT + l] = l == 0 ? (bit ? MONTY_R1 : 0u) : (l == 1 ? kp_monty : 0u);      // BIT, R_KP, then the unused flags
    return x;
}
__device__ __forceinline__ void p2r_chains16_kernel_body(const p2chip::P2RArgs& a) {
    using namespace p2chip;
    static_assert(R_KP == BIT + 1 && R_WIDTH == BIT + 9, "the sixteen-lane writer knows the flags' places");
    const uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l = (int)(threadIdx.x & 15u);
    if (g >= a.n_chains) return;                                  // (whole DPP rows leave together)
    const uint32_t* d = a.desc + 6 * g;
    uint32_t* t = a.trace + (uint64_t)d[0] * a.ld;
    const uint32_t blocks = d[1], depth = d[3], index = d[4];
    const uint32_t *vals = a.data + d[2], *sib = a.data + d[5];
    const Coop16Consts kc = coop16_load_consts(l);
    uint32_t out = 0u;
    for (uint32_t k = 0; k < blocks; k++, t += a.ld) {
        const uint32_t in = l < 8 ? to_monty(vals[8 * k + (uint32_t)l]) : out;
        out = p2chip_fill_row16(t, in, l, 0u, k + 1 == blocks ? to_monty(2u * index) : 0u, kc);
    }
    for (uint32_t lvl = 0; lvl < depth; lvl++, t += a.ld) {
        const uint32_t bit = (index >> lvl) & 1u;
        const uint32_t osw = dpp<0x128>(out), sv = to_monty(sib[8 * lvl + (uint32_t)(l & 7)]);
        const uint32_t in = bit ? (l < 8 ? sv : osw) : (l < 8 ? out : sv);
        out = p2chip_fill_row16(t, in, l, bit, to_monty(index >> lvl), kc);
    }
    if (l < 8) a.roots[8 * g + (uint64_t)l] = from_monty(out);
}
__global__ void __launch_bounds__(256) p2r_chains16_kernel(p2chip::P2RArgs a) { p2r_chains16_kernel_body(a); }
struct p2r_chains16_kernel_bargs { p2chip::P2RArgs a; static p2r_chains16_kernel_bargs make(p2chip::P2RArgs a) { return p2r_chains16_kernel_bargs{a}; } };
__global__ void __launch_bounds__(256) p2r_chains16_kernel_batch(const p2r_chains16_kernel_bargs* __restrict__ zk_arr) { const p2r_chains16_kernel_bargs& zk_b = zk_arr[blockIdx.z]; p2r_chains16_kernel_body(zk_b.a); }

// machine mode's chains the same way (mrec_chains_kernel above is the one-lane-per-chain form that hands rows to the row kernel): sixteen lanes per chain, every row
// written where it is computed.  A commitment's sponge digests wait in LDS for their injection levels.
__global__ void __launch_bounds__(256) mrec_chains16_kernel(p2chip::MrecChainArgs a) {
    using namespace p2chip;
    __shared__ uint32_t dgs[16][16][8];
    const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l = (int)(threadIdx.x & 15u), grp = (int)(threadIdx.x >> 4);
    const uint32_t per_q = a.n_trees + a.R;
    if (gid >= (uint64_t)a.NP * a.Q * per_q) return;              // (whole DPP rows leave together)
    const uint32_t c = (uint32_t)(gid % per_q), q = (uint32_t)((gid / per_q) % a.Q), p = (uint32_t)(gid / ((uint64_t)per_q * a.Q));
    const uint32_t* w = a.proofs + (uint64_t)p * a.proof_words;
    const uint32_t qidx = a.vals[(uint64_t)p * a.vals_stride + q];
    const uint32_t* qw = w + a.o_queries + (uint64_t)q * a.per_query;
    uint32_t out = 0u;
    const Coop16Consts kc = coop16_load_consts(l);
    auto emit = [&](uint64_t row, uint32_t in, uint32_t bit, uint32_t kp) { out = p2chip_fill_row16(a.trace + row * a.ld, in, l, bit, to_monty(kp), kc); };
    // true when some lane of this chain's sixteen says `bad` (the other chains of the wave may be elsewhere in the code: their bits are not looked at)
    auto any16 = [&](bool bad) { const unsigned long long m = __ballot(bad); return ((m >> (16u * ((threadIdx.x & 63u) >> 4))) & 0xFFFFull) != 0; };
    if (c >= a.n_trees) {                                          // ---- a FRI layer: the pair's leaf, then its path
        const uint32_t lay = c - a.n_trees;
        const uint64_t at = ((uint64_t)p * a.Q + q) * a.R + lay;
        const uint32_t k = a.pair_k[at];
        uint64_t fat = a.fri_off;
        for (uint32_t i = 0; i < lay; i++) fat += 4 + 8 * (uint64_t)(a.H - 1 - i);
        const uint32_t* path = qw + fat + 4;
        const uint64_t r0 = (uint64_t)lay + (uint64_t)lay * (uint64_t)(2 * (a.H - 1) - (lay - 1)) / 2;
        uint64_t row = (uint64_t)p * a.p2_rows + a.p2_fri0 + (uint64_t)q * a.fri_rows + r0;
        emit(row++, l < 8 ? a.pairs[8 * at + (uint64_t)l] : 0u, 0u, 2u * k);
        const uint32_t depth = a.H - 1 - lay;
        for (uint32_t lvl = 0; lvl < depth; lvl++) {
            const uint32_t b = (k >> lvl) & 1u;
            const uint32_t osw = dpp<0x128>(out), sv = to_monty(path[8 * lvl + (uint32_t)(l & 7)]);
            emit(row++, b ? (l < 8 ? sv : osw) : (l < 8 ? out : sv), b, k >> lvl);
        }
        const bool bad = l < 8 && from_monty(out) != w[a.o_lroots + 8 * lay + (uint32_t)l];
        if (any16(bad) && l == 0) atomicCAS(a.err + p, 0u, 1u);
        return;
    }
    // ---- a commitment: sponges (shorter heights first, the tallest last), then the path with the injections
    const MrecTreePlan& tp = a.trees[c];
    const uint32_t index = qidx >> tp.shift;
    uint64_t row = (uint64_t)p * a.p2_rows + tp.row0 + (uint64_t)q * tp.rows_per_query;
    for (uint32_t sidx = 0; sidx < tp.n_sponges; sidx++) {
        const uint32_t words = tp.sp_words[sidx], nb = (words + 7) / 8;
        const int32_t* so = a.src + tp.sp_src[sidx];
        uint32_t state = 0u;                                       // (overwrite mode: a block replaces the first words of the state, the rest goes on)
        for (uint32_t b = 0; b < nb; b++) {
            const uint32_t kk = words - 8 * b < 8 ? words - 8 * b : 8u;
            uint32_t in = state;
            if ((uint32_t)l < kk) { const int32_t o = so[8 * b + (uint32_t)l]; in = o < 0 ? 0u : to_monty(qw[o]); }
            emit(row++, in, 0u, (b == nb - 1 && sidx + 1 == tp.n_sponges) ? 2u * index : 0u);
            state = out;
        }
        if (l < 8) dgs[grp][sidx][l] = out;
    }
    __threadfence_block();
    uint32_t cur = out;                                            // lanes 0 .. 7: the node on the way up
    const uint32_t* path = qw + tp.path_off;
    for (uint32_t lvl = 0; lvl < tp.depth; lvl++) {
        const uint32_t b = (index >> lvl) & 1u;
        const uint32_t csw = dpp<0x128>(cur), sv = to_monty(path[8 * lvl + (uint32_t)(l & 7)]);
        emit(row++, b ? (l < 8 ? sv : csw) : (l < 8 ? cur : sv), b, index >> lvl);
        cur = out;
        const int32_t js = tp.inj[lvl];
        if (js >= 0) {
            emit(row++, l < 8 ? cur : dgs[grp][js][l - 8], 0u, index >> (lvl + 1));
            cur = out;
        }
    }
    const bool bad = l < 8 && from_monty(cur) != (tp.root_off < 0 ? a.key_root[l] : w[tp.root_off + l]);
    if (any16(bad) && l == 0) atomicCAS(a.err + p, 0u, 3u);
}
static hipError_t launch_mrec_chains16(const p2chip::MrecChainArgs& a, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(mrec_chains16_kernel, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_p2r_rows(const p2chip::P2RArgs& a, hipStream_t s) {
    if (a.n_chains && a.ld == p2chip::R_WIDTH) {
        // the chains sixteen lanes each; everything else (transcript rows, padding) through the row kernel below, which then sees no chain
        ZK_LAUNCH(p2r_chains16_kernel, p2r_chains16_kernel_batch, p2r_chains16_kernel_bargs, dim3((unsigned)(((uint64_t)a.n_chains * 16 + 255) / 256)), dim3(256), 0, s, a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        p2chip::P2RArgs rest = a;
        rest.n_chains = 0;
        if (rest.n_transcript == 0 && rest.rows == rest.used_rows) return hipSuccess;
        return launch_p2r_rows(rest, s);
    }
    p2chip::P2RArgs b = a;
    const uint64_t pad = a.rows - a.used_rows;
    const bool replicate = pad > 1 && a.ld % 4 == 0 && ((uintptr_t)a.trace & 15) == 0;
    if (replicate) b.rows = a.used_rows + 1;                    // (the kernel fills the first padding row only)
    const uint64_t lanes = (uint64_t)b.n_chains + b.n_transcript + (b.rows - b.used_rows);
    ZK_LAUNCH(p2r_rows_kernel, p2r_rows_kernel_batch, p2r_rows_kernel_bargs, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s, b);
    if (replicate) {
        const uint64_t vecs = (pad - 1) * (a.ld / 4);
        ZK_LAUNCH(p2r_pad_rows_kernel, p2r_pad_rows_kernel_batch, p2r_pad_rows_kernel_bargs, dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, s, a.trace, (uint32_t)a.ld, (uint64_t)a.used_rows, pad - 1);
    }
    return hipGetLastError();
}

hipError_t launch_p2chip_layer_paths(const p2chip::LayerPathsArgs& a, hipStream_t s) {
    const uint64_t lanes = a.n_paths + (a.rows - a.used_rows) + a.n_transcript + a.n_query_rows;
    ZK_LAUNCH(p2chip_layer_paths_kernel, p2chip_layer_paths_kernel_batch, p2chip_layer_paths_kernel_bargs, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_p2chip_merkle(const p2chip::MerkleTraceArgs& a, hipStream_t s) {
    const uint64_t lanes = a.n_paths + (a.rows - a.n_paths * ((uint64_t)a.row_width / 8 + a.depth));
    ZK_LAUNCH(p2chip_merkle_kernel, p2chip_merkle_kernel_batch, p2chip_merkle_kernel_bargs, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace zk
