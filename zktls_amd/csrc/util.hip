// util.hip -- synthetic shard generation on the device and form conversion.
//
// The synthetic workload is the one SURVEY.md section 8(d) / BASELINE.md fix for the
// benchmark ("synthetic 2^20-row shards"): the reference's real trace comes from
// executing the zkTLS guest ELF inside sp1-core-executor, which is out of scope and
// unobtainable offline (crates/guest-prover-sp1/src/sp1.rs:113,116 are the call sites).
// Values are a counter-based splitmix64 stream reduced mod p, written in Montgomery form.
#include "babybear.cuh"
#include "kernels.h"
#include "batch.h"

namespace zk {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// canonical value: the (index+1)-th splitmix64 output for `seed`, mod p
__device__ __forceinline__ uint32_t synth_value(uint64_t seed, uint64_t index) {
    return (uint32_t)(mix64(seed + index * 0x9E3779B97F4A7C15ull) % P);
}

__device__ __forceinline__ void fill_uniform_kernel_body(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width) {
    const uint64_t total = rows * width;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = i / width, c = i % width;
        out[r * ld + c] = to_monty(synth_value(seed, i));
    }
}
__global__ void fill_uniform_kernel(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width) { fill_uniform_kernel_body(out, ld, seed, rows, width); }
struct fill_uniform_kernel_bargs { uint32_t* out; uint64_t ld; uint64_t seed; uint64_t rows; uint32_t width; static fill_uniform_kernel_bargs make(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width) { return fill_uniform_kernel_bargs{out, ld, seed, rows, width}; } };
__global__ void fill_uniform_kernel_batch(const fill_uniform_kernel_bargs* __restrict__ zk_arr) { const fill_uniform_kernel_bargs& zk_b = zk_arr[blockIdx.z]; fill_uniform_kernel_body(zk_b.out, zk_b.ld, zk_b.seed, zk_b.rows, zk_b.width); }

hipError_t launch_fill_uniform(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width, hipStream_t s) {
    if (rows == 0 || width == 0) return hipSuccess;
    uint64_t total = rows * width;
    unsigned blocks = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ZK_LAUNCH(fill_uniform_kernel, fill_uniform_kernel_batch, fill_uniform_kernel_bargs, dim3(blocks), dim3(256), 0, s, out, ld, seed, rows, width);
    return hipGetLastError();
}

// synthetic AIR trace: group g = columns 4g..4g+3 = (a, b, c, d)
//   a, b uniform;  c = a*a*b + (g+1);  d[0] = 5g+7;  d[i] = a[i-1]*b[i-1] + c[i-1] + (2g+3)
__device__ __forceinline__ void gen_trace_kernel_body(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width) {
    const uint32_t G = width / 4;
    const uint64_t total = rows * G;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = t / G;
        const uint32_t g = (uint32_t)(t % G);
        const uint32_t k1 = to_monty((g + 1) % P), k2 = to_monty((2 * g + 3) % P);
        uint32_t a = to_monty(synth_value(seed, i * width + 4 * g));
        uint32_t b = to_monty(synth_value(seed, i * width + 4 * g + 1));
        uint32_t c = fadd(fmul(fmul(a, a), b), k1);
        uint32_t d;
        if (i == 0) d = to_monty((5 * g + 7) % P);
        else {
            uint32_t pa = to_monty(synth_value(seed, (i - 1) * width + 4 * g));
            uint32_t pb = to_monty(synth_value(seed, (i - 1) * width + 4 * g + 1));
            uint32_t pc = fadd(fmul(fmul(pa, pa), pb), k1);
            d = fadd(fadd(fmul(pa, pb), pc), k2);
        }
        uint4 v = make_uint4(a, b, c, d);
        uint32_t* p = out + i * ld + 4 * g;
        if ((ld & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) *reinterpret_cast<uint4*>(p) = v;
        else { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
    }
}
__global__ void gen_trace_kernel(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width) { gen_trace_kernel_body(out, ld, seed, rows, width); }
struct gen_trace_kernel_bargs { uint32_t* out; uint64_t ld; uint64_t seed; uint64_t rows; uint32_t width; static gen_trace_kernel_bargs make(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width) { return gen_trace_kernel_bargs{out, ld, seed, rows, width}; } };
__global__ void gen_trace_kernel_batch(const gen_trace_kernel_bargs* __restrict__ zk_arr) { const gen_trace_kernel_bargs& zk_b = zk_arr[blockIdx.z]; gen_trace_kernel_body(zk_b.out, zk_b.ld, zk_b.seed, zk_b.rows, zk_b.width); }

hipError_t launch_gen_trace(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width, hipStream_t s) {
    if (rows == 0 || width < 4) return hipSuccess;
    uint64_t total = rows * (width / 4);
    unsigned blocks = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ZK_LAUNCH(gen_trace_kernel, gen_trace_kernel_batch, gen_trace_kernel_bargs, dim3(blocks), dim3(256), 0, s, out, ld, seed, rows, width);
    return hipGetLastError();
}

// same AIR, but the odd group of each of the first `pairs` group pairs RECEIVES the even group's
// (a, b) under the row permutation pi(i) = 5 i + 3 mod N (the LogUp workload)
// recv_seed / recv_width: the stream and row pitch the RECEIVER groups read (the table itself, or -- lookups between two
// tables of equal height -- the partner table)
__device__ __forceinline__ void gen_trace_logup_kernel_body(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width, uint32_t pairs, uint64_t recv_seed, uint32_t recv_width) {
    const uint32_t G = width / 4;
    const uint64_t total = rows * G;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = t / G;
        const uint32_t g = (uint32_t)(t % G);
        const bool recv = (g & 1u) && (g / 2u) < pairs;
        const uint32_t sg = recv ? g - 1 : g;
        const uint64_t ri = recv ? ((5 * i + 3) & (rows - 1)) : i;
        const uint32_t k1 = to_monty((g + 1) % P), k2 = to_monty((2 * g + 3) % P);
        const uint64_t sd = recv ? recv_seed : seed;
        const uint64_t sw = recv ? recv_width : width;
        uint32_t a = to_monty(synth_value(sd, ri * sw + 4 * sg));
        uint32_t b = to_monty(synth_value(sd, ri * sw + 4 * sg + 1));
        uint32_t c = fadd(fmul(fmul(a, a), b), k1);
        uint32_t d;
        if (i == 0) d = to_monty((5 * g + 7) % P);
        else {
            const uint64_t pi = recv ? ((5 * (i - 1) + 3) & (rows - 1)) : i - 1;
            uint32_t pa = to_monty(synth_value(sd, pi * sw + 4 * sg));
            uint32_t pb = to_monty(synth_value(sd, pi * sw + 4 * sg + 1));
            uint32_t pc = fadd(fmul(fmul(pa, pa), pb), k1);
            d = fadd(fadd(fmul(pa, pb), pc), k2);
        }
        uint32_t* p = out + i * ld + 4 * g;
        p[0] = a; p[1] = b; p[2] = c; p[3] = d;
    }
}
__global__ void gen_trace_logup_kernel(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width, uint32_t pairs, uint64_t recv_seed, uint32_t recv_width) { gen_trace_logup_kernel_body(out, ld, seed, rows, width, pairs, recv_seed, recv_width); }
struct gen_trace_logup_kernel_bargs { uint32_t* out; uint64_t ld; uint64_t seed; uint64_t rows; uint32_t width; uint32_t pairs; uint64_t recv_seed; uint32_t recv_width; static gen_trace_logup_kernel_bargs make(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width, uint32_t pairs, uint64_t recv_seed, uint32_t recv_width) { return gen_trace_logup_kernel_bargs{out, ld, seed, rows, width, pairs, recv_seed, recv_width}; } };
__global__ void gen_trace_logup_kernel_batch(const gen_trace_logup_kernel_bargs* __restrict__ zk_arr) { const gen_trace_logup_kernel_bargs& zk_b = zk_arr[blockIdx.z]; gen_trace_logup_kernel_body(zk_b.out, zk_b.ld, zk_b.seed, zk_b.rows, zk_b.width, zk_b.pairs, zk_b.recv_seed, zk_b.recv_width); }

hipError_t launch_gen_trace_logup(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width, uint32_t pairs,
                                  uint64_t recv_seed, uint32_t recv_width, hipStream_t s) {
    if (rows == 0 || width < 4) return hipSuccess;
    uint64_t total = rows * (width / 4);
    unsigned blocks = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ZK_LAUNCH(gen_trace_logup_kernel, gen_trace_logup_kernel_batch, gen_trace_logup_kernel_bargs, dim3(blocks), dim3(256), 0, s, out, ld, seed, rows, width, pairs, recv_seed, recv_width);
    return hipGetLastError();
}

// ------------------------------------------------------------------ layout adapter (row a11)
// out[c][r'] = in[r][c] with r' = r or bitrev(r): moves between RISC Zero's column-major
// [count][size] polynomials and the row-major [size][count] matrices the NTT kernels stream.
// 32 x 32 tiles through LDS (pitch 33): 128-byte segments on both sides.
__device__ __forceinline__ void transpose_kernel_body(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t rows, uint64_t cols, int rev_bits_in, int rev_bits_out) {
    __shared__ uint32_t tile[32][33];
    const uint64_t r0 = (uint64_t)blockIdx.y * 32, c0 = (uint64_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const uint64_t r = r0 + k, c = c0 + tx;
        if (r < rows && c < cols) {
            const uint64_t cs = rev_bits_in ? (uint64_t)(__brev((uint32_t)c) >> (32 - rev_bits_in)) : c;
            tile[k][tx] = in[r * cols + cs];
        }
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const uint64_t c = c0 + k, r = r0 + tx;                  // out[c][r]
        if (r < rows && c < cols) {
            const uint64_t cd = rev_bits_out ? (uint64_t)(__brev((uint32_t)c) >> (32 - rev_bits_out)) : c;
            out[cd * rows + r] = tile[tx][k];
        }
    }
}
__global__ void __launch_bounds__(256) transpose_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t rows, uint64_t cols, int rev_bits_in, int rev_bits_out) { transpose_kernel_body(in, out, rows, cols, rev_bits_in, rev_bits_out); }
struct transpose_kernel_bargs { const uint32_t* in; uint32_t* out; uint64_t rows; uint64_t cols; int rev_bits_in; int rev_bits_out; static transpose_kernel_bargs make(const uint32_t* in, uint32_t* out, uint64_t rows, uint64_t cols, int rev_bits_in, int rev_bits_out) { return transpose_kernel_bargs{in, out, rows, cols, rev_bits_in, rev_bits_out}; } };
__global__ void __launch_bounds__(256) transpose_kernel_batch(const transpose_kernel_bargs* __restrict__ zk_arr) { const transpose_kernel_bargs& zk_b = zk_arr[blockIdx.z]; transpose_kernel_body(zk_b.in, zk_b.out, zk_b.rows, zk_b.cols, zk_b.rev_bits_in, zk_b.rev_bits_out); }

// in: [rows][cols] row-major; out: [cols][rows].  rev_bits_in / rev_bits_out (0 = off) bit-reverse the
// COLUMN index of `in` on the read side / on the write side (it becomes the row index of `out`).
hipError_t launch_transpose(const uint32_t* in, uint32_t* out, uint64_t rows, uint64_t cols, int rev_bits_in, int rev_bits_out, hipStream_t s) {
    if (rows == 0 || cols == 0) return hipSuccess;
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    ZK_LAUNCH(transpose_kernel, transpose_kernel_batch, transpose_kernel_bargs, grid, dim3(256), 0, s, in, out, rows, cols, rev_bits_in, rev_bits_out);
    return hipGetLastError();
}

__device__ __forceinline__ void convert_kernel_body(const uint32_t* in, uint32_t* out, uint64_t n, bool to_m) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = to_m ? fmul(in[i], MONTY_R2) : from_monty(in[i]);
}
__global__ void convert_kernel(const uint32_t* in, uint32_t* out, uint64_t n, bool to_m) { convert_kernel_body(in, out, n, to_m); }
struct convert_kernel_bargs { const uint32_t* in; uint32_t* out; uint64_t n; bool to_m; static convert_kernel_bargs make(const uint32_t* in, uint32_t* out, uint64_t n, bool to_m) { return convert_kernel_bargs{in, out, n, to_m}; } };
__global__ void convert_kernel_batch(const convert_kernel_bargs* __restrict__ zk_arr) { const convert_kernel_bargs& zk_b = zk_arr[blockIdx.z]; convert_kernel_body(zk_b.in, zk_b.out, zk_b.n, zk_b.to_m); }

hipError_t launch_convert(const uint32_t* in, uint32_t* out, uint64_t n, bool to_monty_form, hipStream_t s) {
    if (n == 0) return hipSuccess;
    unsigned blocks = (unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    ZK_LAUNCH(convert_kernel, convert_kernel_batch, convert_kernel_bargs, dim3(blocks), dim3(256), 0, s, in, out, n, to_monty_form);
    return hipGetLastError();
}

// ------------------------------------------------------------------ transforms below the tile minimum (N < 32)
// By definition, one thread per output element: these sizes only occur at the edges of the operator API (the provers
// need N >= 32), so O(N^2) with N <= 16 is the simplest correct thing.
// out[row(k)][c] = scale * sum_j in[j][c] * (shift * w^k)^j,  k < 2^log_out;  w of order 2^log_out (or its inverse),
// row(k) = k or bitrev(k).  Forward DFT: log_out = log_n, shift 1.  Inverse: w^-1, scale 1/N.  LDE: coefficients in.
__device__ __forceinline__ void small_eval_kernel_body(const uint32_t* in, uint64_t in_ld, uint32_t* out, uint64_t out_ld, int log_n, int log_out, uint32_t width, uint32_t w, uint32_t shift, uint32_t scale, int bitrev_out) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t rows_out = 1u << log_out, n = 1u << log_n;
    if (idx >= rows_out * width) return;
    const uint32_t k = idx / width, c = idx % width;
    const uint32_t x = fmul(shift, fpow(w, k));
    uint32_t acc = 0;
    for (uint32_t j = n; j-- > 0;) acc = fadd(fmul(acc, x), in[(uint64_t)j * in_ld + c]);      // Horner
    const uint32_t row = bitrev_out ? (log_out ? (__brev(k) >> (32 - log_out)) : 0u) : k;
    out[(uint64_t)row * out_ld + c] = fmul(acc, scale);
}
__global__ void __launch_bounds__(256) small_eval_kernel(const uint32_t* in, uint64_t in_ld, uint32_t* out, uint64_t out_ld, int log_n, int log_out, uint32_t width, uint32_t w, uint32_t shift, uint32_t scale, int bitrev_out) { small_eval_kernel_body(in, in_ld, out, out_ld, log_n, log_out, width, w, shift, scale, bitrev_out); }
struct small_eval_kernel_bargs { const uint32_t* in; uint64_t in_ld; uint32_t* out; uint64_t out_ld; int log_n; int log_out; uint32_t width; uint32_t w; uint32_t shift; uint32_t scale; int bitrev_out; static small_eval_kernel_bargs make(const uint32_t* in, uint64_t in_ld, uint32_t* out, uint64_t out_ld, int log_n, int log_out, uint32_t width, uint32_t w, uint32_t shift, uint32_t scale, int bitrev_out) { return small_eval_kernel_bargs{in, in_ld, out, out_ld, log_n, log_out, width, w, shift, scale, bitrev_out}; } };
__global__ void __launch_bounds__(256) small_eval_kernel_batch(const small_eval_kernel_bargs* __restrict__ zk_arr) { const small_eval_kernel_bargs& zk_b = zk_arr[blockIdx.z]; small_eval_kernel_body(zk_b.in, zk_b.in_ld, zk_b.out, zk_b.out_ld, zk_b.log_n, zk_b.log_out, zk_b.width, zk_b.w, zk_b.shift, zk_b.scale, zk_b.bitrev_out); }

hipError_t launch_small_eval(const uint32_t* in, uint64_t in_ld, uint32_t* out, uint64_t out_ld, int log_n, int log_out, uint32_t width,
                             uint32_t w, uint32_t shift, uint32_t scale, int bitrev_out, hipStream_t s) {
    const uint32_t total = (1u << log_out) * width;
    ZK_LAUNCH(small_eval_kernel, small_eval_kernel_batch, small_eval_kernel_bargs, dim3((total + 255) / 256), dim3(256), 0, s, in, in_ld, out, out_ld, log_n, log_out, width, w, shift, scale, bitrev_out);
    return hipGetLastError();
}


// ---- plain copies and fills as kernels: in a lock-step batch (batch.h) the members' small copies and memsets merge into one launch
__device__ __forceinline__ void copy_bytes_kernel_body(uint8_t* dst, const uint8_t* src, uint64_t bytes) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (uint64_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)dst | (uintptr_t)src) & 3) == 0) {
        const uint64_t words = bytes >> 2;
        for (uint64_t i = tid; i < words; i += nthr) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i];
        for (uint64_t i = (words << 2) + tid; i < bytes; i += nthr) dst[i] = src[i];
    } else {
        for (uint64_t i = tid; i < bytes; i += nthr) dst[i] = src[i];
    }
}
__global__ void __launch_bounds__(256) copy_bytes_kernel(uint8_t* dst, const uint8_t* src, uint64_t bytes) { copy_bytes_kernel_body(dst, src, bytes); }
struct copy_bytes_kernel_bargs { uint8_t* dst; const uint8_t* src; uint64_t bytes; static copy_bytes_kernel_bargs make(uint8_t* dst, const uint8_t* src, uint64_t bytes) { return copy_bytes_kernel_bargs{dst, src, bytes}; } };
__global__ void __launch_bounds__(256) copy_bytes_kernel_batch(const copy_bytes_kernel_bargs* __restrict__ zk_arr) { const copy_bytes_kernel_bargs& zk_b = zk_arr[blockIdx.z]; copy_bytes_kernel_body(zk_b.dst, zk_b.src, zk_b.bytes); }

hipError_t launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    uint64_t blocks = (bytes / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    ZK_LAUNCH(copy_bytes_kernel, copy_bytes_kernel_batch, copy_bytes_kernel_bargs, dim3((unsigned)blocks), dim3(256), 0, s, (uint8_t*)dst, (const uint8_t*)src, (uint64_t)bytes);
    return hipGetLastError();
}
__device__ __forceinline__ void fill_bytes_kernel_body(uint8_t* dst, uint32_t byte, uint64_t bytes) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (uint64_t)gridDim.x * blockDim.x;
    if (((uintptr_t)dst & 3) == 0) {
        const uint64_t words = bytes >> 2;
        const uint32_t w = byte * 0x01010101u;
        for (uint64_t i = tid; i < words; i += nthr) ((uint32_t*)dst)[i] = w;
        for (uint64_t i = (words << 2) + tid; i < bytes; i += nthr) dst[i] = (uint8_t)byte;
    } else {
        for (uint64_t i = tid; i < bytes; i += nthr) dst[i] = (uint8_t)byte;
    }
}
__global__ void __launch_bounds__(256) fill_bytes_kernel(uint8_t* dst, uint32_t byte, uint64_t bytes) { fill_bytes_kernel_body(dst, byte, bytes); }
struct fill_bytes_kernel_bargs { uint8_t* dst; uint32_t byte; uint64_t bytes; static fill_bytes_kernel_bargs make(uint8_t* dst, uint32_t byte, uint64_t bytes) { return fill_bytes_kernel_bargs{dst, byte, bytes}; } };
__global__ void __launch_bounds__(256) fill_bytes_kernel_batch(const fill_bytes_kernel_bargs* __restrict__ zk_arr) { const fill_bytes_kernel_bargs& zk_b = zk_arr[blockIdx.z]; fill_bytes_kernel_body(zk_b.dst, zk_b.byte, zk_b.bytes); }

hipError_t launch_fill_bytes(void* dst, int byte, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    uint64_t blocks = (bytes / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    ZK_LAUNCH(fill_bytes_kernel, fill_bytes_kernel_batch, fill_bytes_kernel_bargs, dim3((unsigned)blocks), dim3(256), 0, s, (uint8_t*)dst, (uint32_t)(byte & 0xff), (uint64_t)bytes);
    return hipGetLastError();
}

// rows x width words between pitched matrices (what hipMemcpy2DAsync does, as a kernel that merges in a lock-step batch)
__device__ __forceinline__ void copy2d_kernel_body(uint32_t* dst, uint64_t dst_ld, const uint32_t* src, uint64_t src_ld, uint32_t width, uint64_t rows) {
    const uint64_t total = rows * width, nthr = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += nthr) {
        const uint64_t r = i / width, c = i - r * width;
        dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}
__global__ void __launch_bounds__(256) copy2d_kernel(uint32_t* dst, uint64_t dst_ld, const uint32_t* src, uint64_t src_ld, uint32_t width, uint64_t rows) { copy2d_kernel_body(dst, dst_ld, src, src_ld, width, rows); }
struct copy2d_kernel_bargs { uint32_t* dst; uint64_t dst_ld; const uint32_t* src; uint64_t src_ld; uint32_t width; uint64_t rows; static copy2d_kernel_bargs make(uint32_t* dst, uint64_t dst_ld, const uint32_t* src, uint64_t src_ld, uint32_t width, uint64_t rows) { return copy2d_kernel_bargs{dst, dst_ld, src, src_ld, width, rows}; } };
__global__ void __launch_bounds__(256) copy2d_kernel_batch(const copy2d_kernel_bargs* __restrict__ zk_arr) { const copy2d_kernel_bargs& zk_b = zk_arr[blockIdx.z]; copy2d_kernel_body(zk_b.dst, zk_b.dst_ld, zk_b.src, zk_b.src_ld, zk_b.width, zk_b.rows); }

hipError_t launch_copy2d(uint32_t* dst, uint64_t dst_ld, const uint32_t* src, uint64_t src_ld, uint32_t width, uint64_t rows, hipStream_t s) {
    if (rows == 0 || width == 0) return hipSuccess;
    uint64_t blocks = (rows * width + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ZK_LAUNCH(copy2d_kernel, copy2d_kernel_batch, copy2d_kernel_bargs, dim3((unsigned)blocks), dim3(256), 0, s, dst, dst_ld, src, src_ld, width, rows);
    return hipGetLastError();
}
}  // namespace zk
