// tools/microbench.hip -- gfx950 micro-benchmarks that size the shard-prover kernels:
//   (1) integer / fp64 VALU instruction throughput (the guides give no int-mul rate),
//   (2) Montgomery-multiply variants,
//   (3) HBM streaming with the tile access patterns of the NTT passes (chunk widths).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip
// Run on the GPU box: ./tools/microbench   (prints one line per measurement)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../zktls_amd/csrc/babybear.cuh"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ITER = 512;     // loop trips
constexpr int UNROLL = 16;    // instructions per accumulator chain per trip
constexpr int NACC = 8;       // independent chains

#define OP_KERNEL(NAME, ASM_BODY)                                                        \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {          \
        uint32_t a[NACC];                                                                \
        uint32_t b = seed | 1u, c = seed * 3u + 5u;                                      \
        _Pragma("unroll") for (int i = 0; i < NACC; i++) a[i] = threadIdx.x * 7u + i + seed; \
        for (int it = 0; it < ITER; it++) {                                              \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                         \
                _Pragma("unroll") for (int i = 0; i < NACC; i++) { ASM_BODY; }           \
            }                                                                            \
        }                                                                                \
        uint32_t r = 0;                                                                  \
        _Pragma("unroll") for (int i = 0; i < NACC; i++) r ^= a[i];                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r + c;                              \
    }

OP_KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_mul_lo_u32, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_mul_hi_u32, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_mul_u32_u24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_mad_u32_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
OP_KERNEL(k_lshl_add_u32, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_min_u32, asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_add3_u32, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
OP_KERNEL(k_sub_u32, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_and_b32, asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_xor_b32, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_lshrrev_b32, asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i])))
OP_KERNEL(k_ashrrev_i32, asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[i])))
OP_KERNEL(k_max_u32, asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_mov_b32, asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_add_lit, asm volatile("v_add_u32 %0, 0x87ffffff, %0" : "+v"(a[i])))
OP_KERNEL(k_add_sgpr, asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(seed)))
OP_KERNEL(k_add_co, asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc"))
OP_KERNEL(k_sub_co, asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc"))
OP_KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"))
OP_KERNEL(k_cmp_cnd, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"))
OP_KERNEL(k_subco_cnd, asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"))
OP_KERNEL(k_mul_hi_i32, asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_mul_lo_sgpr, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(seed)))
OP_KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %1, 5" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_bfe_u32, asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[i])))
OP_KERNEL(k_sub_dpp, asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_pk_add_u16, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_pk_mul_lo_u16, asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)))
OP_KERNEL(k_dot4_u32_u8, asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))

// 64-bit result ops
__global__ void __launch_bounds__(256) k_mad_u64_u32(uint32_t* out, uint32_t seed) {
    uint64_t a[NACC];
    uint32_t b = seed | 1u;
#pragma unroll
    for (int i = 0; i < NACC; i++) a[i] = threadIdx.x * 7u + i + seed;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                uint32_t lo = (uint32_t)a[i];
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(lo), "v"(b) : "vcc");
            }
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(r ^ (r >> 32));
}

__global__ void __launch_bounds__(256) k_fma_f64(uint32_t* out, uint32_t seed) {
    double a[NACC];
    double b = 1.0 + seed * 1e-9, c = 1e-7;
#pragma unroll
    for (int i = 0; i < NACC; i++) a[i] = threadIdx.x * 0.5 + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    double r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r;
}

__global__ void __launch_bounds__(256) k_fma_f32(uint32_t* out, uint32_t seed) {
    float a[NACC];
    float b = 1.0f + seed * 1e-9f, c = 1e-7f;
#pragma unroll
    for (int i = 0; i < NACC; i++) a[i] = threadIdx.x * 0.5f + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r;
}

// Montgomery multiply variants (compiler-scheduled): a <- a * b
__global__ void __launch_bounds__(256) k_fmul(uint32_t* out, uint32_t seed) {
    uint32_t a[NACC];
    uint32_t b = (seed | 1u) % zk::P;
#pragma unroll
    for (int i = 0; i < NACC; i++) a[i] = (threadIdx.x * 7u + i + seed) % zk::P;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) a[i] = zk::fmul(a[i], b);
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// butterfly: (a0, a1) <- (a0 + a1, (a0 - a1) * w)
__global__ void __launch_bounds__(256) k_butterfly(uint32_t* out, uint32_t seed) {
    uint32_t a[NACC];
    uint32_t b = (seed | 1u) % zk::P;
#pragma unroll
    for (int i = 0; i < NACC; i++) a[i] = (threadIdx.x * 7u + i + seed) % zk::P;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i += 2) {
                uint32_t s = zk::fadd(a[i], a[i + 1]);
                uint32_t d = zk::fmul(zk::fsub(a[i], a[i + 1]), b);
                a[i] = s; a[i + 1] = d;
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_fadd(uint32_t* out, uint32_t seed) {
    uint32_t a[NACC];
    uint32_t b = (seed | 1u) % zk::P;
#pragma unroll
    for (int i = 0; i < NACC; i++) a[i] = (threadIdx.x * 7u + i + seed) % zk::P;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) a[i] = zk::fadd(a[i], b);
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// tuned device primitives (babybear.cuh): dmul = 2 mad_u64 + mul_lo + subrev_co + cndmask
__global__ void __launch_bounds__(256) k_dmul(uint32_t* out, uint32_t seed) {
    uint32_t a[NACC];
    uint32_t b = (seed | 1u) % zk::P;
#pragma unroll
    for (int i = 0; i < NACC; i++) a[i] = (threadIdx.x * 7u + i + seed) % zk::P;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) a[i] = zk::dmul(a[i], b);
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_dadd(uint32_t* out, uint32_t seed) {
    uint32_t a[NACC];
    uint32_t b = (threadIdx.x * 13u + seed) % zk::P;
#pragma unroll
    for (int i = 0; i < NACC; i++) a[i] = (threadIdx.x * 7u + i + seed) % zk::P;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) a[i] = zk::dadd(a[i], b);
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_dbutterfly(uint32_t* out, uint32_t seed) {
    uint32_t a[NACC];
    uint32_t b = (threadIdx.x * 13u + seed) % zk::P;
#pragma unroll
    for (int i = 0; i < NACC; i++) a[i] = (threadIdx.x * 7u + i + seed) % zk::P;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i += 2) {
                uint32_t s = zk::dadd(a[i], a[i + 1]);
                uint32_t d = zk::dmul(zk::dsub_lazy(a[i], a[i + 1]), b);
                a[i] = s; a[i + 1] = d;
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

typedef void (*opk_t)(uint32_t*, uint32_t);

static void run_op(const char* name, opk_t k, double ops_per_thread_inst, int blocks_per_cu) {
    int blocks = 256 * blocks_per_cu;
    uint32_t* out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 12345u + rep);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double lane_ops = (double)blocks * 256 * ITER * UNROLL * NACC * ops_per_thread_inst;
    double tops = lane_ops / (best * 1e-3) / 1e12;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal: lanes/clk/SIMD = tops*1e12 / (1024 SIMDs * 2.4e9)
    double lanes_per_clk_simd = tops * 1e12 / (1024.0 * 2.4e9);
    printf("OP %-16s blocks/CU=%d  %.3f ms  %.2f T lane-ops/s  %.1f lanes/clk/SIMD@2.4GHz  (%.1f clk per wave64 instr)\n",
           name, blocks_per_cu, best, tops, lanes_per_clk_simd, 64.0 / lanes_per_clk_simd);
    CK(hipFree(out));
}

// ------------------------------------------------------------------ memory patterns
// copy a [rows][cols] u32 matrix in tiles of TR rows x CW words, where tile rows are
// `stride` rows apart (strided = first NTT pass) or adjacent (second pass); each lane
// moves one dword per access like the NTT kernel (CW words per row chunk).
__global__ void __launch_bounds__(1024) k_tile_copy(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                   uint64_t ld, uint32_t tiles_r, uint32_t ncg, uint32_t log_cw,
                                                   uint64_t tile_mul, uint64_t row_stride, int xcd_map) {
    const uint32_t CW = 1u << log_cw;
    const uint32_t c = threadIdx.x & (CW - 1), u = threadIdx.x >> log_cw;
    const uint32_t P_ = blockDim.x >> log_cw;
    uint32_t tile, cg;
    if (xcd_map) { uint32_t x = blockIdx.x & 7, l = blockIdx.x >> 3; cg = l % ncg; tile = (l / ncg) * 8 + x; }
    else { cg = blockIdx.x % ncg; tile = blockIdx.x / ncg; }
    const uint64_t base = ((uint64_t)tile * tile_mul + (uint64_t)u * row_stride) * ld + cg * CW + c;
    const uint64_t step = (uint64_t)P_ * row_stride * ld;
    uint32_t v[32];
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = in[base + i * step];
#pragma unroll
    for (int i = 0; i < 32; i++) out[base + i * step] = v[i] + 1;
}

// plain streaming copy, 16 B per lane
__global__ void __launch_bounds__(256) k_stream_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = in[i];
        v.x += 1;
        out[i] = v;
    }
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

// streaming variants: U x 16 B per lane in flight, optional nontemporal hints
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_stream_copy_u(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(&in[i + k * stride]) : in[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) {
            v[k].x += 1;
            if (NT) __builtin_nontemporal_store(v[k], &out[i + k * stride]); else out[i + k * stride] = v[k];
        }
    }
}
__global__ void __launch_bounds__(256) k_stream_read(const uint4* __restrict__ in, uint32_t* out, size_t n) {
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream_write(uint4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

static void run_mem(void) {
    const uint64_t rows = 1 << 20, cols = 256;
    size_t bytes = rows * cols * 4;
    uint32_t *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define TIME_IT(LABEL, BYTES, LAUNCH)                                                   \
    {                                                                                    \
        float best = 1e30f;                                                              \
        for (int rep = 0; rep < 4; rep++) {                                              \
            CK(hipEventRecord(e0)); LAUNCH; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
            float ms = time_ms(e0, e1); if (rep > 0 && ms < best) best = ms;             \
        }                                                                                \
        CK(hipGetLastError());                                                           \
        printf("MEM %-44s %.3f ms  %.2f TB/s\n", LABEL, best, (double)(BYTES) / best / 1e9); \
    }
    for (int bpc = 2; bpc <= 16; bpc *= 2) {
        char lab[96];
        snprintf(lab, sizeof lab, "stream_copy U=1 blocks/CU=%d", bpc);
        TIME_IT(lab, 2.0 * bytes, hipLaunchKernelGGL(k_stream_copy, dim3(256 * bpc), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16));
        snprintf(lab, sizeof lab, "stream_copy U=4 blocks/CU=%d", bpc);
        TIME_IT(lab, 2.0 * bytes, hipLaunchKernelGGL((k_stream_copy_u<4, false>), dim3(256 * bpc), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, bytes / 16));
        snprintf(lab, sizeof lab, "stream_copy U=4 nontemporal blocks/CU=%d", bpc);
        TIME_IT(lab, 2.0 * bytes, hipLaunchKernelGGL((k_stream_copy_u<4, true>), dim3(256 * bpc), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, bytes / 16));
    }
    TIME_IT("stream_read U=4 blocks/CU=8", 1.0 * bytes, hipLaunchKernelGGL(k_stream_read, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)a, b, bytes / 16));
    TIME_IT("stream_write blocks/CU=8", 1.0 * bytes, hipLaunchKernelGGL(k_stream_write, dim3(256 * 8), dim3(256), 0, 0, (uint4*)b, bytes / 16));
    // tile patterns: 32 rows per thread, P = threads / CW threads per column, chunk width CW words
    for (int strided = 1; strided >= 0; strided--) {
        for (int log_cw = 4; log_cw <= 6; log_cw++) {
            for (int xcd = 0; xcd <= 1; xcd++) {
                uint32_t CW = 1u << log_cw;
                uint32_t threads = 32 * CW > 1024 ? 1024 : 32 * CW;
                uint32_t Pn = threads / CW, trows = 32 * Pn;
                uint32_t ncg = cols / CW, tiles = rows / trows;
                uint64_t tile_mul = strided ? 1 : trows, row_stride = strided ? tiles : 1;
                char lab[96];
                snprintf(lab, sizeof lab, "tile_copy %s chunk=%3uB thr=%4u rows=%4u xcd=%d", strided ? "strided" : "contig ", CW * 4, threads, trows, xcd);
                TIME_IT(lab, 2.0 * bytes, hipLaunchKernelGGL(k_tile_copy, dim3(tiles * ncg), dim3(threads), 0, 0, a, b, cols, tiles, ncg, log_cw, tile_mul, row_stride, xcd));
            }
        }
    }
    CK(hipFree(a)); CK(hipFree(b));
}

int main(int argc, char** argv) {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
    printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int bpc = 8; bpc <= 8; bpc += 4) {
        run_op("v_add_u32", k_add_u32, 1, bpc);
        run_op("v_min_u32", k_min_u32, 1, bpc);
        run_op("v_add3_u32", k_add3_u32, 1, bpc);
        run_op("v_lshl_add_u32", k_lshl_add_u32, 1, bpc);
        run_op("v_sub_u32", k_sub_u32, 1, bpc);
        run_op("v_and_b32", k_and_b32, 1, bpc);
        run_op("v_xor_b32", k_xor_b32, 1, bpc);
        run_op("v_lshrrev_b32", k_lshrrev_b32, 1, bpc);
        run_op("v_ashrrev_i32", k_ashrrev_i32, 1, bpc);
        run_op("v_max_u32", k_max_u32, 1, bpc);
        run_op("v_mov_b32", k_mov_b32, 1, bpc);
        run_op("v_add_u32 literal", k_add_lit, 1, bpc);
        run_op("v_add_u32 sgpr", k_add_sgpr, 1, bpc);
        run_op("v_add_co_u32", k_add_co, 1, bpc);
        run_op("v_sub_co_u32", k_sub_co, 1, bpc);
        run_op("v_cndmask_b32", k_cndmask, 1, bpc);
        run_op("cmp+cndmask(x2)", k_cmp_cnd, 2, bpc);
        run_op("sub_co+cndmask(x2)", k_subco_cnd, 2, bpc);
        run_op("v_mul_hi_i32", k_mul_hi_i32, 1, bpc);
        run_op("v_mul_lo_u32 sgpr", k_mul_lo_sgpr, 1, bpc);
        run_op("v_alignbit_b32", k_alignbit, 1, bpc);
        run_op("v_bfe_u32", k_bfe_u32, 1, bpc);
        run_op("v_add_u32_dpp", k_sub_dpp, 1, bpc);
        run_op("v_pk_add_u16", k_pk_add_u16, 1, bpc);
        run_op("v_pk_mul_lo_u16", k_pk_mul_lo_u16, 1, bpc);
        run_op("v_dot4_u32_u8", k_dot4_u32_u8, 1, bpc);
        run_op("v_mul_lo_u32", k_mul_lo_u32, 1, bpc);
        run_op("v_mul_hi_u32", k_mul_hi_u32, 1, bpc);
        run_op("v_mul_u32_u24", k_mul_u32_u24, 1, bpc);
        run_op("v_mad_u32_u24", k_mad_u32_u24, 1, bpc);
        run_op("v_mad_u64_u32", k_mad_u64_u32, 1, bpc);
        run_op("v_fma_f32", k_fma_f32, 1, bpc);
        run_op("v_fma_f64", k_fma_f64, 1, bpc);
        run_op("monty fmul", k_fmul, 1, bpc);
        run_op("monty fadd", k_fadd, 1, bpc);
        run_op("butterfly(x0.5)", k_butterfly, 0.5, bpc);
        run_op("tuned dmul", k_dmul, 1, bpc);
        run_op("tuned dadd", k_dadd, 1, bpc);
        run_op("tuned butterfly(x0.5)", k_dbutterfly, 0.5, bpc);
    }
    run_mem();
    return 0;
}
