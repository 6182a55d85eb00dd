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
__global__ void __launch_bounds__(512) k_tile_copy(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
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

static void run_mem(void) {
    const uint64_t rows = 1 << 20, cols = 256;
    size_t bytes = rows * cols * 4;
    uint32_t *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // stream copy
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_stream_copy, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        if (rep == 2) printf("MEM stream_copy 1GiB+1GiB: %.3f ms  %.2f TB/s\n", time_ms(e0, e1), 2.0 * bytes / time_ms(e0, e1) / 1e9);
    }
    // tile patterns: 1024-row tiles (32 per thread x P=32 threads) with chunk width CW words
    for (int strided = 1; strided >= 0; strided--) {
        for (int log_cw = 4; log_cw <= 6; log_cw++) {
            for (int xcd = 0; xcd <= 1; xcd++) {
                uint32_t CW = 1u << log_cw;
                uint32_t threads = 32 * CW;   // P = 32
                if (threads > 1024) {
                    continue;
                }
                uint32_t ncg = cols / CW, tiles = rows / 1024;
                uint64_t tile_mul = strided ? 1 : 1024, row_stride = strided ? 1024 : 1;
                float best = 1e30f;
                for (int rep = 0; rep < 4; rep++) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_tile_copy, dim3(tiles * ncg), dim3(threads), 0, 0, a, b, cols, tiles, ncg, log_cw, tile_mul, row_stride, xcd);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms = time_ms(e0, e1);
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("MEM tile_copy %s chunk=%3uB threads=%4u xcd_map=%d: %.3f ms  %.2f TB/s\n",
                       strided ? "strided   " : "contiguous", CW * 4, threads, xcd, best, 2.0 * bytes / best / 1e9);
            }
        }
    }
    CK(hipGetLastError());
    CK(hipFree(a)); CK(hipFree(b));
}

int main(int argc, char** argv) {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
    printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int bpc = 4; bpc <= 8; bpc += 4) {
        run_op("v_add_u32", k_add_u32, 1, bpc);
        run_op("v_min_u32", k_min_u32, 1, bpc);
        run_op("v_add3_u32", k_add3_u32, 1, bpc);
        run_op("v_lshl_add_u32", k_lshl_add_u32, 1, bpc);
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
    }
    run_mem();
    return 0;
}
