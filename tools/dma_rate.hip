// Micro-benchmark of the L2 -> LDS path (buffer_load_dwordx4 ... lds) by ACCESS PATTERN: what does a CU get when every 1-KiB piece is
//   16 rows x 64 B (the GEMM kernels' stage layout: 32 k of 16 activation / weight rows - half a 128-B line per row and step),
//    8 rows x 128 B (whole lines: what a 64-k stage would fetch), 4 x 256 B, 1 x 1024 B (contiguous)?
// One block per CU (8 waves), every wave keeps DEPTH pieces in flight into a wave-private LDS ring and issues the next one as soon as the
// oldest has landed (counted vmcnt) - nothing else runs, so the number is the path's own ceiling for that pattern.
//   hipcc -O3 --offload-arch=gfx950 tools/dma_rate.hip -o tools/bin/dma_rate ;  tools/bin/dma_rate            (prints one table)
// "rows" of a block: 192 (A-like, its own rows, streamed from HBM / Infinity Cache) or shared by all blocks (W-like, L2-resident after the first touch).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __amdgpu_buffer_rsrc_t bufrsrc_t;

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

constexpr int NW = 8;      // (wave slots of the LDS ring / row assignment; blocks of 4 waves use half of them)

// RPP = rows per piece (16 / 8 / 4 / 1); bytes per row and piece = 1024 / RPP.  A wave owns RG = 2 groups of RPP rows and walks their k
// (address arithmetic per piece: one scalar add - the first version divided per piece and measured its own VALU, 20.0 B/clk for everything)
constexpr int RG = 2;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int RPP, int DEPTH, bool REG, int WAVES = 8>
__global__ __launch_bounds__(64 * WAVES, 2) void dma_rate_kernel(const char* src, unsigned bytes, unsigned stride, int shared, int total, unsigned long long* out, int rot) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NW * DEPTH * 1024];
    static_assert(DEPTH % RG == 0, "");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bufrsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)bytes, 0x00020000);
    constexpr int BPR = 1024 / RPP;           // bytes per row in one piece
    constexpr int LPR = BPR / 16;             // lanes per row
    const int row = lane / LPR, chunk = lane % LPR;
    const unsigned row0 = shared ? 0u : (unsigned)blockIdx.x * (unsigned)(NW * RG * RPP);
    unsigned vo[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) vo[g] = (row0 + (unsigned)((wave * RG + g) * RPP + row)) * stride + (unsigned)chunk * 16u;
    int so = rot ? (int)(((blockIdx.x >> 3) * (unsigned)rot * BPR) % stride) : 0;     // rot: the CUs of an XCD start at different k
    u32x4 r[DEPTH];
    auto issue = [&](int slot) __attribute__((always_inline)) {
        if (REG)
            r[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo[slot % RG], so, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (wave * DEPTH + slot) * 1024), 16, (int)vo[slot % RG], so, 0, 0);
        if (slot % RG == RG - 1) {
            so += BPR;
            if (so >= (int)stride) so = 0;
        }
    };
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) issue(i);
    for (int it = DEPTH; it < total; it += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            if (REG) {
                *reinterpret_cast<u32x4*>(lds + (wave * DEPTH + i) * 1024 + lane * 16) = r[i];     // (the compiler's own counted wait on r[i])
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
            }
            issue(i);
        }
    }
    if (REG) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) *reinterpret_cast<u32x4*>(lds + (wave * DEPTH + i) * 1024 + lane * 16) = r[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (total < 0) out[0] = lds[threadIdx.x];
}

template <int RPP, int DEPTH, bool REG = false, int WAVES = 8>
void run(const char* what, const char* src, size_t bytes, unsigned stride, int shared, int laps, int ncu, unsigned long long* dout, int rot = 0, int grid = 0) {
    if (!grid) grid = ncu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    const int ksteps = laps * (int)(stride / (1024 / RPP));
    const int total = ksteps * RG;               // pieces per wave
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((dma_rate_kernel<RPP, DEPTH, REG, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, src, (unsigned)bytes, stride, shared, total, dout, rot);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double per_cu = (double)total * WAVES * 1024.0;     // bytes one block moved
    printf("%-30s %s %2d rows x %4d B  depth %2d  blocks %3d x %d waves  rows/block %3d  %8.1f us  %6.2f TB/s aggregate  %5.1f B/clk/CU (2.4 GHz)\n", what,
           REG ? "VGPR+ds_write" : "LDS-DMA      ", RPP, 1024 / RPP, DEPTH, grid, WAVES, NW * RG * RPP, best * 1e3, per_cu * grid / (best * 1e-3) / 1e12,
           per_cu * grid / ncu / (best * 1e-3 * 2.4e9));
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const unsigned stride = 5120;                       // K = 2560 bf16
    const size_t bytes = (size_t)ncu * NW * RG * 16 * stride;    // 256 rows per block at 16 rows per piece: 335 MB
    char* src;
    CK(hipMalloc(reinterpret_cast<void**>(&src), bytes));
    CK(hipMemset(src, 1, bytes));
    unsigned long long* dout;
    CK(hipMalloc(reinterpret_cast<void**>(&dout), 4096));
    printf("%d CUs; one 512-thread block per CU, each wave keeps `depth` 1-KiB pieces in flight; a block's rows are 5120 B long\n", ncu);
    printf("-- own rows per block, walked once (activation-like: streamed from HBM / the Infinity Cache)\n");
    run<16, 8>("own rows, 1 lap", src, bytes, stride, 0, 1, ncu, dout);
    run<8, 8>("own rows, 1 lap", src, bytes, stride, 0, 1, ncu, dout);
    run<4, 8>("own rows, 1 lap", src, bytes, stride, 0, 1, ncu, dout);
    run<1, 8>("own rows, 1 lap", src, bytes, stride, 0, 1, ncu, dout);
    run<16, 16>("own rows, 1 lap", src, bytes, stride, 0, 1, ncu, dout);
    run<8, 16>("own rows, 1 lap", src, bytes, stride, 0, 1, ncu, dout);
    run<16, 8, true>("own rows, 1 lap", src, bytes, stride, 0, 1, ncu, dout);
    run<8, 8, true>("own rows, 1 lap", src, bytes, stride, 0, 1, ncu, dout);
    printf("-- the same rows for every block, 16 laps (weight-like: L2-resident)\n");
    run<16, 8>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout);
    run<8, 8>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout);
    run<4, 8>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout);
    run<1, 8>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout);
    run<16, 16>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout);
    run<8, 16>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout);
    run<16, 8>("shared rows, rotated start", src, bytes, stride, 1, 16, ncu, dout, 5);
    run<8, 8>("shared rows, rotated start", src, bytes, stride, 1, 16, ncu, dout, 3);
    run<16, 8, true>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout);
    run<8, 8, true>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout);
    run<16, 4, true>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout);
    printf("-- fewer CUs active (per-CU or per-chip ceiling?)\n");
    run<16, 8>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout, 0, 128);
    run<16, 8>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout, 0, 64);
    run<16, 8>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout, 0, 8);
    run<16, 8, true>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout, 0, 8);
    printf("-- how the rate scales with the waves that issue (B/clk per CU; 512 blocks = two 8-wave blocks per CU)\n");
    run<16, 8, false, 4>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout);
    run<8, 8, false, 4>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout);
    run<16, 16, false, 4>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout);
    run<16, 8, false, 8>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout, 0, 2 * ncu);
    run<8, 8, false, 8>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout, 0, 2 * ncu);
    run<16, 8, false, 4>("shared rows, 16 laps", src, bytes, stride, 1, 16, ncu, dout, 0, 64);
    return 0;
}
