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

constexpr int NW = 8;

// RPP = rows per piece (16 / 8 / 4 / 1); bytes per row and piece = 1024 / RPP
template <int RPP, int DEPTH>
__global__ __launch_bounds__(512, 2) void dma_rate_kernel(const char* src, unsigned bytes, unsigned stride, int rows_per_block, int shared, int steps,
                                                         unsigned long long* out, int rot) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NW * DEPTH * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bufrsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)bytes, 0x00020000);
    constexpr int BPR = 1024 / RPP;           // bytes per row in one piece
    constexpr int LPR = BPR / 16;             // lanes per row
    const int row = lane / LPR, chunk = lane % LPR;
    const int pps = rows_per_block / RPP;     // pieces per "step" (one k slice of BPR bytes over all rows of the block)
    const unsigned row0 = shared ? 0u : (unsigned)blockIdx.x * (unsigned)rows_per_block;
    // piece q of the block (flat over steps): rows (q % pps) * RPP ..., k offset (q / pps) * BPR; wave w takes q = w, w + 8, ...
    int q = wave;
    const int k_rot = rot ? (int)((blockIdx.x >> 3) * rot) % (int)(stride / BPR) : 0;     // rot: blocks of one XCD start their walk at different k
    auto issue = [&](int slot) __attribute__((always_inline)) {
        const int pr = q % pps, pk = (q / pps + k_rot) % (int)(stride / BPR);     // (walks past the row end wrap: laps over the same bytes)
        const unsigned vo = (row0 + (unsigned)(pr * RPP + row)) * stride + (unsigned)chunk * 16u;
        const int so = __builtin_amdgcn_readfirstlane(pk * BPR);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (wave * DEPTH + slot) * 1024), 16, (int)vo, so, 0, 0);
        q += NW;
    };
    const int total = steps * pps / NW;        // pieces per wave
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) issue(i);
    int slot = 0;
    for (int i = DEPTH; i < total; ++i) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        issue(slot);
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * NW + wave] = t1 - t0;
    // keep the LDS contents observable
    if (steps < 0) out[0] = lds[threadIdx.x];
}


// the same walk through registers: buffer_load_dwordx4 -> VGPR -> ds_write_b128 (what the DMA path replaces)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int RPP, int DEPTH>
__global__ __launch_bounds__(512, 2) void reg_rate_kernel(const char* src, unsigned bytes, unsigned stride, int rows_per_block, int shared, int steps,
                                                         unsigned long long* out, int rot) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NW * DEPTH * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bufrsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)bytes, 0x00020000);
    constexpr int BPR = 1024 / RPP;
    constexpr int LPR = BPR / 16;
    const int row = lane / LPR, chunk = lane % LPR;
    const int pps = rows_per_block / RPP;
    const unsigned row0 = shared ? 0u : (unsigned)blockIdx.x * (unsigned)rows_per_block;
    int q = wave;
    const int k_rot = rot ? (int)((blockIdx.x >> 3) * rot) % (int)(stride / BPR) : 0;
    u32x4 r[DEPTH];
    auto issue = [&](int i) __attribute__((always_inline)) {
        const int pr = q % pps, pk = (q / pps + k_rot) % (int)(stride / BPR);
        const unsigned vo = (row0 + (unsigned)(pr * RPP + row)) * stride + (unsigned)chunk * 16u;
        const int so = __builtin_amdgcn_readfirstlane(pk * BPR);
        r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, so, 0);
        q += NW;
    };
    const int total = steps * pps / NW;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) issue(i);
    for (int it = DEPTH; it < total; it += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            *reinterpret_cast<u32x4*>(lds + (wave * DEPTH + i) * 1024 + lane * 16) = r[i];
            issue(i);
        }
    }
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) *reinterpret_cast<u32x4*>(lds + (wave * DEPTH + i) * 1024 + lane * 16) = r[i];
    __syncthreads();
    if (steps < 0) out[0] = lds[threadIdx.x];
}

template <int RPP, int DEPTH, bool REG = false>
void run(const char* what, const char* src, size_t bytes, unsigned stride, int rows_per_block, int shared, int steps, int ncu, unsigned long long* dout, int rot = 0, int grid = 0) {
    if (!grid) grid = ncu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    std::vector<unsigned long long> h(ncu * NW);
    double cyc = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        if (REG)
            hipLaunchKernelGGL((reg_rate_kernel<RPP, DEPTH>), dim3(grid), dim3(512), 0, 0, src, (unsigned)bytes, stride, rows_per_block, shared, steps, dout, rot);
        else
            hipLaunchKernelGGL((dma_rate_kernel<RPP, DEPTH>), dim3(grid), dim3(512), 0, 0, src, (unsigned)bytes, stride, rows_per_block, shared, steps, dout, rot);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            cyc = 0;
        }
    }
    const double per_cu = (double)steps * rows_per_block * (1024 / RPP) * 1.0;     // bytes one CU moved
    // s_memtime ticks at 100 MHz on this part; the rate in B/clk uses the event time and the 2.4 GHz shader clock
    printf("%-34s %s rows x bytes %2d x %4d  depth %2d  blocks %3d  %8.1f us  %6.2f TB/s aggregate  %5.1f B/clk/CU (2.4 GHz)\n", what, REG ? "VGPR+ds_write" : "LDS-DMA      ",
           RPP, 1024 / RPP, DEPTH, grid, best * 1e3, per_cu * grid / (best * 1e-3) / 1e12, per_cu / (best * 1e-3 * 2.4e9));
    (void)cyc;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const unsigned stride = 5120;                       // K = 2560 bf16
    const int rpb = 768;
    const size_t bytes = (size_t)ncu * rpb * stride;    // 1 GB for 256 CUs
    char* src;
    CK(hipMalloc(reinterpret_cast<void**>(&src), bytes));
    CK(hipMemset(src, 1, bytes));
    unsigned long long* dout;
    CK(hipMalloc(reinterpret_cast<void**>(&dout), (size_t)ncu * NW * 8));
    const int steps64 = stride / 64;                    // a block walks all k of its rows
    printf("device: %s, %d CUs; one 512-thread block per CU, each wave keeps `depth` 1-KiB pieces in flight\n", prop.name, ncu);
    printf("-- own 768 rows per block (activation-like: 1 GB streamed from HBM)\n");
    run<16, 8>("own 768 rows", src, bytes, stride, rpb, 0, steps64, ncu, dout);
    run<8, 8>("own 768 rows", src, bytes, stride, rpb, 0, steps64 / 2, ncu, dout);
    run<4, 8>("own 768 rows", src, bytes, stride, rpb, 0, steps64 / 4, ncu, dout);
    run<1, 8>("own 768 rows", src, bytes, stride, rpb, 0, steps64 / 16, ncu, dout);
    run<16, 16>("own 768 rows", src, bytes, stride, rpb, 0, steps64, ncu, dout);
    run<8, 16>("own 768 rows", src, bytes, stride, rpb, 0, steps64 / 2, ncu, dout);
    run<1, 16>("own 768 rows", src, bytes, stride, rpb, 0, steps64 / 16, ncu, dout);
    printf("-- own 192 rows per block (252 MB: fits the Infinity Cache, second and later repetitions come from it)\n");
    run<16, 8>("own 192 rows", src, bytes, stride, 192, 0, steps64, ncu, dout);
    run<8, 8>("own 192 rows", src, bytes, stride, 192, 0, steps64 / 2, ncu, dout);
    run<16, 16>("own 192 rows", src, bytes, stride, 192, 0, steps64, ncu, dout);
    run<8, 16>("own 192 rows", src, bytes, stride, 192, 0, steps64 / 2, ncu, dout);
    printf("-- the same 320 rows (1.6 MB) for every block (weight-like: L2-resident), walked 4 times\n");
    run<16, 8>("shared 320 rows x 4 laps", src, bytes, stride, 320, 1, 4 * steps64, ncu, dout);
    run<8, 8>("shared 320 rows x 4 laps", src, bytes, stride, 320, 1, 4 * steps64 / 2, ncu, dout);
    run<4, 8>("shared 320 rows x 4 laps", src, bytes, stride, 320, 1, 4 * steps64 / 4, ncu, dout);
    run<1, 8>("shared 320 rows x 4 laps", src, bytes, stride, 320, 1, 4 * steps64 / 16, ncu, dout);
    run<16, 16>("shared 320 rows x 4 laps", src, bytes, stride, 320, 1, 4 * steps64, ncu, dout);
    run<8, 16>("shared 320 rows x 4 laps", src, bytes, stride, 320, 1, 4 * steps64 / 2, ncu, dout);
    printf("-- the same 320 rows, but block b starts its walk (b / 8) * rot k-steps in (wraps): no two CUs of an XCD ask for the same line at the same time\n");
    run<16, 8>("shared 320 rows, rotated start", src, bytes, stride, 320, 1, 4 * steps64, ncu, dout, 5);
    run<8, 8>("shared 320 rows, rotated start", src, bytes, stride, 320, 1, 4 * steps64 / 2, ncu, dout, 3);
    run<1, 8>("shared 320 rows, rotated start", src, bytes, stride, 320, 1, 4 * steps64 / 16, ncu, dout, 1);
    printf("-- 16 own rows per block (80 KB each, 2.6 MB per XCD: L2-resident, nothing shared), 48 laps\n");
    run<16, 8>("own 16 rows x 48 laps", src, bytes, stride, 16, 0, 48 * steps64, ncu, dout);
    run<8, 8>("own 16 rows x 48 laps", src, bytes, stride, 16, 0, 48 * steps64 / 2, ncu, dout);
    run<1, 8>("own 16 rows x 48 laps", src, bytes, stride, 16, 0, 48 * steps64 / 16, ncu, dout);
    run<16, 16>("own 16 rows x 48 laps", src, bytes, stride, 16, 0, 48 * steps64, ncu, dout);
    printf("-- through registers instead (buffer_load_dwordx4 -> VGPR -> ds_write_b128)\n");
    run<16, 8, true>("own 16 rows x 48 laps", src, bytes, stride, 16, 0, 48 * steps64, ncu, dout);
    run<8, 8, true>("own 16 rows x 48 laps", src, bytes, stride, 16, 0, 48 * steps64 / 2, ncu, dout);
    run<1, 8, true>("own 16 rows x 48 laps", src, bytes, stride, 16, 0, 48 * steps64 / 16, ncu, dout);
    run<16, 8, true>("shared 320 rows x 4 laps", src, bytes, stride, 320, 1, 4 * steps64, ncu, dout);
    run<8, 8, true>("shared 320 rows x 4 laps", src, bytes, stride, 320, 1, 4 * steps64 / 2, ncu, dout);
    run<16, 8, true>("own 768 rows", src, bytes, stride, rpb, 0, steps64, ncu, dout);
    run<8, 8, true>("own 768 rows", src, bytes, stride, rpb, 0, steps64 / 2, ncu, dout);
    printf("-- fewer CUs active (is the ceiling per CU or per chip?)\n");
    run<16, 8>("own 16 rows x 48 laps", src, bytes, stride, 16, 0, 48 * steps64, ncu, dout, 0, 128);
    run<16, 8>("own 16 rows x 48 laps", src, bytes, stride, 16, 0, 48 * steps64, ncu, dout, 0, 64);
    run<16, 8>("own 16 rows x 48 laps", src, bytes, stride, 16, 0, 48 * steps64, ncu, dout, 0, 8);
    run<16, 8, true>("own 16 rows x 48 laps", src, bytes, stride, 16, 0, 48 * steps64, ncu, dout, 0, 64);
    run<16, 8, true>("own 16 rows x 48 laps", src, bytes, stride, 16, 0, 48 * steps64, ncu, dout, 0, 8);
    return 0;
}
