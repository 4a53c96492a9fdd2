// libv3d_comm.so: the frame-axis exchanges of a frame-sharded evaluation over RCCL behind a C ABI (include/v3d_comm.h).  Mirrors
// v3d_amd/dist.py::FrameShard (allgather_frames, exchange_halo_and_sums); the reference has no distributed code on this path.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/v3d_comm.h"

namespace {
thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#define COMM_REQUIRE(cond, ...)      \
    do {                             \
        if (!(cond)) {               \
            set_error(__VA_ARGS__);  \
            return -1;               \
        }                            \
    } while (0)
#define NCCL_TRY(call)                                                               \
    do {                                                                             \
        ncclResult_t r_ = (call);                                                    \
        if (r_ != ncclSuccess) {                                                     \
            set_error("%s: %s", #call, ncclGetErrorString(r_));                      \
            return -3;                                                               \
        }                                                                            \
    } while (0)
#define HIP_TRY(call)                                                                \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            set_error("%s: %s", #call, hipGetErrorString(e_));                       \
            return -2;                                                               \
        }                                                                            \
    } while (0)

void frame_range(int T, int world, int rank, int& t0, int& tl) {
    const int base = T / world, rem = T % world;
    tl = base + (rank < rem ? 1 : 0);
    t0 = rank * base + (rank < rem ? rank : rem);
}

// total[i] = allsums[0][i] + allsums[1][i] + ... in rank order (fp64): the same bits on every rank
__global__ void sum_rank_order_kernel(const double* __restrict__ allsums, double* __restrict__ total, int world, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = allsums[i];
    for (int r = 1; r < world; ++r) a += allsums[(long long)r * n + i];
    total[i] = a;
}
}  // namespace

struct v3d_comm_s {
    ncclComm_t comm;
    int rank, world;
};

extern "C" int v3d_comm_abi_version(void) { return V3D_COMM_ABI_VERSION; }
extern "C" const char* v3d_comm_last_error(void) { return g_err; }

extern "C" int v3d_comm_unique_id(void* id_out) {
    COMM_REQUIRE(id_out != nullptr, "v3d_comm_unique_id: null output");
    static_assert(sizeof(ncclUniqueId) == V3D_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int v3d_comm_init(const void* unique_id, int32_t rank, int32_t world, v3d_comm_t* comm_out) {
    COMM_REQUIRE(unique_id && comm_out, "v3d_comm_init: null pointer");
    COMM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "v3d_comm_init: bad rank %d of %d", rank, world);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c;
    NCCL_TRY(ncclCommInitRank(&c, world, id, rank));
    *comm_out = new v3d_comm_s{c, rank, world};
    return 0;
}

extern "C" int v3d_comm_destroy(v3d_comm_t comm) {
    COMM_REQUIRE(comm != nullptr, "v3d_comm_destroy: null communicator");
    NCCL_TRY(ncclCommDestroy(comm->comm));
    delete comm;
    return 0;
}

extern "C" int v3d_comm_frame_range(int32_t T_global, int32_t world, int32_t rank, int32_t* t0, int32_t* t_local) {
    COMM_REQUIRE(t0 && t_local, "v3d_comm_frame_range: null output");
    COMM_REQUIRE(world >= 1 && T_global >= world && rank >= 0 && rank < world, "v3d_comm_frame_range: cannot shard %d frames over %d ranks (rank %d)",
                 T_global, world, rank);
    int a, b;
    frame_range(T_global, world, rank, a, b);
    *t0 = a;
    *t_local = b;
    return 0;
}

extern "C" int v3d_comm_allgather_frames(v3d_comm_t comm, const void* local, void* out, int64_t B, int32_t T_global, int64_t frame_bytes,
                                         v3d_comm_stream_t stream) {
    COMM_REQUIRE(comm && local && out, "v3d_comm_allgather_frames: null pointer");
    COMM_REQUIRE(B > 0 && frame_bytes > 0 && T_global >= comm->world, "v3d_comm_allgather_frames: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    int t0, tl;
    frame_range(T_global, comm->world, comm->rank, t0, tl);
    const char* src = (const char*)local;
    char* dst = (char*)out;
    // own frames: [B][T_local] rows of `local` into frames t0 .. of every sample
    HIP_TRY(hipMemcpy2DAsync(dst + (size_t)t0 * frame_bytes, (size_t)T_global * frame_bytes, src, (size_t)tl * frame_bytes, (size_t)tl * frame_bytes, (size_t)B,
                             hipMemcpyDeviceToDevice, st));
    if (comm->world == 1) return 0;
    NCCL_TRY(ncclGroupStart());
    for (int r = 0; r < comm->world; ++r) {
        if (r == comm->rank) continue;
        int rt0, rtl;
        frame_range(T_global, comm->world, r, rt0, rtl);
        for (int64_t b = 0; b < B; ++b) {      // messages between one pair of ranks match in issue order on both sides
            NCCL_TRY(ncclSend(src + (size_t)b * tl * frame_bytes, (size_t)tl * frame_bytes, ncclChar, r, comm->comm, st));
            NCCL_TRY(ncclRecv(dst + ((size_t)b * T_global + rt0) * frame_bytes, (size_t)rtl * frame_bytes, ncclChar, r, comm->comm, st));
        }
    }
    NCCL_TRY(ncclGroupEnd());
    return 0;
}

extern "C" int v3d_comm_exchange_halo_and_sums(v3d_comm_t comm, void* buf, int64_t B, int32_t T_global, int64_t frame_bytes, const double* sums,
                                               double* allsums, double* total, int64_t nsums, v3d_comm_stream_t stream) {
    COMM_REQUIRE(comm != nullptr, "v3d_comm_exchange_halo_and_sums: null communicator");
    COMM_REQUIRE(buf || sums, "v3d_comm_exchange_halo_and_sums: nothing to exchange");
    COMM_REQUIRE(!buf || (B > 0 && frame_bytes > 0 && T_global >= comm->world), "v3d_comm_exchange_halo_and_sums: bad halo geometry");
    COMM_REQUIRE(!sums || (allsums && total && nsums > 0), "v3d_comm_exchange_halo_and_sums: sums need allsums[world][nsums], total[nsums], nsums > 0");
    hipStream_t st = (hipStream_t)stream;
    const int rank = comm->rank, world = comm->world;
    if (sums) HIP_TRY(hipMemcpyAsync(allsums + (size_t)rank * nsums, sums, (size_t)nsums * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (world > 1) {
        int t0, tl;
        frame_range(T_global > 0 ? T_global : world, world, rank, t0, tl);
        char* base = (char*)buf;
        char* mid = base + (size_t)B * frame_bytes;                                     // [B][T_local] local frames
        char* hi = mid + (size_t)B * tl * frame_bytes;                                  // [B] frames of the next rank
        NCCL_TRY(ncclGroupStart());
        if (buf && rank > 0)
            for (int64_t b = 0; b < B; ++b) {                                           // my first frames -> previous rank; its last frames -> my frame -1 slab
                NCCL_TRY(ncclSend(mid + (size_t)b * tl * frame_bytes, (size_t)frame_bytes, ncclChar, rank - 1, comm->comm, st));
                NCCL_TRY(ncclRecv(base + (size_t)b * frame_bytes, (size_t)frame_bytes, ncclChar, rank - 1, comm->comm, st));
            }
        if (buf && rank + 1 < world)
            for (int64_t b = 0; b < B; ++b) {                                           // my last frames -> next rank; its first frames -> my frame T_local slab
                NCCL_TRY(ncclSend(mid + ((size_t)b * tl + (tl - 1)) * frame_bytes, (size_t)frame_bytes, ncclChar, rank + 1, comm->comm, st));
                NCCL_TRY(ncclRecv(hi + (size_t)b * frame_bytes, (size_t)frame_bytes, ncclChar, rank + 1, comm->comm, st));
            }
        if (sums)
            for (int r = 0; r < world; ++r) {
                if (r == rank) continue;
                NCCL_TRY(ncclSend(sums, (size_t)nsums, ncclDouble, r, comm->comm, st));
                NCCL_TRY(ncclRecv(allsums + (size_t)r * nsums, (size_t)nsums, ncclDouble, r, comm->comm, st));
            }
        NCCL_TRY(ncclGroupEnd());
    }
    if (sums) {
        hipLaunchKernelGGL(sum_rank_order_kernel, dim3((unsigned)((nsums + 255) / 256)), dim3(256), 0, st, (const double*)allsums, total, world, (long long)nsums);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" int v3d_comm_selftest(v3d_comm_t comm, const void* src, void* dst, int64_t bytes, v3d_comm_stream_t stream) {
    COMM_REQUIRE(comm && src && dst && bytes > 0, "v3d_comm_selftest: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    NCCL_TRY(ncclGroupStart());
    NCCL_TRY(ncclSend(src, (size_t)bytes, ncclChar, comm->rank, comm->comm, st));
    NCCL_TRY(ncclRecv(dst, (size_t)bytes, ncclChar, comm->rank, comm->comm, st));
    NCCL_TRY(ncclGroupEnd());
    return 0;
}
