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

// ---- the message schedule of one grouped exchange, as plain data --------------------------------------------------------------------------------
// Both exchanges build their send / recv list with these host functions and then issue it; v3d_comm_debug_schedule hands the same list to the CPU
// tests (tests/test_comm_schedule.py), which check for world 2 / 4 / 8 that every pair of ranks agrees on (sizes, order) and that the list is
// dist.py::FrameShard's.  Offsets are byte offsets into the buffer the message reads / writes (local / out / buf / sums / allsums).
struct Msg {
    int send;              // 1 = ncclSend, 0 = ncclRecv
    int peer;
    int buffer;            // allgather: 0 = local, 1 = out; halo + sums: 0 = buf, 1 = sums, 2 = allsums
    long long offset, bytes;
};
int build_allgather(int rank, int world, long long B, int T_global, long long frame_bytes, Msg* out, int cap) {
    int n = 0, t0, tl;
    frame_range(T_global, world, rank, t0, tl);
    for (int r = 0; r < world; ++r) {
        if (r == rank) continue;
        int rt0, rtl;
        frame_range(T_global, world, r, rt0, rtl);
        for (long long b = 0; b < B; ++b) {      // messages between one pair of ranks match in issue order on both sides
            if (n + 2 > cap) return -1;
            out[n++] = Msg{1, r, 0, b * tl * frame_bytes, tl * frame_bytes};
            out[n++] = Msg{0, r, 1, (b * T_global + rt0) * frame_bytes, rtl * frame_bytes};
        }
    }
    return n;
}
int build_halo_and_sums(int rank, int world, int have_buf, long long B, int T_global, long long frame_bytes, long long nsums, Msg* out, int cap) {
    int n = 0, t0, tl;
    frame_range(T_global > 0 ? T_global : world, world, rank, t0, tl);
    const long long mid = B * frame_bytes, hi = mid + B * tl * frame_bytes;      // [B] previous rank's frames | [B][T_local] local | [B] next rank's frames
    if (have_buf && rank > 0)
        for (long long b = 0; b < B; ++b) {                                       // my first frames -> previous rank; its last frames -> my frame -1 slab
            if (n + 2 > cap) return -1;
            out[n++] = Msg{1, rank - 1, 0, mid + b * tl * frame_bytes, frame_bytes};
            out[n++] = Msg{0, rank - 1, 0, b * frame_bytes, frame_bytes};
        }
    if (have_buf && rank + 1 < world)
        for (long long b = 0; b < B; ++b) {                                       // my last frames -> next rank; its first frames -> my frame T_local slab
            if (n + 2 > cap) return -1;
            out[n++] = Msg{1, rank + 1, 0, mid + (b * tl + (tl - 1)) * frame_bytes, frame_bytes};
            out[n++] = Msg{0, rank + 1, 0, hi + b * frame_bytes, frame_bytes};
        }
    if (nsums > 0)
        for (int r = 0; r < world; ++r) {
            if (r == rank) continue;
            if (n + 2 > cap) return -1;
            out[n++] = Msg{1, r, 1, 0, nsums * 8};
            out[n++] = Msg{0, r, 2, (long long)r * nsums * 8, nsums * 8};
        }
    return n;
}
// issue a list inside ONE group.  A failing send / recv still closes the group (an open group would queue - or hang - every later RCCL call of this thread).
int issue_group(const Msg* m, int n, char* const* bases, ncclComm_t comm, hipStream_t st, const char* what) {
    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess) { set_error("%s: ncclGroupStart: %s", what, ncclGetErrorString(r)); return -3; }
    ncclResult_t bad = ncclSuccess;
    for (int i = 0; i < n && bad == ncclSuccess; ++i) {
        char* ptr = bases[m[i].buffer] + m[i].offset;
        bad = m[i].send ? ncclSend(ptr, (size_t)m[i].bytes, ncclChar, m[i].peer, comm, st) : ncclRecv(ptr, (size_t)m[i].bytes, ncclChar, m[i].peer, comm, st);
    }
    r = ncclGroupEnd();
    if (bad != ncclSuccess) { set_error("%s: ncclSend / ncclRecv: %s", what, ncclGetErrorString(bad)); return -3; }
    if (r != ncclSuccess) { set_error("%s: ncclGroupEnd: %s", what, ncclGetErrorString(r)); return -3; }
    return 0;
}
constexpr int kMaxMsgs = 4096;

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
    static thread_local Msg msgs[kMaxMsgs];
    const int n = build_allgather(comm->rank, comm->world, B, T_global, frame_bytes, msgs, kMaxMsgs);
    COMM_REQUIRE(n >= 0, "v3d_comm_allgather_frames: more than %d messages (B too large)", kMaxMsgs);
    char* bases[2] = {const_cast<char*>(src), dst};
    return issue_group(msgs, n, bases, comm->comm, st, "v3d_comm_allgather_frames");
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
        static thread_local Msg msgs[kMaxMsgs];
        const int n = build_halo_and_sums(rank, world, buf != nullptr, B, T_global, frame_bytes, sums ? nsums : 0, msgs, kMaxMsgs);
        COMM_REQUIRE(n >= 0, "v3d_comm_exchange_halo_and_sums: more than %d messages (B too large)", kMaxMsgs);
        char* bases[3] = {(char*)buf, (char*)const_cast<double*>(sums), (char*)allsums};
        const int rc = issue_group(msgs, n, bases, comm->comm, st, "v3d_comm_exchange_halo_and_sums");
        if (rc != 0) return rc;
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
    Msg m[2] = {Msg{1, comm->rank, 0, 0, bytes}, Msg{0, comm->rank, 1, 0, bytes}};
    char* bases[2] = {(char*)const_cast<void*>(src), (char*)dst};
    return issue_group(m, 2, bases, comm->comm, st, "v3d_comm_selftest");
}

// tests / diagnostics (no GPU, no communicator): the message list rank `rank` of `world` would issue.  kind 0 = v3d_comm_allgather_frames,
// 1 = v3d_comm_exchange_halo_and_sums (have_buf: halos, nsums > 0: statistics).  out[5 * i ..] = (send, peer, buffer, offset, bytes); returns the
// message count, or -1 when `cap` messages are not enough.
extern "C" int v3d_comm_debug_schedule(int32_t kind, int32_t rank, int32_t world, int32_t have_buf, int64_t B, int32_t T_global, int64_t frame_bytes, int64_t nsums,
                                       int64_t* out, int32_t cap) {
    COMM_REQUIRE(out && cap > 0 && world >= 1 && rank >= 0 && rank < world && T_global >= world, "v3d_comm_debug_schedule: bad arguments");
    static thread_local Msg msgs[kMaxMsgs];
    const int lim = cap < kMaxMsgs ? cap : kMaxMsgs;
    const int n = kind == 0 ? build_allgather(rank, world, B, T_global, frame_bytes, msgs, lim) : build_halo_and_sums(rank, world, have_buf, B, T_global, frame_bytes, nsums, msgs, lim);
    if (n < 0) { set_error("v3d_comm_debug_schedule: more than %d messages", lim); return -1; }
    for (int i = 0; i < n; ++i) {
        out[5 * i] = msgs[i].send; out[5 * i + 1] = msgs[i].peer; out[5 * i + 2] = msgs[i].buffer; out[5 * i + 3] = msgs[i].offset; out[5 * i + 4] = msgs[i].bytes;
    }
    return n;
}
