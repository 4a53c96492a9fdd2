/* v3d_comm.h - C ABI of libv3d_comm.so: the three frame-axis exchanges of a frame-sharded V3D evaluation over RCCL, for hosts that do not
 * bring torch.distributed (SURVEY.md 8b last row: v3d_comm_init / v3d_allgather_frames / v3d_halo_exchange / v3d_gn_stats_allreduce).
 *
 * The reference has NO distributed code on this path: these entries mirror the new design of v3d_amd/dist.py::FrameShard, which
 * is what the Python host uses (torch.distributed P2P ops on the same buffers).  One process per GPU; contiguous frame ranges per rank
 * (18 frames over 8 ranks = 3,3,2,2,2,2,2,2); weights replicated; every exchange is ONE grouped point-to-point launch (ncclGroupStart / End
 * around exact-size ncclSend / ncclRecv pairs: xGMI is a point-to-point mesh, no ring collective on the evaluation path).
 *   (i)   temporal self-attention  (video_attention.py:114,122-125)   K|V of every rank's frames to every rank        v3d_comm_allgather_frames
 *   (ii)  (3,1,1) temporal conv    (video_model.py:42-55)              +-1 frame halo with the ring neighbours         } v3d_comm_exchange_halo_and_sums
 *   (iii) 3-D GroupNorm statistics (openaimodel.py:267-271,302-305)    every rank's fp64 (sum, sumsq) to every rank    } (one grouped call)
 * STATUS: verified on hardware with ONE rank only (communicator creation, grouped self send / recv, the copy / reduction kernels) - the
 * build boxes have one GPU and RCCL refuses two ranks per device; the multi-rank message schedule is the one dist.py runs under gloo with
 * 2 / 4 / 8 ranks in tests/test_dist_gloo.py; tests/test_comm_schedule.py holds this library's own send / recv lists (v3d_comm_debug_schedule) to
 * dist.py's for world 2 / 4 / 8 and checks that every pair of ranks agrees on sizes and order.  No RCCL timing exists for this path.
 * All pointers are device pointers on the current device; return 0 or a negative code, message through v3d_comm_last_error(). */
#ifndef V3D_COMM_H
#define V3D_COMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define V3D_COMM_ABI_VERSION 1
#define V3D_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct v3d_comm_s* v3d_comm_t;
typedef void* v3d_comm_stream_t; /* hipStream_t */

int v3d_comm_abi_version(void);
const char* v3d_comm_last_error(void);
/* rank 0 creates the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by whatever channel the host has */
int v3d_comm_unique_id(void* id_out);
/* ncclCommInitRank on the CURRENT device; collective over all `world` ranks */
int v3d_comm_init(const void* unique_id, int32_t rank, int32_t world, v3d_comm_t* comm_out);
int v3d_comm_destroy(v3d_comm_t comm);
/* contiguous, as-even-as-possible frame ranges: rank r of `world` owns frames [t0, t0 + t_local) of T_global (dist.py frame_partition) */
int v3d_comm_frame_range(int32_t T_global, int32_t world, int32_t rank, int32_t* t0, int32_t* t_local);
/* (i) local [B][T_local][frame_bytes] -> out [B][T_global][frame_bytes] on every rank: rank r's frames land directly in frames parts[r] of every
 * sample (uneven shards move exactly their own bytes), the own frames are copied on `stream`.  frame_bytes = S * C * 2 for bf16 K|V rows. */
int v3d_comm_allgather_frames(v3d_comm_t comm, const void* local, void* out, int64_t B, int32_t T_global, int64_t frame_bytes,
                              v3d_comm_stream_t stream);
/* (ii) + (iii) in ONE grouped call.  buf = split-halo activation buffer of v3d_gemm's CONVT3 halo_rows layout,
 *   [B frames: the previous rank's last frames | B * T_local local frames | B frames: the next rank's first frames] of frame_bytes each;
 * this rank's first / last local frame of every sample goes to the previous / next rank and theirs arrive in the outer slabs (nothing is sent
 * or received at the global ends).  buf == NULL: statistics only.  sums (nsums doubles: [n_stat][groups][2] of v3d_groupnorm_finalize) goes to
 * every other rank, theirs arrive in allsums[world][nsums]; total[nsums] = the ranks' tables added in RANK ORDER (same bits on every rank).
 * sums == NULL: halos only. */
int v3d_comm_exchange_halo_and_sums(v3d_comm_t comm, void* buf, int64_t B, int32_t T_global, int64_t frame_bytes, const double* sums,
                                    double* allsums, double* total, int64_t nsums, v3d_comm_stream_t stream);
/* one grouped ncclSend + ncclRecv of `bytes` bytes from this rank to itself: exercises the grouped point-to-point path on a single GPU */
int v3d_comm_selftest(v3d_comm_t comm, const void* src, void* dst, int64_t bytes, v3d_comm_stream_t stream);

/* tests / diagnostics (no GPU, no communicator): the (send, peer, buffer, byte offset, bytes) list rank `rank` of `world` would issue inside its ONE group -
 * kind 0: v3d_comm_allgather_frames (buffer 0 = local, 1 = out); kind 1: v3d_comm_exchange_halo_and_sums (have_buf: halos; nsums > 0: statistics;
 * buffer 0 = buf, 1 = sums, 2 = allsums).  out holds 5 int64 per message; returns the message count or -1.  The exchanges issue exactly this list. */
int v3d_comm_debug_schedule(int32_t kind, int32_t rank, int32_t world, int32_t have_buf, int64_t B, int32_t T_global, int64_t frame_bytes, int64_t nsums,
                            int64_t* out, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif
