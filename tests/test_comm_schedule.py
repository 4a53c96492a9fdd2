"""The two implementations of the frame-axis message schedule - libv3d_comm.so (csrc_comm/comm.hip, what a non-Python host runs over RCCL) and
v3d_amd/dist.py::FrameShard (what the Python host runs over torch.distributed) - must not drift.  No GPU, no
communicator: the library exports the list it would issue (v3d_comm_debug_schedule: the exchanges issue exactly that list), FrameShard runs with a
recording transport.  Checked for world 2 / 4 / 8 and B = 1 / 2:
  * pairwise matching inside each implementation: for every ordered pair (a, b) the sizes a sends to b, in issue order, are the sizes b posts
    receives for from a, in issue order (the property a grouped ncclSend / ncclRecv exchange needs to complete);
  * the two implementations agree: per rank the sequence of (direction, peer, bytes) is the same once consecutive messages of one direction to one
    peer are coalesced (dist.py packs a neighbour's B halo frames into ONE message, the C library sends them as B messages from the strided buffer).
The real multi-rank RCCL execution of either path only happens on the driver's 8-GPU node; nothing here claims a hardware number."""
import pytest
import torch

from test_comm_abi import _lib
from v3d_amd import comm
from v3d_amd.dist import SimFrameShard, _Handle

T = 18
S, C = 6, 8                     # rows per frame, channels (bf16): frame_bytes = S * C * 2
FRAME_BYTES = S * C * 2
NSUMS = 2 * 32 * 2              # [n_stat][groups][2] doubles


class RecordingShard(SimFrameShard):
    def __init__(self, T_global, world, rank):
        super().__init__(T_global, world, rank)
        self.log = []

    def _exchange(self, sends, recvs, async_op=False):
        # FrameShard._exchange posts every send, then every receive, of ONE grouped call
        self.log.append([("send", r, t.numel() * t.element_size()) for t, r in sends] + [("recv", r, t.numel() * t.element_size()) for t, r in recvs])
        return _Handle(())


def dist_schedules(world, B):
    """per rank: [allgather list, halo + sums list, sums-only list] as (direction, peer, bytes)"""
    out = []
    for rank in range(world):
        sh = RecordingShard(T, world, rank)
        x = torch.zeros(B, sh.T_local, S, C, dtype=torch.bfloat16)
        sh.allgather_frames(x)
        buf, _ = sh.halo_buffer(B, S, C, torch.bfloat16, "cpu")
        sums = torch.zeros(NSUMS, dtype=torch.float64)
        sh.exchange_halo_and_sums(buf, B, S, sums)
        sh.exchange_halo_and_sums(None, 0, 0, sums)
        assert len(sh.log) == 3
        out.append(sh.log)
    return out


def lib_schedules(lib, world, B):
    def conv(msgs):
        return [("send" if m[0] else "recv", m[1], m[4]) for m in msgs]
    out = []
    for rank in range(world):
        out.append([conv(comm.debug_schedule(lib, 0, rank, world, B=B, T_global=T, frame_bytes=FRAME_BYTES)),
                    conv(comm.debug_schedule(lib, 1, rank, world, have_buf=True, B=B, T_global=T, frame_bytes=FRAME_BYTES, nsums=NSUMS)),
                    conv(comm.debug_schedule(lib, 1, rank, world, have_buf=False, B=B, T_global=T, frame_bytes=FRAME_BYTES, nsums=NSUMS))])
    return out


def assert_pairs_match(per_rank, what):
    world = len(per_rank)
    for a in range(world):
        for b in range(world):
            if a == b:
                continue
            sent = [n for d, p, n in per_rank[a] if d == "send" and p == b]
            want = [n for d, p, n in per_rank[b] if d == "recv" and p == a]
            assert sent == want, (what, a, b, sent, want)


def coalesce(msgs):
    """bytes per (direction, peer), and the order in which the peers of each direction first appear"""
    tot, order = {}, {"send": [], "recv": []}
    for d, p, n in msgs:
        tot[(d, p)] = tot.get((d, p), 0) + n
        if p not in order[d]:
            order[d].append(p)
    return tot, order


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("B", [1, 2])
def test_c_library_and_dist_py_issue_the_same_schedule(world, B):
    lib = _lib()
    d, l = dist_schedules(world, B), lib_schedules(lib, world, B)
    for k, what in enumerate(("allgather_frames", "halo+sums", "sums only")):
        assert_pairs_match([r[k] for r in d], "dist.py " + what)
        assert_pairs_match([r[k] for r in l], "libv3d_comm " + what)
        for rank in range(world):
            assert coalesce(d[rank][k]) == coalesce(l[rank][k]), (what, rank, d[rank][k], l[rank][k])
    # the all-gather and the statistics are message-for-message identical (sizes per peer in order); only the halo is split differently
    for rank in range(world):
        for k in (0, 2):
            for direction in ("send", "recv"):
                assert [m for m in d[rank][k] if m[0] == direction] == [m for m in l[rank][k] if m[0] == direction], (k, rank, direction)


def test_uneven_shard_moves_exact_bytes():
    """18 frames over 8 ranks = 3,3,2,2,2,2,2,2: a rank sends its own frames once per peer and receives every peer's exact frame count"""
    lib = _lib()
    for rank in range(8):
        tl = comm.frame_range(lib, T, 8, rank)[1]
        msgs = comm.debug_schedule(lib, 0, rank, 8, B=1, T_global=T, frame_bytes=FRAME_BYTES)
        assert sum(m[4] for m in msgs if m[0] == 1) == 7 * tl * FRAME_BYTES
        assert sum(m[4] for m in msgs if m[0] == 0) == (T - tl) * FRAME_BYTES
        # receives land in disjoint frame ranges of `out` that do not touch the own frames
        spans = sorted((m[3], m[3] + m[4]) for m in msgs if m[0] == 0)
        t0 = comm.frame_range(lib, T, 8, rank)[0]
        spans.append((t0 * FRAME_BYTES, (t0 + tl) * FRAME_BYTES))
        spans.sort()
        assert spans[0][0] == 0 and spans[-1][1] == T * FRAME_BYTES and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
