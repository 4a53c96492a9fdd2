"""libv3d_comm.so (include/v3d_comm.h): the C ABI of the frame-axis exchanges over RCCL (SURVEY.md 8b last row).
CPU: the library loads, exports every symbol the header declares, and its frame partition is dist.py's.
-m gpu: ONE rank on the leased GPU (RCCL refuses two ranks per device): communicator creation, the all-gather's own-frames placement, the
statistics path (rank-order fp64 sum), a no-op halo exchange at the global ends, and a grouped self send / recv through RCCL."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    """the library, or a skip where it cannot exist (no hipcc / no RCCL under the ROCm tree / librccl unresolvable): it is optional (v3d_amd/build.py)"""
    from v3d_amd import comm
    from v3d_amd.build import build_comm
    try:
        build_comm(verbose=False)
        return comm.load_library()
    except (RuntimeError, OSError) as e:
        pytest.skip(f"libv3d_comm.so unavailable: {str(e).splitlines()[0]}")


def test_header_symbols_are_exported_and_bound():
    from v3d_amd import comm
    lib = _lib()
    hdr = open(os.path.join(ROOT, "include", "v3d_comm.h")).read()
    declared = set(re.findall(r"\b(v3d_comm_[a-z_0-9]+)\s*\(", hdr)) - {"v3d_comm_s"}
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/v3d_comm.h but not exported"
    assert declared == set(comm.SIGNATURES), (declared ^ set(comm.SIGNATURES))
    assert lib.v3d_comm_abi_version() == comm.ABI_VERSION


def test_frame_range_is_the_partition_of_dist_py():
    from v3d_amd import comm
    from v3d_amd.dist import frame_partition
    lib = _lib()
    for T in (1, 2, 3, 14, 18, 24, 25):
        for world in range(1, min(T, 9) + 1):
            parts = frame_partition(T, world)
            for r in range(world):
                assert comm.frame_range(lib, T, world, r) == (parts[r].start, len(parts[r])), (T, world, r)
    assert [comm.frame_range(lib, 18, 8, r)[1] for r in range(8)] == [3, 3, 2, 2, 2, 2, 2, 2]
    with pytest.raises(ValueError, match="cannot shard"):
        comm.frame_range(lib, 3, 4, 0)


@pytest.mark.gpu
def test_single_rank_communicator_on_hardware():
    from v3d_amd import comm
    lib = _lib()
    torch.cuda.set_device(0)
    c = comm.Comm(comm.Comm.unique_id(lib), 0, 1, lib)
    try:
        g = torch.Generator(device="cuda").manual_seed(3)
        kv = torch.randn(2, 18, 64, 128, device="cuda", generator=g).to(torch.bfloat16)           # [B, T, S, 2C]
        assert torch.equal(c.allgather_frames(kv, 18), kv)                                          # one rank: its frames are all frames
        sums = torch.randn(2, 32, 2, device="cuda", generator=g, dtype=torch.float64)
        buf = torch.randn((2 + 2 * 18 + 2) * 64, 128, device="cuda", generator=g).to(torch.bfloat16)
        before = buf.clone()
        total = c.exchange_halo_and_sums(buf, 2, 18, 64 * 128 * 2, sums)
        assert torch.equal(total, sums) and torch.equal(buf, before)                                # global ends: nothing sent, nothing received
        assert torch.equal(c.exchange_halo_and_sums(None, 0, 0, 0, sums), sums)                     # statistics only
        src = torch.randn(1 << 20, device="cuda", generator=g)
        dst = c.selftest(src)                                                                       # grouped ncclSend + ncclRecv to self
        torch.cuda.synchronize()
        assert torch.equal(dst, src)
    finally:
        c.destroy()
