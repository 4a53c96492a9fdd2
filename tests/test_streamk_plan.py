"""The stream-K decomposition of a persistent launch's last round (v3d_amd/csrc/gemm_common.h sk_build_table), checked on the CPU for every
(tiles, granules, blocks) the kernels can meet: the library exports the device code's own table builder as a host function.

Properties (they are what makes the hand-off deadlock-free and the result complete):
  * the pieces of all blocks cover every granule of every tile of the last round exactly once, whole tiles of the full rounds exactly once;
  * a block has at most one donor piece and it is its first item; at most one owner piece and it is its last;
  * an owner's donors d0 .. d1 are exactly the blocks that hold the rest of its tile, each as its donor piece - so donors never wait for
    anybody and every owner waits only for blocks that run their donor piece first: no cycle;
  * no block of a planned launch is left without work in the tail (a donor with nothing to publish would never raise its flag)."""
import ctypes
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    path = os.path.join(ROOT, "v3d_amd", "lib", "libv3d_hip.so")
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    lb = ctypes.CDLL(path)
    lb.v3d_debug_sk_table.restype = ctypes.c_int
    lb.v3d_debug_sk_table.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_int)]
    return lb


def plan(lib, ntiles, units, G, min_units, min_saved):
    out = (ctypes.c_int * 14)()
    blocks = []
    for b in range(G):
        if not lib.v3d_debug_sk_table(ntiles, units, G, b, min_units, min_saved, out):
            return None
        n, donor = out[0], out[1]
        pieces = [tuple(out[2 + 6 * k + j] for j in range(6)) for k in range(2)]
        blocks.append((n, donor, pieces))
    return blocks


def check(lib, ntiles, units, G, min_units=1, min_saved=0):
    blocks = plan(lib, ntiles, units, G, min_units, min_saved)
    if blocks is None:
        return False
    full, R = ntiles // G, ntiles % G
    cover = {}                                        # (tile, granule) -> block
    donor_of, owners = {}, []
    for b, (n, donor, pieces) in enumerate(blocks):
        ntail = n - full
        assert 1 <= ntail <= 2, f"block {b}: {ntail} pieces of the last round (every block of a planned launch gets one or two)"
        tail = pieces[:ntail]
        assert donor == (1 if tail[0][3] == 1 else 0)
        for k, (tile, u0, u1, role, d0, d1) in enumerate(tail):
            assert full * G <= tile < ntiles and 0 <= u0 < u1 <= units
            assert role == (1 if u0 > 0 else (2 if u1 < units else 0))
            assert not (role == 1 and k != 0), "a donor piece must be the block's first"
            assert not (role == 2 and k != ntail - 1), "an owner piece must be the block's last"
            for u in range(u0, u1):
                assert (tile, u) not in cover, f"granule {u} of tile {tile} is covered twice"
                cover[(tile, u)] = b
            if role == 1:
                donor_of[b] = tile
            if role == 2:
                owners.append((b, tile, d0, d1))
    assert len(cover) == R * units, "the pieces do not cover the last round"
    for b, tile, d0, d1 in owners:
        rest = sorted({cover[(tile, u)] for u in range(units)} - {b})
        assert rest == list(range(d0, d1 + 1)), f"owner {b} of tile {tile} waits for {d0}..{d1}, the tile's other pieces are on {rest}"
        assert all(donor_of.get(d) == tile for d in rest), "an owner waits for a block whose donor piece is another tile's"
        assert d0 == b + 1
    # every partial tile has exactly one owner
    partial = {t for (t, _), _ in cover.items()} - {t for b, (n, d, ps) in enumerate(blocks) for (t, u0, u1, r, _, _) in ps[:n - full] if r == 0}
    assert partial == {t for _, t, _, _ in owners}
    return True


def test_v3d_shapes(lib):
    """The launches of the V3D_512 U-Net that take a tail on 256 CUs (192 x 320 tiles): 32 x 32 level (384 tiles), 16 x 16 level (192 tiles),
    8 x 8 temporal (48 tiles)."""
    assert check(lib, 384, 20, 256, 4, 2)            # conv3x3 640 -> 640
    assert check(lib, 384, 60, 256, 4, 2)            # 1920 -> 640
    assert check(lib, 192, 40, 256, 4, 2)            # 1280 -> 1280
    assert check(lib, 192, 80, 256, 4, 2)
    assert check(lib, 48, 40, 256, 4, 2)             # five or six blocks per tile
    assert not check(lib, 768, 10, 256, 4, 2)        # three full rounds: classic
    assert not check(lib, 250, 40, 256, 4, 2)        # the last round is nearly full: classic
    assert not check(lib, 64, 4, 256, 4, 2)          # pieces would be shorter than 4 chunks: classic


def test_every_small_configuration(lib):
    """Exhaustive over small launches (blocks 2..12, tiles up to 3 rounds, 1..9 granules): every planned one satisfies the properties."""
    planned = 0
    for G in range(2, 13):
        for ntiles in range(1, 3 * G + 1):
            for units in range(1, 10):
                planned += bool(check(lib, ntiles, units, G))
    assert planned > 500


@pytest.mark.parametrize("G", [256, 304, 64, 8])
def test_large_grids(lib, G):
    for ntiles in (G // 4 + 1, G // 2, G - G // 8, G + 1, G + G // 2, 2 * G + G // 3, 5 * G - G // 8):
        for units in (4, 10, 37, 160, 1440):
            check(lib, ntiles, units, G, 1, 0)
