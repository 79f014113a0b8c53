"""BGZF blocks inflated on the GPU (include/xenomapper_bgzf.h) against zlib -- byte for byte, plus the CRC-32 kernel against
the member trailers: the reference's own BAM fixtures, the long-CIGAR fixture, the fixtures tiled into thousands of
blocks, and raw-DEFLATE streams zlib wrote at every level / strategy (stored, fixed and dynamic blocks, distance-1 runs,
matches beyond the output ring), at shifted alignments.  Damaged streams must end with a status and must not write
outside their own block."""
import gzip
import os
import struct
import sys
import zlib

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

DATA = os.path.join(H.REPO, "tests", "golden", "ref_data")
sys.path.insert(0, os.path.join(H.REPO, "tools"))


@pytest.fixture(scope="module")
def ctx():
    from xenomapper_amd import _ffi
    c = _ffi.Context(0)
    yield c
    c.close()


def gpu_inflate(ctx, image, blocks, total, want_crc=True, poison=0xEE):
    """-> (inflated bytes as a host array incl. 64 guard bytes at either end, status per block, crc per block)"""
    import torch
    from xenomapper_amd import _ffi
    dev = torch.device("cuda:0")
    comp = torch.zeros(image.shape[0] + _ffi.BGZF_COMP_PAD, dtype=torch.uint8, device=dev)
    comp[:image.shape[0]] = torch.from_numpy(np.ascontiguousarray(image)).to(dev)
    shifted = blocks.copy()
    shifted["out_off"] += 64                                               # guard bytes in front
    d_blocks = torch.from_numpy(shifted.view(np.uint8)).to(dev)
    out = torch.full((total + 128,), poison, dtype=torch.uint8, device=dev)
    status = torch.full((max(len(blocks), 1),), -1, dtype=torch.int32, device=dev)
    work = torch.zeros(1, dtype=torch.int32, device=dev)
    crc = torch.zeros(max(len(blocks), 1), dtype=torch.int32, device=dev)
    ctx.bgzf_inflate_dev(comp, d_blocks, out, status, work)
    if want_crc:
        ctx.bgzf_crc32_dev(out, d_blocks, crc)
    torch.cuda.synchronize()
    return out.cpu().numpy(), status.cpu().numpy()[:len(blocks)], crc.cpu().numpy().view(np.uint32)[:len(blocks)]


def check_image(ctx, image):
    from xenomapper_amd import _ffi
    blocks, crc, nxt, total = _ffi.bgzf_index(image)
    assert nxt == image.shape[0]
    want = np.frombuffer(gzip.decompress(image.tobytes()), dtype=np.uint8)
    assert want.shape[0] == total
    out, status, got_crc = gpu_inflate(ctx, image, blocks, total)
    assert (status == 0).all(), [(int(b), _ffi.bgzf_strerror(s)) for b, s in enumerate(status) if s][:5]
    assert np.array_equal(out[64:64 + total], want)
    assert (out[:64] == 0xEE).all() and (out[64 + total:] == 0xEE).all()   # nothing outside the blocks
    assert np.array_equal(got_crc, crc)
    for b in (0, len(blocks) // 2, len(blocks) - 1):                        # the kernel's CRC against zlib's, too
        o, n = int(blocks["out_off"][b]), int(blocks["isize"][b])
        assert int(got_crc[b]) == zlib.crc32(want[o:o + n].tobytes())
    return len(blocks), total


def test_reference_bam_fixtures_and_long_cigars(ctx):
    for path in (os.path.join(DATA, "paired_end_testdata_human.bam"), os.path.join(DATA, "paired_end_testdata_mouse.bam"),
                 os.path.join(H.GOLDEN, "long_cigar_cg.bam")):
        n, total = check_image(ctx, np.fromfile(path, dtype=np.uint8))
        assert n >= 4 and total > 100_000


def test_tiled_fixture_thousands_of_blocks(ctx, tmp_path):
    import bench_bam
    path = str(tmp_path / "tiled.bam")
    bench_bam.tiled_bam(os.path.join(DATA, "paired_end_testdata_human.bam"), path, 1500)
    n, total = check_image(ctx, np.fromfile(path, dtype=np.uint8))
    assert n > 2500 and total > 150_000_000


def bgzf_member(deflated, raw):
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(deflated) + 25) + deflated +
            struct.pack("<II", zlib.crc32(raw), len(raw)))


def deflate_raw(raw, level, strategy):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return c.compress(raw) + c.flush()


def make_payload(rng, kind, n):
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()                           # incompressible: stored blocks
    if kind == 1:
        return rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), n).tobytes()
    if kind == 2:
        a = np.full(n, ord("F"), dtype=np.uint8)
        a[rng.random(n) < 0.1] = ord(",")
        return a.tobytes()                                                               # long runs: distance-1 matches
    if kind == 3:
        unit = rng.integers(0, 256, 700, dtype=np.uint8).tobytes()
        return (unit * (n // 700 + 1))[:n]                                               # far matches, period 700
    if kind == 4:
        return bytes(n)
    a = rng.integers(0, 256, n, dtype=np.uint8)
    if n > 9000:
        a[-3000:] = a[:3000]                                                             # a match far beyond the ring
    return a.tobytes()


def test_zlib_streams_every_level_and_strategy(ctx):
    rng = np.random.default_rng(2024)
    strategies = (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED)
    members, raws = [], []
    for it in range(400):
        n = int(rng.choice([0, 1, 2, 15, 16, 17, 127, 128, 129, 1000, 65280, 65536])) if it % 4 == 0 else int(rng.integers(0, 65536))
        raw = make_payload(rng, int(rng.integers(0, 6)), n)
        level, strategy = int(rng.choice([0, 1, 4, 6, 9])), strategies[int(rng.integers(0, 5))]
        d = deflate_raw(raw, level, strategy)
        if len(d) + 26 > 65536:                                                           # does not fit a BGZF member: stored form, split
            raw = raw[:60000]
            d = deflate_raw(raw, level, strategy)
        members.append(bgzf_member(d, raw))
        raws.append(raw)
    image = np.frombuffer(b"".join(members), dtype=np.uint8)
    n, total = check_image(ctx, image)
    assert n == 400 and total == sum(len(r) for r in raws)


def test_damaged_streams_end_with_a_status_and_stay_inside_their_block(ctx):
    from xenomapper_amd import _ffi
    rng = np.random.default_rng(7)
    image = bytearray(np.fromfile(os.path.join(H.GOLDEN, "long_cigar_cg.bam"), dtype=np.uint8).tobytes())
    blocks, crc, _, total = _ffi.bgzf_index(np.frombuffer(bytes(image), dtype=np.uint8))
    hit = []
    for b in range(0, len(blocks), 2):                                                    # every other block gets one flipped bit
        at = int(blocks["cdata_off"][b]) + int(rng.integers(0, int(blocks["cdata_len"][b])))
        image[at] ^= 1 << int(rng.integers(0, 8))
        hit.append(b)
    want = None
    out, status, got_crc = gpu_inflate(ctx, np.frombuffer(bytes(image), dtype=np.uint8), blocks, total)
    clean = np.frombuffer(gzip.decompress(np.fromfile(os.path.join(H.GOLDEN, "long_cigar_cg.bam"), dtype=np.uint8).tobytes()), dtype=np.uint8)
    assert (out[:64] == 0xEE).all() and (out[64 + total:] == 0xEE).all()
    for b in range(len(blocks)):
        o, n = 64 + int(blocks["out_off"][b]), int(blocks["isize"][b])
        if b not in hit:                                                                  # untouched blocks are untouched by their neighbours
            assert status[b] == 0 and np.array_equal(out[o:o + n], clean[o - 64:o - 64 + n]) and got_crc[b] == crc[b]
        else:                                                                             # a flipped bit is caught by the decoder or by the CRC
            assert status[b] != 0 or got_crc[b] != crc[b] or np.array_equal(out[o:o + n], clean[o - 64:o - 64 + n])


def test_empty_input_and_empty_blocks(ctx):
    from xenomapper_amd import _ffi
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    image = np.frombuffer(eof * 3, dtype=np.uint8)
    blocks, crc, nxt, total = _ffi.bgzf_index(image)
    assert len(blocks) == 3 and total == 0 and nxt == len(image)
    out, status, got_crc = gpu_inflate(ctx, image, blocks, 0)
    assert (status == 0).all() and (out == 0xEE).all() and (got_crc == 0).all()
