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


def test_periods_around_the_batch_width_and_codes_beyond_the_root_tables(ctx):
    """What the wide token loop (xm_inflate_core.h: a lane per bit offset decodes, a lane per output byte produces, 64 bytes a
    batch) has edges of its own for: repeating units of 1 .. 66, 127 .. 130 and 257 .. 260 bytes (matches that overlap themselves,
    matches whose source is a token of the same batch, matches of 258 bytes that run over five batches), short literal / match
    mixes, and alphabets skewed enough for codes longer than the 10 / 8 bits of the root tables (tokens the serial reader takes
    in the middle of a window); every level that changes zlib's parsing, dynamic and fixed codes."""
    rng = np.random.default_rng(99)
    members, raws = [], []

    def add(raw, level, strategy=zlib.Z_DEFAULT_STRATEGY):
        d = deflate_raw(raw, level, strategy)
        assert len(d) + 26 <= 65536
        members.append(bgzf_member(d, raw))
        raws.append(raw)

    for period in list(range(1, 67)) + [127, 128, 129, 130, 257, 258, 259, 260]:
        unit = rng.integers(0, 256, period, dtype=np.uint8).tobytes()
        n = int(rng.integers(300, 3000))
        raw = bytearray((unit * (n // period + 1))[:n])
        for k in range(0, n, int(rng.integers(40, 400))):                                 # a literal now and then breaks the runs
            raw[k] = int(rng.integers(0, 256))
        for level, strategy in ((1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED)):
            add(bytes(raw), level, strategy)
    # geometric symbol frequencies: code lengths up to 15 bits in the literal / length code, long distance codes too
    for k in range(40):
        n = int(rng.integers(2000, 60000))
        ranks = np.minimum(rng.geometric(0.5 if k % 2 else 0.3, n) - 1, 255).astype(np.uint8)
        perm = rng.permutation(256).astype(np.uint8)
        raw = bytearray(perm[ranks].tobytes())
        for _ in range(int(rng.integers(0, 200))):                                        # copies at skewed distances
            ln, src = int(rng.integers(3, 259)), int(rng.integers(0, max(n - 300, 1)))
            dst = min(n - ln, src + int(min(rng.geometric(0.002), 32000)))
            if dst > src and dst + ln <= n:
                raw[dst:dst + ln] = raw[src:src + ln]
        add(bytes(raw), int(rng.choice([1, 6, 9])), zlib.Z_HUFFMAN_ONLY if k % 5 == 0 else zlib.Z_DEFAULT_STRATEGY)
    image = np.frombuffer(b"".join(members), dtype=np.uint8)
    n, total = check_image(ctx, image)
    assert n == len(members) and total == sum(len(r) for r in raws)
    # ... and every one of those streams cut short or with a flipped bit ends with a status or a CRC mismatch, inside its block
    from xenomapper_amd import _ffi
    blocks, crc, _nxt, total = _ffi.bgzf_index(image)
    damaged = bytearray(image.tobytes())
    for b in range(len(blocks)):
        at = int(blocks["cdata_off"][b]) + int(rng.integers(0, int(blocks["cdata_len"][b])))
        damaged[at] ^= 1 << int(rng.integers(0, 8))
    out, status, got_crc = gpu_inflate(ctx, np.frombuffer(bytes(damaged), dtype=np.uint8), blocks, total)
    assert (out[:64] == 0xEE).all() and (out[64 + total:] == 0xEE).all()
    want = np.frombuffer(b"".join(raws), dtype=np.uint8)
    for b in range(len(blocks)):
        o, nb = int(blocks["out_off"][b]), int(blocks["isize"][b])
        assert status[b] != 0 or got_crc[b] != crc[b] or np.array_equal(out[64 + o:64 + o + nb], want[o:o + nb]), b


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


def test_random_bytes_as_deflate_data_end_with_a_status(ctx):
    """BGZF members whose DEFLATE data are random bytes, or a valid stream cut short, or a valid stream with a wrong ISIZE: every
    loop of the decoder ends whatever the data holds (the wide token loop's own argument is written at its head), every such
    block ends with a status or a CRC that does not match, and nothing is written outside the blocks' own output."""
    from xenomapper_amd import _ffi
    rng = np.random.default_rng(31)
    members, sizes = [], []
    good = deflate_raw(make_payload(rng, 1, 30000), 6, zlib.Z_DEFAULT_STRATEGY)
    for k in range(600):
        kind = k % 4
        if kind == 0:
            d = rng.integers(0, 256, int(rng.integers(1, 4000)), dtype=np.uint8).tobytes()
        elif kind == 1:                                                                   # a dynamic-block header, then noise
            d = bytes([0b101]) + rng.integers(0, 256, int(rng.integers(8, 3000)), dtype=np.uint8).tobytes()
        elif kind == 2:
            d = good[:int(rng.integers(1, len(good)))]                                    # cut short
        else:
            d = good
        isize = int(rng.integers(0, 65537)) if kind != 3 else int(rng.choice([0, 1, 29999, 30001, 65536]))
        members.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(d) + 25) + d +
                       struct.pack("<II", 0xDEADBEEF, isize))
        sizes.append(isize)
    image = np.frombuffer(b"".join(members), dtype=np.uint8)
    blocks, crc, nxt, total = _ffi.bgzf_index(image)
    assert len(blocks) == 600 and nxt == image.shape[0] and total == sum(sizes)
    out, status, got_crc = gpu_inflate(ctx, image, blocks, total)
    assert (out[:64] == 0xEE).all() and (out[64 + total:] == 0xEE).all()
    assert all(status[b] != 0 or got_crc[b] != 0xDEADBEEF for b in range(600) if sizes[b])
    assert int((status != 0).sum()) > 400


def test_crc32_of_blocks_of_every_size_class_at_every_alignment(ctx):
    """xm_bgzf_crc32_dev by itself (its pieces are counted from the block's END, fetched 64 / 16 / 4 / 1 bytes a trip, and combined
    with powers of x shared by the workgroup): blocks of 0 .. 65 536 bytes around every boundary of those trips, each at an
    address of its own modulo 16, against zlib.crc32."""
    import torch
    from xenomapper_amd import _ffi
    rng = np.random.default_rng(7)
    sizes = [0, 1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 63, 64, 65, 127, 255, 256, 257, 511, 512, 513, 1020, 1023, 1024, 1025, 1027,
             4095, 4096, 4097, 16383, 16384, 16385, 16639, 16640, 16641, 65023, 65279, 65280, 65281, 65535, 65536]
    sizes += [int(x) for x in rng.integers(0, 65537, 60)]
    data = rng.integers(0, 256, sum(sizes) + 17 * len(sizes) + 64, dtype=np.uint8)
    blocks = np.zeros(len(sizes), dtype=_ffi.BGZF_BLOCK)
    at = 0
    for k, n in enumerate(sizes):
        at += 1 + (k * 5) % 16                                           # every residue modulo 16 comes up
        blocks["out_off"][k], blocks["isize"][k] = at, n
        at += n
    dev = torch.device("cuda:0")
    out = torch.from_numpy(data).to(dev)
    d_blocks = torch.from_numpy(blocks.view(np.uint8)).to(dev)
    crc = torch.zeros(len(sizes), dtype=torch.int32, device=dev)
    ctx.bgzf_crc32_dev(out, d_blocks, crc)
    torch.cuda.synchronize()
    got = crc.cpu().numpy().view(np.uint32)
    want = np.array([zlib.crc32(data[int(b["out_off"]):int(b["out_off"]) + int(b["isize"])].tobytes()) for b in blocks], dtype=np.uint32)
    assert np.array_equal(got, want), [(sizes[k], int(blocks["out_off"][k]) % 16) for k in np.nonzero(got != want)[0]]
    assert len(set(int(o) % 16 for o in blocks["out_off"])) == 16


def test_empty_input_and_empty_blocks(ctx):
    from xenomapper_amd import _ffi
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    image = np.frombuffer(eof * 3, dtype=np.uint8)
    blocks, crc, nxt, total = _ffi.bgzf_index(image)
    assert len(blocks) == 3 and total == 0 and nxt == len(image)
    out, status, got_crc = gpu_inflate(ctx, image, blocks, 0)
    assert (status == 0).all() and (out == 0xEE).all() and (got_crc == 0).all()


def test_records_found_and_read_by_the_inflating_chains(ctx, tmp_path):
    """xm_bgzf_inflate_walk_dev: every chain of lanes also follows the alignment records' block_size chain through the block it
    wrote and reads the stripper's fields out of its records.  Record-aligned blocks (as samtools writes them): counts, record
    starts and exits equal a plain walk of the inflated bytes on the host, block by block, the exits land on the next block's
    first byte, the first block is entered behind the BAM header, a window that ends inside a record stops in front of it; the
    fields (name position and length, AS, XS) are those the text of the record holds; blocks that cut records (fixed-size
    blocks) report exits that are NOT the next block's start."""
    import torch
    import bench_bam
    from xenomapper_amd import _ffi, _host
    dev = torch.device("cuda:0")
    for aligned in (True, False):
        path = str(tmp_path / ("walk_%d.bam" % aligned))
        bench_bam.tiled_bam(os.path.join(DATA, "paired_end_testdata_human.bam"), path, 40, aligned=aligned)
        image = np.fromfile(path, dtype=np.uint8)
        blocks, _crc, _nxt, total = _ffi.bgzf_index(image)
        want = np.frombuffer(gzip.decompress(image.tobytes()), dtype=np.uint8)
        reader = _host.BamReader(image, 1, header_only=True)
        first = reader.records_start()
        n_raw = total - 1000                                               # the window ends inside a record
        comp = torch.zeros(image.shape[0] + _ffi.BGZF_COMP_PAD, dtype=torch.uint8, device=dev)
        comp[:image.shape[0]] = torch.from_numpy(image).to(dev)
        d_blocks = torch.from_numpy(blocks.view(np.uint8)).to(dev)
        out = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
        nb = len(blocks)
        status = torch.full((nb,), -1, dtype=torch.int32, device=dev)
        work = torch.zeros(1, dtype=torch.int32, device=dev)
        cnt = torch.full((nb,), -7, dtype=torch.int32, device=dev)
        ext = torch.full((nb,), -7, dtype=torch.int32, device=dev)
        cap = 65536 // 36 + 2
        slots, name_off, name_len, a, x, ncig, cig_at = (torch.full((nb * cap,), -1, dtype=torch.int32, device=dev) for _ in range(7))
        flag = torch.full((nb * cap,), 0xEE, dtype=torch.uint8, device=dev)
        walk = np.zeros(nb, dtype=_ffi.BGZF_WALK)
        starts = blocks["out_off"].astype(np.int64).copy()
        ends = starts + blocks["isize"].astype(np.int64)
        j0 = int(np.searchsorted(ends, first, side="right"))             # the block the first record lies in
        for k in range(nb):
            if k < j0 or blocks["isize"][k] == 0:
                continue                                                   # header blocks, the empty end-of-file block: no walk
            at = 4 * cap * k
            walk[k] = (0, max(int(starts[k]), first) if k == j0 else int(starts[k]), int(ends[k]), n_raw, cap,
                       cnt.data_ptr() + 4 * k, ext.data_ptr() + 4 * k, slots.data_ptr() + at, name_off.data_ptr() + at,
                       name_len.data_ptr() + at, a.data_ptr() + at, x.data_ptr() + at, flag.data_ptr() + cap * k,
                       ncig.data_ptr() + at, cig_at.data_ptr() + at, _ffi.BGZF_TAGS_AS_XS, 0)
        d_walk = torch.from_numpy(walk.view(np.uint8)).to(dev)
        ctx.bgzf_inflate_dev(comp, d_blocks, out, status, work, walk=d_walk)
        torch.cuda.synchronize()
        assert (status.cpu().numpy() == 0).all()
        assert np.array_equal(out[:total].cpu().numpy(), want)
        h_cnt, h_ext = cnt.cpu().numpy(), ext.cpu().numpy().view(np.uint32)
        h = [t.cpu().numpy().reshape(nb, cap) for t in (slots, name_off, name_len, a, x, flag, ncig, cig_at)]
        # the text of all records, to read names and tags from (the printer is pinned to the oracle in test_host_fuzz)
        rec_all = np.empty(total // 36 + 8, dtype=np.uint32)
        n_all, _stop = _host.bam_walk(want.ctypes.data, total, first, rec_all)
        text = np.empty(8 * total, dtype=np.uint8)
        loff, llen = np.empty(n_all + 1, dtype=np.uint32), np.empty(n_all + 1, dtype=np.uint32)
        reader.print_records(want.ctypes.data, rec_all.ctypes.data, n_all, text, loff, llen)
        line_of = {int(rec_all[i]): bytes(text[int(loff[i]):int(loff[i]) + int(llen[i])]) for i in range(n_all)}
        reader.close()
        lands = 0
        for k in range(nb):
            if not walk["end"][k]:
                assert h_cnt[k] == -7                                      # skipped entries are not written
                continue
            p, hi, recs = int(walk["start"][k]), int(walk["end"][k]), []
            while p < hi:                                                  # the host restatement of the walk (xm_bamdev.hip walk_kernel)
                if hi - p < 4:
                    break
                size = int(want[p]) | int(want[p + 1]) << 8 | int(want[p + 2]) << 16 | int(want[p + 3]) << 24
                if size > n_raw - p - 4:
                    break
                recs.append(p)
                p += 4 + size
            assert int(h_cnt[k]) == len(recs) and int(h_ext[k]) == p, (aligned, k)
            assert h[0][k, :len(recs)].view(np.uint32).tolist() == recs
            if aligned:                                                    # whole records inside the block: the fields are the text's
                for i, r in enumerate(recs[:5] + recs[-3:]):
                    i = recs.index(r)
                    fields = line_of[r].split(b"\t")
                    assert int(h[1][k, i]) == r + 36 and int(h[2][k, i]) == len(fields[0])
                    tags = {f[:2]: f for f in fields[11:]}
                    for col, tag in ((3, b"AS"), (4, b"XS")):
                        wantv = int(tags[tag].split(b":")[2]) if tag in tags else -2**31
                        assert int(h[col][k, i]) == wantv, (k, i, tag)
                    assert int(h[5][k, i]) == 0
                    # the CIGAR words where the record says they are, as many as its header says
                    n_cigar = int(want[r + 4 + 12]) | int(want[r + 4 + 13]) << 8
                    assert int(h[6][k, i]) == n_cigar and int(h[7][k, i]) == r + 36 + int(want[r + 4 + 8])
            lands += int(k + 1 < nb and p == int(walk["end"][k]))
        walked = int((walk["end"] > 0).sum())
        if aligned:
            assert lands >= walked - 2                                     # all but the block the window ends in (and the last)
        else:
            assert lands < walked // 2
