#!/usr/bin/env python3
"""Randomised soak of the device paths against the C oracle (runs on the GPU box; not part of the test-suite because it
runs for minutes):  python tools/stress_gpu.py --seconds 240 [--seed S]

Every round draws a size, a mode, a unit-mask style and a score / CIGAR style (short CIGARs, long ones, escaped records
of 255+ operations at random places, tiles of 1000+ operations, absent NM, op codes 0..15), runs
  * xm_classify_compact (AS/XS columns, host buffers: category bytes or compact stream between K1 and K2c),
  * xm_classify_compact_cigar (CSR in, packed inside) and xm_classify_compact_cigar_packed_dev (every output form)
and compares codes, counts, bin offsets and index lists with the oracle.  Prints one line per 50 rounds; exits non-zero
at the first difference, with the seed of the round."""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
ABSENT = -2**31


def unit_flags(rng, n, mode):
    style = rng.integers(0, 5)
    if mode == 0:
        return (rng.random(n) < (1.0 if style < 2 else rng.random())).astype(np.uint8)
    if style == 0:                                   # strictly interleaved mates
        f = np.zeros(n, dtype=np.uint8)
        f[1::2] = 1
        return f
    if style == 1:                                   # singletons / triples mixed in
        sizes = rng.choice([1, 2, 3], size=n, p=[0.05, 0.9, 0.05])
        names = np.repeat(np.arange(n), sizes)[:n]
        f = np.zeros(n, dtype=np.uint8)
        f[1:] = names[1:] == names[:-1]
        return f
    if style == 2:                                   # dense: long runs of equal names
        f = np.ones(n, dtype=np.uint8)
        f[0] = 0
        f[rng.random(n) < 0.01] = 0
        return f
    f = (rng.random(n) < rng.random()).astype(np.uint8)
    if n:
        f[0] = 0
    return f


def random_csr(rng, n):
    style = rng.integers(0, 4)
    k = rng.integers(0, 5, n).astype(np.int64)
    k[rng.random(n) < 0.5] = 1
    if style >= 1 and n:
        for _ in range(int(rng.integers(1, 6))):     # escaped records at random places, sometimes adjacent
            at = int(rng.integers(0, n))
            k[at] = int(rng.choice([254, 255, 256, 300, 1000, 5000]))
            if rng.random() < 0.3 and at + 1 < n:
                k[at + 1] = 255
    if style == 2 and n >= 256:
        t = int(rng.integers(0, n // 256)) * 256     # a whole tile of 4 .. 6-operation records: 1000+ operations in the stretch
        k[t:t + 256] = rng.integers(4, 7, 256)
    if style == 3:
        k[:] = 0                                     # nothing mapped
    off = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum(k, out=off[1:])
    total = int(off[-1])
    ops = (rng.integers(1, 40, total).astype(np.uint32) << 4) | rng.integers(0, 16 if rng.random() < 0.3 else 9, total).astype(np.uint32)
    nm = np.where(rng.random(n) < 0.2, ABSENT, rng.integers(0, 8, n)).astype(np.int32)
    return {"nm": nm, "cig_off": off, "cig_oplen": ops}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args()
    import torch
    from tests import helpers as H
    from xenomapper_amd import _ffi
    seed0 = a.seed if a.seed is not None else int(time.time())
    ctx = _ffi.Context(0)
    dev = torch.device("cuda:0")
    t_end = time.time() + a.seconds
    rounds = 0
    while time.time() < t_end:
        seed = seed0 + rounds
        rng = np.random.default_rng(seed)
        n = int(rng.choice([1, 5, 255, 256, 257, 2047, 2048, 2049, 4096, 6000, 20_000, 70_001, 300_003]))
        mode = int(rng.integers(0, 3))
        flags = unit_flags(rng, n, mode)
        bits = H.synth.pack_unit_bits(flags)
        mi = int(rng.choice([ABSENT, -40, 0, 3]))
        what = "?"
        try:
            # --- AS/XS columns -------------------------------------------------------------------------------------
            what = "AS/XS"
            vals = np.concatenate([[ABSENT, ABSENT], np.arange(-8, 9)]).astype(np.int64)
            cols = [vals[rng.integers(0, len(vals), n)].astype(np.int32) for _ in range(4)]
            want_code, want_counts = H.c_classify(mode, *cols, bits, mi)
            want_idx, want_off = H.c_compact(mode, want_code)
            for want_bytes in (True, False):
                code, idx, off, counts = ctx.classify_compact(mode, *cols, bits, mi, want_code=want_bytes)
                assert code is None or np.array_equal(code, want_code)
                assert np.array_equal(counts, want_counts) and np.array_equal(off, want_off) and np.array_equal(idx, want_idx)
            # --- CIGAR columns -------------------------------------------------------------------------------------
            what = "CIGAR host"
            c1, c2 = random_csr(rng, n), random_csr(rng, n)
            xs = [np.where(rng.random(n) < 0.8, ABSENT, -rng.integers(0, 300, n)).astype(np.int32) for _ in range(2)]
            a1, b1 = H.c_cigar_scores(c1["nm"], c1["cig_off"], c1["cig_oplen"])
            a2, b2 = H.c_cigar_scores(c2["nm"], c2["cig_off"], c2["cig_oplen"])
            assert b1 == 0 and b2 == 0
            want_code, want_counts = H.c_classify(mode, a1, xs[0], a2, xs[1], bits, mi)
            want_idx, want_off = H.c_compact(mode, want_code)
            code, idx, off, counts = ctx.classify_compact_cigar(mode, c1["nm"], c1["cig_off"], c1["cig_oplen"], xs[0],
                                                                c2["nm"], c2["cig_off"], c2["cig_oplen"], xs[1], bits, mi)
            assert np.array_equal(code, want_code) and np.array_equal(counts, want_counts)
            assert np.array_equal(off, want_off) and np.array_equal(idx, want_idx)
            what = "CIGAR packed dev"
            d = []
            for c, x in ((c1, xs[0]), (c2, xs[1])):
                cnt, tile, ops = _ffi.cigar_pack(c["cig_off"], c["cig_oplen"])
                if ops.shape[0] == 0:
                    ops = np.zeros(1, dtype=np.uint32)
                d += [torch.from_numpy(v).to(dev) for v in (c["nm"], cnt, tile.view(np.int32), ops.view(np.int32), x)]
            d.append(torch.from_numpy(bits.view(np.int64)).to(dev))
            want_bins = np.full(n, 7, dtype=np.uint8)
            for b in range(7):
                want_bins[want_idx[int(want_off[b]):int(want_off[b + 1])]] = b
            form = int(rng.integers(0, 3))
            codet = torch.full((n + 16,), 0xAA, dtype=torch.uint8, device=dev) if form != 1 else None
            bins4 = torch.full((_ffi.bins4_bytes(n),), 0xAA, dtype=torch.uint8, device=dev) if form != 0 else None
            idxt = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
            offt = torch.zeros(8, dtype=torch.int64, device=dev)
            cntt = torch.zeros(64, dtype=torch.int64, device=dev)
            flag = torch.zeros(4, dtype=torch.int32, device=dev)
            ctx.classify_compact_cigar_packed_dev(mode, *d, mi, codet, idxt, offt, cntt, bins4=bins4, range_flag=flag)
            torch.cuda.synchronize()
            assert int(flag[0].item()) == 0
            assert codet is None or np.array_equal(codet[:n].cpu().numpy(), want_code)
            assert bins4 is None or np.array_equal(_ffi.unpack_bins4(bins4.cpu().numpy(), n), want_bins)
            assert np.array_equal(cntt.cpu().numpy().astype(np.uint64), want_counts)
            assert np.array_equal(offt.cpu().numpy().astype(np.uint64), want_off)
            assert np.array_equal(idxt[:int(want_off[7])].cpu().numpy().view(np.uint32), want_idx)
        except AssertionError:
            print("MISMATCH in %s: round seed %d (n=%d mode=%d min_score=%d)" % (what, seed, n, mode, mi), flush=True)
            raise
        rounds += 1
        if rounds % 50 == 0:
            print("%d rounds ok (last: n=%d mode=%d)" % (rounds, n, mode), flush=True)
    print("stress ok: %d rounds, seeds %d .. %d" % (rounds, seed0, seed0 + rounds - 1))


if __name__ == "__main__":
    main()
