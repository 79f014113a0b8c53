"""N > 1 path on CPU: world_size 2 and 3 over gloo (127.0.0.1).  Each rank takes its read block with
the one-record halo, classifies it (here with the oracle standing in for the device, since this
container has no GPU; tests/test_shard_gpu.py does the same with the HIP kernels), all-reduces
category_counts and gathers its bin lists; rank 0 checks the result equals the unsharded oracle."""
import os
import socket

import numpy as np
import pytest

from tests import helpers as H


class OracleCtx(object):
    """Test stand-in with the Context.classify / compact / classify_compact signatures, backed by the C oracle."""

    def classify_compact(self, mode, as1, xs1, as2, xs2, unit_bits, m, want_code=True):
        code, counts = H.c_classify(mode, as1, xs1, as2, xs2, unit_bits, m)
        idx, off = H.c_compact(mode, code)
        return (code if want_code else None), idx, off, counts

    def classify(self, mode, as1, xs1, as2, xs2, unit_bits, m):
        return H.c_classify(mode, as1, xs1, as2, xs2, unit_bits, m)

    def compact(self, mode, code):
        idx, off = H.c_compact(mode, code)
        counts = np.bincount(code[code != 0xFF], minlength=64).astype(np.uint64)
        return idx, off, counts


def make_input(n, seed):
    rng = np.random.default_rng(seed)
    vals = np.concatenate([[-2**31, -2**31], np.arange(-6, 7)]).astype(np.int64)
    cols = [vals[rng.integers(0, len(vals), n)].astype(np.int32) for _ in range(4)]
    # names: mostly pairs, some singletons and triples -> unit flags
    sizes = rng.choice([1, 2, 3], size=n, p=[0.1, 0.8, 0.1])
    names = np.repeat(np.arange(n), sizes)[:n]
    flags = np.zeros(n, dtype=np.uint8)
    flags[1:] = names[1:] == names[:-1]
    return cols, H.synth.pack_unit_bits(flags)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, mode, use_gpu, ret):
    import torch
    import torch.distributed as dist
    from xenomapper_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cols, bits = make_input(n, 42)
        if use_gpu:
            from xenomapper_amd import _ffi
            ctx = _ffi.Context(0)
        else:
            ctx = OracleCtx()
        blocks = shard.plan_blocks(n, world)
        start, end = blocks[rank]
        lists, counts = shard.classify_block(ctx, mode, cols, bits, n, start, end, -2**31)
        t = torch.from_numpy(counts.astype(np.int64))
        shard.allreduce_counts(t)
        all_lists = shard.gather_bin_lists(lists, dst=0)
        if rank == 0:
            code, want_counts = H.c_classify(mode, *cols, bits, -2**31)
            want_idx, want_off = H.c_compact(mode, code)
            assert np.array_equal(t.numpy().astype(np.uint64), want_counts)
            for b in range(7):
                assert np.array_equal(all_lists[b], want_idx[int(want_off[b]):int(want_off[b + 1])].astype(np.int64)), b
            ret.put("ok")
    finally:
        dist.destroy_process_group()


def run_world(world, n, mode, use_gpu=False):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, mode, use_gpu, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ret.get(timeout=5) == "ok"


def test_plan_blocks_cover_and_align():
    from xenomapper_amd import shard
    for n in (0, 1, 63, 64, 65, 1000, 100_003):
        for world in (1, 2, 3, 8):
            blocks = shard.plan_blocks(n, world)
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (a0, a1), (b0, b1) in zip(blocks, blocks[1:]):
                assert a1 == b0 and a0 <= a1
            assert all(s % 64 == 0 or s == n for s, _ in blocks)


def test_halo_keeps_overlapping_units():
    """A triple QNAME run that straddles the cut still yields both overlapping pairs, once each."""
    from xenomapper_amd import shard
    n = 130
    cols = [np.arange(n, dtype=np.int32) % 7 for _ in range(4)]
    flags = np.zeros(n, dtype=np.uint8)
    flags[63] = flags[64] = flags[65] = 1          # records 62..65 share a name: units at 63, 64, 65
    bits = H.synth.pack_unit_bits(flags)
    got = []
    for start, end in shard.plan_blocks(n, 2):
        lists, _ = shard.classify_block(OracleCtx(), 1, cols, bits, n, start, end, -2**31)
        got += [int(i) for l in lists for i in l]
    assert sorted(got) == [63, 64, 65]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_unsharded_gloo(world):
    run_world(world, 50_007, 1)
    run_world(world, 4_100, 2)


def test_eight_ranks_like_config_4():
    """BASELINE.json configs[3]'s shape -- eight read blocks, one per rank, halo at every cut, one all-reduce -- at a
    size the CPU runs in seconds (the full 400 M pairs: tests/test_shard_gpu.py)."""
    run_world(8, 80_011, 1)
