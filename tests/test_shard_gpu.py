"""The N > 1 path with the real kernels on the one-GPU box:
* two ranks (gloo rendezvous on 127.0.0.1) share the GPU, each classifies its read block through the C ABI; counts are
  all-reduced, lists gathered;
* the collectives themselves on DEVICE memory with a one-rank communicator: the library's own RCCL entry point
  (xm_comm_init + xm_allreduce_counts) and torch.distributed's nccl backend, the call bench.py makes;
* BASELINE.json configs[3] at full size: 400 M read pairs as eight 50 M-pair read blocks with the one-record halo, run
  back to back through the device-resident fused entry point, against the unsharded C oracle."""
import os

import numpy as np
import pytest

from tests import helpers as H
from tests.test_shard_gloo import _free_port, run_world

pytestmark = pytest.mark.gpu


def test_two_ranks_one_gpu():
    run_world(2, 300_001, 1, use_gpu=True)
    run_world(2, 70_000, 2, use_gpu=True)


def test_library_rccl_allreduce_on_device_memory():
    """xm_comm_unique_id / xm_comm_init / xm_allreduce_counts / xm_comm_destroy: librccl is found, a communicator comes
    up on the box's GPU and the all-reduce runs on a device buffer (one rank: the sum is the input)."""
    import torch
    from xenomapper_amd import _ffi
    with _ffi.Context(0) as ctx:
        assert ctx.comm_size() == 0
        with pytest.raises(ValueError):
            ctx.allreduce_counts(torch.zeros(64, dtype=torch.int64, device="cuda:0"))      # no communicator yet
        uid = _ffi.comm_unique_id()
        assert len(uid) == _ffi.UNIQUE_ID_BYTES
        ctx.comm_init(1, 0, uid)
        assert ctx.comm_size() == 1
        counts = torch.arange(64, dtype=torch.int64, device="cuda:0") * 1_000_000_007
        ctx.allreduce_counts(counts)
        torch.cuda.synchronize()
        assert torch.equal(counts.cpu(), torch.arange(64, dtype=torch.int64) * 1_000_000_007)
        ctx.comm_destroy()
        assert ctx.comm_size() == 0
        ctx.comm_init(1, 0, _ffi.comm_unique_id())                                          # and again, then by ctx.close()
        assert ctx.comm_size() == 1


def test_torch_nccl_allreduce_of_a_device_tensor():
    """shard.allreduce_counts on a CUDA tensor under backend "nccl" (= RCCL): the exact call bench.py makes at N > 1,
    here in a one-rank group."""
    import torch
    import torch.distributed as dist
    from xenomapper_amd import shard
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        counts = torch.arange(64, dtype=torch.int64, device="cuda:0") + 5
        shard.allreduce_counts(counts)
        torch.cuda.synchronize()
        assert torch.equal(counts.cpu(), torch.arange(64, dtype=torch.int64) + 5)
        one = torch.ones(1, dtype=torch.int64, device="cuda:0")
        dist.all_reduce(one)
        assert int(one.item()) == dist.get_world_size() == 1
    finally:
        dist.destroy_process_group()


def _pack_bits_device(flags):
    """uint8 0/1 flags (length a multiple of 64) -> packed little-endian int64 words, on the device."""
    import torch
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=flags.device)
    by = (flags.view(-1, 8) * w).sum(dim=1, dtype=torch.uint8)
    return by.view(torch.int64)


@pytest.mark.parametrize("pairs_per_block", [int(os.environ.get("XM_CFG4_PAIRS_PER_BLOCK", "50000000"))])
def test_config4_eight_blocks_with_halo_full_size(pairs_per_block):
    """BASELINE.json configs[3]: 8 x 50 M pairs (seed 4004 + block, as each GPU would generate its own), treated as ONE
    800 M-record input whose units straddle the cuts (mates at (2k-1, 2k), so every block's first record closes a
    unit with the halo record in front of it).  Each block goes through shard.plan_blocks' range + halo and the
    device-resident fused entry point; category_counts are summed, bin lists put back into whole-input numbering and
    compared block by block with the unsharded C oracle -- bit for bit."""
    import torch
    from xenomapper_amd import _ffi, shard, synth
    world = 8
    dev = torch.device("cuda", 0)
    n = 2 * pairs_per_block * world
    cols = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in ("as1", "xs1", "as2", "xs2")}
    for g in range(world):
        part = synth.score_columns_torch(pairs_per_block, seed=4004 + g, device=dev)
        lo = 2 * pairs_per_block * g
        for k in cols:
            cols[k][lo:lo + 2 * pairs_per_block] = part[k]
        del part
    flags = torch.zeros(n, dtype=torch.uint8, device=dev)
    flags[2::2] = 1                                                   # units close at even records >= 2
    host = {k: v.cpu().numpy() for k, v in cols.items()}
    host_bits = _pack_bits_device(flags).cpu().numpy().view(np.uint64)       # n is a multiple of 64
    mode = _ffi.MODE_PE_LIBERAL
    want_code, want_counts = H.c_classify(mode, host["as1"], host["xs1"], host["as2"], host["xs2"], host_bits, _ffi.ABSENT)
    want_idx, want_off = H.c_compact(mode, want_code)
    del host
    blocks = shard.plan_blocks(n, world)
    assert [e - s for s, e in blocks] == [n // world] * world
    total = np.zeros(64, dtype=np.uint64)
    n_local_max = n // world + 1
    code = torch.empty(n_local_max + 64, dtype=torch.uint8, device=dev)
    idx = torch.empty(n_local_max, dtype=torch.int32, device=dev)
    off = torch.zeros(8, dtype=torch.int64, device=dev)
    counts = torch.zeros(64, dtype=torch.int64, device=dev)
    with _ffi.Context(0) as ctx:
        for g, (start, end) in enumerate(blocks):
            halo = 1 if start > 0 else 0
            lo = start - halo
            local = {k: v[lo:end].clone() for k, v in cols.items()}                 # what rank g would hold (16-byte aligned)
            lf = torch.zeros(((end - lo + 63) // 64) * 64, dtype=torch.uint8, device=dev)
            lf[:end - lo] = flags[lo:end]
            if halo:
                lf[0] = 0                                                           # the halo closes no unit here
            bits = _pack_bits_device(lf)
            ctx.classify_compact_dev(mode, local["as1"], local["xs1"], local["as2"], local["xs2"], bits, _ffi.ABSENT,
                                     code, idx, off, counts)
            torch.cuda.synchronize()
            o = off.cpu().numpy().astype(np.int64)
            total += counts.cpu().numpy().astype(np.uint64)
            got = idx[:int(o[7])].cpu().numpy().view(np.uint32).astype(np.int64) + lo
            # category bytes of the block's own records (its first one needed the halo's state)
            assert np.array_equal(code[halo:end - lo].cpu().numpy(), want_code[start:end]), g
            for b in range(7):
                seg = want_idx[int(want_off[b]):int(want_off[b + 1])]
                a, z = np.searchsorted(seg, start), np.searchsorted(seg, end)
                assert np.array_equal(got[o[b]:o[b + 1]], seg[a:z].astype(np.int64)), (g, b)
            del local, lf, bits
    assert np.array_equal(total, want_counts)
    assert int(total.sum()) == n // 2 - 1


def test_bench_starts_its_own_ranks():
    """The driver's command shape, `python bench.py --gpus N ...` from a plain start: bench.py must bring up N ranks
    itself and print exactly one JSON line.  Two ranks sharing this box's GPU over gloo (XM_BENCH_REHEARSAL=1; the
    RCCL build of the same path needs N GPUs), small batch."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, XM_BENCH_REHEARSAL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(H.REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                           "--pairs", "1000000", "--no-cpu-baseline", "--no-e2e"], env=env, capture_output=True, text=True,
                          timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert rec["verified_vs_oracle"] is True
    # the printed line rounds to 5-6 significant digits (the full record keeps everything)
    assert rec["value"] > 0 and abs(rec["value"] - 2 * 1_000_000 * 3 / (rec["ms_per_step"] * 3e-3)) / rec["value"] < 1e-4
    assert {"roofline", "roofline_step", "kernel_ms", "config"} <= set(rec)
