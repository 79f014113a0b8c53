"""The N > 1 path on real devices: lights up by itself on a box with two or more GPUs (skips cleanly on one).

N ranks (fresh child processes, one device each), ONE shared seeded input cut with shard.plan_blocks / take_block --
halo at every cut, a run of equal QNAMEs straddling every cut (two overlapping pairs across it, ref
xenomapper.py:402-405, :451-452) -- category_counts reduced through both collectives (torch nccl = RCCL, and the C
ABI's xm_allreduce_counts), lists gathered, compared with the unsharded C oracle.  tests/multi_gpu_rank.py is the rank.
The same script is rehearsed on CPU (gloo, the oracle standing in for the device) so that its logic is exercised here.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers as H
from tests.test_shard_gloo import _free_port, make_input


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()          # does not initialise the GPU
    except Exception:                             # noqa: BLE001
        return 0


def _shared_input(path, n, world):
    from xenomapper_amd import shard
    cols, bits = make_input(n, 4242)
    flags = shard.unpack_unit_bits(bits, n).copy()
    for start, _ in shard.plan_blocks(n, world)[1:]:
        flags[start - 1:start + 2] = 1            # records start-2 .. start+1 share a name: units at start-1, start, start+1
    bits = H.synth.pack_unit_bits(flags)
    np.savez(path, as1=cols[0], xs1=cols[1], as2=cols[2], xs2=cols[3], unit_bits=bits, n=n)


def _run_ranks(world, path, mode, extra_env, timeout=300):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1", **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(H.REPO, "tests", "multi_gpu_rank.py"), path, str(mode)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    out = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=timeout)
            out.append((p.returncode, o, e))
    except subprocess.TimeoutExpired:
        for p in procs:                            # the exact processes started above
            p.kill()
        raise AssertionError("ranks did not finish within %d s: %s" % (timeout, [p.poll() for p in procs]))
    assert all(rc == 0 for rc, _, _ in out), "\n".join("rc %s\n%s\n%s" % (rc, o[-2000:], e[-2000:]) for rc, o, e in out)
    # (gloo / RCCL print banners on stdout: the rank's own line is the last one)
    assert sorted(o.strip().splitlines()[-1] for _, o, _ in out) == sorted("rank %d ok" % r for r in range(world))


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs torch.cuda.device_count() >= 2 (one process per GPU over RCCL)")
@pytest.mark.parametrize("mode", [1, 2])
def test_sharded_input_on_real_devices_both_collectives(tmp_path, mode):
    world = min(_gpu_count(), 4)
    path = str(tmp_path / "input.npz")
    _shared_input(path, 1_000_003, world)
    _run_ranks(world, path, mode, {})


@pytest.mark.parametrize("world", [2, 3])
def test_rank_script_rehearsal_on_cpu(tmp_path, world):
    """The rank script's own logic (block + halo, counts, gather, oracle comparison) with gloo and the oracle as the
    device: what the GPU test above runs, minus RCCL and the kernels."""
    path = str(tmp_path / "input.npz")
    _shared_input(path, 20_011, world)
    _run_ranks(world, path, 1, {"XM_TEST_BACKEND": "gloo", "XM_TEST_ORACLE_CTX": "1"}, timeout=120)
