"""One rank of tests/test_multi_gpu.py (started as a fresh process per GPU, never a re-exec of a process that has touched
the GPU).  Takes its read block of ONE shared input (plan_blocks cut + halo), classifies it on its own device through the
C ABI, reduces category_counts through BOTH collectives -- torch.distributed's nccl backend and the library's own RCCL
communicator (xm_comm_init / xm_allreduce_counts) -- gathers the bin lists, and rank 0 compares everything with the
unsharded C oracle.  Exit status 0 = this rank saw everything agree.

    python tests/multi_gpu_rank.py <input.npz> <mode>      (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT in the environment)
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    path, mode = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("XM_TEST_BACKEND", "nccl")          # "gloo": the CPU rehearsal of this script (one GPU or none)
    import torch
    import torch.distributed as dist
    from xenomapper_amd import _ffi, shard
    phase = "start"
    try:
        data = np.load(path)
        cols = [data[k] for k in ("as1", "xs1", "as2", "xs2")]
        bits, n = data["unit_bits"], int(data["n"])
        on_gpu = backend == "nccl"
        dev_index = rank if on_gpu else 0
        phase = "init_process_group(%s)" % backend
        if on_gpu:
            torch.cuda.set_device(dev_index)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        phase = "classify"
        if os.environ.get("XM_TEST_ORACLE_CTX") == "1":           # CPU rehearsal: the oracle stands in for the device
            from tests.test_shard_gloo import OracleCtx
            ctx = OracleCtx()
        else:
            ctx = _ffi.Context(dev_index)
        start, end = shard.plan_blocks(n, world)[rank]
        lists, counts = shard.classify_block(ctx, mode, cols, bits, n, start, end, -2**31)
        dev = torch.device("cuda", dev_index) if on_gpu else torch.device("cpu")
        t_torch = torch.from_numpy(counts.astype(np.int64)).to(dev)
        t_lib = t_torch.clone()
        phase = "dist.all_reduce"
        shard.allreduce_counts(t_torch)
        if on_gpu:
            phase = "xm_comm_unique_id / broadcast"
            uid = [_ffi.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            phase = "xm_comm_init"
            ctx.comm_init(world, rank, uid[0])
            assert ctx.comm_size() == world
            phase = "xm_allreduce_counts"
            ctx.allreduce_counts(t_lib)
            torch.cuda.synchronize()
            assert torch.equal(t_lib, t_torch), "library RCCL all-reduce != torch.distributed all-reduce"
            ctx.comm_destroy()
        phase = "gather_bin_lists"
        all_lists = shard.gather_bin_lists(lists, dst=0)
        if rank == 0:
            phase = "compare with the unsharded oracle"
            from tests import helpers as H
            code, want_counts = H.c_classify(mode, *cols, bits, -2**31)
            want_idx, want_off = H.c_compact(mode, code)
            assert np.array_equal(t_torch.cpu().numpy().astype(np.uint64), want_counts)
            for b in range(7):
                assert np.array_equal(all_lists[b], want_idx[int(want_off[b]):int(want_off[b + 1])].astype(np.int64)), b
        phase = "barrier"
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:                                      # noqa: BLE001 -- the parent reads this line
        print("rank %d failed in phase %s: %s: %s" % (rank, phase, type(e).__name__, e), file=sys.stderr, flush=True)
        os._exit(1)
    print("rank %d ok" % rank, flush=True)


if __name__ == "__main__":
    main()
