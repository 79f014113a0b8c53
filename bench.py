#!/usr/bin/env python3
"""Benchmark of the classification hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic input that is already resident
in HBM: K1 classify (score columns -> category byte per record) + K2 compact (category_counts +
stable split of the pair indices into the six bins); every step adds its category_counts to the
job's running total on the device.  On N > 1 GPUs the job ends -- inside the timed region -- with the
one RCCL all-reduce of the final category_counts (the reference also only reports them at the end of
a run).  Workload = BASELINE.json configs[1]: 50 M paired-end 2x150 bp read pairs with AS/XS scores
per GPU (weak scaling: every rank holds its own 50 M-pair read block).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (classify): algorithmic bytes
(33 B per pair: 4 int32 scores x 2 mates in, 1 category byte out) / the kernel's mean
duration, measured with HIP events on the launch stream inside the timed region.  `cpu_baseline`
is the oracle's Python restatement of the reference's whole CPU path (parse + classify + write) on
SAM text, timed on one host core at N = 1 on a bounded sample.
"""
import argparse
import io
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0                 # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
BYTES_PER_PAIR_CLASSIFY = 33.0         # algorithmic: 2 mates x 4 int32 scores in + 1 category byte out
BYTES_PER_PAIR_COMPACT = 5.0           # 1 category byte in + one u32 pair index out


def cpu_baseline_python(sample_pairs=20000, passes=0, budget_s=12.0):
    """The oracle's pure-Python restatement of the reference CPU path on a SAM text twin of the
    workload (same score model), one core.  Repeats whole passes over the text until ~budget_s."""
    from oracle import xm_oracle as O
    from xenomapper_amd import synth
    t1, t2, _ = synth.sam_text_pair(n_pairs=sample_pairs, seed=2002, profile="bowtie2", paired=True,
                                    read_len=150)
    done = 0
    t0 = time.perf_counter()
    n_pass = 0
    while True:
        s1, s2 = io.StringIO(t1), io.StringIO(t2)
        outs = [io.StringIO() for _ in range(6)]
        O.write_headers(s1, s2, outs)
        res = O.run_paired_end(O.read_pairs(s1, s2), outs)
        done += len(res.units)
        n_pass += 1
        el = time.perf_counter() - t0
        if (passes and n_pass >= passes) or (not passes and el >= budget_s):
            break
    return {"value": done / el, "unit": "read-pairs/s", "cores": 1, "kind": "port",
            "sample": "%d passes over a %d-pair 2x150 bp SAM text twin (seed 2002): parse + classify + "
                      "write six bins, pure-Python restatement of the reference loop" % (n_pass, sample_pairs),
            "seconds": round(el, 2)}


def cpu_baseline_c(cols_host, n_pairs):
    """The C oracle on score columns already parsed (classify + compact only), one core."""
    from tests import helpers as H
    t0 = time.perf_counter()
    code, _ = H.c_classify(1, cols_host["as1"], cols_host["xs1"], cols_host["as2"], cols_host["xs2"],
                           cols_host["unit_bits"], -2**31)
    H.c_compact(1, code)
    el = time.perf_counter() - t0
    return {"value": n_pairs / el, "unit": "read-pairs/s", "cores": 1, "kind": "port",
            "sample": "%d pairs of the same columns, scalar C restatement, classify + compact only "
                      "(no SAM parsing, no output)" % n_pairs, "seconds": round(el, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=50_000_000, help="read pairs per GPU")
    ap.add_argument("--mode", choices=("liberal", "conservative"), default=None)
    ap.add_argument("--workload", choices=("cfg2", "cfg3", "cfg5"), default="cfg2",
                    help="BASELINE.json configs[1] (default, the quoted metric), [2] --cigar_scores path, [4] HISAT ZS + conservative")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from xenomapper_amd import _ffi, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the classifier has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # Rehearsal on a one-GPU box (not for reported numbers): XM_BENCH_REHEARSAL=1 lets every rank use cuda:0
    # and swaps RCCL for gloo, so that the N > 1 code path can be exercised where only one GPU is visible.
    rehearsal = os.environ.get("XM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n_pairs = args.pairs
    n = 2 * n_pairs
    if args.mode is None:
        args.mode = "conservative" if args.workload == "cfg5" else "liberal"
    mode = _ffi.MODE_PE_LIBERAL if args.mode == "liberal" else _ffi.MODE_PE_CONSERVATIVE
    ctx = _ffi.Context(local_rank)
    bytes_per_pair = BYTES_PER_PAIR_CLASSIFY
    cig = None
    if args.workload == "cfg3":
        # no AS tag: AS is synthesised in-kernel from NM + CIGAR (CSR); XS mostly absent
        cig = [synth.cigar_columns_torch(n, seed=3003 + 2 * rank, device=dev),
               synth.cigar_columns_torch(n, seed=3004 + 2 * rank, device=dev, mapped_p=0.3)]
        g = torch.Generator(device=dev)
        g.manual_seed(3005 + rank)
        xs1 = torch.where(torch.rand(n, generator=g, device=dev) < 0.95, torch.full((n,), _ffi.ABSENT, dtype=torch.int32, device=dev),
                          -torch.randint(0, 40, (n,), generator=g, device=dev, dtype=torch.int32))
        cols = {"xs1": xs1, "xs2": torch.full((n,), _ffi.ABSENT, dtype=torch.int32, device=dev),
                "unit_bits": torch.from_numpy(synth.interleaved_unit_bits(n).view(np.int64)).to(dev)}
        k_bar = (cig[0]["cig_oplen"].numel() + cig[1]["cig_oplen"].numel()) / (2.0 * n)
        bytes_per_pair = 4 * (12 + 4 * k_bar) + 1            # per record and species NM + XS + offset + ops
        range_flag = torch.zeros(4, dtype=torch.int32, device=dev)
    else:
        cols = synth.score_columns_torch(n_pairs, seed=(2002 if args.workload == "cfg2" else 5005) + rank, device=dev,
                                         profile="hisat" if args.workload == "cfg5" else "bowtie2")   # own read block per rank
    code = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    off = torch.zeros(8, dtype=torch.int64, device=dev)
    counts = torch.zeros(64, dtype=torch.int64, device=dev)
    job_counts = torch.zeros(64, dtype=torch.int64, device=dev)
    floor_min = _ffi.ABSENT                                                       # min_score = -inf

    def step():
        if cig is not None:
            ctx.classify_cigar_dev(mode, cig[0]["nm"], cig[0]["cig_off"], cig[0]["cig_oplen"], cols["xs1"],
                                   cig[1]["nm"], cig[1]["cig_off"], cig[1]["cig_oplen"], cols["xs2"], cols["unit_bits"],
                                   floor_min, code, range_flag=range_flag)
        else:
            ctx.classify_dev(mode, cols["as1"], cols["xs1"], cols["as2"], cols["xs2"], cols["unit_bits"],
                             floor_min, code)
        ctx.compact_dev(mode, code[:n], idx, off, counts)
        job_counts.add_(counts)                                  # category_counts of the job so far

    def finish():
        if world > 1:
            dist.all_reduce(job_counts, op=dist.ReduceOp.SUM)    # RCCL over xGMI: 64 x int64, once per job

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    finish()                                                      # warms the RCCL communicator up as well
    fence()
    job_counts.zero_()
    fence()
    ctx.timing_select(["classify"])          # the timed region brackets only the kernel the roofline is about
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish()
    fence()
    elapsed = time.perf_counter() - t0
    timing = ctx.timing_read()
    job_total = int(job_counts.sum().item())
    # diagnostic pass outside the timed region: every kernel bracketed (the event pairs cost stream time)
    ctx.timing_select(None)
    ctx.timing_reset()
    for _ in range(min(args.steps, 20)):
        step()
    fence()
    timing_all = ctx.timing_read()
    ctx.timing_enable(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # parity of what was just timed (rank-local): against the C oracle on the same columns
    verified = None
    host_cols = None
    if not args.no_verify:
        from tests import helpers as H
        host_cols = {k: v.cpu().numpy() for k, v in cols.items()}
        host_cols["unit_bits"] = host_cols["unit_bits"].view(np.uint64)
        if cig is not None:
            for f, key in ((0, "as1"), (1, "as2")):
                hc = {k: v.cpu().numpy() for k, v in cig[f].items()}
                host_cols[key], bad = H.c_cigar_scores(hc["nm"], hc["cig_off"].view(np.uint32), hc["cig_oplen"].view(np.uint32))
                assert bad == 0
        want_code, want_counts = H.c_classify(mode, host_cols["as1"], host_cols["xs1"], host_cols["as2"],
                                              host_cols["xs2"], host_cols["unit_bits"], floor_min)
        want_idx, want_off = H.c_compact(mode, want_code)
        ok = bool((code[:n].cpu().numpy() == want_code).all())
        ok &= bool((off.cpu().numpy().astype(np.uint64) == want_off).all())
        ok &= bool((idx[:int(want_off[7])].cpu().numpy().view(np.uint32) == want_idx).all())
        ok &= bool((counts.cpu().numpy().astype(np.uint64) == want_counts).all())
        ok &= job_total == world * n_pairs * args.steps
        verified = ok
        flag = torch.tensor([1 if ok else 0], device=dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        verified = bool(flag.item())

    if rank == 0:
        k_cls = timing["classify"]
        cls_ms = k_cls["ms"] / max(1, k_cls["launches"])
        achieved = bytes_per_pair * n_pairs / (cls_ms * 1e-3) / 1e9
        kernels = {k: round(v["ms"] / max(1, v["launches"]), 5) for k, v in timing_all.items() if v["launches"]}
        traffic = None
        pmc_file = os.path.join(REPO, "profiles", {"cfg2": "pmc_classify.json", "cfg3": "pmc_classify_cigar.json"}.get(
            args.workload, "none"))
        if os.path.exists(pmc_file) and n_pairs == 50_000_000:
            try:
                traffic = json.load(open(pmc_file)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "read-pairs/sec classified",
            "value": world * n_pairs * args.steps / elapsed,
            "unit": "read-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic" + (" (REHEARSAL: ranks share one GPU, gloo)" if rehearsal else ""),
            "config": {"workload": {"cfg2": "configs[1]: %d paired-end 2x150 bp read pairs per GPU, AS/XS present, %s "
                                           "pair rule, min_score=-inf, score columns resident in HBM",
                                    "cfg3": "configs[2]: %d paired-end pairs per GPU on the --cigar_scores path (no AS/XS; NM + "
                                            "CIGAR as CSR, AS synthesised in the classify kernel), %s pair rule, columns resident in HBM",
                                    "cfg5": "configs[4]: %d paired-end pairs per GPU, HISAT-style scores with ZS as second-best "
                                            "(AS=0/ZS=0 present), %s pair rule, columns resident in HBM"}[args.workload]
                                   % (n_pairs, args.mode),
                       "pairs_per_gpu": n_pairs, "records_per_species_per_gpu": n,
                       "sharding": "read-block per GPU, no halo exchange" + (", one RCCL all-reduce of the final category_counts" if world > 1 else "")},
            "roofline": {"bound": "hbm",
                         "kernel": "classify_cigar_kernel<paired>" if cig is not None else "classify_kernel<int32, paired>",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "algorithmic_bytes_per_pair": bytes_per_pair, "kernel_ms": cls_ms},
            "kernel_ms": kernels, "kernel_ms_note": "all kernels bracketed in a separate pass after the timed region",
            "verified_vs_oracle": verified,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_python()
            if host_cols is not None and cig is None:
                m = min(n_pairs, 5_000_000)
                sub = {k: np.ascontiguousarray(v[:2 * m]) for k, v in host_cols.items() if k != "unit_bits"}
                sub["unit_bits"] = np.ascontiguousarray(host_cols["unit_bits"][:(2 * m + 63) // 64])
                line["cpu_baseline_c"] = cpu_baseline_c(sub, m)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
