#!/usr/bin/env python3
"""Benchmark of the classification hot path on MI355X (BASELINE.json metric).

    python bench.py                              # 1 GPU, configs[1]
    python bench.py --gpus N --steps K --warmup W   # N > 1: starts its own N ranks (one per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W            # ... or is started as a rank

One step = one pass of the hot path over one batch of synthetic input that is already resident in HBM, through
ONE C-ABI call (xm_classify_compact_dev): K1 classify + count (score columns -> category byte per record,
category_counts, per-granule bin counts), K2b scan, K2c scatter (stable split of the pair indices into the six
bins); every step leaves its category_counts in its own 64-word slot on the device, and the job ends -- inside the timed
region -- by summing the slots into the job's category_counts and, on N > 1 GPUs, with the one RCCL all-reduce of them
(the reference also only reports them at the end of a run).  Workload = BASELINE.json configs[1]: 50 M paired-end 2x150 bp read pairs with
AS/XS scores per GPU (weak scaling: every rank holds its own 50 M-pair read block).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (classify): algorithmic bytes (33 B per pair:
4 int32 scores x 2 mates in, 1 category byte out) / the kernel's mean duration, measured with HIP
events on the launch stream inside the timed region.  `roofline_step` is SURVEY 8d's whole-step figure: 38 B per pair
/ the sum of the kernels' mean durations.  `cpu_baseline` is the oracle's Python restatement of the reference's whole
CPU path (parse + classify + write) on SAM text, timed on one host core at N = 1 on a bounded sample.  `e2e` holds
the two transfer-inclusive rates SURVEY 8d asks for beside it (never `value`): host columns in / bin lists out over
PCIe, and SAM text in / six SAM files out.
"""
import argparse
import contextlib
import io
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0                 # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
BYTES_PER_PAIR_CLASSIFY = 33.0         # algorithmic: 2 mates x 4 int32 scores in + 1 category byte out
BYTES_PER_PAIR_COMPACT = 5.0           # 1 category byte in + one u32 pair index out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=50_000_000, help="read pairs per GPU")
    ap.add_argument("--strong-total", type=int, default=0, metavar="PAIRS",
                    help="strong scaling instead: PAIRS read pairs in all, PAIRS / N per GPU (SURVEY 8d: 400000000)")
    ap.add_argument("--mode", choices=("liberal", "conservative"), default=None)
    ap.add_argument("--workload", choices=("cfg2", "cfg3", "cfg5", "f64", "se"), default="cfg2",
                    help="BASELINE.json configs[1] (default, the quoted metric), [2] --cigar_scores path, [4] HISAT ZS + "
                         "conservative; f64 = configs[1] through the binary64 kernel; se = the single-end loop (units = reads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the transfer-inclusive side measurements")
    return ap.parse_args()


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on stdout when its first communicator comes up; stdout is for the one JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def spawn_ranks(args):
    """`python bench.py --gpus N` from a plain start: N fresh rank processes under torch.distributed.run, before this
    process has touched the GPU (it never does); rank 0's JSON line comes through on stdout."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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
    return {"value": done / el, "unit": "read-pairs/s", "cores": 1, "cpu_model": cpu_model(),
            "host_cpus_visible": os.cpu_count(), "kind": "port",
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
    return {"value": n_pairs / el, "unit": "read-pairs/s", "cores": 1, "cpu_model": cpu_model(), "kind": "port",
            "sample": "%d pairs of the same columns, scalar C restatement, classify + compact only "
                      "(no SAM parsing, no output)" % n_pairs, "seconds": round(el, 3)}


def e2e_h2d_inclusive(ctx, mode, host_cols, pairs, reps=3):
    """Host columns in, bin lists + counts out, through the host-buffer C ABI (pageable NumPy memory): H2D of the four
    score columns and the unit mask, the fused pass, D2H of the index lists.  Never `value`."""
    import numpy as np
    n = 2 * pairs
    cols = [np.ascontiguousarray(host_cols[k][:n]) for k in ("as1", "xs1", "as2", "xs2")]
    bits = np.ascontiguousarray(host_cols["unit_bits"][:(n + 63) // 64])
    best, units = None, 0
    for _ in range(reps):
        t0 = time.perf_counter()
        _, idx, off, _ = ctx.classify_compact(mode, *cols, bits, -2**31, want_code=False)
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
        units = int(off[7])
    moved = 16 * n + n // 8 + 4 * units
    return {"read_pairs_per_s": units / best, "GBps_over_pcie": moved / best / 1e9, "pairs": pairs,
            "bytes_over_pcie": moved, "seconds": round(best, 4),
            "what": "xm_classify_compact on pageable host arrays: H2D 32.25 B/pair, fused pass, D2H 4 B/pair (bin lists)"}


def e2e_sam_text(pairs=4_000_000, to_files=True):
    """SAM text in -> six SAM files out through the file fast path (C++ stripper -> GPU -> C++ writer); outputs on
    tmpfs (`to_files`) or /dev/null (what is left is parser-bound).  Never `value`."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import bench_e2e
    shm = os.path.isdir("/dev/shm")
    r = bench_e2e.run(pairs=pairs, threads=0, mode="liberal", workdir="/dev/shm" if shm else "/tmp",
                      out_dir=("/dev/shm" if shm else "/tmp") if to_files else None)
    return {"read_pairs_per_s": r["value"], "input_GBps": r["input_GBps"], "pairs": r["units"], "threads": r["threads"],
            "seconds": round(r["seconds"], 4), "outputs": r["outputs"], "output_bytes": r["output_bytes"],
            "what": "two SAM text files (2x150 bp, tiled 50 k-pair twin) -> stripper -> H2D -> fused pass -> D2H -> six SAM outputs"}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    from xenomapper_amd import _ffi, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
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
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if rehearsal else "nccl"
        with stdout_to_stderr():
            if rehearsal:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def allreduce(t, op):
        if world == 1:
            return t
        if backend == "gloo":                      # rehearsal only: gloo reduces host tensors
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)
        return t

    # how many ranks the collective backend really connected (the driver checks this against --gpus)
    one = torch.ones(1, dtype=torch.int64, device=dev)
    with stdout_to_stderr():
        n_ranks_seen = int(allreduce(one, dist.ReduceOp.SUM).item()) if world > 1 else 1

    n_pairs = args.pairs
    if args.strong_total:
        n_pairs = (args.strong_total // world + 3) // 4 * 4          # a read block per GPU, whole 64-record words of the unit mask
    n = 2 * n_pairs
    if args.mode is None:
        args.mode = "conservative" if args.workload == "cfg5" else "liberal"
    if args.workload == "se":
        mode = _ffi.MODE_SE
    else:
        mode = _ffi.MODE_PE_LIBERAL if args.mode == "liberal" else _ffi.MODE_PE_CONSERVATIVE
    units_per_step = n if args.workload == "se" else n_pairs            # se: a unit is a read
    ctx = _ffi.Context(local_rank)
    bytes_per_unit = BYTES_PER_PAIR_CLASSIFY
    bytes_per_unit_compact = BYTES_PER_PAIR_COMPACT
    dtype = "int32"
    cig = cigp = None
    cigar_csr = os.environ.get("XM_BENCH_CIGAR_CSR") == "1"     # A/B only: the CSR-column kernel (K1c + stand-alone histogram)
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
        if cigar_csr:
            bytes_per_unit = 4 * (12 + 4 * k_bar) + 1        # per record and species NM + XS + offset + ops
        else:
            # packed CIGAR columns: NM + XS + one count byte per record and species, one tile base per 256 records, the ops
            cigp = [synth.cigar_pack_torch(c) for c in cig]
            bytes_per_unit = 4 * (9 + 1.0 / 64 + 4 * k_bar) + 1
        range_flag = torch.zeros(4, dtype=torch.int32, device=dev)
    else:
        cols = synth.score_columns_torch(n_pairs, seed=(5005 if args.workload == "cfg5" else 2002) + rank, device=dev,
                                         profile="hisat" if args.workload == "cfg5" else "bowtie2")   # own read block per rank
        if args.workload == "se":
            cols["unit_bits"] = torch.full(((n + 63) // 64,), -1, dtype=torch.int64, device=dev)      # every record is a unit
            bytes_per_unit, bytes_per_unit_compact = 17.0, 5.0                   # 4 int32 in + 1 byte out; 1 byte in + 1 index out
        if args.workload == "f64":
            for k in ("as1", "xs1", "as2", "xs2"):
                c = cols[k]
                cols[k] = torch.where(c == _ffi.ABSENT, torch.full((), float("-inf"), dtype=torch.float64, device=dev),
                                      c.to(torch.float64))
            bytes_per_unit = 65.0                                                # 2 mates x 4 binary64 scores in + 1 byte out
            dtype = "f64"
    code = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev)      # compact category stream: a nibble per record
    # XM_BENCH_CATEGORY_BYTES=1 (A/B): the per-record output of the step is the category byte (fwd*8+rev) instead
    category_bytes = os.environ.get("XM_BENCH_CATEGORY_BYTES") == "1" or (args.workload == "cfg3" and cigar_csr)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    off = torch.zeros(8, dtype=torch.int64, device=dev)
    n_slots = max(args.steps, args.warmup, 1)
    step_counts = torch.zeros((n_slots, 64), dtype=torch.int64, device=dev)       # category_counts of every step of the job
    counts = step_counts[0]
    job_counts = torch.zeros(64, dtype=torch.int64, device=dev)
    step_no = [0]
    floor_min = float("-inf") if args.workload == "f64" else _ffi.ABSENT          # min_score = -inf

    unfused = os.environ.get("XM_BENCH_UNFUSED") == "1"          # A/B only: two C-ABI calls (classify, then compact with its own histogram)

    def step_unfused():
        counts = step_counts[step_no[0] % n_slots]
        step_no[0] += 1
        if cig is not None:
            ctx.classify_cigar_dev(mode, cig[0]["nm"], cig[0]["cig_off"], cig[0]["cig_oplen"], cols["xs1"],
                                   cig[1]["nm"], cig[1]["cig_off"], cig[1]["cig_oplen"], cols["xs2"], cols["unit_bits"],
                                   floor_min, code, range_flag=range_flag)
        else:
            ctx.classify_dev(mode, cols["as1"], cols["xs1"], cols["as2"], cols["xs2"], cols["unit_bits"], floor_min, code)
        ctx.compact_dev(mode, code[:n], idx, off, counts)

    def step():
        if unfused:
            return step_unfused()
        counts = step_counts[step_no[0] % n_slots]
        step_no[0] += 1
        if cigp is not None:
            ctx.classify_compact_cigar_packed_dev(mode, cigp[0]["nm"], cigp[0]["cig_cnt"], cigp[0]["cig_tile"], cigp[0]["cig_oplen"],
                                                  cols["xs1"], cigp[1]["nm"], cigp[1]["cig_cnt"], cigp[1]["cig_tile"],
                                                  cigp[1]["cig_oplen"], cols["xs2"], cols["unit_bits"], floor_min,
                                                  code if category_bytes else None, idx, off, counts,
                                                  bins4=None if category_bytes else bins4, range_flag=range_flag)
        elif cig is not None:
            ctx.classify_compact_cigar_dev(mode, cig[0]["nm"], cig[0]["cig_off"], cig[0]["cig_oplen"], cols["xs1"],
                                           cig[1]["nm"], cig[1]["cig_off"], cig[1]["cig_oplen"], cols["xs2"],
                                           cols["unit_bits"], floor_min, code, idx, off, counts, range_flag=range_flag)
        else:
            ctx.classify_compact_dev(mode, cols["as1"], cols["xs1"], cols["as2"], cols["xs2"], cols["unit_bits"], floor_min,
                                     code if category_bytes else None, idx, off, counts,
                                     bins4=None if category_bytes else bins4)

    def finish(n_steps):
        torch.sum(step_counts[:n_steps], dim=0, out=job_counts)  # category_counts of the job: the sum over its steps
        allreduce(job_counts, dist.ReduceOp.SUM)                 # RCCL over xGMI: 64 x int64, once per job

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    with stdout_to_stderr():
        finish(max(args.warmup, 1))                               # warms the RCCL communicator up as well
    fence()
    step_counts.zero_()
    job_counts.zero_()
    step_no[0] = 0
    fence()
    ctx.timing_select(["classify"])          # the timed region brackets only the kernel the roofline is about
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    timing = ctx.timing_read()
    job_total = int(job_counts.sum().item())
    job_final = job_counts.clone()
    # diagnostic pass outside the timed region: every kernel bracketed (the event pairs cost stream time)
    ctx.timing_select(None)
    ctx.timing_reset()
    for _ in range(min(args.steps, 20)):
        step()
    fence()
    timing_all = ctx.timing_read()
    ctx.timing_enable(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    elapsed = float(allreduce(t, dist.ReduceOp.MAX).item()) if world > 1 else elapsed

    # parity of what was just timed (rank-local): against the C oracle on the same columns
    verified = None
    host_cols = None
    if not args.no_verify:
        from tests import helpers as H
        if rank == 0:
            H.c_oracle()                       # (re)builds oracle/libxm_oracle.so if stale: once, not in N ranks at a time
        if world > 1:
            dist.barrier()
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
        if not category_bytes and not unfused:
            # the timed step left the compact stream: check it, then ask for the category bytes as well (same kernels,
            # both outputs) so that they are checked too
            want_bins = np.full(n, 7, dtype=np.uint8)
            for b in range(7):
                want_bins[want_idx[int(want_off[b]):int(want_off[b + 1])]] = b
            ok_bins = bool((_ffi.unpack_bins4(bins4.cpu().numpy(), n) == want_bins).all())
            if cigp is not None:
                ctx.classify_compact_cigar_packed_dev(mode, cigp[0]["nm"], cigp[0]["cig_cnt"], cigp[0]["cig_tile"], cigp[0]["cig_oplen"],
                                                      cols["xs1"], cigp[1]["nm"], cigp[1]["cig_cnt"], cigp[1]["cig_tile"],
                                                      cigp[1]["cig_oplen"], cols["xs2"], cols["unit_bits"], floor_min,
                                                      code, idx, off, counts, bins4=bins4, range_flag=range_flag)
            else:
                ctx.classify_compact_dev(mode, cols["as1"], cols["xs1"], cols["as2"], cols["xs2"], cols["unit_bits"], floor_min,
                                         code, idx, off, counts, bins4=bins4)
            torch.cuda.synchronize()
        else:
            ok_bins = True
        ok = ok_bins and bool((code[:n].cpu().numpy() == want_code).all())
        ok &= bool((off.cpu().numpy().astype(np.uint64) == want_off).all())
        ok &= bool((idx[:int(want_off[7])].cpu().numpy().view(np.uint32) == want_idx).all())
        ok &= bool((counts.cpu().numpy().astype(np.uint64) == want_counts).all())
        ok &= job_total == world * units_per_step * args.steps
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
        verified = bool(allreduce(flag, dist.ReduceOp.MIN).item()) if world > 1 else ok

    if rank == 0:
        unit_name = "read" if args.workload == "se" else "read-pair"
        k_cls = timing["classify"]
        cls_ms = k_cls["ms"] / max(1, k_cls["launches"])
        achieved = bytes_per_unit * units_per_step / (cls_ms * 1e-3) / 1e9
        kernels = {k: round(v["ms"] / max(1, v["launches"]), 5) for k, v in timing_all.items() if v["launches"]}
        sum_ms = sum(kernels.values())
        step_bytes = (bytes_per_unit + bytes_per_unit_compact) * units_per_step
        step_achieved = step_bytes / (sum_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        pmc_name = {"cfg2": "pmc_classify.json", "cfg3": "pmc_classify_cigar.json"}.get(args.workload)
        if pmc_name and n_pairs == 50_000_000:
            try:
                with open(os.path.join(REPO, "profiles", pmc_name)) as fh:
                    traffic = json.load(fh).get("hbm_bytes_per_launch")
                traffic_source = "profiles/%s (rocprofv3 --pmc passes of this command, replayed; not measured in this run)" % pmc_name
            except Exception:
                traffic = None
        step_traffic = None
        if args.workload == "cfg2" and n_pairs == 50_000_000 and not unfused and not category_bytes:
            try:
                with open(os.path.join(REPO, "profiles", "pmc_step.json")) as fh:
                    step_traffic = json.load(fh).get("hbm_bytes_per_step")
            except Exception:
                step_traffic = None
        ms_per_step = 1e3 * elapsed / args.steps
        line = {
            "metric": "reads/sec classified" if args.workload == "se" else "read-pairs/sec classified",
            "value": world * units_per_step * args.steps / elapsed,
            "unit": unit_name + "s/s",
            "n_gpus": world, "n_ranks_seen": n_ranks_seen, "collective_backend": backend,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if args.strong_total else "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic" + (" (REHEARSAL: ranks share one GPU, gloo)" if rehearsal else ""),
            "config": {"workload": {"cfg2": "configs[1]: %d paired-end 2x150 bp read pairs per GPU, AS/XS present, %s "
                                           "pair rule, min_score=-inf, score columns resident in HBM",
                                    "cfg3": "configs[2]: %d paired-end pairs per GPU on the --cigar_scores path (no AS/XS; NM + "
                                            "CIGAR ops in " + ("CSR" if cigar_csr else "packed") + " columns, AS synthesised in the classify kernel), %s pair rule, columns resident in HBM",
                                    "cfg5": "configs[4]: %d paired-end pairs per GPU, HISAT-style scores with ZS as second-best "
                                            "(AS=0/ZS=0 present), %s pair rule, columns resident in HBM",
                                    "f64": "configs[1] columns as binary64 (the reference's own arithmetic; used for non-integral "
                                           "scores): %d paired-end pairs per GPU, %s pair rule, columns resident in HBM",
                                    "se": "configs[0]'s kernel shape at size: single-end loop over 2 x %d reads per GPU, every "
                                          "record a unit (%s ignored), columns resident in HBM"}[args.workload]
                                   % (n_pairs, args.mode),
                       "job": ("%d read pairs in all, one %d-pair read block per GPU%s" % (world * n_pairs, n_pairs,
                               " = BASELINE.json configs[3] (400 M pairs sharded across 8 GPUs, RCCL count all-reduce)"
                               if world * n_pairs == 400_000_000 and world == 8 and args.workload == "cfg2" else "")),
                       "pairs_per_gpu": n_pairs, "records_per_species_per_gpu": n,
                       "step": ("A/B: xm_classify_dev + xm_compact_dev (classify, hist, scan, scatter)" if unfused else
                                "one xm_classify_compact%s_dev call: classify+count, scan, scatter" % ("_cigar_packed" if cigp is not None else "_cigar" if cig is not None else "")),
                       "category_per_record": ("category byte (fwd*8+rev, 1 B per record)" if category_bytes or unfused else
                                               "compact stream bins4: the output bin as a nibble per record = 1 B per pair (SURVEY 8d's "
                                               "algorithmic figure); the category byte is produced on request and checked outside the timed region"),
                       "sharding": "read-block per GPU, no halo exchange" + (", one RCCL all-reduce of the final category_counts" if world > 1 else "")},
            "roofline": {"bound": "hbm",
                         "kernel": "classify_cigp_kernel<paired, counts, bins4>" if cigp is not None else "classify_cigar_kernel<paired>" if cig is not None else "classify_kernel<%s, %s, counts>" % (dtype, "single" if args.workload == "se" else "paired"),
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_unit": bytes_per_unit, "kernel_ms": cls_ms},
            "roofline_step": {"bound": "hbm", "what": "SURVEY 8d: (classify + compact) algorithmic bytes per %s / sum of the kernels' mean durations" % unit_name,
                              "algorithmic_bytes_per_unit": bytes_per_unit + bytes_per_unit_compact,
                              "sum_kernel_ms": round(sum_ms, 5), "achieved": step_achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": step_achieved / HBM_PEAK_GBPS,
                              "traffic": step_traffic,
                              "traffic_source": "profiles/pmc_step.json (replayed)" if step_traffic else None,
                              "frac_by_ms_per_step": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            "kernel_ms": kernels, "kernel_ms_note": "all kernels bracketed in a separate pass after the timed region (scan = its two launches)",
            "verified_vs_oracle": verified,
            "xm_allreduce_counts": None,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_python()
            if host_cols is not None and args.workload in ("cfg2", "cfg5"):
                m = min(n_pairs, 5_000_000)
                sub = {k: np.ascontiguousarray(v[:2 * m]) for k, v in host_cols.items() if k != "unit_bits"}
                sub["unit_bits"] = np.ascontiguousarray(host_cols["unit_bits"][:(2 * m + 63) // 64])
                line["cpu_baseline_c"] = cpu_baseline_c(sub, m)
        if world == 1 and not args.no_e2e and args.workload == "cfg2":
            e2e = {"note": "transfer-inclusive rates beside `value` (which is HBM-resident); measured after the timed region"}
            try:
                hc = host_cols if host_cols is not None else {k: v.cpu().numpy() for k, v in cols.items()}
                if hc["unit_bits"].dtype != np.uint64:
                    hc["unit_bits"] = hc["unit_bits"].view(np.uint64)
                e2e["h2d_inclusive"] = e2e_h2d_inclusive(ctx, mode, hc, min(n_pairs, 25_000_000))
            except Exception as e:                               # noqa: BLE001
                e2e["h2d_inclusive"] = {"error": "%s: %s" % (type(e).__name__, e)}
            for key, to_files in (("sam_text", True), ("sam_text_devnull", False)):
                try:
                    e2e[key] = e2e_sam_text(to_files=to_files)
                except Exception as e:                           # noqa: BLE001
                    e2e[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            line["e2e"] = e2e
    # The same reduction through the library's own RCCL communicator (xm_allreduce_counts, the C ABI's collective), on
    # every rank, after everything else and under a watchdog: the job total above came from torch.distributed, so
    # whatever happens here the line is printed -- with the outcome (or the error, or "timed out") in it.
    if rank != 0:
        line = None

    def give_up():
        if rank == 0:
            line["xm_allreduce_counts"] = {"error": "no answer within 120 s"}
            print(json.dumps(line), flush=True)
        os._exit(0)

    if not rehearsal:
        import threading
        watchdog = threading.Timer(120.0, give_up)
        watchdog.daemon = True
        watchdog.start()
        outcome = None
        try:
            with stdout_to_stderr():
                uid = [None]
                if rank == 0:
                    try:
                        uid = [_ffi.comm_unique_id()]
                    except Exception as e:                       # noqa: BLE001 -- every rank must still reach the broadcast
                        uid = ["%s: %s" % (type(e).__name__, e)]
                if world > 1:
                    dist.broadcast_object_list(uid, src=0)
                if not isinstance(uid[0], bytes):
                    raise RuntimeError("xm_comm_unique_id on rank 0: %s" % uid[0])
                ctx.comm_init(world, rank, uid[0])
                mine = counts * args.steps                       # every step saw the same block: this rank's job total
                ctx.allreduce_counts(mine)
                torch.cuda.synchronize()
                outcome = {"ranks": ctx.comm_size(), "matches_torch_distributed": bool(torch.equal(mine, job_final))}
                ctx.comm_destroy()
        except Exception as e:                                   # noqa: BLE001 -- reported, not raised
            outcome = {"error": "%s: %s" % (type(e).__name__, e)}
        watchdog.cancel()
        if rank == 0:
            line["xm_allreduce_counts"] = outcome
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
