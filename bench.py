#!/usr/bin/env python3
"""Benchmark of the classification hot path on MI355X (BASELINE.json metric).

    python bench.py                              # 1 GPU: configs[1] timed; configs[2] and [4] reported under `workloads`
    python bench.py --workload cfg3|cfg5|se|f64  # another workload as the timed one
    python bench.py --gpus N --steps K --warmup W   # N > 1: starts its own N ranks (one per GPU, RCCL)
    python bench.py --gpus N --sharded-input     # configs[3] as ONE input cut into N read blocks with halo (strong scaling)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W            # ... or is started as a rank

One step = one pass of the hot path over one batch of synthetic input that is already resident in HBM, through
ONE C-ABI call (xm_classify_compact_dev; --cigar_scores: xm_classify_compact_cigar_packed_dev): K1 classify + count
(columns -> category per record as the compact stream, category_counts, per-granule and per-part bin counts), K2b scan,
K2c scatter (stable split of the pair indices into the six bins); every step leaves its category_counts in its own
64-word slot on the device, and the job ends -- inside the timed region -- by summing the slots into the job's
category_counts and, on N > 1 GPUs, with the one RCCL all-reduce of them (the reference also only reports them at the
end of a run).  Workload = BASELINE.json configs[1]: 50 M paired-end 2x150 bp read pairs with AS/XS scores per GPU
(weak scaling: every rank holds its own 50 M-pair read block).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (classify): algorithmic bytes (33 B per pair:
4 int32 scores x 2 mates in, 1 category byte out; packed CIGAR columns: 4 (9 + 1/64 + 4 k) + 1) / the
kernel's mean duration, measured with HIP events on the launch stream inside the timed region; `traffic` = HBM bytes
per launch from the committed rocprofv3 --pmc passes, printed only while the kernel sources still hash to what they
were measured on (tools/kernel_hash.py).  `roofline_step` is SURVEY 8d's whole-step figure: 38 B per pair / the sum of
the kernels' mean durations (and / ms_per_step).  `workloads` (default run): configs[2] and configs[4] timed the same
way in the same process.  `cpu_baseline` is the oracle's Python restatement of the reference's whole CPU path (parse +
classify + write) on SAM text, timed on one host core at N = 1 on a bounded sample.  `e2e` holds the transfer-inclusive
rates SURVEY 8d asks for beside it (never `value`) with their own ceilings: host columns in / bin lists out over PCIe
(pageable and page-locked buffers; the link's measured rate), SAM text in / six SAM files out (the host side's memory,
write and parse ceilings).  A hang in the library's own RCCL collective (run after everything else, under a watchdog)
still delivers the line -- with the error in it -- and exits with status 3.
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
EXTRA_WORKLOADS = (("configs[2]", "cfg3"), ("configs[4]", "cfg5"))     # timed after the default run, reported under `workloads`


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
    ap.add_argument("--sharded-input", action="store_true",
                    help="configs[3] as ONE input: --strong-total pairs (default 400000000) cut into read blocks with the one-record "
                         "halo (shard.plan_blocks / take_block rule), one block per GPU; strong scaling")
    ap.add_argument("--singletons-pct", type=float, default=0.0,
                    help="cfg2/cfg5: this percentage of the reads are singletons (no mate), so the mates are not strictly "
                         "interleaved and the scatter takes its general path")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="default run only: skip timing configs[2] and configs[4] after the timed region (the `workloads` key)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the transfer-inclusive side measurements")
    return ap.parse_args()


_REAL_STDOUT = [None]      # while stdout_to_stderr() is active: a duplicate of the real stdout (the watchdog writes there)


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on stdout when its first communicator comes up; stdout is for the one JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    _REAL_STDOUT[0] = saved
    try:
        yield
    finally:
        sys.stdout.flush()
        _REAL_STDOUT[0] = None
        os.dup2(saved, 1)
        os.close(saved)


WATCHDOG_EXIT = 3


def run_with_watchdog(fn, seconds, line, what, rank=0):
    """Run fn() -- something that may block for ever inside a collective -- under a timer.  When the timer fires, rank 0
    writes the JSON line (with the failure recorded under `what`) to the REAL stdout, whatever fd 1 currently points
    at, and the process exits with WATCHDOG_EXIT: a hung collective must never reach the driver as rc 0 without a line."""
    import threading

    def give_up():
        if rank == 0 and line is not None:
            line[what] = {"error": "no answer within %g s" % seconds, "rank": rank}
            data = (json.dumps(line) + "\n").encode()
            fd = _REAL_STDOUT[0] if _REAL_STDOUT[0] is not None else 1
            try:
                sys.stdout.flush()
            except Exception:                                         # noqa: BLE001
                pass
            while data:
                data = data[os.write(fd, data):]
        sys.stderr.write("bench.py: rank %d gave up on %s after %g s\n" % (rank, what, seconds))
        sys.stderr.flush()
        os._exit(WATCHDOG_EXIT)

    timer = threading.Timer(seconds, give_up)
    timer.daemon = True
    timer.start()
    try:
        return fn()
    finally:
        timer.cancel()


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
            "sample_short": "%d x %d-pair SAM text twin, parse+classify+write, %.0f s" % (n_pass, sample_pairs, el),
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
    idx_buf = [None]

    def timed():
        best, units = None, 0
        for _ in range(reps):
            t0 = time.perf_counter()
            _, idx, off, _ = ctx.classify_compact(mode, *cols, bits, -2**31, want_code=False, idx_out=idx_buf[0])
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
            units = int(off[7])
        return best, units
    pageable, units = timed()
    moved = 16 * n + n // 8 + 4 * units
    out = {"read_pairs_per_s": units / pageable, "GBps_over_pcie": moved / pageable / 1e9, "pairs": pairs,
           "bytes_over_pcie": moved, "seconds": round(pageable, 4),
           "what": "xm_classify_compact on pageable host arrays: H2D 32.25 B/pair, fused pass, D2H 4 B/pair (bin lists)"}
    # the same with the buffers page-locked once (xm_host_register: long-lived buffers, e.g. a parser's columns and a
    # writer's index list): the copies are direct DMA, the rate is the link's
    try:
        idx_buf[0] = np.zeros(n, dtype=np.uint32)
        t0 = time.perf_counter()
        with ctx.registered(*(cols + [bits, idx_buf[0]])):           # unlocks what it locked, also when locking stops half way
            reg = time.perf_counter() - t0
            pinned, units = timed()
        out["registered_buffers"] = {"read_pairs_per_s": units / pinned, "GBps_over_pcie": moved / pinned / 1e9,
                                     "seconds": round(pinned, 4), "register_seconds_once": round(reg, 4),
                                     "what": "the same call with the five input arrays and the index output page-locked "
                                             "beforehand (xm_host_register)"}
    except Exception as e:                                       # noqa: BLE001
        out["registered_buffers"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def pcie_ceiling(dev, mb=512):
    """What the link itself delivers on this box: one pinned-host <-> device copy of `mb` MB each way (torch, best of 3)."""
    import torch
    host = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    devt = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    out = {}
    for name, dst, src in (("h2d_pinned_GBps", devt, host), ("d2h_pinned_GBps", host, devt)):
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dst.copy_(src, non_blocking=False)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        out[name] = (mb << 20) / best / 1e9
    pageable = torch.empty(mb << 20, dtype=torch.uint8)
    pageable.fill_(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    devt.copy_(pageable)
    torch.cuda.synchronize()
    out["h2d_pageable_GBps"] = (mb << 20) / (time.perf_counter() - t0) / 1e9
    out["what"] = "single %d MB copies through torch (hipMemcpy): the ceiling of e2e.h2d_inclusive on this box" % mb
    return out


def e2e_sam_text(pairs=4_000_000, to_files=True, gpu_strip=True):
    """SAM text in -> six SAM files out through the file fast path (text -> page-locked staging -> GPU stripper -> fused
    pass -> C++ writer; gpu_strip=False: the C++ host stripper instead); outputs on tmpfs (`to_files`) or /dev/null.
    Never `value`."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import bench_e2e
    shm = os.path.isdir("/dev/shm")
    before = os.environ.get("XENOMAPPER_GPU_STRIP")
    os.environ["XENOMAPPER_GPU_STRIP"] = "1" if gpu_strip else "0"
    try:
        r = bench_e2e.run(pairs=pairs, threads=0, mode="liberal", workdir="/dev/shm" if shm else "/tmp",
                          out_dir=("/dev/shm" if shm else "/tmp") if to_files else None)
    finally:
        if before is None:
            del os.environ["XENOMAPPER_GPU_STRIP"]
        else:
            os.environ["XENOMAPPER_GPU_STRIP"] = before
    strip = None
    if gpu_strip and r["phases"].get("strip_upload_ms"):
        # the stripper's own roofline is the link: text bytes / device time from the first piece's upload to the last
        strip = {"bound": "pcie", "upload_GBps": round(r["input_bytes"] / (r["phases"]["strip_upload_ms"] / 1e3) / 1e9, 2),
                 "kernels_ms_total": round(r["phases"].get("strip_kernels_ms", 0.0), 2),
                 "kernels_GBps_of_text": round(r["input_bytes"] / (max(r["phases"].get("strip_kernels_ms", 0.0), 1e-9) / 1e3) / 1e9, 1),
                 "what": "SAM text read over the link by the first strip kernel, piece by piece while the host threads read the next piece "
                         "into page-locked memory (compare e2e.pcie_ceiling.h2d_pinned_GBps); the other strip kernels run behind each window"}
    return {"read_pairs_per_s": r["value"], "input_GBps": r["input_GBps"], "pairs": r["units"], "threads": r["threads"], "phases": r["phases"],
            "strip": strip,
            "seconds": round(r["seconds"], 4), "outputs": r["outputs"], "output_bytes": r["output_bytes"],
            "text_bytes_in_per_pair": round(r["input_bytes"] / max(r["units"], 1), 1),
            "text_bytes_out_per_pair": round(r["output_bytes"] / max(r["units"], 1), 1),
            "output_GBps": r["output_bytes"] / r["seconds"] / 1e9,
            "stripper": "gpu" if gpu_strip else "host",
            "what": ("two SAM text files (2x150 bp, tiled 50 k-pair twin) -> pread into page-locked memory -> the TEXT read over the link by the "
                     "strip kernels -> fused pass on the columns in HBM -> the six outputs gathered on the GPU (xm_strip_fetch_bins), one stream per "
                     "window back over the link -> six SAM outputs") if gpu_strip
                    else "two SAM text files (2x150 bp, tiled 50 k-pair twin) -> C++ host stripper -> H2D of columns -> fused pass "
                         "-> D2H -> six SAM outputs"}


class Workload(object):
    """One BASELINE.json workload on one GPU: synthetic columns resident in HBM, the output buffers, step() = one
    C-ABI call over the whole batch, verify() = the last step's outputs against the C oracle on the same columns."""

    DESCR = {
        "cfg2": "configs[1]: %d paired-end 2x150 bp read pairs per GPU, AS/XS present, %s pair rule, min_score=-inf, score "
                "columns resident in HBM",
        "cfg3": "configs[2]: %d paired-end pairs per GPU on the --cigar_scores path (no AS/XS; NM + CIGAR ops in %s columns, "
                "AS synthesised in the classify kernel), %s pair rule, columns resident in HBM",
        "cfg5": "configs[4]: %d paired-end pairs per GPU, HISAT-style scores with ZS as second-best (AS=0/ZS=0 present), %s "
                "pair rule, columns resident in HBM",
        "f64": "configs[1] columns as binary64 (the reference's own arithmetic; used for non-integral scores): %d paired-end "
               "pairs per GPU, %s pair rule, columns resident in HBM",
        "se": "configs[0]'s kernel shape at size: single-end loop over 2 x %d reads per GPU, every record a unit (%s ignored), "
              "columns resident in HBM",
    }

    def __init__(self, name, ctx, dev, n_pairs, rank, mode_name=None, n_slots=1, singletons_pct=0.0, shard=None, form=None):
        import numpy as np
        import torch
        from xenomapper_amd import _ffi, synth
        self.name, self.ctx, self.dev = name, ctx, dev
        self.mode_name = mode_name or ("conservative" if name == "cfg5" else "liberal")
        self.mode = _ffi.MODE_SE if name == "se" else (_ffi.MODE_PE_LIBERAL if self.mode_name == "liberal" else _ffi.MODE_PE_CONSERVATIVE)
        self.bytes_classify, self.bytes_compact, self.dtype = BYTES_PER_PAIR_CLASSIFY, BYTES_PER_PAIR_COMPACT, "int32"
        self.cig = self.cigp = self.range_flag = None
        self.cigar_csr = os.environ.get("XM_BENCH_CIGAR_CSR") == "1"     # A/B only: the CSR-column kernel (K1c + stand-alone histogram)
        self.unfused = os.environ.get("XM_BENCH_UNFUSED") == "1"        # A/B only: xm_classify_dev + xm_compact_dev
        self.shard = shard
        # the six-list form (xm_classify_place*_dev): the sharded-input entry (large read blocks go through it in chunks), and XM_BENCH_PLACE=1 for A/B runs
        self.place = (form == "place" or os.environ.get("XM_BENCH_PLACE") == "1") and name in ("cfg2", "cfg5", "se", "f64")
        # the segmented-lists form (xm_classify_runs*_dev, one launch): its own workload entry, and XM_BENCH_RUNS=1 for A/B runs
        self.runs = (form == "runs" or os.environ.get("XM_BENCH_RUNS") == "1") and name in ("cfg2", "cfg5", "se", "f64") and not self.place
        self.layout = "strictly interleaved mates"
        n = 2 * n_pairs
        units = n_pairs
        if shard is not None:
            cols, n, units = self._sharded_block(shard, rank)
        elif name == "cfg3":
            # no AS tag: AS is synthesised in-kernel from NM + CIGAR; XS mostly absent
            self.cig = [synth.cigar_columns_torch(n, seed=3003 + 2 * rank, device=dev),
                        synth.cigar_columns_torch(n, seed=3004 + 2 * rank, device=dev, mapped_p=0.3)]
            g = torch.Generator(device=dev)
            g.manual_seed(3005 + rank)
            xs1 = torch.where(torch.rand(n, generator=g, device=dev) < 0.95, torch.full((n,), _ffi.ABSENT, dtype=torch.int32, device=dev),
                              -torch.randint(0, 40, (n,), generator=g, device=dev, dtype=torch.int32))
            cols = {"xs1": xs1, "xs2": torch.full((n,), _ffi.ABSENT, dtype=torch.int32, device=dev),
                    "unit_bits": torch.from_numpy(synth.interleaved_unit_bits(n).view(np.int64)).to(dev)}
            self.k_bar = (self.cig[0]["cig_oplen"].numel() + self.cig[1]["cig_oplen"].numel()) / (2.0 * n)
            if self.cigar_csr:
                self.bytes_classify = 4 * (12 + 4 * self.k_bar) + 1      # per record and species NM + XS + CSR offset + ops
            else:
                # packed CIGAR columns: NM + XS + one count byte per record and species, one tile base per 256 records, the ops
                self.cigp = [synth.cigar_pack_torch(c) for c in self.cig]
                self.bytes_classify = 4 * (9 + 1.0 / 64 + 4 * self.k_bar) + 1
            self.range_flag = torch.zeros(4, dtype=torch.int32, device=dev)
        else:
            cols = synth.score_columns_torch(n_pairs, seed=(5005 if name == "cfg5" else 2002) + rank, device=dev,
                                             profile="hisat" if name == "cfg5" else "bowtie2")      # own read block per rank
            if name == "se":
                cols["unit_bits"] = torch.full(((n + 63) // 64,), -1, dtype=torch.int64, device=dev)     # every record is a unit
                self.bytes_classify, self.bytes_compact, units = 17.0, 5.0, n     # 4 int32 in + 1 byte out; 1 byte in + 1 index out
            elif singletons_pct > 0.0:
                flags = singleton_unit_flags(n, singletons_pct / 100.0, seed=77 + rank)
                units = int(flags.sum())
                cols["unit_bits"] = torch.from_numpy(synth.pack_unit_bits(flags).view(np.int64)).to(dev)
                self.layout = "%.3g %% of the reads are singletons: mates not strictly interleaved" % singletons_pct
            if name == "f64":
                for k in ("as1", "xs1", "as2", "xs2"):
                    c = cols[k]
                    cols[k] = torch.where(c == _ffi.ABSENT, torch.full((), float("-inf"), dtype=torch.float64, device=dev),
                                          c.to(torch.float64))
                self.bytes_classify, self.dtype = 65.0, "f64"            # 2 mates x 4 binary64 scores in + 1 byte out
        self.cols, self.n, self.n_pairs, self.units_per_step = cols, n, n_pairs, units
        self.floor_min = float("-inf") if name == "f64" else _ffi.ABSENT             # min_score = -inf
        # XM_BENCH_CATEGORY_BYTES=1 (A/B): the per-record output of the step is the category byte (fwd*8+rev) instead
        self.category_bytes = os.environ.get("XM_BENCH_CATEGORY_BYTES") == "1" or (name == "cfg3" and self.cigar_csr) or self.unfused
        self.code = torch.empty(n + 16, dtype=torch.uint8, device=dev)
        self.bins4 = torch.empty(_ffi.bins4_bytes(n), dtype=torch.uint8, device=dev)  # compact category stream: a nibble per record
        self.idx = torch.empty(n, dtype=torch.int32, device=dev)
        self.off = torch.zeros(8, dtype=torch.int64, device=dev)
        if self.place:
            self.lists = [torch.empty(units + 64, dtype=torch.int32, device=dev) for _ in range(7 if name == "f64" else 6)]
            self.n_out = torch.zeros(8, dtype=torch.int64, device=dev)
        if self.runs:
            # its OWN algorithmic bytes: the scores in, 2 B per unit of run entries + 16 B of counts per 2048 records out; no
            # second pass (SURVEY 8d's 38 B/pair is the flat contract's figure and stays with the flat step)
            g = _ffi.runs_granules(n)
            self.runs16 = torch.empty(g * _ffi.RUNS_GRAN, dtype=torch.int16, device=dev)
            self.gran_counts = torch.empty(g * 8, dtype=torch.int16, device=dev)
            self.n_out = torch.zeros(8, dtype=torch.int64, device=dev)
            per_unit_in = self.bytes_classify - 1.0
            self.bytes_classify = per_unit_in + 2.0 + 16.0 * g / max(units, 1)
            self.bytes_compact = 0.0
        self.n_slots = max(n_slots, 1)
        self.step_counts = torch.zeros((self.n_slots, 64), dtype=torch.int64, device=dev)   # category_counts of every step of the job
        self.step_no = 0
        self.counts = self.step_counts[0]

    # configs[3] as one input: PARTS seeded parts (seed 4004 + part) put end to end, mates at records (2k-1, 2k) so that
    # every 64-record-aligned cut separates a pair; this rank's read block + the halo record in front of it
    PARTS = 8

    def _sharded_block(self, shard, rank):
        import torch
        from xenomapper_amd import shard as sh, synth
        total_pairs, world = shard
        part_pairs = total_pairs // self.PARTS
        n_all = 2 * part_pairs * self.PARTS
        start, end = sh.plan_blocks(n_all, world)[rank]
        halo = 1 if start > 0 else 0
        lo = start - halo
        part_rec = 2 * part_pairs
        pieces = {k: [] for k in ("as1", "xs1", "as2", "xs2")}
        for p in range(lo // part_rec, (end - 1) // part_rec + 1):
            part = synth.score_columns_torch(part_pairs, seed=4004 + p, device=self.dev)
            a, b = max(lo, p * part_rec) - p * part_rec, min(end, (p + 1) * part_rec) - p * part_rec
            for k in pieces:
                pieces[k].append(part[k][a:b].clone())
            del part
        cols = {k: torch.cat(v) if len(v) > 1 else v[0] for k, v in pieces.items()}
        gidx = torch.arange(lo, end, device=self.dev, dtype=torch.int64)
        flags = ((gidx % 2 == 0) & (gidx >= 2)).to(torch.uint8)
        if halo:
            flags[0] = 0                                   # the halo's own unit belongs to the previous block (shard.take_block)
        units = int(flags.sum().item())
        pad = (-flags.numel()) % 64
        if pad:
            flags = torch.cat([flags, torch.zeros(pad, dtype=torch.uint8, device=self.dev)])
        w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=self.dev)
        cols["unit_bits"] = (flags.view(-1, 8) * w).sum(dim=1, dtype=torch.uint8).view(torch.int64)
        self.layout = "mates at records (2k-1, 2k) of the whole input: every cut separates a pair (halo record in front of the block)"
        self.block = (start, end, halo)
        return cols, end - lo, units

    def short(self):
        return {"cfg2": "configs[1]: %d PE 2x150 pairs/GPU, AS/XS, %s, HBM-resident", "cfg3": "configs[2]: %d PE pairs/GPU, --cigar_scores packed columns, %s",
                "cfg5": "configs[4]: %d PE pairs/GPU, HISAT ZS, %s", "f64": "configs[1] as binary64: %d PE pairs/GPU, %s",
                "se": "single-end loop: 2 x %d reads/GPU (%s ignored)"}[self.name] % (self.n_pairs, self.mode_name)

    def step_short(self):
        if self.runs:
            return "xm_classify_runs%s_dev: classify+count+sort per granule, ONE launch (segmented lists)" % ("_f64" if self.dtype == "f64" else "")
        if self.place:
            return "xm_classify_place%s_dev: classify+count, scan, scatter into six lists" % ("_f64" if self.dtype == "f64" else "")
        if self.unfused:
            return "A/B: xm_classify_dev + xm_compact_dev"
        return "xm_classify_compact%s_dev: classify+count, scan, scatter (3 launches)" % (
            "_cigar_packed" if self.cigp is not None else "_cigar" if self.cig is not None else "_f64" if self.dtype == "f64" else "")

    def describe(self):
        if self.name == "cfg3":
            return self.DESCR["cfg3"] % (self.n_pairs, "CSR" if self.cigar_csr else "packed", self.mode_name)
        return self.DESCR[self.name] % (self.n_pairs, self.mode_name)

    def kernel_name(self):
        if self.runs:
            return "classify_runs_kernel<%s, %s>" % (self.dtype, "single" if self.name == "se" else "paired")
        if self.cigp is not None:
            return "classify_cigp_kernel<paired, counts, bins4>"
        if self.cig is not None:
            return "classify_cigar_kernel<paired>"
        return "classify_kernel<%s, %s, counts>" % (self.dtype, "single" if self.name == "se" else "paired")

    def call_name(self):
        if self.runs:
            return "one xm_classify_runs%s_dev call: classify, count, sort by bin inside the granule (one launch + a 1-workgroup launch for the counts)" % (
                "_f64" if self.dtype == "f64" else "")
        if self.place:
            return "one xm_classify_place%s_dev call: classify+count, scan, scatter into six lists" % ("_f64" if self.dtype == "f64" else "")
        if self.unfused:
            return "A/B: xm_classify_dev + xm_compact_dev (classify, hist, scan, scatter)"
        return "one xm_classify_compact%s_dev call: classify+count, scan, scatter" % (
            "_cigar_packed" if self.cigp is not None else "_cigar" if self.cig is not None else "_f64" if self.dtype == "f64" else "")

    def _packed_call(self, code, bins4, counts):
        c, p = self.cols, self.cigp
        self.ctx.classify_compact_cigar_packed_dev(self.mode, p[0]["nm"], p[0]["cig_cnt"], p[0]["cig_tile"], p[0]["cig_oplen"], c["xs1"],
                                                   p[1]["nm"], p[1]["cig_cnt"], p[1]["cig_tile"], p[1]["cig_oplen"], c["xs2"],
                                                   c["unit_bits"], self.floor_min, code, self.idx, self.off, counts,
                                                   bins4=bins4, range_flag=self.range_flag)

    def step(self):
        c, ctx, mode = self.cols, self.ctx, self.mode
        counts = self.counts = self.step_counts[self.step_no % self.n_slots]
        self.step_no += 1
        code = self.code if self.category_bytes else None
        bins4 = None if self.category_bytes else self.bins4
        if self.runs:
            ctx.classify_runs_dev(mode, c["as1"], c["xs1"], c["as2"], c["xs2"], c["unit_bits"], self.floor_min, self.runs16,
                                  self.gran_counts, self.n_out, counts)
        elif self.place:
            ctx.classify_place_dev(mode, c["as1"], c["xs1"], c["as2"], c["xs2"], c["unit_bits"], self.floor_min, self.lists[:6],
                                   self.n_out, counts, code_out=code, bins4=bins4,
                                   list_state6=self.lists[6] if len(self.lists) > 6 else None)
        elif self.unfused:
            if self.cig is not None:
                g = self.cig
                ctx.classify_cigar_dev(mode, g[0]["nm"], g[0]["cig_off"], g[0]["cig_oplen"], c["xs1"], g[1]["nm"], g[1]["cig_off"],
                                       g[1]["cig_oplen"], c["xs2"], c["unit_bits"], self.floor_min, self.code, range_flag=self.range_flag)
            else:
                ctx.classify_dev(mode, c["as1"], c["xs1"], c["as2"], c["xs2"], c["unit_bits"], self.floor_min, self.code)
            ctx.compact_dev(mode, self.code[:self.n], self.idx, self.off, counts)
        elif self.cigp is not None:
            self._packed_call(code, bins4, counts)
        elif self.cig is not None:
            g = self.cig
            ctx.classify_compact_cigar_dev(mode, g[0]["nm"], g[0]["cig_off"], g[0]["cig_oplen"], c["xs1"], g[1]["nm"], g[1]["cig_off"],
                                           g[1]["cig_oplen"], c["xs2"], c["unit_bits"], self.floor_min, self.code, self.idx, self.off,
                                           counts, range_flag=self.range_flag)
        else:
            ctx.classify_compact_dev(mode, c["as1"], c["xs1"], c["as2"], c["xs2"], c["unit_bits"], self.floor_min, code,
                                     self.idx, self.off, counts, bins4=bins4)

    def reset_counts(self):
        self.step_counts.zero_()
        self.step_no = 0

    def host_columns(self):
        import numpy as np
        hc = {k: v.cpu().numpy() for k, v in self.cols.items()}
        hc["unit_bits"] = hc["unit_bits"].view(np.uint64)
        return hc

    def verify(self):
        """What the last step left on the device against the C oracle on the same columns (rank-local)."""
        import numpy as np
        import torch
        from tests import helpers as H
        from xenomapper_amd import _ffi
        n, mode = self.n, self.mode
        hc = self.host_columns()
        if self.cig is not None:
            for f, key in ((0, "as1"), (1, "as2")):
                g = {k: v.cpu().numpy() for k, v in self.cig[f].items()}
                hc[key], bad = H.c_cigar_scores(g["nm"], g["cig_off"].view(np.uint32), g["cig_oplen"].view(np.uint32))
                assert bad == 0
        want_code, want_counts = H.c_classify(mode, hc["as1"], hc["xs1"], hc["as2"], hc["xs2"], hc["unit_bits"], self.floor_min)
        want_idx, want_off = H.c_compact(mode, want_code)
        ok = True
        if self.runs:
            n_out = self.n_out.cpu().numpy().astype(np.uint64)
            runs16 = self.runs16.cpu().numpy().view(np.uint16)
            gcnt = self.gran_counts.cpu().numpy().view(np.uint16)
            for b in range(7):
                want = want_idx[int(want_off[b]):int(want_off[b + 1])]
                ok &= int(n_out[b]) == want.shape[0]
                got = _ffi.runs_expand(n, runs16, gcnt, b)
                ok &= got.shape[0] == want.shape[0] and bool((got == want).all())
            ok &= int(n_out[7]) == int(want_off[7]) == self.units_per_step
            ok &= bool((self.counts.cpu().numpy().astype(np.uint64) == want_counts).all())
            self.host_cols = hc
            return ok
        if self.place:
            n_out = self.n_out.cpu().numpy().astype(np.uint64)
            for b in range(len(self.lists)):
                want = want_idx[int(want_off[b]):int(want_off[b + 1])]
                ok &= int(n_out[b]) == want.shape[0]
                ok &= bool((self.lists[b][:want.shape[0]].cpu().numpy().view(np.uint32) == want).all())
            ok &= int(n_out[7]) == int(want_off[7]) == self.units_per_step
            ok &= bool((self.counts.cpu().numpy().astype(np.uint64) == want_counts).all())
            self.host_cols = hc
            return ok
        if not self.category_bytes:
            # the timed step left the compact stream: check it, then ask for the category bytes as well (same kernels,
            # both outputs) so that they are checked too
            want_bins = np.full(n, 7, dtype=np.uint8)
            for b in range(7):
                want_bins[want_idx[int(want_off[b]):int(want_off[b + 1])]] = b
            ok = bool((_ffi.unpack_bins4(self.bins4.cpu().numpy(), n) == want_bins).all())
            c = self.cols
            if self.cigp is not None:
                self._packed_call(self.code, self.bins4, self.counts)
            else:
                self.ctx.classify_compact_dev(mode, c["as1"], c["xs1"], c["as2"], c["xs2"], c["unit_bits"], self.floor_min,
                                              self.code, self.idx, self.off, self.counts, bins4=self.bins4)
            torch.cuda.synchronize()
        ok &= bool((self.code[:n].cpu().numpy() == want_code).all())
        ok &= bool((self.off.cpu().numpy().astype(np.uint64) == want_off).all())
        ok &= bool((self.idx[:int(want_off[7])].cpu().numpy().view(np.uint32) == want_idx).all())
        ok &= bool((self.counts.cpu().numpy().astype(np.uint64) == want_counts).all())
        ok &= int(want_off[7]) == self.units_per_step
        if self.range_flag is not None:
            ok &= int(self.range_flag[0].item()) == 0
        self.host_cols = hc
        return ok


def singleton_unit_flags(n_records, p_single, seed):
    """Unit flags of a read stream in which a fraction p_single of the reads has no mate: pairs occupy two adjacent
    records (the second closes the unit), singletons one (closes none), so the parity of the mates flips at every
    singleton."""
    import numpy as np
    rng = np.random.default_rng(seed)
    items = int(n_records / (2.0 - p_single)) + 1024
    single = rng.random(items) < p_single                          # per read template
    length = np.where(single, 1, 2)
    begin = np.cumsum(length) - length
    keep = (begin + length <= n_records)
    second = begin[keep & ~single] + 1
    flags = np.zeros(n_records, dtype=np.uint8)
    flags[second] = 1
    return flags


def time_steps(ctx, wl, steps, warmup, fence, finish=None):
    """`warmup` untimed steps, then exactly `steps` steps (+ finish()) between two fences, HIP events around the classify
    kernel only; then a diagnostic pass with every kernel bracketed.  -> (elapsed s, classify timing, all-kernel timing)"""
    for _ in range(warmup):
        wl.step()
    if finish is not None:
        with stdout_to_stderr():
            finish(max(warmup, 1))                                # warms the RCCL communicator up as well
    fence()
    wl.reset_counts()
    fence()
    ctx.timing_select(["classify"])          # the timed region brackets only the kernel the roofline is about
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    if finish is not None:
        finish(steps)
    fence()
    elapsed = time.perf_counter() - t0
    timing = ctx.timing_read()
    # diagnostic pass outside the timed region: every kernel bracketed (the event pairs cost stream time)
    ctx.timing_select(None)
    ctx.timing_reset()
    for _ in range(min(steps, 20)):
        wl.step()
    fence()
    timing_all = ctx.timing_read()
    ctx.timing_enable(False)
    return elapsed, timing, timing_all


def median_step_ms(wl, steps=20):
    """Median duration of one step: `steps` further steps, each between two events on the stream the library launches
    on (torch's current stream), after the timed region ("report median"; `ms_per_step` stays elapsed / K)."""
    import torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for k in range(steps):
        wl.step()
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(steps))
    return ms[steps // 2]


def stream_ceiling_gbps(ctx, wl, reps=11):
    """The box's own ceiling for the classify kernel's access pattern: xm_stream_probe_dev reads the workload's four
    score columns (16 B per record) and writes 1/2 B per record, no arithmetic; median of `reps`; GB/s of the bytes it
    moves.  None for workloads without four int32 score columns."""
    import torch
    c = wl.cols
    if wl.dtype != "int32" or any(k not in c for k in ("as1", "xs1", "as2", "xs2")):
        return None
    n = wl.n - wl.n % 4
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ctx.stream_probe_dev(c["as1"][:n], c["xs1"][:n], c["as2"][:n], c["xs2"][:n], wl.bins4)
    ev[0].record()
    for k in range(reps):
        ctx.stream_probe_dev(c["as1"][:n], c["xs1"][:n], c["as2"][:n], c["xs2"][:n], wl.bins4)
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(reps))[reps // 2]
    return 16.5 * n / (ms * 1e-3) / 1e9


def copy_ceiling_gbps(dev, bytes_moved, reps=11):
    """What a plain streaming copy reaches on THIS box with the classify kernel's footprint: a device-to-device copy
    that reads bytes_moved / 2 and writes as much (torch -> hipMemcpyDtoD), median of `reps`; GB/s of read + written bytes."""
    import torch
    half = int(bytes_moved // 2) & ~15
    src = torch.empty(half, dtype=torch.uint8, device=dev)
    dst = torch.empty(half, dtype=torch.uint8, device=dev)
    src.fill_(3)
    dst.copy_(src)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        dst.copy_(src)
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(reps))[reps // 2]
    del src, dst
    return 2.0 * half / (ms * 1e-3) / 1e9


def _r(x, digits=5):
    """Round to `digits` significant digits (the printed line is short; the full record keeps everything)."""
    if x is None or isinstance(x, (bool, int, str)):
        return x
    return float("%.*g" % (digits, x))


def compact_line(full):
    """The ONE JSON line rank 0 prints: numbers only, well under 2 000 characters, so that a record which keeps the
    line's tail still holds every workload.  The sentences (what each number is) live in profiles/README.md; the full
    record of the run is written next to it (`full_record`)."""
    roof, step = full["roofline"], full["roofline_step"]
    line = {k: full[k] for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline",
                                 "dtype", "data")}
    line["value"] = _r(full["value"], 6)
    line["ms_per_step"] = _r(full["ms_per_step"])
    line["ms_per_step_median"] = _r(full.get("ms_per_step_median"))
    line["config"] = {"workload": full["config"]["workload_short"], "pairs_per_gpu": full["config"]["pairs_per_gpu"],
                      "step": full["config"]["step_short"], "sharding": full["config"]["sharding_short"]}
    line["roofline"] = {"bound": "hbm", "kernel": roof["kernel"].split("<")[0], "achieved": _r(roof["achieved"]), "peak": roof["peak"],
                        "unit": "GB/s", "frac": _r(roof["frac"], 4), "traffic": _r(roof["traffic"]), "kernel_ms": _r(roof["kernel_ms"]),
                        "bytes_per_unit": _r(roof["algorithmic_bytes_per_unit"]), "copy_ceiling_GBps": _r(roof.get("copy_ceiling_GBps"), 4),
                        "memcpy_d2d_GBps": _r(roof.get("memcpy_d2d_GBps"), 4), "frac_of_copy": _r(roof.get("frac_of_copy"), 4)}
    line["roofline_step"] = {"frac": _r(step["frac"], 4), "frac_by_ms_per_step": _r(step["frac_by_ms_per_step"], 4),
                             "sum_kernel_ms": _r(step["sum_kernel_ms"]), "bytes_per_unit": _r(step["algorithmic_bytes_per_unit"]),
                             "traffic": _r(step["traffic"])}
    line["kernel_ms"] = {k: _r(v, 4) for k, v in full["kernel_ms"].items()}
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb["sample_short"]}
    line["verified_vs_oracle"] = full["verified_vs_oracle"]
    line["n_ranks_seen"] = full["n_ranks_seen"]
    x = full.get("xm_allreduce_counts")
    if x is not None:
        line["xm_allreduce_counts"] = ({"ranks": x.get("ranks"), "ok": x.get("matches_torch_distributed")} if "error" not in x
                                       else {"error": str(x["error"])[:120]})
    w = {}
    for key, short in (("configs[2]", "cfg3"), ("configs[4]", "cfg5")):
        e = (full.get("workloads") or {}).get(key)
        if e is not None:      # [ms_per_step, roofline.frac (classify kernel), frac by ms_per_step (whole step), verified]
            w[short] = ([_r(e["ms_per_step"]), _r(e["roofline"]["frac"], 4), _r(e["roofline_step"]["frac_by_ms_per_step"], 4),
                         e["verified_vs_oracle"]] if "error" not in e else [None, None, None, str(e["error"])[:80]])
    e = (full.get("workloads") or {}).get("sharded_input")
    if e is not None:          # [ms_per_step, read-pairs/s of the whole job, verified]: configs[3]'s input cut into N blocks with halo
        w["sharded"] = ([_r(e["ms_per_step"]), _r(e["value"], 6), e["verified_vs_oracle"]] if "error" not in e
                        else [None, None, str(e["error"])[:80]])
    e = (full.get("workloads") or {}).get("runs")
    if e is not None:          # [ms_per_step, median, frac by ms_per_step on its OWN bytes (34 B/pair), verified]: segmented lists, one launch
        w["runs"] = ([_r(e["ms_per_step"]), _r(e.get("ms_per_step_median")), _r(e["roofline_step"]["frac_by_ms_per_step"], 4),
                      e["verified_vs_oracle"]] if "error" not in e else [None, None, None, str(e["error"])[:80]])
    e = (full.get("workloads") or {}).get("graph")
    if e is not None:          # [ms_per_step of the same call replayed from a HIP graph, verified]
        w["graph"] = [_r(e["ms_per_step"]), e["verified_vs_oracle"]] if "error" not in e else [None, str(e["error"])[:80]]
    if w:
        line["workloads"] = w
    e2e = full.get("e2e")
    if e2e:                    # [G read-pairs/s host columns -> lists over PCIe (page-locked), M read-pairs/s SAM text -> six SAM files (GPU stripper),
                               #  M pairs/s BAM -> outputs (BAM front end on the GPU), M pairs/s SAM text -> files with the host stripper,
                               #  M pairs/s BAM -> outputs with the host decoder, GB/s of inflated bytes of the GPU BGZF decoder]
        def rate(d, *path):
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        line["e2e"] = [_r((rate(e2e, "h2d_inclusive", "registered_buffers", "read_pairs_per_s") or 0) / 1e9, 4),
                       _r((rate(e2e, "sam_text", "read_pairs_per_s") or 0) / 1e6, 4),
                       _r((rate(e2e, "bam", "read_pairs_per_s") or 0) / 1e6, 4),
                       _r((rate(e2e, "sam_text_host_stripper", "read_pairs_per_s") or 0) / 1e6, 4),
                       _r((rate(e2e, "bam_host_decoder", "read_pairs_per_s") or 0) / 1e6, 4),
                       _r(rate(e2e, "inflate", "value") or 0, 4)]
    line["full_record"] = full.get("full_record")
    return line


def rooflines(wl, elapsed, steps, timing, timing_all, unit_name):
    """The `roofline` (dominant kernel) and `roofline_step` (whole step) objects + per-kernel means of one workload."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import kernel_hash
    k_cls = timing["classify"]
    cls_ms = k_cls["ms"] / max(1, k_cls["launches"])
    achieved = wl.bytes_classify * wl.units_per_step / (cls_ms * 1e-3) / 1e9
    kernels = {k: round(v["ms"] / max(1, v["launches"]), 5) for k, v in timing_all.items() if v["launches"]}
    sum_ms = sum(kernels.values())
    step_bytes = (wl.bytes_classify + wl.bytes_compact) * wl.units_per_step
    step_achieved = step_bytes / (sum_ms * 1e-3) / 1e9
    ms_per_step = 1e3 * elapsed / steps
    traffic = step_traffic = source = None
    if not (wl.unfused or wl.place or wl.runs or wl.category_bytes or wl.shard or "singletons" in wl.layout):
        traffic, step_traffic, source = kernel_hash.load_traffic(wl.name, wl.n_pairs)
    roof = {"bound": "hbm", "kernel": wl.kernel_name(), "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": source,
            "algorithmic_bytes_per_unit": wl.bytes_classify, "kernel_ms": cls_ms}
    step = {"bound": "hbm", "what": "SURVEY 8d: (classify + compact) algorithmic bytes per %s / sum of the kernels' mean durations" % unit_name,
            "algorithmic_bytes_per_unit": wl.bytes_classify + wl.bytes_compact, "sum_kernel_ms": round(sum_ms, 5),
            "achieved": step_achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": step_achieved / HBM_PEAK_GBPS,
            "traffic": step_traffic, "traffic_source": source,
            "frac_by_ms_per_step": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    return roof, step, kernels


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    from xenomapper_amd import _ffi

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
    if args.sharded_input and args.workload != "cfg2":
        print("bench.py: --sharded-input is configs[3], i.e. the cfg2 columns", file=sys.stderr)
        sys.exit(2)
    # Rehearsal on a one-GPU box (not for reported numbers): XM_BENCH_REHEARSAL=1 lets every rank use cuda:0
    # and swaps RCCL for gloo, so that the N > 1 code path can be exercised where only one GPU is visible.
    rehearsal = os.environ.get("XM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    watchdog_s = float(os.environ.get("XM_BENCH_WATCHDOG_S", "120"))
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
    shard = None
    if args.sharded_input:
        total = args.strong_total or 400_000_000
        total -= total % (Workload.PARTS * 32)                       # whole 64-record words per part
        shard = (total, world)
        n_pairs = total // world
        args.strong_total = total
    elif args.strong_total:
        n_pairs = (args.strong_total // world + 3) // 4 * 4          # a read block per GPU, whole 64-record words of the unit mask
    ctx = _ffi.Context(local_rank)
    n_slots = max(args.steps, args.warmup, 1)
    wl = Workload(args.workload, ctx, dev, n_pairs, rank, args.mode, n_slots, args.singletons_pct, shard)
    job_counts = torch.zeros(64, dtype=torch.int64, device=dev)

    def finish(n_steps):
        torch.sum(wl.step_counts[:n_steps], dim=0, out=job_counts)   # category_counts of the job: the sum over its steps
        allreduce(job_counts, dist.ReduceOp.SUM)                     # RCCL over xGMI: 64 x int64, once per job

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed, timing, timing_all = time_steps(ctx, wl, args.steps, args.warmup, fence, finish)
    step_median_ms = median_step_ms(wl, 20)
    job_total = int(job_counts.sum().item())
    job_final = job_counts.clone()
    last_counts = wl.counts.clone()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    elapsed = float(allreduce(t, dist.ReduceOp.MAX).item()) if world > 1 else elapsed
    u = torch.tensor([wl.units_per_step], dtype=torch.int64, device=dev)
    units_all_ranks = int(allreduce(u, dist.ReduceOp.SUM).item()) if world > 1 else wl.units_per_step

    # parity of what was just timed (rank-local): against the C oracle on the same columns
    verified = None
    if not args.no_verify:
        from tests import helpers as H
        if rank == 0:
            H.c_oracle()                       # (re)builds oracle/libxm_oracle.so if stale: once, not in N ranks at a time
        if world > 1:
            dist.barrier()
        ok = wl.verify()
        ok &= job_total == units_all_ranks * args.steps
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
        verified = bool(allreduce(flag, dist.ReduceOp.MIN).item()) if world > 1 else ok

    line = None
    if rank == 0:
        unit_name = "read" if args.workload == "se" else "read-pair"
        roof, roof_step, kernels = rooflines(wl, elapsed, args.steps, timing, timing_all, unit_name)
        try:        # an on-box streaming-copy ceiling beside the spec peak, same footprint as the classify kernel
            roof["memcpy_d2d_GBps"] = copy_ceiling_gbps(dev, wl.bytes_classify * wl.units_per_step)
            probe = stream_ceiling_gbps(ctx, wl)
            roof["copy_ceiling_GBps"] = probe if probe is not None else roof["memcpy_d2d_GBps"]
            roof["copy_ceiling_kind"] = "xm_stream_probe_dev" if probe is not None else "hipMemcpyDtoD"
            roof["frac_of_copy"] = roof["achieved"] / roof["copy_ceiling_GBps"]
        except Exception as e:                                   # noqa: BLE001
            roof["copy_ceiling_GBps"], roof["frac_of_copy"], roof["copy_ceiling_error"] = None, None, "%s: %s" % (type(e).__name__, e)
        if args.sharded_input:
            job = ("ONE %d-pair input (%d seeded parts end to end) cut into %d read blocks of ~%d pairs with a one-record halo%s"
                   % (args.strong_total, Workload.PARTS, world, n_pairs,
                      " = BASELINE.json configs[3] (400 M pairs sharded across 8 GPUs, RCCL count all-reduce)"
                      if args.strong_total == 400_000_000 and world == 8 else ""))
            sharding = "sharded input: shard.plan_blocks cut + halo record per block (units belong to the block holding their second record)"
        else:
            job = ("%d read pairs in all, one %d-pair read block per GPU%s" % (world * n_pairs, n_pairs,
                   " = BASELINE.json configs[3] (400 M pairs sharded across 8 GPUs, RCCL count all-reduce)"
                   if world * n_pairs == 400_000_000 and world == 8 and args.workload == "cfg2" else ""))
            sharding = "independent read block per GPU (weak), no halo exchange"
        line = {
            "metric": "reads/sec classified" if args.workload == "se" else "read-pairs/sec classified",
            "value": units_all_ranks * args.steps / elapsed,
            "unit": unit_name + "s/s",
            "n_gpus": world, "n_ranks_seen": n_ranks_seen, "collective_backend": backend,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_step_median": step_median_ms,
            "higher_is_better": True, "scaling": "strong" if args.strong_total else "weak", "vs_baseline": None,
            "dtype": wl.dtype, "data": "synthetic" + (" (REHEARSAL: ranks share one GPU, gloo)" if rehearsal else ""),
            "config": {"workload": wl.describe(), "workload_short": wl.short(), "step_short": wl.step_short(),
                       "sharding_short": ("sharded input (%d pairs, halo)" % args.strong_total if args.sharded_input else
                                          "strong" if args.strong_total else "weak: own block per GPU") + (", 1 RCCL all-reduce of counts" if world > 1 else ""),
                       "job": job,
                       "pairs_per_gpu": n_pairs, "records_per_species_per_gpu": wl.n, "record_layout": wl.layout,
                       "step": wl.call_name(),
                       "category_per_record": ("category byte (fwd*8+rev, 1 B per record)" if wl.category_bytes else
                                               "compact stream bins4: the output bin as a nibble per record = 1 B per pair (SURVEY 8d's "
                                               "algorithmic figure); the category byte is produced on request and checked outside the timed region"),
                       "sharding": sharding + (", one RCCL all-reduce of the final category_counts" if world > 1 else "")},
            "roofline": roof, "roofline_step": roof_step,
            "kernel_ms": kernels, "kernel_ms_note": "all kernels bracketed in a separate pass after the timed region (scan = its launches)",
            "verified_vs_oracle": verified,
            "xm_allreduce_counts": None,
        }
        host_cols = getattr(wl, "host_cols", None)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_python()
            if host_cols is not None and args.workload in ("cfg2", "cfg5") and not args.sharded_input:
                m = min(n_pairs, 5_000_000)
                sub = {k: np.ascontiguousarray(v[:2 * m]) for k, v in host_cols.items() if k != "unit_bits"}
                sub["unit_bits"] = np.ascontiguousarray(host_cols["unit_bits"][:(2 * m + 63) // 64])
                line["cpu_baseline_c"] = cpu_baseline_c(sub, m)
        if world == 1 and not args.no_e2e and args.workload == "cfg2" and not args.sharded_input and not args.singletons_pct:
            e2e = {"note": "transfer-inclusive rates beside `value` (which is HBM-resident); measured after the timed region"}
            try:
                hc = host_cols if host_cols is not None else wl.host_columns()
                e2e["h2d_inclusive"] = e2e_h2d_inclusive(ctx, wl.mode, hc, min(n_pairs, 25_000_000))
            except Exception as e:                               # noqa: BLE001
                e2e["h2d_inclusive"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                e2e["pcie_ceiling"] = pcie_ceiling(dev)
            except Exception as e:                               # noqa: BLE001
                e2e["pcie_ceiling"] = {"error": "%s: %s" % (type(e).__name__, e)}
            for key, to_files, gpu_strip in (("sam_text", True, True), ("sam_text_devnull", False, True),
                                             ("sam_text_host_stripper", True, False)):
                try:
                    e2e[key] = e2e_sam_text(to_files=to_files, gpu_strip=gpu_strip)
                except Exception as e:                           # noqa: BLE001
                    e2e[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                sys.path.insert(0, os.path.join(REPO, "tools"))
                import bench_bam
                wd = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
                r = bench_bam.run(copies=48000, workdir=wd)
                e2e["bam"] = {"read_pairs_per_s": r["value"], "pairs": r["units"], "bam_GBps": r["bam_GBps"], "seconds": round(r["seconds"], 4),
                              "threads": r["threads"], "phases": r["phases"],
                              "what": "two BAM files (the reference's fixtures tiled 48 000 times, record-aligned BGZF blocks as samtools writes "
                                      "them) -> BGZF blocks inflated, records found and stripped ON THE GPU (xm_bamdev), columns stay in HBM -> "
                                      "fused pass -> the records' SAM text printed and the six outputs gathered ON THE GPU (xm_bamdev_fetch_bins), one stream per "
                                      "window back over the link -> six SAM outputs on /dev/null"}
                for key, kw, what in (("bam_cigar_scores", {"cigar_scores": True}, "the same input with the --cigar_scores plugin (NM + the records' CIGAR "
                                                                              "words packed into the CIGAR columns on the device)"),
                                      ("bam_single_end", {"single_end": True}, "the same files as single-end input: the skipping walk, as the "
                                                                          "command line runs it (reads/s)")):
                    rk = bench_bam.run(copies=48000, workdir=wd, **kw)
                    e2e[key] = {"per_s": rk["value"], "units": rk["units"], "seconds": round(rk["seconds"], 4), "what": what}
                os.environ["XENOMAPPER_GPU_BAM"] = "0"
                try:
                    r0 = bench_bam.run(copies=48000, workdir=wd)
                    e2e["bam_host_decoder"] = {"read_pairs_per_s": r0["value"], "seconds": round(r0["seconds"], 4), "phases": r0["phases"],
                                               "what": "the same input through the host decoder (XENOMAPPER_GPU_BAM=0: inflate and printing on CPU threads)"}
                finally:
                    os.environ.pop("XENOMAPPER_GPU_BAM", None)
            except Exception as e:                               # noqa: BLE001
                e2e["bam"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                sys.path.insert(0, os.path.join(REPO, "tools"))
                import bench_inflate
                e2e["inflate"] = bench_inflate.run(out_gb=1.0, reps=5, ctx=ctx)
            except Exception as e:                               # noqa: BLE001
                e2e["inflate"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                sys.path.insert(0, os.path.join(REPO, "tools"))
                import bench_e2e
                e2e["host_ceilings"] = bench_e2e.host_ceilings("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
                e2e["host_ceilings"]["what"] = ("the file path's own rooflines on this box: memory copy (one core / all granted threads), "
                                                "one write(2) stream into tmpfs, posix_fallocate alone, pread(2) of a tmpfs file into page-locked memory by thread "
                                                "count (what feeds the link in e2e.sam_text*.strip.upload_GBps), the writer gathering lines into a "
                                                "mapped tmpfs file and into memory, the stripper's GB/s of SAM text against its thread count")
            except Exception as e:                               # noqa: BLE001
                e2e["host_ceilings"] = {"error": "%s: %s" % (type(e).__name__, e)}
            line["e2e"] = e2e

    # The other BASELINE configs, timed the same way in this process after the headline run (default invocation only):
    # configs[2] --cigar_scores and configs[4] HISAT ZS + conservative (one GPU), and -- at every N -- configs[3]'s input:
    # 400 M pairs as ONE input cut into N read blocks with the halo record (strong scaling), so that one scaling run of
    # the default command yields both curves of SURVEY 8d.  `value`/`config` stay configs[1].
    plain_default = (args.workload == "cfg2" and not args.no_extra_workloads and not args.sharded_input
                     and not args.singletons_pct and not args.strong_total and args.pairs == 50_000_000
                     and not wl.unfused and not wl.place and not wl.runs and os.environ.get("XM_BENCH_CATEGORY_BYTES") != "1")
    default_run = world == 1 and plain_default
    extra = {}
    if plain_default:
        del wl
        torch.cuda.empty_cache()
    if default_run:
        for key, name in EXTRA_WORKLOADS:
            try:
                w2 = Workload(name, ctx, dev, n_pairs, rank, None, 20)
                el2, tm2, tma2 = time_steps(ctx, w2, 20, 5, fence)
                ok2 = None if args.no_verify else bool(w2.verify())
                r2, rs2, k2 = rooflines(w2, el2, 20, tm2, tma2, "read-pair")
                extra[key] = {"workload": w2.describe(), "step": w2.call_name(), "steps": 20, "warmup": 5,
                              "ms_per_step": 1e3 * el2 / 20, "value": w2.units_per_step * 20 / el2, "unit": "read-pairs/s",
                              "dtype": w2.dtype, "roofline": r2, "roofline_step": rs2, "kernel_ms": k2,
                              "verified_vs_oracle": ok2}
                if name == "cfg3":
                    extra[key]["mean_cigar_ops_per_record"] = w2.k_bar
                    # the timed step takes PACKED CIGAR columns (packed once, during set-up); callers that hold CSR columns pay
                    # xm_cigar_pack on the host per block (the host-buffer entry points do it themselves) -- its rate, one thread:
                    try:
                        m = min(w2.n, 10_000_000)
                        off = w2.cig[0]["cig_off"][:m + 1].cpu().numpy().view(np.uint32)
                        ops = w2.cig[0]["cig_oplen"][:int(off[-1])].cpu().numpy().view(np.uint32)
                        t0 = time.perf_counter()
                        _ffi.cigar_pack(off, ops)
                        el = time.perf_counter() - t0
                        extra[key]["host_cigar_pack"] = {"records": m, "seconds": round(el, 4), "M_records_per_s": m / el / 1e6,
                                                         "what": "xm_cigar_pack (CSR -> packed columns) on one host thread; outside the timed region"}
                    except Exception as e:                       # noqa: BLE001
                        extra[key]["host_cigar_pack"] = {"error": "%s: %s" % (type(e).__name__, e)}
                del w2
                torch.cuda.empty_cache()
            except Exception as e:                               # noqa: BLE001 -- reported, the headline line still goes out
                extra[key] = {"error": "%s: %s" % (type(e).__name__, e)}
    if default_run:
        # the segmented-lists form of configs[1] (one launch, no scan / scatter): its own entry, its own algorithmic bytes
        try:
            w4 = Workload("cfg2", ctx, dev, n_pairs, rank, None, 20, form="runs")
            el4, tm4, tma4 = time_steps(ctx, w4, 20, 5, fence)
            med4 = median_step_ms(w4, 20)
            ok4 = None if args.no_verify else bool(w4.verify())
            r4, rs4, k4 = rooflines(w4, el4, 20, tm4, tma4, "read-pair")
            extra["runs"] = {"workload": w4.describe(), "step": w4.call_name(), "steps": 20, "warmup": 5, "ms_per_step": 1e3 * el4 / 20,
                             "ms_per_step_median": med4, "value": w4.units_per_step * 20 / el4, "unit": "read-pairs/s", "dtype": w4.dtype,
                             "roofline": r4, "roofline_step": rs4, "kernel_ms": k4, "verified_vs_oracle": ok4,
                             "note": "output contract = per-granule runs (2 B per unit) + counts, NOT the flat lists of SURVEY 8b (4): "
                                     "bytes_per_unit is this form's own (32 in + 2 out + counts), `value`/`roofline_step` of the line stay flat"}
            del w4
            torch.cuda.empty_cache()
        except Exception as e:                                   # noqa: BLE001
            extra["runs"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if default_run and world == 1:
        # the same fused call captured ONCE into a HIP graph and replayed (the *_dev entry points only enqueue work; a caller that
        # classifies batch after batch into the same buffers can do this): what the three launches' gaps cost.  Its own entry --
        # `value` stays the eager call's.
        try:
            w5 = Workload("cfg2", ctx, dev, n_pairs, rank, None, 1)
            ctx.timing_enable(False)                                 # (no event pairs inside the capture)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                            # warm-up outside the capture, as torch asks for
                for _ in range(3):
                    w5.step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                w5.step()
            for _ in range(5):
                graph.replay()
            fence()
            t0 = time.perf_counter()
            for _ in range(20):
                graph.replay()
            fence()
            el5 = time.perf_counter() - t0
            ok5 = None if args.no_verify else bool(w5.verify())
            extra["graph"] = {"workload": w5.describe(), "step": w5.call_name() + " captured into a HIP graph, replayed", "steps": 20, "warmup": 5,
                              "ms_per_step": 1e3 * el5 / 20, "value": w5.units_per_step * 20 / el5, "unit": "read-pairs/s",
                              "verified_vs_oracle": ok5}
            del graph, w5
            torch.cuda.empty_cache()
        except Exception as e:                                   # noqa: BLE001
            extra["graph"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if plain_default:
        try:
            total = 400_000_000 - 400_000_000 % (Workload.PARTS * 32)
            w3 = Workload("cfg2", ctx, dev, total // world, rank, None, 10, 0.0, (total, world), form="place")
            el3, _, tma3 = time_steps(ctx, w3, 10, 3, fence)
            t3 = torch.tensor([el3], dtype=torch.float64, device=dev)
            el3 = float(allreduce(t3, dist.ReduceOp.MAX).item()) if world > 1 else el3
            u3 = torch.tensor([w3.units_per_step], dtype=torch.int64, device=dev)
            units3 = int(allreduce(u3, dist.ReduceOp.SUM).item()) if world > 1 else w3.units_per_step
            ok3 = None
            if not args.no_verify:
                f3 = torch.tensor([1 if w3.verify() else 0], dtype=torch.int64, device=dev)
                ok3 = bool(allreduce(f3, dist.ReduceOp.MIN).item()) if world > 1 else bool(f3.item())
            extra["sharded_input"] = {"total_pairs": total, "blocks": world, "steps": 10, "warmup": 3, "ms_per_step": 1e3 * el3 / 10,
                                      "value": units3 * 10 / el3, "unit": "read-pairs/s", "scaling": "strong", "units_per_step": units3,
                                      "step": w3.call_name(),
                                      # per STEP (a large read block takes several launches of each kernel: xm_api.hip place_steps)
                                      "kernel_ms": {k: round(v["ms"] / 10, 5) for k, v in tma3.items() if v["launches"]},
                                      "launches_per_step": {k: v["launches"] / 10 for k, v in tma3.items() if v["launches"]},
                                      "verified_vs_oracle": ok3}
            del w3
            torch.cuda.empty_cache()
        except Exception as e:                                   # noqa: BLE001
            extra["sharded_input"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and extra:
        line["workloads"] = extra
        line["workloads_note"] = ("timed in this process after the headline run, same protocol: configs[2] / configs[4] 5 warm-up + 20 "
                                  "steps (one GPU only); sharded_input = configs[3]'s 400 M-pair input cut into N read blocks with halo, six-list output (xm_classify_place_dev), "
                                  "3 warm-up + 10 steps, max over ranks")

    # The same reduction through the library's own RCCL communicator (xm_allreduce_counts, the C ABI's collective), on
    # every rank, after everything else and under a watchdog: the job total above came from torch.distributed, so
    # whatever happens here the line is printed -- with the outcome (or the error) in it -- and a hang exits non-zero.
    def library_allreduce():
        with stdout_to_stderr():
            uid = [None]
            if rank == 0:
                try:
                    uid = [_ffi.comm_unique_id()]
                except Exception as e:                           # noqa: BLE001 -- every rank must still reach the broadcast
                    uid = ["%s: %s" % (type(e).__name__, e)]
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            if not isinstance(uid[0], bytes):
                raise RuntimeError("xm_comm_unique_id on rank 0: %s" % uid[0])
            if os.environ.get("XM_BENCH_FAKE_HANG") == "1":      # test hook: a collective that never returns
                time.sleep(3600)
            ctx.comm_init(world, rank, uid[0])
            mine = last_counts * args.steps                      # every step saw the same block: this rank's job total
            ctx.allreduce_counts(mine)
            torch.cuda.synchronize()
            out = {"ranks": ctx.comm_size(), "matches_torch_distributed": bool(torch.equal(mine, job_final))}
            ctx.comm_destroy()
            return out

    if not rehearsal or os.environ.get("XM_BENCH_FAKE_HANG") == "1":
        try:
            outcome = run_with_watchdog(library_allreduce, watchdog_s, line, "xm_allreduce_counts", rank)
        except Exception as e:                                   # noqa: BLE001 -- reported, not raised
            outcome = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            line["xm_allreduce_counts"] = outcome
    if rank == 0:
        # the full record goes to a file, the printed line stays short (numbers only)
        path = os.environ.get("XM_BENCH_FULL_JSON") or os.path.join(REPO, "gpurun_out", "bench_full_%dgpu_%s.json" % (world, args.workload))
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "wt") as fh:
                json.dump(line, fh, indent=1)
            line["full_record"] = os.path.relpath(path, REPO)
        except OSError:
            line["full_record"] = None
        out = compact_line(line)
        if os.environ.get("XM_BENCH_VERBOSE") == "1":
            out = line
        print(json.dumps(out, separators=(",", ":")), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
