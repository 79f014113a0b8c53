"""bench.py's host-side machinery that must work before any GPU is involved: the watchdog around the library
collective (a hang must put the JSON line on the REAL stdout and exit non-zero) and the tie between the replayed PMC
traffic figures and the kernel sources they were measured on."""
import json
import os
import shutil
import subprocess
import sys

from tests import helpers as H


def test_watchdog_delivers_the_line_on_stdout_and_exits_nonzero():
    """A collective that never returns, entered -- as bench.py does -- while fd 1 is redirected to stderr."""
    prog = (
        "import sys, time; sys.path.insert(0, %r); import bench\n"
        "line = {'metric': 'm', 'value': 1.0, 'xm_allreduce_counts': None}\n"
        "def hang():\n"
        "    with bench.stdout_to_stderr():\n"
        "        print('RCCL banner that must not reach stdout')\n"
        "        time.sleep(60)\n"
        "bench.run_with_watchdog(hang, 0.5, line, 'xm_allreduce_counts', rank=0)\n" % H.REPO)
    p = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3
    rec = json.loads(p.stdout)                                  # exactly the line, nothing else
    assert rec["value"] == 1.0 and "no answer within" in rec["xm_allreduce_counts"]["error"]
    assert "RCCL banner" in p.stderr and "gave up" in p.stderr


def test_watchdog_other_ranks_exit_nonzero_without_a_line():
    prog = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "bench.run_with_watchdog(lambda: time.sleep(60), 0.3, None, 'xm_allreduce_counts', rank=1)\n" % H.REPO)
    p = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3 and p.stdout == ""


def test_watchdog_is_transparent_when_the_call_returns():
    sys.path.insert(0, H.REPO)
    import bench
    line = {"x": None}
    assert bench.run_with_watchdog(lambda: 42, 5.0, line, "x") == 42 and line == {"x": None}


def test_replayed_traffic_goes_stale_when_a_kernel_source_changes(tmp_path):
    """profiles/pmc_traffic.json carries the hash of xm_kernels.hip + xm_kernels.h; editing a comment flips bench.py's
    traffic to null / "stale"."""
    sys.path.insert(0, os.path.join(H.REPO, "tools"))
    import kernel_hash
    repo = tmp_path / "repo"
    for rel in kernel_hash.KERNEL_SOURCES:
        (repo / os.path.dirname(rel)).mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join(H.REPO, rel), repo / rel)
    (repo / "profiles").mkdir()
    rec = {"kernel_src_sha256": kernel_hash.kernel_src_sha256(str(repo), flags=""),
           "workloads": {"cfg2": {"pairs": 50_000_000, "classify_hbm_bytes_per_launch": 1.7e9, "step_hbm_bytes": 2.0e9,
                                  "profile": "profiles/rXX_cfg2_pmc.json"}}}
    (repo / "profiles" / "pmc_traffic.json").write_text(json.dumps(rec))
    os.environ.pop("XENOMAPPER_HIPCC_FLAGS", None)
    assert kernel_hash.load_traffic("cfg2", 50_000_000, str(repo))[:2] == (1.7e9, 2.0e9)
    assert kernel_hash.load_traffic("cfg2", 1_000_000, str(repo)) == (None, None, None)        # another size: no claim
    assert kernel_hash.load_traffic("cfg3", 50_000_000, str(repo)) == (None, None, None)       # never collected
    with open(repo / kernel_hash.KERNEL_SOURCES[0], "a") as fh:
        fh.write("// a comment\n")
    assert kernel_hash.load_traffic("cfg2", 50_000_000, str(repo)) == (None, None, "stale")
    os.environ["XENOMAPPER_HIPCC_FLAGS"] = "-DXM_CIGP_WPE=0"
    try:
        assert kernel_hash.kernel_src_sha256(str(repo)) != kernel_hash.kernel_src_sha256(str(repo), flags="")
    finally:
        del os.environ["XENOMAPPER_HIPCC_FLAGS"]


def test_committed_traffic_file_matches_its_schema():
    path = os.path.join(H.REPO, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return
    rec = json.load(open(path))
    assert len(rec["kernel_src_sha256"]) == 64
    for name, ent in rec["workloads"].items():
        assert name in ("cfg2", "cfg3", "cfg5", "f64", "se") and ent["pairs"] > 0
        assert ent["classify_hbm_bytes_per_launch"] > 0 and ent["step_hbm_bytes"] >= ent["classify_hbm_bytes_per_launch"]


def test_singleton_flags_model():
    sys.path.insert(0, H.REPO)
    import bench
    f = bench.singleton_unit_flags(200_000, 0.01, seed=3)
    assert f[0] == 0 and f.dtype.itemsize == 1
    pairs = int(f.sum())
    singles = 200_000 - 2 * pairs
    assert 0.005 < singles / (pairs + singles) < 0.02
    import numpy as np
    assert not (f[1:] & f[:-1]).any()                          # two units never touch: a unit's first record closes none
    assert np.flatnonzero(f)[:50].tolist() != list(range(1, 100, 2))   # parity flips: not strictly interleaved


def test_printed_line_stays_short_enough_for_a_truncating_record():
    """The ONE line bench.py prints must survive a record that keeps only its last 2 000 characters: numbers only,
    every workload in it (configs[2] / configs[4] / the sharded input / the transfer-inclusive rates)."""
    import json
    import bench
    roof = {"bound": "hbm", "kernel": "classify_kernel<int32, paired, counts>", "achieved": 5825.123456, "peak": 8000.0, "unit": "GB/s",
            "frac": 0.72814043, "traffic": 1724123456.0, "traffic_source": "x" * 300, "algorithmic_bytes_per_unit": 33.0,
            "kernel_ms": 0.283262, "copy_ceiling_GBps": 6234.5678, "memcpy_d2d_GBps": 5377.123, "frac_of_copy": 0.93421}
    step = {"frac": 0.694612, "frac_by_ms_per_step": 0.677912, "sum_kernel_ms": 0.34194, "algorithmic_bytes_per_unit": 38.0,
            "traffic": 1998123456.0, "what": "y" * 300}
    sub = {"ms_per_step": 0.514341234, "roofline": dict(roof), "roofline_step": dict(step), "verified_vs_oracle": True, "workload": "z" * 400}
    full = {"metric": "read-pairs/sec classified", "value": 142726123456.789, "unit": "read-pairs/s", "n_gpus": 8, "steps": 50, "warmup": 5,
            "ms_per_step": 0.350321234, "ms_per_step_median": 0.339481234, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload_short": "configs[1]: 50000000 PE 2x150 pairs/GPU, AS/XS, liberal, HBM-resident", "pairs_per_gpu": 50_000_000,
                       "step_short": "xm_classify_compact_dev: classify+count, scan, scatter (3 launches)",
                       "sharding_short": "weak: own block per GPU, 1 RCCL all-reduce of counts", "workload": "w" * 500},
            "roofline": roof, "roofline_step": step, "kernel_ms": {"classify": 0.280412, "scan": 0.008012, "scatter": 0.053551},
            "cpu_baseline": {"value": 135060.123, "unit": "read-pairs/s", "cores": 1, "kind": "port", "sample": "s" * 300,
                             "sample_short": "82 x 20000-pair SAM text twin, parse+classify+write, 12 s"},
            "verified_vs_oracle": True, "n_ranks_seen": 8, "xm_allreduce_counts": {"ranks": 8, "matches_torch_distributed": True},
            "workloads": {"configs[2]": sub, "configs[4]": sub, "runs": dict(sub, ms_per_step_median=0.289412345),
                          "sharded_input": {"ms_per_step": 3.127712, "value": 127890123456.0, "verified_vs_oracle": True}},
            "e2e": {"h2d_inclusive": {"registered_buffers": {"read_pairs_per_s": 1.56e9}}, "sam_text": {"read_pairs_per_s": 6.869e6},
                    "bam": {"read_pairs_per_s": 6.2e6}, "sam_text_host_stripper": {"read_pairs_per_s": 6.1e6},
                    "bam_host_decoder": {"read_pairs_per_s": 3.5e6}, "inflate": {"value": 28.3312345}, "note": "n" * 500},
            "full_record": "gpurun_out/bench_full_8gpu_cfg2.json"}
    line = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(line) < 1900, len(line)
    rec = json.loads(line)
    assert rec["workloads"]["cfg3"][:3] == [0.51434, 0.7281, 0.6779] and rec["workloads"]["sharded"][2] is True
    assert rec["roofline"]["frac"] == 0.7281 and rec["cpu_baseline"]["cores"] == 1 and rec["e2e"] == [1.56, 6.869, 6.2, 6.1, 3.5, 28.33]
    assert rec["workloads"]["runs"] == [0.51434, 0.28941, 0.6779, True]
    # an error in a side measurement is carried as text, the line still parses and stays short
    full["workloads"]["configs[2]"] = {"error": "RuntimeError: " + "e" * 500}
    full["xm_allreduce_counts"] = {"error": "watchdog: " + "h" * 500}
    line = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(line) < 1900 and json.loads(line)["workloads"]["cfg3"][0] is None
