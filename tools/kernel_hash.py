"""Identity of the kernel sources a measurement belongs to.

`profiles/pmc_traffic.json` (HBM bytes per launch from rocprofv3 --pmc passes) is replayed by bench.py next to freshly
measured times; it carries the hash below, and bench.py prints the traffic only while the tree still hashes to it
(otherwise `traffic: null, traffic_source: "stale"`).  The hash covers the files the device code is compiled from, the
file that decides which launches make up a step (xm_api.hip) with the header it implements, and any extra hipcc flags of
a tuning build.
"""
import hashlib
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("xenomapper_amd/csrc/xm_kernels.hip", "xenomapper_amd/csrc/xm_kernels.h", "xenomapper_amd/csrc/xm_api.hip",
                  "include/xenomapper_hip.h")
# the strip kernels' PMC table (profiles/*_strip_pmc.json) is guarded the same way, by the sources IT was measured on
STRIP_SOURCES = ("xenomapper_amd/csrc/xm_strip.hip", "include/xenomapper_strip.h")


def kernel_src_sha256(repo=REPO, flags=None, sources=None):
    h = hashlib.sha256()
    for rel in (sources or KERNEL_SOURCES):
        with open(os.path.join(repo, rel), "rb") as fh:
            h.update(rel.encode() + b"\0" + fh.read() + b"\0")
    flags = os.environ.get("XENOMAPPER_HIPCC_FLAGS", "") if flags is None else flags
    h.update(" ".join(flags.split()).encode())
    return h.hexdigest()


def load_traffic(workload, n_pairs, repo=REPO, path=None):
    """(classify bytes per launch, step bytes, source text) for `workload` from profiles/pmc_traffic.json, or
    (None, None, reason) when there is no entry, the size differs, or the kernels have changed since it was collected."""
    import json
    path = path or os.path.join(repo, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None, None, None
    ent = rec.get("workloads", {}).get(workload)
    if not ent or ent.get("pairs") != n_pairs:
        return None, None, None
    if rec.get("kernel_src_sha256") != kernel_src_sha256(repo):
        return None, None, "stale"
    src = "profiles/pmc_traffic.json <- %s (rocprofv3 --pmc passes of this command on these kernel sources, replayed; not measured in this run)" % ent.get("profile")
    return ent.get("classify_hbm_bytes_per_launch"), ent.get("step_hbm_bytes"), src


def strip_src_sha256(repo=REPO, flags=None):
    return kernel_src_sha256(repo, flags, STRIP_SOURCES)


if __name__ == "__main__":
    import sys
    print(strip_src_sha256() if "--strip" in sys.argv else kernel_src_sha256())
