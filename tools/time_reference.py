#!/usr/bin/env python3
"""Time the *reference* (imported from /root/reference, build container only) on the synthetic text twins, for
BASELINE.md section 2 (SURVEY.md 8d "CPU baseline beside it", item 1).  Prints a markdown table."""
import io, os, sys, time
REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.isdir(os.path.join(REF, "xenomapper")):
    sys.exit("reference not mounted")
sys.dont_write_bytecode = True
sys.path.insert(0, REF); sys.path.insert(0, REPO)
from xenomapper import xenomapper as ref
from xenomapper_amd import synth

def run(t1, t2, loop, tag_func, reps, **kw):
    units = 0
    t0 = time.perf_counter()
    for _ in range(reps):
        s1, s2 = io.StringIO(t1), io.StringIO(t2)
        outs = {k: open(os.devnull, "wt") for k in ("primary_specific", "secondary_specific", "primary_multi",
                                                     "secondary_multi", "unassigned", "unresolved")}
        ref.process_headers(s1, s2, **outs)
        c = loop(ref.getReadPairs(s1, s2, **kw), tag_func=tag_func, **outs)
        units += sum(c.values())
        for o in outs.values():
            o.close()
    return units / (time.perf_counter() - t0)

rows = []
t1, t2, _ = synth.sam_text_pair(n_pairs=20000, seed=2002, profile="bowtie2", paired=True, read_len=150)
rows.append(("cfg2 twin: main_paired_end + get_tag, 20 k pairs x 10", run(t1, t2, ref.main_paired_end, ref.get_tag, 10)))
rows.append(("cfg2 twin: conservative_main_paired_end + get_tag", run(t1, t2, ref.conservative_main_paired_end, ref.get_tag, 10)))
t1, t2, _ = synth.sam_text_pair(n_pairs=20000, seed=3003, profile="cigar", paired=True, read_len=150)
rows.append(("cfg3 twin: main_paired_end + get_cigarbased_AS_tag", run(t1, t2, ref.main_paired_end, ref.get_cigarbased_AS_tag, 10)))
t1, t2, _ = synth.sam_text_pair(n_pairs=20000, seed=5005, profile="hisat", paired=True, read_len=150)
rows.append(("cfg5 twin: conservative_main_paired_end + get_tag_with_ZS_as_XS", run(t1, t2, ref.conservative_main_paired_end, ref.get_tag_with_ZS_as_XS, 10)))
t1, t2, _ = synth.sam_text_pair(n_pairs=40000, seed=1001, profile="bowtie2", paired=False, read_len=50, mixed_ws=0.2)
rows.append(("cfg1 twin: main_single_end + get_tag (reads/s), skip_repeated", run(t1, t2, ref.main_single_end, ref.get_tag, 10, skip_repeated_reads=True)))
print("| path (reference v1.0.2 imported in place, 1 core) | units/s |\n|---|---|")
for name, v in rows:
    print("| %s | %.0f |" % (name, v))
