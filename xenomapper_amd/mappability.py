"""Drop-in for the reference's experimental `xenomappability` tool (xenomapper/mappability.py, v1.0.2;
SURVEY.md 8f-4): paired-end mappability estimated from single-end mappability and the observed insert-size
distribution.  Same names and behaviour; the one numeric loop -- Mappability.single_end_to_paired, a
correlation of every chromosome track with the mate density (ref :94-124) -- runs on the GPU through
xm_mate_correlate (bit-identical to the Python loop: binary64, multiply and add rounded separately, left to
right).  Everything else here is text handling and small-list statistics and stays on the host.
"""
from __future__ import annotations

import argparse
import itertools
import sys
from collections import Counter
from statistics import mean

import numpy as np

from .xenomapper import default_context, get_sam_header

__version__ = "1.0.2"


class Mappability(dict):
    """{chromosome: per-position mappability list}, with wiggle I/O (ref :38-124)."""

    def __init__(self, chromosome_sizes={}):
        if chromosome_sizes:
            for chrom in chromosome_sizes:
                self[chrom] = [0] * chromosome_sizes[chrom]
        self.chromosome_sizes = chromosome_sizes

    def to_wiggle(self, wigglefile=sys.stdout, chromosomes=[]):
        """fixedStep wiggle, start=1 step=1, chromosomes in sorted order (ref :46-57)."""
        for chrom in sorted(self):
            if not chromosomes or chrom in chromosomes:
                wigglefile.write("fixedStep\tchrom={0}\tstart=1\tstep=1\n".format(chrom))
                wigglefile.write("".join(str(score) + "\n" for score in self[chrom]))

    def from_wiggle(self, wigglefile=sys.stdin, datatype=float):
        """Load chromosomes from a fixedStep wiggle (declaration lines "fixedStep<TAB>chrom=NAME<TAB>start=1<TAB>step=1",
        one value per line after them), replacing tracks of the same name; may be called again for more files.
        Behaviour pinned by the reference's tests (ref :59-92): a declared track that stays empty is dropped unless it is
        the last one of the file, which is always stored -- under the key None when the file declares nothing."""
        tracks = [(None, [])]                         # (name, values) in file order; values go to the last one
        for text in wigglefile:
            if not text.startswith("fixedStep"):
                tracks[-1][1].append(datatype(text))
                continue
            declared = text.strip().split("\t")
            key_and_name = declared[1].split("=")
            if key_and_name[0] != "chrom" or declared[2] != "start=1" or declared[3] != "step=1":
                raise ValueError('Unsupported wiggle fixed step format [must be in the format "fixedStep '
                                 'chrom=chrX start=1 step=1"] {0}'.format(declared))
            tracks.append((key_and_name[1], []))
        for name, values in tracks[:-1]:
            if name and values:
                self[name] = values
        self[tracks[-1][0]] = tracks[-1][1]
        self.chromosome_sizes.update((name, len(track)) for name, track in self.items())

    def single_end_to_paired(self, mate_density=[1]):
        """Paired-end mappability: a uniquely mappable position stays 1.0, any other gets the density-weighted
        mappability of where its mate may fall (ref :94-124).  One GPU launch per chromosome."""
        assert abs(sum(mate_density) - 1.0) < 0.000001
        ctx = default_context()
        density = np.asarray(mate_density, dtype=np.float64)
        paired = Mappability(chromosome_sizes=self.chromosome_sizes)
        for chrom in self:
            track = np.asarray(self[chrom], dtype=np.float64)
            values = ctx.mate_correlate(track, density).tolist()
            if len(values) != len(paired[chrom]):          # the reference would raise IndexError while assigning
                raise IndexError("list assignment index out of range")
            paired[chrom] = values
        return paired


def parse_fasta(fastafile, token=">"):
    """(name, sequence) of every record of a (multi-)FASTA handle, streamed record by record; the handle is closed when
    the generator finishes.  As the reference's parser (ref :127-146): lines are stripped, a header is a line that starts
    with `token` and its name is that line without its first character, lines in front of the first header are ignored,
    and a record whose name is empty is not reported."""
    def numbered(handle):                                  # every line with the number of headers seen so far
        seen = 0
        for raw in handle:
            line = raw.strip()
            seen += line.startswith(token)
            yield seen, line
    with fastafile as handle:
        for record, lines in itertools.groupby(numbered(handle), key=lambda pair: pair[0]):
            if record == 0:
                continue
            header = next(lines)[1]
            sequence = "".join(line for _, line in lines)
            if header[1:]:
                yield (header[1:], sequence)


def make_blocklist(seqstring, block_size=80):
    """The sequence cut into block_size pieces (ref :148-157)."""
    return [seqstring[start:start + block_size] for start in range(0, len(seqstring), block_size)]


def slice_string_in_blocks(seqstring, block_size=80):
    return "\n".join(make_blocklist(seqstring, block_size)) + "\n"


def format_fasta(name, seq, block_size=80):
    return ">" + name + "\n" + slice_string_in_blocks(seq, block_size)


def simulate_reads(fastafile, readlength=100, outfile=sys.stdout):
    """Every readlength window of every sequence as a FASTA read named chrom_<1-based start> (ref :168-173)."""
    for name, seq in parse_fasta(fastafile):
        chrom = name.split()[0]
        for start in range(len(seq) - readlength + 1):
            outfile.write(format_fasta("{0}_{1}".format(chrom, start + 1), seq[start:start + readlength]))


def single_end_mappability_from_sam(samfile, outfile=sys.stdout, fill_sequence_gaps=True, chromosome_sizes={}):
    """Single-end mappability wiggle from the name-sorted SAM of simulated reads: 1 where the read maps back to
    its origin with MAPQ 42 (ref :175-214)."""
    mappable = Mappability(chromosome_sizes=chromosome_sizes)
    get_sam_header(samfile)
    current, position, values = None, 0, []
    for line in samfile:
        name, _flag, chrom, pos, mapq = line.strip("\n").split()[:5]
        true_chrom = "_".join(name.split("_")[:-1])
        if current != true_chrom:
            if current and values:
                mappable[current] = values
            current, position, values = chrom, 0, []          # keyed by RNAME, as the reference does (:189)
        position += 1
        name_pos = int(name.split("_")[-1])
        if name_pos != position:
            if not name_pos > position:
                raise ValueError("Name is not sequential.  SAM must be in name sorted order Name: {0} Expected: "
                                 "{1}_{2}".format(name, current, position))
            values.extend([0] * (name_pos - position))
            position = name_pos
        values.append(1 if (true_chrom, name_pos) == (chrom, int(pos)) and mapq == "42" else 0)
    if current and values:
        mappable[current] = values
    mappable.to_wiggle(wigglefile=outfile)


def paired_end_mappability(wiggle, mate_density, outfile=sys.stdout, chromosome_sizes={}):
    """Paired-end mappability wiggle from a single-end wiggle and a mate density (ref :216-235).  Only the chromosomes
    named in chromosome_sizes BEFORE the wiggle is read are written (all of them when it is empty): reading adds the
    wiggle's own chromosomes to that dictionary."""
    wanted = [*chromosome_sizes]
    single_end = Mappability(chromosome_sizes=chromosome_sizes)
    single_end.from_wiggle(wiggle, datatype=float)
    paired = single_end.single_end_to_paired(mate_density=mate_density)        # the GPU part
    paired.to_wiggle(wigglefile=outfile, chromosomes=wanted)


def smoothed_list(the_list, width=10):
    """Running mean over the window [x - width - 1, x + width) clipped at the front (ref :237-238); statistics.mean, i.e.
    the exactly rounded mean, as there."""
    out = []
    for x in range(len(the_list)):
        first = x - width - 1
        out.append(mean(the_list[first if first > 0 else 0:x + width]))
    return out


def normalised_list(the_list):
    """Every value divided by the sum of all (ref :240-242)."""
    scale = sum(the_list)
    return [value / scale for value in the_list]


def remove_small_values(the_list, relative_limit=0.1):
    """Values not above relative_limit * max become 0 (ref :244-253)."""
    floor = max(the_list) * relative_limit
    return [x if x > floor else 0 for x in the_list]


def mate_distribution_from_sam(samfile=sys.stdin, sample_size=10000):
    """Mate density from the |TLEN| of the first sample_size + 1 records with a non-zero TLEN (ref :255-273)."""
    sizes = []
    for line in samfile:
        if not line or line[0] == "@":
            continue
        insert_size = line.strip("\n").split()[8]
        if insert_size != "0":
            sizes.append(abs(int(insert_size)))
        if sample_size and len(sizes) > sample_size:
            break
    frequencies = Counter(sizes)
    histogram = [frequencies[i] if i in frequencies else 0 for i in range(0, max(sizes))]
    return normalised_list(remove_small_values(smoothed_list(histogram)))


def command_line_interface():
    parser = argparse.ArgumentParser(
        description="Caution: experimental.  Paired end mappability inferred from single end mappability: (1) --fasta "
                    "turns a genome into simulated reads, (2) map them, (3) --mapped_test_data turns the name sorted SAM "
                    "into a single end mappability wiggle, (4) --single_end_wiggle with --sam_for_sizes gives the paired "
                    "end wiggle.")
    parser.add_argument("--fasta", type=argparse.FileType("rt"), help="genome FASTA to cut into simulated reads (to stdout)")
    parser.add_argument("--readlength", type=int, default=100, help="the readlength to simulate")
    parser.add_argument("--mapped_test_data", type=argparse.FileType("rt"),
                        help="name sorted SAM of the mapped simulated reads; writes a fixed step wiggle to stdout")
    parser.add_argument("--single_end_wiggle", type=argparse.FileType("rt"), help="wiggle file of single end mappabilities")
    parser.add_argument("--sam_for_sizes", type=argparse.FileType("rt"), help="SAM file for the insert size distribution")
    parser.add_argument("--version", action="store_true", help="print version information and exit")
    args = parser.parse_args()
    if args.version:
        print(__version__)
        sys.exit()
    if (not args.fasta) and (not args.mapped_test_data) and (not args.single_end_wiggle):
        print("ERROR: Insufficient arguments provided")
        parser.print_help()
        sys.exit(1)
    return args


def main(args=None):
    if not args:
        args = command_line_interface()
    out = sys.stdout                       # looked up now: the functions' defaults were bound when the module was imported
    if args.fasta:
        simulate_reads(fastafile=args.fasta, readlength=args.readlength, outfile=out)
    elif args.mapped_test_data:
        single_end_mappability_from_sam(samfile=args.mapped_test_data, outfile=out)
    elif args.single_end_wiggle:
        if not args.sam_for_sizes:
            raise RuntimeError("You must provide a sam file to estimate the mate pair distance distribution")
        paired_end_mappability(wiggle=args.single_end_wiggle, mate_density=mate_distribution_from_sam(args.sam_for_sizes),
                               outfile=out)


if __name__ == "__main__":  # pragma: no cover
    main()
