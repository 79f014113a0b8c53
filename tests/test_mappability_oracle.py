"""Pins oracle/mappability_oracle.py (and the C correlate) to the reference's mappability module: the digests and
canned values of xenomapper/tests/test_mappability.py quoted literally, plus golden vectors recorded from the
imported reference (tests/golden/g6_mappability.json).  CPU only."""
import hashlib
import io
import os

import numpy as np

from tests import helpers as H
from oracle import mappability_oracle as MO

G6 = H.golden("g6_mappability.json")
DATA = os.path.join(H.GOLDEN, "ref_data")


def test_simulate_reads_digest():
    with open(os.path.join(DATA, "test_from_EcoliK12DH10B.fasta")) as fh:
        text = MO.simulated_reads_text(fh, readlength=150)
    # xenomapper/tests/test_mappability.py:40
    assert hashlib.sha224(text.encode("latin-1")).hexdigest() == "227d299d0b0d2a348a41d6a5397668ca6a9ac5218ab4f85d68fd5c53"
    from string import ascii_lowercase, ascii_uppercase
    canned = ('>testing_1\nabcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWX\n'
              '>testing_2\nbcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXY\n'
              '>testing_3\ncdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ\n')                 # :42-48
    assert MO.simulated_reads_text(io.StringIO('>testing\n' + ascii_lowercase + ascii_uppercase), 50) == canned
    assert list(MO.read_fasta(io.StringIO('>firstsequence\nGACAT\n>secondsequence\nGNATCAT'))) == \
        [('firstsequence', 'GACAT'), ('secondsequence', 'GNATCAT')]                                 # :32-33
    assert MO.blocks(ascii_lowercase + ascii_uppercase, 10) == ['abcdefghij', 'klmnopqrst', 'uvwxyzABCD', 'EFGHIJKLMN',
                                                               'OPQRSTUVWX', 'YZ']                 # :53-56
    assert MO.fasta_text('fooGene', ascii_lowercase + ascii_uppercase, 10) == \
        '>fooGene\nabcdefghij\nklmnopqrst\nuvwxyzABCD\nEFGHIJKLMN\nOPQRSTUVWX\nYZ\n'              # :66-68


def _sam_body(path):
    with open(path) as fh:
        return [l for l in fh if not l.startswith("@")]


def test_single_end_wiggle_digest():
    tracks = {"Chromosome": [0] * 2752, "A_Repeat": [0] * 991}
    tracks.update(MO.single_end_track_from_sam(_sam_body(os.path.join(DATA, "test_from_EcoliK12DH10B_150reads.sam"))))
    text = MO.wiggle_text(tracks)
    # xenomapper/tests/test_mappability.py:109
    assert hashlib.sha224(text.encode("latin-1")).hexdigest() == "e8e8557a16c05aaa436c2c0fe616450d82a955e0f6de8eb3d190cdf4"
    assert text == G6["single_end_wiggle_text"]


def test_mate_density():
    with open(os.path.join(DATA, "paired_end_testdata_human.sam")) as fh:
        got = MO.mate_density_from_sam(fh, sample_size=3)
    assert got == [0.0] * 164 + [0.047619047619047596] * 21 + [0.0] * 266          # test_mappability.py:101
    assert [v.hex() for v in got] == G6["mate_density_sample3"]
    with open(os.path.join(DATA, "paired_end_testdata_human.sam")) as fh:
        assert [v.hex() for v in MO.mate_density_from_sam(fh)] == G6["mate_density_default"]
    assert [repr(v) for v in MO.smoothed([1, 2, 3] * 10 + [100] + [1, 2, 3] * 10)] == G6["smoothed_list"]


def test_correlate_reference_rows():
    # test_mappability.py:115-118, :128-135
    assert MO.correlate_track([0] * 20, [0, 0.5, 0.5]) == [0.0] * 20
    assert MO.correlate_track([0, 1, 1] * 10, [0, 0.4, 0.5, 0.1]) == [0.9, 1.0, 1.0] * 10
    # :138-143
    dens = [0, 0, 0, 0, 0, 0, 0, 0, 0, 0.01, 0.45, 0.41, 0.13, 0, 0, 0, 0]
    wig = io.StringIO('fixedStep\tchrom=Chromosome\tstart=1\tstep=1\n' + '1\n0\n' * 50 +
                      'fixedStep\tchrom=Repeat\tstart=1\tstep=1\n' + '0\n' * 10)
    want = ('fixedStep\tchrom=Chromosome\tstart=1\tstep=1\n' + '1.0\n0.42\n' * 44 + '1.0\n0.01\n' + '1.0\n0.0\n' * 5 +
            'fixedStep\tchrom=X\tstart=1\tstep=1\n' + '0.0\n' * 10)
    assert MO.paired_wiggle_text(wig, dens, {'Chromosome': 100, 'X': 10}) == want


def test_correlate_golden_python_and_c():
    for case in G6["single_end_to_paired"]:
        track = [float.fromhex(v) for v in case["track"]]
        if case["track_is_int"]:
            track = [int(v) for v in track]
        density = [float.fromhex(v) for v in case["density"]]
        want = [float.fromhex(v) for v in case["expect"]]
        got = MO.correlate_track(track, density)
        assert [float(v).hex() for v in got] == case["expect"]
        c = H.c_mate_correlate(np.array(track, dtype=np.float64), np.array(density, dtype=np.float64))
        assert [float(v).hex() for v in c] == [float(v).hex() for v in want]


def test_paired_wiggle_of_the_fixture_track():
    dens = [float.fromhex(v) for v in G6["mate_density_default"]]
    text = MO.paired_wiggle_text(io.StringIO(G6["single_end_wiggle_text"]), dens, {"Chromosome": 2752, "A_Repeat": 991})
    assert len(text) == G6["paired_wiggle_len"]
    assert hashlib.sha224(text.encode("latin-1")).hexdigest() == G6["paired_wiggle_sha224"]
