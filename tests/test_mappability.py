"""The xenomappability drop-in (xenomapper_amd/mappability.py), driven like the reference's own
tests/test_mappability.py: host text/statistics functions on CPU, the GPU correlation bit for bit."""
import hashlib
import io
import os
from string import ascii_lowercase, ascii_uppercase

import numpy as np
import pytest

from tests import helpers as H

G6 = H.golden("g6_mappability.json")
DATA = os.path.join(H.GOLDEN, "ref_data")


def test_host_functions_like_the_reference_suite():
    from xenomapper_amd import mappability as mp
    assert list(mp.parse_fasta(io.StringIO('>firstsequence\nGACAT\n>secondsequence\nGNATCAT'))) == \
        [('firstsequence', 'GACAT'), ('secondsequence', 'GNATCAT')]
    out = io.StringIO()
    mp.simulate_reads(open(os.path.join(DATA, "test_from_EcoliK12DH10B.fasta")), readlength=150, outfile=out)
    assert hashlib.sha224(out.getvalue().encode('latin-1')).hexdigest() == G6["simulate_reads_150_sha224"] == \
        '227d299d0b0d2a348a41d6a5397668ca6a9ac5218ab4f85d68fd5c53'
    out = io.StringIO('')
    mp.simulate_reads(io.StringIO('>testing\n' + ascii_lowercase + ascii_uppercase), readlength=50, outfile=out)
    assert out.getvalue() == ('>testing_1\nabcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWX\n'
                              '>testing_2\nbcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXY\n'
                              '>testing_3\ncdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ\n')
    assert mp.make_blocklist(ascii_lowercase + ascii_uppercase, 10) == ['abcdefghij', 'klmnopqrst', 'uvwxyzABCD',
                                                                       'EFGHIJKLMN', 'OPQRSTUVWX', 'YZ']
    assert mp.slice_string_in_blocks(ascii_lowercase + ascii_uppercase, 10) == \
        'abcdefghij\nklmnopqrst\nuvwxyzABCD\nEFGHIJKLMN\nOPQRSTUVWX\nYZ\n'
    assert mp.format_fasta('fooGene', ascii_lowercase + ascii_uppercase, 10) == \
        '>fooGene\nabcdefghij\nklmnopqrst\nuvwxyzABCD\nEFGHIJKLMN\nOPQRSTUVWX\nYZ\n'
    sm = mp.smoothed_list([1, 2, 3] * 10 + [100] + [1, 2, 3] * 10)
    assert (sum(sm), max(sm), min(sm), len(sm)) == (219.8090909090909, 6.714285714285714, 1.9, 61)
    assert [repr(v) for v in sm] == G6["smoothed_list"]
    nl = mp.normalised_list([1, 2, 3] * 10 + [10] + [1, 2, 3] * 10)
    assert (max(nl), min(nl), nl[0] / nl[2], len(nl)) == (0.07692307692307693, 0.007692307692307693, 1 / 3, 61)
    rs = mp.remove_small_values(list(range(100)))
    assert rs[:10] == [0] * 10 and rs[10:] == list(range(100))[10:]
    with open(os.path.join(DATA, "paired_end_testdata_human.sam")) as fh:
        assert mp.mate_distribution_from_sam(samfile=fh, sample_size=3) == [0.0] * 164 + [0.047619047619047596] * 21 + [0.0] * 266
    with open(os.path.join(DATA, "paired_end_testdata_human.sam")) as fh:
        assert [v.hex() for v in mp.mate_distribution_from_sam(samfile=fh)] == G6["mate_density_default"]
    res = io.StringIO()
    with open(os.path.join(DATA, "test_from_EcoliK12DH10B_150reads.sam")) as fh:
        mp.single_end_mappability_from_sam(fh, outfile=res, chromosome_sizes={'Chromosome': 2752, 'A_Repeat': 991})
    assert hashlib.sha224(res.getvalue().encode('latin-1')).hexdigest() == 'e8e8557a16c05aaa436c2c0fe616450d82a955e0f6de8eb3d190cdf4'
    # wiggle round trip
    m = mp.Mappability(chromosome_sizes={'X': 30})
    m['X'] = [0, 1, 1] * 10
    buf = io.StringIO()
    m.to_wiggle(wigglefile=buf)
    assert buf.getvalue() == 'fixedStep\tchrom=X\tstart=1\tstep=1\n' + '0\n1\n1\n' * 10
    m2 = mp.Mappability(chromosome_sizes={})
    m2.from_wiggle(io.StringIO(buf.getvalue()), datatype=float)
    assert m2['X'] == [0.0, 1.0, 1.0] * 10 and m2.chromosome_sizes == {'X': 30}


@pytest.mark.gpu
def test_single_end_to_paired_like_the_reference_suite():
    from xenomapper_amd import mappability as mp
    mappable = mp.Mappability(chromosome_sizes={'Chromosome': 20})
    assert mappable['Chromosome'] == [0] * 20
    assert mappable.single_end_to_paired(mate_density=[0, 0.5, 0.5])['Chromosome'] == [0.0] * 20       # :115-118
    mappable = mp.Mappability(chromosome_sizes={'X': 30})
    mappable['X'] = [0, 1, 1] * 10
    assert mappable.single_end_to_paired(mate_density=[0, 0.4, 0.5, 0.1])['X'] == [0.9, 1.0, 1.0] * 10  # :134-135
    out = io.StringIO()
    dens = [0, 0, 0, 0, 0, 0, 0, 0, 0, 0.01, 0.45, 0.41, 0.13, 0, 0, 0, 0]
    wig = io.StringIO('fixedStep\tchrom=Chromosome\tstart=1\tstep=1\n' + '1\n0\n' * 50 +
                      'fixedStep\tchrom=Repeat\tstart=1\tstep=1\n' + '0\n' * 10)
    mp.paired_end_mappability(wig, dens, outfile=out, chromosome_sizes={'Chromosome': 100, 'X': 10})
    assert out.getvalue() == ('fixedStep\tchrom=Chromosome\tstart=1\tstep=1\n' + '1.0\n0.42\n' * 44 + '1.0\n0.01\n' +
                              '1.0\n0.0\n' * 5 + 'fixedStep\tchrom=X\tstart=1\tstep=1\n' + '0.0\n' * 10)   # :138-143
    # the fixture track with the fixture's own mate density: digest recorded from the reference
    out = io.StringIO()
    dens = [float.fromhex(v) for v in G6["mate_density_default"]]
    mp.paired_end_mappability(io.StringIO(G6["single_end_wiggle_text"]), dens, outfile=out,
                              chromosome_sizes={"Chromosome": 2752, "A_Repeat": 991})
    assert len(out.getvalue()) == G6["paired_wiggle_len"]
    assert hashlib.sha224(out.getvalue().encode("latin-1")).hexdigest() == G6["paired_wiggle_sha224"]


@pytest.mark.gpu
def test_correlate_golden_and_large_bit_exact():
    from xenomapper_amd import _ffi
    with _ffi.Context(0) as ctx:
        for case in G6["single_end_to_paired"]:
            track = np.array([float.fromhex(v) for v in case["track"]], dtype=np.float64)
            density = np.array([float.fromhex(v) for v in case["density"]], dtype=np.float64)
            got = ctx.mate_correlate(track, density)
            assert [float(v).hex() for v in got] == case["expect"]
        rng = np.random.default_rng(8)
        for n, m in ((3_000_001, 451), (100_000, 5000), (1, 7), (255, 2049), (70_000, 1)):
            track = np.where(rng.random(n) < 0.6, 1.0, rng.choice([0.0, 0.0, 0.25, 0.7], n))
            density = rng.random(m)
            density /= density.sum()
            got = ctx.mate_correlate(track, density)
            want = H.c_mate_correlate(track, density)
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (n, m)
        # non-finite taps: the loop must stop at the end of the track, not multiply padding by inf
        track = np.array([0.0, 0.5, 0.0, 0.25])
        density = np.array([0.5, float("inf"), float("nan")])
        got = ctx.mate_correlate(track, density)
        want = H.c_mate_correlate(track, density)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


G10 = H.golden("g10_mappability_cli.json")["cases"]


def _run_mappability_cli(case, tmp_path, monkeypatch, capsys):
    import hashlib
    import os
    from xenomapper_amd import mappability as mp
    data = os.path.join(H.GOLDEN, "ref_data")
    by = {c["name"]: c for c in G10}
    (tmp_path / "single.wig").write_text(by["single_end_wiggle"]["stdout"]["text"])
    argv = ["xenomappability"] + [a.replace("{D}", data).replace("{T}", str(tmp_path)) for a in case["argv"]]
    monkeypatch.setattr("sys.argv", argv)
    code, raised = 0, None
    try:
        mp.main()
    except SystemExit as exc:
        code = exc.code or 0
    except Exception as exc:
        code, raised = 1, "%s: %s" % (type(exc).__name__, exc)
    out = capsys.readouterr().out
    assert (code, raised) == (case["returncode"], case["exception"])
    if case["name"] != "no_arguments":                           # its stdout is the help text, which is this build's own
        assert (hashlib.sha224(out.encode("latin-1")).hexdigest(), len(out)) == (case["stdout"]["sha224"], case["stdout"]["len"])


@pytest.mark.parametrize("case", [c for c in G10 if c["name"] != "paired_end_wiggle"], ids=lambda c: c["name"])
def test_g10_command_line_host_steps(case, tmp_path, monkeypatch, capsys):
    """The companion tool's command line as the reference ran it (G10): the steps that are host text work."""
    _run_mappability_cli(case, tmp_path, monkeypatch, capsys)


@pytest.mark.gpu
def test_g10_command_line_paired_step(tmp_path, monkeypatch, capsys):
    """... and the step whose inner loop is the GPU correlation kernel."""
    _run_mappability_cli({c["name"]: c for c in G10}["paired_end_wiggle"], tmp_path, monkeypatch, capsys)
