"""the legacy-DE binary reader (`outfit_hip_ephem_open`, host-only) against the numpy
restatement of the reference's reader (oracle/de_reader.py) on synthetic DE-layout files, and both against the
reference's own header KAT (jpl_ephem/horizon/horizon_data.rs:863-896: DE440 IPT -> recsize 8144)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import de_writer
import oracle_lib as O
import outfit_amd
from outfit_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import de_reader  # noqa: E402

DE440_IPT = [[3, 14, 4], [171, 10, 2], [231, 13, 2], [309, 11, 1], [342, 8, 1], [366, 7, 1], [387, 6, 1],
             [405, 6, 1], [423, 6, 1], [441, 13, 8], [753, 11, 2], [819, 10, 4], [899, 10, 4], [1019, 0, 0],
             [1019, 0, 0]]


def test_reference_header_kat_recsize():
    """horizon_data.rs:863-896: the DE440 IPT gives 8144-byte records; the synthetic table uses the DE440
    (n_coeffs, n_subintervals) of EMB / Moon / Sun."""
    assert de_reader.compute_recsize(DE440_IPT) == 8144
    assert de_writer.recsize_of(DE440_IPT) == 8144
    assert (synth.NCOEF, synth.NSUB) == ((13, 13, 11), (2, 8, 2)) or \
        (tuple(synth.NCOEF), tuple(synth.NSUB)) == ((13, 13, 11), (2, 8, 2))


@pytest.fixture(scope="module")
def table():
    return synth.make_ephemeris(n_blocks=5)


def test_reader_matches_oracle_and_source_table(tmp_path, table):
    path = tmp_path / "synthetic.440"
    assert de_writer.write_de(path, table) == 8144
    ora = de_reader.read(path)
    eph = outfit_amd.Ephemeris.from_file(path)
    h = eph.header
    # header: what HorizonHeader holds (horizon_data.rs:130-160)
    assert h["jpl_version"] == "DE440" and h["ipt"] == DE440_IPT and h["recsize"] == 8144
    assert (h["start_period"], h["end_period"], h["period_lenght"]) == (table["jd_start"], table["jd_end"], 32.0)
    assert h["earth_moon_mass_ratio"] == table["emrat"] and h["n_blocks"] == 5 and h["ncon"] == 645
    assert np.array_equal(np.array(h["ipt"]), ora["ipt"]) and ora["recsize"] == 8144 and ora["numde"] == 440
    # table: product == oracle == the table the file was written from, bit for bit
    for k in ("jd_start", "jd_end", "block_days", "emrat", "n_blocks"):
        assert getattr(eph, k) == ora[k] == table[k], k
    assert eph.ncoef == ora["ncoef"] == tuple(table["ncoef"]) and eph.nsub == ora["nsub"] == tuple(table["nsub"])
    assert eph.coef.tobytes() == ora["coef"].tobytes() == np.asarray(table["coef"]).tobytes()
    # ... and it drives the Chebyshev evaluation exactly like the in-memory table
    direct = outfit_amd.Ephemeris.from_dict(table)
    for mjd in (59990.25, 60050.0, 60100.123456):
        a, b = O.D3(), O.D3()
        assert O.lib().orc_earth_ephemeris(C.byref(eph.as_c()), mjd, a) == 0
        assert O.lib().orc_earth_ephemeris(C.byref(direct.as_c()), mjd, b) == 0
        assert list(a) == list(b)


def test_pre_de440_layout_has_no_rows_13_14(tmp_path, table):
    """read_ipt_13_14 (horizon_data.rs:382-405): rows 13/14 only for DE440+ with more than 400 constants."""
    ipt = [r[:] for r in DE440_IPT]
    ipt[13], ipt[14] = [0, 0, 0], [0, 0, 0]
    for numde, ncon in ((430, 572), (440, 400)):
        path = tmp_path / f"synthetic.{numde}.{ncon}"
        de_writer.write_de(path, table, ipt=ipt, numde=numde, ncon=ncon)
        eph = outfit_amd.Ephemeris.from_file(path)
        assert eph.header["jpl_version"] == f"DE{numde}" and eph.header["ipt"][13:] == [[0, 0, 0], [0, 0, 0]]
        assert eph.coef.tobytes() == de_reader.read(path)["coef"].tobytes()


def test_rows_13_14_change_the_record_size(tmp_path, table):
    """DE440t-style: a TT-TDB row (14) lengthens every record; the reader must follow compute_recsize."""
    ipt = [r[:] for r in DE440_IPT]
    ipt[14] = [1019, 13, 8]
    path = tmp_path / "synthetic.440t"
    rs = de_writer.write_de(path, table, ipt=ipt)
    assert rs == 8144 + 4 * 2 * 13 * 8
    eph = outfit_amd.Ephemeris.from_file(path)
    assert eph.header["recsize"] == rs and eph.header["ipt"][14] == [1019, 13, 8]
    assert eph.coef.tobytes() == np.asarray(table["coef"]).tobytes() == de_reader.read(path)["coef"].tobytes()


def test_reader_errors(tmp_path, table):
    with pytest.raises(outfit_amd.EphemerisFileError):
        outfit_amd.Ephemeris.from_file(tmp_path / "missing.440")
    short = tmp_path / "short.440"
    short.write_bytes(b"\0" * 1000)
    with pytest.raises(outfit_amd.EphemerisFileError):
        outfit_amd.Ephemeris.from_file(short)
    good = tmp_path / "good.440"
    de_writer.write_de(good, table)
    data = good.read_bytes()
    cut = tmp_path / "cut.440"
    cut.write_bytes(data[: 2 * 8144 + 100])          # header records only: no complete data record
    with pytest.raises(outfit_amd.EphemerisFileError):
        outfit_amd.Ephemeris.from_file(cut)
    partial = tmp_path / "partial.440"
    partial.write_bytes(data[: 2 * 8144 + 3 * 8144 + 17])   # trailing partial record is ignored (len / recsize)
    assert outfit_amd.Ephemeris.from_file(partial).n_blocks == 3 == de_reader.read(partial)["n_blocks"]
    junk = tmp_path / "junk.440"
    junk.write_bytes(bytes(range(256)) * 64)          # IPT rows point outside any plausible record
    with pytest.raises(outfit_amd.EphemerisFileError):
        outfit_amd.Ephemeris.from_file(junk)


def test_body_records_of_any_ipt_row(tmp_path, table):
    """`outfit_hip_ephem_body_record` (extract_body_records, horizon_data.rs:468-491) against the numpy restatement:
    every body of the file, first / last block; the three bodies of the path equal the table's segments; the
    restated `interpolate` (horizon_records.rs:204-298) reproduces the C oracle's Earth on the same file and its
    velocity / acceleration are the derivatives of its position (central differences)."""
    path = tmp_path / "bodies.440"
    de_writer.write_de(path, table)
    hdr = de_reader.read(path)
    for row in range(13):
        for block in (0, table["n_blocks"] - 1):
            a, ja = outfit_amd.Ephemeris.body_records(path, row, block)
            b, jb = de_reader.body_records(path, row, block)
            assert a.shape == b.shape == (DE440_IPT[row][2], 2 if row == 11 else 3, DE440_IPT[row][1])
            assert a.tobytes() == b.tobytes() and ja == jb == (table["jd_start"] + 32.0 * block, table["jd_start"] + 32.0 * block + 32.0)
    per = sum(3 * c * s for c, s in zip(table["ncoef"], table["nsub"]))
    coef = np.asarray(table["coef"]).reshape(table["n_blocks"], per)
    o = 0
    for b, row in enumerate((2, 9, 10)):
        n = 3 * table["ncoef"][b] * table["nsub"][b]
        assert outfit_amd.Ephemeris.body_records(path, row, 2)[0].reshape(-1).tobytes() == coef[2, o:o + n].tobytes()
        o += n
    with pytest.raises(outfit_amd.InvalidBatch):
        outfit_amd.Ephemeris.body_records(path, 13, 0)           # row 13 holds no coefficients in DE440
    with pytest.raises(outfit_amd.InvalidBatch):
        outfit_amd.Ephemeris.body_records(path, 2, table["n_blocks"])
    eph = outfit_amd.Ephemeris.from_file(path)
    for mjd in (59990.25, 60050.0, 60100.123456):
        want = O.D3()
        assert O.lib().orc_earth_ephemeris(C.byref(eph.as_c()), mjd, want) == 0
        pos, vel, acc = de_reader.ephemeris(path, hdr, 2, 10, mjd)
        assert pos == list(want)
        h = 1e-3
        p1, v1, _ = de_reader.ephemeris(path, hdr, 2, 10, mjd + h)
        p0, v0, _ = de_reader.ephemeris(path, hdr, 2, 10, mjd - h)
        assert np.allclose((np.array(p1) - np.array(p0)) / (2 * h), vel, rtol=1e-6, atol=1e-12)
        assert np.allclose((np.array(v1) - np.array(v0)) / (2 * h), acc, rtol=1e-5, atol=1e-12)


@pytest.mark.gpu
def test_reader_suite_runs_in_gputest(tmp_path):
    """The host-only reader checks above, collected under -m gpu as well so that they execute on the GPU box's
    toolchain and filesystem in the round-end run (they need no device)."""
    table = synth.make_ephemeris(n_blocks=5)
    test_reference_header_kat_recsize()
    for i, fn in enumerate((test_reader_matches_oracle_and_source_table, test_pre_de440_layout_has_no_rows_13_14,
                            test_rows_13_14_change_the_record_size, test_reader_errors, test_body_records_of_any_ipt_row)):
        d = tmp_path / f"case{i}"
        d.mkdir()
        fn(d, table)


@pytest.mark.gpu
def test_iod_with_file_ephemeris_equals_in_memory_table(tmp_path):
    table = synth.make_ephemeris()
    path = tmp_path / "bench.440"
    de_writer.write_de(path, table)
    ds = synth.make_dataset(200, ephem=table, topocentric=True)
    batch = outfit_amd.ObsBatch.from_dict(ds)
    params = synth.bench_params(outfit_amd.IODParams())
    a = outfit_amd.fit_full_iod_records(batch, outfit_amd.Ephemeris.from_file(path), params, 1)
    b = outfit_amd.fit_full_iod_records(batch, outfit_amd.Ephemeris.from_dict(table), params, 1)
    assert a.tobytes() == b.tobytes() and (a["status"] == 0).sum() > 150
