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
    """read_ipt_13_14 (horizon_data.rs:323-347): rows 13/14 only for DE440+ with more than 400 constants."""
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
