"""topolow_amd.rda: the .rda (RDX3/RDX2 XDR) reader.  The streams are built here, byte by byte, after
R's serialisation format (R Internals, section 'Serialization Formats'); when the reference checkout
is present its bundled data sets are parsed as well (data only, nothing executed)."""
import bz2
import gzip
import lzma
import os
import struct

import numpy as np
import pytest

from topolow_amd import antigenic, rda

NA_INT = -2 ** 31


def _i(x):
    return struct.pack(">i", x)


def _chars(s):
    if s is None:
        return _i(9) + _i(-1)
    b = s.encode("utf-8")
    return _i(9 | (1 << 15)) + _i(len(b)) + b          # CHARSXP, gp = UTF8 mask << 12 (bit 3)


def _strs(v, attrs=b""):
    return _i(16 | (0x200 if attrs else 0)) + _i(len(v)) + b"".join(_chars(s) for s in v) + attrs


def _ints(v, attrs=b"", obj=False):
    return _i(13 | (0x200 if attrs else 0) | (0x100 if obj else 0)) + _i(len(v)) + b"".join(_i(x) for x in v) + attrs


def _reals(v, attrs=b""):
    out = b""
    for x in v:
        out += struct.pack(">Q", 0x7FF00000000007A2) if x is None else struct.pack(">d", x)
    return _i(14 | (0x200 if attrs else 0)) + _i(len(v)) + out + attrs


class _Out:
    """Sequential emitter: symbols get their reference numbers in stream order, as R assigns them."""
    def __init__(self):
        self.b = b""
        self.seen = {}

    def raw(self, x):
        self.b += x
        return self

    def sym(self, name):
        if name in self.seen:
            return self.raw(_i(255 | (self.seen[name] << 8)))
        self.seen[name] = len(self.seen) + 1
        return self.raw(_i(1) + _chars(name))

    def pairlist(self, items):
        """items: (name, emit) with emit(out) writing the value."""
        for name, emit in items:
            self.raw(_i(2 | 0x400)).sym(name)
            emit(self)
        return self.raw(_i(254))


def _data_frame_stream(version=3):
    o = _Out()
    head = b"RDX%d\nX\n" % version + _i(version) + _i(0x040300) + _i(0x030500 if version == 3 else 0x020300)
    if version == 3:
        head += _i(5) + b"UTF-8"
    o.raw(head)

    def factor(out):
        out.raw(_i(13 | 0x200 | 0x100) + _i(4) + b"".join(_i(x) for x in [1, 2, NA_INT, 1]))
        out.pairlist([("levels", lambda q: q.raw(_strs(["HK68", "EN72"]))),
                      ("class", lambda q: q.raw(_strs(["factor"])))])

    def table(out):
        out.raw(_i(19 | 0x200 | 0x100) + _i(4))
        out.raw(_strs(["A/x/1", None, "B/y/2", "C"])).raw(_ints([1968, NA_INT, 1972, 1975]))
        out.raw(_reals([1.5, None, float("nan"), -2.0]))
        factor(out)
        out.pairlist([("names", lambda q: q.raw(_strs(["strain", "year", "value", "cluster"]))),
                      ("class", lambda q: q.raw(_strs(["data.frame"]))),
                      ("row.names", lambda q: q.raw(_ints([NA_INT, -4])))])

    def altseq(out):          # ALTREP: info pairlist (class symbol, package symbol, type), state, attributes
        out.raw(_i(238) + _i(2)).sym("compact_intseq").raw(_i(2)).sym("base").raw(_i(2) + _ints([13]) + _i(254))
        out.raw(_reals([5.0, 3.0, 1.0]) + _i(254))

    o.pairlist([("tbl", table), ("seq", altseq), ("vec", lambda q: q.raw(_reals([0.25, 4.0])))])
    return o.b


@pytest.mark.parametrize("pack", ["plain", "gzip", "bzip2", "xz"])
@pytest.mark.parametrize("version", [2, 3])
def test_reads_a_data_frame_with_na_factor_and_altrep(tmp_path, pack, version):
    raw = _data_frame_stream(version)
    data = {"plain": lambda b: b, "gzip": gzip.compress, "bzip2": bz2.compress, "xz": lzma.compress}[pack](raw)
    path = tmp_path / "t.rda"
    path.write_bytes(data)
    objs = rda.read_rda(str(path))
    assert set(objs) == {"tbl", "seq", "vec"}
    df = objs["tbl"]
    assert isinstance(df, rda.DataFrame) and df.names == ["strain", "year", "value", "cluster"] and len(df) == 4
    assert df.column("strain") == ["A/x/1", None, "B/y/2", "C"]
    assert np.array_equal(df.column("year"), [1968.0, np.nan, 1972.0, 1975.0], equal_nan=True)
    assert np.array_equal(df.column("value"), [1.5, np.nan, np.nan, -2.0], equal_nan=True)
    assert df.column("cluster") == ["HK68", "EN72", None, "HK68"] and df.row_names is None
    assert df.rows()[2] == {"strain": "B/y/2", "year": 1972.0, "value": pytest.approx(np.nan, nan_ok=True),
                            "cluster": None}
    assert np.array_equal(objs["seq"], [3, 4, 5, 6, 7]) and np.array_equal(objs["vec"], [0.25, 4.0])


def test_refuses_what_it_cannot_parse(tmp_path):
    p = tmp_path / "bad.rda"
    p.write_bytes(b"RDA2\nA\n")                     # ascii format
    with pytest.raises(rda.RdaError):
        rda.read_rda(str(p))
    raw = _data_frame_stream(3)
    p.write_bytes(raw[:len(raw) // 2])
    with pytest.raises(rda.RdaError):
        rda.read_rda(str(p))
    closure = _Out().raw(b"RDX3\nX\n" + _i(3) + _i(0x040300) + _i(0x030500) + _i(5) + b"UTF-8")
    closure.raw(_i(2 | 0x400)).sym("f").raw(_i(3) + _i(254) + _i(254) + _i(254) + _i(254))
    p.write_bytes(closure.b)
    with pytest.raises(rda.RdaError):
        rda.read_rda(str(p))


REF_DATA = "/root/reference/data"


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference checkout not present")
def test_reference_data_sets_parse_and_feed_the_pipeline():
    """h3n2_data = 3 542 rows (the same table as data-raw/Smith2004-data.csv),
    hiv_titers = 178 912 rows of Antibody / Virus / IC50 with censored values such as ">150"."""
    import csv
    h3 = rda.read_rda(os.path.join(REF_DATA, "h3n2_data.rda"))["h3n2_data"]
    assert len(h3) == 3542 and h3.names[:3] == ["virusStrain", "serumStrain", "titer"]
    raw = list(csv.DictReader(open("/root/reference/data-raw/Smith2004-data.csv", encoding="utf-8-sig")))
    rows = h3.rows()
    assert [r["titer"] for r in rows] == [r["titer"] for r in raw]
    assert [r["virusStrain"] for r in rows] == [r["virusStrain"] for r in raw]
    a = antigenic.process_antigenic_data(rows, "virusStrain", "serumStrain", "titer", is_similarity=True,
                                         base=2, scale_factor=10)[0]
    b = antigenic.process_antigenic_data(raw, "virusStrain", "serumStrain", "titer", is_similarity=True,
                                         base=2, scale_factor=10)[0]
    assert [r["distance"] for r in a] == [r["distance"] for r in b]
    hiv = rda.read_rda(os.path.join(REF_DATA, "hiv_titers.rda"))["hiv_titers"]
    assert len(hiv) == 178912 and hiv.names == ["Antibody", "Virus", "IC50"]
    ic50 = hiv.column("IC50")
    assert any(v.startswith(">") for v in ic50) and any(v.startswith("<") for v in ic50)
