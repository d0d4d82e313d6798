"""Reader for R's `.rda` / `.RData` files (XDR serialisation, format version 2 or 3: "RDX2"/"RDX3",
gzip / bzip2 / xz compressed or plain) -- the on-disk form of the reference's bundled data sets
(`data/h3n2_data.rda`, `data/hiv_titers.rda`, ...), so the input tables can be
fed to `topolow_amd.antigenic` without R.

A parser only: nothing in the file is evaluated.  Supported values are the ones data sets are made
of -- NULL, logical / integer / double / character vectors, factors, lists and data frames, with
attributes; compact integer / real sequences and wrapped vectors of the ALTREP framework.  Closures,
environments, promises, byte code and external pointers raise `RdaError`.

    objs = read_rda("hiv_titers.rda")            # {"hiv_titers": DataFrame(...)}
    rows = objs["hiv_titers"].rows()             # list of dicts, like csv.DictReader
"""
from __future__ import annotations

import bz2
import gzip
import lzma
import struct
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import numpy as np

NA_INTEGER = -2 ** 31
_NA_REAL_BITS = 0x7FF00000000007A2          # R's NA_real_: a NaN whose low word is 1954


class RdaError(ValueError):
    pass


@dataclass
class RObject:
    """A vector (or list) with its attributes."""
    value: Any
    attributes: Dict[str, Any] = field(default_factory=dict)


@dataclass
class DataFrame:
    names: List[str]
    columns: List[Any]                 # numpy arrays (numbers, NaN = NA) or lists of str / None
    row_names: Optional[List[Any]] = None

    def __len__(self) -> int:
        return len(self.columns[0]) if self.columns else 0

    def column(self, name: str):
        return self.columns[self.names.index(name)]

    def rows(self) -> List[Dict[str, Any]]:
        return [dict(zip(self.names, vals)) for vals in zip(*[list(c) for c in self.columns])]


# SEXP types of the serialisation format (R internals, serialize.c)
_NILSXP, _SYMSXP, _LISTSXP, _CLOSXP, _ENVSXP, _PROMSXP, _LANGSXP = 0, 1, 2, 3, 4, 5, 6
_CHARSXP, _LGLSXP, _INTSXP, _REALSXP, _CPLXSXP, _STRSXP, _VECSXP, _EXPRSXP, _RAWSXP = 9, 10, 13, 14, 15, 16, 19, 20, 24
_ALTREP, _ATTRLISTSXP, _ATTRLANGSXP = 238, 239, 240
_EMPTYENV, _BASEENV, _GLOBALENV, _UNBOUND, _MISSINGARG, _BASENAMESPACE = 242, 241, 253, 252, 251, 247
_NAMESPACESXP, _PACKAGESXP, _PERSISTSXP, _NILVALUE, _REFSXP = 249, 250, 248, 254, 255


class _Reader:
    def __init__(self, data: bytes):
        self.b = data
        self.o = 0
        self.refs: List[Any] = []

    def take(self, n: int) -> bytes:
        if self.o + n > len(self.b):
            raise RdaError("truncated file")
        out = self.b[self.o:self.o + n]
        self.o += n
        return out

    def int(self) -> int:
        return struct.unpack(">i", self.take(4))[0]

    def length(self) -> int:
        n = self.int()
        if n == -1:                                   # long vector: two more ints
            hi, lo = struct.unpack(">II", self.take(8))
            n = (hi << 32) | lo
        return n

    # --- one item -------------------------------------------------------------------------
    def item(self) -> Any:
        flags = self.int()
        t = flags & 0xFF
        has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
        if t == _NILVALUE or t == _NILSXP:
            return None
        if t in (_EMPTYENV, _BASEENV, _GLOBALENV, _UNBOUND, _MISSINGARG, _BASENAMESPACE):
            return None
        if t == _REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self.int()
            return self.refs[idx - 1]
        if t == _SYMSXP:
            name = self.item()                        # a CHARSXP
            self.refs.append(name)
            return name
        if t in (_NAMESPACESXP, _PACKAGESXP, _PERSISTSXP):
            self.item_strings_block()
            self.refs.append(None)
            return None
        if t in (_LISTSXP, _LANGSXP, _ATTRLISTSXP, _ATTRLANGSXP):
            # pairlist: walked iteratively (data frames of many columns nest deeply otherwise)
            out: List[tuple] = []
            while True:
                attrs = self.attributes() if (has_attr or t in (_ATTRLISTSXP, _ATTRLANGSXP)) else {}
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                del attrs
                flags = self.int()
                t = flags & 0xFF
                has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
                if t in (_NILVALUE, _NILSXP):
                    return out
                if t not in (_LISTSXP, _LANGSXP, _ATTRLISTSXP, _ATTRLANGSXP):
                    raise RdaError("malformed pairlist")
        if t == _CHARSXP:
            n = self.int()
            if n == -1:
                return None                           # NA_character_
            raw = self.take(n)
            enc = "latin-1" if flags & (1 << 14) else "utf-8"      # gp bit 2 (<< 12): latin1
            return raw.decode(enc, errors="replace")
        if t == _ALTREP:
            info, state, attr = self.item(), self.item(), self.item()
            return self.altrep(info, state, attr)
        if t in (_CLOSXP, _ENVSXP, _PROMSXP, 7, 8, 21, 22, 23, 25, 243, 244, 245, 246):
            raise RdaError(f"unsupported R object in data file (type {t})")
        # vectors
        if t == _LGLSXP or t == _INTSXP:
            n = self.length()
            v = np.frombuffer(self.take(4 * n), dtype=">i4").astype(np.int32)
        elif t == _REALSXP:
            n = self.length()
            v = np.frombuffer(self.take(8 * n), dtype=">f8").astype(np.float64)
        elif t == _CPLXSXP:
            n = self.length()
            v = np.frombuffer(self.take(16 * n), dtype=">c16").astype(np.complex128)
        elif t == _RAWSXP:
            n = self.length()
            v = np.frombuffer(self.take(n), dtype=np.uint8).copy()
        elif t == _STRSXP:
            n = self.length()
            v = [self.item() for _ in range(n)]
        elif t in (_VECSXP, _EXPRSXP):
            n = self.length()
            v = [self.item() for _ in range(n)]
        else:
            raise RdaError(f"unknown item type {t}")
        attrs = self.attributes() if has_attr else {}
        return self.finish(t, v, attrs)

    def item_strings_block(self):
        if self.int() != 0:
            raise RdaError("names in persistent name vectors not supported")
        n = self.int()
        return [self.item() for _ in range(n)]

    def attributes(self) -> Dict[str, Any]:
        pl = self.item()
        return {k: v for k, v in (pl or []) if k is not None}

    def altrep(self, info, state, attr) -> Any:
        cls = info[0][1] if info else None
        attrs = {k: v for k, v in (attr or []) if k is not None} if isinstance(attr, list) else {}
        if cls in ("compact_intseq", "compact_realseq"):
            n, start, step = (float(x) for x in _plain(state))
            seq = start + step * np.arange(int(n))
            v = seq.astype(np.int32) if cls == "compact_intseq" else seq.astype(np.float64)
            return self.finish(_INTSXP if cls == "compact_intseq" else _REALSXP, v, attrs)
        if cls in ("wrap_real", "wrap_integer", "wrap_logical", "wrap_string", "wrap_list", "wrap_complex",
                   "wrap_raw"):
            inner = state[0] if isinstance(state, list) else state
            if isinstance(inner, tuple):
                inner = inner[1]
            if attrs and isinstance(inner, RObject):
                inner.attributes.update(attrs)
            elif attrs:
                inner = RObject(inner, attrs)
            return inner
        if cls == "deferred_string":
            src = state[0][1] if isinstance(state, list) else state
            return [None if _is_na(x) else _r_format(x) for x in _plain(src)]
        raise RdaError(f"unsupported ALTREP class {cls!r}")

    def finish(self, t: int, v: Any, attrs: Dict[str, Any]) -> Any:
        """Vector + attributes -> python value: factors become strings, data frames DataFrame."""
        cls = _plain(attrs.get("class")) or []
        if t == _INTSXP and "factor" in cls:
            levels = _plain(attrs.get("levels")) or []
            return [None if k == NA_INTEGER else levels[k - 1] for k in v.tolist()]
        if t == _VECSXP and "data.frame" in cls:
            names = list(_plain(attrs.get("names")) or [])
            cols = [_column(c) for c in v]
            rn = _plain(attrs.get("row.names"))
            if isinstance(rn, np.ndarray):
                # compact form c(NA, -n): automatic row names 1..n
                rn = None if (rn.size == 2 and rn[0] == NA_INTEGER) else rn.tolist()
            return DataFrame(names, cols, rn)
        if t == _REALSXP:
            v = v.copy()
        if not attrs:
            return v
        return RObject(v, attrs)


def _plain(x: Any) -> Any:
    return x.value if isinstance(x, RObject) else x


def _is_na(x: Any) -> bool:
    return x is None or (isinstance(x, float) and x != x) or x == NA_INTEGER


def _r_format(x: Any) -> str:
    return str(int(x)) if float(x) == int(x) else repr(float(x))


def _column(c: Any) -> Any:
    c = _plain(c)
    if isinstance(c, np.ndarray) and c.dtype == np.int32:
        out = c.astype(np.float64)
        out[c == NA_INTEGER] = np.nan
        return out
    return c


def _decompress(raw: bytes) -> bytes:
    if raw[:6] == b"\xfd7zXZ\x00":
        return lzma.decompress(raw)
    if raw[:2] == b"\x1f\x8b":
        return gzip.decompress(raw)
    if raw[:3] == b"BZh":
        return bz2.decompress(raw)
    return raw


def read_rda(path: str) -> Dict[str, Any]:
    """Objects saved in an .rda / .RData file, by name."""
    with open(path, "rb") as fh:
        data = _decompress(fh.read())
    if data[:5] not in (b"RDX2\n", b"RDX3\n"):
        raise RdaError("not an R data file in XDR format (RDX2 / RDX3)")
    r = _Reader(data)
    r.o = 5
    if r.take(2) != b"X\n":
        raise RdaError("only the XDR (binary, big-endian) serialisation is supported")
    version = r.int()
    r.int()                      # R version that wrote the file
    r.int()                      # minimal R version that can read it
    if version == 3:
        r.take(r.int())          # native encoding of the writing session
    elif version != 2:
        raise RdaError(f"serialisation version {version} not supported")
    top = r.item()
    if not isinstance(top, list):
        raise RdaError("file does not hold a list of saved objects")
    return {name: value for name, value in top if name is not None}
