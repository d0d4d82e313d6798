"""Input construction for the relaxation path: long titer / IC50
tables -> the dissimilarity matrix `euclidean_embedding()` takes.

Mirrors two functions of the reference's R/data_preprocessing.R:
  * `process_antigenic_data()`  (:488-684) -- log transform, Smith's per-serum distance
    `max(log titer of the serum) - log titer`, threshold sign inversion ("<" titer -> ">" distance,
    :601-609), averaging of repeated measurements with the threshold sign re-applied (:575-583,
    :633-647);
  * `titers_list_to_matrix()`   (:743-844) -- "V/"/"S/" name prefixes, optional ordering by year,
    symmetric fill, zero diagonal.
Only the pieces the embedding needs are reproduced (the `raw_value` column and metadata joins
that do not influence the matrix are carried along but not otherwise used).  Numbers that R
pastes into strings go through R's default 15-significant-digit formatting; so do they here.
"""
from __future__ import annotations

import math
import re
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .core import RMatrix


def _r_num_str(x: float) -> str:
    """as.character(<double>) / paste0(<double>): 15 significant digits."""
    if x == int(x) and abs(x) < 1e15:
        return str(int(x))
    return repr(float(f"{x:.15g}"))


def _remove_sign(s: str) -> float:
    return float(re.sub(r"[<>]", "", s))


def _reapply_sign(values: Sequence[str], avg: float) -> str:
    """R/data_preprocessing.R:575-583."""
    if any(("<" in v) or (">" in v) for v in values):
        sign = "<" if any("<" in v for v in values) else ">"
        return sign + _r_num_str(avg)
    return _r_num_str(avg)


def process_antigenic_data(rows: Sequence[Dict[str, object]], antigen_col: str, serum_col: str,
                           value_col: str, is_similarity: bool = False,
                           base: Optional[float] = None, scale_factor: float = 1.0,
                           antigen_year_col: Optional[str] = "virusYear",
                           serum_year_col: Optional[str] = "serumYear"):
    """rows: list of dicts (e.g. csv.DictReader).  Returns (long_rows, RMatrix) like the
    reference's list(long=, matrix=)."""
    if base is None:
        base = 2.0 if is_similarity else math.e
    clean = []
    for r in rows:
        v = r.get(value_col)
        if v is None:
            continue
        v = str(v).strip()
        if v == "" or v == "NA" or not re.match(r"^[0-9<>]", v):
            continue
        rr = dict(r)
        rr[value_col] = v
        clean.append(rr)
    if not clean:
        raise ValueError("No valid measurements remaining after cleaning")
    has_ay = antigen_year_col is not None and antigen_year_col in clean[0]
    has_sy = serum_year_col is not None and serum_year_col in clean[0]

    def logb(x):
        return math.log(x) / math.log(base)

    # per-row distance strings (thresholds keep a prefix)
    if is_similarity:
        processed = []
        prefix = []
        for r in clean:
            v = r[value_col]
            p = v[0] if v[0] in "<>" else ""
            num = float(v[1:]) if p else float(v)
            processed.append(logb(num / scale_factor))
            prefix.append(p)
        max_by_serum: Dict[object, float] = {}
        for r, pv in zip(clean, processed):
            s = r[serum_col]
            max_by_serum[s] = max(max_by_serum.get(s, -math.inf), pv)
        dist_str = []
        for r, pv, p in zip(clean, processed, prefix):
            d = max_by_serum[r[serum_col]] - pv
            if p == "<":
                dist_str.append(">" + _r_num_str(d))      # "<" titer -> ">" distance
            elif p == ">":
                dist_str.append("<" + _r_num_str(d))
            else:
                dist_str.append(_r_num_str(d))
    else:
        dist_str = []
        for r in clean:
            v = r[value_col]
            p = v[0] if v[0] in "<>" else ""
            num = float(v[1:]) if p else float(v)
            dist_str.append(p + _r_num_str(logb(1.0 + num)))

    # combine repeated (antigen, serum) measurements; group order = sorted keys (dplyr::group_by)
    groups: "OrderedDict[Tuple[str, str], List[int]]" = OrderedDict()
    for q, r in enumerate(clean):
        groups.setdefault((str(r[antigen_col]), str(r[serum_col])), []).append(q)
    long_rows = []
    for key in sorted(groups):
        idx = groups[key]
        ds = [dist_str[q] for q in idx]
        avg = float(np.mean([_remove_sign(d) for d in ds]))
        row = {antigen_col: key[0], serum_col: key[1], "distance": _reapply_sign(ds, avg)}
        if has_ay:
            row[antigen_year_col] = clean[idx[0]][antigen_year_col]
        if has_sy:
            row[serum_year_col] = clean[idx[0]][serum_year_col]
        long_rows.append(row)
    if has_ay:
        long_rows.sort(key=lambda r: float(r[antigen_year_col]))   # stable, like order()
    matrix = titers_list_to_matrix(long_rows, antigen_col, antigen_year_col if has_ay else None,
                                   serum_col, serum_year_col if has_sy else None, "distance",
                                   rc=False, sort=has_ay or has_sy)
    return long_rows, matrix


def titers_list_to_matrix(rows: Sequence[Dict[str, object]], chnames: str, chorder: Optional[str],
                          rnames: str, rorder: Optional[str], values_column: str, rc: bool = False,
                          sort: bool = False) -> RMatrix:
    """R/data_preprocessing.R:743-844.  Names sort in code-point order (R sorts in the session
    locale; the embedding re-orders by mean dissimilarity anyway unless preserve_order=TRUE)."""
    ch = [(("" if rc else "V/") + str(r[chnames])) for r in rows]
    rf = [(("" if rc else "S/") + str(r[rnames])) for r in rows]
    all_points = sorted(set(ch) | set(rf))
    if sort:
        ranks = []
        for name in all_points:
            yr = 0.0
            if chorder is not None:
                ys = [float(r[chorder]) for r, c in zip(rows, ch) if c == name]
                if ys:
                    yr = min(ys)
            if yr == 0 and rorder is not None:
                ys = [float(r[rorder]) for r, c in zip(rows, rf) if c == name]
                if ys:
                    yr = min(ys)
            ranks.append(yr)
        order = np.argsort(np.asarray(ranks), kind="stable")
        all_points = [all_points[q] for q in order]
    pos = {nm: q for q, nm in enumerate(all_points)}
    n = len(all_points)
    m = np.full((n, n), None, dtype=object)
    for r, c_, f_ in zip(rows, ch, rf):
        val = r[values_column]
        val = val if isinstance(val, str) else _r_num_str(float(val))
        a, b = pos[c_], pos[f_]
        m[a, b] = val
        m[b, a] = val
    for q in range(n):
        m[q, q] = "0"
    return RMatrix(m, all_points)
