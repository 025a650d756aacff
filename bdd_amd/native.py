"""ctypes front end of the C++ host-side input stage (include/bdd_ilp.h, bdd_amd/csrc/host/): .lp reader,
ILP -> QBDD conversion and long-BDD splitting as native code.

The pure-Python versions in ilp.py / bdd_collection.py stay the readable specification (and are what the tests
pin against oracle/_ref); these do the same work in C++ — the language of the reference's own input stage
(src/ILP/ILP_parser.cpp, src/bdd_conversion/bdd_preprocessor.cpp) — and are what large instances should use.
"""
from __future__ import annotations

import os
import ctypes as C

import numpy as np

from . import capi
from .bdd_collection import BddCollection
from .ilp import ILP, Constraint

_INEQ = {-1: "<=", 0: "=", 1: ">="}
_CODE = {"<=": -1, "=": 0, ">=": 1}


def _check(rc):
    if rc != 0:
        msg = capi.lib().bddilp_last_error().decode()
        raise (RuntimeError if rc == -2 else ValueError)(msg)


def _collection_from_handle(L, h) -> BddCollection:
    n, nb = int(L.bddilp_bdds_nr_instructions(h)), int(L.bddilp_bdds_nr_bdds(h))
    instr = np.zeros((0, 3), np.uint64)
    if n:
        buf = (C.c_uint64 * (3 * n)).from_address(L.bddilp_bdds_instructions(h))
        instr = np.frombuffer(buf, dtype=np.uint64).reshape(n, 3)
    delims = np.frombuffer((C.c_uint64 * (nb + 1)).from_address(L.bddilp_bdds_delimiters(h)), dtype=np.uint64)
    return BddCollection.from_arrays(instr, delims)


def parse_lp(text: str, fmt: str = "lp") -> ILP:
    """ilp.parse_lp / parse_opb / parse_lp_or_opb through the C++ readers (fmt: "lp", "opb" or "auto")."""
    L = capi.lib()
    h = C.c_void_p()
    fn = {"lp": L.bddilp_parse_lp, "opb": L.bddilp_parse_opb, "auto": L.bddilp_parse}[fmt]
    _check(fn(text.encode(), C.byref(h)))
    try:
        ilp = ILP()
        V = int(L.bddilp_nr_variables(h))
        for v in range(V):
            ilp.var(L.bddilp_variable_name(h, v).decode())
        obj = np.zeros(V)
        const = C.c_double(0)
        L.bddilp_objective(h, obj.ctypes.data_as(C.c_void_p), C.byref(const))
        ilp.objective = obj.tolist()
        ilp.constant = const.value
        for c in range(int(L.bddilp_nr_constraints(h))):
            n = int(L.bddilp_constraint_size(h, c))
            co, vs = np.zeros(n, np.int64), np.zeros(n, np.uint64)
            ineq, rhs = C.c_int(0), C.c_int64(0)
            _check(L.bddilp_constraint(h, c, co.ctypes.data_as(C.c_void_p), vs.ctypes.data_as(C.c_void_p), C.byref(ineq), C.byref(rhs)))
            ilp.constraints.append(Constraint(co.tolist(), [int(x) for x in vs], _INEQ[ineq.value], int(rhs.value),
                                              L.bddilp_constraint_name(h, c).decode()))
        return ilp
    finally:
        L.bddilp_destroy(h)


def lp_to_bdd_collection(text: str, split: bool = False, split_length: int = 0, normalize: bool = False, with_implication_bdd: bool = False):
    """.lp text -> (ILP, BddCollection) entirely in C++ (parse, optional normalisation, conversion, optional splitting)."""
    L = capi.lib()
    h = C.c_void_p()
    _check(L.bddilp_parse_lp(text.encode(), C.byref(h)))
    try:
        if normalize:
            L.bddilp_normalize(h)
        b = C.c_void_p()
        _check(L.bddilp_to_bdds(h, (2 if with_implication_bdd else 1) if split else 0, int(split_length or 0), C.byref(b)))
        try:
            col = _collection_from_handle(L, b)
        finally:
            L.bddilp_bdds_destroy(b)
    finally:
        L.bddilp_destroy(h)
    return col


def rows_to_bdd_collection(rows, split_length: int = None, nr_variables: int = 0, with_implication_bdd: bool = False) -> BddCollection:
    """rows: iterable of (coefficients, variables, ineq, rhs); trivially true rows are skipped, an infeasible one raises."""
    L = capi.lib()
    b = C.c_void_p()
    _check(L.bddilp_bdds_create(C.byref(b)))
    try:
        for co, vs, ineq, rhs in rows:
            co = np.ascontiguousarray(co, dtype=np.int64)
            vs = np.ascontiguousarray(vs, dtype=np.uint64)
            st = C.c_int(0)
            _check(L.bddilp_bdds_add_row(b, co.ctypes.data_as(C.c_void_p), vs.ctypes.data_as(C.c_void_p), co.size, _CODE[ineq], int(rhs), C.byref(st)))
            if st.value == 2:
                raise RuntimeError("problem is infeasible")
        if split_length is not None:
            ns, nv = C.c_uint64(0), C.c_uint64(0)
            _check(L.bddilp_bdds_split(b, max(int(nr_variables), int(L.bddilp_bdds_nr_variables(b))), int(split_length),
                                       int(with_implication_bdd), C.byref(ns), C.byref(nv)))
        return _collection_from_handle(L, b)
    finally:
        L.bddilp_bdds_destroy(b)


def write_bdd_lp(col: BddCollection, costs, path: str) -> None:
    """csrc/host: bdd_store::write_bdd_lp — the arc-flow LP of a BDD collection ("export bdd lp" of the driver)"""
    L = capi.lib()
    instr = np.ascontiguousarray(col.instr, dtype=np.uint64)
    delims = np.ascontiguousarray(col.delims, dtype=np.uint64)
    c = np.ascontiguousarray(costs, dtype=np.float64)
    _check(L.bddilp_write_bdd_lp(instr.ctypes.data_as(C.c_void_p), delims.ctypes.data_as(C.c_void_p), col.nr_bdds(),
                                 c.ctypes.data_as(C.c_void_p), c.size, os.fsencode(path)))


def export_graphviz(col: BddCollection, bdd_nr: int, path: str) -> None:
    """csrc/host: bdd_store::export_graphviz — one BDD as a Graphviz digraph ("export bdd graph" of the driver)"""
    L = capi.lib()
    instr = np.ascontiguousarray(col.instr, dtype=np.uint64)
    delims = np.ascontiguousarray(col.delims, dtype=np.uint64)
    _check(L.bddilp_export_graphviz(instr.ctypes.data_as(C.c_void_p), delims.ctypes.data_as(C.c_void_p), col.nr_bdds(), int(bdd_nr), os.fsencode(path)))
