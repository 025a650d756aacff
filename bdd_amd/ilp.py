"""0/1 ILP model, a reader for the .lp subset the reference parses, and ILP -> QBDD conversion.

Mirrors (host-side, Python) the input stage either side of the hot path:
  * ``ILP`` / ``parse_lp``: reference ``LPMP::ILP_input`` + the PEGTL grammar in
    src/ILP/ILP_parser.cpp:24-140 (``Minimize``, objective terms, ``Subject To``, optionally
    named rows, ``Bounds`` = variable fixations applied as ``ILP_input::reduce`` does, ``Binaries`` /
    ``Generals`` lists skipped, ``End``; products of variables in a row are refused).  Variable indices are
    assigned in order of first appearance, objective first (ILP_parser.cpp:246-254, :316-327).
  * ``to_bdd_collection``: ``bdd_preprocessor::add_ilp`` (src/bdd_conversion/bdd_preprocessor.cpp:123-336)
    for linear rows: simplex rows -> ``simplex_constraint`` (:172-188), everything else ->
    canonical QBDD over the row's variables in the order they are written (:190-226).
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import List

import numpy as np

from .bdd_collection import BddCollection


@dataclass
class Constraint:
    coefficients: List[int]
    variables: List[int]
    ineq: str  # "<=", "=", ">="
    rhs: int
    name: str = ""

    def is_simplex(self) -> bool:
        # ILP_input::constraint::is_simplex, src/ILP/ILP_input.cpp:81-93
        return self.ineq == "=" and self.rhs != 0 and all(c == self.rhs for c in self.coefficients)


@dataclass
class ILP:
    var_names: List[str] = field(default_factory=list)
    objective: List[float] = field(default_factory=list)
    constraints: List[Constraint] = field(default_factory=list)
    constant: float = 0.0
    _index: dict = field(default_factory=dict, repr=False)

    def var(self, name: str) -> int:
        i = self._index.get(name)
        if i is None:
            i = len(self.var_names)
            self._index[name] = i
            self.var_names.append(name)
            self.objective.append(0.0)
        return i

    def nr_variables(self) -> int:
        return len(self.var_names)

    def add_constraint(self, terms, ineq, rhs, name=""):
        """terms: iterable of (coeff, var_name_or_index)."""
        cs, vs = [], []
        for c, v in terms:
            cs.append(int(c))
            vs.append(self.var(v) if isinstance(v, str) else int(v))
        self.constraints.append(Constraint(cs, vs, ineq, int(rhs), name))

    def evaluate(self, x) -> float:
        return float(np.dot(self.objective, x)) + self.constant

    def feasible(self, x) -> bool:
        for c in self.constraints:
            s = sum(a * int(x[v]) for a, v in zip(c.coefficients, c.variables))
            ok = s <= c.rhs if c.ineq == "<=" else (s == c.rhs if c.ineq == "=" else s >= c.rhs)
            if not ok:
                return False
        return True

    def reduce(self, zeros, ones) -> "ILP":
        """ILP_input::reduce (src/ILP/ILP_input.cpp:508-591): the model without the fixed variables.  A variable
        fixed to 1 moves its objective coefficient into the constant and its row coefficients to the right-hand
        sides; a row that loses every term is checked (`0 <rel> rhs`) and dropped; the others keep their order."""
        zeros, ones = set(zeros), set(ones)
        r = ILP()
        r.constant = self.constant
        vmap = {}
        for i, name in enumerate(self.var_names):
            if i in zeros and i in ones:
                raise ValueError(f"variable '{name}' is fixed to 0 and to 1")
            if i in ones:
                r.constant += self.objective[i]
            elif i not in zeros:
                vmap[i] = r.var(name)
                r.objective[vmap[i]] = self.objective[i]
        for c in self.constraints:
            cs, vs, rhs = [], [], c.rhs
            for a, v in zip(c.coefficients, c.variables):
                if v in zeros:
                    continue
                if v in ones:
                    rhs -= a
                    continue
                cs.append(a)
                vs.append(vmap[v])
            if vs:
                r.constraints.append(Constraint(cs, vs, c.ineq, rhs, c.name))
            elif not (0 <= rhs if c.ineq == "<=" else (0 == rhs if c.ineq == "=" else 0 >= rhs)):
                raise ValueError(f"reduced model not feasible due to violated constraint {c.name}")
        return r

    def write_lp(self) -> str:
        def term(c, name, first):
            sign = "-" if c < 0 else ("" if first else "+")
            mag = abs(c)
            mag_s = repr(mag) if isinstance(mag, float) and mag != int(mag) else str(int(mag))
            return f"{sign} {mag_s} {name}".strip()

        lines = ["Minimize"]
        obj = [term(c, n, i == 0) for i, (c, n) in enumerate(zip(self.objective, self.var_names))]
        for i in range(0, len(obj), 8):
            lines.append(" ".join(obj[i:i + 8]))
        lines.append("Subject To")
        for c in self.constraints:
            lhs = " ".join(term(a, self.var_names[v], i == 0) for i, (a, v) in enumerate(zip(c.coefficients, c.variables)))
            pre = f"{c.name}: " if c.name else ""
            lines.append(f"{pre}{lhs} {c.ineq} {c.rhs}")
        lines.append("End")
        return "\n".join(lines) + "\n"


_NUM = r"(?:\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)"
_VAR = r"[A-Za-z][A-Za-z0-9_\-/(){},#;\[\].']*"
_TERM = re.compile(rf"\s*([+-])?\s*({_NUM})?\s*\*?\s*({_VAR})")
_INEQ = re.compile(r"(<=|>=|=)")
_SECTION = re.compile(r"^\s*(bounds|binaries|binary|generals|general|coalesce)\s*$", re.I)


def _parse_terms(s: str, what: str):
    terms, pos = [], 0
    s = s.strip()
    while pos < len(s):
        m = _TERM.match(s, pos)
        if not m or m.end() == pos:
            rest = s[pos:].strip()
            if not rest:
                break
            raise ValueError(f"cannot parse {what} near '{rest[:40]}'")
        if m.group(1) is None and terms:
            # every term but the first carries a sign (ILP_parser.cpp:54-60,100-104; OPB_parser.cpp:43,57); in a row,
            # `x y`, `x*y`, `2 x * y` are products of variables (inequality_monomial, :86-98): refused, not read as sums
            near = s[pos:].strip()[:40]
            if what == "constraint":
                raise ValueError(f"nonlinear constraint (product of variables) near '{near}' is not supported")
            raise ValueError(f"cannot parse {what}: term without a sign near '{near}'")
        sign = -1.0 if m.group(1) == "-" else 1.0
        coeff = float(m.group(2)) if m.group(2) is not None else 1.0
        terms.append((sign * coeff, m.group(3)))
        pos = m.end()
    return terms


# operands and relations in turn; a lone `<` or `>` is a token of its own (refused below), exactly as host/ilp.cpp: scan_bound —
# findall must never skip a character: `x < = 1` is an error, not the fixation `x = 1`
_BOUND_TOK = re.compile(rf"\s*(<=|>=|=|{_VAR}|[^\s<>=]+|[<>])")


def _parse_bound(ilp: "ILP", line: str, zeros: set, ones: set) -> None:
    """One line of the Bounds section: the reference's four forms (ILP_parser.cpp:128-131: `x = v`, `x <= v`,
    `v <= x`, `lb <= x <= ub`, v in {0, 1}) and their mirror images with `>=` (the reference's own test input has
    `x2 >= 0`, test/test_ILP_parser.cpp:15).  Anything else raises: it would change the model if it meant something."""
    tok = _BOUND_TOK.findall(line)

    def bad(why):
        return ValueError(f"cannot read Bounds line '{line}': {why}")

    if "".join(tok) != "".join(line.split()):      # the scanner consumed the whole line, or the line is refused
        raise bad("unreadable characters")

    def value(t):
        if t in ("0", "+0", "0.0"):
            return 0
        if t in ("1", "+1", "1.0"):
            return 1
        raise bad("a bound of a binary variable is 0 or 1")

    def variable(t):
        if t not in ilp._index:
            raise bad(f"variable '{t}' is in no row and not in the objective")
        return ilp._index[t]

    def lower(v, b):      # b <= x
        if b == 1:
            ones.add(v)

    def upper(v, b):      # x <= b
        if b == 0:
            zeros.add(v)

    rel = ("<=", ">=", "=")
    if len(tok) == 3 and tok[1] in rel:
        var_first = tok[0][0].isalpha()
        v = variable(tok[0] if var_first else tok[2])
        b = value(tok[2] if var_first else tok[0])
        if tok[1] == "=":
            lower(v, b)
            upper(v, b)
        elif (tok[1] == "<=") == var_first:
            upper(v, b)
        else:
            lower(v, b)
    elif len(tok) == 5 and tok[1] == tok[3] and tok[1] in ("<=", ">="):
        v = variable(tok[2])
        a, c = value(tok[0]), value(tok[4])
        lb, ub = (a, c) if tok[1] == "<=" else (c, a)
        if lb > ub:
            raise bad("lower bound above upper bound")
        lower(v, lb)
        upper(v, ub)
    else:
        raise bad("expected `x = v`, `x <= v`, `x >= v`, `v <= x`, `v >= x` or `lb <= x <= ub`")
    if v in zeros and v in ones:
        raise bad("variable is fixed to 0 and to 1")


def parse_lp(text: str) -> ILP:
    """Parse the .lp subset of the reference's ILP_parser (src/ILP/ILP_parser.cpp:24-140)."""
    lines = [ln for ln in text.splitlines() if not ln.lstrip().startswith("\\")]
    body = "\n".join(lines)
    m = re.search(r"^\s*(Minimize|Minimise|min)\s*$", body, re.I | re.M)
    if not m:
        raise ValueError("LP text has no 'Minimize' line")
    st = re.search(r"^\s*Subject To\s*$|^\s*s\.t\.\s*$|^\s*st\s*$", body[m.end():], re.I | re.M)
    if not st:
        raise ValueError("LP text has no 'Subject To' line")
    obj_text = body[m.end(): m.end() + st.start()]
    rest = body[m.end() + st.end():]
    ilp = ILP()
    obj_text = re.sub(r"^\s*[A-Za-z_][\w]*\s*:", "", obj_text.strip())  # optional objective name
    # a trailing constant (sign number with no variable) is the objective constant
    mconst = re.search(rf"([+-])\s*({_NUM})\s*$", obj_text)
    if mconst and not re.search(rf"{_VAR}\s*$", obj_text):
        ilp.constant = float(mconst.group(2)) * (-1.0 if mconst.group(1) == "-" else 1.0)
        obj_text = obj_text[: mconst.start()]
    for c, name in _parse_terms(obj_text.replace("\n", " "), "objective"):
        ilp.objective[ilp.var(name)] += c
    # constraints: a row may span several lines; it ends at the line holding the relation + rhs
    pending = ""
    rest_lines = rest.splitlines()
    i_sec = len(rest_lines)
    for i, ln in enumerate(rest_lines):
        s = ln.strip()
        if not s:
            continue
        if re.match(r"^end\s*$", s, re.I) or _SECTION.match(s):
            i_sec = i
            break
        pending = (pending + " " + s).strip()
        mi = _INEQ.search(pending)
        if not mi:
            continue
        rhs_s = pending[mi.end():].strip()
        if not re.fullmatch(rf"[+-]?\s*{_NUM}", rhs_s):
            continue  # rhs not complete yet
        lhs = pending[: mi.start()]
        name = ""
        mn = re.match(r"^\s*([^\s:]+)\s*:", lhs)
        if mn:
            name = mn.group(1)
            lhs = lhs[mn.end():]
        terms = _parse_terms(lhs, "constraint")
        rhs = float(rhs_s.replace(" ", ""))
        if rhs != int(rhs) or any(c != int(c) for c, _ in terms):
            raise ValueError("only integer constraint coefficients are supported (as the reference, ILP_parser.cpp:262-300)")
        ilp.add_constraint([(int(c), n) for c, n in terms], mi.group(1), int(rhs), name)
        pending = ""
    if pending:
        raise ValueError(f"incomplete constraint: '{pending[:60]}'")
    # sections behind the rows: Bounds lines fix variables (ILP_parser.cpp:128-131,343-436), the lists of the other
    # sections are skipped — every variable is binary anyway (:144, `until<end_line>`)
    zeros, ones, in_bounds = set(), set(), False
    for ln in rest_lines[i_sec:]:
        s = ln.strip()
        if not s:
            continue
        if re.match(r"^end\s*$", s, re.I):
            break
        ms = _SECTION.match(s)
        if ms:
            in_bounds = ms.group(1).lower() == "bounds"
            continue
        if in_bounds:
            _parse_bound(ilp, s, zeros, ones)
    if zeros or ones:
        return ilp.reduce(zeros, ones)
    return ilp


def parse_opb(text: str) -> ILP:
    """The OPB (pseudo-Boolean) subset of the reference's OPB_parser (src/ILP/OPB_parser.cpp:23-60): leading
    `* comment` lines, `min: <terms> ;`, then `<terms> {<=,>=,=} <integer> ;` rows, which may span lines; whatever
    follows the last `;` is ignored (`until<eof>`)."""
    lines, header = [], True
    for ln in text.splitlines():
        if header and ln.strip().startswith("*"):
            continue
        if ln.strip():
            header = False
        lines.append(ln)
    t = " ".join(lines).strip()
    if not t.startswith("min:"):
        raise ValueError("could not read input: OPB text must start with 'min:'")
    stmts = t[4:].split(";")
    if len(stmts) < 2:
        raise ValueError("could not read input: objective is not terminated by ';'")
    ilp = ILP()
    for c, name in _parse_terms(stmts[0], "objective"):
        ilp.objective[ilp.var(name)] += c
    for row in stmts[1:-1]:                      # the piece after the last ';' is not a complete row
        row = row.strip()
        if not row:
            continue
        mi = _INEQ.search(row)
        if not mi:
            raise ValueError(f"cannot parse constraint near '{row[:40]}'")
        rhs_s = row[mi.end():].strip()
        if not re.fullmatch(r"[+-]?\d+", rhs_s):
            raise ValueError("only integer constraint coefficients are supported (OPB_parser.cpp:55)")
        terms = _parse_terms(row[: mi.start()], "constraint")
        if any(c != int(c) for c, _ in terms):
            raise ValueError("only integer constraint coefficients are supported (OPB_parser.cpp:47-48)")
        ilp.add_constraint([(int(c), n) for c, n in terms], mi.group(1), int(rhs_s))
    return ilp


def parse_lp_or_opb(text: str) -> ILP:
    """bdd_solver::read_ILP for strings (bdd_solver.cpp:59-63): the .lp grammar first, then OPB."""
    try:
        return parse_lp(text)
    except ValueError as lp_error:
        try:
            return parse_opb(text)
        except ValueError:
            raise lp_error


def to_bdd_collection(ilp: ILP) -> BddCollection:
    """bdd_preprocessor::add_ilp for linear rows (bdd_preprocessor.cpp:165-226).
    Rows that are trivially true are skipped (:213-214); infeasible rows raise (:215-216)."""
    col = BddCollection()
    for c in ilp.constraints:
        if len(set(c.variables)) != len(c.variables):
            raise ValueError(f"constraint '{c.name}' repeats a variable")
        if c.is_simplex():
            col.add_simplex(c.variables)
            continue
        try:
            col.add_linear(c.coefficients, c.ineq, c.rhs, c.variables)
        except ValueError as e:
            if "trivially true" in str(e):
                continue
            raise RuntimeError("problem is infeasible") from e
    return col


# --------------------------------------------------------------------------------------- long-BDD splitting
# bdd_preprocessor.cpp:25-31 takes (SM count * max threads per SM) / 10 as the number of BDD nodes the device
# works on concurrently; the same figure for MI355X: 256 CUs * 2048 resident work-items / 10.
MI355X_CONCURRENT_NODES = 256 * 2048 // 10


def compute_split_length(col: BddCollection, parallelism: int = MI355X_CONCURRENT_NODES) -> int:
    """Largest chunk length >= 200 layers for which the hop-wise node counts, folded at that length, keep the
    device at >= 50 % average occupancy (compute_split_length, bdd_preprocessor.cpp:32-121)."""
    import numpy as np

    widths = np.zeros(0, dtype=np.int64)
    for b in range(col.nr_bdds()):
        w = np.asarray(col.layer_widths(b), dtype=np.int64)
        if w.size > widths.size:
            widths = np.concatenate((widths, np.zeros(w.size - widths.size, np.int64)))
        widths[: w.size] += w
    if widths.size == 0:
        return 0
    widths = np.maximum.accumulate(widths[::-1])[::-1]          # a hop counts as busy as the busiest later hop

    def occupancy(w):
        return float(np.minimum(w, parallelism).sum()) / parallelism / w.size

    length = int(widths.size)
    while length >= 200:
        folded = np.zeros(length, np.int64)
        for start in range(0, widths.size, length):
            part = widths[start:start + length]
            folded[: part.size] += part
        if occupancy(folded) >= 0.5:
            break
        length -= 1
    return length


def split_long_bdds(col: BddCollection, nr_variables: int, split_length: int = None,
                    with_implication_bdd: bool = False, parallelism: int = MI355X_CONCURRENT_NODES):
    """The splitting stage of bdd_preprocessor::add_ilp (bdd_preprocessor.cpp:372-415): every BDD with more than
    `split_length` variables (None: compute_split_length) is replaced by its split_qbdd chunks; auxiliary
    variables are numbered from `nr_variables`.  Returns (#BDDs split, number of variables afterwards)."""
    if split_length is None:
        split_length = compute_split_length(col, parallelism)
    if split_length <= 0:
        return 0, nr_variables
    next_var, removed = nr_variables, []
    for b in range(col.nr_bdds()):
        if len(col.layer_widths(b)) > split_length:
            new_nrs, next_var = col.split_qbdd(b, split_length, next_var, with_implication_bdd)
            if len(new_nrs) > 1:
                removed.append(b)
    col.remove(removed)
    return len(removed), next_var
