"""Flat BDD interchange format + closed-form QBDD builders.

The storage is bit-compatible with the reference's ``BDD::bdd_collection``
(reference: include/bdd_collection/bdd_collection.h:14-36,122-288): a flat array of
``bdd_instruction{size_t lo, hi, index}`` with ABSOLUTE lo/hi indices, one
``bdd_delimiters`` entry per BDD, nodes grouped by variable in BDD order and the two
terminals (index == TOPSINK / BOTSINK) last.  This is the *input type* of the hot path
(SURVEY.md §8 a-0); the HIP library consumes it through ``bddmma_create``.

The builders restate the canned constraints of the reference
(src/bdd_collection/bdd_collection.cpp:2039-2134 ``simplex_constraint``,
``not_all_false_constraint`` followed by ``make_qbdd`` :1670-1812) in closed form so
that benchmark instances can be generated without any reference code on the GPU box.
tests/test_bdd_builders.py checks them node-for-node against oracle/_ref.
"""
from __future__ import annotations

import numpy as np

TOPSINK = np.uint64(2**64 - 1)
BOTSINK = np.uint64(2**64 - 2)
# local placeholders used inside templates (resolved to absolute indices on append)
_T = -1  # top sink
_B = -2  # bot sink


class BddCollection:
    """Growable flat QBDD store (reference: BDD::bdd_collection)."""

    def __init__(self):
        self._chunks = []  # list of (n,3) uint64 arrays with ABSOLUTE indices
        self._delims = [np.zeros(1, dtype=np.uint64)]
        self._n = 0
        self._nb = 0

    @classmethod
    def from_arrays(cls, instr, delims) -> "BddCollection":
        """A collection over existing storage in the reference's flat format: instr (n, 3) = {lo, hi, index} rows with absolute
        node indices (bdd_collection.h:14-36), delims (nr_bdds + 1) = first instruction of every BDD.  The arrays are copied."""
        instr = np.ascontiguousarray(np.asarray(instr), dtype=np.uint64).reshape(-1, 3).copy()
        delims = np.ascontiguousarray(np.asarray(delims), dtype=np.uint64).reshape(-1).copy()
        if delims.size == 0 or delims[0] != 0 or int(delims[-1]) != instr.shape[0] or np.any(np.diff(delims.astype(np.int64)) < 0):
            raise ValueError("delims must start at 0, end at the number of instructions and be non-decreasing")
        col = cls()
        col._chunks = [instr] if instr.shape[0] else []
        col._delims = [delims]
        col._n, col._nb = int(instr.shape[0]), int(delims.size) - 1
        return col

    # ------------------------------------------------------------------ access
    @property
    def instr(self) -> np.ndarray:
        if len(self._chunks) != 1:
            self._chunks = [np.concatenate(self._chunks, axis=0) if self._chunks else np.zeros((0, 3), np.uint64)]
        return self._chunks[0]

    @property
    def delims(self) -> np.ndarray:
        if len(self._delims) != 1:
            self._delims = [np.concatenate(self._delims)]
        return self._delims[0]

    def nr_bdds(self) -> int:
        return self._nb

    def nr_bdd_nodes(self) -> int:
        return self._n

    def nr_variables(self) -> int:
        ins = self.instr
        nt = ins[:, 2] < BOTSINK
        return int(ins[nt, 2].max()) + 1 if nt.any() else 0

    def variables(self, b: int) -> list:
        d = self.delims
        idx = self.instr[int(d[b]):int(d[b + 1]), 2]
        idx = idx[idx < BOTSINK]
        out = []
        for v in idx.tolist():
            if not out or out[-1] != v:
                out.append(v)
        return out

    # ---------------------------------------------------------------- building
    def _append_local(self, lo, hi, var_of_node, n_bdds, nodes_per_bdd, top_first):
        """Append n_bdds BDDs that share one local template.

        lo/hi: (n_bdds, nodes_per_bdd) int64 local child indices (or _T/_B);
        var_of_node: (n_bdds, nodes_per_bdd) uint64 global variable ids.
        Two terminals are appended after the nodes of every BDD.
        """
        per = nodes_per_bdd + 2
        base = self._n + np.arange(n_bdds, dtype=np.int64)[:, None] * per
        top_local = nodes_per_bdd + (0 if top_first else 1)
        bot_local = nodes_per_bdd + (1 if top_first else 0)

        def resolve(x):
            x = np.where(x == _T, top_local, x)
            x = np.where(x == _B, bot_local, x)
            return (x + base).astype(np.uint64)

        out = np.empty((n_bdds, per, 3), dtype=np.uint64)
        out[:, :nodes_per_bdd, 0] = resolve(lo)
        out[:, :nodes_per_bdd, 1] = resolve(hi)
        out[:, :nodes_per_bdd, 2] = var_of_node
        out[:, top_local, :] = TOPSINK
        out[:, bot_local, :] = BOTSINK
        self._chunks.append(out.reshape(-1, 3))
        self._delims.append((self._n + (np.arange(n_bdds, dtype=np.int64) + 1) * per).astype(np.uint64))
        first = self._nb
        self._n += n_bdds * per
        self._nb += n_bdds
        return first

    def add_simplex(self, variables) -> int:
        """sum_i x_i = 1  (bdd_collection.cpp:2039-2103).  variables: (n,) or (n_bdds, n)."""
        v = np.atleast_2d(np.asarray(variables, dtype=np.uint64))
        nb, n = v.shape
        if n == 1:
            lo = np.full((nb, 1), _B, np.int64)
            hi = np.full((nb, 1), _T, np.int64)
            return self._append_local(lo, hi, v, nb, 1, top_first=False)
        nn = 2 * n - 1
        lo = np.empty(nn, np.int64)
        hi = np.empty(nn, np.int64)
        layer = np.empty(nn, np.int64)
        lo[0], hi[0], layer[0] = 1, 2, 0
        for i in range(1, n - 1):
            a, c = 2 * i - 1, 2 * i  # sum == 0, sum == 1
            lo[a], hi[a] = 2 * i + 1, 2 * i + 2
            lo[c], hi[c] = 2 * i + 2, _B
            layer[a] = layer[c] = i
        a, c = 2 * n - 3, 2 * n - 2
        lo[a], hi[a] = _B, _T
        lo[c], hi[c] = _T, _B
        layer[a] = layer[c] = n - 1
        return self._append_local(np.broadcast_to(lo, (nb, nn)), np.broadcast_to(hi, (nb, nn)), v[:, layer], nb, nn,
                                  top_first=False)

    def add_covering(self, variables) -> int:
        """sum_i x_i >= 1 as the QBDD that not_all_false_constraint(n) + make_qbdd yields
        (bdd_collection.cpp:2105-2134, :1670-1812; SURVEY.md §8c dump for n = 4)."""
        v = np.atleast_2d(np.asarray(variables, dtype=np.uint64))
        nb, n = v.shape
        if n == 1:
            lo = np.full((nb, 1), _B, np.int64)
            hi = np.full((nb, 1), _T, np.int64)
            # a single-node not_all_false BDD already is a QBDD: make_qbdd is skipped, sinks stay bot, top
            return self._append_local(lo, hi, v, nb, 1, top_first=False)
        nn = 2 * n - 1
        lo = np.empty(nn, np.int64)
        hi = np.empty(nn, np.int64)
        layer = np.empty(nn, np.int64)
        lo[0], hi[0], layer[0] = 1, 2, 0
        for i in range(1, n - 1):
            a, c = 2 * i - 1, 2 * i  # uncovered, covered
            lo[a], hi[a] = 2 * i + 1, 2 * i + 2
            lo[c], hi[c] = 2 * i + 2, 2 * i + 2
            layer[a] = layer[c] = i
        a, c = 2 * n - 3, 2 * n - 2
        lo[a], hi[a] = _B, _T
        lo[c], hi[c] = _T, _T
        layer[a] = layer[c] = n - 1
        return self._append_local(np.broadcast_to(lo, (nb, nn)), np.broadcast_to(hi, (nb, nn)), v[:, layer], nb, nn,
                                  top_first=True)

    def add_linear(self, coefficients, ineq: str, rhs: int, variables) -> int:
        """sum_i a_i x_i {<=,=,>=} rhs as the canonical (minimal) QBDD over `variables` in the
        given order — the function the reference builds with bdd_converter + make_qbdd
        (bdd_preprocessor.cpp:196-226).  Returns the BDD number, or raises on a constraint that is
        trivially true/false (the reference skips / throws, :213-216)."""
        a = [int(c) for c in coefficients]
        vs = [int(x) for x in variables]
        n = len(a)
        assert n == len(vs) and n > 0
        cmp = {"<=": lambda s: s <= rhs, "=": lambda s: s == rhs, ">=": lambda s: s >= rhs,
               "<": lambda s: s < rhs, ">": lambda s: s > rhs}[ineq]
        # level-wise reachable partial sums
        levels = [{0}]
        for i in range(n):
            nxt = set()
            for s in levels[i]:
                nxt.add(s)
                nxt.add(s + a[i])
            levels.append(nxt)
        # bottom-up canonical ids: id -1 = top, -2 = bot, else index into level's unique table
        T, Bt = _T, _B
        cur = {s: (T if cmp(s) else Bt) for s in levels[n]}
        tables = [None] * n  # per level: list of (lo_id, hi_id)
        for i in range(n - 1, -1, -1):
            uniq = {}
            tab = []
            new = {}
            for s in sorted(levels[i]):
                key = (cur[s], cur[s + a[i]])
                if key == (Bt, Bt):
                    new[s] = Bt
                    continue
                if key not in uniq:
                    uniq[key] = len(tab)
                    tab.append(key)
                new[s] = uniq[key]
            tables[i] = tab
            cur = new
        root = cur[0]
        if root == Bt:
            raise ValueError("constraint is infeasible")
        # T appears as a child only at the last level, so a sub-function that is constant TRUE below
        # level i keeps explicit nodes on every level (QBDD: no level skipping).
        if all(cmp(s) for s in levels[n]):
            raise ValueError("constraint is trivially true")
        # drop nodes unreachable from the root after bot-pruning
        reach = [set() for _ in range(n)]
        reach[0].add(root)
        for i in range(n - 1):
            for k in reach[i]:
                for ch in tables[i][k]:
                    if ch >= 0:
                        reach[i + 1].add(ch)
        # A variable the function does not depend on is absent from the reference's BDD: the reduced
        # BDD built by bdd_mgr does not contain it and make_qbdd only fills in variables that occur.
        # Such a variable has lo == hi on every node of its level; f is then the row without that term.
        dead = [i for i in range(n) if all(tables[i][k][0] == tables[i][k][1] for k in reach[i])]
        if dead:
            keep = [i for i in range(n) if i not in set(dead)]
            return self.add_linear([a[i] for i in keep], ineq, rhs, [vs[i] for i in keep])
        remap = []
        offs = []
        total = 0
        for i in range(n):
            order = sorted(reach[i])
            remap.append({k: j for j, k in enumerate(order)})
            offs.append(total)
            total += len(order)
        lo = np.empty(total, np.int64)
        hi = np.empty(total, np.int64)
        layer = np.empty(total, np.int64)
        for i in range(n):
            for k, j in remap[i].items():
                l, h = tables[i][k]
                lo[offs[i] + j] = l if l < 0 else offs[i + 1] + remap[i + 1][l]
                hi[offs[i] + j] = h if h < 0 else offs[i + 1] + remap[i + 1][h]
                layer[offs[i] + j] = i
        v = np.asarray(vs, dtype=np.uint64)[None, :]
        return self._append_local(lo[None, :], hi[None, :], v[:, layer], 1, total, top_first=False)

    def add_cardinality(self, variables, k: int) -> int:
        """sum_i x_i = k (bdd_collection::cardinality_constraint)."""
        return self.add_linear([1] * len(variables), "=", k, variables)

    # ---------------------------------------------------------------- splitting
    # ---- text exports ("export bdd lp" / "export bdd graph" of the driver; reference include/bdd_collection/bdd_collection.h:731-830, :663-729)
    def write_bdd_lp(self, costs) -> str:
        """The network-flow LP over the arcs of all BDDs, linked by the original variables; same rows, names and order as the reference's."""
        ins, d = self.instr, self.delims
        out = ["Minimize\n"]
        for i, c in enumerate(costs):
            out.append(f"{'-' if c < 0 else '+'}{abs(float(c)):g} x_{i}\n")
        out.append("Subject To\n")
        is_bot = lambda k: ins[k, 2] == BOTSINK
        is_sink = lambda k: ins[k, 2] >= BOTSINK
        arc = lambda b, k, v: f"arc_{b}_{k - int(d[b])}_{v}"
        for b in range(self.nr_bdds()):
            d0, d1 = int(d[b]), int(d[b + 1])
            lo, hi = int(ins[d0, 0]), int(ins[d0, 1])
            row = f"R_{b}: "
            if not is_bot(lo):
                row += arc(b, d0, 0)
            if not is_bot(hi):
                row += " + " + arc(b, d0, 1)
            out.append(row + " = 1\n")
            incoming = [[] for _ in range(d1 - d0)]
            if not is_sink(lo):
                incoming[lo - d0].append((d0, 0))
            if not is_sink(hi):
                incoming[hi - d0].append((d0, 1))
            for i in range(d0 + 1, d1 - 2):
                lo, hi = int(ins[i, 0]), int(ins[i, 1])
                row = f"FC_{b}_{i - d0}: "
                if not is_bot(lo):
                    row += arc(b, i, 0)
                if not is_bot(hi):
                    row += " + " + arc(b, i, 1)
                for n, v in incoming[i - d0]:
                    row += " - " + arc(b, n, v)
                out.append(row + " = 0\n")
                if not is_sink(lo):
                    incoming[lo - d0].append((i, 0))
                if not is_sink(hi):
                    incoming[hi - d0].append((i, 1))
        for b in range(self.nr_bdds()):
            d0, d1 = int(d[b]), int(d[b + 1])
            cur = int(ins[d0, 2])
            row = ""
            for i in range(d0, d1 - 2):
                if int(ins[i, 2]) != cur:
                    out.append(row + f" - x_{cur} = 0\n")
                    row = ""
                    cur = int(ins[i, 2])
                if not is_bot(int(ins[i, 1])):
                    row += " + " + arc(b, i, 1)
            out.append(row + f" - x_{cur} = 0\n")
        out.append("Bounds\nBinaries\n")
        for b in range(self.nr_bdds()):
            for i in range(int(d[b]), int(d[b + 1]) - 2):
                if not is_bot(int(ins[i, 0])):
                    out.append(arc(b, i, 0) + "\n")
                if not is_bot(int(ins[i, 1])):
                    out.append(arc(b, i, 1) + "\n")
        out.append("End\n")
        return "".join(out)

    def export_graphviz(self, b: int) -> str:
        """BDD b as a Graphviz digraph, one cluster per variable (the reference iterates an unordered_map over the clusters: their order in
        the file carries no meaning; here they come in variable order, the sinks first)."""
        ins, d = self.instr, self.delims
        d0, d1 = int(d[b]), int(d[b + 1])
        clusters, last_node = {}, {}
        for i in range(d0, d1):
            idx = ins[i, 2]
            if idx >= BOTSINK:
                clusters.setdefault(2**64 - 1, []).append(f'{i - d0} [label="{"top" if idx == TOPSINK else "bot"}"];\n')
            else:
                clusters.setdefault(int(idx), []).append(f'{i - d0} [label="{int(idx)}"];\n')
                last_node[int(idx)] = i - d0
        out = ["digraph BDD\n{\n"]
        for key in sorted(clusters, key=lambda k: (k != 2**64 - 1, k)):
            out.append(f"subgraph cluster_{key} {{\n" + "".join(clusters[key]) + "color = blue\n}\n")
        order = [last_node[v] for v in sorted(last_node)]
        for a, c in zip(order, order[1:]):
            out.append(f"{a} -> {c} [style=invis];\n")
        for i in range(d0, d1):
            if ins[i, 2] >= BOTSINK:
                continue
            out.append(f"{i - d0} -> {int(ins[i, 1]) - d0};\n")
            out.append(f'{i - d0} -> {int(ins[i, 0]) - d0}[style="dashed"];\n')
        out.append("}\n")
        return "".join(out)

    def layer_widths(self, b: int) -> list:
        """#nodes per variable layer of BDD b (bdd_collection::layer_widths, bdd_collection.h:195)."""
        d = self.delims
        idx = self.instr[int(d[b]):int(d[b + 1]) - 2, 2]
        if idx.size == 0:
            return []
        starts = np.flatnonzero(np.concatenate(([True], idx[1:] != idx[:-1])))
        return np.diff(np.concatenate((starts, [idx.size]))).tolist()

    def remove(self, bdd_nrs) -> None:
        """bdd_collection::remove (bdd_collection.h:200-203): drop the listed BDDs, compacting the storage."""
        drop = sorted(set(int(b) for b in bdd_nrs))
        if not drop:
            return
        ins, d = self.instr, self.delims.astype(np.int64)
        keep = [b for b in range(self._nb) if b not in set(drop)]
        chunks, delims, pos = [], [0], 0
        for b in keep:
            part = ins[d[b]:d[b + 1]].copy()
            nt = part[:, 2] < BOTSINK
            shift = np.int64(pos) - d[b]
            part[nt, 0] = (part[nt, 0].astype(np.int64) + shift).astype(np.uint64)
            part[nt, 1] = (part[nt, 1].astype(np.int64) + shift).astype(np.uint64)
            chunks.append(part)
            pos += part.shape[0]
            delims.append(pos)
        self._chunks = [np.concatenate(chunks, axis=0)] if chunks else []
        self._delims = [np.asarray(delims, dtype=np.uint64)]
        self._n, self._nb = pos, len(keep)

    def split_qbdd(self, b: int, chunk_size: int, aux_var_start: int, with_implication_bdd: bool = False):
        """Cut a long QBDD into chunks of <= chunk_size variable layers that are coupled through auxiliary
        one-hot variables (bdd_collection::split_qbdd, src/bdd_collection/bdd_collection.cpp:507-949).

        For a cut in front of a layer with w nodes, w fresh variables a_0..a_{w-1} say which node the path
        crosses.  The chunk after the cut starts with a triangular "head" over a_0..a_{w-1} that accepts exactly
        the one-hot assignments and enters node j of the layer for the j-th of them; the chunk before the cut
        ends, behind every node j of the cut layer, in a "tail" chain over the same variables that accepts
        only the matching one-hot assignment.  Output BDDs are appended (the original stays; callers remove
        it, bdd_preprocessor.cpp:393-412); node order and terminal order (bot, top) are the reference's, and
        tests/test_bdd_builders.py compares node-for-node with oracle/_ref.

        Returns (new BDD numbers, next free auxiliary variable).  A cut layer of width 1 — asserted against
        by the reference (:583) — gets one auxiliary variable that has to be 1.
        """
        assert chunk_size > 0
        d = self.delims
        off = int(d[b])
        n_nodes = int(d[b + 1]) - off - 2
        ins = self.instr[off:off + n_nodes + 2].astype(np.int64, copy=True)   # sink markers wrap to -1 / -2
        widths = self.layer_widths(b)
        n_layers = len(widths)
        if n_layers <= chunk_size:
            return [b], aux_var_start
        loff = np.concatenate(([0], np.cumsum(widths))).tolist()             # local offsets of the layers, [n_layers] = n_nodes
        n_chunks = (n_layers + chunk_size - 1) // chunk_size
        aux = [aux_var_start]
        for c in range(1, n_chunks - 1):
            aux.append(aux[-1] + widths[c * chunk_size])
        top_local, bot_local = off + n_nodes, off + n_nodes + 1
        if int(ins[n_nodes, 2]) != -1:                                       # terminal order of the input: (bot, top)
            top_local, bot_local = bot_local, top_local

        new_nrs = []
        for c in range(n_chunks):
            first, last = c * chunk_size, min((c + 1) * chunk_size, n_layers) - 1
            w_head = widths[first] if c > 0 else 0
            w_tail = widths[last + 1] if c + 1 < n_chunks else 0
            n_head = w_head * (w_head + 1) // 2
            n_chunk = loff[last + 1] - loff[first]
            n_tail = (w_tail * (w_tail + 1) // 2 + w_tail - 1) if w_tail else 0
            BOT, TOP = n_head + n_chunk + n_tail, n_head + n_chunk + n_tail + 1
            lo, hi, var = [], [], []

            def emit(v, l, h):
                var.append(v); lo.append(l); hi.append(h)

            # head: row i has i+1 nodes; node (i, j) is "j-th candidate still open after a_0..a_{i-1}"
            head = lambda i, j: i * (i + 1) // 2 + j
            for i in range(w_head):
                a = aux[c - 1] + i
                lastrow = i + 1 == w_head
                for j in range(i + 1):
                    if not lastrow:
                        emit(a, head(i + 1, 0), head(i + 1, 1)) if j == 0 else emit(a, head(i + 1, j + 1), BOT)
                    else:
                        emit(a, BOT, n_head + j) if j == 0 else emit(a, n_head + j, BOT)
            # the chunk's own nodes, children shifted; children in layer last+1 land on the tail's first row
            shift = n_head - loff[first]
            for u in range(loff[first], loff[last + 1]):
                l, h, v = (int(t) for t in ins[u])
                cl = TOP if l == top_local else BOT if l == bot_local else l - off + shift
                ch = TOP if h == top_local else BOT if h == bot_local else h - off + shift
                emit(v, cl, ch)
            # tail
            if w_tail == 1:
                emit(aux[c], BOT, TOP)
            elif w_tail > 1:
                W, base = w_tail, n_head + n_chunk
                tail = lambda i, j: base + (j if i == 0 else W + W * (i - 1) + j - (i - 1) * (i - 2) // 2)
                for j in range(W):
                    emit(aux[c], BOT, tail(1, W - 1)) if j + 1 == W else emit(aux[c], tail(1, j), BOT)
                for i in range(1, W - 1):
                    n_row = W - i + 1
                    for j in range(n_row):
                        if j + 1 == n_row:
                            emit(aux[c] + i, tail(i + 1, W - i - 1), BOT)
                        elif j + 2 == n_row:
                            emit(aux[c] + i, BOT, tail(i + 1, j))
                        else:
                            emit(aux[c] + i, tail(i + 1, j), BOT)
                emit(aux[c] + W - 1, BOT, TOP)
                emit(aux[c] + W - 1, TOP, BOT)
            n = len(var)
            assert n == BOT, (n, BOT)
            lo = np.where(np.asarray(lo) == TOP, _T, np.where(np.asarray(lo) == BOT, _B, np.asarray(lo)))
            hi = np.where(np.asarray(hi) == TOP, _T, np.where(np.asarray(hi) == BOT, _B, np.asarray(hi)))
            new_nrs.append(self._append_local(lo[None, :].astype(np.int64), hi[None, :].astype(np.int64),
                                              np.asarray(var, dtype=np.uint64)[None, :], 1, n, top_first=False))
        if with_implication_bdd and len(new_nrs) > 2:
            imp = self._implication_bdd(ins, off, widths, loff, chunk_size, n_chunks, aux, top_local, bot_local)
            if imp is not None:
                new_nrs.append(imp)
        return new_nrs, aux[-1] + widths[(n_chunks - 1) * chunk_size]

    def _implication_bdd(self, ins, off, widths, loff, chunk, n_chunks, aux, top_abs, bot_abs):
        """The optional extra BDD of split_qbdd (bdd_collection.cpp:801-941): over the auxiliary variables only, the
        conjunction of (a) one simplex per cut ("exactly one node of the cut layer is crossed") and (b) for every node of
        a cut and every other cut, the clause "this node is crossed => one of the nodes connected to it by a directed path
        in the original BDD is crossed" (skipped when all nodes of the other cut are connected).  Returns the new BDD
        number, or None when there is no clause of kind (b) (the reference then drops the simplex BDDs again, :915-919).

        The reference forms the conjunction with bdd_mgr (`bdd_and`, `reorder`, `make_qbdd`); here it is built as the
        product automaton of the constraints over the variables in ascending order and reduced bottom-up, which gives
        the same canonical quasi-reduced BDD; the order of the nodes inside a layer (first discovery, hi child first) equals the
        reference's on most instances but not all — it carries no meaning, and the tests compare up to that order."""
        n_cuts = n_chunks - 1
        cut_layer = lambda c: c * chunk                               # cuts are numbered 1 .. n_cuts
        aux_of = lambda c, i: aux[c - 1] + (widths[cut_layer(c)] - 1 - i)   # node i of cut c <-> auxiliary variable (:836,842)
        # sources[c][c2][j]: bitmask of the nodes of cut c that reach node j of the later cut c2
        sources = {}
        for c in range(1, n_cuts + 1):
            mask = [1 << i for i in range(widths[cut_layer(c)])]
            for l in range(cut_layer(c), cut_layer(n_cuts)):
                nxt = [0] * widths[l + 1]
                for k, u in enumerate(range(loff[l], loff[l + 1])):
                    for ch in (int(ins[u, 0]), int(ins[u, 1])):
                        if ch != top_abs and ch != bot_abs:
                            nxt[ch - off - loff[l + 1]] |= mask[k]
                mask = nxt
                if (l + 1) % chunk == 0:
                    sources[(c, (l + 1) // chunk)] = mask
        constraints = []   # ("simplex", vars) | ("clause", negated var, positive vars)
        for c in range(1, n_cuts + 1):
            constraints.append(("simplex", [aux_of(c, i) for i in range(widths[cut_layer(c)])]))
        n_clauses = 0
        for c in range(1, n_cuts):                                    # forward implications (:826-857)
            for c2 in range(c + 1, n_cuts + 1):
                w1, w2 = widths[cut_layer(c)], widths[cut_layer(c2)]
                for i1 in range(w1):
                    pos = [aux_of(c2, i2) for i2 in range(w2) if (sources[(c, c2)][i2] >> i1) & 1]
                    if len(pos) == w2:
                        continue
                    constraints.append(("clause", aux_of(c, i1), pos))
                    n_clauses += 1
        for c1 in range(2, n_cuts + 1):                               # reverse implications (:861-893)
            for c2 in range(1, c1):
                w1, w2 = widths[cut_layer(c1)], widths[cut_layer(c2)]
                for i1 in range(w1):
                    pos = [aux_of(c2, i2) for i2 in range(w2) if (sources[(c2, c1)][i1] >> i2) & 1]
                    if len(pos) == w2:
                        continue
                    constraints.append(("clause", aux_of(c1, i1), pos))
                    n_clauses += 1
        if n_clauses == 0:
            return None
        # ---- product automaton over the auxiliary variables in ascending order
        v_first, v_last = aux[0], aux[-1] + widths[cut_layer(n_cuts)] - 1
        by_var = {v: [] for v in range(v_first, v_last + 1)}
        last_var = []
        for k, con in enumerate(constraints):
            vs = con[1] if con[0] == "simplex" else [con[1]] + con[2]
            last_var.append(max(vs))
            for v in vs:
                by_var[v].append(k)
        DONE = 2                                                        # local state of a finished (accepted) constraint
        start = tuple(0 for _ in constraints)
        levels, trans = [{start: 0}], []                                # per level: state -> id ; per level: [(lo, hi)] with ids of the next level or _T / _B
        for v in range(v_first, v_last + 1):
            cur, nxt, tr = levels[-1], {}, [None] * len(levels[-1])
            for st, sid in cur.items():
                out = []
                for val in (0, 1):
                    ns, dead = list(st), False
                    for k in by_var[v]:
                        con = constraints[k]
                        if con[0] == "simplex":
                            x = ns[k] + val
                            if x > 1 or (v == last_var[k] and x != 1):
                                dead = True
                                break
                            ns[k] = DONE if v == last_var[k] else x
                        else:
                            x = 1 if (ns[k] == 1 or (v == con[1] and val == 0) or (v != con[1] and val == 1)) else 0
                            if v == last_var[k] and x != 1:
                                dead = True
                                break
                            ns[k] = DONE if v == last_var[k] else x
                    if dead:
                        out.append(_B)
                    elif v == v_last:
                        out.append(_T)
                    else:
                        out.append(nxt.setdefault(tuple(ns), len(nxt)))
                tr[sid] = tuple(out)
            trans.append(tr)
            levels.append(nxt)
        # ---- bottom-up reduction to the canonical quasi-reduced form
        n_lev = len(trans)
        canon = [None] * n_lev           # per level: old id -> canonical id (or _B)
        tables = [None] * n_lev          # per level: list of (lo, hi) over canonical ids of the next level
        for l in range(n_lev - 1, -1, -1):
            uniq, tab, cm = {}, [], []
            for lo, hi in trans[l]:
                lo = lo if lo < 0 else canon[l + 1][lo]
                hi = hi if hi < 0 else canon[l + 1][hi]
                if lo == _B and hi == _B:
                    cm.append(_B)
                    continue
                key = (lo, hi)
                if key not in uniq:
                    uniq[key] = len(tab)
                    tab.append(key)
                cm.append(uniq[key])
            canon[l], tables[l] = cm, tab
        assert canon[0][0] != _B, "the implication constraints cannot be infeasible"
        # nodes reachable from the root; inside a layer in order of first discovery when the parents are scanned in
        # order and the hi child is visited before the lo child (the order make_qbdd leaves, observed on oracle/_ref)
        order = [[canon[0][0]]] + [[] for _ in range(n_lev - 1)]
        for l in range(n_lev - 1):
            seen = set()
            for k in order[l]:
                for ch in (tables[l][k][1], tables[l][k][0]):
                    if ch >= 0 and ch not in seen:
                        seen.add(ch)
                        order[l + 1].append(ch)
        remap, offs, total = [], [], 0
        for l in range(n_lev):
            remap.append({k: j for j, k in enumerate(order[l])})
            offs.append(total)
            total += len(order[l])
        lo = np.empty(total, np.int64); hi = np.empty(total, np.int64); var = np.empty(total, np.uint64)
        for l in range(n_lev):
            for k, j in remap[l].items():
                a, b = tables[l][k]
                lo[offs[l] + j] = a if a < 0 else offs[l + 1] + remap[l + 1][a]
                hi[offs[l] + j] = b if b < 0 else offs[l + 1] + remap[l + 1][b]
                var[offs[l] + j] = v_first + l
        return self._append_local(lo[None, :], hi[None, :], var[None, :], 1, total, top_first=True)

    def permute(self, order) -> None:
        """Reorder the BDDs: new BDD i is old BDD order[i] (storage compacted, child indices re-based)."""
        order = np.asarray(order, dtype=np.int64)
        assert sorted(order.tolist()) == list(range(self._nb))
        ins, d = self.instr, self.delims.astype(np.int64)
        sizes = (d[1:] - d[:-1])[order]
        new_d = np.concatenate(([0], np.cumsum(sizes)))
        # source index of every new instruction, and the shift of its BDD
        src = np.repeat(d[:-1][order] - new_d[:-1], sizes) + np.arange(new_d[-1])
        out = ins[src].copy()
        shift = np.repeat(new_d[:-1] - d[:-1][order], sizes)
        nt = out[:, 2] < BOTSINK
        out[nt, 0] = (out[nt, 0].astype(np.int64) + shift[nt]).astype(np.uint64)
        out[nt, 1] = (out[nt, 1].astype(np.int64) + shift[nt]).astype(np.uint64)
        self._chunks = [out]
        self._delims = [new_d.astype(np.uint64)]

    def append(self, other: "BddCollection") -> None:
        ins = other.instr.copy()
        nt = ins[:, 2] < BOTSINK
        ins[nt, 0] += np.uint64(self._n)
        ins[nt, 1] += np.uint64(self._n)
        self._chunks.append(ins)
        self._delims.append(other.delims[1:] + np.uint64(self._n))
        self._n += other._n
        self._nb += other._nb

    def rebase(self, var_map) -> None:
        """bdd_collection::rebase: variable i -> var_map[i] for every BDD."""
        vm = np.asarray(var_map, dtype=np.uint64)
        ins = self.instr
        nt = ins[:, 2] < BOTSINK
        ins[nt, 2] = vm[ins[nt, 2].astype(np.int64)]

    # ------------------------------------------------------------- evaluation
    def evaluate(self, b: int, x) -> bool:
        """bdd_collection::evaluate (bdd_collection.h:293-311)."""
        ins = self.instr
        i = int(self.delims[b])
        while True:
            lo, hi, idx = (int(t) for t in ins[i])
            if idx == int(TOPSINK):
                return True
            if idx == int(BOTSINK):
                return False
            i = hi if x[idx] else lo
