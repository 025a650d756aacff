"""Device-layout builder (bdd_amd/csrc/layout.cpp) through the host-only debug ABI — CPU only.

Decodes the pack / hop / node-word arrays back into a graph and checks it is the input BDD
collection (same children, same variables), plus the structural invariants the kernels rely on.
"""
import ctypes as C

import numpy as np
import pytest

from bdd_amd import BddCollection, capi, to_bdd_collection
from bdd_amd.instances import assignment_ilp, random_set_cover
from util import GOLDEN, load_golden

TOP = np.uint64(2**64 - 1)
BOT = np.uint64(2**64 - 2)


class Layout:
    def __init__(self, col, pack_width=0, wide_pack_width=0, vars_per_bin=0, stage_cap=0, waves_per_block=0, exchange_by_variable=0, pack_fill=0, pack_stagger=0,
                 keep_bdd_order=0, chip=None):
        L = capi.lib()
        instr = np.ascontiguousarray(col.instr, dtype=np.uint64)
        delims = np.ascontiguousarray(col.delims, dtype=np.uint64)
        h = C.c_void_p()
        opts = capi.Options(pack_width, wide_pack_width, 0, vars_per_bin, stage_cap, waves_per_block)
        opts.exchange_by_variable = exchange_by_variable
        opts.pack_fill = pack_fill
        opts.pack_stagger = pack_stagger
        opts.keep_bdd_order = keep_bdd_order
        if chip is None:
            rc = L.bddmma_layout_create(C.byref(h), instr.ctypes.data_as(C.c_void_p), delims.ctypes.data_as(C.c_void_p),
                                        col.nr_bdds(), C.byref(opts))
        else:   # (real_size, n_cus, lds_bytes_per_cu): the layout bddmma_create builds on such a device
            rc = L.bddmma_layout_create_for_chip(C.byref(h), instr.ctypes.data_as(C.c_void_p), delims.ctypes.data_as(C.c_void_p),
                                                 col.nr_bdds(), C.byref(opts), *chip)
        capi.check(rc, None)
        self.L, self.h = L, h
        self.pack_stagger = pack_stagger
        sz = lambda w: int(L.bddmma_layout_size(h, w))
        self.pack_width = sz(16)
        self.n_slots, self.narrow_slots, self.n_layers = sz(0), sz(1), sz(2)
        self.np_n, self.np_w, self.n_hops, self.n_vars = sz(3), sz(4), sz(5), sz(6)
        rec_n, rec_w = sz(7), sz(8)
        self.np_h, rec_h, self.huge_pack_width = sz(17), sz(18), sz(19)

        def get(which, n, dt):
            a = np.zeros(max(n, 1), dt)
            capi.check(L.bddmma_layout_copy(h, which, a.ctypes.data_as(C.c_void_p)), None)
            return a[:n]
        self.nwords = get(0, self.narrow_slots, np.uint32)
        self.wwords = get(1, self.n_slots - self.narrow_slots, np.uint64)
        self.slot_to_instr = get(2, self.n_slots, np.uint64)
        self.layer_var = get(3, self.n_layers, np.int32)
        self.layer_bdd = get(4, self.n_layers, np.int32)
        self.sets = []
        for base, P, rec in ((5, self.np_n, rec_n), (9, self.np_w, rec_w), (27, self.np_h, rec_h)):
            self.sets.append(dict(pack_hop_ptr=get(base, P + 1 if P else 0, np.uint32),
                                  hop_node_off=get(base + 1, rec + 1 if P else 0, np.uint32),
                                  hop_layer_off=get(base + 2, rec + 1 if P else 0, np.uint32),
                                  steps=get(base + 3, P, np.uint8), P=P))
        self.var_ptr = get(13, self.n_vars + 1, np.uint32)
        self.var_layers = get(14, self.n_layers, np.uint32)
        self.root_slot = get(15, col.nr_bdds(), np.uint32)
        self.n_bins, self.vars_per_bin, self.n_groups, self.narrow_layers, self.stage_cap = sz(9), sz(10), sz(11), sz(12), sz(13)
        self.bin_ptr = get(16, self.n_bins + 1, np.uint32)
        self.evar = get(17, self.n_layers, np.uint32)
        self.lpos = get(18, self.n_layers, np.uint32)
        self.vpos = get(19, self.n_layers, np.uint32)
        self.pack_group_ptr = get(20, self.np_n + 1 if self.np_n else 0, np.uint32)
        self.grp_layer_off = get(21, self.n_groups + 1, np.uint32)
        self.grp_hop_end = get(22, self.n_groups, np.uint32)
        self.wpb, n_rounds = sz(14), sz(15)
        n_quads = (self.np_n + self.wpb - 1) // self.wpb if self.np_n else 0
        self.quad_round_ptr = get(23, n_quads + 1 if n_quads else 0, np.uint32)
        self.cs_ptr = get(24, n_rounds + 1, np.uint32)
        self.cs_entry = get(25, self.narrow_layers, np.uint32)
        self.cs_slot = get(26, self.narrow_layers, np.uint16)
        self.entry_by_var, self.res_ok, self.res_max_slots, self.res_max_layers = bool(sz(21)), bool(sz(22)), sz(23), sz(24)
        self.pack_hdr = get(35, 8 * self.np_n, np.uint32).reshape(-1, 8)
        self.quad_hdr = get(36, 4 * n_quads, np.uint32).reshape(-1, 4)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.bddmma_layout_destroy(self.h)

    def decode(self):
        """-> dict slot -> (lo_slot|'T'|'B', hi_slot|..., layer_global, head)"""
        out = {}
        widths = {}
        twos = {}
        self.widths = widths
        self.twos = twos
        for wide, S in enumerate(self.sets):
            for p in range(S["P"]):
                q0, q1 = int(S["pack_hop_ptr"][p]), int(S["pack_hop_ptr"][p + 1])
                for q in range(q0, q1):
                    nb, ne = int(S["hop_node_off"][q]), int(S["hop_node_off"][q + 1])
                    lb = int(S["hop_layer_off"][q])
                    lcount = 0
                    for j in range(ne - nb):
                        slot = nb + j
                        if wide:
                            w = int(self.wwords[slot - self.narrow_slots])
                            lo, hi, l = w & 0x1FFFFF, (w >> 21) & 0x1FFFFF, (w >> 42) & 0x1FFFFF
                            head, pad, BOTC, TOPC = bool(w >> 63), False, 0x1FFFFF, 0x1FFFFE
                        else:
                            w = int(self.nwords[slot])
                            lo, hi, pos, lidx, two = w & 511, (w >> 9) & 511, (w >> 18) & 63, (w >> 24) & 63, bool((w >> 30) & 1)
                            pad, TOPC, BOTC = bool(w >> 31), self.pack_width, self.pack_width + 1
                            head = pos == 0
                            if j % 64 == 0:
                                gbase = lcount              # layers of the lower 64-lane groups of this hop
                            if not pad:
                                if head:
                                    lcount += 1
                                l = lcount - 1
                                assert lidx == l - gbase, "layer index inside the lane group"
                                widths[(q, l)] = widths.get((q, l), 0) + 1
                                twos[(q, l)] = two
                            else:
                                l = 0
                        if pad:
                            continue
                        assert q + 1 < q1 or (lo >= TOPC and hi >= TOPC), "last hop must only reach terminals"
                        cv = lambda c: "B" if c == BOTC else ("T" if c == TOPC else ne + c)
                        out[slot] = (cv(lo), cv(hi), lb + l, head, (wide, p, q - q0, j))
        return out


def check_exchange(lay):
    """Invariants of the variable <-> layer exchange tables (layout.hpp, struct Exchange)."""
    L, VB = lay.n_layers, lay.vars_per_bin
    assert sorted(lay.lpos.tolist()) == list(range(L))          # layer -> entry is a permutation
    assert lay.bin_ptr[0] == 0 and lay.bin_ptr[-1] == L and np.all(np.diff(lay.bin_ptr.astype(np.int64)) >= 0)
    ebin = np.searchsorted(lay.bin_ptr, np.arange(L), side="right") - 1
    for l in range(L):
        e = int(lay.lpos[l])
        assert int(lay.evar[e]) == int(lay.layer_var[l]) and int(ebin[e]) == int(lay.evar[e]) // VB
    np.testing.assert_array_equal(lay.vpos, lay.lpos[lay.var_layers])
    # stage groups tile the hops of every narrow pack and hold <= stage_cap contiguous layers
    S = lay.sets[0]
    for p in range(lay.np_n):
        g0, g1 = int(lay.pack_group_ptr[p]), int(lay.pack_group_ptr[p + 1])
        q = int(S["pack_hop_ptr"][p])
        for g in range(g0, g1):
            qe = int(lay.grp_hop_end[g])
            assert qe > q
            l0, l1 = int(S["hop_layer_off"][q]), int(S["hop_layer_off"][qe])
            assert (int(lay.grp_layer_off[g]), int(lay.grp_layer_off[g + 1])) == (l0, l1)
            assert 0 < l1 - l0 <= lay.stage_cap
            # push exchange: inside one bin the group's entries are consecutive and ordered by layer
            e = lay.lpos[l0:l1].astype(np.int64)
            b = ebin[e]
            for bb in np.unique(b):
                ee = e[b == bb]
                assert lay.entry_by_var or np.all(np.diff(ee) == 1)
            q = qe
        assert q == int(S["pack_hop_ptr"][p + 1])
    assert int(lay.grp_layer_off[-1]) == lay.narrow_layers
    # cooperative staging: every narrow layer is staged exactly once; inside a (quad, round) the items are
    # sorted by entry and the slot addresses the owning wave's region
    seen = np.zeros(lay.narrow_layers, bool)
    inv = np.empty(L, np.int64); inv[lay.lpos] = np.arange(L)
    for Q in range(len(lay.quad_round_ptr) - 1):
        for k, r in enumerate(range(int(lay.quad_round_ptr[Q]), int(lay.quad_round_ptr[Q + 1]))):
            c0, c1 = int(lay.cs_ptr[r]), int(lay.cs_ptr[r + 1])
            ent = lay.cs_entry[c0:c1].astype(np.int64)
            assert np.all(np.diff(ent) > 0)
            layers = inv[ent]
            for e_layer, slot in zip(layers, lay.cs_slot[c0:c1].astype(np.int64)):
                w, off = divmod(int(slot), lay.stage_cap)
                p = Q * lay.wpb + w
                g = int(lay.pack_group_ptr[p]) + k
                assert g < int(lay.pack_group_ptr[p + 1]) and int(lay.grp_layer_off[g]) + off == e_layer
                assert not seen[e_layer]
                seen[e_layer] = True
    assert seen.all()
    if lay.entry_by_var:
        # entries are the (variable, bdd)-sorted layers, so a variable's entries are var_ptr[v] .. var_ptr[v + 1]
        np.testing.assert_array_equal(lay.lpos[lay.var_layers], np.arange(L))
        np.testing.assert_array_equal(lay.vpos, np.arange(L))
    # headers of the resident sweeps
    S = lay.sets[0]
    for p in range(lay.np_n):
        q0, q1 = int(S["pack_hop_ptr"][p]), int(S["pack_hop_ptr"][p + 1])
        h = lay.pack_hdr[p]
        assert (int(h[0]), int(h[0] + h[1])) == (int(S["hop_node_off"][q0]), int(S["hop_node_off"][q1]))
        assert (int(h[2]), int(h[2] + h[3])) == (int(S["hop_layer_off"][q0]), int(S["hop_layer_off"][q1]))
        assert int(h[4]) == q0 and int(h[5]) & 0xFFFF == q1 - q0 and int(h[5]) >> 16 == int(S["steps"][p])
        assert int(h[1]) <= lay.res_max_slots and int(h[3]) <= lay.res_max_layers
    for Q in range(len(lay.quad_round_ptr) - 1):
        r0, r1 = int(lay.quad_round_ptr[Q]), int(lay.quad_round_ptr[Q + 1])
        assert int(lay.quad_hdr[Q][2]) == r1 - r0
        if r1 > r0:
            assert (int(lay.quad_hdr[Q][0]), int(lay.quad_hdr[Q][1])) == (int(lay.cs_ptr[r0]), int(lay.cs_ptr[r0 + 1] - lay.cs_ptr[r0]))
    one_group = all(int(lay.pack_group_ptr[p + 1] - lay.pack_group_ptr[p]) == 1 for p in range(lay.np_n))
    short = all(int(lay.pack_hdr[p][5]) & 0xFFFF <= 63 for p in range(lay.np_n))
    if lay.pack_stagger >= 2 and not lay.res_ok:
        pass   # a pack with a BDD that starts below its first hop rules the resident sweeps out (they know roots at the first hop only)
    else:
        assert lay.res_ok == (lay.np_n > 0 and one_group and short)


def check_roundtrip(col, **kw):
    if "exchange_by_variable" not in kw:   # both entry orders: (bin, group, layer) and (variable, bdd)
        off = Layout(col, **kw)
        assert not off.entry_by_var
        check_exchange(off)
        kw = dict(kw, exchange_by_variable=2)
    lay = Layout(col, **kw)
    check_exchange(lay)
    dec = lay.decode()
    ins = col.instr
    n_nonterm = int((ins[:, 2] < BOT).sum())
    assert len(dec) == n_nonterm == lay.n_slots - sum(1 for s in range(lay.n_slots) if s not in dec)
    inv = {int(lay.slot_to_instr[s]): s for s in dec}
    assert len(inv) == n_nonterm
    seen_heads = {}
    for s, (lo, hi, lg, head, where) in dec.items():
        i = int(lay.slot_to_instr[s])
        for side, c in ((0, lo), (1, hi)):
            ci = int(ins[i, side])
            if ins[ci, 2] == TOP:
                assert c == "T"
            elif ins[ci, 2] == BOT:
                assert c == "B"
            else:
                assert c == inv[ci]
        assert lay.layer_var[lg] == int(ins[i, 2])
        seen_heads[lg] = seen_heads.get(lg, 0) + (1 if head else 0)
        if not where[0]:  # narrow: a layer never straddles a 64-lane group
            pass
    assert all(v == 1 for v in seen_heads.values()) and len(seen_heads) == lay.n_layers
    # nodes of one layer are contiguous slots inside one 64-group (narrow)
    by_layer = {}
    for s, (_, _, lg, _, where) in dec.items():
        by_layer.setdefault(lg, []).append((s, where))
    for lg, lst in by_layer.items():
        slots = sorted(s for s, _ in lst)
        assert slots == list(range(slots[0], slots[0] + len(slots)))
        wide, p_, h_, j_ = lst[0][1]
        if not wide:  # the width field of every node of a narrow layer is the layer's width
            S = lay.sets[0]
            q = int(S["pack_hop_ptr"][p_]) + h_
            assert lay.widths[(q, lg - int(S["hop_layer_off"][q]))] == len(slots)
            assert lay.twos[(q, lg - int(S["hop_layer_off"][q]))] == (len(slots) == 2)     # the "two-node layer" bit of the words
        if not lst[0][1][0]:
            js = [w[3] for _, w in lst]
            assert min(js) // 64 == max(js) // 64
    # var CSR sorted by (var, bdd)
    for v in range(lay.n_vars):
        ls = lay.var_layers[lay.var_ptr[v]:lay.var_ptr[v + 1]]
        assert all(lay.layer_var[l] == v for l in ls)
        b = lay.layer_bdd[ls]
        assert np.all(b[1:] > b[:-1])
    # roots
    d = col.delims.astype(np.int64)
    for b in range(col.nr_bdds()):
        assert int(lay.slot_to_instr[lay.root_slot[b]]) == d[b]
    return lay


@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("pw", [64, 128, 256])
def test_roundtrip_golden(name, pw):
    col, _ = load_golden(name)
    check_roundtrip(col, pack_width=pw)


def test_roundtrip_many_packs_and_wide():
    col, _ = random_set_cover(400, 300, 6, seed=1)
    lay = check_roundtrip(col, pack_width=64, vars_per_bin=64, stage_cap=128)
    assert lay.np_n > 1 and lay.np_w == 0 and lay.n_bins == 7 and lay.n_groups > lay.np_n
    # a knapsack row with a wide layer goes to a wide pack
    col2 = BddCollection()
    co = [27, 32, 1, 32, 19, 21, 25, 12, 39, 3, 11, 15, 23, 16, 6, 2, 1, 2]  # widest layer: 97 nodes
    col2.add_linear(co, "<=", sum(co) // 2, list(range(18)))
    col2.add_covering([0, 5, 9])
    lay2 = check_roundtrip(col2, pack_width=64)
    assert lay2.np_w == 1 and lay2.np_n == 1


@pytest.mark.parametrize("pack_width,wide_pack_width", [(64, 192), (128, 256), (64, 0)])
def test_roundtrip_chained_packs(pack_width, wide_pack_width):
    """pack_stagger: BDDs of a pack may start below its first hop, in narrow and in wide packs (hop_root); every node, child, layer and root
    still decodes to the input."""
    rng = np.random.Generator(np.random.PCG64(77))
    col = BddCollection()
    V = 60
    for _ in range(14):                      # layers of 59-135 nodes: wide packs (and narrow ones at pack width 128)
        k = int(rng.integers(15, 20))
        co = rng.integers(1, 40, size=k)
        col.add_linear(co, "<=", int(co.sum() // 2), np.sort(rng.choice(V, size=k, replace=False)))
    for _ in range(30):                      # layers of up to ~40 nodes: narrow packs
        k = int(rng.integers(8, 13))
        co = rng.integers(1, 12, size=k)
        col.add_linear(co, "<=", int(co.sum() // 2), np.sort(rng.choice(V, size=k, replace=False)))
    for _ in range(40):
        col.add_covering(np.sort(rng.choice(V, size=5, replace=False)))
    side_by_side = Layout(col, pack_width=pack_width, wide_pack_width=wide_pack_width, pack_stagger=1)
    chained = check_roundtrip(col, pack_width=pack_width, wide_pack_width=wide_pack_width, pack_stagger=60)
    assert chained.np_w < side_by_side.np_w and chained.np_n < side_by_side.np_n      # fewer, longer packs of both kinds
    assert chained.n_slots <= side_by_side.n_slots + 64 * chained.n_layers             # same nodes (narrow slots may differ by group padding)


def test_layout_rejects_bad_input():
    L = capi.lib()
    col = BddCollection()
    col.add_simplex([0, 1, 2])
    ins = col.instr.copy()
    ins[0, 0] = 3  # root lo -> node of layer 2: skips a layer
    bad = BddCollection.from_arrays(ins, col.delims)
    with pytest.raises(capi.BddMmaError, match="QBDD"):
        Layout(bad)
    with pytest.raises(capi.BddMmaError):
        Layout(col, pack_width=100)


def test_layers_wider_than_the_lds_frontier_form_huge_packs():
    """A BDD with a layer wider than wide_pack_width is not rejected: it goes to a 'huge' pack, whose frontier
    the workgroup-per-pack kernels keep in global memory."""
    col3 = BddCollection()
    co = [27, 32, 1, 32, 19, 21, 25, 12, 39, 3, 11, 15, 23, 16, 6, 2, 1, 2]   # widest layer: 97 nodes
    col3.add_linear(co, "<=", sum(co) // 2, list(range(18)))
    col3.add_covering([0, 5, 9])
    lay = check_roundtrip(col3, pack_width=64, wide_pack_width=64)
    assert (lay.np_n, lay.np_w, lay.np_h) == (1, 0, 1) and lay.huge_pack_width == 104   # 97 rounded up to 8
    from bdd_amd import native
    n = 28
    co = np.random.Generator(np.random.PCG64(1)).integers(1, 5000, size=n)   # widest layer: 7843 nodes
    col4 = native.rows_to_bdd_collection([(co, np.arange(n), "<=", int(co.sum() // 2)), ([1, 1, 1], [0, 1, 2], "=", 1)])
    assert max(col4.layer_widths(0)) > 2048
    lay = Layout(col4)
    assert (lay.np_n, lay.np_w, lay.np_h) == (1, 0, 1) and lay.huge_pack_width >= max(col4.layer_widths(0))


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bdd_amd.solver import bdd_hip_parallel_mma
    ilp = assignment_ilp(3)
    with pytest.raises(capi.BddMmaError, match="no CPU fallback"):
        bdd_hip_parallel_mma(to_bdd_collection(ilp), ilp.objective)


def test_roundtrip_long_bdds_many_groups():
    rng = np.random.Generator(np.random.PCG64(4))
    col = BddCollection()
    col.add_simplex(np.sort(rng.choice(500, size=300, replace=False)))
    for _ in range(60):
        col.add_covering(np.sort(rng.choice(500, size=int(rng.integers(2, 90)), replace=False)))
    lay = check_roundtrip(col, pack_width=64, stage_cap=64, vars_per_bin=64)
    assert lay.n_groups > 10 * lay.np_n and lay.n_hops == 300
    for w in (1, 2, 8):
        check_roundtrip(col, pack_width=64, stage_cap=64, vars_per_bin=64, waves_per_block=w)



def test_packs_of_equal_structure_share_their_node_words():
    """Structure templates: the device keeps each distinct pack word sequence once (layout.hpp: narrow_words_unique);
    reading pack p's words at narrow_word_off[p] reproduces the per-slot words exactly."""
    from bdd_amd.instances import random_set_cover
    col, _ = random_set_cover(4000, 3000, 6, seed=2)          # 3000 covering rows of the same size
    for extra in ([0, 1, 2], [5, 9], [3, 4, 7, 8, 11]):       # and a few rows of other shapes
        col.add_simplex(extra)
    lay = Layout(col, pack_width=64)
    L, h = lay.L, lay.h
    n_unique = int(L.bddmma_layout_size(h, 20))
    uniq = np.zeros(max(n_unique, 1), np.uint32)
    woff = np.zeros(max(lay.np_n, 1), np.uint32)
    capi.check(L.bddmma_layout_copy(h, 31, uniq.ctypes.data_as(C.c_void_p)), None)
    capi.check(L.bddmma_layout_copy(h, 32, woff.ctypes.data_as(C.c_void_p)), None)
    S = lay.sets[0]
    for p in range(lay.np_n):
        s0, s1 = int(S["hop_node_off"][S["pack_hop_ptr"][p]]), int(S["hop_node_off"][S["pack_hop_ptr"][p + 1]])
        np.testing.assert_array_equal(uniq[int(woff[p]):int(woff[p]) + s1 - s0], lay.nwords[s0:s1])
    assert lay.np_n > 50 and n_unique < lay.narrow_slots // 10   # ~all full packs of 6-variable covering rows share one sequence


@pytest.mark.parametrize("pack_width", [64, 128, 256])
def test_uniform_shape_runs_are_packed_in_closed_form(pack_width):
    """Long runs of one BDD shape take the closed-form packing path of build_layout (one simulated pack, stamped out):
    same invariants as the greedy path, packs of a family are full and share one stored word sequence."""
    rng = np.random.Generator(np.random.PCG64(3))
    col = BddCollection()
    V = 4000
    rows7 = np.sort(np.array([rng.choice(V, 7, replace=False) for _ in range(700)]), axis=1).astype(np.uint64)
    rows4 = np.sort(np.array([rng.choice(V, 4, replace=False) for _ in range(300)]), axis=1).astype(np.uint64)
    col.add_covering(rows7)
    for r in rows4:
        col.add_simplex(r)
    col.add_covering(np.sort(rng.choice(V, 5, replace=False)).astype(np.uint64)[None, :])   # a class of one
    col.permute(rng.permutation(col.nr_bdds()))
    lay = check_roundtrip(col, pack_width=pack_width)
    per_pack7 = pack_width // 2                       # covering BDDs are 2 nodes wide
    S = lay.sets[0]
    sizes = [int(S["hop_node_off"][S["pack_hop_ptr"][p + 1]] - S["hop_node_off"][S["pack_hop_ptr"][p]]) for p in range(lay.np_n)]
    full7 = (2 * 7 - 1) * per_pack7                  # 13 nodes per 7-variable covering BDD
    assert sizes.count(full7) == 700 // per_pack7
    n_unique = int(lay.L.bddmma_layout_size(lay.h, 20))
    assert n_unique < sum(sizes) / 3                 # the family's packs share their word sequence


def test_pack_fill_trades_lanes_for_packs():
    """pack_fill: BDDs join an open pack only below `fill` slots per hop (more, emptier packs for latency-bound instances); an empty
    pack still takes any narrow BDD.  Both packing paths (greedy and closed form) honour it, all layout invariants hold."""
    from bdd_amd.instances import random_set_cover
    col, _ = random_set_cover(3000, 2000, 6, seed=7)         # 2000 covering rows, 2 nodes wide
    col.add_linear(np.arange(1, 13), "<=", 30, np.arange(12))  # a knapsack row wider than the fill
    full = check_roundtrip(col, pack_width=64)
    half = check_roundtrip(col, pack_width=64, pack_fill=32)
    quarter = check_roundtrip(col, pack_width=64, pack_fill=16, waves_per_block=2)
    assert full.np_n < half.np_n < quarter.np_n and quarter.np_n >= 3 * full.np_n
    S = quarter.sets[0]
    widths = np.diff(S["hop_node_off"])
    wide_hops = widths > 16                                   # only the pack that starts with the knapsack row may exceed the fill
    packs_over = {int(np.searchsorted(S["pack_hop_ptr"], q, side="right") - 1) for q in np.nonzero(wide_hops)[0]}
    assert len(packs_over) <= 1
    with pytest.raises(capi.BddMmaError, match="pack_fill"):
        Layout(col, pack_width=64, pack_fill=65)


def test_automatic_stagger_leaves_flat_rows_side_by_side():
    """flat BDDs (covering rows) never narrow again, so chaining them behind a full pack only builds a staircase — more
    and longer packs, one pack per workgroup, no resident sweeps.  With keep_bdd_order the rows bypass the closed-form packing of uniform
    shape runs and go through PackBuilder::add: the automatic mode must give the side-by-side layout there too, at a size where it is on."""
    col, _ = random_set_cover(300_000, 150_000, 10, seed=1)     # 2.85 M nodes: automatic chaining allows 13 hops per pack
    def hops(**kw):
        lay = Layout(col, **kw)
        s = lay.sets[0]
        return lay.np_n, int(s["pack_hop_ptr"][-1]), int(np.diff(s["pack_hop_ptr"]).max()), lay.wpb
    auto, side, forced = hops(keep_bdd_order=1), hops(keep_bdd_order=1, pack_stagger=1), hops(keep_bdd_order=1, pack_stagger=30)
    assert auto == side                        # same packs, same total wave-hops, four packs per workgroup
    assert auto[2] == 10 and auto[3] == 4
    assert forced[1] > 3 * side[1] and forced[2] > 10 and forced[3] == 1   # what the explicit option still does (and the old automatic mode did)
    assert hops() == side                      # grouped by shape (closed-form packing): unchanged


def test_packs_are_formed_longest_first():
    """Round 5: the blocks of a launch start in pack order and a sweep ends with its last wave, so (a) diamond-shaped BDDs of small shape
    classes (general linear rows) come first, widest peak first — neighbours of similar peak chain at shorter offsets — and (b) the other
    classes follow longest BDDs first.  keep_bdd_order = 2 keeps the order before (classes by first appearance), 1 the input order; every
    order decodes back to the input."""
    from bdd_amd.instances import random_set_cover_mixed
    col, _ = random_set_cover_mixed(6_000, 4_000, 3, 9, seed=4)           # rows of 3 ... 9 variables, interleaved at random
    def pack_hops(lay):
        return np.diff(lay.sets[0]["pack_hop_ptr"]).astype(int)
    new, old = check_roundtrip(col), Layout(col, keep_bdd_order=2)
    h_new, h_old = pack_hops(new), pack_hops(old)
    assert np.all(h_new[1:] <= h_new[:-1]) and h_new[0] == 9 and h_new[-1] == 3         # longest packs first
    assert not np.all(h_old[1:] <= h_old[:-1])                                         # first appearance: whatever the input starts with
    assert sorted(h_new) == sorted(h_old) and new.narrow_slots == old.narrow_slots      # the same packs, another order
    # general linear rows (every BDD its own shape) behind covering rows in the input: the chained packs come first, and the peak widths of
    # the BDDs they hold do not increase from pack to pack
    rng = np.random.Generator(np.random.PCG64(5))
    col = BddCollection()
    V = 400
    for _ in range(600):
        col.add_covering(np.sort(rng.choice(V, size=6, replace=False)))
    rows = []
    for _ in range(300):
        k = 12
        co = rng.integers(1, 30, size=k)
        col.add_linear(co, "<=", int(co.sum() // 2), np.sort(rng.choice(V, size=k, replace=False)))
    lay = check_roundtrip(col, pack_width=64, pack_stagger=36)
    S = lay.sets[0]
    hops = pack_hops(lay)
    first_short = int(np.argmax(hops <= 6))                                            # covering rows: 6 hops
    assert first_short > 0 and np.all(hops[:first_short] > 6) and np.all(hops[first_short:] <= 6)
    d = col.delims.astype(np.int64)
    ins = col.instr
    peak = {}
    for b in range(600, 900):                                                          # widest layer of every knapsack BDD
        idx = ins[d[b]:d[b + 1] - 2, 2]
        peak[b] = int(np.unique(idx, return_counts=True)[1].max())
    narrow = {b: w for b, w in peak.items() if w <= 64}
    first_slot = {b: int(lay.root_slot[b]) for b in narrow}                            # slots grow with the pack index
    by_pack = sorted(narrow, key=lambda b: first_slot[b])
    pack_of = {b: int(np.searchsorted(S["hop_node_off"][S["pack_hop_ptr"]], first_slot[b], side="right") - 1) for b in by_pack}
    per_pack = {}
    for b in by_pack:
        per_pack.setdefault(pack_of[b], []).append(narrow[b])
    maxima = [max(v) for _, v in sorted(per_pack.items())]
    minima = [min(v) for _, v in sorted(per_pack.items())]
    assert all(minima[i] >= maxima[i + 1] for i in range(len(maxima) - 1))             # widest first, pack by pack
    older = Layout(col, pack_width=64, pack_stagger=36, keep_bdd_order=2)                # the order before: covering rows (first in the input) first
    assert pack_hops(older)[0] == 6 and older.n_layers == lay.n_layers
    # (fewer wave-hops than the order before is a statistical statement — 5.6 % on 40 000 rows, profiles/r05_pack_order_widebench.txt —, not one about 300)


def test_large_instances_get_eight_packs_per_workgroup():
    """The entries a sweep workgroup stages per bin form one run in the entry arrays; where four packs give runs shorter than 4.5 entries
    (many bins: 105 M nodes, or here a small bin size; threshold 4.5 entries) the automatic choice is eight packs per workgroup (layout.cpp; measured +5-9 % at
    105 M nodes).  Few bins, few packs or an explicit option: unchanged.  (Eight-pack tables decode back to the input: test_roundtrip_long_bdds_many_groups.)"""
    col, _ = random_set_cover(300_000, 280_000, 10, seed=2)     # 4 375 packs of 128 slots
    assert Layout(col).wpb == 4                                 # 74 bins: a workgroup of four stages ~35 entries per bin
    many = Layout(col, vars_per_bin=256)                        # 1 172 bins: ~2 entries
    assert many.np_n >= 4096 and many.wpb == 8
    assert Layout(col, vars_per_bin=256, waves_per_block=4).wpb == 4
    small, _ = random_set_cover(100_000, 50_000, 10, seed=2)
    assert Layout(small, vars_per_bin=64).wpb <= 4              # too few packs for workgroups of eight to fill the chip


def test_pack_stagger_is_bounded():
    col, _ = random_set_cover(200, 100, 5, seed=2)
    with pytest.raises(capi.BddMmaError, match="pack_stagger"):
        Layout(col, pack_stagger=70_000)


def res2_records(lay, real_size):
    info = np.zeros(5, np.uint32)
    capi.check(lay.L.bddmma_layout_res2_records(lay.h, real_size, info.ctypes.data_as(C.c_void_p), None, None), None)
    words, off = np.zeros(max(int(info[1]), 1), np.uint32), np.zeros(max(lay.np_n, 1), np.uint32)
    capi.check(lay.L.bddmma_layout_res2_records(lay.h, real_size, info.ctypes.data_as(C.c_void_p), words.ctypes.data_as(C.c_void_p),
                                                off.ctypes.data_as(C.c_void_p)), None)
    return bool(info[0]), words[:int(info[1])].reshape(-1, 4), off[:lay.np_n], int(info[2]), int(info[3])


@pytest.mark.parametrize("real_size", [4, 8])
def test_res2_records_restate_the_node_words(real_size):
    """The per-lane records of the second-generation resident sweeps (layout.hpp: Res2Records) against the 4-byte node words they are
    derived from: same children (as slots of the pack), same layer, head and two-node flags; padding lanes point at harmless places;
    two-node layers start at even lanes (the DPP pair swap of kernels.hpp: pair_min_aligned relies on it)."""
    col = BddCollection()
    rng = np.random.Generator(np.random.PCG64(9))
    for _ in range(300):
        k = int(rng.integers(2, 12))
        vs = np.sort(rng.choice(400, size=k, replace=False))
        (col.add_covering if rng.random() < 0.6 else col.add_simplex)(vs)
    for _ in range(20):
        col.add_simplex([int(rng.integers(0, 400))])          # single-variable rows: one-node layers between the pairs
    lay = Layout(col, pack_width=64)
    ok, rec, rec_off, ns, nl = res2_records(lay, real_size)
    assert ok and lay.pack_width == 64
    S = real_size
    T_OFF, F_OFF = 0, (ns + 4) * S
    N = lay.sets[0]
    seen_two = seen_one = 0
    for p in range(lay.np_n):
        q0, q1 = int(N["pack_hop_ptr"][p]), int(N["pack_hop_ptr"][p + 1])
        s0, l0 = int(N["hop_node_off"][q0]), int(N["hop_layer_off"][q0])
        for h in range(q1 - q0):
            nb, ne = int(N["hop_node_off"][q0 + h]) - s0, int(N["hop_node_off"][q0 + h + 1]) - s0
            lb = int(N["hop_layer_off"][q0 + h]) - l0
            for j in range(64):
                r = rec[int(rec_off[p]) + h * 64 + j]
                w = int(lay.nwords[s0 + nb + j]) if j < ne - nb else 1 << 31
                if w >> 31:
                    assert r[3] == 0xFFFFFFFF
                    assert r[0] & 0xFFFF == r[0] >> 16 == T_OFF + S * (ns + 1)                       # cost to terminal +inf on both sides
                    assert r[1] & 0xFFFF == r[1] >> 16 == F_OFF + S * (ns + j)                       # pushes into the lane's own dummy entry
                    assert (r[2] >> 16) == S * (ns + j)                                             # own slot past the pack: stores are dropped
                    continue
                lo, hi, pos, lidx, two = w & 511, (w >> 9) & 511, (w >> 18) & 63, (w >> 24) & 63, (w >> 30) & 1
                for side, c in ((0, lo), (1, hi)):
                    t = (r[0] >> (16 * side)) & 0xFFFF
                    f = (r[1] >> (16 * side)) & 0xFFFF
                    if c < 64:
                        assert t == T_OFF + S * (ne + c) and f == F_OFF + S * (ne + c)
                    else:
                        assert t == T_OFF + S * (ns + (0 if c == 64 else 1)) and f == F_OFF + S * (ns + j)
                ll = lb + lidx
                assert r[2] & 0xFFFF == ll * 2 * S and r[2] >> 16 == S * (nb + j)
                assert r[3] & 0xFFFF == (ll * 2 * S if pos == 0 else 0xFFF0)
                assert (r[3] >> 16) == two
                if two:
                    assert (j - pos) % 2 == 0                                                       # aligned pair
                    seen_two += 1
                else:
                    seen_one += 1
    assert seen_two > 1000 and seen_one > 300
    # packs of one structure template share their records; wider packs / wider layers have none
    col2, _ = random_set_cover(3000, 2000, 8, seed=3)
    lay2 = Layout(col2, pack_width=64)
    ok2, rec2, off2, _, _ = res2_records(lay2, real_size)
    assert ok2 and len(set(off2.tolist())) < lay2.np_n / 4 and rec2.shape[0] < lay2.np_n * 8 * 64 / 4
    assert not res2_records(Layout(col2, pack_width=128), real_size)[0]
    col3 = BddCollection()
    col3.add_linear([3, 5, 7, 2, 4, 6, 1], "<=", 14, np.arange(7))
    col3.add_covering([0, 1, 2])
    assert not res2_records(Layout(col3, pack_width=64), real_size)[0]


@pytest.mark.parametrize("real_size,pack_width", [(4, 64), (4, 128), (8, 128), (4, 256)])
def test_stream_records_restate_the_node_words(real_size, pack_width):
    """The records of the second-generation streaming sweeps (layout.hpp: StreamRecords): offsets into the hop's LDS buffers, the layer's
    index inside the hop (over all 64-lane groups), the head-only store offset, against the node words."""
    col = BddCollection()
    rng = np.random.Generator(np.random.PCG64(19))
    for _ in range(900):
        k = int(rng.integers(2, 12))
        vs = np.sort(rng.choice(600, size=k, replace=False))
        (col.add_covering if rng.random() < 0.6 else col.add_simplex)(vs)
    for _ in range(30):
        col.add_simplex([int(rng.integers(0, 600))])
    lay = Layout(col, pack_width=pack_width)
    info = np.zeros(5, np.uint32)
    capi.check(lay.L.bddmma_layout_stream_records(lay.h, real_size, info.ctypes.data_as(C.c_void_p), None, None), None)
    assert info[0] == 1
    words, off = np.zeros(int(info[1]), np.uint32), np.zeros(lay.np_n, np.uint32)
    capi.check(lay.L.bddmma_layout_stream_records(lay.h, real_size, info.ctypes.data_as(C.c_void_p), words.ctypes.data_as(C.c_void_p),
                                                  off.ctypes.data_as(C.c_void_p)), None)
    rec = words.reshape(-1, 4)
    S, W = real_size, pack_width
    N = lay.sets[0]
    checked = 0
    for p in range(lay.np_n):
        q0, q1 = int(N["pack_hop_ptr"][p]), int(N["pack_hop_ptr"][p + 1])
        s0 = int(N["hop_node_off"][q0])
        for h in range(q1 - q0):
            nb, ne = int(N["hop_node_off"][q0 + h]) - s0, int(N["hop_node_off"][q0 + h + 1]) - s0
            heads = 0
            for j in range(W):
                r = rec[int(off[p]) + h * W + j]
                w = int(lay.nwords[s0 + nb + j]) if j < ne - nb else 1 << 31
                if w >> 31:
                    assert r[3] == 0x80000000 and r[0] & 0xFFFF == r[0] >> 16 == (W + 1) * S
                    assert r[1] & 0xFFFF == r[1] >> 16 == (W + 2 + j) * S and r[2] >> 16 == 0xFFF0
                    continue
                lo, hi, pos, two = w & 511, (w >> 9) & 511, (w >> 18) & 63, (w >> 30) & 1
                assert r[0] & 0xFFFF == lo * S and r[0] >> 16 == hi * S
                assert r[1] & 0xFFFF == (lo * S if lo < W else (W + 2 + j) * S) and r[1] >> 16 == (hi * S if hi < W else (W + 2 + j) * S)
                if pos == 0:
                    heads += 1
                lq = (heads - 1) * 2 * S                                   # layers are numbered left to right over the whole hop
                assert r[2] & 0xFFFF == lq and r[2] >> 16 == (lq if pos == 0 else 0xFFF0) and r[3] == (two | (pos << 8))
                assert not two or (j - pos) % 2 == 0
                checked += 1
            assert heads == int(N["hop_layer_off"][q0 + h + 1]) - int(N["hop_layer_off"][q0 + h])
    assert checked > 5000
    # layers wider than two nodes and staggered packs have records too (position inside the layer for the LDS segmented minimum)
    col3 = BddCollection()
    for _ in range(40):
        co = rng.integers(1, 9, size=9)
        col3.add_linear(co, "<=", int(co.sum() // 2), np.sort(rng.choice(60, size=9, replace=False)))
    lay3 = Layout(col3, pack_width=64, pack_stagger=24)
    capi.check(lay3.L.bddmma_layout_stream_records(lay3.h, real_size, info.ctypes.data_as(C.c_void_p), None, None), None)
    assert info[0] == 1
    w3 = np.zeros(int(info[1]), np.uint32)
    capi.check(lay3.L.bddmma_layout_stream_records(lay3.h, real_size, info.ctypes.data_as(C.c_void_p), w3.ctypes.data_as(C.c_void_p), None), None)
    r3 = w3.reshape(-1, 4)
    real = (r3[:, 3] & 0x80000000) == 0
    assert ((r3[real, 3] >> 8) & 63).max() >= 3          # nodes at position >= 3 of their layer exist


@pytest.mark.parametrize("real_size", [4, 8])
def test_layer_records_restate_the_node_words(real_size):
    """The records of the third-generation streaming sweeps (layout.hpp: LayerRecords): lane l of a hop owns the hop's l-th layer — its one or
    two nodes (neighbouring slots), their children as offsets into a hop buffer with per-lane sink entries, the store offsets — against the
    node words; and the shapes the builder must refuse."""
    col = BddCollection()
    rng = np.random.Generator(np.random.PCG64(23))
    for _ in range(900):
        k = int(rng.integers(2, 14))
        vs = np.sort(rng.choice(600, size=k, replace=False))
        (col.add_covering if rng.random() < 0.6 else col.add_simplex)(vs)
    for _ in range(30):
        col.add_simplex([int(rng.integers(0, 600))])
    lay = Layout(col, pack_width=128)
    info = np.zeros(5, np.uint32)
    capi.check(lay.L.bddmma_layout_layer_records(lay.h, real_size, info.ctypes.data_as(C.c_void_p), None, None), None)
    assert info[0] == 1
    words, off = np.zeros(int(info[1]), np.uint32), np.zeros(lay.np_n, np.uint32)
    capi.check(lay.L.bddmma_layout_layer_records(lay.h, real_size, info.ctypes.data_as(C.c_void_p), words.ctypes.data_as(C.c_void_p),
                                                 off.ctypes.data_as(C.c_void_p)), None)
    rec = words.reshape(-1, 4)
    S, W = real_size, 128
    N = lay.sets[0]
    checked = two_node = 0
    for p in range(lay.np_n):
        q0, q1 = int(N["pack_hop_ptr"][p]), int(N["pack_hop_ptr"][p + 1])
        s0 = int(N["hop_node_off"][q0])
        for h in range(q1 - q0):
            nb, ne = int(N["hop_node_off"][q0 + h]) - s0, int(N["hop_node_off"][q0 + h + 1]) - s0
            nl = int(N["hop_layer_off"][q0 + h + 1]) - int(N["hop_layer_off"][q0 + h])
            layers = []          # [(slot of the head, [words of the layer's nodes])] in slot order
            for j in range(ne - nb):
                w = int(lay.nwords[s0 + nb + j])
                if w >> 31:
                    continue
                if (w >> 18) & 63 == 0:
                    layers.append((j, [w]))
                else:
                    assert layers[-1][0] == j - 1 and len(layers[-1][1]) == 1      # second node: the slot behind its head
                    layers[-1][1].append(w)
            assert len(layers) == nl <= 64
            for l in range(64):
                r = [int(x) for x in rec[int(off[p]) + h * 64 + l]]
                top, bot = (W + 2 * l) * S, (W + 2 * l + 1) * S
                ch = lambda c: (c if c < W else W + 2 * l + (c - W)) * S
                if l >= nl:
                    assert r == [bot | bot << 16, bot | bot << 16, top, 0xFFF0 | 0xFFF0 << 16]
                    continue
                j, ws = layers[l]
                a = ws[0]
                assert r[0] == ch(a & 511) | ch((a >> 9) & 511) << 16
                assert r[2] & 0xFFFF == j * S and r[2] >> 16 == (2 | (1 if len(ws) == 2 else 0))
                if len(ws) == 2:
                    b = ws[1]
                    assert r[1] == ch(b & 511) | ch((b >> 9) & 511) << 16 and r[3] == (j * S) | ((j + 1) * S) << 16
                    two_node += 1
                else:
                    assert r[1] == bot | bot << 16 and r[3] == (j * S) | 0xFFF0 << 16
                checked += 1
    assert checked > 3000 and two_node > 1000
    # refused: other pack widths, layers wider than two nodes, staggered packs
    for lay2 in (Layout(col, pack_width=64), Layout(col, pack_width=256)):
        capi.check(lay2.L.bddmma_layout_layer_records(lay2.h, real_size, info.ctypes.data_as(C.c_void_p), None, None), None)
        assert info[0] == 0
    col3 = BddCollection()
    for _ in range(40):
        co = rng.integers(1, 9, size=9)
        col3.add_linear(co, "<=", int(co.sum() // 2), np.sort(rng.choice(60, size=9, replace=False)))
    for lay3 in (Layout(col3, pack_width=128), Layout(col3, pack_width=128, pack_stagger=24)):
        capi.check(lay3.L.bddmma_layout_layer_records(lay3.h, real_size, info.ctypes.data_as(C.c_void_p), None, None), None)
        assert info[0] == 0


def seg_exchange(lay, threads, real_size):
    info = np.zeros(8, np.uint32)
    capi.check(lay.L.bddmma_layout_seg_exchange(lay.h, threads, real_size, info.ctypes.data_as(C.c_void_p), None, None, None), None)
    b, p, t = np.zeros(max(int(info[1]), 1), np.uint32), np.zeros(max(int(info[2]), 1), np.uint16), np.zeros(max(int(info[3]), 1), np.uint32)
    capi.check(lay.L.bddmma_layout_seg_exchange(lay.h, threads, real_size, info.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                                p.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p)), None)
    return bool(info[0]), b[: info[1]].reshape(-1, 4), p[: info[2]], t[: info[3]].reshape(-1, 2), info[4:7].tolist()


@pytest.mark.parametrize("threads,real_size,vars_per_bin", [(256, 4, 64), (256, 8, 128), (512, 4, 0), (1024, 8, 192)])
def test_seg_exchange_schedule_is_the_per_variable_gather(threads, real_size, vars_per_bin):
    """layout.hpp: SegExchange — the fixed schedule of k_exchange_seg, executed here in numpy exactly as the kernel walks it, gives the
    per-variable sums of k_delta_gather (same entries, same order), every entry gets its variable's slot, and the runs are balanced."""
    from bdd_amd.instances import random_set_cover
    col, _ = random_set_cover(900, 700, 6, seed=5)
    lay = Layout(col, vars_per_bin=vars_per_bin)
    ok, hdr, perm, thr, (max_e, max_s, max_g) = seg_exchange(lay, threads, real_size)
    assert ok and hdr.shape[0] == lay.n_bins and thr.shape[0] == lay.n_bins * threads
    VEC = 16 // real_size
    rng = np.random.Generator(np.random.PCG64(3))
    mm = rng.standard_normal(lay.n_layers).astype(np.float32 if real_size == 4 else np.float64)
    mm[rng.random(lay.n_layers) < 0.2] = 0
    dt = mm.dtype.type
    seen = np.zeros(lay.n_layers, np.int64)
    pairs_by_entry = np.zeros((lay.n_layers, 2), mm.dtype)
    for b in range(lay.n_bins):
        first, gy, e0, E = (int(x) for x in hdr[b])
        groups, slots = gy & 0xFF, gy >> 8
        assert e0 == int(lay.bin_ptr[b]) and E == int(lay.bin_ptr[b + 1]) - e0 and groups <= 4
        Z = (E + VEC - 1) // VEC * VEC
        lds = np.zeros(Z + 1, mm.dtype); lds[:E] = mm[e0:e0 + E]
        tile, cnt, slot_of = np.zeros((slots, 2), mm.dtype), np.zeros(slots, np.int64), np.full(Z + 1, -1, np.int64)
        runs = []
        for t in range(threads):
            ends, slot = int(thr[b * threads + t][0]), int(thr[b * threads + t][1])
            p = np.concatenate([perm[(first + g * threads + t) * 8:(first + g * threads + t) * 8 + 8] for g in range(groups)]).astype(np.int64) if groups else np.zeros(0, np.int64)
            run_len = ends.bit_length()
            assert np.all(p[run_len:] == Z) and np.all(p[:run_len] < E)
            runs.append(run_len)
            lo, hi, n = dt(0), dt(0), 0
            for k in range(groups * 8):
                m = lds[p[k]]
                if m > 0:
                    hi = dt(hi + m)
                elif m < 0:
                    lo = dt(lo + -m)
                n += 1
                if (ends >> k) & 1:
                    tile[slot] = (lo, hi); cnt[slot] = n; slot += 1
                    lo, hi, n = dt(0), dt(0), 0
                if k < run_len:
                    slot_of[p[k]] = int(thr[b * threads + t][1]) + bin(ends & ((1 << k) - 1)).count("1")
            seen[e0 + p[:run_len]] += 1
        assert np.all(cnt > 0) and np.all(slot_of[:E] >= 0)
        tile = (tile / cnt[:, None].astype(mm.dtype)).astype(mm.dtype)
        pairs_by_entry[e0:e0 + E] = tile[slot_of[:E]]
        # balance: no run is longer than the mean by more than the largest variable
        nz = [r for r in runs if r]
        per_var = np.diff(lay.var_ptr.astype(np.int64))[b * lay.vars_per_bin:(b + 1) * lay.vars_per_bin]
        assert max(runs) <= -(-E // threads) + int(per_var.max()) and max(runs) <= 32 and (not nz or len(nz) == min(threads, int((per_var > 0).sum())))
    assert np.all(seen == 1)                                   # every entry is in exactly one run
    # the reference of the schedule: k_delta_gather's sum per variable in (variable, bdd) order, then the broadcast
    for v in range(lay.n_vars):
        k0, k1 = int(lay.var_ptr[v]), int(lay.var_ptr[v + 1])
        lo, hi = dt(0), dt(0)
        for e in lay.vpos[k0:k1]:
            m = mm[e]
            if m > 0:
                hi = dt(hi + m)
            elif m < 0:
                lo = dt(lo + -m)
        if k1 > k0:
            want = np.array([lo / dt(k1 - k0), hi / dt(k1 - k0)], mm.dtype)
            assert np.array_equal(pairs_by_entry[lay.vpos[k0:k1]], np.tile(want, (k1 - k0, 1)))


def test_seg_exchange_refuses_what_it_cannot_hold():
    from bdd_amd.instances import random_set_cover
    col, _ = random_set_cover(40, 400, 6, seed=1)              # 60 entries per variable: runs of more than 32 with 64 threads per 64 variables
    assert not seg_exchange(Layout(col, vars_per_bin=64), 64, 4)[0]
    assert not seg_exchange(Layout(col, exchange_by_variable=2), 256, 4)[0]


def test_layout_for_a_smaller_chip_differs_where_the_rules_read_the_chip():
    """the CPU-side layout entry point can be asked for the layout bddmma_create builds on the device at hand
    (bddmma_device_chip -> bddmma_layout_create_for_chip).  ~2 200 narrow packs of 64 slots: one pack per workgroup where the chip holds
    them all at once (256 CUs; layout.cpp: 3 700 packs per 256 CUs, scaled with the CU count), the streaming sweeps' four on a
    32-CU partition.  Zeros mean MI355X: the same layout as bddmma_layout_create."""
    from bdd_amd.instances import random_set_cover
    col, _ = random_set_cover(150_000, 70_400, 10, seed=5)
    default = Layout(col, pack_width=64)
    same = Layout(col, pack_width=64, chip=(4, 0, 0))
    small = Layout(col, pack_width=64, chip=(4, 32, 160 * 1024))
    assert default.pack_width == 64 and 2048 <= default.np_n <= 3700
    assert (default.wpb, same.wpb) == (1, 1) and small.wpb == 4
    np.testing.assert_array_equal(default.nwords, same.nwords)
    np.testing.assert_array_equal(default.cs_entry, same.cs_entry)
    assert small.n_slots == default.n_slots and small.n_layers == default.n_layers
    L, h = capi.lib(), C.c_void_p()
    assert L.bddmma_layout_create_for_chip(C.byref(h), None, None, 0, None, 2, 0, 0) != 0   # real_size is 4 or 8


@pytest.mark.parametrize("packs,real_size,want", [(80, 8, 640), (100, 8, 448), (116, 8, 384), (126, 8, 320), (140, 8, 640), (100, 4, 640)])
def test_stage_groups_shrink_to_fit_the_launch_into_one_round_of_workgroups(packs, real_size, want):
    """layout.cpp (round 6): rows of 50 variables are 32 BDDs per 64-slot pack and 1 600 layers per pack — three stage groups of <= 640.  On an 8-CU chip
    with 160 KB of LDS per CU the forward sweep's one-pack workgroups (16 B per staged pair + 3 904 B static + ~0.6 KB) fit 11 per CU with groups of
    640 layers, 13 with 448, 15 with 384, 16 (the register budget) with 320: the rule takes the LARGEST size that puts every workgroup of the launch on
    the chip at once, keeps 640 where that already fits (80 packs <= 88) or no size achieves it (140 > 128), and never fires in float (20 per CU either way).
    An explicit stage_cap is never overridden."""
    from bdd_amd.instances import random_set_cover
    col, _ = random_set_cover(4000, 32 * packs, 50, seed=packs)
    chip = (real_size, 8, 160 * 1024)
    lay = Layout(col, pack_width=64, chip=chip)
    assert lay.np_n == packs and lay.wpb == 1
    assert lay.stage_cap == want
    assert Layout(col, pack_width=64, stage_cap=640, chip=chip).stage_cap == 640
    check_roundtrip(col, pack_width=64, chip=chip)
    # four packs per workgroup: 57 KB with groups of 640 layers (two workgroups per CU), 53 KB with 576 (three) — where no size fits the launch into one round
    # the rule takes that one step; a launch that fits anyway, and float, keep 640
    four = Layout(col, pack_width=64, waves_per_block=4, chip=chip)
    assert four.wpb == 4
    n_wg = (packs + 3) // 4
    per_cu = lambda cap: min(160 * 1024 // (4 * (cap * 2 * real_size + (4 * 66 + 128) * real_size + 768) + 640), (16 if real_size == 8 else 20) // 4)
    fits = [cap for cap in range(640, 255, -64) if n_wg <= 8 * per_cu(cap)]
    want4 = 640 if fits[:1] == [640] else fits[0] if fits else 576 if (per_cu(640), per_cu(576)) == (2, 3) else 640
    assert four.stage_cap == want4, (n_wg, four.stage_cap, want4)
    assert want4 == {(80, 8): 576, (100, 8): 384, (116, 8): 384, (126, 8): 384, (140, 8): 576}.get((packs, real_size), 640)


def test_device_chip_query_without_a_device_is_an_error_not_a_default():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_device_abi.py")
    n, lds, thr = C.c_uint32(7), C.c_uint32(7), C.c_uint64(7)
    assert capi.lib().bddmma_device_chip(0, C.byref(n), C.byref(lds), C.byref(thr)) != 0
    assert (n.value, lds.value, thr.value) == (7, 7, 7)
