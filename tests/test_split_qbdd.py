"""Long-BDD splitting: bdd_collection::split_qbdd restated in bdd_amd/bdd_collection.py,
pinned node-for-node against the reference's own compiled code (tests/golden/split_*.npz, made by
oracle/make_golden.py: record_split) and checked semantically by brute force — CPU only."""
import itertools

import numpy as np
import pytest

from bdd_amd import BddCollection
from bdd_amd.ilp import compute_split_length, split_long_bdds
from oracle import oracle as O
from util import GOLDEN_DIR, SPLIT_GOLDEN, canonical_nodes, collection_from_arrays

needs_ref = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("name", SPLIT_GOLDEN)
def test_split_matches_reference_golden(name):
    z = np.load(f"{GOLDEN_DIR}/{name}.npz")
    col = collection_from_arrays(z["in_instr"], z["in_delims"])
    implication = bool(z["implication"]) if "implication" in z else False
    new_nrs, next_aux = col.split_qbdd(0, int(z["chunk"]), int(z["aux0"]), implication)
    assert len(new_nrs) == int(z["n_new"]) and next_aux == int(z["next_aux"])
    if len(new_nrs) > 1:
        col.remove([0])
    np.testing.assert_array_equal(col.delims, z["out_delims"])
    if implication and len(new_nrs) == -(-len(collection_from_arrays(z["in_instr"], z["in_delims"]).layer_widths(0)) // int(z["chunk"])) + 1:
        # the chunks node for node; the implication BDD (last) up to the order of the nodes inside its layers, which
        # in the reference comes out of bdd_mgr's bdd_and / make_qbdd
        ref = collection_from_arrays(z["out_instr"], z["out_delims"])
        last = int(z["out_delims"][-2])
        np.testing.assert_array_equal(col.instr[:last], z["out_instr"][:last])
        assert col.layer_widths(col.nr_bdds() - 1) == ref.layer_widths(ref.nr_bdds() - 1)
        assert canonical_nodes(col, col.nr_bdds() - 1) == canonical_nodes(ref, ref.nr_bdds() - 1)
    else:
        np.testing.assert_array_equal(col.instr, z["out_instr"])


def test_implication_bdd_semantics():
    """The implication BDD accepts exactly the auxiliary assignments that some path of the original BDD induces
    (it is the conjunction of necessary conditions; on these instances it is exact), so adding it does not change the
    feasible set of the split system."""
    n = 8
    col = BddCollection()
    col.add_linear([2, 1, 3, 1, 2, 1, 2, 3], "<=", 7, list(range(n)))
    want = {x for x in itertools.product((0, 1), repeat=n) if col.evaluate(0, x)}
    nsplit, n_all = split_long_bdds(col, n, 2, with_implication_bdd=True)
    assert nsplit == 1 and col.nr_bdds() == 4 + 1
    imp = col.nr_bdds() - 1
    assert set(col.variables(imp)) == set(range(n, n_all))       # over the auxiliary variables only
    # every feasible x has exactly one auxiliary completion, and the implication BDD accepts it
    chunks = list(range(col.nr_bdds() - 1))
    aux_n = n_all - n
    assert aux_n <= 16
    for x in list(want)[:12]:
        sols = [a for a in itertools.product((0, 1), repeat=aux_n) if all(col.evaluate(b, list(x) + list(a)) for b in chunks)]
        assert len(sols) == 1 and col.evaluate(imp, list(x) + list(sols[0]))
    # and it rejects one-hot assignments that no path realises (if any exist for this instance)
    accepted = sum(col.evaluate(imp, [0] * n + list(a)) for a in itertools.product((0, 1), repeat=aux_n))
    assert 0 < accepted < 2 ** aux_n


@needs_ref
def test_split_matches_reference_live():
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(12):
        k = int(rng.integers(6, 14))
        co = rng.integers(1, 5, size=k)
        rhs = int(co.sum() // 2)
        rc = O.RefCollection()
        assert rc.add_linear(co, ["<=", ">=", "="][int(rng.integers(0, 3))], rhs, np.arange(k)) >= -2
        if rc.nr_bdds() == 0:
            continue
        col = rc.export()
        widths = col.layer_widths(0)
        chunk = int(rng.integers(2, max(3, k - 1)))
        # the reference asserts that cut layers have more than one node (bdd_collection.cpp:583)
        if any(widths[c] == 1 for c in range(chunk, len(widths), chunk)):
            continue
        n, na = rc.split_qbdd(0, chunk, k + 3)
        new_nrs, na2 = col.split_qbdd(0, chunk, k + 3)
        if len(new_nrs) > 1:
            col.remove([0])
        ref = rc.export()
        assert (n, na) == (len(new_nrs), na2)
        np.testing.assert_array_equal(col.instr, ref.instr)
        np.testing.assert_array_equal(col.delims, ref.delims)


def feasible_projection(col, n_orig, n_all):
    """set of x in {0,1}^n_orig for which some assignment of the auxiliary variables satisfies every BDD"""
    ok = set()
    for x in itertools.product((0, 1), repeat=n_orig):
        for a in itertools.product((0, 1), repeat=n_all - n_orig):
            full = list(x) + list(a)
            if all(col.evaluate(b, full) for b in range(col.nr_bdds())):
                ok.add(x)
                break
    return ok


@pytest.mark.parametrize("chunk", [1, 2, 3, 5])
def test_split_preserves_the_feasible_set(chunk):
    n = 5 if chunk == 1 else 7
    col = BddCollection()
    col.add_linear([2, 1, 3, 1, 2, 1, 2][:n], "<=", 4 if chunk == 1 else 6, list(range(n)))
    want = {x for x in itertools.product((0, 1), repeat=n) if col.evaluate(0, x)}
    nsplit, n_all = split_long_bdds(col, n, chunk)
    assert nsplit == 1 and col.nr_bdds() == -(-n // chunk)
    assert n_all - n <= 13, "keep the enumeration small"
    assert feasible_projection(col, n, n_all) == want
    # exactly one auxiliary assignment per feasible x (the auxiliary variables are one-hot per cut)
    for x in list(want)[:5]:
        cnt = sum(all(col.evaluate(b, list(x) + list(a)) for b in range(col.nr_bdds()))
                  for a in itertools.product((0, 1), repeat=n_all - n))
        assert cnt == 1


def test_split_with_width_one_cut_layer():
    # x0 + x1 = 1 followed by free structure: covering BDD's first layer has a single node; chunk 1 cuts in
    # front of width-2 layers only, but a simplex over 2 variables cut at 1 sees width 2; force a width-1 cut
    # with a BDD whose second layer has one node: (x0 = 0) AND (x1 + x2 >= 1)  ==  -3 x0 + x1 + x2 >= 1
    col = BddCollection()
    col.add_linear([-3, 1, 1], ">=", 1, [0, 1, 2])
    assert col.layer_widths(0)[1] == 1
    want = {x for x in itertools.product((0, 1), repeat=3) if col.evaluate(0, x)}
    nsplit, n_all = split_long_bdds(col, 3, 1)
    assert nsplit == 1 and col.nr_bdds() == 3
    assert feasible_projection(col, 3, n_all) == want


def test_split_long_bdds_only_touches_long_ones_and_numbers_aux_variables_densely():
    col = BddCollection()
    col.add_covering(list(range(12)))          # long
    col.add_simplex([0, 5, 7])                 # short: untouched
    col.add_covering(list(range(2, 10)))       # long
    nsplit, n_all = split_long_bdds(col, 12, 4)
    assert nsplit == 2
    assert col.variables(0) == [0, 5, 7]       # survivors keep their order, chunks are appended
    assert col.nr_bdds() == 1 + 3 + 2
    used = set()
    for b in range(col.nr_bdds()):
        used |= set(col.variables(b))
    assert used == set(range(n_all))           # auxiliary ids are dense above the original variables
    for b in range(1, col.nr_bdds()):
        assert len(col.layer_widths(b)) <= 4 + 2 * 2   # chunk + head/tail auxiliary layers (cut width 2)


def test_compute_split_length_rule():
    col = BddCollection()
    for _ in range(40):
        col.add_covering(list(range(1000)))
    # 40 BDDs x 2 nodes per hop: far below any realistic parallelism -> folding until the floor of 200 layers
    assert compute_split_length(col, parallelism=1 << 20) == 199
    # parallelism 100 nodes: already >= 50 % busy un-split -> keep the full length
    assert compute_split_length(col, parallelism=100) == 1000


def test_remove_compacts_storage():
    col = BddCollection()
    col.add_simplex([0, 1, 2]); col.add_covering([1, 2, 3]); col.add_simplex([2, 3, 4, 5])
    want = [(x, col.evaluate(2, x)) for x in itertools.product((0, 1), repeat=6)]
    col.remove([0])
    assert col.nr_bdds() == 2 and col.variables(0) == [1, 2, 3] and int(col.delims[-1]) == col.nr_bdd_nodes()
    assert all(col.evaluate(1, x) == r for x, r in want)
