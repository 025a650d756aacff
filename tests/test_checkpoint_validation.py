"""Checkpoint validation without a GPU (capi.cpp: bddmma_load).  A file is checked — magic, record sizes, checksum of the layout
section, every index / offset / size array against its target — BEFORE a device is touched, so on a box without a GPU a good file
gets as far as "no HIP device" and a bad one is refused as corrupt.  tests/golden/checkpoint_small_v07.bin was written by
tools/make_checkpoint_fixture.py on an MI355X (120 variables, 93 BDDs, narrow + wide packs, 39 KB)."""
import os

import numpy as np
import pytest

from bdd_amd import capi
from bdd_amd.solver import bdd_hip_parallel_mma
from util import CHECKPOINT_ARRAY_IDS as IDS, GOLDEN_DIR, Checksum, parse_checkpoint, write_checkpoint

FIXTURE = os.path.join(GOLDEN_DIR, "checkpoint_small_v07.bin")


def load_error(path):
    """'' when the file passed validation (the load then fails for want of a GPU, or succeeds on a GPU box), else the message"""
    try:
        bdd_hip_parallel_mma.load(path)
        return ""
    except capi.BddMmaError as e:
        msg = str(e)
        return "" if "no HIP device" in msg else msg


def test_fixture_is_accepted_and_the_checksum_is_restated_correctly(tmp_path):
    raw = open(FIXTURE, "rb").read()
    assert load_error(FIXTURE) == ""
    head, sc, opts, recs, stored, tail = parse_checkpoint(raw)
    cs = Checksum(); cs.add(sc); cs.add(opts)
    for i, es, cnt, data in recs:
        cs.add(i.to_bytes(8, "little") + es.to_bytes(8, "little") + cnt.to_bytes(8, "little")); cs.add(data)
    assert cs.value() == stored
    p = str(tmp_path / "same.bin")
    write_checkpoint(p, head, sc, opts, recs, tail)
    assert open(p, "rb").read() == raw and load_error(p) == ""


def test_truncated_flipped_and_foreign_files_are_refused(tmp_path):
    raw = open(FIXTURE, "rb").read()
    p = str(tmp_path / "bad.bin")
    for cut in (0, 7, 20, 200, len(raw) // 3, len(raw) - 16):
        open(p, "wb").write(raw[:cut])
        assert load_error(p) != ""
    head, sc, opts, recs, stored, tail = parse_checkpoint(raw)
    layout_end = len(raw) - len(tail)
    for pos in (48, 170, layout_end // 3, layout_end // 2, layout_end - 20):   # one flipped bit anywhere in the layout section
        b = bytearray(raw); b[pos] ^= 0x10
        open(p, "wb").write(bytes(b))
        assert "corrupt" in load_error(p), pos
    b = bytearray(raw); b[7] = ord("4")                                       # the previous format's magic
    open(p, "wb").write(bytes(b))
    assert "another version" in load_error(p)


def poke(recs, name, index, value, dtype):
    out = []
    for i, es, cnt, data in recs:
        if i == IDS[name]:
            a = np.frombuffer(data, dtype=dtype).copy()
            a[index] = value
            data = a.tobytes()
        out.append([i, es, cnt, data])
    return out


CASES = [("evar", 7, 10**6, np.uint32), ("bvar", 5, 60000, np.uint16), ("lpos", 0, 2**31, np.uint32), ("vpos", 3, 2**31, np.uint32),
         ("var_layers", 9, 2**30, np.uint32), ("layer_var", 2, -1, np.int32), ("layer_var", 2, 10**6, np.int32), ("bdd_root_slot", 1, 2**32 - 1, np.uint32),
         ("cs_entry", 4, 2**30, np.uint32), ("cs_slot", 4, 65000, np.uint16), ("pack_hdr", 0, 5, np.uint32), ("pack_hdr", 1, 10**6, np.uint32),
         ("pack_hdr", 6, 2**31, np.uint32), ("quad_hdr", 0, 77, np.uint32), ("quad_hdr", 1, 10**6, np.uint32), ("grp_hop_end", 0, 2**30, np.uint32),
         ("grp_layer_off", 1, 2**30, np.uint32), ("num_bdds_per_var", 0, -5, np.int32), ("bin_ptr", 1, 2**31, np.uint32),
         ("narrow_words", 0, 0x1FF | (0x1FF << 9), np.uint32)]


@pytest.mark.parametrize("name,index,value,dtype", CASES, ids=[f"{c[0]}[{c[1]}]={c[2]}" for c in CASES])
def test_an_out_of_range_index_with_a_correct_checksum_is_refused(tmp_path, name, index, value, dtype):
    head, sc, opts, recs, _, tail = parse_checkpoint(open(FIXTURE, "rb").read())
    p = str(tmp_path / "bad.bin")
    write_checkpoint(p, head, sc, opts, poke(recs, name, index, value, dtype), tail)
    assert "corrupt" in load_error(p)


def test_bad_scalars_and_short_per_hop_tables_are_refused(tmp_path):
    head, sc, opts, recs, _, tail = parse_checkpoint(open(FIXTURE, "rb").read())
    p = str(tmp_path / "bad.bin")
    # LayoutScalars: 7 x u64, then pack_width, wide_pack_width, huge_pack_width, narrow_slots, vars_per_bin, n_bins, stage_cap, waves_per_block, ...
    for off, val in ((56, 100), (56 + 28, 3), (56 + 24, 0), (56 + 16, 0), (56 + 20, 10**6), (48, 2**31), (8, 10**9)):
        b = bytearray(sc); b[off:off + 4] = int(val).to_bytes(4, "little")
        write_checkpoint(p, head, bytes(b), opts, recs, tail)
        assert "corrupt" in load_error(p), (off, val)
    for name in ("nodes_per_hop", "layers_per_hop"):   # bddmma_layers_per_hop copies size() entries into a caller buffer of n_hops
        short = [[i, es, cnt - 1, data[:-es]] if i == IDS[name] else [i, es, cnt, data] for i, es, cnt, data in recs]
        write_checkpoint(p, head, sc, opts, short, tail)
        assert "corrupt" in load_error(p)
