"""Cube4 — the remaining environment of the reference's C++ core (cpp/environments.cpp:263-370) — on the CPU box: the
oracle's restatement (explicit 4-cycles) and the product's compile-time table (face rotation + strip description,
csrc/dca_common.h) against EACH OTHER and against the REFERENCE's own compiled class (oracle/_ref, `make -C oracle ref`):
every move on random sticker arrangements, getNextStates, isSolved (any same-colour arrangement counts), move / inverse."""
import hashlib

import numpy as np
import pytest

from oracle import c_oracle as co

PERM_SHA256 = "7eb3d1a82e1beb130fccfc3fc40e22255869adb6feea87e3d99d36cc38445f59"  # the 24 x 96 gather table, recorded from oracle/_ref


def _states(n, seed):
    rng = np.random.default_rng(seed)
    return rng.permuted(np.tile(np.arange(96, dtype=np.uint8), (n, 1)), axis=1)


def _oracle_perm():
    ident = np.arange(96, dtype=np.uint8)[None]
    return np.stack([co.next_state("cube4", ident, a)[0] for a in range(24)])


def test_cube4_tables_oracle_product_reference_agree():
    from deepcubea_amd import _lib
    perm = _oracle_perm()
    assert hashlib.sha256(perm.tobytes()).hexdigest() == PERM_SHA256
    assert np.array_equal(_lib.cube4_perm_table(), perm)  # the product's generator (host-readable copy of the device table)
    for a in range(24):
        assert sorted(perm[a].tolist()) == list(range(96))
        assert np.array_equal(perm[a][perm[a ^ 1]], np.arange(96))  # move a ^ 1 undoes move a
        moved = int((perm[a] != np.arange(96)).sum())
        assert moved == (32 if a < 12 else 16)  # outer turn: 12 face + 16 side stickers move... (4 centres turn too); inner slice: 16
    assert _lib.env_ids("cube4") == (3, 0, 96, 24, 6)


@pytest.mark.skipif(co.ref_lib() is None, reason="oracle/_ref not built (needs /root/reference)")
def test_cube4_oracle_vs_reference_class():
    s = _states(300, 1)
    for a in range(24):
        assert np.array_equal(co.next_state("cube4", s, a), co.ref_next_state("cube4", s, a)), a
    ch, sv, _ = co.expand("cube4", s[:64])
    rch, rsv = co.ref_expand("cube4", s[:64])
    assert ch.shape == (64, 24, 96) and np.array_equal(ch, rch) and np.array_equal(sv, rsv)
    # isSolved: goal, goal under whole-face-preserving relabelling (same colours, other stickers), near-solved, scrambled
    goal = np.arange(96, dtype=np.uint8)
    same_colours = goal.copy()
    rng = np.random.default_rng(2)
    for f in range(6):
        same_colours[f * 16:(f + 1) * 16] = f * 16 + rng.permutation(16)
    swapped_faces = np.concatenate([goal[16:32], goal[0:16], goal[32:]])  # faces exchanged: still one colour per face
    one_off = goal.copy()
    one_off[[5, 21]] = one_off[[21, 5]]
    cases = np.stack([goal, same_colours, swapped_faces, one_off, s[0], co.next_state("cube4", goal[None], 7)[0]])
    want = np.array([True, True, True, False, False, False])
    assert np.array_equal(co.is_solved("cube4", cases), want)
    r = co.ref_lib()
    import ctypes as C
    got = np.empty(len(cases), np.uint8)
    r.ref_is_solved(3, 0, cases.ctypes.data_as(C.c_void_p), C.c_int64(len(cases)), got.ctypes.data_as(C.c_void_p))
    assert np.array_equal(got.astype(bool), want)
    # a scramble and its inverse sequence
    t = goal[None].copy()
    seq = [0, 13, 6, 22, 9, 15, 3]
    for a in seq:
        t = co.next_state("cube4", t, a)
    assert not co.is_solved("cube4", t)[0]
    for a in reversed(seq):
        t = co.next_state("cube4", t, a ^ 1)
    assert np.array_equal(t[0], goal)
