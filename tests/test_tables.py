"""The product's constant tables (csrc/icer_tables.hpp), entry by entry against the reference build."""
import ctypes as C

import numpy as np


class CoderTables(C.Structure):     # mirrors icer::CoderTables
    _fields_ = [("v2v", (C.c_uint16 * 32) * 8), ("v2v_term", (C.c_uint32 * 8) * 8), ("v2v_step6", ((C.c_uint16 * 64) * 8) * 8), ("v2v_tail", ((C.c_uint8 * 64) * 8) * 8),
                ("node_c", (C.c_uint8 * 32) * 8), ("node_full", (C.c_uint8 * 8) * 8),
                ("cand_bin", C.c_uint8 * 64), ("cand_node", C.c_uint8 * 64), ("cand_lane", (C.c_uint8 * 32) * 8), ("v2v_flush", ((C.c_uint8 * 6) * 9) * 8),
                ("gm", C.c_uint16 * 17), ("gl", C.c_uint16 * 17), ("gi", C.c_uint16 * 17),
                ("ginv", C.c_uint32 * 17), ("cut", C.c_uint32 * 16), ("binlut", C.c_uint32 * 257),
                ("x2n", C.c_uint32 * 32)]


def _tables(emu):
    emu.lib.emu_get_tables.restype = C.c_size_t
    emu.lib.emu_get_tables.argtypes = [C.c_void_p, C.c_size_t]
    t = CoderTables()
    assert emu.lib.emu_get_tables(C.byref(t), C.sizeof(t)) == C.sizeof(t)
    return t


def test_tables_equal_reference(emu, reference):
    t = _tables(emu)
    for b in range(1, 8):
        for pre in range(32):
            nin, nout, code = reference.custom_code(b, pre)
            e = t.v2v[b][pre]
            assert (e & 15, (e >> 4) & 15, e >> 8) == ((nin, nout, code) if nin else (0, 0, 0)), (b, pre)
            for n in range(1, 6):       # termination masks agree with the look-up rule of icer_encoding.c:91
                assert ((t.v2v_term[b][n] >> pre) & 1) == (1 if nin == n else 0)
        for pre in range(9):
            for nb in range(6):
                fb, fn = reference.flush_entry(b, pre, nb)
                assert (t.v2v_flush[b][pre][nb] & 15, t.v2v_flush[b][pre][nb] >> 4) == (fb, fn), (b, pre, nb)
    for b in range(8, 17):
        assert (t.gm[b], t.gl[b], t.gi[b]) == reference.golomb(b)
        assert all(((z * t.ginv[b]) >> 20) == z // t.gm[b] for z in range(2048))     # exact reciprocal
    for i in range(16):
        assert t.cut[i] == reference.lib.ref_tap_cutoff(i)


def _tree_nodes(reference, b):
    partial = {1}                                       # root + every proper prefix of a code word's input
    for pre in range(32):
        nin, _, _ = reference.custom_code(b, pre)
        for k in range(1, nin):
            partial.add((pre & ((1 << k) - 1)) | (1 << k))
    return partial


def _walk(reference, b, node, bits, nbits):
    """nbits applications of the reference's per-bit rule (icer_encoding.c:87-98: a prefix is complete when the table
    entry's input length equals the bits consumed); returns (node after, start flags)"""
    nin = node.bit_length() - 1
    acc, starts = node ^ (1 << nin), 0
    for k in range(nbits):
        if nin == 0:
            starts |= 1 << k
        acc |= ((bits >> k) & 1) << nin
        nin += 1
        if reference.custom_code(b, acc)[0] == nin:
            acc, nin = 0, 0
        assert nin < 5
    return acc | (1 << nin), starts


def test_six_bit_step_tables_equal_single_steps(emu, reference):
    """v2v_step6 / v2v_tail (compact node numbers) against repeated single-bit steps of the reference's rule."""
    t = _tables(emu)
    for b in range(1, 8):
        nodes = _tree_nodes(reference, b)
        assert len(nodes) <= 8
        assert sorted(n for n in range(32) if t.node_c[b][n] != 0xFF) == sorted(nodes) and t.node_c[b][1] == 0
        for node in nodes:
            cn = t.node_c[b][node]
            assert t.node_full[b][cn] == node
            for bits in range(64):
                after, starts = _walk(reference, b, node, bits, 6)
                e = t.v2v_step6[b][cn][bits]
                assert (t.node_full[b][e & 7], e >> 4) == (after, starts), (b, node, bits)
            for k in range(1, 6):
                for bits in range(1 << k):
                    after, starts = _walk(reference, b, node, bits, k)
                    assert t.node_full[b][t.v2v_tail[b][cn][(1 << k) | bits]] == after
                    assert (t.v2v_step6[b][cn][bits] >> 4) & ((1 << k) - 1) == starts


def test_walker_candidate_lanes_cover_every_tree_node(emu, reference):
    """Every node of a bin's code tree has exactly one candidate lane (the split walk relies on it)."""
    t = _tables(emu)
    for b in range(1, 8):
        nodes = _tree_nodes(reference, b)
        lanes = {node: t.cand_lane[b][node] for node in nodes}
        assert all(8 <= ln < 64 for ln in lanes.values()) and len(set(lanes.values())) == len(nodes), (b, lanes)
        assert all(t.cand_bin[ln] == b and t.cand_node[ln] == node for node, ln in lanes.items())
    assert sum(1 for ln in range(64) if t.cand_bin[ln]) == 46


def test_pick_bin_equals_reference(emu, reference):
    t = _tables(emu)
    assert all(e != 0xFFFFFFFF for e in t.binlut)          # at most one cut-off per bucket of 256
    for total in range(1, 640):                            # (counts are rescaled at 500)
        for zero in range(total // 2, total + 1):
            assert emu.lib.emu_pick_bin(zero, total) == reference.lib.icer_compute_bin(zero, total), (zero, total)


def test_crc_shift_table(emu):
    """x2n[k] = x^(2^k) mod P: advancing a CRC state over 2^(k-3) zero bytes must equal multiplying by it."""
    import zlib
    t = _tables(emu)

    def mulmod(a, b):
        p = 0
        m = 1 << 31
        while m:
            if a & m:
                p ^= b
                if a & (m - 1) == 0:
                    break
            m >>= 1
            b = (b >> 1) ^ 0xEDB88320 if b & 1 else b >> 1
        return p
    a, bb = b"hello ICER", bytes(range(200)) * 3
    op = 1 << 31
    n, k = len(bb), 3
    while n:
        if n & 1:
            op = mulmod(t.x2n[k & 31], op)
        n >>= 1
        k += 1
    assert mulmod(op, zlib.crc32(a)) ^ zlib.crc32(bb) == zlib.crc32(a + bb)
