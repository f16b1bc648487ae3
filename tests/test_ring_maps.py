"""CPU checks of the ring GEMM's address maps (csrc/gemm_ring.h), restated in numpy: the LDS-DMA writes lane-linearly, so the
LDS image is defined by which GLOBAL address each lane requests; the fragment reads must then find element (row, k) where
the MFMA operand map expects it.  Also: the images are bank-conflict free under the guide's LDS service groups, and the
host's tile plans cover every row exactly once.  (The GPU parity tests are in test_gpu_kernels.py; this file needs no GPU.)"""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "detr-tensorflow_amd", "lib", "libdetr_hip.so")
BK = 64


def dma_k_image(A, row0, row_end, k0, img_rows8, pieces_per_wave):
    """RingDmaK: piece P = wave + 8 i covers stage rows 8P .. 8P+7; lane -> (row 8P + lane / 8, slot position lane % 8) and
    requests source chunk (lane % 8) ^ ((row >> 1) & 7).  Returns the stage image as [rows][64] (uint32 ids, 0 = not written)."""
    img = np.zeros(img_rows8 * BK, dtype=np.int64)
    for wave in range(8):
        for i in range(pieces_per_wave):
            P = wave + 8 * i
            if 8 * P >= img_rows8:
                continue                      # (dump area)
            for lane in range(64):
                r = 8 * P + (lane >> 3)
                c = (lane & 7) ^ ((r >> 1) & 7)
                g = row0 + r
                src = A[g, k0 + 8 * c:k0 + 8 * c + 8] if g < row_end else np.zeros(8, dtype=np.int64)
                dst = P * 512 + lane * 8      # element offset: piece base + lane * 16 bytes
                img[dst:dst + 8] = src
    return img


def frag_k(img, row_base, kk, lane):
    """the A / K-contiguous-B fragment of MFMA lane `lane` for k-step kk of the stage: 16 bytes at row * 128 + 16 * ((2 kk + h) ^ sw)"""
    l31, h = lane & 31, lane >> 5
    sw = (l31 >> 1) & 7
    byte = (row_base + l31) * 128 + 16 * ((2 * kk + h) ^ sw)
    return img[byte // 2: byte // 2 + 8], byte


@pytest.mark.parametrize("tm,tile_rows", [(3, 132), (2, 68), (4, 256), (1, 8), (3, 176)])
def test_k_contiguous_image_and_fragments(tm, tile_rows):
    rng = np.random.default_rng(tm * 1000 + tile_rows)
    M, K = 3 * tile_rows + 5, 256
    A = rng.integers(1, 2 ** 40, size=(M, K), dtype=np.int64)
    for m0 in (0, tile_rows, 3 * tile_rows):           # the last tile is ragged (5 rows)
        row_end = min(M, m0 + tile_rows)
        a_rows8 = (tile_rows + 7) & ~7
        for k0 in (0, 64, 192):
            img = dma_k_image(A, m0, row_end, k0, a_rows8, tm)
            for wm in range(2):
                my_rows = row_end - (m0 + wm * 32 * tm)
                nmi = 0 if my_rows <= 0 else (tm if my_rows >= 32 * tm else (my_rows + 31) >> 5)
                for mi in range(nmi):
                    rb = wm * 32 * tm + mi * 32
                    for kk in range(4):
                        for lane in range(64):
                            row = m0 + rb + (lane & 31)
                            if row >= row_end:
                                continue       # rows past the tile: whatever the image holds there is never stored
                            got, _ = frag_k(img, rb, kk, lane)
                            want = A[row, k0 + 16 * kk + 8 * (lane >> 5): k0 + 16 * kk + 8 * (lane >> 5) + 8]
                            assert np.array_equal(got, want), (m0, k0, wm, mi, kk, lane)


def test_k_contiguous_image_is_bank_conflict_free():
    """ds_read_b128 is serviced in four groups of 16 lanes (MI355X_MICROARCH.md, LDS table); inside a group the 16-byte slots
    (address / 16 mod 16 -- the 64 four-byte banks of a 256-byte row) must be distinct."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    dummy = np.zeros(1 << 16, dtype=np.int64)
    for rb in (0, 32, 96):
        for kk in range(4):
            for g in groups:
                slots = {(frag_k(dummy, rb, kk, lane)[1] // 16) % 16 for lane in g}
                assert len(slots) == 16, (rb, kk, g)


def tr_lane_off(NH, col_base, lane):
    g, t = lane >> 4, lane & 15
    kr, q = t >> 2, t & 3
    ibg = (col_base >> 4) + (g & 1)
    nh, ib = ibg >> 3, ibg & 7
    return (2 * (g >> 1) * NH + nh) * 1024 + kr * 256 + (((2 * ib + (q >> 1)) ^ (4 * kr)) * 16) + (q & 1) * 8


def test_transposed_image_and_fragments():
    """RingDmaMN + ring_frag_tr: [k][n] operand in row-major pieces (4 k rows x 128 columns, chunk pc of row kr at position
    pc ^ 4 kr).  ds_read_b64_tr_b16 semantics as pinned in round 1 (scripts/experiments/tr_b16_probe.*): the 16 lanes of a group hand
    in sixteen 8-byte chunks -- chunk t = (k row t / 4, four columns 4 (t % 4) ..) of a 4 x 16 block -- and lane c receives column c:
    its 4 consecutive k.  Also: sixteen consecutive DMA lanes request one contiguous 256-byte segment, and the 32 lanes of a
    transpose-read service group touch every one of the 64 banks exactly once."""
    rng = np.random.default_rng(7)
    for BN in (128, 256):
        NH = BN // 128
        K, N, n0 = 192, BN + 24, 0
        B = rng.integers(1, 2 ** 40, size=(K, N), dtype=np.int64)
        for k0 in (0, 128):
            img = np.zeros(BN * BK, dtype=np.int64)               # 2-byte elements
            for wave in range(8):
                for i in range(BN // 64):
                    P = wave + 8 * i
                    kb, nh = P // NH, P % NH
                    cols = []
                    for lane in range(64):
                        kr, pc = lane >> 4, lane & 15
                        col = n0 + nh * 128 + 8 * (pc ^ (4 * kr))
                        cols.append((kr, col))
                        src = B[k0 + 4 * kb + kr, col:col + 8] if col + 8 <= N else np.zeros(8, dtype=np.int64)
                        img[P * 512 + lane * 8: P * 512 + lane * 8 + 8] = src
                    for kr in range(4):                           # 16 consecutive lanes: one row, one aligned 256-byte segment
                        seg = sorted(c for r, c in cols[16 * kr:16 * kr + 16])
                        assert all(r == kr for r, _ in cols[16 * kr:16 * kr + 16]) and seg == list(range(n0 + nh * 128, n0 + nh * 128 + 128, 8))
            for col_base in range(0, BN, 32):
                for kk in range(4):
                    banks = {0: [], 1: []}
                    for lane in range(64):
                        off = tr_lane_off(NH, col_base, lane) + kk * 4 * NH * 1024
                        banks[lane >> 5] += [(off // 4) % 64, (off // 4 + 1) % 64]
                        g, t = lane >> 4, lane & 15
                        got = []
                        for half in range(2):                     # the two transpose reads: k groups 4 kk + 2 (g >> 1) + half
                            base = off + half * NH * 1024 - (t >> 2) * 256 - ((t & 3) & 1) * 8       # (undo this lane's own chunk: rebuild the block)
                            # what lane t RECEIVES: column t of the block = element t % 4 of the chunks (kr, t / 4), kr = 0 .. 3
                            blk_lane0 = (lane & ~15)
                            vals = []
                            for kr in range(4):
                                src_lane = blk_lane0 + 4 * kr + (t >> 2)          # the lane that handed in chunk (kr, quarter t / 4)
                                o = tr_lane_off(NH, col_base, src_lane) + kk * 4 * NH * 1024 + half * NH * 1024
                                vals.append(img[o // 2 + (t & 3)])
                            got += vals
                        n = col_base + (lane & 31)                # MFMA B operand map: column n = lane & 31, k = 16 kk + 8 (lane >> 5) .. + 7
                        want = B[k0 + 16 * kk + 8 * (lane >> 5): k0 + 16 * kk + 8 * (lane >> 5) + 8, n0 + n]
                        assert np.array_equal(np.array(got), want), (BN, k0, col_base, kk, lane)
                    for grp in banks.values():
                        assert sorted(grp) == list(range(64)), (BN, col_base, kk)


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_ring_plans_cover_every_row_once():
    lib = ctypes.CDLL(LIB)
    out = (ctypes.c_int32 * 8)()
    for M, N, K in [(33600, 256, 1024), (8400, 256, 2048), (133600, 128, 512), (133600, 256, 512), (8400, 2048, 512), (8400, 512, 2048),
                    (33600, 1024, 512), (8400, 256, 768), (4100, 384, 128), (257, 128, 64), (100000, 1024, 4096)]:
        assert lib.detr_hip_gemm_ring_plan(M, N, K, out) == 1, (M, N, K)
        tm, tn, ns, rows, tiles_m, tiles_n, wgs, lds = list(out)
        assert 1 <= tm <= 4 and tn in (1, 2) and ns in (2, 3) and rows <= 64 * tm and rows % 4 == 0
        assert tiles_m == -(-M // rows) and tiles_n == -(-N // (128 * tn)) and wgs == tiles_m * tiles_n
        a_rows8 = (rows + 7) & ~7
        dump = 0 if a_rows8 == 64 * tm else 1024             # (an A image at full capacity has no piece past its end: no dump area)
        assert ns * (a_rows8 + 128 * tn) * 128 + dump <= lds <= 160 * 1024
        assert lds >= 8 * 32 * (32 * tn + 4) * 4              # the epilogue's staging strips alias the ring
        # 2 x ceil(blocks / 2) >= blocks: the two row waves reach every 32-row block of the pitch
        assert 2 * 32 * tm >= rows
