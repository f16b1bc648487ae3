"""CPU checks of the LDS images of round 5's two other LDS-DMA kernels, restated in numpy from csrc/bwd_fused.hip and csrc/conv_halo_dma.h (the
ring GEMM's are in test_ring_maps.py).  An LDS-DMA piece writes lane-linearly (lane l -> 16 bytes at piece base + 16 l), so an image is DEFINED
by which global address each lane requests; the fragment reads must then find element (row, k) where the MFMA operand map expects it, and the
layouts claim to be bank-conflict free under the guide's LDS service groups (MI355X_MICROARCH.md: ds_read_b128 in four groups of 16 lanes,
ds_read_b64_tr_b16 in two groups of 32 lanes, 64 banks of 4 bytes).  No GPU needed; the GPU parity tests are in test_gpu_kernels.py."""
import numpy as np

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS = B128_GROUPS + [[l + 32 for l in g] for g in B128_GROUPS]


def tr_fragment(img, lane_addr):
    """ds_read_b64_tr_b16 as pinned in round 1 (scripts/experiments/tr_b16_probe.*) for one 16-lane group: lane t hands in the 8-byte chunk at
    lane_addr(t) -- (k row t >> 2, four columns 4 (t & 3) ..) of a 4 x 16 block -- and lane c receives column c: element c & 3 of the chunks
    handed in by lanes 4 kr + (c >> 2), kr = 0 .. 3.  img: int64 array of 2-byte elements; returns [16 lanes][4 k]."""
    out = np.zeros((16, 4), dtype=np.int64)
    for c in range(16):
        for kr in range(4):
            out[c, kr] = img[lane_addr(4 * kr + (c >> 2)) // 2 + (c & 3)]
    return out


# ---------------------------------------------------------------------------------------------------------------------------------------
# csrc/bwd_fused.hip
# ---------------------------------------------------------------------------------------------------------------------------------------
def fused_images(G, Y):
    """the dY strip image (16 pieces) and the `a` strip image (4 pieces) as the four requesting waves fill them"""
    g_img = np.zeros(32 * 256, dtype=np.int64)
    y_img = np.zeros(32 * 64, dtype=np.int64)
    for wave in range(4):
        for i in range(4):
            P = wave + 4 * i
            for lane in range(64):
                kr, ps = lane >> 4, lane & 15
                pc = ps ^ (4 * kr) ^ (P >> 2)
                row, col = 4 * (P >> 1) + kr, (P & 1) * 128 + pc * 8
                g_img[(P * 1024 + lane * 16) // 2:(P * 1024 + lane * 16) // 2 + 8] = G[row, col:col + 8]
        for lane in range(64):
            r, pos = 8 * wave + (lane >> 3), lane & 7
            c = pos ^ (4 * ((r >> 1) & 1))
            y_img[(wave * 1024 + lane * 16) // 2:(wave * 1024 + lane * 16) // 2 + 8] = Y[r, 8 * c:8 * c + 8]
    return g_img, y_img


def test_fused_backward_strip_images_serve_both_read_patterns():
    rng = np.random.default_rng(5)
    G = rng.integers(1, 2 ** 40, size=(32, 256), dtype=np.int64)
    Y = rng.integers(1, 2 ** 40, size=(32, 64), dtype=np.int64)
    g_img, y_img = fused_images(G, Y)
    # (a) input gradient: lane (row l31, half hh) reads the 16-byte chunk 2 kk + hh of its row -- ds_read_b128, conflict free per service group
    for kk in range(16):
        slots = {}
        for lane in range(64):
            l31, hh = lane & 31, lane >> 5
            c32 = 2 * kk + hh
            addr = (l31 >> 2) * 2048 + (l31 & 3) * 256 + (kk >> 3) * 1024 + (((c32 & 15) ^ (4 * (l31 & 3)) ^ (l31 >> 3)) * 16)
            assert np.array_equal(g_img[addr // 2:addr // 2 + 8], G[l31, 8 * c32:8 * c32 + 8]), (kk, lane)
            slots[lane] = (addr // 16) % 16
        for grp in B128_GROUPS:
            assert len({slots[l] for l in grp}) == 16, (kk, grp)
    # (b) weight gradient, B operand: wave w, 32-column block nb, k-step kk: lane (column l31, k = 16 kk + 8 hh ..) through two transpose reads
    for wave in range(4):
        for nb in range(2):
            for kk in range(2):
                banks = {0: [], 1: []}
                frag = {}
                for gq in range(4):
                    for half in range(2):
                        def addr(t, gq=gq, half=half):
                            kr, q = t >> 2, t & 3
                            ibg = 4 * wave + 2 * nb + (gq & 1)
                            nh, ib = ibg >> 3, ibg & 7
                            goff = (2 * (gq >> 1) * 2 + nh) * 1024 + kr * 256 + ((((2 * ib + (q >> 1)) ^ (4 * kr)) ^ (gq >> 1)) * 16) + (q & 1) * 8
                            return ((goff ^ (32 * kk)) + kk * 8192) + half * 2048
                        got = tr_fragment(g_img, addr)
                        for t in range(16):
                            frag.setdefault(16 * gq + t, []).extend(got[t])
                            if half == 0:
                                a = addr(t)
                                banks[gq >> 1] += [(a // 4) % 64, (a // 4 + 1) % 64]
                for lane in range(64):
                    col, k0 = 64 * wave + 32 * nb + (lane & 31), 16 * kk + 8 * (lane >> 5)
                    assert np.array_equal(np.array(frag[lane]), G[k0:k0 + 8, col]), (wave, nb, kk, lane)
                for grp in banks.values():
                    assert sorted(grp) == list(range(64)), (wave, nb, kk)
    # (c) weight gradient, A operand (`a` strip, [row][64 ch]): lane (channel 32 mb + l31, k = rows 16 kk + 8 hh ..)
    for mb in range(2):
        for kk in range(2):
            banks = {0: [], 1: []}
            frag = {}
            for gq in range(4):
                for half in range(2):
                    def addr(t, gq=gq, half=half):
                        kr, q = t >> 2, t & 3
                        yoff = (gq >> 1) * 1024 + kr * 128 + (((4 * mb + 2 * (gq & 1) + (q >> 1)) ^ (4 * (kr >> 1))) * 16) + (q & 1) * 8
                        return yoff + kk * 2048 + half * 512
                    got = tr_fragment(y_img, addr)
                    for t in range(16):
                        frag.setdefault(16 * gq + t, []).extend(got[t])
                        if half == 0:
                            a = addr(t)
                            banks[gq >> 1] += [(a // 4) % 64, (a // 4 + 1) % 64]
            for lane in range(64):
                ch, k0 = 32 * mb + (lane & 31), 16 * kk + 8 * (lane >> 5)
                assert np.array_equal(np.array(frag[lane]), Y[k0:k0 + 8, ch]), (mb, kk, lane)
            for grp in banks.values():
                assert sorted(grp) == list(range(64)), (mb, kk)
    # (d) the mask of the input gradient's epilogue: lane (row l31, half hh) reads columns 32 d + 8 j + 4 hh .. + 3 of `a`
    for d in range(2):
        for j in range(4):
            for lane in range(64):
                l31, hh = lane & 31, lane >> 5
                addr = (l31 >> 3) * 1024 + (l31 & 7) * 128 + hh * 8 + (((4 * d + j) ^ (4 * ((l31 >> 1) & 1))) * 16)
                c0 = 32 * d + 8 * j + 4 * hh
                assert np.array_equal(y_img[addr // 2:addr // 2 + 4], Y[l31, c0:c0 + 4]), (d, j, lane)


def test_fused_backward_weight_image():
    """W [64][256] resident in LDS: row n = 512 bytes, chunk c at position c ^ (n & 15); the input gradient's A fragments (lane n, chunk 2 kk + hh)"""
    rng = np.random.default_rng(6)
    W = rng.integers(1, 2 ** 40, size=(64, 256), dtype=np.int64)
    img = np.zeros(64 * 256, dtype=np.int64)
    for wave in range(4):
        for i in range(8):
            P = wave + 4 * i
            for lane in range(64):
                n, pos = 2 * P + (lane >> 5), lane & 31
                c = pos ^ (n & 15)
                img[(P * 1024 + lane * 16) // 2:(P * 1024 + lane * 16) // 2 + 8] = W[n, 8 * c:8 * c + 8]
    for d in range(2):
        for kk in range(16):
            slots = {}
            for lane in range(64):
                n, c32 = 32 * d + (lane & 31), 2 * kk + (lane >> 5)
                addr = n * 512 + ((c32 ^ (n & 15)) * 16)
                assert np.array_equal(img[addr // 2:addr // 2 + 8], W[n, 8 * c32:8 * c32 + 8])
                slots[lane] = (addr // 16) % 16
            for grp in B128_GROUPS:
                assert len({slots[l] for l in grp}) == 16


# ---------------------------------------------------------------------------------------------------------------------------------------
# csrc/conv_halo_dma.h
# ---------------------------------------------------------------------------------------------------------------------------------------
def test_halo_dma_patch_image_and_tap_fragments():
    """patch of a 32-channel chunk: [204 pixels][32 ch] in 64-byte rows, chunk c of pixel p at position c ^ ((p >> 2) & 3); 13 pieces of 16 pixels.
    Every tap's A fragments (tile row, pixel l31 + tap shift) must find their channels and be conflict free for all nine shifts."""
    rng = np.random.default_rng(7)
    PW, PIX = 34, 6 * 34
    X = rng.integers(1, 2 ** 40, size=(PIX, 32), dtype=np.int64)
    img = np.zeros(13 * 512, dtype=np.int64)
    for wave in range(4):
        for i in range(4):
            P = wave + 4 * i
            if P >= 13:
                continue                                   # (dump area)
            for lane in range(64):
                px = 16 * P + (lane >> 2)
                c = (lane & 3) ^ ((px >> 2) & 3)
                src = X[px, 8 * c:8 * c + 8] if px < PIX else np.zeros(8, dtype=np.int64)
                img[(P * 1024 + lane * 16) // 2:(P * 1024 + lane * 16) // 2 + 8] = src
    for wm in range(2):
        for mi in range(2):
            for tp in range(9):
                toff = (tp // 3) * PW + tp % 3
                for kk in range(2):
                    slots = {}
                    for lane in range(64):
                        l31, hh = lane & 31, lane >> 5
                        p = (2 * wm + mi) * PW + l31 + toff
                        addr = (p * 64 + 16 * (hh ^ ((p >> 2) & 3))) ^ (32 * kk)
                        ch = 16 * kk + 8 * hh
                        assert np.array_equal(img[addr // 2:addr // 2 + 8], X[p, ch:ch + 8]), (wm, mi, tp, kk, lane)
                        slots[lane] = (addr // 16) % 16
                    for grp in B128_GROUPS:
                        assert len({slots[l] for l in grp}) == 16, (wm, mi, tp, kk)


def test_halo_dma_kernel_tile_images():
    """kernel tile of a step, 128 output columns x 32 k: forward [k][n] as the row-major transpose-read image (pieces of 4 k x 128 n), input
    gradient [n][k] as 64-byte rows with chunk c of row r at position c ^ ((r >> 2) & 3)."""
    rng = np.random.default_rng(8)
    # forward
    Wf = rng.integers(1, 2 ** 40, size=(32, 128), dtype=np.int64)          # [k][n]
    img = np.zeros(8 * 512, dtype=np.int64)
    for wave in range(4):
        for i in range(2):
            P = wave + 4 * i
            for lane in range(64):
                kr, pc = lane >> 4, lane & 15
                col = 8 * (pc ^ (4 * kr))
                img[(P * 1024 + lane * 16) // 2:(P * 1024 + lane * 16) // 2 + 8] = Wf[4 * P + kr, col:col + 8]
    for col_base in range(0, 128, 32):
        for kk in range(2):
            frag, banks = {}, {0: [], 1: []}
            for gq in range(4):
                for half in range(2):
                    def addr(t, gq=gq, half=half):       # ring_tr_lane_off<1> + ring_frag_tr<1>
                        kr, q = t >> 2, t & 3
                        ib = (col_base >> 4) + (gq & 1)
                        return (2 * (gq >> 1)) * 1024 + kr * 256 + (((2 * ib + (q >> 1)) ^ (4 * kr)) * 16) + (q & 1) * 8 + kk * 4096 + half * 1024
                    got = tr_fragment(img, addr)
                    for t in range(16):
                        frag.setdefault(16 * gq + t, []).extend(got[t])
                        if half == 0:
                            a = addr(t)
                            banks[gq >> 1] += [(a // 4) % 64, (a // 4 + 1) % 64]
            for lane in range(64):
                n, k0 = col_base + (lane & 31), 16 * kk + 8 * (lane >> 5)
                assert np.array_equal(np.array(frag[lane]), Wf[k0:k0 + 8, n]), (col_base, kk, lane)
            for grp in banks.values():
                assert sorted(grp) == list(range(64))
    # input gradient
    Wd = rng.integers(1, 2 ** 40, size=(128, 32), dtype=np.int64)          # [n][k]
    img = np.zeros(8 * 512, dtype=np.int64)
    for wave in range(4):
        for i in range(2):
            P = wave + 4 * i
            for lane in range(64):
                r = 16 * P + (lane >> 2)
                c = (lane & 3) ^ ((r >> 2) & 3)
                img[(P * 1024 + lane * 16) // 2:(P * 1024 + lane * 16) // 2 + 8] = Wd[r, 8 * c:8 * c + 8]
    for wn in range(2):
        for ni in range(2):
            for kk in range(2):
                slots = {}
                for lane in range(64):
                    l31, hh = lane & 31, lane >> 5
                    r = wn * 64 + ni * 32 + l31
                    addr = (r * 64 + 16 * (hh ^ ((r >> 2) & 3))) ^ (32 * kk)
                    assert np.array_equal(img[addr // 2:addr // 2 + 8], Wd[r, 16 * kk + 8 * hh:16 * kk + 8 * hh + 8])
                    slots[lane] = (addr // 16) % 16
                for grp in B128_GROUPS:
                    assert len({slots[l] for l in grp}) == 16


# ---------------------------------------------------------------------------------------------------------------------------------------
# csrc/conv_halo.h: grid order of the stride-2 input gradient's parity classes
# ---------------------------------------------------------------------------------------------------------------------------------------
def s2_class_of_workgroup(wid, nk):
    """restatement of the cls_mix branch of conv3x3_halo_bf16_kernel<.., S2C = true>: workgroup id -> (class, tile index inside the class)"""
    m = min(nk)
    if wid < 4 * m:
        return wid & 3, wid >> 2
    j, k = wid - 4 * m, 0
    for q in range(3):
        if k == q and j >= nk[q] - m:
            j -= nk[q] - m
            k = q + 1
    return k, m + j


def test_stride2_class_round_robin_covers_every_tile_once():
    """Every (class, tile) pair is produced by exactly one workgroup id, for equal, unequal and EMPTY classes (a one-row map has no (1, .) class;
    sizes as launch_conv_halo_s2classes computes them: N * ceil(Hc / 4) * ceil(Wc / 32) * channel panels)."""
    import itertools
    cases = [[5, 5, 5, 5], [6, 6, 5, 5], [7, 5, 6, 4], [0, 3, 0, 3], [0, 0, 0, 2], [4, 0, 4, 0], [1, 2, 3, 4], [9, 1, 1, 1], [0, 0, 0, 0]]
    for N, H, W, panels in itertools.product((1, 3), (1, 2, 7, 50), (1, 5, 84, 167), (1, 4)):
        nk = []
        for k in range(4):
            ph, pw = (1 if k in (0, 2) else 0), (1 if k in (0, 1) else 0)
            Hc, Wc = (H - ph + 1) // 2, (W - pw + 1) // 2
            nk.append(N * -(-Hc // 4) * -(-Wc // 32) * panels)
        cases.append(nk)
    for nk in cases:
        seen = set()
        for wid in range(sum(nk)):
            k, idx = s2_class_of_workgroup(wid, nk)
            assert 0 <= k < 4 and 0 <= idx < nk[k], (nk, wid, k, idx)
            assert (k, idx) not in seen, (nk, wid)
            seen.add((k, idx))
        assert len(seen) == sum(nk)


# ---------------------------------------------------------------------------------------------------------------------------------------
# csrc/conv_f32.hip: conv3x3_wgrad_fused_bf16_kernel -- the haloed input patch image (round 6: rows of a sub-block rotated by the ci block)
# ---------------------------------------------------------------------------------------------------------------------------------------
WRITE_B128_GROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]       # ds_write_b128: 8 contiguous lanes per LDS cycle, 32 banks
WRITE_B64_GROUPS = [list(range(16 * g, 16 * g + 16)) for g in range(4)]     # ds_write_b64: 16 contiguous lanes, 32 banks


def wgrad_patch_offset(c, cb, q, rotate=True):
    """shorts offset inside one patch row of (pixel slot c, 16-channel block cb, channel quad q)"""
    row = ((c & 3) + cb) & 3 if rotate else c & 3
    return ((c >> 2) * 4 + cb) * 64 + row * 16 + q * 4


def _write_conflicts(rotate, s16):
    """extra LDS cycles of one patch store instruction of wave 0 (register i = 0): lanes of a service group on the same bank, distinct addresses"""
    groups = WRITE_B128_GROUPS if s16 else WRITE_B64_GROUPS
    worst = 0
    for grp in groups:
        use = {}
        for lane in grp:
            v = lane
            pp, c4 = (v >> 3, 2 * (v & 7)) if s16 else (v >> 4, v & 15)
            o = wgrad_patch_offset(pp, c4 >> 2, c4 & 3, rotate)               # (stride 1, patch row 0: slot = pp)
            for d in range(4 if s16 else 2):                                   # dwords of the store
                use.setdefault((o // 2 + d) % 32, set()).add(o // 2 + d)
        worst = max(worst, max(len(a) for a in use.values()))
    return worst


def test_wgrad_fused_patch_stores_are_conflict_free_since_round_6():
    assert _write_conflicts(rotate=False, s16=True) == 4 and _write_conflicts(rotate=False, s16=False) == 4      # what rounds 3-5 ran
    assert _write_conflicts(rotate=True, s16=True) == 1 and _write_conflicts(rotate=True, s16=False) == 1


def test_wgrad_fused_patch_fragments_read_the_right_pixels_without_conflicts():
    """the A fragment of tap kw, pixel half s2: lane (g, t16) hands in (pixel 16 s2 + 8 (g >> 1) + (t16 >> 2) + kw + 4 h, ci block 2 wm + (g & 1),
    quad t16 & 3); after the transpose read lane l of the MFMA holds ci = 32 wm + (l & 31), k = 8 (l >> 5) + 4 h .. + 4 = consecutive output pixels"""
    rng = np.random.default_rng(11)
    PC = 34
    X = rng.integers(1, 2 ** 40, size=(PC, 64), dtype=np.int64)               # [patch pixel][ci]
    img = np.zeros(((PC + 3) // 4) * 4 * 64, dtype=np.int64)
    for c in range(PC):
        for cb in range(4):
            for q in range(4):
                o = wgrad_patch_offset(c, cb, q)
                img[o:o + 4] = X[c, 16 * cb + 4 * q:16 * cb + 4 * q + 4]
    for wm in range(2):
        for s2 in range(2):
            for kw in range(3):
                for h in range(2):
                    banks = {0: [], 1: []}
                    for g in range(4):
                        cib = 2 * wm + (g & 1)

                        def addr(t, g=g, cib=cib):
                            c = 16 * s2 + 8 * (g >> 1) + (t >> 2) + kw + 4 * h
                            return 2 * wgrad_patch_offset(c, cib, t & 3)
                        got = tr_fragment(img, addr)
                        for t in range(16):
                            ci = 16 * cib + t
                            pix0 = 16 * s2 + 8 * (g >> 1) + kw + 4 * h
                            assert np.array_equal(got[t], X[pix0:pix0 + 4, ci]), (wm, s2, kw, h, g, t)
                            a = addr(t)
                            banks[g >> 1] += [(a // 4) % 64, (a // 4 + 1) % 64]
                    for half in (0, 1):                                        # ds_read_b64_tr_b16: two service groups of 32 lanes, 64 banks
                        assert len(set(banks[half])) == 64, (wm, s2, kw, h, half)
