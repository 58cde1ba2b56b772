#!/usr/bin/env python3
"""CPU emulation of wgrad_bf16dma_kernel's addressing (csrc/wgrad_bf16.hip, OSVOS_WGRAD_FORM=4 / 5): the LDS-DMA placement of the XOR-swizzled
pixel-major tiles, the ds_read_b64_tr_b16 gathers (semantics as probed in profiles/r01_tr_b16_probe.txt), the tap / half-patch geometry and
the MFMA operand roles -- run BEFORE the kernel's first GPU launch to check that every fragment holds the elements the weight gradient needs,
that the bias sums come out, and that no 32-lane half of a gather hits a bank twice.  It restates the kernel's index formulas in numpy; it
does not execute the HIP code (tests/test_gpu_ops.py::test_wgrad_bf16_forms_are_bit_identical does that on the GPU).
usage: python tools/emu_wgrad_dma.py"""
import numpy as np
PW, PH, HP, BCI, BCOT, WAVES = 32, 8, 4, 64, 128, 8
DYP, XP, XW = 256, 128, 36
XPIX = (HP + 2) * XW
X_B, DY_B = XPIX * XP, HP * PW * DYP
X_OFF, DY_OFF = 0, 2 * X_B
NDY, XI = DY_B // 1024 // WAVES, X_B // 1024
NX = (XI + WAVES - 1) // WAVES
LDS = 2 * X_B + 2 * DY_B
OOB = None

def run(H, W, Cin_s, Cout, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.integers(-3, 4, size=(H, W, Cin_s)).astype(np.float64)
    dY = rng.integers(-3, 4, size=(H, W, Cout)).astype(np.float64)
    Xb, dYb = X.reshape(-1), dY.reshape(-1)          # element index = byte offset / 2
    npx, npy = -(-W // PW), -(-H // PH)
    co0, ci0 = 0, 0
    lds = np.zeros(LDS // 2)                          # 16-bit elements
    acc = np.zeros((WAVES, 9, 32, 32))                # [wave][tap][m = ci local][n = co local]
    bsum = np.zeros((WAVES, 64))

    def dma_half(px, py, h, buf):
        x0, y0 = px * PW, py * PH
        for wv in range(WAVES):
            for d in range(NDY + NX):
                for lane in range(64):
                    if d < NDY:
                        dcol = 4 * wv + (lane >> 4)
                        doct = (lane & 15) ^ (4 * (dcol & 3))
                        y = y0 + HP * h + d
                        ok = co0 + 8 * doct < Cout and x0 + dcol < W and y < H
                        off = ((dcol * Cout + co0 + 8 * doct) * 2 + (y * W + x0) * Cout * 2) if ok else OOB
                        dst = DY_OFF + buf * DY_B + (wv + WAVES * d) * 1024 + lane * 16
                        src = dYb
                    else:
                        j = d - NDY
                        if wv + WAVES * j >= XI:
                            continue
                        e = 64 * (wv + WAVES * j) + lane
                        hp = e >> 3; hy = hp // XW; hx = hp - hy * XW
                        octv = (e & 7) ^ (4 * ((hx >> 1) & 1))
                        xrel = ((hy * W + hx) * Cin_s + ci0 + 8 * octv) * 2
                        xcol = hx - 1 if (hx < PW + 2 and ci0 + 8 * octv < Cin_s) else 0x40000000
                        xrow = hy - 1
                        yb = y0 + HP * h
                        ok = 0 <= x0 + xcol < W and 0 <= yb + xrow < H
                        off = (((yb - 1) * W + x0 - 1) * Cin_s * 2 + xrel) if ok else OOB
                        dst = X_OFF + buf * X_B + (wv + WAVES * j) * 1024 + lane * 16
                        src = Xb
                    if off is None:
                        lds[dst // 2: dst // 2 + 8] = 0
                    else:
                        assert off >= 0 and off % 16 == 0 and off // 2 + 8 <= src.size, (off, d)
                        lds[dst // 2: dst // 2 + 8] = src[off // 2: off // 2 + 8]

    def tr_read(addrs):
        """addrs[64] byte addresses -> out[64][4] per the probed semantics (16-lane groups)."""
        out = np.zeros((64, 4))
        for lane in range(64):
            g, i = lane // 16, lane % 16
            for j in range(4):
                src_lane = g * 16 + 4 * j + i // 4
                a = addrs[src_lane]
                assert a % 8 == 0
                out[lane, j] = lds[a // 2 + (i % 4)]
        return out

    def tr8(addrs, pitch):
        return np.concatenate([tr_read(addrs), tr_read([a + 4 * pitch for a in addrs])], axis=1)

    def half_patch(B):
        for wave in range(WAVES):
            wc, wi = wave >> 1, wave & 1
            a_base, b_base = [], [[], [], []]
            for lane in range(64):
                fi, fg, lh = lane & 15, (lane >> 4) & 1, lane >> 5
                fp, fo, fb = fi >> 2, 2 * fg + ((fi & 3) >> 1), (fi & 1) * 8
                a_base.append(DY_OFF + (8 * lh + fp) * DYP + ((4 * wc + fo) ^ (4 * (fp & 3))) * 16 + fb)
                for s2 in range(3):
                    b_base[s2].append(X_OFF + (8 * lh + fp) * XP + ((4 * wi + fo) ^ (4 * (((fp + s2) >> 1) & 1))) * 16 + fb)
            for ks in range(HP * 2):
                af = tr8([a + B * DY_B + ((ks >> 1) * PW + (ks & 1) * 16) * DYP for a in a_base], DYP)      # [lane][8]
                dYf = np.zeros((32, 16))
                for lane in range(64):
                    dYf[lane & 31, 8 * (lane >> 5): 8 * (lane >> 5) + 8] = af[lane]
                if wi == 0:
                    bsum[wave] += af.sum(axis=1)
                for r in range(3):
                    for s2 in range(3):
                        bf = tr8([b + B * X_B + (((ks >> 1) + r) * XW + (ks & 1) * 16 + s2) * XP for b in b_base[s2]], XP)
                        Xf = np.zeros((32, 16))
                        for lane in range(64):
                            Xf[lane & 31, 8 * (lane >> 5): 8 * (lane >> 5) + 8] = bf[lane]
                        acc[wave, r * 3 + s2] += Xf @ dYf.T

    for py in range(npy):
        for px in range(npx):
            dma_half(px, py, 0, 0)
            dma_half(px, py, 1, 1)
            half_patch(0)
            half_patch(1)
    # expected
    Xp = np.zeros((H + 2, W + 2, Cin_s)); Xp[1:-1, 1:-1] = X
    ok = True
    for wave in range(WAVES):
        wc, wi = wave >> 1, wave & 1
        for r in range(3):
            for s in range(3):
                exp = np.einsum('hwi,hwo->io', Xp[r:r + H, s:s + W, 32 * wi:32 * wi + 32], dY[:, :, 32 * wc:32 * wc + 32])
                if not np.array_equal(exp, acc[wave, r * 3 + s]):
                    ok = False
                    print("MISMATCH wave", wave, "tap", r, s, np.abs(exp - acc[wave, r * 3 + s]).max())
    # bias: lane (fi, fg, lh) of waves wi=0 holds cout 32 wc + 16 fg + fi
    for wave in range(0, WAVES, 2):
        wc = wave >> 1
        for lane in range(32):
            tot = bsum[wave, lane] + bsum[wave, lane + 32]
            c = 32 * wc + lane
            if tot != dY[:, :, c].sum():
                ok = False; print("bias mismatch", c)
    print("H=%d W=%d Cin=%d Cout=%d: %s" % (H, W, Cin_s, Cout, "OK" if ok else "FAILED"))

run(9, 40, 64, 128)
run(16, 64, 64, 128, 1)
run(5, 33, 64, 128, 2)

def conflicts():
    worst = 0
    for wave in range(WAVES):
        wc, wi = wave >> 1, wave & 1
        for B in range(2):
            for ks in range(8):
                for kind in range(10):
                    addrs = []
                    for lane in range(64):
                        fi, fg, lh = lane & 15, (lane >> 4) & 1, lane >> 5
                        fp, fo, fb = fi >> 2, 2 * fg + ((fi & 3) >> 1), (fi & 1) * 8
                        if kind == 9:
                            a = DY_OFF + (8 * lh + fp) * DYP + ((4 * wc + fo) ^ (4 * (fp & 3))) * 16 + fb + B * DY_B + ((ks >> 1) * PW + (ks & 1) * 16) * DYP
                        else:
                            r, s2 = kind // 3, kind % 3
                            a = X_OFF + (8 * lh + fp) * XP + ((4 * wi + fo) ^ (4 * (((fp + s2) >> 1) & 1))) * 16 + fb + B * X_B + (((ks >> 1) + r) * XW + (ks & 1) * 16 + s2) * XP
                        addrs.append(a)
                    for extra in (0, 4 * (DYP if kind == 9 else XP)):
                        for half in range(2):
                            banks = []
                            for lane in range(32 * half, 32 * half + 32):
                                a = addrs[lane] + extra
                                banks += [(a // 4) % 64, (a // 4 + 1) % 64]
                            worst = max(worst, max(banks.count(b) for b in set(banks)))
    print("worst bank multiplicity per 32-lane half:", worst)
conflicts()
