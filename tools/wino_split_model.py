"""Go / no-go arithmetic for Winograd convolutions in split-operand arithmetic on the fp16 MFMA (round-4 verdict, item 1).

    python tools/wino_split_model.py            (CPU only; prints profiles/r05_experiments/wino_split_model.txt)

Part 1 -- ACCURACY.  A 128-channel 3x3 layer in float64 (reference), and in the arithmetic each candidate kernel would execute:
the input transform in float32, the transformed weights computed in float64, both split h = fp16(a), l = fp16((a - h) * 2^11),
the three fp16 x fp16 products accumulated in float32, the output transform in float32.  (Accuracy is NOT what decides.)

Part 2 -- RESOURCES of one workgroup step on gfx950, from the constants of /opt/skills/guides/MI355X_MICROARCH.md:
160 KiB LDS per CU, 512 registers per lane and SIMD (256 per wave at two waves per SIMD), v_mfma_f32_32x32x16_f16 = 16,384 MAC in
32 cycles, LDS reads 256 B/clk/CU, wide LDS writes ~80 B/clk/CU, VALU 16 lanes/clk/SIMD with 8 of an MFMA's 32 cycles taken by its
issue.  For each candidate: accumulator registers, LDS bytes of ONE step's operands (the kernel needs two buffers to overlap the
LDS-DMA with the MFMAs), weight bytes streamed from L2 per MFMA cycle, LDS traffic per MFMA cycle, vector instructions per MFMA.
The direct kernel (conv_split_kernel<1,12,64,3,2,2>) is the first row: the model reproduces its known numbers.
"""
import numpy as np

F16, F32, F64 = np.float16, np.float32, np.float64


def split(a):
    a = a.astype(F32)
    h = a.astype(F16)
    l = ((a - h.astype(F32)) * F32(2048.0)).astype(F16)
    return h, l


def split_dot(wh, wl, xh, xl):
    """sum_k w x with three exact fp16 products per term, fp32 accumulation (two accumulators, as conv_split_kernel.h)."""
    f = lambda a, b: np.einsum('ok,kp->op', a.astype(F32), b.astype(F32), dtype=F32)
    return f(wh, xh) + (f(wh, xl) + f(wl, xh)) * F32(2.0 ** -11)


def direct64(x, w):
    C, H, W = x.shape
    y = np.zeros((w.shape[0], H - 2, W - 2), F64)
    for dy in range(3):
        for dx in range(3):
            y += np.einsum('oc,chw->ohw', w[:, :, dy, dx].astype(F64), x[:, dy:dy + H - 2, dx:dx + W - 2].astype(F64))
    return y


def direct_split(x, w):
    C, H, W = x.shape
    xh, xl = split(x)
    wh, wl = split(w)
    y = np.zeros((w.shape[0], H - 2, W - 2), F32)
    for dy in range(3):
        for dx in range(3):
            s = (slice(None), slice(dy, dy + H - 2), slice(dx, dx + W - 2))
            y += split_dot(wh[:, :, dy, dx], wl[:, :, dy, dx], xh[s].reshape(C, -1), xl[s].reshape(C, -1)).reshape(y.shape)
    return y


# Winograd matrices (Lavin & Gray): F(2,3) and F(4,3)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], F64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], F64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], F64)
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], F64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], F64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], F64)


def wino2d_split(x, w):
    """F(2x2, 3x3): 16 positions per 2x2 patch."""
    C, H, W = x.shape
    O = w.shape[0]
    U = np.einsum('ai,ocij,bj->aboc', G2, w.astype(F64), G2)          # float64 on the host, then split
    Uh, Ul = split(U)
    ph, pw = (H - 2) // 2, (W - 2) // 2
    d = np.stack([np.stack([x[:, i:i + 2 * ph:2, j:j + 2 * pw:2] for j in range(4)]) for i in range(4)])      # [4][4][C][ph][pw]
    B = BT2.astype(F32)
    V = np.einsum('ai,ijcpq->ajcpq', B, d.astype(F32), dtype=F32)
    V = np.einsum('bj,ajcpq->abcpq', B, V, dtype=F32)
    Vh, Vl = split(V)
    M = np.zeros((4, 4, O, ph * pw), F32)
    for a in range(4):
        for b in range(4):
            M[a, b] = split_dot(Uh[a, b], Ul[a, b], Vh[a, b].reshape(C, -1), Vl[a, b].reshape(C, -1))
    A = AT2.astype(F32)
    Y = np.einsum('ia,abop->ibop', A, M, dtype=F32)
    Y = np.einsum('jb,ibop->ijop', A, Y, dtype=F32).reshape(2, 2, O, ph, pw)
    y = np.zeros((O, 2 * ph, 2 * pw), F32)
    for i in range(2):
        for j in range(2):
            y[:, i::2, j::2] = Y[i, j]
    return y


def wino1d_split(x, w, BT, G, AT):
    """F(m, 3) along x, the three kernel rows direct: alpha = m + 2 positions per m output columns."""
    m, alpha = AT.shape
    C, H, W = x.shape
    O = w.shape[0]
    U = np.einsum('aj,ocyj->yaoc', G, w.astype(F64))                   # [dy][position][O][C]
    Uh, Ul = split(U)
    nb = (W - 2) // m
    d = np.stack([x[:, :, j:j + m * nb:m] for j in range(alpha)])        # [alpha][C][H][nb]
    V = np.einsum('aj,jchq->achq', BT.astype(F32), d.astype(F32), dtype=F32)
    Vh, Vl = split(V)
    Mq = np.zeros((alpha, O, H - 2, nb), F32)
    for a in range(alpha):
        for dy in range(3):
            Mq[a] += split_dot(Uh[dy, a], Ul[dy, a], Vh[a][:, dy:dy + H - 2].reshape(C, -1), Vl[a][:, dy:dy + H - 2].reshape(C, -1)).reshape(O, H - 2, nb)
    Y = np.einsum('ja,aohq->johq', AT.astype(F32), Mq, dtype=F32)
    y = np.zeros((O, H - 2, m * nb), F32)
    for j in range(m):
        y[:, :, j::m] = Y[j]
    return y


def accuracy():
    rng = np.random.default_rng(5)
    C = 128
    x = rng.standard_normal((C, 14, 26)).astype(F32)
    x = (x / (1 + np.exp(-x))).astype(F32)                              # SiLU'd activations, as conv2 sees them
    w = (rng.standard_normal((C, C, 3, 3)) / (3 * C ** 0.5)).astype(F32)
    ref = direct64(x, w)
    rows = []
    for name, y in [("direct, split operands (the shipped kernel's arithmetic)", direct_split(x, w)),
                    ("Winograd F(2x2,3x3), split operands", wino2d_split(x, w)),
                    ("Winograd F(2,3) along x, split operands", wino1d_split(x, w, BT2, G2, AT2)),
                    ("Winograd F(4,3) along x, split operands", wino1d_split(x, w, BT4, G4, AT4))]:
        r = ref[:, :y.shape[1], :y.shape[2]]
        e = np.abs(y.astype(F64) - r)
        rows.append((name, e.max() / np.abs(r).max(), np.sqrt((e ** 2).mean()) / np.abs(r).max()))
    return rows


# ------------------------------------------------------------------------------------------------------------------------------
LDS, REGS_WAVE, MAC_PER_MFMA, CYC_PER_MFMA = 160 * 1024, 256, 32 * 32 * 16, 32


def candidate(name, px, cout, positions, taps_per_pos, acc_per_px, parts_w, parts_v, halo, valu_per_px_cin, frag_pairs_per_block, n_acc):
    """One 16-channel step of a 512-thread workgroup (8 waves, two per SIMD).
    px: output pixels of the tile; positions: Winograd positions (1 = direct); taps_per_pos: products per position, pixel-equivalent
    and (cin, cout); acc_per_px: accumulators per output pixel and output channel relative to the direct form; halo: staged input
    pixels per output pixel; frag_pairs_per_block: (weight + pixel) fragment pairs read from LDS per product block of the wave's loop."""
    blocks = px * acc_per_px / 32 * (cout / 32)                       # 32 x 32 accumulator blocks of the tile
    acc_regs = blocks * 16 * n_acc / 8
    mfma = blocks * taps_per_pos * 3                                  # three MFMAs per fp32 product block
    cyc = mfma * CYC_PER_MFMA / 4                                     # per SIMD at 100 % issue
    w_bytes = positions * taps_per_pos * 16 * cout * 2 * parts_w
    v_bytes = px * halo * acc_per_px * 16 * 2 * parts_v               # the staged (transformed) input of the step
    lds_read = mfma / 3 * frag_pairs_per_block * 1024 * (parts_w + parts_v) / 2
    valu = px * halo * 16 * valu_per_px_cin / 64 / 4 * 4              # cycles per SIMD (a wave instruction issues in 4)
    return dict(name=name, px=px, cout=cout, acc_regs=acc_regs, lds_step=w_bytes + v_bytes, lds_two=2 * (w_bytes + v_bytes), mfma_cyc=cyc,
                w_stream=w_bytes / cyc, lds_rd=lds_read / cyc, lds_wr=(w_bytes + v_bytes) / cyc, valu_frac=valu / (cyc * 24 / 32))


def resources():
    rows = [
        # direct: 12 x 32 px x 64 cout, 9 taps; a pixel fragment serves three (row, dy) pairs: (15 + 9) / 27 pairs per block
        candidate("direct 3x3, 12x32 px x 64 (shipped; conv2 form: input by LDS-DMA)", 384, 64, 1, 9, 1, 2, 2, 14 * 34 / 384, 0, 24 / 27, 2),
        # F(2x2,3x3): 64 patches = 256 px, 16 positions, one product per position; accumulators 16 per 4 px; V = 16 values per 4 px
        candidate("F(2x2,3x3), 64 patches x 64, one accumulator (weights P,h,l)", 256, 64, 16, 1, 4, 3, 2, 1.0, 100 / 4, 2.0, 1),
        candidate("F(2x2,3x3), 64 patches x 64, two accumulators", 256, 64, 16, 1, 4, 2, 2, 1.0, 100 / 4, 2.0, 2),
        candidate("F(2x2,3x3), 128 patches x 32, one accumulator", 512, 32, 16, 1, 4, 3, 2, 1.0, 100 / 4, 2.0, 1),
        # F(4,3) along x: 6 positions per 4 columns, three kernel rows direct: 18 products per 4 px; accumulators 6 per 4 px
        candidate("F(4,3) along x, 2 rows x 128 px x 64, two accumulators", 256, 64, 6, 3, 1.5, 2, 2, 4 / 2, 15, (4 + 3) / 6, 2),
        candidate("F(4,3) along x, 4 rows x 128 px x 64, one accumulator (weights P,h,l)", 512, 64, 6, 3, 1.5, 3, 2, 6 / 4, 15, (6 + 3) / 12, 1),
        # F(2,3) along x: 4 positions per 2 columns: 12 products per 2 px; accumulators 4 per 2 px
        candidate("F(2,3) along x, 3 rows x 64 px x 64, two accumulators", 192, 64, 4, 3, 2, 2, 2, 5 / 3, 11, (5 + 3) / 9, 2),
        candidate("F(2,3) along x, 6 rows x 64 px x 64, one accumulator (weights P,h,l)", 384, 64, 4, 3, 2, 3, 2, 8 / 6, 11, (8 + 3) / 18, 1),
    ]
    return rows


if __name__ == "__main__":
    print("== accuracy against a float64 convolution, 128 -> 128 channels, relative to max|ref| (max / rms) ==")
    for name, mx, rms in accuracy():
        print(f"  {name:62s} {mx:9.2e} {rms:9.2e}")
    print()
    print("== one 16-channel step of a 512-thread workgroup: what it needs against what a CU has ==")
    print("   (limits: accumulators + ~90 other registers <= 256; TWO operand buffers <= 160 KiB; weight stream: the shipped kernel takes 7 B/clk at 100 % MFMA issue --")
    print("    3-5 in practice -- on every one of 256 CUs; LDS reads <= 256 B/clk, writes (DMA + ds_write) <= ~80-128 B/clk; VALU share of the issue")
    print("    slots the MFMAs leave free <= 1)")
    print(f"  {'candidate':74s} {'acc regs':>8s} {'LDS 1 buf':>9s} {'2 bufs':>8s} {'W B/clk':>8s} {'rd B/clk':>8s} {'wr B/clk':>8s} {'VALU':>5s}")
    for r in resources():
        flag = []
        if r['acc_regs'] + 90 > REGS_WAVE: flag.append("registers")
        if r['lds_two'] > LDS: flag.append("LDS capacity")
        if r['w_stream'] > 40: flag.append("L2 weight stream")
        if r['lds_wr'] > 100: flag.append("LDS write rate")
        if r['valu_frac'] > 1.0: flag.append("VALU")
        print(f"  {r['name']:74s} {r['acc_regs']:8.0f} {r['lds_step'] / 1024:8.0f}K {r['lds_two'] / 1024:7.0f}K {r['w_stream']:8.1f} {r['lds_rd']:8.0f} {r['lds_wr']:8.0f} "
              f"{r['valu_frac']:5.2f}   {'NO: ' + ', '.join(flag) if flag else 'fits'}")
