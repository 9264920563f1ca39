"""Seeded synthetic CU-batch streams.

The reference ships no bitstreams and no encoder, so the path's input - the per-CU records that entropy
decoding and MV derivation produce (XEVD_CU_DATA, src_base/xevd_def.h:1145-1190) - is generated directly:
a quad-tree per 64x64 CTU (boundary CUs forced to split, xevd.c:918-1017 behaviour), inter / intra mode,
reference indices, quarter-pel MVs with all four sub-pel classes and a share of vectors that point outside
the picture (MV-clip path), per-CU QP, coded-block flags and sparse low-frequency coefficient blocks.
Everything is numpy-vectorised so that 8K pictures (about 2 M SCUs) generate in well under a second.
"""
import numpy as np

from .abi import MODE_INTER, MODE_INTRA


def _morton(xs, ys):
    """z-order key of SCU coordinates inside a CTU (4 bits each is enough for 128/4 = 32 -> use 5)."""
    k = np.zeros(xs.shape, np.int64)
    for b in range(5):
        k |= ((xs >> b) & 1) << (2 * b)
        k |= ((ys >> b) & 1) << (2 * b + 1)
    return k


def gen_partition(rng, width, height, log2_ctu=6, split_prob=0.5, min_log2=2):
    """Quad-tree leaf CUs of one picture in decode order (CTU raster, z-order inside a CTU)."""
    ctu = 1 << log2_ctu
    w_ctu, h_ctu = (width + ctu - 1) // ctu, (height + ctu - 1) // ctu
    gx, gy = np.meshgrid(np.arange(w_ctu) * ctu, np.arange(h_ctu) * ctu)
    x, y = gx.ravel().astype(np.int64), gy.ravel().astype(np.int64)
    size = ctu
    leaves = []
    while True:
        inside = (x < width) & (y < height)
        x, y = x[inside], y[inside]
        crosses = (x + size > width) | (y + size > height)
        if size > (1 << min_log2):
            split = crosses | (rng.random(x.shape) < split_prob)
        else:
            split = np.zeros(x.shape, bool)
            assert not crosses.any(), "picture size must be a multiple of the minimum CU size"
        leaves.append((x[~split], y[~split], np.full((~split).sum(), size)))
        if not split.any():
            break
        half = size // 2
        px, py = x[split], y[split]
        x = np.concatenate([px, px + half, px, px + half])
        y = np.concatenate([py, py, py + half, py + half])
        size = half
    x = np.concatenate([l[0] for l in leaves])
    y = np.concatenate([l[1] for l in leaves])
    s = np.concatenate([l[2] for l in leaves])
    ctu_idx = (y >> log2_ctu) * w_ctu + (x >> log2_ctu)
    key = ctu_idx * (1 << 12) + _morton((x & (ctu - 1)) >> 2, (y & (ctu - 1)) >> 2)
    order = np.argsort(key, kind="stable")
    x, y, s, ctu_idx = x[order], y[order], s[order], ctu_idx[order]
    start = np.searchsorted(ctu_idx, np.arange(w_ctu * h_ctu + 1)).astype(np.uint32)
    log2s = np.log2(s).astype(np.uint8)
    return x.astype(np.uint16), y.astype(np.uint16), log2s, log2s.copy(), start


def gen_partition_btt(rng, width, height, log2_ctu=6, split_prob=0.5, btt_frac=0.6, min_log2=2, max_ratio=4):
    """Leaf CUs of a quad + binary + ternary split tree (the Main profile's BTT, split modes of src_base/xevd_def.h:776-785)
    in decode order: children of a node are visited first-to-last, so the order key is the path of child indices."""
    ctu = 1 << log2_ctu
    mn = 1 << min_log2
    w_ctu, h_ctu = (width + ctu - 1) // ctu, (height + ctu - 1) // ctu
    gx, gy = np.meshgrid(np.arange(w_ctu) * ctu, np.arange(h_ctu) * ctu)
    x, y = gx.ravel().astype(np.int64), gy.ravel().astype(np.int64)
    w = np.full(x.shape, ctu, np.int64)
    h = w.copy()
    key = np.arange(len(x), dtype=np.int64)                 # CTU raster index, then 2 bits per tree level
    depth = np.zeros(len(x), np.int64)
    MAXD = 12
    leaves = []
    while len(x):
        inside = (x < width) & (y < height)
        x, y, w, h, key, depth = (v[inside] for v in (x, y, w, h, key, depth))
        cx, cy = x + w > width, y + h > height
        # candidate modes: 0 none, 1 quad, 2 binary ver, 3 binary hor, 4 ternary ver, 5 ternary hor
        avail = np.stack([~(cx | cy),
                          (w == h) & (w > mn),
                          (w > mn) & (h <= max_ratio * (w // 2)),
                          (h > mn) & (w <= max_ratio * (h // 2)),
                          (w >= 4 * mn) & (h <= max_ratio * (w // 4)) & ~(cx | cy),
                          (h >= 4 * mn) & (w <= max_ratio * (h // 4)) & ~(cx | cy)], 1)
        # a node crossing the picture border must split towards the border
        forced_v = cx & ~cy & (w > mn)
        forced_h = cy & ~cx & (h > mn)
        avail[forced_v] &= np.array([0, 1, 1, 0, 0, 0], bool)
        avail[forced_h] &= np.array([0, 1, 0, 1, 0, 0], bool)
        avail[forced_v, 2] = True
        avail[forced_h, 3] = True
        both = cx & cy
        avail[both] &= np.array([0, 1, 1, 1, 0, 0], bool)
        avail[both & (w >= h) & (w > mn), 2] = True
        avail[both & (h > w), 3] = True
        weight = np.array([0.0, 1.0 - btt_frac, btt_frac / 3, btt_frac / 3, btt_frac / 6, btt_frac / 6])
        score = rng.random(avail.shape) * weight * avail
        mode = np.where(score.max(1) > 0, score.argmax(1), 0)
        stay = avail[:, 0] & ((rng.random(len(x)) >= split_prob) | (mode == 0))
        mode = np.where(stay, 0, mode)
        assert (avail[np.arange(len(x)), mode]).all(), "picture size must be a multiple of the minimum CU size"
        lf = mode == 0
        leaves.append((x[lf], y[lf], w[lf], h[lf], key[lf] << (2 * (MAXD - depth[lf]))))
        nx, ny, nw, nh, nk, nd = [], [], [], [], [], []

        def child(m, i, dx, dy, cw, ch):
            nx.append(x[m] + dx); ny.append(y[m] + dy); nw.append(cw); nh.append(ch)
            nk.append(key[m] * 4 + i); nd.append(depth[m] + 1)
        m = mode == 1
        for i in range(4):
            child(m, i, (i & 1) * (w[m] // 2), (i >> 1) * (h[m] // 2), w[m] // 2, h[m] // 2)
        m = mode == 2
        for i in range(2):
            child(m, i, i * (w[m] // 2), 0 * w[m], w[m] // 2, h[m])
        m = mode == 3
        for i in range(2):
            child(m, i, 0 * w[m], i * (h[m] // 2), w[m], h[m] // 2)
        m = mode == 4
        for i, (o, f) in enumerate(((0, 4), (1, 2), (3, 4))):
            child(m, i, o * (w[m] // 4), 0 * w[m], w[m] // f, h[m])
        m = mode == 5
        for i, (o, f) in enumerate(((0, 4), (1, 2), (3, 4))):
            child(m, i, 0 * w[m], o * (h[m] // 4), w[m], h[m] // f)
        x, y, w, h, key, depth = (np.concatenate(v) for v in (nx, ny, nw, nh, nk, nd))
        assert len(depth) == 0 or depth.max() <= MAXD
    x, y, w, h, key = (np.concatenate([l[i] for l in leaves]) for i in range(5))
    order = np.argsort(key, kind="stable")
    x, y, w, h, key = x[order], y[order], w[order], h[order], key[order]
    ctu_idx = (y >> log2_ctu) * w_ctu + (x >> log2_ctu)
    assert (np.diff(ctu_idx) >= 0).all()
    start = np.searchsorted(ctu_idx, np.arange(w_ctu * h_ctu + 1)).astype(np.uint32)
    return x.astype(np.uint16), y.astype(np.uint16), np.log2(w).astype(np.uint8), np.log2(h).astype(np.uint8), start


def gen_frame(rng, width, height, bit_depth=8, log2_ctu=6, inter_frac=1.0, bi_frac=0.0, coded_frac=0.6,
              n_refs=(1, 0), qp_range=(22, 37), mv_sigma_px=8.0, oob_frac=0.05, split_prob=0.5,
              chroma_qp_table=None, max_level=24, amp=2.0, ats_frac=0.0, ats_inter_frac=0.0, btt_frac=0.0, min_log2=2, eipd=False, partition=None, admvp=False):
    """One picture's CU batch as a dict of numpy arrays (layout of xgpu_cu_batch, include/xevd_hip.h)."""
    tree = None
    if partition is not None:      # leaf CUs given by the caller (x, y, log2w, log2h, ctu_cu_start[, tree: 0 / 1 luma-only / 2 chroma-only CUs of local dual trees])
        x, y, l2w, l2h, start = partition[:5]
        tree = partition[5] if len(partition) > 5 else None
    elif btt_frac > 0:
        x, y, l2w, l2h, start = gen_partition_btt(rng, width, height, log2_ctu, split_prob, btt_frac)
    else:
        x, y, l2w, l2h, start = gen_partition(rng, width, height, log2_ctu, split_prob, min_log2)
    n = len(x)
    w = (1 << l2w.astype(np.int64))
    h = (1 << l2h.astype(np.int64))

    pred_mode = np.where(rng.random(n) < inter_frac, MODE_INTER, MODE_INTRA).astype(np.uint8)
    if tree is not None:
        pred_mode[tree != 0] = MODE_INTRA
    inter = pred_mode != MODE_INTRA

    # reference indices: uni L0, uni L1 or bi
    refi = np.full((n, 2), -1, np.int8)
    has_l1 = n_refs[1] > 0
    bi = inter & has_l1 & (rng.random(n) < bi_frac)
    if admvp:      # with sps->tool_admvp a CU of width + height <= 12 (4x4, 4x8, 8x4) cannot be bi-predicted (xevdm_check_bi_applicability, src_main/xevdm_util.c:1086-1098):
        bi &= (w + h) > 12      # the parser never hands such a CU over, so the benchmark's batches do not hold one either
    use_l1_only = inter & has_l1 & ~bi & (rng.random(n) < 0.3)
    l0 = inter & ~use_l1_only
    refi[l0, 0] = rng.integers(0, max(n_refs[0], 1), l0.sum())
    l1 = bi | use_l1_only
    if has_l1:
        refi[l1, 1] = rng.integers(0, n_refs[1], l1.sum())

    # motion vectors: quarter-pel, 25% each of integer / H-only / V-only / 2-D phase, some far outside
    mv = np.zeros((n, 2, 2), np.int16)
    for lst in range(2):
        ipart = np.round(rng.normal(0, mv_sigma_px, (n, 2))).astype(np.int64)
        cls = rng.integers(0, 4, n)
        frac = rng.integers(1, 4, (n, 2))
        frac[:, 0] *= ((cls == 1) | (cls == 3))
        frac[:, 1] *= ((cls == 2) | (cls == 3))
        v = ipart * 4 + frac
        oob = rng.random(n) < oob_frac
        far = rng.integers(-(max(width, height) + 300), max(width, height) + 300, (n, 2)) * 4 + rng.integers(0, 4, (n, 2))
        v[oob] = far[oob]
        v = np.clip(v, -32768, 32767)
        mv[:, lst, :] = np.where((refi[:, lst] >= 0)[:, None], v, 0)

    # QPs: per-CU luma QP (sequence range), chroma through the mapping table, +6*(bd-8) like core->qp_*
    qp_y = rng.integers(qp_range[0], qp_range[1] + 1, n)
    tbl = np.asarray(chroma_qp_table if chroma_qp_table is not None else DEFAULT_CHROMA_QP_BASE, np.int64)
    qp_c = tbl[np.clip(qp_y, 0, 57)]
    boff = 6 * (bit_depth - 8)
    qp = np.stack([qp_y + boff, qp_c + boff, qp_c + boff], 1).astype(np.uint8)

    # coded block flags
    coded = rng.random(n) < coded_frac
    cbf = np.zeros(n, np.uint8)
    cbf |= (coded & (rng.random(n) < 0.9)).astype(np.uint8)
    cbf |= (coded & (rng.random(n) < 0.5)).astype(np.uint8) << 1
    cbf |= (coded & (rng.random(n) < 0.5)).astype(np.uint8) << 2
    if tree is not None:
        cbf = np.where(tree == 1, cbf & 1, np.where(tree == 2, cbf & 6, cbf)).astype(np.uint8)

    # ATS-inter (Main): a coded inter CU of 8..64 may code one half/quarter TU (xevdm_check_ats_inter_info_coded,
    # src_main/xevdm_util.c:3565-3583); ats_inter_info = idx | pos << 4, idx 1/3 vertical split, 2/4 horizontal split
    ats_inter = None
    tu_w, tu_h = w.copy(), h.copy()
    if ats_inter_frac > 0:
        ok = inter & (cbf != 0) & (l2w <= 6) & (l2h <= 6) & (rng.random(n) < ats_inter_frac)
        avail = np.stack([w >= 8, h >= 8, w >= 16, h >= 16], 1) & ok[:, None]       # idx 1, 2, 3, 4
        pick = rng.random((n, 4)) * avail
        idx = np.where(avail.any(1), pick.argmax(1) + 1, 0)
        pos = rng.integers(0, 2, n)
        ats_inter = np.where(idx > 0, idx | (pos << 4), 0).astype(np.uint8)
        tu_w = np.where(idx == 1, w // 2, np.where(idx == 3, w // 4, w))
        tu_h = np.where(idx == 2, h // 2, np.where(idx == 4, h // 4, h))

    # coefficient arena: CU-contiguous, components in Y,U,V order, coded ones only
    area_y = tu_w * tu_h
    area_c = area_y // 4
    sizes = np.stack([area_y * (cbf & 1), area_c * ((cbf >> 1) & 1), area_c * ((cbf >> 2) & 1)], 1)
    cu_size = sizes.sum(1)
    coef_off = np.concatenate([[0], np.cumsum(cu_size)[:-1]]).astype(np.int64)
    n_coef = int(cu_size.sum())
    coef = np.zeros(max(n_coef, 1), np.int16)
    # transform-block table: one entry per coded TB.  TBs are at most 64x64 (chroma 32x32): a larger CU carries up to
    # four sub-blocks sb = (j<<1)|i inside its CU-strided coefficient block (xevd_itdq.c:544-621), each with its own
    # coded flag (cbf_sub bit 4*c+sb)
    big = (l2w > 6) | (l2h > 6)
    cbf_sub = np.zeros(n, np.uint16)
    if big.any():
        sub = rng.integers(1, 16, (n, 3)).astype(np.uint16)          # at least one coded sub-block per coded component
        exists = np.zeros((n, 4), bool)
        for sb in range(4):
            exists[:, sb] = ((sb & 1) < np.where(l2w > 6, 2, 1)) & ((sb >> 1) < np.where(l2h > 6, 2, 1))
        emask = (exists * (1 << np.arange(4))).sum(1).astype(np.uint16)
        for c in range(3):
            v = sub[:, c] & emask
            v = np.where(v == 0, emask & (~emask + 1), v)              # keep the lowest existing one if the draw missed
            cbf_sub |= (np.where(((cbf >> c) & 1).astype(bool), v, 0).astype(np.uint16) << (4 * c))
        cbf_sub = np.where(big, cbf_sub, 0).astype(np.uint16)
    tb_off, tb_w, tb_h, tb_qp, tb_stride = [], [], [], [], []
    run = coef_off.copy()
    for c in range(3):
        cw_c, ch_c = (tu_w >> (1 if c else 0)), (tu_h >> (1 if c else 0))
        tw = np.minimum(cw_c, 32 if c else 64)
        th = np.minimum(ch_c, 32 if c else 64)
        for sb in range(4):
            si, sj = sb & 1, sb >> 1
            m = ((cbf >> c) & 1).astype(bool) & (si * tw < cw_c) & (sj * th < ch_c)
            m &= ~big | (((cbf_sub >> (4 * c + sb)) & 1).astype(bool))
            tb_off.append((run + sj * th * cw_c + si * tw)[m])
            tb_w.append(tw[m]); tb_h.append(th[m]); tb_stride.append(cw_c[m])
            tb_qp.append(qp[m, c].astype(np.int64))
        run = run + sizes[:, c]
    tb_off, tb_w, tb_h, tb_qp, tb_stride = (np.concatenate(v) for v in (tb_off, tb_w, tb_h, tb_qp, tb_stride))
    # level cap per block: keep the dequantised magnitude near the sample range (amp * 2^bd) like a real
    # encoder's output, which is also the range where the reference's C and SIMD paths agree;
    # amp=None leaves levels uncapped (stress cases, compared against the normative C path only)
    if amp is not None and len(tb_off):
        l2 = np.log2(tb_w).astype(np.int64) + np.log2(tb_h).astype(np.int64)
        odd = l2 & 1
        shift = 6 - (15 - bit_depth - (l2 >> 1)) + 8 * odd
        gain = (np.array([40, 45, 51, 57, 64, 71])[tb_qp % 6] << (tb_qp // 6)) * np.where(odd, 181, 1) / (2.0 ** shift)
        tb_cap = np.clip(np.floor(amp * (1 << bit_depth) / gain), 1, max_level).astype(np.int64)
    else:
        tb_cap = np.full(len(tb_off), max_level, np.int64)
    if len(tb_off):
        nnz = np.minimum(1 + rng.geometric(0.25, len(tb_off)), 24)
        tb = np.repeat(np.arange(len(tb_off)), nnz)
        # positions: concentrated at low frequencies, occasionally anywhere in the block
        fx = np.abs(rng.normal(0, 1.0, len(tb))) * tb_w[tb] / 5.0
        fy = np.abs(rng.normal(0, 1.0, len(tb))) * tb_h[tb] / 5.0
        anywhere = rng.random(len(tb)) < 0.05
        fx = np.where(anywhere, rng.random(len(tb)) * tb_w[tb], fx)
        fy = np.where(anywhere, rng.random(len(tb)) * tb_h[tb], fy)
        # 64-point dimensions carry coefficients only in their first 32 positions (the range real streams use;
        # the reference's AVX 64-point IQT stage drops the rest while its C version does not)
        px = np.minimum(fx.astype(np.int64), np.minimum(tb_w[tb], 32) - 1)
        py = np.minimum(fy.astype(np.int64), np.minimum(tb_h[tb], 32) - 1)
        lev = np.round(rng.laplace(0, 2.0, len(tb))).astype(np.int64)
        lev[lev == 0] = 1
        lev = np.clip(lev, -tb_cap[tb], tb_cap[tb])
        coef[tb_off[tb] + py * tb_stride[tb] + px] = lev
        # the first coefficient of every coded block is DC-ish and non-zero
        dc = np.round(rng.laplace(0, 6.0, len(tb_off))).astype(np.int64)
        dc[dc == 0] = 1
        coef[tb_off] = np.clip(dc, -tb_cap, tb_cap)

    # ATS (Main): intra CUs up to 32x32 may code their luma TB with DST-VII / DCT-VIII (bit0 on, bit1 v type, bit2 h type)
    ats = None
    if ats_frac > 0:
        on = (~inter) & (l2w <= 5) & (l2h <= 5) & (rng.random(n) < ats_frac)
        ats = (on.astype(np.uint8) | (rng.integers(0, 4, n).astype(np.uint8) << 1) * on).astype(np.uint8)
    ipm = np.zeros((n, 2), np.uint8)
    ipm[:, 0] = rng.integers(0, 5, n)
    # the Baseline syntax has no chroma mode of its own (core->ipm[1] = luma mode, xevd_eco.c:1154); a few CUs differ anyway
    ipm[:, 1] = np.where(rng.random(n) < 0.8, ipm[:, 0], rng.integers(0, 5, n))
    if eipd:      # sps->tool_eipd: 33 luma modes (DC, planar, bilinear, angular), chroma DM / BI / DC / HOR / VER
        ipm[:, 0] = np.where(rng.random(n) < 0.25, rng.integers(0, 3, n), rng.integers(0, 33, n))
        ipm[:, 1] = rng.integers(0, 5, n)
    return {
        "x": x, "y": y, "log2w": l2w, "log2h": l2h, "pred_mode": pred_mode, "refi": refi, "mv": mv, "qp": qp,
        "tree": tree if tree is not None and tree.any() else None, "cbf": cbf, "cbf_sub": cbf_sub if big.any() else None, "ats": ats, "ats_inter": ats_inter, "ipm": ipm, "coef_off": coef_off.astype(np.uint32), "coef": coef[:max(n_coef, 1)],
        "ctu_cu_start": start, "n_coef": n_coef,
    }


# xevd_tbl_qp_chroma_adjust_base (src_base/xevd_tbl.c:345-354): the Baseline default chroma QP mapping, a
# constant of the MPEG-5 EVC specification.
DEFAULT_CHROMA_QP_BASE = [
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19,
    20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 29, 30, 31, 32, 32, 33, 33, 34, 34,
    35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 39, 39, 40, 40, 40, 41, 41, 41]


def gen_partition_tree(rng, width, height, allowed, split_prob=0.5, log2_ctu=6, inter_only=None, dual_tree=False):
    """Leaf CUs (decode order) of a split tree whose nodes only take the splits `allowed(x, y, log2w, log2h)` reports - [none, binary with a vertical cut,
    binary horizontal, ternary vertical, ternary horizontal], entries 0 / 1 / 2 - e.g. StreamWriter.split_allowed for a stream with sps_btt_flag.  A split
    reported as 2 is only taken when `inter_only` is a list: the leaves below it must then be inter CUs, their indices are appended to that list.
    With `dual_tree` a split reported as 2 or 3 may also start a local dual tree: the leaves below it become luma-only CUs (tree 1) and the node's chroma-only CU
    (tree 2) follows them; a sixth array `tree` is then returned.
    -> x, y, log2w, log2h, ctu_cu_start like gen_partition [, tree]"""
    ctu = 1 << log2_ctu
    w_ctu, h_ctu = (width + ctu - 1) // ctu, (height + ctu - 1) // ctu
    xs, ys, lws, lhs, start, trees = [], [], [], [], [], []

    def node(x, y, lw, lh, forced=False, dual=False):
        if x >= width or y >= height:
            return
        raw = allowed(x, y, lw, lh)
        a = [int(v == 1 or (v == 2 and (inter_only is not None or forced or dual or dual_tree)) or (v == 3 and (dual or (dual_tree and not forced)))) for v in raw]
        inside = x + (1 << lw) <= width and y + (1 << lh) <= height
        opts = [m for m in range(1, 5) if a[m]]
        if inside and a[0] and (not opts or rng.random() >= split_prob):
            if forced:
                inter_only.append(len(xs))
            xs.append(x); ys.append(y); lws.append(lw); lhs.append(lh); trees.append(1 if dual else 0)
            return
        assert opts, f"no way to split the node {x},{y} {1 << lw}x{1 << lh}"
        m = opts[int(rng.integers(0, len(opts)))] if inside else opts[0]
        ver = m in (1, 3)
        sizes = [1, 1] if m < 3 else [2, 1, 2]
        off = 0
        # a split that constrains its children: inter CUs only (P / B pictures), or a local dual tree
        starts_dual = False
        if raw[m] >= 2 and not forced and not dual:
            starts_dual = raw[m] == 3 or inter_only is None or (dual_tree and rng.random() < 0.5)
        for sh in sizes:
            clw, clh = (lw - sh, lh) if ver else (lw, lh - sh)
            node(x + off if ver else x, y if ver else y + off, clw, clh, forced or (raw[m] == 2 and not starts_dual and not dual), dual or starts_dual)
            off += 1 << (clw if ver else clh)
        if starts_dual:
            xs.append(x); ys.append(y); lws.append(lw); lhs.append(lh); trees.append(2)
    for cy in range(h_ctu):
        for cx in range(w_ctu):
            start.append(len(xs))
            node(cx * ctu, cy * ctu, log2_ctu, log2_ctu)
    start.append(len(xs))
    out = (np.array(xs, np.uint16), np.array(ys, np.uint16), np.array(lws, np.uint8), np.array(lhs, np.uint8), np.array(start, np.uint32))
    return out + (np.array(trees, np.uint8),) if dual_tree else out


def add_affine(rng, batch, frac=0.5):
    """Turn a share of the inter CUs of at least 8x8 into affine CUs (Main, xevdm_affine_mc): 2 or 3 control points, the first = the CU's vector,
    the others a few quarter-pel steps away so that every branch occurs - no spread (the whole CU as one block), spreads of a sample across the CU
    (32/16/8-sample sub-blocks), stronger ones (per-sample interpolation, EIF), and rotations/shears too strong for EIF (8x8 sub-blocks)."""
    n = len(batch["x"])
    inter = (batch["pred_mode"] != MODE_INTRA) & (batch["pred_mode"] != 6)      # not intra, not intra block copy
    ok = inter & (batch["log2w"] >= 3) & (batch["log2h"] >= 3) & (rng.random(n) < frac)
    aff = np.where(ok, rng.integers(2, 4, n), 0).astype(np.uint8)
    cp = np.zeros((n, 2, 3, 2), np.int64)
    scale = np.array([0, 1, 2, 3, 6, 12, 24, 60, 160])[rng.integers(0, 9, n)]
    for lst in range(2):
        base = batch["mv"][:, lst, :].astype(np.int64)
        cp[:, lst, 0] = base
        for v in (1, 2):
            d = rng.integers(-1, 2, (n, 2)) * scale[:, None] + rng.integers(-1, 2, (n, 2)) * (scale[:, None] // 3)
            cp[:, lst, v] = base + d
        cp[:, lst] *= (batch["refi"][:, lst] >= 0)[:, None, None]
    batch["affine"] = aff
    batch["affine_mv"] = (np.clip(cp, -32768, 32767) * (aff != 0)[:, None, None, None]).astype(np.int16)
    return batch


def add_dmvr(rng, batch, frac=0.6, mirror_frac=0.5):
    """Flag a share of the plain inter CUs as DMVR candidates (merge-mode CUs of a stream with sps->tool_dmvr: xgpu_cu_batch.dmvr).  Most of them
    bi-predicted and at least 8x8 (the backend refines those whose two references lie at equal POC distances on either side of the picture),
    some not - the flag alone must not change them.  The flagged bi-predicted CUs get moderate vectors (inside the picture + margin), `mirror_frac`
    of them list-1 vectors that mirror list 0 up to a sample or two - the situation the refinement is for."""
    n = len(batch["x"])
    aff = batch.get("affine")
    plain = (batch["pred_mode"] != MODE_INTRA) & (batch["pred_mode"] != 6) & ((aff == 0) if aff is not None else True)
    flag = plain & (rng.random(n) < frac)
    bi = flag & (batch["refi"][:, 0] >= 0) & (batch["refi"][:, 1] >= 0)
    mv = batch["mv"].astype(np.int64)
    mv[bi] = np.clip(mv[bi], -96, 96)
    mir = bi & (rng.random(n) < mirror_frac)
    mv[mir, 1] = -mv[mir, 0] + rng.integers(-6, 7, (int(mir.sum()), 2))
    batch["mv"] = mv.astype(np.int16)
    batch["dmvr"] = flag.astype(np.uint8)
    return batch


MODE_IBC = 6


def add_ibc(rng, batch, width, height, log2_ctu=6, frac=0.3, max_log2=6):
    """Turn a share of the CUs into intra-block-copy CUs (Main, xevdm_IBC_mc): pred_mode 6, mv[0] = a whole-sample block vector into the part of
    the CURRENT picture that is already reconstructed when the CU's turn comes - CTU rows above, CTUs to the left in the same row, or an earlier,
    larger CU of the same CTU (decoding order = batch order).  Odd vectors included (chroma uses the halved vector)."""
    n = len(batch["x"])
    S = 1 << log2_ctu
    x, y = batch["x"].astype(np.int64), batch["y"].astype(np.int64)
    w, h = 1 << batch["log2w"].astype(np.int64), 1 << batch["log2h"].astype(np.int64)
    ctu = (y // S) * ((width + S - 1) // S) + x // S
    ai = batch.get("ats_inter")
    aff = batch.get("affine")
    pick = rng.random(n) < frac
    for i in range(n):
        if not pick[i] or (ai is not None and ai[i]) or (w[i] > (1 << max_log2) or h[i] > (1 << max_log2)):
            continue
        opts = []
        cy0, cx0 = (y[i] // S) * S, (x[i] // S) * S
        if cy0 >= h[i]:
            opts.append((0, width - w[i], 0, cy0 - h[i]))
        if cx0 >= w[i]:
            opts.append((0, cx0 - w[i], cy0, min(cy0 + S, height) - h[i]))
        same = np.nonzero((ctu[:i] == ctu[i]) & (w[:i] >= w[i]) & (h[:i] >= h[i]))[0]
        if len(same):
            j = int(same[rng.integers(0, len(same))])
            opts.append((x[j], x[j] + w[j] - w[i], y[j], y[j] + h[j] - h[i]))
        if not opts:
            continue
        x0, x1, y0, y1 = opts[rng.integers(0, len(opts))]
        sx, sy = int(rng.integers(x0, x1 + 1)), int(rng.integers(y0, y1 + 1))
        batch["pred_mode"][i] = MODE_IBC
        batch["refi"][i] = -1
        batch["mv"][i] = 0
        batch["mv"][i, 0] = (sx - x[i], sy - y[i])
        if batch.get("ats") is not None:
            batch["ats"][i] = 0
        if aff is not None:
            aff[i] = 0
            batch["affine_mv"][i] = 0
    return batch


def gen_picture(rng, width, height, bit_depth=8, smooth=True):
    """A synthetic reference picture (active area only): smooth gradients + texture + noise, 4:2:0."""
    maxv = (1 << bit_depth) - 1
    planes = []
    for (pw, ph) in ((width, height), (width // 2, height // 2), (width // 2, height // 2)):
        yy, xx = np.mgrid[0:ph, 0:pw]
        base = 0.5 + 0.25 * np.sin(xx / (17.0 + 13 * rng.random())) * np.cos(yy / (11.0 + 9 * rng.random()))
        base += 0.15 * np.sin((xx + 2 * yy) / (5.0 + 4 * rng.random()))
        base += rng.normal(0, 0.06, (ph, pw))
        planes.append(np.clip(np.round(base * maxv), 0, maxv).astype(np.int16))
    return planes


def gen_alf_params(rng, n_ctu, across_tiles=0, ctb_on_frac=0.8, enable=(1, 1, 1), strength=40):
    """Random ALF parameters in the form alf_recon_coef leaves them (src_main/xevdm_alf.c:700-794): 25 luma filters
    of 13 coefficients and one chroma filter of 7, each with unity gain at 9 fractional bits (the centre tap is
    512 minus twice the sum of the others)."""
    luma = rng.integers(-strength, strength + 1, (25, 13)).astype(np.int64)
    luma[:, 12] = 512 - 2 * luma[:, :12].sum(1)
    chroma = rng.integers(-strength, strength + 1, 7).astype(np.int64)
    chroma[6] = 512 - 2 * chroma[:6].sum()
    flag = (rng.random(n_ctu) < ctb_on_frac).astype(np.uint8)
    return {"enable": tuple(enable), "luma_coef": luma.astype(np.int16), "chroma_coef": chroma.astype(np.int16),
            "ctb_flag": flag, "across_tiles": across_tiles}
