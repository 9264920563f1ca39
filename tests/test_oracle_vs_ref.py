"""Pins the CPU oracle (oracle/xevd_oracle.c) against the REAL reference compiled in oracle/_ref.

Runs only where oracle/_ref exists (development container and, because built .so files travel, the GPU box).
Block level: the reference's exported per-block functions (plain-C tables = normative; AVX also checked where
the survey found them identical).  Picture level: oracle/ref_harness.c drives the reference's own
xevd_sub_block_itdq / xevd_mc / xevd_recon / xevd_deblock_cu_* over the same CU batch.
"""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

import oracle_lib as ol
from xevd_amd import abi, synth  # noqa: F401

pytestmark = pytest.mark.ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_transform_tables_match_reference():
    lib, orc = ol.ref(), ol.oracle()
    for l in range(1, 7):
        n = 1 << l
        t_ref = np.frombuffer((C.c_int8 * (n * n)).in_dll(lib, f"xevd_tbl_tm{n}"), np.int8)
        t_orc = np.ctypeslib.as_array(orc.orc_tm(l), (n * n,))
        assert np.array_equal(t_ref, t_orc), f"tm{n}"


def test_ats_tables_match_reference():
    import ctypes as C
    lib, orc = ol.ref(), ol.oracle()
    lib.xevdm_init_multi_tbl()
    orc.orc_ats_tm.restype = C.POINTER(C.c_int16)
    orc.orc_ats_tm.argtypes = [C.c_int, C.c_int]
    for l in range(2, 6):
        n = 1 << l
        t_ref = np.frombuffer((C.c_int16 * (2 * n * n)).in_dll(lib, f"xevd_tbl_tr{n}"), np.int16).reshape(2, n * n)
        for typ in (0, 1):
            assert np.array_equal(t_ref[typ], np.ctypeslib.as_array(orc.orc_ats_tm(typ, l), (n * n,))), (n, typ)


def _set_mc_tables(lib, admvp):
    lp = C.c_void_p.in_dll(lib, "tbl_mc_l_coeff")
    cp = C.c_void_p.in_dll(lib, "tbl_mc_c_coeff")
    lp.value = C.addressof((C.c_int16 * 128).in_dll(lib, "tbl_mc_l_coeff_main" if admvp else "xevd_tbl_mc_l_coeff"))
    cp.value = C.addressof((C.c_int16 * 128).in_dll(lib, "tbl_mc_c_coeff_main" if admvp else "xevd_tbl_mc_c_coeff"))


@pytest.mark.parametrize("admvp", [0, 1])
@pytest.mark.parametrize("bd", [8, 10, 12])
def test_mc_blocks(admvp, bd):
    lib, orc = ol.ref(), ol.oracle()
    rng = np.random.default_rng(100 + admvp * 2 + bd)
    _set_mc_tables(lib, admvp)
    plane = rng.integers(0, 1 << bd, (200, 260)).astype(np.int16)
    s = plane.shape[1]
    names = {(0, 0): "00", (1, 0): "n0", (0, 1): "0n", (1, 1): "nn"}
    for luma in (1, 0):
        for trial in range(120):
            w = 1 << rng.integers(2 if luma else 1, 8 if luma else 7)
            h = 1 << rng.integers(2 if luma else 1, 8 if luma else 7)
            w, h = int(min(w, 128 if luma else 64)), int(min(h, 128 if luma else 64))
            has_dx, has_dy = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            prec = 4 if luma else 5
            step = 4  # baseline tables only populate every 4th phase; keep main on the same grid half the time
            if admvp and rng.random() < 0.5:
                step = 1
            fx = int(rng.integers(0, (1 << prec) // step)) * step
            fy = int(rng.integers(0, (1 << prec) // step)) * step
            ix, iy = int(rng.integers(8, 260 - 8 - w - 8)), int(rng.integers(8, 200 - 8 - h - 8))
            gx, gy = (ix << prec) + fx, (iy << prec) + fy
            a = np.zeros((h, w), np.int16)
            b = np.zeros((h, w), np.int16)
            fn = getattr(lib, f"xevd_mc_{'l' if luma else 'c'}_{names[(has_dx, has_dy)]}")
            fn(_p(plane), gx, gy, s, w, _p(a), w, h, bd)
            (orc.orc_mc_l if luma else orc.orc_mc_c)(_p(plane), gx, gy, s, w, _p(b), w, h, bd, has_dx, has_dy, admvp)
            assert np.array_equal(a, b), (luma, w, h, has_dx, has_dy, fx, fy)
            # the AVX/SSE variants agree with plain C on in-range samples (survey 4)
            if w >= 4 and h >= 4:
                suffix = "_avx" if (luma and names[(has_dx, has_dy)] != "00") or (not luma and names[(has_dx, has_dy)] == "nn") else "_sse"
                simd = getattr(lib, f"xevd_mc_{'l' if luma else 'c'}_{names[(has_dx, has_dy)]}{suffix}", None)
                if simd is not None:
                    c = np.zeros((h, w), np.int16)
                    simd(_p(plane), gx, gy, s, w, _p(c), w, h, bd)
                    assert np.array_equal(a, c), ("simd", luma, w, h, has_dx, has_dy)


def _dq_params(log2w, log2h, qp, bd, iqt):
    tbl = [40, 45, 51, 57, 64, 72] if iqt else [40, 45, 51, 57, 64, 71]
    scale = tbl[qp % 6] << (qp // 6)
    tr_shift = 15 - bd - ((log2w + log2h) >> 1)
    shift = 20 - 14 - tr_shift + (8 if (log2w + log2h) & 1 else 0)
    offset = 0 if shift == 0 else 1 << (shift - 1)
    return scale, offset, shift


@pytest.mark.parametrize("iqt", [0, 1])
@pytest.mark.parametrize("bd", [8, 10, 12])
def test_itdq_all_sizes(iqt, bd):
    lib, orc = ol.ref(), ol.oracle()
    rng = np.random.default_rng(7 + iqt + bd)
    itxb = (C.c_void_p * 6).in_dll(lib, "xevd_tbl_itxb")
    itx = (C.c_void_p * 6).in_dll(lib, "xevdm_tbl_itx")
    f_itxb = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int)
    f_itx = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int)
    lib.xevd_dquant.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_uint8]
    for log2w in range(1, 7):
        for log2h in range(1, 7):
            w, h = 1 << log2w, 1 << log2h
            for trial in range(6):
                qp = int(rng.integers(0, 52)) + 6 * (bd - 8)
                coef = np.zeros((h, w), np.int16)
                # last trial: a few huge levels -> exercises the dequant s16 clip.  Kept sparse because the
                # reference's second stage multiplies s8*s32 in 32-bit int (xevd_itdq.c:60-72,...): dense
                # saturated blocks overflow int there (undefined behaviour, outside the conformant range).
                mag = 2000 if trial == 5 else 40
                nnz = int(rng.integers(1, 4)) if trial == 5 else int(rng.integers(1, max(2, w * h // 4)))
                ys, xs = rng.integers(0, h, nnz), rng.integers(0, w, nnz)
                coef[ys, xs] = rng.integers(-mag, mag + 1, nnz)
                scale, offset, shift = _dq_params(log2w, log2h, qp, bd, iqt)
                if not iqt:
                    # stay where the reference's 32-bit second-stage products are defined: halve the levels
                    # until sum_k 90*|stage1[k]| < 2^31 (stage-1 magnitudes bounded through the dequantised block)
                    while True:
                        ns = 181 if (log2w + log2h) & 1 else 1
                        dq = np.clip((coef.astype(np.int64) * scale * ns + offset) >> shift, -32768, 32767)
                        stage1_bound = 90 * np.abs(dq).sum(0).max()           # per column, all rows
                        if 90 * stage1_bound * min(w, int((np.abs(dq).sum(0) > 0).sum())) < 2 ** 31:
                            break
                        coef = (coef // 2).astype(np.int16)
                a = coef.copy()
                b = coef.copy()
                lib.xevd_dquant(_p(a), log2w, log2h, scale, offset, shift)
                if iqt:
                    t = np.zeros(w * h, np.int16)
                    f_itx(itx[log2h - 1])(_p(a), _p(t), 7, w)
                    f_itx(itx[log2w - 1])(_p(t), _p(a), 12 - (bd - 8), h)
                else:
                    t = np.zeros(w * h, np.int32)
                    f_itxb(itxb[log2h - 1])(_p(a), _p(t), 0, w, 0)
                    f_itxb(itxb[log2w - 1])(_p(t), _p(a), 7 + 12 - (bd - 8), h, 1)
                orc.orc_itdq(_p(b), log2w, log2h, qp, bd, iqt)
                assert np.array_equal(a, b), (log2w, log2h, qp, trial)


def test_recon_wraps_like_reference():
    lib, orc = ol.ref(), ol.oracle()
    rng = np.random.default_rng(3)
    for bd in (8, 10, 12):
        for is_coef in (0, 1):
            w, h = 16, 8
            pred = rng.integers(0, 1 << bd, (h, w)).astype(np.int16)
            coef = rng.integers(-32768, 32768, (h, w)).astype(np.int16)   # includes sums that wrap in s16
            a = np.zeros((h, 40), np.int16)
            b = np.zeros((h, 40), np.int16)
            lib.xevd_recon(_p(coef), _p(pred), is_coef, w, h, 40, _p(a), bd)
            orc.orc_recon(_p(coef), _p(pred), is_coef, w, h, 40, _p(b), bd)
            assert np.array_equal(a, b)


def test_deblock_segments():
    lib, orc = ol.ref(), ol.oracle()
    rng = np.random.default_rng(11)
    for bd in (8, 10, 12):
        for trial in range(400):
            st = int(rng.integers(1, 13)) << (bd - 8)
            base = rng.integers(0, 1 << bd)
            blk = np.clip(base + rng.integers(-40, 41, (12, 12)) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(np.int16)
            for is_ver in (0, 1):
                a, b = blk.copy(), blk.copy()
                off = 4 * 12 + 4
                fn = lib.deblock_scu_ver if is_ver else lib.deblock_scu_hor
                fn(C.c_void_p(a.ctypes.data + 2 * off), st, 12, bd - 8, 1)
                orc.orc_dbk_luma(C.c_void_p(b.ctypes.data + 2 * off), st, 12, bd, is_ver)
                assert np.array_equal(a, b)
                au, av, bu, bv = blk.copy(), blk.T.copy(), blk.copy(), blk.T.copy()
                st_u, st_v = int(rng.integers(0, 13)) << (bd - 8), int(rng.integers(0, 13)) << (bd - 8)
                fn = lib.deblock_scu_ver_chroma if is_ver else lib.deblock_scu_hor_chroma
                fn(C.c_void_p(au.ctypes.data + 2 * off), C.c_void_p(av.ctypes.data + 2 * off), st_u, st_v, 12, bd - 8, 1)
                orc.orc_dbk_chroma(C.c_void_p(bu.ctypes.data + 2 * off), C.c_void_p(bv.ctypes.data + 2 * off), st_u, st_v, 12, bd, is_ver)
                assert np.array_equal(au, bu) and np.array_equal(av, bv)


import cases


@pytest.mark.parametrize("case", cases.CASES, ids=[c[0] for c in cases.CASES])
def test_picture_level_oracle_equals_reference(case):
    cs = cases.build_case(*case)
    a, a_pre, ma, ra = cases.run_cpu("oracle", cs)
    b, b_pre, mb, rb = cases.run_cpu("ref", cs)
    assert np.array_equal(ra, rb), "residual arena"
    for c in range(3):
        assert np.array_equal(a_pre.active(c), b_pre.active(c)), f"recon plane {c}"
    assert np.array_equal(ma.map_scu & 0x7FFFFFFF, mb.map_scu & 0x7FFFFFFF)
    assert np.array_equal(ma.map_refi, mb.map_refi) and np.array_equal(ma.map_mv, mb.map_mv)
    for c in range(3):
        assert np.array_equal(a.bufs[c], b.bufs[c]), f"deblocked+padded plane {c}"
    # the reference's AVX/SSE tables give the same picture on these conformant-range inputs (all-inter: the
    # SIMD recon kernels scribble past narrow blocks, which a real decode repairs with the next CU)
    if cs["addb"]:
        return      # ADDB is scalar C only in the reference
    tools1 = dict(case[8]) if len(case) > 8 else {}
    tools1["inter_frac"] = 1.0
    cs1 = cases.build_case(*case[:8], tools1)
    c0, _, _, r0 = cases.run_cpu("ref", cs1, simd=0)
    s0, _, _, r1 = cases.run_cpu("ref", cs1, simd=1)
    assert np.array_equal(r0, r1), "simd residual"
    for c in range(3):
        assert np.array_equal(s0.bufs[c], c0.bufs[c]), f"simd plane {c}"


def test_bench_workload_4k_oracle_equals_reference():
    """the oracle pinned at size on the benchmark's own Main 4K batch (10 bit, admvp tables, IQT, ADDB, ALF, two lists): the picture the
    reference's functions produce; the 8K batch of the same generator is compared on the GPU box (bench.py's bit_exact field)"""
    cs = cases.bench_case("cfg3_main_4k_10b_ra")
    a, _, _, ra = cases.run_cpu("oracle", cs)
    b, _, _, rb = cases.run_cpu("ref", cs)
    assert np.array_equal(ra, rb)
    for c in range(3):
        assert np.array_equal(a.bufs[c], b.bufs[c])


@pytest.mark.parametrize("split_prob", [0.0, 1.0])
def test_picture_level_extreme_partitions(split_prob):
    """all-64x64 CUs and all-4x4 CUs (the longest chroma deblocking dependency chains)"""
    cs = cases.build_case("extreme", 136, 72, 8, 0, 0, (1, 1), 0.3, None, seed=int(split_prob), split_prob=split_prob, qp_range=(30, 50))
    a, _, _, ra = cases.run_cpu("oracle", cs)
    b, _, _, rb = cases.run_cpu("ref", cs)
    assert np.array_equal(ra, rb)
    for c in range(3):
        assert np.array_equal(a.bufs[c], b.bufs[c])


@pytest.mark.parametrize("offs", [(0, 0), (6, -4), (-6, 5)])
def test_addb_with_slice_offsets_and_high_qp(offs):
    """alpha/beta offsets go through get_index()'s u8 arguments (negative ones saturate the index); QP up to 51"""
    cs = cases.build_case("addb_offs", 136, 72, 10, 1, 1, (2, 2), 0.5, {"addb": 1, "alpha_off": offs[0], "beta_off": offs[1]},
                          seed=offs[0] + 7, qp_range=(30, 51))
    a, _, _, _ = cases.run_cpu("oracle", cs)
    b, _, _, _ = cases.run_cpu("ref", cs)
    for c in range(3):
        assert np.array_equal(a.bufs[c], b.bufs[c]), f"plane {c}"


@pytest.mark.parametrize("src_bd,dst_bd", [(10, 8), (8, 8), (8, 10), (10, 10), (12, 10), (12, 8), (10, 12)])
def test_output_conversion(src_bd, dst_bd):
    """orc_output_convert == the application's imgb_cpy_codec_to_out (app/xevd_app_util.h:665-708) on whole pictures,
    extremes included."""
    if not os.path.exists(ol.REF_OUTPUT_SO):
        pytest.skip("oracle/_ref/libref_output.so not built")
    rng = np.random.default_rng(src_bd * 16 + dst_bd)
    w, h = 72, 40
    planes = [rng.integers(0, 1 << src_bd, (h >> (i > 0), w >> (i > 0))).astype(np.int16) for i in range(3)]
    planes[0][0, :4] = [0, (1 << src_bd) - 1, (1 << src_bd) - 2, 1]
    assert np.array_equal(ol.output_convert(planes, src_bd, dst_bd), ol.ref_output_convert(planes, src_bd, dst_bd))


@pytest.mark.parametrize("name", sorted(ol.DRA_SETS))
def test_dra_apply(name):
    """orc_dra_apply == the reference's DRA sample processing (xevdm_dra.c:272-355, order of xevd_apply_filter) with the tables the
    real xevd_init_dra builds from the signalled parameters; the tables themselves become a golden fixture for the GPU test."""
    rng = np.random.default_rng(len(name))
    w, h = 72, 40
    planes = [rng.integers(0, 1024, (h >> (i > 0), w >> (i > 0))).astype(np.int16) for i in range(3)]
    planes[0][0, :4] = [0, 1023, 1, 1022]
    planes[1][0, :4] = [0, 1023, 512, 511]
    luts, ref = ol.ref_dra(name, 10, planes)
    ours = ol.dra_apply(planes, luts)
    for c in range(3):
        assert np.array_equal(ours[c], ref[c]), f"plane {c}"
    assert not np.array_equal(ours[0], planes[0]) and not np.array_equal(ours[1], planes[1])     # the filter does something


def test_md5_goldens_are_plain_md5_of_the_tight_planes():
    """The picture-signature fixture (tests/golden/md5_pictures.json: the reference's xevd_md5_imgb, src_base/xevd_util.c:985-1002, run on seeded pictures by
    tests/golden/make_md5_golden.py) == RFC 1321 over every plane's rows of width x 2 bytes - the message definition k_md5.hip's GPU test is held to; when
    oracle/_ref is present the fixture is also re-made and compared."""
    import hashlib
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_md5_golden as g
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "md5_pictures.json")))
    assert len(gold) == len(g.CASES)
    have_ref = os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "libxevd_ref.so"))
    for seed, w, h, bd in g.CASES:
        planes = g.md5_picture(seed, w, h, bd)
        key = f"{seed}_{w}x{h}_{bd}b"
        assert [hashlib.md5(p.astype("<i2").tobytes()).hexdigest() for p in planes] == gold[key]
        if have_ref and w <= 400:
            assert g.reference_digests(planes) == gold[key]
