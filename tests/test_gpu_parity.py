"""GPU suite (run with -m gpu on an MI355X): the HIP backend, driven through the C ABI, against
  (1) the committed golden vectors the reference produced (tests/golden/*.npz),
  (2) the CPU oracle on further seeded inputs incl. extreme partitions and out-of-range stress levels,
  (3) size-independent properties at BASELINE.json's full picture sizes.
Bar: bit-exact (integer path)."""
import os

import numpy as np
import pytest

import cases
import golden_io
from xevd_amd import abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def decs():
    from xevd_amd.decoder import XgpuDecoder
    d = {(admvp, iqt): XgpuDecoder(64, 64, 8, admvp=admvp, iqt=iqt, max_pics=2) for admvp in (0, 1) for iqt in (0, 1)}
    yield d
    for v in d.values():
        v.close()


def test_native_library_is_loaded():
    lib = abi.load()
    assert lib.xgpu_version().startswith(b"xevd_amd")
    maps = open("/proc/self/maps").read()
    assert "libxevd_hip.so" in maps


@pytest.mark.parametrize("fname", ["blocks_mc.npz", "blocks_mc_12b.npz"])
def test_gpu_mc_blocks_golden(decs, fname):
    d = np.load(os.path.join(golden_io.GOLDEN, fname))
    for bd, admvp, luma, has_dx, has_dy, w, h, gx, gy, off in d["recs"]:
        plane = d[f"plane_bd{bd}"]
        out = decs[(int(admvp), 0)].test_mc(plane, 0, 0, int(has_dx), int(has_dy), int(gx), int(gy), int(w), int(h), int(bd), bool(luma))
        assert np.array_equal(out.ravel(), d["pred"][off:off + w * h]), (bd, admvp, luma, has_dx, has_dy, w, h, gx, gy)


@pytest.mark.parametrize("fname", ["blocks_itdq.npz", "blocks_itdq_12b.npz"])
def test_gpu_itdq_blocks_golden(decs, fname):
    d = np.load(os.path.join(golden_io.GOLDEN, fname))
    for iqt, bd, log2w, log2h, qp, off in d["recs"]:
        n = 1 << (log2w + log2h)
        out = decs[(0, int(iqt))].test_itdq(d["coef"][off:off + n], int(log2w), int(log2h), [int(qp)], int(bd))
        assert np.array_equal(out, d["resid"][off:off + n]), (iqt, bd, log2w, log2h, qp)


def test_gpu_fine_grained_recon_and_deblock_shims(decs):
    """The remaining per-block slots of the reference's function table - fn_recon, fn_dbk[HOR / VER], fn_dbk_chroma[HOR / VER] (src_base/xevd_def.h:
    363-364, 1466-1468) - through the kernels' own residual add and line filters, against the oracle's xevd_recon / deblock_scu_* (which
    tests/test_oracle_vs_ref.py pins to the reference's functions): sums that wrap in s16, strengths 0..12 scaled by the bit depth."""
    import ctypes as C
    import oracle_lib as ol
    orc = ol.oracle()
    dec = decs[(0, 0)]
    rng = np.random.default_rng(31)
    for bd in (8, 10, 12):
        for is_coef in (0, 1):
            for (w, h) in ((16, 8), (4, 4), (64, 32), (2, 2)):
                # predictions outside the sample range too: without coefficients the reference clips them (xevd_recon.c:41-48), e.g. the
                # Baseline DC of a non-square block next to an unavailable side
                pred = rng.integers(-300, (1 << bd) + 300, (h, w)).astype(np.int16)
                coef = rng.integers(-32768, 32768, (h, w)).astype(np.int16)
                exp = rng.integers(0, 100, (h, w + 24)).astype(np.int16)
                got = dec.test_recon(coef, pred, is_coef, exp, bd)
                orc.orc_recon(coef.ctypes.data, pred.ctypes.data, is_coef, w, h, w + 24, exp.ctypes.data, bd)
                assert np.array_equal(got, exp), (bd, is_coef, w, h)
        for trial in range(60):
            st = int(rng.integers(1, 13)) << (bd - 8)
            base = rng.integers(0, 1 << bd)
            blk = np.clip(base + rng.integers(-40, 41, (12, 12)) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(np.int16)
            for is_ver in (0, 1):
                exp = blk.copy()
                orc.orc_dbk_luma(C.c_void_p(exp.ctypes.data + 2 * (4 * 12 + 4)), st, 12, bd, is_ver)
                assert np.array_equal(dec.test_dbk(blk, 4, 4, st, not is_ver, bd), exp), (bd, st, is_ver)
                eu, ev = blk.copy(), blk.T.copy()
                st_u, st_v = int(rng.integers(0, 13)) << (bd - 8), int(rng.integers(0, 13)) << (bd - 8)
                orc.orc_dbk_chroma(C.c_void_p(eu.ctypes.data + 2 * (4 * 12 + 4)), C.c_void_p(ev.ctypes.data + 2 * (4 * 12 + 4)), st_u, st_v, 12, bd, is_ver)
                gu, gv = dec.test_dbk(blk, 4, 4, st_u, not is_ver, bd, plane_v=blk.T.copy(), st_v=st_v)
                assert np.array_equal(gu, eu) and np.array_equal(gv, ev), (bd, st_u, st_v, is_ver)


def test_gpu_itdq_many_blocks_per_wave(decs):
    """several TBs share a wave (64/W per wave): every block of a batch must come out like its golden twin"""
    d = np.load(os.path.join(golden_io.GOLDEN, "blocks_itdq.npz"))
    recs = [r for r in d["recs"] if r[0] == 0 and r[1] == 8 and r[2] == 2 and r[3] == 3]
    coef = np.concatenate([d["coef"][r[5]:r[5] + 32] for r in recs] * 7)
    exp = np.concatenate([d["resid"][r[5]:r[5] + 32] for r in recs] * 7)
    qp = [int(r[4]) for r in recs] * 7
    out = decs[(0, 0)].test_itdq(coef, 2, 3, qp, 8)
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("name", golden_io.PICTURE_CASES)
def test_gpu_pictures_golden(name):
    case, exp = golden_io.load_picture_case(name)
    pre = cases.run_gpu(case, deblock=False, pad=False, alf=False)
    for c in range(3):
        pad = abi.PAD_L if c == 0 else abi.PAD_C
        got = pre[c][pad:-pad, pad:-pad]
        assert np.array_equal(got, exp["pre"][c]), f"recon plane {c}: {np.argwhere(got != exp['pre'][c])[:4]}"
    out, resid = cases.run_gpu(case, resid=True)
    assert np.array_equal(resid[:len(exp["resid"])], exp["resid"]), "residual arena (dequant + inverse transform of every coded TB, intra CUs included)"
    for c in range(3):
        assert np.array_equal(out[c], exp["out"][c]), f"final plane {c}: {np.argwhere(out[c] != exp['out'][c])[:4]}"
    if exp["dmvr_mv"] is not None:      # the vectors the reference keeps for temporal prediction (refined where DMVR ran)
        _, mvs = cases.run_gpu(case, dmvr=True)
        assert np.array_equal(mvs, exp["dmvr_mv"]), f"DMVR vectors: {np.argwhere(mvs != exp['dmvr_mv'])[:4]}"


@pytest.mark.parametrize("name", [n for n in golden_io.PICTURE_CASES if "addb" in n or "alf" in n or "all_tools" in n or "ctu128_10b" in n or "12b" in n])
def test_gpu_pictures_golden_scalar_deblocking(name, monkeypatch):
    """k_addb_alf<false>: the scalar ADDB line filters - the instantiation every picture above 10 bits takes - on the ADDB / ALF goldens of every bit depth
    (XEVD_HIP_ADDB_SCALAR is read when a context opens)"""
    monkeypatch.setenv("XEVD_HIP_ADDB_SCALAR", "1")
    case, exp = golden_io.load_picture_case(name)
    out = cases.run_gpu(case)
    for c in range(3):
        assert np.array_equal(out[c], exp["out"][c]), f"final plane {c}: {np.argwhere(out[c] != exp['out'][c])[:4]}"


@pytest.mark.parametrize("w,h,bd,tools", [(64, 192, 8, {}), (1088, 256, 10, {"addb": 1, "alf": 1}), (64, 128, 10, {"addb": 1, "alf": 1}), (2112, 128, 8, {"addb": 1, "alf": 1}),
                                          (8, 264, 8, {}), (1032, 72, 10, {"addb": 1, "alf": 1, "log2_ctu": 7})],
                         ids=["64x192", "1088x256", "64x128_filters", "2112x128", "8x264", "1032x72_ctu128"])
def test_gpu_one_region_wide_last_strip(w, h, bd, tools):
    """pictures whose last strip of 64x64 regions is ONE region wide (regions_x % 16 == 1: widths 1..64, 1025..1088, 2049..2112 - e.g. portrait 1080 x 1920) and pictures
    one filter tile wide: the index -> (row, column) maps of k_inter and k_addb_alf divide by that width with a multiplication, and floor(2^32 / 1) + 1 does not fit"""
    main = 1 if tools else 0
    cs = cases.build_case(f"strip_{w}x{h}", w, h, bd, main, main, (1, 1), 0.4, dict(tools, inter_frac=0.8))
    ref, _, _, _ = cases.run_cpu("oracle", cs)
    out = cases.run_gpu(cs)
    for c in range(3):
        assert np.array_equal(out[c], ref.bufs[c]), f"{w}x{h} plane {c}: {np.argwhere(out[c] != ref.bufs[c])[:4]}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_io.PICTURE_CASES)
def test_gpu_pictures_golden_residual_pass_ahead(name):
    """xgpu_batch_recon_ahead: the residual pass of a batch queued with the PREVIOUS picture's kernels (inside its data-flow intra launch, k_intra_itdq, when it
    has one and the sequence uses IQT; as a plain launch otherwise) leaves the same residuals and the same picture"""
    case, exp = golden_io.load_picture_case(name)
    out, resid = cases.run_gpu(case, resid=True, ahead=True)
    assert np.array_equal(resid[:len(exp["resid"])], exp["resid"])
    for c in range(3):
        assert np.array_equal(out[c], exp["out"][c]), f"final plane {c}: {np.argwhere(out[c] != exp['out'][c])[:4]}"


@pytest.mark.gpu
@pytest.mark.parametrize("admvp,bd", [(0, 8), (1, 10)], ids=["base_taps_8b", "main_taps_10b"])
def test_gpu_split_role_whole_sample_vectors(admvp, bd):
    """k_inter's split role: a lane whose vector has a whole-sample component sits out of the window rows / samples its identity taps multiply by zero
    (mc_scu_list).  Pictures of small CUs (4x4 .. 16x16, mixed inside every 32x32 tile) whose vectors are forced, CU by CU, through every combination of
    whole / fractional luma and chroma phases in x and y, both lists, against the oracle (xevd_mc.c:469-557, xevdm_mc.c:1860-2038)."""
    for seed in range(3):
        cs = cases.build_case(f"split_phases_{admvp}", 200, 136, bd, admvp, 0, (2, 2), 0.5, {"inter_frac": 1.0, "split_prob": 0.85 if seed else 1.0, "coded_frac": 0.3}, seed=seed, oob_frac=0.2)
        mv = cs["batch"]["mv"]                                   # [n_cu][list][x / y], quarter samples
        k = np.arange(mv.shape[0])
        for l in range(2):
            for d in range(2):
                sel = (k >> (2 * l + d + seed)) % 4              # 0: as drawn, 1: whole luma sample, 2: whole chroma sample (multiple of 8), 3: half a luma sample
                v = mv[:, l, d].astype(np.int32)
                v = np.where(sel == 1, v & ~3, np.where(sel == 2, v & ~7, np.where(sel == 3, (v & ~3) | 2, v)))
                mv[:, l, d] = v.astype(mv.dtype)
        ref, _, _, _ = cases.run_cpu("oracle", cs)
        out = cases.run_gpu(cs)
        for c in range(3):
            assert np.array_equal(out[c], ref.bufs[c]), f"seed {seed} plane {c}: {np.argwhere(out[c] != ref.bufs[c])[:4]}"


@pytest.mark.gpu
def test_gpu_intra_level1_sixteen_lanes_per_cu(monkeypatch):
    """k_intra_l1: level-1 CUs of at most 16 SCUs reconstructed by 16 lanes each, four per wave (Baseline predictors; xevd_ipred.c:95-161, 587-676).  Large pictures take
    that launch by themselves (cfg3 / cfg4 workloads, the 4K and 8K tests); here XEVD_HIP_INTRA_SMALL_MIN=1 (read per context) sends every picture golden without EIPD
    / IBC / HTDF through it, and random pictures of small and oddly shaped CUs (binary / ternary splits: 32x8, 64x4, 4x16 ...) next to unavailable picture borders."""
    monkeypatch.setenv("XEVD_HIP_INTRA_SMALL_MIN", "1")
    n = 0
    for case_name in golden_io.PICTURE_CASES:
        case, exp = golden_io.load_picture_case(case_name)
        if case.get("eipd") or not (case["batch"]["pred_mode"] == 0).any():
            continue
        out = cases.run_gpu(case)
        n += 1
        for c in range(3):
            assert np.array_equal(out[c], exp["out"][c]), f"{case_name}: final plane {c}"
    assert n >= 5
    for seed, (w, h, bd, kw) in enumerate([(264, 136, 8, {"split_prob": 0.9, "inter_frac": 0.6}), (200, 328, 10, {"split_prob": 0.7, "inter_frac": 0.5, "btt_frac": 0.8}),
                                            (136, 72, 8, {"split_prob": 1.0, "inter_frac": 0.3}), (328, 200, 10, {"split_prob": 0.5, "inter_frac": 0.8, "btt_frac": 0.6, "log2_ctu": 7})]):
        cs = cases.build_case(f"l1_small_{seed}", w, h, bd, 1 if bd == 10 else 0, 0, (1, 1), 0.3, kw, seed=seed)
        ref, _, _, _ = cases.run_cpu("oracle", cs)
        out = cases.run_gpu(cs)
        for c in range(3):
            assert np.array_equal(out[c], ref.bufs[c]), f"random {seed} plane {c}: {np.argwhere(out[c] != ref.bufs[c])[:4]}"


RANDOM = [
    # name, w, h, bd, admvp, iqt, n_refs, bi_frac, kwargs
    ("rnd_a", 264, 136, 8, 0, 0, (2, 1), 0.3, {}),
    ("rnd_b", 72, 200, 10, 1, 1, (2, 2), 0.6, {}),
    ("all_4x4", 136, 72, 8, 0, 0, (1, 1), 0.3, {"split_prob": 1.0, "qp_range": (30, 50)}),
    ("all_64", 192, 128, 8, 0, 0, (1, 1), 0.3, {"split_prob": 0.0, "qp_range": (30, 50)}),
    ("all_inter_oob", 128, 128, 10, 1, 0, (1, 1), 0.5, {"inter_frac": 1.0, "oob_frac": 0.6}),
    ("stress_levels", 128, 72, 8, 0, 0, (1, 0), 0.0, {"amp": 40.0}),
    ("stress_levels_iqt", 128, 72, 10, 0, 1, (1, 0), 0.0, {"amp": 40.0}),
    ("tiny", 8, 8, 8, 0, 0, (1, 0), 0.0, {}),
    ("btt_all_tools", 328, 200, 10, 1, 1, (2, 2), 0.5, {"inter_frac": 1.0, "tools": {"addb": 1, "alf": 1, "btt_frac": 0.8, "ats_inter_frac": 0.7, "coded_frac": 0.8}}),
    ("btt_ctu128_base_dbk", 264, 264, 8, 1, 0, (1, 1), 0.3, {"tools": {"log2_ctu": 7, "btt_frac": 0.8, "ats_inter_frac": 0.5}}),
    ("affine_all_tools", 328, 200, 10, 1, 1, (2, 2), 0.5, {"inter_frac": 0.9, "oob_frac": 0.3, "tools": {"addb": 1, "alf": 1, "btt_frac": 0.5, "ats_inter_frac": 0.5, "affine_frac": 0.9,
                                                                                                     "split_prob": 0.3, "log2_ctu": 7, "coded_frac": 0.8}}),
    ("affine_small_8b", 72, 136, 8, 1, 0, (1, 1), 0.4, {"inter_frac": 1.0, "oob_frac": 0.5, "tools": {"affine_frac": 1.0, "split_prob": 0.6}}),
    ("ibc_chains", 328, 200, 10, 1, 1, (1, 1), 0.4, {"inter_frac": 0.2, "tools": {"addb": 1, "alf": 1, "ibc_frac": 0.7, "btt_frac": 0.5, "split_prob": 0.6}}),
    ("ibc_all_tools_ctu128", 264, 264, 8, 1, 1, (2, 2), 0.5, {"inter_frac": 0.6, "tools": {"addb": 1, "ibc_frac": 0.4, "log2_ctu": 7, "eipd": 1, "affine_frac": 0.5, "ats_frac": 0.4,
                                                                                       "ats_inter_frac": 0.4, "btt_frac": 0.5, "split_prob": 0.45}}),
    ("ibc_base_dbk", 136, 72, 8, 1, 0, (1, 0), 0.0, {"inter_frac": 0.3, "tools": {"ibc_frac": 0.6, "split_prob": 0.7}}),
    ("htdf_b", 328, 200, 10, 1, 1, (2, 2), 0.5, {"inter_frac": 0.8, "tools": {"addb": 1, "alf": 1, "htdf_qp": 30, "coded_frac": 0.9, "split_prob": 0.45, "btt_frac": 0.5}}),
    ("htdf_all_tools_ctu128", 264, 264, 8, 1, 1, (1, 1), 0.4, {"inter_frac": 0.6, "tools": {"addb": 1, "htdf_qp": 40, "log2_ctu": 7, "eipd": 1, "affine_frac": 0.4, "ats_frac": 0.4, "ibc_frac": 0.2,
                                                                                        "ats_inter_frac": 0.4, "btt_frac": 0.5, "split_prob": 0.4, "coded_frac": 0.8, "constrained_intra": 1}}),
    ("htdf_i_qp20", 136, 72, 8, 1, 0, (1, 0), 0.0, {"inter_frac": 0.0, "tools": {"htdf_qp": 20, "split_prob": 0.5}}),
    ("dmvr_all_tools", 328, 200, 10, 1, 1, (2, 2), 0.8, {"inter_frac": 0.9, "oob_frac": 0.2, "tools": {"addb": 1, "alf": 1, "dmvr_frac": 0.8, "btt_frac": 0.5, "ats_inter_frac": 0.5, "coded_frac": 0.8,
                                                                                                   "split_prob": 0.35, "affine_frac": 0.2}}),
    ("dmvr_big_cus_8b", 264, 264, 8, 1, 0, (2, 2), 0.9, {"inter_frac": 1.0, "tools": {"dmvr_frac": 1.0, "log2_ctu": 7, "split_prob": 0.15}}),
    ("htdf_off_by_qp", 136, 72, 8, 1, 0, (1, 0), 0.0, {"inter_frac": 0.5, "tools": {"htdf_qp": 17}}),
]


@pytest.mark.parametrize("spec", RANDOM, ids=[s[0] for s in RANDOM])
def test_gpu_vs_oracle_random(spec):
    *case, kw = spec
    kw = dict(kw)
    tools = kw.pop("tools", None)
    for seed in range(2):
        cs = cases.build_case(*case[:8], tools, seed=seed, **kw)
        ref, _, _, _ = cases.run_cpu("oracle", cs)
        out = cases.run_gpu(cs)
        for c in range(3):
            assert np.array_equal(out[c], ref.bufs[c]), f"{spec[0]} seed {seed} plane {c}: {np.argwhere(out[c] != ref.bufs[c])[:4]}"


@pytest.mark.parametrize("size", [(1920, 1080), (3840, 2160)], ids=["1080p", "4k"])
def test_gpu_full_size_vs_oracle(size):
    """BASELINE.json picture sizes, whole pipeline, against the oracle (a few seconds of CPU)."""
    w, h = size
    cs = cases.build_case("full", w, h, 8, 0, 0, (1, 0), 0.0, None, seed=w, inter_frac=0.95)
    ref, _, _, _ = cases.run_cpu("oracle", cs)
    out = cases.run_gpu(cs)
    for c in range(3):
        assert np.array_equal(out[c], ref.bufs[c]), f"plane {c}"


def test_gpu_all_intra_1080p_vs_oracle():
    """BASELINE.json configs[0] shape (Baseline 1080p I-only): every CU intra, dependency chains a thousand CUs deep
    through the data-flow kernel; the resident batch is decoded three times (epoch-valued done flags, running ticket counter)."""
    cs = cases.build_case("intra1080", 1920, 1080, 8, 0, 0, (1, 0), 0.0, {"inter_frac": 0.0}, seed=3)
    ref, _, _, _ = cases.run_cpu("oracle", cs)
    out = cases.run_gpu(cs, repeat=3)
    for c in range(3):
        assert np.array_equal(out[c], ref.bufs[c]), f"plane {c}"


def test_gpu_ibc_intra_1080p_vs_oracle():
    """a 1080p picture of intra and intra-block-copy CUs only (Main, EIPD, ADDB): IBC CUs wait for the CUs under their source block, intra CUs for
    their neighbours - one dependency graph through the data-flow kernel; decoded three times from the resident batch"""
    cs = cases.build_case("ibc1080", 1920, 1080, 10, 1, 1, (1, 0), 0.0, {"inter_frac": 0.0, "ibc_frac": 0.5, "eipd": 1, "addb": 1, "split_prob": 0.55}, seed=5)
    assert (cs["batch"]["pred_mode"] == 6).sum() > 2000
    ref, _, _, _ = cases.run_cpu("oracle", cs)
    out = cases.run_gpu(cs, repeat=3)
    for c in range(3):
        assert np.array_equal(out[c], ref.bufs[c]), f"plane {c}"


def test_gpu_batch_rejects_malformed_tool_fields():
    """xgpu_batch_create validates what the Main-tool fields claim: an error code (never a crash), and the context keeps working"""
    from xevd_amd.decoder import XgpuDecoder, XgpuError
    cs = cases.build_case("reject", 136, 72, 10, 1, 1, (1, 1), 0.4, {"inter_frac": 0.7, "affine_frac": 0.6, "ibc_frac": 0.3, "addb": 1}, seed=2)
    good = cs["batch"]
    ibc = np.nonzero(good["pred_mode"] == 6)[0]
    aff = np.nonzero(good["affine"] != 0)[0]
    assert len(ibc) and len(aff)

    def variant(fn):
        b = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in good.items()}
        fn(b)
        return b
    bad = [
        variant(lambda b: b["mv"].__setitem__((ibc[0], 0), (-4000, 0))),                          # IBC source left of the picture
        variant(lambda b: b["mv"].__setitem__((ibc[0], 0), (0, 0))),                              # IBC source = the CU itself: not reconstructed before it
        variant(lambda b: b["mv"].__setitem__((ibc[-1], 0), (0, 64))),                            # IBC source below: later in decoding order (or outside)
        variant(lambda b: b["affine"].__setitem__(aff[0], 5)),                                    # control-point count
        variant(lambda b: b["refi"].__setitem__(aff[0], (-1, -1))),                               # affine CU without a reference
        variant(lambda b: b["affine"].__setitem__(np.nonzero(b["pred_mode"] == 0)[0][0], 2)),     # affine intra CU
        variant(lambda b: b.__setitem__("htdf_slice_qp", 77)),                                    # slice QP range
        variant(lambda b: b["pred_mode"].__setitem__(aff[0], 5)),                                 # unknown prediction mode
        variant(lambda b: b.__setitem__("tree", np.where(np.arange(len(b["x"])) == aff[0], 2, 0).astype(np.uint8))),      # a chroma-only CU of a dual tree that is an inter CU
        variant(lambda b: b.__setitem__("tree", np.where(np.arange(len(b["x"])) == np.nonzero((b["pred_mode"] == 0) & ((b["cbf"] & 6) == 0))[0][0], 1, 0).astype(np.uint8))),  # luma-only CU, no chroma-only CU after it
        variant(lambda b: b.__setitem__("tree", np.full(len(b["x"]), 3, np.uint8))),               # unknown tree type
    ]
    dec = XgpuDecoder(cs["w"], cs["h"], cs["bd"], iqt=1, admvp=1, addb=1)
    for k, b in enumerate(bad):
        with pytest.raises(XgpuError):
            dec.batch_create(b)
    dec.close() if hasattr(dec, "close") else None
    ref, _, _, _ = cases.run_cpu("oracle", cs)
    out = cases.run_gpu(cs)
    for c in range(3):
        assert np.array_equal(out[c], ref.bufs[c])


def test_gpu_htdf_1080p_vs_oracle():
    """a 1080p B picture with HTDF: every intra CU and every coded inter CU of a filterable size is a node of the dependency graph (each reads the
    final border samples of the CUs before it) - tens of thousands of nodes through the data-flow kernel; decoded three times from the resident batch"""
    cs = cases.build_case("htdf1080", 1920, 1080, 10, 1, 1, (1, 1), 0.4, {"inter_frac": 0.85, "htdf_qp": 34, "addb": 1, "coded_frac": 0.8, "split_prob": 0.5}, seed=6)
    ref, _, _, _ = cases.run_cpu("oracle", cs)
    out = cases.run_gpu(cs, repeat=3)
    for c in range(3):
        assert np.array_equal(out[c], ref.bufs[c]), f"plane {c}"


@pytest.mark.parametrize("path", sorted(__import__("glob").glob(os.path.join(golden_io.GOLDEN, "stream_*.npz"))), ids=os.path.basename)
def test_gpu_golden_streams(path):
    """real bitstreams: committed .evc bytes -> our parser -> HIP backend == the pictures the reference decoder produced"""
    import stream_util as su
    d = np.load(path)
    ours = su.decode_gpu(d["bytes"].tobytes(), verify_md5=True)      # signed streams: the player checks the MD5 SEIs as well
    assert len(ours) == int(d["n"])
    for k in range(len(ours)):
        for c in range(3):
            assert np.array_equal(ours[k][c], d[f"p{k}_{c}"]), f"picture {k} plane {c}"


def test_gpu_tiles_parsed_on_threads():
    """a tiled golden stream with the tiles of every picture parsed by four host threads: still the reference decoder's pictures"""
    import stream_util as su
    d = np.load(os.path.join(golden_io.GOLDEN, "stream_main_tiles_3x2_all_tools_10b.npz"))
    ours = su.decode_gpu(d["bytes"].tobytes(), parser_threads=4)
    assert len(ours) == int(d["n"])
    for k in range(len(ours)):
        for c in range(3):
            assert np.array_equal(ours[k][c], d[f"p{k}_{c}"]), f"picture {k} plane {c}"


def test_gpu_stream_1080p_vs_oracle():
    """BASELINE.json configs[1] shape as a real stream: 1080p Baseline IPPP written, parsed, decoded on the GPU and by the oracle"""
    import stream_util as su
    data = su.make_stream(1920, 1080, 3, seed=21, max_refs=1)
    ours, ref = su.decode_gpu(data), su.decode_oracle(data)
    assert len(ours) == 3
    for k in range(3):
        for c in range(3):
            assert np.array_equal(ours[k][c], ref[k][c]), f"picture {k} plane {c}"


SUCO_CONFIGS = [
    # sps_suco_flag beyond the golden streams, GPU against parser + oracle (the oracle is pinned to the reference decoder on such streams, tests/test_stream.py):
    # local dual trees under the BASELINE deblocking filter (order-aware chroma edges), every intra tool at once, a 1080p all-intra picture pair
    (264, 200, 5, dict(main=True, suco=(0, 2), btt=(2, 0, 0, 0), admvp=True, dual_tree=True, eipd=True, split_prob=0.7, inter_frac=0.6, max_refs=2)),
    (392, 264, 4, dict(main=True, suco=(0, 3), eipd=True, htdf=True, ibc_log_max=4, cm_init=True, adcc=True, iqt=True, ats=True, inter_frac=0.4, split_prob=0.7, bit_depth=10)),
    (392, 264, 5, dict(main=True, suco=(1, 2), admvp=True, affine=True, hmvp=True, eipd=True, addb=True, alf=True, inter_frac=0.8, split_prob=0.5, max_refs=2, log2_sub_gop=2, tiles=(2, 2, 1))),
    (1920, 1080, 2, dict(main=True, suco=(0, 2), eipd=True, idr_period=1, split_prob=0.75)),
]


@pytest.mark.parametrize("cfg", SUCO_CONFIGS, ids=[f"{c[0]}x{c[1]}x{c[2]}_{i}" for i, c in enumerate(SUCO_CONFIGS)])
def test_gpu_suco_streams_vs_oracle(cfg):
    import stream_util as su
    w, h, n, kw = cfg
    data = su.make_stream(w, h, n, seed=w + 3 * n, **kw)
    ours, ref = su.decode_gpu(data), su.decode_oracle(data)
    assert len(ours) == n and len(ref) == n
    for k in range(n):
        for c in range(3):
            assert np.array_equal(ours[k][c], ref[k][c]), f"picture {k} plane {c}: {np.argwhere(ours[k][c] != ref[k][c])[:4].tolist()}"


def test_gpu_8k_properties():
    """8K (7680x4320): identity property - zero motion, no residual, deblocking off: the picture equals its
    reference, padding included; and the run is deterministic."""
    from xevd_amd.decoder import XgpuDecoder
    w, h, bd = 7680, 4320, 10
    rng = np.random.default_rng(8)
    planes = synth.gen_picture(rng, w, h, bd)
    batch = synth.gen_frame(rng, w, h, bd, inter_frac=1.0, coded_frac=0.0, mv_sigma_px=0.0, oob_frac=0.0)
    batch["mv"][:] = 0
    with XgpuDecoder(w, h, bd, max_pics=3) as dec:
        r = dec.pic_alloc()
        dec.pic_upload(r, planes)
        dec.frame_begin(r, 0, {})
        dec.pad()
        dec.frame_end()
        cur = dec.pic_alloc()
        hb = dec.batch_create(batch)
        dec.decode_picture(cur, 1, {(0, 0): (r, 0)}, hb, deblock=False)
        a = dec.pic_download_padded(cur)
        b = dec.pic_download_padded(r)
        for c in range(3):
            assert np.array_equal(a[c], b[c])
            pad = abi.PAD_L if c == 0 else abi.PAD_C
            assert np.array_equal(a[c][pad:-pad, pad:-pad], planes[c])
            assert np.array_equal(a[c], np.pad(planes[c], pad, mode="edge"))


@pytest.mark.gpu
@pytest.mark.parametrize("src_bd,dst_bd,crop", [(10, 8, (0, 0, 0, 0)), (10, 8, (2, 6, 4, 2)), (8, 8, (0, 0, 0, 0)), (8, 10, (0, 2, 0, 0)),
                                                (10, 10, (4, 0, 2, 6)), (12, 10, (0, 0, 0, 0)), (12, 8, (2, 2, 2, 2)), (10, 12, (0, 0, 0, 0))])
def test_gpu_output_conversion(src_bd, dst_bd, crop):
    """xgpu_pic_output (device crop + bit-depth conversion + packing) == the oracle's restatement of the application's
    imgb_cpy_codec_to_out, extremes included; the oracle is pinned to the real function in test_oracle_vs_ref.py."""
    from xevd_amd.decoder import XgpuDecoder
    import oracle_lib as ol
    rng = np.random.default_rng(src_bd * 100 + dst_bd)
    w, h = 136, 72
    planes = [rng.integers(0, 1 << src_bd, (h >> (i > 0), w >> (i > 0))).astype(np.int16) for i in range(3)]
    planes[0][0, :4] = [0, (1 << src_bd) - 1, (1 << src_bd) - 2, 1]
    dec = XgpuDecoder(w, h, src_bd, device=0)
    try:
        pic = dec.pic_alloc()
        dec.pic_upload(pic, planes)
        got = dec.pic_output(pic, dst_bd, crop)
    finally:
        dec.close()
    assert np.array_equal(got, ol.output_convert(planes, src_bd, dst_bd, crop))


@pytest.mark.gpu
def test_gpu_output_rejects_bad_crop():
    from xevd_amd.decoder import XgpuDecoder
    dec = XgpuDecoder(64, 64, 8, device=0)
    try:
        pic = dec.pic_alloc()
        with pytest.raises(ValueError):
            dec.pic_output(pic, 8, (1, 0, 0, 0))          # odd crop
        with pytest.raises(ValueError):
            dec.pic_output(pic, 8, (32, 32, 0, 0))        # nothing left
    finally:
        dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["three_ranges_idx58", "five_ranges_idx40", "one_range_idx30"])
def test_gpu_dra_output(name):
    """xgpu_pic_output with DRA tables == the reference's own output (golden: xevd_init_dra tables, xevd_apply_dra_* and
    imgb_cpy_codec_to_out run in the development container) and == the oracle, with and without crop."""
    from xevd_amd.decoder import XgpuDecoder
    import oracle_lib as ol
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dra.npz"))
    planes = [d[f"in_{c}"] for c in range(3)]
    luts = d[f"{name}_luts"]
    h, w = planes[0].shape
    dec = XgpuDecoder(w, h, 10, device=0)
    try:
        pic = dec.pic_alloc()
        dec.pic_upload(pic, planes)
        out8, out10 = dec.pic_output(pic, 8, dra=luts), dec.pic_output(pic, 10, dra=luts)
        crop = (4, 2, 2, 6)
        out_crop = dec.pic_output(pic, 8, crop, dra=luts)
        plain = dec.pic_output(pic, 10)
    finally:
        dec.close()
    assert np.array_equal(out8, d[f"{name}_out8"]) and np.array_equal(out10, d[f"{name}_out10"])
    mapped = ol.dra_apply(planes, luts)
    assert np.array_equal(out_crop, ol.output_convert(mapped, 10, 8, crop))
    assert np.array_equal(plain, ol.output_convert(planes, 10, 10))      # and without tables the picture is untouched


@pytest.mark.gpu
def test_gpu_picture_md5():
    """xgpu_pic_md5 (k_md5.hip: the three chains of a picture in three lanes of one wave) == the reference's own xevd_md5_imgb on the same pictures
    (tests/golden/md5_pictures.json, made by tests/golden/make_md5_golden.py from oracle/_ref) - message lengths with and without a partial last block,
    8- and 10-bit pictures (two bytes per sample either way), a 1080p picture; and of the DRA-mapped picture == MD5 of what xgpu_pic_output hands out with
    the same tables (the copy the Main decoder signs, src_main/xevdm.c:3256-3287)."""
    import hashlib
    import json
    import sys
    from xevd_amd.decoder import XgpuDecoder
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_md5_golden as g
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "md5_pictures.json")))
    for seed, w, h, bd in g.CASES:
        planes = g.md5_picture(seed, w, h, bd)
        with XgpuDecoder(w, h, bd, device=0) as dec:
            pic = dec.pic_alloc()
            dec.pic_upload(pic, planes)
            got = [d.hex() for d in dec.pic_md5(pic)]
            again = [d.hex() for d in dec.pic_md5(pic)]
        assert got == gold[f"{seed}_{w}x{h}_{bd}b"] == again, (seed, w, h, bd, got)
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dra.npz"))
    planes, luts = [d[f"in_{c}"] for c in range(3)], d["three_ranges_idx58_luts"]
    hh, ww = planes[0].shape
    with XgpuDecoder(ww, hh, 10, device=0) as dec:
        pic = dec.pic_alloc()
        dec.pic_upload(pic, planes)
        mapped = dec.pic_output(pic, 10, dra=luts)
        got = dec.pic_md5(pic, dra=luts)
    n_y, n_c = ww * hh * 2, (ww // 2) * (hh // 2) * 2
    assert got == [hashlib.md5(mapped[:n_y].tobytes()).digest(), hashlib.md5(mapped[n_y:n_y + n_c].tobytes()).digest(), hashlib.md5(mapped[n_y + n_c:].tobytes()).digest()]


@pytest.mark.gpu
def test_gpu_stream_cropped_8bit_output():
    """a stream with SPS chroma QP tables and a conformance window: player output with the crop applied and 10 -> 8 bit conversion on
    the device == oracle pictures through the oracle's conversion"""
    import stream_util as su
    import oracle_lib as ol
    from xevd_amd.player import StreamDecoder
    d = np.load(os.path.join(golden_io.GOLDEN, "stream_cqt_crop_10b.npz"))
    data = d["bytes"].tobytes()
    frames = [f for _, f in StreamDecoder(data, apply_crop=True).output_order(output_bit_depth=8)]
    assert len(frames) == int(d["n"])
    for k, f in enumerate(frames):
        ref = [d[f"p{k}_{c}"] for c in range(3)]
        assert np.array_equal(f, ol.output_convert(ref, 10, 8, (2, 4, 0, 6))), f"picture {k}"


@pytest.mark.gpu
def test_gpu_decoder_survives_mutated_streams():
    """damaged streams that still parse reach the backend: every call returns (pictures or an error), the device stays usable and a
    clean stream decodes bit-exactly afterwards"""
    import glob
    import stream_util as su
    from xevd_amd.player import StreamDecoder
    rng = np.random.default_rng(7)
    paths = sorted(glob.glob(os.path.join(golden_io.GOLDEN, "stream_*.npz")))
    seeds = [np.load(p)["bytes"].tobytes() for p in paths]
    decoded = failed = 0
    for it in range(48):
        data = bytearray(seeds[it % len(seeds)])
        for _ in range(int(rng.integers(1, 3))):
            pos = int(rng.integers(len(data) // 4, len(data)))
            data[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            for _pic in StreamDecoder(bytes(data)).pictures(download=True):      # pictures in front of the damage still reach the backend
                decoded += 1
        except Exception:
            failed += 1
    assert decoded > 0 and failed > 0
    d = np.load(paths[0])
    ours = su.decode_gpu(d["bytes"].tobytes())
    for k in range(len(ours)):
        for c in range(3):
            assert np.array_equal(ours[k][c], d[f"p{k}_{c}"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hier_b_gop4", "main_eipd_all_tools_10b", "cqt_crop_10b", "idr_period_skip", "main_dra_10b", "main_htdf_all_tools_10b", "main_tiles_explicit_10b", "main_affine_all_tools_10b",
                                  "main_dmvr_hmvp_mmvd_b_8b", "main_every_tool_10b", "main_every_tool_tiles_8b", "main_slices_4rows_all_tools_10b", "main_slices_columns_arbitrary_8b"])
def test_gpu_plain_c_decoder(name, tmp_path):
    """examples/evc_decode - a decoder in plain C on the two C ABIs, no Python in the loop - writes the reference decoder's pictures"""
    import subprocess
    exe = os.path.normpath(os.path.join(golden_io.GOLDEN, "..", "..", "examples", "evc_decode"))
    assert os.path.exists(exe), "examples/evc_decode is not built (python -c 'import __graft_entry__ as g; g.build()')"
    d = np.load(os.path.join(golden_io.GOLDEN, f"stream_{name}.npz"))
    src, dst = tmp_path / "s.evc", tmp_path / "s.yuv"
    src.write_bytes(d["bytes"].tobytes())
    r = subprocess.run([exe, str(src), str(dst)], stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-300:]
    w, h = (int(v) for v in d["size"])
    ten_bit = int(d["p0_0"].max()) > 255 or "10b" in name
    got = np.fromfile(dst, "<u2" if ten_bit else np.uint8)
    expect = np.concatenate([d[f"p{k}_{c}"].ravel() for k in range(int(d["n"])) for c in range(3)])
    assert got.size == expect.size and np.array_equal(got.astype(np.int32), expect.astype(np.int32))


APP_ON_HIP = os.path.normpath(os.path.join(golden_io.GOLDEN, "..", "..", "oracle", "_ref", "xevd_app_on_hip"))


@pytest.mark.gpu
@pytest.mark.parametrize("name,args", [("hier_b_gop8_10b", ["--output-bit-depth", "10"]), ("main_eipd_all_tools_10b", ["--output-bit-depth", "10"]),
                                       ("main_all_tools_10b", []), ("signed_main_alf_10b", ["-s", "--output-bit-depth", "10"]),
                                       ("main_dra_10b", ["--output-bit-depth", "10"]), ("main_htdf_all_tools_10b", ["--output-bit-depth", "10"]),
                                       ("main_ibc_all_tools_10b", ["--output-bit-depth", "10"]), ("main_admvp_all_tools_10b", ["--output-bit-depth", "10"]), ("main_dmvr_all_tools_10b", ["--output-bit-depth", "10"]),
                                       ("main_tiles_3x2_all_tools_10b", ["--output-bit-depth", "10"]), ("main_tiles_explicit_10b", []),
                                       ("main_affine_all_tools_10b", ["--output-bit-depth", "10"]), ("main_every_tool_10b", ["--output-bit-depth", "10"]),
                                       ("main_slices_4rows_all_tools_10b", ["--output-bit-depth", "10"])])
def test_gpu_reference_application_on_our_api(name, args, tmp_path):
    """The reference's OWN sample application (app/xevd_app.c, compiled from its source where it lies) linked against libxevd_amd_api.so - this
    repository's implementation of the public xevd_create / xevd_decode / xevd_pull API - instead of libxevd: it decodes the golden streams on
    the GPU and writes the reference decoder's pictures (16-bit copy, or its own 10 -> 8 bit conversion), and verifies the MD5 SEIs with -s."""
    import subprocess
    import oracle_lib as ol
    if not os.path.exists(APP_ON_HIP):
        pytest.skip("oracle/_ref/xevd_app_on_hip is built only where the reference sources are (development container)")
    d = np.load(os.path.join(golden_io.GOLDEN, f"stream_{name}.npz"))
    src, dst = tmp_path / "s.evc", tmp_path / "s.yuv"
    src.write_bytes(d["bytes"].tobytes())
    r = subprocess.run([APP_ON_HIP, "-i", str(src), "-o", str(dst)] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0, r.stdout.decode()[-400:]
    pics = [[d[f"p{k}_{c}"] for c in range(3)] for k in range(int(d["n"]))]
    if "--output-bit-depth" in args:
        got = np.fromfile(dst, "<u2")
        expect = np.concatenate([p.ravel() for pic in pics for p in pic]).astype(np.uint16)
    else:
        got = np.fromfile(dst, np.uint8)
        expect = np.concatenate([ol.output_convert(pic, 10, 8) for pic in pics])
    assert got.size == expect.size and np.array_equal(got, expect)


REF_DECODE_HIP = os.path.normpath(os.path.join(golden_io.GOLDEN, "..", "..", "oracle", "_ref", "ref_decode_hip"))
# the Main-profile streams: the reference's Main library itself does not survive the Baseline ones (its entropy pass writes past ctx->cod_eco,
# src_main/xevdm.c:1450-1455, on CTU rows that cross the picture's bottom edge) - those are the Baseline library's, tests/test_stream.py
# Streams with tool_dmvr together with tool_hmvp / tool_mmvd - the refined vectors steer the reference parser's own candidate lists CU by CU - run too: the binding
# calls the refinement search on the host (xhost_dmvr_search) at every such CU and the backend repeats it for the prediction (round 4)
HOST_DMVR_STREAMS = set()
STREAM_NAMES = sorted(f[len("stream_"):-len(".npz")] for f in os.listdir(golden_io.GOLDEN)
                      if f.startswith("stream_") and f.endswith(".npz") and "main_" in f and f[len("stream_"):-len(".npz")] not in HOST_DMVR_STREAMS)


@pytest.mark.gpu
@pytest.mark.parametrize("name", STREAM_NAMES)
def test_gpu_reference_parser_feeds_hip_backend(name, tmp_path):
    """INTEGRATION.md section 4 as a program: the reference decoder itself (Main library objects; oracle/ref_binding.c compiles its xevdm.c with
    the backend installed in ctx->fn_dec_slice / fn_deblock / fn_alf / fn_picbuf_expand) - its NAL / parameter-set / slice-header / SBAC / CU
    parser, motion derivation, DPB and xevd_pull run unchanged, every picture is reconstructed and filtered by libxevd_hip.so - on every golden
    stream: the reference's pictures, sample for sample (and, where the stream carries MD5 SEIs, verified by the reference's own check)."""
    import subprocess
    if not os.path.exists(REF_DECODE_HIP):
        pytest.skip("oracle/_ref/ref_decode_hip is built only where the reference sources are (development container)")
    d = np.load(os.path.join(golden_io.GOLDEN, f"stream_{name}.npz"))
    src, dst = tmp_path / "s.evc", tmp_path / "s.raw"
    src.write_bytes(d["bytes"].tobytes())
    h, w = d["p0_0"].shape
    r = subprocess.run([REF_DECODE_HIP, str(src), str(dst), str(w), str(h), "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0, r.stdout.decode()[-400:]
    n = int(d["n"])
    assert int(r.stdout.split()[-2]) == n
    got = np.fromfile(dst, "<i2")
    expect = np.concatenate([d[f"p{k}_{c}"].ravel() for k in range(min(n, 64)) for c in range(3)]).astype(np.int16)
    assert got.size == expect.size and np.array_equal(got, expect)


REF_DECODE_HIP_BASE = os.path.normpath(os.path.join(golden_io.GOLDEN, "..", "..", "oracle", "_ref", "ref_decode_hip_base"))
BASE_STREAM_NAMES = sorted(f[len("stream_"):-len(".npz")] for f in os.listdir(golden_io.GOLDEN)
                           if f.startswith("stream_") and f.endswith(".npz") and "main_" not in f)


@pytest.mark.gpu
@pytest.mark.parametrize("name", BASE_STREAM_NAMES)
def test_gpu_reference_baseline_parser_feeds_hip_backend(name, tmp_path):
    """The same for the reference's BASELINE library (libxevdb: src_base/xevd.c; oracle/ref_binding_base.c compiles it with the backend in ctx->fn_dec_slice /
    fn_deblock / fn_picbuf_expand): its own Baseline parser, motion derivation, DPB and xevd_pull, every picture reconstructed by libxevd_hip.so - the golden
    Baseline streams, the reference's pictures sample for sample."""
    import subprocess
    if not os.path.exists(REF_DECODE_HIP_BASE):
        pytest.skip("oracle/_ref/ref_decode_hip_base is built only where the reference sources are (development container)")
    d = np.load(os.path.join(golden_io.GOLDEN, f"stream_{name}.npz"))
    src, dst = tmp_path / "s.evc", tmp_path / "s.raw"
    src.write_bytes(d["bytes"].tobytes())
    h, w = d["p0_0"].shape
    r = subprocess.run([REF_DECODE_HIP_BASE, str(src), str(dst), str(w), str(h), "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0, r.stdout.decode()[-400:]
    n = int(d["n"])
    assert int(r.stdout.split()[-2]) == n
    got = np.fromfile(dst, "<i2")
    expect = np.concatenate([d[f"p{k}_{c}"].ravel() for k in range(min(n, 64)) for c in range(3)]).astype(np.int16)
    assert got.size == expect.size and np.array_equal(got, expect)


@pytest.mark.gpu
@pytest.mark.parametrize("md5_on_device", [False, True], ids=["host_md5", "device_md5"])
def test_gpu_reference_application_rejects_bad_signature(md5_on_device, tmp_path):
    """the reference's application with -s on the public API: the intact stream passes, a flipped digest bit is XEVD_ERR_BAD_CRC - with the signatures made by the
    host's MD5, and (XEVD_AMD_MD5_ON_DEVICE=1) by xgpu_pic_md5 on the device picture"""
    import subprocess
    if not os.path.exists(APP_ON_HIP):
        pytest.skip("oracle/_ref/xevd_app_on_hip is not built")
    env = dict(os.environ)
    env.pop("XEVD_AMD_MD5_ON_DEVICE", None)
    if md5_on_device:
        env["XEVD_AMD_MD5_ON_DEVICE"] = "1"
    d = np.load(os.path.join(golden_io.GOLDEN, "stream_signed_main_alf_10b.npz"))
    good = tmp_path / "g.evc"
    good.write_bytes(d["bytes"].tobytes())
    r = subprocess.run([APP_ON_HIP, "-i", str(good), "-s"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120, env=env)
    assert r.returncode == 0 and b"MD5 check mismatch" not in r.stdout, r.stdout.decode()[-400:]
    bad = bytearray(d["bytes"].tobytes())
    bad[len(bad) - 3] ^= 0x40                          # inside the last SEI's V-plane digest
    src = tmp_path / "s.evc"
    src.write_bytes(bytes(bad))
    r = subprocess.run([APP_ON_HIP, "-i", str(src), "-s"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120, env=env)
    assert r.returncode != 0 and b"MD5 check mismatch" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfg2_base_1080p_8b_ippp", "cfg3_main_4k_10b_ra", "cfg4_main_8k_10b_ra",
                                  # the Main tools outside BASELINE's configurations, at 7680x4320 too: k_dmvr's item list (hundreds of thousands of sub-blocks), k_affine's
                                  # tile split, the data-flow kernel with most CUs of the picture as HTDF nodes
                                  "main_8k_10b_ra_dmvr", "main_8k_10b_ra_affine30", "main_8k_10b_ra_htdf"])
def test_gpu_bench_workload_vs_oracle(name):
    """The configurations the metric is quoted on, at size: the exact CU batch, reference pictures and ALF parameters bench.py times (Main 10 bit,
    admvp 8-tap tables, IQT, ADDB, ALF on every CTU, two lists, 50 % bi-prediction at 3840x2160 and 7680x4320; Baseline 1080p) through the whole
    pipeline against the oracle - XCD band / strip mapping, 32-bit arena offsets at 55 M coefficients, ALF at 120 x 68 CTUs included
    (src_main/xevdm.c:3136-3219)."""
    cs = cases.bench_case(name)
    ref, _, _, _ = cases.run_cpu("oracle", cs)
    out = cases.run_gpu(cs)
    for c in range(3):
        assert np.array_equal(out[c], ref.bufs[c]), f"{name} plane {c}: {np.argwhere(out[c] != ref.bufs[c])[:4]}"
    out = cases.run_gpu(cs, ahead=True)      # as bench.py's timed loop runs it: the residual pass inside the previous picture's data-flow launch
    for c in range(3):
        assert np.array_equal(out[c], ref.bufs[c]), f"{name} (residual pass ahead) plane {c}: {np.argwhere(out[c] != ref.bufs[c])[:4]}"


@pytest.mark.gpu
def test_gpu_plain_c_decoder_work_queue(tmp_path):
    """examples/evc_decode --gpus 2 on two inputs: every input is cut into closed GOPs, the GOPs of both go through the C work queue
    (include/xevd_wq.h) to one worker thread per device - on a one-GPU box the second worker finds no device and leaves its share to the
    first - and every GOP lands at its own offset of its output file: the reference decoder's pictures, byte for byte"""
    import subprocess
    exe = os.path.normpath(os.path.join(golden_io.GOLDEN, "..", "..", "examples", "evc_decode"))
    names = ["idr_period_skip", "hier_b_gop4"]
    cmd, expect = [exe, "--gpus", "2"], []
    for name in names:
        d = np.load(os.path.join(golden_io.GOLDEN, f"stream_{name}.npz"))
        src, dst = tmp_path / f"{name}.evc", tmp_path / f"{name}.yuv"
        src.write_bytes(d["bytes"].tobytes())
        cmd += [str(src), str(dst)]
        expect.append((dst, np.concatenate([d[f"p{k}_{c}"].ravel() for k in range(int(d["n"])) for c in range(3)])))
    r = subprocess.run(cmd, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    assert b"2 stream(s)" in r.stderr
    for dst, want in expect:
        got = np.fromfile(dst, np.uint8)
        assert got.size == want.size and np.array_equal(got.astype(np.int32), want.astype(np.int32)), dst


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["--workers", "1", "--tile-threads", "8"], ["--workers", "1", "--tile-threads", "8", "--no-pipeline"], ["--workers", "3", "--tile-threads", "2"],
                                  ["--workers", "1", "--tile-threads", "8", "--builders", "1"], ["--workers", "1", "--tile-threads", "4", "--builders", "4", "--build-threads", "2"],
                                  ["--workers", "2", "--tile-threads", "4", "--builders", "3", "--build-threads", "2"]],
                         ids=["pipelined", "back_to_back", "gop_parallel", "one_builder", "four_builders", "gop_parallel_three_builders"])
def test_gpu_bench_stream_plain_c_decoder(args, tmp_path):
    """the real-bitstream leg of bench.py at 1920x1088: random-access Main (hierarchical B, two lists of two references, tool_admvp, IQT, ADDB, ALF, 4x4 tiles),
    three IDR periods, decoded by examples/evc_decode - parser thread one picture ahead of the device thread, back to back, and GOP-parallel - must be the
    pictures parser + oracle reconstruct (which tests/test_stream.py pins to the reference decoder for this stream shape)"""
    import subprocess
    import sys
    import stream_util as su
    root = os.path.normpath(os.path.join(golden_io.GOLDEN, "..", ".."))
    sys.path.insert(0, root)
    import bench
    wl = dict(bench.WORKLOADS["cfg4_main_8k_10b_ra"])
    wl["w"], wl["h"] = 1920, 1088
    one, data, _ = bench.write_bench_stream(wl, 17, 3, seed=9)
    exe = os.path.join(root, "examples", "evc_decode")
    src, dst = tmp_path / "s.evc", tmp_path / "s.yuv"
    src.write_bytes(data)
    r = subprocess.run([exe] + args + [str(src), str(dst)], stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-300:]
    ref = su.decode_oracle(one)
    expect = np.concatenate([ref[k][c].ravel() for k in range(17) for c in range(3)]).astype(np.int32)
    got = np.fromfile(dst, "<u2").astype(np.int32)
    assert got.size == 3 * expect.size
    for period in range(3):
        g = got[period * expect.size:(period + 1) * expect.size]
        if not np.array_equal(g, expect):
            el = 1920 * 1088 * 3 // 2
            bad = [k for k in range(17) if not np.array_equal(g[k * el:(k + 1) * el], expect[k * el:(k + 1) * el])]
            k = bad[0]
            d = np.flatnonzero(g[k * el:k * el + 1920 * 1088] != expect[k * el:k * el + 1920 * 1088])
            assert False, f"IDR period {period}: pictures {bad} differ; first luma differences of picture {k} at (y, x) {[(int(i) // 1920, int(i) % 1920) for i in d[:6]]} ({len(d)} samples)"


@pytest.mark.gpu
def test_gpu_bench_streams_leg_configs4():
    """bench.py's multi-GPU leg (BASELINE configs[4] literally: independent Main streams -> examples/evc_decode --gpus N through the C work queue, compared with
    the reference decoder) on a small picture size and however many devices the box has: every stream's first IDR period bit-exact, every picture accounted for"""
    import bench
    wl = dict(bench.WORKLOADS["cfg3_main_4k_10b_ra"], w=512, h=256)
    out = bench.streams_leg(2, n_streams=3, gop_pictures=9, repeats=3, wl=wl)
    assert "error" not in out.get("evc_decode", {}), out
    assert out["pictures"] == 3 * 9 * 3 and sum(out["pictures_per_device"]) == out["pictures"]
    assert out["devices_used"] >= 1 and out["host_cpu_quota"] >= 1
    if os.path.exists(os.path.join(os.path.dirname(bench.__file__), "oracle", "_ref", "ref_decode_main")):
        assert out["bit_exact"] is True, out


def _permute_inside_ctus(batch, seed, log2_ctu=6):
    """the CUs of every CTU (quad-tree partitions) in the order of a tree walk that takes the two columns of ANY split node right to left at random - SUCO without the
    SPS limits on which nodes may choose (sizes 8 .. 64).  Still a tree order: all neighbours along one side of a CU lie in one sibling subtree, so they are all decoded
    before the CU or all after it, which the reference takes for granted (its deblocking walk filters a CU's whole side when the FIRST neighbouring SCU is done; a side
    that is half done would be filtered twice).  The order is what tells the backend which neighbours are reconstructed first."""
    b = dict(batch)
    rng = np.random.default_rng(seed)
    start = np.asarray(b["ctu_cu_start"])
    x, y, lw = np.asarray(b["x"]).astype(np.int64), np.asarray(b["y"]).astype(np.int64), np.asarray(b["log2w"]).astype(np.int64)
    perm = []

    def walk(idx, x0, y0, ls, out):
        if len(idx) == 0:
            return
        if len(idx) == 1 and lw[idx[0]] == ls:
            out.append(int(idx[0]))
            return
        h = 1 << (ls - 1)
        for q in ([1, 0, 3, 2] if rng.random() < 0.6 else [0, 1, 2, 3]):
            qx, qy = x0 + (q & 1) * h, y0 + (q >> 1) * h
            walk(idx[(x[idx] >= qx) & (x[idx] < qx + h) & (y[idx] >= qy) & (y[idx] < qy + h)], qx, qy, ls - 1, out)
    for k in range(len(start) - 1):
        idx = np.arange(start[k], start[k + 1])
        if len(idx) == 0:
            continue
        out = []
        walk(idx, int(x[idx].min()) >> log2_ctu << log2_ctu, int(y[idx].min()) >> log2_ctu << log2_ctu, log2_ctu, out)
        assert sorted(out) == list(idx)
        perm += out
    perm = np.asarray(perm, np.int64)
    n = len(b["x"])
    for k, v in list(b.items()):
        if isinstance(v, np.ndarray) and k != "ctu_cu_start" and k != "coef" and v.ndim >= 1 and v.shape[0] == n:
            b[k] = np.ascontiguousarray(v[perm])
    return b


@pytest.mark.parametrize("cfg", [
    ("perm_eipd_htdf_constrained", 200, 136, 10, 1, 1, (1, 1), 0.3, {"addb": 1, "alf": 1, "eipd": 1, "inter_frac": 0.5, "htdf_qp": 30, "constrained_intra": 1, "split_prob": 0.6, "coded_frac": 0.8}),
    ("perm_eipd_noaddb_i", 136, 136, 8, 1, 1, (1, 0), 0.0, {"eipd": 1, "inter_frac": 0.0, "split_prob": 0.7}),
    ("perm_eipd_noaddb_p", 264, 200, 10, 1, 0, (2, 0), 0.0, {"eipd": 1, "inter_frac": 0.6, "split_prob": 0.7}),
    ("perm_base_modes_b", 200, 136, 8, 0, 0, (1, 1), 0.4, {"inter_frac": 0.6, "split_prob": 0.7}),
], ids=lambda c: c[0])
def test_gpu_reversed_tree_orders_vs_oracle(cfg):
    """Right-hand neighbours in every shape: the CUs of each CTU in a tree order with reversed nodes of every size through the whole pipeline against the oracle (which sets its COD flags CU by CU like
    the reference) - EIPD prediction from left / right / both columns for every CU size, constrained intra prediction, HTDF borders, and the baseline deblocking filter's
    order-aware chroma chains, far beyond what SUCO split trees produce."""
    name, w, h, bd, admvp, iqt, n_refs, bi_frac, tools = cfg
    for seed in (1, 2):
        cs = cases.build_case(name, w, h, bd, admvp, iqt, n_refs, bi_frac, tools, seed=seed)
        cs["batch"] = _permute_inside_ctus(cs["batch"], 40 + seed)
        ref, _, _, _ = cases.run_cpu("oracle", cs)
        out = cases.run_gpu(cs)
        for c in range(3):
            assert np.array_equal(out[c], ref.bufs[c]), f"{name} seed {seed} plane {c}: {np.argwhere(out[c] != ref.bufs[c])[:4].tolist()}"
