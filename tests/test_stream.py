"""Frame-level parity on real bitstreams: synthetic EVC Baseline streams written by our writer are decoded by
the REAL reference decoder through its public API (xevd_create / xevd_decode / xevd_pull, oracle/_ref) and by our parser + the
CPU oracle; every output picture must be identical.  Also pins the committed golden streams (tests/golden/stream_*.npz)."""
import glob
import os

import numpy as np
import pytest

import golden_io
import stream_util as su
from xevd_amd import abi, stream

CONFIGS = [
    # w, h, pictures, kwargs
    (64, 64, 2, dict(split_prob=0.9)),
    (136, 72, 4, dict()),
    (208, 120, 6, dict(max_refs=2)),
    (144, 88, 5, dict(bit_depth=10, qp_offsets=(1, -2))),
    (72, 136, 7, dict(max_refs=4, skip_frac=0.4, idr_period=4)),
    (136, 136, 3, dict(deblock=False, cu_qp_delta=False)),
    (200, 72, 4, dict(inter_frac=0.5, split_prob=0.7)),
    # hierarchical sub-GOPs (temporal layers): B slices with bi-prediction, temporal direct mode, two-list skip; whole sub-GOPs only -
    # the reference never outputs the pictures of a sub-GOP whose lower POCs are missing
    (136, 72, 7, dict(log2_sub_gop=1, max_refs=2)),
    (208, 120, 9, dict(log2_sub_gop=2, max_refs=2)),
    (144, 136, 17, dict(log2_sub_gop=3, max_refs=3, bit_depth=10)),
    (136, 136, 9, dict(log2_sub_gop=2, max_refs=4, direct_frac=0.4, skip_frac=0.3)),
    # Main profile with the tools of the back half switched on - decoded by the reference's MAIN-profile library: frame-level parity of
    # IQT (incl. its chroma QP mapping default), ATS-intra / ATS-inter syntax and transforms, ADDB with slice offsets
    (136, 72, 3, dict(main=True)),
    (136, 72, 4, dict(main=True, iqt=True)),
    (136, 72, 4, dict(main=True, addb=True)),
    (208, 120, 5, dict(main=True, iqt=True, addb=True, addb_offsets=(2, -1), max_refs=2)),
    (136, 136, 5, dict(main=True, iqt=True, ats=True, addb=True)),
    (144, 136, 9, dict(main=True, iqt=True, ats=True, addb=True, log2_sub_gop=2, max_refs=2, bit_depth=10)),
    # ... and ALF: parameter sets in APS NAL units (5x5 / 7x7 luma shapes, merged classes, delta / prediction coding, chroma filter),
    # slice-level switches, per-CTU map or none, pictures without ALF in between
    (136, 72, 4, dict(main=True, alf=True)),
    (264, 136, 11, dict(main=True, alf=True, addb=True)),
    (144, 136, 9, dict(main=True, iqt=True, ats=True, addb=True, alf=True, log2_sub_gop=2, max_refs=2, bit_depth=10)),
    # chroma QP mapping tables in the SPS (one shared table from 16 up / two tables from the bottom of the range), with QP offsets; cropping
    (136, 72, 4, dict(chroma_qp_points=(1, [[(4, 0), (5, -2), (9, -5), (20, -3)]]), qp_offsets=(2, -3))),
    (144, 88, 5, dict(bit_depth=10, chroma_qp_points=(0, [[(30, 0), (10, -4), (15, -6)], [(25, 1), (6, -2), (30, -10)]]), max_refs=2, crop=(2, 4, 0, 6))),
    (136, 72, 4, dict(main=True, iqt=True, addb=True, chroma_qp_points=(1, [[(10, -1), (12, -6)]]), qp_offsets=(-2, 1))),
    # DRA: parameter sets in APS NAL units of type 1, switched on by the PPS; the reference applies the post-filter when it outputs a picture
    (136, 72, 4, dict(main=True, iqt=True, bit_depth=10, dra="three_ranges_idx58", max_refs=2)),
    (144, 88, 5, dict(main=True, bit_depth=10, dra="five_ranges_idx40", addb=True, log2_sub_gop=2, max_refs=2)),
    (136, 72, 3, dict(main=True, iqt=True, bit_depth=10, dra="one_range_idx30", chroma_qp_points=(1, [[(10, -1), (12, -6)]]), qp_offsets=(-2, 1))),
    # ... and EIPD: 33 luma / 5 chroma intra modes with their most-probable-mode syntax, all-intra and mixed pictures
    (136, 72, 2, dict(main=True, eipd=True, idr_period=1, split_prob=0.8)),
    (200, 136, 5, dict(main=True, eipd=True, inter_frac=0.4, max_refs=2)),
    (144, 136, 9, dict(main=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, inter_frac=0.5, log2_sub_gop=2, max_refs=2, bit_depth=10)),
    # tool_htdf: every intra CU and every coded inter CU is filtered after its reconstruction - the real decoder against parser + oracle
    (136, 72, 3, dict(main=True, htdf=True, inter_frac=0.6)),
    (200, 136, 9, dict(main=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, inter_frac=0.6, log2_sub_gop=2, max_refs=2, bit_depth=10)),
    # sps->tool_admvp: merge candidates (spatial, temporal with POC scaling and the picture-border clip, combined bi-predictive, zero) for skip and
    # merge-mode CUs, the resolution-indexed predictor + bi_idx for explicitly coded motion, intra-only 4x4 CUs, Main interpolation tables
    (136, 72, 3, dict(main=True, admvp=True, inter_frac=0.8, max_refs=1)),
    (136, 136, 8, dict(main=True, admvp=True, inter_frac=0.9, max_refs=4)),
    (200, 136, 9, dict(main=True, admvp=True, inter_frac=0.8, max_refs=2, log2_sub_gop=2)),
    (264, 136, 17, dict(main=True, admvp=True, inter_frac=0.9, max_refs=3, log2_sub_gop=3, bit_depth=10, direct_frac=0.3, skip_frac=0.3)),
    (200, 136, 9, dict(main=True, admvp=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, ibc_log_max=5, inter_frac=0.7, max_refs=2, log2_sub_gop=2, bit_depth=10)),
    # ... with sps->tool_hmvp (history candidates in the merge list and the fallback predictor, reset per CTU row) and sps->tool_amvr (mvr_idx: coarser
    # vector grids, the predictor's neighbour position coupled with the index)
    (200, 136, 9, dict(main=True, admvp=True, hmvp=True, inter_frac=0.9, max_refs=2, log2_sub_gop=2)),
    (200, 136, 4, dict(main=True, admvp=True, amvr=True, inter_frac=0.9, max_refs=2)),
    (264, 136, 9, dict(main=True, admvp=True, amvr=True, hmvp=True, iqt=True, addb=True, alf=True, inter_frac=0.9, max_refs=3, log2_sub_gop=3, bit_depth=10)),
    # ... and sps->tool_dmvr: skip / merge-mode CUs refined by the backend; the refined vectors come back to the parser (xhost_parser_set_dmvr_mvs)
    # for the temporal candidates of later pictures, the baseline deblocking filter sees the refined vectors, ADDB the unrefined ones
    (200, 136, 9, dict(main=True, admvp=True, dmvr=True, inter_frac=0.9, max_refs=2, log2_sub_gop=2, skip_frac=0.3, direct_frac=0.3)),
    (136, 136, 6, dict(main=True, admvp=True, dmvr=True, inter_frac=0.9, max_refs=2, skip_frac=0.3, direct_frac=0.3)),
    (264, 136, 17, dict(main=True, admvp=True, dmvr=True, amvr=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, ibc_log_max=4, inter_frac=0.9, max_refs=3,
                        log2_sub_gop=3, bit_depth=10, skip_frac=0.3, direct_frac=0.3)),
    # ... and sps->tool_mmvd: merge with vector difference (group / base candidate / distance / direction syntax, the three prediction types per
    # group with mirrored and POC-scaled vectors, the P-slice variants for 1, 2 and more references)
    (200, 136, 4, dict(main=True, admvp=True, mmvd=True, inter_frac=0.9, skip_frac=0.35, direct_frac=0.3, max_refs=1)),
    (136, 136, 8, dict(main=True, admvp=True, mmvd=True, inter_frac=0.9, skip_frac=0.35, direct_frac=0.3, max_refs=4)),
    (264, 136, 17, dict(main=True, admvp=True, mmvd=True, amvr=True, hmvp=True, iqt=True, addb=True, alf=True, inter_frac=0.9, skip_frac=0.35, direct_frac=0.3, max_refs=3,
                        log2_sub_gop=3, bit_depth=10)),
    # ALF parameter sets that start from the standard's fixed filters (usage pattern 1: every class, 2: flagged classes; 4-bit set index per class)
    (264, 136, 8, dict(main=True, alf=True, addb=True, alf_fixed=True)),
    (200, 136, 9, dict(main=True, iqt=True, ats=True, addb=True, alf=True, alf_fixed=True, log2_sub_gop=2, max_refs=2, bit_depth=10)),
    # sps->ibc_flag: ibc_flag + block-vector syntax in I / P / B slices (size limits 8 .. 64), the above-right neighbour quirk of the Main library's
    # predictor availability included (an IBC CU above-right lends its block vector as a motion vector predictor, xevdm_util.c:1499-1503)
    (136, 72, 3, dict(main=True, eipd=True, ibc_log_max=4, idr_period=1)),
    (264, 136, 6, dict(main=True, eipd=True, addb=True, ibc_log_max=3, ibc_frac=0.6, inter_frac=0.5)),
    (200, 136, 9, dict(main=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, ibc_log_max=6, inter_frac=0.5, log2_sub_gop=3, max_refs=2, bit_depth=10)),
    # sps->tool_affine: affine merge CUs (model inherited from affine neighbours, constructed from corner vectors incl. the co-located ones, zero
    # candidates) and affine inter CUs (two predictors per list, 2 or 3 coded control points), sub-block vectors in the motion maps (merge / predictor
    # candidates of later CUs, temporal candidates of later pictures, history), with every other tool, tiles and DMVR
    (200, 136, 3, dict(main=True, admvp=True, affine=True, inter_frac=0.9, split_prob=0.3)),
    (392, 264, 6, dict(main=True, admvp=True, affine=True, inter_frac=0.95, split_prob=0.35, skip_frac=0.3, direct_frac=0.3, max_refs=2)),
    (392, 264, 9, dict(main=True, admvp=True, affine=True, affine_frac=0.7, inter_frac=0.95, split_prob=0.35, skip_frac=0.3, direct_frac=0.3, max_refs=2, log2_sub_gop=2)),
    (328, 264, 17, dict(main=True, admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, inter_frac=0.9, split_prob=0.4,
                        skip_frac=0.3, direct_frac=0.3, max_refs=3, log2_sub_gop=3, bit_depth=10)),
    (392, 264, 9, dict(main=True, admvp=True, affine=True, dmvr=True, addb=True, inter_frac=0.95, split_prob=0.35, skip_frac=0.3, direct_frac=0.3, max_refs=2, log2_sub_gop=2, tiles=(2, 2, 0))),
    (264, 264, 8, dict(main=True, admvp=True, affine=True, affine_frac=0.9, inter_frac=1.0, split_prob=0.25, skip_frac=0.3, direct_frac=0.3, max_refs=4)),
    # sps->tool_rpl / tool_pocs: POC from poc_lsb, reference lists and marking from RPLs (explicit in the slice header or candidates of the SPS by index),
    # list sizes from the override - low delay, hierarchical B, IDR periods, with the motion tools that read the lists (TMVP, merge, DMVR) and tiles
    (136, 72, 5, dict(main=True, rpl=True, max_refs=2)),
    (136, 72, 5, dict(main=True, pocs=True, max_refs=2)),
    (136, 136, 11, dict(main=True, rpl=True, pocs=True, max_refs=4, idr_period=4, skip_frac=0.4)),
    (200, 136, 17, dict(main=True, rpl=True, max_refs=3, log2_sub_gop=3)),
    (200, 136, 10, dict(main=True, pocs=True, admvp=True, max_refs=2, log2_sub_gop=2, idr_period=5)),
    (200, 136, 12, dict(main=True, rpl=True, pocs=True, admvp=True, hmvp=True, max_refs=4, rpl_in_sps=True, idr_period=7)),
    (264, 136, 17, dict(main=True, rpl=True, pocs=True, admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, inter_frac=0.9,
                        skip_frac=0.3, direct_frac=0.3, max_refs=3, log2_sub_gop=3, bit_depth=10, qp_delta_area=8)),
    (392, 264, 9, dict(main=True, rpl=True, admvp=True, dmvr=True, addb=True, inter_frac=0.95, skip_frac=0.3, direct_frac=0.3, max_refs=2, log2_sub_gop=2, tiles=(2, 2, 0))),
    # sps->tool_cm_init: contexts initialised from the standard's tables (slice kind, QP), neighbour / level / shape dependent context choices
    (136, 72, 2, dict(main=True, cm_init=True, idr_period=1)),
    (136, 72, 4, dict(main=True, cm_init=True, max_refs=2)),
    (200, 136, 9, dict(main=True, cm_init=True, max_refs=2, log2_sub_gop=2)),
    (264, 136, 17, dict(main=True, cm_init=True, rpl=True, pocs=True, admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True,
                        ibc_log_max=5, inter_frac=0.9, skip_frac=0.3, direct_frac=0.3, max_refs=3, log2_sub_gop=3, bit_depth=10, qp_delta_area=8)),
    (392, 264, 9, dict(main=True, cm_init=True, admvp=True, dmvr=True, addb=True, inter_frac=0.95, skip_frac=0.3, direct_frac=0.3, max_refs=2, log2_sub_gop=2, tiles=(2, 2, 0))),
    # sps->tool_adcc: coefficient blocks as last position + significance / greater-than-1 / greater-than-2 flags + Golomb-Rice remainders + signs per group
    # of 16 scan positions, contexts from the coded neighbours; all block sizes 2x2 .. 64x64, small and large levels (low QPs)
    (64, 64, 1, dict(main=True, adcc=True, idr_period=1, split_prob=0.9)),
    (136, 72, 4, dict(main=True, adcc=True, max_refs=2)),
    (200, 136, 5, dict(main=True, adcc=True, iqt=True, max_refs=2, max_level=3000, qp_range=(0, 8))),
    (200, 136, 9, dict(main=True, adcc=True, iqt=True, ats=True, max_refs=2, log2_sub_gop=2)),
    (264, 136, 17, dict(main=True, adcc=True, rpl=True, pocs=True, admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True,
                        ibc_log_max=5, inter_frac=0.9, skip_frac=0.3, direct_frac=0.3, max_refs=3, log2_sub_gop=3, bit_depth=10, qp_delta_area=8)),
    # sps_btt_flag: binary / ternary split trees (no quad split), non-square CUs 4x8 .. 64x16 through every CU-level derivation, the SPS limits of the tree,
    # implicit splits at the picture border, split-flag contexts from the neighbours' sizes (cm_init), "inter only" mode constraints (admvp), QP groups of TT nodes
    (64, 64, 1, dict(main=True, btt=(2, 0, 0, 0), idr_period=1, split_prob=0.6)),
    (136, 72, 4, dict(main=True, btt=(2, 0, 0, 0), max_refs=2, split_prob=0.7)),
    (264, 200, 4, dict(main=True, btt=(3, 1, 1, 1), max_refs=2, split_prob=0.8)),
    (200, 136, 9, dict(main=True, btt=(2, 0, 0, 0), max_refs=2, log2_sub_gop=2, split_prob=0.7)),
    (200, 136, 5, dict(main=True, btt=(2, 0, 1, 0), adcc=True, max_refs=2, split_prob=0.75)),
    (200, 136, 5, dict(main=True, btt=(2, 0, 0, 0), eipd=True, htdf=True, qp_delta_area=8, max_refs=2, split_prob=0.7)),
    (200, 136, 5, dict(main=True, btt=(2, 0, 0, 0), admvp=True, max_refs=2, split_prob=0.7)),
    (264, 200, 9, dict(main=True, btt=(2, 0, 0, 0), admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, cm_init=True, adcc=True,
                       rpl=True, pocs=True, qp_delta_area=8, max_refs=2, log2_sub_gop=2, split_prob=0.7, bit_depth=10, inter_frac=0.9, skip_frac=0.3, direct_frac=0.3)),
    (392, 264, 5, dict(main=True, btt=(2, 0, 0, 0), iqt=True, addb=True, tiles=(2, 2, 0), max_refs=2, split_prob=0.7)),
    # sps->dquant_flag: one QP delta per quantisation group of pps.cu_qp_delta_area samples (8x8 ... 64x64; an odd area never matches a square node)
    (264, 200, 5, dict(main=True, iqt=True, qp_delta_area=6, split_prob=0.7, inter_frac=0.7)),
    (264, 200, 5, dict(main=True, iqt=True, qp_delta_area=10, split_prob=0.7, inter_frac=0.7)),
    (264, 200, 5, dict(main=True, iqt=True, qp_delta_area=7, split_prob=0.7, inter_frac=0.7)),
    (328, 264, 9, dict(main=True, admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, inter_frac=0.9, split_prob=0.6,
                       skip_frac=0.3, direct_frac=0.3, max_refs=3, log2_sub_gop=2, bit_depth=10, qp_delta_area=8)),
    (392, 264, 5, dict(main=True, iqt=True, addb=True, tiles=(2, 2, 0), qp_delta_area=12, split_prob=0.7)),
    # several tiles per picture (PPS grid uniform / explicit, one slice with entry points): every tile its own arithmetic-coder run, no neighbour
    # across a tile border (intra samples, HTDF border, motion candidates, most probable modes), the history reset per tile CTU row, deblocking with
    # and without loop_filter_across_tiles, the ALF windows ending at the tile (mirrored / replicated)
    (256, 192, 3, dict(main=True, iqt=True, tiles=(2, 2, 1))),
    (256, 256, 4, dict(main=True, tiles=(4, 4, 0))),
    (320, 200, 4, dict(main=True, iqt=True, addb=True, alf=True, eipd=True, bit_depth=10, tiles=(3, 2, 0))),
    (392, 264, 9, dict(main=True, iqt=True, ats=True, addb=True, alf=True, alf_fixed=True, eipd=True, htdf=True, admvp=True, amvr=True, hmvp=True, mmvd=True, log2_sub_gop=2, max_refs=2,
                       tiles=(2, 3, 1))),
    (384, 256, 9, dict(main=True, iqt=True, addb=True, alf=True, eipd=True, admvp=True, dmvr=True, log2_sub_gop=2, max_refs=2, bit_depth=10, tiles=(2, 2, 0))),
    (512, 320, 5, dict(main=True, iqt=True, addb=True, alf=True, eipd=True, admvp=True, bit_depth=10, tiles=(3, 3, 0, (1, 5), (2, 1)))),
    (712, 72, 4, dict(main=True, iqt=True, alf=True, addb=True, tiles=(7, 1, 0))),
    # several slices per picture (slice NAL units of tile rectangles, sps_pocs_flag): own slice QP per slice (the tiles' QP predictors start there), the in-loop
    # filters of the whole picture with the LAST slice's header - deblocking switched off there switches it off everywhere (src_main/xevdm.c:3138-3199)
    (256, 256, 4, dict(main=True, pocs=True, tiles=(4, 4, 0), slices=[(0, 7), (8, 15)])),
    (256, 256, 5, dict(main=True, pocs=True, rpl=True, iqt=True, addb=True, alf=True, admvp=True, hmvp=True, max_refs=2, log2_sub_gop=2, bit_depth=10, tiles=(4, 4, 0),
                       slices=[(0, 3, 28, 1), (4, 7, 33, 0), (8, 11, -1, 1), (12, 15, 35, 1)])),
    (256, 192, 3, dict(main=True, pocs=True, iqt=True, addb=True, tiles=(2, 2, 1), slices=[(0, 1, 30, 1), (2, 2, 26, 1), (3, 3, 38, 0)])),
    (384, 256, 4, dict(main=True, pocs=True, rpl=True, iqt=True, addb=True, alf=True, eipd=True, admvp=True, max_refs=2, tiles=(3, 2, 0), slices=[(0, 3, 31), (1, 5, 25)])),
    (384, 256, 3, dict(main=True, pocs=True, iqt=True, addb=True, admvp=True, max_refs=2, tiles=(3, 2, 0), slices=[(0, 3, 31), (1, 5, 25)], arbitrary_slices=True)),
    (256, 256, 3, dict(main=True, pocs=True, tiles=(4, 4, 0), slices=[(0, 7), (8, 11, 35), (12, 15, 24)], arbitrary_slices=True)),
    (200, 328, 4, dict(main=True, iqt=True, alf=True, tiles=(1, 5, 1))),
    # tool_dmvr TOGETHER with tool_hmvp / tool_mmvd (what Main-profile encoders switch on): the refined vectors are state of the picture's own parse (history
    # buffer, merge list of MMVD CUs), so the front end runs the refinement search itself on the decoded reference samples (xevd_amd/host/dmvr_search.h;
    # decode_oracle registers them with xhost_parser_set_ref_luma, the stream writer gets them through xhost_writer_set_ref_luma)
    (136, 136, 9, dict(main=True, admvp=True, dmvr=True, hmvp=True, iqt=True, addb=True, log2_sub_gop=2, max_refs=2)),
    (264, 200, 17, dict(main=True, admvp=True, dmvr=True, mmvd=True, iqt=True, addb=True, log2_sub_gop=3, max_refs=2, bit_depth=10)),
    (200, 136, 9, dict(main=True, admvp=True, dmvr=True, hmvp=True, mmvd=True, amvr=True, log2_sub_gop=2, max_refs=4, skip_frac=0.3, direct_frac=0.4)),
    (200, 136, 8, dict(main=True, admvp=True, dmvr=True, hmvp=True, mmvd=True, inter_frac=0.9, max_refs=4, skip_frac=0.3, direct_frac=0.3)),
    (392, 264, 9, dict(main=True, admvp=True, amvr=True, hmvp=True, mmvd=True, dmvr=True, affine=True, iqt=True, addb=True, alf=True, inter_frac=0.95, skip_frac=0.3, direct_frac=0.3,
                       max_refs=2, log2_sub_gop=2, tiles=(2, 2, 0))),
    # sps_suco_flag: split nodes coded right to left (the flag's syntax and inheritance, quad / binary / ternary order) - right-hand neighbours in the flag and split contexts,
    # the most probable intra modes (a third neighbour mode), merge / AMVP / affine candidates and their temporal positions, EIPD prediction from a right reference column
    # (LR_01 / LR_11 forms of DC, horizontal, planar, bilinear, angular), HTDF borders; with quad trees (many reversed nodes) and BTT
    (136, 72, 2, dict(main=True, suco=(0, 2), idr_period=1, split_prob=0.8)),
    (136, 72, 2, dict(main=True, suco=(0, 2), eipd=True, idr_period=1, split_prob=0.8)),
    (200, 136, 5, dict(main=True, suco=(0, 2), eipd=True, inter_frac=0.4, max_refs=2)),
    (200, 136, 9, dict(main=True, suco=(0, 2), cm_init=True, max_refs=2, log2_sub_gop=2)),
    (136, 72, 3, dict(main=True, suco=(0, 2), htdf=True, inter_frac=0.6)),
    (200, 136, 9, dict(main=True, suco=(0, 2), admvp=True, hmvp=True, amvr=True, inter_frac=0.9, max_refs=2, log2_sub_gop=2)),
    (200, 136, 4, dict(main=True, suco=(0, 2), admvp=True, mmvd=True, inter_frac=0.9, skip_frac=0.35, direct_frac=0.3, max_refs=1)),
    (392, 264, 6, dict(main=True, suco=(0, 2), admvp=True, affine=True, inter_frac=0.95, split_prob=0.35, skip_frac=0.3, direct_frac=0.3, max_refs=2)),
    (264, 200, 4, dict(main=True, suco=(1, 1), btt=(3, 1, 1, 1), max_refs=2, split_prob=0.8)),
    (264, 136, 6, dict(main=True, suco=(0, 2), eipd=True, addb=True, ibc_log_max=3, ibc_frac=0.6, inter_frac=0.5)),
    (264, 200, 9, dict(main=True, suco=(0, 2), btt=(2, 0, 0, 0), admvp=True, dual_tree=True, affine=True, amvr=True, hmvp=True, mmvd=True, dmvr=True, iqt=True, ats=True, addb=True, alf=True, eipd=True,
                       htdf=True, ibc_log_max=4, cm_init=True, adcc=True, qp_delta_area=8, max_refs=2, log2_sub_gop=2, split_prob=0.7, bit_depth=10, inter_frac=0.7, skip_frac=0.3, direct_frac=0.3)),
    (392, 264, 9, dict(main=True, suco=(0, 3), cm_init=True, admvp=True, dmvr=True, addb=True, inter_frac=0.95, skip_frac=0.3, direct_frac=0.3, max_refs=2, log2_sub_gop=2, tiles=(2, 2, 0))),
    # every Main tool the front end knows at once - the shape of a real Main-profile encode
    (264, 200, 17, dict(main=True, btt=(2, 0, 0, 0), admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, dmvr=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True,
                        cm_init=True, adcc=True, rpl=True, pocs=True, qp_delta_area=8, max_refs=2, log2_sub_gop=3, split_prob=0.7, bit_depth=10, inter_frac=0.9, skip_frac=0.3, direct_frac=0.3)),
    (264, 200, 9, dict(main=True, btt=(2, 0, 0, 0), admvp=True, dual_tree=True, affine=True, amvr=True, hmvp=True, mmvd=True, dmvr=True, iqt=True, ats=True, addb=True, alf=True, eipd=True,
                       htdf=True, ibc_log_max=4, cm_init=True, adcc=True, qp_delta_area=8, max_refs=2, log2_sub_gop=2, split_prob=0.7, bit_depth=10, inter_frac=0.7, skip_frac=0.3, direct_frac=0.3)),
]


def _right_first(b, w, h):
    """number of CUs of a parsed batch whose right-hand neighbour SCU belongs to a CU that comes before them in decoding order"""
    ws, hs = (w + 3) // 4, (h + 3) // 4
    own = -np.ones((hs, ws), np.int64)
    n = 0
    for i, (x, y, lw, lh) in enumerate(zip(b["x"], b["y"], b["log2w"], b["log2h"])):
        xs, ys, sw, sh = int(x) // 4, int(y) // 4, (1 << int(lw)) // 4, (1 << int(lh)) // 4
        n += int(xs + sw < ws and own[ys, xs + sw] >= 0)
        own[ys:ys + sh, xs:xs + sw] = i
    return n


@pytest.mark.ref
@pytest.mark.parametrize("cfg", CONFIGS, ids=[f"{c[0]}x{c[1]}x{c[2]}" for c in CONFIGS])
def test_stream_reference_decoder_equals_parser_plus_oracle(cfg):
    if not su.have_ref_decoder():
        pytest.skip("oracle/_ref is not built")
    w, h, n, kw = cfg
    data = su.make_stream(w, h, n, seed=w * 7 + n, **kw)
    ref = su.decode_reference(data, w, h, main=bool(kw.get("main")))
    pics = []
    ours = su.decode_oracle(data, keep_params=pics)
    assert len(ref) == n and len(ours) == n
    if kw.get("dmvr") and (kw.get("hmvp") or kw.get("mmvd")) and kw.get("log2_sub_gop"):      # refinement candidates exist: merge-mode CUs with two references, 8x8 and up
        assert sum(int(((p["batch"]["dmvr"] > 0) & (p["batch"]["refi"].min(1) >= 0) & (p["batch"]["log2w"] >= 3) & (p["batch"]["log2h"] >= 3)).sum()) for p in pics if p["batch"]["dmvr"] is not None) >= 10
    if kw.get("ats"):      # the stream really carries both kinds of ATS CUs
        assert sum(int((p["batch"]["ats"] & 1).sum()) for p in pics) > 0 and sum(int((p["batch"]["ats_inter"] != 0).sum()) for p in pics) > 0
    if kw.get("suco"):     # CUs with their right-hand neighbour decoded first exist
        assert sum(_right_first(p["batch"], w, h) for p in pics) >= 20
    if kw.get("eipd"):     # angular luma modes and all five chroma modes occur
        intra = np.concatenate([p["batch"]["ipm"][p["batch"]["pred_mode"] == 0] for p in pics])
        assert len(set(intra[:, 0].tolist())) > 20 and set(intra[:, 1].tolist()) == {0, 1, 2, 3, 4}
    for k in range(n):
        for c in range(3):
            assert np.array_equal(ref[k][c], ours[k][c]), f"picture {k} plane {c}: {np.argwhere(ref[k][c] != ours[k][c])[:4]}"


@pytest.mark.ref
def test_stream_reference_decoder_threads_agree():
    if not su.have_ref_decoder():
        pytest.skip("oracle/_ref is not built")
    # (the reference's row threading needs at least as many CTU rows as threads: with more threads than rows it leaves CTU rows
    #  unreconstructed - a finding, not a requirement - so the multi-threaded CPU baseline is only used on tall enough pictures)
    data = su.make_stream(136, 328, 4, seed=5)
    a = su.decode_reference(data, 136, 328, threads=1)
    b = su.decode_reference(data, 136, 328, threads=4)
    assert all(np.array_equal(a[k][c], b[k][c]) for k in range(4) for c in range(3))


def test_tiles_parse_the_same_on_any_number_of_threads():
    """xhost_parser_set_threads: the tiles of a picture are parsed by parallel host threads; the batch (CU order tile by tile, coefficient offsets,
    CTU starts, the grid) does not depend on the thread count"""
    data = su.make_stream(392, 264, 5, seed=6, main=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, admvp=True, amvr=True, hmvp=True, mmvd=True, log2_sub_gop=2, max_refs=2,
                          tiles=(3, 3, 1))
    one, many = stream.parse_stream(data), stream.parse_stream(data, threads=5)
    assert len(one) == len(many) == 5 and one[0]["batch"]["tiles"]["col_bd"] == [0, 2, 4, 7]
    for p, q in zip(one, many):
        assert p["poc"] == q["poc"]
        for k, v in p["batch"].items():
            assert np.array_equal(v, q["batch"][k]) if isinstance(v, np.ndarray) else v == q["batch"][k], k
        # tile by tile: the CTU starts rise through the whole batch and every CU lies in the tile its CTU belongs to
        assert (np.diff(p["batch"]["ctu_cu_start"].astype(np.int64)) >= 0).all()


def test_writer_is_deterministic_and_parser_round_trips():
    data = su.make_stream(136, 72, 3, seed=11)
    assert data == su.make_stream(136, 72, 3, seed=11)
    pics = stream.parse_stream(data)
    assert [p["poc"] for p in pics] == [0, 1, 2] and pics[0]["is_idr"] and pics[1]["refs"][0] == [0]
    # every CU of every picture lies inside the picture and the CTU index is consistent
    for p in pics:
        b = p["batch"]
        assert (b["x"].astype(int) + (1 << b["log2w"].astype(int)) <= 136).all() and (b["y"].astype(int) + (1 << b["log2h"].astype(int)) <= 72).all()
        assert b["ctu_cu_start"][-1] == len(b["x"])
        area = ((1 << b["log2w"].astype(np.int64)) * (1 << b["log2h"].astype(np.int64))).sum()
        assert area == 136 * 72


def test_parser_rejects_garbage():
    data = bytearray(su.make_stream(64, 64, 2, seed=3))
    with pytest.raises(RuntimeError):
        stream.parse_stream(bytes(data[:-3]))             # truncated last NAL
    data[4] ^= 0x80                                        # forbidden_zero_bit of the first NAL header
    with pytest.raises(RuntimeError):
        stream.parse_stream(bytes(data))


def _nals(data):
    pos, out = 0, []
    while pos + 4 <= len(data):
        n = int.from_bytes(data[pos:pos + 4], "big")
        out.append(data[pos:pos + 4 + n])
        pos += 4 + n
    return out


def test_several_slices_per_picture_boundaries_and_refusals():
    """a picture in several slice NAL units: the NAL scanner (what the GOP splitter uses) tells first slices from further ones; a picture that loses one of
    its slices is refused when the next picture starts (not assembled from another picture's tiles); the writer refuses such pictures without sps_pocs_flag
    (the reference decoder would count a picture per slice NAL, src_main/xevdm.c:3030-3040)"""
    import ctypes as C
    kw = dict(main=True, pocs=True, tiles=(4, 4, 0), slices=[(0, 7), (8, 11, 35), (12, 15, 24)])
    data = su.make_stream(256, 256, 3, seed=11, **kw)
    lib = stream.load()
    lib.xhost_scan_open.restype = C.c_void_p
    lib.xhost_scan_nal.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.xhost_scan_close.argtypes = [C.c_void_p]
    sc = lib.xhost_scan_open()
    kinds = [lib.xhost_scan_nal(sc, bytes(n[4:]), len(n) - 4) for n in _nals(data)]
    lib.xhost_scan_close(sc)
    assert [k for k in kinds if k] == [1, 2, 2] * 3
    assert len(stream.parse_stream(data)) == 3
    nals = _nals(data)
    slices = [i for i, k in enumerate(kinds) if k]
    with pytest.raises(RuntimeError, match="another picture|twice"):      # picture 1 without its last slice: picture 2's first slice must not complete it
        stream.parse_stream(b"".join(n for i, n in enumerate(nals) if i != slices[5]))
    # a parameter set between two slices of one picture would change the geometry under the open picture's maps / tile state: refused, the
    # partial picture dropped; so is a stream that ends inside a picture
    is_ps = [i for i, n in enumerate(nals) if ((n[4] << 8 | n[5]) >> 9 & 63) - 1 in (24, 25)]
    assert len(is_ps) >= 2
    for ps in is_ps[:2]:
        with pytest.raises(RuntimeError, match="parameter set between"):
            stream.parse_stream(b"".join(nals[:slices[1]] + [nals[ps]] + nals[slices[1]:]))
    with pytest.raises(RuntimeError, match="ends inside a picture"):
        stream.parse_stream(b"".join(nals[:slices[7] + 1]))
    # the scanner resynchronises behind a lost slice: the next picture's first slice brings tiles the open picture already has -> kind 1 again
    sc = lib.xhost_scan_open()
    kinds2 = [lib.xhost_scan_nal(sc, bytes(n[4:]), len(n) - 4) for i, n in enumerate(nals) if i != slices[5]]
    lib.xhost_scan_close(sc)
    assert [k for k in kinds2 if k] == [1, 2, 2, 1, 2, 1, 2, 2]
    with pytest.raises(RuntimeError):
        su.make_stream(256, 256, 2, seed=11, main=True, tiles=(4, 4, 0), slices=[(0, 7), (8, 15)])      # no sps_pocs_flag
    with pytest.raises(RuntimeError):
        su.make_stream(256, 256, 2, seed=11, main=True, pocs=True, tiles=(4, 4, 0), slices=[(0, 7), (4, 15)])      # overlapping tile rectangles


def test_ref_luma_registered_late_from_another_thread():
    """xhost_parser_set_ref_luma_wait: the luma planes of a tool_dmvr + tool_hmvp / tool_mmvd stream are registered from a second thread, each one only after
    the parser has been inside later pictures for a while (or, for a picture nothing refers to soon, pictures later) - every picture parses to what the
    picture-by-picture hand-over gives; a cancelled wait fails the parser instead of hanging it"""
    import queue
    import threading
    import time
    w, h, n = 264, 200, 17
    data = su.make_stream(w, h, n, seed=21, main=True, admvp=True, dmvr=True, hmvp=True, mmvd=True, amvr=True, iqt=True, addb=True, log2_sub_gop=3, max_refs=2, tiles=(2, 2, 0),
                          inter_frac=0.95, skip_frac=0.3, direct_frac=0.3)
    lumas, want = [], []
    su.decode_oracle(data, order="decoding", keep_luma=lumas, keep_params=want)
    assert len(want) == n and sum(int(p["needs_ref_luma"]) for p in want) >= 8
    assert sum(int(((p["batch"]["dmvr"] > 0) & (p["batch"]["refi"].min(1) >= 0)).sum()) for p in want if p["batch"]["dmvr"] is not None) >= 10

    def run(delay_s, lag, cancel_at=None, threads=1):
        handed, got, err = queue.Queue(), [], []

        def parse():
            try:
                for p in stream.iter_stream(data, threads=threads, luma_wait=True):
                    got.append(p)
                    handed.put(p)
            except RuntimeError as e:
                err.append(str(e))
            handed.put(None)
        t = threading.Thread(target=parse)
        t.start()
        pending, k = [], 0
        while True:
            p = handed.get()
            if p is None:
                break
            pending.append((k, p))
            k += 1
            while len(pending) > lag:                    # `lag` pictures behind the parser, and late
                j, q = pending.pop(0)
                if cancel_at is not None and j == cancel_at:
                    q["cancel_wait"]()
                    pending = []
                    break
                time.sleep(delay_s)
                if q["needs_ref_luma"]:
                    assert lumas[j][0] == q["poc"]
                    q["set_ref_luma"](q["poc"], lumas[j][1], abi.PAD_L)
            if cancel_at is None and handed.empty() and pending and not t.is_alive():
                break
            # nothing handed out for a while: the parser waits for a plane - give it the oldest one
            while cancel_at is None and pending and handed.empty() and t.is_alive():
                time.sleep(0.002)
                if handed.empty() and t.is_alive():
                    j, q = pending.pop(0)
                    if q["needs_ref_luma"]:
                        q["set_ref_luma"](q["poc"], lumas[j][1], abi.PAD_L)
        t.join(60)
        assert not t.is_alive()
        return got, err

    for delay, lag, threads in ((0.004, 0, 1), (0.0, 2, 1), (0.002, 1, 4)):
        got, err = run(delay, lag, threads=threads)
        assert not err and len(got) == n
        for p, q in zip(want, got):
            assert p["poc"] == q["poc"]
            for k, v in p["batch"].items():
                assert np.array_equal(v, q["batch"][k]) if isinstance(v, np.ndarray) else (v == q["batch"][k] or k == "tiles"), (p["poc"], k)
    got, err = run(0.0, 0, cancel_at=3)
    assert len(err) == 1 and "reference picture" in err[0] and len(got) < n


def test_parser_rebind_equals_a_new_parser():
    """xhost_parser_rebind: one parser object (its picture maps, motion-field pool, tile batches and tile threads kept) on one independent byte string after the
    other hands out exactly the pictures a new parser per string hands out - different geometries, profiles and tool sets in turn (what a work-queue worker
    does with the GOP jobs it draws)"""
    import ctypes as C
    lib = stream.load()
    lib.xhost_parser_rebind.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    a = su.make_stream(256, 192, 5, seed=3, main=True, iqt=True, addb=True, alf=True, admvp=True, max_refs=2, tiles=(2, 2, 1))
    b = su.make_stream(192, 128, 4, seed=4)

    def parse_all(h):
        out = []
        while True:
            hp = stream.HostPicture()
            rc = lib.xhost_parser_next(h, C.byref(hp))
            if rc <= 0:
                assert rc == 0, lib.xhost_parser_error(h)
                return out
            n = hp.batch.n_cu
            out.append((hp.poc, n, np.ctypeslib.as_array(hp.batch.mv, (n * 4,)).tobytes(), np.ctypeslib.as_array(hp.batch.pred_mode, (n,)).tobytes(), hp.batch.n_coef,
                        np.ctypeslib.as_array(hp.batch.coef, (max(hp.batch.n_coef, 1),)).tobytes(), [hp.refp_poc[i][0] for i in range(hp.num_refp[0])]))

    def fresh(d):
        h = lib.xhost_parser_open(d, len(d))
        lib.xhost_parser_set_threads(h, 3)
        r = parse_all(h)
        lib.xhost_parser_close(h)
        return r
    ra, rb = fresh(a), fresh(b)
    assert len(ra) == 5 and len(rb) == 4
    h = lib.xhost_parser_open(a, len(a))
    lib.xhost_parser_set_threads(h, 3)
    assert parse_all(h) == ra
    for d, r in ((b, rb), (a, ra), (a, ra), (b, rb)):
        assert lib.xhost_parser_rebind(h, d, len(d)) == 0
        assert parse_all(h) == r
    # a unit that ends in an error leaves a parser that can be rebound
    assert lib.xhost_parser_rebind(h, a[:len(a) // 2], len(a) // 2) == 0
    hp = stream.HostPicture()
    while True:
        rc = lib.xhost_parser_next(h, C.byref(hp))
        if rc <= 0:
            break
    assert lib.xhost_parser_rebind(h, b, len(b)) == 0 and parse_all(h) == rb
    lib.xhost_parser_close(h)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(golden_io.GOLDEN, "stream_*.npz"))), ids=os.path.basename)
def test_golden_streams_parser_plus_oracle(path):
    """committed streams + the reference decoder's pictures (made by tests/golden/make_golden.py with oracle/_ref)"""
    d = np.load(path)
    ours = su.decode_oracle(d["bytes"].tobytes())
    assert len(ours) == int(d["n"])
    for k in range(len(ours)):
        for c in range(3):
            assert np.array_equal(ours[k][c], d[f"p{k}_{c}"]), f"picture {k} plane {c}"


@pytest.mark.parametrize("kw", [dict(log2_sub_gop=2, max_refs=2), dict(main=True, iqt=True, addb=True, alf=True, bit_depth=10)],
                         ids=["base_hier_b", "main_alf_addb_10b"])
def test_md5_sei_round_trip(kw):
    """The MD5 round trip: the writer signs every picture with the MD5s of the oracle's reconstruction; the REFERENCE
    decoder, told to verify signatures (XEVD_CFG_SET_USE_PIC_SIGNATURE), accepts the stream - and rejects it once a digest is
    damaged (XEVD_ERR_BAD_CRC).  Our parser hands the digests out with the picture."""
    if not su.have_ref_decoder():
        pytest.skip("oracle/_ref/ref_decode not built")
    main = kw.get("main", False)
    data = su.make_stream(136, 72, 5, seed=77, sign=True, **kw)
    pics = stream.parse_stream(data)
    assert all(p["md5"] is not None for p in pics)
    ref = su.decode_reference(data, 136, 72, main=main)              # raises if the reference's own MD5 check fails
    ora = su.decode_oracle(data)
    assert len(ref) == len(ora) == 5
    for a, b in zip(ref, ora):
        for c in range(3):
            assert np.array_equal(a[c], b[c])
    bad = bytearray(data)
    bad[len(bad) - 3] ^= 0x40                                          # inside the last SEI's V-plane digest
    with pytest.raises(RuntimeError):
        su.decode_reference(bytes(bad), 136, 72, main=main)


def test_parser_survives_mutated_streams():
    """robustness: bit flips, overwritten bytes and truncations of valid streams end in an error or in parsed pictures - never in a hang
    or a crash (the same mutations ran 21000 times under ASan/UBSan while this was written)"""
    rng = np.random.default_rng(2024)
    seeds = [np.load(p)["bytes"].tobytes() for p in sorted(glob.glob(os.path.join(golden_io.GOLDEN, "stream_*.npz")))]
    outcomes = {"error": 0, "pictures": 0}
    for it in range(400):
        data = bytearray(seeds[it % len(seeds)])
        for _ in range(int(rng.integers(1, 5))):
            pos, kind = int(rng.integers(0, len(data))), int(rng.integers(0, 4))
            if kind == 0:
                data[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                data[pos] = int(rng.integers(0, 256))
            elif kind == 2:
                del data[1 + int(rng.integers(0, len(data) - 1)):]
            elif pos + 1 < len(data):
                data[pos] = data[pos + 1] = 0xFF
        try:
            for _ in stream.iter_stream(bytes(data)):
                outcomes["pictures"] += 1
        except RuntimeError:
            outcomes["error"] += 1
    assert outcomes["error"] > 100 and outcomes["pictures"] > 100


@pytest.mark.ref
def test_bench_stream_reference_decoder_equals_parser_plus_oracle():
    """the stream bench.py decodes in its real-bitstream leg (write_bench_stream: random-access Main, two lists of two references, tool_admvp, IQT, ADDB,
    ALF, 4x4 tiles; one closed GOP repeated): at a small size, the reference decoder == parser + oracle, and every repeated IDR period decodes to the
    same pictures - which is what lets the bench compare all periods against ONE period of the reference decoder"""
    if not su.have_ref_decoder():
        pytest.skip("oracle/_ref is not built")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    wl = dict(bench.WORKLOADS["cfg4_main_8k_10b_ra"])
    wl["w"], wl["h"] = 328, 264
    one, data, what = bench.write_bench_stream(wl, 17, 2, seed=5)
    assert "random access" in what
    pics = stream.parse_stream(data)
    assert len(pics) == 34 and sum(p["slice_type"] == stream.SLICE_B for p in pics) == 2 * 14 and all(p["batch"]["tiles"] is not None for p in pics)
    ref = su.decode_reference(data, wl["w"], wl["h"], main=True)
    ours = su.decode_oracle(data)
    assert len(ref) == 34 and len(ours) == 34
    for k in range(34):
        for c in range(3):
            assert np.array_equal(ref[k][c], ours[k][c]), f"picture {k} plane {c}"
            assert np.array_equal(ours[k][c], ours[k % 17][c]), f"IDR period 2, picture {k % 17}, plane {c}"
