// Constants of the EVC specification (ISO/IEC 23094-1, initialisation of the context variables with sps_cm_init_flag): initValue per context and slice kind
// ([0] I / P slices, [1] B slices; init_* in src_main/xevdm_tbl.c:64-377), in the order of the front end's context arrays.  Data only; generated once from the
// standard's tables.
#pragma once
#include <stdint.h>
static const int16_t k_cm_split[2][1] = { { 0 }, { 0 } };
static const int16_t k_cm_run[2][24] = { { 48, 112, 128, 0, 321, 82, 419, 160, 385, 323, 353, 129, 225, 193, 387, 389, 453, 227, 453, 161, 421, 161, 481, 225 }, { 129, 178, 453, 97, 583, 259, 517, 259, 453, 227, 871, 355, 291, 227, 195, 97, 161, 65, 97, 33, 65, 1, 1003, 227 } };
static const int16_t k_cm_last[2][2] = { { 421, 337 }, { 33, 790 } };
static const int16_t k_cm_level[2][24] = { { 416, 98, 128, 66, 32, 82, 17, 48, 272, 112, 52, 50, 448, 419, 385, 355, 161, 225, 82, 97, 210, 0, 416, 224 }, { 805, 775, 775, 581, 355, 389, 65, 195, 48, 33, 224, 225, 775, 227, 355, 161, 129, 97, 33, 65, 16, 1, 841, 355 } };
static const int16_t k_cm_cbf_luma[2][1] = { { 664 }, { 368 } };
static const int16_t k_cm_cbf_cb[2][1] = { { 384 }, { 416 } };
static const int16_t k_cm_cbf_cr[2][1] = { { 320 }, { 288 } };
static const int16_t k_cm_cbf_all[2][1] = { { 0 }, { 794 } };
static const int16_t k_cm_pred_mode[2][3] = { { 64, 0, 0 }, { 481, 16, 368 } };
static const int16_t k_cm_direct[2][1] = { { 0 }, { 0 } };
static const int16_t k_cm_inter_dir[2][2] = { { 0, 0 }, { 242, 80 } };
static const int16_t k_cm_intra_dir[2][2] = { { 0, 0 }, { 0, 0 } };
static const int16_t k_cm_mvp_idx[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } };
static const int16_t k_cm_mvd[2][1] = { { 0 }, { 18 } };
static const int16_t k_cm_refi[2][2] = { { 0, 0 }, { 288, 0 } };
static const int16_t k_cm_dqp[2][1] = { { 4 }, { 4 } };
static const int16_t k_cm_skip[2][2] = { { 0, 0 }, { 711, 233 } };
static const int16_t k_cm_ats_mode[2][1] = { { 512 }, { 673 } };
static const int16_t k_cm_ats_inter_flag[2][2] = { { 0, 0 }, { 0, 0 } };
static const int16_t k_cm_ats_inter_quad[2][1] = { { 0 }, { 0 } };
static const int16_t k_cm_ats_inter_hor[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } };
static const int16_t k_cm_ats_inter_pos[2][1] = { { 0 }, { 0 } };
static const int16_t k_cm_alf_ctb[2][1] = { { 0 }, { 0 } };
static const int16_t k_cm_mmvd_flag[2][1] = { { 0 }, { 194 } };
static const int16_t k_cm_mmvd_merge_idx[2][3] = { { 0, 0, 0 }, { 49, 129, 82 } };
static const int16_t k_cm_mmvd_dist_idx[2][7] = { { 0, 0, 0, 0, 0, 0, 0 }, { 179, 5, 133, 131, 227, 64, 128 } };
static const int16_t k_cm_mmvd_dir_idx[2][2] = { { 0, 0 }, { 161, 33 } };
static const int16_t k_cm_mmvd_group_idx[2][2] = { { 0, 0 }, { 453, 48 } };
static const int16_t k_cm_mvr_idx[2][4] = { { 0, 0, 0, 496 }, { 773, 101, 421, 199 } };
static const int16_t k_cm_merge_mode[2][1] = { { 0 }, { 464 } };
static const int16_t k_cm_merge_idx[2][5] = { { 0, 0, 0, 496, 496 }, { 18, 128, 146, 37, 69 } };
static const int16_t k_cm_bi_idx[2][2] = { { 0, 0 }, { 49, 17 } };
static const int16_t k_cm_ibc_flag[2][2] = { { 0, 0 }, { 711, 233 } };
static const int16_t k_cm_affine_flag[2][2] = { { 0, 0 }, { 320, 210 } };
static const int16_t k_cm_affine_mode[2][1] = { { 0 }, { 225 } };
static const int16_t k_cm_affine_mrg[2][5] = { { 0, 0, 0, 0, 0 }, { 193, 129, 32, 323, 0 } };
static const int16_t k_cm_affine_mvp_idx[2][1] = { { 0 }, { 161 } };
static const int16_t k_cm_affine_mvd_flag[2][2] = { { 0, 0 }, { 547, 645 } };
static const int16_t k_cm_ipm_mpm_flag[2][1] = { { 263 }, { 225 } };
static const int16_t k_cm_ipm_mpm_idx[2][1] = { { 436 }, { 724 } };
static const int16_t k_cm_ipm_chroma[2][1] = { { 465 }, { 560 } };
static const int16_t k_cm_btt_split_flag[2][15] = { { 145, 560, 528, 308, 594, 560, 180, 500, 626, 84, 406, 662, 320, 36, 340 }, { 536, 726, 594, 66, 338, 528, 258, 404, 464, 98, 342, 370, 384, 256, 65 } };
static const int16_t k_cm_btt_split_dir[2][5] = { { 0, 417, 389, 99, 0 }, { 0, 128, 81, 49, 0 } };
static const int16_t k_cm_btt_split_type[2][1] = { { 257 }, { 225 } };
static const int16_t k_cm_mode_cons[2][3] = { { 64, 0, 0 }, { 481, 16, 368 } };
static const int16_t k_cm_suco_flag[2][14] = { { 0, 0, 0, 0, 0, 0, 545, 0, 481, 515, 0, 32, 0, 0 }, { 0, 0, 0, 0, 0, 0, 577, 0, 481, 2, 0, 97, 0, 0 } };      // init_suco_flag, xevdm_tbl.c:328-332
static const int16_t k_cm_sig_coeff[2][47] = { { 387, 98, 233, 346, 717, 306, 233, 37, 321, 293, 244, 37, 329, 645, 408, 493, 164, 781, 101, 179, 369, 871, 585, 244, 361, 147, 416, 408, 628, 352, 406, 502, 566, 466, 54, 97, 521, 113, 147, 519, 36, 297, 132, 457, 308, 231, 534 }, { 66, 34, 241, 321, 293, 113, 35, 83, 226, 519, 553, 229, 751, 224, 129, 133, 162, 227, 178, 165, 532, 417, 357, 33, 489, 199, 387, 939, 133, 515, 32, 131, 3, 305, 579, 323, 65, 99, 425, 453, 291, 329, 679, 683, 391, 751, 51 } };
static const int16_t k_cm_gt_ab[2][18] = { { 40, 225, 306, 272, 85, 120, 389, 664, 209, 322, 291, 536, 338, 709, 54, 244, 19, 566 }, { 38, 352, 340, 19, 305, 258, 18, 33, 209, 773, 517, 406, 719, 741, 613, 295, 37, 498 } };
static const int16_t k_cm_last_x[2][21] = { { 762, 310, 288, 828, 342, 451, 502, 51, 97, 416, 662, 890, 340, 146, 20, 337, 468, 975, 216, 66, 54 }, { 892, 84, 581, 600, 278, 419, 372, 568, 408, 485, 338, 632, 666, 732, 17, 178, 180, 585, 581, 34, 257 } };
static const int16_t k_cm_last_y[2][21] = { { 81, 440, 4, 534, 406, 226, 370, 370, 259, 38, 598, 792, 860, 312, 88, 662, 924, 161, 248, 20, 54 }, { 470, 376, 323, 276, 602, 52, 340, 600, 376, 378, 598, 502, 730, 538, 17, 195, 504, 378, 320, 160, 572 } };
