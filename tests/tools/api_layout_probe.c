/* api_layout_probe.c - prints the value of every constant and the size / offset of every struct field of the public decoder API as the header named by
 * -DPROBE_HEADER declares them; tests/test_abi.py builds it against include/xevd_api.h and (development container) the reference's inc/xevd.h and compares. */
#include PROBE_HEADER
#include <stddef.h>
#include <stdio.h>
#define C(n) printf("%s %lld\n", #n, (long long)(n))
#define S(t) printf("sizeof %s %zu\n", #t, sizeof(t))
#define F(t, f) printf("%s.%s %zu %zu\n", #t, #f, offsetof(t, f), sizeof(((t *)0)->f))
int main(void)
{
    C(XEVD_MAX_TASK_CNT); C(XEVD_OK); C(XEVD_WARN_CRC_IGNORED); C(XEVD_OK_FRM_DELAYED); C(XEVD_OK_DIM_CHANGED); C(XEVD_OK_OUT_NOT_AVAILABLE); C(XEVD_OK_NO_MORE_FRM);
    C(XEVD_ERR); C(XEVD_ERR_INVALID_ARGUMENT); C(XEVD_ERR_OUT_OF_MEMORY); C(XEVD_ERR_REACHED_MAX); C(XEVD_ERR_UNSUPPORTED); C(XEVD_ERR_UNEXPECTED);
    C(XEVD_ERR_UNSUPPORTED_COLORSPACE); C(XEVD_ERR_MALFORMED_BITSTREAM); C(XEVD_ERR_THREAD_ALLOCATION); C(XEVD_ERR_BAD_CRC); C(XEVD_ERR_UNKNOWN);
    C(XEVD_SUCCEEDED(XEVD_WARN_CRC_IGNORED)); C(XEVD_FAILED(XEVD_ERR_BAD_CRC));
    C(XEVD_CF_UNKNOWN); C(XEVD_CF_YCBCR400); C(XEVD_CF_YCBCR420); C(XEVD_CF_YCBCR422); C(XEVD_CF_YCBCR444); C(XEVD_CF_YCBCR422N); C(XEVD_CF_YCBCR422W);
    C(XEVD_CS_UNKNOWN); C(XEVD_CS_YCBCR400); C(XEVD_CS_YCBCR420); C(XEVD_CS_YCBCR422); C(XEVD_CS_YCBCR444); C(XEVD_CS_YCBCR400_10LE); C(XEVD_CS_YCBCR420_10LE);
    C(XEVD_CS_YCBCR422_10LE); C(XEVD_CS_YCBCR444_10LE); C(XEVD_CS_YCBCR400_12LE); C(XEVD_CS_YCBCR420_12LE); C(XEVD_CS_YCBCR400_14LE); C(XEVD_CS_YCBCR420_14LE);
    C(XEVD_CS_GET_FORMAT(0x4A0B)); C(XEVD_CS_GET_BIT_DEPTH(0x4A0B)); C(XEVD_CS_GET_BYTE_DEPTH(0x4A0B)); C(XEVD_CS_GET_ENDIAN(0x4A0B));
    C(XEVD_CS_SET_FORMAT(0x4A0B, 13)); C(XEVD_CS_SET_BIT_DEPTH(0x4A0B, 12)); C(XEVD_CS_SET_ENDIAN(0x4A0B, 0));
    C(XEVD_CFG_SET_USE_PIC_SIGNATURE); C(XEVD_CFG_GET_CODEC_BIT_DEPTH); C(XEVD_CFG_GET_WIDTH); C(XEVD_CFG_GET_HEIGHT); C(XEVD_CFG_GET_CODED_WIDTH);
    C(XEVD_CFG_GET_CODED_HEIGHT); C(XEVD_CFG_GET_COLOR_SPACE); C(XEVD_CFG_GET_MAX_CODING_DELAY);
    C(XEVD_NAL_UNIT_LENGTH_BYTE); C(XEVD_NUT_NONIDR); C(XEVD_NUT_IDR); C(XEVD_NUT_SPS); C(XEVD_NUT_PPS); C(XEVD_NUT_APS); C(XEVD_NUT_FD); C(XEVD_NUT_SEI);
    C(XEVD_ST_UNKNOWN); C(XEVD_ST_B); C(XEVD_ST_P); C(XEVD_ST_I);
    C(XEVD_SEI_BUFFERING_PERIOD); C(XEVD_SEI_PICTURE_TIMING); C(XEVD_SEI_USER_DATA_REGISTERED_ITU_T_T35); C(XEVD_SEI_USER_DATA_UNREGISTERED); C(XEVD_SEI_RECOVERY_POINT);
    C(XEVD_SEI_MASTERING_DISPLAY_INFO); C(XEVD_SEI_CONTENT_LIGHT_LEVEL_INFO); C(XEVD_SEI_AMBIENT_VIEWING_ENVIRONMENT); C(XEVD_IMGB_SEI_SLOT); C(XEVD_SEI_MAGIC);
    C(XEVD_TS_PTS); C(XEVD_TS_DTS); C(XEVD_TS_NUM); C(XEVD_NDATA_NUM); C(XEVD_PDATA_NUM); C(XEVD_IMGB_MAX_PLANE);
    S(XEVD_MTIME); S(XEVD); S(XEVD_SEI_PAYLOAD_TYPE);
    S(XEVD_SEI_PAYLOAD); F(XEVD_SEI_PAYLOAD, payload_size); F(XEVD_SEI_PAYLOAD, payload_type); F(XEVD_SEI_PAYLOAD, payload);
    S(XEVD_SEI); F(XEVD_SEI, num_payloads); F(XEVD_SEI, payloads);
    S(XEVD_IMGB); F(XEVD_IMGB, cs); F(XEVD_IMGB, np); F(XEVD_IMGB, w); F(XEVD_IMGB, h); F(XEVD_IMGB, x); F(XEVD_IMGB, y); F(XEVD_IMGB, s); F(XEVD_IMGB, e); F(XEVD_IMGB, a);
    F(XEVD_IMGB, ts); F(XEVD_IMGB, ndata); F(XEVD_IMGB, pdata); F(XEVD_IMGB, aw); F(XEVD_IMGB, ah); F(XEVD_IMGB, padl); F(XEVD_IMGB, padr); F(XEVD_IMGB, padu); F(XEVD_IMGB, padb);
    F(XEVD_IMGB, baddr); F(XEVD_IMGB, bsize); F(XEVD_IMGB, refcnt); F(XEVD_IMGB, addref); F(XEVD_IMGB, getref); F(XEVD_IMGB, release);
    F(XEVD_IMGB, crop_idx); F(XEVD_IMGB, crop_l); F(XEVD_IMGB, crop_r); F(XEVD_IMGB, crop_t); F(XEVD_IMGB, crop_b); F(XEVD_IMGB, imgb_active_pps_id); F(XEVD_IMGB, imgb_active_aps_id);
    S(XEVD_BITB); F(XEVD_BITB, addr); F(XEVD_BITB, pddr); F(XEVD_BITB, bsize); F(XEVD_BITB, ssize); F(XEVD_BITB, err); F(XEVD_BITB, ndata); F(XEVD_BITB, pdata); F(XEVD_BITB, ts);
    S(XEVD_CDSC); F(XEVD_CDSC, threads);
    S(XEVD_STAT); F(XEVD_STAT, read); F(XEVD_STAT, nalu_type); F(XEVD_STAT, stype); F(XEVD_STAT, fnum); F(XEVD_STAT, poc); F(XEVD_STAT, tid); F(XEVD_STAT, refpic_num); F(XEVD_STAT, refpic);
    S(XEVD_INFO); F(XEVD_INFO, nalu_len); F(XEVD_INFO, nalu_type); F(XEVD_INFO, nalu_tid);
    /* the six entry points have the reference's prototypes: assignments to typed pointers fail to compile otherwise */
    { XEVD (*f0)(XEVD_CDSC *, int *) = xevd_create; void (*f1)(XEVD) = xevd_delete; int (*f2)(XEVD, XEVD_BITB *, XEVD_STAT *) = xevd_decode;
      int (*f3)(XEVD, XEVD_IMGB **) = xevd_pull; int (*f4)(XEVD, int, void *, int *) = xevd_config; int (*f5)(void *, int, int, XEVD_INFO *) = xevd_info;
      printf("prototypes %d\n", f0 && f1 && f2 && f3 && f4 && f5); }
    return 0;
}
