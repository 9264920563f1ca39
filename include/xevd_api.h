/* xevd_api.h - the decoder API that xevd_amd/compat/xevd_api.cc (libxevd_amd_api.so) implements on the MI355X backend.
 *
 * This is an ABI restatement written for this repository, not the reference's header: an application built against the reference's
 * public header (inc/xevd.h:48-374 of mpeg5/xevd v0.7.0) links against libxevd_amd_api.so unchanged, and an application that includes
 * this file instead compiles unchanged, because every constant has the reference's value and every struct its layout (sizes and offsets:
 * tests/test_abi.py::test_public_api_header_matches_reference_layout compiles one probe against both headers in the development container).
 * Only what the six entry points exchange is declared; the library needs no generated export header.
 *
 *   xevd_create(cdsc, &err) -> id      inc/xevd.h:369        one decoder instance = one xgpu_ctx + one bitstream parser
 *   xevd_decode(id, bitb, stat)        inc/xevd.h:371        ONE NAL unit (no length prefix) per call
 *   xevd_pull(id, &imgb)               inc/xevd.h:372        next picture in output order, borrowed + addref'ed; the caller release()s it
 *   xevd_config / xevd_info / xevd_delete                    inc/xevd.h:370,373,374
 */
#ifndef XEVD_AMD_XEVD_API_H
#define XEVD_AMD_XEVD_API_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#ifndef XEVD_EXPORT
#define XEVD_EXPORT __attribute__((visibility("default")))
#endif

#define XEVD_MAX_TASK_CNT 8                          /* most threads XEVD_CDSC.threads may ask for */

/* status codes: 0 and positive = success, negative = failure */
enum {
    XEVD_OK = 0,
    XEVD_WARN_CRC_IGNORED = 200,                     /* a picture signature came with the picture and was not checked */
    XEVD_OK_FRM_DELAYED = 202, XEVD_OK_DIM_CHANGED = 203, XEVD_OK_OUT_NOT_AVAILABLE = 204, XEVD_OK_NO_MORE_FRM = 205,
    XEVD_ERR = -1,
    XEVD_ERR_INVALID_ARGUMENT = -101, XEVD_ERR_OUT_OF_MEMORY = -102, XEVD_ERR_REACHED_MAX = -103, XEVD_ERR_UNSUPPORTED = -104, XEVD_ERR_UNEXPECTED = -105,
    XEVD_ERR_UNSUPPORTED_COLORSPACE = -201, XEVD_ERR_MALFORMED_BITSTREAM = -202, XEVD_ERR_THREAD_ALLOCATION = -203,
    XEVD_ERR_BAD_CRC = -300,
    XEVD_ERR_UNKNOWN = -32767
};
#define XEVD_SUCCEEDED(r) ((r) >= XEVD_OK)
#define XEVD_FAILED(r)    ((r) < XEVD_OK)

/* colour space word: format in bits 0-7, bit depth in bits 8-13, endianness in bit 14 */
enum { XEVD_CF_UNKNOWN = 0, XEVD_CF_YCBCR400 = 10, XEVD_CF_YCBCR420 = 11, XEVD_CF_YCBCR422 = 12, XEVD_CF_YCBCR444 = 13, XEVD_CF_YCBCR422N = 12, XEVD_CF_YCBCR422W = 18 };
#define XEVD_CS_SET(fmt, depth, big_endian) (((big_endian) << 14) | ((depth) << 8) | (fmt))
#define XEVD_CS_GET_FORMAT(cs)              ((cs) & 0xFF)
#define XEVD_CS_GET_BIT_DEPTH(cs)           (((cs) >> 8) & 0x3F)
#define XEVD_CS_GET_BYTE_DEPTH(cs)          ((XEVD_CS_GET_BIT_DEPTH(cs) + 7) >> 3)
#define XEVD_CS_GET_ENDIAN(cs)              (((cs) >> 14) & 1)
#define XEVD_CS_SET_FORMAT(cs, v)           (((cs) & ~0xFF) | (v))
#define XEVD_CS_SET_BIT_DEPTH(cs, v)        (((cs) & ~(0x3F << 8)) | ((v) << 8))
#define XEVD_CS_SET_ENDIAN(cs, v)           (((cs) & ~(1 << 14)) | ((v) << 14))
#define XEVD_CS_UNKNOWN        XEVD_CS_SET(0, 0, 0)
#define XEVD_CS_YCBCR400       XEVD_CS_SET(XEVD_CF_YCBCR400, 8, 0)
#define XEVD_CS_YCBCR420       XEVD_CS_SET(XEVD_CF_YCBCR420, 8, 0)
#define XEVD_CS_YCBCR422       XEVD_CS_SET(XEVD_CF_YCBCR422, 8, 0)
#define XEVD_CS_YCBCR444       XEVD_CS_SET(XEVD_CF_YCBCR444, 8, 0)
#define XEVD_CS_YCBCR400_10LE  XEVD_CS_SET(XEVD_CF_YCBCR400, 10, 0)
#define XEVD_CS_YCBCR420_10LE  XEVD_CS_SET(XEVD_CF_YCBCR420, 10, 0)
#define XEVD_CS_YCBCR422_10LE  XEVD_CS_SET(XEVD_CF_YCBCR422, 10, 0)
#define XEVD_CS_YCBCR444_10LE  XEVD_CS_SET(XEVD_CF_YCBCR444, 10, 0)
#define XEVD_CS_YCBCR400_12LE  XEVD_CS_SET(XEVD_CF_YCBCR400, 12, 0)
#define XEVD_CS_YCBCR420_12LE  XEVD_CS_SET(XEVD_CF_YCBCR420, 12, 0)
#define XEVD_CS_YCBCR400_14LE  XEVD_CS_SET(XEVD_CF_YCBCR400, 14, 0)
#define XEVD_CS_YCBCR420_14LE  XEVD_CS_SET(XEVD_CF_YCBCR420, 14, 0)

/* xevd_config selectors */
enum {
    XEVD_CFG_SET_USE_PIC_SIGNATURE = 301,            /* int: verify the MD5 picture signature SEI */
    XEVD_CFG_GET_CODEC_BIT_DEPTH = 401, XEVD_CFG_GET_WIDTH = 402, XEVD_CFG_GET_HEIGHT = 403, XEVD_CFG_GET_CODED_WIDTH = 404, XEVD_CFG_GET_CODED_HEIGHT = 405,
    XEVD_CFG_GET_COLOR_SPACE = 406, XEVD_CFG_GET_MAX_CODING_DELAY = 407
};

/* NAL units: a 4-byte big-endian length in front of each in a file; nal_unit_type_plus1 - 1 below */
enum { XEVD_NAL_UNIT_LENGTH_BYTE = 4, XEVD_NUT_NONIDR = 0, XEVD_NUT_IDR = 1, XEVD_NUT_SPS = 24, XEVD_NUT_PPS = 25, XEVD_NUT_APS = 26, XEVD_NUT_FD = 27, XEVD_NUT_SEI = 28 };
enum { XEVD_ST_UNKNOWN = -1, XEVD_ST_B = 0, XEVD_ST_P = 1, XEVD_ST_I = 2 };      /* XEVD_STAT.stype */

/* SEI messages handed out with a pulled picture: imgb->ndata[XEVD_IMGB_SEI_SLOT] == XEVD_SEI_MAGIC says imgb->pdata[same slot] is an XEVD_SEI owned by the library */
typedef enum _XEVD_SEI_PAYLOAD_TYPE {
    XEVD_SEI_BUFFERING_PERIOD = 0, XEVD_SEI_PICTURE_TIMING = 1, XEVD_SEI_USER_DATA_REGISTERED_ITU_T_T35 = 4, XEVD_SEI_USER_DATA_UNREGISTERED = 5,
    XEVD_SEI_RECOVERY_POINT = 6, XEVD_SEI_MASTERING_DISPLAY_INFO = 137, XEVD_SEI_CONTENT_LIGHT_LEVEL_INFO = 144, XEVD_SEI_AMBIENT_VIEWING_ENVIRONMENT = 148
} XEVD_SEI_PAYLOAD_TYPE;
typedef struct _XEVD_SEI_PAYLOAD { int payload_size; XEVD_SEI_PAYLOAD_TYPE payload_type; unsigned char *payload; } XEVD_SEI_PAYLOAD;
typedef struct _XEVD_SEI { int num_payloads; XEVD_SEI_PAYLOAD *payloads; } XEVD_SEI;
#define XEVD_IMGB_SEI_SLOT 3
#define XEVD_SEI_MAGIC     0x58534549

typedef long long XEVD_MTIME;                        /* time stamps, 100 ns units */
enum { XEVD_TS_PTS = 0, XEVD_TS_DTS = 1, XEVD_TS_NUM = 2 };
#define XEVD_NDATA_NUM      4
#define XEVD_PDATA_NUM      4
#define XEVD_IMGB_MAX_PLANE 4

/* A picture in host memory.  Per plane: the visible w x h samples start x / y samples into an aligned aw x ah area inside an allocation of s bytes per row and
 * e rows' worth of bytes (baddr / bsize = the allocation, a = the first visible sample, pad* = the border around the aligned area).  Reference-counted. */
typedef struct _XEVD_IMGB XEVD_IMGB;
struct _XEVD_IMGB {
    int cs, np;
    int w[XEVD_IMGB_MAX_PLANE], h[XEVD_IMGB_MAX_PLANE], x[XEVD_IMGB_MAX_PLANE], y[XEVD_IMGB_MAX_PLANE], s[XEVD_IMGB_MAX_PLANE], e[XEVD_IMGB_MAX_PLANE];
    void *a[XEVD_IMGB_MAX_PLANE];
    XEVD_MTIME ts[XEVD_TS_NUM];
    int ndata[XEVD_NDATA_NUM];
    void *pdata[XEVD_PDATA_NUM];
    int aw[XEVD_IMGB_MAX_PLANE], ah[XEVD_IMGB_MAX_PLANE];
    int padl[XEVD_IMGB_MAX_PLANE], padr[XEVD_IMGB_MAX_PLANE], padu[XEVD_IMGB_MAX_PLANE], padb[XEVD_IMGB_MAX_PLANE];
    void *baddr[XEVD_IMGB_MAX_PLANE];
    int bsize[XEVD_IMGB_MAX_PLANE];
    int refcnt;
    int (*addref)(XEVD_IMGB *imgb);
    int (*getref)(XEVD_IMGB *imgb);
    int (*release)(XEVD_IMGB *imgb);
    int crop_idx, crop_l, crop_r, crop_t, crop_b;
    int imgb_active_pps_id, imgb_active_aps_id;
};

/* The bytes handed to xevd_decode: addr / ssize = one NAL unit; the caller owns the buffer */
typedef struct _XEVD_BITB {
    void *addr, *pddr;
    int bsize, ssize, err;
    int ndata[XEVD_NDATA_NUM];
    void *pdata[XEVD_PDATA_NUM];
    XEVD_MTIME ts[XEVD_TS_NUM];
} XEVD_BITB;

typedef struct _XEVD_CDSC { int threads; } XEVD_CDSC;

/* What xevd_decode reports about the NAL unit it consumed (fnum < 0: not a picture) */
typedef struct _XEVD_STAT {
    int read, nalu_type, stype, fnum, poc, tid;
    unsigned char refpic_num[2];
    int refpic[2][16];
} XEVD_STAT;

typedef struct _XEVD_INFO { int nalu_len, nalu_type, nalu_tid; } XEVD_INFO;

typedef void *XEVD;
XEVD XEVD_EXPORT xevd_create(XEVD_CDSC *cdsc, int *err);
void XEVD_EXPORT xevd_delete(XEVD id);
int  XEVD_EXPORT xevd_decode(XEVD id, XEVD_BITB *bitb, XEVD_STAT *stat);
int  XEVD_EXPORT xevd_pull(XEVD id, XEVD_IMGB **img);
int  XEVD_EXPORT xevd_config(XEVD id, int cfg, void *buf, int *size);
int  XEVD_EXPORT xevd_info(void *bits, int bits_size, int is_annexb, XEVD_INFO *info);

#ifdef __cplusplus
}
#endif
#endif
