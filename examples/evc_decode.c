/*
 * evc_decode.c - a complete decoder in plain C on the two C ABIs of this repository (no Python, no reference code):
 *   include/xevd_host.h  bitstream -> per-picture CU batches + DPB bookkeeping by POC      (libxevd_host.so, host only)
 *   include/xevd_hip.h   CU batches -> pictures on the MI355X                              (libxevd_hip.so)
 * It is the loop xevd_app runs around xevd_decode / xevd_pull (app/xevd_app.c:455-640), with the pictures living in HBM: parse a picture,
 * map its reference POCs to device picture slots, reconstruct + filter + pad it, release the pictures the stream unmarked, write the
 * output in POC order inside every IDR period (what xevd_pull's bumping yields).
 * usage: evc_decode in.evc out.yuv [output_bit_depth]        (0 / omitted: the coding bit depth; 8: one byte per sample)
 * build: oracle-free; see examples/Makefile.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "xevd_host.h"

#define MAX_SLOTS 24
#define CHECK(call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, g ? xgpu_last_error(g) : ""); return 1; } } while (0)

typedef struct { int poc, pic, in_use; } slot_t;                 /* DPB: POC -> device picture slot */
typedef struct { int epoch, poc; size_t off; } out_t;              /* one output picture: position in the staging file */

static int cmp_out(const void *a, const void *b)
{
    const out_t *x = (const out_t *)a, *y = (const out_t *)b;
    return x->epoch != y->epoch ? x->epoch - y->epoch : x->poc - y->poc;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.evc out.yuv [output_bit_depth]\n", argv[0]); return 2; }
    const int out_bd_arg = argc > 3 ? atoi(argv[3]) : 0;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *bytes = (uint8_t *)malloc((size_t)size);
    if (fread(bytes, 1, (size_t)size, f) != (size_t)size) return 2;
    fclose(f);

    xhost_parser *ps = xhost_parser_open(bytes, (size_t)size);
    xgpu_ctx *g = NULL;
    slot_t dpb[MAX_SLOTS];
    int free_pic[MAX_SLOTS], n_free = 0, n_pics = 0, epoch = -1, rc;
    out_t *outs = NULL;
    uint8_t *frames = NULL;                                         /* decoded pictures in decoding order, packed */
    size_t frame_bytes = 0, cap = 0;
    xhost_picture p;
    memset(dpb, 0, sizeof(dpb));

    while ((rc = xhost_parser_next(ps, &p)) == 1) {
        if (!g) {                                                    /* first picture: the sequence parameters are known */
            xgpu_seq_params sp;
            memset(&sp, 0, sizeof(sp));
            sp.device = 0; sp.width = p.width; sp.height = p.height;
            sp.bit_depth_luma = p.bit_depth_luma; sp.bit_depth_chroma = p.bit_depth_chroma; sp.chroma_format_idc = 1; sp.log2_ctu = 6;
            sp.tool_iqt = p.tool_iqt; sp.tool_addb = p.tool_addb; sp.tool_alf = p.tool_alf; sp.tool_eipd = p.tool_eipd;
            sp.max_pics = MAX_SLOTS + 2;
            sp.chroma_qp_table[0] = p.chroma_qp_table[0]; sp.chroma_qp_table[1] = p.chroma_qp_table[1];
            CHECK(xgpu_open(&sp, &g));
            for (int i = 0; i < MAX_SLOTS; i++) { const int id = xgpu_pic_alloc(g); if (id < 0) return 1; free_pic[n_free++] = id; }
            frame_bytes = xgpu_pic_output_size(g, out_bd_arg ? out_bd_arg : p.bit_depth_luma, 0, 0, 0, 0);
        }
        if (p.is_idr) {                                              /* an IDR empties the DPB */
            for (int i = 0; i < MAX_SLOTS; i++) if (dpb[i].in_use) { free_pic[n_free++] = dpb[i].pic; dpb[i].in_use = 0; }
            epoch++;
        }
        if (n_free == 0) { fprintf(stderr, "DPB overflow\n"); return 1; }
        const int cur = free_pic[--n_free];

        xgpu_frame_params fp;
        memset(&fp, 0, sizeof(fp));
        fp.pic = cur; fp.poc = p.poc;
        for (int l = 0; l < 2; l++) {
            fp.num_refp[l] = p.num_refp[l];
            for (int i = 0; i < p.num_refp[l]; i++) {
                int s = -1;
                for (int k = 0; k < MAX_SLOTS; k++) if (dpb[k].in_use && dpb[k].poc == p.refp_poc[i][l]) s = dpb[k].pic;
                if (s < 0) { fprintf(stderr, "reference POC %d is not in the DPB\n", p.refp_poc[i][l]); return 1; }
                fp.refp_pic[i][l] = s; fp.refp_poc[i][l] = p.refp_poc[i][l];
            }
        }
        fp.qp_u_offset = p.qp_u_offset; fp.qp_v_offset = p.qp_v_offset;
        fp.deblock_alpha_offset = p.deblock_alpha_offset; fp.deblock_beta_offset = p.deblock_beta_offset;
        fp.deblock_on = p.deblock_on; fp.alf_on = p.alf_on;

        xgpu_dbatch *db = NULL;
        CHECK(xgpu_batch_create(g, &p.batch, &db));                 /* the parser's arrays are consumed before the call returns */
        CHECK(xgpu_frame_begin(g, &fp));
        CHECK(xgpu_batch_recon(g, db));
        if (p.deblock_on) CHECK(xgpu_deblock(g));
        if (p.alf_on) CHECK(xgpu_alf(g, &p.alf));
        CHECK(xgpu_pad(g));
        CHECK(xgpu_frame_end(g));
        xgpu_batch_destroy(g, db);

        if ((size_t)(n_pics + 1) * frame_bytes > cap) {
            cap = cap ? cap * 2 : 16 * frame_bytes;
            frames = (uint8_t *)realloc(frames, cap);
            outs = (out_t *)realloc(outs, sizeof(out_t) * (cap / frame_bytes));
            if (!frames || !outs) return 1;
        }
        xgpu_dra_luts dra = { p.dra_lut[0], { p.dra_lut[1], p.dra_lut[2] } };          /* the DRA post-filter, when the PPS switches it on */
        CHECK(xgpu_pic_output(g, cur, p.dra_lut[0] ? &dra : NULL, out_bd_arg ? out_bd_arg : p.bit_depth_luma, 0, 0, 0, 0, frames + (size_t)n_pics * frame_bytes, frame_bytes));
        outs[n_pics].epoch = epoch; outs[n_pics].poc = p.poc; outs[n_pics].off = (size_t)n_pics * frame_bytes;
        n_pics++;

        for (int r = 0; r < p.n_release; r++)                        /* pictures the stream unmarked when this one was stored */
            for (int k = 0; k < MAX_SLOTS; k++)
                if (dpb[k].in_use && dpb[k].poc == p.release_poc[r]) { free_pic[n_free++] = dpb[k].pic; dpb[k].in_use = 0; }
        if (p.is_ref) {
            for (int k = 0; k < MAX_SLOTS; k++) if (!dpb[k].in_use) { dpb[k].in_use = 1; dpb[k].poc = p.poc; dpb[k].pic = cur; break; }
        } else free_pic[n_free++] = cur;
    }
    if (rc < 0) { fprintf(stderr, "parser: %s\n", xhost_parser_error(ps)); return 1; }

    qsort(outs, (size_t)n_pics, sizeof(out_t), cmp_out);            /* output order: ascending POC inside every IDR period */
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 2; }
    for (int i = 0; i < n_pics; i++) fwrite(frames + outs[i].off, 1, frame_bytes, f);
    fclose(f);
    fprintf(stderr, "%d pictures\n", n_pics);
    xhost_parser_close(ps);
    if (g) xgpu_close(g);
    free(frames); free(outs); free(bytes);
    return 0;
}
