// xevd_api.cc - the reference's PUBLIC decoder API (inc/xevd.h:369-374: xevd_create / xevd_decode / xevd_pull / xevd_config / xevd_delete /
// xevd_info) implemented on this repository's two C ABIs: an application written for libxevd - the reference's own xevd_app included - links
// against libxevd_amd_api.so instead and decodes on the MI355X.  No reference source is part of this file; it is compiled against this repository's own
// restatement of that ABI (include/xevd_api.h: the reference's constants and struct layouts, pinned by tests/test_abi.py), so it builds anywhere.
// Streams: what include/xevd_host.h parses.
//
//   xevd_decode(one NAL unit)  -> xhost_parser_nal; a picture: map reference POCs to device slots, reconstruct + filter + pad on the GPU,
//                                 download it into a host XEVD_IMGB (16-bit planes, what the reference's pictures are) and queue it for output
//   xevd_pull                  -> pictures in POC order inside every IDR period.  A picture leaves when the next temporal-layer-0 picture
//                                 has arrived (all pictures of its sub-GOP are decoded by then); a pull that is not preceded by a decode is the
//                                 application's bumping phase: everything left, then XEVD_ERR_UNEXPECTED (src_base/xevd.c:2042-2071 behaviour)
//   picture-signature SEI      -> with XEVD_CFG_SET_USE_PIC_SIGNATURE the MD5 of every plane is checked (XEVD_ERR_BAD_CRC), else
//                                 XEVD_WARN_CRC_IGNORED, as xevd.c:2010-2026
#include "xevd_api.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#include "xevd_host.h"

namespace {

// ---- MD5 (RFC 1321), for the picture signatures ----
struct Md5 {
    uint32_t h[4] = { 0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u };
    uint8_t buf[64];
    uint64_t len = 0;
    static uint32_t rol(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }
    void block(const uint8_t *p)
    {
        static const int S[64] = { 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                   4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21 };
        static uint32_t K[64];
        static bool init = false;
        if (!init) { for (int i = 0; i < 64; i++) { double v = __builtin_fabs(__builtin_sin((double)(i + 1))) * 4294967296.0; K[i] = (uint32_t)v; } init = true; }
        uint32_t m[16], a = h[0], b = h[1], c = h[2], d = h[3];
        for (int i = 0; i < 16; i++) m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        for (int i = 0; i < 64; i++) {
            uint32_t f; int g;
            if (i < 16) { f = (b & c) | (~b & d); g = i; }
            else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
            else { f = c ^ (b | ~d); g = (7 * i) & 15; }
            const uint32_t t = d; d = c; c = b; b = b + rol(a + f + K[i] + m[g], S[i]); a = t;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d;
    }
    void update(const uint8_t *p, size_t n)
    {
        size_t fill = (size_t)(len & 63);
        len += n;
        if (fill) { const size_t take = std::min(n, 64 - fill); memcpy(buf + fill, p, take); p += take; n -= take; if (fill + take < 64) return; block(buf); }
        for (; n >= 64; p += 64, n -= 64) block(p);
        memcpy(buf, p, n);
    }
    void finish(uint8_t out[16])
    {
        const uint64_t bits = len * 8;
        const uint8_t pad = 0x80, zero = 0;
        update(&pad, 1);
        while ((len & 63) != 56) update(&zero, 1);
        uint8_t l[8];
        for (int i = 0; i < 8; i++) l[i] = (uint8_t)(bits >> (8 * i));
        update(l, 8);
        for (int i = 0; i < 4; i++) for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(h[i] >> (8 * k));
    }
};

struct Image {                      // an XEVD_IMGB that owns three tight 16-bit planes
    XEVD_IMGB img;
    std::vector<int16_t> mem;
    int epoch, poc;
    bool ready;                     // may leave through xevd_pull (the next temporal-layer-0 picture has arrived)
    bool have_dig = false;          // XEVD_AMD_MD5_ON_DEVICE: the picture's signature was made on the device (xgpu_pic_md5) when it was decoded
    uint8_t dig[3][16];
};
int img_addref(XEVD_IMGB *i) { return ++i->refcnt; }
int img_getref(XEVD_IMGB *i) { return i->refcnt; }
int img_release(XEVD_IMGB *i)
{
    if (--i->refcnt > 0) return i->refcnt;
    delete (Image *)i->pdata[0];
    return 0;
}

enum { MAX_SLOTS = 24 };
struct Slot { int poc, pic, in_use; };

struct Decoder {
    xhost_parser *ps = nullptr;
    xgpu_ctx *g = nullptr;
    Slot dpb[MAX_SLOTS];
    std::vector<int> free_pic;
    std::vector<Image *> pending;          // decoded, not yet pulled
    int epoch = -1, last_key_poc = -1, pic_cnt = 0;
    bool decoded_since_pull = false, use_sig = false;
    bool md5_on_device = getenv("XEVD_AMD_MD5_ON_DEVICE") != nullptr;      // picture signatures by xgpu_pic_md5 instead of the host's MD5 (read per decoder)
    Image *last = nullptr;                 // the picture a signature SEI refers to (it stays in `pending` or with the caller)
    std::map<int, std::vector<int16_t>> ref_luma;      // by device picture slot: host copies of decoded luma planes for the parser's own DMVR search
    int w = 0, h = 0, bd = 8;
    Decoder() { memset(dpb, 0, sizeof(dpb)); }
};

int map_err(int rc) { return rc == XHOST_ERR_MALFORMED ? XEVD_ERR_MALFORMED_BITSTREAM : (rc < 0 ? rc : XEVD_OK); }

int decode_picture(Decoder *d, const xhost_picture &p, Image **out)
{
    if (!d->g) {
        xgpu_seq_params sp;
        memset(&sp, 0, sizeof(sp));
        sp.width = p.width; sp.height = p.height; sp.bit_depth_luma = p.bit_depth_luma; sp.bit_depth_chroma = p.bit_depth_chroma;
        sp.chroma_format_idc = 1; sp.log2_ctu = 6; sp.max_pics = MAX_SLOTS + 2;
        sp.tool_iqt = p.tool_iqt; sp.tool_addb = p.tool_addb; sp.tool_alf = p.tool_alf; sp.tool_eipd = p.tool_eipd; sp.tool_admvp = p.tool_admvp;
        sp.chroma_qp_table[0] = p.chroma_qp_table[0]; sp.chroma_qp_table[1] = p.chroma_qp_table[1];
        int rc = xgpu_open(&sp, &d->g);
        if (rc < 0) return rc;
        for (int i = 0; i < MAX_SLOTS; i++) { const int id = xgpu_pic_alloc(d->g); if (id < 0) return id; d->free_pic.push_back(id); }
        d->w = p.width; d->h = p.height; d->bd = p.bit_depth_luma;
    }
    if (p.is_idr) {
        for (Slot &s : d->dpb) if (s.in_use) { d->free_pic.push_back(s.pic); s.in_use = 0; }
        d->epoch++;
    }
    if (d->free_pic.empty()) return XEVD_ERR_UNEXPECTED;
    for (int l = 0; l < 2; l++) if (p.num_refp[l] < 0 || p.num_refp[l] > XGPU_MAX_REFS) return XEVD_ERR_MALFORMED_BITSTREAM;
    const int cur = d->free_pic.back();
    d->free_pic.pop_back();
    // every failure below hands the slot back: a damaged picture must not cost the decoder a picture buffer
    struct SlotGuard { Decoder *d; int pic; bool keep; ~SlotGuard() { if (!keep) d->free_pic.push_back(pic); } } guard = { d, cur, false };
    xgpu_frame_params fp;
    memset(&fp, 0, sizeof(fp));
    fp.pic = cur; fp.poc = p.poc;
    for (int l = 0; l < 2; l++) {
        fp.num_refp[l] = p.num_refp[l];
        for (int i = 0; i < p.num_refp[l]; i++) {
            int s = -1;
            for (const Slot &k : d->dpb) if (k.in_use && k.poc == p.refp_poc[i][l]) s = k.pic;
            if (s < 0) return XEVD_ERR_MALFORMED_BITSTREAM;
            fp.refp_pic[i][l] = s; fp.refp_poc[i][l] = p.refp_poc[i][l];
        }
    }
    fp.qp_u_offset = p.qp_u_offset; fp.qp_v_offset = p.qp_v_offset;
    fp.deblock_alpha_offset = p.deblock_alpha_offset; fp.deblock_beta_offset = p.deblock_beta_offset;
    fp.deblock_on = p.deblock_on; fp.alf_on = p.alf_on;
    xgpu_dbatch *db = nullptr;
    int rc = xgpu_batch_create(d->g, &p.batch, &db);
    if (rc >= 0) rc = xgpu_frame_begin(d->g, &fp);
    if (rc >= 0) rc = xgpu_batch_recon(d->g, db);
    if (rc >= 0 && p.deblock_on) rc = xgpu_deblock(d->g);
    if (rc >= 0 && p.alf_on) rc = xgpu_alf(d->g, &p.alf);
    if (rc >= 0) rc = xgpu_pad(d->g);
    if (rc >= 0) rc = xgpu_frame_end(d->g);
    if (rc >= 0 && p.n_dmvr_sub > 0) {      // sps->tool_dmvr: this picture's refined vectors, for the temporal candidates of the pictures to come
        std::vector<int16_t> mv((size_t)p.n_dmvr_sub * 4);
        rc = xgpu_batch_dmvr_mvs(d->g, db, mv.data(), p.n_dmvr_sub);
        if (rc == p.n_dmvr_sub) rc = xhost_parser_set_dmvr_mvs(d->ps, mv.data(), p.n_dmvr_sub);
        else if (rc >= 0) rc = XGPU_ERR_UNEXPECTED;
    }
    if (rc >= 0 && p.needs_ref_luma) {      // tool_dmvr with tool_hmvp / tool_mmvd: the parser refines vectors itself and reads this picture's decoded luma for it
        std::vector<int16_t> &buf = d->ref_luma[cur];
        const int stride = p.width + 2 * XGPU_PAD_L;
        buf.resize((size_t)stride * (p.height + 2 * XGPU_PAD_L));
        rc = xgpu_pic_download_padded(d->g, cur, buf.data(), nullptr, nullptr);
        if (rc >= 0) rc = xhost_parser_set_ref_luma(d->ps, p.poc, buf.data() + (size_t)XGPU_PAD_L * stride + XGPU_PAD_L, stride);
    }
    if (db) xgpu_batch_destroy(d->g, db);
    if (rc < 0) return rc;

    Image *im = new Image();
    const int w = p.width, h = p.height;
    im->mem.resize((size_t)w * h * 3 / 2);
    im->epoch = d->epoch; im->poc = p.poc; im->ready = false;
    XEVD_IMGB &g = im->img;
    memset(&g, 0, sizeof(g));
    g.cs = XEVD_CS_SET(XEVD_CF_YCBCR420, p.bit_depth_luma, 0);
    g.np = 3;
    int16_t *pl[3] = { im->mem.data(), im->mem.data() + (size_t)w * h, im->mem.data() + (size_t)w * h * 5 / 4 };
    for (int c = 0; c < 3; c++) {
        g.w[c] = g.aw[c] = c ? w / 2 : w; g.h[c] = g.ah[c] = c ? h / 2 : h;
        g.s[c] = g.w[c] * 2; g.e[c] = g.h[c];
        g.a[c] = g.baddr[c] = pl[c]; g.bsize[c] = g.s[c] * g.h[c];
    }
    g.refcnt = 1; g.addref = img_addref; g.getref = img_getref; g.release = img_release;
    g.pdata[0] = im;
    if (p.crop[0] | p.crop[1] | p.crop[2] | p.crop[3]) { g.crop_idx = 1; g.crop_l = p.crop[0]; g.crop_r = p.crop[1]; g.crop_t = p.crop[2]; g.crop_b = p.crop[3]; }
    g.imgb_active_aps_id = -1;
    if (p.dra_lut[0] && p.bit_depth_luma > 8) {
        // xevd_pull_frm hands out a DRA-mapped COPY of the picture (src_main/xevdm.c:3376-3383): the output kernel applies the tables and packs the
        // planes exactly in this image's layout (16-bit samples, tight rows)
        const xgpu_dra_luts dra = { p.dra_lut[0], { p.dra_lut[1], p.dra_lut[2] } };
        rc = xgpu_pic_output(d->g, cur, &dra, p.bit_depth_luma, 0, 0, 0, 0, im->mem.data(), im->mem.size() * sizeof(int16_t));
    } else
        rc = xgpu_pic_download(d->g, cur, pl[0], w, pl[1], pl[2], w / 2);
    if (rc >= 0 && d->use_sig && d->md5_on_device) {
        // the signature of the picture the application gets - the DRA-mapped copy when there is one (src_main/xevdm.c:3256-3287) - made by three lanes of the device
        // instead of a host pass over the planes (k_md5.hip; slower than a host core per picture, but it is not the host's time)
        const xgpu_dra_luts dra = { p.dra_lut[0], { p.dra_lut[1], p.dra_lut[2] } };
        rc = xgpu_pic_md5(d->g, cur, (p.dra_lut[0] && p.bit_depth_luma > 8) ? &dra : nullptr, im->dig);
        im->have_dig = rc >= 0;
    }
    if (rc < 0) { delete im; return rc; }

    for (int r = 0; r < p.n_release; r++)
        for (Slot &k : d->dpb) if (k.in_use && k.poc == p.release_poc[r]) { d->free_pic.push_back(k.pic); k.in_use = 0; }
    if (p.is_ref) { for (Slot &k : d->dpb) if (!k.in_use) { k.in_use = 1; k.poc = p.poc; k.pic = cur; guard.keep = true; break; } }
    *out = im;
    return XEVD_OK;
}

}   // namespace

extern "C" {

XEVD xevd_create(XEVD_CDSC *cdsc, int *err)
{
    (void)cdsc;                                  // cdsc->threads: the reference's CPU thread pool; the pictures are reconstructed on the GPU
    Decoder *d = new Decoder();
    d->ps = xhost_parser_open_nal();
    if (err) *err = XEVD_OK;
    return (XEVD)d;
}

void xevd_delete(XEVD id)
{
    Decoder *d = (Decoder *)id;
    if (!d) return;
    for (Image *im : d->pending) im->img.release(&im->img);
    if (d->g) xgpu_close(d->g);
    xhost_parser_close(d->ps);
    delete d;
}

int xevd_config(XEVD id, int cfg, void *buf, int *size)
{
    Decoder *d = (Decoder *)id;
    if (!d || !buf || !size || *size < 4) return XEVD_ERR_INVALID_ARGUMENT;
    int *v = (int *)buf;
    switch (cfg) {
    case XEVD_CFG_SET_USE_PIC_SIGNATURE: d->use_sig = *v != 0; return XEVD_OK;
    case XEVD_CFG_GET_CODEC_BIT_DEPTH: *v = d->bd; return XEVD_OK;
    case XEVD_CFG_GET_WIDTH: case XEVD_CFG_GET_CODED_WIDTH: *v = d->w; return XEVD_OK;
    case XEVD_CFG_GET_HEIGHT: case XEVD_CFG_GET_CODED_HEIGHT: *v = d->h; return XEVD_OK;
    case XEVD_CFG_GET_COLOR_SPACE: *v = XEVD_CF_YCBCR420; return XEVD_OK;
    default: return XEVD_ERR_UNSUPPORTED;
    }
}

int xevd_info(void *bits, int bits_size, int is_annexb, XEVD_INFO *info)
{
    // what is present is reported, the rest stays -1 (src_base/xevd_util.c:1693-1728): the application first asks with the 4 length bytes only
    const uint8_t *p = (const uint8_t *)bits;
    if (!p || !info) return XEVD_ERR_INVALID_ARGUMENT;
    info->nalu_len = info->nalu_type = info->nalu_tid = -1;
    if (is_annexb && bits_size >= 4) {
        info->nalu_len = (int)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]);
        p += 4; bits_size -= 4;
    }
    if (bits_size >= 2) {
        if (p[0] & 0x80) return XEVD_ERR_MALFORMED_BITSTREAM;
        info->nalu_type = (p[0] >> 1) & 0x3F;             // the coded value (type + 1), as the reference reports it
        info->nalu_tid = ((p[0] & 1) << 2) | (p[1] >> 6);
    }
    return XEVD_OK;
}

int xevd_decode(XEVD id, XEVD_BITB *bitb, XEVD_STAT *stat)
{
    Decoder *d = (Decoder *)id;
    if (!d || !bitb || !bitb->addr || bitb->ssize < 2) return XEVD_ERR_INVALID_ARGUMENT;
    const uint8_t *nal = (const uint8_t *)bitb->addr;
    const int nut = ((nal[0] >> 1) & 0x3F) - 1, tid = ((nal[0] & 1) << 2) | (nal[1] >> 6);
    if (stat) { stat->nalu_type = nut; stat->stype = 0; stat->fnum = -1; stat->read += bitb->ssize; }
    d->decoded_since_pull = true;

    if (nut == XEVD_NUT_SEI) {                       // payload type 0x10, size 16, then 16 bytes per plane (src_base/xevd_eco.c:1617-1678)
        if (bitb->ssize >= 2 + 2 + 48 && nal[2] == 0x10 && nal[3] == 16 && d->last) {
            if (!d->use_sig) return XEVD_WARN_CRC_IGNORED;
            const XEVD_IMGB &g = d->last->img;
            for (int c = 0; c < 3; c++) {
                Md5 m;
                uint8_t dig[16];
                if (d->last->have_dig) memcpy(dig, d->last->dig[c], 16);
                else { m.update((const uint8_t *)g.a[c], (size_t)g.s[c] * g.h[c]); m.finish(dig); }
                if (memcmp(dig, nal + 4 + 16 * c, 16) != 0) return XEVD_ERR_BAD_CRC;
            }
        }
        return XEVD_OK;
    }

    xhost_picture p;
    const int rc = xhost_parser_nal(d->ps, nal, (size_t)bitb->ssize, &p);
    if (rc < 0) return map_err(rc);
    if (rc == 0) return XEVD_OK;

    Image *im = nullptr;
    const int r2 = decode_picture(d, p, &im);
    if (r2 < 0) return r2;
    im->img.ts[XEVD_TS_DTS] = bitb->ts[XEVD_TS_DTS]; im->img.ts[XEVD_TS_PTS] = bitb->ts[XEVD_TS_PTS];
    d->pending.push_back(im);
    d->last = im;
    if (p.temporal_id == 0) {                        // everything up to the previous layer-0 picture (or of an earlier IDR period) can leave now
        for (Image *q : d->pending)
            if (q != im && (q->epoch < d->epoch || q->poc <= d->last_key_poc)) q->ready = true;
        d->last_key_poc = p.poc;
    }
    if (stat) {
        stat->fnum = d->pic_cnt++; stat->stype = p.slice_type; stat->poc = p.poc; stat->tid = tid;
        for (int l = 0; l < 2; l++) {
            stat->refpic_num[l] = (unsigned char)p.num_refp[l];
            for (int i = 0; i < p.num_refp[l] && i < 16; i++) stat->refpic[l][i] = p.refp_poc[i][l];
        }
    }
    return XEVD_OK;
}

int xevd_pull(XEVD id, XEVD_IMGB **img)
{
    Decoder *d = (Decoder *)id;
    if (!d || !img) return XEVD_ERR_INVALID_ARGUMENT;
    *img = nullptr;
    const bool bumping = !d->decoded_since_pull;     // a second pull without a decode in between: the application drains the decoder
    d->decoded_since_pull = false;
    int best = -1;
    for (size_t k = 0; k < d->pending.size(); k++) {
        const Image *q = d->pending[k];
        if (!bumping && !q->ready) continue;
        if (best < 0 || q->epoch < d->pending[best]->epoch || (q->epoch == d->pending[best]->epoch && q->poc < d->pending[best]->poc)) best = (int)k;
    }
    if (best < 0) return bumping ? XEVD_ERR_UNEXPECTED : XEVD_OK_OUT_NOT_AVAILABLE;
    Image *im = d->pending[best];
    d->pending.erase(d->pending.begin() + best);
    if (d->last == im) d->last = nullptr;            // its buffer now belongs to the caller
    *img = &im->img;
    return XEVD_OK;
}

}   // extern "C"
