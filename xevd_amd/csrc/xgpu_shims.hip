// xgpu_shims.hip - the fine-grained test shims of include/xevd_hip.h (xgpu_test_*): one block per call through the kernels' device functions, with the reference's
// per-block function-table call shapes.
#include "xgpu_host.h"

// ------------------------------------------------------------------------------------------------ test shims
static int test_mc(xgpu_ctx *c, const int16_t *plane, int pw, int ph, int ref_x, int ref_y, int has_dx, int has_dy,
                   int gmv_x, int gmv_y, int16_t *pred, int w, int h, int bd, int luma)
{
    ARGCHK(c, c != NULL); ARGCHK(c, plane && pred && pw > 0 && ph > 0 && w > 0 && h > 0);
    ARGCHK(c, luma ? ((w & 3) == 0 && (h & 3) == 0) : ((w & 1) == 0 && (h & 1) == 0));
    int16_t *dp = NULL, *dq = NULL;
    const size_t pb = sizeof(int16_t) * (size_t)pw * ph + 64, qb = sizeof(int16_t) * (size_t)w * h;
    HIPCHK(c, hipMalloc((void **)&dp, pb));
    HIPCHK(c, hipMalloc((void **)&dq, qb));
    HIPCHK(c, hipMemcpyAsync(dp, plane, pb - 64, hipMemcpyHostToDevice, c->stream));
    launch_test_mc(c, dp, pw, ref_x, ref_y, has_dx, has_dy, gmv_x, gmv_y, dq, w, h, bd, luma);
    HIPCHK(c, hipMemcpyAsync(pred, dq, qb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(dp); (void)hipFree(dq);
    return XGPU_OK;
}
int xgpu_test_mc_l(xgpu_ctx *c, const int16_t *plane, int pw, int ph, int ref_x, int ref_y, int has_dx, int has_dy,
                   int gmv_x, int gmv_y, int16_t *pred, int w, int h, int bd)
{ return test_mc(c, plane, pw, ph, ref_x, ref_y, has_dx, has_dy, gmv_x, gmv_y, pred, w, h, bd, 1); }
int xgpu_test_mc_c(xgpu_ctx *c, const int16_t *plane, int pw, int ph, int ref_x, int ref_y, int has_dx, int has_dy,
                   int gmv_x, int gmv_y, int16_t *pred, int w, int h, int bd)
{ return test_mc(c, plane, pw, ph, ref_x, ref_y, has_dx, has_dy, gmv_x, gmv_y, pred, w, h, bd, 0); }

int xgpu_test_recon(xgpu_ctx *c, const int16_t *coef, const int16_t *pred, int is_coef, int cuw, int cuh, int s_rec, int16_t *rec, int bit_depth)
{
    ARGCHK(c, c != NULL); ARGCHK(c, coef && pred && rec && cuw >= 2 && cuh >= 1 && !(cuw & 1) && cuw <= 128 && cuh <= 128 && s_rec >= cuw && bit_depth >= 8 && bit_depth <= 12);
    const size_t nb = sizeof(int16_t) * (size_t)cuw * cuh, rb = sizeof(int16_t) * (size_t)s_rec * cuh;
    int16_t *d = NULL;
    HIPCHK(c, hipMalloc((void **)&d, 2 * nb + rb));
    int16_t *dc = d, *dp = d + (size_t)cuw * cuh, *dr = dp + (size_t)cuw * cuh;
    HIPCHK(c, hipMemcpyAsync(dc, coef, nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dp, pred, nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dr, rec, rb, hipMemcpyHostToDevice, c->stream));
    launch_test_recon(c, dc, dp, is_coef, cuw, cuh, dr, s_rec, bit_depth);
    HIPCHK(c, hipMemcpyAsync(rec, dr, rb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d);
    return XGPU_OK;
}
// plane(s): pw x ph samples, host, in and out; (x, y) = the first sample on the far side of the edge
static int test_dbk(xgpu_ctx *c, int16_t *plane, int16_t *plane_v, int pw, int ph, int x, int y, int st, int st_v, int hor, int bd, int chroma)
{
    ARGCHK(c, c != NULL); ARGCHK(c, plane && (!chroma || plane_v) && pw > 0 && ph > 0 && bd >= 8 && bd <= 12 && st >= 0 && st_v >= 0);
    const int len = chroma ? 2 : 4;
    ARGCHK(c, hor ? (y >= 2 && y + 2 <= ph && x >= 0 && x + len <= pw) : (x >= 2 && x + 2 <= pw && y >= 0 && y + len <= ph));
    const size_t nb = sizeof(int16_t) * (size_t)pw * ph;
    int16_t *d = NULL;
    HIPCHK(c, hipMalloc((void **)&d, 2 * nb));
    HIPCHK(c, hipMemcpyAsync(d, plane, nb, hipMemcpyHostToDevice, c->stream));
    if (chroma) HIPCHK(c, hipMemcpyAsync(d + (size_t)pw * ph, plane_v, nb, hipMemcpyHostToDevice, c->stream));
    launch_test_dbk(c, d + (size_t)y * pw + x, d + (size_t)pw * ph + (size_t)y * pw + x, st, st_v, pw, bd, hor, chroma);
    HIPCHK(c, hipMemcpyAsync(plane, d, nb, hipMemcpyDeviceToHost, c->stream));
    if (chroma) HIPCHK(c, hipMemcpyAsync(plane_v, d + (size_t)pw * ph, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d);
    return XGPU_OK;
}
int xgpu_test_dbk(xgpu_ctx *c, int16_t *plane, int pw, int ph, int x, int y, int st, int hor, int bit_depth)
{ return test_dbk(c, plane, NULL, pw, ph, x, y, st, 0, hor, bit_depth, 0); }
int xgpu_test_dbk_chroma(xgpu_ctx *c, int16_t *u, int16_t *v, int pw, int ph, int x, int y, int st_u, int st_v, int hor, int bit_depth)
{ return test_dbk(c, u, v, pw, ph, x, y, st_u, st_v, hor, bit_depth, 1); }

int xgpu_test_batch_resid(xgpu_ctx *c, xgpu_dbatch *db, int16_t *resid)
{
    ARGCHK(c, c != NULL); ARGCHK(c, db != NULL && resid != NULL);
    if (db->n_coef) HIPCHK(c, hipMemcpyAsync(resid, db->d_resid, sizeof(int16_t) * db->n_coef, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return XGPU_OK;
}

int xgpu_test_itdq(xgpu_ctx *c, int16_t *coef, int n_blocks, int log2w, int log2h, const uint8_t *qp, int bit_depth)
{
    ARGCHK(c, c != NULL); ARGCHK(c, coef && qp && n_blocks > 0 && log2w >= 1 && log2w <= 6 && log2h >= 1 && log2h <= 6);
    const size_t per = (size_t)1 << (log2w + log2h), nb = per * n_blocks * sizeof(int16_t);
    std::vector<TbRec> tbs(n_blocks);
    std::vector<TbWave> wv;
    for (int i = 0; i < n_blocks; i++) { tbs[i].off = (uint32_t)(per * i); tbs[i].log2w = (uint8_t)log2w; tbs[i].log2h = (uint8_t)log2h; tbs[i].qp = qp[i]; tbs[i].log2s = (uint8_t)log2w; }
    const int pw = itdq_group_size(log2w, log2h);
    for (int f = 0; f < n_blocks; f += pw) wv.push_back({ (uint32_t)f, (uint16_t)std::min(pw, n_blocks - f), (uint8_t)log2w, (uint8_t)log2h, 0, 0, { 0, 0 } });
    int16_t *dc = NULL, *dr = NULL; TbRec *dt = NULL; TbWave *dw = NULL;
    HIPCHK(c, hipMalloc((void **)&dc, nb)); HIPCHK(c, hipMalloc((void **)&dr, nb));
    HIPCHK(c, hipMalloc((void **)&dt, sizeof(TbRec) * tbs.size())); HIPCHK(c, hipMalloc((void **)&dw, sizeof(TbWave) * wv.size()));
    HIPCHK(c, hipMemcpyAsync(dc, coef, nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dt, tbs.data(), sizeof(TbRec) * tbs.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dw, wv.data(), sizeof(TbWave) * wv.size(), hipMemcpyHostToDevice, c->stream));
    ItdqArgs ia; ia.coef = dc; ia.resid = dr; ia.tbs = dt; ia.waves = dw; ia.n_waves = (int)wv.size(); ia.bd = bit_depth; ia.iqt = c->sp.tool_iqt;
    launch_itdq(c, ia, c->stream);
    HIPCHK(c, hipMemcpyAsync(coef, dr, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(dc); (void)hipFree(dr); (void)hipFree(dt); (void)hipFree(dw);
    return XGPU_OK;
}

