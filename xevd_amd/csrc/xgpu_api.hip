// xgpu_api.hip - the C ABI of include/xevd_hip.h, part 1: context, device pictures, output, frame begin / end, HIP-event kernel timing.
// Host-side code only; kernels live in k_*.hip, the batch builder in xgpu_builder.hip, the launch sequencing in xgpu_launch.hip, the test shims in xgpu_shims.hip.
#include "xgpu_host.h"

// xevd_tbl_qp_chroma_adjust_base (src_base/xevd_tbl.c:345-354): default Baseline chroma QP mapping.
// ... and xevd_tbl_qp_chroma_adjust_main (xevd_tbl.c:334-342): the default when sps->tool_iqt is on (xevdm.c:471-479)
static const int8_t k_chroma_qp_main[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 30, 31, 32, 33, 34, 35, 36, 37, 37, 38, 39, 40, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54 };
static const int8_t k_chroma_qp_base[58] = {
     0,  1,  2,  3,  4,  5,  6,  7,  8,  9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19,
    20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 29, 30, 31, 32, 32, 33, 33, 34, 34,
    35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 39, 39, 40, 40, 40, 41, 41, 41 };


const char *xgpu_version(void) { return "xevd_amd 0.1 (gfx950)"; }
const char *xgpu_last_error(const xgpu_ctx *c) { return c ? c->err : "null ctx"; }

// ------------------------------------------------------------------------------------------------ timing
void time_begin(xgpu_ctx *c, int k, hipEvent_t *a, hipEvent_t *b)
{
    if (!c->timing) return;
    auto get = [&]() {
        hipEvent_t e;
        if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    *a = get(); *b = get();
    (void)hipEventRecord(*a, c->stream);
    (void)k;
}
void time_end(xgpu_ctx *c, int k, hipEvent_t a, hipEvent_t b)
{
    if (!c->timing) return;
    (void)hipEventRecord(b, c->stream);
    c->ev_pending.push_back({ a, b, k });
}
int xgpu_timing_enable(xgpu_ctx *c, int on) { ARGCHK(c, c != NULL); c->timing = on; return XGPU_OK; }
int xgpu_timing_reset(xgpu_ctx *c)
{
    ARGCHK(c, c != NULL);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto &e : c->ev_pending) { c->ev_pool.push_back(e.a); c->ev_pool.push_back(e.b); }
    c->ev_pending.clear();
    memset(c->t_ms, 0, sizeof(c->t_ms));
    memset(c->t_n, 0, sizeof(c->t_n));
    return XGPU_OK;
}
int xgpu_timing_get(xgpu_ctx *c, double ms[XGPU_K_COUNT], long long n[XGPU_K_COUNT])
{
    ARGCHK(c, c != NULL);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto &e : c->ev_pending) {
        float t = 0;
        HIPCHK(c, hipEventElapsedTime(&t, e.a, e.b));
        c->t_ms[e.k] += t; c->t_n[e.k]++;
        c->ev_pool.push_back(e.a); c->ev_pool.push_back(e.b);
    }
    c->ev_pending.clear();
    for (int i = 0; i < XGPU_K_COUNT; i++) { ms[i] = c->t_ms[i]; n[i] = c->t_n[i]; }
    return XGPU_OK;
}

// ------------------------------------------------------------------------------------------------ lifetime
static void init_transform_tables(xgpu_ctx *c)
{
    // xevd_tbl_tm2..64 (src_base/xevd_tbl.c:89-243) = round(64*sqrt(2)*cos((2n+1)k*pi/2N)), row 0 = 64; checked
    // entry by entry against the reference's tables in tests/test_oracle_vs_ref.py (same closed form as the oracle)
    static int tm[5460];
    int o = 0;
    for (int l = 1; l <= 6; l++) {
        const int N = 1 << l;
        for (int k = 0; k < N; k++)
            for (int n = 0; n < N; n++) {
                const double v = k == 0 ? 64.0 : 64.0 * sqrt(2.0) * cos((2 * n + 1) * k * 3.14159265358979323846 / (2.0 * N));
                tm[o++] = (int)(v >= 0 ? floor(v + 0.5) : -floor(-v + 0.5));
            }
    }
    // ATS matrices exactly as xevdm_init_multi_tbl builds them in double precision (src_main/xevdm_itdq.c:81-119); the
    // 4-point kernels of the reference are factorised around three entries of the first row (:163-190, :284-312) -
    // the equivalent 4x4 matrices are stored instead.  Order: [DST7, DCT8][4, 8, 16, 32], M[k][n].
    static int16_t ats[2 * 1360];
    for (int t = 0; t < 2; t++) {
        int q = t * 1360;
        for (int l = 2; l <= 5; l++) {
            const int N = 1 << l;
            const double sc = sqrt((double)N) * 64;
            int16_t m[32 * 32];
            for (int k = 0; k < N; k++)
                for (int n = 0; n < N; n++) {
                    const double v = t == 0 ? sin(3.14159265358979323846 * (k + 0.5) * (n + 1) / (N + 0.5)) * sqrt(2.0 / (N + 0.5))
                                            : cos(3.14159265358979323846 * (k + 0.5) * (n + 0.5) / (N + 0.5)) * sqrt(2.0 / (N + 0.5));
                    m[k * N + n] = (int16_t)(sc * v + (v > 0 ? 0.5 : -0.5));
                }
            if (N == 4) {
                const int a = m[0], b = m[1], cc = m[2], d = m[3];
                const int e7[16] = { a, b, cc, b + a,  cc, cc, 0, -cc,  a + b, -a, -cc, b,  b, -(b + a), cc, -a };
                const int e8[16] = { d + cc, b, cc, d,  b, 0, -b, -b,  cc, -b, -d, d + cc,  d, -b, d + cc, -cc };
                for (int i = 0; i < 16; i++) m[i] = (int16_t)(t == 0 ? e7[i] : e8[i]);
            }
            for (int i = 0; i < N * N; i++) ats[q++] = m[i];
        }
    }
    upload_transform_tables(tm, ats, c->stream);
}

int xgpu_open(const xgpu_seq_params *sp, xgpu_ctx **out)
{
    if (!sp || !out) return XGPU_ERR_INVALID_ARGUMENT;
    *out = NULL;
    if (sp->chroma_format_idc != 1) return XGPU_ERR_UNSUPPORTED;
    if (sp->width <= 0 || sp->height <= 0 || (sp->width & 7) || (sp->height & 7)) return XGPU_ERR_INVALID_ARGUMENT;
    if (sp->bit_depth_luma < 8 || sp->bit_depth_luma > 12 || sp->bit_depth_chroma < 8 || sp->bit_depth_chroma > 12) return XGPU_ERR_UNSUPPORTED;
    if (sp->log2_ctu < 5 || sp->log2_ctu > 7) return XGPU_ERR_INVALID_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || sp->device < 0 || sp->device >= ndev) return XGPU_ERR_UNEXPECTED;

    xgpu_ctx *c = new xgpu_ctx();
    c->sp = *sp;
    c->sp.chroma_qp_table[0] = c->sp.chroma_qp_table[1] = NULL;
    c->builder_threads = 1;
    c->err[0] = 0; c->timing = 0; c->have_frame = 0; c->d_maps = NULL; c->d_dra = NULL; c->d_ctb_flag = NULL; c->stream = 0; c->up_stream = 0; c->down_stream = 0; c->side_stream = 0; c->after_inter = 0; c->have_after_inter = 0; c->where = 0; c->addb_pending = 0;
    c->fork_ev = c->join_ev = 0;
    c->intra_small_min = getenv("XEVD_HIP_INTRA_SMALL_MIN") ? std::max(1, atoi(getenv("XEVD_HIP_INTRA_SMALL_MIN"))) : 2048;      // (k_intra.hip: launch_intra; read per context, tests set 1)
    c->addb_scalar = getenv("XEVD_HIP_ADDB_SCALAR") != NULL;
    c->split_addb_alf = getenv("XEVD_HIP_SPLIT_ADDB_ALF") != NULL;      // measurement knob: ADDB and ALF as two kernels (the round-2 chain) instead of k_addb_alf
    for (int i = 0; i < 2; i++) { c->d_out[i] = NULL; c->out_caps[i] = 0; c->out_ready[i] = c->out_done[i] = 0; c->out_busy[i] = 0; }
    c->d_md5 = NULL; c->md5_ready = 0;
    c->out_next = 0;
    memset(c->t_ms, 0, sizeof(c->t_ms)); memset(c->t_n, 0, sizeof(c->t_n));
    // chroma QP mapping: caller table starts at qp = -6*(bdc-8); default = Baseline static table with the
    // identity extension below 0 (xevd_set_chroma_qp_tbl_loc, xevd_tbl.c:364-372)
    const int boff = 6 * (sp->bit_depth_chroma - 8);
    for (int t = 0; t < 2; t++)
        for (int q = -boff; q <= 57; q++)
            c->chroma_qp[t][q + boff] = sp->chroma_qp_table[t] ? sp->chroma_qp_table[t][q + boff] : (int8_t)(q < 0 ? q : (sp->tool_iqt ? k_chroma_qp_main[q] : k_chroma_qp_base[q]));

    auto fail = [&](int code) { xgpu_close(c); return code; };
    if (hipSetDevice(sp->device) != hipSuccess) return fail(XGPU_ERR_UNEXPECTED);
    // Host waits (hipEventSynchronize / hipStreamSynchronize) spin by default: a decoder thread that waits for a picture's download burns a whole CPU doing so, and on
    // a host with a CPU quota (the GPU boxes of this pool: cgroup cpu.max = 16 CPUs under 256 hardware threads) the spinning of several workers throttles the
    // parser threads next to them.  XEVD_HIP_BLOCKING_SYNC=1: the runtime sleeps on an interrupt instead (a few microseconds more latency per wait).
    static const bool blocking = getenv("XEVD_HIP_BLOCKING_SYNC") != NULL && atoi(getenv("XEVD_HIP_BLOCKING_SYNC")) != 0;
    if (blocking) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    const unsigned ev_flags = hipEventDisableTiming | (blocking ? hipEventBlockingSync : 0);
    {
        // The runtime hands its hardware queues (four by default, GPU_MAX_HW_QUEUES) to the streams of a process in the order they are created.  Every context creates
        // four streams, so created in one fixed order the KERNEL streams of all contexts of a process shared one hardware queue, and the pictures of independent decoders ran
        // strictly one after the other (two contexts in one process: 2814 pictures/s at 8K against 2841 for one; 3087 - 3118 with the queues apart, 4K 7865 -> 11 300).  Every
        // other context creates its kernel stream LAST instead of first: the kernel streams alternate between the first and the fourth queue, the upload and download streams
        // of all contexts stay on the second and third - a picture's output copy never queues behind another decoder's kernels (rotating all four positions cost the host-bound
        // many-stream decode 4 %; stream priority classes changed nothing) -, and the rarely used side stream shares with the other parity's kernels.
        // XEVD_HIP_NO_QUEUE_ROTATION=1: the fixed order (A/B measurements).
        static std::atomic<int> n_ctx(0);
        static const bool no_rot = getenv("XEVD_HIP_NO_QUEUE_ROTATION") != NULL;
        const bool swap = !no_rot && (n_ctx.fetch_add(1) & 1);
        hipStream_t *const order[4] = { swap ? &c->side_stream : &c->stream, &c->up_stream, &c->down_stream, swap ? &c->stream : &c->side_stream };
        for (int i = 0; i < 4; i++)
            if (hipStreamCreateWithFlags(order[i], hipStreamNonBlocking) != hipSuccess) return fail(XGPU_ERR_UNEXPECTED);
    }
    if (hipEventCreateWithFlags(&c->after_inter, hipEventDisableTiming) != hipSuccess) return fail(XGPU_ERR_UNEXPECTED);
    if (hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming) != hipSuccess) return fail(XGPU_ERR_UNEXPECTED);
    for (int i = 0; i < 2; i++)
        if (hipEventCreateWithFlags(&c->out_ready[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->out_done[i], ev_flags) != hipSuccess)
            return fail(XGPU_ERR_UNEXPECTED);

    c->w_scu = sp->width >> 2; c->h_scu = sp->height >> 2;
    const int ctu = 1 << sp->log2_ctu;
    c->w_ctu = (sp->width + ctu - 1) / ctu; c->h_ctu = (sp->height + ctu - 1) / ctu;
    c->s_l = align_up(XGPU_MARGIN_L + sp->width + XGPU_PAD_L, 64);
    c->s_c = align_up(XGPU_MARGIN_C + (sp->width >> 1) + XGPU_PAD_C, 64);
    c->rows_l = sp->height + 2 * XGPU_PAD_L;
    c->rows_c = (sp->height >> 1) + 2 * XGPU_PAD_C;
    c->off_u = (size_t)c->s_l * c->rows_l;
    c->off_v = c->off_u + (size_t)c->s_c * c->rows_c;
    c->pic_elems = c->off_v + (size_t)c->s_c * c->rows_c + 64;   // +64: slack for the 16-byte window over-read of the last row

    if (hipMalloc((void **)&c->d_maps, sizeof(ScuRec) * (size_t)c->w_scu * c->h_scu) != hipSuccess) return fail(XGPU_ERR_OUT_OF_MEMORY);
    if (hipMemsetAsync(c->d_maps, 0, sizeof(ScuRec) * (size_t)c->w_scu * c->h_scu, c->stream) != hipSuccess) return fail(XGPU_ERR_UNEXPECTED);
    if (hipMalloc((void **)&c->d_ctb_flag, (size_t)c->w_ctu * c->h_ctu + 16) != hipSuccess) return fail(XGPU_ERR_OUT_OF_MEMORY);
    init_transform_tables(c);
    // slot 0 of `pics` is the private scratch picture of the deblocking passes
    c->pics.resize(1 + std::max(1, std::min(sp->max_pics, 34)));
    for (auto &p : c->pics) { p.base = NULL; p.used = 0; }
    {
        DevPic &p = c->pics[0];
        if (hipMalloc((void **)&p.base, c->pic_elems * sizeof(int16_t)) != hipSuccess) return fail(XGPU_ERR_OUT_OF_MEMORY);
        (void)hipMemsetAsync(p.base, 0, c->pic_elems * sizeof(int16_t), c->stream);
        p.s_l = c->s_l; p.s_c = c->s_c; p.used = 1;
        p.y = p.base + (size_t)XGPU_PAD_L * c->s_l + XGPU_MARGIN_L;
        p.u = p.base + c->off_u + (size_t)XGPU_PAD_C * c->s_c + XGPU_MARGIN_C;
        p.v = p.base + c->off_v + (size_t)XGPU_PAD_C * c->s_c + XGPU_MARGIN_C;
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(XGPU_ERR_UNEXPECTED);
    *out = c;
    return XGPU_OK;
}

void xgpu_close(xgpu_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->sp.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    if (c->down_stream) (void)hipStreamSynchronize(c->down_stream);
    if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
    for (auto &p : c->pics) if (p.base) (void)hipFree(p.base);
    if (c->d_maps) (void)hipFree(c->d_maps);
    for (BatchBlock &k : c->pool) { (void)hipFree(k.d_base); (void)hipHostFree(k.h_stage); (void)hipEventDestroy(k.uploaded); (void)hipEventDestroy(k.done); (void)hipEventDestroy(k.itdq_done); }
    for (auto &h : c->pinned) (void)hipHostFree(h.p);
    c->pinned.clear();
    c->pool.clear();
    if (c->d_md5) (void)hipFree(c->d_md5);
    if (c->md5_ready) (void)hipEventDestroy(c->md5_ready);
    for (int i = 0; i < 2; i++) { if (c->d_out[i]) (void)hipFree(c->d_out[i]); if (c->out_ready[i]) (void)hipEventDestroy(c->out_ready[i]); if (c->out_done[i]) (void)hipEventDestroy(c->out_done[i]); }
    if (c->d_dra) (void)hipFree(c->d_dra);
    if (c->d_ctb_flag) (void)hipFree(c->d_ctb_flag);
    for (auto &e : c->ev_pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->after_inter) (void)hipEventDestroy(c->after_inter);
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->join_ev) (void)hipEventDestroy(c->join_ev);
    delete c;
}

int xgpu_sync(xgpu_ctx *c)
{
    ARGCHK(c, c != NULL);
    HIPCHK(c, hipStreamSynchronize(c->up_stream));
    HIPCHK(c, hipStreamSynchronize(c->side_stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->down_stream));
    return XGPU_OK;
}

// pinned host memory for the arrays a batch points at: xgpu_batch_create sends a coefficient arena inside such a range straight from the caller's buffer
int xgpu_host_alloc(xgpu_ctx *c, size_t bytes, void **out)
{
    ARGCHK(c, c != NULL); ARGCHK(c, out != NULL && bytes > 0);
    *out = NULL;
    HIPCHK(c, hipSetDevice(c->sp.device));
    void *p = NULL;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { snprintf(c->err, sizeof(c->err), "host_alloc: cannot pin %zu bytes", bytes); return XGPU_ERR_OUT_OF_MEMORY; }
    std::lock_guard<std::mutex> g(c->pool_mu);
    c->pinned.push_back({ (uint8_t *)p, bytes });
    *out = p;
    return XGPU_OK;
}
void xgpu_host_free(xgpu_ctx *c, void *p)
{
    if (!c || !p) return;
    std::lock_guard<std::mutex> g(c->pool_mu);
    for (size_t i = 0; i < c->pinned.size(); i++)
        if (c->pinned[i].p == (uint8_t *)p) { (void)hipHostFree(p); c->pinned.erase(c->pinned.begin() + (long)i); return; }
}

// ------------------------------------------------------------------------------------------------ pictures

int xgpu_pic_alloc(xgpu_ctx *c)
{
    ARGCHK(c, c != NULL);
    HIPCHK(c, hipSetDevice(c->sp.device));
    for (size_t i = 1; i < c->pics.size(); i++) {
        DevPic &p = c->pics[i];
        if (p.used) continue;
        if (!p.base) {
            if (hipMalloc((void **)&p.base, c->pic_elems * sizeof(int16_t)) != hipSuccess) {
                snprintf(c->err, sizeof(c->err), "hipMalloc of a %zu-byte picture failed", c->pic_elems * sizeof(int16_t));
                return XGPU_ERR_OUT_OF_MEMORY;
            }
            HIPCHK(c, hipMemsetAsync(p.base, 0, c->pic_elems * sizeof(int16_t), c->stream));
        }
        p.s_l = c->s_l; p.s_c = c->s_c;
        p.y = p.base + (size_t)XGPU_PAD_L * c->s_l + XGPU_MARGIN_L;
        p.u = p.base + c->off_u + (size_t)XGPU_PAD_C * c->s_c + XGPU_MARGIN_C;
        p.v = p.base + c->off_v + (size_t)XGPU_PAD_C * c->s_c + XGPU_MARGIN_C;
        p.used = 1;
        return (int)i - 1;
    }
    snprintf(c->err, sizeof(c->err), "no free picture slot (max_pics=%d)", c->sp.max_pics);
    return XGPU_ERR_OUT_OF_MEMORY;
}

int xgpu_pic_free(xgpu_ctx *c, int pic)
{
    ARGCHK(c, c != NULL);
    ARGCHK(c, valid_pic(c, pic));
    dpic(c, pic).used = 0;          // memory is kept for reuse (a DPB recycles its buffers, xevd_picman.c)
    return XGPU_OK;
}

static int copy_planes(xgpu_ctx *c, int pic, int16_t *y, int s_y, int16_t *u, int16_t *v, int s_c, int ext_l, int ext_c, bool up)
{
    // ext_*: how many samples of padding around the active area take part (0 = active area only)
    DevPic &p = dpic(c, pic);
    int16_t *host[3] = { y, u, v };
    int16_t *dev[3] = { p.y, p.u, p.v };
    for (int i = 0; i < 3; i++) {
        if (!host[i]) continue;                           // a plane the caller did not ask for (xgpu_pic_download_padded with luma only)
        const int e = i ? ext_c : ext_l, hs = i ? s_c : s_y, ds = i ? p.s_c : p.s_l;
        const int w = (i ? c->sp.width >> 1 : c->sp.width) + 2 * e, h = (i ? c->sp.height >> 1 : c->sp.height) + 2 * e;
        int16_t *d = dev[i] - (size_t)e * ds - e;
        if (up) HIPCHK(c, hipMemcpy2DAsync(d, ds * 2, host[i], hs * 2, w * 2, h, hipMemcpyHostToDevice, c->stream));
        else    HIPCHK(c, hipMemcpy2DAsync(host[i], hs * 2, d, ds * 2, w * 2, h, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return XGPU_OK;
}

int xgpu_pic_upload(xgpu_ctx *c, int pic, const int16_t *y, int s_y, const int16_t *u, const int16_t *v, int s_c)
{
    ARGCHK(c, c != NULL); ARGCHK(c, valid_pic(c, pic)); ARGCHK(c, y && u && v && s_y >= c->sp.width && s_c >= c->sp.width / 2);
    return copy_planes(c, pic, (int16_t *)y, s_y, (int16_t *)u, (int16_t *)v, s_c, 0, 0, true);
}
int xgpu_pic_download(xgpu_ctx *c, int pic, int16_t *y, int s_y, int16_t *u, int16_t *v, int s_c)
{
    ARGCHK(c, c != NULL); ARGCHK(c, valid_pic(c, pic)); ARGCHK(c, y && u && v && s_y >= c->sp.width && s_c >= c->sp.width / 2);
    return copy_planes(c, pic, y, s_y, u, v, s_c, 0, 0, false);
}
int xgpu_pic_download_padded(xgpu_ctx *c, int pic, int16_t *by, int16_t *bu, int16_t *bv)
{
    ARGCHK(c, c != NULL); ARGCHK(c, valid_pic(c, pic)); ARGCHK(c, by && (bu != NULL) == (bv != NULL));
    return copy_planes(c, pic, by, c->sp.width + 2 * XGPU_PAD_L, bu, bv, (c->sp.width >> 1) + 2 * XGPU_PAD_C, XGPU_PAD_L, XGPU_PAD_C, false);
}
int xgpu_pic_upload_padded(xgpu_ctx *c, int pic, const int16_t *by, const int16_t *bu, const int16_t *bv)
{
    ARGCHK(c, c != NULL); ARGCHK(c, valid_pic(c, pic)); ARGCHK(c, by && bu && bv);
    return copy_planes(c, pic, (int16_t *)by, c->sp.width + 2 * XGPU_PAD_L, (int16_t *)bu, (int16_t *)bv,
                       (c->sp.width >> 1) + 2 * XGPU_PAD_C, XGPU_PAD_L, XGPU_PAD_C, true);
}

static bool valid_output(const xgpu_ctx *c, int out_bd, int cl, int cr, int ct, int cb)
{
    return out_bd >= 8 && out_bd <= 16 && cl >= 0 && cr >= 0 && ct >= 0 && cb >= 0 && !((cl | cr | ct | cb) & 1) &&
           cl + cr < c->sp.width && ct + cb < c->sp.height;
}
size_t xgpu_pic_output_size(const xgpu_ctx *c, int out_bit_depth, int crop_l, int crop_r, int crop_t, int crop_b)
{
    if (!c || !valid_output(c, out_bit_depth, crop_l, crop_r, crop_t, crop_b)) return 0;
    const size_t w = c->sp.width - crop_l - crop_r, h = c->sp.height - crop_t - crop_b;
    return (w * h + 2 * (w >> 1) * (h >> 1)) * (out_bit_depth == 8 ? 1 : 2);
}
static int upload_dra(xgpu_ctx *c, const xgpu_dra_luts *dra)      // the inverse-mapping tables behind the picture's kernels on their stream
{
    ARGCHK(c, dra->luma_inv_scale_lut && dra->chroma_inv_scale_lut[0] && dra->chroma_inv_scale_lut[1]);
    ARGCHK(c, c->sp.bit_depth_luma <= 10);                         // the tables have 1024 entries (DRA_LUT_MAXSIZE)
    if (!c->d_dra && hipMalloc((void **)&c->d_dra, sizeof(int32_t) * 3 * 1024) != hipSuccess) { snprintf(c->err, sizeof(c->err), "pic_output: cannot allocate the DRA tables"); return XGPU_ERR_OUT_OF_MEMORY; }
    const int32_t *src[3] = { dra->luma_inv_scale_lut, dra->chroma_inv_scale_lut[0], dra->chroma_inv_scale_lut[1] };
    for (int i = 0; i < 3; i++) HIPCHK(c, hipMemcpyAsync(c->d_dra + 1024 * i, src[i], sizeof(int32_t) * 1024, hipMemcpyHostToDevice, c->stream));
    return XGPU_OK;
}
int xgpu_pic_output_async(xgpu_ctx *c, int pic, const xgpu_dra_luts *dra, int out_bit_depth, int crop_l, int crop_r, int crop_t, int crop_b, void *dst, size_t dst_size,
                          int *ticket)
{
    ARGCHK(c, c != NULL); ARGCHK(c, valid_pic(c, pic)); ARGCHK(c, dst != NULL && ticket != NULL);
    ARGCHK(c, valid_output(c, out_bit_depth, crop_l, crop_r, crop_t, crop_b));
    const size_t need = xgpu_pic_output_size(c, out_bit_depth, crop_l, crop_r, crop_t, crop_b);
    ARGCHK(c, dst_size >= need);
    const int k = c->out_next;
    if (c->out_busy[k]) { HIPCHK(c, hipEventSynchronize(c->out_done[k])); c->out_busy[k] = 0; }      // two outputs in flight at most
    if (c->out_caps[k] < need) {
        if (c->d_out[k]) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_out[k]); c->d_out[k] = NULL; c->out_caps[k] = 0; }
        if (hipMalloc((void **)&c->d_out[k], need) != hipSuccess) { snprintf(c->err, sizeof(c->err), "pic_output: cannot allocate the %zu-byte staging buffer", need); return XGPU_ERR_OUT_OF_MEMORY; }
        c->out_caps[k] = need;
    }
    if (dra) { const int rc = upload_dra(c, dra); if (rc < 0) return rc; }
    // conversion + packing behind the picture's kernels on their stream; the copy to the host on the download stream behind an event, so it
    // overlaps the kernels of the next picture
    launch_output(c, dpic(c, pic), dra ? c->d_dra : NULL, out_bit_depth, crop_l, crop_r, crop_t, crop_b, c->d_out[k]);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->out_ready[k], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->down_stream, c->out_ready[k], 0));
    HIPCHK(c, hipMemcpyAsync(dst, c->d_out[k], need, hipMemcpyDeviceToHost, c->down_stream));
    HIPCHK(c, hipEventRecord(c->out_done[k], c->down_stream));
    c->out_busy[k] = 1;
    c->out_next = k ^ 1;
    *ticket = k;
    return XGPU_OK;
}
int xgpu_pic_output_wait(xgpu_ctx *c, int ticket)
{
    ARGCHK(c, c != NULL); ARGCHK(c, ticket == 0 || ticket == 1);
    if (c->out_busy[ticket]) { HIPCHK(c, hipEventSynchronize(c->out_done[ticket])); c->out_busy[ticket] = 0; }
    return XGPU_OK;
}
int xgpu_pic_output(xgpu_ctx *c, int pic, const xgpu_dra_luts *dra, int out_bit_depth, int crop_l, int crop_r, int crop_t, int crop_b, void *dst, size_t dst_size)
{
    int ticket = 0;
    const int rc = xgpu_pic_output_async(c, pic, dra, out_bit_depth, crop_l, crop_r, crop_t, crop_b, dst, dst_size, &ticket);
    return rc < 0 ? rc : xgpu_pic_output_wait(c, ticket);
}

// The picture signature on the device (k_md5.hip): the planes packed as the signature's message behind the picture's kernels (k_output, samples as they are), the three
// chains on the transfer stream - the kernel stream is free for the next picture while they run -, 48 bytes to the host.  Blocking.
int xgpu_pic_md5(xgpu_ctx *c, int pic, const xgpu_dra_luts *dra, uint8_t digest[3][16])
{
    ARGCHK(c, c != NULL); ARGCHK(c, valid_pic(c, pic)); ARGCHK(c, digest != NULL);
    const int w = c->sp.width, h = c->sp.height;
    const size_t need = ((size_t)w * h + 2 * (size_t)(w >> 1) * (h >> 1)) * 2;
    if (!c->d_md5) {
        if (!c->md5_ready) HIPCHK(c, hipEventCreateWithFlags(&c->md5_ready, hipEventDisableTiming));      // the event first: d_md5 != NULL means both exist
        if (hipMalloc((void **)&c->d_md5, need + 64) != hipSuccess) { c->d_md5 = NULL; snprintf(c->err, sizeof(c->err), "pic_md5: cannot allocate the %zu-byte message buffer", need + 64); return XGPU_ERR_OUT_OF_MEMORY; }
    }
    if (dra) { const int rc = upload_dra(c, dra); if (rc < 0) return rc; }
    uint32_t *d_digest = (uint32_t *)(c->d_md5 + ((need + 15) & ~(size_t)15));
    launch_output(c, dpic(c, pic), dra ? c->d_dra : NULL, c->sp.bit_depth_luma, 0, 0, 0, 0, c->d_md5, true);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->md5_ready, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->down_stream, c->md5_ready, 0));
    launch_md5(c, c->down_stream, c->d_md5, w, h, d_digest);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(digest, d_digest, 48, hipMemcpyDeviceToHost, c->down_stream));
    HIPCHK(c, hipStreamSynchronize(c->down_stream));
    return XGPU_OK;
}

// ------------------------------------------------------------------------------------------------ per picture
// ADDB deblocking directly followed by ALF: xgpu_deblock only prepares its arguments and xgpu_alf launches k_addb_alf, which deblocks the 72 x 72 region around

int xgpu_frame_begin(xgpu_ctx *c, const xgpu_frame_params *fp)
{
    ARGCHK(c, c != NULL); ARGCHK(c, fp != NULL); ARGCHK(c, valid_pic(c, fp->pic));
    for (int l = 0; l < 2; l++) {
        ARGCHK(c, fp->num_refp[l] >= 0 && fp->num_refp[l] <= XGPU_MAX_REFS);
        for (int i = 0; i < fp->num_refp[l]; i++) ARGCHK(c, valid_pic(c, fp->refp_pic[i][l]));
    }
    c->fp = *fp;
    c->have_frame = 1;
    c->where = 0;
    c->pad_done = 0;
    c->addb_pending = 0;
    c->order_rl = 0;
    return XGPU_OK;
}

int xgpu_frame_end(xgpu_ctx *c)
{
    ARGCHK(c, c != NULL);
    c->have_frame = 0;
    if (c->where != 0 || c->addb_pending) {
        c->addb_pending = 0;
        snprintf(c->err, sizeof(c->err), "frame_end: the in-loop filters announced in xgpu_frame_params (deblock_on=%d alf_on=%d) were not all run",
                 c->fp.deblock_on, c->fp.alf_on);
        c->where = 0;
        return XGPU_ERR_UNEXPECTED;
    }
    return XGPU_OK;
}
