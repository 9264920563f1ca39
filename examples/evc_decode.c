/*
 * evc_decode.c - a complete decoder in plain C on the C ABIs of this repository (no Python, no reference code):
 *   include/xevd_host.h  bitstream -> per-picture CU batches + DPB bookkeeping by POC      (libxevd_host.so, host only)
 *   include/xevd_hip.h   CU batches -> pictures on the MI355X                              (libxevd_hip.so)
 *   include/xevd_wq.h    closed GOPs of all inputs -> the GPUs of the node                 (libxevd_host.so)
 * It is the loop xevd_app runs around xevd_decode / xevd_pull (app/xevd_app.c:455-640), with the pictures living in HBM: parse a picture,
 * map its reference POCs to device picture slots, reconstruct + filter + pad it, release the pictures the stream unmarked, write the
 * output in POC order inside every IDR period (what xevd_pull's bumping yields).  The output of picture k (conversion, packing, download)
 * overlaps the kernels of picture k + 1 (xgpu_pic_output_async).
 * With --gpus N every input is cut into closed GOPs (IDR to IDR: independent units) and the GOPs of all inputs go through one
 * host work queue to N worker threads, one per device, each with its own context and DPB; every GOP lands at its own offset of its output
 * file.  No collective, no RCCL: there is nothing to exchange.
 * --workers W puts W worker threads (each with its own parser, context and DPB) on every device: one stream's entropy decoding is a serial
 * chain on one host core (an 8K picture takes tens of milliseconds to parse, its kernels half a millisecond), so a GPU is fed by parsing
 * several GOPs at once.
 * --tile-threads T lets every worker's parser use T threads for the tiles of one picture (xhost_parser_set_threads).
 * Every worker is a pipeline of three stages: the parser thread (entropy decoding, picture k + 1 + N), N batch-builder threads (--builders; default 2 for one or two workers, 1 for more: xgpu_batch_create of
 * pictures k + 1 .. k + N side by side, picture j on thread j mod N) and the device thread (kernel launches and output of picture k); --no-pipeline: back to back on one thread, as xevd_dec_nalu does it.  A worker keeps its parser
 * (xhost_parser_rebind), its context, its pinned buffers and its threads from unit to unit.
 * usage: evc_decode [--gpus N] [--workers W] [--tile-threads T] [--build-threads B] [--builders N] [--keep-units K] [--hash-units] [--no-pipeline] [--trace] [--bd D] in.evc out.yuv [in2.evc out2.yuv ...]      (--bd 0 / omitted: the coding bit depth; 8: one byte per sample)
 *        evc_decode in.evc out.yuv D                                               (the round-1 form)
 * build: oracle-free; see examples/Makefile.
 */
#define _XOPEN_SOURCE 700
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <fcntl.h>
#include <pthread.h>
#include <malloc.h>
#include <sys/resource.h>
#include "xevd_host.h"
#include "xevd_wq.h"

#define MAX_SLOTS 33             /* the parser keeps at most 32 reference pictures + the current one */
#define MAX_STREAMS 64
#define MAX_GOPS 4096
#define CHECK(call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, w->g ? xgpu_last_error(w->g) : ""); return rc_; } } while (0)
static double now_s(void);
static int g_hash_units = 0;       /* --hash-units: a 64-bit hash of every unit's pictures (output order) in the --json line: units that are not kept (--keep-units) are checked all the same */
static uint64_t g_unit_hash[64][256];
static int g_keep_units = -1;      /* --keep-units K: only the first K units (closed GOPs) of every input are written to its output file; the rest is decoded all the same (long timing runs) */
static int g_builders = 0;          /* --builders N: builder threads per worker (pictures built side by side); 0 = not given: 2 for one or two workers, 1 for more (the host's CPUs are the workers' to share) */
static int g_depth = 3;             /* 2 + g_builders */
static int g_build_threads = 8;     /* --build-threads B: host threads xgpu_batch_create spreads its per-CU passes over */

#define MAX_BUILDERS 4
typedef struct { int poc, pic, in_use; } slot_t;                 /* DPB: POC -> device picture slot */
typedef struct { int epoch, poc; size_t off; } out_t;              /* one output picture: position in the unit's buffer */
typedef struct { const uint8_t *bytes; size_t size; int fd; int out_bd; } stream_t;
typedef struct {                                                    /* one worker = one device */
    int device;
    xgpu_ctx *g;
    xgpu_seq_params sp;
    int slots[MAX_SLOTS], n_slots;                                  /* device pictures allocated so far (on demand) */
    const stream_t *streams;
    long pictures;
    double busy_s, setup_s;                                         /* time inside the units / inside context creation and picture allocation */
    uint8_t *frames; size_t frames_cap; int frames_pinned;          /* the output pictures of a unit (pinned): kept across units - pinning gigabytes per GOP costs more than decoding it */
    void *arena[4]; size_t arena_bytes[4]; int arena_busy[4];     /* pinned coefficient arenas handed to the parsers of this worker's units (kept across units) */
    pthread_mutex_t arena_mu;
    int arena_ok;
    int ref_luma_pinned[MAX_SLOTS + 2];                             /* from xgpu_host_alloc: freed with the context */
    int16_t *ref_luma[MAX_SLOTS + 2];                               /* host copies of decoded luma planes, by device picture (only for streams whose parser asks) */
    double parse_s, build_s;                                        /* inside xhost_parser_next (its own thread with the pipeline) / inside xgpu_batch_create */
    xhost_parser *ps;                                               /* the worker's parser, rebound to every unit (xhost_parser_rebind): its memory and tile threads stay */
    /* the worker's builder threads live as long as the worker: xgpu_batch_create keeps its worker pool and its scratch (owner map, records, dependency plan: tens of
       megabytes at 8K) per calling thread, and threads created per unit threw both away at every GOP */
    pthread_t bth[MAX_BUILDERS];
    int n_bth;
    pthread_mutex_t b_mu;
    pthread_cond_t b_cv;
    void *b_unit;                                                   /* the pipe_t of the unit being decoded (NULL between units) */
    long b_gen;                                                     /* a new unit: the builders leave their wait */
    int b_left, b_quit;                                             /* builders still inside the unit; the worker goes away */
} worker_t;

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int cmp_out(const void *a, const void *b)
{
    const out_t *x = (const out_t *)a, *y = (const out_t *)b;
    return x->epoch != y->epoch ? x->epoch - y->epoch : x->poc - y->poc;
}

/* the worker's context fits the stream of this picture?  Else (re)open it: one context per sequence, reused by every unit of it */
static int worker_context(worker_t *w, const xhost_picture *p)
{
    xgpu_seq_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.device = w->device; sp.width = p->width; sp.height = p->height;
    sp.bit_depth_luma = p->bit_depth_luma; sp.bit_depth_chroma = p->bit_depth_chroma; sp.chroma_format_idc = 1; sp.log2_ctu = 6;
    sp.tool_iqt = p->tool_iqt; sp.tool_addb = p->tool_addb; sp.tool_alf = p->tool_alf; sp.tool_eipd = p->tool_eipd; sp.tool_admvp = p->tool_admvp;
    sp.max_pics = MAX_SLOTS + 1;
    if (w->g && !p->chroma_qp_table[0] && !memcmp(&sp, &w->sp, sizeof(sp))) return 0;
    const double t0 = now_s();
    pthread_mutex_lock(&w->arena_mu);                               /* the arenas belong to the context that goes away (a parser that still holds one only returns it) */
    if (w->g) { xgpu_close(w->g); w->g = NULL; }
    for (int i = 0; i < 4; i++) { w->arena[i] = NULL; w->arena_bytes[i] = 0; }
    pthread_mutex_unlock(&w->arena_mu);
    if (w->frames && !w->frames_pinned) free(w->frames);
    w->frames = NULL; w->frames_cap = 0;
    for (int i = 0; i < MAX_SLOTS + 2; i++) { if (!w->ref_luma_pinned[i]) free(w->ref_luma[i]); w->ref_luma[i] = NULL; w->ref_luma_pinned[i] = 0; }      /* sized for the old sequence (pinned ones went with the context) */
    w->sp = sp;                                                     /* compared without the table pointers */
    w->n_slots = 0;
    sp.chroma_qp_table[0] = p->chroma_qp_table[0]; sp.chroma_qp_table[1] = p->chroma_qp_table[1];
    xgpu_ctx *g = NULL;
    CHECK(xgpu_open(&sp, &g));
    if (g_build_threads > 1) (void)xgpu_set_builder_threads(g, g_build_threads > 64 ? 64 : g_build_threads);
    pthread_mutex_lock(&w->arena_mu);
    w->g = g;
    pthread_mutex_unlock(&w->arena_mu);
    w->setup_s += now_s() - t0;
    return 0;
}

/* xhost_parser_set_arena callbacks: pinned host memory of the worker's context, so that xgpu_batch_create sends a picture's coefficients from where the
   parser gathered them.  Called on the parser thread; the context appears when the first picture of the worker's first unit has been parsed - until
   then there is nothing to allocate from and the parser keeps that picture's coefficients in its own memory. */
static void *arena_alloc(void *user, size_t bytes)
{
    worker_t *w = (worker_t *)user;
    void *out = NULL;
    pthread_mutex_lock(&w->arena_mu);
    xgpu_ctx *g = w->arena_ok ? w->g : NULL;                        /* not before this unit's context is settled (worker_context may replace it) */
    for (int i = 0; i < 4 && !out && g; i++)
        if (w->arena[i] && !w->arena_busy[i] && w->arena_bytes[i] >= bytes) { w->arena_busy[i] = 1; out = w->arena[i]; }
    for (int i = 0; i < 4 && !out && g; i++)
        if (!w->arena_busy[i]) {
            if (w->arena[i]) { xgpu_host_free(g, w->arena[i]); w->arena[i] = NULL; }
            if (xgpu_host_alloc(g, bytes, &w->arena[i]) == 0) { w->arena_bytes[i] = bytes; w->arena_busy[i] = 1; out = w->arena[i]; }
            break;
        }
    pthread_mutex_unlock(&w->arena_mu);
    return out;
}
static void arena_release(void *user, void *mem)
{
    worker_t *w = (worker_t *)user;
    pthread_mutex_lock(&w->arena_mu);
    for (int i = 0; i < 4; i++) if (w->arena[i] == mem) w->arena_busy[i] = 0;
    pthread_mutex_unlock(&w->arena_mu);
}

static int g_tile_threads = 1;
static int g_trace = 0;             /* --trace: one line per picture on stderr (slice type, CUs, stage times) */
static int g_pipeline = 1;          /* --no-pipeline: parse and reconstruct every picture back to back on the worker's thread (the round-2 loop) */

/* ---- the parser of a unit on its own thread, one picture ahead of the thread that drives the device ------------------------------------------------
 * xevd_dec_nalu runs entropy decoding and reconstruction of a picture back to back (src_base/xevd.c:1905-1983).  Here they are two stages of a
 * pipeline: while this worker turns picture k into a device batch and launches its kernels, the parser thread is already inside xhost_parser_next
 * for picture k + 1 (xhost_parser_set_depth(2): the arrays of two pictures stay valid).  Pictures with DMVR candidates break the overlap for one
 * step: the parser needs their refined vectors from the device before it goes on (xhost_parser_set_dmvr_mvs). */
#define PIPE_DEPTH (2 + MAX_BUILDERS)  /* most pictures in flight between the parser, the builders and the device thread; in use: 2 + --builders (g_depth) */
typedef struct {
    xhost_parser *ps;
    xhost_picture pic[PIPE_DEPTH];
    int rc[PIPE_DEPTH];
    double parse_ms[PIPE_DEPTH];
    xgpu_dbatch *db[PIPE_DEPTH];     /* built by the builder thread */
    int build_rc[PIPE_DEPTH];
    double build_ms[PIPE_DEPTH];
    long built_k[PIPE_DEPTH];        /* k + 1 once picture k (in slot k % depth) has been built */
    int n_builders;
    long produced, released;         /* pictures handed over by the parser thread / given back by the consumer */
    int stop;
    xgpu_ctx *g;                     /* the worker's context once the device thread has settled it for this unit (the builder thread waits for it) */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    double parse_s, build_s;
} pipe_t;

static void *parser_thread(void *arg)
{
    pipe_t *q = (pipe_t *)arg;
    for (long k = 0;; k++) {
        const int s = (int)(k % g_depth);
        pthread_mutex_lock(&q->mu);
        while (!q->stop && q->released < k - (g_depth - 1)) pthread_cond_wait(&q->cv, &q->mu);      /* slot k % depth is free once picture k - depth has been given back */
        const int stop = q->stop;
        pthread_mutex_unlock(&q->mu);
        if (stop) break;
        const double t0 = now_s();
        const int rc = xhost_parser_next(q->ps, &q->pic[s]);
        q->parse_s += now_s() - t0;
        q->parse_ms[s] = 1e3 * (now_s() - t0);
        pthread_mutex_lock(&q->mu);
        q->rc[s] = rc;
        q->produced = k + 1;
        pthread_cond_broadcast(&q->cv);
        /* DMVR feedback: picture k's refined vectors reach the parser (from the consumer, while this thread waits here) before picture k + 1 is parsed.
           (When the parser refines itself - tool_dmvr with tool_hmvp / tool_mmvd, needs_ref_luma - it goes on: xhost_parser_next waits inside the first
           CU that searches in a plane the consumer has not registered yet, xhost_parser_set_ref_luma_wait.) */
        if (rc == 1 && q->pic[s].n_dmvr_sub > 0) while (!q->stop && q->released < k + 1) pthread_cond_wait(&q->cv, &q->mu);
        pthread_mutex_unlock(&q->mu);
        if (rc != 1) break;
    }
    return NULL;
}

/* The middle stage: picture k's device batch (xgpu_batch_create: records, transform-block lists, dependency plan, staging block, upload) is built here while the
   device thread launches picture k - 1 and the parser is inside picture k + 1.  xgpu_batch_create may run next to the thread that drives the context. */
typedef struct { void *w; int id; } builder_arg_t;
static void builder_unit(pipe_t *q, int id)
{
    /* --builders N: N of these threads, thread i on pictures i, i + N, ... - a picture's batch does not depend on the one before it, and at 8K the build
       (13 ms on 4 threads, largely its dependency plan) was the longest stage of the pipeline */
    for (long k = id;; k += q->n_builders) {
        const int s = (int)(k % g_depth);
        pthread_mutex_lock(&q->mu);
        while (!q->stop && q->produced <= k) pthread_cond_wait(&q->cv, &q->mu);
        const int prc = q->stop ? 0 : q->rc[s];
        while (!q->stop && prc == 1 && !q->g) pthread_cond_wait(&q->cv, &q->mu);       /* the device thread opens the context at the unit's first picture */
        xgpu_ctx *g = q->g;
        const int stop = q->stop;
        pthread_mutex_unlock(&q->mu);
        if (stop) break;
        int brc = 0;
        q->db[s] = NULL;
        if (prc == 1) {
            const double t0 = now_s();
            brc = xgpu_batch_create(g, &q->pic[s].batch, &q->db[s]);
            q->build_ms[s] = 1e3 * (now_s() - t0);
            q->build_s += now_s() - t0;
        }
        pthread_mutex_lock(&q->mu);
        q->build_rc[s] = brc;
        q->built_k[s] = k + 1;
        pthread_cond_broadcast(&q->cv);
        pthread_mutex_unlock(&q->mu);
        if (prc != 1 || brc < 0) break;
    }
}
static void *builder_thread(void *arg)
{
    worker_t *w = (worker_t *)((builder_arg_t *)arg)->w;
    const int id = ((builder_arg_t *)arg)->id;
    long seen = 0;
    free(arg);
    for (;;) {
        pthread_mutex_lock(&w->b_mu);
        while (!w->b_quit && w->b_gen == seen) pthread_cond_wait(&w->b_cv, &w->b_mu);
        if (w->b_quit) { pthread_mutex_unlock(&w->b_mu); break; }
        seen = w->b_gen;
        pipe_t *q = (pipe_t *)w->b_unit;
        pthread_mutex_unlock(&w->b_mu);
        if (q && id < q->n_builders) builder_unit(q, id);
        pthread_mutex_lock(&w->b_mu);
        if (--w->b_left == 0) pthread_cond_broadcast(&w->b_cv);
        pthread_mutex_unlock(&w->b_mu);
    }
    return NULL;
}

/* Decode `bytes` (parameter sets + one or more GOPs) on the worker's device -> packed pictures in output order */
static int decode_unit(worker_t *w, const uint8_t *bytes, size_t size, int out_bd_arg, int expected, uint8_t **frames_out, out_t **outs_out, int *n_out, size_t *frame_bytes_out, int *pinned_out)
{
    slot_t dpb[MAX_SLOTS];
    int free_pic[MAX_SLOTS], n_free = 0, n_pics = 0, epoch = -1, rc = 0, have_ctx = 0, ticket = -1, pinned = 0, thread_on = 0;
    uint8_t *frames = NULL;                                         /* decoded pictures in decoding order, packed: pinned memory, so that the */
    size_t frame_bytes = 0;                                         /* download of picture k runs while picture k + 1 is parsed and launched     */
    xgpu_dbatch *db = NULL;
    int16_t *mv = NULL;
    pthread_t th;
    int builder_on = 0;
    pipe_t q;
    memset(&q, 0, sizeof(q));
    memset(dpb, 0, sizeof(dpb));
    out_t *outs = (out_t *)malloc(sizeof(out_t) * (size_t)(expected > 0 ? expected : 1));
    pthread_mutex_init(&q.mu, NULL); pthread_cond_init(&q.cv, NULL);      /* before the first goto done, which destroys them */
    pthread_mutex_lock(&w->arena_mu); w->arena_ok = 0; pthread_mutex_unlock(&w->arena_mu);
    if (w->ps) { if (xhost_parser_rebind(w->ps, bytes, size) < 0) { rc = -1; goto done; } }
    else {
        w->ps = xhost_parser_open(bytes, size);
        if (w->ps) {
            if (g_tile_threads > 1) xhost_parser_set_threads(w->ps, g_tile_threads);      /* the tiles of a picture on parallel host threads */
            xhost_parser_set_arena(w->ps, arena_alloc, arena_release, w);
            if (g_pipeline) xhost_parser_set_depth(w->ps, g_depth);
            if (g_pipeline) xhost_parser_set_ref_luma_wait(w->ps, 1);    /* luma planes (needs_ref_luma) are registered by this thread while the parser thread runs ahead */
        }
    }
    q.ps = w->ps;
    if (!outs || !q.ps) { rc = -1; goto done; }
    if (g_pipeline) {
        if (pthread_create(&th, NULL, parser_thread, &q) != 0) { rc = -1; goto done; }
        thread_on = 1;
        q.n_builders = g_builders;
        for (; w->n_bth < g_builders; w->n_bth++) {                  /* (first unit of the worker) */
            builder_arg_t *ba = (builder_arg_t *)malloc(sizeof(builder_arg_t));
            if (!ba) { rc = -1; goto done; }
            ba->w = w; ba->id = w->n_bth;
            if (pthread_create(&w->bth[w->n_bth], NULL, builder_thread, ba) != 0) { free(ba); rc = -1; goto done; }
        }
        pthread_mutex_lock(&w->b_mu);                                 /* the unit goes to the worker's builder threads */
        w->b_unit = &q; w->b_left = w->n_bth; w->b_gen++;
        pthread_cond_broadcast(&w->b_cv);
        pthread_mutex_unlock(&w->b_mu);
        builder_on = g_builders;
    }
#define FAIL(code) do { rc = (code); goto done; } while (0)
#define TRY(call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, w->g ? xgpu_last_error(w->g) : ""); FAIL(rc_); } } while (0)

    for (long k = 0;; k++) {
        const int ks = (int)(k % g_depth);
        xhost_picture *pp = &q.pic[ks];
        int prc;
        if (thread_on) {
            pthread_mutex_lock(&q.mu);
            while (q.produced <= k) pthread_cond_wait(&q.cv, &q.mu);
            prc = q.rc[ks];
            pthread_mutex_unlock(&q.mu);
        } else {
            const double t0 = now_s();
            prc = xhost_parser_next(q.ps, pp);
            q.parse_s += now_s() - t0;
            q.parse_ms[ks] = 1e3 * (now_s() - t0);
        }
        if (prc < 0) { fprintf(stderr, "parser: %s\n", xhost_parser_error(q.ps)); FAIL(prc); }
        if (prc == 0) break;
        const xhost_picture p = *pp;                                 /* the arrays it points at stay valid until picture k is given back below */
        if (!have_ctx) {                                             /* first picture: the sequence parameters are known */
            TRY(worker_context(w, &p));
            pthread_mutex_lock(&w->arena_mu); w->arena_ok = 1; pthread_mutex_unlock(&w->arena_mu);
            for (int i = 0; i < w->n_slots; i++) free_pic[n_free++] = w->slots[i];
            frame_bytes = xgpu_pic_output_size(w->g, out_bd_arg ? out_bd_arg : p.bit_depth_luma, 0, 0, 0, 0);
            const size_t need = (size_t)(expected > 0 ? expected : 1) * frame_bytes;
            if (w->frames_cap < need) {                              /* the worker's output buffer grows to the largest unit it has seen */
                if (w->frames) { if (w->frames_pinned) xgpu_host_free(w->g, w->frames); else free(w->frames); }
                void *pin = NULL;
                w->frames_pinned = xgpu_host_alloc(w->g, need, &pin) == 0;
                w->frames = w->frames_pinned ? (uint8_t *)pin : (uint8_t *)malloc(need);       /* pageable: the copies block, the result is the same */
                w->frames_cap = w->frames ? need : 0;
            }
            frames = w->frames; pinned = w->frames_pinned;
            if (!frames) FAIL(-1);
            have_ctx = 1;
            pthread_mutex_lock(&q.mu); q.g = w->g; pthread_cond_broadcast(&q.cv); pthread_mutex_unlock(&q.mu);      /* the builder thread may start */
        }
        if (p.is_idr) {                                              /* an IDR empties the DPB */
            for (int i = 0; i < MAX_SLOTS; i++) if (dpb[i].in_use) { free_pic[n_free++] = dpb[i].pic; dpb[i].in_use = 0; }
            epoch++;
        }
        if (n_free == 0) {                                           /* device pictures are allocated when the stream first needs them */
            if (w->n_slots == MAX_SLOTS) { fprintf(stderr, "DPB overflow\n"); FAIL(-1); }
            const double t0 = now_s();
            const int id = xgpu_pic_alloc(w->g);
            if (id < 0) FAIL(id);
            w->setup_s += now_s() - t0;
            w->slots[w->n_slots++] = id; free_pic[n_free++] = id;
        }
        const int cur = free_pic[--n_free];

        xgpu_frame_params fp;
        memset(&fp, 0, sizeof(fp));
        fp.pic = cur; fp.poc = p.poc;
        for (int l = 0; l < 2; l++) {
            fp.num_refp[l] = p.num_refp[l];
            for (int i = 0; i < p.num_refp[l]; i++) {
                int s = -1;
                for (int j = 0; j < MAX_SLOTS; j++) if (dpb[j].in_use && dpb[j].poc == p.refp_poc[i][l]) s = dpb[j].pic;
                if (s < 0) { fprintf(stderr, "reference POC %d is not in the DPB\n", p.refp_poc[i][l]); FAIL(-1); }
                fp.refp_pic[i][l] = s; fp.refp_poc[i][l] = p.refp_poc[i][l];
            }
        }
        fp.qp_u_offset = p.qp_u_offset; fp.qp_v_offset = p.qp_v_offset;
        fp.deblock_alpha_offset = p.deblock_alpha_offset; fp.deblock_beta_offset = p.deblock_beta_offset;
        fp.deblock_on = p.deblock_on; fp.alf_on = p.alf_on;

        double build_ms;
        if (builder_on) {                                            /* built by the builder thread while picture k - 1 was being launched */
            pthread_mutex_lock(&q.mu);
            while (q.built_k[ks] != k + 1) pthread_cond_wait(&q.cv, &q.mu);
            const int brc = q.build_rc[ks];
            db = q.db[ks]; q.db[ks] = NULL;
            build_ms = q.build_ms[ks];
            pthread_mutex_unlock(&q.mu);
            if (brc < 0) { fprintf(stderr, "xgpu_batch_create -> %d (%s)\n", brc, xgpu_last_error(w->g)); FAIL(brc); }
        } else {
            const double tb = now_s();
            TRY(xgpu_batch_create(w->g, &p.batch, &db));            /* the parser's arrays are consumed before the call returns */
            q.build_s += now_s() - tb;
            build_ms = 1e3 * (now_s() - tb);
        }
        if (g_trace) fprintf(stderr, "picture %ld: poc %d slice %s, %d CUs, %zu coefficients: parse %.2f ms, batch build %.2f ms\n", k, p.poc,
                             p.slice_type == XHOST_SLICE_I ? "I" : p.slice_type == XHOST_SLICE_P ? "P" : "B", p.batch.n_cu, p.batch.n_coef, q.parse_ms[ks], build_ms);
        TRY(xgpu_frame_begin(w->g, &fp));
        TRY(xgpu_batch_recon(w->g, db));
        if (p.deblock_on) TRY(xgpu_deblock(w->g));
        if (p.alf_on) TRY(xgpu_alf(w->g, &p.alf));
        TRY(xgpu_pad(w->g));
        TRY(xgpu_frame_end(w->g));
        if (p.n_dmvr_sub > 0) {
            /* sps->tool_dmvr: the refined vectors of this picture go back to the parser before it parses the next one (temporal merge candidates);
               the parser thread is parked until this picture is given back */
            mv = (int16_t *)malloc(sizeof(int16_t) * 4 * (size_t)p.n_dmvr_sub);
            const int got = mv ? xgpu_batch_dmvr_mvs(w->g, db, mv, p.n_dmvr_sub) : -1;
            const int fed = got == p.n_dmvr_sub ? xhost_parser_set_dmvr_mvs(q.ps, mv, got) : -1;
            free(mv); mv = NULL;
            if (fed < 0) { fprintf(stderr, "DMVR vectors: backend %d, parser %d (%s)\n", got, fed, xhost_parser_error(q.ps)); FAIL(-1); }
        }
        if (p.needs_ref_luma) {
            /* tool_dmvr with tool_hmvp / tool_mmvd: the syntax of later pictures depends on refined vectors, so the parser runs the refinement search
               itself (xevd_amd/host/dmvr_search.h) on this picture's decoded luma: one padded plane per device picture, downloaded behind the kernels */
            const int stride = p.width + 2 * XGPU_PAD_L, slot = cur;                  /* device picture ids are 0 .. max_pics - 1 = MAX_SLOTS */
            if (!w->ref_luma[slot]) {                                /* pinned: the parser may be waiting for this plane, and a pageable download takes three times as long */
                const size_t bytes = sizeof(int16_t) * (size_t)stride * (size_t)(p.height + 2 * XGPU_PAD_L);
                void *m = NULL;
                if (xgpu_host_alloc(w->g, bytes, &m) >= 0 && m) { w->ref_luma[slot] = (int16_t *)m; w->ref_luma_pinned[slot] = 1; }
                else w->ref_luma[slot] = (int16_t *)malloc(bytes);
            }
            if (!w->ref_luma[slot]) FAIL(-1);
            TRY(xgpu_pic_download_padded(w->g, cur, w->ref_luma[slot], NULL, NULL));
            TRY(xhost_parser_set_ref_luma(q.ps, p.poc, w->ref_luma[slot] + (size_t)XGPU_PAD_L * stride + XGPU_PAD_L, stride));
        }
        TRY(xgpu_batch_wait_upload(w->g, db));                      /* the coefficient arena (the parser's, pinned) has left host memory: its slot may be parsed into again */
        xgpu_batch_destroy(w->g, db); db = NULL;

        if (n_pics >= expected) { fprintf(stderr, "more pictures than slice NAL units\n"); FAIL(-1); }
        /* output of this picture behind its kernels, overlapping the parsing and the kernels of the next one; the DRA post-filter, when the
           PPS switches it on, is part of it */
        xgpu_dra_luts dra = { p.dra_lut[0], { p.dra_lut[1], p.dra_lut[2] } };
        const int prev = ticket;
        TRY(xgpu_pic_output_async(w->g, cur, p.dra_lut[0] ? &dra : NULL, out_bd_arg ? out_bd_arg : p.bit_depth_luma, 0, 0, 0, 0,
                                  frames + (size_t)n_pics * frame_bytes, frame_bytes, &ticket));
        if (prev >= 0 && prev != ticket) TRY(xgpu_pic_output_wait(w->g, prev));
        outs[n_pics].epoch = epoch; outs[n_pics].poc = p.poc; outs[n_pics].off = (size_t)n_pics * frame_bytes;
        n_pics++;

        for (int r = 0; r < p.n_release; r++)                        /* pictures the stream unmarked when this one was stored */
            for (int j = 0; j < MAX_SLOTS; j++)
                if (dpb[j].in_use && dpb[j].poc == p.release_poc[r]) { free_pic[n_free++] = dpb[j].pic; dpb[j].in_use = 0; }
        if (p.is_ref) {
            for (int j = 0; j < MAX_SLOTS; j++) if (!dpb[j].in_use) { dpb[j].in_use = 1; dpb[j].poc = p.poc; dpb[j].pic = cur; break; }
        } else free_pic[n_free++] = cur;

        if (thread_on) {                                             /* picture k's arrays go back to the parser */
            pthread_mutex_lock(&q.mu);
            q.released = k + 1;
            pthread_cond_broadcast(&q.cv);
            pthread_mutex_unlock(&q.mu);
        }
    }
    if (w->g) TRY(xgpu_sync(w->g));                                 /* the last outputs have landed */
    qsort(outs, (size_t)n_pics, sizeof(out_t), cmp_out);            /* output order: ascending POC inside every IDR period */
#undef TRY
#undef FAIL
done:
    if (thread_on) {                                                /* every path: stop and join the parser thread before its parser goes away */
        pthread_mutex_lock(&q.mu);
        q.stop = 1;
        pthread_cond_broadcast(&q.cv);
        pthread_mutex_unlock(&q.mu);
        if (rc < 0) xhost_parser_cancel_wait(q.ps);                 /* a parser waiting for a luma plane that will not come any more */
        pthread_join(th, NULL);
        if (builder_on) {                                           /* the builders have left the unit (its pipe_t lives on this stack frame) */
            pthread_mutex_lock(&w->b_mu);
            while (w->b_left) pthread_cond_wait(&w->b_cv, &w->b_mu);
            w->b_unit = NULL;
            pthread_mutex_unlock(&w->b_mu);
        }
        for (int i = 0; i < PIPE_DEPTH; i++) if (q.db[i]) { xgpu_batch_destroy(w->g, q.db[i]); q.db[i] = NULL; }      /* built, never launched (an error further down the pipeline) */
    }
    pthread_mutex_destroy(&q.mu); pthread_cond_destroy(&q.cv);
    if (rc < 0 && w->ps) { xhost_parser_close(w->ps); w->ps = NULL; }      /* a failed unit: the next one starts with a new parser */
    w->parse_s += q.parse_s; w->build_s += q.build_s;
    free(mv);
    if (db) xgpu_batch_destroy(w->g, db);
    if (rc < 0) {                                                   /* nothing is handed out on failure */
        if (w->g) (void)xgpu_sync(w->g);                            /* queued downloads still write into the worker's output buffer */
        free(outs);
        return rc;
    }
    *frames_out = frames; *outs_out = outs; *n_out = n_pics; *frame_bytes_out = frame_bytes; *pinned_out = pinned;
    return 0;
}

/* ---- work queue callbacks: one worker per device ---- */
static void *worker_init(int device, void *user)
{
    worker_t *w = (worker_t *)calloc(1, sizeof(worker_t));
    if (!w) return NULL;
    w->device = device; w->streams = (const stream_t *)user;
    pthread_mutex_init(&w->arena_mu, NULL);
    pthread_mutex_init(&w->b_mu, NULL); pthread_cond_init(&w->b_cv, NULL);
    /* is the device there?  A worker without one leaves the queue to the others instead of failing their jobs */
    xgpu_seq_params sp;
    xgpu_ctx *probe = NULL;
    memset(&sp, 0, sizeof(sp));
    sp.device = device; sp.width = 64; sp.height = 64; sp.bit_depth_luma = sp.bit_depth_chroma = 8; sp.chroma_format_idc = 1; sp.log2_ctu = 6; sp.max_pics = 1;
    if (xgpu_open(&sp, &probe) < 0) { fprintf(stderr, "device %d is not usable: its jobs go to the other workers\n", device); free(w); return NULL; }
    xgpu_close(probe);
    return w;
}
static int worker_job(void *state, const xwq_job *job)
{
    worker_t *w = (worker_t *)state;
    const stream_t *s = &w->streams[job->stream];
    const size_t cap = (size_t)job->offset + (size_t)job->size + 16;
    uint8_t *unit = (uint8_t *)malloc(cap), *frames = NULL;
    out_t *outs = NULL;
    int n = 0, pinned = 0;
    size_t frame_bytes = 0;
    if (!unit) return -1;
    const size_t len = xwq_unit_bytes(s->bytes, s->size, job, unit, cap);
    const double t0 = now_s();
    int rc = len ? decode_unit(w, unit, len, s->out_bd, job->n_pictures, &frames, &outs, &n, &frame_bytes, &pinned) : -1;
    w->busy_s += now_s() - t0;
    free(unit);
    if (rc < 0) return rc;
    if (n != job->n_pictures) { fprintf(stderr, "stream %d unit %d: %d pictures decoded, %d expected\n", job->stream, job->unit, n, job->n_pictures); rc = -1; }
    else if (g_keep_units < 0 || job->unit < g_keep_units)
        for (int i = 0; i < n && rc >= 0; i++)                       /* picture by picture, in output order, at the unit's place in the file */
            if (pwrite(s->fd, frames + outs[i].off, frame_bytes, (off_t)((size_t)(job->first_picture + i) * frame_bytes)) != (ssize_t)frame_bytes) { perror("pwrite"); rc = -1; }
    if (rc >= 0 && g_hash_units && job->stream < 64 && job->unit < 256) {
        /* word-wise multiply-xorshift over the unit's pictures in output order (not a cryptographic hash: equal units of a repeated GOP must agree, and the
           first unit's pictures are compared with the reference decoder's by the caller) */
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < n; i++) {
            const uint64_t *p = (const uint64_t *)(frames + outs[i].off);
            for (size_t k = 0; k < frame_bytes / 8; k++) { h = (h ^ p[k]) * 0xD6E8FEB86659FD93ull; h ^= h >> 32; }
        }
        g_unit_hash[job->stream][job->unit] = h;
    }
    (void)pinned;                                                   /* `frames` is the worker's buffer: reused by its next unit */
    free(outs);
    w->pictures += n;
    return rc;
}
static double g_busy[64], g_setup[64], g_parse[64], g_build[64];
static long g_pics[64];
static int g_dev[64];
static int g_json = 0;              /* --json: one JSON line with the run's figures on stdout (bench.py's multi-GPU leg reads it) */
static int g_workers;
static void worker_fini(void *state)
{
    worker_t *w = (worker_t *)state;
    { const int k = __sync_fetch_and_add(&g_workers, 1); g_busy[k & 63] = w->busy_s - w->setup_s; g_setup[k & 63] = w->setup_s; g_parse[k & 63] = w->parse_s; g_build[k & 63] = w->build_s; g_pics[k & 63] = w->pictures; g_dev[k & 63] = w->device; }
    if (w->n_bth) {                                                 /* the builder threads go first: they use the context */
        pthread_mutex_lock(&w->b_mu); w->b_quit = 1; pthread_cond_broadcast(&w->b_cv); pthread_mutex_unlock(&w->b_mu);
        for (int i = 0; i < w->n_bth; i++) pthread_join(w->bth[i], NULL);
    }
    if (w->frames && !w->frames_pinned) free(w->frames);            /* (pinned memory goes with the context) */
    if (w->ps) xhost_parser_close(w->ps);                           /* before the context: it gives its arenas back */
    if (w->g) xgpu_close(w->g);
    for (int i = 0; i < MAX_SLOTS + 2; i++) if (!w->ref_luma_pinned[i]) free(w->ref_luma[i]);
    free(w);
}

int main(int argc, char **argv)
{
    int gpus = 1, workers = 1, out_bd = 0, a = 1;
    /* the per-picture arrays of the parser and the batch builder are tens of megabytes: kept in the heap instead of being mapped and unmapped per
       picture (every munmap interrupts all the threads of the process - the tile threads of the parser - to flush their TLBs) */
    mallopt(M_MMAP_THRESHOLD, 32 << 20);        /* the largest value glibc takes */
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    while (a < argc && argv[a][0] == '-' && argv[a][1] == '-') {
        if (!strcmp(argv[a], "--gpus") && a + 1 < argc) { gpus = atoi(argv[a + 1]); a += 2; }
        else if (!strcmp(argv[a], "--workers") && a + 1 < argc) { workers = atoi(argv[a + 1]); a += 2; }
        else if (!strcmp(argv[a], "--bd") && a + 1 < argc) { out_bd = atoi(argv[a + 1]); a += 2; }
        else if (!strcmp(argv[a], "--tile-threads") && a + 1 < argc) { g_tile_threads = atoi(argv[a + 1]); a += 2; }
        else if (!strcmp(argv[a], "--no-pipeline")) { g_pipeline = 0; a += 1; }
        else if (!strcmp(argv[a], "--trace")) { g_trace = 1; a += 1; }
        else if (!strcmp(argv[a], "--build-threads") && a + 1 < argc) { g_build_threads = atoi(argv[a + 1]); a += 2; }
        else if (!strcmp(argv[a], "--builders") && a + 1 < argc) { g_builders = atoi(argv[a + 1]); if (g_builders < 1) g_builders = 1; if (g_builders > MAX_BUILDERS) g_builders = MAX_BUILDERS; g_depth = 2 + g_builders; a += 2; }
        else if (!strcmp(argv[a], "--keep-units") && a + 1 < argc) { g_keep_units = atoi(argv[a + 1]); a += 2; }
        else if (!strcmp(argv[a], "--hash-units")) { g_hash_units = 1; a += 1; }
        else if (!strcmp(argv[a], "--json")) { g_json = 1; a++; }
        else break;
    }
    int n_pos = argc - a;
    if (n_pos == 3 && strspn(argv[a + 2], "0123456789") == strlen(argv[a + 2])) { out_bd = atoi(argv[a + 2]); n_pos = 2; }      /* in out D */
    if (n_pos < 2 || (n_pos & 1) || gpus < 1 || workers < 1 || gpus * workers > 64 || n_pos / 2 > MAX_STREAMS) {
        fprintf(stderr, "usage: %s [--gpus N] [--workers W] [--tile-threads T] [--build-threads B] [--builders N] [--bd D] in.evc out.yuv [in2.evc out2.yuv ...]\n", argv[0]);
        return 2;
    }
    static stream_t streams[MAX_STREAMS];
    static xwq_job jobs[MAX_GOPS];
    const int n_streams = n_pos / 2;
    xwq *q = xwq_create();
    if (!q) return 1;
    long total_pictures = 0;
    int n_jobs = 0;
    for (int s = 0; s < n_streams; s++) {
        FILE *f = fopen(argv[a + 2 * s], "rb");
        if (!f) { perror(argv[a + 2 * s]); return 2; }
        fseek(f, 0, SEEK_END);
        const long size = ftell(f);
        fseek(f, 0, SEEK_SET);
        uint8_t *bytes = (uint8_t *)malloc((size_t)size + 1);
        if (!bytes || fread(bytes, 1, (size_t)size, f) != (size_t)size) return 2;
        fclose(f);
        streams[s].bytes = bytes; streams[s].size = (size_t)size; streams[s].out_bd = out_bd;
        streams[s].fd = open(argv[a + 2 * s + 1], O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (streams[s].fd < 0) { perror(argv[a + 2 * s + 1]); return 2; }
        /* one job per closed GOP: the unit of the queue, and the bound on the pinned output buffer of a worker */
        int n = xwq_split_gops(bytes, (size_t)size, s, jobs, MAX_GOPS);
        if (n == -203) { fprintf(stderr, "%s: more than %d closed GOPs\n", argv[a + 2 * s], MAX_GOPS); return 1; }
        if (n < 0) { fprintf(stderr, "%s: damaged NAL length prefix\n", argv[a + 2 * s]); return 1; }
        if (n == 0) continue;                                       /* no picture: an empty output file */
        for (int k = 0; k < n; k++) { xwq_push(q, &jobs[k]); total_pictures += jobs[k].n_pictures; }
        n_jobs += n;
    }
    xwq_close(q);
    int devices[64], done[64];
    if (g_builders == 0) { g_builders = gpus * workers <= 2 ? 2 : 1; g_depth = 2 + g_builders; }
    gpus *= workers;                                                /* worker i runs on device i / workers */
    for (int i = 0; i < gpus; i++) devices[i] = i / workers;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int rc = xwq_run(q, devices, gpus, worker_init, worker_job, worker_fini, streams, done);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    for (int s = 0; s < n_streams; s++) { close(streams[s].fd); free((void *)streams[s].bytes); }
    xwq_destroy(q);
    if (rc < 0) { fprintf(stderr, "decoding failed: %d\n", rc); return 1; }
    fprintf(stderr, "%ld pictures\n", total_pictures);
    fprintf(stderr, "%d stream(s), %d job(s) on %d worker(s), %d per device:", n_streams, n_jobs, gpus, workers);
    for (int i = 0; i < gpus; i++) fprintf(stderr, " %d", done[i]);
    double busy = 0, setup = 0, parse = 0, build = 0;
    for (int i = 0; i < g_workers && i < 64; i++) { if (g_busy[i] > busy) busy = g_busy[i]; if (g_setup[i] > setup) setup = g_setup[i]; parse += g_parse[i]; build += g_build[i]; }
    fprintf(stderr, " jobs; %.3f s wall (device start-up included), %.2f pictures/s; decoding alone (parsing + kernels + output, the span xevd_app times: "
            "app/xevd_app.c:492-501,612-624; slowest worker) %.3f s, %.2f pictures/s; context + picture allocation %.3f s\n",
            secs, secs > 0 ? (double)total_pictures / secs : 0.0, busy, busy > 0 ? (double)total_pictures / busy : 0.0, setup);
    fprintf(stderr, "stages per picture: parse %.2f ms (%s), batch build %.2f ms\n", total_pictures ? 1e3 * parse / (double)total_pictures : 0.0,
            g_pipeline ? "own thread, one picture ahead" : "same thread", total_pictures ? 1e3 * build / (double)total_pictures : 0.0);
    if (g_json) {
        struct rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        long per_dev[64];
        memset(per_dev, 0, sizeof(per_dev));
        for (int i = 0; i < g_workers && i < 64; i++) if (g_dev[i] >= 0 && g_dev[i] < 64) per_dev[g_dev[i]] += g_pics[i];
        printf("{\"pictures\": %ld, \"streams\": %d, \"jobs\": %d, \"devices\": %d, \"workers_per_device\": %d, \"tile_threads\": %d, \"build_threads\": %d, \"builders\": %d, \"pipeline\": %d, "
               "\"wall_s\": %.4f, \"decode_only_s\": %.4f, \"fps_wall\": %.2f, \"fps_decode_only\": %.2f, \"setup_s\": %.3f, \"parse_ms_per_picture\": %.3f, \"build_ms_per_picture\": %.3f, "
               "\"cpu_user_s\": %.3f, \"cpu_sys_s\": %.3f, \"pictures_per_device\": [",
               total_pictures, n_streams, n_jobs, gpus / workers, workers, g_tile_threads, g_build_threads, g_builders, g_pipeline, secs, busy, secs > 0 ? (double)total_pictures / secs : 0.0,
               busy > 0 ? (double)total_pictures / busy : 0.0, setup, total_pictures ? 1e3 * parse / (double)total_pictures : 0.0, total_pictures ? 1e3 * build / (double)total_pictures : 0.0,
               (double)ru.ru_utime.tv_sec + 1e-6 * (double)ru.ru_utime.tv_usec, (double)ru.ru_stime.tv_sec + 1e-6 * (double)ru.ru_stime.tv_usec);
        for (int d = 0; d < gpus / workers; d++) printf("%s%ld", d ? ", " : "", per_dev[d]);
        printf("]");
        if (g_hash_units) {
            printf(", \"unit_hashes\": [");
            for (int st = 0; st < n_streams && st < 64; st++) {
                int units = 0;
                for (int u = 0; u < 256; u++) if (g_unit_hash[st][u]) units = u + 1;
                printf("%s[", st ? ", " : "");
                for (int u = 0; u < units; u++) printf("%s\"%016llx\"", u ? ", " : "", (unsigned long long)g_unit_hash[st][u]);
                printf("]");
            }
            printf("]");
        }
        printf("}\n");
    }
    return 0;
}
