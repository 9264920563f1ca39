// xgpu_host.h - what the host-side translation units of the backend share (xgpu_api.hip: context, pictures, output; xgpu_builder.hip: the batch builder and its
// dependency plan; xgpu_launch.hip: the per-picture launch sequencing; xgpu_shims.hip: the fine-grained test shims).  Private to xevd_amd/csrc.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include "xgpu_internal.h"

#define HIPCHK(c, expr)                                                                                         \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            snprintf((c)->err, sizeof((c)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return XGPU_ERR_UNEXPECTED;                                                                         \
        }                                                                                                       \
    } while (0)
#define ARGCHK(c, cond)                                                                                         \
    do {                                                                                                        \
        if (!(cond)) {                                                                                          \
            if (c) snprintf((c)->err, sizeof((c)->err), "%s:%d invalid argument: %s", __FILE__, __LINE__, #cond); \
            return XGPU_ERR_INVALID_ARGUMENT;                                                                   \
        }                                                                                                       \
    } while (0)


static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
static inline bool valid_pic(const xgpu_ctx *c, int pic) { return pic >= 0 && pic + 1 < (int)c->pics.size() && c->pics[pic + 1].used; }
static inline DevPic &dpic(xgpu_ctx *c, int pic) { return c->pics[pic + 1]; }
// ADDB directly followed by ALF runs as ONE kernel (k_addb_alf)
static inline bool addb_alf_fused(const xgpu_ctx *c) { return c->fp.deblock_on && c->sp.tool_addb && c->fp.alf_on && !c->split_addb_alf; }

// HIP-event kernel timing (xgpu_api.hip)
void time_begin(xgpu_ctx *c, int k, hipEvent_t *a, hipEvent_t *b);
void time_end(xgpu_ctx *c, int k, hipEvent_t a, hipEvent_t b);
#define TIMED(c, k, stmt)                      \
    do {                                       \
        hipEvent_t ta_ = 0, tb_ = 0;           \
        time_begin(c, k, &ta_, &tb_);          \
        stmt;                                  \
        time_end(c, k, ta_, tb_);              \
    } while (0)


// xgpu_tile_grid -> TileMask; false: not a partition of the picture's CTU grid
inline bool tile_mask(const xgpu_ctx *c, const xgpu_tile_grid *g, TileMask &m)
{
    memset(&m, 0, sizeof(m));
    if (!g) return true;
    if (g->n_cols < 1 || g->n_cols > XGPU_MAX_TILE_COLS || g->n_rows < 1 || g->n_rows > XGPU_MAX_TILE_ROWS || c->w_ctu > 256 || c->h_ctu > 256) return false;
    if (g->col_bd[0] != 0 || g->row_bd[0] != 0 || g->col_bd[g->n_cols] != c->w_ctu || g->row_bd[g->n_rows] != c->h_ctu) return false;
    for (int i = 0; i < g->n_cols; i++) if (g->col_bd[i + 1] <= g->col_bd[i]) return false;
    for (int j = 0; j < g->n_rows; j++) if (g->row_bd[j + 1] <= g->row_bd[j]) return false;
    for (int i = 1; i < g->n_cols; i++) m.vb[g->col_bd[i] >> 5] |= 1u << (g->col_bd[i] & 31);
    for (int j = 1; j < g->n_rows; j++) m.hb[g->row_bd[j] >> 5] |= 1u << (g->row_bd[j] & 31);
    return true;
}

// The batch builder: SoA batch of the ABI -> 32-byte CU records, the TB list sorted by size class and the
// wave work items of the itdq kernel, written into ONE pinned staging block and sent with one async copy per
// array.  (xevd_ctu_row_rec_mt's per-CU cu_init + coef_rect_to_series, xevd.c:567-676, become this pass.)
// host_only (xgpu_test_build_batch): everything but the device - the staging block comes from malloc, nothing is uploaded; `segs` receives (offset, bytes) of
// every array in the staging block.  The CPU suite pins the builder's output with it (digests, thread-count independence).
