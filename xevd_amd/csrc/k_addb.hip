// k_addb.hip - ADDB deblocking (Main profile, sps->tool_addb), vertical-edge pass and horizontal-edge pass.
//
// Replaces xevdm_deblock -> deblock_tree -> xevdm_deblock_cu_ver / _hor -> deblock_addb_cu_* -> deblock_scu_line_*
// (src_main/xevdm.c:1935-2103, src_main/xevdm_df.c:361-1135).  Semantics reproduced: only CU edges on the 8x8 luma
// grid; boundary strength 0..4 from intra / CTU crossing / luma cbf / reference PICTURES + MVs (get_bs); QP =
// average of both sides; alpha/beta/clip tables indexed through get_index()'s u8 arguments; luma strong (bS 4)
// and normal filters over 3 samples per side, chroma over 1; all vertical edges before all horizontal ones.
//
// MI355X mapping - out of place (SRC -> DST) and order-free: grid edges are 8 samples apart and touch 3 samples per side, so edges never interact and
// the 8-sample windows centred on the grid lines tile the picture in both directions: one LANE per 4-sample edge SEGMENT owns the SCU on either side
// (P = left / above, Q = right / below), filters once and writes both halves - every sample is read once and written once, no filter is evaluated
// twice.  Both edge directions run in ONE kernel (k_addb_fused below; round 2 had a kernel per direction: two reads and two writes of the picture).
// All loads are issued before any decision; decisions are lane-local integer tests, tables in LDS.
#include "xgpu_internal.h"
#include "addb_filter.h"

// ---------------------------------------------------------------------------------------------------------
// k_addb_fused - both edge directions in ONE kernel: one read and one write of the picture instead of two of each.
//
// Grid edges are 8 samples apart and a filter touches at most 4 samples on either side (3 written), so the 8-sample windows centred on the grid lines
// tile the picture in BOTH directions.  A workgroup therefore owns the tile [X0 - 4, X0 + 252) x [Y0 - 4, Y0 + 12) - shifted by half a window against
// the 8x8 grid - which holds 32 x 4 complete vertical-edge windows (8 samples x 4 rows) and, at the same time, 64 x 2 complete horizontal-edge windows
// (4 samples x 8 rows): no halo in either direction, nothing is read or filtered twice.
//   phase V: lane = one vertical-edge segment: SCU records + its luma / chroma windows from HBM (a wave's loads are two 512-byte runs per
//            row), filter, windows -> LDS, the two SCU records -> LDS (the horizontal phase needs the same 64 x 8 records);
//   barrier;
//   phase H: lane = one horizontal-edge segment: windows and records from LDS, filter, stores to DST (512-byte runs per row).
// All vertical edges of the picture before all horizontal ones (xevdm_deblock, src_main/xevdm.c:2048-2103) holds per sample: a horizontal filter reads
// only its own window, which phase V of the same workgroup has completed.
// ---------------------------------------------------------------------------------------------------------
#define AF_LS 264                 // LDS luma row stride in samples: 256 + 8 (rows 4 apart land 16 banks apart; rows stay 16-byte aligned)
#define AF_CS 136                 // chroma: 128 + 8
template <int SR>                 // SCU rows of the tile: 4 = 128 threads, 256 x 16 samples (8 = 256 threads, 256 x 32 samples measured the same)
__global__ __launch_bounds__(32 * SR) void k_addb_fused(const AddbArgs a, const int16_t *__restrict__ sy_, const int16_t *__restrict__ su_,
                                                    const int16_t *__restrict__ sv_, int16_t *__restrict__ dy_, int16_t *__restrict__ du_,
                                                    int16_t *__restrict__ dv_, int tiles_x)
{
    __shared__ uint8_t s_alpha[52], s_beta[52], s_clip[52 * 5], s_pic[XGPU_MAX_REFS * 2];
    __shared__ int8_t s_cqp[2 * 96];
    constexpr int NT = 32 * SR;
    __shared__ __attribute__((aligned(16))) int16_t s_y[4 * SR * AF_LS];
    __shared__ __attribute__((aligned(16))) int16_t s_c[2][2 * SR * AF_CS];
    __shared__ uint4 s_map[SR][64];
    const int t = threadIdx.x;
    for (int i = t; i < 52; i += NT) { s_alpha[i] = k_alpha[i]; s_beta[i] = k_beta[i]; }
    for (int i = t; i < 260; i += NT) s_clip[i] = ((const uint8_t *)k_clip)[i];
    for (int i = t; i < XGPU_MAX_REFS * 2; i += NT) s_pic[i] = a.pic_id[i];
    for (int i = t; i < 192; i += NT) s_cqp[i] = a.chroma_qp[i];
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int n_ex = (a.w_scu >> 1) + 1, n_ey = (a.h_scu >> 1) + 1;     // grid lines incl. the picture borders (which only copy their inner half)
#define PK2(lo, hi) ((uint32_t)(uint16_t)(lo) | ((uint32_t)(uint16_t)(hi) << 16))

    // ---------------------------------------------------------------- phase V: vertical edges
    {
        const int wx = t & 31, sr = t >> 5;                  // window along x, SCU row of the tile
        const int ex = (tx << 5) + wx;                       // grid line x = 8 * ex
        const int srow = ty * SR - 1 + sr;                   // SCU row in the picture
        const int sxq = ex << 1;                             // the Q-side SCU column
        const bool ok = ex < n_ex && srow >= 0 && srow < a.h_scu;
        const bool has_p = ok && ex > 0, has_q = ok && sxq < a.w_scu;
        const uint4 *maps = (const uint4 *)a.maps;
        uint4 rq = make_uint4(0, 0, 0, 0), rp = rq;
        int L[4][8], Cc[2][2][4];
        if (ok) {
            const int kq = srow * a.w_scu + sxq;
            if (has_q) rq = maps[kq];
            if (has_p) rp = maps[kq - 1];
            const int x = sxq << 2, y = srow << 2, cy = srow << 1;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const U32x4a8 v = *(const U32x4a8 *)(sy_ + (y + r) * a.s_l + x - 4);
                L[r][0] = (int16_t)(v.a & 0xFFFF); L[r][1] = (int16_t)(v.a >> 16); L[r][2] = (int16_t)(v.b & 0xFFFF); L[r][3] = (int16_t)(v.b >> 16);
                L[r][4] = (int16_t)(v.c & 0xFFFF); L[r][5] = (int16_t)(v.c >> 16); L[r][6] = (int16_t)(v.d & 0xFFFF); L[r][7] = (int16_t)(v.d >> 16);
            }
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const U32x2a4 v = *(const U32x2a4 *)((pl ? sv_ : su_) + (cy + r) * a.s_c + (x >> 1) - 2);
                    Cc[pl][r][0] = (int16_t)(v.a & 0xFFFF); Cc[pl][r][1] = (int16_t)(v.a >> 16);
                    Cc[pl][r][2] = (int16_t)(v.b & 0xFFFF); Cc[pl][r][3] = (int16_t)(v.b >> 16);
                }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) L[r][k] = 0;
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int k = 0; k < 4; k++) Cc[pl][r][k] = 0;
        }
        __syncthreads();                                     // the tables (the loads above are in flight across it)
        if (has_p && has_q) addb_edge<0>(a, rq, rp, sxq, L, Cc, s_alpha, s_beta, s_clip, s_cqp, s_pic);
        s_map[sr][2 * wx] = rp; s_map[sr][2 * wx + 1] = rq;
#pragma unroll
        for (int r = 0; r < 4; r++)
            *(uint4 *)(s_y + (4 * sr + r) * AF_LS + 8 * wx) = make_uint4(PK2(L[r][0], L[r][1]), PK2(L[r][2], L[r][3]), PK2(L[r][4], L[r][5]), PK2(L[r][6], L[r][7]));
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int r = 0; r < 2; r++)
                *(uint2 *)(s_c[pl] + (2 * sr + r) * AF_CS + 4 * wx) = make_uint2(PK2(Cc[pl][r][0], Cc[pl][r][1]), PK2(Cc[pl][r][2], Cc[pl][r][3]));
    }
    __syncthreads();

    // ---------------------------------------------------------------- phase H: horizontal edges
    {
        const int sx = t & 63, g = t >> 6;                   // SCU column of the tile, grid line of the tile
        const int scol = (tx << 6) - 1 + sx;                 // SCU column in the picture
        const int ey = ty * (SR / 2) + g;                    // grid line y = 8 * ey
        if (scol < 0 || scol >= a.w_scu || ey >= n_ey) return;
        const int syq = ey << 1;                             // the Q-side SCU row
        const bool has_p = ey > 0, has_q = syq < a.h_scu;
        const uint4 rq = s_map[2 * g + 1][sx], rp = s_map[2 * g][sx];
        int L[4][8], Cc[2][2][4];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint2 v = *(const uint2 *)(s_y + (8 * g + r) * AF_LS + 4 * sx);
            L[0][r] = (int16_t)(v.x & 0xFFFF); L[1][r] = (int16_t)(v.x >> 16); L[2][r] = (int16_t)(v.y & 0xFFFF); L[3][r] = (int16_t)(v.y >> 16);
        }
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t v = *(const uint32_t *)(s_c[pl] + (4 * g + r) * AF_CS + 2 * sx);
                Cc[pl][0][r] = (int16_t)(v & 0xFFFF); Cc[pl][1][r] = (int16_t)(v >> 16);
            }
        if (has_p && has_q) addb_edge<1>(a, rq, rp, syq, L, Cc, s_alpha, s_beta, s_clip, s_cqp, s_pic);
        const int x = scol << 2, y = syq << 2, cx = scol << 1, cy = syq << 1;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (r < 4 ? !has_p : !has_q) continue;
            *(uint2 *)(dy_ + (y - 4 + r) * a.s_l + x) = make_uint2(PK2(L[0][r], L[1][r]), PK2(L[2][r], L[3][r]));
        }
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (r < 2 ? !has_p : !has_q) continue;
                *(uint32_t *)((pl ? dv_ : du_) + (cy - 2 + r) * a.s_c + cx) = PK2(Cc[pl][0][r], Cc[pl][1][r]);
            }
    }
#undef PK2
}

void launch_addb_fused(xgpu_ctx *c, const AddbArgs &a, const DevPic &src, const DevPic &dst)
{
    const int n_ex = (a.w_scu >> 1) + 1, n_ey = (a.h_scu >> 1) + 1;
    // tile height: 4 SCU rows (128 threads, 17 KB of LDS: nine workgroups per CU in different phases of load - filter - store).  Measured against 8 rows (256
    // threads, 34 KB): 64.9 / 65.0 us at 8K, 25.0 / 26.4 us at 4K - the kernel is bound by neither VALU issue (an edge-compacted variant with 12.7 M instead of
    // 21.7 M VALU instructions took the same 66 us) nor workgroup granularity; 233 MB in 65 us is 75 % of what a plain copy kernel reaches on this part.
    const int tiles_x = (n_ex + 31) >> 5;
    hipLaunchKernelGGL(k_addb_fused<4>, dim3(tiles_x * ((n_ey + 1) >> 1)), dim3(128), 0, c->stream, a, src.y, src.u, src.v, dst.y, dst.u, dst.v, tiles_x);
}

