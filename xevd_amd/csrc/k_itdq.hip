// k_itdq.hip - dequantisation + 2-D inverse DCT-II of every coded transform block of a picture, one launch.
//
// Replaces xevd_sub_block_itdq -> xevd_itdq -> xevd_dquant + xevd_itrans (src_base/xevd_itdq.c:473-621) and the
// IQT variant xevdm_itdq / xevdm_itrans (src_main/xevdm_itdq.c:708-788).
//
// MI355X mapping: the host sorts the picture's TBs by size class and cuts them into wave-sized work items
// (TbWave).  One 64-lane workgroup per item:
//   stage 1 (vertical):   lane = one COLUMN of one TB (64/W TBs side by side), coefficient rows streamed from
//                         HBM with coalesced loads, dequantised on the fly, accumulated against transform-matrix
//                         rows held in SGPRs (uniform scalar loads from constant memory); all-zero coefficient
//                         rows are skipped wave-uniformly (ballot) - most high-frequency rows are zero;
//   transpose through LDS (row stride H+1 dwords: conflict-free for both the column writes and the row reads);
//   stage 2 (horizontal): lane = one ROW of one TB, 16 outputs at a time, 64-bit accumulation in the
//                         non-IQT path exactly like the reference's s64 sums, packed 16-byte stores.
// The butterflies of the reference are an evaluation order of exact integer dot products; a direct product
// with the same matrices is bit-identical (tests/test_oracle_vs_ref.py pins the matrices and the arithmetic).
// No MFMA: north_star scopes these as integer butterflies; the kernel is bound by issue rate on large TBs and
// by HBM on small ones.
#include "xgpu_internal.h"

// transform matrices xevd_tbl_tm2..64 as int32, row-major [k][n], filled by the host from the closed form
// round(64*sqrt(2)*cos((2n+1)k*pi/2N)) (row 0 = 64), see xgpu_api.hip:init_transform_tables
__constant__ int k_tm[5460];
__host__ __device__ constexpr int tm_base(int log2n) { return log2n == 1 ? 0 : log2n == 2 ? 4 : log2n == 3 ? 20 : log2n == 4 ? 84 : log2n == 5 ? 340 : 1364; }

void upload_transform_tables(const int *tm, hipStream_t s)
{
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(k_tm), tm, sizeof(int) * 5460, 0, hipMemcpyHostToDevice, s);
}

__device__ __forceinline__ int clip16(int v) { return min(max(v, -32768), 32767); }

template <int LW, int LH>
__device__ __forceinline__ void itdq_item(const ItdqArgs &a, const TbWave wv, int *lds)
{
    constexpr int W = 1 << LW, H = 1 << LH;
    constexpr int P = 64 / W;                       // TBs per wave (W <= 64)
    constexpr int LS = H + 1;                       // LDS row stride in dwords
    const int lane = threadIdx.x;
    const int *tmh = k_tm + tm_base(LH);
    const int *tmw = k_tm + tm_base(LW);

    // ------------------------------------------------ stage 1: columns ------------------------------------
    {
        const int p = lane >> LW, j = lane & (W - 1);
        const bool valid = p < wv.count;
        const TbRec tb = a.tbs[wv.first + (valid ? p : 0)];
        // xevd_itdq.c:511-515 / xevd_dquant :480-492
        const int qp = tb.qp;
        const int sidx = qp % 6;
        // xevd_tbl_dq_scale {..,72} with tool_iqt, xevd_tbl_dq_scale_b {..,71} without (xevd_tbl.c:255-256)
        const int sbase = sidx == 0 ? 40 : sidx == 1 ? 45 : sidx == 2 ? 51 : sidx == 3 ? 57 : sidx == 4 ? 64 : (a.iqt ? 72 : 71);
        const int scale = sbase << (qp / 6);
        constexpr int odd = (LW + LH) & 1;
        const int shift = 20 - 14 - (15 - a.bd - ((LW + LH) >> 1)) + (odd ? 8 : 0);
        const long long offset = shift == 0 ? 0 : 1ll << (shift - 1);
        const long long mul = (long long)scale * (odd ? 181 : 1);
        const int16_t *src = a.coef + tb.off + j;

        int acc[H];
#pragma unroll
        for (int n = 0; n < H; n++) acc[n] = 0;
        for (int k = 0; k < H; k++) {
            int c = valid ? (int)src[k * W] : 0;
            if (__ballot(c != 0) == 0) continue;                       // whole coefficient row zero in this wave
            long long lev = ((long long)c * mul + offset) >> shift;
            const int v = (int)min(max(lev, -32768ll), 32767ll);
#pragma unroll
            for (int n = 0; n < H; n++) acc[n] = (__mul24(tmh[k * H + n], v) + acc[n]);   // |tm|<=90, |v|<2^15: exact in 24x24
        }
        int *dst = lds + lane * LS;
#pragma unroll
        for (int n = 0; n < H; n++) dst[n] = a.iqt ? clip16((acc[n] + 64) >> 7) : acc[n];
    }
    __syncthreads();

    // ------------------------------------------------ stage 2: rows ---------------------------------------
    constexpr int NC = W < 16 ? W : 16;             // outputs per chunk
    const int shift2 = a.iqt ? 12 - (a.bd - 8) : 7 + 12 - (a.bd - 8);
    for (int ri = lane; ri < P * H; ri += 64) {
        const int p = ri >> LH, r = ri & (H - 1);
        if (p >= wv.count) break;
        const TbRec tb = a.tbs[wv.first + p];
        const int *in = lds + (p * W) * LS + r;
        int16_t *out = a.resid + tb.off + r * W;
#pragma unroll 1
        for (int n0 = 0; n0 < W; n0 += NC) {
            int res[NC];
            if (a.iqt) {
                int s[NC];
#pragma unroll
                for (int n = 0; n < NC; n++) s[n] = 1 << (shift2 - 1);
                for (int k = 0; k < W; k++) {
                    const int v = in[k * LS];
#pragma unroll
                    for (int n = 0; n < NC; n++) s[n] = (__mul24(tmw[k * W + n0 + n], v) + s[n]);
                }
#pragma unroll
                for (int n = 0; n < NC; n++) res[n] = clip16(s[n] >> shift2);
            } else {
                long long s[NC];
#pragma unroll
                for (int n = 0; n < NC; n++) s[n] = 1ll << (shift2 - 1);
                for (int k = 0; k < W; k++) {
                    const int v = in[k * LS];
#pragma unroll
                    for (int n = 0; n < NC; n++) s[n] += (long long)tmw[k * W + n0 + n] * v;
                }
#pragma unroll
                for (int n = 0; n < NC; n++) res[n] = (int)min(max(s[n] >> shift2, -32768ll), 32767ll);
            }
            if constexpr (NC >= 8) {
#pragma unroll
                for (int n = 0; n < NC; n += 8) {
                    uint4 v;
                    v.x = (uint32_t)(uint16_t)res[n + 0] | ((uint32_t)(uint16_t)res[n + 1] << 16);
                    v.y = (uint32_t)(uint16_t)res[n + 2] | ((uint32_t)(uint16_t)res[n + 3] << 16);
                    v.z = (uint32_t)(uint16_t)res[n + 4] | ((uint32_t)(uint16_t)res[n + 5] << 16);
                    v.w = (uint32_t)(uint16_t)res[n + 6] | ((uint32_t)(uint16_t)res[n + 7] << 16);
                    *(uint4 *)(out + n0 + n) = v;
                }
            } else if constexpr (NC == 4) {
                uint2 v;
                v.x = (uint32_t)(uint16_t)res[0] | ((uint32_t)(uint16_t)res[1] << 16);
                v.y = (uint32_t)(uint16_t)res[2] | ((uint32_t)(uint16_t)res[3] << 16);
                *(uint2 *)(out + n0) = v;
            } else {
                *(uint32_t *)(out + n0) = (uint32_t)(uint16_t)res[0] | ((uint32_t)(uint16_t)res[1] << 16);
            }
        }
    }
}

#define ITDQ_MAX_LDS_INTS (64 * 65)

__global__ __launch_bounds__(64) void k_itdq(const ItdqArgs a)
{
    __shared__ int lds[ITDQ_MAX_LDS_INTS];
    const int wi = blockIdx.x;
    if (wi >= a.n_waves) return;
    const TbWave wv = a.waves[wi];
#define CASE(lw, lh) case (lw) * 8 + (lh): itdq_item<lw, lh>(a, wv, lds); break;
#define ROW(lw) CASE(lw, 1) CASE(lw, 2) CASE(lw, 3) CASE(lw, 4) CASE(lw, 5) CASE(lw, 6)
    switch (wv.log2w * 8 + wv.log2h) {
        ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6)
        default: break;
    }
}

void launch_itdq(xgpu_ctx *c, const ItdqArgs &a)
{
    if (a.n_waves <= 0) return;
    hipLaunchKernelGGL(k_itdq, dim3(a.n_waves), dim3(64), 0, c->stream, a);
}
