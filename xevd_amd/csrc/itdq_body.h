// itdq_body.h - the device side of the residual pass (dequantisation + inverse transforms of one 256-thread work item), shared by k_itdq.hip (the pass as
// a launch of its own) and k_intra.hip (k_intra_itdq: the NEXT picture's residual pass rides in the workgroups of this picture's data-flow intra launch).
// The transform matrices are constant memory of the including translation unit: each unit uploads its own copy (upload_transform_tables_tu).
// What is restated and how it is mapped: see the header of k_itdq.hip.
#pragma once
#include "xgpu_internal.h"

typedef short v2s __attribute__((ext_vector_type(2)));

// transform matrices xevd_tbl_tm2..64 packed as row pairs: entry [k2][n] = (tm[2*k2][n], tm[2*k2+1][n]) as two s16;
// filled by the host from the closed form (xgpu_api.hip: init_transform_tables).  Offsets: N*N/2 dwords per size.
static __constant__ uint32_t k_tmp[2730];
__host__ __device__ constexpr int tmp_base(int log2n) { return log2n == 1 ? 0 : log2n == 2 ? 2 : log2n == 3 ? 10 : log2n == 4 ? 42 : log2n == 5 ? 170 : 682; }

// ATS matrices, same packing, [type DST7=0 / DCT8=1][size 4,8,16,32]: offsets 8, 32, 128, 512 dwords per size.
// The 4-point entries hold the 4x4 matrices equivalent to the reference's factorised kernels (xevdm_itdq.c:163-190, 284-312).
static __constant__ uint32_t k_atsp[2][680];
__host__ __device__ constexpr int atsp_base(int log2n) { return log2n == 2 ? 0 : log2n == 3 ? 8 : log2n == 4 ? 40 : 168; }

static void upload_transform_tables_tu(const int *tm, const int16_t *ats, hipStream_t s)      // this translation unit's copy of the tables
{
    // ats: [type DST7=0/DCT8=1][log2n 2..5] row-major s16 matrices M[k][n] back to back (16+64+256+1024 per type)
    static uint32_t apk[2][680];
    for (int t = 0; t < 2; t++) {
        int src = 0;
        for (int l = 2; l <= 5; l++) {
            const int N = 1 << l, dst = atsp_base(l);
            for (int k2 = 0; k2 < N / 2; k2++)
                for (int n = 0; n < N; n++)
                    apk[t][dst + k2 * N + n] = (uint32_t)(uint16_t)ats[t * 1360 + src + (2 * k2) * N + n] |
                                               ((uint32_t)(uint16_t)ats[t * 1360 + src + (2 * k2 + 1) * N + n] << 16);
            src += N * N;
        }
    }
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(k_atsp), apk, sizeof(apk), 0, hipMemcpyHostToDevice, s);
    // tm: int32 row-major matrices 2,4,..,64 back to back (5460 entries)
    static uint32_t packed[2730];
    int src = 0;
    for (int l = 1; l <= 6; l++) {
        const int N = 1 << l, dst = tmp_base(l);
        for (int k2 = 0; k2 < N / 2; k2++)
            for (int n = 0; n < N; n++)
                packed[dst + k2 * N + n] = (uint32_t)(uint16_t)(int16_t)tm[src + (2 * k2) * N + n] |
                                           ((uint32_t)(uint16_t)(int16_t)tm[src + (2 * k2 + 1) * N + n] << 16);
        src += N * N;
    }
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(k_tmp), packed, sizeof(packed), 0, hipMemcpyHostToDevice, s);
    (void)hipStreamSynchronize(s);
}

__device__ __forceinline__ int clip16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b), c, false);
}

// geometry of a size class, shared with the host-side batch builder (xgpu_builder.hip)
__host__ __device__ constexpr int itdq_group(int lw, int lh)
{
    const int W = 1 << lw, H = 1 << lh;
    const int l1 = W * (H > 16 ? H / 16 : 1), l2 = H * (W > 16 ? W / 16 : 1);
    return 256 / (l1 > l2 ? l1 : l2);
}

#define ITDQ_PLANES_DWORDS 4608  // max over size classes of 2 planes x G*H*(W/2+1) dwords (16x16: 2*2304)
#define ITDQ_LDS_DWORDS (ITDQ_PLANES_DWORDS + 2048)   // + 4096 dequantised s16 coefficients
#define ITDQ_MAX_G 128           // TBs per work item (2x2 chroma blocks)

// OR over the 64 lanes of the wave (every lane must take part)
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v |= (uint32_t)__shfl_xor((int)v, m, 64);
    return v;
}

// Sparsity masks (round 3): a coded block of a real stream holds a handful of non-zero coefficients at low frequencies.  Stage 0 records, per TB, which
// coefficient ROW pairs and COLUMN pairs hold a non-zero value (LDS atomic OR, only for the few non-zero dwords); stage 1 then walks only the set row
// pairs (union over the wave's TBs, a scalar bit loop - the matrix rows stay wave-uniform SGPR loads) and stage 2 only the set column pairs: a column
// without coefficients stays zero through the vertical transform.  The round-2 loop tested every row pair with an LDS read + ballot + branch - 32
// dependent LDS round trips per stage for a 64-point transform, which is what the kernel's time was (the arithmetic itself is a few dozen dot2).
// IQT: the sequence uses the 16-bit two-stage transforms (sps->tool_iqt) - every work item keeps a clipped s16 intermediate, so only ONE intermediate plane
// exists in LDS (17 KB instead of 27 KB per workgroup) and the 32-bit split path is not compiled in: more workgroups per CU for a kernel that is bound by the
// latency of its dependent loads, not by arithmetic.
template <int LW, int LH, bool IQT>
__device__ __forceinline__ void itdq_item(const ItdqArgs &a, const TbWave wv, uint32_t *lds, uint32_t *s_rm, uint32_t *s_cm)
{
    constexpr int W = 1 << LW, H = 1 << LH;
    constexpr int N1 = H > 16 ? 16 : H, C1 = H / N1;          // stage-1 outputs per lane, chunks
    constexpr int N2 = W > 16 ? 16 : W, C2 = W / N2;
    constexpr int G = itdq_group(LW, LH);
    constexpr int RS = W / 2 + 1;                              // LDS row stride in dwords (s16 pairs), odd
    constexpr int PLANE = G * H * RS;                          // dwords per intermediate plane
    constexpr bool UNI1 = (G * W) % 64 == 0, UNI2 = (G * H) % 64 == 0;
    static_assert(2 * PLANE <= ITDQ_PLANES_DWORDS && G * W * H <= 4096 && G <= ITDQ_MAX_G, "LDS budget");
    constexpr int PLANES_DWORDS = IQT ? ITDQ_PLANES_DWORDS / 2 : ITDQ_PLANES_DWORDS;
    constexpr bool RMASK = H >= 16, CMASK = W >= 16;          // shorter transforms: the masks would cost more than the 2..4 loop rounds they can save
    const int t = threadIdx.x;
    if (RMASK || CMASK) {
        if (t < G) { s_rm[t] = 0; s_cm[t] = 0; }
        __syncthreads();
    }
    // wave-uniform matrix choice: DCT-II, or for ATS work items (4..32 only) DST-VII / DCT-VIII
    const uint32_t *tmh = (wv.tr_v == TR_DCT2 || LH < 2 || LH > 5) ? k_tmp + tmp_base(LH) : k_atsp[wv.tr_v - 1] + atsp_base(LH);
    const uint32_t *tmw = (wv.tr_h == TR_DCT2 || LW < 2 || LW > 5) ? k_tmp + tmp_base(LW) : k_atsp[wv.tr_h - 1] + atsp_base(LW);
    const bool s16_mid = IQT || wv.tr_v != TR_DCT2 || wv.tr_h != TR_DCT2;     // ATS keeps a clipped s16 intermediate like IQT (:406-421)
    int16_t *ldsh = (int16_t *)lds;                            // plane 0: hi (or the IQT intermediate), plane 1: lo
    int16_t *ldsl = (int16_t *)(lds + PLANE);
    uint32_t *ldsc = lds + PLANES_DWORDS;                      // dequantised coefficients, [p][row][col] s16

    // ------------------------------------------------ stage 0: load + dequantise ---------------------------
    // all coefficients of the G blocks in one coalesced sweep (one memory round trip for the whole work item),
    // dequantised once (xevd_dquant, xevd_itdq.c:480-492; shift/offset :511-515; scale tables xevd_tbl.c:255-256:
    // {..,72} with tool_iqt, {..,71} without) and parked in LDS as s16
    {
        constexpr int S = W * H, UN = S >= 8 ? 8 : 4;           // samples per load unit
        constexpr int odd = (LW + LH) & 1;
        const int shift = 20 - 14 - (15 - a.bd - ((LW + LH) >> 1)) + (odd ? 8 : 0);
        const long long offset = shift == 0 ? 0 : 1ll << (shift - 1);
        for (int u = t; u < G * S / UN; u += 256) {
            const int p = (u * UN) / S, o = (u * UN) % S;
            if (p >= wv.count) break;
            const TbRec tb = a.tbs[wv.first + p];
            const int qp = tb.qp, sidx = qp % 6;
            const int sbase = sidx == 0 ? 40 : sidx == 1 ? 45 : sidx == 2 ? 51 : sidx == 3 ? 57 : sidx == 4 ? 64 : (IQT ? 72 : 71);
            const long long mul = (long long)(sbase << (qp / 6)) * (odd ? 181 : 1);
            // row-major TB with row stride 2^log2s (a sub-block of a >64 CU keeps the CU's stride, xevd_itdq.c:573-584)
            const int16_t *src = a.coef + tb.off + ((o >> LW) << tb.log2s) + (o & (W - 1));
            uint32_t raw[UN / 2];
            if constexpr (UN == 8) { const uint4 v = *(const uint4 *)src; raw[0] = v.x; raw[1] = v.y; raw[2] = v.z; raw[3] = v.w; }
            else { const uint2 v = *(const uint2 *)src; raw[0] = v.x; raw[1] = v.y; }
#pragma unroll
            for (int i = 0; i < UN / 2; i++) {
                if (raw[i] == 0) continue;
                const int c0 = (int16_t)(raw[i] & 0xFFFF), c1 = (int16_t)(raw[i] >> 16);
                const long long l0 = ((long long)c0 * mul + offset) >> shift, l1 = ((long long)c1 * mul + offset) >> shift;
                const int v0 = (int)min(max(l0, -32768ll), 32767ll), v1 = (int)min(max(l1, -32768ll), 32767ll);
                raw[i] = (uint32_t)(uint16_t)v0 | ((uint32_t)(uint16_t)v1 << 16);
                if (RMASK) atomicOr(&s_rm[p], 1u << ((o + 2 * i) >> (LW + 1)));                  // row pair of this dword
                if (CMASK) atomicOr(&s_cm[p], 1u << (((o + 2 * i) & (W - 1)) >> 1));              // column pair
            }
#pragma unroll
            for (int i = 0; i < UN / 2; i++) ldsc[(p * S + o) / 2 + i] = raw[i];
        }
    }
    __syncthreads();

    // ------------------------------------------------ stage 1: columns ------------------------------------
    uint32_t rows1 = 0;
    if (RMASK) {
        const int p1 = (t % (G * W)) >> LW;
        rows1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_or((t < G * W * C1 && p1 < wv.count) ? s_rm[p1] : 0u));
    }
    if (t < G * W * C1) {
        const int idx = t % (G * W);
        int chunk = t / (G * W);
        if (UNI1) chunk = __builtin_amdgcn_readfirstlane(chunk);
        const int p = idx >> LW, j = idx & (W - 1);
        const bool valid = p < wv.count;
        const int16_t *src = (const int16_t *)ldsc + p * (W * H) + j;

        int acc[N1];
#pragma unroll
        for (int n = 0; n < N1; n++) acc[n] = 0;
        if (RMASK) {
            for (uint32_t m = rows1; m; m &= m - 1) {          // the row pairs that hold a coefficient in one of this wave's TBs
                const int k2 = __builtin_ctz(m);
                const uint32_t vp = valid ? ((uint32_t)(uint16_t)src[(2 * k2) * W] | ((uint32_t)(uint16_t)src[(2 * k2 + 1) * W] << 16)) : 0u;
                const uint32_t *row = tmh + k2 * H + chunk * N1;
#pragma unroll
                for (int n = 0; n < N1; n++) acc[n] = dot2(row[n], vp, acc[n]);
            }
        } else {
            for (int k2 = 0; k2 < H / 2; k2++) {
                const uint32_t vp = valid ? ((uint32_t)(uint16_t)src[(2 * k2) * W] | ((uint32_t)(uint16_t)src[(2 * k2 + 1) * W] << 16)) : 0u;
                if (__ballot(vp != 0) == 0) continue;              // both coefficient rows zero across this wave
                const uint32_t *row = tmh + k2 * H + chunk * N1;
#pragma unroll
                for (int n = 0; n < N1; n++) acc[n] = dot2(row[n], vp, acc[n]);
            }
        }
        // transposed store: element [p][row = chunk*N1+n][col = j] - only columns stage 2 will read (its column pair holds a coefficient)
        const int base = (p * H + chunk * N1) * (2 * RS) + j;
        if (CMASK && !(valid && ((s_cm[p] >> (j >> 1)) & 1))) {
        } else if (s16_mid) {
#pragma unroll
            for (int n = 0; n < N1; n++) ldsh[base + n * (2 * RS)] = (int16_t)clip16((acc[n] + 64) >> 7);   // xevdm_itdq.c ITX_SHIFT1 = 7
        } else {
#pragma unroll
            for (int n = 0; n < N1; n++) {
                ldsh[base + n * (2 * RS)] = (int16_t)(acc[n] >> 15);          // |acc| < 2^28 -> hi in s16 range
                ldsl[base + n * (2 * RS)] = (int16_t)(acc[n] & 0x7FFF);
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------ stage 2: rows ---------------------------------------
    uint32_t cols2 = 0;
    if (CMASK) {
        const int p2 = (t % (G * H)) >> LH;
        cols2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_or((t < G * H * C2 && p2 < wv.count) ? s_cm[p2] : 0u));
    }
    if (t < G * H * C2) {
        const int idx = t % (G * H);
        int chunk = t / (G * H);
        if (UNI2) chunk = __builtin_amdgcn_readfirstlane(chunk);
        const int p = idx >> LH, r = idx & (H - 1);
        if (p >= wv.count) return;
        // a column pair outside this TB's own mask was not written by stage 1 (it may be set for another TB of the wave): reads as zero
        const uint32_t own = CMASK ? s_cm[p] : 0xFFFFFFFFu;
        const TbRec tb = a.tbs[wv.first + p];
        const int shift2 = s16_mid ? 12 - (a.bd - 8) : 7 + 12 - (a.bd - 8);
        const uint32_t *inh = lds + (p * H + r) * RS;
        const uint32_t *inl = inh + PLANE;
        int res[N2];
        if (s16_mid) {
            int s[N2];
#pragma unroll
            for (int n = 0; n < N2; n++) s[n] = 1 << (shift2 - 1);
            if (CMASK) {
                for (uint32_t m = cols2; m; m &= m - 1) {
                    const int k2 = __builtin_ctz(m);
                    const uint32_t vp = ((own >> k2) & 1) ? inh[k2] : 0u;
                    const uint32_t *row = tmw + k2 * W + chunk * N2;
#pragma unroll
                    for (int n = 0; n < N2; n++) s[n] = dot2(row[n], vp, s[n]);
                }
            } else {
                for (int k2 = 0; k2 < W / 2; k2++) {
                    const uint32_t vp = inh[k2];
                    if (__ballot(vp != 0) == 0) continue;
                    const uint32_t *row = tmw + k2 * W + chunk * N2;
#pragma unroll
                    for (int n = 0; n < N2; n++) s[n] = dot2(row[n], vp, s[n]);
                }
            }
#pragma unroll
            for (int n = 0; n < N2; n++) res[n] = clip16(s[n] >> shift2);
        } else {
            int sh[N2], sl[N2];
#pragma unroll
            for (int n = 0; n < N2; n++) { sh[n] = 0; sl[n] = 0; }
            if (CMASK) {
                for (uint32_t m = cols2; m; m &= m - 1) {
                    const int k2 = __builtin_ctz(m);
                    const bool mine = (own >> k2) & 1;
                    const uint32_t vh = mine ? inh[k2] : 0u, vl = mine ? inl[k2] : 0u;
                    const uint32_t *row = tmw + k2 * W + chunk * N2;
#pragma unroll
                    for (int n = 0; n < N2; n++) { sh[n] = dot2(row[n], vh, sh[n]); sl[n] = dot2(row[n], vl, sl[n]); }
                }
            } else {
                for (int k2 = 0; k2 < W / 2; k2++) {
                    const uint32_t vh = inh[k2], vl = inl[k2];
                    if (__ballot((vh | vl) != 0) == 0) continue;
                    const uint32_t *row = tmw + k2 * W + chunk * N2;
#pragma unroll
                    for (int n = 0; n < N2; n++) { sh[n] = dot2(row[n], vh, sh[n]); sl[n] = dot2(row[n], vl, sl[n]); }
                }
            }
            const long long add = 1ll << (shift2 - 1);
#pragma unroll
            for (int n = 0; n < N2; n++) {
                const long long s = (long long)sh[n] * 32768 + sl[n] + add;     // == the reference's s64 sum + rounding offset
                res[n] = (int)min(max(s >> shift2, -32768ll), 32767ll);
            }
        }
        int16_t *out = a.resid + tb.off + (r << tb.log2s) + chunk * N2;
        if constexpr (N2 >= 8) {
#pragma unroll
            for (int n = 0; n < N2; n += 8) {
                uint4 v;
                v.x = (uint32_t)(uint16_t)res[n + 0] | ((uint32_t)(uint16_t)res[n + 1] << 16);
                v.y = (uint32_t)(uint16_t)res[n + 2] | ((uint32_t)(uint16_t)res[n + 3] << 16);
                v.z = (uint32_t)(uint16_t)res[n + 4] | ((uint32_t)(uint16_t)res[n + 5] << 16);
                v.w = (uint32_t)(uint16_t)res[n + 6] | ((uint32_t)(uint16_t)res[n + 7] << 16);
                *(uint4 *)(out + n) = v;
            }
        } else if constexpr (N2 == 4) {
            uint2 v;
            v.x = (uint32_t)(uint16_t)res[0] | ((uint32_t)(uint16_t)res[1] << 16);
            v.y = (uint32_t)(uint16_t)res[2] | ((uint32_t)(uint16_t)res[3] << 16);
            *(uint2 *)out = v;
        } else {
            *(uint32_t *)out = (uint32_t)(uint16_t)res[0] | ((uint32_t)(uint16_t)res[1] << 16);
        }
    }
}

// one work item; lds = ITDQ_LDS_DWORDS dwords (IQT: minus half the planes), s_rm / s_cm = ITDQ_MAX_G dwords each; all 256 threads of the workgroup call it
template <bool IQT>
__device__ __forceinline__ void itdq_dispatch(const ItdqArgs &a, int wi, uint32_t *lds, uint32_t *s_rm, uint32_t *s_cm)
{
    const TbWave wv = a.waves[wi];
#define CASE(lw, lh) case (lw) * 8 + (lh): itdq_item<lw, lh, IQT>(a, wv, lds, s_rm, s_cm); break;
#define ROW(lw) CASE(lw, 1) CASE(lw, 2) CASE(lw, 3) CASE(lw, 4) CASE(lw, 5) CASE(lw, 6)
    switch (wv.log2w * 8 + wv.log2h) {
        ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6)
        default: break;
    }
#undef ROW
#undef CASE
}
