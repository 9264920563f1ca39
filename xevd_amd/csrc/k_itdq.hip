// k_itdq.hip - dequantisation + 2-D inverse DCT-II of every coded transform block of a picture, one launch.
//
// Replaces xevd_sub_block_itdq -> xevd_itdq -> xevd_dquant + xevd_itrans (src_base/xevd_itdq.c:473-621) and the
// IQT variant xevdm_itdq / xevdm_itrans (src_main/xevdm_itdq.c:708-788).
//
// MI355X mapping: the host sorts the picture's TBs by size class and cuts them into 256-thread work items
// (TbWave = a group of G same-size TBs, 4096 samples for everything up to 64x64).  Inside a workgroup
//   stage 1 (vertical):   lane = one COLUMN of one TB, wave = one chunk of 16 output rows, so the transform
//                         matrix entries a wave needs are wave-uniform and live in SGPRs (scalar loads from
//                         constant memory, packed as s16 row PAIRS); two taps per v_dot2c_i32_i16; coefficient
//                         row pairs that are zero across the whole wave are skipped (ballot) - most
//                         high-frequency rows are; dequantisation (s64 like xevd_dquant) happens on the fly;
//   transpose through LDS as packed s16 (row stride W/2+1 dwords: conflict-free column writes and row reads);
//   stage 2 (horizontal): lane = one ROW of one TB, wave = one chunk of 16 output columns.  The reference's
//                         non-IQT path keeps a 32-bit intermediate and a 64-bit sum: the intermediate t
//                         (|t| < 2^28) is split exactly into t = hi*2^15 + lo with both halves in s16 range,
//                         two dot2 chains accumulate sum(tm*hi) and sum(tm*lo) in 32 bits without overflow and
//                         the 64-bit value hi*2^15+lo is formed once per output - bit-identical to the s64 sum.
//                         IQT keeps a clipped s16 intermediate, one chain.
// The reference's partial butterflies are an evaluation order of exact integer dot products; a direct product
// with the same matrices is bit-identical (matrices and arithmetic pinned in tests/test_oracle_vs_ref.py).
// No MFMA (north_star: integer butterflies, not dense contractions); bound by VALU issue on 32/64-point TBs
// and by HBM on small ones.
#include "itdq_body.h"

template <bool IQT>
__global__ __launch_bounds__(256) void k_itdq(const ItdqArgs a)
{
    __shared__ uint32_t lds[IQT ? ITDQ_LDS_DWORDS - ITDQ_PLANES_DWORDS / 2 : ITDQ_LDS_DWORDS];
    __shared__ uint32_t s_rm[ITDQ_MAX_G], s_cm[ITDQ_MAX_G];      // per TB: coefficient row pairs / column pairs that are not all zero
    const int wi = blockIdx.x;
    if (wi >= a.n_waves) return;
    itdq_dispatch<IQT>(a, wi, lds, s_rm, s_cm);
}

int itdq_group_size(int lw, int lh) { return itdq_group(lw, lh); }

void upload_transform_tables_intra(const int *tm, const int16_t *ats, hipStream_t s);      // k_intra.hip's copy (k_intra_itdq)
void upload_transform_tables(const int *tm, const int16_t *ats, hipStream_t s)
{
    upload_transform_tables_tu(tm, ats, s);
    upload_transform_tables_intra(tm, ats, s);
}

void launch_itdq(xgpu_ctx *c, const ItdqArgs &a, hipStream_t s)
{
    if (a.n_waves <= 0) return;
    if (a.iqt) hipLaunchKernelGGL(k_itdq<true>, dim3(a.n_waves), dim3(256), 0, s, a);
    else       hipLaunchKernelGGL(k_itdq<false>, dim3(a.n_waves), dim3(256), 0, s, a);
}
