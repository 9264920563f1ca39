// micro-benchmark (round 6): issue cost of the VALU instructions k_addb_alf is made of, gfx950.  Eight independent chains per lane, 4 waves per SIMD-quarter resident:
// cycles per wave-instruction and SIMD = what one instruction costs a kernel that is bound by VALU issue.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N_IT 2048
#define CHAIN8(ASM) \
    for (int i = 0; i < N_IT; i++) { \
        asm volatile(ASM : "+v"(r0) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r1) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r2) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r3) : "v"(a), "v"(b)); \
        asm volatile(ASM : "+v"(r4) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r5) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r6) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r7) : "v"(a), "v"(b)); }
#define CHAIN8_64(ASM) \
    for (int i = 0; i < N_IT; i++) { \
        asm volatile(ASM : "+v"(q0) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(q1) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(q2) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(q3) : "v"(a), "v"(b)); \
        asm volatile(ASM : "+v"(q4) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(q5) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(q6) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(q7) : "v"(a), "v"(b)); }
template <int OP> __global__ void k(int *out, int a0, int b0)
{
    int a = a0 + threadIdx.x, b = b0;
    int r0 = 0, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;
    unsigned long long q0 = 0, q1 = 1, q2 = 2, q3 = 3, q4 = 4, q5 = 5, q6 = 6, q7 = 7;
    if (OP == 0) CHAIN8("v_add_u32 %0, %0, %1")
    if (OP == 1) CHAIN8("v_mul_lo_u32 %0, %0, %1")
    if (OP == 2) CHAIN8("v_mul_u32_u24 %0, %0, %1")
    if (OP == 3) CHAIN8("v_mad_i32_i16 %0, %1, %2, %0")
    if (OP == 4) CHAIN8("v_pk_add_u16 %0, %0, %1")
    if (OP == 5) CHAIN8("v_dot2_u32_u16 %0, %1, %2, %0")
    if (OP == 6) CHAIN8("v_dot2c_i32_i16 %0, %1, %2")
    if (OP == 7) CHAIN8("v_sad_u16 %0, %1, %2, %0")
    if (OP == 8) CHAIN8("v_med3_i32 %0, %0, %1, %2")
    if (OP == 9) CHAIN8("v_cvt_pk_i16_i32 %0, %0, %1")
    if (OP == 10) CHAIN8("v_perm_b32 %0, %0, %1, %2")
    if (OP == 11) CHAIN8("v_lshl_or_b32 %0, %0, %1, %2")
    if (OP == 12) CHAIN8("v_pk_max_i16 %0, %0, %1")
    if (OP == 13) CHAIN8_64("v_lshl_add_u64 %0, %0, 1, %0")
    if (OP == 14) CHAIN8_64("v_lshrrev_b64 %0, %1, %0")
    if (OP == 15) CHAIN8("v_mad_u32_u24 %0, %0, %1, %2")
    if (OP == 16) CHAIN8("v_ashrrev_i32 %0, %1, %0")
    if (OP == 17) CHAIN8("v_bfe_u32 %0, %0, %1, %2")
    if (OP == 18) CHAIN8("v_cndmask_b32 %0, %0, %1, vcc")
    if (OP == 19) CHAIN8("v_alignbit_b32 %0, %0, %1, 16")
    if (OP == 20) CHAIN8("v_pk_mul_lo_u16 %0, %0, %1")
    if (OP == 21) CHAIN8("v_pk_mad_i16 %0, %1, %2, %0")
    if (OP == 22) CHAIN8("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
    if (OP == 100) CHAIN8("v_and_b32 %0, %0, %1")
    if (OP == 101) CHAIN8("v_or_b32 %0, %0, %1")
    if (OP == 102) CHAIN8("v_xor_b32 %0, %0, %1")
    if (OP == 103) CHAIN8("v_lshlrev_b32 %0, %1, %0")
    if (OP == 104) CHAIN8("v_lshrrev_b32 %0, %1, %0")
    if (OP == 105) CHAIN8("v_min_i32 %0, %0, %1")
    if (OP == 106) CHAIN8("v_max_i32 %0, %0, %1")
    if (OP == 107) CHAIN8("v_sub_u32 %0, %0, %1")
    if (OP == 108) CHAIN8("v_mov_b32 %0, %1")
    if (OP == 109) CHAIN8("v_add3_u32 %0, %0, %1, %2")
    if (OP == 110) CHAIN8("v_bfi_b32 %0, %0, %1, %2")
    if (OP == 111) CHAIN8("v_and_or_b32 %0, %0, %1, %2")
    if (OP == 112) CHAIN8("v_lshl_add_u32 %0, %0, 1, %1")
    if (OP == 113) CHAIN8("v_add_lshl_u32 %0, %0, %1, 1")
    if (OP == 114) CHAIN8("v_add_u16 %0, %0, %1")
    if (OP == 115) CHAIN8("v_max_i16 %0, %0, %1")
    if (OP == 116) CHAIN8("v_min_u16 %0, %0, %1")
    if (OP == 117) CHAIN8("v_pk_min_u16 %0, %0, %1")
    if (OP == 118) CHAIN8("v_pk_ashrrev_i16 %0, %1, %0")
    if (OP == 119) CHAIN8("v_mad_u16 %0, %0, %1, %2")
    if (OP == 120) CHAIN8("v_mul_i32_i24 %0, %0, %1")
    if (OP == 121) CHAIN8("v_add_co_u32 %0, vcc, %0, %1")
    if (OP == 122) CHAIN8("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %1, %2, vcc")
    if (OP == 123) CHAIN8("v_add_u32 %0, s4, %0")
    if (OP == 124) CHAIN8("v_add_u32 %0, 0x12345, %0")
    if (OP == 125) CHAIN8("v_mul_lo_u32 %0, %0, %1")
    if (OP == 126) CHAIN8("v_mul_hi_u32 %0, %0, %1")
    if (OP == 127) CHAIN8("v_sub_u16 %0, %0, %1")
    if (OP == 128) CHAIN8("v_ashrrev_i16 %0, %1, %0")
    if (OP == 129) CHAIN8("v_lshlrev_b16 %0, %1, %0")
    if (OP == 130) CHAIN8("v_mul_lo_u16 %0, %0, %1")
    if (OP == 131) CHAIN8("v_mad_i32_i24 %0, %0, %1, %2")
    if (OP == 132) CHAIN8("v_sad_u32 %0, %0, %1, %2")
    if (OP == 133) CHAIN8("v_min3_i32 %0, %0, %1, %2")
    if (OP == 134) CHAIN8("v_add_f32 %0, %0, %1")
    if (OP == 135) CHAIN8("v_fma_f32 %0, %0, %1, %2")
    if (OP == 136) CHAIN8("v_pk_fma_f16 %0, %0, %1, %2")
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (int)(q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7);
}
template <int OP> void run(const char *name)
{
    int *d; hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 3, 5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 3, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = 1024.0 * 4 * N_IT * 8 / 1024;          // 1024 workgroups x 4 waves over 1024 SIMDs
    printf("%-22s %8.3f ms   %6.2f ns per wave-instruction and SIMD  (= %.2f x v_add_u32 slots if that is 4 cycles)\n", name, ms, ms * 1e6 / wave_instr_per_simd, 0.0);
    hipFree(d);
}
int main()
{
    run<0>("v_add_u32"); run<1>("v_mul_lo_u32"); run<2>("v_mul_u32_u24"); run<3>("v_mad_i32_i16"); run<4>("v_pk_add_u16"); run<5>("v_dot2_u32_u16"); run<6>("v_dot2c_i32_i16");
    run<7>("v_sad_u16"); run<8>("v_med3_i32"); run<9>("v_cvt_pk_i16_i32"); run<10>("v_perm_b32"); run<11>("v_lshl_or_b32"); run<12>("v_pk_max_i16"); run<13>("v_lshl_add_u64");
    run<14>("v_lshrrev_b64"); run<15>("v_mad_u32_u24"); run<16>("v_ashrrev_i32"); run<17>("v_bfe_u32"); run<18>("v_cndmask_b32"); run<19>("v_alignbit_b32"); run<20>("v_pk_mul_lo_u16");
    run<21>("v_pk_mad_i16"); run<22>("v_mov_b32_dpp");
    run<100>("v_and_b32"); run<101>("v_or_b32"); run<102>("v_xor_b32"); run<103>("v_lshlrev_b32"); run<104>("v_lshrrev_b32"); run<105>("v_min_i32"); run<106>("v_max_i32"); run<107>("v_sub_u32"); run<108>("v_mov_b32"); run<109>("v_add3_u32"); run<110>("v_bfi_b32"); run<111>("v_and_or_b32"); run<112>("v_lshl_add_u32"); run<113>("v_add_lshl_u32"); run<114>("v_add_u16"); run<115>("v_max_i16"); run<116>("v_min_u16"); run<117>("v_pk_min_u16"); run<118>("v_pk_ashrrev_i16"); run<119>("v_mad_u16"); run<120>("v_mul_i32_i24"); run<121>("v_add_co_u32"); run<122>("v_cmp+cndmask"); run<123>("v_add_u32 sgpr"); run<124>("v_add_u32 const"); run<125>("v_mul_lo_u32 (rpt)"); run<126>("v_mul_hi_u32"); run<127>("v_sub_u16"); run<128>("v_ashrrev_i16"); run<129>("v_lshlrev_b16"); run<130>("v_mul_lo_u16"); run<131>("v_mad_i32_i24"); run<132>("v_sad_u32"); run<133>("v_min3_i32"); run<134>("v_readfirstlane-free v_add_f32"); run<135>("v_fma_f32"); run<136>("v_pk_add_f32?");
    return 0;
}
