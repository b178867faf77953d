// op_rate_microbench.hip — what each VALU opcode class costs a SIMD on gfx950 when 1, 2 or 4 waves share it.
// hipcc -O2 --offload-arch=gfx950 -o op_rate tools/op_rate_microbench.hip; ./op_rate
// Output per op: cycles of SIMD time per wave-instruction (launch time x 2.4 GHz / instructions per wave / waves per SIMD)
// for 1024 / 2048 / 4096 single-wave workgroups (= 1 / 2 / 4 waves per SIMD on 1024 SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)
#define LOOP(body) for (int it = 0; it < iters; it++) asm volatile(".rept " STR(REP) "\n" body "\n.endr\n" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "s"(sb) : "vcc", "s20", "s21", "v10", "v11")

#define OPS(X) \
    X(0, "v_add_u32 %0, %0, %4") \
    X(1, "v_sub_u32 %0, %0, %4") \
    X(2, "v_and_b32 %0, %0, %4") \
    X(3, "v_xor_b32 %0, %0, %4") \
    X(4, "v_lshlrev_b32 %0, 1, %0") \
    X(5, "v_ashrrev_i32 %0, 1, %0") \
    X(6, "v_max_i32 %0, %0, %4") \
    X(7, "v_min_u32 %0, %0, %4") \
    X(8, "v_mul_i32_i24 %0, %0, %4") \
    X(9, "v_mul_u32_u24 %0, %0, %4") \
    X(10, "v_mad_i32_i24 %0, %0, %4, %0") \
    X(11, "v_mad_u32_u24 %0, %0, %4, %0") \
    X(12, "v_add3_u32 %0, %0, %4, %0") \
    X(13, "v_lshl_add_u32 %0, %0, 1, %4") \
    X(14, "v_and_or_b32 %0, %0, %4, %0") \
    X(15, "v_bfe_i32 %0, %0, 0, 17") \
    X(16, "v_med3_i32 %0, %0, -1, 1") \
    X(17, "v_cndmask_b32 %0, %0, %4, vcc") \
    X(18, "v_cmp_gt_i32 vcc, %0, %4") \
    X(19, "v_mul_lo_u32 %0, %0, %4") \
    X(20, "v_mov_b32 %0, %0") \
    X(21, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    X(22, "v_add_u32_dpp %0, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    X(23, "v_ffbh_u32 %0, %0") \
    X(24, "v_alignbit_b32 %0, %0, %4, 3") \
    X(25, "v_mul_hi_u32 %0, %0, %4") \
    X(26, "v_lshlrev_b64 v[10:11], 1, v[10:11]") \
    X(27, "v_pk_add_i16 %0, %0, %4") \
    X(28, "v_pk_mad_i16 %0, %0, %4, %0") \
    X(29, "v_pk_max_i16 %0, %0, %4") \
    X(30, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4") \
    X(31, "v_mad_i32_i24 %0, %0, %4, %0\n v_mad_i32_i24 %1, %1, %4, %1\n v_mad_i32_i24 %2, %2, %4, %2\n v_mad_i32_i24 %3, %3, %4, %3") \
    X(32, "v_cmp_gt_i32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc") \
    X(33, "v_cmp_gt_i32 s[20:21], %0, %4\n v_cndmask_b32 %0, %0, %4, s[20:21]") \
    X(34, "v_sub_u32 %0, %4, %0\n v_mul_i32_i24 %1, %0, %4\n v_add_u32 %0, %1, %0") \
    X(35, "v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %4\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %4") \
    X(36, "v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4") \
    X(37, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3") \
    X(38, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc") \
    X(39, "v_sub_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD") \
    X(40, "v_dot2_i32_i16 %0, %0, %4, %0") \
    X(41, "v_bfi_b32 %0, %0, %4, %0") \
    X(42, "v_sub_u32 %0, %0, %4\n s_nop 0") \
    X(43, "v_add_u32 %0, %0, %4\n s_add_u32 s20, s20, 3") \
    X(44, "v_mad_i32_i24 %0, %0, %4, %0\n s_add_u32 s20, s20, 3") \
    X(45, "ds_bpermute_b32 %0, %0, %0\n s_waitcnt lgkmcnt(0)") \
    X(46, "v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0") \
    X(47, "v_sad_u32 %0, %0, %4, %0") \
    X(48, "v_lshrrev_b32 %0, 1, %0") \
    X(49, "v_lshrrev_b32 %0, %4, %0") \
    X(50, "v_ashrrev_i32 %0, %4, %0") \
    X(51, "v_lshlrev_b32 %0, %4, %0") \
    X(52, "v_or_b32 %0, %0, %4") \
    X(53, "v_not_b32 %0, %0") \
    X(54, "v_subrev_u32 %0, %0, %4") \
    X(55, "v_bfe_u32 %0, %0, 3, 9") \
    X(56, "v_cmp_gt_i32 vcc, %0, %4\n v_addc_co_u32 %0, vcc, 0, %0, vcc") \
    X(57, "v_cmp_gt_i32 vcc, %0, %4\n v_subb_co_u32 %0, vcc, %0, %4, vcc") \
    X(58, "v_perm_b32 %0, %0, %4, %0") \
    X(59, "v_xad_u32 %0, %0, %4, %0") \
    X(60, "v_lshl_or_b32 %0, %0, 1, %4") \
    X(61, "v_mul_i32_i24 %0, %0, %4\n v_add_u32 %1, %1, %4")

static const int kInstr[] = {1,1,1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,1,1, 4,4,2,2,3,4,4,4,4,1, 1,1,2,2,2,2,2,1, 1,1,1,1,1,1,1,1,2,2, 1,1,1,2};

template <int T>
__global__ void k(uint64_t *out, int iters, int seed)
{
    int a = seed + threadIdx.x, b = seed * 3 + 1, c = 5, d = 7, e = 9;
    int sb = seed + 1;
    (void)sb;
#define X(N, BODY) if (T == N) LOOP(BODY);
    OPS(X)
#undef X
    if (threadIdx.x == 0 && a + c + d + e == 0x12345) out[0] = a;
}

static const char *kName[] = {
#define X(N, BODY) BODY,
    OPS(X)
#undef X
};

template <int T>
void run(uint64_t *d)
{
    const int iters = 100;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double cyc[4];
    const int grids[4] = {1024, 2048, 4096, 8192};
    for (int g = 0; g < 4; g++) {
        k<T><<<grids[g], 64>>>(d, 5, 1);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 3; r++) {
            hipEventRecord(e0);
            k<T><<<grids[g], 64>>>(d, iters, 1);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double n = (double)iters * REP * kInstr[T];
        cyc[g] = best * 1e6 * 2.4 / n / (grids[g] / 1024);
    }
    char nm[80];
    int i = 0;
    for (const char *p = kName[T]; *p && i < 70; p++) nm[i++] = *p == '\n' ? ';' : *p;
    nm[i] = 0;
    printf("%-72s %6.2f %6.2f %6.2f %6.2f\n", nm, cyc[0], cyc[1], cyc[2], cyc[3]);
    fflush(stdout);
}

template <int T>
void run_all(uint64_t *d)
{
    if constexpr (T < 62) {
        run<T>(d);
        run_all<T + 1>(d);
    }
}

int main(int argc, char **)
{
    uint64_t *d;
    hipMalloc(&d, 8192);
    hipMemset(d, 0, 8192);
    printf("SIMD cycles (at 2.4 GHz) per wave-instruction with 1 / 2 / 4 / 8 waves per SIMD\n");
    if (argc > 1)
        run_all<48>(d);  // only the opcodes added in the second batch
    else
        run_all<0>(d);
    return 0;
}
