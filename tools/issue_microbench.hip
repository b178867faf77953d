// issue_microbench.hip — what a lone wave64 pays per instruction on gfx950 (cycles, s_memtime): hipcc -O2 --offload-arch=gfx950 -o issue tools/issue_microbench.hip; ./issue [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)
#define LOOP(body, outs...) for (int it = 0; it < iters; it++) asm volatile(".rept " STR(REP) "\n" body ".endr\n" : outs)

template <int T>
__global__ void k(uint64_t *out, int iters, int seed)
{
    int a = seed + threadIdx.x, b = seed * 3 + 1, c = 5, d = 7, e = 9, f = 11, g = 13, h = 15;
    int sa = seed, sb = seed + 1;
    __shared__ int lds[256];
    lds[threadIdx.x] = (threadIdx.x * 4 + 4) & 255;
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    uint64_t w0 = wall_clock64();
    if (T == 0) LOOP("v_add_u32 %0, %0, %1\n", "+v"(a) : "v"(b));
    if (T == 1) LOOP("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n", "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(sa) : "v"(b));
    if (T == 2) LOOP("v_mad_i32_i24 %0, %0, %1, %0\n", "+v"(a) : "v"(b));
    if (T == 3) LOOP("v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n", "+v"(a) : "v"(b), "v"(c) : "vcc");
    if (T == 4) LOOP("v_cmp_gt_i32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %2, s[20:21]\n", "+v"(a) : "v"(b), "v"(c) : "s20", "s21");
    if (T == 5) LOOP("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 3\n", "+v"(a), "+s"(sa) : "v"(b) : "scc");
    if (T == 6) LOOP("v_med3_i32 %0, %0, %1, %2\n", "+v"(a) : "v"(b), "v"(c));
    if (T == 7) LOOP("v_add_u32 %0, %0, %1\n s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n", "+v"(a) : "v"(b));
    if (T == 8) LOOP("v_cmp_gt_i32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %2\n s_or_b64 exec, exec, s[20:21]\n", "+v"(a) : "v"(b), "v"(c) : "vcc", "s20", "s21");
    if (T == 9) LOOP("v_mul_i32_i24_sdwa %0, sext(%0), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n", "+v"(a) : "v"(b));
    if (T == 10) LOOP("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n", "+v"(a) : : "memory");
    if (T == 11) LOOP("s_add_u32 %0, %0, 3\n", "+s"(sa) : : "scc");
    if (T == 12) LOOP("v_add_u32 %0, %0, %1\n s_cmp_lg_u32 %2, 0\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, %1\n1:\n", "+v"(a) : "v"(b), "s"(sb) : "scc");   // taken branch (sb != 0)
    if (T == 13) LOOP("v_add_u32 %0, %0, %1\n s_cmp_eq_u32 %2, 0\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, %1\n1:\n", "+v"(a) : "v"(b), "s"(sb) : "scc");   // not taken
    if (T == 14) LOOP("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n", "+v"(a), "+v"(c) : "v"(b));  // 2 independent chains
    if (T == 15) LOOP("v_cmp_gt_i32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, %2\n1:\n s_or_b64 exec, exec, s[20:21]\n", "+v"(a) : "v"(b), "v"(c) : "vcc", "s20", "s21");
    if (T == 16) LOOP("v_lshlrev_b64 %0, 1, %0\n", "+v"(*(long long *)&out[8 + threadIdx.x]) :);
    if (T == 17) LOOP("v_cmp_gt_i32 vcc, %0, %1\n v_add_u32 %3, %3, %1\n v_cndmask_b32 %0, %0, %2, vcc\n", "+v"(a) : "v"(b), "v"(c), "v"(d) : "vcc");
    if (T == 18) LOOP("v_add_u32 %0, %0, %1\n v_readfirstlane_b32 s20, %0\n s_add_u32 s20, s20, 1\n v_add_u32 %0, s20, %0\n", "+v"(a) : "v"(b) : "s20", "scc");
    if (T == 19) LOOP("v_add_u32 %0, %0, %1\n v_mov_b32_dpp %2, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32 %0, %0, %2\n", "+v"(a), "+v"(b), "+v"(c) :);
    uint64_t t1 = __builtin_readcyclecounter();
    uint64_t w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = w1 - w0;
        out[2] = a + c + d + e + f + g + h + sa + b;
    }
}

static const int kInstr[] = {1, 8, 1, 2, 2, 2, 1, 3, 4, 1, 2, 1, 3, 4, 2, 5, 1, 3, 4, 3};
static const char *kName[] = {"dep v_add", "8 indep v_add", "dep v_mad_i24", "v_cmp vcc -> v_cndmask dep", "v_cmp sgpr -> v_cndmask dep",
                              "v_add + s_add alternating", "dep v_med3", "v_add, nop, dpp mov (dep)", "cmp, saveexec, add, or-exec", "dep sdwa mul",
                              "dep ds_read + wait", "dep s_add", "taken branch (3 instr)", "untaken branch (4 instr)", "2 indep chains",
                              "cmp,saveexec,cbranch_execz(nt),add,or", "dep v_lshlrev_b64", "cmp, indep add, cndmask", "valu->readfirstlane->salu->valu",
                              "add, dpp mov, add (dep, auto nops)"};

template <int T>
void run(uint64_t *d, int launches)
{
    const int iters = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<T><<<launches, 64>>>(d, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<T><<<launches, 64>>>(d, iters, 1);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[3];
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    const double n = (double)iters * REP * kInstr[T];
    printf("%-42s  cyc/instr %6.2f   (100MHz ticks -> ns/instr %6.2f)  event ns/instr %6.2f\n", kName[T], h[0] / n, h[1] * 10.0 / n, ms * 1e6 / n);
}

int main(int argc, char **argv)
{
    int launches = argc > 1 ? atoi(argv[1]) : 1;
    uint64_t *d;
    hipMalloc(&d, 8192);
    hipMemset(d, 0, 8192);
    printf("workgroups per launch: %d\n", launches);
    run<0>(d, launches); run<1>(d, launches); run<2>(d, launches); run<3>(d, launches); run<4>(d, launches); run<5>(d, launches);
    run<6>(d, launches); run<7>(d, launches); run<8>(d, launches); run<9>(d, launches); run<10>(d, launches); run<11>(d, launches);
    run<12>(d, launches); run<13>(d, launches); run<14>(d, launches); run<15>(d, launches); run<16>(d, launches); run<17>(d, launches);
    run<18>(d, launches); run<19>(d, launches);
    return 0;
}
