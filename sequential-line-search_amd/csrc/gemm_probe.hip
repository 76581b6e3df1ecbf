// tiny probe for PMC runs: gemm_probe M N K pad group_m  -> two launches of the plain 128x128-tile GEMM (debug tool)
#define main ubench_main
#include "ubench.hip"
#undef main
int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), pad = atoi(argv[4]), gm = atoi(argv[5]);
    CK(hipFuncSetAttribute((const void*)gemm_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    bench_gemm<false, false>(M, N, K, gm, 1, pad);
    return 0;
}
