// Stand-alone ablation harness for the f32-MFMA GEMM: times the collab forward shape with parts of
// the kernel compiled out (-DABL_NOSTORE / -DABL_NOGLOAD / -DABL_NOBARRIER ...).  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DABL_...] gemm_ablate.cpp -o gemm_ablate
#include "../../plnlp_amd/csrc/gemm_f32.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    int64_t M = argc > 1 ? atol(argv[1]) : 235868, N = argc > 2 ? atol(argv[2]) : 256, K = argc > 3 ? atol(argv[3]) : 512;
    float *a, *b, *c;
    hipMalloc(&a, M * K * 4); hipMalloc(&b, N * K * 4); hipMalloc(&c, M * N * 4);
    std::vector<float> h(M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(a, h.data(), M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, h.data(), N * K * 4, hipMemcpyHostToDevice);
    plnlp_gemm_operand seg{a, K, b, K, K};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) plnlp_gemm_f32(&seg, 1, 0, 1, c, N, M, N, nullptr, 1, nullptr, 0, nullptr);
    hipEventRecord(e0);
    const int it = 10;
    for (int i = 0; i < it; ++i) plnlp_gemm_f32(&seg, 1, 0, 1, c, N, M, N, nullptr, 1, nullptr, 0, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("M=%ld N=%ld K=%ld  %.4f ms  %.1f TFLOP/s\n", (long)M, (long)N, (long)K, ms, 2.0 * M * N * K / ms / 1e9);
    return 0;
}
