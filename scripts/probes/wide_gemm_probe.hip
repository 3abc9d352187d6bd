// wide_gemm_probe.hip -- harness of the hand-written prefill GEMM (wide_gemm_device.h; an experiment of round 5, NOT part of the library:
// profiles/r05_wide_gemm.md says why): launched (mode 1) stream-K over a persistent grid, or (mode 0) with whole tiles only, one range per workgroup
// round-robin, to see the kernel's rate without the decomposition; checked against a plain fp32 GPU product on a sample of the outputs.
// usage: wide_gemm_probe <M> <N> <K> [mode: 0 = whole tiles, 1 = stream-K] [iters] [check] [epilogue: 0 plain, 1 silu(gate) * up]
#include "wide_gemm_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace widegemm;
typedef _Float16 E;

// whole tiles, round by round over a persistent grid
__global__ __launch_bounds__(512, 1) void k_wide_dp(const E *__restrict__ A, const E *__restrict__ W, E *__restrict__ C, int M, int N, int K, int tiles_m, int n_tiles) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    for (int id0 = blockIdx.x; id0 < n_tiles; id0 += gridDim.x) {
        const int round = id0 / gridDim.x, in_round = n_tiles - round * gridDim.x < (int)gridDim.x ? n_tiles - round * gridDim.x : gridDim.x;
        const int id = round * gridDim.x + xcd_logical(blockIdx.x, in_round);
        const int tn = id / tiles_m, tm = id % tiles_m;
        floatx4 acc[8][4];
        zero_acc(acc);
        fragment<F16>(A, W + (size_t)tn * WT * K, M, K, tm * AT, 128, 64, 0, K / BK, acc, lds_base);
        store_tile<F16, EPI_PLAIN>(C, M, N, tn * WT, tm * AT, acc);
    }
}

__global__ void k_ref(const E *A, const E *W, float *C, int M, int N, int K, int m_step, int n_step, int row_off) {
    const int n = (blockIdx.x * blockDim.x + threadIdx.x) * n_step, m = blockIdx.y * m_step;
    if (n >= N || m >= M) return;
    float s = 0.f;
    for (int k = 0; k < K; k++) s += (float)A[(size_t)m * K + k] * (float)W[(size_t)(n + row_off) * K + k];
    C[(size_t)(m / m_step) * (N / n_step + 1) + n / n_step] = s;
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1024, N = argc > 2 ? atoi(argv[2]) : 16384, K = argc > 3 ? atoi(argv[3]) : 4096;
    const int mode = argc > 4 ? atoi(argv[4]) : 1, iters = argc > 5 ? atoi(argv[5]) : 20, check = argc > 6 ? atoi(argv[6]) : 1, epi = argc > 7 ? atoi(argv[7]) : 0;
    if (N % WT || K % BK || (epi && mode == 0)) { printf("N %% 256, K %% 64; the epilogue needs mode 1\n"); return 1; }
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<E> hA((size_t)M * K), hW((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto &x : hA) x = (E)(rnd() * 2.0f);
    for (auto &x : hW) x = (E)(rnd() * 0.1f);
    E *A, *W[3], *C; float *R, *R2;
    CHK(hipMalloc(&A, hA.size() * 2)); CHK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    for (int i = 0; i < 3; i++) { CHK(hipMalloc(&W[i], hW.size() * 2)); CHK(hipMemcpy(W[i], hW.data(), hW.size() * 2, hipMemcpyHostToDevice)); }
    const int ldc = epi ? N / 2 : N;
    CHK(hipMalloc(&C, (size_t)M * ldc * 2)); CHK(hipMemset(C, 0xff, (size_t)M * ldc * 2));
    const int tiles_m = (M + AT - 1) / AT, tiles_n = N / WT, n_tiles = tiles_m * tiles_n;
    char *ws; const size_t ws_head = 4096 * 4 + 64, ws_bytes = ws_head + (size_t)cus * 2 * SLOT_FLOATS * 4;
    CHK(hipMalloc(&ws, ws_bytes)); CHK(hipMemset(ws, 0, ws_head));
    Args a;
    a.A = A; a.C = C; a.counters = (int *)ws; a.fault = a.counters + 4096; a.slots = (float *)(ws + ws_head);
    a.M = M; a.N = N; a.K = K; a.tiles_m = tiles_m; a.tiles_n = tiles_n; a.KT = K / BK; a.ldc = ldc; a.iters = (long long)n_tiles * a.KT;
    CHK(hipFuncSetAttribute((const void *)k_wide_dp, hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES));
    CHK(hipFuncSetAttribute((const void *)k_wide_gemm<F16, EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES));
    CHK(hipFuncSetAttribute((const void *)k_wide_gemm<F16, EPI_SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES));
    const int grid_sk = a.iters < cus ? (int)a.iters : cus;
    if (getenv("WIDE_PROBE_ADDR")) { printf("A %p..%p W0 %p..%p C %p..%p ws %p..%p slots %p\n", (void *)A, (void *)(A + hA.size()), (void *)W[0], (void *)(W[0] + hW.size()), (void *)C, (void *)(C + (size_t)M * ldc), (void *)ws, (void *)(ws + ws_bytes), (void *)a.slots); fflush(stdout); }
    auto launch = [&](int it) {
        if (mode == 0) hipLaunchKernelGGL(k_wide_dp, dim3(n_tiles < cus ? n_tiles : cus), dim3(512), RING_BYTES, 0, A, W[it % 3], C, M, N, K, tiles_m, n_tiles);
        else {
            a.W = W[it % 3];
            if (epi) hipLaunchKernelGGL((k_wide_gemm<F16, EPI_SILU>), dim3(grid_sk), dim3(512), RING_BYTES, 0, a);
            else hipLaunchKernelGGL((k_wide_gemm<F16, EPI_PLAIN>), dim3(grid_sk), dim3(512), RING_BYTES, 0, a);
        }
    };
    launch(0); CHK(hipDeviceSynchronize());
    if (check) {
        const int m_step = 7, n_step = 5, NO = epi ? N / 2 : N;
        const int rm = (M + m_step - 1) / m_step, rn = NO / n_step + 1;
        CHK(hipMalloc(&R, (size_t)rm * rn * 4)); CHK(hipMalloc(&R2, (size_t)rm * rn * 4));
        hipLaunchKernelGGL(k_ref, dim3((NO / n_step + 255) / 256 + 1, rm), dim3(256), 0, 0, A, W[0], R, M, NO, K, m_step, n_step, 0);
        if (epi) hipLaunchKernelGGL(k_ref, dim3((NO / n_step + 255) / 256 + 1, rm), dim3(256), 0, 0, A, W[0], R2, M, NO, K, m_step, n_step, NO);
        std::vector<float> hR((size_t)rm * rn), hR2((size_t)rm * rn); std::vector<E> hC((size_t)M * ldc);
        CHK(hipMemcpy(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost));
        if (epi) CHK(hipMemcpy(hR2.data(), R2, hR2.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0; long bad = 0, seen = 0;
        for (int m = 0; m < M; m += m_step) for (int n = 0; n < NO; n += n_step) {
            float ref = hR[(size_t)(m / m_step) * rn + n / n_step];
            if (epi) { const float g = (float)(E)ref, u = (float)(E)hR2[(size_t)(m / m_step) * rn + n / n_step]; ref = (float)(E)(g / (1.f + expf(-g))) * u; }
            const float got = (float)hC[(size_t)m * ldc + n];
            const double e = fabs((double)ref - got); seen++;
            if (!(e <= 4e-3 * fabs(ref) + 3e-2)) { if (bad < 5) printf("  mismatch m %d n %d: ref %f got %f\n", m, n, ref, got); bad++; }
            if (e > worst) worst = e; if (fabs(ref) > scale) scale = fabs(ref);
        }
        int fault = 0; CHK(hipMemcpy(&fault, a.fault, 4, hipMemcpyDeviceToHost));
        std::vector<int> cnt(n_tiles); CHK(hipMemcpy(cnt.data(), a.counters, n_tiles * 4, hipMemcpyDeviceToHost));
        long dirty = 0; for (int x : cnt) dirty += x != 0;
        printf("check: %ld of %ld sampled outputs off; worst |d| %.4f at scale %.2f; fault word %d, %ld counters left non-zero\n", bad, seen, worst, scale, fault, dirty);
    }
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    for (int it = 0; it < iters; it++) launch(it + 1);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    const double fl = 2.0 * M * N * K;
    printf("M %d N %d K %d mode %d epi %d: %d tiles on %d CUs (%.2f rounds): %.1f us, %.3f PFLOP/s\n", M, N, K, mode, epi, n_tiles, cus, (double)n_tiles / cus, ms * 1e3, fl / ms / 1e12);
    return 0;
}
