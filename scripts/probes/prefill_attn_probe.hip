// prefill_attn_probe.hip -- harness of csrc/prefill_attn_device.h outside the library: random q / K / V in the runner's layouts, the kernel's output
// against a plain fp32 causal softmax-attention on the GPU (every row, every head), and its launch time.
// usage: prefill_attn_probe <rows> [heads 32] [kv_heads 32] [pos0 0] [iters 20] [max_len 2048] [form: 4 = 4 waves x 2 fragments, 8 = 8 waves x 1, 42 = two key groups of 4 x 2]
#include "../../sam-decoding_amd/csrc/prefill_attn_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace prefillattn;
typedef _Float16 E;

// one workgroup of 128 threads per (row, head): thread d accumulates output element d over the visible keys with a two-pass softmax
__global__ void k_ref(const E *q, const E *kc, const E *vc, float *out, int rows, int pos0, int H, int Hkv, long long max_len, float scale) {
    const int row = blockIdx.x, h = blockIdx.y, d = threadIdx.x, kvh = h / (H / Hkv), nk = pos0 + row + 1;
    __shared__ float sc[4096];
    __shared__ float red[128];
    const E *qp = q + ((size_t)row * H + h) * 128;
    float mx = -INFINITY;
    for (int k = d; k < nk; k += 128) {
        const E *kp = kc + ((size_t)kvh * max_len + k) * 128;
        float s = 0.f;
        for (int j = 0; j < 128; j++) s += (float)qp[j] * (float)kp[j];
        s *= scale; sc[k] = s; mx = fmaxf(mx, s);
    }
    red[d] = mx; __syncthreads();
    for (int o = 64; o > 0; o >>= 1) { if (d < o) red[d] = fmaxf(red[d], red[d + o]); __syncthreads(); }
    mx = red[0]; __syncthreads();
    float sum = 0.f;
    for (int k = d; k < nk; k += 128) { const float p = expf(sc[k] - mx); sc[k] = p; sum += p; }
    red[d] = sum; __syncthreads();
    for (int o = 64; o > 0; o >>= 1) { if (d < o) red[d] += red[d + o]; __syncthreads(); }
    sum = red[0];
    float acc = 0.f;
    for (int k = 0; k < nk; k++) acc += sc[k] * (float)vc[((size_t)kvh * max_len + k) * 128 + d];
    out[((size_t)row * H + h) * 128 + d] = acc / sum;
}

int main(int argc, char **argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 1536, H = argc > 2 ? atoi(argv[2]) : 32, Hkv = argc > 3 ? atoi(argv[3]) : 32, pos0 = argc > 4 ? atoi(argv[4]) : 0;
    const int iters = argc > 5 ? atoi(argv[5]) : 20;
    const long long max_len = argc > 6 ? atoll(argv[6]) : 2048;
    const int waves = argc > 7 ? atoi(argv[7]) : 8;
    if (pos0 + rows > max_len || pos0 + rows > 4096) { printf("pos0 + rows must fit the cache (and 4096)\n"); return 1; }
    const size_t nq = (size_t)rows * H * 128, nkv = (size_t)Hkv * max_len * 128;
    std::vector<E> hq(nq), hk(nkv), hv(nkv);
    uint32_t s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto &x : hq) x = (E)(rnd() * 4.0f);
    for (auto &x : hk) x = (E)(rnd() * 4.0f);
    for (auto &x : hv) x = (E)(rnd() * 2.0f);
    // what lies behind the prompt in the cache must not matter: NaN there
    for (int h = 0; h < Hkv; h++) for (long long k = pos0 + rows; k < max_len; k++) for (int d = 0; d < 128; d++) { hk[((size_t)h * max_len + k) * 128 + d] = (E)NAN; hv[((size_t)h * max_len + k) * 128 + d] = (E)NAN; }
    E *q, *kc, *vc, *out; float *ref;
    CHK(hipMalloc(&q, nq * 2)); CHK(hipMalloc(&kc, nkv * 2)); CHK(hipMalloc(&vc, nkv * 2)); CHK(hipMalloc(&out, nq * 2)); CHK(hipMalloc(&ref, nq * 4));
    CHK(hipMemcpy(q, hq.data(), nq * 2, hipMemcpyHostToDevice)); CHK(hipMemcpy(kc, hk.data(), nkv * 2, hipMemcpyHostToDevice)); CHK(hipMemcpy(vc, hv.data(), nkv * 2, hipMemcpyHostToDevice));
    CHK(hipMemset(out, 0xff, nq * 2));
    const float scale = 1.0f / sqrtf(128.f);
    CHK(hipFuncSetAttribute((const void *)k_prefill_attention<F16, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CHK(hipFuncSetAttribute((const void *)k_prefill_attention<F16, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CHK(hipFuncSetAttribute((const void *)(k_prefill_attention<F16, 4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BYTES));
    const int n_blocks = (rows + QB - 1) / QB, pair = getenv("PA_PAIR") ? atoi(getenv("PA_PAIR")) : (n_blocks * H > 256);
    const dim3 grid(pair ? (n_blocks + 1) / 2 : n_blocks, H);
    auto launch = [&]() {
        if (waves == 42) hipLaunchKernelGGL((k_prefill_attention<F16, 4, 2, 2>), grid, dim3(512), 2 * LDS_BYTES, 0, q, kc, vc, out, rows, pos0, H, Hkv, max_len, scale * 1.4426950408889634f, pair);
        else if (waves == 4) hipLaunchKernelGGL((k_prefill_attention<F16, 4, 2>), grid, dim3(256), LDS_BYTES, 0, q, kc, vc, out, rows, pos0, H, Hkv, max_len, scale * 1.4426950408889634f, pair);
        else hipLaunchKernelGGL((k_prefill_attention<F16, 8, 1>), grid, dim3(512), LDS_BYTES, 0, q, kc, vc, out, rows, pos0, H, Hkv, max_len, scale * 1.4426950408889634f, pair);
    };
    launch(); CHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_ref, dim3(rows, H), dim3(128), 0, 0, q, kc, vc, ref, rows, pos0, H, Hkv, max_len, scale);
    CHK(hipDeviceSynchronize());
    std::vector<E> ho(nq); std::vector<float> hr(nq);
    CHK(hipMemcpy(ho.data(), out, nq * 2, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hr.data(), ref, nq * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale_o = 0; long bad = 0;
    for (size_t i = 0; i < nq; i++) {
        const double e = fabs((double)hr[i] - (double)(float)ho[i]);
        if (!(e <= 2e-3 + 4e-3 * fabs(hr[i]))) { if (bad < 5) printf("  mismatch row %zu head %zu d %zu: ref %f got %f\n", i / (H * 128), (i / 128) % H, i % 128, hr[i], (float)ho[i]); bad++; }
        if (e > worst) worst = e; if (fabs(hr[i]) > scale_o) scale_o = fabs(hr[i]);
    }
    printf("check: %ld of %zu outputs off; worst |d| %.5f at scale %.3f\n", bad, nq, worst, scale_o);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    for (int it = 0; it < iters; it++) launch();
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    double fl = 0; for (int r = 0; r < rows; r++) fl += 4.0 * 128 * (pos0 + r + 1);
    fl *= H;
    printf("rows %d heads %d/%d pos0 %d waves %d: %d workgroups: %.1f us, %.3f PFLOP/s (causal flops)\n", rows, H, Hkv, pos0, waves, grid.x * grid.y, ms * 1e3, fl / ms / 1e12);
    return 0;
}
