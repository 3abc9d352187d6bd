// l2_retain_probe.hip -- does data read by one kernel stay in the XCD L2s for the NEXT kernel on the same stream?
// Decides whether a glue kernel (RMSNorm, RoPE, attention merge: 16-512 workgroups, HBM idle) can usefully pull the head of the
// next projection's weight stream into the L2 of the XCD that will consume it.
//   reader: 256 workgroups x 512 threads, workgroup b reads slice (b + shift) % 256 of the buffer with dwordx4 loads
//           (plain or nt policy), 8 loads per lane in flight.  Workgroup b runs on XCD b % 8 (observed placement), so shift = 0
//           repeats the placement of the warming pass and shift = 1 moves every slice to another XCD (what is then still fast
//           comes from the Infinity Cache, not from L2).
//   sequence per measurement: flush (1 GiB read) -> [warm pass] -> timed pass; HIP events around the timed pass only.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/l2_retain_probe scripts/probes/l2_retain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(512) void k_read(const u4 *__restrict__ base, size_t vec_per_wg, int shift, int n_wg_data, int *__restrict__ sink) {
    const int slice = (blockIdx.x + shift) % n_wg_data;
    const u4 *p = base + (size_t)slice * vec_per_wg;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < vec_per_wg; i += 512 * 8) {
        u4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const size_t k = i + (size_t)j * 512;
            if (k < vec_per_wg) v[j] = NT ? __builtin_nontemporal_load(p + k) : p[k]; else v[j] = u4{0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345u) sink[0] = 1;
}

int main() {
    const size_t flush_bytes = 1ull << 30;
    u4 *buf, *flush; int *sink;
    hipMalloc(&buf, 256ull << 20); hipMalloc(&flush, flush_bytes); hipMalloc(&sink, 4);
    hipMemset(buf, 1, 256ull << 20); hipMemset(flush, 2, flush_bytes);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int sizes_mb[] = {4, 8, 16, 24, 32, 64, 128};
    printf("%8s %10s %10s %10s %10s %10s %10s   (us per pass, median of 9; 256 workgroups)\n", "MB", "cold", "hot", "hot-nt", "hot-shift1", "warm-nt/hot", "cold-nt");
    for (int mb : sizes_mb) {
        const size_t bytes = (size_t)mb << 20, vec_per_wg = bytes / 16 / 256;
        auto run = [&](int warm_mode, bool nt, int shift) {
            std::vector<float> t;
            for (int r = 0; r < 9; r++) {
                hipLaunchKernelGGL(k_read<false>, dim3(256), dim3(512), 0, st, flush, flush_bytes / 16 / 256, 0, 256, sink);
                if (warm_mode == 1) hipLaunchKernelGGL(k_read<false>, dim3(256), dim3(512), 0, st, buf, vec_per_wg, 0, 256, sink);
                if (warm_mode == 2) hipLaunchKernelGGL(k_read<true>, dim3(256), dim3(512), 0, st, buf, vec_per_wg, 0, 256, sink);
                hipEventRecord(e0, st);
                if (nt) hipLaunchKernelGGL(k_read<true>, dim3(256), dim3(512), 0, st, buf, vec_per_wg, shift, 256, sink);
                else hipLaunchKernelGGL(k_read<false>, dim3(256), dim3(512), 0, st, buf, vec_per_wg, shift, 256, sink);
                hipEventRecord(e1, st); hipStreamSynchronize(st);
                float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1000.f);
            }
            std::sort(t.begin(), t.end());
            return t[t.size() / 2];
        };
        const float cold = run(0, false, 0), hot = run(1, false, 0), hot_nt = run(1, true, 0), hot_s1 = run(1, false, 1), wnt = run(2, false, 0), cold_nt = run(0, true, 0);
        printf("%8d %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f\n", mb, cold, hot, hot_nt, hot_s1, wnt, cold_nt);
    }
    return 0;
}
