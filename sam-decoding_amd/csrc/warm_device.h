// warm_device.h -- L2 warm-up of the NEXT projection's weight stream from inside a glue launch (gfx950).
//
// A decoder layer at <= 64 draft rows is four weight streams (k_gemm_skinny) with glue launches in between (RMSNorm, RoPE + K/V
// write, tree attention, its merge) during which HBM idles.  scripts/probes/l2_retain_probe.hip: what a kernel reads with plain
// loads STAYS in the XCD L2s for the next kernel on the stream (32 MB re-read by the same workgroup -> XCD placement: at the
// launch floor; shifted by one XCD: Infinity-Cache speed; warmed with nt loads: cold), and nt loads hit lines that are there.
// So a glue launch carries extra workgroups that do nothing but read the head of the next projection's stream -- the first
// `bytes` of every k_gemm_skinny workgroup's run of 64 KiB blocks -- on the XCD that will consume it: workgroup `lin` of the
// projection (lin = column tile + tiles * split, dispatched x-fastest) runs on XCD (lin + o) % 8, where the offset o is the same for
// every launch of a stream segment -- 0 for plain launches, another constant inside a hipGraph replay (scripts/probes/xcc_probe.hip:
// 7 for all nine launches of a layer-shaped sequence) -- so the warm workgroup with linear id m in ITS launch warms projection
// workgroups lin == m (mod 8).  Observed placement, used for speed only: a wrong guess costs the hit, never correctness.  The
// projection then starts on L2 hits while its HBM requests for the rest of the stream are already queued.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct WarmArgs {
    const char *w;          // packed weights of the next projection (samd_gemm_pack_weights), NULL = nothing to warm
    int tiles;              // its column tiles (N / 128) = gridDim.x of its launch
    int n_chunks;           // K / 256: 64 KiB blocks per column tile
    int splits;             // its split-K factor = gridDim.y
    int bytes;              // bytes to warm per projection workgroup (a multiple of 2048)
    int delay;              // s_sleep units (64 cycles each) before the first load: the glue workgroups' own loads queue first
};

#ifdef SAMD_HIP_H
static inline WarmArgs warm_args(const samd_warm_t *next) {
    WarmArgs a; a.w = nullptr; a.tiles = a.n_chunks = a.splits = a.bytes = a.delay = 0;
    if (next && next->d_packed_w && next->kb_per_workgroup > 0 && next->N >= 128 && next->K >= 256 && next->splits >= 1) {
        a.w = (const char *)next->d_packed_w; a.tiles = next->N / 128; a.n_chunks = next->K / 256; a.splits = next->splits;
        a.bytes = (next->kb_per_workgroup * 1024) & ~2047;
        a.delay = next->delay < 0 ? 0 : next->delay;
    }
    return a;
}
#endif

// number of warm workgroups a glue launch adds (one per workgroup of the projection)
static inline int warm_blocks(const WarmArgs &a) { return a.w && a.bytes > 0 ? a.tiles * a.splits : 0; }

// body of warm workgroup `index` (0 .. warm_blocks - 1); every thread of the block takes part.  Returns a value that depends on
// every loaded byte: the caller stores it under a condition that never holds, which keeps the loads alive without a sink buffer.
__device__ __forceinline__ unsigned warm_next_projection(const WarmArgs &a, int index) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    // this workgroup's own linear id in its launch decides the XCD it runs on, and the projection's workgroup with the same id mod 8
    // will run on the same one (see the header comment); 8 consecutive warm workgroups cover the 8 residues
    const int mylin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int lin = (mylin & 7) + 8 * (index >> 3);
    if (lin >= a.tiles * a.splits) return 0u;
    for (int d = a.delay; d > 0; d--) __builtin_amdgcn_s_sleep(1);
    const int tile = lin % a.tiles, split = lin / a.tiles;
    const int c0 = (int)((long long)split * a.n_chunks / a.splits), c1 = (int)((long long)(split + 1) * a.n_chunks / a.splits);
    long long bytes = (long long)(c1 - c0) * 65536;
    bytes = bytes < a.bytes ? bytes : a.bytes;
    const u4 *p = reinterpret_cast<const u4 *>(a.w + ((size_t)tile * a.n_chunks + c0) * 65536);
    const int n16 = (int)(bytes >> 4), T = blockDim.x * blockDim.y;
    const int tid = threadIdx.x + threadIdx.y * blockDim.x;
    unsigned acc = 0;
    for (int i = tid; i < n16; i += 8 * T) {
        u4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (i + j * T < n16) ? p[i + j * T] : u4{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; j++) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    return acc;
}
