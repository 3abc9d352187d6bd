"""update_walk_pmc.py <walk_summary.json> -- refresh profiles/walk_pmc.json (bench.py's committed fallback / cross-check of roofline.traffic)
from the summary scripts/pmc_walk.sh wrote on the GPU box"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = json.load(open(sys.argv[1]))
b = s.get("bench_roofline", {})
out = {
    "kernel": "k_static_walk<8, 2> through samd_static_lookup_batch (round 6: EDGE BLOCKS with fail headers + hot words, DISPLACED bits, bigram table, "
              "flagged chain words, 8 waves per SIMD; the launch stores every stream's (index, length); every table at 16 slots per entry)",
    "round": 6,
    "config": {"corpus_tokens": 1 << 22, "streams": b.get("streams", 1 << 20), "tokens_per_stream": b.get("tokens_per_stream", 16)},
    "fetch_bytes_per_launch": s["FETCH_SIZE"] * 1024.0,
    "write_bytes_per_launch": s["WRITE_SIZE"] * 1024.0,
    "fetch_size_correction": 1.0,
    "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/pmc_walk.sh); FETCH_SIZE is exact for this access pattern "
            "(profiles/r01_hbm_probe.md) and agrees with TCC_EA0_RDREQ_sum x 64 B; WRITE_SIZE = the 8 MB of results the launch stores; "
            "kernel_source_sha16 = sha256 of csrc/sam_kernels.hip + sam_device.h + samd_common.h at collection time (bench.py nulls roofline.traffic when the tree differs)",
    "rdreq_per_launch": s.get("TCC_EA0_RDREQ_sum"),
    "rdreq_per_visited_state": round(s["TCC_EA0_RDREQ_sum"] / b["visited_states"], 4) if b.get("visited_states") and s.get("TCC_EA0_RDREQ_sum") else None,
    "tcc_hit": s.get("TCC_HIT_sum"), "tcc_miss": s.get("TCC_MISS_sum"), "tcp_tcc_read_req": s.get("TCP_TCC_READ_REQ_sum"),
    "sq_wave_cycles": s.get("SQ_WAVE_CYCLES"), "sq_wait_any": s.get("SQ_WAIT_ANY"), "sq_insts_vmem_rd": s.get("SQ_INSTS_VMEM_RD"),
    # the duration convention: WARM launches (all but the first of the probe run) from the kernel trace; avg_kernel_ns = their mean
    "avg_kernel_ns": float(s["warm_kernel_ns"]["mean"]) if s.get("warm_kernel_ns") else float(s["kernel_stats"]["AverageNs"]),
    "min_kernel_ns": float(s["warm_kernel_ns"]["min"]) if s.get("warm_kernel_ns") else float(s["kernel_stats"]["MinNs"]),
    "cold_first_launch_ns": s.get("warm_kernel_ns", {}).get("cold_first"), "warm_launches": s.get("warm_kernel_ns", {}).get("launches"),
    "duration_convention": "warm launches under rocprofv3 --kernel-trace (the probe's first, cold launch excluded); bench.py times 20 launches with HIP events after one warm-up launch",
    "kernel_source_sha16": s["kernel_source_sha16"],
    # the commit whose tree holds the kernel sources with that hash (the HEAD this file was refreshed at)
    "commit": __import__("subprocess").run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None,
}
json.dump(out, open(os.path.join(ROOT, "profiles", "walk_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
