"""walk_sweep.py -- the SAM traversal kernel (k_static_walk through samd_static_lookup_batch) over BASELINE.md section 3's corpus sizes
and over a Zipfian corpus (VERDICT r04 #3c): frac of the HBM peak, HBM requests per visited state (rocprofv3 FETCH_SIZE / WRITE_SIZE,
separate child passes), derived-table bytes and the bigram table's slots per pair actually obtained under its memory cap.
usage (GPU box): python3 scripts/walk_sweep.py [out.json] [--sizes 20,22,24] [--dists markov,zipf] [--slots 16,4] [--no-pmc]
Writes the JSON (profiles/walk_sweep.json is what bench.py attaches as roofline.corpus_sweep) and prints a markdown table."""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
import samd_hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?", default=os.path.join(ROOT, "gpurun_out", "walk_sweep.json"))
ap.add_argument("--sizes", default="20,22,24")
ap.add_argument("--dists", default="markov,zipf")
ap.add_argument("--slots", default="16,4")
ap.add_argument("--streams", type=int, default=1 << 20)
ap.add_argument("--tokens", type=int, default=16)
ap.add_argument("--no-pmc", action="store_true")
args = ap.parse_args()

rows = []
for dist in args.dists.split(","):
    for lg in [int(x) for x in args.sizes.split(",")]:
        n_tok = 1 << lg
        t0 = time.perf_counter()
        flat, off, docs = bench.synth_corpus(n_tok) if dist == "markov" else bench.synth_corpus_zipf(n_tok)
        auto = samd_hip.StaticAutomaton.build_flat(flat, off, bench.EOS, samd_hip.KIND_COUNT).upload()
        build_s = time.perf_counter() - t0
        info = auto.info()
        e = auto.export()
        deg, length = e["deg"], e["length"]
        profile = {"non_branching": round(float((deg <= 1).mean()), 4), "deg_gt_5": round(float((deg > 5).mean()), 5),
                   "hubs_deg_ge_100_at_depth_1_2_3": [int(((length == d) & (deg >= 100)).sum()) for d in (1, 2, 3)],
                   "max_deg_at_depth_2_3": [int(deg[length == d].max()) for d in (2, 3)]}
        del e
        for slots in [int(x) for x in args.slots.split(",")]:
            cdf = None if dist == "markov" else bench.zipf_cdf(bench.VOCAB)
            roof, _ = bench.walk_roofline(auto, docs, np.random.default_rng(7), args.streams, args.tokens, 20, n_tok, slots_per_pair=slots, noise_cdf=cdf)
            d = roof["derived_bytes"]
            pairs = None
            row = {"tokens": n_tok, "dist": dist, "states": int(info["n_states"]), "image_bytes": int(info["device_bytes"]),
                   "slots_per_pair_asked": slots, "bigram_slots": int(d["bigram_slots"]), "derived_bytes": int(d["chain_bytes"] + d["bigram_bytes"] + d["topk_count_bytes"] + d.get("edge_table_bytes", 0) + d.get("hot_word_bytes", 0) + d.get("edge_block_bytes", 0)),
                   "edge_table_bytes": int(d.get("edge_table_bytes", 0)), "hot_word_bytes": int(d.get("hot_word_bytes", 0)), "edge_block_bytes": int(d.get("edge_block_bytes", 0)),
                   "edge_block_states": int(d.get("edge_block_states", 0)),
                   "bigram_bytes": int(d["bigram_bytes"]), "launch_ms": roof["launch_ms"], "frac": roof["frac"], "achieved_gbps": roof["achieved"],
                   "visited_states": roof["visited_states"], "visits_per_token": round(roof["visited_states"] / (args.streams * args.tokens), 3),
                   "transitions_per_s": roof["transitions_per_s"], "degree_profile": profile, "build_s": round(build_s, 1)}
            if not args.no_pmc:
                det = {}
                live, how = bench.live_walk_traffic(n_tok, args.streams, args.tokens, timeout_s=400, dist=dist, slots_per_pair=slots, detail=det)
                row["traffic"] = live
                row["fetch_bytes"], row["write_bytes"] = det.get("FETCH_SIZE"), det.get("WRITE_SIZE")
                row["req_per_visit"] = round(live / 64.0 / roof["visited_states"], 4) if live else None
                row["traffic_over_algorithmic"] = round(live / roof["alg_bytes_per_launch"], 3) if live else None
                row["frac_of_request_ceiling"] = round(live / 64.0 / (roof["launch_ms"] * 1e-3) / bench.HBM_REQUESTS_PER_S, 4) if live else None
                if live is None:
                    row["pmc_note"] = how
            rows.append(row)
            print(json.dumps(row), flush=True)
        del auto
        torch.cuda.empty_cache()

h = hashlib.sha256()
for name in ("sam_kernels.hip", "sam_device.h", "samd_common.h"):
    h.update(open(os.path.join(ROOT, "sam-decoding_amd", "csrc", name), "rb").read())
out = {"kernel_source_sha16": h.hexdigest()[:16], "streams": args.streams, "tokens_per_stream": args.tokens,
       "entry_point": "samd_static_lookup_batch (the launch stores every stream's (index, length))", "rows": rows}
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(out, open(args.out, "w"), indent=1)
print("| corpus | dist | states | slots/pair asked -> got | derived MB | launch ms | frac | visits/token | req/visit | traffic/alg |")
print("|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    pairs_got = "-"
    print(f"| 2^{int(np.log2(r['tokens']))} | {r['dist']} | {r['states']} | {r['slots_per_pair_asked']} -> {r['bigram_slots']} slots | {r['derived_bytes'] / 1e6:.0f} | "
          f"{r['launch_ms']:.4f} | {r['frac']:.4f} | {r['visits_per_token']} | {r.get('req_per_visit')} | {r.get('traffic_over_algorithmic')} |")
