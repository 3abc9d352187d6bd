#!/bin/bash
# ab_trees_run.sh <rounds> <tree> [<tree> ...] -- alternate the self-contained trees under ab_trees/ (scripts/ab_tree.sh; "." = this working
# tree) over scripts/ab_step.py (whole verify forward, 8- and 64-row buckets) and scripts/attn_ab.py (attention pair alone), same box.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
ulimit -c 0
R=$1; shift
for t in "$@"; do [ "$t" = "." ] || cp scripts/ab_step.py "ab_trees/$t/scripts/ab_step.py"; done
for r in $(seq 1 "$R"); do
  for t in "$@"; do
    d=$([ "$t" = "." ] && echo . || echo "ab_trees/$t")
    echo "== $t ($(cat "$d/REV" 2>/dev/null || echo working-tree))"
    timeout 300 python "$d/scripts/ab_step.py" 7 13 30 61 2>&1 | tail -1
    timeout 300 python "$d/scripts/attn_ab.py" 800 2>&1 | tail -1
  done
done
