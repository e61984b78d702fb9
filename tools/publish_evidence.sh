#!/bin/bash
# Copies one evidence set from the scratch gpurun_out/ (written by tools/collect_evidence.sh TAG on the GPU box) into the
# tracked profiles/ under the round's names:   bash tools/publish_evidence.sh r02c r02
set -eu
TAG=$1
ROUND=$2
SRC=gpurun_out
DST=profiles
for f in $SRC/${TAG}_*; do
    name=$(basename "$f")
    name=${name#${TAG}_}
    case "$name" in
        *.err|sha.txt) continue ;;
        kt_bench.json) out=${ROUND}_bench_cartpole_under_rocprof.json ;;
        kt_graph_bench.json) out=${ROUND}_bench_cartpole_graph_under_rocprof.json ;;
        pmc_traffic.json) out=pmc_traffic.json ;;
        devcount_traffic.json) out=devcount_traffic.json ;;
        pmc_valu.json) out=pmc_valu.json ;;
        *) out=${ROUND}_$name ;;
    esac
    [ -d "$f" ] && continue
    cp "$f" "$DST/$out"
done
cat $SRC/${TAG}_sha.txt
