#!/bin/bash
# The native Beagle reader (host code, threads) under AddressSanitizer + UBSan and under ThreadSanitizer, on the CPU:
#   bash tools/reader_sanitize.sh [sites] [individuals]
# Generates one plain-gzip and one BGZF file, then for each: index pass (with and without names), indexed opens at three
# rows with 1 / 2 / 7 / 16 threads (full reads, checksums must agree with the single-threaded one), plain open + skip.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d)
python3 - "$D" "${1:-60000}" "${2:-60}" <<PY
import sys
sys.path.insert(0, "$R/tests"); sys.path.insert(0, "$R/tools")
import synth, bench_cli
d, m, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
L, IDs = synth.make_beagle(m, n, 3, seed=3)
bench_cli.write_beagle(d + "/a.gz", L, d + "/ids", IDs, "gzip")
bench_cli.write_beagle(d + "/b.gz", L, d + "/ids", IDs, "bgzf")
PY
for san in address,undefined thread; do
    g++ -std=c++17 -O1 -g -fsanitize=$san -fno-omit-frame-pointer -I"$R/include" -o "$D/drv" "$R/tests/c_abi/reader_sanitize.cpp" \
        "$R/wgsassign_amd/csrc/reader.cpp" -lz -lpthread -ldl
    for f in a.gz b.gz; do
        for span in 40000 3000000; do
            echo "== -fsanitize=$san $f span=$span"
            "$D/drv" "$D/$f" $span
        done
    done
done
rm -rf "$D"
echo "reader: clean under ASan+UBSan and TSan"
