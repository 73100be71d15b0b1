# rocprofv3 kernel trace of the device-resident BGZF ingest (tools/bench_reader.py --lowdepth --device) with the ids of the code it ran;
# run on the GPU box: bash tools/prof_ingest.sh [sites] [individuals]; then python tools/summarize_ingest_prof.py gpurun_out profiles/r04_ingest
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
S=${1:-400000}; N=${2:-2000}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
python3 -c "import sys, json; sys.path.insert(0, '$R'); from wgsassign_amd import _lib; l = _lib.load(); print(json.dumps({'build_id': l.wgs_build_id().decode(), 'kernels_id': l.wgs_kernels_id().decode(), 'ingest_kernels_id': l.wgs_ingest_kernels_id().decode(), 'sites': $S, 'individuals': $N}))" > $R/gpurun_out/ing_ids.json || exit 1
cd /tmp
rm -rf $R/gpurun_out/ing_kt
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ing_kt -- python3 $R/tools/bench_reader.py --inds $N --sites $S --lowdepth --device --only-device-inflate > $R/gpurun_out/ing_kt.log 2>&1 || exit 1
timeout -k 10 300 python3 $R/tools/bench_reader.py --inds $N --sites $S --lowdepth --device --only-device-inflate > $R/gpurun_out/ing_plain.log 2>&1 || exit 1
echo "profiled ingest"
