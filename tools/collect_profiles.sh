# After, on the GPU box:  tools/gpu_prof.sh r05 /root/repo/bench.py --no-coded --no-paths --no-projection --no-cpu --steps 10 --warmup 3
#                         tools/prof_codes.sh r05c ; tools/prof_em_coded.sh ; tools/prof_ingest.sh ; tools/prof_loo.sh
# gather gpurun_out/ into profiles/r05_* (run here, in the repository root).
set -e
python tools/summarize_profile.py r05 r05_final "10000000x1000xK10 exact, float32 kernels (bench.py --no-coded)" 10000000 1000 10 exact > /dev/null
python tools/summarize_profile.py r05c r05_coded "10000000x1000xK10 exact, tools/check_codes.py: float32 and class-coded kernels, the encoder, with the LDS counters of the coded scoring sweep (a launch of em_coded_kernel is not always the same work here: see r05_em_coded)" > /dev/null
python tools/summarize_em_coded_prof.py r05_em_coded > /dev/null
if [ -f gpurun_out/ing_ids.json ]; then python tools/summarize_ingest_prof.py gpurun_out profiles/r05_ingest > /dev/null; fi
if [ -f gpurun_out/loo_counters.txt ]; then mkdir -p profiles/r05_loo; cp gpurun_out/loo_counters.txt profiles/r05_loo/counters.txt; cp gpurun_out/r05_ids.json profiles/r05_loo/ids.json; fi
(for f in em_kernels assign_kernels codes_kernels beagle_kernels ingest inflate; do echo "== $f.hip"; python tools/kernel_resources.py wgsassign_amd/csrc/$f.hip; done) > profiles/r05_kernel_resources.txt 2>&1
cat gpurun_out/r05_ids.json
