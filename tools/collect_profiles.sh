# After tools/gpu_prof.sh r04 ... && tools/prof_encode.sh && tools/prof_loo.sh && tools/gpu_prof.sh r04c ... && tools/prof_ingest.sh on the GPU box:
# gather gpurun_out/ into profiles/r04_* (run here, in the repository root).
set -e
python tools/summarize_profile.py r04 r04_final "10000000x1000xK10 exact, float32 kernels (bench.py --no-coded)" 10000000 1000 10 exact > /dev/null
python tools/summarize_profile.py r04c r04_coded "10000000x1000xK10 exact, all legs of bench.py (float32 and class-coded kernels, the encoder; a launch of a kernel is not always the same work here: see r04_final for the headline kernel)" > /dev/null
python tools/summarize_encode_prof.py > /dev/null
python tools/summarize_ingest_prof.py gpurun_out profiles/r04_ingest > /dev/null
cp gpurun_out/r04_ids.json profiles/r04_encoder/ids.json
cp gpurun_out/r04_ids.json profiles/r04_loo/ids.json
python tools/sum_counters.py gpurun_out/loo_sq gpurun_out/loo_sq2 > profiles/r04_loo/counters.txt
(for f in em_kernels assign_kernels codes_kernels beagle_kernels ingest inflate; do echo "== $f.hip"; python tools/kernel_resources.py wgsassign_amd/csrc/$f.hip; done) > profiles/r04_kernel_resources.txt 2>&1
cat gpurun_out/r04_ids.json
