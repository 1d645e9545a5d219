#!/bin/bash
# Collects the per-round profile set on the GPU box (run from the repo root through gpurun):
#   bash tools/collect_profiles.sh r03      -> gpurun_out/profiles_r03/*   (copy what is to be judged into profiles/)
tag=${1:-rXX}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd $root
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/bench.err
python bench.py --scheduler ddpm --ddim-steps 1000 --points 4096 --grasps 200 --clouds-per-gpu 64 --steps 3 --warmup 1 > $out/${tag}_bench_c5.json 2>> $out/bench.err
python bench.py --scheduler ddpm --ddim-steps 1000 --points 4096 --grasps 200 --clouds-per-gpu 64 --steps 3 --warmup 1 --minimal --noise kernel > $out/${tag}_bench_c5_noise_in_kernel.json 2>> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_s1 -o run -- /usr/bin/python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --streams 1 --minimal > $out/${tag}_bench_streams1_under_rocprof.json 2> $out/prof.err
cd $root
python3 - $out $tag <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/prof_s1/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(f)))
with open(f"{out}/{tag}_bench_streams1_kernel_stats.csv", "w", newline="") as g:
    w = csv.writer(g)
    for r in rows:
        r[0] = r[0][:140]
        w.writerow(r)
PY
bash tools/pmc_denoise.sh $tag > $out/pmc.log 2>&1
cp gpurun_out/pmc_$tag.json $out/${tag}_denoise_pmc.json
cp gpurun_out/pmc_$tag.txt $out/${tag}_denoise_pmc_counters.txt
rm -rf $out/prof_s1
head -c 600 $out/${tag}_bench.json; echo; head -c 300 $out/${tag}_bench_c5.json; echo; head -5 $out/${tag}_bench_streams1_kernel_stats.csv | cut -c1-160
# round 4 additions: the ppc experiment, the decoder alone, the set-abstraction kernels and the encoder kernels under counters
cd $root
python bench.py --experiment ppc --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_ppc.json 2>> $out/bench.err
bash tools/pmc_kernels.sh ${tag}_sa tools/run_sa_once.py 256 > $out/pmck_sa.log 2>&1
cp gpurun_out/pmck_${tag}_sa.json $out/${tag}_point_ops_pmc_raw.json
bash tools/prof_kernels.sh ${tag}_enc tools/bench_encoders.py --shipped --only "PVCNNEncoder(fpc)" --batch-sizes 256 --iterations 5 > $out/${tag}_encoder_kernels.txt 2>&1
bash tools/prof_kernels.sh ${tag}_ssg tools/bench_encoders.py --only PointNet2 --batch-sizes 256 --iterations 5 > $out/${tag}_shootout_ssg_kernels.txt 2>&1
bash tools/prof_kernels.sh ${tag}_pvcnn2 tools/bench_encoders.py --only PVCNN2 --batch-sizes 256 --iterations 5 > $out/${tag}_shootout_pvcnn2_kernels.txt 2>&1
# the shipped encoder's kernels under counters (matrix-pipe busy, effective clock, waits, LDS, L2: profiles/<tag>_encoder_pmc.*)
PMCK_MORE=1 bash tools/pmc_kernels.sh ${tag}_enc tools/run_encoder_once.py 256 3 > $out/${tag}_encoder_pmc.txt 2>&1
cp gpurun_out/pmck_${tag}_enc.json $out/${tag}_encoder_pmc.json
