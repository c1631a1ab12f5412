tag=r04c
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd $root
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $out/${tag}_pytest_gpu.txt
bash scripts/prof.sh ${tag}_cornell1080p --workload cornell > $out/${tag}_prof_cornell.log 2>&1
python bench.py --workload cornell --steps 3 --warmup 1 > $out/${tag}_bench_cornell1080p.json 2> $out/${tag}_bench_cornell.err
bash scripts/all_workloads.sh > $out/${tag}_all_workloads.txt 2>&1
echo "== cornell" > $out/${tag}_full_frame_parity.txt
timeout 900 python scripts/full_frame_parity.py cornell >> $out/${tag}_full_frame_parity.txt 2>&1
cat $out/${tag}_pytest_gpu.txt $out/${tag}_all_workloads.txt; tail -5 $out/${tag}_full_frame_parity.txt
