R=$GRAFT_REPO_ROOT
cd $R
bash tools/collect_profiles.sh round4_b ba9480b
cd $R
timeout -k 5 900 python bench.py --gpus 2 --steps 40 > gpurun_out/round4_b_bench_gpus2_on_one_device.json 2> /dev/null; tail -c 1500 gpurun_out/round4_b_bench_gpus2_on_one_device.json
