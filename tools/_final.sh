python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5
timeout 700 python tools/fuzz_parity.py 400 8107 > gpurun_out/fuzz_8107.log 2>&1; tail -4 gpurun_out/fuzz_8107.log
bash tools/profile_round.sh r02 > gpurun_out/profile_r02.log 2>&1; tail -8 gpurun_out/profile_r02.log
python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err; tail -1 gpurun_out/bench_r02.json
