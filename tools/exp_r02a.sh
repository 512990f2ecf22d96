set -e
mkdir -p gpurun_out/r02a
timeout -k 10 300 python tools/layer_report.py --reps 5 > gpurun_out/r02a/layer_base.txt 2>&1
timeout -k 10 200 python tools/layer_report.py --reps 3 --tile $((8<<16)) > gpurun_out/r02a/layer_noA.txt 2>&1
timeout -k 10 200 python tools/layer_report.py --reps 3 --tile $((16<<16)) > gpurun_out/r02a/layer_noB.txt 2>&1
timeout -k 10 200 python tools/layer_report.py --reps 3 --tile $((32<<16)) > gpurun_out/r02a/layer_noEpi.txt 2>&1
timeout -k 10 200 python tools/layer_report.py --reps 3 --tile $((1<<16)) > gpurun_out/r02a/layer_Azero.txt 2>&1
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r02a/bench.json 2>gpurun_out/r02a/bench.err
tail -1 gpurun_out/r02a/layer_base.txt
