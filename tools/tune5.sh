set -e
for b in 1 2 4; do python tools/tune_gemm.py --batch $b --cfgs 17,18,19 --out gpurun_out/tune5_b$b.json > gpurun_out/tune5_b$b.log 2>&1; done
timeout -k 10 400 python tools/tune_ineval.py --batch 16 --res 256 --out gpurun_out/ineval_b16_r256.json > gpurun_out/ineval_b16.log 2>&1
timeout -k 10 400 python tools/tune_ineval.py --batch 8 --res 512 --out gpurun_out/ineval_b8_r512.json > gpurun_out/ineval_512.log 2>&1
tail -2 gpurun_out/ineval_b16.log gpurun_out/ineval_512.log; tail -1 gpurun_out/tune5_b*.log
