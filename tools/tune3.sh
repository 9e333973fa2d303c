set -e
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "every_tile or 160_wide" > gpurun_out/t.log 2>&1
for b in 8 4 2 1 16; do python tools/tune_gemm.py --batch $b --cfgs 14,15,16 --out gpurun_out/tune3_b$b.json > gpurun_out/tune3_b$b.log 2>&1; done
python tools/tune_gemm.py --batch 8 --res 512 --cfgs 14,15,16 --out gpurun_out/tune3_b8_r512.json > gpurun_out/tune3_b8_r512.log 2>&1
python tools/tune_gemm.py --vae --batch 8 --cfgs 14,15,16 --out gpurun_out/tune3_vae_b8.json > gpurun_out/tune3_vae_b8.log 2>&1
python tools/tune_gemm.py --vae --batch 8 --res 512 --cfgs 14,15,16 --out gpurun_out/tune3_vae_b8_r512.json > gpurun_out/tune3_vae_b8_r512.log 2>&1
