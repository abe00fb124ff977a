#!/bin/bash
# round 4: two half-batch chains - table lines for the half-batch problems, then A/B and a few parity tests
mkdir -p gpurun_out
timeout 1500 python tools/make_tile_table.py --add-missing gpurun_out/tiles_chains.txt > gpurun_out/chains_table.log 2>&1
tail -2 gpurun_out/chains_table.log
export K22_TILE_TABLE=$PWD/gpurun_out/tiles_chains.txt
for c in 2 1 2 1; do
  K22_CHAINS=$c timeout 600 python tools/x3_check.py bf16 2>&1 | grep dtype | cut -c1-560 | sed "s/^/chains=$c /" | tee -a gpurun_out/chains_ab.txt
done
for c in 2 1; do
  K22_CHAINS=$c timeout 600 python tools/x3_check.py f16x3,fp32 2>&1 | grep dtype | cut -c1-560 | sed "s/^/chains=$c /" | tee -a gpurun_out/chains_ab.txt
done
timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/chains_pytest.txt
