cd tools/micro
echo "--- HBM regime: all blocks, 64 walkers each"; timeout 120 ./gather_bench 2048 900000 64
echo "--- few blocks (MALL-resident 64 x 3.6 MB = 230 MB), 1024 walkers per block, one WG per block"; timeout 120 ./gather_bench 64 900000 1024
echo "--- 32 blocks, 1024 walkers"; timeout 120 ./gather_bench 32 900000 1024
echo "--- 128 blocks, 1024 walkers (460 MB > MALL)"; timeout 120 ./gather_bench 128 900000 1024
echo "--- 256 blocks, 1024 walkers"; timeout 120 ./gather_bench 256 900000 1024
echo "--- mode 2 XCD-affine: 64 blocks, 32 WG x 256 = 8192 walkers per block, 4 passes"; timeout 120 ./gather_bench 64 900000 256 0 4
echo "--- mode 2: 64 blocks, 32 WG x 64 = 2048 walkers, 4 passes"; timeout 120 ./gather_bench 64 900000 64 0 4
echo "--- mode 2: 8 blocks (one per XCD), 32 x 256 walkers, 8 passes"; timeout 120 ./gather_bench 8 900000 256 0 8
echo "--- mode 2: 16 blocks (two per XCD: 7.2 MB per L2), 32 x 256 walkers, 8 passes"; timeout 120 ./gather_bench 16 900000 256 0 8
