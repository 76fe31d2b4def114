cd tools/micro
for M in 64 32 16; do for PC in 1 2; do for W in 256 512 1024; do
echo "--- tickets: 1024 blocks, M=$M, $PC workgroups of $W per CU"; timeout 120 ./gather_bench 1024 900000 $W 0 $M $PC | tail -2
done; done; done
