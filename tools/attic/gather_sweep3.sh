cd tools/micro
for M in 64 32; do for CFG in "1 1024" "2 512" "2 1024"; do set -- $CFG
echo "--- tickets + segment buffers: 1024 blocks, M=$M, $1 workgroups of $2 per CU"; timeout 120 ./gather_bench 1024 900000 $2 0 $M $1 1 | tail -1
done; done
