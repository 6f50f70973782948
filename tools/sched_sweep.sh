set -e
for m in teapot_normal.obj water.obj; do
echo "== $m K=1 BW=4 (MAX_BLOCK 1024)"
RT_TILES_PER_WAVE=1 RT_BLOCK_WAVES=4 timeout -k 10 120 python -u tools/kbench.py $m 20 2>&1 | grep -v amdgpu.ids
done
