for w in 0 2 4 6 8 12; do
  if [ $w = 0 ]; then unset RT_BLOCK_WAVES; else export RT_BLOCK_WAVES=$w; fi
  echo "== RT_BLOCK_WAVES=$w"
  RT_TRACE_GEOM=1 timeout -k 10 120 python -u tools/kbench.py teapot_normal.obj 20 2>&1 | grep -v amdgpu | sort -u | grep "shadow=1\|trace geometry" || exit 1
done
