for tb in 768 1536; do echo "TARGET_BLOCKS=$tb"; SEMPYR_TARGET_BLOCKS=$tb timeout -k 10 120 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids;  SEMPYR_TARGET_BLOCKS=$tb timeout -k 10 120 python tools/bench_conv.py mid 2>&1 | grep -v "amdgpu.ids\|shape"; done
echo "TARGET_BLOCKS=512 mid";  timeout -k 10 120 python tools/bench_conv.py mid 2>&1 | grep -v "amdgpu.ids\|shape"
