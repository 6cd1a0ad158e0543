for a in 0 1 2 3 4 8 12 15; do echo "ABLATE=$a"; SEMPYR_DEBUG_ABLATE=$a timeout -k 10 120 python tools/bench_conv.py 2>&1 | sed -n '2,3p;12p'; done
