for a in 0 1 2 3 4 7 8 15 16 20 28; do echo "ABLATE=$a"; SEMPYR_DEBUG_ABLATE=$a timeout -k 10 120 python tools/bench_conv.py 2>&1 | sed -n '3,4p;8p'; done
