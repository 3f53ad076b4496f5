set -e
for pad in 0 64 128 32; do echo "== PAD=$pad"; PAD=$pad timeout -k 10 250 python tools/mb_yardstick.py 2>/dev/null; done
