#!/bin/bash
# A/B of an environment switch on one box: tools/ab_env.sh VAR scene...   (book1 | grid | hall); arms: VAR unset, VAR=1, unset again
set -e
export RT_CHUNK=1
var=$1; shift
run() {
  for sc in "$@"; do
    case $sc in
      book1) a="3840 500 50 0 book1";; cpu) a="1920 100 50 0 cpu";; grid) a="1920 64 16 0 grid";; hall) a="7680 256 100 0 hall";;
    esac
    timeout -k 10 300 python -u tools/quick_bench.py $a 2>&1 | tail -2 | sed 's/.*kernel/kernel/; s/crc.*rt_/rt_/'
  done
}
echo "== $var unset"; run "$@"
echo "== $var=1"; export $var=1; run "$@"; unset $var
echo "== $var unset"; run "$@"
