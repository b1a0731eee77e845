#!/bin/bash
# Sanitizer runs of the host-side C code (CPU build only; no GPU involved):
#   tools/san/run.sh <a .fastq.gz> [more files...]
# 1. the DEFLATE decoder against zlib on random streams, corrupted and truncated ones, under ASan + UBSan
# 2. the feeder (plain / gzip / BGZF, 4 threads, three window sizes) under ThreadSanitizer and under ASan + UBSan
set -e
C=$(cd "$(dirname "$0")/../../tatajuba_amd/csrc" && pwd)
T=${TMPDIR:-/tmp}
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=gnu11 -I$C $(dirname "$0")/tji_fuzz.c $C/tj_inflate.c -o $T/tji_fuzz -lz -lpthread
ASAN_OPTIONS=detect_leaks=0 $T/tji_fuzz
[ $# -gt 0 ] || exit 0
gcc -O1 -g -fsanitize=thread -std=gnu11 -I$C $(dirname "$0")/feeder_driver.c $C/feeder.c $C/fastq_reader.c $C/tj_inflate.c -o $T/feeder_tsan -lz -lpthread
$T/feeder_tsan "$@"
gcc -O1 -g -fsanitize=address,undefined -std=gnu11 -I$C $(dirname "$0")/feeder_driver.c $C/feeder.c $C/fastq_reader.c $C/tj_inflate.c -o $T/feeder_asan -lz -lpthread
$T/feeder_asan "$@"
