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
# 1b. entering a deflate stream in the middle (what a one-member .gz is read with on several threads), same sanitizers
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=gnu11 -I$C $(dirname "$0")/tjp_fuzz.c $C/tj_inflate.c -o $T/tjp_fuzz -lz -lpthread
$T/tjp_fuzz
[ $# -gt 0 ] || exit 0
# a one-member gzip file big enough for several rounds of stretches (TATAJUBA_AMD_GZ_STRETCH=65536), next to the caller's files
python3 - "$T/san_one_member.fq.gz" <<'PY'
import gzip, random, sys
r = random.Random(3)
recs = []
for i in range(40000):
    L = r.randrange(60, 200)
    recs.append("@r%d\n%s\n+\n%s\n" % (i, "".join(r.choice("ACGT") for _ in range(L)), "".join(r.choice("FFF:,#") for _ in range(L))))
open(sys.argv[1], "wb").write(gzip.compress("".join(recs).encode(), 6))
PY
export TATAJUBA_AMD_GZ_STRETCH=65536
set -- "$@" "$T/san_one_member.fq.gz"
gcc -O1 -g -fsanitize=thread -std=gnu11 -I$C $(dirname "$0")/feeder_driver.c $C/feeder.c $C/fastq_reader.c $C/tj_inflate.c -o $T/feeder_tsan -lz -lpthread
$T/feeder_tsan "$@"
gcc -O1 -g -fsanitize=address,undefined -std=gnu11 -I$C $(dirname "$0")/feeder_driver.c $C/feeder.c $C/fastq_reader.c $C/tj_inflate.c -o $T/feeder_asan -lz -lpthread
$T/feeder_asan "$@"
