#!/bin/bash
# AddressSanitizer + UBSan over the host layer (libffmodel.so, the dlrm driver, the rank launcher) with the CPU oracle as the
# kernel library -- the only sanitizer run this pool allows (no GPU ASan).  Builds into gpurun_out/san (scratch, git-ignored):
#   bash tools/sanitize_cpu.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/san
rm -rf "$S"; mkdir -p "$S/dlrm_flexflow_amd" "$S/oracle"
cp -r "$R/dlrm_flexflow_amd/host" "$S/dlrm_flexflow_amd/host"
cp -r "$R/include" "$S/include"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer"
cd "$S/dlrm_flexflow_amd/host"
make clean > /dev/null
sed -i "s/-shared -pthread -o/-shared -pthread $SAN -o/" Makefile
make -j8 CXXFLAGS="-O1 -g -std=c++17 -fPIC -pthread -Wall -Wextra -Wno-unused-parameter $SAN" 2>&1 | grep -E "error|warning" || true
gcc -O1 -g -mavx2 -mfma -ffp-contract=off -fopenmp -fPIC $SAN -shared -o "$S/oracle/libffh_oracle_san.so" "$R/oracle/ffh_oracle.c" -I"$R/include" -lm
export ASAN_OPTIONS=detect_leaks=1:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4
EXE=$S/dlrm_flexflow_amd/host/dlrm
LIB=$S/oracle/libffh_oracle_san.so
run() { echo "== $*"; "$@" > "$S/out.txt" 2> "$S/err.txt" || { echo "FAILED rc=$?"; tail -40 "$S/err.txt"; exit 1; }; grep -E "THROUGHPUT|ERROR|runtime error" "$S/out.txt" "$S/err.txt" | head -5; }
SMALL="-b 64 --arch-sparse-feature-size 8 --arch-embedding-size 100-200-50 --arch-mlp-bot 13-16-8 --arch-mlp-top 32-16-1 --data-size 256 --epochs 2"
run $EXE --backend $LIB $SMALL
run $EXE --backend $LIB $SMALL --arch-interaction-op dot
run $EXE --backend $LIB -b 64 --arch-sparse-feature-size 8 --arch-embedding-size 100-200-50 --arch-mlp-bot 13-16-8 --arch-mlp-top 14-16-1 --data-size 256 --epochs 2 --arch-interaction-op dot-tril
run $EXE --backend $LIB $SMALL --no-trace --profiling
run $EXE --backend $LIB $SMALL --allow-tensor-op-math-conversion --deterministic
run $EXE --backend $LIB $SMALL --export $S/strategy.txt
run $EXE --backend $LIB $SMALL --import $S/strategy.txt
FFM_LAUNCH_DRYRUN=1 run $S/dlrm_flexflow_amd/host/dlrm_testing -ll:gpu 3 $SMALL      # (the -DFFM_TESTING build of the launcher: the dry run is compiled out of dlrm)
echo "sanitizer run clean"
